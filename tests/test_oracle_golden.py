"""The CPU oracle (oracle/bdof_oracle.py) replayed against vectors captured from the
reference's own functions (tests/golden/make_golden.py).  float64: tolerances are
round-off only."""
import os

import numpy as np
import pytest

from oracle import bdof_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_get_kernel(golden_dir):
    g = _load(golden_dir, 'g1_get_kernel.npz')
    for key in g.files:
        _, Y, X, dist = key.split('_')
        H = orc.get_kernel(float(dist), 0.248, [1., 1., 1.], [int(Y), int(X), 4])
        assert H.shape == (int(Y), int(X))
        np.testing.assert_allclose(H, g[key], rtol=0, atol=1e-13)


@pytest.mark.parametrize('name,fp', [('none', None), ('near', 1e-4), ('inf', 'inf')])
def test_g2_forward_16(golden_dir, name, fp):
    g = _load(golden_dir, 'g2_forward_16.npz')
    w, pa = orc.multislice_propagate_batch_numpy(g['delta'], g['beta'], np.ones((16, 16)), np.zeros((16, 16)),
                                                 5000., 1e-7, free_prop_cm=fp, obj_batch_shape=g['delta'].shape)
    ref = g['wave_' + name]
    np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    if name == 'none':
        np.testing.assert_allclose(pa, g['probe_array'], rtol=0, atol=1e-12)


def test_g2_forward_16_probe(golden_dir):
    g = _load(golden_dir, 'g2_forward_16.npz')
    w, _ = orc.multislice_propagate_batch_numpy(g['delta'], g['beta'], g['probe_real'], g['probe_imag'], 5000., 1e-7,
                                                free_prop_cm=1e-4, obj_batch_shape=g['delta'].shape)
    np.testing.assert_allclose(w, g['wave_near_probe'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('name,fp', [('none', None), ('near', 1e-4), ('inf', 'inf')])
def test_g2_forward_cfg1(golden_dir, name, fp):
    """cfg1: 64^3 tube phantom, 1 angle, 32 slices (BASELINE.json configs[0])."""
    g = _load(golden_dir, 'g2_forward_cfg1.npz')
    delta = g['delta']
    w, pa = orc.multislice_propagate_batch_numpy(delta, 0.1 * delta, np.ones((64, 64)), np.zeros((64, 64)), 5000.,
                                                 1e-7, free_prop_cm=fp, obj_batch_shape=delta.shape)
    ref = g['wave_' + name]
    np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    if name == 'none':
        np.testing.assert_allclose(np.abs(pa).sum(axis=(1, 2, 3)), g['probe_array_abs_sum'], rtol=1e-13)
        np.testing.assert_allclose(pa[-1], g['probe_array_last'], rtol=0, atol=1e-12)


def test_g3_rotation(golden_dir):
    g = _load(golden_dir, 'g3_rotation.npz')
    for size, n in [((8, 8, 8), 5), ((64, 64, 64), 4), ((6, 10, 10), 7)]:
        key = 'x'.join(map(str, size)) + '_n{}'.format(n)
        coords = orc.rotation_lookup(list(size), n)
        assert np.array_equal(np.stack(coords), g['coords_' + key])        # integer tables: bit exact
        if 'obj_' + key in g.files:
            obj = g['obj_' + key]
            rot = np.stack([orc.apply_rotation(obj, c) for c in coords])
            assert np.array_equal(rot, g['rot_' + key])


def test_g4_adam(golden_dir):
    g = _load(golden_dir, 'g4_adam.npz')
    x = g['x0']
    m = v = None
    for it in range(3):
        x, m, v = orc.apply_gradient_adam(x, g['g{}'.format(it)], it, m, v, step_size=1e-7)
        np.testing.assert_allclose(x, g['x{}'.format(it + 1)], rtol=1e-14)
        np.testing.assert_allclose(m, g['m{}'.format(it + 1)], rtol=1e-14)
        np.testing.assert_allclose(v, g['v{}'.format(it + 1)], rtol=1e-14)


def test_g5_tv(golden_dir):
    g = _load(golden_dir, 'g5_tv.npz')
    np.testing.assert_allclose(orc.total_variation_3d(g['arr']), g['tv'], rtol=1e-14)


def test_g6_split_tasks(golden_dir):
    g = _load(golden_dir, 'g6_split_tasks.npz')
    parts = orc.split_tasks(g['arr'], int(g['split_size']))
    assert [len(p) for p in parts] == list(g['lengths'])
    assert np.array_equal(np.concatenate(parts), g['concat'])


def test_g8_kernel_ir_and_upsample(golden_dir):
    """get_kernel_ir (cnn_propagator/util.py:105-127) and upsample_2x (:350-360): oracle and host code against the
    reference's own outputs."""
    from beyond_dof_amd import util
    g = np.load(os.path.join(golden_dir, 'g8_kernel_ir_upsample.npz'))
    n = 0
    for key in g.files:
        if not key.startswith('Hir_'):
            continue
        _, Y, X, dist = key.split('_')
        for fn in (orc.get_kernel_ir, util.get_kernel_ir):
            h = fn(float(dist), 0.248, [1., 1., 1.], [int(Y), int(X), 4])
            np.testing.assert_allclose(h, g[key], rtol=1e-12, atol=1e-12 * np.abs(g[key]).max())
        n += 1
    assert n == 6
    for tag in ('3', '4'):
        for fn in (orc.upsample_2x, util.upsample_2x):
            np.testing.assert_allclose(fn(g['up_in' + tag]), g['up_out' + tag], rtol=0, atol=1e-15)


def test_g9_real_space_propagator(golden_dir):
    """multislice_propagate_cnn (cnn_propagator/propagation.py:18-133) executed from the reference's own file (its
    autograd.scipy.signal.convolve stood in by scipy.signal.convolve2d, see make_golden.py): the oracle's restatement
    reproduces kernel construction and crop, the running padding constant, the slice loop, the corner renormalisation and
    the detector step to round-off."""
    g = np.load(os.path.join(golden_dir, 'g9_conv_propagator.npz'))
    d, b = g['delta16'], g['beta16']
    one, zero = np.ones(d.shape[1:3]), np.zeros(d.shape[1:3])
    for name, fp in [('none', None), ('near', 1e-4), ('inf', 'inf')]:
        w = orc.multislice_propagate_cnn(d, b, one, zero, 5000., [1e-7] * 3, kernel_size=5, free_prop_cm=fp)
        ref = g['wave16_k5_' + name]
        np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    w = orc.multislice_propagate_cnn(d, b, g['probe_real16'], g['probe_imag16'], 5000., [1e-7] * 3, kernel_size=9, free_prop_cm=1e-4)
    np.testing.assert_allclose(w, g['wave16_k9_probe_near'], rtol=0, atol=1e-12 * np.abs(g['wave16_k9_probe_near']).max())
    # cfg1's object, the entry points' default kernel_size = 17.  No file of the phantom travels: delta is rebuilt from G2b
    gd = np.load(os.path.join(golden_dir, 'g2_forward_cfg1.npz'))['delta']
    for name, fp in [('none', None), ('near', 1e-4)]:
        w = orc.multislice_propagate_cnn(gd, 0.1 * gd, np.ones((64, 64)), np.zeros((64, 64)), 5000., [1e-7] * 3, kernel_size=17, free_prop_cm=fp)
        ref = g['wave_cfg1_k17_' + name]
        np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def _oracle_conv_loop(g, n_theta, mb, n_epochs, seed, lr, kernel_size, fp, shrink_cycle=None, **reg):
    """The loop of cnn_propagator/fullfield.py:337-362 driven by the oracle's pieces (real-space propagator)."""
    from beyond_dof_amd.comm import minibatch_schedule
    n = g['mask'].shape[0]
    mask = g['mask'].astype(np.float64)
    coords = orc.rotation_lookup([n, n, n], n_theta)
    sched = minibatch_schedule(n_theta, 1, mb, rng=np.random.RandomState(seed))
    one, zero = np.ones((n, n)), np.zeros((n, n))
    x = np.clip(np.array([g['init_delta'] * mask, g['init_beta'] * mask]), 0, None)
    prj = g['prj']
    for i_epoch in range(n_epochs):
        m = v = None
        for i_batch, idx in enumerate(sched):
            rot = np.stack([orc.apply_rotation(np.stack([x[0], x[1]], axis=3), coords[j]) for j in idx])
            _, gd_rot, gb_rot = orc.cnn_loss_and_grad(rot[..., 0], rot[..., 1], one, zero, 5000., [1e-7] * 3, np.abs(prj[idx]),
                                                      kernel_size=kernel_size, free_prop_cm=fp)
            gd = sum(orc.apply_rotation_adjoint(gd_rot[b], coords[j]) for b, j in enumerate(idx))
            gb = sum(orc.apply_rotation_adjoint(gb_rot[b], coords[j]) for b, j in enumerate(idx))
            rd, rb = orc.regularizer_grad(x[0], x[1], **reg)
            x, m, v = orc.apply_gradient_adam(x, np.array([gd + rd, gb + rb]), i_batch, m, v, step_size=lr)
            x = np.clip(x * mask, 0, None)
            if shrink_cycle is not None and i_epoch >= shrink_cycle:             # shrink wrap, fullfield.py:369-372
                mask = mask * (x[0] > 1e-15)
    return x


@pytest.mark.parametrize('tag,n_epochs,seed,reg', [('a', 2, 7, dict(alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)),
                                                   ('b', 1, 3, dict(alpha=1e-8, alpha_d=None, alpha_b=None, gamma=0.)),
                                                   ('c', 2, 11, dict(alpha_d=4e-6, alpha_b=1.5e-9, gamma=0., shrink_cycle=0))])
def test_g10_reconstruct_fullfield_loop(golden_dir, tag, n_epochs, seed, reg):
    """The reference's own reconstruct_fullfield executed on 8^3 (make_golden.py --g10: file traffic in memory, autograd.grad
    stood in by float64 central differences of the reference's calculate_loss).  The oracle-driven loop — the one the GPU
    tests compare the product with — lands on the same volume: schedule from the seed, per-epoch Adam restart, bias-correction
    exponent, regulariser branch (b: quirk Q6), mask and clip order, shrink-wrap (c), analytic gradient."""
    g = np.load(os.path.join(golden_dir, 'g10_reconstruct_fullfield.npz'))
    lr = 1e-7
    x = _oracle_conv_loop(g, 4, 2, n_epochs, seed, lr, 5, 1e-4, **reg)
    for got, key in ((x[0], 'delta_' + tag), (x[1], 'beta_' + tag)):
        want = g[key]
        moved = np.abs(want - np.clip(g['init_' + key.split('_')[0]] * g['mask'], 0, None)).max()
        assert moved >= 0.9 * lr                                         # the volume did move by whole steps
        # finite differences vs analytic gradient: a thousandth of a step at worst (measured 3e-4 for beta in case b, where no
        # regulariser term lifts small gradients above the differences' round-off), typically 1e-6
        assert np.abs(got - want).max() <= 1e-3 * lr, (key, np.abs(got - want).max() / lr)
        assert np.sqrt(np.mean((got - want) ** 2)) <= 2e-5 * lr


def test_g11_reconstruct_ptychography_loop(golden_dir):
    """The reference's own reconstruct_ptychography executed (make_golden.py --g11: 10^3 object, 18 x 18 gaussian probe,
    3 positions x 2 angles, minibatches of 2, two epochs, seed 42 through the frozen clock; autograd.grad stood in by
    central differences).  The product's schedule (ptychography.epoch_schedule / batches_of_epoch) driving the oracle's
    window cut, real-space forward, gradient, Adam and clip lands on the reference's volume — including the quirks of
    the padded position lists and of the batch count taken from the UNPADDED number of spots (:257-260,269-282)."""
    from beyond_dof_amd.ptychography import batches_of_epoch
    g = np.load(os.path.join(golden_dir, 'g11_reconstruct_ptychography.npz'))
    n, n_theta, mb, lr = g['init_delta'].shape[0], g['prj'].shape[0], 2, 2e-7
    pos, psz = g['probe_pos'], tuple(int(v) for v in g['probe_size'])
    coords = orc.rotation_lookup([n, n, n], n_theta)
    pr, pi_ = orc.gaussian_probe(psz, 4., 4., 0.5)
    x = np.array([g['init_delta'], g['init_beta']])
    rng = np.random.RandomState(42)
    for _ in range(2):
        m = v = None
        for i_batch, (i_theta, ind) in enumerate(batches_of_epoch(n_theta, len(pos), mb, 1, 0, rng)):
            _, gd, gb = orc.ptycho_loss_and_grad(x[0], x[1], coords[i_theta], pos, pos[ind], g['prj'][i_theta, ind], pr, pi_, psz,
                                                 5000., 1e-7, propagator='conv', kernel_size=17)
            x, m, v = orc.apply_gradient_adam(x, np.array([gd, gb]), i_batch, m, v, step_size=lr)
            x = np.clip(x, 0, None)
    for got, key in ((x[0], 'delta'), (x[1], 'beta')):
        assert np.abs(g[key] - g['init_' + key]).max() >= 0.9 * lr
        assert np.abs(got - g[key]).max() <= 1e-3 * lr, (key, np.abs(got - g[key]).max() / lr)
        assert np.sqrt(np.mean((got - g[key]) ** 2)) <= 2e-5 * lr


def test_g10_gradient_of_the_reference_loss(golden_dir):
    """The gradient itself: float64 central differences of the REFERENCE's calculate_loss (real-space forward, rotation, loss,
    L1 + TV; captured at the first minibatch of G10's case a) against the oracle's analytic adjoint — what stands in for
    autograd.grad everywhere else."""
    g = np.load(os.path.join(golden_dir, 'g10_reconstruct_fullfield.npz'))
    n = g['mask'].shape[0]
    coords = orc.rotation_lookup([n, n, n], 4)
    idx = g['grad0_ind']
    d, b = g['grad0_delta_in'], g['grad0_beta_in']
    rot = np.stack([orc.apply_rotation(np.stack([d, b], axis=3), coords[j]) for j in idx])
    one, zero = np.ones((n, n)), np.zeros((n, n))
    _, gd_rot, gb_rot = orc.cnn_loss_and_grad(rot[..., 0], rot[..., 1], one, zero, 5000., [1e-7] * 3, np.abs(g['prj'][idx]), kernel_size=5,
                                              free_prop_cm=1e-4)
    gd = sum(orc.apply_rotation_adjoint(gd_rot[k], coords[j]) for k, j in enumerate(idx))
    gb = sum(orc.apply_rotation_adjoint(gb_rot[k], coords[j]) for k, j in enumerate(idx))
    rd, rb = orc.regularizer_grad(d, b, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    assert rel(gd + rd, g['grad0_gd']) <= 1e-6 and rel(gb + rb, g['grad0_gb']) <= 1e-6, (rel(gd + rd, g['grad0_gd']), rel(gb + rb, g['grad0_gb']))


def _g20_setup(golden_dir):
    import sys
    sys.path.insert(0, golden_dir)
    import g13_inputs
    g = np.load(os.path.join(golden_dir, 'g20_directional_derivatives_64.npz'))
    shape = (64, 64, 64)
    mask = g13_inputs.mask(shape)
    init_d, init_b = g13_inputs.initial_guess(shape)
    return g, init_d * mask, init_b * mask, g13_inputs.g20_directions(shape), orc.rotation_lookup(list(shape), 4)


def test_g20_gradient_of_the_reference_loss_at_64_cubed(golden_dir):
    """The oracle's analytic gradient against central differences of the REFERENCE's own calculate_loss along six directions at
    (64, 64, 64) — a size the HIP kernels take — for both forward models the reference has (np_funcs' FFT propagator and the
    real-space propagator its entry point calls): golden vector G20, numbers that no oracle code produced."""
    g, d, b, dirs, coords = _g20_setup(golden_dir)
    one, zero = np.ones((64, 64)), np.zeros((64, 64))
    for model, fixture in (('fft', 'g15_reconstruct_fullfield_fft_64.npz'), ('conv', 'g13_reconstruct_fullfield_64.npz')):
        prj = np.load(os.path.join(golden_dir, fixture))['prj']
        ind = g[model + '_ind']
        if model == 'fft':
            loss, gd, gb = orc.fullfield_loss_and_grad(d, b, coords, ind, prj[ind], one, zero, 5000., 1e-7, free_prop_cm=1e-4, with_reg=False)
        else:
            rot = np.stack([orc.apply_rotation(np.stack([d, b], axis=3), coords[j]) for j in ind])
            loss, gd_rot, gb_rot = orc.cnn_loss_and_grad(rot[..., 0], rot[..., 1], one, zero, 5000., [1e-7] * 3, np.abs(prj[ind]), kernel_size=17,
                                                         free_prop_cm=1e-4)
            gd = sum(orc.apply_rotation_adjoint(gd_rot[k], coords[j]) for k, j in enumerate(ind))
            gb = sum(orc.apply_rotation_adjoint(gb_rot[k], coords[j]) for k, j in enumerate(ind))
        assert abs(loss - float(g[model + '_loss'])) <= 1e-10 * abs(loss)
        got = np.array([[np.sum(gd * v), np.sum(gb * v)] for v in dirs])
        ref = g[model + '_dd']
        assert np.max(np.abs(got - ref) / np.abs(ref)) <= 1e-6, (model, got, ref)


@pytest.mark.parametrize('model', ['fft', 'conv'])
def test_g21_gradient_of_the_reference_ptychography_loss_at_64_cubed(golden_dir, model):
    """G20 for ptychography: central differences of the REFERENCE's calculate_loss (cnn_propagator/ptychography.py:30-81 —
    rotation, padding, window cut, forward, far-field loss) along six directions at the first minibatch of the G17 (FFT forward
    bound in) and G14 (real-space propagator) runs, against the oracle's analytic gradient: numbers no oracle code produced."""
    import sys
    sys.path.insert(0, golden_dir)
    import g13_inputs
    g = np.load(os.path.join(golden_dir, 'g21_ptycho_directional_derivatives_64.npz'))
    f = np.load(os.path.join(golden_dir, 'g17_reconstruct_ptychography_fft_64.npz' if model == 'fft' else 'g14_reconstruct_ptychography_64.npz'))
    obj_size, psz, sigma = tuple(int(v) for v in f['obj_size']), tuple(int(v) for v in f['probe_size']), float(f['probe_sigma'])
    d, b = g13_inputs.initial_guess(obj_size)
    coords = orc.rotation_lookup(list(obj_size), f['prj'].shape[0])
    pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
    i_theta, batch = int(g[model + '_i_theta']), g[model + '_pos_batch']
    ind = [int(np.where((f['probe_pos'] == p).all(axis=1))[0][0]) for p in batch]
    loss, gd, gb = orc.ptycho_loss_and_grad(d.astype(np.float64), b.astype(np.float64), coords[i_theta], f['probe_pos'], batch, f['prj'][i_theta, ind],
                                            pr, pi_, psz, 5000., 1e-7, propagator=model, kernel_size=17)
    assert abs(loss - float(g[model + '_loss'])) <= 1e-10 * abs(loss)
    got = np.array([[np.sum(gd * v), np.sum(gb * v)] for v in g13_inputs.g20_directions(obj_size)])
    ref = g[model + '_dd']
    # 5e-6: the differences' own floor here (kinks of |psi| at dark far-field bins against round-off; make_golden.py fd_directional)
    assert np.max(np.abs(got - ref) / np.abs(ref)) <= 5e-6, (model, np.abs(got - ref) / np.abs(ref))
