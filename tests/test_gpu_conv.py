"""Real-space truncated-kernel propagator on the GPU vs the oracle's restatement of
cnn_propagator/propagation.py:18-133, forward and gradient, incl. the corner-pixel renormalisation, all three detector
modes and the fused rotation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope='module')
def engine_mod():
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import engine
    return engine


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
@pytest.mark.parametrize('ks,probe', [(17, 'plane'), (5, 'random')])
def test_conv_forward_and_gradient(engine_mod, fp, ks, probe):
    rng = np.random.default_rng(3)
    B, Y, X, S = 2, 64, 128, 6
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    if probe == 'plane':
        pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    else:
        pr, pi = 1 + 0.1 * rng.normal(size=(Y, X)), 0.1 * rng.normal(size=(Y, X))
    psize = [1e-7] * 3
    eng = engine_mod.MultisliceEngine(Y, X, S, B, with_grad=True)
    eng.set_physics(5000., 1e-7, fp)
    eng.set_conv(5000., psize, ks)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    wave = eng.forward(B, conv=True)
    ref = orc.multislice_propagate_cnn(delta, beta, pr.astype(np.complex64).real, pi.astype(np.float32), 5000., psize,
                                       kernel_size=ks, free_prop_cm=fp)
    assert rel(np.abs(wave) ** 2, np.abs(ref) ** 2) <= 1e-5
    assert rel(wave, ref) <= 5e-6
    meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
    loss = eng.loss_grad(B, meas, conv=True)
    gd, gb = eng.grad_batch_to_host(B)
    rl, rgd, rgb = orc.cnn_loss_and_grad(delta, beta, pr.astype(np.float32), pi.astype(np.float32), 5000., psize, meas,
                                         kernel_size=ks, free_prop_cm=fp)
    assert abs(loss - rl) <= 2e-5 * abs(rl)
    # The corner-pixel renormalisation (propagation.py:109-110) sends -(sum_j G_j conj(q_j)) / conj(P000) into pixel
    # (0,0,0); with zero-mean residuals that sum cancels to ~1 % of its terms, so float32 resolves it to ~1e-3 only.
    # It is checked separately (and to 1e-6 with a one-signed residual below); everything else to the usual bound.
    tol = 1e-3 if fp == 'inf' else 2e-4
    away = np.ones(gd.shape, dtype=bool)
    away[0, :ks, :ks, :] = False
    assert rel(gd[away], rgd[away]) <= tol and rel(gb[away], rgb[away]) <= tol
    assert rel(gd[~away], rgd[~away]) <= 2e-2 and rel(gb[~away], rgb[~away]) <= 2e-2
    if fp is None:
        meas2 = np.abs(ref) * 1.2
        eng.loss_grad(B, meas2, conv=True)
        gd, gb = eng.grad_batch_to_host(B)
        _, rgd, rgb = orc.cnn_loss_and_grad(delta, beta, pr.astype(np.float32), pi.astype(np.float32), 5000., psize, meas2,
                                            kernel_size=ks, free_prop_cm=fp)
        assert rel(gd, rgd) <= 1e-5 and rel(gb, rgb) <= 1e-5


def test_drop_in_multislice_propagate_cnn(engine_mod):
    from beyond_dof_amd.propagation import multislice_propagate_cnn
    rng = np.random.default_rng(4)
    delta = rng.uniform(0, 2e-6, size=(1, 64, 64, 32))
    beta = 0.1 * delta
    one, zero = np.ones((64, 64)), np.zeros((64, 64))
    w = multislice_propagate_cnn(delta, beta, one, zero, 5000., [1e-7] * 3, kernel_size=17, free_prop_cm=1e-4)
    ref = orc.multislice_propagate_cnn(delta, beta, one, zero, 5000., [1e-7] * 3, kernel_size=17, free_prop_cm=1e-4)
    assert rel(np.abs(w) ** 2, np.abs(ref) ** 2) <= 1e-5
    w2, pa, dt = multislice_propagate_cnn(delta, beta, one, zero, 5000., [1e-7] * 3, kernel_size=17, free_prop_cm=1e-4, debug=True)
    assert np.array_equal(w, w2) and dt >= 0


def test_fullfield_with_conv_propagator(engine_mod):
    """cnn_propagator/fullfield.py:93-106 as written there: rotation gather + conv propagator, gradient to the volume."""
    from beyond_dof_amd.solver import FullfieldSolver
    n, n_theta, mb, fp, ks = 64, 6, 2, 1e-4, 9
    rng = np.random.default_rng(0)
    od = rng.uniform(0, 2e-6, size=(n, n, n))
    ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    idx = np.array([1, 4])
    rot = np.stack([orc.apply_rotation(np.stack([od, ob], axis=3), coords[j]) for j in idx])
    one, zero = np.ones((n, n)), np.zeros((n, n))
    psize = [1e-7] * 3
    ref = orc.multislice_propagate_cnn(rot[..., 0], rot[..., 1], one, zero, 5000., psize, kernel_size=ks, free_prop_cm=fp)
    prj = np.zeros((n_theta, n, n))
    prj[idx] = np.abs(ref) * 1.1
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp, coord_ls=coords, propagator='conv', kernel_size=ks)
    s.set_volume(od, ob)
    s.set_measurements(prj)
    w = s.forward_angles(idx)
    assert rel(np.abs(w) ** 2, np.abs(ref) ** 2) <= 1e-5
    loss = s.loss_and_grad(idx)
    gd, gb = s.gradient_to_host()
    rl, gd_rot, gb_rot = orc.cnn_loss_and_grad(rot[..., 0], rot[..., 1], one, zero, 5000., psize, prj[idx], kernel_size=ks, free_prop_cm=fp)
    rgd = sum(orc.apply_rotation_adjoint(gd_rot[b], coords[j]) for b, j in enumerate(idx))
    rgb = sum(orc.apply_rotation_adjoint(gb_rot[b], coords[j]) for b, j in enumerate(idx))
    assert abs(loss - rl) <= 2e-5 * rl
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4


@pytest.mark.parametrize('name,fp', [('none', None), ('near', 1e-4)])
def test_conv_forward_vs_reference_golden_vector(engine_mod, name, fp):
    """The HIP path against golden vector G9 directly: the reference's own multislice_propagate_cnn on cfg1's object
    (64 x 64 tube phantom, 32 slices, 17 taps; tests/golden/make_golden.py --g9) — 1e-5 on intensities, the north-star bound."""
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    delta = np.load(os.path.join(gdir, 'g2_forward_cfg1.npz'))['delta']
    ref = np.load(os.path.join(gdir, 'g9_conv_propagator.npz'))['wave_cfg1_k17_' + name]
    B, Y, X, S = delta.shape
    eng = engine_mod.MultisliceEngine(Y, X, S, B, with_grad=False)
    eng.set_physics(5000., 1e-7, fp)
    eng.set_conv(5000., [1e-7] * 3, 17)
    eng.set_probe(np.ones((Y, X)), np.zeros((Y, X)))
    eng.set_object_batch(delta, 0.1 * delta)
    wave = eng.forward(B, conv=True)
    assert rel(np.abs(wave) ** 2, np.abs(ref) ** 2) <= 1e-5
    assert rel(wave, ref) <= 5e-6


@pytest.mark.parametrize('adjoint_precision', ['float32', 'first-step'])
def test_reconstruct_fullfield_vs_the_reference_loop(engine_mod, tmp_path, monkeypatch, adjoint_precision):
    """The product's entry point against golden vector G13 DIRECTLY: the reference's own reconstruct_fullfield executed at
    (Y, X, Z) = (64, 64, 64), 4 angles in minibatches of 2, two epochs, real-space propagator with 17 taps, L1 + TV, mask,
    clip (tests/golden/make_golden.py --g13; volumes stored on every second voxel).  Same data file, mask files, initial
    guess and seed in; the float32 HIP path against the float64 loop: every voxel within a hundredth of an Adam step after
    four steps, the volume within the north star's 1e-5."""
    import os
    import sys
    from beyond_dof_amd import h5io, tiffio
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g13_reconstruct_fullfield_64.npz'))
    init_d, init_b = g13_inputs.initial_guess()
    monkeypatch.chdir(tmp_path)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', g['prj'])
    tiffio.write_tiff_stack(g13_inputs.mask(), 'case/fin_sup_mask/mask', dtype='float32', overwrite=True)
    lr = 1e-7
    d, b = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=2, learning_rate=lr, minibatch_size=2, energy_ev=5000,
                                 psize_cm=1e-7, free_prop_cm=1e-4, save_path='case', output_folder='out', initial_guess=[init_d, init_b],
                                 shrink_cycle=None, kernel_size=17, propagator='conv', seed=5, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11,
                                 adjoint_precision=adjoint_precision)
    assert float(g['delta_moved_max']) >= 3.9 * lr                       # four whole steps were taken
    d, b = d[::2, ::2, ::2], b[::2, ::2, ::2]
    dev = np.abs(d - g['delta_sub'])
    devb = np.abs(b - g['beta_sub'])
    stats = (dev.max() / lr, rel(d, g['delta_sub']), devb.max() / lr, rel(b, g['beta_sub']))
    print('G13 stats', adjoint_precision, stats)
    # measured: 0.0016 of a step at worst, delta 5.4e-6, beta 1.2e-5 relative (before the residual of the real-space
    # propagator's loss was split off its carrier, bdof_loss_grad_conv: 0.2 of a step, 5.9e-4)
    assert stats[0] <= 0.01 and stats[1] <= 1e-5, stats                 # the north-star bound on the reconstructed delta
    assert stats[2] <= 0.01 and stats[3] <= 5e-5, stats


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
@pytest.mark.parametrize('ks,probe', [(5, 'random'), (17, 'gaussian')])
def test_float64_real_space_path_vs_oracle(engine_mod, fp, ks, probe):
    """bdof_loss_grad_conv_f64 (csrc/bdof_conv64.h): the real-space propagator's forward + loss + gradient entirely in float64 — pad
    + 'valid' convolution as an overlap-save transform pair on the padded grid, the corner-pixel renormalisation and its adjoint,
    magnitude loss — against the oracle's restatement of cnn_propagator/propagation.py:18-133 + its hand-derived gradient (pinned
    by G9 and by finite differences).  Float64 throughout: the loss to 1e-12; the gradient rows are stored as float32 (6e-8)."""
    rng = np.random.default_rng(21)
    B, N, S = 3, 64, 7
    delta = rng.uniform(0, 2e-5, size=(B, N, N, S))
    beta = 0.1 * delta
    if probe == 'gaussian':
        pr, pi = orc.gaussian_probe((N, N), 40., 40., 0.5)
    else:
        pr, pi = 0.8 + 0.1 * rng.normal(size=(N, N)), 0.1 * rng.normal(size=(N, N))
    ref = orc.multislice_propagate_cnn(delta, beta, pr, pi, 5000., [1e-7] * 3, kernel_size=ks, free_prop_cm=fp)
    meas = (np.abs(ref) * (1 + 0.02 * rng.normal(size=ref.shape))).astype(np.float32).astype(np.float64)     # what the device is handed
    rl, rgd, rgb = orc.cnn_loss_and_grad(delta, beta, pr, pi, 5000., [1e-7] * 3, meas, kernel_size=ks, free_prop_cm=fp)
    eng = engine_mod.MultisliceEngine(N, N, S, B, with_grad=True)
    eng.set_physics(5000., 1e-7, fp)
    eng.set_conv(5000., [1e-7] * 3, ks)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    eng.enable_conv_f64()
    loss = eng.loss_grad(B, meas, conv=True, f64=True)
    gd, gb = eng.grad_batch_to_host(B)
    e = (abs(loss - rl) / rl, rel(gd, rgd), rel(gb, rgb))
    l32 = eng.loss_grad(B, meas, conv=True)
    gd32, gb32 = eng.grad_batch_to_host(B)
    print('float64 real-space path, detector', fp, ks, 'taps:', e, ' float32 kernels:', (abs(l32 - rl) / rl, rel(gd32, rgd), rel(gb32, rgb)))
    # float64 arithmetic; the gradient rows are float32 numbers (6e-8)
    assert e[0] <= 1e-8 and e[1] <= 2e-7 and e[2] <= 2e-7, e


@pytest.mark.parametrize('streams,adjoint_precision', [(None, 'float32'), ('2', 'float32'), (None, 'first-step'), (None, 'float64')])
def test_reconstruct_ptychography_vs_the_reference_loop(engine_mod, tmp_path, monkeypatch, streams, adjoint_precision):
    """(streams = '2': the minibatch split over two streams, as large minibatches run — windows, carrier field and gradient rows
    of the second group at their offsets.)  The ptychography entry point against golden vector G14 directly: the reference's own reconstruct_ptychography executed
    (make_golden.py --g14: (64, 64, 64) object, 64 x 64 wide gaussian probe, 4 positions x 2 angles in minibatches of 2, two
    epochs = 8 Adam steps, real-space propagator with 17 taps, far field; seed 42 through the frozen clock)."""
    import os
    import sys
    from beyond_dof_amd import h5io
    from beyond_dof_amd.ptychography import reconstruct_ptychography
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    if streams:
        monkeypatch.setenv('BDOF_STREAMS', streams)
    g = np.load(os.path.join(gdir, 'g14_reconstruct_ptychography_64.npz'))
    obj_size, psz = tuple(int(v) for v in g['obj_size']), tuple(int(v) for v in g['probe_size'])
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    monkeypatch.chdir(tmp_path)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', g['prj'])
    lr = 2e-7
    d, b = reconstruct_ptychography('data.h5', [tuple(int(v) for v in p) for p in g['probe_pos']], psz, obj_size, theta_st=0, theta_end=2 * np.pi,
                                    n_epochs=2, learning_rate=lr, minibatch_size=2, energy_ev=5000, psize_cm=1e-7, save_path='case',
                                    output_folder='out', initial_guess=[init_d, init_b], probe_type='gaussian', dynamic_dropping=False,
                                    propagator='conv', kernel_size=17, seed=42, probe_mag_sigma=40., probe_phase_sigma=40., probe_phase_max=0.5,
                                    adjoint_precision=adjoint_precision)
    assert float(g['delta_moved_max']) >= 5 * lr                        # the volume moved by several whole steps
    d, b = d[::2, ::2, ::2], b[::2, ::2, ::2]
    dev, devb = np.abs(d - g['delta_sub']), np.abs(b - g['beta_sub'])
    stats = (dev.max() / lr, rel(d, g['delta_sub']), devb.max() / lr, rel(b, g['beta_sub']), float(np.mean(dev > 0.5 * lr)))
    print('G14 stats', adjoint_precision, stats)
    # Round 2 measured delta 7.5e-3 here (a few voxels 3 steps off), beta 7e-4: the whole wave of this wide probe ran through the
    # float32 convolutions and |d| - m was taken in float32.  Round 3 gave the real-space path a carrier FIELD (the probe carried
    # through empty space by the padded convolution in float64, bdof_set_conv_probe_stack) and the float64 residual at the
    # detector: first-minibatch gradient 1.8e-4 -> 4.9e-6, delta after the eight steps 1.6e-5, beta 3.1e-6, no voxel more than
    # 0.009 of a step away — where the transfer-function path stands with float32 adjoint arithmetic (G17: 1.7e-5).
    # Round 4: the float64 real-space path (bdof_loss_grad_conv_f64) for the first minibatch of every epoch — the entry point's
    # default, adjoint_precision='first-step': delta 4.8e-6, beta 1.8e-6 — or throughout ('float64'): 1.2e-7 / 6.6e-8.
    if adjoint_precision == 'float32':
        assert stats[0] <= 0.05 and stats[1] <= 3e-5 and stats[3] <= 1e-5 and stats[4] == 0.0, stats
    else:
        assert stats[0] <= 0.05 and stats[1] <= 1e-5 and stats[3] <= 1e-5 and stats[4] == 0.0, stats


@pytest.mark.parametrize('ks,probe,shape', [(17, 'plane', (128, 128)), (17, 'random', (64, 256)), (9, 'random', (128, 64)),
                                            (5, 'random', (64, 64)), (17, 'random', (512, 512))])
def test_second_tiling_equals_the_first(engine_mod, monkeypatch, ks, probe, shape):
    """k_conv2 (64 x 32 tiles by LDS-DMA, csrc/bdof_conv2.h) keeps k_conv's order of the tap sums; only the epilogue's
    multiply-adds may be contracted differently by the compiler: forward wave, loss and gradient agree to float32 rounding,
    incl. the tiles that touch the field's edge with a non-zero padding constant (cnn_propagator/propagation.py:80-107)."""
    rng = np.random.default_rng(11)
    Y, X = shape
    B, S = (3, 5) if Y < 512 else (2, 3)
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    if probe == 'plane':
        pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    else:
        pr, pi = 0.8 + 0.1 * rng.normal(size=(Y, X)), 0.1 * rng.normal(size=(Y, X))
    out = {}
    for tiling in ('1', '2', '2s'):          # '2s': the second tiling with the batch split over two streams (as large batches run)
        monkeypatch.setenv('BDOF_CONV_TILING', tiling[0])
        monkeypatch.setenv('BDOF_STREAMS', '2' if tiling == '2s' else '1')
        eng = engine_mod.MultisliceEngine(Y, X, S, B, with_grad=True)
        eng.set_physics(5000., 1e-7, 1e-4)
        eng.set_conv(5000., [1e-7] * 3, ks)
        eng.set_probe(pr, pi)
        eng.set_object_batch(delta, beta)
        wave = eng.forward(B, conv=True)
        meas = np.abs(wave) * (1 + 0.05 * np.random.default_rng(5).normal(size=wave.shape))
        if tiling == '1':
            meas1 = meas
        loss = eng.loss_grad(B, meas1, conv=True)
        gd, gb = eng.grad_batch_to_host(B)
        out[tiling] = (wave, loss, gd, gb)
        del eng
    w1, l1, gd1, gb1 = out['1']
    assert np.isfinite(w1).all() and np.abs(w1).max() > 0
    for tiling in ('2',):
        w2, l2, gd2, gb2 = out[tiling]
        assert np.abs(w1 - w2).max() <= 4e-7 * np.abs(w1).max() and rel(w2, w1) <= 1e-7
        assert abs(l1 - l2) <= 1e-6 * abs(l1)
        assert rel(gd2, gd1) <= 2e-6 and rel(gb2, gb1) <= 2e-6
        assert np.abs(gd1 - gd2).max() <= 1e-5 * np.abs(gd1).max()
    # sub-batches on two streams: the same kernels on the same fields — the same bits
    for a, b in zip(out['2'], out['2s']):
        assert np.array_equal(a, b)
