"""End-to-end: reconstruct_fullfield (HDF5 in, TIFF out, Adam loop on the GPU) against an oracle-driven loop with
the same schedule.  Adam's update is sign-like in its first steps (m_hat/sqrt(v_hat) = +-1), so voxels whose
gradient lies within float32 noise of zero may move by up to 2*lr differently; the comparison therefore bounds the
fraction of such voxels and the L2 distance, and checks the loss, instead of asking for 1e-5 per voxel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def _phantom(n, rng):
    z, y, x = np.mgrid[:n, :n, :n]
    d = np.zeros((n, n, n))
    for _ in range(6):
        c = rng.uniform(n * 0.3, n * 0.7, size=3)
        r = rng.uniform(n * 0.08, n * 0.2)
        d += 1e-6 * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * r ** 2))
    return d, 0.1 * d


def test_reconstruct_fullfield_end_to_end(tmp_path, monkeypatch):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io, tiffio
    from beyond_dof_amd.comm import minibatch_schedule
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    monkeypatch.chdir(tmp_path)                                  # rotation tables are written to the CWD like the reference
    rng = np.random.default_rng(0)
    n, n_theta, mb, fp = 64, 8, 4, 1e-4
    true_d, true_b = _phantom(n, rng)
    coords = orc.rotation_lookup([n, n, n], n_theta)
    one, zero = np.ones((n, n)), np.zeros((n, n))
    rot = np.stack([orc.apply_rotation(np.stack([true_d, true_b], axis=3), c) for c in coords])
    prj, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, fp, rot[..., 0].shape,
                                                  return_probe_array=False)
    prj = prj.astype(np.complex64)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', prj)
    mask = np.ones((n, n, n), dtype=np.float32)
    mask[:4] = 0
    tiffio.write_tiff_stack(mask, 'case/fin_sup_mask/mask', dtype='float32', overwrite=True)
    init_d = np.clip(rng.normal(8.7e-7, 1e-7, size=(n, n, n)), 0, None)
    init_b = np.clip(rng.normal(5.1e-8, 1e-8, size=(n, n, n)), 0, None)
    kw = dict(alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    lr = 1e-7
    d, b = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=1, learning_rate=lr, minibatch_size=mb,
                                 energy_ev=5000, psize_cm=1e-7, free_prop_cm=fp, save_path='case', output_folder='out',
                                 initial_guess=[init_d, init_b], shrink_cycle=None, seed=7, center=32, unknown_key=1, **kw)
    out = os.path.join('case', 'out')
    assert os.path.exists(os.path.join(out, 'summary.txt'))
    assert np.array_equal(tiffio.read_tiff(os.path.join(out, 'delta_ds_1.tiff')), d.astype(np.float32))
    assert tiffio.read_tiff(os.path.join(out, 'beta_ds_1.tiff')).shape == (n, n, n)
    assert np.all(d[:4] == 0) and np.all(d >= 0)                 # finite support + non-negativity

    # oracle loop, same schedule (cnn_propagator/fullfield.py:337-362)
    x = np.array([init_d * mask, init_b * mask])
    sched = minibatch_schedule(n_theta, 1, mb, rng=np.random.RandomState(7))
    m = v = None
    losses = []
    for i_batch, idx in enumerate(sched):
        loss, g1, g2 = orc.fullfield_loss_and_grad(x[0], x[1], coords, idx, prj[idx], one, zero, 5000., 1e-7, free_prop_cm=fp,
                                                   with_reg=True, **kw)
        losses.append(loss)
        x, m, v = orc.apply_gradient_adam(x, np.array([g1, g2]), i_batch, m, v, step_size=lr)
        x = np.clip(x * mask, 0, None)
    diff = np.abs(d - x[0])
    assert np.mean(diff > 0.05 * lr) < 2e-3                      # almost every voxel took the same (sign-like) step
    assert diff.max() <= 2.5 * lr * len(sched)
    assert np.linalg.norm(d - x[0]) <= 2e-3 * np.linalg.norm(x[0])
    assert np.linalg.norm(b - x[1]) <= 2e-2 * np.linalg.norm(x[1])

    # a second run with more epochs lowers the data-term loss (and exercises 'auto' stop + shrink-wrap + no mask files)
    from beyond_dof_amd.solver import FullfieldSolver
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp, coord_ls=coords)
    s.set_measurements(np.abs(prj))
    s.set_volume(init_d * mask, init_b * mask)
    l0 = s.loss_and_grad(sched[0])
    d2, b2 = reconstruct_fullfield('data.h5', n_epochs='auto', max_nepochs=3, crit_conv_rate=1e-9, learning_rate=lr,
                                   minibatch_size=mb, energy_ev=5000, psize_cm=1e-7, free_prop_cm=fp, save_path='case',
                                   output_folder='out2', initial_guess=[init_d, init_b], shrink_cycle=1, seed=7,
                                   save_intermediate=True, debug=True, **kw)
    s.set_volume(d2, b2)
    l1 = s.loss_and_grad(sched[0])
    assert l1 < l0
    assert os.path.exists(os.path.join('case', 'out2', 'intermediate', 'current.tiff'))
    # debug=True: |exit waves| of every tenth minibatch (cnn_propagator/fullfield.py:372-374) — here minibatch 0 of each epoch;
    # the last dump is the forward model of the volume after the first step of the last epoch
    dumps = sorted(os.listdir(os.path.join('case', 'out2', 'exits')))
    assert dumps and all(name.endswith('-0.tiff') for name in dumps), dumps
    ex = tiffio.read_tiff(os.path.join('case', 'out2', 'exits', dumps[0]))
    assert ex.shape == (mb, n, n) and np.all(np.isfinite(ex)) and abs(ex.mean() - 1) < 0.1


def test_reconstruct_fullfield_optimizable_probe_and_accumulation(tmp_path, monkeypatch):
    """probe_type='optimizable' with probe_learning_rate and pupil_function, and n_batch_per_update accumulation
    (tensorflow_recon/fullfield.py:311-327,413-425,442-455) through the entry point: files out, probe moved, loss down."""
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io, tiffio
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(3)
    n, n_theta, mb, fp = 64, 8, 2, 1e-4
    true_d, true_b = _phantom(n, rng)
    coords = orc.rotation_lookup([n, n, n], n_theta)
    one, zero = np.ones((n, n)), np.zeros((n, n))
    rot = np.stack([orc.apply_rotation(np.stack([true_d, true_b], axis=3), c) for c in coords])
    prj, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, fp, rot[..., 0].shape,
                                                  return_probe_array=False)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', prj.astype(np.complex64))
    init_d = np.clip(rng.normal(8.7e-7, 1e-7, size=(n, n, n)), 0, None)
    init_b = np.clip(rng.normal(5.1e-8, 1e-8, size=(n, n, n)), 0, None)
    mag0 = 1 + 0.05 * rng.normal(size=(n, n))
    pupil = np.ones((n, n))
    pupil[:2] = 0
    common = dict(theta_st=0, theta_end=2 * np.pi, learning_rate=1e-7, minibatch_size=mb, energy_ev=5000, psize_cm=1e-7, free_prop_cm=fp,
                  save_path='case', initial_guess=[init_d, init_b], shrink_cycle=None, seed=7, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    d, b = reconstruct_fullfield('data.h5', n_epochs=2, output_folder='out_p', probe_type='optimizable', probe_initial=[mag0, np.zeros((n, n))],
                                 probe_learning_rate=2e-3, pupil_function=pupil, **common)
    mag = tiffio.read_tiff(os.path.join('case', 'out_p', 'probe_mag_ds_1.tiff'))
    assert mag.shape == (n, n) and np.all(mag[:2] == 0)                       # pupil enforced
    assert np.abs(mag[2:] - mag0[2:]).max() > 1e-3                            # the probe moved ...
    assert np.abs(mag[2:] - 1).mean() < np.abs(mag0[2:] - 1).mean()           # ... towards the true (unit) probe
    assert os.path.exists(os.path.join('case', 'out_p', 'probe_phase_ds_1.tiff'))
    # accumulation: 4 minibatches per epoch, applied every 2 -> the same as minibatches of twice the size (mean of means)
    d1, b1 = reconstruct_fullfield('data.h5', n_epochs=1, output_folder='out_a', n_batch_per_update=2, accumulate_gradients=True,
                                   random_theta=False, **common)
    common2 = dict(common, minibatch_size=2 * mb)
    d2, b2 = reconstruct_fullfield('data.h5', n_epochs=1, output_folder='out_b', random_theta=False, **common2)
    assert np.mean(np.abs(d1 - d2) > 0.05 * 1e-7) < 2e-3 and np.linalg.norm(d1 - d2) <= 1e-3 * np.linalg.norm(d2)


class _LoopbackComm(object):
    """One rank that still runs the exchange branch of the solver's tail (slab loop, sharded Adam, all-gather calls) with
    collectives that are the identity — what they are on a single rank."""
    size, rank, local_rank, always_reduce, sharded, backend = 1, 0, 0, True, True, 'loopback'

    def __init__(self):
        self.calls = []

    def attach(self, ctx):
        pass

    def Barrier(self):
        pass

    def allreduce_max_host(self, arr):
        return arr

    def start_allreduce(self, ctx, buf, lo, hi):
        self.calls.append(('ar', lo, hi))

    def start_reduce_scatter(self, ctx, buf, lo, per):
        self.calls.append(('rs', lo, per))

    def start_allgather(self, ctx, buf, lo, per):
        self.calls.append(('ag', lo, per))

    def wait(self, ctx, ticket):
        pass


def _two_steps(solver, sched, **kw):
    solver.reset_moments()
    for i, idx in enumerate(sched):
        solver.step(i, idx, 1e-7, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, **kw)
    return solver.get_volume()


def _slab_case(comm):
    from beyond_dof_amd.solver import FullfieldSolver
    rng = np.random.default_rng(1)
    n, n_theta, mb = 64, 8, 4
    coords = orc.rotation_lookup([n, n, n], n_theta)
    meas = 1 + 0.05 * rng.normal(size=(n_theta, n, n))
    init_d = np.clip(rng.normal(8.7e-7, 1e-7, size=(n, n, n)), 0, None)
    sched = [np.arange(0, 4), np.arange(4, 8)]
    vols = []
    for n_slabs, sharded, c in ((1, False, None), (5, False, comm), (64, False, comm), (8, True, comm), (1, True, comm)):
        s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, coord_ls=coords, comm=c)
        s.set_measurements(meas)
        s.set_volume(init_d, 0.1 * init_d)
        vols.append(_two_steps(s, sched, n_slabs=n_slabs, sharded=sharded))
    for v in vols[1:]:
        assert np.array_equal(v[0], vols[0][0]) and np.array_equal(v[1], vols[0][1])
    assert not np.array_equal(vols[0][0], init_d)
    # the tuner times the candidate plans of the tail in a dry run: it must pick one of them and leave the volume alone
    before = s.get_volume()
    n_calls = len(getattr(comm, 'calls', []))
    plan = s.tune_tail(candidates=(1, 8))
    assert plan in ((1, True), (8, True)) and len(s.tuned) == 2 and s.tail_plan() == plan
    after = s.get_volume()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    v2 = _two_steps(s, sched)                        # and step() then runs the tuned plan
    s2 = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, coord_ls=coords)
    s2.set_measurements(meas)
    s2.set_volume(before[0], before[1])
    v1 = _two_steps(s2, sched)
    assert np.array_equal(v1[0], v2[0]) and np.array_equal(v1[1], v2[1])
    # the shard layout of the Adam moments is frozen for an epoch: a different sharded plan, the all-reduce form or a
    # whole-volume adam_update inside it must be refused, not run on stale moments (round-2 advice)
    s.reset_moments()
    s.step(0, sched[0], 1e-7, n_slabs=8, sharded=True)
    with pytest.raises(RuntimeError, match='exchange plan changed'):
        s.step(1, sched[1], 1e-7, n_slabs=4, sharded=True)
    with pytest.raises(RuntimeError, match='exchange plan changed'):
        s.step(1, sched[1], 1e-7, n_slabs=8, sharded=False)
    with pytest.raises(RuntimeError, match='adam_update'):
        s.adam_update(1, 1e-7)
    g_sharded = s.gradient_to_host()                 # gathers the parts first (identity on one rank)
    s.reset_moments()
    s.step(0, sched[1], 1e-7, n_slabs=4, sharded=False)      # fine again after the reset
    if hasattr(comm, 'calls'):
        del comm.calls[n_calls:]


def test_slab_pipelined_step_is_bit_identical():
    """rotation adjoint -> exchange -> Adam, slab by slab, in the all-reduce and in the reduce-scatter / sharded-Adam /
    all-gather form (FullfieldSolver.step with more than one rank), gives the very same volume as the whole-volume
    sequence (loop-back comm: the slab kernels, their ordering and the ranges handed to the collectives)."""
    import __graft_entry__ as entry
    entry.build()
    comm = _LoopbackComm()
    _slab_case(comm)
    per_x = 64 * 64 * 2
    kinds = [c[0] for c in comm.calls]
    assert kinds.count('ar') == 2 * (5 + 64) and kinds.count('rs') == 2 * (8 + 1) and kinds.count('ag') == 2 * (8 + 1)
    ar = [c for c in comm.calls if c[0] == 'ar'][:5]
    assert ar[0][1] == 0 and ar[-1][2] == 64 * per_x and all(ar[i][2] == ar[i + 1][1] for i in range(4))      # slabs tile the volume
    rs = [c for c in comm.calls if c[0] == 'rs'][:8]
    assert [c[1] for c in rs] == [i * 8 * per_x for i in range(8)] and all(c[2] == 8 * per_x for c in rs)


def test_slab_pipelined_step_through_rccl():
    """The same through the library's own RCCL communicator on one rank (bdof_comm_*: ncclCommInitRank with a unique id,
    reduce-scatter / all-gather / all-reduce per slab on the communicator's stream, ordered against the ctx stream by
    events, no host synchronisation).  In a child process so that librccl is only mapped there."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path.insert(0, {0!r}); sys.path.insert(0, {1!r})\n'
            'import test_gpu_fullfield as t\n'
            'from beyond_dof_amd.comm import RcclComm\n'
            'c = RcclComm(); assert c.always_reduce and c.size == 1\n'
            't._slab_case(c); c.close(); assert "torch" not in sys.modules; print("SLAB_RCCL_OK")\n').format(root, os.path.join(root, 'tests'))
    env = dict(os.environ, BDOF_FORCE_COMM='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b'SLAB_RCCL_OK' in r.stdout, r.stdout.decode()[-3000:]


def test_reconstruct_fullfield_vs_the_reference_loop_fft(tmp_path, monkeypatch):
    """The product's DEFAULT entry point (transfer-function propagator, the north-star path) against golden vector G15: the
    reference's own reconstruct_fullfield loop executed at 64^3 with the name it calls for the forward model bound to the
    reference's np_funcs.multislice_propagate_batch_numpy (tests/golden/make_golden.py --g15) — 4 angles in minibatches of 2,
    two epochs = four Adam steps, L1 + TV, mask, clip.  Same data file, mask files, initial guess and seed in, volume out."""
    import sys
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io, tiffio
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g15_reconstruct_fullfield_fft_64.npz'))
    init_d, init_b = g13_inputs.initial_guess()
    monkeypatch.chdir(tmp_path)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', g['prj'])
    tiffio.write_tiff_stack(g13_inputs.mask(), 'case/fin_sup_mask/mask', dtype='float32', overwrite=True)
    lr = 1e-7
    d, b = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=2, learning_rate=lr, minibatch_size=2, energy_ev=5000,
                                 psize_cm=1e-7, free_prop_cm=1e-4, save_path='case', output_folder='out', initial_guess=[init_d, init_b],
                                 shrink_cycle=None, seed=5, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    assert float(g['delta_moved_max']) >= 3.9 * lr                       # four whole steps were taken
    d, b = d[::2, ::2, ::2], b[::2, ::2, ::2]
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    stats = (np.abs(d - g['delta_sub']).max() / lr, rel(d, g['delta_sub']), np.abs(b - g['beta_sub']).max() / lr, rel(b, g['beta_sub']))
    print('G15 stats', stats)
    assert stats[0] <= 0.01 and stats[1] <= 1e-5, stats                 # the north-star bound on the reconstructed delta
    assert stats[2] <= 0.01 and stats[3] <= 5e-5, stats


@pytest.mark.parametrize('fixture,noise,adjoint_precision', [('g18_reconstruct_fullfield_fft_256.npz', 0.0, 'float32'),
                                                             ('g19_reconstruct_fullfield_fft_256_noisy.npz', 0.02, 'float32'),
                                                             ('g19_reconstruct_fullfield_fft_256_noisy.npz', 0.02, 'first-step')])
def test_reconstruct_fullfield_vs_the_reference_loop_at_cfg2_size(tmp_path, monkeypatch, fixture, noise, adjoint_precision):
    """Golden vector G18: the comparison of G15 at BASELINE config 2's volume size — 256^3, 256 slices, 4 angles in minibatches
    of 2, two epochs.  The data (an input: both sides only have to read the same array) come from the oracle's forward model
    on a formula phantom, computed here as the generator did; the reference loop's volumes are stored on every eighth voxel.
    G19 is the same with 2 % (seeded) amplitude noise on the data, the regime measured data are in."""
    import sys
    from beyond_dof_amd import h5io, tiffio
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, fixture))
    n = 256
    init_d, init_b = g13_inputs.initial_guess((n, n, n))
    prj = g13_inputs.data_from_phantom(orc, (n, n, n), 4, noise)
    monkeypatch.chdir(tmp_path)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', prj)
    tiffio.write_tiff_stack(g13_inputs.mask((n, n, n)), 'case/fin_sup_mask/mask', dtype='float32', overwrite=True)
    lr = 1e-7
    d, b = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=2, learning_rate=lr, minibatch_size=2, energy_ev=5000,
                                 psize_cm=1e-7, free_prop_cm=1e-4, save_path='case', output_folder='out', initial_guess=[init_d, init_b],
                                 shrink_cycle=None, seed=5, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, adjoint_precision=adjoint_precision)
    assert float(g['delta_moved_max']) >= 3.5 * lr
    d, b = d[::8, ::8, ::8], b[::8, ::8, ::8]
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    stats = (np.abs(d - g['delta_sub']).max() / lr, rel(d, g['delta_sub']), np.abs(b - g['beta_sub']).max() / lr, rel(b, g['beta_sub']))
    print('G18 stats' if not noise else 'G19 stats', adjoint_precision, stats)
    # measured (round 3, dithered transform constants — the default): G18 (noise-free data) delta 6.6e-6, beta 4.3e-5; G19 (2 % noise)
    # delta 7.4e-6, beta 2.7e-5 — both inside the north star's 1e-5 (hi + lo tables in every transform, -DBDOF_EXACT_TRANSFORMS:
    # 5.8e-6 / 7.0e-6; one plain table, BDOF_TW_DITHER=0: 2.1e-5 / 1.68e-5).  Round 4 (dithered transfer function): 2.0e-6 /
    # 4.2e-6; adjoint_precision='first-step' (the first minibatch of each epoch through bdof_loss_grad_tf_f64) is offered as well
    assert stats[0] <= 0.05 and stats[1] <= 1e-5 and stats[3] <= 1e-4, stats


@pytest.mark.parametrize('model', ['fft', 'conv'])
def test_device_gradient_vs_directional_derivatives_of_the_reference_loss(model):
    """Golden vector G20: central differences of the REFERENCE's own calculate_loss (cnn_propagator/fullfield.py:93-121 executed at
    (64, 64, 64), first minibatch of the G15 / G13 runs; tests/golden/make_golden.py --g20) along six directions — numbers no
    oracle code produced.  The device's volume gradient (rotation gather, multislice forward, loss, adjoint sweep, rotation
    adjoint), projected on the same directions, for the transfer-function propagator and for the real-space one."""
    import sys
    from beyond_dof_amd.solver import FullfieldSolver
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g20_directional_derivatives_64.npz'))
    shape = (64, 64, 64)
    mask = g13_inputs.mask(shape)
    init_d, init_b = g13_inputs.initial_guess(shape)
    d, b = init_d * mask, init_b * mask
    prj = np.load(os.path.join(gdir, 'g15_reconstruct_fullfield_fft_64.npz' if model == 'fft' else 'g13_reconstruct_fullfield_64.npz'))['prj']
    ind = g[model + '_ind']
    s = FullfieldSolver(64, 64, 64, 4, len(ind), 5000., 1e-7, free_prop_cm=1e-4, propagator=model, kernel_size=17)
    s.set_measurements(np.abs(prj))
    s.set_volume(d, b)
    loss = s.loss_and_grad(ind)
    gd, gb = s.gradient_to_host()
    assert abs(loss - float(g[model + '_loss'])) <= 1e-5 * abs(loss)
    dirs = g13_inputs.g20_directions(shape)
    got = np.array([[np.sum(gd.astype(np.float64) * v), np.sum(gb.astype(np.float64) * v)] for v in dirs])
    ref = g[model + '_dd']
    # each number against the larger of itself and a tenth of the largest projection of its parameter: a projection is a sum of
    # 262144 signed terms, and the ones that cancel to a few per cent of the others carry the same ABSOLUTE float32 error
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 0.1 * np.abs(ref).max(axis=0))
    print('G20', model, 'loss', abs(loss - float(g[model + '_loss'])) / abs(loss), 'directional derivatives rel err', err.ravel())
    # measured: transfer-function model 2.6e-6 at worst (dithered transform constants; 5.2e-6 with hi + lo tables), real-space model
    # 1.4e-5 (3.1e-5 before its taps were dithered over the slices too)
    assert np.max(err) <= (1.5e-5 if model == 'fft' else 5e-5), (got, ref)
