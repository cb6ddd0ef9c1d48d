"""Ptychography path: per-position windows cut by index math in the kernels (zero padding beyond the volume, windows
overlapping), far-field detector, adjoint through windows + rotation — against the oracle's pad/slice/scatter-add
restatement of cnn_propagator/ptychography.py:30-81."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _setup(psz=(64, 64)):
    rng = np.random.default_rng(0)
    n, n_theta = 96, 5
    # positions include windows hanging over every edge (zero padding) and heavy overlap
    pos = np.array([(y, x) for y in (10, 48, 90) for x in (5, 40, 70, 95)])
    od = rng.uniform(0, 2e-5, size=(n, n, n))
    ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    prr, pii = orc.gaussian_probe(psz, 6., 6., 0.5)
    return rng, n, n_theta, psz, pos, od, ob, coords, prr, pii


@pytest.mark.parametrize('psz,force_resident', [((64, 64), False), ((64, 64), True), ((72, 72), False), ((60, 60), False)],
                         ids=['64-streaming', '64-resident', '72-resident', '60-rocfft'])
def test_ptycho_forward_and_gradient_vs_oracle(psz, force_resident, monkeypatch):
    """Rotation table + per-position windows (zero padding beyond the volume) through each of the three device engines."""
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd.solver import PtychoSolver
    if force_resident:
        monkeypatch.setenv('BDOF_FORCE_RESIDENT', '1')
    elif psz == (64, 64):
        monkeypatch.setenv('BDOF_NO_RESIDENT_PIN', '1')      # PtychoSolver pins the resident engine where it exists: not here
    rng, n, n_theta, psz, pos, od, ob, coords, prr, pii = _setup(psz)
    mb = 6
    s = PtychoSolver((n, n, n), psz, pos, n_theta, mb, 5000., 1e-7, prr, pii, coord_ls=coords)
    s.set_volume(od, ob)
    sel = np.array([0, 3, 5, 6, 10, 11])
    i_theta = 2
    # oracle forward of the windows
    pad, half = orc.ptycho_pad_amounts(pos, psz, (n, n, n))
    rot = orc.apply_rotation(np.stack([od, ob], axis=3), coords[i_theta])
    obj_pad = np.pad(rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
    subs = np.stack([obj_pad[p[0] + pad[0, 0] - half[0]:p[0] + pad[0, 0] - half[0] + psz[0],
                             p[1] + pad[1, 0] - half[1]:p[1] + pad[1, 0] - half[1] + psz[1]] for p in pos[sel]])
    ref, _ = orc.multislice_propagate_batch_numpy(subs[..., 0], subs[..., 1], prr, pii, 5000., 1e-7, 'inf', subs[..., 0].shape,
                                                  return_probe_array=False)
    w = s.forward(i_theta, sel)
    # 96 slices, localised probe (no carrier to split off): float32 round-off of the 192 transforms accumulates to
    # 4e-6 (60^2) ... 1.0e-5 (72^2 on the rocFFT engine) in the intensities (tools/gpu_check_pty_err.py) — it was 1.4e-5
    # before the systematic energy drift of float32 transform chains was taken out; the bound stays at 1.2e-5,
    # above the 1e-5 of the short-stack comparisons in test_gpu_parity.py / test_gpu_resident.py.
    on_resident = s.eng.probe_stack            # carrier field of the probe (bdof_set_probe_stack): every engine carries it
    assert on_resident
    assert rel(np.abs(w) ** 2, np.abs(ref) ** 2) <= (5e-7 if on_resident else 1.2e-5)
    meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
    loss = s.loss_and_grad(i_theta, sel, meas)
    gd, gb = s.gradient_to_host()
    rl, rgd, rgb = orc.ptycho_loss_and_grad(od, ob, coords[i_theta], pos, pos[sel], meas, prr, pii, psz, 5000., 1e-7)
    assert abs(loss - rl) <= (2e-6 if on_resident else 5e-5) * rl
    # far field, 96 slices, localised probe: without a carrier to split off the |D| - |m| cancellation in float32 costs
    # ~3e-4 of the gradient (see tests/test_gpu_parity.py header), bound 1e-3; on the resident engine the probe's free-space
    # propagation is the carrier (bdof_set_probe_stack) and the whole comparison tightens by two orders of magnitude
    # (BDOF_NO_PROBE_STACK=1 restores the plain float32 chain; tools/gpu_check_stream_stack.py prints both)
    gtol = 1e-4 if on_resident else 1e-3
    assert rel(gd, rgd) <= gtol and rel(gb, rgb) <= gtol


def test_reconstruct_ptychography_end_to_end(tmp_path, monkeypatch):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io, tiffio
    from beyond_dof_amd.ptychography import reconstruct_ptychography, epoch_schedule
    from beyond_dof_amd.solver import PtychoSolver
    monkeypatch.chdir(tmp_path)
    rng, n, n_theta, psz, pos, od, ob, coords, prr, pii = _setup()
    n_theta = 3
    coords = orc.rotation_lookup([n, n, n], n_theta)
    # data from the oracle forward model, (n_theta, n_pos, py, px) complex64   (cnn_propagator/simulation.py:363)
    pad, half = orc.ptycho_pad_amounts(pos, psz, (n, n, n))
    data = np.zeros((n_theta, len(pos), psz[0], psz[1]), dtype=np.complex64)
    for t in range(n_theta):
        rot = orc.apply_rotation(np.stack([od, ob], axis=3), coords[t])
        obj_pad = np.pad(rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
        subs = np.stack([obj_pad[p[0] + pad[0, 0] - half[0]:p[0] + pad[0, 0] - half[0] + psz[0],
                                 p[1] + pad[1, 0] - half[1]:p[1] + pad[1, 0] - half[1] + psz[1]] for p in pos])
        w, _ = orc.multislice_propagate_batch_numpy(subs[..., 0], subs[..., 1], prr, pii, 5000., 1e-7, 'inf', subs[..., 0].shape,
                                                    return_probe_array=False)
        data[t] = w
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', data)
    init_d = np.full((n, n, n), 8e-6)
    init_b = np.full((n, n, n), 8e-7)
    kw = dict(probe_mag_sigma=6., probe_phase_sigma=6., probe_phase_max=0.5)
    d, b = reconstruct_ptychography('data.h5', [tuple(p) for p in pos], psz, (n, n, n), theta_st=0, theta_end=2 * np.pi,
                                    n_epochs=2, learning_rate=2e-7, minibatch_size=5, energy_ev=5000, psize_cm=1e-7,
                                    save_path='case', output_folder='out', initial_guess=[init_d, init_b],
                                    probe_type='gaussian', seed=3, n_dp_batch=20, **kw)
    assert os.path.exists('case/out/summary.txt')
    assert np.array_equal(tiffio.read_tiff('case/out/delta_ds_1.tiff'), d.astype(np.float32))
    assert np.all(d >= 0) and np.all(b >= 0)
    # the data-term loss over the whole data set went down
    s = PtychoSolver((n, n, n), psz, pos, n_theta, len(pos), 5000., 1e-7, prr, pii, coord_ls=coords)

    def total(dd, bb):
        s.set_volume(dd, bb)
        return sum(s.loss_and_grad(t, np.arange(len(pos)), np.abs(data[t])) for t in range(n_theta))
    assert total(d, b) < total(init_d, init_b)
    # schedule shape: every theta's list is padded to a multiple of the minibatch (quirk Q10)
    sched = epoch_schedule(3, 12, 5, np.random.RandomState(0))
    assert len(sched) == 3 * 15 and all(len(set(sched[i * 15:(i + 1) * 15, 0])) == 1 for i in range(3))


@pytest.mark.parametrize('psz', [(64, 64), (35, 35)])
def test_resident_measurements_equal_per_step_upload(psz):
    """PtychoSolver.set_measurements + step(prj_abs_batch=None): the minibatch picked out of the device-resident stack by one
    gather launch gives the very same volume as uploading it per step.  35 x 35 (generic engine): 4900 bytes per field, not
    a multiple of 16 — the gather then copies 4-byte words (round-2 advice: it refused)."""
    from beyond_dof_amd.solver import PtychoSolver
    rng, n, n_theta, psz, pos, od, ob, coords, prr, pii = _setup(psz)
    meas = np.abs(rng.normal(1.0, 0.1, size=(n_theta, len(pos)) + psz)) * 30
    vols = []
    for resident in (False, True):
        s = PtychoSolver((n, n, n), psz, pos, n_theta, 6, 5000., 1e-7, prr, pii, coord_ls=coords)
        s.set_volume(od, ob)
        if resident:
            s.set_measurements(meas)
        s.reset_moments()
        for i, (i_theta, sel) in enumerate(((2, np.array([0, 3, 5, 6, 10, 11])), (4, np.array([1, 2, 4, 7, 8, 9])))):
            s.step(i, i_theta, sel, None if resident else meas[i_theta, sel], 1e-7)
        vols.append(s.get_volume())
    assert np.array_equal(vols[0][0], vols[1][0]) and np.array_equal(vols[0][1], vols[1][1])


@pytest.mark.parametrize('adjoint_precision', ['float32', 'float64', 'first-step'])
def test_reconstruct_ptychography_vs_the_reference_loop_fft(tmp_path, monkeypatch, adjoint_precision):
    """The DEFAULT ptychography entry point (transfer-function propagator, LDS-resident engine for the 64 x 64 probe, far field)
    against golden vector G17: the reference's own reconstruct_ptychography loop executed with the name it calls for the forward
    model bound to the reference's np_funcs.multislice_propagate_batch_numpy (tests/golden/make_golden.py --g17) — 64^3 object,
    gaussian probe (sigma 10), 4 positions x 2 angles in minibatches of 2, two epochs = 8 Adam steps, 2 % noise on the data.

    What separates a float32 device from the reference's float64 loop here is Adam's FIRST step of every epoch,
    lr g / (|g| + 1e-8): at the ~100 voxels (of 262144) where the gradient changes sign within 1e-7 of zero an absolute error of
    1e-8 — 1e-6 of the gradient's rms — is a fraction of a whole step (tools/gpu_diag_g17_steps.py).  Round 3 took the
    largest term out (the residual |d| - m is formed in float64 against the float64 carrier field: 4.9e-5 -> 1.7e-5); what is
    left is the float32 rounding of the adjoint sweep's transforms (3e-6 on the gradient), which has no known part to split
    off.  adjoint_precision='float64' runs that sweep in float64 and lands within 1.6e-6 — inside the north star's 1e-5."""
    import sys
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io
    from beyond_dof_amd.ptychography import reconstruct_ptychography
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g17_reconstruct_ptychography_fft_64.npz'))
    obj_size, psz, sigma = tuple(int(v) for v in g['obj_size']), tuple(int(v) for v in g['probe_size']), float(g['probe_sigma'])
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    monkeypatch.chdir(tmp_path)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', g['prj'])
    lr = 2e-7
    d, b = reconstruct_ptychography('data.h5', [tuple(int(v) for v in p) for p in g['probe_pos']], psz, obj_size, theta_st=0, theta_end=2 * np.pi,
                                    n_epochs=2, learning_rate=lr, minibatch_size=2, energy_ev=5000, psize_cm=1e-7, save_path='case',
                                    output_folder='out', initial_guess=[init_d, init_b], probe_type='gaussian', dynamic_dropping=False,
                                    seed=42, probe_mag_sigma=sigma, probe_phase_sigma=sigma, probe_phase_max=0.5,
                                    adjoint_precision=adjoint_precision)
    assert float(g['delta_moved_max']) >= 5 * lr
    summary = dict(line.split(None, 1) for line in open('case/out/summary.txt').read().splitlines() if len(line.split(None, 1)) == 2)
    assert summary['adjoint_precision'].strip() == adjoint_precision and summary['adjoint_precision_effective'].strip() == adjoint_precision
    d, b = d[::2, ::2, ::2], b[::2, ::2, ::2]
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    dev, devb = np.abs(d - g['delta_sub']), np.abs(b - g['beta_sub'])
    stats = (dev.max() / lr, rel(d, g['delta_sub']), devb.max() / lr, rel(b, g['beta_sub']), float(np.mean(dev > 0.05 * lr)))
    print('G17 stats', adjoint_precision, stats)
    if adjoint_precision == 'float64':
        # measured: delta 1.6e-6, beta 2.6e-7, no voxel more than 0.001 of a step away
        assert stats[0] <= 0.01 and stats[1] <= 5e-6 and stats[3] <= 2e-6 and stats[4] == 0.0, stats
    elif adjoint_precision == 'first-step':
        # float64 for the first minibatch of each epoch only (2 of the 8 steps here, 1 in hundreds at cfg5's size): 2.1e-6
        assert stats[0] <= 0.02 and stats[1] <= 5e-6 and stats[3] <= 2e-6 and stats[4] == 0.0, stats
    else:
        # all float32, the LDS-resident kernel alone.  Round 4: the kernel multiplies by the slice's DITHERED copy of the transfer
        # function (bdof_set_transfer_f64) — a fixed float32 H was the same perturbation in all 64 slices of both sweeps:
        # delta 1.7e-5 -> 7.0e-6, beta 1.7e-6 -> 9.4e-7, no voxel more than 0.005 of a step away.  Inside the north star's 1e-5
        # without any float64 sweep (round 2, residual in float32 as well: 4.9e-5)
        assert stats[0] <= 0.02 and stats[1] <= 1e-5 and stats[3] <= 5e-6 and stats[4] == 0.0, stats


def test_first_step_engine_that_cannot_be_set_up_is_reported_and_recorded(tmp_path, monkeypatch, capsys):
    """The default adjoint_precision='first-step' allocates the float64 path's wave and tape beside the engine's; where that fails
    the run continues in float32 — with less margin under the 1e-5 bound — so it must SAY so and RECORD it: one message
    that names the consequence, `adjoint_precision_effective  float32` in summary.txt.  An explicit request fails instead."""
    import sys
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io, ptychography
    from beyond_dof_amd._lib import BdofError
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g17_reconstruct_ptychography_fft_64.npz'))
    obj_size, psz, sigma = tuple(int(v) for v in g['obj_size']), tuple(int(v) for v in g['probe_size']), float(g['probe_sigma'])
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    monkeypatch.chdir(tmp_path)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', g['prj'])
    real, asked = ptychography.PtychoSolver, []

    def failing(*a, **kw):
        asked.append(kw.get('adjoint64'))
        if kw.get('adjoint64') == 'first':
            raise BdofError('bdof_configure: out of device memory (forced by the test)')
        return real(*a, **kw)
    monkeypatch.setattr(ptychography, 'PtychoSolver', failing)
    run = lambda **kw: ptychography.reconstruct_ptychography(
        'data.h5', [tuple(int(v) for v in p) for p in g['probe_pos']], psz, obj_size, theta_st=0, theta_end=2 * np.pi, n_epochs=1,
        learning_rate=2e-7, minibatch_size=2, energy_ev=5000, psize_cm=1e-7, save_path='case', output_folder='out',
        initial_guess=[init_d, init_b], probe_type='gaussian', dynamic_dropping=False, seed=42, probe_mag_sigma=sigma,
        probe_phase_sigma=sigma, probe_phase_max=0.5, **kw)
    d, _ = run()
    assert asked == ['first', None] and np.all(np.isfinite(d))
    out = capsys.readouterr().out
    assert "adjoint_precision='first-step' could not be set up" in out and "continuing with 'float32'" in out and 'forced by the test' in out
    summary = dict(line.split(None, 1) for line in open('case/out/summary.txt').read().splitlines() if len(line.split(None, 1)) == 2)
    assert summary['adjoint_precision'].strip() == 'first-step' and summary['adjoint_precision_effective'].strip() == 'float32'
    with pytest.raises(BdofError):
        run(adjoint_precision='first-step')


def test_cfg5_solver_step_at_full_shape():
    """BASELINE configs[4] through the SOLVER at its stated shape: 256^3 volume, 400 probe positions of 72 x 72 (20 x 20 grid,
    step 12), 256 slices, far field — PtychoSolver.loss_and_grad / step with the window + rotation adjoint of a 256^3 volume,
    what cnn_propagator/ptychography.py:285-310 runs per minibatch (the engine alone is test_gpu_resident's
    test_cfg5_full_size_properties).  The float64 oracle does not finish 400 wavefields of 256 slices in seconds, so:
    (1) three positions against the oracle's gradient (full depth, full volume, rotation included);
    (2) the loss is a mean over the minibatch's pixels, so its gradient over all 400 positions is the average of the
        gradients over the two halves, and the loss the average of the losses — a size-independent property of the whole
        pipeline (forward, loss, adjoint, overlap-add of 400 windows, rotation adjoint);
    (3) one Adam step moves exactly the voxels the 400 windows illuminate and leaves the volume non-negative."""
    from beyond_dof_amd.solver import PtychoSolver
    n, psz, n_theta, side = 256, (72, 72), 4, 20
    rng = np.random.default_rng(5)
    pos = np.array([(y, x) for y in np.arange(side) * 12 + 14 for x in np.arange(side) * 12 + 14])
    assert len(pos) == 400
    od = rng.random((n, n, n)) * 1e-6
    ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    prr, pii = orc.gaussian_probe(psz, 6., 6., 0.5)
    i_theta = 1
    # (1) three positions against the oracle
    sel = np.array([0, 211, 399])
    meas3 = (np.abs(rng.normal(1.0, 0.1, size=(3,) + psz)) * 40).astype(np.float32)
    s3 = PtychoSolver((n, n, n), psz, pos, n_theta, 3, 5000., 1e-7, prr, pii, coord_ls=coords)
    s3.set_volume(od, ob)
    loss3 = s3.loss_and_grad(i_theta, sel, meas3)
    gd, gb = s3.gradient_to_host()
    rl, rgd, rgb = orc.ptycho_loss_and_grad(od, ob, coords[i_theta], pos, pos[sel], meas3.astype(np.complex128), prr, pii, psz, 5000., 1e-7)
    assert abs(loss3 - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4, (rel(gd, rgd), rel(gb, rgb))
    del s3
    # (2) all 400 positions = the average of the two halves
    meas = (np.abs(rng.normal(1.0, 0.1, size=(n_theta, 400) + psz)) * 40).astype(np.float32)
    s = PtychoSolver((n, n, n), psz, pos, n_theta, 400, 5000., 1e-7, prr, pii, coord_ls=coords, adjoint64='first')
    s.set_volume(od, ob)
    s.set_measurements(meas)
    l_all = s.loss_and_grad(i_theta, np.arange(400))
    g_all = s.gradient_to_host()
    # (2b) ... and against the float64 path on the same context (bdof_loss_grad_tf_f64; 3e-15 / 2.5e-8 from the oracle at the sizes
    # the oracle reaches, tests/test_gpu_parity.py): the whole 400-position minibatch of the LDS-resident kernel, on amplitudes
    # the model can produce (its own diffraction patterns with 5 % noise).  With the random amplitudes above the float32 gradient is
    # 4e-5 from the float64 one, and rightly so: they put a residual of 40 on detector pixels where the wave is 1e-7 of its
    # peak, and the seed (|d| - m) d / |d| then needs the PHASE of a number that float32 transforms do not resolve.
    consistent = (np.abs(s.forward(i_theta, np.arange(400))) * (1 + 0.05 * rng.normal(size=(400,) + psz))).astype(np.float32)
    l32 = s.loss_and_grad(i_theta, np.arange(400), consistent)
    g32 = s.gradient_to_host()
    l64 = s.loss_and_grad(i_theta, np.arange(400), consistent, f64=True)
    g64 = s.gradient_to_host()
    e64 = (abs(l32 - l64) / abs(l64), rel(g32[0], g64[0]), rel(g32[1], g64[1]))
    print('cfg5 whole minibatch, resident float32 kernel vs the float64 path: loss', e64[0], 'gradient', e64[1:])
    assert e64[0] <= 1e-6 and e64[1] <= 1e-5 and e64[2] <= 1e-5, e64
    del g32, g64
    h = PtychoSolver((n, n, n), psz, pos, n_theta, 200, 5000., 1e-7, prr, pii, coord_ls=coords)
    h.set_volume(od, ob)
    h.set_measurements(meas)
    halves = []
    for part in (np.arange(0, 400, 2), np.arange(1, 400, 2)):
        lp = h.loss_and_grad(i_theta, part)
        halves.append((lp, h.gradient_to_host()))
    del h
    assert abs(l_all - 0.5 * (halves[0][0] + halves[1][0])) <= 1e-6 * abs(l_all)
    for c in range(2):
        avg = 0.5 * (halves[0][1][c] + halves[1][1][c])
        assert rel(g_all[c], avg) <= 2e-6, (c, rel(g_all[c], avg))
    # (3) one Adam step of the solver
    lit = np.abs(g_all[0]) > 0
    s.reset_moments()
    s.step(0, i_theta, np.arange(400), None, 1e-7)
    d1, b1 = s.get_volume()
    d0, b0 = od.astype(np.float32), ob.astype(np.float32)
    moved = d1 != d0
    assert not np.any(moved & ~lit)                                                      # nothing outside the windows moved
    assert np.mean(moved[lit]) > 0.99 and np.all(d1 >= 0) and np.all(b1 >= 0)
    assert np.abs(d1 - d0).max() <= 1.0001e-7                                             # Adam's first step: at most lr


@pytest.mark.parametrize('model,adjoint64', [('fft', False), ('fft', True), ('conv', False), ('conv', True)])
def test_device_gradient_vs_directional_derivatives_of_the_reference_loss(model, adjoint64):
    """Golden vector G21: central differences of the REFERENCE's own calculate_loss (cnn_propagator/ptychography.py:30-81 executed
    at (64, 64, 64), first minibatch of the G17 / G14 runs; tests/golden/make_golden.py --g21) along six directions — numbers no
    oracle code produced.  The device's volume gradient (rotation gather, zero-padded window cut, multislice forward, far-field
    loss, adjoint sweep, window overlap-add, rotation adjoint) projected on the same directions."""
    import sys
    from beyond_dof_amd.solver import PtychoSolver
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g21_ptycho_directional_derivatives_64.npz'))
    f = np.load(os.path.join(gdir, 'g17_reconstruct_ptychography_fft_64.npz' if model == 'fft' else 'g14_reconstruct_ptychography_64.npz'))
    obj_size, psz, sigma = tuple(int(v) for v in f['obj_size']), tuple(int(v) for v in f['probe_size']), float(f['probe_sigma'])
    d, b = g13_inputs.initial_guess(obj_size)
    pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
    i_theta, batch = int(g[model + '_i_theta']), g[model + '_pos_batch']
    ind = np.array([int(np.where((f['probe_pos'] == p).all(axis=1))[0][0]) for p in batch])
    s = PtychoSolver(obj_size, psz, f['probe_pos'], f['prj'].shape[0], len(ind), 5000., 1e-7, pr, pi_, propagator=model, kernel_size=17,
                     adjoint64=adjoint64 or None)
    s.set_volume(d, b)
    loss = s.loss_and_grad(i_theta, ind, np.abs(f['prj'][i_theta, ind]))
    gd, gb = s.gradient_to_host()
    dirs = g13_inputs.g20_directions(obj_size)
    got = np.array([[np.sum(gd.astype(np.float64) * v), np.sum(gb.astype(np.float64) * v)] for v in dirs])
    ref = g[model + '_dd']
    # each number against the larger of itself and a tenth of the largest projection of its parameter: a projection is a sum of
    # 262144 signed terms, and the ones that cancel to a few per cent of the others carry the same ABSOLUTE float32 error
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 0.1 * np.abs(ref).max(axis=0))
    print('G21', model, 'adjoint64' if adjoint64 else '', 'loss', abs(loss - float(g[model + '_loss'])) / abs(loss), 'directional derivatives rel err',
          err.ravel())
    assert abs(loss - float(g[model + '_loss'])) <= 1e-5 * abs(loss)
    # measured: transfer-function model 7.7e-6 (float64 adjoint sweep 3.0e-6: the differences' own floor is 5e-6), real-space
    # model 7.5e-5 (float32 kernel taps; far-field loss of a sigma-40 probe) and, through the float64 real-space path of the same
    # context (adjoint64=True: bdof_loss_grad_conv_f64, what the default 'first-step' runs once per epoch), at the differences' floor
    assert np.max(err) <= (1.5e-5 if model == 'fft' or adjoint64 else 2e-4), (got, ref)
