"""rotation='bilinear': the TF twin's tf_rotate(obj, theta, 'BILINEAR') (tensorflow_recon/fullfield.py:96) in place of the cnn
variant's nearest-neighbour tables — bdof_rotate_bilinear / bdof_rotate_bilinear_adjoint through FullfieldSolver, against the
oracle's restatement (tests/test_oracle_bilinear.py; TF itself is not installed: parity unpinned by execution)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _case(n=64, n_theta=7, mb=3, fp=1e-4, seed=0):
    rng = np.random.default_rng(seed)
    od = np.zeros((n, n, n))
    od[:, 8:-8, 8:-8] = rng.uniform(0, 2e-6, size=(n, n - 16, n - 16))
    ob = 0.1 * od
    theta = -np.linspace(0, 2 * np.pi, n_theta).astype(np.float32)          # tensorflow_recon/fullfield.py:216
    idx = np.array([1, 3, 6])[:mb]
    return rng, od, ob, theta, idx


def _oracle(od, ob, theta, idx, prj, fp, one, zero):
    obj = np.stack([od, ob], axis=3)
    rot = np.stack([orc.rotate_bilinear(obj, float(theta[j])) for j in idx])
    loss, gd, gb = orc.multislice_loss_and_grad(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, prj, fp)
    g = sum(orc.rotate_bilinear_adjoint(np.stack([gd[b], gb[b]], axis=3), float(theta[j])) for b, j in enumerate(idx))
    return loss, g[..., 0], g[..., 1], rot


@pytest.mark.parametrize('fp', [1e-4, None])
def test_bilinear_rotation_forward_gradient_and_adam(fp):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd.solver import FullfieldSolver
    n, n_theta, mb = 64, 7, 3
    rng, od, ob, theta, idx = _case(n, n_theta, mb, fp)
    one, zero = np.ones((n, n)), np.zeros((n, n))
    obj = np.stack([od, ob], axis=3)
    rot = np.stack([orc.rotate_bilinear(obj, float(theta[j])) for j in idx])
    ref, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, fp, rot[..., 0].shape, return_probe_array=False)
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp, rotation='bilinear', theta=theta)
    s.set_volume(od, ob)
    w = s.forward_angles(idx)
    assert rel(np.abs(w) ** 2, np.abs(ref) ** 2) <= 1e-5 and rel(w, ref) <= 1e-5
    # the rotated objects themselves (bdof_rotate_bilinear; the solver's one-pass bdof_set_object_bilinear writes their
    # modulation factors instead) ...
    from beyond_dof_amd._lib import DeviceBuffer
    lib, h = s.ctx.lib, s.ctx.handle
    rows = DeviceBuffer(s.ctx, mb * n * n * n * 8, np.float32, (mb, n, n, n, 2))
    prm = DeviceBuffer.from_host(s.ctx, np.ascontiguousarray(s.rot_prm[idx]))
    s.ctx.check(lib.bdof_rotate_bilinear(h, s.x[s.cur].ptr, n, n, n, prm.ptr, mb, rows.ptr))
    s.ctx.sync()
    got = rows.download()                                                 # [b][z][x][y][2]
    assert rel(got[..., 0].transpose(0, 3, 2, 1), rot[..., 0]) <= 2e-6
    # ... and bound the two-pass way they give the same wave, bit for bit
    s.eng.set_volume(rows, mb * n * n, n, None, 0, 0)
    assert np.array_equal(s.eng.forward(mb), w)
    prj = np.zeros((n_theta, n, n))
    prj[idx] = (np.abs(ref) * (1 + 0.03 * rng.normal(size=ref.shape))).astype(np.float32)
    s.set_measurements(prj)
    loss = s.loss_and_grad(idx)
    gd, gb = s.gradient_to_host()
    rl, rgd, rgb, _ = _oracle(od, ob, theta, idx, prj[idx], fp, one, zero)
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4, (rel(gd, rgd), rel(gb, rgb))
    # Adam steps: whole volume and slab-wise give the same volume; the loss goes down
    vols = []
    for comm, kw in ((None, {}), ('loop', dict(n_slabs=4, sharded=True))):
        c = None
        if comm:
            import sys, os
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from test_gpu_fullfield import _LoopbackComm
            c = _LoopbackComm()
        s2 = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp, rotation='bilinear', theta=theta, comm=c)
        s2.set_volume(od * 0.9, ob * 0.9)
        s2.set_measurements(prj)
        s2.reset_moments()
        l0 = s2.step(0, idx, 1e-8, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, want_loss=True, **kw)
        s2.step(1, idx, 1e-8, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, **kw)
        l2 = s2.loss_and_grad(idx)
        assert l2 < l0
        vols.append(s2.get_volume())
    assert np.array_equal(vols[0][0], vols[1][0]) and np.array_equal(vols[0][1], vols[1][1])


def test_reconstruct_fullfield_bilinear(tmp_path, monkeypatch):
    """The entry point with rotation='bilinear' (true angles theta = -linspace(theta_st, theta_end))."""
    import os
    from beyond_dof_amd import h5io
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    monkeypatch.chdir(tmp_path)
    n, n_theta = 64, 6
    rng, od, ob, theta, _ = _case(n, n_theta)
    one, zero = np.ones((n, n)), np.zeros((n, n))
    obj = np.stack([od, ob], axis=3)
    rot = np.stack([orc.rotate_bilinear(obj, float(t)) for t in theta])
    prj, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, 1e-4, rot[..., 0].shape, return_probe_array=False)
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', prj.astype(np.complex64))
    d, b = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=1, learning_rate=1e-8, minibatch_size=3, energy_ev=5000,
                                 psize_cm=1e-7, free_prop_cm=1e-4, save_path='case', output_folder='out', initial_guess=[od * 0.9, ob * 0.9],
                                 shrink_cycle=None, seed=1, alpha_d=0., alpha_b=0., gamma=0., rotation='bilinear')
    assert os.path.exists(os.path.join('case', 'out', 'delta_ds_1.tiff')) and d.shape == (n, n, n)
    assert not os.path.exists('arrsize_64_64_64_ntheta_6')                  # no lookup tables are made in this mode
    assert np.linalg.norm(d - od) < np.linalg.norm(0.9 * od - od)           # moved towards the truth


def test_bilinear_binding_for_the_real_space_propagator():
    """bdof_set_object_bilinear(conv=1) writes the factors for the real-space propagator's k (numpy's pi, not the FFT path's
    literal — the two agree to 1e-8, i.e. they are the same float32 here): same waves as rotating first
    (bdof_rotate_bilinear) and binding the rows (bdof_set_object)."""
    from beyond_dof_amd._lib import DeviceBuffer
    from beyond_dof_amd.solver import FullfieldSolver
    n, n_theta, mb = 64, 7, 3
    rng, od, ob, theta, idx = _case(n, n_theta, mb)
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, rotation='bilinear', theta=theta, propagator='conv')
    s.set_volume(od, ob)
    w = s.forward_angles(idx)                                             # fused binding, conv sweep
    lib, h = s.ctx.lib, s.ctx.handle
    rows = DeviceBuffer(s.ctx, mb * n * n * n * 8, np.float32, (mb, n, n, n, 2))
    prm = DeviceBuffer.from_host(s.ctx, np.ascontiguousarray(s.rot_prm[idx]))
    s.ctx.check(lib.bdof_rotate_bilinear(h, s.x[s.cur].ptr, n, n, n, prm.ptr, mb, rows.ptr))
    s.eng.set_volume(rows, mb * n * n, n, None, 0, 0)
    assert np.array_equal(s.eng.forward(mb, conv=True), w)
