"""Full-field parity against the float64 oracle through the product's own solver (FullfieldSolver: fused rotation, forward,
loss, adjoint, rotation adjoint, regulariser + Adam + mask + clip):
  * plane probe + far-field detector (reconstruct_fullfield(probe_type='plane', free_prop_cm='inf')), the case whose
    gradient was wrong by 25 % before the adjoint carrier (bdof_kernels.h: AdjCarrier);
  * reconstructed delta after three Adam steps in all three detector modes, at the bound DESIGN.md states.
Measured amplitudes are rounded to float32 before either side sees them: the reference's datasets are complex64
(cnn_propagator/fullfield.py:139-140), so |prj| is a float32 quantity there too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _case(n, n_theta, mb, fp, seed=0, noise=0.02):
    rng = np.random.default_rng(seed)
    od = rng.uniform(0, 2e-6, size=(n, n, n))
    ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    idx = np.sort(rng.choice(n_theta, mb, replace=False))
    rot = np.stack([orc.apply_rotation(np.stack([od, ob], axis=3), coords[j]) for j in idx])
    one, zero = np.ones((n, n)), np.zeros((n, n))
    ref_wave, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, fp, rot[..., 0].shape,
                                                       return_probe_array=False)
    prj = np.zeros((n_theta, n, n))
    prj[idx] = (np.abs(ref_wave) * (1 + noise * rng.normal(size=ref_wave.shape))).astype(np.float32)
    return od, ob, coords, idx, prj, ref_wave, one, zero, rng


def _solver(n, n_theta, mb, fp, coords, od, ob, prj):
    from beyond_dof_amd.solver import FullfieldSolver
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp, coord_ls=coords)
    s.set_volume(od, ob)
    s.set_measurements(prj)
    return s


@pytest.mark.parametrize('n', [64, 128])
def test_plane_probe_far_field_gradient(n):
    import __graft_entry__ as entry
    entry.build()
    n_theta, mb, fp = 6, 2, 'inf'
    od, ob, coords, idx, prj, ref_wave, one, zero, _ = _case(n, n_theta, mb, fp)
    s = _solver(n, n_theta, mb, fp, coords, od, ob, prj)
    w = s.forward_angles(idx)
    assert rel(np.abs(w) ** 2, np.abs(ref_wave) ** 2) <= 1e-5            # forward intensities, north-star tolerance
    assert rel(w, ref_wave) <= 1e-5
    loss = s.loss_and_grad(idx)
    gd, gb = s.gradient_to_host()
    rl, rgd, rgb = orc.fullfield_loss_and_grad(od, ob, coords, idx, prj[idx], one, zero, 5000., 1e-7, free_prop_cm=fp, with_reg=False)
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4


def test_plane_probe_far_field_other_engines():
    """The same case on the generic (rocFFT) engine — directly, and as the engine a resident-plan size (72 x 72) is routed to
    for this probe / detector combination — against the oracle's batch gradient."""
    from beyond_dof_amd.engine import MultisliceEngine
    rng = np.random.default_rng(3)
    for (Y, X, S, kw) in ((64, 64, 24, dict(engine='generic')), (72, 72, 20, dict()), (60, 100, 12, dict())):
        B = 2
        delta = rng.uniform(0, 2e-6, size=(B, Y, X, S))
        beta = 0.1 * delta
        pr, pi = np.ones((Y, X)), np.zeros((Y, X))
        ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, 'inf', delta.shape)
        meas = (np.abs(ref) * (1 + 0.02 * rng.normal(size=ref.shape))).astype(np.float32)
        eng = MultisliceEngine(Y, X, S, B, with_grad=True, **kw)
        eng.set_physics(5000., 1e-7, 'inf')
        eng.set_probe(pr, pi)
        eng.set_object_batch(delta, beta)
        w = eng.forward(B)
        assert rel(w, ref) <= 1e-5
        loss = eng.loss_grad(B, meas)
        gd, gb = eng.grad_batch_to_host(B)
        rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas.astype(np.float64), 'inf')
        assert abs(loss - rl) <= 1e-5 * abs(rl), (Y, X)
        assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4, (Y, X, rel(gd, rgd), rel(gb, rgb))


@pytest.mark.parametrize('fp,n_theta,mb,bound_d,bound_step', [(1e-4, 8, 2, 2e-5, 0.1), (None, 5, 3, 2e-5, 0.1), ('inf', 6, 2, 3e-3, 2.5)])
def test_reconstructed_delta_after_three_adam_steps(fp, n_theta, mb, bound_d, bound_step):
    """Reconstructed delta against the oracle-driven loop (cnn_propagator/fullfield.py:345-362) after three Adam steps with
    the regulariser, mask and clip.  Real-space detectors: <= 2e-5 relative L2 (measured 1.4-1.6e-5), no voxel more than a
    tenth of a step away.  What separates that from the north star's 1e-5: Adam's first steps are
    lr * g / (|g| + 1e-8), so at voxels whose gradient is within float32 round-off of zero (absolute error 2e-6 of the
    typical |g|) the step differs by up to lr * dg / 1e-8.
    Plane probe + far field: the detector is one bright bin that does not see phase to first order; the delta-gradient is
    6000 x smaller than the beta-gradient and rests on the 2 % residuals of the weak scattered bins, where the float32
    forward sweep (2e-6 of the scattered wave after 128 slices) is amplified 50 x — gradient 1e-4, and Adam turns gradient
    noise at small |g| into whole +-lr steps: 1e-3 in delta after three steps.  Bound stated, not hidden."""
    n = 64
    od, ob, coords, idx, prj, _, one, zero, rng = _case(n, n_theta, mb, fp)
    s = _solver(n, n_theta, mb, fp, coords, od, ob, prj)
    mask = (rng.uniform(size=(n, n, n)) > 0.1).astype(np.float32)
    s.set_mask(mask)
    kw = dict(alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    lr = 1e-7
    x = np.array([od, ob])
    m = v = None
    for it in range(3):
        s.step(it, idx, lr, **kw)
        _, g1, g2 = orc.fullfield_loss_and_grad(x[0], x[1], coords, idx, prj[idx], one, zero, 5000., 1e-7, free_prop_cm=fp,
                                                with_reg=True, **kw)
        x, m, v = orc.apply_gradient_adam(x, np.array([g1, g2]), it, m, v, step_size=lr)
        x = np.clip(x * mask, 0, None)
    d, b = s.get_volume()
    assert rel(d, x[0]) <= bound_d, rel(d, x[0])
    assert np.abs(d - x[0]).max() <= bound_step * lr
    assert rel(b, x[1]) <= 1e-3, rel(b, x[1])


def test_regularizer_value_on_device():
    """alpha_d sum|delta| + alpha_b sum|beta| + gamma TV(delta) (cnn_propagator/fullfield.py:109-118) against the oracle's
    restatement (pinned by golden vector G5)."""
    n = 64
    od, ob, coords, idx, prj, _, one, zero, rng = _case(n, 4, 2, 1e-4)
    s = _solver(n, 4, 2, 1e-4, coords, od, ob, prj)
    d32, b32 = od.astype(np.float32).astype(np.float64), ob.astype(np.float32).astype(np.float64)
    want = orc.regularizer(d32, b32, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    got = s.regularizer(1.5e-8, 1.5e-9, 1e-11)
    assert abs(got - want) <= 1e-6 * abs(want), (got, want)
