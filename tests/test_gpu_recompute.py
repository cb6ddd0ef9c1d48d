"""Tape-free adjoint (bdof_configure flag 16, MultisliceEngine(recompute=True)): the forward wave is marched back beside the
adjoint field (phi_{z-1} = P^H (phi_z / c_z)) instead of being kept per slice — autograd's tape
(cnn_propagator/fullfield.py:329) without the memory.  Same loss, same gradient as the taped form and as the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _grad(recompute, delta, beta, pr, pi, fp, variant, meas):
    from beyond_dof_amd.engine import MultisliceEngine
    B, Y, X, S = delta.shape
    eng = MultisliceEngine(Y, X, S, B, with_grad=True, engine='streaming', recompute=recompute)
    eng.set_physics(5000., 1e-7, fp, variant=variant)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    loss = eng.loss_grad(B, meas)
    gd, gb = eng.grad_batch_to_host(B)
    return loss, gd, gb


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
@pytest.mark.parametrize('variant', ['numpy_skip_last', 'tf_all'])
@pytest.mark.parametrize('probe', ['plane', 'gaussian'])
def test_recompute_matches_tape_and_oracle(fp, variant, probe):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import util
    rng = np.random.default_rng(5)
    B, Y, X, S = 3, 64, 128, 20
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    if probe == 'plane':
        pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    else:
        pr, pi = util.gaussian_probe((Y, X), 12., 12., 0.5)          # localised probe: the carrier-field instances
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, variant=variant,
                                                  return_probe_array=False)
    meas = (np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))).astype(np.float32)
    l0, gd0, gb0 = _grad(False, delta, beta, pr, pi, fp, variant, meas)
    l1, gd1, gb1 = _grad(True, delta, beta, pr, pi, fp, variant, meas)
    assert l1 == l0                                                       # the forward sweep is the same launches
    assert rel(gd1, gd0) <= 2e-5 and rel(gb1, gb0) <= 2e-5
    rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas.astype(np.float64), fp, variant=variant)
    assert abs(l1 - rl) <= 1e-5 * abs(rl)
    tol = 2e-4
    assert rel(gd1, rgd) <= tol and rel(gb1, rgb) <= tol, (rel(gd1, rgd), rel(gb1, rgb))


@pytest.mark.parametrize('S', [1, 2, 3])
def test_recompute_short_stacks(S):
    rng = np.random.default_rng(S)
    B, Y, X = 2, 64, 64
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    meas = (1 + 0.05 * rng.normal(size=(B, Y, X))).astype(np.float32)
    _, gd0, gb0 = _grad(False, delta, beta, pr, pi, 1e-4, 'numpy_skip_last', meas)
    _, gd1, gb1 = _grad(True, delta, beta, pr, pi, 1e-4, 'numpy_skip_last', meas)
    assert rel(gd1, gd0) <= 2e-5 and rel(gb1, gb0) <= 2e-5


def test_recompute_full_depth_512():
    """512 x 512 x 512 slices, one wavefield, both forms against each other: the float32 march back over 511 slices."""
    rng = np.random.default_rng(9)
    B, Y, X, S = 1, 512, 512, 512
    delta = rng.uniform(0, 2e-6, size=(B, Y, X, S)).astype(np.float32)
    beta = (0.1 * delta).astype(np.float32)
    pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    meas = (1 + 0.02 * rng.normal(size=(B, Y, X))).astype(np.float32)
    l0, gd0, gb0 = _grad(False, delta, beta, pr, pi, 1e-4, 'numpy_skip_last', meas)
    l1, gd1, gb1 = _grad(True, delta, beta, pr, pi, 1e-4, 'numpy_skip_last', meas)
    assert l1 == l0
    assert rel(gd1, gd0) <= 1e-4 and rel(gb1, gb0) <= 1e-4, (rel(gd1, gd0), rel(gb1, gb0))


def test_recompute_through_the_solver_and_sub_batch_streams():
    """FullfieldSolver(recompute=True): fused rotation, two sub-batch streams, Adam — same volume as the taped run."""
    from beyond_dof_amd.solver import FullfieldSolver
    rng = np.random.default_rng(1)
    n, n_theta, mb = 64, 8, 4
    coords = orc.rotation_lookup([n, n, n], n_theta)
    meas = 1 + 0.05 * rng.normal(size=(n_theta, n, n))
    init_d = np.clip(rng.normal(8.7e-7, 1e-7, size=(n, n, n)), 0, None)
    vols = []
    for rc in (False, True):
        s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, coord_ls=coords, recompute=rc)
        s.eng.set_streams(2)
        s.set_measurements(meas)
        s.set_volume(init_d, 0.1 * init_d)
        s.reset_moments()
        for i, idx in enumerate([np.arange(0, 4), np.arange(4, 8)]):
            s.step(i, idx, 1e-7, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
        vols.append(s.get_volume())
        g = s.gradient_to_host()
        vols.append(g)
    assert rel(vols[3][0], vols[1][0]) <= 2e-5 and rel(vols[3][1], vols[1][1]) <= 2e-5
    assert rel(vols[2][0], vols[0][0]) <= 2e-5
