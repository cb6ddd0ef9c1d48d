"""Generic-size engine (rocFFT + point-wise kernels): sizes without a hand-written FFT plan — notably the 72 x 72 probe
of the reference's ptychography drivers (cnn_propagator/reconstruct_ptycho.py:106) — against the oracle, and against the
fused engine on a size both support."""
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


@pytest.mark.parametrize('Y,X', [(72, 72), (60, 100), (45, 33)])
@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
@pytest.mark.parametrize('variant', ['numpy_skip_last', 'tf_all'])
def test_generic_sizes_vs_oracle(engine_mod, Y, X, fp, variant):
    rng = np.random.default_rng(1)
    B, S = 3, 5
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    if fp == 'inf':
        pr, pi = orc.gaussian_probe((Y, X), Y / 8., Y / 8., 0.5)
    else:
        pr, pi = 1 + 0.1 * rng.normal(size=(Y, X)), 0.1 * rng.normal(size=(Y, X))
    eng = engine_mod.MultisliceEngine(Y, X, S, B, with_grad=True)
    eng.set_physics(5000., 1e-7, fp, variant=variant)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    wave = eng.forward(B)
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, variant=variant,
                                                  return_probe_array=False)
    assert rel(np.abs(wave) ** 2, np.abs(ref) ** 2) <= 1e-5
    assert rel(wave, ref) <= 5e-6
    meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
    loss = eng.loss_grad(B, meas)
    gd, gb = eng.grad_batch_to_host(B)
    rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp, variant)
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
def test_generic_engine_cross_checks_fused_engine(engine_mod, fp):
    """Same inputs through the fused hand-written FFT kernels and through rocFFT: two independent device paths."""
    rng = np.random.default_rng(2)
    B, Y, X, S = 2, 128, 64, 12
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    pr, pi = orc.gaussian_probe((Y, X), 12., 12., 0.5)
    out = []
    for force in (False, True):
        eng = engine_mod.MultisliceEngine(Y, X, S, B, with_grad=True, force_generic=force)
        eng.set_physics(5000., 1e-7, fp)
        eng.set_probe(pr, pi)
        eng.set_object_batch(delta, beta)
        w = eng.forward(B)
        meas = np.abs(w) * (1 + 0.05 * rng.normal(size=w.shape)) if not out else out[0][3]
        loss = eng.loss_grad(B, meas)
        gd, gb = eng.grad_batch_to_host(B)
        out.append((w, loss, gd, meas, gb))
    assert rel(out[1][0], out[0][0]) <= 2e-6
    assert abs(out[1][1] - out[0][1]) <= 1e-5 * abs(out[0][1])
    assert rel(out[1][2], out[0][2]) <= 1e-4 and rel(out[1][4], out[0][4]) <= 1e-4


def test_ptychography_with_72x72_probe(engine_mod):
    """BASELINE.json configs[4] geometry in small: 72 x 72 probe windows over a rotated volume, far field."""
    from beyond_dof_amd.solver import PtychoSolver
    rng = np.random.default_rng(0)
    n, n_theta, psz = 96, 4, (72, 72)
    pos = np.array([(y, x) for y in (20, 60) for x in (15, 48, 80)])
    od = rng.uniform(0, 2e-5, size=(n, n, n))
    ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    prr, pii = orc.gaussian_probe(psz, 6., 6., 0.5)
    s = PtychoSolver((n, n, n), psz, pos, n_theta, len(pos), 5000., 1e-7, prr, pii, coord_ls=coords)
    s.set_volume(od, ob)
    i_theta = 1
    sel = np.arange(len(pos))
    pad, half = orc.ptycho_pad_amounts(pos, psz, (n, n, n))
    rot = orc.apply_rotation(np.stack([od, ob], axis=3), coords[i_theta])
    obj_pad = np.pad(rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
    subs = np.stack([obj_pad[p[0] + pad[0, 0] - half[0]:p[0] + pad[0, 0] - half[0] + psz[0],
                             p[1] + pad[1, 0] - half[1]:p[1] + pad[1, 0] - half[1] + psz[1]] for p in pos])
    ref, _ = orc.multislice_propagate_batch_numpy(subs[..., 0], subs[..., 1], prr, pii, 5000., 1e-7, 'inf', subs[..., 0].shape,
                                                  return_probe_array=False)
    w = s.forward(i_theta, sel)
    # 96 slices in float32 with a localised probe (nothing for carrier splitting to split off): round-off grows like
    # sqrt(S) and reaches 1.3e-5 on the intensities here; the 1e-5 bound holds up to ~50 slices in this regime
    assert rel(np.abs(w) ** 2, np.abs(ref) ** 2) <= 2e-5
    meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
    loss = s.loss_and_grad(i_theta, sel, meas)
    gd, gb = s.gradient_to_host()
    rl, rgd, rgb = orc.ptycho_loss_and_grad(od, ob, coords[i_theta], pos, pos, meas, prr, pii, psz, 5000., 1e-7)
    assert abs(loss - rl) <= 5e-5 * rl
    assert rel(gd, rgd) <= 1e-3 and rel(gb, rgb) <= 1e-3


@pytest.mark.parametrize('n,S,fp', [(2048, 3, None), (4096, 2, 1e-4)])
def test_cfg4_field_sizes_vs_oracle(engine_mod, n, S, fp):
    """cfg4 (BASELINE configs[3]): a 512^2 probe zero-padded into a 2k / 4k field, whole-field FFT propagation — the sizes
    beyond the fused plans run on the rocFFT engine.  Forward wave and loss + gradient against the float64 oracle."""
    rng = np.random.default_rng(n)
    delta = np.zeros((1, n, n, S), dtype=np.float32)
    c0 = n // 2 - 256
    delta[0, c0:c0 + 512, c0:c0 + 512, :] = rng.uniform(0, 5e-5, size=(512, 512, S))      # zone-plate-like strong phase object
    beta = 0.1 * delta
    yy, xx = np.mgrid[:n, :n]
    inside = (abs(yy - n // 2) < 256) & (abs(xx - n // 2) < 256)
    pr = inside * np.exp(-((yy - n / 2.) ** 2 + (xx - n / 2.) ** 2) / (2 * 90. ** 2))       # 512^2 probe in the padded field
    pi = np.zeros((n, n))
    eng = engine_mod.MultisliceEngine(n, n, S, 1, with_grad=True)
    eng.set_physics(5000., 1e-7, fp)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    wave = eng.forward(1)
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, return_probe_array=False)
    assert rel(wave, ref) <= 5e-6
    meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
    loss = eng.loss_grad(1, meas)
    gd, gb = eng.grad_batch_to_host(1)
    rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp)
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4
