"""The tiled propagation algorithm on the CPU (float64): what it costs in accuracy against the whole-field propagator
(golden-pinned np_funcs restatement), as a function of halo and stitch interval.  The reference's own tiled code is on a
branch that is not in the checkout (README.md:1-11): parity unpinned by reference code, the whole-field result is the oracle."""
import numpy as np

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _problem(n=96, S=24, seed=0):
    rng = np.random.default_rng(seed)
    delta = np.zeros((n, n, S))
    c = n // 4
    delta[c:-c, c:-c, :] = rng.uniform(0, 5e-5, size=(n - 2 * c, n - 2 * c, S))       # strong phase object inside a vacuum frame
    beta = 0.1 * delta
    yy, xx = np.mgrid[:n, :n]
    probe = np.exp(-((yy - n / 2.) ** 2 + (xx - n / 2.) ** 2) / (2 * (n / 6.) ** 2)).astype(np.complex128)
    return delta, beta, probe


def test_tile_kernel_equals_field_kernel_on_the_field_grid():
    for shape in ((8, 8), (9, 12), (64, 48)):
        a = orc.get_kernel(3.0, 0.248, [1., 1., 1.], shape)
        b = orc.get_kernel_tile(3.0, 0.248, [1., 1., 1.], shape, shape)
        assert np.abs(a - b).max() < 1e-13
    # and the product-side helper is the same function
    from beyond_dof_amd import util
    t = util.get_kernel_tile(2.0, 0.248, [1., 1., 1.], (32, 32), (96, 80))
    assert np.abs(t - orc.get_kernel_tile(2.0, 0.248, [1., 1., 1.], (32, 32), (96, 80))).max() < 1e-13


def test_one_tile_without_halo_is_the_whole_field():
    delta, beta, probe = _problem(32, 5)
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe.real, probe.imag, 5000., 1e-7, None, (1,) + delta.shape,
                                                  return_probe_array=False)
    # the oracle rounds the probe to complex64 (np_funcs.py:20-21): feed the tiled form the same probe
    p64 = probe.astype(np.complex64).astype(np.complex128)
    out = orc.tiled_multislice_propagate(delta, beta, p64, 5000., 1e-7, tile=32, halo=0, slices_per_exchange=5)
    assert rel(out, ref[0]) < 1e-13


def test_tiling_error_falls_with_the_halo():
    """Error against the whole field, worst-case object (white spectrum up to the band edge), 5 keV / 1 nm (the band-edge ray
    moves 0.124 px per slice).  A hard-edged tile is limited by the diffraction of the jump at its periodic boundary (1/x
    tail); with the outer half of the halo ramped to zero the error follows the halo."""
    delta, beta, probe = _problem(128, 32)
    p64 = probe.astype(np.complex64).astype(np.complex128)
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe.real, probe.imag, 5000., 1e-7, None, (1,) + delta.shape,
                                                  return_probe_array=False)
    run = lambda tile, halo, seg, taper, **kw: rel(orc.tiled_multislice_propagate(delta, beta, p64, 5000., 1e-7, tile, halo, seg,
                                                                                   taper=taper, **kw), ref[0])
    hard = run(64, 16, 16, 0)
    e16 = run(64, 16, 16, 8)
    e32 = run(128, 32, 32, 16)
    assert hard > 3e-4 and e16 < 0.1 * hard and e32 < 1e-5, (hard, e16, e32)
    ref2, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe.real, probe.imag, 5000., 1e-7, None, (1,) + delta.shape,
                                                   variant='tf_all', return_probe_array=False)
    both = orc.tiled_multislice_propagate(delta, beta, p64, 5000., 1e-7, 128, 32, 32, taper=16, variant='tf_all')
    assert rel(both, ref2[0]) < 1e-5


def test_hard_edged_wave_is_where_tiles_and_whole_field_part():
    """Documented limit: a jump in the wave makes the whole-field FFT propagator ring across the entire field (the inverse
    transform of its sharply band-limited symbol has alternating 0.04 / x^2 tails per slice); a tile never sees a source
    beyond its halo.  The two then differ by those tails, whatever the stitch interval — and agree again once the tile is the
    field or the wave is smooth."""
    n, S = 192, 24
    yy, xx = np.mgrid[:n, :n].astype(np.float64)
    hard = ((abs(yy - n / 2) < 40) & (abs(xx - n / 2) < 40)).astype(np.float64)
    smooth = np.exp(-((yy - n / 2) ** 2 + (xx - n / 2) ** 2) / (2 * 14. ** 2))
    vac = np.zeros((n, n, S))
    res = {}
    for name, probe in (('hard', hard), ('smooth', smooth)):
        p64 = probe.astype(np.complex64).astype(np.complex128)
        ref, _ = orc.multislice_propagate_batch_numpy(vac[None], vac[None], probe, np.zeros_like(probe), 5000., 1e-7, None, (1,) + vac.shape,
                                                      return_probe_array=False)
        res[name] = [rel(orc.tiled_multislice_propagate(vac, vac, p64, 5000., 1e-7, 96, 24, seg, taper=12), ref[0]) for seg in (24, 1)]
        res[name + '_one_tile'] = rel(orc.tiled_multislice_propagate(vac, vac, p64, 5000., 1e-7, 192, 24, 24, taper=12), ref[0])
    assert res['hard'][0] > 1e-4 and abs(res['hard'][1] / res['hard'][0] - 1) < 0.2        # not a matter of the stitch interval
    assert max(res['smooth']) < 1e-5 and res['hard_one_tile'] < 1e-5 and res['smooth_one_tile'] < 1e-5, res
