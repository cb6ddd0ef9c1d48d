"""Tiled ("pfft") propagation on the device (beyond_dof_amd.tiling.TiledPropagator: bdof_tiles_gather / bdof_forward_range /
bdof_tiles_scatter) against (a) the float64 restatement of the same tiled algorithm — float32 round-off only — and (b) the
whole-field propagator of np_funcs.py:15-65, the oracle the tiling error is measured against (the reference's own tiled code is
absent from the checkout: parity unpinned by reference code)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _problem(n, S, seed=0):
    rng = np.random.default_rng(seed)
    delta = np.zeros((n, n, S))
    c = n // 4
    delta[c:-c, c:-c, :] = rng.uniform(0, 5e-5, size=(n - 2 * c, n - 2 * c, S))
    beta = 0.1 * delta
    yy, xx = np.mgrid[:n, :n]
    probe = np.exp(-((yy - n / 2.) ** 2 + (xx - n / 2.) ** 2) / (2 * (n / 6.) ** 2))
    return delta, beta, probe


@pytest.mark.parametrize('variant', ['numpy_skip_last', 'tf_all'])
def test_tiled_forward_matches_the_algorithm_and_the_whole_field(variant):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, tile, halo, seg = 256, 40, 128, 32, 16
    delta, beta, probe = _problem(n, S)
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, slices_per_exchange=seg, variant=variant)
    assert tp.taper == 16 and tp.n_tiles == 16 and tp.segments() == [(0, 16), (16, 16), (32, 8)]
    tp.set_object(delta, beta)
    out = tp.forward(probe, np.zeros_like(probe))
    p64 = probe.astype(np.complex64).astype(np.complex128)
    alg = orc.tiled_multislice_propagate(delta, beta, p64, 5000., 1e-7, tile, halo, seg, taper=16, variant=variant)
    assert rel(out, alg) <= 2e-6                                   # the same algorithm in float64: round-off only
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe, np.zeros_like(probe), 5000., 1e-7, None,
                                                  (1,) + delta.shape, variant=variant, return_probe_array=False)
    assert rel(out, ref[0]) <= 1e-5                                # tiling error + round-off against the whole field
    assert rel(np.abs(out) ** 2, np.abs(ref[0]) ** 2) <= 1e-5


def test_tiled_slab_object_and_default_interval():
    """The same slab in every slice (thick zone plate, BASELINE cfg4's object) through set_object_slab; default stitch interval."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S = 512, 96
    yy, xx = np.mgrid[:n, :n].astype(np.float64)
    r2 = (yy - n / 2) ** 2 + (xx - n / 2) ** 2
    zone = (np.floor(r2 / (2 * 64 * 4.0)).astype(np.int64) % 2 == 0) & (r2 < 64.0 ** 2)     # outermost zone 4 px at radius 64
    slab = np.where(zone, 5e-5, 0.0)
    # a 192-pixel square probe in the zero-padded field with 32-pixel raised-cosine edges.  (A HARD-edged probe is a different
    # matter: the whole-field FFT propagator answers a jump with alternating-sign tails 0.04 / x^2 per slice — the inverse
    # transform of its sharply band-limited symbol — that reach across the whole field; tiles do not see sources beyond their
    # halo, and the two methods then differ by those tails: 1e-4 at 128-pixel tiles in float64, tests/test_oracle_tiling.py.)
    edge = lambda t: np.clip((96. + 16. - np.abs(t - n / 2)) / 32., 0., 1.)
    soft = lambda t: 0.5 - 0.5 * np.cos(np.pi * edge(t))
    probe = soft(yy) * soft(xx)
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=256, halo=48)
    assert tp.seg == int(0.5 * (48 - 24) / 0.124)
    tp.set_object_slab(slab, 0.1 * slab)
    out = tp.forward(probe, np.zeros_like(probe))
    delta = np.repeat(slab[:, :, None], S, axis=2)
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], 0.1 * delta[None], probe, np.zeros_like(probe), 5000., 1e-7, None,
                                                  (1,) + delta.shape, return_probe_array=False)
    assert rel(out, ref[0]) <= 2e-5, rel(out, ref[0])


def _torch_tiled_loss_grad(delta, beta, probe, meas, tile, halo, seg, taper, long_range=False):
    """The tiled algorithm (oracle.tiled_multislice_propagate, variant tf_all) in torch float64 on the CPU, differentiated by
    autograd: pins the device's hand-derived tiled adjoint.  long_range: with the correction of DESIGN §8 — per range
    new field = (the field's free-space step over the range) + stitch(tiles through the object - tiles through vacuum)."""
    import torch
    torch.set_num_threads(8)          # a GPU box shows all of its host's cores and grants 16: the default pool oversubscribes them
    fy, fx, S = delta.shape
    voxel = np.array([1., 1., 1.])
    lmbda = 1240. / 5000.
    k = 2. * orc.PI * voxel[-1] / lmbda
    h = torch.from_numpy(np.fft.ifftshift(orc.get_kernel_tile(voxel[-1], lmbda, voxel, (tile, tile), (fy, fx))))
    hf = torch.from_numpy(np.fft.ifftshift(orc.get_kernel(voxel[-1], lmbda, voxel, (fy, fx))))
    w1 = np.ones(tile)
    ramp = 0.5 - 0.5 * np.cos(np.pi * (np.arange(taper) + 0.5) / taper)
    w1[:taper], w1[tile - taper:] = ramp, ramp[::-1]
    win = torch.from_numpy(w1[:, None] * w1[None, :])
    # one leaf per slice: the gradient of a window cut then allocates (fy, fx) zeros, not (fy, fx, S) — 10 x faster at 1024^2 x 20
    td = [torch.tensor(np.ascontiguousarray(delta[:, :, z]), requires_grad=True) for z in range(S)]
    tb = [torch.tensor(np.ascontiguousarray(beta[:, :, z]), requires_grad=True) for z in range(S)]
    field = torch.tensor(probe.astype(np.complex64).astype(np.complex128))
    core = tile - 2 * halo
    oy, ox = orc.tile_origins(fy, tile, halo), orc.tile_origins(fx, tile, halo)
    for z0 in range(0, S, seg):
        nz = min(seg, S - z0)
        new = torch.zeros_like(field)
        for y0 in oy:
            ry = np.arange(y0, y0 + tile)
            iy = torch.from_numpy(ry % fy)
            for x0 in ox:
                rx = np.arange(x0, x0 + tile)
                ix = torch.from_numpy(rx % fx)
                w = field[iy][:, ix] * win
                if long_range:
                    wfree = torch.fft.ifft2(torch.fft.fft2(w) * torch.from_numpy(h.numpy() ** nz))
                inside = torch.from_numpy((((ry >= 0) & (ry < fy))[:, None] & ((rx >= 0) & (rx < fx))[None, :]).astype(np.float64))
                for z in range(z0, z0 + nz):
                    d = td[z][iy][:, ix] * inside
                    b = tb[z][iy][:, ix] * inside
                    w = w * torch.exp(1j * k * d) * torch.exp(-k * b)
                    w = torch.fft.ifft2(torch.fft.fft2(w) * h)
                ny_c, nx_c = min(core, fy - (y0 + halo)), min(core, fx - (x0 + halo))
                pad = torch.zeros_like(field)
                piece = w - wfree if long_range else w
                pad[y0 + halo:y0 + halo + ny_c, x0 + halo:x0 + halo + nx_c] = piece[halo:halo + ny_c, halo:halo + nx_c]
                new = new + pad
        if long_range:
            new = new + torch.fft.ifft2(torch.fft.fft2(field) * torch.from_numpy(hf.numpy() ** nz))
        field = new
    loss = torch.mean((torch.abs(field) - torch.from_numpy(meas)) ** 2)
    loss.backward()
    return (loss.item(), np.stack([t.grad.numpy() for t in td], axis=2), np.stack([t.grad.numpy() for t in tb], axis=2),
            field.detach().numpy())


def test_tiled_gradient_vs_autograd_of_the_algorithm():
    """TiledPropagator.loss_and_grad (stitch adjoints + tape-free range sweeps + overlap-add of the tiles' gradient rows)
    against torch autograd of the same tiled forward in float64, and against the whole-field oracle's gradient."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, tile, halo, seg = 128, 12, 64, 16, 5
    delta, beta, probe = _problem(n, S, seed=3)
    rng = np.random.default_rng(1)
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, slices_per_exchange=seg, variant='tf_all', with_grad=True)
    assert tp.n_tiles == 16 and tp.segments() == [(0, 5), (5, 5), (10, 2)]
    tp.set_object(delta, beta)
    out = tp.forward(probe, np.zeros_like(probe))
    meas = (np.abs(out) * (1 + 0.05 * rng.normal(size=out.shape))).astype(np.float32).astype(np.float64)
    loss, gd, gb = tp.loss_and_grad(probe, np.zeros_like(probe), meas)
    rl, rgd, rgb, rfield = _torch_tiled_loss_grad(delta, beta, probe, meas, tile, halo, seg, tp.taper)
    assert rel(out, rfield) <= 2e-6
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd) <= 2e-4 and rel(gb, rgb) <= 2e-4, (rel(gd, rgd), rel(gb, rgb))
    # the whole-field model's gradient (np_funcs forward + hand-derived adjoint): the tiling error of a 16-pixel halo on top
    wl, wgd, wgb = orc.multislice_loss_and_grad(delta[None], beta[None], probe, np.zeros_like(probe), 5000., 1e-7, meas[None], None, 'tf_all')
    assert abs(loss - wl) <= 1e-3 * abs(wl)
    assert rel(gd, wgd[0]) <= 2e-2, rel(gd, wgd[0])


def test_tiled_gradient_with_the_long_range_correction_vs_autograd():
    """TiledPropagator(with_grad=True, long_range=True): loss and gradient through the CORRECTED tiled model — per range
    psi_out = W psi_in + Stitch(T - F) Cut psi_in — against torch autograd of the same algorithm in float64 (three ranges), i.e. the
    hand-derived adjoint G_in = W^H G + Cut^H (T^H - F^H) Stitch^H G with the field-level adjoint in complex128; and the corrected
    exit wave is closer to the whole-field oracle than the uncorrected one."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, tile, halo, seg = 128, 12, 64, 16, 4
    delta, beta, probe = _problem(n, S, seed=5)
    rng = np.random.default_rng(2)
    zero = np.zeros_like(probe)
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe, zero, 5000., 1e-7, None, (1,) + delta.shape, variant='tf_all',
                                                  return_probe_array=False)
    errs = {}
    for lr in (False, True):
        tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, slices_per_exchange=seg, variant='tf_all', with_grad=True,
                             long_range=lr)
        assert tp.long_range == lr and tp.segments() == [(0, 4), (4, 4), (8, 4)]
        tp.set_object(delta, beta)
        out = tp.forward(probe, zero)
        errs[lr] = rel(out, ref[0])
        if not lr:
            continue
        meas = (np.abs(out) * (1 + 0.05 * rng.normal(size=out.shape))).astype(np.float32).astype(np.float64)
        loss, gd, gb = tp.loss_and_grad(probe, zero, meas)
        rl, rgd, rgb, rfield = _torch_tiled_loss_grad(delta, beta, probe, meas, tile, halo, seg, tp.taper, long_range=True)
        e = (rel(out, rfield), abs(loss - rl) / abs(rl), rel(gd, rgd), rel(gb, rgb))
        print('corrected tiled model vs autograd of it: wave', e[0], 'loss', e[1], 'g_delta', e[2], 'g_beta', e[3])
        assert e[0] <= 2e-6 and e[1] <= 1e-5 and e[2] <= 2e-4 and e[3] <= 2e-4, e
    print('exit wave vs the whole-field oracle: uncorrected', errs[False], 'corrected', errs[True])
    assert errs[True] < errs[False]


@pytest.mark.parametrize('precision', ['float32', 'float64'])
def test_long_range_correction_with_a_last_range_of_one_slice(precision):
    """Variant numpy_skip_last with a last stitch range of ONE slice (no transfer-function step in it: nothing to correct, the
    cores are written straight into the float64 field), non-square field, both tile precisions: the corrected exit wave against
    the whole-field oracle, and closer to it than the uncorrected one."""
    from beyond_dof_amd.tiling import TiledPropagator
    fy, fx, S, tile, halo, seg = 128, 192, 9, 64, 16, 4
    rng = np.random.default_rng(9)
    delta = np.zeros((fy, fx, S))
    delta[32:96, 48:144, :] = rng.uniform(0, 5e-5, size=(64, 96, S))
    beta = 0.1 * delta
    yy, xx = np.mgrid[:fy, :fx]
    probe = np.exp(-((yy - fy / 2.) ** 2 / (2 * (fy / 6.) ** 2) + (xx - fx / 2.) ** 2 / (2 * (fx / 6.) ** 2)))
    zero = np.zeros_like(probe)
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe, zero, 5000., 1e-7, None, (1,) + delta.shape, return_probe_array=False)
    err = {}
    for lr in (False, True):
        tp = TiledPropagator((fy, fx), S, 5000., 1e-7, tile=tile, halo=halo, slices_per_exchange=seg, long_range=lr, precision=precision)
        assert tp.segments() == [(0, 4), (4, 4), (8, 1)]
        tp.set_object(delta, beta)
        out = tp.forward(probe, zero)
        assert out.dtype == (np.complex128 if precision == 'float64' else np.complex64)
        err[lr] = rel(out, ref[0])
    print('last range of one slice,', precision, ': uncorrected', err[False], 'corrected', err[True])
    assert err[True] <= 5e-6 and err[True] < 0.5 * err[False], err


def test_tiled_gradient_slab_object():
    """The same slab in every slice: the gradient rows of all slices and all tiles accumulate into one (FY, FX) pair of maps."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, tile, halo, seg = 128, 6, 64, 16, 4
    rng = np.random.default_rng(7)
    slab = np.zeros((n, n))
    slab[32:96, 32:96] = rng.uniform(0, 5e-5, size=(64, 64))
    yy, xx = np.mgrid[:n, :n]
    probe = np.exp(-((yy - n / 2.) ** 2 + (xx - n / 2.) ** 2) / (2 * (n / 6.) ** 2))
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, slices_per_exchange=seg, variant='tf_all', with_grad=True)
    tp.set_object_slab(slab, 0.1 * slab)
    out = tp.forward(probe, np.zeros_like(probe))
    meas = (np.abs(out) * (1 + 0.05 * rng.normal(size=out.shape))).astype(np.float32).astype(np.float64)
    loss, gd, gb = tp.loss_and_grad(probe, np.zeros_like(probe), meas)
    delta = np.repeat(slab[:, :, None], S, axis=2)
    rl, rgd, rgb, _ = _torch_tiled_loss_grad(delta, 0.1 * delta, probe, meas, tile, halo, seg, tp.taper)
    assert abs(loss - rl) <= 1e-5 * abs(rl)
    assert rel(gd, rgd.sum(axis=2)) <= 2e-4 and rel(gb, rgb.sum(axis=2)) <= 2e-4


def _cfg4_inputs(n, r_zp=256.0, half=256.0):
    """BASELINE cfg4's inputs (tools/bench_cfg4.py): a zone plate of radius r_zp whose outermost zone is 4 pixels wide, delta
    5e-5 in the open zones, and a square probe of 2 * half pixels with 32-pixel raised-cosine edges, zero-padded into n^2."""
    t = np.arange(n, dtype=np.float64)
    r2 = (t[:, None] - n / 2) ** 2 + (t[None, :] - n / 2) ** 2
    zone = (np.floor(r2 / (2 * r_zp * 4.0)).astype(np.int64) % 2 == 0) & (r2 < r_zp ** 2)
    slab = np.where(zone, 5e-5, 0.0)
    edge = np.clip((half + 16. - np.abs(t - n / 2)) / 32., 0., 1.)
    soft = 0.5 - 0.5 * np.cos(np.pi * edge)
    return slab, soft[:, None] * soft[None, :]


_CFG4_REF = {}


def _cfg4_float64_whole_field(n, S):
    """The float64 whole-field propagation of np_funcs.py:36-43 of cfg4's probe through its zone-plate slab on the host (H from
    the oracle's golden-pinned get_kernel; scipy's threaded FFT: a 4096^2 complex128 transform takes ~0.15 s on the box's cores,
    96 slices half a minute).  Returns (exit wave of variant numpy_skip_last, of tf_all = one more step); computed once per session."""
    if (n, S) not in _CFG4_REF:
        import scipy.fft as sfft
        slab, probe = _cfg4_inputs(n)
        slab = slab.astype(np.float32).astype(np.float64)
        k = 2. * orc.PI * 1.0 / (1240. / 5000.)
        h = np.fft.ifftshift(orc.get_kernel(1.0, 1240. / 5000., np.array([1., 1., 1.]), (n, n)))
        cmod = np.exp(1j * k * slab) * np.exp(-k * 0.1 * slab)
        w = probe.astype(np.complex64).astype(np.complex128)
        for z in range(S):
            w = w * cmod
            if z < S - 1:
                w = sfft.ifft2(sfft.fft2(w, workers=-1) * h, workers=-1)
        _CFG4_REF[(n, S)] = (w, sfft.ifft2(sfft.fft2(w, workers=-1) * h, workers=-1))
    return _CFG4_REF[(n, S)]


@pytest.mark.parametrize('seg,ranges', [(None, 1), (32, 3)])
def test_cfg4_tiles_512_on_the_4096_field_vs_float64(seg, ranges):
    """BASELINE configs[3] at its stated shape: a 512^2 probe zero-padded into a 4096^2 field through a zone-plate slab, tiles of
    512^2 with a 64-pixel halo (121 tiles in one batch), 96 slices — against the float64 whole-field propagation of
    np_funcs.py:36-43.  seg=None: the default stitch interval (129 slices: one range, no stitch inside the stack);
    seg=32: THREE stitch ranges — bdof_tiles_scatter writes the 121 cores back into the 4096^2 field and bdof_tiles_gather
    re-cuts the tiles with fresh tapered halos twice inside the stack, the step that defines the method, at the stated shape.
    Forward wave and intensity within the north star's 1e-5."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S = 4096, 96
    slab, probe = _cfg4_inputs(n)
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=512, halo=64, slices_per_exchange=seg)
    assert tp.n_tiles == 121 and tp.core == 384 and tp.taper == 32 and len(tp.segments()) == ranges
    if seg is None:
        assert tp.seg == int(0.5 * 32 / 0.124)
    tp.set_object_slab(slab, 0.1 * slab)
    out = tp.forward(probe, np.zeros_like(probe))
    del tp
    w = _cfg4_float64_whole_field(n, S)[0]
    e_wave, e_int = rel(out, w), rel(np.abs(out) ** 2, np.abs(w) ** 2)
    print('cfg4 tiles 512/64 on 4096^2 x 96 slices in', ranges, 'stitch range(s) vs float64: wave', e_wave, 'intensity', e_int)
    assert e_wave <= 1e-5 and e_int <= 1e-5, (e_wave, e_int)      # measured 4.0e-6 / 3.8e-6 with one range (round 3)


def test_cfg4_tiled_loss_and_gradient_on_the_4096_field():
    """Loss + gradient through the tiles at cfg4's stated shape: 121 tiles of 512^2 on the 4096^2 field, 96 slices in TWO stitch
    ranges (bdof_tiles_scatter_adjoint -> bdof_adjoint_range at 121 x 512^2 -> bdof_tiles_grad_add -> bdof_tiles_gather_adjoint,
    last range first), variant tf_all.  No float64 reference differentiates a 4096^2 x 96 stack in seconds, so the gradient is
    checked through properties that do not depend on the size:
    (1) the loss of the tiled model (device reduction) equals the loss formed on the host from the tiled exit wave, and is
        within 1e-4 of the loss of the float64 WHOLE-FIELD model (the data are that model's exit wave + 2 % noise);
    (2) the gradient is a slab map that vanishes where no wave passes and is finite elsewhere;
    (3) descent: along -g the loss falls at the rate |g|^2 — the central difference (L(x + t g) - L(x - t g)) / (2 t |g|^2) of
        the device's own tiled forward model (losses formed on the host in float64) is 1 within 1e-3 (measured 1 + 2e-7) at a
        step that changes the loss by 5 % either way, at an object that is not the data's (0.9 of the plate's thickness).  The
        ratio is <grad L, g> / |g|^2: a wrong stitch adjoint, a range taken in the wrong order or tile gradients added at the
        wrong rows all move it away from 1."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, seg = 4096, 96, 48
    slab, probe = _cfg4_inputs(n)
    zero = np.zeros_like(probe)
    ref = _cfg4_float64_whole_field(n, S)[1]                                     # tf_all: a step after the last slice too
    rng = np.random.default_rng(8)
    meas = (np.abs(ref) * (1 + 0.02 * rng.normal(size=ref.shape))).astype(np.float32).astype(np.float64)
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=512, halo=64, slices_per_exchange=seg, variant='tf_all', with_grad=True)
    assert tp.n_tiles == 121 and tp.segments() == [(0, 48), (48, 48)]
    host_loss = lambda wave: float(np.mean((np.abs(wave.astype(np.complex128)) - meas) ** 2))
    # (1) at the data's own object: the tiled loss against the whole-field float64 loss
    tp.set_object_slab(slab, 0.1 * slab)
    out = tp.forward(probe, zero)
    assert rel(out, ref) <= 1e-5, rel(out, ref)
    l_true, _, _ = tp.loss_and_grad(probe, zero, meas)
    l_ref = float(np.mean((np.abs(ref) - meas) ** 2))
    assert abs(l_true - host_loss(out)) <= 1e-5 * l_true and abs(l_true - l_ref) <= 1e-4 * l_ref, (l_true, host_loss(out), l_ref)
    # (2), (3) at 0.9 of the plate's thickness
    x0 = 0.9 * slab
    tp.set_object_slab(x0, 0.1 * x0)
    loss, gd, gb = tp.loss_and_grad(probe, zero, meas)
    assert gd.shape == (n, n) and np.all(np.isfinite(gd)) and np.all(np.isfinite(gb))
    dark = np.abs(probe) == 0
    dark[:, 1024:3072] = False                                                   # keep only columns the wave cannot reach in 96 slices
    dark[1024:3072, :] = False
    assert np.abs(gd[dark]).max() <= 1e-6 * np.abs(gd).max()
    g2 = float(np.sum(gd.astype(np.float64) ** 2))                               # step along the delta-gradient alone (beta held)
    t = 0.05 * loss / g2
    lp = host_loss(_tiled_forward_with(tp, x0 + t * gd, 0.1 * x0, probe))
    lm = host_loss(_tiled_forward_with(tp, x0 - t * gd, 0.1 * x0, probe))
    ratio = (lp - lm) / (2 * t * g2)
    print('cfg4 tiled gradient on the 4096^2 field, two ranges: loss', loss, 'at the true object', l_true, 'whole-field float64', l_ref,
          'descent ratio', ratio, '(L+ - L-)/L', (lp - lm) / loss)
    assert abs((lp - lm) / loss - 0.1) <= 1e-3 and abs(ratio - 1) <= 1e-3, ratio


def test_cfg4_at_its_stated_depth_1024_slices():
    """BASELINE configs[3] as stated: 512^2 probe in the 4096^2 field, zone-plate slab, **1024 slices**, 512^2 tiles / 64-pixel halo.
    The host's float64 run of this depth takes 393 s, so the reference is the library's own float64 WHOLE-FIELD propagation
    (WholeFieldPropagator: rocFFT double transforms of the 4096^2 field, float64 modulation) — first validated here against the
    host's float64 run at 96 slices (<= 1e-12), then run at 1024.
    * fused float32 tiles with dithered constants and transfer function, the long-range correction with the field in float64,
      stitched every 16 slices (64 ranges), the full wave through the float32 sweeps: exit wave and intensity within the north
      star's 1e-5 (measured 5.8e-6) with 64-pixel halos (121 tiles), with 24-pixel halos (81 tiles) and with the vacuum tiles left out;
    * the DEFAULT plan: the same with every tile riding on its own carrier field (bdof_set_range_carrier) — within 2e-6 (7.7e-7);
    * precision='float64' (rocFFT double tiles): within 2e-6 (measured 9.1e-7: what is left of the tiling error);
    * float64 tiles WITHOUT the correction at round 3's interval: the tiling error proper, 2.2e-5 — the whole-field
      propagator's long-range tails, which no tile sees; on record with its bounds, it is what the correction removes."""
    from beyond_dof_amd.tiling import TiledPropagator, WholeFieldPropagator
    n = 4096
    slab, probe = _cfg4_inputs(n)
    zero = np.zeros_like(probe)
    wf = WholeFieldPropagator((n, n), 96, 5000., 1e-7)
    wf.set_object_slab(slab, 0.1 * slab)
    e96 = rel(wf.forward(probe, zero), _cfg4_float64_whole_field(n, 96)[0])
    del wf
    assert e96 <= 1e-12, e96
    S = 1024
    wf = WholeFieldPropagator((n, n), S, 5000., 1e-7)
    wf.set_object_slab(slab, 0.1 * slab)
    ref = wf.forward(probe, zero)
    del wf
    res = {}
    for name, kw in (('default', dict(halo=64, skip_vacuum=False, carrier=False)), ('auto_halo', dict(halo=24, skip_vacuum=False, carrier=False)),
                     ('skip_vacuum', dict(halo=24, carrier=False)), ('carrier', {}),
                     ('float64', dict(halo=64, precision='float64')), ('no_correction', dict(halo=64, precision='float64', long_range=False))):
        tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=512, **kw)
        if name == 'default':
            assert tp.precision == 'float32' and tp.long_range and tp.seg == 16 and len(tp.segments()) == 64 and tp.n_tiles == 121
        elif name in ('auto_halo', 'skip_vacuum'):   # 24-pixel halos: what corrected ranges of full-amplitude float32 sweeps need
            assert tp.halo == 24 and tp.taper == 12 and tp.long_range and tp.seg == 16 and tp.n_tiles == 81 and not tp.carrier
        elif name == 'carrier':                      # the constructor's defaults: per-tile carriers, 48-pixel halos, vacuum tiles left out
            assert tp.carrier and tp.skip_vacuum and tp.halo == 48 and tp.precision == 'float32' and tp.seg == 16 and tp.n_tiles == 100
        tp.set_object_slab(slab, 0.1 * slab)
        if name == 'skip_vacuum':                    # only the tiles whose window touches the zone plate run
            assert len(tp._active) == 64 and all(len(a) == 6 for a in tp._active), [len(a) for a in tp._active]
        if name == 'carrier':
            assert all(len(a) <= 9 for a in tp._active), [len(a) for a in tp._active]
        out = tp.forward(probe, zero)
        del tp
        res[name] = (rel(out, ref), rel(np.abs(out) ** 2, np.abs(ref) ** 2))
    print('cfg4 at 1024 slices vs the float64 whole field (wave, intensity):', res, '; float64 engine vs the host at 96 slices:', e96)
    assert max(res['default']) <= 1e-5 and max(res['auto_halo']) <= 1e-5, res          # measured 5.82e-6 / 5.83e-6 (intensity 5.9e-6 / 6.3e-6)
    # 6 of the 81 tiles touch the zone plate.  Its result is another draw of the same float32 rounding noise, not the same numbers:
    # 16 slices of the fused kernels carry ~7e-7 of round-off whose realisation changes with the last bit of the input (measured:
    # two histories 1e-15 apart leave T psi 6.8e-7 apart after one range), and the 75 vacuum tiles move the double-precision field
    # by 1e-19.  5.8e-6 with every tile, 6.5e-6 with the six (intensity 6.3e-6 / 7.7e-6): both 0.7e-6 x sqrt(64 ranges).
    assert max(res['skip_vacuum']) <= 1e-5, res
    # per-tile carriers (the default): each tile of a range rides on its own free-space propagation in double and the float32
    # kernels carry the scattered part only — the round-off above is gone and what is left is the tiling error of the halo:
    # measured 7.7e-7 (intensity 8.7e-7) at 48 pixels, the figure of float64 tiles
    assert max(res['carrier']) <= 2e-6, res
    assert max(res['float64']) <= 2e-6, res
    assert 1.5e-5 <= res['no_correction'][0] <= 3e-5, res


def _tiled_forward_with(tp, delta2d, beta2d, probe):
    tp.set_object_slab(delta2d, beta2d)
    return tp.forward(probe, np.zeros_like(probe))


def test_cfg4_tile_size_two_ranges_and_gradient_vs_oracle():
    """The 512^2 / 64 plan on a field the float64 references can differentiate: 1024^2 (9 tiles), 20 slices of a (FY, FX, S) object
    under the zone plate's aperture in two stitch ranges (16 + 4).  Forward against the whole-field oracle; loss and gradient
    (bdof_adjoint_range at tile 512, stitch adjoints, overlap-add of the tiles' gradient rows) against torch autograd of the
    same tiled algorithm in float64 — the device's hand-derived tiled adjoint at cfg4's tile size — and against the oracle's
    gradient of the WHOLE-FIELD model.  The latter differs by the tiling itself, not by round-off: the adjoint of the stitch
    hands every tile a hard-edged piece of the (noise-driven, white) adjoint field, whose edge fringes wrap around the tile's
    period — in float64 on the CPU 3.3e-3 / 1.7e-3 / 1.2e-3 of the gradient at halos of 16 / 32 / 48 pixels (128^2 tiles) while
    the forward wave and the loss agree to 3e-6 / 2e-7 (MEASUREMENTS.md, round 3, cfg4): the exact gradient of a loss that is
    2e-7 from the whole field's is still 1e-3 from its gradient."""
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, seg = 1024, 20, 16
    slab, probe = _cfg4_inputs(n, r_zp=128.0, half=160.0)
    rng = np.random.default_rng(4)
    delta = slab[:, :, None] * rng.uniform(0.5, 1.0, size=(1, 1, S))      # the plate, its strength varying from slice to slice
    beta = 0.1 * delta
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=512, halo=64, slices_per_exchange=seg, variant='tf_all', with_grad=True)
    assert tp.n_tiles == 9 and tp.segments() == [(0, 16), (16, 4)]
    tp.set_object(delta, beta)
    out = tp.forward(probe, np.zeros_like(probe))
    zero = np.zeros_like(probe)
    ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe, zero, 5000., 1e-7, None, (1,) + delta.shape, variant='tf_all',
                                                  return_probe_array=False)
    assert rel(out, ref[0]) <= 1e-5, rel(out, ref[0])
    meas = (np.abs(ref[0]) * (1 + 0.02 * rng.normal(size=probe.shape))).astype(np.float32).astype(np.float64)
    loss, gd, gb = tp.loss_and_grad(probe, zero, meas)
    rl, rgd, rgb, rfield = _torch_tiled_loss_grad(delta, beta, probe, meas, 512, 64, seg, tp.taper)
    e = (rel(out, rfield), abs(loss - rl) / abs(rl), rel(gd, rgd), rel(gb, rgb))
    print('tile 512 vs autograd of the tiled algorithm: wave', e[0], 'loss', e[1], 'g_delta', e[2], 'g_beta', e[3])
    assert e[0] <= 2e-6 and e[1] <= 1e-5 and e[2] <= 2e-4 and e[3] <= 2e-4, e
    wl, wgd, wgb = orc.multislice_loss_and_grad(delta[None], beta[None], probe, zero, 5000., 1e-7, meas[None], None, 'tf_all')
    w = (abs(loss - wl) / abs(wl), rel(gd, wgd[0]), rel(gb, wgb[0]))
    print('tile 512 vs the whole-field oracle: loss', w[0], 'g_delta', w[1], 'g_beta', w[2])
    assert w[0] <= 1e-5 and w[1] <= 3e-3 and w[2] <= 3e-3, w


def test_range_carrier_stack_in_one_call_equals_the_step_by_step_one():
    """bdof_range_carrier_build (p_z = F^-1(H^z F p_0): one forward transform, the spectra by a running product, one batched inverse
    transform) against the same stack made slice by slice with bdof_fields_free_step + bdof_c_convert, and against numpy."""
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import _lib, util
    from beyond_dof_amd.engine import MultisliceEngine
    B, T, nz = 3, 128, 6
    rng = np.random.default_rng(8)
    p0 = (rng.normal(size=(B, T, T)) + 1j * rng.normal(size=(B, T, T))).astype(np.complex128)          # [b][x][y]
    eng = MultisliceEngine(T, T, 8, B, with_grad=False)
    ctx, lib, h = eng.ctx, eng.lib, eng.h
    hk = np.fft.ifftshift(util.get_kernel(1., 0.248, np.array([1., 1., 1.]), (T, T))) / float(T * T)
    ht = np.ascontiguousarray(hk.T.astype(np.complex128))                                               # [kx][ky]
    hbuf = _lib.DeviceBuffer.from_host(ctx, ht)
    a = _lib.DeviceBuffer.from_host(ctx, p0)
    one = _lib.DeviceBuffer(ctx, nz * B * T * T * 8, np.complex64, (nz, B, T, T))
    ctx.check(lib.bdof_range_carrier_build(h, a.ptr, one.ptr, B, T, T, hbuf.ptr, nz))
    b = _lib.DeviceBuffer.from_host(ctx, p0)
    steps = _lib.DeviceBuffer(ctx, nz * B * T * T * 8, np.complex64, (nz, B, T, T))
    for j in range(nz):
        ctx.check(lib.bdof_c_convert(h, steps.ptr + j * B * T * T * 8, b.ptr, B * T * T, 0))
        if j < nz - 1:
            ctx.check(lib.bdof_fields_free_step(h, b.ptr, B, T, T, hbuf.ptr, 0, 1))
    ctx.sync()
    s1, s2 = one.download(), steps.download()
    ref = np.stack([np.fft.ifft2(np.fft.fft2(p0) * (ht * T * T) ** z) for z in range(nz)])              # axes (x, y) <-> (kx, ky)
    e = (rel(s1, s2), rel(s1, ref), rel(s2, ref))
    print('carrier stack: one call vs step by step', e[0], ' vs numpy float64:', e[1:])
    assert e[0] <= 2e-7 and e[1] <= 1e-7 and e[2] <= 1e-7, e        # complex64 storage: 6e-8
