"""cfg4 (BASELINE configs[3]): a 512^2 probe in a 4k x 4k zero-padded field through a 1024-slice zone-plate object, forward
model.  Tiled propagation on the fused kernels (beyond_dof_amd.tiling) against the whole-field transform on the rocFFT engine:
time per 1024 slices and the deviation of the exit waves, for several tile / halo choices (the error-vs-halo curve of DESIGN).
usage: python tools/bench_cfg4.py [n=4096] [slices=1024] [json out]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from beyond_dof_amd import _lib  # noqa: E402
from beyond_dof_amd.engine import MultisliceEngine  # noqa: E402
from beyond_dof_amd.tiling import TiledPropagator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out_json = sys.argv[3] if len(sys.argv) > 3 else None
yy, xx = np.mgrid[:n, :n].astype(np.float32)
r2 = (yy - n / 2) ** 2 + (xx - n / 2) ** 2
# Fresnel zone plate, outermost zone 4 px wide at radius 256: zones at r_k^2 = k * r1^2 with r1^2 = 2 * 256 * 4
zone = (np.floor(r2 / (2 * 256 * 4.0)).astype(np.int64) % 2 == 0) & (r2 < 256.0 ** 2)
slab = np.where(zone, 5e-5, 0.0).astype(np.float32)
del yy, xx, r2
# the 512^2 probe in the zero-padded field, with 32-pixel raised-cosine edges: a hard-edged one would make the whole-field FFT
# propagator ring across the entire field (alternating 0.04 / x^2 tails of its band-limited kernel), which tiles do not copy
t = np.arange(n, dtype=np.float32)
edge = np.clip((256. + 16. - np.abs(t - n / 2)) / 32., 0., 1.)
soft = (0.5 - 0.5 * np.cos(np.pi * edge)).astype(np.float32)
pr = np.ascontiguousarray(soft[:, None] * soft[None, :])
res = {'field': n, 'slices': S, 'runs': []}

# ---- whole field, rocFFT engine (the oracle of the tiling error on the device; float32) ---------------------------------
eng = MultisliceEngine(n, n, S, 1, with_grad=False)
eng.set_physics(5000., 1e-7, None)
eng.set_probe(pr, np.zeros_like(pr))
rows = np.ascontiguousarray(np.stack([slab.T, 0.1 * slab.T], axis=-1).astype(np.float32))          # [x][y] pairs
vol = _lib.DeviceBuffer.from_host(eng.ctx, rows)
tab = np.tile(np.arange(n, dtype=np.int32), (1, S, 1))                                         # [1 angle][S][x] -> row x
eng.set_volume(vol, n, n, _lib.DeviceBuffer.from_host(eng.ctx, tab), n, 1)
for it in range(2):
    eng.ctx.sync()
    t0 = time.perf_counter()
    out = eng.forward(1, angle_idx=[0], to_host=False)
    eng.ctx.sync()
    dt_whole = time.perf_counter() - t0
whole = eng._wave_to_host(out, 1)[0]
print('whole field %d^2 x %d slices (rocFFT engine): %.1f ms (%.0f slices/s)' % (n, S, dt_whole * 1e3, S / dt_whole))
res['whole_field_ms'] = dt_whole * 1e3
del eng, vol, out

rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for tile, halo in ((512, 32), (512, 64), (512, 96), (1024, 64), (1024, 128)):
    if tile > n:
        continue
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo)
    tp.set_object_slab(slab, 0.1 * slab)
    tp.field.upload(np.ascontiguousarray(pr.T.astype(np.complex64)))
    tp.forward_device()
    tp.ctx.sync()
    tp.field.upload(np.ascontiguousarray(pr.T.astype(np.complex64)))
    tp.ctx.sync()
    t0 = time.perf_counter()
    tp.forward_device()
    tp.ctx.sync()
    dt = time.perf_counter() - t0
    w = np.ascontiguousarray(tp.field.download().T)
    px = tp.n_tiles * tile * tile
    run = {'tile': tile, 'halo': halo, 'taper': tp.taper, 'tiles': tp.n_tiles, 'slices_per_exchange': tp.seg, 'ms': dt * 1e3,
           'overhead_px': px / float(n * n), 'GBps_40B_model': 40.0 * px * S / dt / 1e9,
           'rel_err_wave_vs_whole_field': rel(w, whole), 'rel_err_intensity': rel(np.abs(w) ** 2, np.abs(whole) ** 2)}
    res['runs'].append(run)
    print('tiles %4d^2 halo %3d (taper %2d, %3d tiles, stitch every %3d slices): %7.1f ms = %.2fx whole field; wave err %.2e, intensity err %.2e, '
          '%.0f GB/s at 40 B/px' % (tile, halo, tp.taper, tp.n_tiles, tp.seg, dt * 1e3, dt / dt_whole, run['rel_err_wave_vs_whole_field'],
                                   run['rel_err_intensity'], run['GBps_40B_model']))
    del tp
if out_json:
    json.dump(res, open(out_json, 'w'), indent=1)

# ---- which of the two float32 results is closer to float64?  (whole-field restatement of np_funcs.py:36-43 with scipy's
# threaded FFT, at a reduced slice count: 4096^2 complex128 transforms take seconds each on the host) ----------------------
n_or = int(os.environ.get('CFG4_ORACLE_SLICES', '0'))
if n_or > 0:
    import scipy.fft as sfft
    from beyond_dof_amd import util
    S2 = n_or
    h = np.fft.ifftshift(util.get_kernel(1.0, 0.248, [1., 1., 1.], (n, n)))
    cmod = np.exp(1j * (2 * util.PI * 1.0 / 0.248) * slab.astype(np.float64)) * np.exp(-(2 * util.PI * 1.0 / 0.248) * 0.1 * slab.astype(np.float64))
    w = pr.astype(np.complex64).astype(np.complex128)
    t0 = time.perf_counter()
    for z in range(S2):
        w = w * cmod
        if z < S2 - 1:
            w = sfft.ifft2(sfft.fft2(w, workers=-1) * h, workers=-1)
        if z % 64 == 63:
            print('  float64 host propagation: slice %d of %d, %.0f s' % (z + 1, S2, time.perf_counter() - t0), flush=True)
    print('float64 whole field, %d slices on the host: %.0f s' % (S2, time.perf_counter() - t0))
    eng = MultisliceEngine(n, n, S2, 1, with_grad=False)
    eng.set_physics(5000., 1e-7, None)
    eng.set_probe(pr, np.zeros_like(pr))
    vol = _lib.DeviceBuffer.from_host(eng.ctx, rows)
    eng.set_volume(vol, n, n, _lib.DeviceBuffer.from_host(eng.ctx, np.ascontiguousarray(tab[:, :S2])), n, 1)
    whole2 = eng.forward(1, angle_idx=[0])[0]
    res['float64_check'] = {'slices': S2, 'whole_field_rocfft_vs_float64': rel(whole2, w), 'tiled': []}
    print('  whole field (rocFFT, float32) vs float64: %.2e' % res['float64_check']['whole_field_rocfft_vs_float64'])
    del eng, vol
    # the last entry repeats 512 / 64 with ONE nearest-rounded twiddle table per transform (BDOF_TW_DITHER=0) instead of the
    # dithered copies: what the coherent table error is worth at this depth
    for tile, halo, dither in ((512, 32, None), (512, 64, None), (512, 96, None), (512, 128, None), (1024, 64, None), (1024, 128, None), (512, 64, '0')):
        if dither is not None:
            os.environ['BDOF_TW_DITHER'] = dither
        tp = TiledPropagator((n, n), S2, 5000., 1e-7, tile=tile, halo=halo)
        tp.set_object_slab(slab, 0.1 * slab)
        o = tp.forward(pr, np.zeros_like(pr))
        e = rel(o, w)
        res['float64_check']['tiled'].append({'tile': tile, 'halo': halo, 'one_table': dither is not None, 'vs_float64': e,
                                              'vs_whole_field_float32': rel(o, whole2)})
        print('  tiles %d^2 halo %d%s vs float64: %.2e   (vs the float32 whole field: %.2e)' % (tile, halo, ' (one twiddle table)' if dither is not None else '',
                                                                                             e, rel(o, whole2)), flush=True)
        del tp
        os.environ.pop('BDOF_TW_DITHER', None)
    if out_json:
        json.dump(res, open(out_json, 'w'), indent=1)

# ---- forward + adjoint through the tiles (variant tf_all; tape-free range sweeps) --------------------------------------------
if os.environ.get('CFG4_GRAD'):
    for tile, halo in ((512, 64), (512, 32)):
        tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, variant='tf_all', with_grad=True)
        tp.set_object_slab(slab, 0.1 * slab)
        exit_wave = tp.forward(pr, np.zeros_like(pr))
        rng = np.random.default_rng(0)
        meas = (np.abs(exit_wave) * (1 + 0.02 * rng.normal(size=exit_wave.shape))).astype(np.float32)
        loss, gd, gb = tp.loss_and_grad(pr, np.zeros_like(pr), meas)       # buffers, first launches; host copies of the gradient
        used = tp.ctx.mem_used() / 2.0 ** 30
        meas_dev = _lib.DeviceBuffer.from_host(tp.ctx, np.ascontiguousarray(meas.T))
        probe_dev = np.ascontiguousarray(pr.T.astype(np.complex64))
        tp.field.upload(probe_dev)                                         # inputs resident in HBM when the timed region starts
        tp.ctx.sync()
        t0 = time.perf_counter()
        loss2, _ = tp.loss_and_grad_device(meas_dev)
        tp.ctx.sync()
        dt = time.perf_counter() - t0
        assert loss2 == loss
        px = tp.n_tiles * tile * tile
        run = {'tile': tile, 'halo': halo, 'tiles': tp.n_tiles, 'slices_per_exchange': tp.seg, 'fwd_adjoint_ms': dt * 1e3,
               'slice_steps_per_s': S / dt, 'GBps_136B_model': 136.0 * px * S / dt / 1e9, 'hbm_used_GiB': used, 'loss': loss,
               'grad_norms': [float(np.linalg.norm(gd)), float(np.linalg.norm(gb))]}
        res.setdefault('fwd_adjoint', []).append(run)
        print('tiles %d^2 halo %d: loss + gradient of %d slices of %d^2 in %.0f ms (%.0f slice-steps/s of the whole field; %.0f GB/s at '
              '40 + 96 B per tile pixel; %.1f GiB of HBM in use); loss %.4e' % (tile, halo, S, n, dt * 1e3, S / dt, run['GBps_136B_model'], used, loss))
        del tp
    if out_json:
        json.dump(res, open(out_json, 'w'), indent=1)
