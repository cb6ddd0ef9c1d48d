"""cfg4 (BASELINE configs[3]): a 512^2 probe in a 4k x 4k zero-padded field through a 1024-slice zone-plate object.
Tiled propagation on the fused kernels (beyond_dof_amd.tiling) against the whole field: time per stack and the deviation of the
exit waves from the library's float64 whole-field propagation (WholeFieldPropagator: rocFFT double; checked against the host's
float64 run at 96 slices in tests/test_gpu_tiling.py), for the plans of DESIGN §8.
usage: python tools/bench_cfg4.py [n=4096] [slices=1024] [json out]        (env CFG4_GRAD=1: loss + gradient timings too)"""
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
from beyond_dof_amd.tiling import TiledPropagator, WholeFieldPropagator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out_json = sys.argv[3] if len(sys.argv) > 3 else None
t = np.arange(n, dtype=np.float64)
r2 = (t[:, None] - n / 2) ** 2 + (t[None, :] - n / 2) ** 2
# Fresnel zone plate, outermost zone 4 px wide at radius 256: zones at r_k^2 = k * r1^2 with r1^2 = 2 * 256 * 4
slab = np.where((np.floor(r2 / (2 * 256 * 4.0)).astype(np.int64) % 2 == 0) & (r2 < 256.0 ** 2), 5e-5, 0.0)
del r2
# the 512^2 probe in the zero-padded field, with 32-pixel raised-cosine edges (a hard-edged one makes the whole-field FFT propagator
# ring across the entire field, which tiles do not copy)
edge = np.clip((256. + 16. - np.abs(t - n / 2)) / 32., 0., 1.)
soft = 0.5 - 0.5 * np.cos(np.pi * edge)
pr = np.ascontiguousarray(soft[:, None] * soft[None, :])
zero = np.zeros_like(pr)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
res = {'field': n, 'slices': S, 'build': _lib.build_id(), 'runs': []}

# ---- float64 whole field on the device: the reference of every error below ------------------------------------------------
wf = WholeFieldPropagator((n, n), S, 5000., 1e-7)
wf.set_object_slab(slab, 0.1 * slab)
ref = wf.forward(pr, zero)
t0 = time.perf_counter()
ref = wf.forward(pr, zero)
res['whole_field_float64_ms'] = (time.perf_counter() - t0) * 1e3          # includes the 2 x 268-MB host copies
print('whole field float64 (rocFFT double): %.0f ms' % res['whole_field_float64_ms'], flush=True)
del wf

# ---- whole field, rocFFT engine in float32 -----------------------------------------------------------------------------------
eng = MultisliceEngine(n, n, S, 1, with_grad=False)
eng.set_physics(5000., 1e-7, None)
eng.set_probe(pr, zero)
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
res['whole_field_float32_ms'] = dt_whole * 1e3
res['whole_field_float32_vs_float64'] = rel(whole, ref)
print('whole field float32 (rocFFT engine): %.1f ms, %.2e from float64' % (dt_whole * 1e3, res['whole_field_float32_vs_float64']), flush=True)
del eng, vol, out, whole

plans = [('default: float32 tiles on per-tile carriers + correction, float64 field; vacuum tiles left out', dict(tile=512)),   # halo='auto': 48
         ('the same, halo 24', dict(tile=512, halo=24)),
         ('the same, halo 32', dict(tile=512, halo=32)),
         ('the same, halo 64', dict(tile=512, halo=64)),
         ('the same, halo 96', dict(tile=512, halo=96)),
         ('halo 24, the full wave through the float32 sweeps (no carriers), vacuum tiles left out', dict(tile=512, halo=24, carrier=False)),
         ('the same with every tile running', dict(tile=512, halo=24, skip_vacuum=False, carrier=False)),
         ('every tile, halo 64 (the halo of plain stitching), no carriers', dict(tile=512, halo=64, skip_vacuum=False, carrier=False)),
         ('float32 tiles, no correction (round 3 plan)', dict(tile=512, halo=64, long_range=False)),
         ('float64 tiles + correction', dict(tile=512, halo=64, precision='float64')),
         ('1024^2 tiles, halo 64', dict(tile=1024, halo=64))]
for name, kw in plans:
    if kw['tile'] > n:
        continue
    tp = TiledPropagator((n, n), S, 5000., 1e-7, **kw)
    tp.set_object_slab(slab, 0.1 * slab)
    o = tp.forward(pr, zero)
    tp.field.upload(np.ascontiguousarray(pr.T.astype(np.complex64).astype(tp.field.dtype)))
    tp.ctx.sync()
    t0 = time.perf_counter()
    tp.forward_device()
    tp.ctx.sync()
    dt = time.perf_counter() - t0
    run = {'plan': name, 'tile': kw['tile'], 'halo': tp.halo, 'tiles': tp.n_tiles, 'slices_per_exchange': tp.seg, 'ranges': len(tp.segments()),
           'long_range': tp.long_range, 'precision': tp.precision, 'tiles_run_per_range': (float(np.mean([len(a) for a in tp._active]))
                                                                                         if (tp._active is not None and tp.skip_vacuum and tp.long_range and not tp.dbl) else float(tp.n_tiles)),
           'ms': dt * 1e3, 'wave_vs_float64': rel(o, ref),
           'intensity_vs_float64': rel(np.abs(o) ** 2, np.abs(ref) ** 2)}
    res['runs'].append(run)
    print('%-100s %4d tiles, every %3d slices: %7.1f ms  wave %.2e intensity %.2e' % (
        name, tp.n_tiles, tp.seg, dt * 1e3, run['wave_vs_float64'], run['intensity_vs_float64']), flush=True)
    del tp
if out_json:
    json.dump(res, open(out_json, 'w'), indent=1)

# ---- forward + adjoint through the tiles (variant tf_all; tape-free range sweeps) --------------------------------------------
if os.environ.get('CFG4_GRAD'):
    for name, kw in (('corrected model', dict(tile=512)), ('corrected model, halo 64', dict(tile=512, halo=64)),
                     ('no correction', dict(tile=512, halo=64, long_range=False))):
        tp = TiledPropagator((n, n), S, 5000., 1e-7, variant='tf_all', with_grad=True, **kw)
        tp.set_object_slab(slab, 0.1 * slab)
        exit_wave = tp.forward(pr, zero)
        rng = np.random.default_rng(0)
        meas = (np.abs(exit_wave) * (1 + 0.02 * rng.normal(size=exit_wave.shape))).astype(np.float32)
        loss, gd, gb = tp.loss_and_grad(pr, zero, meas)       # buffers, first launches; host copies of the gradient
        used = tp.ctx.mem_used() / 2.0 ** 30
        meas_dev = _lib.DeviceBuffer.from_host(tp.ctx, np.ascontiguousarray(meas.T))
        tp.field.upload(np.ascontiguousarray(pr.T.astype(np.complex64)))      # inputs resident in HBM when the timed region starts
        tp.ctx.sync()
        t0 = time.perf_counter()
        loss2, _ = tp.loss_and_grad_device(meas_dev)
        tp.ctx.sync()
        dt = time.perf_counter() - t0
        assert loss2 == loss
        run = {'plan': name, 'tiles': tp.n_tiles, 'slices_per_exchange': tp.seg, 'fwd_adjoint_ms': dt * 1e3, 'slice_steps_per_s': S / dt,
               'hbm_used_GiB': used, 'loss': loss, 'grad_norms': [float(np.linalg.norm(gd)), float(np.linalg.norm(gb))]}
        res.setdefault('fwd_adjoint', []).append(run)
        print('tiled loss + gradient, %s (every %d slices): %d slices of %d^2 in %.0f ms (%.0f slice-steps/s of the whole field; %.1f GiB of HBM '
              'in use); loss %.4e' % (name, tp.seg, S, n, dt * 1e3, S / dt, used, loss), flush=True)
        del tp
    if out_json:
        json.dump(res, open(out_json, 'w'), indent=1)
