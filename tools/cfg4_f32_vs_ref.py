"""cfg4 error budget, device side: TiledPropagator (float32 fused kernels) for a list of tile:halo:taper:seg plans against the
float64 whole-field exit wave written by tools/cfg4_f64_torch.py.
usage: python tools/cfg4_f32_vs_ref.py REF.npy [n=4096] [S=1024] [tile:halo:taper:seg ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from beyond_dof_amd.tiling import TiledPropagator  # noqa: E402

ref = np.load(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cfgs = [tuple(int(v) for v in a.split(':')) for a in sys.argv[4:]]
kw = {}
kw['precision'] = os.environ.get('CFG4_PRECISION', 'float32')
kw['long_range'] = bool(int(os.environ.get('CFG4_LONG_RANGE', '0')))


def cfg4_inputs(n, r_zp=256.0, half=256.0):
    t = np.arange(n, dtype=np.float64)
    r2 = (t[:, None] - n / 2) ** 2 + (t[None, :] - n / 2) ** 2
    zone = (np.floor(r2 / (2 * r_zp * 4.0)).astype(np.int64) % 2 == 0) & (r2 < r_zp ** 2)
    slab = np.where(zone, 5e-5, 0.0)
    edge = np.clip((half + 16. - np.abs(t - n / 2)) / 32., 0., 1.)
    soft = 0.5 - 0.5 * np.cos(np.pi * edge)
    return slab, soft[:, None] * soft[None, :]


slab, probe = cfg4_inputs(n)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for tile, halo, taper, seg in cfgs:
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, taper=taper, slices_per_exchange=seg, **kw)
    tp.set_object_slab(slab, 0.1 * slab)
    o = tp.forward(probe, np.zeros_like(probe))
    tp.field.upload(np.ascontiguousarray(probe.T.astype(np.complex64).astype(tp.field.dtype)))
    tp.ctx.sync()
    t0 = time.perf_counter()
    tp.forward_device()
    tp.ctx.sync()
    dt = time.perf_counter() - t0
    print('device %s%s tiles %d/%d taper %d stitch every %d (%d tiles, %d ranges): wave %.3e intensity %.3e   %.0f ms' % (
        kw['precision'], ' + long-range correction' if kw['long_range'] else '', tile, halo, taper, seg, tp.n_tiles, len(tp.segments()), rel(o, ref), rel(np.abs(o) ** 2, np.abs(ref) ** 2), dt * 1e3), flush=True)
    del tp
