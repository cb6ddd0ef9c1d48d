"""cfg4 timing (BASELINE configs[3]): a 512^2 probe in a 4k x 4k zero-padded field through a 1024-slice zone-plate object,
whole-field FFT propagation on the rocFFT engine, forward model only (the tape + gradient of 1024 slices of 4096^2 would
need 2 x 137 GB).  usage: python tools/bench_cfg4.py [n=4096] [slices=1024]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from beyond_dof_amd import _lib, util  # noqa: E402
from beyond_dof_amd.engine import MultisliceEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
yy, xx = np.mgrid[:n, :n].astype(np.float32)
r2 = (yy - n / 2) ** 2 + (xx - n / 2) ** 2
# Fresnel zone plate, outermost zone 4 px wide at radius 256: zones at r_k^2 = k * r1^2 with r1^2 = 2 * 256 * 4
zone = (np.floor(r2 / (2 * 256 * 4.0)).astype(np.int64) % 2 == 0) & (r2 < 256.0 ** 2)
slab = np.where(zone, 5e-5, 0.0).astype(np.float32)
del yy, xx, r2
pr = np.zeros((n, n), dtype=np.float32)
c0 = n // 2 - 256
pr[c0:c0 + 512, c0:c0 + 512] = 1.0
eng = MultisliceEngine(n, n, S, 1, with_grad=False)
eng.set_physics(5000., 1e-7, 1e-4)
eng.set_probe(pr, np.zeros_like(pr))
# the object is the same 2-D slab in every slice: one (delta, beta) row set, a table that maps every slice to it
rows = np.ascontiguousarray(np.stack([slab.T, 0.1 * slab.T], axis=-1).astype(np.float32))          # [x][y] pairs
vol = _lib.DeviceBuffer.from_host(eng.ctx, rows)
tab = np.tile(np.arange(n, dtype=np.int32), (1, S, 1))                                         # [1 angle][S][x] -> row x
eng.set_volume(vol, n, n, _lib.DeviceBuffer.from_host(eng.ctx, tab), n, 1)
for it in range(2):
    eng.ctx.sync()
    t0 = time.perf_counter()
    out = eng.forward(1, angle_idx=[0], to_host=False)
    eng.ctx.sync()
    dt = time.perf_counter() - t0
w = eng._wave_to_host(out, 1)
print('%d^2 field, %d slices, forward: %.1f ms (%.0f slices/s, %.1f GB/s at 40 B/px forward model), |wave|^2 sum %.6e' %
      (n, S, dt * 1e3, S / dt, 40.0 * n * n * S / dt / 1e9, float((np.abs(w) ** 2).sum())))
