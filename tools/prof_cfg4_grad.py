"""Development aid: one tiled loss + gradient at cfg4 size for rocprofv3."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_dof_amd.tiling import TiledPropagator
n, S = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 258
yy, xx = np.mgrid[:n, :n].astype(np.float32)
r2 = (yy - n / 2) ** 2 + (xx - n / 2) ** 2
slab = np.where((np.floor(r2 / 2048.0).astype(np.int64) % 2 == 0) & (r2 < 65536.0), 5e-5, 0.0).astype(np.float32)
t = np.arange(n, dtype=np.float32)
soft = (0.5 - 0.5 * np.cos(np.pi * np.clip((272. - np.abs(t - n / 2)) / 32., 0., 1.))).astype(np.float32)
pr = np.ascontiguousarray(soft[:, None] * soft[None, :])
tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=512, halo=64, variant='tf_all', with_grad=True)
tp.set_object_slab(slab, 0.1 * slab)
meas = np.ones((n, n), dtype=np.float32)
for _ in range(2):
    print(tp.loss_and_grad(pr, np.zeros_like(pr), meas)[0])
import time
lib, h, p = tp.lib, tp.h, tp.idx.ptr
a, xo, yo = p, p + 4 * tp.n_tiles, p + 8 * tp.n_tiles
T, B = tp.tile, tp.n_tiles
def timed(name, f):
    tp.ctx.sync(); t0 = time.perf_counter(); f(); tp.ctx.sync(); print('%-28s %8.1f ms' % (name, (time.perf_counter() - t0) * 1e3))
z0, nz = tp.segments()[0]
end = tp._ends[0]
timed('gather', lambda: tp.ctx.check(lib.bdof_tiles_gather(h, tp.field.ptr, tp.fx, tp.fy, tp.tiles_in.ptr, B, T, T, xo, yo, tp.taper)))
timed('forward_range %d' % nz, lambda: tp.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, tp.tiles_in.ptr, end.ptr, 1)))
timed('scatter', lambda: tp.ctx.check(lib.bdof_tiles_scatter(h, end.ptr, tp.field.ptr, tp.fx, tp.fy, B, T, T, xo, yo, tp.halo, tp.halo)))
timed('scatter_adjoint', lambda: tp.ctx.check(lib.bdof_tiles_scatter_adjoint(h, tp.field.ptr, tp.fx, tp.fy, tp.tiles_in.ptr, B, T, T, xo, yo, tp.halo, tp.halo)))
timed('adjoint_range %d' % nz, lambda: tp.ctx.check(lib.bdof_adjoint_range(h, B, a, xo, yo, z0, nz, end.ptr, tp.tiles_in.ptr, tp.tiles_out.ptr, tp._grot.ptr)))
from beyond_dof_amd._lib import DeviceBuffer
gvol = DeviceBuffer.zeros(tp.ctx, tp.eng._keep['obj'].shape, np.float32)
timed('grad_add', lambda: tp.ctx.check(lib.bdof_tiles_grad_add(h, tp._grot.ptr, gvol.ptr, B, T, T, xo, yo, z0, nz)))
timed('gather_adjoint', lambda: tp.ctx.check(lib.bdof_tiles_gather_adjoint(h, tp.tiles_out.ptr, tp.field.ptr, tp.fx, tp.fy, B, T, T, xo, yo, tp.taper)))
t0 = time.perf_counter(); tp.loss_and_grad(pr, np.zeros_like(pr), meas); print('whole loss_and_grad %.1f ms' % ((time.perf_counter() - t0) * 1e3))
