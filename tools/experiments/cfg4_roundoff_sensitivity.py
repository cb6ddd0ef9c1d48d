"""After three ranges with two histories (1e-16 apart): how many float32 tile inputs of the six central tiles differ, and what does
one more range make of it?"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import __graft_entry__ as entry
entry.build()
from beyond_dof_amd.tiling import TiledPropagator
import test_gpu_tiling as t
n, S = 4096, 48
slab, probe = t._cfg4_inputs(n)
zero = np.zeros_like(probe)
tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=512, halo=24, long_range=True, slices_per_exchange=16, variant='tf_all')
tp.set_object_slab(slab, 0.1 * slab)
A = tp._active[0]
far = np.array([i for i in range(tp.n_tiles) if i not in set(A.tolist())][:10], dtype=np.int32)
A2 = np.sort(np.concatenate([A, far]))
lib, h, T = tp.lib, tp.h, tp.tile
got = []
for sets in ([A, A, A], [A2, A2, A2]):
    tp._active = [np.ascontiguousarray(s) for s in sets]
    tp.forward(probe, zero)
    f64 = tp.field64.download().copy()
    tp._active = [np.ascontiguousarray(A)] * 3
    B, a, xo, yo, va, vx, vy = tp._range_tiles(0)
    tp.ctx.check(lib.bdof_tiles_gather_mixed(h, tp.field64.ptr, n, n, tp.tiles_in.ptr, B, T, T, xo, yo, tp.taper))
    tin = tp.tiles_in.download()[:B].copy()
    t_tab, _ = tp._tables(16)
    tp.ctx.check(lib.bdof_forward_range_h(h, B, va, vx, vy, 0, 1, tp.tiles_in.ptr, tp.tiles_free.ptr, 1, t_tab.ptr))
    tp.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, 0, 16, tp.tiles_in.ptr, tp.tiles_out.ptr, 1))
    tp.ctx.sync()
    got.append((f64, tin, tp.tiles_free.download()[:B].copy(), tp.tiles_out.download()[:B].copy()))
f0, f1 = got[0][0], got[1][0]
print('field after 3 ranges: %d of %d complex values differ; max |diff| %.3e' % (np.count_nonzero(f0 != f1), f0.size, np.abs(f0 - f1).max()))
for j, what in ((1, 'float32 tile inputs'), (2, 'T_free'), (3, 'T')):
    a0, a1 = got[0][j], got[1][j]
    print('%-20s: %d of %d values differ; max |diff| %.3e; |diff| / |value| = %.3e' % (what, np.count_nonzero(a0 != a1), a0.size, np.abs(a0 - a1).max(),
                                                                                     np.linalg.norm(a0 - a1) / np.linalg.norm(a0)))
d = (got[0][3].astype(np.complex128) - got[0][2]) - (got[1][3].astype(np.complex128) - got[1][2])
print('T - T_free: |diff| / |T| = %.3e' % (np.linalg.norm(d) / np.linalg.norm(got[0][3])))
