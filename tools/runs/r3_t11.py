import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import bdof_oracle as orc
from beyond_dof_amd.tiling import TiledPropagator
import test_gpu_tiling as T
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
n, S = 1024, 80
slab, probe = T._cfg4_inputs(n, r_zp=128.0, half=160.0)
rng = np.random.default_rng(4)
delta = slab[:, :, None] * rng.uniform(0.5, 1.0, size=(1, 1, S)); beta = 0.1 * delta
zero = np.zeros_like(probe)
ref, _ = orc.multislice_propagate_batch_numpy(delta[None], beta[None], probe, zero, 5000., 1e-7, None, (1,) + delta.shape, variant='tf_all', return_probe_array=False)
meas = (np.abs(ref[0]) * (1 + 0.02 * rng.normal(size=probe.shape))).astype(np.float32).astype(np.float64)
wl, wgd, wgb = orc.multislice_loss_and_grad(delta[None], beta[None], probe, zero, 5000., 1e-7, meas[None], None, 'tf_all')
print('|wgd| by slice (first, mid, last):', np.linalg.norm(wgd[0][..., 0]), np.linalg.norm(wgd[0][..., 40]), np.linalg.norm(wgd[0][..., 79]), flush=True)
for tile, halo, seg in ((512, 64, 64), (512, 64, 16), (512, 96, 64), (256, 64, 64), (512, 64, 80)):
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=tile, halo=halo, slices_per_exchange=seg, variant='tf_all', with_grad=True)
    tp.set_object(delta, beta)
    out = tp.forward(probe, zero)
    loss, gd, gb = tp.loss_and_grad(probe, zero, meas)
    e = gd - wgd[0]
    per_slice = [np.linalg.norm(e[..., z]) / np.linalg.norm(wgd[0][..., z]) for z in (0, 20, 40, 63, 64, 79)]
    # where: error energy by x position modulo the core
    ex = np.sqrt(np.sum(e ** 2, axis=(0, 2)))
    top = np.argsort(ex)[-5:]
    print('tile', tile, 'halo', halo, 'seg', seg, 'n_tiles', tp.n_tiles, 'fwd', rel(out, ref[0]), 'loss', abs(loss - wl) / wl, 'gd', rel(gd, wgd[0]), 'gb', rel(gb, wgb[0]),
          'per-slice', ['%.1e' % v for v in per_slice], 'worst x columns', top, flush=True)
    del tp
