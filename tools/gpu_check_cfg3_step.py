"""cfg3 at its full size through the SOLVER: 512^3 volume, the 200-angle rotation tables, a two-angle minibatch — rotation gather,
forward through all 512 slices, loss, adjoint sweep, rotation adjoint and one Adam step with L1 + TV, mask and clip — against the
float64 oracle (the oracle needs a few minutes and ~20 GB of host memory: a tool, not a test; the 256^3 twin is
tests/test_gpu_parity.py::test_cfg2_full_size_fullfield_step_vs_oracle).
usage: python tools/gpu_check_cfg3_step.py [n=512] [n_theta=200]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd.solver import FullfieldSolver  # noqa: E402
from scipy.ndimage import uniform_filter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_theta = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fp, lr = 1e-4, 1e-7
reg = dict(alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
rng = np.random.default_rng(3)
od = uniform_filter(rng.random((n, n, n)) * 2e-6, size=3, mode='wrap')          # bench.py's charcoal-like volume
ob = 0.1 * od
t0 = time.time()
coords = orc.rotation_lookup([n, n, n], n_theta)
idx = np.array([7 % n_theta, (n_theta * 2) // 3 + 1])
one, zero = np.ones((n, n)), np.zeros((n, n))
rot = np.stack([orc.apply_rotation(np.stack([od, ob], axis=3), coords[j]) for j in idx])
ref_wave, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, fp, rot[..., 0].shape, return_probe_array=False)
del rot
prj = np.zeros((n_theta, n, n), dtype=np.float32)
meas = np.abs(ref_wave) * (1 + 0.05 * rng.normal(size=ref_wave.shape))
prj[idx] = meas
print('oracle forward %.0f s' % (time.time() - t0), flush=True)
s = FullfieldSolver(n, n, n, n_theta, len(idx), 5000., 1e-7, free_prop_cm=fp, coord_ls=coords)
s.set_volume(od, ob)
s.set_measurements(prj)
print('forward intensities, two angles of %d^3 x %d slices: rel err %.2e' % (n, n, rel(np.abs(s.forward_angles(idx)) ** 2, np.abs(ref_wave) ** 2)), flush=True)
loss = s.loss_and_grad(idx)
gd, gb = s.gradient_to_host()
t0 = time.time()
rl, rgd, rgb = orc.fullfield_loss_and_grad(od, ob, coords, idx, prj[idx].astype(np.float64), one, zero, 5000., 1e-7, free_prop_cm=fp, with_reg=False)
print('oracle loss + gradient %.0f s' % (time.time() - t0), flush=True)
print('loss rel err %.2e; volume gradient (rotation adjoint included) delta %.2e beta %.2e' % (abs(loss - rl) / rl, rel(gd, rgd), rel(gb, rgb)), flush=True)
# one Adam step (first of an epoch: the step in which a float32 gradient shows) with the regulariser, mask, clip
mask = np.ones((n, n, n), dtype=np.float32)
mask[:, :4, :] = 0
s.set_mask(mask)
s.reset_moments()
s.step(0, idx, lr, reg['alpha_d'], reg['alpha_b'], reg['gamma'])
d1, b1 = s.get_volume()
rd, rb = orc.regularizer_grad(od, ob, **reg)
x, _, _ = orc.apply_gradient_adam(np.array([od, ob]), np.array([rgd + rd, rgb + rb]), 0, None, None, step_size=lr)
x = np.clip(x * mask, 0, None)
dev = np.abs(d1 - x[0])
print('after one Adam step: delta rel err %.2e (max %.4f of a step, %.2e of the voxels more than 0.01 of a step away); beta rel err %.2e' % (
    rel(d1, x[0]), dev.max() / lr, float(np.mean(dev > 0.01 * lr)), rel(b1, x[1])))
