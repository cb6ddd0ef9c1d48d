"""cfg5 timing: 256^3 object, gaussian probe, 20 x 20 scan positions, far field, all positions of one angle
per Adam step.  usage: python tools/bench_ptycho.py [probe_size] [n_pos_side] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from beyond_dof_amd import util  # noqa: E402
from beyond_dof_amd.solver import PtychoSolver  # noqa: E402
gaussian_probe = util.gaussian_probe

ps = int(sys.argv[1]) if len(sys.argv) > 1 else 72
side = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, n_theta = 256, 8
rng = np.random.default_rng(5)
pos = np.array([(y, x) for y in np.arange(side) * 12 + 14 for x in np.arange(side) * 12 + 14])
mb = len(pos)
pr, pi = gaussian_probe((ps, ps), 6., 6., 0.5)
t0 = time.time()
coords = util.rotation_lookup([n, n, n], n_theta)
s = PtychoSolver([n, n, n], [ps, ps], pos, n_theta, mb, 5000., 1e-7, pr, pi, coord_ls=coords)
d = rng.random((n, n, n), dtype=np.float32) * 1e-6
s.set_volume(d, 0.1 * d)
meas = np.abs(rng.normal(1.0, 0.1, size=(n_theta, mb, ps, ps))).astype(np.float32) * ps
s.set_measurements(meas)               # the amplitudes of all angles resident in HBM (inputs resident when the timed region starts)
print('setup %.1f s' % (time.time() - t0), flush=True)
s.reset_moments()
for it in range(1 + steps):
    if it == 1:
        s.ctx.sync()
        t0 = time.perf_counter()
    s.step(it, it % n_theta, np.arange(mb), None, 1e-7)       # what reconstruct_ptychography runs per minibatch
s.ctx.sync()
dt = (time.perf_counter() - t0) / steps
px = ps * ps
# the LDS-resident kernel moves 40 B per pixel per slice-step (modulation factor 8 + tape write 8; tape read 8 + factor 8 +
# gradient write 8) — that is the figure its roofline fraction is priced at, not the streaming engines' 104 B
print('probe %d^2, %d positions, %d slices: %.2f ms per Adam step, %.0f slice-steps/s, %.1f GB/s at the resident kernel\'s 40 B/px (%.3f of 8 TB/s)' %
      (ps, mb, n, dt * 1e3, mb * n / dt, 40.0 * px * mb * n / dt / 1e9, 40.0 * px * mb * n / dt / 8e12))

# what adjoint_precision='first-step' (reconstruct_ptychography's default) adds: the first minibatch of every epoch through the
# model's float64 path on the same context (bdof_loss_grad_tf_f64: rocFFT double precision, unfused)
if os.environ.get('BDOF_BENCH_FIRST_STEP'):
    del s
    s = PtychoSolver([n, n, n], [ps, ps], pos, n_theta, mb, 5000., 1e-7, pr, pi, coord_ls=coords, adjoint64='first')
    s.set_volume(d, 0.1 * d)
    s.set_measurements(meas)
    used = s.ctx.mem_used() / 2.0 ** 30
    for rep in range(2):
        s.reset_moments()
        s.ctx.sync()
        t0 = time.perf_counter()
        s.step(0, 0, np.arange(mb), None, 1e-7)
        s.ctx.sync()
        t1 = time.perf_counter()
        s.step(1, 1, np.arange(mb), None, 1e-7)
        s.ctx.sync()
        t2 = time.perf_counter()
    print('first step of an epoch through the float64 path: %.1f ms (the following float32 step %.2f ms); %.1f GiB of HBM in use with its buffers'
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, used))
