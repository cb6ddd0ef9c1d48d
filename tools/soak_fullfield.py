"""Development aid: a medium-size reconstruct_fullfield run end to end (synthetic data from the product's own forward model)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_dof_amd import h5io
from beyond_dof_amd.fullfield import reconstruct_fullfield
from beyond_dof_amd.solver import FullfieldSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_theta = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(0)
z, y, x = np.mgrid[:n, :n, :n]
d = np.zeros((n, n, n), dtype=np.float32)
for _ in range(8):
    c = rng.uniform(n * 0.3, n * 0.7, size=3); r = rng.uniform(n * 0.05, n * 0.15)
    d += (1e-6 * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * r ** 2))).astype(np.float32)
with tempfile.TemporaryDirectory() as td:
    os.chdir(td)
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4)
    s.set_volume(d, 0.1 * d)
    prj = s.forward_angles(np.arange(n_theta))
    del s
    os.makedirs('case')
    h5io.write_dataset('case/data.h5', 'exchange/data', prj.astype(np.complex64))
    t0 = time.time()
    rd, rb = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=3, learning_rate=1e-7, minibatch_size=mb, energy_ev=5000,
                                   psize_cm=1e-7, free_prop_cm=1e-4, save_path='case', output_folder='out', shrink_cycle=None, seed=3,
                                   alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    print('reconstruct_fullfield %d^3, %d angles, 3 epochs: %.1f s; files: %s' % (n, n_theta, time.time() - t0, sorted(os.listdir('case/out'))))
    print('delta: recon mean %.3e vs truth mean %.3e; corr %.3f' % (rd.mean(), d.mean(), np.corrcoef(rd.ravel(), d.ravel())[0, 1]))
