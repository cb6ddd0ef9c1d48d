"""A complete small reconstruction, the way the reference's driver scripts run one (cnn_propagator/reconstruct_fullfield.py):
simulate a full-field dataset of a phantom with the product's forward model, write it as exchange/data, reconstruct with
reconstruct_fullfield and compare with the phantom.  The detector distance decides how much of the phase the intensities
carry: at 10 um (free_prop_cm=1e-3) 100 epochs recover delta to a correlation of 0.988 with the phantom (relative L2 error
0.16) in 2.6 s on one MI355X; at 1 um the low spatial frequencies are barely encoded and the same run stalls at 0.82; in
the far field with a plane probe (one bright bin) it does not converge — properties of the measurement, the same for the
reference.

    python examples/reconstruct_phantom.py [n=128] [n_theta=60] [n_epochs=100] [learning_rate=2e-8] [free_prop_cm=1e-3] [minibatch=10]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_dof_amd import h5io                                   # noqa: E402
from beyond_dof_amd.fullfield import reconstruct_fullfield       # noqa: E402
from beyond_dof_amd.solver import FullfieldSolver                # noqa: E402


def phantom(n, rng):
    """A few soft blobs inside the central cylinder: delta up to ~1.5e-6, beta = delta / 10."""
    z, y, x = np.mgrid[:n, :n, :n].astype(np.float32)
    d = np.zeros((n, n, n), dtype=np.float32)
    for _ in range(10):
        c = rng.uniform(n * 0.3, n * 0.7, size=3)
        r = rng.uniform(n * 0.04, n * 0.12)
        d += 1e-6 * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * r ** 2))
    return d


def run(n=128, n_theta=60, n_epochs=100, lr=2e-8, fp=1e-3, quiet=False, mb=10, propagator='fft'):
    """Simulate, write exchange/data, reconstruct, compare with the phantom: returns the figures main() prints
    (tests/test_gpu_convergence.py asserts them)."""
    import contextlib
    import io
    rng = np.random.default_rng(0)
    d = phantom(n, rng)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp, propagator=propagator)
        s.set_volume(d, 0.1 * d)
        prj = s.forward_angles(np.arange(n_theta))
        del s
        os.makedirs('case')
        h5io.write_dataset('case/data.h5', 'exchange/data', prj.astype(np.complex64))
        t0 = time.time()
        with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()):
            rd, rb = reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=n_epochs, learning_rate=lr,
                                           minibatch_size=mb, energy_ev=5000, psize_cm=1e-7, free_prop_cm=fp, save_path='case',
                                           output_folder='out', shrink_cycle=None, seed=3, alpha_d=1e-9, alpha_b=1e-10, gamma=0,
                                           initial_guess=[np.zeros_like(d), np.zeros_like(d)], propagator=propagator)
        dt = time.time() - t0
        os.chdir(cwd)
    inner = (slice(n // 4, -n // 4),) * 3
    return {'n': n, 'n_theta': n_theta, 'n_epochs': n_epochs, 'seconds': dt,
            'delta_corr': float(np.corrcoef(rd.ravel(), d.ravel())[0, 1]),
            'delta_corr_inner': float(np.corrcoef(rd[inner].ravel(), d[inner].ravel())[0, 1]),
            'delta_rel_l2': float(np.linalg.norm(rd - d) / np.linalg.norm(d)), 'delta_peak': float(rd.max()), 'phantom_peak': float(d.max()),
            'beta_corr': float(np.corrcoef(rb.ravel(), 0.1 * d.ravel())[0, 1])}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_theta = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    n_epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lr = float(sys.argv[4]) if len(sys.argv) > 4 else 2e-8
    fp = sys.argv[5] if len(sys.argv) > 5 else '1e-3'
    fp = 'inf' if fp == 'inf' else float(fp)
    mb = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    propagator = sys.argv[7] if len(sys.argv) > 7 else 'fft'
    r = run(n, n_theta, n_epochs, lr, fp, mb=mb, propagator=propagator)
    print('reconstruct_fullfield (propagator {}) {}^3, {} angles, {} epochs in minibatches of {}: {:.1f} s ({:.1f} ms per Adam step, entry point to files)'.format(
        propagator, n, n_theta, n_epochs, mb, r['seconds'], 1e3 * r['seconds'] / (n_epochs * max(1, n_theta // mb))))
    print('delta: correlation with the phantom {:.4f} (central half {:.4f}); relative L2 error {:.3f}; peak {:.3e} vs {:.3e}'.format(
        r['delta_corr'], r['delta_corr_inner'], r['delta_rel_l2'], r['delta_peak'], r['phantom_peak']))
    print('beta : correlation {:.4f}'.format(r['beta_corr']))


if __name__ == '__main__':
    main()
