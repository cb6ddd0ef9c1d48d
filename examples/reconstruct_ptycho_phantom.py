"""A complete small ptychographic reconstruction, the way cnn_propagator/reconstruct_ptycho.py runs one: far-field diffraction
patterns of a gaussian probe scanned over a phantom at a set of angles (simulated with the product's forward model), written
as exchange/data (n_theta, n_pos, py, px), reconstructed with reconstruct_ptychography and compared with the phantom.  On one MI355X: 128^3, 121 positions x 30 angles,
40 epochs at learning rate 5e-8 in 4.7 s — delta correlation 0.996, relative L2 error 0.09, beta correlation 0.88.

    python examples/reconstruct_ptycho_phantom.py [n=128] [n_theta=30] [n_epochs=20] [learning_rate=1e-7]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_dof_amd import h5io, util                                # noqa: E402
from beyond_dof_amd.ptychography import reconstruct_ptychography    # noqa: E402
from beyond_dof_amd.solver import PtychoSolver                      # noqa: E402


def phantom(n, rng):
    z, y, x = np.mgrid[:n, :n, :n].astype(np.float32)
    d = np.zeros((n, n, n), dtype=np.float32)
    for _ in range(10):
        c = rng.uniform(n * 0.3, n * 0.7, size=3)
        r = rng.uniform(n * 0.04, n * 0.12)
        d += 1e-5 * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * r ** 2))
    return d


def run(n=128, n_theta=30, n_epochs=20, lr=1e-7, quiet=False):
    """Simulate, write exchange/data, reconstruct, compare with the phantom (tests/test_gpu_convergence.py asserts the figures)."""
    import contextlib
    import io
    psz, step = (72, 72), 12                                          # the reference drivers' probe (reconstruct_ptycho.py:106,129-131)
    rng = np.random.default_rng(0)
    d = phantom(n, rng)
    pos = [(y, x) for y in range(0, n, step) for x in range(0, n, step)]
    kw = dict(probe_mag_sigma=6., probe_phase_sigma=6., probe_phase_max=0.5)
    prr, pii = util.gaussian_probe(psz, kw['probe_mag_sigma'], kw['probe_phase_sigma'], kw['probe_phase_max'])
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        s = PtychoSolver((n, n, n), psz, pos, n_theta, len(pos), 5000., 1e-7, prr, pii)
        s.set_volume(d, 0.1 * d)
        data = np.stack([s.forward(t, np.arange(len(pos))) for t in range(n_theta)]).astype(np.complex64)
        del s
        os.makedirs('case')
        h5io.write_dataset('case/data.h5', 'exchange/data', data)
        t0 = time.time()
        with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()):
            rd, rb = reconstruct_ptychography('data.h5', pos, psz, (n, n, n), theta_st=0, theta_end=2 * np.pi, n_epochs=n_epochs,
                                              learning_rate=lr, minibatch_size=len(pos), energy_ev=5000, psize_cm=1e-7, save_path='case',
                                              output_folder='out', initial_guess=[np.zeros_like(d), np.zeros_like(d)],
                                              probe_type='gaussian', seed=3, n_dp_batch=len(pos), alpha_d=0, alpha_b=0, **kw)
        dt = time.time() - t0
        os.chdir(cwd)
    return {'n': n, 'n_pos': len(pos), 'n_theta': n_theta, 'n_epochs': n_epochs, 'seconds': dt,
            'delta_corr': float(np.corrcoef(rd.ravel(), d.ravel())[0, 1]), 'delta_rel_l2': float(np.linalg.norm(rd - d) / np.linalg.norm(d)),
            'delta_peak': float(rd.max()), 'phantom_peak': float(d.max()), 'beta_corr': float(np.corrcoef(rb.ravel(), 0.1 * d.ravel())[0, 1])}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_theta = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n_epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-7
    r = run(n, n_theta, n_epochs, lr)
    print('reconstruct_ptychography {}^3, {} positions x {} angles, {} epochs: {:.1f} s'.format(n, r['n_pos'], n_theta, n_epochs, r['seconds']))
    print('delta: correlation with the phantom {:.4f}; relative L2 error {:.3f}; peak {:.3e} vs {:.3e}'.format(
        r['delta_corr'], r['delta_rel_l2'], r['delta_peak'], r['phantom_peak']))
    print('beta : correlation {:.4f}'.format(r['beta_corr']))


if __name__ == '__main__':
    main()
