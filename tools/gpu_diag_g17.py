"""Development aid (GPU box): where does the reconstructed-delta error of far-field ptychography (golden vector G17) come from?
For each device engine: gradient of the first minibatch against the oracle (window frame and volume), then the eight Adam
steps of G17 against the reference loop's volume.  BDOF_LIB selects a differently built library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd.ptychography import batches_of_epoch  # noqa: E402
from beyond_dof_amd.solver import PtychoSolver  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def main():
    gdir = os.path.join(ROOT, 'tests', 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    which = sys.argv[1] if len(sys.argv) > 1 else 'g17'
    g = np.load(os.path.join(gdir, 'g17_reconstruct_ptychography_fft_64.npz' if which == 'g17' else 'g14_reconstruct_ptychography_64.npz'))
    conv = which != 'g17'
    obj_size, psz, sigma = tuple(int(v) for v in g['obj_size']), tuple(int(v) for v in g['probe_size']), float(g['probe_sigma'])
    pos, n_theta, mb, lr = g['probe_pos'], 2, 2, 2e-7
    coords = orc.rotation_lookup(list(obj_size), n_theta)
    pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    engines = (('conv', {}),) if conv else (('resident', {}), ('streaming', {'BDOF_NO_RESIDENT_PIN': '1'}),
                                            ('generic', {'BDOF_NO_RESIDENT_PIN': '1', 'BDOF_FORCE_GENERIC': '1'}),
                                            ('adjoint64', {'BDOF_ADJOINT64': '1'}))
    print('lib', os.environ.get('BDOF_LIB', 'default'), 'BDOF_NO_F64_DET' if os.environ.get('BDOF_NO_F64_DET') else '')
    ind0 = np.array([0, 1])
    _, rgd, rgb = orc.ptycho_loss_and_grad(init_d, init_b, coords[0], pos, pos[ind0], g['prj'][0, ind0], pr, pi_, psz, 5000., 1e-7,
                                           propagator='conv' if conv else 'fft')
    for name, env in engines:
        for k in ('BDOF_NO_RESIDENT_PIN', 'BDOF_FORCE_GENERIC', 'BDOF_ADJOINT64'):
            os.environ.pop(k, None)
        os.environ.update(env)
        s = PtychoSolver(obj_size, psz, pos, n_theta, mb, 5000., 1e-7, pr, pi_, coord_ls=coords, propagator='conv' if conv else 'fft')
        s.set_volume(init_d, init_b)
        s.loss_and_grad(0, ind0, np.abs(g['prj'][0, ind0]))
        gd, gb = s.gradient_to_host()
        e = gd - rgd
        nz = np.abs(rgd) > 0
        # multiplicative part: per-voxel relative error where the gradient is not tiny
        big = np.abs(rgd) > 0.1 * np.sqrt(np.mean(rgd[nz] ** 2))
        print('{:10s} volume gradient: delta {:.2e} beta {:.2e} | delta rel err at big voxels rms {:.2e} | max|e|/rms(g) {:.2e}'.format(
            name, rel(gd, rgd), rel(gb, rgb), np.sqrt(np.mean((e[big] / rgd[big]) ** 2)), np.abs(e).max() / np.sqrt(np.mean(rgd[nz] ** 2))))
        # the eight Adam steps of G17
        s.set_volume(init_d, init_b)
        s.set_measurements(np.abs(g['prj']))
        rng = np.random.RandomState(42)
        for _ in range(2):
            s.reset_moments()
            for i_batch, (i_theta, ind) in enumerate(batches_of_epoch(n_theta, len(pos), mb, 1, 0, rng)):
                s.step(i_batch, i_theta, ind, None, lr)
        d, b = s.get_volume()
        d, b = d[::2, ::2, ::2], b[::2, ::2, ::2]
        print('{:10s} after 8 Adam steps: delta {:.2e} beta {:.2e} max dev {:.3f} / {:.3f} lr'.format(
            name, rel(d, g['delta_sub']), rel(b, g['beta_sub']), np.abs(d - g['delta_sub']).max() / lr, np.abs(b - g['beta_sub']).max() / lr))
        del s


if __name__ == '__main__':
    main()
