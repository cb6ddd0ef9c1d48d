"""Development aid (GPU box): G17 step by step.  The oracle's loop runs on the CPU; at every Adam step the device gets the ORACLE's
current volume and moments, computes its own gradient and its own update, and both are compared with the oracle's — separates
the error of the gradient from the error of the Adam kernel and from the accumulation over steps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd import util  # noqa: E402
from beyond_dof_amd.ptychography import batches_of_epoch  # noqa: E402
from beyond_dof_amd.solver import PtychoSolver  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def main():
    gdir = os.path.join(ROOT, 'tests', 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g17_reconstruct_ptychography_fft_64.npz'))
    obj_size, psz, sigma = tuple(int(v) for v in g['obj_size']), tuple(int(v) for v in g['probe_size']), float(g['probe_sigma'])
    pos, n_theta, mb, lr = g['probe_pos'], 2, 2, 2e-7
    coords = orc.rotation_lookup(list(obj_size), n_theta)
    pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    s = PtychoSolver(obj_size, psz, pos, n_theta, mb, 5000., 1e-7, pr, pi_, coord_ls=coords)
    s.set_measurements(np.abs(g['prj']))
    x = np.array([init_d, init_b])
    xdev = x.copy()                      # the device's own trajectory
    rng = np.random.RandomState(42)
    for ep in range(2):
        m = v = None
        mdev = vdev = None
        for i_batch, (i_theta, ind) in enumerate(batches_of_epoch(n_theta, len(pos), mb, 1, 0, rng)):
            _, gd, gb = orc.ptycho_loss_and_grad(x[0], x[1], coords[i_theta], pos, pos[ind], g['prj'][i_theta, ind], pr, pi_, psz, 5000., 1e-7)
            # device gradient at the oracle's volume
            s.set_volume(x[0], x[1])
            s.loss_and_grad(i_theta, ind)
            dgd, dgb = s.gradient_to_host()
            e_d, e_b = dgd - gd, dgb - gb
            th_d = np.sum(e_d * gb) / np.sum(gb * gb)           # phase-rotation model: e_delta = -theta g_beta, e_beta = theta g_delta
            th_b = np.sum(e_b * gd) / np.sum(gd * gd)
            sc_d = np.sum(e_d * gd) / np.sum(gd * gd)           # scale error
            sc_b = np.sum(e_b * gb) / np.sum(gb * gb)
            print('ep {} step {}: |gd| {:.3e} |gb| {:.3e}  grad err delta {:.2e} beta {:.2e} | e_d~g_b coeff {:.2e}, e_b~g_d coeff {:.2e} | scale d {:.2e} b {:.2e}'
                  ' | after removing both: d {:.2e} b {:.2e}'.format(
                      ep, i_batch, np.sqrt(np.mean(gd ** 2)), np.sqrt(np.mean(gb ** 2)), rel(dgd, gd), rel(dgb, gb), th_d, th_b, sc_d, sc_b,
                      np.linalg.norm(e_d - th_d * gb - sc_d * gd) / np.linalg.norm(gd), np.linalg.norm(e_b - th_b * gd - sc_b * gb) / np.linalg.norm(gb)))
            # oracle step
            xn, m, v = orc.apply_gradient_adam(x, np.array([gd, gb]), i_batch, m, v, step_size=lr)
            xn = np.clip(xn, 0, None)
            # device's Adam from the ORACLE's state with the DEVICE's gradient (moments of the oracle's previous step uploaded)
            s.g.upload(util.volume_to_rows(dgd, dgb))
            if i_batch == 0:
                s.reset_moments()
            else:
                s.m.upload(util.volume_to_rows(mprev[0], mprev[1]))
                s.v.upload(util.volume_to_rows(vprev[0], vprev[1]))
            s.adam_update(i_batch, lr)
            d1, b1 = s.get_volume()
            # and the oracle's Adam with the device's gradient: separates the kernel from the gradient
            xg, _, _ = orc.apply_gradient_adam(x, np.array([dgd, dgb]), i_batch, None if i_batch == 0 else mprev.copy(),
                                               None if i_batch == 0 else vprev.copy(), step_size=lr)
            xg = np.clip(xg, 0, None)
            print('      one step from the same state: device vs oracle delta {:.2e} ({:.2e} lr rms, max {:.3f} lr) beta {:.2e} | oracle Adam on the device gradient: '
                  'delta {:.2e} beta {:.2e} | device Adam vs oracle Adam on the same gradient: delta {:.2e} beta {:.2e}'.format(
                      rel(d1, xn[0]), np.sqrt(np.mean((d1 - xn[0]) ** 2)) / lr, np.abs(d1 - xn[0]).max() / lr, rel(b1, xn[1]),
                      rel(xg[0], xn[0]), rel(xg[1], xn[1]), rel(d1, xg[0]), rel(b1, xg[1])))
            mprev, vprev = m.copy(), v.copy()
            x = xn
    print('final oracle loop vs golden: delta {:.2e} beta {:.2e}'.format(rel(x[0][::2, ::2, ::2], g['delta_sub']), rel(x[1][::2, ::2, ::2], g['beta_sub'])))


if __name__ == '__main__':
    main()
