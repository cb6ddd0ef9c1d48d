"""Development aid (GPU box): full-field gradient + Adam step vs oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beyond_dof_amd.solver import FullfieldSolver
from oracle import bdof_oracle as orc

def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

def case(n, n_theta, mb, fp, seed=0):
    rng = np.random.default_rng(seed)
    od = rng.uniform(0, 2e-6, size=(n, n, n)); ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    idx = np.sort(rng.choice(n_theta, mb, replace=False))
    obj_stack = np.stack([od, ob], axis=3)
    rot = np.stack([orc.apply_rotation(obj_stack, coords[j]) for j in idx])
    ref_wave, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], np.ones((n, n)), np.zeros((n, n)), 5000., 1e-7, fp, rot[..., 0].shape, return_probe_array=False)
    prj = np.zeros((n_theta, n, n))
    prj[idx] = np.abs(ref_wave) * (1 + 0.02 * rng.normal(size=ref_wave.shape))
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=fp)
    s.set_volume(od, ob)
    s.set_measurements(prj)
    w = s.forward_angles(idx)
    print('n{} fwd via fused rotation: wave {:.2e}'.format(n, rel(w, ref_wave)))
    loss = s.loss_and_grad(idx)
    gd, gb = s.gradient_to_host()
    rl, rgd, rgb = orc.fullfield_loss_and_grad(od, ob, coords, idx, prj[idx], np.ones((n, n)), np.zeros((n, n)), 5000., 1e-7, free_prop_cm=fp, with_reg=False)
    print('   loss {:.2e} gd {:.2e} gb {:.2e}'.format(abs(loss - rl) / rl, rel(gd, rgd), rel(gb, rgb)))
    # Adam steps with regulariser
    mask = (rng.uniform(size=(n, n, n)) > 0.1).astype(np.float32)
    s.set_mask(mask)
    kw = dict(alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
    x = np.array([od, ob]); m = v = None
    for it in range(3):
        s.step(it, idx, 1e-7, **kw)
        _, g1, g2 = orc.fullfield_loss_and_grad(x[0], x[1], coords, idx, prj[idx], np.ones((n, n)), np.zeros((n, n)), 5000., 1e-7, free_prop_cm=fp, with_reg=True, **kw)
        x, m, v = orc.apply_gradient_adam(x, np.array([g1, g2]), it, m, v, step_size=1e-7)
        x = np.clip(x * mask, 0, None)
        d, b = s.get_volume()
        print('   adam it{}: delta {:.2e} beta {:.2e}  (max abs diff delta {:.2e} of max {:.2e})'.format(it, rel(d, x[0]), rel(b, x[1]), np.abs(d - x[0]).max(), x[0].max()))

case(64, 8, 2, 1e-4)
case(64, 5, 3, None)
case(128, 6, 2, 'inf')
