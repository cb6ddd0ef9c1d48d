"""Development aid (GPU box): structure of the gradient error, plane probe + far field."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beyond_dof_amd.engine import MultisliceEngine
from oracle import bdof_oracle as orc

def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

rng = np.random.default_rng(3)
for (Y, X, S, fp) in ((128, 128, 128, 'inf'), (128, 128, 128, 1e-4)):
    B = 2
    delta = rng.uniform(0, 2e-6, size=(B, Y, X, S)); beta = 0.1 * delta
    pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, return_probe_array=False)
    meas = (np.abs(ref) * (1 + 0.02 * rng.normal(size=ref.shape))).astype(np.float32)
    eng = MultisliceEngine(Y, X, S, B, with_grad=True, engine='streaming')
    eng.set_physics(5000., 1e-7, fp); eng.set_probe(pr, pi); eng.set_object_batch(delta, beta)
    loss = eng.loss_grad(B, meas)
    gd, gb = eng.grad_batch_to_host(B)
    rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas.astype(np.float64), fp)
    err = gd - rgd
    print(fp, 'loss rel', abs(loss - rl) / rl, 'gd', rel(gd, rgd), 'gb', rel(gb, rgb))
    print('  |rgd| rms', np.sqrt(np.mean(rgd ** 2)), ' mean rgd', rgd.mean(), ' |rgb| rms', np.sqrt(np.mean(rgb ** 2)))
    print('  err rms', np.sqrt(np.mean(err ** 2)), ' err mean', err.mean())
    m_bs = err.mean(axis=(1, 2))          # (B, S): constant offset per wavefield and slice
    print('  rms of per-(b,slice) mean of err', np.sqrt(np.mean(m_bs ** 2)), ' -> after removing it', np.sqrt(np.mean((err - m_bs[:, None, None, :]) ** 2)))
    print('  per-slice mean err, b=0, first/last 4:', m_bs[0, :4], m_bs[0, -4:])
    print('  per-slice mean of rgd, b=0:', rgd.mean(axis=(1, 2))[0, :3], rgd.mean(axis=(1, 2))[0, -3:])
    e_s = np.sqrt(np.mean(err ** 2, axis=(0, 1, 2)))
    print('  err rms per slice:', e_s[::16], e_s[-1])
    r_s = np.sqrt(np.mean(rgd ** 2, axis=(0, 1, 2)))
    print('  rgd rms per slice:', r_s[::16], r_s[-1])
    # smooth or white?  correlation of the error between neighbouring pixels / slices
    e0 = err[0]
    print('  corr(x+1)', np.mean(e0[:, 1:, :] * e0[:, :-1, :]) / np.mean(e0 ** 2), ' corr(slice+1)', np.mean(e0[:, :, 1:] * e0[:, :, :-1]) / np.mean(e0 ** 2))
    al = np.array([np.sum(gd[..., z] * rgd[..., z]) / np.sum(rgd[..., z] ** 2) - 1 for z in range(S)])
    res = np.array([np.linalg.norm(gd[..., z] - (1 + al[z]) * rgd[..., z]) / np.linalg.norm(rgd[..., z]) for z in range(S)])
    print('  scale error alpha_z:', al[::16], al[-1])
    print('  residual after removing the scale:', res[::16], res[-1])
    alb = np.array([np.sum(gb[..., z] * rgb[..., z]) / np.sum(rgb[..., z] ** 2) - 1 for z in range(S)])
    print('  beta scale error:', alb[::16], alb[-1])
