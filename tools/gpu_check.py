"""Development aid (GPU box): print HIP-vs-oracle errors for a sweep of cases."""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beyond_dof_amd.engine import MultisliceEngine
from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def case(B, Y, X, S, fp, variant, seed=0, dmax=2e-5, grad=True, plane=False, noise=0.05):
    rng = np.random.default_rng(seed)
    delta = rng.uniform(0, dmax, size=(B, Y, X, S))
    beta = 0.1 * delta
    pr = 1 + 0.1 * rng.normal(size=(Y, X))
    pi = 0.1 * rng.normal(size=(Y, X))
    if plane:
        pr, pi = np.ones((Y, X)), np.zeros((Y, X))
    eng = MultisliceEngine(Y, X, S, B, with_grad=True)
    eng.set_physics(5000., 1e-7, fp, variant=variant)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    t0 = time.time()
    wave = eng.forward(B)
    t1 = time.time()
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, variant=variant,
                                                  return_probe_array=False)
    ew = rel(wave, ref)
    ei = rel(np.abs(wave) ** 2, np.abs(ref) ** 2)
    msg = ('PLANE ' if plane else '') + 'B{} {}x{}x{} det={} {}: wave {:.2e} intensity {:.2e}'.format(B, Y, X, S, fp, variant, ew, ei)
    if grad:
        meas = np.abs(ref) * (1 + noise * rng.normal(size=ref.shape))
        loss = eng.loss_grad(B, meas)
        gd, gb = eng.grad_batch_to_host(B)
        rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp, variant)
        msg += ' | loss {:.2e} gd {:.2e} gb {:.2e}'.format(abs(loss - rl) / abs(rl), rel(gd, rgd), rel(gb, rgb))
    print(msg, flush=True)


if __name__ == '__main__':
    for fp in [None, 1e-4, 'inf']:
        for variant in ['numpy_skip_last', 'tf_all']:
            case(2, 128, 128, 64, fp, variant, dmax=2e-6, plane=True, noise=0.0 if fp != 'inf' else 0.05)
    case(2, 512, 512, 128, 1e-4, 'numpy_skip_last', dmax=2e-6, plane=True, noise=0.0)
    case(2, 512, 512, 128, 1e-4, 'numpy_skip_last', dmax=2e-6, plane=False, noise=0.0)
    for (Y, X) in [(64, 64), (128, 128), (64, 256), (256, 128)]:
        for fp in [None, 1e-4, 'inf']:
            for variant in ['numpy_skip_last', 'tf_all']:
                case(2, Y, X, 5, fp, variant)
    case(3, 256, 256, 16, 1e-4, 'numpy_skip_last')
    case(2, 512, 512, 8, 1e-4, 'numpy_skip_last')
    case(1, 1024, 1024, 3, 'inf', 'numpy_skip_last')
    case(1, 64, 64, 1, None, 'numpy_skip_last')
    case(1, 64, 64, 1, 1e-4, 'tf_all')
