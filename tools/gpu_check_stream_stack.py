"""Carrier field on the streaming engine: localised probe on power-of-two fields, all detector modes, history tape."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd.engine import MultisliceEngine  # noqa: E402

rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
rng = np.random.default_rng(4)
for (Y, X, S) in ((64, 128, 40), (256, 256, 96)):
    B = 2
    delta = rng.uniform(0, 2e-6, size=(B, Y, X, S))
    beta = 0.1 * delta
    pr, pi = orc.gaussian_probe((Y, X), Y / 10., X / 10., 0.5)
    for fp, var in ((None, 'numpy_skip_last'), (None, 'tf_all'), (1e-4, 'numpy_skip_last'), (1e-4, 'tf_all'), ('inf', 'numpy_skip_last'),
                    ('inf', 'tf_all')):
        ref, pa = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, variant=var)
        meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
        rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp, var)
        for stack in (True, False):
            os.environ.pop('BDOF_NO_PROBE_STACK', None)
            if not stack:
                os.environ['BDOF_NO_PROBE_STACK'] = '1'
            eng = MultisliceEngine(Y, X, S, B, with_grad=True, engine='streaming')
            eng.set_physics(5000., 1e-7, fp, variant=var)
            eng.set_probe(pr, pi)
            eng.set_object_batch(delta, beta)
            w = eng.forward(B)
            loss = eng.loss_grad(B, meas)
            gd, gb = eng.grad_batch_to_host(B)
            hist = ''
            if var == 'numpy_skip_last':
                eng.forward(B, keep_tape=True)
                hist = ' history %.2e' % rel(eng.probe_array(B), pa)
            print('%dx%dx%d %-5s %-15s stack %-5s wave %.2e loss %.2e gd %.2e gb %.2e%s' %
                  (Y, X, S, fp, var, eng.probe_stack, rel(w, ref), abs(loss - rl) / rl, rel(gd, rgd), rel(gb, rgb), hist))
os.environ.pop('BDOF_NO_PROBE_STACK', None)
