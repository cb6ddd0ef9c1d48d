"""Intensity error against the float64 oracle at cfg5 depth (72 x 72, 256 slices, far field), per engine, with and without
the free-space energy calibration, for a random and for a smooth object."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd.engine import MultisliceEngine  # noqa: E402
from scipy.ndimage import gaussian_filter  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


n, S, B = 72, 256, 3
rng = np.random.default_rng(55)
objs = {'random voxels': rng.uniform(0, 2e-6, size=(B, n, n, S))}
objs['smooth (sigma 3)'] = np.stack([gaussian_filter(o, 3) for o in objs['random voxels']])
pr, pi = orc.gaussian_probe((n, n), 6., 6., 0.5)
for name, delta in objs.items():
    beta = 0.1 * delta
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, 'inf', delta.shape, return_probe_array=False)
    for engine in ('resident', 'generic'):
        for cal in (True,):        # (the free-space energy calibration of round 1 is gone: every engine carries the probe field)
            eng = MultisliceEngine(n, n, S, B, with_grad=False, engine=engine)
            eng.set_physics(5000., 1e-7, 'inf')
            eng.set_probe(pr, pi)
            eng.set_object_batch(delta, beta)
            w = eng.forward(B)
            e = np.sum(np.abs(w.astype(np.complex128)) ** 2) / np.sum(np.abs(ref) ** 2) - 1
            print('%-18s %-9s calibrated %-5s intensity rel err %.3e  wave rel err %.3e  energy ratio - 1 %+.3e' %
                  (name, engine, cal, rel(np.abs(w) ** 2, np.abs(ref) ** 2), rel(w, ref), e))
