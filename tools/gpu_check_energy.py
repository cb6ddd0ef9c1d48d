"""Energy drift of a pure-phase multislice stack per engine (float32 round-off bias check)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd.engine import MultisliceEngine  # noqa: E402

for n, engines in ((72, ('resident', 'generic')), (64, ('resident', 'streaming', 'generic'))):
    for S in (64, 256):
        rng = np.random.default_rng(5)
        B = 4
        delta = (rng.random((B, n, n, S), dtype=np.float32) * np.float32(2e-6))
        beta = np.zeros_like(delta)
        pr, pi = orc.gaussian_probe((n, n), 6., 6., 0.5)
        e0 = float(np.sum(np.abs((pr + 1j * pi).astype(np.complex64).astype(np.complex128)) ** 2))
        for engine in engines:
            eng = MultisliceEngine(n, n, S, B, with_grad=False, engine=engine)
            eng.set_physics(5000., 1e-7, None)
            eng.set_probe(pr, pi)
            eng.set_object_batch(delta, beta)
            w = eng.forward(B)
            e = np.sum(np.abs(w.astype(np.complex128)) ** 2, axis=(1, 2))
            print('n %d S %d %-9s energy/e0 - 1 = %+.3e' % (n, S, engine, float(np.mean(e / e0 - 1))))
    # the same stack in numpy complex64 arithmetic with the float32-rounded transfer function
    S = 256
    h = np.fft.ifftshift(orc.get_kernel(1.0, 0.248, [1., 1., 1.], [n, n, S])).astype(np.complex64)
    print('n %d  mean |h|^2 - 1 = %+.3e' % (n, float(np.mean(np.abs(h.astype(np.complex128)) ** 2) - 1)))
