"""Development aid: conv propagator gradient diagnostics."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_dof_amd.engine import MultisliceEngine
from oracle import bdof_oracle as orc

def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

rng = np.random.default_rng(3)
B, Y, X, S, ks, fp = 2, 64, 128, 6, 17, None
delta = rng.uniform(0, 2e-5, size=(B, Y, X, S)); beta = 0.1 * delta
pr, pi = np.ones((Y, X)), np.zeros((Y, X))
psize = [1e-7] * 3
eng = MultisliceEngine(Y, X, S, B, with_grad=True)
eng.set_physics(5000., 1e-7, fp); eng.set_conv(5000., psize, ks); eng.set_probe(pr, pi); eng.set_object_batch(delta, beta)
wave = eng.forward(B, conv=True)
ref = orc.multislice_propagate_cnn(delta, beta, pr, pi, 5000., psize, kernel_size=ks, free_prop_cm=fp)
meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
loss = eng.loss_grad(B, meas, conv=True)
gd, gb = eng.grad_batch_to_host(B)
rl, rgd, rgb = orc.cnn_loss_and_grad(delta, beta, pr, pi, 5000., psize, meas, kernel_size=ks, free_prop_cm=fp)
print('all', rel(gd, rgd), rel(gb, rgb))
m = np.ones_like(gd, dtype=bool); m[0, :20, :20, :] = False
print('away from corner', rel(gd[m], rgd[m]), rel(gb[m], rgb[m]))
print('corner', rel(gd[~m], rgd[~m]), gd[0, 0, 0], rgd[0, 0, 0])
# same with noise-free-ish big residual
meas2 = np.abs(ref) * 1.2
loss = eng.loss_grad(B, meas2, conv=True)
gd, gb = eng.grad_batch_to_host(B)
rl, rgd, rgb = orc.cnn_loss_and_grad(delta, beta, pr, pi, 5000., psize, meas2, kernel_size=ks, free_prop_cm=fp)
print('uniform residual: all', rel(gd, rgd), rel(gb, rgb), 'corner', gd[0, 0, 0], rgd[0, 0, 0])
