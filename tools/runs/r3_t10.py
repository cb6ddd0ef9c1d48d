import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import bdof_oracle as orc
from beyond_dof_amd import engine
Y = X = 72; fp = None; variant = 'numpy_skip_last'
rng = np.random.default_rng(1)
B, S = 3, 5
delta = rng.uniform(0, 2e-5, size=(B, Y, X, S)); beta = 0.1 * delta
pr, pi = 1 + 0.1 * rng.normal(size=(Y, X)), 0.1 * rng.normal(size=(Y, X))
eng = engine.MultisliceEngine(Y, X, S, B, with_grad=True)
eng.set_physics(5000., 1e-7, fp, variant=variant); eng.set_probe(pr, pi); eng.set_object_batch(delta, beta)
wave = eng.forward(B)
ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, variant=variant, return_probe_array=False)
print('wave err', np.linalg.norm(wave - ref) / np.linalg.norm(ref))
meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
loss = eng.loss_grad(B, meas)
print('loss', loss, 'oracle', np.mean((np.abs(ref) - meas) ** 2))
ml = eng.meas_layout(meas)          # [b][x][y]
rt = ref.transpose(0, 2, 1)
for g in range(3):
    for t in (0, 1, 63, 64, 65, 100, 300, 511, 512, 513, 703):
        x, y = divmod(t, 72)
        print('expect wg', g, 'tid', t, 'meas', ml[g, x, y], '|d|', abs(rt[g, x, y]), 'o', g * 5184 + t)
