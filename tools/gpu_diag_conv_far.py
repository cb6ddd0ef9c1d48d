"""Where the real-space propagator's far-field ptychography gradient loses accuracy (golden vector G21's configuration: 64^3 object,
64 x 64 gaussian probe of sigma 40, two positions, 17 taps): device gradient against the float64 oracle — as a whole, away from
the wake of the corner pixel (through which the renormalisation of propagation.py:109-110 sends its scalar), slice by slice, and
for the three detectors.  usage: python tools/gpu_diag_conv_far.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from oracle import bdof_oracle as orc  # noqa: E402
from beyond_dof_amd.engine import MultisliceEngine  # noqa: E402
import g13_inputs  # noqa: E402

rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
f = np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_reconstruct_ptychography_64.npz'))
obj_size, psz, sigma = tuple(int(v) for v in f['obj_size']), tuple(int(v) for v in f['probe_size']), float(f['probe_sigma'])
d, b = g13_inputs.initial_guess(obj_size)
pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
n = obj_size[0]
# the two windows of the first minibatch cover the whole object here (probe 64 = object 64): take the object itself as both
# wavefields' sub-volumes, with the fixture's amplitudes — the forward model and loss of cnn_propagator/ptychography.py:74-79
delta = np.stack([d, d[::-1]]).astype(np.float64)
beta = np.stack([b, b[::-1]]).astype(np.float64)
S = delta.shape[-1]
for fp in ('inf', 1e-4, None):
    ref = orc.multislice_propagate_cnn(delta, beta, pr, pi_, 5000., [1e-7] * 3, kernel_size=17, free_prop_cm=fp)
    rng = np.random.default_rng(1)
    meas = np.abs(ref) * (1 + 0.02 * rng.normal(size=ref.shape))
    rl, rgd, rgb = orc.cnn_loss_and_grad(delta, beta, pr, pi_, 5000., [1e-7] * 3, meas, kernel_size=17, free_prop_cm=fp)
    eng = MultisliceEngine(n, n, S, 2, with_grad=True)
    eng.set_physics(5000., 1e-7, fp)
    eng.set_conv(5000., [1e-7] * 3, 17)
    eng.set_probe(pr, pi_)
    eng.set_object_batch(delta, beta)
    loss = eng.loss_grad(2, meas, conv=True)
    gd, gb = eng.grad_batch_to_host(2)
    print('detector %s: loss rel err %.2e; gradient delta %.2e beta %.2e' % (fp, abs(loss - rl) / rl, rel(gd, rgd), rel(gb, rgb)))
    # scale and phase of the best complex multiple: a systematic factor shows here
    for name, g, r in (('delta', gd, rgd), ('beta', gb, rgb)):
        a = float(np.sum(g * r) / np.sum(r * r))
        print('   %s: best scale of the device gradient on the oracle\'s %.8f (1 - %.2e); residual after scaling %.2e' % (name, a, 1 - a, rel(g / a, r)))
    err_z = [rel(gd[..., z], rgd[..., z]) for z in range(S)]
    print('   per slice (delta): z=0 %.2e, z=%d %.2e, z=%d %.2e; max %.2e at z=%d' % (err_z[0], S // 2, err_z[S // 2], S - 1, err_z[-1], max(err_z), int(np.argmax(err_z))))
    # error map summed over z and b: where in the window
    e2 = np.sum((gd - rgd) ** 2, axis=(0, 3))
    tot = e2.sum()
    print('   share of the squared error in the 16 x 16 corner block %.3f, in the outer 8-pixel frame %.3f' % (e2[:16, :16].sum() / tot,
          1 - e2[8:-8, 8:-8].sum() / tot))
    del eng
