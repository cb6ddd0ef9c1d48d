"""Forward-intensity error of the ptychography windows vs the float64 oracle, per engine and probe size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import bdof_oracle as orc  # noqa: E402
import test_gpu_ptycho as t  # noqa: E402
from beyond_dof_amd.solver import PtychoSolver  # noqa: E402

for psz, env in [((64, 64), {}), ((64, 64), {'BDOF_FORCE_RESIDENT': '1'}), ((72, 72), {}), ((72, 72), {'BDOF_NO_RESIDENT': '1'}),
                 ((96, 96), {}), ((96, 96), {'BDOF_NO_RESIDENT': '1'}), ((60, 60), {})]:
    for k in ('BDOF_FORCE_RESIDENT', 'BDOF_NO_RESIDENT'):
        os.environ.pop(k, None)
    os.environ.update(env)
    rng, n, n_theta, psz, pos, od, ob, coords, prr, pii = t._setup(psz)
    s = PtychoSolver((n, n, n), psz, pos, n_theta, 6, 5000., 1e-7, prr, pii, coord_ls=coords)
    s.set_volume(od, ob)
    sel = np.array([0, 3, 5, 6, 10, 11])
    pad, half = orc.ptycho_pad_amounts(pos, psz, (n, n, n))
    rot = orc.apply_rotation(np.stack([od, ob], axis=3), coords[2])
    obj_pad = np.pad(rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
    subs = np.stack([obj_pad[p[0] + pad[0, 0] - half[0]:p[0] + pad[0, 0] - half[0] + psz[0],
                             p[1] + pad[1, 0] - half[1]:p[1] + pad[1, 0] - half[1] + psz[1]] for p in pos[sel]])
    ref, _ = orc.multislice_propagate_batch_numpy(subs[..., 0], subs[..., 1], prr, pii, 5000., 1e-7, 'inf', subs[..., 0].shape,
                                                  return_probe_array=False)
    w = s.forward(2, sel)
    print(psz, env, 'intensity rel err %.3e  wave rel err %.3e' % (t.rel(np.abs(w) ** 2, np.abs(ref) ** 2), t.rel(w, ref)))
