"""Worker of tests/test_gpu_dist.py: one rank of a two-rank full-field reconstruction step sharing ONE GPU.
The collective runs through gloo (RCCL refuses two ranks on one device); everything else — sharding of the minibatch,
the HIP path, the slab-pipelined rotation adjoint / all-reduce / Adam — is what an N-GPU run executes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem():
    rng = np.random.default_rng(11)
    n, n_theta, mb = 64, 8, 2
    meas = 1 + 0.05 * rng.normal(size=(n_theta, n, n))
    init_d = np.clip(rng.normal(8.7e-7, 1e-7, size=(n, n, n)), 0, None)
    return n, n_theta, mb, meas, init_d


def main(out_dir):
    from beyond_dof_amd.comm import TorchComm, get_comm, minibatch_schedule
    from beyond_dof_amd.solver import FullfieldSolver
    from beyond_dof_amd import util
    from beyond_dof_amd.comm import RcclComm, comm_backend
    comm = get_comm()                      # BDOF_COMM_BACKEND=gloo in the environment (rccl: one device per rank, >= 2 GPUs)
    assert isinstance(comm, RcclComm if comm_backend() == 'rccl' else TorchComm)
    device = comm.local_rank if comm_backend() == 'rccl' else 0
    sharded = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    n, n_theta, mb, meas, init_d = problem()
    coords = util.rotation_lookup([n, n, n], n_theta)
    s = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, coord_ls=coords, comm=comm, device=device)
    s.set_measurements(meas)
    s.set_volume(init_d, 0.1 * init_d)
    s.reset_moments()
    sched = minibatch_schedule(n_theta, comm.size, mb, rng=np.random.RandomState(3))
    losses = []
    for i, chunk in enumerate(sched):
        mine = chunk[comm.rank * mb:(comm.rank + 1) * mb]
        losses.append(s.step(i, mine, 1e-7, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, want_loss=True, n_slabs=4, sharded=sharded))
    g = s.gradient_to_host()
    d, b = s.get_volume()
    # a sharded step followed by loss_and_grad: self.g then holds this rank's own whole gradient, and gradient_to_host must
    # read it back as it is (the shard layout of the step before no longer applies)
    s.step(len(sched), mine, 0.0, want_loss=False, n_slabs=4, sharded=sharded)
    s.loss_and_grad(mine)
    lg = s.gradient_to_host()
    np.savez(os.path.join(out_dir, 'rank{}_{}.npz'.format(comm.rank, int(sharded))), d=d, b=b, gd=g[0], gb=g[1], losses=np.array(losses),
             sched=np.array(sched), lgd=lg[0], lgb=lg[1])
    comm.Barrier()
    comm.close()


if __name__ == '__main__':
    main(sys.argv[1])
