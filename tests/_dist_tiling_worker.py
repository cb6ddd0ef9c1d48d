"""Worker of tests/test_gpu_dist.py::test_two_rank_tiled_propagation: one rank of a tile-sharded propagation (forward, and
loss + gradient) sharing ONE GPU with the other rank; gloo carries the field / gradient sums (RCCL refuses two ranks on one
device)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem():
    rng = np.random.default_rng(21)
    n, S = 256, 24
    delta = np.zeros((n, n, S))
    c = n // 4
    delta[c:-c, c:-c, :] = rng.uniform(0, 5e-5, size=(n - 2 * c, n - 2 * c, S))
    beta = 0.1 * delta
    yy, xx = np.mgrid[:n, :n]
    probe = np.exp(-((yy - n / 2.) ** 2 + (xx - n / 2.) ** 2) / (2 * (n / 6.) ** 2))
    meas = (1 + 0.05 * rng.normal(size=(n, n))) * probe
    return n, S, delta, beta, probe, meas


def run(comm=None):
    from beyond_dof_amd.tiling import TiledPropagator
    n, S, delta, beta, probe, meas = problem()
    zero = np.zeros_like(probe)
    tp = TiledPropagator((n, n), S, 5000., 1e-7, tile=128, halo=32, slices_per_exchange=8, variant='tf_all', with_grad=True, comm=comm)
    tp.set_object(delta, beta)
    wave = tp.forward(probe, zero)
    loss, gd, gb = tp.loss_and_grad(probe, zero, meas)
    gprobe = np.ascontiguousarray(tp.field.download().T)
    return tp, dict(wave=wave, loss=np.array(loss), gd=gd, gb=gb, gprobe=gprobe)


def main(out_dir):
    from beyond_dof_amd.comm import TorchComm, get_comm
    comm = get_comm()                      # BDOF_COMM_BACKEND=gloo in the environment
    assert isinstance(comm, TorchComm)
    tp, res = run(comm)
    assert tp.n_tiles == tp.n_tiles_field // comm.size
    np.savez(os.path.join(out_dir, 'tiled_rank{}.npz'.format(comm.rank)), **res)
    comm.Barrier()
    comm.close()


if __name__ == '__main__':
    main(sys.argv[1])
