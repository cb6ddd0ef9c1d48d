"""Multi-rank path on CPU: 2 processes, gloo.  Each rank produces the gradient of ITS angles (here with the
oracle, the GPU is not needed to test the exchange), the volume gradient is SUM-all-reduced through
beyond_dof_amd.comm.TorchComm and divided by size exactly as the solver does; the result must equal the
single-rank gradient of the union minibatch (cnn_propagator/fullfield.py:343-351)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import bdof_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    rng = np.random.default_rng(0)
    n, n_theta, mb = 8, 6, 2
    od = rng.uniform(0, 1e-3, size=(n, n, n))
    ob = 0.1 * od
    coords = orc.rotation_lookup([n, n, n], n_theta)
    prj = 1 + 0.05 * rng.normal(size=(n_theta, n, n))
    return n, n_theta, mb, od, ob, coords, prj


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from beyond_dof_amd.comm import TorchComm, minibatch_schedule
    comm = TorchComm('gloo')
    assert comm.size == world and comm.rank == rank
    n, n_theta, mb, od, ob, coords, prj = _problem()
    sched = minibatch_schedule(n_theta, world, mb, rng=np.random.RandomState(5))
    chunk = sched[0]
    mine = chunk[rank * mb:(rank + 1) * mb]
    one, zero = np.ones((n, n)), np.zeros((n, n))
    _, gd, gb = orc.fullfield_loss_and_grad(od, ob, coords, mine, prj[mine], one, zero, 5000., 1e-7, free_prop_cm=1e-4,
                                            with_reg=False)
    g = torch.from_numpy(np.stack([gd, gb]))
    comm.allreduce_sum_device(g)
    g = g.numpy() / comm.size
    # the slab pipeline of FullfieldSolver.step: produce slab c, all-reduce it, consume it — same result, fixed order
    g2 = torch.zeros(g.size, dtype=torch.float64)
    src = torch.from_numpy(np.stack([gd, gb])).reshape(-1)
    bounds = [0, 100, 101, 700, g2.numel()]
    log, seen = [], []

    def produce(c):
        log.append(('p', c))
        g2[bounds[c]:bounds[c + 1]] = src[bounds[c]:bounds[c + 1]]

    def consume(c):
        log.append(('c', c))
        seen.append(g2[bounds[c]:bounds[c + 1]].clone())

    comm.pipelined_allreduce(g2, bounds, produce, consume)
    assert [e for e in log if e[0] == 'p'] == [('p', c) for c in range(4)]
    assert [e for e in log if e[0] == 'c'] == [('c', c) for c in range(4)]
    assert all(log.index(('p', c)) < log.index(('c', c)) for c in range(4))
    assert np.array_equal(torch.cat(seen).numpy() / comm.size, g.reshape(-1))
    loss_sum = comm.allreduce_sum_host(np.array([float(rank + 1)]))
    idx = comm.bcast_host(np.arange(4) if rank == 0 else np.zeros(4, dtype=np.int64), root=0)
    comm.Barrier()
    np.savez(os.path.join(out_dir, 'rank{}.npz'.format(rank)), g=g, chunk=chunk, loss_sum=loss_sum, idx=idx)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_union(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n, n_theta, mb, od, ob, coords, prj = _problem()
    r0 = np.load(str(tmp_path / 'rank0.npz'))
    r1 = np.load(str(tmp_path / 'rank1.npz'))
    assert np.array_equal(r0['chunk'], r1['chunk'])              # every rank derives the same schedule
    assert np.array_equal(r0['g'], r1['g'])                      # and holds the same reduced gradient
    assert r0['loss_sum'][0] == 3.0 and np.array_equal(r1['idx'], np.arange(4))
    chunk = r0['chunk']
    assert len(chunk) == world * mb and len(set(chunk.tolist())) == world * mb
    one, zero = np.ones((n, n)), np.zeros((n, n))
    _, gd, gb = orc.fullfield_loss_and_grad(od, ob, coords, chunk, prj[chunk], one, zero, 5000., 1e-7, free_prop_cm=1e-4,
                                            with_reg=False)
    np.testing.assert_allclose(r0['g'][0], gd, rtol=0, atol=1e-12 * np.abs(gd).max())
    np.testing.assert_allclose(r0['g'][1], gb, rtol=0, atol=1e-12 * np.abs(gb).max())


def test_single_rank_fallback():
    from beyond_dof_amd.comm import PseudoComm, get_comm
    os.environ.pop('WORLD_SIZE', None)
    c = get_comm()
    assert isinstance(c, PseudoComm) and c.size == 1 and c.rank == 0
    c.Barrier()
    x = np.arange(3.0)
    assert c.allreduce_sum_host(x) is x
