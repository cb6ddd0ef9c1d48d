"""Multi-rank path on CPU: 2 processes.  (a) gloo:  Each rank produces the gradient of ITS angles (here with the
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
    comm.allreduce_sum_device(None, g)
    g = g.numpy() / comm.size
    # the collectives of the solver's sharded tail on a slab [lo, lo + size*per): reduce-scatter leaves this rank's part
    # summed, the all-gather hands every rank's part to all — together they equal the all-reduce
    src = torch.from_numpy(np.stack([gd, gb])).reshape(-1)
    g2 = src.clone()
    lo, per = 96, 200
    assert comm.start_reduce_scatter(None, g2, lo, per) is None
    mine_lo = lo + rank * per
    assert np.array_equal(g2[mine_lo:mine_lo + per].numpy() / comm.size, g.reshape(-1)[mine_lo:mine_lo + per])
    g3 = torch.full_like(src, -7.0)
    g3[mine_lo:mine_lo + per] = g2[mine_lo:mine_lo + per]
    comm.start_allgather(None, g3, lo, per)
    comm.wait(None, None)
    assert np.array_equal(g3[lo:lo + world * per].numpy() / comm.size, g.reshape(-1)[lo:lo + world * per])
    assert float(g3[0]) == -7.0 and float(g3[lo + world * per]) == -7.0          # nothing outside the slab is touched
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


# ---- control plane of the native (RCCL) path: unix-socket rendezvous + host collectives, no GPU needed ----------------
def _socket_worker(rank, world, path, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), BDOF_RDZV=path)
    from beyond_dof_amd.comm import RcclComm, get_comm
    comm = get_comm()
    assert isinstance(comm, RcclComm) and comm.size == world and comm.rank == rank and comm.backend == 'rccl'
    tot = comm.allreduce_sum_host(np.array([rank + 1.0, 10.0 * (rank + 1)]))
    mx = comm.allreduce_max_host(np.array([float(rank), -float(rank)]))
    big = comm.bcast_host(np.arange(300000, dtype=np.float64) if rank == 0 else None, root=0)      # > one socket buffer
    uid = comm.group.bcast(b'x' * 128 if rank == 0 else None)
    where = comm.group.allgather(('host', rank))
    comm.Barrier()
    np.savez(os.path.join(out_dir, 'sock{}.npz'.format(rank)), tot=tot, mx=mx, big_sum=big.sum(), uid_len=len(uid),
             where=np.array([w[1] for w in where]))
    comm.close()


@pytest.mark.timeout(120)
def test_socket_rendezvous_and_host_collectives(tmp_path):
    world = 3
    path = str(tmp_path / 'rdzv.sock')
    open(path, 'w').close()                       # a stale file of a dead job must not be in the way
    mp.spawn(_socket_worker, args=(world, path, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = np.load(str(tmp_path / 'sock{}.npz'.format(r)))
        assert np.array_equal(d['tot'], [6.0, 60.0]) and np.array_equal(d['mx'], [2.0, 0.0])
        assert d['big_sum'] == 300000 * 299999 / 2 and d['uid_len'] == 128 and np.array_equal(d['where'], [0, 1, 2])
    assert not os.path.exists(path)               # rank 0 removes the socket file once everyone is connected


def test_solver_tail_plan_without_a_gpu():
    """Which form of the exchange a volume of dim_x planes takes (solver._VolumeSolver.tail_plan): pure host logic."""
    from beyond_dof_amd.solver import _VolumeSolver

    class C(object):
        size, rank, sharded = 8, 0, True

    s = _VolumeSolver.__new__(_VolumeSolver)
    s.comm, s._plan = C(), None
    for dim_x, want in ((512, (8, True)), (64, (8, True)), (72, (3, True)), (48, (6, True)), (8, (1, True)), (20, (8, False)), (7, (7, False))):
        s.dim_x = dim_x
        got = s.tail_plan()
        if want[1]:
            assert got == want, (dim_x, got)
            assert dim_x % (got[0] * 8) == 0
        else:
            assert got[1] is False
    s.dim_x = 512
    assert s.tail_plan(n_slabs=3, sharded=False) == (3, False)
    assert sum(nx for _, nx in s.slab_bounds(3)) == 512


def test_control_plane_messages_are_parsed_not_executed(tmp_path):
    """The rendezvous socket carries None / bytes / numpy arrays / JSON values in a typed encoding (no pickle), lives in a
    directory only this user can enter, and both ends check the peer's uid (round-2 advice)."""
    from beyond_dof_amd import comm
    for obj in (None, b'\x00\x01' * 64, [3, 'host', 1.5], {'a': [1, 2]}, 7):
        assert comm._decode(comm._encode(obj)) == obj
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    b = comm._decode(comm._encode(a))
    assert b.dtype == a.dtype and np.array_equal(a, b)
    items = comm._decode(comm._encode_list([comm._encode(None), comm._encode(a), comm._encode([1, 2])]))
    assert items[0] is None and np.array_equal(items[1], a) and items[2] == [1, 2]
    with pytest.raises(TypeError):
        comm._encode(np.array([object()]))
    with pytest.raises(ValueError):
        comm._decode(b'\x80\x04junk')                 # a pickle stream is just a malformed message
    src = open(comm.__file__).read()
    assert 'import pickle' not in src and 'pickle.loads' not in src
    d = comm.rendezvous_dir()
    st = os.stat(d)
    assert st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0
    os.environ.pop('BDOF_RDZV', None)
    assert os.path.dirname(comm.rendezvous_path()) == d


@pytest.mark.timeout(300)
@pytest.mark.parametrize('backend', ['rccl', 'gloo'])
def test_bench_self_launch_eight_ranks_rehearsal(backend):
    """`python bench.py --gpus 8` end to end without a GPU (--rehearse-cpu): the parent starts eight ranks, they meet (native
    control plane: socket star in a private directory; gloo: TCP store), cover the 200 angles of cfg3 between them, exchange
    through the backend, take the max-over-ranks time, and rank 0 prints exactly one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'BDOF_RDZV')}
    env['BDOF_COMM_BACKEND'] = backend
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--rehearse-cpu', '--steps', '3', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['rehearsal'] and d['config']['angles_covered'] == 200 and d['config']['exchange_ok']
