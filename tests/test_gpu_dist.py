"""Two ranks, one GPU: the N-rank Adam step (angle-sharded minibatch, slab-pipelined rotation adjoint -> exchange ->
Adam, in the all-reduce and in the reduce-scatter / sharded-Adam / all-gather form) against the single-rank step on the
union minibatch.  gloo carries the collective because RCCL does not accept two ranks on one device; the RCCL flavour of
the same pipeline (the library's own communicator) runs on one rank in test_gpu_fullfield.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch_two(worker, *args, **kw):
    port = _free_port()
    procs = []
    backend = kw.get('backend', 'gloo')
    stub = kw.get('stub')
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2',
                   LOCAL_RANK='0' if stub else str(rank), BDOF_COMM_BACKEND=backend)
        if backend == 'rccl':
            env['BDOF_RDZV'] = os.path.join(str(args[0]), 'rdzv_{}.sock'.format(port))
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if stub:
            env['BDOF_RCCL_LIB'] = stub
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', worker)] + [str(a) for a in args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    return procs


def _build_rccl_stub(tmp_path):
    """tests/rccl_stub/rccl_stub.cpp -> a shared library with librccl's eight entry points the product binds."""
    out = os.path.join(str(tmp_path), 'librccl_stub.so')
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-O2', '-std=c++17', '-shared', '-fPIC', '-o', out,
                           os.path.join(ROOT, 'tests', 'rccl_stub', 'rccl_stub.cpp'), '-I/opt/rocm/include'])
    return out


def _run_two_ranks(tmp_path, sharded, backend='gloo', stub=None):
    procs = _launch_two('_dist_gpu_worker.py', tmp_path, int(sharded), backend=backend, stub=stub)
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-3000:] for o in outs)
    return [np.load(str(tmp_path / 'rank{}_{}.npz'.format(r, int(sharded)))) for r in range(2)]


def test_two_ranks_on_one_gpu_match_the_union_minibatch(tmp_path):
    import __graft_entry__ as entry
    entry.build()
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _dist_gpu_worker as w
    r0, r1 = _run_two_ranks(tmp_path, sharded=False)
    # every rank ends with the same volume, bit for bit (same reduced gradient, same Adam)
    assert np.array_equal(r0['d'], r1['d']) and np.array_equal(r0['b'], r1['b'])
    assert np.array_equal(r0['gd'], r1['gd'])
    # reduce-scatter -> Adam on this rank's half of every slab -> all-gather: the same volume as the all-reduce form
    s0, s1 = _run_two_ranks(tmp_path, sharded=True)
    assert np.array_equal(s0['d'], s1['d']) and np.array_equal(s0['b'], s1['b'])
    assert np.array_equal(s0['d'], r0['d']) and np.array_equal(s0['b'], r0['b'])
    # gradient_to_host() after a sharded step gathers the ranks' parts: the all-reduce form's gradient, on both ranks
    assert np.array_equal(s0['gd'], r0['gd']) and np.array_equal(s1['gd'], r0['gd']) and np.array_equal(s0['gb'], r0['gb'])
    # step(sharded) -> loss_and_grad -> gradient_to_host: the rank's own unreduced gradient, not a gather of stale shards over it
    for k in ('lgd', 'lgb'):
        assert np.array_equal(s0[k], r0[k]) and np.array_equal(s1[k], r1[k]), k
        assert not np.array_equal(s0[k], s1[k])           # the two ranks hold different angles

    # single rank, union minibatch (cnn_propagator/fullfield.py:343-351: the ranks' chunks tile the sorted minibatch)
    from beyond_dof_amd import util
    from beyond_dof_amd.solver import FullfieldSolver
    n, n_theta, mb, meas, init_d = w.problem()
    coords = util.rotation_lookup([n, n, n], n_theta)
    s = FullfieldSolver(n, n, n, n_theta, 2 * mb, 5000., 1e-7, free_prop_cm=1e-4, coord_ls=coords)
    s.set_measurements(meas)
    s.set_volume(init_d, 0.1 * init_d)
    s.reset_moments()
    sched = r0['sched']
    losses = []
    for i, chunk in enumerate(sched):
        losses.append(s.step(i, chunk, 1e-7, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, want_loss=True))
    g = s.gradient_to_host()
    # rank gradients are means over mb angles, summed over ranks and divided by size inside Adam: compare g / size
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(r0['gd'] / 2, g[0]) <= 1e-5 and rel(r0['gb'] / 2, g[1]) <= 1e-5
    # losses: mean over the rank's own angles; the union loss is the mean of the two
    assert abs(0.5 * (r0['losses'][-1] + r1['losses'][-1]) - losses[-1]) <= 1e-5 * abs(losses[-1])
    d, b = s.get_volume()
    lr = 1e-7
    diff = np.abs(d - r0['d'])
    assert np.mean(diff > 0.05 * lr) < 2e-3          # Adam's first steps are sign-like: see test_gpu_fullfield.py
    assert rel(r0['d'], d) <= 2e-3


def test_two_ranks_on_two_gpus_through_rccl(tmp_path):
    """The product's own multi-GPU path — RcclComm: socket rendezvous, ncclCommInitRank, reduce-scatter / all-gather in place
    at rank * count offsets on the communicator's stream — with two real ranks, one device each, in both forms of the
    exchange: bit-identical to the gloo rehearsal of the same two ranks.  Needs a box with >= 2 GPUs (RCCL refuses two ranks
    on one device); the single-GPU boxes of this pool skip it, the driver's multi-GPU node runs it."""
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import _lib
    if _lib.load().bdof_device_count() < 2:
        pytest.skip('needs two GPUs: RCCL does not accept two ranks on one device')
    for sharded in (False, True):
        g0, g1 = _run_two_ranks(tmp_path, sharded, backend='gloo')
        r0, r1 = _run_two_ranks(tmp_path, sharded, backend='rccl')
        for k in ('d', 'b', 'gd', 'gb'):
            assert np.array_equal(r0[k], r1[k]), (sharded, k)          # both ranks end alike
            assert np.array_equal(r0[k], g0[k]), (sharded, k)          # and like the gloo run
        assert np.allclose(r0['losses'], g0['losses'], rtol=0, atol=0)


def test_two_ranks_on_one_gpu_through_the_library_collectives(tmp_path):
    """The same product path as the test above — RcclComm: socket rendezvous, bdof_comm_create, reduce-scatter / all-gather
    in place at rank * count offsets, tickets on the communicator's stream (cnn_propagator/fullfield.py:343-353) — with
    nranks = 2 on ONE GPU: librccl is replaced by tests/rccl_stub (NCCL's semantics over shared memory, sums in rank order),
    because RCCL itself refuses two ranks on one device.  Everything on the library's side of the nccl* calls runs as it
    will on a multi-GPU node; the results equal the gloo rehearsal bit for bit in both forms of the exchange."""
    import __graft_entry__ as entry
    entry.build()
    stub = _build_rccl_stub(tmp_path)
    for sharded in (False, True):
        g0, g1 = _run_two_ranks(tmp_path, sharded, backend='gloo')
        r0, r1 = _run_two_ranks(tmp_path, sharded, backend='rccl', stub=stub)
        for k in ('d', 'b', 'gd', 'gb'):
            assert np.array_equal(r0[k], r1[k]), (sharded, k)          # both ranks end alike
            assert np.array_equal(r0[k], g0[k]), (sharded, k)          # and like the gloo run
        assert np.allclose(r0['losses'], g0['losses'], rtol=0, atol=0)


def test_bench_with_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` as the driver starts it (self-launch, socket rendezvous, RcclComm, tail tuning, barrier, max
    over ranks, one JSON line from rank 0) on a single GPU, the collectives bound to tests/rccl_stub: a rehearsal of the
    first N > 1 bench run at a small size."""
    import json
    import __graft_entry__ as entry
    entry.build()
    env = dict(os.environ, BDOF_RCCL_LIB=_build_rccl_stub(tmp_path), BDOF_STUB_SLOT_MB='128')
    env.pop('BDOF_COMM_BACKEND', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '128', '--angles-per-gpu', '4',
                        '--n-theta', '16', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-profile'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['value'] > 0 and d['scaling'] == 'weak'
    assert 'rccl' in d['config']['exchange'].lower() or 'reduce' in d['config']['exchange'].lower(), d['config']['exchange']
    assert d['config']['exchange_ms'] > 0
    err = p.stderr.decode()
    assert 'rccl communicator of 2 ranks ready' in err          # one line per rank with its device (comm.RcclComm.attach)


def _run_two_ptycho_ranks(tmp_path, sharded, backend='gloo', stub=None):
    procs = _launch_two('_dist_ptycho_worker.py', tmp_path, int(sharded), backend=backend, stub=stub)
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-3000:] for o in outs)
    return [np.load(str(tmp_path / 'pty{}_{}.npz'.format(r, int(sharded)))) for r in range(2)]


def test_two_rank_ptychography_through_the_library_collectives(tmp_path):
    """Probe positions sharded over two ranks (cnn_propagator/ptychography.py:292-306) through RcclComm and the library's
    collectives bound to tests/rccl_stub, on one GPU: the volumes of the gloo rehearsal, bit for bit, in both forms."""
    import __graft_entry__ as entry
    entry.build()
    stub = _build_rccl_stub(tmp_path)
    for sharded in (False, True):
        g0, _ = _run_two_ptycho_ranks(tmp_path, sharded)
        r0, r1 = _run_two_ptycho_ranks(tmp_path, sharded, backend='rccl', stub=stub)
        assert np.array_equal(r0['d'], r1['d']) and np.array_equal(r0['b'], r1['b'])
        assert np.array_equal(r0['d'], g0['d']) and np.array_equal(r0['b'], g0['b'])


def test_two_rank_ptychography_matches_the_union_minibatch(tmp_path):
    """Probe positions sharded over two ranks (cnn_propagator/ptychography.py:292-306): window/rotation adjoint, exchange and
    Adam through PtychoSolver.step in both forms of the exchange, against one rank holding all positions."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _dist_ptycho_worker as w
    vols = {}
    for sharded in (False, True):
        port = _free_port()
        procs = []
        for rank in range(2):
            env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank),
                       BDOF_COMM_BACKEND='gloo')
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_dist_ptycho_worker.py'), str(tmp_path), str(int(sharded))],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), '\n'.join(o[-3000:] for o in outs)
        r0, r1 = [np.load(str(tmp_path / 'pty{}_{}.npz'.format(r, int(sharded)))) for r in range(2)]
        assert np.array_equal(r0['d'], r1['d']) and np.array_equal(r0['b'], r1['b'])
        vols[sharded] = r0
    assert np.array_equal(vols[False]['d'], vols[True]['d']) and np.array_equal(vols[False]['b'], vols[True]['b'])
    from beyond_dof_amd import util
    from beyond_dof_amd.solver import PtychoSolver
    n, n_theta, psz, pos, init_d, meas, pr, pi = w.problem()
    s = PtychoSolver((n, n, n), psz, pos, n_theta, len(pos), 5000., 1e-7, pr, pi, coord_ls=util.rotation_lookup([n, n, n], n_theta))
    s.set_volume(init_d, 0.1 * init_d)
    s.reset_moments()
    for i, i_theta in enumerate((1, 3)):
        s.step(i, i_theta, np.arange(len(pos)), meas[i_theta], 1e-7)
    d, b = s.get_volume()
    lr = 1e-7
    # the union loss is the mean over 12 patterns, each rank's over its 6, summed and divided by size: the same gradient
    assert np.mean(np.abs(d - vols[True]['d']) > 0.05 * lr) < 2e-3
    assert np.linalg.norm(d - vols[True]['d']) <= 2e-3 * np.linalg.norm(d)


def test_bench_py_two_rank_rehearsal(tmp_path):
    """Plain `python bench.py --gpus 2`: the parent starts the two ranks itself (no launcher, no RANK / WORLD_SIZE in its
    environment), sharded schedule, tuned exchange, max-over-ranks timing, exactly one JSON line from rank 0.  The two
    ranks share this box's one GPU, so gloo carries the collective (BDOF_COMM_BACKEND=gloo); everything else is what an
    N-GPU run executes."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['BDOF_COMM_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '64', '--angles-per-gpu', '4',
                        '--n-theta', '16', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                           # exactly one JSON line on stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['config']['global_batch_angles'] == 8 and d['config']['allreduce_slabs'] in (1, 8, 16, 32) and d['config']['sharded_adam']
    assert d['config']['exchange'].startswith('gloo') and np.isfinite(d['final_loss'])
    assert d['roofline'] is not None and d['roofline']['frac'] > 0


def test_bench_py_single_rank_matches_contract():
    """`python bench.py` (N = 1): one JSON line with roofline + cpu_baseline, no collective."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--size', '64', '--angles-per-gpu', '4', '--n-theta', '16',
                        '--steps', '2', '--warmup', '1', '--cpu-slices', '8'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['config']['exchange'].startswith('none') and d['vs_baseline'] is None
    ro = d['roofline']
    assert ro['bound'] == 'hbm' and abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-12
    assert abs(ro['achieved'] * 1e9 - ro['algorithmic_bytes_per_launch'] / (ro['avg_launch_ms_events'] * 1e-3)) <= 1e-6 * ro['achieved'] * 1e9
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0


def test_two_rank_tiled_propagation(tmp_path):
    """cfg4's tiled propagation with the tiles dealt to two ranks (spatial decomposition): the stitched field,
    the loss, the object gradient and G(probe) equal the single-rank run's — the forward field bit for bit (a sum with zeros),
    the adjoint quantities to float32 summation order."""
    import __graft_entry__ as entry
    entry.build()
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _dist_tiling_worker as w
    procs = _launch_two('_dist_tiling_worker.py', tmp_path)
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-3000:] for o in outs)
    r0, r1 = [np.load(str(tmp_path / 'tiled_rank{}.npz'.format(r))) for r in range(2)]
    for k in ('wave', 'loss', 'gd', 'gb', 'gprobe'):
        assert np.array_equal(r0[k], r1[k]), k                     # every rank ends with the same field and gradient
    tp, one = w.run(None)
    assert tp.n_tiles == tp.n_tiles_field == 16
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert np.array_equal(r0['wave'], one['wave'])
    assert abs(float(r0['loss']) - float(one['loss'])) <= 1e-6 * abs(float(one['loss']))
    assert rel(r0['gd'], one['gd']) <= 2e-6 and rel(r0['gb'], one['gb']) <= 2e-6
    assert rel(r0['gprobe'], one['gprobe']) <= 2e-6
