#!/usr/bin/env python
"""bench.py — Adam iterations of the full-field multislice reconstruction on synthetic data.

One "step" = one Adam iteration of cnn_propagator/fullfield.py:340-362 on this rank's minibatch of projection angles:
fused rotation + multislice forward, magnitude loss, hand-derived adjoint, rotation adjoint, gradient exchange across
ranks (RCCL reduce-scatter -> Adam on 1/N -> all-gather when N > 1), fused regulariser + Adam + mask + clip — the same
`FullfieldSolver.step` the entry point `reconstruct_fullfield` runs.
Workload (BASELINE.json configs[2] per GPU, weak scaling): 512^3 charcoal-like random (delta, beta) volume, 25 angles per
GPU per step, 5 keV, 1 nm voxels, free_prop_cm = 1e-4.

`python bench.py --gpus N` starts N ranks by itself (one process per GPU; this parent never touches a GPU) unless it is
already running as one rank of a launcher (`python -m torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE set).
Rank 0 prints ONE JSON line.  value = slice-steps/s = n_gpus * angles_per_gpu * slices * steps / time.

The roofline leg is a separate pass after the timed region: one step with the sub-batch streams off and every per-slice launch
made with hipExtLaunchKernelGGL, which stamps a HIP-event pair with the dispatch's own begin and end — a launch is the whole
minibatch and `bytes / duration` needs no assumption about what else shares the chip; the matching rocprofv3 summary is
profiles/<round>_kernel_stats_1stream.csv (same command with BDOF_STREAMS=1) and agrees with the events to 1-3 %.  The timed region itself runs the production configuration (two sub-batch streams).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROUND = 'r04'
# algorithmic bytes per pixel per launch (they sum to SURVEY §8(d)'s 104 B per slice-step)
BYTES_PER_PX = {'row_fwd': 24.0, 'col_prop': 16.0, 'row_bwd': 40.0, 'rot_adjoint': 8.0}
HBM_PEAK = 8.0e12
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/pmc_summary.py applies the gfx950
# corrections); a process cannot collect them on itself, so the committed summary is read back here
PMC_SUMMARY = os.path.join(ROOT, 'profiles', ROUND + '_pmc_traffic_bench.json')
STATS_1STREAM = os.path.join(ROOT, 'profiles', ROUND + '_kernel_stats_1stream.csv')
PMC_KERNEL = {'row_fwd': 'k_row_fwd<512, false, true', 'col_prop': 'k_row_prop<512', 'row_bwd': 'k_row_bwd<512, 1',
              'rot_adjoint': 'k_rot_adjoint'}


# the real-space path's kernels (bench.py --propagator conv, 17 taps) and its committed summaries
PMC_KERNEL_CONV = {'row_fwd': 'k_conv2<false, 8', 'row_bwd': 'k_conv2<true, 8', 'rot_adjoint': 'k_rot_adjoint'}
PMC_SUMMARY_CONV = os.path.join(ROOT, 'profiles', ROUND + '_pmc_traffic_conv.json')
STATS_1STREAM_CONV = os.path.join(ROOT, 'profiles', ROUND + '_kernel_stats_conv.csv')


def same_build(recorded):
    """A committed profile summary belongs to this run only if it was taken on the same kernel sources and compiler flags
    (beyond_dof_amd._lib.build_id: the summaries carry it, bench.py's line carries it)."""
    from beyond_dof_amd._lib import build_id
    mine = build_id()
    return bool(recorded) and recorded.get('source_sha256') == mine['source_sha256'] and recorded.get('flags') == mine['flags']


def pmc_traffic(kernel_class, n, mb, conv=False):
    """Measured L2<->fabric bytes per launch of the class's kernel (512^3, 25 angles, whole-batch launches), or None — also when
    the committed summary was taken on other sources or flags than the library being timed."""
    summary = PMC_SUMMARY_CONV if conv else PMC_SUMMARY
    if (n, mb) != (512, 25) or not os.path.exists(summary):
        return None
    doc = json.load(open(summary))
    if not same_build(doc.get('build')):
        return None
    want = (PMC_KERNEL_CONV if conv else PMC_KERNEL)[kernel_class]
    for name, k in doc['kernels'].items():
        if name.startswith(want):
            return k['total_bytes_per_launch']
    return None


def rocprof_avg_ms(kernel_class, n, mb, conv=False):
    """Average dispatch duration of the class's kernel in the committed single-stream rocprofv3 --kernel-trace --stats summary
    of this command (512^3, 25 angles only), or None."""
    stats = STATS_1STREAM_CONV if conv else STATS_1STREAM
    stamp = os.path.splitext(stats)[0] + '.build.json'
    if (n, mb) != (512, 25) or not os.path.exists(stats) or not os.path.exists(stamp) or not same_build(json.load(open(stamp))):
        return None
    import csv
    want = (PMC_KERNEL_CONV if conv else PMC_KERNEL)[kernel_class]
    calls = total = 0.0                       # a class may have several instances (the transfer-function step: forward and exact adjoint)
    for row in csv.DictReader(open(stats)):
        name = row['Name'].replace('void ', '')
        if name.startswith(want):
            total += float(row['TotalDurationNs'])
            # the rotation adjoint is TWO kernels per call of the class (k_rot_adjoint, then k_rot_adjoint_heavy for the clamped
            # border rows): their times add up, the calls are those of the first
            if not (kernel_class == 'rot_adjoint' and name.startswith('k_rot_adjoint_heavy')):
                calls += float(row['Calls'])
    return total / calls * 1e-6 if calls else None


def make_phantom(n, seed=3):
    """'charcoal-like': U(0, 2e-6) box-smoothed over 3 voxels, beta = 0.1 delta."""
    from scipy.ndimage import uniform_filter
    rng = np.random.default_rng(seed)
    d = rng.random((n, n, n), dtype=np.float32) * np.float32(2e-6)
    d = uniform_filter(d, size=3, mode='wrap')
    return d, (0.1 * d).astype(np.float32)


def _cpu_sample(args):
    """One worker of the CPU baseline: oracle forward + adjoint of one angle (numpy is single-threaded here)."""
    size, n_slice, seed = args
    for var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[var] = '1'
    from oracle import bdof_oracle as orc
    rng = np.random.default_rng(seed)
    delta = rng.uniform(0, 2e-6, size=(1, size, size, n_slice))
    beta = 0.1 * delta
    pr, pi = np.ones((size, size)), np.zeros((size, size))
    meas = np.ones((1, size, size))
    t0 = time.perf_counter()
    orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, 1e-4)
    return time.perf_counter() - t0


def cpu_baseline(size, n_slice, seed=11):
    """The oracle (numpy restatement of np_funcs.py + its adjoint, complex128) timed on the host: one process per core,
    one projection angle each — the reference's one-MPI-rank-per-core model (cnn_propagator/fullfield.py:343).
    Workers are plain subprocesses of this file (`--cpu-worker`), killed after a deadline: they never touch the GPU."""
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(size), str(n_slice), str(seed + i)],
                              stdout=subprocess.PIPE, env=env) for i in range(cores)]
    times = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
            times.append(float(out.decode().strip().splitlines()[-1]))
        except Exception:                      # noqa: BLE001  (deadline or a worker that died: drop it)
            p.kill()
    wall = time.perf_counter() - t0
    if not times:
        return None
    return {'value': len(times) * n_slice / wall, 'unit': 'slice-steps/s', 'cores': len(times), 'kind': 'port',
            'per_core_value': n_slice / (sum(times) / len(times)),
            'sample': '{0}x{0} wavefield, {1} slices, fwd+adjoint, complex128 numpy; {2} processes x 1 angle in {3:.1f} s wall '
                      '(incl. interpreter start)'.format(size, n_slice, len(times), wall)}


def self_launch(n_ranks):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this same command, one per GPU.  This process only
    spawns and waits — it never loads the HIP library — and rank 0's stdout is this process's stdout (one JSON line).
    The native control plane meets on a unix socket in a directory made here (0700, BDOF_RDZV): no port to lose between
    picking it and using it.  The gloo rehearsal backend still needs a TCP port; a rank-0 bind failure there is retried
    with a fresh one.  Every rank's stderr goes to a file of its own and is passed on afterwards, rank by rank."""
    import shutil
    import socket
    import tempfile
    rc = 1
    for attempt in range(3):
        tmp = tempfile.mkdtemp(prefix='bdof_bench_')               # 0700
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        procs, errs = [], []
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), BDOF_RDZV=os.path.join(tmp, 'rdzv.sock'))
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            errs.append(open(os.path.join(tmp, 'rank{}.err'.format(r)), 'w+'))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
        rc = 0
        try:
            while any(p.poll() is None for p in procs):
                time.sleep(0.2)
                bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
                if bad:                                    # one rank died: the others would wait for it forever
                    rc = bad[0]
                    break
        finally:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    p.kill()
        rc = rc or max((p.returncode or 0) for p in procs)
        texts = []
        for r, f in enumerate(errs):
            f.seek(0)
            texts.append(f.read())
            f.close()
        shutil.rmtree(tmp, ignore_errors=True)
        in_use = rc != 0 and any('EADDRINUSE' in t or 'ddress already in use' in t for t in texts)
        if in_use and attempt < 2:
            print('[bench] port {} was taken between picking and binding it: starting the ranks again'.format(port), file=sys.stderr)
            continue
        for r, t in enumerate(texts):
            for line in t.splitlines():
                print('[rank {}] {}'.format(r, line), file=sys.stderr)
        break
    return rc


def rehearse_cpu(args, out):
    """`--rehearse-cpu`: everything of an N-rank run that does not need a GPU — environment, rendezvous, schedule, barriers,
    max-over-ranks timing, one JSON line from rank 0 — with the collectives on host arrays (gloo, or the native control
    plane's socket star).  What tests/test_dist_gloo.py runs with 8 processes in the build container."""
    from beyond_dof_amd.comm import SocketGroup, comm_backend, minibatch_schedule
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
    mb = args.angles_per_gpu
    sched = minibatch_schedule(args.n_theta, world, mb, rng=np.random.default_rng(1234))
    mine = [chunk[rank * mb:(rank + 1) * mb] for chunk in sched]
    if comm_backend() == 'gloo' and world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo')
        def allmax(v):
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0])
        def allsum(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            dist.all_reduce(t)
            return t.numpy()
        barrier, close = dist.barrier, dist.destroy_process_group
    else:
        grp = SocketGroup(rank, world)
        allmax = lambda v: float(np.max(grp.allgather(np.array([v]))))
        allsum = lambda a: np.sum(grp.allgather(np.asarray(a)), axis=0)
        barrier, close = grp.barrier, grp.close
    barrier()
    t0 = time.perf_counter()
    g = np.zeros(1 << 16)
    for i in range(args.warmup + args.steps):
        g = allsum(np.full(1 << 16, float(rank + i)))              # stands in for the gradient exchange
    barrier()
    elapsed = allmax(time.perf_counter() - t0)
    seen = allsum(np.bincount(np.concatenate(mine), minlength=args.n_theta).astype(np.float64))
    if rank == 0:
        expect = sum(range(world)) + world * (args.warmup + args.steps - 1)
        out.write(json.dumps({'metric': 'rehearsal (no GPU work)', 'value': 0.0, 'unit': 'slice-steps/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': elapsed / max(1, args.steps) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'rehearsal': True,
                          'config': {'workload': 'cpu rehearsal of the launch / rendezvous / schedule path', 'backend': comm_backend(),
                                     'angles_covered': int((seen > 0).sum()), 'exchange_ok': bool(np.all(g == expect))}}) + '\n')
        out.flush()
    barrier()
    close()
    return 0


def device_copy_rate(ctx, nbytes=1 << 30, reps=4):
    """What a plain device-to-device copy reaches on THIS box right now, bytes read + written per second (stream-ordered time
    stamps around `reps` copies of `nbytes`): the practical ceiling next to the 8 TB/s datasheet figure the fractions are priced at."""
    import ctypes
    from beyond_dof_amd._lib import DeviceBuffer
    lib, h = ctx.lib, ctx.handle
    a, b = DeviceBuffer(ctx, nbytes), DeviceBuffer(ctx, nbytes)
    ctx.check(lib.bdof_memset(h, a.ptr, 1, nbytes))
    ctx.check(lib.bdof_memcpy_d2d(h, b.ptr, a.ptr, nbytes))
    ctx.sync()
    ctx.check(lib.bdof_timer_mark(h, 2))
    for _ in range(reps):
        ctx.check(lib.bdof_memcpy_d2d(h, b.ptr, a.ptr, nbytes))
    ctx.check(lib.bdof_timer_mark(h, 3))
    ctx.sync()
    ms = ctypes.c_double(0)
    ctx.check(lib.bdof_timer_elapsed(h, 2, 3, ctypes.byref(ms)))
    a.free()
    b.free()
    return 2.0 * nbytes * reps / (ms.value * 1e-3)


def roofline_pass(solver, batch, hyper, n, mb, S, conv=False):
    """One step with the sub-batch streams off and every launch bracketed by HIP events on the stream it is launched on."""
    eng = solver.eng
    eng.set_streams(1)
    solver.step(0, batch, **hyper)                     # same kernels, now as whole-batch launches: warm
    solver.ctx.sync()
    eng.profile_enable(True, stride=1)
    solver.step(0, batch, **hyper)
    solver.ctx.sync()
    prof = eng.profile_read()
    eng.profile_enable(False)
    eng.set_streams(-1)
    px = mb * n * n
    launches_per_step = {'row_fwd': S, 'col_prop': 2 * S, 'row_bwd': S, 'rot_adjoint': 1}
    # the real-space propagator has one kernel per slice and direction (k_conv: 24 B/px forward, 40 B/px backward),
    # timed under the same two classes; no transfer-function launches
    byte_model = {k: v for k, v in BYTES_PER_PX.items() if not (conv and k == 'col_prop')}
    per_class = {}
    for name, bpp in byte_model.items():
        cnt, ms = prof[name]
        if cnt:
            nbytes = bpp * px * (S if name == 'rot_adjoint' else 1)
            per_class[name] = {'timed_launches': cnt, 'avg_ms': ms / cnt, 'bytes_per_launch': nbytes,
                               'GBps': nbytes / (ms / cnt * 1e-3) / 1e9, 'frac': nbytes / (ms / cnt * 1e-3) / HBM_PEAK,
                               'avg_ms_rocprof_1stream': rocprof_avg_ms(name, n, mb, conv)}
    if not per_class:
        return None
    share = {k: per_class[k]['avg_ms'] * launches_per_step[k] for k in per_class}
    # dominant kernel = largest share of the step; classes within 5 % of the largest count as tied (the transfer-function
    # step and the adjoint row kernel are, at 24 ms each) and the tie goes to the one FURTHER from the roofline
    dom = min((k for k in share if share[k] >= 0.95 * max(share.values())), key=lambda k: per_class[k]['frac'])
    d = per_class[dom]
    tb = pmc_traffic(dom, n, mb, conv)
    return {'bound': 'hbm', 'kernel': dom, 'achieved': d['GBps'], 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': d['frac'],
            'traffic': None if tb is None else tb / (d['avg_ms'] * 1e-3) / 1e9,
            'traffic_bytes_per_launch': tb, 'algorithmic_bytes_per_launch': d['bytes_per_launch'],
            'avg_launch_ms_events': d['avg_ms'], 'avg_launch_ms_rocprof': d['avg_ms_rocprof_1stream'],
            'rocprof_summary': os.path.relpath(STATS_1STREAM_CONV if conv else STATS_1STREAM, ROOT) if d['avg_ms_rocprof_1stream'] else None,
            'traffic_source': (os.path.relpath(PMC_SUMMARY_CONV if conv else PMC_SUMMARY, ROOT) + ' (rocprofv3 --pmc FETCH_SIZE x2, WRITE_SIZE x1)') if tb else None,
            'share_of_step_ms': share,
            'note': 'dominant kernel = largest share of the step (ties within 5 % go to the lower frac); achieved = algorithmic bytes of '
                    'one whole-minibatch launch / its average duration from the HIP events hipExtLaunchKernelGGL stamps with the '
                    'dispatch begin and end, in a single-stream pass after the timed region (the timed region runs two sub-batch '
                    'streams: its figure is whole_step_frac)',
            'per_kernel': per_class}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == '--cpu-worker':
        print(_cpu_sample((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))))
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--angles-per-gpu', type=int, default=25)
    ap.add_argument('--n-theta', type=int, default=200)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-slices', type=int, default=192)
    ap.add_argument('--no-profile', action='store_true', help='skip the roofline pass')
    ap.add_argument('--recompute', action='store_true', help='tape-free adjoint (bdof_configure flag 16): psi_z is marched back')
    ap.add_argument('--rotation', default='nearest', choices=['nearest', 'bilinear'],
                    help="'bilinear': the TF twin's tf_rotate instead of the cnn variant's fused nearest-neighbour tables, for comparison")
    ap.add_argument('--propagator', default='fft', choices=['fft', 'conv'],
                    help="'conv': the reference entry points' truncated real-space kernel (17 taps), for comparison")
    ap.add_argument('--rehearse-cpu', action='store_true', help='launch, rendezvous, schedule and timing path of an N-rank run without GPU work')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)

    # stdout carries exactly one JSON line: native libraries that print there (RCCL's version banner does) are sent to
    # stderr by pointing fd 1 at fd 2 for the rest of the run; the JSON line goes to a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if args.rehearse_cpu:
        return rehearse_cpu(args, real_stdout)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))

    from beyond_dof_amd import _lib
    from beyond_dof_amd.comm import comm_backend, get_comm, minibatch_schedule
    from beyond_dof_amd.solver import FullfieldSolver

    comm = get_comm()                      # N > 1: RcclComm (RCCL behind the C ABI); BDOF_COMM_BACKEND=gloo: rehearsal
    device = local_rank
    if world > 1 and (comm_backend() == 'gloo' or os.environ.get('BDOF_RCCL_LIB')):
        # several rehearsal ranks may share one GPU: through gloo, or through the library's own collectives bound to a stand-in
        # for librccl (tests/rccl_stub; RCCL itself refuses two ranks on one device)
        device = local_rank % max(1, _lib.load().bdof_device_count())

    n, mb, n_theta = args.size, args.angles_per_gpu, args.n_theta
    sched = minibatch_schedule(n_theta, world, mb, rng=np.random.default_rng(1234))
    my_batches = [chunk[rank * mb:(rank + 1) * mb] for chunk in sched]
    my_angles = np.unique(np.concatenate(my_batches[:min(len(my_batches), args.steps + args.warmup)]))

    t_setup = time.time()
    solver = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, comm=comm, device=device,
                             propagator=args.propagator, recompute=args.recompute, rotation=args.rotation,
                             theta=-np.linspace(0, 2 * np.pi, n_theta) if args.rotation == 'bilinear' else None)
    true_d, true_b = make_phantom(n)
    solver.set_volume(true_d, true_b)
    meas = np.zeros((n_theta, n, n), dtype=np.float32)
    meas[my_angles] = np.abs(solver.forward_angles(my_angles))       # synthetic data from our own forward model
    solver.set_measurements(meas)
    del meas
    rng = np.random.default_rng(100)                                   # same initial guess on every rank
    init_d = np.clip(rng.normal(8.7e-7, 1e-7, size=(n, n, n)), 0, None).astype(np.float32)   # fullfield.py:250-253
    init_b = np.clip(rng.normal(5.1e-8, 1e-8, size=(n, n, n)), 0, None).astype(np.float32)
    solver.set_volume(init_d, init_b)
    solver.set_mask(np.ones((n, n, n), dtype=np.float32))
    del true_d, true_b, init_d, init_b
    hyper = dict(learning_rate=1e-7, alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)     # params_cone, reconstruct_fullfield.py:50-56
    n_slabs, sharded = solver.tune_tail()     # slab count of the exchange + Adam pipeline: timed here, outside the run
    if rank == 0:
        print('[bench] setup {:.1f} s; exchange: {} slab(s){} {}'.format(time.time() - t_setup, n_slabs, ', sharded Adam' if sharded else '',
                                                                       solver.tuned or ''), file=sys.stderr)

    def run(i):
        solver.step(i % len(my_batches), my_batches[i % len(my_batches)], want_loss=False, **hyper)

    for i in range(args.warmup):
        run(i)
    solver.ctx.sync()
    comm.Barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        run(i)
    solver.ctx.sync()
    comm.Barrier()
    elapsed = time.perf_counter() - t0
    if comm.size > 1:
        elapsed = float(comm.allreduce_max_host(np.array([elapsed]))[0])
    # where the tail of a step goes (rotation adjoint, gradient exchange, Adam, all-gather), from stream-ordered time stamps in
    # two extra steps after the timed region; max over ranks
    solver.time_tail = True
    tail = []
    for i in range(2):
        run(args.warmup + args.steps + i)
        tail.append(solver.tail_ms())
    solver.time_tail = False
    tail_ms = float(np.mean(tail))
    if comm.size > 1:
        tail_ms = float(comm.allreduce_max_host(np.array([tail_ms]))[0])
    groups = solver.eng.batch_groups(mb)
    S = n
    roof = None if args.no_profile or args.rotation != 'nearest' else roofline_pass(solver, my_batches[0], hyper, n, mb, S, conv=args.propagator == 'conv')
    loss = solver.loss_and_grad(my_batches[0], want_loss=True)
    copy_rate = device_copy_rate(solver.ctx) if (rank == 0 and roof is not None) else None
    comm.Barrier()

    if rank == 0:
        slice_steps = world * mb * S * args.steps
        if roof is not None:
            roof['whole_step_frac'] = (72.0 if args.propagator == 'conv' else 104.0) * n * n * (slice_steps / world) / elapsed / HBM_PEAK
            roof['sub_batch_streams_in_timed_region'] = groups
            roof['device_copy_GBps'] = copy_rate / 1e9
            roof['device_copy_note'] = ('a plain device-to-device copy of 1 GiB on this box, read + written bytes per second: the practical '
                                        'HBM ceiling beside the 8 TB/s the fractions are priced at (whole_step_frac x 8000 / this = the step '
                                        'relative to a copy)')
        out = {'metric': 'multislice fwd+adjoint slice-steps/s (full Adam iteration: rotation, forward, loss, adjoint, '
                         'gradient exchange, regulariser+Adam)',
               'value': slice_steps / elapsed, 'unit': 'slice-steps/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'adam_iters_per_s': args.steps / elapsed, 'final_loss': loss,
               'config': {'workload': '{3}: {0}^3 charcoal-like random (delta,beta) volume, {1} of {2} angles per GPU per '
                                      'Adam step, {0} slices, 5 keV, 1 nm, free_prop_cm=1e-4, plane probe'.format(
                                          n, mb, n_theta, 'cfg3' if (n, n_theta) == (512, 200) else 'cfg3-shaped (non-default size)'),
                          'global_batch_angles': world * mb, 'parallelism': 'angle-sharded dp{}'.format(world),
                          'exchange': ('none (1 rank)' if world == 1 and not getattr(comm, 'always_reduce', False) else
                                       '{} ({} slab(s), {})'.format(comm.backend, n_slabs, 'reduce-scatter + sharded Adam + all-gather'
                                                                    if sharded else 'all-reduce')),
                          'allreduce_slabs': n_slabs, 'sharded_adam': bool(sharded), 'exchange_ms': tail_ms, 'tail_tuning_ms': solver.tuned,
                          'exchange_ms_note': 'tail of a step on the ctx stream: rotation adjoint + gradient exchange + Adam (+ all-gather), '
                                              'slab-pipelined; max over ranks, two steps after the timed region',
                          'propagator': args.propagator, 'rotation': args.rotation,
                          'adjoint': 'recompute (tape-free)' if args.recompute else 'tape',
                          'transform_constants': 'one float32 table' if os.environ.get('BDOF_TW_DITHER') in ('0', '1') else
                          'dithered over the slices, {} copies (DESIGN §5)'.format(os.environ.get('BDOF_TW_DITHER', '64')),
                          'transfer_function': 'one float32 table' if os.environ.get('BDOF_H_DITHER') in ('0', '1') else
                          'dithered over the slices, {} copies'.format(os.environ.get('BDOF_H_DITHER', '64')),
                          'build': _lib.build_id(),
                          'hbm_used_GiB': solver.ctx.mem_used() / 2.0 ** 30},
               'roofline': roof}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n, args.cpu_slices)
        real_stdout.write(json.dumps(out) + '\n')
        real_stdout.flush()
    sys.stdout.flush()
    comm.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
