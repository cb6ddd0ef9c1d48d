#!/usr/bin/env python
"""bench.py — Adam iterations of the full-field multislice reconstruction on synthetic data.

One "step" = one Adam iteration of cnn_propagator/fullfield.py:340-362 on this rank's minibatch of
projection angles: fused rotation + multislice forward, magnitude loss, hand-derived adjoint, rotation
adjoint, (RCCL all-reduce of the volume gradient when N > 1), fused regulariser + Adam + mask + clip.
Workload (BASELINE.json configs[2] per GPU, weak scaling): 512^3 charcoal-like random (delta, beta)
volume, 25 angles per GPU per step, 5 keV, 1 nm voxels, free_prop_cm = 1e-4.

Prints ONE JSON line (rank 0).  value = slice-steps/s = n_gpus * angles_per_gpu * slices * steps / time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per pixel per launch (they sum to SURVEY §8(d)'s 104 B per slice-step)
BYTES_PER_PX = {'row_fwd': 24.0, 'col_prop': 16.0, 'row_bwd': 40.0, 'rot_adjoint': 8.0}
HBM_PEAK = 8.0e12
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/pmc_summary.py applies the gfx950
# corrections); a process cannot collect them on itself, so the committed summary is read back here
PMC_SUMMARY = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_bench.json')
PMC_KERNEL = {'row_fwd': 'k_row_fwd<512, false, true>', 'col_prop': 'k_row_prop<512>', 'row_bwd': 'k_row_bwd<512, 1>',
              'rot_adjoint': 'k_rot_adjoint'}


def pmc_traffic(kernel_class, n, mb):
    """Measured L2<->fabric bytes per launch of the class's kernel (512^3, 25 angles only), or None."""
    if (n, mb) != (512, 25) or not os.path.exists(PMC_SUMMARY):
        return None
    k = json.load(open(PMC_SUMMARY))['kernels'].get(PMC_KERNEL[kernel_class])
    return k['total_bytes_per_launch'] if k else None


def rocprof_avg_ms(kernel_class, n, mb):
    """Average dispatch duration of the class's kernel in the committed rocprofv3 --kernel-trace --stats summary of this
    same command (profiles/r01_kernel_stats.csv; 512^3, 25 angles only), or None."""
    path = os.path.join(ROOT, 'profiles', 'r01_kernel_stats.csv')
    if (n, mb) != (512, 25) or not os.path.exists(path):
        return None
    import csv
    want = PMC_KERNEL[kernel_class]
    for row in csv.DictReader(open(path)):
        if row['Name'].replace('void ', '').startswith(want):
            return float(row['AverageNs']) * 1e-6
    return None


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
    import subprocess
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


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == '--cpu-worker':
        print(_cpu_sample((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--angles-per-gpu', type=int, default=25)
    ap.add_argument('--n-theta', type=int, default=200)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-slices', type=int, default=192)
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--propagator', default='fft', choices=['fft', 'conv'],
                    help="'conv': the reference entry points' truncated real-space kernel (17 taps), for comparison")
    ap.add_argument('--profile-stride', type=int, default=64,
                    help='HIP-event time every n-th launch of the per-slice kernels (each timed launch costs ~9 us of stream time)')
    args = ap.parse_args()

    # stdout carries exactly one JSON line: native libraries that print there (RCCL's version banner does) are sent to
    # stderr by pointing fd 1 at fd 2 for the rest of the run; the JSON line goes to a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))

    from beyond_dof_amd.comm import PseudoComm, TorchComm, minibatch_schedule
    from beyond_dof_amd.solver import FullfieldSolver

    comm = PseudoComm()
    if world > 1 or os.environ.get('BDOF_FORCE_TORCH_COMM'):      # the env switch exercises the RCCL path on one GPU
        import torch
        torch.cuda.set_device(local_rank)
        comm = TorchComm(os.environ.get('BDOF_COMM_BACKEND', 'nccl'))      # gloo: rehearsal with several ranks on one GPU

    n, mb, n_theta = args.size, args.angles_per_gpu, args.n_theta
    sched = minibatch_schedule(n_theta, world, mb, rng=np.random.default_rng(1234))
    my_batches = [chunk[rank * mb:(rank + 1) * mb] for chunk in sched]
    my_angles = np.unique(np.concatenate(my_batches[:min(len(my_batches), args.steps + args.warmup)]))

    t_setup = time.time()
    solver = FullfieldSolver(n, n, n, n_theta, mb, 5000., 1e-7, free_prop_cm=1e-4, comm=comm, device=local_rank,
                             propagator=args.propagator)
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
    n_slabs = solver.tune_allreduce()          # slab-pipelined vs whole-volume all-reduce tail: timed here, outside the run
    if rank == 0:
        print('[bench] setup {:.1f} s; all-reduce tail: {} slab(s) {}'.format(time.time() - t_setup, n_slabs, solver.tuned or ''),
              file=sys.stderr)

    def run(i):
        solver.step(i % len(my_batches), my_batches[i % len(my_batches)], want_loss=False, **hyper)

    for i in range(args.warmup):
        run(i)
    solver.ctx.sync()
    comm.Barrier()
    if not args.no_profile:
        solver.eng.profile_enable(True, stride=args.profile_stride)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        run(i)
    solver.ctx.sync()
    comm.Barrier()
    elapsed = time.perf_counter() - t0
    if comm.size > 1:
        elapsed = float(comm.allreduce_sum_host(np.array([elapsed if r == rank else 0.0 for r in range(world)])).max())
    prof = solver.eng.profile_read() if not args.no_profile else {}
    loss = solver.loss_and_grad(my_batches[0], want_loss=True)

    if rank == 0:
        S = n
        slice_steps = world * mb * S * args.steps
        px = mb * n * n
        roof = None
        if prof:
            # A batch may run as `groups` sub-batches on concurrent streams (bdof_set_streams): every per-slice launch
            # then carries 1/groups of the batch and shares the chip with the same launch of the other groups, so the
            # chip-wide rate of the kernel is groups x (bytes of one launch / its duration).  Both figures are reported.
            groups = solver.eng.batch_groups(mb)
            per_class = {}
            for name, bpp in BYTES_PER_PX.items():
                cnt, ms = prof[name]
                if cnt:
                    g = 1 if name == 'rot_adjoint' else groups
                    nbytes = bpp * px * (S if name == 'rot_adjoint' else 1) / g
                    per_class[name] = {'timed_launches': cnt, 'avg_ms': ms / cnt, 'bytes_per_launch': nbytes,
                                       'concurrent_launches': g, 'GBps_one_launch': nbytes / (ms / cnt * 1e-3) / 1e9,
                                       'GBps': g * nbytes / (ms / cnt * 1e-3) / 1e9}
            launches_per_step = {'row_fwd': S, 'col_prop': 2 * S, 'row_bwd': S, 'rot_adjoint': 1}
            dom = max(per_class, key=lambda k: per_class[k]['avg_ms'] * launches_per_step[k] * (1 if k == 'rot_adjoint' else groups))
            ach = per_class[dom]['GBps']
            tb = pmc_traffic(dom, n, mb)
            g = per_class[dom]['concurrent_launches']
            if tb is not None:
                # the PMC passes serialise kernels, so the library may have run them un-split (one launch per batch) while
                # the timed run splits the batch in g sub-batches: bring the measured bytes to the timed run's launch size
                full = BYTES_PER_PX[dom] * px * (S if dom == 'rot_adjoint' else 1)
                g_pmc = 1 if tb > 0.75 * full else g
                tb = tb * g_pmc / g
            roof = {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                    'frac': ach * 1e9 / HBM_PEAK,
                    'traffic': None if tb is None else g * tb / (per_class[dom]['avg_ms'] * 1e-3) / 1e9,
                    'concurrent_launches': g,
                    'avg_launch_ms_events': per_class[dom]['avg_ms'],
                    'avg_launch_ms_rocprof': rocprof_avg_ms(dom, n, mb),
                    'note': ('achieved = concurrent_launches x algorithmic bytes of one launch / its event interval; with two '
                             'sub-batch streams the event interval on a stream (end of its previous kernel -> end of this one) '
                             'includes the wait for CU slots the other stream holds, rocprofv3 reports the dispatch alone '
                             '(avg_launch_ms_rocprof, from the committed profiles/r01_kernel_stats.csv)')
                            if g > 1 else 'achieved = algorithmic bytes of one launch / its average duration',
                    'traffic_bytes_per_launch': tb, 'algorithmic_bytes_per_launch': per_class[dom]['bytes_per_launch'],
                    'traffic_source': 'profiles/r01_pmc_traffic_bench.json (rocprofv3 --pmc FETCH_SIZE x2, WRITE_SIZE x1)' if tb else None,
                    'per_kernel': per_class,
                    'whole_step_frac': 104.0 * n * n * (slice_steps / world) / elapsed / HBM_PEAK}
        out = {'metric': 'multislice fwd+adjoint slice-steps/s (full Adam iteration: rotation, forward, loss, adjoint, '
                         'gradient all-reduce, regulariser+Adam)',
               'value': slice_steps / elapsed, 'unit': 'slice-steps/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'adam_iters_per_s': args.steps / elapsed, 'final_loss': loss,
               'config': {'workload': 'cfg3: {0}^3 charcoal-like random (delta,beta) volume, {1} of {2} angles per GPU per '
                                      'Adam step, {0} slices, 5 keV, 1 nm, free_prop_cm=1e-4, plane probe'.format(n, mb, n_theta),
                          'global_batch_angles': world * mb, 'parallelism': 'angle-sharded dp{}'.format(world), 'allreduce_slabs': n_slabs,
                          'propagator': args.propagator},
               'roofline': roof}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n, args.cpu_slices)
        real_stdout.write(json.dumps(out) + '\n')
        real_stdout.flush()
    sys.stdout.flush()
    comm.close()


if __name__ == '__main__':
    main()
