"""Device-resident state of a reconstruction: the (delta, beta) volume, its Adam moments, the rotation tables and the
measured amplitudes, plus the Adam iteration built from libbdof calls.  This is the loop body of
cnn_propagator/fullfield.py:340-362 and ptychography.py:285-310 with every array kept in HBM.

One Adam iteration (`step`):
    forward + adjoint sweeps of this rank's wavefields                     bdof_loss_grad
    then, pipelined over x-slabs of the [X][Z][Y] volume (the "tail"):
      rotation adjoint of slab c                                          bdof_rotation_adjoint_rows
      gradient exchange of slab c across ranks                            bdof_reduce_scatter_grad | bdof_allreduce_grad
      regulariser + Adam + mask + clip on slab c (sharded: on this rank's 1/N of it)   bdof_adam_step_slab
      sharded: all-gather of the updated slab                             bdof_allgather_volume
The collectives run on the communicator's stream; the ctx stream only waits for slab c two slabs later, so the producer
of the next slabs and the consumer of the previous ones run while a slab is on the wire.  The TV stencil reads the
pre-update volume (x_old), which no slab overwrites: slab-wise, sharded and whole-volume execution give identical results.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import DeviceBuffer
from . import util
from .comm import PseudoComm
from .engine import MultisliceEngine


class _VolumeSolver(object):
    """What the full-field and ptychography solvers share: volume double buffer, gradient, moments, the tail of the step."""

    lookahead = 2              # slabs in flight between a collective's start and the Adam pass that consumes it

    def _init_volume(self):
        shape = (self.dim_x, self.dim_z, self.dim_y, 2)
        self.x = [DeviceBuffer.zeros(self.ctx, shape, np.float32), DeviceBuffer.zeros(self.ctx, shape, np.float32)]
        self.cur = 0
        self.g = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.m = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.v = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.mask = None
        self.nvox = self.dim_x * self.dim_z * self.dim_y
        self._plan = None
        self.tuned = None
        self._acc = 0              # minibatches accumulated in self.g since the last update (n_batch_per_update)
        self._epoch_plan = None    # shard layout of the Adam moments since the last reset_moments (None: none taken yet)
        self._g_shards = None      # slabs whose gradient is reduced on this rank's 1/size only (after a sharded step)
        self.probe = None          # enable_probe_optimization
        self.comm.attach(self.ctx)

    def _bind_volume(self):
        self.eng.set_volume(self.x[self.cur], self.dim_x * self.dim_z, self.dim_y, self.tab, self.dim_x, self.n_theta)

    # ---- state -------------------------------------------------------------------------------
    def set_volume(self, obj_delta, obj_beta):
        self.x[self.cur].upload(util.volume_to_rows(obj_delta, obj_beta))
        self._bind_volume()

    def get_volume(self):
        self.ctx.sync()
        return util.rows_to_volume(self.x[self.cur].download())

    def set_mask(self, mask):
        self.mask = None if mask is None else DeviceBuffer.from_host(
            self.ctx, np.ascontiguousarray(np.asarray(mask, dtype=np.float32).transpose(1, 2, 0)))

    def reset_moments(self):
        """m, v = (None, None) at the start of every epoch (fullfield.py:338, quirk Q10)."""
        lib, h = self.ctx.lib, self.ctx.handle
        self.ctx.check(lib.bdof_memset(h, self.m.ptr, 0, self.m.nbytes))
        self.ctx.check(lib.bdof_memset(h, self.v.ptr, 0, self.v.nbytes))
        self._acc = 0
        self._epoch_plan = None

    def bcast_volume(self, root=0):
        """Rank `root`'s volume to every rank (the init_*_temp.npy hand-over of ptychography.py:169-208)."""
        if self.comm.size > 1:
            self.comm.bcast_device(self.ctx, self.x[self.cur], root)
            self._bind_volume()

    def _get_loss(self):
        loss = ctypes.c_double(0)
        self.ctx.check(self.ctx.lib.bdof_get_loss(self.ctx.handle, ctypes.byref(loss)))
        return loss.value

    def _g_is_local(self):
        """self.g is about to be rewritten whole with this rank's own, unreduced gradient: the shard layout a sharded step
        left behind no longer describes it (gradient_to_host would otherwise gather stale parts over it)."""
        self._g_shards = None

    def gradient_to_host(self):
        """The gradient of the last step / loss_and_grad as (g_delta, g_beta), each (Y, X, Z).  After a sharded step every rank
        holds the reduced gradient of its own 1/size of each slab only: the parts are gathered first — the call is then a
        COLLECTIVE, every rank must make it.  After loss_and_grad (or a step in the all-reduce form) it is a local read-back."""
        if self._g_shards:
            per_x = self.dim_z * self.dim_y * 2
            for x0, nx in self._g_shards:
                self.comm.wait(self.ctx, self.comm.start_allgather(self.ctx, self.g, x0 * per_x, (nx // self.comm.size) * per_x))
            self._g_shards = None
        self.ctx.sync()
        return util.rows_to_volume(self.g.download())

    def regularizer(self, alpha_d=0.0, alpha_b=0.0, gamma=0.0):
        """alpha_d sum|delta| + alpha_b sum|beta| + gamma TV(delta) of the current volume (fullfield.py:109-118), on the device."""
        sums = (ctypes.c_double * 3)()
        self.ctx.check(self.ctx.lib.bdof_regularizer_value(self.ctx.handle, self.x[self.cur].ptr, self.dim_x, self.dim_z, self.dim_y, sums))
        return alpha_d * sums[0] + alpha_b * sums[1] + gamma * sums[2]

    # ---- the tail of the step ----------------------------------------------------------------
    def _reduces(self):
        return self.comm.size > 1 or getattr(self.comm, 'always_reduce', False)

    def slab_bounds(self, n_slabs):
        """x-plane ranges [(x0, nx)] of n_slabs nearly equal slabs of the [X][Z][Y] volume."""
        n_slabs = max(1, min(int(n_slabs), self.dim_x))
        edges = [(self.dim_x * i) // n_slabs for i in range(n_slabs + 1)]
        return [(edges[i], edges[i + 1] - edges[i]) for i in range(n_slabs)]

    def tail_plan(self, n_slabs=None, sharded=None):
        """(n_slabs, sharded) of the exchange.  sharded = reduce-scatter -> Adam on this rank's 1/size of each slab ->
        all-gather: needs every slab to split evenly over the ranks, else the all-reduce form is used.
        BDOF_ALLREDUCE_SLABS / BDOF_SHARDED_ADAM override; tune_tail() picks the slab count by measurement."""
        if not self._reduces():
            return 1, False
        if n_slabs is None and sharded is None and self._plan is not None:
            return self._plan
        if n_slabs is None:
            n_slabs = int(os.environ.get('BDOF_ALLREDUCE_SLABS', '8'))
        if sharded is None:
            sharded = getattr(self.comm, 'sharded', False) and os.environ.get('BDOF_SHARDED_ADAM', '1') != '0'
        n_slabs = max(1, min(int(n_slabs), self.dim_x))
        if sharded:
            n = n_slabs
            while n > 1 and self.dim_x % (n * self.comm.size):
                n -= 1
            if self.dim_x % (n * self.comm.size) == 0:
                n_slabs = n
            else:
                sharded = False
        return n_slabs, bool(sharded)

    def _adam_slab(self, i_update, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask, slab, g_scale):
        new = 1 - self.cur
        self.eng.adam_step(self.x[self.cur], self.x[new], self.g, self.m, self.v, self.mask if use_mask else None,
                           (self.dim_x, self.dim_z, self.dim_y), i_update, learning_rate, g_scale=g_scale,
                           alpha_d=alpha_d, alpha_b=alpha_b, gamma=gamma, clip=clip, slab=slab)

    def _flip(self):
        self.cur = 1 - self.cur
        self._bind_volume()

    def _whole_volume_guard(self):
        """adam_update runs Adam on the whole volume with this rank's moments: wrong once a sharded step has left them valid
        on 1/size of every slab only."""
        if self._epoch_plan is not None and self._epoch_plan[1] is not None:
            raise RuntimeError('adam_update() after a sharded step(): the Adam moments of this epoch are sharded over the ranks '
                               '(call reset_moments() first, or keep using step())')
        self._epoch_plan = ('taken', None)

    time_tail = False          # True: every _tail is bracketed by stream-ordered time stamps (bdof_timer_mark slots 0 / 1)

    def tail_ms(self):
        """Duration of the last tail on the ctx stream (rotation adjoint + exchange + Adam + all-gather), with time_tail on."""
        ms = ctypes.c_double(0)
        self.ctx.check(self.ctx.lib.bdof_timer_elapsed(self.ctx.handle, 0, 1, ctypes.byref(ms)))
        return ms.value

    def _tail(self, produce, i_update, learning_rate, alpha_d=0.0, alpha_b=0.0, gamma=0.0, clip=True, use_mask=True,
              n_slabs=None, sharded=None, flip=True, n_acc=1):
        if not self.time_tail:
            return self._tail_run(produce, i_update, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask, n_slabs, sharded, flip, n_acc)
        lib, h = self.ctx.lib, self.ctx.handle
        self.ctx.check(lib.bdof_timer_mark(h, 0))
        self._tail_run(produce, i_update, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask, n_slabs, sharded, flip, n_acc)
        self.ctx.check(lib.bdof_timer_mark(h, 1))

    def _tail_run(self, produce, i_update, learning_rate, alpha_d=0.0, alpha_b=0.0, gamma=0.0, clip=True, use_mask=True,
                  n_slabs=None, sharded=None, flip=True, n_acc=1):
        """produce(x0, nx): enqueue the kernels that leave this rank's gradient of x-planes [x0, x0+nx) in self.g.
        Then exchange + regulariser + Adam (+ mask, clip), slab by slab.  g_scale = 1 / (size * n_acc)
        (grads /= size, fullfield.py:351; accumulated minibatches are averaged, tensorflow_recon/fullfield.py:424)."""
        comm, ctx = self.comm, self.ctx
        g_scale = 1.0 / (comm.size * n_acc)
        args = (i_update, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask)
        if not self._reduces():
            produce(0, self.dim_x)
            self._adam_slab(*args, slab=None, g_scale=g_scale)
            if flip:
                self._flip()
            return
        n_slabs, sharded = self.tail_plan(n_slabs, sharded)
        # Sharded form: the Adam moments (and, after the reduce-scatter, the gradient) are valid on this rank's 1/size of every
        # slab only, so the shard layout must not move while the moments live — i.e. until the next reset_moments().  The
        # all-reduce form keeps them whole on every rank, whatever its slab count.
        layout = (n_slabs, True) if sharded else None
        if self._epoch_plan is None:
            self._epoch_plan = ('taken', layout)
        elif self._epoch_plan[1] != layout:
            raise RuntimeError('the exchange plan changed from {} to {} inside an epoch: the Adam moments are sharded by the first one '
                               '(call reset_moments() first)'.format(self._epoch_plan[1], layout))
        slabs = self.slab_bounds(n_slabs)
        self._g_shards = list(slabs) if sharded else None
        per_x = self.dim_z * self.dim_y * 2                  # floats per x-plane of the gradient / volume
        size, rank = comm.size, comm.rank
        tickets, gathers = [None] * len(slabs), []
        x_new = self.x[1 - self.cur]

        def finish(c):
            x0, nx = slabs[c]
            comm.wait(ctx, tickets[c])
            if sharded:
                w = nx // size
                self._adam_slab(*args, slab=(x0 + rank * w, w), g_scale=g_scale)
                gathers.append(comm.start_allgather(ctx, x_new, x0 * per_x, w * per_x))
            else:
                self._adam_slab(*args, slab=(x0, nx), g_scale=g_scale)

        for c, (x0, nx) in enumerate(slabs):
            produce(x0, nx)
            if sharded:
                tickets[c] = comm.start_reduce_scatter(ctx, self.g, x0 * per_x, (nx // size) * per_x)
            else:
                tickets[c] = comm.start_allreduce(ctx, self.g, x0 * per_x, (x0 + nx) * per_x)
            if c >= self.lookahead:
                finish(c - self.lookahead)
        for c in range(max(0, len(slabs) - self.lookahead), len(slabs)):
            finish(c)
        for t in gathers:
            comm.wait(ctx, t)
        if flip:
            self._flip()

    def tune_tail(self, candidates=(1, 8, 16, 32), reps=2):
        """Pick the number of slabs of the pipelined tail by timing it on this machine and process layout (the collective,
        the streams it runs on and the kernels either side interact in ways that differ between runtimes).  Dry run with
        learning rate 0 on a zeroed rotated-frame gradient: the gradient buffer, the spare volume buffer and the Adam
        moments are scratch afterwards (the moments are zeroed again), the volume is not touched.  Every rank takes the
        same decision (max of the timings over ranks).  `step` uses the result."""
        import time
        if not self._reduces() or os.environ.get('BDOF_ALLREDUCE_SLABS'):
            self._plan = self.tail_plan()
            return self._plan
        self._zero_rotated_gradient()
        plans = []
        for n in candidates:
            p = self.tail_plan(n)
            if p not in plans:
                plans.append(p)
        times = []
        for p in plans:
            self._epoch_plan = None                           # dry runs: the moments are scratch (zeroed again below)
            self._dry_tail(*p)                                # first use: communicator / stream set-up
            self.ctx.sync()
            self.comm.Barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                self._dry_tail(*p)
            self.ctx.sync()
            times.append(time.perf_counter() - t0)
        self._g_shards = None
        worst = self.comm.allreduce_max_host(np.array(times))
        self._plan = plans[int(np.argmin(worst))]
        self.tuned = {'{}slab{}'.format(p[0], '_sharded' if p[1] else ''): float(t) / reps for p, t in zip(plans, worst)}
        self.reset_moments()
        return self._plan

    def _zero_rotated_gradient(self):
        lib, h = self.ctx.lib, self.ctx.handle
        n = self.mb * self.dim_z * self.eng.nx * self.eng.ny * 8
        self.ctx.check(lib.bdof_memset(h, lib.bdof_grot(h), 0, n))

    # ---- optimisable probe (probe_type='optimizable', tensorflow_recon/fullfield.py:311-327,442-455) ---------------------
    def enable_probe_optimization(self, probe_real, probe_imag, probe_learning_rate=1e-3, pupil_function=None):
        """probe_real / probe_imag become optimisation variables with their own Adam (learning rate probe_learning_rate; its
        step counter runs over the whole level, as a tf.train.AdamOptimizer's does).  Every step() then also takes the
        gradient w.r.t. the probe (bdof_probe_grad: G(psi_0) summed over the minibatch), averages it over the ranks,
        updates the probe on the host in float64 — it is one (Y, X) field — multiplies by the pupil function
        (tensorflow_recon/fullfield.py:546-548) and hands it back to the engine, which re-derives its carrier (scalar, or
        the carrier field propagated on the device in float64)."""
        self.probe = (np.asarray(probe_real, dtype=np.float64) + 1j * np.asarray(probe_imag, dtype=np.float64)) * np.ones((self.eng.ny, self.eng.nx))
        self.probe_lr = float(probe_learning_rate)
        self.pupil = None if pupil_function is None else np.asarray(pupil_function, dtype=np.float64)
        self.probe_m = np.zeros_like(self.probe)
        self.probe_v = np.zeros_like(self.probe)
        self.probe_t = 0
        self._pacc = 0
        self.eng.enable_probe_grad(True)
        # The resident amplitudes were laid out as m - |a0| for the probe in force at set_measurements (residual splitting,
        # engine._set_meas_mode).  A probe that moves every step would leave that reference behind — every residual biased by
        # |a0_new| - |a0_old|, or by |a0| itself once the carrier changes kind — so the splitting is switched off here and
        # amplitudes that are already resident are put back to plain m.
        old_ref = getattr(self.eng, 'meas_ref', 0.0)
        self.eng.residual_split = False
        self.eng.set_probe(self.probe.real, self.probe.imag)
        if old_ref and getattr(self, 'meas', None) is not None:
            self.ctx.sync()
            self.meas.upload((self.meas.download().astype(np.float64) + old_ref).astype(np.float32))

    def _probe_collect(self):
        """Add this minibatch's probe gradient to the device accumulator (call after every loss_grad)."""
        if getattr(self, 'probe', None) is None:
            return
        self.eng.probe_grad(accumulate=self._pacc > 0, to_host=False)
        self._pacc += 1

    def _probe_apply(self, b1=0.9, b2=0.999, eps=1e-8):
        """Adam on (probe_real, probe_imag) with the accumulated gradient (mean over accumulated minibatches and ranks)."""
        if getattr(self, 'probe', None) is None or self._pacc == 0:
            return
        self.ctx.sync()
        g = np.ascontiguousarray(self.eng._gprobe.download().T).astype(np.complex128) / self._pacc
        self._pacc = 0
        if self.comm.size > 1:
            g = self.comm.allreduce_sum_host(g) / self.comm.size
        self.probe_t += 1
        self.probe_m = b1 * self.probe_m + (1 - b1) * g
        self.probe_v = b2 * self.probe_v + (1 - b2) * (g.real ** 2 + 1j * g.imag ** 2)
        mh = self.probe_m / (1 - b1 ** self.probe_t)
        vh = self.probe_v / (1 - b2 ** self.probe_t)
        self.probe = self.probe - self.probe_lr * (mh.real / (np.sqrt(vh.real) + eps) + 1j * mh.imag / (np.sqrt(vh.imag) + eps))
        if self.pupil is not None:
            self.probe = self.probe * self.pupil
        self.eng.set_probe(self.probe.real, self.probe.imag)

    def get_probe(self):
        return self.probe.real.copy(), self.probe.imag.copy()

    def shrink_wrap(self, thresh=1e-15):
        """mask = mask * (obj_delta > 1e-15)   (cnn_propagator/fullfield.py:365-368, intended behaviour, quirk Q8)."""
        if self.mask is not None:
            self.ctx.check(self.ctx.lib.bdof_mask_shrink(self.ctx.handle, self.x[self.cur].ptr, self.mask.ptr, self.nvox, thresh))


class FullfieldSolver(_VolumeSolver):
    def __init__(self, dim_y, dim_x, dim_z, n_theta, minibatch_size, energy_ev, psize_cm, free_prop_cm=None,
                 probe_real=None, probe_imag=None, variant='numpy_skip_last', comm=None, device=0, stream=None,
                 coord_ls=None, propagator='fft', kernel_size=17, recompute=None, rotation='nearest', theta=None, adjoint64=None,
                 detector_kernel='TF'):
        """propagator='fft': the transfer-function step of np_funcs.py (north-star path); 'conv': the truncated real-space
        kernel of propagation.py, what cnn_propagator/fullfield.py:87,102 calls (kernel_size taps per axis).
        rotation='nearest': the cnn variant's lookup tables, fused into the kernels (cnn_propagator/util.py:294-402);
        'bilinear': the TF twin's tf_rotate(obj, theta[i], 'BILINEAR') (tensorflow_recon/fullfield.py:96) with the true angles
        `theta` (radians, one per projection): the minibatch's rotated objects are materialised per step
        (bdof_rotate_bilinear), its adjoint is a gather (bdof_rotate_bilinear_adjoint)."""
        self.conv = propagator == 'conv'
        self.bilinear = rotation == 'bilinear'
        if adjoint64 not in (None, False, True, 'first'):
            raise ValueError("adjoint64 must be None, False, True or 'first'")
        if adjoint64 and rotation == 'bilinear' and (self.conv or adjoint64 == 'first'):
            raise ValueError("the float64 paths (adjoint64='first'; propagator='conv') run with the lookup-table rotation")
        # 'first': the first minibatch of every epoch through the model's float64 path on the same context (bdof_loss_grad_tf_f64 /
        # bdof_loss_grad_conv_f64), as PtychoSolver does it; True: every minibatch — the engine's float64 adjoint sweep
        # (transfer-function model) or the float64 path (real-space model, which has no such sweep)
        self.f64 = 'first' if adjoint64 == 'first' else (True if (adjoint64 is True and self.conv) else None)
        adjoint64 = adjoint64 is True and not self.conv
        if rotation not in ('nearest', 'bilinear'):
            raise ValueError("rotation must be 'nearest' or 'bilinear'")
        if self.bilinear and theta is None:
            raise ValueError("rotation='bilinear' needs the projection angles theta")
        self.dim_y, self.dim_x, self.dim_z = int(dim_y), int(dim_x), int(dim_z)
        self.n_theta, self.mb = int(n_theta), int(minibatch_size)
        self.comm = comm or PseudoComm()
        self.eng = MultisliceEngine(self.dim_y, self.dim_x, self.dim_z, self.mb, with_grad=True, device=device, stream=stream,
                                    recompute=recompute, adjoint64=adjoint64)
        self.ctx = self.eng.ctx
        self.eng.set_physics(energy_ev, psize_cm, free_prop_cm, variant=variant, detector_kernel=detector_kernel)
        if self.conv:
            self.eng.set_conv(energy_ev, psize_cm, kernel_size)
        if probe_real is None:
            probe_real, probe_imag = np.ones((dim_y, dim_x)), np.zeros((dim_y, dim_x))   # 'plane', fullfield.py:276-278
        self.eng.set_probe(probe_real, probe_imag)
        if self.f64:                                 # allocates the float64 wave + tape for the minibatch: fails here, not mid-run
            try:
                if self.conv:
                    self.eng.enable_conv_f64()
                else:
                    self.eng.enable_tf_f64()
            except ValueError as err:
                from ._lib import BdofError
                raise BdofError(str(err))
        if self.bilinear:
            # per-angle projective transform of tf.contrib.image.rotate for images of height X and width Z, in float64
            th = np.asarray(theta, dtype=np.float64)
            assert len(th) == self.n_theta
            H, W = self.dim_x, self.dim_z
            c, sn = np.cos(th), np.sin(th)
            self.rot_prm = np.stack([c, sn, ((W - 1) - (c * (W - 1) - sn * (H - 1))) / 2.0, ((H - 1) - (sn * (W - 1) + c * (H - 1))) / 2.0], axis=1)
            self.prm_buf = DeviceBuffer(self.ctx, self.mb * 32, np.float64, (self.mb, 4))
            self.tab = self.off = self.order = None
        else:
            # rotation lookup tables (cnn_propagator/util.py:294-347), uploaded once
            if coord_ls is None:
                coord_ls = util.rotation_lookup([dim_y, dim_x, dim_z], n_theta)
            tab, off, order = util.device_rotation_tables(coord_ls, self.dim_x, self.dim_z)
            self.tab = DeviceBuffer.from_host(self.ctx, tab)
            self.off = DeviceBuffer.from_host(self.ctx, off)
            self.order = DeviceBuffer.from_host(self.ctx, order)
        self._init_volume()
        self.meas = None
        self.meas_stage = DeviceBuffer(self.ctx, self.mb * self.dim_x * self.dim_y * 4, np.float32,
                                       (self.mb, self.dim_x, self.dim_y))
        self.angle_buf = DeviceBuffer(self.ctx, self.mb * 4, np.int32, (self.mb,))
        self._bind_volume()
        if not self.bilinear:
            self.eng.set_rotation_adjoint(self.off, self.order, self.dim_x * self.dim_z)

    def _bind_volume(self):
        if self.bilinear:
            return                   # the engine's object is the minibatch's rotated copy, rebuilt at every step (_rotate_batch)
        self.eng.set_volume(self.x[self.cur], self.dim_x * self.dim_z, self.dim_y, self.tab, self.dim_x, self.n_theta)

    def _rotate_batch(self, idx, B):
        """Bilinear rotation of the current volume to the B angles idx, bound as a batch of rotated objects: one pass that
        writes the rotated objects' modulation factors into the ctx (bdof_set_object_bilinear)."""
        lib, h = self.ctx.lib, self.ctx.handle
        prm = np.zeros((self.mb, 4))
        prm[:B] = self.rot_prm[np.asarray(idx)]
        self.prm_buf.upload(prm)
        self.ctx.check(lib.bdof_set_object_bilinear(h, self.x[self.cur].ptr, self.dim_x, self.dim_z, self.dim_y, self.prm_buf.ptr, B,
                                                    int(self.conv)))

    def set_measurements(self, prj_abs):
        """|prj| for every angle, (n_theta, Y, X) (loss uses np.abs(this_prj_batch), fullfield.py:106)."""
        self.meas = DeviceBuffer.from_host(self.ctx, self.eng.meas_layout(prj_abs))

    # ---- one Adam iteration ------------------------------------------------------------------
    def _stage_batch(self, angle_idx):
        lib, h = self.ctx.lib, self.ctx.handle
        idx = np.asarray(angle_idx, dtype=np.int32)
        assert len(idx) == self.mb
        self.angle_buf.upload(idx)
        # this_prj_batch = prj[this_ind_batch] (fullfield.py:344): one gather launch on the resident stack
        self.ctx.check(lib.bdof_gather_fields(h, self.meas_stage.ptr, self.meas.ptr, self.angle_buf.ptr, self.mb,
                                              self.dim_x * self.dim_y * 4))

    def _rot_loss_grad(self, angle_idx, f64=False):
        """Forward + adjoint sweeps of this rank's angles: the gradient w.r.t. the rotated objects stays in the ctx.
        f64: through the transfer-function model's float64 path on the same context (bdof_loss_grad_tf_f64)."""
        lib, h = self.ctx.lib, self.ctx.handle
        self._stage_batch(angle_idx)
        if f64:
            if self.bilinear:
                raise ValueError("f64: the float64 paths run with the lookup-table rotation")
            if self.conv and not getattr(self.eng, 'conv_f64', False):
                self.eng.enable_conv_f64()
            if not self.conv and not getattr(self.eng, 'tf_f64', False):
                self.eng.enable_tf_f64()
            fn = lib.bdof_loss_grad_conv_f64 if self.conv else lib.bdof_loss_grad_tf_f64
            self.ctx.check(fn(h, self.mb, self.angle_buf.ptr, None, None, self.meas_stage.ptr, float(getattr(self.eng, 'meas_ref', 0.0))))
            return
        fn = lib.bdof_loss_grad_conv if self.conv else lib.bdof_loss_grad
        if self.bilinear:
            self._rotate_batch(angle_idx, self.mb)
        self.ctx.check(fn(h, self.mb, None if self.bilinear else self.angle_buf.ptr, None, None, self.meas_stage.ptr, None))

    def _produce(self, accumulate=False):
        lib, h = self.ctx.lib, self.ctx.handle

        def produce(x0, nx):
            if self.bilinear:
                self.ctx.check(lib.bdof_rotate_bilinear_adjoint(h, lib.bdof_grot(h), self.dim_x, self.dim_z, self.dim_y, self.prm_buf.ptr,
                                                                self.mb, self.g.ptr, x0 * self.dim_z, nx * self.dim_z, int(accumulate), 1.0))
            else:
                self.ctx.check(lib.bdof_rotation_adjoint_rows(h, self.mb, self.angle_buf.ptr, self.g.ptr, x0 * self.dim_z,
                                                              nx * self.dim_z, int(accumulate), 1.0))
        return produce

    def loss_and_grad(self, angle_idx, want_loss=True, f64=False):
        """Data-term loss and its gradient w.r.t. the volume for this rank's angles (left in self.g, not reduced).  f64: both
        sweeps through the model's float64 path (a float64 twin of the fused kernels on the same context; the tests hold it to 3e-15 /
        2.5e-8 of the host's float64 restatement at the sizes that reaches: tests/test_gpu_parity.py)."""
        self._rot_loss_grad(angle_idx, f64=f64 or self.f64 is True)
        self._g_is_local()
        self._produce()(0, self.dim_x)
        return self._get_loss() if want_loss else None

    def adam_update(self, i_batch, learning_rate, alpha_d=0.0, alpha_b=0.0, gamma=0.0, clip=True, use_mask=True):
        """Regulariser + Adam on the gradient in self.g (whole volume, no exchange)."""
        self._whole_volume_guard()
        self._adam_slab(i_batch, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask, None, 1.0 / self.comm.size)
        self._flip()

    def _dry_tail(self, n_slabs, sharded):
        self.angle_buf.upload(np.arange(self.mb, dtype=np.int32) % self.n_theta)
        if self.bilinear:
            prm = self.rot_prm[np.arange(self.mb) % self.n_theta]
            self.prm_buf.upload(np.ascontiguousarray(prm))
        self._tail(self._produce(), 0, 0.0, n_slabs=n_slabs, sharded=sharded, flip=False)

    def step(self, i_batch, angle_idx, learning_rate, alpha_d=0.0, alpha_b=0.0, gamma=0.0, want_loss=False, n_slabs=None,
             sharded=None, use_mask=True, clip=True, n_batch_per_update=1, last_of_epoch=False):
        """grads = loss_grad(...); Allreduce; /size; Adam; mask; clip   (fullfield.py:345-362).

        n_batch_per_update > 1 (tensorflow_recon/fullfield.py:413-425,512-530): the volume gradient of consecutive
        minibatches is accumulated and applied (averaged) every n-th minibatch or at the last one of the epoch; the Adam
        bias-correction exponent then counts updates, not minibatches."""
        self._rot_loss_grad(angle_idx, f64=self.f64 is True or (self.f64 == 'first' and i_batch == 0 and getattr(self, 'probe', None) is None))
        self._probe_collect()
        nb = max(1, int(n_batch_per_update))
        if nb > 1:
            self._g_is_local()
            self._produce(accumulate=self._acc > 0)(0, self.dim_x)
            self._acc += 1
            if self._acc == nb or last_of_epoch:
                n_acc, self._acc = self._acc, 0
                self._tail(lambda x0, nx: None, i_batch // nb, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask, n_slabs,
                           sharded, n_acc=n_acc)
                self._probe_apply()
        else:
            self._tail(self._produce(), i_batch, learning_rate, alpha_d, alpha_b, gamma, clip, use_mask, n_slabs, sharded)
            self._probe_apply()
        return self._get_loss() if want_loss else None          # the loss of THIS minibatch stays on the device until the next one

    def forward_angles(self, angle_idx):
        """Detector waves (len(idx), Y, X) of the current volume — the forward_pass of fullfield.py:79-91."""
        idx = np.asarray(angle_idx, dtype=np.int32)
        out = []
        for i in range(0, len(idx), self.mb):
            chunk = idx[i:i + self.mb]
            if self.bilinear:
                self._rotate_batch(chunk, len(chunk))
                out.append(self.eng.forward(len(chunk), conv=self.conv))
            else:
                out.append(self.eng.forward(len(chunk), angle_idx=chunk, conv=self.conv))
        return np.concatenate(out, axis=0)


class PtychoSolver(_VolumeSolver):
    """Device-resident state of a ptychographic reconstruction (cnn_propagator/ptychography.py:285-310): volume,
    Adam moments, rotation tables, all diffraction amplitudes; windows are cut by index math inside the kernels."""

    def __init__(self, obj_size, probe_size, probe_pos, n_theta, minibatch_size, energy_ev, psize_cm, probe_real, probe_imag,
                 variant='numpy_skip_last', comm=None, device=0, stream=None, coord_ls=None, propagator='fft', kernel_size=17,
                 adjoint64=None):
        self.conv = propagator == 'conv'
        self.dim_y, self.dim_x, self.dim_z = [int(s) for s in obj_size]
        self.py, self.px = int(probe_size[0]), int(probe_size[1])
        self.n_theta, self.mb = int(n_theta), int(minibatch_size)
        self.comm = comm or PseudoComm()
        self.probe_pos = np.asarray(probe_pos).astype(int)
        self.half = (np.array(probe_size) / 2).astype('int')            # ptychography.py:138
        # small square probes: the LDS-resident engine (one launch per minibatch) also for the minibatches of ~20 positions the
        # drivers use, where the automatic choice would take the launch-bound streaming kernels for 64^2 / 128^2
        from .engine import RESIDENT_SIZES
        pin = 'resident' if (self.py == self.px and self.py in RESIDENT_SIZES and not self.conv
                             and not os.environ.get('BDOF_NO_RESIDENT_PIN')) else 'auto'
        self.eng = MultisliceEngine(self.py, self.px, self.dim_z, self.mb, with_grad=True, device=device, stream=stream, engine=pin,
                                    adjoint64=adjoint64 is True and not self.conv)
        self.ctx = self.eng.ctx
        self.eng.set_physics(energy_ev, psize_cm, 'inf', variant=variant)   # free_prop_cm='inf', ptychography.py:76
        if self.conv:
            self.eng.set_conv(energy_ev, psize_cm, kernel_size)
        self.eng.set_probe(probe_real, probe_imag)
        # adjoint64='first': the FIRST minibatch of every epoch runs through the model's float64 path on the SAME context
        # (bdof_loss_grad_tf_f64 / bdof_loss_grad_conv_f64: whole sweeps in double, no second engine) — Adam's first step after a
        # restart is lr g / (|g| + 1e-8), the only one in which a 1e-8 error of the gradient moves a voxel by a fraction of a
        # whole step; every later step runs on the fast engine.  f64 = 'first' / True says when it runs.
        self.f64 = None
        if adjoint64 == 'first' or (self.conv and adjoint64 is True):
            try:
                if self.conv:
                    self.eng.enable_conv_f64()
                else:
                    self.eng.enable_tf_f64()
            except ValueError as err:                      # a probe the float64 path does not take: let the caller decide
                from ._lib import BdofError
                raise BdofError(str(err))
            self.f64 = adjoint64
        elif adjoint64 not in (None, False, True):
            raise ValueError("adjoint64 must be None, False, True or 'first'")
        if coord_ls is None:
            coord_ls = util.rotation_lookup([self.dim_y, self.dim_x, self.dim_z], n_theta)
        tab, off, order = util.device_rotation_tables(coord_ls, self.dim_x, self.dim_z)
        self.tab = DeviceBuffer.from_host(self.ctx, tab)
        self.off = DeviceBuffer.from_host(self.ctx, off)
        self.order = DeviceBuffer.from_host(self.ctx, order)
        self._init_volume()
        self.meas_stage = DeviceBuffer(self.ctx, self.mb * self.py * self.px * 4, np.float32, (self.mb, self.py, self.px))
        self.idx_buf = DeviceBuffer(self.ctx, 4 * self.mb * 4, np.int32, (4, self.mb))
        self._bind_volume()
        self.eng.set_rotation_adjoint(self.off, self.order, self.dim_x * self.dim_z)
        self._last = None

    def set_measurements(self, prj_abs_all):
        """All diffraction amplitudes |prj| (n_theta, n_pos, py, px) resident on the device (a cfg5-sized dataset is 0.75 GB):
        the minibatch is then picked out by one gather launch instead of an HDF5 read + upload per step
        (this_prj_batch = prj[this_i_theta, this_ind_batch], ptychography.py:295)."""
        a = np.asarray(prj_abs_all)
        self.n_pos_all = a.shape[1]
        self.meas_all = DeviceBuffer.from_host(self.ctx, self.eng.meas_layout(a.reshape((-1,) + a.shape[2:])))

    def _stage(self, i_theta, pos_idx, prj_abs_batch):
        pos = self.probe_pos[np.asarray(pos_idx)]
        idx = np.empty((4, self.mb), dtype=np.int32)
        idx[0] = i_theta
        idx[1] = pos[:, 1] - self.half[1]            # window origin in x (axis 1), ptychography.py:68-70
        idx[2] = pos[:, 0] - self.half[0]            # window origin in y (axis 0)
        idx[3] = 0
        p = self.idx_buf.ptr
        if prj_abs_batch is None:
            if getattr(self, 'meas_all', None) is None:
                raise ValueError('no measurements: pass prj_abs_batch or call set_measurements first')
            idx[3] = int(i_theta) * self.n_pos_all + np.asarray(pos_idx)
            self.idx_buf.upload(idx)
            self.ctx.check(self.ctx.lib.bdof_gather_fields(self.ctx.handle, self.meas_stage.ptr, self.meas_all.ptr, p + 12 * self.mb, self.mb,
                                                           self.py * self.px * 4))
        else:
            self.idx_buf.upload(idx)
            self.meas_stage.upload(self.eng.meas_layout(prj_abs_batch))
        return p, p + 4 * self.mb, p + 8 * self.mb

    def _win_loss_grad(self, i_theta, pos_idx, prj_abs_batch, use64=False):
        a, xo, yo = self._stage(i_theta, pos_idx, prj_abs_batch)
        ctx = self.ctx
        if self.f64 is True or (use64 and self.f64 == 'first'):
            fn = ctx.lib.bdof_loss_grad_conv_f64 if self.conv else ctx.lib.bdof_loss_grad_tf_f64
            ctx.check(fn(ctx.handle, self.mb, a, xo, yo, self.meas_stage.ptr, float(getattr(self.eng, 'meas_ref', 0.0))))
        else:
            fn = ctx.lib.bdof_loss_grad_conv if self.conv else ctx.lib.bdof_loss_grad
            ctx.check(fn(ctx.handle, self.mb, a, xo, yo, self.meas_stage.ptr, None))
        self._last = (int(i_theta), xo, yo)

    def _produce_all(self):
        i_theta, xo, yo = self._last
        self._g_is_local()
        self.ctx.check(self.ctx.lib.bdof_window_rotation_adjoint(self.ctx.handle, self.mb, i_theta, xo, yo, self.g.ptr, 0, 1.0))

    def _get_loss(self):
        loss = ctypes.c_double(0)
        self.ctx.check(self.ctx.lib.bdof_get_loss(self.ctx.handle, ctypes.byref(loss)))
        return loss.value

    def loss_and_grad(self, i_theta, pos_idx, prj_abs_batch=None, want_loss=True, f64=False):
        """prj_abs_batch: |this_prj_batch| (mb, py, px) for probe positions pos_idx at angle i_theta (None: from the resident
        stack of set_measurements).  f64: through the model's float64 path (solver built with adjoint64='first')."""
        if f64 and self.f64 is None:
            raise ValueError("f64: build the solver with adjoint64='first' (the float64 path's buffers are allocated there)")
        self._win_loss_grad(i_theta, pos_idx, prj_abs_batch, use64=bool(f64))
        self._produce_all()
        return self._get_loss() if want_loss else None

    def _dry_tail(self, n_slabs, sharded):
        self._tail(lambda x0, nx: None, 0, 0.0, n_slabs=n_slabs, sharded=sharded, flip=False, use_mask=False)

    def step(self, i_batch, i_theta, pos_idx, prj_abs_batch=None, learning_rate=1.0, want_loss=False, n_slabs=None, sharded=None, clip=True):
        """One Adam iteration of ptychography.py:301-310: loss_grad, Allreduce, /size, Adam, clip (no regulariser, no mask).
        The window/rotation adjoint produces the whole volume gradient in one pass; exchange and Adam are still slab-wise."""
        self._win_loss_grad(i_theta, pos_idx, prj_abs_batch,
                            use64=self.f64 == 'first' and i_batch == 0 and self.probe is None)
        self._probe_collect()
        self._produce_all()
        self._tail(lambda x0, nx: None, i_batch, learning_rate, clip=clip, use_mask=False, n_slabs=n_slabs, sharded=sharded)
        self._probe_apply()
        return self._get_loss() if want_loss else None

    def forward(self, i_theta, pos_idx):
        pos = self.probe_pos[np.asarray(pos_idx)]
        return self.eng.forward(len(pos), angle_idx=[i_theta] * len(pos), xoff=pos[:, 1] - self.half[1],
                                yoff=pos[:, 0] - self.half[0], conv=self.conv)

    def adam_update(self, i_batch, learning_rate, clip=True):
        self._whole_volume_guard()
        self._adam_slab(i_batch, learning_rate, 0.0, 0.0, 0.0, clip, False, None, 1.0 / self.comm.size)
        self._flip()
