"""Device-resident state of a full-field reconstruction: the (delta, beta) volume, its Adam moments,
the rotation tables and the measured amplitudes, plus the Adam iteration built from libbdof calls.
This is the loop body of cnn_propagator/fullfield.py:340-362 with every array kept in HBM."""
import os

import numpy as np

from . import _lib
from ._lib import DeviceBuffer
from . import util
from .comm import PseudoComm
from .engine import MultisliceEngine


class FullfieldSolver(object):
    def __init__(self, dim_y, dim_x, dim_z, n_theta, minibatch_size, energy_ev, psize_cm, free_prop_cm=None,
                 probe_real=None, probe_imag=None, variant='numpy_skip_last', comm=None, device=0, stream=None,
                 coord_ls=None, propagator='fft', kernel_size=17):
        """propagator='fft': the transfer-function step of np_funcs.py (north-star path); 'conv': the truncated real-space
        kernel of propagation.py, what cnn_propagator/fullfield.py:87,102 calls (kernel_size taps per axis)."""
        self.conv = propagator == 'conv'
        self.dim_y, self.dim_x, self.dim_z = int(dim_y), int(dim_x), int(dim_z)
        self.n_theta, self.mb = int(n_theta), int(minibatch_size)
        self.comm = comm or PseudoComm()
        self.eng = MultisliceEngine(self.dim_y, self.dim_x, self.dim_z, self.mb, with_grad=True, device=device, stream=stream)
        self.ctx = self.eng.ctx
        self.eng.set_physics(energy_ev, psize_cm, free_prop_cm, variant=variant)
        if self.conv:
            self.eng.set_conv(energy_ev, psize_cm, kernel_size)
        if probe_real is None:
            probe_real, probe_imag = np.ones((dim_y, dim_x)), np.zeros((dim_y, dim_x))   # 'plane', fullfield.py:276-278
        self.eng.set_probe(probe_real, probe_imag)
        # rotation lookup tables (cnn_propagator/util.py:294-347), uploaded once
        if coord_ls is None:
            coord_ls = util.rotation_lookup([dim_y, dim_x, dim_z], n_theta)
        tab, off, order = util.device_rotation_tables(coord_ls, self.dim_x, self.dim_z)
        self.tab = DeviceBuffer.from_host(self.ctx, tab)
        self.off = DeviceBuffer.from_host(self.ctx, off)
        self.order = DeviceBuffer.from_host(self.ctx, order)
        nvox = self.dim_x * self.dim_z * self.dim_y
        shape = (self.dim_x, self.dim_z, self.dim_y, 2)
        self.x = [DeviceBuffer.zeros(self.ctx, shape, np.float32), DeviceBuffer.zeros(self.ctx, shape, np.float32)]
        self.cur = 0
        self.g = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.m = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.v = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.mask = None
        self.meas = None
        self.meas_stage = DeviceBuffer(self.ctx, self.mb * self.dim_x * self.dim_y * 4, np.float32,
                                       (self.mb, self.dim_x, self.dim_y))
        self.angle_buf = DeviceBuffer(self.ctx, self.mb * 4, np.int32, (self.mb,))
        self.nvox = nvox
        self._n_slabs = None
        self.tuned = None
        self._bind_volume()
        self.eng.set_rotation_adjoint(self.off, self.order, self.dim_x * self.dim_z)

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

    def set_measurements(self, prj_abs):
        """|prj| for every angle, (n_theta, Y, X) (loss uses np.abs(this_prj_batch), fullfield.py:106)."""
        self.meas = DeviceBuffer.from_host(self.ctx, self.eng.meas_layout(prj_abs))

    def reset_moments(self):
        """m, v = (None, None) at the start of every epoch (fullfield.py:338, quirk Q10)."""
        lib, h = self.ctx.lib, self.ctx.handle
        self.ctx.check(lib.bdof_memset(h, self.m.ptr, 0, self.m.nbytes))
        self.ctx.check(lib.bdof_memset(h, self.v.ptr, 0, self.v.nbytes))

    # ---- one Adam iteration ------------------------------------------------------------------
    def _stage_batch(self, angle_idx):
        lib, h = self.ctx.lib, self.ctx.handle
        idx = np.asarray(angle_idx, dtype=np.int32)
        assert len(idx) == self.mb
        self.angle_buf.upload(idx)
        per = self.dim_x * self.dim_y * 4
        for b, j in enumerate(idx):
            self.ctx.check(lib.bdof_memcpy_d2d(h, self.meas_stage.ptr + b * per, self.meas.ptr + int(j) * per, per))

    def _get_loss(self):
        import ctypes
        loss = ctypes.c_double(0)
        self.ctx.check(self.ctx.lib.bdof_get_loss(self.ctx.handle, ctypes.byref(loss)))
        return loss.value

    def _rot_loss_grad(self, angle_idx):
        """Forward + adjoint sweeps of this rank's angles: the gradient w.r.t. the rotated objects stays in the ctx."""
        lib, h = self.ctx.lib, self.ctx.handle
        self._stage_batch(angle_idx)
        fn = lib.bdof_loss_grad_conv if self.conv else lib.bdof_loss_grad
        self.ctx.check(fn(h, self.mb, self.angle_buf.ptr, None, None, self.meas_stage.ptr, None))

    def loss_and_grad(self, angle_idx, want_loss=True):
        """Data-term loss and its gradient w.r.t. the volume for this rank's angles (left in self.g)."""
        lib, h = self.ctx.lib, self.ctx.handle
        self._rot_loss_grad(angle_idx)
        self.ctx.check(lib.bdof_rotation_adjoint(h, self.mb, self.angle_buf.ptr, self.g.ptr, 0, 1.0))
        return self._get_loss() if want_loss else None

    def adam_update(self, i_batch, learning_rate, alpha_d=0.0, alpha_b=0.0, gamma=0.0, clip=True, use_mask=True, slab=None,
                    flip=True):
        new = 1 - self.cur
        self.eng.adam_step(self.x[self.cur], self.x[new], self.g, self.m, self.v, self.mask if use_mask else None,
                           (self.dim_x, self.dim_z, self.dim_y), i_batch, learning_rate, g_scale=1.0 / self.comm.size,
                           alpha_d=alpha_d, alpha_b=alpha_b, gamma=gamma, clip=clip, slab=slab)
        if flip:
            self.cur = new
            self._bind_volume()

    def slab_bounds(self, n_slabs):
        """x-plane ranges [(x0, nx)] of n_slabs nearly equal slabs of the [X][Z][Y] volume."""
        n_slabs = max(1, min(int(n_slabs), self.dim_x))
        edges = [(self.dim_x * i) // n_slabs for i in range(n_slabs + 1)]
        return [(edges[i], edges[i + 1] - edges[i]) for i in range(n_slabs)]

    def _tail(self, i_batch, learning_rate, alpha_d, alpha_b, gamma, n_slabs, flip=True):
        """Rotation adjoint -> all-reduce -> regulariser + Adam, whole volume (n_slabs <= 1) or pipelined over x-slabs."""
        lib, h = self.ctx.lib, self.ctx.handle
        reduce = self.comm.size > 1 or getattr(self.comm, 'always_reduce', False)
        if not reduce or n_slabs <= 1:
            self.ctx.check(lib.bdof_rotation_adjoint(h, self.mb, self.angle_buf.ptr, self.g.ptr, 0, 1.0))
            if reduce:
                self.comm.allreduce_sum_device(self.g, stream_sync=self.ctx.sync)
            self.adam_update(i_batch, learning_rate, alpha_d, alpha_b, gamma, flip=flip)
            return
        slabs = self.slab_bounds(n_slabs)
        per_x = self.dim_z * self.dim_y * 2               # floats per x-plane of the gradient
        bounds = [s[0] * per_x for s in slabs] + [self.dim_x * per_x]

        def produce(c):
            x0, nx = slabs[c]
            self.ctx.check(lib.bdof_rotation_adjoint_rows(h, self.mb, self.angle_buf.ptr, self.g.ptr, x0 * self.dim_z,
                                                          nx * self.dim_z, 0, 1.0))

        def consume(c):
            self.adam_update(i_batch, learning_rate, alpha_d, alpha_b, gamma, slab=slabs[c], flip=False)

        self.comm.pipelined_allreduce(self.g, bounds, produce, consume, stream_ptr=self.eng.stream_ptr())
        if flip:
            self.cur = 1 - self.cur
            self._bind_volume()

    def default_slabs(self):
        if self._n_slabs is None:
            reduce = self.comm.size > 1 or getattr(self.comm, 'always_reduce', False)
            self._n_slabs = int(os.environ.get('BDOF_ALLREDUCE_SLABS', '8')) if reduce else 1
        return self._n_slabs

    def tune_allreduce(self, candidates=(1, 8), reps=2):
        """Pick the number of slabs of the pipelined tail by timing it on this machine and process layout (the collective,
        the streams it runs on and the kernels either side interact in ways that differ between runtimes).  Dry run: the
        gradient buffer and the Adam moments are scratch afterwards (the moments are zeroed again), the volume is not
        touched.  Every rank takes the same decision (max of the timings over ranks).  Call before the epoch loop."""
        import time
        reduce = self.comm.size > 1 or getattr(self.comm, 'always_reduce', False)
        if not reduce or os.environ.get('BDOF_ALLREDUCE_SLABS'):
            return self.default_slabs()
        self.angle_buf.upload(np.arange(self.mb, dtype=np.int32) % self.n_theta)
        times = []
        for n in candidates:
            self._tail(0, 0.0, 0.0, 0.0, 0.0, n, flip=False)             # first use: communicator / stream set-up
            self.ctx.sync()
            self.comm.Barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                self._tail(0, 0.0, 0.0, 0.0, 0.0, n, flip=False)
            self.ctx.sync()
            times.append(time.perf_counter() - t0)
        worst = self.comm.allreduce_max_host(np.array(times))
        self._n_slabs = int(candidates[int(np.argmin(worst))])
        self.tuned = dict(zip([int(c) for c in candidates], [float(t) / reps for t in worst]))
        self.reset_moments()
        return self._n_slabs

    def step(self, i_batch, angle_idx, learning_rate, alpha_d=0.0, alpha_b=0.0, gamma=0.0, want_loss=False, n_slabs=None):
        """grads = loss_grad(...); Allreduce; /size; Adam; mask; clip   (fullfield.py:345-362).

        With more than one rank the tail of the step is pipelined over x-slabs of the volume: rotation adjoint of slab
        c -> all-reduce of slab c (RCCL, asynchronous) -> regulariser + Adam of slab c, so that the 8 B/voxel collective
        overlaps the kernels either side of it (the TV stencil reads the pre-update volume, which no slab overwrites).
        Slab-wise and whole-volume execution give identical results."""
        if n_slabs is None:
            n_slabs = self.default_slabs()
        self._rot_loss_grad(angle_idx)
        self._tail(i_batch, learning_rate, alpha_d, alpha_b, gamma, n_slabs)
        return self._get_loss() if want_loss else None

    def shrink_wrap(self, thresh=1e-15):
        """mask = mask * (obj_delta > 1e-15)   (cnn_propagator/fullfield.py:365-368, intended behaviour, quirk Q8)."""
        if self.mask is not None:
            self.ctx.check(self.ctx.lib.bdof_mask_shrink(self.ctx.handle, self.x[self.cur].ptr, self.mask.ptr, self.nvox, thresh))

    def gradient_to_host(self):
        self.ctx.sync()
        return util.rows_to_volume(self.g.download())

    def forward_angles(self, angle_idx):
        """Detector waves (len(idx), Y, X) of the current volume — the forward_pass of fullfield.py:79-91."""
        idx = np.asarray(angle_idx, dtype=np.int32)
        out = []
        for i in range(0, len(idx), self.mb):
            chunk = idx[i:i + self.mb]
            out.append(self.eng.forward(len(chunk), angle_idx=chunk, conv=self.conv))
        return np.concatenate(out, axis=0)


class PtychoSolver(object):
    """Device-resident state of a ptychographic reconstruction (cnn_propagator/ptychography.py:285-310): volume,
    Adam moments, rotation tables, all diffraction amplitudes; windows are cut by index math inside the kernels."""

    def __init__(self, obj_size, probe_size, probe_pos, n_theta, minibatch_size, energy_ev, psize_cm, probe_real, probe_imag,
                 variant='numpy_skip_last', comm=None, device=0, stream=None, coord_ls=None, propagator='fft', kernel_size=17):
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
        self.eng = MultisliceEngine(self.py, self.px, self.dim_z, self.mb, with_grad=True, device=device, stream=stream, engine=pin)
        self.ctx = self.eng.ctx
        self.eng.set_physics(energy_ev, psize_cm, 'inf', variant=variant)   # free_prop_cm='inf', ptychography.py:76
        if self.conv:
            self.eng.set_conv(energy_ev, psize_cm, kernel_size)
        self.eng.set_probe(probe_real, probe_imag)
        if coord_ls is None:
            coord_ls = util.rotation_lookup([self.dim_y, self.dim_x, self.dim_z], n_theta)
        tab, off, order = util.device_rotation_tables(coord_ls, self.dim_x, self.dim_z)
        self.tab = DeviceBuffer.from_host(self.ctx, tab)
        self.off = DeviceBuffer.from_host(self.ctx, off)
        self.order = DeviceBuffer.from_host(self.ctx, order)
        shape = (self.dim_x, self.dim_z, self.dim_y, 2)
        self.x = [DeviceBuffer.zeros(self.ctx, shape, np.float32), DeviceBuffer.zeros(self.ctx, shape, np.float32)]
        self.cur = 0
        self.g = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.m = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.v = DeviceBuffer.zeros(self.ctx, shape, np.float32)
        self.meas_stage = DeviceBuffer(self.ctx, self.mb * self.py * self.px * 4, np.float32, (self.mb, self.py, self.px))
        self.idx_buf = DeviceBuffer(self.ctx, 3 * self.mb * 4, np.int32, (3, self.mb))
        self._bind_volume()
        self.eng.set_rotation_adjoint(self.off, self.order, self.dim_x * self.dim_z)

    def _bind_volume(self):
        self.eng.set_volume(self.x[self.cur], self.dim_x * self.dim_z, self.dim_y, self.tab, self.dim_x, self.n_theta)

    def set_volume(self, obj_delta, obj_beta):
        self.x[self.cur].upload(util.volume_to_rows(obj_delta, obj_beta))
        self._bind_volume()

    def get_volume(self):
        self.ctx.sync()
        return util.rows_to_volume(self.x[self.cur].download())

    def reset_moments(self):
        lib, h = self.ctx.lib, self.ctx.handle
        self.ctx.check(lib.bdof_memset(h, self.m.ptr, 0, self.m.nbytes))
        self.ctx.check(lib.bdof_memset(h, self.v.ptr, 0, self.v.nbytes))

    def _stage(self, i_theta, pos_idx, prj_abs_batch):
        pos = self.probe_pos[np.asarray(pos_idx)]
        idx = np.empty((3, self.mb), dtype=np.int32)
        idx[0] = i_theta
        idx[1] = pos[:, 1] - self.half[1]            # window origin in x (axis 1), ptychography.py:68-70
        idx[2] = pos[:, 0] - self.half[0]            # window origin in y (axis 0)
        self.idx_buf.upload(idx)
        self.meas_stage.upload(self.eng.meas_layout(prj_abs_batch))
        p = self.idx_buf.ptr
        return p, p + 4 * self.mb, p + 8 * self.mb

    def loss_and_grad(self, i_theta, pos_idx, prj_abs_batch, want_loss=True):
        """prj_abs_batch: |this_prj_batch| (mb, py, px) for probe positions pos_idx at angle i_theta."""
        import ctypes
        lib, h = self.ctx.lib, self.ctx.handle
        a, xo, yo = self._stage(i_theta, pos_idx, prj_abs_batch)
        fn = lib.bdof_loss_grad_conv if self.conv else lib.bdof_loss_grad
        self.ctx.check(fn(h, self.mb, a, xo, yo, self.meas_stage.ptr, None))
        self.ctx.check(lib.bdof_window_rotation_adjoint(h, self.mb, int(i_theta), xo, yo, self.g.ptr, 0, 1.0))
        if want_loss:
            loss = ctypes.c_double(0)
            self.ctx.check(lib.bdof_get_loss(h, ctypes.byref(loss)))
            return loss.value
        return None

    def forward(self, i_theta, pos_idx):
        pos = self.probe_pos[np.asarray(pos_idx)]
        return self.eng.forward(len(pos), angle_idx=[i_theta] * len(pos), xoff=pos[:, 1] - self.half[1],
                                yoff=pos[:, 0] - self.half[0], conv=self.conv)

    def adam_update(self, i_batch, learning_rate, clip=True):
        new = 1 - self.cur
        self.eng.adam_step(self.x[self.cur], self.x[new], self.g, self.m, self.v, None, (self.dim_x, self.dim_z, self.dim_y),
                           i_batch, learning_rate, g_scale=1.0 / self.comm.size, clip=clip)
        self.cur = new
        self._bind_volume()

    def gradient_to_host(self):
        self.ctx.sync()
        return util.rows_to_volume(self.g.download())
