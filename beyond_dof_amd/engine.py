"""Python host of the multislice engine: owns a libbdof context and converts between the reference's
array conventions ((B, Y, X, S) objects, (B, Y, X) waves, (Y, X, Z) volumes) and the device layout.
All arithmetic of the hot path runs in libbdof.so; there is no CPU fallback."""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import DeviceBuffer
from . import util

_DET = {None: _lib.DET_NONE, 'inf': _lib.DET_FAR}
_VARIANT = {'numpy_skip_last': _lib.VARIANT_NUMPY_SKIP_LAST, 'tf_all': _lib.VARIANT_TF_ALL}


def _idx_buf(ctx, values):
    return DeviceBuffer.from_host(ctx, np.asarray(values, dtype=np.int32))


RESIDENT_SIZES = (32, 36, 48, 64, 72, 80, 96, 128)      # square fields with an LDS-resident plan (csrc/bdof_resident.h)


class MultisliceEngine(object):
    """One wavefield geometry (NY x NX x S) on one GPU."""

    def __init__(self, ny, nx, n_slice, batch_max, with_grad=True, device=0, stream=None, force_generic=False, engine='auto',
                 recompute=None, no_grot=False, adjoint64=None):
        """Engines (include/bdof.h, bdof_configure): powers of two in 64..1024 run on the fused streaming kernels; small
        square fields (32..128, e.g. the 72 x 72 ptychography probe) on the LDS-resident kernel when there is no fused plan
        or the batch is large; every other size on the generic engine (rocFFT).  engine='generic' (= force_generic=True),
        'streaming' (never resident) or 'resident' (resident for every batch size) pin the choice for cross-checks.
        adjoint64=True (env BDOF_ADJOINT64=1): the adjoint sweep in float64 (bdof_configure flag 64; generic engine) — the
        accuracy option for reconstructions that must follow the reference's float64 loop voxel by voxel."""
        if engine not in ('auto', 'generic', 'streaming', 'resident'):
            raise ValueError('engine must be auto, generic, streaming or resident')
        force_generic = force_generic or engine == 'generic'
        if recompute is None:
            recompute = bool(int(os.environ.get('BDOF_RECOMPUTE', '0')))
        self.recompute = bool(recompute)
        if adjoint64 is None:
            adjoint64 = bool(int(os.environ.get('BDOF_ADJOINT64', '0')))
        self.adjoint64 = bool(adjoint64) and bool(with_grad)
        self.ctx = _lib.Context(device, stream)
        self.lib = self.ctx.lib
        self.h = self.ctx.handle
        self.ny, self.nx, self.n_slice, self.batch_max = int(ny), int(nx), int(n_slice), int(batch_max)
        self.with_grad = bool(with_grad)
        self.ctx.check(self.lib.bdof_configure(self.h, self.ny, self.nx, self.n_slice, self.batch_max,
                                               int(bool(with_grad)) | (2 if (force_generic or os.environ.get('BDOF_FORCE_GENERIC')) else 0)
                                               | (4 if engine == 'streaming' else 0) | (8 if engine == 'resident' else 0)
                                               | (16 if self.recompute else 0) | (32 if no_grot else 0) | (64 if self.adjoint64 else 0)))
        self._engine_arg = 'generic' if force_generic else engine
        self._device = device
        self.det_mode = _lib.DET_NONE
        self._keep = {}          # device buffers that must outlive the calls that registered them
        self._tables = None

    # ---- physics -------------------------------------------------------------------------------
    def set_physics(self, energy_ev, psize_cm, free_prop_cm=None, variant='numpy_skip_last', pi=util.PI, field_shape=None,
                    detector_kernel='TF'):
        """k and H exactly as cnn_propagator/np_funcs.py:19-32,45-57 derive them from energy / pixel size.  field_shape: this
        engine's wavefields are tiles of a (FY, FX) field and apply that field's propagator (util.get_kernel_tile).
        detector_kernel: 'TF' (what np_funcs.py:55 forces), 'IR' (get_kernel_ir, np_funcs.py:59-61) or 'auto' (the sampling
        criterion of np_funcs.py:51-53) for the step to a detector at a finite distance."""
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        delta_nm = voxel_nm[-1]
        k = 2. * pi * delta_nm / lmbda_nm
        hs64 = util.device_transfer_function(delta_nm, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, field_shape=field_shape, dtype=np.complex128)
        hs = hs64.astype(np.complex64)
        h00 = np.array(util.transfer_function_dc(delta_nm, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, field_shape=field_shape))
        hdet = hdet00 = None
        self.det_kernel = 'TF'
        if free_prop_cm is None:
            det = _lib.DET_NONE
        elif isinstance(free_prop_cm, str):
            if free_prop_cm != 'inf':
                raise ValueError("free_prop_cm must be None, a distance in cm or 'inf'")
            det = _lib.DET_FAR
        else:
            det = _lib.DET_NEAR
            self.det_kernel = util.detector_kernel_kind(detector_kernel, free_prop_cm * 1e7, lmbda_nm, voxel_nm, (self.ny, self.nx))
            hdet = util.device_transfer_function(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, kernel=self.det_kernel)
            hdet00 = np.array(util.transfer_function_dc(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, kernel=self.det_kernel))
        self.det_mode = det
        self.variant = variant
        self.k = k
        self._physics_args = (energy_ev, psize_cm, free_prop_cm, variant, pi)
        self._field_shape = field_shape
        self._tf64_args = (hs64, free_prop_cm, lmbda_nm, voxel_nm, pi, float(k))
        self.tf_f64 = self.conv_f64 = False     # a float64 twin bound before this call held the previous tables
        # tf_all + far field: the last transfer-function step only multiplies the far field by the
        # unit-modulus H (F P phi = H . F phi); libbdof skips it and the host applies it to returned waves
        self._far_phase = None
        if det == _lib.DET_FAR and variant == 'tf_all':
            self._far_phase = (hs.astype(np.complex128) * (self.nx * self.ny)).astype(np.complex64)
        self.ctx.check(self.lib.bdof_set_physics(self.h, k, hs.ctypes.data, hdet.ctypes.data if hdet is not None else None,
                                                 h00.ctypes.data, hdet00.ctypes.data if hdet00 is not None else None,
                                                 det, _VARIANT[variant]))
        # the same table in float64: the streaming kernels multiply by dithered float32 copies of it (bdof_set_transfer_f64)
        self.ctx.check(self.lib.bdof_set_transfer_f64(self.h, hs64.ctypes.data))
        if self.adjoint64:
            hd64 = None
            if det == _lib.DET_NEAR:
                hd64 = util.device_transfer_function(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, dtype=np.complex128,
                                                     kernel=self.det_kernel)
            self.ctx.check(self.lib.bdof_set_physics_f64(self.h, hs64.ctypes.data, hd64.ctypes.data if hd64 is not None else None))
        if getattr(self, '_probe_args', None) is not None:
            self.set_probe(*self._probe_args)      # the carrier (field, calibration) of the probe depends on the physics

    def _probe_stack(self, probe_c64):
        """The probe propagated through free space to the entrance of every slice and to the detector, in float64 on the
        host (np_funcs.py:42-61 without an object) — the carrier field of bdof_set_probe_stack (include/bdof.h)."""
        energy_ev, psize_cm, free_prop_cm, variant, pi = self._physics_args
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        h = np.fft.ifftshift(util.get_kernel(voxel_nm[-1], lmbda_nm, voxel_nm, (self.ny, self.nx), pi=pi))
        p = probe_c64.astype(np.complex128)
        stack = np.empty((self.n_slice, self.nx, self.ny), dtype=np.complex64)
        for z in range(self.n_slice):
            stack[z] = p.T
            if z < self.n_slice - 1:
                p = np.fft.ifft2(np.fft.fft2(p) * h)
        if self.det_mode == _lib.DET_FAR:
            det = np.fft.fft2(p)                      # un-shifted, un-normalised; a tf_all step before it is applied on the host
        else:
            if variant == 'tf_all':
                p = np.fft.ifft2(np.fft.fft2(p) * h)
            if self.det_mode == _lib.DET_NEAR:
                hd = np.fft.ifftshift(util.centred_kernel(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, kernel=self.det_kernel))
                p = np.fft.ifft2(np.fft.fft2(p) * hd)
            det = p
        return np.ascontiguousarray(stack), np.ascontiguousarray(det.T.astype(np.complex64))

    def _probe_field_device(self, probe_c64):
        """The same carrier field computed by the library on the device in float64 (bdof_set_probe_field): the host only forms
        the two transfer functions (float64, transposed to [kx][ky])."""
        energy_ev, psize_cm, free_prop_cm, variant, pi = self._physics_args
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        fs = getattr(self, '_field_shape', None)
        kern = (lambda d: util.get_kernel_tile(d, lmbda_nm, voxel_nm, (self.ny, self.nx), fs, pi=pi)) if fs is not None else \
            (lambda d: util.get_kernel(d, lmbda_nm, voxel_nm, (self.ny, self.nx), pi=pi))
        hT = np.ascontiguousarray(np.fft.ifftshift(kern(voxel_nm[-1])).T.astype(np.complex128))
        hdT = None
        if self.det_mode == _lib.DET_NEAR:
            kd = kern(free_prop_cm * 1e7) if self.det_kernel == 'TF' else \
                util.centred_kernel(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, kernel='IR')
            hdT = np.ascontiguousarray(np.fft.ifftshift(kd).T.astype(np.complex128))
        p = np.ascontiguousarray(probe_c64.T.astype(np.complex128))
        self.ctx.check(self.lib.bdof_set_probe_field(self.h, p.ctypes.data, hT.ctypes.data, hdT.ctypes.data if hdT is not None else None))

    def set_probe(self, probe_real, probe_imag):
        self._probe_args = (np.array(probe_real, copy=True), np.array(probe_imag, copy=True))
        self.tf_f64 = self.conv_f64 = False     # the float64 twins (enable_tf_f64 / enable_conv_f64) hold the previous probe
        probe = (np.asarray(probe_real) + 1j * np.asarray(probe_imag)) * np.ones((self.ny, self.nx))
        probe = probe.astype(np.complex64)                         # the reference rounds to complex64 too (np_funcs.py:20)
        # Carrier splitting: the wave is held as carrier + eps and only eps runs through the float32 transforms.
        #  - a (nearly) uniform probe rides on its mean a0, propagated exactly as a scalar inside the library;
        #  - a localised probe rides on its own free-space propagation, a carrier FIELD per slice, which the library computes
        #    on the device in float64 (bdof_set_probe_field): eps is the scattered wave alone;
        #  - otherwise (BDOF_NO_PROBE_STACK, the real-space propagator) a0 = 0 and the whole wave is float32.
        mean = complex(probe.astype(np.complex128).mean())
        a0 = mean if np.abs(probe - mean).max() <= 0.25 * abs(mean) else 0j
        self.probe_stack = False
        if getattr(self, '_conv_set', False) and hasattr(self, '_physics_args'):
            if self._set_conv_probe_stack(probe, a0):
                return
        use_stack = (a0 == 0 and hasattr(self, '_physics_args') and not os.environ.get('BDOF_NO_PROBE_STACK')
                     and not getattr(self, '_conv_set', False)              # the real-space propagator has its own carrier
                     and self.n_slice * self.nx * self.ny <= (1 << 32)      # 32 GiB of stack at most
                     and self.lib.bdof_probe_stack_supported(self.h) == 1)
        if use_stack:
            if os.environ.get('BDOF_HOST_PROBE_STACK'):                     # cross-check: the float64 propagation on the host
                stack, det = self._probe_stack(probe)
                zero = np.zeros((self.nx, self.ny), dtype=np.complex64)
                self.ctx.check(self.lib.bdof_set_probe(self.h, zero.ctypes.data, 0.0, 0.0))
                self.ctx.check(self.lib.bdof_set_probe_stack(self.h, stack.ctypes.data, det.ctypes.data))
            else:
                self._probe_field_device(probe)
            self.probe_stack, self.probe_gain = True, 1.0
            self._set_meas_mode(0j)
            return
        if self.lib.bdof_probe_stack_supported(self.h) == 1:
            self.ctx.check(self.lib.bdof_set_probe_stack(self.h, None, None))
        self.probe_gain = 1.0
        eps = np.ascontiguousarray((probe.astype(np.complex128) - a0).T.astype(np.complex64))
        self.ctx.check(self.lib.bdof_set_probe(self.h, eps.ctypes.data, a0.real, a0.imag))
        self._set_meas_mode(a0)

    def _set_conv_probe_stack(self, probe, a0):
        """Real-space propagator with a probe that has no dominant constant part (a0 == 0, e.g. a ptychography probe): the
        carrier FIELD of bdof_set_conv_probe_stack — the probe carried through empty space by the padded convolution itself,
        in float64 on the host — so that only the scattered wave runs through the float32 convolutions and the residual
        |d| - m is taken in float64.  Returns True if the stack was set (False: scalar carrier, stack removed)."""
        lib, h = self.lib, self.h
        small = (self.n_slice + 1) * self.nx * self.ny <= (1 << 28)
        if a0 != 0 or not small or os.environ.get('BDOF_NO_PROBE_STACK'):
            self.ctx.check(lib.bdof_set_conv_probe_stack(h, None, None, 0., 0., 0., 0.))
            return False
        energy_ev, psize_cm, free_prop_cm, variant, pi = self._physics_args
        ky, kx, e = self._conv_kernel
        planes = util.conv_probe_stack(probe.astype(np.complex128), ky, kx, e, self.n_slice)          # (S + 1, Y, X)
        p_end = planes[-1]
        if self.det_mode == _lib.DET_FAR:
            det = np.fft.fft2(p_end)                                                                   # [ky][kx], un-shifted
        elif self.det_mode == _lib.DET_NEAR:
            voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
            hd = np.fft.ifftshift(util.centred_kernel(free_prop_cm * 1e7, 1240. / energy_ev, voxel_nm, self.ny, self.nx, pi=pi, kernel=self.det_kernel))
            det = np.fft.ifft2(np.fft.fft2(p_end) * hd).T                                              # [x][y]
        else:
            det = p_end.T
        stack = np.ascontiguousarray(planes.transpose(0, 2, 1).astype(np.complex64))                   # [S + 1][x][y]
        det = np.ascontiguousarray(det.astype(np.complex128))
        zero = np.zeros((self.nx, self.ny), dtype=np.complex64)
        if lib.bdof_probe_stack_supported(h) == 1:
            self.ctx.check(lib.bdof_set_probe_stack(h, None, None))
        self.ctx.check(lib.bdof_set_probe(h, zero.ctypes.data, 0.0, 0.0))
        p0, ps = complex(planes[0][0, 0]), complex(p_end[0, 0])
        self.ctx.check(lib.bdof_set_conv_probe_stack(h, stack.ctypes.data, det.ctypes.data, p0.real, p0.imag, ps.real, ps.imag))
        self.probe_stack, self.probe_gain = False, 1.0
        self._set_meas_mode(0j)
        return True

    # ---- gradient w.r.t. the probe (probe_type='optimizable', tensorflow_recon/fullfield.py:311-327) -------------------
    def enable_probe_grad(self, on=True):
        self.ctx.check(self.lib.bdof_enable_probe_grad(self.h, int(bool(on))))
        self._gprobe = DeviceBuffer.zeros(self.ctx, (self.nx, self.ny), np.complex64) if on else None

    def probe_grad(self, accumulate=False, to_host=True):
        """dL/d(probe_real) + i dL/d(probe_imag) of the last loss_grad, summed over its wavefields: (Y, X) complex."""
        self.ctx.check(self.lib.bdof_probe_grad(self.h, self._gprobe.ptr, int(bool(accumulate))))
        if not to_host:
            return self._gprobe
        self.ctx.sync()
        return np.ascontiguousarray(self._gprobe.download().T)

    def set_probe_none(self):
        """No probe of the ctx's own: every wavefield starts from a caller-supplied field (bdof_forward_range), no carrier."""
        self._probe_args = None
        zero = np.zeros((self.nx, self.ny), dtype=np.complex64)
        self.ctx.check(self.lib.bdof_set_probe(self.h, zero.ctypes.data, 0.0, 0.0))
        self.ctx.check(self.lib.bdof_set_probe_stack(self.h, None, None))
        self.probe_stack, self.probe_gain = False, 1.0
        self._set_meas_mode(0j)

    residual_split = True      # False: amplitudes go to the device as they are (a probe that changes between steps)

    def _set_meas_mode(self, a0):
        """Residual splitting at the detector (include/bdof.h, bdof_set_meas_mode): with a plane-wave carrier and a real-space
        detector the measured amplitudes go to the device as m - |a0|.  The amplitudes a caller keeps resident on the device
        are laid out for the reference in force when they were uploaded (meas_layout), so a probe that is re-set between
        steps (probe_type='optimizable') must not move it: the solvers switch the splitting off for that case
        (residual_split = False) before they upload."""
        self.meas_ref = 0.0
        if a0 != 0 and self.det_mode != _lib.DET_FAR and self.residual_split and not os.environ.get('BDOF_NO_RESIDUAL_SPLIT'):
            self.meas_ref = abs(a0)
        self.ctx.check(self.lib.bdof_set_meas_mode(self.h, 1 if self.meas_ref else 0))

    def set_conv(self, energy_ev, psize_cm, kernel_size=17):
        """Switch the slice-to-slice step to the truncated real-space kernel of multislice_propagate_cnn
        (cnn_propagator/propagation.py:18-44): k uses numpy's pi there (:25), the kernel the reference's PI literal."""
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        delta_nm = voxel_nm[-1]
        ky, kx, e = util.conv_kernel_separable(delta_nm, lmbda_nm, voxel_nm, (self.ny, self.nx), kernel_size)
        ksum = e * ky.sum() * kx.sum()
        self._conv_set = True
        self._conv_kernel = (ky, kx, e)
        k = 2. * np.pi * delta_nm / lmbda_nm
        kyf = np.ascontiguousarray(ky.astype(np.complex64))
        kxf = np.ascontiguousarray(kx.astype(np.complex64))
        self.ctx.check(self.lib.bdof_set_conv(self.h, kyf.ctypes.data, kxf.ctypes.data, int(kernel_size), e.real, e.imag,
                                              ksum.real, ksum.imag, k))
        # the taps in float64 as well: dithered copies, one per slice (include/bdof.h)
        ky64 = np.ascontiguousarray(ky.astype(np.complex128))
        kx64 = np.ascontiguousarray(kx.astype(np.complex128))
        self.ctx.check(self.lib.bdof_set_conv_taps_f64(self.h, ky64.ctypes.data, kx64.ctypes.data, float(e.real), float(e.imag)))
        self._conv_k64 = k
        if getattr(self, '_probe_args', None) is not None:
            self.set_probe(*self._probe_args)      # the carrier (scalar or field) of the probe follows the propagator

    def enable_tf_f64(self):
        """The transfer-function model's loss + gradient entirely in float64 on this context (bdof_loss_grad_tf_f64;
        loss_grad(..., f64=True)): what the reference's autograd differentiates (np_funcs.py:15-65 in numpy float64).  The
        accuracy path of the first minibatch of an epoch (adjoint_precision='first-step') — no second engine — and a float64 twin
        of the fused kernels for tests.  Hands the probe and the transfer function(s) over in float64; call again after
        set_physics / set_probe."""
        if getattr(self, '_tf64_args', None) is None or getattr(self, '_probe_args', None) is None:
            raise RuntimeError('set_physics and set_probe first')
        hs64, free_prop_cm, lmbda_nm, voxel_nm, pi, k = self._tf64_args
        hd = None
        if self.det_mode == _lib.DET_NEAR:
            hd64 = util.device_transfer_function(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi, dtype=np.complex128,
                                                 kernel=self.det_kernel)
            hd = np.ascontiguousarray(hd64.T)
        ht = np.ascontiguousarray(hs64.T)                                       # [kx][ky]
        pr, pi_ = self._probe_args
        # the reference's wavefront starts as complex64 (np_funcs.py:20-21) and becomes complex128 at the first product
        probe = np.ascontiguousarray(((np.asarray(pr) + 1j * np.asarray(pi_)) * np.ones((self.ny, self.nx))).T.astype(np.complex64)
                                     .astype(np.complex128))
        self.ctx.check(self.lib.bdof_set_tf_f64(self.h, probe.ctypes.data, ht.ctypes.data, hd.ctypes.data if hd is not None else None, k))
        self.tf_f64 = True

    def enable_conv_f64(self):
        """The real-space propagator's loss + gradient entirely in float64 (bdof_loss_grad_conv_f64; loss_grad(..., conv=True,
        f64=True)): the accuracy path for the first minibatch of an epoch (adjoint_precision='first-step' / 'float64' with
        propagator='conv').  Square fields.  Hands the probe and the transform of the zero-padded
        ks x ks kernel over in float64 (overlap-save on the padded (N + ks - 1)^2 grid)."""
        if not getattr(self, '_conv_set', False) or getattr(self, '_probe_args', None) is None:
            raise RuntimeError('set_conv and set_probe first')
        if self.nx != self.ny:
            raise ValueError('the float64 real-space path takes square wavefields')
        ky, kx, e = self._conv_kernel
        ks = len(ky)
        m = self.nx + ks - 1
        kpad = np.zeros((m, m), dtype=np.complex128)
        kpad[:ks, :ks] = e * np.outer(ky, kx)                                    # K[p][q] = e ky[p] kx[q]   (numpy (Y, X) order)
        khat = np.ascontiguousarray((np.fft.fft2(kpad) / float(m * m)).T)       # [kx][ky]
        ksum = e * ky.sum() * kx.sum()
        pr, pi = self._probe_args
        probe = np.ascontiguousarray(((np.asarray(pr) + 1j * np.asarray(pi)) * np.ones((self.ny, self.nx))).T.astype(np.complex128))
        self.ctx.check(self.lib.bdof_set_conv_f64(self.h, probe.ctypes.data, khat.ctypes.data, ks, float(ksum.real), float(ksum.imag),
                                                  float(self._conv_k64)))
        hd = None
        if self.det_mode == _lib.DET_NEAR:                                       # propagation.py:122-124: one transfer-function step
            _, free_prop_cm, lmbda_nm, voxel_nm, pi64, _ = self._tf64_args
            hd = np.ascontiguousarray(util.device_transfer_function(free_prop_cm * 1e7, lmbda_nm, voxel_nm, self.ny, self.nx, pi=pi64,
                                                                    dtype=np.complex128, kernel=self.det_kernel).T)
        self.ctx.check(self.lib.bdof_set_conv_f64_detector(self.h, hd.ctypes.data if hd is not None else None))
        self.conv_f64 = True

    # ---- object --------------------------------------------------------------------------------
    def set_object_batch(self, grid_delta_batch, grid_beta_batch):
        """Already rotated objects, (B, Y, X, S) each (the np_funcs.py:15 argument convention)."""
        rows = util.batch_to_rows(grid_delta_batch, grid_beta_batch)
        buf = DeviceBuffer.from_host(self.ctx, rows)
        self._keep['obj'] = buf
        self.ctx.check(self.lib.bdof_set_object(self.h, buf.ptr, rows.shape[0] * rows.shape[1] * rows.shape[2], self.ny, None, 0, 0))
        return buf

    def set_volume(self, vol_buf, n_rows, vol_ny, tab_buf, vol_nx, n_angles):
        """Un-rotated volume rows [n_rows = X*Z][vol_ny] pairs + rotation table [n_angles][S][vol_nx] (device).
        Must be called again after the volume memory changed (the modulation table is rebuilt)."""
        self._keep['obj'] = vol_buf
        self._keep['tab'] = tab_buf
        self.ctx.check(self.lib.bdof_set_object(self.h, _lib._ptr(vol_buf), int(n_rows), int(vol_ny), _lib._ptr(tab_buf), int(vol_nx), int(n_angles)))

    def set_rotation_adjoint(self, off_buf, order_buf, n_dest):
        self._keep['off'] = off_buf
        self._keep['order'] = order_buf
        self.ctx.check(self.lib.bdof_set_rotation_adjoint(self.h, _lib._ptr(off_buf), _lib._ptr(order_buf), int(n_dest)))

    # ---- forward -------------------------------------------------------------------------------
    def _wave_to_host(self, buf, B):
        if self.det_mode == _lib.DET_FAR:
            w = buf.download((B, self.ny, self.nx), np.complex64)        # un-shifted fft2, [b][ky][kx]
            if self._far_phase is not None:
                w = w * self._far_phase
            return np.ascontiguousarray(np.fft.fftshift(w, axes=(1, 2)))   # np_funcs.py:48
        return np.ascontiguousarray(buf.download((B, self.nx, self.ny), np.complex64).transpose(0, 2, 1))

    def meas_layout(self, meas_abs):
        """|measured| (n, Y, X) in the index order libbdof's loss kernels read it."""
        ref = getattr(self, 'meas_ref', 0.0)
        if ref:
            m = (np.asarray(meas_abs, dtype=np.float64) - ref).astype(np.float32)      # subtract in float64, round once
        else:
            m = np.asarray(meas_abs, dtype=np.float32)
        if self.det_mode == _lib.DET_FAR:
            return np.ascontiguousarray(np.fft.ifftshift(m, axes=(1, 2)))
        return np.ascontiguousarray(m.transpose(0, 2, 1))

    def _meas_to_device(self, meas_abs):
        return DeviceBuffer.from_host(self.ctx, self.meas_layout(meas_abs))

    def forward(self, B, angle_idx=None, xoff=None, yoff=None, keep_tape=False, to_host=True, conv=False):
        out = DeviceBuffer(self.ctx, B * self.nx * self.ny * 8, np.complex64, (B, self.nx, self.ny))
        a = _idx_buf(self.ctx, angle_idx) if angle_idx is not None else None
        xo = _idx_buf(self.ctx, xoff) if xoff is not None else None
        yo = _idx_buf(self.ctx, yoff) if yoff is not None else None
        if conv:
            self.ctx.check(self.lib.bdof_forward_conv(self.h, B, _lib._ptr(a), _lib._ptr(xo), _lib._ptr(yo), out.ptr))
        else:
            self.ctx.check(self.lib.bdof_forward(self.h, B, _lib._ptr(a), _lib._ptr(xo), _lib._ptr(yo), out.ptr, int(keep_tape)))
        self.ctx.sync()
        return self._wave_to_host(out, B) if to_host else out

    def probe_array(self, B):
        """Per-slice wavefields, (S, B, Y, X) — the second return value of np_funcs.py:65."""
        out = DeviceBuffer(self.ctx, B * self.nx * self.ny * 8, np.complex64, (B, self.nx, self.ny))
        res = np.empty((self.n_slice, B, self.ny, self.nx), dtype=np.complex64)
        for i in range(self.n_slice):
            self.ctx.check(self.lib.bdof_tape_to_real(self.h, i, B, out.ptr))
            res[i] = out.download((B, self.nx, self.ny), np.complex64).transpose(0, 2, 1)
        return res

    # ---- loss + gradient -----------------------------------------------------------------------
    def loss_grad(self, B, meas_abs, angle_idx=None, xoff=None, yoff=None, meas_on_device=False, conv=False, f64=False):
        """Runs forward + loss + adjoint; returns the loss.  The gradient stays on the device.  f64: the float64 path of the
        model on the same context (enable_tf_f64 / with conv: enable_conv_f64)."""
        m = meas_abs if meas_on_device else self._meas_to_device(meas_abs)
        a = _idx_buf(self.ctx, angle_idx) if angle_idx is not None else None
        xo = _idx_buf(self.ctx, xoff) if xoff is not None else None
        yo = _idx_buf(self.ctx, yoff) if yoff is not None else None
        if f64:
            if conv and not getattr(self, 'conv_f64', False):
                raise RuntimeError('the real-space propagator\'s float64 path: enable_conv_f64() first')
            if not conv and not getattr(self, 'tf_f64', False):
                raise RuntimeError('the transfer-function model\'s float64 path: enable_tf_f64() first')
            fn = self.lib.bdof_loss_grad_conv_f64 if conv else self.lib.bdof_loss_grad_tf_f64
            self.ctx.check(fn(self.h, B, _lib._ptr(a), _lib._ptr(xo), _lib._ptr(yo), _lib._ptr(m), float(getattr(self, 'meas_ref', 0.0))))
        else:
            fn = self.lib.bdof_loss_grad_conv if conv else self.lib.bdof_loss_grad
            self.ctx.check(fn(self.h, B, _lib._ptr(a), _lib._ptr(xo), _lib._ptr(yo), _lib._ptr(m), None))
        loss = ctypes.c_double(0)
        self.ctx.check(self.lib.bdof_get_loss(self.h, ctypes.byref(loss)))
        self._keep['last_idx'] = (a, xo, yo, m)
        return loss.value

    def grad_batch_to_host(self, B):
        """Gradient w.r.t. the rotated object batch: (g_delta, g_beta), each (B, Y, X, S)."""
        n = B * self.n_slice * self.nx * self.ny
        out = np.empty((B, self.n_slice, self.nx, self.ny, 2), dtype=np.float32)
        self.ctx.check(self.lib.bdof_memcpy_d2h(self.h, out.ctypes.data, self.lib.bdof_grot(self.h), n * 8))
        return util.rows_to_batch(out)

    def rotation_adjoint(self, B, angle_idx, gvol_buf, accumulate=False, scale=1.0):
        a = _idx_buf(self.ctx, angle_idx)
        self.ctx.check(self.lib.bdof_rotation_adjoint(self.h, B, a.ptr, _lib._ptr(gvol_buf), int(accumulate), float(scale)))
        self.ctx.sync()

    def adam_step(self, x_old, x_new, g, m, v, mask, shape_xzy, i_batch, lr, g_scale=1.0, alpha_d=0.0, alpha_b=0.0,
                  gamma=0.0, b1=0.9, b2=0.999, eps=1e-8, clip=True, slab=None):
        """slab = (x0, nx): update only that range of x-planes (pipelined with the gradient all-reduce)."""
        nxv, nzv, nyv = [int(s) for s in shape_xzy]
        x0, nx = (0, nxv) if slab is None else (int(slab[0]), int(slab[1]))
        self.ctx.check(self.lib.bdof_adam_step_slab(self.h, _lib._ptr(x_old), _lib._ptr(x_new), _lib._ptr(g), _lib._ptr(m),
                                                    _lib._ptr(v), _lib._ptr(mask), nxv, nzv, nyv, g_scale, alpha_d, alpha_b,
                                                    gamma, lr, b1, b2, eps, int(i_batch), int(clip), x0, nx))

    def stream_ptr(self):
        return int(self.lib.bdof_stream(self.h) or 0)

    # ---- profiling -----------------------------------------------------------------------------
    def set_streams(self, n=-1):
        """Number of concurrent sub-batches of the fused FFT engine (-1: automatic, see include/bdof.h)."""
        self.ctx.check(self.lib.bdof_set_streams(self.h, int(n)))

    def batch_groups(self, B):
        return int(self.lib.bdof_batch_groups(self.h, int(B)))

    def profile_enable(self, on=True, stride=1):
        self.ctx.check(self.lib.bdof_profile_enable(self.h, int(stride) if on else 0))

    def profile_read(self):
        res = {}
        for cls, name in enumerate(_lib.KERNEL_CLASS_NAMES):
            n = ctypes.c_int(0)
            ms = ctypes.c_double(0)
            self.ctx.check(self.lib.bdof_profile_read(self.h, cls, ctypes.byref(n), ctypes.byref(ms)))
            res[name] = (n.value, ms.value)
        return res

    def sync(self):
        self.ctx.sync()
