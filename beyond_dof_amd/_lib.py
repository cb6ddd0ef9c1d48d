"""ctypes binding of libbdof.so (C ABI: include/bdof.h).  No fallback: if the HIP library is
missing or no GPU is present, using the product path raises."""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BDOF_LIB') or os.path.join(_HERE, 'libbdof.so')      # BDOF_LIB: a differently built copy (kernel experiments)

DET_NONE, DET_NEAR, DET_FAR = 0, 1, 2
VARIANT_NUMPY_SKIP_LAST, VARIANT_TF_ALL = 0, 1
K_ROW_FWD, K_COL_PROP, K_ROW_BWD, K_LOSS, K_ROT_ADJ, K_ADAM = range(6)
KERNEL_CLASS_NAMES = ['row_fwd', 'col_prop', 'row_bwd', 'loss', 'rot_adjoint', 'adam']

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_float_p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/bdof.h one to one
_SIGNATURES = {
    'bdof_ctx_create': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, _vp]),
    'bdof_ctx_destroy': (None, [_vp]),
    'bdof_last_error': (ctypes.c_char_p, [_vp]),
    'bdof_sync': (ctypes.c_int, [_vp]),
    'bdof_stream': (_vp, [_vp]),
    'bdof_device_count': (ctypes.c_int, []),
    'bdof_twiddle_tables': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'bdof_device_pci_bus_id': (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    'bdof_timer_mark': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bdof_timer_elapsed': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    'bdof_configure': (ctypes.c_int, [_vp] + [ctypes.c_int] * 5),
    'bdof_set_physics': (ctypes.c_int, [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_set_physics_f64': (ctypes.c_int, [_vp, _vp, _vp]),
    'bdof_set_probe': (ctypes.c_int, [_vp, _vp, ctypes.c_double, ctypes.c_double]),
    'bdof_set_meas_mode': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bdof_probe_stack_supported': (ctypes.c_int, [_vp]),
    'bdof_set_probe_field': (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    'bdof_set_probe_stack': (ctypes.c_int, [_vp, _vp, _vp]),
    'bdof_set_object': (ctypes.c_int, [_vp, _vp, ctypes.c_longlong, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_set_rotation_adjoint': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    'bdof_forward': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int]),
    'bdof_forward_range': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    'bdof_tiles_gather': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    'bdof_tiles_scatter': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_adjoint_range': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]),
    'bdof_tiles_scatter_adjoint': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_tiles_gather_adjoint': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    'bdof_tiles_grad_add': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_field_loss_seed': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_set_transfer_f64': (ctypes.c_int, [_vp, _vp]),
    'bdof_forward_range_h': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp]),
    'bdof_fields_free_step': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_caxpy': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, ctypes.c_size_t, ctypes.c_int]),
    'bdof_c_convert': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int]),
    'bdof_tiles_gather_f64': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    'bdof_tiles_scatter_f64': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_tiles_gather_mixed': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    'bdof_tiles_scatter_diff64': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int]),
    'bdof_tiles_scatter_adjoint_mixed': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int,
                                                        ctypes.c_int]),
    'bdof_tiles_gather_adjoint_diff64': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp,
                                                        ctypes.c_int, ctypes.c_int]),
    'bdof_forward_range_f64': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_double, ctypes.c_int]),
    'bdof_tape_to_real': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    'bdof_loss_grad': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    'bdof_set_conv': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int] + [ctypes.c_double] * 5),
    'bdof_set_conv_taps_f64': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, ctypes.c_double]),
    'bdof_set_conv_probe_stack': (ctypes.c_int, [_vp, _vp, _vp] + [ctypes.c_double] * 4),
    'bdof_set_conv_f64': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]),
    'bdof_loss_grad_conv_f64': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_double]),
    'bdof_set_conv_f64_detector': (ctypes.c_int, [_vp, _vp]),
    'bdof_fields_free_step_aux': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_aux_join': (ctypes.c_int, [_vp]),
    'bdof_range_carrier_build': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int]),
    'bdof_set_range_carrier': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'bdof_set_tf_f64': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_double]),
    'bdof_loss_grad_tf_f64': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_double]),
    'bdof_forward_conv': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    'bdof_loss_grad_conv': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    'bdof_get_loss': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double)]),
    'bdof_enable_probe_grad': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bdof_probe_grad': (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    'bdof_grot': (_vp, [_vp]),
    'bdof_rotation_adjoint': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_float]),
    'bdof_rotation_adjoint_rows': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]),
    'bdof_rotate_bilinear': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    'bdof_set_object_bilinear': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int]),
    'bdof_rotate_bilinear_adjoint': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]),
    'bdof_window_rotation_adjoint': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_float]),
    'bdof_adam_step': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                       + [ctypes.c_float] * 8 + [ctypes.c_int, ctypes.c_int]),
    'bdof_adam_step_slab': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                            + [ctypes.c_float] * 8 + [ctypes.c_int] * 4),
    'bdof_regularizer_value': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    'bdof_mask_shrink': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, ctypes.c_float]),
    'bdof_gather_fields': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_size_t]),
    'bdof_set_streams': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bdof_batch_groups': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bdof_profile_enable': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bdof_profile_read': (ctypes.c_int, [_vp, ctypes.c_int, _c_int_p, ctypes.POINTER(ctypes.c_double)]),
    'bdof_malloc': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    'bdof_device_mem': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    'bdof_ctx_malloc': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_size_t]),
    'bdof_comm_unique_id': (ctypes.c_int, [_vp, ctypes.c_size_t]),
    'bdof_comm_create': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t]),
    'bdof_comm_destroy': (None, [_vp]),
    'bdof_comm_last_error': (ctypes.c_char_p, [_vp]),
    'bdof_comm_size': (ctypes.c_int, [_vp]),
    'bdof_comm_rank': (ctypes.c_int, [_vp]),
    'bdof_allreduce_grad': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _c_int_p]),
    'bdof_reduce_scatter_grad': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _c_int_p]),
    'bdof_allgather_volume': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _c_int_p]),
    'bdof_bcast_volume': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int, _c_int_p]),
    'bdof_comm_wait': (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    'bdof_comm_sync': (ctypes.c_int, [_vp]),
    'bdof_free': (ctypes.c_int, [_vp]),
    'bdof_memcpy_h2d': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    'bdof_memcpy_d2h': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    'bdof_memset': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_size_t]),
    'bdof_memcpy_d2d': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
}
EXPORTED_SYMBOLS = sorted(_SIGNATURES)

_lib = None


def build_id():
    """What the measured numbers of a run belong to: a hash of the kernel sources (csrc/*, include/bdof.h), the compiler flags
    libbdof.so was built with (libbdof.so.flags) and a hash of the library file itself.  bench.py puts it in its JSON line and
    the profile summaries under profiles/ carry it (tools/pmc_summary.py, tools/rocpd_summary.py): a traffic figure or a
    rocprofv3 duration read back from a committed file is only quoted next to a run of the same sources and flags."""
    import hashlib
    root = os.path.dirname(_HERE)
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, 'csrc')
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + [os.path.join(root, 'include', 'bdof.h')]
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    flags = open(LIB_PATH + '.flags').read() if os.path.exists(LIB_PATH + '.flags') else None
    lib = hashlib.sha256(open(LIB_PATH, 'rb').read()).hexdigest()[:16] if os.path.exists(LIB_PATH) else None
    return {'source_sha256': h.hexdigest()[:16], 'flags': flags, 'lib_sha256': lib}


class BdofError(RuntimeError):
    pass


TORCH_FIRST = False


def load():
    """Load libbdof.so (build it with `python -c "import __graft_entry__ as g; g.build()"`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BdofError('HIP extension not built: {} is missing (run __graft_entry__.build())'.format(LIB_PATH))
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's, requested under another file name): whichever of
    # torch and libbdof.so is loaded first decides which HIP runtime the process gets, and when libbdof.so comes first
    # torch loads a SECOND runtime that cannot see the GPU.  The product path does not use torch (RCCL sits behind the C
    # ABI); only the gloo rehearsal backend does, and then torch is imported before the library.
    global TORCH_FIRST
    if 'torch' in sys.modules:
        TORCH_FIRST = True
    elif ((int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.environ.get('BDOF_COMM_BACKEND', '').lower() == 'gloo')
          or os.environ.get('BDOF_PRELOAD_TORCH')):
        import torch  # noqa: F401
        TORCH_FIRST = True
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, 'data_ptr'):       # torch tensor on the device
        return x.data_ptr()
    raise TypeError('expected DeviceBuffer / device pointer, got {}'.format(type(x)))


class DeviceBuffer(object):
    """A caller-owned device allocation (hipMalloc through the C ABI)."""

    def __init__(self, ctx, nbytes, dtype=np.uint8, shape=None):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else (self.nbytes // self.dtype.itemsize,)
        p = _vp()
        rc = ctx.lib.bdof_ctx_malloc(ctx.handle, ctypes.byref(p), max(self.nbytes, 1))
        if rc != 0 or not p.value:
            raise BdofError('bdof_ctx_malloc({} bytes) failed with hip error {}'.format(nbytes, rc))
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, arr.nbytes, arr.dtype, arr.shape)
        buf.upload(arr)
        return buf

    @classmethod
    def zeros(cls, ctx, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = cls(ctx, n, dtype, shape)
        ctx.check(ctx.lib.bdof_memset(ctx.handle, buf.ptr, 0, buf.nbytes))
        return buf

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes == self.nbytes, (arr.nbytes, self.nbytes)
        self.ctx.check(self.ctx.lib.bdof_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, shape=None, dtype=None):
        dtype = np.dtype(dtype or self.dtype)
        shape = tuple(shape) if shape is not None else self.shape
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.bdof_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    @property
    def __cuda_array_interface__(self):
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (self.ptr, False), 'version': 2}

    def free(self):
        if getattr(self, 'ptr', None):
            self.ctx.lib.bdof_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context(object):
    """Owns one bdof_ctx (one device, one stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        if self.lib.bdof_device_count() < 1:
            raise BdofError('no HIP device visible: the multislice engine has no CPU fallback')
        h = _vp()
        rc = self.lib.bdof_ctx_create(ctypes.byref(h), int(device), stream)
        if rc != 0:
            raise BdofError('bdof_ctx_create(device={}) failed: {}'.format(device, rc))
        self.handle = h.value
        self.device = device

    def check(self, rc):
        if rc != 0:
            msg = self.lib.bdof_last_error(self.handle)
            raise BdofError('libbdof error {}: {}'.format(rc, msg.decode() if msg else ''))

    def sync(self):
        self.check(self.lib.bdof_sync(self.handle))

    def mem_used(self):
        """Bytes of device memory in use on the ctx's device (all processes)."""
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self.check(self.lib.bdof_device_mem(self.handle, ctypes.byref(free), ctypes.byref(total)))
        return total.value - free.value

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.bdof_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
