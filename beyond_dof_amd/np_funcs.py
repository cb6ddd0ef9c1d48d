"""Drop-in for cnn_propagator/np_funcs.py: same function name, arguments and return values, computed by
the HIP engine (libbdof.so) instead of numpy FFTs."""
import numpy as np

from .engine import MultisliceEngine
from . import util

PI = util.PI

_engine_cache = {}


def _engine(ny, nx, n_slice, batch, with_grad):
    key = (ny, nx, n_slice, with_grad)
    eng = _engine_cache.get(key)
    if eng is None or eng.batch_max < batch:
        eng = MultisliceEngine(ny, nx, n_slice, batch, with_grad=with_grad)
        _engine_cache[key] = eng
    return eng


def multislice_propagate_batch_numpy(grid_delta_batch, grid_beta_batch, probe_real, probe_imag, energy_ev, psize_cm,
                                     free_prop_cm=None, obj_batch_shape=None, variant='numpy_skip_last',
                                     return_probe_array=True):
    """cnn_propagator/np_funcs.py:15-65.  Returns (wavefront[B,Y,X] complex64, probe_array[S,B,Y,X]).

    `variant` and `return_probe_array` are extensions: 'tf_all' propagates after the last slice too
    (tensorflow_recon/util.py:465-483); return_probe_array=False skips the per-slice history."""
    if obj_batch_shape is None:
        obj_batch_shape = grid_delta_batch.shape
    B, ny, nx, n_slice = [int(s) for s in obj_batch_shape]
    want_tape = bool(return_probe_array) and variant == 'numpy_skip_last'
    eng = _engine(ny, nx, n_slice, B, with_grad=want_tape)
    eng.set_physics(energy_ev, psize_cm, free_prop_cm, variant=variant)
    eng.set_probe(probe_real, probe_imag)
    eng.set_object_batch(np.asarray(grid_delta_batch)[:B], np.asarray(grid_beta_batch)[:B])
    wave = eng.forward(B, keep_tape=want_tape)
    probe_array = eng.probe_array(B) if want_tape else np.zeros((0,), dtype=np.complex64)
    return wave, probe_array
