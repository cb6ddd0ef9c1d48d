"""Drop-in for cnn_propagator/propagation.py: multislice_propagate_cnn with the same arguments, computed by the HIP
engine's real-space truncated-kernel propagator (separable ks-tap passes instead of a ks x ks convolution)."""
import time

import numpy as np

from .engine import MultisliceEngine

_engine_cache = {}


def multislice_propagate_cnn(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, kernel_size=17,
                             free_prop_cm=None, debug=False):
    """cnn_propagator/propagation.py:18-133.  Returns the detector wave (B, Y, X) complex64; with debug=True also an empty
    per-slice list and the elapsed time, like the reference's (probe, probe_array, time) tuple."""
    assert kernel_size % 2 == 1, 'kernel_size must be an odd number.'
    t0 = time.time()
    B, ny, nx, n_slice = [int(s) for s in np.shape(grid_delta)]
    key = (ny, nx, n_slice)
    eng = _engine_cache.get(key)
    if eng is None or eng.batch_max < B:
        eng = MultisliceEngine(ny, nx, n_slice, B, with_grad=False)
        _engine_cache[key] = eng
    eng.set_physics(energy_ev, psize_cm, free_prop_cm)
    eng.set_conv(energy_ev, psize_cm, kernel_size)
    eng.set_probe(probe_real, probe_imag)
    eng.set_object_batch(np.asarray(grid_delta), np.asarray(grid_beta))
    wave = eng.forward(B, conv=True)
    if debug:
        return wave, [], time.time() - t0
    return wave
