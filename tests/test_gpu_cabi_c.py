"""The drop-in boundary is a C ABI: examples/cabi_forward.c — plain C, compiled with gcc against include/bdof.h, no HIP or
Python in it — drives libbdof.so through forward + loss + gradient; its outputs must equal what the Python host gets
from the same library (bit for bit) and agree with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize('fp', [None, 1e-4])
def test_c_host_drives_the_library(tmp_path, fp, monkeypatch):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import util
    from beyond_dof_amd.engine import MultisliceEngine
    exe = str(tmp_path / 'cabi_forward')
    subprocess.check_call(['gcc', '-O2', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'cabi_forward.c'), '-o', exe,
                           '-L', os.path.join(ROOT, 'beyond_dof_amd'), '-lbdof',
                           '-Wl,-rpath,' + os.path.join(ROOT, 'beyond_dof_amd')])
    rng = np.random.default_rng(3)
    B, Y, X, S = 2, 128, 64, 6
    delta = rng.uniform(0, 2e-5, size=(B, Y, X, S))
    beta = 0.1 * delta
    pr, pi = 1 + 0.1 * rng.normal(size=(Y, X)), 0.1 * rng.normal(size=(Y, X))
    ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, return_probe_array=False)
    meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))

    # what the host hands over (the same recipe as beyond_dof_amd/engine.py: float64 on the host, rounded once)
    voxel_nm = np.array([1e-7] * 3) * 1e7
    lmbda_nm = 1240. / 5000.
    k = 2. * util.PI * voxel_nm[-1] / lmbda_nm
    hs = util.device_transfer_function(voxel_nm[-1], lmbda_nm, voxel_nm, Y, X)
    h00 = util.transfer_function_dc(voxel_nm[-1], lmbda_nm, voxel_nm, Y, X)
    det, hdet, hdet00 = 0, None, (1.0, 0.0)
    if fp is not None:
        det = 1
        hdet = util.device_transfer_function(fp * 1e7, lmbda_nm, voxel_nm, Y, X)
        hdet00 = util.transfer_function_dc(fp * 1e7, lmbda_nm, voxel_nm, Y, X)
    probe = (pr + 1j * pi).astype(np.complex64)
    a0 = 0j                                                           # no carrier splitting: the probe is not near-uniform
    eps = np.ascontiguousarray(probe.T)
    rows = util.batch_to_rows(delta, beta)
    meas_dev = np.ascontiguousarray(meas.astype(np.float32).transpose(0, 2, 1))     # [b][x][y]
    with open(str(tmp_path / 'in.bin'), 'wb') as f:
        f.write(struct.pack('<6i', Y, X, S, B, det, 0))
        f.write(struct.pack('<7d', k, h00[0], h00[1], hdet00[0], hdet00[1], a0.real, a0.imag))
        f.write(hs.tobytes())
        f.write(util.device_transfer_function(voxel_nm[-1], lmbda_nm, voxel_nm, Y, X, dtype=np.complex128).tobytes())
        if hdet is not None:
            f.write(hdet.tobytes())
        f.write(eps.tobytes())
        f.write(rows.tobytes())
        f.write(meas_dev.tobytes())
    r = subprocess.run([exe, str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    raw = open(str(tmp_path / 'out.bin'), 'rb').read()
    loss_c = struct.unpack('<d', raw[:8])[0]
    nw = B * X * Y * 8
    wave_c = np.frombuffer(raw[8:8 + nw], dtype=np.complex64).reshape(B, X, Y).transpose(0, 2, 1)
    grad_c = util.rows_to_batch(np.frombuffer(raw[8 + nw:], dtype=np.float32).reshape(B, S, X, Y, 2))

    # the Python host on the same library
    # the C driver hands the probe over as it is: no carrier field, no energy calibration (engine.py: set_probe)
    monkeypatch.setenv('BDOF_NO_PROBE_STACK', '1')
    eng = MultisliceEngine(Y, X, S, B, with_grad=True)
    eng.calibrate_energy = False
    eng.set_physics(5000., 1e-7, fp)
    eng.set_probe(pr, pi)
    eng.set_object_batch(delta, beta)
    wave_py = eng.forward(B)
    loss_py = eng.loss_grad(B, meas)
    gd_py, gb_py = eng.grad_batch_to_host(B)
    assert np.array_equal(wave_c, wave_py) and loss_c == loss_py
    assert np.array_equal(grad_c[0], gd_py) and np.array_equal(grad_c[1], gb_py)
    # and the oracle
    rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp)
    assert rel(wave_c, ref) <= 5e-6 and abs(loss_c - rl) <= 1e-5 * abs(rl)
    assert rel(grad_c[0], rgd) <= 2e-4 and rel(grad_c[1], rgb) <= 2e-4
