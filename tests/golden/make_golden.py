"""Generate golden vectors by running the REFERENCE's own functions.

Run only in the build container (needs /root/reference, which never travels):

    python tests/golden/make_golden.py            # G1-G6  -> tests/golden/*.npz
    /opt/conda/bin/python3.9 tests/golden/make_golden.py --h5   # G7 (h5py 3.3.0 lives there)
    python tests/golden/make_golden.py --g8       # G8 (get_kernel_ir, upsample_2x)
    python tests/golden/make_golden.py --g9       # G9 (propagation.multislice_propagate_cnn, the real-space propagator)
    python tests/golden/make_golden.py --g10      # G10 (fullfield.reconstruct_fullfield: the whole loop, ~1 min)
    python tests/golden/make_golden.py --g11      # G11 (ptychography.reconstruct_ptychography: the whole loop, ~4 min)
    python tests/golden/make_golden.py --g12      # G12 (simulation.create_fullfield_data_numpy / create_ptychography_data_batch_numpy)
    python tests/golden/make_golden.py --g13      # G13 (reconstruct_fullfield at a size the GPU's real-space kernels take)
    python tests/golden/make_golden.py --g14      # G14 (reconstruct_ptychography likewise: 64 x 64 probe, 64^3 object)
    python tests/golden/make_golden.py --g15      # G15 (reconstruct_fullfield's loop around np_funcs' FFT forward model)
    python tests/golden/make_golden.py --g16      # G16 (tensorflow_recon/create_noisy_data.py: Poisson noise, ptychography branch)
    python tests/golden/make_golden.py --g17      # G17 (reconstruct_ptychography's loop around np_funcs' FFT forward model)
    python tests/golden/make_golden.py --g18      # G18 (G15 at BASELINE config 2's size, 256^3; ~2 min)
    python tests/golden/make_golden.py --g19      # G19 (G18 with 2 % amplitude noise on the data, as measured data have)
    python tests/golden/make_golden.py --g21 fft; python tests/golden/make_golden.py --g21 conv      # G21 (G20 for ptychography)
    python tests/golden/make_golden.py --g20      # G20 (directional derivatives of the reference's calculate_loss at 64^3, both forward models)

The reference modules are imported unmodified; only third-party imports that
the hot path never touches (dxchange, h5py, tensorflow, matplotlib backends) are
stubbed, pyfftw's numpy interface is aliased to numpy.fft (the reference's own
fallback, tensorflow_recon/util.py:7-14) and autograd.numpy to numpy.
G9 runs propagation.multislice_propagate_cnn unmodified.  Its one third-party primitive, HIPS autograd's
autograd.scipy.signal.convolve (not installed, no pinned version anywhere in the reference), is stood in by
scipy.signal.convolve2d per batch element for the one call form the function uses (mode='valid',
axes=([1, 2], [0, 1]): true convolution of every [b] image with the 2-D kernel).  The vector therefore pins everything
the reference itself writes — kernel construction and crop, the running padding constant, the slice loop, the corner
renormalisation, the detector step — up to the definition of that primitive.
G10 runs fullfield.reconstruct_fullfield unmodified on an 8^3 problem: h5py / dxchange are in-memory stand-ins for the
file traffic, and autograd.grad — reverse-mode differentiation by HIPS autograd, absent — is stood in by float64 central
finite differences of the reference's own calculate_loss (1024 variables, relative accuracy ~1e-8).  The vector pins the
loop itself: minibatch schedule from the global numpy seed, per-epoch Adam restart, mask, clip, regulariser variants
(incl. quirk Q6), the order of all of it — with the reference's own forward model, rotation and Adam underneath.
G11 does the same for ptychography.reconstruct_ptychography (10^3 object, 18 x 18 gaussian probe, 3 positions x 2 angles,
minibatches of 2): additionally mpi4py's COMM_WORLD is a one-rank stand-in and the module's clock is frozen, because the
function seeds numpy from int(time.time() / 60) right before it draws the epoch's schedule.
G13 runs reconstruct_fullfield once more at (Y, X, Z) = (64, 64, 64) with 17 taps — a size the HIP real-space kernels
accept, so the product's entry point can be compared with the reference's loop DIRECTLY — where 524288 finite-difference
variables are out of reach: there autograd.grad is stood in by the oracle's analytic gradient of the same loss (itself pinned
by finite differences, torch autograd and, through G10, by this very loop at 8^3).
G15 is G13 with one substitution inside the reference: the name multislice_propagate_cnn that fullfield.py calls is bound to
the reference's OTHER forward model, np_funcs.multislice_propagate_batch_numpy (the transfer-function propagator that the
north star names) — two pieces of reference code composed, so that the product's default entry point has a reference-run
loop to be compared with.
The fixtures are data only (inputs + outputs); no reference source is stored.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/cnn_propagator'


def _import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub('dxchange')
    stub('h5py')
    stub('tensorflow')
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    pyfftw = stub('pyfftw')
    interfaces = stub('pyfftw.interfaces')
    nfft = stub('pyfftw.interfaces.numpy_fft', fft2=np.fft.fft2, ifft2=np.fft.ifft2, fftn=np.fft.fftn,
                ifftn=np.fft.ifftn, fftshift=np.fft.fftshift, ifftshift=np.fft.ifftshift)
    pyfftw.interfaces = interfaces
    interfaces.numpy_fft = nfft
    autograd = stub('autograd')
    sys.modules['autograd.numpy'] = np
    autograd.numpy = np
    if not hasattr(np, 'int'):
        np.int = int                      # cnn_propagator/util.py:326-327 uses the removed alias
    sys.path.insert(0, REF)
    import util as ref_util
    import np_funcs as ref_np_funcs
    return ref_util, ref_np_funcs


def main():
    ref_util, ref_np = _import_reference()
    out = {}

    # G1: get_kernel
    g1 = {}
    for (Y, X) in [(8, 8), (9, 12), (64, 64)]:
        for dist in [1.0, 1000.0]:
            g1['H_{}_{}_{}'.format(Y, X, int(dist))] = ref_util.get_kernel(dist, 0.248, [1., 1., 1.], [Y, X, 4])
    np.savez_compressed(os.path.join(HERE, 'g1_get_kernel.npz'), **g1)

    # G2a: forward on a 16x16x6 random object, three detector modes, B=2
    rng = np.random.default_rng(0)
    delta = rng.uniform(0, 1e-5, size=(2, 16, 16, 6))
    beta = 0.1 * delta
    pr = np.ones((16, 16))
    pi_ = np.zeros((16, 16))
    g2 = {'delta': delta, 'beta': beta}
    for name, fp in [('none', None), ('near', 1e-4), ('inf', 'inf')]:
        w, pa = ref_np.multislice_propagate_batch_numpy(delta, beta, pr, pi_, 5000., 1e-7, free_prop_cm=fp,
                                                        obj_batch_shape=delta.shape)
        g2['wave_' + name] = w
        if name == 'none':
            g2['probe_array'] = pa
    # non-trivial probe
    prr = rng.normal(size=(16, 16))
    pii = rng.normal(size=(16, 16))
    w, _ = ref_np.multislice_propagate_batch_numpy(delta, beta, prr, pii, 5000., 1e-7, free_prop_cm=1e-4,
                                                   obj_batch_shape=delta.shape)
    g2['probe_real'] = prr
    g2['probe_imag'] = pii
    g2['wave_near_probe'] = w
    np.savez_compressed(os.path.join(HERE, 'g2_forward_16.npz'), **g2)

    # G2b: cfg1 — 64^3 tube phantom (tensorflow_recon/grid_delta.npy), first 32 slices, beta := 0.1 delta
    gd = np.load('/root/reference/tensorflow_recon/grid_delta.npy')[..., :32]
    gd = gd.reshape(1, *gd.shape)
    gb = 0.1 * gd
    g2b = {'delta': gd.astype(np.float64)}
    for name, fp in [('none', None), ('near', 1e-4), ('inf', 'inf')]:
        w, pa = ref_np.multislice_propagate_batch_numpy(gd, gb, np.ones((64, 64)), np.zeros((64, 64)), 5000., 1e-7,
                                                        free_prop_cm=fp, obj_batch_shape=gd.shape)
        g2b['wave_' + name] = w
        if name == 'none':
            g2b['probe_array_abs_sum'] = np.abs(pa).sum(axis=(1, 2, 3))
            g2b['probe_array_last'] = pa[-1]
    np.savez_compressed(os.path.join(HERE, 'g2_forward_cfg1.npz'), **g2b)

    # G3: rotation tables + apply_rotation
    import tempfile
    g3 = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for size, n_theta in [([8, 8, 8], 5), ([64, 64, 64], 4), ([6, 10, 10], 7)]:
                folder = 'arrsize_{}_{}_{}_ntheta_{}'.format(*size, n_theta)
                ref_util.save_rotation_lookup(size, n_theta)
                coords = ref_util.read_all_origin_coords(folder, n_theta)
                key = 'x'.join(map(str, size)) + '_n{}'.format(n_theta)
                g3['coords_' + key] = np.stack(coords)
                if size[0] <= 8:
                    rngo = np.random.default_rng(3)
                    obj = rngo.normal(size=(size[0], size[1], size[2], 2))
                    g3['obj_' + key] = obj
                    g3['rot_' + key] = np.stack([ref_util.apply_rotation(obj, c, folder) for c in coords])
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, 'g3_rotation.npz'), **g3)

    # G4: three Adam steps incl. the m = v = None start.
    # NB util.py:285 does np.zeros_like(v) with v=None -> 0-d object array; arithmetic still works.
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1e-6, size=(2, 4, 5, 6))
    g4 = {'x0': x}
    m = v = None
    for it in range(3):
        g = rng.normal(size=x.shape) * 1e-3
        x, m, v = ref_util.apply_gradient_adam(x, g, it, m, v, step_size=1e-7)
        g4['g{}'.format(it)] = g
        g4['x{}'.format(it + 1)] = np.asarray(x, dtype=np.float64)
        g4['m{}'.format(it + 1)] = np.asarray(m, dtype=np.float64)
        g4['v{}'.format(it + 1)] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'g4_adam.npz'), **g4)

    # G5: total_variation_3d
    rng = np.random.default_rng(5)
    arr = rng.normal(size=(6, 7, 8))
    np.savez_compressed(os.path.join(HERE, 'g5_tv.npz'), arr=arr, tv=ref_util.total_variation_3d(arr))

    # G6: split_tasks
    arr = np.arange(23)
    parts = ref_util.split_tasks(arr, 5)
    np.savez_compressed(os.path.join(HERE, 'g6_split_tasks.npz'), arr=arr, split_size=5,
                        lengths=np.array([len(p) for p in parts]), concat=np.concatenate(parts))
    print('golden vectors written to', HERE)


def main_g8():
    """G8: the two helpers off the hot path that the entry points can still reach (kernel='IR', multiscale)."""
    ref_util, _ = _import_reference()
    g8 = {}
    for (Y, X) in [(8, 8), (9, 12), (32, 32)]:
        for dist in [50.0, 1000.0]:
            g8['Hir_{}_{}_{}'.format(Y, X, int(dist))] = ref_util.get_kernel_ir(dist, 0.248, [1., 1., 1.], [Y, X, 4])
    rng = np.random.default_rng(8)
    a3 = rng.normal(size=(3, 4, 5))
    a4 = rng.normal(size=(2, 3, 2, 2))
    g8['up_in3'], g8['up_out3'] = a3, ref_util.upsample_2x(a3)
    g8['up_in4'], g8['up_out4'] = a4, ref_util.upsample_2x(a4)
    np.savez_compressed(os.path.join(HERE, 'g8_kernel_ir_upsample.npz'), **g8)
    print('wrote g8')


def main_g9():
    """G9: the real-space truncated-kernel propagator (cnn_propagator/propagation.py:18-133)."""
    _import_reference()
    from scipy.signal import convolve2d

    def convolve(a, b, mode='full', axes=None):
        if mode != 'valid' or axes != ([1, 2], [0, 1]):
            raise NotImplementedError('stand-in covers the call form of propagation.py:93 only')
        return np.stack([convolve2d(img, b, mode='valid') for img in a])

    autograd = sys.modules['autograd']
    autograd.grad = lambda *a, **k: None
    sys.modules['autograd.numpy.random'] = np.random
    ascipy = types.ModuleType('autograd.scipy')
    asignal = types.ModuleType('autograd.scipy.signal')
    asignal.convolve = convolve
    ascipy.signal = asignal
    autograd.scipy = ascipy
    sys.modules['autograd.scipy'] = ascipy
    sys.modules['autograd.scipy.signal'] = asignal
    import propagation as ref_prop
    ref_prop.trange = range                                   # no progress bar in the log
    g9 = {}
    rng = np.random.default_rng(9)
    delta = rng.uniform(0, 1e-5, size=(2, 16, 20, 6))
    beta = 0.1 * delta
    g9['delta16'], g9['beta16'] = delta, beta
    prr, pii = 1 + 0.1 * rng.normal(size=(16, 20)), 0.1 * rng.normal(size=(16, 20))
    g9['probe_real16'], g9['probe_imag16'] = prr, pii
    for name, fp in [('none', None), ('near', 1e-4), ('inf', 'inf')]:
        g9['wave16_k5_' + name] = ref_prop.multislice_propagate_cnn(delta, beta, np.ones((16, 20)), np.zeros((16, 20)), 5000., [1e-7] * 3,
                                                                    kernel_size=5, free_prop_cm=fp)
    g9['wave16_k9_probe_near'] = ref_prop.multislice_propagate_cnn(delta, beta, prr, pii, 5000., [1e-7] * 3, kernel_size=9, free_prop_cm=1e-4)
    # cfg1: 64^3 tube phantom, first 32 slices, beta := 0.1 delta, the entry points' default kernel_size = 17
    gd = np.load('/root/reference/tensorflow_recon/grid_delta.npy')[..., :32]
    gd = gd.reshape(1, *gd.shape).astype(np.float64)
    for name, fp in [('none', None), ('near', 1e-4)]:
        g9['wave_cfg1_k17_' + name] = ref_prop.multislice_propagate_cnn(gd, 0.1 * gd, np.ones((64, 64)), np.zeros((64, 64)), 5000., [1e-7] * 3,
                                                                        kernel_size=17, free_prop_cm=fp)
    np.savez_compressed(os.path.join(HERE, 'g9_conv_propagator.npz'), **g9)
    print('wrote g9')


def _setup_conv_reference():
    """Import propagation.py with the stand-in of G9 for autograd's convolve; returns (stubs dict, module)."""
    _import_reference()
    from scipy.signal import convolve2d

    def convolve(a, b, mode='full', axes=None):
        if mode != 'valid' or axes != ([1, 2], [0, 1]):
            raise NotImplementedError('stand-in covers the call form of propagation.py:93 only')
        return np.stack([convolve2d(img, b, mode='valid') for img in a])

    autograd = sys.modules['autograd']
    sys.modules['autograd.numpy.random'] = np.random
    ascipy = types.ModuleType('autograd.scipy')
    asignal = types.ModuleType('autograd.scipy.signal')
    asignal.convolve = convolve
    ascipy.signal = asignal
    autograd.scipy = ascipy
    sys.modules['autograd.scipy'] = ascipy
    sys.modules['autograd.scipy.signal'] = asignal
    return autograd


def main_g10():
    """G10: reconstruct_fullfield (cnn_propagator/fullfield.py:19-390), the whole optimisation loop, on 8^3."""
    import contextlib
    import io
    import tempfile
    autograd = _setup_conv_reference()

    def fd_grad(fn, argnums):
        assert list(argnums) == [0, 1]

        def g(obj_delta, obj_beta, *rest):
            out = []
            with contextlib.redirect_stdout(io.StringIO()):           # calculate_loss prints the loss on every call
                for which in (0, 1):
                    args = [np.array(obj_delta, dtype=np.float64), np.array(obj_beta, dtype=np.float64)]
                    gr = np.zeros_like(args[which])
                    h = 1e-9 if which == 0 else 1e-10
                    it = np.nditer(args[which], flags=['multi_index'])
                    for _ in it:
                        i = it.multi_index
                        keep = args[which][i]
                        args[which][i] = keep + h
                        lp = fn(args[0], args[1], *rest)
                        args[which][i] = keep - h
                        lm = fn(args[0], args[1], *rest)
                        args[which][i] = keep
                        gr[i] = (lp - lm) / (2 * h)
                    out.append(gr)
            first_call.append((np.array(obj_delta), np.array(obj_beta), np.array(rest[0]), out[0].copy(), out[1].copy()))
            return tuple(out)
        return g

    first_call = []
    autograd.grad = fd_grad
    import propagation as ref_prop
    ref_prop.trange = range
    import fullfield as ref_ff
    ref_ff.trange = range

    n, n_theta, mb = 8, 4, 2
    rng = np.random.default_rng(10)
    true_d = np.zeros((n, n, n))
    true_d[2:6, 2:6, 2:6] = rng.uniform(2e-6, 6e-6, size=(4, 4, 4))
    true_b = 0.1 * true_d
    mask = np.ones((n, n, n), dtype=np.float32)
    mask[0] = 0
    init_d = np.clip(rng.normal(3e-6, 1e-6, size=(n, n, n)), 0, None)
    init_b = np.clip(rng.normal(3e-7, 1e-7, size=(n, n, n)), 0, None)
    store = {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, key):
            return self.arr[key]

    class _File(object):
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, key):
            assert key == 'exchange/data'
            return _Dataset(store['prj'])

    sys.modules['h5py'].File = _File
    ref_ff.h5py.File = _File
    dx = sys.modules['dxchange']
    written = {}
    dx.read_tiff_stack = lambda fname, ind, digit=5: np.array(mask)
    dx.read_tiff = lambda fname: np.array(mask)
    dx.write_tiff = lambda arr, fname=None, dtype=None, overwrite=False: written.__setitem__(os.path.basename(fname), np.array(arr))
    dx.write_tiff_stack = lambda *a, **k: None

    g10 = {'true_delta': true_d, 'true_beta': true_b, 'mask': mask, 'init_delta': init_d, 'init_beta': init_b}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            # the data: the reference's own forward model on its own rotation tables
            ref_ff.save_rotation_lookup([n, n, n], n_theta)
            folder = 'arrsize_{0}_{0}_{0}_ntheta_{1}'.format(n, n_theta)
            coords = ref_ff.read_all_origin_coords(folder, n_theta)
            obj = np.stack([true_d, true_b], axis=3)
            rot = np.stack([ref_ff.apply_rotation(obj, c, folder) for c in coords])
            prj = ref_prop.multislice_propagate_cnn(rot[..., 0], rot[..., 1], np.ones((n, n)), np.zeros((n, n)), 5000., [1e-7] * 3,
                                                    kernel_size=5, free_prop_cm=1e-4)
            store['prj'] = prj.astype('complex64')
            g10['prj'] = store['prj']
            cases = {'a': dict(alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11, n_epochs=2, seed=7),
                     'b': dict(alpha=1e-8, alpha_d=None, alpha_b=None, gamma=0., n_epochs=1, seed=3),      # quirk Q6 branch
                     # shrink-wrap from the first epoch on (with shrink_cycle > i_epoch the reference dies on an unbound name, :369-372)
                     'c': dict(alpha_d=4e-6, alpha_b=1.5e-9, gamma=0., n_epochs=2, seed=11, shrink_cycle=0)}
            for tag, kw in cases.items():
                np.random.seed(kw.pop('seed'))                        # the schedule is shuffled from the global state (:196-197)
                written.clear()
                with contextlib.redirect_stdout(io.StringIO()):
                    ref_ff.reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, learning_rate=1e-7, minibatch_size=mb,
                                                 energy_ev=5000, psize_cm=1e-7, free_prop_cm=1e-4, save_path='.', output_folder='out',
                                                 initial_guess=[init_d.copy(), init_b.copy()], kernel_size=5, **dict(dict(shrink_cycle=None), **kw))
                if tag == 'a':            # the gradient of the first minibatch: finite differences of the reference's calculate_loss
                    g10['grad0_delta_in'], g10['grad0_beta_in'], g10['grad0_ind'], g10['grad0_gd'], g10['grad0_gb'] = first_call[0]
                g10['delta_' + tag] = np.asarray(written['delta_ds_1'], dtype=np.float64)
                g10['beta_' + tag] = np.asarray(written['beta_ds_1'], dtype=np.float64)
                print('case', tag, 'done: |delta - init| =', np.abs(g10['delta_' + tag] - init_d * mask).max())
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, 'g10_reconstruct_fullfield.npz'), **g10)
    print('wrote g10')


def _fd_grad(fn, argnums):
    """Stand-in for autograd.grad(fn, [0, 1]): float64 central differences, one variable at a time."""
    import contextlib
    import io
    assert list(argnums) == [0, 1]

    def g(obj_delta, obj_beta, *rest):
        out = []
        with contextlib.redirect_stdout(io.StringIO()):
            for which in (0, 1):
                args = [np.array(obj_delta, dtype=np.float64), np.array(obj_beta, dtype=np.float64)]
                gr = np.zeros_like(args[which])
                h = 1e-9 if which == 0 else 1e-10
                it = np.nditer(args[which], flags=['multi_index'])
                for _ in it:
                    i = it.multi_index
                    keep = args[which][i]
                    args[which][i] = keep + h
                    lp = fn(args[0], args[1], *rest)
                    args[which][i] = keep - h
                    lm = fn(args[0], args[1], *rest)
                    args[which][i] = keep
                    gr[i] = (lp - lm) / (2 * h)
                out.append(gr)
        return tuple(out)
    return g


def main_g11():
    """G11: reconstruct_ptychography (cnn_propagator/ptychography.py:19-365), the whole loop, on a 10^3 object."""
    import contextlib
    import io
    import tempfile
    autograd = _setup_conv_reference()
    autograd.grad = _fd_grad

    class _Comm(object):
        def Get_size(self):
            return 1

        def Get_rank(self):
            return 0

        def Barrier(self):
            pass

        def Allreduce(self, src, dst):
            dst[...] = src

    mpi4py = types.ModuleType('mpi4py')
    mpi4py.MPI = types.SimpleNamespace(COMM_WORLD=_Comm())
    sys.modules['mpi4py'] = mpi4py
    import propagation as ref_prop
    ref_prop.trange = range
    import ptychography as ref_pt
    ref_pt.trange = range
    frozen = types.SimpleNamespace(time=lambda: 42 * 60.0 + 1.0)      # seed = int(time.time() / 60) = 42  (:165-166)
    ref_pt.time = frozen

    n, n_theta, psz, mb = 10, 2, (18, 18), 2
    pos = [(3, 4), (6, 5), (5, 7)]                                    # 3 positions, minibatches of 2: the padding quirk is exercised
    rng = np.random.default_rng(11)
    true_d = np.zeros((n, n, n))
    true_d[2:8, 2:8, 2:8] = rng.uniform(2e-6, 6e-6, size=(6, 6, 6))
    true_b = 0.1 * true_d
    init_d = np.clip(rng.normal(3e-6, 1e-6, size=(n, n, n)), 0, None)
    init_b = np.clip(rng.normal(3e-7, 1e-7, size=(n, n, n)), 0, None)
    kw = dict(probe_mag_sigma=4., probe_phase_sigma=4., probe_phase_max=0.5)
    store, written = {}, {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr
            self.shape = arr.shape

        def __getitem__(self, key):
            return self.arr[key]

    class _File(object):
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, key):
            assert key == 'exchange/data'
            return _Dataset(store['prj'])

    ref_pt.h5py.File = _File
    dx = sys.modules['dxchange']
    dx.write_tiff = lambda arr, fname=None, dtype=None, overwrite=False: written.__setitem__(os.path.basename(fname), np.array(arr))

    g11 = {'true_delta': true_d, 'true_beta': true_b, 'init_delta': init_d, 'init_beta': init_b, 'probe_pos': np.array(pos),
           'probe_size': np.array(psz)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            # the data: the reference's own window cut + forward model, evaluated through its calculate_loss ingredients
            ref_pt.save_rotation_lookup([n, n, n], n_theta)
            folder = 'arrsize_{0}_{0}_{0}_ntheta_{1}'.format(n, n_theta)
            coords = ref_pt.read_all_origin_coords(folder, n_theta)
            py = np.arange(psz[0]) - (psz[0] - 1.) / 2
            px = np.arange(psz[1]) - (psz[1] - 1.) / 2
            pxx, pyy = np.meshgrid(px, py)
            pr, pi_ = ref_pt.mag_phase_to_real_imag(np.exp(-(pxx ** 2 + pyy ** 2) / (2 * 4. ** 2)), 0.5 * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * 4. ** 2)))
            half = (np.array(psz) / 2).astype('int')
            prj = np.zeros((n_theta, len(pos), psz[0], psz[1]), dtype='complex64')
            for t in range(n_theta):
                rot = ref_pt.apply_rotation(np.stack([true_d, true_b], axis=3), coords[t], folder)
                rot = np.pad(rot, ((half[0], half[0]), (half[1], half[1]), (0, 0), (0, 0)), mode='constant')
                subs = np.stack([rot[p[0]:p[0] + psz[0], p[1]:p[1] + psz[1]] for p in pos])     # window starts at pos - half in the un-padded frame
                prj[t] = ref_prop.multislice_propagate_cnn(subs[..., 0], subs[..., 1], pr, pi_, 5000., [1e-7] * 3, free_prop_cm='inf')
            store['prj'] = prj
            g11['prj'] = prj
            with contextlib.redirect_stdout(io.StringIO()):
                ref_pt.reconstruct_ptychography('data.h5', [tuple(p) for p in pos], psz, (n, n, n), theta_st=0, theta_end=2 * np.pi, n_epochs=2,
                                                learning_rate=2e-7, minibatch_size=mb, energy_ev=5000, psize_cm=1e-7, save_path='.',
                                                output_folder='out', initial_guess=[init_d.copy(), init_b.copy()], probe_type='gaussian',
                                                dynamic_dropping=False, **kw)
            g11['delta'] = np.asarray(written['delta_ds_1'], dtype=np.float64)
            g11['beta'] = np.asarray(written['beta_ds_1'], dtype=np.float64)
            print('done: |delta - init| max =', np.abs(g11['delta'] - init_d).max())
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, 'g11_reconstruct_ptychography.npz'), **g11)
    print('wrote g11')


def main_g12():
    """G12: the dataset simulators (simulation.py:80-161,283-386).  cnn_propagator/simulation.py imports a module `npfuncs`
    that only tensorflow_recon/ holds; the two simulation.py are identical in these functions, so the TF twin's directory —
    simulation.py, util.py, npfuncs.py side by side — is the one imported (tensorflow and tensorflow.contrib.image are empty
    stand-ins: the numpy simulators never touch them; h5py is an in-memory stand-in that keeps what is written)."""
    import contextlib
    import io
    import tempfile

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    written = {}

    class _Dataset(object):
        def __init__(self, shape, dtype):
            self.arr = np.zeros(shape, dtype=dtype)

        def __setitem__(self, key, val):
            self.arr[key] = val

    class _Group(object):
        def __init__(self, fname):
            self.fname = fname

        def create_dataset(self, name, shape=None, dtype=None):
            d = _Dataset(shape, dtype)
            written[os.path.basename(self.fname)] = d
            return d

    class _File(object):
        def __init__(self, fname, mode='r'):
            self.fname = fname

        def create_group(self, name):
            return _Group(self.fname)

        def close(self):
            pass

    stub('dxchange', write_tiff=lambda *a, **k: None)
    stub('h5py', File=_File)
    tf = stub('tensorflow')
    contrib = stub('tensorflow.contrib')
    image = stub('tensorflow.contrib.image', rotate=lambda *a, **k: None)
    tf.contrib, contrib.image = contrib, image
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    if not hasattr(np, 'int'):
        np.int = int
    sys.path.insert(0, '/root/reference/tensorflow_recon')
    import simulation as sim
    sim.tqdm = lambda x: x

    n = 16
    rng = np.random.default_rng(12)
    z, y, x = np.mgrid[:n, :n, :n].astype(np.float64)
    d = np.zeros((n, n, n))
    for _ in range(4):
        c = rng.uniform(5, 11, size=3)
        r = rng.uniform(1.5, 3.0)
        d += 4e-6 * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * r ** 2))
    g12 = {'grid_delta': d, 'grid_beta': 0.1 * d}
    pos = [(4, 4), (8, 9), (12, 6), (15, 15)]
    g12['probe_pos'] = np.array(pos)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            os.makedirs('phantom')
            np.save('phantom/grid_delta.npy', d)
            np.save('phantom/grid_beta.npy', 0.1 * d)
            with contextlib.redirect_stdout(io.StringIO()):
                sim.create_fullfield_data_numpy(5000., 1e-7, 1e-4, 3, 'phantom', '.', 'ff_plane.h5', batch_size=2, probe_type='plane',
                                                theta_st=0, theta_end=2 * np.pi)
                sim.create_fullfield_data_numpy(5000., 1e-7, None, 2, 'phantom', '.', 'ff_gauss.h5', batch_size=1, probe_type='gaussian',
                                                theta_st=0, theta_end=np.pi, probe_mag_sigma=4., probe_phase_sigma=4., probe_phase_max=0.5)
                sim.create_ptychography_data_batch_numpy(5000., 1e-7, 2, 'phantom', '.', 'pt.h5', pos, probe_type='gaussian', probe_size=(8, 8),
                                                         theta_st=0, theta_end=2 * np.pi, probe_circ_mask=None, minibatch_size=3,
                                                         probe_mag_sigma=2., probe_phase_sigma=2., probe_phase_max=0.5)
        finally:
            os.chdir(cwd)
    for k, v in written.items():
        g12[k.replace('.h5', '')] = v.arr
        print(k, v.arr.shape, v.arr.dtype, np.abs(v.arr).max())
    np.savez_compressed(os.path.join(HERE, 'g12_simulators.npz'), **g12)
    print('wrote g12')


def main_g13():
    """G13: reconstruct_fullfield at (64, 64, 64), kernel_size 17; gradient by the oracle's analytic adjoint.  Initial guess and
    mask are formulas (g13_inputs.py); the reconstructed volumes are stored on every second voxel per axis, as float32."""
    import contextlib
    import io
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import bdof_oracle as orc
    autograd = _setup_conv_reference()

    def oracle_grad(fn, argnums):
        assert list(argnums) == [0, 1]
        cl = dict(zip(fn.__code__.co_freevars, [c.cell_contents for c in fn.__closure__]))

        def g(obj_delta, obj_beta, this_ind_batch, this_prj_batch):
            coords = cl['coord_ls']
            rot = np.stack([orc.apply_rotation(np.stack([obj_delta, obj_beta], axis=3), coords[j]) for j in this_ind_batch])
            _, gd_rot, gb_rot = orc.cnn_loss_and_grad(rot[..., 0], rot[..., 1], cl['probe_real'], cl['probe_imag'], cl['energy_ev'],
                                                      [cl['psize_cm'] * cl['ds_level']] * 3, np.abs(this_prj_batch),
                                                      kernel_size=cl['kernel_size'], free_prop_cm=cl['free_prop_cm'])
            gd = sum(orc.apply_rotation_adjoint(gd_rot[b], coords[j]) for b, j in enumerate(this_ind_batch))
            gb = sum(orc.apply_rotation_adjoint(gb_rot[b], coords[j]) for b, j in enumerate(this_ind_batch))
            rd, rb = orc.regularizer_grad(obj_delta, obj_beta, alpha=cl['alpha'], alpha_d=cl['alpha_d'], alpha_b=cl['alpha_b'], gamma=cl['gamma'])
            return gd + rd, gb + rb
        return g

    autograd.grad = oracle_grad
    import propagation as ref_prop
    ref_prop.trange = range
    import fullfield as ref_ff
    ref_ff.trange = range

    sys.path.insert(0, HERE)
    import g13_inputs
    ny, nx, n_theta, mb = 64, 64, 4, 2
    rng = np.random.default_rng(13)
    yy, xx, zz = np.mgrid[:ny, :nx, :nx].astype(np.float64)
    true_d = np.zeros((ny, nx, nx))
    for _ in range(5):
        c = (rng.uniform(16, 48), rng.uniform(20, 44), rng.uniform(20, 44))
        r = rng.uniform(4, 8)
        true_d += 3e-6 * np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2 + (zz - c[2]) ** 2) / (2 * r ** 2))
    true_b = 0.1 * true_d
    mask = g13_inputs.mask()
    init_d, init_b = g13_inputs.initial_guess()
    store, written = {}, {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, key):
            return self.arr[key]

    class _File(object):
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, key):
            return _Dataset(store['prj'])

    ref_ff.h5py.File = _File
    dx = sys.modules['dxchange']
    dx.read_tiff_stack = lambda fname, ind, digit=5: np.array(mask)
    dx.read_tiff = lambda fname: np.array(mask)
    dx.write_tiff = lambda arr, fname=None, dtype=None, overwrite=False: written.__setitem__(os.path.basename(fname), np.array(arr))
    dx.write_tiff_stack = lambda *a, **k: None
    g13 = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            ref_ff.save_rotation_lookup([ny, nx, nx], n_theta)
            folder = 'arrsize_{}_{}_{}_ntheta_{}'.format(ny, nx, nx, n_theta)
            coords = ref_ff.read_all_origin_coords(folder, n_theta)
            obj = np.stack([true_d, true_b], axis=3)
            rot = np.stack([ref_ff.apply_rotation(obj, c, folder) for c in coords])
            prj = ref_prop.multislice_propagate_cnn(rot[..., 0], rot[..., 1], np.ones((ny, nx)), np.zeros((ny, nx)), 5000., [1e-7] * 3,
                                                    kernel_size=17, free_prop_cm=1e-4)
            store['prj'] = prj.astype('complex64')
            g13['prj'] = store['prj']
            np.random.seed(5)
            with contextlib.redirect_stdout(io.StringIO()):
                ref_ff.reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=2, learning_rate=1e-7, minibatch_size=mb,
                                             energy_ev=5000, psize_cm=1e-7, free_prop_cm=1e-4, save_path='.', output_folder='out',
                                             initial_guess=[init_d.copy(), init_b.copy()], shrink_cycle=None, kernel_size=17,
                                             alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
            g13['delta_sub'] = np.asarray(written['delta_ds_1'])[::2, ::2, ::2].astype(np.float32)
            g13['beta_sub'] = np.asarray(written['beta_ds_1'])[::2, ::2, ::2].astype(np.float32)
            g13['delta_moved_max'] = np.abs(np.asarray(written['delta_ds_1']) - init_d * mask).max()
            print('done: |delta - init| max =', g13['delta_moved_max'])
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, 'g13_reconstruct_fullfield_64.npz'), **g13)
    print('wrote g13')


class _StopAfterFirstGradient(Exception):
    pass


def main_g14(fft=False, directional=None):
    """directional: a dict -> G21 mode (see main_g21): the gradient stand-in evaluates central differences of the reference's
    calculate_loss along g13_inputs.g20_directions at the first minibatch, fills the dict and stops the loop; nothing is written.
    G14: reconstruct_ptychography with a 64 x 64 gaussian probe on a (64, 64, 64) object, 4 positions x 2 angles, minibatches
    of 2, two epochs; gradient by the oracle's analytic adjoint (as G13).  Volumes stored on every second voxel, float32.
    fft=True is G17: the name multislice_propagate_cnn that ptychography.py calls is bound to np_funcs' transfer-function
    forward model (as G15 does for full field), with the drivers' kind of probe (sigma 10: it has decayed at the window's
    edge, which that model does not mind)."""
    import contextlib
    import io
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import bdof_oracle as orc
    autograd = _setup_conv_reference()

    def oracle_grad(fn, argnums):
        assert list(argnums) == [0, 1]
        cl = dict(zip(fn.__code__.co_freevars, [c.cell_contents for c in fn.__closure__]))

        def g(obj_delta, obj_beta, this_i_theta, this_pos_batch, this_prj_batch):
            _, gd, gb = orc.ptycho_loss_and_grad(obj_delta, obj_beta, cl['coord_ls'][this_i_theta], cl['probe_pos'], this_pos_batch,
                                                 this_prj_batch, cl['probe_real'], cl['probe_imag'], tuple(cl['probe_size']), cl['energy_ev'],
                                                 cl['psize_cm'] * cl['ds_level'], propagator='fft' if fft else 'conv', kernel_size=17)
            return gd, gb
        return g

    def fd_directional(fn, argnums):
        assert list(argnums) == [0, 1]
        sys.path.insert(0, HERE)
        import g13_inputs as gi

        def g(obj_delta, obj_beta, this_i_theta, this_pos_batch, this_prj_batch):
            d0, b0 = np.array(obj_delta, dtype=np.float64), np.array(obj_beta, dtype=np.float64)
            out = []
            with contextlib.redirect_stdout(io.StringIO()):
                base = float(fn(d0, b0, this_i_theta, this_pos_batch, this_prj_batch))
                for v in gi.g20_directions(d0.shape):
                    for which, h in ((0, 1e-10), (1, 3e-11)):
                        # steps: |psi| of the far-field loss has a kink at every dark bin, so the central difference is off by a term
                        # LINEAR in h (3e-5 of the derivative at 1e-9, 1.5e-6 at 1e-10), and round-off takes over below (1.5e-5 at
                        # 1e-11): 1e-10 is the best this loss gives — the test's bound is 5e-6
                        vals = []
                        for sgn in (+1.0, -1.0):
                            d, b = d0.copy(), b0.copy()
                            if which == 0:
                                d += sgn * h * v
                            else:
                                b += sgn * h * v
                            vals.append(float(fn(d, b, this_i_theta, this_pos_batch, this_prj_batch)))
                        out.append((vals[0] - vals[1]) / (2 * h))
            directional.update(loss=base, dd=np.array(out).reshape(-1, 2), i_theta=int(this_i_theta), pos_batch=np.array(this_pos_batch),
                               delta_in=d0, beta_in=b0)
            raise _StopAfterFirstGradient()
        return g

    autograd.grad = oracle_grad if directional is None else fd_directional

    class _Comm(object):
        def Get_size(self):
            return 1

        def Get_rank(self):
            return 0

        def Barrier(self):
            pass

        def Allreduce(self, src, dst):
            dst[...] = src

    mpi4py = types.ModuleType('mpi4py')
    mpi4py.MPI = types.SimpleNamespace(COMM_WORLD=_Comm())
    sys.modules['mpi4py'] = mpi4py
    import propagation as ref_prop
    ref_prop.trange = range
    import ptychography as ref_pt
    ref_pt.trange = range
    ref_pt.time = types.SimpleNamespace(time=lambda: 42 * 60.0 + 1.0)          # seed 42
    sigma = 10. if fft else 40.
    if fft:
        ref_np = sys.modules['np_funcs']

        def fft_forward(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, kernel_size=17, free_prop_cm=None, debug=False):
            return ref_np.multislice_propagate_batch_numpy(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm[0],
                                                           free_prop_cm=free_prop_cm, obj_batch_shape=grid_delta.shape)[0]
        ref_pt.multislice_propagate_cnn = fft_forward
    forward = ref_pt.multislice_propagate_cnn

    obj_size, psz, n_theta, mb = (64, 64, 64), (64, 64), 2, 2     # X = Z: the reference's lookup tables are only sound for square (x, z)
    pos = [(20, 22), (24, 40), (42, 26), (40, 44)]
    yy, xx, zz = np.mgrid[:obj_size[0], :obj_size[1], :obj_size[2]].astype(np.float64)
    rng = np.random.default_rng(14)
    true_d = np.zeros(obj_size)
    for _ in range(5):
        c = (rng.uniform(16, 48), rng.uniform(16, 48), rng.uniform(16, 48))
        r = rng.uniform(3, 7)
        true_d += 3e-6 * np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2 + (zz - c[2]) ** 2) / (2 * r ** 2))
    true_b = 0.1 * true_d
    sys.path.insert(0, HERE)
    import g13_inputs
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    # a wide probe: the real-space propagator pads with the constant 1 and renormalises by the corner pixel
    # (propagation.py:79,91,109-110), so a probe that has decayed at the window's corner is scaled to nothing
    kw = dict(probe_mag_sigma=sigma, probe_phase_sigma=sigma, probe_phase_max=0.5)
    store, written = {}, {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr
            self.shape = arr.shape

        def __getitem__(self, key):
            return self.arr[key]

    class _File(object):
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, key):
            return _Dataset(store['prj'])

    ref_pt.h5py.File = _File
    dx = sys.modules['dxchange']
    dx.write_tiff = lambda arr, fname=None, dtype=None, overwrite=False: written.__setitem__(os.path.basename(fname), np.array(arr))
    g14 = {'probe_pos': np.array(pos), 'probe_size': np.array(psz), 'obj_size': np.array(obj_size)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            n = obj_size
            ref_pt.save_rotation_lookup(list(n), n_theta)
            folder = 'arrsize_{}_{}_{}_ntheta_{}'.format(n[0], n[1], n[2], n_theta)
            coords = ref_pt.read_all_origin_coords(folder, n_theta)
            pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
            half = (np.array(psz) / 2).astype('int')
            prj = np.zeros((n_theta, len(pos), psz[0], psz[1]), dtype='complex64')
            for t in range(n_theta):
                rot = ref_pt.apply_rotation(np.stack([true_d, true_b], axis=3), coords[t], folder)
                rot = np.pad(rot, ((half[0], half[0]), (half[1], half[1]), (0, 0), (0, 0)), mode='constant')
                subs = np.stack([rot[p[0]:p[0] + psz[0], p[1]:p[1] + psz[1]] for p in pos])
                prj[t] = forward(subs[..., 0], subs[..., 1], pr, pi_, 5000., [1e-7] * 3, free_prop_cm='inf')
            # 2 % amplitude noise: a noise-free far-field pattern of this weak object differs from the initial guess's by 4e-6 of
            # its size, a residual no float32 detector wave can resolve (measured data are not like that)
            prj = (prj * (1 + 0.02 * rng.normal(size=prj.shape))).astype('complex64')
            store['prj'] = prj
            g14['prj'] = prj
            if directional is not None:
                # G21: the data are those of the committed G14 / G17 fixture (the noise above is seeded, the check is cheap)
                stored = np.load(os.path.join(HERE, 'g17_reconstruct_ptychography_fft_64.npz' if fft else 'g14_reconstruct_ptychography_64.npz'))['prj']
                assert np.array_equal(stored, prj)
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        ref_pt.reconstruct_ptychography('data.h5', [tuple(p) for p in pos], psz, obj_size, theta_st=0, theta_end=2 * np.pi, n_epochs=1,
                                                        learning_rate=2e-7, minibatch_size=mb, energy_ev=5000, psize_cm=1e-7, save_path='.',
                                                        output_folder='out', initial_guess=[init_d.copy(), init_b.copy()], probe_type='gaussian',
                                                        dynamic_dropping=False, **kw)
                except _StopAfterFirstGradient:
                    pass
                os.chdir(cwd)
                return
            with contextlib.redirect_stdout(io.StringIO()):
                ref_pt.reconstruct_ptychography('data.h5', [tuple(p) for p in pos], psz, obj_size, theta_st=0, theta_end=2 * np.pi, n_epochs=2,
                                                learning_rate=2e-7, minibatch_size=mb, energy_ev=5000, psize_cm=1e-7, save_path='.',
                                                output_folder='out', initial_guess=[init_d.copy(), init_b.copy()], probe_type='gaussian',
                                                dynamic_dropping=False, **kw)
            full_d = np.asarray(written['delta_ds_1'])
            g14['delta_sub'] = full_d[::2, ::2, ::2].astype(np.float32)
            g14['beta_sub'] = np.asarray(written['beta_ds_1'])[::2, ::2, ::2].astype(np.float32)
            g14['delta_moved_max'] = np.abs(full_d - init_d).max()
            print('done: |delta - init| max =', g14['delta_moved_max'])
        finally:
            os.chdir(cwd)
    g14['probe_sigma'] = np.array(sigma)
    np.savez_compressed(os.path.join(HERE, 'g17_reconstruct_ptychography_fft_64.npz' if fft else 'g14_reconstruct_ptychography_64.npz'), **g14)
    print('wrote', 'g17' if fft else 'g14')


def main_g15(n=64, noise=0.0):
    """G15: reconstruct_fullfield at (64, 64, 64) around np_funcs' FFT forward; gradient by the oracle's analytic adjoint.
    n = 256 is G18: the same at BASELINE config 2's volume size (256^3, 256 slices).  There the measured data are not stored
    (4 x 256^2 complex values): generator and test both compute them with the oracle's forward model from the phantom formula
    of g13_inputs.phantom — they are an input, any array would do as long as both sides read the same one; the stored outputs
    are the reference loop's volumes on every eighth voxel per axis."""
    import contextlib
    import io
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import bdof_oracle as orc
    autograd = _setup_conv_reference()
    ref_util, ref_np = sys.modules.get('util'), sys.modules.get('np_funcs')

    def oracle_grad(fn, argnums):
        assert list(argnums) == [0, 1]
        cl = dict(zip(fn.__code__.co_freevars, [c.cell_contents for c in fn.__closure__]))

        def g(obj_delta, obj_beta, this_ind_batch, this_prj_batch):
            _, gd, gb = orc.fullfield_loss_and_grad(obj_delta, obj_beta, cl['coord_ls'], this_ind_batch, this_prj_batch, cl['probe_real'],
                                                    cl['probe_imag'], cl['energy_ev'], cl['psize_cm'] * cl['ds_level'], free_prop_cm=cl['free_prop_cm'],
                                                    alpha=cl['alpha'], alpha_d=cl['alpha_d'], alpha_b=cl['alpha_b'], gamma=cl['gamma'], with_reg=True)
            return gd, gb
        return g

    autograd.grad = oracle_grad
    import propagation as ref_prop
    ref_prop.trange = range
    import fullfield as ref_ff

    def fft_forward(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, kernel_size=17, free_prop_cm=None, debug=False):
        return ref_np.multislice_propagate_batch_numpy(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm[0],
                                                       free_prop_cm=free_prop_cm, obj_batch_shape=grid_delta.shape)[0]
    ref_ff.multislice_propagate_cnn = fft_forward

    sys.path.insert(0, HERE)
    import g13_inputs
    ny, nx, n_theta, mb = n, n, 4, 2
    sub = 2 if n == 64 else 8
    if n == 64:
        rng = np.random.default_rng(15)
        yy, xx, zz = np.mgrid[:ny, :nx, :nx].astype(np.float64)
        true_d = np.zeros((ny, nx, nx))
        for _ in range(5):
            c = (rng.uniform(16, 48), rng.uniform(20, 44), rng.uniform(20, 44))
            r = rng.uniform(4, 8)
            true_d += 3e-6 * np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2 + (zz - c[2]) ** 2) / (2 * r ** 2))
    else:
        true_d = g13_inputs.phantom((n, n, n))
    true_b = 0.1 * true_d
    mask = g13_inputs.mask((n, n, n))
    init_d, init_b = g13_inputs.initial_guess((n, n, n))
    store, written = {}, {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, key):
            return self.arr[key]

    class _File(object):
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, key):
            return _Dataset(store['prj'])

    ref_ff.h5py.File = _File
    dx = sys.modules['dxchange']
    dx.read_tiff_stack = lambda fname, ind, digit=5: np.array(mask)
    dx.read_tiff = lambda fname: np.array(mask)
    dx.write_tiff = lambda arr, fname=None, dtype=None, overwrite=False: written.__setitem__(os.path.basename(fname), np.array(arr))
    dx.write_tiff_stack = lambda *a, **k: None
    g15 = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            ref_ff.save_rotation_lookup([ny, nx, nx], n_theta)
            folder = 'arrsize_{}_{}_{}_ntheta_{}'.format(ny, nx, nx, n_theta)
            coords = ref_ff.read_all_origin_coords(folder, n_theta)
            obj = np.stack([true_d, true_b], axis=3)
            rot = np.stack([ref_ff.apply_rotation(obj, c, folder) for c in coords]) if n == 64 else None
            if n == 64:
                prj = fft_forward(rot[..., 0], rot[..., 1], np.ones((ny, nx)), np.zeros((ny, nx)), 5000., [1e-7] * 3, free_prop_cm=1e-4)
                store['prj'] = prj.astype('complex64')
                g15['prj'] = store['prj']
            else:
                store['prj'] = g13_inputs.data_from_phantom(orc, (n, n, n), n_theta, noise)
            np.random.seed(5)
            with contextlib.redirect_stdout(io.StringIO()):
                ref_ff.reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=2, learning_rate=1e-7, minibatch_size=mb,
                                             energy_ev=5000, psize_cm=1e-7, free_prop_cm=1e-4, save_path='.', output_folder='out',
                                             initial_guess=[init_d.copy(), init_b.copy()], shrink_cycle=None,
                                             alpha_d=1.5e-8, alpha_b=1.5e-9, gamma=1e-11)
            full_d = np.asarray(written['delta_ds_1'])
            g15['delta_sub'] = full_d[::sub, ::sub, ::sub].astype(np.float32)
            g15['beta_sub'] = np.asarray(written['beta_ds_1'])[::sub, ::sub, ::sub].astype(np.float32)
            g15['delta_moved_max'] = np.abs(full_d - init_d * mask).max()
            print('done: |delta - init| max =', g15['delta_moved_max'])
        finally:
            os.chdir(cwd)
    name = 'g15_reconstruct_fullfield_fft_64.npz' if n == 64 else ('g19_reconstruct_fullfield_fft_256_noisy.npz' if noise else 'g18_reconstruct_fullfield_fft_256.npz')
    np.savez_compressed(os.path.join(HERE, name), **g15)
    print('wrote', name)


def main_g20():
    """G20: the GRADIENT at a size the GPU kernels take, from the reference alone.  reconstruct_fullfield is started at (64, 64, 64)
    exactly as for G13 (real-space propagator, what fullfield.py calls) and G15 (np_funcs' FFT forward bound to the name it calls);
    the stand-in for autograd.grad receives the reference's own calculate_loss closure at the first minibatch, evaluates central
    differences of it along six directions (float64; 24 evaluations of the reference's loss per forward model) and stops the
    loop.  No oracle code is involved in the stored numbers: they pin the oracle's analytic gradient (CPU test) and the device
    gradient (GPU test) at this size, which G13-G19 — reference loop + oracle gradient — do not."""
    import contextlib
    import io
    import tempfile
    autograd = _setup_conv_reference()
    ref_np = sys.modules.get('np_funcs')
    sys.path.insert(0, HERE)
    import g13_inputs
    ny = nx = 64
    n_theta, mb = 4, 2
    dirs = g13_inputs.g20_directions((ny, nx, nx))
    got = {}

    class _Stop(Exception):
        pass

    def fd_directional(fn, argnums):
        assert list(argnums) == [0, 1]

        def g(obj_delta, obj_beta, this_ind_batch, this_prj_batch):
            d0, b0 = np.array(obj_delta, dtype=np.float64), np.array(obj_beta, dtype=np.float64)
            out = []
            with contextlib.redirect_stdout(io.StringIO()):
                base = float(fn(d0, b0, this_ind_batch, this_prj_batch))
                for v in dirs:
                    for which, h in ((0, 1e-9), (1, 1e-10)):
                        vals = []
                        for sgn in (+1.0, -1.0):
                            d, b = d0.copy(), b0.copy()
                            if which == 0:
                                d += sgn * h * v
                            else:
                                b += sgn * h * v
                            vals.append(float(fn(d, b, this_ind_batch, this_prj_batch)))
                        out.append((vals[0] - vals[1]) / (2 * h))
            got['loss'] = base
            got['dd'] = np.array(out).reshape(len(dirs), 2)          # [direction][delta, beta]: dL/dt along t * v
            got['ind'] = np.array(this_ind_batch)
            got['delta_in'], got['beta_in'] = d0, b0
            raise _Stop()
        return g

    autograd.grad = fd_directional
    import propagation as ref_prop
    ref_prop.trange = range
    import fullfield as ref_ff
    ref_ff.trange = range
    real_conv = ref_ff.multislice_propagate_cnn

    def fft_forward(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, kernel_size=17, free_prop_cm=None, debug=False):
        return ref_np.multislice_propagate_batch_numpy(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm[0],
                                                       free_prop_cm=free_prop_cm, obj_batch_shape=grid_delta.shape)[0]

    mask = g13_inputs.mask((ny, nx, nx))
    init_d, init_b = g13_inputs.initial_guess((ny, nx, nx))
    store = {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, key):
            return self.arr[key]

    class _File(object):
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, key):
            return _Dataset(store['prj'])

    ref_ff.h5py.File = _File
    dx = sys.modules['dxchange']
    dx.read_tiff_stack = lambda fname, ind, digit=5: np.array(mask)
    dx.read_tiff = lambda fname: np.array(mask)
    dx.write_tiff = lambda *a, **k: None
    dx.write_tiff_stack = lambda *a, **k: None
    g20 = {}
    cwd = os.getcwd()
    for model, fixture in (('fft', 'g15_reconstruct_fullfield_fft_64.npz'), ('conv', 'g13_reconstruct_fullfield_64.npz')):
        ref_ff.multislice_propagate_cnn = fft_forward if model == 'fft' else real_conv
        store['prj'] = np.load(os.path.join(HERE, fixture))['prj']       # the data of G15 / G13 (outputs of the reference's forward models)
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                ref_ff.save_rotation_lookup([ny, nx, nx], n_theta)
                np.random.seed(5)
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        ref_ff.reconstruct_fullfield('data.h5', theta_st=0, theta_end=2 * np.pi, n_epochs=1, learning_rate=1e-7, minibatch_size=mb,
                                                     energy_ev=5000, psize_cm=1e-7, free_prop_cm=1e-4, save_path='.', output_folder='out',
                                                     initial_guess=[init_d.copy(), init_b.copy()], shrink_cycle=None, kernel_size=17,
                                                     alpha_d=0., alpha_b=0., gamma=0.)
                except _Stop:
                    pass
            finally:
                os.chdir(cwd)
        assert np.array_equal(got['delta_in'], init_d * mask) and np.array_equal(got['beta_in'], init_b * mask)
        g20[model + '_loss'] = np.array(got['loss'])
        g20[model + '_dd'] = got['dd']
        g20[model + '_ind'] = got['ind']
        print(model, 'loss', got['loss'], 'batch', got['ind'], 'dL/dt:', got['dd'].ravel())
    np.savez_compressed(os.path.join(HERE, 'g20_directional_derivatives_64.npz'), **g20)
    print('wrote g20')


def main_g21():
    """G21: G20 for ptychography — central differences of the reference's own calculate_loss (cnn_propagator/ptychography.py:30-81:
    rotation, zero padding, window cut, forward model, far-field loss) along the six directions, at the first minibatch of the
    G17 run (np_funcs' FFT forward bound in, sigma-10 probe) and of the G14 run (the real-space propagator it literally calls,
    sigma-40 probe).  Each in a process of its own state: run as two invocations (--g21 fft | --g21 conv), merged on the second."""
    which = sys.argv[sys.argv.index('--g21') + 1]
    out = {}
    main_g14(fft=which == 'fft', directional=out)
    path = os.path.join(HERE, 'g21_ptycho_directional_derivatives_64.npz')
    g21 = dict(np.load(path)) if os.path.exists(path) else {}
    g21[which + '_loss'] = np.array(out['loss'])
    g21[which + '_dd'] = out['dd']
    g21[which + '_i_theta'] = np.array(out['i_theta'])
    g21[which + '_pos_batch'] = out['pos_batch']
    np.savez_compressed(path, **{k: g21[k] for k in sorted(g21)})
    print(which, 'loss', out['loss'], 'theta', out['i_theta'], 'positions', out['pos_batch'].tolist(), 'dL/dt', out['dd'].ravel())


def main_g16():
    """G16: tensorflow_recon/create_noisy_data.py run as the script it is (runpy), in a directory laid out with the file
    names it hard-codes; its clock is frozen (it seeds numpy from int(time.time())) and h5py / dxchange are in-memory
    stand-ins.  Its source path contains 'ptycho', so the ptychography branch (:45-72) is the one that runs."""
    import contextlib
    import io
    import runpy
    import tempfile
    import time as _time

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    rng = np.random.default_rng(16)
    n = 12
    gd = np.zeros((n, n, n))
    gd[3:9, 3:9, 3:9] = rng.uniform(1e-6, 3e-6, size=(6, 6, 6))
    src = (rng.uniform(0.5, 3.0, size=(2, 3, 8, 8)) * np.exp(1j * rng.uniform(0, 6.28, size=(2, 3, 8, 8)))).astype('complex64')
    written = {}

    class _Dataset(object):
        def __init__(self, arr):
            self.arr = arr
            self.shape = arr.shape

        def __getitem__(self, key):
            return self.arr[key]

        def __setitem__(self, key, val):
            self.arr[key] = val

    class _Group(object):
        def __init__(self, fname):
            self.fname = fname

        def create_dataset(self, name, dtype=None, shape=None):
            d = _Dataset(np.zeros(shape, dtype=dtype))
            written[os.path.basename(self.fname)] = d
            return d

    class _File(object):
        def __init__(self, fname, mode='r'):
            self.fname, self.mode = fname, mode

        def __getitem__(self, key):
            assert key == 'exchange/data' and self.mode == 'r'
            return _Dataset(src)

        def create_group(self, name):
            return _Group(self.fname)

    stub('h5py', File=_File)
    stub('dxchange', write_tiff=lambda *a, **k: None)
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    cwd = os.getcwd()
    real_time = _time.time
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            os.makedirs('cell/phantom')
            os.makedirs('cell/ptychography')
            np.save('cell/phantom/grid_delta.npy', gd)
            _time.time = lambda: 1234.5
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                runpy.run_path('/root/reference/tensorflow_recon/create_noisy_data.py', run_name='__main__')
        finally:
            _time.time = real_time
            os.chdir(cwd)
    g16 = {'grid_delta': gd, 'src': src, 'seed': np.array(1234)}
    for k, v in written.items():
        g16[k.replace('.h5', '').replace('.', 'p')] = v.arr
        print(k, v.arr.shape, float(np.abs(v.arr).max()))
    np.savez_compressed(os.path.join(HERE, 'g16_noisy_data.npz'), **g16)
    print('wrote g16')


def main_h5():
    """G7: exchange/data files written by h5py 3.3.0 (run under /opt/conda/bin/python3.9)."""
    import h5py
    rng = np.random.default_rng(7)
    dat = (rng.normal(size=(3, 8, 8)) + 1j * rng.normal(size=(3, 8, 8))).astype('complex64')
    path = os.path.join(HERE, 'g7_fullfield_3x8x8.h5')
    with h5py.File(path, 'w') as f:          # same calls as cnn_propagator/simulation.py:124-126,159
        grp = f.create_group('exchange')
        d = grp.create_dataset('data', (3, 8, 8), dtype='complex64')
        d[...] = dat
    np.save(os.path.join(HERE, 'g7_fullfield_3x8x8.npy'), dat)
    dat4 = (rng.normal(size=(2, 3, 4, 6)) + 1j * rng.normal(size=(2, 3, 4, 6))).astype('complex64')
    path = os.path.join(HERE, 'g7_ptycho_2x3x4x6.h5')
    with h5py.File(path, 'w') as f:          # cnn_propagator/simulation.py:363
        grp = f.create_group('exchange')
        d = grp.create_dataset('data', (2, 3, 4, 6), dtype='complex64')
        for i in range(2):
            d[i] = dat4[i]
    np.save(os.path.join(HERE, 'g7_ptycho_2x3x4x6.npy'), dat4)
    print('h5 fixtures written with h5py', h5py.__version__)


if __name__ == '__main__':
    if '--h5' in sys.argv:
        main_h5()
    elif '--g8' in sys.argv:
        main_g8()
    elif '--g9' in sys.argv:
        main_g9()
    elif '--g10' in sys.argv:
        main_g10()
    elif '--g11' in sys.argv:
        main_g11()
    elif '--g12' in sys.argv:
        main_g12()
    elif '--g13' in sys.argv:
        main_g13()
    elif '--g14' in sys.argv:
        main_g14()
    elif '--g15' in sys.argv:
        main_g15()
    elif '--g17' in sys.argv:
        main_g14(fft=True)
    elif '--g18' in sys.argv:
        main_g15(n=256)
    elif '--g19' in sys.argv:
        main_g15(n=256, noise=0.02)
    elif '--g16' in sys.argv:
        main_g16()
    elif '--g20' in sys.argv:
        main_g20()
    elif '--g21' in sys.argv:
        main_g21()
    else:
        main()
