"""Inputs of golden vectors G13 / G14 that are formulas rather than stored arrays (shared by make_golden.py and the GPU tests):
initial guess and support mask of the (64, 64, 64) reconstruct_fullfield run, initial guess of the (64, 64, 64)
reconstruct_ptychography run."""
import numpy as np

SHAPE = (64, 64, 64)


def initial_guess(shape=SHAPE):
    y, x, z = np.mgrid[:shape[0], :shape[1], :shape[2]].astype(np.float64)
    w = np.sin(0.37 * y + 0.91 * x + 1.7 * z) * np.cos(0.53 * x - 0.29 * z + 0.11 * y)
    d = (8.7e-7 * (1 + 0.2 * w)).astype(np.float32)
    b = (5.1e-8 * (1 - 0.2 * w)).astype(np.float32)
    return d.astype(np.float64), b.astype(np.float64)          # exactly what a float32 volume holds


def mask(shape=SHAPE):
    m = np.ones(shape, dtype=np.float32)
    m[:3] = 0
    return m


def phantom(shape):
    """Five soft blobs (delta up to ~3e-6) placed by fixed fractions of the volume: the G18 phantom."""
    y, x, z = np.mgrid[:shape[0], :shape[1], :shape[2]].astype(np.float64)
    d = np.zeros(shape)
    for cy, cx, cz, r in ((0.35, 0.40, 0.45, 0.08), (0.60, 0.55, 0.40, 0.11), (0.45, 0.62, 0.60, 0.07), (0.55, 0.38, 0.58, 0.09),
                          (0.50, 0.50, 0.50, 0.05)):
        d += 3e-6 * np.exp(-((y - cy * shape[0]) ** 2 + (x - cx * shape[1]) ** 2 + (z - cz * shape[2]) ** 2) / (2 * (r * shape[0]) ** 2))
    return d


def data_from_phantom(orc, shape, n_theta, noise=0.0):
    """exchange/data of G18, (n_theta, Y, X) complex64: the oracle's forward model on the phantom, 1 um to the detector;
    noise: relative amplitude noise (seeded), as measured data have."""
    d = phantom(shape)
    coords = orc.rotation_lookup(list(shape), n_theta)
    one, zero = np.ones(shape[:2]), np.zeros(shape[:2])
    out = []
    for c in coords:
        rot = orc.apply_rotation(np.stack([d, 0.1 * d], axis=3), c)[None]
        w, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], one, zero, 5000., 1e-7, 1e-4, rot[..., 0].shape,
                                                    return_probe_array=False)
        out.append(w[0])
    out = np.array(out)
    if noise:
        out = out * (1 + noise * np.random.default_rng(18).normal(size=out.shape))
    return out.astype(np.complex64)


def g20_directions(shape, n_dir=6):
    """The directions of golden vector G20 (shared by generator and tests): unit-less (Y, X, Z) fields, four of random signs
    (seeded) and two smooth blobs; applied as delta += t * 1e-9 * v, beta += t * 1e-10 * v."""
    rng = np.random.default_rng(20)
    y, x, z = np.mgrid[:shape[0], :shape[1], :shape[2]].astype(np.float64)
    out = [np.sign(rng.normal(size=shape)) for _ in range(n_dir - 2)]
    for cy, cx, cz, r in ((0.45, 0.55, 0.5, 0.12), (0.6, 0.4, 0.45, 0.2)):
        out.append(np.exp(-((y - cy * shape[0]) ** 2 + (x - cx * shape[1]) ** 2 + (z - cz * shape[2]) ** 2) / (2 * (r * shape[0]) ** 2)))
    return out
