"""Host-side helpers of the hot path, mirroring the names of the reference's cnn_propagator/util.py.

Everything here is set-up work that runs once per reconstruction (transfer functions, rotation
lookup tables, task splitting); the per-step arithmetic lives in libbdof.so.  Tables are computed in
float64 exactly as the reference does so that the integer lookups are bit-identical.
"""
import os
import sys

import numpy as np

PI = 3.1415927   # the reference's literal (cnn_propagator/util.py:20); it enters k and H


def get_kernel(dist_nm, lmbda_nm, voxel_nm, grid_shape, pi=PI):
    """Fresnel transfer function H(u, v) on the reference's inclusive linspace mesh
    (cnn_propagator/util.py:73-102).  Returns a centred (Y, X) complex128 array."""
    ny, nx = int(grid_shape[0]), int(grid_shape[1])
    v = np.linspace(-1. / (2. * voxel_nm[1]), 1. / (2. * voxel_nm[1]), ny)     # rows: v_max = 1/(2 voxel[1])
    u = np.linspace(-1. / (2. * voxel_nm[0]), 1. / (2. * voxel_nm[0]), nx)     # cols: u_max = 1/(2 voxel[0])
    uu, vv = np.meshgrid(u, v)
    k = 2 * pi / lmbda_nm
    return np.exp(1j * k * dist_nm) * np.exp(-1j * pi * lmbda_nm * dist_nm * (uu ** 2 + vv ** 2))


def get_kernel_ir(dist_nm, lmbda_nm, voxel_nm, grid_shape, pi=PI):
    """Impulse-response form of the Fresnel kernel (cnn_propagator/util.py:105-127): the real-space kernel sampled on the
    pixel grid, Fourier transformed, times the pixel area.  The reference's propagator never selects it (np_funcs.py:55
    forces 'TF' after computing the sampling criterion of :51-53); here it is the detector step's `detector_kernel='IR'`
    option (detector_kernel below, MultisliceEngine.set_physics), 'auto' applies the reference's criterion."""
    ny, nx = int(grid_shape[0]), int(grid_shape[1])
    k = 2 * pi / lmbda_nm
    sy, sx = voxel_nm[0] * ny, voxel_nm[1] * nx
    x = np.arange(-sx / 2., -sx / 2. + sx, voxel_nm[1])
    y = np.arange(-sy / 2., -sy / 2. + sy, voxel_nm[0])
    xx, yy = np.meshgrid(x, y)
    h = np.exp(1j * k * dist_nm) / (1j * lmbda_nm * dist_nm) * np.exp(1j * k / (2 * dist_nm) * (xx ** 2 + yy ** 2))
    return np.fft.fftshift(np.fft.fft2(h)) * voxel_nm[0] * voxel_nm[1]


def upsample_2x(arr):
    """Multiscale hand-over (cnn_propagator/util.py:350-360): zero-stuffing by 2 along the three spatial axes followed by a
    sigma-1 gaussian filter; 4-D arrays channel by channel."""
    from scipy.ndimage import gaussian_filter
    arr = np.asarray(arr)
    if arr.ndim == 4:
        return np.stack([upsample_2x(arr[..., i]) for i in range(arr.shape[3])], axis=3)
    out = np.zeros([2 * n for n in arr.shape])
    out[::2, ::2, ::2] = arr
    return gaussian_filter(out, 1)


def get_kernel_tile(dist_nm, lmbda_nm, voxel_nm, tile_shape, field_shape, pi=PI):
    """The transfer function of a (FY, FX) FIELD's propagator, sampled at the FFT frequencies of a (TY, TX) tile of it.

    get_kernel puts H on an inclusive linspace of N points (quirk Q4): centred index k of an N-point axis, i.e. FFT frequency
    f = (k - N//2) / (N dx), carries u(f) = -u_max + (f N dx + N//2) * 2 u_max / (N - 1).  A tile cut out of the field must apply
    the SAME operator, so its bins (FFT frequencies (j - T//2) / (T dx)) take H at u(f) of the field's mesh, not of a T-point
    mesh of its own (whose effective step is (T - 1) / T instead of (N - 1) / N of the nominal one: 0.34 % in distance between
    512 and 4096 points, 0.7 rad at the band edge after 1024 slices).  Equals get_kernel when tile_shape == field_shape."""
    def axis(t, n, vox):
        u_max = 1. / (2. * vox)
        f = (np.arange(t) - t // 2) / (t * vox)
        return -u_max + (f * n * vox + n // 2) * 2. * u_max / (n - 1)
    v = axis(int(tile_shape[0]), int(field_shape[0]), voxel_nm[1])
    u = axis(int(tile_shape[1]), int(field_shape[1]), voxel_nm[0])
    uu, vv = np.meshgrid(u, v)
    k = 2 * pi / lmbda_nm
    return np.exp(1j * k * dist_nm) * np.exp(-1j * pi * lmbda_nm * dist_nm * (uu ** 2 + vv ** 2))


def detector_kernel_kind(kind, dist_nm, lmbda_nm, voxel_nm, grid_shape):
    """'TF' or 'IR' for the detector step.  'auto' is the criterion cnn_propagator/np_funcs.py:51-53 computes (and then
    overrides with 'TF' at :55): transfer function where the mean voxel size exceeds lambda z / L, impulse response otherwise."""
    if kind in ('TF', 'IR'):
        return kind
    if kind != 'auto':
        raise ValueError("detector_kernel must be 'TF', 'IR' or 'auto'")
    voxel_nm = np.asarray(voxel_nm, dtype=float)
    size_nm = np.array(grid_shape) * voxel_nm[:len(grid_shape)]
    crit_samp = lmbda_nm * dist_nm / np.prod(size_nm) ** (1. / len(size_nm))
    return 'TF' if np.prod(voxel_nm) ** (1. / 3) > crit_samp else 'IR'


def centred_kernel(dist_nm, lmbda_nm, voxel_nm, ny, nx, pi=PI, field_shape=None, kernel='TF'):
    """The centred (ny, nx) multiplier of one propagation step: get_kernel, get_kernel_tile (a tile of a larger field) or
    get_kernel_ir (kernel='IR', whole fields only)."""
    if kernel == 'IR':
        if field_shape is not None:
            raise ValueError("kernel='IR' is defined for whole fields, not for tiles of a larger one")
        return get_kernel_ir(dist_nm, lmbda_nm, voxel_nm, (ny, nx), pi=pi)
    if kernel != 'TF':
        raise ValueError("kernel must be 'TF' or 'IR'")
    if field_shape is None:
        return get_kernel(dist_nm, lmbda_nm, voxel_nm, (ny, nx), pi=pi)
    return get_kernel_tile(dist_nm, lmbda_nm, voxel_nm, (ny, nx), field_shape, pi=pi)


def device_transfer_function(dist_nm, lmbda_nm, voxel_nm, ny, nx, pi=PI, field_shape=None, dtype=np.complex64, kernel='TF'):
    """H prepared for libbdof: un-shifted [ky][kx], 1/(NX*NY) folded in, complex64 (complex128: the float64 adjoint sweep's
    table, bdof_set_physics_f64).  field_shape: the (ny, nx) wavefield is a tile of a larger field whose propagator it applies
    (get_kernel_tile).  kernel='IR': the impulse-response form (get_kernel_ir, cnn_propagator/util.py:105-127)."""
    h = centred_kernel(dist_nm, lmbda_nm, voxel_nm, ny, nx, pi, field_shape, kernel)
    hs = np.fft.ifftshift(h) / float(nx * ny)
    return np.ascontiguousarray(hs.astype(dtype))


def transfer_function_dc(dist_nm, lmbda_nm, voxel_nm, ny, nx, pi=PI, field_shape=None, kernel='TF'):
    """(re, im) of ifftshift(H)[0][0]: the factor a constant wave picks up in one transfer-function step."""
    h = centred_kernel(dist_nm, lmbda_nm, voxel_nm, ny, nx, pi, field_shape, kernel)
    v = complex(np.fft.ifftshift(h)[0, 0])
    return v.real, v.imag


def conv_kernel_separable(delta_nm, lmbda_nm, voxel_nm, grid_shape, kernel_size, pi=PI):
    """The truncated real-space Fresnel kernel of cnn_propagator/propagation.py:35-44 in separable form:
    K[p][q] = e * ky[p] * kx[q].  H on the (Y-1, X-1) mesh is e * outer(fv, fu), so its inverse FFT, fftshift and centre
    crop factorise exactly.  Returns (ky, kx, e) in complex128."""
    ny, nx = int(grid_shape[0]) - 1, int(grid_shape[1]) - 1
    half = int((kernel_size - 1) / 2)

    def one(n, vox):
        f = np.exp(-1j * pi * lmbda_nm * delta_nm * np.linspace(-1. / (2. * vox), 1. / (2. * vox), n) ** 2)
        k1 = np.fft.fftshift(np.fft.ifft(np.fft.ifftshift(f)))
        mid = int((n - 1) / 2)
        return k1[mid - half:mid + half + 1]

    ky = one(ny, voxel_nm[1])       # rows of H follow v_max = 1/(2 voxel[1])  (get_kernel)
    kx = one(nx, voxel_nm[0])
    e = np.exp(1j * (2 * pi / lmbda_nm) * delta_nm)
    return ky, kx, e


def conv_probe_stack(probe, ky, kx, e, n_slice):
    """The probe carried through EMPTY space by the real-space propagator's own step (cnn_propagator/propagation.py:79-104
    without an object), in float64: p_0 = probe, p_{z+1} = K * pad(p_z, edge_z) ('valid' true convolution, K = e ky (x) kx),
    edge_{z+1} = sum(K) edge_z, edge_0 = 1.  Returns the S + 1 planes (S + 1, Y, X) complex128 — the carrier field of
    bdof_set_conv_probe_stack.  Two 1-D passes of shifted adds: cheap enough for a 512^2 x 512 stack."""
    h = (len(ky) - 1) // 2
    ksum = e * ky.sum() * kx.sum()
    p = np.array(probe, dtype=np.complex128)
    ny, nx = p.shape
    planes = np.empty((n_slice + 1, ny, nx), dtype=np.complex128)
    planes[0] = p
    edge = 1.0 + 0j
    for z in range(n_slice):
        padded = np.pad(p, h, mode='constant', constant_values=edge)
        t = np.zeros((ny, nx + 2 * h), dtype=np.complex128)
        for a in range(2 * h + 1):                    # out[y] = sum_a ky[a] in[y + 2h - a]: a true convolution
            t += ky[a] * padded[2 * h - a:2 * h - a + ny, :]
        q = np.zeros((ny, nx), dtype=np.complex128)
        for b in range(2 * h + 1):
            q += kx[b] * t[:, 2 * h - b:2 * h - b + nx]
        p = e * q
        edge = edge * ksum
        planes[z + 1] = p
    return planes


def rotation_lookup(array_size, n_theta):
    """Nearest-neighbour rotation source coordinates for every angle, as save_rotation_lookup builds
    them (cnn_propagator/util.py:294-332) but kept in memory: list of (X*Z, 2) int arrays.  Angles are
    linspace(0, 2*pi, n_theta) whatever theta_st/theta_end say (reference behaviour)."""
    ny, nx, nz = [int(a) for a in array_size]
    cx, cz = np.floor(nx / 2), np.floor(nz / 2)
    xs = np.repeat(np.arange(nx), nz) - cx
    zs = np.tile(np.arange(nz), nx) - cz
    new = np.stack([xs, zs]).astype(np.float32)
    out = []
    for theta in np.linspace(0, 2 * np.pi, n_theta):
        rot = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
        old = np.matmul(rot, new)
        c1 = np.clip(np.round(old[0] + cx).astype(int), 0, nx - 1)
        c2 = np.clip(np.round(old[1] + cz).astype(int), 0, nz - 1)
        out.append(np.stack([c1, c2], axis=1))
    return out


def save_rotation_lookup(array_size, n_theta, dest_folder=None):
    """Drop-in for cnn_propagator/util.py:294-347: writes the same .npy files and returns the tables.  The folder appears
    atomically (written under a temporary name, then renamed), so a concurrent reader never sees half of it."""
    coords = rotation_lookup(array_size, n_theta)
    if dest_folder is None:
        dest_folder = 'arrsize_{}_{}_{}_ntheta_{}'.format(array_size[0], array_size[1], array_size[2], n_theta)
    tmp = '{}.tmp{}'.format(dest_folder.rstrip('/'), os.getpid())
    os.makedirs(tmp, exist_ok=True)
    for i, arr in enumerate(coords):
        np.save(os.path.join(tmp, '{:04}'.format(i)), arr)
    ny, nx, nz = [int(a) for a in array_size]
    coord0 = np.repeat(np.arange(ny), nx * nz)
    coord1 = np.tile(np.repeat(np.arange(nx), nz), ny).astype(float)
    coord2 = np.tile(np.tile(np.arange(nz), nx), ny).astype(float)
    for i, coord in enumerate([coord0, coord1, coord2]):
        np.save(os.path.join(tmp, 'coord{}_vec'.format(i)), coord)
    if os.path.isdir(dest_folder):                      # refresh an existing (possibly incomplete) folder file by file
        for f in os.listdir(tmp):
            os.replace(os.path.join(tmp, f), os.path.join(dest_folder, f))
        os.rmdir(tmp)
    else:
        os.rename(tmp, dest_folder)
    return coords


def rotation_lookup_files(array_size, n_theta, comm):
    """The tables of the run, through the reference's files (cnn_propagator/fullfield.py:209-215, ptychography.py:149-158):
    rank 0 writes the folder if it is missing or incomplete, EVERY rank passes the same barrier, then every rank reads."""
    folder = 'arrsize_{}_{}_{}_ntheta_{}'.format(array_size[0], array_size[1], array_size[2], n_theta)
    if comm.rank == 0:
        complete = all(os.path.exists(os.path.join(folder, '{:04}.npy'.format(i))) for i in range(n_theta))
        if not complete:
            save_rotation_lookup(array_size, n_theta, folder)
    comm.Barrier()
    return read_all_origin_coords(folder, n_theta)


def read_all_origin_coords(src_folder, n_theta):
    """cnn_propagator/util.py:369-374."""
    return [np.load(os.path.join(src_folder, '{:04}.npy'.format(i))) for i in range(n_theta)]


def device_rotation_tables(coords, nx, nz):
    """int32 tables for libbdof from the reference-style coordinate lists.

    tab[a][z][x]   = source row (x' * Z + z') of the [X][Z][Y] volume feeding rotated row (x, z)
    off/order      = per-angle inverse (CSR): for every volume row the rotated rows (z*X + x) it feeds
    """
    n = len(coords)
    tab = np.empty((n, nz, nx), dtype=np.int32)
    off = np.empty((n, nx * nz + 1), dtype=np.int32)
    order = np.empty((n, nz * nx), dtype=np.int32)
    for a, c in enumerate(coords):
        src = (c[:, 0] * nz + c[:, 1]).reshape(nx, nz)       # indexed [x][z]
        t = np.ascontiguousarray(src.T)                       # [z][x]
        tab[a] = t
        dest = t.reshape(-1)                                  # indexed by s = z*X + x
        order[a] = np.argsort(dest, kind='stable')
        off[a, 0] = 0
        np.cumsum(np.bincount(dest, minlength=nx * nz), out=off[a, 1:])
    return tab, off, order


def split_tasks(arr, split_size):
    """cnn_propagator/util.py:271-277."""
    return [arr[i:i + split_size] for i in range(0, len(arr), split_size)]


def mag_phase_to_real_imag(mag, phase):
    """cnn_propagator/util.py:265-268."""
    a = mag * np.exp(1j * phase)
    return a.real, a.imag


def gaussian_probe(shape, probe_mag_sigma, probe_phase_sigma, probe_phase_max):
    """Gaussian probe of cnn_propagator/fullfield.py:299-310 / ptychography.py:212-222."""
    py = np.arange(shape[0]) - (shape[0] - 1.) / 2
    px = np.arange(shape[1]) - (shape[1] - 1.) / 2
    pxx, pyy = np.meshgrid(px, py)
    mag = np.exp(-(pxx ** 2 + pyy ** 2) / (2 * probe_mag_sigma ** 2))
    phase = probe_phase_max * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * probe_phase_sigma ** 2))
    return mag_phase_to_real_imag(mag, phase)


def print_flush(a, designate_rank=None, this_rank=None):
    """cnn_propagator/util.py:248-256."""
    if designate_rank is None or this_rank == designate_rank:
        print(a)
    sys.stdout.flush()


# ---- layout conversion between the reference's arrays and libbdof's device layout ------------------
def volume_to_rows(obj_delta, obj_beta):
    """(Y, X, Z) delta and beta -> [X][Z][Y] (delta, beta) pairs, float32."""
    pair = np.stack([obj_delta, obj_beta], axis=-1)
    return np.ascontiguousarray(pair.transpose(1, 2, 0, 3).astype(np.float32))


def rows_to_volume(rows):
    """Inverse of volume_to_rows: [X][Z][Y][2] -> (delta, beta) each (Y, X, Z)."""
    v = rows.transpose(2, 0, 1, 3)
    return np.ascontiguousarray(v[..., 0]), np.ascontiguousarray(v[..., 1])


def batch_to_rows(grid_delta_batch, grid_beta_batch):
    """(B, Y, X, S) rotated object batches -> [b][z][x][y] pairs, float32."""
    pair = np.stack([grid_delta_batch, grid_beta_batch], axis=-1)
    return np.ascontiguousarray(pair.transpose(0, 3, 2, 1, 4).astype(np.float32))


def rows_to_batch(rows):
    """[b][z][x][y][2] -> (g_delta, g_beta) each (B, Y, X, S)."""
    v = rows.transpose(0, 3, 2, 1, 4)
    return np.ascontiguousarray(v[..., 0]), np.ascontiguousarray(v[..., 1])
