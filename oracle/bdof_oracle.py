"""CPU oracle: float64 numpy restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``beyond_dof_amd/`` imports this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may use it, and there only as the checker / the timed CPU
baseline, never as the product path.

Parity pinning: every function below that restates an importable reference
function is checked against golden vectors captured from the reference itself
(``tests/golden/make_golden.py`` imports ``/root/reference/cnn_propagator`` with
stubs for its unused third-party imports and stores inputs + outputs as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).  The
gradient (the reference uses HIPS autograd, which is not installed) is pinned by
finite differences of the golden-pinned forward and by torch autograd
(``tests/test_oracle_adjoint.py``).

Reference citations use paths relative to ``/root/reference``.
"""
import numpy as np

PI = 3.1415927  # cnn_propagator/util.py:20, cnn_propagator/np_funcs.py:12 (quirk Q1)


# ---------------------------------------------------------------------------
# Fresnel transfer function                       cnn_propagator/util.py:73-102
# ---------------------------------------------------------------------------
def gen_mesh(max_, shape):
    """cnn_propagator/util.py:73-79 — inclusive symmetric linspace mesh (quirk Q4)."""
    yy = np.linspace(-max_[0], max_[0], shape[0])
    xx = np.linspace(-max_[1], max_[1], shape[1])
    return np.meshgrid(xx, yy)


def get_kernel(dist_nm, lmbda_nm, voxel_nm, grid_shape, pi=PI):
    """cnn_propagator/util.py:82-102 — centred transfer function H, shape (Y, X)."""
    k = 2 * pi / lmbda_nm
    u_max = 1. / (2. * voxel_nm[0])
    v_max = 1. / (2. * voxel_nm[1])
    u, v = gen_mesh([v_max, u_max], grid_shape[0:2])
    return np.exp(1j * k * dist_nm) * np.exp(-1j * pi * lmbda_nm * dist_nm * (u ** 2 + v ** 2))


def get_kernel_ir(dist_nm, lmbda_nm, voxel_nm, grid_shape, pi=PI):
    """cnn_propagator/util.py:105-127 — impulse-response kernel: the real-space Fresnel kernel sampled on the pixel grid
    (origin at -size/2), fft2 + fftshift, times the pixel area.  Dead code on the hot path (np_funcs.py:55 forces the
    transfer-function form); restated for the `kernel='IR'` option."""
    size_nm = np.array(voxel_nm) * np.array(grid_shape)
    k = 2 * pi / lmbda_nm
    ymin, xmin = np.array(size_nm)[:2] / -2.
    dy, dx = voxel_nm[0:2]
    x = np.arange(xmin, xmin + size_nm[1], dx)
    y = np.arange(ymin, ymin + size_nm[0], dy)
    x, y = np.meshgrid(x, y)
    h = np.exp(1j * k * dist_nm) / (1j * lmbda_nm * dist_nm) * np.exp(1j * k / (2 * dist_nm) * (x ** 2 + y ** 2))
    return np.fft.fftshift(np.fft.fft2(h)) * voxel_nm[0] * voxel_nm[1]


def upsample_2x(arr):
    """cnn_propagator/util.py:350-360 — multiscale hand-over: zero-stuff by 2 along the three spatial axes, gaussian
    filter sigma 1 (scipy default mode 'reflect'); a 4-D array is treated channel by channel."""
    from scipy.ndimage import gaussian_filter
    if arr.ndim == 4:
        return np.stack([upsample_2x(arr[..., i]) for i in range(arr.shape[3])], axis=3)
    out = np.zeros([2 * n for n in arr.shape])
    out[::2, ::2, ::2] = arr
    return gaussian_filter(out, 1)


def _propagate(wavefront, h):
    """cnn_propagator/np_funcs.py:42 — fft2 / fftshift / *h / ifftshift / ifft2."""
    return np.fft.ifft2(np.fft.ifftshift(np.fft.fftshift(np.fft.fft2(wavefront), axes=[1, 2]) * h, axes=[1, 2]))


# ---------------------------------------------------------------------------
# Forward multislice                              cnn_propagator/np_funcs.py:15-65
# ---------------------------------------------------------------------------
def multislice_propagate_batch_numpy(grid_delta_batch, grid_beta_batch, probe_real, probe_imag, energy_ev,
                                     psize_cm, free_prop_cm=None, obj_batch_shape=None,
                                     variant='numpy_skip_last', pi=PI, return_probe_array=True, detector_kernel='TF'):
    """Restates cnn_propagator/np_funcs.py:15-65.  detector_kernel='IR': the branch of :58-61 that :55 disables.

    ``variant='numpy_skip_last'`` is the reference numpy behaviour (no propagation
    after the last slice, np_funcs.py:41); ``'tf_all'`` propagates after every
    slice (tensorflow_recon/util.py:465-483, quirk Q3).
    Returns (wavefront[B,Y,X] complex128, probe_array[S,B,Y,X]).
    """
    if obj_batch_shape is None:
        obj_batch_shape = grid_delta_batch.shape
    minibatch_size = obj_batch_shape[0]
    grid_shape = obj_batch_shape[1:]
    voxel_nm = np.array([psize_cm] * 3) * 1.e7
    wavefront = np.zeros([minibatch_size, obj_batch_shape[1], obj_batch_shape[2]], dtype='complex64')
    # in-place add: the probe is ROUNDED to complex64 here exactly as np_funcs.py:20-21 does; the first
    # `wavefront * c` below then promotes everything to complex128 (quirk Q2)
    wavefront += (probe_real + 1j * probe_imag)

    lmbda_nm = 1240. / energy_ev
    size_nm = np.array(grid_shape) * voxel_nm
    n_slice = obj_batch_shape[-1]
    delta_nm = voxel_nm[-1]

    h = get_kernel(delta_nm, lmbda_nm, voxel_nm, grid_shape, pi=pi)
    k = 2. * pi * delta_nm / lmbda_nm

    probe_array = []
    for i in range(n_slice):
        delta_slice = grid_delta_batch[:, :, :, i]
        beta_slice = grid_beta_batch[:, :, :, i]
        c = np.exp(1j * k * delta_slice) * np.exp(-k * beta_slice)
        wavefront = wavefront * c
        if i < n_slice - 1 or variant == 'tf_all':
            wavefront = _propagate(wavefront, h)
        if return_probe_array:
            probe_array.append(wavefront)

    if free_prop_cm is not None:
        if free_prop_cm == 'inf':
            wavefront = np.fft.fftshift(np.fft.fft2(wavefront), axes=[1, 2])
        else:
            dist_nm = free_prop_cm * 1e7
            if detector_kernel == 'TF':                                      # np_funcs.py:55 forces 'TF'
                h = get_kernel(dist_nm, lmbda_nm, voxel_nm, grid_shape, pi=pi)
            else:                                                            # np_funcs.py:58-61
                h = get_kernel_ir(dist_nm, lmbda_nm, voxel_nm, grid_shape, pi=pi)
            wavefront = _propagate(wavefront, h)
    return wavefront, np.array(probe_array)


# ---------------------------------------------------------------------------
# Loss + hand-derived adjoint                      SURVEY.md §3.3
#   loss: cnn_propagator/fullfield.py:106, cnn_propagator/ptychography.py:79
#   gradient: what autograd.grad(calculate_loss,[0,1]) returns
#             (cnn_propagator/fullfield.py:329,345) for the FFT forward above
# ---------------------------------------------------------------------------
def multislice_loss_and_grad(grid_delta_batch, grid_beta_batch, probe_real, probe_imag, energy_ev, psize_cm,
                             meas_abs, free_prop_cm=None, variant='numpy_skip_last', pi=PI,
                             return_probe_grad=False, detector_kernel='TF'):
    """loss = mean((|d| - meas_abs)**2) over (B,Y,X) and its gradient w.r.t. the
    (already rotated / windowed) delta and beta batches, shape [B,Y,X,S]."""
    B, Y, X, S = grid_delta_batch.shape
    voxel_nm = np.array([psize_cm] * 3) * 1.e7
    lmbda_nm = 1240. / energy_ev
    delta_nm = voxel_nm[-1]
    h = get_kernel(delta_nm, lmbda_nm, voxel_nm, (Y, X, S), pi=pi)
    k = 2. * pi * delta_nm / lmbda_nm

    psi = np.zeros([B, Y, X], dtype=np.complex64)
    psi += (probe_real + 1j * probe_imag)          # complex64 rounding of the probe, np_funcs.py:20-21
    psi = psi.astype(np.complex128)
    phis, cs = [], []
    for i in range(S):
        c = np.exp(1j * k * grid_delta_batch[..., i]) * np.exp(-k * grid_beta_batch[..., i])
        phi = psi * c
        phis.append(phi)
        cs.append(c)
        psi = _propagate(phi, h) if (i < S - 1 or variant == 'tf_all') else phi

    if free_prop_cm is None:
        d = psi
    elif free_prop_cm == 'inf':
        d = np.fft.fftshift(np.fft.fft2(psi), axes=[1, 2])
    else:
        hd = (get_kernel if detector_kernel == 'TF' else get_kernel_ir)(free_prop_cm * 1e7, lmbda_nm, voxel_nm, (Y, X, S), pi=pi)
        d = _propagate(psi, hd)

    absd = np.abs(d)
    resid = absd - meas_abs
    loss = np.mean(resid ** 2)
    with np.errstate(divide='ignore', invalid='ignore'):
        unit = np.where(absd > 0, d / absd, 0)
    G = 2.0 * resid * unit / (B * Y * X)           # G(d) = dL/dRe + i dL/dIm

    def prop_adj(G, hh):
        # P = F^-1 diag(ifftshift hh) F: P^H = F^-1 diag(conj hh) F for any multiplier (unitary when |hh| = 1)
        return np.fft.ifft2(np.fft.ifftshift(np.fft.fftshift(np.fft.fft2(G), axes=[1, 2]) * np.conj(hh), axes=[1, 2]))

    if free_prop_cm is None:
        pass
    elif free_prop_cm == 'inf':
        G = (Y * X) * np.fft.ifft2(np.fft.ifftshift(G, axes=[1, 2]))
    else:
        G = prop_adj(G, hd)

    g_delta = np.zeros((B, Y, X, S))
    g_beta = np.zeros((B, Y, X, S))
    for i in range(S - 1, -1, -1):
        if i < S - 1 or variant == 'tf_all':
            G = prop_adj(G, h)
        t = np.conj(phis[i]) * G
        g_delta[..., i] = k * t.imag
        g_beta[..., i] = -k * t.real
        G = np.conj(cs[i]) * G
    if return_probe_grad:
        return loss, g_delta, g_beta, G
    return loss, g_delta, g_beta


# ---------------------------------------------------------------------------
# Rotation lookup tables + gather                  cnn_propagator/util.py:294-402
# ---------------------------------------------------------------------------
def rotation_lookup(array_size, n_theta):
    """cnn_propagator/util.py:294-332 without the disk I/O: list over theta of
    (X*Z, 2) integer source coordinates (coord1_old, coord2_old).  theta_j =
    linspace(0, 2*pi, n_theta)[j] (quirk Q5), centre floor(n/2), round half-even, clip."""
    image_center = [np.floor(x / 2) for x in array_size]
    coord1 = np.arange(array_size[1])
    coord2 = np.arange(array_size[2])
    coord2_vec = np.tile(coord2, array_size[1])
    coord1_vec = np.tile(coord1, array_size[2])
    coord1_vec = np.reshape(coord1_vec, [array_size[1], array_size[2]])
    coord1_vec = np.reshape(np.transpose(coord1_vec), [-1])
    coord1_vec = coord1_vec - image_center[1]
    coord2_vec = coord2_vec - image_center[2]
    coord_new = np.stack([coord1_vec, coord2_vec]).astype(np.float32)
    theta_ls = np.linspace(0, 2 * np.pi, n_theta)
    coord_old_ls = []
    for theta in theta_ls:
        m_rot = np.array([[np.cos(theta), -np.sin(theta)],
                          [np.sin(theta), np.cos(theta)]])
        coord_old = np.matmul(m_rot, coord_new)
        coord1_old = np.round(coord_old[0, :] + image_center[1]).astype(int)
        coord2_old = np.round(coord_old[1, :] + image_center[2]).astype(int)
        coord1_old = np.clip(coord1_old, 0, array_size[1] - 1)
        coord2_old = np.clip(coord2_old, 0, array_size[2] - 1)
        coord_old_ls.append(np.stack([coord1_old, coord2_old], axis=1))
    return coord_old_ls


def apply_rotation(obj, coord_old):
    """cnn_propagator/util.py:377-402: rot[i0,i1,i2,c] = obj[i0, c1(i1,i2), c2(i1,i2), c]."""
    s = obj.shape
    c1 = coord_old[:, 0].reshape(s[1], s[2])
    c2 = coord_old[:, 1].reshape(s[1], s[2])
    return obj[:, c1, c2, ...]


def apply_rotation_adjoint(g_rot, coord_old):
    """Transpose of apply_rotation: scatter-add g_rot[i0,i1,i2,...] into [i0,c1,c2,...]."""
    s = g_rot.shape
    out = np.zeros_like(g_rot)
    flat = coord_old[:, 0] * s[2] + coord_old[:, 1]
    src = g_rot.reshape(s[0], s[1] * s[2], -1)
    dst = out.reshape(s[0], s[1] * s[2], -1)
    for i0 in range(s[0]):
        np.add.at(dst[i0], flat, src[i0])
    return out


# ---------------------------------------------------------------------------
# Regularisers, optimiser, scheduler              cnn_propagator/util.py, fullfield.py
# ---------------------------------------------------------------------------
def total_variation_3d(arr):
    """cnn_propagator/util.py:61-70 — periodic anisotropic TV."""
    res = np.sum(np.abs(np.roll(arr, 1, axis=0) - arr))
    res = res + np.sum(np.abs(np.roll(arr, 1, axis=1) - arr))
    res = res + np.sum(np.abs(np.roll(arr, 1, axis=2) - arr))
    return res


def total_variation_3d_grad(arr):
    """d total_variation_3d / d arr with d|x|/dx = sign(x) (autograd's convention, sign(0)=0)."""
    g = np.zeros_like(arr)
    for ax in range(3):
        s = np.sign(np.roll(arr, 1, axis=ax) - arr)     # term j: |a[j-1] - a[j]|
        g += -s + np.roll(s, -1, axis=ax)
    return g


def regularizer(obj_delta, obj_beta, alpha=1e-7, alpha_d=None, alpha_b=None, gamma=1e-6):
    """cnn_propagator/fullfield.py:109-118 (incl. quirk Q6: the alpha branch counts delta twice)."""
    if alpha_d is None:
        return alpha * (np.sum(np.abs(obj_delta)) + np.sum(np.abs(obj_delta))) + gamma * total_variation_3d(obj_delta)
    if gamma == 0:
        return alpha_d * np.sum(np.abs(obj_delta)) + alpha_b * np.sum(np.abs(obj_beta))
    return alpha_d * np.sum(np.abs(obj_delta)) + alpha_b * np.sum(np.abs(obj_beta)) + gamma * total_variation_3d(obj_delta)


def regularizer_grad(obj_delta, obj_beta, alpha=1e-7, alpha_d=None, alpha_b=None, gamma=1e-6):
    if alpha_d is None:
        gd = 2 * alpha * np.sign(obj_delta) + gamma * total_variation_3d_grad(obj_delta)
        return gd, np.zeros_like(obj_beta)
    gd = alpha_d * np.sign(obj_delta)
    gb = alpha_b * np.sign(obj_beta)
    if gamma != 0:
        gd = gd + gamma * total_variation_3d_grad(obj_delta)
    return gd, gb


def apply_gradient_adam(x, g, i_batch, m=None, v=None, step_size=0.001, b1=0.9, b2=0.999, eps=1e-8):
    """cnn_propagator/util.py:280-291 (bias-correction exponent = i_batch + 1, quirk Q10)."""
    g = np.array(g)
    if m is None or v is None:
        m = np.zeros_like(x)
        v = np.zeros_like(x)
    m = (1 - b1) * g + b1 * m
    v = (1 - b2) * (g ** 2) + b2 * v
    mhat = m / (1 - b1 ** (i_batch + 1))
    vhat = v / (1 - b2 ** (i_batch + 1))
    x = x - step_size * mhat / (np.sqrt(vhat) + eps)
    return x, m, v


def split_tasks(arr, split_size):
    """cnn_propagator/util.py:271-277."""
    res = []
    ind = 0
    while ind < len(arr):
        res.append(arr[ind:min(ind + split_size, len(arr))])
        ind += split_size
    return res


# ---------------------------------------------------------------------------
# Full-field loss / gradient of one minibatch      cnn_propagator/fullfield.py:93-121
# (FFT forward of np_funcs.py in place of the conv forward, as north_star asks)
# ---------------------------------------------------------------------------
def fullfield_loss_and_grad(obj_delta, obj_beta, coord_ls, this_ind_batch, this_prj_batch, probe_real, probe_imag,
                            energy_ev, psize_cm, free_prop_cm=None, alpha=1e-7, alpha_d=None, alpha_b=None,
                            gamma=1e-6, variant='numpy_skip_last', pi=PI, with_reg=True):
    obj_stack = np.stack([obj_delta, obj_beta], axis=3)
    rot = np.stack([apply_rotation(obj_stack, coord_ls[j]) for j in this_ind_batch])
    loss, gd_rot, gb_rot = multislice_loss_and_grad(rot[..., 0], rot[..., 1], probe_real, probe_imag, energy_ev,
                                                    psize_cm, np.abs(this_prj_batch), free_prop_cm, variant, pi)
    gd = np.zeros_like(obj_delta)
    gb = np.zeros_like(obj_beta)
    for b, j in enumerate(this_ind_batch):
        gd += apply_rotation_adjoint(gd_rot[b], coord_ls[j])
        gb += apply_rotation_adjoint(gb_rot[b], coord_ls[j])
    if with_reg:
        loss = loss + regularizer(obj_delta, obj_beta, alpha, alpha_d, alpha_b, gamma)
        rd, rb = regularizer_grad(obj_delta, obj_beta, alpha, alpha_d, alpha_b, gamma)
        gd, gb = gd + rd, gb + rb
    return loss, gd, gb


# ---------------------------------------------------------------------------
# Ptychography loss / gradient of one minibatch    cnn_propagator/ptychography.py:30-81
# ---------------------------------------------------------------------------
def ptycho_pad_amounts(probe_pos, probe_size, obj_size):
    """cnn_propagator/ptychography.py:44-59 (incl. the probe_size_half[0] slip at :57, quirk Q14)."""
    probe_pos = np.asarray(probe_pos)
    half = (np.array(probe_size) / 2).astype('int')
    pad = np.array([[0, 0], [0, 0]])
    if probe_pos[:, 0].min() - half[0] < 0:
        pad[0, 0] = half[0] - probe_pos[:, 0].min()
    if probe_pos[:, 0].max() + half[0] > obj_size[0]:
        pad[0, 1] = probe_pos[:, 0].max() + half[0] - obj_size[0]
    if probe_pos[:, 1].min() - half[1] < 0:
        pad[1, 0] = half[1] - probe_pos[:, 1].min()
    if probe_pos[:, 1].max() + half[1] > obj_size[1]:
        pad[1, 1] = probe_pos[:, 1].max() + half[0] - obj_size[1]
    return pad, half


def ptycho_loss_and_grad(obj_delta, obj_beta, coord_old, probe_pos_all, this_pos_batch, this_prj_batch,
                         probe_real, probe_imag, probe_size, energy_ev, psize_cm, variant='numpy_skip_last', pi=PI,
                         propagator='fft', kernel_size=17):
    """Loss of cnn_propagator/ptychography.py:30-81, far field ('inf'), and its gradient w.r.t. the un-rotated (delta, beta).
    propagator='fft': the transfer-function forward of np_funcs.py (the north-star path); 'conv': the real-space propagator the
    reference function literally calls (:74-76, kernel_size 17) — the form golden vector G11 pins."""
    obj_size = obj_delta.shape
    obj_stack = np.stack([obj_delta, obj_beta], axis=3)
    obj_rot = apply_rotation(obj_stack, coord_old)
    pad, half = ptycho_pad_amounts(probe_pos_all, probe_size, obj_size)
    obj_pad = np.pad(obj_rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
    subs = []
    for pos in this_pos_batch:
        p0 = int(pos[0]) + pad[0, 0]
        p1 = int(pos[1]) + pad[1, 0]
        subs.append(obj_pad[p0 - half[0]:p0 - half[0] + probe_size[0], p1 - half[1]:p1 - half[1] + probe_size[1]])
    subs = np.stack(subs)
    if propagator == 'conv':
        loss, gd_sub, gb_sub = cnn_loss_and_grad(subs[..., 0], subs[..., 1], probe_real, probe_imag, energy_ev,
                                                 [psize_cm] * 3 if np.isscalar(psize_cm) else psize_cm, np.abs(this_prj_batch),
                                                 kernel_size=kernel_size, free_prop_cm='inf')
    else:
        loss, gd_sub, gb_sub = multislice_loss_and_grad(subs[..., 0], subs[..., 1], probe_real, probe_imag, energy_ev,
                                                        psize_cm, np.abs(this_prj_batch), 'inf', variant, pi)
    g_pad = np.zeros(obj_pad.shape)
    for b, pos in enumerate(this_pos_batch):
        p0 = int(pos[0]) + pad[0, 0]
        p1 = int(pos[1]) + pad[1, 0]
        g_pad[p0 - half[0]:p0 - half[0] + probe_size[0], p1 - half[1]:p1 - half[1] + probe_size[1], :, 0] += gd_sub[b]
        g_pad[p0 - half[0]:p0 - half[0] + probe_size[0], p1 - half[1]:p1 - half[1] + probe_size[1], :, 1] += gb_sub[b]
    g_rot = g_pad[pad[0, 0]:pad[0, 0] + obj_size[0], pad[1, 0]:pad[1, 0] + obj_size[1]]
    g = apply_rotation_adjoint(g_rot, coord_old)
    return loss, g[..., 0], g[..., 1]


# ---------------------------------------------------------------------------
# Probe construction                               cnn_propagator/fullfield.py:299-310
# ---------------------------------------------------------------------------
def gaussian_probe(shape, mag_sigma, phase_sigma, phase_max):
    py = np.arange(shape[0]) - (shape[0] - 1.) / 2
    px = np.arange(shape[1]) - (shape[1] - 1.) / 2
    pxx, pyy = np.meshgrid(px, py)
    mag = np.exp(-(pxx ** 2 + pyy ** 2) / (2 * mag_sigma ** 2))
    phase = phase_max * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * phase_sigma ** 2))
    a = mag * np.exp(1j * phase)            # util.py:265-268 mag_phase_to_real_imag
    return a.real, a.imag


# ---------------------------------------------------------------------------
# Real-space truncated-kernel propagator            cnn_propagator/propagation.py:18-133   
# Pinned by golden vector G9 (tests/golden/make_golden.py --g9): the reference's own function executed from its file, with
# its one absent third-party primitive — HIPS autograd's autograd.scipy.signal.convolve, un-versioned in the reference —
# stood in by scipy.signal.convolve2d for the single call form it uses ('valid', axes=([1, 2], [0, 1])).  The restatement
# below reproduces those outputs to 1e-12; what stays unpinned is that primitive's definition (a true convolution).
# ---------------------------------------------------------------------------
def conv_kernel_2d(delta_nm, lmbda_nm, voxel_nm, grid_shape, kernel_size):
    """propagation.py:35-44: H on the (Y-1, X-1) mesh -> real space -> centre crop of kernel_size^2."""
    kernel = get_kernel(delta_nm, lmbda_nm, voxel_nm, np.array(grid_shape) - 1)
    kernel = np.fft.fftshift(np.fft.ifft2(np.fft.ifftshift(kernel)))
    kernel_mid = ((np.array(kernel.shape) - 1) / 2).astype('int')
    half = int((kernel_size - 1) / 2)
    return kernel[kernel_mid[0] - half:kernel_mid[0] + half + 1, kernel_mid[1] - half:kernel_mid[1] + half + 1]


def _conv_valid(field, kernel):
    """True 2-D convolution, mode='valid', per batch element (autograd.scipy.signal.convolve with
    axes=([1, 2], [0, 1]), propagation.py:93)."""
    from scipy.signal import convolve2d
    return np.stack([convolve2d(f, kernel, mode='valid') for f in field])


def multislice_propagate_cnn(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, kernel_size=17,
                             free_prop_cm=None, return_tape=False):
    """Restates cnn_propagator/propagation.py:18-133.  psize_cm is a 3-vector there (fullfield.py:89); k uses np.pi
    (propagation.py:25, quirk Q1) while the kernel uses PI = 3.1415927 through get_kernel."""
    assert kernel_size % 2 == 1
    n_batch, shape_y, shape_x, n_slice = grid_delta.shape
    lmbda_nm = 1240. / energy_ev
    voxel_nm = np.array(psize_cm) * 1.e7 if np.ndim(psize_cm) else np.array([psize_cm] * 3) * 1.e7
    delta_nm = voxel_nm[-1]
    k = 2. * np.pi * delta_nm / lmbda_nm
    grid_shape = np.array(grid_delta.shape[1:])
    kernel = conv_kernel_2d(delta_nm, lmbda_nm, voxel_nm, grid_shape, kernel_size)
    pad_len = (kernel_size - 1) // 2
    probe = np.tile(probe_real + 1j * probe_imag, [n_batch, 1, 1]).astype(np.complex128)
    edge_val = 1.0
    initial_int = probe[0, 0, 0]
    tape = []
    for i_slice in range(n_slice):
        c = np.exp(1j * k * grid_delta[:, :, :, i_slice] - k * grid_beta[:, :, :, i_slice])
        probe = probe * c
        tape.append((probe, c))
        padded = np.pad(probe, [[0, 0], [pad_len, pad_len], [pad_len, pad_len]], mode='constant', constant_values=edge_val)
        probe = _conv_valid(padded, kernel)
        edge_val = np.sum(kernel.flatten() * edge_val)
    final_int = probe[0, 0, 0]
    pre_norm = probe
    probe = probe * (initial_int / final_int)
    if free_prop_cm is not None:
        if free_prop_cm == 'inf':
            probe = np.fft.fftshift(np.fft.fft2(probe), axes=[1, 2])
        else:
            h = get_kernel(free_prop_cm * 1e7, lmbda_nm, voxel_nm, grid_shape)
            probe = np.fft.ifft2(np.fft.ifftshift(np.fft.fftshift(np.fft.fft2(probe), axes=[1, 2]) * h, axes=[1, 2]))
    if return_tape:
        return probe, tape, kernel, pre_norm, initial_int, k
    return probe


def cnn_loss_and_grad(grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, meas_abs, kernel_size=17,
                      free_prop_cm=None):
    """loss = mean((|d| - meas_abs)^2) for the conv forward and its gradient w.r.t. (delta, beta) — what
    autograd.grad(calculate_loss) differentiates in cnn_propagator/fullfield.py:102-106,329 (hand-derived; checked
    against finite differences in tests/test_oracle_adjoint.py)."""
    from scipy.signal import correlate2d
    d, tape, kernel, pre_norm, initial_int, k = multislice_propagate_cnn(
        grid_delta, grid_beta, probe_real, probe_imag, energy_ev, psize_cm, kernel_size, free_prop_cm, return_tape=True)
    B, Y, X, S = grid_delta.shape
    lmbda_nm = 1240. / energy_ev
    voxel_nm = np.array(psize_cm) * 1.e7 if np.ndim(psize_cm) else np.array([psize_cm] * 3) * 1.e7
    absd = np.abs(d)
    resid = absd - meas_abs
    loss = np.mean(resid ** 2)
    with np.errstate(divide='ignore', invalid='ignore'):
        unit = np.where(absd > 0, d / absd, 0)
    G = 2.0 * resid * unit / (B * Y * X)                       # G(d)
    if free_prop_cm is None:
        Gq = G
    elif free_prop_cm == 'inf':
        Gq = (Y * X) * np.fft.ifft2(np.fft.ifftshift(G, axes=[1, 2]))
    else:
        h = get_kernel(free_prop_cm * 1e7, lmbda_nm, voxel_nm, (Y, X, S))
        Gq = np.fft.ifft2(np.fft.ifftshift(np.fft.fftshift(np.fft.fft2(G), axes=[1, 2]) * np.conj(h), axes=[1, 2]))
    # q = s * P,  s = initial_int / P[0,0,0]  (propagation.py:109-110; differentiable through the corner pixel)
    P000 = pre_norm[0, 0, 0]
    s = initial_int / P000
    q = s * pre_norm
    Gp = np.conj(s) * Gq
    Gp[0, 0, 0] += -np.sum(Gq * np.conj(q)) / np.conj(P000)
    g_delta = np.zeros((B, Y, X, S))
    g_beta = np.zeros((B, Y, X, S))
    for i in range(S - 1, -1, -1):
        phi, c = tape[i]
        # adjoint of (constant pad, valid convolution): full correlation with conj(kernel), cropped to the field
        Gphi = np.stack([correlate2d(g, kernel, mode='same') for g in Gp])
        t = np.conj(phi) * Gphi
        g_delta[..., i] = k * t.imag
        g_beta[..., i] = -k * t.real
        Gp = np.conj(c) * Gphi
    return loss, g_delta, g_beta


# ---------------------------------------------------------------------------
# Tiled ("pfft") propagation                       /root/reference/README.md:1-11
#   The reference's tiled propagator lives on a branch that is not in the checkout: PARITY UNPINNED by reference code.
#   What is stated here is the algorithm of the paper's title — overlapping tiles, periodic tile FFTs, cores stitched back
#   before the wrap-around crosses the halo — in float64, so that (a) its deviation from the whole-field propagator above
#   (which IS golden-pinned) can be measured on the CPU, and (b) the device implementation can be checked against the same
#   algorithm to float32 round-off.
# ---------------------------------------------------------------------------
def get_kernel_tile(dist_nm, lmbda_nm, voxel_nm, tile_shape, field_shape, pi=PI):
    """H of the (FY, FX) field's propagator (get_kernel above, mesh quirk Q4 of the FIELD) at the FFT frequencies of a tile."""
    def axis(t, n, vox):
        u_max = 1. / (2. * vox)
        f = (np.arange(t) - t // 2) / (t * vox)
        return -u_max + (f * n * vox + n // 2) * 2. * u_max / (n - 1)
    v = axis(tile_shape[0], field_shape[0], voxel_nm[1])
    u = axis(tile_shape[1], field_shape[1], voxel_nm[0])
    uu, vv = np.meshgrid(u, v)
    return np.exp(1j * (2 * pi / lmbda_nm) * dist_nm) * np.exp(-1j * pi * lmbda_nm * dist_nm * (uu ** 2 + vv ** 2))


def tile_origins(field_n, tile, halo):
    core = tile - 2 * halo
    return [i * core - halo for i in range(-(-field_n // core))]


def tiled_multislice_propagate(grid_delta, grid_beta, probe, energy_ev, psize_cm, tile, halo, slices_per_exchange,
                               taper=0, variant='numpy_skip_last', pi=PI):
    """Exit wave (FY, FX) of `probe` (FY, FX) through the object (FY, FX, S) by tiled propagation.  Every
    `slices_per_exchange` slices: cut periodic (tile x tile) windows at tile_origins, run the slices on each window with
    its own periodic FFT and the field's transfer function, write the cores back.  The outermost `taper` pixels of every
    window are ramped to zero first (raised cosine), so that the window's periodic boundary carries no jump."""
    fy, fx, n_slice = grid_delta.shape
    voxel_nm = np.array([psize_cm] * 3) * 1.e7
    lmbda_nm = 1240. / energy_ev
    delta_nm = voxel_nm[-1]
    k = 2. * pi * delta_nm / lmbda_nm
    h = np.fft.ifftshift(get_kernel_tile(delta_nm, lmbda_nm, voxel_nm, (tile, tile), (fy, fx), pi=pi))
    field = np.array(probe, dtype=np.complex128)
    w1 = np.ones(tile)
    if taper > 0:
        ramp = 0.5 - 0.5 * np.cos(np.pi * (np.arange(taper) + 0.5) / taper)
        w1[:taper] = ramp
        w1[tile - taper:] = ramp[::-1]
    win = w1[:, None] * w1[None, :]
    oy, ox = tile_origins(fy, tile, halo), tile_origins(fx, tile, halo)
    core = tile - 2 * halo
    for z0 in range(0, n_slice, slices_per_exchange):
        nz = min(slices_per_exchange, n_slice - z0)
        new = np.empty_like(field)
        for y0 in oy:
            iy = np.arange(y0, y0 + tile) % fy
            for x0 in ox:
                ix = np.arange(x0, x0 + tile) % fx
                w = field[np.ix_(iy, ix)] * win
                # the object is NOT periodic: outside the volume the tile sees vacuum (what the device window does)
                inside = ((np.arange(y0, y0 + tile) >= 0) & (np.arange(y0, y0 + tile) < fy))[:, None] & \
                         ((np.arange(x0, x0 + tile) >= 0) & (np.arange(x0, x0 + tile) < fx))[None, :]
                for z in range(z0, z0 + nz):
                    d = np.where(inside, grid_delta[np.ix_(iy, ix)][..., z], 0.0)
                    b = np.where(inside, grid_beta[np.ix_(iy, ix)][..., z], 0.0)
                    w = w * (np.exp(1j * k * d) * np.exp(-k * b))
                    if z < n_slice - 1 or variant == 'tf_all':
                        w = np.fft.ifft2(np.fft.fft2(w) * h)
                # cores tile the field from 0; the part of a last core that overhangs the field's edge is dropped
                ny_c, nx_c = min(core, fy - (y0 + halo)), min(core, fx - (x0 + halo))
                new[y0 + halo:y0 + halo + ny_c, x0 + halo:x0 + halo + nx_c] = w[halo:halo + ny_c, halo:halo + nx_c]
        field = new
    return field


# ---------------------------------------------------------------------------
# Bilinear rotation of the TF twin                  tensorflow_recon/fullfield.py:96
#   tf_rotate = tf.contrib.image.rotate (TensorFlow 1.x, un-vendored and absent here: no version is pinned anywhere in the
#   reference; PARITY UNPINNED by execution).  Restated from its published definition: the (Y, X, Z, C) object is a batch of
#   NHWC images of height H = X and width W = Z; angles_to_projective_transforms gives, for angle t,
#       x_offset = ((W-1) - (cos t (W-1) - sin t (H-1))) / 2 ,  y_offset = ((H-1) - (sin t (W-1) + cos t (H-1))) / 2
#   and the OUTPUT pixel (h, w) takes the input at (h', w') = (sin t w + cos t h + y_offset, cos t w - sin t h + x_offset),
#   bilinearly, every tap outside the image reading 0.  Cross-checked against scipy.ndimage.map_coordinates(order=1).
# ---------------------------------------------------------------------------
def rotate_bilinear_params(theta, height, width):
    c, s = np.cos(theta), np.sin(theta)
    x_off = ((width - 1) - (c * (width - 1) - s * (height - 1))) / 2.0
    y_off = ((height - 1) - (s * (width - 1) + c * (height - 1))) / 2.0
    return c, s, x_off, y_off


def _bilinear_taps(theta, height, width):
    c, s, x_off, y_off = rotate_bilinear_params(theta, height, width)
    hh, ww = np.mgrid[:height, :width].astype(np.float64)
    sw = c * ww - s * hh + x_off
    sh = s * ww + c * hh + y_off
    fw, fh = np.floor(sw), np.floor(sh)
    aw, ah = sw - fw, sh - fh
    taps = []
    for dh, dw, wt in ((0, 0, (1 - ah) * (1 - aw)), (0, 1, (1 - ah) * aw), (1, 0, ah * (1 - aw)), (1, 1, ah * aw)):
        h2, w2 = fh.astype(int) + dh, fw.astype(int) + dw
        inside = (h2 >= 0) & (h2 < height) & (w2 >= 0) & (w2 < width)
        taps.append((np.where(inside, h2, 0), np.where(inside, w2, 0), np.where(inside, wt, 0.0)))
    return taps


def rotate_bilinear(obj, theta):
    """obj (Y, X, Z, C) -> rotated (Y, X, Z, C)."""
    Y, H, W = obj.shape[:3]
    out = np.zeros_like(obj, dtype=np.float64)
    for h2, w2, wt in _bilinear_taps(theta, H, W):
        out += wt[None, :, :, None] * obj[:, h2, w2]
    return out


def rotate_bilinear_adjoint(g_rot, theta):
    """Adjoint of rotate_bilinear: scatter-add of the rotated-frame gradient with the same weights."""
    Y, H, W = g_rot.shape[:3]
    out = np.zeros_like(g_rot, dtype=np.float64)
    for h2, w2, wt in _bilinear_taps(theta, H, W):
        contrib = wt[None, :, :, None] * g_rot
        np.add.at(out, (slice(None), h2, w2), contrib)
    return out
