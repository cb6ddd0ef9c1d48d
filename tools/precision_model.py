"""Development aid (CPU): which float32 quantity limits the delta-gradient of far-field ptychography?

The G17 configuration (64 x 64 gaussian probe with phase, 64 slices, far field, 2 % noise on the data) is run through a numpy
model of the DEVICE algorithm — carrier field p_z in float64 rounded once to float32, scattered wave eps through the
transforms, modulation factors c - 1, seed, adjoint sweep — with one ingredient at a time held at float32 and everything else
at float64.  Prints the relative error of g_delta / g_beta against the all-float64 run (= oracle.multislice_loss_and_grad).
"""
import os
import sys

import numpy as np
import scipy.fft as sfft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bdof_oracle as orc  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def c64(a):
    return a.astype(np.complex64).astype(np.complex128)


def f32(a):
    return a.astype(np.float32).astype(np.float64)


def model(delta, beta, probe, meas, k, h, opt):
    """delta, beta (B, Y, X, S); probe (Y, X) complex; h: un-shifted transfer function (Y, X) complex128, |h| = 1.
    opt: set of strings naming what is held at float32:
      'h'       transfer function rounded to complex64
      'cm1'     modulation factors c - 1 rounded
      'pz'      carrier field planes rounded
      'fwd_fft' forward sweep's transforms in complex64 (scipy single precision)
      'adj_fft' adjoint sweep's transforms in complex64
      'seed'    detector wave, residual and seed in float32
      'tape'    stored phi_z rounded to complex64
      'grad'    the products conj(phi) G in complex64
    """
    B, Y, X, S = delta.shape
    hh = c64(h) if 'h' in opt else h

    def prop(f, conj=False, single=False):
        if single:
            f = f.astype(np.complex64)
            hq = (np.conj(hh) if conj else hh).astype(np.complex64)
            return sfft.ifft2(sfft.fft2(f, axes=(1, 2)) * hq, axes=(1, 2)).astype(np.complex128)
        return np.fft.ifft2(np.fft.fft2(f, axes=(1, 2)) * (np.conj(hh) if conj else hh), axes=(1, 2))

    # carrier field: float64 propagation of the probe, one rounding per plane
    p = np.zeros((Y, X), dtype=np.complex64)
    p += probe
    p = p.astype(np.complex128)
    pz = []
    for z in range(S):
        pz.append(c64(p) if 'pz' in opt else p)
        if z < S - 1:
            p = np.fft.ifft2(np.fft.fft2(p) * h)         # the carrier is always propagated with the exact h in float64
    pdet = np.fft.fft2(p)
    if 'pz' in opt:
        pdet = c64(pdet)
    eps = np.zeros((B, Y, X), dtype=np.complex128)
    phis, cm1s = [], []
    for z in range(S):
        x = k * delta[..., z]
        y = -k * beta[..., z]
        cm1 = np.expm1(y) * np.cos(x) + (np.cos(x) - 1.0) + 1j * np.exp(y) * np.sin(x)
        if 'cm1' in opt:
            cm1 = c64(cm1)
        epsp = eps + cm1 * (pz[z] + eps)                  # phi_z = p_z + eps'
        if 'fwd_fft' in opt:
            epsp = c64(epsp)
        phi = pz[z] + epsp
        phis.append(c64(phi) if 'tape' in opt else phi)
        cm1s.append(cm1)
        eps = prop(epsp, single='fwd_fft' in opt) if z < S - 1 else epsp
    if 'fwd_fft' in opt:
        ed = sfft.fft2(eps.astype(np.complex64), axes=(1, 2)).astype(np.complex128)
    else:
        ed = np.fft.fft2(eps, axes=(1, 2))
    d = pdet + ed
    m = np.fft.ifftshift(meas, axes=(1, 2))
    if 'seed' in opt:
        d32 = d.astype(np.complex64)
        ab = np.abs(d32).astype(np.float32)
        r = ab - m.astype(np.float32)
        seed = ((np.float32(2.0 / (B * Y * X)) * r / ab) * d32).astype(np.complex128)
        loss = float(np.mean(r.astype(np.float64) ** 2))
    else:
        ab = np.abs(d)
        r = ab - m
        seed = 2.0 * r * d / ab / (B * Y * X)
        loss = float(np.mean(r ** 2))
    if 'adj_fft' in opt:
        G = (sfft.ifft2(seed.astype(np.complex64), axes=(1, 2)) * np.float32(Y * X)).astype(np.complex128)
    else:
        G = np.fft.ifft2(seed, axes=(1, 2)) * (Y * X)
    gd = np.zeros((B, Y, X, S))
    gb = np.zeros((B, Y, X, S))
    for z in range(S - 1, -1, -1):
        if z < S - 1:
            G = prop(G, conj=True, single='adj_fft' in opt)
        if 'grad' in opt:
            t = (np.conj(phis[z]).astype(np.complex64) * G.astype(np.complex64)).astype(np.complex128)
        else:
            t = np.conj(phis[z]) * G
        gd[..., z] = k * t.imag
        gb[..., z] = -k * t.real
        G = np.conj(1.0 + cm1s[z]) * G
        if 'adj_fft' in opt:
            G = c64(G)
    return loss, gd, gb


def main():
    gdir = os.path.join(ROOT, 'tests', 'golden')
    sys.path.insert(0, gdir)
    import g13_inputs
    g = np.load(os.path.join(gdir, 'g17_reconstruct_ptychography_fft_64.npz'))
    obj_size, psz, sigma = tuple(int(v) for v in g['obj_size']), tuple(int(v) for v in g['probe_size']), float(g['probe_sigma'])
    init_d, init_b = g13_inputs.initial_guess(obj_size)
    pos = g['probe_pos']
    coords = orc.rotation_lookup(list(obj_size), 2)
    pr, pi_ = orc.gaussian_probe(psz, sigma, sigma, 0.5)
    # the windows of angle 0, all four positions
    rot = orc.apply_rotation(np.stack([init_d, init_b], axis=3), coords[0])
    pad, half = orc.ptycho_pad_amounts(pos, psz, obj_size)
    obj_pad = np.pad(rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
    subs = np.stack([obj_pad[p[0] + pad[0, 0] - half[0]:p[0] + pad[0, 0] - half[0] + psz[0],
                             p[1] + pad[1, 0] - half[1]:p[1] + pad[1, 0] - half[1] + psz[1]] for p in pos])
    meas = np.abs(g['prj'][0]).astype(np.float64)
    delta, beta = subs[..., 0], subs[..., 1]
    voxel_nm = np.array([1e-7] * 3) * 1e7
    lmbda_nm = 1240. / 5000.
    h = np.fft.ifftshift(orc.get_kernel(voxel_nm[-1], lmbda_nm, voxel_nm, psz))
    k = 2. * orc.PI * voxel_nm[-1] / lmbda_nm
    rl, rgd, rgb = orc.multislice_loss_and_grad(delta, beta, pr, pi_, 5000., 1e-7, meas, 'inf')
    l0, gd0, gb0 = model(delta, beta, pr + 1j * pi_, meas, k, h, set())
    print('model(float64) vs oracle: loss {:.2e} gd {:.2e} gb {:.2e}'.format(abs(l0 - rl) / rl, rel(gd0, rgd), rel(gb0, rgb)))
    print('|g_delta| rms {:.3e}  |g_beta| rms {:.3e}'.format(np.sqrt(np.mean(rgd ** 2)), np.sqrt(np.mean(rgb ** 2))))
    for opt in (['h'], ['cm1'], ['pz'], ['tape'], ['seed'], ['grad'], ['fwd_fft'], ['adj_fft'],
                ['h', 'cm1', 'pz', 'tape', 'seed', 'grad', 'fwd_fft', 'adj_fft'],
                ['h', 'cm1', 'pz', 'tape', 'grad', 'fwd_fft']):
        l1, gd, gb = model(delta, beta, pr + 1j * pi_, meas, k, h, set(opt))
        print('{:60s} loss {:.2e}  g_delta {:.2e}  g_beta {:.2e}'.format('+'.join(opt), abs(l1 - rl) / rl, rel(gd, rgd), rel(gb, rgb)))


if __name__ == '__main__':
    main()
