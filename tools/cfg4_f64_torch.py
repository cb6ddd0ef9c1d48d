"""cfg4 error budget, float64 side (TEST INFRASTRUCTURE, not product): the whole-field propagation of np_funcs.py:36-43 and the
tiled algorithm of oracle.tiled_multislice_propagate, both in complex128 with torch on the GPU (the host needs 393 s for the
1024 slices of a 4096^2 field; hipFFT in double needs seconds).  Validated against the host's scipy float64 run at 96 slices
(`check`).  Writes the whole-field exit wave to a .npy for the float32 runs to be compared with, and prints the TILING error
of the float64 tiled algorithm for every (tile, halo, taper, interval) given — the part of cfg4's error no arithmetic removes.
usage: python tools/cfg4_f64_torch.py OUT.npy [n=4096] [S=1024] [tile:halo:taper:seg ...]    (env CFG4_CHECK=1: host check)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bdof_oracle as orc  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cfgs = [tuple(int(v) for v in a.split(':')) for a in sys.argv[4:]]          # tile:halo:taper:seg[:1 = with the long-range correction]
dev = torch.device('cuda:0')


def cfg4_inputs(n, r_zp=256.0, half=256.0):
    t = np.arange(n, dtype=np.float64)
    r2 = (t[:, None] - n / 2) ** 2 + (t[None, :] - n / 2) ** 2
    zone = (np.floor(r2 / (2 * r_zp * 4.0)).astype(np.int64) % 2 == 0) & (r2 < r_zp ** 2)
    slab = np.where(zone, 5e-5, 0.0)
    edge = np.clip((half + 16. - np.abs(t - n / 2)) / 32., 0., 1.)
    soft = 0.5 - 0.5 * np.cos(np.pi * edge)
    return slab, soft[:, None] * soft[None, :]


slab, probe = cfg4_inputs(n)
slab = slab.astype(np.float32).astype(np.float64)                      # what the device is handed
k = 2. * orc.PI * 1.0 / (1240. / 5000.)
cmod = torch.from_numpy(np.exp(1j * k * slab) * np.exp(-k * 0.1 * slab)).to(dev)
p0 = torch.from_numpy(probe.astype(np.complex64).astype(np.complex128)).to(dev)
rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def whole_field(S):
    h = torch.from_numpy(np.fft.ifftshift(orc.get_kernel(1.0, 1240. / 5000., np.array([1., 1., 1.]), (n, n)))).to(dev)
    w = p0.clone()
    for z in range(S):
        w = w * cmod
        if z < S - 1:
            w = torch.fft.ifft2(torch.fft.fft2(w) * h)
    return w


def tiled(S, tile, halo, taper, seg, correct=False):
    """correct: after every range add D psi_in, D = (whole-field free-space step over the range) - (the tiles' free-space step
    over the range, stitched): the long-range part of the band-limited propagator that no tile sees, to first order in the
    object's phase over one range.  One whole-field transform pair + one transform pair per tile per RANGE."""
    hN = torch.from_numpy(np.fft.ifftshift(orc.get_kernel(1.0, 1240. / 5000., np.array([1., 1., 1.]), (n, n)))).to(dev)
    h = torch.from_numpy(np.fft.ifftshift(orc.get_kernel_tile(1.0, 1240. / 5000., np.array([1., 1., 1.]), (tile, tile), (n, n)))).to(dev)
    hexact = h
    hmode = os.environ.get('CFG4_H32', '')
    if hmode == 'round':                       # what a fixed float32 table of the transfer function costs: everything else float64
        h = h.to(torch.complex64).to(torch.complex128)
    elif hmode == 'dither':                    # D copies, each part rounded down or up so that the mean over D slices is exact
        D = 64
        def dith(x):
            lo = x.to(torch.float32)
            lo = torch.where(lo.to(torch.float64) > x, torch.nextafter(lo, torch.full_like(lo, -float('inf'))), lo)
            hi = torch.nextafter(lo, torch.full_like(lo, float('inf')))
            p = (x - lo.to(torch.float64)) / (hi.to(torch.float64) - lo.to(torch.float64))
            ph = torch.rand_like(p)
            d = torch.arange(D, device=dev, dtype=torch.float64)[:, None, None]
            up = torch.floor((d + 1) * p + ph) > torch.floor(d * p + ph)
            return torch.where(up, hi.to(torch.float64), lo.to(torch.float64))
        hd = torch.complex(dith(h.real), dith(h.imag))      # [D][T][T]
    w1 = np.ones(tile)
    if taper > 0:
        ramp = 0.5 - 0.5 * np.cos(np.pi * (np.arange(taper) + 0.5) / taper)
        w1[:taper], w1[tile - taper:] = ramp, ramp[::-1]
    win = torch.from_numpy(w1[:, None] * w1[None, :]).to(dev)
    o = np.array(orc.tile_origins(n, tile, halo))
    core = tile - 2 * halo
    y0 = np.repeat(o, len(o))
    x0 = np.tile(o, len(o))
    ry = y0[:, None] + np.arange(tile)[None, :]
    rx = x0[:, None] + np.arange(tile)[None, :]
    iy = torch.from_numpy(ry % n).to(dev)
    ix = torch.from_numpy(rx % n).to(dev)
    inside = torch.from_numpy(((ry >= 0) & (ry < n))[:, :, None] & ((rx >= 0) & (rx < n))[:, None, :]).to(dev)
    cm = torch.where(inside, cmod[iy[:, :, None], ix[:, None, :]], torch.ones((), dtype=cmod.dtype, device=dev))
    field = p0.clone()
    for z0 in range(0, S, seg):
        nz = min(seg, S - z0)
        w = field[iy[:, :, None], ix[:, None, :]] * win
        nprop = nz if z0 + nz < S else nz - 1                    # transfer-function steps in this range (none after the last slice)
        if correct and nprop > 0:
            wf = torch.fft.ifft2(torch.fft.fft2(w) * hexact ** nprop)        # the tiles' free-space step over the range
            whole = torch.fft.ifft2(torch.fft.fft2(field) * hN ** nprop)      # the field's
        for z in range(z0, z0 + nz):
            w = w * cm
            if z < S - 1:
                w = torch.fft.ifft2(torch.fft.fft2(w) * (hd[z % 64] if hmode == 'dither' else h))
        new = torch.empty_like(field)
        for t in range(len(y0)):
            ya, xa = int(y0[t]) + halo, int(x0[t]) + halo
            nyc, nxc = min(core, n - ya), min(core, n - xa)
            new[ya:ya + nyc, xa:xa + nxc] = w[t, halo:halo + nyc, halo:halo + nxc]
            if correct and nprop > 0:
                new[ya:ya + nyc, xa:xa + nxc] -= wf[t, halo:halo + nyc, halo:halo + nxc]
        if correct and nprop > 0:
            new = new + whole
        field = new
    return field


if os.environ.get('CFG4_CHECK'):
    import scipy.fft as sfft
    S2 = 96
    hh = np.fft.ifftshift(orc.get_kernel(1.0, 1240. / 5000., np.array([1., 1., 1.]), (n, n)))
    cm = cmod.cpu().numpy()
    w = probe.astype(np.complex64).astype(np.complex128)
    for z in range(S2):
        w = w * cm
        if z < S2 - 1:
            w = sfft.ifft2(sfft.fft2(w, workers=-1) * hh, workers=-1)
    g = whole_field(S2)
    print('torch complex128 on the GPU vs scipy float64 on the host, %d slices of %d^2: %.2e' % (S2, n, rel(g, torch.from_numpy(w).to(dev))), flush=True)

t0 = time.perf_counter()
ref = whole_field(S)
torch.cuda.synchronize()
print('whole field float64, %d slices of %d^2: %.1f s' % (S, n, time.perf_counter() - t0), flush=True)
np.save(out, ref.cpu().numpy())
for c in cfgs:
    tile, halo, taper, seg = c[:4]
    corr = len(c) > 4 and c[4] == 1
    t0 = time.perf_counter()
    f = tiled(S, tile, halo, taper, seg, corr)
    torch.cuda.synchronize()
    print('float64 TILED %d/%d taper %d stitch every %d%s: tiling error wave %.3e intensity %.3e   (%.1f s)' % (
        tile, halo, taper, seg, ' + long-range correction' if corr else '', rel(f, ref), rel(f.abs() ** 2, ref.abs() ** 2), time.perf_counter() - t0), flush=True)
    if os.environ.get('CFG4_ERRMAP'):
        e = (f - ref).abs() ** 2
        tot = float(e.sum())
        c0 = n // 2
        for r in (128, 256, 272, 320, 512, 1024):
            print('    error energy within +-%d of the centre: %.3f' % (r, float(e[c0 - r:c0 + r, c0 - r:c0 + r].sum()) / tot))
        core = tile - 2 * halo
        d = torch.arange(n, device=dev) % core
        d = torch.minimum(d, core - 1 - d)                       # distance to the nearest core boundary along one axis
        dd = torch.minimum(d[:, None], d[None, :])
        for lo, hi in ((0, 4), (4, 16), (16, 64), (64, 1 << 20)):
            m = (dd >= lo) & (dd < hi)
            print('    distance %d..%d to a core boundary: %.3f of the error energy on %.3f of the pixels' % (lo, hi, float(e[m].sum()) / tot, float(m.float().mean())))
    del f
