"""numpy model of the device FFT plan used by beyond_dof_amd/csrc (Stockham autosort,
8 elements per thread, radix plan [8, mid..., 8]).  Validates the index math against
numpy.fft and counts LDS bank conflicts of candidate layouts.  Development aid only."""
import numpy as np

PLANS = {64: [8, 8], 128: [8, 2, 8], 256: [8, 4, 8], 512: [8, 8, 8], 1024: [8, 2, 8, 8], 2048: [8, 4, 8, 8],
         4096: [8, 8, 8, 8]}


def pad_row(idx):
    return idx + (idx >> 4)


def model_fft(x, sign, conflicts=None, lds_index=pad_row, lanes_of=None):
    """x: (N,) complex. Emulates T = N/8 threads; returns natural-order DFT (unnormalised)."""
    N = len(x)
    radices = PLANS[N]
    T = N // 8
    regs = np.zeros((T, 8), complex)
    for tid in range(T):
        for m in range(8):
            regs[tid, m] = x[tid + m * T]
    lds = np.zeros(2 * N, complex)
    p = 1
    for s, r in enumerate(radices):
        nb = 8 // r
        t = N // r
        first = s == 0
        last = s == len(radices) - 1
        if not first:
            # read phase
            for tid in range(T):
                for j in range(nb):
                    i = tid + j * T
                    for m in range(r):
                        regs[tid, j * r + m] = lds[lds_index(i + m * t)]
            if conflicts is not None:
                for j in range(nb):
                    for m in range(r):
                        conflicts('read', s, np.array([lds_index(tid + j * T + m * t) for tid in range(T)]))
        out = np.zeros((T, 8), complex)
        waddr = np.zeros((T, 8), int)
        for tid in range(T):
            for j in range(nb):
                i = tid + j * T
                k = i % p
                u = regs[tid, j * r:(j + 1) * r].copy()
                u = u * np.exp(sign * 2j * np.pi * np.arange(r) * k / (p * r))
                v = np.array([np.sum(u * np.exp(sign * 2j * np.pi * np.arange(r) * q / r)) for q in range(r)])
                for q in range(r):
                    out[tid, j * r + q] = v[q]
                    waddr[tid, j * r + q] = (i - k) * r + k + q * p
        if last:
            # positions must be tid + q*T
            for tid in range(T):
                for q in range(8):
                    assert waddr[tid, q] == tid + q * T, (N, tid, q, waddr[tid, q])
            regs = out
        else:
            for tid in range(T):
                for e in range(8):
                    lds[lds_index(waddr[tid, e])] = out[tid, e]
            if conflicts is not None:
                for e in range(8):
                    conflicts('write', s, np.array([lds_index(waddr[tid, e]) for tid in range(T)]))
        p *= r
    y = np.zeros(N, complex)
    for tid in range(T):
        for q in range(8):
            y[tid + q * T] = regs[tid, q]
    return y


class ConflictCounter:
    """ds_write_b64: 16-lane groups, 32 banks x 4 B; ds_read_b64: 32-lane groups, 64 banks x 4 B."""
    def __init__(self):
        self.stats = {}

    def __call__(self, kind, stage, slots8):          # slots8: per-thread 8-byte slot indices (row-local)
        grp = 16 if kind == 'write' else 32
        nbank_slots = 16 if kind == 'write' else 32    # 8-byte slots per bank row
        worst = 1
        for g0 in range(0, len(slots8), grp):
            sl = slots8[g0:g0 + grp]
            banks = {}
            for a in set(sl.tolist()):
                banks.setdefault(a % nbank_slots, set()).add(a)
            worst = max(worst, max(len(v) for v in banks.values()))
        key = (kind, stage)
        self.stats[key] = max(self.stats.get(key, 1), worst)


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for N in PLANS:
        x = rng.normal(size=N) + 1j * rng.normal(size=N)
        cc = ConflictCounter()
        for sign in (-1, +1):
            y = model_fft(x, sign, conflicts=cc)
            ref = np.fft.fft(x) if sign < 0 else np.fft.ifft(x) * N
            err = np.abs(y - ref).max() / np.abs(ref).max()
            assert err < 1e-12, (N, sign, err)
        print(N, PLANS[N], 'ok; worst-way LDS conflicts (row layout, pad idx>>4):', cc.stats)


def model_tile_transposed(x_rows, sign):
    """Emulates the tile kernels: TILE rows, all stages but the last wave-local per row, the last stage done
    by threads mapped (r = q % TILE, j = q // TILE) and stored transposed.  Returns out[pos][r]."""
    TILE, N = x_rows.shape
    radices = PLANS[N]
    T = N // 8
    RS = ((N + N // 16 + 29) // 32) * 32 + 2
    lds = np.zeros(TILE * RS, complex)
    for r in range(TILE):
        regs = np.array([[x_rows[r, tid + m * T] for m in range(8)] for tid in range(T)])
        p = 1
        for s, rad in enumerate(radices[:-1]):
            nb, t = 8 // rad, N // rad
            if s > 0:
                for tid in range(T):
                    for j in range(nb):
                        for m in range(rad):
                            regs[tid, j * rad + m] = lds[r * RS + pad_row(tid + j * T + m * t)]
            out = np.zeros_like(regs)
            for tid in range(T):
                for j in range(nb):
                    i = tid + j * T
                    k = i % p
                    u = regs[tid, j * rad:(j + 1) * rad] * np.exp(sign * 2j * np.pi * np.arange(rad) * k / (p * rad))
                    v = np.array([np.sum(u * np.exp(sign * 2j * np.pi * np.arange(rad) * q / rad)) for q in range(rad)])
                    for q in range(rad):
                        lds[r * RS + pad_row((i - k) * rad + k + q * p)] = v[q]
            p *= rad
    out = np.zeros((N, TILE), complex)
    worst = 1
    for q0 in range(0, TILE * T, 32):            # 32-lane read groups of the transposed stage
        slots = {}
        for q in range(q0, q0 + 32):
            r, j = q % TILE, q // TILE
            for m in range(1):
                a = r * RS + pad_row(j + m * T)
                slots.setdefault(a % 32, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    for q in range(TILE * T):
        r, j = q % TILE, q // TILE
        u = np.array([lds[r * RS + pad_row(j + m * T)] for m in range(8)])
        u = u * np.exp(sign * 2j * np.pi * np.arange(8) * j / N)
        v = np.array([np.sum(u * np.exp(sign * 2j * np.pi * np.arange(8) * qq / 8)) for qq in range(8)])
        for qq in range(8):
            out[j + qq * T, r] = v[qq]
    return out, worst


def check_affine_slots():
    """slot(base + c) == slot(base) + c + (c >> 4) for every access the plans generate."""
    for N, radices in PLANS.items():
        T = N // 8
        p = 1
        for s_, r in enumerate(radices):
            nb, t = 8 // r, N // r
            for tid in range(T):
                for j in range(nb):
                    i = tid + j * T
                    if s_ > 0:
                        for m in range(r):
                            assert pad_row(i + m * t) == pad_row(i) + m * t + ((m * t) >> 4), (N, 'read', s_, i, m)
                    if s_ < len(radices) - 1:
                        k = i % p
                        b = (i - k) * r + k
                        for q in range(r):
                            assert pad_row(b + q * p) == pad_row(b) + q * p + ((q * p) >> 4), (N, 'write', s_, i, q)
            p *= r
    print('affine slot property holds for all plans')


if __name__ == '__main__':
    check_affine_slots()
    rng = np.random.default_rng(1)
    for N, TILE in [(64, 64), (128, 32), (256, 16), (512, 16), (1024, 16)]:
        x = rng.normal(size=(TILE, N)) + 1j * rng.normal(size=(TILE, N))
        for sign in (-1, 1):
            out, worst = model_tile_transposed(x, sign)
            ref = (np.fft.fft(x, axis=1) if sign < 0 else np.fft.ifft(x, axis=1) * N).T
            assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-12
        print('transposed tail ok N={} TILE={} worst read conflict {}-way'.format(N, TILE, worst))
