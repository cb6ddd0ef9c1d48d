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
