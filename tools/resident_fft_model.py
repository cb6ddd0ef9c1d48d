"""Index/butterfly model of the LDS-resident 2-D FFT (csrc/bdof_resident.h): in-place Stockham passes over a square field
with pitch P, butterflies of radix 2/3/4/5/8/9 written exactly as in the kernel.  Checks against numpy.fft."""
import numpy as np

PLANS = {32: (8, 4), 36: (4, 9), 48: (8, 2, 3), 64: (8, 8), 72: (9, 8), 80: (8, 2, 5), 96: (8, 4, 3), 128: (8, 4, 4)}


def dft3(a, s):
    t1 = a[1] + a[2]
    t2 = a[0] - 0.5 * t1
    t3 = s * 1j * 0.86602540378443865 * (a[1] - a[2])
    return [a[0] + t1, t2 + t3, t2 - t3]


def dft5(a, s):
    c1, c2, s1, s2 = 0.30901699437494742, -0.80901699437494742, 0.95105651629515357, 0.58778525229247313
    t1, t2, t3, t4 = a[1] + a[4], a[2] + a[3], a[1] - a[4], a[2] - a[3]
    m1 = a[0] + c1 * t1 + c2 * t2
    m2 = a[0] + c2 * t1 + c1 * t2
    n1 = s * 1j * (s1 * t3 + s2 * t4)
    n2 = s * 1j * (s2 * t3 - s1 * t4)
    return [a[0] + t1 + t2, m1 + n1, m2 + n2, m2 - n2, m1 - n1]


def dft9(a, s):
    u = list(a)
    for n2 in range(3):
        u[n2], u[n2 + 3], u[n2 + 6] = dft3([u[n2], u[n2 + 3], u[n2 + 6]], s)
    w = lambda p: np.exp(s * 2j * np.pi * p / 9)
    u[4] *= w(1); u[7] *= w(2); u[5] *= w(2); u[8] *= w(4)
    out = [0] * 9
    for k1 in range(3):
        x = dft3([u[3 * k1], u[3 * k1 + 1], u[3 * k1 + 2]], s)
        for k2 in range(3):
            out[k1 + 3 * k2] = x[k2]
    return out


def dftn(a, s):
    n = len(a)
    if n == 3: return dft3(a, s)
    if n == 5: return dft5(a, s)
    if n == 9: return dft9(a, s)
    return list(np.array([sum(a[m] * np.exp(s * 2j * np.pi * m * k / n) for m in range(n)) for k in range(n)]))


def fft_pass(f, N, P, R, NS, s, along_y, T):
    NB = N * (N // R)
    cnt = (NB + T - 1) // T
    tw = np.exp(s * 2j * np.pi * np.arange(N) / N)
    regs = {}
    for tid in range(T):
        for c in range(cnt):
            q = tid + c * T
            if q >= NB: continue
            line, j = q % N, q // N
            base, es = (line * P, 1) if along_y else (line, P)
            k = j % NS
            u = [f[base + (j + m * (N // R)) * es] for m in range(R)]
            for m in range(1, R):
                u[m] = u[m] * tw[k * m * (N // (NS * R))]
            regs[q] = dftn(u, s)
    for q, u in regs.items():
        line, j = q % N, q // N
        base, es = (line * P, 1) if along_y else (line, P)
        k = j % NS
        j0 = (j // NS) * NS * R + k
        for m in range(R):
            f[base + (j0 + m * NS) * es] = u[m]


def fft2(f, N, P, s, T):
    for along_y in (True, False):
        ns = 1
        for R in PLANS[N]:
            fft_pass(f, N, P, R, ns, s, along_y, T)
            ns *= R


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for r in (3, 5, 9):
        a = rng.normal(size=r) + 1j * rng.normal(size=r)
        for s in (-1, 1):
            ref = np.fft.fft(a) if s < 0 else np.fft.ifft(a) * r
            assert np.allclose(dftn(list(a), s), ref), (r, s)
    for N in PLANS:
        P = N | 1
        x = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))      # [x][y]
        f = np.zeros(N * P, dtype=complex)
        for i in range(N):
            f[i * P:i * P + N] = x[i]
        fft2(f, N, P, -1, 256)
        got = np.array([f[i * P:i * P + N] for i in range(N)])
        assert np.allclose(got, np.fft.fft2(x)), N
        fft2(f, N, P, +1, 256)
        got = np.array([f[i * P:i * P + N] for i in range(N)])
        assert np.allclose(got, x * N * N), N
        print('N = %d plan %s ok' % (N, PLANS[N]))
