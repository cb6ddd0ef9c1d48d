// LDS-resident engine for small square wavefields (N <= 128: the 72 x 72 probes of cnn_propagator/reconstruct_ptycho.py:106).
//
// A whole N x N complex field fits in one CU's LDS (72 x 73 x 8 B = 42 KB), so ONE workgroup carries one wavefield through
// ALL slices without the field ever leaving the CU: forward sweep (np_funcs.py:37-43), detector step (:45-61), loss and
// seed (ptychography.py:79 / fullfield.py:106), adjoint sweep — one launch per minibatch instead of ~8 per
// slice.  HBM traffic per pixel per slice-step drops from the 104 B of the streaming engines to 40 B:
//   forward  : modulation factor 8 + phi tape write 8                          (the transfer function is L2-resident)
//   backward : tape read 8 + modulation factor 8 + gradient write 8
// The 2-D transforms are in-place Stockham passes over the LDS image (radices 2/3/4/5/8/9; every thread reads all its
// butterflies, barrier, writes them to the autosort positions, barrier).  Lines-fastest thread mapping + odd row pitch
// make every LDS access of a pass conflict-free in both directions.  Index math: tools/resident_fft_model.py.
#pragma once
#include "bdof_generic.h"

// Exact twiddles (hi + lo pairs) in the ADJOINT passes of this kernel (what -DBDOF_EXACT_TRANSFORMS gives the streaming kernels): measured in
// round 3 on far-field ptychography (golden vector G17's configuration, 64^2 x 64 slices): gradient error 1.86e-6 -> 1.70e-6 for
// +2.3 % (72^2) / +3.3 % (64^2) kernel time — the float32 rounding of the transforms themselves is the floor here, not the
// tables (tools/precision_model.py).  Off by default; -DBDOF_RES_EXACT_ADJOINT switches it on.
#ifdef BDOF_RES_EXACT_ADJOINT
constexpr bool BDOF_EX_RES = true;
#else
constexpr bool BDOF_EX_RES = false;
#endif

template <int N> struct ResPlan;      // radices of the Stockham passes of one line, and the workgroup size
template <> struct ResPlan<32> { static constexpr int n = 2, R0 = 8, R1 = 4, R2 = 1, T = 64, WPE = 2; static constexpr bool FUSE = true; };
template <> struct ResPlan<36> { static constexpr int n = 2, R0 = 4, R1 = 9, R2 = 1, T = 128, WPE = 2; static constexpr bool FUSE = true; };
template <> struct ResPlan<48> { static constexpr int n = 3, R0 = 8, R1 = 2, R2 = 3, T = 192, WPE = 3; static constexpr bool FUSE = true; };
template <> struct ResPlan<64> { static constexpr int n = 2, R0 = 8, R1 = 8, R2 = 1, T = 512, WPE = 4; static constexpr bool FUSE = true; };
// 72 (the reference drivers' probe): wave-local lines (ResWave<72>), radix 9 first — the stride-8 writes of an (8, 9) plan are
// 4-way bank conflicts in that lane mapping; the point-wise epilogues of the fused form do not fit the 80 registers that
// two 11-wave workgroups per CU leave
template <> struct ResPlan<72> { static constexpr int n = 2, R0 = 9, R1 = 8, R2 = 1, T = 704, WPE = 6; static constexpr bool FUSE = false; };
template <> struct ResPlan<80> { static constexpr int n = 3, R0 = 8, R1 = 2, R2 = 5, T = 512, WPE = 4; static constexpr bool FUSE = false; };
template <> struct ResPlan<96> { static constexpr int n = 3, R0 = 8, R1 = 4, R2 = 3, T = 768, WPE = 3; static constexpr bool FUSE = true; };
template <> struct ResPlan<128> { static constexpr int n = 3, R0 = 8, R1 = 4, R2 = 4, T = 1024, WPE = 4; static constexpr bool FUSE = true; };

// Wave-local lines: L lanes of one wave own one line (L >= butterflies per line in every pass), LPW lines per wave.  Both
// passes of a line then run in the same wave with no workgroup barrier in between (a wave's LDS accesses execute in
// program order), and the x-lines of a propagation step go through forward pass 0, 1 (x h), inverse pass 0, 1 without one:
// 4 barriers per step instead of 17.  L = 0: workgroup-wide passes (res_pass).
template <int N> struct ResWave { static constexpr int L = 0, LPW = 0; };
template <> struct ResWave<72> { static constexpr int L = 9, LPW = 7; };       // 7 x 9 = 63 lanes; 11 waves cover 77 >= 72 lines

static inline bool resident_supported(int n) {
    return n == 32 || n == 36 || n == 48 || n == 64 || n == 72 || n == 80 || n == 96 || n == 128;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() on gfx9 also drains vmcnt, i.e. every barrier would wait
// for the global loads prefetched for the next slice and for the tape / gradient stores of the previous one — one HBM round
// trip per barrier.  Inside a sweep nothing communicates between threads through global memory, so waiting for the LDS
// counter is sufficient; the one hand-over through global memory (forward tape -> adjoint sweep) has a full barrier.
__device__ __forceinline__ void res_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// e / N for the element indices of the point-wise sweeps (0 <= e < N * N + 1024), as a 24-bit multiply and a shift (both
// full rate; the compiler's v_mul_hi_i32 is quarter rate).  The magic constant is checked exhaustively at compile time.
template <int N> struct ResDiv {
    static constexpr unsigned SH = 20, M = ((1u << SH) + N - 1) / N;
    static constexpr bool ok() {
        for (unsigned e = 0; e < (unsigned)(N * N + 1024); ++e)
            if (((e * M) >> SH) != e / (unsigned)N) return false;
        return true;
    }
    static_assert(M < (1u << 24) && (unsigned long long)(N * N + 1024) * M < (1ull << 32), "24 x 24 -> 32 bit multiply");
};
template <int N> __device__ __forceinline__ int res_div(int e) {
    static_assert(ResDiv<N>::ok(), "magic division constant is not exact over the element range");
    return (int)(__umul24((unsigned)e, ResDiv<N>::M) >> ResDiv<N>::SH);
}

// ROUND: how sqrt(3)/2 is represented (see mul_sqrt_half in bdof_fft.h): 0 hi + lo pair, 1 nearest float32 (1.8e-8 short),
// 2 its upper neighbour (5.1e-8 long)
template <int ROUND> __device__ __forceinline__ float mul_sqrt3_half(float t) {
    if constexpr (ROUND == 0) return fmaf(t, 0.86602540378443865f, t * 1.5543624e-8f);
    else if constexpr (ROUND == 1) return t * 0.86602540378443865f;
    else return t * 0.86602544784545898f;
}

template <int SIGN, int ROUND = 0> __device__ __forceinline__ void dft3(cf& a0, cf& a1, cf& a2) {
    const cf t1 = cadd(a1, a2);
    const cf t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
    const cf d = csub(a1, a2);
    const float s = (float)SIGN;
    const cf t3 = make_float2(-s * mul_sqrt3_half<ROUND>(d.y), s * mul_sqrt3_half<ROUND>(d.x));      // SIGN * i * sqrt(3)/2 * (a1 - a2)
    a0 = cadd(a0, t1);
    a1 = cadd(t2, t3);
    a2 = csub(t2, t3);
}

template <int SIGN> __device__ __forceinline__ void dft5(cf& a0, cf& a1, cf& a2, cf& a3, cf& a4) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f * (float)SIGN, s2 = 0.58778525229247313f * (float)SIGN;
    const cf t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
    const cf m1 = make_float2(a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y);
    const cf m2 = make_float2(a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y);
    const cf p1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    const cf p2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    const cf n1 = make_float2(-p1.y, p1.x), n2 = make_float2(-p2.y, p2.x);      // i * p (SIGN already in s1, s2)
    a0 = cadd(a0, cadd(t1, t2));
    a1 = cadd(m1, n1);
    a4 = csub(m1, n1);
    a2 = cadd(m2, n2);
    a3 = csub(m2, n2);
}

template <int SIGN, int ROUND = 0> __device__ __forceinline__ void dft9(cf (&u)[9]) {
    dft3<SIGN, ROUND>(u[0], u[3], u[6]);
    dft3<SIGN, ROUND>(u[1], u[4], u[7]);
    dft3<SIGN, ROUND>(u[2], u[5], u[8]);
    const float s = (float)SIGN;
    // (hi + lo pairs for these three constants were measured too: no further gain in accuracy)
    const cf w1 = make_float2(0.76604444311897804f, s * 0.64278760968653933f);
    const cf w2 = make_float2(0.17364817766693035f, s * 0.98480775301220806f);
    const cf w4 = make_float2(-0.93969262078590838f, s * 0.34202014332566873f);
    u[4] = cmul(u[4], w1);
    u[7] = cmul(u[7], w2);
    u[5] = cmul(u[5], w2);
    u[8] = cmul(u[8], w4);
    dft3<SIGN, ROUND>(u[0], u[1], u[2]);
    dft3<SIGN, ROUND>(u[3], u[4], u[5]);
    dft3<SIGN, ROUND>(u[6], u[7], u[8]);
    // u[3 k1 + k2] holds X[k1 + 3 k2]: transpose
    cf t;
    t = u[1]; u[1] = u[3]; u[3] = t;
    t = u[2]; u[2] = u[6]; u[6] = t;
    t = u[5]; u[5] = u[7]; u[7] = t;
}

// ROUND: representation of the irrational butterfly constants (bdof_fft.h).  This engine serves localised probes (no carrier
// to split off), where the accuracy of the transform chain IS the accuracy of the result: it uses 0, the hi + lo pairs
// (+10 % kernel time; intensity error at 72^2 x 256 slices 1.5e-5 instead of 2.4e-5 with the alternating 1 / 2 scheme).
template <int R, int SIGN, int ROUND> __device__ __forceinline__ void res_dft(cf (&u)[R]) {
    if constexpr (R == 2) dft2<SIGN>(u[0], u[1]);
    else if constexpr (R == 3) dft3<SIGN, ROUND>(u[0], u[1], u[2]);
    else if constexpr (R == 4) dft4<SIGN>(u[0], u[1], u[2], u[3]);
    else if constexpr (R == 5) dft5<SIGN>(u[0], u[1], u[2], u[3], u[4]);
    else if constexpr (R == 8) dft8<SIGN, ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
    else if constexpr (R == 9) dft9<SIGN, ROUND>(u);
}

// One Stockham pass (radix R, NS = product of the earlier radices) over the N lines of the field, in place.
// ALONG_Y: lines are the rows x (elements contiguous); else lines are the columns y (element stride P).
// Epilogue hooks of a pass: `pre(c, m, line, pos)` is called in the read phase for the element the thread will write at
// position `pos` of line `line` (global loads are issued here), `post(c, m, line, pos, v)` returns the value to store.
// Only used on passes along x (lines = columns: consecutive lanes are consecutive y, so the global accesses coalesce).
struct EpiNone {
    static constexpr bool active = false, pre_in_pass = false;
    __device__ __forceinline__ void pre(int, int, int, int) {}
    __device__ __forceinline__ cf post(int, int, int, int, cf v) { return v; }
};

// EX: exact transform constants — the twiddle table's lo parts (tw[N + j], bdof_fft.h) are multiplied in as well; the passes of
// the ADJOINT sweep run with it (the gradient's error is made there, the forward sweep rides on its carrier).
template <int N, int R, int NS, int SIGN, bool ALONG_Y, int T, bool EX = false, class Epi>
__device__ __forceinline__ void res_pass(cf* f, const cf* tw, int tid, Epi& epi) {
    // every index below is invariant across slices: without this the compiler hoists the address math of all 16 passes
    // out of the slice loop and keeps it in registers (250+ VGPRs, spills).  Recomputing it per pass is a few VALU ops.
    asm volatile("" : "+v"(tid));
    constexpr int P = N | 1;
    constexpr int NB = N * (N / R);
    constexpr int CNT = (NB + T - 1) / T;
    constexpr int ES = ALONG_Y ? 1 : P;
    cf u[CNT][R];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int q = tid + c * T;
        if (CNT * T == NB || q < NB) {
            const int line = q % N, j = q / N;
            const cf* src = f + (ALONG_Y ? line * P : line) + j * ES;
#pragma unroll
            for (int m = 0; m < R; ++m) u[c][m] = src[m * (N / R) * ES];
            if constexpr (Epi::active && Epi::pre_in_pass) {
                const int j0 = (j / NS) * NS * R + j % NS;
#pragma unroll
                for (int m = 0; m < R; ++m) epi.pre(c, m, line, j0 + m * NS);
            }
            if constexpr (NS > 1) {
                const int k = j % NS;
#pragma unroll
                for (int m = 1; m < R; ++m) {
                    const int ti = k * m * (N / (NS * R));
                    u[c][m] = tw_mul<SIGN, EX>(u[c][m], tw[ti], tw[(EX ? N : 0) + ti]);
                }
            }
            res_dft<R, SIGN, 0>(u[c]);
        }
    }
    res_sync();
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int q = tid + c * T;
        if (CNT * T == NB || q < NB) {
            const int line = q % N, j = q / N;
            const int j0 = (j / NS) * NS * R + j % NS;
            cf* dst = f + (ALONG_Y ? line * P : line) + j0 * ES;
#pragma unroll
            for (int m = 0; m < R; ++m) dst[m * NS * ES] = epi.post(c, m, line, j0 + m * NS, u[c][m]);
        }
    }
    res_sync();
}

// Wave-local Stockham pass: lane (li, j) of wave w holds butterfly j of line w * LPW + li.  In place without a barrier: every
// lane of the wave has issued its reads before any lane's write (one instruction stream), and only this wave touches
// these lines while the line direction does not change.
template <int N, int R, int NS, int SIGN, bool ALONG_Y, int L, int LPW, bool EX = false, class Epi>
__device__ __forceinline__ void res_wpass(cf* f, const cf* tw, int tid, Epi& epi) {
    asm volatile("" : "+v"(tid));
    constexpr int P = N | 1, ES = ALONG_Y ? 1 : P, NBL = N / R;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane / L, j = lane - li * L;
    const int line = wave * LPW + li;
    const bool act = li < LPW && line < N && j < NBL;
    cf u[R];
    cf* base = f + (ALONG_Y ? line * P : line);
    const int k = j % NS;
    const int j0 = (j / NS) * NS * R + k;
    if (act) {
#pragma unroll
        for (int m = 0; m < R; ++m) u[m] = base[(j + m * NBL) * ES];
        if constexpr (Epi::active) {
#pragma unroll
            for (int m = 0; m < R; ++m) epi.pre(0, m, line, j0 + m * NS);
        }
        if constexpr (NS > 1) {
#pragma unroll
            for (int m = 1; m < R; ++m) {
                const int ti = k * m * (N / (NS * R));
                u[m] = tw_mul<SIGN, EX>(u[m], tw[ti], tw[(EX ? N : 0) + ti]);
            }
        }
        res_dft<R, SIGN, 0>(u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (act) {
#pragma unroll
        for (int m = 0; m < R; ++m) base[(j0 + m * NS) * ES] = epi.post(0, m, line, j0 + m * NS, u[m]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// both passes of every line of one direction (two-pass plans only); no workgroup barrier inside
template <int N, int SIGN, bool ALONG_Y, bool EX = false, class Epi> __device__ __forceinline__ void res_wlines(cf* f, const cf* tw, int tid, Epi& epi) {
    typedef ResPlan<N> Pl;
    static_assert(Pl::n == 2, "wave-local lines are written for two-pass plans");
    constexpr int L = ResWave<N>::L, LPW = ResWave<N>::LPW;
    static_assert(L >= N / Pl::R0 && L >= N / Pl::R1 && L * LPW <= 64, "lanes per line");
    EpiNone none;
    res_wpass<N, Pl::R0, 1, SIGN, ALONG_Y, L, LPW, EX>(f, tw, tid, none);
    res_wpass<N, Pl::R1, Pl::R0, SIGN, ALONG_Y, L, LPW, EX>(f, tw, tid, epi);
}

// radix and butterflies per thread of the LAST pass of a line (the one that carries an epilogue)
template <int N, int T> struct ResLast {
    typedef ResPlan<N> Pl;
    static constexpr int R = Pl::n > 2 ? Pl::R2 : Pl::R1;
    static constexpr int CNT = (N * (N / R) + T - 1) / T;
};

template <int N, int T, int SIGN, bool ALONG_Y, bool EX = false, class Epi>
__device__ __forceinline__ void res_lines(cf* f, const cf* tw, int tid, Epi& epi) {
    typedef ResPlan<N> Pl;
    EpiNone none;
    res_pass<N, Pl::R0, 1, SIGN, ALONG_Y, T, EX>(f, tw, tid, none);
    if constexpr (Pl::n > 2) {
        res_pass<N, Pl::R1, Pl::R0, SIGN, ALONG_Y, T, EX>(f, tw, tid, none);
        res_pass<N, Pl::R2, Pl::R0 * Pl::R1, SIGN, ALONG_Y, T, EX>(f, tw, tid, epi);
    } else {
        res_pass<N, Pl::R1, Pl::R0, SIGN, ALONG_Y, T, EX>(f, tw, tid, epi);
    }
}

// un-normalised 2-D DFT of the field image f[x * P + y], SIGN = -1 forward, +1 inverse; `epi` rides on the last pass
template <int N, int T, int SIGN, bool EX = false, class Epi> __device__ __forceinline__ void res_fft2(cf* f, const cf* tw, int tid, Epi& epi) {
    EpiNone none;
    if constexpr (ResWave<N>::L > 0) {
        static_assert(T >= 64 * ((N + ResWave<N>::LPW - 1) / ResWave<N>::LPW), "not enough waves for the lines");
        res_wlines<N, SIGN, true, EX>(f, tw, tid, none);
        res_sync();
        res_wlines<N, SIGN, false, EX>(f, tw, tid, epi);
        res_sync();
    } else {
        res_lines<N, T, SIGN, true, EX>(f, tw, tid, none);
        res_lines<N, T, SIGN, false, EX>(f, tw, tid, epi);
    }
}
template <int N, int T, int SIGN, bool EX = false> __device__ __forceinline__ void res_fft2(cf* f, const cf* tw, int tid) {
    EpiNone none;
    res_fft2<N, T, SIGN, EX>(f, tw, tid, none);
}

struct ResArgs {
    const cf* probe;       // [N][N] eps part of the probe, [x][y]
    const cf* hsT;         // [kx][ky] transfer function / (N*N); hD > 0: hD dithered copies of it, one after the other — the
                           // step after slice z (and its adjoint) multiplies by copy z mod hD (bdof_set_transfer_f64)
    const cf* hdetT;       // [kx][ky] detector transfer function / (N*N)   (near field)
    cf* tape;              // nullable: phi_z tape, slice z of wavefield b at tape + z * tape_stride + b * N * N
    size_t tape_stride;
    float2* grot;          // [B][S][N][N]
    ObjView obj;
    const cf* carrier;     // device [2 S]: a_z, then a_z (cbar - 1) (mean-refraction carrier, modulate_eps_s)
    cf carrier_det;        // constant part of the detector wave (far field: DC bin value)
    // Carrier FIELD (bdof_set_probe_stack): the probe propagated through free space, p_z [S][N][N] at the entrance of every
    // slice and pdet [N][N] at the detector (far field: its un-normalised fft2, [kx][ky]), computed by the host in
    // float64.  The wave is then held as psi_z = p_z + eps_z and only the SCATTERED part eps goes through the float32
    // transforms — the generalisation of the scalar carrier to a localised probe.  nullptr: scalar carrier above.
    const cf* pstack;
    const cf* pdet;
    const float* meas;     // nullable; real detectors [b][x][y], far field [b][ky][kx]
    cf* out_wave;          // nullable; same order as meas
    double* partial;       // [2 * gridDim.x * T / 64]: per workgroup and wave, (sum r^2, sum r |d|)
    const cf* twiddle;     // [N] exp(-2 pi i k / N)
    int B, S, det_mode, tf_all, do_grad;
    float k, seed_scale;
    int meas_dev;          // `meas` holds m - |carrier_det| (loss_seed_dev, bdof_kernels.h)
    int hD;                // dithered copies of hsT (0: one table)
    float dref;
    cf* gpsi0;             // nullable [B][N][N]: G(psi_0), the probe gradient per wavefield
    const double2* pdet64; // nullable: `pdet` in float64 — the residual |d| - m is then formed in float64 (loss_seed_f64)
};

// transfer-function multiply folded into the last pass of the forward transform: the thread writing element (kx, ky)
// multiplies it by hT[kx][ky] on the way (no separate read-modify-write sweep of the LDS image)
template <int N, int T, bool CONJ> struct EpiH {
    static constexpr bool active = true, pre_in_pass = true;      // L2-resident table: the latency of one pass is enough
    const cf* hT;
    cf h[ResLast<N, T>::CNT][ResLast<N, T>::R];
    __device__ __forceinline__ void pre(int c, int m, int line, int pos) { h[c][m] = hT[pos * N + line]; }
    __device__ __forceinline__ cf post(int c, int m, int, int, cf v) {
        cf w = h[c][m];
        if constexpr (CONJ) w.y = -w.y;
        return cmul(v, w);
    }
};

// Issue the loads of an epilogue ahead of the pass that consumes them: same (butterfly, output) -> (line, position) map as
// the last pass along x of res_lines.
template <int N, int T, class Epi> __device__ __forceinline__ void res_epi_prefetch(int tid, Epi& epi) {
    if constexpr (Epi::active && !Epi::pre_in_pass) {
        asm volatile("" : "+v"(tid));
        constexpr int R = ResLast<N, T>::R, CNT = ResLast<N, T>::CNT, NS = N / R, NB = N * (N / R);
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int q = tid + c * T;
            if (CNT * T == NB || q < NB) {
                const int line = q % N, j = q / N;
                const int j0 = (j / NS) * NS * R + j % NS;
#pragma unroll
                for (int m = 0; m < R; ++m) epi.pre(c, m, line, j0 + m * NS);
            }
        }
    }
}

// F^-1 (h .) F   (CONJ: the adjoint step, conj(h)); `epi` rides on the last pass of the inverse transform
struct MidNone { __device__ __forceinline__ void operator()() {} };

// `mid` runs before the LAST line set of the step (the inverse transform along y): where the non-fused plans issue the global
// loads of the next slice.  Issued before the whole step their 4 * EPT registers were live across all eight passes — on top
// of a radix-9 butterfly that is 80 spilled VGPRs for 72^2 — while two passes (~4 us) already cover the HBM latency.
template <int N, int T, bool CONJ, class Epi, class Mid>
__device__ __forceinline__ void res_prop(cf* f, const cf* hT, const cf* tw, int tid, Epi& epi, Mid&& mid) {
    EpiH<N, T, CONJ> eh;
    eh.hT = hT;
    constexpr bool EX = CONJ && BDOF_EX_RES;        // the adjoint step's transforms are exact (bdof_fft.h)
    if constexpr (ResWave<N>::L > 0) {
        // wave-local lines: the inverse transform runs x first, so that a column goes forward, x h, and back in one wave
        static_assert(!Epi::active, "the wave-local form carries no point-wise epilogue");
        EpiNone none;
        res_wlines<N, -1, true, EX>(f, tw, tid, none);
        res_sync();
        res_wlines<N, -1, false, EX>(f, tw, tid, eh);
        res_wlines<N, +1, false, EX>(f, tw, tid, none);
        res_sync();
        {   // the last line set, its two passes written out so that `mid` sits before the lighter (radix R1) one
            typedef ResPlan<N> Pl;
            constexpr int L = ResWave<N>::L, LPW = ResWave<N>::LPW;
            res_wpass<N, Pl::R0, 1, +1, true, L, LPW, EX>(f, tw, tid, none);
            mid();
            res_wpass<N, Pl::R1, Pl::R0, +1, true, L, LPW, EX>(f, tw, tid, none);
        }
        res_sync();
    } else {
        res_fft2<N, T, -1, EX>(f, tw, tid, eh);
        res_epi_prefetch<N, T>(tid, epi);
        EpiNone none;
        res_lines<N, T, +1, true, EX>(f, tw, tid, none);
        mid();
        res_lines<N, T, +1, false, EX>(f, tw, tid, epi);
    }
}
template <int N, int T, bool CONJ, class Epi>
__device__ __forceinline__ void res_prop(cf* f, const cf* hT, const cf* tw, int tid, Epi& epi) {
    res_prop<N, T, CONJ>(f, hT, tw, tid, epi, MidNone());
}
template <int N, int T, bool CONJ> __device__ __forceinline__ void res_prop(cf* f, const cf* hT, const cf* tw, int tid) {
    EpiNone none;
    res_prop<N, T, CONJ>(f, hT, tw, tid, none, MidNone());
}

// Global loads of a slice are issued one propagation step ahead of their use (software pipeline): the modulation factors
// (and, in the adjoint sweep, the tape) of the NEXT slice are in flight while the current slice's transforms run in LDS.
// The object rows feeding each field row (rotation table + window offset) are looked up two slices ahead and parked in a
// 3-slot LDS ring, so that the factor loads themselves never wait on a dependent table load.
template <int N, int T> struct ResPipe {
    static constexpr int P = N | 1;
    static constexpr int EPT = (N * N + T - 1) / T;      // field elements per thread

    // object row of field row x = tid at slice z (-1: outside the volume / the sweep)
    // ... as the element offset of that row in the modulation table (row * volNY: the 64-bit multiply is done once per row
    // and slice by the thread that fills the ring, not once per element by every thread that reads it)
    static __device__ __forceinline__ long long row_of(const ResArgs& a, int b, int z, int tid) {
        if (tid >= N || z < 0 || z >= a.S) return -1;
        const long long r = obj_src_row(a.obj, b, tid, z, N);
        return r >= 0 ? r * a.obj.volNY : -1;
    }
    static __device__ __forceinline__ void load_factors(const ResArgs& a, const long long* rows, int y0, int tid, float2 (&m)[EPT]) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = min(tid + i * T, N * N - 1);
            const int x = res_div<N>(e), y = e - x * N;
            const long long srow = rows[x];
            const int yg = y + y0;
            const int yc = min(max(yg, 0), a.obj.volNY - 1);
            const float2 v = a.obj.vol[(size_t)(srow >= 0 ? srow : 0) + yc];
            const bool in = srow >= 0 && yg == yc;
            m[i] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
        }
    }
    static __device__ __forceinline__ void load_field(const cf* src, int tid, cf (&t)[EPT]) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < EPT; ++i) t[i] = src[min(tid + i * T, N * N - 1)];
    }
};

// Modulation of the NEXT slice folded into the last pass of a propagation step: the element (x, y) leaves the inverse
// transform as psi_{z+1}(x, y), is multiplied by that slice's factor and stored as phi_{z+1} (LDS + tape).
template <int N, int T> struct EpiMod {
    static constexpr bool active = true, pre_in_pass = false;     // HBM loads: issued by res_epi_prefetch, four passes early
    const ResArgs* a;
    const long long* rows;     // object rows of slice z+1
    cf* tape;                  // nullable: tape of slice z+1 for this wavefield
    cf car, csh;
    const cf* pz;              // nullable: carrier field of slice z+1
    int y0;
    float2 fac[ResLast<N, T>::CNT][ResLast<N, T>::R];
    __device__ __forceinline__ void pre(int c, int m, int line, int pos) {
        const long long srow = rows[pos];
        const int yg = line + y0;
        const int yc = min(max(yg, 0), a->obj.volNY - 1);
        const float2 v = a->obj.vol[(size_t)(srow >= 0 ? srow : 0) + yc];
        const bool in = srow >= 0 && yg == yc;
        fac[c][m] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
    }
    __device__ __forceinline__ cf post(int c, int m, int line, int pos, cf v) {
        const cf pc = pz ? pz[pos * N + line] : car;
        const cf phi = modulate_eps_s(v, pc, fac[c][m], csh);
        if (tape) tape[pos * N + line] = pz ? cadd(phi, pc) : phi;      // carrier field: the tape holds the FULL phi (see adjoint)
        return phi;
    }
};

// Point-wise adjoint step of slice z folded into the last pass of the adjoint propagation that precedes it.
template <int N, int T> struct EpiBwd {
    static constexpr bool active = true, pre_in_pass = false;
    const ResArgs* a;
    const long long* rows;     // object rows of slice z
    const cf* tape;
    float2* gdst;
    cf car;
    const cf* pz;              // nullable: carrier field of slice z
    int y0;
    float2 fac[ResLast<N, T>::CNT][ResLast<N, T>::R];
    cf tp[ResLast<N, T>::CNT][ResLast<N, T>::R];
    __device__ __forceinline__ void pre(int c, int m, int line, int pos) {
        const long long srow = rows[pos];
        const int yg = line + y0;
        const int yc = min(max(yg, 0), a->obj.volNY - 1);
        const float2 v = a->obj.vol[(size_t)(srow >= 0 ? srow : 0) + yc];
        const bool in = srow >= 0 && yg == yc;
        fac[c][m] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
        tp[c][m] = tape[pos * N + line];
    }
    __device__ __forceinline__ cf post(int c, int m, int line, int pos, cf G) {
        const cf phi = pz ? tp[c][m] : cadd(tp[c][m], car);            // with a carrier field the tape already holds p + eps
        const cf q = cmulc(G, phi);
        gdst[pos * N + line] = make_float2(a->k * q.y, -a->k * q.x);
        return cmulc(G, make_float2(1.f + fac[c][m].x, fac[c][m].y));
    }
};

// Stand-alone point-wise sweeps over the LDS image (element e = tid + i * T -> (x, y) = (e / N, e % N), coalesced globals).
template <int N, int T> struct ResPoint {
    typedef ResPipe<N, T> Pipe;
    static constexpr int P = N | 1, EPT = Pipe::EPT;

    // phi = c psi : modulation of the slice whose factors are in m; phi goes to the LDS image and to the tape
    // pz: nullable carrier field of the slice ([x][y], read where it is used: an L2-resident table shared by all wavefields)
    static __device__ __forceinline__ void modulate(cf* f, cf* tape, cf car, cf csh, const cf* pz, const float2 (&m)[EPT], int tid) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * T;
            if (EPT * T == N * N || e < N * N) {
                const int x = res_div<N>(e), y = e - x * N;
                const cf pc = pz ? pz[e] : car;
                const cf phi = modulate_eps_s(f[x * P + y], pc, m[i], csh);
                f[x * P + y] = phi;
                if (tape) tape[e] = pz ? cadd(phi, pc) : phi;      // carrier field: the tape holds the FULL phi, so that the
                                                                    // adjoint sweep does not have to read p_z again
            }
        }
        res_sync();
    }
    // adjoint of the modulation: gradient rows out, G(psi) = conj(c) G(phi) left in the LDS image
    static __device__ __forceinline__ void adjoint(cf* f, const cf (&t)[EPT], const float2 (&m)[EPT], cf car, const cf* pz, float k,
                                                   float2* gdst, int tid) {
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * T;
            if (EPT * T == N * N || e < N * N) {
                const int x = res_div<N>(e), y = e - x * N;
                const cf G = f[x * P + y];
                const cf phi = pz ? t[i] : cadd(t[i], car);
                const cf q = cmulc(G, phi);
                gdst[e] = make_float2(k * q.y, -k * q.x);
                f[x * P + y] = cmulc(G, make_float2(1.f + m[i].x, m[i].y));
            }
        }
        res_sync();
    }
};

// WPE = waves per SIMD the register allocation must leave room for (two workgroups per CU where the LDS image allows)
template <int N, int T, int WPE>
__global__ __launch_bounds__(T, WPE) void k_resident(ResArgs a) {
    typedef ResPipe<N, T> Pipe;
    typedef ResPoint<N, T> Point;
    constexpr int P = N | 1, EPT = Pipe::EPT;
    constexpr bool FUSE = ResPlan<N>::FUSE;      // point-wise steps folded into the neighbouring passes (EpiMod / EpiBwd)
    extern __shared__ __align__(16) unsigned char res_smem[];
    cf* f = reinterpret_cast<cf*>(res_smem);
    cf* tw = f + N * P;
    long long* rowbuf = reinterpret_cast<long long*>(tw + 2 * N);  // [3][N] object rows of slices z, z+1, z+2 (ring)
    const int tid = threadIdx.x;
    for (int e = tid; e < 2 * N; e += T) tw[e] = a.twiddle[e];      // hi parts, then lo parts (upload_twiddle)
    const bool far = a.det_mode == BDOF_DET_FAR;
    const size_t fsz = (size_t)N * N;
    // the table of the step after slice z: its dithered copy z mod hD (a fixed float32 table is the same perturbation in every
    // slice; bdof_set_transfer_f64), the adjoint of that step the conjugate of the same copy
    auto hs_of = [&](int z) -> const cf* { return a.hD > 0 ? a.hsT + (size_t)(z % a.hD) * fsz : a.hsT; };
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int y0 = a.obj.yoff ? a.obj.yoff[b] : 0;
        cf* tape0 = a.tape ? a.tape + b * fsz : nullptr;           // slice z of this wavefield: tape0 + z * tape_stride
        res_sync();
        for (int e = tid; e < N * N; e += T) {
            const int x = res_div<N>(e), y = e - x * N;
            f[x * P + y] = a.probe[e];
        }
        if (tid < N) {
            rowbuf[tid] = Pipe::row_of(a, b, 0, tid);
            rowbuf[N + tid] = Pipe::row_of(a, b, 1, tid);
        }
        res_sync();

        // ---- forward sweep --------------------------------------------------------------------
        float2 m[EPT];
        Pipe::load_factors(a, rowbuf, y0, tid, m);
        if constexpr (FUSE) {
            // slice 0 is modulated on its own; every later slice inside the propagation step that produces it (EpiMod)
            Point::modulate(f, tape0, a.carrier[0], a.carrier[a.S], a.pstack, m, tid);
            for (int z = 0; z < a.S; ++z) {
                const long long r2 = Pipe::row_of(a, b, z + 2, tid);
                if (z + 1 < a.S) {
                    EpiMod<N, T> em;
                    em.a = &a;
                    em.rows = rowbuf + ((z + 1) % 3) * N;
                    em.tape = tape0 ? tape0 + (size_t)(z + 1) * a.tape_stride : nullptr;
                    em.car = a.carrier[z + 1];
                    em.csh = a.carrier[a.S + z + 1];
                    em.pz = a.pstack ? a.pstack + (size_t)(z + 1) * fsz : nullptr;
                    em.y0 = y0;
                    res_prop<N, T, false>(f, hs_of(z), tw, tid, em);
                } else if (a.tf_all && !far) {
                    res_prop<N, T, false>(f, hs_of(z), tw, tid);
                }
                if (tid < N) rowbuf[((z + 2) % 3) * N + tid] = r2;      // read two propagation steps from now
            }
        } else {
            EpiNone none;
            for (int z = 0; z < a.S; ++z) {
                const long long r2 = Pipe::row_of(a, b, z + 2, tid);
                Point::modulate(f, tape0 ? tape0 + (size_t)z * a.tape_stride : nullptr, a.carrier[z], a.carrier[a.S + z],
                                a.pstack ? a.pstack + (size_t)z * fsz : nullptr, m, tid);
                if (tid < N) rowbuf[((z + 2) % 3) * N + tid] = r2;
                if (z + 1 < a.S) {
                    // the next slice's factors: in flight during the last two passes of the step
                    res_prop<N, T, false>(f, hs_of(z), tw, tid, none, [&]() { Pipe::load_factors(a, rowbuf + ((z + 1) % 3) * N, y0, tid, m); });
                } else if (a.tf_all && !far) {
                    res_prop<N, T, false>(f, hs_of(z), tw, tid);
                }
            }
        }
        if (a.det_mode == BDOF_DET_NEAR) res_prop<N, T, false>(f, a.hdetT, tw, tid);
        else if (far) res_fft2<N, T, -1>(f, tw, tid);

        // ---- detector wave, loss, seed --------------------------------------------------------
        // The sums of this wavefield are reduced per wave right here and added to the wave's slot of `partial` in global memory
        // (same lane every time: a plain read-modify-write, fixed order).  They used to be carried in registers to the end of
        // the kernel — values that live across both sweeps get spilled, and hipcc 7.2 placed the spill of the zero-initialised
        // accumulators in the flow block of an `if (tid < N)` region, BEFORE exec was restored: waves with no lane in that
        // region never stored them and reloaded whatever an earlier kernel had left in that scratch slot (loss off by 5 % at
        // N = 128, depending on what ran before).  tools/check_spills.py scans the ISA of every kernel for that pattern.
        // The loop is counted and the kind of detector / carrier is one wave-uniform switch: the earlier form (`for (e = tid;
        // e < N * N; e += T)` with a `continue` per case) compiled, at N = 72 after the change above, into code whose sums were
        // wrong for most waves — and right again with a store added to the loop body (a code-generation problem that moved
        // with register allocation, like the spill; the structured form has nothing for it to act on).
        double acc = 0.0, acc2 = 0.0;
        {
            const int mode = !a.meas ? 0 : ((a.meas_dev && !far && !a.pdet) ? 1 : (a.pdet64 ? 2 : 3));
            const float abs_car = sqrtf(a.carrier_det.x * a.carrier_det.x + a.carrier_det.y * a.carrier_det.y);
            int t0 = tid;
            asm volatile("" : "+v"(t0));
#pragma nounroll
            for (int i = 0; i < EPT; ++i) {
                const int e = t0 + i * T;
                if (EPT * T == N * N || e < N * N) {
                    const int x = res_div<N>(e), y = e - x * N;
                    const size_t o = b * fsz + (far ? y * N + x : e);
                    cf d = f[x * P + y], seed;
                    if (mode == 1) {               // plane-wave carrier, real-space detector, residual splitting (loss_seed_dev)
                        seed = loss_seed_dev(d, a.carrier_det, abs_car, a.meas[o], a.seed_scale, acc, acc2, a.dref);
                        d = cadd(d, a.carrier_det);
                    } else if (mode == 2) {        // carrier field in float64: |d| - m in float64 (loss_seed_f64)
                        cf dw;
                        seed = loss_seed_f64(d, a.pdet64[e], a.meas[o], a.seed_scale, acc, acc2, dw);
                        d = dw;
                    } else {
                        if (a.pdet) d = cadd(d, a.pdet[e]);
                        else if (!far || e == 0) d = cadd(d, a.carrier_det);
                        seed = d;
                        if (mode == 3) seed = loss_seed(d, a.meas[o], a.seed_scale, acc, acc2);
                    }
                    if (a.out_wave) a.out_wave[o] = d;
                    if (mode != 0) f[x * P + y] = seed;
                }
            }
        }
        if (a.meas) {
            acc = wave_reduce_sum(acc);
            acc2 = wave_reduce_sum(acc2);
            if ((tid & 63) == 0) {
                double* pw = a.partial + 2 * ((size_t)blockIdx.x * (T / 64) + (tid >> 6));
                const bool first = b == (int)blockIdx.x;
                pw[0] = (first ? 0.0 : pw[0]) + acc;
                pw[1] = (first ? 0.0 : pw[1]) + acc2;
            }
        }
        if (!a.do_grad || !a.meas) continue;

        // ---- adjoint sweep --------------------------------------------------------------------
        if (tid < N) {
            rowbuf[((a.S - 1) % 3) * N + tid] = Pipe::row_of(a, b, a.S - 1, tid);
            rowbuf[((a.S + 1) % 3) * N + tid] = Pipe::row_of(a, b, a.S - 2, tid);      // (S - 2) mod 3
        }
        // The adjoint sweep reads tape elements in a different thread mapping than the forward sweep wrote them whenever a
        // fused plan has CNT * T != N * N / R butterflies (N = 36) — communication between threads through global memory.  Once per
        // wavefield the barrier therefore also drains the stores (vmcnt) and makes them visible workgroup-wide; the
        // per-pass barriers (res_sync) stay LDS-only.
        __threadfence();
        __syncthreads();
        cf t[EPT];
        if constexpr (!FUSE) {
            Pipe::load_factors(a, rowbuf + ((a.S - 1) % 3) * N, y0, tid, m);
            Pipe::load_field(tape0 + (size_t)(a.S - 1) * a.tape_stride, tid, t);
        }
        if (a.det_mode == BDOF_DET_NEAR) res_prop<N, T, true>(f, a.hdetT, tw, tid);
        else if (far) res_fft2<N, T, +1, BDOF_EX_RES>(f, tw, tid);     // F^H = un-normalised inverse
        for (int z = a.S - 1; z >= 0; --z) {
            const long long r2 = Pipe::row_of(a, b, z - 2, tid);
            const bool prop_after = z < a.S - 1 || (a.tf_all && !far);
            const cf* tape = tape0 + (size_t)z * a.tape_stride;
            float2* gdst = a.grot + ((size_t)b * a.S + z) * fsz;
            if constexpr (FUSE) {
                if (prop_after) {
                    EpiBwd<N, T> eb;
                    eb.a = &a;
                    eb.rows = rowbuf + (z % 3) * N;
                    eb.tape = tape;
                    eb.gdst = gdst;
                    eb.car = cadd(a.carrier[z], a.carrier[a.S + z]);      // cbar a_z: constant part of phi_z
                    eb.pz = a.pstack ? a.pstack + (size_t)z * fsz : nullptr;
                    eb.y0 = y0;
                    res_prop<N, T, true>(f, hs_of(z), tw, tid, eb);
                } else {
                    // the slice the adjoint sweep starts from when no transfer-function step follows the last slice
                    Pipe::load_factors(a, rowbuf + (z % 3) * N, y0, tid, m);
                    Pipe::load_field(tape, tid, t);
                    Point::adjoint(f, t, m, cadd(a.carrier[z], a.carrier[a.S + z]), a.pstack ? a.pstack + (size_t)z * fsz : nullptr, a.k, gdst, tid);
                }
            } else {
                if (prop_after) {
                    EpiNone none;
                    if (z < a.S - 1) {
                        // this slice's factors and tape: in flight during the last two passes of the adjoint step
                        res_prop<N, T, true>(f, hs_of(z), tw, tid, none, [&]() {
                            Pipe::load_factors(a, rowbuf + (z % 3) * N, y0, tid, m);
                            Pipe::load_field(tape, tid, t);
                        });
                    } else {
                        res_prop<N, T, true>(f, hs_of(z), tw, tid);       // slice S - 1 was loaded before the detector step
                    }
                }
                Point::adjoint(f, t, m, cadd(a.carrier[z], a.carrier[a.S + z]), a.pstack ? a.pstack + (size_t)z * fsz : nullptr, a.k, gdst, tid);
            }
            if (tid < N) rowbuf[((z + 1) % 3) * N + tid] = r2;           // slot of slice z - 2
        }
        if (a.gpsi0) {                                                   // the image now holds G(psi_0)
            for (int e = tid; e < N * N; e += T) {
                const int x = res_div<N>(e), y = e - x * N;
                a.gpsi0[b * fsz + e] = f[x * P + y];
            }
        }
    }
}

