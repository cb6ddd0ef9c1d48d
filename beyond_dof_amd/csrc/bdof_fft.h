// Device-side 1-D FFT building blocks for the multislice kernels (gfx950, wave64).
//
// One "line" (a wavefield row along y, or a column along x) of N = 2^n complex points is
// transformed by T = N/8 threads, 8 points per thread, with a Stockham autosort plan
// [8, mid..., 8] (mid = radix 2/4/8 stages).  Invariant at BOTH ends of every transform:
// register u[m] of thread `tid` holds the point at natural position tid + m*T.  That is what
// lets the kernels chain inverse-FFT -> pointwise physics -> forward-FFT with no exchange in
// between, and lets global loads/stores stay coalesced (lane-contiguous for fixed m).
// Exchanges between stages go through LDS (index math validated by tools/fft_plan_model.py).
#pragma once
#include <hip/hip_runtime.h>

typedef float2 cf;

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
__device__ __forceinline__ cf cmulc(cf a, cf b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }

// Transform constants.  float32-rounded twiddle tables and float32(sqrt(1/2)) are the SAME small perturbation of every transform
// of a multislice stack: their errors add up coherently along hundreds of slices (gradient error at 512 slices 1.61e-5,
// reconstructed delta against the reference's loop at 256^3 2.1e-5).  Two remedies are built in:
//
//  * dithered constants (the streaming engine's per-slice kernels, default): the host uploads D = 64 copies of each table in which
//    entry j is rounded DOWN or UP so that the mean over the copies is the float64 value to ulp / D, slice z runs with copy
//    z mod D, and sqrt(1/2) arrives per launch the same way (mul_sqrt_half's `sq`).  Plain float32 arithmetic, no extra
//    instruction — and the table errors of neighbouring slices cancel instead of adding up: gradient error at 512 slices 6.0e-6,
//    reconstructed delta 6.6e-6 (G18) / 3.4e-6 (G15) / 7.4e-6 (G19).  BDOF_TW_DITHER=0 in the environment switches it off.
//  * hi + lo pairs (EX = true: table[N + j] = the float32 rounding error of table[j], sqrt(1/2) likewise): every product with a
//    constant is exact to 1e-15 for 2 more FMAs — 6.1e-6 / 5.8e-6 / 3.7e-6 / 7.0e-6 on the same four numbers for +3.2 ms (4.8 %)
//    of the cfg3 step.  The kernels that run once per step (detector plane, loss) use it; -DBDOF_EXACT_TRANSFORMS builds the
//    per-slice kernels with it too (round 3's first default, kept for comparison).
#if defined(BDOF_EXACT_TWIDDLES) || defined(BDOF_FAST_ADJOINT) || defined(BDOF_FAST_FORWARD)
#error "BDOF_EXACT_TWIDDLES / BDOF_FAST_ADJOINT / BDOF_FAST_FORWARD (rounds 2-3) no longer exist: the per-slice kernels run with dithered constants; -DBDOF_EXACT_TRANSFORMS builds them with hi + lo tables"
#endif
#ifndef BDOF_EXACT_CONSTANTS
#define BDOF_EXACT_CONSTANTS 1          // sqrt(1/2) as a hi + lo pair wherever no dithered value is handed in (mul_sqrt_half)
#endif
#ifdef BDOF_EXACT_TRANSFORMS
constexpr bool BDOF_EX_ALL = true;
#else
constexpr bool BDOF_EX_ALL = false;
#endif
constexpr bool BDOF_EX_ADJ = BDOF_EX_ALL;       // adjoint sweep (A', B')
constexpr bool BDOF_EX_FWD_A = BDOF_EX_ALL;     // forward sweep
constexpr bool BDOF_EX_FWD_B = BDOF_EX_ALL;
constexpr bool BDOF_EX_DET = true;              // once per step: detector plane, loss, seed
// u * (w + wl), the twiddle conjugated for the inverse transform
template <int SIGN, bool EX> __device__ __forceinline__ cf tw_mul(cf u, cf w, cf wl) {
    if (SIGN > 0) { w.y = -w.y; wl.y = -wl.y; }
    if constexpr (EX) {
        // u w + u wl with the small product innermost: 2 mul + 6 fma instead of two complex multiplies and an add
        const float lx = fmaf(u.x, wl.x, -(u.y * wl.y)), ly = fmaf(u.x, wl.y, u.y * wl.x);
        return make_float2(fmaf(u.x, w.x, fmaf(-u.y, w.y, lx)), fmaf(u.x, w.y, fmaf(u.y, w.x, ly)));
    } else return cmul(u, w);
}

// multiply by SIGN*i
template <int SIGN> __device__ __forceinline__ cf mul_si(cf a) {
    return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int SIGN> __device__ __forceinline__ void dft2(cf& a, cf& b) {
    cf t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// natural-order outputs X0..X3 left in a0..a3
template <int SIGN> __device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = mul_si<SIGN>(csub(a1, a3));
    a0 = cadd(s0, s2);
    a2 = csub(s0, s2);
    a1 = cadd(s1, s3);
    a3 = csub(s1, s3);
}

// Irrational butterfly constants.  float32(sqrt(1/2)) is 1.7e-8 short (float32(sqrt(3)/2) 1.8e-8): every product with it
// shrinks the amplitude by that much, the defects add up along a chain of transforms and a multislice stack drifts in
// energy (-1.25e-7 per slice at 72^2).  The remedies:
//   sq != nullptr (the per-slice kernels of the streaming engine): the constant of THIS launch comes from the host, sq[0] for the
//      transforms instantiated with ROUND 1 and sq[1] for those with ROUND 2 — the host walks the two neighbouring float32 values
//      over the slices so that their mean is sqrt(1/2) (dithered constants, above), at no cost in instructions;
//   ROUND 0, or BDOF_EXACT_CONSTANTS (the default) without sq: hi + lo pair, one more FMA per product, exact to 1e-15;
//   without BDOF_EXACT_CONSTANTS (-UBDOF_EXACT_CONSTANTS is not offered; round 2's scheme, kept for the record): ROUND 1 = the
//      nearest float32 (rounds DOWN), ROUND 2 = its upper neighbour; a kernel whose propagation step runs four line transforms uses
//      2 in one of them and 1 in the other three: (3 x -1.71 + 6.72)e-8 — the defects cancel to a quarter.
template <int ROUND_> __device__ __forceinline__ float mul_sqrt_half(float t, const float* sq = nullptr) {
    if constexpr (ROUND_ == 0) return fmaf(t, 0.70710678118654752f, t * 1.2101617e-8f);
    else if (sq) return t * sq[ROUND_ - 1];
    else {
#ifdef BDOF_EXACT_CONSTANTS
        return fmaf(t, 0.70710678118654752f, t * 1.2101617e-8f);
#else
        if constexpr (ROUND_ == 1) return t * 0.70710678118654752f;
        else return t * 0.70710682868957520f;
#endif
    }
}

template <int SIGN, int ROUND = 1>
__device__ __forceinline__ void dft8(cf& a0, cf& a1, cf& a2, cf& a3, cf& a4, cf& a5, cf& a6, cf& a7, const float* sq = nullptr) {
    dft4<SIGN>(a0, a2, a4, a6);   // E0..E3 -> a0,a2,a4,a6
    dft4<SIGN>(a1, a3, a5, a7);   // O0..O3 -> a1,a3,a5,a7
    const float s = (float)SIGN;
    cf o0 = a1;
    cf o1 = make_float2(mul_sqrt_half<ROUND>(a3.x - s * a3.y, sq), mul_sqrt_half<ROUND>(a3.y + s * a3.x, sq));       // * (1 + s i)/sqrt2
    cf o2 = mul_si<SIGN>(a5);                                                                                        // * s i
    cf o3 = make_float2(mul_sqrt_half<ROUND>(-a7.x - s * a7.y, sq), mul_sqrt_half<ROUND>(-a7.y + s * a7.x, sq));     // * (-1 + s i)/sqrt2
    cf e0 = a0, e1 = a2, e2 = a4, e3 = a6;
    a0 = cadd(e0, o0); a4 = csub(e0, o0);
    a1 = cadd(e1, o1); a5 = csub(e1, o1);
    a2 = cadd(e2, o2); a6 = csub(e2, o2);
    a3 = cadd(e3, o3); a7 = csub(e3, o3);
}

// ---------------------------------------------------------------------------------------------
// Plans.  radix of stage s and p = product of earlier radices.
// ---------------------------------------------------------------------------------------------
template <int N> struct FftPlan;
template <> struct FftPlan<64>   { static constexpr int NS = 2, R0 = 8, R1 = 8, R2 = 1, R3 = 1; };
template <> struct FftPlan<128>  { static constexpr int NS = 3, R0 = 8, R1 = 2, R2 = 8, R3 = 1; };
template <> struct FftPlan<256>  { static constexpr int NS = 3, R0 = 8, R1 = 4, R2 = 8, R3 = 1; };
template <> struct FftPlan<512>  { static constexpr int NS = 3, R0 = 8, R1 = 8, R2 = 8, R3 = 1; };
template <> struct FftPlan<1024> { static constexpr int NS = 4, R0 = 8, R1 = 2, R2 = 8, R3 = 8; };

// Twiddles.  The LAST stage's factors exp(-2 pi i m tid / N) depend on the lane and stay in registers (7 values);
// the middle stages' factors depend only on k = butterfly % p (8 or 16 distinct lanes' worth) and live in a small
// LDS table [k][m-1] (row length R-1 is odd -> the distinct k of a wave hit distinct banks, equal k broadcast).
template <int N> struct FftTw {
    typedef FftPlan<N> P;
    static constexpr int RL = P::NS == 2 ? P::R1 : (P::NS == 3 ? P::R2 : P::R3);      // last radix (8)
    static constexpr int PL = N / RL;                                                  // its p
    // middle stages: (R1, p = R0) when NS > 2; (R2, p = R0*R1) when NS > 3
    static constexpr int M1_R = P::NS > 2 ? P::R1 : 1, M1_P = P::R0;
    static constexpr int M2_R = P::NS > 3 ? P::R2 : 1, M2_P = P::R0 * P::R1;
    static constexpr int M1_OFF = 0, M1_CNT = P::NS > 2 ? M1_P * (M1_R - 1) : 0;
    static constexpr int M2_OFF = M1_CNT, M2_CNT = P::NS > 3 ? M2_P * (M2_R - 1) : 0;
    static constexpr int LDS_CNT = M1_CNT + M2_CNT > 0 ? M1_CNT + M2_CNT : 1;
    static constexpr int D1 = M1_R > 1 ? M1_R - 1 : 1, D2 = M2_R > 1 ? M2_R - 1 : 1;   // row lengths (never 0 as divisors)
    cf w[7];            // last stage, forward sign
    const cf* mid;      // LDS table of the middle stages
    const cf* tail;     // LDS table [m-1][j] of the last stage (load<true>: its lo parts follow at + 7 T, the middle
                        // stages' at mid + LDS_CNT)
    const float* sq = nullptr;      // dithered sqrt(1/2) of this launch (mul_sqrt_half); nullptr: the compiled-in constants

    // table[j] = exp(-2 pi i j / N), j in [0, N).  Must be called by every thread of the workgroup.
    // lds_mid: LDS_CNT slots; lds_tail: 7*N/8 slots laid out [m-1][j] = table[m*j] (also the tail stage's twiddles):
    // one cooperative fill, one barrier, and the per-lane last-stage factors are read back from LDS instead of being
    // gathered from global memory by every wave (kernel start-up is amortised over only ~2 tiles per workgroup).
    template <bool LO = BDOF_EX_ALL>
    __device__ __forceinline__ void load(const cf* __restrict__ table, int tid, cf* lds_mid, cf* lds_tail) {
        constexpr int T = N / 8;
        for (int e = threadIdx.x; e < 7 * T; e += blockDim.x) {
            const int m = e / T + 1, j = e - (m - 1) * T;
            lds_tail[e] = table[m * j];
            if constexpr (LO) lds_tail[7 * T + e] = table[N + m * j];
        }
        for (int e = threadIdx.x; e < M1_CNT + M2_CNT; e += blockDim.x) {
            int k, m, step;
            if (e < M1_CNT) { k = e / D1; m = e % D1 + 1; step = N / (M1_P * M1_R); }
            else { const int f = e - M1_CNT; k = f / D2; m = f % D2 + 1; step = N / (M2_P * M2_R); }
            lds_mid[e] = table[m * k * step];
            if constexpr (LO) lds_mid[LDS_CNT + e] = table[N + m * k * step];
        }
        mid = lds_mid;
        tail = lds_tail;
        __syncthreads();
#pragma unroll
        for (int m = 1; m < 8; ++m) w[m - 1] = lds_tail[(m - 1) * T + tid];
    }
};

template <int R, int SIGN, int ROUND = 1> __device__ __forceinline__ void dftR(cf (&u)[8], int j, const float* sq = nullptr) {
    if constexpr (R == 8) dft8<SIGN, ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], sq);
    else if constexpr (R == 4) {
        if (j == 0) dft4<SIGN>(u[0], u[1], u[2], u[3]); else dft4<SIGN>(u[4], u[5], u[6], u[7]);
    } else {
        if (j == 0) dft2<SIGN>(u[0], u[1]); else if (j == 1) dft2<SIGN>(u[2], u[3]);
        else if (j == 2) dft2<SIGN>(u[4], u[5]); else dft2<SIGN>(u[6], u[7]);
    }
}

// One stage, split in its three phases.  L provides slot(idx) = idx + idx/16 and ld_at / st_at(base_slot, c),
// which address slot(base + c) as base_slot + c + c/16: for every (base, c) pair the plans generate the low
// four bits never carry (checked exhaustively in tools/fft_plan_model.py), so each access is one base register
// plus an instruction-immediate offset.
template <int N, int R, class L> __device__ __forceinline__ void stage_read(cf (&u)[8], int tid, L& lds) {
    constexpr int T = N / 8, NB = 8 / R, TT = N / R;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int bs = lds.slot(tid + j * T);
#pragma unroll
        for (int m = 0; m < R; ++m) u[j * R + m] = lds.ld_at(bs, m * TT);
    }
}

// WHICH: 0 = first stage (no twiddles), 1 / 2 = middle stage (LDS table), 3 = last stage (registers)
template <int N, int SIGN, int R, int PP, int WHICH, int ROUND = 1, bool EX = BDOF_EX_ALL>
__device__ __forceinline__ void stage_compute(cf (&u)[8], const FftTw<N>& tw, int tid) {
    constexpr int T = N / 8, NB = 8 / R;
    typedef FftTw<N> TW;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if constexpr (WHICH == 3) {
#pragma unroll
            for (int m = 1; m < R; ++m)
                u[j * R + m] = tw_mul<SIGN, EX>(u[j * R + m], tw.w[m - 1], EX ? tw.tail[(7 + m - 1) * T + tid] : tw.w[m - 1]);
        } else if constexpr (WHICH == 1 || WHICH == 2) {
            const int k = (tid + j * T) % PP;
            const cf* row = tw.mid + (WHICH == 1 ? TW::M1_OFF : TW::M2_OFF) + k * (R - 1);
#pragma unroll
            for (int m = 1; m < R; ++m) u[j * R + m] = tw_mul<SIGN, EX>(u[j * R + m], row[m - 1], row[(EX ? TW::LDS_CNT : 0) + m - 1]);
        }
        dftR<R, SIGN, EX ? 0 : ROUND>(u, j, tw.sq);
    }
}

template <int N, int R, int PP, class L> __device__ __forceinline__ void stage_write(const cf (&u)[8], int tid, L& lds) {
    constexpr int T = N / 8, NB = 8 / R;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i = tid + j * T;
        const int k = i % PP;
        const int bs = lds.slot((i - k) * R + k);
#pragma unroll
        for (int q = 0; q < R; ++q) lds.st_at(bs, q * PP, u[j * R + q]);
    }
}

// Full transform of one line.  u[m] <-> position tid + m*T on entry and on exit.
// Unnormalised; SIGN = -1 forward DFT, +1 inverse.  lds.sync_w2r() orders a stage's stores
// before the next stage's loads; lds.sync_r2w() orders loads before the stores that reuse the image.
template <int N, int SIGN, int ROUND = 1, bool EX = BDOF_EX_ALL, class L>
__device__ __forceinline__ void line_fft(cf (&u)[8], const FftTw<N>& tw, int tid, L& lds) {
    typedef FftPlan<N> P;
    stage_compute<N, SIGN, P::R0, 1, 0, ROUND, EX>(u, tw, tid);
    lds.sync_r2w();
    stage_write<N, P::R0, 1>(u, tid, lds);
    lds.sync_w2r();
    stage_read<N, P::R1>(u, tid, lds);
    stage_compute<N, SIGN, P::R1, P::R0, (P::NS == 2 ? 3 : 1), ROUND, EX>(u, tw, tid);
    if constexpr (P::NS > 2) {
        lds.sync_r2w();
        stage_write<N, P::R1, P::R0>(u, tid, lds);
        lds.sync_w2r();
        stage_read<N, P::R2>(u, tid, lds);
        stage_compute<N, SIGN, P::R2, P::R0 * P::R1, (P::NS == 3 ? 3 : 2), ROUND, EX>(u, tw, tid);
    }
    if constexpr (P::NS > 3) {
        lds.sync_r2w();
        stage_write<N, P::R2, P::R0 * P::R1>(u, tid, lds);
        lds.sync_w2r();
        stage_read<N, P::R3>(u, tid, lds);
        stage_compute<N, SIGN, P::R3, P::R0 * P::R1 * P::R2, 3, ROUND, EX>(u, tw, tid);
    }
}

// All stages but the last; the last stage's inputs are left in the line's LDS image (no sync after the
// final store: the caller's workgroup barrier orders it before the transposed readers).
template <int N, int SIGN, int ROUND = 1, bool EX = BDOF_EX_ALL, class L>
__device__ __forceinline__ void line_fft_partial(cf (&u)[8], const FftTw<N>& tw, int tid, L& lds) {
    typedef FftPlan<N> P;
    stage_compute<N, SIGN, P::R0, 1, 0, ROUND, EX>(u, tw, tid);
    lds.sync_r2w();
    stage_write<N, P::R0, 1>(u, tid, lds);
    if constexpr (P::NS > 2) {
        lds.sync_w2r();
        stage_read<N, P::R1>(u, tid, lds);
        stage_compute<N, SIGN, P::R1, P::R0, 1, ROUND, EX>(u, tw, tid);
        lds.sync_r2w();
        stage_write<N, P::R1, P::R0>(u, tid, lds);
    }
    if constexpr (P::NS > 3) {
        lds.sync_w2r();
        stage_read<N, P::R2>(u, tid, lds);
        stage_compute<N, SIGN, P::R2, P::R0 * P::R1, 2, ROUND, EX>(u, tw, tid);
        lds.sync_r2w();
        stage_write<N, P::R2, P::R0 * P::R1>(u, tid, lds);
    }
}

// The last stage (radix 8, p = N/8) for butterfly j of a line whose image is `lds`; `tail` is the LDS copy
// [m-1][j] of the twiddles exp(-2 pi i m j / N) (fill_tail_table).  On exit u[q] is the output at position j + q*N/8.
template <int N, int SIGN, int ROUND = 1, bool EX = BDOF_EX_ALL, class L>
__device__ __forceinline__ void last_stage(cf (&u)[8], int j, L& lds, const cf* tail, const float* sq = nullptr) {
    constexpr int T = N / 8;
    cf w[7], wl[7];
#pragma unroll
    for (int m = 1; m < 8; ++m) {
        w[m - 1] = tail[(m - 1) * T + j];
        wl[m - 1] = tail[((EX ? 7 : 0) + m - 1) * T + j];
    }
    stage_read<N, 8>(u, j, lds);
#pragma unroll
    for (int m = 1; m < 8; ++m) u[m] = tw_mul<SIGN, EX>(u[m], w[m - 1], wl[m - 1]);
    dft8<SIGN, EX ? 0 : ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], sq);
}
