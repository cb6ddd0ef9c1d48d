// Experimental variants of the propagation row kernel for A/B timing in kbench (development aid).
#pragma once
#include "../beyond_dof_amd/csrc/bdof_kernels.h"
#include "experiments/bdof_fft_reg.h"

// MODE 0: full; 1: memory only (load, one LDS hop for the transposition, transposed store; no butterflies);
// 2: compute only (no global loads of u/h, result stored by one lane only); 3: full but plain (non-transposed) store
// 4: full with the stage exchanges in registers (bdof_fft_reg.h; h read as if laid out in the permuted order: timing);
// 5: the same with h gathered through fft512_perm from the natural-order table (correct results, for the comparison with 0)
template <int NX, int MODE>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NX>::MIN_WAVES) void kv_prop(RowPropArgs a) {
    typedef RowCfg<NX> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    __shared__ cf smem_tw[FftTw<NX>::LDS_CNT];
    FftTw<NX> tw;
    __shared__ cf smem_tail[7 * C::T];
    tw.load(a.twiddle, tid, smem_tw, smem_tail);
    const int ntiles = a.B * a.NY / C::TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NY, ky0 = row0 - b * a.NY;
#pragma nounroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            RowLds<C::T> lds{smem + r * C::RS};
            cf u[8], hv[8];
            const cf* src = a.in + (size_t)(row0 + r) * NX;
            const cf* hrow = a.h + (size_t)(ky0 + r) * NX;
            if (MODE != 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = src[tid + m * C::T];
#pragma unroll
                for (int m = 0; m < 8; ++m) hv[m] = hrow[tid + m * C::T];
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) { u[m] = make_float2(tid * 0.001f + m, 1.f - m); hv[m] = make_float2(0.5f, 0.25f * m); }
            }
            if constexpr ((MODE == 4 || MODE == 5) && NX == 512) {
                if (MODE == 5) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) hv[m] = hrow[fft512_perm(tid, m)];
                }
                fft512_reg_forward<1, false>(u, tw, tid);
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = cmul(u[m], cscale(hv[m], a.scale));
                fft512_reg_inverse_partial<2, false>(u, tw, tid, lds);
            } else
            if (MODE == 1) {
                const int bs = lds.slot(tid);
#pragma unroll
                for (int m = 0; m < 8; ++m) lds.st_at(bs, m * C::T, cmul(u[m], hv[m]));
            } else {
                line_fft<NX, -1>(u, tw, tid, lds);
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = cmul(u[m], cscale(hv[m], a.scale));
                if (MODE == 3) {
                    line_fft<NX, +1>(u, tw, tid, lds);
                    cf* dst = a.out + (size_t)(row0 + r) * NX;
#pragma unroll
                    for (int m = 0; m < 8; ++m) dst[tid + m * C::T] = u[m];
                } else {
                    line_fft_partial<NX, +1>(u, tw, tid, lds);
                }
            }
        }
        if (MODE != 3) {
            __syncthreads();
            if (MODE == 1) {
#pragma nounroll
                for (int pass = 0; pass < C::PASSES; ++pass) {
                    const int q = threadIdx.x + pass * BDOF_THREADS;
                    const int r = q % C::TILE, j = q / C::TILE;
                    RowLds<C::T> lds{smem + r * C::RS};
                    cf* dst = a.out + (size_t)b * NX * a.NY + ky0;
                    const int bs = lds.slot(j);
#pragma unroll
                    for (int m = 0; m < 8; ++m) dst[(size_t)(j + m * C::T) * a.NY + r] = lds.ld_at(bs, m * C::T);
                }
            } else if (MODE == 2) {
#pragma nounroll
                for (int pass = 0; pass < C::PASSES; ++pass) {
                    const int q = threadIdx.x + pass * BDOF_THREADS;
                    const int r = q % C::TILE, j = q / C::TILE;
                    RowLds<C::T> lds{smem + r * C::RS};
                    cf u[8];
                    last_stage<NX, +1>(u, j, lds, smem_tail);
                    cf acc = u[0];
#pragma unroll
                    for (int m = 1; m < 8; ++m) acc = cadd(acc, u[m]);
                    if (acc.x == 123.456f) a.out[q] = acc;       // keeps the work alive, practically never stores
                }
            } else {
                transposed_tail<NX, +1>(smem, a.out + (size_t)b * NX * a.NY + ky0, a.NY, 1.f, smem_tail);
            }
            __syncthreads();
        }
    }
}
