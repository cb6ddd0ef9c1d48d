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

// Tile order experiment (round 4): which tiles a persistent workgroup takes.  ORDER 0: blockIdx + i * grid (production);
// 1: the adjacent pair 2 * blockIdx + i (its two 128-byte store segments of every output row make one 256-byte block);
// 2: adjacent pairs, and the two pairs of an XCD's consecutive workgroups adjacent too (512 bytes of every output row per XCD);
// 3: as 0 but the second tile first loaded before the first one's tail (row loads hoisted across the tile barrier)
template <int NX, int ORDER>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NX>::MIN_WAVES) void kv_prop_order(RowPropArgs a) {
    typedef RowCfg<NX> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    __shared__ cf smem_tw[FftTw<NX>::LDS_CNT];
    FftTw<NX> tw;
    __shared__ cf smem_tail[7 * C::T];
    tw.load(a.twiddle, tid, smem_tw, smem_tail);
    tw.sq = a.sq;
    const int ntiles = a.B * a.NY / C::TILE;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    for (int i = 0; i < per; ++i) {
        int tile;
        if (ORDER == 1) tile = per * blockIdx.x + i;
        else if (ORDER == 2) {
            const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
            const int p = (k >> 1) * 16 + xcd * 2 + (k & 1);
            tile = per * p + i;
        } else tile = blockIdx.x + i * gridDim.x;
        if (tile >= ntiles) break;
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NY, ky0 = row0 - b * a.NY;
#pragma nounroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            RowLds<C::T> lds{smem + r * C::RS};
            cf u[8], hv[8];
            const cf* src = a.in + (size_t)(row0 + r) * NX;
            const cf* hrow = a.h + (size_t)(ky0 + r) * NX;
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = src[tid + m * C::T];
#pragma unroll
            for (int m = 0; m < 8; ++m) hv[m] = hrow[tid + m * C::T];
            line_fft<NX, -1>(u, tw, tid, lds);
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = cmul(u[m], cscale(hv[m], a.scale));
            line_fft_partial<NX, +1, 2>(u, tw, tid, lds);
        }
        __syncthreads();
        transposed_tail<NX, +1, 2>(smem, a.out + (size_t)b * NX * a.NY + ky0, a.NY, 1.f, smem_tail, a.sq);
        __syncthreads();
    }
}

// Memory-only skeleton with the transposition on the LOAD side (round 4 question: stores like one contiguous piece per workgroup —
// 5.9 against 4.5-5.0 TB/s in the calibration kernels — so would "strided 128-byte reads, whole-row writes" beat the production
// "whole-row reads, strided 128-byte writes"?).  Tile = 16 lines (columns of the input, 128-byte segments of every one of its
// N rows) -> LDS -> 16 whole output rows.  TREAD 0: the production direction (as kv_prop MODE 1), 1: transposed reads.
template <int NX, int TREAD>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NX>::MIN_WAVES) void kv_prop_side(RowPropArgs a) {
    typedef RowCfg<NX> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    const int ntiles = a.B * a.NY / C::TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NY, ky0 = row0 - b * a.NY;
        if (TREAD) {
            // in: [b][x][ky] (ld = NY); line r of the tile = column ky0 + r; element j + m T of the line = row j + m T
            const cf* src = a.in + (size_t)b * NX * a.NY + ky0;
#pragma nounroll
            for (int pass = 0; pass < C::PASSES; ++pass) {
                const int q = threadIdx.x + pass * BDOF_THREADS;
                const int r = q % C::TILE, j = q / C::TILE;
                RowLds<C::T> lds{smem + r * C::RS};
                const int bs = lds.slot(j);
                cf u[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = src[(size_t)(j + m * C::T) * a.NY + r];
#pragma unroll
                for (int m = 0; m < 8; ++m) lds.st_at(bs, m * C::T, u[m]);
            }
            __syncthreads();
#pragma nounroll
            for (int pass = 0; pass < C::PASSES; ++pass) {
                const int r = pass * C::RPP + rl;
                RowLds<C::T> lds{smem + r * C::RS};
                const cf* hrow = a.h + (size_t)(ky0 + r) * NX;
                cf* dst = a.out + (size_t)(row0 + r) * NX;
                const int bs = lds.slot(tid);
                cf hv[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) hv[m] = hrow[tid + m * C::T];
#pragma unroll
                for (int m = 0; m < 8; ++m) dst[tid + m * C::T] = cmul(lds.ld_at(bs, m * C::T), hv[m]);
            }
            __syncthreads();
        } else {
#pragma nounroll
            for (int pass = 0; pass < C::PASSES; ++pass) {
                const int r = pass * C::RPP + rl;
                RowLds<C::T> lds{smem + r * C::RS};
                cf u[8], hv[8];
                const cf* src = a.in + (size_t)(row0 + r) * NX;
                const cf* hrow = a.h + (size_t)(ky0 + r) * NX;
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = src[tid + m * C::T];
#pragma unroll
                for (int m = 0; m < 8; ++m) hv[m] = hrow[tid + m * C::T];
                const int bs = lds.slot(tid);
#pragma unroll
                for (int m = 0; m < 8; ++m) lds.st_at(bs, m * C::T, cmul(u[m], hv[m]));
            }
            __syncthreads();
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
            __syncthreads();
        }
    }
}
