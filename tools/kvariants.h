// Experimental variants of the column kernel for A/B timing in kbench (development aid).
#pragma once
#include "../beyond_dof_amd/csrc/bdof_kernels.h"

// MODE 0: full; 1: load+store only; 2: load + LDS exchanges (no butterflies) + store; 3: full but no h multiply
template <int NX, int MODE>
__global__ __launch_bounds__((NX / 8) * ColTile<NX>::W) void kv_col(ColPropArgs a) {
    constexpr int T = NX / 8, W = ColTile<NX>::W;
    __shared__ cf smem[NX * W];
    const int w = threadIdx.x % W, i = threadIdx.x / W;
    ColLds<W> lds{smem, w};
    FftTw<NX> tw;
    if (MODE == 0 || MODE == 3) tw.load(a.twiddle, i);
    const int tiles_per_b = a.NY / W;
    const int ntiles = a.B * tiles_per_b;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int y = (tile - b * tiles_per_b) * W + w;
        const size_t base = (size_t)b * NX * a.NY + y;
        cf u[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = a.in[base + (size_t)(i + m * T) * a.NY];
        if (MODE == 0 || MODE == 3) {
            line_fft<NX, -1>(u, tw, i, lds);
            if (MODE == 0) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    cf hv = a.h[(size_t)(i + m * T) * a.NY + y];
                    if (a.conj_h) hv.y = -hv.y;
                    u[m] = cmul(u[m], cscale(hv, a.scale));
                }
            }
            line_fft<NX, +1>(u, tw, i, lds);
        } else if (MODE == 2) {
            for (int rep = 0; rep < 4; ++rep) {
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 8; ++m) lds.st(i * 8 + m, u[m]);
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = lds.ld(i + m * T);
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) a.out[base + (size_t)(i + m * T) * a.NY] = u[m];
    }
}
