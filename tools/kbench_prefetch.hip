// Development aid (not part of the product): the transfer-function kernel with a software pipeline over its rows — the next
// row's 8 points per lane are requested before the current row's transforms run (16 more registers), barriers wait on LDS only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/kbench_prefetch tools/kbench_prefetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../beyond_dof_amd/csrc/bdof_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static int balanced(int tiles, int cap) { if (tiles <= cap) return tiles; int r = (tiles + cap - 1) / cap; return (tiles + r - 1) / r; }

template <int NX, bool HPRE>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NX>::MIN_WAVES) void kp_prop(RowPropArgs a) {
    typedef RowCfg<NX> C;
    constexpr bool EX = false;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    __shared__ cf smem_tw[FftTw<NX>::LDS_CNT];
    FftTw<NX> tw;
    __shared__ cf smem_tail[7 * C::T];
    const int ntiles = a.B * a.NY / C::TILE;
    cf un[8], hn[8];
    if ((int)blockIdx.x < ntiles) {
        const int row0 = blockIdx.x * C::TILE;
        const cf* src = a.in + (size_t)(row0 + rl) * NX;
#pragma unroll
        for (int m = 0; m < 8; ++m) un[m] = src[tid + m * C::T];
        if (HPRE) {
            const cf* hrow = a.h + (size_t)(row0 % a.NY + rl) * NX;
#pragma unroll
            for (int m = 0; m < 8; ++m) hn[m] = hrow[tid + m * C::T];
        }
    }
    tw.template load<EX>(a.twiddle, tid, smem_tw, smem_tail);
    tw.sq = a.sq;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NY, ky0 = row0 - b * a.NY;
#pragma unroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            RowLds<C::T> lds{smem + r * C::RS};
            cf u[8], hv[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = un[m];
            if (HPRE) {
#pragma unroll
                for (int m = 0; m < 8; ++m) hv[m] = hn[m];
            }
            if (!HPRE) {        // h first: vmcnt retires in order, and waiting for h must not wait for the next row
                const cf* hrow = a.h + (size_t)(ky0 + r) * NX;
#pragma unroll
                for (int m = 0; m < 8; ++m) hv[m] = hrow[tid + m * C::T];
            }
            // the row after this one: the next pass of the tile, or the first pass of this workgroup's next tile
            const int nrow = pass + 1 < C::PASSES ? row0 + (pass + 1) * C::RPP + rl : (tile + (int)gridDim.x) * C::TILE + rl;
            if (pass + 1 < C::PASSES || tile + (int)gridDim.x < ntiles) {
                const cf* src = a.in + (size_t)nrow * NX;
#pragma unroll
                for (int m = 0; m < 8; ++m) un[m] = src[tid + m * C::T];
                if (HPRE) {
                    const cf* hrow = a.h + (size_t)(nrow % a.NY) * NX;
#pragma unroll
                    for (int m = 0; m < 8; ++m) hn[m] = hrow[tid + m * C::T];
                }
            }
            line_fft<NX, -1, 1, EX>(u, tw, tid, lds);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                cf t = hv[m];
                if (a.conj_h) t.y = -t.y;
                u[m] = cmul(u[m], cscale(t, a.scale));
            }
            line_fft_partial<NX, +1, 2, EX>(u, tw, tid, lds);
        }
        conv_sync();                       // LDS only: the row in flight stays in flight
        transposed_tail<NX, +1, 2, EX>(smem, a.out + (size_t)b * NX * a.NY + ky0, a.NY, 1.f, smem_tail, a.sq);
        conv_sync();
    }
}

// XCD-aware tile order for the same kernel: workgroups are dealt to the 8 XCDs round-robin, so with tile = blockIdx.x + k * grid
// the 32 tiles of a wavefield — which write the 32 adjacent 128-byte segments of every output row — sit in 8 different L2s.
// Here every XCD takes a contiguous eighth of the tiles (whole wavefields), walked in order by its workgroups.
template <int NX>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NX>::MIN_WAVES) void kx_prop(RowPropArgs a) {
    typedef RowCfg<NX> C;
    constexpr bool EX = false;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    __shared__ cf smem_tw[FftTw<NX>::LDS_CNT];
    FftTw<NX> tw;
    __shared__ cf smem_tail[7 * C::T];
    tw.template load<EX>(a.twiddle, tid, smem_tw, smem_tail);
    tw.sq = a.sq;
    const int ntiles = a.B * a.NY / C::TILE;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, nw = ((int)gridDim.x - xcd + 7) >> 3;
    const int t0 = (int)((long long)xcd * ntiles / 8), t1 = (int)((long long)(xcd + 1) * ntiles / 8);
    for (int tile = t0 + w; tile < t1; tile += nw) {
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
            line_fft<NX, -1, 1, EX>(u, tw, tid, lds);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                cf t = hv[m];
                if (a.conj_h) t.y = -t.y;
                u[m] = cmul(u[m], cscale(t, a.scale));
            }
            line_fft_partial<NX, +1, 2, EX>(u, tw, tid, lds);
        }
        __syncthreads();
        transposed_tail<NX, +1, 2, EX>(smem, a.out + (size_t)b * NX * a.NY + ky0, a.NY, 1.f, smem_tail, a.sq);
        __syncthreads();
    }
}

template <class F> static float run(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / iters;
}

int main() {
    constexpr int N = 512;
    const int Bmax = 100;
    const size_t fld = (size_t)Bmax * N * N;
    cf *in, *out, *out2, *h, *tw;
    CK(hipMalloc(&in, fld * 8)); CK(hipMalloc(&out, fld * 8)); CK(hipMalloc(&out2, fld * 8));
    CK(hipMalloc(&h, (size_t)N * N * 8)); CK(hipMalloc(&tw, 2 * N * 8));
    std::vector<cf> hw(2 * N, make_float2(0.f, 0.f));
    for (int j = 0; j < N; ++j) hw[j] = make_float2((float)cos(-2 * M_PI * j / N), (float)sin(-2 * M_PI * j / N));
    CK(hipMemcpy(tw, hw.data(), 2 * N * 8, hipMemcpyHostToDevice));
    std::vector<float> rnd(fld * 2);
    for (auto& v : rnd) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(in, rnd.data(), fld * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(h, rnd.data(), (size_t)N * N * 8, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    for (int B : {25, 100}) {
        RowPropArgs a{in, out, h, B, N, 1.f / (N * N), 0, tw, {0.70710678f, 0.70710678f}};
        RowPropArgs a2 = a; a2.out = out2;
        const int tiles = B * N / RowCfg<N>::TILE, grid = balanced(tiles, ncu * 2);
        const int iters = 40;
        for (int rep = 0; rep < 2; ++rep) {
            const float t0 = run([&] { hipLaunchKernelGGL((k_row_prop<N, false>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters);
            const float t1 = run([&] { hipLaunchKernelGGL((kp_prop<N, false>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a2); }, iters);
            const float t2 = run([&] { hipLaunchKernelGGL((kp_prop<N, true>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a2); }, iters);
            const float t3 = run([&] { hipLaunchKernelGGL((kp_prop<N, false>), dim3(ncu * 2), dim3(BDOF_THREADS), 0, 0, a2); }, iters);
            const float t4 = run([&] { hipLaunchKernelGGL((kx_prop<N>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a2); }, iters);
            const float t5 = run([&] { hipLaunchKernelGGL((k_row_prop<N, false>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters);
            printf("B %4d grid %4d: production %7.2f us per 25   row prefetch %7.2f   row + h prefetch %7.2f   row prefetch, grid %d: %7.2f   XCD-aware tile order %7.2f   production again %7.2f\n",
                   B, grid, t0 * 1e3 * 25 / B, t1 * 1e3 * 25 / B, t2 * 1e3 * 25 / B, ncu * 2, t3 * 1e3 * 25 / B, t4 * 1e3 * 25 / B, t5 * 1e3 * 25 / B);
        }
        // same results?
        hipLaunchKernelGGL((k_row_prop<N, false>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a);
        hipLaunchKernelGGL((kx_prop<N>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a2);
        CK(hipDeviceSynchronize());
        std::vector<float> r0((size_t)B * N * N * 2), r1(r0.size());
        CK(hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r1.data(), out2, r1.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; double mx = 0;
        for (size_t i = 0; i < r0.size(); ++i) { if (r0[i] != r1[i]) ++diff; mx = std::max(mx, (double)fabsf(r0[i])); }
        printf("B %4d: %zu of %zu values differ (max |out| %.3g)\n", B, diff, r0.size(), mx);
    }
    return 0;
}
