// Micro-benchmark of the hot kernels in isolation (development aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/kbench tools/kbench.hip && ./tools/kbench [B] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../beyond_dof_amd/csrc/bdof_kernels.h"
#include "kvariants.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F> static float time_it(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

// calibration kernels for the PMC byte counters: known traffic with this library's access widths
__global__ void k_calib_copy8(const float2* in, float2* out, size_t n) {       // 8 B per lane, like the row loads/stores
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_calib_copy16(const float4* in, float4* out, size_t n) {      // 16 B per lane
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// read-only and write-only ceilings (the rotation adjoint is 96 % reads, the tape is written once)
__global__ void k_calib_read16(const float4* in, float* out, size_t n) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
// the same with four independent 16-byte loads per lane in flight
__global__ void k_calib_read16x4(const float4* in, float* out, size_t n) {
    float4 acc[4];
    for (auto& a : acc) a = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k].x += v[k].x; acc[k].y += v[k].y; acc[k].z += v[k].z; acc[k].w += v[k].w; }
    }
    for (; i < n; i += stride) { const float4 v = in[i]; acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w; }
    float t = 0.f;
    for (auto& a : acc) t += a.x + a.y + a.z + a.w;
    if (t == 123.456f) out[0] = t;
}
__global__ void k_calib_write16(float4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <class T> __global__ void k_calib_write(T* out, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
template <class T> __global__ void k_calib_write_nt(T* out, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, out + i);
}
// each workgroup writes its own contiguous chunk (rows of 4 KB one after the other), not a grid-stride interleave
template <class T> __global__ void k_calib_write_chunk(T* out, size_t n, T v) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = per * blockIdx.x, i1 = i0 + per < n ? i0 + per : n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) out[i] = v;
}

static int balanced(int tiles, int cap) { if (tiles <= cap) return tiles; int r = (tiles + cap - 1) / cap; return (tiles + r - 1) / r; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 25;
    constexpr int N = 512;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int S = 8;
    const size_t fld = (size_t)B * N * N;
    cf *in, *out, *tape, *h, *tw, *probe; float2 *vol, *grot; int *tab, *ang;
    CK(hipMalloc(&in, fld * 8)); CK(hipMalloc(&out, fld * 8)); CK(hipMalloc(&tape, fld * 8));
    CK(hipMalloc(&h, (size_t)N * N * 8)); CK(hipMalloc(&tw, N * 8)); CK(hipMalloc(&probe, (size_t)N * N * 8));
    CK(hipMalloc(&vol, (size_t)N * N * N * 8)); CK(hipMalloc(&grot, fld * S * 8));
    CK(hipMalloc(&tab, (size_t)B * S * N * 4)); CK(hipMalloc(&ang, B * 4));
    std::vector<cf> hw(N);
    for (int j = 0; j < N; ++j) hw[j] = make_float2((float)cos(-2 * M_PI * j / N), (float)sin(-2 * M_PI * j / N));
    CK(hipMemcpy(tw, hw.data(), N * 8, hipMemcpyHostToDevice));
    std::vector<float> rnd(fld * 2);
    for (auto& v : rnd) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(in, rnd.data(), fld * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tape, rnd.data(), fld * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(h, rnd.data(), (size_t)N * N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(probe, rnd.data(), (size_t)N * N * 8, hipMemcpyHostToDevice));
    {
        std::vector<float> v((size_t)N * N * N * 2);
        for (auto& x : v) x = 1e-9f * (float)(rand() % 1000);
        CK(hipMemcpy(vol, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        std::vector<int> t((size_t)B * S * N), a(B);
        for (auto& x : t) x = rand() % (N * N);      // random source rows, like a generic rotation angle
        for (int b = 0; b < B; ++b) a[b] = b;
        CK(hipMemcpy(tab, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(ang, a.data(), B * 4, hipMemcpyHostToDevice));
    }
    ObjView obj{vol, tab, ang, nullptr, nullptr, S, N, N};
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double px = (double)B * N * N;
    {   // 2 x 1 GiB buffers: far beyond the 256 MiB Infinity Cache
        const size_t nb = (size_t)1 << 30;
        float2 *ca, *cb;
        CK(hipMalloc(&ca, nb)); CK(hipMalloc(&cb, nb));
        CK(hipMemset(ca, 1, nb));
        float ms = time_it([&] { hipLaunchKernelGGL(k_calib_copy8, dim3(ncu * 8), dim3(256), 0, 0, ca, cb, nb / 8); }, 3);
        printf("calib copy8  1 GiB read + 1 GiB write: %8.2f us  %7.1f GB/s\n", ms * 1e3, 2.0 * nb / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_calib_copy16, dim3(ncu * 8), dim3(256), 0, 0, (const float4*)ca, (float4*)cb, nb / 16); }, 3);
        printf("calib copy16 1 GiB read + 1 GiB write: %8.2f us  %7.1f GB/s\n", ms * 1e3, 2.0 * nb / ms / 1e6);
        for (int g : {8, 16, 32}) {
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_read16, dim3(ncu * g), dim3(256), 0, 0, (const float4*)ca, (float*)cb, nb / 16); }, 3);
            printf("calib read16   1 GiB read  (grid %2d x CUs): %8.2f us  %7.1f GB/s\n", g, ms * 1e3, 1.0 * nb / ms / 1e6);
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_read16x4, dim3(ncu * g), dim3(256), 0, 0, (const float4*)ca, (float*)cb, nb / 16); }, 3);
            printf("calib read16x4 1 GiB read  (grid %2d x CUs): %8.2f us  %7.1f GB/s\n", g, ms * 1e3, 1.0 * nb / ms / 1e6);
        }
        ms = time_it([&] { hipLaunchKernelGGL(k_calib_write16, dim3(ncu * 8), dim3(256), 0, 0, (float4*)cb, nb / 16); }, 3);
        printf("calib write16  1 GiB write: %8.2f us  %7.1f GB/s\n", ms * 1e3, 1.0 * nb / ms / 1e6);
        for (int g : {4, 8, 16}) {
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_write<float>, dim3(ncu * g), dim3(256), 0, 0, (float*)cb, nb / 4, 1.f); }, 3);
            printf("calib write  4 B/lane (grid %2d x CUs): %7.1f GB/s", g, 1.0 * nb / ms / 1e6);
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_write<float2>, dim3(ncu * g), dim3(256), 0, 0, (float2*)cb, nb / 8, make_float2(1.f, 2.f)); }, 3);
            printf("   8 B/lane: %7.1f GB/s", 1.0 * nb / ms / 1e6);
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_write<float4>, dim3(ncu * g), dim3(256), 0, 0, (float4*)cb, nb / 16, make_float4(1.f, 2.f, 3.f, 4.f)); }, 3);
            printf("   16 B/lane: %7.1f GB/s", 1.0 * nb / ms / 1e6);
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_write_nt<float>, dim3(ncu * g), dim3(256), 0, 0, (float*)cb, nb / 4, 1.f); }, 3);
            printf("   non-temporal 4 B: %7.1f", 1.0 * nb / ms / 1e6);
            ms = time_it([&] { hipLaunchKernelGGL(k_calib_write_chunk<float2>, dim3(ncu * g), dim3(256), 0, 0, (float2*)cb, nb / 8, make_float2(1.f, 2.f)); }, 3);
            printf("   8 B/lane, a contiguous chunk per workgroup: %7.1f GB/s\n", 1.0 * nb / ms / 1e6);
        }
        CK(hipFree(ca)); CK(hipFree(cb));
    }
    const int tiles = B * N / RowCfg<N>::TILE;
    if (argc > 3 && atoi(argv[3]) == 1) {      // grid sweep: how many workgroups should share the launch's tiles (alternated, best of 5)
        RowFwdArgs fa{in, probe, out, tape, obj, B, N, 3, 25.3f, make_float2(1.f, 0.f), tw};
        RowPropArgs pa{in, out, h, B, N, 1.f, 0, tw};
        RowBwdArgs ba{in, tape, out, grot, obj, B, N, 3, 25.3f, make_float2(1.f, 0.f), tw};
        fa.sq[0] = fa.sq[1] = pa.sq[0] = pa.sq[1] = ba.sq[0] = ba.sq[1] = 0.70710678f;
        const int grids[] = {400, 448, 512, 640, 800};
        float best[3][5];
        for (auto& r : best) for (float& v : r) v = 1e9f;
        for (int rep = 0; rep < 5; ++rep)
            for (int gi = 0; gi < 5; ++gi) {
                const int g = grids[gi];
                best[0][gi] = std::min(best[0][gi], time_it([&] { hipLaunchKernelGGL((k_row_fwd<N, false, true>), dim3(g), dim3(BDOF_THREADS), 0, 0, fa); }, iters));
                best[1][gi] = std::min(best[1][gi], time_it([&] { hipLaunchKernelGGL((k_row_prop<N>), dim3(g), dim3(BDOF_THREADS), 0, 0, pa); }, iters));
                best[2][gi] = std::min(best[2][gi], time_it([&] { hipLaunchKernelGGL((k_row_bwd<N, 1>), dim3(g), dim3(BDOF_THREADS), 0, 0, ba); }, iters));
            }
        const char* nm[3] = {"row_fwd ", "row_prop", "row_bwd "};
        for (int k = 0; k < 3; ++k) {
            printf("%s (%d tiles)", nm[k], tiles);
            for (int gi = 0; gi < 5; ++gi) printf("  grid %d: %6.2f us", grids[gi], best[k][gi] * 1e3);
            printf("\n");
        }
        return 0;
    }
    if (argc > 3 && atoi(argv[3]) == 2) {      // tile-order experiment (kvariants.h: kv_prop_order), alternated, best of 7
        RowPropArgs pa{in, out, h, B, N, 1.f, 0, tw};
        pa.sq[0] = pa.sq[1] = 0.70710678f;
        const int g = tiles / 2;
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 7; ++rep) {
            best[0] = std::min(best[0], time_it([&] { hipLaunchKernelGGL((kv_prop_order<N, 0>), dim3(g), dim3(BDOF_THREADS), 0, 0, pa); }, iters));
            best[1] = std::min(best[1], time_it([&] { hipLaunchKernelGGL((kv_prop_order<N, 1>), dim3(g), dim3(BDOF_THREADS), 0, 0, pa); }, iters));
            best[2] = std::min(best[2], time_it([&] { hipLaunchKernelGGL((kv_prop_order<N, 2>), dim3(g), dim3(BDOF_THREADS), 0, 0, pa); }, iters));
        }
        printf("tile order (grid %d, %d tiles): round-robin %6.2f us   adjacent pairs %6.2f us   pairs of pairs per XCD %6.2f us\n", g, tiles, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3);
        {   // which side should carry the transposition? (memory-only skeletons, alternated, best of 7; same-buffer = Infinity Cache
            // resident like the sweep's wavefields, then the ring of fresh buffers)
            float bs[2] = {1e9f, 1e9f};
            for (int rep = 0; rep < 7; ++rep) {
                bs[0] = std::min(bs[0], time_it([&] { hipLaunchKernelGGL((kv_prop_side<N, 0>), dim3(g), dim3(BDOF_THREADS), 0, 0, pa); }, iters));
                bs[1] = std::min(bs[1], time_it([&] { hipLaunchKernelGGL((kv_prop_side<N, 1>), dim3(g), dim3(BDOF_THREADS), 0, 0, pa); }, iters));
            }
            printf("memory-only skeleton: rows in, 128-byte segments out (production) %6.2f us   128-byte segments in, rows out %6.2f us\n", bs[0] * 1e3, bs[1] * 1e3);
        }
        // the same with every launch on fresh buffers (a ring of 8 x in/out: 840 MB, beyond the Infinity Cache), as the sweep sees them
        const int ring = 8;
        cf *rin, *rout;
        CK(hipMalloc(&rin, fld * 8 * ring)); CK(hipMalloc(&rout, fld * 8 * ring));
        for (int r = 0; r < ring; ++r) CK(hipMemcpy(rin + fld * r, in, fld * 8, hipMemcpyDeviceToDevice));
        float bst[3] = {1e9f, 1e9f, 1e9f};
        int turn = 0;
        auto launch = [&](int order) {
            RowPropArgs q = pa; q.in = rin + fld * (turn % ring); q.out = rout + fld * (turn % ring); ++turn;
            if (order == 0) hipLaunchKernelGGL((kv_prop_order<N, 0>), dim3(g), dim3(BDOF_THREADS), 0, 0, q);
            else if (order == 1) hipLaunchKernelGGL((kv_prop_order<N, 1>), dim3(g), dim3(BDOF_THREADS), 0, 0, q);
            else hipLaunchKernelGGL((kv_prop_order<N, 2>), dim3(g), dim3(BDOF_THREADS), 0, 0, q);
        };
        for (int rep = 0; rep < 7; ++rep)
            for (int o = 0; o < 3; ++o) bst[o] = std::min(bst[o], time_it([&] { launch(o); }, iters));
        {
            float bs[2] = {1e9f, 1e9f};
            auto launch2 = [&](int side) {
                RowPropArgs q = pa; q.in = rin + fld * (turn % ring); q.out = rout + fld * (turn % ring); ++turn;
                if (side == 0) hipLaunchKernelGGL((kv_prop_side<N, 0>), dim3(g), dim3(BDOF_THREADS), 0, 0, q);
                else hipLaunchKernelGGL((kv_prop_side<N, 1>), dim3(g), dim3(BDOF_THREADS), 0, 0, q);
            };
            for (int rep = 0; rep < 7; ++rep)
                for (int o = 0; o < 2; ++o) bs[o] = std::min(bs[o], time_it([&] { launch2(o); }, iters));
            printf("  skeletons on the ring: production direction %6.2f us   transposed reads %6.2f us\n", bs[0] * 1e3, bs[1] * 1e3);
        }
        printf("  on a ring of %d buffer pairs:    round-robin %6.2f us   adjacent pairs %6.2f us   pairs of pairs per XCD %6.2f us\n", ring, bst[0] * 1e3, bst[1] * 1e3, bst[2] * 1e3);
        return 0;
    }
    for (int per_cu : {2, 3, 4}) {
        const int grid = balanced(tiles, ncu * per_cu);
        printf("-- %d WG/CU cap, grid %d (tiles %d)\n", per_cu, grid, tiles);
        {
            RowFwdArgs a{in, probe, out, tape, obj, B, N, 3, 25.3f, make_float2(1.f, 0.f), tw};
            float ms = time_it([&] { hipLaunchKernelGGL((k_row_fwd<N, false, true>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters);
            printf("row_fwd   %8.2f us  %7.1f GB/s (24 B/px)\n", ms * 1e3, 24 * px / ms / 1e6);
        }
        {
            RowPropArgs a{in, out, h, B, N, 1.f, 0, tw};
            float ms = time_it([&] { hipLaunchKernelGGL((k_row_prop<N>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters);
            printf("row_prop  %8.2f us  %7.1f GB/s (16 B/px)\n", ms * 1e3, 16 * px / ms / 1e6);
        }
        {
            RowPropArgs a{in, out, h, B, N, 1.f, 0, tw};
            float ms;
            ms = time_it([&] { hipLaunchKernelGGL((kv_prop<N, 0>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters); printf("  prop full        %8.2f us\n", ms * 1e3);
            ms = time_it([&] { hipLaunchKernelGGL((kv_prop<N, 1>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters); printf("  prop memory-only %8.2f us\n", ms * 1e3);
            ms = time_it([&] { hipLaunchKernelGGL((kv_prop<N, 2>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters); printf("  prop compute-only%8.2f us\n", ms * 1e3);
            ms = time_it([&] { hipLaunchKernelGGL((kv_prop<N, 3>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters); printf("  prop plain-store %8.2f us\n", ms * 1e3);
            ms = time_it([&] { hipLaunchKernelGGL((kv_prop<N, 4>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters); printf("  prop in-register exchanges %8.2f us\n", ms * 1e3);
            if (per_cu == 2) {          // the in-register variant against the production form, same inputs
                std::vector<cf> r0(fld), r5(fld);
                hipLaunchKernelGGL((kv_prop<N, 0>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a);
                CK(hipMemcpy(r0.data(), out, fld * 8, hipMemcpyDeviceToHost));
                CK(hipMemset(out, 0, fld * 8));
                hipLaunchKernelGGL((kv_prop<N, 5>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a);
                CK(hipMemcpy(r5.data(), out, fld * 8, hipMemcpyDeviceToHost));
                double num = 0, den = 0;
                for (size_t i = 0; i < fld; ++i) {
                    num += (double)(r0[i].x - r5[i].x) * (r0[i].x - r5[i].x) + (double)(r0[i].y - r5[i].y) * (r0[i].y - r5[i].y);
                    den += (double)r0[i].x * r0[i].x + (double)r0[i].y * r0[i].y;
                }
                printf("  in-register vs production result: rel L2 %.3e\n", sqrt(num / den));
            }
        }
        {
            RowBwdArgs a{in, tape, out, grot, obj, B, N, 3, 25.3f, make_float2(1.f, 0.f), tw};
            float ms = time_it([&] { hipLaunchKernelGGL((k_row_bwd<N, 1>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a); }, iters);
            printf("row_bwd   %8.2f us  %7.1f GB/s (40 B/px)\n", ms * 1e3, 40 * px / ms / 1e6);
        }
    }
    return 0;
}
