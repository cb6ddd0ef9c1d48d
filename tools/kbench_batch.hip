// Development aid (not part of the product): what would a persistent kernel gain on the transfer-function step?
// The same k_row_prop instance over B = 25, 50, ... 800 wavefields of 512^2 in ONE launch (no dependencies between tiles), as
// microseconds per 25 wavefields: the difference to the B = 25 launch is the per-launch ramp / tail / first-tile latency that a
// dataflow kernel could at best remove.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/kbench_batch tools/kbench_batch.hip && ./tools/kbench_batch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../beyond_dof_amd/csrc/bdof_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static int balanced(int tiles, int cap) { if (tiles <= cap) return tiles; int r = (tiles + cap - 1) / cap; return (tiles + r - 1) / r; }

template <bool EX> static float run(const RowPropArgs& a, int grid, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_row_prop<512, EX>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_row_prop<512, EX>), dim3(grid), dim3(BDOF_THREADS), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / iters;
}

int main() {
    constexpr int N = 512;
    const int Bmax = 800;
    const size_t fld = (size_t)Bmax * N * N;
    cf *in, *out, *h, *tw;
    CK(hipMalloc(&in, fld * 8)); CK(hipMalloc(&out, fld * 8)); CK(hipMalloc(&h, (size_t)N * N * 8)); CK(hipMalloc(&tw, 2 * N * 8));
    std::vector<cf> hw(2 * N, make_float2(0.f, 0.f));
    for (int j = 0; j < N; ++j) hw[j] = make_float2((float)cos(-2 * M_PI * j / N), (float)sin(-2 * M_PI * j / N));
    CK(hipMemcpy(tw, hw.data(), 2 * N * 8, hipMemcpyHostToDevice));
    CK(hipMemset(in, 0, fld * 8)); CK(hipMemset(h, 0, (size_t)N * N * 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("k_row_prop<512>, 2 workgroups per CU, us per 25 wavefields (one launch over B wavefields; 105 MB algorithmic per 25)\n");
    for (int B : {25, 50, 100, 200, 400, 800}) {
        RowPropArgs a{in, out, h, B, N, 1.f, 0, tw};
        const int tiles = B * N / RowCfg<N>::TILE, grid = balanced(tiles, ncu * 2);
        const int iters = B <= 100 ? 40 : 10;
        const float fast = run<false>(a, grid, iters), exact = run<true>(a, grid, iters);
        printf("B %4d grid %4d: plain tables %7.2f us per 25 (%.0f GB/s)   exact tables %7.2f us per 25 (%.0f GB/s)\n", B, grid,
               fast * 1e3 * 25 / B, 16.0 * 25 * N * N / (fast * 25 / B) / 1e6, exact * 1e3 * 25 / B, 16.0 * 25 * N * N / (exact * 25 / B) / 1e6);
    }
    return 0;
}
