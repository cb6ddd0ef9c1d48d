// Development aid (not part of the product): where a workgroup of k_conv2 spends its cycles, by in-kernel stamps (s_memtime at the
// phase boundaries of wave 0 of every workgroup), on the bench's shape — 25 fields of 512 x 512, 17 taps, object bound per wavefield.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DBDOF_CONV2_STAMP=0 -o tools/kbench_conv2 tools/kbench_conv2.hip
// (-DBDOF_CONV2_STAMP=w: the stamps of wave w; without it only the launch time; -DBDOF_CONV2_WHATIF=1|2|3: timing experiments that
//  drop the operand loads (1) and / or the forward kernel's stores (2) — wrong results, what the kernel costs without them)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <unistd.h>
#include "../beyond_dof_amd/csrc/bdof_kernels.h"
#include "../beyond_dof_amd/csrc/bdof_conv2.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool BWD> static void run(const ConvArgs& a, int grid, int iters, const char* name) {
    typedef Conv2Cfg<8> C;
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st[8];
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_conv2<BWD, 8, false>), dim3(grid), dim3(C::THREADS), 0, 0, a);
    CK(hipDeviceSynchronize());
#ifdef BDOF_CONV2_STAMP
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_conv2_stamp), zero, sizeof(zero)));
#endif
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_conv2<BWD, 8, false>), dim3(grid), dim3(C::THREADS), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
#ifdef BDOF_CONV2_STAMP
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_conv2_stamp), sizeof(st)));
#else
    (void)zero;
    printf("%s: %.2f us per launch (no stamps), grid %d\n", name, ms * 1e3 / iters, grid);
    // the same launch with the GPU idle for `gap` microseconds before it: does the back-to-back figure depend on the clock the
    // chip holds under sustained load?
    for (int gap : {0, 200, 2000}) {
        double sum = 0; float best = 1e9f;
        for (int i = 0; i < 30; ++i) {
            if (gap) usleep(gap);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_conv2<BWD, 8, false>), dim3(grid), dim3(C::THREADS), 0, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            sum += t; best = t < best ? t : best;
        }
        printf("   one launch at a time, %4d us idle before each: mean %.2f us, best %.2f us\n", gap, sum / 30 * 1e3, best * 1e3);
    }
    return;
#endif
    double tot = 0;
    for (int k = 0; k < 8; ++k) tot += (double)st[k];
    static const char* ph[8] = {"operand requests + padding patch", "barrier (halo tile complete)", "y pass", "wait for the epilogue's operands",
                                "barrier (y pass complete)", "DMA issue + table rows", "x pass + epilogue + stores issued", "wait for the next halo tile"};
    printf("%s: %.2f us per launch (with stamps), grid %d; cycles of wave 0 per workgroup and launch: %.0f\n", name, ms * 1e3 / iters, grid, tot / iters / grid);
    for (int k = 0; k < 8; ++k) printf("   %-38s %5.1f %%\n", ph[k], 100.0 * st[k] / tot);
}

int main() {
    const int B = 25, N = 512, S = 2;
    const size_t fld = (size_t)B * N * N;
    cf *in, *out, *tape; float2 *vol, *grot; ConvTaps* taps;
    CK(hipMalloc(&in, fld * 8)); CK(hipMalloc(&out, fld * 8)); CK(hipMalloc(&tape, fld * 8));
    CK(hipMalloc(&vol, fld * S * 8)); CK(hipMalloc(&grot, fld * S * 8)); CK(hipMalloc(&taps, sizeof(ConvTaps)));
    std::vector<float> rnd(fld * 2);
    for (auto& v : rnd) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(in, rnd.data(), fld * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tape, rnd.data(), fld * 8, hipMemcpyHostToDevice));
    CK(hipMemset(vol, 0, fld * S * 8));
    ConvTaps t{};
    for (int i = 0; i < 17; ++i) { t.ky[i] = make_float2(0.05f * (i + 1), -0.01f * i); t.kx[i] = make_float2(0.03f * (17 - i), 0.02f * i); }
    t.e = make_float2(1.f, 0.f); t.ks = 17;
    CK(hipMemcpy(taps, &t, sizeof(t), hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    ObjView obj{vol, nullptr, nullptr, nullptr, nullptr, S, N, N};
#ifndef BDOF_CONV2_PER_CU
#define BDOF_CONV2_PER_CU 2
#endif
    const int nstrips = B * (N / 64), run_tiles = ((nstrips + 7) / 8) * (N / 32), slots = prop.multiProcessorCount * BDOF_CONV2_PER_CU / 8;
    const int rounds = (run_tiles + slots - 1) / slots, grid = 8 * ((run_tiles + rounds - 1) / rounds);
    ConvArgs af{in, out, nullptr, nullptr, obj, B, N, N, 1, make_float2(0.f, 0.f), make_float2(1.f, 0.f), 1e-3f, taps, 17, nullptr};
    ConvArgs ab{in, out, tape, grot, obj, B, N, N, 1, make_float2(0.f, 0.f), make_float2(1.f, 0.f), 1e-3f, taps, 17, nullptr};
    for (int rep = 0; rep < 2; ++rep) {
        run<false>(af, grid, 40, "forward");
        run<true>(ab, grid, 40, "backward");
    }
    return 0;
}
