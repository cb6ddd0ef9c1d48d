// Real-space propagator, second tiling (round 3): the same two 1-D passes and epilogue as k_conv (bdof_kernels.h), same
// order of the tap sums — results equal to float32 rounding of the epilogue — with the data movement rebuilt around LDS-DMA:
//   * 64 x 32 tile (x by y) instead of 32 x 64: the y pass, which also has to produce the x pass's halo rows, computes
//     (64 + 2H) x 32 outputs for 64 x 32 (25 % more at 17 taps) instead of (32 + 2H) x 64 for 32 x 64 (50 % more);
//   * the halo tile goes global -> LDS by global_load_lds_dwordx4 (no staging registers, no stash pass, no per-element
//     index arithmetic: the per-lane source offsets are tile-invariant and computed once); tiles that touch the field's
//     edge clamp their addresses and overwrite what lies outside with the padding constant after the data have landed;
//   * raw image row-major in 16-byte units (two consecutive y), RU = (TY + 2H) / 2 + 1 units per row: an odd unit count
//     makes the y pass's ds_read_b128 (lanes = 8 windows x 8 rows, 32 bytes apart along a row) conflict-free, and the
//     DMA image stays lane-linear (the one dummy unit per row is a masked lane);
//   * windows of 4 outputs in both passes: 8 (64 + 2H) y-pass windows in two rounds, 512 x-pass windows — every wave
//     issues the same FMA count in the x pass and the y pass's second round is 2H / 8 waves;
//   * 52 KB of LDS and <= 80 VGPRs: three workgroups per CU (k_conv: two).
// Reference: cnn_propagator/propagation.py:80-107 (the convolution of one slice), :109-110 (renormalisation, k_conv_final).
#pragma once

template <int H> struct Conv2Cfg {
    static constexpr int TX = 64, TY = 32, R = 4, SM = 33, THREADS = 512;
    static constexpr int TXH = TX + 2 * H, TYH = TY + 2 * H, NP = TYH / 2, RU = NP | 1;
    static constexpr int UNITS = TXH * RU, NLOADS = (UNITS + 63) / 64, MP = (NLOADS + 7) / 8;
    static constexpr int A_BYTES = NLOADS * 1024, M_BYTES = TXH * SM * 8, LDS = A_BYTES + M_BYTES;
    static constexpr int MINW = 4;
    static_assert(H % 2 == 0 && NP % 2 == 0, "halo of even width: 16-byte units must not straddle the field's edge");
    static_assert(TX * TY == THREADS * R, "one x-pass window per thread");
};

template <bool BWD, int H, bool PF = false>
__global__ __launch_bounds__(Conv2Cfg<H>::THREADS, Conv2Cfg<H>::MINW) void k_conv2(ConvArgs a) {
    typedef Conv2Cfg<H> C;
    constexpr int TX = C::TX, TY = C::TY, R = C::R, TXH = C::TXH, RU = C::RU, NP = C::NP, SM = C::SM, MP = C::MP;
    typedef const __attribute__((address_space(4))) ConvTaps* TapsPtr;
    typedef const __attribute__((address_space(1))) void* GPtr;
    typedef __attribute__((address_space(3))) void* LPtr;
    // two LDS objects of their own: hipcc then knows that a read of M cannot alias the DMA's destination, and does not
    // wait for the DMA before the x pass (with one dynamic array it puts s_waitcnt vmcnt(0) in front of the first read)
    __shared__ float4 A4[C::NLOADS * 64];                    // raw halo tile [TXH][RU] units of two complex values
    __shared__ cf M[TXH * SM];                               // y-pass result [TXH][SM]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = a.NX / TX, tiles_y = a.NY / TY, tpf = tiles_x * tiles_y;
    const int ntiles = a.B * tpf;

    // ---- LDS-DMA plan of this lane: unit u = 64 k + lane of load k = wave + 8 m holds row i = u / RU, pair c = u % RU
    unsigned rel[MP];            // byte offset of the unit from the tile's halo origin (interior tiles)
    unsigned act = 0;            // bit m: the lane takes part in load m
#pragma unroll
    for (int m = 0; m < MP; ++m) {
        const int u = (wave + 8 * m) * 64 + lane;
        const int i = u / RU, c = u - i * RU;
        rel[m] = (unsigned)(min(i, TXH - 1) * a.NY + 2 * min(c, NP - 1)) * 8u;
        act |= (wave + 8 * m < C::NLOADS && i < TXH && c < NP) ? 1u << m : 0u;
    }
    unsigned oob = 0;            // bit m: the unit of load m lies outside the field (padding constant after landing)
    auto issue = [&](int tile) {
        const int b = tile / tpf, t2 = tile - b * tpf;
        const int x0 = (t2 / tiles_y) * TX, y0 = (t2 % tiles_y) * TY;
        const char* src = (const char*)(a.in + (size_t)b * a.NX * a.NY);
        const bool interior = x0 >= H && x0 + TX + H <= a.NX && y0 >= H && y0 + TY + H <= a.NY;      // uniform
        oob = 0;
        if (interior) {
            const char* base = src + ((size_t)(x0 - H) * a.NY + (y0 - H)) * 8;
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if ((act >> m) & 1u)
                    __builtin_amdgcn_global_load_lds((GPtr)(base + rel[m]), (LPtr)(A4 + (wave + 8 * m) * 64), 16, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const int u = (wave + 8 * m) * 64 + lane;
                const int i = u / RU, c = u - i * RU;
                const int x = x0 - H + i, y = y0 - H + 2 * c;
                const bool in = (unsigned)x < (unsigned)a.NX && (unsigned)y < (unsigned)a.NY;
                const unsigned off = (__umul24(min(max(x, 0), a.NX - 1), a.NY) + min(max(y, 0), a.NY - 2)) * 8u;
                if ((act >> m) & 1u) {
                    oob |= in ? 0u : 1u << m;
                    __builtin_amdgcn_global_load_lds((GPtr)(src + off), (LPtr)(A4 + (wave + 8 * m) * 64), 16, 0, 0);
                }
            }
        }
    };
    // modulation-table rows of this thread's x-pass outputs (obj_src_row with the per-wavefield part as a scalar base): the
    // table entries are requested at the END of the previous tile (behind its stores and the DMA, so that nothing waits for
    // them while the DMA is in flight) and turned into rows at the top of the tile, after the wait that retires the DMA
    const int i0 = (tid / TY) * R, j = tid % TY;
    const bool use_tab = a.obj.tab != nullptr && a.zmod >= 0;                 // uniform
    int sraw[R];
    unsigned xin = 0;
    int yo = 0;                  // window origin in y of the tile's wavefield (ptychography)
    auto request_rows = [&](int tile) {
        const int b = tile / tpf, t2 = tile - b * tpf;
        const int x0 = (t2 / tiles_y) * TX;
        xin = 0;
        yo = a.obj.yoff ? a.obj.yoff[b] : 0;
        if (use_tab) {
            const int xg0 = x0 + i0 + (a.obj.xoff ? a.obj.xoff[b] : 0);
            const int* tabrow = a.obj.tab + ((long long)a.obj.angle_of_b[b] * a.obj.S + a.zmod) * a.obj.volNX;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int xg = xg0 + q;
                xin |= (unsigned)xg < (unsigned)a.obj.volNX ? 1u << q : 0u;
                sraw[q] = tabrow[min(max(xg, 0), a.obj.volNX - 1)];
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) sraw[q] = a.zmod < 0 || a.obj.tab ? 0 : (b * a.obj.S + a.zmod) * a.NX + x0 + i0 + q;
            xin = a.zmod < 0 ? 0u : (1u << R) - 1u;               // zmod < 0: no modulation (row 0 is read and not used)
        }
    };

    if ((int)blockIdx.x < ntiles) {
        issue(blockIdx.x);
        request_rows(blockIdx.x);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tpf, t2 = tile - b * tpf;
        const int x0 = (t2 / tiles_y) * TX, y0 = (t2 % tiles_y) * TY;
        // the halo tile has landed (and the table rows with it); outside the field the padding constant
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (oob) {
            const float4 pp = make_float4(a.pad.x, a.pad.y, a.pad.x, a.pad.y);
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if ((oob >> m) & 1u) A4[(wave + 8 * m) * 64 + lane] = pp;
        }
        conv_sync();
        // loads the epilogue needs, in flight during the y pass
        const int y = y0 + j;
        const int yg = y + yo;
        const int yc = min(max(yg, 0), a.obj.volNY - 1);
        float2 m1[R];
        cf tp[R], pf[R];
        const cf* tape_b = BWD ? a.tape + (size_t)b * a.NX * a.NY : nullptr;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (sraw[q] < 0) xin &= ~(1u << q);                          // a table entry that points nowhere
            m1[q] = a.obj.vol[(size_t)max(sraw[q], 0) * a.obj.volNY + yc];
            if constexpr (BWD) tp[q] = tape_b[__umul24(x0 + i0 + q, a.NY) + y];
            if constexpr (PF) pf[q] = a.pfield[__umul24(x0 + i0 + q, a.NY) + y];       // L2-resident plane shared by all wavefields
        }
        TapsPtr kt = (TapsPtr)a.taps;
        asm volatile("" : "+s"(kt));
        // pass along y: lanes = 8 windows of 4 outputs along a row, then rows.  forward: o[y] = sum_d K[h+d] f[y-d]
#pragma unroll 1
        for (int t = tid; t < TXH * (TY / R); t += C::THREADS) {
            const int w = t % (TY / R), i = t / (TY / R);
            const float4* p = A4 + i * RU + 2 * w;
            cf win[R + 2 * H], o[R];
#pragma unroll
            for (int q = 0; q < (R + 2 * H) / 2; ++q) {
                const float4 v = p[q];
                win[2 * q] = make_float2(v.x, v.y);
                win[2 * q + 1] = make_float2(v.z, v.w);
            }
            conv_window<BWD, H, R>(win, kt->ky, o);
#pragma unroll
            for (int q = 0; q < R; ++q) M[i * SM + R * w + q] = o[q];
        }
        // the epilogue's operands are in registers before the next DMA is queued behind them (vmcnt retires in order)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            asm volatile("" : "+v"(m1[q].x), "+v"(m1[q].y));
            if constexpr (BWD) asm volatile("" : "+v"(tp[q].x), "+v"(tp[q].y));
            if constexpr (PF) asm volatile("" : "+v"(pf[q].x), "+v"(pf[q].y));
        }
        conv_sync();
        const unsigned xin_t = xin;
        const int next = tile + gridDim.x;
        if (next < ntiles) issue(next);                 // A is free: in flight during the x pass and the epilogue
        asm volatile("" : "+s"(kt));
        // pass along x (window of R consecutive x for one y), then the pointwise physics
        {
            cf o[R];
            const cf ke = make_float2(kt->e.x, kt->e.y);
            cf* out_b = a.out + (size_t)b * a.NX * a.NY;
            float2* grot_b = BWD ? a.grot + ((size_t)b * a.obj.S + a.zmod) * a.NX * a.NY : nullptr;
            cf win[R + 2 * H];
#pragma unroll
            for (int q = 0; q < R + 2 * H; ++q) win[q] = M[(i0 + q) * SM + j];
            conv_window<BWD, H, R>(win, kt->kx, o);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const cf acc = BWD ? cmulc(o[q], ke) : cmul(o[q], ke);
                const int x = x0 + i0 + q;
                const unsigned off = __umul24(x, a.NY) + y;
                const bool in = ((xin_t >> q) & 1u) && yg == yc;
                const float2 mm = make_float2(in ? m1[q].x : 0.f, in ? m1[q].y : 0.f);
                cf car = a.carrier;
                if constexpr (PF) car = pf[q];
                if constexpr (!BWD) {
                    out_b[off] = modulate_eps(acc, car, mm);
                } else {
                    const cf phi = cadd(tp[q], car);
                    const cf tt = cmulc(acc, phi);
                    grot_b[off] = make_float2(a.k * tt.y, -a.k * tt.x);
                    out_b[off] = cmulc(acc, make_float2(1.f + mm.x, mm.y));
                }
            }
        }
        if (next < ntiles) request_rows(next);
    }
}
