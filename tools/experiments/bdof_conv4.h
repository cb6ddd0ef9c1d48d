// EXPERIMENT, not built into the library (measured in round 3: correct — the parity tests of tests/test_gpu_conv.py passed with it,
// bit-identical on one and two streams — and SLOWER than k_conv2: forward 49.6 against 45.2 us, backward 61.8 against 54.4 us per
// launch).  To try it again: include it after bdof_conv2.h in csrc/bdof_capi.hip and launch k_conv4<BWD, PF> with
// k_conv2's grid for 17 taps.
// Real-space propagator, double-buffered form of k_conv2 (bdof_conv2.h; round 3).  What k_conv2's what-if timings showed
// (profiles/r03_conv2_phase_stamps.txt): without its tap sums the forward kernel runs at the memory floor, 25.8 us; with them
// 47 us = that floor + the 21 us the FMAs take — a workgroup has memory in flight only between its DMA issue (after the y pass:
// the halo buffer is in use until then) and its stores, and none during the y pass.  Here the halo tile has TWO buffers and the DMA
// of tile T + 1 is queued at the top of tile T, in flight during all of its arithmetic.  To keep two workgroups per CU (2 x 32 KB
// of halo buffers + the 21 KB of the y-pass result would not fit twice into 160 KB) the y-pass result overwrites the tile's own
// buffer in place: all windows are read into registers, a barrier, then the results are stored over the first 32 columns of their
// rows.  The epilogue's operands are requested after the y pass, behind the DMA in the queue: waiting for them waits for the DMA
// too, which by then has had the whole tile to land.  17 taps; the other tap counts run k_conv2.
#pragma once

template <bool BWD, bool PF = false>
__global__ __launch_bounds__(Conv2Cfg<8>::THREADS, 4) void k_conv4(ConvArgs a) {
    constexpr int H = 8;
    typedef Conv2Cfg<H, 64> C;
    static_assert(C::NLOADS == 32 && C::MP == 4, "every wave issues exactly four DMA pieces per halo tile: the counted wait below relies on it");
    constexpr int NW = C::NW;
    constexpr int TX = C::TX, TY = C::TY, R = C::R, TXH = C::TXH, RU = C::RU, NP = C::NP, SM = C::SM, MP = C::MP;
    typedef const __attribute__((address_space(4))) ConvTaps* TapsPtr;
    __shared__ float4 A4[2 * C::NLOADS * 64];                // two halo-tile buffers [TXH][RU] units; the y-pass result of a tile
    constexpr int ABUF = C::NLOADS * 64, MS = 2 * RU;        // overwrites the first 32 columns of its rows in place (row stride MS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned a4_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)A4;      // LDS byte address of the raw image
    // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin (b and b + 8 share one): the strips (one row of
    // tiles across a field; B * tiles_x of them) are split into 8 contiguous runs, one per value of blockIdx.x % 8, and the
    // workgroups of a run walk its tiles in order — the tiles in flight on an XCD at any time are neighbours in y and x, so
    // the halos they share (7/8 of what a tile reads beyond its own pixels) are hits in that XCD's L2 instead of second
    // fetches through the fabric.  Placement is the dispatcher's: a different one costs speed, never correctness.
    const int tiles_x = a.NX / TX, tiles_y = a.NY / TY;
    const int nstrips = a.B * tiles_x;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;       // the host launches a multiple of 8
    const int strip0 = xcd * nstrips / 8;
    const int ntiles = ((xcd + 1) * nstrips / 8 - strip0) * tiles_y;                   // tiles of this XCD's run
    struct TilePos { int b, x0, y0; };
    auto tile_pos = [&](int l) -> TilePos {
        const int s = strip0 + l / tiles_y;
        const int b = s / tiles_x;
        return TilePos{b, (s - b * tiles_x) * TX, (l % tiles_y) * TY};
    };

    // ---- LDS-DMA plan of this lane: unit u = 64 k + lane of load k = wave + NW m holds row i = u / RU, pair c = u % RU
    unsigned rel[MP];            // byte offset of the unit from the tile's halo origin (interior tiles)
    unsigned act = 0;            // bit m: the lane takes part in load m
#pragma unroll
    for (int m = 0; m < MP; ++m) {
        const int u = (wave + NW * m) * 64 + lane;
        const int i = u / RU, c = u - i * RU;
        rel[m] = (unsigned)(min(i, TXH - 1) * a.NY + 2 * min(c, NP - 1)) * 8u;
        act |= (wave + NW * m < C::NLOADS && i < TXH && c < NP) ? 1u << m : 0u;
    }
    unsigned oob = 0;            // bit m: the unit of load m lies outside the field (padding constant after landing)
    auto issue = [&](int tile, int buf) {
        const unsigned dst0 = a4_lds + (unsigned)buf * (ABUF * 16);
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0, y0 = tp_.y0;
        const char* src = (const char*)(a.in + (size_t)b * a.NX * a.NY);
        const bool interior = x0 >= H && x0 + TX + H <= a.NX && y0 >= H && y0 + TY + H <= a.NY;      // uniform
        oob = 0;
        if (interior) {
            const char* base = src + ((size_t)(x0 - H) * a.NY + (y0 - H)) * 8;
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if ((act >> m) & 1u)
                    conv2_dma16(base, rel[m], dst0 + (wave + NW * m) * 1024);
        } else {
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const int u = (wave + NW * m) * 64 + lane;
                const int i = u / RU, c = u - i * RU;
                const int x = x0 - H + i, y = y0 - H + 2 * c;
                const bool in = (unsigned)x < (unsigned)a.NX && (unsigned)y < (unsigned)a.NY;
                const unsigned off = (__umul24(min(max(x, 0), a.NX - 1), a.NY) + min(max(y, 0), a.NY - 2)) * 8u;
                if ((act >> m) & 1u) {
                    oob |= in ? 0u : 1u << m;
                    conv2_dma16(src + off, dst0 + (wave + NW * m) * 1024);
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
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0;
        xin = 0;
        yo = 0;
        if (a.obj.yoff) yo = conv2_ld32(a.obj.yoff + b);
        if (use_tab) {
            const int xg0 = x0 + i0 + (a.obj.xoff ? a.obj.xoff[b] : 0);
            const int* tabrow = a.obj.tab + ((long long)a.obj.angle_of_b[b] * a.obj.S + a.zmod) * a.obj.volNX;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int xg = xg0 + q;
                xin |= (unsigned)xg < (unsigned)a.obj.volNX ? 1u << q : 0u;
                sraw[q] = conv2_ld32(tabrow + min(max(xg, 0), a.obj.volNX - 1));
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) sraw[q] = a.zmod < 0 || a.obj.tab ? 0 : (b * a.obj.S + a.zmod) * a.NX + x0 + i0 + q;
            xin = a.zmod < 0 ? 0u : (1u << R) - 1u;               // zmod < 0: no modulation (row 0 is read and not used)
        }
    };

    // operands of a tile's epilogue: modulation factors, tape (backward) and carrier-field plane of this thread's R outputs
    struct Epi {
        float2 m1[R];
        cf tp[R], pf[R];
        unsigned xin;      // bit q: output row q has a modulation row
        bool yin;          // this thread's y lies inside the volume
    };
    // requests them for `tile` from the table rows in sraw / xin / yo (which have landed)
    auto request_epi = [&](int tile, Epi& e) {
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0, y0 = tp_.y0;
        const int y = y0 + j, yg = y + yo;
        const int yc = min(max(yg, 0), a.obj.volNY - 1);
        e.xin = xin;
        e.yin = yg == yc;
        const cf* tape_b = BWD ? a.tape + (size_t)b * a.NX * a.NY : nullptr;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (sraw[q] < 0) e.xin &= ~(1u << q);                          // a table entry that points nowhere
            const unsigned off = __umul24(x0 + i0 + q, a.NY) + y;
            e.m1[q] = a.obj.vol[(size_t)max(sraw[q], 0) * a.obj.volNY + yc];
            if constexpr (BWD) e.tp[q] = tape_b[off];
            if constexpr (PF) e.pf[q] = a.pfield[off];                   // L2-resident plane shared by all wavefields
        }
    };
    // Order of a tile (vmcnt retires in order):
    //   barrier | DMA of the next halo tile into the OTHER buffer | y pass | barrier | y-pass result in place | barrier |
    //   operands of this tile's epilogue (plain loads) | table rows of the next tile (asm) | x pass | epilogue: hipcc waits
    //   vmcnt(0) for the operands — the DMA, queued before them, has had the whole tile to land — | NS stores | vmcnt(NS).
    // The DMA of tile T + 1 is in flight during ALL of tile T's arithmetic — in k_conv2 only during the x pass, and the what-if
    // timings of that kernel (profiles/r03_conv2_phase_stamps.txt) show memory and arithmetic adding up instead of overlapping.
    constexpr int NS = BWD ? 2 * R : R;          // global stores of one tile's epilogue (distinct rows: never merged)
    if (wg < ntiles) {
        issue(wg, 0);
        request_rows(wg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < R; ++q) asm volatile("" : "+v"(sraw[q]));           // no use of these moves above the wait
    asm volatile("" : "+v"(yo));
    int buf = 0;
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0, y0 = tp_.y0;
        float4* Ab = A4 + buf * ABUF;
        cf* Mb = reinterpret_cast<cf*>(Ab);
        // the halo tile has landed (this wave's pieces: the wait at the end of the previous tile); outside the field the padding constant
        if (oob) {
            const float4 pp = make_float4(a.pad.x, a.pad.y, a.pad.x, a.pad.y);
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if ((oob >> m) & 1u) Ab[(wave + NW * m) * 64 + lane] = pp;
        }
        conv_sync();                     // every wave's pieces are in, and nobody reads the other buffer any more
        const int next = tile + nwg;
        const bool has_next = next < ntiles;
        if (has_next) issue(next, buf ^ 1);
        TapsPtr kt = (TapsPtr)a.taps;
        asm volatile("" : "+s"(kt));
        // pass along y: lanes = 8 windows of 4 outputs along a row, then rows; rows 0..63 by every thread, rows 64..TXH-1 by
        // the first 8 * 2H threads.  Results stay in registers until every window of the tile has been read.
        cf o1[R], o2[R];
        const int w = tid % (TY / R), i1 = tid / (TY / R), i2 = i1 + C::THREADS / (TY / R);
        const bool second = tid < (TXH - C::THREADS / (TY / R)) * (TY / R);
        {
            const float4* p = Ab + i1 * RU + 2 * w;
            cf win[R + 2 * H];
#pragma unroll
            for (int q = 0; q < (R + 2 * H) / 2; ++q) {
                const float4 v = p[q];
                win[2 * q] = make_float2(v.x, v.y);
                win[2 * q + 1] = make_float2(v.z, v.w);
            }
            conv_window<BWD, H, R>(win, kt->ky, o1);
        }
        if (second) {
            const float4* p = Ab + i2 * RU + 2 * w;
            cf win[R + 2 * H];
#pragma unroll
            for (int q = 0; q < (R + 2 * H) / 2; ++q) {
                const float4 v = p[q];
                win[2 * q] = make_float2(v.x, v.y);
                win[2 * q + 1] = make_float2(v.z, v.w);
            }
            conv_window<BWD, H, R>(win, kt->ky, o2);
        }
        conv_sync();                     // all windows read: their rows can be overwritten
#pragma unroll
        for (int q = 0; q < R; ++q) Mb[i1 * MS + R * w + q] = o1[q];
        if (second) {
#pragma unroll
            for (int q = 0; q < R; ++q) Mb[i2 * MS + R * w + q] = o2[q];
        }
        conv_sync();
        Epi cur;
        request_epi(tile, cur);          // in flight during the x pass's tap sums
        if (has_next) request_rows(next);
        asm volatile("" : "+s"(kt));
        // pass along x (window of R consecutive x for one y), then the pointwise physics
        {
            cf o[R];
            const cf ke = make_float2(kt->e.x, kt->e.y);
            cf* out_b = a.out + (size_t)b * a.NX * a.NY;
            float2* grot_b = BWD ? a.grot + ((size_t)b * a.obj.S + a.zmod) * a.NX * a.NY : nullptr;
            cf win[R + 2 * H];
#pragma unroll
            for (int q = 0; q < R + 2 * H; ++q) win[q] = Mb[(i0 + q) * MS + j];
            conv_window<BWD, H, R>(win, kt->kx, o);
            const int y = y0 + j;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const cf acc = BWD ? cmulc(o[q], ke) : cmul(o[q], ke);
                const int x = x0 + i0 + q;
                const unsigned off = __umul24(x, a.NY) + y;
                const bool in = ((cur.xin >> q) & 1u) && cur.yin;
                const float2 mm = make_float2(in ? cur.m1[q].x : 0.f, in ? cur.m1[q].y : 0.f);
                cf car = a.carrier;
                if constexpr (PF) car = cur.pf[q];
                if constexpr (!BWD) {
                    out_b[off] = modulate_eps(acc, car, mm);
                } else {
                    const cf phi = cadd(cur.tp[q], car);
                    const cf tt = cmulc(acc, phi);
                    grot_b[off] = make_float2(a.k * tt.y, -a.k * tt.x);
                    out_b[off] = cmulc(acc, make_float2(1.f + mm.x, mm.y));
                }
            }
        }
        // everything queued before this tile's NS stores has retired: this wave's pieces of the next halo tile too
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
#pragma unroll
        for (int q = 0; q < R; ++q) asm volatile("" : "+v"(sraw[q]));
        asm volatile("" : "+v"(yo));
        buf ^= 1;
    }
}
