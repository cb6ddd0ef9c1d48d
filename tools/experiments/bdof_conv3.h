// EXPERIMENT, not built into the library (measured in round 3: correct — the parity tests of tests/test_gpu_conv.py passed with
// it — and no faster than k_conv2: forward 45.9 against 45.3 us, backward 54.2 against 53.6 us per launch).  To try it
// again: include it after bdof_conv2.h in csrc/bdof_capi.hip and launch k_conv3<BWD, PF> with k_conv2's grid for 17 taps.
// Real-space propagator, third form (round 3): k_conv2 (bdof_conv2.h) with the halo tile's overlap between consecutive tiles kept
// in LDS.  Measured on k_conv2 (in-kernel stamps): its waves wait ~1 % for memory and spend a quarter of a tile getting
// their DMA pieces accepted — the kernel runs at ~80 % of the rate at which a CU's load path takes bytes, and a third of those
// bytes are halo.  Here a workgroup walks CONSECUTIVE tiles of a strip (y direction); of the 48 columns of the next halo tile the
// first 16 are the last 16 of the current one: they are copied inside LDS and only the 32 new columns come by DMA — 20 pieces of
// 1 KB per tile instead of 30 (the first tile of a workgroup and of a strip fetches all 30).
// For that the image is kept in blocks of 4 column pairs x 16 rows (one DMA piece each: 16 segments of 64 B): 6 pair blocks
// x 5 row blocks = 30 blocks, exactly the 80 x 48 halo tile of 17 taps — no pad units, no masked lanes; a slide moves pair
// blocks 4, 5 to 0, 1 and fills 2..5.  The y pass reads pairs 2w .. 2w + 9 of a row with lanes on 16 consecutive rows
// (ds_read_b128, 256 contiguous bytes per 16 lanes: conflict-free); where a window crosses pair blocks depends on the parity
// of w, so a wave takes windows of one parity (two instances of the pass, immediate offsets in both).
// 17 taps only (H = 8); the other tap counts and fields that are not multiples of 64 x 32 run k_conv2 / k_conv.
// Everything else — passes, epilogue, asm-issued DMA and loop-carried loads, counted wait, XCD-aware runs — as in k_conv2.
#pragma once

template <int V> struct Conv3Int { static constexpr int value = V; };

struct Conv3Cfg {
    static constexpr int H = 8, TX = 64, TY = 32, R = 4, SM = 33, THREADS = 512, NW = 8;
    static constexpr int TXH = TX + 2 * H, TYH = TY + 2 * H, NP = TYH / 2;        // 80 rows, 24 pairs
    static constexpr int RB = TXH / 16, PB = NP / 4, NBLK = RB * PB;               // 5 x 6 = 30 blocks of 64 units
    static constexpr int SLIDE_PB = (TYH - TY) / 2 / 4;                            // 2 pair blocks carried over
    static constexpr int NFULL = NBLK, NSLIDE = (PB - SLIDE_PB) * RB;              // 30 / 20 DMA pieces
    static constexpr int MPF = (NFULL + NW - 1) / NW, MPS = (NSLIDE + NW - 1) / NW;
    static_assert(TXH % 16 == 0 && NP % 4 == 0 && (TYH - TY) % 8 == 0, "block layout");
};

template <bool BWD, bool PF = false>
__global__ __launch_bounds__(Conv3Cfg::THREADS, 4) void k_conv3(ConvArgs a) {
    typedef Conv3Cfg C;
    constexpr int H = C::H, NW = C::NW;
    constexpr int TX = C::TX, TY = C::TY, R = C::R, TXH = C::TXH, SM = C::SM, RB = C::RB, PB = C::PB;
    typedef const __attribute__((address_space(4))) ConvTaps* TapsPtr;
    __shared__ float4 A4[C::NBLK * 64];                      // halo tile: block (pb, rb) at (pb * RB + rb) * 64, unit (dp, di) at dp * 16 + di
    __shared__ cf M[TXH * SM];                               // y-pass result [TXH][SM]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned a4_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)A4;
    // XCD-aware runs of strips as in k_conv2; inside a run a workgroup takes a BLOCK of consecutive tiles (the host sizes the
    // grid so that the blocks are as long as k_conv2's rounds): consecutive tiles are neighbours in y except where a strip ends
    const int tiles_x = a.NX / TX, tiles_y = a.NY / TY;
    const int nstrips = a.B * tiles_x;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const int strip0 = xcd * nstrips / 8;
    const int run_tiles = ((xcd + 1) * nstrips / 8 - strip0) * tiles_y;
    const int per_wg = (run_tiles + nwg - 1) / nwg;
    const int first = wg * per_wg, ntiles = min(run_tiles, first + per_wg);            // this workgroup's tiles: [first, ntiles)
    struct TilePos { int b, x0, y0; };
    auto tile_pos = [&](int l) -> TilePos {
        const int s = strip0 + l / tiles_y;
        const int b = s / tiles_x;
        return TilePos{b, (s - b * tiles_x) * TX, (l % tiles_y) * TY};
    };
    // DMA plan: piece kk = wave + 8 m of a fetch is block (pb, rb) = (pb0 + kk / RB, kk % RB), pb0 = 0 (full) or SLIDE_PB
    // (slide); this lane's unit in a block is row di = lane % 16, pair dp = lane / 16
    const int di = lane & 15, dp = lane >> 4;
    const unsigned lane_off = (unsigned)(di * a.NY + 2 * dp) * 8u;        // bytes from the block's first element
    unsigned oob = 0;            // bit m: this lane's unit of piece m lies outside the field
    auto issue = [&](int tile) {
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0, y0 = tp_.y0;
        const char* src = (const char*)(a.in + (size_t)b * a.NX * a.NY);
        const bool interior = x0 >= H && x0 + TX + H <= a.NX && y0 >= H && y0 + TY + H <= a.NY;      // uniform
        const bool slide = tile != first && y0 != 0;                                                    // uniform
        const int pb0 = slide ? C::SLIDE_PB : 0, npieces = slide ? C::NSLIDE : C::NFULL;
        oob = 0;
#pragma unroll
        for (int m = 0; m < C::MPF; ++m) {
            const int kk = wave + NW * m;
            if (kk < npieces) {                                              // uniform
                const int pb = pb0 + kk / RB, rb = kk - (kk / RB) * RB;
                const unsigned dst = a4_lds + (unsigned)(pb * RB + rb) * 1024u;
                if (slide && pb >= PB - C::SLIDE_PB) {
                    // the block this piece overwrites is the next tile's block (pb - (PB - SLIDE_PB), rb): moved by the wave that
                    // overwrites it, before it does
                    const float4 v = A4[(pb * RB + rb) * 64 + lane];
                    A4[((pb - (PB - C::SLIDE_PB)) * RB + rb) * 64 + lane] = v;
                }
                if (interior) {
                    const char* base = src + ((size_t)(x0 - H + rb * 16) * a.NY + (y0 - H + pb * 8)) * 8;
                    conv2_dma16(base, lane_off, dst);
                } else {
                    const int x = x0 - H + rb * 16 + di, y = y0 - H + pb * 8 + 2 * dp;
                    const bool in = (unsigned)x < (unsigned)a.NX && (unsigned)y < (unsigned)a.NY;
                    const unsigned off = (__umul24(min(max(x, 0), a.NX - 1), a.NY) + min(max(y, 0), a.NY - 2)) * 8u;
                    oob |= in ? 0u : 1u << m;
                    conv2_dma16(src + off, dst);
                }
            }
        }
    };
    // where the padding constant goes once the pieces have landed (same walk as issue)
    auto patch = [&](int tile) {
        const TilePos tp_ = tile_pos(tile);
        const bool slide = tile != first && tp_.y0 != 0;
        const int pb0 = slide ? C::SLIDE_PB : 0;
        const float4 pp = make_float4(a.pad.x, a.pad.y, a.pad.x, a.pad.y);
#pragma unroll
        for (int m = 0; m < C::MPF; ++m)
            if ((oob >> m) & 1u) {
                const int kk = wave + NW * m;
                const int pb = pb0 + kk / RB, rb = kk - (kk / RB) * RB;
                A4[(pb * RB + rb) * 64 + lane] = pp;
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
    constexpr int NS = BWD ? 2 * R : R;          // global stores of one tile's epilogue (checked on the ISA: tools/check_spills.py)
    if (first < ntiles) {
        issue(first);
        request_rows(first);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < R; ++q) asm volatile("" : "+v"(sraw[q]));
    asm volatile("" : "+v"(yo));
    // y pass of one wave-round: rows rb * 16 + di, windows w = 2 wi + PAR (wi = lane / 16): pairs 2w .. 2w + 9 of the row
    auto ypass = [&](int rb, auto par, TapsPtr kt) {
        constexpr int PAR = decltype(par)::value;
        const int wi = lane >> 4, w = 2 * wi + PAR, i = rb * 16 + di;
        const float4* p = A4 + (wi * RB + rb) * 64 + di;                         // row i in pair block wi (pairs 4 wi ..)
        cf win[R + 2 * H], o[R];
#pragma unroll
        for (int q = 0; q < (R + 2 * H) / 2; ++q) {
            const int pq = q + 2 * PAR;                                            // pair 2w + q, counted from the start of block wi
            const float4 v = p[(pq / 4) * (RB * 64) + (pq % 4) * 16];
            win[2 * q] = make_float2(v.x, v.y);
            win[2 * q + 1] = make_float2(v.z, v.w);
        }
        conv_window<BWD, H, R>(win, kt->ky, o);
#pragma unroll
        for (int q = 0; q < R; ++q) M[i * SM + R * w + q] = o[q];
    };
    for (int tile = first; tile < ntiles; ++tile) {
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0, y0 = tp_.y0;
        Epi cur;
        request_epi(tile, cur);
        if (oob) patch(tile);
        conv_sync();
        TapsPtr kt = (TapsPtr)a.taps;
        asm volatile("" : "+s"(kt));
        // 10 wave-rounds (5 row blocks x 2 parities) on 8 waves: round r of wave v is item v + 8 r; parity = item / RB
#pragma unroll 1
        for (int item = wave; item < 2 * RB; item += NW) {
            if (item < RB) ypass(item, Conv3Int<0>(), kt);
            else ypass(item - RB, Conv3Int<1>(), kt);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            asm volatile("" : "+v"(cur.m1[q].x), "+v"(cur.m1[q].y));
            if constexpr (BWD) asm volatile("" : "+v"(cur.tp[q].x), "+v"(cur.tp[q].y));
            if constexpr (PF) asm volatile("" : "+v"(cur.pf[q].x), "+v"(cur.pf[q].y));
        }
        conv_sync();
        const int next = tile + 1;
        if (next < ntiles) {
            issue(next);
            request_rows(next);
        }
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
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
#pragma unroll
        for (int q = 0; q < R; ++q) asm volatile("" : "+v"(sraw[q]));
        asm volatile("" : "+v"(yo));
    }
}
