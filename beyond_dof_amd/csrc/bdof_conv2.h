// Real-space propagator, second tiling (round 3): the same two 1-D passes and epilogue as k_conv (bdof_kernels.h), same
// order of the tap sums — results equal to float32 rounding of the epilogue — with the data movement rebuilt around LDS-DMA:
//   * 64 x 32 tile (x by y) instead of 32 x 64: the y pass, which also has to produce the x pass's halo rows, computes
//     (64 + 2H) x 32 outputs for 64 x 32 (25 % more at 17 taps) instead of (32 + 2H) x 64 for 32 x 64 (50 % more);
//   * the halo tile goes global -> LDS by global_load_lds_dwordx4 (no staging registers, no stash pass, no per-element
//     index arithmetic: the per-lane source offsets are tile-invariant and computed once); tiles that touch the field's
//     edge clamp their addresses and overwrite what lies outside with the padding constant after the data have landed;
//   * raw image row-major in 16-byte units (two consecutive y), RU = (TY + 2H) / 2 + 1 units per row: an odd unit count
//     puts the y pass's ds_read_b128 (lanes = 8 windows x 8 rows, 32 bytes apart along a row) of neighbouring rows on
//     alternating 16-byte slots, and the DMA image stays lane-linear (the one dummy unit per row is a masked lane);
//   * windows of 4 outputs in both passes: 8 (64 + 2H) y-pass windows in two rounds, 512 x-pass windows — every wave
//     issues the same FMA count in the x pass and the y pass's second round is 2H / 8 waves;
//   * 53.9 KB of LDS in two static objects, 97-117 VGPRs: two workgroups per CU (three need <= 80 registers: 64-112 bytes of
//     scratch per lane, 59.9 / 94.9 us against 54.0 / 57.8 us forward / backward in tools/kbench_conv2.hip);
//   * XCD-aware tile order (runs of strips per blockIdx % 8) so that the halos neighbouring tiles share are L2 hits;
//   * the DMA and the loads carried into the next tile are issued by asm, with one counted wait per tile (below).
// Reference: cnn_propagator/propagation.py:80-107 (the convolution of one slice), :109-110 (renormalisation, k_conv_final).
#pragma once

// Loads whose results are carried into the NEXT tile are issued by asm: hipcc keeps no s_waitcnt bookkeeping for them (met
// again across the loop's back edge and the branches of the DMA issue it can only answer with vmcnt(0), which drains the DMA
// as soon as it is queued, or this tile's stores); the counted wait at the end of a tile retires them, and names their
// registers so that no use moves above it.  "memory": they keep their place between the DMA and the stores.
__device__ __forceinline__ int conv2_ld32(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// The DMA itself is asm for the same reason: hipcc tracks a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the next read of the destination array — at the top of a tile that also waits for the
// previous tile's stores.  M0 (the destination base) is saved and restored around the statement (cdna_hip_programming.md).
// lds_dst: wave-uniform LDS byte address; each lane's 16 bytes land at lds_dst + 16 * lane.
__device__ __forceinline__ void conv2_dma16(const char* base, unsigned voff, unsigned lds_dst) {        // base uniform
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void conv2_dma16(const char* src, unsigned lds_dst) {                        // per-lane address
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// TXV: rows of a tile.  64 (512 threads, two workgroups per CU) is what runs; 32 (256 threads, four workgroups per CU: more
// halo per output, twice as many independent pipelines per CU) was measured slower, 47.0 / 54.5 against 45.0 / 53.6 us
template <int H, int TXV = 64> struct Conv2Cfg {
    static constexpr int TX = TXV, TY = 32, R = 4, SM = 33, THREADS = TX * TY / R, NW = THREADS / 64;
    static constexpr int TXH = TX + 2 * H, TYH = TY + 2 * H, NP = TYH / 2, RU = NP | 1;
    static constexpr int UNITS = TXH * RU, NLOADS = (UNITS + 63) / 64, MP = (NLOADS + NW - 1) / NW;
    static constexpr int A_BYTES = NLOADS * 1024, M_BYTES = TXH * SM * 8, LDS = A_BYTES + M_BYTES;
#ifdef BDOF_CONV2_MINW
    static constexpr int MINW = BDOF_CONV2_MINW; // timing experiment (tools/kbench_conv2.hip)
#else
    static constexpr int MINW = 4;               // waves per SIMD asked of the register allocator: two workgroups of 8 waves per CU
#endif
    static_assert(H % 2 == 0 && NP % 2 == 0, "halo of even width: 16-byte units must not straddle the field's edge");
    static_assert(TX * TY == THREADS * R, "one x-pass window per thread");
};

// In-kernel phase stamps (tools/kbench_conv2.hip builds with -DBDOF_CONV2_STAMP): wave 0 of every workgroup adds the
// cycles (s_memtime) it spent in each phase of its tiles to g_conv2_stamp[phase]; compiled out of the library.
#ifdef BDOF_CONV2_STAMP
__device__ unsigned long long g_conv2_stamp[8];
#define CONV2_STAMP(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                            st_acc[k] += t_ - st_t; st_t = t_; } while (0)
#else
#define CONV2_STAMP(k) do { } while (0)
#endif

template <bool BWD, int H, bool PF = false, int TXV = 64>
__global__ __launch_bounds__((Conv2Cfg<H, TXV>::THREADS), (Conv2Cfg<H, TXV>::MINW)) void k_conv2(ConvArgs a) {
    typedef Conv2Cfg<H, TXV> C;
    constexpr int NW = C::NW;
    constexpr int TX = C::TX, TY = C::TY, R = C::R, TXH = C::TXH, RU = C::RU, NP = C::NP, SM = C::SM, MP = C::MP;
    typedef const __attribute__((address_space(4))) ConvTaps* TapsPtr;
    __shared__ float4 A4[C::NLOADS * 64];                    // raw halo tile [TXH][RU] units of two complex values
    __shared__ cf M[TXH * SM];                               // y-pass result [TXH][SM]
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
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 4)
        rel[m] = (unsigned)(min(i, TXH - 1) * a.NY + 2 * (min(c, NP - 1) % 16) + H) * 8u;    // timing experiment: every row's units from
#else                                                                                          // TWO aligned 128-byte lines
        rel[m] = (unsigned)(min(i, TXH - 1) * a.NY + 2 * min(c, NP - 1)) * 8u;
#endif
        act |= (wave + NW * m < C::NLOADS && i < TXH && c < NP) ? 1u << m : 0u;
    }
    unsigned oob = 0;            // bit m: the unit of load m lies outside the field (padding constant after landing)
    auto issue = [&](int tile) {
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
                    conv2_dma16(base, rel[m], a4_lds + (wave + NW * m) * 1024);
        } else {
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const int u = (wave + NW * m) * 64 + lane;
                const int i = u / RU, c = u - i * RU;
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 4)
                const int x = x0 - H + i, y = y0 + 2 * (c % 16);
#else
                const int x = x0 - H + i, y = y0 - H + 2 * c;
#endif
                const bool in = (unsigned)x < (unsigned)a.NX && (unsigned)y < (unsigned)a.NY;
                const unsigned off = (__umul24(min(max(x, 0), a.NX - 1), a.NY) + min(max(y, 0), a.NY - 2)) * 8u;
                if ((act >> m) & 1u) {
                    oob |= in ? 0u : 1u << m;
                    conv2_dma16(src + off, a4_lds + (wave + NW * m) * 1024);
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
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 1)
            e.m1[q] = make_float2(1e-3f * (float)q, 0.f);                 // timing experiment (tools/kbench_conv2.hip): no operand loads
#else
            e.m1[q] = a.obj.vol[(size_t)max(sraw[q], 0) * a.obj.volNY + yc];
#endif
            if constexpr (BWD) e.tp[q] = tape_b[off];
            if constexpr (PF) e.pf[q] = a.pfield[off];                   // L2-resident plane shared by all wavefields
        }
    };
    // Order of a tile (vmcnt retires in order): operands of its epilogue | y pass | wait for them | DMA of the next halo
    // tile | table rows of the next tile | x pass, epilogue, NS stores | vmcnt(NS): everything but the stores has retired.
    // (Measured and dropped, round 3: the epilogue's operands of the NEXT tile prefetched behind the DMA as well — 16 to 24
    // more live registers, fwd 46.3 -> 51 us, bwd 55.6 -> 63 us per launch of 25 fields of 512 x 512; the arguments read
    // through a laundered kernarg pointer instead of SGPRs spilled to VGPR lanes — 69 v_readlane per tile gone, same time.)
    constexpr int NS = BWD ? 2 * R : R;          // global stores of one tile's epilogue (distinct rows: never merged)
    // (that the NS youngest vector-memory operations before the wait ARE stores is checked on the ISA of every instance by
    // tools/check_spills.py: scan_counted_waits, part of the CPU test suite)
    if (wg < ntiles) {
        issue(wg);
        request_rows(wg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < R; ++q) asm volatile("" : "+v"(sraw[q]));           // no use of these moves above the wait
    asm volatile("" : "+v"(yo));
#ifdef BDOF_CONV2_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const TilePos tp_ = tile_pos(tile);
        const int b = tp_.b, x0 = tp_.x0, y0 = tp_.y0;
        // the epilogue's operands: requested first, in flight across the barrier and the y pass
        Epi cur;
        request_epi(tile, cur);
        // the halo tile has landed; outside the field the padding constant
        if (oob) {
            const float4 pp = make_float4(a.pad.x, a.pad.y, a.pad.x, a.pad.y);
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if ((oob >> m) & 1u) A4[(wave + NW * m) * 64 + lane] = pp;
        }
        CONV2_STAMP(0);          // operand requests, padding patch
        conv_sync();
        CONV2_STAMP(1);          // barrier: the other waves' DMA pieces
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
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 8)
                const float4 v = make_float4(1e-3f * (float)(lane + q), 0.5f, (float)w, 1.f);     // timing experiment: no LDS reads in the passes
#else
                const float4 v = p[q];
#endif
                win[2 * q] = make_float2(v.x, v.y);
                win[2 * q + 1] = make_float2(v.z, v.w);
            }
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 16)
            for (int q = 0; q < R; ++q) o[q] = cadd(win[q], win[q + 2 * H]);                       // timing experiment: no tap sums
#else
            conv_window<BWD, H, R>(win, kt->ky, o);
#endif
#pragma unroll
            for (int q = 0; q < R; ++q) M[i * SM + R * w + q] = o[q];
        }
        CONV2_STAMP(2);          // y pass
        // the epilogue's operands are in registers before the next DMA is queued behind them
#pragma unroll
        for (int q = 0; q < R; ++q) {
            asm volatile("" : "+v"(cur.m1[q].x), "+v"(cur.m1[q].y));
            if constexpr (BWD) asm volatile("" : "+v"(cur.tp[q].x), "+v"(cur.tp[q].y));
            if constexpr (PF) asm volatile("" : "+v"(cur.pf[q].x), "+v"(cur.pf[q].y));
        }
        CONV2_STAMP(3);          // wait for the epilogue's operands
        conv_sync();
        CONV2_STAMP(4);          // barrier: M complete
        const int next = tile + nwg;
        if (next < ntiles) {
            issue(next);                 // A is free: in flight during the x pass and the epilogue
            request_rows(next);
        }
        CONV2_STAMP(5);          // DMA issue
        asm volatile("" : "+s"(kt));
        // pass along x (window of R consecutive x for one y), then the pointwise physics
        {
            cf o[R];
            const cf ke = make_float2(kt->e.x, kt->e.y);
            cf* out_b = a.out + (size_t)b * a.NX * a.NY;
            float2* grot_b = BWD ? a.grot + ((size_t)b * a.obj.S + a.zmod) * a.NX * a.NY : nullptr;
            cf win[R + 2 * H];
#pragma unroll
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 8)
            for (int q = 0; q < R + 2 * H; ++q) win[q] = make_float2(1e-3f * (float)(lane + q), 0.5f);
#else
            for (int q = 0; q < R + 2 * H; ++q) win[q] = M[(i0 + q) * SM + j];
#endif
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 16)
            for (int q = 0; q < R; ++q) o[q] = cadd(win[q], win[q + 2 * H]);
#else
            conv_window<BWD, H, R>(win, kt->kx, o);
#endif
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
#if defined(BDOF_CONV2_WHATIF) && (BDOF_CONV2_WHATIF & 2)
                    const cf res = modulate_eps(acc, car, mm);             // timing experiment: no stores (kept alive by a test
                    if (res.x == 123.456f) out_b[off] = res;               // that practically never passes)
#else
                    out_b[off] = modulate_eps(acc, car, mm);
#endif
                } else {
                    const cf phi = cadd(cur.tp[q], car);
                    const cf tt = cmulc(acc, phi);
                    grot_b[off] = make_float2(a.k * tt.y, -a.k * tt.x);
                    out_b[off] = cmulc(acc, make_float2(1.f + mm.x, mm.y));
                }
            }
        }
        CONV2_STAMP(6);          // x pass, epilogue, stores issued
        // everything queued before this tile's NS stores has retired: the next halo tile and its table rows
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
        CONV2_STAMP(7);          // wait for the next halo tile
#pragma unroll
        for (int q = 0; q < R; ++q) asm volatile("" : "+v"(sraw[q]));
        asm volatile("" : "+v"(yo));
    }
#ifdef BDOF_CONV2_STAMP
    if (tid == 64 * BDOF_CONV2_STAMP)          // -DBDOF_CONV2_STAMP=w: the stamps of wave w
        for (int k = 0; k < 8; ++k) atomicAdd(&g_conv2_stamp[k], st_acc[k]);
#endif
}
