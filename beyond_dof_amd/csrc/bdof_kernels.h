// Multislice hot-path kernels (gfx950).  See DESIGN.md for the data layout and the byte model.
//
// Two wavefield layouts alternate so that EVERY kernel reads whole contiguous rows:
//   L1[b][x][ky]  rows along the rotation axis y (ky: DFT index along y);  psi_hat = R psi / NY
//   L2[b][ky][x]  rows along x
//   A_i (k_row_fwd) : L1 -> L2   psi = R^-1' psi_hat ; phi = c_i psi ; out = R phi         24 B/px
//   B   (k_row_prop): L2 -> L1   out = C^-1'( h * C in ),  h = ifftshift(H)/(NX NY)        16 B/px
//   A'_i(k_row_bwd) : L1 -> L2   adjoint of A_i, writes the (delta,beta) gradient rows     40 B/px
// A row is transformed by one wave (N <= 512) with LDS exchanges that need no barrier; the LAST
// stage of each kernel's final FFT is executed with a transposed thread mapping (16 rows x N/8
// butterflies), so the transposition costs no extra LDS pass and only one workgroup barrier per
// 16-row tile, and the stores are 128-byte segments.  B's output is directly the tape entry.
#pragma once
#include "bdof_fft.h"

#define BDOF_THREADS 512

template <int N> struct RowCfg {
    static constexpr int T = N / 8;                          // lanes per row
    static constexpr int RPP = BDOF_THREADS / T;             // rows per pass
    // 16 rows: the transposed stores are then 128-byte segments.  8-row tiles (64-byte segments, 39 KB of LDS, three
    // workgroups per CU at 80 VGPRs) were measured in round 2: A 35 -> 55 us, B 26 -> 33 us, A' 50 -> 103 us per launch.
    static constexpr int TILE = RPP > 16 ? RPP : 16;         // rows per tile = transposed segment length
    static constexpr int PASSES = TILE / RPP;
    static constexpr int NPAD = N + N / 16;
    static constexpr int RS = ((NPAD + 29) / 32) * 32 + 2;   // LDS row stride (cf), == 2 mod 32: conflict-free transposed reads
    static constexpr int LDS_CF = TILE * RS;
    static constexpr int MIN_WAVES = N >= 1024 ? 2 : 4;      // waves per SIMD asked from the register allocator
};

// ---------------------------------------------------------------------------------------------
// LDS image of one row: slot(i) = i + (i >> 4)
// ---------------------------------------------------------------------------------------------
template <int T> struct RowLds {
    cf* base;
    __device__ __forceinline__ int slot(int i) const { return i + (i >> 4); }
    __device__ __forceinline__ cf ld_at(int base_slot, int c) const { return base[base_slot + c + (c >> 4)]; }
    __device__ __forceinline__ void st_at(int base_slot, int c, cf v) { base[base_slot + c + (c >> 4)] = v; }
    __device__ __forceinline__ void sync_w2r() { sync(); }
    __device__ __forceinline__ void sync_r2w() { sync(); }
    __device__ __forceinline__ void sync() {
        if constexpr (T <= 64) {
            // the row lives inside one wave: LDS operations of a wave execute in order, only the
            // compiler has to be kept from reordering across the exchange
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Transposed tail: last FFT stage of the TILE rows whose images are in `smem`, then
//   dst[(pos) * ld + r] = scale * out(row r, position pos),  pos = j + q*T
// Called by all threads between two workgroup barriers.
// ---------------------------------------------------------------------------------------------
template <int N, int SIGN, int ROUND = 1, bool EX = BDOF_EX_ALL>
__device__ __forceinline__ void transposed_tail(cf* smem, cf* __restrict__ dst, size_t ld, float scale, const cf* tail, const float* sq = nullptr) {
    typedef RowCfg<N> C;
#pragma nounroll
    for (int pass = 0; pass < C::PASSES; ++pass) {
        const int q = threadIdx.x + pass * BDOF_THREADS;
        const int r = q % C::TILE, j = q / C::TILE;
        RowLds<C::T> lds{smem + r * C::RS};
        cf u[8];
        last_stage<N, SIGN, ROUND, EX>(u, j, lds, tail, sq);
#pragma unroll
        for (int m = 0; m < 8; ++m) dst[(size_t)(j + m * C::T) * ld + r] = cscale(u[m], scale);
    }
}

// ---------------------------------------------------------------------------------------------
// Object access: which (delta,beta) row feeds wavefield row (b, x) at slice z
// ---------------------------------------------------------------------------------------------
struct ObjView {
    const float2* vol;       // rows of volNY modulation factors c - 1 (k_modulation_table of the (delta, beta) rows)
    const int* tab;          // nullable: [n_angles][S][volNX] -> source row (rotation lookup, K1)
    const int* angle_of_b;   // [B] angle index per batch element (with tab)
    const int* xoff;         // nullable [B]: window origin in x (ptychography, K11)
    const int* yoff;         // nullable [B]: window origin in y
    int S, volNX, volNY;
};

__device__ __forceinline__ long long obj_src_row(const ObjView& o, int b, int x, int z, int NX) {
    if (o.tab) {
        int xg = x + (o.xoff ? o.xoff[b] : 0);
        if (xg < 0 || xg >= o.volNX) return -1;
        return (long long)o.tab[((long long)o.angle_of_b[b] * o.S + z) * o.volNX + xg];
    }
    return ((long long)b * o.S + z) * NX + x;
}

// sin x and cos x - 1 with a 3-term Cody-Waite reduction by pi/2 and the cephes minimax polynomials: ~1 ulp for
// |x| < 1e5 (k*delta is the phase picked up in ONE slice).  cos - 1 comes straight out of the polynomial in the
// first quadrant (no cancellation for small arguments); in the other quadrants it is not small anyway.
__device__ __forceinline__ void sin_cosm1(float x, float& s, float& cm1) {
    const float n = rintf(x * 0.636619772367581343f);
    float r = fmaf(-n, 1.5703125f, x);
    r = fmaf(-n, 4.837512969970703125e-4f, r);
    r = fmaf(-n, 7.54978995489188216e-8f, r);
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cpm1 = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                            -0.5f * r2);                      // cos r - 1
    const int q = (int)n & 3;
    const float cp = 1.0f + cpm1;
    s = q == 0 ? sp : (q == 1 ? cp : (q == 2 ? -sp : -cp));
    cm1 = q == 0 ? cpm1 : (q == 1 ? -sp - 1.0f : (q == 2 ? -2.0f - cpm1 : sp - 1.0f));
}

// c = exp(i k delta) * exp(-k beta)                      cnn_propagator/np_funcs.py:39
__device__ __forceinline__ cf slice_modulation(float2 db, float k) {
    float s, cm1;
    sin_cosm1(k * db.x, s, cm1);
    const float e = __expf(-k * db.y);
    return make_float2(fmaf(e, cm1, e), e * s);
}

// cm1 = exp(i k delta) * exp(-k beta) - 1, free of cancellation for small arguments   (c: cnn_propagator/np_funcs.py:39)
//   Re = expm1(-k beta) cos(k delta) + (cos(k delta) - 1) ;  Im = exp(-k beta) sin(k delta)
__device__ __forceinline__ cf slice_modulation_m1(float2 db, float k) {
    float s, cm1;
    sin_cosm1(k * db.x, s, cm1);
    const float y = -k * db.y;
    const float e = __expf(y);
    const float poly = y * fmaf(y * 0.5f, fmaf(y * (1.f / 3.f), fmaf(y * 0.25f, fmaf(y, 0.2f, 1.f), 1.f), 1.f), 1.f);
    const float em1 = fabsf(y) < 0.1f ? poly : e - 1.f;
    return make_float2(fmaf(em1, 1.0f + cm1, cm1), e * s);
}

// Carrier splitting: the wave is held as psi = a + eps with a known complex scalar a per slice (the plane-wave part,
// propagated exactly on the host in float64) and only eps goes through the float32 FFTs, so round-off scales with the
// scattered field instead of the full wave.  c (a + eps) = a + [eps + (c-1)(a + eps)].
__device__ __forceinline__ cf modulate_eps(cf eps, cf carrier, cf cm1) {
    return cadd(eps, cmul(cm1, cadd(carrier, eps)));
}
// Mean-refraction carrier.  The plane-wave part is not only propagated (a_{z+1} = a_z H00) but also modulated by the MEAN
// modulation factor cbar of the object (mean of c over the volume, k_modulation_table): entering slice z the wave is
// a_z + eps_z with a_z = a_0 (cbar H00)^z, leaving it cbar a_z + eps'_z with
//     eps' = c (a + eps) - cbar a = eps + (c - 1)(a + eps) - a (cbar - 1)          (cshift = a (cbar - 1), host float64).
// Exact algebra for ANY cbar; with cbar the object's mean, eps no longer collects the mean phase shift / absorption of the
// slices passed (20 x the structured part for a random 512-slice object), so the round-off of the float32 transforms,
// which scales with |eps|, drops by that factor — what limits far-field residuals and the gradient.
__device__ __forceinline__ cf modulate_eps_s(cf eps, cf carrier, cf cm1, cf cshift) {
    return csub(cadd(eps, cmul(cm1, cadd(carrier, eps))), cshift);
}

// The modulation factors of the whole object, once per object update (one pass, 16 B/voxel): every voxel row is used by
// ~25 angles per Adam step, so evaluating sincos/exp here instead of in A and A' removes a third of their arithmetic.
// partial (nullable): [gridDim.x] per-workgroup sums of c - 1 in float64, summed in a fixed order by k_sum_mean (deterministic)
__global__ __launch_bounds__(256) void k_modulation_table(const float2* __restrict__ db, float2* __restrict__ cm1, size_t n, float k,
                                                           double2* __restrict__ partial) {
    double sx = 0.0, sy = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = slice_modulation_m1(db[i], k);
        cm1[i] = v;
        sx += (double)v.x;
        sy += (double)v.y;
    }
    if (partial) {
        __shared__ double w[2][4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off, 64); sy += __shfl_down(sy, off, 64); }
        if ((threadIdx.x & 63) == 0) { w[0][threadIdx.x >> 6] = sx; w[1][threadIdx.x >> 6] = sy; }
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = make_double2(w[0][0] + w[0][1] + w[0][2] + w[0][3], w[1][0] + w[1][1] + w[1][2] + w[1][3]);
    }
}
// 256 threads, fixed order (thread t sums partial[t], partial[t + 256], ...; then a fixed tree): deterministic.  (One thread
// summing the 4096 partials of a 512^3 table took 0.40 ms per Adam step — as long as the table pass itself.)
__global__ __launch_bounds__(256) void k_sum_mean(const double2* partial, int n, double inv_count, double2* out) {
    __shared__ double sx[256], sy[256];
    double ax = 0.0, ay = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) { ax += partial[j].x; ay += partial[j].y; }
    sx[threadIdx.x] = ax;
    sy[threadIdx.x] = ay;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { sx[threadIdx.x] += sx[threadIdx.x + w]; sy[threadIdx.x] += sy[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = make_double2(sx[0] * inv_count, sy[0] * inv_count);
}

// Loads are unconditional (clamped address) and zeroed afterwards: a branch around each load makes hipcc wait
// vmcnt(0) per element, i.e. eight dependent memory round trips per row.
__device__ __forceinline__ void load_obj_row(const ObjView& o, long long srow, int y0, int tid, int T, float2 (&db)[8]) {
    const float2* row = o.vol + (size_t)(srow >= 0 ? srow : 0) * o.volNY;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int yg = tid + m * T + y0;
        const int yc = min(max(yg, 0), o.volNY - 1);
        const float2 v = row[yc];
        const bool in = srow >= 0 && yg == yc;
        db[m] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// A: forward row kernel.                              cnn_propagator/np_funcs.py:37-40 (+ FFT along y)
// ---------------------------------------------------------------------------------------------
struct RowFwdArgs {
    const cf* in;      // L1 psi_hat_i [B][NX][NY]; ignored when FIRST (probe used)
    const cf* probe;   // [NX][NY] real-space probe
    cf* out;           // TSTORE: L2 [B][NY][NX] = R phi_i ; else L1 [B][NX][NY]
    cf* phi_out;       // nullable: tape of the real-space scattered part of phi_i, [B][NX][NY] (read back by A'_i)
    ObjView obj;
    int B, NX, z;
    float k;
    cf carrier;        // a_z: constant part of the wave entering slice z
    const cf* twiddle;
    const cf* pz;      // PF kernels: carrier FIELD of slice z, [NX][NY] (bdof_set_probe_stack); replaces `carrier`
    cf cshift;         // a_z (cbar - 1): see modulate_eps_s (0 with a carrier field)
    int real_in;       // INV kernels: `in` is real-space (no inverse transform first)
    float in_scale;    // INV kernels: factor on the (transformed) input
    int probe_batched; // FIRST kernels: `probe` is [B][NX][NY], one starting field per wavefield (bdof_forward_range)
    float sq[2];       // sqrt(1/2) of this launch for the transforms with ROUND 1 / ROUND 2 (dithered over the slices, bdof_fft.h)
    int pz_b;          // PF kernels: rows per wavefield in `pz` — 0: one carrier field shared by the batch; NX: [B][NX][NY], one per
                       // wavefield (the tiles of a stitch range, each riding on its own free-space propagation: bdof_set_range_carrier)
};

// Inverse of the modulation, for the tape-free adjoint (bdof_configure flag 16): from the scattered part of phi_z = c psi_z
// back to the scattered part of psi_z,  eps = (eps' - carrier (c - 1) + cshift) / c     (modulate_eps_s solved for eps).
__device__ __forceinline__ cf unmodulate_eps(cf epsp, cf carrier, cf cm1, cf cshift) {
    const cf num = cadd(csub(epsp, cmul(cm1, carrier)), cshift);
    const float cr = 1.f + cm1.x, ci = cm1.y;
    const float inv = 1.f / fmaf(cr, cr, ci * ci);
    return make_float2((num.x * cr + num.y * ci) * inv, (num.y * cr - num.x * ci) * inv);
}

// PF: the carrier is a field (localised probe), one more coalesced 8-B read per pixel; the plane-wave instances are untouched
// INV: A_z^-1 — rows of the hybrid (or, real_in, real-space) scattered part of phi_z in, R eps(psi_z) out: the first half
// of the step phi_z -> phi_{z-1} = P^H (phi_z / c_z) that marches the forward wave back beside the adjoint field.
template <int NY, bool FIRST, bool TSTORE, bool PF = false, bool INV = false>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NY>::MIN_WAVES) void k_row_fwd(RowFwdArgs a) {
    typedef RowCfg<NY> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    constexpr bool EX = BDOF_EX_FWD_A;
    __shared__ cf smem_tw[FftTw<NY>::LDS_CNT * (EX ? 2 : 1)];
    FftTw<NY> tw;
    __shared__ cf smem_tail[7 * C::T * (EX ? 2 : 1)];
    tw.template load<EX>(a.twiddle, tid, smem_tw, smem_tail);
    tw.sq = a.sq;
    const int ntiles = a.B * a.NX / C::TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NX, x0 = row0 - b * a.NX;
        const int y0 = a.obj.yoff ? a.obj.yoff[b] : 0;
#pragma nounroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            const int x = x0 + r;
            RowLds<C::T> lds{smem + r * C::RS};
            cf u[8];
            float2 db[8];
            const cf* src = FIRST ? a.probe + (a.probe_batched ? (size_t)(row0 + r) : (size_t)x) * NY : a.in + (size_t)(row0 + r) * NY;
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = src[tid + m * C::T];
            load_obj_row(a.obj, obj_src_row(a.obj, b, x, a.z, a.NX), y0, tid, C::T, db);
            if constexpr (PF) {
                cf pc[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) pc[m] = a.pz[((size_t)b * a.pz_b + x) * NY + tid + m * C::T];
                if constexpr (INV) {
                    if (!a.real_in) line_fft<NY, +1, 1, EX>(u, tw, tid, lds);
#pragma unroll
                    for (int m = 0; m < 8; ++m) u[m] = unmodulate_eps(cscale(u[m], a.in_scale), pc[m], db[m], make_float2(0.f, 0.f));
                } else {
                    if constexpr (!FIRST) line_fft<NY, +1, 1, EX>(u, tw, tid, lds);
#pragma unroll
                    for (int m = 0; m < 8; ++m) u[m] = modulate_eps(u[m], pc[m], db[m]);
                }
            } else if constexpr (INV) {
                if (!a.real_in) line_fft<NY, +1, 1, EX>(u, tw, tid, lds);
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = unmodulate_eps(cscale(u[m], a.in_scale), a.carrier, db[m], a.cshift);
            } else {
                if constexpr (!FIRST) line_fft<NY, +1, 1, EX>(u, tw, tid, lds);
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = modulate_eps_s(u[m], a.carrier, db[m], a.cshift);
            }
            if (a.phi_out) {
                cf* pdst = a.phi_out + (size_t)(row0 + r) * NY;
#pragma unroll
                for (int m = 0; m < 8; ++m) pdst[tid + m * C::T] = u[m];
            }
            if constexpr (TSTORE) {
                line_fft_partial<NY, -1, 1, EX>(u, tw, tid, lds);
            } else {
                line_fft<NY, -1, 1, EX>(u, tw, tid, lds);
                cf* dst = a.out + (size_t)(row0 + r) * NY;
#pragma unroll
                for (int m = 0; m < 8; ++m) dst[tid + m * C::T] = u[m];
            }
        }
        if constexpr (TSTORE) {
            __syncthreads();
            transposed_tail<NY, -1, 1, EX>(smem, a.out + (size_t)b * NY * a.NX + x0, a.NX, 1.f, smem_tail, a.sq);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// B: Fresnel transfer-function step along x.          cnn_propagator/np_funcs.py:42 (K3-K5)
// ---------------------------------------------------------------------------------------------
struct RowPropArgs {
    const cf* in;    // L2 [B][NY][NX]
    cf* out;         // L1 [B][NX][NY]
    const cf* h;     // h[ky][kx] = ifftshift(H)[ky][kx] / (NX NY)
    int B, NY;
    float scale;     // extra factor applied with h
    int conj_h;      // adjoint step uses conj(h)
    const cf* twiddle;
    float sq[2];     // sqrt(1/2) of this launch (RowFwdArgs)
};

// EX: exact transform constants (hi + lo twiddles, bdof_fft.h) — the instance the adjoint sweep launches
template <int NX, bool EX = BDOF_EX_ALL>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NX>::MIN_WAVES) void k_row_prop(RowPropArgs a) {
    typedef RowCfg<NX> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    __shared__ cf smem_tw[FftTw<NX>::LDS_CNT * (EX ? 2 : 1)];
    FftTw<NX> tw;
    __shared__ cf smem_tail[7 * C::T * (EX ? 2 : 1)];
    // (Filling the tables under the first row's loads instead of in front of them — `if (!tables) tw.load(...)` after the loads — was
    // measured twice: round 3 in the bench, round 4 alternating in tools/kbench, best of 7: 23.2-24.5 us as here, 24.1-25.1 us with
    // the fill deferred.  The fill is short; the branch and the later barrier cost more than the overlap returns.)
    tw.template load<EX>(a.twiddle, tid, smem_tw, smem_tail);
    tw.sq = a.sq;
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
            line_fft_partial<NX, +1, 2, EX>(u, tw, tid, lds);  // constants rounded up here, down elsewhere (bdof_fft.h)
        }
        __syncthreads();
        transposed_tail<NX, +1, 2, EX>(smem, a.out + (size_t)b * NX * a.NY + ky0, a.NY, 1.f, smem_tail, a.sq);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// block reduction of a per-thread double: wavefront shuffles, then one LDS hop
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// two sums at once: dst[0] = sum v, dst[1] = sum v2
__device__ __forceinline__ void block_store_sum2(double v, double v2, double* dst) {
    __shared__ double wsum[2][BDOF_THREADS / 64];
    v = wave_reduce_sum(v);
    v2 = wave_reduce_sum(v2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { wsum[0][wid] = v; wsum[1][wid] = v2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0, s2 = 0;
        for (int j = 0; j < BDOF_THREADS / 64; ++j) { s += wsum[0][j]; s2 += wsum[1][j]; }
        dst[0] = s;
        dst[1] = s2;
    }
}

// ---------------------------------------------------------------------------------------------
// Detector plane: magnitude loss + adjoint seed.
//   loss: cnn_propagator/fullfield.py:106 ; seed G(d) = 2 (|d| - |m|) d/|d| / n   
// k_row_loss   (free_prop_cm None / distance): rows of L1, d = R^-1' in * in_scale (real space).
// k_row_loss_far (free_prop_cm 'inf'):         rows of L2, d = C in * in_scale  = un-shifted fft2;
//   the fftshift of np_funcs.py:48 is folded into the order in which the host lays out `meas`.
// ---------------------------------------------------------------------------------------------
struct LossArgs {
    const cf* in;
    cf* out_hyb;         // nullable: seed transformed back (* out_scale); layout by TSTORE
    cf* out_wave;        // nullable: detector wave, same row layout as `in`
    const float* meas;   // nullable: |measured|, same row layout as `in`
    double* partial;     // [2 * gridDim.x] per-workgroup sums of (|d|-|m|)^2 and of (|d|-|m|) |d|
    int B, R;            // R rows per batch element
    float in_scale, out_scale, seed_scale;
    cf carrier;          // constant part of the detector wave (far field: its DC bin value a*NX*NY)
    const cf* twiddle;
    const cf* pfield;    // nullable: carrier field at this plane in the row layout of `in` ([R][N], the same for every batch
                         // element); replaces `carrier`
    int meas_dev;        // `meas` holds m - |carrier| (loss_seed_dev); real-space detectors with a scalar carrier only
    // Adjoint carrier (far field + plane-wave carrier, see AdjCarrier below): the DC bin is evaluated in float64 and its seed
    // is NOT sent through the transforms; it is left in gcar[b] (and conj(a_end) * seed in gt0[b]) for the adjoint kernels.
    double2* gcar;       // nullable [B]
    double2* gt0;        // [B]
    double2 carrier_dd;  // a_end * NX * NY in float64 (= `carrier`)
    double2 a_end;       // a_end: constant part of phi_{S-1}
    float dref;          // meas_dev: |carrier| - reference subtracted by the host (loss_seed_dev)
    const double2* pfield64;   // nullable: `pfield` in float64 — the residual is then formed in float64 (loss_seed_f64)
    double2 pscale;            // complex factor on pfield64 (the real-space propagator's renormalisation s, else 1)
};

// Adjoint carrier.  With a plane-wave probe and a far-field detector nearly all of the detector wave sits in ONE bin (DC,
// |d| ~ NX NY), so the adjoint field G = F^H seed is a large constant gamma plus a small structured part Gamma.  The
// gradient is k Im / -k Re of conj(phi) G with phi = a + e, and the product of the two constants, conj(a) gamma, is
// (nearly) real: in float32 its imaginary part is round-off of size 1e-7 |a| |gamma|, which swamps the true delta-gradient
// (seen: 25 % error at 128^2).  So gamma is split off exactly as the forward carrier is: the DC seed is formed in float64 by
// the loss kernel and never enters the float32 transforms; it is carried as a scalar per wavefield, gamma_z = gamma_S
// conj(cbar H00)^(slices back) (float64), conj(c)(gamma + Gamma) = conj(cbar) gamma + [conj(c) Gamma + conj(c - cbar) gamma],
// and conj(a_z) gamma_z = conj(a_end) gamma_S =: t0 (constant parts of phi_z and of G(phi_z)) is the same at every slice and
// is formed once, in float64.
struct AdjCarrier {
    const double2* gcar;   // nullable [B]: gamma at the plane the adjoint sweep starts from
    const double2* gt0;    // [B]: conj(a_end) * gamma
    double2 fac;           // conj(cbar H00)^(number of slices between that plane and this one): what gamma picks up on the way
    cf cbm1;               // cbar - 1 (mean-refraction carrier, modulate_eps_s)
};
__device__ __forceinline__ void adj_carrier_load(const AdjCarrier& ac, int b, cf& gam, cf& t0) {
    const double2 g = ac.gcar[b], t = ac.gt0[b];
    gam = make_float2((float)(g.x * ac.fac.x - g.y * ac.fac.y), (float)(g.x * ac.fac.y + g.y * ac.fac.x));
    t0 = make_float2((float)t.x, (float)t.y);
}

__device__ __forceinline__ cf loss_seed(cf d, float m, float seed_scale, double& acc, double& acc2) {
    const float a = sqrtf(d.x * d.x + d.y * d.y);
    const float r = a - m;
    acc += (double)r * (double)r;
    acc2 += (double)r * (double)a;
    const float f = a > 0.f ? seed_scale * r / a : 0.f;
    return make_float2(d.x * f, d.y * f);
}

// Residual splitting at the detector (bdof_set_meas_mode 1): with the wave held as d = a + e (a the plane-wave part, known
// exactly) the residual |d| - m is the difference of two float32 numbers of size one and carries 7e-8 of absolute error,
// i.e. 3e-6 of a 2 % residual.  |a + e| - |a| = (2 Re(conj(a) e) + |e|^2) / (|a + e| + |a|) has no cancellation, and the host
// hands the measurement over as mdev = m - |a| (float64 subtraction, then float32): r = (|d| - |a|) - mdev is as accurate
// as the scattered wave e itself.
// dref = |a| - (the reference the host subtracted): the host subtracts |a_0|, the carrier at the detector has modulus
// |a_0| |cbar|^S (mean-refraction carrier), the difference (float64 on the host) is added back here.
__device__ __forceinline__ cf loss_seed_dev(cf e, cf a, float abs_a, float mdev, float seed_scale, double& acc, double& acc2, float dref = 0.f) {
    const cf d = cadd(a, e);
    const float ab = sqrtf(d.x * d.x + d.y * d.y);
    const float q = fmaf(2.f * a.x, e.x, fmaf(2.f * a.y, e.y, fmaf(e.x, e.x, e.y * e.y)));
    const float r = q / (ab + abs_a) - (mdev - dref);
    acc += (double)r * (double)r;
    acc2 += (double)r * (double)ab;
    const float f = ab > 0.f ? seed_scale * r / ab : 0.f;
    return make_float2(d.x * f, d.y * f);
}

// Carrier FIELD at the detector (localised probe): the wave is d = p + e with p the probe's own propagation, known in float64
// (bdof_set_probe_field), and e the scattered part that came through the float32 transforms.  Rounding p to float32 first
// costs 6e-8 |d| — 3e-6 of a 2 % residual, the largest single term of the gradient's error in far-field ptychography
// (tools/precision_model.py) — so |d| - m is formed here in float64 and only the finished seed is rounded.
__device__ __forceinline__ cf loss_seed_f64(cf e, double2 p, float m, float seed_scale, double& acc, double& acc2, cf& d_out) {
    const double dx = p.x + (double)e.x, dy = p.y + (double)e.y;
    const double ab = sqrt(dx * dx + dy * dy), r = ab - (double)m;
    acc += r * r;
    acc2 += r * ab;
    const double f = ab > 0.0 ? (double)seed_scale * r / ab : 0.0;
    d_out = make_float2((float)dx, (float)dy);
    return make_float2((float)(dx * f), (float)(dy * f));
}

// FAR = false: inverse FFT first (rows of L1); FAR = true: forward FFT first (rows of L2).
template <int N, bool FAR, bool TSTORE>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<N>::MIN_WAVES) void k_row_loss(LossArgs a) {
    typedef RowCfg<N> C;
    constexpr int S1 = FAR ? -1 : +1;     // direction of the first transform; the second is the opposite
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    constexpr bool EX = BDOF_EX_DET;
    __shared__ cf smem_tw[FftTw<N>::LDS_CNT * (EX ? 2 : 1)];
    FftTw<N> tw;
    __shared__ cf smem_tail[7 * C::T * (EX ? 2 : 1)];
    tw.template load<EX>(a.twiddle, tid, smem_tw, smem_tail);
    const int ntiles = a.B * a.R / C::TILE;
    double acc = 0.0, acc2 = 0.0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.R, r0 = row0 - b * a.R;
#pragma nounroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            RowLds<C::T> lds{smem + r * C::RS};
            const size_t off = (size_t)(row0 + r) * N;
            cf u[8];
            float mm[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = a.in[off + tid + m * C::T];
            if (a.meas) {
#pragma unroll
                for (int m = 0; m < 8; ++m) mm[m] = a.meas[off + tid + m * C::T];
            }
            line_fft<N, S1, 1, EX>(u, tw, tid, lds);
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = cscale(u[m], a.in_scale);
            const bool dev = !FAR && a.meas_dev && a.meas && !a.pfield;
            cf e0 = make_float2(0.f, 0.f);        // FAR: scattered part of the DC bin
            if (a.pfield64) {
                const double2* pf = a.pfield64 + (size_t)(r0 + r) * N;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const double2 q = pf[tid + m * C::T];
                    const double2 p = make_double2(q.x * a.pscale.x - q.y * a.pscale.y, q.x * a.pscale.y + q.y * a.pscale.x);
                    cf dw;
                    if (a.meas) u[m] = loss_seed_f64(u[m], p, mm[m], a.seed_scale, acc, acc2, dw);
                    else dw = u[m] = make_float2((float)(p.x + (double)u[m].x), (float)(p.y + (double)u[m].y));
                    if (a.out_wave) a.out_wave[off + tid + m * C::T] = dw;
                }
            } else if (dev) {
                // u is still the scattered part: the seed comes from (e, a) directly, the detector wave is a + e
                const float abs_a = sqrtf(a.carrier.x * a.carrier.x + a.carrier.y * a.carrier.y);
                if (a.out_wave) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) a.out_wave[off + tid + m * C::T] = cadd(u[m], a.carrier);
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) u[m] = loss_seed_dev(u[m], a.carrier, abs_a, mm[m], a.seed_scale, acc, acc2, a.dref);
            } else {
                if (a.pfield) {
                    const cf* pf = a.pfield + (size_t)(r0 + r) * N;
#pragma unroll
                    for (int m = 0; m < 8; ++m) u[m] = cadd(u[m], pf[tid + m * C::T]);
                } else if constexpr (FAR) {
                    if (r0 + r == 0 && tid == 0) { e0 = u[0]; u[0] = cadd(u[0], a.carrier); }       // DC bin of the un-shifted fft2
                } else {
#pragma unroll
                    for (int m = 0; m < 8; ++m) u[m] = cadd(u[m], a.carrier);
                }
                if (a.out_wave) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) a.out_wave[off + tid + m * C::T] = u[m];
                }
                if (a.meas) {
                    if constexpr (FAR) {
                        if (a.gcar && r0 + r == 0 && tid == 0) {
                            // DC bin in float64; its seed stays out of the transforms (AdjCarrier).  u[0] = (m, 0) makes the
                            // float32 path below contribute exactly nothing for this bin.
                            const double dx = a.carrier_dd.x + (double)e0.x, dy = a.carrier_dd.y + (double)e0.y;
                            const double ab = sqrt(dx * dx + dy * dy), rr = ab - (double)mm[0];
                            acc += rr * rr;
                            acc2 += rr * ab;
                            const double f = ab > 0.0 ? (double)a.seed_scale * rr / ab : 0.0;
                            const double2 s0 = make_double2(dx * f, dy * f);
                            a.gcar[b] = s0;
                            a.gt0[b] = make_double2(a.a_end.x * s0.x + a.a_end.y * s0.y, a.a_end.x * s0.y - a.a_end.y * s0.x);
                            u[0] = make_float2(mm[0], 0.f);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < 8; ++m) u[m] = loss_seed(u[m], mm[m], a.seed_scale, acc, acc2);
                }
            }
            if (a.out_hyb) {
                if constexpr (TSTORE) {
                    line_fft_partial<N, -S1, 1, EX>(u, tw, tid, lds);
                } else {
                    line_fft<N, -S1, 1, EX>(u, tw, tid, lds);
#pragma unroll
                    for (int m = 0; m < 8; ++m) a.out_hyb[off + tid + m * C::T] = cscale(u[m], a.out_scale);
                }
            }
        }
        if constexpr (TSTORE) {
            if (a.out_hyb) {
                __syncthreads();
                transposed_tail<N, -S1, 1, EX>(smem, a.out_hyb + (size_t)b * N * a.R + r0, a.R, a.out_scale, smem_tail);
                __syncthreads();
            }
        }
    }
    if (a.meas) block_store_sum2(acc, acc2, a.partial + 2 * blockIdx.x);
}

__global__ void k_sum_partials(const double* partial, int n, double scale, double* out) {
    double v = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) v += partial[2 * j];
    __shared__ double wsum[4];
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale;
}

// ---------------------------------------------------------------------------------------------
// A': backward row kernel (hand-derived adjoint of A):
//   G(phi) = R^-1' g_hat ; phi = a_z + tape (written by A_z) ; t = conj(phi) G(phi)
//   g_delta = k Im t ; g_beta = -k Re t ; G(psi) = conj(c) G(phi) ; out = R G(psi)
// ---------------------------------------------------------------------------------------------
struct RowBwdArgs {
    const cf* gin;     // L1 g_hat(phi_i) [B][NX][NY]
    const cf* tape;    // real-space scattered part of phi_i, [B][NX][NY]
    cf* gout;          // nullable: L2 R G(psi_i)
    float2* grot;      // [B][S][NX][NY] (g_delta, g_beta) in the rotated / windowed frame
    ObjView obj;
    int B, NX, z;
    float k;
    cf carrier;        // a_z: constant part of psi_z
    const cf* twiddle;
    const cf* pz;      // PF kernels: carrier field of slice z, [NX][NY]
    AdjCarrier ac;     // GC kernels: constant part of the adjoint field (far field + plane-wave carrier)
    cf cshift;         // a_z (cbar - 1)
    cf carrier_phi;    // cbar a_z: constant part of phi_z
    float tape_scale;  // HIST 3
    cf* gpsi0;         // nullable: G(psi_z) in real space, [B][NX][NY] — at z = 0 the gradient w.r.t. the probe per wavefield
                       // (probe_real / probe_imag of tensorflow_recon/fullfield.py:311-327 as optimisation variables)
    int grot_S, grot_z;  // gradient rows go to grot[b][grot_z][x][y] of a [B][grot_S][NX][NY] buffer (= obj.S, z unless the
                         // sweep covers a slice range with a buffer of its own, bdof_adjoint_range)
    float sq[2];         // sqrt(1/2) of this launch (RowFwdArgs)
};

// HIST = 0: phi_z (scattered part) is read from the tape A_z wrote.  HIST = 1: the tape holds the per-slice history
// psi_hat_z (the INPUT of A_z, written by the transfer-function kernel anyway) and phi_z is recomputed here with A_z's own
// operations (inverse transform, modulation) — one more transform per launch, 8 B per pixel less traffic in A.
// HIST = 2: slice 0 of that mode, phi_0 from the probe (no transform).
// HIST = 3: tape-free adjoint — `tape` is the hybrid scattered part of phi_z itself, marched back by A_{z+1}^-1 and the
// adjoint transfer-function step (one inverse transform here, no modulation); tape_scale multiplies it.
template <int NY, int HIST, bool PF = false, bool GC = false>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NY>::MIN_WAVES) void k_row_bwd(RowBwdArgs a) {
    typedef RowCfg<NY> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    // the transforms of the ADJOINT field run with exact constants (EX, bdof_fft.h); re-deriving phi from the tape does not
    constexpr bool EX = BDOF_EX_ADJ;
    // phi_z is re-derived from the tape with plain float32 tables: it is ONE transform of the scattered part (no chain for a
    // table error to add up along, and phi = carrier + scattered part enters the gradient only through conj(phi) G), so the
    // exact constants of A_z's own inverse transform buy nothing here and cost 2.4 us per launch (1.2 ms per cfg3 step)
    constexpr bool EXP = false;
    constexpr bool LO = EX || EXP;
    __shared__ cf smem_tw[FftTw<NY>::LDS_CNT * (LO ? 2 : 1)];
    FftTw<NY> tw;
    __shared__ cf smem_tail[7 * C::T * (LO ? 2 : 1)];
    tw.template load<LO>(a.twiddle, tid, smem_tw, smem_tail);
    tw.sq = a.sq;
    const int ntiles = a.B * a.NX / C::TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NX, x0 = row0 - b * a.NX;
        const int y0 = a.obj.yoff ? a.obj.yoff[b] : 0;
        cf gam = make_float2(0.f, 0.f), t0 = make_float2(0.f, 0.f);
        if constexpr (GC) adj_carrier_load(a.ac, b, gam, t0);
#pragma nounroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            const int x = x0 + r;
            RowLds<C::T> lds{smem + r * C::RS};
            const size_t off = (size_t)(row0 + r) * NY;
            cf g[8], p[8];
            float2 db[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) g[m] = a.gin[off + tid + m * C::T];
            const cf* psrc = HIST == 2 ? a.tape + (size_t)x * NY : a.tape + off;      // HIST 2: `tape` is the probe [NX][NY]
#pragma unroll
            for (int m = 0; m < 8; ++m) p[m] = psrc[tid + m * C::T];
            load_obj_row(a.obj, obj_src_row(a.obj, b, x, a.z, a.NX), y0, tid, C::T, db);
            cf pc[PF ? 8 : 1];
            if constexpr (PF) {
#pragma unroll
                for (int m = 0; m < 8; ++m) pc[m] = a.pz[(size_t)x * NY + tid + m * C::T];
            }
            if constexpr (HIST == 1 || HIST == 3) line_fft<NY, +1, 1, EXP>(p, tw, tid, lds);   // psi_hat_z -> psi_z (scattered part)
            if constexpr (HIST == 3) {
#pragma unroll
                for (int m = 0; m < 8; ++m) p[m] = cscale(p[m], a.tape_scale);
            }
            if constexpr (HIST == 1 || HIST == 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) p[m] = PF ? modulate_eps(p[m], pc[m], db[m]) : modulate_eps_s(p[m], a.carrier, db[m], a.cshift);
            }
            line_fft<NY, +1, 1, EX>(g, tw, tid, lds);
            float2* gdst = a.grot + (((size_t)b * a.grot_S + a.grot_z) * a.NX + x) * NY;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const cf phi = cadd(p[m], PF ? pc[m] : a.carrier_phi);
                cf t = cmulc(g[m], phi);                             // G * conj(phi)
                if constexpr (GC) t = cadd(cadd(t, cmulc(gam, p[m])), t0);      // + gamma conj(e) + gamma conj(a)
                gdst[tid + m * C::T] = make_float2(a.k * t.y, -a.k * t.x);
                g[m] = cmulc(g[m], make_float2(1.f + db[m].x, db[m].y));    // conj(c) G,  c = 1 + (c - 1)
                if constexpr (GC) g[m] = cadd(g[m], cmulc(gam, csub(db[m], a.ac.cbm1)));   // + conj(c - cbar) gamma (conj(cbar) gamma rides on)
            }
            if (a.gpsi0) {
                cf* pd = a.gpsi0 + off;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    pd[tid + m * C::T] = GC ? cadd(g[m], cmulc(gam, make_float2(1.f + a.ac.cbm1.x, a.ac.cbm1.y))) : g[m];
            }
            if (a.gout) line_fft_partial<NY, -1, 1, EX>(g, tw, tid, lds);
        }
        if (a.gout) {
            __syncthreads();
            transposed_tail<NY, -1, 1, EX>(smem, a.gout + (size_t)b * NY * a.NX + x0, a.NX, 1.f, smem_tail, a.sq);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Adjoint of the rotation gather (K1^T), atomics-free and deterministic: every destination row of
// the volume gradient sums the rotated-frame gradient rows that were gathered from it, through a
// per-angle inverse (CSR) table.                  adjoint of cnn_propagator/util.py:377-402
// One workgroup per destination row: the (angle, source) list is first collected in LDS by one
// thread per batch element, then all threads stream the listed rows with independent loads.
// ---------------------------------------------------------------------------------------------
struct RotAdjArgs {
    const float2* grot;       // [B][S][NX][NY]
    float2* gvol;             // [n_dest][NY]
    const int* off;           // [n_angles][n_dest + 1]
    const int* order;         // [n_angles][S*NX] source rows (z*NX + x) sorted by destination
    const int* angle_of_b;    // [B]
    int B, n_src, n_dest, NY, accumulate;
    float scale;
    int* heavy_count;         // device counter (zeroed before the launch)
    int* heavy_rows;          // [n_dest] destination rows whose source list does not fit a wave's LDS list
    int d0, d1;               // destination rows handled by this launch: [d0, d1)
};

#define BDOF_ROTADJ_MAXLIST 256      // per wave; longer lists (clamped border rows) take the direct path

__device__ __forceinline__ void f4acc(float4& a, const float4 s) { a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w; }

// One WAVE per destination row (4 rows per workgroup, no workgroup barrier): lane b looks up batch element b's CSR
// range, a wave scan places the (batch, source-row) pairs into the wave's LDS list in a fixed order, then the 64 lanes
// stream the listed 4-KB rows with 16-byte loads, several rows in flight.
__global__ __launch_bounds__(256) void k_rot_adjoint(RotAdjArgs a) {
    __shared__ int lists[4][BDOF_ROTADJ_MAXLIST];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* list = lists[wave];
    const int nv = a.NY / 2;                    // float4 = two (delta,beta) pairs
    for (int d = a.d0 + blockIdx.x * 4 + wave; d < a.d1; d += gridDim.x * 4) {
        int total = 0;
        for (int b0 = 0; b0 < a.B; b0 += 64) {
            const int b = b0 + lane;
            int e0 = 0, cnt = 0;
            const int* order = a.order;
            if (b < a.B) {
                const int ang = a.angle_of_b[b];
                const int* off = a.off + (size_t)ang * (a.n_dest + 1);
                e0 = off[d];
                cnt = off[d + 1] - e0;
                order = a.order + (size_t)ang * a.n_src;
            }
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            const int chunk_total = __shfl(incl, 63, 64);
            if (total + chunk_total <= BDOF_ROTADJ_MAXLIST) {
                int pos = total + incl - cnt;
                for (int e = 0; e < cnt; ++e) list[pos++] = b * a.n_src + order[e0 + e];
            }
            total += chunk_total;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (total > BDOF_ROTADJ_MAXLIST) {
            // clamped border rows collect thousands of sources: one wave would become the kernel's tail.  Defer them to
            // k_rot_adjoint_heavy, which gives each such row a whole workgroup.
            if (lane == 0) a.heavy_rows[atomicAdd(a.heavy_count, 1)] = d;
            continue;
        }
        float4* drow = reinterpret_cast<float4*>(a.gvol + (size_t)d * a.NY);
        for (int v0 = 0; v0 < nv; v0 += 256) {
            float4 acc[4];
            bool ok[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc[c] = make_float4(0.f, 0.f, 0.f, 0.f); ok[c] = v0 + c * 64 + lane < nv; }
            int e = 0;
            for (; e + 2 <= total; e += 2) {
                const float4* r0 = reinterpret_cast<const float4*>(a.grot + (size_t)list[e] * a.NY) + v0 + lane;
                const float4* r1 = reinterpret_cast<const float4*>(a.grot + (size_t)list[e + 1] * a.NY) + v0 + lane;
                float4 s0[4], s1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) if (ok[c]) { s0[c] = r0[c * 64]; s1[c] = r1[c * 64]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) if (ok[c]) { f4acc(acc[c], s0[c]); f4acc(acc[c], s1[c]); }
            }
            for (; e < total; ++e) {
                const float4* r0 = reinterpret_cast<const float4*>(a.grot + (size_t)list[e] * a.NY) + v0 + lane;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (ok[c]) f4acc(acc[c], r0[c * 64]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!ok[c]) continue;
                float4 o = make_float4(acc[c].x * a.scale, acc[c].y * a.scale, acc[c].z * a.scale, acc[c].w * a.scale);
                float4* dst = drow + v0 + c * 64 + lane;
                if (a.accumulate) f4acc(o, *dst);
                *dst = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Heavy destination rows: one workgroup per row; the source indices are staged through LDS 256 at a time and all
// threads stream the rows (16 B per lane, 4 rows in flight).  Fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void k_rot_adjoint_heavy(RotAdjArgs a) {
    __shared__ int stage[256];
    const int nheavy = *a.heavy_count;
    const int nv = a.NY / 2;
    for (int i = blockIdx.x; i < nheavy; i += gridDim.x) {
        const int d = a.heavy_rows[i];
        for (int v0 = 0; v0 < nv; v0 += 256) {
            const int v = v0 + threadIdx.x;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = 0; b < a.B; ++b) {
                const int ang = a.angle_of_b[b];
                const int* off = a.off + (size_t)ang * (a.n_dest + 1);
                const int* order = a.order + (size_t)ang * a.n_src;
                const int e0 = off[d], e1 = off[d + 1];
                for (int c0 = e0; c0 < e1; c0 += 256) {
                    const int n = min(256, e1 - c0);
                    __syncthreads();
                    if ((int)threadIdx.x < n) stage[threadIdx.x] = b * a.n_src + order[c0 + threadIdx.x];
                    __syncthreads();
                    if (v < nv) {
                        int e = 0;
                        for (; e + 4 <= n; e += 4) {
                            const float4 s0 = reinterpret_cast<const float4*>(a.grot + (size_t)stage[e] * a.NY)[v];
                            const float4 s1 = reinterpret_cast<const float4*>(a.grot + (size_t)stage[e + 1] * a.NY)[v];
                            const float4 s2 = reinterpret_cast<const float4*>(a.grot + (size_t)stage[e + 2] * a.NY)[v];
                            const float4 s3 = reinterpret_cast<const float4*>(a.grot + (size_t)stage[e + 3] * a.NY)[v];
                            f4acc(acc, s0); f4acc(acc, s1); f4acc(acc, s2); f4acc(acc, s3);
                        }
                        for (; e < n; ++e) f4acc(acc, reinterpret_cast<const float4*>(a.grot + (size_t)stage[e] * a.NY)[v]);
                    }
                }
            }
            if (v < nv) {
                float4* dst = reinterpret_cast<float4*>(a.gvol + (size_t)d * a.NY) + v;
                float4 o = make_float4(acc.x * a.scale, acc.y * a.scale, acc.z * a.scale, acc.w * a.scale);
                if (a.accumulate) f4acc(o, *dst);
                *dst = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused regulariser gradient + Adam + finite-support mask + non-negativity (K9 + K10).
//   cnn_propagator/fullfield.py:109-118,352-362 ; cnn_propagator/util.py:61-70,280-291
// Volume layout [X][Z][Y] of (delta, beta) pairs.  Reads x_old, writes x_new (the TV stencil
// needs the neighbours' pre-update values).
// ---------------------------------------------------------------------------------------------
struct AdamArgs {
    const float2* x_old;
    float2* x_new;
    const float2* g;       // data-term gradient (already summed over ranks)
    float2* m;
    float2* v;
    const float* mask;     // nullable [X][Z][Y]
    int NXv, NZv, NYv;
    float g_scale;         // 1/size                       cnn_propagator/fullfield.py:351
    float alpha_d, alpha_b, gamma;
    float lr, b1, b2, eps, inv_bc1, inv_bc2;   // inv_bc = 1 / (1 - b^(i_batch+1))
    float om_b1, om_b2;                        // 1 - b1, 1 - b2 formed in float64 on the host (1 - 0.999f is 1.3e-5 off 1e-3)
    int clip;              // max(x, 0)
    int x0, x1;            // slab of the volume updated by this launch: x in [x0, x1)  (the stencil reads beyond it)
};

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

// (Round 3 measured the alternative the round-2 review suggested — one thread per (z, y) column walking x with the x-neighbours
// of the TV stencil in registers: L2 <-> fabric traffic of a 512^3 step 10.2 -> 8.7 GB as intended, kernel time 1.58 -> 1.97 ms:
// every workgroup then reads 2-KB pieces 2 MB apart, and what the stencil's re-fetches cost in bytes the linear walk below wins
// back in DRAM locality.  Kept: this element-per-thread form.)
__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    const size_t slab = (size_t)a.NZv * a.NYv;
    // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin, so workgroup b works in the (b % 8)-th eighth of the
    // range and walks it linearly with its XCD's other workgroups — the z+-1 / x+-1 neighbours of the TV stencil (4 KB and
    // NZ*NY*8 B away) are then lines the same XCD's L2 has just seen, instead of another XCD's.
    const size_t first = (size_t)a.x0 * slab, total = (size_t)(a.x1 - a.x0) * slab;
    const size_t per_xcd = (total + 7) / 8;
    const size_t lane0 = (size_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x, stride = (size_t)(gridDim.x >> 3) * blockDim.x;
    const size_t c0 = (size_t)(blockIdx.x & 7) * per_xcd, c1 = c0 + per_xcd < total ? c0 + per_xcd : total;
    for (size_t loc = c0 + lane0; loc < c1; loc += stride) {
        const size_t idx = first + loc;
        const int y = idx % a.NYv;
        const size_t r = idx / a.NYv;
        const int z = r % a.NZv;
        const int x = r / a.NZv;
        const float2 xv = a.x_old[idx];
        float gd = a.g[idx].x * a.g_scale + a.alpha_d * sgn(xv.x);
        float gb = a.g[idx].y * a.g_scale + a.alpha_b * sgn(xv.y);
        if (a.gamma != 0.f) {
            const size_t sy = 1, sz = a.NYv, sx = (size_t)a.NZv * a.NYv;
            const float ym = a.x_old[idx - y * sy + ((y + a.NYv - 1) % a.NYv) * sy].x;
            const float yp = a.x_old[idx - y * sy + ((y + 1) % a.NYv) * sy].x;
            const float zm = a.x_old[idx - z * sz + ((z + a.NZv - 1) % a.NZv) * sz].x;
            const float zp = a.x_old[idx - z * sz + ((z + 1) % a.NZv) * sz].x;
            const float xm = a.x_old[idx - x * sx + ((x + a.NXv - 1) % a.NXv) * sx].x;
            const float xp = a.x_old[idx - x * sx + ((x + 1) % a.NXv) * sx].x;
            const float c = xv.x;
            gd += a.gamma * (sgn(c - ym) - sgn(yp - c) + sgn(c - zm) - sgn(zp - c) + sgn(c - xm) - sgn(xp - c));
        }
        float2 m = a.m[idx], v = a.v[idx];
        m.x = a.om_b1 * gd + a.b1 * m.x;
        m.y = a.om_b1 * gb + a.b1 * m.y;
        v.x = a.om_b2 * gd * gd + a.b2 * v.x;
        v.y = a.om_b2 * gb * gb + a.b2 * v.y;
        a.m[idx] = m;
        a.v[idx] = v;
        float nd = xv.x - a.lr * (m.x * a.inv_bc1) / (sqrtf(v.x * a.inv_bc2) + a.eps);
        float nb = xv.y - a.lr * (m.y * a.inv_bc1) / (sqrtf(v.y * a.inv_bc2) + a.eps);
        if (a.mask) { const float mk = a.mask[idx]; nd *= mk; nb *= mk; }
        if (a.clip) { nd = fmaxf(nd, 0.f); nb = fmaxf(nb, 0.f); }
        a.x_new[idx] = make_float2(nd, nb);
    }
}

// ---------------------------------------------------------------------------------------------
// Tiled ("pfft") propagation: a field too large for one fused plan is cut into overlapping tiles that run through the
// per-slice kernels as a batch; every few slices the tiles' cores are stitched back and the halos refilled (README.md:1-11
// of the reference: "tiling-based Fresnel multislice propagation"; its source is on a branch that is not in the checkout).
// Field [FX][FY] complex; tile b covers field rows x0[b] .. x0[b] + TX - 1 and columns y0[b] .. + TY - 1, PERIODICALLY
// (the whole-field FFT propagator it stands in for is periodic).
// ---------------------------------------------------------------------------------------------
struct TileArgs {
    cf* field;
    cf* tiles;          // [B][TX][TY]
    const int* x0;
    const int* y0;
    int B, FX, FY, TX, TY, hx, hy;      // scatter: only the core [hx, TX - hx) x [hy, TY - hy) of a tile is written back
    int taper;                          // gather: the outermost `taper` pixels of a tile are ramped to zero (raised cosine)
};
// A tile is propagated with its own periodic FFT: left and right edge meet, and a jump there diffracts into the tile with a
// 1/distance tail (Fresnel edge fringes) — 7e-4 of error at a 16-pixel halo.  Ramping the outer part of the halo to zero
// removes the jump; what is left travels inwards at the geometric rate only (3e-5 at the same halo, 3e-6 at 32 pixels).
__device__ __forceinline__ float taper_weight(int i, int n, int taper) {
    const int e = min(i, n - 1 - i);
    return e < taper ? 0.5f - 0.5f * __cosf(3.14159265358979f * ((float)e + 0.5f) / (float)taper) : 1.f;
}
__device__ __forceinline__ int wrap_idx(int i, int n) { i %= n; return i < 0 ? i + n : i; }

// mode 0: tile = field (periodic) x taper window.  mode 1: tile = field on the tile's CORE, zero on the halo and beyond the
// field's edge — the adjoint of k_tiles_scatter (which writes cores, without wrapping).
__global__ __launch_bounds__(256) void k_tiles_gather(TileArgs a, int mode) {
    const int b = blockIdx.z;
    const int ox = a.x0[b], oy = a.y0[b];
    for (int x = blockIdx.y; x < a.TX; x += gridDim.y) {
        cf* dst = a.tiles + ((size_t)b * a.TX + x) * a.TY;
        if (mode == 1) {
            const int xg = ox + x;
            const bool xin = x >= a.hx && x < a.TX - a.hx && xg >= 0 && xg < a.FX;
            const cf* src = a.field + (size_t)(xin ? xg : 0) * a.FY;
            for (int y = blockIdx.x * blockDim.x + threadIdx.x; y < a.TY; y += gridDim.x * blockDim.x) {
                const int yg = oy + y;
                const bool in = xin && y >= a.hy && y < a.TY - a.hy && yg >= 0 && yg < a.FY;
                dst[y] = in ? src[yg] : make_float2(0.f, 0.f);
            }
            continue;
        }
        const cf* src = a.field + (size_t)wrap_idx(ox + x, a.FX) * a.FY;
        const float wx = taper_weight(x, a.TX, a.taper);
        for (int y = blockIdx.x * blockDim.x + threadIdx.x; y < a.TY; y += gridDim.x * blockDim.x)
            dst[y] = cscale(src[wrap_idx(oy + y, a.FY)], wx * taper_weight(y, a.TY, a.taper));
    }
}
// cores back into the field; a core pixel beyond the field's edge is dropped (cores tile the field from 0, the last ones
// overhang), so every field pixel has exactly one writer
__global__ __launch_bounds__(256) void k_tiles_scatter(TileArgs a) {
    const int b = blockIdx.z;
    const int ox = a.x0[b], oy = a.y0[b];
    for (int x = a.hx + blockIdx.y; x < a.TX - a.hx; x += gridDim.y) {
        const int xg = ox + x;
        if (xg < 0 || xg >= a.FX) continue;
        cf* dst = a.field + (size_t)xg * a.FY;
        const cf* src = a.tiles + ((size_t)b * a.TX + x) * a.TY;
        for (int y = a.hy + blockIdx.x * blockDim.x + threadIdx.x; y < a.TY - a.hy; y += gridDim.x * blockDim.x) {
            const int yg = oy + y;
            if (yg >= 0 && yg < a.FY) dst[yg] = src[y];
        }
    }
}
// Adjoint of the tapered periodic gather: field[xg][yg] = sum over the tiles b and tile pixels (x, y) that were cut from
// (xg, yg) — periodically — of w(x) w(y) tiles[b][x][y].  One workgroup per field row; it first lists the (tile, x) pairs
// that map onto its row, then every thread sums its columns over the list in a fixed order (deterministic, no atomics).
#define BDOF_TILE_MAXLIST 1024
__global__ __launch_bounds__(256) void k_tiles_gather_adjoint(TileArgs a) {
    __shared__ int lb[BDOF_TILE_MAXLIST], lx[BDOF_TILE_MAXLIST];
    __shared__ int nlist;
    for (int xg = blockIdx.x; xg < a.FX; xg += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int n = 0;
            for (int b = 0; b < a.B; ++b) {
                // tile rows x with (x0[b] + x) mod FX == xg
                int x = wrap_idx(xg - a.x0[b], a.FX);
                for (; x < a.TX && n < BDOF_TILE_MAXLIST; x += a.FX) { lb[n] = b; lx[n] = x; ++n; }
            }
            nlist = n;
        }
        __syncthreads();
        const int n = nlist;
        for (int yg = threadIdx.x; yg < a.FY; yg += blockDim.x) {
            float sx = 0.f, sy = 0.f;
            for (int e = 0; e < n; ++e) {
                const int b = lb[e], x = lx[e];
                const float wx = taper_weight(x, a.TX, a.taper);
                for (int y = wrap_idx(yg - a.y0[b], a.FY); y < a.TY; y += a.FY) {
                    const float w = wx * taper_weight(y, a.TY, a.taper);
                    const cf v = a.tiles[((size_t)b * a.TX + x) * a.TY + y];
                    sx = fmaf(w, v.x, sx);
                    sy = fmaf(w, v.y, sy);
                }
            }
            a.field[(size_t)xg * a.FY + yg] = make_float2(sx, sy);
        }
    }
}
// Object gradient of a slice range of the tiled propagation: the window-frame gradient rows grot[b][z - z0][x][y] of every
// tile are added into the volume gradient rows gvol[tab[z][xg]][yg], xg = x0[b] + x, yg = y0[b] + y (the object is not
// periodic: tile pixels beyond the volume saw vacuum and contribute nothing).  One workgroup per volume column xg; a thread
// owns (xg, yg) for all z of the range, so rows shared by several slices (a slab repeated over z) accumulate in a register.
// Requires tab[z][.] to be injective in xg for every z (no rotation), which is what the tiled path runs.
struct TileGradArgs {
    const float2* grot;     // [B][nz][TX][TY]
    float2* gvol;           // [rows][volNY]
    const int* tab;         // [S][volNX] of the (single) angle
    const int* x0;
    const int* y0;
    int B, TX, TY, volNX, volNY, z0, nz, accumulate;
};
__global__ __launch_bounds__(256) void k_tiles_grad_add(TileGradArgs a) {
    __shared__ int lb[BDOF_TILE_MAXLIST];
    __shared__ int nlist;
    for (int xg = blockIdx.x; xg < a.volNX; xg += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int n = 0;
            for (int b = 0; b < a.B && n < BDOF_TILE_MAXLIST; ++b) {
                const int x = xg - a.x0[b];
                if (x >= 0 && x < a.TX) lb[n++] = b;
            }
            nlist = n;
        }
        __syncthreads();
        const int n = nlist;
        for (int yg = threadIdx.x; yg < a.volNY; yg += blockDim.x) {
            // the tiles that cover (xg, yg): found once, outside the slice loop (a handful: (T / core)^2)
            constexpr int MAXM = 9;
            size_t moff[MAXM];
            int nm = 0, spill_from = n;
            for (int e = 0; e < n; ++e) {
                const int b = lb[e];
                const int y = yg - a.y0[b];
                if (y < 0 || y >= a.TY) continue;
                if (nm == MAXM) { spill_from = e; break; }
                moff[nm++] = (((size_t)b * a.nz) * a.TX + (xg - a.x0[b])) * a.TY + y;
            }
            const size_t zstride = (size_t)a.TX * a.TY;
            int cur = -1;
            float2 acc = make_float2(0.f, 0.f);
            for (int zr = 0; zr < a.nz; ++zr) {
                const int dest = a.tab[(size_t)(a.z0 + zr) * a.volNX + xg];
                if (dest != cur) {
                    if (cur >= 0) {
                        float2* d = a.gvol + (size_t)cur * a.volNY + yg;
                        *d = make_float2(d->x + acc.x, d->y + acc.y);
                    }
                    cur = dest;
                    acc = make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int m = 0; m < MAXM; ++m) {
                    if (m < nm) {
                        const float2 g = a.grot[moff[m] + zr * zstride];
                        acc.x += g.x;
                        acc.y += g.y;
                    }
                }
                for (int e = spill_from; e < n; ++e) {          // more than MAXM covering tiles (very large halos)
                    const int b = lb[e];
                    const int y = yg - a.y0[b];
                    if (y < 0 || y >= a.TY) continue;
                    const float2 g = a.grot[(((size_t)b * a.nz + zr) * a.TX + (xg - a.x0[b])) * a.TY + y];
                    acc.x += g.x;
                    acc.y += g.y;
                }
            }
            if (cur >= 0) {
                float2* d = a.gvol + (size_t)cur * a.volNY + yg;
                *d = make_float2(d->x + acc.x, d->y + acc.y);
            }
        }
    }
}

// out[i] (+)= sum over b of src[b][i]: the probe gradient of the minibatch from the per-wavefield G(psi_0)
__global__ __launch_bounds__(256) void k_sum_fields(const cf* __restrict__ src, cf* __restrict__ out, int B, size_t n, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float sx = 0.f, sy = 0.f;
        for (int b = 0; b < B; ++b) { const cf v = src[(size_t)b * n + i]; sx += v.x; sy += v.y; }
        if (accumulate) { sx += out[i].x; sy += out[i].y; }
        out[i] = make_float2(sx, sy);
    }
}

// dst[b][:] = src[idx[b]][:] for B fields of n16 16-byte words: the minibatch's measured amplitudes picked out of the
// resident stack of all angles in one launch (this_prj_batch = prj[this_ind_batch], cnn_propagator/fullfield.py:344).
__global__ __launch_bounds__(256) void k_gather_fields(float4* __restrict__ dst, const float4* __restrict__ src,
                                                        const int* __restrict__ idx, int B, size_t n16) {
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float4* s = src + (size_t)idx[b] * n16;
        float4* d = dst + (size_t)b * n16;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
}

// the same in 4-byte words, for fields whose size is not a multiple of 16 bytes (a 75 x 75 projection after ::ds_level
// downsampling, a 35 x 35 probe on the generic engine): such fields do not start on 16-byte boundaries either
__global__ __launch_bounds__(256) void k_gather_fields4(float* __restrict__ dst, const float* __restrict__ src,
                                                         const int* __restrict__ idx, int B, size_t n4) {
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* s = src + (size_t)idx[b] * n4;
        float* d = dst + (size_t)b * n4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Bilinear rotation of the object (the TF twin's tf_rotate(..., interpolation='BILINEAR'), tensorflow_recon/fullfield.py:96):
// tf.contrib.image.rotate treats the (Y, X, Z, 2) object as NHWC images of height X and width Z and maps every OUTPUT pixel
// (h, w) to the input point (h', w') = (sin w + cos h + y_off, cos w - sin h + x_off), sampled bilinearly with zeros outside.
// In the [X][Z][Y] row layout a rotated row (x, z) is a weighted sum of at most four volume rows — still whole contiguous
// rows.  prm[b] = (cos, sin, x_off, y_off) in float64 (coordinates are formed per ROW, so float64 costs nothing).
// Output rows [b][z][x][y]: the layout of already rotated objects (bdof_set_object without a table).
// ---------------------------------------------------------------------------------------------
struct RotBilinArgs {
    const float2* vol;      // [NXv][NZv][NYv]
    float2* rot;            // forward: out [B][NZv][NXv][NYv] ; adjoint: in, the rotated-frame gradient
    float2* gvol;           // adjoint: out [NXv][NZv][NYv]
    const double4* prm;     // [B]
    int B, NXv, NZv, NYv;
    int d0, d1, accumulate; // adjoint: destination rows [d0, d1)
    float scale;
};
// source point and the four (row index, weight) pairs of output row (h = x, w = z); rows outside the volume get weight 0
__device__ __forceinline__ void bilin_taps(const double4 p, int h, int w, int H, int W, int (&row)[4], float (&wt)[4]) {
    const double sw = p.x * w - p.y * h + p.z, sh = p.y * w + p.x * h + p.w;
    const double fw = floor(sw), fh = floor(sh);
    const int w0 = (int)fw, h0 = (int)fh;
    const float aw = (float)(sw - fw), ah = (float)(sh - fh);
    const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
    const float cw[4] = {(1.f - ah) * (1.f - aw), (1.f - ah) * aw, ah * (1.f - aw), ah * aw};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool in = hs[i] >= 0 && hs[i] < H && ws[i] >= 0 && ws[i] < W;
        row[i] = in ? hs[i] * W + ws[i] : 0;
        wt[i] = in ? cw[i] : 0.f;
    }
}
// one wave per output row (4 rows per workgroup).  MOD: the rows are written as modulation factors c - 1 of the interpolated
// (delta, beta) (what k_modulation_table would make of them in a second pass over B volumes), with the per-workgroup
// float64 sums of c - 1 in `partial` when the mean-refraction carrier needs them.
template <bool MOD>
__global__ __launch_bounds__(256) void k_rot_bilinear(RotBilinArgs a, float k, double2* __restrict__ partial) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nrows = (size_t)a.B * a.NZv * a.NXv;
    const int nv = a.NYv / 2;                      // float4 = two (delta, beta) pairs
    double sx = 0.0, sy = 0.0;
    for (size_t o = (size_t)blockIdx.x * 4 + wave; o < nrows; o += (size_t)gridDim.x * 4) {
        const int x = o % a.NXv;
        const size_t r = o / a.NXv;
        const int z = r % a.NZv, b = r / a.NZv;
        int row[4];
        float wt[4];
        bilin_taps(a.prm[b], x, z, a.NXv, a.NZv, row, wt);
        float4* dst = reinterpret_cast<float4*>(a.rot + o * a.NYv);
        for (int v = lane; v < nv; v += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (wt[i] != 0.f) {
                    const float4 s = reinterpret_cast<const float4*>(a.vol + (size_t)row[i] * a.NYv)[v];
                    acc.x = fmaf(wt[i], s.x, acc.x); acc.y = fmaf(wt[i], s.y, acc.y);
                    acc.z = fmaf(wt[i], s.z, acc.z); acc.w = fmaf(wt[i], s.w, acc.w);
                }
            }
            if constexpr (MOD) {
                const float2 m0 = slice_modulation_m1(make_float2(acc.x, acc.y), k), m1 = slice_modulation_m1(make_float2(acc.z, acc.w), k);
                acc = make_float4(m0.x, m0.y, m1.x, m1.y);
                sx += (double)m0.x + (double)m1.x;
                sy += (double)m0.y + (double)m1.y;
            }
            dst[v] = acc;
        }
    }
    if constexpr (MOD) {
        if (partial) {
            __shared__ double w[2][4];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off, 64); sy += __shfl_down(sy, off, 64); }
            if (lane == 0) { w[0][wave] = sx; w[1][wave] = sy; }
            __syncthreads();
            if (threadIdx.x == 0) partial[blockIdx.x] = make_double2(w[0][0] + w[0][1] + w[0][2] + w[0][3], w[1][0] + w[1][1] + w[1][2] + w[1][3]);
        }
    }
}
// Adjoint, as a gather (deterministic, no atomics): volume row d = (x', z') collects w * rotated-frame rows (x, z) whose
// source point lies within one pixel of it.  A rotation preserves distances, so those (x, z) lie within sqrt(2) of the
// inverse-rotated (x', z'): the 4 x 4 lattice points around it are examined, the weight of each comes from the FORWARD map
// (the same numbers the forward kernel used).
// Per destination row the candidates (angle b, lattice point (jh, jw)) are examined 64 at a time, one per lane (the float64
// tap arithmetic is done once per candidate, not once per lane); the ones with a non-zero weight are compacted, in candidate
// order, into a per-wave list, and the wave then streams those rows — four float4 columns per lane, two rows in flight.
template <int NV4>       // float4 columns per lane: NYv / 2 <= 64 * NV4
__global__ __launch_bounds__(256) void k_rot_bilinear_adjoint(RotBilinArgs a) {
    __shared__ int l_row[4][64];
    __shared__ float l_wgt[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = a.NYv / 2;
    const int H = a.NXv, W = a.NZv;
    for (int d = a.d0 + blockIdx.x * 4 + wave; d < a.d1; d += gridDim.x * 4) {
        const int hp = d / W, wp = d - hp * W;            // (x', z')
        float4 acc[NV4];
#pragma unroll
        for (int i = 0; i < NV4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b0 = 0; b0 < a.B; b0 += 4) {
            const int b = b0 + (lane >> 4), jh = (lane >> 2) & 3, jw = lane & 3;
            float wgt = 0.f;
            int row = 0;
            if (b < a.B) {
                const double4 p = a.prm[b];
                // inverse of (h', w') = (s w + c h + yo, c w - s h + xo):  w = c (w' - xo) + s (h' - yo),  h = -s (w' - xo) + c (h' - yo)
                const double dw = wp - p.z, dh = hp - p.w;
                const double wi = p.x * dw + p.y * dh, hi = -p.y * dw + p.x * dh;
                const int w = (int)floor(wi) - 1 + jw, h = (int)floor(hi) - 1 + jh;
                if (h >= 0 && h < H && w >= 0 && w < W) {
                    const double sw = p.x * w - p.y * h + p.z, sh = p.y * w + p.x * h + p.w;
                    const double fw = floor(sw), fh = floor(sh);
                    const int iw = wp - (int)fw, ih = hp - (int)fh;         // which of the 2 x 2 taps of (h, w) is (h', w'): 0 or 1
                    if (iw >= 0 && iw <= 1 && ih >= 0 && ih <= 1) {
                        const float aw = (float)(sw - fw), ah = (float)(sh - fh);
                        wgt = (ih ? ah : 1.f - ah) * (iw ? aw : 1.f - aw);
                        row = (b * W + w) * H + h;
                    }
                }
            }
            const unsigned long long live = __ballot(wgt != 0.f);
            const int n = __popcll(live);
            if (wgt != 0.f) {
                const int pos = __popcll(live & ((1ull << lane) - 1ull));      // candidate order is kept: deterministic sums
                l_row[wave][pos] = row;
                l_wgt[wave][pos] = wgt;
            }
            // a wave's LDS accesses execute in program order: no hardware barrier between the list's writes and reads
            __builtin_amdgcn_wave_barrier();
            int e = 0;
            for (; e + 1 < n; e += 2) {
                const float w0 = l_wgt[wave][e], w1 = l_wgt[wave][e + 1];
                const float4* r0 = reinterpret_cast<const float4*>(a.rot + (size_t)l_row[wave][e] * a.NYv);
                const float4* r1 = reinterpret_cast<const float4*>(a.rot + (size_t)l_row[wave][e + 1] * a.NYv);
                float4 s0[NV4], s1[NV4];
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const int v = min(lane + 64 * i, nv - 1);
                    s0[i] = r0[v];
                    s1[i] = r1[v];
                }
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    acc[i].x = fmaf(w0, s0[i].x, acc[i].x); acc[i].y = fmaf(w0, s0[i].y, acc[i].y);
                    acc[i].z = fmaf(w0, s0[i].z, acc[i].z); acc[i].w = fmaf(w0, s0[i].w, acc[i].w);
                    acc[i].x = fmaf(w1, s1[i].x, acc[i].x); acc[i].y = fmaf(w1, s1[i].y, acc[i].y);
                    acc[i].z = fmaf(w1, s1[i].z, acc[i].z); acc[i].w = fmaf(w1, s1[i].w, acc[i].w);
                }
            }
            if (e < n) {
                const float w0 = l_wgt[wave][e];
                const float4* r0 = reinterpret_cast<const float4*>(a.rot + (size_t)l_row[wave][e] * a.NYv);
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const float4 s = r0[min(lane + 64 * i, nv - 1)];
                    acc[i].x = fmaf(w0, s.x, acc[i].x); acc[i].y = fmaf(w0, s.y, acc[i].y);
                    acc[i].z = fmaf(w0, s.z, acc[i].z); acc[i].w = fmaf(w0, s.w, acc[i].w);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        float4* drow = reinterpret_cast<float4*>(a.gvol + (size_t)d * a.NYv);
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                float4 o = make_float4(acc[i].x * a.scale, acc[i].y * a.scale, acc[i].z * a.scale, acc[i].w * a.scale);
                if (a.accumulate) f4acc(o, drow[v]);
                drow[v] = o;
            }
        }
    }
}

// Value of the regulariser  alpha_d sum|delta| + alpha_b sum|beta| + gamma TV(delta)  (cnn_propagator/fullfield.py:109-118,
// total_variation_3d util.py:61-70: periodic, anisotropic, sum over the three axes of |roll(x, 1) - x|).  Per-workgroup float64
// partials [3 * gridDim.x]: sum|delta|, sum|beta|, TV.
__global__ __launch_bounds__(256) void k_reg_value(const float2* __restrict__ x, int NXv, int NZv, int NYv, double* __restrict__ partial) {
    const size_t n = (size_t)NXv * NZv * NYv;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % NYv;
        const size_t r = idx / NYv;
        const int z = r % NZv;
        const int xx = r / NZv;
        const float2 v = x[idx];
        s0 += fabsf(v.x);
        s1 += fabsf(v.y);
        const size_t sz = NYv, sx = (size_t)NZv * NYv;
        const float ym = x[idx - y + (y + NYv - 1) % NYv].x;
        const float zm = x[idx - z * sz + ((z + NZv - 1) % NZv) * sz].x;
        const float xm = x[idx - xx * sx + ((xx + NXv - 1) % NXv) * sx].x;
        s2 += (double)fabsf(ym - v.x) + (double)fabsf(zm - v.x) + (double)fabsf(xm - v.x);
    }
    __shared__ double w[3][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if ((threadIdx.x & 63) == 0) { w[0][threadIdx.x >> 6] = s0; w[1][threadIdx.x >> 6] = s1; w[2][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) partial[3 * blockIdx.x + threadIdx.x] = w[threadIdx.x][0] + w[threadIdx.x][1] + w[threadIdx.x][2] + w[threadIdx.x][3];
}

// shrink-wrap of the finite-support mask: mask *= (delta > thresh)      cnn_propagator/fullfield.py:365-368
__global__ __launch_bounds__(256) void k_mask_shrink(const float2* x, float* mask, size_t n, float thresh) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
        mask[idx] = x[idx].x > thresh ? mask[idx] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// Ptychography: adjoint of "rotate, zero-pad, cut a window per probe position" (K1^T . K11^T).
//   cnn_propagator/ptychography.py:32-34,42-73.  Destination row d of the volume gradient collects, for every rotated
// row (z, xg) gathered from it and every window b that covers xg, the window-frame gradient row shifted by yoff[b].
// All batch elements share one rotation angle.  Deterministic (no atomics).
// ---------------------------------------------------------------------------------------------
struct WinAdjArgs {
    const float2* grot;       // [B][S][NX][NY]  window-frame gradient
    float2* gvol;             // [n_dest][volNY]
    const int* off;           // [n_dest + 1] of the batch's angle
    const int* order;         // [S*volNX] source rows (z*volNX + xg) sorted by destination
    const int* xoff;          // [B] window origin in x (may be negative / beyond the volume: zero padding)
    const int* yoff;          // [B]
    int B, S, NX, NY, volNX, volNY, n_dest, accumulate;
    float scale;
};

// Stage 1: overlap-add of the window-frame gradients into the rotated frame, pad[z][xg][y] = sum over the windows b that
// cover (xg, y) of grot[b][z][xg - xoff[b]][y - yoff[b]]  (adjoint of the window cut + zero padding; pixels of a window
// that lie outside the volume are dropped).  One workgroup per (xg, chunk of z): it first collects the windows covering
// its xg in LDS (a scan over B offsets), then streams their rows.  Fixed summation order (by yoff, b) -> deterministic.
// Stage 2 is the plain rotation adjoint (k_rot_adjoint) of `pad` as a one-element batch.
#define BDOF_WIN_MAXLIST 1024
__global__ __launch_bounds__(256) void k_window_overlap_add(WinAdjArgs a, float2* pad, int z_per_wg) {
    __shared__ int ub[BDOF_WIN_MAXLIST], uy[BDOF_WIN_MAXLIST];                              // unsorted: window, yoff
    __shared__ int lb[BDOF_WIN_MAXLIST], ly[BDOF_WIN_MAXLIST], lx[BDOF_WIN_MAXLIST];        // sorted by (yoff, window)
    __shared__ int nlist;
    const int xg = blockIdx.x;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int b = 0; b < a.B; ++b) {
            const int xw = xg - a.xoff[b];
            if (xw >= 0 && xw < a.NX && n < BDOF_WIN_MAXLIST) { ub[n] = b; uy[n] = a.yoff[b]; ++n; }
        }
        nlist = n;
    }
    __syncthreads();
    const int n = nlist;
    // rank sort by (yoff, window index): the windows that cover a given y are then one contiguous run of the list, so the
    // inner loop below has no per-element test and its loads can be issued back to back
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int ye = uy[e], be = ub[e];
        int rank = 0;
        for (int o = 0; o < n; ++o) rank += (uy[o] < ye) || (uy[o] == ye && ub[o] < be);
        lb[rank] = be;
        ly[rank] = ye;
        lx[rank] = xg - a.xoff[be];
    }
    __syncthreads();
    const int z0 = blockIdx.y * z_per_wg, z1 = min(a.S, z0 + z_per_wg);
    for (int y = threadIdx.x; y < a.volNY; y += blockDim.x) {
        // windows with yoff in (y - NY, y]: first index with yoff > y - NY, first index with yoff > y
        int lo, hi;
        {
            int l = 0, h = n;
            while (l < h) { const int mid = (l + h) >> 1; if (ly[mid] > y - a.NY) h = mid; else l = mid + 1; }
            lo = l;
            l = 0; h = n;
            while (l < h) { const int mid = (l + h) >> 1; if (ly[mid] > y) h = mid; else l = mid + 1; }
            hi = l;
        }
        for (int z = z0; z < z1; ++z) {
            float2 acc = make_float2(0.f, 0.f);
            int e = lo;
            for (; e + 4 <= hi; e += 4) {
                float2 g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    g[q] = a.grot[(((size_t)lb[e + q] * a.S + z) * a.NX + lx[e + q]) * a.NY + (y - ly[e + q])];
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc.x += g[q].x; acc.y += g[q].y; }
            }
            for (; e < hi; ++e) {
                const float2 g = a.grot[(((size_t)lb[e] * a.S + z) * a.NX + lx[e]) * a.NY + (y - ly[e])];
                acc.x += g.x;
                acc.y += g.y;
            }
            pad[((size_t)z * a.volNX + xg) * a.volNY + y] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void k_window_rot_adjoint(WinAdjArgs a) {
    for (int d = blockIdx.x; d < a.n_dest; d += gridDim.x) {
        const int e0 = a.off[d], e1 = a.off[d + 1];
        for (int y = threadIdx.x; y < a.volNY; y += blockDim.x) {
            float2 acc = make_float2(0.f, 0.f);
            for (int e = e0; e < e1; ++e) {
                const int src = a.order[e];
                const int z = src / a.volNX, xg = src - z * a.volNX;
                for (int b = 0; b < a.B; ++b) {
                    const int xw = xg - a.xoff[b];
                    if (xw < 0 || xw >= a.NX) continue;
                    const int yw = y - a.yoff[b];
                    if (yw < 0 || yw >= a.NY) continue;
                    const float2 g = a.grot[(((size_t)b * a.S + z) * a.NX + xw) * a.NY + yw];
                    acc.x += g.x;
                    acc.y += g.y;
                }
            }
            float2* dst = a.gvol + (size_t)d * a.volNY + y;
            float2 o = make_float2(acc.x * a.scale, acc.y * a.scale);
            if (a.accumulate) { o.x += dst->x; o.y += dst->y; }
            *dst = o;
        }
    }
}

// =============================================================================================
// Real-space truncated-kernel propagator (cnn_propagator/propagation.py:18-133):
//   per slice  phi = c psi ; psi' = K * pad(phi, edge_val)   (true 2-D convolution, K = E ky (x) kx separable,
//   ks x ks taps, the padding constant follows edge_val' = sum(K) edge_val), after the last slice the wave is
//   renormalised by the corner pixel of batch element 0 (propagation.py:79,109-110).
// Fields are real-space [b][x][y] with the same carrier splitting as the FFT path: psi = a_z + eps, a_{z+1} = a_z sum(K),
// so the padding constant of eps is (1 - a_0) sum(K)^z — zero for a unit plane wave.
// One launch per slice: 2-D LDS tile with halo, y pass then x pass, then the next slice's modulation is applied to
// the tile's outputs (no halo overhead) — 24 B/px forward; the stored phi_z double as the tape.
// =============================================================================================
#define BDOF_CONV_MAXK 33
#define BDOF_CONV_TX 32
#define BDOF_CONV_TY 64

struct ConvTaps {
    float2 ky[BDOF_CONV_MAXK];
    float2 kx[BDOF_CONV_MAXK];
    float2 e;          // global factor of the kernel
    int ks;
};

struct ConvArgs {
    const cf* in;        // [B][NX][NY]  forward: phi_z (eps part) ; backward: G(psi_{z+1})
    cf* out;             // forward: phi_{z+1} (or psi_S after the last slice) ; backward: G(psi_z)
    const cf* tape;      // backward: phi_z
    float2* grot;        // backward: gradient rows [B][S][NX][NY]
    ObjView obj;         // modulation table rows (c - 1) of slice `zmod`
    int B, NX, NY, zmod; // forward: zmod = z + 1 (or -1: no modulation) ; backward: zmod = z
    cf pad;              // forward: padding constant of the eps field (0 in backward)
    cf carrier;          // forward: a_{z+1} ; backward: a_z
    float k;
    const ConvTaps* taps;   // device copy (bdof_set_conv): read through the scalar cache one pass at a time — by value in the
    int ks;                 // kernel arguments all 134 dwords are fetched at entry and live in spilled SGPRs (v_readlane per use)
    const cf* pfield;       // nullable [NX][NY]: carrier FIELD plane replacing `carrier` (bdof_set_conv_probe_stack): forward the
                            // plane of slice zmod = z + 1, backward that of slice z
};

// 1-D pass over a register window: thread owns R consecutive outputs and the R + 2H inputs they need, so every LDS
// value is read once per R outputs instead of once per tap.
typedef float conv_v2f __attribute__((ext_vector_type(2)));
template <bool BWD, int H, int R>
__device__ __forceinline__ void conv_window(const cf (&win)[R + 2 * H], const __attribute__((address_space(4))) float2* taps, cf (&out)[R]) {
    conv_v2f acc[R];
#pragma unroll
    for (int o = 0; o < R; ++o) acc[o] = (conv_v2f){0.f, 0.f};
#pragma unroll
    for (int d = -H; d <= H; ++d) {
        const cf w = make_float2(taps[H + d].x, taps[H + d].y);
        // complex multiply-accumulate as two packed FMAs (v_pk_fma_f32: 2 x the rate of v_fma_f32): the splat of f.x / f.y,
        // the swap of the tap's halves and the signs all fold into op_sel / neg modifiers, the tap sits in an SGPR pair.
        // backward: multiply by conj(w).  Same operation order as four scalar FMAs, so the results are bit-identical.
        const conv_v2f w1 = BWD ? (conv_v2f){w.x, -w.y} : (conv_v2f){w.x, w.y};
        const conv_v2f w2 = BWD ? (conv_v2f){w.y, w.x} : (conv_v2f){-w.y, w.x};
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const cf f = win[o + H + (BWD ? d : -d)];
            acc[o] = __builtin_elementwise_fma((conv_v2f){f.x, f.x}, w1, acc[o]);
            acc[o] = __builtin_elementwise_fma((conv_v2f){f.y, f.y}, w2, acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) out[o] = make_float2(acc[o].x, acc[o].y);
}

// workgroup barrier that orders LDS traffic only (see res_sync in bdof_resident.h)
__device__ __forceinline__ void conv_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// H = (ks - 1) / 2 as a template parameter (2, 4, 8, 16 instantiated; H = 0 selects the generic runtime-ks loops)
#define BDOF_CONV_THREADS 512
// PF: the carrier is a FIELD plane (a.pfield) instead of the scalar a.carrier — an instance of its own, so that the scalar
// instances keep the register allocation they were tuned with (as a run-time select the extra pointer cost 10 us per launch)
template <bool BWD, int H, bool PF = false>
__global__ __launch_bounds__(BDOF_CONV_THREADS, 4) void k_conv(ConvArgs a) {
    // R outputs per thread in the y pass (384 windows per tile), R2 in the x pass (512 windows: every thread busy)
    constexpr int TX = BDOF_CONV_TX, TY = BDOF_CONV_TY, R = 8, R2 = 4;
    const int h = H > 0 ? H : (a.ks - 1) / 2;
    const int TXH = TX + 2 * h, TYH = TY + 2 * h;
    typedef const __attribute__((address_space(4))) ConvTaps* TapsPtr;      // constant address space: s_load, no VGPRs
    // odd row strides: pass 1 runs with consecutive lanes on consecutive ROWS (each lane slides its own window along y),
    // so the row stride in 8-byte slots must be odd for those lanes to hit distinct LDS banks
    const int SA = TYH | 1, SM = TY + 1;
    extern __shared__ cf lds[];
    cf* A = lds;                       // [TXH][SA]
    cf* M = lds + TXH * SA;            // [TXH][SM]
    const int tiles_x = a.NX / TX, tiles_y = a.NY / TY;
    const int ntiles = a.B * tiles_x * tiles_y;
    // Software pipeline over the tiles of this workgroup: the halo tile of the NEXT tile is fetched into registers while
    // the x pass of the current one runs (A is free once the y pass is done), and the loads the epilogue depends on
    // (rotation-table row -> modulation row, tape) are issued at the top of the tile, two barriers before their use.
    // Barriers order LDS only (conv_sync): __syncthreads() would also drain those loads.
    constexpr int NLD = (BDOF_CONV_MAXK - 1 + BDOF_CONV_TX) * (BDOF_CONV_MAXK - 1 + BDOF_CONV_TY) / BDOF_CONV_THREADS + 1;
    constexpr int NPF = H > 0 ? ((TX + 2 * H) * (TY + 2 * H) + BDOF_CONV_THREADS - 1) / BDOF_CONV_THREADS : NLD;
    cf nx[NPF];
    // per-thread offsets stay 32-bit (one wavefield is < 2^31 elements); the wavefield's base is a scalar: 64-bit
    // multiplies (v_mad_u64_u32) are quarter rate and this kernel is bound by VALU issue
    // returns the mask of the elements that lie inside the field (the others take the padding constant at stash time,
    // so that the loads themselves are unconditional and all in flight together)
    auto fetch = [&](int tile, cf (&v)[NPF]) -> unsigned {
        const int b = tile / (tiles_x * tiles_y);
        const int t2 = tile - b * tiles_x * tiles_y;
        const int x0 = (t2 / tiles_y) * TX, y0 = (t2 % tiles_y) * TY;
        const cf* src = a.in + (size_t)b * a.NX * a.NY;
        unsigned inside = 0;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = min((int)threadIdx.x + q * BDOF_CONV_THREADS, TXH * TYH - 1);
            const int i = e / TYH, j = e - i * TYH;
            const int x = x0 - h + i, y = y0 - h + j;
            inside |= ((unsigned)x < (unsigned)a.NX && (unsigned)y < (unsigned)a.NY) ? 1u << q : 0u;
            v[q] = src[__umul24(min(max(x, 0), a.NX - 1), a.NY) + min(max(y, 0), a.NY - 1)];      // NX, NY <= 4096: 24-bit multiply
        }
        return inside;
    };
    auto stash = [&](const cf (&v)[NPF], unsigned inside) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = (int)threadIdx.x + q * BDOF_CONV_THREADS;
            if (e < TXH * TYH) {
                const int i = e / TYH, j = e - i * TYH;
                A[i * SA + j] = (inside >> q) & 1u ? v[q] : a.pad;
            }
        }
    };
    unsigned nx_in = 0;
    if ((int)blockIdx.x < ntiles) {
        nx_in = fetch(blockIdx.x, nx);
        stash(nx, nx_in);
    }
    conv_sync();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y);
        const int t2 = tile - b * tiles_x * tiles_y;
        const int x0 = (t2 / tiles_y) * TX, y0 = (t2 % tiles_y) * TY;
        // this thread's x-pass item (one per thread: (TX / R2) * TY = 512) and the loads its epilogue needs
        const int i0 = ((int)threadIdx.x / TY) * R2, j = (int)threadIdx.x % TY;
        const int y = y0 + j;
        const int yg = y + (a.obj.yoff ? a.obj.yoff[b] : 0);
        const int yc = min(max(yg, 0), a.obj.volNY - 1);
        // modulation-table row of each output row (obj_src_row, with the per-wavefield part as a scalar base)
        int srow[R2];
        {
            const int xg0 = x0 + i0 + (a.obj.tab && a.obj.xoff ? a.obj.xoff[b] : 0);
            const int* tabrow = a.obj.tab ? a.obj.tab + ((long long)a.obj.angle_of_b[b] * a.obj.S + max(a.zmod, 0)) * a.obj.volNX : nullptr;
            const int base = a.obj.tab ? 0 : (b * a.obj.S + max(a.zmod, 0)) * a.NX;
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                const int xg = xg0 + q;
                if (a.zmod < 0) srow[q] = -1;
                else if (tabrow) srow[q] = (unsigned)xg < (unsigned)a.obj.volNX ? tabrow[xg] : -1;
                else srow[q] = base + xg;
            }
        }
        // the taps are re-read per pass (scalar cache hits): laundering the pointer keeps the loads from being hoisted out
        // of the tile loop, where 68 SGPRs of taps would not fit beside the rest
        TapsPtr kt = (TapsPtr)a.taps;
        asm volatile("" : "+s"(kt));
        // pass along y.  forward: o[y] = sum_d K[h+d] f[y-d] ; backward: o[y] = sum_d conj(K[h+d]) g[y+d]
        if constexpr (H > 0) {
            for (int t = threadIdx.x; t < TXH * (TY / R); t += blockDim.x) {
                const int i = t % TXH, j0 = (t / TXH) * R;
                cf win[R + 2 * H], o[R];
#pragma unroll
                for (int q = 0; q < R + 2 * H; ++q) win[q] = A[i * SA + j0 + q];
                conv_window<BWD, H, R>(win, kt->ky, o);
#pragma unroll
                for (int q = 0; q < R; ++q) M[i * SM + j0 + q] = o[q];
            }
        } else {
            for (int e = threadIdx.x; e < TXH * TY; e += blockDim.x) {
                const int i = e / TY, jj = e - i * TY;
                cf acc = make_float2(0.f, 0.f);
                for (int d = -h; d <= h; ++d) {
                    const cf f = A[i * SA + jj + h + (BWD ? d : -d)];
                    const cf w = make_float2(kt->ky[h + d].x, kt->ky[h + d].y);
                    acc = cadd(acc, BWD ? cmulc(f, w) : cmul(f, w));
                }
                M[i * SM + jj] = acc;
            }
        }
        float2 m1[R2];
        cf tp[R2];
        const cf* tape_b = BWD ? a.tape + (size_t)b * a.NX * a.NY : nullptr;
#pragma unroll
        for (int q = 0; q < R2; ++q) {
            m1[q] = a.obj.vol[(size_t)max(srow[q], 0) * a.obj.volNY + yc];
            if constexpr (BWD) tp[q] = tape_b[__umul24(x0 + i0 + q, a.NY) + y];
        }
        conv_sync();
        const int next = tile + gridDim.x;
        if (next < ntiles) nx_in = fetch(next, nx);  // in flight during the x pass
        // pass along x (window of R2 consecutive x for one y), then the pointwise physics
        asm volatile("" : "+s"(kt));
        {
            cf o[R2];
            const cf ke = make_float2(kt->e.x, kt->e.y);
            cf* out_b = a.out + (size_t)b * a.NX * a.NY;
            float2* grot_b = BWD ? a.grot + ((size_t)b * a.obj.S + a.zmod) * a.NX * a.NY : nullptr;
            if constexpr (H > 0) {
                cf win[R2 + 2 * H];
#pragma unroll
                for (int q = 0; q < R2 + 2 * H; ++q) win[q] = M[(i0 + q) * SM + j];
                conv_window<BWD, H, R2>(win, kt->kx, o);
            } else {
                for (int q = 0; q < R2; ++q) {
                    cf acc = make_float2(0.f, 0.f);
                    for (int d = -h; d <= h; ++d) {
                        const cf f = M[(i0 + q + h + (BWD ? d : -d)) * SM + j];
                        const cf w = make_float2(kt->kx[h + d].x, kt->kx[h + d].y);
                        acc = cadd(acc, BWD ? cmulc(f, w) : cmul(f, w));
                    }
                    o[q] = acc;
                }
            }
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                const cf acc = BWD ? cmulc(o[q], ke) : cmul(o[q], ke);
                const int x = x0 + i0 + q;
                const unsigned off = __umul24(x, a.NY) + y;
                const bool in = srow[q] >= 0 && yg == yc;
                const float2 mm = make_float2(in ? m1[q].x : 0.f, in ? m1[q].y : 0.f);
                cf car = a.carrier;
                if constexpr (PF) car = a.pfield[off];                    // L2-resident plane shared by all wavefields
                if constexpr (!BWD) {
                    out_b[off] = modulate_eps(acc, car, mm);                // phi_{z+1} = c_{z+1} psi_{z+1}  (eps part)
                } else {
                    const cf phi = cadd(tp[q], car);
                    const cf tt = cmulc(acc, phi);                              // G(phi) conj(phi)
                    grot_b[off] = make_float2(a.k * tt.y, -a.k * tt.x);
                    out_b[off] = cmulc(acc, make_float2(1.f + mm.x, mm.y));     // G(psi_z) = conj(c_z) G(phi_z)
                }
            }
        }
        if (next < ntiles) stash(nx, nx_in);         // A was last read before the barrier above
        conv_sync();
    }
}

// phi_0 = c_0 (a_0 + eps_probe): the first modulation, before any convolution
struct ConvInitArgs {
    const cf* probe;     // [NX][NY] eps part
    cf* out;             // [B][NX][NY]
    ObjView obj;
    int B, NX, NY;
    cf carrier;          // a_0
    const cf* pfield;    // nullable [NX][NY]: carrier field p_0 (then `probe` holds zeros)
};
__global__ __launch_bounds__(256) void k_conv_init(ConvInitArgs a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % a.NY;
        const size_t r = idx / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        float2 m1 = make_float2(0.f, 0.f);
        const long long srow = obj_src_row(a.obj, b, x, 0, a.NX);
        const int yg = y + (a.obj.yoff ? a.obj.yoff[b] : 0);
        if (srow >= 0 && yg >= 0 && yg < a.obj.volNY) m1 = a.obj.vol[(size_t)srow * a.obj.volNY + yg];
        a.out[idx] = modulate_eps(a.probe[(size_t)x * a.NY + y], a.pfield ? a.pfield[(size_t)x * a.NY + y] : a.carrier, m1);
    }
}

// s = probe[0,0] / psi_S[0,0,0]   (propagation.py:79,109-110).  scal[0] = s, scal[1] = psi_S[0,0,0]
__global__ void k_conv_scalars(const cf* psi_eps, cf carrier_end, const cf* probe_eps, cf carrier0, cf* scal) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const cf p000 = cadd(psi_eps[0], carrier_end);
        const cf init = cadd(probe_eps[0], carrier0);
        const float den = p000.x * p000.x + p000.y * p000.y;
        scal[0] = cscale(cmulc(init, p000), 1.f / den);
        scal[1] = p000;
    }
}

// Renormalised wave q = s (a_S + eps) and, for the real-space detector without free propagation, loss + seed.
//   MODE 0: q only (out) ; MODE 1: q, loss partials, Gp = conj(s) G(d) (out2)
struct ConvFinalArgs {
    const cf* psi_eps;
    cf* out;             // q  [B][NX][NY] (nullable in MODE 1)
    cf* out2;            // MODE 1: conj(s) G(d)
    const float* meas;   // MODE 1
    double* partial;     // MODE 1: [2 * gridDim.x]: sum r^2, sum r |d|
    const cf* scal;
    cf carrier_end;      // a_S; zero when `split` (only the scattered part e' = s eps is formed)
    size_t n;
    float seed_scale;
    // MODE 1 with residual splitting (bdof_set_meas_mode(1)): d = A + e', A = s a_S from the host in float64, meas = m - |a_0|,
    // dref = |A| - |a_0|
    int split;
    cf A;
    float absA, dref;
    const cf* pfield;    // nullable [plane]: carrier FIELD p_S added to every wavefield (MODE 0 outputs of the full wave)
    size_t plane;
};
template <int MODE>
__global__ __launch_bounds__(256) void k_conv_final(ConvFinalArgs a) {
    const cf s = a.scal[0];
    double acc = 0.0, acc2 = 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.n; idx += (size_t)gridDim.x * blockDim.x) {
        cf full = cadd(a.psi_eps[idx], a.carrier_end);
        if (a.pfield) full = cadd(full, a.pfield[idx % a.plane]);
        const cf q = cmul(full, s);
        if constexpr (MODE == 1) {
            if (a.split) {
                if (a.out) a.out[idx] = cadd(q, a.A);
                a.out2[idx] = cmulc(loss_seed_dev(q, a.A, a.absA, a.meas[idx], a.seed_scale, acc, acc2, a.dref), s);
                continue;
            }
        }
        if (a.out) a.out[idx] = q;
        if constexpr (MODE == 1) {
            const float ab = sqrtf(q.x * q.x + q.y * q.y);
            const float r = ab - a.meas[idx];
            acc += (double)r * r;
            acc2 += (double)r * ab;
            const float f = ab > 0.f ? a.seed_scale * r / ab : 0.f;
            a.out2[idx] = cmulc(cscale(q, f), s);                         // conj(s) G(d)
        }
    }
    if constexpr (MODE == 1) {
        __shared__ double w1[4], w2[4];
        acc = wave_reduce_sum(acc);
        acc2 = wave_reduce_sum(acc2);
        if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = acc; w2[threadIdx.x >> 6] = acc2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            a.partial[2 * blockIdx.x] = w1[0] + w1[1] + w1[2] + w1[3];
            a.partial[2 * blockIdx.x + 1] = w2[0] + w2[1] + w2[2] + w2[3];
        }
    }
}

// Gp = conj(s) G(q) for detectors that went through the FFT machinery (G(q) arrives in real space)
__global__ __launch_bounds__(256) void k_conv_scale_seed(cf* g, const cf* scal, size_t n) {
    const cf s = scal[0];
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
        g[idx] = cmulc(g[idx], s);
}

// loss = sum r^2 / n ; corner fix  Gp[0] -= (sum_j G(q)_j conj(q_j)) / conj(psi_S[0,0,0]),  the sum being
// seed_scale * sum r |d| (G(d) is parallel to d and the detector step is unitary / an exact adjoint pair)
__global__ void k_conv_finish(const double* partial, int nblocks, int stride, double inv_n, double seed_scale, double* loss,
                              cf* gp, const cf* scal) {
    double v = 0.0, v2 = 0.0;
    for (int j = threadIdx.x; j < nblocks; j += blockDim.x) { v += partial[stride * j]; v2 += partial[stride * j + 1]; }
    __shared__ double w1[4], w2[4];
    v = wave_reduce_sum(v);
    v2 = wave_reduce_sum(v2);
    if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = v; w2[threadIdx.x >> 6] = v2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        loss[0] = (w1[0] + w1[1] + w1[2] + w1[3]) * inv_n;
        const double rsum = (w2[0] + w2[1] + w2[2] + w2[3]) * seed_scale;
        const cf p = scal[1];
        const double den = (double)p.x * p.x + (double)p.y * p.y;
        // rsum / conj(p) = rsum * p / |p|^2
        gp[0].x -= (float)(rsum * p.x / den);
        gp[0].y -= (float)(rsum * p.y / den);
    }
}

// real-space rows -> hybrid (FFT along y), transposed into L2: feeds the detector steps of the FFT machinery
struct RealToHybArgs {
    const cf* in;      // [B][NX][NY] real space
    cf* out;           // L2
    int B, NX;
    const cf* twiddle;
};
template <int NY>
__global__ __launch_bounds__(BDOF_THREADS, RowCfg<NY>::MIN_WAVES) void k_row_real_to_hyb(RealToHybArgs a) {
    typedef RowCfg<NY> C;
    __shared__ cf smem[C::LDS_CF];
    const int tid = threadIdx.x % C::T, rl = threadIdx.x / C::T;
    constexpr bool EX = BDOF_EX_DET;
    __shared__ cf smem_tw[FftTw<NY>::LDS_CNT * (EX ? 2 : 1)];
    FftTw<NY> tw;
    __shared__ cf smem_tail[7 * C::T * (EX ? 2 : 1)];
    tw.template load<EX>(a.twiddle, tid, smem_tw, smem_tail);
    const int ntiles = a.B * a.NX / C::TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C::TILE;
        const int b = row0 / a.NX, x0 = row0 - b * a.NX;
#pragma nounroll
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int r = pass * C::RPP + rl;
            RowLds<C::T> lds{smem + r * C::RS};
            cf u[8];
            const cf* src = a.in + (size_t)(row0 + r) * NY;
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = src[tid + m * C::T];
            line_fft_partial<NY, -1, 1, EX>(u, tw, tid, lds);
        }
        __syncthreads();
        transposed_tail<NY, -1, 1, EX>(smem, a.out + (size_t)b * NY * a.NX + x0, a.NX, 1.f, smem_tail);
        __syncthreads();
    }
}
