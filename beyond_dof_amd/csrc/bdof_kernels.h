// Multislice hot-path kernels (gfx950).  See DESIGN.md for the data layout and the byte model.
//
// Layout: a wavefield is psi[b][x][y], y fastest ("rows" run along y, the tomographic rotation
// axis, so that a rotated object row is a contiguous row of the un-rotated volume).
// "Hybrid" fields are transformed along y only:  psi_hat = R psi / NY  (R = unnormalised DFT along y).
//   row kernel  A_i : psi = R^-1' psi_hat ; phi = c_i psi ; out = R phi          (24 B/px)
//   col kernel  B   : out = C^-1' ( hs * C in ),  hs = ifftshift(H)^T / (NX NY)   (16 B/px)
// so one slice is two launches and 40 B/px; B's output is directly the tape entry psi_hat_{i+1}.
#pragma once
#include "bdof_fft.h"

#define BDOF_ROW_THREADS 256

// ---------------------------------------------------------------------------------------------
// LDS images
// ---------------------------------------------------------------------------------------------
template <int T> struct RowLds {
    cf* base;   // this row's image, padded: slot(i) = i + (i >> 4)
    __device__ __forceinline__ cf ld(int i) const { return base[i + (i >> 4)]; }
    __device__ __forceinline__ void st(int i, cf v) { base[i + (i >> 4)] = v; }
    __device__ __forceinline__ void sync_w2r() { sync(); }
    __device__ __forceinline__ void sync_r2w() { sync(); }
    __device__ __forceinline__ void sync() {
        if constexpr (T <= 64) {
            // the line lives inside one wave: LDS ops of a wave execute in order; only the compiler
            // must be kept from reordering across the exchange
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    }
};

template <int W> struct ColLds {
    cf* base;   // tile image [idx][w], w fastest: conflict-free for lanes that differ in w
    int w;
    __device__ __forceinline__ cf ld(int i) const { return base[i * W + w]; }
    __device__ __forceinline__ void st(int i, cf v) { base[i * W + w] = v; }
    __device__ __forceinline__ void sync_w2r() { __syncthreads(); }
    __device__ __forceinline__ void sync_r2w() { __syncthreads(); }
};

template <int N> struct ColTile {   // columns per workgroup tile
    static constexpr int W = N <= 256 ? 32 : (N == 512 ? 16 : 8);
};

// ---------------------------------------------------------------------------------------------
// Object access: which (delta,beta) row feeds wavefield row (b, x) at slice z
// ---------------------------------------------------------------------------------------------
struct ObjView {
    const float2* vol;       // rows of volNY (delta, beta) pairs
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

// c = exp(i k delta) * exp(-k beta)                      cnn_propagator/np_funcs.py:39
__device__ __forceinline__ cf slice_modulation(float2 db, float k) {
    float s, c;
    sincosf(k * db.x, &s, &c);
    float e = expf(-k * db.y);
    return make_float2(e * c, e * s);
}

// ---------------------------------------------------------------------------------------------
// A: forward row kernel.                              cnn_propagator/np_funcs.py:37-40 (+ FFT along y)
// ---------------------------------------------------------------------------------------------
struct RowFwdArgs {
    const cf* in;      // psi_hat_i [B][NX][NY]; ignored when FIRST (probe used)
    const cf* probe;   // [NX][NY] real-space probe
    cf* out;           // R phi_i
    ObjView obj;
    int B, NX, z;
    float k;
    const cf* twiddle;
};

template <int NY, bool FIRST>
__global__ __launch_bounds__(BDOF_ROW_THREADS) void k_row_fwd(RowFwdArgs a) {
    constexpr int T = NY / 8, RPW = BDOF_ROW_THREADS / T, NPAD = NY + NY / 16;
    __shared__ cf smem[RPW * NPAD];
    const int tid = threadIdx.x % T, rl = threadIdx.x / T;
    RowLds<T> lds{smem + rl * NPAD};
    FftTw<NY> tw;
    tw.load(a.twiddle, tid);
    const int nrows = a.B * a.NX;
    for (int rg = blockIdx.x; rg * RPW < nrows; rg += gridDim.x) {
        const int row = rg * RPW + rl;
        const bool valid = row < nrows;
        const int b = valid ? row / a.NX : 0;
        const int x = valid ? row - b * a.NX : 0;
        cf u[8];
        if constexpr (FIRST) {
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = a.probe[(size_t)x * NY + tid + m * T];
        } else {
            const cf* src = a.in + (size_t)(valid ? row : 0) * NY;
#pragma unroll
            for (int m = 0; m < 8; ++m) u[m] = src[tid + m * T];
            line_fft<NY, +1>(u, tw, tid, lds);
        }
        const long long srow = obj_src_row(a.obj, b, x, a.z, a.NX);
        const int y0 = a.obj.yoff ? a.obj.yoff[b] : 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int yg = tid + m * T + y0;
            float2 db = make_float2(0.f, 0.f);
            if (srow >= 0 && yg >= 0 && yg < a.obj.volNY) db = a.obj.vol[(size_t)srow * a.obj.volNY + yg];
            u[m] = cmul(u[m], slice_modulation(db, a.k));
        }
        line_fft<NY, -1>(u, tw, tid, lds);
        if (valid) {
            cf* dst = a.out + (size_t)row * NY;
#pragma unroll
            for (int m = 0; m < 8; ++m) dst[tid + m * T] = u[m];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// B: column kernel, Fresnel transfer-function step.   cnn_propagator/np_funcs.py:42 (K3-K5)
// ---------------------------------------------------------------------------------------------
struct ColPropArgs {
    const cf* in;
    cf* out;
    const cf* h;     // hs[kx][ky] = ifftshift(H)[ky][kx] / (NX NY)
    int B, NY;
    float scale;     // extra factor (NY when the input is already a normalised hybrid field)
    int conj_h;      // adjoint step uses conj(h)
    const cf* twiddle;
};

template <int NX>
__global__ __launch_bounds__((NX / 8) * ColTile<NX>::W) void k_col_prop(ColPropArgs a) {
    constexpr int T = NX / 8, W = ColTile<NX>::W;
    __shared__ cf smem[NX * W];
    const int w = threadIdx.x % W, i = threadIdx.x / W;
    ColLds<W> lds{smem, w};
    FftTw<NX> tw;
    tw.load(a.twiddle, i);
    const int tiles_per_b = a.NY / W;
    const int ntiles = a.B * tiles_per_b;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int y = (tile - b * tiles_per_b) * W + w;
        const size_t base = (size_t)b * NX * a.NY + y;
        cf u[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = a.in[base + (size_t)(i + m * T) * a.NY];
        line_fft<NX, -1>(u, tw, i, lds);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            cf hv = a.h[(size_t)(i + m * T) * a.NY + y];
            if (a.conj_h) hv.y = -hv.y;
            u[m] = cmul(u[m], cscale(hv, a.scale));
        }
        line_fft<NX, +1>(u, tw, i, lds);
#pragma unroll
        for (int m = 0; m < 8; ++m) a.out[base + (size_t)(i + m * T) * a.NY] = u[m];
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

template <int NTHREADS> __device__ __forceinline__ void block_store_sum(double v, double* dst) {
    __shared__ double wsum[NTHREADS / 64];
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wsum[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int j = 0; j < NTHREADS / 64; ++j) s += wsum[j];
        *dst = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Detector plane in real space (free_prop_cm None / distance): magnitude loss + adjoint seed.
//   loss: cnn_propagator/fullfield.py:106 ; seed G(d) = 2 (|d| - |m|) d/|d| / n   
// ---------------------------------------------------------------------------------------------
struct LossArgs {
    const cf* in;        // hybrid field of the detector wave
    cf* out_hyb;         // nullable: R seed * out_scale
    cf* out_wave;        // nullable: detector wave d[b][x][y]
    const float* meas;   // nullable: |measured| [b][x][y] (already in this kernel's index order)
    double* partial;     // [gridDim.x] per-workgroup sums of (|d|-|m|)^2
    int B, NX, NY;
    float in_scale, out_scale, seed_scale;
    const cf* twiddle;
};

__device__ __forceinline__ cf loss_seed(cf d, float m, float seed_scale, double& acc) {
    const float a = sqrtf(d.x * d.x + d.y * d.y);
    const float r = a - m;
    acc += (double)r * (double)r;
    const float f = a > 0.f ? seed_scale * r / a : 0.f;
    return make_float2(d.x * f, d.y * f);
}

template <int NY>
__global__ __launch_bounds__(BDOF_ROW_THREADS) void k_row_loss(LossArgs a) {
    constexpr int T = NY / 8, RPW = BDOF_ROW_THREADS / T, NPAD = NY + NY / 16;
    __shared__ cf smem[RPW * NPAD];
    const int tid = threadIdx.x % T, rl = threadIdx.x / T;
    RowLds<T> lds{smem + rl * NPAD};
    FftTw<NY> tw;
    tw.load(a.twiddle, tid);
    const int nrows = a.B * a.NX;
    double acc = 0.0;
    for (int rg = blockIdx.x; rg * RPW < nrows; rg += gridDim.x) {
        const int row = rg * RPW + rl;
        const bool valid = row < nrows;
        const size_t off = (size_t)(valid ? row : 0) * NY;
        cf u[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = a.in[off + tid + m * T];
        line_fft<NY, +1>(u, tw, tid, lds);
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = cscale(u[m], a.in_scale);
        if (a.out_wave && valid) {
#pragma unroll
            for (int m = 0; m < 8; ++m) a.out_wave[off + tid + m * T] = u[m];
        }
        if (a.meas) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                double dummy = 0.0;
                u[m] = loss_seed(u[m], a.meas[off + tid + m * T], a.seed_scale, valid ? acc : dummy);
            }
            if (a.out_hyb) {
                line_fft<NY, -1>(u, tw, tid, lds);
                if (valid) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) a.out_hyb[off + tid + m * T] = cscale(u[m], a.out_scale);
                }
            }
        }
    }
    if (a.meas) block_store_sum<BDOF_ROW_THREADS>(acc, a.partial + blockIdx.x);
}

// Far-field detector (free_prop_cm == 'inf'): d = fftshift(fft2 psi); here un-shifted, the shift is
// folded into the order in which the host lays out `meas` / reads `out_wave`.
//   cnn_propagator/np_funcs.py:47-48, cnn_propagator/ptychography.py:74-79
template <int NX>
__global__ __launch_bounds__((NX / 8) * ColTile<NX>::W) void k_col_loss_far(LossArgs a) {
    constexpr int T = NX / 8, W = ColTile<NX>::W;
    __shared__ cf smem[NX * W];
    const int w = threadIdx.x % W, i = threadIdx.x / W;
    ColLds<W> lds{smem, w};
    FftTw<NX> tw;
    tw.load(a.twiddle, i);
    const int tiles_per_b = a.NY / W;
    const int ntiles = a.B * tiles_per_b;
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int y = (tile - b * tiles_per_b) * W + w;
        const size_t base = (size_t)b * NX * a.NY + y;
        cf u[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = cscale(a.in[base + (size_t)(i + m * T) * a.NY], a.in_scale);
        line_fft<NX, -1>(u, tw, i, lds);
        if (a.out_wave) {
#pragma unroll
            for (int m = 0; m < 8; ++m) a.out_wave[base + (size_t)(i + m * T) * a.NY] = u[m];
        }
        if (a.meas) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
                u[m] = loss_seed(u[m], a.meas[base + (size_t)(i + m * T) * a.NY], a.seed_scale, acc);
            if (a.out_hyb) {
                line_fft<NX, +1>(u, tw, i, lds);
#pragma unroll
                for (int m = 0; m < 8; ++m) a.out_hyb[base + (size_t)(i + m * T) * a.NY] = cscale(u[m], a.out_scale);
            }
        }
    }
    if (a.meas) block_store_sum<(NX / 8) * ColTile<NX>::W>(acc, a.partial + blockIdx.x);
}

__global__ void k_sum_partials(const double* partial, int n, double scale, double* out) {
    double v = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) v += partial[j];
    __shared__ double wsum[4];
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale;
}

// ---------------------------------------------------------------------------------------------
// A': backward row kernel (hand-derived adjoint of A):
//   G(phi) = R^-1' g_hat ; psi = R^-1' tape ; phi = c psi ; t = conj(phi) G(phi)
//   g_delta = k Im t ; g_beta = -k Re t ; G(psi) = conj(c) G(phi) ; out = R G(psi)
// ---------------------------------------------------------------------------------------------
struct RowBwdArgs {
    const cf* gin;     // g_hat(phi_i) [B][NX][NY]
    const cf* tape;    // psi_hat_i; ignored when FIRST (probe)
    const cf* probe;
    cf* gout;          // nullable: R G(psi_i)
    float2* grot;      // [B][S][NX][NY] (g_delta, g_beta) in the rotated / windowed frame
    ObjView obj;
    int B, NX, z;
    float k;
    const cf* twiddle;
};

template <int NY, bool FIRST>
__global__ __launch_bounds__(BDOF_ROW_THREADS) void k_row_bwd(RowBwdArgs a) {
    constexpr int T = NY / 8, RPW = BDOF_ROW_THREADS / T, NPAD = NY + NY / 16;
    __shared__ cf smem[RPW * NPAD];
    const int tid = threadIdx.x % T, rl = threadIdx.x / T;
    RowLds<T> lds{smem + rl * NPAD};
    FftTw<NY> tw;
    tw.load(a.twiddle, tid);
    const int nrows = a.B * a.NX;
    for (int rg = blockIdx.x; rg * RPW < nrows; rg += gridDim.x) {
        const int row = rg * RPW + rl;
        const bool valid = row < nrows;
        const int b = valid ? row / a.NX : 0;
        const int x = valid ? row - b * a.NX : 0;
        const size_t off = (size_t)(valid ? row : 0) * NY;
        cf g[8], p[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) g[m] = a.gin[off + tid + m * T];
        if constexpr (FIRST) {
#pragma unroll
            for (int m = 0; m < 8; ++m) p[m] = a.probe[(size_t)x * NY + tid + m * T];
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) p[m] = a.tape[off + tid + m * T];
        }
        line_fft<NY, +1>(g, tw, tid, lds);
        if constexpr (!FIRST) line_fft<NY, +1>(p, tw, tid, lds);
        const long long srow = obj_src_row(a.obj, b, x, a.z, a.NX);
        const int y0 = a.obj.yoff ? a.obj.yoff[b] : 0;
        float2* gdst = a.grot + (((size_t)b * a.obj.S + a.z) * a.NX + x) * NY;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int yg = tid + m * T + y0;
            float2 db = make_float2(0.f, 0.f);
            if (srow >= 0 && yg >= 0 && yg < a.obj.volNY) db = a.obj.vol[(size_t)srow * a.obj.volNY + yg];
            const cf c = slice_modulation(db, a.k);
            const cf phi = cmul(p[m], c);
            const cf t = cmulc(g[m], phi);          // G * conj(phi)
            if (valid) gdst[tid + m * T] = make_float2(a.k * t.y, -a.k * t.x);
            g[m] = cmulc(g[m], c);                  // conj(c) G
        }
        if (a.gout) {
            line_fft<NY, -1>(g, tw, tid, lds);
            if (valid) {
#pragma unroll
                for (int m = 0; m < 8; ++m) a.gout[off + tid + m * T] = g[m];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Adjoint of the rotation gather (K1^T), atomics-free and deterministic: every destination row of
// the volume gradient sums the rotated-frame gradient rows that were gathered from it, through a
// per-angle inverse (CSR) table.                  adjoint of cnn_propagator/util.py:377-402
// ---------------------------------------------------------------------------------------------
struct RotAdjArgs {
    const float2* grot;       // [B][S][NX][NY]
    float2* gvol;             // [n_dest][NY]
    const int* off;           // [n_angles][n_dest + 1]
    const int* order;         // [n_angles][S*NX] source rows (z*NX + x) sorted by destination
    const int* angle_of_b;    // [B]
    int B, n_src, n_dest, NY, accumulate;
    float scale;
};

__global__ __launch_bounds__(256) void k_rot_adjoint(RotAdjArgs a) {
    const int nv = a.NY / 2;   // float4 = two (delta,beta) pairs
    for (int d = blockIdx.x; d < a.n_dest; d += gridDim.x) {
        for (int v = threadIdx.x; v < nv; v += blockDim.x) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = 0; b < a.B; ++b) {
                const int ang = a.angle_of_b[b];
                const int* off = a.off + (size_t)ang * (a.n_dest + 1);
                const int e0 = off[d], e1 = off[d + 1];
                const int* order = a.order + (size_t)ang * a.n_src;
                for (int e = e0; e < e1; ++e) {
                    const float4 s = reinterpret_cast<const float4*>(a.grot + ((size_t)b * a.n_src + order[e]) * a.NY)[v];
                    acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
                }
            }
            float4* dst = reinterpret_cast<float4*>(a.gvol + (size_t)d * a.NY) + v;
            float4 o = make_float4(acc.x * a.scale, acc.y * a.scale, acc.z * a.scale, acc.w * a.scale);
            if (a.accumulate) { const float4 q = *dst; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
            *dst = o;
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
    int clip;              // max(x, 0)
};

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    const size_t n = (size_t)a.NXv * a.NZv * a.NYv;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
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
        m.x = (1.f - a.b1) * gd + a.b1 * m.x;
        m.y = (1.f - a.b1) * gb + a.b1 * m.y;
        v.x = (1.f - a.b2) * gd * gd + a.b2 * v.x;
        v.y = (1.f - a.b2) * gb * gb + a.b2 * v.y;
        a.m[idx] = m;
        a.v[idx] = v;
        float nd = xv.x - a.lr * (m.x * a.inv_bc1) / (sqrtf(v.x * a.inv_bc2) + a.eps);
        float nb = xv.y - a.lr * (m.y * a.inv_bc1) / (sqrtf(v.y * a.inv_bc2) + a.eps);
        if (a.mask) { const float mk = a.mask[idx]; nd *= mk; nb *= mk; }
        if (a.clip) { nd = fmaxf(nd, 0.f); nb = fmaxf(nb, 0.f); }
        a.x_new[idx] = make_float2(nd, nb);
    }
}
