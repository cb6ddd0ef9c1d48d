// Whole-field / tile-batch operations of the tiled ("pfft") propagator in either precision (DESIGN: cfg4).
//
//  * free-space steps of a batch of fields in ONE transform pair (rocFFT, batched 2-D, in place): field <- F^-1 ( h * F field )
//    with a caller-supplied table h[kx][ky] (the n-th power of the transfer function, 1 / (NX NY) folded in) — the long-range
//    correction of the tiled propagator applies it once per stitch range to the whole field and to the tile batch;
//  * the float64 tile path (TiledPropagator(precision='float64')): the reference's arithmetic is float64
//    (cnn_propagator/np_funcs.py:20-42, quirk Q2) and a 1024-slice stack run through float32 transforms carries 1.5e-5 of
//    rounding; here modulation, transforms (rocFFT double) and the transfer-function product are float64, unfused;
//  * tile cut-out / write-back and axpy in float64.
#pragma once
#include "bdof_generic.h"

// f[i] *= h[i mod per_field]     (C2 = float2 / double2)
template <class C2>
__global__ __launch_bounds__(256) void k_f_hmul(C2* __restrict__ f, const C2* __restrict__ h, size_t per_field, size_t n, int conj_h) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const C2 a = f[i];
        C2 t = h[i % per_field];
        if (conj_h) t.y = -t.y;
        C2 o;
        o.x = a.x * t.x - a.y * t.y;
        o.y = a.x * t.y + a.y * t.x;
        f[i] = o;
    }
}

// y += alpha x on n real numbers
template <class R>
__global__ __launch_bounds__(256) void k_f_axpy(R* __restrict__ y, const R* __restrict__ x, R alpha, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void k_f_to_double(const cf* __restrict__ src, double2* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_double2((double)src[i].x, (double)src[i].y);
}
__global__ __launch_bounds__(256) void k_f_to_float(const double2* __restrict__ src, cf* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_float2((float)src[i].x, (float)src[i].y);
}

// ---- tiles in float64: the same cut-out / write-back as k_tiles_gather (mode 0) / k_tiles_scatter (bdof_kernels.h) ----------
struct Tile64Args {
    double2* field;     // [FX][FY]
    double2* tiles;     // [B][TX][TY]
    const int* x0;
    const int* y0;
    int B, FX, FY, TX, TY, hx, hy, taper;
};
__device__ __forceinline__ double taper_weight64(int i, int n, int taper) {
    const int e = min(i, n - 1 - i);
    return e < taper ? 0.5 - 0.5 * cos(3.14159265358979323846 * ((double)e + 0.5) / (double)taper) : 1.0;
}
__global__ __launch_bounds__(256) void k_tiles_gather64(Tile64Args a) {
    const int b = blockIdx.z;
    const int ox = a.x0[b], oy = a.y0[b];
    for (int x = blockIdx.y; x < a.TX; x += gridDim.y) {
        double2* dst = a.tiles + ((size_t)b * a.TX + x) * a.TY;
        const double2* src = a.field + (size_t)wrap_idx(ox + x, a.FX) * a.FY;
        const double wx = taper_weight64(x, a.TX, a.taper);
        for (int y = blockIdx.x * blockDim.x + threadIdx.x; y < a.TY; y += gridDim.x * blockDim.x) {
            const double w = wx * taper_weight64(y, a.TY, a.taper);
            const double2 v = src[wrap_idx(oy + y, a.FY)];
            dst[y] = make_double2(v.x * w, v.y * w);
        }
    }
}
__global__ __launch_bounds__(256) void k_tiles_scatter64(Tile64Args a) {
    const int b = blockIdx.z;
    const int ox = a.x0[b], oy = a.y0[b];
    for (int x = a.hx + blockIdx.y; x < a.TX - a.hx; x += gridDim.y) {
        const int xg = ox + x;
        if (xg < 0 || xg >= a.FX) continue;
        double2* dst = a.field + (size_t)xg * a.FY;
        const double2* src = a.tiles + ((size_t)b * a.TX + x) * a.TY;
        for (int y = a.hy + blockIdx.x * blockDim.x + threadIdx.x; y < a.TY - a.hy; y += gridDim.x * blockDim.x) {
            const int yg = oy + y;
            if (yg >= 0 && yg < a.FY) dst[yg] = src[y];
        }
    }
}

// float32 tiles of a float64 field (the field of the float32 tiled path with the long-range correction stays in float64: its
// bulk is carried by the whole-field free-space step in double, the tiles add the object's part)
struct TileMixArgs {
    double2* field;       // [FX][FY]
    const cf* ta;         // [B][TX][TY]
    const cf* tb;         // nullable
    cf* tiles;            // gather target
    const int* x0;
    const int* y0;
    int B, FX, FY, TX, TY, hx, hy, taper, accumulate;
};
// mode 0: tile = field (periodic) x taper window.  mode 1: tile = the field on the tile's CORE, zero on the halo and beyond the
// field's edge — the adjoint of the write-back (k_tiles_gather's two modes, from a float64 field)
__global__ __launch_bounds__(256) void k_tiles_gather_mixed(TileMixArgs a, int mode) {
    const int b = blockIdx.z;
    const int ox = a.x0[b], oy = a.y0[b];
    for (int x = blockIdx.y; x < a.TX; x += gridDim.y) {
        cf* dst = a.tiles + ((size_t)b * a.TX + x) * a.TY;
        if (mode == 1) {
            const int xg = ox + x;
            const bool xin = x >= a.hx && x < a.TX - a.hx && xg >= 0 && xg < a.FX;
            const double2* srow = a.field + (size_t)(xin ? xg : 0) * a.FY;
            for (int y = blockIdx.x * blockDim.x + threadIdx.x; y < a.TY; y += gridDim.x * blockDim.x) {
                const int yg = oy + y;
                const bool in = xin && y >= a.hy && y < a.TY - a.hy && yg >= 0 && yg < a.FY;
                const double2 v = srow[in ? yg : 0];
                dst[y] = in ? make_float2((float)v.x, (float)v.y) : make_float2(0.f, 0.f);
            }
            continue;
        }
        const double2* src = a.field + (size_t)wrap_idx(ox + x, a.FX) * a.FY;
        const double wx = taper_weight64(x, a.TX, a.taper);
        for (int y = blockIdx.x * blockDim.x + threadIdx.x; y < a.TY; y += gridDim.x * blockDim.x) {
            const double w = wx * taper_weight64(y, a.TY, a.taper);
            const double2 v = src[wrap_idx(oy + y, a.FY)];
            dst[y] = make_float2((float)(v.x * w), (float)(v.y * w));
        }
    }
}
// field[core] = (accumulate ? field[core] : 0) + ta - tb   on the cores of the tiles, in float64
__global__ __launch_bounds__(256) void k_tiles_scatter_diff64(TileMixArgs a) {
    const int b = blockIdx.z;
    const int ox = a.x0[b], oy = a.y0[b];
    for (int x = a.hx + blockIdx.y; x < a.TX - a.hx; x += gridDim.y) {
        const int xg = ox + x;
        if (xg < 0 || xg >= a.FX) continue;
        double2* dst = a.field + (size_t)xg * a.FY;
        const size_t row = ((size_t)b * a.TX + x) * a.TY;
        for (int y = a.hy + blockIdx.x * blockDim.x + threadIdx.x; y < a.TY - a.hy; y += gridDim.x * blockDim.x) {
            const int yg = oy + y;
            if (yg < 0 || yg >= a.FY) continue;
            const cf va = a.ta[row + y];
            double dx = (double)va.x, dy = (double)va.y;
            if (a.tb) { const cf vb = a.tb[row + y]; dx -= (double)vb.x; dy -= (double)vb.y; }
            if (a.accumulate) { const double2 o = dst[yg]; dx += o.x; dy += o.y; }
            dst[yg] = make_double2(dx, dy);
        }
    }
}

// Adjoint of the tapered periodic cut-out, accumulated into a float64 field: field[xg][yg] += sum over the tiles b and tile
// pixels (x, y) cut from (xg, yg) — periodically — of w(x) w(y) (ta - tb)[b][x][y].  One workgroup per field row, fixed order.
__global__ __launch_bounds__(256) void k_tiles_gather_adjoint_diff64(TileMixArgs a) {
    __shared__ int lb[BDOF_TILE_MAXLIST], lx[BDOF_TILE_MAXLIST];
    __shared__ int nlist;
    for (int xg = blockIdx.x; xg < a.FX; xg += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int n = 0;
            for (int b = 0; b < a.B; ++b) {
                int x = wrap_idx(xg - a.x0[b], a.FX);
                for (; x < a.TX && n < BDOF_TILE_MAXLIST; x += a.FX) { lb[n] = b; lx[n] = x; ++n; }
            }
            nlist = n;
        }
        __syncthreads();
        const int n = nlist;
        for (int yg = threadIdx.x; yg < a.FY; yg += blockDim.x) {
            double sx = 0.0, sy = 0.0;
            for (int e = 0; e < n; ++e) {
                const int b = lb[e], x = lx[e];
                const double wx = taper_weight64(x, a.TX, a.taper);
                for (int y = wrap_idx(yg - a.y0[b], a.FY); y < a.TY; y += a.FY) {
                    const double w = wx * taper_weight64(y, a.TY, a.taper);
                    const size_t o = ((size_t)b * a.TX + x) * a.TY + y;
                    const cf va = a.ta[o];
                    double dx = (double)va.x, dy = (double)va.y;
                    if (a.tb) { const cf vb = a.tb[o]; dx -= (double)vb.x; dy -= (double)vb.y; }
                    sx += w * dx;
                    sy += w * dy;
                }
            }
            double2* dst = a.field + (size_t)xg * a.FY + yg;
            if (a.accumulate) { sx += dst->x; sy += dst->y; }
            *dst = make_double2(sx, sy);
        }
    }
}

// phi = c psi, c = exp(i k delta) exp(-k beta) from the caller's (delta, beta) rows (cnn_propagator/np_funcs.py:37-40), float64
struct Mod64Args {
    double2* field;      // [B][NX][NY]
    ObjView obj;         // .vol = the (delta, beta) rows themselves (not the float32 table of c - 1)
    int B, NX, NY, z;
    double k;
    double2* tape;       // nullable: phi_z is stored here too (the float64 loss + gradient paths keep every slice's)
};
__global__ __launch_bounds__(256) void k_f64_modulate(Mod64Args a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % a.NY;
        const size_t r = idx / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        const float2 db = g_mod_value(a.obj, b, x, y, a.z, a.NX);
        if (db.x == 0.f && db.y == 0.f) {                           // vacuum: c = 1
            if (a.tape) a.tape[idx] = a.field[idx];
            continue;
        }
        double s, cs;
        sincos(a.k * (double)db.x, &s, &cs);
        const double e = exp(-a.k * (double)db.y);
        const double2 v = a.field[idx];
        const double2 phi = make_double2(e * (v.x * cs - v.y * s), e * (v.x * s + v.y * cs));
        a.field[idx] = phi;
        if (a.tape) a.tape[idx] = phi;
    }
}

// D float32 copies of a float64 complex table whose roundings average to the float64 values (the dithered transform constants
// of bdof_fft.h, applied to the transfer function h): in copy d each part of entry i is rounded DOWN or UP — up in a fraction
// p = (x - lo) / (hi - lo) of the copies, spread evenly over d with a golden-ratio phase per entry — so that the mean over any
// run of L copies is x to ulp / L.  The launches of slice z take copy z mod D: a fixed float32 table is the same small
// perturbation of every slice, and its error adds up coherently (1.4e-5 of the exit wave after 1024 slices, measured with
// everything else in float64); the dithered copies' errors cancel (1e-6).
__device__ __forceinline__ float dither_round(double x, int d, double phase) {
    float lo = (float)x;
    if ((double)lo > x) lo = nextafterf(lo, -INFINITY);
    const float hi = nextafterf(lo, INFINITY);
    if ((double)lo == x) return lo;
    const double p = (x - (double)lo) / ((double)hi - (double)lo);
    return floor((d + 1) * p + phase) > floor(d * p + phase) ? hi : lo;
}
__global__ __launch_bounds__(256) void k_dither_copies(const double2* __restrict__ src, cf* __restrict__ dst, size_t n, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = src[i];
        const double g = (double)(i % 1048573) * 0.6180339887498949;
        const double ph = g - floor(g), ph2 = ph + 0.5 - floor(ph + 0.5);
        for (int d = 0; d < D; ++d) dst[(size_t)d * n + i] = make_float2(dither_round(v.x, d, ph), dither_round(v.y, d, ph2));
    }
}

// Carrier stack of a stitch range from the spectrum of the tiles' input: S_z = s_hat * H^z for z = 1 .. nz - 1, all of them in one
// pass (running product in double); h carries 1 / (NX NY), H = h * (NX NY).  spec [B][per] -> out [nz - 1][B][per]
__global__ __launch_bounds__(256) void k_carrier_spectra(const double2* __restrict__ spec, const double2* __restrict__ h, double2* __restrict__ out,
                                                         size_t per, int B, int nz, double scale_up) {
    const size_t n = per * B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 hk = h[i % per];
        const double hx = hk.x * scale_up, hy = hk.y * scale_up;            // the un-normalised H
        const double2 s0 = spec[i];
        double wx = hk.x, wy = hk.y;                                        // H^z / (NX NY), z = 1
        for (int z = 1; z < nz; ++z) {
            out[(size_t)(z - 1) * n + i] = make_double2(s0.x * wx - s0.y * wy, s0.x * wy + s0.y * wx);
            const double tx = wx * hx - wy * hy, ty = wx * hy + wy * hx;
            wx = tx; wy = ty;
        }
    }
}
