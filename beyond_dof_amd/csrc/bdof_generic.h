// Generic-size engine: any NY x NX (e.g. the 72 x 72 ptychography probe of cnn_propagator/reconstruct_ptycho.py:106).
// The 2-D transforms are rocFFT's (batched, in place); everything else is the same physics as the fused engine written as
// point-wise kernels on real-space fields [b][x][y] (frequency domain [b][kx][ky]).  Unfused, so it moves ~2.5x the bytes
// of the fused row kernels — it exists for coverage of sizes without a hand-written plan and as an on-device cross-check.
#pragma once
#include "bdof_kernels.h"

struct GModArgs {
    cf* field;           // [B][NX][NY] eps part, updated in place
    const cf* probe;     // nullable: [NX][NY] eps part of the probe (first slice: field is written, not read)
    cf* tape;            // nullable: phi_z (eps part)
    ObjView obj;
    int B, NX, NY, z;
    cf carrier;
    const cf* pz;        // nullable: carrier field of the slice, [x][y] (bdof_set_probe_stack); the tape then holds the full phi
    cf cshift;           // a_z (cbar - 1), scalar carrier only (modulate_eps_s)
};

__device__ __forceinline__ float2 g_mod_value(const ObjView& o, int b, int x, int y, int z, int NX) {
    const long long srow = obj_src_row(o, b, x, z, NX);
    const int yg = y + (o.yoff ? o.yoff[b] : 0);
    const int yc = min(max(yg, 0), o.volNY - 1);
    const float2 v = o.vol[(size_t)(srow >= 0 ? srow : 0) * o.volNY + yc];
    const bool in = srow >= 0 && yg == yc;
    return make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
}

__global__ __launch_bounds__(256) void k_g_modulate(GModArgs a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % a.NY;
        const size_t r = idx / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        const cf e = a.probe ? a.probe[(size_t)x * a.NY + y] : a.field[idx];
        const cf pc = a.pz ? a.pz[(size_t)x * a.NY + y] : a.carrier;
        const cf phi = modulate_eps_s(e, pc, g_mod_value(a.obj, b, x, y, a.z, a.NX), a.cshift);
        a.field[idx] = phi;
        if (a.tape) a.tape[idx] = a.pz ? cadd(phi, pc) : phi;
    }
}

// field[b][kx][ky] *= h[ky][kx] (the transfer-function table is kept in the fused engine's [ky][kx] order)
__global__ __launch_bounds__(256) void k_g_hmul(cf* field, const cf* h, int B, int NX, int NY, int conj_h) {
    const size_t n = (size_t)B * NX * NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int ky = idx % NY;
        const int kx = (idx / NY) % NX;
        cf t = h[(size_t)ky * NX + kx];
        if (conj_h) t.y = -t.y;
        field[idx] = cmul(field[idx], t);
    }
}

struct GLossArgs {
    cf* field;           // detector wave (eps part / far: un-shifted fft2), overwritten by the seed when meas != null
    cf* out_wave;        // nullable; real detectors [b][x][y], far field [b][ky][kx]
    const float* meas;   // nullable; same order as out_wave
    double* partial;     // [2 * gridDim.x]
    int B, NX, NY, far;
    cf carrier;          // real detectors: added everywhere; far: added to the DC bin of every batch element
    float seed_scale;
    const cf* pdet;      // nullable: carrier field at the detector, [x][y] / far field [kx][ky] (replaces `carrier`)
    int meas_dev;        // `meas` holds m - |carrier| (loss_seed_dev, bdof_kernels.h)
    float dref;
    double2* gcar;       // nullable [B]: adjoint carrier (AdjCarrier, bdof_kernels.h) — far field with a plane-wave carrier
    double2* gt0;
    double2 carrier_dd, a_end;
    const double2* pdet64;   // nullable: `pdet` in float64 (loss_seed_f64)
    double2* seed64;         // nullable [B][NX][NY]: float64 adjoint sweep (bdof_configure flag 64) — detector wave, residual and
    double meas_ref;         // seed all formed in float64 and left here un-rounded; meas_ref: what the host subtracted (meas_dev)
    double2 pscale;          // complex factor on pdet64 (the real-space propagator's renormalisation s; 0, 0 means 1)
};

__global__ __launch_bounds__(256) void k_g_loss(GLossArgs a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    double acc = 0.0, acc2 = 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % a.NY;
        const size_t r = idx / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        cf d = a.field[idx];
        if (a.seed64 && a.meas) {
            const bool dc = x == 0 && y == 0;
            double2 p = make_double2(0.0, 0.0);
            if (a.pdet64) p = a.pdet64[(size_t)x * a.NY + y];
            else if (a.pdet) { const cf q = a.pdet[(size_t)x * a.NY + y]; p = make_double2((double)q.x, (double)q.y); }
            else if (!a.far || dc) p = a.carrier_dd;
            const size_t oi = a.far ? ((size_t)b * a.NY + y) * a.NX + x : idx;
            const double dx = p.x + (double)d.x, dy = p.y + (double)d.y;
            const double ab = sqrt(dx * dx + dy * dy), rr = ab - ((double)a.meas[oi] + a.meas_ref);
            acc += rr * rr;
            acc2 += rr * ab;
            const double f = ab > 0.0 ? (double)a.seed_scale * rr / ab : 0.0;
            a.seed64[idx] = make_double2(dx * f, dy * f);
            if (a.out_wave) a.out_wave[oi] = make_float2((float)dx, (float)dy);
            continue;
        }
        if (a.meas_dev && a.meas && !a.far && !a.pdet) {
            const size_t oidx = idx;
            if (a.out_wave) a.out_wave[oidx] = cadd(d, a.carrier);
            a.field[idx] = loss_seed_dev(d, a.carrier, sqrtf(a.carrier.x * a.carrier.x + a.carrier.y * a.carrier.y), a.meas[oidx],
                                         a.seed_scale, acc, acc2, a.dref);
            continue;
        }
        const cf e0 = d;
        if (a.pdet64 && a.meas) {
            const size_t oi = a.far ? ((size_t)b * a.NY + y) * a.NX + x : idx;
            cf dw;
            double2 p = a.pdet64[(size_t)x * a.NY + y];
            if (a.pscale.x != 0.0 || a.pscale.y != 0.0) p = make_double2(p.x * a.pscale.x - p.y * a.pscale.y, p.x * a.pscale.y + p.y * a.pscale.x);
            a.field[idx] = loss_seed_f64(d, p, a.meas[oi], a.seed_scale, acc, acc2, dw);
            if (a.out_wave) a.out_wave[oi] = dw;
            continue;
        }
        if (a.pdet) d = cadd(d, a.pdet[(size_t)x * a.NY + y]);
        else if (!a.far || (x == 0 && y == 0)) d = cadd(d, a.carrier);
        const size_t oidx = a.far ? ((size_t)b * a.NY + y) * a.NX + x : idx;
        if (a.out_wave) a.out_wave[oidx] = d;
        if (a.meas && a.gcar && a.far && x == 0 && y == 0) {
            // DC bin in float64, its seed kept out of the transforms (AdjCarrier)
            const double dx = a.carrier_dd.x + (double)e0.x, dy = a.carrier_dd.y + (double)e0.y;
            const double ab = sqrt(dx * dx + dy * dy), rr = ab - (double)a.meas[oidx];
            acc += rr * rr;
            acc2 += rr * ab;
            const double f = ab > 0.0 ? (double)a.seed_scale * rr / ab : 0.0;
            const double2 s0 = make_double2(dx * f, dy * f);
            a.gcar[b] = s0;
            a.gt0[b] = make_double2(a.a_end.x * s0.x + a.a_end.y * s0.y, a.a_end.x * s0.y - a.a_end.y * s0.x);
            a.field[idx] = make_float2(0.f, 0.f);
            continue;
        }
        if (a.meas) a.field[idx] = loss_seed(d, a.meas[oidx], a.seed_scale, acc, acc2);
    }
    if (a.meas) {
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

struct GBwdArgs {
    cf* g;               // G(phi_z) in, G(psi_z) out (in place)
    const cf* tape;      // phi_z (eps part)
    float2* grot;
    ObjView obj;
    int B, NX, NY, z;
    float k;
    cf carrier;          // cbar a_z: constant part of phi_z
    int full_tape;       // the tape holds the full phi (carrier field), not its scattered part
    AdjCarrier ac;       // gcar nullable
};

__global__ __launch_bounds__(256) void k_g_bwd(GBwdArgs a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % a.NY;
        const size_t r = idx / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        const cf G = a.g[idx];
        const cf e = a.tape[idx];
        const cf phi = a.full_tape ? e : cadd(e, a.carrier);
        cf t = cmulc(G, phi);
        const float2 m1 = g_mod_value(a.obj, b, x, y, a.z, a.NX);
        cf Gn = cmulc(G, make_float2(1.f + m1.x, m1.y));
        if (a.ac.gcar) {
            cf gam, t0;
            adj_carrier_load(a.ac, b, gam, t0);
            t = cadd(cadd(t, cmulc(gam, e)), t0);
            Gn = cadd(Gn, cmulc(gam, csub(m1, a.ac.cbm1)));
            // the sweep ends at slice 0: leave the FULL G(psi_0) = scattered part + conj(cbar) gamma for bdof_probe_grad
            if (a.z == 0) Gn = cadd(Gn, cmulc(gam, make_float2(1.f + a.ac.cbm1.x, a.ac.cbm1.y)));
        }
        a.grot[(((size_t)b * a.obj.S + a.z) * a.NX + x) * a.NY + y] = make_float2(a.k * t.y, -a.k * t.x);
        a.g[idx] = Gn;
    }
}


// ---- float64 adjoint sweep (bdof_configure flag 64) -------------------------------------------------------------------------
// The adjoint field G is as large as the wave itself and has no known part to split off (it is driven by the data's noise), so
// its float32 transform chain sets a floor of ~3e-6 under the gradient (64 slices) — and Adam's first step of every epoch,
// lr g / (|g| + 1e-8), turns an absolute error of 1e-8 at a voxel where the gradient changes sign into a fraction of a whole
// step.  With this option the seed, the adjoint transforms (rocFFT double precision), the transfer function and
// the products conj(phi) G are float64; the forward sweep (scattered wave on its carrier) and the tape stay float32.
struct GBwd64Args {
    double2* g;          // G(phi_z) in, G(psi_z) out
    const cf* tape;      // phi_z (scattered part, or the full phi with a carrier field)
    float2* grot;
    ObjView obj;
    int B, NX, NY, z;
    double k;
    double2 carrier;     // constant part of phi_z
    int full_tape;
};
__global__ __launch_bounds__(256) void k_g_bwd64(GBwd64Args a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = idx % a.NY;
        const size_t r = idx / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        const double2 G = a.g[idx];
        const cf e = a.tape[idx];
        const double px = (double)e.x + (a.full_tape ? 0.0 : a.carrier.x), py = (double)e.y + (a.full_tape ? 0.0 : a.carrier.y);
        const double tx = G.x * px + G.y * py, ty = G.y * px - G.x * py;             // G conj(phi)
        a.grot[(((size_t)b * a.obj.S + a.z) * a.NX + x) * a.NY + y] = make_float2((float)(a.k * ty), (float)(-a.k * tx));
        const float2 m1 = g_mod_value(a.obj, b, x, y, a.z, a.NX);
        const double cx = 1.0 + (double)m1.x, cy = (double)m1.y;
        a.g[idx] = make_double2(G.x * cx + G.y * cy, G.y * cx - G.x * cy);           // G conj(c)
    }
}
// field[b][kx][ky] *= h[ky][kx] (or its conjugate), float64
__global__ __launch_bounds__(256) void k_g_hmul64(double2* field, const double2* h, int B, int NX, int NY, int conj_h) {
    const size_t n = (size_t)B * NX * NY;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int ky = idx % NY;
        const int kx = (idx / NY) % NX;
        double2 t = h[(size_t)ky * NX + kx];
        if (conj_h) t.y = -t.y;
        const double2 v = field[idx];
        field[idx] = make_double2(v.x * t.x - v.y * t.y, v.x * t.y + v.y * t.x);
    }
}
__global__ __launch_bounds__(256) void k_d_scale(double2* f, size_t n, double s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        f[i] = make_double2(f[i].x * s, f[i].y * s);
}

// ---- float64 helpers of bdof_set_probe_field: the carrier field of a localised probe, propagated on the device ------------
__global__ __launch_bounds__(256) void k_d_mul(double2* __restrict__ f, const double2* __restrict__ h, size_t n, double scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 a = f[i], b = h[i];
        f[i] = make_double2((a.x * b.x - a.y * b.y) * scale, (a.x * b.y + a.y * b.x) * scale);
    }
}
__global__ __launch_bounds__(256) void k_d_copy(const double2* __restrict__ src, double2* __restrict__ dst, int n0, int n1, int transposed) {
    const size_t n = (size_t)n0 * n1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
        dst[transposed ? (idx % n1) * (size_t)n0 + idx / n1 : idx] = src[idx];
}
// dst[i] = (float2) src[i]; transposed: dst[j * n0 + i] = src[i * n1 + j] for an [n0][n1] source
__global__ __launch_bounds__(256) void k_d_to_f(const double2* __restrict__ src, cf* __restrict__ dst, int n0, int n1, int transposed) {
    const size_t n = (size_t)n0 * n1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const double2 v = src[idx];
        const size_t o = transposed ? (idx % n1) * (size_t)n0 + idx / n1 : idx;
        dst[o] = make_float2((float)v.x, (float)v.y);
    }
}
