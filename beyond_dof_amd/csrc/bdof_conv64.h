// The real-space propagator of cnn_propagator/propagation.py:18-133 in float64 (bdof_loss_grad_conv_f64): an ACCURACY path for the
// one Adam step per epoch in which a float32 gradient shows (adjoint_precision='first-step'), not a fast one.
//
// The reference pads the wave with a running constant and convolves it ('valid') with the ks x ks crop of the real-space Fresnel
// kernel after every slice, renormalises by the corner pixel, and differentiates all of it in float64.  Here the convolution of a
// slice is a circular convolution on the padded (N + ks - 1)^2 grid — one rocFFT double-precision transform pair with the
// transform of the zero-padded kernel (bdof_fields_free_step's table) — whose 'valid' part is exactly the reference's output
// (overlap-save); its adjoint is the circular correlation of the embedded adjoint field, cropped to the un-padded pixels (the
// padding constant does not depend on the object).  Everything else is point-wise, in double: modulation from the caller's
// (delta, beta) rows, the corner-pixel renormalisation s = psi_0[0,0,0] / psi_S[0,0,0] (ONE scalar for the whole batch,
// propagation.py:79,109-110) and its adjoint -sum(G conj q) / conj(P_000) into pixel (0,0,0), the magnitude loss and its seed.
//
// The transfer-function model of np_funcs.py:15-65 runs through the same point-wise kernels (bdof_loss_grad_tf_f64): the step
// after a slice is then one transform pair on the field's own grid with H in float64, nothing is padded and nothing renormalised.
#pragma once
#include "bdof_field.h"

// dst[b][X][Y] (M x M) = src[b][X - off][Y - off] (N x N) inside, `edge` outside
__global__ __launch_bounds__(256) void k_c64_pad(const double2* __restrict__ src, double2* __restrict__ dst, int B, int N, int M, int off, double2 edge) {
    const size_t n = (size_t)B * M * M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int Y = i % M;
        const size_t r = i / M;
        const int X = r % M, b = r / M;
        const int x = X - off, y = Y - off;
        dst[i] = (x >= 0 && x < N && y >= 0 && y < N) ? src[((size_t)b * N + x) * N + y] : edge;
    }
}
// dst[b][x][y] (N x N) = src[b][x + off][y + off] (M x M)
__global__ __launch_bounds__(256) void k_c64_crop(const double2* __restrict__ src, double2* __restrict__ dst, int B, int N, int M, int off) {
    const size_t n = (size_t)B * N * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = i % N;
        const size_t r = i / N;
        const int x = r % N, b = r / N;
        dst[i] = src[((size_t)b * M + x + off) * M + y + off];
    }
}
// every wavefield of the batch starts from the same probe
__global__ __launch_bounds__(256) void k_c64_bcast(const double2* __restrict__ probe, double2* __restrict__ dst, int B, size_t per) {
    const size_t n = per * B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = probe[i % per];
}
// scal[0] = P[0] (batch element 0, pixel (0, 0)); scal[1] = s = init / P[0]
__global__ void k_c64_corner(const double2* P, double2 init, double2* scal) {
    const double2 p = P[0];
    const double inv = 1.0 / (p.x * p.x + p.y * p.y);
    scal[0] = p;
    scal[1] = make_double2((init.x * p.x + init.y * p.y) * inv, (init.y * p.x - init.x * p.y) * inv);
}
// f *= s (conj_s: conj(s))
__global__ __launch_bounds__(256) void k_c64_scale(double2* f, size_t n, const double2* scal, int conj_s) {
    double2 s = scal[1];
    if (conj_s) s.y = -s.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = f[i];
        f[i] = make_double2(v.x * s.x - v.y * s.y, v.x * s.y + v.y * s.x);
    }
}
// loss partials and seed in place: d [B][NX][NY]; far: d is [b][kx][ky] and meas [b][ky][kx] (un-shifted), else both [b][x][y].
// meas holds m - meas_ref (residual splitting of the float32 path); G = 2 (|d| - m) d / |d| * seed_scale
__global__ __launch_bounds__(256) void k_c64_loss(double2* d, const float* __restrict__ meas, double* partial, int B, int NX, int NY, int far,
                                                 double meas_ref, double seed_scale) {
    const size_t n = (size_t)B * NX * NY;
    double acc = 0.0, acc2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = i % NY;
        const size_t r = i / NY;
        const int x = r % NX, b = r / NX;
        const double m = (double)meas[far ? ((size_t)b * NY + y) * NX + x : i] + meas_ref;
        const double2 v = d[i];
        const double a = sqrt(v.x * v.x + v.y * v.y);
        const double res = a - m;
        acc += res * res;
        acc2 += res * a;
        const double f = a > 0.0 ? seed_scale * res / a : 0.0;
        d[i] = make_double2(v.x * f, v.y * f);
    }
    __shared__ double w[2][4];
    acc = wave_reduce_sum(acc);
    acc2 = wave_reduce_sum(acc2);
    if ((threadIdx.x & 63) == 0) { w[0][threadIdx.x >> 6] = acc; w[1][threadIdx.x >> 6] = acc2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = w[0][0] + w[0][1] + w[0][2] + w[0][3];
        partial[2 * blockIdx.x + 1] = w[1][0] + w[1][1] + w[1][2] + w[1][3];
    }
}
// per-workgroup partial sums of G conj(q) (complex), then the renormalisation's adjoint: G <- conj(s) G, G[0] -= T / conj(P000)
__global__ __launch_bounds__(256) void k_c64_dot(const double2* __restrict__ G, const double2* __restrict__ q, size_t n, double2* part) {
    double sx = 0.0, sy = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 g = G[i], v = q[i];
        sx += g.x * v.x + g.y * v.y;
        sy += g.y * v.x - g.x * v.y;
    }
    __shared__ double w[2][4];
    sx = wave_reduce_sum(sx);
    sy = wave_reduce_sum(sy);
    if ((threadIdx.x & 63) == 0) { w[0][threadIdx.x >> 6] = sx; w[1][threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = make_double2(w[0][0] + w[0][1] + w[0][2] + w[0][3], w[1][0] + w[1][1] + w[1][2] + w[1][3]);
}
__global__ void k_c64_corner_adj(double2* G, const double2* part, int npart, const double2* scal) {
    double tx = 0.0, ty = 0.0;
    for (int j = 0; j < npart; ++j) { tx += part[j].x; ty += part[j].y; }      // fixed order
    const double2 p = scal[0];                                                  // P000; divide T by conj(P000)
    const double inv = 1.0 / (p.x * p.x + p.y * p.y);
    // T / conj(p) = T p / |p|^2
    const double qx = (tx * p.x - ty * p.y) * inv, qy = (tx * p.y + ty * p.x) * inv;
    G[0].x -= qx;
    G[0].y -= qy;
}
// point-wise adjoint of slice z: t = conj(phi) G; gradient rows (k Im t, -k Re t); G <- conj(c) G
struct C64BwdArgs {
    double2* G;             // [B][NX][NY] G(phi_z), overwritten by G(psi_z)
    const double2* phi;     // tape of slice z
    float2* grot;           // [B][S][NX][NY]
    ObjView obj;            // .vol = (delta, beta) rows
    int B, NX, NY, S, z;
    double k;
};
__global__ __launch_bounds__(256) void k_c64_bwd(C64BwdArgs a) {
    const size_t n = (size_t)a.B * a.NX * a.NY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = i % a.NY;
        const size_t r = i / a.NY;
        const int x = r % a.NX, b = r / a.NX;
        const double2 g = a.G[i], p = a.phi[i];
        const double tx = p.x * g.x + p.y * g.y, ty = p.x * g.y - p.y * g.x;          // conj(phi) G
        a.grot[(((size_t)b * a.S + a.z) * a.NX + x) * a.NY + y] = make_float2((float)(a.k * ty), (float)(-a.k * tx));
        const float2 db = g_mod_value(a.obj, b, x, y, a.z, a.NX);
        double s, cs;
        sincos(a.k * (double)db.x, &s, &cs);
        const double e = exp(-a.k * (double)db.y);
        // conj(c) G, c = e (cs + i s)
        a.G[i] = make_double2(e * (cs * g.x + s * g.y), e * (cs * g.y - s * g.x));
    }
}
