// libbdof.so — host side of the C ABI declared in include/bdof.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cmath>
#include <algorithm>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bdof.h"
#include "bdof_kernels.h"
#include "bdof_conv2.h"
#include "bdof_generic.h"
#include "bdof_field.h"
#include "bdof_conv64.h"
#include "bdof_resident.h"
#include "bdof_comm.h"
#include <rocfft/rocfft.h>
#include <map>
#include <array>

#define BDOF_ERR_ARG (-1)
#define BDOF_ERR_STATE (-2)
#define BDOF_ERR_SIZE (-3)

#define BDOF_MAX_GROUPS 4
#define BDOF_N_TIMERS 16
#define BDOF_MAX_DEVICES 64
struct bdof_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // dual-stream split of a batch (see batch_groups): the sub-batch the launchers currently work on
    hipStream_t side[BDOF_MAX_GROUPS - 1] = {};   // side streams PROVEN to run concurrently with the ctx stream (probe_side_streams)
    int n_side = -1;                             // -1: not probed yet
    std::vector<hipStream_t> side_all;           // every candidate created (kept until the ctx dies)
    hipEvent_t ev_fork = nullptr, ev_join[BDOF_MAX_GROUPS - 1] = {};
    hipEvent_t timer[BDOF_N_TIMERS] = {};        // bdof_timer_mark
    int n_streams = -1;                          // -1 auto (1 or 2), else the number of groups to split a batch in
    int sub_b0 = 0, sub_part = 0;
    hipStream_t sub_stream = nullptr;
    int ncu = 256;
    int NY = 0, NX = 0, S = 0, Bmax = 0;
    bool with_grad = false;
    bool recompute = false;                      // tape-free adjoint (bdof_configure flag 16): the tape holds 3 fields, not S
    cf *twY = nullptr, *twX = nullptr;
    // dithered twiddle tables (bdof_fft.h; BDOF_TW_DITHER=D, default 64): D copies of each table, entry j of copy d rounded up or
    // down so that the mean over the D copies is the float64 value to ulp / D; the launches of slice z take copy z mod D
    int tw_dither = 0;
    unsigned tw_tick = 0;          // the slice of the last A / A' launch: the transfer-function launch that follows takes the same copy
    cf *hs = nullptr, *hdet = nullptr, *hcomb = nullptr, *probe = nullptr;
    // float64 real-space propagator (bdof_set_conv_f64 / bdof_loss_grad_conv_f64, bdof_conv64.h)
    double2 *c64_probe = nullptr, *c64_khat = nullptr, *c64_psi = nullptr, *c64_q = nullptr, *c64_big = nullptr, *c64_tape = nullptr,
            *c64_scal = nullptr, *c64_part = nullptr;
    int c64_ks = 0, c64_B = 0;
    double2* car_scratch = nullptr;             // bdof_range_carrier_build: [nz - 1][B][NX][NY] complex128
    size_t car_scratch_n = 0;
    hipStream_t aux = nullptr;                  // bdof_fields_free_step_aux: the whole-field step of a stitch range beside the tiles' sweeps
    hipEvent_t ev_aux_fork = nullptr, ev_aux_join = nullptr;
    rocfft_execution_info ginfo_aux = nullptr;
    void* gwork_aux = nullptr;
    size_t gwork_aux_sz = 0;
    bool aux_pending = false;
    const cf* range_car = nullptr;              // bdof_set_range_carrier: [nz][B][NX][NY] carrier fields of the range being swept
    int range_car_B = 0, range_car_z0 = 0;
    bool c64_tf = false;                        // the float64 path holds the transfer-function model (bdof_set_tf_f64), not the real-space one
    double2 *c64_h = nullptr, *c64_hdet = nullptr;
    std::complex<double> c64_ksum{1.0, 0.0};
    double c64_k = 0.0;
    cf* hsT_d = nullptr;           // the same copies in the LDS-resident kernel's [kx][ky] order
    cf* hs_d = nullptr;            // bdof_set_transfer_f64: hs_copies dithered float32 copies of the slice step's table (bdof_field.h)
    int hs_copies = 0;
    const cf* hs_override = nullptr;   // bdof_forward_range_h: the table of this call's transfer-function steps
    cf *bufA = nullptr, *bufB = nullptr, *tape = nullptr;
    float2* grot = nullptr;
    double2 *gcar = nullptr, *gt0 = nullptr;     // adjoint carrier per wavefield (AdjCarrier, bdof_kernels.h)
    cf* gpsi0 = nullptr;                         // [Bmax][NX][NY] G(psi_0) per wavefield (bdof_enable_probe_grad)
    const cf* gpsi_src = nullptr;                // where the last bdof_loss_grad left it (gpsi0, or the generic engine's field)
    int gpsi_B = 0;
    double *partial = nullptr, *loss_dev = nullptr;
    int npartial = 0;
    float k = 0.f;                              // the k the modulation table in c->mod was built with
    float k_fft = 0.f;                          // bdof_set_physics' k (the reference's PI literal, quirk Q1)
    std::complex<double> h00{1.0, 0.0}, hdet00{1.0, 0.0}, a0{0.0, 0.0};   // carrier splitting (bdof_kernels.h)
    std::complex<double> cbm1{0.0, 0.0};        // cbar - 1: mean modulation factor of the object minus one (modulate_eps_s)
    double2* cbar_dev = nullptr;                // [npartial_cb + 1] per-workgroup sums of the modulation table, then the mean
    int ncb = 0;
    // real-space truncated-kernel propagator (bdof_set_conv)
    bool have_conv = false;
    ConvTaps taps{};
    ConvTaps* taps_dev = nullptr;
    int taps_copies = 1;          // dithered copies of the taps in taps_dev (bdof_set_conv_taps_f64), slice z takes copy z mod taps_copies
    std::complex<double> ksum{1.0, 0.0};
    float k_conv = 0.f;
    cf *bufC = nullptr, *conv_scal = nullptr;
    // carrier field of the real-space propagator (bdof_set_conv_probe_stack): p_0 .. p_S, float32 planes [S + 1][NX][NY]; the
    // detector plane of p_S in float64; corner pixels of p_0 and p_S (the renormalisation s = p_0[0,0] / psi_S[0,0,0])
    cf* cstack = nullptr;
    double2* cdet64 = nullptr;
    std::complex<double> c_p0{0.0, 0.0}, c_pS{0.0, 0.0};
    // LDS-resident engine (small square fields, bdof_resident.h)
    bool resident = false, res_dirty = true, res_always = false;
    cf *hsT = nullptr, *hdetT = nullptr, *twR = nullptr, *res_carrier = nullptr;
    int meas_dev = 0;                           // bdof_set_meas_mode
    cf *pstack = nullptr, *pdet = nullptr, *pdetT = nullptr;      // carrier field of a localised probe (bdof_set_probe_stack); pdetT = det transposed
    double2 *pdet64 = nullptr, *pdetT64 = nullptr;                // the same planes in float64 (bdof_set_probe_field): the residual |d| - m is formed in float64
    // generic-size engine (rocFFT): one plan pair per batch size
    bool generic = false;
    std::map<int, std::pair<rocfft_plan, rocfft_plan>> gplans;
    // float64 adjoint sweep (bdof_configure flag 64; generic engine)
    bool adj64 = false, have_h64 = false;
    std::map<int, std::pair<rocfft_plan, rocfft_plan>> gplans64;
    std::map<std::array<int, 4>, std::pair<rocfft_plan, rocfft_plan>> fplans;      // bdof_fields_free_step: (NX, NY, B, double)
    double2 *g64 = nullptr, *hs64 = nullptr, *hdet64 = nullptr;
    rocfft_execution_info ginfo = nullptr;
    void* gwork = nullptr;
    size_t gwork_sz = 0;
    int det_mode = BDOF_DET_NONE, variant = BDOF_VARIANT_NUMPY_SKIP_LAST;
    bool have_physics = false, have_probe = false, tape_valid = false, last_valid = false;
    ObjView obj{};
    const float2* obj_src = nullptr;            // caller's (delta, beta) rows
    bool obj_bound_mod = false;                 // bdof_set_object_bilinear: c->mod was written directly, there is no obj_src
    size_t obj_rows = 0;
    float2* mod = nullptr;                      // c - 1 table of those rows (k_modulation_table)
    size_t mod_cap = 0;
    bool mod_dirty = true;
    int n_angles = 0;
    const int *adj_off = nullptr, *adj_order = nullptr;
    int adj_ndest = 0;
    float2* winpad = nullptr;                   // [S][volNX][volNY] rotated-frame gradient of the ptychography windows
    size_t winpad_sz = 0;
    int* win_angle = nullptr;                   // device copy of the batch's angle index
    int* heavy = nullptr;                       // [1 + n_dest]: counter + deferred rows of the rotation adjoint
    // profiling
    bool prof = false;
    int prof_stride = 1;                        // time every prof_stride-th launch of a class
    unsigned prof_seen[BDOF_K_COUNT] = {0};
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, int>> ev_used;   // (class, index of start event)
    size_t ev_next = 0;
    double prof_ms[BDOF_K_COUNT] = {0};
    int prof_n[BDOF_K_COUNT] = {0};
    std::string err;
};

static int fail(bdof_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPC(c, call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            return fail((c), (int)e_, std::string(#call) + ": " + hipGetErrorString(e_));          \
        }                                                                                          \
    } while (0)

// Mean-refraction carrier (modulate_eps_s): constant part of the wave entering slice z, a_z = a_0 (cbar H00)^z
// (H00 = DC value of the transfer function, cbar = mean modulation factor of the object)
static std::complex<double> carrier_z(const bdof_ctx* c, int z) { return c->a0 * std::pow((1.0 + c->cbm1) * c->h00, z); }
static cf cfl(std::complex<double> a) { return make_float2((float)a.real(), (float)a.imag()); }
static cf carrier_at(const bdof_ctx* c, int z) { return cfl(carrier_z(c, z)); }
static cf cshift_at(const bdof_ctx* c, int z) { return cfl(carrier_z(c, z) * c->cbm1); }                  // a_z (cbar - 1)
static cf carrier_phi_at(const bdof_ctx* c, int z) { return cfl(carrier_z(c, z) * (1.0 + c->cbm1)); }      // constant part of phi_z
// constant part of the detector wave (real-space detectors) / of the wave whose fft2 is the far field
static std::complex<double> carrier_end(const bdof_ctx* c) {
    std::complex<double> a = carrier_z(c, c->S - 1) * (1.0 + c->cbm1);
    if (c->det_mode != BDOF_DET_FAR && c->variant == BDOF_VARIANT_TF_ALL) a *= c->h00;
    if (c->det_mode == BDOF_DET_NEAR) a *= c->hdet00;
    return a;
}
static cf carrier_det(const bdof_ctx* c) {
    std::complex<double> a = carrier_end(c);
    if (c->det_mode == BDOF_DET_FAR) a *= (double)c->NX * (double)c->NY;
    return make_float2((float)a.real(), (float)a.imag());
}
// bdof_set_meas_mode(1): the host subtracted |a_0| from the amplitudes; the detector carrier has modulus |a_0| |cbar|^S
static float meas_dref(const bdof_ctx* c) { return (float)(std::abs(carrier_end(c)) - std::abs(c->a0)); }
// the object's mean modulation factor rides on the carrier when the carrier is a scalar of the transfer-function path
static bool want_cbar(const bdof_ctx* c) {
    static const bool off = std::getenv("BDOF_NO_MEAN_CARRIER") != nullptr;
    return !off && std::abs(c->a0) > 0.0 && !c->pstack && !c->have_conv;
}

// far-field detector + plane-wave carrier: the DC seed of the adjoint is carried as a float64 scalar (AdjCarrier)
static bool use_adj_carrier(const bdof_ctx* c) {
    static const bool off = std::getenv("BDOF_NO_ADJ_CARRIER") != nullptr;
    return !off && c->det_mode == BDOF_DET_FAR && !c->pstack && c->gcar && std::abs(c->a0) > 0.0;
}
static double2 d2(std::complex<double> v) { return make_double2(v.real(), v.imag()); }
// conj(H00)^n: what the adjoint carrier picks up in n adjoint transfer-function steps
static AdjCarrier adj_carrier_at(const bdof_ctx* c, int steps_back) {
    if (!use_adj_carrier(c)) return AdjCarrier{nullptr, nullptr, make_double2(1.0, 0.0), make_float2(0.f, 0.f)};
    return AdjCarrier{c->gcar + c->sub_b0, c->gt0 + c->sub_b0, d2(std::pow(std::conj((1.0 + c->cbm1) * c->h00), steps_back)), cfl(c->cbm1)};
}

static bool supported_n(int n) { return n == 64 || n == 128 || n == 256 || n == 512 || n == 1024; }

// ---- profiling helpers -----------------------------------------------------------------------
struct ProfScope {
    bdof_ctx* c;
    int cls;
    bool on;
    bool ext;             // the launch itself carries the events (BDOF_LAUNCH -> hipExtLaunchKernelGGL): nothing is recorded here
    size_t idx = 0;
    ProfScope(bdof_ctx* c_, int cls_, bool ext_ = false) : c(c_), cls(cls_), on(c_->prof), ext(ext_) {
        if (on && c->prof_stride > 1 && cls <= BDOF_K_ROW_BWD) on = (c->prof_seen[cls]++ % (unsigned)c->prof_stride) == 0;
        if (!on) return;
        if (c->ev_next + 2 > c->ev_pool.size()) {
            size_t old = c->ev_pool.size();
            c->ev_pool.resize(old + 4096);
            for (size_t i = old; i < c->ev_pool.size(); ++i) (void)hipEventCreate(&c->ev_pool[i]);
        }
        idx = c->ev_next;
        c->ev_next += 2;
        if (!ext) (void)hipEventRecord(c->ev_pool[idx], c->sub_stream ? c->sub_stream : c->stream);
    }
    hipEvent_t e0() const { return c->ev_pool[idx]; }
    hipEvent_t e1() const { return c->ev_pool[idx + 1]; }
    ~ProfScope() {
        if (!on) return;
        if (!ext) (void)hipEventRecord(c->ev_pool[idx + 1], c->sub_stream ? c->sub_stream : c->stream);
        c->ev_used.emplace_back(cls, (int)idx);
    }
};
// A timed launch of the per-slice kernel classes: hipExtLaunchKernelGGL stamps the two events with the dispatch's own begin
// and end (what rocprofv3 reports), where an event pair recorded around the launch also contains the ~2.5 us between the
// start marker and the first wave.
#define BDOF_LAUNCH(ps, kernel, grid, blk, shmem, stream, ...)                                                     \
    do {                                                                                                           \
        if ((ps).on) hipExtLaunchKernelGGL(kernel, grid, blk, shmem, stream, (ps).e0(), (ps).e1(), 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, blk, shmem, stream, __VA_ARGS__);                                    \
    } while (0)

static void prof_collect(bdof_ctx* c) {
    for (auto& u : c->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[u.second], c->ev_pool[u.second + 1]) == hipSuccess) {
            c->prof_ms[u.first] += ms;
            c->prof_n[u.first] += 1;
        }
    }
    c->ev_used.clear();
    c->ev_next = 0;
}

// ---- launches --------------------------------------------------------------------------------
// Persistent grids: at most `per_cu` workgroups per CU, and every workgroup gets the same number of tiles.
static int balanced_grid(const bdof_ctx* c, int tiles, int per_cu) {
    const int cap = c->ncu * per_cu;
    if (tiles <= cap) return tiles;
    const int rounds = (tiles + cap - 1) / cap;
    return (tiles + rounds - 1) / rounds;
}
template <int N> static int rows_grid(const bdof_ctx* c, int B, int R) {
    return balanced_grid(c, B * R / RowCfg<N>::TILE, N >= 1024 ? 1 : 2);
}

#define DISPATCH_N(n, EXPR)                                    \
    switch (n) {                                               \
        case 64: { constexpr int N_ = 64; EXPR; } break;       \
        case 128: { constexpr int N_ = 128; EXPR; } break;     \
        case 256: { constexpr int N_ = 256; EXPR; } break;     \
        case 512: { constexpr int N_ = 512; EXPR; } break;     \
        case 1024: { constexpr int N_ = 1024; EXPR; } break;   \
        default: break;                                        \
    }

// ---- sub-batches --------------------------------------------------------------------------------
// A batch whose tile count fills the chip's workgroup slots badly (25 wavefields of 512 rows = 800 tiles on 512 slots:
// every launch runs as two rounds at 78 % occupancy) is split in two groups that run the same kernel sequence on two
// streams: the groups have no dependencies on each other, so one group's kernels fill the slots the other leaves idle
// and the per-launch round quantisation disappears.  Launchers address the current group through c->sub_*.
struct Group { int b0, B; hipStream_t st; };

// HIP maps streams onto a small pool of hardware queues (4 per process by default, GPU_MAX_HW_QUEUES): two streams that
// share a queue run their kernels one after the other, and which streams collide depends on every other stream in the
// process (torch's, RCCL's ...).  Splitting a batch over two such streams only adds launches.  So side streams are
// admitted by measurement: a 200-us spin kernel on each of two streams takes ~200 us when they overlap, ~400 when not.
__global__ void k_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && ticks < 0) *sink = 1;
}

static bool streams_overlap(bdof_ctx* c, hipStream_t a, hipStream_t b) {
    hipEvent_t e0, e1, ef, ej;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    const long long ticks = 20000;                 // wall_clock64 counts at 100 MHz: 200 us
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {            // the first round also pays the lazy creation of the queues
        (void)hipEventRecord(e0, a);
        (void)hipEventRecord(ef, a);
        (void)hipStreamWaitEvent(b, ef, 0);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, (int*)nullptr);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, (int*)nullptr);
        (void)hipEventRecord(ej, b);
        (void)hipStreamWaitEvent(a, ej, 0);
        (void)hipEventRecord(e1, a);
        if (hipStreamSynchronize(a) != hipSuccess) { best = 1e30f; break; }
        float ms = 1e30f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ef); (void)hipEventDestroy(ej);
    (void)c;
    return best < 0.3f;                            // 0.2 ms when concurrent, 0.4 ms when serialised
}

static void probe_side_streams(bdof_ctx* c) {
    c->n_side = 0;
    for (int cand = 0; cand < 8 && c->n_side < BDOF_MAX_GROUPS - 1; ++cand) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
        c->side_all.push_back(s);
        bool ok = streams_overlap(c, c->stream, s);
        for (int i = 0; ok && i < c->n_side; ++i) ok = streams_overlap(c, c->side[i], s);
        if (ok) c->side[c->n_side++] = s;
    }
}

static int batch_groups(bdof_ctx* c, int B, int rows_per_b, int tile, Group (&g)[BDOF_MAX_GROUPS]) {
    g[0] = Group{0, B, c->stream};
    const int slots = c->ncu * 2;
    const long tiles = (long)B * rows_per_b / tile;
    int n = c->n_streams;
    if (n < 0) n = (B >= 4 && tiles >= slots) ? 2 : 1;      // measured: +3..12 % from 512 tiles up, a loss below
    if (n > BDOF_MAX_GROUPS) n = BDOF_MAX_GROUPS;
    if (n > B) n = B;
    if (n <= 1) return 1;
    if (c->n_side < 0) probe_side_streams(c);
    if (n > 1 + c->n_side) n = 1 + c->n_side;
    if (n <= 1) return 1;
    for (int i = 0, b0 = 0; i < n; ++i) {
        const int Bg = B / n + (i < B % n ? 1 : 0);
        g[i] = Group{b0, Bg, i == 0 ? c->stream : c->side[i - 1]};
        b0 += Bg;
    }
    return n;
}
static void use_group(bdof_ctx* c, const Group& g, int part_off = 0) {
    c->sub_b0 = g.b0;
    c->sub_stream = g.st;
    c->sub_part = part_off;
}
static void use_whole(bdof_ctx* c) { c->sub_b0 = 0; c->sub_stream = c->stream; c->sub_part = 0; }
static int fork_streams(bdof_ctx* c, int ngroups) {
    if (ngroups < 2) return 0;
    HIPC(c, hipEventRecord(c->ev_fork, c->stream));
    for (int i = 1; i < ngroups; ++i) HIPC(c, hipStreamWaitEvent(c->side[i - 1], c->ev_fork, 0));
    return 0;
}
static int join_streams(bdof_ctx* c, int ngroups) {
    use_whole(c);
    for (int i = 1; i < ngroups; ++i) {
        HIPC(c, hipEventRecord(c->ev_join[i - 1], c->side[i - 1]));
        HIPC(c, hipStreamWaitEvent(c->stream, c->ev_join[i - 1], 0));
    }
    return 0;
}

static ObjView sub_obj(const bdof_ctx* c) {
    ObjView o = c->obj;
    const int b0 = c->sub_b0;
    if (b0) {
        if (o.tab) { o.angle_of_b += b0; }
        else { o.vol += (size_t)b0 * c->S * c->NX * o.volNY; }
        if (o.xoff) o.xoff += b0;
        if (o.yoff) o.yoff += b0;
    }
    return o;
}
// copy d of a dithered constant: the float32 below or above v, the upper one in a fraction p = (v - below) / ulp of the copies,
// spread evenly (copy d takes it iff floor((d + 1) p + phase) > floor(d p + phase)): the mean over any run of copies is v to
// ulp / (length of the run)
static float dither_pick(double v, int d, double phase) {
    const float r = (float)v;
    float f0 = r, f1 = r;
    if ((double)r > v) f0 = std::nextafterf(r, -4.f); else if ((double)r < v) f1 = std::nextafterf(r, 4.f); else return r;
    const double p = (v - (double)f0) / ((double)f1 - (double)f0);
    return std::floor((d + 1) * p + phase) > std::floor(d * p + phase) ? f1 : f0;
}

// sqrt(1/2) of a slice launch for the transforms compiled with ROUND 1 / ROUND 2 (bdof_fft.h): the nearest float32 and its upper
// neighbour (one transform in four rounds up: the defects cancel to a quarter) — or, with dithered tables, one value for both,
// the upper neighbour in 20.3 % of the slices so that the mean over the slices is sqrt(1/2)
static void sq_of(const bdof_ctx* c, unsigned key, float (&sq)[2]) {
    if (c->tw_dither > 0) sq[0] = sq[1] = dither_pick(0.70710678118654752440, (int)(key % (unsigned)c->tw_dither), 0.5);
    else { sq[0] = 0.70710678118654752f; sq[1] = 0.70710682868957520f; }
}
// the table copy of a slice launch
static const cf* tw_of(bdof_ctx* c, const cf* base, int N, unsigned key) {
    return c->tw_dither > 0 ? base + (size_t)(key % (unsigned)c->tw_dither) * 2 * N : base;
}

template <class T> static T* sub_field(const bdof_ctx* c, T* p) { return p ? p + (size_t)c->sub_b0 * c->NX * c->NY : p; }

// A_z: L1 (or the probe) -> L2 (tstore) or L1 (plain)
static const cf* slice_carrier_field(const bdof_ctx* c, int z) { return c->pstack ? c->pstack + (size_t)z * c->NX * c->NY : nullptr; }

// start != nullptr: slice z starts from the real-space fields start[b] (scattered part) instead of the hybrid `in` — the
// first slice of a range (bdof_forward_range); z == 0 without `start` starts from the probe shared by the batch.
static void launch_row_fwd(bdof_ctx* c, int B, int z, const cf* in, cf* out, bool tstore, cf* phi_out = nullptr, const cf* start = nullptr) {
    ProfScope ps(c, BDOF_K_ROW_FWD, true);
    RowFwdArgs a{sub_field(c, in), start ? sub_field(c, start) : c->probe, sub_field(c, out), sub_field(c, phi_out), sub_obj(c), B, c->NX, z,
                 c->k, carrier_at(c, z), tw_of(c, c->twY, c->NY, (unsigned)z), slice_carrier_field(c, z), cshift_at(c, z), 0, 1.f, start ? 1 : 0};
    sq_of(c, (unsigned)z, a.sq);
    a.pz_b = 0;
    if (c->range_car) {          // every wavefield of the range rides on its own carrier field
        a.pz = c->range_car + ((size_t)(z - c->range_car_z0) * c->range_car_B + c->sub_b0) * c->NX * c->NY;
        a.pz_b = c->NX;
    }
    c->tw_tick = (unsigned)z;
    const bool pf = a.pz != nullptr;
    DISPATCH_N(c->NY, {
        const dim3 grid(rows_grid<N_>(c, B, c->NX));
        const dim3 blk(BDOF_THREADS);
        if (z == 0 || start) {
            if (pf) {
                if (tstore) BDOF_LAUNCH(ps, (k_row_fwd<N_, true, true, true>), grid, blk, 0, c->sub_stream, a);
                else BDOF_LAUNCH(ps, (k_row_fwd<N_, true, false, true>), grid, blk, 0, c->sub_stream, a);
            } else {
                if (tstore) BDOF_LAUNCH(ps, (k_row_fwd<N_, true, true>), grid, blk, 0, c->sub_stream, a);
                else BDOF_LAUNCH(ps, (k_row_fwd<N_, true, false>), grid, blk, 0, c->sub_stream, a);
            }
        } else if (pf) {
            if (tstore) BDOF_LAUNCH(ps, (k_row_fwd<N_, false, true, true>), grid, blk, 0, c->sub_stream, a);
            else BDOF_LAUNCH(ps, (k_row_fwd<N_, false, false, true>), grid, blk, 0, c->sub_stream, a);
        } else {
            if (tstore) BDOF_LAUNCH(ps, (k_row_fwd<N_, false, true>), grid, blk, 0, c->sub_stream, a);
            else BDOF_LAUNCH(ps, (k_row_fwd<N_, false, false>), grid, blk, 0, c->sub_stream, a);
        }
    });
}

// A_z^-1 (tape-free adjoint): scattered part of phi_z (L1 hybrid, or real space) -> R eps(psi_z) in L2
static void launch_row_unmod(bdof_ctx* c, int B, int z, const cf* in, cf* out, bool real_in, float in_scale) {
    ProfScope ps(c, BDOF_K_ROW_FWD, true);
    RowFwdArgs a{sub_field(c, in), c->probe, sub_field(c, out), nullptr, sub_obj(c), B, c->NX, z, c->k, carrier_at(c, z), tw_of(c, c->twY, c->NY, (unsigned)z),
                 slice_carrier_field(c, z), cshift_at(c, z), real_in ? 1 : 0, in_scale, 0};
    a.pz_b = 0;
    sq_of(c, (unsigned)z, a.sq);
    c->tw_tick = (unsigned)z;
    DISPATCH_N(c->NY, {
        const dim3 grid(rows_grid<N_>(c, B, c->NX));
        const dim3 blk(BDOF_THREADS);
        if (a.pz) BDOF_LAUNCH(ps, (k_row_fwd<N_, false, true, true, true>), grid, blk, 0, c->sub_stream, a);
        else BDOF_LAUNCH(ps, (k_row_fwd<N_, false, true, false, true>), grid, blk, 0, c->sub_stream, a);
    });
}

// B: L2 -> L1
// key: the slice whose copy of the dithered constants the launch takes (tw_of); -1 = that of the last A / A' launch.  The step
// between slices z-1 and z runs with copy z-1 in the forward sweep (it follows A_{z-1}); its adjoint, which follows A'_z, is
// handed z-1 explicitly, so that the adjoint step is the transpose of the very operator the forward sweep applied.
static void launch_row_prop(bdof_ctx* c, int B, const cf* in, cf* out, const cf* h, float scale, int conj_h, int key = -1) {
    ProfScope ps(c, BDOF_K_COL_PROP, true);
    const unsigned tick = key >= 0 ? (unsigned)key : c->tw_tick;
    if (h == c->hs) {
        if (c->hs_override) h = c->hs_override;
        else if (c->hs_copies > 0) h = c->hs_d + (size_t)(tick % (unsigned)c->hs_copies) * c->NX * c->NY;      // the slice's dithered copy
    }
    RowPropArgs a{sub_field(c, in), sub_field(c, out), h, B, c->NY, scale, conj_h, tw_of(c, c->twX, c->NX, tick)};
    sq_of(c, tick, a.sq);
    DISPATCH_N(c->NX, {
        // the adjoint step runs the instance with exact transform constants (bdof_fft.h: that is where the gradient's error is made)
        if (conj_h ? BDOF_EX_ADJ : BDOF_EX_FWD_B) BDOF_LAUNCH(ps, (k_row_prop<N_, true>), dim3(rows_grid<N_>(c, B, c->NY)), dim3(BDOF_THREADS), 0, c->sub_stream, a);
        else BDOF_LAUNCH(ps, (k_row_prop<N_, false>), dim3(rows_grid<N_>(c, B, c->NY)), dim3(BDOF_THREADS), 0, c->sub_stream, a);
    });
}

// A'_z: L1 (g) + phi tape -> L2 (g)
// hist: 0 = `tape` is the phi tape of slice z; 1 = `tape` is psi_hat_z of the history tape; 2 = slice 0 of the history mode
struct GradTarget {            // where A'_z leaves the gradient rows / G(psi_z): the ctx's own buffers unless a range sweep says otherwise
    float2* grot = nullptr;    // [B][S_][NX][NY]
    int S_ = 0, z_ = 0;
    cf* gpsi = nullptr;        // real-space G(psi_z), [B][NX][NY]
};
static void launch_row_bwd(bdof_ctx* c, int B, int z, const cf* gin, const cf* tape, cf* gout, int hist = 0, float tape_scale = 1.f,
                           const GradTarget* gt = nullptr) {
    ProfScope ps(c, BDOF_K_ROW_BWD, true);
    float2* grot = gt && gt->grot ? gt->grot + (size_t)c->sub_b0 * gt->S_ * c->NX * c->NY : c->grot + (size_t)c->sub_b0 * c->S * c->NX * c->NY;
    RowBwdArgs a{sub_field(c, gin), hist == 2 ? c->probe : sub_field(c, tape), sub_field(c, gout),
                 grot, sub_obj(c), B, c->NX, z, c->k, carrier_at(c, z), tw_of(c, c->twY, c->NY, (unsigned)z),
                 slice_carrier_field(c, z), adj_carrier_at(c, c->S - 1 - z), cshift_at(c, z), carrier_phi_at(c, z), tape_scale,
                 gt && gt->gpsi ? sub_field(c, gt->gpsi) : (z == 0 && c->gpsi0 ? c->gpsi0 + (size_t)c->sub_b0 * c->NX * c->NY : nullptr),
                 gt && gt->grot ? gt->S_ : c->S, gt && gt->grot ? gt->z_ : z};
    sq_of(c, (unsigned)z, a.sq);
    c->tw_tick = (unsigned)z;
    const bool pf = a.pz != nullptr;
    DISPATCH_N(c->NY, {
        const dim3 grid(rows_grid<N_>(c, B, c->NX));
        const dim3 blk(BDOF_THREADS);
        if (a.ac.gcar) {         // far field + plane-wave carrier (never together with a carrier field)
            if (hist == 0) BDOF_LAUNCH(ps, (k_row_bwd<N_, 0, false, true>), grid, blk, 0, c->sub_stream, a);
            else if (hist == 1) BDOF_LAUNCH(ps, (k_row_bwd<N_, 1, false, true>), grid, blk, 0, c->sub_stream, a);
            else if (hist == 2) BDOF_LAUNCH(ps, (k_row_bwd<N_, 2, false, true>), grid, blk, 0, c->sub_stream, a);
            else BDOF_LAUNCH(ps, (k_row_bwd<N_, 3, false, true>), grid, blk, 0, c->sub_stream, a);
        } else if (pf) {
            if (hist == 0) BDOF_LAUNCH(ps, (k_row_bwd<N_, 0, true>), grid, blk, 0, c->sub_stream, a);
            else if (hist == 1) BDOF_LAUNCH(ps, (k_row_bwd<N_, 1, true>), grid, blk, 0, c->sub_stream, a);
            else if (hist == 2) BDOF_LAUNCH(ps, (k_row_bwd<N_, 2, true>), grid, blk, 0, c->sub_stream, a);
            else BDOF_LAUNCH(ps, (k_row_bwd<N_, 3, true>), grid, blk, 0, c->sub_stream, a);
        } else {
            if (hist == 0) BDOF_LAUNCH(ps, (k_row_bwd<N_, 0>), grid, blk, 0, c->sub_stream, a);
            else if (hist == 1) BDOF_LAUNCH(ps, (k_row_bwd<N_, 1>), grid, blk, 0, c->sub_stream, a);
            else if (hist == 2) BDOF_LAUNCH(ps, (k_row_bwd<N_, 2>), grid, blk, 0, c->sub_stream, a);
            else BDOF_LAUNCH(ps, (k_row_bwd<N_, 3>), grid, blk, 0, c->sub_stream, a);
        }
    });
}

// Real-space detector on L1 rows.  Returns the grid (= number of partial sums when meas != null).
// pf64 / pscale: a float64 carrier field of the caller's own and its complex factor (the real-space propagator's planes and
// renormalisation); by default the ctx's pdet64 goes with pfield == pdet
static const double2 kOne = {1.0, 0.0};
static int launch_loss_real(bdof_ctx* c, int B, const cf* in, cf* out_hyb, bool tstore, cf* out_wave, const float* meas,
                            float in_scale, float out_scale, float seed_scale, cf carrier, const cf* pfield = nullptr,
                            const float* dref_override = nullptr, const double2* pf64 = nullptr, double2 pscale = kOne) {
    ProfScope ps(c, BDOF_K_LOSS);
    LossArgs a{sub_field(c, in), sub_field(c, out_hyb), sub_field(c, out_wave), sub_field(c, meas), c->partial + 2 * c->sub_part, B, c->NX,
               in_scale, out_scale, seed_scale, carrier, c->twY, pfield, c->meas_dev, nullptr, nullptr, make_double2(0.0, 0.0), make_double2(0.0, 0.0),
               c->meas_dev ? (dref_override ? *dref_override : meas_dref(c)) : 0.f,
               pf64 ? pf64 : (pfield && pfield == c->pdet ? c->pdet64 : nullptr), pscale};
    int grid = 0;
    DISPATCH_N(c->NY, {
        grid = rows_grid<N_>(c, B, c->NX);
        if (tstore) hipLaunchKernelGGL((k_row_loss<N_, false, true>), dim3(grid), dim3(BDOF_THREADS), 0, c->sub_stream, a);
        else hipLaunchKernelGGL((k_row_loss<N_, false, false>), dim3(grid), dim3(BDOF_THREADS), 0, c->sub_stream, a);
    });
    return grid;
}

// Far-field detector on L2 rows; the seed goes back transposed into L1.
static int launch_loss_far(bdof_ctx* c, int B, const cf* in, cf* out_hyb, cf* out_wave, const float* meas, float in_scale,
                           float out_scale, float seed_scale, const cf* pfield = nullptr, const double2* pf64 = nullptr, double2 pscale = kOne) {
    ProfScope ps(c, BDOF_K_LOSS);
    const bool gc = meas && out_hyb && use_adj_carrier(c);
    LossArgs a{sub_field(c, in), sub_field(c, out_hyb), sub_field(c, out_wave), sub_field(c, meas), c->partial + 2 * c->sub_part, B, c->NY,
               in_scale, out_scale, seed_scale, carrier_det(c), c->twX, pfield, 0,
               gc ? c->gcar + c->sub_b0 : nullptr, gc ? c->gt0 + c->sub_b0 : nullptr,
               d2(carrier_end(c) * ((double)c->NX * (double)c->NY)), d2(carrier_end(c)), 0.f,
               pf64 ? pf64 : (pfield && pfield == c->pdetT ? c->pdetT64 : nullptr), pscale};
    int grid = 0;
    DISPATCH_N(c->NX, {
        grid = rows_grid<N_>(c, B, c->NY);
        hipLaunchKernelGGL((k_row_loss<N_, true, true>), dim3(grid), dim3(BDOF_THREADS), 0, c->sub_stream, a);
    });
    return grid;
}

// ---- the forward sweep ---------------------------------------------------------------------------
// bufA is the L2-type scratch (row kernels' transposed output), bufB the L1-type scratch.
// On return the field the detector step starts from is
//   DET_NONE, numpy_skip_last : bufA = R phi_{S-1} in L1 order (plain store), un-normalised
//   DET_NONE, tf_all          : bufB = psi_hat_S (L1)
//   DET_NEAR                  : bufB = d_hat (L1)    (tf_all: one step with the combined transfer function)
//   DET_FAR                   : bufA = R phi_{S-1} in L2 order, un-normalised (|fft2| is unchanged by the
//                               unit-modulus transfer function, so tf_all needs no extra step here)
enum { TAPE_NONE = 0, TAPE_HISTORY = 1, TAPE_PHI = 2, TAPE_LAST = 3 };
// TAPE_LAST (tape-free adjoint): only the real-space phi_{S-1} is kept (tape slot 0); the adjoint sweep marches it back.
// TAPE_HISTORY keeps psi_hat_{z+1} (the transfer-function step's output) per slice: probe_array of np_funcs.py:43.
// TAPE_PHI keeps the real-space phi_z written by A_z: what the adjoint needs, without a third transform in A'_z.
static void forward_sweep(bdof_ctx* c, const Group* groups, int ngroups, int tape_mode) {
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    const bool tf_all = c->variant == BDOF_VARIANT_TF_ALL;
    for (int z = 0; z < c->S; ++z) {
        const cf* in = nullptr;
        if (z > 0) in = tape_mode == TAPE_HISTORY ? c->tape + (size_t)(z - 1) * fld : c->bufB;
        const bool last = z == c->S - 1;
        cf* phi = tape_mode == TAPE_PHI ? c->tape + (size_t)z * fld : (tape_mode == TAPE_LAST && last ? c->tape : nullptr);
        for (int gi = 0; gi < ngroups; ++gi) {
            const int B = groups[gi].B;
            use_group(c, groups[gi]);
            if (!last) {
                launch_row_fwd(c, B, z, in, c->bufA, true, phi);
                launch_row_prop(c, B, c->bufA, tape_mode == TAPE_HISTORY ? c->tape + (size_t)z * fld : c->bufB, c->hs, 1.f, 0);
            } else if (c->det_mode == BDOF_DET_NONE && !tf_all) {
                launch_row_fwd(c, B, z, in, c->bufA, false, phi);
            } else {
                launch_row_fwd(c, B, z, in, c->bufA, true, phi);
                if (c->det_mode == BDOF_DET_NONE) launch_row_prop(c, B, c->bufA, c->bufB, c->hs, 1.f, 0);
                else if (c->det_mode == BDOF_DET_NEAR) launch_row_prop(c, B, c->bufA, c->bufB, tf_all ? c->hcomb : c->hdet, 1.f, 0);
            }
        }
    }
}

// (Re)build the modulation table c - 1 = exp(i k delta - k beta) - 1 of the object rows when the object or k changed.
// room for n modulation factors and, if the mean-refraction carrier is in use, for the per-workgroup sums of a pass
static int modulation_room(bdof_ctx* c, size_t n, bool mean) {
    if (n > c->mod_cap) {
        if (c->mod) (void)hipFree(c->mod);
        c->mod = nullptr; c->mod_cap = 0;
        HIPC(c, hipMalloc((void**)&c->mod, sizeof(float2) * n));
        c->mod_cap = n;
    }
    if (mean && c->ncb < c->ncu * 16 + 1) {
        if (c->cbar_dev) (void)hipFree(c->cbar_dev);
        c->cbar_dev = nullptr; c->ncb = 0;
        HIPC(c, hipMalloc((void**)&c->cbar_dev, sizeof(double2) * (size_t)(c->ncu * 16 + 1)));
        c->ncb = c->ncu * 16 + 1;
    }
    return 0;
}
// after a pass that left c - 1 in c->mod (and `grid` partial sums in cbar_dev if mean): mean to the host, object bound
static int modulation_done(bdof_ctx* c, size_t n, int grid, bool mean) {
    HIPC(c, hipGetLastError());
    std::complex<double> cb(0.0, 0.0);
    if (mean) {
        // cbar is needed by the HOST (it forms the carrier scalars in float64): one small read-back per object update
        hipLaunchKernelGGL(k_sum_mean, dim3(1), dim3(256), 0, c->stream, c->cbar_dev, grid, 1.0 / (double)n, c->cbar_dev + grid);
        double2 m;
        HIPC(c, hipMemcpyAsync(&m, c->cbar_dev + grid, sizeof(double2), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        cb = std::complex<double>(m.x, m.y);
    }
    if (cb != c->cbm1) { c->cbm1 = cb; c->res_dirty = true; }
    c->obj.vol = c->mod;
    c->mod_dirty = false;
    return 0;
}

static int ensure_modulation(bdof_ctx* c) {
    if (!c->mod_dirty) return 0;
    if (!c->obj_src) return fail(c, BDOF_ERR_STATE, "the object was bound as modulation factors (bdof_set_object_bilinear): bind it again for this propagator");
    const size_t n = c->obj_rows * (size_t)c->obj.volNY;
    const bool mean = want_cbar(c);
    int r = modulation_room(c, n, mean);
    if (r) return r;
    size_t need = (n + 255) / 256;
    int grid = need < (size_t)c->ncu * 16 ? (int)need : c->ncu * 16;
    hipLaunchKernelGGL(k_modulation_table, dim3(grid), dim3(256), 0, c->stream, c->obj_src, c->mod, n, c->k, mean ? c->cbar_dev : nullptr);
    return modulation_done(c, n, grid, mean);
}

static int ensure_modulation_k(bdof_ctx* c, float k) {
    // the conv propagator's k uses numpy's pi, the FFT path's the reference's literal (quirk Q1): one table per k
    if (c->k != k) { c->k = k; c->mod_dirty = true; }
    return ensure_modulation(c);
}

// =================================================================================================
// Generic-size engine (rocFFT).  Fields are real-space [b][x][y] in bufA; the tape holds phi_z.
// =================================================================================================
static bool g_rocfft_ready = false;

#define RFC(c, call)                                                                                \
    do {                                                                                            \
        rocfft_status s_ = (call);                                                                  \
        if (s_ != rocfft_status_success) return fail((c), BDOF_ERR_STATE, std::string(#call) + ": rocfft status " + std::to_string((int)s_)); \
    } while (0)

static int generic_plans(bdof_ctx* c, int B, rocfft_plan* fwd, rocfft_plan* inv, bool dbl = false) {
    if (!g_rocfft_ready) { RFC(c, rocfft_setup()); g_rocfft_ready = true; }
    auto& plans = dbl ? c->gplans64 : c->gplans;
    const rocfft_precision prec = dbl ? rocfft_precision_double : rocfft_precision_single;
    auto it = plans.find(B);
    if (it == plans.end()) {
        const size_t lengths[2] = {(size_t)c->NY, (size_t)c->NX};      // fastest dimension first
        rocfft_plan pf = nullptr, pi = nullptr;
        RFC(c, rocfft_plan_create(&pf, rocfft_placement_inplace, rocfft_transform_type_complex_forward, prec, 2, lengths, (size_t)B, nullptr));
        RFC(c, rocfft_plan_create(&pi, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, prec, 2, lengths, (size_t)B, nullptr));
        size_t w1 = 0, w2 = 0;
        RFC(c, rocfft_plan_get_work_buffer_size(pf, &w1));
        RFC(c, rocfft_plan_get_work_buffer_size(pi, &w2));
        const size_t need = std::max(w1, w2);
        if (need > c->gwork_sz) {
            HIPC(c, hipStreamSynchronize(c->stream));
            if (c->gwork) (void)hipFree(c->gwork);
            c->gwork = nullptr;
            HIPC(c, hipMalloc(&c->gwork, need));
            c->gwork_sz = need;
        }
        it = plans.emplace(B, std::make_pair(pf, pi)).first;
    }
    if (!c->ginfo) {
        RFC(c, rocfft_execution_info_create(&c->ginfo));
        RFC(c, rocfft_execution_info_set_stream(c->ginfo, (void*)c->stream));
    }
    if (c->gwork_sz) RFC(c, rocfft_execution_info_set_work_buffer(c->ginfo, c->gwork, c->gwork_sz));
    *fwd = it->second.first;
    *inv = it->second.second;
    return 0;
}

// the transfer function of the step after slice z: its dithered float32 copy where the caller handed the float64 table over
static const cf* hs_slice(const bdof_ctx* c, int z) {
    return c->hs_copies > 0 ? c->hs_d + (size_t)((unsigned)z % (unsigned)c->hs_copies) * c->NX * c->NY : c->hs;
}

static int g_elem_grid(const bdof_ctx* c, size_t n) { return (int)std::min<size_t>((n + 255) / 256, (size_t)c->ncu * 16); }

// one transfer-function step in place: field <- F^-1' ( h * F field )
static int generic_prop(bdof_ctx* c, int B, rocfft_plan pf, rocfft_plan pi, cf* field, const cf* h, int conj_h) {
    ProfScope ps(c, BDOF_K_COL_PROP);
    void* buf[1] = {field};
    RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo));
    const size_t n = (size_t)B * c->NX * c->NY;
    hipLaunchKernelGGL(k_g_hmul, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, field, h, B, c->NX, c->NY, conj_h);
    RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo));
    return 0;
}

// Forward sweep; leaves the field the detector starts from in bufA (eps part) and returns its carrier.
static int generic_forward_sweep(bdof_ctx* c, int B, bool tape, rocfft_plan pf, rocfft_plan pi, std::complex<double>* carrier_out) {
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    const size_t n = (size_t)B * c->NX * c->NY;
    const bool tf_all = c->variant == BDOF_VARIANT_TF_ALL;
    std::complex<double> a = c->a0;
    int r;
    for (int z = 0; z < c->S; ++z) {
        {
            ProfScope ps(c, BDOF_K_ROW_FWD);
            GModArgs m{c->bufA, z == 0 ? c->probe : nullptr, tape ? c->tape + (size_t)z * fld : nullptr, c->obj, B, c->NX, c->NY, z,
                       make_float2((float)a.real(), (float)a.imag()), c->pstack ? c->pstack + (size_t)z * c->NX * c->NY : nullptr,
                       cfl(a * c->cbm1)};
            hipLaunchKernelGGL(k_g_modulate, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, m);
            a *= 1.0 + c->cbm1;                  // mean-refraction carrier: phi_z rides on cbar a_z
        }
        const bool last = z == c->S - 1;
        if (!last || (tf_all && c->det_mode != BDOF_DET_FAR)) {
            if ((r = generic_prop(c, B, pf, pi, c->bufA, hs_slice(c, z), 0))) return r;      // the slice's dithered copy of H
            a *= c->h00;
        }
    }
    if (c->det_mode == BDOF_DET_NEAR) {
        if ((r = generic_prop(c, B, pf, pi, c->bufA, c->hdet, 0))) return r;
        a *= c->hdet00;
    } else if (c->det_mode == BDOF_DET_FAR) {
        void* buf[1] = {c->bufA};
        RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo));
        a *= (double)c->NX * (double)c->NY;
    }
    *carrier_out = a;
    return 0;
}

static int generic_forward(bdof_ctx* c, int B, void* out_wave, bool keep_tape) {
    rocfft_plan pf, pi;
    int r = generic_plans(c, B, &pf, &pi);
    if (r) return r;
    std::complex<double> a;
    if ((r = generic_forward_sweep(c, B, keep_tape, pf, pi, &a))) return r;
    if (out_wave) {
        const size_t n = (size_t)B * c->NX * c->NY;
        GLossArgs la{c->bufA, (cf*)out_wave, nullptr, c->partial, B, c->NX, c->NY, c->det_mode == BDOF_DET_FAR,
                     make_float2((float)a.real(), (float)a.imag()), 0.f, c->pdet, 0, 0.f, nullptr, nullptr, make_double2(0.0, 0.0), make_double2(0.0, 0.0), nullptr,
                     nullptr, 0.0, make_double2(0.0, 0.0)};
        hipLaunchKernelGGL(k_g_loss, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, la);
    }
    return 0;
}

static int generic_loss_grad(bdof_ctx* c, int B, const float* meas, void* out_wave) {
    rocfft_plan pf, pi;
    int r = generic_plans(c, B, &pf, &pi);
    if (r) return r;
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    const size_t n = (size_t)B * c->NX * c->NY;
    const bool tf_all = c->variant == BDOF_VARIANT_TF_ALL;
    std::complex<double> a;
    if ((r = generic_forward_sweep(c, B, true, pf, pi, &a))) return r;
    const int egrid = g_elem_grid(c, n);
    const bool f64 = c->adj64;
    if (f64 && !c->have_h64) return fail(c, BDOF_ERR_STATE, "float64 adjoint: bdof_set_physics_f64 has not been called");
    {
        ProfScope ps(c, BDOF_K_LOSS);
        const bool gc = use_adj_carrier(c) && !f64;
        GLossArgs la{c->bufA, (cf*)out_wave, meas, c->partial, B, c->NX, c->NY, c->det_mode == BDOF_DET_FAR,
                     make_float2((float)a.real(), (float)a.imag()), 2.f / ((float)B * (float)c->NX * (float)c->NY), c->pdet, c->meas_dev,
                     c->meas_dev ? meas_dref(c) : 0.f, gc ? c->gcar : nullptr, gc ? c->gt0 : nullptr, d2(a), d2(carrier_end(c)), c->pdet ? c->pdet64 : nullptr,
                     f64 ? c->g64 : nullptr, c->meas_dev ? std::abs(c->a0) : 0.0, make_double2(0.0, 0.0)};
        hipLaunchKernelGGL(k_g_loss, dim3(egrid), dim3(256), 0, c->stream, la);
    }
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, egrid, 1.0 / ((double)B * c->NX * c->NY), c->loss_dev);
    if (f64) {
        // float64 adjoint sweep: g64 holds the seed G(d)
        rocfft_plan pf64, pi64;
        if ((r = generic_plans(c, B, &pf64, &pi64, true))) return r;
        void* b64[1] = {c->g64};
        auto prop64 = [&](const double2* h) -> int {
            ProfScope ps(c, BDOF_K_COL_PROP);
            RFC(c, rocfft_execute(pf64, b64, nullptr, c->ginfo));
            hipLaunchKernelGGL(k_g_hmul64, dim3(egrid), dim3(256), 0, c->stream, c->g64, h, B, c->NX, c->NY, 1);
            RFC(c, rocfft_execute(pi64, b64, nullptr, c->ginfo));
            return 0;
        };
        if (c->det_mode == BDOF_DET_NEAR) { if ((r = prop64(c->hdet64))) return r; }
        else if (c->det_mode == BDOF_DET_FAR) RFC(c, rocfft_execute(pi64, b64, nullptr, c->ginfo));
        for (int z = c->S - 1; z >= 0; --z) {
            const bool prop_after = z < c->S - 1 || (tf_all && c->det_mode != BDOF_DET_FAR);
            if (prop_after && (r = prop64(c->hs64))) return r;
            ProfScope ps(c, BDOF_K_ROW_BWD);
            GBwd64Args ba{c->g64, c->tape + (size_t)z * fld, c->grot, c->obj, B, c->NX, c->NY, z, (double)c->k,
                          d2(carrier_z(c, z) * (1.0 + c->cbm1)), c->pstack ? 1 : 0};
            hipLaunchKernelGGL(k_g_bwd64, dim3(egrid), dim3(256), 0, c->stream, ba);
        }
        // G(psi_0) for bdof_probe_grad, where the float32 sweep leaves it
        hipLaunchKernelGGL(k_d_to_f, dim3(egrid), dim3(256), 0, c->stream, c->g64, c->bufA, B * c->NX, c->NY, 0);
        return 0;
    }
    // adjoint of the detector step: bufA now holds the seed G(d)
    void* buf[1] = {c->bufA};
    if (c->det_mode == BDOF_DET_NEAR) {
        if ((r = generic_prop(c, B, pf, pi, c->bufA, c->hdet, 1))) return r;
    } else if (c->det_mode == BDOF_DET_FAR) {
        RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo));            // F^H = unnormalised inverse
    }
    for (int z = c->S - 1; z >= 0; --z) {
        const bool prop_after = z < c->S - 1 || (tf_all && c->det_mode != BDOF_DET_FAR);
        if (prop_after && (r = generic_prop(c, B, pf, pi, c->bufA, hs_slice(c, z), 1))) return r;      // conj of the forward step's copy
        ProfScope ps(c, BDOF_K_ROW_BWD);
        GBwdArgs ba{c->bufA, c->tape + (size_t)z * fld, c->grot, c->obj, B, c->NX, c->NY, z, c->k, carrier_phi_at(c, z), c->pstack ? 1 : 0,
                    adj_carrier_at(c, c->S - 1 - z)};
        hipLaunchKernelGGL(k_g_bwd, dim3(egrid), dim3(256), 0, c->stream, ba);
    }
    return 0;
}

// ---- LDS-resident engine ---------------------------------------------------------------------------
template <int N, int T, int WPE> static int resident_launch_t(bdof_ctx* c, const ResArgs& a, int grid, int* waves) {
    *waves = T / 64;
    const size_t lds = sizeof(cf) * ((size_t)N * (N | 1) + 2 * N) + sizeof(long long) * 3 * N;
    static bool attr_set[BDOF_MAX_DEVICES] = {};      // the attribute is per device (a process may hold ctxs on several)
    if (!attr_set[c->device % BDOF_MAX_DEVICES]) {
        HIPC(c, hipFuncSetAttribute((const void*)k_resident<N, T, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[c->device % BDOF_MAX_DEVICES] = true;
    }
    hipLaunchKernelGGL((k_resident<N, T, WPE>), dim3(grid), dim3(T), lds, c->stream, a);
    return 0;
}
template <int N> static int resident_launch(bdof_ctx* c, const ResArgs& a, int grid, int* waves) {
    return resident_launch_t<N, ResPlan<N>::T, ResPlan<N>::WPE>(c, a, grid, waves);
}

// One workgroup per wavefield: against the streaming kernels (sizes with a fused plan) the resident engine wins once the
// batch fills a good part of the chip (measured at 64^2 / 128^2: 400 / 100 wavefields 4.6x / 1.5x faster, 25 slower);
// sizes without a fused plan always take it (the alternative is the unfused rocFFT engine).
static bool use_resident(const bdof_ctx* c, int B) {
    if (!c->resident || c->recompute) return false;
    // far field + plane-wave carrier needs the adjoint carrier (AdjCarrier), which the resident kernel does not carry: the
    // streaming / generic engines take that case (plane-wave full-field at a resident-plan size)
    if (c->det_mode == BDOF_DET_FAR && !c->pstack && std::abs(c->a0) > 0.0 && !std::getenv("BDOF_NO_ADJ_CARRIER")) return false;
    return c->generic || c->res_always || B * 4 >= c->ncu;
}

static int resident_run(bdof_ctx* c, int B, const float* meas, void* out_wave, bool do_grad) {
    if (c->res_dirty) {
        std::vector<cf> car(2 * (size_t)c->S);
        for (int z = 0; z < c->S; ++z) { car[z] = carrier_at(c, z); car[c->S + z] = cshift_at(c, z); }
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, hipMemcpy(c->res_carrier, car.data(), sizeof(cf) * car.size(), hipMemcpyHostToDevice));
        c->res_dirty = false;
    }
    ProfScope ps(c, BDOF_K_ROW_FWD);
    const bool grad = do_grad && meas;
    const bool hdith = c->hs_copies > 0 && c->hsT_d;
    ResArgs a{c->probe, hdith ? c->hsT_d : c->hsT, c->hdetT, grad ? c->tape : nullptr, (size_t)c->Bmax * c->NX * c->NY, c->grot, c->obj, c->res_carrier,
              carrier_det(c), c->pstack, c->pdet, meas, (cf*)out_wave, c->partial, c->twR, B, c->S, c->det_mode,
              c->variant == BDOF_VARIANT_TF_ALL ? 1 : 0, grad ? 1 : 0, c->k, 2.f / ((float)B * (float)c->NX * (float)c->NY), c->meas_dev,
              hdith ? c->hs_copies : 0, c->meas_dev ? meas_dref(c) : 0.f, grad ? c->gpsi0 : nullptr, c->pdet ? c->pdet64 : nullptr};
    const int grid = B < c->npartial ? B : c->npartial;
    int r = 0, waves = 1;
    switch (c->NX) {
        case 32: r = resident_launch<32>(c, a, grid, &waves); break;
        case 36: r = resident_launch<36>(c, a, grid, &waves); break;
        case 48: r = resident_launch<48>(c, a, grid, &waves); break;
        case 64: r = resident_launch<64>(c, a, grid, &waves); break;
        case 72: r = resident_launch<72>(c, a, grid, &waves); break;
        case 80: r = resident_launch<80>(c, a, grid, &waves); break;
        case 96: r = resident_launch<96>(c, a, grid, &waves); break;
        case 128: r = resident_launch<128>(c, a, grid, &waves); break;
        default: return fail(c, BDOF_ERR_SIZE, "no resident plan for this size");
    }
    if (r) return r;
    if (meas)
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, grid * waves, 1.0 / ((double)B * c->NX * c->NY),
                           c->loss_dev);
    HIPC(c, hipGetLastError());
    return 0;
}

static int check_ready(bdof_ctx* c, int B) {
    if (!c) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (!c->have_physics) return fail(c, BDOF_ERR_STATE, "bdof_set_physics has not been called");
    if (!c->have_probe) return fail(c, BDOF_ERR_STATE, "bdof_set_probe has not been called");
    if (!c->obj_src && !c->obj_bound_mod) return fail(c, BDOF_ERR_STATE, "bdof_set_object has not been called");
    if (B < 1 || B > c->Bmax) return fail(c, BDOF_ERR_ARG, "batch size outside [1, Bmax]");
    return 0;
}

extern "C" {

int bdof_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bdof_device_pci_bus_id(int device, char* out, int len) {
    if (!out || len < 16) return BDOF_ERR_ARG;
    out[0] = 0;
    return (int)hipDeviceGetPCIBusId(out, len, device);
}

// stream-ordered time stamps on the ctx stream: bdof_timer_mark(slot) now, bdof_timer_elapsed(a, b) after a sync
int bdof_timer_mark(bdof_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= BDOF_N_TIMERS) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    if (!c->timer[slot]) HIPC(c, hipEventCreate(&c->timer[slot]));
    HIPC(c, hipEventRecord(c->timer[slot], c->stream));
    return 0;
}
int bdof_timer_elapsed(bdof_ctx* c, int slot_a, int slot_b, double* ms) {
    if (!c || !ms || slot_a < 0 || slot_a >= BDOF_N_TIMERS || slot_b < 0 || slot_b >= BDOF_N_TIMERS) return BDOF_ERR_ARG;
    if (!c->timer[slot_a] || !c->timer[slot_b]) return fail(c, BDOF_ERR_STATE, "bdof_timer_elapsed: slot never marked");
    HIPC(c, hipEventSynchronize(c->timer[slot_b]));
    float f = 0.f;
    HIPC(c, hipEventElapsedTime(&f, c->timer[slot_a], c->timer[slot_b]));
    *ms = (double)f;
    return 0;
}

int bdof_ctx_create(bdof_ctx** out, int device, void* stream) {
    if (!out) return BDOF_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return e != hipSuccess ? (int)e : (int)hipErrorNoDevice;
    if (device < 0 || device >= n) return BDOF_ERR_ARG;
    bdof_ctx* c = new bdof_ctx();
    c->device = device;
    e = hipSetDevice(device);
    if (e != hipSuccess) { delete c; return (int)e; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->ncu = prop.multiProcessorCount;
    if (stream) { c->stream = (hipStream_t)stream; }
    else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return (int)e; }
        c->own_stream = true;
    }
    (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    for (int i = 0; i < BDOF_MAX_GROUPS - 1; ++i) (void)hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
    c->sub_stream = c->stream;
    if (const char* e = std::getenv("BDOF_STREAMS")) c->n_streams = std::atoi(e);
    *out = c;
    return 0;
}

static void free_generic(bdof_ctx* c) {
    for (auto& kv : c->gplans) { (void)rocfft_plan_destroy(kv.second.first); (void)rocfft_plan_destroy(kv.second.second); }
    c->gplans.clear();
    for (auto& kv : c->gplans64) { (void)rocfft_plan_destroy(kv.second.first); (void)rocfft_plan_destroy(kv.second.second); }
    c->gplans64.clear();
    for (auto& kv : c->fplans) { (void)rocfft_plan_destroy(kv.second.first); (void)rocfft_plan_destroy(kv.second.second); }
    c->fplans.clear();
    for (double2** q : {&c->g64, &c->hs64, &c->hdet64}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    c->have_h64 = false;
    if (c->ginfo) { (void)rocfft_execution_info_destroy(c->ginfo); c->ginfo = nullptr; }
    if (c->gwork) { (void)hipFree(c->gwork); c->gwork = nullptr; c->gwork_sz = 0; }
}

static void free_workspace(bdof_ctx* c) {
    free_generic(c);
    if (c->hs_d) { (void)hipFree(c->hs_d); c->hs_d = nullptr; c->hs_copies = 0; }
    if (c->hsT_d) { (void)hipFree(c->hsT_d); c->hsT_d = nullptr; }
    for (double2** q : {&c->c64_probe, &c->c64_khat, &c->c64_psi, &c->c64_q, &c->c64_big, &c->c64_tape, &c->c64_scal, &c->c64_part, &c->c64_h, &c->c64_hdet}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    c->c64_ks = c->c64_B = 0;
    void* ptrs[] = {c->cstack, c->cdet64, c->pdet64, c->pdetT64, c->pstack, c->pdet, c->pdetT, c->hsT, c->hdetT, c->twR, c->res_carrier, c->bufC, c->conv_scal, c->taps_dev, c->twY, c->twX, c->hs, c->hdet, c->hcomb, c->probe, c->bufA, c->bufB, c->tape, c->grot, c->gcar, c->gt0, c->gpsi0, c->partial, c->loss_dev};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    c->hsT = c->hdetT = c->twR = c->res_carrier = nullptr;
    c->pstack = c->pdet = c->pdetT = nullptr;
    c->pdet64 = c->pdetT64 = nullptr;
    c->resident = false;
    c->bufC = c->conv_scal = nullptr;
    c->cstack = nullptr; c->cdet64 = nullptr;
    c->taps_dev = nullptr;
    c->taps_copies = 1;
    c->have_conv = false;
    c->twY = c->twX = c->hs = c->hdet = c->hcomb = c->probe = c->bufA = c->bufB = c->tape = nullptr;
    c->grot = nullptr;
    c->gcar = c->gt0 = nullptr;
    c->gpsi0 = nullptr; c->gpsi_src = nullptr;
    c->partial = c->loss_dev = nullptr;
}

void bdof_ctx_destroy(bdof_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_workspace(c);
    if (c->heavy) (void)hipFree(c->heavy);
    if (c->winpad) (void)hipFree(c->winpad);
    if (c->win_angle) (void)hipFree(c->win_angle);
    if (c->mod) (void)hipFree(c->mod);
    if (c->cbar_dev) (void)hipFree(c->cbar_dev);
    for (auto& e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto& e : c->timer) if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : c->side_all) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->ev_aux_fork) (void)hipEventDestroy(c->ev_aux_fork);
    if (c->ev_aux_join) (void)hipEventDestroy(c->ev_aux_join);
    if (c->ginfo_aux) (void)rocfft_execution_info_destroy(c->ginfo_aux);
    if (c->gwork_aux) (void)hipFree(c->gwork_aux);
    if (c->car_scratch) (void)hipFree(c->car_scratch);
    for (int i = 0; i < BDOF_MAX_GROUPS - 1; ++i)
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* bdof_last_error(const bdof_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int bdof_sync(bdof_ctx* c) {
    if (!c) return BDOF_ERR_ARG;
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

// exp(-2 pi i j / N) in float32, rounded so that the MODULUS is as close to one as float32 allows: among the roundings
// of (cos, sin) up or down (each candidate within one ulp of the true value) the pair with the smallest | |w|^2 - 1 |.
// Plain round-to-nearest leaves every twiddle with a modulus error of +-3e-8; the factors do not average out along the
// paths of a transform, a chain of hundreds of transforms then drifts in energy.
static cf unit_twiddle(double c, double s) {
    const float c0 = (float)c, s0 = (float)s;
    const float cc[3] = {c0, std::nextafterf(c0, 2.f), std::nextafterf(c0, -2.f)};
    const float ss[3] = {s0, std::nextafterf(s0, 2.f), std::nextafterf(s0, -2.f)};
    float bc = c0, bs = s0;
    double best = std::fabs((double)c0 * c0 + (double)s0 * s0 - 1.0);
    for (float x : cc)
        for (float y : ss) {
            if (std::fabs((double)x - c) > std::fabs((double)std::nextafterf(c0, 2.f) - (double)c0)) continue;      // > 1 ulp away
            if (std::fabs((double)y - s) > std::fabs((double)std::nextafterf(s0, 2.f) - (double)s0)) continue;
            const double e = std::fabs((double)x * x + (double)y * y - 1.0);
            if (e < best) { best = e; bc = x; bs = y; }
        }
    return make_float2(bc, bs);
}

// D copies (D <= 1: one copy, nearest rounding with the modulus kept closest to one) of the twiddle table of length N; per copy
// N hi values, then N lo values (the float32 rounding error of the hi ones), (re, im) pairs
static void fill_twiddle_tables(int N, int D, cf* t) {
    const int copies = D > 1 ? D : 1;
    for (int d = 0; d < copies; ++d)
        for (int j = 0; j < N; ++j) {
            const double ang = -2.0 * M_PI * (double)j / (double)N;
            cf* td = t + (size_t)d * 2 * N;
            if (D > 1) {
                // golden-ratio phases: a different offset of the up / down pattern for every entry and component
                const double ph1 = std::fmod(0.6180339887498949 * (2 * j + 1), 1.0), ph2 = std::fmod(0.6180339887498949 * (2 * j + 2) + 0.37, 1.0);
                td[j] = make_float2(dither_pick(std::cos(ang), d, ph1), dither_pick(std::sin(ang), d, ph2));
            } else td[j] = unit_twiddle(std::cos(ang), std::sin(ang));
            td[(size_t)N + j] = make_float2((float)(std::cos(ang) - (double)td[j].x), (float)(std::sin(ang) - (double)td[j].y));
        }
}

int bdof_twiddle_tables(int N, int D, float* out) {
    if (N < 1 || D < 0 || D > 256 || !out) return BDOF_ERR_ARG;
    fill_twiddle_tables(N, D, (cf*)out);
    return 0;
}

static int upload_twiddle(bdof_ctx* c, int N, cf** dst, bool dither = true) {
    const int D = dither && c->tw_dither > 1 ? c->tw_dither : 1;
    std::vector<cf> t((size_t)N * 2 * D);
    fill_twiddle_tables(N, D, t.data());
    HIPC(c, hipMalloc((void**)dst, sizeof(cf) * t.size()));
    HIPC(c, hipMemcpyAsync(*dst, t.data(), sizeof(cf) * t.size(), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int bdof_configure(bdof_ctx* c, int NY, int NX, int S, int Bmax, int with_grad) {
    if (!c) return BDOF_ERR_ARG;
    if (NY < 1 || NX < 1 || S < 1 || Bmax < 1) return fail(c, BDOF_ERR_ARG, "NY, NX, S and Bmax must be >= 1");
    const bool generic = (with_grad & (2 | 64)) != 0 || !supported_n(NY) || !supported_n(NX);
    if (generic && (size_t)NY * NX > ((size_t)1 << 26)) return fail(c, BDOF_ERR_SIZE, "wavefield too large");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    free_workspace(c);
    c->NY = NY; c->NX = NX; c->S = S; c->Bmax = Bmax; c->with_grad = (with_grad & 1) != 0;
    c->generic = generic;
    // dithered transform constants (bdof_fft.h): 64 copies of each table by default, BDOF_TW_DITHER=0 for one plain table
    { const char* e = std::getenv("BDOF_TW_DITHER"); c->tw_dither = e ? std::max(0, std::min(256, atoi(e))) : 64; if (c->tw_dither == 1) c->tw_dither = 0; }
    c->recompute = (with_grad & 16) != 0 && !generic;      // the streaming engine's option; the others keep their tapes
    c->adj64 = (with_grad & 64) != 0 && (with_grad & 1) != 0;
    c->resident = (with_grad & (6 | 64)) == 0 && NX == NY && resident_supported(NX) && !std::getenv("BDOF_NO_RESIDENT");
    c->res_always = (with_grad & 8) != 0 || std::getenv("BDOF_FORCE_RESIDENT");
    c->res_dirty = true;
    c->have_physics = c->have_probe = c->tape_valid = false;
    int r;
    if (c->resident) {
        if ((r = upload_twiddle(c, NX, &c->twR, false))) return r;       // one launch runs all slices: one table
        HIPC(c, hipMalloc((void**)&c->hsT, sizeof(cf) * NX * NY));
        HIPC(c, hipMalloc((void**)&c->hdetT, sizeof(cf) * NX * NY));
        HIPC(c, hipMalloc((void**)&c->res_carrier, sizeof(cf) * 2 * (size_t)S));
    }
    if (!generic) {
        if ((r = upload_twiddle(c, NY, &c->twY))) return r;
        if ((r = upload_twiddle(c, NX, &c->twX))) return r;
    }
    const size_t fld = (size_t)Bmax * NX * NY;
    HIPC(c, hipMalloc((void**)&c->hs, sizeof(cf) * NX * NY));
    HIPC(c, hipMalloc((void**)&c->hdet, sizeof(cf) * NX * NY));
    HIPC(c, hipMalloc((void**)&c->hcomb, sizeof(cf) * NX * NY));
    HIPC(c, hipMalloc((void**)&c->probe, sizeof(cf) * NX * NY));
    HIPC(c, hipMalloc((void**)&c->bufA, sizeof(cf) * fld));
    HIPC(c, hipMalloc((void**)&c->bufB, sizeof(cf) * fld));
    if (c->with_grad) {
        // tape-free adjoint: phi_{S-1} (real space) + the two fields the marched-back wave alternates between
        const bool small_tape = c->recompute && !(c->resident && ((with_grad & 8) != 0 || std::getenv("BDOF_FORCE_RESIDENT")));
        c->recompute = small_tape;
        HIPC(c, hipMalloc((void**)&c->tape, sizeof(cf) * fld * (size_t)(small_tape ? std::min(S, 3) : S)));
        // flag 32: the caller sweeps slice ranges into gradient buffers of its own (bdof_adjoint_range, the tiled path), where
        // [Bmax][S] rows would not fit — 260 GB for 121 tiles of 512^2 x 1024 slices
        if ((with_grad & 32) == 0) HIPC(c, hipMalloc((void**)&c->grot, sizeof(float2) * fld * (size_t)S));
        HIPC(c, hipMalloc((void**)&c->gcar, sizeof(double2) * (size_t)Bmax));
        HIPC(c, hipMalloc((void**)&c->gt0, sizeof(double2) * (size_t)Bmax));
    }
    if (c->adj64) {
        HIPC(c, hipMalloc((void**)&c->g64, sizeof(double2) * fld));
        HIPC(c, hipMalloc((void**)&c->hs64, sizeof(double2) * NX * NY));
        HIPC(c, hipMalloc((void**)&c->hdet64, sizeof(double2) * NX * NY));
    }
    c->npartial = c->ncu * 16 + 64;
    // the resident kernel leaves one pair per workgroup AND wave (up to 16 waves)
    HIPC(c, hipMalloc((void**)&c->partial, sizeof(double) * 2 * c->npartial * (c->resident ? 16 : 1)));
    HIPC(c, hipMalloc((void**)&c->loss_dev, sizeof(double)));
    HIPC(c, hipMemsetAsync(c->loss_dev, 0, sizeof(double), c->stream));
    return 0;
}

int bdof_set_physics(bdof_ctx* c, double k, const float* hs, const float* hs_det, const double* h00, const double* hdet00,
                     int det_mode, int variant) {
    if (!c || !hs || !h00) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (det_mode < 0 || det_mode > 2 || variant < 0 || variant > 1) return fail(c, BDOF_ERR_ARG, "bad det_mode / variant");
    if (det_mode == BDOF_DET_NEAR && !hs_det) return fail(c, BDOF_ERR_ARG, "hs_det required for BDOF_DET_NEAR");
    const size_t bytes = sizeof(cf) * c->NX * c->NY;
    c->c64_tf = false; c->c64_ks = 0;     // a float64 twin handed over before (bdof_set_tf_f64 / bdof_set_conv_f64) held the previous model
    HIPC(c, hipMemcpyAsync(c->hs, hs, bytes, hipMemcpyHostToDevice, c->stream));
    if (hs_det) {
        HIPC(c, hipMemcpyAsync(c->hdet, hs_det, bytes, hipMemcpyHostToDevice, c->stream));
        // tf_all + near detector: the two consecutive steps F^-1 hdet F F^-1 hs F collapse into one
        std::vector<float> comb((size_t)2 * c->NX * c->NY);
        const double n = (double)c->NX * c->NY;
        for (size_t i = 0; i < (size_t)c->NX * c->NY; ++i) {
            const double ar = hs[2 * i], ai = hs[2 * i + 1], br = hs_det[2 * i], bi = hs_det[2 * i + 1];
            comb[2 * i] = (float)((ar * br - ai * bi) * n);
            comb[2 * i + 1] = (float)((ar * bi + ai * br) * n);
        }
        HIPC(c, hipMemcpyAsync(c->hcomb, comb.data(), bytes, hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
    }
    if (c->resident) {
        // the resident kernel multiplies the field image [kx][ky] element by element: tables in that order
        const int N = c->NX;
        std::vector<float> t((size_t)2 * N * N);
        for (int pass = 0; pass < (hs_det ? 2 : 1); ++pass) {
            const float* src = pass ? hs_det : hs;
            for (int ky = 0; ky < N; ++ky)
                for (int kx = 0; kx < N; ++kx) {
                    t[2 * ((size_t)kx * N + ky)] = src[2 * ((size_t)ky * N + kx)];
                    t[2 * ((size_t)kx * N + ky) + 1] = src[2 * ((size_t)ky * N + kx) + 1];
                }
            HIPC(c, hipMemcpy(pass ? c->hdetT : c->hsT, t.data(), bytes, hipMemcpyHostToDevice));
        }
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    c->res_dirty = true;
    c->k = c->k_fft = (float)k;
    c->h00 = std::complex<double>(h00[0], h00[1]);
    c->hdet00 = hdet00 ? std::complex<double>(hdet00[0], hdet00[1]) : std::complex<double>(1.0, 0.0);
    c->det_mode = det_mode;
    c->variant = variant;
    c->have_physics = true;
    c->have_h64 = false;             // float64 adjoint: the caller follows up with bdof_set_physics_f64
    c->hs_copies = 0;                // dithered copies of the old table: the caller follows up with bdof_set_transfer_f64
    c->mod_dirty = true;
    return 0;
}

// The slice step's transfer function in float64 ([ky][kx], 1 / (NX NY) folded in — what bdof_set_physics' hs was rounded from):
// the streaming engine's per-slice launches then multiply by DITHERED float32 copies of it, copy z mod D for slice z
// (bdof_field.h: k_dither_copies; D = BDOF_H_DITHER, default 64 capped at 256 MiB of tables, 0 = off), the adjoint step by the
// conjugate of the copy the forward step used.  A fixed float32 table is the same perturbation of every slice: its error grows
// linearly with the number of slices (np_funcs.py:42 runs float64).
int bdof_set_transfer_f64(bdof_ctx* c, const double* hs64) {
    if (!c || !hs64) return BDOF_ERR_ARG;
    if (!c->have_physics) return fail(c, BDOF_ERR_STATE, "bdof_set_physics has not been called");
    const char* env = std::getenv("BDOF_H_DITHER");
    const size_t n = (size_t)c->NX * c->NY;
    int D = env ? std::max(0, std::min(256, atoi(env))) : 64;
    while (D > 1 && (size_t)D * n * sizeof(cf) > ((size_t)256 << 20)) D /= 2;
    if (D < 2) { c->hs_copies = 0; return 0; }
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->hs_d) { (void)hipFree(c->hs_d); c->hs_d = nullptr; }
    if (c->hsT_d) { (void)hipFree(c->hsT_d); c->hsT_d = nullptr; }
    c->hs_copies = 0;
    double2* tmp = nullptr;
    HIPC(c, hipMalloc(&tmp, n * sizeof(double2)));
    hipError_t e = hipMalloc(&c->hs_d, (size_t)D * n * sizeof(cf));
    if (e != hipSuccess) { (void)hipFree(tmp); c->hs_d = nullptr; return fail(c, (int)e, "hipMalloc of the dithered transfer-function copies failed"); }
    e = hipMemcpy(tmp, hs64, n * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_dither_copies, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, tmp, c->hs_d, n, D);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess && c->resident) {
        // the LDS-resident kernel multiplies its field image [kx][ky] element by element: the same copies, transposed
        std::vector<double> t(2 * n);
        for (int ky = 0; ky < c->NY; ++ky)
            for (int kx = 0; kx < c->NX; ++kx) {
                t[2 * ((size_t)kx * c->NY + ky)] = hs64[2 * ((size_t)ky * c->NX + kx)];
                t[2 * ((size_t)kx * c->NY + ky) + 1] = hs64[2 * ((size_t)ky * c->NX + kx) + 1];
            }
        e = hipMalloc(&c->hsT_d, (size_t)D * n * sizeof(cf));
        if (e == hipSuccess) e = hipMemcpy(tmp, t.data(), n * sizeof(double2), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_dither_copies, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, tmp, c->hsT_d, n, D);
            e = hipStreamSynchronize(c->stream);
        }
        c->res_dirty = true;
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, (int)e, "building the dithered transfer-function copies failed");
    c->hs_copies = D;
    return 0;
}

int bdof_set_physics_f64(bdof_ctx* c, const double* hs, const double* hs_det) {
    if (!c || !hs) return BDOF_ERR_ARG;
    if (!c->adj64) return fail(c, BDOF_ERR_STATE, "bdof_set_physics_f64 needs bdof_configure with flag 64 (float64 adjoint sweep)");
    if (!c->have_physics) return fail(c, BDOF_ERR_STATE, "bdof_set_physics has not been called");
    if (c->det_mode == BDOF_DET_NEAR && !hs_det) return fail(c, BDOF_ERR_ARG, "hs_det required for BDOF_DET_NEAR");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    const size_t bytes = sizeof(double2) * (size_t)c->NX * c->NY;
    HIPC(c, hipMemcpy(c->hs64, hs, bytes, hipMemcpyHostToDevice));
    if (hs_det) HIPC(c, hipMemcpy(c->hdet64, hs_det, bytes, hipMemcpyHostToDevice));
    c->have_h64 = true;
    return 0;
}

int bdof_set_probe(bdof_ctx* c, const float* probe, double a0_re, double a0_im) {
    if (!c || !probe) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipMemcpyAsync(c->probe, probe, sizeof(cf) * c->NX * c->NY, hipMemcpyHostToDevice, c->stream));
    c->c64_tf = false; c->c64_ks = 0;     // a float64 twin handed over before (bdof_set_tf_f64 / bdof_set_conv_f64) held the previous model
    HIPC(c, hipStreamSynchronize(c->stream));
    c->a0 = std::complex<double>(a0_re, a0_im);
    c->res_dirty = true;
    c->mod_dirty = true;            // whether the mean modulation rides on the carrier depends on a0 (want_cbar)
    c->have_probe = true;
    return 0;
}

int bdof_set_meas_mode(bdof_ctx* c, int mode) {
    if (!c) return BDOF_ERR_ARG;
    if (mode != 0 && mode != 1) return fail(c, BDOF_ERR_ARG, "bdof_set_meas_mode: mode must be 0 or 1");
    c->meas_dev = mode;
    return 0;
}

int bdof_probe_stack_supported(bdof_ctx* c) {
    // every engine of the transfer-function path carries it (the real-space propagator of bdof_set_conv does not)
    return c && c->NY > 0 ? 1 : 0;
}

int bdof_set_probe_stack(bdof_ctx* c, const float* stack, const float* det) {
    if (!c) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->pstack) { (void)hipFree(c->pstack); c->pstack = nullptr; }
    if (c->pdet) { (void)hipFree(c->pdet); c->pdet = nullptr; }
    if (c->pdetT) { (void)hipFree(c->pdetT); c->pdetT = nullptr; }
    if (c->pdet64) { (void)hipFree(c->pdet64); c->pdet64 = nullptr; }          // a host-supplied stack comes in float32 only
    if (c->pdetT64) { (void)hipFree(c->pdetT64); c->pdetT64 = nullptr; }
    c->mod_dirty = true;
    if (!stack && !det) return 0;
    if (!stack || !det) return fail(c, BDOF_ERR_ARG, "bdof_set_probe_stack: both arrays or neither");
    const size_t fld = sizeof(cf) * (size_t)c->NX * c->NY;
    HIPC(c, hipMalloc((void**)&c->pstack, fld * (size_t)c->S));
    HIPC(c, hipMalloc((void**)&c->pdet, fld));
    HIPC(c, hipMemcpy(c->pstack, stack, fld * (size_t)c->S, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->pdet, det, fld, hipMemcpyHostToDevice));
    // the streaming far-field detector works on rows [ky][kx]: the same field transposed
    std::vector<float> t((size_t)2 * c->NX * c->NY);
    for (int i = 0; i < c->NX; ++i)
        for (int j = 0; j < c->NY; ++j) {
            t[2 * ((size_t)j * c->NX + i)] = det[2 * ((size_t)i * c->NY + j)];
            t[2 * ((size_t)j * c->NX + i) + 1] = det[2 * ((size_t)i * c->NY + j) + 1];
        }
    HIPC(c, hipMalloc((void**)&c->pdetT, fld));
    HIPC(c, hipMemcpy(c->pdetT, t.data(), fld, hipMemcpyHostToDevice));
    return 0;
}

// The carrier field of a localised probe (bdof_set_probe_stack) computed ON THE DEVICE in float64: S - 1 transfer-function steps
// of one field with rocFFT's double-precision plans (np_funcs.py:42 without an object), each plane rounded once into the
// float32 stack.  A probe that changes every Adam step (probe_type='optimizable') then costs ~4 small launches per slice
// instead of 2 S host FFTs and a 1-GiB upload.
int bdof_set_probe_field(bdof_ctx* c, const double* probe, const double* hT, const double* hdetT) {
    if (!c || !probe || !hT) return BDOF_ERR_ARG;
    if (!c->have_physics) return fail(c, BDOF_ERR_STATE, "bdof_set_physics has not been called");
    if (c->det_mode == BDOF_DET_NEAR && !hdetT) return fail(c, BDOF_ERR_ARG, "hdetT required for BDOF_DET_NEAR");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (!g_rocfft_ready) { RFC(c, rocfft_setup()); g_rocfft_ready = true; }
    const size_t n = (size_t)c->NX * c->NY, fld = sizeof(cf) * n, dbytes = sizeof(double2) * n;
    rocfft_plan pf = nullptr, pi = nullptr;
    rocfft_execution_info info = nullptr;
    double2 *dp = nullptr, *dh = nullptr, *dhd = nullptr;
    void* work = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        if (pf) (void)rocfft_plan_destroy(pf);
        if (pi) (void)rocfft_plan_destroy(pi);
        if (info) (void)rocfft_execution_info_destroy(info);
        for (void* q : {(void*)dp, (void*)dh, (void*)dhd, work}) if (q) (void)hipFree(q);
    };
#define PF_TRY(expr) do { rc = (expr); if (rc) { cleanup(); return rc; } } while (0)
#define PF_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(c, (int)e_, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
#define PF_RF(call) do { rocfft_status s_ = (call); if (s_ != rocfft_status_success) { cleanup(); return fail(c, BDOF_ERR_STATE, std::string(#call) + ": rocfft status " + std::to_string((int)s_)); } } while (0)
    const size_t lengths[2] = {(size_t)c->NY, (size_t)c->NX};      // fastest dimension first: fields are [x][y]
    PF_RF(rocfft_plan_create(&pf, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 2, lengths, 1, nullptr));
    PF_RF(rocfft_plan_create(&pi, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, 2, lengths, 1, nullptr));
    size_t w1 = 0, w2 = 0;
    PF_RF(rocfft_plan_get_work_buffer_size(pf, &w1));
    PF_RF(rocfft_plan_get_work_buffer_size(pi, &w2));
    PF_RF(rocfft_execution_info_create(&info));
    PF_RF(rocfft_execution_info_set_stream(info, (void*)c->stream));
    if (std::max(w1, w2)) {
        PF_HIP(hipMalloc(&work, std::max(w1, w2)));
        PF_RF(rocfft_execution_info_set_work_buffer(info, work, std::max(w1, w2)));
    }
    PF_HIP(hipMalloc((void**)&dp, dbytes));
    PF_HIP(hipMalloc((void**)&dh, dbytes));
    PF_HIP(hipMemcpy(dp, probe, dbytes, hipMemcpyHostToDevice));
    PF_HIP(hipMemcpy(dh, hT, dbytes, hipMemcpyHostToDevice));
    if (hdetT) {
        PF_HIP(hipMalloc((void**)&dhd, dbytes));
        PF_HIP(hipMemcpy(dhd, hdetT, dbytes, hipMemcpyHostToDevice));
    }
    if (c->pstack) { (void)hipFree(c->pstack); c->pstack = nullptr; }
    if (c->pdet) { (void)hipFree(c->pdet); c->pdet = nullptr; }
    if (c->pdetT) { (void)hipFree(c->pdetT); c->pdetT = nullptr; }
    if (c->pdet64) { (void)hipFree(c->pdet64); c->pdet64 = nullptr; }
    if (c->pdetT64) { (void)hipFree(c->pdetT64); c->pdetT64 = nullptr; }
    PF_HIP(hipMalloc((void**)&c->pstack, fld * (size_t)c->S));
    PF_HIP(hipMalloc((void**)&c->pdet, fld));
    PF_HIP(hipMalloc((void**)&c->pdetT, fld));
    static const bool no_f64_det = std::getenv("BDOF_NO_F64_DET") != nullptr;      // A/B switch: residual from float32 planes as before
    if (!no_f64_det) {
        PF_HIP(hipMalloc((void**)&c->pdet64, dbytes));
        PF_HIP(hipMalloc((void**)&c->pdetT64, dbytes));
    }
    const int grid = g_elem_grid(c, n);
    void* buf[1] = {dp};
    auto step = [&](const double2* h) -> int {
        if (rocfft_execute(pf, buf, nullptr, info) != rocfft_status_success) return fail(c, BDOF_ERR_STATE, "rocfft_execute (double) failed");
        hipLaunchKernelGGL(k_d_mul, dim3(grid), dim3(256), 0, c->stream, dp, h, n, 1.0 / (double)n);
        if (rocfft_execute(pi, buf, nullptr, info) != rocfft_status_success) return fail(c, BDOF_ERR_STATE, "rocfft_execute (double) failed");
        return 0;
    };
    for (int z = 0; z < c->S; ++z) {
        hipLaunchKernelGGL(k_d_to_f, dim3(grid), dim3(256), 0, c->stream, dp, c->pstack + (size_t)z * n, c->NX, c->NY, 0);
        if (z < c->S - 1) PF_TRY(step(dh));
    }
    if (c->det_mode == BDOF_DET_FAR) {
        if (rocfft_execute(pf, buf, nullptr, info) != rocfft_status_success) { cleanup(); return fail(c, BDOF_ERR_STATE, "rocfft_execute (double) failed"); }
    } else {
        if (c->variant == BDOF_VARIANT_TF_ALL) PF_TRY(step(dh));
        if (c->det_mode == BDOF_DET_NEAR) PF_TRY(step(dhd));
    }
    hipLaunchKernelGGL(k_d_to_f, dim3(grid), dim3(256), 0, c->stream, dp, c->pdet, c->NX, c->NY, 0);
    hipLaunchKernelGGL(k_d_to_f, dim3(grid), dim3(256), 0, c->stream, dp, c->pdetT, c->NX, c->NY, 1);
    // ... and unrounded: the detector kernels add the scattered wave to these and take |d| - m in float64 (loss_seed_f64)
    if (c->pdet64) {
        hipLaunchKernelGGL(k_d_copy, dim3(grid), dim3(256), 0, c->stream, dp, c->pdet64, c->NX, c->NY, 0);
        hipLaunchKernelGGL(k_d_copy, dim3(grid), dim3(256), 0, c->stream, dp, c->pdetT64, c->NX, c->NY, 1);
    }
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(c->stream));
#undef PF_TRY
#undef PF_HIP
#undef PF_RF
    cleanup();
    // the wave is carried as p_z + eps_z with eps_0 = 0: no probe of its own, no scalar carrier
    HIPC(c, hipMemsetAsync(c->probe, 0, fld, c->stream));
    c->a0 = 0.0;
    c->res_dirty = true;
    c->mod_dirty = true;
    c->have_probe = true;
    return 0;
}

int bdof_set_object(bdof_ctx* c, const void* vol, long long n_rows, int volNY, const int* tab, int volNX, int n_angles) {
    if (!c || !vol || n_rows < 1) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (volNY < 1) return fail(c, BDOF_ERR_ARG, "volNY must be >= 1");
    if (tab && (volNX < 1 || n_angles < 1)) return fail(c, BDOF_ERR_ARG, "volNX and n_angles must be >= 1 with a table");
    if (!tab && volNY != c->NY) return fail(c, BDOF_ERR_ARG, "without a rotation table volNY must equal NY");
    c->obj_src = (const float2*)vol;
    c->obj_bound_mod = false;
    c->obj_rows = (size_t)n_rows;
    c->mod_dirty = true;
    c->obj.vol = nullptr;
    c->obj.volNY = volNY;
    c->obj.tab = tab;
    c->obj.volNX = tab ? volNX : c->NX;
    c->obj.S = c->S;
    c->n_angles = tab ? n_angles : 0;
    return 0;
}

int bdof_set_rotation_adjoint(bdof_ctx* c, const int* off, const int* order, int n_dest) {
    if (!c || !off || !order || n_dest < 1) return BDOF_ERR_ARG;
    c->adj_off = off; c->adj_order = order; c->adj_ndest = n_dest;
    HIPC(c, hipSetDevice(c->device));
    if (c->heavy) (void)hipFree(c->heavy);
    c->heavy = nullptr;
    HIPC(c, hipMalloc((void**)&c->heavy, sizeof(int) * ((size_t)n_dest + 1)));
    return 0;
}

static void set_batch_views(bdof_ctx* c, const int* angle_of_b, const int* xoff, const int* yoff) {
    c->obj.angle_of_b = angle_of_b;
    c->obj.xoff = xoff;
    c->obj.yoff = yoff;
}

int bdof_forward(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, void* out_wave, int keep_tape) {
    int r = check_ready(c, B);
    if (r) return r;
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    if (keep_tape && !c->with_grad) return fail(c, BDOF_ERR_STATE, "keep_tape needs bdof_configure(with_grad=1)");
    if (keep_tape && c->recompute) return fail(c, BDOF_ERR_STATE, "the per-slice history is not kept in tape-free (recompute) mode");
    HIPC(c, hipSetDevice(c->device));
    set_batch_views(c, angle_of_b, xoff, yoff);
    if ((r = ensure_modulation_k(c, c->k_fft))) return r;
    if (use_resident(c, B) && !keep_tape) {
        if ((r = resident_run(c, B, nullptr, out_wave, false))) return r;
        c->tape_valid = c->last_valid = false;
        return 0;
    }
    if (c->generic) {
        if (keep_tape) return fail(c, BDOF_ERR_STATE, "the per-slice history is not kept by the generic-size engine");
        if ((r = generic_forward(c, B, out_wave, false))) return r;
        c->tape_valid = c->last_valid = false;
        HIPC(c, hipGetLastError());
        return 0;
    }
    const bool tf_all = c->variant == BDOF_VARIANT_TF_ALL;
    Group groups[BDOF_MAX_GROUPS];
    const int ng = batch_groups(c, B, c->NX, 16, groups);
    if ((r = fork_streams(c, ng))) return r;
    forward_sweep(c, groups, ng, keep_tape ? TAPE_HISTORY : TAPE_NONE);
    if ((r = join_streams(c, ng))) return r;
    c->tape_valid = keep_tape != 0;
    c->last_valid = false;
    if (out_wave) {
        if (c->det_mode == BDOF_DET_FAR)
            launch_loss_far(c, B, c->bufA, nullptr, (cf*)out_wave, nullptr, 1.f, 1.f, 0.f, c->pdetT);
        else if (c->det_mode == BDOF_DET_NONE && !tf_all)
            launch_loss_real(c, B, c->bufA, nullptr, false, (cf*)out_wave, nullptr, 1.f / c->NY, 1.f, 0.f, carrier_det(c), c->pdet);
        else
            launch_loss_real(c, B, c->bufB, nullptr, false, (cf*)out_wave, nullptr, 1.f, 1.f, 0.f, carrier_det(c), c->pdet);
    }
    if (keep_tape && !tf_all) {
        // probe_array[S-1] = phi_{S-1} (np_funcs.py:41-43): keep R phi_{S-1} in L1 order in bufA
        if (!(c->det_mode == BDOF_DET_NONE)) {
            const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
            const cf* in = c->S > 1 ? c->tape + (size_t)(c->S - 2) * fld : nullptr;   // psi_hat_{S-1}
            launch_row_fwd(c, B, c->S - 1, in, c->bufA, false);
        }
        c->last_valid = true;
    }
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_forward_range(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                       const void* in_real, void* out_real, int prop_last) {
    int r = check_ready(c, B);
    if (r) return r;
    if (!in_real || !out_real) return BDOF_ERR_ARG;
    if (z0 < 0 || nz < 1 || z0 + nz > c->S) return fail(c, BDOF_ERR_ARG, "slice range outside [0, S)");
    if (c->generic) return fail(c, BDOF_ERR_SIZE, "bdof_forward_range runs on the fused streaming kernels (NY, NX powers of two in 64..1024)");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    if (c->range_car && (B != c->range_car_B || z0 != c->range_car_z0))
        return fail(c, BDOF_ERR_ARG, "the range carriers (bdof_set_range_carrier) were handed over for another batch / first slice");
    HIPC(c, hipSetDevice(c->device));
    set_batch_views(c, angle_of_b, xoff, yoff);
    if ((r = ensure_modulation_k(c, c->k_fft))) return r;
    Group groups[BDOF_MAX_GROUPS];
    const int ng = batch_groups(c, B, c->NX, 16, groups);
    if ((r = fork_streams(c, ng))) return r;
    for (int z = z0; z < z0 + nz; ++z) {
        const bool last = z == z0 + nz - 1;
        for (int gi = 0; gi < ng; ++gi) {
            const int Bg = groups[gi].B;
            use_group(c, groups[gi]);
            const cf* start = z == z0 ? (const cf*)in_real : nullptr;
            if (!last || prop_last) {
                launch_row_fwd(c, Bg, z, c->bufB, c->bufA, true, nullptr, start);
                launch_row_prop(c, Bg, c->bufA, c->bufB, c->hs, 1.f, 0);
            } else {
                launch_row_fwd(c, Bg, z, c->bufB, c->bufA, false, nullptr, start);
            }
        }
    }
    // real-space wave after the range: psi_{z0+nz} (prop_last) or phi_{z0+nz-1}
    const int zl = z0 + nz - 1;
    for (int gi = 0; gi < ng; ++gi) {
        use_group(c, groups[gi]);
        if (prop_last)
            launch_loss_real(c, groups[gi].B, c->bufB, nullptr, false, (cf*)out_real, nullptr, 1.f, 1.f, 0.f, carrier_at(c, zl + 1),
                             zl + 1 < c->S ? slice_carrier_field(c, zl + 1) : c->pdet);
        else
            launch_loss_real(c, groups[gi].B, c->bufA, nullptr, false, (cf*)out_real, nullptr, 1.f / c->NY, 1.f, 0.f, carrier_phi_at(c, zl),
                             slice_carrier_field(c, zl));
    }
    if ((r = join_streams(c, ng))) return r;
    c->tape_valid = c->last_valid = false;
    HIPC(c, hipGetLastError());
    return 0;
}

// Carrier fields for the NEXT bdof_forward_range calls over slices z0 .. z0 + nz - 1 of B wavefields: stack [nz][B][NX][NY] complex64
// (device), wavefield b's free-space propagation to the entrance of each of those slices.  The range is then swept on
// psi_z = p_z + eps_z with only eps in the float32 transforms, `in_real` is the scattered part entering the range (zeros when the
// carrier IS the incoming wave) and `out_real` receives the scattered part leaving it — for the tiles of a corrected stitch range
// exactly T psi - T_free psi, without the round-off of two full-amplitude sweeps.  NULL removes the stack.
int bdof_set_range_carrier(bdof_ctx* c, const void* stack, int B, int z0, int nz) {
    if (!c) return BDOF_ERR_ARG;
    if (!stack) { c->range_car = nullptr; c->range_car_B = 0; return 0; }
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (c->pstack || c->a0 != std::complex<double>(0.0, 0.0))
        return fail(c, BDOF_ERR_STATE, "range carriers replace the ctx's own probe carrier: bind a zero probe (bdof_set_probe, a0 = 0) and no probe stack");
    if (B < 1 || B > c->Bmax || z0 < 0 || nz < 1 || z0 + nz > c->S) return fail(c, BDOF_ERR_ARG, "batch / slice range outside the configuration");
    c->range_car = (const cf*)stack;
    c->range_car_B = B;
    c->range_car_z0 = z0;
    return 0;
}

int bdof_forward_range_h(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                         const void* in_real, void* out_real, int prop_last, const void* h) {
    if (!c || !h) return BDOF_ERR_ARG;
    c->hs_override = (const cf*)h;
    const int r = bdof_forward_range(c, B, angle_of_b, xoff, yoff, z0, nz, in_real, out_real, prop_last);
    c->hs_override = nullptr;
    return r;
}

// Adjoint of bdof_forward_range(prop_last = 1) in the tape-free form: the forward wave is marched back from the range's end
// state beside the adjoint field.  end_real: psi_{z0+nz} (what bdof_forward_range returned), g_end_real: G(psi_{z0+nz});
// g_start_real receives G(psi_{z0}); the gradient rows of the range go to grot_range [B][nz][NX][NY] (pairs).
int bdof_adjoint_range(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                       const void* end_real, const void* g_end_real, void* g_start_real, void* grot_range) {
    int r = check_ready(c, B);
    if (r) return r;
    if (!end_real || !g_end_real || !g_start_real || !grot_range) return BDOF_ERR_ARG;
    if (z0 < 0 || nz < 1 || z0 + nz > c->S) return fail(c, BDOF_ERR_ARG, "slice range outside [0, S)");
    if (c->generic) return fail(c, BDOF_ERR_SIZE, "bdof_adjoint_range runs on the fused streaming kernels");
    if (!c->with_grad || !c->tape) return fail(c, BDOF_ERR_STATE, "bdof_adjoint_range needs bdof_configure(with_grad=1)");
    if (c->S < 2) return fail(c, BDOF_ERR_SIZE, "bdof_adjoint_range needs at least two tape fields (S >= 2)");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    HIPC(c, hipSetDevice(c->device));
    set_batch_views(c, angle_of_b, xoff, yoff);
    if ((r = ensure_modulation_k(c, c->k_fft))) return r;
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    cf* const rc1 = c->tape;                       // marched-back phi_hat_z (L1)
    cf* const rc2 = c->tape + fld;                 // R eps(psi_z) (L2)
    Group groups[BDOF_MAX_GROUPS];
    const int ng = batch_groups(c, B, c->NX, 16, groups);
    if ((r = fork_streams(c, ng))) return r;
    const int zt = z0 + nz - 1;
    for (int gi = 0; gi < ng; ++gi) {
        const int Bg = groups[gi].B;
        use_group(c, groups[gi]);
        // phi_{zt} = P^H psi_{zt+1} and G(phi_{zt}) = P^H G(psi_{zt+1}): real space -> R (transposed) -> adjoint step
        RealToHybArgs ra{sub_field(c, (const cf*)end_real), sub_field(c, c->bufA), Bg, c->NX, c->twY};
        DISPATCH_N(c->NY, { hipLaunchKernelGGL((k_row_real_to_hyb<N_>), dim3(rows_grid<N_>(c, Bg, c->NX)), dim3(BDOF_THREADS), 0, c->sub_stream, ra); });
        launch_row_prop(c, Bg, c->bufA, rc1, c->hs, 1.f, 1, zt);
        RealToHybArgs rg{sub_field(c, (const cf*)g_end_real), sub_field(c, c->bufA), Bg, c->NX, c->twY};
        DISPATCH_N(c->NY, { hipLaunchKernelGGL((k_row_real_to_hyb<N_>), dim3(rows_grid<N_>(c, Bg, c->NX)), dim3(BDOF_THREADS), 0, c->sub_stream, rg); });
        launch_row_prop(c, Bg, c->bufA, c->bufB, c->hs, 1.f, 1, zt);
    }
    for (int z = zt; z >= z0; --z) {
        for (int gi = 0; gi < ng; ++gi) {
            const int Bg = groups[gi].B;
            use_group(c, groups[gi]);
            GradTarget gt;
            gt.grot = (float2*)grot_range; gt.S_ = nz; gt.z_ = z - z0;
            gt.gpsi = z == z0 ? (cf*)g_start_real : nullptr;
            launch_row_bwd(c, Bg, z, c->bufB, rc1, z > z0 ? c->bufA : nullptr, 3, 1.f, &gt);
            if (z > z0) {
                launch_row_prop(c, Bg, c->bufA, c->bufB, c->hs, 1.f, 1, z - 1);
                launch_row_unmod(c, Bg, z, rc1, rc2, false, 1.f);
                launch_row_prop(c, Bg, rc2, rc1, c->hs, 1.f, 1, z - 1);
            }
        }
    }
    if ((r = join_streams(c, ng))) return r;
    c->tape_valid = c->last_valid = false;
    c->gpsi_src = nullptr;
    HIPC(c, hipGetLastError());
    return 0;
}

// loss = mean((|field| - meas)^2) over n = FX * FY pixels (left on the device, bdof_get_loss) and, in place of the field,
// the adjoint seed 2 (|d| - m) d / |d| / n — the detector-less (free_prop_cm None) loss of a whole field
int bdof_field_loss_seed(bdof_ctx* c, void* field, const float* meas, int FX, int FY) {
    if (!c || !field || !meas || FX < 1 || FY < 1) return BDOF_ERR_ARG;
    if (!c->partial) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipSetDevice(c->device));
    const size_t n = (size_t)FX * FY;
    const int egrid = g_elem_grid(c, n);
    GLossArgs la{(cf*)field, nullptr, meas, c->partial, 1, FX, FY, 0, make_float2(0.f, 0.f), (float)(2.0 / (double)n), nullptr, 0, 0.f,
                 nullptr, nullptr, make_double2(0.0, 0.0), make_double2(0.0, 0.0), nullptr, nullptr, 0.0, make_double2(0.0, 0.0)};
    hipLaunchKernelGGL(k_g_loss, dim3(egrid), dim3(256), 0, c->stream, la);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, egrid, 1.0 / (double)n, c->loss_dev);
    HIPC(c, hipGetLastError());
    return 0;
}

static int tiles_check(bdof_ctx* c, const void* field, const void* tiles, int B, int FX, int FY, int TX, int TY, const int* x0, const int* y0) {
    if (!c || !field || !tiles || !x0 || !y0) return BDOF_ERR_ARG;
    if (B < 1 || FX < 1 || FY < 1 || TX < 1 || TY < 1) return fail(c, BDOF_ERR_ARG, "bad tile / field shape");
    return 0;
}

int bdof_tiles_gather(bdof_ctx* c, const void* field, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0, const int* y0,
                      int taper) {
    int r = tiles_check(c, field, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    HIPC(c, hipSetDevice(c->device));
    if (taper < 0 || 2 * taper > TX || 2 * taper > TY) return fail(c, BDOF_ERR_ARG, "taper must fit the tile");
    TileArgs a{(cf*)field, (cf*)tiles, x0, y0, B, FX, FY, TX, TY, 0, 0, taper};
    hipLaunchKernelGGL(k_tiles_gather, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a, 0);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_tiles_scatter(bdof_ctx* c, const void* tiles, void* field, int FX, int FY, int B, int TX, int TY, const int* x0, const int* y0,
                       int halo_x, int halo_y) {
    int r = tiles_check(c, field, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (halo_x < 0 || halo_y < 0 || 2 * halo_x >= TX || 2 * halo_y >= TY) return fail(c, BDOF_ERR_ARG, "halo must leave a core");
    HIPC(c, hipSetDevice(c->device));
    TileArgs a{(cf*)field, (cf*)tiles, x0, y0, B, FX, FY, TX, TY, halo_x, halo_y, 0};
    hipLaunchKernelGGL(k_tiles_scatter, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

// adjoint of bdof_tiles_scatter: tiles = the field on every tile's core, zero elsewhere
int bdof_tiles_scatter_adjoint(bdof_ctx* c, const void* field, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0, const int* y0,
                               int halo_x, int halo_y) {
    int r = tiles_check(c, field, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (halo_x < 0 || halo_y < 0 || 2 * halo_x >= TX || 2 * halo_y >= TY) return fail(c, BDOF_ERR_ARG, "halo must leave a core");
    HIPC(c, hipSetDevice(c->device));
    TileArgs a{(cf*)field, (cf*)tiles, x0, y0, B, FX, FY, TX, TY, halo_x, halo_y, 0};
    hipLaunchKernelGGL(k_tiles_gather, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a, 1);
    HIPC(c, hipGetLastError());
    return 0;
}

// adjoint of bdof_tiles_gather: field = sum of the tiles' pixels, weighted with the taper, at the positions they were cut from
int bdof_tiles_gather_adjoint(bdof_ctx* c, const void* tiles, void* field, int FX, int FY, int B, int TX, int TY, const int* x0, const int* y0,
                              int taper) {
    int r = tiles_check(c, field, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (taper < 0 || 2 * taper > TX || 2 * taper > TY) return fail(c, BDOF_ERR_ARG, "taper must fit the tile");
    HIPC(c, hipSetDevice(c->device));
    TileArgs a{(cf*)field, (cf*)tiles, x0, y0, B, FX, FY, TX, TY, 0, 0, taper};
    hipLaunchKernelGGL(k_tiles_gather_adjoint, dim3(std::min(FX, c->ncu * 8)), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

// object gradient of a slice range of tiles, added into the volume gradient rows (see k_tiles_grad_add)
int bdof_tiles_grad_add(bdof_ctx* c, const void* grot_range, void* gvol, int B, int TX, int TY, const int* x0, const int* y0, int z0, int nz) {
    if (!c || !grot_range || !gvol || !x0 || !y0 || B < 1) return BDOF_ERR_ARG;
    if (!c->obj.tab) return fail(c, BDOF_ERR_STATE, "bdof_set_object with a table has not been called");
    if (z0 < 0 || nz < 1 || z0 + nz > c->S) return fail(c, BDOF_ERR_ARG, "slice range outside [0, S)");
    HIPC(c, hipSetDevice(c->device));
    TileGradArgs a{(const float2*)grot_range, (float2*)gvol, c->obj.tab, x0, y0, B, TX, TY, c->obj.volNX, c->obj.volNY, z0, nz, 1};
    hipLaunchKernelGGL(k_tiles_grad_add, dim3(std::min(c->obj.volNX, c->ncu * 8)), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

// ---- whole-field / tile-batch operations in either precision (bdof_field.h) ------------------------------------------------
static int field_plans(bdof_ctx* c, int NX, int NY, int B, bool dbl, rocfft_plan* fwd, rocfft_plan* inv) {
    if (!g_rocfft_ready) { RFC(c, rocfft_setup()); g_rocfft_ready = true; }
    const std::array<int, 4> key{NX, NY, B, dbl ? 1 : 0};
    auto it = c->fplans.find(key);
    if (it == c->fplans.end()) {
        const size_t lengths[2] = {(size_t)NY, (size_t)NX};             // fastest dimension first
        const rocfft_precision prec = dbl ? rocfft_precision_double : rocfft_precision_single;
        rocfft_plan pf = nullptr, pi = nullptr;
        RFC(c, rocfft_plan_create(&pf, rocfft_placement_inplace, rocfft_transform_type_complex_forward, prec, 2, lengths, (size_t)B, nullptr));
        RFC(c, rocfft_plan_create(&pi, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, prec, 2, lengths, (size_t)B, nullptr));
        size_t w1 = 0, w2 = 0;
        RFC(c, rocfft_plan_get_work_buffer_size(pf, &w1));
        RFC(c, rocfft_plan_get_work_buffer_size(pi, &w2));
        const size_t need = std::max(w1, w2);
        if (need > c->gwork_sz) {
            HIPC(c, hipStreamSynchronize(c->stream));
            if (c->gwork) (void)hipFree(c->gwork);
            c->gwork = nullptr;
            c->gwork_sz = 0;
            HIPC(c, hipMalloc(&c->gwork, need));
            c->gwork_sz = need;
        }
        it = c->fplans.emplace(key, std::make_pair(pf, pi)).first;
    }
    if (!c->ginfo) {
        RFC(c, rocfft_execution_info_create(&c->ginfo));
        RFC(c, rocfft_execution_info_set_stream(c->ginfo, (void*)c->stream));
    }
    if (c->gwork_sz) RFC(c, rocfft_execution_info_set_work_buffer(c->ginfo, c->gwork, c->gwork_sz));
    *fwd = it->second.first;
    *inv = it->second.second;
    return 0;
}

// fields[b] <- F^-1 ( h * F fields[b] ) for B fields [NX][NY] in place; h[kx][ky] carries 1 / (NX NY) (and any power of the
// transfer function: np_funcs.py:42 composes in free space); is_double: complex128 fields and table, else complex64
int bdof_fields_free_step(bdof_ctx* c, void* fields, int B, int NX, int NY, const void* h, int conj_h, int is_double) {
    if (!c || !fields || !h || B < 1 || NX < 1 || NY < 1) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    rocfft_plan pf, pi;
    int r = field_plans(c, NX, NY, B, is_double != 0, &pf, &pi);
    if (r) return r;
    void* buf[1] = {fields};
    const size_t per = (size_t)NX * NY, n = per * B;
    RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo));
    if (is_double) hipLaunchKernelGGL(k_f_hmul<double2>, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (double2*)fields, (const double2*)h, per, n, conj_h);
    else hipLaunchKernelGGL(k_f_hmul<float2>, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (float2*)fields, (const float2*)h, per, n, conj_h);
    RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo));
    HIPC(c, hipGetLastError());
    return 0;
}

// The same step on an AUXILIARY stream: it starts after everything queued on the ctx's stream so far (first copying `src` into
// `fields` there, if given) and runs beside what the ctx's stream is handed next — the whole-field step of a stitch range beside the
// tiles' carrier steps and sweeps (tiling.TiledPropagator).  bdof_aux_join makes the ctx's stream wait for it.  One step in
// flight at a time; own rocFFT execution info and work buffer.
int bdof_fields_free_step_aux(bdof_ctx* c, void* fields, const void* src, int B, int NX, int NY, const void* h, int conj_h, int is_double) {
    if (!c || !fields || !h || B < 1 || NX < 1 || NY < 1) return BDOF_ERR_ARG;
    if (c->aux_pending) return fail(c, BDOF_ERR_STATE, "an auxiliary step is in flight: bdof_aux_join first");
    HIPC(c, hipSetDevice(c->device));
    rocfft_plan pf, pi;
    int r = field_plans(c, NX, NY, B, is_double != 0, &pf, &pi);
    if (r) return r;
    if (!c->aux) {
        HIPC(c, hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
        HIPC(c, hipEventCreateWithFlags(&c->ev_aux_fork, hipEventDisableTiming));
        HIPC(c, hipEventCreateWithFlags(&c->ev_aux_join, hipEventDisableTiming));
        RFC(c, rocfft_execution_info_create(&c->ginfo_aux));
        RFC(c, rocfft_execution_info_set_stream(c->ginfo_aux, (void*)c->aux));
    }
    size_t w1 = 0, w2 = 0;
    RFC(c, rocfft_plan_get_work_buffer_size(pf, &w1));
    RFC(c, rocfft_plan_get_work_buffer_size(pi, &w2));
    const size_t need = std::max(w1, w2);
    if (need > c->gwork_aux_sz) {
        HIPC(c, hipStreamSynchronize(c->aux));
        if (c->gwork_aux) (void)hipFree(c->gwork_aux);
        c->gwork_aux = nullptr; c->gwork_aux_sz = 0;
        HIPC(c, hipMalloc(&c->gwork_aux, need));
        c->gwork_aux_sz = need;
    }
    if (c->gwork_aux_sz) RFC(c, rocfft_execution_info_set_work_buffer(c->ginfo_aux, c->gwork_aux, c->gwork_aux_sz));
    const size_t per = (size_t)NX * NY, n = per * B, esz = is_double ? sizeof(double2) : sizeof(float2);
    HIPC(c, hipEventRecord(c->ev_aux_fork, c->stream));
    HIPC(c, hipStreamWaitEvent(c->aux, c->ev_aux_fork, 0));
    c->aux_pending = true;
    if (src) HIPC(c, hipMemcpyAsync(fields, src, n * esz, hipMemcpyDeviceToDevice, c->aux));
    void* buf[1] = {fields};
    RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo_aux));
    if (is_double) hipLaunchKernelGGL(k_f_hmul<double2>, dim3(g_elem_grid(c, n)), dim3(256), 0, c->aux, (double2*)fields, (const double2*)h, per, n, conj_h);
    else hipLaunchKernelGGL(k_f_hmul<float2>, dim3(g_elem_grid(c, n)), dim3(256), 0, c->aux, (float2*)fields, (const float2*)h, per, n, conj_h);
    RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo_aux));
    HIPC(c, hipGetLastError());
    return 0;
}

// The carrier stack of a stitch range (bdof_set_range_carrier) in one call: p0 [B][NX][NY] complex128 (device; overwritten) are
// the wavefields entering the range; stack [nz][B][NX][NY] complex64 receives p_z = F^-1(H^z F p0), z = 0 .. nz - 1 — the
// free-space propagation of each to the entrance of every slice of the range, formed in double: one forward transform, the nz - 1
// spectra s_hat H^z by a running product, ONE batched inverse transform of all of them, one conversion.  h: complex128 [kx][ky],
// ifftshift(H) / (NX NY).
int bdof_range_carrier_build(bdof_ctx* c, void* p0, void* stack, int B, int NX, int NY, const void* h, int nz) {
    if (!c || !p0 || !stack || !h || B < 1 || NX < 1 || NY < 1 || nz < 1) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    const size_t per = (size_t)NX * NY, n = per * B;
    hipLaunchKernelGGL(k_f_to_float, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (const double2*)p0, (cf*)stack, n);
    if (nz > 1) {
        const size_t need = n * (size_t)(nz - 1);
        if (need > c->car_scratch_n) {
            HIPC(c, hipStreamSynchronize(c->stream));
            if (c->car_scratch) (void)hipFree(c->car_scratch);
            c->car_scratch = nullptr; c->car_scratch_n = 0;
            HIPC(c, hipMalloc(&c->car_scratch, need * sizeof(double2)));
            c->car_scratch_n = need;
        }
        rocfft_plan pf, pi, qf, qi;
        int r = field_plans(c, NX, NY, B, true, &pf, &pi);
        if (r) return r;
        if ((r = field_plans(c, NX, NY, B * (nz - 1), true, &qf, &qi))) return r;       // (sizes the shared work buffer for both)
        if ((r = field_plans(c, NX, NY, B, true, &pf, &pi))) return r;
        void* b0[1] = {p0};
        RFC(c, rocfft_execute(pf, b0, nullptr, c->ginfo));
        hipLaunchKernelGGL(k_carrier_spectra, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (const double2*)p0, (const double2*)h, c->car_scratch,
                           per, B, nz, (double)NX * (double)NY);
        void* b1[1] = {c->car_scratch};
        RFC(c, rocfft_execute(qi, b1, nullptr, c->ginfo));
        hipLaunchKernelGGL(k_f_to_float, dim3(g_elem_grid(c, need)), dim3(256), 0, c->stream, (const double2*)c->car_scratch, (cf*)stack + n, need);
    }
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_aux_join(bdof_ctx* c) {
    if (!c) return BDOF_ERR_ARG;
    if (!c->aux_pending) return 0;
    HIPC(c, hipSetDevice(c->device));
    c->aux_pending = false;
    HIPC(c, hipEventRecord(c->ev_aux_join, c->aux));
    HIPC(c, hipStreamWaitEvent(c->stream, c->ev_aux_join, 0));
    return 0;
}

// y += alpha x on n complex numbers (either precision)
int bdof_caxpy(bdof_ctx* c, void* y, const void* x, double alpha, size_t n, int is_double) {
    if (!c || !y || !x) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    if (is_double) hipLaunchKernelGGL(k_f_axpy<double>, dim3(g_elem_grid(c, 2 * n)), dim3(256), 0, c->stream, (double*)y, (const double*)x, alpha, 2 * n);
    else hipLaunchKernelGGL(k_f_axpy<float>, dim3(g_elem_grid(c, 2 * n)), dim3(256), 0, c->stream, (float*)y, (const float*)x, (float)alpha, 2 * n);
    HIPC(c, hipGetLastError());
    return 0;
}

// complex64 <-> complex128 copies of n numbers
int bdof_c_convert(bdof_ctx* c, void* dst, const void* src, size_t n, int to_double) {
    if (!c || !dst || !src) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    if (to_double) hipLaunchKernelGGL(k_f_to_double, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (const cf*)src, (double2*)dst, n);
    else hipLaunchKernelGGL(k_f_to_float, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (const double2*)src, (cf*)dst, n);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_tiles_gather_f64(bdof_ctx* c, const void* field, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0, const int* y0,
                          int taper) {
    int r = tiles_check(c, field, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (taper < 0 || 2 * taper > TX || 2 * taper > TY) return fail(c, BDOF_ERR_ARG, "taper must fit the tile");
    HIPC(c, hipSetDevice(c->device));
    Tile64Args a{(double2*)field, (double2*)tiles, x0, y0, B, FX, FY, TX, TY, 0, 0, taper};
    hipLaunchKernelGGL(k_tiles_gather64, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_tiles_scatter_f64(bdof_ctx* c, const void* tiles, void* field, int FX, int FY, int B, int TX, int TY, const int* x0, const int* y0,
                           int halo_x, int halo_y) {
    int r = tiles_check(c, field, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (halo_x < 0 || halo_y < 0 || 2 * halo_x >= TX || 2 * halo_y >= TY) return fail(c, BDOF_ERR_ARG, "halo must leave a core");
    HIPC(c, hipSetDevice(c->device));
    Tile64Args a{(double2*)field, (double2*)tiles, x0, y0, B, FX, FY, TX, TY, halo_x, halo_y, 0};
    hipLaunchKernelGGL(k_tiles_scatter64, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

// complex64 tiles cut out of a complex128 field (tapered, periodic) / written back into it:
// field[core] = (accumulate ? field[core] : 0) + tiles_a - tiles_b (tiles_b nullable), the sum formed in float64
int bdof_tiles_gather_mixed(bdof_ctx* c, const void* field64, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0, const int* y0,
                            int taper) {
    int r = tiles_check(c, field64, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (taper < 0 || 2 * taper > TX || 2 * taper > TY) return fail(c, BDOF_ERR_ARG, "taper must fit the tile");
    HIPC(c, hipSetDevice(c->device));
    TileMixArgs a{(double2*)field64, nullptr, nullptr, (cf*)tiles, x0, y0, B, FX, FY, TX, TY, 0, 0, taper, 0};
    hipLaunchKernelGGL(k_tiles_gather_mixed, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a, 0);
    HIPC(c, hipGetLastError());
    return 0;
}

// adjoint of bdof_tiles_scatter_diff64 w.r.t. tiles_a: complex64 tiles = the complex128 field on every tile's core, zero elsewhere
int bdof_tiles_scatter_adjoint_mixed(bdof_ctx* c, const void* field64, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0,
                                     const int* y0, int halo_x, int halo_y) {
    int r = tiles_check(c, field64, tiles, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (halo_x < 0 || halo_y < 0 || 2 * halo_x >= TX || 2 * halo_y >= TY) return fail(c, BDOF_ERR_ARG, "halo must leave a core");
    HIPC(c, hipSetDevice(c->device));
    TileMixArgs a{(double2*)field64, nullptr, nullptr, (cf*)tiles, x0, y0, B, FX, FY, TX, TY, halo_x, halo_y, 0, 0};
    hipLaunchKernelGGL(k_tiles_gather_mixed, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a, 1);
    HIPC(c, hipGetLastError());
    return 0;
}

// adjoint of bdof_tiles_gather_mixed, on a difference of tiles: field64 (+)= sum of the tapered (tiles_a - tiles_b) pixels at the
// positions they were cut from (periodically); tiles_b nullable
int bdof_tiles_gather_adjoint_diff64(bdof_ctx* c, const void* tiles_a, const void* tiles_b, void* field64, int FX, int FY, int B, int TX, int TY,
                                     const int* x0, const int* y0, int taper, int accumulate) {
    int r = tiles_check(c, field64, tiles_a, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (taper < 0 || 2 * taper > TX || 2 * taper > TY) return fail(c, BDOF_ERR_ARG, "taper must fit the tile");
    HIPC(c, hipSetDevice(c->device));
    TileMixArgs a{(double2*)field64, (const cf*)tiles_a, (const cf*)tiles_b, nullptr, x0, y0, B, FX, FY, TX, TY, 0, 0, taper, accumulate};
    hipLaunchKernelGGL(k_tiles_gather_adjoint_diff64, dim3(std::min(FX, c->ncu * 8)), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_tiles_scatter_diff64(bdof_ctx* c, const void* tiles_a, const void* tiles_b, void* field64, int FX, int FY, int B, int TX, int TY,
                              const int* x0, const int* y0, int halo_x, int halo_y, int accumulate) {
    int r = tiles_check(c, field64, tiles_a, B, FX, FY, TX, TY, x0, y0);
    if (r) return r;
    if (halo_x < 0 || halo_y < 0 || 2 * halo_x >= TX || 2 * halo_y >= TY) return fail(c, BDOF_ERR_ARG, "halo must leave a core");
    HIPC(c, hipSetDevice(c->device));
    TileMixArgs a{(double2*)field64, (const cf*)tiles_a, (const cf*)tiles_b, nullptr, x0, y0, B, FX, FY, TX, TY, halo_x, halo_y, 0, accumulate};
    hipLaunchKernelGGL(k_tiles_scatter_diff64, dim3((TY + 255) / 256, std::min(TX, 64), B), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

// bdof_forward_range in float64 on caller-owned complex128 fields [B][NX][NY], in place: slices z0 .. z0+nz-1 of the bound object
// seen through the windows (xoff, yoff); modulation from the caller's (delta, beta) rows with k in float64, transforms by
// rocFFT in double precision, h[kx][ky] = the transfer function / (NX NY) in float64.  cnn_propagator/np_funcs.py:36-43.
int bdof_forward_range_f64(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz, void* fields,
                           const void* h, double k, int prop_last) {
    if (!c || !fields || !h) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (!c->obj_src) return fail(c, BDOF_ERR_STATE, "bdof_set_object (with (delta, beta) rows) has not been called");
    if (B < 1) return fail(c, BDOF_ERR_ARG, "batch size must be positive");
    if (z0 < 0 || nz < 1 || z0 + nz > c->S) return fail(c, BDOF_ERR_ARG, "slice range outside [0, S)");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    HIPC(c, hipSetDevice(c->device));
    rocfft_plan pf, pi;
    int r = field_plans(c, c->NX, c->NY, B, true, &pf, &pi);
    if (r) return r;
    ObjView o = c->obj;
    o.vol = c->obj_src;
    o.angle_of_b = angle_of_b;
    o.xoff = xoff;
    o.yoff = yoff;
    const size_t per = (size_t)c->NX * c->NY, n = per * B;
    void* buf[1] = {fields};
    for (int z = z0; z < z0 + nz; ++z) {
        Mod64Args m{(double2*)fields, o, B, c->NX, c->NY, z, k, nullptr};
        hipLaunchKernelGGL(k_f64_modulate, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, m);
        if (z < z0 + nz - 1 || prop_last) {
            RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo));
            hipLaunchKernelGGL(k_f_hmul<double2>, dim3(g_elem_grid(c, n)), dim3(256), 0, c->stream, (double2*)fields, (const double2*)h, per, n, 0);
            RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo));
        }
    }
    HIPC(c, hipGetLastError());
    return 0;
}

// ---- the real-space propagator in float64 (bdof_conv64.h) -------------------------------------------------------------------
// Buffers of the float64 paths for B wavefields: the wave, the tape of S slices and, for the real-space model, the renormalised
// exit wave and the padded grid.  Allocated when the model is handed over (for Bmax), so that a device without room for them says
// so at set-up and not in the middle of a run.
static int c64_room(bdof_ctx* c, int B, bool tf, size_t M) {
    const size_t n = (size_t)c->NX * c->NY * B;
    const bool have = c->c64_B >= B && c->c64_psi && c->c64_tape && (tf || (c->c64_q && c->c64_big));
    if (have) return 0;
    HIPC(c, hipStreamSynchronize(c->stream));
    for (double2** q : {&c->c64_psi, &c->c64_q, &c->c64_big, &c->c64_tape}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    c->c64_B = 0;
    HIPC(c, hipMalloc(&c->c64_psi, n * sizeof(double2)));
    HIPC(c, hipMalloc(&c->c64_tape, (size_t)c->S * n * sizeof(double2)));
    if (!tf) {
        HIPC(c, hipMalloc(&c->c64_q, n * sizeof(double2)));
        HIPC(c, hipMalloc(&c->c64_big, M * M * (size_t)B * sizeof(double2)));
    }
    c->c64_B = B;
    return 0;
}

// probe: host complex128 [NX][NY]; khat: host complex128 [M][M], M = NX + ks - 1 = NY + ks - 1, the 2-D transform of the ks x ks
// kernel zero-padded to M x M, transposed to [kx][ky] and divided by M^2; ksum = sum of the kernel's taps (the padding constant's
// recursion, propagation.py:91,99); k = 2 pi dz / lambda with numpy's pi (propagation.py:25)
int bdof_set_conv_f64(bdof_ctx* c, const double* probe, const double* khat, int ks, double ksum_re, double ksum_im, double k) {
    if (!c || !probe || !khat) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (c->NX != c->NY) return fail(c, BDOF_ERR_SIZE, "the float64 real-space path takes square wavefields");
    if (ks < 1 || ks % 2 == 0 || ks >= c->NX) return fail(c, BDOF_ERR_ARG, "kernel_size must be odd and smaller than the field");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->NX * c->NY, M = (size_t)c->NX + ks - 1;
    if (c->c64_ks != ks) {
        for (double2** q : {&c->c64_khat, &c->c64_big}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        c->c64_B = 0;
    }
    if (!c->c64_probe) HIPC(c, hipMalloc(&c->c64_probe, n * sizeof(double2)));
    if (!c->c64_khat) HIPC(c, hipMalloc(&c->c64_khat, M * M * sizeof(double2)));
    if (!c->c64_scal) HIPC(c, hipMalloc(&c->c64_scal, 2 * sizeof(double2)));
    if (!c->c64_part) HIPC(c, hipMalloc(&c->c64_part, (size_t)c->ncu * 16 * sizeof(double2)));
    HIPC(c, hipMemcpy(c->c64_probe, probe, n * sizeof(double2), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->c64_khat, khat, M * M * sizeof(double2), hipMemcpyHostToDevice));
    c->c64_ks = ks;
    c->c64_tf = false;
    c->c64_ksum = std::complex<double>(ksum_re, ksum_im);
    c->c64_k = k;
    return c64_room(c, c->Bmax, false, M);
}

// the step to a detector at a finite distance for the float64 real-space path (propagation.py:122-127: one transfer-function
// step of the renormalised exit wave): hdetT host complex128 [kx][ky], ifftshift(H_det) / (NX NY); NULL removes it
int bdof_set_conv_f64_detector(bdof_ctx* c, const double* hdetT) {
    if (!c) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->NX * c->NY;
    if (!hdetT) { if (c->c64_hdet) (void)hipFree(c->c64_hdet); c->c64_hdet = nullptr; return 0; }
    if (!c->c64_hdet) HIPC(c, hipMalloc(&c->c64_hdet, n * sizeof(double2)));
    HIPC(c, hipMemcpy(c->c64_hdet, hdetT, n * sizeof(double2), hipMemcpyHostToDevice));
    return 0;
}

// bdof_loss_grad_conv with every quantity in float64: loss left for bdof_get_loss, gradient rows in the ctx's rotated-frame
// buffer (bdof_grot) like every other engine, so bdof_rotation_adjoint / bdof_window_rotation_adjoint follow unchanged.
// meas: device float, laid out as for bdof_loss_grad_conv (far field: [b][ky][kx] un-shifted, else [b][x][y]); meas_ref: what
// the host subtracted from the amplitudes (bdof_set_meas_mode 1), 0 otherwise.  Detector: none, far field, or near field after
// bdof_set_conv_f64_detector.
int bdof_loss_grad_conv_f64(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, double meas_ref) {
    if (!c || !meas) return BDOF_ERR_ARG;
    if (!c->c64_ks || c->c64_tf) return fail(c, BDOF_ERR_STATE, "bdof_set_conv_f64 has not been called");
    if (!c->obj_src) return fail(c, BDOF_ERR_STATE, "bdof_set_object (with (delta, beta) rows) has not been called");
    if (!c->grot || !c->partial) return fail(c, BDOF_ERR_STATE, "bdof_configure(with_grad = 1) needed");
    if (B < 1 || B > c->Bmax) return fail(c, BDOF_ERR_ARG, "batch size outside [1, Bmax]");
    if (c->det_mode == BDOF_DET_NEAR && !c->c64_hdet)
        return fail(c, BDOF_ERR_STATE, "near-field detector: bdof_set_conv_f64_detector has not been called");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    HIPC(c, hipSetDevice(c->device));
    const int N = c->NX, ks = c->c64_ks, p = (ks - 1) / 2, M = N + ks - 1, S = c->S;
    const size_t per = (size_t)N * N, n = per * B, nbig = (size_t)M * M * B;
    int r;
    if ((r = c64_room(c, B, false, (size_t)M))) return r;
    rocfft_plan pf, pi;
    if ((r = field_plans(c, N, N, B, true, &pf, &pi))) return r;           // detector transforms
    ObjView o = c->obj;
    o.vol = c->obj_src;
    o.angle_of_b = angle_of_b;
    o.xoff = xoff;
    o.yoff = yoff;
    const int eg = g_elem_grid(c, n), egb = g_elem_grid(c, nbig);
    double2 *psi = c->c64_psi, *big = c->c64_big;
    hipLaunchKernelGGL(k_c64_bcast, dim3(eg), dim3(256), 0, c->stream, c->c64_probe, psi, B, per);
    std::complex<double> edge(1.0, 0.0);
    for (int z = 0; z < S; ++z) {
        Mod64Args m{psi, o, B, N, N, z, c->c64_k, c->c64_tape + (size_t)z * n};
        hipLaunchKernelGGL(k_f64_modulate, dim3(eg), dim3(256), 0, c->stream, m);
        hipLaunchKernelGGL(k_c64_pad, dim3(egb), dim3(256), 0, c->stream, psi, big, B, N, M, p, make_double2(edge.real(), edge.imag()));
        if ((r = bdof_fields_free_step(c, big, B, M, M, c->c64_khat, 0, 1))) return r;
        hipLaunchKernelGGL(k_c64_crop, dim3(eg), dim3(256), 0, c->stream, big, psi, B, N, M, ks - 1);
        edge *= c->c64_ksum;
    }
    // q = s P, s = psi_0[0,0,0] / P[0,0,0] (one scalar for the batch)
    double2 init;
    HIPC(c, hipMemcpyAsync(&init, c->c64_probe, sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(k_c64_corner, dim3(1), dim3(1), 0, c->stream, psi, init, c->c64_scal);
    hipLaunchKernelGGL(k_c64_scale, dim3(eg), dim3(256), 0, c->stream, psi, n, c->c64_scal, 0);
    HIPC(c, hipMemcpyAsync(c->c64_q, psi, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    const bool far = c->det_mode == BDOF_DET_FAR;
    void* buf[1] = {psi};
    const bool near = c->det_mode == BDOF_DET_NEAR;
    if (far) RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo));           // un-shifted, un-normalised fft2 (propagation.py:114-115)
    else if (near) { if ((r = bdof_fields_free_step(c, psi, B, N, N, c->c64_hdet, 0, 1))) return r; }      // propagation.py:122-124
    const int lgrid = std::min(eg, c->npartial);
    hipLaunchKernelGGL(k_c64_loss, dim3(lgrid), dim3(256), 0, c->stream, psi, meas, c->partial, B, N, N, far ? 1 : 0, meas_ref, 2.0 / (double)n);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, lgrid, 1.0 / (double)n, c->loss_dev);
    if (far) RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo));           // G(q) = N^2 ifft2(G(d)): the un-normalised inverse
    else if (near) { if ((r = bdof_fields_free_step(c, psi, B, N, N, c->c64_hdet, 1, 1))) return r; }
    // adjoint of the renormalisation through the corner pixel
    const int dgrid = std::min(eg, c->ncu * 16);
    hipLaunchKernelGGL(k_c64_dot, dim3(dgrid), dim3(256), 0, c->stream, psi, c->c64_q, n, c->c64_part);
    hipLaunchKernelGGL(k_c64_scale, dim3(eg), dim3(256), 0, c->stream, psi, n, c->c64_scal, 1);
    hipLaunchKernelGGL(k_c64_corner_adj, dim3(1), dim3(1), 0, c->stream, psi, c->c64_part, dgrid, c->c64_scal);
    for (int z = S - 1; z >= 0; --z) {
        // adjoint of (constant pad, valid convolution): embed, circular correlation, crop to the un-padded pixels
        hipLaunchKernelGGL(k_c64_pad, dim3(egb), dim3(256), 0, c->stream, psi, big, B, N, M, ks - 1, make_double2(0.0, 0.0));
        if ((r = bdof_fields_free_step(c, big, B, M, M, c->c64_khat, 1, 1))) return r;
        hipLaunchKernelGGL(k_c64_crop, dim3(eg), dim3(256), 0, c->stream, big, psi, B, N, M, p);
        C64BwdArgs ba{psi, c->c64_tape + (size_t)z * n, c->grot, o, B, N, N, S, z, c->c64_k};
        hipLaunchKernelGGL(k_c64_bwd, dim3(eg), dim3(256), 0, c->stream, ba);
    }
    c->tape_valid = c->last_valid = false;
    c->gpsi_src = nullptr;
    HIPC(c, hipGetLastError());
    return 0;
}

// ---- the transfer-function model in float64 on the same context -------------------------------------------------------------
// probe: host complex128 [NX][NY]; hT / hdetT (nullable: no near-field detector step): host complex128 [kx][ky], the
// ifftshift-ed transfer functions / (NX NY); k = 2 pi dz / lambda as bdof_set_physics has it
int bdof_set_tf_f64(bdof_ctx* c, const double* probe, const double* hT, const double* hdetT, double k) {
    if (!c || !probe || !hT) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (c->det_mode == BDOF_DET_NEAR && !hdetT) return fail(c, BDOF_ERR_ARG, "a near-field detector needs its transfer function in float64 too");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->NX * c->NY;
    if (!c->c64_probe) HIPC(c, hipMalloc(&c->c64_probe, n * sizeof(double2)));
    if (!c->c64_h) HIPC(c, hipMalloc(&c->c64_h, n * sizeof(double2)));
    if (hdetT && !c->c64_hdet) HIPC(c, hipMalloc(&c->c64_hdet, n * sizeof(double2)));
    HIPC(c, hipMemcpy(c->c64_probe, probe, n * sizeof(double2), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->c64_h, hT, n * sizeof(double2), hipMemcpyHostToDevice));
    if (hdetT) HIPC(c, hipMemcpy(c->c64_hdet, hdetT, n * sizeof(double2), hipMemcpyHostToDevice));
    c->c64_tf = true;
    c->c64_ks = 1;
    c->c64_k = k;
    return c64_room(c, c->Bmax, true, 0);
}

// bdof_loss_grad with every quantity in float64 (np_funcs.py:15-65 as autograd differentiates it in the reference: modulation
// from the (delta, beta) rows, F^-1 H F per slice by rocFFT in double, magnitude loss, adjoint sweep): loss for bdof_get_loss,
// gradient rows in bdof_grot.  Unfused — the accuracy path of the first minibatch of an epoch, and a float64 twin of the fused
// kernels on the device.  meas: device float, laid out as for bdof_loss_grad (+ meas_ref under bdof_set_meas_mode(1)).
int bdof_loss_grad_tf_f64(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, double meas_ref) {
    if (!c || !meas) return BDOF_ERR_ARG;
    if (!c->c64_tf) return fail(c, BDOF_ERR_STATE, "bdof_set_tf_f64 has not been called");
    if (!c->obj_src) return fail(c, BDOF_ERR_STATE, "bdof_set_object (with (delta, beta) rows) has not been called");
    if (!c->grot || !c->partial) return fail(c, BDOF_ERR_STATE, "bdof_configure(with_grad = 1) needed");
    if (B < 1 || B > c->Bmax) return fail(c, BDOF_ERR_ARG, "batch size outside [1, Bmax]");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    HIPC(c, hipSetDevice(c->device));
    const int NX = c->NX, NY = c->NY, S = c->S;
    const size_t per = (size_t)NX * NY, n = per * B;
    int r;
    if ((r = c64_room(c, B, true, 0))) return r;
    rocfft_plan pf, pi;
    if ((r = field_plans(c, NX, NY, B, true, &pf, &pi))) return r;
    ObjView o = c->obj;
    o.vol = c->obj_src;
    o.angle_of_b = angle_of_b;
    o.xoff = xoff;
    o.yoff = yoff;
    const int eg = g_elem_grid(c, n);
    const bool far = c->det_mode == BDOF_DET_FAR;
    // a step after the last slice: variant tf_all, except in front of a far-field detector (|F P phi| = |H F phi| = |F phi|)
    const bool prop_last = c->variant == BDOF_VARIANT_TF_ALL && !far;
    double2* psi = c->c64_psi;
    hipLaunchKernelGGL(k_c64_bcast, dim3(eg), dim3(256), 0, c->stream, c->c64_probe, psi, B, per);
    for (int z = 0; z < S; ++z) {
        Mod64Args m{psi, o, B, NX, NY, z, c->c64_k, c->c64_tape + (size_t)z * n};
        hipLaunchKernelGGL(k_f64_modulate, dim3(eg), dim3(256), 0, c->stream, m);
        if (z < S - 1 || prop_last)
            if ((r = bdof_fields_free_step(c, psi, B, NX, NY, c->c64_h, 0, 1))) return r;
    }
    void* buf[1] = {psi};
    if (c->det_mode == BDOF_DET_NEAR) { if ((r = bdof_fields_free_step(c, psi, B, NX, NY, c->c64_hdet, 0, 1))) return r; }
    else if (far) RFC(c, rocfft_execute(pf, buf, nullptr, c->ginfo));       // un-shifted, un-normalised fft2
    const int lgrid = std::min(eg, c->npartial);
    hipLaunchKernelGGL(k_c64_loss, dim3(lgrid), dim3(256), 0, c->stream, psi, meas, c->partial, B, NX, NY, far ? 1 : 0, meas_ref, 2.0 / (double)n);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, lgrid, 1.0 / (double)n, c->loss_dev);
    if (c->det_mode == BDOF_DET_NEAR) { if ((r = bdof_fields_free_step(c, psi, B, NX, NY, c->c64_hdet, 1, 1))) return r; }
    else if (far) RFC(c, rocfft_execute(pi, buf, nullptr, c->ginfo));       // F^H: the un-normalised inverse
    for (int z = S - 1; z >= 0; --z) {
        if (z < S - 1 || prop_last)
            if ((r = bdof_fields_free_step(c, psi, B, NX, NY, c->c64_h, 1, 1))) return r;
        C64BwdArgs ba{psi, c->c64_tape + (size_t)z * n, c->grot, o, B, NX, NY, S, z, c->c64_k};
        hipLaunchKernelGGL(k_c64_bwd, dim3(eg), dim3(256), 0, c->stream, ba);
    }
    c->tape_valid = c->last_valid = false;
    c->gpsi_src = nullptr;
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_tape_to_real(bdof_ctx* c, int i, int B, void* out) {
    int r = check_ready(c, B);
    if (r) return r;
    if (!out) return BDOF_ERR_ARG;
    if (!c->tape_valid) return fail(c, BDOF_ERR_STATE, "no history: run bdof_forward(keep_tape=1) first");
    if (i < 0 || i >= c->S) return fail(c, BDOF_ERR_ARG, "slice index outside [0, S)");
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    if (i < c->S - 1) {
        launch_loss_real(c, B, c->tape + (size_t)i * fld, nullptr, false, (cf*)out, nullptr, 1.f, 1.f, 0.f, carrier_at(c, i + 1),
                         slice_carrier_field(c, i + 1));
    } else {
        if (!c->last_valid)
            return fail(c, BDOF_ERR_STATE, "the last slice's wave is only kept after bdof_forward(keep_tape=1) with the numpy_skip_last variant");
        launch_loss_real(c, B, c->bufA, nullptr, false, (cf*)out, nullptr, 1.f / c->NY, 1.f, 0.f, carrier_phi_at(c, c->S - 1),
                         slice_carrier_field(c, c->S - 1));
    }
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_loss_grad(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, void* out_wave) {
    int r = check_ready(c, B);
    if (r) return r;
    if (!meas) return BDOF_ERR_ARG;
    if (c->meas_dev && (c->det_mode == BDOF_DET_FAR || c->pstack))
        return fail(c, BDOF_ERR_STATE, "bdof_set_meas_mode(1) needs a real-space detector and a scalar carrier");
    if (!c->with_grad || !c->grot) return fail(c, BDOF_ERR_STATE, "bdof_loss_grad needs bdof_configure(with_grad=1) with the gradient workspace (not flag 32)");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    HIPC(c, hipSetDevice(c->device));
    set_batch_views(c, angle_of_b, xoff, yoff);
    if ((r = ensure_modulation_k(c, c->k_fft))) return r;
    c->gpsi_src = c->gpsi0;
    c->gpsi_B = B;
    if (use_resident(c, B)) {
        if ((r = resident_run(c, B, meas, out_wave, true))) return r;
        c->tape_valid = c->last_valid = false;
        return 0;
    }
    if (c->generic) {
        if ((r = generic_loss_grad(c, B, meas, out_wave))) return r;
        c->tape_valid = c->last_valid = false;
        c->gpsi_src = c->bufA;                  // the adjoint sweep ends with G(psi_0) in the field buffer, [b][x][y]
        HIPC(c, hipGetLastError());
        return 0;
    }
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    const float NYf = (float)c->NY;
    const bool tf_all = c->variant == BDOF_VARIANT_TF_ALL;
    Group groups[BDOF_MAX_GROUPS];
    const int ng = batch_groups(c, B, c->NX, 16, groups);
    if ((r = fork_streams(c, ng))) return r;
    // The tape is the per-slice history psi_hat_z that the transfer-function kernel writes anyway; A'_z recomputes phi_z from
    // it (one more transform per launch, no tape write in A_z: 104 instead of 112 B per pixel per slice-step, 67.6 -> 65.3 ms
    // per step at 512^3 x 25).  BDOF_TAPE_PHI=1 selects the older form (A_z stores phi_z) for comparison; same bits.
    static const bool hist_env = std::getenv("BDOF_TAPE_PHI") == nullptr;
    const bool recompute = c->recompute;
    const bool hist_tape = hist_env && !recompute;
    forward_sweep(c, groups, ng, recompute ? TAPE_LAST : (hist_tape ? TAPE_HISTORY : TAPE_PHI));
    c->tape_valid = false;      // the tape holds phi_z (or an incomplete history), not what bdof_tape_to_real expects
    c->last_valid = false;
    const float seed_scale = 2.f / ((float)B * (float)c->NX * (float)c->NY);
    int npart = 0;
    // Detector + seed.  Afterwards bufB holds g_hat(phi_{S-1}) (L1 order, normalised hybrid).
    for (int gi = 0; gi < ng; ++gi) {
        const int Bg = groups[gi].B;
        use_group(c, groups[gi], npart);
        if (c->det_mode == BDOF_DET_FAR) {
            npart += launch_loss_far(c, Bg, c->bufA, c->bufB, (cf*)out_wave, meas, 1.f, 1.f, seed_scale, c->pdetT);
        } else if (c->det_mode == BDOF_DET_NONE && !tf_all) {
            npart += launch_loss_real(c, Bg, c->bufA, c->bufB, false, (cf*)out_wave, meas, 1.f / NYf, 1.f / NYf, seed_scale, carrier_det(c), c->pdet);
        } else {
            // the detector wave came out of a transfer-function step: seed -> R (transposed) -> adjoint step
            const cf* h = c->det_mode == BDOF_DET_NONE ? c->hs : (tf_all ? c->hcomb : c->hdet);
            npart += launch_loss_real(c, Bg, c->bufB, c->bufA, true, (cf*)out_wave, meas, 1.f, 1.f, seed_scale, carrier_det(c), c->pdet);
            launch_row_prop(c, Bg, c->bufA, c->bufB, h, 1.f, 1, c->S - 1);
        }
    }
    // backward sweep: A'_z (L1 -> L2), then the adjoint transfer-function step (L2 -> L1)
    // Tape-free form (bdof_configure flag 16): P is unitary and c_z is invertible, so the forward wave is marched
    // BACK beside the adjoint field instead of being stored per slice: phi_{z-1} = P^H (phi_z / c_z) — one A_z^-1 launch and
    // one more adjoint transfer-function launch per slice (+40 B per pixel per slice-step), S - 3 fields per wavefield less
    // memory.  phi_{S-1} is kept by the forward sweep (real space, tape slot 0); slots 1 / 2 hold the marched-back wave in L1 /
    // L2 order; slice 0 takes phi_0 from the probe as the history form does.
    cf* const rc_last = c->tape;
    cf* const rc1 = c->tape + fld;
    cf* const rc2 = c->tape + 2 * fld;
    for (int z = c->S - 1; z >= 0; --z) {
        for (int gi = 0; gi < ng; ++gi) {
            use_group(c, groups[gi]);
            if (recompute) {
                const bool top = z == c->S - 1;
                if (z == 0) launch_row_bwd(c, groups[gi].B, z, c->bufB, nullptr, nullptr, 2);
                else if (top) launch_row_bwd(c, groups[gi].B, z, c->bufB, rc_last, c->bufA, 0);
                else launch_row_bwd(c, groups[gi].B, z, c->bufB, rc1, c->bufA, 3, 1.f);
                if (z > 0) launch_row_prop(c, groups[gi].B, c->bufA, c->bufB, c->hs, 1.f, 1, z - 1);
                if (z > 1) {
                    launch_row_unmod(c, groups[gi].B, z, top ? rc_last : rc1, rc2, top, 1.f);
                    launch_row_prop(c, groups[gi].B, rc2, rc1, c->hs, 1.f, 1, z - 1);
                }
                continue;
            }
            if (hist_tape)
                launch_row_bwd(c, groups[gi].B, z, c->bufB, z > 0 ? c->tape + (size_t)(z - 1) * fld : nullptr, z > 0 ? c->bufA : nullptr,
                               z > 0 ? 1 : 2);
            else
                launch_row_bwd(c, groups[gi].B, z, c->bufB, c->tape + (size_t)z * fld, z > 0 ? c->bufA : nullptr);
            if (z > 0) launch_row_prop(c, groups[gi].B, c->bufA, c->bufB, c->hs, 1.f, 1, z - 1);
        }
    }
    if ((r = join_streams(c, ng))) return r;
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, npart,
                       1.0 / ((double)B * c->NX * c->NY), c->loss_dev);
    HIPC(c, hipGetLastError());
    return 0;
}


// =================================================================================================
// Real-space truncated-kernel propagator (cnn_propagator/propagation.py:18-133)
// =================================================================================================
int bdof_set_conv(bdof_ctx* c, const float* ky, const float* kx, int ks, double e_re, double e_im, double ksum_re,
                  double ksum_im, double k) {
    if (!c || !ky || !kx) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (ks < 1 || ks > BDOF_CONV_MAXK || ks % 2 == 0) return fail(c, BDOF_ERR_ARG, "kernel_size must be odd and <= 33");
    if (c->generic || c->NX % BDOF_CONV_TX || c->NY % BDOF_CONV_TY)
        return fail(c, BDOF_ERR_SIZE, "the conv propagator needs power-of-two wavefields (tiles are 32 x 64)");
    HIPC(c, hipSetDevice(c->device));
    c->c64_tf = false; c->c64_ks = 0;     // a float64 twin handed over before (bdof_set_tf_f64 / bdof_set_conv_f64) held the previous model
    for (int i = 0; i < ks; ++i) {
        c->taps.ky[i] = make_float2(ky[2 * i], ky[2 * i + 1]);
        c->taps.kx[i] = make_float2(kx[2 * i], kx[2 * i + 1]);
    }
    c->taps.e = make_float2((float)e_re, (float)e_im);
    c->taps.ks = ks;
    c->ksum = std::complex<double>(ksum_re, ksum_im);
    c->k_conv = (float)k;
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    if (!c->bufC) HIPC(c, hipMalloc((void**)&c->bufC, sizeof(cf) * fld));
    if (!c->conv_scal) HIPC(c, hipMalloc((void**)&c->conv_scal, sizeof(cf) * 4));
    HIPC(c, hipStreamSynchronize(c->stream));                  // a sweep still in flight reads the previous taps
    if (c->taps_dev && c->taps_copies != 1) { (void)hipFree(c->taps_dev); c->taps_dev = nullptr; }
    if (!c->taps_dev) HIPC(c, hipMalloc((void**)&c->taps_dev, sizeof(ConvTaps)));
    c->taps_copies = 1;
    HIPC(c, hipMemcpy(c->taps_dev, &c->taps, sizeof(ConvTaps), hipMemcpyHostToDevice));
    c->have_conv = true;
    c->mod_dirty = true;
    return 0;
}

int bdof_set_conv_taps_f64(bdof_ctx* c, const double* ky, const double* kx, double e_re, double e_im) {
    if (!c || !ky || !kx) return BDOF_ERR_ARG;
    if (!c->have_conv) return fail(c, BDOF_ERR_STATE, "bdof_set_conv has not been called");
    const int D = c->tw_dither;
    if (D < 2) return 0;                                       // BDOF_TW_DITHER=0: the nearest-rounded taps of bdof_set_conv stay
    HIPC(c, hipSetDevice(c->device));
    std::vector<ConvTaps> t((size_t)D, c->taps);
    const int ks = c->taps.ks;
    for (int d = 0; d < D; ++d) {
        for (int i = 0; i < ks; ++i) {
            // a phase of its own for each of the 4 ks + 2 numbers (golden-ratio sequence), as for the twiddle tables
            const double ph = 0.6180339887498949 * (4 * i + 1);
            t[d].ky[i] = make_float2(dither_pick(ky[2 * i], d, std::fmod(ph, 1.0)), dither_pick(ky[2 * i + 1], d, std::fmod(ph + 0.6180339887498949, 1.0)));
            t[d].kx[i] = make_float2(dither_pick(kx[2 * i], d, std::fmod(ph + 2 * 0.6180339887498949, 1.0)),
                                     dither_pick(kx[2 * i + 1], d, std::fmod(ph + 3 * 0.6180339887498949, 1.0)));
        }
        t[d].e = make_float2(dither_pick(e_re, d, 0.25), dither_pick(e_im, d, 0.75));
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->taps_dev) (void)hipFree(c->taps_dev);
    c->taps_dev = nullptr;
    HIPC(c, hipMalloc((void**)&c->taps_dev, sizeof(ConvTaps) * D));
    HIPC(c, hipMemcpy(c->taps_dev, t.data(), sizeof(ConvTaps) * D, hipMemcpyHostToDevice));
    c->taps_copies = D;
    return 0;
}

}  // extern "C"  (helpers below have C++ linkage)

static cf conv_carrier(const bdof_ctx* c, int z) {        // a_z = a_0 sum(K)^z
    const std::complex<double> a = c->a0 * std::pow(c->ksum, z);
    return make_float2((float)a.real(), (float)a.imag());
}
static cf conv_pad(const bdof_ctx* c, int z) {            // padding constant of eps: edge_val_z - a_z = (1 - a_0) sum(K)^z
    const std::complex<double> a = (1.0 - c->a0) * std::pow(c->ksum, z);
    return make_float2((float)a.real(), (float)a.imag());
}

static int conv_lds_bytes(const bdof_ctx* c) {
    const int h = (c->taps.ks - 1) / 2;
    const int TXH = BDOF_CONV_TX + 2 * h, TYH = BDOF_CONV_TY + 2 * h;
    return (TXH * (TYH | 1) + TXH * (BDOF_CONV_TY + 1)) * (int)sizeof(cf);
}

// BDOF_CONV_TILING=1 keeps the first tiling (k_conv, 32 x 64 tiles through staging registers) for every size; read at every
// launch so that one process can run both (tests/test_gpu_conv.py compares them)
static int conv_tiling() {
    const char* e = getenv("BDOF_CONV_TILING");
    return e && e[0] == '1' ? 1 : 2;
}

template <bool BWD, int H, bool PF> static int launch_conv_hp(bdof_ctx* c, ConvArgs& a, ProfScope& ps) {
    if constexpr (H == 2 || H == 4 || H == 8) {
        // second tiling (bdof_conv2.h): 64 x 32 tiles by LDS-DMA, static LDS, two workgroups per CU by registers
        if (conv_tiling() != 1 && a.NX % Conv2Cfg<H>::TX == 0 && a.NY % Conv2Cfg<H>::TY == 0) {
            // 8 runs of strips (bdof_conv2.h, XCD-aware order), nwg workgroups each: as few rounds as the workgroups per CU
            // allow, and the workgroup count that fills the last round best
            auto plan = [&](int tx, int per_cu) {
                const int nstrips = a.B * (a.NX / tx);
                const int run_tiles = ((nstrips + 7) / 8) * (a.NY / Conv2Cfg<H>::TY);
                const int slots = std::max(1, c->ncu * per_cu / 8);
                const int rounds = (run_tiles + slots - 1) / slots;
                return 8 * ((run_tiles + rounds - 1) / rounds);
            };
            BDOF_LAUNCH(ps, (k_conv2<BWD, H, PF>), dim3(plan(Conv2Cfg<H>::TX, 2)), dim3(Conv2Cfg<H>::THREADS), 0, c->sub_stream, a);
            return 0;
        }
    }
    const int lds = conv_lds_bytes(c);
    static bool attr_set[BDOF_MAX_DEVICES] = {};
    if (!attr_set[c->device % BDOF_MAX_DEVICES]) {
        HIPC(c, hipFuncSetAttribute((const void*)k_conv<BWD, H, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_set[c->device % BDOF_MAX_DEVICES] = true;
    }
    const int tiles = a.B * (a.NX / BDOF_CONV_TX) * (a.NY / BDOF_CONV_TY);
    const int grid = balanced_grid(c, tiles, 2);
    BDOF_LAUNCH(ps, (k_conv<BWD, H, PF>), dim3(grid), dim3(BDOF_CONV_THREADS), lds, c->sub_stream, a);
    return 0;
}
template <bool BWD, int H> static int launch_conv_h(bdof_ctx* c, ConvArgs& a, ProfScope& ps) {
    return a.pfield ? launch_conv_hp<BWD, H, true>(c, a, ps) : launch_conv_hp<BWD, H, false>(c, a, ps);
}

template <bool BWD> static int launch_conv(bdof_ctx* c, ConvArgs& a) {
    ProfScope ps(c, BWD ? BDOF_K_ROW_BWD : BDOF_K_ROW_FWD, true);
    switch ((c->taps.ks - 1) / 2) {           // register-window fast paths for the common kernel sizes 5, 9, 17, 33
        case 2: return launch_conv_h<BWD, 2>(c, a, ps);
        case 4: return launch_conv_h<BWD, 4>(c, a, ps);
        case 8: return launch_conv_h<BWD, 8>(c, a, ps);
        case 16: return launch_conv_h<BWD, 16>(c, a, ps);
        default: return launch_conv_h<BWD, 0>(c, a, ps);
    }
}

// forward sweep of the conv propagator; leaves psi_S (eps part) in bufB and the scalars in conv_scal
static int conv_forward_sweep(bdof_ctx* c, int B, bool tape) {
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    const size_t n = (size_t)B * c->NX * c->NY, plane = (size_t)c->NX * c->NY;
    const int egrid = (int)std::min<size_t>((n + 255) / 256, (size_t)c->ncu * 16);
    const bool cs = c->cstack != nullptr;           // carrier field: eps is the scattered wave alone, padded with zeros
    const cf zero = make_float2(0.f, 0.f);
    ObjView obj = c->obj;
    cf* cur = tape ? c->tape : c->bufA;
    ConvInitArgs ia{c->probe, cur, obj, B, c->NX, c->NY, conv_carrier(c, 0), cs ? c->cstack : nullptr};
    hipLaunchKernelGGL(k_conv_init, dim3(egrid), dim3(256), 0, c->stream, ia);
    int r;
    // sub-batches on streams of their own, as on the transfer-function path (batch_groups): the ramp and tail of one group's
    // launch are filled by the other's; the slices of a group follow each other on its stream, the groups meet again at the
    // detector (the renormalisation reads batch element 0)
    Group groups[BDOF_MAX_GROUPS];
    const int ng = batch_groups(c, B, (c->NX / 32) * (c->NY / 32), 2, groups);
    if ((r = fork_streams(c, ng))) return r;
    for (int gi = 0; gi < ng; ++gi) {
        use_group(c, groups[gi]);
        const size_t off = (size_t)groups[gi].b0 * plane;
        const ObjView gobj = sub_obj(c);
        cf* gcur = cur;
        for (int z = 0; z < c->S; ++z) {
            const bool last = z == c->S - 1;
            cf* out = last ? c->bufB : (tape ? c->tape + (size_t)(z + 1) * fld : (gcur == c->bufA ? c->bufC : c->bufA));
            ConvArgs a{gcur + off, out + off, nullptr, nullptr, gobj, groups[gi].B, c->NX, c->NY, last ? -1 : z + 1, cs ? zero : conv_pad(c, z),
                       conv_carrier(c, z + 1), c->k_conv, c->taps_dev + z % c->taps_copies, c->taps.ks,
                       cs && !last ? c->cstack + (size_t)(z + 1) * plane : nullptr};
            if ((r = launch_conv<false>(c, a))) { use_whole(c); return r; }
            gcur = out;
        }
    }
    if ((r = join_streams(c, ng))) return r;
    hipLaunchKernelGGL(k_conv_scalars, dim3(1), dim3(64), 0, c->stream, c->bufB, cs ? cfl(c->c_pS) : conv_carrier(c, c->S), c->probe,
                       cs ? cfl(c->c_p0) : conv_carrier(c, 0), c->conv_scal);
    return 0;
}

// carrier field: the renormalisation s = p_0[0,0] / (p_S[0,0] + eps_S[0,0,0]) in float64 (one 8-byte read-back)
static int conv_scale64(bdof_ctx* c, double2* s64) {
    cf e0;
    HIPC(c, hipMemcpyAsync(&e0, c->bufB, sizeof(cf), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    const std::complex<double> sd = c->c_p0 / (c->c_pS + std::complex<double>(e0.x, e0.y));
    *s64 = make_double2(sd.real(), sd.imag());
    return 0;
}

static int conv_check(bdof_ctx* c, int B, const int* angle_of_b) {
    int r = check_ready(c, B);
    if (r) return r;
    if (!c->have_conv) return fail(c, BDOF_ERR_STATE, "bdof_set_conv has not been called");
    if (c->obj.tab && !angle_of_b) return fail(c, BDOF_ERR_ARG, "angle_of_b required with a rotation table");
    return 0;
}


extern "C" {

int bdof_set_conv_probe_stack(bdof_ctx* c, const float* stack, const double* det64, double p0_re, double p0_im, double pS_re, double pS_im) {
    if (!c) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->cstack) { (void)hipFree(c->cstack); c->cstack = nullptr; }
    if (c->cdet64) { (void)hipFree(c->cdet64); c->cdet64 = nullptr; }
    if (!stack && !det64) return 0;
    if (!stack || !det64) return fail(c, BDOF_ERR_ARG, "bdof_set_conv_probe_stack: both arrays or neither");
    if (!c->have_conv) return fail(c, BDOF_ERR_STATE, "bdof_set_conv has not been called");
    const size_t plane = (size_t)c->NX * c->NY;
    HIPC(c, hipMalloc((void**)&c->cstack, sizeof(cf) * plane * (size_t)(c->S + 1)));
    HIPC(c, hipMalloc((void**)&c->cdet64, sizeof(double2) * plane));
    HIPC(c, hipMemcpy(c->cstack, stack, sizeof(cf) * plane * (size_t)(c->S + 1), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->cdet64, det64, sizeof(double2) * plane, hipMemcpyHostToDevice));
    c->c_p0 = std::complex<double>(p0_re, p0_im);
    c->c_pS = std::complex<double>(pS_re, pS_im);
    return 0;
}

int bdof_forward_conv(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, void* out_wave) {
    int r = conv_check(c, B, angle_of_b);
    if (r) return r;
    if (!out_wave) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    set_batch_views(c, angle_of_b, xoff, yoff);
    if ((r = ensure_modulation_k(c, c->k_conv))) return r;
    if ((r = conv_forward_sweep(c, B, false))) return r;
    c->tape_valid = c->last_valid = false;
    const size_t n = (size_t)B * c->NX * c->NY;
    const int egrid = (int)std::min<size_t>((n + 255) / 256, (size_t)c->ncu * 16);
    const cf zero = make_float2(0.f, 0.f);
    const bool cs = c->cstack != nullptr;
    const size_t plane = (size_t)c->NX * c->NY;
    const cf* pS = cs ? c->cstack + (size_t)c->S * plane : nullptr;
    if (c->det_mode == BDOF_DET_NONE) {
        ConvFinalArgs fa{c->bufB, (cf*)out_wave, nullptr, nullptr, nullptr, c->conv_scal, cs ? zero : conv_carrier(c, c->S), n, 0.f, 0, zero, 0.f, 0.f, pS, plane};
        hipLaunchKernelGGL((k_conv_final<0>), dim3(egrid), dim3(256), 0, c->stream, fa);
    } else {
        // carrier field: only the scattered part s eps_S goes through the float32 detector transforms; the carrier's detector
        // plane comes in float64, times s (conv_scale64)
        double2 s64 = kOne;
        if (cs && (r = conv_scale64(c, &s64))) return r;
        ConvFinalArgs fa{c->bufB, c->bufA, nullptr, nullptr, nullptr, c->conv_scal, cs ? zero : conv_carrier(c, c->S), n, 0.f, 0, zero, 0.f, 0.f, nullptr, plane};
        hipLaunchKernelGGL((k_conv_final<0>), dim3(egrid), dim3(256), 0, c->stream, fa);
        RealToHybArgs ra{c->bufA, c->bufC, B, c->NX, c->twY};
        DISPATCH_N(c->NY, { hipLaunchKernelGGL((k_row_real_to_hyb<N_>), dim3(rows_grid<N_>(c, B, c->NX)), dim3(BDOF_THREADS), 0, c->stream, ra); });
        if (c->det_mode == BDOF_DET_NEAR) {
            launch_row_prop(c, B, c->bufC, c->bufA, c->hdet, 1.f, 0);
            launch_loss_real(c, B, c->bufA, nullptr, false, (cf*)out_wave, nullptr, 1.f, 1.f, 0.f, zero, nullptr, nullptr, cs ? c->cdet64 : nullptr, s64);
        } else {
            const std::complex<double> keep = c->a0;      // the far-field kernel adds the carrier's DC bin: none here
            c->a0 = 0.0;
            launch_loss_far(c, B, c->bufC, nullptr, (cf*)out_wave, nullptr, 1.f, 1.f, 0.f, nullptr, cs ? c->cdet64 : nullptr, s64);
            c->a0 = keep;
        }
    }
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_loss_grad_conv(bdof_ctx* c, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, void* out_wave) {
    int r = conv_check(c, B, angle_of_b);
    if (r) return r;
    if (!meas) return BDOF_ERR_ARG;
    // residual splitting at the detector (bdof_set_meas_mode(1): meas holds m - |a_0|), real-space detectors: the renormalised
    // wave is kept as A + e', A = s a_S a float64 scalar formed on the host from the corner pixel, and |A + e'| - m is
    // evaluated without the cancellation of two numbers of size one (loss_seed_dev) — as on the transfer-function path
    const bool split = c->meas_dev != 0;
    if (split && (c->det_mode == BDOF_DET_FAR || std::abs(c->a0) == 0.0))
        return fail(c, BDOF_ERR_STATE, "bdof_set_meas_mode(1) needs a plane-wave carrier and a real-space detector");
    if (!c->with_grad) return fail(c, BDOF_ERR_STATE, "bdof_loss_grad_conv needs bdof_configure(with_grad=1)");
    c->gpsi_src = nullptr;                   // the real-space propagator does not export the probe gradient
    HIPC(c, hipSetDevice(c->device));
    set_batch_views(c, angle_of_b, xoff, yoff);
    if ((r = ensure_modulation_k(c, c->k_conv))) return r;
    if ((r = conv_forward_sweep(c, B, true))) return r;
    c->tape_valid = c->last_valid = false;
    const size_t fld = (size_t)c->Bmax * c->NX * c->NY;
    const size_t n = (size_t)B * c->NX * c->NY;
    const int egrid = (int)std::min<size_t>((n + 255) / 256, (size_t)c->ncu * 16);
    const double seed_scale = 2.0 / ((double)B * c->NX * c->NY);
    const cf zero = make_float2(0.f, 0.f);
    cf* gp;
    int npart;
    cf carA = zero;                 // split: A (times Hdet[0,0] for the near detector), the constant part of the detector wave
    float absA = 0.f, dref = 0.f;
    if (split) {
        cf e0, p0;                  // corner pixel of batch element 0: scattered part of psi_S and of the probe
        HIPC(c, hipMemcpyAsync(&e0, c->bufB, sizeof(cf), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipMemcpyAsync(&p0, c->probe, sizeof(cf), hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        const std::complex<double> aS = c->a0 * std::pow(c->ksum, c->S);
        const std::complex<double> sd = (c->a0 + std::complex<double>(p0.x, p0.y)) / (aS + std::complex<double>(e0.x, e0.y));
        std::complex<double> A = sd * aS;
        if (c->det_mode == BDOF_DET_NEAR) A *= c->hdet00;
        carA = make_float2((float)A.real(), (float)A.imag());
        absA = (float)std::abs(A);
        dref = (float)(std::abs(A) - std::abs(c->a0));
    }
    const bool cs = c->cstack != nullptr;
    const size_t plane = (size_t)c->NX * c->NY;
    double2 s64 = kOne;
    if (cs && (r = conv_scale64(c, &s64))) return r;
    const cf car_end = (split || cs) ? zero : conv_carrier(c, c->S);       // split / carrier field: only the scattered part s eps goes on
    if (c->det_mode == BDOF_DET_NONE && cs) {
        // q = s (p_S + eps_S): the scattered part in float32, the residual against s p_S in float64 (loss_seed_f64), seed in place
        ConvFinalArgs fa{c->bufB, c->bufA, nullptr, nullptr, nullptr, c->conv_scal, zero, n, 0.f, 0, zero, 0.f, 0.f, nullptr, plane};
        hipLaunchKernelGGL((k_conv_final<0>), dim3(egrid), dim3(256), 0, c->stream, fa);
        GLossArgs la{c->bufA, (cf*)out_wave, meas, c->partial, B, c->NX, c->NY, 0, zero, (float)seed_scale, c->cstack + (size_t)c->S * plane, 0, 0.f,
                     nullptr, nullptr, make_double2(0.0, 0.0), make_double2(0.0, 0.0), c->cdet64, nullptr, 0.0, s64};
        hipLaunchKernelGGL(k_g_loss, dim3(egrid), dim3(256), 0, c->stream, la);
        hipLaunchKernelGGL(k_conv_scale_seed, dim3(egrid), dim3(256), 0, c->stream, c->bufA, c->conv_scal, n);
        npart = egrid;
        gp = c->bufA;
    } else if (c->det_mode == BDOF_DET_NONE) {
        ConvFinalArgs fa{c->bufB, (cf*)out_wave, c->bufA, meas, c->partial, c->conv_scal, car_end, n, (float)seed_scale, split ? 1 : 0, carA, absA, dref, nullptr, plane};
        hipLaunchKernelGGL((k_conv_final<1>), dim3(egrid), dim3(256), 0, c->stream, fa);
        npart = egrid;
        gp = c->bufA;
    } else {
        ConvFinalArgs fa{c->bufB, c->bufA, nullptr, nullptr, nullptr, c->conv_scal, car_end, n, 0.f, 0, zero, 0.f, 0.f, nullptr, plane};
        hipLaunchKernelGGL((k_conv_final<0>), dim3(egrid), dim3(256), 0, c->stream, fa);
        RealToHybArgs ra{c->bufA, c->bufC, B, c->NX, c->twY};
        DISPATCH_N(c->NY, { hipLaunchKernelGGL((k_row_real_to_hyb<N_>), dim3(rows_grid<N_>(c, B, c->NX)), dim3(BDOF_THREADS), 0, c->stream, ra); });
        if (c->det_mode == BDOF_DET_NEAR) {
            launch_row_prop(c, B, c->bufC, c->bufA, c->hdet, 1.f, 0);                                     // d_hat (L1)
            npart = launch_loss_real(c, B, c->bufA, c->bufC, true, (cf*)out_wave, meas, 1.f, 1.f, (float)seed_scale, carA, nullptr,
                                     split ? &dref : nullptr, cs ? c->cdet64 : nullptr, s64);
            launch_row_prop(c, B, c->bufC, c->bufA, c->hdet, 1.f, 1);                                     // g_hat(q) (L1)
        } else {
            const std::complex<double> keep = c->a0;
            c->a0 = 0.0;
            npart = launch_loss_far(c, B, c->bufC, c->bufA, (cf*)out_wave, meas, 1.f, 1.f, (float)seed_scale, nullptr,
                                    cs ? c->cdet64 : nullptr, s64);                                           // g_hat(q) (L1)
            c->a0 = keep;
        }
        launch_loss_real(c, B, c->bufA, nullptr, false, c->bufC, nullptr, 1.f, 1.f, 0.f, zero);            // G(q), real space
        hipLaunchKernelGGL(k_conv_scale_seed, dim3(egrid), dim3(256), 0, c->stream, c->bufC, c->conv_scal, n);
        gp = c->bufC;
    }
    hipLaunchKernelGGL(k_conv_finish, dim3(1), dim3(256), 0, c->stream, c->partial, npart, 2, 1.0 / ((double)B * c->NX * c->NY),
                       seed_scale, c->loss_dev, gp, c->conv_scal);
    // backward sweep, in the same sub-batches
    Group groups[BDOF_MAX_GROUPS];
    const int ng = batch_groups(c, B, (c->NX / 32) * (c->NY / 32), 2, groups);
    if ((r = fork_streams(c, ng))) return r;
    for (int gi = 0; gi < ng; ++gi) {
        use_group(c, groups[gi]);
        const size_t off = (size_t)groups[gi].b0 * plane;
        const ObjView gobj = sub_obj(c);
        cf* gcur = gp;
        for (int z = c->S - 1; z >= 0; --z) {
            cf* gout = gcur == c->bufB ? c->bufA : c->bufB;
            ConvArgs a{gcur + off, gout + off, c->tape + (size_t)z * fld + off, c->grot + (size_t)groups[gi].b0 * c->S * plane, gobj, groups[gi].B, c->NX, c->NY, z,
                       zero, conv_carrier(c, z), c->k_conv, c->taps_dev + z % c->taps_copies, c->taps.ks, cs ? c->cstack + (size_t)z * plane : nullptr};
            if ((r = launch_conv<true>(c, a))) { use_whole(c); return r; }
            gcur = gout;
        }
    }
    if ((r = join_streams(c, ng))) return r;
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_enable_probe_grad(bdof_ctx* c, int enable) {
    if (!c) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipSetDevice(c->device));
    if (enable && !c->gpsi0) {
        if (!c->with_grad) return fail(c, BDOF_ERR_STATE, "the probe gradient needs bdof_configure(with_grad=1)");
        HIPC(c, hipMalloc((void**)&c->gpsi0, sizeof(cf) * (size_t)c->Bmax * c->NX * c->NY));
    } else if (!enable && c->gpsi0) {
        HIPC(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->gpsi0);
        c->gpsi0 = nullptr;
    }
    c->gpsi_src = nullptr;
    return 0;
}

int bdof_probe_grad(bdof_ctx* c, void* out, int accumulate) {
    if (!c || !out) return BDOF_ERR_ARG;
    if (!c->gpsi0) return fail(c, BDOF_ERR_STATE, "bdof_enable_probe_grad(1) has not been called");
    if (!c->gpsi_src) return fail(c, BDOF_ERR_STATE, "no gradient sweep since the probe gradient was enabled (bdof_loss_grad first)");
    HIPC(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->NX * c->NY;
    hipLaunchKernelGGL(k_sum_fields, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)c->ncu * 8)), dim3(256), 0, c->stream, c->gpsi_src,
                       (cf*)out, c->gpsi_B, n, accumulate);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_get_loss(bdof_ctx* c, double* loss) {
    if (!c || !loss) return BDOF_ERR_ARG;
    HIPC(c, hipMemcpyAsync(loss, c->loss_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

void* bdof_grot(bdof_ctx* c) { return c ? (void*)c->grot : nullptr; }

int bdof_rotation_adjoint_rows(bdof_ctx* c, int B, const int* angle_of_b, void* gvol, int row0, int n_rows, int accumulate,
                               float scale) {
    if (!c || !gvol || !angle_of_b) return BDOF_ERR_ARG;
    if (!c->grot) return fail(c, BDOF_ERR_STATE, "no gradient workspace (configure with_grad=1)");
    if (!c->adj_off) return fail(c, BDOF_ERR_STATE, "bdof_set_rotation_adjoint has not been called");
    if (B < 1 || B > c->Bmax) return fail(c, BDOF_ERR_ARG, "batch size outside [1, Bmax]");
    if (c->NY % 2) return fail(c, BDOF_ERR_SIZE, "the rotation adjoint needs an even NY");
    if (row0 < 0 || n_rows < 0 || (long long)row0 + n_rows > c->adj_ndest) return fail(c, BDOF_ERR_ARG, "destination rows outside the volume");
    if (n_rows == 0) return 0;
    HIPC(c, hipSetDevice(c->device));
    ProfScope ps(c, BDOF_K_ROT_ADJ);
    RotAdjArgs a{c->grot, (float2*)gvol, c->adj_off, c->adj_order, angle_of_b, B, c->S * c->NX, c->adj_ndest, c->NY, accumulate, scale,
                 c->heavy, c->heavy + 1, row0, row0 + n_rows};
    HIPC(c, hipMemsetAsync(c->heavy, 0, sizeof(int), c->stream));
    int need = (n_rows + 3) / 4;
    int grid = need < c->ncu * 8 ? need : c->ncu * 8;
    hipLaunchKernelGGL(k_rot_adjoint, dim3(grid), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_rot_adjoint_heavy, dim3(c->ncu * 8), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_rotation_adjoint(bdof_ctx* c, int B, const int* angle_of_b, void* gvol, int accumulate, float scale) {
    if (!c) return BDOF_ERR_ARG;
    return bdof_rotation_adjoint_rows(c, B, angle_of_b, gvol, 0, c->adj_ndest, accumulate, scale);
}

int bdof_window_rotation_adjoint(bdof_ctx* c, int B, int angle, const int* xoff, const int* yoff, void* gvol,
                                 int accumulate, float scale) {
    if (!c || !gvol || !xoff || !yoff) return BDOF_ERR_ARG;
    if (!c->grot) return fail(c, BDOF_ERR_STATE, "no gradient workspace (configure with_grad=1)");
    if (!c->adj_off || !c->obj.tab) return fail(c, BDOF_ERR_STATE, "rotation tables have not been set");
    if (B < 1 || B > c->Bmax) return fail(c, BDOF_ERR_ARG, "batch size outside [1, Bmax]");
    if (angle < 0 || angle >= c->n_angles) return fail(c, BDOF_ERR_ARG, "angle index outside the rotation tables");
    HIPC(c, hipSetDevice(c->device));
    if (c->obj.volNY % 2) return fail(c, BDOF_ERR_SIZE, "the rotation adjoint needs an even volume NY");
    ProfScope ps(c, BDOF_K_ROT_ADJ);
    const int n_src = c->S * c->obj.volNX;
    WinAdjArgs a{c->grot, (float2*)gvol, c->adj_off + (size_t)angle * (c->adj_ndest + 1), c->adj_order + (size_t)angle * n_src,
                 xoff, yoff, B, c->S, c->NX, c->NY, c->obj.volNX, c->obj.volNY, c->adj_ndest, accumulate, scale};
    if (B > BDOF_WIN_MAXLIST) {
        // more windows than the overlap-add kernel lists per column: direct (slow) gather
        int grid = c->adj_ndest < c->ncu * 16 ? c->adj_ndest : c->ncu * 16;
        hipLaunchKernelGGL(k_window_rot_adjoint, dim3(grid), dim3(256), 0, c->stream, a);
        HIPC(c, hipGetLastError());
        return 0;
    }
    const size_t need = sizeof(float2) * (size_t)n_src * c->obj.volNY;
    if (c->winpad_sz < need) {
        HIPC(c, hipStreamSynchronize(c->stream));
        if (c->winpad) (void)hipFree(c->winpad);
        c->winpad = nullptr; c->winpad_sz = 0;
        HIPC(c, hipMalloc((void**)&c->winpad, need));
        c->winpad_sz = need;
    }
    if (!c->win_angle) HIPC(c, hipMalloc((void**)&c->win_angle, sizeof(int)));
    HIPC(c, hipMemsetD32Async((hipDeviceptr_t)c->win_angle, angle, 1, c->stream));
    // stage 1: windows -> rotated frame
    const int z_per_wg = 8;
    hipLaunchKernelGGL(k_window_overlap_add, dim3(c->obj.volNX, (c->S + z_per_wg - 1) / z_per_wg), dim3(256), 0, c->stream, a,
                       c->winpad, z_per_wg);
    // stage 2: rotated frame -> volume (the full-field rotation adjoint with a batch of one)
    RotAdjArgs r{c->winpad, (float2*)gvol, c->adj_off, c->adj_order, c->win_angle, 1, n_src, c->adj_ndest, c->obj.volNY, accumulate,
                 scale, c->heavy, c->heavy + 1, 0, c->adj_ndest};
    HIPC(c, hipMemsetAsync(c->heavy, 0, sizeof(int), c->stream));
    const int needwg = (c->adj_ndest + 3) / 4;
    hipLaunchKernelGGL(k_rot_adjoint, dim3(needwg < c->ncu * 8 ? needwg : c->ncu * 8), dim3(256), 0, c->stream, r);
    hipLaunchKernelGGL(k_rot_adjoint_heavy, dim3(c->ncu * 8), dim3(256), 0, c->stream, r);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_adam_step_slab(bdof_ctx* c, const void* x_old, void* x_new, const void* g, void* m, void* v, const float* mask,
                        int NXv, int NZv, int NYv, float g_scale, float alpha_d, float alpha_b, float gamma,
                        float lr, float b1, float b2, float eps, int i_batch, int clip, int x0, int nx) {
    if (!c || !x_old || !x_new || !g || !m || !v) return BDOF_ERR_ARG;
    if (x_old == x_new) return fail(c, BDOF_ERR_ARG, "x_new must not alias x_old (the TV stencil reads pre-update neighbours)");
    if (NXv < 1 || NZv < 1 || NYv < 1 || i_batch < 0) return fail(c, BDOF_ERR_ARG, "bad volume shape / i_batch");
    if (x0 < 0 || nx < 0 || (long long)x0 + nx > NXv) return fail(c, BDOF_ERR_ARG, "slab outside the volume");
    if (nx == 0) return 0;
    HIPC(c, hipSetDevice(c->device));
    ProfScope ps(c, BDOF_K_ADAM);
    // b1, b2 arrive as float32 (0.999f = 0.99900001287...); the reference's are Python floats.  The decimal the caller meant is
    // recovered (7 significant digits) so that 1 - b and the bias corrections are those of the float64 reference.
    const double b1d = std::round((double)b1 * 1e7) / 1e7, b2d = std::round((double)b2 * 1e7) / 1e7;
    const double bc1 = 1.0 - std::pow(b1d, (double)(i_batch + 1));
    const double bc2 = 1.0 - std::pow(b2d, (double)(i_batch + 1));
    AdamArgs a{(const float2*)x_old, (float2*)x_new, (const float2*)g, (float2*)m, (float2*)v, mask, NXv, NZv, NYv,
               g_scale, alpha_d, alpha_b, gamma, lr, (float)b1d, (float)b2d, eps, (float)(1.0 / bc1), (float)(1.0 / bc2),
               (float)(1.0 - b1d), (float)(1.0 - b2d), clip, x0, x0 + nx};
    const size_t n = (size_t)nx * NZv * NYv;
    size_t need = (n + 255) / 256;
    int grid = need < (size_t)c->ncu * 16 ? (int)need : c->ncu * 16;
    grid = (grid + 7) / 8 * 8;                       // k_adam deals the range to the 8 XCDs by blockIdx % 8
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, c->stream, a);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_adam_step(bdof_ctx* c, const void* x_old, void* x_new, const void* g, void* m, void* v, const float* mask,
                   int NXv, int NZv, int NYv, float g_scale, float alpha_d, float alpha_b, float gamma,
                   float lr, float b1, float b2, float eps, int i_batch, int clip) {
    return bdof_adam_step_slab(c, x_old, x_new, g, m, v, mask, NXv, NZv, NYv, g_scale, alpha_d, alpha_b, gamma, lr, b1, b2, eps,
                               i_batch, clip, 0, NXv);
}

int bdof_rotate_bilinear(bdof_ctx* c, const void* vol, int NXv, int NZv, int NYv, const double* prm, int B, void* out_rows) {
    if (!c || !vol || !prm || !out_rows || B < 1) return BDOF_ERR_ARG;
    if (NXv < 1 || NZv < 1 || NYv < 2 || NYv % 2) return fail(c, BDOF_ERR_SIZE, "bdof_rotate_bilinear needs an even NY");
    HIPC(c, hipSetDevice(c->device));
    RotBilinArgs a{(const float2*)vol, (float2*)out_rows, nullptr, (const double4*)prm, B, NXv, NZv, NYv, 0, 0, 0, 1.f};
    const size_t nrows = (size_t)B * NZv * NXv;
    const int grid = (int)std::min<size_t>((nrows + 3) / 4, (size_t)c->ncu * 16);
    hipLaunchKernelGGL((k_rot_bilinear<false>), dim3(grid), dim3(256), 0, c->stream, a, 0.f, (double2*)nullptr);
    HIPC(c, hipGetLastError());
    return 0;
}

// bdof_rotate_bilinear + bdof_set_object in one pass: the B rotated objects are written straight into the ctx's modulation
// table as factors c - 1 (no rotated (delta, beta) copy, no second pass over B volumes) and bound as the batch's objects.
int bdof_set_object_bilinear(bdof_ctx* c, const void* vol, int NXv, int NZv, int NYv, const double* prm, int B, int conv) {
    if (!c || !vol || !prm || B < 1) return BDOF_ERR_ARG;
    if (c->NY == 0) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    if (NXv < 1 || NZv < 1 || NYv < 2 || NYv % 2) return fail(c, BDOF_ERR_SIZE, "bdof_set_object_bilinear needs an even NY");
    if (NYv != c->NY || NXv != c->NX || NZv != c->S) return fail(c, BDOF_ERR_ARG, "the volume must be (NX, S, NY) of the configured wavefields");
    if (B > c->Bmax) return fail(c, BDOF_ERR_ARG, "batch size outside [1, Bmax]");
    if (conv && !c->have_conv) return fail(c, BDOF_ERR_STATE, "bdof_set_conv has not been called");
    HIPC(c, hipSetDevice(c->device));
    const size_t nrows = (size_t)B * NZv * NXv, n = nrows * NYv;
    // set the binding first: want_cbar looks at the physics and probe only
    c->obj_src = nullptr;
    c->obj_bound_mod = true;
    c->obj_rows = nrows;
    c->obj.volNY = NYv;
    c->obj.tab = nullptr;
    c->obj.volNX = c->NX;
    c->obj.S = c->S;
    c->n_angles = 0;
    c->k = conv ? c->k_conv : c->k_fft;
    const bool mean = want_cbar(c);
    int r = modulation_room(c, n, mean);
    if (r) return r;
    RotBilinArgs a{(const float2*)vol, c->mod, nullptr, (const double4*)prm, B, NXv, NZv, NYv, 0, 0, 0, 1.f};
    const int grid = (int)std::min<size_t>((nrows + 3) / 4, (size_t)c->ncu * 16);
    hipLaunchKernelGGL((k_rot_bilinear<true>), dim3(grid), dim3(256), 0, c->stream, a, c->k, mean ? c->cbar_dev : (double2*)nullptr);
    return modulation_done(c, n, grid, mean);
}

int bdof_rotate_bilinear_adjoint(bdof_ctx* c, const void* grot, int NXv, int NZv, int NYv, const double* prm, int B, void* gvol, int row0,
                                 int n_rows, int accumulate, float scale) {
    if (!c || !grot || !prm || !gvol || B < 1) return BDOF_ERR_ARG;
    if (NXv < 1 || NZv < 1 || NYv < 2 || NYv % 2) return fail(c, BDOF_ERR_SIZE, "bdof_rotate_bilinear_adjoint needs an even NY");
    if (row0 < 0 || n_rows < 0 || (long long)row0 + n_rows > (long long)NXv * NZv) return fail(c, BDOF_ERR_ARG, "destination rows outside the volume");
    if (n_rows == 0) return 0;
    HIPC(c, hipSetDevice(c->device));
    ProfScope ps(c, BDOF_K_ROT_ADJ);
    RotBilinArgs a{nullptr, (float2*)grot, (float2*)gvol, (const double4*)prm, B, NXv, NZv, NYv, row0, row0 + n_rows, accumulate, scale};
    const int grid = std::min((n_rows + 3) / 4, c->ncu * 16);
    const int nv4 = (NYv / 2 + 63) / 64;
    if (nv4 <= 1) hipLaunchKernelGGL((k_rot_bilinear_adjoint<1>), dim3(grid), dim3(256), 0, c->stream, a);
    else if (nv4 <= 2) hipLaunchKernelGGL((k_rot_bilinear_adjoint<2>), dim3(grid), dim3(256), 0, c->stream, a);
    else if (nv4 <= 4) hipLaunchKernelGGL((k_rot_bilinear_adjoint<4>), dim3(grid), dim3(256), 0, c->stream, a);
    else if (nv4 <= 8) hipLaunchKernelGGL((k_rot_bilinear_adjoint<8>), dim3(grid), dim3(256), 0, c->stream, a);
    else return fail(c, BDOF_ERR_SIZE, "bdof_rotate_bilinear_adjoint: NY <= 1024");
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_regularizer_value(bdof_ctx* c, const void* x, int NXv, int NZv, int NYv, double* sums) {
    if (!c || !x || !sums) return BDOF_ERR_ARG;
    if (NXv < 1 || NZv < 1 || NYv < 1) return fail(c, BDOF_ERR_ARG, "bad volume shape");
    if (!c->partial) return fail(c, BDOF_ERR_STATE, "bdof_configure has not been called");
    HIPC(c, hipSetDevice(c->device));
    const size_t n = (size_t)NXv * NZv * NYv;
    const int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)(2 * c->npartial / 3));
    hipLaunchKernelGGL(k_reg_value, dim3(grid), dim3(256), 0, c->stream, (const float2*)x, NXv, NZv, NYv, c->partial);
    HIPC(c, hipGetLastError());
    std::vector<double> p(3 * (size_t)grid);
    HIPC(c, hipMemcpyAsync(p.data(), c->partial, sizeof(double) * p.size(), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    sums[0] = sums[1] = sums[2] = 0.0;
    for (int b = 0; b < grid; ++b) for (int j = 0; j < 3; ++j) sums[j] += p[3 * (size_t)b + j];
    return 0;
}

void* bdof_stream(bdof_ctx* c) { return c ? (void*)c->stream : nullptr; }

int bdof_mask_shrink(bdof_ctx* c, const void* x, float* mask, size_t n, float thresh) {
    if (!c || !x || !mask) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    size_t need = (n + 255) / 256;
    int grid = need < (size_t)c->ncu * 16 ? (int)need : c->ncu * 16;
    hipLaunchKernelGGL(k_mask_shrink, dim3(grid), dim3(256), 0, c->stream, (const float2*)x, mask, n, thresh);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_gather_fields(bdof_ctx* c, void* dst, const void* src, const int* idx, int B, size_t bytes_per_field) {
    if (!c || !dst || !src || !idx || B < 1) return BDOF_ERR_ARG;
    if (bytes_per_field % 4) return fail(c, BDOF_ERR_SIZE, "bdof_gather_fields: fields must be multiples of 4 bytes");
    HIPC(c, hipSetDevice(c->device));
    if (bytes_per_field % 16) {
        const size_t n4 = bytes_per_field / 4;
        const int gx4 = (int)std::min<size_t>((n4 + 255) / 256, 64);
        hipLaunchKernelGGL(k_gather_fields4, dim3(gx4, B < 1024 ? B : 1024), dim3(256), 0, c->stream, (float*)dst, (const float*)src, idx, B, n4);
        HIPC(c, hipGetLastError());
        return 0;
    }
    const size_t n16 = bytes_per_field / 16;
    const int gx = (int)std::min<size_t>((n16 + 255) / 256, 64);
    hipLaunchKernelGGL(k_gather_fields, dim3(gx, B < 1024 ? B : 1024), dim3(256), 0, c->stream, (float4*)dst, (const float4*)src, idx, B, n16);
    HIPC(c, hipGetLastError());
    return 0;
}

int bdof_set_streams(bdof_ctx* c, int n) {
    if (!c) return -1;
    if (n == 0 || n > BDOF_MAX_GROUPS) return fail(c, -2, "bdof_set_streams: n must be -1 or 1..4");
    c->n_streams = n < 0 ? -1 : n;
    return 0;
}

int bdof_batch_groups(bdof_ctx* c, int B) {
    if (!c || B < 1 || !c->NX) return -1;
    if (c->generic) return 1;
    Group g[BDOF_MAX_GROUPS];
    return batch_groups(c, B, c->NX, 16, g);
}

int bdof_profile_enable(bdof_ctx* c, int enable) {
    if (!c) return BDOF_ERR_ARG;
    HIPC(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (enable) { for (int i = 0; i < BDOF_K_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_seen[i] = 0; } }
    c->prof = enable != 0;
    c->prof_stride = enable > 1 ? enable : 1;
    return 0;
}

int bdof_profile_read(bdof_ctx* c, int kernel_class, int* n_launches, double* total_ms) {
    if (!c || kernel_class < 0 || kernel_class >= BDOF_K_COUNT) return BDOF_ERR_ARG;
    HIPC(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (n_launches) *n_launches = c->prof_n[kernel_class];
    if (total_ms) *total_ms = c->prof_ms[kernel_class];
    return 0;
}

int bdof_device_mem(bdof_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    if (!c || !free_bytes || !total_bytes) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}

int bdof_malloc(void** ptr, size_t bytes) {
    if (!ptr) return BDOF_ERR_ARG;
    return (int)hipMalloc(ptr, bytes);
}
int bdof_ctx_malloc(bdof_ctx* c, void** ptr, size_t bytes) {
    if (!c || !ptr) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMalloc(ptr, bytes));
    return 0;
}
int bdof_free(void* ptr) { return (int)hipFree(ptr); }
int bdof_memcpy_h2d(bdof_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int bdof_memcpy_d2h(bdof_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int bdof_memset(bdof_ctx* c, void* dst, int value, size_t bytes) {
    if (!c) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return 0;
}

int bdof_memcpy_d2d(bdof_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return BDOF_ERR_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}


// =================================================================================================
// Collectives (RCCL over xGMI): comm.Allreduce(this_grads, grads) of cnn_propagator/fullfield.py:348-351
// =================================================================================================
static int comm_fail(bdof_comm* m, int code, const std::string& msg) {
    if (m) m->err = msg;
    g_rccl.err = msg;
    return code;
}
#define NCCLC(m, call)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return comm_fail((m), 1000 + (int)r_, std::string(#call) + ": " + g_rccl.GetErrorString(r_));  \
    } while (0)
#define HIPM(m, call)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return comm_fail((m), (int)e_, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int bdof_comm_unique_id(void* id, size_t bytes) {
    if (!id || bytes < sizeof(ncclUniqueId)) return BDOF_ERR_ARG;
    if (!rccl_load()) return BDOF_ERR_STATE;
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) { g_rccl.err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return 1000 + (int)r; }
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

const char* bdof_comm_last_error(const bdof_comm* m) { return m ? m->err.c_str() : g_rccl.err.c_str(); }

void bdof_comm_destroy(bdof_comm* m);

int bdof_comm_create(bdof_comm** out, int device, int nranks, int rank, const void* id, size_t bytes) {
    if (!out || !id || bytes < sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks) return BDOF_ERR_ARG;
    *out = nullptr;
    if (!rccl_load()) return BDOF_ERR_STATE;
    bdof_comm* m = new bdof_comm();
    m->device = device; m->nranks = nranks; m->rank = rank;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming);
    for (int i = 0; e == hipSuccess && i < BDOF_COMM_TICKETS; ++i) e = hipEventCreateWithFlags(&m->ticket[i], hipEventDisableTiming);
    if (e != hipSuccess) { g_rccl.err = std::string("bdof_comm_create: ") + hipGetErrorString(e); bdof_comm_destroy(m); return (int)e; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = g_rccl.CommInitRank(&m->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        g_rccl.err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
        m->comm = nullptr;
        bdof_comm_destroy(m);                  // releases the stream and the events created above
        return 1000 + (int)r;
    }
    *out = m;
    return 0;
}

void bdof_comm_destroy(bdof_comm* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm) (void)g_rccl.CommDestroy(m->comm);
    for (int i = 0; i < BDOF_COMM_TICKETS; ++i) if (m->ticket[i]) (void)hipEventDestroy(m->ticket[i]);
    if (m->ev_in) (void)hipEventDestroy(m->ev_in);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int bdof_comm_size(const bdof_comm* m) { return m ? m->nranks : 0; }
int bdof_comm_rank(const bdof_comm* m) { return m ? m->rank : -1; }

// the communicator's stream picks up behind everything the ctx stream holds now
static int comm_enter(bdof_comm* m, bdof_ctx* c) {
    if (!m || !c) return BDOF_ERR_ARG;
    if (m->device != c->device) return comm_fail(m, BDOF_ERR_ARG, "communicator and ctx live on different devices");
    HIPM(m, hipSetDevice(m->device));
    HIPM(m, hipEventRecord(m->ev_in, c->stream));
    HIPM(m, hipStreamWaitEvent(m->stream, m->ev_in, 0));
    return 0;
}
static int comm_leave(bdof_comm* m, int* ticket) {
    const unsigned t = m->next++ % BDOF_COMM_TICKETS;
    HIPM(m, hipEventRecord(m->ticket[t], m->stream));
    if (ticket) *ticket = (int)t;
    return 0;
}

int bdof_allreduce_grad(bdof_comm* m, bdof_ctx* c, void* buf, size_t count, int* ticket) {
    int r = comm_enter(m, c);
    if (r) return r;
    if (!buf) return comm_fail(m, BDOF_ERR_ARG, "null buffer");
    if (count) NCCLC(m, g_rccl.AllReduce(buf, buf, count, ncclFloat, ncclSum, m->comm, m->stream));
    return comm_leave(m, ticket);
}

int bdof_reduce_scatter_grad(bdof_comm* m, bdof_ctx* c, void* buf, size_t count_per_rank, int* ticket) {
    int r = comm_enter(m, c);
    if (r) return r;
    if (!buf) return comm_fail(m, BDOF_ERR_ARG, "null buffer");
    if (count_per_rank)
        NCCLC(m, g_rccl.ReduceScatter(buf, (float*)buf + (size_t)m->rank * count_per_rank, count_per_rank, ncclFloat, ncclSum, m->comm, m->stream));
    return comm_leave(m, ticket);
}

int bdof_allgather_volume(bdof_comm* m, bdof_ctx* c, void* buf, size_t count_per_rank, int* ticket) {
    int r = comm_enter(m, c);
    if (r) return r;
    if (!buf) return comm_fail(m, BDOF_ERR_ARG, "null buffer");
    if (count_per_rank)
        NCCLC(m, g_rccl.AllGather((const float*)buf + (size_t)m->rank * count_per_rank, buf, count_per_rank, ncclFloat, m->comm, m->stream));
    return comm_leave(m, ticket);
}

int bdof_bcast_volume(bdof_comm* m, bdof_ctx* c, void* buf, size_t count, int root, int* ticket) {
    int r = comm_enter(m, c);
    if (r) return r;
    if (!buf || root < 0 || root >= m->nranks) return comm_fail(m, BDOF_ERR_ARG, "bad buffer / root");
    if (count) NCCLC(m, g_rccl.Broadcast(buf, buf, count, ncclFloat, root, m->comm, m->stream));
    return comm_leave(m, ticket);
}

int bdof_comm_wait(bdof_comm* m, bdof_ctx* c, int ticket) {
    if (!m || !c || ticket < 0 || ticket >= BDOF_COMM_TICKETS) return BDOF_ERR_ARG;
    HIPM(m, hipSetDevice(m->device));
    HIPM(m, hipStreamWaitEvent(c->stream, m->ticket[ticket], 0));
    return 0;
}

int bdof_comm_sync(bdof_comm* m) {
    if (!m) return BDOF_ERR_ARG;
    HIPM(m, hipSetDevice(m->device));
    HIPM(m, hipStreamSynchronize(m->stream));
    return 0;
}

}  // extern "C"
