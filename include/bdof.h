/* bdof.h — C ABI of libbdof.so: MI355X-native multislice Fresnel forward + adjoint + Adam.
 *
 * The reference (mdw771/beyond_dof) has no FFI layer; its only seam is a set of Python call
 * signatures.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).  The Python host in beyond_dof_amd/ binds these
 * with ctypes and re-exposes the reference's own function names.
 *
 * Conventions
 *   - every call returns int: 0 ok, <0 bdof argument/state error, >0 hipError_t; text via
 *     bdof_last_error(ctx).
 *   - device buffers passed in are caller-owned raw device pointers; the library never frees them.
 *   - one ctx <-> one device <-> one stream; calls are asynchronous on that stream until
 *     bdof_sync / bdof_get_loss; a ctx is not thread-safe.
 *   - complex = interleaved float (re, im); "pair" = interleaved float (delta, beta).
 *
 * Internal data layout:
 *   wavefields        [b][x][y]            y fastest (y = tomographic rotation axis, reference axis 0)
 *   object rows       [row][y]   of pairs  full-field: row = x*Z + z of the un-rotated volume
 *   rotated gradient  [b][z][x][y] of pairs
 */
#ifndef BDOF_H
#define BDOF_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bdof_ctx bdof_ctx;

enum { BDOF_DET_NONE = 0, BDOF_DET_NEAR = 1, BDOF_DET_FAR = 2 };      /* free_prop_cm None / float / 'inf'  (np_funcs.py:45-61) */
enum { BDOF_VARIANT_NUMPY_SKIP_LAST = 0, BDOF_VARIANT_TF_ALL = 1 };   /* np_funcs.py:41 vs tensorflow_recon/util.py:465-483 */
enum { BDOF_K_ROW_FWD = 0, BDOF_K_COL_PROP = 1, BDOF_K_ROW_BWD = 2, BDOF_K_LOSS = 3, BDOF_K_ROT_ADJ = 4, BDOF_K_ADAM = 5, BDOF_K_COUNT = 6 };

/* lifecycle.  `stream` is a hipStream_t (NULL: the library creates its own). */
int bdof_ctx_create(bdof_ctx** out, int device, void* stream);
void bdof_ctx_destroy(bdof_ctx* ctx);
const char* bdof_last_error(const bdof_ctx* ctx);
int bdof_sync(bdof_ctx* ctx);
void* bdof_stream(bdof_ctx* ctx);   /* the ctx's hipStream_t, for ordering foreign work (collectives) against it */
int bdof_device_count(void);

/* Host only (no device is touched): the twiddle tables bdof_configure uploads for transforms of length N, as D copies of
 * [N hi (re, im)][N lo (re, im)] float32 — exp(-2 pi i j / N) with lo = the rounding error of hi.  D >= 2: the dithered copies
 * (entry j of copy d is the float32 below or above the float64 value, the upper one in the fraction of the copies that makes the
 * mean over the copies equal the float64 value to ulp / D; DESIGN §5 "Dithered transform constants"); D = 0 or 1: one copy,
 * rounded to nearest with the modulus kept closest to one.  out: 4 N max(D, 1) floats.  Exposed for the tests. */
int bdof_twiddle_tables(int N, int D, float* out);
int bdof_device_pci_bus_id(int device, char* out, int len);      /* "0000:c1:00.0": which physical GPU a rank really got */
/* Stream-ordered time stamps on the ctx stream (16 slots): mark now, read the interval after the work has run — how the
 * bench reports the tail of a step (rotation adjoint, gradient exchange, Adam) without a host synchronisation inside it. */
int bdof_timer_mark(bdof_ctx* ctx, int slot);
int bdof_timer_elapsed(bdof_ctx* ctx, int slot_a, int slot_b, double* ms);

/* Workspace for wavefields of NY x NX, S slices, up to Bmax wavefields per launch.  Three engines sit behind the same calls:
 *   - fused streaming kernels (hand-written FFTs, field in HBM): NY, NX powers of two in 64..1024;
 *   - LDS-resident kernel (the whole field stays in one CU's LDS through all slices, one launch per minibatch): square
 *     fields of 32, 36, 48, 64, 72, 80, 96 or 128 pixels — the probes of the ptychography drivers
 *     (cnn_propagator/reconstruct_ptycho.py:106); chosen when no fused plan exists or the batch has >= CUs/4 wavefields;
 *   - generic-size engine (rocFFT + point-wise kernels): every other size.
 * with_grad bits: 0 allocate the tape (S fields per wavefield) and the rotated-frame gradient; 1 force the generic-size
 * engine (cross-checks); 2 never use the resident engine; 3 use it for every batch size; 4 (value 16) tape-free adjoint of the
 * streaming engine: 3 tape fields instead of S, the forward wave is marched back beside the adjoint field;
 * 5 (value 32) no rotated-frame gradient workspace (range sweeps with caller-owned buffers, bdof_adjoint_range);
 * 6 (value 64) float64 adjoint sweep (accuracy option; runs on the generic-size engine whatever the size): seed, adjoint
 * transforms (rocFFT double precision), transfer function and the products conj(phi) G in float64, forward sweep and tape in
 * float32 — what autograd's float64 tape gives the reference (cnn_propagator/fullfield.py:329, ptychography.py:248);
 * follow bdof_set_physics with bdof_set_physics_f64.
 * Environment read here: BDOF_TW_DITHER=D — number of dithered copies of the transform constants the per-slice kernels walk
 * (default 64; 0: one plain float32 table, round 2's behaviour; DESIGN §5 "Dithered transform constants").
 * Replaces the per-call allocations of multislice_propagate_batch_numpy
 * (cnn_propagator/np_funcs.py:20,43) and of autograd's tape (cnn_propagator/fullfield.py:329). */
int bdof_configure(bdof_ctx* ctx, int NY, int NX, int S, int Bmax, int with_grad);

/* Physics.  k = 2*PI*delta_nm/lambda_nm (np_funcs.py:32).  hs / hs_det: HOST arrays [NX][NY] complex,
 * hs[kx][ky] = ifftshift(get_kernel(...))[ky][kx] / (NX*NY)   (cnn_propagator/util.py:82-102, np_funcs.py:42);
 * the host computes them in float64 exactly as the reference does and rounds once to float32.
 * hs_det may be NULL unless det_mode == BDOF_DET_NEAR.  h00 / hdet00: (re, im) of the un-scaled DC values
 * ifftshift(H)[0][0] in float64 — the factor a constant wave picks up in one step (carrier splitting, below). */
int bdof_set_physics(bdof_ctx* ctx, double k, const float* hs, const float* hs_det, const double* h00, const double* hdet00,
                     int det_mode, int variant);

/* The same two tables in float64 (HOST arrays of complex128, layout and 1/(NX*NY) scaling of hs / hs_det) for the float64
 * adjoint sweep of bdof_configure flag 64; call after every bdof_set_physics. */
int bdof_set_physics_f64(bdof_ctx* ctx, const double* hs, const double* hs_det);

/* Probe wavefront (np_funcs.py:20-21; cnn_propagator/fullfield.py:276-314), split as probe = a0 + eps: `probe_eps` is
 * the HOST array [NX][NY] complex of eps, a0 any complex constant (0 for a general probe, the plane-wave amplitude for a
 * plane probe).  The library carries a0 through the slices exactly (a_{z+1} = a_z * h00, float64 on the host) and runs
 * only eps through the float32 FFTs, so round-off scales with the scattered field, not with the full wave. */
int bdof_set_probe(bdof_ctx* ctx, const float* probe_eps, double a0_re, double a0_im);

/* Carrier FIELD for a localised probe (all three engines of the transfer-function path).  stack: HOST array [S][NX][NY] complex, the probe propagated
 * through free space to the entrance of every slice, p_0 = probe, p_{z+1} = ifft2(ifftshift(fftshift(fft2 p_z) H))
 * (np_funcs.py:42 without an object), computed by the host in float64; det: HOST [NX][NY], the same wave at the detector
 * (no detector: p_{S-1}, or p_S with the tf_all variant; near field: one more step with the detector kernel; far field:
 * the un-shifted, un-normalised fft2 of it, indexed [kx][ky]).  The wave is then carried as psi_z = p_z + eps_z and only
 * the scattered part eps runs through the float32 transforms, whose round-off therefore scales with the scattered field —
 * what the scalar a0 of bdof_set_probe does for a plane wave.  Call bdof_set_probe with a zero array (eps_0 = 0, a0 = 0).
 * NULL, NULL removes the stack.  bdof_probe_stack_supported: 1 once bdof_configure has run (the real-space propagator of
 * bdof_set_conv does not use the stack). */
int bdof_probe_stack_supported(bdof_ctx* ctx);
int bdof_set_probe_stack(bdof_ctx* ctx, const float* stack, const float* det);

/* The same carrier field computed by the library, on the device, in float64: probe HOST [NX][NY] complex128; hT / hdetT HOST
 * [kx][ky] complex128 = ifftshift(get_kernel(...)) transposed, NOT divided by NX*NY (hdetT: the detector distance, NULL unless
 * det_mode == BDOF_DET_NEAR).  Replaces bdof_set_probe + bdof_set_probe_stack for a localised probe: no host FFTs, so a probe
 * that changes every step (probe_type='optimizable', tensorflow_recon/fullfield.py:311-327) stays cheap. */
int bdof_set_probe_field(bdof_ctx* ctx, const double* probe, const double* hT, const double* hdetT);

/* Object.  vol: n_rows device rows of volNY (delta, beta) pairs.  Call again whenever that memory has been modified: the
 * library keeps a table of the modulation factors exp(i k delta - k beta) - 1 of these rows and rebuilds it lazily.
 *  tab == NULL: row(b,z,x) = (b*S+z)*NX+x, i.e. the caller
 * supplies already rotated objects (the grid_delta_batch/grid_beta_batch arguments of
 * multislice_propagate_batch_numpy).  tab != NULL (device, [n_angles][S][volNX] int32): fused
 * nearest-neighbour rotation gather, row = tab[angle][z][x]  (apply_rotation, cnn_propagator/util.py:377-402
 * with the tables of save_rotation_lookup, util.py:294-347). */
int bdof_set_object(bdof_ctx* ctx, const void* vol, long long n_rows, int volNY, const int* tab, int volNX, int n_angles);

/* Inverse rotation tables for the gradient (device): off [n_angles][n_dest+1], order [n_angles][S*volNX]. */
int bdof_set_rotation_adjoint(bdof_ctx* ctx, const int* off, const int* order, int n_dest);

/* Forward model only: replaces multislice_propagate_batch_numpy (np_funcs.py:15-65).
 * angle_of_b / xoff / yoff: device int32 [B] or NULL.  out_wave: device [B][NX][NY] complex, detector
 * wave (far field: un-shifted fft2, the caller applies fftshift).  keep_tape != 0 keeps the per-slice
 * hybrid fields for bdof_tape_to_real (needs with_grad). */
int bdof_forward(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, void* out_wave, int keep_tape);

/* Slices z0 .. z0+nz-1 only, from caller-supplied real-space fields: in_real [B][NX][NY] complex is the (scattered part of
 * the) wave entering slice z0, one field per wavefield; out_real receives the wave after the range in real space —
 * psi_{z0+nz} if prop_last (a transfer-function step follows the last slice of the range), else phi_{z0+nz-1}.  Fused
 * streaming kernels only.  The building block of the tiled propagation below; np_funcs.py:36-43 restricted to a range. */
int bdof_forward_range(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                       const void* in_real, void* out_real, int prop_last);

/* Tiled ("pfft") Fresnel propagation (the reference's README.md:1-11; its scripts are on a branch absent from the checkout,
 * BASELINE cfg4): a field [FX][FY] complex too large for one fused plan is cut into B overlapping tiles [TX][TY] with origins
 * (x0[b], y0[b]) (device int32; periodic in the field, like the whole-field FFT propagator they replace); the tiles run
 * through bdof_forward_range as a batch with the object windowed by the same origins; every few slices the cores (tile minus
 * a halo of halo_x / halo_y pixels per side) are written back and the tiles re-cut, which refreshes the halos before the
 * wrap-around of a tile's own periodic boundary has crossed them (it advances lambda dz / (2 dx^2) pixels per slice).  The
 * outermost `taper` pixels of every gathered tile are ramped to zero (raised cosine): without the ramp the jump where a tile's
 * left and right edge meet diffracts into it with a 1/distance tail. */
int bdof_tiles_gather(bdof_ctx* ctx, const void* field, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0, const int* y0,
                      int taper);
int bdof_tiles_scatter(bdof_ctx* ctx, const void* tiles, void* field, int FX, int FY, int B, int TX, int TY, const int* x0, const int* y0,
                       int halo_x, int halo_y);

/* Gradient of the tiled path (tape-free, range by range, last range first).  bdof_adjoint_range is the adjoint of
 * bdof_forward_range(prop_last = 1): end_real = the psi_{z0+nz} that call returned, g_end_real = G(psi_{z0+nz}) (both device
 * [B][NX][NY] complex); g_start_real receives G(psi_{z0}); the gradient rows of the range's slices go to grot_range, device
 * [B][nz][NX][NY] pairs (a ctx configured with with_grad | 16 | 32 has no [B][S] gradient workspace of its own: 260 GB for
 * 121 tiles of 512^2 x 1024 slices).  The forward wave is marched back from end_real beside the adjoint field.
 * bdof_tiles_scatter_adjoint / bdof_tiles_gather_adjoint: adjoints of the two stitching steps; bdof_tiles_grad_add: the
 * range's gradient rows added into the volume gradient gvol (rows of volNY pairs, the object's own layout) through the table
 * of bdof_set_object — which must be injective in x for every slice (no rotation: the tiled path runs one pre-rotated object).
 * bdof_field_loss_seed: loss mean((|field| - meas)^2) (bdof_get_loss) and, in place, its seed — fullfield.py:106 on a field. */
int bdof_adjoint_range(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                       const void* end_real, const void* g_end_real, void* g_start_real, void* grot_range);
int bdof_tiles_scatter_adjoint(bdof_ctx* ctx, const void* field, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0,
                               const int* y0, int halo_x, int halo_y);
int bdof_tiles_gather_adjoint(bdof_ctx* ctx, const void* tiles, void* field, int FX, int FY, int B, int TX, int TY, const int* x0,
                              const int* y0, int taper);
int bdof_tiles_grad_add(bdof_ctx* ctx, const void* grot_range, void* gvol, int B, int TX, int TY, const int* x0, const int* y0, int z0, int nz);
int bdof_field_loss_seed(bdof_ctx* ctx, void* field, const float* meas, int FX, int FY);

/* Long-range correction and float64 path of the tiled propagator (round 4; DESIGN "cfg4").
 * The band-limited whole-field propagator of np_funcs.py:42 has alternating tails ~ lambda dz n / (2 pi x^2) after n slices that
 * reach across the whole field; a tile never sees sources beyond its halo, and at 1024 slices that alone is 2.2e-5 of the exit
 * wave — in float64, whatever the stitch interval.  Free space composes (P^n = F^-1 H^n F), so once per stitch range the part
 * the tiles miss is added back: D psi_in = (whole-field free-space step over the range) - (the tiles' own, stitched), exact
 * to first order in the object's phase over one range; with ranges shorter than the band edge's phase-winding length
 * (4 dx^2 / (lambda dz) = 16 slices at 5 keV / 1 nm) the tiling error drops to 1e-6.
 * bdof_fields_free_step: fields[b] <- F^-1 ( h * F fields[b] ) for B fields [NX][NY] in place (rocFFT, one batched transform
 *   pair); h[kx][ky] (device) carries 1 / (NX NY) and whatever power of the transfer function the caller folded in;
 *   conj_h: the adjoint step; is_double: complex128 fields and table, else complex64.  Any NX, NY rocFFT takes.
 * bdof_caxpy: y += alpha x on n complex numbers; bdof_c_convert: complex64 <-> complex128 copy.
 * bdof_tiles_gather_f64 / bdof_tiles_scatter_f64: bdof_tiles_gather / bdof_tiles_scatter on complex128 fields and tiles.
 * bdof_forward_range_f64: bdof_forward_range on caller-owned complex128 fields [B][NX][NY] IN PLACE, everything in float64 —
 *   what the reference computes in (np_funcs.py:20-42 promote to complex128, quirk Q2): c = exp(i k delta) exp(-k beta) from
 *   the (delta, beta) rows of bdof_set_object, rocFFT double-precision transforms, h[kx][ky] / (NX NY) in float64 (device).
 *   Unfused (an accuracy path, ~4x the time of the fused float32 kernels): a 1024-slice stack through float32 transforms
 *   carries 1.5e-5 of rounding, this path none. */
/* The slice step's transfer function in float64 ([ky][kx], 1 / (NX NY) folded in: what bdof_set_physics' hs was rounded from;
 * call it after bdof_set_physics).  The streaming engine's per-slice launches then multiply by dithered float32 copies of it —
 * copy z mod D for slice z, the adjoint step by the conjugate of the copy its forward step used — whose roundings average to the
 * float64 values: a fixed float32 table is the same perturbation in every slice and its error grows linearly with depth
 * (1.4e-5 of the exit wave after 1024 slices with everything else in float64; 1e-6 with the copies).  np_funcs.py:42 multiplies
 * by a complex128 H.  D = env BDOF_H_DITHER (default 64, at most 256 MiB of copies; 0 switches it off). */
int bdof_set_transfer_f64(bdof_ctx* ctx, const double* hs64);
/* bdof_forward_range with the table of its transfer-function steps supplied by the caller (device, [ky][kx] complex64 like hs):
 * the tiled propagator's free-space step over a whole stitch range (H^n) through the fused kernels. */
/* Carrier fields for the next bdof_forward_range calls over slices z0 .. z0 + nz - 1 of B wavefields: device stack [nz][B][NX][NY]
 * complex64, wavefield b's free-space propagation to the entrance of each slice (formed in double by the caller:
 * bdof_fields_free_step + bdof_c_convert).  The range is swept on psi_z = p_z + eps_z with only eps in the float32 transforms;
 * in_real is the scattered part entering the range, out_real the scattered part leaving it — for the tiles of a corrected stitch
 * range (np_funcs.py:36-43 on a tile) T psi - T_free psi itself.  Needs a zero ctx probe (bdof_set_probe with a0 = 0, no probe
 * stack).  NULL removes the stack. */
int bdof_set_range_carrier(bdof_ctx* ctx, const void* stack, int B, int z0, int nz);
/* bdof_fields_free_step on an auxiliary stream: starts after everything queued on the ctx's stream so far (first copying `src`
 * into `fields` there, if not NULL) and runs beside what the ctx's stream is handed next; bdof_aux_join makes the ctx's stream
 * wait for it.  One step in flight at a time.  (The whole-field free-space step of a stitch range, F^-1 H^n F of np_funcs.py:42,
 * beside the tiles' sweeps.) */
int bdof_fields_free_step_aux(bdof_ctx* ctx, void* fields, const void* src, int B, int NX, int NY, const void* h, int conj_h, int is_double);
int bdof_aux_join(bdof_ctx* ctx);
/* The carrier stack of bdof_set_range_carrier in one call: p0 device complex128 [B][NX][NY] (the wavefields entering the range;
 * overwritten), stack device complex64 [nz][B][NX][NY] <- F^-1(H^z F p0), z = 0 .. nz - 1, formed in double (one forward
 * transform, the spectra by a running product, one batched inverse transform); h complex128 [kx][ky], ifftshift(H) / (NX NY). */
int bdof_range_carrier_build(bdof_ctx* ctx, void* p0, void* stack, int B, int NX, int NY, const void* h, int nz);
int bdof_forward_range_h(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                         const void* in_real, void* out_real, int prop_last, const void* h);
int bdof_fields_free_step(bdof_ctx* ctx, void* fields, int B, int NX, int NY, const void* h, int conj_h, int is_double);
int bdof_caxpy(bdof_ctx* ctx, void* y, const void* x, double alpha, size_t n, int is_double);
int bdof_c_convert(bdof_ctx* ctx, void* dst, const void* src, size_t n, int to_double);
int bdof_tiles_gather_f64(bdof_ctx* ctx, const void* field, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0,
                          const int* y0, int taper);
int bdof_tiles_scatter_f64(bdof_ctx* ctx, const void* tiles, void* field, int FX, int FY, int B, int TX, int TY, const int* x0,
                           const int* y0, int halo_x, int halo_y);
/* The float32 tiled path with the long-range correction keeps its FIELD in complex128: the whole-field free-space step carries
 * the bulk of the wave in double, the complex64 tiles (fused kernels) add the object's part, T psi - T_free psi.
 * bdof_tiles_gather_mixed: complex64 tiles cut out of the complex128 field (tapered, periodic);
 * bdof_tiles_scatter_diff64: field[core] = (accumulate ? field[core] : 0) + tiles_a - tiles_b (tiles_b nullable), in float64. */
int bdof_tiles_gather_mixed(bdof_ctx* ctx, const void* field64, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0,
                            const int* y0, int taper);
int bdof_tiles_scatter_diff64(bdof_ctx* ctx, const void* tiles_a, const void* tiles_b, void* field64, int FX, int FY, int B, int TX, int TY,
                              const int* x0, const int* y0, int halo_x, int halo_y, int accumulate);
/* Their adjoints, for the gradient through the corrected tiled model: bdof_tiles_scatter_adjoint_mixed (complex64 tiles = the
 * complex128 field on every tile's core, zero elsewhere) and bdof_tiles_gather_adjoint_diff64 (field64 (+)= the tapered, periodic
 * scatter-add of tiles_a - tiles_b; deterministic gather form). */
int bdof_tiles_scatter_adjoint_mixed(bdof_ctx* ctx, const void* field64, int FX, int FY, void* tiles, int B, int TX, int TY, const int* x0,
                                     const int* y0, int halo_x, int halo_y);
int bdof_tiles_gather_adjoint_diff64(bdof_ctx* ctx, const void* tiles_a, const void* tiles_b, void* field64, int FX, int FY, int B, int TX,
                                     int TY, const int* x0, const int* y0, int taper, int accumulate);
int bdof_forward_range_f64(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, int z0, int nz,
                           void* fields, const void* h, double k, int prop_last);

/* probe_array[i] of np_funcs.py:43 (wave after slice i), device out [B][NX][NY]; valid after a
 * bdof_forward(keep_tape=1) (bdof_loss_grad reuses the tape for its own purposes and invalidates it). */
int bdof_tape_to_real(bdof_ctx* ctx, int i, int B, void* out);

/* Loss + gradient: replaces loss_grad = autograd.grad(calculate_loss,[0,1]) for the multislice part
 * (cnn_propagator/fullfield.py:93-106,329,345; cnn_propagator/ptychography.py:30-81,248,301).
 * meas: device float [B][NX][NY] = |measured| in this library's index order (far field: un-shifted).
 * loss = mean((|d|-meas)^2) is left on the device (bdof_get_loss); the gradient w.r.t. the rotated /
 * windowed (delta,beta) is left in the ctx (bdof_grot). out_wave may be NULL. */
int bdof_loss_grad(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, void* out_wave);
int bdof_get_loss(bdof_ctx* ctx, double* loss);
/* Gradient w.r.t. the probe: with bdof_enable_probe_grad(1) every bdof_loss_grad also keeps G(psi_0) = dL/dRe psi_0 + i dL/dIm psi_0
 * of each wavefield; bdof_probe_grad sums it over the batch into out (device [NX][NY] complex, accumulate != 0: added to it).
 * The probe is shared by the wavefields of a minibatch, so this is the gradient of the loss w.r.t. (probe_real, probe_imag) —
 * the trainable probe of tensorflow_recon/fullfield.py:311-327,442-455 (probe_type='optimizable'). */
int bdof_enable_probe_grad(bdof_ctx* ctx, int enable);
int bdof_probe_grad(bdof_ctx* ctx, void* out, int accumulate);
/* mode 1: the `meas` arrays of bdof_loss_grad hold |measured| - |a0| instead of |measured| (a0: the plane-wave part handed
 * to bdof_set_probe; every transfer-function step has unit modulus at DC, so |a0| is the carrier's modulus at the detector
 * too).  The residual is then formed as (|a + e| - |a|) - (m - |a|) with |a + e| - |a| evaluated without cancellation —
 * three times more accurate gradients for plane-wave illumination.  Real-space detectors (none / near) and a scalar carrier
 * only; mode 0 (default): plain amplitudes.  bdof_loss_grad_conv honours it too: there the constant part of the detector
 * wave is A = s a_S with s the corner-pixel renormalisation of propagation.py:79,109-110, formed in float64 per call. */
int bdof_set_meas_mode(bdof_ctx* ctx, int mode);

/* Real-space truncated-kernel propagator: replaces multislice_propagate_cnn (cnn_propagator/propagation.py:18-133), the
 * forward model cnn_propagator/fullfield.py:87,102 and ptychography.py:74 literally call.  The cropped kernel is separable,
 * K[p][q] = e * ky[p] * kx[q] (ky, kx: HOST complex arrays of ks taps, ks odd <= 33); ksum = sum of K (the factor of the
 * padding constant per slice, propagation.py:104); k = 2*np.pi*delta_nm/lambda_nm (propagation.py:25).  The detector step
 * and the probe are those of bdof_set_physics / bdof_set_probe.  bdof_forward_conv / bdof_loss_grad_conv mirror
 * bdof_forward / bdof_loss_grad (same arguments, gradient left in bdof_grot). */
int bdof_set_conv(bdof_ctx* ctx, const float* ky, const float* kx, int ks, double e_re, double e_im, double ksum_re,
                  double ksum_im, double k);
/* Optional, after bdof_set_conv: the same taps in float64 (HOST complex128 arrays of ks taps, and e).  The library then keeps
 * BDOF_TW_DITHER (default 64) copies of the float32 taps whose roundings average to these values and runs slice z with copy
 * z mod D, as it does with the transform constants of the transfer-function path (bdof_configure) — the taps are the same small
 * perturbation of every slice otherwise.  A no-op with BDOF_TW_DITHER=0. */
int bdof_set_conv_taps_f64(bdof_ctx* ctx, const double* ky, const double* kx, double e_re, double e_im);
/* Carrier FIELD of the real-space propagator, for a probe with no dominant constant part (a localised ptychography probe —
 * what cnn_propagator/ptychography.py:74-76 runs): stack = HOST [S + 1][NX][NY] complex64, the probe carried through EMPTY
 * space by the same padded convolution, p_0 = probe, p_{z+1} = K * pad(p_z, edge_z), edge_{z+1} = sum(K) edge_z, edge_0 = 1
 * (propagation.py:79-104 without an object), computed by the host in float64; det64 = HOST [NX][NY] complex128, the detector
 * plane of p_S before the renormalisation (no detector: p_S; near field: one transfer-function step of it; far field: its
 * un-shifted, un-normalised fft2 indexed [ky][kx]); p0 / pS: the corner pixels p_0[0,0], p_S[0,0] in float64 (the
 * renormalisation s = probe[0,0] / psi_S[0,0,0], propagation.py:109-110, is formed against them).  The wave is then carried as
 * psi_z = p_z + eps_z, eps zero-padded, and |d| - m is taken in float64 against s * det64.  Call bdof_set_probe with a zero
 * array and a0 = 0 first, and again after every bdof_set_conv.  NULL, NULL removes the stack. */
int bdof_set_conv_probe_stack(bdof_ctx* ctx, const float* stack, const double* det64, double p0_re, double p0_im, double pS_re, double pS_im);
/* The same loss + gradient entirely in float64 (bdof_conv64.h): the reference differentiates this forward model in float64
 * (autograd on numpy float64, cnn_propagator/ptychography.py:248,301), and for far-field ptychography the float32 kernels stop at
 * 1.5e-5 of its reconstructed delta (golden vector G14) — all of it in the wake of the corner pixel through which the
 * renormalisation of propagation.py:109-110 feeds sum(G conj q) back into the stack.  An accuracy path for the first minibatch of
 * an epoch (adjoint_precision='first-step'), unfused: every slice's pad + 'valid' convolution is one rocFFT double-precision
 * transform pair on the padded (N + ks - 1)^2 grid (overlap-save; its adjoint the circular correlation of the embedded adjoint
 * field), everything else point-wise in double.  bdof_set_conv_f64: probe host complex128 [NX][NY]; khat host complex128
 * [M][M], M = N + ks - 1: fft2 of the kernel zero-padded to M x M, transposed to [kx][ky], / M^2; ksum = sum of its taps; k as
 * for bdof_set_conv.  bdof_loss_grad_conv_f64: meas as for bdof_loss_grad_conv (+ meas_ref, what the host subtracted under
 * bdof_set_meas_mode(1)); loss by bdof_get_loss; gradient rows in bdof_grot.  Square fields; detector none or far field — a
 * near-field detector (propagation.py:122-127: one transfer-function step of the renormalised exit wave) after
 * bdof_set_conv_f64_detector(ctx, hdetT): host complex128 [kx][ky], ifftshift(H_det) / (NX NY). */
int bdof_set_conv_f64(bdof_ctx* ctx, const double* probe, const double* khat, int ks, double ksum_re, double ksum_im, double k);
int bdof_loss_grad_conv_f64(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas,
                            double meas_ref);
int bdof_set_conv_f64_detector(bdof_ctx* ctx, const double* hdetT);
/* The transfer-function model (cnn_propagator/np_funcs.py:15-65) in float64 on the same context: what autograd differentiates in
 * the reference (cnn_propagator/ptychography.py:248,301; fullfield.py:329,345) — modulation from the (delta, beta) rows, the step
 * after a slice as one rocFFT double-precision transform pair with H in float64, detector step (none / near field / far field),
 * magnitude loss, adjoint sweep.  Unfused; the accuracy path for the first minibatch of an epoch (adjoint_precision='first-step')
 * and a float64 twin of the fused kernels on the device.  bdof_set_tf_f64: probe host complex128 [NX][NY]; hT / hdetT (NULL
 * without a near-field detector): host complex128 [kx][ky], ifftshift(H) / (NX NY); k as for bdof_set_physics.
 * bdof_loss_grad_tf_f64: arguments as bdof_loss_grad_conv_f64; loss by bdof_get_loss, gradient rows in bdof_grot.
 * bdof_set_physics, bdof_set_probe and bdof_set_conv drop what either float64 path was handed (it described the previous model):
 * hand it over again after them, or the float64 calls fail with BDOF_ERR_STATE. */
int bdof_set_tf_f64(bdof_ctx* ctx, const double* probe, const double* hT, const double* hdetT, double k);
int bdof_loss_grad_tf_f64(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, double meas_ref);
int bdof_forward_conv(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, void* out_wave);
int bdof_loss_grad_conv(bdof_ctx* ctx, int B, const int* angle_of_b, const int* xoff, const int* yoff, const float* meas, void* out_wave);
void* bdof_grot(bdof_ctx* ctx);      /* device [B][S][NX][NY] pairs */

/* Adjoint of the rotation gather: gvol[dest][y] (+)= scale * sum over the batch of the rows gathered
 * from dest.  gvol: device [n_dest][NY] pairs. */
int bdof_rotation_adjoint(bdof_ctx* ctx, int B, const int* angle_of_b, void* gvol, int accumulate, float scale);
/* The same for destination rows [row0, row0 + n_rows) only (gvol is still the base of the whole gradient volume): lets
 * the host pipeline rotation adjoint -> all-reduce -> Adam slab by slab, hiding them behind the collective
 * (comm.Allreduce(this_grads, grads), cnn_propagator/fullfield.py:348-351). */
int bdof_rotation_adjoint_rows(bdof_ctx* ctx, int B, const int* angle_of_b, void* gvol, int row0, int n_rows, int accumulate,
                               float scale);

/* Bilinear rotation — the TF twin's tf_rotate(obj, theta, interpolation='BILINEAR') (tensorflow_recon/fullfield.py:96; the cnn
 * variant's nearest-neighbour tables are the fused path of bdof_set_object).  vol: device [NXv][NZv][NYv] pairs; prm: DEVICE
 * float64 [B][4] = (cos, sin, x_off, y_off) of tf.contrib.image.angles_to_projective_transforms for images of height NXv and
 * width NZv: output pixel (h, w) = (x, z) samples the input at (sin w + cos h + y_off, cos w - sin h + x_off), zeros outside.
 * out_rows: device [B][NZv][NXv][NYv] pairs — hand it to bdof_set_object(out_rows, B*NZv*NXv, NYv, NULL, 0, 0) as a batch of
 * already rotated objects.  The adjoint takes the rotated-frame gradient bdof_grot() back to volume rows [row0, row0+n_rows)
 * of gvol (deterministic gather; same slab interface as bdof_rotation_adjoint_rows). */
int bdof_rotate_bilinear(bdof_ctx* ctx, const void* vol, int NXv, int NZv, int NYv, const double* prm, int B, void* out_rows);
/* The two calls above in one pass (what FullfieldSolver(rotation='bilinear') runs every step): the B rotated objects are
 * written straight into the ctx's table of modulation factors exp(i k (delta + i beta) dz) - 1 and bound as the batch's objects
 * — no rotated (delta, beta) copy, no second pass over B volumes.  The volume must be (NX, S, NY) of the configured wavefields.
 * conv != 0: factors for the real-space propagator's k (bdof_set_conv), which the conv entry points then use. */
int bdof_set_object_bilinear(bdof_ctx* ctx, const void* vol, int NXv, int NZv, int NYv, const double* prm, int B, int conv);
int bdof_rotate_bilinear_adjoint(bdof_ctx* ctx, const void* grot, int NXv, int NZv, int NYv, const double* prm, int B, void* gvol, int row0,
                                 int n_rows, int accumulate, float scale);

/* Ptychography: adjoint of rotate + zero-pad + per-position window (cnn_propagator/ptychography.py:32-34,42-73) for a
 * batch whose elements all use rotation angle `angle` and windows at (xoff[b], yoff[b]); gvol: device [n_dest][volNY] pairs. */
int bdof_window_rotation_adjoint(bdof_ctx* ctx, int B, int angle, const int* xoff, const int* yoff, void* gvol,
                                 int accumulate, float scale);

/* Fused regulariser gradient + Adam + mask + clip on a volume [NXv][NZv][NYv] of pairs.
 * Replaces cnn_propagator/fullfield.py:109-118 (gradient of the L1 + TV terms), apply_gradient_adam
 * (cnn_propagator/util.py:280-291) and the constraints of fullfield.py:359-362.  g is the data-term
 * gradient summed over ranks; g_scale = 1/size (fullfield.py:351). */
int bdof_adam_step(bdof_ctx* ctx, const void* x_old, void* x_new, const void* g, void* m, void* v, const float* mask,
                   int NXv, int NZv, int NYv, float g_scale, float alpha_d, float alpha_b, float gamma,
                   float lr, float b1, float b2, float eps, int i_batch, int clip);
/* The same for the slab x in [x0, x0 + nx) of the [X][Z][Y] volume (all pointers are still those of the whole volume). */
int bdof_adam_step_slab(bdof_ctx* ctx, const void* x_old, void* x_new, const void* g, void* m, void* v, const float* mask,
                        int NXv, int NZv, int NYv, float g_scale, float alpha_d, float alpha_b, float gamma,
                        float lr, float b1, float b2, float eps, int i_batch, int clip, int x0, int nx);

/* dst[b] = src[idx[b]] for B fields of bytes_per_field bytes (a multiple of 4; 16-byte copies when it is a multiple of 16;
 * idx: device int32 [B]) in one launch:
 * this_prj_batch = prj[this_ind_batch] (cnn_propagator/fullfield.py:344) on the device-resident stack of amplitudes. */
int bdof_gather_fields(bdof_ctx* ctx, void* dst, const void* src, const int* idx, int B, size_t bytes_per_field);

/* Value of the regulariser terms of the loss (cnn_propagator/fullfield.py:109-118; total_variation_3d, util.py:61-70) on a
 * volume [NXv][NZv][NYv] of pairs: sums[0] = sum|delta|, sums[1] = sum|beta|, sums[2] = TV(delta) (periodic, anisotropic).
 * The caller weights them (alpha_d, alpha_b, gamma).  Synchronous (three numbers come back). */
int bdof_regularizer_value(bdof_ctx* ctx, const void* x, int NXv, int NZv, int NYv, double* sums);

/* Shrink-wrap (cnn_propagator/fullfield.py:365-368): mask[i] *= (delta[i] > thresh) over n voxels. */
int bdof_mask_shrink(bdof_ctx* ctx, const void* x, float* mask, size_t n, float thresh);

/* Sub-batch streams of the fused FFT engine.  A batch whose tiles (B*NX/16 workgroups per launch) cover at least the
 * chip's resident workgroup slots is split in groups that run the same kernel sequence concurrently on side streams
 * (no dependencies between wavefields of one minibatch: fullfield.py:95-103 is a plain loop over the batch), so the
 * partly filled last round of one launch is covered by the other group's kernels.  n = -1 automatic (1 or 2 groups),
 * 1..4 fixed.  bdof_batch_groups returns the number of groups a batch of B wavefields runs as. */
int bdof_set_streams(bdof_ctx* ctx, int n);
int bdof_batch_groups(bdof_ctx* ctx, int B);

/* Per-kernel-class timing with HIP events on the stream of each launch (bench.py roofline leg).  enable = 0 off, 1 every launch,
 * n > 1 every n-th launch of the per-slice kernel classes (two event records per launch cost ~9 us of stream time). */
int bdof_profile_enable(bdof_ctx* ctx, int enable);
int bdof_profile_read(bdof_ctx* ctx, int kernel_class, int* n_launches, double* total_ms);

/* ---- Collectives: the gradient exchange of the data-parallel loop ------------------------------------------------
 * Replaces comm.Allreduce(this_grads, grads) (mpi4py, host float64 buffers: cnn_propagator/fullfield.py:348-351,
 * ptychography.py:302-306) with RCCL over xGMI on device float32 buffers, one process per GPU.  librccl is opened at
 * run time by the first bdof_comm_* call; single-GPU use never loads it.
 *   bdof_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId); the host hands it to every rank over any
 *                         channel it likes (the Python host: a unix-domain socket rendezvous, comm.py);
 *   bdof_comm_create      ncclCommInitRank on `device` (collective: every rank calls it);
 *   bdof_allreduce_grad   in-place SUM of `count` floats;
 *   bdof_reduce_scatter_grad / bdof_allgather_volume   in place on a buffer of nranks * count_per_rank floats: after the
 *                         reduce-scatter rank r holds the sum in its part [r*count, (r+1)*count); the all-gather
 *                         distributes every rank's part — the "reduce-scatter -> Adam on 1/N -> all-gather" form of
 *                         the step, same wire bytes as the all-reduce and 1/N of the Adam traffic;
 *   bdof_bcast_volume     in-place broadcast from `root` (initial guess of rank 0, ptychography.py:169-208).
 * Every collective runs on the communicator's own stream, ordered BEHIND the work the ctx stream holds at the call, and
 * returns a ticket; bdof_comm_wait(comm, ctx, ticket) makes the ctx stream wait for it (stream-side, no host block), so
 * kernels enqueued between the two calls overlap the transfer.  Tickets are a ring of 256.  Errors: 1000 + ncclResult_t,
 * text via bdof_comm_last_error (NULL: the error of a failed create / unique_id). */
typedef struct bdof_comm bdof_comm;
int bdof_comm_unique_id(void* id, size_t bytes);
int bdof_comm_create(bdof_comm** out, int device, int nranks, int rank, const void* id, size_t bytes);
void bdof_comm_destroy(bdof_comm* comm);
const char* bdof_comm_last_error(const bdof_comm* comm);
int bdof_comm_size(const bdof_comm* comm);
int bdof_comm_rank(const bdof_comm* comm);
int bdof_allreduce_grad(bdof_comm* comm, bdof_ctx* ctx, void* buf, size_t count, int* ticket);
int bdof_reduce_scatter_grad(bdof_comm* comm, bdof_ctx* ctx, void* buf, size_t count_per_rank, int* ticket);
int bdof_allgather_volume(bdof_comm* comm, bdof_ctx* ctx, void* buf, size_t count_per_rank, int* ticket);
int bdof_bcast_volume(bdof_comm* comm, bdof_ctx* ctx, void* buf, size_t count, int root, int* ticket);
int bdof_comm_wait(bdof_comm* comm, bdof_ctx* ctx, int ticket);
int bdof_comm_sync(bdof_comm* comm);

/* Device memory helpers for hosts without an allocator of their own.  bdof_malloc allocates on the calling thread's
 * current device, bdof_ctx_malloc on the ctx's device. */
int bdof_ctx_malloc(bdof_ctx* ctx, void** ptr, size_t bytes);
int bdof_device_mem(bdof_ctx* ctx, size_t* free_bytes, size_t* total_bytes);   /* hipMemGetInfo of the ctx's device */
int bdof_malloc(void** ptr, size_t bytes);
int bdof_free(void* ptr);
int bdof_memcpy_h2d(bdof_ctx* ctx, void* dst, const void* src, size_t bytes);
int bdof_memcpy_d2h(bdof_ctx* ctx, void* dst, const void* src, size_t bytes);
int bdof_memset(bdof_ctx* ctx, void* dst, int value, size_t bytes);
int bdof_memcpy_d2d(bdof_ctx* ctx, void* dst, const void* src, size_t bytes);   /* asynchronous on the ctx stream */

#ifdef __cplusplus
}
#endif
#endif
