/* cabi_forward.c — libbdof.so driven from plain C (no HIP headers, no Python): the forward model + loss + gradient of a
 * batch of already rotated objects, i.e. what cnn_propagator/np_funcs.py:15-65 and the autograd.grad call of
 * cnn_propagator/fullfield.py:329,345 compute.  Inputs come from a small binary file written by the caller
 * (tests/test_gpu_cabi_c.py); the detector waves, the loss and the gradient go back to a second file.
 *
 *   gcc -O2 -I include examples/cabi_forward.c -o cabi_forward -L beyond_dof_amd -lbdof -Wl,-rpath,$PWD/beyond_dof_amd
 *
 * File layout (little endian): int32 NY, NX, S, B, det_mode, variant; double k, h00[2], hdet00[2], a0[2];
 * float hs[NX*NY*2]; double hs64[NX*NY*2] (the same table before it was rounded: bdof_set_transfer_f64); float hdet[NX*NY*2]
 * (det_mode 1 only); float probe_eps[NX*NY*2]; float rows[B*S*NX*NY*2]; float meas[B*NX*NY]. */
#include <stdio.h>
#include <stdlib.h>
#include "bdof.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != 0) {                                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? bdof_last_error(ctx) : "(no ctx)");  \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)

static void* slurp(FILE* f, size_t bytes) {
    void* p = malloc(bytes);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read (%zu bytes)\n", bytes); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    bdof_ctx* ctx = NULL;
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int hdr[6];
    double dbl[7];
    if (fread(hdr, sizeof(int), 6, f) != 6 || fread(dbl, sizeof(double), 7, f) != 7) { fprintf(stderr, "bad header\n"); return 3; }
    const int NY = hdr[0], NX = hdr[1], S = hdr[2], B = hdr[3], det = hdr[4], variant = hdr[5];
    const size_t fld = (size_t)NX * NY;
    float* hs = slurp(f, fld * 2 * sizeof(float));
    double* hs64 = slurp(f, fld * 2 * sizeof(double));
    float* hdet = det == BDOF_DET_NEAR ? slurp(f, fld * 2 * sizeof(float)) : NULL;
    float* probe = slurp(f, fld * 2 * sizeof(float));
    float* rows = slurp(f, (size_t)B * S * fld * 2 * sizeof(float));
    float* meas = slurp(f, (size_t)B * fld * sizeof(float));
    fclose(f);

    if (bdof_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 4; }
    CHECK(bdof_ctx_create(&ctx, 0, NULL));
    CHECK(bdof_configure(ctx, NY, NX, S, B, 1));
    CHECK(bdof_set_physics(ctx, dbl[0], hs, hdet, &dbl[1], &dbl[3], det, variant));
    CHECK(bdof_set_transfer_f64(ctx, hs64));         /* the slice step's H in float64: the kernels use dithered float32 copies of it */
    CHECK(bdof_set_probe(ctx, probe, dbl[5], dbl[6]));

    void *d_rows = NULL, *d_meas = NULL, *d_wave = NULL;
    const size_t rows_bytes = (size_t)B * S * fld * 2 * sizeof(float);
    CHECK(bdof_malloc(&d_rows, rows_bytes));
    CHECK(bdof_malloc(&d_meas, (size_t)B * fld * sizeof(float)));
    CHECK(bdof_malloc(&d_wave, (size_t)B * fld * 2 * sizeof(float)));
    CHECK(bdof_memcpy_h2d(ctx, d_rows, rows, rows_bytes));
    CHECK(bdof_memcpy_h2d(ctx, d_meas, meas, (size_t)B * fld * sizeof(float)));
    CHECK(bdof_set_object(ctx, d_rows, (long long)B * S * NX, NY, NULL, 0, 0));

    double loss = 0.0;
    CHECK(bdof_loss_grad(ctx, B, NULL, NULL, NULL, (const float*)d_meas, d_wave));
    CHECK(bdof_get_loss(ctx, &loss));
    float* wave = malloc((size_t)B * fld * 2 * sizeof(float));
    float* grad = malloc(rows_bytes);
    CHECK(bdof_memcpy_d2h(ctx, wave, d_wave, (size_t)B * fld * 2 * sizeof(float)));
    CHECK(bdof_memcpy_d2h(ctx, grad, bdof_grot(ctx), rows_bytes));
    CHECK(bdof_sync(ctx));

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fwrite(&loss, sizeof(double), 1, o);
    fwrite(wave, sizeof(float), (size_t)B * fld * 2, o);
    fwrite(grad, 1, rows_bytes, o);
    fclose(o);
    printf("loss %.9e\n", loss);
    bdof_free(d_rows); bdof_free(d_meas); bdof_free(d_wave);
    bdof_ctx_destroy(ctx);
    return 0;
}
