/* gsr_client.c -- a plain C program (gcc, no torch, no Python, no C++) that drives libgsr_hip.so (or, built with
 * -DGSR_CLIENT_CPU, the oracle's libgsr_cpu.so) through include/gsr.h:
 * the drop-in boundary is a C ABI over caller-owned device memory and this is the proof.  It reads one scene + camera +
 * dL/dpixels from a flat binary file, runs gsr_forward_count / gsr_forward_render / gsr_backward with buffers it allocates
 * with hipMalloc, and writes every output to a flat binary file; tests/test_gpu_c_abi.py compares that file with the CPU
 * oracle under the tolerances of tests/parity.py.
 *
 * input : int64 N, int32 W, H, degree, 3 x int32 pad; GsrCamera (raw struct); float means[3N], scales[3N], rots[4N],
 *         opacity[N], sh[48N], dpix[3WH]
 * output: int64 D; int32 radii[N], offsets[N]; float xy[2N], depths[N], cov3D[6N], rgb[3N], conic[4N], clamped[3N];
 *         int32 point_list[D], ranges[2T]; float image[3WH], inv_depth[WH], final_T[WH]; int32 n_contrib[WH];
 *         float dmean3D[3N], dscale[3N], drot[4N], dopacity[N], dshs[48N], dcolor[3N], dmean2D[3N], dconic[4N] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsr.h"

#define CHECK_GSR(x) do { int r_ = (x); if (r_ != GSR_OK) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, gsr_strerror(r_)); return 3; } } while (0)

/* The same program is built twice (tests/c_abi/Makefile): `gsr_client` against libgsr_hip.so with the buffers in device
 * memory, and -DGSR_CLIENT_CPU `gsr_client_cpu` against oracle/libgsr_cpu.so -- the CPU oracle behind the identical ABI --
 * with the buffers in host memory.  Everything below the three helpers is shared. */
#ifdef GSR_CLIENT_CPU
#define CHECK_HIP(x) do { if ((x) != 0) return 2; } while (0)
typedef void *hipStream_t;
static int hipStreamCreate(hipStream_t *s) { *s = NULL; return 0; }
static int hipStreamSynchronize(hipStream_t s) { (void)s; return 0; }
static void *dev_alloc(size_t bytes) { return aligned_alloc(64, ((bytes ? bytes : 4) + 63) & ~(size_t)63); }
static void *dev_upload(FILE *f, size_t bytes)
{
    void *d = dev_alloc(bytes);
    if (!d || (bytes && fread(d, 1, bytes, f) != bytes)) { fprintf(stderr, "short read\n"); exit(4); }
    return d;
}
static void dump(FILE *f, const void *d, size_t bytes) { fwrite(d, 1, bytes, f); }
#else
#include <hip/hip_runtime_api.h>
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d HIP error %d\n", __FILE__, __LINE__, (int)e_); return 2; } } while (0)
static void *dev_alloc(size_t bytes) { void *p = NULL; return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess ? p : NULL; }
static void *dev_upload(FILE *f, size_t bytes)
{
    void *h = malloc(bytes ? bytes : 4), *d = dev_alloc(bytes);
    if (!h || !d || (bytes && fread(h, 1, bytes, f) != bytes)) { fprintf(stderr, "short read\n"); exit(4); }
    if (hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) exit(5);
    free(h);
    return d;
}
static void dump(FILE *f, const void *d, size_t bytes)
{
    void *h = malloc(bytes ? bytes : 4);
    if (bytes && hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) exit(6);
    fwrite(h, 1, bytes, f);
    free(h);
}
/* columns [c0, c0 + nc) of n rows of 16 floats in device memory, written packed */
static void dump_cols(FILE *f, const void *rows, size_t n, int c0, int nc)
{
    float *h = malloc(64 * (n ? n : 1));
    if (n && hipMemcpy(h, rows, 64 * n, hipMemcpyDeviceToHost) != hipSuccess) exit(6);
    for (size_t i = 0; i < n; ++i) fwrite(h + 16 * i + c0, 4, (size_t)nc, f);
    free(h);
}
#endif

int main(int argc, char **argv)
{
    /* third argument "records" (HIP build only, ABI 7): xy / conic_opacity / rgb are not given to the forward at all -- it keeps its
     * blend records in a buffer of ours and we read the three arrays back as its columns; likewise dL_dcolor / dL_dmean2D / dL_dconic
     * are not given to the backward and are read back as columns of the accumulator records in the backward workspace.  The dump has
     * the same layout either way, so the two modes can be diffed. */
    const int records_mode = argc == 4 && strcmp(argv[3], "records") == 0;
    if (argc != 3 && !records_mode) { fprintf(stderr, "usage: %s in.bin out.bin [records]\n", argv[0]); return 1; }
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 1;
    int64_t N; int32_t hdr[6]; GsrCamera cam;
    if (fread(&N, 8, 1, in) != 1 || fread(hdr, 4, 6, in) != 6 || fread(&cam, sizeof cam, 1, in) != 1) return 4;
    const int32_t W = hdr[0], H = hdr[1], degree = hdr[2];
    const size_t P = (size_t)W * H, n = (size_t)N;
    const int tiles = ((W + 15) / 16) * ((H + 15) / 16);
    if (gsr_abi_version() != GSR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    GsrScene sc = {N, 0, 0, 0, 0, 0, degree, 1.0f, 1};
    sc.means = dev_upload(in, 12 * n); sc.scales = dev_upload(in, 12 * n); sc.rotations = dev_upload(in, 16 * n);
    sc.opacity = dev_upload(in, 4 * n); sc.sh = dev_upload(in, 192 * n);
    float *dpix = dev_upload(in, 12 * P);
    fclose(in);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    GsrGeom g = {0};
    g.radii = dev_alloc(4 * n); g.tiles_touched = dev_alloc(4 * n); g.point_offsets = dev_alloc(4 * n); g.xy = dev_alloc(8 * n);
    g.depths = dev_alloc(4 * n); g.cov3D = dev_alloc(24 * n); g.rgb = dev_alloc(12 * n); g.conic_opacity = dev_alloc(16 * n);
    g.clamped_state = dev_alloc(12 * n);
#ifndef GSR_CLIENT_CPU
    void *records = NULL;
    if (records_mode) { records = dev_alloc(64 * n); g.blend_records = records; g.xy = g.rgb = g.conic_opacity = NULL; }
    g.sh_dir_grad = dev_alloc(36 * n); /* forward -> backward: the SH backward then skips the 192-byte coefficient rows */
#endif
    const size_t geom_bytes = gsr_geom_workspace_bytes(N);
    void *geom_ws = dev_alloc(geom_bytes);
    int64_t D = -1;
    CHECK_GSR(gsr_forward_count(&sc, &cam, &g, geom_ws, geom_bytes, &D, stream));

    const size_t bwd_bytes = gsr_backward_workspace_bytes(N, D, W, H);
    void *bwd_ws = dev_alloc(bwd_bytes);
    GsrBinning bin = {D, dev_alloc(4 * (size_t)D), dev_alloc(8 * (size_t)tiles), dev_alloc((size_t)D + 16), /* block_masks: forward -> backward */
                      dev_alloc(4 * gsr_block_order_ints(W, H)),                                           /* block_order: likewise */
                      NULL, 0};
#ifndef GSR_CLIENT_CPU
    bin.backward_ws = bwd_ws; /* the forward blend's spare workgroups clear the backward's accumulators in it */
#endif
    GsrImage img = {dev_alloc(12 * P), dev_alloc(4 * P), dev_alloc(4 * P), dev_alloc(4 * P)};
    const size_t bin_bytes = gsr_binning_workspace_bytes(N, D, W, H);
    void *bin_ws = dev_alloc(bin_bytes);
    CHECK_GSR(gsr_forward_render(&sc, &cam, &g, &bin, &img, geom_ws, geom_bytes, bin_ws, bin_bytes, stream));

    GsrGrads gr = {dev_alloc(12 * n), dev_alloc(12 * n), dev_alloc(16 * n), dev_alloc(4 * n), dev_alloc(192 * n), dev_alloc(12 * n),
                   dev_alloc(12 * n), dev_alloc(16 * n), NULL};
#ifndef GSR_CLIENT_CPU
    if (records_mode) gr.dL_dcolor = gr.dL_dmean2D = gr.dL_dconic = NULL; /* read back below as columns of the accumulator records */
    else g.blend_records = geom_ws; /* still untouched: the backward reuses the forward's records */
    bin.backward_ws_cleared = 1; /* ... and bwd_ws has not been touched since gsr_forward_render cleared its accumulators */
#endif
    CHECK_GSR(gsr_backward(&sc, &cam, &g, &bin, &img, dpix, &gr, bwd_ws, bwd_bytes, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    FILE *out = fopen(argv[2], "wb");
    if (!out) return 1;
    fwrite(&D, 8, 1, out);
#ifndef GSR_CLIENT_CPU
    if (records_mode) {
        dump(out, g.radii, 4 * n); dump(out, g.point_offsets, 4 * n); dump_cols(out, records, n, 0, 2); dump(out, g.depths, 4 * n);
        dump(out, g.cov3D, 24 * n); dump_cols(out, records, n, 6, 3); dump_cols(out, records, n, 2, 4); dump(out, g.clamped_state, 12 * n);
    } else
#endif
    {
    dump(out, g.radii, 4 * n); dump(out, g.point_offsets, 4 * n); dump(out, g.xy, 8 * n); dump(out, g.depths, 4 * n);
    dump(out, g.cov3D, 24 * n); dump(out, g.rgb, 12 * n); dump(out, g.conic_opacity, 16 * n); dump(out, g.clamped_state, 12 * n);
    }
    dump(out, bin.point_list, 4 * (size_t)D); dump(out, bin.ranges, 8 * (size_t)tiles);
    dump(out, img.image, 12 * P); dump(out, img.inv_depth, 4 * P); dump(out, img.final_T, 4 * P); dump(out, img.n_contrib, 4 * P);
    dump(out, gr.dL_dmean3D, 12 * n); dump(out, gr.dL_dscale, 12 * n); dump(out, gr.dL_drot, 16 * n); dump(out, gr.dL_dopacity, 4 * n);
    dump(out, gr.dL_dshs, 192 * n);
#ifndef GSR_CLIENT_CPU
    if (records_mode) {
        const char *acc = (const char *)bwd_ws + gsr_backward_accumulators_offset(N);
        dump_cols(out, acc, n, 0, 3); dump_cols(out, acc, n, 3, 3); dump_cols(out, acc, n, 6, 4);
    } else
#endif
    { dump(out, gr.dL_dcolor, 12 * n); dump(out, gr.dL_dmean2D, 12 * n); dump(out, gr.dL_dconic, 16 * n); }
    fclose(out);
    printf("gsr_client ok: N=%lld D=%lld %dx%d\n", (long long)N, (long long)D, W, H);
    return 0;
}
