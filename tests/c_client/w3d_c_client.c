/* A host without PyTorch: plain C over include/w3d.h and the HIP runtime.
 *
 * Test infrastructure (tests/test_gpu_c_client.py): reads one view's inputs from a binary file, runs
 * w3d_forward_stage1 -> w3d_forward_stage2 -> w3d_backward (deterministic mode) with buffers from hipMalloc, and
 * writes the outputs to a second file.  The test feeds the same inputs through the Python binding and compares
 * the two bit for bit: the shared library needs nothing from torch — not its allocator, not its streams.
 *
 * Input file (little endian):  int32 P, H, W, sh_degree, sh_coeffs;  float tanfovx, tanfovy, scale_modifier;
 *   float bg[3], viewmatrix[16], projmatrix[16], campos[3];
 *   float means3D[P*3], shs[P*sh_coeffs*3], opacities[P], scales[P*3], rotations[P*4], dL_dcolor[3*H*W]
 * Output file: uint32 num_visible, num_rendered;  int32 radii[P];  float color[3*H*W], depth[H*W], alpha[H*W];
 *   float dL_dmeans3D[P*3], dL_dmeans2D[P*3], dL_dshs[P*sh_coeffs*3], dL_dopacity[P], dL_dscales[P*3], dL_drots[P*4]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "w3d.h"

#define HIP(x)                                                                          \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define W3D(x)                                                                  \
    do {                                                                        \
        int rc_ = (x);                                                          \
        if (rc_ != W3D_OK) {                                                    \
            fprintf(stderr, "%s: rc %d: %s\n", #x, rc_, w3d_last_error());      \
            return 3;                                                           \
        }                                                                       \
    } while (0)

static void *dev_from(FILE *f, size_t bytes) {
    void *h = malloc(bytes ? bytes : 1), *d = NULL;
    if (!h || fread(h, 1, bytes, f) != bytes) { fprintf(stderr, "short input file\n"); exit(4); }
    if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "hipMalloc / hipMemcpy failed\n");
        exit(5);
    }
    free(h);
    return d;
}

static void *dev_alloc(size_t bytes) {
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(5); }
    return d;
}

static int dump(FILE *f, const void *d, size_t bytes) {
    void *h = malloc(bytes ? bytes : 1);
    if (!h || hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const int bad = fwrite(h, 1, bytes, f) != bytes;
    free(h);
    return bad;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s inputs.bin outputs.bin\n", argv[0]); return 1; }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 1; }
    int32_t hdr[5];
    float fl[3];
    if (fread(hdr, 4, 5, fi) != 5 || fread(fl, 4, 3, fi) != 3) { fprintf(stderr, "bad header\n"); return 4; }
    const int32_t P = hdr[0], H = hdr[1], W = hdr[2], M = hdr[4];
    const size_t HW = (size_t)H * W;

    hipStream_t stream;
    HIP(hipStreamCreate(&stream));            /* a stream of our own: nothing here comes from torch */

    w3d_view v;
    memset(&v, 0, sizeof v);
    v.struct_size = (uint32_t)sizeof v;
    if (w3d_version() / 100 != W3D_ABI_VERSION / 100) { fprintf(stderr, "libw3d_hip ABI %d, header %d\n", w3d_version(), W3D_ABI_VERSION); return 3; }
    v.image_height = H; v.image_width = W;
    v.tanfovx = fl[0]; v.tanfovy = fl[1]; v.scale_modifier = fl[2];
    v.sh_degree = hdr[3]; v.sh_coeffs = M;
    v.tile_cull = 1;
    v.bg = (const float *)dev_from(fi, 3 * 4);
    v.viewmatrix = (const float *)dev_from(fi, 16 * 4);
    v.projmatrix = (const float *)dev_from(fi, 16 * 4);
    v.campos = (const float *)dev_from(fi, 3 * 4);
    float *means3D = (float *)dev_from(fi, (size_t)P * 3 * 4), *shs = (float *)dev_from(fi, (size_t)P * M * 3 * 4);
    float *opac = (float *)dev_from(fi, (size_t)P * 4), *scales = (float *)dev_from(fi, (size_t)P * 3 * 4);
    float *rots = (float *)dev_from(fi, (size_t)P * 4 * 4), *dL_dcolor = (float *)dev_from(fi, 3 * HW * 4);
    fclose(fi);

    uint64_t state_b = 0, scratch_b = 0;
    W3D(w3d_forward_sizes(P, H, W, &state_b, &scratch_b));
    void *state = dev_alloc(state_b), *scratch = dev_alloc(scratch_b);
    int32_t *radii = (int32_t *)dev_alloc((size_t)P * 4);
    uint32_t counts[2] = {0, 0};
    W3D(w3d_forward_stage1(&v, P, means3D, shs, NULL, opac, scales, rots, NULL, radii, state, scratch, counts, stream));
    const uint64_t cap = counts[1] ? counts[1] : 1;       /* stage 1 synchronised: the exact list length is known */
    uint32_t *point_list = (uint32_t *)dev_alloc(cap * 4);
    float *color = (float *)dev_alloc(3 * HW * 4), *depth = (float *)dev_alloc(HW * 4), *alpha = (float *)dev_alloc(HW * 4);
    W3D(w3d_forward_stage2(&v, P, state, scratch, point_list, cap, color, depth, alpha, NULL, 0, NULL, NULL, NULL, NULL, stream));

    /* backward, deterministic mode: the comparison with the other host can then be bit for bit */
    v.deterministic = 1;
    v.det_list_capacity = cap;
    uint64_t bwd_b = 0;
    W3D(w3d_backward_det_sizes(P, cap, &bwd_b));
    void *bscratch = dev_alloc(bwd_b);
    float *g3 = (float *)dev_alloc((size_t)P * 3 * 4), *g2 = (float *)dev_alloc((size_t)P * 3 * 4);
    float *gsh = (float *)dev_alloc((size_t)P * M * 3 * 4), *gop = (float *)dev_alloc((size_t)P * 4);
    float *gsc = (float *)dev_alloc((size_t)P * 3 * 4), *grot = (float *)dev_alloc((size_t)P * 4 * 4);
    W3D(w3d_backward(&v, P, means3D, shs, NULL, opac, scales, rots, NULL, state, point_list, dL_dcolor, NULL, NULL, g3, g2, NULL,
                     gsh, gop, gsc, grot, NULL, bscratch, stream));
    HIP(hipStreamSynchronize(stream));

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 1; }
    int bad = fwrite(counts, 4, 2, fo) != 2;
    bad |= dump(fo, radii, (size_t)P * 4);
    bad |= dump(fo, color, 3 * HW * 4) | dump(fo, depth, HW * 4) | dump(fo, alpha, HW * 4);
    bad |= dump(fo, g3, (size_t)P * 3 * 4) | dump(fo, g2, (size_t)P * 3 * 4) | dump(fo, gsh, (size_t)P * M * 3 * 4);
    bad |= dump(fo, gop, (size_t)P * 4) | dump(fo, gsc, (size_t)P * 3 * 4) | dump(fo, grot, (size_t)P * 4 * 4);
    fclose(fo);
    if (bad) { fprintf(stderr, "could not write the outputs\n"); return 6; }
    printf("w3d_c_client: P=%d %dx%d visible=%u rendered=%u\n", P, W, H, counts[0], counts[1]);
    return 0;
}
