// valu_microbench.hip — what does one SIMD of gfx950 sustain in wave64 VALU instructions per second, as a
// function of the waves resident on it?  Settles the "2 or 4 cycles per wave64 VALU instruction" question that
// prices the blend kernels' VALU roofline (MI355X_MICROARCH.md rows 'v_fma_f32 (wave64)'
// and 'vector-instruction ISSUE cost').
//
// Method: every wave runs `iters` iterations of a loop body of 64 INDEPENDENT instructions of one kind (8 accumulator
// chains, each instruction depends only on the one 8 slots earlier), written in inline asm so the compiler can neither
// pack, reorder nor drop them.  Workgroups of 256*w threads, one per CU (grid = #CUs): the w*4 waves of a workgroup are
// dealt round-robin over the CU's 4 SIMDs, so each SIMD hosts w waves.  The placement is VERIFIED, not assumed: every
// wave records HW_REG_HW_ID / HW_REG_XCC_ID and the host counts waves per (xcc, se, cu, simd).
// Output: one JSON object per (op, waves/SIMD) with chip-wide wave-instructions per second, and the same number as
// "cycles per wave-instruction per SIMD" at the nominal clock.  Timing by HIP events around the launch.
//
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_bin/valu_microbench profiles/valu_microbench.hip
//   profiles/_bin/valu_microbench > profiles/r02/valu_microbench.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

enum Op { OP_FMA = 0, OP_EXP, OP_RCP, OP_DPP_QUAD, OP_DPP_ROWMIRROR, OP_DPP_BCAST, OP_CMP, OP_CNDMASK, OP_MIX, OP_PKFMA,
          OP_MUL, OP_MAX, OP_CNDMASK_SGPR, OP_CMP_SGPR, OP_READFIRSTLANE, OP_MOV_DPP, OP_LDS_READ128_BCAST, OP_LDS_ADD_ROWLEADERS,
          OP_CNDMASK_VCC_SET, OP_CNDMASK_E64_VCC, OP_CNDMASK_E32_MIXED, OP_CMP_CNDMASK_PAIRS, N_OPS };
static const char *op_name[N_OPS] = {"v_fma_f32",           "v_exp_f32",           "v_rcp_f32",   "v_add_f32_dpp quad_perm",
                                     "v_add_f32_dpp row_mirror", "v_add_f32_dpp row_bcast15", "v_cmp_lt_f32", "v_cndmask_b32",
                                     "blend-bwd mix (1 exp + 1 rcp + 4 dpp + 2 cmp + 56 fma)", "v_pk_fma_f32",
                                     "v_mul_f32", "v_max_f32", "v_cndmask_b32 e64 (SGPR-pair mask)", "v_cmp_lt_f32 e64 (SGPR-pair result)",
                                     "v_readfirstlane_b32", "v_mov_b32_dpp row_mirror", "ds_read_b128 (all lanes one address)",
                                     "ds_add_f32 (4 lanes, one address)", "v_cndmask_b32 (vcc written before the loop)",
                                     "v_cndmask_b32_e64 with vcc as the explicit mask operand", "4 v_cndmask_b32_e32 + 4 v_fma_f32 alternating",
                                     "4 x (v_cmp_lt_f32 vcc ; v_cndmask_b32_e32 vcc) pairs"};

template <int OP>
__global__ void __launch_bounds__(1024) valu_kernel(float *out, uint32_t *hwid, int iters) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = 0.25f + 1e-3f * (float)((threadIdx.x + i) & 63);
    float b = 0.999f, c = 1e-3f;
    asm volatile("" : "+v"(b), "+v"(c));
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    // (per-wave LDS address, identical in all lanes of the wave)
    const unsigned lds_addr = (unsigned)(uintptr_t)(&lds[(threadIdx.x >> 6) * 64]);
    unsigned long long smask = 0x5555555555555555ull, sink = 0;
    asm volatile("" : "+s"(smask));
    if (OP == OP_CNDMASK_VCC_SET || OP == OP_CNDMASK_E64_VCC || OP == OP_CNDMASK_E32_MIXED) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == OP_FMA) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b), "v"(c));
            } else if (OP == OP_EXP) {
                asm volatile("v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
            } else if (OP == OP_RCP) {
                asm volatile("v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
            } else if (OP == OP_DPP_QUAD) {
#define D8(CTRL)                                                                                                              \
    asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL "\nv_add_f32_dpp %1, %1, %1 " CTRL "\nv_add_f32_dpp %2, %2, %2 " CTRL "\n" \
                 "v_add_f32_dpp %3, %3, %3 " CTRL "\nv_add_f32_dpp %4, %4, %4 " CTRL "\nv_add_f32_dpp %5, %5, %5 " CTRL "\n" \
                 "v_add_f32_dpp %6, %6, %6 " CTRL "\nv_add_f32_dpp %7, %7, %7 " CTRL "\n"                                     \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
                D8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
            } else if (OP == OP_DPP_ROWMIRROR) {
                D8("row_mirror row_mask:0xf bank_mask:0xf");
            } else if (OP == OP_DPP_BCAST) {
                D8("row_bcast:15 row_mask:0xa bank_mask:0xf");
            } else if (OP == OP_CMP) {
                // compare into VCC (no VGPR result): 8 per group
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\n"
                             "v_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b)
                             : "vcc");
            } else if (OP == OP_CNDMASK) {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                             "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                             "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b)
                             : "vcc");
            } else if (OP == OP_PKFMA) {
                // packed fp32: 4 instructions on register pairs = the lane-work of 8 v_fma_f32 (counted as 4 instructions)
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 *p = reinterpret_cast<f2 *>(a);
                f2 bb = {b, b}, cc = {c, c};
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                             : "v"(bb), "v"(cc));
            } else if (OP == OP_MUL) {
                asm volatile("v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b));
            } else if (OP == OP_MAX) {
                asm volatile("v_max_f32 %0, %0, %8\nv_max_f32 %1, %1, %8\nv_max_f32 %2, %2, %8\nv_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\nv_max_f32 %5, %5, %8\nv_max_f32 %6, %6, %8\nv_max_f32 %7, %7, %8\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b));
            } else if (OP == OP_CNDMASK_SGPR) {
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\n"
                             "v_cndmask_b32_e64 %3, %3, %8, %9\nv_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\n"
                             "v_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b), "s"(smask));
            } else if (OP == OP_CNDMASK_VCC_SET) {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                             "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                             "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b));
            } else if (OP == OP_CNDMASK_E64_VCC) {
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\nv_cndmask_b32_e64 %1, %1, %8, vcc\nv_cndmask_b32_e64 %2, %2, %8, vcc\n"
                             "v_cndmask_b32_e64 %3, %3, %8, vcc\nv_cndmask_b32_e64 %4, %4, %8, vcc\nv_cndmask_b32_e64 %5, %5, %8, vcc\n"
                             "v_cndmask_b32_e64 %6, %6, %8, vcc\nv_cndmask_b32_e64 %7, %7, %8, vcc\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b));
            } else if (OP == OP_CNDMASK_E32_MIXED) {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_fma_f32 %1, %1, %8, %9\nv_cndmask_b32 %2, %2, %8, vcc\nv_fma_f32 %3, %3, %8, %9\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\nv_fma_f32 %5, %5, %8, %9\nv_cndmask_b32 %6, %6, %8, vcc\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b), "v"(c));
            } else if (OP == OP_CMP_CNDMASK_PAIRS) {
                asm volatile("v_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %2, %2, %8, vcc\n"
                             "v_cmp_lt_f32 vcc, %5, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cmp_lt_f32 vcc, %7, %8\nv_cndmask_b32 %6, %6, %8, vcc\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b)
                             : "vcc");
            } else if (OP == OP_CMP_SGPR) {
                unsigned long long r0, r1, r2, r3;
                asm volatile("v_cmp_lt_f32_e64 %0, %4, %8\nv_cmp_lt_f32_e64 %1, %5, %8\nv_cmp_lt_f32_e64 %2, %6, %8\nv_cmp_lt_f32_e64 %3, %7, %8\n"
                             "v_cmp_lt_f32_e64 %0, %5, %8\nv_cmp_lt_f32_e64 %1, %6, %8\nv_cmp_lt_f32_e64 %2, %7, %8\nv_cmp_lt_f32_e64 %3, %4, %8\n"
                             : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3)
                             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b));
                sink ^= r0 ^ r1 ^ r2 ^ r3;
            } else if (OP == OP_READFIRSTLANE) {
                unsigned r0, r1, r2, r3;
                asm volatile("v_readfirstlane_b32 %0, %4\nv_readfirstlane_b32 %1, %5\nv_readfirstlane_b32 %2, %6\nv_readfirstlane_b32 %3, %7\n"
                             "v_readfirstlane_b32 %0, %5\nv_readfirstlane_b32 %1, %6\nv_readfirstlane_b32 %2, %7\nv_readfirstlane_b32 %3, %4\n"
                             : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3)
                             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
                sink ^= (unsigned long long)(r0 ^ r1 ^ r2 ^ r3);
            } else if (OP == OP_MOV_DPP) {
                asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 row_mirror row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 row_mirror row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 row_mirror row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
            } else if (OP == OP_LDS_READ128_BCAST) {
                // what the blend loops do per list entry: every lane reads the same 16 B of the staged record
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:16\nds_read_b128 %2, %8 offset:32\nds_read_b128 %3, %8 offset:48\n"
                             "ds_read_b128 %4, %8 offset:64\nds_read_b128 %5, %8 offset:80\nds_read_b128 %6, %8 offset:96\nds_read_b128 %7, %8 offset:112\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                             : "v"(lds_addr)
                             : "memory");
                a[0] += t0.x + t1.y + t2.z + t3.w + t4.x + t5.y + t6.z + t7.w;
            } else if (OP == OP_LDS_ADD_ROWLEADERS) {
                // the blend backward's per-entry accumulation: the 4 row leaders add into ONE LDS word (exec = lanes 0,16,32,48)
                if ((threadIdx.x & 15) == 0) {
                    asm volatile("ds_add_f32 %0, %1\nds_add_f32 %0, %1 offset:4\nds_add_f32 %0, %1 offset:8\nds_add_f32 %0, %1 offset:12\n"
                                 "ds_add_f32 %0, %1 offset:16\nds_add_f32 %0, %1 offset:20\nds_add_f32 %0, %1 offset:24\nds_add_f32 %0, %1 offset:28\n"
                                 :
                                 : "v"(lds_addr), "v"(a[0])
                                 : "memory");
                }
            } else {   // OP_MIX: the instruction classes of one (entry, quadrant) step of the blend backward, 64 per iteration
                if (r == 0) {
                    asm volatile("v_exp_f32 %0, %0\nv_rcp_f32 %1, %1\n"
                                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                 "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                                 "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                                 "v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
                                 "v_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                                 : "v"(b)
                                 : "vcc");
                } else {
                    asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                                 : "v"(b), "v"(c));
                }
            }
        }
    }
    float s = (float)(sink & 1ull) + lds[threadIdx.x & 1023];
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        uint32_t id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        hwid[2 * w] = id;
        hwid[2 * w + 1] = xcc;
    }
}

template <int OP>
static void run(int n_cu, int w, int iters, float *out, uint32_t *hwid, double nominal_hz, bool first) {
    // w waves per SIMD: blocks of min(w,4)*256 threads, ceil(w/4) blocks per CU
    const int per_block = w > 4 ? 4 : w, blocks_per_cu = (w + 3) / 4;
    const int threads = 256 * per_block, grid = n_cu * blocks_per_cu;
    const size_t n_waves = (size_t)grid * threads / 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(grid), dim3(threads), 0, 0, out, hwid, iters / 8);      // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(valu_kernel<OP>, dim3(grid), dim3(threads), 0, 0, out, hwid, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // placement census: waves per SIMD actually observed
    std::vector<uint32_t> ids(2 * n_waves);
    CHECK(hipMemcpy(ids.data(), hwid, ids.size() * 4, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> per_simd;
    for (size_t i = 0; i < n_waves; i++) {
        const uint32_t id = ids[2 * i], xcc = ids[2 * i + 1] & 0xF;
        // HW_ID: [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se
        const uint32_t key = (xcc << 16) | (((id >> 13) & 7) << 12) | (((id >> 12) & 1) << 11) | (((id >> 8) & 15) << 4) | ((id >> 4) & 3);
        per_simd[key]++;
    }
    int mn = 1 << 30, mx = 0;
    for (auto &kv : per_simd) { mn = kv.second < mn ? kv.second : mn; mx = kv.second > mx ? kv.second : mx; }
    const double instr_per_wave = (double)iters * (OP == OP_PKFMA ? 32.0 : 64.0);      // (LDS ops: 64 DS instructions per iteration)
    const double total = instr_per_wave * (double)n_waves;
    const double rate = total / (best * 1e-3);                       // wave-instructions per second, whole chip
    const double simds = (double)per_simd.size();
    const double cyc = nominal_hz * (best * 1e-3) / (instr_per_wave * (double)n_waves / simds);
    printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"simds_used\": %d, \"waves_per_simd_observed\": [%d, %d], \"ms\": %.4f, "
           "\"wave_instr_per_s\": %.4e, \"cycles_per_wave_instr_per_simd_at_nominal_clock\": %.3f}",
           first ? "" : ",\n", op_name[OP], w, (int)simds, mn, mx, best, rate, cyc);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double hz = (double)prop.clockRate * 1e3;
    float *out;
    uint32_t *hwid;
    CHECK(hipMalloc(&out, (size_t)n_cu * 2 * 1024 * 4));
    CHECK(hipMalloc(&hwid, (size_t)n_cu * 2 * 16 * 2 * 4));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"nominal_clock_hz\": %.4e, \"iters\": %d, \"instr_per_iter\": 64,\n \"results\": [\n",
           prop.gcnArchName, n_cu, hz, iters);
    bool first = true;
    const int ws[] = {1, 2, 4, 8};
#define RUN(OP)                                                                \
    for (int w : ws) { run<OP>(n_cu, w, iters, out, hwid, hz, first); first = false; }
    RUN(OP_FMA) RUN(OP_EXP) RUN(OP_RCP) RUN(OP_DPP_QUAD) RUN(OP_DPP_ROWMIRROR) RUN(OP_DPP_BCAST) RUN(OP_CMP) RUN(OP_CNDMASK)
    RUN(OP_MIX) RUN(OP_PKFMA) RUN(OP_MUL) RUN(OP_MAX) RUN(OP_CNDMASK_SGPR) RUN(OP_CNDMASK_VCC_SET) RUN(OP_CMP_SGPR) RUN(OP_READFIRSTLANE)
    RUN(OP_MOV_DPP) RUN(OP_LDS_READ128_BCAST) RUN(OP_LDS_ADD_ROWLEADERS) RUN(OP_CNDMASK_E64_VCC) RUN(OP_CNDMASK_E32_MIXED) RUN(OP_CMP_CNDMASK_PAIRS)
    printf("\n ]}\n");
    return 0;
}
