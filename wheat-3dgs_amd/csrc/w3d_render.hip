// w3d_render.hip — per-tile alpha compositing, forward and backward (SURVEY.md Appendix
// A.3, A.4, A.6; replaces the render stage behind reference gaussian_renderer/__init__.py:89-97
// and :194-204).
//
// CDNA4 mapping: ONE wave64 owns ONE 16x16 tile; each lane carries four pixels (the same
// (lx,ly) position inside each 8x8 quadrant of the tile).  Consequences:
//   * no __syncthreads() anywhere — staging through LDS and consuming it are wave-synchronous;
//   * a staged Gaussian (three 16-B LDS broadcast reads) is amortised over 4 pixels per lane;
//   * quadrant-level culling: a whole 8x8 quadrant skips exp/accumulate for a Gaussian when no
//     lane of it can reach alpha >= 1/255 (wave-uniform branch on a ballot);
//   * in the backward pass the per-Gaussian partial sums of the 4 pixels are added in registers
//     first, then reduced over each 16-lane row with bank-masked DPP adds (a reduce-scatter: 21
//     DPP adds for 9-10 values), the row sums go to per-(entry, row) LDS slots with plain stores,
//     and the flush adds the four rows and issues 4 records (36-40 B each) per global atomic
//     instruction — instead of 9 atomics per (pixel, Gaussian) pair.  With view.deterministic the
//     flush stores into the slot of the list entry instead (det_gather_kernel adds them per
//     Gaussian in tile order).
// One tile-wave per workgroup (W3D_RW); the wave -> tile map keeps the tiles of one XCD contiguous so
// neighbouring tiles (which share Gaussians) hit the same L2.
#include "w3d_common.h"

#include <type_traits>

namespace {

#ifndef W3D_RW
#define W3D_RW 1          // tile-waves per workgroup of the blend kernels: 7500 one-wave workgroups balance better over 256 CUs
                          // than 1875 four-wave ones (blend backward 0.538 -> 0.518 ms; 2 waves: 0.525)
#endif
#ifndef W3D_TILE_ORDER
#define W3D_TILE_ORDER 1  // blend backward: every XCD takes its tiles longest walk first (tile_order_kernel)
#endif
#define LOG2E W3D_LOG2E
#define W3D_FLASH_LABELS 4     // FlashSplat: labels per tile that take the LDS row-sum path (more: one wave reduction per label and entry)
#define W3D_ACC_SLOTS 128      // backward: (entry of a 32-entry half batch) x (16-lane row) slots per accumulated value
#ifndef W3D_ACC_PITCH
#define W3D_ACC_PITCH 132      // ... and the distance (floats) between two values' slot arrays.  With 128 every value's array starts
                               // on the same LDS bank: the 16 storing lanes of an entry (4 values x 4 rows) hit 4 banks 4-way, and
                               // the flush's 16 value-lanes read one bank group 16-way — SQ_LDS_BANK_CONFLICT = 12 % of the
                               // kernel's cycles on the densified scene (profiles/r04/lds_counters.txt).  132 (16-B aligned for
                               // the flush's b128 reads) spreads the values over banks 4k mod 32.
#endif
#ifndef W3D_BWD_SFORM
#define W3D_BWD_SFORM 1        // blend backward: the suffix colour carried as its product with dL/dpixel (one scalar per pixel)
#endif
#ifndef W3D_FWD_HOIST_IDX
#define W3D_FWD_HOIST_IDX 1
#endif
#ifndef W3D_VCC_SELECT
#define W3D_VCC_SELECT 1       // per-lane selects of the blend loops through VCC: v_cndmask_b32_e32 issues at full rate, the e64 form
                               // (mask in an SGPR pair — what the compiler picks when several lane masks are alive) at half rate
                               // (profiles/r02/valu_microbench.json); one s_mov_b64 vcc on the scalar pipe feeds 2-3 selects
#endif

// Selects with ONE lane mask in their VCC form (see W3D_VCC_SELECT).  fwd_apply: w = m ? aT : 0, T = m ? T_new : T,
// last = m ? idx : last;  kill_where: x = m ? v : x;  bwd_mask2: a = m ? a : 0, g = m ? g : 0.
__device__ __forceinline__ float fwd_apply(uint64_t m, float aT, float &T, float T_new, uint32_t &last, uint32_t idx) {
    float w;
    asm volatile("s_mov_b64 vcc, %3\n\t"
                 "v_cndmask_b32_e32 %0, 0, %4, vcc\n\t"
                 "v_cndmask_b32_e32 %1, %1, %5, vcc\n\t"
                 "v_cndmask_b32_e32 %2, %2, %6, vcc"
                 : "=&v"(w), "+v"(T), "+v"(last) : "s"(m), "v"(aT), "v"(T_new), "v"(idx) : "vcc");   // (w is written before T_new / idx are read)
    return w;
}
__device__ __forceinline__ void kill_where(uint64_t m, float &x, float v) {
    asm volatile("s_mov_b64 vcc, %1\n\t"
                 "v_cndmask_b32_e32 %0, %0, %2, vcc"
                 : "+v"(x) : "s"(m), "v"(v) : "vcc");
}
__device__ __forceinline__ void bwd_mask2(uint64_t m, float &a, float &g) {
    asm volatile("s_mov_b64 vcc, %2\n\t"
                 "v_cndmask_b32_e32 %0, 0, %0, vcc\n\t"
                 "v_cndmask_b32_e32 %1, 0, %1, vcc"
                 : "+v"(a), "+v"(g) : "s"(m) : "vcc");
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, ROW_MASK, 0xF, true));
}
// sum over the 64 lanes; the result is valid in lane 63 (and returned broadcast through readlane)
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);       // row_half_mirror
    v += dpp_mov<0x140>(v);       // row_mirror  -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

// wave -> (tile, part of the tile).  Without a schedule: tile = wave index with each XCD's tiles contiguous (blocks are dealt round-robin
// over the 8 XCDs; speed only, never correctness), the whole tile.  With one (tile_schedule_kernel): block b runs entry
// order[(b % 8) * cap + b / 8] = tile | part << 29, 0xFFFFFFFF = nothing.  part: 0 the whole tile, 1 / 2 its upper / lower two
// 8x8 quadrants, 3..6 one quadrant — `qmask` is the set of quadrants (bit k = quadrant k) this wave owns.
__device__ __forceinline__ uint32_t wave_to_tile(uint32_t T, uint32_t &tile, uint32_t &qmask, const uint32_t *__restrict__ order = nullptr,
                                                 uint32_t cap = 0) {
    const uint32_t b = blockIdx.x;
    qmask = 0xFu;
    if (order) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)order[(b & 7u) * cap + (b >> 3)]);
        if (e == 0xFFFFFFFFu) return 0u;
        tile = e & 0x1FFFFFFFu;
        const uint32_t part = e >> 29;
        qmask = part == 0u ? 0xFu : part == 1u ? 0x3u : part == 2u ? 0xCu : (1u << (part - 3u));
        return tile < T;
    }
    const uint32_t per_xcd = (gridDim.x + 7) / 8;
    const uint32_t logical_block = (b & 7u) * per_xcd + (b >> 3);
    // (an SGPR: the per-tile loads become scalar loads and every loop bound derived from them stays scalar)
    tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(logical_block * W3D_RW + (threadIdx.x >> 6)));
    return tile < T;
}

// tile -> list cell (w3d_view.list_share).  The grid is the one the forward that wrote this state BUILT its lists on — counters[6] =
// lsx | lsy << 8, left by the tile scan — not what the caller's view says now: a backward or a re-blend handed a different
// list_share / deterministic flag than the forward would otherwise index tile_start on the wrong grid and silently return
// garbage (the host-side values only size the launch).
__device__ __forceinline__ uint32_t list_of_tile(uint32_t tile, uint32_t gx, const uint32_t *__restrict__ counters) {
    const uint32_t code = counters[6];
    if (code == 0u) return tile;
    const uint32_t lsx = code & 0xFFu, lsy = (code >> 8) & 0xFFu;
    const uint32_t lgx = (gx + (1u << lsx) - 1u) >> lsx;
    const uint32_t ty = tile / gx, tx = tile - ty * gx;
    return (ty >> lsy) * lgx + (tx >> lsx);
}

struct StagedLDS {
    float4 a[64];  // x, y, pmin2 (log2-domain power below which alpha < 1/255 for sure), id bits
    float4 b[64];  // conic.x, conic.y, conic.z, opacity            (only the backward's flush reads it)
    float4 c[64];  // r, g, b, depth
    float4 d[64];  // -0.5*log2e*conic.x, -log2e*conic.y, -0.5*log2e*conic.z, opacity: the exponent in the
                   // log2 domain is dx*(d.x*dx + d.y*dy) + d.z*dy*dy — 5 VALU instead of 7 + 1 for the exp2 scale
};

// Which of the tile's four 8x8 quadrants can this Gaussian touch?  The minimum of q(d) = 0.5 d^T Conic d over
// the quadrant's pixel rectangle is compared with tau = log(255 o) + ~1e-3 (exactly the per-tile test of the
// binning stage, w3d_preprocess.hip footprint_hits_tile, on a quarter tile): the minimum sits at the centre if
// that is inside, otherwise on one of the four edges, where it is a clamped 1-D parabola minimum.  Evaluated
// once per list entry by the staging lane; the blend loops then skip a quadrant on a scalar bit test instead
// of evaluating the exponent for 64 pixels.  Conservative by the 1e-3 margin plus the exponent's own fp32 evaluation noise
// over the tile, so no result changes.
__device__ __forceinline__ uint32_t quadrant_mask(float mx, float my, float A, float B, float C, float pmin,
                                                  float tx0, float ty0) {
    if (!(A > 0.f && C > 0.f)) return pmin <= 0.f ? 0xFu : 0u;
#ifdef W3D_NO_QUADMASK          // (test aid: profiles/cull_exactness_probe.py compares the product against a build without the masks)
    return 0xFu;
#endif
    // pmin = -log(255 o) - 1e-4; + the fp32 evaluation noise of the exponent over this tile (w3d_q_noise: needles only)
    const float Dx = fmaxf(fabsf(mx - tx0), fabsf(mx - (tx0 + 15.f))), Dy = fmaxf(fabsf(my - ty0), fabsf(my - (ty0 + 15.f)));
    const float tau = -pmin + 9e-4f + w3d_q_noise(A, B, C, Dx, Dy);
    const float iA = __builtin_amdgcn_rcpf(A), iC = __builtin_amdgcn_rcpf(C);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float x0 = tx0 + (float)((k & 1) * 8), y0 = ty0 + (float)((k >> 1) * 8);
        const float dxl = mx - (x0 + 7.f), dxh = mx - x0, dyl = my - (y0 + 7.f), dyh = my - y0;
        const bool inside = dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f;
        float dy = fminf(fmaxf(-B * dxl * iC, dyl), dyh);
        float best = 0.5f * (A * dxl * dxl + C * dy * dy) + B * dxl * dy;
        dy = fminf(fmaxf(-B * dxh * iC, dyl), dyh);
        best = fminf(best, 0.5f * (A * dxh * dxh + C * dy * dy) + B * dxh * dy);
        float dx = fminf(fmaxf(-B * dyl * iA, dxl), dxh);
        best = fminf(best, 0.5f * (A * dx * dx + C * dyl * dyl) + B * dx * dyl);
        dx = fminf(fmaxf(-B * dyh * iA, dxl), dxh);
        best = fminf(best, 0.5f * (A * dx * dx + C * dyh * dyh) + B * dx * dyh);
        if (inside || !(best > tau)) m |= 1u << k;
    }
    return m;
}

// Returns this lane's entry mask: bits 0..3 = quadrants it can touch.  The masks stay in registers; the loops read
// them with v_readlane and skip entries that touch no quadrant on a scalar bit scan.
// record words 2, 3 of a Gaussian (its published tile rect) -> may this tile blend it at all?  (only a tile that reads a SHARED
// list ever meets an entry whose rect it is not in)
__device__ __forceinline__ bool tile_in_rect(float lo_bits, float hi_bits, uint32_t tcol, uint32_t trow) {
    const uint32_t lo = __float_as_uint(lo_bits), hi = __float_as_uint(hi_bits);
    return tcol >= (lo & 0xFFFFu) && tcol < (hi & 0xFFFFu) && trow >= (lo >> 16) && trow < (hi >> 16);
}
// alpha = min(0.99, o*exp(power)) >= 1/255 needs power >= -log(255 o) =: pmin; 1e-4 slack covers the rounding of the fast exp,
// so skipping below pmin never changes a result
__device__ __forceinline__ float pmin_of(float opacity) { return (opacity > 0.f) ? (-__logf(255.0f * opacity) - 1e-4f) : 1.0f; }

__device__ __forceinline__ uint32_t stage_entries(StagedLDS &s, uint32_t lane, uint32_t n, const uint32_t *__restrict__ list,
                                              const float4 *__restrict__ grec, uint32_t tcol, uint32_t trow) {
    uint32_t q = 0u;
    if (lane < n) {
        // ONE 64-B line per entry, (almost) in the staged layout (W3DLayout::o_grec)
        const uint32_t g = list[lane];
        const float4 *r = grec + 4 * (size_t)g;
        const float4 a = r[0], co = r[1], cd = r[2], d = r[3];
        const float pmin = pmin_of(co.w);
        s.a[lane] = make_float4(a.x, a.y, pmin * LOG2E, __uint_as_float(g));
        s.b[lane] = co;
        s.c[lane] = cd;
        s.d[lane] = d;
        if (tile_in_rect(a.z, a.w, tcol, trow))
            q = quadrant_mask(a.x, a.y, co.x, co.y, co.z, pmin, (float)(tcol * W3D_TILE), (float)(trow * W3D_TILE));
    }
    __builtin_amdgcn_wave_barrier();
    return q;
}

// ------------------------------------------------------------------------------ forward
#ifndef W3D_FWD_OCC
#define W3D_FWD_OCC 6     // waves per SIMD the forward is compiled for (80 VGPRs as it falls out; 8 would need <= 64)
#endif
// DA: the depth and alpha images are wanted (every API path but the fused training step, which only feeds the colour image to its
// loss: w3d_forward_stage2 with out_depth = out_alpha = NULL drops their two accumulations per pixel and entry and their stores)
template <bool FLASH, bool DA = true>
__global__ void __launch_bounds__(64 * W3D_RW, FLASH ? 4 : W3D_FWD_OCC)
render_fwd_kernel(uint32_t T, uint32_t gx, int W, int H, const uint32_t *__restrict__ tile_start,
                  const uint32_t *__restrict__ point_list, const float4 *__restrict__ grec, const float *__restrict__ bg,
                  float *__restrict__ out_color, float *__restrict__ out_depth, float *__restrict__ out_alpha,
                  float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                  const float *__restrict__ gt_mask, int num_obj, int P, float *__restrict__ used_count,
                  int32_t *__restrict__ contrib_num, uint32_t list_cap, uint32_t *__restrict__ counters,
                  uint32_t *__restrict__ tile_walk, const uint32_t *__restrict__ tile_order, uint32_t order_cap,
                  uint32_t *__restrict__ walk_hint, uint32_t lshift, uint32_t lgx) {
    __shared__ StagedLDS lds[W3D_RW];
    __shared__ int s_labels[W3D_RW][FLASH ? 256 : 1];
    // FlashSplat: row sums of the per-entry, per-label weights of the current batch: [label slot][entry][16-lane row]
    __shared__ __align__(16) float s_facc[W3D_RW][FLASH ? W3D_FLASH_LABELS * 64 * 4 : 4];
    uint32_t tile, qmask;
    if (!wave_to_tile(T, tile, qmask, tile_order, order_cap)) return;
    // the list buffer may be smaller than the lists (speculative sizing, see w3d_forward_stage2): never read
    // past it; the capacity is published for the backward pass
    if (tile == 0 && (threadIdx.x & 63) == 0) counters[3] = list_cap;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    StagedLDS &s = lds[wv];
    const uint32_t tx0 = (tile % gx) * W3D_TILE, ty0 = (tile / gx) * W3D_TILE;
    const uint32_t lx = lane & 7, ly = lane >> 3;
    float pxf[4], pyf[4];
    // hi[k]: upper bound of the exponent a pixel still accepts — 0 while it is live, -inf once it is done (saturated or
    // outside the image).  A float so that "live && power <= 0" is ONE v_cmp whose lane mask feeds the ballot directly
    // (a bool array would be decoded from a 0/1 VGPR and re-encoded for every ballot).
    bool inside[4];
    float hi[4];
    float Tr[4], C0[4], C1[4], C2[4], D[4], A[4];
    uint32_t last[4];
    int napplied[4];
    int label[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t px = tx0 + (k & 1) * 8 + lx, py = ty0 + (k >> 1) * 8 + ly;
        pxf[k] = (float)px; pyf[k] = (float)py;
        inside[k] = (px < (uint32_t)W) && (py < (uint32_t)H) && ((qmask >> k) & 1u);       // (a part-wave owns some quadrants only)
        hi[k] = inside[k] ? 0.f : -INFINITY;
        Tr[k] = 1.f; C0[k] = C1[k] = C2[k] = D[k] = A[k] = 0.f;
        last[k] = 0; napplied[k] = 0;
        label[k] = -1;
        if (FLASH && gt_mask && inside[k]) {
            const int l = (int)gt_mask[(size_t)py * W + px];
            label[k] = (l >= 0 && l <= num_obj) ? l : -1;
        }
    }
    // distinct labels present in this tile (FlashSplat scatter loops over them), ascending; the first W3D_FLASH_LABELS of
    // them also live in scalar registers: a wheat-head instance map puts one to three labels into a tile (background, a
    // head, its neighbour), and those tiles take the row-sum path below for every label
    int nlabels = 0;
    int lab[W3D_FLASH_LABELS];
#pragma unroll
    for (int i = 0; i < W3D_FLASH_LABELS; i++) lab[i] = -1;
    if (FLASH && gt_mask && used_count) {
        int cur = -1;
        for (;;) {
            int m = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (label[k] > cur) m = min(m, label[k]);
            m = __builtin_amdgcn_readfirstlane(wave_min_i32(m));
            if (m == 0x7fffffff) break;
            if (lane == 0) s_labels[wv][nlabels] = m;
#pragma unroll
            for (int i = 0; i < W3D_FLASH_LABELS; i++)
                if (nlabels == i) lab[i] = m;
            nlabels++;
            cur = m;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the list this tile reads: its own, or (w3d_view.list_share) the one it shares with its neighbours in the list cell —
    // the entries that cannot touch THIS tile have an empty quadrant mask and are skipped on the scalar bit scan below
    (void)lshift; (void)lgx;
    const uint32_t ltile = list_of_tile(tile, gx, counters);
    const uint32_t start = min(tile_start[ltile], list_cap), end = min(tile_start[ltile + 1], list_cap);
    for (uint32_t base = start; base < end; base += 64) {
        if ((w3d_ballot(hi[0] == 0.f) | w3d_ballot(hi[1] == 0.f) | w3d_ballot(hi[2] == 0.f) | w3d_ballot(hi[3] == 0.f)) == 0ull) break;
        const uint32_t n = min(64u, end - base);
        uint32_t myq = stage_entries(s, lane, n, point_list + base, grec, tx0 / W3D_TILE, ty0 / W3D_TILE);
        {
            // quadrants whose 64 pixels are all done (saturated or outside the image) take no further entries: their bit is
            // cleared from every entry mask of the batch, so entries that only touch finished quadrants cost nothing
            uint32_t live = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) live |= (w3d_ballot(hi[k] == 0.f) != 0ull) ? (1u << k) : 0u;
            myq &= live;
        }
        uint64_t todo = w3d_ballot(myq != 0u);
        // FlashSplat scatter, tiles with up to W3D_FLASH_LABELS labels (every tile of a binary mask, nearly every tile of an
        // instance map): per label the weights of an entry are summed over each 16-lane row in registers (4 DPP stages), the
        // four row leaders store the row sums into the entry's LDS slots (plain stores: every slot is written once per
        // batch), and the batch is flushed with ONE 64-lane atomic per label instead of one single-lane atomic per entry
        const bool facc_path = FLASH && gt_mask && used_count && nlabels >= 1 && nlabels <= W3D_FLASH_LABELS;
        uint64_t ftouched = 0ull;
        while (todo) {
            const uint32_t j = (uint32_t)__builtin_ctzll(todo);            // front to back
            todo &= todo - 1ull;
            const uint32_t qm = (uint32_t)__builtin_amdgcn_readlane((int)myq, (int)j);
            const float4 ea = s.a[j], ed = s.d[j], ec = s.c[j];
#if W3D_FWD_HOIST_IDX
            // (a VGPR copy made ONCE per entry: handed to the select helper as a scalar, the index is re-materialised by a v_mov in
            //  every quadrant block)
            uint32_t contributor;
            asm volatile("v_mov_b32 %0, %1" : "=v"(contributor) : "s"(base - start + j + 1));
#else
            const uint32_t contributor = base - start + j + 1;
#endif
            float wk[4] = {0.f, 0.f, 0.f, 0.f};
            bool any_applied = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!(qm & (1u << k))) continue;
                const float dx = ea.x - pxf[k], dy = ea.y - pyf[k];
                const float power = fmaf(ed.z * dy, dy, fmaf(ed.y, dy, ed.x * dx) * dx);      // log2 domain
                // (ballots of plain compares ANDed as scalars: the ballot of a compound predicate is materialised in a VGPR)
                const uint64_t m_cand = w3d_ballot(power <= hi[k]) & w3d_ballot(power >= ea.z);
                if (m_cand == 0ull) continue;                                                      // whole quadrant untouched
                float alpha = ed.w * __builtin_amdgcn_exp2f(power);
                alpha = fminf(0.99f, alpha);
                const float test_T = Tr[k] * (1.f - alpha);
#if W3D_VCC_SELECT
                // ok = cand && alpha >= 1/255; stop = ok && T' < 1e-4 (the entry is NOT applied and the pixel is done);
                // apply = ok && !stop — as lane masks on the scalar pipe, the selects in their full-rate VCC form
                const uint64_t m_ok = m_cand & w3d_ballot(alpha >= (1.0f / 255.0f));
                const uint64_t m_small = w3d_ballot(test_T < 0.0001f);
                const float w = fwd_apply(m_ok & ~m_small, alpha * Tr[k], Tr[k], test_T, last[k], contributor);
                kill_where(m_ok & m_small, hi[k], -INFINITY);
                const bool apply = FLASH && w != 0.f;           // (alpha >= 1/255 and T >= 1e-4: an applied weight is never 0)
#else
                const bool cand = power <= hi[k] && power >= ea.z;
                const bool ok = cand && alpha >= (1.0f / 255.0f);
                const bool stop = ok && test_T < 0.0001f;
                const bool apply = ok && !stop;
                const float w = apply ? alpha * Tr[k] : 0.f;
                Tr[k] = apply ? test_T : Tr[k];
                last[k] = apply ? contributor : last[k];
                hi[k] = stop ? -INFINITY : hi[k];
#endif
                C0[k] += ec.x * w; C1[k] += ec.y * w; C2[k] += ec.z * w;
                if (DA) { D[k] += ec.w * w; A[k] += w; }
                if (FLASH) { wk[k] = w; napplied[k] += apply ? 1 : 0; any_applied = any_applied || apply; }
            }
            if (FLASH && gt_mask && used_count) {
                if (w3d_ballot(any_applied) != 0ull && facc_path) {
                    ftouched |= 1ull << j;
#pragma unroll
                    for (int li = 0; li < W3D_FLASH_LABELS; li++) {
                        if (li >= nlabels) break;
                        float part = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; k++) part += (label[k] == lab[li]) ? wk[k] : 0.f;
                        part += dpp_mov<0xB1>(part);
                        part += dpp_mov<0x4E>(part);
                        part += dpp_mov<0x141>(part);
                        part += dpp_mov<0x140>(part);
                        if ((lane & 15u) == 0u) s_facc[wv][(li * 64 + j) * 4 + (lane >> 4)] = part;
                    }
                } else if (w3d_ballot(any_applied) != 0ull) {
                    // more labels in one tile than slots: one wave reduction and one atomic per label and entry
                    const uint32_t g = __float_as_uint(ea.w);
                    for (int li = 0; li < nlabels; li++) {
                        const int L = s_labels[wv][li];
                        float part = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; k++) part += (label[k] == L) ? wk[k] : 0.f;
                        const float tot = wave_sum(part);
                        if (lane == 0 && tot != 0.f) atomicAdd(&used_count[(size_t)L * P + g], tot);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (facc_path && ((ftouched >> lane) & 1ull)) {
            const uint32_t g = __float_as_uint(s.a[lane].w);
#pragma unroll
            for (int li = 0; li < W3D_FLASH_LABELS; li++) {
                if (li >= nlabels) break;
                const float4 r4 = *reinterpret_cast<const float4 *>(&s_facc[wv][(li * 64 + lane) * 4]);
                const float v = (r4.x + r4.y) + (r4.z + r4.w);
                if (v != 0.f) atomicAdd(&used_count[(size_t)lab[li] * P + g], v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t HW = (size_t)H * W;
    {
        // how far into its list this tile blended anything: the length of the backward's reverse walk (its work)
        const uint32_t m = wave_max_u32(max(max(last[0], last[1]), max(last[2], last[3])));
        // (the part-waves of a split tile each report their own quadrants: the tile's length is the maximum — tile_walk is zeroed by
        //  the tile scan of this forward; the per-camera hint for the NEXT render of this view is not, and takes whichever part's
        //  length is stored last: a lower bound of the tile's, good enough for a hint)
        if (lane == 0) {
            if (qmask == 0xFu) tile_walk[tile] = m; else atomicMax(&tile_walk[tile], m);
            if (walk_hint) walk_hint[tile] = m;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (inside[k]) {
            const size_t pix = (size_t)(uint32_t)pyf[k] * W + (uint32_t)pxf[k];
            out_color[pix] = C0[k] + Tr[k] * bg0;
            out_color[HW + pix] = C1[k] + Tr[k] * bg1;
            out_color[2 * HW + pix] = C2[k] + Tr[k] * bg2;
            if (DA) { out_depth[pix] = D[k]; out_alpha[pix] = A[k]; }
            final_T[pix] = Tr[k];
            n_contrib[pix] = last[k];
            if (FLASH && contrib_num) contrib_num[pix] = napplied[k];
        }
    }
}

// ------------------------------------------------------------------------------ backward
// Nine (ten with depth) per-Gaussian sums are reduced over the wave stage by stage so that the
// DPP adds of different values interleave (no hazard nops); the totals land in lane 63.
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0xB1>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x4E>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x141>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x140>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x142, 0xA>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x143, 0xC>(v[i]);
}

// first four DPP stages only: every lane ends up with the sum over its 16-lane row
template <int N>
__device__ __forceinline__ void row_sum_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0xB1>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x4E>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x141>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += dpp_mov<0x140>(v[i]);
}

struct Staged { float4 a, b, c; uint32_t g; };
__device__ __forceinline__ Staged gather_entry(uint32_t g, const float4 *__restrict__ grec) {
    const float4 *r = grec + 4 * (size_t)g;
    Staged s;
    s.a = r[0]; s.b = r[1]; s.c = r[2];       // (the fourth quarter of the line is three multiples of b: recomputed at staging, 4 VGPRs less in flight)
    s.g = g;
    return s;
}

// HAS_DA: gradients w.r.t. the depth and alpha images are present.
#ifndef W3D_BWD_OCC
#define W3D_BWD_OCC 4
#endif
template <bool HAS_DA, bool DET>
__global__ void __launch_bounds__(64 * W3D_RW, HAS_DA ? 3 : W3D_BWD_OCC)   // 2nd argument = waves per SIMD: caps VGPRs at 168 / 128
render_bwd_kernel(uint32_t T, uint32_t gx, int W, int H, const uint32_t *__restrict__ tile_start,
                  const uint32_t *__restrict__ point_list, const float4 *__restrict__ grec, const float *__restrict__ bg,
                  const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                  const float *__restrict__ dL_dcolor, const float *__restrict__ dL_ddepth,
                  const float *__restrict__ dL_dalpha_px, float *__restrict__ grad2d,
                  const uint32_t *__restrict__ counters, float *__restrict__ inst, uint32_t inst_cap,
                  const uint32_t *__restrict__ tile_order, uint32_t order_cap, uint32_t lshift, uint32_t lgx) {
    constexpr int NV = HAS_DA ? 10 : 9;
    __shared__ StagedLDS lds[W3D_RW];
    // row sums of the current half batch: acc[value][entry * 4 + row].  Every (entry, row) slot is written exactly once
    // per half batch by that row's leader lane — plain stores, no LDS atomics (measured on gfx950: a ds_add_f32 of four lanes
    // on one address occupies the CU's LDS for ~15 cycles, nine of them per entry cost more than the entry's arithmetic)
    __shared__ __align__(16) float acc_all[W3D_RW][NV * W3D_ACC_PITCH];
    uint32_t tile, qmask;
    if (!wave_to_tile(T, tile, qmask, tile_order, order_cap)) return;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    StagedLDS &s = lds[wv];
    float *acc = acc_all[wv];
    const uint32_t tx0 = (tile % gx) * W3D_TILE, ty0 = (tile / gx) * W3D_TILE;
    const uint32_t lx = lane & 7, ly = lane >> 3;
    const size_t HW = (size_t)H * W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    // per-pixel state, 4 pixels per lane.  ar* = colour (depth, alpha) composited BEHIND the entry being
    // visited; it is advanced right after an entry is processed (A <- a*c + (1-a)*A), which is the same
    // arithmetic, in the same order, as the textbook "last_alpha / last_color" formulation but needs no
    // copies of the previous contributor.  Pixel coordinates are rebuilt from the lane id on the fly.
    const float pxb = (float)(tx0 + lx), pyb = (float)(ty0 + ly);
    float Tr[4], Tfin[4];
    float dp0[4], dp1[4], dp2[4], dpd[4], dpa[4];
#if W3D_BWD_SFORM
    // The colour (depth, alpha) composited BEHIND the entry being visited enters dL/dalpha only through its product with
    // dL/dpixel, so ONE scalar per pixel is carried, S = A . dL/dpixel, advanced by the same recurrence (A <- a c + (1-a) A
    // gives S <- S + a (c . dL/dpixel - S)): 5 instructions per pixel and entry instead of 9 (15 -> 7 with depth / alpha
    // gradients), 8 (16) registers fewer.  The other legal form of the suffix recurrence (the oracle's exp-mode bit 3).
    float S[4];
#else
    float ar0[4], ar1[4], ar2[4], ard[4], ara[4];
#endif
    uint32_t last[4];
    uint32_t maxc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t px = tx0 + (k & 1) * 8 + lx, py = ty0 + (k >> 1) * 8 + ly;
        const bool in = (px < (uint32_t)W) && (py < (uint32_t)H) && ((qmask >> k) & 1u);    // (a part-wave walks for its own quadrants only)
        const size_t pix = (size_t)py * W + px;
        Tfin[k] = in ? final_T[pix] : 0.f;
        Tr[k] = Tfin[k];
        last[k] = in ? n_contrib[pix] : 0u;
        dp0[k] = in ? dL_dcolor[pix] : 0.f;
        dp1[k] = in ? dL_dcolor[HW + pix] : 0.f;
        dp2[k] = in ? dL_dcolor[2 * HW + pix] : 0.f;
        dpd[k] = (HAS_DA && in && dL_ddepth) ? dL_ddepth[pix] : 0.f;
        dpa[k] = (HAS_DA && in && dL_dalpha_px) ? dL_dalpha_px[pix] : 0.f;
#if W3D_BWD_SFORM
        S[k] = 0.f;
#else
        ar0[k] = ar1[k] = ar2[k] = ard[k] = ara[k] = 0.f;
#endif
        maxc = max(maxc, last[k]);
    }
    maxc = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(maxc));   // (an SGPR: everything derived from it stays scalar)
    if (maxc == 0) return;
    const uint32_t cap = counters[3];
    (void)lshift; (void)lgx;
    const uint32_t start = min(tile_start[list_of_tile(tile, gx, counters)], cap);     // (shared lists: see the forward)
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const bool has_bg = (bg0 != 0.f) || (bg1 != 0.f) || (bg2 != 0.f);     // wave-uniform: black background skips the term
    const int nb = (int)((maxc + 63) / 64);
    // (Scalar bounds on the quadrants' walks were measured and dropped: "idx0 < last[k] holds for the whole quadrant while
    //  idx0 < min last[k]" saved a compare per quadrant at the price of a branch — 3-4 % slower than always comparing — and
    //  "skip quadrant k while idx0 >= max last[k]" 5 % slower.  Every wave-uniform branch in this loop costs the wave a
    //  VALU -> SALU round trip that 2-4 resident waves do not hide: the 0.99 cap, too, is an unconditional v_min now, and
    //  the second early exit (no lane reaches alpha >= 1/255 after the exponential) is gone — together another 3 %.)
    // where the row sums of this lane go (see the reduce-scatter below): the first lane of every bank (4 lanes) stores the
    // three values its bank ended up with
    const uint32_t l4 = (lane >> 2) & 3u, row = lane >> 4;
    const bool storer = (lane & 3u) == 0u;
    const uint32_t offA = ((l4 >> 1) + 5u * (l4 & 1u)) * W3D_ACC_PITCH + row;      // values 0, 5, 1, 6
    const uint32_t offB = offA + 2u * W3D_ACC_PITCH;                               // values 2, 7, 3, 8
    const uint32_t offC = (4u + 5u * (l4 & 1u)) * W3D_ACC_PITCH + row;             // values 4, 9
    const bool storeC = storer && (l4 == 0u || (HAS_DA && l4 == 1u));
    // software pipeline over the 64-entry batches (walked back to front): while batch b is consumed from
    // LDS, the records of batch b-1 are already in flight to registers and the ids of batch b-2 to `ids`.
    auto batch_id = [&](int b) -> uint32_t {
        const uint32_t i = (uint32_t)b * 64u + lane;
        if (!(b >= 0 && i < maxc)) return 0xFFFFFFFFu;
        return point_list[min(start + i, cap - 1u)];
    };
#ifdef W3D_BWD_STATS
    uint32_t st_entries = 0, st_quads = 0, st_exp = 0, st_full = 0, st_staged = 0;
#endif
    // The walk exists twice — black background (the reference's default, arguments/__init__.py: white_background False; the
    // background term and its per-quadrant scalar branch are compiled out) and any other — chosen once per wave: the time of
    // this kernel is the number of instructions it issues, and `if (has_bg)` cost two of them per
    // quadrant evaluation even when the background is black.
    auto walk = [&](auto bg_tag) {
    constexpr bool BG = decltype(bg_tag)::value;
    Staged nxt;
    {
        const uint32_t g = batch_id(nb - 1);
        nxt = (g != 0xFFFFFFFFu) ? gather_entry(g, grec) : Staged{};
    }
    uint32_t ids = batch_id(nb - 2);
    for (int b = nb - 1; b >= 0; b--) {
        const uint32_t n = min(64u, maxc - (uint32_t)b * 64u);
        const float pmin = pmin_of(nxt.b.w);
        s.a[lane] = make_float4(nxt.a.x, nxt.a.y, pmin * LOG2E, __uint_as_float(nxt.g)); s.b[lane] = nxt.b; s.c[lane] = nxt.c;
        s.d[lane] = make_float4(-0.5f * LOG2E * nxt.b.x, -LOG2E * nxt.b.y, -0.5f * LOG2E * nxt.b.z, nxt.b.w);
        // this lane's entry: quadrant mask (bits 0..3); it stays in the register — the walk below reads it with v_readlane
        // and skips entries without any quadrant on a scalar bit scan
        uint32_t myq = 0u;
        if (lane < n && tile_in_rect(nxt.a.z, nxt.a.w, tx0 / W3D_TILE, ty0 / W3D_TILE)) {
            myq = quadrant_mask(nxt.a.x, nxt.a.y, nxt.b.x, nxt.b.y, nxt.b.z, pmin, (float)tx0, (float)ty0) & qmask;
        }
        const uint64_t todo_all = w3d_ballot(myq != 0u);
#ifdef W3D_BWD_STATS
        st_staged += n;
#endif
        __builtin_amdgcn_wave_barrier();
        if (b > 0) {
            nxt = (ids != 0xFFFFFFFFu) ? gather_entry(ids, grec) : Staged{};
            ids = batch_id(b - 2);
        }
        // the batch is consumed in two halves of 32 entries, each followed by its flush (the row-sum slots are per half)
#pragma unroll 1
        for (int jlo = 32; jlo >= 0; jlo -= 32) {
            uint32_t todo = (uint32_t)(todo_all >> jlo);
            uint32_t touched = 0u;
            while (todo) {
                const uint32_t jl = 31u - (uint32_t)__builtin_clz(todo);             // highest entry first: back to front
                todo &= ~(1u << jl);
                const uint32_t j = (uint32_t)jlo + jl;
                const uint32_t qm = (uint32_t)__builtin_amdgcn_readlane((int)myq, (int)j);
                const float4 ea = s.a[j], ed = s.d[j], ec = s.c[j];
                const uint32_t idx0 = (uint32_t)b * 64u + j;            // 0-based position in the tile list
                // per-lane partial sums of this tile instance.  Geometry enters through the five moments of
                // m = dL/dG * G:  S1 = sum m dx, S2 = sum m dy, Sxx = sum m dx^2, Sxy = sum m dx dy, Syy = sum m dy^2;
                // dL/dmean2D and dL/dconic are linear in them and are formed AFTER the wave reduction.
                // v: S1, S2, Sxx, Sxy, Syy, opacity, r, g, b [, depth]
                float v[10];
#pragma unroll
                for (int i = 0; i < 10; i++) v[i] = 0.f;
                bool any = false;
#ifdef W3D_BWD_STATS
                st_entries++;
#endif
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (!(qm & (1u << k))) continue;
#ifdef W3D_BWD_STATS
                    st_quads++;
#endif
                    const float dx = ea.x - (pxb + (float)((k & 1) * 8)), dy = ea.y - (pyb + (float)((k >> 1) * 8));
                    const float power = fmaf(ed.z * dy, dy, fmaf(ed.y, dy, ed.x * dx) * dx);      // log2 domain
                    // (ballots of plain compares, ANDed as scalars — see the forward kernel)
                    bool cand = power <= 0.f;
                    uint64_t mc = w3d_ballot(cand) & w3d_ballot(power >= ea.z);
                    cand = cand && power >= ea.z;
                    mc &= w3d_ballot(idx0 < last[k]);
                    cand = cand && idx0 < last[k];
                    if (mc == 0ull) continue;
#ifdef W3D_BWD_STATS
                    st_exp++;
#endif
                    const float Graw = __builtin_amdgcn_exp2f(power);
                    float araw = ed.w * Graw;
                    araw = fminf(0.99f, araw);
                    const uint64_t mk = mc & w3d_ballot(araw >= (1.0f / 255.0f));
                    any = any || (mk != 0ull);
#ifdef W3D_BWD_STATS
                    st_full += mk != 0ull ? 1u : 0u;
#endif
                    // Branch-free per lane: a lane that does not blend this Gaussian runs the same recurrences
                    // with alpha = G = 0, which leaves T and the suffix accumulators untouched and adds zeros.
                    const bool ok = cand && araw >= (1.0f / 255.0f);    // (its lane mask is mk, already in an SGPR pair)
#if W3D_VCC_SELECT
                    float alpha = araw, G = Graw;
                    (void)ok;
                    bwd_mask2(mk, alpha, G);
#else
                    const float alpha = ok ? araw : 0.f;
                    const float G = ok ? Graw : 0.f;
#endif
                    const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
                    const float Tn = Tr[k] * inv;
                    const float dch = alpha * Tn;
#if W3D_BWD_SFORM
                    float cdp = ec.x * dp0[k] + ec.y * dp1[k] + ec.z * dp2[k];
                    if (HAS_DA) { cdp += ec.w * dpd[k] + dpa[k]; v[9] += dch * dpd[k]; }
                    float dL_dalpha = cdp - S[k];
                    S[k] = fmaf(alpha, dL_dalpha, S[k]);
#else
                    const float d0 = ec.x - ar0[k], d1 = ec.y - ar1[k], d2 = ec.z - ar2[k];
                    float dL_dalpha = d0 * dp0[k] + d1 * dp1[k] + d2 * dp2[k];
                    ar0[k] += alpha * d0; ar1[k] += alpha * d1; ar2[k] += alpha * d2;   // A <- a c + (1-a) A
                    if (HAS_DA) {
                        const float dd = ec.w - ard[k], da = 1.f - ara[k];
                        dL_dalpha += dd * dpd[k] + da * dpa[k];
                        ard[k] += alpha * dd; ara[k] += alpha * da;
                        v[9] += dch * dpd[k];
                    }
#endif
                    Tr[k] = Tn;
                    dL_dalpha *= Tn;
                    if (BG) dL_dalpha -= (Tfin[k] * inv) * (bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k]);
                    const float m = ed.w * dL_dalpha * G;      // dL/dG * G
                    const float mx = m * dx, my = m * dy;
                    v[0] += mx; v[1] += my;
                    v[2] += mx * dx; v[3] += mx * dy; v[4] += my * dy;
                    v[5] += G * dL_dalpha;
                    v[6] += dch * dp0[k]; v[7] += dch * dp1[k]; v[8] += dch * dp2[k];
                }
                if (!any) continue;
                // Row sums by reduce-scatter.  A DPP "bank" is a group of four consecutive lanes, and a DPP add with a bank
                // mask writes only the enabled banks.  Stage 1 (partner = the neighbouring bank, lane +- 4) therefore halves the
                // register count while it adds: banks 0 and 2 take values 0..4, banks 1 and 3 values 5..9; stage 2 (lane +- 8)
                // halves it again: bank q of s0 / s1 / s2 ends up with value {0,5,1,6}[q] / {2,7,3,8}[q] / {4,9}[q], summed over
                // the four lanes of the row that share lane % 4; two quad-permute stages on those 3 registers finish the row
                // sums.  21 DPP adds per entry instead of 36.
                float r0, r1, r2, r3, r4, s0, s1, s2 = 0.f;
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                             "v_add_f32_dpp %1, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                             "v_add_f32_dpp %2, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                             "v_add_f32_dpp %3, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                             "v_add_f32_dpp %4, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                             "v_add_f32_dpp %0, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                             "v_add_f32_dpp %1, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                             "v_add_f32_dpp %2, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                             "v_add_f32_dpp %3, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                             "v_add_f32_dpp %4, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4)
                             : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]));
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %3, %3 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                             "v_add_f32_dpp %1, %5, %5 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                             "v_add_f32_dpp %2, %7, %7 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                             "v_add_f32_dpp %0, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                             "v_add_f32_dpp %1, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                             "s_nop 1"           // (the compiler's DPP reads of s0..s2 follow: it cannot see the hazard behind inline asm)
                             : "=&v"(s0), "=&v"(s1), "+v"(s2)
                             : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4));
                s0 += dpp_mov<0xB1>(s0); s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);      // quad_perm [1,0,3,2]
                s0 += dpp_mov<0x4E>(s0); s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);      // quad_perm [2,3,0,1]
                touched |= 1u << jl;
                if (storer) {
                    acc[offA + jl * 4u] = s0;
                    acc[offB + jl * 4u] = s1;
                    if (storeC) acc[offC + jl * 4u] = s2;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // flush: lane -> (entry e = 4*pass + lane/16, value k = lane%16) adds the four row sums; the five moments become
            // dL/dmean2D and dL/dconic here.  record: [0] dL/dmean2D.x [1] .y [2] dL/dconic.x [3] .y (half) [4] .z
            // [5] opacity [6..8] rgb [9] depth
            if (touched) {
                const uint32_t k = lane & 15u, sub = lane >> 4;
#pragma unroll 4
                for (uint32_t pass = 0; pass < 8; pass++) {
                    const uint32_t e = pass * 4u + sub;
                    if (((touched >> e) & 1u) && k < (uint32_t)NV) {
                        const float4 r4v = *reinterpret_cast<const float4 *>(&acc[k * W3D_ACC_PITCH + e * 4u]);
                        const float sk = (r4v.x + r4v.y) + (r4v.z + r4v.w);
                        const float so = dpp_mov<0xB1>(sk);            // k = 0 <-> k = 1 exchange their sums (S1, S2)
                        const float4 co = s.b[jlo + e];
                        float val;
                        if (k == 0u) val = -(co.x * sk + co.y * so) * ddelx_dx;
                        else if (k == 1u) val = -(co.z * sk + co.y * so) * ddely_dy;
                        else if (k < 5u) val = -0.5f * sk;
                        else val = sk;
                        if (DET) {
                            // deterministic mode: the contribution goes to the slot of its list entry (written once);
                            // det_gather_kernel adds a Gaussian's slots in tile order
                            const uint32_t pos = start + (uint32_t)b * 64u + (uint32_t)jlo + e;
                            if (pos < inst_cap) inst[(size_t)pos * W3D_G2D_STRIDE + k] = val;
                        } else {
                            const uint32_t g = __float_as_uint(s.a[jlo + e].w);
                            atomicAdd(&grad2d[(size_t)g * W3D_G2D_STRIDE + k], val);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    };
    if (has_bg) walk(std::true_type{}); else walk(std::false_type{});
#ifdef W3D_BWD_STATS
    if (lane == 0) {
        uint32_t *cw = const_cast<uint32_t *>(counters);
        atomicAdd(&cw[8], st_staged); atomicAdd(&cw[9], st_entries); atomicAdd(&cw[10], st_quads);
        atomicAdd(&cw[11], st_exp); atomicAdd(&cw[12], st_full); atomicAdd(&cw[13], 1u);
    }
#endif
}

// Zeroes the 64-B gradient records of the VISIBLE Gaussians (an all-zero rect marks a culled one, whose record nobody
// adds to or reads): 4 threads per record, whole lines.
__global__ void __launch_bounds__(256) zero_visible_records_kernel(float4 *__restrict__ rec, const uint2 *__restrict__ rect, size_t P) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * P) return;
    const uint2 r = rect[i >> 2];
    if (r.x | r.y) rec[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Deterministic mode, second half: one thread per visible Gaussian finds its entry in the list of every tile it was binned
// to (binary search on the list order: depth bits, then id) and adds the slots in ascending tile order.
__global__ void __launch_bounds__(256)
det_gather_kernel(int P, int gx, const uint2 *__restrict__ rect, const uint4 *__restrict__ rect_mask, int cull,
                  const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                  const float4 *__restrict__ grec, const uint32_t *__restrict__ counters, const float *__restrict__ inst,
                  uint32_t inst_cap, float *__restrict__ grad2d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    uint2 rc = rect[g];
    if (!(rc.x | rc.y)) return;
    // (tile_cull: the lists were built from the rect of the 16-B rect / mask record, which the preprocess may have shrunk to
    //  the footprint's own extent; the mask bits count the tiles of THAT rect)
    uint4 rm = make_uint4(0u, 0u, 0u, 0u);
    if (cull) { rm = rect_mask[g]; rc = make_uint2(rm.x, rm.y); }
    const uint32_t minx = rc.x & 0xFFFFu, miny = rc.x >> 16, maxx = rc.y & 0xFFFFu, maxy = rc.y >> 16;
    const uint32_t w = maxx - minx, nt = w * (maxy - miny);
    uint64_t mask = ~0ull;
    if (cull && nt <= 64u) mask = (uint64_t)rm.z | ((uint64_t)rm.w << 32);
    const uint32_t cap = min(counters[3], inst_cap);
    const uint32_t mykey = __float_as_uint(grec[4 * (size_t)g + 2].w);
    float sum[10];
#pragma unroll
    for (int k = 0; k < 10; k++) sum[k] = 0.f;
    for (uint32_t ty = miny; ty < maxy; ty++)
        for (uint32_t tx = minx; tx < maxx; tx++) {
            const uint32_t kbit = (ty - miny) * w + (tx - minx);
            if (nt <= 64u && !((mask >> kbit) & 1ull)) continue;
            const uint32_t t = ty * (uint32_t)gx + tx;
            uint32_t lo = min(tile_start[t], cap), hi = min(tile_start[t + 1], cap);
            while (lo < hi) {                        // first entry whose (depth bits, id) is not below mine
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t e = point_list[mid];
                const uint32_t ek = __float_as_uint(grec[4 * (size_t)e + 2].w);
                if (ek < mykey || (ek == mykey && e < (uint32_t)g)) lo = mid + 1; else hi = mid;
            }
            if (lo < min(tile_start[t + 1], cap) && point_list[lo] == (uint32_t)g) {
                const float4 *sl = reinterpret_cast<const float4 *>(inst + (size_t)lo * W3D_G2D_STRIDE);
                const float4 a = sl[0], bq = sl[1], c = sl[2];
                sum[0] += a.x; sum[1] += a.y; sum[2] += a.z; sum[3] += a.w;
                sum[4] += bq.x; sum[5] += bq.y; sum[6] += bq.z; sum[7] += bq.w;
                sum[8] += c.x; sum[9] += c.y;
            }
        }
    float4 *out = reinterpret_cast<float4 *>(grad2d + (size_t)g * W3D_G2D_STRIDE);
    out[0] = make_float4(sum[0], sum[1], sum[2], sum[3]);
    out[1] = make_float4(sum[4], sum[5], sum[6], sum[7]);
    out[2] = make_float4(sum[8], sum[9], 0.f, 0.f);
}

// Block -> tile map of the blend backward.  Its waves differ a lot in length (the reverse walk of a tile is as long as the
// forward got into its list), the chip holds only 4096 of the 7500 tile-waves at once, and whatever starts last finishes
// last: with tiles in image order the kernel ends on a tail of half-empty SIMDs.  So every XCD processes ITS OWN contiguous
// range of tiles (the L2 locality of the static map stays) longest walks first: a stable counting sort of the range by a
// 6-bit quantised walk length — neighbours of similar length stay neighbours.  One workgroup per XCD; frames with more than
// 1024 tiles per XCD keep the image order.
__global__ void __launch_bounds__(1024)
tile_order_kernel(const uint32_t *__restrict__ tile_walk, uint32_t T, uint32_t per_xcd, uint32_t *__restrict__ order) {
    __shared__ uint32_t hist[16][64];
    __shared__ uint32_t base_s[16][64];
    __shared__ uint32_t red[16];
    const uint32_t x = blockIdx.x, t0 = x * per_xcd, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (per_xcd > 1024u) {
        for (uint32_t i = threadIdx.x; i < per_xcd; i += 1024u) order[t0 + i] = t0 + i;
        return;
    }
    const uint32_t i = threadIdx.x;
    const bool valid = i < per_xcd && t0 + i < T;
    const uint32_t w = valid ? tile_walk[t0 + i] : 0u;
    uint32_t m = wave_max_u32(w);
    if (lane == 0) red[wv] = m;
    for (uint32_t b = lane; b < 64; b += 64) hist[wv][b] = 0;
    __syncthreads();
    uint32_t wmax = 1u;
#pragma unroll
    for (int k = 0; k < 16; k++) wmax = max(wmax, red[k]);
    // bucket 0 = longest; tiles past the end of the image (padding of the last XCD) go last
    const uint32_t q = valid ? 63u - min(63u, (uint32_t)(((uint64_t)w * 64u) / ((uint64_t)wmax + 1u))) : 63u;
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const uint64_t mb = w3d_ballot((q >> b) & 1u);
        peers &= ((q >> b) & 1u) ? mb : ~mb;
    }
    const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (rank == 0) hist[wv][q] = (uint32_t)__popcll(peers);
    __syncthreads();
    if (wv == 0) {
        // lane = bucket: totals over the 16 waves, exclusive scan over the buckets, then the per-wave bases
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) tot += hist[k][lane];
        uint32_t incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)lane >= off) incl += up;
        }
        uint32_t run = incl - tot;
#pragma unroll
        for (int k = 0; k < 16; k++) { base_s[k][lane] = run; run += hist[k][lane]; }
    }
    __syncthreads();
    if (i < per_xcd) order[t0 + base_s[wv][q] + rank] = t0 + i;
}

// The block -> (tile, part) schedule of a blend kernel for frames of up to W3D_SCHED_MAX_TILES tiles (one workgroup; the kernel above
// serves larger frames).  The one-wave-per-tile schedule loses twice (profiles/r06/walk_schedule.jsonl: makespan over the perfectly
// divisible bound 1.08-1.34 on the translucent scenes, 1.5-2.3 on the opaque one):
//   * every XCD gets the same NUMBER of tiles, not the same work: here the 8 contiguous tile ranges are cut at equal cumulated cost
//     (cost of a tile = its walk length + W3D_SCHED_C0 for the wave's fixed part), so they differ in tile count — the grid has
//     `cap` = 1.5x the even share of blocks per XCD and the blocks past a range's entries exit at once;
//   * a tile whose wave alone takes longer than the whole chip needs per wave slot (cost > total / slots) finishes last whatever
//     the order: such a tile is cut into two waves (upper / lower pair of 8x8 quadrants, each ~0.6 of the cost: the staging is paid
//     twice) or, beyond 2.2x, four (one quadrant each, ~0.35).  Part-waves add into the same Gaussian records (atomic backward
//     only: allow_split = 0 for the deterministic one) and write disjoint pixels.
// Inside a range: longest entry first (counting sort by a 6-bit quantised cost).  If a cost-balanced range would need more than `cap`
// entries (a heavy-tailed profile) even ranges are used with the long tiles still split, and if that does not fit either, even ranges
// of whole tiles.  Frames of up to 1024 tiles have cap = 4x the even share: every tile may run as four quadrant waves.
#define W3D_SCHED_C0 16u
#ifndef W3D_QUARTER_X10
#define W3D_QUARTER_X10 22u    // ... and in four beyond this / 10 x the halving threshold
#endif
#ifndef W3D_SPLIT_X10
#define W3D_SPLIT_X10 15u      // a tile is cut in two when its cost exceeds this / 10 x (total cost / wave slots), in four beyond 2.2x that
#endif
// cost of a tile's wave, its part code (0 whole, 1 two halves, 3 four quadrants) and the cost of one of its entries
__device__ __forceinline__ uint32_t sched_cost(uint32_t walk) { return walk ? walk + W3D_SCHED_C0 : 2u; }
__device__ __forceinline__ uint32_t sched_code(uint32_t c, uint32_t bound, int allow_split, uint32_t &ce) {
    ce = c;
    if (!allow_split || 10u * c <= W3D_SPLIT_X10 * bound) return 0u;
    if (100u * c > W3D_QUARTER_X10 * W3D_SPLIT_X10 * bound) { ce = (35u * c) / 100u; return 3u; }
    ce = (6u * c) / 10u;
    return 1u;
}
// Eight workgroups, one per range.  Every one of them reads ALL walk lengths (30 KB) and scans cost and entry count in tile order —
// so that all eight agree on the cuts and on whether the balanced ranges fit `cap` — then orders the entries of its own range.
__global__ void __launch_bounds__(1024)
tile_schedule_kernel(const uint32_t *__restrict__ walk, uint32_t T, uint32_t cap, uint32_t slots, int allow_split,
                     uint32_t *__restrict__ order) {
    constexpr uint32_t PER = W3D_SCHED_MAX_TILES / 1024u;
    __shared__ uint32_t wave_c[16], wave_e[16], last_x[1025];
    __shared__ uint32_t cut_t[9], cut_e[9];          // first tile of range k and the entries in front of it (k = 8: T, all entries)
    __shared__ uint32_t cnt[64], base[64], s_cmax[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, x = blockIdx.x;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t w[PER];
    {
        const uint4 *w4 = reinterpret_cast<const uint4 *>(walk);
        const uint32_t t0 = tid * PER;
        if (t0 + PER <= T) {
            const uint4 a = w4[2 * tid], b4 = w4[2 * tid + 1];
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b4.x; w[5] = b4.y; w[6] = b4.z; w[7] = b4.w;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < PER; i++) w[i] = t0 + i < T ? walk[t0 + i] : 0u;
        }
    }
    if (tid < 9) { cut_t[tid] = T; cut_e[tid] = 0u; }
    // ---- pass 1: total cost -> the bound a single wave should stay under -> part codes
    uint32_t c[PER], csum = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) { const uint32_t t = tid * PER + i; c[i] = t < T ? sched_cost(w[i]) : 0u; csum += c[i]; }
    uint32_t cinc = csum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)cinc, off, 64); if ((int)lane >= off) cinc += u; }
    if (lane == 63) wave_c[wv] = cinc;
    __syncthreads();
    uint32_t cpre = cinc - csum, Wtot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) { const uint32_t v = wave_c[k]; cpre += k < wv ? v : 0u; Wtot += v; }
    const uint32_t bound = max(Wtot / max(slots, 1u), 1u);
    const float range_scale = 8.0f / (float)max(Wtot, 1u);
    // ---- pass 2: entries in tile order, range of every tile (monotone in tile order), the cuts.  Up to three attempts, each decided
    // identically by all eight workgroups: cost-balanced ranges with long tiles split; if some range would need more than `cap` entries
    // (a heavy-tailed profile: the ranges without a long tile hold too many short ones), EVEN ranges with long tiles split — the
    // splits are what such a profile needs most; if that does not fit either, even ranges of whole tiles (always fits).
    bool bad = false;
    uint32_t my_s = 0, my_e = 0, my_n = 0;
    int split = allow_split;
    const uint32_t even = (T + 7u) / 8u;
    for (int attempt = 0; attempt < 3; attempt++) {
        const bool balanced = attempt == 0;
        if (attempt == 2) split = 0;
        if (tid < 9) { cut_t[tid] = T; cut_e[tid] = 0u; }
        uint32_t esum = 0, xl = 0, run = cpre;
        uint32_t xr[PER], ne[PER];
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) {
            const uint32_t t = tid * PER + i;
            uint32_t ce;
            const uint32_t code = sched_code(c[i], bound, split, ce);
            ne[i] = t < T ? (code == 0u ? 1u : (code == 1u ? 2u : 4u)) : 0u;
            esum += ne[i];
            xr[i] = balanced ? min(7u, (uint32_t)((float)(run + c[i] / 2u) * range_scale)) : min(7u, t / even);
            run += c[i];
            if (t < T) xl = xr[i];
        }
        uint32_t einc = esum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)einc, off, 64); if ((int)lane >= off) einc += u; }
        if (lane == 63) wave_e[wv] = einc;
        last_x[tid + 1] = xl;               // (threads behind the last tile repeat the last range)
        if (tid == 0) last_x[0] = 0u;
        __syncthreads();
        uint32_t epre = einc - esum, Etot = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) { const uint32_t v = wave_e[k]; epre += k < wv ? v : 0u; Etot += v; }
        {
            uint32_t prev = last_x[tid], er = epre;
#pragma unroll
            for (uint32_t i = 0; i < PER; i++) {
                const uint32_t t = tid * PER + i;
                if (t < T) {
                    for (uint32_t k = prev + 1u; k <= xr[i]; k++) { cut_t[k] = t; cut_e[k] = er; }     // ranges (prev, xr] start at this tile
                    prev = xr[i];
                }
                er += ne[i];
            }
            if (tid == 0) { cut_t[0] = 0u; cut_e[0] = 0u; cut_e[8] = Etot; }
        }
        __syncthreads();
        // (a range nobody starts keeps cut_t = T: empty — and every later one as well, since the ranges are monotone; its cut_e must
        //  then be the total)
        bad = false;
        {
            uint32_t ct[9], cee[9];
#pragma unroll
            for (int k = 0; k < 9; k++) { ct[k] = cut_t[k]; cee[k] = cut_t[k] >= T ? Etot : cut_e[k]; }
#pragma unroll
            for (int k = 0; k < 8; k++) bad = bad || (cee[k + 1] - cee[k]) > cap;
            my_s = ct[0]; my_e = ct[1]; my_n = cee[1] - cee[0];
#pragma unroll
            for (int k = 1; k < 8; k++) if (x == (uint32_t)k) { my_s = ct[k]; my_e = ct[k + 1]; my_n = cee[k + 1] - cee[k]; }
        }
        __syncthreads();               // (the cuts are rewritten by the next attempt)
        if (!bad) break;               // (uniform: the same data in all threads of all eight workgroups)
        if (!split) attempt = 1;       // (nothing to split: the next attempt is the last one)
    }
    // ---- pass 3: this range's entries, longest first
    if (tid < 64) cnt[tid] = 0u;
    uint32_t cm = 1u;
    for (uint32_t t = my_s + tid; t < my_e; t += 1024u) { uint32_t ce; sched_code(sched_cost(walk[t]), bound, split, ce); cm = max(cm, ce); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cm = max(cm, (uint32_t)__shfl_xor((int)cm, off, 64));
    if (lane == 0) s_cmax[wv] = cm;
    __syncthreads();
    uint32_t cmax = 1u;
#pragma unroll
    for (int k = 0; k < 16; k++) cmax = max(cmax, s_cmax[k]);
    const float q_scale = 64.0f / (float)(cmax + 1u);
    // (at most two rounds: a range holds at most cap <= 1.5 x 1024 + 1 entries)  The entries of one bucket among the 64 tiles of a
    // wave's round are found by ballots and ONE lane adds their number: neighbouring tiles mostly share a bucket, and 64 LDS atomics
    // on one address serialise.
    constexpr int ROUNDS = 2;
    uint32_t rq[ROUNDS], rcode[ROUNDS], rbefore[ROUNDS], rtot[ROUNDS];
    uint64_t rgrp[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t t = my_s + (uint32_t)r * 1024u + tid;
        const bool valid = t < my_e;
        uint32_t ce = 0u;
        rcode[r] = valid ? sched_code(sched_cost(walk[t]), bound, split, ce) : 0u;
        rq[r] = 63u - min(63u, (uint32_t)((float)ce * q_scale));                 // bucket 0 = longest
        uint64_t peers = w3d_ballot(valid);
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const uint64_t m = w3d_ballot((rq[r] >> b) & 1u);
            peers &= ((rq[r] >> b) & 1u) ? m : ~m;
        }
        const uint64_t m2 = w3d_ballot(rcode[r] == 1u), m4 = w3d_ballot(rcode[r] == 3u);
        auto entries = [&](uint64_t set) -> uint32_t { return (uint32_t)__popcll(set) + (uint32_t)__popcll(set & m2) + 3u * (uint32_t)__popcll(set & m4); };
        rgrp[r] = valid ? peers : 0ull;
        rbefore[r] = entries(peers & lt);
        rtot[r] = entries(peers);
        if (valid && (peers & lt) == 0ull) atomicAdd(&cnt[rq[r]], rtot[r]);
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t v = cnt[lane];
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, off, 64); if ((int)lane >= off) inc += u; }
        base[lane] = inc - v;
        cnt[lane] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t t = my_s + (uint32_t)r * 1024u + tid;
        const uint64_t peers = rgrp[r];
        uint32_t old = 0u;
        if (peers && (peers & lt) == 0ull) old = atomicAdd(&cnt[rq[r]], rtot[r]);       // the group's leader claims its entries
        old = (uint32_t)__shfl((int)old, peers ? (int)__builtin_ctzll(peers) : 0, 64);
        if (peers) {
            const uint32_t n = rcode[r] == 0u ? 1u : (rcode[r] == 1u ? 2u : 4u);
            const uint32_t pos = base[rq[r]] + old + rbefore[r];
            for (uint32_t e = 0; e < n; e++) order[x * cap + pos + e] = t | ((rcode[r] + e) << 29);
        }
    }
    for (uint32_t j = my_n + tid; j < cap; j += 1024u) order[x * cap + j] = 0xFFFFFFFFu;
}

__global__ void copy_pixel_state_kernel(const float *__restrict__ fT, const uint32_t *__restrict__ nc, size_t n,
                                        float *__restrict__ fT_out, uint32_t *__restrict__ nc_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (fT_out) fT_out[i] = fT[i];
        if (nc_out) nc_out[i] = nc[i];
    }
}

}  // namespace

int w3d_launch_render(const W3DLayout &L, const w3d_view &v, char *state, const uint32_t *point_list, uint64_t list_capacity,
                      float *out_color, float *out_depth, float *out_alpha, const float *gt_mask, int32_t num_obj,
                      float *used_count, int32_t *contrib_num, hipStream_t stream) {
    const uint32_t T = (uint32_t)L.T;
    uint32_t blocks = (T + W3D_RW - 1) / W3D_RW;
    blocks = (blocks + 7) / 8 * 8;   // the XCD-contiguous map needs a multiple of 8 blocks
    const bool flash = (gt_mask != nullptr) || (used_count != nullptr) || (contrib_num != nullptr);
    const uint32_t *ts = reinterpret_cast<const uint32_t *>(state + L.o_tile_start);
    // The forward does not know its walk lengths yet (list lengths are no stand-in: measured, no gain — most walks stop
    // early); a caller that renders the same camera repeatedly hands in the lengths of its previous render (w3d.h)
    uint32_t *order = nullptr;
    const uint32_t cap = L.order_cap;
#if W3D_TILE_ORDER
    if (W3D_RW == 1 && v.tile_walk_hint) {
        order = reinterpret_cast<uint32_t *>(state + L.o_tile_order);
        // (slots: 1024 SIMDs x the waves per SIMD the kernel is compiled for)
        if (T <= W3D_SCHED_MAX_TILES)
            hipLaunchKernelGGL(tile_schedule_kernel, dim3(8), dim3(1024), 0, stream, (const uint32_t *)v.tile_walk_hint, T, cap,
                               1024u * (flash ? 4u : (uint32_t)W3D_FWD_OCC), 1, order);
        else
            hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(1024), 0, stream, (const uint32_t *)v.tile_walk_hint, T, cap, order);
        blocks = 8 * cap;
    }
#endif
#define ARGS                                                                                                          \
    T, (uint32_t)L.gx, L.W, L.H, ts, point_list,                                                                      \
        reinterpret_cast<const float4 *>(state + L.o_grec), v.bg, out_color, out_depth, out_alpha,                    \
        reinterpret_cast<float *>(state + L.o_final_T), reinterpret_cast<uint32_t *>(state + L.o_n_contrib), gt_mask, \
        num_obj, L.P, used_count, contrib_num, (uint32_t)(list_capacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : list_capacity),      \
        reinterpret_cast<uint32_t *>(state + L.o_counters), reinterpret_cast<uint32_t *>(state + L.o_tile_walk), order, cap,   \
        v.tile_walk_hint, (uint32_t)L.lsx | ((uint32_t)L.lsy << 4), (uint32_t)L.lgx
    {
        W3D_PROF("render_fwd", stream);
        if (flash) hipLaunchKernelGGL((render_fwd_kernel<true>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
        else if (out_depth && out_alpha) hipLaunchKernelGGL((render_fwd_kernel<false>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
        else hipLaunchKernelGGL((render_fwd_kernel<false, false>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
    }
#undef ARGS
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

int w3d_launch_render_backward(const W3DLayout &L, const w3d_view &v, const char *state, const uint32_t *point_list,
                               const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha, float *grad2d,
                               hipStream_t stream) {
    const uint32_t T = (uint32_t)L.T;
    uint32_t blocks = (T + W3D_RW - 1) / W3D_RW;
    blocks = (blocks + 7) / 8 * 8;
    const bool det = v.deterministic != 0;
    // block -> (tile, part) map from the walk lengths the forward of this view measured (tile_schedule_kernel)
    uint32_t *order = nullptr;
    const uint32_t cap = L.order_cap;
#if W3D_TILE_ORDER
    if (W3D_RW == 1) {
        order = reinterpret_cast<uint32_t *>(const_cast<char *>(state) + L.o_tile_order);
        uint32_t *walk = reinterpret_cast<uint32_t *>(const_cast<char *>(state) + L.o_tile_walk);
        const bool da_ = dL_ddepth || dL_dalpha;
        if (T <= W3D_SCHED_MAX_TILES)
            hipLaunchKernelGGL(tile_schedule_kernel, dim3(8), dim3(1024), 0, stream, (const uint32_t *)walk, T, cap,
                               1024u * (da_ ? 3u : (uint32_t)W3D_BWD_OCC), det ? 0 : 1, order);
        else
            hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(1024), 0, stream, (const uint32_t *)walk, T, cap, order);
        blocks = 8 * cap;
    }
#endif
    // deterministic mode: [P records][det_list_capacity slots] in the scratch buffer (w3d_backward_det_sizes)
    float *inst = nullptr;
    uint32_t inst_cap = 0;
    static_assert(W3D_G2D_STRIDE == 16, "4 float4 per record");
    const size_t Pz = (size_t)(L.P > 0 ? L.P : 0);
    if (det) {
        const uint64_t cap = v.det_list_capacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : v.det_list_capacity;
        inst = reinterpret_cast<float *>(reinterpret_cast<char *>(grad2d) + w3d_align_up((uint64_t)(Pz ? Pz : 1) * W3D_G2D_STRIDE * sizeof(float)));
        inst_cap = (uint32_t)cap;
        // entries the reverse walk never reaches contribute nothing: their slots stay zero
        if (cap) W3D_HIP_CHECK(hipMemsetAsync(inst, 0, cap * W3D_G2D_STRIDE * sizeof(float), stream));
    } else if (Pz && !v.records_kept_clean) {
        // the records the atomics add to start at zero (own kernel rather than hipMemsetAsync: strictly stream-ordered);
        // records_kept_clean: the caller's buffer is zero already and preprocess_bwd_kernel leaves it so (w3d.h)
        hipLaunchKernelGGL(zero_visible_records_kernel, dim3((unsigned)((4 * Pz + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<float4 *>(grad2d), reinterpret_cast<const uint2 *>(state + L.o_rect), Pz);
    }
#define ARGS                                                                                                      \
    T, (uint32_t)L.gx, L.W, L.H, reinterpret_cast<const uint32_t *>(state + L.o_tile_start), point_list,          \
        reinterpret_cast<const float4 *>(state + L.o_grec), v.bg,                                                 \
        reinterpret_cast<const float *>(state + L.o_final_T), reinterpret_cast<const uint32_t *>(state + L.o_n_contrib), \
        dL_dcolor, dL_ddepth, dL_dalpha, grad2d, reinterpret_cast<const uint32_t *>(state + L.o_counters), inst, inst_cap, order, cap, \
        (uint32_t)L.lsx | ((uint32_t)L.lsy << 4), (uint32_t)L.lgx
    {
        W3D_PROF("render_bwd", stream);
        const bool da = dL_ddepth || dL_dalpha;
        if (det) {
            if (da) hipLaunchKernelGGL((render_bwd_kernel<true, true>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
            else hipLaunchKernelGGL((render_bwd_kernel<false, true>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
        } else {
            if (da) hipLaunchKernelGGL((render_bwd_kernel<true, false>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
            else hipLaunchKernelGGL((render_bwd_kernel<false, false>), dim3(blocks), dim3(64 * W3D_RW), 0, stream, ARGS);
        }
    }
    if (det && Pz) {
        W3D_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(det_gather_kernel, dim3((unsigned)((Pz + 255) / 256)), dim3(256), 0, stream, (int)Pz, (int)L.gx,
                           reinterpret_cast<const uint2 *>(state + L.o_rect), reinterpret_cast<const uint4 *>(state + L.o_tile_mask),
                           (int)v.tile_cull, reinterpret_cast<const uint32_t *>(state + L.o_tile_start), point_list,
                           reinterpret_cast<const float4 *>(state + L.o_grec),
                           reinterpret_cast<const uint32_t *>(state + L.o_counters), inst, inst_cap, grad2d);
    }
#undef ARGS
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

int w3d_debug_pixel_state_impl(const W3DLayout &L, const char *state, float *final_T_out, uint32_t *n_contrib_out,
                               hipStream_t stream) {
    const size_t n = (size_t)L.H * L.W;
    hipLaunchKernelGGL(copy_pixel_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float *>(state + L.o_final_T),
                       reinterpret_cast<const uint32_t *>(state + L.o_n_contrib), n, final_T_out, n_contrib_out);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
