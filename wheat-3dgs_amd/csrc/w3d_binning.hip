// w3d_binning.hip — depth sort of the Gaussians and construction of the per-tile lists
// (replaces the scan / duplicate-with-keys / 64-bit radix sort /
// identify-ranges stage of the reference's CUDA submodule).
//
// MI355X-first formulation.  Instead of materialising R = sum(tiles touched) 64-bit
// (tile|depth) keys and radix-sorting all of them through HBM, the order contract
// "within a tile ascending depth bits, ties by ascending Gaussian index" is met in two steps:
//   1. ONE stable LSD radix sort of the (depth bits, index) pairs — P << R, 16 B per Gaussian; its first pass drops the
//      culled Gaussians and publishes V, the later passes sort the V survivors;
//   2. a single-pass stable counting sort of the tile instances by tile id: the depth order is cut into C chunks and the
//      image into bands of tile rows; one wave per (chunk, band) keeps the band's per-tile counters (count pass) or
//      cursors + 64-bit rank bitmaps (fill pass) in LDS and bins its records 64 at a time, one record per lane —
//      the rank of a record among the batch's hits of a tile is a popcount of the tile's bitmap, so every list slot is
//      written exactly once, already in its final position, without any serial walk (chunk_walk_kernel).
// HBM traffic: 4 B per tile instance (the list itself) + the C x T counter matrices, instead
// of >= 6 passes x 24 B per instance.  No global atomics on the instance path.
//
// Everything here is wave-synchronous (wave64): the LDS arrays of a wave are private to it,
// so there is no __syncthreads() in the hot loops.
#include "w3d_common.h"

namespace {

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// ------------------------------------------------------------------------------ radix sort
// One pass = histogram, row scan, scatter.  A "run" is the contiguous slice of keys one wave owns.
//
// The keys are the bit patterns of positive floats (view depths), and a view's depths rarely span more than two octaves
// (the wheat-plot cameras hover 2-2.5 units above the canopy: depths 1.7-3.2), so their upper 16 bits take at most 256
// consecutive values.  The first pass's histogram kernel therefore also finds min / max of the visible keys, and when
// (kmax >> 16) - (kmin >> 16) <= 255 the third pass sorts by (key >> 16) - (kmin >> 16) — ALL the remaining bits in one
// 8-bit digit — and the fourth pass does not run: 3 passes instead of 4, same order (the digit is a monotone function of the
// upper half).  Wider depth ranges take the four plain 8-bit passes.  The decision lives on the device (counters[4..5]):
// kernels of a pass that is not needed exit at once, and the consumers pick the buffer the last executed pass wrote.
#define W3D_CTL_KMIN_HI 4       // counters[4] = kmin >> 16
#define W3D_CTL_THREE 5         // counters[5] = 1: three passes suffice
__device__ __forceinline__ uint32_t radix_digit(uint32_t key, int pass, const uint32_t *__restrict__ ctl) {
    if (pass < 2) return (key >> (W3D_RADIX_BITS * pass)) & (W3D_RADIX_BINS - 1u);
    if (ctl[W3D_CTL_THREE]) return (key >> 16) - ctl[W3D_CTL_KMIN_HI];
    return (key >> (W3D_RADIX_BITS * pass)) & (W3D_RADIX_BINS - 1u);
}

__global__ void __launch_bounds__(256)
radix_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t items, uint32_t n_runs, int pass,
                  uint32_t *__restrict__ hist /* [256][n_runs] */, const uint32_t *__restrict__ n_dev, int drop_invalid,
                  const uint32_t *__restrict__ ctl, uint32_t *__restrict__ minmax /* pass 0: [n_runs][2] */) {
    __shared__ uint32_t h_all[4][W3D_RADIX_BINS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t run = blockIdx.x * 4 + wv;
    if (pass == 3 && ctl[W3D_CTL_THREE]) return;
    uint32_t *h = h_all[wv];
    for (int i = lane; i < W3D_RADIX_BINS; i += 64) h[i] = 0;
    __builtin_amdgcn_wave_barrier();
    // passes after the first sort only the visible Gaussians (the first pass dropped the culled ones and counted the
    // rest): n comes from the device and the runs re-partition it evenly
    if (n_dev) { n = *n_dev; items = max(64u, ((n + n_runs - 1u) / n_runs + 63u) & ~63u); }
    if (run < n_runs) {
        const uint32_t beg = min(n, run * items), end = min(n, beg + items);
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        // (the digit rule of passes 2 / 3 is wave-uniform and loop-invariant: read once)
        const uint32_t three = pass >= 2 ? ctl[W3D_CTL_THREE] : 0u, kmin_hi = pass >= 2 ? ctl[W3D_CTL_KMIN_HI] : 0u;
        const int shift = W3D_RADIX_BITS * pass;
#pragma unroll 8
        for (uint32_t i = beg + lane; i < end; i += 64) {
            const uint32_t key = keys[i];
            const uint32_t d = three ? (key >> 16) - kmin_hi : (key >> shift) & (W3D_RADIX_BINS - 1u);
            if (!drop_invalid || key != W3D_INVALID_KEY) {
                atomicAdd(const_cast<uint32_t *>(&h_all[wv][d & (W3D_RADIX_BINS - 1u)]), 1u);
                kmin = min(kmin, key); kmax = max(kmax, key);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < W3D_RADIX_BINS; i += 64) hist[(size_t)i * n_runs + run] = h[i];
        if (minmax) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
                kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
            }
            if (lane == 0) { minmax[2 * run] = kmin; minmax[2 * run + 1] = kmax; }
        }
    }
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// exclusive scan over the threads of a block (<= 1024 threads); returns the block total in `total`
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot /* LDS [17] */, uint32_t &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        const uint32_t t = lane < nw ? wave_tot[lane] : 0u;
        const uint32_t ti = wave_inclusive_scan(t);
        if (lane < nw) wave_tot[lane] = ti - t;
        if (lane == nw - 1) wave_tot[16] = ti;
    }
    __syncthreads();
    const uint32_t res = wave_tot[wv] + inc - v;
    total = wave_tot[16];
    __syncthreads();
    return res;
}

// one block per digit: exclusive scan of that digit's row of per-run counts (in place) + row total
__global__ void __launch_bounds__(256)
radix_rowscan_kernel(uint32_t *__restrict__ hist, uint32_t n_runs, uint32_t *__restrict__ rowtot, int pass,
                     uint32_t *__restrict__ ctl, const uint32_t *__restrict__ minmax) {
    __shared__ uint32_t wave_tot[17];
    if (pass == 3 && ctl[W3D_CTL_THREE]) return;
    if (pass == 0 && blockIdx.x == 0) {
        // depth range of the visible Gaussians -> how many passes the sort needs (see radix_digit)
        __shared__ uint32_t red[8];
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        for (uint32_t i = threadIdx.x; i < n_runs; i += 256) { kmin = min(kmin, minmax[2 * i]); kmax = max(kmax, minmax[2 * i + 1]); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = kmin; red[2 * (threadIdx.x >> 6) + 1] = kmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++) { kmin = min(kmin, red[2 * w]); kmax = max(kmax, red[2 * w + 1]); }
            const bool none = kmin > kmax;                       // nothing visible
            ctl[W3D_CTL_KMIN_HI] = none ? 0u : (kmin >> 16);
            ctl[W3D_CTL_THREE] = (none || ((kmax >> 16) - (kmin >> 16)) < (uint32_t)W3D_RADIX_BINS) ? 1u : 0u;
        }
        __syncthreads();
    }
    uint32_t *row = hist + (size_t)blockIdx.x * n_runs;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_runs; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_runs ? row[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, wave_tot, tot);
        if (i < n_runs) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}

// CAN_FINAL (passes 2 and 3): the pass that turns out to be the LAST one — the third when three suffice, else the fourth —
// does not write (key, id) pairs any more but the depth-ordered packed records {id, rect lo, rect hi, depth} + tile mask the
// (chunk, band) walkers stream (they re-read their chunk once per band and per pass: coalesced records instead of gathers).
// The 16-B rect / mask line of every Gaussian is gathered one batch ahead of the ranking, the (key, id) pairs two ahead.
template <bool CAN_FINAL>
__global__ void __launch_bounds__(256)
radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                     uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n, uint32_t items,
                     uint32_t n_runs, int pass, const uint32_t *__restrict__ offs /* row-scanned [256][n_runs] */,
                     const uint32_t *__restrict__ rowtot /* [BINS] */, uint32_t *__restrict__ num_visible,
                     const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ ctl,
                     const uint2 *__restrict__ rect, const uint4 *__restrict__ rect_mask, uint4 *__restrict__ rec,
                     uint2 *__restrict__ rec_mask, int cull) {
    __shared__ uint32_t cur_all[4][W3D_RADIX_BINS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t run = blockIdx.x * 4 + wv;
    if (run >= n_runs) return;
    if (pass == 3 && ctl[W3D_CTL_THREE]) return;
    const uint32_t three = pass >= 2 ? ctl[W3D_CTL_THREE] : 0u, kmin_hi = pass >= 2 ? ctl[W3D_CTL_KMIN_HI] : 0u;
    const bool fin = CAN_FINAL && (pass == 3 || three != 0u);
    const int shift = W3D_RADIX_BITS * pass;
    // num_visible != NULL: FIRST pass — culled Gaussians (key 0xFFFFFFFF) are dropped and the number of survivors is
    // published; n_dev != NULL: later pass over those survivors only (same re-partition as radix_hist_kernel)
    const bool drop_invalid = num_visible != nullptr;
    if (n_dev) { n = *n_dev; items = max(64u, ((n + n_runs - 1u) / n_runs + 63u) & ~63u); }
    uint32_t *cur = cur_all[wv];
    {
        // digit bases = exclusive scan of the row totals, BINS/64 consecutive digits per lane
        constexpr int PER = W3D_RADIX_BINS / 64;
        uint32_t tot[PER];
        uint32_t lsum = 0;
#pragma unroll
        for (int i = 0; i < PER; i += 4) {
            const uint4 t4 = reinterpret_cast<const uint4 *>(rowtot)[(lane * PER + i) / 4];
            tot[i] = t4.x; tot[i + 1] = t4.y; tot[i + 2] = t4.z; tot[i + 3] = t4.w;
            lsum += t4.x + t4.y + t4.z + t4.w;
        }
        const uint32_t incl = wave_inclusive_scan(lsum);
        uint32_t base = incl - lsum;
        if (num_visible && run == 0 && lane == 63) *num_visible = incl;      // all counted keys = the visible Gaussians
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t d = lane * PER + i;
            cur[d] = base + offs[(size_t)d * n_runs + run];
            base += tot[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t beg = min(n, run * items), end = min(n, beg + items);
    const uint64_t lt = lanemask_lt();
    auto gather = [&](uint32_t g) -> uint4 {
        if (cull) return rect_mask[g];                   // one 16-B record per Gaussian: a single random line
        const uint2 rc = rect[g];
        return make_uint4(rc.x, rc.y, 0xFFFFFFFFu, 0xFFFFFFFFu);
    };
    // software pipeline: the (key, id) pairs of the next two batches and the rect line of the next batch are in flight while
    // the current batch is ranked
    // (first pass: the value of key i is i — the preprocess writes the Gaussians in index order — so no value array is read)
    // pass 0 (drop_invalid) reads keys in Gaussian order: the value of key i IS i — the preprocess writes no id array, and
    // vals_in must not be read in that pass
    auto val_at = [&](uint32_t i) -> uint32_t { return drop_invalid ? i : vals_in[i]; };
    uint32_t k1 = (beg + lane < end) ? keys_in[beg + lane] : 0u, v1 = (beg + lane < end) ? val_at(beg + lane) : 0u;
    uint32_t k2 = (beg + 64 + lane < end) ? keys_in[beg + 64 + lane] : 0u, v2 = (beg + 64 + lane < end) ? val_at(beg + 64 + lane) : 0u;
    uint4 g1 = make_uint4(0u, 0u, 0u, 0u);
    if (CAN_FINAL && fin && beg + lane < end) g1 = gather(v1);
    for (uint32_t base = beg; base < end; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t key = k1, val = v1;
        const uint4 gg = g1;
        const bool valid = i < end && !(drop_invalid && key == W3D_INVALID_KEY);
        k1 = k2; v1 = v2;
        k2 = 0u; v2 = 0u;
        if (i + 128 < end) { k2 = keys_in[i + 128]; v2 = val_at(i + 128); }
        if (CAN_FINAL && fin && i + 64 < end) g1 = gather(v1);
        const uint32_t d = (three ? (key >> 16) - kmin_hi : (key >> shift)) & (W3D_RADIX_BINS - 1u);
        // lanes holding the same digit (stable rank = number of such lanes below me)
        uint64_t peers = w3d_ballot(valid);
#pragma unroll
        for (int b = 0; b < W3D_RADIX_BITS; b++) {
            const uint64_t m = w3d_ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt);
        uint32_t pos = 0;
        if (valid) pos = cur[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) cur[d] = pos + (uint32_t)__popcll(peers);   // group leader advances the cursor
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            if (CAN_FINAL && fin) {
                rec[pos] = make_uint4(val, gg.x, gg.y, key);          // (the sort key IS the view depth)
                rec_mask[pos] = make_uint2(gg.z, gg.w);
            } else {
                keys_out[pos] = key; vals_out[pos] = val;
            }
        }
    }
}

// ------------------------------------------------------------------------------ tile counting
// One wave per (chunk, band of tile rows).  The wave streams the chunk's depth-ordered records, keeps the
// ones whose rect reaches into its band (about one in eight) in a 128-entry LDS ring, and whenever 64 are
// queued it bins all 64 AT ONCE, one record per lane:
//   count (MODE 0): every lane walks the set bits of ITS record's tile mask and adds 1 to the tile's packed
//                   u16 LDS counter (ds_add_u32 — counting needs no order);
//   fill  (MODE 1): phase A: every lane ORs its lane bit into a 64-bit LDS bitmap per tile (ds_or_b64);
//                   phase B: it walks its tiles again, position = cursor[tile] + popcount(bitmap below my bit),
//                            and writes the list entry — ring slots are in depth order, so the rank within the
//                            batch IS the depth rank and the counting sort stays stable without any serial walk;
//                   phase C: lanes <-> tiles of the band: cursor += popcount(bitmap), bitmap = 0.
// Records with more than W3D_WALK_SMALL tiles in the band are binned by the whole wave (lanes <-> the tiles
// of that record) in the same three phases, so one big footprint does not stall 63 lanes.
#ifndef W3D_WALK_SMALL
#define W3D_WALK_SMALL 16
#endif
#define W3D_WALK_QUEUE 128

#ifndef W3D_WW
#define W3D_WW 4          // (chunk, band) waves per workgroup of the walk: the 4 band-waves of a chunk share its records in L1
                          // (measured fill: 1 wave 0.255 ms, 2 -> 0.198, 4 -> 0.172, 8 -> 0.202, 16 -> 0.234)
#endif
template <int MODE, bool CULL>
__global__ void __launch_bounds__(64 * W3D_WW)
chunk_walk_kernel(const uint4 *__restrict__ rec, const uint2 *__restrict__ rec_mask,
                  const uint32_t *__restrict__ counters,
                  uint32_t chunk, uint32_t C, uint32_t T, uint32_t gx, uint32_t gy, uint32_t band_rows,
                  uint16_t *__restrict__ cnt, const uint32_t *__restrict__ off, uint32_t *__restrict__ point_list,
                  uint64_t capacity, uint32_t wave_bytes) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the 4 waves of a workgroup walk the SAME chunk for 4 neighbouring bands: they stream the same records at
    // about the same time, so three of the four reads hit the CU's vector L1
    const uint32_t c = blockIdx.x;
    const uint32_t band = blockIdx.y * W3D_WW + wv;
    if (c >= C || band * band_rows >= gy) return;
    const uint32_t y0 = band * band_rows, y1 = min(gy, y0 + band_rows);
    const uint32_t tb0 = y0 * gx, Tb = (y1 - y0) * gx;           // first tile / tile count of the band
    const uint32_t Tbpad = (band_rows * gx + 63u) & ~63u;
    const uint32_t V = counters[0];
    (void)T;
    // per-wave LDS: [ring 128 x 24 B][MODE 1: bitmap 8 B x Tbpad][counters / cursors]
    unsigned char *base_w = smem + (size_t)wv * wave_bytes;
    uint4 *qa = reinterpret_cast<uint4 *>(base_w);               // raw records {id, rect lo, rect hi, depth}
    uint2 *qb = reinterpret_cast<uint2 *>(qa + W3D_WALK_QUEUE);   // their tile masks
    unsigned char *p = base_w + W3D_WALK_QUEUE * (sizeof(uint4) + sizeof(uint2));
    unsigned long long *bm = reinterpret_cast<unsigned long long *>(p);
    if (MODE == 1) p += (size_t)Tbpad * 8;
    uint32_t *h32 = reinterpret_cast<uint32_t *>(p);             // MODE 0: Tbpad/2 words of two u16 counters; MODE 1: cursors
    // the chunks partition the V VISIBLE records evenly (V is only known on the device), not the P slots the host sized
    // the matrices for — all C chunk-waves of a band get work, each a 1/C-th of it
    chunk = min(chunk, max(64u, ((V + C - 1u) / C + 63u) & ~63u));
    const uint32_t s_beg = min(V, c * chunk), s_end = min(V, s_beg + chunk);
    // the first records are requested before the LDS set-up below, which hides their latency; NB 64-record
    // batches are kept in flight (rotating registers, so the loop body — and process() — exists once)
    constexpr int NB = 4;
    uint4 nx_rec[NB];
    uint2 nx_mask[NB];
    auto fetch_one = [&](uint32_t from, uint4 &r, uint2 &mk) {
        const uint32_t s = from + lane;
        r = make_uint4(0u, 0u, 0u, 0u);
        mk = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        if (s < s_end) { r = rec[s]; if (CULL) mk = rec_mask[s]; }
    };
#pragma unroll
    for (int i = 0; i < NB; i++) fetch_one(s_beg + (uint32_t)i * 64u, nx_rec[i], nx_mask[i]);
    if (MODE == 0) {
        for (uint32_t t = lane; t < Tbpad / 2; t += 64) h32[t] = 0;
    } else {
        const uint32_t *row = off + (size_t)c * T + tb0;
        for (uint32_t t0 = 0; t0 < Tb; t0 += 512) {              // 8 independent loads in flight per lane
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const uint32_t t = t0 + (uint32_t)i * 64u + lane; v[i] = t < Tb ? row[t] : 0u; }
#pragma unroll
            for (int i = 0; i < 8; i++) { const uint32_t t = t0 + (uint32_t)i * 64u + lane; if (t < Tb) h32[t] = v[i]; }
        }
        for (uint32_t t = lane; t < Tbpad; t += 64) bm[t] = 0ull;
    }
    __builtin_amdgcn_wave_barrier();
#define RL(x, i) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (int)(i)))
    uint32_t q_head = 0, q_len = 0;                               // wave-uniform ring state

    // bins the first nq (<= 64) queued records
    auto process = [&](uint32_t nq) {
        const uint32_t slot = (q_head + lane) & (W3D_WALK_QUEUE - 1u);
        // the ring holds the raw records; everything derived from them is computed HERE, once per queued record, not in
        // the scan loop where seven of eight lanes would compute it for records that miss the band
        uint4 er = make_uint4(0u, 0u, 0u, 0u);
        uint2 em = make_uint2(0u, 0u);
        if (lane < nq) { er = qa[slot]; em = qb[slot]; }
        const uint32_t g = er.x, minx = er.y & 0xFFFFu, miny = er.y >> 16, maxx = er.z & 0xFFFFu, maxy = er.z >> 16;
        const uint32_t w = maxx - minx, nt = w * (maxy - miny);
        uint64_t rm = 0ull;
        uint32_t kind = 0u;                                       // 0 = lane-parallel, 1 = whole wave (<= 64 tiles), 2 = whole wave (big rect)
        if (lane < nq) {
            if (nt <= 64u) {
                // bit k: k-th tile of the rect (row-major) can be reached — restricted to the rows of this band
                const uint32_t k_lo = (max(miny, y0) - miny) * w, k_hi = (min(maxy, y1) - miny) * w;
                rm = (k_hi >= 64u ? ~0ull : ((1ull << k_hi) - 1ull)) & ~((1ull << k_lo) - 1ull);
                if (CULL) rm &= (uint64_t)em.x | ((uint64_t)em.y << 32);
                kind = __popcll(rm) > W3D_WALK_SMALL ? 1u : 0u;
            } else {
                kind = 2u;
            }
        }
        // floor(k / w) == (k * magic) >> 16 for k < 64, w <= 64 with magic = floor(65536 / w) + 1; through the float
        // reciprocal (65536 / w is either an integer, where rcp is exact enough, or >= 1/64 away from one)
        const uint32_t magic = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)max(w, 1u)) + 0.004f) + 1u;
        const uint4 eb = make_uint4((uint32_t)rm, (uint32_t)(rm >> 32), er.w, er.z);     // (layout the whole-wave part reads)
        const uint64_t coop = w3d_ballot(lane < nq && kind != 0u);
        // lane-parallel part: the (at most W3D_WALK_SMALL) tiles of this lane's record, derived once and kept in registers
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        uint32_t tls[W3D_WALK_SMALL];
        uint32_t kmax = 0;                                         // wave-uniform: slots [0, kmax) hold a tile for some lane
#pragma unroll
        for (int i = 0; i < W3D_WALK_SMALL; i++) tls[i] = NONE;
        {
            uint64_t m = (lane < nq && kind == 0u) ? rm : 0ull;
            // tile of rect slot k (row ty = k / w): (miny + ty - y0) * gx + minx + k - ty * w = base + k + ty * (gx - w);
            // base may wrap below zero for a rect that starts above the band — the sum is taken modulo 2^32
            const uint32_t base = (miny - y0) * gx + minx, rowskip = gx - w;
#pragma unroll
            for (int i = 0; i < W3D_WALK_SMALL; i++) {
                const bool v = m != 0ull;
                if (w3d_ballot(v) == 0ull) break;                  // every lane has run out of tiles
                kmax = (uint32_t)i + 1u;
                const uint32_t k = (uint32_t)__ffsll((unsigned long long)m) - 1u;
                m &= m - 1ull;
                const uint32_t ty = __umul24(k, magic) >> 16;
                const uint32_t tl = base + k + __umul24(ty, rowskip);
                tls[i] = v ? tl : NONE;
            }
        }
        // whole-wave part: records with many tiles, one at a time, lanes <-> tiles.  op(tile index in band, ring position, id)
        auto traverse_coop = [&](auto op) {
            uint64_t todo = coop;
            while (todo) {
                const uint32_t j = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
                todo &= todo - 1ull;
                const uint32_t jg = RL(g, j), jw = RL(w, j), jminx = RL(minx, j), jminy = RL(miny, j);
                if (RL(kind, j) == 1u) {
                    const uint64_t jm = (uint64_t)RL(eb.x, j) | ((uint64_t)RL(eb.y, j) << 32);
                    const uint32_t ty = __umul24(lane, RL(magic, j)) >> 16;
                    const uint32_t tl = __umul24(jminy + ty - y0, gx) + jminx + (lane - __umul24(ty, jw));
                    if ((jm >> lane) & 1ull) op(tl, j, jg);
                } else {
                    // rect of more than 64 tiles (never culled): rows of the band only, generic division
                    const uint32_t hi = RL(eb.w, j), jmaxx = hi & 0xFFFFu, jmaxy = hi >> 16;
                    const uint32_t r0 = max(jminy, y0), r1 = min(jmaxy, y1), ww = jmaxx - jminx;
                    const uint32_t n = (r1 - r0) * ww;
                    for (uint32_t kb = 0; kb < n; kb += 64) {
                        const uint32_t k = kb + lane;
                        const uint32_t ty = k / ww, tl = (r0 + ty - y0) * gx + jminx + (k - ty * ww);
                        if (k < n) op(tl, j, jg);
                    }
                }
            }
        };
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < W3D_WALK_SMALL; i++) {
                if ((uint32_t)i >= kmax) break;
                if (tls[i] != NONE) atomicAdd(&h32[tls[i] >> 1], 1u << ((tls[i] & 1u) * 16u));
            }
            traverse_coop([&](uint32_t tl, uint32_t, uint32_t) { atomicAdd(&h32[tl >> 1], 1u << ((tl & 1u) * 16u)); });
        } else {
            const unsigned long long mybit = 1ull << lane;
#pragma unroll
            for (int i = 0; i < W3D_WALK_SMALL; i++) {
                if ((uint32_t)i >= kmax) break;
                if (tls[i] != NONE) atomicOr(&bm[tls[i]], mybit);
            }
            traverse_coop([&](uint32_t tl, uint32_t src, uint32_t) { atomicOr(&bm[tl], 1ull << src); });
            __builtin_amdgcn_wave_barrier();
            // all cursor / bitmap reads of the batch are issued back to back, then the list entries are written
#pragma unroll
            for (int i0 = 0; i0 < W3D_WALK_SMALL; i0 += 8) {
                if ((uint32_t)i0 >= kmax) break;
                uint32_t hh[8];
                unsigned long long bb[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    hh[i] = 0u; bb[i] = 0ull;
                    if (tls[i0 + i] != NONE) { hh[i] = h32[tls[i0 + i]]; bb[i] = bm[tls[i0 + i]]; }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t pos = hh[i] + (uint32_t)__popcll(bb[i] & (mybit - 1ull));
                    if (tls[i0 + i] != NONE && pos < capacity) point_list[pos] = g;
                }
            }
            traverse_coop([&](uint32_t tl, uint32_t src, uint32_t id) {
                const uint32_t pos = h32[tl] + (uint32_t)__popcll(bm[tl] & ((1ull << src) - 1ull));
                if (pos < capacity) point_list[pos] = id;
            });
            __builtin_amdgcn_wave_barrier();
            // (round 4 tried advancing the cursors through the touched tiles' LEADER records — lowest bit of the bitmap — instead of
            //  this sweep over the band: 0.152 -> 0.175 ms; sixteen exec-masked LDS read-modify-writes per lane cost more than
            //  five coalesced sweeps of 64 tiles)
            for (uint32_t t0 = 0; t0 < Tb; t0 += 512) {          // cursors advance by the batch's hits; 8 reads in flight
                unsigned long long b8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) { const uint32_t t = t0 + (uint32_t)i * 64u + lane; b8[i] = t < Tb ? bm[t] : 0ull; }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t t = t0 + (uint32_t)i * 64u + lane;
                    if (b8[i]) { h32[t] += (uint32_t)__popcll(b8[i]); bm[t] = 0ull; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (uint32_t base = s_beg; base < s_end; base += 64u) {
        const uint32_t nb = min(64u, s_end - base);
        const uint4 cur_rec = nx_rec[0];
        const uint2 cur_mask = nx_mask[0];
#pragma unroll
        for (int i = 0; i + 1 < NB; i++) { nx_rec[i] = nx_rec[i + 1]; nx_mask[i] = nx_mask[i + 1]; }
        fetch_one(base + 64u * NB, nx_rec[NB - 1], nx_mask[NB - 1]);
        // rect reaches into this band?  (whether its tile mask does is settled when the record is binned)
        const bool relevant = lane < nb && (cur_rec.y >> 16) < y1 && (cur_rec.z >> 16) > y0;
        const uint64_t bal = w3d_ballot(relevant);
        if (bal) {
            if (relevant) {
                const uint32_t slot = (q_head + q_len + (uint32_t)__popcll(bal & lanemask_lt())) & (W3D_WALK_QUEUE - 1u);
                qa[slot] = cur_rec; qb[slot] = cur_mask;
            }
            q_len += (uint32_t)__popcll(bal);
            __builtin_amdgcn_wave_barrier();
        }
        // bin 64 queued records — or, after the last batch, whatever is left (the same code, so process() is instantiated once)
        const bool last = base + 64u >= s_end;
        while (q_len >= 64u || (last && q_len)) {
            const uint32_t nq = min(q_len, 64u);
            process(nq);
            q_head = (q_head + nq) & (W3D_WALK_QUEUE - 1u);
            q_len -= nq;
        }
    }
#undef RL
    if (MODE == 0) {
        __builtin_amdgcn_wave_barrier();
        uint16_t *row = cnt + (size_t)c * T + tb0;
        const uint16_t *h16 = reinterpret_cast<const uint16_t *>(h32);
        for (uint32_t t = lane; t < Tb; t += 64) row[t] = h16[t];
    }
}

// ------------------------------------------------------------------------------ offset scan
// part[sg][t] = sum over the chunks of segment sg of cnt[c][t]
__global__ void __launch_bounds__(256)
seg_sum_kernel(const uint16_t *__restrict__ cnt, uint32_t C, uint32_t T, uint32_t seg, uint32_t *__restrict__ part) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, sg = blockIdx.y;
    if (t >= T) return;
    const uint32_t c0 = sg * seg, c1 = min(C, sg * seg + seg);
    uint32_t s = 0;
#pragma unroll 16
    for (uint32_t c = c0; c < c1; c++) s += cnt[(size_t)c * T + t];      // independent loads: keep many in flight
    part[(size_t)sg * T + t] = s;
}

// one block: totals per tile -> exclusive scan -> tile_start[T+1]; counters[1] = total list length.  The totals of up to 8192
// tiles are first summed COALESCED (thread <-> tile, all loads of a thread independent) into LDS; then every thread owns a run
// of 8 consecutive tiles, so the block scans once over the 1024 run sums instead of once per 1024 tiles (the chain of
// eight block scans with their barriers took 15 us to move 0.5 MB).
__global__ void __launch_bounds__(1024)
tile_scan_kernel(const uint32_t *__restrict__ part, uint32_t T, uint32_t nseg, uint32_t *__restrict__ tile_start,
                 uint32_t *__restrict__ counters, uint32_t share_code) {
    constexpr uint32_t PER = 8, SPAN = 1024u * PER;
    __shared__ uint32_t wave_tot[17];
    __shared__ uint32_t tot_s[SPAN];
    uint32_t carry = 0u;
    for (uint32_t sweep0 = 0; sweep0 < T; sweep0 += SPAN) {     // (one sweep up to 8192 tiles; 4K frames take several)
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) {
            const uint32_t t = sweep0 + i * 1024u + threadIdx.x;
            uint32_t a = 0;
            if (t < T) {
#pragma unroll 16
                for (uint32_t sg = 0; sg < nseg; sg++) a += part[(size_t)sg * T + t];
            }
            tot_s[i * 1024u + threadIdx.x] = a;
        }
        __syncthreads();
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) { v[i] = tot_s[threadIdx.x * PER + i]; sum += v[i]; }
        uint32_t tot;
        uint32_t run = carry + block_exclusive_scan(sum, wave_tot, tot);
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) { tot_s[threadIdx.x * PER + i] = run; run += v[i]; }
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) {
            const uint32_t t = sweep0 + i * 1024u + threadIdx.x;
            if (t < T) tile_start[t] = tot_s[i * 1024u + threadIdx.x];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tile_start[T] = carry;
        counters[1] = carry;
        counters[6] = share_code;       // lsx | lsy << 8 of the list grid these ranges live on (w3d_debug_tile_ranges)
    }
}

// off[c][t] = start of chunk c's entries inside tile t's list
__global__ void __launch_bounds__(256)
chunk_off_kernel(const uint16_t *__restrict__ cnt, const uint32_t *__restrict__ part, const uint32_t *__restrict__ tile_start,
                 uint32_t C, uint32_t T, uint32_t seg, uint32_t *__restrict__ off) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, sg = blockIdx.y;
    if (t >= T) return;
    const uint32_t c0 = sg * seg, c1 = min(C, sg * seg + seg);
    uint32_t run = tile_start[t];
    for (uint32_t s = 0; s < sg; s++) run += part[(size_t)s * T + t];
    uint32_t c = c0;
    for (; c + 8 <= c1; c += 8) {         // 8 count loads in flight per step of the running sum
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = cnt[(size_t)(c + i) * T + t];
#pragma unroll
        for (int i = 0; i < 8; i++) { off[(size_t)(c + i) * T + t] = run; run += v[i]; }
    }
    for (; c < c1; c++) {
        off[(size_t)c * T + t] = run;
        run += cnt[(size_t)c * T + t];
    }
}

// per 16x16 tile: the range of the list it reads (with w3d_view.list_share the tiles of one list cell report the same range;
// the share mode of the forward that wrote this state is kept in counters[6])
__global__ void copy_ranges_kernel(const uint32_t *__restrict__ tile_start, uint32_t T, uint32_t gx,
                                   const uint32_t *__restrict__ counters, uint32_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lsx = counters[6] & 0xFFu, lsy = (counters[6] >> 8) & 0xFFu;
    const uint32_t lgx = (gx + (1u << lsx) - 1u) >> lsx;
    if (t < T) {
        const uint32_t l = ((t / gx) >> lsy) * lgx + ((t % gx) >> lsx);
        out[2 * t] = tile_start[l]; out[2 * t + 1] = tile_start[l + 1];
    }
}

// Bands of tile rows for the chunk walk: each (chunk, band) pair is one wave.  Narrow bands shrink the wave's LDS
// footprint (more waves per CU behind the latency of the walk) and the share of a chunk's records that concern it;
// every band adds a pass over the chunk's 24-B records.
struct W3DBands { uint32_t rows, count, tbpad; };
W3DBands w3d_pick_bands(const W3DLayout &L, int mode) {
    W3DBands b;
#ifndef W3D_COUNT_BAND_TILES
#define W3D_COUNT_BAND_TILES 1024u
#endif
#ifndef W3D_FILL_BAND_TILES
#define W3D_FILL_BAND_TILES 320u
#endif
    uint32_t band_tiles = mode == 0 ? W3D_COUNT_BAND_TILES : W3D_FILL_BAND_TILES;  // tiles per band (the fill pass holds 12 B of LDS per tile, the count pass 2 B;
                                                     // measured fill at 1600x1200: 200 -> 0.188 ms, 300 -> 0.171, 500 -> 0.179, 700 -> 0.193)
    uint32_t rows = band_tiles / (uint32_t)L.lgx;
    if (rows < 1) rows = 1;
    if (rows > (uint32_t)L.lgy) rows = (uint32_t)L.lgy;
    b.rows = rows;
    b.count = ((uint32_t)L.lgy + rows - 1) / rows;
    b.tbpad = (rows * (uint32_t)L.lgx + 63u) & ~63u;
    return b;
}
}  // namespace

// stable depth sort of the Gaussians + depth-ordered packed records
int w3d_launch_depth_sort(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, hipStream_t stream) {
    uint32_t *counters = reinterpret_cast<uint32_t *>(state + L.o_counters);
    uint32_t *keys[2] = {reinterpret_cast<uint32_t *>(scratch + L.s_keys0), reinterpret_cast<uint32_t *>(scratch + L.s_keys1)};
    uint32_t *vals[2] = {reinterpret_cast<uint32_t *>(scratch + L.s_vals0), reinterpret_cast<uint32_t *>(scratch + L.s_vals1)};
    uint32_t *hist = reinterpret_cast<uint32_t *>(scratch + L.s_hist);
    uint32_t *rowtot = reinterpret_cast<uint32_t *>(scratch + L.s_rowtot);
    if (L.P == 0) return W3D_OK;
    uint32_t *minmax = reinterpret_cast<uint32_t *>(scratch + L.s_minmax);
    {
        // ---- stable LSD radix sort of (depth bits, id): 3 or 4 passes of 8 bits (radix_digit); culled Gaussians carry key
        // 0xFFFFFFFF and are dropped by the first pass
        const uint32_t n = (uint32_t)L.P, runs = L.sort_waves, blocks = (runs + 3) / 4;
        int src = 0;
        W3D_PROF("depth_sort", stream);
        for (int pass = 0; pass < W3D_RADIX_PASSES; pass++) {
            // pass 0 reads all P keys, drops the culled ones and publishes V = counters[0]; passes 1.. sort V keys
            const uint32_t *n_dev = pass == 0 ? (const uint32_t *)nullptr : counters;
            hipLaunchKernelGGL(radix_hist_kernel, dim3(blocks), dim3(256), 0, stream, keys[src], n, L.sort_items, runs, pass, hist,
                               n_dev, pass == 0 ? 1 : 0, counters, pass == 0 ? minmax : (uint32_t *)nullptr);
            W3D_LAUNCH_CHECK(v.debug, stream);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(W3D_RADIX_BINS), dim3(256), 0, stream, hist, runs, rowtot, pass, counters,
                               minmax);
            W3D_LAUNCH_CHECK(v.debug, stream);
#define SCATTER_ARGS                                                                                                          \
    keys[src], vals[src], keys[src ^ 1], vals[src ^ 1], n, L.sort_items, runs, pass, hist, rowtot,                            \
        pass == 0 ? counters : (uint32_t *)nullptr, n_dev, counters, reinterpret_cast<const uint2 *>(state + L.o_rect),       \
        reinterpret_cast<const uint4 *>(state + L.o_tile_mask), reinterpret_cast<uint4 *>(scratch + L.s_rec),                 \
        reinterpret_cast<uint2 *>(scratch + L.s_rec_mask), (int)v.tile_cull
            if (pass >= 2) hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(blocks), dim3(256), 0, stream, SCATTER_ARGS);
            else hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(blocks), dim3(256), 0, stream, SCATTER_ARGS);
#undef SCATTER_ARGS
            W3D_LAUNCH_CHECK(v.debug, stream);
            src ^= 1;
        }
    }
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

template <int MODE>
static void launch_walk(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, uint32_t *point_list,
                        uint64_t capacity, hipStream_t stream) {
    const W3DBands bands = w3d_pick_bands(L, MODE);
    const dim3 grid(L.C, (bands.count + W3D_WW - 1) / W3D_WW);
    const uint32_t wave_bytes = W3D_WALK_QUEUE * 24u + bands.tbpad * (MODE == 0 ? 2u : 12u);
    const size_t lds = (size_t)wave_bytes * W3D_WW;
    const uint4 *rec = reinterpret_cast<const uint4 *>(scratch + L.s_rec);
    const uint2 *rmask = reinterpret_cast<const uint2 *>(scratch + L.s_rec_mask);
    const uint32_t *counters = reinterpret_cast<const uint32_t *>(state + L.o_counters);
    uint16_t *cnt = reinterpret_cast<uint16_t *>(scratch + L.s_cnt);
    const uint32_t *off = reinterpret_cast<const uint32_t *>(scratch + L.s_off);
    if (v.tile_cull)
        hipLaunchKernelGGL((chunk_walk_kernel<MODE, true>), grid, dim3(64 * W3D_WW), lds, stream, rec, rmask, counters, L.chunk, L.C,
                           (uint32_t)L.LT, (uint32_t)L.lgx, (uint32_t)L.lgy, bands.rows, cnt, off, point_list, capacity, wave_bytes);
    else
        hipLaunchKernelGGL((chunk_walk_kernel<MODE, false>), grid, dim3(64 * W3D_WW), lds, stream, rec, rmask, counters, L.chunk, L.C,
                           (uint32_t)L.LT, (uint32_t)L.lgx, (uint32_t)L.lgy, bands.rows, cnt, off, point_list, capacity, wave_bytes);
}

// per-chunk per-tile counts and the list offsets
int w3d_launch_tile_count(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, hipStream_t stream) {
    uint32_t *counters = reinterpret_cast<uint32_t *>(state + L.o_counters);
    uint32_t *tile_start = reinterpret_cast<uint32_t *>(state + L.o_tile_start);
    uint16_t *cnt = reinterpret_cast<uint16_t *>(scratch + L.s_cnt);
    uint32_t *part = reinterpret_cast<uint32_t *>(scratch + L.s_part);
    uint32_t *off = reinterpret_cast<uint32_t *>(scratch + L.s_off);
    const uint32_t T = (uint32_t)L.LT;           // (lists live on the list grid: w3d_view.list_share)
    W3D_PROF("tile_count_scan", stream);
    launch_walk<0>(L, v, state, scratch, nullptr, 0, stream);
    W3D_LAUNCH_CHECK(v.debug, stream);
    const uint32_t tb = (T + 255) / 256;
    hipLaunchKernelGGL(seg_sum_kernel, dim3(tb, W3D_SCAN_SEGS), dim3(256), 0, stream, cnt, L.C, T, L.seg, part);
    W3D_LAUNCH_CHECK(v.debug, stream);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, part, T, (uint32_t)W3D_SCAN_SEGS, tile_start, counters,
                       (uint32_t)L.lsx | ((uint32_t)L.lsy << 8));
    W3D_LAUNCH_CHECK(v.debug, stream);
    hipLaunchKernelGGL(chunk_off_kernel, dim3(tb, W3D_SCAN_SEGS), dim3(256), 0, stream, cnt, part, tile_start, L.C, T, L.seg, off);
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

int w3d_launch_fill_lists(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, uint32_t *point_list,
                          uint64_t list_capacity, hipStream_t stream) {
    W3D_PROF("fill_lists", stream);
    launch_walk<1>(L, v, state, scratch, point_list, list_capacity, stream);
    W3D_LAUNCH_CHECK(v.debug, stream);
    return W3D_OK;
}

int w3d_debug_tile_ranges_impl(const W3DLayout &L, const char *state, uint32_t *ranges_out, hipStream_t stream) {
    const uint32_t T = (uint32_t)L.T;
    hipLaunchKernelGGL(copy_ranges_kernel, dim3((T + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(state + L.o_tile_start), T, (uint32_t)L.gx,
                       reinterpret_cast<const uint32_t *>(state + L.o_counters), ranges_out);
    W3D_HIP_CHECK(hipGetLastError());
    return W3D_OK;
}
