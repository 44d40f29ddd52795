// w3d_binning.hip — depth sort of the Gaussians and construction of the per-tile lists
// (replaces the scan / duplicate-with-keys / 64-bit radix sort /
// identify-ranges stage of the reference's CUDA submodule).
//
// MI355X-first formulation.  Instead of materialising R = sum(tiles touched) 64-bit
// (tile|depth) keys and radix-sorting all of them through HBM, the order contract
// "within a tile ascending depth bits, ties by ascending Gaussian index" is met in two steps:
//   1. ONE stable sort of the (depth bits, index) pairs — P << R, 16 B per Gaussian — as a 1024-way bucket split over the view's
//      depth interval + an in-LDS sort of every bucket (five launches, "depth sort" below); the split drops the culled
//      Gaussians and publishes V;
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

// ------------------------------------------------------------------------------ depth sort
// Stable sort of the visible Gaussians by (depth bits, index) in FIVE launches (rounds 1-5: an LSD radix sort, 3-4 passes x
// {histogram, row scan, scatter} = 12 launches, 0.162 ms for 1.2 M keys — a chain of short latency-bound kernels, at the level of
// rocPRIM's one-sweep sort, 139 us, on this part).
//
// The keys are the bit patterns of positive floats (view depths > 0.2), monotone in the depth, and ONE view's depths are a narrow
// interval of them.  Every workgroup of the preprocess kernel leaves {min, max} of its visible keys and two of its keys (a block reduction and one 16-B
// store, nothing to initialise).  Then
//   0. depth_grid             one workgroup: the view's interval [kmin, kmax] from those entries and the bucket grid over it (below:
//                             W3D_DB_BINS = 1024 buckets, piecewise linear in the key, even in POPULATION as far as a sample tells);
//   1. depth_bucket_hist      one histogram of bucket(key) per run (= the slice of the Gaussian order one wave owns);
//   2. depth_bucket_scan      per bucket: exclusive scan of the run counts + bucket total;
//   3. depth_bucket_scatter   stable multi-split of the (key, index) pairs into the buckets (culled Gaussians dropped, V published);
//   4. depth_bucket_sort      one workgroup per bucket: the bucket (1 200 pairs on the benchmark view) is sorted IN LDS by the key's
//                             offset from the bucket's lower bound (`rbits` bits: the width of a bucket) — ceil(rbits / 8) stable
//                             8-bit passes, 2 for the benchmark cameras — and written out as
//                             the depth-ordered packed records {id, rect lo, rect hi, depth} + tile mask the (chunk, band) walkers
//                             stream.  A bucket that does not fit the LDS arrays (everything at one depth, 40 M Gaussians, a pile
//                             the grid's sample did not see) takes the same passes through its own slice of the two global
//                             (key, id) buffers — slower, same result (with round 6's first, equal-width grid the densified
//                             benchmark scene had 140-230 such buckets on some cameras: 0.141 ms against 0.099; a second launch
//                             with 128 KB of LDS for those buckets was measured and is slower still: its 1024 one-per-CU
//                             workgroups cost 12 us when idle).
// Bucket b holds exactly the keys of [kmin + lo(b), kmin + lo(b) + width(b)) (depth_bucket_range); the scatter keeps index order inside a
// bucket and the in-bucket passes are stable, so the result is THE stable order by (depth bits, index), whatever the depth
// distribution (tests/test_gpu_parity.py::test_depth_sort_paths_and_tie_order_at_size: lists bit-identical to the oracle's).
#define W3D_CTL_KMIN 4          // counters[4] = smallest visible depth key
#define W3D_CTL_MUL 5           // counters[5] = the grid's segment multiplier (0: the interval has fewer than 1024 key values, bucket = key - kmin)
#define W3D_DB_CAP 4096         // items a bucket may hold to be sorted in LDS (one 32-bit word each, 2 x 16 KB of ping-pong arrays: 4 workgroups per CU)
#define W3D_DB_FAST 2048        // ... and up to this many take the path that keeps the pairs in registers (depth_bucket_sort_kernel)

// ---- the bucket grid of a view.  Equal-WIDTH buckets over [kmin, kmax] are only balanced when the depths are spread evenly: one
// Gaussian in a thousand far behind the scene (a background 30-60 units away) stretches the interval to five octaves, the scene keeps
// 14 % of the buckets, 50-90 of them hold 8-20 k keys and fall out of the LDS paths — depth sort 0.094 -> 0.235 ms
// (profiles/skewed_depth_probe.py).  So the grid is piecewise linear: the interval is cut into W3D_DB_SEGS = 64 equal segments, a
// fixed SAMPLE of the keys (two per preprocess workgroup, at most 16384, spread over the Gaussian order: ~150 visible samples per
// segment on an even view — with 256 segments and ~10 the sampling noise alone doubled some buckets) estimates each segment's
// population, every segment gets one bucket plus its proportional share of the other 960, and inside a segment the buckets are
// equal-width again:  seg = ((key - kmin) * mseg) >> 32,  bucket = first[seg] + min(count[seg] - 1, ((key - kmin - start[seg]) *
// slope[seg]) >> 32).  Monotone in the key whatever the sample says, so the sort stays exact; the sample only decides how even the
// buckets come out.  The table (one uint4 per segment: start, slope, first | count << 16, end) is built by depth_grid_kernel.
#define W3D_DB_SEGS 64
#define W3D_DB_SEG_BITS 6
struct DepthGridHead { uint32_t kmin, mseg, nbuckets, pad; };       // mseg == 0: fewer than 1024 key values, bucket = key - kmin
// LDS image of the grid: tab[seg] = {start, slope (0: one key value per bucket), first | count << 16, end}
__device__ __forceinline__ uint32_t depth_bucket(uint32_t key, const DepthGridHead &h, const uint4 *tab) {
    const uint32_t x = key - h.kmin;
    if (h.mseg == 0u) return x;
    const uint4 t = tab[__umulhi(x, h.mseg)];
    const uint32_t loc = t.y ? __umulhi(x - t.x, t.y) : x - t.x;
    return (t.z & 0xFFFFu) + min(loc, (t.z >> 16) - 1u);
}
// [lo, hi) of the key offsets (key - kmin) bucket b holds; false: the bucket is past the grid's last one
__device__ __forceinline__ bool depth_bucket_range(uint32_t b, const DepthGridHead &h, const uint4 *tab, uint32_t &lo, uint32_t &hi) {
    if (h.mseg == 0u) { lo = b; hi = b + 1u; return b < h.nbuckets; }
    if (b >= h.nbuckets) return false;
    uint32_t s0 = 0, s1 = W3D_DB_SEGS;                       // last segment whose first bucket is <= b
    while (s1 - s0 > 1u) { const uint32_t mid = (s0 + s1) >> 1; if ((tab[mid].z & 0xFFFFu) <= b) s0 = mid; else s1 = mid; }
    const uint4 t = tab[s0];
    const uint32_t loc = b - (t.z & 0xFFFFu), cnt = t.z >> 16;
    // smallest offset x in the segment with ((x - start) * slope) >> 32 >= l: ceil(l * 2^32 / slope) — a double-precision
    // estimate, then made exact against the very expression depth_bucket() evaluates
    auto lower = [&](uint32_t l) -> uint32_t {
        if (t.y == 0u) return t.x + l;
        if (l == 0u) return t.x;
        uint32_t x = (uint32_t)fmin(4294967295.0, (double)l * 4294967296.0 / (double)t.y);
        while (__umulhi(x, t.y) < l) x++;
        while (x > 0u && __umulhi(x - 1u, t.y) >= l) x--;
        return t.x + x;
    };
    lo = lower(loc);
    hi = loc + 1u >= cnt ? t.w : lower(loc + 1u);
    return true;
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

// The view's grid: ONE workgroup of 1024 threads (built inside the histogram kernel by each of its 256 one-per-CU workgroups the
// interval reduction + sample + scans sat on every workgroup's critical path: 25 -> 39 us).  Publishes head + table (grid_out) and
// every bucket's interval of key offsets {lo, width} (brange).  A single workgroup lives on latency, so: every thread keeps its
// entries in registers (one round of loads up to 2 M Gaussians), the 64-bit integer divisions (a software routine of ~200
// instructions) are left to the one wave that owns the segments, and the sample is counted into 16 interleaved copies of the
// segment histogram (neighbours in the Gaussian order lie at about the same depth: 64 lanes adding to one LDS word serialize).
#define W3D_DG_THREADS 1024
#define W3D_DG_HOLD 8            // entries a thread keeps in registers: one round of loads up to 8192 preprocess workgroups = 2 M Gaussians
#define W3D_DG_COPIES 16
__global__ void __launch_bounds__(W3D_DG_THREADS)
depth_grid_kernel(const uint4 *__restrict__ entries, uint32_t nb,
                  uint4 *__restrict__ grid_out /* [1 + W3D_DB_SEGS] */, uint2 *__restrict__ brange /* [W3D_DB_BINS] */,
                  uint32_t *__restrict__ counters) {
    __shared__ __align__(16) uint4 tab[W3D_DB_SEGS];
    __shared__ uint32_t red[32], chist[W3D_DG_COPIES][W3D_DB_SEGS + 1], s_head[4];
    static_assert(W3D_DB_SEGS == 64, "one wave owns the segments");
    static_assert(W3D_DB_BINS == W3D_DG_THREADS, "one thread per bucket");
    const uint32_t tid = threadIdx.x;
    // the per-workgroup entries the preprocess left: nb = ceil(P / W3D_PRE_BLOCK) x {min, max, two of the workgroup's keys} — 125 KB of
    // coalesced reads at 2 M Gaussians.  The two keys are the grid's population sample: one Gaussian in 128, spread evenly over the
    // Gaussian order (in Morton order: over space); gathering a sample from the key array here cost 20 us — 8192 scattered lines
    // through ONE CU.  Beyond W3D_DG_HOLD x 1024 entries the sample is every stride-th entry and the rest only feed the interval.
    const uint32_t stride = (nb + W3D_DG_HOLD * W3D_DG_THREADS - 1) / (W3D_DG_HOLD * W3D_DG_THREADS);
    const uint32_t ns = stride ? (nb + stride - 1) / stride : 0u;
    uint4 m[W3D_DG_HOLD];
#pragma unroll
    for (int u = 0; u < W3D_DG_HOLD; u++) {
        const uint32_t i = tid + (uint32_t)W3D_DG_THREADS * u;
        m[u] = i < ns ? entries[(size_t)i * stride] : make_uint4(0xFFFFFFFFu, 0u, W3D_INVALID_KEY, W3D_INVALID_KEY);
    }
    for (uint32_t i = tid; i < W3D_DG_COPIES * (W3D_DB_SEGS + 1); i += W3D_DG_THREADS) (&chist[0][0])[i] = 0u;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int u = 0; u < W3D_DG_HOLD; u++) { kmin = min(kmin, m[u].x); kmax = max(kmax, m[u].y); }
    if (stride > 1u)
        for (uint32_t i = tid; i < nb; i += W3D_DG_THREADS) { const uint4 e = entries[i]; kmin = min(kmin, e.x); kmax = max(kmax, e.y); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    if ((tid & 63) == 0) { red[tid >> 6] = kmin; red[16 + (tid >> 6)] = kmax; }
    __syncthreads();
    if (tid < 64) {
        kmin = red[tid & 15]; kmax = red[16 + (tid & 15)];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        }
        if (tid == 0) {
            uint32_t mseg = 0u, nbk = 0u;
            const bool any = kmin <= kmax;
            const uint32_t span = any ? kmax - kmin : 0u;
            if (!any) kmin = 0u;                                                // nothing visible: no buckets
            else if (span < W3D_DB_BINS) nbk = span + 1u;                       // one key value per bucket: nothing to sort inside
            else mseg = (uint32_t)((1ull << (32 + W3D_DB_SEG_BITS)) / ((uint64_t)span + 1ull));   // (span * mseg) >> 32 < W3D_DB_SEGS; every segment >= 16 key values wide
            s_head[0] = kmin; s_head[1] = mseg; s_head[2] = nbk; s_head[3] = span;
        }
    }
    __syncthreads();
    DepthGridHead h;
    h.kmin = s_head[0]; h.mseg = s_head[1]; h.nbuckets = s_head[2]; h.pad = 0u;
    if (h.mseg) {
        const uint32_t span = s_head[3];
        uint32_t *mine = chist[tid & (W3D_DG_COPIES - 1)];
#pragma unroll
        for (int u = 0; u < 2 * W3D_DG_HOLD; u++) {
            const uint32_t key = (u & 1) ? m[u >> 1].w : m[u >> 1].z;
            if (key != W3D_INVALID_KEY) atomicAdd(&mine[__umulhi(key - h.kmin, h.mseg)], 1u);
        }
        __syncthreads();
        if (tid < W3D_DB_SEGS) {
            // lane <-> segment: key-offset interval [start, end), share of the buckets, first bucket
            const uint32_t start = (uint32_t)((((uint64_t)tid << 32) + h.mseg - 1u) / h.mseg);          // smallest x with (x * mseg) >> 32 == tid
            const uint32_t nxt = (uint32_t)__shfl_down((int)start, 1, 64);
            const uint32_t end = tid + 1u == W3D_DB_SEGS ? span + 1u : nxt;
            uint32_t c = 0u;
#pragma unroll
            for (int k = 0; k < W3D_DG_COPIES; k++) c += chist[k][tid];
            const uint32_t nsamp = (uint32_t)__shfl((int)wave_inclusive_scan(c), 63, 64);
            const uint32_t width = end - start;
            // (c <= 16384, the share is below 2^24: 32-bit arithmetic)
            uint32_t cnt = 1u + ((uint32_t)(W3D_DB_BINS - W3D_DB_SEGS) * c) / max(nsamp, 1u);
            cnt = min(cnt, width);                                             // never more buckets than key values
            const uint32_t inc = wave_inclusive_scan(cnt);
            // slope: ((x - start) * slope) >> 32 in [0, cnt) for x - start < width; cnt == width: one key value per bucket (slope 0)
            const uint32_t slope = cnt == width ? 0u : (uint32_t)(((uint64_t)cnt << 32) / width);
            const uint4 t = make_uint4(start, slope, (inc - cnt) | (cnt << 16), end);
            tab[tid] = t; grid_out[1 + tid] = t;
            if (tid == W3D_DB_SEGS - 1) s_head[2] = inc;
        }
        __syncthreads();
        h.nbuckets = s_head[2];
    }
    if (tid == 0) { grid_out[0] = make_uint4(h.kmin, h.mseg, h.nbuckets, 0u); counters[W3D_CTL_KMIN] = h.kmin; counters[W3D_CTL_MUL] = h.mseg; }
    // thread <-> bucket: its interval of key offsets (the in-bucket sort reads these 8 bytes instead of the table)
    uint32_t lo = 0u, hi = 0u;
    const bool ok = depth_bucket_range(tid, h, tab, lo, hi);
    brange[tid] = ok ? make_uint2(lo, hi - lo) : make_uint2(0u, 0u);
}

// the published grid -> LDS (256 threads; includes the barrier)
__device__ __forceinline__ DepthGridHead depth_grid_load(const uint4 *__restrict__ grid, uint4 *tab) {
    const uint4 h4 = grid[0];
    DepthGridHead h;
    h.kmin = h4.x; h.mseg = h4.y; h.nbuckets = h4.z; h.pad = 0u;
    if (h.mseg && threadIdx.x < W3D_DB_SEGS) tab[threadIdx.x] = grid[1 + threadIdx.x];
    __syncthreads();
    return h;
}

// one run per wave, four runs per workgroup; hist[bucket][run] (row pitch = runs rounded up to 4: the four counts of a workgroup
// leave as one 16-B store per bucket)
__global__ void __launch_bounds__(256)
depth_bucket_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t items, uint32_t n_runs, uint32_t pitch,
                         uint32_t *__restrict__ hist, const uint4 *__restrict__ grid) {
    __shared__ __align__(16) uint32_t h_all[4][W3D_DB_BINS];
    __shared__ __align__(16) uint4 tab[W3D_DB_SEGS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t run = blockIdx.x * 4 + wv;
    uint4 *hz = reinterpret_cast<uint4 *>(h_all[wv]);
#pragma unroll
    for (int i = 0; i < W3D_DB_BINS / 256; i++) hz[lane + 64 * i] = make_uint4(0u, 0u, 0u, 0u);
    const DepthGridHead dg = depth_grid_load(grid, tab);
    if (run < n_runs) {
        // (per-lane LDS atomics: finding the lanes of one bucket by ballots first — in Morton storage order a batch falls into a
        //  handful of buckets — was measured and is slower, 25 -> 32 us)
        const uint32_t beg = min(n, run * items), end = min(n, beg + items);
        for (uint32_t i0 = beg + lane; i0 < end; i0 += 64u * 16u) {        // (sixteen loads in flight: the scatter kernel explains)
            uint32_t kk[16];
#pragma unroll
            for (int u = 0; u < 16; u++) kk[u] = i0 + 64u * u < end ? keys[i0 + 64u * u] : W3D_INVALID_KEY;
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (kk[u] != W3D_INVALID_KEY) atomicAdd(&h_all[wv][depth_bucket(kk[u], dg, tab)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < W3D_DB_BINS; b += 256)
        *reinterpret_cast<uint4 *>(&hist[(size_t)b * pitch + 4u * blockIdx.x]) = make_uint4(h_all[0][b], h_all[1][b], h_all[2][b], h_all[3][b]);
}

// one workgroup per bucket: exclusive scan of its row of run counts (in place, four runs per thread) + the bucket's total
__global__ void __launch_bounds__(256)
depth_bucket_scan_kernel(uint32_t *__restrict__ hist, uint32_t pitch, uint32_t *__restrict__ rowtot) {
    __shared__ uint32_t wave_tot[17];
    uint4 *row = reinterpret_cast<uint4 *>(hist + (size_t)blockIdx.x * pitch);
    uint32_t carry = 0;
    for (uint32_t base = 0; base < pitch / 4u; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint4 v = i < pitch / 4u ? row[i] : make_uint4(0u, 0u, 0u, 0u);
        uint32_t tot;
        const uint32_t ex = carry + block_exclusive_scan(v.x + v.y + v.z + v.w, wave_tot, tot);
        if (i < pitch / 4u) row[i] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        carry += tot;
    }
    if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}

// stable multi-split of the (key, index) pairs into the buckets; the value of key i IS i (the preprocess writes the keys in
// Gaussian order and no id array); culled Gaussians (key 0xFFFFFFFF) are dropped; V and the bucket boundaries are published
__global__ void __launch_bounds__(256)
depth_bucket_scatter_kernel(const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                            uint32_t n, uint32_t items, uint32_t n_runs, uint32_t pitch, const uint32_t *__restrict__ offs,
                            const uint32_t *__restrict__ rowtot, uint32_t *__restrict__ counters, uint32_t *__restrict__ bstart,
                            const uint4 *__restrict__ grid) {
    __shared__ uint32_t cur_all[4][W3D_DB_BINS];
    __shared__ __align__(16) uint4 tab[W3D_DB_SEGS];
    __shared__ uint32_t wave_tot[17];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t run = blockIdx.x * 4 + wv;
    const DepthGridHead dg = depth_grid_load(grid, tab);
    {
        // bucket bases = exclusive scan of the bucket totals (four consecutive buckets per thread); cursor of run r in bucket b =
        // base[b] + (counts of the runs before r in b)
        const uint4 t4 = reinterpret_cast<const uint4 *>(rowtot)[threadIdx.x];
        uint32_t total;
        uint32_t base = block_exclusive_scan(t4.x + t4.y + t4.z + t4.w, wave_tot, total);
        const uint32_t tot[4] = {t4.x, t4.y, t4.z, t4.w};
        if (blockIdx.x == 0 && threadIdx.x == 0) { counters[0] = total; bstart[W3D_DB_BINS] = total; }   // all counted keys = the visible Gaussians
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t b = threadIdx.x * 4u + i;
            const uint4 o4 = *reinterpret_cast<const uint4 *>(&offs[(size_t)b * pitch + 4u * blockIdx.x]);
            cur_all[0][b] = base + o4.x; cur_all[1][b] = base + o4.y; cur_all[2][b] = base + o4.z; cur_all[3][b] = base + o4.w;
            if (blockIdx.x == 0) bstart[b] = base;
            base += tot[i];
        }
    }
    __syncthreads();
    if (run >= n_runs) return;
    uint32_t *cur = cur_all[wv];
    const uint32_t beg = min(n, run * items), end = min(n, beg + items);
    const uint64_t lt = lanemask_lt();
    // sixteen batches of keys (1024: a whole run at P <= 2 M) are requested at once, then ranked one after the other: with two
    // waves per SIMD nothing else hides the latency of the loads (two batches in flight: 27 us, sixteen: see DESIGN.md section 2b)
    constexpr int NBATCH = 16;
    for (uint32_t base0 = beg; base0 < end; base0 += 64u * NBATCH) {
        uint32_t kk[NBATCH];
#pragma unroll
        for (int u = 0; u < NBATCH; u++) {
            const uint32_t i = base0 + 64u * u + lane;
            kk[u] = i < end ? keys_in[i] : W3D_INVALID_KEY;
        }
#pragma unroll
        for (int u = 0; u < NBATCH; u++) {
            const uint32_t i = base0 + 64u * u + lane;
            if (base0 + 64u * u >= end) break;                              // (wave-uniform)
            const uint32_t key = kk[u];
            const bool valid = key != W3D_INVALID_KEY;
            const uint32_t d = valid ? depth_bucket(key, dg, tab) : 0u;
            // lanes holding the same bucket (stable rank = number of such lanes below me)
            uint64_t peers = w3d_ballot(valid);
#pragma unroll
            for (int b = 0; b < W3D_DB_BITS; b++) {
                const uint64_t m = w3d_ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const uint32_t rank = __popcll(peers & lt);
            uint32_t pos = 0;
            if (valid) pos = cur[d] + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) cur[d] = pos + (uint32_t)__popcll(peers);   // group leader advances the cursor
            __builtin_amdgcn_wave_barrier();
            if (valid) { keys_out[pos] = key; vals_out[pos] = i; }
        }
    }
}

// One stable 8-bit pass of a workgroup (256 threads) over n items of ITS bucket, src -> dst; the items are cut into four contiguous
// quarters, one per wave.  PAIRS: (key, id) pairs in the bucket's slice of the global buffers, digit = ((key - kmin) >> dshift) & 255
// (kmin: the bucket's smallest possible key).  !PAIRS: single words c = offset of the key in the bucket << 12 | position after the
// split (n <= 4096) in LDS, digit = (c >> dshift) & 255.  (The address spaces are resolved after inlining.)
template <int MODE, uint32_t POSBITS = 12>
__device__ __forceinline__ void bucket_pass(const uint32_t *ksrc, const uint32_t *vsrc, uint32_t *kdst, uint32_t *vdst, uint32_t n,
                                            uint32_t kmin, uint32_t dshift, uint32_t (*hw)[256], uint32_t *wave_tot) {
    constexpr bool PAIRS = MODE == 1, MAKE = MODE == 2;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t per = ((n + 3u) / 4u + 63u) & ~63u;
    const uint32_t beg = min(n, (uint32_t)wv * per), end = min(n, beg + per);
    auto digit = [&](uint32_t k) -> uint32_t { return ((((PAIRS || MAKE) ? k - kmin : k)) >> dshift) & 255u; };
#pragma unroll
    for (int i = 0; i < 4; i++) hw[wv][lane + 64 * i] = 0u;
    __builtin_amdgcn_wave_barrier();
    // (several loads in flight: a bucket that goes through global memory reads its keys from L2, a non-unrolled loop would pay
    //  the latency once per 64 keys)
    for (uint32_t i0 = beg + lane; i0 < end; i0 += 64u * 8u) {
        uint32_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) kk[u] = i0 + 64u * u < end ? ksrc[i0 + 64u * u] : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++) if (i0 + 64u * u < end) atomicAdd(&hw[wv][digit(kk[u])], 1u);
    }
    __syncthreads();
    {
        // thread <-> digit: digit bases, then the cursor of every wave
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = hw[0][d], c1 = hw[1][d], c2 = hw[2][d], c3 = hw[3][d];
        uint32_t total;
        const uint32_t base = block_exclusive_scan(c0 + c1 + c2 + c3, wave_tot, total);
        hw[0][d] = base; hw[1][d] = base + c0; hw[2][d] = base + c0 + c1; hw[3][d] = base + c0 + c1 + c2;
    }
    __syncthreads();
    const uint64_t lt = lanemask_lt();
    uint32_t *cur = hw[wv];
    // software pipeline: the items of the next two batches are in flight while the current batch is ranked
    uint32_t k1 = beg + lane < end ? ksrc[beg + lane] : 0u, v1 = (PAIRS && beg + lane < end) ? vsrc[beg + lane] : 0u;
    uint32_t k2 = beg + 64 + lane < end ? ksrc[beg + 64 + lane] : 0u, v2 = (PAIRS && beg + 64 + lane < end) ? vsrc[beg + 64 + lane] : 0u;
    for (uint32_t base = beg; base < end; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < end;
        const uint32_t key = k1, val = v1;
        k1 = k2; v1 = v2;
        k2 = i + 128 < end ? ksrc[i + 128] : 0u;
        v2 = (PAIRS && i + 128 < end) ? vsrc[i + 128] : 0u;
        const uint32_t d = digit(key);
        uint64_t peers = w3d_ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = w3d_ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt);
        uint32_t pos = 0;
        if (valid) pos = cur[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) cur[d] = pos + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (valid) { kdst[pos] = MAKE ? (((key - kmin) << POSBITS) | i) : key; if (PAIRS) vdst[pos] = val; }
    }
    __syncthreads();
}

// one workgroup per bucket: sort by the offset from the bucket's lower bound, write the depth-ordered packed records
__global__ void __launch_bounds__(256)
depth_bucket_sort_kernel(uint32_t *__restrict__ keys_a, uint32_t *__restrict__ vals_a, uint32_t *__restrict__ keys_b,
                         uint32_t *__restrict__ vals_b, const uint4 *__restrict__ grid, const uint2 *__restrict__ brange,
                         const uint32_t *__restrict__ bstart,
                         const uint2 *__restrict__ rect, const uint4 *__restrict__ rect_mask, uint4 *__restrict__ rec,
                         uint2 *__restrict__ rec_mask, int cull) {
    __shared__ uint32_t lc[2][W3D_DB_CAP];
    __shared__ uint32_t hw[4][256];
    __shared__ uint32_t wave_tot[17];
    const uint32_t beg = bstart[blockIdx.x], n = bstart[blockIdx.x + 1] - beg;
    if (n == 0) return;
    // this bucket's interval of key offsets: its width decides how many 8-bit passes the bucket needs (a bucket of a densely
    // populated segment is narrow whatever the view's whole interval is)
    const uint2 rng = brange[blockIdx.x];
    const uint32_t klo = grid[0].x + rng.x;                                        // smallest key this bucket can hold
    const uint32_t width = rng.y;
    const uint32_t rbits = width > 1u ? 32u - (uint32_t)__builtin_clz(width - 1u) : 0u;
    const uint32_t npass = (rbits + 7u) / 8u;
    auto gather = [&](uint32_t g) -> uint4 {
        if (cull) return rect_mask[g];                   // one 16-B record per Gaussian: a single random line
        const uint2 rc = rect[g];
        return make_uint4(rc.x, rc.y, 0xFFFFFFFFu, 0xFFFFFFFFu);
    };
    // the records of the sorted bucket, four per thread at a time so that the four dependent chains (position -> id -> rect line)
    // are in flight together; src(i) = position of the i-th smallest item in the split's output
    auto emit_all = [&](const uint32_t *ks, const uint32_t *vs, auto src) {
        for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 1024) {
            uint32_t key[4], val[4];
            uint4 gg[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + 256u * u;
                if (i < n) { const uint32_t j = src(i); key[u] = ks[j]; val[u] = vs[j]; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + 256u * u < n) gg[u] = gather(val[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + 256u * u;
                if (i < n) {
                    rec[beg + i] = make_uint4(val[u], gg[u].x, gg[u].y, key[u]);          // (the sort key IS the view depth)
                    rec_mask[beg + i] = make_uint2(gg[u].z, gg[u].w);
                }
            }
        }
    };
    if (npass == 0) {
        emit_all(keys_a + beg, vals_a + beg, [](uint32_t i) { return i; });
    } else if (n <= W3D_DB_FAST && rbits <= 20u) {
        // The usual bucket (the grid aims at ~1 200 keys): every thread keeps its up to 8 (key, id) pairs in registers and asks
        // for their rect / mask lines BEFORE the sort — which lines are needed does not depend on the order, only where their
        // records go — so the random gathers are in flight during the LDS passes; the sorted order is then inverted in LDS
        // (source position -> rank) and every thread writes its own items' records to their ranks.  Two dependent memory
        // round trips (pairs, gathers) instead of four (keys; sorted positions -> pairs; gathers).
        constexpr int IT = W3D_DB_FAST / 256;
        uint32_t *a0 = &lc[0][0], *a1 = &lc[0][W3D_DB_FAST], *inv = &lc[1][0];
        uint32_t kk[IT], vv[IT];
        uint4 gg[IT];
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const uint32_t i = threadIdx.x + 256u * u;
            kk[u] = i < n ? keys_a[beg + i] : 0u;
            vv[u] = i < n ? vals_a[beg + i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const uint32_t i = threadIdx.x + 256u * u;
            if (i < n) a0[i] = ((kk[u] - klo) << 12) | i;
        }
#pragma unroll
        for (int u = 0; u < IT; u++) if (threadIdx.x + 256u * u < n) gg[u] = gather(vv[u]);
        __syncthreads();                                       // (waits for LDS traffic only: the gathers stay in flight)
        for (uint32_t p = 0; p < npass; p++) {
            bucket_pass<0>(a0, nullptr, a1, nullptr, n, 0u, 12u + 8u * p, hw, wave_tot);
            uint32_t *t = a0; a0 = a1; a1 = t;
        }
        for (uint32_t i = threadIdx.x; i < n; i += 256) inv[a0[i] & 4095u] = i;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const uint32_t i = threadIdx.x + 256u * u;
            if (i < n) {
                const uint32_t d = beg + inv[i];
                rec[d] = make_uint4(vv[u], gg[u].x, gg[u].y, kk[u]);          // (the sort key IS the view depth)
                rec_mask[d] = make_uint2(gg[u].z, gg[u].w);
            }
        }
    } else if (n <= W3D_DB_CAP && rbits <= 20u) {
        // in LDS, one word per item: (offset of the key in the bucket) << 12 | position after the split
        for (uint32_t i = threadIdx.x; i < n; i += 256) lc[0][i] = ((keys_a[beg + i] - klo) << 12) | i;
        __syncthreads();
        uint32_t s = 0;
        for (uint32_t p = 0; p < npass; p++, s ^= 1u) bucket_pass<0>(lc[s], nullptr, lc[s ^ 1u], nullptr, n, 0u, 12u + 8u * p, hw, wave_tot);
        const uint32_t *fin = lc[s];
        emit_all(keys_a + beg, vals_a + beg, [&](uint32_t i) { return fin[i] & 4095u; });
    } else if (n <= 2u * W3D_DB_CAP && npass == 2u && rbits <= 19u) {
        // up to twice the capacity, two passes: ONE LDS array holds the bucket.  Pass 0 reads the raw keys (global, twice: count and
        // scatter) and leaves the words offset << 13 | position sorted by the low digit in LDS; pass 1 ranks them by the high digit
        // and scatters the words into the bucket's slice of the second global buffer, from where the records are emitted — one
        // L2 round trip instead of the three of the global path (the densified benchmark scene: 140-230 such buckets per view)
        uint32_t *big = &lc[0][0];
        bucket_pass<2, 13>(keys_a + beg, nullptr, big, nullptr, n, klo, 0u, hw, wave_tot);
        bucket_pass<0>(big, nullptr, keys_b + beg, nullptr, n, 0u, 13u + 8u, hw, wave_tot);
        const uint32_t *fin = keys_b + beg;
        emit_all(keys_a + beg, vals_a + beg, [&](uint32_t i) { return fin[i] & 8191u; });
    } else {
        // a bucket beyond the LDS array (or offsets of 21+ bits): the same passes on (key, id) pairs through its slice of the two global buffers
        uint32_t *ks = keys_a + beg, *vs = vals_a + beg, *kd = keys_b + beg, *vd = vals_b + beg;
        for (uint32_t p = 0; p < npass; p++) {
            bucket_pass<1>(ks, vs, kd, vd, n, klo, 8u * p, hw, wave_tot);
            uint32_t *t = ks; ks = kd; kd = t;
            t = vs; vs = vd; vd = t;
        }
        emit_all(ks, vs, [](uint32_t i) { return i; });
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
#ifndef W3D_WALK_SMALL_SHARED
#define W3D_WALK_SMALL_SHARED 8
#endif
#define W3D_WALK_QUEUE 128

#ifndef W3D_WW
#define W3D_WW 4          // (chunk, band) waves per workgroup of the walk: the 4 band-waves of a chunk share its records in L1
                          // (measured fill: 1 wave 0.255 ms, 2 -> 0.198, 4 -> 0.172, 8 -> 0.202, 16 -> 0.234)
#endif
template <int MODE, bool CULL, int SMALL>
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
#ifndef W3D_WALK_NB
#define W3D_WALK_NB 2     // (measured, untrained / densified scene, count + fill in us: 1 -> 28.0 + 72.2 / 71.5 + 138.1, 2 -> 27.7 + 72.8 / 72.0 + 138.9,
#endif                   //  3 -> 31.4 + 73.5 / 74.0 + 165.4, 4 -> 32.1 + 73.7 / 74.7 + 156.3, 8 -> 36.7 + 81.3 / 83.3 + 183.6)
    constexpr int NB = W3D_WALK_NB;
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
                kind = __popcll(rm) > SMALL ? 1u : 0u;
            } else {
                kind = 2u;
            }
        }
        // floor(k / w) == (k * magic) >> 16 for k < 64, w <= 64 with magic = floor(65536 / w) + 1; through the float
        // reciprocal (65536 / w is either an integer, where rcp is exact enough, or >= 1/64 away from one)
        const uint32_t magic = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)max(w, 1u)) + 0.004f) + 1u;
        const uint4 eb = make_uint4((uint32_t)rm, (uint32_t)(rm >> 32), er.w, er.z);     // (layout the whole-wave part reads)
        const uint64_t coop = w3d_ballot(lane < nq && kind != 0u);
        // lane-parallel part: the (at most SMALL) tiles of this lane's record, derived once and kept in registers
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        static_assert(SMALL % 8 == 0, "the fill pass reads the slots eight at a time");
        uint32_t tls[SMALL];
        uint32_t kmax = 0;                                         // wave-uniform: slots [0, kmax) hold a tile for some lane
#pragma unroll
        for (int i = 0; i < SMALL; i++) tls[i] = NONE;
        {
            uint64_t m = (lane < nq && kind == 0u) ? rm : 0ull;
            // tile of rect slot k (row ty = k / w): (miny + ty - y0) * gx + minx + k - ty * w = base + k + ty * (gx - w);
            // base may wrap below zero for a rect that starts above the band — the sum is taken modulo 2^32
            const uint32_t base = (miny - y0) * gx + minx, rowskip = gx - w;
#pragma unroll
            for (int i = 0; i < SMALL; i++) {
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
            for (int i = 0; i < SMALL; i++) {
                if ((uint32_t)i >= kmax) break;
                if (tls[i] != NONE) atomicAdd(&h32[tls[i] >> 1], 1u << ((tls[i] & 1u) * 16u));
            }
            traverse_coop([&](uint32_t tl, uint32_t, uint32_t) { atomicAdd(&h32[tl >> 1], 1u << ((tl & 1u) * 16u)); });
        } else {
            const unsigned long long mybit = 1ull << lane;
#pragma unroll
            for (int i = 0; i < SMALL; i++) {
                if ((uint32_t)i >= kmax) break;
                if (tls[i] != NONE) atomicOr(&bm[tls[i]], mybit);
            }
            traverse_coop([&](uint32_t tl, uint32_t src, uint32_t) { atomicOr(&bm[tl], 1ull << src); });
            __builtin_amdgcn_wave_barrier();
            // all cursor / bitmap reads of the batch are issued back to back, then the list entries are written
#pragma unroll
            for (int i0 = 0; i0 < SMALL; i0 += 8) {
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
                 uint32_t *__restrict__ counters, uint32_t share_code, uint32_t *__restrict__ tile_walk, uint32_t n_tiles) {
    // (the blend forward's walk lengths: the part-waves of a split tile report theirs with atomicMax, w3d_render.hip)
    for (uint32_t t = threadIdx.x; t < n_tiles; t += 1024u) tile_walk[t] = 0u;
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
    for (; c + 32 <= c1; c += 32) {       // 32 count loads in flight per step of the running sum
        uint32_t v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = cnt[(size_t)(c + i) * T + t];
#pragma unroll
        for (int i = 0; i < 32; i++) { off[(size_t)(c + i) * T + t] = run; run += v[i]; }
    }
    for (; c + 8 <= c1; c += 8) {
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

// stable depth sort of the visible Gaussians + depth-ordered packed records: five launches (see "depth sort" above)
int w3d_launch_depth_sort(const W3DLayout &L, const w3d_view &v, char *state, char *scratch, hipStream_t stream) {
    uint32_t *counters = reinterpret_cast<uint32_t *>(state + L.o_counters);
    uint32_t *keys[2] = {reinterpret_cast<uint32_t *>(scratch + L.s_keys0), reinterpret_cast<uint32_t *>(scratch + L.s_keys1)};
    uint32_t *vals[2] = {reinterpret_cast<uint32_t *>(scratch + L.s_vals0), reinterpret_cast<uint32_t *>(scratch + L.s_vals1)};
    uint32_t *hist = reinterpret_cast<uint32_t *>(scratch + L.s_hist);
    uint32_t *rowtot = reinterpret_cast<uint32_t *>(scratch + L.s_rowtot);
    uint32_t *bstart = reinterpret_cast<uint32_t *>(scratch + L.s_bstart);
    if (L.P == 0) return W3D_OK;
    const uint32_t n = (uint32_t)L.P, runs = L.sort_waves, blocks = (runs + 3) / 4, pitch = blocks * 4;
    W3D_PROF("depth_sort", stream);
    uint4 *grid = reinterpret_cast<uint4 *>(scratch + L.s_grid);
    uint2 *brange = reinterpret_cast<uint2 *>(scratch + L.s_grid + 257 * 16);        // (the table region is sized for 256 segments)
    hipLaunchKernelGGL(depth_grid_kernel, dim3(1), dim3(W3D_DG_THREADS), 0, stream, reinterpret_cast<const uint4 *>(scratch + L.s_minmax),
                       (n + W3D_PRE_BLOCK - 1) / W3D_PRE_BLOCK, grid, brange, counters);
    hipLaunchKernelGGL(depth_bucket_hist_kernel, dim3(blocks), dim3(256), 0, stream, keys[0], n, L.sort_items, runs, pitch, hist, (const uint4 *)grid);
    W3D_LAUNCH_CHECK(v.debug, stream);
    hipLaunchKernelGGL(depth_bucket_scan_kernel, dim3(W3D_DB_BINS), dim3(256), 0, stream, hist, pitch, rowtot);
    W3D_LAUNCH_CHECK(v.debug, stream);
    hipLaunchKernelGGL(depth_bucket_scatter_kernel, dim3(blocks), dim3(256), 0, stream, keys[0], keys[1], vals[1], n, L.sort_items, runs,
                       pitch, hist, rowtot, counters, bstart, (const uint4 *)grid);
    W3D_LAUNCH_CHECK(v.debug, stream);
    // (keys[0] / vals[0] — the unsorted keys are not needed any more — serve as the second buffer of a bucket that is too large for LDS)
    hipLaunchKernelGGL(depth_bucket_sort_kernel, dim3(W3D_DB_BINS), dim3(256), 0, stream, keys[1], vals[1], keys[0], vals[0],
                       (const uint4 *)grid, (const uint2 *)brange, bstart, reinterpret_cast<const uint2 *>(state + L.o_rect), reinterpret_cast<const uint4 *>(state + L.o_tile_mask),
                       reinterpret_cast<uint4 *>(scratch + L.s_rec), reinterpret_cast<uint2 *>(scratch + L.s_rec_mask), (int)v.tile_cull);
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
    // records binned one per lane may touch at most SMALL cells of the band (larger ones go through the whole-wave part): 16 on
    // the tile grid, 8 on a shared list grid, whose cells are 2-4 tiles (measured on the untrained scene's 32x32 cells: count
    // 27.8 -> 25.6 us, fill 72.7 -> 70.9; on the densified scene's one list per tile 8 costs +14 us)
    const bool shared = (L.lsx | L.lsy) != 0;
#define W3D_WALK_LAUNCH(CULL_, SMALL_)                                                                                              \
    hipLaunchKernelGGL((chunk_walk_kernel<MODE, CULL_, SMALL_>), grid, dim3(64 * W3D_WW), lds, stream, rec, rmask, counters, L.chunk, L.C, \
                       (uint32_t)L.LT, (uint32_t)L.lgx, (uint32_t)L.lgy, bands.rows, cnt, off, point_list, capacity, wave_bytes)
    if (v.tile_cull) { if (shared) W3D_WALK_LAUNCH(true, W3D_WALK_SMALL_SHARED); else W3D_WALK_LAUNCH(true, W3D_WALK_SMALL); }
    else             { if (shared) W3D_WALK_LAUNCH(false, W3D_WALK_SMALL_SHARED); else W3D_WALK_LAUNCH(false, W3D_WALK_SMALL); }
#undef W3D_WALK_LAUNCH
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
                       (uint32_t)L.lsx | ((uint32_t)L.lsy << 8), reinterpret_cast<uint32_t *>(state + L.o_tile_walk), (uint32_t)L.T);
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
