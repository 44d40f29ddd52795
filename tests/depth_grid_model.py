"""The depth sort's bucket grid (w3d_binning.hip depth_grid_kernel) restated in integer numpy, for the tests and probes that need to
know which in-bucket path a scene takes: the table the kernel builds from the preprocess workgroups' {min, max, two keys} entries, then every key's bucket."""
import numpy as np

SEGS, BINS, PRE_BLOCK, HOLD = 64, 1024, 256, 8 * 1024
INVALID = 0xFFFFFFFF


def grid_buckets(keys, return_buckets=False):
    """keys: (P,) uint32 depth keys in storage order, 0xFFFFFFFF = culled.  Returns (populations[1024], widths[1024], nbuckets)
    and, with return_buckets, the bucket of every visible key (in storage order) as a fourth item."""
    keys = np.asarray(keys, dtype=np.int64)
    P = keys.shape[0]
    nb = (P + PRE_BLOCK - 1) // PRE_BLOCK
    vis = keys != INVALID
    kv = keys[vis]
    if kv.size == 0:
        out = (np.zeros(BINS, np.int64), np.zeros(BINS, np.int64), 0)
        return out + (np.zeros(0, np.int64),) if return_buckets else out
    kmin, kmax = int(kv.min()), int(kv.max())
    span = kmax - kmin
    if span < BINS:
        out = (np.bincount(kv - kmin, minlength=BINS), (np.arange(BINS) <= span).astype(np.int64), span + 1)
        return out + (kv - kmin,) if return_buckets else out
    mseg = (1 << (32 + 6)) // (span + 1)
    stride = (nb + HOLD - 1) // HOLD
    ent = np.arange(0, nb, stride) * PRE_BLOCK
    samp = np.concatenate([keys[np.minimum(ent + 64, P - 1)][ent + 64 < P], keys[np.minimum(ent + 192, P - 1)][ent + 192 < P]])
    samp = samp[samp != INVALID]
    c = np.bincount(((samp - kmin) * mseg) >> 32, minlength=SEGS)
    start = ((np.arange(SEGS, dtype=object) << 32) + mseg - 1) // mseg
    start = np.array(start, dtype=np.int64)
    end = np.append(start[1:], span + 1)
    width = end - start
    cnt = np.minimum(1 + ((BINS - SEGS) * c) // max(int(c.sum()), 1), width)
    first = np.cumsum(cnt) - cnt
    slope = np.where(cnt == width, 0, (cnt.astype(object) << 32) // width.astype(object)).astype(np.int64)
    x = kv - kmin
    seg = (x * mseg) >> 32
    d = x - start[seg]
    loc = np.where(slope[seg] > 0, (d * slope[seg]) >> 32, d)           # (d < 2^32 / 64, slope < 2^32: the product fits int64)
    b = first[seg] + np.minimum(loc, cnt[seg] - 1)
    pop = np.bincount(b, minlength=BINS)
    # bucket widths: lo(l) = ceil(l * 2^32 / slope)
    widths = np.zeros(BINS, np.int64)
    for s in range(SEGS):
        l = np.arange(int(cnt[s]) + 1, dtype=object)
        lo = l if slope[s] == 0 else ((l << 32) + int(slope[s]) - 1) // int(slope[s])
        lo = np.array(lo, dtype=np.int64)
        lo[-1] = width[s]
        widths[first[s]:first[s] + cnt[s]] = np.diff(lo)
    out = (pop, widths, int(cnt.sum()))
    return out + (b,) if return_buckets else out


def summary(pop, widths):
    rbits = np.where(widths > 1, np.ceil(np.log2(np.maximum(widths, 2))).astype(np.int64), 0)
    npass = (rbits + 7) // 8
    V = max(int(pop.sum()), 1)
    return {"max": int(pop.max()), "over_4096": int((pop > 4096).sum()), "over_8192": int((pop > 8192).sum()), "empty": int((pop == 0).sum()),
            "keys_by_passes": {str(p): int(pop[npass == p].sum()) for p in sorted(set(npass.tolist()))},
            "key_passes_per_key": round(float((pop * npass).sum()) / V, 3)}


def bucket_paths(pop, widths):
    """How many non-empty buckets take which branch of depth_bucket_sort_kernel: `direct` (one key value per bucket: no pass),
    `fast` (<= 2048 keys: pairs in registers), `lds4096` (<= 4096: records emitted from the sorted positions), `mid` (<= 8192 keys,
    two passes: one LDS array + one trip through the second global buffer), `global` (everything else)."""
    rbits = np.where(widths > 1, np.ceil(np.log2(np.maximum(widths, 2))).astype(np.int64), 0)
    npass = (rbits + 7) // 8
    nz = pop > 0
    direct = nz & (npass == 0)
    fast = nz & ~direct & (pop <= 2048) & (rbits <= 20)
    lds = nz & ~direct & ~fast & (pop <= 4096) & (rbits <= 20)
    mid = nz & ~direct & ~fast & ~lds & (pop <= 8192) & (npass == 2) & (rbits <= 19)
    glob = nz & ~direct & ~fast & ~lds & ~mid
    return {"direct": int(direct.sum()), "fast": int(fast.sum()), "lds4096": int(lds.sum()), "mid": int(mid.sum()), "global": int(glob.sum())}
