"""Host (CPU-tensor) twins of the HIP-backed operations — a registry that is EMPTY in the product.

Every operation of this package that has a kernel behind it (Adam, the photometric loss, the densification statistics, the
row compaction, the gradient-row pack / apply, the SH update of the low-rank exchange) refuses CPU tensors, as the rasterizer
does: `twin(name, what)` raises unless somebody registered a stand-in.  Nobody in the product does.  tests/cpu_twins.py
registers torch restatements of the same formulas so that the host LOGIC — block bookkeeping, schedules, the view-parallel
protocol over gloo — can be unit-tested on a box without a GPU (`-m "not gpu"`); the GPU tests then hold the kernels against
the same restatements.  (Rounds 1-5 kept those formulas inside the product classes as `if not x.is_cuda:` branches.)"""
_TWINS = {}


def register(name, fn):
    _TWINS[name] = fn


def twin(name, what):
    fn = _TWINS.get(name)
    if fn is None:
        raise RuntimeError(f"{what} needs GPU tensors: this package has no CPU path (the kernels live in libw3d_hip.so; the "
                           f"host-logic tests register torch stand-ins through tests/cpu_twins.py)")
    return fn
