"""exchange leg (N > 1): the step's collectives alone and the replica bit-identity check."""
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def exchange_bandwidth(model, world, dev, rows=0):
    """The step's gradient collectives alone (N > 1): achieved bus bandwidth per GPU.
    Low-rank form — all-gather of the (P,3) colour gradients: every rank receives (world-1)*12P bytes; all-reduce of the
    11-float geometry span: ring model 2*(world-1)/world * 44P bytes per rank.  Sparse form (rows > 0: the largest per-view
    row count of the last step) — all-gather of `rows` 64-byte rows per rank: (world-1)*64*rows bytes received."""
    P = model.num_points
    d = torch.randn(P, 3, device=dev)
    d_all = torch.empty(world, P, 3, device=dev)
    geo = torch.randn(11 * P, device=dev)
    out = {}
    cases = [("all_gather_dcolor", lambda: dist.all_gather_into_tensor(d_all.view(-1), d.view(-1)), (world - 1) * 12.0 * P),
             ("all_reduce_geometry", lambda: dist.all_reduce(geo), 2.0 * (world - 1) / world * 44.0 * P)]
    if rows > 0:
        r_own = torch.randn(rows, 16, device=dev)
        r_all = torch.empty(world, rows, 16, device=dev)
        cases.append(("all_gather_rows", lambda: dist.all_gather_into_tensor(r_all.view(-1), r_own.view(-1)), (world - 1) * 64.0 * rows))
    for name, fn, nbytes in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / 10], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        out[name] = {"ms": round(1e3 * float(dt), 4), "bus_GBps_per_gpu": round(nbytes / float(dt) / 1e9, 1)}
    return out


def replicas_identical(model, world, dev):
    """Every rank must hold bit-identical parameters (nothing re-synchronises them): compare a checksum of the bits."""
    bits = model.flat.detach().view(torch.int32).to(torch.int64)
    s = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()])
    allsums = [torch.zeros_like(s) for _ in range(world)]
    dist.all_gather(allsums, s)
    return all(bool(torch.equal(allsums[0], x)) for x in allsums)

