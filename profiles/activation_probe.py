#!/usr/bin/env python3
"""Which arithmetic do torch's device kernels for F.normalize / sigmoid / exp perform?  (the fused raw path
must reproduce them bit for bit.)  Candidates are emulated with torch's own elementwise kernels (each op individually rounded;
the fma chain through float64: exact product, one rounding to float64, then to float32) on 4 M random rows over 60 binades and
compared bit for bit with torch's result.  Prints one JSON line."""
import json

import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n = 4_000_000
x = (torch.randn(n, 4, generator=g) * torch.exp2(torch.randint(-30, 30, (n, 1), generator=g).float())).to(dev)
ref = F.normalize(x)
s = [x[:, i] * x[:, i] for i in range(4)]
xd = x.double()
cands = {
    "((s0+s1)+s2)+s3": ((s[0] + s[1]) + s[2]) + s[3],
    "(s0+s1)+(s2+s3)": (s[0] + s[1]) + (s[2] + s[3]),
    "fma chain": (xd[:, 3] * xd[:, 3] + (xd[:, 2] * xd[:, 2] + (xd[:, 1] * xd[:, 1] + s[0].double()).float().double()).float().double()).float(),
    "float64 sum": (xd * xd).sum(1).float(),
}
out = {"normalize": {}}
for name, ss in cands.items():
    d = torch.sqrt(ss).clamp_min(1e-12)[:, None]
    out["normalize"][name + " ; x / d"] = int((torch.ne((x / d).view(torch.int32), ref.view(torch.int32))).any(1).sum())
    out["normalize"][name + " ; x * (1/d)"] = int((torch.ne((x * (1.0 / d)).view(torch.int32), ref.view(torch.int32))).any(1).sum())
out["normalize"]["norm == linalg.vector_norm"] = int((torch.linalg.vector_norm(x, 2, 1) != torch.sqrt(cands["((s0+s1)+s2)+s3"])).sum())
t = torch.randn(n, generator=g).to(dev) * 6
out["sigmoid"] = {"1/(1+exp(-x))": int((torch.sigmoid(t).view(torch.int32) != (1.0 / (1.0 + torch.exp(-t))).view(torch.int32)).sum())}
out["rows"] = n
print(json.dumps(out))
