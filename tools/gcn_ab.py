#!/usr/bin/env python3
"""A/B of the two forward graph-conv kernels on the same inputs (STEP_GCN_FWD_V1=1 selects the first form)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib, ops  # noqa: E402
from step_amd.net.utils.graph import Graph  # noqa: E402

d = "cuda:0"
A0 = torch.tensor(Graph(strategy="spatial").A, dtype=torch.float32)
g = torch.Generator().manual_seed(1)
worst = 0.0
for (N, Ci, Co, T, st, res, tr, dense) in [(5, 32, 64, 75, 2, True, False, False), (5, 64, 64, 75, 1, False, False, False),
                                            (3, 64, 64, 37, 2, True, False, False), (4, 32, 32, 13, 1, False, True, False),
                                            (4, 64, 32, 29, 1, True, False, False), (4, 64, 3, 75, 1, True, True, False),
                                            (4, 32, 64, 75, 1, True, True, True), (600, 32, 64, 240, 2, True, False, False),
                                            (8, 64, 64, 38, 1, False, False, False), (8, 32, 64, 75, 2, True, False, False),
                                            (8, 64, 64, 8, 1, False, False, False), (8, 64, 64, 41, 1, False, False, False),
                                            (8, 64, 64, 40, 1, False, False, False), (700, 64, 64, 38, 1, False, False, False)]:
    A = (A0 * (torch.rand(3, 16, 16, generator=g) + 0.5) if not dense else torch.randn(3, 16, 16, generator=g)).to(d)
    nnz = ops.a_nnz_max(A)
    To = (T - 1) // st + 1
    x = torch.randn(N, Ci, T, 16, generator=g).to(d)
    w = (torch.randn(Ci, 3 * Co, generator=g) if tr else torch.randn(3 * Co, Ci, generator=g)).to(d)
    b = torch.randn(3 * Co, generator=g).to(d)
    rw = ((torch.randn(Ci, Co, generator=g) if tr else torch.randn(Co, Ci, generator=g)).to(d)) if res else None
    rb = torch.randn(Co, generator=g).to(d) if res else None
    outs = []
    for v1 in (True, False):
        if v1:
            os.environ["STEP_GCN_FWD_V1"] = "1"
        else:
            os.environ.pop("STEP_GCN_FWD_V1", None)
        y = torch.full((N, Co, T, 16), float("nan"), device=d)
        r = torch.full((N, Co, To, 16), float("nan"), device=d) if res else None
        _lib.call("step_gcn_fwd", x, None, None, w, b, int(tr), A, nnz, y, rw, rb, int(tr), r, N, Ci, Co, 3, T, st)
        torch.cuda.synchronize()
        outs.append((y, r))
    ey = float((outs[0][0] - outs[1][0]).abs().max() / outs[0][0].abs().max())
    er = float((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max()) if res else 0.0
    nan = bool(torch.isnan(outs[1][0]).any()) or (res and bool(torch.isnan(outs[1][1]).any()))
    print("N=%d %d->%d T=%d s=%d res=%s transposed=%s denseA=%s: y rel diff %.2e, res rel diff %.2e, nan=%s"
          % (N, Ci, Co, T, st, res, tr, dense, ey, er, nan), flush=True)
    worst = max(worst, ey, er)
    assert not nan
assert worst < 2e-6, worst
print("ok")
