#!/usr/bin/env python3
"""Micro-benchmarks of the HBM-bound kernels around the temporal conv (graph conv fwd/bwd, BN
reduce/apply, affine), HIP events on the launch stream, with the algorithmic bytes of each call.
usage: python tools/gbench.py [--n 4096] [--block 0|1|2] [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib, ops  # noqa: E402
from step_amd.net.utils.graph import Graph  # noqa: E402

BLOCKS = {0: (3, 32, 240, 1, True), 1: (32, 64, 240, 2, True), 2: (64, 64, 120, 1, False)}


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--block", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    d = "cuda:0"
    N = a.n
    Ci, Co, T, st, resconv = BLOCKS[a.block]
    To = (T - 1) // st + 1
    K = 3
    g = torch.Generator().manual_seed(3)
    A = torch.tensor(Graph(strategy="spatial").A, dtype=torch.float32).to(d)
    nnz = ops.a_nnz_max(A)
    x = torch.randn(N, Ci, T, 16, generator=g).to(d)
    w = (torch.randn(K * Co, Ci, generator=g) * 0.1).to(d)
    b = torch.zeros(K * Co, device=d)
    y = torch.empty(N, Co, T, 16, device=d)
    dy = torch.randn(N, Co, T, 16, generator=g).to(d)
    rw = (torch.randn(Co, Ci, generator=g) * 0.1).to(d) if resconv else None
    rb = torch.zeros(Co, device=d) if resconv else None
    r = torch.empty(N, Co, To, 16, device=d) if resconv else None
    dr = torch.randn(N, Co, To, 16, generator=g).to(d) if resconv else None
    dx = torch.empty_like(x)
    dw, db, dA = torch.empty_like(w), torch.empty_like(b), torch.empty_like(A)
    drw = torch.empty_like(rw) if resconv else None
    drb = torch.empty(Co, device=d) if resconv else None
    nb = _lib.query("step_gcn_bwd_ws_bytes", N, Ci, Co, K, T)
    ws = torch.empty(nb, dtype=torch.uint8, device=d)
    el = 4.0
    ysums = torch.empty(2 * Co + 1, dtype=torch.float64, device=d)
    rsums = torch.empty(2 * Co + 1, dtype=torch.float64, device=d) if resconv else None
    snb = _lib.query("step_gcn_fwd_stats_ws_bytes", Co)
    sws = torch.empty(snb, dtype=torch.uint8, device=d)
    calls = {
        "gcn_fwd": (lambda: _lib.call("step_gcn_fwd", x, None, None, w, b, 0, A, nnz, y, rw, rb, 0, r,
                                      N, Ci, Co, K, T, st),
                    el * (x.numel() + y.numel() + (r.numel() if resconv else 0))),
        "gcn_fwd_stats": (lambda: _lib.call("step_gcn_fwd_stats", x, None, None, w, b, 0, A, nnz, y, rw, rb, 0, r,
                                            N, Ci, Co, K, T, st, ysums, rsums, sws, snb),
                          el * (x.numel() + y.numel() + (r.numel() if resconv else 0))),
        "gcn_fwd_split": (lambda: _lib.call("step_gcn_fwd_split", x, None, None, w, b, 0, A, nnz, y, rw, rb, 0, r,
                                            N, Ci, Co, K, T, st),
                          el * (x.numel() + y.numel() + (r.numel() if resconv else 0))),
        "gcn_bwd(all)": (lambda: _lib.call("step_gcn_bwd", dy, dr, x, None, None, w, b, 0, A, nnz, rw, 0, dx, dw,
                                           db, dA, drw, drb, N, Ci, Co, K, T, st, ws, nb),
                         el * (2 * (x.numel() + dy.numel() + (dr.numel() if resconv else 0)) + dx.numel())),
    }
    if not resconv and Ci == Co:        # identity residual: dx += dout * [out > 0] in the same launch
        idg, ido = torch.randn(N, Ci, T, 16, generator=g).to(d), torch.randn(N, Ci, T, 16, generator=g).to(d)
        calls["gcn_bwd_idres"] = (lambda: _lib.call("step_gcn_bwd_idres", dy, dr, x, None, None, w, b, 0, A, nnz, rw, 0,
                                                     idg, ido, dx, dw, db, dA, drw, drb, N, Ci, Co, K, T, st, ws, nb),
                                  el * (2 * (x.numel() + dy.numel()) + 3 * dx.numel()))
    idg_s = torch.randn(N, Ci, T, 16, generator=g).to(d) if (not resconv and Ci == Co) else None
    ido_s = torch.randn(N, Ci, T, 16, generator=g).to(d) if (not resconv and Ci == Co) else None
    calls["gcn_bwd_split"] = (lambda: _lib.call("step_gcn_bwd_split", dy, dr, x, None, None, w, b, 0, A, nnz, rw, 0,
                                                 idg_s, ido_s, dx, dw, db, dA, drw, drb, N, Ci, Co, K, T, st, ws, nb),
                              el * (2 * (x.numel() + dy.numel() + (dr.numel() if resconv else 0)) + dx.numel()
                                    + (2 * dx.numel() if idg_s is not None else 0)))
    # BN / elementwise kernels on the block's (N, Co, T, 16) tensor
    bnw, bnb = torch.ones(Co, device=d), torch.zeros(Co, device=d)
    rm, rv = torch.zeros(Co, device=d), torch.ones(Co, device=d)
    mean, invstd, sc, sh = ops.bn_train_prepare(dy, 0, bnw, bnb, rm, rv)
    calls["bn_train_prepare"] = (lambda: ops.bn_train_prepare(dy, 0, bnw, bnb, rm, rv), el * dy.numel())
    calls["affine_act"] = (lambda: ops.affine_act(dy, sc, sh, None, None, None, True, 0), el * 2 * dy.numel())
    calls["bn_backward(relu mask)"] = (
        lambda: ops.bn_backward(dy, y, None, sc, sh, 2, mean, invstd, bnw, True, 0),
        el * (2 * 2 * dy.numel() + dy.numel()))
    for name, (fn, nbytes) in calls.items():
        if a.only and not (name + " ").startswith(a.only):
            continue
        ms = timeit(fn, a.reps)
        print("%-24s blk%d N=%d: %8.3f ms  %7.1f GB/s algorithmic (%.2f of 8 TB/s)" %
              (name, a.block, N, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)


if __name__ == "__main__":
    main()
