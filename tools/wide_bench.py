#!/usr/bin/env python3
"""MFMA utilisation of the temporal-conv AND graph-conv kernels, and of whole st_gcn blocks, on the WIDE layers north_star names (C = 128 / 256: the blocks
the reference leaves commented out, classifier_stgcn_real_only/net/classifier.py:42-49).  Shapes follow the
original ST-GCN schedule at T_in = 240: 128 channels at T = 60 (stride-2 entry from T = 120), 256 at T = 30.
The graph conv of a block maps Ci -> C (64 -> 128 and 128 -> 256 with a stride-2 residual conv at the entries).
usage: python tools/wide_bench.py [--n 2048] [--reps 3] [--no-blocks]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib, ops  # noqa: E402
from step_amd.net.utils.graph import Graph  # noqa: E402

PEAK = 157.3e12


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
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-blocks", action="store_true")
    a = ap.parse_args()
    d = "cuda:0"
    g = torch.Generator().manual_seed(7)
    n = a.n
    for c, s, tin in [(128, 2, 120), (128, 1, 60), (256, 2, 60), (256, 1, 30)]:
        tout = (tin - 1) // s + 1
        y = torch.randn(n, c, tin, 16, generator=g).to(d)
        du = torch.randn(n, c, tout, 16, generator=g).to(d)
        w = (torch.randn(c, c, 75, 1, generator=g) * 0.02).to(d)
        b = torch.zeros(c, device=d)
        sc, sh = torch.ones(c, device=d), torch.zeros(c, device=d)
        wp = ops.tcn_pack(w, c, c, 75, s, 0)
        wpd = ops.tcn_pack(w, c, c, 75, s, 2)
        calls = {
            "tcn_fwd": lambda: ops.tcn_conv(y, sc, sh, True, wp, b, None, None, None, None, None, False,
                                            n, c, c, tin, tout, 75, s, 0),
            "tcn_dgrad": lambda: ops.tcn_conv(du, None, None, False, wpd, None, None, None, None, None, None, False,
                                              n, c, c, tin, tout, 75, s, 2),
            "tcn_wgrad": lambda: ops.tcn_wgrad(du, y, sc, sh, True, n, c, c, tin, tout, 75, s, False, w.shape),
        }
        fl = 2.0 * c * c * 75 * tout * 16 * n
        for name, fn in calls.items():
            ms = timeit(fn, a.reps)
            print(json.dumps({"kernel": name, "C": c, "stride": s, "T_in": tin, "N": n, "ms": round(ms, 3),
                              "tflops": round(fl / ms / 1e9, 1), "frac_fp32_mfma": round(fl / ms / 1e9 / (PEAK / 1e12), 4)}),
                  flush=True)
        del y, du

    # graph conv of the same layers: (Ci, C, T_in, stride of the residual conv or 0)
    A = torch.tensor(Graph(strategy="spatial").A, dtype=torch.float32).to(d)
    nnz = ops.a_nnz_max(A)
    for ci, c, tin, rs in [(64, 128, 120, 2), (128, 128, 60, 0), (128, 256, 60, 2), (256, 256, 30, 0)]:
        to = (tin - 1) // rs + 1 if rs else tin
        x = torch.randn(n, ci, tin, 16, generator=g).to(d)
        w = (torch.randn(3 * c, ci, generator=g) * 0.05).to(d)
        b = torch.zeros(3 * c, device=d)
        y = torch.empty(n, c, tin, 16, device=d)
        dy = torch.randn(n, c, tin, 16, generator=g).to(d)
        rw = (torch.randn(c, ci, generator=g) * 0.05).to(d) if rs else None
        rb = torch.zeros(c, device=d) if rs else None
        r = torch.empty(n, c, to, 16, device=d) if rs else None
        dr = torch.randn(n, c, to, 16, generator=g).to(d) if rs else None
        dx, dw, db, dA = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b), torch.empty_like(A)
        drw = torch.empty_like(rw) if rs else None
        drb = torch.empty(c, device=d) if rs else None
        nb = _lib.query("step_gcn_bwd_ws_bytes", n, ci, c, 3, tin)
        ws = torch.empty(nb, dtype=torch.uint8, device=d)
        st = rs if rs else 1
        fl = (2.0 * 3 * ci * c * tin * 16 + (2.0 * ci * c * to * 16 if rs else 0.0)) * n
        calls = {
            "gcn_fwd": (lambda: _lib.call("step_gcn_fwd", x, None, None, w, b, 0, A, nnz, y, rw, rb, 0, r, n, ci, c, 3, tin, st), fl),
            "gcn_bwd": (lambda: _lib.call("step_gcn_bwd", dy, dr, x, None, None, w, b, 0, A, nnz, rw, 0, dx, dw, db, dA, drw,
                                          drb, n, ci, c, 3, tin, st, ws, nb), 2 * fl),
        }
        for name, (fn, f) in calls.items():
            ms = timeit(fn, a.reps)
            print(json.dumps({"kernel": name, "Ci": ci, "C": c, "res_stride": rs, "T_in": tin, "N": n, "ms": round(ms, 3),
                              "tflops": round(f / ms / 1e9, 1), "frac_fp32_mfma": round(f / ms / 1e9 / (PEAK / 1e12), 4)}),
                  flush=True)
        del x, y, dy, dx, r, dr, ws

    if a.no_blocks:
        return
    # whole st_gcn blocks (train-mode forward + backward), the 64 -> 128 -> 128 -> 256 -> 256 stack of the original
    # ST-GCN schedule; FLOPs = 3 x forward (gcn + residual conv + temporal conv)
    from step_amd.net.blocks import st_gcn
    for ci, c, s, tin in [(64, 128, 2, 120), (128, 128, 1, 60), (128, 256, 2, 60), (256, 256, 1, 30)]:
        tout = (tin - 1) // s + 1
        torch.manual_seed(0)
        blk = st_gcn(ci, c, (75, 3), s).to(d).train()
        blk.a_nnz_max = nnz
        x = torch.randn(n, ci, tin, 16, generator=g).to(d).requires_grad_(True)
        gy = torch.randn(n, c, tout, 16, generator=g).to(d)

        def step():
            out, _ = blk(x, A)
            out.backward(gy)
            blk.zero_grad(set_to_none=True)
            x.grad = None

        ms = timeit(step, a.reps)
        fl = 3.0 * n * (2.0 * 3 * ci * c * tin * 16 + 2.0 * c * c * 75 * tout * 16 + (2.0 * ci * c * tout * 16 if (ci != c or s != 1) else 0.0))
        print(json.dumps({"kernel": "st_gcn block fwd+bwd", "Ci": ci, "C": c, "stride": s, "T_in": tin, "N": n, "ms": round(ms, 3),
                          "tflops": round(fl / ms / 1e9, 1), "frac_fp32_mfma": round(fl / ms / 1e9 / (PEAK / 1e12), 4)}), flush=True)
        del x, gy, blk


if __name__ == "__main__":
    main()
