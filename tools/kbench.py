#!/usr/bin/env python3
"""Kernel micro-benchmarks (HIP events on the launch stream) for the temporal-conv kernels.
usage: python tools/kbench.py [fwd|dgrad|wgrad|all] [--n 4096] [--c 64] [--stride 1] [--t 120] [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--c", type=int, default=64)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--t", type=int, default=120)
    ap.add_argument("--kt", type=int, default=75)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--bf16", action="store_true", help="bf16-operand MFMA kernels (fp32 tensors)")
    ap.add_argument("--split", action="store_true", help="wgrad on split fp32 operands (three bf16 MFMAs per product)")
    ap.add_argument("--noact", action="store_true", help="forward conv without BN1 + ReLU on load (A/B of the staging work)")
    a = ap.parse_args()
    d = "cuda:0"
    n, c, s, tin, kt = a.n, a.c, a.stride, a.t, a.kt
    tout = (tin - 1) // s + 1
    g = torch.Generator().manual_seed(7)
    y = torch.randn(n, c, tin, 16, generator=g).to(d)
    du = torch.randn(n, c, tout, 16, generator=g).to(d)
    w = (torch.randn(c, c, kt, 1, generator=g) * 0.02).to(d)
    b = torch.zeros(c, device=d)
    sc, sh = torch.ones(c, device=d), torch.zeros(c, device=d)
    bf = "split" if a.split else a.bf16
    wp = ops.tcn_pack(w, c, c, kt, s, 0, bf16=bf)
    wpd = ops.tcn_pack(w, c, c, kt, s, 2, bf16=bf)
    peak = 2500.0 if bf is True else 157.3
    calls = {
        "fwd": lambda: ops.tcn_conv(y, None if a.noact else sc, None if a.noact else sh, not a.noact, wp, b, None, None,
                                    None, None, None, False, n, c, c, tin, tout, kt, s, 0, bf16=bf),
        "dgrad": lambda: ops.tcn_conv(du, None, None, False, wpd, None, None, None, None, None, None,
                                      False, n, c, c, tin, tout, kt, s, 2, bf16=bf),
        "wgrad": lambda: ops.tcn_wgrad(du, y, sc, sh, True, n, c, c, tin, tout, kt, s, False, w.shape,
                                       bf16=bf),
    }
    fl = 2.0 * c * c * kt * tout * 16 * n
    for name, fn in calls.items():
        if a.which not in ("all", name):
            continue
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print("%-6s C=%d s=%d Tin=%d N=%d: %8.3f ms  %7.2f TFLOP/s  %.3f of %.1f" %
              (name, c, s, tin, n, ms, fl / ms / 1e9, fl / ms / 1e9 / peak, peak), flush=True)


if __name__ == "__main__":
    main()
