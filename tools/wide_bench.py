#!/usr/bin/env python3
"""MFMA utilisation of the temporal-conv kernels on the WIDE layers north_star names (C = 128 / 256: the blocks
the reference leaves commented out, classifier_stgcn_real_only/net/classifier.py:42-49).  Shapes follow the
original ST-GCN schedule at T_in = 240: 128 channels at T = 60 (stride-2 entry from T = 120), 256 at T = 30.
usage: python tools/wide_bench.py [--n 2048] [--reps 3]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402

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


if __name__ == "__main__":
    main()
