#!/usr/bin/env python3
"""Stage timings of the frequency-domain temporal conv (step_amd/dft.py) beside the direct fp32 kernels.
usage: python tools/dft_bench.py [--n 4096] [--c 64] [--stride 2] [--t 240] [--reps 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import dft, ops  # noqa: E402


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
    ap.add_argument("--c", type=int, default=64)
    ap.add_argument("--stride", type=int, default=2)
    ap.add_argument("--t", type=int, default=240)
    ap.add_argument("--kt", type=int, default=75)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-direct", action="store_true")
    ap.add_argument("--only", default="", help="run only the stages whose name contains this")
    a = ap.parse_args()
    d = "cuda:0"
    n, c, s, t, kt = a.n, a.c, a.stride, a.t, a.kt
    to = (t - 1) // s + 1
    g = torch.Generator().manual_seed(7)
    y = torch.randn(n, c, t, 16, generator=g).to(d)
    du = torch.randn(n, c, to, 16, generator=g).to(d)
    w = (torch.randn(c, c, kt, 1, generator=g) * 0.02).to(d)
    b = torch.zeros(c, device=d)
    sc, sh = torch.ones(c, device=d), torch.zeros(c, device=d)
    ph = dft.phases(c, t, kt, s, False)
    if ph == 2:
        tp, ktp = dft.phase_geometry(t, kt)
        rows, NF, L = dft.geometry(tp, ktp)
    else:
        tp, ktp = t, kt
        rows, NF, L = dft.geometry(t, kt)
    wf, wb = dft.weight_spectrum(w, c, t, kt, False, to, s)
    sums = torch.zeros(2 * c + 1, dtype=torch.float64, device=d)
    want = lambda *names: not a.only or any(a.only in x for x in names)      # set up only what the chosen stages read
    tr_a = (lambda: dft.transform_phases(y, sc, sh, True, t, kt)) if ph == 2 else (lambda: dft.transform(y, sc, sh, True, 1, t, kt))
    mix_f = (lambda: dft.mix(sa, wf, 2 * c, c)) if ph == 2 else (lambda: dft.mix(sa, wf, c))
    inv_f = (lambda: dft.inverse(sm, b, to, 1, tp, ktp, sums)) if ph == 2 else (lambda: dft.inverse(sm, b, to, s, t, kt, sums))
    mix_b = (lambda: dft.mix(sdu, wb, c, 2 * c)) if ph == 2 else (lambda: dft.mix(sdu, wb, c))
    sa = tr_a() if want("fwd: mix", "bwd: wgrad", "fwd: inverse(+stats)") or not a.only else None
    sm = mix_f() if want("fwd: inverse(+stats)") else None
    sdu = dft.transform_dout(du, s, t, kt) if want("bwd: mix", "bwd: inverse", "bwd: wgrad") else None
    smb = mix_b() if want("bwd: inverse") else None
    inv_b = (lambda: dft._lib.call("step_dft_inv_ph", smb, dft.inv_matrix(tp, 1, tp, ktp, d), torch.empty_like(y), n, c, rows, tp, 2)) \
        if ph == 2 else (lambda: dft.inverse(smb, None, t, 1, t, kt))
    fl = 2.0 * c * c * kt * to * 16 * n
    stages = [
        ("wspec", lambda: dft.weight_spectrum(w, c, t, kt, False, to, s), 0),
        ("fwd: transform(a)", tr_a, 1.0 * rows * t * c * 16 * n),
        ("fwd: mix", mix_f, 2.0 * 3 * c * c * ph * 16 * NF * n),
        ("fwd: inverse(+stats)", inv_f, 1.0 * rows * to * c * 16 * n),
        ("bwd: transform(du)", lambda: dft.transform_dout(du, s, t, kt), 1.0 * rows * to * c * 16 * n),
        ("bwd: mix", mix_b, 2.0 * 3 * c * c * ph * 16 * NF * n),
        ("bwd: inverse", inv_b, 1.0 * rows * t * c * 16 * n),
        ("bwd: wgrad", lambda: dft.conv_wgrad(sdu, sa, t, kt, False, tuple(w.shape), to, s), 2.0 * 3 * c * c * ph * 16 * NF * n),
    ]
    tot = {"fwd": 0.0, "bwd": 0.0}
    print("C=%d s=%d T=%d N=%d Kt=%d phases=%d: rows=%d L=%d, spectrum of the input %.2f GB" % (c, s, t, n, kt, ph, rows, L, n * rows * c * ph * 64 / 1e9))
    for name, fn, flops in stages:
        if a.only and a.only not in name:
            continue
        ms = timeit(fn, a.reps)
        if name[:3] in tot:
            tot[name[:3]] += ms
        print("  %-22s %8.3f ms  %7.1f TFLOP/s executed" % (name, ms, flops / ms / 1e9), flush=True)
    if not a.only:
        print("  forward %.2f ms, backward (dgrad + wgrad) %.2f ms; direct-conv FLOPs per pass %.2f T -> %.0f / %.0f TFLOP/s equivalent"
              % (tot["fwd"], tot["bwd"], fl / 1e12, fl / tot["fwd"] / 1e9, 2 * fl / tot["bwd"] / 1e9))
    if kt == 75 and not a.no_direct:
        wp = ops.tcn_pack(w, c, c, kt, s, 0)
        wpd = ops.tcn_pack(w, c, c, kt, s, 2)
        f = timeit(lambda: ops.tcn_conv(y, sc, sh, True, wp, b, None, None, None, None, None, False, n, c, c, t, to, kt, s, 0), a.reps)
        dg = timeit(lambda: ops.tcn_conv(du, None, None, False, wpd, None, None, None, None, None, None, False, n, c, c, t, to, kt, s, 2), a.reps)
        wg = timeit(lambda: ops.tcn_wgrad(du, y, sc, sh, True, n, c, c, t, to, kt, s, False, w.shape), a.reps)
        print("  direct kernels: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % (f, dg, wg))


if __name__ == "__main__":
    main()
