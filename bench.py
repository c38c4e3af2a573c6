#!/usr/bin/env python3
"""bench.py - ST-GCN fwd+bwd sequences/s on MI355X (BASELINE.json's metric).

A "step" is one training iteration of the hot path over one batch of synthetic skeletons that
is already resident in HBM: training-mode forward of the real_only Classifier (Kt=75, 3->32->64->64),
CrossEntropyLoss, backward through every HIP kernel, [one RCCL all-reduce of the flat gradient
bucket when world > 1], Adam.step().  Workload = BASELINE configs[1]'s tensor
(N=4096 x 3 x 240 x 16, fp32) per GPU; weak scaling (per-GPU batch fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu] [--no-kernels]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the temporal-conv MFMA
kernels), timed live with HIP events on the launch stream; `cpu_baseline` is the CPU oracle
(a from-scratch port of the reference's PyTorch path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (nominal clock)


def make_model(device, kt=75):
    from step_amd.net.classifier import Classifier
    torch.manual_seed(0)
    m = Classifier(3, 4, {"strategy": "spatial"}, temporal_kernel_size=kt)
    # weights_init of the reference harness (processor.py:117-129)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.normal_(0.0, 0.02)
                mod.bias.zero_()
            elif isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.weight.normal_(1.0, 0.02)
                mod.bias.zero_()
    return m.to(device)


def synth(n, t, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, t, 16, 1, generator=g) * 0.5
    y = torch.randint(0, 4, (n,), generator=g)
    return x.to(device), y.to(device)


def tcn_flops(n, c, t_out, kt=75):
    return 2.0 * c * c * kt * t_out * 16 * n


# north_star's wide layers (C = 128 / 256; the reference leaves these blocks commented out,
# classifier_stgcn_real_only/net/classifier.py:42-49): (channels, stride, T_in) of the original ST-GCN schedule
WIDE_BLOCKS = [(128, 2, 120), (128, 1, 60), (256, 2, 60), (256, 1, 30)]


def time_kernels(n, t, device, reps=3, bf16=False, blocks=None, tag="b"):
    """HIP-event timing of each temporal-conv kernel of the step (fwd, dgrad, wgrad x 3 blocks),
    launched alone on the current stream."""
    from step_amd import ops
    out = []
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    if blocks is None:
        blocks = [(32, 1, t), (64, 2, t), (64, 1, (t - 1) // 2 + 1)]
    g = torch.Generator().manual_seed(7)
    for bi, (c, s, tin) in enumerate(blocks):
        tout = (tin - 1) // s + 1
        y = torch.randn(n, c, tin, 16, generator=g).to(device)
        du = torch.randn(n, c, tout, 16, generator=g).to(device)
        w = (torch.randn(c, c, 75, 1, generator=g) * 0.02).to(device)
        b = torch.zeros(c, device=device)
        sc = torch.ones(c, device=device)
        sh = torch.zeros(c, device=device)
        wp = ops.tcn_pack(w, c, c, 75, s, 0, bf16=bf16)
        wpd = ops.tcn_pack(w, c, c, 75, s, 2, bf16=bf16)
        calls = {
            "tcn_fwd": lambda: ops.tcn_conv(y, sc, sh, True, wp, b, None, None, None, None, None, False,
                                            n, c, c, tin, tout, 75, s, 0, bf16=bf16),
            "tcn_dgrad": lambda: ops.tcn_conv(du, None, None, False, wpd, None, None, None, None, None,
                                              None, False, n, c, c, tin, tout, 75, s, 2, bf16=bf16),
            "tcn_wgrad": lambda: ops.tcn_wgrad(du, y, sc, sh, True, n, c, c, tin, tout, 75, s, False,
                                               w.shape, bf16=bf16),
        }
        for name, fn in calls.items():
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = tcn_flops(n, c, tout)
            out.append({"kernel": "%s_%s%d" % (name, tag, bi), "C": c, "stride": s, "T_in": tin, "ms": round(ms, 3),
                        "tflops": round(fl / ms / 1e9, 2),
                        "frac": round(fl / ms / 1e9 / peak, 4)})
        del y, du
    return out


def time_dft_kernels(n, t, device, reps=3):
    """HIP-event timing of every stage of the frequency-domain temporal conv (the default fp32 path; step_amd/dft.py,
    csrc/tcn_dft.hip) for the three blocks of the step, each with the FLOPs its GEMM executes:
    transform / inverse rows * frames * C * 16 per sequence (half-size forms: rows x frames / 2 multiply-adds per column),
    mixing and weight gradient 2 * 3 C^2 * 16 * NF (complex products as three real ones)."""
    from step_amd import dft
    out = []
    g = torch.Generator().manual_seed(11)
    for bi, (c, s, tin) in enumerate([(32, 1, t), (64, 2, t), (64, 1, (t - 1) // 2 + 1)]):
        tout = (tin - 1) // s + 1
        # stride 2 runs as two stride-1 convs on the even / odd frames (dft.phases): transforms of tin / 2 frames on a
        # grid of ~half the bins, 2 C -> C mixing per bin
        ph = dft.phases(c, tin, 75, s, False)
        tg, ktg = dft.phase_geometry(tin, 75) if ph == 2 else (tin, 75)
        rows, nf, L = dft.geometry(tg, ktg)
        y = torch.randn(n, c, tin, 16, generator=g).to(device)
        du = torch.randn(n, c, tout, 16, generator=g).to(device)
        w = (torch.randn(c, c, 75, 1, generator=g) * 0.02).to(device)
        b = torch.zeros(c, device=device)
        sc, sh = torch.ones(c, device=device), torch.zeros(c, device=device)
        wf, wb = dft.weight_spectrum(w, c, tin, 75, False, tout, s)
        sums = torch.zeros(2 * c + 1, dtype=torch.float64, device=device)
        if ph == 2:
            tr_a = lambda: dft.transform_phases(y, sc, sh, True, tin, 75)
            mix_f = lambda: dft.mix(sa, wf, 2 * c, c)
            inv_f = lambda: dft.inverse(sm, b, tout, 1, tg, ktg, sums)
            mix_b = lambda: dft.mix(sdu, wb, c, 2 * c)
            da = torch.empty_like(y)
            inv_b = lambda: dft._lib.call("step_dft_inv_ph", smb, dft.inv_matrix(tg, 1, tg, ktg, device), da, n, c, rows, tg, 2)
        else:
            tr_a = lambda: dft.transform(y, sc, sh, True, 1, tin, 75)
            mix_f = lambda: dft.mix(sa, wf, c)
            inv_f = lambda: dft.inverse(sm, b, tout, s, tin, 75, sums)
            mix_b = lambda: dft.mix(sdu, wb, c)
            inv_b = lambda: dft.inverse(smb, None, tin, 1, tin, 75)
        sa = tr_a()
        sm = mix_f()
        sdu = dft.transform_dout(du, s, tin, 75)
        smb = mix_b()
        fmix = 2.0 * 3 * c * c * ph * 16 * nf * n
        sa_bytes, sdu_bytes = 4.0 * sa.numel(), 4.0 * sdu.numel()      # (sa holds both phases: ph * C channels)
        stages = [
            ("dft_transform", tr_a, 1.0 * rows * tin * c * 16 * n, 4.0 * y.numel() + sa_bytes),
            ("dft_mix", mix_f, fmix, sa_bytes + sdu_bytes),
            ("dft_inverse", inv_f, 1.0 * rows * tout * c * 16 * n, sdu_bytes + 4.0 * du.numel()),
            ("dft_transform_dout", lambda: dft.transform_dout(du, s, tin, 75), 1.0 * rows * tout * c * 16 * n, 4.0 * du.numel() + sdu_bytes),
            ("dft_mix_bwd", mix_b, fmix, sdu_bytes + sa_bytes),
            ("dft_inverse_bwd", inv_b, 1.0 * rows * tin * c * 16 * n, sa_bytes + 4.0 * y.numel()),
            ("dft_wgrad", lambda: dft.conv_wgrad(sdu, sa, tin, 75, False, tuple(w.shape), tout, s), fmix, sa_bytes + sdu_bytes),
        ]
        for name, fn, fl, nbytes in stages:
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out.append({"kernel": "%s_b%d" % (name, bi), "C": c, "stride": s, "T_in": tin, "rows": rows, "phases": ph, "ms": round(ms, 3),
                        "tflops": round(fl / ms / 1e9, 2), "frac": round(fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "flops_per_launch": fl, "gbs": round(nbytes / ms / 1e6, 1),
                        "frac_hbm": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)})
        del y, du, sa, sm, sdu, smb
        torch.cuda.empty_cache()
    return out


def time_hbm_kernels(n, t, device, reps=3):
    """HIP-event timing of the HBM-bound stages of each block (graph conv fwd / bwd, BN statistics,
    BN+residual+ReLU apply, BN backward), with their algorithmic bytes (each tensor read or written
    once; 4 B per element) against the 8 TB/s HBM3E peak."""
    from step_amd import _lib, ops
    from step_amd.net.utils.graph import Graph
    out = []
    A = torch.tensor(Graph(strategy="spatial").A, dtype=torch.float32).to(device)
    nnz = ops.a_nnz_max(A)
    g = torch.Generator().manual_seed(3)
    for bi, (ci, co, tt, st, resconv) in enumerate([(3, 32, t, 1, True), (32, 64, t, 2, True),
                                                    (64, 64, (t - 1) // 2 + 1, 1, False)]):
        to = (tt - 1) // st + 1
        x = torch.randn(n, ci, tt, 16, generator=g).to(device)
        if bi > 0:
            x.relu_()        # blocks 1-2 read the previous block's output, a ReLU result (classifier.py:166,191): half zeros
        w = (torch.randn(3 * co, ci, generator=g) * 0.1).to(device)
        b = torch.zeros(3 * co, device=device)
        y = torch.empty(n, co, tt, 16, device=device)
        dy = torch.randn(n, co, tt, 16, generator=g).to(device)
        rw = (torch.randn(co, ci, generator=g) * 0.1).to(device) if resconv else None
        rb = torch.zeros(co, device=device) if resconv else None
        r = torch.empty(n, co, to, 16, device=device) if resconv else None
        dr = torch.randn(n, co, to, 16, generator=g).to(device) if resconv else None
        dx, dw, db, dA = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b), torch.empty_like(A)
        drw = torch.empty_like(rw) if resconv else None
        drb = torch.empty(co, device=device) if resconv else None
        nb = _lib.query("step_gcn_bwd_ws_bytes", n, ci, co, 3, tt)
        ws = torch.empty(nb, dtype=torch.uint8, device=device)
        bnw, bnb = torch.ones(co, device=device), torch.zeros(co, device=device)
        rm, rv = torch.zeros(co, device=device), torch.ones(co, device=device)
        mean, invstd, sc, sh = ops.bn_train_prepare(dy, 0, bnw, bnb, rm, rv)
        rn = r.numel() if resconv else 0
        ysums = torch.empty(2 * co + 1, dtype=torch.float64, device=device)
        rsums = torch.empty(2 * co + 1, dtype=torch.float64, device=device) if resconv else None
        snb = _lib.query("step_gcn_fwd_stats_ws_bytes", co)
        sws = torch.empty(snb, dtype=torch.uint8, device=device)
        calls = {
            # the train step's call: graph conv + residual conv WITH the BatchNorm statistics of both outputs in the
            # epilogue (step_gcn_fwd_stats); FLOPs / bytes counted are the convolution's alone
            "gcn_fwd": (lambda: _lib.call("step_gcn_fwd_stats", x, None, None, w, b, 0, A, nnz, y, rw, rb, 0, r,
                                          n, ci, co, 3, tt, st, ysums, rsums, sws, snb), x.numel() + y.numel() + rn),
            "gcn_bwd": (lambda: _lib.call("step_gcn_bwd", dy, dr, x, None, None, w, b, 0, A, nnz, rw, 0, dx, dw, db,
                                          dA, drw, drb, n, ci, co, 3, tt, st, ws, nb),
                        2 * (x.numel() + dy.numel() + rn) + dx.numel()),
            "bn_stats": (lambda: ops.bn_train_prepare(dy, 0, bnw, bnb, rm, rv), dy.numel()),
            "bn_res_relu": (lambda: ops.affine_act(dy, sc, sh, None, None, None, True, 0), 2 * dy.numel()),
            "bn_bwd": (lambda: ops.bn_backward(dy, y, None, sc, sh, 2, mean, invstd, bnw, True, 0), 5 * dy.numel()),
        }
        for name, (fn, nel) in calls.items():
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            gbs = 4.0 * nel / ms / 1e6
            row = {"kernel": "%s_b%d" % (name, bi), "bound": "hbm", "ms": round(ms, 3),
                   "gbs": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)}
            if name.startswith("gcn") and ci >= 32:
                # blocks 1-2: arithmetic intensity ~40 flop/B is above the fp32 ridge (~20), so the binding roofline
                # is the fp32 MFMA peak; algorithmic FLOPs = 1x1 conv to K*Co channels + residual conv (x2 backward)
                fl = (2.0 * 3 * ci * co * tt * 16 + (2.0 * ci * co * to * 16 if resconv else 0.0)) * n
                fl *= 2.0 if name == "gcn_bwd" else 1.0
                row.update({"bound": "mfma", "tflops": round(fl / ms / 1e9, 2),
                            "frac": round(fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4), "frac_hbm": row["frac"]})
            out.append(row)
        del x, y, dy, dx, r, dr, ws
    return out


def cpu_baseline(t, kt=75, n=192):
    """The oracle (port of the reference's PyTorch CPU path) on a bounded sample: fwd+bwd of n
    sequences at the bench's T, on the box's host cores; median of 3 passes."""
    from oracle import stgcn
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    m = make_model("cpu", kt)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    x, y = synth(n, t, 99, "cpu")
    stgcn.classifier_loss_and_grads(P, x[:2], y[:2], "real_only")          # warm-up
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        stgcn.classifier_loss_and_grads(P, x, y, "real_only")
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    # the reference's DEFAULT CPU path runs its convolutions through oneDNN; only the forward is usable that way on this
    # image (oracle/stgcn.py:31-37), so it is timed forward-only (eval), beside the fwd+bwd figure above
    ne = min(64, n)
    stgcn.set_onednn(True)
    try:
        with torch.no_grad():
            stgcn.classifier_forward(P, x[:2], "real_only")
            des = []
            for _ in range(3):
                t0 = time.perf_counter()
                stgcn.classifier_forward(P, x[:ne], "real_only")
                des.append(time.perf_counter() - t0)
    finally:
        stgcn.set_onednn(False)
    de = sorted(des)[1]
    onednn_eval = {"value": round(ne / de, 3), "unit": "sequences/s", "cores": threads, "kind": "port",
                   "sample": "oracle eval forward only, oneDNN ENABLED (the reference's default CPU convolution path), "
                             "N=%d x 3 x %d x 16 fp32, median of 3 passes (%.1f s each); compare with "
                             "inference_sequences_per_s_per_gpu, not with `value`" % (ne, t, de)}
    return {"value": round(n / dt, 3), "unit": "sequences/s", "cores": threads, "kind": "port",
            "onednn_eval_forward": onednn_eval,
            "sample": "oracle fwd+bwd (CrossEntropy, no optimizer), N=%d x 3 x %d x 16 fp32, median of 3 passes "
                      "(%.1f s each); torch CPU kernels with oneDNN DISABLED (its conv backward is unusable for "
                      "these (Kt,1) kernels on this image, oracle/stgcn.py:31-37) - the reference's default CPU "
                      "path would run its convolutions through oneDNN" % (n, t, dt)}


def model_flops_per_sequence_dft(t, kt=75, k=3):
    """forward FLOPs of one sequence as the DEFAULT fp32 path executes them: the temporal conv as transform + mixing +
    inverse transform (see time_dft_kernels), the rest as in model_flops_per_sequence; second value: fwd + bwd
    (forward + transform of dout + mixing + full-rate inverse + weight-gradient GEMM + 2x the graph conv)"""
    from step_amd import dft
    fwd, both, ci, tt = 0.0, 0.0, 3, t
    for co, s in ((32, 1), (64, 2), (64, 1)):
        to = (tt - 1) // s + 1
        ph = dft.phases(co, tt, kt, s, False)          # stride 2: two stride-1 convs on the even / odd frames
        rows, nf, _ = dft.geometry(*dft.phase_geometry(tt, kt)) if ph == 2 else dft.geometry(tt, kt)
        gcn = 2.0 * ci * k * co * tt * 16 + 2.0 * k * co * tt * 16 * 16 + (2.0 * ci * co * to * 16 if (ci != co or s != 1) else 0.0)
        mixf = 2.0 * 3 * co * co * ph * 16 * nf
        f = 1.0 * rows * tt * co * 16 + mixf + 1.0 * rows * to * co * 16
        b = 1.0 * rows * to * co * 16 + mixf + 1.0 * rows * tt * co * 16 + mixf
        fwd += gcn + f
        both += 3.0 * gcn + f + b
        ci, tt = co, to
    return fwd, both


def model_flops_per_sequence(t, kt=75, k=3):
    """algorithmic forward FLOPs of one sequence through the real_only stack (SURVEY 8d formulas):
    gcn 2*Ci*K*Co*T*V, dense aggregation 2*K*Co*T*V^2, tcn 2*Co^2*Kt*To*V, residual conv 2*Ci*Co*To*V"""
    fl, ci, tt = 0.0, 3, t
    for co, s in ((32, 1), (64, 2), (64, 1)):
        to = (tt - 1) // s + 1
        fl += 2.0 * ci * k * co * tt * 16 + 2.0 * k * co * tt * 16 * 16 + 2.0 * co * co * kt * to * 16
        if ci != co or s != 1:
            fl += 2.0 * ci * co * to * 16
        ci, tt = co, to
    return fl


def _sha256(paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(os.path.join(ROOT, p), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(kernel, batch, frames):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/gpu_round.sh
    (profiles/pmc_traffic.json).  Used only when that file was measured on THIS kernel source (sha256 of the
    .hip files recorded in it) at this shape; otherwise null - a stale number is worse than none."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        ent = pmc.get(kernel)
        if ent and pmc.get("batch") == batch and pmc.get("frames") == frames \
                and pmc.get("source_sha256") == _sha256(pmc.get("sources", [])):
            return ent["hbm_bytes"], {"file": "profiles/pmc_traffic.json", "git_sha": pmc.get("git_sha"),
                                      "source_sha256": pmc.get("source_sha256")[:16]}
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None, None


def self_launch(n):
    """`python bench.py --gpus N` outside torchrun: run `python -m torch.distributed.run --nproc-per-node N
    bench.py <same flags>` as a child process (one rank per GPU, RCCL) and return its exit code; rank 0's JSON
    line goes to our stdout.  On a box with fewer than N GPUs (rehearsal only) the ranks share GPU 0 and the
    collectives run over gloo through host memory - the line then says so in `rehearsal`."""
    import socket
    import subprocess
    env = dict(os.environ)
    have = torch.cuda.device_count()          # counts devices without initialising the GPU runtime
    if have < n:
        print("bench.py: %d GPUs asked, %d present: REHEARSAL, %d ranks share GPU 0 over gloo" % (n, have, n),
              file=sys.stderr)
        env.update(STEP_DIST_BACKEND="gloo", STEP_LOCAL_DEVICE="0", STEP_BENCH_REHEARSAL="1")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (sequences)")
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--mfma", choices=["f32", "f32_direct", "bf16", "f32_split", "f32_split_wgrad", "f32_split_gcn", "f32_dft"], default="f32",
                    help="MFMA operand type of the temporal convs for the headline number (f32 = BASELINE "
                         "configs[1]); with f32 a short bf16-operand run is reported beside it")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16-operand run")
    ap.add_argument("--sync-bn", action="store_true",
                    help="BatchNorm statistics over all ranks (the reference's single-device batch), "
                         "step_amd.parallel.sync_batchnorm; default is per-rank statistics (DDP semantics)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start the N ranks ourselves - a fresh child, before anything here touches the GPU
        sys.exit(self_launch(args.gpus))

    from step_amd import parallel
    import torch.distributed as dist
    if os.environ.get("STEP_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["STEP_BENCH_DEBUG"]), exit=True)
    rank, local, world = parallel.init_from_env()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    import step_amd
    model = step_amd.set_mfma_dtype(make_model(device).train(), args.mfma)
    parallel.broadcast_parameters(model)
    parallel.sync_batchnorm(args.sync_bn and world > 1)
    opt = torch.optim.Adam(model.parameters(), lr=0.1, weight_decay=1e-4)
    bucket = parallel.GradBucket(model.parameters())
    x, y = synth(args.batch, args.frames, 1234 + rank, device)

    def step():
        out, _ = model(x)
        loss = F.cross_entropy(out, y)
        bucket.zero_grad()                  # one memset of the flat buffer every p.grad is a view of
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(warmup, steps):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tdt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
            dt = float(tdt)
        return dt

    dt = timed(args.warmup, args.steps)
    seq_per_s = args.batch * world * args.steps / dt

    # the same step with bf16 MFMA operands in the temporal convs (BASELINE's bf16 training config)
    bf16_run = None
    if args.mfma == "f32" and not args.no_bf16:
        step_amd.set_mfma_dtype(model, "bf16")
        k = min(3, args.steps)
        dtb = timed(1, k)
        bf16_run = {"sequences_per_s": round(args.batch * world * k / dtb, 1),
                    "ms_per_step": round(dtb / k * 1e3, 2), "steps": k,
                    "note": "same model and step, temporal convs on v_mfma_f32_32x32x16_bf16 (fp32 tensors, "
                            "fp32 accumulate); logits within 1e-2 of fp32 (tests/test_gpu_bf16.py)"}
        step_amd.set_mfma_dtype(model, "f32")

    # the fp32 step with the temporal kernels' products taken on the bf16 matrix cores from fp32 operands split into
    # bf16 hi + lo (opt-in modes, held to the fp32 parity bars of the reference fixtures: tests/test_gpu_models.py,
    # tests/test_gpu_ops.py); reported beside the headline, NEVER as it
    split_run = None
    if args.mfma == "f32" and not args.no_bf16:
        split_run = {"note": "same model and step on fp32 tensors; products as bf16 MFMA terms on operands split into "
                             "bf16 hi + lo (16 mantissa bits: hi*hi + hi*lo + lo*hi, the dropped lo*lo is 2^-16 "
                             "relative), fp32 accumulate; loss, outputs and every gradient of the reference's "
                             "train-step fixtures hold at the fp32 bars"}
        for key, mode in (("wgrad_only", "f32_split_wgrad"), ("all_temporal_kernels", "f32_split"),
                          ("temporal_and_graph_conv", "f32_split_gcn")):
            step_amd.set_mfma_dtype(model, mode)
            k = min(3, args.steps)
            dts = timed(1, k)
            split_run[key] = {"mode": mode, "sequences_per_s": round(args.batch * world * k / dts, 1),
                              "ms_per_step": round(dts / k * 1e3, 2), "steps": k}
            if mode != "f32_split_wgrad" and not args.no_infer:
                model.eval()
                with torch.no_grad():
                    model(x)
                    torch.cuda.synchronize()
                    ti = time.perf_counter()
                    for _ in range(3):
                        model(x)
                    torch.cuda.synchronize()
                    split_run[key]["inference_sequences_per_s_per_gpu"] = round(args.batch * 3 / (time.perf_counter() - ti), 1)
                model.train()
        step_amd.set_mfma_dtype(model, "f32")

    # the same fp32 step on the DIRECT implicit-GEMM temporal kernels (what the headline ran until round 3)
    direct_run = None
    if args.mfma == "f32" and not args.no_bf16:
        step_amd.set_mfma_dtype(model, "f32_direct")
        k = min(3, args.steps)
        dtd = timed(1, k)
        direct_run = {"mode": "f32_direct", "sequences_per_s": round(args.batch * world * k / dtd, 1),
                      "ms_per_step": round(dtd / k * 1e3, 2), "steps": k,
                      "step_tflops_per_gpu": round(3.0 * model_flops_per_sequence(args.frames) * args.batch * k / dtd / 1e12, 2),
                      "note": "same model, tensors and arithmetic type; the temporal conv and its gradients as direct "
                              "implicit GEMMs (75 multiply-adds per output and channel pair) instead of in the frequency domain"}
        step_amd.set_mfma_dtype(model, "f32")

    # forward-only (eval, fused epilogues) rate of the same tensor, for reference
    infer = 0.0
    model.eval()
    with torch.no_grad():
      if not args.no_infer:
        model(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        infer = args.batch * 3 / (time.perf_counter() - t1)

    kernels, roof = None, None
    if rank == 0 and not args.no_kernels:
        del x
        torch.cuda.empty_cache()
        kernels = time_kernels(args.batch, args.frames, device, bf16=(args.mfma == "bf16"))
        dom_direct = max(kernels, key=lambda k: k["ms"])
        dom = dom_direct
        torch.cuda.empty_cache()
        if args.mfma in ("f32", "f32_dft"):
            # the default fp32 step runs the frequency-domain stages: its dominant kernel is one of those
            kd = time_dft_kernels(args.batch, args.frames, device)
            dom = max(kd, key=lambda k: k["ms"])
            kernels = kd + kernels
            torch.cuda.empty_cache()
        kernels = kernels + time_hbm_kernels(args.batch, args.frames, device)
        if args.mfma == "f32":          # MFMA utilisation of the wide layers (kernel-level bench points, N = batch / 2)
            torch.cuda.empty_cache()
            kernels = kernels + time_kernels(max(1, args.batch // 2), args.frames, device, blocks=WIDE_BLOCKS, tag="wide")
        if bf16_run is not None:
            # roofline of the bf16-operand step's dominant kernel, against the nominal dense bf16 MFMA peak and against
            # what register-only bf16 MFMAs sustain on this box at the clock they pull (tools/mfma_peak.hip)
            torch.cuda.empty_cache()
            kb = time_kernels(args.batch, args.frames, device, bf16=True)
            domb = max(kb, key=lambda k: k["ms"])
            bf16_run["kernels"] = kb
            bf16_run["roofline"] = {"bound": "mfma", "kernel": domb["kernel"], "achieved": domb["tflops"],
                                    "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": domb["frac"],
                                    "sustained_on_this_box": [1580.0, 1840.0],
                                    "frac_of_sustained": [round(domb["tflops"] / 1840.0, 4), round(domb["tflops"] / 1580.0, 4)],
                                    "traffic": None}
            torch.cuda.empty_cache()
        traffic, traffic_src = (pmc_traffic(dom["kernel"], args.batch, args.frames) if args.mfma == "f32"
                                else (None, None))
        direct_flops = tcn_flops(args.batch, dom_direct["C"], (args.frames - 1) // 2 + 1
                                 if dom_direct["kernel"].endswith(("b1", "b2")) else args.frames)
        roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                "peak": PEAK_BF16_MFMA_TFLOPS if args.mfma == "bf16" else PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_src,
                "flops_per_launch": dom.get("flops_per_launch", direct_flops)}
        if dom.get("frac_hbm", 0.0) > dom["frac"]:
            # the frequency-domain stages stream a spectrum per launch: where the bytes' fraction of the HBM peak exceeds
            # the FLOPs' fraction of the MFMA peak, HBM is the binding roof (the other fraction is kept beside it)
            roof.update({"bound": "hbm", "achieved": dom["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dom["frac_hbm"],
                         "bytes_per_launch": round(dom["gbs"] * dom["ms"] * 1e6), "mfma_tflops": dom["tflops"], "frac_mfma": dom["frac"]})
        elif "frac_hbm" in dom:
            roof.update({"hbm_gbs": dom["gbs"], "frac_hbm": dom["frac_hbm"]})
        if direct_run is not None:
            td, tsd = pmc_traffic(dom_direct["kernel"], args.batch, args.frames)
            direct_run["roofline"] = {"bound": "mfma", "kernel": dom_direct["kernel"], "achieved": dom_direct["tflops"],
                                      "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom_direct["frac"],
                                      "traffic": td, "traffic_source": tsd, "flops_per_launch": direct_flops}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:       # CPU baseline: rank 0 at N=1 only
        cpu = cpu_baseline(args.frames)

    if rank == 0:
        line = {
            "metric": "ST-GCN fwd+bwd sequences/sec", "value": round(seq_per_s, 1), "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32_dft": "f32", "f32_direct": "f32"}.get(args.mfma, args.mfma), "data": "synthetic",
            "config": {"workload": "configs[1] tensor: real_only ST-GCN classifier (Kt=75, 3->32->64->64), "
                                   "N=%d x 3 x %d x 16 per GPU, fp32 tensors%s; step = train-mode fwd + "
                                   "CrossEntropy + bwd + grad all-reduce + Adam"
                                   % (args.batch, args.frames,
                                      {"f32": ", fp32 arithmetic (temporal convs in the frequency domain: fp32 MFMA GEMMs)",
                                       "f32_dft": ", fp32 arithmetic (temporal convs in the frequency domain: fp32 MFMA GEMMs)",
                                       "f32_direct": ", fp32 arithmetic (direct implicit-GEMM temporal convs)",
                                       "bf16": ", bf16 MFMA operands in the temporal convs",
                                       "f32_split": ", fp32 operands as bf16 hi + lo on the bf16 matrix cores",
                                       "f32_split_gcn": ", fp32 operands as bf16 hi + lo on the bf16 matrix cores (graph-conv forward too)",
                                       "f32_split_wgrad": ", wgrad operands as bf16 hi + lo"}.get(args.mfma, "")),
                       "global_batch": args.batch * world, "frames": args.frames,
                       "parallelism": "dp%d%s" % (world, "+syncbn" if args.sync_bn and world > 1 else "")},
            "inference_sequences_per_s_per_gpu": round(infer, 1),
            # FLOPs the step EXECUTES (frequency-domain temporal convs in the default mode); the direct-conv count of the
            # same operators is kept beside it - it exceeds the MFMA peak by construction, 4-10x fewer multiply-adds run
            "step_tflops_per_gpu": round((model_flops_per_sequence_dft(args.frames)[1] if args.mfma in ("f32", "f32_dft")
                                          else 3.0 * model_flops_per_sequence(args.frames)) * args.batch * args.steps / dt / 1e12, 2),
            "direct_conv_equivalent_tflops_per_gpu": round(3.0 * model_flops_per_sequence(args.frames) * args.batch * args.steps / dt / 1e12, 2),
            "rehearsal": ("%d ranks share one GPU, collectives over gloo through host memory: not a scaling number"
                          % world) if os.environ.get("STEP_BENCH_REHEARSAL") else None,
            "f32_direct": direct_run, "bf16_mfma": bf16_run, "f32_split_operands": split_run, "roofline": roof, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                                     # rank 0's extra measurements are done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
