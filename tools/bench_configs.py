#!/usr/bin/env python3
"""Single-GPU throughput of every GPU configuration BASELINE.json names (configs[1..4]), beside the
headline of bench.py.  One JSON object per configuration, one per line; each states the algorithmic
FLOPs per sequence it was priced with and the fraction of the fp32 (or nominal bf16) MFMA peak.

    python tools/bench_configs.py [--reps 3] [--only 2,3,4,5]

configs[1]  real_only classifier inference, N=4096 x 3 x 240 x 16 fp32, eval mode
configs[2]  real_and_synth training step, bf16 MFMA operands, per-GPU batch 512, T=240, Adam
configs[3]  CVAE decoder bulk synthesis, micro-batch 2048, T=75 (12 500 gaits per GPU of the 100k job)
configs[4]  hybrid classifier eval, N=1024, T=1024, F=27 affective features
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_FP32 = 157.3e12
PEAK_BF16 = 2500.0e12
D = torch.device("cuda:0")
MODE = "f32"        # --mfma: step_amd.set_mfma_dtype mode applied to every model below (where it is not set explicitly)


def _mode(m):
    import step_amd
    return step_amd.set_mfma_dtype(m, MODE) if MODE != "f32" else m


def stack_flops(widths, strides, T, kt=75, res=None, K=3, V=16):
    """algorithmic forward FLOPs of one sequence (SURVEY 8d formulas)"""
    fl, t = 0.0, T
    for i, ((ci, co), s) in enumerate(zip(widths, strides)):
        to = (t - 1) // s + 1
        fl += 2.0 * ci * K * co * t * V            # 1x1 conv to K*Co
        fl += 2.0 * 46 / 16 * co * t * V           # aggregation over the 46 non-zeros of A (per w: 46/16)
        fl += 2.0 * co * co * kt * to * V          # temporal conv
        if res is None or res[i]:
            if ci != co or s != 1:
                fl += 2.0 * ci * co * to * V       # residual 1x1
        t = to
    return fl


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def init_like_reference(m):
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.normal_(0.0, 0.02)
                mod.bias.zero_()
            elif isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.weight.normal_(1.0, 0.02)
                mod.bias.zero_()
    return m


def config2(reps):
    from step_amd.net.classifier import Classifier
    torch.manual_seed(0)
    m = _mode(init_like_reference(Classifier(3, 4, {"strategy": "spatial"})).to(D).eval())
    n, t = 4096, 240
    x = (torch.randn(n, 3, t, 16, 1, generator=torch.Generator().manual_seed(1234)) * 0.5).to(D)
    with torch.no_grad():
        dt = timeit(lambda: m(x), reps)
    fl = stack_flops([(3, 32), (32, 64), (64, 64)], [1, 2, 1], t)
    return {"config": "configs[1] real_only inference N=4096 x 3 x 240 x 16 fp32 eval", "sequences_per_s": n / dt,
            "ms": dt * 1e3, "gflop_per_seq": fl / 1e9, "frac_fp32_mfma": n * fl / dt / PEAK_FP32}


def config3(reps):
    import step_amd
    from step_amd.net.classifier_real_and_synth import Classifier
    torch.manual_seed(0)
    n, t = 512, 240
    fl = 3 * stack_flops([(3, 32), (32, 64), (64, 64)], [1, 2, 2], t, res=[False, True, True])
    out = []
    for mode in ("bf16", "f32"):
        m = _mode(init_like_reference(Classifier(3, 4, {"strategy": "spatial"})).to(D).train())
        step_amd.set_mfma_dtype(m, mode)
        opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=1e-4)
        g = torch.Generator().manual_seed(1234)
        x = (torch.randn(n, 3, t, 16, 1, generator=g) * 0.5).to(D)
        y = torch.randint(0, 4, (n,), generator=g).to(D)

        def step():
            o, _ = m(x)
            loss = F.cross_entropy(o, y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        dt = timeit(step, reps)
        peak = PEAK_BF16 if mode == "bf16" else PEAK_FP32
        out.append({"config": "configs[2] real_and_synth train step (fwd+CE+bwd+Adam), per-GPU batch 512, T=240, "
                              "%s MFMA operands" % mode, "sequences_per_s": n / dt, "ms": dt * 1e3,
                    "gflop_per_seq_fwd_bwd": fl / 1e9, "frac_%s_mfma" % mode: n * fl / dt / peak})
    return out


def config4(reps):
    from step_amd.net.CVAE_stgcn import CVAE
    torch.manual_seed(0)
    m = _mode(CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"}).to(D).eval())
    n, t = 2048, 75
    z = (torch.randn(n, 32, generator=torch.Generator().manual_seed(5)) * 3 - 1.5).to(D)     # processor_stgcn.py:319
    l = F.one_hot(torch.arange(n) % 4, 4).float().to(D)
    with torch.no_grad():
        dt = timeit(lambda: m.decoder(z, l, t, 16), reps)
    fl = stack_flops([(32, 32), (32, 64), (64, 3)], [1, 1, 1], t)
    # decoder[0] sees z.repeat([1,1,T,V]): its graph conv runs on one frame and its temporal conv in closed form
    # (step_tcn_const_fwd), so only the other two blocks execute the reference's FLOP count
    fl_exec = stack_flops([(32, 64), (64, 3)], [1, 1], t) + 2.0 * 32 * 32 * t * 16
    return {"config": "configs[3] CVAE decoder synthesis, micro-batch 2048, T=75, eval", "gaits_per_s": n / dt,
            "ms": dt * 1e3, "seconds_for_12500_gaits": 12500 / (n / dt), "gflop_per_gait_reference": fl / 1e9,
            "gflop_per_gait_executed": fl_exec / 1e9, "frac_fp32_mfma": n * fl_exec / dt / PEAK_FP32,
            "note": "frac counts executed FLOPs; the reference's formulation of decoder[0] would add %.2f GFLOP/gait"
                    % ((fl - fl_exec) / 1e9)}


def cvae_train(reps):
    """generator_cvae training iteration (utils/processor_stgcn.py:140-152): encoder -> reparametrisation ->
    decoder -> between_frame_loss + KLD -> backward -> Adam; the reference logs ~16 s per iteration of batch 8
    for this (its loss is an O(B T V C) Python loop)."""
    from step_amd import losses
    from step_amd.net.CVAE_stgcn import CVAE
    torch.manual_seed(0)
    m = _mode(CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"}).to(D).train())
    opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=1e-4)
    n, t = 1024, 75
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(n, 3, t, 16, 1, generator=g) * 0.5).to(D)
    onehot = F.one_hot(torch.arange(n) % 4, 4).float().to(D)
    lenc = onehot[:, :, None, None, None].repeat(1, 1, t, 16, 1)

    def step():
        rec, mean, lsig, _ = m(x, lenc, onehot)
        loss = losses.vae_loss(x, rec, mean, lsig)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    dt = timeit(step, reps)
    fl = 3 * (stack_flops([(7, 64), (64, 32), (32, 32)], [1, 1, 1], t) + stack_flops([(32, 32), (32, 64), (64, 3)], [1, 1, 1], t))
    return {"config": "CVAE train step (encoder + decoder fwd/bwd, between_frame_loss + KLD, Adam), batch 1024, T=75",
            "gaits_per_s": n / dt, "ms": dt * 1e3, "gflop_per_gait_fwd_bwd": fl / 1e9, "frac_fp32_mfma": n * fl / dt / PEAK_FP32}


def config5(reps):
    from step_amd.net.classifier_hybrid import Classifier
    torch.manual_seed(0)
    m = _mode(init_like_reference(Classifier(3, 27, 4, {"strategy": "spatial"})).to(D).eval())
    n, t = 1024, 1024
    g = torch.Generator().manual_seed(9)
    xa = (torch.rand(n, 27, generator=g) * 2 - 1).to(D)
    xg = (torch.randn(n, 3, t, 16, 1, generator=g) * 0.5).to(D)
    with torch.no_grad():
        dt = timeit(lambda: m(xa, xg), reps)
    fl = stack_flops([(3, 32), (32, 64), (64, 64)], [1, 2, 2], t, res=[False, True, True])
    return {"config": "configs[4] hybrid eval N=1024, T=1024, F=27", "sequences_per_s": n / dt, "ms": dt * 1e3,
            "gflop_per_seq": fl / 1e9, "frac_fp32_mfma": n * fl / dt / PEAK_FP32}


def kt9(reps):
    """north_star's literal 9 x 1 temporal kernel (the reference uses 75; SURVEY 8d "extra bench points"):
    real_only architecture at Kt = 9, N = 4096 x 3 x 240 x 16 - the model drops below the fp32 ridge only in
    bf16, so the fp32 fractions are still against the MFMA peak; HBM bytes per sequence 7.22 MB forward."""
    import step_amd
    from step_amd.net.classifier import Classifier
    torch.manual_seed(0)
    n, t = 4096, 240
    g = torch.Generator().manual_seed(1234)
    x = (torch.randn(n, 3, t, 16, 1, generator=g) * 0.5).to(D)
    y = torch.randint(0, 4, (n,), generator=g).to(D)
    fl = stack_flops([(3, 32), (32, 64), (64, 64)], [1, 2, 1], t, kt=9)
    out = []
    m = _mode(init_like_reference(Classifier(3, 4, {"strategy": "spatial"}, temporal_kernel_size=9)).to(D).eval())
    with torch.no_grad():
        dt = timeit(lambda: m(x), reps)
    out.append({"config": "Kt=9 real_only inference N=4096 x 3 x 240 x 16 fp32 eval", "sequences_per_s": n / dt, "ms": dt * 1e3,
                "gflop_per_seq": fl / 1e9, "frac_fp32_mfma": n * fl / dt / PEAK_FP32,
                "hbm_GBps_algorithmic": n * 7.22e6 / dt / 1e9})
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=1e-4)

    def step():
        o, _ = m(x)
        loss = F.cross_entropy(o, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    dt = timeit(step, reps)
    out.append({"config": "Kt=9 real_only train step (fwd+CE+bwd+Adam) N=4096 x 3 x 240 x 16 fp32", "sequences_per_s": n / dt,
                "ms": dt * 1e3, "gflop_per_seq_fwd_bwd": 3 * fl / 1e9, "frac_fp32_mfma": 3 * n * fl / dt / PEAK_FP32,
                "hbm_GBps_algorithmic": 3 * n * 7.22e6 / dt / 1e9})
    return out


def aff(reps):
    """affective features + normalisation of N=4096 gaits x T=240 (HBM-bound: each coordinate once)"""
    from step_amd import affective
    n, t = 4096, 240
    out = []
    for dt_ in (torch.float32, torch.float64):
        g = torch.cumsum(0.3 * torch.randn(n, t, 48, generator=torch.Generator().manual_seed(2)), 1).to(dt_).to(D)
        for stride_f in (False, True):
            dt = timeit(lambda: affective.normalize_features(affective.compute_features(g, 1 / 30.0, stride_f, D)), reps)
            out.append({"config": "affective features + normalise, N=4096 x T=240 x 48 %s, stride features %s"
                                  % (str(dt_).replace("torch.", ""), stride_f),
                        "gaits_per_s": n / dt, "ms": dt * 1e3, "GBps_algorithmic": g.numel() * g.element_size() / dt / 1e9})
    return out


def batch1(reps):
    """The reference's prediction loop (processor.py:457-473: one batch-1 forward per gait, T = 75): latency per gait
    through step_amd.export.generate_predictions with batch=1, and the same 256 gaits in one batch"""
    import numpy as np
    from step_amd import export
    from step_amd.net.classifier import Classifier
    torch.manual_seed(0)
    m = _mode(init_like_reference(Classifier(3, 4, {"strategy": "spatial"})).to(D).eval())
    n, t = 256, 75
    data = (torch.randn(n, t, 48, generator=torch.Generator().manual_seed(5)) * 0.3).numpy()
    out = []
    for batch in (1, n):
        dt = timeit(lambda: export.generate_predictions(m, data, 4, 16, 3, batch=batch), reps)
        out.append({"config": "generate_predictions, %d gaits x T=75, batch %d (host numpy in, labels out)" % (n, batch),
                    "ms_per_gait": dt * 1e3 / n, "gaits_per_s": n / dt})
    x1 = torch.from_numpy(np.ascontiguousarray(np.moveaxis(data[:1].reshape(1, t, 16, 3, 1), [1, 2, 3], [2, 3, 1]))).float().to(D)
    with torch.no_grad():
        dt = timeit(lambda: m(x1), 200)
    out.append({"config": "one batch-1 eval forward, input resident (7 launches: 2 per block + head)", "ms_per_gait": dt * 1e3,
                "gaits_per_s": 1 / dt})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="2,3,4,5")
    ap.add_argument("--mfma", default="f32", help="f32 | f32_split | f32_split_wgrad for the configs that do not fix it")
    a = ap.parse_args()
    torch.cuda.set_device(D)
    global MODE
    MODE = a.mfma
    for k in a.only.split(","):
        r = {"2": config2, "3": config3, "4": config4, "5": config5, "aff": aff, "kt9": kt9, "cvae_train": cvae_train, "batch1": batch1}[k](a.reps)
        for line in (r if isinstance(r, list) else [r]):
            if MODE != "f32":
                line["mfma"] = MODE
            print(json.dumps({k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in line.items()}), flush=True)


if __name__ == "__main__":
    main()
