"""GPU: blocks and whole models of step_amd (HIP path through the C ABI) against the golden
vectors of the real reference and against the CPU oracle.  fp32, 1e-4 relative (north_star)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import stgcn
from tests.util import load_npz, sub, ckpt, assert_close, assert_grad_close, feeds_train_bn, grad_noise
from tests.test_oracle_golden import _block_cases, block_meta, trajectory_state_check

pytestmark = pytest.mark.gpu
D = "cuda:0"


def _load(module, P):
    missing, unexpected = module.load_state_dict({k: v for k, v in P.items()}, strict=False)
    assert not unexpected, unexpected
    return module.to(D)


@pytest.mark.parametrize("fname,tag", _block_cases())
@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("impl", ["f32", "f32_direct"])       # default (frequency domain where it pays) | direct kernels
def test_block_vs_reference_golden(fname, tag, mode, impl):
    from step_amd.net.blocks import st_gcn, st_gctn
    z = load_npz(fname)
    transposed, cin, cout, stride, residual = block_meta(tag)
    kt = int(tag.rsplit("k", 1)[1])
    P0 = sub(z, tag + ".P0.")
    d = sub(z, "%s.%s." % (tag, mode))
    blk = (st_gctn if transposed else st_gcn)(cin, cout, (kt, 3), stride, residual=residual)
    blk.load_state_dict(P0)
    blk = blk.to(D).train(mode == "train")
    blk.a_nnz_max = 4
    blk.mfma_dtype = impl
    x = d["x"].to(D).requires_grad_(True)
    A = d["A"].to(D).requires_grad_(True)
    y, _ = blk(x, A)
    assert_close(y, d["y"], 1e-4, "y")
    y.backward(d["gy"].to(D))
    assert_close(x.grad, d["gx"], 3e-4, "gx")
    supp = (d["A"] != 0).float()
    assert_close(A.grad.cpu() * supp, d["gA"] * supp, 3e-4, "gA")
    for name, p in blk.named_parameters():
        g = p.grad
        if "G." + name in d:
            noise = 1e-7 * float(d["gy"].abs().sum()) if (mode == "train" and feeds_train_bn(name)) else 1e-6
            assert_grad_close(g, d["G." + name], 3e-4, name, noise)
        else:
            assert_close(g[:2], d["Gsl." + name], 3e-4, "Gsl " + name)
            n = float(d["Gnorm." + name])
            assert abs(float(g.double().norm()) - n) <= 3e-4 * n, name
    sd = blk.state_dict()
    for k, v in d.items():
        if k.startswith("P1."):
            assert_grad_close(sd[k[3:]], v, 2e-5, k, 1e-7)
    # the no-grad eval path takes the fused MFMA epilogue: must agree with the autograd path
    if mode == "eval":
        with torch.no_grad():
            y2, _ = blk(d["x"].to(D), d["A"].to(D))
        assert_close(y2, d["y"], 1e-4, "fused eval y")


@pytest.mark.parametrize("mfma", ["f32", "f32_direct", "f32_split", "f32_split_gcn", "f32_dft"])
def test_classifier_real_only_ckpt395_eval(mfma):
    import step_amd
    from step_amd.net.classifier import Classifier
    m = _load(Classifier(3, 4, {"strategy": "spatial"}), ckpt("ckpt_classifier_real_only_epoch395")).eval()
    step_amd.set_mfma_dtype(m, mfma)        # (f32_split: split fp32 operands on the bf16 matrix cores, same bars)
    d = sub(load_npz("classifier_real_only_eval"), "")
    with torch.no_grad():
        probs, feat = m(d["h5_x"].to(D))
        assert_close(probs, d["h5_probs"], 1e-4, "h5 probs")
        assert_close(feat, d["h5_feat"], 1e-4, "h5 feat")
        # "classification accuracy identical": same arg-max on every shipped gait
        assert torch.equal(probs.argmax(1).cpu(), d["h5_probs"].argmax(1))
        probs, feat = m(d["rnd_x"].to(D))
        assert_close(probs, d["rnd_probs"], 1e-4, "rnd probs")
        assert_close(feat, d["rnd_feat"], 1e-4, "rnd feat")
        o, f = m.extract_feature(d["rnd_x"][:2].to(D))
        assert_close(o, d["ef_out"], 1e-4, "extract_feature out")
        assert_close(f, d["ef_feat"], 1e-4, "extract_feature feat")
        p1, f1 = m(d["rnd_x"][:1].to(D))                 # N == 1: feature loses its batch axis
        assert tuple(f1.shape) == (64,)


def test_real_and_synth_and_hybrid_eval_kt75():
    from step_amd.net import classifier_real_and_synth, classifier_hybrid
    base = ckpt("ckpt_classifier_real_only_epoch395")
    d = sub(load_npz("real_and_synth_eval_T240"), "")
    P = {k: v for k, v in base.items() if not k.startswith("st_gcn_networks.0.residual")}
    P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    m = classifier_real_and_synth.Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = m.to(D).eval()
    with torch.no_grad():
        o, f = m(d["x"].to(D))
    assert_close(o, d["out"], 1e-4, "logits")
    assert_close(f, d["feat"], 1e-4, "feat")

    d = sub(load_npz("hybrid_eval_T1024"), "")
    P = {k.replace("data_bn.", "data_bn1."): v for k, v in base.items()
         if not k.startswith(("st_gcn_networks.0.residual", "fcn."))}
    P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    m = classifier_hybrid.Classifier(3, 29, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = m.to(D).eval()
    with torch.no_grad():
        o = m(d["x_aff"].to(D), d["x_gait"].to(D))
    assert_close(o, d["out"], 1e-4, "hybrid logits")


def _variant_model(variant, kt=75):
    from step_amd.net import classifier, classifier_real_and_synth, classifier_hybrid
    g = {"strategy": "spatial"}
    if variant == "hybrid":
        return classifier_hybrid.Classifier(3, 29, 4, g, temporal_kernel_size=kt)
    mod = classifier if variant == "real_only" else classifier_real_and_synth
    return mod.Classifier(3, 4, g, temporal_kernel_size=kt)


@pytest.mark.parametrize("variant", ["real_only", "real_and_synth", "hybrid"])
def test_train_step_kt9_vs_reference(variant):
    """processor.py:316-328: forward, CrossEntropyLoss, backward, Adam(lr=0.1, wd=1e-4).step()"""
    z = load_npz("train_steps")
    d = sub(z, variant + "_k9.")
    P0 = {k[3:]: v for k, v in d.items() if k.startswith("P0.")}
    m = _variant_model(variant, 9)
    m.load_state_dict(P0)
    m = m.to(D).train()
    opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=1e-4)
    args = (d["x_aff"].to(D), d["x"].to(D)) if variant == "hybrid" else (d["x"].to(D),)
    out = m(*args)
    out = out if variant == "hybrid" else out[0]
    loss = torch.nn.CrossEntropyLoss()(out, d["labels"].to(D))
    opt.zero_grad()
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 2e-5
    assert_close(out, d["out"], 1e-4, "out")
    # 5e-3, not 1e-3: with 8 samples one ReLU whose pre-activation lies within an ulp of zero flips under any
    # change of summation order and moves whole gradient tensors by ~4e-3 of their max (measured: two fp32-exact
    # graph-conv kernels, outputs equal to 3e-7, every backward kernel's inputs equal to 3e-7 up to the flipped
    # element of block 1's output ReLU; tools/gcn_ab.py).  Block-level gradients and the Kt=75 step keep 1e-3.
    for k, p in m.named_parameters():
        assert_grad_close(p.grad, d["G." + k], 5e-3, k, grad_noise(k))
    opt.step()
    sd = m.state_dict()
    for k, v in sd.items():
        ref1 = d["P1." + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref1)
        elif "G." + k in d:
            mask = d["G." + k].abs() > 1e-5        # Adam's first step is lr*sign(g)
            assert float(((v.cpu() - ref1).abs() * mask).max()) < 2e-3, k
        else:
            assert_grad_close(v, ref1, 1e-4, k, 1e-7)


@pytest.mark.parametrize("mfma", ["f32", "f32_direct", "f32_split_wgrad", "f32_split", "f32_split_gcn", "f32_dft"])
def test_train_step_kt75_ckpt395_vs_reference(mfma):
    """(f32_split_wgrad: the weight gradient's products as three bf16 MFMAs on split fp32 operands - held to the
    same bars against the reference's own gradients)"""
    import step_amd
    z = load_npz("train_steps")
    d = sub(z, "real_only_k75_ckpt395.")
    m = _variant_model("real_only")
    m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
    m = step_amd.set_mfma_dtype(m.to(D).train(), mfma)
    out, _ = m(d["x"].to(D))
    loss = F.cross_entropy(out, d["labels"].to(D))
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 2e-5
    assert_close(out, d["out"], 1e-4, "out")
    for k, p in m.named_parameters():
        g = p.grad
        if "G." + k in d:
            assert_grad_close(g, d["G." + k], 1e-3, k, 2e-6 if feeds_train_bn(k) else 2e-7)
        else:
            assert_close(g[:2], d["Gsl." + k], 1e-3, k)
            n = float(d["Gnorm." + k])
            assert abs(float(g.double().norm()) - n) <= 1e-3 * n, k
    sd = m.state_dict()
    for k, v in d.items():
        if k.startswith("P1.") and "running" in k:
            assert_grad_close(sd[k[3:]], v, 1e-4, k, 1e-7)


@pytest.mark.parametrize("mfma", ["f32", "f32_direct", "f32_split", "f32_split_gcn", "f32_dft"])
def test_cvae_ckpt105_eval_and_train(mfma):
    import step_amd
    from step_amd.net.CVAE_stgcn import CVAE
    m = CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    m.load_state_dict(ckpt("ckpt_cvae_epoch105"))
    m = step_amd.set_mfma_dtype(m.to(D).eval(), mfma)      # (f32_split: the st_gctn ConvTranspose kernels on split operands)
    d = sub(load_npz("cvae_eval"), "", device=D)
    with torch.no_grad():
        mean, lsig = m.encoder(d["x"], d["lenc"])
        assert_close(mean, d["mean"], 1e-4, "mean")
        assert_close(lsig, d["lsig"], 1e-4, "lsig")
        assert_close(m.decoder(d["z"], d["onehot"], 75, 16), d["dec75"], 1e-4, "dec75")
        assert_close(m.decoder(d["z"][:2], d["onehot"][:2], 240, 16), d["dec240"], 1e-4, "dec240")
        # those two ran the synthesis path that never builds z.repeat([1,1,T,V]); the general path (taken whenever
        # autograd records) builds it and must give the same frames
        fast = m.decoder(d["z"], d["onehot"], 75, 16)
    slow = m.decoder(d["z"], d["onehot"], 75, 16)
    assert slow.requires_grad and not fast.requires_grad
    assert_close(fast, slow.detach(), 1e-5, "decoder: repeated-input path vs materialised repeat")
    # training-mode forward/backward (encoder -> reparam with the fixture's eps -> decoder)
    m.train()
    d = sub(load_npz("cvae_train"), "", device=D)
    T = d["x"].shape[2]
    mean, lsig = m.encoder(d["x"], d["lenc"])
    zz = d["eps"] * torch.exp(0.5 * lsig) + mean
    rec = m.decoder(zz, d["onehot"], T, 16)
    assert_close(mean, d["mean"], 1e-4, "mean")
    assert_close(rec, d["rec"], 1e-4, "rec")
    loss = ((rec - d["x"]) ** 2).mean() - 0.5 * torch.mean(1 + lsig - mean.pow(2) - lsig.exp())
    assert abs(float(loss) - float(d["loss"])) < 2e-5
    loss.backward()
    for k, p in m.named_parameters():
        if "G." + k in d:
            assert_grad_close(p.grad, d["G." + k], 1e-3, k, grad_noise(k))
        elif "Gsl." + k in d:
            assert_close(p.grad[:2], d["Gsl." + k], 1e-3, k)


def test_full_size_properties():
    """BASELINE config 2 size (N=4096, T=240 is too slow for the oracle): size-independent
    properties instead - batch-split invariance in eval mode (sequences are independent), and
    agreement of a strided subsample with the oracle."""
    from step_amd.net.classifier import Classifier
    P = ckpt("ckpt_classifier_real_only_epoch395")
    m = _load(Classifier(3, 4, {"strategy": "spatial"}), P).eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(512, 3, 240, 16, 1, generator=g) * 0.5
    with torch.no_grad():
        full, feat = m(x.to(D))
        a, _ = m(x[:200].to(D))
        b, _ = m(x[200:].to(D))
        assert torch.equal(full, torch.cat((a, b)))
        idx = torch.arange(0, 512, 64)
        ref, rfeat = stgcn.classifier_forward(P, x[idx], "real_only")
    assert_close(full[idx], ref, 1e-4, "probs")
    assert_close(feat[idx], rfeat, 1e-4, "feat")
    assert torch.allclose(full.sum(1).cpu(), torch.ones(512), atol=1e-5)


@pytest.mark.parametrize("variant,N,mfma", [("real_only", 4096, "f32"), ("real_and_synth", 512, "f32"),
                                            ("real_only", 4096, "f32_split_gcn"), ("real_only", 4096, "f32_direct"),
                                            ("real_and_synth", 512, "f32_direct")])
def test_full_size_train_step_replicated_batch(variant, N, mfma):
    """The bench workloads' own shapes in train mode (configs[1]: real_only, N=4096 x 3 x 240 x 16; configs[2]:
    real_and_synth, per-GPU batch 512): a batch made of N/8 copies of 8 gaits has the same batch statistics
    (mean, biased variance), the same mean CrossEntropy and therefore the same parameter gradients as the 8
    gaits alone, which the oracle does in seconds.  Exercises every kernel of the step with the full-size grids
    against the oracle; also eval-mode batch-split invariance at that size."""
    P = ckpt("ckpt_classifier_real_only_epoch395")
    if variant == "real_and_synth":
        d = sub(load_npz("real_and_synth_eval_T240"), "")
        P = {k: v for k, v in P.items() if not k.startswith("st_gcn_networks.0.residual")}
        P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    g = torch.Generator().manual_seed(N)
    x8 = torch.randn(8, 3, 240, 16, 1, generator=g) * 0.5
    l8 = torch.arange(8) % 4
    loss_ref, out_ref, grads = stgcn.classifier_loss_and_grads({k: v.clone() for k, v in P.items()}, x8, l8, variant)
    import step_amd
    m = step_amd.set_mfma_dtype(_load(_variant_model(variant), P).train(), mfma)   # (split operands: same bars)
    x = x8.repeat(N // 8, 1, 1, 1, 1).to(D)
    out, _ = m(x)
    loss = F.cross_entropy(out, l8.repeat(N // 8).to(D))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) < 2e-5
    assert_close(out[:8], out_ref, 1e-4, "train-mode output")
    assert torch.equal(out[:8], out[N - 8:])                      # copies are treated alike
    for k, p in m.named_parameters():
        assert_grad_close(p.grad, grads[k], 1e-3, k, 1e-5 if feeds_train_bn(k) else 1e-6)
    m.eval()
    with torch.no_grad():
        xe = torch.randn(N, 3, 240, 16, 1, generator=g).to(D) * 0.5
        full, _ = m(xe)
        parts = torch.cat([m(xe[i:i + N // 4 - 24])[0] for i in range(0, N, N // 4 - 24)])
    assert torch.equal(full, parts)


@pytest.mark.parametrize("N,T,M", [(1, 37, 1), (3, 9, 1), (2, 101, 1), (2, 40, 2), (5, 75, 1)])
def test_ragged_shapes_train_and_eval_vs_oracle(N, T, M):
    """Edge shapes: N == 1, T not a multiple of the 8/40-frame tiles, T odd under stride 2,
    T < Kt (kernel longer than the sequence), and M > 1 (persons folded into the batch)."""
    from step_amd.net.classifier import Classifier
    P = ckpt("ckpt_classifier_real_only_epoch395")
    g = torch.Generator().manual_seed(N * 1000 + T)
    x = torch.randn(N, 3, T, 16, M, generator=g) * 0.5
    m = _load(Classifier(3, 4, {"strategy": "spatial"}), P).eval()
    xo = x.permute(0, 4, 1, 2, 3).reshape(N * M, 3, T, 16, 1)       # oracle folds M itself
    with torch.no_grad():
        out, feat = m(x.to(D))
        ref, rfeat = stgcn.classifier_forward({k: v.clone() for k, v in P.items()}, xo, "real_only")
    if M == 1:
        assert_close(out, ref, 1e-4, "eval probs")
    else:
        # the reference averages pooled features over M before fcn (classifier.py:82)
        w = P["fcn.weight"].view(4, 64)
        pooled = rfeat.view(N, M, 64).mean(1)
        assert_close(out, torch.softmax(pooled @ w.t() + P["fcn.bias"], 1), 1e-4, "eval probs M>1")
        return
    if N == 1:
        return                                   # batch-of-one BN statistics: eval only
    labels = torch.arange(N) % 4
    Pt = {k: v.clone() for k, v in P.items()}
    loss_ref, out_ref, grads = stgcn.classifier_loss_and_grads(Pt, x, labels, "real_only")
    m.train()
    out, _ = m(x.to(D))
    loss = F.cross_entropy(out, labels.to(D))
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 2e-5
    for k, p in m.named_parameters():
        assert_grad_close(p.grad, grads[k], 1e-3, k, 2e-6 if feeds_train_bn(k) else 2e-7)


def test_graph_conv_modules_standalone():
    """ConvTemporalGraphical / ConvTransposeTemporalGraphical keep the reference call surface:
    forward(x, A) -> (y, A), with autograd."""
    from step_amd.net.utils.tgcn import ConvTemporalGraphical, ConvTransposeTemporalGraphical
    from oracle import graph_adjacency
    A = torch.tensor(graph_adjacency("spatial"), dtype=torch.float32)
    for cls, tr in ((ConvTemporalGraphical, False), (ConvTransposeTemporalGraphical, True)):
        torch.manual_seed(3)
        mod = cls(5, 7, 3)
        x = torch.randn(2, 5, 11, 16)
        ref = stgcn.conv_temporal_graphical(x, A, mod.conv.weight.detach(), mod.conv.bias.detach(), tr)
        mod = mod.to(D)
        xg = x.to(D).requires_grad_(True)
        y, A_out = mod(xg, A.to(D))
        assert A_out.shape == A.shape
        assert_close(y, ref, 1e-4, "y")
        y.sum().backward()
        assert xg.grad is not None and mod.conv.weight.grad is not None


def test_batched_prediction_and_feature_export():
    """processor.py:457-473 / :489-515 in large batches == the reference's batch-1 loop
    (its outputs on the shipped 60-gait fixture are in classifier_real_only_eval.npz)."""
    import numpy as np
    from step_amd import export
    from step_amd.net.classifier import Classifier
    m = _load(Classifier(3, 4, {"strategy": "spatial"}), ckpt("ckpt_classifier_real_only_epoch395")).train()
    d = sub(load_npz("classifier_real_only_eval"), "")
    x = d["h5_x"]                                                   # (N, 3, T, 16, 1)
    N, _, T, V, _ = x.shape
    raw = x[..., 0].permute(0, 2, 3, 1).reshape(N, T, V * 3).numpy()    # the loader's (N, T, 48)
    labels, out = export.generate_predictions(m, raw, 4, 16, 3, batch=25)
    assert m.training                                               # mode restored
    assert_close(torch.tensor(out), d["h5_probs"], 1e-4, "batched probs")
    assert np.array_equal(labels, d["h5_probs"].argmax(1).numpy().astype(np.float64))
    feats, by_key = export.extract_features(m, raw, 16, 3, keys=["gait%03d" % i for i in range(N)], batch=32)
    assert_close(torch.tensor(feats), d["h5_feat"], 1e-4, "batched features")
    assert feats.shape == (N, 64) and len(by_key) == N and by_key["gait007"].shape == (64,)
    one, _ = export.generate_predictions(m, raw[:3], 4, 16, 3, batch=1)    # the reference's own batching
    assert np.array_equal(one, labels[:3])


@pytest.mark.parametrize("mfma", ["f32", "f32_direct", "f32_split_wgrad", "f32_split", "f32_split_gcn", "f32_dft"])
def test_five_step_trajectory_vs_reference(mfma):
    """processor.py:316-335 over several iterations: 5 x (forward, CrossEntropyLoss, backward, Adam.step) on the
    real_only Classifier (Kt = 75, N = 8, T = 75), lr 0.1 dropped to 0.01 before the 4th step (adjust_lr,
    processor.py:260-268).  Weights are re-packed from the updated parameters every step; loss and train-mode
    output of EVERY step, the final state_dict, the BN running statistics and num_batches_tracked are held
    against the reference's run."""
    d = sub(load_npz("trajectory"), "")
    m = _variant_model("real_only")
    m.load_state_dict({k[3:]: v for k, v in d.items() if k.startswith("P0.")})
    import step_amd
    m = step_amd.set_mfma_dtype(m.to(D).train(), mfma)
    opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=1e-4)
    lossf = torch.nn.CrossEntropyLoss()
    for it in range(5):
        if it == 3:
            for g in opt.param_groups:
                g["lr"] = 0.01
        out, _ = m(d["x"][it].to(D))
        loss = lossf(out, d["labels"][it].to(D))
        opt.zero_grad()
        loss.backward()
        opt.step()
        # f32_split (all three temporal kernels on split operands, ~5e-6 per kernel instead of ~1e-6): the lr = 0.1
        # Adam steps amplify that - every step stays inside north_star's 1e-4 on the loss and 1e-3 on the outputs,
        # not inside the fp32 kernels' own 1e-5 / 1e-4, and the final-state comparison is left to the fp32 modes
        # (f32_split_gcn adds the graph-conv forward, whose aggregated z is itself rounded to 16 mantissa bits: its
        # step-1 output sits at 1.3e-3, measured, so its bars are 2e-4 / 2e-3)
        k = {"f32_split": 10.0, "f32_split_gcn": 20.0}.get(mfma, 1.0)
        assert abs(float(loss.detach()) - float(d["losses"][it])) <= k * 1e-5 * max(1.0, abs(float(d["losses"][it]))), it
        assert_close(out, d["outs"][it], k * 1e-4, "train-mode output of step %d" % it)
    if mfma not in ("f32_split", "f32_split_gcn"):
        assert trajectory_state_check(m.state_dict(), d) >= 55


def test_dense_A_loaded_into_a_sparse_graph_model():
    """The sparse (EMAX = 4) graph-conv kernels are selected from the A buffer's non-zero count; a checkpoint may
    overwrite A (it is a persistent buffer).  A dense A loaded into a model built with the sparse skeleton must
    give the oracle's numbers in every block, forward and backward."""
    P = ckpt("ckpt_classifier_real_only_epoch395")
    g = torch.Generator().manual_seed(17)
    P = dict(P, A=torch.rand(3, 16, 16, generator=g) * 0.2)
    x = torch.randn(4, 3, 40, 16, 1, generator=g) * 0.5
    labels = torch.arange(4) % 4
    loss_ref, out_ref, grads = stgcn.classifier_loss_and_grads({k: v.clone() for k, v in P.items()}, x, labels, "real_only")
    m = _variant_model("real_only")
    assert m.st_gcn_networks[0].a_nnz_max < 16
    m.load_state_dict(P)
    m = m.to(D).train()
    out, _ = m(x.to(D))
    assert m.st_gcn_networks[0].a_nnz_max == 16 and m.st_gcn_networks[2].gcn.a_nnz_max == 16
    F.cross_entropy(out, labels.to(D)).backward()
    assert_close(out, out_ref, 1e-4, "output with a dense A")
    for k, p in m.named_parameters():
        assert_grad_close(p.grad, grads[k], 1e-3, k, 2e-6 if feeds_train_bn(k) else 2e-7)


def test_first_block_identity_residual_behind_data_bn_is_refused():
    """ops.StGcnBlockFn: with data_bn folded into the graph-conv load, an identity residual would add the raw x
    where the reference adds the normalised x (classifier.py:68-78,170-171) - refused, not silently wrong"""
    from step_amd._lib import StepHipError
    from step_amd.net.classifier import Classifier
    m = Classifier(32, 4, {"strategy": "spatial"}).to(D)         # 32 -> 32, stride 1: identity residual in block 0
    with pytest.raises(StepHipError, match="identity residual"):
        m(torch.zeros(2, 32, 20, 16, 1, device=D))


def test_bn_rows_near_constant_column():
    """hybrid head's data_bn2 (classifier_hybrid/net/classifier.py:46,83): a column such as 100 + 1e-2 * randn must
    keep its variance (two-pass fp64 statistics), checked against torch's BatchNorm1d in float64"""
    from step_amd.ops import BnRowsFn
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, 7, generator=g)
    x[:, 2] = 100.0 + 1e-2 * torch.randn(256, generator=g)
    x[:, 5] = -3000.0 + 1e-1 * torch.randn(256, generator=g)
    bn = torch.nn.BatchNorm1d(7).to(D).train()
    ref_bn = torch.nn.BatchNorm1d(7).double().train()
    y = BnRowsFn.apply(bn, True, x.to(D), bn.weight, bn.bias)
    ref = ref_bn(x.double())
    assert float((y.cpu().double() - ref).abs().max()) < 2e-3      # fp32 inputs near 100 carry ~1e-5/1e-2 relative
    assert_close(bn.running_var, ref_bn.running_var, 1e-5, "running_var")
    assert_close(bn.running_mean, ref_bn.running_mean, 1e-6, "running_mean")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_second_gpu_without_set_device():
    """_lib.call launches on the tensors' device and stream, not on torch's current device"""
    from step_amd._lib import StepHipError
    P = ckpt("ckpt_classifier_real_only_epoch395")
    m = _load(_variant_model("real_only"), P).eval()
    x = torch.randn(3, 3, 30, 16, 1) * 0.5
    with torch.no_grad():
        a, _ = m(x.to(D))
        m1 = m.to("cuda:1")
        assert torch.cuda.current_device() == 0
        b, _ = m1(x.to("cuda:1"))
        assert torch.equal(a.cpu(), b.cpu())
        with pytest.raises(StepHipError, match="different devices"):
            m1(x.to(D))


@pytest.mark.parametrize("cin,cout,stride,T,N", [(64, 128, 2, 21, 3), (128, 128, 1, 13, 2), (128, 256, 2, 16, 2),
                                                 (256, 256, 1, 9, 2)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_wide_st_gcn_block_vs_oracle(cin, cout, stride, T, N, mode):
    """The 128 / 256-channel st_gcn blocks of the original ST-GCN schedule that the reference leaves commented out
    (classifier.py:42-49; north_star's MFMA bench layers): a whole block - wide graph conv + residual conv, BN,
    wide temporal conv, BN, add, ReLU - forward and backward against the oracle's st_gcn_block (no reference
    fixture can exist for them: the reference never builds these widths)."""
    from step_amd.net.blocks import st_gcn
    from step_amd.net.utils.graph import Graph
    torch.manual_seed(cin + cout + stride)
    blk = st_gcn(cin, cout, (75, 3), stride)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0.0, 0.05)
                m.bias.normal_(0.0, 0.1)
            elif isinstance(m, torch.nn.BatchNorm2d):
                m.weight.normal_(1.0, 0.1)
                m.bias.normal_(0.0, 0.1)
                m.running_mean.normal_(0.0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    P = {"b." + k: v.detach().clone() for k, v in blk.state_dict().items()}
    A = torch.tensor(Graph(strategy="spatial").A, dtype=torch.float32) * (torch.rand(3, 16, 16) + 0.5)
    x = torch.randn(N, cin, T, 16)
    gy_shape = (N, cout, (T - 1) // stride + 1, 16)
    gy = torch.randn(gy_shape)
    # oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if v.is_floating_point() and "running" not in k}
    Q = dict(P)
    Q.update(leaves)
    xo, Ao = x.clone().requires_grad_(True), A.clone().requires_grad_(True)
    yo = stgcn.st_gcn_block(xo, Ao, Q, "b", stride, True, mode == "train")
    yo.backward(gy)
    # product
    blk = blk.to(D).train(mode == "train")
    blk.a_nnz_max = 4
    xg, Ag = x.to(D).requires_grad_(True), A.to(D).requires_grad_(True)
    yg, _ = blk(xg, Ag)
    assert_close(yg, yo, 1e-4, "y")
    yg.backward(gy.to(D))
    assert_close(xg.grad, xo.grad, 3e-4, "gx")
    supp = (A != 0).float()
    assert_close(Ag.grad.cpu() * supp, Ao.grad * supp, 3e-4, "gA")
    for name, prm in blk.named_parameters():
        noise = 1e-7 * float(gy.abs().sum()) if (mode == "train" and feeds_train_bn(name)) else 1e-6
        assert_grad_close(prm.grad, leaves["b." + name].grad, 3e-4, name, noise)
    if mode == "train":
        sd = blk.state_dict()
        for k in ("tcn.0.running_mean", "tcn.0.running_var", "tcn.3.running_mean", "tcn.3.running_var"):
            assert_grad_close(sd[k], Q["b." + k], 2e-5, k, 1e-7)
    else:
        with torch.no_grad():
            y2, _ = blk(x.to(D), A.to(D))
        assert_close(y2, yo, 1e-4, "fused eval y")


@pytest.mark.parametrize("mfma", ["f32", "f32_direct"])
def test_eval_under_no_grad_takes_the_fused_epilogue(monkeypatch, mfma):
    """Eval-mode blocks under torch.no_grad() finish inside the temporal conv's epilogue (BN2 + residual + ReLU,
    classifier.py:189-191; the direct kernel's, or the inverse transform's in the frequency-domain default) although
    the parameters require grad; with autograd recording they keep the separate pass.  (ctx.needs_input_grad mirrors
    requires_grad whatever the grad mode is - this path once never ran.)"""
    import step_amd
    from step_amd import ops
    from step_amd.net.classifier import Classifier
    m = Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
    m = step_amd.set_mfma_dtype(m.to(D).eval(), mfma)
    x = torch.randn(3, 3, 75, 16, 1, generator=torch.Generator().manual_seed(3)).to(D)
    calls = []
    real = ops.affine_act
    monkeypatch.setattr(ops, "affine_act", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        fused, _ = m(x)
    assert not calls, "separate BN2 + residual + ReLU pass under no_grad"
    unfused, _ = m(x)
    assert len(calls) == 3 and unfused.requires_grad
    assert_close(fused, unfused.detach(), 1e-5, "fused vs separate epilogue")


def test_full_size_decoder_synthesis_replicated_batch():
    """BASELINE configs[3] at its stated size (Decoder.forward CVAE_stgcn.py:198-232, micro-batch 2048, T = 75, eval,
    z = randn * 3 - 1.5 and one-hot classes cycling 0-3 as processor_stgcn.py:311-333): 8 distinct (z, class) rows
    x 256.  Eval is per-sample, so copies must agree BIT FOR BIT wherever they sit in the batch, rows 0-7 must be the
    oracle's, and the reference's own decoder fixture rows must come out the same inside the big batch."""
    from step_amd.net.CVAE_stgcn import CVAE
    P = ckpt("ckpt_cvae_epoch105")
    m = CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = m.to(D).eval()
    g = torch.Generator().manual_seed(2048)
    z8 = torch.randn(8, 32, generator=g) * 3 - 1.5
    l8 = F.one_hot(torch.arange(8) % 4, 4).float()
    N = 2048
    with torch.no_grad():
        out = m.decoder(z8.repeat(N // 8, 1).to(D), l8.repeat(N // 8, 1).to(D), 75, 16)
        ref = stgcn.decoder_forward({k: v.clone() for k, v in P.items()}, z8, l8, 75, 16)
    assert tuple(out.shape) == (N, 3, 75, 16, 1)
    assert_close(out[:8], ref, 1e-4, "decoder rows vs oracle")
    blocks = out.view(N // 8, 8, 3, 75, 16, 1)
    assert torch.equal(blocks, blocks[:1].expand_as(blocks)), "copies of a (z, class) row differ inside the batch"
    # the reference's own fixture rows inside a full micro-batch (positions 0.. and N-4..)
    d = sub(load_npz("cvae_eval"), "")
    k = d["z"].shape[0]
    zz = torch.cat((d["z"], z8.repeat((N - 2 * k) // 8 + 1, 1)[:N - 2 * k], d["z"]))
    ll = torch.cat((d["onehot"], l8.repeat((N - 2 * k) // 8 + 1, 1)[:N - 2 * k], d["onehot"]))
    with torch.no_grad():
        big = m.decoder(zz.to(D), ll.to(D), 75, 16)
    assert_close(big[:k], d["dec75"], 1e-4, "reference fixture rows at the head of a 2048 batch")
    assert torch.equal(big[:k], big[N - k:])


def test_full_size_hybrid_eval_f27_replicated_batch():
    """BASELINE configs[4] at its stated size and with the reference's DEFAULT feature count
    (classifier_hybrid/net/classifier.py:61-89; F = 27 = len(data_train[0][0]) of classifier_hybrid/main.py:62 without
    the stride features): N = 1024, T = 1024, eval.  8 distinct (gait, affective vector) pairs x 128 against the
    oracle on the 8 (the oracle is pinned to the reference at F = 29 by hybrid_eval_T1024.npz; the best hybrid
    checkpoint is missing, so the head is seeded random), copies bit-equal, batch-split invariance at that size."""
    from step_amd.net import classifier_hybrid
    base = ckpt("ckpt_classifier_real_only_epoch395")
    torch.manual_seed(27)
    m = classifier_hybrid.Classifier(3, 27, 4, {"strategy": "spatial"})
    sd = m.state_dict()
    for k, v in base.items():
        k2 = k.replace("data_bn.", "data_bn1.")
        if k2 in sd and not k.startswith("fcn."):
            sd[k2] = v.clone()
    g = torch.Generator().manual_seed(1024)
    sd["data_bn2.running_mean"] = torch.randn(91, generator=g) * 0.1
    sd["data_bn2.running_var"] = torch.rand(91, generator=g) + 0.5
    sd["data_bn2.weight"] = torch.rand(91, generator=g) + 0.5
    sd["data_bn2.bias"] = torch.randn(91, generator=g) * 0.1
    m.load_state_dict(sd)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(D).eval()
    x8 = torch.randn(8, 3, 1024, 16, 1, generator=g) * 0.5
    a8 = torch.rand(8, 27, generator=g) * 2 - 1
    N = 1024
    with torch.no_grad():
        ref = stgcn.classifier_forward(P, x8, "hybrid", x_aff=a8)
        xg, ag = x8.repeat(N // 8, 1, 1, 1, 1).to(D), a8.repeat(N // 8, 1).to(D)
        out = m(ag, xg)
        assert tuple(out.shape) == (N, 4)
        assert_close(out[:8], ref, 1e-4, "hybrid logits, F = 27, T = 1024")
        assert torch.equal(out.view(N // 8, 8, 4), out[:8].expand(N // 8, 8, 4))
        parts = torch.cat([m(ag[i:i + 344], xg[i:i + 344]) for i in range(0, N, 344)])
    assert torch.equal(out, parts)


def test_epoch_level_caller_train_save_reload_eval(tmp_path):
    """BASELINE configs[0] through the HIP modules: processor.py:260-268 (adjust_lr), :309-344 (per_train), :346-379
    (per_test, top-1), :381-406 (train, checkpoint naming), :447-455 (load_best_model) as step_amd.harness.EpochRunner
    on 342 synthetic gaits at T = 75, batch 8, drop_last, Adam(0.1, 1e-4), 4 epochs."""
    import os
    import re
    from step_amd import harness
    from step_amd.net.classifier import Classifier
    torch.manual_seed(0)
    train_set, test_set = harness.synthetic_gaits(342, 75, 4, seed=0)
    assert train_set[0].shape == (307, 3, 75, 16, 1) and test_set[0].shape == (35, 3, 75, 16, 1)
    run = harness.EpochRunner(Classifier(3, 4, {"strategy": "spatial"}), train_set, test_set, str(tmp_path), 4,
                              device=D, seed=0)
    assert run.steps == [2, 3, 4]
    hist = run.train()
    assert len(hist) == 4 and run.meta["iter"] == 4 * 38                    # 307 // 8 batches per epoch
    assert [round(h["lr"], 6) for h in hist] == [0.1, 0.1, 0.01, 0.001]     # ceil(4 * [0.5, 0.75, 0.875]) = 2, 3, 4
    losses = [h["train_loss"] for h in hist]
    assert all(l == l and l < 10 for l in losses) and losses[-1] < losses[0], losses
    for name, mod in run.model.named_modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            assert int(mod.num_batches_tracked) == run.meta["iter"], name
    saved = [h["saved"] for h in hist if "saved" in h]
    assert saved and all(re.fullmatch(r"epoch\d+_acc\d+\.\d\d_model\.pth\.tar", os.path.basename(p)) for p in saved)
    assert os.path.basename(saved[-1]) == harness.CKPT_FORMAT.format(run.best_epoch, run.best_accuracy.item())
    # the model as it was when the best checkpoint was written: reload by file-name parsing in a FRESH runner
    best_epoch, best_acc = run.best_epoch, run.best_accuracy.item()
    fresh = harness.EpochRunner(Classifier(3, 4, {"strategy": "spatial"}), train_set, test_set, str(tmp_path), 4,
                                device=D, seed=0)
    path = fresh.load_best_model()
    assert path == saved[-1] and fresh.best_epoch == best_epoch
    sd_file = torch.load(path, map_location=D, weights_only=True)
    for k, v in fresh.model.state_dict().items():
        assert torch.equal(v, sd_file[k]), k
    # evaluation of the reloaded model reproduces the saved accuracy (same 32 of the 35 test gaits: drop_last with a
    # shuffled order, so compare on the full, unshuffled test set in one batch instead of the loader's sample)
    fresh.model.eval()
    with torch.no_grad():
        a, _ = fresh.model(fresh.test_set[0])
        b, _ = fresh.model(fresh.test_set[0])
        one, _ = fresh.model(fresh.test_set[0][:5])
    assert torch.equal(a, b) and torch.equal(a[:5], one)                    # bit-for-bit logits after the reload
    _, accs = fresh.per_test()
    assert 0.0 <= accs[1] <= 100.0
    if best_epoch == len(hist) - 1:                                         # best == last: the live model IS the file
        run.model.eval()
        with torch.no_grad():
            live, _ = run.model(run.test_set[0])
        assert torch.equal(live, a)


@pytest.mark.parametrize("mfma", ["f32", "f32_direct"])
def test_second_eval_forward_launches_no_pack_or_prepare(monkeypatch, mfma):
    """Inference host path (the reference's batch-1 prediction loop, processor.py:457-473): packed temporal weights,
    eval-mode BatchNorm coefficients and A * importance are cached per parameter version, pooling + fcn + softmax are
    one launch.  Second no-grad eval forward: 2 launches per block + 1 head (4 per block in the frequency-domain
    default: graph conv, transform, mixing, inverse transform with the block's tail in its epilogue), none of them a
    pack / weight-spectrum / prepare kernel; an in-place parameter update or train() refreshes the caches."""
    import step_amd
    from step_amd import _lib
    from step_amd.net.classifier import Classifier
    P = ckpt("ckpt_classifier_real_only_epoch395")
    m = step_amd.set_mfma_dtype(_load(Classifier(3, 4, {"strategy": "spatial"}), P).eval(), mfma)
    pack = "step_dft_wspec" if mfma == "f32" else "step_tcn_pack_weight"
    conv = ["step_dft_fwd", "step_dft_mix", "step_dft_inv"] if mfma == "f32" else ["step_tcn_conv"]
    x = torch.randn(3, 3, 75, 16, 1, generator=torch.Generator().manual_seed(3)).to(D)
    names = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (names.append(name), real(name, *a))[1])
    with torch.no_grad():
        o1, f1 = m(x)
        first = list(names)
        del names[:]
        o2, f2 = m(x)
    assert any(n.startswith(pack) for n in first) and "step_bn_eval_prepare" in first
    assert sorted(names) == sorted((["step_gcn_fwd"] + conv) * 3 + ["step_head_fwd"]), names
    assert torch.equal(o1, o2) and torch.equal(f1, f2)
    ref, rfeat = stgcn.classifier_forward({k: v.clone() for k, v in P.items()}, x.cpu(), "real_only")
    assert_close(o2, ref, 1e-4, "cached eval probs")
    assert_close(f2, rfeat, 1e-4, "cached eval feat")
    # the one-launch head against the three-launch autograd path: same sums, bit for bit
    o3, f3 = m(x)
    assert o3.requires_grad
    assert_close(o2, o3.detach(), 1e-5, "fused epilogue vs separate pass")
    # an in-place update (what optimizer.step / load_state_dict do) is seen through the tensor version ...
    with torch.no_grad():
        m.st_gcn_networks[1].tcn[2].weight.mul_(1.5)
        m.st_gcn_networks[2].tcn[3].running_var.mul_(2.0)
        m.edge_importance[0].mul_(0.5)
        del names[:]
        o4, _ = m(x)
    assert names.count(pack) == 1 and names.count("step_bn_eval_prepare") == 1
    P2 = {k: v.clone() for k, v in m.state_dict().items()}
    ref4, _ = stgcn.classifier_forward({k: v.cpu() for k, v in P2.items()}, x.cpu(), "real_only")
    assert_close(o4, ref4, 1e-4, "after in-place parameter updates")
    # ... and train() / eval() drop every cache (weights_init writes through .data, which bumps no version)
    m.train()
    m.eval()
    with torch.no_grad():
        del names[:]
        m(x)
    assert names.count(pack) == 3


def test_gradient_saliency_vs_reference_hook():
    """`--smap` (processor.py:434-445): step_amd.saliency.GuidedBackprop against tests/golden/saliency.npz - the
    reference's own Classifier (checkpoint 395, eval) with processor.py's hook on st_gcn_networks[0].gcn.conv and its
    one-hot backward, on 8 shipped gaits (T = 18) and 4 random ones (T = 75): the gradient at the first graph conv's
    input, whole batch; the call returns sample 0 as the reference does."""
    from step_amd.net.classifier import Classifier
    from step_amd.saliency import GuidedBackprop
    m = _load(Classifier(3, 4, {"strategy": "spatial"}), ckpt("ckpt_classifier_real_only_epoch395"))
    d = sub(load_npz("saliency"), "")
    gbp = GuidedBackprop(m)
    assert not m.training
    for tag in ("h5", "rnd"):
        first = gbp.generate_gradients(d[tag + "_x"].to(D), d[tag + "_labels"])
        ref = d[tag + "_grad"]
        assert_close(gbp.gradients, ref, 1e-3, tag + " saliency")
        assert first.shape == tuple(ref.shape[1:])
        assert_close(torch.from_numpy(first), ref[0], 1e-3, tag + " saliency, first sample")
