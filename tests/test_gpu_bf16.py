"""GPU: the bf16-operand temporal kernels (BASELINE config 'real_and_synth training, bf16').

Two bars per case: (a) against the oracle run on operands rounded to bf16 the way the kernel rounds
them (after BN + ReLU; weights) the result must agree to fp32 round-off (1e-4) - this pins the
kernel's indexing and accumulation exactly; (b) against the plain fp32 oracle it must stay within
north_star's bf16 tolerance, 1e-2 relative."""
import pytest
import torch

from oracle import stgcn
from tests.util import assert_close

pytestmark = pytest.mark.gpu
TOL_EXACT = 1e-4
TOL_BF16 = 1e-2


def dev():
    return torch.device("cuda:0")


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("C,stride,T,N", [
    (64, 1, 48, 2), (64, 1, 23, 3), (32, 1, 96, 2), (32, 1, 17, 1), (64, 2, 41, 2), (64, 2, 80, 1),
    (64, 1, 240, 1), (64, 2, 240, 1), (32, 1, 240, 1),
])
def test_tcn_bf16_forward_and_epilogue(C, stride, T, N):
    from step_amd import ops
    Kt = 75
    assert ops.tcn_bf16_supported(C, C, Kt, stride, 0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, T, 16, generator=g)
    w = torch.randn(C, C, Kt, 1, generator=g) * 0.05
    b = torch.randn(C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    To = (T - 1) // stride + 1
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref_exact = stgcn.temporal_conv(rb(a), rb(w), b, stride)
    ref_f32 = stgcn.temporal_conv(a, w, b, stride)
    d = dev()
    wp = ops.tcn_pack(w.to(d), C, C, Kt, stride, 0, bf16=True)
    out = ops.tcn_conv(x.to(d), sc.to(d), sh.to(d), True, wp, b.to(d), None, None, None, None, None,
                       False, N, C, C, T, To, Kt, stride, 0, bf16=True)
    assert_close(out, ref_exact, TOL_EXACT, "bf16 tcn vs bf16-rounded oracle")
    assert_close(out, ref_f32, TOL_BF16, "bf16 tcn vs fp32 oracle")
    osc, osh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    res = torch.randn(N, C, To, 16, generator=g)
    rsc, rsh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    v = lambda t: t.view(1, -1, 1, 1)
    ref2 = torch.relu(ref_exact * v(osc) + v(osh) + res * v(rsc) + v(rsh))
    out2 = ops.tcn_conv(x.to(d), sc.to(d), sh.to(d), True, wp, b.to(d), osc.to(d), osh.to(d),
                        res.to(d), rsc.to(d), rsh.to(d), True, N, C, C, T, To, Kt, stride, 0, bf16=True)
    assert_close(out2, ref2, TOL_EXACT, "bf16 tcn fused epilogue")


@pytest.mark.parametrize("C,stride,T,N", [(64, 1, 33, 2), (32, 1, 50, 2), (64, 2, 41, 2), (64, 2, 40, 1),
                                          (64, 2, 240, 1), (64, 1, 120, 1)])
def test_tcn_bf16_dgrad(C, stride, T, N):
    from step_amd import ops
    Kt = 75
    assert ops.tcn_bf16_supported(C, C, Kt, stride, 2)
    g = torch.Generator().manual_seed(12)
    To = (T - 1) // stride + 1
    w = torch.randn(C, C, Kt, 1, generator=g) * 0.05
    du = torch.randn(N, C, To, 16, generator=g)

    def dgrad(wt, dut):
        a = torch.zeros(N, C, T, 16, requires_grad=True)
        stgcn.temporal_conv(a, wt, None, stride).backward(dut)
        return a.grad

    d = dev()
    wpd = ops.tcn_pack(w.to(d), C, C, Kt, stride, 2, bf16=True)
    da = ops.tcn_conv(du.to(d), None, None, False, wpd, None, None, None, None, None, None, False,
                      N, C, C, T, To, Kt, stride, 2, bf16=True)
    assert_close(da, dgrad(rb(w), rb(du)), TOL_EXACT, "bf16 dgrad vs bf16-rounded oracle")
    assert_close(da, dgrad(w, du), TOL_BF16, "bf16 dgrad vs fp32 oracle")


@pytest.mark.parametrize("C,stride,T,N", [(64, 1, 33, 2), (64, 1, 16, 5), (32, 1, 50, 3), (64, 2, 41, 2),
                                          (64, 2, 40, 3), (64, 1, 240, 1), (32, 1, 240, 2), (64, 2, 240, 1),
                                          (64, 1, 7, 300), (32, 2, 21, 2)])
def test_tcn_bf16_wgrad(C, stride, T, N):
    from step_amd import ops
    Kt = 75
    assert ops.tcn_wgrad_bf16_supported(C, C, Kt, stride)
    g = torch.Generator().manual_seed(13)
    To = (T - 1) // stride + 1
    y = torch.randn(N, C, T, 16, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    du = torch.randn(N, C, To, 16, generator=g)
    a = torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))

    def wgrad(at, dut):
        w = torch.zeros(C, C, Kt, 1, requires_grad=True)
        b = torch.zeros(C, requires_grad=True)
        stgcn.temporal_conv(at, w, b, stride).backward(dut)
        return w.grad, b.grad

    d = dev()
    dw, db = ops.tcn_wgrad(du.to(d), y.to(d), sc.to(d), sh.to(d), True, N, C, C, T, To, Kt, stride, False,
                           (C, C, Kt, 1), bf16=True)
    dw_e, _ = wgrad(rb(a), rb(du))
    dw_f, db_f = wgrad(a, du)
    assert_close(dw, dw_e, TOL_EXACT, "bf16 wgrad vs bf16-rounded oracle")
    assert_close(dw, dw_f, TOL_BF16, "bf16 wgrad vs fp32 oracle")
    assert_close(db, db_f, TOL_EXACT, "bias gradient (exact fp32 sums)")


def test_tcn_bf16_unsupported_shapes_say_so():
    from step_amd import ops
    assert not ops.tcn_bf16_supported(3, 3, 75, 1, 0)
    assert not ops.tcn_bf16_supported(64, 64, 9, 1, 0)
    assert ops.tcn_bf16_supported(64, 64, 75, 1, 1)              # ConvTranspose2d forward: stride 1 only
    assert not ops.tcn_bf16_supported(64, 64, 75, 2, 1)
    assert not ops.tcn_wgrad_bf16_supported(3, 3, 75, 1)
    assert not ops.tcn_wgrad_bf16_supported(64, 64, 9, 1)


# ----------------------------------------------------------------------- whole models, bf16 MFMA
def _load_models_helpers():
    from tests.util import load_npz, sub, ckpt
    return load_npz, sub, ckpt


def test_bf16_classifier_eval_within_1e2_and_same_labels():
    """north_star: logits within 1e-2 relative in bf16, identical classification."""
    import step_amd
    from step_amd.net import classifier, classifier_real_and_synth
    load_npz, sub, ckpt = _load_models_helpers()
    D = dev()
    m = classifier.Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
    m = step_amd.set_mfma_dtype(m.to(D).eval(), "bf16")
    d = sub(load_npz("classifier_real_only_eval"), "")
    with torch.no_grad():
        probs, feat = m(d["h5_x"].to(D))
        assert_close(probs, d["h5_probs"], TOL_BF16, "bf16 h5 probs")
        assert_close(feat, d["h5_feat"], TOL_BF16, "bf16 h5 feat")
        assert torch.equal(probs.argmax(1).cpu(), d["h5_probs"].argmax(1))
        probs, feat = m(d["rnd_x"].to(D))
        assert_close(probs, d["rnd_probs"], TOL_BF16, "bf16 rnd probs")
        assert_close(feat, d["rnd_feat"], TOL_BF16, "bf16 rnd feat")

    base = ckpt("ckpt_classifier_real_only_epoch395")
    d = sub(load_npz("real_and_synth_eval_T240"), "")
    P = {k: v for k, v in base.items() if not k.startswith("st_gcn_networks.0.residual")}
    P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    m = classifier_real_and_synth.Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = step_amd.set_mfma_dtype(m.to(D).eval(), "bf16")
    with torch.no_grad():
        o, f = m(d["x"].to(D))
    assert_close(o, d["out"], TOL_BF16, "bf16 real_and_synth logits")
    assert_close(f, d["feat"], TOL_BF16, "bf16 real_and_synth feat")


def _check_grads_exact(m, grads, rtol=1e-3):
    from tests.util import assert_grad_close, feeds_train_bn
    n = 0
    for k, p in m.named_parameters():
        assert_grad_close(p.grad, grads[k], rtol, k, 1e-5 if feeds_train_bn(k) else 1e-6)
        n += 1
    return n


def test_bf16_train_step_kt75_vs_bf16_operand_oracle():
    """One training step (ckpt395, Kt=75, train-mode BN) with bf16 MFMA operands.  Bar (a): against the oracle
    that rounds the temporal conv's operands to bf16 where the kernels do (oracle.stgcn tcn_bf16=True: a after
    BN1+ReLU, the weights, dout in dgrad/wgrad) loss, outputs and EVERY parameter gradient agree to fp32
    round-off (1e-4 / 1e-3, the fp32 step's own bars).  Bar (b): against the reference's fp32 fixture the loss
    and outputs stay within north_star's 1e-2."""
    import torch.nn.functional as F
    import step_amd
    from step_amd.net import classifier
    load_npz, sub, ckpt = _load_models_helpers()
    D = dev()
    d = sub(load_npz("train_steps"), "real_only_k75_ckpt395.")
    P = ckpt("ckpt_classifier_real_only_epoch395")
    loss_ref, out_ref, grads = stgcn.classifier_loss_and_grads({k: v.clone() for k, v in P.items()}, d["x"],
                                                               d["labels"], "real_only", tcn_bf16=True)
    m = classifier.Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = step_amd.set_mfma_dtype(m.to(D).train(), "bf16")
    out, _ = m(d["x"].to(D))
    loss = F.cross_entropy(out, d["labels"].to(D))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) < 2e-5
    assert_close(out, out_ref, TOL_EXACT, "bf16 train out vs bf16-operand oracle")
    assert _check_grads_exact(m, grads) >= 30
    assert abs(float(loss.detach()) - float(d["loss"])) < 1e-2 * max(1.0, abs(float(d["loss"])))
    assert_close(out, d["out"], TOL_BF16, "bf16 train out vs fp32 reference")


def test_bf16_real_and_synth_train_step_per_gpu_batch_512():
    """BASELINE configs[2] as stated: classifier_stgcn_real_and_synth (classifier.py:40-51,65-89 of that app),
    bf16 MFMA operands, TRAIN mode, per-GPU batch 512 x 3 x 240 x 16.  The batch is 64 copies of 8 gaits: same
    batch statistics, same mean CrossEntropy and the same parameter gradients as the 8 gaits alone, which the
    bf16-operand oracle does in seconds.  Loss / logits to 1e-4, every gradient to 1e-3 against that oracle;
    logits within 1e-2 of the fp32 oracle (north_star's bf16 bar)."""
    import torch.nn.functional as F
    import step_amd
    from step_amd.net import classifier_real_and_synth
    load_npz, sub, ckpt = _load_models_helpers()
    D = dev()
    N = 512
    P = ckpt("ckpt_classifier_real_only_epoch395")
    d = sub(load_npz("real_and_synth_eval_T240"), "")
    P = {k: v for k, v in P.items() if not k.startswith("st_gcn_networks.0.residual")}
    P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    g = torch.Generator().manual_seed(N)
    x8 = torch.randn(8, 3, 240, 16, 1, generator=g) * 0.5
    l8 = torch.arange(8) % 4
    loss_ref, out_ref, grads = stgcn.classifier_loss_and_grads({k: v.clone() for k, v in P.items()}, x8, l8,
                                                               "real_and_synth", tcn_bf16=True)
    _, out_f32, _ = stgcn.classifier_loss_and_grads({k: v.clone() for k, v in P.items()}, x8, l8, "real_and_synth")
    m = classifier_real_and_synth.Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = step_amd.set_mfma_dtype(m.to(D).train(), "bf16")
    from step_amd import ops
    for b in m.st_gcn_networks:                                   # every temporal conv really is on the bf16 kernels
        _, co, kt, st = b._geom
        assert b.mfma_dtype == "bf16" and ops.tcn_bf16_supported(co, co, kt, st, 0) \
            and ops.tcn_bf16_supported(co, co, kt, st, 2) and ops.tcn_wgrad_bf16_supported(co, co, kt, st)
    out, _ = m(x8.repeat(N // 8, 1, 1, 1, 1).to(D))
    loss = F.cross_entropy(out, l8.repeat(N // 8).to(D))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) < 2e-5
    assert_close(out[:8], out_ref, TOL_EXACT, "bf16 train logits vs bf16-operand oracle")
    assert torch.equal(out[:8], out[N - 8:])
    assert_close(out[:8], out_f32, TOL_BF16, "bf16 train logits vs fp32 oracle")
    assert _check_grads_exact(m, grads) >= 30


def test_bf16_cvae_train_step_vs_bf16_operand_oracle():
    """The bf16 mode on the CVAE (st_gctn blocks: ConvTranspose2d temporal convs, generator_cvae/net/CVAE_stgcn.py:340-394):
    pack modes 1 / 3 of the bf16 conv kernels and the transposed bf16 weight gradient (step_tcn_wgrad_bf16_t).  Training
    forward + backward of encoder -> reparameterisation (the fixture's eps) -> decoder on the reference's cvae_train
    inputs with the shipped epoch-105 weights, against the oracle with bf16-rounded temporal-conv operands in every
    block the kernels cover (square 32 / 64 channels; the decoder's last 3 -> 3 conv stays fp32 in both): fp32 bars.
    And within north_star's 1e-2 of the reference's fp32 fixture."""
    import step_amd
    from step_amd.net.CVAE_stgcn import CVAE
    from step_amd import ops
    from tests.util import assert_grad_close, feeds_train_bn
    load_npz, sub, ckpt = _load_models_helpers()
    D = dev()
    P = ckpt("ckpt_cvae_epoch105")
    d = sub(load_npz("cvae_train"), "")
    T = d["x"].shape[2]
    keys = stgcn.trainable_keys(P)
    leaves = {k: P[k].detach().clone().requires_grad_(True) for k in keys}
    Q = {k: v.clone() for k, v in P.items()}
    Q.update(leaves)
    rec_r, mean_r, lsig_r, _ = stgcn.cvae_forward(Q, d["x"], d["lenc"], d["onehot"], d["eps"], T, 16, training=True,
                                                  tcn_bf16=lambda c: c in (32, 64))
    loss_r = ((rec_r - d["x"]) ** 2).mean() - 0.5 * torch.mean(1 + lsig_r - mean_r.pow(2) - lsig_r.exp())
    gl = torch.autograd.grad(loss_r, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: g for k, g in zip(keys, gl) if g is not None}

    m = CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    m.load_state_dict(P)
    m = step_amd.set_mfma_dtype(m.to(D).train(), "bf16")
    nb = 0
    for blk in list(m.encoder.encoder) + list(m.decoder.decoder):
        _, co, kt, st = blk._geom
        if co in (32, 64):
            tr = blk.__class__.__name__ == "st_gctn"
            assert blk.mfma_dtype == "bf16" and ops.tcn_bf16_supported(co, co, kt, st, 1 if tr else 0) \
                and ops.tcn_bf16_supported(co, co, kt, st, 3 if tr else 2)
            nb += 1
    assert nb == 5
    dd = {k: v.to(D) for k, v in d.items() if k in ("x", "lenc", "onehot", "eps")}
    mean, lsig = m.encoder(dd["x"], dd["lenc"])
    zz = dd["eps"] * torch.exp(0.5 * lsig) + mean
    rec = m.decoder(zz, dd["onehot"], T, 16)
    loss = ((rec - dd["x"]) ** 2).mean() - 0.5 * torch.mean(1 + lsig - mean.pow(2) - lsig.exp())
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) < 2e-5
    assert_close(mean, mean_r.detach(), TOL_EXACT, "bf16 CVAE mean vs bf16-operand oracle")
    assert_close(rec, rec_r.detach(), TOL_EXACT, "bf16 CVAE reconstruction vs bf16-operand oracle")
    n = 0
    for k, p in m.named_parameters():
        if k in grads:
            assert_grad_close(p.grad, grads[k], 1e-3, k, 1e-5 if feeds_train_bn(k) else 1e-6)
            n += 1
    assert n >= 40
    assert_close(mean, d["mean"], TOL_BF16, "bf16 CVAE mean vs fp32 reference")
    assert_close(rec, d["rec"], TOL_BF16, "bf16 CVAE reconstruction vs fp32 reference")


@pytest.mark.parametrize("C,stride,T,N,tr,split", [(32, 1, 33, 3, False, 0), (64, 2, 41, 2, False, 0), (64, 1, 240, 5, False, 0),
                                                   (64, 2, 240, 3, False, 1), (32, 1, 24, 2, True, 0), (64, 1, 75, 300, True, 1),
                                                   (64, 1, 120, 300, False, 0)])
def test_tcn_conv_stats_epilogue_bf16_and_split(C, stride, T, N, tr, split):
    """step_tcn_conv_stats_bf16: u bit-equal to step_tcn_conv_bf16's / step_tcn_conv_split's, [SUM u, SUM u^2] per channel
    equal to float64 sums of u (ragged T, stride 2, ConvTranspose packing, more tiles than blocks so that the per-wave
    fp64 accumulators carry over tiles), bit-identical over two calls"""
    from step_amd import _lib, ops
    d = dev()
    g = torch.Generator().manual_seed(C + T + N)
    To = (T - 1) // stride + 1
    y = torch.randn(N, C, T, 16, generator=g).to(d)
    w = (torch.randn(C, C, 75, 1, generator=g) * 0.05).to(d)
    b = torch.randn(C, generator=g).to(d)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(d), torch.randn(C, generator=g).to(d) * 0.3
    mode = 1 if tr else 0
    bf = "split" if split else True
    wp = ops.tcn_pack(w, C, C, 75, stride, mode, bf16=bf)
    ref = ops.tcn_conv(y, sc, sh, True, wp, b, None, None, None, None, None, False, N, C, C, T, To, 75, stride, mode, bf16=bf)
    nb = _lib.query("step_tcn_conv_stats_bf16_ws_bytes", C)
    res = []
    for _ in range(2):
        u = torch.full((N, C, To, 16), float("nan"), device=d)
        sums = torch.zeros(2 * C + 1, dtype=torch.float64, device=d)
        ws = torch.empty(nb, dtype=torch.uint8, device=d)
        _lib.call("step_tcn_conv_stats_bf16", y, sc, sh, 1, wp, b, u, N, C, C, T, To, 75, stride, mode, split, sums, ws, nb)
        res.append((u, sums))
    assert torch.equal(res[0][0], ref) and torch.equal(res[0][1], res[1][1])
    ud = ref.double()
    got = res[0][1][:2 * C].view(C, 2)
    assert float((got[:, 0] - ud.sum(dim=(0, 2, 3))).abs().max()) <= 2e-6 * float(ud.abs().sum(dim=(0, 2, 3)).max())
    assert float(((got[:, 1] - (ud * ud).sum(dim=(0, 2, 3))) / (ud * ud).sum(dim=(0, 2, 3))).abs().max()) <= 2e-6
