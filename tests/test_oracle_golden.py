"""CPU: the oracle (oracle/) against every golden vector produced by the real reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import graph_adjacency, stgcn
from tests.util import load_npz, sub, ckpt, assert_close, assert_grad_close, feeds_train_bn

F32 = 2e-5      # fp32 restatement vs fp32 reference: op-order noise only


def test_graph_all_strategies():
    g = load_npz("graph")
    for strat in ("uniform", "distance", "spatial"):
        np.testing.assert_array_equal(graph_adjacency(strat), g[strat])
    np.testing.assert_array_equal(graph_adjacency("spatial", max_hop=2), g["spatial_hop2"])
    np.testing.assert_array_equal(graph_adjacency("distance", max_hop=2), g["distance_hop2"])
    A = graph_adjacency("spatial")
    assert A.shape == (3, 16, 16) and int((A != 0).sum()) == 46
    np.testing.assert_allclose(A.sum(axis=(0, 1)), 1.0)


def test_product_graph_all_strategies():
    """The PRODUCT's Graph (step_amd/net/utils/graph.py; host numpy, no GPU needed) against every array the
    reference's Graph produced (net/utils/graph.py:26-100): strategies uniform / distance / spatial, max_hop 1 and 2."""
    from step_amd.net.utils.graph import Graph
    g = load_npz("graph")
    for strat in ("uniform", "distance", "spatial"):
        np.testing.assert_array_equal(Graph(strategy=strat).A, g[strat])
    np.testing.assert_array_equal(Graph(strategy="spatial", max_hop=2).A, g["spatial_hop2"])
    np.testing.assert_array_equal(Graph(strategy="distance", max_hop=2).A, g["distance_hop2"])
    assert Graph(strategy="spatial").A.dtype == g["spatial"].dtype
    with pytest.raises(ValueError):
        Graph(strategy="nope")


@pytest.mark.parametrize("tag", ["c3", "c1"])
def test_assemble_oracle_vs_reference_lines(tag):
    """utils/loader.py:33-37 (pad / tile / cycles, then the alignment) - the reference's own statements were run
    on ragged cuts of the shipped gaits by make_golden.gen_assemble"""
    from oracle import pose
    z = load_npz("assemble")
    lens = z[tag + ".lens"]
    frames = z[tag + ".frames"]
    cuts = np.concatenate([[0], np.cumsum(lens)])
    data_list = [frames[cuts[i]:cuts[i + 1]] for i in range(len(lens))]
    with np.errstate(all="ignore"):
        aligned, padded = pose.assemble(data_list, int(z[tag + ".cycles"]))
    np.testing.assert_array_equal(padded, z[tag + ".padded"])
    np.testing.assert_allclose(aligned, z[tag + ".aligned"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", ["f64_small", "f32_b0", "f32_b1", "T_f64_small", "T_f32_b0", "T_f32_b1"])
def test_graph_conv_fwd_bwd(tag):
    d = sub(load_npz("gcn"), tag + ".")
    tr = tag.startswith("T_")
    x = d["x"].clone().requires_grad_(True)
    A = d["A"].clone().requires_grad_(True)
    w = d["w"].clone().requires_grad_(True)
    b = d["b"].clone().requires_grad_(True)
    y = stgcn.conv_temporal_graphical(x, A, w, b, transposed=tr)
    tol = 1e-12 if "f64" in tag else F32
    assert_close(y, d["y"], tol, "y")
    y.backward(d["gy"])
    for got, key in ((x.grad, "gx"), (A.grad, "gA"), (w.grad, "gw"), (b.grad, "gb")):
        assert_close(got, d[key], tol * 5, key)


def _block_cases():
    out = []
    for f in ("st_gcn_blocks", "st_gctn_blocks"):
        tags = sorted({k.split(".")[0] for k in load_npz(f).files})
        out += [(f, t) for t in tags]
    return out


def block_meta(tag):
    parts = tag.split("_")
    transposed = parts[0] == "t"
    cin, cout = int(parts[1]), int(parts[2])
    stride = 2 if "s2" in parts else 1
    residual = "nores" not in parts
    return transposed, cin, cout, stride, residual


@pytest.mark.parametrize("fname,tag", _block_cases())
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_st_gcn_block(fname, tag, mode):
    z = load_npz(fname)
    transposed, cin, cout, stride, residual = block_meta(tag)
    P = sub(z, tag + ".P0.")
    P = {"blk." + k: v.clone() for k, v in P.items()}
    d = sub(z, "%s.%s." % (tag, mode))
    keys = [k for k in P if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    for k in keys:
        P[k].requires_grad_(True)
    x = d["x"].clone().requires_grad_(True)
    A = d["A"].clone().requires_grad_(True)
    y = stgcn.st_gcn_block(x, A, P, "blk", stride, residual, mode == "train", transposed)
    assert_close(y, d["y"], 5e-5, "y")
    y.backward(d["gy"])
    assert_close(x.grad, d["gx"], 2e-4, "gx")
    assert_close(A.grad, d["gA"], 2e-4, "gA")
    for k in keys:
        name = k[4:]
        g = P[k].grad
        if "G." + name in d:
            noise = 1e-7 * float(d["gy"].abs().sum()) if (mode == "train" and feeds_train_bn(name)) else 1e-6
            assert_grad_close(g, d["G." + name], 2e-4, name, noise)
        else:
            assert_close(g[:2], d["Gsl." + name], 3e-4, "Gsl " + name)
            assert abs(float(g.double().norm()) - float(d["Gnorm." + name])) <= 3e-4 * float(d["Gnorm." + name])
    for k, v in d.items():
        if k.startswith("P1."):
            assert_close(P["blk." + k[3:]], v, 2e-5, k)


def test_classifier_real_only_ckpt395_eval():
    P = ckpt("ckpt_classifier_real_only_epoch395")
    d = sub(load_npz("classifier_real_only_eval"), "")
    with torch.no_grad():
        probs, feat = stgcn.classifier_forward(P, d["h5_x"], "real_only")
        assert_close(probs, d["h5_probs"], 1e-4, "h5 probs")
        assert_close(feat, d["h5_feat"], 1e-4, "h5 feat")
        assert torch.equal(probs.argmax(1), d["h5_probs"].argmax(1))
        probs, feat = stgcn.classifier_forward(P, d["rnd_x"], "real_only")
        assert_close(probs, d["rnd_probs"], 1e-4, "rnd probs")
        assert_close(feat, d["rnd_feat"], 1e-4, "rnd feat")
        o, f = stgcn.classifier_extract_feature(P, d["rnd_x"][:2], "real_only")
        assert_close(o, d["ef_out"], 1e-4, "extract_feature out")
        assert_close(f, d["ef_feat"], 1e-4, "extract_feature feat")


def test_real_and_synth_and_hybrid_eval_kt75():
    base = ckpt("ckpt_classifier_real_only_epoch395")
    d = sub(load_npz("real_and_synth_eval_T240"), "")
    P = {k: v for k, v in base.items() if not k.startswith("st_gcn_networks.0.residual")}
    P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    with torch.no_grad():
        o, f = stgcn.classifier_forward(P, d["x"], "real_and_synth")
    assert_close(o, d["out"], 1e-4, "logits")
    assert_close(f, d["feat"], 1e-4, "feat")

    d = sub(load_npz("hybrid_eval_T1024"), "")
    P = {k.replace("data_bn.", "data_bn1."): v for k, v in base.items()
         if not k.startswith(("st_gcn_networks.0.residual", "fcn."))}
    P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
    with torch.no_grad():
        o = stgcn.classifier_forward(P, d["x_gait"], "hybrid", x_aff=d["x_aff"])
    assert_close(o, d["out"], 1e-4, "hybrid logits")


@pytest.mark.parametrize("variant", ["real_only", "real_and_synth", "hybrid"])
def test_train_step_kt9(variant):
    z = load_npz("train_steps")
    p = variant + "_k9."
    d = sub(z, p)
    P = {k[3:]: v.clone() for k, v in d.items() if k.startswith("P0.")}
    loss, out, grads = stgcn.classifier_loss_and_grads(P, d["x"], d["labels"], variant,
                                                       x_aff=d.get("x_aff"))
    assert abs(float(loss) - float(d["loss"])) < 1e-5
    assert_close(out, d["out"], 1e-4, "out")
    for k, g in grads.items():
        assert_grad_close(g, d["G." + k], 5e-4, k, 1e-6 if feeds_train_bn(k) else 1e-7)
    # Adam(lr=0.1, wd=1e-4), one step; BN buffers were updated by the forward above
    params = {k: P[k] for k in grads}
    stgcn.adam_step(params, grads, {}, lr=0.1, weight_decay=1e-4, step=1)
    for k in P:
        if k.endswith("num_batches_tracked"):
            assert int(P[k]) == int(d["P1." + k])
        elif k in grads:
            # Adam's first step is lr*sign(g) wherever |g| >> eps: compare where the
            # reference gradient is well away from zero
            ref1, ref0 = d["P1." + k], d["P0." + k]
            mask = d["G." + k].abs() > 1e-6
            assert float(((P[k] - ref1).abs() * mask).max()) < 2e-3, k
        else:
            assert_grad_close(P[k], d["P1." + k], 1e-4, k, 1e-7)


def test_train_step_kt75_ckpt395():
    z = load_npz("train_steps")
    d = sub(z, "real_only_k75_ckpt395.")
    P = ckpt("ckpt_classifier_real_only_epoch395")
    loss, out, grads = stgcn.classifier_loss_and_grads(P, d["x"], d["labels"], "real_only")
    assert abs(float(loss) - float(d["loss"])) < 1e-5
    assert_close(out, d["out"], 1e-4, "out")
    for k, g in grads.items():
        if "G." + k in d:
            assert_grad_close(g, d["G." + k], 1e-3, k, 1e-6 if feeds_train_bn(k) else 1e-7)
        else:
            assert_close(g[:2], d["Gsl." + k], 1e-3, k)
            n = float(d["Gnorm." + k])
            assert abs(float(g.double().norm()) - n) <= 1e-3 * n, k
    for k, v in d.items():
        if k.startswith("P1.") and "running" in k:
            assert_close(P[k[3:]], v, 1e-4, k)


def test_cvae_ckpt105():
    P = ckpt("ckpt_cvae_epoch105")
    d = sub(load_npz("cvae_eval"), "")
    with torch.no_grad():
        mean, lsig = stgcn.encoder_forward(P, d["x"], d["lenc"])
        assert_close(mean, d["mean"], 1e-4, "mean")
        assert_close(lsig, d["lsig"], 1e-4, "lsig")
        dec = stgcn.decoder_forward(P, d["z"], d["onehot"], 75, 16)
        assert_close(dec, d["dec75"], 1e-4, "dec75")
        dec = stgcn.decoder_forward(P, d["z"][:2], d["onehot"][:2], 240, 16)
        assert_close(dec, d["dec240"], 1e-4, "dec240")


def test_cvae_train_fwd_bwd():
    P = ckpt("ckpt_cvae_epoch105")
    d = sub(load_npz("cvae_train"), "")
    keys = stgcn.trainable_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    T = d["x"].shape[2]
    rec, mean, lsig, z = stgcn.cvae_forward(P, d["x"], d["lenc"], d["onehot"], d["eps"], T, 16,
                                            training=True)
    assert_close(mean, d["mean"], 1e-4, "mean")
    assert_close(rec, d["rec"], 1e-4, "rec")
    loss = ((rec - d["x"]) ** 2).mean() - 0.5 * torch.mean(1 + lsig - mean.pow(2) - lsig.exp())
    assert abs(float(loss) - float(d["loss"])) < 1e-5
    loss.backward()
    for k in keys:
        g = P[k].grad
        if "G." + k in d:
            assert_grad_close(g, d["G." + k], 1e-3, k, 1e-6 if feeds_train_bn(k) else 1e-7)
        elif "Gsl." + k in d:
            assert_close(g[:2], d["Gsl." + k], 1e-3, k)


# ---------------------------------------------------------------- CVAE loss
@pytest.mark.parametrize("tag", list("abcde"))
def test_vae_loss_oracle_vs_reference(tag):
    """between_frame_loss + KLD (losses.py:7-42, processor_stgcn.py:34-40): value and gradients"""
    from oracle import losses
    d = sub(load_npz("vae_loss"), tag + ".")
    x_out = d["x_out"].clone().requires_grad_(True)
    mean = d["mean"].clone().requires_grad_(True)
    lsig = d["lsig"].clone().requires_grad_(True)
    bf = losses.between_frame_loss(d["x_in"], x_out)
    loss = losses.vae_loss(d["x_in"], x_out, mean, lsig, 1.0)
    loss.backward()
    assert abs(float(bf) - float(d["bf"])) <= 2e-6 * max(1.0, abs(float(d["bf"])))
    assert abs(float(loss) - float(d["loss"])) <= 2e-6 * max(1.0, abs(float(d["loss"])))
    assert_close(x_out.grad, d["d_x_out"], 1e-5, "d_x_out")
    assert_close(mean.grad, d["d_mean"], 1e-6, "d_mean")
    assert_close(lsig.grad, d["d_lsig"], 1e-6, "d_lsig")
    if d["x_in"].shape[2] <= 3:     # the reference's own statement order, term by term
        ref = losses.between_frame_loss_loops(d["x_in"], d["x_out"])
        assert abs(float(ref) - float(d["bf"])) <= 1e-6 * max(1.0, abs(float(d["bf"])))


# ------------------------------------------------------------- pose alignment
@pytest.mark.parametrize("tag", ["a", "b"])
def test_pose_alignment_oracle_and_host_transforms_vs_reference(tag):
    """common.py:91-112 (first 48 features) + get_transformation :230-279"""
    from oracle import pose
    from step_amd import pipeline
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_align.npz"))
    gaits, ref = z[tag + ".gaits"], z[tag + ".aligned"]
    assert np.abs(pose.align_gaits(gaits) - ref).max() < 1e-12
    # the product's batched host part (one SVD call for all gaits) gives the same transforms
    M = pipeline.similarity_transforms(gaits)
    for s in range(gaits.shape[0]):
        R, c, t = pose.get_transformation(pose.get_joints(gaits, s, 0).T, pose.get_joints(gaits, 0, 0).T)
        assert np.abs(M[s, :, :3] - c * R).max() < 1e-10 and np.abs(M[s, :, 3] - t).max() < 1e-10


# -------------------------------------------------------- affective features
def _same(got, ref, rtol):
    """equal where the reference is finite (to rtol of the column scale), same NaN / +-inf elsewhere"""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~fin & ~np.isnan(ref)],
                                                                         ref[~fin & ~np.isnan(ref)])
    scale = np.maximum(np.abs(np.where(fin, ref, 0.0)).max(axis=0, keepdims=True), 1e-30)
    err = np.where(fin, np.abs(np.where(fin, got, 0.0) - np.where(fin, ref, 0.0)) / scale, 0.0)
    assert err.max() <= rtol, "max error / column scale = %.3e > %.1e" % (err.max(), rtol)


@pytest.mark.parametrize("tag", ["h5", "walk", "short", "mono"])
def test_affective_features_oracle_vs_reference(tag):
    """compute_features.py:428-468 and normalize_features.py:1-22 on the shipped 60 gaits and on seeded
    gaits that reach every branch (plateaus and ties of the strike rule, a foot that never strikes,
    T = 4, a frame whose box is not a bounding box)"""
    from oracle import affective
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "affective.npz"))
    g, ts = z[tag + ".gaits"], 1.0 / 30.0
    _same(affective.compute_features_batch(g, ts), z[tag + ".f27"], 1e-10)
    f29 = affective.compute_features_batch(g, ts, True)
    _same(f29, z[tag + ".f29"], 1e-10)
    assert np.array_equal(f29[:, 27], z[tag + ".f29"][:, 27])          # stride length: exact
    _same(affective.normalize_features(z[tag + ".f29"]), z[tag + ".norm29"], 1e-12)


def trajectory_state_check(P, d):
    """final state_dict of the 5-step run against the reference's.  Tolerance per tensor = 1e-3 of its largest
    entry + 8 x `noise.*` of the fixture = the largest distance between the reference's own fp32 run and five
    re-runs of it (one in float64, four in fp32 with the inputs moved by 1e-7 relative): at lr = 0.1 Adam turns
    round-off-level gradients into +-lr steps, so that spread - not fp32 epsilon - is how reproducible the
    reference's trajectory is (conv biases in front of a train-mode BN: not at all).  Loss and train-mode output
    of every step are held to the usual 1e-5 / 1e-4 by the callers."""
    n = 0
    for k in P:
        ref = d["P5." + k]
        if k.endswith("num_batches_tracked"):
            assert int(P[k]) == int(ref) == 5
        elif k == "A" or k.endswith(".A"):
            assert torch.equal(torch.as_tensor(P[k]).cpu(), ref)
        else:
            assert_grad_close(P[k], ref, 1e-3, k, 8.0 * float(d["noise." + k]) + 1e-7)
            n += 1
    return n


def test_trajectory_oracle_vs_reference():
    """processor.py:316-335 over 5 iterations (Adam lr 0.1 -> 0.01 before the 4th, wd 1e-4): the oracle's loss and
    train-mode output per step and its final state against the reference run (make_golden.gen_trajectory)."""
    d = sub(load_npz("trajectory"), "")
    P = {k[3:]: v.clone() for k, v in d.items() if k.startswith("P0.")}
    keys = stgcn.trainable_keys(P)
    state = {}
    for it in range(5):
        loss, out, grads = stgcn.classifier_loss_and_grads(P, d["x"][it], d["labels"][it], "real_only")
        # BN running stats / num_batches_tracked were updated in place by the forward above
        stgcn.adam_step({k: P[k] for k in keys}, grads, state, 0.1 if it < 3 else 0.01, 1e-4, it + 1)
        assert abs(float(loss) - float(d["losses"][it])) <= 1e-5 * max(1.0, abs(float(d["losses"][it]))), it
        assert_close(out, d["outs"][it], 1e-4, "train-mode output of step %d" % it)
    assert trajectory_state_check(P, d) >= 55


def test_saliency_fixture_vs_oracle():
    """tests/golden/saliency.npz (the reference's Classifier + processor.py's backward hook on
    st_gcn_networks[0].gcn.conv, eval mode, one-hot output gradient): the oracle's gradient at the first graph conv's
    input through the graph-conv branch alone."""
    P = ckpt("ckpt_classifier_real_only_epoch395")
    d = sub(load_npz("saliency"), "")
    for tag in ("h5", "rnd"):
        x = d[tag + "_x"]
        h = stgcn.data_bn(x[..., 0], P, "data_bn", False).detach().requires_grad_(True)
        A = P["A"].to(x.dtype)
        y = stgcn.st_gcn_block(h, A * P["edge_importance.0"], P, "st_gcn_networks.0", 1, True, False, x_res=h.detach())
        for i, (_, _, stride, residual) in list(enumerate(stgcn.ARCH["real_only"]))[1:]:
            y = stgcn.st_gcn_block(y, A * P["edge_importance.%d" % i], P, "st_gcn_networks.%d" % i, stride, residual, False)
        w = P["fcn.weight"]
        out = torch.softmax(y.mean(dim=(2, 3)) @ w.reshape(w.shape[0], -1).t() + P["fcn.bias"], dim=1)
        assert_close(out, d[tag + "_out"], 1e-5, tag + " out")
        one_hot = torch.zeros_like(out)
        one_hot[torch.arange(out.shape[0]), d[tag + "_labels"].long()] = 1
        out.backward(gradient=one_hot)
        assert_close(h.grad, d[tag + "_grad"], 1e-4, tag + " gradient at gcn.conv's input")
