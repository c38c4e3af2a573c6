#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own ``net`` packages (pure torch + numpy), runs them
on the CPU with fixed seeds and stores inputs, parameters and outputs as
``.npz`` (data only: no reference source is written anywhere).  The shipped
checkpoints are read with ``torch.load(weights_only=True)`` and the HDF5
fixture with the image's ``h5dump`` tool (h5py is not installed).
"""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)
# oneDNN's convolution backward never returns when the 75-tap kernel is longer than the
# sequence (T < Kt); the fixtures include such shapes, so use the native kernels here.
torch.backends.mkldnn.enabled = False


def ref_import(app, modname):
    """Import ``net.<modname>`` from one app directory (each app carries its own
    ``net``/``utils`` packages, so purge them between imports)."""
    for k in list(sys.modules):
        if k == "net" or k.startswith("net.") or k == "utils" or k.startswith("utils."):
            del sys.modules[k]
    sys.path[:] = [p for p in sys.path if not p.startswith(REF)]
    sys.path.insert(0, os.path.join(REF, app))
    return importlib.import_module("net." + modname)


def npd(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-34s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def flat(prefix, d):
    return {prefix + k: v for k, v in d.items()}


def h5_read(path, dataset):
    txt = subprocess.run(["/opt/conda/bin/h5dump", "-d", dataset, "-y", "-w", "0", path],
                         check=True, capture_output=True, text=True).stdout
    body = txt[txt.index("DATA {") + 6:]
    body = body[:body.index("}")]
    vals = [float(t) for t in re.split(r"[,\s]+", body.strip()) if t]
    return np.asarray(vals, dtype=np.float64)


def h5_names(path):
    txt = subprocess.run(["/opt/conda/bin/h5dump", "-n", path], check=True,
                         capture_output=True, text=True).stdout
    return [ln.split()[1] for ln in txt.splitlines() if ln.strip().startswith("dataset")]


# --------------------------------------------------------------------------- 1
def gen_graph():
    g = ref_import("classifier_stgcn_real_only", "utils.graph")
    out = {}
    for strat in ("uniform", "distance", "spatial"):
        out[strat] = g.Graph(strategy=strat).A
    out["spatial_hop2"] = g.Graph(strategy="spatial", max_hop=2).A
    out["distance_hop2"] = g.Graph(strategy="distance", max_hop=2).A
    save("graph", **out)


# --------------------------------------------------------------------------- 2
def gen_gcn():
    tg = ref_import("classifier_stgcn_real_only", "utils.tgcn")
    gr = ref_import("classifier_stgcn_real_only", "utils.graph")
    A0 = torch.tensor(gr.Graph(strategy="spatial").A)
    out = {}
    cases = [("f64_small", torch.float64, 2, 5, 7, 11),
             ("f32_b0", torch.float32, 2, 3, 32, 30),
             ("f32_b1", torch.float32, 2, 32, 64, 30)]
    for transposed in (False, True):
        cls = tg.ConvTransposeTemporalGraphical if transposed else tg.ConvTemporalGraphical
        for tag, dt, n, ci, co, t in cases:
            torch.manual_seed(0)
            m = cls(ci, co, 3).to(dt)
            x = torch.randn(n, ci, t, 16, dtype=dt, requires_grad=True)
            imp = (torch.rand(3, 16, 16, dtype=dt) * 2 - 0.5)
            A = (A0.to(dt) * imp).requires_grad_(True)
            y, _ = m(x, A)
            gy = torch.randn_like(y)
            y.backward(gy)
            p = ("T_" if transposed else "") + tag + "."
            out.update({p + "x": x.detach().numpy(), p + "A": A.detach().numpy(),
                        p + "w": m.conv.weight.detach().numpy(),
                        p + "b": m.conv.bias.detach().numpy(),
                        p + "y": y.detach().numpy(), p + "gy": gy.numpy(),
                        p + "gx": x.grad.numpy(), p + "gA": A.grad.numpy(),
                        p + "gw": m.conv.weight.grad.numpy(),
                        p + "gb": m.conv.bias.grad.numpy()})
    save("gcn", **out)


# ------------------------------------------------------------------------ 3, 4
BIG = 5000      # tensors above this many elements are stored as slices + norms


def _slim_grads(prefix, grads, out):
    for k, g in grads.items():
        if g.size > BIG:
            out[prefix + "Gsl." + k] = g[:2].copy()
            out[prefix + "Gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            out[prefix + "Gsum." + k] = np.float64(g.astype(np.float64).sum())
        else:
            out[prefix + "G." + k] = g


def _block_case(block_cls, cin, cout, kt, stride, residual, n, t, out, tag, A0):
    for train in (True, False):
        torch.manual_seed(1)                       # same parameters in both modes
        blk = block_cls(cin, cout, (kt, 3), stride, residual=residual)
        with torch.no_grad():                      # non-trivial BN state
            for mod in blk.modules():
                if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.normal_(0, 0.3)
                    mod.running_mean.normal_(0, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
        blk.train(train)
        if train:
            out.update(flat(tag + ".P0.", npd(blk.state_dict())))
        torch.manual_seed(2)
        x = torch.randn(n, cin, t, 16, requires_grad=True)
        imp = torch.rand(3, 16, 16) * 2 - 0.5
        A = (A0 * imp).requires_grad_(True)
        p = "%s.%s." % (tag, "train" if train else "eval")
        y, _ = blk(x, A)
        gy = torch.randn_like(y)
        y.backward(gy)
        out.update({p + "x": x.detach().numpy(), p + "A": A.detach().numpy(),
                    p + "y": y.detach().numpy(), p + "gy": gy.numpy(),
                    p + "gx": x.grad.numpy(), p + "gA": A.grad.numpy()})
        out.update(flat(p + "P1.", {k: v for k, v in npd(blk.state_dict()).items()
                                    if "running" in k}))        # running stats after
        _slim_grads(p, {k: v.grad.numpy() for k, v in blk.named_parameters()}, out)


def gen_blocks():
    cl = ref_import("classifier_stgcn_real_only", "classifier")
    gr = ref_import("classifier_stgcn_real_only", "utils.graph")
    A0 = torch.tensor(gr.Graph(strategy="spatial").A, dtype=torch.float32)
    out = {}
    for kt in (9, 75):
        t = 10 if kt == 75 else 12
        _block_case(cl.st_gcn, 3, 32, kt, 1, True, 3, t, out, "b_3_32_s1_res_k%d" % kt, A0)
        _block_case(cl.st_gcn, 32, 64, kt, 2, True, 3, t + 1, out, "b_32_64_s2_res_k%d" % kt, A0)
        _block_case(cl.st_gcn, 64, 64, kt, 1, True, 2, t, out, "b_64_64_s1_id_k%d" % kt, A0)
        _block_case(cl.st_gcn, 3, 32, kt, 1, False, 3, t, out, "b_3_32_s1_nores_k%d" % kt, A0)
        if kt == 9:
            _block_case(cl.st_gcn, 64, 64, kt, 2, True, 2, t, out, "b_64_64_s2_res_k%d" % kt, A0)
    save("st_gcn_blocks", **out)

    cv = ref_import("generator_cvae", "CVAE_stgcn")
    out = {}
    for kt in (9, 75):
        t = 10 if kt == 75 else 12
        _block_case(cv.st_gctn, 32, 32, kt, 1, True, 2, t, out, "t_32_32_k%d" % kt, A0)
        if kt == 9:
            _block_case(cv.st_gctn, 32, 64, kt, 1, True, 2, t, out, "t_32_64_k%d" % kt, A0)
        _block_case(cv.st_gctn, 64, 3, kt, 1, True, 2, t, out, "t_64_3_k%d" % kt, A0)
        if kt == 9:
            _block_case(cv.st_gcn, 7, 64, kt, 1, True, 2, t, out, "e_7_64_k%d" % kt, A0)
        _block_case(cv.st_gcn, 64, 32, kt, 1, True, 2, t, out, "e_64_32_k%d" % kt, A0)
    save("st_gctn_blocks", **out)


# --------------------------------------------------------------------------- 5
def load_h5_gaits():
    fpath = os.path.join(REF, "classifier_stgcn_real_only", "features4DCVAEGCN.h5")
    lpath = os.path.join(REF, "classifier_stgcn_real_only", "labels4DCVAEGCN.h5")
    names = h5_names(fpath)
    feats = np.stack([h5_read(fpath, n).reshape(-1, 48) for n in names])
    labels = np.asarray([int(h5_read(lpath, n)[0]) for n in names], dtype=np.int64)
    return names, feats, labels


def gen_ckpt_and_eval():
    cl = ref_import("classifier_stgcn_real_only", "classifier")
    ck = os.path.join(REF, "classifier_stgcn_real_only", "model_classifier_stgcn", "features",
                      "epoch395_acc81.25_model.pth.tar")
    sd = torch.load(ck, map_location="cpu", weights_only=True)
    model = cl.Classifier(3, 4, {"strategy": "spatial"})
    model.load_state_dict(sd)
    model.eval()
    save("ckpt_classifier_real_only_epoch395", **npd(sd))

    names, feats, labels = load_h5_gaits()                     # (60, 18, 48) f64
    # TrainTestLoader reshape/moveaxis, loader.py:65-66 (Procrustes step skipped)
    data = feats.reshape(feats.shape[0], feats.shape[1], 16, 3, 1)
    data = np.moveaxis(data, [1, 2, 3], [2, 3, 1]).astype(np.float32)
    torch.manual_seed(0)
    xr = torch.randn(6, 3, 240, 16, 1) * 0.5
    with torch.no_grad():
        p1, f1 = model(torch.from_numpy(data))
        p2, f2 = model(xr)
        o3, e3 = model.extract_feature(xr[:2])
    save("classifier_real_only_eval", h5_x=data, h5_labels=labels, h5_probs=p1.numpy(),
         h5_feat=f1.numpy(), rnd_x=xr.numpy(), rnd_probs=p2.numpy(), rnd_feat=f2.numpy(),
         ef_out=o3.numpy(), ef_feat=e3.numpy())


# --------------------------------------------------------------------------- 6
def _train_step(model, args, labels, lr=0.1, wd=1e-4):
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    out = model(*args)
    out = out[0] if isinstance(out, tuple) else out
    loss = torch.nn.CrossEntropyLoss()(out, labels)
    opt.zero_grad()
    loss.backward()
    grads = {k: v.grad.detach().clone().numpy() for k, v in model.named_parameters()}
    opt.step()
    return loss.item(), out.detach().numpy(), grads


def _init_like_processor(model):
    """The harness initialises every Conv1d/Conv2d weight ~ N(0, 0.02) with zero bias
    and every BatchNorm weight ~ N(1, 0.02) with zero bias (processor.py:117-129;
    ConvTranspose2d keeps its default init)."""
    nn = torch.nn
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                torch.nn.init.normal_(m.weight, 0.0, 0.02)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                torch.nn.init.normal_(m.weight, 1.0, 0.02)
                m.bias.zero_()


def gen_train_steps():
    out = {}
    for variant, app in (("real_only", "classifier_stgcn_real_only"),
                         ("real_and_synth", "classifier_stgcn_real_and_synth"),
                         ("hybrid", "classifier_hybrid")):
        cl = ref_import(app, "classifier")
        kt, n, t = 9, 8, 75
        torch.manual_seed(3)
        if variant == "hybrid":
            model = cl.Classifier(3, 29, 4, {"strategy": "spatial"}, temporal_kernel_size=kt)
        else:
            model = cl.Classifier(3, 4, {"strategy": "spatial"}, temporal_kernel_size=kt)
        _init_like_processor(model)
        with torch.no_grad():          # importance away from 1 so its gradient path is live
            for p in model.edge_importance:
                p.add_(torch.randn_like(p) * 0.2)
        x = torch.randn(n, 3, t, 16, 1) * 0.5
        labels = torch.randint(0, 4, (n,))
        p = "%s_k%d." % (variant, kt)
        P0 = npd(model.state_dict())
        args = (torch.rand(n, 29) * 2 - 1, x) if variant == "hybrid" else (x,)
        loss, o, grads = _train_step(model, args, labels)
        P1 = npd(model.state_dict())
        out[p + "x"] = x.numpy()
        if variant == "hybrid":
            out[p + "x_aff"] = args[0].numpy()
        out[p + "labels"] = labels.numpy()
        out[p + "loss"] = np.float64(loss)
        out[p + "out"] = o
        out.update(flat(p + "P0.", P0))
        out.update(flat(p + "G.", grads))
        out.update(flat(p + "P1.", P1))

    # Kt = 75, the shipped architecture and checkpoint: parameters come from
    # ckpt_classifier_real_only_epoch395.npz, so only data, loss and gradients are stored.
    cl = ref_import("classifier_stgcn_real_only", "classifier")
    ck = os.path.join(REF, "classifier_stgcn_real_only", "model_classifier_stgcn", "features",
                      "epoch395_acc81.25_model.pth.tar")
    model = cl.Classifier(3, 4, {"strategy": "spatial"})
    model.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True))
    torch.manual_seed(4)
    x = torch.randn(4, 3, 75, 16, 1) * 0.5
    labels = torch.randint(0, 4, (4,))
    loss, o, grads = _train_step(model, (x,), labels)
    p = "real_only_k75_ckpt395."
    out[p + "x"] = x.numpy()
    out[p + "labels"] = labels.numpy()
    out[p + "loss"] = np.float64(loss)
    out[p + "out"] = o
    _slim_grads(p, grads, out)
    out.update(flat(p + "P1.", {k: v for k, v in npd(model.state_dict()).items()
                                if v.size <= BIG}))
    save("train_steps", **out)


# --------------------------------------------------------------------------- 7
def gen_cvae():
    cv = ref_import("generator_cvae", "CVAE_stgcn")
    ck = os.path.join(REF, "generator_cvae", "model_gait_cvae_stgcn", "features",
                      "epoch105_model.pth.tar")
    sd = torch.load(ck, map_location="cpu", weights_only=True)
    model = cv.CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    print("cvae load:", model.load_state_dict(sd))
    model.eval()
    save("ckpt_cvae_epoch105", **npd(sd))
    torch.manual_seed(5)
    n = 4
    x = torch.rand(n, 3, 75, 16, 1)
    cls = torch.tensor([0, 1, 2, 3])
    onehot = torch.eye(4)[cls]
    lenc = onehot[:, :, None, None, None].repeat(1, 1, 75, 16, 1)
    z = torch.randn(n, 32) * 3 - 1.5
    with torch.no_grad():
        mean, lsig = model.encoder(x, lenc)
        dec75 = model.decoder(z, onehot, 75, 16)
        dec240 = model.decoder(z[:2], onehot[:2], 240, 16)
    save("cvae_eval", x=x.numpy(), lenc=lenc.numpy(), onehot=onehot.numpy(), z=z.numpy(),
         mean=mean.numpy(), lsig=lsig.numpy(), dec75=dec75.numpy(), dec240=dec240.numpy())

    # one training-mode forward/backward of the whole CVAE, parameters = ckpt105
    # (CVAE.__init__ does not forward temporal_kernel_size, so Kt is always 75).
    model.train()
    torch.manual_seed(6)
    T = 24
    x = torch.rand(3, 3, T, 16, 1)
    cls = torch.tensor([1, 3, 0])
    onehot = torch.eye(4)[cls]
    lenc = onehot[:, :, None, None, None].repeat(1, 1, T, 16, 1)
    mean, lsig = model.encoder(x, lenc)
    eps = torch.randn(3, 32)
    zz = eps * torch.exp(0.5 * lsig) + mean
    rec = model.decoder(zz, onehot, T, 16)
    loss = ((rec - x) ** 2).mean() - 0.5 * torch.mean(1 + lsig - mean.pow(2) - lsig.exp())
    loss.backward()
    out = dict(x=x.numpy(), lenc=lenc.numpy(), onehot=onehot.numpy(), eps=eps.numpy(),
               mean=mean.detach().numpy(), lsig=lsig.detach().numpy(),
               rec=rec.detach().numpy(), loss=np.float64(loss.item()))
    _slim_grads("", {k: v.grad.numpy() for k, v in model.named_parameters()}, out)
    out.update(flat("P1.", {k: v for k, v in npd(model.state_dict()).items() if "running" in k}))
    save("cvae_train", **out)


# --------------------------------------------------------------------------- 8
def gen_hybrid():
    """Kt = 75 eval of the two other classifier variants.  Their best checkpoints are
    missing from the reference (.MISSING_LARGE_BLOBS), so the ST-GCN stack weights are
    borrowed from ckpt395 (same shapes; block 0 simply has no residual branch) and only
    the small head tensors are stored here."""
    ck = os.path.join(REF, "classifier_stgcn_real_only", "model_classifier_stgcn", "features",
                      "epoch395_acc81.25_model.pth.tar")
    sd = torch.load(ck, map_location="cpu", weights_only=True)

    cl = ref_import("classifier_hybrid", "classifier")
    torch.manual_seed(7)
    model = cl.Classifier(3, 29, 4, {"strategy": "spatial"})
    _init_like_processor(model)
    own = model.state_dict()
    borrowed = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
    borrowed.update({k.replace("data_bn.", "data_bn1."): v for k, v in sd.items()
                     if k.startswith("data_bn.")})
    print("hybrid load:", model.load_state_dict(borrowed, strict=False))
    model.eval()
    with torch.no_grad():
        model.data_bn2.running_mean.normal_(0, 0.1)
        model.data_bn2.running_var.uniform_(0.6, 1.4)
    xa = torch.rand(3, 29) * 2 - 1
    xg = torch.randn(3, 3, 1024, 16, 1) * 0.5
    with torch.no_grad():
        o = model(xa, xg)
    out = dict(x_aff=xa.numpy(), x_gait=xg.numpy(), out=o.numpy())
    out.update(flat("P.", {k: v for k, v in npd(model.state_dict()).items()
                           if k.startswith(("data_bn2", "combined_networks",
                                            "st_gcn_networks.2.residual"))}))
    save("hybrid_eval_T1024", **out)

    cl2 = ref_import("classifier_stgcn_real_and_synth", "classifier")
    model = cl2.Classifier(3, 4, {"strategy": "spatial"})
    own = model.state_dict()
    print("real_and_synth load:", model.load_state_dict(
        {k: v for k, v in sd.items() if k in own}, strict=False))
    model.eval()
    torch.manual_seed(8)
    x = torch.randn(4, 3, 240, 16, 1) * 0.5
    with torch.no_grad():
        o, f = model(x)
    out = dict(x=x.numpy(), out=o.numpy(), feat=f.numpy())
    out.update(flat("P.", {k: v for k, v in npd(model.state_dict()).items() if k not in sd}))
    save("real_and_synth_eval_T240", **out)


# --------------------------------------------------------------------------- 9
def gen_vae_loss():
    """generator_cvae/utils/losses.py between_frame_loss and the vae_loss of
    processor_stgcn.py:34-40 (between_frame_loss + KLD), values and autograd gradients.
    losses.py calls ``np.int`` (removed in numpy 1.24); the alias it had is restored for the call."""
    cv = ref_import("generator_cvae", "CVAE_stgcn")
    losses = importlib.import_module("utils.losses")
    if not hasattr(np, "int"):
        np.int = int          # numpy < 1.24: ``np.int is int``
    out = {}
    torch.manual_seed(21)
    for tag, (n, t) in {"a": (3, 12), "b": (2, 75), "c": (2, 1), "d": (1, 2), "e": (2, 3)}.items():
        x_in = torch.rand(n, 3, t, 16, 1)
        x_out = (x_in + 0.3 * torch.randn(n, 3, t, 16, 1)).requires_grad_(True)
        mean = (torch.randn(n, 32) * 0.5).requires_grad_(True)
        lsig = (torch.randn(n, 32) * 0.5).requires_grad_(True)
        bf = losses.between_frame_loss(x_in, x_out)
        kld = -0.5 * torch.sum(1 + lsig - mean.pow(2) - lsig.exp())
        loss = bf + 1.0 * kld
        loss.backward()
        out.update({tag + ".x_in": x_in.numpy(), tag + ".x_out": x_out.detach().numpy(),
                    tag + ".mean": mean.detach().numpy(), tag + ".lsig": lsig.detach().numpy(),
                    tag + ".bf": np.float64(bf.item()), tag + ".loss": np.float64(loss.item()),
                    tag + ".d_x_out": x_out.grad.numpy().copy(), tag + ".d_mean": mean.grad.numpy().copy(),
                    tag + ".d_lsig": lsig.grad.numpy().copy()})
    save("vae_loss", **out)

    # one CVAE training forward/backward with the reference's own loss (ckpt105, T = 24)
    ck = os.path.join(REF, "generator_cvae", "model_gait_cvae_stgcn", "features", "epoch105_model.pth.tar")
    sd = torch.load(ck, map_location="cpu", weights_only=True)
    model = cv.CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    model.load_state_dict(sd)
    model.train()
    torch.manual_seed(22)
    T = 24
    x = torch.rand(3, 3, T, 16, 1)
    onehot = torch.eye(4)[torch.tensor([2, 0, 3])]
    lenc = onehot[:, :, None, None, None].repeat(1, 1, T, 16, 1)
    mean, lsig = model.encoder(x, lenc)
    eps = torch.randn(3, 32)
    zz = eps * torch.exp(0.5 * lsig) + mean
    rec = model.decoder(zz, onehot, T, 16)
    bf = losses.between_frame_loss(x, rec)
    loss = bf + (-0.5 * torch.sum(1 + lsig - mean.pow(2) - lsig.exp()))
    loss.backward()
    out = dict(x=x.numpy(), lenc=lenc.numpy(), onehot=onehot.numpy(), eps=eps.numpy(),
               rec=rec.detach().numpy(), loss=np.float64(loss.item()), bf=np.float64(bf.item()))
    _slim_grads("", {k: v.grad.numpy() for k, v in model.named_parameters()}, out)
    save("cvae_vae_loss_train", **out)


# --------------------------------------------------------------------------- 10
def gen_pose_align():
    """the loader's Procrustes alignment, utils/common.py:91-112 (first 48 of the 175
    features, the only ones utils/loader.py:37 keeps) with get_transformation (:230-279)."""
    for k in list(sys.modules):
        if k == "utils" or k.startswith("utils."):
            del sys.modules[k]
    sys.path[:] = [q for q in sys.path if not q.startswith(REF)]
    sys.path.insert(0, os.path.join(REF, "classifier_stgcn_real_only"))
    common = importlib.import_module("utils.common")
    rng = np.random.default_rng(31)
    base = rng.standard_normal((16, 3))                       # a "skeleton"
    out = {}
    for tag, (n, t) in {"a": (6, 10), "b": (3, 1)}.items():
        rot = np.linalg.qr(rng.standard_normal((n, 3, 3)))[0]
        gaits = (base[None, None] * rng.uniform(0.5, 2.0, (n, 1, 1, 1))) @ np.swapaxes(rot, 1, 2)[:, None]
        gaits = gaits + rng.standard_normal((n, 1, 1, 3)) + 0.05 * rng.standard_normal((n, t, 16, 3))
        feats = common.get_affective_features(gaits)
        out[tag + ".gaits"] = gaits
        out[tag + ".aligned"] = feats[:, :, :48].copy()
    save("pose_align", **out)


# --------------------------------------------------------------------------- 10b
def gen_assemble():
    """SURVEY 8(f)3, utils/loader.py:21-37: the pad / tile / `cycles` assembly of load_data followed by the
    alignment.  loader.py itself cannot be imported here (h5py, torchvision are absent), so the statements of
    load_data between the HDF5 opens and train_test_split are read from the reference file at generation time
    and executed as they stand, with plain dicts of numpy arrays standing in for the two h5py.File objects
    (they only need .keys(), item access, len() and [()]).  Nothing of the source is stored: the fixture holds
    the ragged inputs and the reference's outputs."""
    for k in list(sys.modules):
        if k == "utils" or k.startswith("utils."):
            del sys.modules[k]
    sys.path[:] = [q for q in sys.path if not q.startswith(REF)]
    sys.path.insert(0, os.path.join(REF, "classifier_stgcn_real_only"))
    common = importlib.import_module("utils.common")
    src = open(os.path.join(REF, "classifier_stgcn_real_only", "utils", "loader.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.strip().startswith("data_list = []"))
    pad_end = next(i for i, l in enumerate(src) if "common.get_affective_features" in l)
    import textwrap
    part1 = textwrap.dedent("\n".join(src[start:pad_end]))          # read loop + pad/tile/cycle
    part2 = textwrap.dedent(src[pad_end])                             # the alignment line
    names, feats, labels = load_h5_gaits()                           # (60, 18, 48) f64
    rng = np.random.default_rng(77)
    out = {}
    for tag, (n, cycles) in {"c3": (60, 3), "c1": (7, 1)}.items():
        lens = rng.integers(5, 19, n)
        lens[0] = 18 if tag == "c3" else 11                          # the longest gait need not be 18
        lens[1] = 1 if tag == "c1" else lens[1]                      # a one-frame gait
        ff = {names[i]: feats[i, :lens[i]].copy() for i in range(n)}
        fl = {names[i]: np.asarray(labels[i]) for i in range(n)}
        ns = dict(np=np, common=common, ff=ff, fl=fl, cycles=cycles, joints=16, coords=3)
        exec(part1, ns)
        out[tag + ".padded"] = ns["data"].copy()
        exec(part2, ns)
        out[tag + ".aligned"] = ns["data"].copy()
        out[tag + ".lens"] = lens.astype(np.int64)
        out[tag + ".frames"] = np.concatenate([ff[k] for k in ff], 0)
        out[tag + ".labels"] = ns["labels"].copy()
        out[tag + ".cycles"] = np.int64(cycles)
        assert ns["data"].shape == (n, int(lens.max()) * cycles, 48)
    save("assemble", **out)


# --------------------------------------------------------------------------- 6b
def gen_trajectory():
    """processor.py:316-335 over several iterations: 5 steps of Adam(lr=0.1, wd=1e-4) (processor.py:251-255) on
    the real_only Classifier, Kt = 75, N = 8, T = 75, with the LR dropped x0.1 before step 4 (adjust_lr,
    processor.py:260-268).  Stores the loss of every step, the final state_dict (incl. num_batches_tracked and
    the BN running statistics) and the initial parameters."""
    cl = ref_import("classifier_stgcn_real_only", "classifier")
    torch.manual_seed(21)
    model = cl.Classifier(3, 4, {"strategy": "spatial"})
    _init_like_processor(model)
    n, t = 8, 75
    xs = torch.randn(5, n, 3, t, 16, 1) * 0.5
    labels = torch.randint(0, 4, (5, n))
    P0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def run(dtype, jitter=None):
        m = cl.Classifier(3, 4, {"strategy": "spatial"})
        m.load_state_dict(P0)
        m = m.to(dtype)
        opt = torch.optim.Adam(m.parameters(), lr=0.1, weight_decay=1e-4)
        lossf = torch.nn.CrossEntropyLoss()
        losses, outs = [], []
        m.train()
        for it in range(5):
            if it == 3:
                for g in opt.param_groups:
                    g["lr"] = 0.01
            xin = xs[it] if jitter is None else xs[it] * (1.0 + 1e-7 * torch.randn(xs[it].shape, generator=jitter))
            o, _ = m(xin.to(dtype))
            loss = lossf(o, labels[it])
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
            outs.append(o.detach().float().numpy().copy())
        return m, losses, outs

    model, losses, outs = run(torch.float32)
    m64, losses64, _ = run(torch.float64)
    out = dict(x=xs.numpy(), labels=labels.numpy(), losses=np.asarray(losses, dtype=np.float64),
               losses_f64=np.asarray(losses64, dtype=np.float64), outs=np.stack(outs))
    out.update(flat("P0.", npd(P0)))
    P5, P5d = npd(model.state_dict()), npd(m64.state_dict())
    out.update(flat("P5.", P5))
    # how far the reference's OWN fp32 run is from its float64 run, per tensor: Adam divides by sqrt(v), so a
    # parameter whose true gradient is zero (conv biases in front of a train-mode BN) or at round-off level takes
    # +-lr steps along the SIGN of rounding noise - its fp32 trajectory is not reproducible by any implementation.
    # Tests allow that much on top of their own tolerance.
    # ... and from four fp32 runs whose inputs are moved by one part in 10^7 (less than an fp32 ulp of most values)
    others = [P5d]
    for seed in range(4):
        mj, _, _ = run(torch.float32, torch.Generator().manual_seed(100 + seed))
        others.append(npd(mj.state_dict()))
    for k in P5:
        if P5[k].dtype.kind == "f":
            out["noise." + k] = np.float64(max(np.abs(P5[k].astype(np.float64) - o[k]).max() for o in others))
    save("trajectory", **out)


# --------------------------------------------------------------------------- 11
def gen_affective():
    """compute_aff_features/compute_features.py:428-468 (27 posture/movement features per
    gait, 29 with the stride features) and normalize_features.py:1-22 (min-max to [-1, 1] over the set)."""
    sys.path[:] = [q for q in sys.path if not q.startswith(REF)]
    sys.path.insert(0, REF)
    cf = importlib.import_module("compute_aff_features.compute_features")
    nf = importlib.import_module("compute_aff_features.normalize_features")
    ts = 1.0 / 30.0                                              # compute_aff_features/main.py:19
    rng = np.random.default_rng(47)
    _, h5g, _ = load_h5_gaits()                                  # the 60 shipped gaits, (60, 18, 48) f64
    walk = np.cumsum(0.3 * rng.standard_normal((12, 64, 48)), axis=1) + 5.0 * rng.standard_normal((12, 1, 48))
    # foot heights on a coarse grid: plateaus and ties exercise every branch of the strike-point rule
    for j in (12, 15):
        walk[:, :, 3 * j + 1] = np.round(walk[:, :, 3 * j + 1])
    walk[3, :, 3 * 12 + 1] = 1.0                                 # right foot never moves: one strike run
    walk[4, :, 3 * 12 + 1] = np.arange(64)                       # right foot rises monotonically: < 2 strikes
    walk[4, :, 3 * 15 + 1] = -np.arange(64)                      # ... and so does the left: "no stride" branch
    walk[5, :, 3 * 15 + 1] = np.arange(64)                       # left foot only
    short = rng.standard_normal((5, 4, 48))                      # T = 4: one jerk sample
    mono = rng.standard_normal((3, 6, 48))
    mono[1, 2, 0::3] = -np.arange(16.0)                          # x strictly decreasing over the joints of a frame:
    out = {}                                                     # the elif of :21-24 never runs, max stays -inf
    for tag, g in {"h5": h5g, "walk": walk, "short": short, "mono": mono}.items():
        with np.errstate(all="ignore"):
            f27 = np.asarray([cf.compute_features(x, ts) for x in g], dtype=np.float64)
            f29 = np.asarray([cf.compute_features(x, ts, add_stride_features=True) for x in g], dtype=np.float64)
            rows = [["k%d" % i] + list(r) for i, r in enumerate(f29)]
            normed = []
            nf.normalize_features(rows, normed)
        out[tag + ".gaits"] = g
        out[tag + ".f27"] = f27
        out[tag + ".f29"] = f29
        out[tag + ".norm29"] = np.asarray([r[1:] for r in normed], dtype=np.float64)
    save("affective", **out)


def gen_saliency():
    """GuidedBackprop as the reference ships it (classifier_stgcn_real_only/utils/processor.py:17-36,98-113; its
    `update_relus` is commented out at :28, so what `--smap` computes is the plain gradient of the target-class output
    with respect to the input of the first graph conv's 1x1 Conv2d, taken by a backward hook on
    `st_gcn_networks[0].gcn.conv`, model in eval mode).  processor.py itself imports h5py / torchlight, which are
    absent here, so its hook / one-hot / backward statements are restated on the reference's own Classifier."""
    cl = ref_import("classifier_stgcn_real_only", "classifier")
    ck = os.path.join(REF, "classifier_stgcn_real_only", "model_classifier_stgcn", "features",
                      "epoch395_acc81.25_model.pth.tar")
    model = cl.Classifier(3, 4, {"strategy": "spatial"})
    model.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True))
    model.eval()                                                            # processor.py:26
    got = {}

    def hook_function(module, grad_in, grad_out):                           # processor.py:32-33
        got["g"] = grad_in[0]
    first_layer = list(model._modules.items())[1][1][0].gcn.conv            # processor.py:35
    first_layer.register_backward_hook(hook_function)                       # processor.py:36

    def generate(x, target):                                                # processor.py:98-113
        output, _ = model(x)
        model.zero_grad()
        one_hot = torch.zeros(output.size())
        for idx in range(output.shape[0]):
            one_hot[idx, target[idx]] = 1
        output.backward(gradient=one_hot)
        return output.detach().numpy().copy(), got["g"].detach().numpy().copy()

    names, feats, labels = load_h5_gaits()
    data = feats.reshape(feats.shape[0], feats.shape[1], 16, 3, 1)
    data = np.moveaxis(data, [1, 2, 3], [2, 3, 1]).astype(np.float32)[:8]
    tg = labels[:8] % 4                                              # (the file holds classes 1..4)
    out_h5, g_h5 = generate(torch.from_numpy(data), tg)
    torch.manual_seed(3)
    xr = torch.randn(4, 3, 75, 16, 1) * 0.5
    lr = np.asarray([0, 1, 2, 3], dtype=np.int64)
    out_r, g_r = generate(xr, lr)
    save("saliency", h5_x=data, h5_labels=tg, h5_out=out_h5, h5_grad=g_h5,
         rnd_x=xr.numpy(), rnd_labels=lr, rnd_out=out_r, rnd_grad=g_r)


if __name__ == "__main__":
    which = sys.argv[1:] or ["graph", "gcn", "blocks", "ckpt", "train", "cvae", "hybrid", "vae_loss", "pose", "affective", "assemble", "trajectory", "saliency"]
    fns = dict(graph=gen_graph, gcn=gen_gcn, blocks=gen_blocks, ckpt=gen_ckpt_and_eval,
               train=gen_train_steps, cvae=gen_cvae, hybrid=gen_hybrid, vae_loss=gen_vae_loss, pose=gen_pose_align,
               affective=gen_affective, assemble=gen_assemble, trajectory=gen_trajectory, saliency=gen_saliency)
    for w in which:
        fns[w]()
