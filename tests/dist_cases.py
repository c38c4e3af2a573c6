"""Model / batch / loss builders shared by the 2-rank workers of tests/test_gpu_dist.py and their single-process
references (not a test module)."""
import torch
import torch.nn.functional as F

from tests.util import ckpt, load_npz, sub

G = {"strategy": "spatial"}
NB = 12                     # global batch of every case (2 ranks x 6)


def build(kind):
    """the model on the CPU, deterministic parameters"""
    import step_amd
    torch.manual_seed(7)
    if kind == "real_only":
        from step_amd.net.classifier import Classifier
        m = Classifier(3, 4, G)
        m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
    elif kind == "ras_bf16":
        from step_amd.net.classifier_real_and_synth import Classifier
        P = {k: v for k, v in ckpt("ckpt_classifier_real_only_epoch395").items()
             if not k.startswith("st_gcn_networks.0.residual")}
        d = sub(load_npz("real_and_synth_eval_T240"), "")
        P.update({k[2:]: v for k, v in d.items() if k.startswith("P.")})
        m = Classifier(3, 4, G)
        m.load_state_dict(P)
        step_amd.set_mfma_dtype(m, "bf16")
    elif kind == "hybrid":
        from step_amd.net.classifier_hybrid import Classifier
        m = Classifier(3, 27, 4, G)
        sd = m.state_dict()
        for k, v in ckpt("ckpt_classifier_real_only_epoch395").items():
            k2 = k.replace("data_bn.", "data_bn1.")
            if k2 in sd and not k.startswith("fcn."):
                sd[k2] = v.clone()
        m.load_state_dict(sd)
    elif kind == "cvae":
        from step_amd.net.CVAE_stgcn import CVAE
        m = CVAE(3, 75, 16, 32, 4, G)
        m.load_state_dict(ckpt("ckpt_cvae_epoch105"))
    else:
        raise ValueError(kind)
    return m


def batch(kind):
    g = torch.Generator().manual_seed(5)
    T = 75
    X = torch.randn(NB, 3, T, 16, 1, generator=g) * 0.5
    Y = torch.randint(0, 4, (NB,), generator=g)
    extra = {}
    if kind == "hybrid":
        extra["aff"] = torch.rand(NB, 27, generator=g) * 2 - 1
    if kind == "cvae":
        extra["eps"] = torch.randn(NB, 32, generator=g)
    return X, Y, extra


def loss_of(kind, m, X, Y, extra, a, b, dev):
    """training loss of rows [a, b) and the model output the tests compare"""
    x, y = X[a:b].to(dev), Y[a:b].to(dev)
    if kind == "hybrid":
        out = m(extra["aff"][a:b].to(dev), x)
        return F.cross_entropy(out, y), out
    if kind == "cvae":
        onehot = F.one_hot(y, 4).float()
        lenc = onehot[:, :, None, None, None].repeat(1, 1, x.shape[2], 16, 1)
        mean, lsig = m.encoder(x, lenc)
        z = extra["eps"][a:b].to(dev) * torch.exp(0.5 * lsig) + mean
        rec = m.decoder(z, onehot, x.shape[2], 16)
        loss = ((rec - x) ** 2).mean() - 0.5 * torch.mean(1 + lsig - mean.pow(2) - lsig.exp())
        return loss, rec
    out, _ = m(x)
    return F.cross_entropy(out, y), out
