"""GPU: the fused CVAE loss against the reference's own values and gradients
(tests/golden/vae_loss.npz, cvae_vae_loss_train.npz) and against the oracle on larger inputs."""
import pytest
import torch

from oracle import losses as oracle_losses
from tests.util import load_npz, sub, ckpt, assert_close, assert_grad_close, feeds_train_bn

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


@pytest.mark.parametrize("tag", list("abcde"))
def test_vae_loss_vs_reference_fixture(tag):
    from step_amd import losses
    d = sub(load_npz("vae_loss"), tag + ".")
    x_in = d["x_in"].to(D)
    x_out = d["x_out"].to(D).requires_grad_(True)
    mean = d["mean"].to(D).requires_grad_(True)
    lsig = d["lsig"].to(D).requires_grad_(True)
    bf = losses.between_frame_loss(x_in, x_out.detach())
    loss = losses.vae_loss(x_in, x_out, mean, lsig, 1.0)
    loss.backward()
    assert abs(float(bf) - float(d["bf"])) <= 1e-5 * max(1.0, abs(float(d["bf"])))
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    assert_close(x_out.grad, d["d_x_out"], 1e-5, "d_x_out")
    assert_close(mean.grad, d["d_mean"], 1e-6, "d_mean")
    assert_close(lsig.grad, d["d_lsig"], 1e-6, "d_lsig")


@pytest.mark.parametrize("N,T,beta", [(64, 75, 1.0), (7, 240, 0.25), (5, 320, 2.0)])
def test_vae_loss_vs_oracle_larger(N, T, beta):
    from step_amd import losses
    g = torch.Generator().manual_seed(31)
    x_in = torch.rand(N, 3, T, 16, 1, generator=g)
    x_out = (x_in + 0.2 * torch.randn(N, 3, T, 16, 1, generator=g)).requires_grad_(True)
    mean = (torch.randn(N, 32, generator=g) * 0.5).requires_grad_(True)
    lsig = (torch.randn(N, 32, generator=g) * 0.5).requires_grad_(True)
    ref = oracle_losses.vae_loss(x_in.double(), x_out.double(), mean.double(), lsig.double(), beta)
    gx, gm, gs = torch.autograd.grad(ref, (x_out, mean, lsig))
    xo = x_out.detach().to(D).requires_grad_(True)
    m = mean.detach().to(D).requires_grad_(True)
    s = lsig.detach().to(D).requires_grad_(True)
    loss = losses.vae_loss(x_in.to(D), xo, m, s, beta)
    (3.0 * loss).backward()                       # a non-unit upstream gradient
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert_close(xo.grad, 3.0 * gx, 1e-5, "d_x_out")
    assert_close(m.grad, 3.0 * gm, 1e-5, "d_mean")
    assert_close(s.grad, 3.0 * gs, 1e-5, "d_lsig")


def test_vae_loss_rejects_what_it_cannot_hold():
    from step_amd import losses, _lib
    x = torch.zeros(1, 3, 400, 16, 1, device=D)
    with pytest.raises(_lib.StepHipError):
        losses.between_frame_loss(x, x)
    with pytest.raises(_lib.StepHipError):
        losses.between_frame_loss(torch.zeros(1, 3, 8, 16, 2, device=D), torch.zeros(1, 3, 8, 16, 2, device=D))


def test_cvae_train_step_with_reference_loss():
    """generator_cvae/utils/processor_stgcn.py:140-152 with ckpt105: encoder -> reparametrisation (the
    fixture's eps) -> decoder -> vae_loss -> backward; every parameter gradient against the reference."""
    from step_amd import losses
    from step_amd.net.CVAE_stgcn import CVAE
    m = CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    m.load_state_dict(ckpt("ckpt_cvae_epoch105"))
    m = m.to(D).train()
    d = sub(load_npz("cvae_vae_loss_train"), "", device=D)
    T = d["x"].shape[2]
    mean, lsig = m.encoder(d["x"], d["lenc"])
    zz = d["eps"] * torch.exp(0.5 * lsig) + mean
    rec = m.decoder(zz, d["onehot"], T, 16)
    assert_close(rec, d["rec"], 1e-4, "rec")
    loss = losses.vae_loss(d["x"], rec, mean, lsig)
    assert abs(float(loss) - float(d["loss"])) <= 1e-4 * abs(float(d["loss"]))
    loss.backward()
    n = 0
    for k, p in m.named_parameters():
        if "G." + k in d:
            assert_grad_close(p.grad, d["G." + k], 1e-3, k, 2e-4 if feeds_train_bn(k) else 2e-5)
            n += 1
        elif "Gsl." + k in d:
            assert_close(p.grad[:2], d["Gsl." + k], 1e-3, k)
            n += 1
    assert n > 40
