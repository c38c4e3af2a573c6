"""GPU: the temporal convolution in the frequency domain (csrc/tcn_dft.hip, step_amd/dft.py) against the oracle's conv
(oracle.stgcn.temporal_conv = tcn[2] of classifier.py:156-162 / CVAE_stgcn.py:359-365) in float64: forward with BN + ReLU
on the load, bias and the BatchNorm statistics of the result; input gradient; weight and bias gradients.  fp32 bars."""
import pytest
import torch

from oracle import stgcn
from tests.util import assert_close

pytestmark = pytest.mark.gpu
D = "cuda:0"

CASES = [(64, 1, 33, 2, False, 75), (32, 1, 50, 3, False, 75), (64, 2, 41, 2, False, 75), (64, 2, 240, 3, False, 75),
         (64, 1, 120, 5, False, 75), (32, 1, 24, 2, True, 75), (64, 1, 75, 4, True, 75), (64, 1, 18, 3, False, 75),
         (64, 2, 120, 3, False, 9), (32, 1, 240, 2, False, 9), (64, 1, 300, 2, False, 75),
         # stride 2 with an even T runs as two stride-1 convs on the even / odd frames (dft.phases): 2 C -> C mixing
         (32, 2, 60, 3, False, 75), (64, 2, 18, 5, False, 75), (32, 2, 240, 2, False, 9), (64, 2, 150, 1, False, 75),
         # ... with several row-tile groups per phase (T = 1024: 9 tiles in 3 groups), one 5-tile group (T = 600), and the
         # smallest grids (one frame per phase; 4 spectrum rows)
         (32, 2, 1024, 1, False, 75), (64, 2, 600, 1, False, 75), (64, 2, 2, 3, False, 3), (32, 2, 4, 2, False, 9)]


@pytest.mark.parametrize("C,stride,T,N,tr,Kt", CASES)
def test_dft_conv_forward_dgrad_wgrad_vs_oracle(C, stride, T, N, tr, Kt):
    from step_amd import dft
    g = torch.Generator().manual_seed(C + T + N + Kt)
    To = (T - 1) // stride + 1
    y = torch.randn(N, C, T, 16, generator=g)
    w = torch.randn(C, C, Kt, 1, generator=g) * 0.05
    b = torch.randn(C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    du = torch.randn(N, C, To, 16, generator=g)
    # oracle, float64
    a64 = torch.relu(y.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    u64 = stgcn.temporal_conv(a64, w64, b64, stride, tr)
    da64, dw64, db64 = torch.autograd.grad(u64, (a64, w64, b64), du.double())

    yd, wd, bd, scd, shd, dud = (t.to(D) for t in (y, w, b, sc, sh, du))
    wf, wb = dft.weight_spectrum(wd, C, T, Kt, tr, To, stride)
    sums = torch.zeros(2 * C + 1, dtype=torch.float64, device=D)
    u, sa = dft.conv_forward(yd, scd, shd, True, wf, bd, To, stride, Kt, sums, transposed=tr)
    assert dft.phases(C, T, Kt, stride, tr) == (2 if stride == 2 and T % 2 == 0 else 1)
    assert sa.shape[2] == C * dft.phases(C, T, Kt, stride, tr)
    assert_close(u, u64, 1e-4, "u")
    ud = u.double()
    got = sums[:2 * C].view(C, 2)
    assert float((got[:, 0] - ud.sum(dim=(0, 2, 3))).abs().max()) <= 2e-6 * float(ud.abs().sum(dim=(0, 2, 3)).max())
    assert float(((got[:, 1] - (ud * ud).sum(dim=(0, 2, 3))) / (ud * ud).sum(dim=(0, 2, 3))).abs().max()) <= 2e-6
    sdu = dft.transform_dout(dud, stride, T, Kt, tr)
    da = dft.conv_dgrad(sdu, wb, T, Kt, stride, tr)
    assert_close(da, da64, 1e-4, "da")
    dw, db = dft.conv_wgrad(sdu, sa, T, Kt, tr, tuple(w.shape), To, stride)
    assert_close(dw, dw64, 1e-4, "dw")
    assert_close(db, db64, 1e-4, "db")
    # no statistics asked, no activation: the plain paths
    u2, _ = dft.conv_forward(yd, None, None, False, wf, None, To, stride, Kt, transposed=tr)
    assert_close(u2, stgcn.temporal_conv(y.double(), w.double(), None, stride, tr), 1e-4, "u, plain")


@pytest.mark.parametrize("C,T,N,tr,Kt,stride", [(64, 1024, 2, False, 75, 1), (32, 700, 3, False, 75, 1), (64, 530, 2, True, 75, 1),
                                                 (32, 300, 2, False, 9, 1), (64, 1024, 2, False, 75, 2), (64, 701, 2, False, 75, 2)])
def test_dft_segmented_eval_forward_vs_oracle(C, T, N, tr, Kt, stride):
    """overlap-save form of the forward (long sequences: step_dft_fwd_seg / _inv_seg), strides 1 and 2, with the last
    segment partial, ConvTranspose taps and the eval-mode tail (BN2 affine, residual with its affine, ReLU)"""
    from step_amd import dft
    g = torch.Generator().manual_seed(C + T + N + Kt + stride)
    To = (T - 1) // stride + 1
    y = torch.randn(N, C, T, 16, generator=g)
    w = torch.randn(C, C, Kt, 1, generator=g) * 0.05
    b = torch.randn(C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    res = torch.randn(N, C, To, 16, generator=g)
    osc, osh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rsc, rsh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    v = lambda t: t.double().view(1, -1, 1, 1)
    a64 = torch.relu(y.double() * v(sc) + v(sh))
    u64 = stgcn.temporal_conv(a64, w.double(), b.double(), stride, tr)
    seg = 256 if Kt == 75 else 128
    S = (T + seg - 1) // seg
    K = seg + Kt - 1
    plan = (S, seg, K, K - (Kt - 1) // 2, stride)
    if T == 1024:
        assert dft.seg_plan(C, T, To, stride, Kt) == plan          # what the model takes at configs[4]'s length
    yd, wd, bd, scd, shd = (t.to(D) for t in (y, w, b, sc, sh))
    wf = dft.weight_spectrum_seg(wd, C, Kt, tr, plan)
    u = dft.conv_forward_seg(yd, scd, shd, True, wf, bd, Kt, plan)
    assert_close(u, u64, 1e-4, "u, segmented")
    out64 = torch.relu(u64 * v(osc) + v(osh) + res.double() * v(rsc) + v(rsh))
    out = dft.conv_forward_seg(yd, scd, shd, True, wf, bd, Kt, plan,
                               epilogue=tuple(t.to(D) for t in (osc, osh, res, rsc, rsh)) + (True,))
    assert_close(out, out64, 1e-4, "eval tail, segmented")
