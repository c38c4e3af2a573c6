"""CPU (no GPU needed): the host side of the frequency-domain temporal conv (step_amd/dft.py).  The cos / sin matrices
and the tap tables the kernels are handed are built on the host; here they are taken out of their tiled layout again
and the same algebra (csrc/tcn_dft.hip: half-size transform about the centre of the frames -> per-bin complex product
with the phase-shifted weight spectrum -> half-size inverse) is run in numpy against the oracle's temporal conv
(oracle/stgcn.py, classifier_stgcn_real_only/net/classifier.py:156-162) and its gradients.  That pins the tables, the
origin shift of strided outputs and the layout contract of step_dft_table_geom without a kernel in the loop."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import stgcn
from step_amd import _lib, dft


def untile(t, rows, inverse, parts):
    """[groups][kpad][slots * 32] -> `parts` matrices [kpad][rows] (the inverse of dft._tile)"""
    wp, groups = ctypes.c_int(0), ctypes.c_int(0)
    mg = _lib.load().step_dft_table_geom(rows, int(inverse), ctypes.byref(wp), ctypes.byref(groups))
    wp, groups = wp.value, groups.value
    t = t.double().numpy()
    assert t.shape[0] == groups and t.shape[2] == wp * 32
    kpad = t.shape[1]
    nq = wp // 4
    slots = np.zeros((groups, kpad, wp, 32))
    slots[:, :, :4 * nq] = t[:, :, :nq * 128].reshape(groups, kpad, nq, 32, 4).transpose(0, 1, 2, 4, 3).reshape(groups, kpad, 4 * nq, 32)
    if wp % 4:
        slots[:, :, 4 * nq:] = t[:, :, nq * 128:].reshape(groups, kpad, 32, wp - 4 * nq).transpose(0, 1, 3, 2)
    out = []
    for h in range(parts):
        m = slots[:, :, h * mg:(h + 1) * mg].transpose(1, 0, 2, 3).reshape(kpad, groups * mg * 32)
        assert not m[:, rows:].any()
        out.append(m[:, :rows])
    assert not slots[:, :, parts * mg:].any()               # padding slots are zero
    return out


def transform(x, step, T, Kt):
    """(re, im) [NF] of x[K] at frames step * k, about their centre"""
    K = len(x)
    rows, NF, L = dft.geometry(T, Kt)
    cosm, sinm = untile(dft.fwd_matrix(K, step, T, Kt, "cpu"), NF, False, 2)
    Kh = (K + 1) // 2
    assert not cosm[Kh:].any() and not sinm[Kh:].any()       # k padding
    xp = np.concatenate([x, np.zeros(2 * Kh - K)])
    j = np.arange(Kh)
    return (xp[Kh + j] + xp[Kh - 1 - j]) @ cosm[:Kh], (xp[Kh + j] - xp[Kh - 1 - j]) @ sinm[:Kh]


def inverse(re, im, T_out, step, T, Kt):
    rows, NF, L = dft.geometry(T, Kt)
    Th = (T_out + 1) // 2
    (m,) = untile(dft.inv_matrix(T_out, step, T, Kt, "cpu"), Th, True, 1)
    NFp = (NF + 15) // 16 * 16                                # each half padded with zero rows to whole chunks of 16 k
    assert m.shape[0] == 2 * NFp and not m[NF:NFp].any() and not m[NFp + NF:].any()
    E, O = re @ m[:NF], im @ m[NFp:NFp + NF]
    out = np.zeros(2 * Th)
    j = np.arange(Th)
    out[Th + j], out[Th - 1 - j] = E + O, E - O
    return out[:T_out]


@pytest.mark.parametrize("T,stride,Kt", [(240, 2, 75), (240, 1, 75), (120, 1, 75), (31, 2, 75), (33, 2, 9), (18, 1, 9), (75, 2, 75)])
def test_tables_reproduce_the_temporal_conv_and_its_gradients(T, stride, Kt):
    rng = np.random.default_rng(T + stride + Kt)
    To = (T - 1) // stride + 1
    a, w, du = rng.standard_normal(T), rng.standard_normal(Kt) * 0.1, rng.standard_normal(To)
    at = torch.tensor(a).view(1, 1, T, 1).requires_grad_(True)
    wt = torch.tensor(w).view(1, 1, Kt, 1).requires_grad_(True)
    u = stgcn.temporal_conv(at, wt, None, stride, False)
    da_ref, dw_ref = torch.autograd.grad(u, (at, wt), torch.tensor(du).view(1, 1, To, 1))
    rows, NF, L = dft.geometry(T, Kt)
    assert rows == 2 * NF and L == rows - 1 and L >= T + (Kt - 1) // 2
    tc, ts = (t.double().numpy() for t in dft.tap_tables(T, Kt, To, stride, "cpu"))
    W = tc @ w + 1j * (ts @ w)                                   # w^ with the input -> output origin shift as a phase
    are, aim = transform(a, 1, T, Kt)
    R = W * (are + 1j * aim)
    tol = 2e-5 * max(1.0, np.abs(u.detach().numpy()).max())      # the tables are float32
    assert np.abs(inverse(R.real, R.imag, To, stride, T, Kt) - u.detach().numpy().ravel()).max() < tol
    dre, dim_ = transform(du, stride, T, Kt)
    dA = np.conj(W) * (dre + 1j * dim_)
    assert np.abs(inverse(dA.real, dA.imag, T, 1, T, Kt) - da_ref.numpy().ravel()).max() < 2e-5 * max(1.0, np.abs(da_ref.numpy()).max())
    C = np.conj(dre + 1j * dim_) * (are + 1j * aim)
    kap = np.full(NF, 2.0 / L)
    kap[0] = 1.0 / L
    dw = (kap * C.real) @ tc - (kap * C.imag) @ ts
    assert np.abs(dw - dw_ref.numpy().ravel()).max() < 2e-5 * max(1.0, np.abs(dw_ref.numpy()).max())


@pytest.mark.parametrize("T,Kt", [(240, 75), (18, 75), (60, 9), (150, 75), (2, 3)])
def test_phase_tables_reproduce_the_stride_2_conv_and_its_gradients(T, Kt):
    """stride 2 as two stride-1 convs on the even / odd frames (dft.phases): the tap split (_phase_taps), the phase tap
    tables and the phase grid's geometry, in numpy against the oracle's strided conv and both gradients"""
    assert dft.phases(64, T, Kt, 2, False) == 2 and dft.phases(64, T + 1, Kt, 2, False) == 1
    assert dft.phases(64, T, Kt, 2, True) == 1 and dft.phases(64, T, Kt, 1, False) == 1
    rng = np.random.default_rng(T + Kt)
    To = T // 2
    a, w, du = rng.standard_normal(T), rng.standard_normal(Kt) * 0.1, rng.standard_normal(To)
    at = torch.tensor(a).view(1, 1, T, 1).requires_grad_(True)
    wt = torch.tensor(w).view(1, 1, Kt, 1).requires_grad_(True)
    u = stgcn.temporal_conv(at, wt, None, 2, False)
    assert u.shape[2] == To
    da_ref, dw_ref = torch.autograd.grad(u, (at, wt), torch.tensor(du).view(1, 1, To, 1))
    Tp, Ktp = dft.phase_geometry(T, Kt)
    P = (Kt - 1) // 2
    assert Tp == To and Ktp == 2 * ((P + 1) // 2) + 1
    rows, NF, L = dft.geometry(Tp, Ktp)
    assert L >= Tp + (P + 1) // 2
    masks, m = dft._phase_taps(Kt, "cpu")
    masks = masks.double().numpy()
    d = np.arange(Kt) - P
    assert ((2 * m.numpy() + masks[1]) == d).all() and (masks.sum(0) == 1).all()      # frame 2 t' + d = 2 (t' + m) + phase
    tc, ts = (t.double().numpy() for t in dft.phase_tap_tables(T, Kt, "cpu"))
    W = [tc @ (w * masks[ph]) + 1j * (ts @ (w * masks[ph])) for ph in range(2)]
    A = []
    for ph in range(2):
        re, im = transform(a[ph::2], 1, Tp, Ktp)
        A.append(re + 1j * im)
    R = W[0] * A[0] + W[1] * A[1]
    tol = 2e-5 * max(1.0, np.abs(u.detach().numpy()).max())
    assert np.abs(inverse(R.real, R.imag, To, 1, Tp, Ktp) - u.detach().numpy().ravel()).max() < tol
    dre, dim_ = transform(du, 1, Tp, Ktp)
    D = dre + 1j * dim_
    da = np.zeros(T)
    for ph in range(2):
        dA = np.conj(W[ph]) * D
        da[ph::2] = inverse(dA.real, dA.imag, Tp, 1, Tp, Ktp)
    assert np.abs(da - da_ref.numpy().ravel()).max() < 2e-5 * max(1.0, np.abs(da_ref.numpy()).max())
    kap = np.full(NF, 2.0 / L)
    kap[0] = 1.0 / L
    dw = np.zeros(Kt)
    for ph in range(2):
        C = np.conj(D) * A[ph]
        dw += masks[ph] * ((kap * C.real) @ tc - (kap * C.imag) @ ts)
    assert np.abs(dw - dw_ref.numpy().ravel()).max() < 2e-5 * max(1.0, np.abs(dw_ref.numpy()).max())


def test_cost_model_and_geometry():
    # 75 taps: the frequency-domain path at every sequence length BASELINE.json names; 9 taps and 3 channels: direct
    for C in (32, 64):
        for T, To in ((240, 240), (240, 120), (120, 120), (75, 75), (75, 38), (1024, 1024), (512, 512)):
            assert dft.preferred(C, 75, T, To)
        for T, To in ((240, 240), (240, 120), (1024, 512), (75, 75)):
            assert not dft.preferred(C, 9, T, To)
    assert not dft.preferred(3, 75, 240, 240) and not dft.supported(3, 75) and not dft.supported(64, 8)
    l = _lib.load()
    assert l.step_dft_rows(240, 75) == 280 and l.step_dft_rows(120, 75) == 160 and l.step_dft_rows(120, 39) == 140 and l.step_dft_rows(240, 8) == 0
    assert l.step_dft_groups(240) == 1 and l.step_dft_groups(1024) == 4
    assert dft.centre2(240, 1) == 239 and dft.centre2(120, 2) == 238 and dft.centre2(15, 1) == 15


def test_host_side_shape_guards_refuse_mismatched_spectra():
    """the kernels trust their arguments, so step_amd/dft.py checks shapes before it calls them: weights of the other
    mixing form, a spectrum on the wrong grid and a one-phase spectrum handed to the two-phase weight gradient raise
    (no GPU needed: the guards fire before any launch)"""
    C, T, Kt = 64, 240, 75
    rows_full = dft.geometry(T, Kt)[0]
    rows_ph = dft.geometry(*dft.phase_geometry(T, Kt))[0]
    assert (rows_full, rows_ph) == (280, 140)
    spec = torch.zeros(1, rows_full, C, 16)
    with pytest.raises(_lib.StepHipError, match="mix"):
        dft.mix(spec, torch.zeros((rows_full // 2) * 4 * C * 2 * C), C)        # 2C -> C weights, C -> C call
    with pytest.raises(_lib.StepHipError, match="mix"):
        dft.mix(torch.zeros(1, rows_ph, C, 16), torch.zeros(7), 2 * C, C)      # channel count / weight size mismatch
    with pytest.raises(_lib.StepHipError, match="conv_wgrad"):
        dft.conv_wgrad(torch.zeros(1, rows_ph, C, 16), torch.zeros(1, rows_ph, C, 16), T, Kt, False, (C, C, Kt, 1), T // 2, 2)
    with pytest.raises(_lib.StepHipError, match="conv_dgrad"):
        dft.conv_dgrad(spec, torch.zeros(1), T, Kt, 2, False)                  # dout's spectrum must be on the phase grid
    with pytest.raises(_lib.StepHipError, match="inverse"):
        dft.inverse(torch.zeros(1, rows_ph, C, 16), None, T, 1, T, Kt)
