"""Host side of the frequency-domain temporal convolution (csrc/tcn_dft.hip): the fixed cos / sin matrices, built in
float64 with integer-reduced angles and cached per shape and device, and the three operators on top of the C ABI.

Same statements of the reference as ops.tcn_conv / ops.tcn_wgrad: tcn[2] of st_gcn / st_gctn
(classifier_stgcn_real_only/net/classifier.py:156-162, generator_cvae/net/CVAE_stgcn.py:359-365) and its gradients."""
import math
import os

import torch

from . import _lib

_TABLES = {}


def _require(ok, msg):
    """shapes are checked here, on the host: the kernels trust their arguments"""
    if not ok:
        raise _lib.StepHipError("dft." + msg)


def supported(C, Kt):
    """square 32- / 64-channel temporal convs with an odd kernel of at most 79 taps (the model's: Kt = 75 and 9)"""
    return C in (32, 64) and Kt % 2 == 1 and Kt <= 79


def preferred(C, Kt, T, T_out):
    """cost model for the default fp32 mode: the frequency-domain path when its multiply-adds (two half-size transforms
    per pass + the per-bin mixing) are under 60 % of the direct kernels' - its GEMMs run at 0.5-0.65 of the fp32 MFMA
    peak, the direct kernels at 0.88 (75 taps: 7-14x fewer at T = 120 / 240, 2-4x at T = 1024; 9 taps: the direct kernels)"""
    if not supported(C, Kt):
        return False
    rows = _lib.query("step_dft_rows", T, Kt)
    cost_dft = 1.0 * rows * (T + T_out) * C * 16 + 8.0 * C * C * 16 * (rows // 2)
    cost_direct = 2.0 * C * C * Kt * T_out * 16
    return cost_dft < (0.6 if Kt >= 25 else 0.3) * cost_direct       # (short kernels: only with a wide margin)


def geometry(T, Kt):
    rows = _lib.query("step_dft_rows", T, Kt)
    return rows, rows // 2, rows - 1            # rows = 2 NF, NF, L


def _tile(mats, rows_out, inverse):
    """k-major float64 matrices [K][rows_out] -> the kernels' layout [groups][Kpad][slots x 32] (step_dft_table_geom):
    one matrix for the inverse, (cos, sin) for the forward transform, whose tiles share a group (slots i and mg + i);
    per k the slots go in fours [q][32][4], then a last pair [32][2]"""
    import ctypes
    wp, groups = ctypes.c_int(0), ctypes.c_int(0)
    mg = _lib.load().step_dft_table_geom(rows_out, int(inverse), ctypes.byref(wp), ctypes.byref(groups))
    wp, groups = wp.value, groups.value
    K = mats[0].shape[0]
    kpad = (K + 15) // 16 * 16
    slots = torch.zeros(groups, kpad, wp, 32, dtype=torch.float64)
    for h, m in enumerate(mats):
        full = torch.zeros(kpad, groups * mg * 32, dtype=torch.float64)
        full[:K, :m.shape[1]] = m
        slots[:, :, h * mg:(h + 1) * mg] = full.view(kpad, groups, mg, 32).permute(1, 0, 2, 3)     # [groups][kpad][mg][32]
    nq = wp // 4
    parts = [slots[:, :, :4 * nq].reshape(groups, kpad, nq, 4, 32).permute(0, 1, 2, 4, 3).reshape(groups, kpad, nq * 128)]
    if wp % 4:
        parts.append(slots[:, :, 4 * nq:].permute(0, 1, 3, 2).reshape(groups, kpad, 64))
    return torch.cat(parts, dim=2)


def _angles2(a, b, L):
    """2 pi (a b mod 2L) / (2L) for integer grids a (rows) and b (columns): angles in half steps of the grid, float64"""
    prod = torch.remainder(a.to(torch.int64)[:, None] * b.to(torch.int64)[None, :], 2 * L).to(torch.float64)
    return prod * (math.pi / L)


def centre2(K, step):
    """twice the time origin of a K-sample signal taken at frames step * k: its frames are paired about
    step * (ceil(K / 2) - 1/2) (csrc/tcn_dft.hip), and its spectrum is taken about that point"""
    return step * (2 * ((K + 1) // 2) - 1)


def fwd_matrix(K, step, T, Kt, device):
    """transform matrix for K samples taken at frames step * k of a length-T grid, in the half-size form: pairs
    j < ceil(K/2) of frames about the centre; cos(2 pi f step (j + 1/2) / L) rows (applied to the pair sums) tiled by
    and the -sin(...) rows (applied to the pair differences), tiled together by _tile"""
    key = ("fwd", K, step, T, Kt, str(device))
    if key not in _TABLES:
        rows, NF, L = geometry(T, Kt)
        Kh = (K + 1) // 2
        ang = _angles2(step * (2 * torch.arange(Kh) + 1), torch.arange(NF), L)          # [Kh][NF]
        t = _tile((torch.cos(ang), -torch.sin(ang)), NF, False)
        _TABLES[key] = t.to(torch.float32).to(device).contiguous()
    return _TABLES[key]


def inv_matrix(T_out, step, T, Kt, device):
    """inverse matrix onto the pairs j < ceil(T_out/2) of frames step * t about their centre: k < NF (real rows of the
    spectrum): kap_f / L cos(2 pi f step (j + 1/2) / L) -> E[j]; the imaginary rows: -kap_f / L sin(...) -> O[j]; the
    frames are E + O and E - O.  kap_0 = 1, else 2 (L is odd: no Nyquist bin).  Each half is padded with zero rows to
    NFp = round_up(NF, 16): the kernel walks the table in chunks of 16 k (csrc/tcn_dft.hip)"""
    key = ("inv", T_out, step, T, Kt, str(device))
    if key not in _TABLES:
        rows, NF, L = geometry(T, Kt)
        Th = (T_out + 1) // 2
        ang = _angles2(torch.arange(NF), step * (2 * torch.arange(Th) + 1), L)          # [NF][Th]
        kap = torch.full((NF, 1), 2.0 / L, dtype=torch.float64)
        kap[0, 0] = 1.0 / L
        zpad = torch.zeros(((NF + 15) // 16 * 16 - NF, Th), dtype=torch.float64)      # each half in whole chunks of 16 k
        m = torch.cat((kap * torch.cos(ang), zpad, -kap * torch.sin(ang), zpad), dim=0)
        _TABLES[key] = _tile((m,), Th, True).to(torch.float32).to(device).contiguous()
    return _TABLES[key]


def tap_tables(T, Kt, T_out, stride, device, d2=None):
    """tc / ts [NF][Kt]: cos / sin of 2 pi f (j - P + delta) / L; delta = the output's time origin minus the input's
    (0 at stride 1, -1/2 at stride 2 with even T): the phase between the two spectra lives in the weights.  d2 = 2 delta
    may be given (segments: the outputs sit P frames into the input window)"""
    if d2 is None:
        d2 = centre2(T_out, stride) - centre2(T, 1)
    key = ("taps", T, Kt, d2, str(device))
    if key not in _TABLES:
        rows, NF, L = geometry(T, Kt)
        ang = _angles2(torch.arange(NF), 2 * (torch.arange(Kt) - (Kt - 1) // 2) + d2, L)
        _TABLES[key] = (torch.cos(ang).to(torch.float32).to(device).contiguous(),
                        torch.sin(ang).to(torch.float32).to(device).contiguous())
    return _TABLES[key]


# ------------------------------------------------------------------ stride 2 as two stride-1 convs (polyphase)
def phases(C, T, Kt, stride, transposed):
    """2 when the conv runs as two stride-1 convs on the even and the odd input frames, else 1.  Output frame t' of a
    stride-2 conv reads the frames 2 t' + d, d = j - P: the taps with even d meet the even frames (phase 0) at offset
    m = d / 2, the taps with odd d the odd frames (phase 1) at m = (d - 1) / 2, so
        u[t'] = SUM_ph SUM_m w_ph[m] a_ph[t' + m]
    is a stride-1 correlation of Tp = T / 2 frames with |m| <= ceil(P / 2): every transform works on half the frames
    and ~half the bins (T = 240, Kt = 75: 160 rows instead of 288), the mixing takes 2 C -> C channels per bin."""
    if os.environ.get("STEP_DFT_NO_PHASES"):            # same-box A/B against the strided-matrix form
        return 1
    return 2 if (stride == 2 and not transposed and T % 2 == 0 and C in (32, 64) and Kt >= 3) else 1


def phase_geometry(T, Kt):
    """(Tp, Ktp): frames per phase and the tap extent 2 ceil(P / 2) + 1 that sets the phase grid's length"""
    P = (Kt - 1) // 2
    return T // 2, 2 * ((P + 1) // 2) + 1


def _phase_taps(Kt, device):
    """per tap j: its phase (d mod 2) as masks [2][Kt] and its offset m within the phase"""
    key = ("phase_taps", Kt, str(device))
    if key not in _TABLES:
        d = torch.arange(Kt) - (Kt - 1) // 2
        ph = torch.remainder(d, 2)
        m = (d - ph) // 2
        masks = torch.stack((ph == 0, ph == 1)).to(torch.float32).to(device)
        _TABLES[key] = (masks, m)
    return _TABLES[key]


def phase_tap_tables(T, Kt, device):
    """tc / ts [NF][Kt] on the phase grid: column j = cos / sin of 2 pi f m_j / L (input and output share their origin)"""
    key = ("phase_tap_tables", T, Kt, str(device))
    if key not in _TABLES:
        Tp, Ktp = phase_geometry(T, Kt)
        rows, NF, L = geometry(Tp, Ktp)
        ang = _angles2(torch.arange(NF), 2 * _phase_taps(Kt, device)[1], L)
        _TABLES[key] = (torch.cos(ang).to(torch.float32).to(device).contiguous(),
                        torch.sin(ang).to(torch.float32).to(device).contiguous())
    return _TABLES[key]


def weight_spectrum(w, C, T, Kt, transposed, T_out=None, stride=1):
    """(forward, backward) mixing weights (NF * 4 C^2 floats each: the triples of csrc/tcn_dft.hip) of the taps w (Co,Ci,Kt,1) [or (Ci,Co,Kt,1) when transposed]
    for a conv from T frames to T_out frames at `stride`"""
    if phases(C, T, Kt, stride, transposed) == 2:
        Tp, Ktp = phase_geometry(T, Kt)
        rows, NF, L = geometry(Tp, Ktp)
        tc, ts = phase_tap_tables(T, Kt, w.device)
        masks = _phase_taps(Kt, w.device)[0]
        w2 = (w.reshape(C, C, 1, Kt) * masks.view(1, 1, 2, Kt)).reshape(C, 2 * C, Kt)      # (Co, 2 ci + ph, Kt)
        wf = torch.empty(NF * 4 * C * 2 * C, dtype=torch.float32, device=w.device)
        wb = torch.empty_like(wf)
        _lib.call("step_dft_wspec", w2, tc, ts, wf, wb, 2 * C, C, NF, Kt, 0)
        return wf, wb
    rows, NF, L = geometry(T, Kt)
    tc, ts = tap_tables(T, Kt, T if T_out is None else T_out, stride, w.device)
    wf = torch.empty(NF * 4 * C * C, dtype=torch.float32, device=w.device)
    wb = torch.empty_like(wf)
    _lib.call("step_dft_wspec", w, tc, ts, wf, wb, C, C, NF, Kt, int(transposed))
    return wf, wb


def transform(x, scale, shift, relu, step, T, Kt):
    """spectrum [N][rows][C][16] of act(x) (N,C,K,16), whose K frames sit at step * k of a length-T grid"""
    N, C, K, V = x.shape
    rows, NF, L = geometry(T, Kt)
    spec = torch.empty((N, rows, C, 16), dtype=torch.float32, device=x.device)
    _lib.call("step_dft_fwd", x, scale, shift, int(relu), fwd_matrix(K, step, T, Kt, x.device), spec, N, C, K, rows)
    return spec


def mix(spec, wmat, C, Co=None):
    """per-bin channel mixing of spec [N][rows][C][16] -> [N][rows][Co][16] (Co = C unless given: the two phases)"""
    N, rows = spec.shape[0], spec.shape[1]
    Co = C if Co is None else Co
    _require(tuple(spec.shape[2:]) == (C, 16) and wmat.numel() == (rows // 2) * 4 * C * Co,
             "mix: spectrum %s / %d weights do not belong to a %d -> %d mixing of %d bins" % (tuple(spec.shape), wmat.numel(), C, Co, rows // 2))
    out = torch.empty((N, rows, Co, 16), dtype=torch.float32, device=spec.device)
    _lib.call("step_dft_mix", spec, wmat, out, N, C, Co, rows // 2)
    return out


def transform_phases(x, scale, shift, relu, T, Kt):
    """spectrum [N][rows][2 C][16] of the even and the odd frames of act(x) (N,C,T,16) (channel 2 c + ph)"""
    N, C = x.shape[0], x.shape[1]
    Tp, Ktp = phase_geometry(T, Kt)
    rows, NF, L = geometry(Tp, Ktp)
    spec = torch.empty((N, rows, 2 * C, 16), dtype=torch.float32, device=x.device)
    _lib.call("step_dft_fwd_ph", x, scale, shift, int(relu), fwd_matrix(Tp, 1, Tp, Ktp, x.device), spec, N, C, Tp, rows, 2)
    return spec


def inverse(spec, bias, T_out, step, T, Kt, sums=None, epilogue=None):
    """(N,C,T_out,16) at frames step * t of the length-T grid (+ bias); sums: float64 [2C + 1] to receive the BatchNorm
    statistics of the result (SUM, SUM of squares per channel); epilogue: (out_scale, out_shift, res, res_scale,
    res_shift, relu) - the eval-mode tail of a block"""
    N, rows, C, V = spec.shape
    _require(rows == geometry(T, Kt)[0] and V == 16 and step * (T_out - 1) < T,
             "inverse: spectrum %s for T = %d, Kt = %d, %d frames at step %d" % (tuple(spec.shape), T, Kt, T_out, step))
    out = torch.empty((N, C, T_out, 16), dtype=torch.float32, device=spec.device)
    ws, nb = None, 0
    if sums is not None:
        nb = _lib.query("step_dft_inv_ws_bytes", N, C, T_out)
        ws = torch.empty(nb, dtype=torch.uint8, device=spec.device)
    osc, osh, res, rsc, rsh, relu = epilogue if epilogue is not None else (None, None, None, None, None, False)
    _lib.call("step_dft_inv", spec, inv_matrix(T_out, step, T, Kt, spec.device), bias, osc, osh, res, rsc, rsh, int(relu),
              out, N, C, rows, T_out, sums, ws, nb)
    return out


def conv_forward(y, scale, shift, relu, wf, bias, T_out, stride, Kt, sums=None, epilogue=None, transposed=False):
    """u = conv(act(y)) + bias -> (u, spectrum of act(y)); wf from weight_spectrum (same T, stride, transposed)"""
    N, C, T, V = y.shape
    if phases(C, T, Kt, stride, transposed) == 2:
        Tp, Ktp = phase_geometry(T, Kt)
        sa = transform_phases(y, scale, shift, relu, T, Kt)
        u = inverse(mix(sa, wf, 2 * C, C), bias, T_out, 1, Tp, Ktp, sums, epilogue)
        return u, sa
    sa = transform(y, scale, shift, relu, 1, T, Kt)
    u = inverse(mix(sa, wf, C), bias, T_out, stride, T, Kt, sums, epilogue)
    return u, sa


def transform_dout(du, stride, T, Kt, transposed=False):
    """spectrum of the output gradient du (N,C,T_out,16) of a conv over T input frames"""
    C = du.shape[1]
    if phases(C, T, Kt, stride, transposed) == 2:
        Tp, Ktp = phase_geometry(T, Kt)                 # T_out = Tp frames on the phase grid
        return transform(du, None, None, False, 1, Tp, Ktp)
    return transform(du, None, None, False, stride, T, Kt)


def conv_dgrad(sdu, wb, T, Kt, stride=1, transposed=False):
    """gradient with respect to the activated input, (N,C,T,16), from the spectrum of dout"""
    N, rows, C, V = sdu.shape
    if phases(C, T, Kt, stride, transposed) == 2:
        Tp, Ktp = phase_geometry(T, Kt)
        _require(rows == geometry(Tp, Ktp)[0], "conv_dgrad: the spectrum of dout is not on the phase grid (transform_dout)")
        sda = mix(sdu, wb, C, 2 * C)
        da = torch.empty((N, C, T, 16), dtype=torch.float32, device=sdu.device)
        _lib.call("step_dft_inv_ph", sda, inv_matrix(Tp, 1, Tp, Ktp, sdu.device), da, N, C, rows, Tp, 2)
        return da
    _require(rows == geometry(T, Kt)[0], "conv_dgrad: spectrum rows %d for T = %d" % (rows, T))
    return inverse(mix(sdu, wb, C), None, T, 1, T, Kt)


def conv_wgrad(sdu, sa, T, Kt, transposed, wshape, T_out=None, stride=1):
    N, rows, C, V = sdu.shape
    ph = phases(C, T, Kt, stride, transposed)
    _require(tuple(sa.shape) == (N, rows, ph * C, 16) and V == 16,
             "conv_wgrad: spectra %s (dout) / %s (input) do not belong to this conv (%d phase(s))" % (tuple(sdu.shape), tuple(sa.shape), ph))
    if ph == 2:
        tc, ts = phase_tap_tables(T, Kt, sa.device)
        dw2 = torch.empty((C, 2 * C, Kt), dtype=torch.float32, device=sa.device)
        db = torch.empty(C, dtype=torch.float32, device=sa.device)
        nb = _lib.query("step_dft_wgrad_ws_bytes", N, 2 * C, C, rows // 2)
        ws = torch.empty(nb, dtype=torch.uint8, device=sa.device)
        _lib.call("step_dft_wgrad", sdu, sa, tc, ts, dw2, db, N, 2 * C, C, rows // 2, Kt, 0, ws, nb)
        odd = _phase_taps(Kt, sa.device)[0][1].view(1, 1, Kt) > 0           # every tap's gradient sits in its own phase
        dw2 = dw2.view(C, C, 2, Kt)
        return torch.where(odd, dw2[:, :, 1], dw2[:, :, 0]).reshape(wshape), db
    tc, ts = tap_tables(T, Kt, T if T_out is None else T_out, stride, sa.device)
    dw = torch.empty(wshape, dtype=torch.float32, device=sa.device)
    db = torch.empty(C, dtype=torch.float32, device=sa.device)
    nb = _lib.query("step_dft_wgrad_ws_bytes", N, C, C, rows // 2)
    ws = torch.empty(nb, dtype=torch.uint8, device=sa.device)
    _lib.call("step_dft_wgrad", sdu, sa, tc, ts, dw, db, N, C, C, rows // 2, Kt, int(transposed), ws, nb)
    return dw, db


# ------------------------------------------------------------------ long sequences: overlap-save (eval forward)
SEGMENT = 256


def seg_plan(C, T, T_out, stride, Kt, N=1):
    """(S, seg, K, Tv, stride) when the eval forward of a conv over T frames is cheaper in S segments of `seg` input
    frames = seg / stride output frames (each a transform of K = seg + Kt - 1 input frames with L >= K bins: S K^2
    instead of T^2), else None.  Cost = multiply-adds + 20 x bytes (the two roofs' ratio: these kernels pay for both,
    DESIGN 4.6); taken under 80 % of the whole-row form - T = 1024 (configs[4]) yes, T <= 512 no.  Tv = K - P is the
    sequence length whose geometry (step_dft_rows) gives the segment's L."""
    if stride not in (1, 2) or T_out != (T - 1) // stride + 1 or not supported(C, Kt) or T < 2 * SEGMENT:
        return None
    P = (Kt - 1) // 2
    seg = SEGMENT
    S = (T + seg - 1) // seg
    if N * S > 65535:                       # (the segment index rides in the grid's z dimension with the sample)
        return None
    K = seg + Kt - 1
    Tv = K - P
    rows_w = _lib.query("step_dft_rows", T, Kt)
    rows_s = _lib.query("step_dft_rows", Tv, Kt)

    def cost(rows, frames_in, frames_out, nseg):
        flops = nseg * (rows * (frames_in + frames_out) * C * 16 + 6.0 * C * C * 16 * (rows // 2))
        nbytes = (4.0 * nseg * rows + T + T_out) * C * 64
        return flops + 20.0 * nbytes
    if cost(rows_s, K, seg // stride, S) < 0.8 * cost(rows_w, T, T_out, 1):
        return S, seg, K, Tv, stride
    return None


def weight_spectrum_seg(w, C, Kt, transposed, plan):
    """forward mixing weights for the segments of `plan`: the outputs of a segment sit P frames into its input window"""
    S, seg, K, Tv, stride = plan
    rows, NF, L = geometry(Tv, Kt)
    d2 = (Kt - 1) + centre2(seg // stride, stride) - centre2(K, 1)
    tc, ts = tap_tables(Tv, Kt, seg // stride, stride, w.device, d2)
    wf = torch.empty(NF * 4 * C * C, dtype=torch.float32, device=w.device)
    wb = torch.empty_like(wf)
    _lib.call("step_dft_wspec", w, tc, ts, wf, wb, C, C, NF, Kt, int(transposed))
    return wf


def conv_forward_seg(y, scale, shift, relu, wf, bias, Kt, plan, epilogue=None):
    """u = conv(act(y)) + bias (+ the eval-mode tail), segment by segment; wf from weight_spectrum_seg"""
    N, C, T, V = y.shape
    S, seg, K, Tv, stride = plan
    T_out = (T - 1) // stride + 1
    rows, NF, L = geometry(Tv, Kt)
    spec = torch.empty((N * S, rows, C, 16), dtype=torch.float32, device=y.device)
    _lib.call("step_dft_fwd_seg", y, scale, shift, int(relu), fwd_matrix(K, 1, Tv, Kt, y.device), spec, N, C, T, K, rows,
              S, seg, -((Kt - 1) // 2))
    spec = mix(spec, wf, C)
    out = torch.empty((N, C, T_out, 16), dtype=torch.float32, device=y.device)
    osc, osh, res, rsc, rsh, orelu = epilogue if epilogue is not None else (None, None, None, None, None, False)
    _lib.call("step_dft_inv_seg", spec, inv_matrix(seg // stride, stride, Tv, Kt, y.device), bias, osc, osh, res, rsc, rsh,
              int(orelu), out, N, C, rows, seg // stride, T_out, S, seg // stride, None, None, 0)
    return out
