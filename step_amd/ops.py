"""Autograd bindings of the HIP kernels: one Function per fused stage of the ST-GCN stack.

Everything here runs on the GPU through the C ABI (``_lib.call``); there is no CPU path.
Tensor layout is (N, C, T, V=16) fp32 contiguous throughout.
"""
import torch

from . import _lib
from . import dft as _dft_mod

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# A/B switch (tools/gpu_ab_step.sh): STEP_SEPARATE_STATS=1 keeps every BatchNorm's statistics as a separate pass over
# the tensor, as before round 3, instead of taking them in the producing kernel's epilogue
import os as _os
_FUSED_STATS = not _os.environ.get("STEP_SEPARATE_STATS")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _f32(t):
    if t.dtype != torch.float32:
        raise _lib.StepHipError("step_amd kernels are fp32; got %s" % t.dtype)
    return t.contiguous()


def a_nnz_max(A):
    """Largest number of non-zeros in any (k, column) or (k, row) of the fixed adjacency A (host
    side, once per model).  A * edge_importance keeps A's zeros, so this bounds A_eff too."""
    nz = A != 0
    return int(max(nz.sum(dim=1).max().item(), nz.sum(dim=2).max().item()))


# ------------------------------------------------------------------------ BatchNorm
_SYNC_BN = {"on": False, "group": None}


def set_sync_bn(enabled, group=None):
    """Train-mode BatchNorm statistics over all ranks of `group` (step_amd.parallel.sync_batchnorm)."""
    _SYNC_BN["on"], _SYNC_BN["group"] = bool(enabled), group


def _sync_world():
    import torch.distributed as dist
    if not _SYNC_BN["on"] or not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(_SYNC_BN["group"])


def _sync_sums(sums):
    """all-reduce [sums..., element count] (fp64) over the ranks; nothing comes back to the host"""
    from . import parallel
    parallel._all_reduce_sum(sums, _SYNC_BN["group"])


def _sync_many(bufs):
    """ONE all-reduce for several [sums..., count] buffers (statistics of BatchNorms whose inputs exist at the same
    time: tcn[0] and residual[1] after the graph conv, tcn[3] and residual[1] in the block's backward tail)"""
    if len(bufs) == 1:
        _sync_sums(bufs[0])
        return
    flat = torch.cat(bufs)
    _sync_sums(flat)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()])
        off += b.numel()


def bn_finalize_sums(sums, count, bn, synced, momentum=BN_MOMENTUM):
    """(mean, invstd, scale, shift) + running-stat update from [SUM x, SUM x^2] per group (2G doubles, slot 2G = the
    element count when the sums were combined over ranks): the second half of step_bn_train_prepare, for
    statistics a producer kernel took in its epilogue (step_gcn_fwd_stats)"""
    G = bn.weight.numel()
    dev = sums.device
    mean = torch.empty(G, dtype=torch.float32, device=dev)
    invstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    _lib.call("step_bn_train_finalize", sums, 0.0 if synced else float(count), bn.weight, bn.bias, bn.running_mean,
              bn.running_var, momentum, BN_EPS, mean, invstd, scale, shift, G)
    bn.num_batches_tracked += 1
    return mean, invstd, scale, shift


def tensor_key(*tensors):
    """identity + in-place version of tensors, for the inference caches below: an optimizer step, load_state_dict
    (copy_) and any other in-place write bump `_version`, .to() / re-assignment change `data_ptr`.  Writes through
    `.data` do NOT bump it: the blocks drop their caches on train() and offer invalidate_caches() for that case."""
    return tuple((t.data_ptr(), t._version, str(t.device)) if t is not None else None for t in tensors)


def cached(cache, name, key, build):
    """cache[name] if it was built from `key`, else build() (cache None: always build)"""
    if cache is None:
        return build()
    ent = cache.get(name)
    if ent is None or ent[0] != key:
        ent = (key, build())
        cache[name] = ent
    return ent[1]


def bn_eval_prepare(weight, bias, running_mean, running_var):
    G = running_mean.numel()
    scale = torch.empty(G, dtype=torch.float32, device=running_mean.device)
    shift = torch.empty_like(scale)
    _lib.call("step_bn_eval_prepare", weight, bias, running_mean, running_var, BN_EPS, scale, shift, G)
    return scale, shift


def bn_train_prepare(x, bn_mode, weight, bias, running_mean, running_var, momentum=BN_MOMENTUM):
    N, C, T, V = x.shape
    G = C * V if bn_mode else C
    dev = x.device
    mean = torch.empty(G, dtype=torch.float32, device=dev)
    invstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    nb = _lib.query("step_bn_ws_bytes", G)
    ws = _ws(nb, dev)
    if _sync_world() > 1:
        # synchronised BatchNorm: local SUM x, SUM x^2 and the local element count travel in one fp64 buffer;
        # the finalize kernel reads the combined count from its last slot (count argument 0)
        sums = torch.empty(2 * G + 1, dtype=torch.float64, device=dev)
        _lib.call("step_bn_train_stats", x, N, C, T, V, bn_mode, sums, ws, nb)
        sums[2 * G] = float(N * T if bn_mode else N * T * V)
        _sync_sums(sums)
        _lib.call("step_bn_train_finalize", sums, 0.0, weight, bias, running_mean, running_var, momentum, BN_EPS,
                  mean, invstd, scale, shift, G)
        return mean, invstd, scale, shift
    _lib.call("step_bn_train_prepare", x, N, C, T, V, bn_mode, weight, bias, running_mean, running_var,
              momentum, BN_EPS, mean, invstd, scale, shift, ws, nb)
    return mean, invstd, scale, shift


def bn_prepare(x, bn_mode, bn, training):
    """(mean, invstd, scale, shift) of a torch BatchNorm parameter container ``bn`` for input x."""
    if training:
        out = bn_train_prepare(x, bn_mode, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        bn.num_batches_tracked += 1
        return out
    scale, shift = bn_eval_prepare(bn.weight, bn.bias, bn.running_mean, bn.running_var)
    invstd = torch.rsqrt(bn.running_var + BN_EPS)
    return bn.running_mean, invstd, scale, shift


def bn_eval_cached(cache, name, bn):
    """eval-mode (scale, shift) of a BatchNorm container, computed once per parameter version"""
    return cached(cache, name, tensor_key(bn.weight, bn.bias, bn.running_mean, bn.running_var),
                  lambda: bn_eval_prepare(bn.weight, bn.bias, bn.running_mean, bn.running_var))


def affine_act(x, scale, shift, res, rscale, rshift, relu, bn_mode=0):
    N, C, T, V = x.shape
    out = torch.empty_like(x)
    _lib.call("step_affine_act", x, scale, shift, res, rscale, rshift, int(relu), out, N, C, T, V, bn_mode)
    return out


def bn_backward(dy, x, mask_src, scale, shift, mask_mode, mean, invstd, gamma, training, bn_mode,
                want_dx=True, dx_accum=None):
    """returns (dx, dgamma, dbeta) of y = act(bn(x)); see step_bn_bwd_* in step_hip.h"""
    N, C, T, V = x.shape
    G = C * V if bn_mode else C
    dev = x.device
    nb = _lib.query("step_bn_ws_bytes", G)
    ws = _ws(nb, dev)
    sums = torch.empty(2 * G + 1, dtype=torch.float64, device=dev)
    _lib.call("step_bn_bwd_reduce", dy, x, mask_src, scale, shift, mask_mode, N, C, T, V, bn_mode, sums, ws, nb)
    dgamma = torch.empty(G, dtype=torch.float32, device=dev)
    dbeta = torch.empty_like(dgamma)
    dx = None
    accumulate = 0
    if want_dx:
        if dx_accum is not None:
            dx, accumulate = dx_accum, 1
        else:
            dx = torch.empty_like(x)
    if training and _sync_world() > 1:
        # synchronised BatchNorm: dgamma / dbeta come from this rank's sums (the gradient all-reduce averages them
        # like every other parameter gradient); dx needs SUM g and SUM g*x over ALL ranks.  The apply kernel divides
        # by the local element count, so the combined sums are scaled by local / combined count on the device.
        _lib.call("step_bn_bwd_apply", dy, x, mask_src, scale, shift, mask_mode, mean, invstd, gamma, sums,
                  1, 0, None, dgamma, dbeta, N, C, T, V, bn_mode, ws, nb)
        if dx is not None:
            local = float(N * T if bn_mode else N * T * V)
            sums[2 * G] = local
            _sync_sums(sums)
            sums[:2 * G] *= local / sums[2 * G]
            _lib.call("step_bn_bwd_apply", dy, x, mask_src, scale, shift, mask_mode, mean, invstd, gamma, sums,
                      1, accumulate, dx, None, None, N, C, T, V, bn_mode, ws, nb)
        return dx, dgamma, dbeta
    _lib.call("step_bn_bwd_apply", dy, x, mask_src, scale, shift, mask_mode, mean, invstd, gamma, sums,
              int(training), accumulate, dx, dgamma, dbeta, N, C, T, V, bn_mode, ws, nb)
    return dx, dgamma, dbeta


def bn_backward_tail_synced(dout, out, u, r, m2, i2, g2, mr, ir, gr):
    """bn_backward_tail_pair under synchronised BatchNorm: the two reduces, ONE all-reduce of both [SUM g, SUM g*x,
    count] buffers, the two applies (dgamma / dbeta from this rank's sums, dx from the combined ones)"""
    N, C, T, V = u.shape
    dev = u.device
    nb = _lib.query("step_bn_ws_bytes", C)
    local = float(N * T * V)
    res = []
    bufs = []
    for x, mean, invstd, gamma in ((u, m2, i2, g2), (r, mr, ir, gr)):
        ws = _ws(nb, dev)
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        _lib.call("step_bn_bwd_reduce", dout, x, out, None, None, 1, N, C, T, V, 0, sums, ws, nb)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty_like(dgamma)
        _lib.call("step_bn_bwd_apply", dout, x, out, None, None, 1, mean, invstd, gamma, sums,
                  1, 0, None, dgamma, dbeta, N, C, T, V, 0, ws, nb)
        sums[2 * C] = local
        bufs.append(sums)
        res.append([x, mean, invstd, gamma, ws, dgamma, dbeta])
    _sync_many(bufs)
    outs = []
    for sums, (x, mean, invstd, gamma, ws, dgamma, dbeta) in zip(bufs, res):
        sums[:2 * C] *= local / sums[2 * C]
        dx = torch.empty_like(x)
        _lib.call("step_bn_bwd_apply", dout, x, out, None, None, 1, mean, invstd, gamma, sums,
                  1, 0, dx, None, None, N, C, T, V, 0, ws, nb)
        outs += [dx, dgamma, dbeta]
    return tuple(outs)


def bn_backward_tail_pair(dout, out, u, r, m2, i2, g2, mr, ir, gr, training):
    """(du, dgamma2, dbeta2, dr, dgamma_r, dbeta_r) of out = relu(bn2(u) + bn_r(r)): both BN backward passes
    in one reduce + one apply (they share dout and the ReLU mask of out)"""
    N, C, T, V = u.shape
    dev = u.device
    nb = 2 * _lib.query("step_bn_ws_bytes", C)
    ws = _ws(nb, dev)
    du, dr = torch.empty_like(u), torch.empty_like(r)
    dg2 = torch.empty(C, dtype=torch.float32, device=dev)
    db2, dgr, dbr = torch.empty_like(dg2), torch.empty_like(dg2), torch.empty_like(dg2)
    _lib.call("step_bn_bwd_tail_pair", dout, out, u, r, m2, i2, g2, mr, ir, gr, int(training), du, dr,
              dg2, db2, dgr, dbr, N, C, T, V, ws, nb)
    return du, dg2, db2, dr, dgr, dbr


# --------------------------------------------------------------------- temporal conv
def tcn_bf16_supported(Co, Ci, Kt, stride, mode):
    """True when the bf16-operand MFMA kernels cover this temporal conv (else use fp32)."""
    return _lib.query("step_tcn_packed_bytes_bf16", Co, Ci, Kt, stride, mode) > 0


def tcn_wgrad_bf16_supported(Co, Ci, Kt, stride):
    return _lib.query("step_tcn_wgrad_bf16_ws_bytes", 1, Ci, Co, 1, 1, Kt, stride) > 0


def tcn_pack(w, Co, Ci, Kt, stride, mode, bf16=False):
    if bf16:                                    # True: bf16 operands; "split": fp32 operands as bf16 hi + lo
        sfx = "split" if bf16 == "split" else "bf16"
        nb = _lib.query("step_tcn_packed_bytes_" + sfx, Co, Ci, Kt, stride, mode)
        if nb == 0:
            raise _lib.StepHipError("no bf16 temporal kernel for C=%d Kt=%d stride=%d mode=%d" % (Co, Kt, stride, mode))
        wp = torch.empty(nb, dtype=torch.uint8, device=w.device)
        _lib.call("step_tcn_pack_weight_" + sfx, w, wp, Co, Ci, Kt, stride, mode)
        return wp
    n = _lib.query("step_tcn_packed_elems", Co, Ci, Kt, stride, mode)
    if n == 0:
        raise _lib.StepHipError("temporal kernel Kt=%d stride=%d is not supported" % (Kt, stride))
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.call("step_tcn_pack_weight", w, wp, Co, Ci, Kt, stride, mode)
    return wp


def tcn_conv(inp, in_scale, in_shift, in_relu, wp, bias, out_scale, out_shift, res, res_scale, res_shift,
             out_relu, N, Ci, Co, Tin, Tout, Kt, stride, mode, bf16=False):
    if mode < 2:
        out = torch.empty((N, Co, Tout, 16), dtype=torch.float32, device=inp.device)
    else:
        out = torch.empty((N, Ci, Tin, 16), dtype=torch.float32, device=inp.device)
    fn = "step_tcn_conv_split" if bf16 == "split" else ("step_tcn_conv_bf16" if bf16 else "step_tcn_conv")
    _lib.call(fn, inp, in_scale, in_shift, int(in_relu), wp,
              bias, out_scale, out_shift, res, res_scale, res_shift, int(out_relu), out, N, Ci, Co, Tin, Tout,
              Kt, stride, mode)
    return out


def tcn_wgrad(dout, inp, in_scale, in_shift, in_relu, N, Ci, Co, Tin, Tout, Kt, stride, transposed, wshape,
              bf16=False):
    dev = dout.device
    dw = torch.empty(wshape, dtype=torch.float32, device=dev)
    db = torch.empty(Co, dtype=torch.float32, device=dev)
    if bf16:                                    # True / "bf16": bf16 operands; "split": fp32 operands as bf16 hi + lo
        nb = _lib.query("step_tcn_wgrad_bf16_ws_bytes", N, Ci, Co, Tin, Tout, Kt, stride)
        if nb == 0:
            raise _lib.StepHipError("no bf16 wgrad kernel for C=%d Kt=%d stride=%d" % (Co, Kt, stride))
        ws = _ws(nb, dev)
        if bf16 == "split":
            _lib.call("step_tcn_wgrad_split", dout, inp, in_scale, in_shift, int(in_relu), dw, db, N, Ci, Co, Tin,
                      Tout, Kt, stride, int(transposed), ws, nb)
        else:
            _lib.call("step_tcn_wgrad_bf16_t", dout, inp, in_scale, in_shift, int(in_relu), dw, db, N, Ci, Co, Tin,
                      Tout, Kt, stride, int(transposed), ws, nb)
        return dw, db
    nb = _lib.query("step_tcn_wgrad_ws_bytes", N, Ci, Co, Tin, Tout, Kt)
    ws = _ws(nb, dev)
    _lib.call("step_tcn_wgrad", dout, inp, in_scale, in_shift, int(in_relu), dw, db, N, Ci, Co, Tin, Tout,
              Kt, stride, int(transposed), ws, nb)
    return dw, db


# ------------------------------------------------------------------ one ST-GCN block
class BlockSpec:
    """Static description of one st_gcn / st_gctn block plus its buffers (no grad)."""

    def __init__(self, Ci, Co, Kt, stride, residual, transposed, nnz, bn1, bn2, bnr, data_bn=None,
                 bf16=False):
        self.Ci, self.Co, self.Kt, self.stride = Ci, Co, Kt, stride
        # bf16 MFMA operands for the temporal conv where those kernels exist (fp32 otherwise)
        # bf16 == "split": fp32 operands as bf16 hi + lo on the bf16 matrix cores (step_tcn_conv_split,
        # step_tcn_wgrad_split); "split_wgrad": only the weight gradient that way, the two convs on fp32 MFMAs
        # bf16 == "dft": fp32 arithmetic, temporal conv and its gradients in the frequency domain (step_amd/dft.py)
        from . import dft as _dft
        #   "dft": wherever those kernels exist; "auto_dft" (the default fp32 mode): where their cost model says they win
        self.dft = {"dft": "force", "auto_dft": "auto"}.get(bf16, False) if _dft.supported(Co, Kt) else False
        if bf16 in ("dft", "auto_dft"):
            bf16 = False
        split = bf16 in ("split", "split_wgrad", "split_gcn")
        conv_bf = True if bf16 is True else ("split" if bf16 in ("split", "split_gcn") else False)
        tr_ok = True        # st_gctn (ConvTranspose2d) blocks: pack modes 1 / 3, `transposed` in the weight gradients
        m0, m2 = (1, 3) if transposed else (0, 2)
        self.bf16_fwd = bool(conv_bf) and tr_ok and tcn_bf16_supported(Co, Co, Kt, stride, m0)
        self.bf16_dgrad = bool(conv_bf) and tr_ok and tcn_bf16_supported(Co, Co, Kt, stride, m2)
        self.bf16_wgrad = bool(bf16) and tr_ok and tcn_wgrad_bf16_supported(Co, Co, Kt, stride)
        if split:
            self.bf16_fwd = "split" if self.bf16_fwd else False
            self.bf16_dgrad = "split" if self.bf16_dgrad else False
            self.bf16_wgrad = "split" if self.bf16_wgrad else False
        # "split_gcn": "split" plus the graph conv that way (step_gcn_fwd_split / step_gcn_bwd_split; the first block
        # stays fp32)
        self.gcn_fwd = "step_gcn_fwd_split" if bf16 == "split_gcn" else "step_gcn_fwd"
        self.gcn_bwd = "step_gcn_bwd_split" if bf16 == "split_gcn" else "step_gcn_bwd_idres"
        self.residual = residual            # 'none' | 'identity' | 'conv'
        self.transposed = transposed
        self.nnz = nnz
        self.bn1, self.bn2, self.bnr, self.data_bn = bn1, bn2, bnr, data_bn
        self.grad_enabled = torch.is_grad_enabled()     # of the caller that builds the spec, right before apply()
        self.cache = None       # dict owned by the block module: packed weights / eval BN coefficients (inference)


class StGcnBlockFn(torch.autograd.Function):
    """[data_bn ->] graph conv -> BN -> ReLU -> temporal conv -> BN -> (+ residual) -> ReLU.

    st_gcn.forward classifier.py:185-191 / st_gctn.forward CVAE_stgcn.py:388-394, with the
    ``data_bn`` of Classifier.forward (classifier.py:68-74) folded in for the first block.
    Argument order after (spec, training, x, A_eff):
        [dbn_w, dbn_b,] gcn_w, gcn_b, bn1_w, bn1_b, tcn_w, tcn_b, bn2_w, bn2_b [, res_w, res_b, bnr_w, bnr_b]
    """

    @staticmethod
    def forward(ctx, spec, training, x, A_eff, *params):
        s = spec
        x = _f32(x)
        A_eff = _f32(A_eff)
        N, Ci, T, V = x.shape
        if V != 16 or Ci != s.Ci:
            raise _lib.StepHipError("block expects (N,%d,T,16), got %s" % (s.Ci, tuple(x.shape)))
        params = [_f32(p) for p in params]
        i = 0
        dbn_w = dbn_b = None
        if s.data_bn is not None:
            dbn_w, dbn_b = params[0], params[1]
            i = 2
        gcn_w, gcn_b, bn1_w, bn1_b, tcn_w, tcn_b, bn2_w, bn2_b = params[i:i + 8]
        res_w = res_b = bnr_w = bnr_b = None
        if s.residual == "conv":
            res_w, res_b, bnr_w, bnr_b = params[i + 8:i + 12]
        Co, Kt, st = s.Co, s.Kt, s.stride
        To = (T - 1) // st + 1
        K = A_eff.shape[0]
        if s.data_bn is not None and s.residual == "identity":
            # the reference's identity residual adds the NORMALISED input (data_bn runs first,
            # classifier.py:68-78); with data_bn folded into the graph-conv load only the raw x exists
            raise _lib.StepHipError("a first block with an identity residual (in == out channels, stride 1) "
                                    "behind a folded data_bn is not supported")
        # needs_input_grad mirrors tensor.requires_grad whatever the grad mode is: under torch.no_grad() nothing
        # will be differentiated even though the parameters require grad (and inside forward() the grad mode is
        # always off, so the caller's mode travels in the spec)
        need_grad = s.grad_enabled and any(ctx.needs_input_grad)
        dev = x.device

        fused = (not training) and (not need_grad)
        if fused:
            return StGcnBlockFn._forward_inference(s, x, A_eff, gcn_w, gcn_b, tcn_w, tcn_b, res_w, res_b)

        # data_bn statistics (training) or running stats (eval); applied inside the gcn load
        m0 = i0 = s0 = h0 = None
        if s.data_bn is not None:
            m0, i0, s0, h0 = bn_prepare(x, 1, s.data_bn, training)

        y = torch.empty((N, Co, T, 16), dtype=torch.float32, device=dev)
        r = torch.empty((N, Co, To, 16), dtype=torch.float32, device=dev) if res_w is not None else None
        rstats = None
        if training and s.gcn_fwd == "step_gcn_fwd" and _FUSED_STATS:
            # train mode: the graph conv also returns SUM y, SUM y^2 (and the residual conv's) per channel - taken in
            # its epilogue where that kernel form exists, so BN1 / the residual BN never read y / r for statistics
            ysums = torch.empty(2 * Co + 1, dtype=torch.float64, device=dev)
            rsums = torch.empty(2 * Co + 1, dtype=torch.float64, device=dev) if res_w is not None else None
            nbw = _lib.query("step_gcn_fwd_stats_ws_bytes", Co)
            _lib.call("step_gcn_fwd_stats", x, s0, h0, gcn_w, gcn_b, int(s.transposed), A_eff, s.nnz, y,
                      res_w, res_b, int(s.transposed), r, N, Ci, Co, K, T, st, ysums, rsums, _ws(nbw, dev), nbw)
            synced = _sync_world() > 1
            if synced:
                ysums[2 * Co] = float(N * T * 16)
                if rsums is not None:
                    rsums[2 * Co] = float(N * To * 16)
                _sync_many([ysums] if rsums is None else [ysums, rsums])
            m1, i1, s1, h1 = bn_finalize_sums(ysums, N * T * 16, s.bn1, synced)
            if rsums is not None:
                rstats = bn_finalize_sums(rsums, N * To * 16, s.bnr, synced)
        else:
            _lib.call(s.gcn_fwd, x, s0, h0, gcn_w, gcn_b, int(s.transposed), A_eff, s.nnz, y,
                      res_w, res_b, int(s.transposed), r, N, Ci, Co, K, T, st)
            m1, i1, s1, h1 = bn_prepare(y, 0, s.bn1, training)
        mode = 1 if s.transposed else 0
        bf = s.bf16_fwd
        mr = ir = sr = hr = None
        sa = wb_dft = None
        use_dft = s.dft == "force" or (s.dft == "auto" and _dft_mod.preferred(Co, Kt, T, To))
        if use_dft:
            # frequency domain: the spectrum of relu(bn1(y)) is kept for the weight gradient
            wf_dft, wb_dft = _dft_mod.weight_spectrum(tcn_w, Co, T, Kt, s.transposed, To, st)
            usums = torch.empty(2 * Co + 1, dtype=torch.float64, device=dev) if (training and _FUSED_STATS) else None
            u, sa = _dft_mod.conv_forward(y, s1, h1, True, wf_dft, tcn_b, To, st, Kt, usums, transposed=s.transposed)
            del wf_dft
            if not need_grad:
                sa = wb_dft = None
            if usums is not None:
                synced = _sync_world() > 1
                if synced:
                    usums[2 * Co] = float(N * To * 16)
                    _sync_sums(usums)
                m2, i2, s2, h2 = bn_finalize_sums(usums, N * To * 16, s.bn2, synced)
            else:
                m2, i2, s2, h2 = bn_prepare(u, 0, s.bn2, training)
            wp = None
        else:
            wp = tcn_pack(tcn_w, Co, Co, Kt, st, mode, bf16=bf)
        if use_dft:
            pass
        elif training and _FUSED_STATS:
            # train mode: SUM u, SUM u^2 per channel come out of the conv's epilogue
            u = torch.empty((N, Co, To, 16), dtype=torch.float32, device=dev)
            usums = torch.empty(2 * Co + 1, dtype=torch.float64, device=dev)
            if bf:
                nbt = _lib.query("step_tcn_conv_stats_bf16_ws_bytes", Co)
                _lib.call("step_tcn_conv_stats_bf16", y, s1, h1, 1, wp, tcn_b, u, N, Co, Co, T, To, Kt, st, mode,
                          int(bf == "split"), usums, _ws(nbt, dev), nbt)
            else:
                nbt = _lib.query("step_tcn_conv_stats_ws_bytes", N, Co, To)
                _lib.call("step_tcn_conv_stats", y, s1, h1, 1, wp, tcn_b, u, N, Co, Co, T, To, Kt, st, mode,
                          usums, _ws(nbt, dev), nbt)
            synced = _sync_world() > 1
            if synced:
                usums[2 * Co] = float(N * To * 16)
                _sync_sums(usums)
            m2, i2, s2, h2 = bn_finalize_sums(usums, N * To * 16, s.bn2, synced)
        else:
            u = tcn_conv(y, s1, h1, True, wp, tcn_b, None, None, None, None, None, False,
                         N, Co, Co, T, To, Kt, st, mode, bf16=bf)
            m2, i2, s2, h2 = bn_prepare(u, 0, s.bn2, training)
        res_t = None
        if s.residual == "conv":
            mr, ir, sr, hr = rstats if rstats is not None else bn_prepare(r, 0, s.bnr, training)
            res_t = r
        elif s.residual == "identity":
            res_t = x
        out = affine_act(u, s2, h2, res_t, sr, hr, True, 0)

        if need_grad:
            ctx.spec, ctx.training = s, training
            ctx.shape = (N, Ci, T, To, K)
            ctx.nparams = len(params)
            ctx.save_for_backward(x, A_eff, y, u, r, out, gcn_w, gcn_b, tcn_w, res_w, bn1_w, bn2_w, bnr_w,
                                  dbn_w, m0, i0, s0, h0, m1, i1, s1, h1, m2, i2, s2, h2, mr, ir, sr, hr, sa, wb_dft)
            ctx.x_needs_grad = ctx.needs_input_grad[2]
        return out

    @staticmethod
    def _forward_inference(s, x, A_eff, gcn_w, gcn_b, tcn_w, tcn_b, res_w, res_b):
        """eval mode, nothing to differentiate: TWO launches per block - the graph conv (data_bn folded into its load)
        and the temporal conv with BN1 + ReLU on its load and BN2, the residual (its BN folded) and the final ReLU in
        its MFMA epilogue.  The packed temporal weights and every BatchNorm's (scale, shift) are cached on the block
        per parameter version (s.cache), so a second forward launches no pack / prepare kernel."""
        N, Ci, T, V = x.shape
        Co, Kt, st = s.Co, s.Kt, s.stride
        To = (T - 1) // st + 1
        K = A_eff.shape[0]
        dev = x.device
        c = s.cache
        s0 = h0 = None
        if s.data_bn is not None:
            s0, h0 = bn_eval_cached(c, "data_bn", s.data_bn)
        y = torch.empty((N, Co, T, 16), dtype=torch.float32, device=dev)
        r = torch.empty((N, Co, To, 16), dtype=torch.float32, device=dev) if res_w is not None else None
        _lib.call(s.gcn_fwd, x, s0, h0, gcn_w, gcn_b, int(s.transposed), A_eff, s.nnz, y,
                  res_w, res_b, int(s.transposed), r, N, Ci, Co, K, T, st)
        s1, h1 = bn_eval_cached(c, "bn1", s.bn1)
        mode = 1 if s.transposed else 0
        bf = s.bf16_fwd
        s2, h2 = bn_eval_cached(c, "bn2", s.bn2)
        res_t = sr = hr = None
        if s.residual == "conv":
            sr, hr = bn_eval_cached(c, "bnr", s.bnr)
            res_t = r
        elif s.residual == "identity":
            res_t = x
        if s.dft == "force" or (s.dft == "auto" and _dft_mod.preferred(Co, Kt, T, To)):
            # frequency domain: the weights' spectrum is cached per parameter version and sequence length
            plan = _dft_mod.seg_plan(Co, T, To, st, Kt, N)
            if plan is not None:
                # long sequence: overlap-save, the transforms cost S K^2 instead of T^2 (dft.seg_plan)
                wf_seg = cached(c, "wf_dft_seg", tensor_key(tcn_w) + plan + (int(s.transposed),),
                                lambda: _dft_mod.weight_spectrum_seg(tcn_w, Co, Kt, s.transposed, plan))
                return _dft_mod.conv_forward_seg(y, s1, h1, True, wf_seg, tcn_b, Kt, plan,
                                                 epilogue=(s2, h2, res_t, sr, hr, True))
            wf_dft = cached(c, "wf_dft", tensor_key(tcn_w) + (T, To, st, int(s.transposed)),
                            lambda: _dft_mod.weight_spectrum(tcn_w, Co, T, Kt, s.transposed, To, st)[0])
            out, _ = _dft_mod.conv_forward(y, s1, h1, True, wf_dft, tcn_b, To, st, Kt,
                                           epilogue=(s2, h2, res_t, sr, hr, True), transposed=s.transposed)
            return out
        wp = cached(c, "wp", tensor_key(tcn_w) + (str(bf), mode, st),
                    lambda: tcn_pack(tcn_w, Co, Co, Kt, st, mode, bf16=bf))
        return tcn_conv(y, s1, h1, True, wp, tcn_b, s2, h2, res_t, sr, hr, True,
                        N, Co, Co, T, To, Kt, st, mode, bf16=bf)

    @staticmethod
    def backward(ctx, dout):
        s, training = ctx.spec, ctx.training
        (x, A_eff, y, u, r, out, gcn_w, gcn_b, tcn_w, res_w, bn1_w, bn2_w, bnr_w, dbn_w,
         m0, i0, s0, h0, m1, i1, s1, h1, m2, i2, s2, h2, mr, ir, sr, hr, sa, wb_dft) = ctx.saved_tensors
        N, Ci, T, To, K = ctx.shape
        Co, Kt, st = s.Co, s.Kt, s.stride
        dout = _f32(dout)
        dev = dout.device

        # tail: out = relu(bn2(u) + res)
        dr = dgr = dbr = None
        if s.residual == "conv" and not (training and _sync_world() > 1):
            du, dg2, db2, dr, dgr, dbr = bn_backward_tail_pair(dout, out, u, r, m2, i2, bn2_w, mr, ir, bnr_w, training)
        elif s.residual == "conv":
            # synchronised BatchNorm: the fused pair cannot stop between its reduce and its apply for the all-reduce;
            # its two halves share ONE all-reduce
            du, dg2, db2, dr, dgr, dbr = bn_backward_tail_synced(dout, out, u, r, m2, i2, bn2_w, mr, ir, bnr_w)
        else:
            du, dg2, db2 = bn_backward(dout, u, out, None, None, 1, m2, i2, bn2_w, training, 0)

        # temporal conv
        if sa is not None:
            sdu = _dft_mod.transform_dout(du, st, T, Kt, s.transposed)
            del du
            dtw, dtb = _dft_mod.conv_wgrad(sdu, sa, T, Kt, s.transposed, tuple(tcn_w.shape), To, st)
            da = _dft_mod.conv_dgrad(sdu, wb_dft, T, Kt, st, s.transposed)
            del sdu
        else:
            dtw, dtb = tcn_wgrad(du, y, s1, h1, True, N, Co, Co, T, To, Kt, st, s.transposed, tcn_w.shape,
                                 bf16=s.bf16_wgrad)
            dmode = 3 if s.transposed else 2
            wpd = tcn_pack(tcn_w, Co, Co, Kt, st, dmode, bf16=s.bf16_dgrad)
            da = tcn_conv(du, None, None, False, wpd, None, None, None, None, None, None, False,
                          N, Co, Co, T, To, Kt, st, dmode, bf16=s.bf16_dgrad)
            del du

        # BN1 + ReLU
        dy, dg1, db1 = bn_backward(da, y, None, s1, h1, 2, m1, i1, bn1_w, training, 0)
        del da

        # graph conv (+ residual conv)
        need_dx = ctx.x_needs_grad or s.data_bn is not None
        dx = torch.empty_like(x) if need_dx else None
        dgw = torch.empty_like(gcn_w)
        dgb = torch.empty_like(gcn_b)
        dA = torch.empty_like(A_eff)
        drw = torch.empty_like(res_w) if res_w is not None else None
        drb = torch.empty(Co, dtype=torch.float32, device=dev) if res_w is not None else None
        nb = _lib.query("step_gcn_bwd_ws_bytes", N, Ci, Co, K, T)
        ws = _ws(nb, dev)
        idres = s.residual == "identity" and dx is not None        # dx += dout * [out > 0], in the same launch
        res_w_dx = res_w
        if getattr(s, "dx_skip_residual", False):
            # saliency.GuidedBackprop: the reference's hook sits on gcn.conv, so the gradient it reports leaves the
            # residual branch out - the kernels use the residual weight for dx only (dWr is dres x^T)
            idres = False
            res_w_dx = torch.zeros_like(res_w) if res_w is not None else None
        _lib.call(s.gcn_bwd, dy, dr, x, s0, h0, gcn_w, gcn_b, int(s.transposed), A_eff, s.nnz,
                  res_w_dx, int(s.transposed), dout if idres else None, out if idres else None,
                  dx, dgw, dgb, dA, drw, drb, N, Ci, Co, K, T, st, ws, nb)

        grads = []
        if s.data_bn is not None:
            # dx currently is the gradient w.r.t. the normalised input
            dx_raw, dg0, db0 = bn_backward(dx, x, None, None, None, 0, m0, i0, dbn_w, training, 1,
                                           want_dx=ctx.x_needs_grad)
            dx = dx_raw
            grads += [dg0, db0]
        grads += [dgw, dgb, dg1, db1, dtw, dtb, dg2, db2]
        if s.residual == "conv":
            grads += [drw, drb, dgr, dbr]
        return (None, None, dx if ctx.x_needs_grad else None, dA) + tuple(grads)


def block_repeated_input_eval(spec, h, A_eff, T, gcn_w, gcn_b, tcn_w, tcn_b, res_w=None, res_b=None):
    """Eval-mode st_gcn / st_gctn block whose input is ONE vector per sequence repeated over all T frames and 16
    joints (CVAE_stgcn.py:209 `z.repeat([1, 1, T, V])` feeding decoder[0]) -> (N, Co, T, 16).

    The repeated tensor is never built: graph conv, BN1 + ReLU (and a residual conv) run on a single frame, and
    the temporal conv of that frame-independent activation is one Co x Co matrix per output frame (the taps that
    fall inside the sequence there, step_tcn_const_weights / step_tcn_const_fwd) with BN2, the residual and the
    final ReLU in its epilogue."""
    s = spec
    h = _f32(h)
    A_eff = _f32(A_eff)
    N, Ci = h.shape
    if Ci != s.Ci or s.stride != 1 or s.data_bn is not None:
        raise _lib.StepHipError("repeated-input block: needs (N,%d) input, stride 1, no data_bn" % s.Ci)
    Co, Kt, K = s.Co, s.Kt, A_eff.shape[0]
    dev = h.device
    x1 = h[:, :, None, None].expand(N, Ci, 1, 16).contiguous()            # one frame
    y0 = torch.empty((N, Co, 1, 16), dtype=torch.float32, device=dev)
    r0 = torch.empty((N, Co, 1, 16), dtype=torch.float32, device=dev) if s.residual == "conv" else None
    gcn_w, gcn_b, tcn_w, tcn_b = _f32(gcn_w), _f32(gcn_b), _f32(tcn_w), _f32(tcn_b)
    if s.residual == "conv":
        res_w, res_b = _f32(res_w), _f32(res_b)
    else:
        res_w = res_b = None
    _lib.call("step_gcn_fwd", x1, None, None, gcn_w, gcn_b, int(s.transposed), A_eff, s.nnz, y0,
              res_w, res_b, int(s.transposed), r0, N, Ci, Co, K, 1, 1)
    c = s.cache
    s1, h1 = bn_eval_cached(c, "bn1", s.bn1)
    a0 = affine_act(y0, s1, h1, None, None, None, True, 0)

    def build_wcum():
        wc = torch.empty((T, Co, Co), dtype=torch.float32, device=dev)
        _lib.call("step_tcn_const_weights", tcn_w, wc, Co, Co, Kt, T, int(s.transposed))
        return wc
    wcum = cached(c, "wcum", tensor_key(tcn_w) + (T,), build_wcum)
    s2, h2 = bn_eval_cached(c, "bn2", s.bn2)
    res0 = sr = hr = None
    if s.residual == "conv":
        sr, hr = bn_eval_cached(c, "bnr", s.bnr)
        res0 = r0
    elif s.residual == "identity":
        res0 = x1
    out = torch.empty((N, Co, T, 16), dtype=torch.float32, device=dev)
    _lib.call("step_tcn_const_fwd", a0, wcum, tcn_b, s2, h2, res0, sr, hr, 1, out, N, Co, Co, T)
    return out


# ---------------------------------------------------------------------------- heads
class PoolFn(torch.autograd.Function):
    """F.avg_pool2d(x, x.size()[2:]) (+ mean over M == 1), classifier.py:81-82 -> (N, C)"""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        N, C = x.shape[0], x.shape[1]
        inner = x.numel() // (N * C)
        feat = torch.empty((N, C), dtype=torch.float32, device=x.device)
        _lib.call("step_pool_fwd", x, feat, N, C, inner)
        ctx.shape = tuple(x.shape)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        N, C = ctx.shape[0], ctx.shape[1]
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dfeat.device)
        _lib.call("step_pool_bwd", _f32(dfeat), dx, N, C, dx.numel() // (N * C))
        return dx


def head_inference(h, weight, bias, softmax):
    """(out, feat) of avg_pool2d -> fcn (-> softmax) for a no-grad eval forward, one launch (step_head_fwd)"""
    h = _f32(h)
    N, C = h.shape[0], h.shape[1]
    O = weight.shape[0]
    feat = torch.empty((N, C), dtype=torch.float32, device=h.device)
    out = torch.empty((N, O), dtype=torch.float32, device=h.device)
    _lib.call("step_head_fwd", h, _f32(weight), bias, 0, int(softmax), feat, out, N, C, h.numel() // (N * C), O)
    return out, feat


class DenseFn(torch.autograd.Function):
    """y = x W^T + b for the 1x1 convs on pooled features (weight (O,I,1,1), or (I,O,1,1) for a
    ConvTranspose2d when ``transposed``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, transposed):
        x, weight = _f32(x), _f32(weight)
        N, I = x.shape
        O = weight.shape[1] if transposed else weight.shape[0]
        y = torch.empty((N, O), dtype=torch.float32, device=x.device)
        _lib.call("step_dense_fwd", x, weight, bias, int(transposed), y, N, I, O)
        ctx.save_for_backward(x, weight)
        ctx.transposed, ctx.has_bias = transposed, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32(dy)
        N, I = x.shape
        O = dy.shape[1]
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight)
        db = torch.empty(O, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        _lib.call("step_dense_bwd", dy, x, weight, int(ctx.transposed), dx, dw, db, N, I, O)
        return dx, dw, db, None


class SoftmaxFn(torch.autograd.Function):
    """F.softmax(x, dim=1), classifier.py:87"""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        _lib.call("step_softmax_fwd", x, y, x.shape[0], x.shape[1])
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        _lib.call("step_softmax_bwd", _f32(dy), y, dx, y.shape[0], y.shape[1])
        return dx


class DataBnFn(torch.autograd.Function):
    """A stand-alone ``data_bn`` (BatchNorm1d on the (N, V*C, T) view): the Decoder's last step,
    CVAE_stgcn.py:224-229."""

    @staticmethod
    def forward(ctx, bn, training, x, weight, bias):
        x = _f32(x)
        mean, invstd, scale, shift = bn_prepare(x, 1, bn, training)
        out = affine_act(x, scale, shift, None, None, None, False, 1)
        ctx.training = training
        ctx.save_for_backward(x, mean, invstd, weight)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, weight = ctx.saved_tensors
        dx, dg, db = bn_backward(_f32(dy), x, None, None, None, 0, mean, invstd, weight, ctx.training, 1)
        return None, None, dx, dg, db


class BnRowsFn(torch.autograd.Function):
    """nn.BatchNorm1d on (N, G): data_bn2 of the hybrid head, classifier_hybrid/net/classifier.py:83"""

    @staticmethod
    def forward(ctx, bn, training, x, weight, bias):
        x = _f32(x)
        N, G = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(G, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        ctx.synced = bool(training and _sync_world() > 1)
        if ctx.synced:
            # synchronised BatchNorm: [SUM x, SUM x^2, row count] over all ranks in one fp64 all-reduce
            sums = torch.empty(2 * G + 1, dtype=torch.float64, device=x.device)
            _lib.call("step_bn_rows_stats", x, sums, N, G)
            sums[2 * G] = float(N)
            _sync_sums(sums)
            _lib.call("step_bn_rows_apply", x, sums, weight, bias, bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS,
                      y, mean, invstd, N, G)
        else:
            _lib.call("step_bn_rows_fwd", x, weight, bias, bn.running_mean, bn.running_var, int(training),
                      BN_MOMENTUM, BN_EPS, y, mean, invstd, N, G)
        if training:
            bn.num_batches_tracked += 1
        ctx.training = training
        ctx.save_for_backward(x, mean, invstd, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, weight = ctx.saved_tensors
        N, G = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty(G, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        if ctx.synced:
            # dgamma / dbeta from this rank's sums (averaged with the other gradients later); dx from the combined ones
            dy = _f32(dy)
            sums = torch.empty(2 * G + 1, dtype=torch.float64, device=x.device)
            _lib.call("step_bn_rows_bwd_sums", dy, x, mean, invstd, sums, dg, db, N, G)
            sums[2 * G] = float(N)
            _sync_sums(sums)
            _lib.call("step_bn_rows_bwd_apply", dy, x, mean, invstd, weight, sums, dx, N, G)
            return None, None, dx, dg, db
        _lib.call("step_bn_rows_bwd", _f32(dy), x, mean, invstd, weight, int(ctx.training), dx, dg, db, N, G)
        return None, None, dx, dg, db
