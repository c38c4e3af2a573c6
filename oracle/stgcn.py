"""Oracle (test infrastructure): functional CPU restatement of STEP's ST-GCN stack.

Everything is a pure function of a parameter dict ``P`` whose keys are exactly
the reference's ``state_dict`` keys, so a shipped checkpoint or a golden
fixture plugs in directly.  Written with elementary torch ops (einsum, matmul,
unfold-free explicit temporal correlation through ``F.conv2d``); no
``nn.Module`` is used.  Works in float32 or float64 and is differentiable by
autograd, which is how gradient parity of the HIP backward kernels is checked.

Reference anchors (all under /root/reference):
  ConvTemporalGraphical.forward          classifier_stgcn_real_only/net/utils/tgcn.py:58-67
  ConvTransposeTemporalGraphical.forward classifier_stgcn_real_only/net/utils/tgcn.py:121-130
  st_gcn.__init__/forward                classifier_stgcn_real_only/net/classifier.py:137-191
  st_gctn.__init__/forward               generator_cvae/net/CVAE_stgcn.py:340-394
  Classifier.forward (real_only)         classifier_stgcn_real_only/net/classifier.py:65-89
  Classifier.forward (real_and_synth)    classifier_stgcn_real_and_synth/net/classifier.py:65-89
  Classifier.forward (hybrid)            classifier_hybrid/net/classifier.py:61-89
  Classifier.extract_feature             classifier_stgcn_real_only/net/classifier.py:91-113
  Encoder/Decoder/CVAE.forward           generator_cvae/net/CVAE_stgcn.py:25-37,108-136,198-232
  weights_init                           classifier_stgcn_real_only/utils/processor.py:117-129
  train step (CE loss + Adam)            classifier_stgcn_real_only/utils/processor.py:316-328
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# oneDNN's CPU convolution BACKWARD on this image is not trustworthy for the (Kt,1)
# kernels of this network: it never returns when T < Kt and returns ~1e20 garbage
# weight gradients for the stride-2, Kt=75, T=75 case (seen while pinning this oracle
# against the reference, which was itself run with oneDNN off).  The oracle therefore
# runs on PyTorch's native CPU kernels.  ``set_onednn(True)`` may be used for
# forward-only timing, whose results were checked against the golden vectors.
torch.backends.mkldnn.enabled = False


def set_onednn(enabled):
    torch.backends.mkldnn.enabled = bool(enabled)

# (in, out, stride, residual) per block; residual True means "identity if
# in==out and stride==1 else 1x1 conv + BN" (classifier.py:167-181).
ARCH = {
    # classifier_stgcn_real_only/net/classifier.py:40-51
    "real_only": ((None, 32, 1, True), (32, 64, 2, True), (64, 64, 1, True)),
    # classifier_stgcn_real_and_synth/net/classifier.py:40-51
    "real_and_synth": ((None, 32, 1, False), (32, 64, 2, True), (64, 64, 2, True)),
    # classifier_hybrid/net/classifier.py:40-44
    "hybrid": ((None, 32, 1, False), (32, 64, 2, True), (64, 64, 2, True)),
    # generator_cvae/net/CVAE_stgcn.py:82-93
    "cvae_encoder": ((None, 64, 1, True), (64, 32, 1, True), (32, 32, 1, True)),
    # generator_cvae/net/CVAE_stgcn.py:173-184 (out of the last block = in_channels)
    "cvae_decoder": ((32, 32, 1, True), (32, 64, 1, True), (64, None, 1, True)),
}


# --------------------------------------------------------------------------- BN
def batch_norm(x, P, prefix, training, dims, shape, update_running=True):
    """nn.BatchNorm{1,2}d semantics.  ``dims`` are the reduced axes and
    ``shape`` broadcasts a per-channel vector onto ``x``."""
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if training:
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if update_running:
            with torch.no_grad():
                n = x.numel() // mean.numel()
                rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(rm.dtype))
                unb = var.detach() * (n / max(n - 1, 1))
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb.to(rv.dtype))
                key = prefix + ".num_batches_tracked"
                if key in P:
                    P[key] += 1
    else:
        mean = P[prefix + ".running_mean"].to(x.dtype)
        var = P[prefix + ".running_var"].to(x.dtype)
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(shape)) * (inv * w).view(shape) + b.view(shape)


def bn2d(x, P, prefix, training, update_running=True):
    return batch_norm(x, P, prefix, training, (0, 2, 3), (1, -1, 1, 1), update_running)


def data_bn(x, P, prefix, training, update_running=True):
    """classifier.py:68-74: BatchNorm1d over (N*M, V*C, T); channel = v*C + c.
    ``x`` is (N, C, T, V) (M == 1 folded into N)."""
    n, c, t, v = x.shape
    xp = x.permute(0, 3, 1, 2).reshape(n, v * c, t)
    xp = batch_norm(xp, P, prefix, training, (0, 2), (1, -1, 1), update_running)
    return xp.reshape(n, v, c, t).permute(0, 2, 3, 1)


# ----------------------------------------------------------------- graph conv
def conv_temporal_graphical(x, A, weight, bias, transposed=False):
    """tgcn.py:58-67 (and :121-130 when ``transposed``).
    x (N,Ci,T,V), A (K,V,V), weight (K*Co,Ci,1,1) or (Ci,K*Co,1,1) if transposed."""
    k = A.shape[0]
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    if transposed:
        w2 = w2.t()
    kc = w2.shape[0]
    h = torch.einsum("oc,nctv->notv", w2, x)
    if bias is not None:
        h = h + bias.view(1, -1, 1, 1)
    n, _, t, v = h.shape
    h = h.view(n, k, kc // k, t, v)
    return torch.einsum("nkctv,kvw->nctw", h, A)


def temporal_conv(a, weight, bias, stride, transposed=False):
    """tcn[2]: Conv2d(C,C,(Kt,1),(s,1),((Kt-1)//2,0)) classifier.py:156-162, or
    the ConvTranspose2d twin CVAE_stgcn.py:359-365 (which, at stride 1, is the
    correlation with the kernel reversed in time and in/out channels swapped).

    When the sequence is shorter than the kernel, oneDNN's convolution backward
    never returns on this image, so that case is computed as an explicit
    window x weight contraction instead of through ``F.conv2d``."""
    kt = weight.shape[2]
    pad = (kt - 1) // 2
    if a.shape[2] >= kt:
        if transposed:
            return F.conv_transpose2d(a, weight, bias, stride=(stride, 1), padding=(pad, 0))
        return F.conv2d(a, weight, bias, stride=(stride, 1), padding=(pad, 0))
    w3 = weight[..., 0]
    if transposed:
        assert stride == 1, "st_gctn is only ever built with stride 1"
        w3 = w3.permute(1, 0, 2).flip(2)
    win = F.pad(a, (0, 0, pad, pad)).unfold(2, kt, stride)       # (N, C, To, V, Kt)
    out = torch.einsum("nctvk,ock->notv", win, w3)
    return out if bias is None else out + bias.view(1, -1, 1, 1)


def _rb(t):
    """round to bf16 and back (round-to-nearest-even, what v_cvt_pk_bf16_f32 does)"""
    return t.to(torch.bfloat16).to(t.dtype)


class _TemporalConvBf16Operands(torch.autograd.Function):
    """temporal_conv with its MFMA operands rounded to bf16 exactly where step_amd's bf16 kernels round
    them (tcn_bf16.hip, tcn_wgrad_sw.hip BF16 mode; BASELINE config "real_and_synth training, bf16"):
      forward : conv(bf16(a), bf16(w)) + b                  a = relu(bn1(y)) rounded on its way into LDS
      dgrad   : conv_input_grad(bf16(w), bf16(dout))
      wgrad   : conv_weight_grad(bf16(a), bf16(dout));  db = SUM dout (exact fp32)
    Accumulation stays fp32.  This is what the model-level bf16 parity tests are held against (1e-3),
    beside north_star's 1e-2 against the plain fp32 path."""

    @staticmethod
    def forward(ctx, a, weight, bias, stride, transposed=False):
        ctx.save_for_backward(a, weight)
        ctx.stride, ctx.transposed = stride, transposed
        return temporal_conv(_rb(a), _rb(weight), bias, stride, transposed)

    @staticmethod
    def backward(ctx, dout):
        a, weight = ctx.saved_tensors
        ar = _rb(a).detach().requires_grad_(True)
        wr = _rb(weight).detach().requires_grad_(True)
        with torch.enable_grad():
            out = temporal_conv(ar, wr, None, ctx.stride, ctx.transposed)
        da, dw = torch.autograd.grad(out, (ar, wr), _rb(dout))
        return da, dw, dout.sum(dim=(0, 2, 3)), None, None


def st_gcn_block(x, A_eff, P, prefix, stride, residual, training,
                 transposed=False, update_running=True, return_parts=False, tcn_bf16=False, x_res=None):
    """st_gcn.forward classifier.py:185-191 / st_gctn.forward CVAE_stgcn.py:388-394.
    ``x_res``: a separate tensor (same values) for the residual branch, so that a gradient can be taken through the
    graph-conv branch alone - what a backward hook on gcn.conv sees (processor.py:31-36)."""
    xr = x if x_res is None else x_res
    gname = "gctn" if transposed else "gcn"
    has_res_conv = (prefix + ".residual.0.weight") in P
    if not residual:
        res = None
    elif has_res_conv:
        rw = P[prefix + ".residual.0.weight"]
        rb = P[prefix + ".residual.0.bias"]
        if transposed:
            res = F.conv_transpose2d(xr, rw, rb, stride=(stride, 1))
        else:
            res = F.conv2d(xr, rw, rb, stride=(stride, 1))
        res = bn2d(res, P, prefix + ".residual.1", training, update_running)
    else:
        res = xr
    y = conv_temporal_graphical(x, A_eff, P[prefix + "." + gname + ".conv.weight"],
                                P[prefix + "." + gname + ".conv.bias"], transposed)
    a = torch.relu(bn2d(y, P, prefix + ".tcn.0", training, update_running))
    # tcn_bf16: True, or a predicate on the conv's channel count (the kernels cover square 32- / 64-channel convs;
    # the CVAE decoder's last block, 3 -> 3, stays on fp32 operands)
    tw = P[prefix + ".tcn.2.weight"]
    if tcn_bf16(tw.shape[0]) if callable(tcn_bf16) else tcn_bf16:
        u = _TemporalConvBf16Operands.apply(a, P[prefix + ".tcn.2.weight"], P[prefix + ".tcn.2.bias"], stride,
                                            transposed)
    else:
        u = temporal_conv(a, P[prefix + ".tcn.2.weight"], P[prefix + ".tcn.2.bias"],
                          stride, transposed)
    v = bn2d(u, P, prefix + ".tcn.3", training, update_running)   # Dropout(p=0) is identity
    out = torch.relu(v if res is None else v + res)
    if return_parts:
        return out, dict(y=y, a=a, u=u, res=res)
    return out


def _stack(x, P, arch, net_prefix, imp_prefix, A, training, transposed=False,
           update_running=True, tcn_bf16=False):
    for i, (_, _, stride, residual) in enumerate(ARCH[arch]):
        key = "%s.%d" % (imp_prefix, i) if imp_prefix else None
        A_eff = A * P[key] if key is not None and key in P else A
        x = st_gcn_block(x, A_eff, P, "%s.%d" % (net_prefix, i), stride, residual,
                         training, transposed, update_running, tcn_bf16=tcn_bf16)
    return x


# ----------------------------------------------------------------- classifiers
def classifier_forward(P, x, variant="real_only", training=False, x_aff=None,
                       update_running=True, tcn_bf16=False):
    """x (N,C,T,V,M) with M == 1 (loader.py:65).  Returns (out, feature) for the
    two ST-GCN classifiers, logits only for the hybrid one."""
    n, c, t, v, m = x.shape
    assert m == 1, "the reference only ever feeds M == 1"
    x4 = x[..., 0]
    bn_name = "data_bn1" if variant == "hybrid" else "data_bn"
    h = data_bn(x4, P, bn_name, training, update_running)
    h = _stack(h, P, variant, "st_gcn_networks", "edge_importance", P["A"].to(x.dtype),
               training, False, update_running, tcn_bf16=tcn_bf16)
    pooled = h.mean(dim=(2, 3))                                  # avg_pool2d + mean over M
    if variant == "hybrid":
        z = torch.cat((pooled, x_aff), dim=1)
        z = batch_norm(z, P, "data_bn2", training, (0,), (1, -1), update_running)
        for i in (0, 1):                                         # no nonlinearity in between
            w = P["combined_networks.%d.weight" % i]
            z = z @ w.reshape(w.shape[0], -1).t() + P["combined_networks.%d.bias" % i]
        return z
    w = P["fcn.weight"]
    logits = pooled @ w.reshape(w.shape[0], -1).t() + P["fcn.bias"]
    feat = pooled.squeeze()                                      # drops N when N == 1
    if variant == "real_only":
        return torch.softmax(logits, dim=1), feat
    return logits, feat


def classifier_extract_feature(P, x, variant="real_only"):
    """classifier.py:91-113 (eval semantics of whatever mode; here eval)."""
    n, c, t, v, m = x.shape
    x4 = x[..., 0]
    h = data_bn(x4, P, "data_bn", False)
    h = _stack(h, P, variant, "st_gcn_networks", "edge_importance", P["A"].to(x.dtype), False)
    w = P["fcn.weight"]
    out = torch.einsum("oc,nctv->notv", w.reshape(w.shape[0], -1), h) + P["fcn.bias"].view(1, -1, 1, 1)
    return out.unsqueeze(-1), h.unsqueeze(-1)


# ------------------------------------------------------------------------ CVAE
def encoder_forward(P, x, l, training=False, prefix="encoder", update_running=True, tcn_bf16=False):
    """Encoder.forward CVAE_stgcn.py:108-136.  x (N,3,T,V,1), l (N,cls,T,V,1)."""
    h = torch.cat((x, l), dim=1)[..., 0]
    h = data_bn(h, P, prefix + ".data_bn", training, update_running)
    h = _stack(h, P, "cvae_encoder", prefix + ".encoder", prefix + ".edge_importance",
               P[prefix + ".A"].to(x.dtype), training, False, update_running, tcn_bf16=tcn_bf16)
    pooled = h.mean(dim=(2, 3))
    wm, wl = P[prefix + ".z_mean.weight"], P[prefix + ".z_lsig.weight"]
    mean = pooled @ wm.reshape(wm.shape[0], -1).t() + P[prefix + ".z_mean.bias"]
    lsig = pooled @ wl.reshape(wl.shape[0], -1).t() + P[prefix + ".z_lsig.bias"]
    return mean, lsig


def decoder_forward(P, z, l, T, V, training=False, prefix="decoder", update_running=True, tcn_bf16=False):
    """Decoder.forward CVAE_stgcn.py:198-232.  Returns (N,C,T,V,1)."""
    zl = torch.cat((z, l), dim=1)
    w = P[prefix + ".fcn.weight"]                  # ConvTranspose2d weight (in, out, 1, 1)
    h = zl @ w.reshape(w.shape[0], w.shape[1]) + P[prefix + ".fcn.bias"]
    h = h[:, :, None, None].expand(-1, -1, T, V)
    h = _stack(h, P, "cvae_decoder", prefix + ".decoder", prefix + ".edge_importance",
               P[prefix + ".A"].to(z.dtype), training, True, update_running, tcn_bf16=tcn_bf16)
    h = data_bn(h, P, prefix + ".data_bn", training, update_running)
    return h.unsqueeze(-1)


def cvae_forward(P, x, lenc, ldec, eps, T, V, training=False, update_running=True, tcn_bf16=False):
    """CVAE.forward CVAE_stgcn.py:25-37 with the reparameterisation noise given."""
    mean, lsig = encoder_forward(P, x, lenc, training, "encoder", update_running, tcn_bf16=tcn_bf16)
    z = eps * torch.exp(0.5 * lsig) + mean
    recon = decoder_forward(P, z, ldec, T, V, training, "decoder", update_running, tcn_bf16=tcn_bf16)
    return recon, mean, lsig, z


# ------------------------------------------------------------- init / training
def weights_init_(P, generator=None):
    """processor.py:117-129 applied to a parameter dict: every Conv2d weight
    N(0, 0.02) and bias 0, every BatchNorm weight N(1, 0.02) and bias 0;
    ConvTranspose2d is NOT matched by the reference's class-name test."""
    for k in sorted(P):
        t = P[k]
        leaf = k.rsplit(".", 1)[-1]
        is_bn = (k.rsplit(".", 1)[0] + ".running_mean") in P
        if is_bn:
            if leaf == "weight":
                t.normal_(1.0, 0.02, generator=generator)
            elif leaf == "bias":
                t.zero_()
        elif leaf == "weight" and t.dim() == 4 and ".gctn." not in k \
                and not k.startswith("decoder."):
            t.normal_(0.0, 0.02, generator=generator)
            P[k[:-6] + "bias"].zero_()
    return P


def adam_step(params, grads, state, lr, weight_decay, step, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (L2 weight decay folded into the gradient), one step."""
    b1, b2 = betas
    for k in params:
        g = grads[k] + weight_decay * params[k]
        m = state.setdefault(k + ".m", torch.zeros_like(g))
        v = state.setdefault(k + ".v", torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        params[k].addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


def trainable_keys(P):
    return [k for k in P if k != "A" and not k.endswith(".A") and
            not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def classifier_loss_and_grads(P, x, labels, variant="real_only", x_aff=None, tcn_bf16=False):
    """One training-mode forward + backward: processor.py:322-327 (CrossEntropyLoss
    on the model output, which for real_only is already a softmax)."""
    keys = trainable_keys(P)
    leaves = {k: P[k].detach().clone().requires_grad_(True) for k in keys}
    Q = dict(P)
    Q.update(leaves)
    out = classifier_forward(Q, x, variant, training=True, x_aff=x_aff, tcn_bf16=tcn_bf16)
    out = out if variant == "hybrid" else out[0]
    loss = F.cross_entropy(out, labels)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys])
    return loss.detach(), out.detach(), dict(zip(keys, grads))
