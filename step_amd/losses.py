"""CVAE training loss on the GPU with the reference's names and argument order:
``between_frame_loss(gait1, gait2, thres)`` (generator_cvae/utils/losses.py:7-42) and
``vae_loss(x_in, x_out, mean, lsig, beta)`` (generator_cvae/utils/processor_stgcn.py:34-40).
One HIP launch computes the value and every gradient (step_vae_loss); the reference spends
O(B*T*V*C) scalar mse_loss calls (~16 s per iteration in its logs) on the same number."""
import torch

from . import _lib


class _VaeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_in, x_out, mean, lsig, beta):
        if x_in.dim() == 5:
            if x_in.shape[4] != 1:
                raise _lib.StepHipError("vae_loss expects M == 1 gaits (N,C,T,V,1)")
        elif x_in.dim() != 4:
            raise _lib.StepHipError("vae_loss expects (N,C,T,V[,1]) gaits, got %s" % (tuple(x_in.shape),))
        if x_in.shape != x_out.shape:
            raise _lib.StepHipError("x_in %s and x_out %s differ in shape" % (tuple(x_in.shape), tuple(x_out.shape)))
        N, C, T, V = x_in.shape[:4]
        xi = x_in.detach().float().contiguous()
        xo = x_out.detach().float().contiguous()
        dev = xo.device
        have_z = mean is not None
        Z = mean.shape[1] if have_z else 0
        m = mean.detach().float().contiguous() if have_z else None
        s = lsig.detach().float().contiguous() if have_z else None
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        d_xo = torch.empty_like(xo)
        d_m = torch.empty_like(m) if have_z else None
        d_s = torch.empty_like(s) if have_z else None
        nb = _lib.query("step_vae_loss_ws_bytes", N)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        _lib.call("step_vae_loss", xi, xo, m, s, float(beta), loss, d_xo, d_m, d_s, N, C, T, V, Z, ws, nb)
        ctx.save_for_backward(d_xo, d_m, d_s)
        ctx.shape = tuple(x_out.shape)
        ctx.parts = loss
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        d_xo, d_m, d_s = ctx.saved_tensors
        need = ctx.needs_input_grad
        gxo = (d_xo * g).view(ctx.shape) if need[1] else None
        gxi = (-(d_xo * g)).view(ctx.shape) if need[0] else None
        gm = d_m * g if (need[2] and d_m is not None) else None
        gs = d_s * g if (need[3] and d_s is not None) else None
        return gxi, gxo, gm, gs, None


def between_frame_loss(gait1, gait2, thres=0.01):
    """losses.py:7-42 (``thres`` is unused in the reference as well)"""
    return _VaeLossFn.apply(gait1, gait2, None, None, 0.0)


def vae_loss(x_in, x_out, mean, lsig, beta=1.):
    """processor_stgcn.py:34-40: between_frame_loss(x_in, x_out) + beta * KLD"""
    return _VaeLossFn.apply(x_in, x_out, mean, lsig, beta)
