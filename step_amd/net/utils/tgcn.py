"""Graph-convolution units with the reference's module surface (net/utils/tgcn.py:7-130).

``ConvTemporalGraphical(in, out, K).forward(x, A) -> (y, A)``; ``self.conv`` is a plain
``nn.Conv2d`` / ``nn.ConvTranspose2d`` used ONLY as the parameter container (so checkpoints
and ``weights_init`` see the same names and class names); the arithmetic is the fused HIP
kernel ``step_gcn_fwd`` / ``step_gcn_bwd``.
"""
import torch
import torch.nn as nn

from ... import _lib
from ...ops import _f32, _ws


class GcnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, A, weight, bias, transposed, nnz):
        x, A, weight = _f32(x), _f32(A), _f32(weight)
        N, Ci, T, V = x.shape
        K = A.shape[0]
        Co = (weight.shape[1] if transposed else weight.shape[0]) // K
        if V != 16:
            raise _lib.StepHipError("V must be 16, got %d" % V)
        y = torch.empty((N, Co, T, 16), dtype=torch.float32, device=x.device)
        _lib.call("step_gcn_fwd", x, None, None, weight, bias, int(transposed), A, nnz, y,
                  None, None, 0, None, N, Ci, Co, K, T, 1)
        ctx.save_for_backward(x, A, weight, bias)
        ctx.meta = (N, Ci, Co, K, T, transposed, nnz)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, A, weight, bias = ctx.saved_tensors
        N, Ci, Co, K, T, transposed, nnz = ctx.meta
        dy = _f32(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight)
        db = torch.empty_like(bias) if bias is not None else None
        dA = torch.empty_like(A)
        nb = _lib.query("step_gcn_bwd_ws_bytes", N, Ci, Co, K, T)
        ws = _ws(nb, x.device)
        _lib.call("step_gcn_bwd", dy, None, x, None, None, weight, bias, int(transposed), A, nnz,
                  None, 0, dx, dw, db, dA, None, None, N, Ci, Co, K, T, 1, ws, nb)
        return dx, dA, dw, db, None, None


class _GraphConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1,
                 t_padding=0, t_dilation=1, bias=True):
        super().__init__()
        if (t_kernel_size, t_stride, t_padding, t_dilation) != (1, 1, 0, 1):
            raise NotImplementedError("the reference only builds the 1x1 form (tgcn.py:41-44 defaults)")
        self.kernel_size = kernel_size
        conv_cls = nn.ConvTranspose2d if self.transposed else nn.Conv2d
        self.conv = conv_cls(in_channels, out_channels * kernel_size, kernel_size=(1, 1), bias=bias)
        # dense adjacency is always valid; models narrow this to the graph's sparsity
        self.a_nnz_max = 16

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size
        y = GcnFn.apply(x, A, self.conv.weight, self.conv.bias, self.transposed, self.a_nnz_max)
        return y, A


class ConvTemporalGraphical(_GraphConvBase):
    """tgcn.py:7-67"""
    transposed = False


class ConvTransposeTemporalGraphical(_GraphConvBase):
    """tgcn.py:70-130"""
    transposed = True
