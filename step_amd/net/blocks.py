"""st_gcn / st_gctn blocks with the reference's module surface and state_dict keys
(classifier.py:116-191, CVAE_stgcn.py:235-394).  The ``nn.BatchNorm2d`` / ``nn.Conv2d``
children are parameter containers only; forward is one fused autograd Function
(``ops.StGcnBlockFn``) over the HIP kernels.
"""
import torch.nn as nn

from ..ops import BlockSpec, StGcnBlockFn, block_repeated_input_eval
from .utils.tgcn import ConvTemporalGraphical, ConvTransposeTemporalGraphical


class _Zero:
    """`lambda x: 0` of the reference (classifier.py:168), picklable"""

    def __call__(self, x):
        return 0


class _Identity:
    def __call__(self, x):
        return x


class _StBlock(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dropout=0, residual=True):
        super().__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0] % 2 == 1
        if dropout != 0:
            raise NotImplementedError("the reference always uses dropout=0 (classifier.py:142)")
        padding = ((kernel_size[0] - 1) // 2, 0)
        conv_cls = nn.ConvTranspose2d if self.transposed else nn.Conv2d
        gcn_cls = ConvTransposeTemporalGraphical if self.transposed else ConvTemporalGraphical
        setattr(self, "gctn" if self.transposed else "gcn",
                gcn_cls(in_channels, out_channels, kernel_size[1]))
        self.tcn = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            conv_cls(out_channels, out_channels, (kernel_size[0], 1), (stride, 1), padding),
            nn.BatchNorm2d(out_channels),
            nn.Dropout(dropout, inplace=True),
        )
        if not residual:
            self.residual = _Zero()
            self._res_kind = "none"
        elif in_channels == out_channels and stride == 1:
            self.residual = _Identity()
            self._res_kind = "identity"
        else:
            self.residual = nn.Sequential(
                conv_cls(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                nn.BatchNorm2d(out_channels),
            )
            self._res_kind = "conv"
        self.relu = nn.ReLU(inplace=True)
        self._geom = (in_channels, out_channels, kernel_size[0], stride)
        self.a_nnz_max = 16
        self.mfma_dtype = "f32"             # "bf16": step_amd.set_mfma_dtype(model, "bf16")
        self._infer_cache = {}              # packed temporal weights, eval-mode BN coefficients (ops.cached)

    def invalidate_caches(self):
        """Drop the inference caches.  They follow in-place parameter updates by themselves (tensor versions); call this
        after writing parameters through `.data`, which bumps no version."""
        self._infer_cache.clear()

    def train(self, mode=True):
        self._infer_cache.clear()           # (weights_init writes through .data before training starts)
        return super().train(mode)

    @property
    def graph_conv(self):
        return self.gctn if self.transposed else self.gcn

    def _refuse_child_hooks(self):
        """The children are parameter containers: their forward never runs, so a hook on them would silently never
        fire (GuidedBackprop of utils/processor.py:35,62-66 registers such hooks).  Fail loudly instead."""
        for name, m in self.named_modules():
            if m is self:
                continue
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
                from .. import _lib
                raise _lib.StepHipError(
                    "a hook is registered on %s.%s, but a step_amd block runs as ONE fused autograd Function and never "
                    "calls its children: hook the block itself (INTEGRATION.md, 'Hooks on the container children')"
                    % (type(self).__name__, name))

    def forward(self, x, A, data_bn=None):
        self._refuse_child_hooks()
        ci, co, kt, stride = self._geom
        bnr = self.residual[1] if self._res_kind == "conv" else None
        spec = BlockSpec(ci, co, kt, stride, self._res_kind, self.transposed, self.a_nnz_max,
                         self.tcn[0], self.tcn[3], bnr, data_bn,
                         bf16={"bf16": True, "f32_split": "split", "f32_split_wgrad": "split_wgrad",
                               "f32_split_gcn": "split_gcn", "f32_dft": "dft", "f32": "auto_dft"}.get(self.mfma_dtype, False))
        spec.cache = self._infer_cache
        spec.dx_skip_residual = getattr(self, "dx_skip_residual", False)     # set by saliency.GuidedBackprop only
        g = self.graph_conv.conv
        params = []
        if data_bn is not None:
            params += [data_bn.weight, data_bn.bias]
        params += [g.weight, g.bias, self.tcn[0].weight, self.tcn[0].bias, self.tcn[2].weight,
                   self.tcn[2].bias, self.tcn[3].weight, self.tcn[3].bias]
        if bnr is not None:
            params += [self.residual[0].weight, self.residual[0].bias, bnr.weight, bnr.bias]
        out = StGcnBlockFn.apply(spec, self.training, x, A, *params)
        return out, A

    def forward_repeated(self, h, A, T):
        """Eval-mode forward of `x = h[:, :, None, None].repeat(1, 1, T, 16)` without building x
        (the decoder's first block, CVAE_stgcn.py:209-212); no autograd graph is recorded."""
        self._refuse_child_hooks()
        ci, co, kt, stride = self._geom
        bnr = self.residual[1] if self._res_kind == "conv" else None
        spec = BlockSpec(ci, co, kt, stride, self._res_kind, self.transposed, self.a_nnz_max,
                         self.tcn[0], self.tcn[3], bnr, None)
        spec.cache = self._infer_cache
        g = self.graph_conv.conv
        rw = rb = None
        if bnr is not None:
            rw, rb = self.residual[0].weight, self.residual[0].bias
        out = block_repeated_input_eval(spec, h.detach(), A.detach(), T, g.weight.detach(), g.bias.detach(),
                                        self.tcn[2].weight.detach(), self.tcn[2].bias.detach(),
                                        None if rw is None else rw.detach(), None if rb is None else rb.detach())
        return out, A


class st_gcn(_StBlock):
    """classifier.py:116-191"""
    transposed = False


class st_gctn(_StBlock):
    """CVAE_stgcn.py:316-394"""
    transposed = True
