"""Classifier of classifier_stgcn_real_only (net/classifier.py:9-113): same constructor,
attributes, state_dict keys and return values; blocks 3->32 (s1), 32->64 (s2), 64->64 (s1),
softmax output."""
import torch
import torch.nn as nn

from .. import _lib
from ..ops import PoolFn, DenseFn, SoftmaxFn, a_nnz_max
from .blocks import st_gcn
from .utils.graph import Graph
from .utils.tgcn import ConvTemporalGraphical  # noqa: F401  (reference re-exports it)


class _ClassifierBase(nn.Module):
    # (out_channels, stride, extra kwargs) per block; in_channels chains
    BLOCKS = ((32, 1, {}), (64, 2, {}), (64, 1, {}))
    DATA_BN = "data_bn"

    def _build_stack(self, in_channels, graph_args, temporal_kernel_size, edge_importance_weighting,
                     **kwargs):
        self.graph = Graph(**graph_args)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer("A", A)
        kernel_size = (temporal_kernel_size, A.size(0))
        setattr(self, self.DATA_BN, nn.BatchNorm1d(in_channels * A.size(1)))
        blocks, cin = [], in_channels
        for cout, stride, extra in self.BLOCKS:
            kw = dict(kwargs)
            kw.update(extra)
            blocks.append(st_gcn(cin, cout, kernel_size, stride, **kw))
            cin = cout
        self.st_gcn_networks = nn.ModuleList(blocks)
        self._sync_nnz()
        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList(
                [nn.Parameter(torch.ones(self.A.size())) for _ in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)

    def _sync_nnz(self):
        """The EMAX = 4 sparse variants of the graph-conv kernels drop entries beyond ``a_nnz_max`` per
        column / row, so the bound follows the A buffer: recomputed whenever A was written
        (load_state_dict, .to(), in-place edits bump ``_version``; a new tensor has a new data_ptr)."""
        key = (self.A.data_ptr(), self.A._version, self.A.device)
        if getattr(self, "_nnz_key", None) != key:
            nnz = a_nnz_max(self.A)
            for b in self.st_gcn_networks:
                b.a_nnz_max = nnz
                b.gcn.a_nnz_max = nnz
            self._nnz_key = key

    def _a_eff(self):
        """[A * importance] per block (classifier.py:78); in a no-grad eval forward the products are cached per
        version of A and of the importance parameters"""
        if self.training or torch.is_grad_enabled():
            return [self.A * imp for imp in self.edge_importance]
        from ..ops import tensor_key
        key = tensor_key(self.A, *[imp for imp in self.edge_importance if isinstance(imp, torch.Tensor)])
        ent = getattr(self, "_a_eff_cache", None)
        if ent is None or ent[0] != key:
            with torch.no_grad():
                ent = (key, [(self.A * imp).contiguous() for imp in self.edge_importance])
            self._a_eff_cache = ent
        return ent[1]

    def train(self, mode=True):
        self._a_eff_cache = None
        return super().train(mode)

    def _stack(self, x):
        N, C, T, V, M = x.size()
        if M != 1:
            # the reference loaders only ever produce M == 1 (utils/loader.py:65)
            x = x.permute(0, 4, 1, 2, 3).contiguous().view(N * M, C, T, V)
        else:
            x = x.reshape(N, C, T, V)
        if not x.is_cuda:
            raise _lib.StepHipError("step_amd models run on the GPU only (input is on %s)" % x.device)
        x = x.float().contiguous()
        data_bn = getattr(self, self.DATA_BN)
        self._sync_nnz()
        for i, (gcn, a_eff) in enumerate(zip(self.st_gcn_networks, self._a_eff())):
            x, _ = gcn(x, a_eff, data_bn if i == 0 else None)
        return x, (N, M)

    def _pooled(self, x):
        h, (N, M) = self._stack(x)
        f = PoolFn.apply(h)                       # (N*M, C)
        if M != 1:
            f = f.view(N, M, -1).mean(dim=1)
        return f


class Classifier(_ClassifierBase):
    def __init__(self, in_channels, num_classes, graph_args, temporal_kernel_size=75,
                 edge_importance_weighting=True, **kwargs):
        super().__init__()
        self._build_stack(in_channels, graph_args, temporal_kernel_size, edge_importance_weighting,
                          **kwargs)
        self.fcn = nn.Conv2d(64, num_classes, kernel_size=1)

    RETURNS_SOFTMAX = True

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled() and x.size(4) == 1:
            # inference: pooling, fcn and softmax in one launch (same sums, bit-identical to the path below)
            from ..ops import head_inference
            h, _ = self._stack(x)
            out, pooled = head_inference(h, self.fcn.weight.detach(), self.fcn.bias.detach(), self.RETURNS_SOFTMAX)
            return out, pooled.view(pooled.size(0), -1, 1, 1).squeeze()
        pooled = self._pooled(x)
        f = pooled.view(pooled.size(0), -1, 1, 1).squeeze()
        logits = DenseFn.apply(pooled, self.fcn.weight, self.fcn.bias, False)
        out = SoftmaxFn.apply(logits) if self.RETURNS_SOFTMAX else logits
        return out, f

    def extract_feature(self, x):
        N, C, T, V, M = x.size()
        h, _ = self._stack(x)
        _, c, t, v = h.size()
        feature = h.view(N, M, c, t, v).permute(0, 2, 3, 4, 1)
        # per-node fcn: a 1x1 conv over (N*M, c, t, v) == dense over the channel axis
        rows = h.permute(0, 2, 3, 1).reshape(-1, c)
        o = DenseFn.apply(rows, self.fcn.weight, self.fcn.bias, False)
        o = o.view(N * M, t, v, -1).permute(0, 3, 1, 2)
        output = o.reshape(N, M, -1, t, v).permute(0, 2, 3, 4, 1)
        return output, feature
