"""CVAE / Encoder / Decoder of generator_cvae (net/CVAE_stgcn.py:10-232): same constructors,
attribute names and state_dict keys (encoder.A, encoder.encoder.{i}..., decoder.decoder.{i}.gctn...)."""
import torch
import torch.nn as nn

from .. import _lib
from ..ops import PoolFn, DenseFn, DataBnFn, a_nnz_max
from .blocks import st_gcn, st_gctn
from .utils.graph import Graph


def _check(x):
    if not x.is_cuda:
        raise _lib.StepHipError("step_amd models run on the GPU only (input is on %s)" % x.device)


class CVAE(nn.Module):
    def __init__(self, in_channels, T, V, n_z, num_classes, graph_args,
                 edge_importance_weighting=False, **kwargs):
        super().__init__()
        self.T, self.V, self.n_z = T, V, n_z
        self.encoder = Encoder(in_channels + num_classes, n_z, graph_args, edge_importance_weighting)
        self.decoder = Decoder(in_channels, n_z + num_classes, graph_args, edge_importance_weighting)

    def forward(self, x, lenc, ldec):
        batch_size = x.size(0)
        mean, lsig = self.encoder(x, lenc)
        sig = torch.exp(0.5 * lsig)
        eps = torch.randn([batch_size, self.n_z], device=mean.device)
        z = eps * sig + mean
        recon_x = self.decoder(z, ldec, self.T, self.V)
        return recon_x, mean, lsig, z

    def inference(self, n=1, ldec=None):
        z = torch.randn([n, self.n_z], device=ldec.device)
        return self.decoder(z, ldec, self.T, self.V)


def _sync_nnz(module, blocks, attr):
    """see _ClassifierBase._sync_nnz: the sparse kernels' bound follows the A buffer"""
    key = (module.A.data_ptr(), module.A._version, module.A.device)
    if getattr(module, "_nnz_key", None) != key:
        nnz = a_nnz_max(module.A)
        for b in blocks:
            b.a_nnz_max = nnz
            getattr(b, attr).a_nnz_max = nnz
        module._nnz_key = key


def _importance(module, blocks, flag):
    if flag:
        return nn.ParameterList([nn.Parameter(torch.ones(module.A.size())) for _ in blocks])
    return [1] * len(blocks)


class Encoder(nn.Module):
    def __init__(self, in_channels, n_z, graph_args, edge_importance_weighting=False,
                 temporal_kernel_size=75, **kwargs):
        super().__init__()
        self.graph = Graph(**graph_args)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer("A", A)
        kernel_size = (temporal_kernel_size, A.size(0))
        self.data_bn = nn.BatchNorm1d(in_channels * A.size(1))
        self.encoder = nn.ModuleList((
            st_gcn(in_channels, 64, kernel_size, 1, **kwargs),
            st_gcn(64, 32, kernel_size, 1, **kwargs),
            st_gcn(32, 32, kernel_size, 1, **kwargs),
        ))
        _sync_nnz(self, self.encoder, "gcn")
        self.edge_importance = _importance(self, self.encoder, edge_importance_weighting)
        self.z_mean = nn.Conv2d(32, n_z, kernel_size=1)
        self.z_lsig = nn.Conv2d(32, n_z, kernel_size=1)

    def forward(self, x, l):
        _check(x)
        x = torch.cat((x, l), dim=1)
        N, C, T, V, M = x.size()
        assert M == 1, "the reference only ever feeds M == 1"
        h = x.reshape(N, C, T, V).float().contiguous()
        _sync_nnz(self, self.encoder, "gcn")
        for i, (gcn, importance) in enumerate(zip(self.encoder, self.edge_importance)):
            h, _ = gcn(h, self.A * importance, self.data_bn if i == 0 else None)
        pooled = PoolFn.apply(h)
        mean = DenseFn.apply(pooled, self.z_mean.weight, self.z_mean.bias, False)
        lsig = DenseFn.apply(pooled, self.z_lsig.weight, self.z_lsig.bias, False)
        return mean, lsig


class Decoder(nn.Module):
    def __init__(self, in_channels, n_z, graph_args, edge_importance_weighting=False,
                 temporal_kernel_size=75, **kwargs):
        super().__init__()
        self.graph = Graph(**graph_args)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer("A", A)
        kernel_size = (temporal_kernel_size, A.size(0))
        self.fcn = nn.ConvTranspose2d(n_z, 32, kernel_size=1)
        self.decoder = nn.ModuleList((
            st_gctn(32, 32, kernel_size, 1, **kwargs),
            st_gctn(32, 64, kernel_size, 1, **kwargs),
            st_gctn(64, in_channels, kernel_size, 1, **kwargs),
        ))
        _sync_nnz(self, self.decoder, "gctn")
        self.edge_importance = _importance(self, self.decoder, edge_importance_weighting)
        self.data_bn = nn.BatchNorm1d(in_channels * A.size(1))
        self.out = nn.Sigmoid()          # constructed but unused, as in the reference (:196,230)

    def forward(self, z, l, T, V):
        _check(z)
        N = z.size(0)
        zl = torch.cat((z, l), dim=1).float().contiguous()
        h = DenseFn.apply(zl, self.fcn.weight, self.fcn.bias, True)         # (N, 32)
        _sync_nnz(self, self.decoder, "gctn")
        blocks = list(zip(self.decoder, self.edge_importance))
        if not self.training and not torch.is_grad_enabled() and V == 16:
            # synthesis: z.repeat([1,1,T,V]) (:209) is never built - the first block works on the (N, 32) vectors
            h, _ = blocks[0][0].forward_repeated(h, self.A * blocks[0][1], T)
            blocks = blocks[1:]
        else:
            h = h[:, :, None, None].expand(N, h.size(1), T, V).contiguous()     # z.repeat([1,1,T,V])
        for gcn, importance in blocks:
            h, _ = gcn(h, self.A * importance)
        h = DataBnFn.apply(self.data_bn, self.training, h, self.data_bn.weight, self.data_bn.bias)
        return h.unsqueeze(4)
