"""Classifier of classifier_hybrid (net/classifier.py:9-89 there): the real_and_synth stack,
`data_bn1`, and the affective-feature head cat -> BatchNorm1d(64+F) -> 1x1 conv (96) ->
1x1 conv (classes) with no nonlinearity in between; forward(x_aff, x_gait) -> logits."""
import torch
import torch.nn as nn

from ..ops import DenseFn, BnRowsFn
from .classifier import _ClassifierBase
from .blocks import st_gcn  # noqa: F401


class Classifier(_ClassifierBase):
    BLOCKS = ((32, 1, {"residual": False}), (64, 2, {}), (64, 2, {}))
    DATA_BN = "data_bn1"

    def __init__(self, in_channels, in_features, num_classes, graph_args, temporal_kernel_size=75,
                 edge_importance_weighting=True, **kwargs):
        super().__init__()
        self._build_stack(in_channels, graph_args, temporal_kernel_size, edge_importance_weighting,
                          **kwargs)
        self.data_bn2 = nn.BatchNorm1d(64 + in_features)
        self.combined_networks = nn.ModuleList((
            nn.Conv2d(64 + in_features, 96, kernel_size=1),
            nn.Conv2d(96, num_classes, kernel_size=1),
        ))
        # keep the reference's registration order of parameters (edge_importance last)
        ei = self._parameters.pop("edge_importance", None) or self._modules.pop("edge_importance", None)
        if ei is not None:
            self.edge_importance = ei

    def forward(self, x_aff, x_gait):
        pooled = self._pooled(x_gait)
        z = torch.cat((pooled, x_aff.to(pooled.device).float()), dim=1).contiguous()
        z = BnRowsFn.apply(self.data_bn2, self.training, z, self.data_bn2.weight, self.data_bn2.bias)
        for net in self.combined_networks:
            z = DenseFn.apply(z, net.weight, net.bias, False)
        return z.view(z.size(0), -1)
