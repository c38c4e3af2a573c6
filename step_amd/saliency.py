"""Gradient saliency of the classifier, with the interface of the reference's `GuidedBackprop`
(classifier_stgcn_real_only/utils/processor.py:17-113, driven by `Processor.smap`, :434-445).

What the reference computes as shipped: `update_relus()` is commented out (:28), so `generate_gradients` returns the
plain gradient of the target-class outputs with respect to the INPUT of the first graph conv's 1x1 Conv2d - the output
of `data_bn` - captured by a backward hook on `st_gcn_networks[0].gcn.conv` (:31-36), model in eval mode (:26).

Here a block is one autograd node and its container children are never called (a hook on them cannot fire: blocks.py
refuses to run while one is registered), so the same tensor is obtained without hooks: the gradient with respect to the
raw input, which the first block's backward produces, divided by the eval-mode `data_bn` scale gamma / sqrt(var + eps) of
its (joint, channel) row - `data_bn` in eval mode is a per-row affine map, so this is the chain rule run backwards.
The hook sits on gcn.conv, not on the block: the first block's residual branch reads the same tensor, and its share
of the gradient is NOT in what the reference reports; the first block therefore leaves the residual term out of its
input gradient for this call (`dx_skip_residual`; parameter gradients are unaffected)."""
import torch


class GuidedBackprop:
    def __init__(self, model):
        self.model = model
        self.gradients = None
        self.model.eval()                                                   # processor.py:26

    def generate_gradients(self, input_image, target_class):
        """processor.py:98-113: forward, one-hot of `target_class` as the output gradient, backward; returns the first
        sample's gradient as a numpy array (the reference's `[0]`); `self.gradients` keeps the whole batch, (N, C, T, V)."""
        x = input_image.detach().clone().float().requires_grad_(True)
        first = self.model.st_gcn_networks[0]
        first.dx_skip_residual = True
        try:
            out = self.model(x)
            output = out[0] if isinstance(out, tuple) else out
            self.model.zero_grad()
            one_hot = torch.zeros_like(output)
            idx = torch.as_tensor(target_class, device=output.device, dtype=torch.long)
            one_hot[torch.arange(output.shape[0], device=output.device), idx] = 1
            output.backward(gradient=one_hot)
        finally:
            first.dx_skip_residual = False
        N, C, T, V, M = x.shape
        bn = getattr(self.model, getattr(self.model, "DATA_BN", "data_bn"))
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()      # row v * C + c (classifier.py:68-72)
        g = x.grad.permute(0, 4, 1, 2, 3).reshape(N * M, C, T, V) / scale.view(V, C).t().reshape(1, C, 1, V)
        self.gradients = g
        return g.cpu().numpy()[0]
