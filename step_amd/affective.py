"""Affective features of the hybrid classifier on the GPU.

``compute_features`` / ``normalize_features`` keep the names and meaning of
compute_aff_features/compute_features.py:428-468 and compute_aff_features/normalize_features.py:1-22,
but take the whole data set at once: (N, T, 48) in, (N, 27 | 29) out, one HIP launch each, instead of
the reference's per-gait, per-frame Python loops.  The result of ``normalize_features`` is the
``x_aff`` input of classifier_hybrid's ``Classifier.forward(x_aff, x_gait)``.
"""
import numpy as np
import torch

from . import _lib

NUM_FEATURES = 27          # + 2 with add_stride_features (compute_features.py:465-467)


def _to_device(gaits, device):
    if not isinstance(gaits, torch.Tensor):
        gaits = torch.as_tensor(np.ascontiguousarray(gaits))
    if gaits.dim() == 4:                                   # (N, T, 16, 3) -> (N, T, 48)
        gaits = gaits.reshape(gaits.shape[0], gaits.shape[1], -1)
    if gaits.dim() != 3 or gaits.shape[2] != 48:
        raise _lib.StepHipError("compute_features expects (N, T, 48) gaits, got %s" % (tuple(gaits.shape),))
    if gaits.dtype not in (torch.float32, torch.float64):
        gaits = gaits.double()
    return gaits.to(device).contiguous()


def compute_features(gaits, time_step, add_stride_features=False, device="cuda:0"):
    """(N, T, 48) gaits (array or tensor, float32 / float64) -> (N, 27 | 29) float64 tensor on ``device``"""
    g = _to_device(gaits, device)
    N, T, _ = g.shape
    F = NUM_FEATURES + (2 if add_stride_features else 0)
    feat = torch.empty((N, F), dtype=torch.float64, device=g.device)
    _lib.call("step_aff_features", g, int(g.dtype == torch.float64), feat, N, T, float(time_step),
              int(bool(add_stride_features)))
    return feat


def normalize_features(features, dtype=torch.float32):
    """(N, F) float64 device tensor -> min-max normalised (N, F) tensor in [-1, 1] (constant columns -> 0)"""
    if not isinstance(features, torch.Tensor) or features.dtype != torch.float64 or features.dim() != 2:
        raise _lib.StepHipError("normalize_features expects the (N, F) float64 tensor of compute_features")
    if dtype not in (torch.float32, torch.float64):
        raise _lib.StepHipError("normalize_features writes float32 or float64")
    f = features.contiguous()
    out = torch.empty(f.shape, dtype=dtype, device=f.device)
    _lib.call("step_aff_normalize", f, out, int(dtype == torch.float64), f.shape[0], f.shape[1])
    return out
