"""Batched prediction and deep-feature export.

The reference walks the data set one gait at a time with batch-1 forwards
(classifier_stgcn_real_only/utils/processor.py:457-473 ``generate_predictions`` and :489-515
``save_best_feature``), i.e. N kernel-launch-bound forwards.  In eval mode every op of the stack is
per-sample (BatchNorm uses running statistics), so the same numbers come out of large batches; these
helpers keep the reference's argument meaning and return values and run the model in chunks of
``batch`` gaits.  Data layout as in the reference: ``data`` is (N, T, joints*coords) and is viewed as
(N, coords, T, joints, 1) (processor.py:465-467).
"""
import numpy as np
import torch


def _as_model_input(chunk, joints, coords, device):
    n, t = chunk.shape[0], chunk.shape[1]
    x = np.reshape(chunk, (n, t, joints, coords, 1))
    x = np.moveaxis(x, [1, 2, 3], [2, 3, 1])                 # (n, coords, t, joints, 1)
    return torch.from_numpy(np.ascontiguousarray(x)).float().to(device)


def _run(model, data, joints, coords, batch, device):
    was_training = model.training
    model.eval()
    outs, feats = [], []
    try:
        with torch.no_grad():
            for i in range(0, data.shape[0], batch):
                x = _as_model_input(np.asarray(data[i:i + batch]), joints, coords, device)
                o, f = model(x)
                outs.append(o.detach().reshape(x.shape[0], -1).cpu())
                feats.append(f.detach().reshape(x.shape[0], -1).cpu())
    finally:
        model.train(was_training)
    return torch.cat(outs).numpy(), torch.cat(feats).numpy()


def generate_predictions(model, data, num_classes, joints, coords, batch=1024, device="cuda:0"):
    """processor.py:457-473: returns (labels_pred (N,), output (N, num_classes))"""
    output, _ = _run(model, data, joints, coords, batch, device)
    if output.shape[1] != num_classes:
        raise ValueError("model returns %d classes, expected %d" % (output.shape[1], num_classes))
    output = output.astype(np.float64)                        # the reference fills np.zeros (float64)
    return np.argmax(output, axis=1).astype(np.float64), output


def extract_features(model, data, joints, coords, keys=None, batch=1024, device="cuda:0"):
    """processor.py:489-515: the (N, 64) array ``save_best_feature`` returns; with ``keys`` also the
    per-sample mapping it writes (one dataset per key in the reference's deepFeatures h5 file)."""
    _, feats = _run(model, data, joints, coords, batch, device)
    feats = feats.astype(np.float64)
    if keys is None:
        return feats
    keys = list(keys)
    if len(keys) != feats.shape[0]:
        raise ValueError("%d keys for %d gaits" % (len(keys), feats.shape[0]))
    return feats, {k: feats[i].astype(np.float32) for i, k in enumerate(keys)}
