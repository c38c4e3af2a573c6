"""Input pipeline of the classifiers on the GPU.

``align_gaits`` is what ``utils/loader.py:37`` keeps of ``common.get_affective_features``
(classifier_stgcn_real_only/utils/common.py:91-112): every frame of gait s, relative to its own root
joint, mapped by the similarity transform that takes frame 0 of gait s onto frame 0 of gait 0
(``get_transformation``, common.py:230-279).  The reference runs it as Python loops over N*T frames
(and computes 127 further features that the loader then drops); here the N 3x3 SVDs are one batched
numpy call on the host and the affine map of all N*T*16 joints is one HIP launch that also writes the
model's (N, C, T, V, 1) layout (the reshape + moveaxis of utils/loader.py:65-66).
"""
import numpy as np
import torch

from . import _lib


def similarity_transforms(gaits):
    """[cR | t] (N, 3, 4) float64 of get_transformation(frame 0 of gait s, frame 0 of gait 0);
    gaits (N, T, V, 3).  Batched restatement of common.py:230-279 (3-D, full-rank or rank-2 poses)."""
    g = np.asarray(gaits, dtype=np.float64)
    X = g[:, 0] - g[:, 0, :1]                               # (N, V, 3) root-relative first frames
    Y = X[0]
    n = X.shape[1]
    mx, my = X.mean(1), Y.mean(0)                           # (N, 3), (3,)
    Xc, Yc = X - mx[:, None], Y - my[None]
    sx = (Xc * Xc).sum(2).mean(1)                           # (N,)
    M = np.einsum("vi,nvj->nij", Yc, Xc) / n                # Yc^T Xc per gait
    U, D, Vt = np.linalg.svd(M)
    rank = np.linalg.matrix_rank(M)
    if (rank < 2).any():
        raise ValueError("degenerate first-frame pose (rank < 2) in gait %d" % int(np.argmax(rank < 2)))
    s3 = np.ones(M.shape[0])
    full = rank > 2
    s3[full & (np.linalg.det(M) < 0)] = -1.0
    s3[~full & (np.linalg.det(U) * np.linalg.det(Vt) < 0)] = -1.0
    S = np.stack([np.ones_like(s3), np.ones_like(s3), s3], 1)        # diag entries
    R = np.einsum("nij,nj,njk->nik", U, S, Vt)
    c = (D * S).sum(1) / sx
    t = my[None] - c[:, None] * np.einsum("nij,nj->ni", R, mx)
    return np.concatenate([c[:, None, None] * R, t[:, :, None]], axis=2)


def align_gaits(gaits, device="cuda:0", transforms=None):
    """gaits (N, T, V, 3) array or tensor -> model input (N, 3, T, V, 1) on ``device``.
    ``.squeeze(-1).permute(0, 2, 3, 1).reshape(N, T, 3 * V)`` of the result is the (N, T, 48) array the
    reference's loader works with."""
    if isinstance(gaits, torch.Tensor):
        g_host = gaits.detach().cpu().numpy()
    else:
        g_host = np.asarray(gaits)
    if g_host.ndim != 4 or g_host.shape[3] != 3:
        raise _lib.StepHipError("align_gaits expects (N, T, V, 3), got %s" % (g_host.shape,))
    N, T, V, _ = g_host.shape
    M = similarity_transforms(g_host[:, :1]) if transforms is None else np.asarray(transforms)
    g_dev = torch.as_tensor(g_host, dtype=torch.float32).contiguous().to(device)
    M_dev = torch.as_tensor(M, dtype=torch.float32).contiguous().to(device)
    out = torch.empty((N, 3, T, V), dtype=torch.float32, device=g_dev.device)
    _lib.call("step_pose_align", g_dev, M_dev, out, N, T, V)
    return out.unsqueeze(-1)


def assemble(data_list, cycles=3, joints=16, coords=3, device="cuda:0"):
    """``load_data`` of utils/loader.py between the HDF5 read and the train/test split (:33-37): a list of
    variable-length gaits, each (len_s, joints*coords), becomes the aligned model input
    (N, 3, time_steps*cycles, 16, 1), time_steps = the longest gait.  The reference tiles every gait to
    time_steps frames with ``np.tile``, writes that block ``cycles`` times into a dense (N, T, 48) array and
    then aligns it; here only the ragged frames go to the GPU and ONE launch indexes the source frame
    ``(t mod time_steps) mod len_s`` while it aligns and re-lays out (``step_pose_align_ragged``).
    Returns (x, time_steps)."""
    if coords != 3:
        raise _lib.StepHipError("assemble: the alignment is 3-D (coords must be 3, got %d)" % coords)
    gaits = [np.asarray(g, dtype=np.float64).reshape(-1, joints, coords) for g in data_list]
    if not gaits or any(g.shape[0] < 1 for g in gaits):
        raise _lib.StepHipError("assemble: every gait needs at least one frame")
    lens = np.asarray([g.shape[0] for g in gaits], dtype=np.int64)
    time_steps = int(lens.max())
    N = len(gaits)
    if int(lens.sum()) >= 2 ** 31:
        raise _lib.StepHipError("assemble: more than 2^31 frames")
    first = np.stack([g[0] for g in gaits])[:, None]                     # (N, 1, V, 3): all the transforms need
    M = similarity_transforms(first)
    frames = torch.as_tensor(np.concatenate(gaits, 0), dtype=torch.float32).contiguous().to(device)
    off = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32).to(device)
    M_dev = torch.as_tensor(M, dtype=torch.float32).contiguous().to(device)
    out = torch.empty((N, 3, time_steps * cycles, joints), dtype=torch.float32, device=frames.device)
    _lib.call("step_pose_align_ragged", frames, off, M_dev, out, N, time_steps, cycles, joints)
    return out.unsqueeze(-1), time_steps
