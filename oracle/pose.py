"""CPU restatement of the loader's pose alignment (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Reference: classifier_stgcn_real_only/utils/common.py:230-279 ``get_transformation`` (Umeyama's
least-squares similarity transform) and :91-112, the first 48 columns of ``get_affective_features`` -
the only ones utils/loader.py:37 keeps: every frame of gait s, taken relative to its own root joint
(common.py:47-66), is mapped by the similarity transform that takes frame 0 of gait s onto frame 0 of
gait 0.  Parity PINNED by tests/golden/pose_align.npz
(outputs of the reference's own functions, tests/golden/make_golden.py gen_pose_align).
"""
import numpy as np


def get_transformation(X, Y):
    """common.py:230-279: R, c, t minimising ||Y - (c R X + t)||; X, Y are (3, n)"""
    m, n = X.shape
    mx, my = X.mean(1), Y.mean(1)
    Xc, Yc = X - mx[:, None], Y - my[:, None]
    sx = np.mean(np.sum(Xc * Xc, 0))
    M = Yc @ Xc.T / n
    U, D, Vt = np.linalg.svd(M)
    V = Vt.T
    r = np.linalg.matrix_rank(M)
    S = np.eye(m)
    if r > m - 1:
        if np.linalg.det(M) < 0:
            S[m - 1, m - 1] = -1
    elif r == m - 1:
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            S[m - 1, m - 1] = -1
    else:
        raise ValueError("degenerate pose (rank %d): the reference returns a 2-D identity here" % r)
    R = U @ S @ V.T
    c = np.trace(np.diag(D) @ S) / sx
    t = my - c * (R @ mx)
    return R, c, t


def get_joints(g, s, t):
    """common.py:47-66: the 16 joints of frame t relative to that frame's root joint"""
    return g[s, t] - g[s, t, 0]


def align_gaits(gaits):
    """gaits (N, T, 16, 3) -> (N, T, 48): common.py:105-112"""
    N, T = gaits.shape[:2]
    out = np.zeros((N, T, 48))
    Y = get_joints(gaits, 0, 0).T
    for s in range(N):
        R, c, t = get_transformation(get_joints(gaits, s, 0).T, Y)
        for k in range(T):
            out[s, k] = ((c * R) @ get_joints(gaits, s, k).T + t[:, None]).T.flatten()
    return out


def assemble(data_list, cycles, joints=16, coords=3):
    """utils/loader.py:33-37: gaits of different lengths, each (len_s, joints*coords), tiled to the longest
    (``np.tile(...)[0:time_steps]``), that block repeated ``cycles`` times, then aligned -> (N, T*cycles, 48).
    Pinned by tests/golden/assemble.npz (the reference's own lines run on ragged cuts of the shipped gaits)."""
    time_steps = max(len(g) for g in data_list)
    data = np.empty((len(data_list), time_steps * cycles, joints * coords))
    for s, g in enumerate(data_list):
        g = np.asarray(g)
        idx = np.arange(time_steps) % len(g)
        data[s] = np.concatenate([g[idx]] * cycles, 0)
    padded = data.copy()
    return align_gaits(data.reshape(len(data_list), -1, joints, coords)), padded
