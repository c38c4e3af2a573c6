"""CPU restatement of the hybrid classifier's affective-feature extractor (TEST INFRASTRUCTURE ONLY,
see oracle/__init__.py).

Reference: compute_aff_features/compute_features.py:428-468 ``compute_features`` (27 per-gait posture
and movement features, 29 with the two stride features), its helpers in compute_aff_features/utils.py
and compute_aff_features/normalize_features.py:1-22 (min-max map of every feature to [-1, 1] over the
data set).  Vectorised numpy in float64, as the reference computes.  Parity PINNED by
tests/golden/affective.npz (outputs of the reference's own functions on the shipped 60-gait HDF5
fixture and on seeded gaits that reach every branch; tests/golden/make_golden.py gen_affective).
"""
import numpy as np

R_HAND, L_HAND, HEAD, R_FOOT, L_FOOT = 6, 9, 3, 12, 15       # compute_features.py:272-294
MOVERS = (R_HAND, L_HAND, HEAD, R_FOOT, L_FOOT)


def _angle(a, b):
    """utils.py:7-19: degrees between a and b, cosine clipped to [-1, 1]"""
    with np.errstate(all="ignore"):
        ua = a / np.linalg.norm(a, axis=-1, keepdims=True)
        ub = b / np.linalg.norm(b, axis=-1, keepdims=True)
        return (180 / np.pi) * np.arccos(np.clip(np.sum(ua * ub, -1), -1.0, 1.0))


def _dist(a, b):
    return np.linalg.norm(a - b, axis=-1)


def _heron(p, q, r):
    """utils.py:24-34"""
    a, b, c = _dist(p, q), _dist(p, r), _dist(q, r)
    s = (a + b + c) / 2
    with np.errstate(all="ignore"):
        return (s * (s - a) * (s - b) * (s - c)) ** 0.5


def _volume(J):
    """compute_features.py:12-37.  The max is only looked at when the joint did NOT lower the min
    (``if min > x ... elif max < x``), so the box is not always the true bounding box."""
    T = J.shape[0]
    lo = np.full((T, 3), np.inf)
    hi = np.full((T, 3), -np.inf)
    for i in range(16):
        x = J[:, i, :]
        new_min = lo > x
        new_max = ~new_min & (hi < x)
        lo = np.where(new_min, x, lo)
        hi = np.where(new_max, x, hi)
    with np.errstate(all="ignore"):
        e = hi - lo
        return e[:, 0] * e[:, 1] * e[:, 2] / 1000


def strike_points(y):
    """compute_features.py:393-404: frames where a foot's height has a local minimum; on a plateau
    every other frame counts"""
    pts = []
    for i in range(1, len(y) - 1):
        p, c, n = y[i - 1], y[i], y[i + 1]
        if p == c == n:
            if not (i - 1 in pts or i in pts):
                pts.append(i)
        elif p >= c <= n:
            pts.append(i)
    return np.asarray(pts)


def stride(frames, time_step):
    """compute_features.py:407-420"""
    r = strike_points(frames[:, 3 * R_FOOT + 1])
    l = strike_points(frames[:, 3 * L_FOOT + 1])
    T = len(frames)
    if len(r) < 2 and len(l) < 2:
        return float(T), T * time_step
    if len(r) < 2:
        m = np.mean(l[1:] - l[:-1] + 1)
    elif len(l) < 2:
        m = np.mean(r[1:] - r[:-1] + 1)
    else:
        m = np.mean(np.concatenate((r[1:] - r[:-1], l[1:] - l[:-1])))
    return float(m), float(m) * time_step


def compute_features(frames, time_step, add_stride_features=False):
    """one gait (T, 48) -> 27 (29) features, compute_features.py:428-468"""
    frames = np.asarray(frames, dtype=np.float64)
    J = frames.reshape(len(frames), 16, 3)
    root, spine, neck, head = J[:, 0], J[:, 1], J[:, 2], J[:, 3]
    rsh, lsh = J[:, 4], J[:, 7]
    up = np.array([0.0, 1.0, 0.0])
    f = [
        _volume(J).mean(),
        _angle(rsh - neck, lsh - neck).mean(),                   # :49-57
        _angle(neck - rsh, lsh - rsh).mean(),                    # :69-77
        _angle(neck - lsh, rsh - lsh).mean(),                    # :89-97
        _angle(head - root, up).mean(),                          # :109-116
        _angle(head - neck, spine - neck).mean(),                # :128-136
        (_dist(J[:, R_HAND], root) / 10).mean(),                 # :148-154
        (_dist(J[:, L_HAND], root) / 10).mean(),
        (_dist(J[:, R_FOOT], root) / 10).mean(),
        (_dist(J[:, L_FOOT], root) / 10).mean(),
        (_heron(J[:, L_HAND], neck, J[:, R_HAND]) / 100).mean(),  # :220-228
        (_heron(J[:, L_FOOT], root, J[:, R_FOOT]) / 100).mean(),  # :240-248
    ]
    with np.errstate(all="ignore"):
        for j in MOVERS:                                         # :260-269
            f.append((np.linalg.norm(np.diff(J[:, j], axis=0), axis=-1) / 10 / time_step).mean())
        vel = {j: np.diff(J[:, j], axis=0) / time_step for j in MOVERS}
        acc = {j: np.diff(vel[j], axis=0) / time_step for j in MOVERS}
        for j in MOVERS:                                         # :298-313
            f.append((np.linalg.norm(acc[j], axis=-1) / 10).mean())
        for j in MOVERS:                                         # :342-364
            f.append((np.linalg.norm(np.diff(acc[j], axis=0) / time_step, axis=-1) / 10).mean())
    if add_stride_features:
        f.extend(stride(frames, time_step))
    return np.asarray(f, dtype=np.float64)


def compute_features_batch(gaits, time_step, add_stride_features=False):
    return np.stack([compute_features(g, time_step, add_stride_features) for g in gaits])


def normalize_features(F):
    """normalize_features.py:1-22 on an (N, F) array: (f - (max+min)/2) / ((max-min)/2), 0 where max == min"""
    F = np.asarray(F, dtype=np.float64)
    lo, hi = np.full(F.shape[1], np.inf), np.full(F.shape[1], -np.inf)
    for row in F:                                                # same comparisons, so NaNs act the same
        lo = np.where(lo > row, row, lo)
        hi = np.where(hi < row, row, hi)
    a, b = (hi + lo) / 2, (hi - lo) / 2
    with np.errstate(all="ignore"):
        out = (F - a) / b
    return np.where(b == 0, 0.0, out)
