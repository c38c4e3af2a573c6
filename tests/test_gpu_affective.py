"""GPU: affective features of the hybrid classifier against the reference's own output."""
import os

import numpy as np
import pytest
import torch

from tests.test_oracle_golden import _same

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
TS = 1.0 / 30.0


def _fixture():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "affective.npz"))


@pytest.mark.parametrize("tag", ["h5", "walk", "short", "mono"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_affective_features_vs_reference_fixture(tag, dtype):
    """compute_features.py:428-468 in one launch; float64 input reproduces the reference to rounding,
    float32 input to float32's input rounding (jerk amplifies it by 1/dt^3)"""
    from step_amd import affective
    z = _fixture()
    g = z[tag + ".gaits"].astype(dtype)
    exact = dtype == np.float64
    f27 = affective.compute_features(g, TS, device=D)
    f29 = affective.compute_features(g, TS, add_stride_features=True, device=D)
    assert f27.dtype == torch.float64 and tuple(f29.shape) == (g.shape[0], 29)
    if exact:
        _same(f27.cpu().numpy(), z[tag + ".f27"], 1e-10)
        _same(f29.cpu().numpy(), z[tag + ".f29"], 1e-10)
        assert np.array_equal(f29.cpu().numpy()[:, 27], z[tag + ".f29"][:, 27])
    else:
        from oracle import affective as oa
        ref = oa.compute_features_batch(g.astype(np.float64), TS, True)   # same float32-rounded input
        _same(f29.cpu().numpy(), ref, 1e-10)
        _same(f29.cpu().numpy()[:, :12], z[tag + ".f29"][:, :12], 1e-4)   # and close to the float64 answer


@pytest.mark.parametrize("tag", ["h5", "walk", "short", "mono"])
def test_affective_normalize_vs_reference_fixture(tag):
    """normalize_features.py:1-22"""
    from step_amd import affective
    z = _fixture()
    f = torch.tensor(z[tag + ".f29"], dtype=torch.float64, device=D)
    _same(affective.normalize_features(f, torch.float64).cpu().numpy(), z[tag + ".norm29"], 1e-12)
    out = affective.normalize_features(f)
    assert out.dtype == torch.float32
    _same(out.cpu().numpy(), z[tag + ".norm29"].astype(np.float32), 1e-6)


def test_affective_long_sequences_vs_oracle():
    """several 128-frame chunks, ragged last chunk, plateaus across chunk boundaries; hybrid T = 1024"""
    from oracle import affective as oa
    from step_amd import affective
    rng = np.random.default_rng(11)
    for T in (129, 300, 1024):
        g = np.cumsum(0.3 * rng.standard_normal((6, T, 48)), axis=1)
        for j in (12, 15):
            g[:, :, 3 * j + 1] = np.round(g[:, :, 3 * j + 1])
        g[1, 100:140, 3 * 12 + 1] = 2.0                       # a plateau over the first chunk boundary
        g[2, :, 3 * 15 + 1] = 0.0
        ref = oa.compute_features_batch(g, TS, True)
        got = affective.compute_features(g, TS, add_stride_features=True, device=D).cpu().numpy()
        _same(got, ref, 1e-10)
        assert np.array_equal(got[:, 27], ref[:, 27])


def test_affective_features_feed_the_hybrid_classifier():
    """features -> normalise -> x_aff of classifier_hybrid's forward (classifier_hybrid/net/classifier.py:61-89)"""
    from step_amd import affective
    from step_amd.net.classifier_hybrid import Classifier
    rng = np.random.default_rng(3)
    g = np.cumsum(0.3 * rng.standard_normal((8, 75, 48)), axis=1)
    x_aff = affective.normalize_features(affective.compute_features(g, TS, device=D))
    assert tuple(x_aff.shape) == (8, 27) and float(x_aff.abs().max()) <= 1.0 + 1e-6
    torch.manual_seed(0)
    m = Classifier(3, 27, 4, {"strategy": "spatial"}).to(D).eval()
    x_gait = torch.tensor(g, dtype=torch.float32).reshape(8, 75, 16, 3, 1).permute(0, 3, 1, 2, 4).contiguous().to(D)
    with torch.no_grad():
        out = m(x_aff, x_gait)
    assert tuple(out.shape) == (8, 4) and bool(torch.isfinite(out).all())
