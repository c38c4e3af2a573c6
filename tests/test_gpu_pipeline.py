"""GPU: pose alignment + layout change against the reference's own output."""
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


# ------------------------------------------------------------- pose alignment
@pytest.mark.parametrize("tag", ["a", "b"])
def test_pose_alignment_vs_reference_fixture(tag):
    """common.py:91-112 + loader.py:65-66 in one launch: values of the reference, model layout"""
    import os
    import numpy as np
    from step_amd import pipeline
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_align.npz"))
    gaits, ref = z[tag + ".gaits"], z[tag + ".aligned"]
    x = pipeline.align_gaits(gaits, device=D)
    N, T = gaits.shape[:2]
    assert tuple(x.shape) == (N, 3, T, 16, 1)
    got = x.squeeze(-1).permute(0, 2, 3, 1).reshape(N, T, 48).cpu()
    assert_close(got, torch.tensor(ref, dtype=torch.float32), 1e-5, "aligned coordinates")
    # the loader's own layout change of the reference array gives the same model input
    want = np.moveaxis(np.reshape(ref, (N, T, 16, 3, 1)), [1, 2, 3], [2, 3, 1])
    assert_close(x.cpu(), torch.tensor(want, dtype=torch.float32), 1e-5, "model layout")


def test_pose_alignment_large_batch_vs_oracle():
    import numpy as np
    from oracle import pose
    from step_amd import pipeline
    rng = np.random.default_rng(5)
    gaits = rng.standard_normal((64, 75, 16, 3)) + 3.0 * rng.standard_normal((64, 1, 1, 3))
    ref = pose.align_gaits(gaits)
    x = pipeline.align_gaits(torch.tensor(gaits), device=D)
    got = x.squeeze(-1).permute(0, 2, 3, 1).reshape(64, 75, 48).cpu()
    assert_close(got, torch.tensor(ref, dtype=torch.float32), 1e-5, "aligned coordinates")
