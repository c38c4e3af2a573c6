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


@pytest.mark.parametrize("tag", ["c3", "c1"])
def test_assemble_ragged_gaits_vs_reference_lines(tag):
    """utils/loader.py:33-37 (tile to the longest gait, `cycles` repeats, alignment) + the layout change of
    :65-66 in ONE launch over the ragged frames, against what the reference's own statements produced"""
    import os
    import numpy as np
    from step_amd import pipeline
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "assemble.npz"))
    lens, frames, cycles = z[tag + ".lens"], z[tag + ".frames"], int(z[tag + ".cycles"])
    cuts = np.concatenate([[0], np.cumsum(lens)])
    data_list = [frames[cuts[i]:cuts[i + 1]] for i in range(len(lens))]
    x, time_steps = pipeline.assemble(data_list, cycles=cycles, device=D)
    ref = z[tag + ".aligned"]
    N, T = ref.shape[:2]
    assert time_steps == int(lens.max()) and tuple(x.shape) == (N, 3, T, 16, 1)
    got = x.squeeze(-1).permute(0, 2, 3, 1).reshape(N, T, 48).cpu()
    assert_close(got, torch.tensor(ref, dtype=torch.float32), 1e-5, "assembled + aligned gaits")
    # equal-length gaits: the ragged launch and the dense one agree bit for bit
    dense = np.stack([g[:5] for g in data_list if len(g) >= 5])
    a, _ = pipeline.assemble(list(dense), cycles=1, device=D)
    b = pipeline.align_gaits(dense.reshape(len(dense), 5, 16, 3), device=D)
    assert torch.equal(a, b)
