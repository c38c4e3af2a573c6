"""CPU (no GPU needed): the C-ABI library loads and exports every symbol include/step_hip.h
declares; the host-side module contract (constructors, state_dict keys, checkpoint loading,
weights_init by class name); loud failure without a GPU; the data-parallel path on gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import load_npz, ckpt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "step_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(step_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from step_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libstep_hip.so lacks %s" % s
        assert s in _lib.SIGNATURES, "ctypes binding lacks %s" % s
    assert set(_lib.SIGNATURES) == set(syms)
    l = _lib.load()
    assert l.step_abi_version() == 1
    # size helpers are pure host code: callable without a GPU
    assert l.step_bn_ws_bytes(48) > 0
    assert l.step_tcn_packed_elems(64, 64, 75, 1, 0) == 64 * 76 * 64 + 256
    assert l.step_tcn_packed_elems(64, 64, 75, 2, 2) == 64 * 38 * 128 + 256
    assert l.step_tcn_packed_elems(64, 64, 8, 1, 0) == 0          # even Kt is rejected
    assert l.step_tcn_wgrad_ws_bytes(512, 64, 64, 240, 120, 75) > 0
    assert l.step_gcn_bwd_ws_bytes(512, 64, 64, 3, 240) > 0


def test_shape_errors_are_reported_before_launch():
    from step_amd import _lib
    l = _lib.load()
    rc = l.step_tcn_conv(None, None, None, 0, None, None, None, None, None, None, None, 0, None,
                         1, 3, 3, 10, 10, 75, 1, 0, None)
    assert rc == -1 and b"null pointer" in l.step_last_error()
    rc = l.step_gcn_fwd(1, None, None, 1, None, 0, 1, 4, 1, None, None, 0, None, 2, 3, 128, 3, 10, 1, None)
    assert rc == -1 and b"Co=128" in l.step_last_error()          # wide layers need Ci % 16 == 0
    # the split-operand twins check the same arguments before anything is launched
    rc = l.step_gcn_fwd_split(1, None, None, 1, None, 0, 1, 4, 1, None, None, 0, None, 2, 32, 64, 4, 10, 1, None)
    assert rc == -1 and b"K=4" in l.step_last_error()
    rc = l.step_gcn_fwd_split(1, None, None, 1, None, 0, 1, 4, 1, 1, None, 0, None, 2, 32, 64, 3, 10, 1, None)
    assert rc == -1 and b"res_w/res_out pair" in l.step_last_error()
    rc = l.step_gcn_bwd_split(1, None, 1, None, None, 1, None, 0, 1, 4, None, 0, 1, None, 1, None, None, None, None,
                              None, 2, 64, 64, 3, 10, 1, 1, 0, None)
    assert rc == -1 and b"id_dout and id_out come together" in l.step_last_error()
    rc = l.step_gcn_bwd_split(1, None, 1, None, None, 1, None, 0, 1, 4, None, 0, None, None, 1, None, None, None, None,
                              None, 2, 64, 64, 3, 10, 1, 1, 0, None)
    assert rc != 0 and b"workspace" in l.step_last_error()
    assert l.step_gcn_bwd_ws_bytes(64, 128, 256, 3, 60) > 0


def test_models_keep_reference_state_dict_contract():
    from step_amd.net import classifier, classifier_real_and_synth, classifier_hybrid, CVAE_stgcn
    g = {"strategy": "spatial"}
    z = load_npz("ckpt_classifier_real_only_epoch395")
    m = classifier.Classifier(3, 4, g)
    sd = m.state_dict()
    assert list(sd) == list(z.files)
    assert all(tuple(sd[k].shape) == z[k].shape for k in z.files)
    m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
    np.testing.assert_array_equal(m.A.numpy(), z["A"])
    assert sum(p.numel() for p in m.parameters()) == 716292
    zc = load_npz("ckpt_cvae_epoch105")
    c = CVAE_stgcn.CVAE(3, 75, 16, 32, 4, g)
    assert list(c.state_dict()) == list(zc.files)
    c.load_state_dict(ckpt("ckpt_cvae_epoch105"))
    assert sum(p.numel() for p in c.parameters()) == 876484
    tr = load_npz("train_steps")
    for variant, mod in (("real_and_synth", classifier_real_and_synth.Classifier(3, 4, g, temporal_kernel_size=9)),
                         ("hybrid", classifier_hybrid.Classifier(3, 29, 4, g, temporal_kernel_size=9))):
        keys = [k[len(variant) + 7:] for k in tr.files if k.startswith(variant + "_k9.P0.")]
        assert list(mod.state_dict()) == keys, variant
    # GuidedBackprop of the reference reaches these attributes (processor.py:35,62-66)
    assert isinstance(m.st_gcn_networks[0].gcn.conv, torch.nn.Conv2d)
    assert isinstance(m.st_gcn_networks[0].relu, torch.nn.ReLU)


def test_weights_init_matches_by_class_name():
    """processor.py:117-129 dispatches on the class NAME: Conv2d and BatchNorm* are re-initialised,
    ConvTranspose2d is not."""
    from step_amd.net import CVAE_stgcn
    names = {type(mod).__name__ for mod in CVAE_stgcn.CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"}).modules()}
    assert {"Conv2d", "ConvTranspose2d", "BatchNorm1d", "BatchNorm2d"} <= names
    for n in names:
        if n.startswith("ConvT"):
            assert n.find("Conv2d") == -1


def test_cpu_input_fails_loudly():
    from step_amd.net import classifier
    from step_amd._lib import StepHipError
    m = classifier.Classifier(3, 4, {"strategy": "spatial"})
    with pytest.raises(StepHipError):
        m(torch.zeros(2, 3, 20, 16, 1))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "step_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), os.path.join(dp, f)


def test_shard_batch_partitions():
    from step_amd.parallel import shard_batch
    for n, w in ((4096, 8), (100000, 8), (7, 3), (5, 8)):
        cuts = [shard_batch(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '__ROOT__')
from step_amd import parallel
rank, local, world = parallel.init_from_env(backend="gloo")
torch.manual_seed(0)
lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
with torch.no_grad():
    for p in lin.parameters():
        p.add_(rank)                      # ranks start different ...
parallel.broadcast_parameters(lin)        # ... and are made equal to rank 0
g = torch.Generator().manual_seed(1)
X = torch.randn(8, 6, generator=g); Y = torch.randint(0, 3, (8,), generator=g)
a, b = parallel.shard_batch(8, rank, world)
loss = torch.nn.functional.cross_entropy(lin(X[a:b]), Y[a:b])
loss.backward()
bucket = parallel.GradBucket(lin.parameters())
bucket.all_reduce_mean()
# single-process reference on the full batch (equal shard sizes -> mean of shard means)
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
torch.manual_seed(0)
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
torch.nn.functional.cross_entropy(ref(X), Y).backward()
for p, q in zip(lin.parameters(), ref.parameters()):
    assert torch.allclose(p.grad, q.grad, atol=1e-6), (rank, (p.grad - q.grad).abs().max())
assert bucket.numel == sum(p.numel() for p in ref.parameters())
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank%d-ok\n" % rank); sys.stdout.flush()
"""


def test_gradient_allreduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.replace('__ROOT__', ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0-ok" in out.stdout and "rank1-ok" in out.stdout


def test_hooks_on_block_children_are_refused():
    """GuidedBackprop (utils/processor.py:35,62-66) hooks st_gcn_networks[0].gcn.conv and every block's .relu; those
    children never run here, so the block refuses instead of silently producing no saliency map"""
    import pytest
    import torch
    from step_amd import _lib
    from step_amd.net.classifier import Classifier
    m = Classifier(3, 4, {"strategy": "spatial"})
    h = m.st_gcn_networks[1].relu.register_forward_hook(lambda mod, i, o: None)
    with pytest.raises(_lib.StepHipError, match="hook"):
        m.st_gcn_networks[1](torch.zeros(1, 32, 4, 16), m.A)
    h.remove()
    h = m.st_gcn_networks[0].gcn.conv.register_full_backward_hook(lambda mod, gi, go: None)
    with pytest.raises(_lib.StepHipError, match="gcn.conv"):
        m.st_gcn_networks[0](torch.zeros(1, 3, 4, 16), m.A)
    h.remove()
    # the decoder's synthesis fast path (forward_repeated) refuses as well
    from step_amd.net.CVAE_stgcn import CVAE
    c = CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    blk = c.decoder.decoder[0]
    h = blk.relu.register_forward_hook(lambda mod, i, o: None)
    with pytest.raises(_lib.StepHipError, match="hook"):
        blk.forward_repeated(torch.zeros(2, 32), c.decoder.A, 75)
    h.remove()


def test_weights_init_moves_exactly_the_conv2d_and_batchnorm_tensors():
    """a11: processor.py:117-129 applied through `model.apply` to a step_amd CVAE re-initialises every Conv2d /
    BatchNorm* weight and bias and NO ConvTranspose2d tensor ('Conv2d' is not a substring of 'ConvTranspose2d'), so the
    reference harness initialises a step_amd model exactly as it initialises its own."""
    from step_amd import harness
    from step_amd.net import CVAE_stgcn
    torch.manual_seed(11)
    m = CVAE_stgcn.CVAE(3, 75, 16, 32, 4, {"strategy": "spatial"})
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.123)                                   # so that "bias filled with 0" is a visible move
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m.apply(harness.weights_init)
    after = m.state_dict()
    expect_moved = set()
    for name, mod in m.named_modules():
        cls = type(mod).__name__
        if cls in ("Conv2d", "BatchNorm1d", "BatchNorm2d"):
            expect_moved |= {name + ".weight", name + ".bias"}
        elif cls == "ConvTranspose2d":
            assert torch.equal(after[name + ".weight"], before[name + ".weight"]), name
    moved = {k for k in after if not torch.equal(after[k], before[k])}
    assert moved == expect_moved and len(moved) > 40
    assert any(".gctn.conv" in k or "decoder.fcn" in k for k in set(after) - moved)
    for k in moved:
        if k.endswith(".bias"):
            assert float(after[k].abs().max()) == 0.0, k
    w = after["encoder.encoder.1.tcn.2.weight"]
    assert abs(float(w.std()) - 0.02) < 2e-3 and abs(float(w.mean())) < 2e-3
    g = after["encoder.encoder.1.tcn.0.weight"]
    assert abs(float(g.mean()) - 1.0) < 2e-2


def test_epoch_harness_host_logic(tmp_path):
    """step_amd.harness: step epochs / adjust_lr (processor.py:238,260-268), top-k (processor.py:294-307), checkpoint
    naming and look-up (processor.py:142-148,403-406), the synthetic 342-gait split"""
    from step_amd import harness
    assert harness.step_epochs(500) == [250, 375, 438]
    assert harness.step_epochs(4) == [2, 3, 4]
    steps = harness.step_epochs(8)
    assert steps == [4, 6, 7]
    lrs, lr = [], 0.1
    for e in range(8):
        lr = harness.lr_at(e, 0.1, steps) or lr
        lrs.append(round(lr, 7))
    assert lrs == [0.1, 0.1, 0.1, 0.1, 0.01, 0.01, 0.001, 0.0001]
    res = np.array([[0.1, 0.7, 0.2], [0.5, 0.3, 0.2], [0.2, 0.3, 0.5], [0.6, 0.3, 0.1]])
    assert harness.topk_accuracy(res, [1, 1, 2, 2], 1) == 50.0
    assert harness.topk_accuracy(res, [1, 1, 2, 2], 2) == 75.0
    for name in ("epoch3_acc62.50_model.pth.tar", "epoch12_acc81.25_model.pth.tar", "log.txt", "epoch1_acc40.62_model.pth.tar"):
        (tmp_path / name).write_text("x")
    assert harness.CKPT_FORMAT.format(12, 81.25) == "epoch12_acc81.25_model.pth.tar"
    e, a, p = harness.best_checkpoint(str(tmp_path))
    assert (e, a) == (12, 81.25) and p.endswith("epoch12_acc81.25_model.pth.tar")
    (tr_x, tr_y), (te_x, te_y) = harness.synthetic_gaits(342, 75, 4, seed=1)
    assert tr_x.shape == (307, 3, 75, 16, 1) and te_x.shape == (35, 3, 75, 16, 1)
    assert float(tr_x[:, :, :, 0].abs().max()) == 0.0          # root-relative
    assert set(tr_y.tolist()) == {0, 1, 2, 3} and -1.5 < float(tr_x.min()) and float(tr_x.max()) < 1.5


_UNUSED_PARAM_WORKER = r"""
import sys, torch
sys.path.insert(0, '__ROOT__')
from step_amd import parallel
torch.manual_seed(0)
a, b = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)          # b is never used in the second step
params = list(a.parameters()) + list(b.parameters())
bucket = parallel.GradBucket(params)
opt = torch.optim.SGD(params, lr=0.1)
x = torch.randn(4, 3)
bucket.zero_grad()
(a(x).sum() + b(x).sum()).backward()
assert float(bucket.flat.abs().sum()) > 0
opt.zero_grad(set_to_none=True)                              # drops every view
a(x).sum().backward()                                        # b gets no gradient this step
bucket._attach()
off = sum(p.numel() for p in a.parameters())
assert float(bucket.flat[off:].abs().max()) == 0.0, "stale gradient of an unused parameter survived"
assert all(p.grad is not None for p in params)
assert float(bucket.flat[:off].abs().sum()) > 0
print("ok")
"""


def test_grad_bucket_zeroes_slices_dropped_by_set_to_none(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_UNUSED_PARAM_WORKER.replace("__ROOT__", ROOT))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
