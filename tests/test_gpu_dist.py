"""GPU: the data-parallel path on the real model.  Two ranks share cuda:0 (RCCL refuses two ranks
on one device, so the collectives run over gloo through host memory - the same GradBucket / broadcast code
reduces the device buffer in place under 'nccl'); each runs the real_only classifier on its contiguous half of
a batch.  The bucket-averaged gradients must equal, BIT FOR BIT, the mean of the two shard gradients a single
process computes, and broadcast_parameters must make rank 1 start from rank 0's weights."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from tests.util import ckpt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch, torch.nn.functional as F, torch.distributed as dist
sys.path.insert(0, '__ROOT__')
from step_amd import parallel
from step_amd.net.classifier import Classifier
from tests.util import ckpt
rank, local, world = parallel.init_from_env()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
m = Classifier(3, 4, {"strategy": "spatial"})
m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
with torch.no_grad():
    for p in m.parameters():
        p.add_(0.01 * rank)                 # ranks start different ...
m = m.to(dev).train()
parallel.broadcast_parameters(m)            # ... and are made equal to rank 0
parallel.sync_batchnorm(os.environ.get("STEP_TEST_SYNCBN") == "1")
g = torch.Generator().manual_seed(5)
X = torch.randn(12, 3, 75, 16, 1, generator=g) * 0.5
Y = torch.randint(0, 4, (12,), generator=g)
a, b = parallel.shard_batch(12, rank, world)
bucket = parallel.GradBucket(m.parameters())
bucket.zero_grad()
out, _ = m(X[a:b].to(dev))
F.cross_entropy(out, Y[a:b].to(dev)).backward()
assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket._views))   # still views
bucket.all_reduce_mean()
torch.cuda.synchronize()
torch.save({"flat": bucket.flat.cpu(), "w0": next(m.parameters()).detach().cpu(),
            "rm": m.st_gcn_networks[0].tcn[0].running_mean.cpu(), "rv": m.st_gcn_networks[2].tcn[3].running_var.cpu(),
            "dbn_rm": m.data_bn.running_mean.cpu(), "out": out.detach().cpu()}, '__OUT__/rank%d.pt' % rank)
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank%d-ok\n" % rank); sys.stdout.flush()
"""


def test_two_rank_gradients_equal_mean_of_shard_gradients(tmp_path):
    from step_amd import parallel
    from step_amd.net.classifier import Classifier
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.replace("__ROOT__", ROOT).replace("__OUT__", str(tmp_path)))
    env = dict(os.environ, STEP_DIST_BACKEND="gloo", STEP_LOCAL_DEVICE="0", OMP_NUM_THREADS="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["flat"], r1["flat"])                    # every rank holds the same averaged gradients
    assert torch.equal(r0["w0"], r1["w0"])                        # broadcast made the ranks equal

    # single process: gradients of each contiguous shard (per-shard BN statistics = DDP semantics), then the mean
    D = "cuda:0"
    g = torch.Generator().manual_seed(5)
    X = torch.randn(12, 3, 75, 16, 1, generator=g) * 0.5
    Y = torch.randint(0, 4, (12,), generator=g)
    flats, rms = [], []
    for rank in range(2):
        m = Classifier(3, 4, {"strategy": "spatial"})
        m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
        m = m.to(D).train()
        a, b = parallel.shard_batch(12, rank, 2)
        o, _ = m(X[a:b].to(D))
        F.cross_entropy(o, Y[a:b].to(D)).backward()
        flats.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu())
        rms.append(m.st_gcn_networks[0].tcn[0].running_mean.cpu())
    want = (flats[0] + flats[1]) / 2
    assert torch.equal(r0["flat"], want), float((r0["flat"] - want).abs().max())
    assert torch.equal(r0["rm"], rms[0]) and torch.equal(r1["rm"], rms[1])    # BN statistics stay per rank
    assert torch.equal(r0["w0"], next(m.parameters()).detach().cpu())


def test_two_ranks_with_synchronised_batchnorm_equal_one_device(tmp_path):
    """parallel.sync_batchnorm: 2 ranks x 6 sequences give what ONE device gives on the 12 (the reference's
    single-device BatchNorm, classifier.py:154-164): outputs, running statistics and the averaged gradients"""
    from step_amd.net.classifier import Classifier
    from tests.util import assert_close
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.replace("__ROOT__", ROOT).replace("__OUT__", str(tmp_path)))
    env = dict(os.environ, STEP_DIST_BACKEND="gloo", STEP_LOCAL_DEVICE="0", OMP_NUM_THREADS="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0", STEP_TEST_SYNCBN="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["flat"], r1["flat"])
    for k in ("rm", "rv", "dbn_rm"):
        assert torch.equal(r0[k], r1[k]), k                       # both ranks track the same statistics

    D = "cuda:0"
    g = torch.Generator().manual_seed(5)
    X = torch.randn(12, 3, 75, 16, 1, generator=g) * 0.5
    Y = torch.randint(0, 4, (12,), generator=g)
    m = Classifier(3, 4, {"strategy": "spatial"})
    m.load_state_dict(ckpt("ckpt_classifier_real_only_epoch395"))
    m = m.to(D).train()
    o, _ = m(X.to(D))
    F.cross_entropy(o, Y.to(D)).backward()
    want = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
    assert_close(torch.cat([r0["out"], r1["out"]]), o.detach().cpu(), 1e-5, "outputs")
    assert_close(r0["rm"], m.st_gcn_networks[0].tcn[0].running_mean.cpu(), 1e-6, "running_mean")
    assert_close(r0["rv"], m.st_gcn_networks[2].tcn[3].running_var.cpu(), 1e-6, "running_var")
    assert_close(r0["dbn_rm"], m.data_bn.running_mean.cpu(), 1e-6, "data_bn running_mean")
    # per parameter: the mean of the two ranks' gradients is the single-device gradient
    off = 0
    from tests.util import feeds_train_bn
    for name, p in m.named_parameters():
        n = p.numel()
        a, b = r0["flat"][off:off + n], want[off:off + n]
        off += n
        if feeds_train_bn(name):
            continue                                              # mathematically zero: rounding noise on both sides
        assert_close(a, b, 1e-4, name)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` outside torchrun starts its own ranks (a child torchrun; here both ranks share
    the one GPU in rehearsal mode) and prints rank 0's JSON line with n_gpus = 2"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "64", "--no-cpu", "--no-kernels", "--no-infer", "--no-bf16"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128 and line["value"] > 0
    if torch.cuda.device_count() < 2:
        assert line["rehearsal"]


_WORKER2 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '__ROOT__')
from step_amd import parallel
from tests import dist_cases
kind = os.environ["STEP_TEST_KIND"]
rank, local, world = parallel.init_from_env()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
m = dist_cases.build(kind).to(dev).train()
parallel.broadcast_parameters(m)
sync = os.environ.get("STEP_TEST_SYNCBN") == "1"
parallel.sync_batchnorm(sync)
calls = []
real = parallel._all_reduce_sum
parallel._all_reduce_sum = lambda flat, group=None: (calls.append(flat.numel()), real(flat, group))[1]
X, Y, extra = dist_cases.batch(kind)
a, b = parallel.shard_batch(dist_cases.NB, rank, world)
bucket = parallel.GradBucket(m.parameters())
bucket.zero_grad()
loss, out = dist_cases.loss_of(kind, m, X, Y, extra, a, b, dev)
loss.backward()
bucket.all_reduce_mean()
torch.cuda.synchronize()
torch.save({"flat": bucket.flat.cpu(), "numel": bucket.numel, "out": out.detach().cpu(), "calls": calls,
            "loss": float(loss)}, '__OUT__/rank%d.pt' % rank)
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank%d-ok\n" % rank); sys.stdout.flush()
"""


def _two_ranks(tmp_path, kind, syncbn, port):
    script = tmp_path / "worker2.py"
    script.write_text(_WORKER2.replace("__ROOT__", ROOT).replace("__OUT__", str(tmp_path)))
    env = dict(os.environ, STEP_DIST_BACKEND="gloo", STEP_LOCAL_DEVICE="0", OMP_NUM_THREADS="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0", STEP_TEST_KIND=kind, STEP_TEST_SYNCBN="1" if syncbn else "0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return (torch.load(tmp_path / "rank0.pt", weights_only=True), torch.load(tmp_path / "rank1.pt", weights_only=True))


@pytest.mark.parametrize("kind,numel,port", [("cvae", 876484, 29541), ("ras_bf16", 720388, 29543)])
def test_two_rank_bucket_other_models(tmp_path, kind, numel, port):
    """the flat gradient bucket on the CVAE (876 484 parameters) and on the real_and_synth classifier
    in its bf16-operand mode (BASELINE configs[2], the 8-GPU training config): the averaged bucket equals the mean of
    the two shard gradients a single process computes, bit for bit"""
    from step_amd import parallel
    from tests import dist_cases
    r0, r1 = _two_ranks(tmp_path, kind, False, port)
    assert r0["numel"] == numel
    assert torch.equal(r0["flat"], r1["flat"])
    assert r0["calls"] == [numel]                                  # ONE collective per step: the gradient bucket
    D = "cuda:0"
    X, Y, extra = dist_cases.batch(kind)
    flats = []
    for rank in range(2):
        m = dist_cases.build(kind).to(D).train()
        a, b = parallel.shard_batch(dist_cases.NB, rank, 2)
        loss, _ = dist_cases.loss_of(kind, m, X, Y, extra, a, b, D)
        loss.backward()
        flats.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                for p in m.parameters() if p.requires_grad]).cpu())
    want = (flats[0] + flats[1]) / 2
    assert torch.equal(r0["flat"], want), float((r0["flat"] - want).abs().max())


def test_two_ranks_hybrid_with_synchronised_batchnorm(tmp_path):
    """classifier_hybrid under sync_batchnorm: the head's BatchNorm1d on (N, 64 + F) rows (data_bn2,
    classifier_hybrid/net/classifier.py:46,83) is synchronised like the 4-D layers - 2 ranks x 6 gaits give what one
    device gives on the 12; and the collectives of a step are counted: BatchNorms whose inputs exist at the same time
    share one all-reduce."""
    from tests import dist_cases
    from tests.util import assert_close, feeds_train_bn
    r0, r1 = _two_ranks(tmp_path, "hybrid", True, 29545)
    assert torch.equal(r0["flat"], r1["flat"])
    # forward: data_bn1 | [bn1] [bn2] | [bn1 + bnr] [bn2] | [bn1 + bnr] [bn2] | data_bn2 = 1 + 2 + 2 + 2 + 1 = 8 collectives
    # for 11 BatchNorms; backward: data_bn2 | [bn2 + bnr] [bn1] | [bn2 + bnr] [bn1] | [bn2] [bn1] = 7 (data_bn1 feeds no
    # gradient further back: dgamma / dbeta are local); then the gradient bucket
    assert len(r0["calls"]) == 8 + 7 + 1, r0["calls"]
    D = "cuda:0"
    X, Y, extra = dist_cases.batch("hybrid")
    m = dist_cases.build("hybrid").to(D).train()
    loss, out = dist_cases.loss_of("hybrid", m, X, Y, extra, 0, dist_cases.NB, D)
    loss.backward()
    assert_close(torch.cat([r0["out"], r1["out"]]), out.detach().cpu(), 1e-5, "logits")
    off = 0
    for name, p in m.named_parameters():
        n = p.numel()
        a = r0["flat"][off:off + n]
        off += n
        if feeds_train_bn(name):
            continue
        assert_close(a, p.grad.reshape(-1).cpu(), 2e-4, name)
