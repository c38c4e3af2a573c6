"""GPU: every HIP entry point against the CPU oracle on seeded inputs (fp32, tolerance 1e-4
relative to the tensor's max magnitude unless stated: north_star's fp32 bar)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import stgcn, graph_adjacency
from tests.util import assert_close, assert_grad_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def A_spatial(dense=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = torch.tensor(graph_adjacency("spatial"), dtype=torch.float32)
    if dense:
        return torch.rand(3, 16, 16, generator=g) - 0.3
    return A * (torch.rand(3, 16, 16, generator=g) * 2 - 0.5)


# --------------------------------------------------------------------------- tcn
@pytest.mark.parametrize("C,Co,Kt,stride,T,N", [
    (32, 32, 75, 1, 40, 2), (64, 64, 75, 1, 23, 2), (64, 64, 75, 2, 41, 2), (32, 32, 75, 2, 30, 1),
    (3, 3, 75, 1, 20, 2), (32, 32, 9, 1, 50, 2), (64, 64, 9, 2, 37, 2), (64, 64, 75, 1, 240, 1),
    # wide layers (the reference's commented-out 128 / 256 blocks, classifier.py:42-49): output-row groups
    (128, 128, 75, 1, 23, 2), (128, 128, 75, 2, 41, 1), (256, 256, 75, 1, 21, 1), (256, 256, 75, 2, 30, 1),
    (128, 128, 9, 1, 50, 1), (256, 256, 9, 2, 37, 1),
])
def test_tcn_forward_and_epilogue(C, Co, Kt, stride, T, N):
    from step_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, T, 16, generator=g)
    w = torch.randn(Co, C, Kt, 1, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    To = (T - 1) // stride + 1
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = stgcn.temporal_conv(a, w, b, stride)
    d = dev()
    wp = ops.tcn_pack(w.to(d), Co, C, Kt, stride, 0)
    out = ops.tcn_conv(x.to(d), sc.to(d), sh.to(d), True, wp, b.to(d), None, None, None, None, None,
                       False, N, C, Co, T, To, Kt, stride, 0)
    assert_close(out, ref, TOL, "tcn raw")
    # fused eval epilogue: BN2 affine + residual affine + ReLU
    osc, osh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2
    res = torch.randn(N, Co, To, 16, generator=g)
    rsc, rsh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2
    v = lambda t: t.view(1, -1, 1, 1)
    ref2 = torch.relu(ref * v(osc) + v(osh) + res * v(rsc) + v(rsh))
    out2 = ops.tcn_conv(x.to(d), sc.to(d), sh.to(d), True, wp, b.to(d), osc.to(d), osh.to(d),
                        res.to(d), rsc.to(d), rsh.to(d), True, N, C, Co, T, To, Kt, stride, 0)
    assert_close(out2, ref2, TOL, "tcn fused")


@pytest.mark.parametrize("C,Kt,stride,T,N", [(32, 75, 1, 33, 2), (64, 75, 2, 41, 2), (64, 75, 2, 40, 1),
                                             (32, 75, 2, 21, 2), (64, 9, 2, 30, 2), (3, 75, 1, 16, 2),
                                             (128, 75, 1, 33, 2), (128, 75, 2, 41, 1), (256, 75, 1, 20, 1),
                                             (256, 75, 2, 30, 1)])
def test_tcn_dgrad_and_wgrad(C, Kt, stride, T, N):
    from step_amd import ops
    g = torch.Generator().manual_seed(2)
    y = torch.randn(N, C, T, 16, generator=g)
    w = (torch.randn(C, C, Kt, 1, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    To = (T - 1) // stride + 1
    a = torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    u = stgcn.temporal_conv(a, w, b, stride)
    du = torch.randn(u.shape, generator=g)
    u.backward(du)
    d = dev()
    wpd = ops.tcn_pack(w.detach().to(d), C, C, Kt, stride, 2)
    da = ops.tcn_conv(du.to(d), None, None, False, wpd, None, None, None, None, None, None, False,
                      N, C, C, T, To, Kt, stride, 2)
    assert_close(da, a.grad, TOL, "dgrad")
    dw, db = ops.tcn_wgrad(du.to(d), y.to(d), sc.to(d), sh.to(d), True, N, C, C, T, To, Kt, stride,
                           False, w.shape)
    assert_close(dw, w.grad, TOL, "wgrad")
    assert_close(db, b.grad, TOL, "bgrad")


@pytest.mark.parametrize("Ci,Co,Kt,T,bf", [(32, 32, 75, 24, False), (32, 64, 9, 30, False), (64, 3, 75, 20, False),
                                            (32, 32, 75, 24, "split"), (32, 64, 75, 75, "split")])
def test_tcn_transposed(Ci, Co, Kt, T, bf):
    """ConvTranspose2d twin (CVAE_stgcn.py:359-365): forward, dgrad, wgrad.  In st_gctn the
    temporal conv is always square (Co x Co); Ci here is ignored except as a label.  (bf = "split": the
    split-operand kernels, same bar.)"""
    from step_amd import ops
    C, N = Co, 2
    g = torch.Generator().manual_seed(3)
    a = torch.randn(N, C, T, 16, generator=g).requires_grad_(True)
    w = (torch.randn(C, C, Kt, 1, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    u = stgcn.temporal_conv(a, w, b, 1, transposed=True)
    du = torch.randn(u.shape, generator=g)
    u.backward(du)
    d = dev()
    wp = ops.tcn_pack(w.detach().to(d), C, C, Kt, 1, 1, bf16=bf)
    out = ops.tcn_conv(a.detach().to(d), None, None, False, wp, b.detach().to(d), None, None, None, None,
                       None, False, N, C, C, T, T, Kt, 1, 1, bf16=bf)
    assert_close(out, u, TOL, "convT fwd")
    wpd = ops.tcn_pack(w.detach().to(d), C, C, Kt, 1, 3, bf16=bf)
    da = ops.tcn_conv(du.to(d), None, None, False, wpd, None, None, None, None, None, None, False,
                      N, C, C, T, T, Kt, 1, 3, bf16=bf)
    assert_close(da, a.grad, TOL, "convT dgrad")
    dw, db = ops.tcn_wgrad(du.to(d), a.detach().to(d), None, None, False, N, C, C, T, T, Kt, 1, True, w.shape, bf16=bf)
    assert_close(dw, w.grad, TOL, "convT wgrad")
    assert_close(db, b.grad, TOL, "convT bgrad")


# --------------------------------------------------------------------------- gcn
@pytest.mark.parametrize("Ci,Co,T,N,transposed,dense", [
    (3, 32, 20, 3, False, False), (32, 64, 17, 2, False, False), (64, 64, 9, 2, False, False),
    (7, 64, 12, 2, False, False), (64, 3, 11, 2, True, False), (32, 32, 8, 2, True, False),
    (32, 64, 10, 2, False, True), (64, 32, 24, 1, False, False),
    (4, 32, 13, 5, True, True), (2, 32, 9, 300, False, False), (1, 32, 7, 2, False, False),   # first-block form
    (3, 64, 11, 2, False, False),
    # wide layers (north_star's C = 128 / 256 bench layers; classifier.py:42-49 leaves these blocks commented out)
    (64, 128, 9, 2, False, False), (128, 128, 7, 3, False, False), (128, 256, 6, 2, False, True),
    (256, 256, 5, 2, False, False), (256, 128, 4, 1, True, False),
])
def test_gcn_fwd_bwd(Ci, Co, T, N, transposed, dense):
    from step_amd.net.utils.tgcn import GcnFn
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, Ci, T, 16, generator=g)
    A = A_spatial(dense)
    w = torch.randn((Ci, 3 * Co, 1, 1) if transposed else (3 * Co, Ci, 1, 1), generator=g) * 0.2
    b = torch.randn(3 * Co, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, A, w, b)]
    y = stgcn.conv_temporal_graphical(leaves[0], leaves[1], leaves[2], leaves[3], transposed)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = dev()
    gl = [t.clone().to(d).requires_grad_(True) for t in (x, A, w, b)]
    nnz = 16 if dense else 4
    yg = GcnFn.apply(gl[0], gl[1], gl[2], gl[3], transposed, nnz)
    assert_close(yg, y, TOL, "gcn y")
    yg.backward(gy.to(d))
    assert_close(gl[0].grad, leaves[0].grad, TOL, "dx")
    assert_close(gl[2].grad, leaves[2].grad, TOL, "dw")
    assert_close(gl[3].grad, leaves[3].grad, TOL, "db")
    supp = (A != 0).float()
    assert_close(gl[1].grad.cpu() * supp, leaves[1].grad * supp, TOL, "dA on supp(A)")


# ---------------------------------------------------------------------------- bn
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,C,T", [(3, 5, 7), (300, 3, 11), (2, 64, 40)])
def test_bn_train_prepare_and_backward(mode, N, C, T):
    from step_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(N, C, T, 16, generator=g) * 2 + 0.7)
    G = C * 16 if mode else C
    gamma, beta = torch.rand(G, generator=g) + 0.5, torch.randn(G, generator=g)
    P = {"bn.weight": gamma.clone().requires_grad_(True), "bn.bias": beta.clone().requires_grad_(True),
         "bn.running_mean": torch.randn(G, generator=g), "bn.running_var": torch.rand(G, generator=g) + 0.5}
    rm0, rv0 = P["bn.running_mean"].clone(), P["bn.running_var"].clone()
    xr = x.clone().requires_grad_(True)
    if mode:
        yr = stgcn.data_bn(xr, P, "bn", True)
    else:
        yr = stgcn.bn2d(xr, P, "bn", True)
    out_ref = torch.relu(yr)
    gy = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gy)
    d = dev()
    rm, rv = rm0.to(d), rv0.to(d)
    mean, invstd, scale, shift = ops.bn_train_prepare(x.to(d), mode, gamma.to(d), beta.to(d), rm, rv)
    assert_close(rm, P["bn.running_mean"], 1e-5, "running_mean")
    assert_close(rv, P["bn.running_var"], 1e-5, "running_var")
    out = ops.affine_act(x.to(d), scale, shift, None, None, None, True, mode)
    assert_close(out, out_ref, TOL, "bn+relu")
    dx, dg, db = ops.bn_backward(gy.to(d), x.to(d), None, scale, shift, 2, mean, invstd, gamma.to(d), True, mode)
    assert_close(dx, xr.grad, 2e-4, "dx")
    assert_close(dg, P["bn.weight"].grad, 2e-4, "dgamma")
    assert_close(db, P["bn.bias"].grad, 2e-4, "dbeta")
    # mask from a saved output tensor gives the same thing
    dx2, _, _ = ops.bn_backward(gy.to(d), x.to(d), out, None, None, 1, mean, invstd, gamma.to(d), True, mode)
    assert torch.equal(dx, dx2)


def test_heads():
    from step_amd import ops
    g = torch.Generator().manual_seed(6)
    d = dev()
    x = torch.randn(5, 64, 30, 16, generator=g)
    xg = x.clone().to(d).requires_grad_(True)
    f = ops.PoolFn.apply(xg)
    assert_close(f, x.mean(dim=(2, 3)), 1e-5, "pool")
    w = torch.randn(4, 64, 1, 1, generator=g)
    b = torch.randn(4, generator=g)
    wg, bg = w.clone().to(d).requires_grad_(True), b.clone().to(d).requires_grad_(True)
    p = ops.SoftmaxFn.apply(ops.DenseFn.apply(f, wg, bg, False))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pr = torch.softmax(xr.mean(dim=(2, 3)) @ wr.view(4, 64).t() + br, dim=1)
    assert_close(p, pr, 1e-5, "softmax")
    lab = torch.tensor([0, 1, 2, 3, 1])
    F.cross_entropy(pr, lab).backward()
    F.cross_entropy(p, lab.to(d)).backward()
    assert_close(xg.grad, xr.grad, 1e-4, "dx")
    assert_close(wg.grad, wr.grad, 1e-4, "dw")
    assert_close(bg.grad, br.grad, 1e-4, "db")


@pytest.mark.parametrize("N,C,T,training", [(3, 5, 7, True), (300, 32, 11, True), (2, 64, 40, False)])
def test_bn_backward_tail_pair_equals_two_passes(N, C, T, training):
    """block tail with a residual conv: one fused reduce + apply for BN2 and the residual BN, bit-identical
    to the two separate step_bn_bwd_reduce / step_bn_bwd_apply passes it replaces"""
    from step_amd import ops
    g = torch.Generator().manual_seed(6)
    d = dev()
    u, r = torch.randn(N, C, T, 16, generator=g).to(d), torch.randn(N, C, T, 16, generator=g).to(d)
    out = torch.relu(torch.randn(N, C, T, 16, generator=g)).to(d)
    dout = torch.randn(N, C, T, 16, generator=g).to(d)
    w2, wr = (torch.rand(C, generator=g) + 0.5).to(d), (torch.rand(C, generator=g) + 0.5).to(d)
    b, rm, rv = torch.zeros(C, device=d), torch.zeros(C, device=d), torch.ones(C, device=d)
    m2, i2, _, _ = ops.bn_train_prepare(u, 0, w2, b, rm.clone(), rv.clone())
    mr, ir, _, _ = ops.bn_train_prepare(r, 0, wr, b, rm.clone(), rv.clone())
    ref_u = ops.bn_backward(dout, u, out, None, None, 1, m2, i2, w2, training, 0)
    ref_r = ops.bn_backward(dout, r, out, None, None, 1, mr, ir, wr, training, 0)
    got = ops.bn_backward_tail_pair(dout, out, u, r, m2, i2, w2, mr, ir, wr, training)
    for a, b_ in zip(got, tuple(ref_u) + tuple(ref_r)):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("strategy,Ci,Co,T", [("uniform", 32, 64, 19), ("distance", 64, 64, 9), ("distance", 3, 32, 12),
                                               ("uniform", 64, 32, 24)])
def test_gcn_other_graph_strategies(strategy, Ci, Co, T):
    """Graph strategies 'uniform' (K = 1) and 'distance' (K = 2), net/utils/graph.py:71-82: fewer partitions than
    the kernels' three slots"""
    from step_amd.net.utils.tgcn import GcnFn
    g = torch.Generator().manual_seed(8)
    from step_amd.net.utils.graph import Graph
    A = torch.tensor(Graph(strategy=strategy).A, dtype=torch.float32)        # the product's own Graph
    K = A.shape[0]
    A = A * (torch.rand(K, 16, 16, generator=g) + 0.5)
    x = torch.randn(2, Ci, T, 16, generator=g)
    w = torch.randn(K * Co, Ci, 1, 1, generator=g) * 0.2
    b = torch.randn(K * Co, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, A, w, b)]
    y = stgcn.conv_temporal_graphical(leaves[0], leaves[1], leaves[2], leaves[3], False)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = dev()
    gl = [t.clone().to(d).requires_grad_(True) for t in (x, A, w, b)]
    yg = GcnFn.apply(gl[0], gl[1], gl[2], gl[3], False, 16)
    assert_close(yg, y, TOL, "gcn y")
    yg.backward(gy.to(d))
    assert_close(gl[0].grad, leaves[0].grad, TOL, "dx")
    assert_close(gl[2].grad, leaves[2].grad, TOL, "dw")
    assert_close(gl[3].grad, leaves[3].grad, TOL, "db")
    supp = (A != 0).float()
    assert_close(gl[1].grad.cpu() * supp, leaves[1].grad * supp, TOL, "dA on supp(A)")


@pytest.mark.parametrize("Ci,Co,T,N,stride,K", [(3, 32, 21, 37, 2, 3), (4, 32, 8, 3, 1, 2), (3, 64, 10, 5, 2, 3)])
def test_first_block_forms_agree(Ci, Co, T, N, stride, K, monkeypatch):
    """The fused first-block kernels (Ci <= 4) against the general tile kernels on a case no model here reaches:
    data_bn scale/shift + residual conv at stride 2 (and K = 2), through the C-ABI."""
    from step_amd import _lib
    d = dev()
    g = torch.Generator().manual_seed(11)
    A = (A_spatial(True)[:K]).contiguous().to(d)
    x = torch.randn(N, Ci, T, 16, generator=g).to(d)
    isc, ish = (torch.rand(16 * Ci, generator=g) + 0.5).to(d), torch.randn(16 * Ci, generator=g).to(d)
    w, b = (torch.randn(K * Co, Ci, generator=g) * 0.3).to(d), torch.randn(K * Co, generator=g).to(d)
    rw, rb = (torch.randn(Co, Ci, generator=g) * 0.3).to(d), torch.randn(Co, generator=g).to(d)
    To = (T - 1) // stride + 1
    dy, dr = torch.randn(N, Co, T, 16, generator=g).to(d), torch.randn(N, Co, To, 16, generator=g).to(d)
    nb = _lib.query("step_gcn_bwd_ws_bytes", N, Ci, Co, K, T)
    ws = torch.empty(nb, dtype=torch.uint8, device=d)

    def run():
        y, r = torch.empty(N, Co, T, 16, device=d), torch.empty(N, Co, To, 16, device=d)
        _lib.call("step_gcn_fwd", x, isc, ish, w, b, 0, A, 16, y, rw, rb, 0, r, N, Ci, Co, K, T, stride)
        out = [torch.empty_like(t) for t in (x, w, b, A, rw, rb)]
        _lib.call("step_gcn_bwd", dy, dr, x, isc, ish, w, b, 0, A, 16, rw, 0, *out, N, Ci, Co, K, T, stride, ws, nb)
        return [y, r] + out

    new = run()
    monkeypatch.setenv("STEP_GCN_FWD_V1", "1")
    monkeypatch.setenv("STEP_GCN_BWD_V1", "1")
    old = run()
    for name, a, o in zip(("y", "res_out", "dx", "dw", "db", "dA", "dres_w", "dres_b"), new, old):
        assert_close(a, o.cpu(), 2e-5, name)


@pytest.mark.parametrize("Ci,Co,T,N,stride,res,tr,K", [
    (64, 64, 75, 9, 1, False, False, 3),      # block 2 of the classifier (identity residual: no residual conv here)
    (32, 64, 75, 5, 2, True, False, 3),       # block 1: residual conv at stride 2, odd T
    (32, 64, 10, 3, 1, True, False, 3),       # residual conv at stride 1 (narrower channel group)
    (64, 32, 7, 4, 1, True, True, 3),         # st_gctn (w as (Ci, K Co)), T % 4 != 0
    (16, 32, 1, 6, 1, False, False, 2),       # one frame, K = 2
    (128, 256, 6, 2, 2, True, False, 3),      # wide: four channel groups
    (64, 96, 5, 2, 1, False, False, 1),       # Co % 64 != 0
    (256, 64, 4, 2, 1, True, False, 3),       # W slice + residual conv's do not fit the LDS: the fp32 kernels run
])
def test_gcn_fwd_split_operands_vs_oracle(Ci, Co, T, N, stride, res, tr, K):
    """step_gcn_fwd_split (x, A, w and the aggregated z as bf16 hi + lo on the bf16 matrix cores) holds the fp32
    bar against the oracle (tgcn.py:58-67 / :121-130 and the residual conv classifier.py:175-179), through the C-ABI"""
    from step_amd import _lib
    d = dev()
    g = torch.Generator().manual_seed(21)
    A = (A_spatial(True)[:K] * (torch.rand(K, 16, 16, generator=g) + 0.5)).contiguous()
    x = torch.randn(N, Ci, T, 16, generator=g)
    w = torch.randn(K * Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    b = torch.randn(K * Co, generator=g)
    wt = w.view(K * Co, Ci).t().contiguous().view(Ci, K * Co, 1, 1) if tr else w
    ref = stgcn.conv_temporal_graphical(x, A, wt, b, tr)
    rw = rb = ref_r = r = None
    To = (T - 1) // stride + 1
    if res:
        rw, rb = torch.randn(Co, Ci, generator=g) / Ci ** 0.5, torch.randn(Co, generator=g)
        ref_r = F.conv2d(x, rw.view(Co, Ci, 1, 1), rb, stride=(stride, 1))
        r = torch.full((N, Co, To, 16), float("nan"), device=d)
    y = torch.full((N, Co, T, 16), float("nan"), device=d)
    rwd = None if rw is None else (rw.t().contiguous() if tr else rw).to(d)
    _lib.call("step_gcn_fwd_split", x.to(d), None, None, wt.to(d), b.to(d), int(tr), A.to(d), 16, y,
              rwd, None if rb is None else rb.to(d), int(tr), r, N, Ci, Co, K, T, stride)
    assert_close(y, ref, TOL, "y")
    if res:
        assert_close(r, ref_r, TOL, "res_out")
    # and it is not the fp32 kernel under another name: the results differ in the last bits
    y32 = torch.empty_like(y)
    r32 = None if r is None else torch.empty_like(r)
    _lib.call("step_gcn_fwd", x.to(d), None, None, wt.to(d), b.to(d), int(tr), A.to(d), 16, y32,
              rwd, None if rb is None else rb.to(d), int(tr), r32, N, Ci, Co, K, T, stride)
    assert_close(y, y32.cpu(), 2e-5, "y vs fp32 kernel")
    assert torch.equal(y, y32) == (Ci == 256 and res)


@pytest.mark.parametrize("Ci,Co,T,N,stride,res,tr,K", [
    (64, 64, 75, 9, 1, "identity", False, 3),     # block 2 of the classifier
    (32, 64, 75, 5, 2, "conv", False, 3),         # block 1: residual conv at stride 2, odd T
    (32, 32, 10, 3, 1, "conv", False, 3),
    (64, 32, 7, 4, 1, "conv", True, 3),           # st_gctn layouts
    (32, 32, 1, 6, 1, None, False, 2),            # one frame, K = 2
    (64, 64, 5, 2, 1, None, False, 1),
])
def test_gcn_bwd_split_operands_vs_oracle(Ci, Co, T, N, stride, res, tr, K):
    """step_gcn_bwd_split (dz = W^T dy, dA += x^T dz, z = x A, dW += dy z^T, dWr += dres x^T on the bf16 matrix cores,
    operands as bf16 hi + lo) holds the fp32 bar against autograd of the oracle (tgcn.py:58-67 / :121-130, residual conv
    classifier.py:175-179, identity residual :170-171), through the C-ABI"""
    from step_amd import _lib
    d = dev()
    g = torch.Generator().manual_seed(22)
    A = (A_spatial(True)[:K] * (torch.rand(K, 16, 16, generator=g) + 0.5)).contiguous()
    x = torch.randn(N, Ci, T, 16, generator=g)
    w = torch.randn(K * Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    b = torch.randn(K * Co, generator=g)
    wt = w.view(K * Co, Ci).t().contiguous().view(Ci, K * Co, 1, 1) if tr else w
    To = (T - 1) // stride + 1
    dy = torch.randn(N, Co, T, 16, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, A, wt, b)]
    total = (stgcn.conv_temporal_graphical(*leaves, tr) * dy).sum()
    rw = rb = dr = dout = out = None
    if res == "conv":
        rw, rb = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).requires_grad_(True), torch.randn(Co, generator=g).requires_grad_(True)
        dr = torch.randn(N, Co, To, 16, generator=g)
        total = total + (F.conv2d(leaves[0], rw.view(Co, Ci, 1, 1), rb, stride=(stride, 1)) * dr).sum()
    total.backward()
    ref_dx = leaves[0].grad
    if res == "identity":
        dout, out = torch.randn(N, Ci, T, 16, generator=g), torch.randn(N, Ci, T, 16, generator=g)
        ref_dx = ref_dx + dout * (out > 0)
    nb = _lib.query("step_gcn_bwd_ws_bytes", N, Ci, Co, K, T)
    ws = torch.empty(nb, dtype=torch.uint8, device=d)
    dev_ = lambda t: None if t is None else t.detach().to(d)
    rwd = None if rw is None else (rw.detach().t().contiguous() if tr else rw.detach()).to(d)
    dx, dw, db, dA = (torch.full(t.shape, float("nan"), device=d) for t in (x, wt, b, A))
    drw = None if rw is None else torch.full(rwd.shape, float("nan"), device=d)
    drb = None if rw is None else torch.full((Co,), float("nan"), device=d)

    def run(fn, outs):
        _lib.call(fn, dev_(dy), dev_(dr), dev_(x), None, None, dev_(wt), dev_(b), int(tr), dev_(A), 16, rwd, int(tr),
                  dev_(dout), dev_(out), *outs, N, Ci, Co, K, T, stride, ws, nb)
    run("step_gcn_bwd_split", (dx, dw, db, dA, drw, drb))
    assert_close(dx, ref_dx, TOL, "dx")
    assert_close(dw, leaves[2].grad, TOL, "dw")
    assert_close(db, leaves[3].grad, TOL, "db")
    supp = (A != 0).float()
    assert_close(dA.cpu() * supp, leaves[1].grad * supp, TOL, "dA on supp(A)")
    if rw is not None:
        assert_close(drw, rw.grad.t() if tr else rw.grad, TOL, "dres_w")
        assert_close(drb, rb.grad, TOL, "dres_b")
    # against the fp32 kernels: close, and not the same code under another name
    f32 = [torch.empty_like(t) for t in (dx, dw, db, dA)] + [None if drw is None else torch.empty_like(drw),
                                                              None if drb is None else torch.empty_like(drb)]
    run("step_gcn_bwd_idres", f32)
    assert_close(dx, f32[0].cpu(), 2e-5, "dx vs fp32 kernel")
    assert_close(dw, f32[1].cpu(), 2e-5, "dw vs fp32 kernel")
    assert not torch.equal(dw, f32[1])


@pytest.mark.parametrize("C,Kt,T,transposed,res", [(32, 75, 75, True, "identity"), (64, 75, 11, True, "conv"),
                                                    (32, 9, 40, False, None), (64, 75, 240, False, "identity")])
def test_tcn_const_input(C, Kt, T, transposed, res):
    """step_tcn_const_weights / step_tcn_const_fwd against the convolution of the materialised repeat
    (CVAE_stgcn.py:209 followed by tcn[2] of decoder[0], :359-365), eval epilogue included"""
    from step_amd import _lib
    d = dev()
    g = torch.Generator().manual_seed(12)
    N = 5
    a0 = torch.relu(torch.randn(N, C, 16, generator=g))
    w = torch.randn(C, C, Kt, 1, generator=g) * 0.05
    b = torch.randn(C, generator=g)
    s2, h2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    a = a0[:, :, None, :].expand(N, C, T, 16)
    pad = ((Kt - 1) // 2, 0)
    u = F.conv_transpose2d(a, w, b, padding=pad) if transposed else F.conv2d(a, w, b, padding=pad)
    ref = u * s2.view(1, C, 1, 1) + h2.view(1, C, 1, 1)
    r0 = sr = hr = None
    if res is not None:
        r0 = torch.randn(N, C, 16, generator=g)
        rr = r0[:, :, None, :]
        if res == "conv":
            sr, hr = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
            rr = rr * sr.view(1, C, 1, 1) + hr.view(1, C, 1, 1)
        ref = ref + rr
    ref = torch.relu(ref)
    wcum = torch.empty(T, C, C, device=d)
    _lib.call("step_tcn_const_weights", w.to(d), wcum, C, C, Kt, T, int(transposed))
    out = torch.empty(N, C, T, 16, device=d)
    dv = lambda t: None if t is None else t.to(d)
    _lib.call("step_tcn_const_fwd", a0.to(d), wcum, b.to(d), s2.to(d), h2.to(d), dv(r0), dv(sr), dv(hr), 1, out,
              N, C, C, T)
    assert_close(out, ref, TOL, "const-input tcn")


@pytest.mark.parametrize("C,T,N", [(64, 9, 3), (32, 20, 2), (128, 6, 2), (16, 7, 2)])
def test_gcn_bwd_identity_residual_in_the_same_launch(C, T, N):
    """step_gcn_bwd_idres = step_gcn_bwd followed by dx += dout * [out > 0] (relu(tcn(x) + x), classifier.py:189-191):
    fused into the data kernel's store at 32 / 64 channels, the elementwise kernel after it elsewhere"""
    from step_amd import _lib
    d = dev()
    g = torch.Generator().manual_seed(13)
    A = A_spatial(False).to(d)
    x = torch.randn(N, C, T, 16, generator=g).to(d)
    w, b = (torch.randn(3 * C, C, generator=g) * 0.2).to(d), torch.randn(3 * C, generator=g).to(d)
    dy = torch.randn(N, C, T, 16, generator=g).to(d)
    dout, out = torch.randn(N, C, T, 16, generator=g).to(d), torch.randn(N, C, T, 16, generator=g).to(d)
    nb = _lib.query("step_gcn_bwd_ws_bytes", N, C, C, 3, T)
    ws = torch.empty(nb, dtype=torch.uint8, device=d)
    ref = [torch.empty_like(t) for t in (x, w, b, A)]
    _lib.call("step_gcn_bwd", dy, None, x, None, None, w, b, 0, A, 4, None, 0, ref[0], ref[1], ref[2], ref[3],
              None, None, N, C, C, 3, T, 1, ws, nb)
    ref[0] = ref[0] + dout * (out > 0)
    got = [torch.empty_like(t) for t in (x, w, b, A)]
    _lib.call("step_gcn_bwd_idres", dy, None, x, None, None, w, b, 0, A, 4, None, 0, dout, out, got[0], got[1],
              got[2], got[3], None, None, N, C, C, 3, T, 1, ws, nb)
    for name, a, r in zip(("dx", "dw", "db", "dA"), got, ref):
        assert_close(a, r.cpu(), 1e-6, name)


@pytest.mark.parametrize("C,stride,T,N,bf16", [(64, 2, 25, 5, False), (32, 1, 30, 3, False), (64, 1, 18, 4, True),
                                                (128, 2, 14, 3, False), (32, 2, 21, 7, True)])
def test_wgrad_two_launch_tap_split(C, stride, T, N, bf16, monkeypatch):
    """the sliding-window wgrad as two launches (taps 0..59 as 4 x 5 per block, taps 60..74 as 4 x 4: no padded tap
    group) against the single padded launch, which the oracle tests above pin; the geometry picks it by cost, here
    both are forced"""
    from step_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(14)
    To = (T - 1) // stride + 1
    y = torch.randn(N, C, T, 16, generator=g).to(d)
    du = torch.randn(N, C, To, 16, generator=g).to(d)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(d), torch.randn(C, generator=g).to(d)
    res = {}
    for mode in ("1", "2"):
        monkeypatch.setenv("STEP_WS_LAUNCHES", mode)
        res[mode] = ops.tcn_wgrad(du, y, sc, sh, True, N, C, C, T, To, 75, stride, False, (C, C, 75, 1), bf16=bf16)
    assert_close(res["2"][0], res["1"][0].cpu(), 1e-5, "dw")
    assert_close(res["2"][1], res["1"][1].cpu(), 1e-6, "db")


@pytest.mark.parametrize("C,stride,T,N", [(32, 1, 33, 2), (64, 2, 41, 2), (64, 1, 40, 6), (32, 2, 21, 2), (128, 1, 33, 2),
                                          (64, 2, 240, 16)])
def test_wgrad_split_operands_vs_oracle(C, stride, T, N):
    """step_tcn_wgrad_split (three bf16 MFMAs per product on fp32 operands split into bf16 hi + lo) against the fp32
    oracle at the fp32 bar of the plain kernel (1e-4 of the largest gradient), plus how far it sits from the fp32-MFMA
    kernel: the dropped lo*lo terms are 2^-16 relative per product"""
    from step_amd import ops
    g = torch.Generator().manual_seed(15)
    y = torch.randn(N, C, T, 16, generator=g) * 1.5 + 0.3
    w = (torch.randn(C, C, 75, 1, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    To = (T - 1) // stride + 1
    a = torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    u = stgcn.temporal_conv(a, w, b, stride)
    du = torch.randn(u.shape, generator=g)
    u.backward(du)
    d = dev()
    args = (du.to(d), y.to(d), sc.to(d), sh.to(d), True, N, C, C, T, To, 75, stride, False, w.shape)
    dw, db = ops.tcn_wgrad(*args, bf16="split")
    assert_close(dw, w.grad, TOL, "wgrad (split operands)")
    assert_close(db, b.grad, TOL, "bgrad")
    dw32, _ = ops.tcn_wgrad(*args)
    from tests.util import relerr
    e_split, e_f32 = relerr(dw, w.grad), relerr(dw32, w.grad)
    assert e_split < 2e-5, (e_split, e_f32)             # typically 2e-6 .. 6e-6; the fp32-MFMA kernel: ~1e-6


@pytest.mark.parametrize("C,stride,T,N", [(64, 1, 48, 2), (64, 1, 23, 3), (32, 1, 96, 2), (32, 1, 17, 1), (64, 2, 41, 2),
                                          (64, 2, 240, 1), (32, 1, 240, 1)])
def test_tcn_split_operands_forward_dgrad_vs_oracle(C, stride, T, N):
    """step_tcn_conv_split (fp32 operands as bf16 hi + lo, w_hi*(v_hi + v_lo) + w_lo*v_hi on the bf16 matrix cores)
    against the fp32 oracle at the fp32 bar of the plain kernels: forward with and without the fused epilogue, and
    the input gradient"""
    from step_amd import ops
    from tests.util import relerr
    Kt = 75
    g = torch.Generator().manual_seed(16)
    x = torch.randn(N, C, T, 16, generator=g) * 1.5 + 0.2
    w = torch.randn(C, C, Kt, 1, generator=g) * 0.05
    b = torch.randn(C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    To = (T - 1) // stride + 1
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    ref = stgcn.temporal_conv(a, w, b, stride)
    du = torch.randn(ref.shape, generator=g)
    ref.backward(du)
    d = dev()
    wp = ops.tcn_pack(w.to(d), C, C, Kt, stride, 0, bf16="split")
    out = ops.tcn_conv(x.to(d), sc.to(d), sh.to(d), True, wp, b.to(d), None, None, None, None, None,
                       False, N, C, C, T, To, Kt, stride, 0, bf16="split")
    assert_close(out, ref.detach(), TOL, "split tcn forward")
    assert relerr(out, ref.detach()) < 2e-5
    osc, osh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    res = torch.randn(N, C, To, 16, generator=g)
    rsc, rsh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    v = lambda t: t.view(1, -1, 1, 1)
    ref2 = torch.relu(ref.detach() * v(osc) + v(osh) + res * v(rsc) + v(rsh))
    out2 = ops.tcn_conv(x.to(d), sc.to(d), sh.to(d), True, wp, b.to(d), osc.to(d), osh.to(d),
                        res.to(d), rsc.to(d), rsh.to(d), True, N, C, C, T, To, Kt, stride, 0, bf16="split")
    assert_close(out2, ref2, TOL, "split tcn fused epilogue")
    wpd = ops.tcn_pack(w.to(d), C, C, Kt, stride, 2, bf16="split")
    da = ops.tcn_conv(du.to(d), None, None, False, wpd, None, None, None, None, None, None, False,
                      N, C, C, T, To, Kt, stride, 2, bf16="split")
    assert_close(da, a.grad, TOL, "split dgrad")
    assert relerr(da, a.grad) < 2e-5


@pytest.mark.parametrize("bf16", [False, "split"])
def test_wgrad_nan_input_without_relu_stays_non_finite(bf16):
    """tcn_wgrad_sw.hip applies the activation as max(x * scale + shift, lo) with lo = -inf when in_relu == 0, and
    v_max(NaN, -inf) is -inf: a NaN in the input of a no-ReLU weight gradient must still leave dw non-finite (no
    model path calls the kernel that way; the reference's conv backward would give NaN there)."""
    from step_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(21)
    N, C, T = 2, 64, 20
    y = torch.randn(N, C, T, 16, generator=g)
    y[1, 5, 7, 3] = float("nan")
    du = torch.randn(N, C, T, 16, generator=g)
    sc, sh = torch.ones(C), torch.zeros(C)
    dw, db = ops.tcn_wgrad(du.to(d), y.to(d), sc.to(d), sh.to(d), False, N, C, C, T, T, 75, 1, False, (C, C, 75, 1),
                           bf16=bf16)
    assert not bool(torch.isfinite(dw[:, 5]).all()), "a NaN activation vanished from the weight gradient"
    others = torch.cat((dw[:, :5], dw[:, 6:]), dim=1)
    assert bool(torch.isfinite(others).all()) and bool(torch.isfinite(db).all())


@pytest.mark.parametrize("Ci,Co,T,N,stride,res,tr", [
    (32, 64, 75, 5, 2, True, False), (64, 64, 41, 3, 1, False, False), (64, 32, 29, 4, 1, True, False),
    (32, 32, 13, 4, 1, False, True), (32, 64, 240, 40, 2, True, False), (64, 64, 120, 37, 1, False, False),
    (3, 32, 30, 6, 1, True, False), (32, 64, 9, 3, 1, True, True), (64, 128, 10, 2, 2, True, False)])
def test_gcn_fwd_stats_epilogue(Ci, Co, T, N, stride, res, tr):
    """step_gcn_fwd_stats: outputs bit-equal to step_gcn_fwd's, and [SUM, SUM of squares] per channel of y and of the
    residual conv's output equal to float64 sums of those tensors (shapes with the epilogue form, T % 4 != 0, odd T
    under stride 2, and shapes that take the separate statistics pass: first block, stride-1 residual at 64 wide,
    wide layers); two calls give bit-identical sums"""
    from step_amd import _lib, ops
    d = dev()
    g = torch.Generator().manual_seed(Ci * 1000 + Co + T)
    A = A_spatial().to(d)
    nnz = ops.a_nnz_max(A)
    To = (T - 1) // stride + 1
    x = (torch.randn(N, Ci, T, 16, generator=g) + 0.3).to(d)
    w = (torch.randn(Ci, 3 * Co, generator=g) if tr else torch.randn(3 * Co, Ci, generator=g)).to(d) * 0.2
    b = torch.randn(3 * Co, generator=g).to(d)
    rw = ((torch.randn(Ci, Co, generator=g) if tr else torch.randn(Co, Ci, generator=g)).to(d) * 0.2) if res else None
    rb = torch.randn(Co, generator=g).to(d) if res else None

    def run(stats):
        y = torch.full((N, Co, T, 16), float("nan"), device=d)
        r = torch.full((N, Co, To, 16), float("nan"), device=d) if res else None
        if not stats:
            _lib.call("step_gcn_fwd", x, None, None, w, b, int(tr), A, nnz, y, rw, rb, int(tr), r, N, Ci, Co, 3, T, stride)
            return y, r, None, None
        ys = torch.zeros(2 * Co + 1, dtype=torch.float64, device=d)
        rs = torch.zeros(2 * Co + 1, dtype=torch.float64, device=d) if res else None
        nb = _lib.query("step_gcn_fwd_stats_ws_bytes", Co)
        ws = torch.empty(nb, dtype=torch.uint8, device=d)
        _lib.call("step_gcn_fwd_stats", x, None, None, w, b, int(tr), A, nnz, y, rw, rb, int(tr), r, N, Ci, Co, 3, T,
                  stride, ys, rs, ws, nb)
        return y, r, ys, rs

    y0, r0, _, _ = run(False)
    y1, r1, ys, rs = run(True)
    y2, r2, ys2, rs2 = run(True)
    assert torch.equal(y0, y1) and (not res or torch.equal(r0, r1))
    assert torch.equal(ys, ys2) and (not res or torch.equal(rs, rs2))
    for t, sm, name in ((y1, ys, "y"), (r1, rs, "r")):
        if t is None:
            continue
        td = t.double()
        ref1, ref2 = td.sum(dim=(0, 2, 3)), (td * td).sum(dim=(0, 2, 3))
        got = sm[:2 * Co].view(Co, 2)
        assert float((got[:, 0] - ref1).abs().max()) <= 2e-6 * float(td.abs().sum(dim=(0, 2, 3)).max()), name
        assert float(((got[:, 1] - ref2) / ref2).abs().max()) <= 2e-6, name


@pytest.mark.parametrize("C,stride,T,N,tr", [(32, 1, 33, 3, False), (64, 2, 41, 2, False), (64, 1, 240, 5, False),
                                             (64, 2, 240, 3, False), (32, 1, 24, 2, True), (128, 1, 30, 2, False),
                                             (3, 1, 20, 2, True)])
def test_tcn_conv_stats_epilogue(C, stride, T, N, tr):
    """step_tcn_conv_stats: u bit-equal to step_tcn_conv's, [SUM u, SUM u^2] per channel equal to float64 sums of u
    (ragged T: blocks whose last waves lie past the sequence end; stride 2; ConvTranspose packing; 128 channels on
    two output groups; the 64 -> 3 decoder layer), and bit-identical over two calls"""
    from step_amd import _lib, ops
    d = dev()
    g = torch.Generator().manual_seed(C + T)
    Ci = 64 if C == 3 else C
    Co = C
    To = (T - 1) // stride + 1
    y = torch.randn(N, Ci, T, 16, generator=g).to(d)
    w = (torch.randn(Ci, Co, 75, 1, generator=g) if tr else torch.randn(Co, Ci, 75, 1, generator=g)).to(d) * 0.05
    b = torch.randn(Co, generator=g).to(d)
    sc, sh = (torch.rand(Ci, generator=g) + 0.5).to(d), torch.randn(Ci, generator=g).to(d) * 0.3
    mode = 1 if tr else 0
    wp = ops.tcn_pack(w, Co, Ci, 75, stride, mode)
    ref = ops.tcn_conv(y, sc, sh, True, wp, b, None, None, None, None, None, False, N, Ci, Co, T, To, 75, stride, mode)
    nb = _lib.query("step_tcn_conv_stats_ws_bytes", N, Co, To)
    res = []
    for _ in range(2):
        u = torch.full((N, Co, To, 16), float("nan"), device=d)
        sums = torch.zeros(2 * Co + 1, dtype=torch.float64, device=d)
        ws = torch.empty(nb, dtype=torch.uint8, device=d)
        _lib.call("step_tcn_conv_stats", y, sc, sh, 1, wp, b, u, N, Ci, Co, T, To, 75, stride, mode, sums, ws, nb)
        res.append((u, sums))
    assert torch.equal(res[0][0], ref) and torch.equal(res[0][1], res[1][1])
    ud = ref.double()
    got = res[0][1][:2 * Co].view(Co, 2)
    assert float((got[:, 0] - ud.sum(dim=(0, 2, 3))).abs().max()) <= 2e-6 * float(ud.abs().sum(dim=(0, 2, 3)).max())
    assert float(((got[:, 1] - (ud * ud).sum(dim=(0, 2, 3))) / (ud * ud).sum(dim=(0, 2, 3))).abs().max()) <= 2e-6
