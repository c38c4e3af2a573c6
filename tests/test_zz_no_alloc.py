"""The entry point that allocates nothing: step_bn_train_prepare_static (include/step_hip.h) against
step_bn_train_prepare, through the C ABI, bit for bit.  Two calls in a row share the library's static
workspace, the second with a smaller input; a shape beyond the static maximum is rejected.  Collected last
(pytest -x at round end must not hide other tests behind it)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(name, x, bn_mode, gamma, beta, with_ws):
    from step_amd import _lib
    N, C, T, V = x.shape
    G = C * V if bn_mode else C
    dev = x.device
    rm = torch.linspace(-1, 1, G, device=dev)
    rv = torch.linspace(0.5, 2, G, device=dev)
    outs = [torch.empty(G, dtype=torch.float32, device=dev) for _ in range(4)]
    args = [x, N, C, T, V, bn_mode, gamma, beta, rm, rv, 0.1, 1e-5] + outs
    if with_ws:
        nb = _lib.query("step_bn_ws_bytes", G)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        args += [ws, nb]
    _lib.call(name, *args)
    torch.cuda.synchronize()
    return outs + [rm, rv]


@pytest.mark.parametrize("bn_mode", [0, 1])
def test_static_workspace_matches_caller_workspace(bn_mode):
    torch.manual_seed(3)
    dev = "cuda:0"
    shapes = [(300, 16, 60, 16), (37, 16, 9, 16)] if bn_mode else [(300, 64, 60, 16), (37, 32, 9, 16)]
    for shape in shapes:                      # second call: smaller input, same static array
        C = shape[1]
        G = C * 16 if bn_mode else C
        x = (torch.randn(shape, device=dev) * 2 + 0.7).contiguous()
        gamma = torch.rand(G, device=dev) + 0.5
        beta = torch.randn(G, device=dev)
        ref = _run("step_bn_train_prepare", x, bn_mode, gamma, beta, True)
        got = _run("step_bn_train_prepare_static", x, bn_mode, gamma, beta, False)
        for a, b, what in zip(got, ref, ("mean", "invstd", "scale", "shift", "running_mean", "running_var")):
            assert torch.equal(a, b), (shape, what, float((a - b).abs().max()))
        # against torch's own batch statistics, so both are not wrong together
        xs = x.double()
        dims = (0, 2) if bn_mode else (0, 2, 3)
        m = xs.mean(dims)
        m = m.t().reshape(-1) if bn_mode else m
        assert float((got[0].double() - m).abs().max()) < 1e-5


def test_static_workspace_rejects_oversize():
    from step_amd import _lib
    dev = "cuda:0"
    x = torch.zeros(2, 32, 4, 16, device=dev)           # data_bn layout: G = 512 > 256
    outs = [torch.empty(512, device=dev) for _ in range(4)]
    with pytest.raises(_lib.StepHipError, match="exceed the static workspace"):
        _lib.call("step_bn_train_prepare_static", x, 2, 32, 4, 16, 1, None, None, None, None, 0.1, 1e-5, *outs)


def test_static_entry_replays_from_a_hip_graph():
    """The callers this entry point is for: captured after ONE warm-up call on the device (the first call looks the
    static array's address up with hipGetSymbolAddress, which is not a capturable call), replayed on new data."""
    from step_amd import _lib
    dev = "cuda:0"
    torch.manual_seed(5)
    N, C, T, V = 64, 32, 30, 16
    x = torch.randn(N, C, T, V, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    outs = [torch.empty(C, device=dev) for _ in range(4)]
    args = [x, N, C, T, V, 0, gamma, beta, rm, rv, 0.1, 1e-5] + outs
    _lib.call("step_bn_train_prepare_static", *args)          # warm-up: resolves the address outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _lib.call("step_bn_train_prepare_static", *args)
    for seed in (1, 2):                                       # replay twice on fresh inputs written in place
        x.copy_(torch.randn(N, C, T, V, device=dev, generator=torch.Generator(dev).manual_seed(seed)) * (1 + seed))
        rm.zero_()
        rv.fill_(1.0)
        graph.replay()
        torch.cuda.synchronize()
        ref = _run("step_bn_train_prepare", x, 0, gamma, beta, True)
        for a, b, what in zip(outs, ref[:4], ("mean", "invstd", "scale", "shift")):
            assert torch.equal(a, b), (seed, what)
