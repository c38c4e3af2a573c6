"""CPU restatement of the CVAE training loss (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Reference: generator_cvae/utils/losses.py:7-42 ``between_frame_loss`` and
generator_cvae/utils/processor_stgcn.py:34-40 ``vae_loss``.  Parity PINNED by
tests/golden/vae_loss.npz and cvae_vae_loss_train.npz (values and autograd gradients produced
by the reference's own functions, tests/golden/make_golden.py gen_vae_loss).

The reference builds the loss from O(B*T*V*C) scalar ``mse_loss`` calls.  Every call is a mean
of squared differences of (g1 - g2), so with D = g1 - g2 viewed as (B, T, F), F = C*V:
    L = mean(D^2)
      + SUM_b SUM_t SUM_{a in {0, mid, T-1}} mean_f (D[b,t] - D[b,a])^2          (losses.py:15-20)
      + SUM_b SUM_{t>=1} SUM_f (D[b,t,f] - D[b,t-1,f])^2                          (:22-24)
      + SUM_b SUM_{t>=2} SUM_f (D[b,t,f] - 2 D[b,t-1,f] + D[b,t-2,f])^2           (:25-29)
with mid = int(T/2) - 1 (a negative index wraps, as Python indexing does in the reference).
The feature order (v, c) of the reference's permute is irrelevant: every term weighs all F alike.
"""
import torch


def _as_btf(g):
    """(N, C, T, V, M) -> (N, T, F) in the reference's own order (losses.py:8-9)."""
    return g.permute(0, 2, 3, 1, 4).contiguous().view(g.shape[0], g.shape[2], g.shape[1] * g.shape[3])


def between_frame_loss(gait1, gait2, thres=0.01):
    """closed form of losses.py:7-42 (``thres`` is unused there too)"""
    g1, g2 = _as_btf(gait1), _as_btf(gait2)
    D = g1 - g2
    T = D.shape[1]
    mid = int(T / 2) - 1
    loss = (D ** 2).mean()
    for a in (0, mid, -1):
        loss = loss + ((D - D[:, a, :].unsqueeze(1)) ** 2).mean(dim=2).sum()
    if T > 1:
        loss = loss + ((D[:, 1:] - D[:, :-1]) ** 2).sum()
    if T > 2:
        loss = loss + ((D[:, 2:] - 2 * D[:, 1:-1] + D[:, :-2]) ** 2).sum()
    return loss


def between_frame_loss_loops(gait1, gait2):
    """the reference's statement order, one scalar term at a time (tiny inputs only)"""
    g1, g2 = _as_btf(gait1), _as_btf(gait2)
    B, T, F = g1.shape
    mid = int(T / 2) - 1
    mse = torch.nn.functional.mse_loss
    loss = mse(g1, g2)
    for b in range(B):
        for t in range(T):
            for a in (0, mid, -1):
                loss = loss + mse(g1[b, t] - g1[b, a], g2[b, t] - g2[b, a])
            for f in range(F):
                if t > 0:
                    loss = loss + mse(g1[b, t, f] - g1[b, t - 1, f], g2[b, t, f] - g2[b, t - 1, f])
                if t > 1:
                    loss = loss + mse(g1[b, t, f] - 2 * g1[b, t - 1, f] + g1[b, t - 2, f],
                                      g2[b, t, f] - 2 * g2[b, t - 1, f] + g2[b, t - 2, f])
    return loss


def kld(mean, lsig):
    """processor_stgcn.py:39"""
    return -0.5 * torch.sum(1 + lsig - mean.pow(2) - lsig.exp())


def vae_loss(x_in, x_out, mean, lsig, beta=1.0):
    """processor_stgcn.py:34-40"""
    return between_frame_loss(x_in, x_out) + beta * kld(mean, lsig)
