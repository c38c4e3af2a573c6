"""Skeleton graph of the 16-joint gait data (host side, runs once per model).

Same public surface as the reference's ``net/utils/graph.py`` (``Graph(strategy, max_hop,
dilation)`` with attributes ``A``, ``edge``, ``hop_dis``, ``num_node``, ``center``;
graph.py:4-100), computed with vectorised numpy.
"""
import numpy as np


class Graph:
    def __init__(self, strategy="uniform", max_hop=1, dilation=1):
        self.max_hop = max_hop
        self.dilation = dilation
        self.get_edge()
        self.hop_dis = get_hop_distance(self.num_node, self.edge, max_hop=max_hop)
        self.get_adjacency(strategy)

    def __str__(self):
        return str(self.A)

    def get_edge(self):
        # 16 joints: root 0, spine 1-2, head 3, two arms (4-6, 7-9), two legs (10-12, 13-15)
        self.num_node = 16
        chains = ((0, 1, 2, 3), (2, 4, 5, 6), (2, 7, 8, 9), (0, 10, 11, 12), (0, 13, 14, 15))
        bones = []
        for ch in chains:
            bones += list(zip(ch[:-1], ch[1:]))
        self.edge = [(i, i) for i in range(self.num_node)] + sorted(set(bones))
        self.center = 0

    def get_adjacency(self, strategy):
        hops = list(range(0, self.max_hop + 1, self.dilation))
        reach = np.isin(self.hop_dis, hops).astype(np.float64)
        norm = normalize_digraph(reach)
        if strategy == "uniform":
            self.A = norm[None].copy()
        elif strategy == "distance":
            self.A = np.stack([np.where(self.hop_dis == h, norm, 0.0) for h in hops])
        elif strategy == "spatial":
            to_center = self.hop_dis[:, self.center]
            dj, di = to_center[:, None], to_center[None, :]
            parts = []
            for h in hops:
                on = self.hop_dis == h
                same = np.where(on & (dj == di), norm, 0.0)
                inward = np.where(on & (dj > di), norm, 0.0)
                outward = np.where(on & ~(dj == di) & ~(dj > di), norm, 0.0)
                if h == 0:
                    parts.append(same)
                else:
                    parts.append(same + inward)
                    parts.append(outward)
            self.A = np.stack(parts)
        else:
            raise ValueError("Do Not Exist This Strategy")


def get_hop_distance(num_node, edge, max_hop=1):
    adj = np.zeros((num_node, num_node))
    for i, j in edge:
        adj[i, j] = adj[j, i] = 1
    hop = np.full((num_node, num_node), np.inf)
    powers = [np.linalg.matrix_power(adj, d) > 0 for d in range(max_hop + 1)]
    for d in range(max_hop, -1, -1):
        hop[powers[d]] = d
    return hop


def normalize_digraph(A):
    col = A.sum(axis=0)
    inv = np.divide(1.0, col, out=np.zeros_like(col), where=col > 0)
    return A @ np.diag(inv)


def normalize_undigraph(A):
    col = A.sum(axis=0)
    inv = np.power(col, -0.5, out=np.zeros_like(col), where=col > 0)
    return np.diag(inv) @ A @ np.diag(inv)
