"""Oracle (test infrastructure): the 16-joint skeleton adjacency of STEP.

Restates ``net/utils/graph.py`` of the reference (identical in all four app
directories): ``Graph.get_edge`` graph.py:40-55, ``get_hop_distance``
graph.py:103-115, ``normalize_digraph`` graph.py:118-126 and the three
partition strategies of ``Graph.get_adjacency`` graph.py:57-100.
Written with vectorised numpy; float64 like the reference.
"""
import numpy as np

NUM_NODE = 16
CENTER = 0
# bones of the gait skeleton, graph.py:43-53
BONES = ((0, 1), (0, 10), (0, 13), (1, 2), (2, 3), (2, 4), (2, 7), (4, 5),
         (5, 6), (7, 8), (8, 9), (10, 11), (11, 12), (13, 14), (14, 15))


def hop_distance(num_node=NUM_NODE, bones=BONES, max_hop=1):
    """graph.py:103-115 - smallest d <= max_hop with (A^d)[i,j] > 0, else inf."""
    adj = np.zeros((num_node, num_node))
    for i, j in bones:
        adj[i, j] = adj[j, i] = 1.0
    for i in range(num_node):          # self links, graph.py:42
        adj[i, i] = 1.0
    hop = np.full((num_node, num_node), np.inf)
    power = np.eye(num_node)
    reach = []
    for d in range(max_hop + 1):
        reach.append(power > 0)
        power = power @ adj
    for d in range(max_hop, -1, -1):
        hop[reach[d]] = d
    return hop


def normalize_digraph(adj):
    """graph.py:118-126 - A @ D^-1 with D = column sums (columns sum to 1)."""
    col = adj.sum(axis=0)
    inv = np.where(col > 0, 1.0 / np.where(col > 0, col, 1.0), 0.0)
    return adj * inv[None, :]


def graph_adjacency(strategy="uniform", max_hop=1, dilation=1):
    """graph.py:57-100 - returns A of shape (K, 16, 16), float64."""
    hop = hop_distance(max_hop=max_hop)
    valid = list(range(0, max_hop + 1, dilation))
    adj = np.zeros((NUM_NODE, NUM_NODE))
    for h in valid:
        adj[hop == h] = 1.0
    norm = normalize_digraph(adj)
    if strategy == "uniform":
        return norm[None].copy()
    if strategy == "distance":
        return np.stack([np.where(hop == h, norm, 0.0) for h in valid])
    if strategy == "spatial":
        dist_c = hop[:, CENTER]                    # hop distance to the centre joint
        dj = dist_c[:, None]                       # row index j
        di = dist_c[None, :]                       # column index i
        parts = []
        for h in valid:
            on = hop == h                          # hop_dis[j, i] == hop
            root = np.where(on & (dj == di), norm, 0.0)
            close = np.where(on & (dj > di), norm, 0.0)
            further = np.where(on & (dj < di), norm, 0.0)
            if h == 0:
                parts.append(root)
            else:
                parts.append(root + close)
                parts.append(further)
        return np.stack(parts)
    raise ValueError("Do Not Exist This Strategy")
