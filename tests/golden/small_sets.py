"""Definitions shared by make_golden.py (which runs the real reference) and the tests."""
import numpy as np

SMALL_SETS = [
    # name, kwargs for grakel_amd.synthetic.random_labelled_graphs
    ("dict_u", dict(N=40, nmin=3, nmax=14, p=0.3, L=3, seed=11, directed=False, fmt="dict")),
    ("adj_u", dict(N=40, nmin=2, nmax=14, p=0.25, L=4, seed=12, directed=False, fmt="adj")),
    ("adj_d", dict(N=30, nmin=3, nmax=12, p=0.3, L=3, seed=13, directed=True, fmt="adj")),
    ("tuples_d", dict(N=30, nmin=3, nmax=12, p=0.35, L=3, seed=14, directed=True, fmt="tuples")),
    ("dense_big", dict(N=12, nmin=40, nmax=70, p=0.5, L=2, seed=15, directed=False, fmt="adj")),
]


def split(G):
    ntr = (2 * len(G)) // 3
    return G[:ntr], G[ntr:]


def sp_inputs(kw, graphs):
    """The reference's Dijkstra route raises KeyError on isolated ``{v: []}`` vertices
    (SURVEY.md 7.5a), so ShortestPath goldens for dict sets use the adjacency route."""
    if kw["fmt"] != "dict":
        return graphs
    out = []
    for g in graphs:
        n = len(g[1])
        A = np.zeros((n, n), dtype=int)
        for a, lst in g[0].items():
            for b in lst:
                A[a, b] = 1
        out.append([A, g[1]])
    return out


def sp_dyadic_graphs(n_graphs=24, seed=11):
    """Weighted graphs whose float edge weights are multiples of 1/8 (0.125 .. 4.0): every path sum is exact
    in float64, so the reference's float distance keys are well defined (graph.py:1767-1794).  Adjacency
    matrices (symmetric) and, for every third graph, a weighted edge dictionary."""
    rs = np.random.RandomState(seed)
    out = []
    for g in range(n_graphs):
        n = int(rs.randint(4, 12))
        A = np.zeros((n, n))
        for i in range(n):
            for j in range(i + 1, n):
                if rs.rand() < 0.35:
                    A[i, j] = A[j, i] = rs.randint(1, 33) / 8.0
        lab = {i: "xyz"[int(rs.randint(0, 3))] for i in range(n)}
        if g % 3 == 2:
            d = {i: {j: float(A[i, j]) for j in range(n) if A[i, j] > 0} for i in range(n)}
            out.append([d, lab])
        else:
            out.append([A, lab])
    return out
