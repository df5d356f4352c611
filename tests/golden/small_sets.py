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


def sp_float_graphs(n_graphs=40, seed=23):
    """Weighted graphs with GENERAL float edge weights (0.1-multiples, two-decimal and random floats): the reference's
    feature keys are then its rounded float path sums, which differ between its floyd_warshall (adjacency input) and its
    dijkstra (dictionary input) and, for dijkstra, between the two directions of a pair.  Symmetric adjacency matrices,
    every third graph a weighted edge dictionary, every seventh a DIRECTED adjacency matrix; graphs 0..3 are hand-made
    cases (0.1 + 0.2 against 0.3; a path whose sums differ by direction)."""
    rs = np.random.RandomState(seed)
    lab3 = {0: 'x', 1: 'y', 2: 'x'}
    out = [[np.array([[0, 0.1, 0], [0.1, 0, 0.2], [0, 0.2, 0]]), dict(lab3)],
           [np.array([[0, 0, 0.3], [0, 0, 0], [0.3, 0, 0]]), dict(lab3)],
           [{0: {1: 0.1}, 1: {0: 0.1, 2: 0.2}, 2: {1: 0.2, 3: 0.3}, 3: {2: 0.3}}, {0: 'x', 1: 'y', 2: 'x', 3: 'y'}],
           [np.array([[0, 0.1, 0, 0], [0.1, 0, 0.2, 0], [0, 0.2, 0, 0.3], [0, 0, 0.3, 0]]), {0: 'x', 1: 'y', 2: 'x', 3: 'y'}]]
    for g in range(4, n_graphs):
        n = int(rs.randint(4, 13))
        A = np.zeros((n, n))
        kind = g % 3
        for i in range(n):
            for j in range(n):
                if i == j or (g % 7 != 0 and j < i):
                    continue
                if rs.rand() < 0.3:
                    w = [rs.randint(1, 30) / 10.0, round(rs.rand() * 3 + 0.01, 2), rs.rand() + 0.05][kind]
                    A[i, j] = w
                    if g % 7 != 0:
                        A[j, i] = w
        lab = {i: "xyz"[int(rs.randint(0, 3))] for i in range(n)}
        if g % 3 == 2:
            d = {i: {j: float(A[i, j]) for j in range(n) if A[i, j] > 0} for i in range(n)}
            out.append([d, lab])
        else:
            out.append([A, lab])
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


def sp_float_big_graphs(seed=5):
    """General float edge weights on graphs ABOVE 143 vertices (round 4: the float64 distance matrix no longer has to fit
    LDS): three sparse graphs of 150-200 vertices -- a symmetric adjacency matrix (the reference's floyd_warshall under
    "auto"), an edge dictionary (its dijkstra), a directed adjacency matrix -- and five small ones so that features are
    shared.  Sparse on purpose: the reference's floyd_warshall is a Python triple loop."""
    rs = np.random.RandomState(seed)
    out = []
    for g, n in enumerate((150, 180, 200, 9, 10, 11, 12, 8)):
        A = np.zeros((n, n))
        p = 2.2 / n if n > 100 else 0.3
        for i in range(n):
            for j in range(n):
                if i == j or (g != 2 and j < i):
                    continue
                if rs.rand() < p:
                    w = [rs.randint(1, 20) / 10.0, round(rs.rand() * 2 + 0.01, 2)][g % 2]
                    A[i, j] = w
                    if g != 2:
                        A[j, i] = w
        lab = {i: "xy"[int(rs.randint(0, 2))] for i in range(n)}
        if g == 1 or g == 4:
            out.append([{i: {j: float(A[i, j]) for j in range(n) if A[i, j] > 0} for i in range(n)}, lab])
        else:
            out.append([A, lab])
    return out


def sp_large_unit_graphs():
    """Unit-weight graphs ABOVE 128 vertices for the bit-parallel breadth-first search (sp.hip: sp_msbfs_kernel, round 5):
    three DIRECTED sparse adjacency matrices of 230-300 vertices (d[u][v] follows the out-edges of u; many unreachable
    pairs), and a 1 300-vertex graph with a vertex of 1 100 out-neighbours, one of 40, a directed chain of 50 and isolated
    vertices (more than 1 024 vertices: a thread owns several; hubs above 32 and above 1 024 neighbours).  `paths()`: two of
    the first with an undirected PATH of 300 vertices (distances up to 299: the search keeps its levels in bytes and has to
    hand the job to the row relaxation)."""
    rs = np.random.RandomState(5)
    G = []
    for n, p in ((230, 0.012), (300, 0.006), (260, 0.02)):
        A = (rs.rand(n, n) < p).astype(np.int64)
        np.fill_diagonal(A, 0)
        G.append([A, dict(enumerate(rs.randint(0, 3, n).tolist()))])
    n = 1300
    A = np.zeros((n, n), np.int64)
    A[0, 100:1200] = 1
    A[1, 0] = A[1, 60:99] = 1
    A[np.arange(1200, 1250), np.arange(1201, 1251)] = 1
    A[150, 1] = A[1250, 1200] = 1
    G.append([A, dict(enumerate((np.arange(n) % 4).tolist()))])
    return G


def sp_large_unit_paths():
    G = sp_large_unit_graphs()
    n = 300
    A = np.zeros((n, n), np.int64)
    A[np.arange(n - 1), np.arange(1, n)] = A[np.arange(1, n), np.arange(n - 1)] = 1
    return G[:2] + [[A, dict(enumerate((np.arange(n) % 2).tolist()))]]
