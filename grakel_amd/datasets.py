"""TU-format graph collections straight to a packed ``GraphBatch``.

The reference's ``read_data`` (``grakel/datasets/base.py:135-290``) builds one Python set of
edge tuples and one label dict per graph; for the accelerated kernels that detour is the
end-to-end bottleneck.  This loader reads the same files

    <dir>/<name>/<name>_A.txt                 "u, v" per line, 1-based global node ids
    <dir>/<name>/<name>_graph_indicator.txt   graph id of node i (1-based, non-decreasing)
    <dir>/<name>/<name>_node_labels.txt       optional discrete node labels
    <dir>/<name>/<name>_graph_labels.txt      optional class per graph

with numpy and emits the CSR batch directly.  Edges are taken as listed (TU files list both
directions of an undirected edge), exactly like ``read_data`` with ``is_symmetric=False``.
Without a node-label file every node gets its out-degree as label when
``produce_labels_nodes`` is set (base.py:222-224), otherwise label 0.
"""
import os

import numpy as np

from .batch import GraphBatch, compress_labels


def read_tu(directory, name, produce_labels_nodes=False, with_classes=True):
    base = os.path.join(directory, name, name)
    gi = np.loadtxt(base + "_graph_indicator.txt", dtype=np.int64, ndmin=1)
    if np.any(np.diff(gi) < 0):
        raise ValueError("graph_indicator must be non-decreasing (nodes of a graph contiguous)")
    V = gi.shape[0]
    with open(base + "_A.txt") as f:
        A = np.loadtxt((line.replace(",", " ") for line in f), dtype=np.int64, ndmin=2)
    src, dst = A[:, 0] - 1, A[:, 1] - 1
    if src.size and (src.min() < 0 or max(src.max(), dst.max()) >= V):
        raise ValueError("edge endpoint outside the node range")
    if np.any(gi[src] != gi[dst]):
        raise ValueError("an edge connects two different graphs")
    key = np.unique(src * V + dst)                       # set semantics: duplicates collapse
    src, dst = key // V, key % V
    row_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=V), out=row_ptr[1:])
    _, first = np.unique(gi, return_index=True)
    graph_ptr = np.concatenate([first, [V]])
    lab_path = base + "_node_labels.txt"
    if os.path.exists(lab_path):
        labels = np.loadtxt(lab_path, dtype=np.int64, ndmin=1)
        if labels.shape[0] != V:
            raise ValueError("node_labels and graph_indicator disagree on the number of nodes")
    elif produce_labels_nodes:
        labels = np.bincount(src[src != dst], minlength=V)
    else:
        labels = np.zeros(V, dtype=np.int64)
    ids, mapping = compress_labels(labels.tolist())
    batch = GraphBatch(graph_ptr, row_ptr, dst, ids, max(len(mapping), 1))
    batch.label_map = mapping                              # original label value -> level-0 id
    classes = None
    cls_path = base + "_graph_labels.txt"
    if with_classes and os.path.exists(cls_path):
        classes = np.loadtxt(cls_path, dtype=np.int64, ndmin=1)
    return batch, classes
