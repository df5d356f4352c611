import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def mutag_graphs():
    """MUTAG in grakel input form [set_of_(u,v), {node: label}] rebuilt from the packed arrays."""
    z = load_golden("mutag.npz")
    n_graphs = int(z["node_graph"].max()) + 1
    labels = [dict() for _ in range(n_graphs)]
    edges = [set() for _ in range(n_graphs)]
    g_of = dict()
    for v, g, l in zip(z["node_id"].tolist(), z["node_graph"].tolist(), z["node_label"].tolist()):
        labels[g][v] = l
        g_of[v] = g
    elabels = [dict() for _ in range(n_graphs)]
    el = z["edge_label"].tolist() if "edge_label" in z.files else [0] * len(z["edge_src"])
    for a, b, l in zip(z["edge_src"].tolist(), z["edge_dst"].tolist(), el):
        edges[g_of[a]].add((a, b))
        elabels[g_of[a]][(a, b)] = l
    return [[edges[g], labels[g], elabels[g]] for g in range(n_graphs)], z


def reference_available():
    """True only in the build container where the real grakel was built (oracle/build_ref.sh)."""
    ref = os.environ.get("GK_REF_BUILD", "/tmp/grakel_oracle")
    return os.path.isdir(os.path.join(ref, "grakel"))


@pytest.fixture(autouse=True, scope="session")
def _poisoned_allocator():
    """GK_TEST_POISON=<byte> (tests/tools/poison_suite.sh): every device block the library hands out is filled with
    that byte first, in every engine of this process -- uninitialised reads become deterministic."""
    if os.environ.get("GK_TEST_GUARD"):              # tests/tools/guard_suite.sh: red zones around every device block,
        from grakel_amd import engine as E            # checked when a block is released and at every gk_synchronize
        orig_g = E.Engine.__init__

        def init_g(self, device=0):
            orig_g(self, device)
            self.set_option("debug.guard", 1)

        E.Engine.__init__ = init_g
    pat = os.environ.get("GK_TEST_POISON")
    if pat:
        from grakel_amd import engine as E
        orig = E.Engine.__init__

        def init(self, device=0):
            orig(self, device)
            self.set_option("debug.poison", 0x100 | int(pat))

        E.Engine.__init__ = init
    yield


@pytest.fixture(autouse=True)
def _guard_verdict_after_every_test():
    """GK_TEST_GUARD=1: a write into a red zone during the test fails THAT test (gk_synchronize -> GK_ERR_STATE)."""
    yield
    if os.environ.get("GK_TEST_GUARD"):
        from grakel_amd import engine as E
        for eng in list(E._engines.values()):
            if eng.handle is not None:
                eng.synchronize()
