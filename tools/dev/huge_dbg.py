import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import grakel_amd as gk
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset
from oracle import grakel_oracle as O
which = sys.argv[1]
rs = np.random.RandomState(5)
def chain(n, chords):
    ed = {i: [] for i in range(n)}
    for i in range(n - 1):
        ed[i].append(i + 1), ed[i + 1].append(i)
    for a, b in zip(rs.randint(0, n, chords).tolist(), rs.randint(0, n, chords).tolist()):
        if a != b and b not in ed[a]:
            ed[a].append(b), ed[b].append(a)
    return [ed, {i: int(rs.randint(0, 3)) for i in range(n)}]
eng = get_engine()
if which == "mid":      # graphs of 129..1000 vertices only: the wave path's table
    Y = [chain(n, n // 6) for n in (130, 500, 1000, 300, 129)] + er_dataset(40, 25, 0.1, 3, 9)
elif which == "huge":
    Y = [chain(1300, 200)] + er_dataset(80, 25, 0.1, 3, 9)
else:
    Y = [chain(1300, 200), chain(700, 100)] + er_dataset(80, 25, 0.1, 3, 9)
for o in sys.argv[2:]:
    eng.set_option(o.split("=")[0], int(o.split("=")[1]))
K = O.WLOracle(n_iter=4).fit_transform(Y)
Kd = gk.WeisfeilerLehman(n_iter=4).fit_transform(Y)
print(which, sys.argv[2:], "equal", np.array_equal(K, Kd), "max diff", np.abs(K - Kd).max())
