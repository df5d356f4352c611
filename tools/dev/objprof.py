import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import grakel_amd
from grakel_amd.synthetic import er_dataset
X = er_dataset(10000, 100, 0.05, 5, 0)
est = grakel_amd.WeisfeilerLehman(n_iter=5)
est.fit_transform(X[:50])
K = est.fit_transform(X); del K
K = est.fit_transform(X); del K
for _ in range(3):
    t0 = time.perf_counter(); K = est.fit_transform(X); print("fit_transform s", time.perf_counter() - t0, flush=True); del K
pr = cProfile.Profile(); pr.enable(); K = est.fit_transform(X); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
