"""``GraphKernel``: the reference's name -> kernel-class dispatcher, for the three kernels
on the MI355X hot path (mirrors ``grakel/graph_kernels.py:452-554``).

    GraphKernel(kernel=[{"name": "weisfeiler_lehman", "n_iter": 5}, "vertex_histogram"], normalize=True)
    GraphKernel(kernel="WL") / "VH" / "ST-WL" / "SP" / {"name": "shortest_path", "with_labels": False}

Kernels outside the hot path and the Nystroem approximation raise ``NotImplementedError``.
"""
import copy
import warnings

from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils.validation import check_is_fitted

from .shortest_path import ShortestPath
from .core_framework import CoreFramework
from .vertex_histogram import VertexHistogram, EdgeHistogram
from .weisfeiler_lehman import WeisfeilerLehman
from .weisfeiler_lehman_optimal_assignment import WeisfeilerLehmanOptimalAssignment

_VH = ("vertex_histogram", "subtree_wl", "VH", "ST-WL")       # graph_kernels.py:38-40
_SP = ("shortest_path", "SP")
_WL = ("weisfeiler_lehman", "WL")
_EH = ("edge_histogram", "EH")
_OTHER_BASE = ("random_walk", "RW", "graphlet_sampling", "GR",
               "subgraph_matching", "SM", "multiscale_laplacian", "ML", "lovasz_theta", "LOVT",
               "svm_theta", "SVMT", "neighborhood_hash", "NH",
               "neighborhood_subgraph_pairwise_distance", "NSPD", "odd_sth", "ODD", "propagation",
               "PR", "pyramid_match", "PM", "graph_hopper", "GH")
_WLOA = ("weisfeiler_lehman_optimal_assignment", "WL-OA")
_OTHER_FRAMEWORKS = ("hadamard_code", "HC")
_CORE = ("core_framework", "CORE")


class GraphKernel(BaseEstimator, TransformerMixin):
    def __init__(self, kernel="shortest_path", normalize=False, verbose=False, n_jobs=None,
                 random_state=None, Nystroem=False):
        self.kernel = kernel
        self.normalize = normalize
        self.verbose = verbose
        self.n_jobs = n_jobs
        self.random_state = random_state
        self.Nystroem = Nystroem

    def make_kernel_(self, kernel_list, hidden_args):
        """graph_kernels.py:452-554 restricted to the accelerated kernels."""
        kernel = kernel_list.pop(0)
        if type(kernel) is str:
            name, kernel = str(kernel), dict()
        elif type(kernel) is not dict:
            raise ValueError('each element of the list of kernels must be a dictionary or a string')
        else:
            if "name" not in kernel:
                raise ValueError('each dictionary concerning a kernel must have a "name" parameter '
                                 'designating the kernel')
            name = kernel.pop("name")
        for key, val in hidden_args.items():
            if key in kernel:
                warnings.warn('Overriding global kernel attribute ' + str(key) + ' with ' + str(val) +
                              '. Please set this attribute as an argument of GraphKernel.')
            kernel[key] = val
        if name in _VH or name in _SP or name in _EH or name in _WLOA:
            if len(kernel_list) != 0:
                warnings.warn('Kernel List not empty while reaching a base-kernel - the rest kernel '
                              'names will be ignored')
            if name in _VH:
                return VertexHistogram, kernel
            if name in _EH:
                return EdgeHistogram, kernel
            if name in _WLOA:
                return WeisfeilerLehmanOptimalAssignment, kernel
            if kernel.pop("as_attributes", False):
                raise NotImplementedError('ShortestPathAttr is outside the MI355X hot path')
            return ShortestPath, kernel
        if name in _WL:
            if len(kernel_list):
                kernel["base_graph_kernel"] = self.make_kernel_(kernel_list, {})
            return WeisfeilerLehman, kernel
        if name in _CORE:                       # graph_kernels.py:543-551: default base is SP
            if len(kernel_list):
                kernel["base_graph_kernel"] = self.make_kernel_(kernel_list, {})
            return CoreFramework, kernel
        if name in _OTHER_BASE or name in _OTHER_FRAMEWORKS:
            raise NotImplementedError('kernel "%s" is outside the MI355X hot path (WL / VH / SP)' % name)
        raise ValueError("Unsupported kernel: " + str(name))

    def initialize(self):
        if self.Nystroem is not False:
            raise NotImplementedError('the Nystroem approximation is outside the MI355X hot path')
        k = self.kernel
        if type(k) is dict or type(k) is str:
            k = [k]
        elif type(k) is not list:
            raise ValueError('A "kernel" must be defined at the __init__ function of the graph kernel '
                             'generic wrapper. Valid kernel types are dict, str, and list of dict or str.')
        hidden = {"verbose": self.verbose, "normalize": self.normalize, "n_jobs": self.n_jobs}
        cls, params = self.make_kernel_(copy.deepcopy(k), hidden)
        self.kernel_ = cls(**params)

    def fit(self, X, y=None):
        self.initialize()
        self.kernel_.fit(X)
        return self

    def fit_transform(self, X, y=None):
        self.initialize()
        return self.kernel_.fit_transform(X)

    def transform(self, X):
        check_is_fitted(self, ["kernel_"])
        return self.kernel_.transform(X)
