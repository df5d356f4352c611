"""grakel_amd: MI355X-native WL-subtree / VertexHistogram / ShortestPath graph kernels
behind GraKeL's ``Kernel`` API (fit / transform / fit_transform / diagonal)."""
from .batch import GraphBatch
from .kernel import Kernel
from .vertex_histogram import VertexHistogram, EdgeHistogram
from .weisfeiler_lehman import WeisfeilerLehman
from .weisfeiler_lehman_optimal_assignment import WeisfeilerLehmanOptimalAssignment
from .shortest_path import ShortestPath
from .core_framework import CoreFramework
from .graph_kernels import GraphKernel

__all__ = ["GraphBatch", "Kernel", "VertexHistogram", "EdgeHistogram", "WeisfeilerLehman", "WeisfeilerLehmanOptimalAssignment", "ShortestPath", "CoreFramework", "GraphKernel"]
__version__ = "0.1.0"
