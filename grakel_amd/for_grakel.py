"""The accelerated kernels as classes the REFERENCE's own frameworks accept as ``base_graph_kernel``.

``grakel.HadamardCode`` / ``CoreFramework`` / ``WeisfeilerLehman`` / ``GraphKernel`` take a base kernel only if
``type(k) is type and issubclass(k, grakel.kernels.Kernel)`` (``hadamard_code.py:73,86``, ``core_framework.py:75,86``,
``weisfeiler_lehman.py:82,95``).  The classes of ``grakel_amd`` implement that protocol -- constructor parameters,
``fit`` / ``fit_transform`` / ``transform`` / ``diagonal``, elements that are ``grakel.Graph`` objects or
``(graph, labels) + extras`` tuples -- without importing the reference.  THIS module imports it (it is the
only one that does, and nothing else in the package imports this module): every class below is
``(accelerated class, grakel.kernels.Kernel)``, so each method resolves to the accelerated class first and the reference's
base is only there for the subclass check.  Plain module-level classes: a fitted framework, which holds the class itself,
pickles (``grakel/tests/test_common.py:53-58``).

    from grakel import HadamardCode, CoreFramework
    from grakel_amd.for_grakel import VertexHistogram, ShortestPath
    HadamardCode(base_graph_kernel=VertexHistogram).fit_transform(graphs)
    CoreFramework(base_graph_kernel=(ShortestPath, {"with_labels": True})).fit_transform(graphs)
"""
from grakel.kernels import Kernel as _ReferenceKernel

from . import vertex_histogram as _vh, shortest_path as _sp, weisfeiler_lehman as _wl
from . import weisfeiler_lehman_optimal_assignment as _oa


class VertexHistogram(_vh.VertexHistogram, _ReferenceKernel):
    __doc__ = _vh.VertexHistogram.__doc__


class EdgeHistogram(_vh.EdgeHistogram, _ReferenceKernel):
    __doc__ = _vh.EdgeHistogram.__doc__


class ShortestPath(_sp.ShortestPath, _ReferenceKernel):
    __doc__ = _sp.ShortestPath.__doc__


class WeisfeilerLehman(_wl.WeisfeilerLehman, _ReferenceKernel):
    __doc__ = _wl.WeisfeilerLehman.__doc__


class WeisfeilerLehmanOptimalAssignment(_oa.WeisfeilerLehmanOptimalAssignment, _ReferenceKernel):
    __doc__ = _oa.WeisfeilerLehmanOptimalAssignment.__doc__


__all__ = ["VertexHistogram", "EdgeHistogram", "ShortestPath", "WeisfeilerLehman", "WeisfeilerLehmanOptimalAssignment"]
