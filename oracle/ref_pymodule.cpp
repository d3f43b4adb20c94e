/*
 * oracle/ref_pymodule.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * pybind11 stand-in for the reference's Boost.Python binding
 * (reference lib/maxflow/src/wrapper.cpp:27-134, pythongraph.h:14-22): Boost.Python is
 * not installed in this image, so only the *binding* is replaced; the solver it exposes
 * is the unmodified reference Graph<> template compiled from /root/reference.
 * Built by oracle/Makefile into oracle/_ref/overlay/medpy/graphcut/maxflow*.so so that
 * `import medpy.graphcut` resolves to the reference's own .py files plus this module
 * (see oracle/overlay.py).  Used in this container to run the reference pipeline and to
 * generate tests/golden/ fixtures (oracle/gen_golden.py).
 *
 * Exposes the same class / method / enum names as wrapper.cpp:59-89.  Unlike
 * pythongraph.h:20-21 (missing `return`, correct only at -O0) the values are returned.
 */
#include <pybind11/pybind11.h>
#include "graph.h"

namespace py = pybind11;

template <typename T>
static void wrap(py::module_& m, const char* name)
{
    typedef Graph<T, T, T> G;
    py::class_<G> c(m, name);
    c.def(py::init([](int nodes, int edges) { return new G(nodes, edges, NULL); }))
        .def("add_node", [](G& g, int num) { return g.add_node(num); }, py::arg("num") = 1)
        .def("add_edge", &G::add_edge)
        .def("sum_edge", &G::sum_edge)
        .def("add_tweights", &G::add_tweights)
        .def("maxflow", [](G& g) { return g.maxflow(); })
        .def("what_segment", [](G& g, int i) { return g.what_segment(i); })
        .def("reset", &G::reset)
        .def("get_edge", &G::get_edge)
        .def("get_node_num", &G::get_node_num)
        .def("get_arc_num", &G::get_arc_num)
        .def("get_trcap", &G::get_trcap)
        .def("set_trcap", &G::set_trcap)
        .def("mark_node", &G::mark_node)
        .def("remove_from_changed_list", &G::remove_from_changed_list);
    py::enum_<typename G::termtype>(c, "termtype")
        .value("SOURCE", G::SOURCE)
        .value("SINK", G::SINK);
}

PYBIND11_MODULE(maxflow, m)
{
    m.doc() = "pybind11 stand-in for medpy.graphcut.maxflow (reference BK v3.01 core, unmodified)";
    wrap<float>(m, "GraphFloat");
    wrap<double>(m, "GraphDouble");
    wrap<int>(m, "GraphInt");
}
