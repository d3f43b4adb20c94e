/*
 * oracle/ref_bulk.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Bulk C entry points over the *unmodified* reference Boykov-Kolmogorov solver
 * (reference lib/maxflow/src/graph.h, graph.cpp, maxflow.cpp), compiled from the
 * sources where they lie under /root/reference by oracle/Makefile into
 * oracle/_ref/libbkref.so.  No reference source is copied into this repository;
 * this file only *calls* the reference class Graph<double,double,double>
 * (the instantiation MedPy exposes as GraphDouble, reference
 * lib/maxflow/src/wrapper.cpp:59-89, instances.inc:15).
 *
 * Why bulk: the reference drives sum_edge()/add_tweights()/what_segment() once
 * per edge / node from Python (energy_voxel.py:660-664, graph.py:551-552,
 * bin/medpy_graphcut_voxel.py:177-181).  At 256^3 that is 5e7 Python calls, so the
 * oracle takes whole arrays per call instead; the per-call semantics are unchanged.
 */
#include <stdint.h>
#include <stddef.h>
#include "graph.h"

typedef Graph<double, double, double> GraphD;

extern "C" {

void* bkref_create(int64_t nodes, int64_t edges)
{
    if (nodes >= 2147483647LL || 2 * edges >= 2147483647LL) return NULL;  /* graph.h:62,82: 32-bit ids */
    GraphD* g = new GraphD((int)nodes, (int)edges, NULL);  /* graph.py:305 */
    if (nodes > 0) g->add_node((int)nodes);               /* graph.py:306 */
    return g;
}

void bkref_destroy(void* h) { delete (GraphD*)h; }

/* GCGraph.set_nweight -> sum_edge (graph.py:438-440, graph.h:457-480) */
void bkref_sum_edges(void* h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev)
{
    GraphD* g = (GraphD*)h;
    for (int64_t k = 0; k < n; ++k) g->sum_edge((int)i[k], (int)j[k], cap[k], rev[k]);
}

/* plain add_edge (graph.h:428-454); parallel arcs allowed */
void bkref_add_edges(void* h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev)
{
    GraphD* g = (GraphD*)h;
    for (int64_t k = 0; k < n; ++k) g->add_edge((int)i[k], (int)j[k], cap[k], rev[k]);
}

/* GCGraph.set_tweight -> add_tweights (graph.py:496-498, graph.h:416-425); idx==NULL means nodes 0..n-1 */
void bkref_add_tweights(void* h, int64_t n, const int64_t* idx, const double* src, const double* snk)
{
    GraphD* g = (GraphD*)h;
    for (int64_t k = 0; k < n; ++k) g->add_tweights((int)(idx ? idx[k] : k), src[k], snk[k]);
}

/*
 * The per-edge insertion loop of __skeleton_base (energy_voxel.py:637-664) for a
 * C-ordered lattice of `ndim` axes: axis 0 first, each axis in C order of the sliced
 * array, symmetric capacities.  w[d] holds the (shape[d]-1) * prod(other) weights of
 * axis d in exactly the order `neighbourhood_intensity_term.ravel()` yields them.
 */
void bkref_sum_lattice(void* h, int ndim, const int64_t* shape, const double* const* w)
{
    GraphD* g = (GraphD*)h;
    for (int d = 0; d < ndim; ++d) {
        int64_t offset = 1;
        for (int k = d + 1; k < ndim; ++k) offset *= shape[k];
        int64_t outer = 1;
        for (int k = 0; k < d; ++k) outer *= shape[k];
        const int64_t inner = (shape[d] - 1) * offset; /* idx_offset_divider, energy_voxel.py:653 */
        const double* wd = w[d];
        int64_t key = 0;
        for (int64_t o = 0; o < outer; ++o) {
            const int64_t base = o * shape[d] * offset;
            for (int64_t r = 0; r < inner; ++r, ++key) {
                const int64_t p = base + r;
                g->sum_edge((int)p, (int)(p + offset), wd[key], wd[key]);
            }
        }
    }
}

double bkref_maxflow(void* h) { return ((GraphD*)h)->maxflow(); }

/* CLI label rule: 0 if SINK == what_segment(idx) else 1 (bin/medpy_graphcut_voxel.py:177-181) */
void bkref_labels(void* h, int64_t n, uint8_t* out)
{
    GraphD* g = (GraphD*)h;
    for (int64_t k = 0; k < n; ++k) out[k] = (g->what_segment((int)k) == GraphD::SINK) ? 0 : 1;
}

int    bkref_what_segment(void* h, int64_t i) { return (int)((GraphD*)h)->what_segment((int)i); }
double bkref_get_trcap(void* h, int64_t i) { return ((GraphD*)h)->get_trcap((int)i); }
double bkref_get_edge(void* h, int64_t i, int64_t j) { return ((GraphD*)h)->get_edge((int)i, (int)j); }
int64_t bkref_get_node_num(void* h) { return ((GraphD*)h)->get_node_num(); }
int64_t bkref_get_arc_num(void* h) { return ((GraphD*)h)->get_arc_num(); }

/* residual graph read-out for the ambiguity check (oracle/cutcheck.py): all arcs in allocation order (sister arcs are
 * adjacent: 2k, 2k + 1, graph.h:428-454) with their residual capacities (graph.h:539-543), and the residual t-links */
void bkref_export(void* h, int32_t* tail, int32_t* head, double* rcap, double* trcap)
{
    GraphD* g = (GraphD*)h;
    const int64_t na = g->get_arc_num();
    GraphD::arc_id a = na ? g->get_first_arc() : NULL;
    for (int64_t k = 0; k < na; ++k) {
        int i, j;
        g->get_arc_ends(a, i, j);
        tail[k] = i; head[k] = j; rcap[k] = g->get_rcap(a);
        a = g->get_next_arc(a);
    }
    const int64_t nn = g->get_node_num();
    for (int64_t k = 0; k < nn; ++k) trcap[k] = g->get_trcap((int)k);
}

} /* extern "C" */
