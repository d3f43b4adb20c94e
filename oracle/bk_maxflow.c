/*
 * oracle/bk_maxflow.c -- TEST INFRASTRUCTURE ONLY: parity checker / CPU baseline ("port").
 *
 * A from-scratch C restatement of the Boykov-Kolmogorov augmenting-path max-flow that the
 * reference ships as lib/maxflow (BK v3.01 as modified by MedPy), for IEEE double
 * capacities (the reference's GraphDouble, lib/maxflow/src/instances.inc:15).  It is the
 * checker the HIP path is compared against when oracle/_ref (the unmodified reference
 * compiled in place) is not available, e.g. on the GPU box.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (medpy_amd/) never does.
 *
 * Pinned: tests/test_oracle_port_vs_ref.py requires bit-identical flow values, labels and
 * t-link residuals against oracle/_ref/libbkref.so on random graphs, the reference's
 * known-answer tests and the synthetic lattices; tests/golden/ holds reference outputs.
 *
 * Layout differs from the reference on purpose (index arrays instead of pointer-linked
 * structs, sister arc = a^1) but every *decision* follows the reference so that the
 * sequence of augmentations -- and therefore every floating point rounding -- is the same:
 *
 *   graph building   graph.h:388-509  (add_node / add_tweights / add_edge / sum_edge / get_edge)
 *   active queues    maxflow.cpp:33-75
 *   orphan lists     maxflow.cpp:79-101
 *   init             maxflow.cpp:119-156
 *   augment          maxflow.cpp:244-311
 *   adoption         maxflow.cpp:316-467
 *   main loop        maxflow.cpp:472-604
 *   what_segment     graph.h:561-571
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P_NONE     (-1) /* parent == NULL : free node              */
#define P_TERMINAL (-2) /* parent == TERMINAL (maxflow.cpp:11)     */
#define P_ORPHAN   (-3) /* parent == ORPHAN   (maxflow.cpp:12)     */
#define Q_NONE     (-1) /* next == NULL : not in an active queue   */
#define INFINITE_D 0x7fffffff /* maxflow.cpp:15 */

typedef struct bkport {
    int32_t n_nodes;
    int32_t n_arcs, arc_cap; /* arcs allocated in sister pairs: sister(a) = a ^ 1 */
    /* per node (graph.h:290-306) */
    int32_t* first;   /* first outgoing arc or -1                          */
    int32_t* parent;  /* arc towards the parent, or P_*                    */
    int32_t* qnext;   /* next active node; self = last in list; Q_NONE     */
    int32_t* ts;      /* timestamp of dist                                 */
    int32_t* dist;    /* distance to the terminal                          */
    uint8_t* is_sink; /* tree membership (valid iff parent != P_NONE)      */
    double*  trcap;   /* >0: residual source->node, <0: -residual node->sink */
    /* per arc (graph.h:308-315) */
    int32_t* head;
    int32_t* next;
    double*  rcap;
    /* solver state */
    double  flow;
    int32_t qfirst[2], qlast[2];
    int32_t time;
    /* orphan list cells (maxflow.cpp:79-101 uses a DBlock pool of nodeptr) */
    int32_t *cell_node, *cell_next;
    int32_t cell_cap, cell_used, cell_free;
    int32_t orphan_first, orphan_last;
    int oom;
} bkport;

/* ------------------------------------------------------------------ build */

void* bkport_create(int64_t nodes, int64_t edges)
{
    if (nodes >= 2147483647LL || 2 * edges >= 2147483647LL || nodes < 0 || edges < 0) return NULL;
    bkport* g = (bkport*)calloc(1, sizeof(bkport));
    if (!g) return NULL;
    int64_t nn = nodes < 16 ? 16 : nodes; /* graph.cpp:17-18 */
    int64_t ee = edges < 16 ? 16 : edges;
    g->n_nodes = (int32_t)nodes; /* GCGraph adds all nodes at once, graph.py:306 */
    g->arc_cap = (int32_t)(2 * ee);
    g->first = (int32_t*)malloc(nn * sizeof(int32_t));
    g->parent = (int32_t*)malloc(nn * sizeof(int32_t));
    g->qnext = (int32_t*)malloc(nn * sizeof(int32_t));
    g->ts = (int32_t*)calloc(nn, sizeof(int32_t));
    g->dist = (int32_t*)calloc(nn, sizeof(int32_t));
    g->is_sink = (uint8_t*)calloc(nn, 1);
    g->trcap = (double*)calloc(nn, sizeof(double)); /* add_node memset, graph.h:406 */
    g->head = (int32_t*)malloc((size_t)g->arc_cap * sizeof(int32_t));
    g->next = (int32_t*)malloc((size_t)g->arc_cap * sizeof(int32_t));
    g->rcap = (double*)malloc((size_t)g->arc_cap * sizeof(double));
    if (!g->first || !g->parent || !g->qnext || !g->ts || !g->dist || !g->is_sink || !g->trcap || !g->head ||
        !g->next || !g->rcap) {
        g->oom = 1;
        return g;
    }
    for (int64_t i = 0; i < nn; ++i) {
        g->first[i] = -1;
        g->parent[i] = P_NONE;
        g->qnext[i] = Q_NONE;
    }
    g->cell_free = -1;
    g->orphan_first = g->orphan_last = -1;
    return g;
}

void bkport_destroy(void* h)
{
    bkport* g = (bkport*)h;
    if (!g) return;
    free(g->first); free(g->parent); free(g->qnext); free(g->ts); free(g->dist); free(g->is_sink);
    free(g->trcap); free(g->head); free(g->next); free(g->rcap); free(g->cell_node); free(g->cell_next);
    free(g);
}

static int grow_arcs(bkport* g) /* graph.cpp:87-114: grow by half */
{
    int64_t cap = (int64_t)g->arc_cap + g->arc_cap / 2;
    if (cap & 1) cap++;
    if (cap >= 2147483647LL) return 0;
    int32_t* h2 = (int32_t*)realloc(g->head, (size_t)cap * sizeof(int32_t));
    if (h2) g->head = h2;
    int32_t* n2 = (int32_t*)realloc(g->next, (size_t)cap * sizeof(int32_t));
    if (n2) g->next = n2;
    double* r2 = (double*)realloc(g->rcap, (size_t)cap * sizeof(double));
    if (r2) g->rcap = r2;
    if (!h2 || !n2 || !r2) return 0;
    g->arc_cap = (int32_t)cap;
    return 1;
}

/* graph.h:428-454: two arcs, each pushed at the FRONT of its tail's adjacency list */
static void add_edge(bkport* g, int32_t i, int32_t j, double cap, double rev)
{
    if (g->n_arcs == g->arc_cap && !grow_arcs(g)) { g->oom = 1; return; }
    int32_t a = g->n_arcs, b = a + 1;
    g->n_arcs += 2;
    g->next[a] = g->first[i]; g->first[i] = a;
    g->next[b] = g->first[j]; g->first[j] = b;
    g->head[a] = j;
    g->head[b] = i;
    g->rcap[a] = cap;
    g->rcap[b] = rev;
}

static int32_t get_arc(const bkport* g, int32_t i, int32_t j) /* graph.h:500-509 */
{
    for (int32_t a = g->first[i]; a >= 0; a = g->next[a])
        if (g->head[a] == j) return a;
    return -1;
}

/* graph.h:457-480: accumulate onto the first existing arc i->j, else create */
static void sum_edge(bkport* g, int32_t i, int32_t j, double cap, double rev)
{
    int32_t a = get_arc(g, i, j);
    if (a >= 0) {
        g->rcap[a] += cap;
        g->rcap[a ^ 1] += rev;
    } else {
        add_edge(g, i, j, cap, rev);
    }
}

void bkport_sum_edges(void* h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev)
{
    bkport* g = (bkport*)h;
    for (int64_t k = 0; k < n; ++k) sum_edge(g, (int32_t)i[k], (int32_t)j[k], cap[k], rev[k]);
}

void bkport_add_edges(void* h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev)
{
    bkport* g = (bkport*)h;
    for (int64_t k = 0; k < n; ++k) add_edge(g, (int32_t)i[k], (int32_t)j[k], cap[k], rev[k]);
}

/* graph.h:416-425: t-links accumulate; the common part goes straight into the flow value */
static void add_tweights(bkport* g, int32_t i, double cap_source, double cap_sink)
{
    double delta = g->trcap[i];
    if (delta > 0) cap_source += delta;
    else           cap_sink -= delta;
    g->flow += (cap_source < cap_sink) ? cap_source : cap_sink;
    g->trcap[i] = cap_source - cap_sink;
}

void bkport_add_tweights(void* h, int64_t n, const int64_t* idx, const double* src, const double* snk)
{
    bkport* g = (bkport*)h;
    for (int64_t k = 0; k < n; ++k) add_tweights(g, (int32_t)(idx ? idx[k] : k), src[k], snk[k]);
}

/* the per-edge loop of __skeleton_base, energy_voxel.py:637-664 (see oracle/ref_bulk.cpp) */
void bkport_sum_lattice(void* h, int ndim, const int64_t* shape, const double* const* w)
{
    bkport* g = (bkport*)h;
    for (int d = 0; d < ndim; ++d) {
        int64_t offset = 1, outer = 1;
        for (int k = d + 1; k < ndim; ++k) offset *= shape[k];
        for (int k = 0; k < d; ++k) outer *= shape[k];
        const int64_t inner = (shape[d] - 1) * offset;
        const double* wd = w[d];
        int64_t key = 0;
        for (int64_t o = 0; o < outer; ++o) {
            const int64_t base = o * shape[d] * offset;
            for (int64_t r = 0; r < inner; ++r, ++key)
                sum_edge(g, (int32_t)(base + r), (int32_t)(base + r + offset), wd[key], wd[key]);
        }
    }
}

/* ------------------------------------------------------------------ active queues (maxflow.cpp:33-75) */

static void set_active(bkport* g, int32_t i)
{
    if (g->qnext[i] == Q_NONE) {
        if (g->qlast[1] >= 0) g->qnext[g->qlast[1]] = i;
        else                  g->qfirst[1] = i;
        g->qlast[1] = i;
        g->qnext[i] = i;
    }
}

static int32_t next_active(bkport* g)
{
    for (;;) {
        int32_t i = g->qfirst[0];
        if (i < 0) {
            g->qfirst[0] = i = g->qfirst[1];
            g->qlast[0] = g->qlast[1];
            g->qfirst[1] = g->qlast[1] = -1;
            if (i < 0) return -1;
        }
        if (g->qnext[i] == i) g->qfirst[0] = g->qlast[0] = -1;
        else                  g->qfirst[0] = g->qnext[i];
        g->qnext[i] = Q_NONE;
        if (g->parent[i] != P_NONE) return i; /* active iff it has a parent */
    }
}

/* ------------------------------------------------------------------ orphan lists (maxflow.cpp:79-101) */

static int32_t cell_new(bkport* g)
{
    if (g->cell_free >= 0) {
        int32_t c = g->cell_free;
        g->cell_free = g->cell_next[c];
        return c;
    }
    if (g->cell_used == g->cell_cap) {
        int32_t cap = g->cell_cap ? g->cell_cap * 2 : 1024;
        int32_t* a = (int32_t*)realloc(g->cell_node, (size_t)cap * sizeof(int32_t));
        if (a) g->cell_node = a;
        int32_t* b = (int32_t*)realloc(g->cell_next, (size_t)cap * sizeof(int32_t));
        if (b) g->cell_next = b;
        if (!a || !b) { g->oom = 1; return -1; }
        g->cell_cap = cap;
    }
    return g->cell_used++;
}

static void cell_delete(bkport* g, int32_t c)
{
    g->cell_next[c] = g->cell_free;
    g->cell_free = c;
}

static void set_orphan_front(bkport* g, int32_t i)
{
    g->parent[i] = P_ORPHAN;
    int32_t c = cell_new(g);
    if (c < 0) return;
    g->cell_node[c] = i;
    g->cell_next[c] = g->orphan_first;
    g->orphan_first = c;
}

static void set_orphan_rear(bkport* g, int32_t i)
{
    g->parent[i] = P_ORPHAN;
    int32_t c = cell_new(g);
    if (c < 0) return;
    g->cell_node[c] = i;
    if (g->orphan_last >= 0) g->cell_next[g->orphan_last] = c;
    else                     g->orphan_first = c;
    g->orphan_last = c;
    g->cell_next[c] = -1;
}

/* ------------------------------------------------------------------ init (maxflow.cpp:119-156) */

static void maxflow_init(bkport* g)
{
    g->qfirst[0] = g->qlast[0] = g->qfirst[1] = g->qlast[1] = -1;
    g->orphan_first = -1;
    g->time = 0;
    for (int32_t i = 0; i < g->n_nodes; ++i) {
        g->qnext[i] = Q_NONE;
        g->ts[i] = g->time;
        if (g->trcap[i] > 0) {
            g->is_sink[i] = 0;
            g->parent[i] = P_TERMINAL;
            set_active(g, i);
            g->dist[i] = 1;
        } else if (g->trcap[i] < 0) {
            g->is_sink[i] = 1;
            g->parent[i] = P_TERMINAL;
            set_active(g, i);
            g->dist[i] = 1;
        } else {
            g->parent[i] = P_NONE;
        }
    }
}

/* ------------------------------------------------------------------ augment (maxflow.cpp:244-311) */

static void augment(bkport* g, int32_t middle)
{
    int32_t i, a;
    double bottleneck = g->rcap[middle];
    /* 1a: bottleneck along the source tree (arcs parent->child are sisters of the parent arcs) */
    for (i = g->head[middle ^ 1];; i = g->head[a]) {
        a = g->parent[i];
        if (a == P_TERMINAL) break;
        if (bottleneck > g->rcap[a ^ 1]) bottleneck = g->rcap[a ^ 1];
    }
    if (bottleneck > g->trcap[i]) bottleneck = g->trcap[i];
    /* 1b: the sink tree */
    for (i = g->head[middle];; i = g->head[a]) {
        a = g->parent[i];
        if (a == P_TERMINAL) break;
        if (bottleneck > g->rcap[a]) bottleneck = g->rcap[a];
    }
    if (bottleneck > -g->trcap[i]) bottleneck = -g->trcap[i];

    /* 2a: push along the source tree */
    g->rcap[middle ^ 1] += bottleneck;
    g->rcap[middle] -= bottleneck;
    for (i = g->head[middle ^ 1];; i = g->head[a]) {
        a = g->parent[i];
        if (a == P_TERMINAL) break;
        g->rcap[a] += bottleneck;
        g->rcap[a ^ 1] -= bottleneck;
        if (!g->rcap[a ^ 1]) set_orphan_front(g, i);
    }
    g->trcap[i] -= bottleneck;
    if (!g->trcap[i]) set_orphan_front(g, i);
    /* 2b: the sink tree */
    for (i = g->head[middle];; i = g->head[a]) {
        a = g->parent[i];
        if (a == P_TERMINAL) break;
        g->rcap[a ^ 1] += bottleneck;
        g->rcap[a] -= bottleneck;
        if (!g->rcap[a]) set_orphan_front(g, i);
    }
    g->trcap[i] += bottleneck;
    if (!g->trcap[i]) set_orphan_front(g, i);

    g->flow += bottleneck;
}

/* ------------------------------------------------------------------ adoption (maxflow.cpp:316-467) */

/* One routine for both trees: `sink` selects which residual must be non-zero on the
 * candidate link (sink tree: i->j i.e. rcap[a0]; source tree: j->i i.e. rcap[a0^1]). */
static void process_orphan(bkport* g, int32_t i, int sink)
{
    int32_t a0, a0_min = -1, a, j;
    int32_t d, d_min = INFINITE_D;

    for (a0 = g->first[i]; a0 >= 0; a0 = g->next[a0]) {
        if (!(sink ? g->rcap[a0] : g->rcap[a0 ^ 1])) continue;
        j = g->head[a0];
        if ((int)g->is_sink[j] != sink || (a = g->parent[j]) == P_NONE) continue;
        /* does j still originate from the terminal? */
        d = 0;
        for (;;) {
            if (g->ts[j] == g->time) { d += g->dist[j]; break; }
            a = g->parent[j];
            d++;
            if (a == P_TERMINAL) { g->ts[j] = g->time; g->dist[j] = 1; break; }
            if (a == P_ORPHAN) { d = INFINITE_D; break; }
            j = g->head[a];
        }
        if (d < INFINITE_D) {
            if (d < d_min) { a0_min = a0; d_min = d; }
            for (j = g->head[a0]; g->ts[j] != g->time; j = g->head[g->parent[j]]) {
                g->ts[j] = g->time;
                g->dist[j] = d--;
            }
        }
    }

    if (a0_min >= 0) {
        g->parent[i] = a0_min;
        g->ts[i] = g->time;
        g->dist[i] = d_min + 1;
    } else {
        g->parent[i] = P_NONE;
        for (a0 = g->first[i]; a0 >= 0; a0 = g->next[a0]) {
            j = g->head[a0];
            if ((int)g->is_sink[j] != sink || (a = g->parent[j]) == P_NONE) continue;
            if (sink ? g->rcap[a0] : g->rcap[a0 ^ 1]) set_active(g, j);
            if (a != P_TERMINAL && a != P_ORPHAN && g->head[a] == i) set_orphan_rear(g, j);
        }
    }
}

/* ------------------------------------------------------------------ main loop (maxflow.cpp:472-604) */

double bkport_maxflow(void* h)
{
    bkport* g = (bkport*)h;
    int32_t i, j, a, current = -1;

    maxflow_init(g);

    for (;;) {
        if ((i = current) >= 0) {
            g->qnext[i] = Q_NONE; /* remove active flag */
            if (g->parent[i] == P_NONE) i = -1;
        }
        if (i < 0) {
            if ((i = next_active(g)) < 0) break;
        }

        /* growth */
        if (!g->is_sink[i]) {
            for (a = g->first[i]; a >= 0; a = g->next[a]) {
                if (!g->rcap[a]) continue;
                j = g->head[a];
                if (g->parent[j] == P_NONE) {
                    g->is_sink[j] = 0;
                    g->parent[j] = a ^ 1;
                    g->ts[j] = g->ts[i];
                    g->dist[j] = g->dist[i] + 1;
                    set_active(g, j);
                } else if (g->is_sink[j]) {
                    break;
                } else if (g->ts[j] <= g->ts[i] && g->dist[j] > g->dist[i]) {
                    g->parent[j] = a ^ 1; /* shorten j's path to the source */
                    g->ts[j] = g->ts[i];
                    g->dist[j] = g->dist[i] + 1;
                }
            }
        } else {
            for (a = g->first[i]; a >= 0; a = g->next[a]) {
                if (!g->rcap[a ^ 1]) continue;
                j = g->head[a];
                if (g->parent[j] == P_NONE) {
                    g->is_sink[j] = 1;
                    g->parent[j] = a ^ 1;
                    g->ts[j] = g->ts[i];
                    g->dist[j] = g->dist[i] + 1;
                    set_active(g, j);
                } else if (!g->is_sink[j]) {
                    a = a ^ 1; /* bridging arc must point source tree -> sink tree */
                    break;
                } else if (g->ts[j] <= g->ts[i] && g->dist[j] > g->dist[i]) {
                    g->parent[j] = a ^ 1;
                    g->ts[j] = g->ts[i];
                    g->dist[j] = g->dist[i] + 1;
                }
            }
        }

        g->time++;

        if (a >= 0) {
            g->qnext[i] = i; /* set active flag */
            current = i;

            augment(g, a);

            /* adoption: each orphan made by augment() is processed together with every
             * orphan it spawns before the next one is looked at (maxflow.cpp:573-589) */
            int32_t np, np_next;
            while ((np = g->orphan_first) >= 0) {
                np_next = g->cell_next[np];
                g->cell_next[np] = -1;
                while ((np = g->orphan_first) >= 0) {
                    g->orphan_first = g->cell_next[np];
                    i = g->cell_node[np];
                    cell_delete(g, np);
                    if (g->orphan_first < 0) g->orphan_last = -1;
                    process_orphan(g, i, g->is_sink[i]);
                }
                g->orphan_first = np_next;
            }
            if (g->oom) break;
        } else {
            current = -1;
        }
    }
    return g->flow;
}

/* ------------------------------------------------------------------ read-out */

/* graph.h:561-571 with default_segm = SOURCE(0); SINK = 1 */
int bkport_what_segment(void* h, int64_t i)
{
    bkport* g = (bkport*)h;
    return (g->parent[i] != P_NONE) ? (g->is_sink[i] ? 1 : 0) : 0;
}

/* bin/medpy_graphcut_voxel.py:177-181: 0 if SINK else 1 */
void bkport_labels(void* h, int64_t n, uint8_t* out)
{
    bkport* g = (bkport*)h;
    for (int64_t k = 0; k < n; ++k) out[k] = (g->parent[k] != P_NONE && g->is_sink[k]) ? 0 : 1;
}

double  bkport_get_trcap(void* h, int64_t i) { return ((bkport*)h)->trcap[i]; }
double  bkport_get_edge(void* h, int64_t i, int64_t j)
{
    bkport* g = (bkport*)h;
    int32_t a = get_arc(g, (int32_t)i, (int32_t)j);
    return a >= 0 ? g->rcap[a] : 0.0;
}
int64_t bkport_get_node_num(void* h) { return ((bkport*)h)->n_nodes; }
int64_t bkport_get_arc_num(void* h) { return ((bkport*)h)->n_arcs; }
int     bkport_oom(void* h) { return ((bkport*)h)->oom; }

/* residual graph read-out for the ambiguity check (oracle/cutcheck.py): all arcs in allocation order (sister arcs are
 * adjacent: a ^ 1) with their residual capacities, and the residual t-links; same contract as bkref_export */
void bkport_export(void* h, int32_t* tail, int32_t* head, double* rcap, double* trcap)
{
    const bkport* g = (const bkport*)h;
    for (int32_t a = 0; a < g->n_arcs; ++a) {
        tail[a] = g->head[a ^ 1];
        head[a] = g->head[a];
        rcap[a] = g->rcap[a];
    }
    for (int32_t i = 0; i < g->n_nodes; ++i) trcap[i] = g->trcap[i];
}
