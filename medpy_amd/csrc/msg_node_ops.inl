/*
 * msg_node_ops.inl -- node operations of the max-flow solver for ARBITRARY sparse graphs (the label / region graph
 * cut of reference medpy/graphcut/generate.py:177-338, voxel graphs of more than three dimensions, and graphs that
 * plug-in energy terms assemble edge by edge through GCGraph.set_nweight, graph.py:382-440).
 *
 * The reference solves these with the same BK code as the voxel graphs (lib/maxflow/src/maxflow.cpp:472-604).  Here:
 * synchronous push-relabel over a CSR residual graph, one thread per node, two kernels per round --
 *   push:   every active node (excess > 0, finite label) first saturates towards the sink, then pushes along its
 *           admissible arcs (label(head) == label - 1) and RECORDS each amount in the arc's own slot `delta`;
 *   gather: every node collects the amounts recorded on its incoming arcs (found through the reverse-arc index),
 *           then, if it still holds excess and has no admissible arc, relabels to 1 + min label over residual arcs.
 * No atomics: an arc's residual and delta slot are written by the arc's tail in `push` and by the same thread again in
 * `gather` (through rev) -- never by two threads in one kernel -- so the floating point result is deterministic.
 * Labels stay valid: pushes use the labels of the previous kernel unchanged; concurrent relabels only raise labels and
 * each uses a lower bound of its neighbours' new labels.
 * Global relabel = chaotic relaxation of label(u) = 1 + min label(head) over residual arcs (1 if the node still has
 * residual capacity to the sink) to the fixpoint; nodes left at MSG_HINF cannot reach the sink and form, at
 * convergence, exactly the set the reference reads out as SOURCE (graph.h:561-571).
 *
 * Single source for the HIP kernels (msg_sparse.hip) and the host simulator of the CPU tests
 * (tests/hostsim/hostsim_sparse.cpp).
 */
#ifndef MSG_NODE_OPS_INL
#define MSG_NODE_OPS_INL

#include <stdint.h>

#if defined(__HIPCC__)
#define MSG_HD __host__ __device__ __forceinline__
#else
#define MSG_HD inline
#endif

#define MSG_HINF 0x3f3f3f3f

struct MsgCsr {
    int64_t nodes, arcs;
    const int64_t* row;   /* [nodes + 1] first arc of each node                         */
    const int32_t* head;  /* [arcs] head node                                            */
    const int64_t* rev;   /* [arcs] index of the reverse arc                             */
    double* rcap;         /* [arcs] residual capacity                                    */
    double* delta;        /* [arcs] amount pushed along the arc in the current round     */
    double* excess;       /* [nodes]                                                     */
    double* sink;         /* [nodes] residual capacity node -> sink                      */
    int32_t* height;      /* [nodes] distance label, MSG_HINF = cannot reach the sink    */
    int32_t* count;       /* [4]: 0 = active nodes seen by the last gather, 1 = relabel pass changed something */
};

/* one node of the push kernel */
MSG_HD void msg_push_node(const MsgCsr& G, int64_t u)
{
    double e = G.excess[u];
    const int32_t h = G.height[u];
    if (!(e > 0.0) || h >= MSG_HINF) return;
    double s = G.sink[u];
    if (s > 0.0) {
        const double d = e < s ? e : s;
        e -= d;
        G.sink[u] = s - d;
    }
    for (int64_t a = G.row[u]; a < G.row[u + 1] && e > 0.0; ++a) {
        const double r = G.rcap[a];
        if (r > 0.0 && G.height[G.head[a]] == h - 1) {
            const double d = e < r ? e : r;
            e -= d;
            G.rcap[a] = r - d;
            G.delta[a] = d;
        }
    }
    G.excess[u] = e;
}

/* one node of the gather + relabel kernel; returns true when the node is still active afterwards */
MSG_HD bool msg_gather_node(const MsgCsr& G, int64_t v)
{
    double e = G.excess[v];
    for (int64_t a = G.row[v]; a < G.row[v + 1]; ++a) {
        const int64_t ra = G.rev[a];
        const double d = G.delta[ra];
        if (d != 0.0) {
            G.delta[ra] = 0.0;
            G.rcap[a] += d;
            e += d;
        }
    }
    G.excess[v] = e;
    const int32_t h = G.height[v];
    if (!(e > 0.0) || h >= MSG_HINF) return false;
    if (G.sink[v] > 0.0) return true; /* will push to the sink next round */
    int32_t best = MSG_HINF;
    for (int64_t a = G.row[v]; a < G.row[v + 1]; ++a)
        if (G.rcap[a] > 0.0) {
            const int32_t hn = G.height[G.head[a]]; /* may be mid-update by its owner: any value read is a lower bound of its new label */
            if (hn == h - 1) return true;           /* admissible arc: push next round */
            if (hn < best) best = hn;
        }
    const int32_t nh = best >= MSG_HINF ? MSG_HINF : best + 1;
    if (nh > h) G.height[v] = nh;
    return nh < MSG_HINF;
}

/* global relabel: start and one relaxation step of one node; returns true when the label went down */
MSG_HD void msg_relabel_init_node(const MsgCsr& G, int64_t u) { G.height[u] = G.sink[u] > 0.0 ? 1 : MSG_HINF; }

MSG_HD bool msg_relabel_relax_node(const MsgCsr& G, int64_t u)
{
    const int32_t h = G.height[u];
    if (h <= 1) return false;
    int32_t best = MSG_HINF;
    for (int64_t a = G.row[u]; a < G.row[u + 1]; ++a)
        if (G.rcap[a] > 0.0) {
            const int32_t hn = G.height[G.head[a]];
            if (hn < best) best = hn;
        }
    if (best < MSG_HINF && best + 1 < h) {
        G.height[u] = best + 1;
        return true;
    }
    return false;
}

MSG_HD bool msg_active_node(const MsgCsr& G, int64_t u) { return G.excess[u] > 0.0 && G.height[u] < MSG_HINF; }

/* schedule shared by the library and the simulator: Dev provides relabel_init(), relabel_pass() -> launches,
 * push(), gather(), zero_count(i), read_counts(int[4]) */
struct MsgSolveStats {
    int64_t rounds, relabels, relabel_passes, converged;
};

template <class Dev>
int msg_solve(Dev& dev, int rounds_per_relabel, int64_t max_rounds, MsgSolveStats& st)
{
    st = MsgSolveStats();
    int cnt[4];
    for (;;) {
        /* exact labels */
        dev.relabel_init();
        for (;;) {
            dev.zero_count(1);
            for (int b = 0; b < 8; ++b) { dev.relabel_pass(); st.relabel_passes++; }
            dev.read_counts(cnt);
            if (!cnt[1]) break;
        }
        st.relabels++;
        dev.zero_count(0);
        dev.count_active();
        dev.read_counts(cnt);
        if (!cnt[0]) { st.converged = 1; return 0; }
        if (st.rounds >= max_rounds) return 1;
        for (int r = 0; r < rounds_per_relabel; ++r) {
            dev.push();
            if (r + 1 == rounds_per_relabel || (r & 15) == 15) dev.zero_count(0);
            dev.gather();
            st.rounds++;
            if ((r & 15) == 15 && r + 1 < rounds_per_relabel) {
                dev.read_counts(cnt);
                if (!cnt[0]) break;
            }
        }
    }
}

#endif /* MSG_NODE_OPS_INL */
