/*
 * hostsim_sparse.cpp -- TEST ONLY.  Runs the sparse-graph solver's node operations and schedule
 * (medpy_amd/csrc/msg_node_ops.inl, the same source the k_msg_* kernels compile) on the host, so the CPU test tier can
 * check the algorithm against the BK oracle without a GPU.  The CSR assembly mirrors msg_sparse.hip:msg_build with
 * std::stable_sort in place of the device radix sort.
 */
#include <algorithm>
#include <numeric>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../medpy_amd/csrc/msg_node_ops.inl"

namespace {

struct HostSparse {
    MsgCsr G;
    std::vector<int64_t> row, rev;
    std::vector<int32_t> head, tail, height;
    std::vector<double> rcap, cap0, delta, excess, sink;
    int32_t count[4];

    void relabel_init() { for (int64_t u = 0; u < G.nodes; ++u) msg_relabel_init_node(G, u); }
    void relabel_pass() { for (int64_t u = 0; u < G.nodes; ++u) if (msg_relabel_relax_node(G, u)) count[1] = 1; }
    void push() { for (int64_t u = 0; u < G.nodes; ++u) msg_push_node(G, u); }
    void gather() { for (int64_t u = 0; u < G.nodes; ++u) if (msg_gather_node(G, u)) count[0] = 1; }
    void count_active() { for (int64_t u = 0; u < G.nodes; ++u) if (msg_active_node(G, u)) count[0] = 1; }
    void zero_count(int i) { count[i] = 0; }
    void read_counts(int* out) { memcpy(out, count, sizeof(count)); }
};

} // namespace

extern "C" int hostsim_sparse_solve(int64_t nodes, int64_t n_edges, const int64_t* ei, const int64_t* ej, const double* ecap, const double* erev,
                                    const uint8_t* eflag, const double* tr, int rounds_per_relabel, uint8_t* labels_out, double* cut_out,
                                    int64_t* stats_out)
{
    HostSparse d;
    const int64_t n2 = 2 * n_edges;
    std::vector<uint64_t> key((size_t)n2);
    std::vector<uint32_t> idx((size_t)n2);
    for (int64_t e = 0; e < n_edges; ++e) {
        key[2 * e] = ((uint64_t)ei[e] << 32) | (uint64_t)ej[e];
        key[2 * e + 1] = ((uint64_t)ej[e] << 32) | (uint64_t)ei[e];
    }
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    for (int64_t k = 0; k < n2;) {
        const uint64_t kk = key[idx[k]];
        double s = 0.0;
        bool once = false;
        int64_t m = k;
        for (; m < n2 && key[idx[m]] == kk; ++m) {
            const uint32_t a = idx[m];
            const double c = (a & 1) ? erev[a >> 1] : ecap[a >> 1];
            if (eflag && (eflag[a >> 1] & 1)) { if (!once) s += c; once = true; }
            else s += c;
        }
        d.head.push_back((int32_t)(kk & 0xffffffffu));
        d.tail.push_back((int32_t)(kk >> 32));
        d.cap0.push_back(s);
        k = m;
    }
    const int64_t A = (int64_t)d.head.size();
    d.row.assign((size_t)nodes + 1, 0);
    for (int64_t a = 0; a < A; ++a) d.row[(size_t)d.tail[a] + 1]++;
    for (int64_t u = 0; u < nodes; ++u) d.row[u + 1] += d.row[u];
    d.rev.assign((size_t)A, 0);
    for (int64_t a = 0; a < A; ++a) {
        const int32_t v = d.head[a], u = d.tail[a];
        const auto b = d.head.begin() + d.row[v], e = d.head.begin() + d.row[v + 1];
        d.rev[a] = std::lower_bound(b, e, u) - d.head.begin();
    }
    d.rcap = d.cap0;
    d.delta.assign((size_t)A, 0.0);
    d.excess.assign((size_t)nodes, 0.0);
    d.sink.assign((size_t)nodes, 0.0);
    d.height.assign((size_t)nodes, MSG_HINF);
    for (int64_t u = 0; u < nodes; ++u) {
        const double t = tr ? tr[u] : 0.0;
        d.excess[u] = t > 0 ? t : 0.0;
        d.sink[u] = t < 0 ? -t : 0.0;
    }
    memset(d.count, 0, sizeof(d.count));
    MsgCsr& G = d.G;
    G.nodes = nodes; G.arcs = A; G.row = d.row.data(); G.head = d.head.data(); G.rev = d.rev.data(); G.rcap = d.rcap.data();
    G.delta = d.delta.data(); G.excess = d.excess.data(); G.sink = d.sink.data(); G.height = d.height.data(); G.count = d.count;
    MsgSolveStats st;
    const int rc = msg_solve(d, rounds_per_relabel > 0 ? rounds_per_relabel : 64, (int64_t)1 << 40, st);
    double cut = 0.0;
    for (int64_t u = 0; u < nodes; ++u) labels_out[u] = d.height[u] < MSG_HINF ? 0 : 1;
    for (int64_t u = 0; u < nodes; ++u) {
        const double t = tr ? tr[u] : 0.0;
        if (labels_out[u]) {
            if (t < 0) cut += -t;
            for (int64_t a = d.row[u]; a < d.row[u + 1]; ++a)
                if (!labels_out[d.head[a]]) cut += d.cap0[a];
        } else if (t > 0) {
            cut += t;
        }
    }
    *cut_out = cut;
    stats_out[0] = st.rounds; stats_out[1] = st.relabels; stats_out[2] = st.relabel_passes; stats_out[3] = st.converged; stats_out[4] = A;
    return rc;
}
