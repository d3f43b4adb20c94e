/*
 * oracle/cutcheck.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Which voxels may a correct max-flow solver label either way?  Given the residual graph of a maximum flow (from the
 * BK oracle, oracle/bk.py:BKGraph.export), every minimum cut (S, T) satisfies
 *        R_s  <=  S  <=  V \ R_t
 * where R_s = nodes reachable FROM the source and R_t = nodes that can REACH the sink through residual arcs
 * (Ford-Fulkerson / Picard-Queyranne).  The reference reports T = R_t (what_segment(), graph.h:561-571), and in exact
 * arithmetic R_t is unique.  In floating point a residual that is 0 in one summation order is a few ulp in another, so
 * two correct solvers can disagree exactly on the nodes whose membership hinges on such arcs.  This file computes R_s
 * and R_t with arcs whose residual is below the rounding granularity treated as saturated; the AMBIGUITY SET is
 * V \ (R_s u R_t).  A voxel outside it on which a solver disagrees with the reference is a bug, not a tie.
 *
 * Rounding granularity of an arc u -> v: tol_rel times the largest arc-pair capacity at u or v (the pair's own
 * included).  The excess of a node is a sum of flows of that size, so it is only known to that many ulp; a residual
 * far below it -- the reference floors zero weights at DBL_MIN = 2.2e-308 (energy_voxel.py:113), next to weights of
 * order 1 -- cannot be told from a saturated arc by ANY floating point solver: pushing 2.2e-308 out of an excess of 0.67
 * leaves the excess unchanged, so such arcs saturate "for free" in one operation order and stay open in another.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* returns 0, or 1 when out of memory.  tail/head/rcap: n_arcs arcs, sister of arc a is a ^ 1 (n_arcs even).
 * trcap: residual t-links (> 0 source -> node, < 0 node -> sink); tol_t: absolute tolerance on them. */
int cut_reach(int64_t n_nodes, int64_t n_arcs, const int32_t* tail, const int32_t* head, const double* rcap, const double* trcap,
              double tol_rel, double tol_t, uint8_t* from_source, uint8_t* to_sink)
{
    int64_t* start = (int64_t*)calloc((size_t)n_nodes + 1, sizeof(int64_t));
    int32_t* adj = (int32_t*)malloc((size_t)(n_arcs ? n_arcs : 1) * sizeof(int32_t));
    int32_t* queue = (int32_t*)malloc((size_t)(n_nodes ? n_nodes : 1) * sizeof(int32_t));
    uint8_t* open = (uint8_t*)malloc((size_t)(n_arcs ? n_arcs : 1));
    if (!start || !adj || !queue || !open) { free(start); free(adj); free(queue); free(open); return 1; }
    {
        double* scale = (double*)calloc((size_t)(n_nodes ? n_nodes : 1), sizeof(double)); /* largest pair capacity at the node */
        if (!scale) { free(start); free(adj); free(queue); free(open); return 1; }
        for (int64_t a = 0; a < n_arcs; ++a) {
            const double pair = rcap[a] + rcap[a ^ 1];
            if (pair > scale[tail[a]]) scale[tail[a]] = pair;
            if (pair > scale[head[a]]) scale[head[a]] = pair;
        }
        for (int64_t a = 0; a < n_arcs; ++a) {
            const double st = scale[tail[a]], sh = scale[head[a]];
            open[a] = rcap[a] > tol_rel * (st > sh ? st : sh) && rcap[a] > 0.0;
        }
        free(scale);
    }
    for (int pass = 0; pass < 2; ++pass) {
        /* pass 0: forward from the source along open arcs (key = tail); pass 1: backward from the sink (key = head) */
        const int32_t* key = pass == 0 ? tail : head;
        const int32_t* other = pass == 0 ? head : tail;
        uint8_t* mark = pass == 0 ? from_source : to_sink;
        memset(start, 0, ((size_t)n_nodes + 1) * sizeof(int64_t));
        for (int64_t a = 0; a < n_arcs; ++a)
            if (open[a]) start[key[a] + 1]++;
        for (int64_t i = 0; i < n_nodes; ++i) start[i + 1] += start[i];
        {
            int64_t* fill = (int64_t*)malloc((size_t)(n_nodes ? n_nodes : 1) * sizeof(int64_t));
            if (!fill) { free(start); free(adj); free(queue); free(open); return 1; }
            memcpy(fill, start, (size_t)n_nodes * sizeof(int64_t));
            for (int64_t a = 0; a < n_arcs; ++a)
                if (open[a]) adj[fill[key[a]]++] = other[a];
            free(fill);
        }
        memset(mark, 0, (size_t)n_nodes);
        int64_t qh = 0, qt = 0;
        for (int64_t i = 0; i < n_nodes; ++i)
            if (pass == 0 ? trcap[i] > tol_t : trcap[i] < -tol_t) { mark[i] = 1; queue[qt++] = (int32_t)i; }
        while (qh < qt) {
            const int32_t u = queue[qh++];
            for (int64_t k = start[u]; k < start[u + 1]; ++k) {
                const int32_t v = adj[k];
                if (!mark[v]) { mark[v] = 1; queue[qt++] = v; }
            }
        }
    }
    free(start); free(adj); free(queue); free(open);
    return 0;
}
