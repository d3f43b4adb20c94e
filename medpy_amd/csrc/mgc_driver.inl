/*
 * mgc_driver.inl -- host-side orchestration of the lattice max-flow solve, written against a
 * "device policy" so the HIP backend (mgc_kernels.hip: one kernel launch per call) and the
 * host simulator of the CPU test tier (tests/hostsim) run the same schedule.
 *
 * Replaces (reference): the serial main loop of Graph::maxflow, lib/maxflow/src/maxflow.cpp:472-604.
 *
 * Schedule
 *   outer:  absorb in-flight outboxes -> GLOBAL RELABEL (tile BFS passes to a fixpoint: exact
 *           distance-to-sink labels, MGC_HINF = cannot reach the sink) -> collect tiles that hold
 *           excess able to reach the sink; none => the preflow is maximum and the labels ARE the cut.
 *   inner:  `rounds` rounds of two colour phases.  Tiles are 3-D checkerboard coloured, so the six
 *           face neighbours of a discharging tile are idle: region discharge is race-free and
 *           bit-reproducible without atomics on the flow data.
 *
 * The work lists and their lengths live on the device; kernels read the length themselves, so the
 * host launches passes / phases in batches and only reads the counter block back every
 * `relabel_batch` passes / `check_rounds` rounds (a pass over an empty list is a no-op).
 *
 * Dev concept (every call is asynchronous on the device's stream unless it returns a value):
 *   fill_heights_inf()  zero_count(i)  read_counts(int out[MGC_NCOUNT])   absorb_all()   suspect_pass()  suspect_batch()
 *   relabel_all(epoch, next_list)  relabel_list(list, epoch, next_list)  first_relabel_dt() -> bool
 *   activate_all(phase)  discharge(list, phase, max_cycles, max_sweeps)  range_push(name) / range_pop() (tracing ranges)
 *   radial labels (mgc_dt_ops.inl; only after first_relabel_dt() returned true):
 *   radial_begin(c_min) -> bool  (distance from the source, C into MGC_CNT_RADIAL_C, exact labels kept aside, labels lowered)
 *   radial_restore_exact()  radial_save_exact()  radial_lower(c_min)  source_open() (counts into MGC_CNT_SOURCE_OPEN)  set_radial(bool)
 */
#ifndef MGC_DRIVER_INL
#define MGC_DRIVER_INL

#include "mgc_common.h"

struct MgcSolveParams {
    int rounds_per_relabel; /* colour-phase rounds between two global relabels            */
    int max_cycles;         /* label/push cycles per tile discharge                       */
    int max_sweeps;         /* push sweeps per cycle                                      */
    int max_outer;          /* safety cap on global relabels                              */
    int relabel_batch;      /* BFS passes launched between two counter read-backs         */
    int check_rounds;       /* colour rounds launched between two counter read-backs      */
    int incremental_relabel;/* 1: later global relabels touch suspect tiles only          */
    int stop_below;         /* the colour rounds of a cycle end early when no more than this many tiles are queued (0: only when none are) */
    int trace;              /* 1: one stderr line per global relabel (tile visits of the relabel, discharges since the one before) */
    int adaptive_rounds;    /* k > 0: the number of rounds between two relabels doubles (up to 4x) while a relabel visits more
                               than k times as many tiles as the discharges of the cycle before it did                   */
    int radial;             /* 1: the flood phase runs on RADIAL labels (mgc_dt_ops.inl): after a first relabel by distance transform
                               the labels are min(exact, C - distance from the source) while excess of the source can still
                               reach the sink; exact labels from the relabel that finds the source sealed in            */
    int radial_min_c;       /* ... only when the shortest source -> sink path has at least this many hops               */
    int radial_rounds0;     /* colour rounds of the first radial cycle; 0: one cycle of the whole budget                 */
    int radial_budget_x16;  /* most colour rounds on radial labels, in sixteenths of (tiles on the shortest source -> sink path) */
};

struct MgcSolveStats {
    int64_t outer;            /* global relabels performed                                 */
    int64_t relabel_passes;   /* tile-BFS passes launched                                  */
    int64_t relabel_tiles;    /* tiles visited by those passes                             */
    int64_t phases;           /* colour phases launched                                    */
    int64_t discharge_tiles;  /* tile discharges                                           */
    int64_t converged;        /* 1 when the preflow is maximum                             */
    int64_t last_active;      /* active tiles found by the last activation pass            */
    int64_t readbacks;        /* counter read-backs (host syncs)                           */
    int64_t radial_cycles;    /* cycles of colour phases that ran on radial labels         */
};

/* where a solver variant keeps its lists and counters (6-neighbourhood: 2 colours, lists 0..3 + 4,5;
 * 26-neighbourhood: 8 colours, lists 0..15 + 16,17) */
struct MgcLayout {
    int ncolours;   /* colour phases per round                   */
    int list_mask;  /* discharge list of phase p = p & list_mask */
    int rl_base;    /* relabel lists rl_base, rl_base + 1        */
    int cnt_active, cnt_dis, cnt_rel;
    int incremental; /* global relabels after the first recompute only suspect tiles (Dev: suspect_pass, reset_suspect) */
    int rl_third;    /* third relabel list (rotation: a pass clears the counter of the list consumed one pass earlier, so
                        no memset sits between two passes) or -1 */
};

static inline MgcLayout mgc_layout6() { MgcLayout l = {2, 3, 4, 6, 8, 9, 1, 7}; return l; }
static inline MgcLayout mgc_layout26() { MgcLayout l = {8, 15, 16, 18, 19, 20, 1, -1}; return l; }

static inline MgcSolveParams mgc_default_params(int ndir = 6)
{
    MgcSolveParams p;
    /* tuned on MI355X at 256^3 / 512^3 (tools/gpu_sweep.py, tools/gpu_sweep26.py; every schedule gives the same labels).
     * One exact in-tile labelling per discharge, then sweeps with local relabels: 83 ms vs 100 ms for (3 cycles x 4
     * sweeps) at 512^3; a sweep of the 26-neighbourhood costs four times as much, so fewer of them pay there. */
    p.rounds_per_relabel = ndir == 26 ? 6 : 8;
    p.max_cycles = ndir == 26 ? -1 : 1; /* < 0: the stored labels (valid lower bounds) instead of an exact in-tile labelling */
    p.max_sweeps = ndir == 26 ? 3 : 12;
    p.max_outer = 100000;
    p.relabel_batch = 8;
    p.check_rounds = 8; /* (4 until round 5; with the flood phase on radial labels a cycle is 8 - 13 rounds and rarely ends early: 22.0 ms at 8, 22.5 at 4) */
    p.incremental_relabel = 1;
    p.stop_below = 0;
    p.trace = 0;
    p.adaptive_rounds = ndir == 26 ? 9 : 2; /* a tile visit of a relabel costs 1/3 of a discharge (8 vs 25 ns), 1/9 in the full neighbourhood (20 vs 175 ns);
                                               measured at 512^3 (round 3): weak contrast 68.7 ms at 3, 66.3 at 2, 66.3 at 1; headline volume 35.9 at 3 and 2, 39.8 at 1 */
    p.radial = ndir == 6 ? 2 : 0; /* 2: decided per graph by whoever calls mgc_solve (mgc_maxflow: wall tiles counted by k_build); the host simulator treats 2 as 1 */
    p.radial_min_c = 8;
    p.radial_budget_x16 = 8;
    p.radial_rounds0 = 0; /* 0: ONE radial cycle as long as the flood may take (radial_budget below).  Measured on MI355X, headline volume 512^3:
                             35.9 ms on exact labels; first radial cycle of 4 rounds (then 8, then 4, a relabel in between) 26.1 ms, 6: 23.4,
                             8: 24.4, 16 (= the budget, one cycle): 22.0 ms; 256^3: 9.5 / 5.3 (4) / 4.8 (8 = the budget).  A short first cycle
                             only pays where a cut hugs the source, and such graphs hold no walls to flood against (parameter radial = 2) */
    return p;
}

template <class Dev>
int mgc_solve(Dev& dev, const MgcLattice& L, const MgcSolveParams& P, MgcSolveStats& st, const MgcLayout lay = mgc_layout6())
{
    uint32_t phase = 4; /* stamps start at 0 */
    uint32_t rep = 2;   /* relabel epoch     */
    int cnt[MGC_NCOUNT];
    int rounds = P.rounds_per_relabel;
    int64_t prev_dis = 0, prev_rel = 0, last_passes = 0;
    bool radial = false; /* the labels in HBM are the radial ones: discharges mark every saturation, relabels start from the exact labels kept aside */
    int radial_done = 0, radial_next = P.radial_rounds0 > 0 ? P.radial_rounds0 : 1 << 20, radial_budget = 0; /* colour rounds run on radial labels / length of the next radial cycle / most such rounds */
    st = MgcSolveStats();
    dev.zero_count(lay.cnt_dis);
    dev.zero_count(lay.cnt_rel);

    for (int outer = 0; outer < P.max_outer; ++outer) {
        /* ---- global relabel ---- */
        dev.range_push("global relabel");
        dev.absorb_all();
        dev.zero_count(lay.rl_base);
        dev.zero_count(lay.rl_base + 1);
        if (lay.rl_third >= 0) dev.zero_count(lay.rl_third); /* (all clears of this stretch go out together, see HipDevT::flush_zero) */
        const bool by_transform = outer == 0 && dev.first_relabel_dt(); /* exact labels in six streaming scans: no passes at all */
        if (by_transform) {
            /* the flood phase on radial labels; whether the graph qualifies (C >= radial_min_c) is decided on the device, the host
             * reads C with the counters of the activation below */
            if (P.radial && lay.incremental && P.incremental_relabel) radial = dev.radial_begin(P.radial_min_c);
        } else if (outer == 0 || !lay.incremental || !P.incremental_relabel) {
            dev.fill_heights_inf();
            dev.relabel_all(rep + 1, lay.rl_base + (int)((rep + 1) & 1u));
        } else {
            /* which tiles could have lost the support of their labels? (tile-level closure, cheap passes) */
            for (;;) {
                dev.zero_count(MGC_CNT_CHANGED);
                for (int b = 0; b < dev.suspect_batch(); ++b) dev.suspect_pass();
                dev.read_counts(cnt);
                st.readbacks++;
                if (cnt[MGC_CNT_CHANGED] == 0) break;
            }
            if (radial) dev.radial_restore_exact(); /* everybody back on the exact labels of the last relabel; the suspect tiles are reset and recomputed from them */
            dev.reset_suspect(rep + 1, lay.rl_base + (int)((rep + 1) & 1u));
            if (P.trace) {
                dev.read_counts(cnt);
                fprintf(stderr, "[mgc] relabel %d: %d suspect tiles\n", outer, cnt[lay.rl_base + (int)((rep + 1) & 1u)]);
            }
        }
        st.relabel_passes++;
        if (by_transform) {
        } else if (lay.rl_third >= 0) {
            /* three lists rotate: pass k consumes lists[k % 3], appends to lists[(k + 1) % 3] and clears the counter of
             * lists[(k + 2) % 3] (consumed by pass k - 1) inside the kernel */
            const int lists[3] = {lay.rl_base, lay.rl_base + 1, lay.rl_third};
            int k = (int)((rep + 1) & 1u); /* where relabel_all / reset_suspect queued their tiles; the other two lists are empty */
            /* passes between two looks at the list length: the label wave of one global relabel needs about as many passes as
             * that of the relabel before it, so most of them go out in one stretch (a pass over an empty list costs ~4 us,
             * a look at the counters a stream drain) */
            int batch = outer > 0 && last_passes > P.relabel_batch ? (int)last_passes : P.relabel_batch;
            int64_t passes_now = 0;
            for (;;) {
                for (int b = 0; b < batch; ++b, ++k) {
                    rep++;
                    dev.relabel_list(lists[k % 3], rep + 1, lists[(k + 1) % 3], lists[(k + 2) % 3]);
                    st.relabel_passes++;
                    passes_now++;
                }
                dev.read_counts(cnt);
                st.readbacks++;
                if (cnt[lists[k % 3]] == 0) { /* the last pass woke nobody: fixpoint */
                    last_passes = passes_now - batch + 1; /* (the wave died somewhere inside the last stretch: what is known to have been needed) */
                    break;
                }
                batch = P.relabel_batch > 4 ? P.relabel_batch / 2 : P.relabel_batch;
            }
        } else {
            for (;;) {
                for (int b = 0; b < P.relabel_batch; ++b) {
                    rep++;
                    const int cur = lay.rl_base + (int)(rep & 1u), nxt = lay.rl_base + (int)((rep + 1) & 1u);
                    dev.zero_count(nxt);
                    dev.relabel_list(cur, rep + 1, nxt, -1);
                    st.relabel_passes++;
                }
                dev.read_counts(cnt);
                st.readbacks++;
                if (cnt[lay.rl_base + (int)((rep + 1) & 1u)] == 0) break; /* the last pass woke nobody: fixpoint */
            }
        }
        st.outer++;
        dev.range_pop();
        const bool after_flood = radial && outer > 0; /* (this relabel followed a cycle on radial labels: its size says nothing about the relabels to come) */
        if (radial && outer > 0 && radial_done >= radial_budget) radial = false; /* the flood has had its rounds: the exact labels of this relabel stay (nothing to keep aside, nobody to ask) */
        if (radial && outer > 0) { /* the labels are exact now: keep them, and ask whether excess of the source still reaches the sink */
            dev.radial_save_exact();
            dev.zero_count(MGC_CNT_SOURCE_OPEN);
            dev.source_open();
        }

        /* ---- who can still push towards the sink? ---- */
        dev.range_push("activation");
        phase += 2 * (uint32_t)(lay.list_mask + 1); /* fresh stamps: anything queued before the relabel is void */
        for (int i = 0; i <= lay.list_mask; ++i) dev.zero_count(i);
        dev.zero_count(lay.cnt_active);
        dev.activate_all(phase);
        dev.read_counts(cnt);
        st.readbacks++;
        st.last_active = cnt[lay.cnt_active];
        st.discharge_tiles = cnt[lay.cnt_dis];
        st.relabel_tiles = cnt[lay.cnt_rel];
        /* A relabel that costs several times what the discharges between two relabels cost is paid too often (weak
         * contrast, tie-heavy inputs: nearly every tile is suspect every time): let the discharges run longer.  Any schedule
         * reaches the same cut; measured at 512^3: `hard` 110 -> 95 ms at 16 rounds, the headline volume is best at 8. */
        {
            const int64_t d_dis = (int64_t)cnt[lay.cnt_dis] - prev_dis; /* discharges since the relabel before this one */
            const int64_t d_rel = (int64_t)cnt[lay.cnt_rel] - prev_rel; /* tile visits of the relabel that just ended  */
            if (P.trace) fprintf(stderr, "[mgc] relabel %d: %lld tile visits in %lld passes so far, %lld discharges before it, %d active tiles\n", outer, (long long)d_rel, (long long)st.relabel_passes, (long long)d_dis, cnt[lay.cnt_active]);
            if (P.adaptive_rounds > 0 && outer > 0 && !radial && !after_flood && d_rel > (int64_t)P.adaptive_rounds * d_dis && rounds < 4 * P.rounds_per_relabel) rounds *= 2;
            prev_dis = cnt[lay.cnt_dis];
            prev_rel = cnt[lay.cnt_rel];
        }
        dev.range_pop();
        if (cnt[lay.cnt_active] == 0) {
            st.converged = 1;
            return 0;
        }
        if (radial) {
            if (outer == 0) {
                if (cnt[MGC_CNT_RADIAL_C] >= MGC_HINF || cnt[MGC_CNT_RADIAL_C] < P.radial_min_c) radial = false; /* (the device left the labels alone for the same reason) */
                /* a flood front moves a tile per colour phase, and nothing a shortest path's length away from the source is still
                 * "behind the cut": half that many ROUNDS (two phases each) reach the far side of a cut that surrounds the source.
                 * Measured on MI355X, headline volume (C = 206 hops = 26 tiles): budget 13 rounds 19.7 ms, 16: 22.5, 20: 22.2, 23: 24.3,
                 * 26: 25.2; 256^3 (C = 103): 7 rounds 4.44 ms, 8: 4.79, 10: 5.31, 13: 6.0 -- what the flood has not closed by then
                 * are holes the exact labels find faster */
                else {
                    radial_budget = (P.radial_budget_x16 * cnt[MGC_CNT_RADIAL_C] / 16 + 7) / 8;
                    if (radial_next > radial_budget) radial_next = radial_budget;
                }
            } else if (cnt[MGC_CNT_SOURCE_OPEN] == 0 || radial_done >= radial_budget) {
                /* the source is sealed in -- or the flood has had its time, and what is still open are holes that only exact
                 * labels find (radial labels lead past them: measured, a solve that re-lowers for ever) */
                radial = false;
            } else {
                dev.radial_lower(P.radial_min_c);
                radial_next = 2 * radial_next < radial_budget - radial_done ? 2 * radial_next : radial_budget - radial_done;
            }
            if (P.trace) fprintf(stderr, "[mgc] relabel %d: radial labels %s (shortest source -> sink path %d hops, %d source tiles open)\n", outer, radial ? "on" : "off", cnt[MGC_CNT_RADIAL_C], outer ? cnt[MGC_CNT_SOURCE_OPEN] : -1);
        }
        dev.set_radial(radial);

        /* ---- colour phases ---- */
        dev.range_push("colour phases");
        const int rounds_now = radial ? radial_next : rounds;
        if (radial) { st.radial_cycles++; radial_done += rounds_now; }
        for (int r = 0; r < rounds_now; ++r) {
            for (int c = 0; c < lay.ncolours; ++c) {
                const int lst = (int)(phase & (uint32_t)lay.list_mask);
                dev.discharge(lst, phase, P.max_cycles, P.max_sweeps);
                dev.zero_count(lst);
                st.phases++;
                phase++;
            }
            if ((r + 1) % P.check_rounds == 0 && r + 1 < rounds_now) {
                dev.read_counts(cnt);
                st.readbacks++;
                int pending = 0;
                for (int i = 0; i <= lay.list_mask; ++i) pending += cnt[i];
                if (pending <= P.stop_below) break; /* (a few stragglers: their tiles keep their excess flag and come back after the relabel) */
            }
        }
        dev.range_pop();
    }
    {   /* not converged within max_outer: leave the work counters of the truncated run */
        dev.read_counts(cnt);
        st.discharge_tiles = cnt[lay.cnt_dis];
        st.relabel_tiles = cnt[lay.cnt_rel];
    }
    return 1;
}

#endif /* MGC_DRIVER_INL */
