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
 *   Z-slabs (a Dev that stands for the slabs of one volume, MgcSlabGroup below): multi() -> bool  exchange(kind, epoch, list)  (border
 *   messages of kind 0 labels / 1 labels + outbox flow / 2 suspect flags to the neighbour slabs; a single device: multi() is false and
 *   exchange() is never called).  read_counts() of such a Dev returns the counters SUMMED OVER ALL SLABS, so every rank takes every
 *   decision of the schedule alike.
 */
#ifndef MGC_DRIVER_INL
#define MGC_DRIVER_INL

#include "mgc_common.h"
#include "../../include/medpy_hip.h" /* mgc_transport */

#include <vector>

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
    int exchange_passes;    /* Z-slabs: relabel passes between two exchanges of the border labels */
    int exchange_rounds;    /* Z-slabs: colour rounds between two exchanges of the border labels + outbox flow (6-neighbourhood; the full
                               neighbourhood pushes into the ghost tiles in place and exchanges after every phase) */
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
    int64_t deferred_drains;  /* Z-slabs: extra exchanges that carried what a full border message had left behind */
};

/* where a solver variant keeps its lists and counters (6-neighbourhood: 2 colours, lists 0..3 + 4,5;
 * 26-neighbourhood: 8 colours, lists 0..15 + 16,17) */
struct MgcLayout {
    int ncolours;   /* colour phases per round                   */
    int list_mask;  /* discharge list of phase p = p & list_mask */
    int rl_base;    /* relabel lists rl_base, rl_base + 1        */
    int cnt_active, cnt_dis, cnt_rel;
    int incremental; /* global relabels after the first recompute only suspect tiles (Dev: suspect_pass, reset_suspect) */
    int cnt_radial_c, cnt_source_open; /* counter slots of the radial labels (mgc_cnt_radial_c / mgc_cnt_source_open, mgc_common.h) */
    int rl_third;    /* third relabel list (rotation: a pass clears the counter of the list consumed one pass earlier, so
                        no memset sits between two passes) or -1 */
};

static inline MgcLayout mgc_layout6() { MgcLayout l = {2, 3, 4, 6, 8, 9, 1, MGC_CNT_RADIAL_C, MGC_CNT_SOURCE_OPEN, 7}; return l; }
static inline MgcLayout mgc_layout26() { MgcLayout l = {8, 15, 16, 18, 19, 20, 1, MGC26_CNT_RADIAL_C, MGC26_CNT_SOURCE_OPEN, -1}; return l; }

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
    /* measured with eight slabs of 2048 x 1024 x 1024 time-multiplexed on one MI355X (profiles/r6_slab_exchange_cadence.jsonl; the label
     * SHA-256 is the single handle's in every row): labels every 4 passes + flow every round 372 exchanges, 124 ms of kernels on the
     * busiest slab; every 8 passes 297 / 126 ms; every 8 passes and every 2nd round 226 / 123 ms */
    p.exchange_passes = 8;
    p.exchange_rounds = 2;
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
        if (dev.multi() && outer > 0) { /* flow a full border message left behind during the colour phases must have crossed before the masks are read */
            for (;;) {
                dev.read_counts(cnt);
                st.readbacks++;
                if (cnt[MGC_CNT_DEFERRED] == 0) break;
                dev.zero_count(MGC_CNT_DEFERRED);
                dev.exchange(1, phase - 1, 0);
                st.deferred_drains++;
            }
        }
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
                if (dev.multi()) dev.exchange(2, 0, 0); /* the DIRTY / SUSPECT flags of the border tiles: the closure crosses the slab borders */
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
            /* Z-slabs: the label wave of a global relabel has to cross every slab border on its way through the volume, so the border
             * labels travel every `xk` passes whatever the slabs' state (labels only go down during a relabel: a ghost label is an upper
             * bound whenever it is read, and a wave that reaches a border is on the other side at most xk passes later); what an
             * exchange wakes goes to the list the next pass consumes.  Every rank runs the same passes (one over an empty list is a
             * no-op of microseconds) and the ranks compare notes at the end of a stretch, behind one more exchange. */
            const bool multi = dev.multi();
            const int xk = P.exchange_passes > 0 ? P.exchange_passes : 4;
            for (;;) {
                if (multi) dev.zero_count(MGC_CNT_DEFERRED); /* (counts what the exchanges of this stretch leave behind) */
                int since = 0;
                for (int b = 0; b < batch; ++b, ++k) {
                    rep++;
                    dev.relabel_list(lists[k % 3], rep + 1, lists[(k + 1) % 3], lists[(k + 2) % 3]);
                    st.relabel_passes++;
                    passes_now++;
                    if (multi && (++since == xk || b + 1 == batch)) { dev.exchange(0, rep + 1, lists[(k + 1) % 3]); since = 0; }
                }
                dev.read_counts(cnt);
                st.readbacks++;
                if (cnt[lists[k % 3]] == 0 && !(multi && cnt[MGC_CNT_DEFERRED])) { /* the last pass (and the exchange behind it) woke nobody, nothing was left behind: fixpoint */
                    last_passes = passes_now - batch + 1; /* (the wave died somewhere inside the last stretch: what is known to have been needed) */
                    break;
                }
                batch = P.relabel_batch > 4 ? P.relabel_batch / 2 : P.relabel_batch;
            }
        } else {
            const bool multi = dev.multi();
            const int xk = P.exchange_passes > 0 ? P.exchange_passes : 4;
            for (;;) {
                if (multi) dev.zero_count(MGC_CNT_DEFERRED);
                int since = 0;
                for (int b = 0; b < P.relabel_batch; ++b) {
                    rep++;
                    const int cur = lay.rl_base + (int)(rep & 1u), nxt = lay.rl_base + (int)((rep + 1) & 1u);
                    dev.zero_count(nxt);
                    dev.relabel_list(cur, rep + 1, nxt, -1);
                    st.relabel_passes++;
                    if (multi && (++since == xk || b + 1 == P.relabel_batch)) { dev.exchange(0, rep + 1, nxt); since = 0; }
                }
                dev.read_counts(cnt);
                st.readbacks++;
                if (cnt[lay.rl_base + (int)((rep + 1) & 1u)] == 0 && !(multi && cnt[MGC_CNT_DEFERRED])) break; /* the last pass woke nobody: fixpoint */
            }
        }
        /* the full neighbourhood has no transform towards the sink (a Chebyshev distance is not separable): its first relabel ran as
         * passes, and the radial labels -- which only need the L1 transform AWAY from the source, mgc_radial_steps -- are put on top of
         * the exact ones here */
        if (outer == 0 && !by_transform && P.radial && lay.incremental && P.incremental_relabel && dev.radial_after_passes()) radial = dev.radial_begin(P.radial_min_c);
        st.outer++;
        dev.range_pop();
        const bool after_flood = radial && outer > 0; /* (this relabel followed a cycle on radial labels: its size says nothing about the relabels to come) */
        if (radial && outer > 0 && radial_done >= radial_budget) radial = false; /* the flood has had its rounds: the exact labels of this relabel stay (nothing to keep aside, nobody to ask) */
        if (radial && outer > 0) { /* the labels are exact now: keep them, and ask whether excess of the source still reaches the sink */
            dev.radial_save_exact();
            dev.zero_count(lay.cnt_source_open);
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
                if (cnt[lay.cnt_radial_c] >= MGC_HINF || cnt[lay.cnt_radial_c] < P.radial_min_c) radial = false; /* (the device left the labels alone for the same reason) */
                /* a flood front moves a tile per colour phase, and nothing a shortest path's length away from the source is still
                 * "behind the cut": half that many ROUNDS (two phases each) reach the far side of a cut that surrounds the source.
                 * Measured on MI355X, headline volume (C = 206 hops = 26 tiles): budget 13 rounds 19.7 ms, 16: 22.5, 20: 22.2, 23: 24.3,
                 * 26: 25.2; 256^3 (C = 103): 7 rounds 4.44 ms, 8: 4.79, 10: 5.31, 13: 6.0 -- what the flood has not closed by then
                 * are holes the exact labels find faster */
                else {
                    radial_budget = (P.radial_budget_x16 * cnt[lay.cnt_radial_c] / 16 + 7) / 8;
                    if (radial_next > radial_budget) radial_next = radial_budget;
                }
            } else if (cnt[lay.cnt_source_open] == 0 || radial_done >= radial_budget) {
                /* the source is sealed in -- or the flood has had its time, and what is still open are holes that only exact
                 * labels find (radial labels lead past them: measured, a solve that re-lowers for ever) */
                radial = false;
            } else {
                dev.radial_lower(P.radial_min_c);
                radial_next = 2 * radial_next < radial_budget - radial_done ? 2 * radial_next : radial_budget - radial_done;
            }
            if (P.trace) fprintf(stderr, "[mgc] relabel %d: radial labels %s (shortest source -> sink path %d hops, %d source tiles open)\n", outer, radial ? "on" : "off", cnt[lay.cnt_radial_c], outer ? cnt[lay.cnt_source_open] : -1);
        }
        dev.set_radial(radial);

        /* ---- colour phases ---- */
        dev.range_push("colour phases");
        const int rounds_now = radial ? radial_next : rounds;
        if (radial) { st.radial_cycles++; radial_done += rounds_now; }
        /* Z-slabs: border labels + outbox flow cross once per ROUND of the two colours (6-neighbourhood; every `exchange_rounds` rounds):
         * what a tile pushed over the slab border waits in its outbox until then -- region discharge only ever assumes a neighbour's
         * labels and outbox as of SOME earlier moment -- and mgc_halo_unpack_tile queues the receiving tile for the next phase of ITS
         * colour.  The full neighbourhood pushes into the ghost tiles in place and exchanges after every phase. */
        const bool multi = dev.multi();
        const int xr = P.exchange_rounds > 0 ? P.exchange_rounds : 1;
        if (multi) dev.zero_count(MGC_CNT_DEFERRED);
        for (int r = 0; r < rounds_now; ++r) {
            for (int c = 0; c < lay.ncolours; ++c) {
                const int lst = (int)(phase & (uint32_t)lay.list_mask);
                dev.discharge(lst, phase, P.max_cycles, P.max_sweeps);
                dev.zero_count(lst);
                if (multi && (lay.ncolours != 2 || (c == 1 && ((r + 1) % xr == 0 || r + 1 == rounds_now)))) dev.exchange(1, phase, 0);
                st.phases++;
                phase++;
            }
            if ((r + 1) % P.check_rounds == 0 && r + 1 < rounds_now) {
                dev.read_counts(cnt);
                st.readbacks++;
                int pending = multi ? cnt[MGC_CNT_DEFERRED] : 0;
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

/* ---------------------------------------------------------------------------------------------------------------------
 * Z-SLABS (no reference counterpart: the reference is single-process and its only splitter, wrapper.py:72-204, is
 * approximate).  The slabs of one volume as ONE Dev of mgc_solve: every operation of the schedule goes to each LOCAL slab (the one
 * slab of this rank on its GPU; or all slabs of the volume when they are time-multiplexed on one GPU or run on the host simulator),
 * the counters come back summed over all slabs of the volume, and the border messages / the carry planes of the distance transforms
 * move through a transport X:
 *     x.local_only()                      every slab of the volume is local (nothing crosses a process border)
 *     x.exchange(kind, epoch, list)       border messages of all local slabs to / from their neighbours (mgc_halo_pack_tile / unpack)
 *     x.allreduce(int64_t* v, n, op)      over all ranks, op 0 = sum, 1 = min (local_only: nothing to do)
 *     x.carry_in(slab, dir) -> const uint16_t*   the carry plane slab's z-scan in direction dir (0 up, 1 down) starts from, ready on
 *                                         the slab's stream; nullptr at the end of the volume
 *     x.carry_out(slab, dir)              the slab's scan has written its carry plane (Slab::carry_buf(dir)): hand it on
 * Slab (HipDevT / the simulator's HostDev) adds to the Dev concept:
 *     dt_applicable() -> bool   dt_scans_xy(seed)   dt_scan_z(bwd, final, c_min, carry_in, want_carry_out)   dt_finish()   shadow_sync()
 *     radial_prepare() -> bool  radial_cmin()  count_get(i) -> int  count_set(i, v)  radial_swap()
 * With one exchange per round of the two colours and the transforms carried across the borders the slabs run the schedule of the
 * single handle -- first relabel by distance transform, flood on radial labels, incremental relabels -- on their share of the tiles.
 * ------------------------------------------------------------------------------------------------------------------- */
template <class Slab, class X>
struct MgcSlabGroup {
    std::vector<Slab*>& d; /* the local slabs in ascending rank order */
    X& x;
    int64_t exchanges = 0, reductions = 0;
    MgcSlabGroup(std::vector<Slab*>& slabs, X& xchg) : d(slabs), x(xchg) {}
    bool multi() const { return true; }
    bool labels_inexact() const { return false; }
    int suspect_batch() const { return d[0]->suspect_batch(); }
    void range_push(const char* name) { d[0]->range_push(name); }
    void range_pop() { d[0]->range_pop(); }
    void fill_heights_inf() { for (Slab* p : d) p->fill_heights_inf(); }
    void zero_count(int i) { for (Slab* p : d) p->zero_count(i); }
    void absorb_all() { for (Slab* p : d) p->absorb_all(); }
    void suspect_pass() { for (Slab* p : d) p->suspect_pass(); }
    void relabel_all(uint32_t epoch, int next) { for (Slab* p : d) p->relabel_all(epoch, next); }
    void relabel_list(int lst, uint32_t epoch, int next, int zero_list = -1) { for (Slab* p : d) p->relabel_list(lst, epoch, next, zero_list); }
    void reset_suspect(uint32_t epoch, int list) { for (Slab* p : d) p->reset_suspect(epoch, list); }
    void activate_all(uint32_t phase) { for (Slab* p : d) p->activate_all(phase); }
    void discharge(int lst, uint32_t phase, int cycles, int sweeps) { for (Slab* p : d) p->discharge(lst, phase, cycles, sweeps); }
    void set_radial(bool on) { for (Slab* p : d) p->set_radial(on); }
    void radial_save_exact() { for (Slab* p : d) p->radial_save_exact(); }
    void radial_restore_exact() { for (Slab* p : d) p->radial_restore_exact(); }
    void radial_lower(int c_min) { for (Slab* p : d) p->radial_lower(c_min); }
    void source_open() { for (Slab* p : d) p->source_open(); }
    bool radial_after_passes() const { return d[0]->radial_after_passes(); }
    void exchange(int kind, uint32_t epoch, int list) { x.exchange(kind, epoch, list); exchanges++; }
    /* the counter block summed over every slab of the volume (MGC_CNT_RADIAL_C is the same word on every slab, see radial_begin) */
    void read_counts(int* out)
    {
        int64_t g[MGC_NCOUNT];
        int c[MGC_NCOUNT];
        for (int i = 0; i < MGC_NCOUNT; ++i) g[i] = 0;
        int radial_c = MGC_HINF;
        const int slot_c = mgc_cnt_radial_c(d[0]->lattice());
        for (Slab* p : d) {
            p->read_counts(c);
            for (int i = 0; i < MGC_NCOUNT; ++i) g[i] += c[i];
            radial_c = c[slot_c];
        }
        g[slot_c] = 0;
        if (!x.local_only()) x.allreduce(g, MGC_NCOUNT, 0);
        reductions++;
        for (int i = 0; i < MGC_NCOUNT; ++i) out[i] = g[i] > 0x7fffffff ? 0x7fffffff : (int)g[i];
        out[slot_c] = radial_c;
    }
    /* the six scans of a transform over all slabs: x and y stay inside a plane; the z-scans run slab after slab, each starting
     * from the carry plane of the slab before (a pipeline: ranks wait for their neighbour's plane on the stream, not on the host) */
    void transform(int seed, int final_kind, int c_min)
    {
        for (Slab* p : d) p->dt_scans_xy(seed);
        for (size_t i = 0; i < d.size(); ++i) { /* upwards */
            const uint16_t* cin = x.carry_in(*d[i], 0);
            const bool more = d[i]->sends_carry(0);
            d[i]->dt_scan_z(false, 0, 0, cin, more);
            if (more) x.carry_out(*d[i], 0);
        }
        for (size_t i = d.size(); i-- > 0;) { /* downwards: the scan that writes the labels */
            const uint16_t* cin = x.carry_in(*d[i], 1);
            const bool more = d[i]->sends_carry(1);
            d[i]->dt_scan_z(true, final_kind, c_min, cin, more);
            if (more) x.carry_out(*d[i], 1);
        }
    }
    bool first_relabel_dt()
    {
        int64_t ok = 1;
        for (Slab* p : d) ok = ok && p->dt_applicable() ? 1 : 0;
        if (!x.local_only()) x.allreduce(&ok, 1, 1);
        if (!ok) return false;
        transform(1, 1, 0);
        for (Slab* p : d) { p->dt_finish(); p->shadow_sync(); }
        return true;
    }
    bool radial_begin(int c_min)
    {
        int64_t ok = 1;
        for (Slab* p : d) ok = ok && p->radial_prepare() ? 1 : 0;
        if (!x.local_only()) x.allreduce(&ok, 1, 1);
        if (!ok) return false;
        /* C = hops of the shortest source -> sink path of the WHOLE volume: the smallest exact label a source voxel carries anywhere */
        int64_t c = MGC_HINF;
        for (Slab* p : d) { p->radial_cmin(); const int v = p->count_get(mgc_cnt_radial_c(p->lattice())); c = v < c ? v : c; }
        if (!x.local_only()) x.allreduce(&c, 1, 1);
        for (Slab* p : d) p->count_set(mgc_cnt_radial_c(p->lattice()), (int)c);
        transform(2, 2, c_min);
        for (Slab* p : d) { p->radial_swap(); p->shadow_sync(); } /* (both sides of a border lowered their copies alike) */
        return true;
    }
};

/* The transport of a slab group.  Either every slab of the volume is local -- messages move between the slabs' own buffers, nothing to
 * reduce -- or this rank holds ONE slab and its neighbours live in other processes: over the slab's native channel when it has one
 * (Slab::has_comm(): RCCL send / receive / all-reduce over xGMI inside the library, stream-ordered), else through the caller's
 * callbacks (mgc_transport, include/medpy_hip.h: host buffers; the CPU test tier over gloo / a directory of files, a development
 * mode of bench.py).  Slab adds to what MgcSlabGroup asks for:
 *     halo_msg_bytes(kind)  halo_pack(side, kind) -> void* (the slab's send buffer)  halo_unpack(side, kind, buf, epoch, list)
 *     recv_buf(side) -> void*  to_host(host, buf, n)  from_host(buf, host, n)  carry_buf(dir)  carry_recv_buf(dir)  carry_bytes()
 *     has_lower() has_upper() needs_carry(dir)  has_comm()  native_exchange(kind, epoch, list)  native_allreduce(v, n, op)
 *     native_send(side, buf, n)  native_recv(side, buf, n) */
template <class Slab>
struct MgcXchg {
    std::vector<Slab*>& d;
    const mgc_transport* cb;
    bool local;
    int error = 0; /* first failure of a native call or a callback (the schedule runs on; the caller reports it) */
    std::vector<char> hs[2], hr[2];
    MgcXchg(std::vector<Slab*>& slabs, const mgc_transport* callbacks, bool all_local) : d(slabs), cb(callbacks), local(all_local) {}
    bool local_only() const { return local; }
    void fail(int rc) { if (rc && !error) error = rc; }
    void exchange(int kind, uint32_t epoch, int list)
    {
        if (local) { /* every border of the volume: pack both sides of all of them, then unpack */
            std::vector<void*> up(d.size(), nullptr), dn(d.size(), nullptr);
            for (size_t i = 0; i + 1 < d.size(); ++i) {
                up[i] = d[i]->halo_pack(1, kind);
                dn[i] = d[i + 1]->halo_pack(0, kind);
            }
            for (size_t i = 0; i + 1 < d.size(); ++i) {
                d[i + 1]->halo_unpack(0, kind, up[i], epoch, list);
                d[i]->halo_unpack(1, kind, dn[i], epoch, list);
            }
            return;
        }
        Slab& s = *d[0];
        if (s.has_comm()) { fail(s.native_exchange(kind, epoch, list)); return; }
        const int64_t nb = s.halo_msg_bytes(kind);
        const bool has[2] = {s.has_lower(), s.has_upper()};
        for (int side = 0; side < 2; ++side)
            if (has[side]) {
                hs[side].resize((size_t)nb); hr[side].resize((size_t)nb);
                s.to_host(hs[side].data(), s.halo_pack(side, kind), nb);
            }
        fail(cb->exchange(cb->ctx, has[0] ? hs[0].data() : nullptr, has[0] ? hr[0].data() : nullptr, has[1] ? hs[1].data() : nullptr, has[1] ? hr[1].data() : nullptr, nb));
        for (int side = 0; side < 2; ++side)
            if (has[side]) {
                s.from_host(s.recv_buf(side), hr[side].data(), nb);
                s.halo_unpack(side, kind, s.recv_buf(side), epoch, list);
            }
    }
    void allreduce(int64_t* v, int n, int op)
    {
        if (local) return;
        if (d[0]->has_comm()) fail(d[0]->native_allreduce(v, n, op));
        else fail(cb->allreduce(cb->ctx, v, n, op));
    }
    /* dir 0: the scan runs upwards, the plane comes from the slab below; dir 1: downwards, from the slab above */
    const uint16_t* carry_in(Slab& s, int dir)
    {
        if (!s.needs_carry(dir)) return nullptr;
        if (local) {
            for (size_t i = 0; i < d.size(); ++i)
                if (d[i] == &s) return dir == 0 ? d[i - 1]->carry_buf(0) : d[i + 1]->carry_buf(1);
            return nullptr;
        }
        const int side = dir == 0 ? 0 : 1;
        if (s.has_comm()) fail(s.native_recv(side, s.carry_recv_buf(dir), s.carry_bytes()));
        else {
            hr[0].resize((size_t)s.carry_bytes());
            fail(cb->recv(cb->ctx, side, hr[0].data(), s.carry_bytes()));
            s.from_host(s.carry_recv_buf(dir), hr[0].data(), s.carry_bytes());
        }
        return s.carry_recv_buf(dir);
    }
    void carry_out(Slab& s, int dir)
    {
        if (local) return; /* (the next slab reads the buffer where it lies) */
        const int side = dir == 0 ? 1 : 0;
        if (s.has_comm()) fail(s.native_send(side, s.carry_buf(dir), s.carry_bytes()));
        else {
            hs[0].resize((size_t)s.carry_bytes());
            s.to_host(hs[0].data(), s.carry_buf(dir), s.carry_bytes());
            fail(cb->send(cb->ctx, side, hs[0].data(), s.carry_bytes()));
        }
    }
};

#endif /* MGC_DRIVER_INL */
