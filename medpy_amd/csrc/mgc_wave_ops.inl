/*
 * mgc_wave_ops.inl -- ONE WAVE PER TILE forms of the two hot tile operations of the 6-neighbourhood
 * solver (region discharge, global-relabel pass).  Same state in HBM, same schedule, same result
 * definition as mgc_tile_ops.inl; what changes is how a tile is mapped onto the machine:
 *
 *   a wave64 owns a whole 8x8x8 tile: lane = (y, x), and every lane keeps its z-COLUMN of eight
 *   voxels in registers (excess, residual sink link, the six residual n-links, the label: 136 VGPRs).
 *
 *   - +-z pushes and label reads never leave the lane; visiting the column in push order moves flow
 *     through all eight layers in ONE step (the 512-thread form needed one sweep, two LDS hand-offs
 *     and three workgroup barriers per layer);
 *   - +-x / +-y hand-offs are lane shifts inside the wave;
 *   - there is no workgroup barrier anywhere: every vote is a ballot, so a slot (z-layer) without
 *     excess costs one compare + one scalar branch, and a direction in which nobody pushed costs
 *     nothing after its vote;
 *   - LDS only holds the 10x10x10 label block (own labels + halo, read by the in-plane neighbours)
 *     and the staged inbox: 7 KiB per tile, so the number of tiles in flight per CU is set by
 *     registers, not by LDS.
 *
 * Why: the 512-thread kernel was bound by instruction issue (about 500 instructions per wave and
 * sweep, 40 % of them scalar control flow and address arithmetic repeated by all eight waves of a
 * tile, plus 24 barriers per discharge; profiles/README.md round 2).  Here the per-tile scalar work is
 * paid once instead of eight times.
 *
 * Replaces (reference): Graph::maxflow, lib/maxflow/src/maxflow.cpp:472-604 -- see mgc_tile_ops.inl
 * for the algorithm (region-discharge push-relabel, Delong & Boykov 2008) and why its labels are the
 * ones what_segment() reports (graph.h:561-571).
 *
 * Written against a "wave executor" W so that the host simulator (tests/hostsim) runs this very
 * source in the CPU test tier:
 *   W::Reg<T, N>          N values per lane; r(l, k)
 *   w.lanes(f)            f(l) for the 64 lanes; lanes only touch their own registers and LDS cells
 *                         nobody else touches in the same step (the host runs them one after another)
 *   w.any(f)              ballot: true if f(l) holds for some lane (wave-uniform)
 *   w.shift(dst, src, k)  dst(l,0) = src(l + k, 0) inside the wave, 0.0 beyond its ends
 *   w.S                   MgcWaveShared& (LDS)
 *   w.atomic_*            device-scope atomics on global words
 *   w.st_stream(p, l, v)  w.st as a streaming (non-temporal) store on the GPU: the write-back of a tile's own state
 *   w.ld(p, l) / w.st(p, l, v)   p[l] for a wave-uniform pointer p: SGPR base + 32-bit lane offset on the GPU, so the
 *                         64 + 70 state accesses of a discharge need ONE address register instead of a 64-bit pair each
 *   w.fresh()             the lane id becomes opaque to the optimiser again (GPU): addresses derived from it before this
 *                         point are recomputed afterwards instead of being kept alive (or spilled) across the sweeps
 *   w.shift_x(dst, src, k) the same for k = +-1 when src is 0.0 on every lane whose source would lie in another row of eight
 *                         (the x-row of a tile layer): two DPP row shifts on the GPU instead of two LDS-crossbar permutes
 *   W::kPrefetch          >= 0: the executor runs ahead of the tile loop -- w.hint_begin() / w.hint_end(L) yield the tile this
 *                         wave will discharge NEXT (or -1) while the current one is being swept, so that no ticket or
 *                         list entry is waited for behind the stores of a visit; > 0: w.prefetch(p, bytes) also starts
 *                         moving [p, p + bytes) of that tile towards the caches without a register or a wait
 *                         (1: excess / labels / masks; 2: the residual planes too).  -1 (host): none of this exists
 *   w.use_here(v)         v; on the GPU the first use of v -- and so the wait for it, if it is still on its way back from
 *                         memory -- cannot be scheduled above this point
 *   w.mark(id)            work-profile hook: counts sections in the simulator; on the GPU nothing, or (development build
 *                         -DMGCW_PROFILE) the cycles since the previous mark, accumulated per section
 */
#ifndef MGC_WAVE_OPS_INL
#define MGC_WAVE_OPS_INL

#include <math.h>
#include <utility>

#include "mgc_tile_ops.inl"

#define MGCW_LANES 64
#define MGCW_REPEAT_MAX 8 /* in-plane push steps of one (slot, direction) per sweep in the REP = MGCW_REPEAT_MAX instance of the discharge (see its sweeps) */
/* every lambda of this file must be inlined into the kernel: a call would force the register arrays it captures into memory */
#define MGCW_INL __attribute__((always_inline))
#define MGCW_BFS 1            /* discharge flag: exact in-tile labels (from scratch) before the sweeps */
#define MGCW_BFS_SINK 2       /* ... for the tiles that hold a sink link: their labels are decided inside the tile, and an exact
                                 labelling per visit saves visits (tie-heavy volume, markers everywhere, 512^3 on MI355X: 1077 -> 710 ms);
                                 elsewhere labels come from far away and the stored ones are as good (weak contrast: 69 -> 80 ms with
                                 MGCW_BFS on every tile) */
#define MGCW_INFLOW_DIRTY 8    /* ... (with MGCW_SAT_DIRTY) a radial cycle that is NOT the first of its solve: flow that comes in marks the tile DIRTY too (see the tail) */
#define MGCW_SAT_DIRTY 4      /* ... the visit runs on RADIAL labels (mgc_dt_ops.inl): any saturated arc marks the tile DIRTY -- whether a voxel
                                 keeps "a residual arc one label down" says nothing about its distance when the labels are not distances */


struct alignas(16) MgcWaveShared {
    int32_t hs[1000];          /* 10x10x10 distance labels: the tile plus a one-voxel halo */
    double  inbox[6][MGC_TF];  /* flow the six neighbours left for this tile, staged by the loading lanes; then the outbox */
    double  snk[MGC_TV];       /* residual sink links of the tile being discharged (tiles that hold any) */
};

/* compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>) -- the slot index of a register array must be a
 * constant, otherwise the array lives in scratch memory */
template <class F, int... I>
MGC_HD void mgcw_static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
MGC_HD void mgcw_static_for(F&& f)
{
    mgcw_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

/* hs[] cell of slot k (z-layer) of lane l = (y, x) */
MGC_HD int mgcw_hs(int l, int k) { return (k + 1) * 100 + ((l >> 3) + 1) * 10 + (l & 7) + 1; }

/* one trip to HBM for what the six neighbours contribute to a relabel visit: lane l fetches, per face, the label of the
 * voxel its face cell touches.  Results go to LDS.  (The discharge fetches labels and inbox with mgcw_halo_issue / commit.) */
template <class W>
MGC_HD void mgcw_load_halo(W& w, const MgcLattice& L, int tile, int l)
{
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
        const int mine = mgc_face_voxel(f, l);
        int32_t hv = MGC_HINF;
        if (nt >= 0) hv = w.ld(L.height + (int64_t)nt * MGC_TV, mgc_face_voxel(f ^ 1, l));
        w.S.hs[mgc_hs_index(mine >> 6, (mine >> 3) & 7, mine & 7) + mgc_hs_step(f)] = hv;
    }
}

/* The same in two steps, for the discharge: ISSUE all twelve loads (per face the label and the outbox slot), COMMIT them to
 * LDS later.  In one step the "empty the slot if it held something" store of face f sits between the loads of face f and
 * those of face f + 1, and the compiler may not move a load across a store it cannot tell apart from it: the six faces
 * became six dependent trips to HBM, each behind the ~76 loads of the tile's own state (that chain WAS the 32 k cycles of the
 * "load + absorb" section of the round-2 profile).  Issued first, the twelve loads come back first (a wave's loads return in
 * issue order) and the commit overlaps with the own state still in flight. */
template <class W, class RegI, class RegD, class RegF>
MGC_HD void mgcw_halo_issue(W& w, const MgcLattice& L, int tile, int l, RegI& hv, RegD& din, RegF& ofl)
{
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    mgcw_static_for<6>([&](auto FF) MGCW_INL {
        constexpr int F = decltype(FF)::value;
        const int nt = mgc_tile_nbr(L, tz, ty, tx, F);
        hv(l, F) = MGC_HINF;
        din(l, F) = 0.0;
        if (nt >= 0) {
            hv(l, F) = w.ld(L.height + (int64_t)nt * MGC_TV, mgc_face_voxel(F ^ 1, l));
            din(l, F) = w.ld(L.obox + ((int64_t)nt * 6 + (F ^ 1)) * MGC_TF, l);
        }
    });
    ofl(l, 0) = 0; /* lane l < 6: the outbox flags of the neighbour across face l */
    if (l < 6) {
        const int nt = mgc_tile_nbr(L, tz, ty, tx, l);
        if (nt >= 0) ofl(l, 0) = (int)L.oflags[nt];
    }
}

template <class W, class RegI, class RegD>
MGC_HD void mgcw_halo_commit(W& w, const MgcLattice& L, int tile, int l, RegI& hv, RegD& din)
{
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    mgcw_static_for<6>([&](auto FF) MGCW_INL {
        constexpr int F = decltype(FF)::value;
        const int nt = mgc_tile_nbr(L, tz, ty, tx, F);
        const int mine = mgc_face_voxel(F, l);
        w.S.hs[mgc_hs_index(mine >> 6, (mine >> 3) & 7, mine & 7) + mgc_hs_step(F)] = hv(l, F);
        w.S.inbox[F][l] = din(l, F);
        if (nt >= 0 && din(l, F) != 0.0) w.st(L.obox + ((int64_t)nt * 6 + (F ^ 1)) * MGC_TF, l, 0.0); /* the slot is emptied */
    });
}

/* label of the neighbour of (lane l, slot K) in direction d: in-plane neighbours and the tile halo from LDS, the
 * lane's own column from registers */
template <int K, int D, class W, class RegI>
MGC_HD int mgcw_nbr_label(W& w, RegI& h, int l)
{
    if constexpr (D == 4) {
        if constexpr (K > 0) return h(l, K - 1);
        else return w.S.hs[mgcw_hs(l, 0) - 100];
    } else if constexpr (D == 5) {
        if constexpr (K < 7) return h(l, K + 1);
        else return w.S.hs[mgcw_hs(l, 7) + 100];
    } else {
        return w.S.hs[mgcw_hs(l, K) + mgc_hs_step(D)];
    }
}

/* ---------------------------------------------------------------------------------------
 * Label relaxation of a whole tile to its fixpoint: h(u) = min(h(u), 1 if u has a sink arc, 1 + h(v) over residual
 * arcs u->v), halo frozen.  arc(l, K, D) says whether the arc of (lane, slot) in direction D (6 = sink) is residual.
 * One round = the lane's z-column swept down and up in registers (information along z crosses all eight layers at
 * once), then one in-plane step for all eight slots against the labels the wave published in LDS after the round
 * before; ONE vote per round ("did any lane lower a label?").  Labels only decrease, so the fixpoint is the same
 * whatever the order.  (Round 2 visited slot by slot with a vote each: 16 votes per round and a wave-uniform branch per
 * slot -- the scalar half of the kernel.)
 * w.S.hs must hold the tile's current labels and the halo when this is called.
 * ------------------------------------------------------------------------------------- */
template <class W, class RegI, class ArcFn>
MGC_HD void mgcw_relax(W& w, RegI& h, ArcFn arc)
{
    typename W::template Reg<int, 1> moved;
    typename W::template Reg<int, 2> hz; /* the halo below slot 0 / above slot 7 (frozen) */
    w.lanes([&](int l) MGCW_INL {
        hz(l, 0) = w.S.hs[mgcw_hs(l, 0) - 100];
        hz(l, 1) = w.S.hs[mgcw_hs(l, 7) + 100];
        mgcw_static_for<8>([&](auto KK) MGCW_INL { /* a sink arc gives 1, once and for all */
            constexpr int K = decltype(KK)::value;
            if (arc(l, KK, std::integral_constant<int, 6>{}) && h(l, K) > 1) {
                h(l, K) = 1;
                w.S.hs[mgcw_hs(l, K)] = 1;
            }
        });
    });
    for (;;) {
        w.lanes([&](int l) MGCW_INL {
            int ch = 0;
            auto lower = [&](auto KK, bool has_arc, int hv) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                const int c = has_arc ? hv + 1 : MGC_HINF; /* hv == MGC_HINF gives a value above every label */
                if (c < h(l, K)) { h(l, K) = c; ch = 1; }
            };
            mgcw_static_for<8>([&](auto KK) MGCW_INL { /* from below, slot 0 upwards */
                constexpr int K = decltype(KK)::value;
                int hb = hz(l, 0);
                if constexpr (K > 0) hb = h(l, K - 1);
                lower(KK, arc(l, KK, std::integral_constant<int, 4>{}), hb);
            });
            mgcw_static_for<8>([&](auto KK) MGCW_INL { /* from above, slot 7 downwards */
                constexpr int K = 7 - decltype(KK)::value;
                constexpr std::integral_constant<int, K> KC{};
                int ha = hz(l, 1);
                if constexpr (K < 7) ha = h(l, K + 1);
                lower(KC, arc(l, KC, std::integral_constant<int, 5>{}), ha);
            });
            mgcw_static_for<8>([&](auto KK) MGCW_INL { /* in-plane: the neighbours as published after the round before */
                constexpr int K = decltype(KK)::value;
                mgcw_static_for<4>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    lower(KK, arc(l, KK, DD), w.S.hs[mgcw_hs(l, K) + mgc_hs_step(D)]);
                });
            });
            moved(l, 0) = ch;
        });
        if (!w.any([&](int l) MGCW_INL -> bool { return moved(l, 0) != 0; })) break;
        w.lanes([&](int l) MGCW_INL { /* publish (a lane that lowered nothing rewrites what is there) */
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                w.S.hs[mgcw_hs(l, K)] = h(l, K);
            });
        });
    }
}

/* start moving the state of the tile this wave discharges next: a wave spends half of a visit waiting for its own loads and
 * stores (profiles/README.md), and all resident waves of a launch do so at the same moments; the next tile's lines are
 * requested while this one is being swept, so that its load phase finds them in the L2 / Infinity Cache */
template <class W>
MGC_HD void mgcw_prefetch_tile(W& w, const MgcLattice& L, int tile)
{
    w.prefetch(L.excess + (int64_t)tile * MGC_TV, MGC_TV * (int)sizeof(double));
    w.prefetch(L.height + (int64_t)tile * MGC_TV, MGC_TV * (int)sizeof(int32_t));
    w.prefetch(L.rmask + (int64_t)tile * MGC_TV, MGC_TV);
    if (W::kPrefetch >= 2) w.prefetch(L.rcap + (int64_t)tile * 6 * MGC_TV, 6 * MGC_TV * (int)sizeof(double));
}

/* ---------------------------------------------------------------------------------------
 * Region discharge of one tile by one wave (colour phase `phase`; the six face neighbours are idle).
 *   load -> absorb inbox -> labels -> sweeps { per slot: sink, -x, +x, -y, +y ; -z down the column ; +z up the
 *   column ; local relabel } -> store (state, masks, labels, outbox, wake-ups).
 * Every hand-off has one sender per receiver and a fixed order of f64 operations: bit-reproducible, no atomics on
 * flow data.  Saturating pushes leave an exact 0.0.
 *
 * SINK = the tile holds residual sink links (status bit MGC_ST_SINK; 2 % of the tiles of a volume whose background
 * markers are its faces).  Only then does the sink plane exist for the kernel at all; it then lives in LDS
 * (w.S.snk), not in registers: the register budget -- excess, six residual planes, labels = 120 VGPRs -- is what
 * sets the number of tiles in flight per SIMD.
 *
 * Control flow is wave-uniform throughout: a lane never branches on its own data, it computes "can I push" as a
 * predicate, the wave votes, and the update runs branch-free (selects) only if somebody can -- no EXEC-mask
 * juggling, and a (slot, direction) pair in which nobody pushes costs three compares and a scalar branch.
 * ------------------------------------------------------------------------------------- */
template <bool SINK, int REP = 1, class W>
MGC_HD void mgcw_discharge_impl(W& w, const MgcLattice& L, int tile, uint32_t phase, int max_sweeps, int flags)
{
    typename W::template Reg<double, 8> e;
    typename W::template Reg<double, 8> r[6];
    typename W::template Reg<int, 8> h;
    typename W::template Reg<double, 1> dl, din;

    double* const t_excess = L.excess + (int64_t)tile * MGC_TV;
    double* const t_sink = L.sink + (int64_t)tile * MGC_TV;
    double* const t_rcap = L.rcap + (int64_t)tile * 6 * MGC_TV;
    double* const t_obox = L.obox + (int64_t)tile * 6 * MGC_TF;
    uint8_t* const t_rmask = L.rmask + (int64_t)tile * MGC_TV;
    int32_t* const t_height = L.height + (int64_t)tile * MGC_TV;
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);

    /* ---- one trip to HBM: label halo + inbox (issued first, see mgcw_halo_issue), own state ---- */
    typename W::template Reg<int, 6> hv;
    typename W::template Reg<double, 6> dnb;
    typename W::template Reg<int, 1> st0; /* the tile's status word (nobody else writes it during this launch) */
    typename W::template Reg<int, 1> ofl;
    w.lanes([&](int l) MGCW_INL {
        mgcw_halo_issue(w, L, tile, l, hv, dnb, ofl);
        st0(l, 0) = (int)L.status[tile];
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            e(l, K) = w.ld(t_excess, K * 64 + l);
            if constexpr (SINK) w.S.snk[K * 64 + l] = w.ld(t_sink, K * 64 + l);
            mgcw_static_for<6>([&](auto DD) MGCW_INL {
                constexpr int D = decltype(DD)::value;
                r[D](l, K) = w.ld(t_rcap + D * MGC_TV, K * 64 + l);
            });
            h(l, K) = w.ld(t_height, K * 64 + l); /* (overwritten when the exact labelling runs) */
        });
        /* the ticket for the tile AFTER this one goes out behind the loads: a wave's memory operations retire in issue order, and
         * the ticket word is the one address every wave of the launch hits (ahead of the loads it would hold them all up) */
        if constexpr (W::kPrefetch >= 0) w.ticket_issue(L);
        mgcw_halo_commit(w, L, tile, l, hv, dnb);
        if (l < 6) { /* retire the outbox flags of the slots just emptied */
            const int nt = mgc_tile_nbr(L, tz, ty, tx, l);
            if (nt >= 0 && (((uint32_t)ofl(l, 0) >> (l ^ 1)) & 1u)) w.atomic_and(&L.oflags[nt], ~(1u << (l ^ 1)));
        }
    });
    w.mark(4); /* loads issued, halo + inbox back and staged */
    /* ---- absorb the staged inbox: e += delta, reverse residual += delta, fixed face order ---- */
    w.lanes([&](int l) MGCW_INL {
        const int y = l >> 3, x = l & 7;
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            /* selects on the VALUES, never on which register is updated: a branch per face makes the optimiser merge
             * the updates into one store through a pointer phi, which pins the residual arrays to scratch memory */
            const double d0 = x == 0 ? w.S.inbox[0][K * 8 + y] : 0.0;
            const double d1 = x == 7 ? w.S.inbox[1][K * 8 + y] : 0.0;
            const double d2 = y == 0 ? w.S.inbox[2][K * 8 + x] : 0.0;
            const double d3 = y == 7 ? w.S.inbox[3][K * 8 + x] : 0.0;
            e(l, K) += d0; r[0](l, K) += d0;
            e(l, K) += d1; r[1](l, K) += d1;
            e(l, K) += d2; r[2](l, K) += d2;
            e(l, K) += d3; r[3](l, K) += d3;
            if constexpr (K == 0) { const double d = w.S.inbox[4][l]; e(l, K) += d; r[4](l, K) += d; }
            if constexpr (K == 7) { const double d = w.S.inbox[5][l]; e(l, K) += d; r[5](l, K) += d; }
        });
    });
    w.mark(5); /* own state back, inbox absorbed */
    /* residual planes this discharge changes (bit D): a plane nobody pushed along, received along or absorbed into goes
     * back to HBM as it came -- so it does not go back at all (a discharge typically moves flow along one or two axes) */
    uint32_t dirty = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f)
        if (w.any([&](int l) MGCW_INL -> bool { return w.S.inbox[f][l] != 0.0; })) dirty |= 1u << f;
    const bool inflow = dirty != 0; /* a neighbour left flow for this tile */
    bool relabelled = (flags & MGCW_BFS) != 0; /* some label of the tile changed */

    /* ---- labels: exact in-tile distances given the frozen halo, or the stored (valid lower-bound) labels ---- */
    if (flags & MGCW_BFS) {
        w.lanes([&](int l) MGCW_INL {
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                h(l, K) = MGC_HINF;
                w.S.hs[mgcw_hs(l, K)] = MGC_HINF;
            });
        });
        mgcw_relax(w, h, [&](int l, auto KK, auto DD) MGCW_INL -> bool {
            constexpr int K = decltype(KK)::value;
            constexpr int D = decltype(DD)::value;
            if constexpr (D == 6) {
                if constexpr (SINK) return w.S.snk[K * 64 + l] > 0.0;
                else return false;
            } else {
                return r[D](l, K) > 0.0;
            }
        });
    } else {
        w.lanes([&](int l) MGCW_INL {
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                w.S.hs[mgcw_hs(l, K)] = h(l, K);
            });
        });
    }

    /* Sweeps.  Data-dependent branches: two batches of slot votes per sweep (which z-layers hold excess that can move) and
     * ONE vote per (active slot, direction) -- "can anybody push?".  Behind a vote everything is branch-free.  Measured on
     * MI355X (512^3, ms of discharge kernels per step): a vote before the push AND around saturation / outflow / hand-off
     * 24.0; no vote at all inside a sweep (every active slot always pushes and shifts in all directions, twice the VALU
     * work, fewer stalls) 28.7 -- but 20 % faster on small volumes, where a launch is one tile deep. */
    typename W::template Reg<double, 8> obx, oby; /* flow pushed out across the x / y faces (meaningful on the face lanes) */
    typename W::template Reg<double, 2> obz;      /* ... across the -z / +z faces */
    typename W::template Reg<int, 16> hn;         /* in-plane neighbour labels of the four slots being swept */
    typename W::template Reg<int, 2> hz;          /* halo labels below slot 0 / above slot 7 (frozen) */
    typename W::template Reg<int, 1> sat;         /* bit K: this lane's voxel of slot K saturated an arc (or its sink link); bit 8: a label of the lane rose */
    w.lanes([&](int l) MGCW_INL {
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            obx(l, K) = 0.0;
            oby(l, K) = 0.0;
        });
        obz(l, 0) = obz(l, 1) = 0.0;
        sat(l, 0) = 0;
        hz(l, 0) = w.S.hs[mgcw_hs(l, 0) - 100];
        hz(l, 1) = w.S.hs[mgcw_hs(l, 7) + 100];
    });
    /* push of (lane, slot K) along D towards a neighbour labelled hnb, branch-free: min(excess, residual) where the arc
     * is admissible, 0.0 elsewhere.  (A label of MGC_HINF never matches: finite labels stay far below MGC_HINF - 1.) */
    auto push = [&](int l, auto KK, auto DD, int hnb) MGCW_INL -> double {
        constexpr int K = decltype(KK)::value;
        constexpr int D = decltype(DD)::value;
        const double rd = r[D](l, K);
        const bool can = e(l, K) > 0.0 && rd > 0.0 && hnb == h(l, K) - 1;
        const double m = w.fmin_pos(e(l, K), rd);
        const double delta = can ? m : 0.0;
        e(l, K) -= delta;
        r[D](l, K) = rd - delta; /* saturating push: rd - rd == 0.0 exactly */
        sat(l, 0) |= (can && delta == rd) ? (1 << K) : 0;
        return delta;
    };
    auto slot_mask = [&]() MGCW_INL -> uint32_t { /* bit K: some voxel of slot K holds excess that can reach the sink */
        uint32_t m = 0;
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            if (w.any([&](int l) MGCW_INL -> bool { return e(l, K) > 0.0 && h(l, K) < MGC_HINF; })) m |= 1u << K;
        });
        return m;
    };
    /* may (lane, slot K) push along direction D towards a neighbour labelled hnb? */
    auto can_push = [&](int l, auto KK, auto DD, int hnb) MGCW_INL -> bool {
        constexpr int K = decltype(KK)::value;
        constexpr int D = decltype(DD)::value;
        return e(l, K) > 0.0 && r[D](l, K) > 0.0 && hnb == h(l, K) - 1;
    };

    w.mark(0); /* load + absorb + label set-up */
    /* the ticket is looked at after the first sweep, the list entry it names after the second (or right after the loop): by
     * then both have long arrived, however many waves queued up on the ticket word */
    int hint_stage = 0;
    auto hint_step = [&]() MGCW_INL {
        if constexpr (W::kPrefetch >= 0) {
            if (hint_stage == 0) w.hint_begin();
            else if (hint_stage == 1) {
                const int coming = w.hint_end(L); /* the tile this wave discharges next */
                if (W::kPrefetch > 0 && coming >= 0) mgcw_prefetch_tile(w, L, coming);
            }
            hint_stage++;
        }
    };
    uint32_t am = slot_mask();
    for (int sw = 0; sw < max_sweeps && am; ++sw) {
        /* ---- per active slot: sink, then the four in-plane directions as lane shifts.  ONE vote per (slot, direction):
         * "can anybody push?" -- if so the push, the hand-off and the receive run branch-free ---- */
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            constexpr int J = K & 3;
            if (!(am & (1u << K))) return;
            w.lanes([&](int l) MGCW_INL { /* the four in-plane neighbour labels: constant during the push steps */
                mgcw_static_for<4>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    hn(l, 4 * J + D) = w.S.hs[mgcw_hs(l, K) + mgc_hs_step(D)];
                });
            });
            if constexpr (SINK) {
                w.lanes([&](int l) MGCW_INL { /* push to the sink first: always admissible (label 1 -> 0) */
                    const double sk = w.S.snk[K * 64 + l];
                    const bool can = e(l, K) > 0.0 && sk > 0.0;
                    const double m = w.fmin_pos(e(l, K), sk);
                    const double delta = can ? m : 0.0;
                    e(l, K) -= delta;
                    w.S.snk[K * 64 + l] = sk - delta;
                    sat(l, 0) |= (can && delta == sk) ? (1 << K) : 0;
                });
            }
            mgcw_static_for<4>([&](auto DD) MGCW_INL {
                constexpr int D = decltype(DD)::value;
                /* REP > 1: the step of a direction is REPEATED while somebody can still push (at most REP times): what a lane received from
                 * its neighbour moves on to the next lane in the same sweep, so flow crosses the tile along x / y in one sweep as it does
                 * along z (the labels of the in-plane neighbours are constant during the push steps: only the local relabel at the end of
                 * a sweep changes any).  Pays where thin flows travel far along exact labels -- weak contrast, integer-valued (CT-like)
                 * volumes: 512^3 ct 74.4 -> 53.1 ms, a third fewer colour phases -- and costs a vote more per pushing direction where
                 * every voxel pushes anyway (the flood on radial labels, small volumes: 256^3 4.42 -> 4.92 ms): the schedule picks the
                 * instance per launch (profiles/r6_ab_repeat.jsonl) */
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int rep = 0; rep < REP; ++rep) {
                if (!w.any([&](int l) MGCW_INL -> bool { return can_push(l, KK, DD, hn(l, 4 * J + D)); })) break;
                dirty |= 3u << (D & ~1);
                w.lanes([&](int l) MGCW_INL {
                    const int y = l >> 3, x = l & 7;
                    const double delta = push(l, KK, DD, hn(l, 4 * J + D));
                    const bool inside = D == 0 ? x > 0 : (D == 1 ? x < 7 : (D == 2 ? y > 0 : y < 7));
                    const double stay = inside ? delta : 0.0;
                    dl(l, 0) = stay;
                    if constexpr (D < 2) obx(l, K) += delta - stay; /* what leaves the tile across face D: delta or 0.0, exactly */
                    else oby(l, K) += delta - stay;
                });
                /* -x: from the lane at x + 1, ...; dl is 0.0 on the lanes whose push left the tile, which are exactly the lanes
                 * at the end of an x-row: the x shifts never carry anything from one row of eight into the next */
                if constexpr (D < 2) w.shift_x(din, dl, D == 0 ? 1 : -1);
                else w.shift(din, dl, D == 2 ? 8 : -8);
                w.lanes([&](int l) MGCW_INL { /* what the neighbour pushed in direction D arrives: reverse residual grows */
                    e(l, K) += din(l, 0);
                    r[D ^ 1](l, K) += din(l, 0);
                });
                }
            });
        });
        /* ---- -z down the column, then +z up the column: flow crosses all eight layers in one pass (the pushes of one
         * lane never leave its registers) ---- */
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = 7 - decltype(KK)::value;
            constexpr std::integral_constant<int, K> KC{};
            constexpr std::integral_constant<int, 4> DC{};
            auto below = [&](int l) MGCW_INL -> int {
                int hb = hz(l, 0);
                if constexpr (K > 0) hb = h(l, K - 1);
                return hb;
            };
            if (!w.any([&](int l) MGCW_INL -> bool { return can_push(l, KC, DC, below(l)); })) return;
            dirty |= 3u << 4;
            w.lanes([&](int l) MGCW_INL {
                const double delta = push(l, KC, DC, below(l));
                if constexpr (K > 0) { e(l, K - 1) += delta; r[5](l, K - 1) += delta; }
                else obz(l, 0) += delta;
            });
        });
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            constexpr std::integral_constant<int, 5> DC{};
            auto above = [&](int l) MGCW_INL -> int {
                int ha = hz(l, 1);
                if constexpr (K < 7) ha = h(l, K + 1);
                return ha;
            };
            if (!w.any([&](int l) MGCW_INL -> bool { return can_push(l, KK, DC, above(l)); })) return;
            dirty |= 3u << 4;
            w.lanes([&](int l) MGCW_INL {
                const double delta = push(l, KK, DC, above(l));
                if constexpr (K < 7) { e(l, K + 1) += delta; r[4](l, K + 1) += delta; }
                else obz(l, 1) += delta;
            });
        });
        /* ---- local relabel (classic push-relabel step): a voxel that still holds excess rises to 1 + the lowest label
         * behind a residual arc.  Labels stay valid lower bounds of the distance (no push runs in this step; an in-plane
         * neighbour's label used here is the one it had before the step).  A voxel with excess either rises here or still
         * has an admissible arc, so "some slot is still active afterwards" == "another sweep will move something". ---- */
        am = slot_mask();
        if (!am) break;
        relabelled = true; /* (conservative: some voxel with excess is looked at; nearly always one of them rises) */
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            constexpr int J = K & 3;
            if (!(am & (1u << K))) return;
            w.lanes([&](int l) MGCW_INL {
                mgcw_static_for<4>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    hn(l, 4 * J + D) = w.S.hs[mgcw_hs(l, K) + mgc_hs_step(D)];
                });
            });
            w.lanes([&](int l) MGCW_INL {
                int c = MGC_HINF;
                if constexpr (SINK) c = w.S.snk[K * 64 + l] > 0.0 ? 1 : MGC_HINF;
                mgcw_static_for<6>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    int hv;
                    if constexpr (D < 4) hv = hn(l, 4 * J + D);
                    else if constexpr (D == 4) { if constexpr (K > 0) hv = h(l, K - 1); else hv = hz(l, 0); }
                    else { if constexpr (K < 7) hv = h(l, K + 1); else hv = hz(l, 1); }
                    const int cd = r[D](l, K) > 0.0 ? hv + 1 : MGC_HINF; /* hv == MGC_HINF gives a value above every candidate */
                    c = cd < c ? cd : c;
                });
                const bool rises = e(l, K) > 0.0 && h(l, K) < c;
                sat(l, 0) |= rises ? 256 : 0;
                h(l, K) = rises ? c : h(l, K);
                w.S.hs[mgcw_hs(l, K)] = h(l, K);
            });
        });
        am = slot_mask();
        hint_step();
        w.mark(2); /* one push sweep */
    }
    hint_step(); /* (a visit of fewer than two sweeps) */
    hint_step();
    const bool active = am != 0; /* sweep budget exhausted with work left: run again in the next phase of this colour */
    w.mark(1); /* (whatever followed the last counted sweep: the vote that ended the loop) */

    /* ---- tail.  Order matters for time, not for the result: a wave's memory operations retire in issue order, and the wake-ups
     * are two dependent returning atomics per woken tile (claim its stamp, then draw a list position).  They frame the
     * tail: the claims go out first, the votes and the staging of the outbox run while they are in flight, the positions
     * are drawn next, then the ~70 stores of the write-back are issued -- nobody waits for those -- and the list entries
     * (which need the positions) go out last.  (Round 2 issued the claims after the stores: every visit ended by waiting
     * for its whole write-back to retire, twice.) ---- */
    uint32_t face = 0; /* bit f: flow left the tile across face f */
    {
        typename W::template Reg<int, 1> nz; /* bit 0 / 1 / 2: this lane pushed something across an x / y / z face */
        w.lanes([&](int l) MGCW_INL {
            int m = 0;
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                m |= (obx(l, K) != 0.0 ? 1 : 0) | (oby(l, K) != 0.0 ? 2 : 0);
            });
            nz(l, 0) = m | (obz(l, 0) != 0.0 ? 4 : 0) | (obz(l, 1) != 0.0 ? 8 : 0);
        });
        if (w.any([&](int l) MGCW_INL -> bool { return (nz(l, 0) & 1) && (l & 7) == 0; })) face |= 1u;
        if (w.any([&](int l) MGCW_INL -> bool { return (nz(l, 0) & 1) && (l & 7) == 7; })) face |= 2u;
        if (w.any([&](int l) MGCW_INL -> bool { return (nz(l, 0) & 2) && (l >> 3) == 0; })) face |= 4u;
        if (w.any([&](int l) MGCW_INL -> bool { return (nz(l, 0) & 2) && (l >> 3) == 7; })) face |= 8u;
        if (w.any([&](int l) MGCW_INL -> bool { return (nz(l, 0) & 4) != 0; })) face |= 16u;
        if (w.any([&](int l) MGCW_INL -> bool { return (nz(l, 0) & 8) != 0; })) face |= 32u;
    }
    /* lane l < 6 wakes the neighbour across face l, lane 6 this tile itself (budget exhausted with work left) */
    typename W::template Reg<int, 4> wk; /* tile to wake (-1: none), its list, claimed?, position */
    int32_t* const list_nbr = L.list[(phase + 1) & 3u];
    int32_t* const list_self = L.list[(phase + 2) & 3u];
    w.fresh();
    w.lanes([&](int l) MGCW_INL {
        int target = -1, lst = 0;
        uint32_t ep = 0;
        if (l < 6 && ((face >> l) & 1u)) { target = mgc_tile_nbr(L, tz, ty, tx, l); ep = phase + 1; }
        if (l == 6 && active) { target = tile; ep = phase + 2; }
        lst = (int)(ep & 3u);
        if (target >= 0 && !mgc_owned(L, target)) target = -1; /* a ghost tile is discharged by the slab that owns it */
        wk(l, 0) = target;
        wk(l, 1) = lst;
        wk(l, 2) = (int)ep;
        if (l < 6 && ((face >> l) & 1u)) w.atomic_or(&L.oflags[tile], 1u << l);
        /* claim: whoever finds another stamp there is the first to queue the tile for that phase.  (What comes back is only
         * looked at further down: the wait for it sits behind the votes and the staging below.) */
        if (target >= 0) wk(l, 2) = (int)w.atomic_exch(&L.stamp[target], ep);
    });
    bool has_sink = false;
    /* excess under a finite label (excess under an INF label is dead for good): what the slot votes that ended the sweeps said */
    const bool has_exc = active;
    if constexpr (SINK) {
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            has_sink = has_sink || w.any([&](int l) MGCW_INL -> bool { return w.S.snk[K * 64 + l] > 0.0; });
        });
    }
    /* DIRTY (the tile's labels may no longer be exact distances: the next global relabel recomputes it and whoever depends on it)
     * iff a label rose or was recomputed, or a voxel that saturated an arc has no residual arc one label down left.  A voxel that
     * keeps one of its supports keeps its distance, and the tiles a small flow merely passes through stay clean. */
    bool saturated = (flags & (MGCW_BFS | MGCW_SAT_DIRTY)) ? w.any([&](int l) MGCW_INL -> bool { return sat(l, 0) != 0; })
                                        : w.any([&](int l) MGCW_INL -> bool { return (sat(l, 0) & 256) != 0; });
    /* On RADIAL labels flow that came in marks the tile DIRTY as well.  The push that sent it was admissible under the radial labels only; the
     * residual arc it opened back towards the sender can undercut the EXACT label kept aside for the receiving voxel (exact(v) > exact(u) + 1 for
     * the new arc v -> u), and a tile that merely passed the flow on would keep that label through the incremental relabel that ends the flood.
     * In the FIRST radial cycle of a solve it cannot happen -- it starts from the distance transform's labels, all arcs residual, neighbours at
     * most one apart -- and marking there would only add 40 k tiles to the relabel behind the flood of the headline volume (+0.4 ms); floods cut
     * into several cycles (radial_rounds0 > 0) start their later ones from graphs with saturated arcs: MGCW_INFLOW_DIRTY. */
    if ((flags & MGCW_INFLOW_DIRTY) && inflow) saturated = true;
    if (!saturated && w.any([&](int l) MGCW_INL -> bool { return sat(l, 0) != 0; })) {
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            if (saturated || !w.any([&](int l) MGCW_INL -> bool { return ((sat(l, 0) >> K) & 1) != 0; })) return;
            saturated = w.any([&](int l) MGCW_INL -> bool {
                const int hk = h(l, K);
                bool kept = false;
                if constexpr (SINK) kept = w.S.snk[K * 64 + l] > 0.0; /* (a label of 1 stands on the sink link) */
                mgcw_static_for<6>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    int hv;
                    if constexpr (D < 4) hv = w.S.hs[mgcw_hs(l, K) + mgc_hs_step(D)];
                    else if constexpr (D == 4) { if constexpr (K > 0) hv = h(l, K - 1); else hv = hz(l, 0); }
                    else { if constexpr (K < 7) hv = h(l, K + 1); else hv = hz(l, 1); }
                    /* (a support in a neighbour tile counts only if the tile's support bits watch that neighbour: mgc_support_watched) */
                    const bool in = D == 0 ? (l & 7) > 0 : (D == 1 ? (l & 7) < 7 : (D == 2 ? (l >> 3) > 0 : (D == 3 ? (l >> 3) < 7 : (D == 4 ? K > 0 : K < 7))));
                    kept = kept || (r[D](l, K) > 0.0 && hv == hk - 1 && (in || (((uint32_t)st0(l, 0) >> (MGC_ST_DEP_SHIFT + D)) & 1u)));
                });
                return ((sat(l, 0) >> K) & 1) && hk < MGC_HINF && !kept;
            });
        });
    }
    w.lanes([&](int l) MGCW_INL { /* outbox staged through LDS (face order: the neighbours read 64 consecutive doubles per face) */
        const int y = l >> 3, x = l & 7;
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            if (x == 0) w.S.inbox[0][K * 8 + y] = obx(l, K);
            if (x == 7) w.S.inbox[1][K * 8 + y] = obx(l, K);
            if (y == 0) w.S.inbox[2][K * 8 + x] = oby(l, K);
            if (y == 7) w.S.inbox[3][K * 8 + x] = oby(l, K);
        });
        w.S.inbox[4][l] = obz(l, 0);
        w.S.inbox[5][l] = obz(l, 1);
    });
    w.mark(6); /* face votes, claims issued, tail votes, outbox staged */
    w.lanes([&](int l) MGCW_INL { /* positions in the lists (region of this workgroup, MgcLattice::scount) */
        const uint32_t ep = l == 6 ? phase + 2 : phase + 1;
        wk(l, 2) = (wk(l, 0) >= 0 && (uint32_t)wk(l, 2) != ep) ? 1 : 0;
        wk(l, 3) = 0;
        if (wk(l, 2)) wk(l, 3) = w.atomic_add(mgc_counter(L, wk(l, 1), w.shard(L)), 1);
    });
    /* ---- ONE block of global stores nobody waits for: state, masks, labels, outbox ---- */
    w.lanes([&](int l) MGCW_INL {
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            w.st_stream(t_excess, K * 64 + l, e(l, K));
            int m = 0;
            if constexpr (SINK) {
                const double sk = w.S.snk[K * 64 + l];
                w.st_stream(t_sink, K * 64 + l, sk);
                m = sk > 0.0 ? MGC_MASK_SINK : 0;
            }
            mgcw_static_for<6>([&](auto DD) MGCW_INL {
                constexpr int D = decltype(DD)::value;
                m |= (r[D](l, K) > 0.0) ? (1 << D) : 0;
            });
            w.st_stream(t_rmask, K * 64 + l, (uint8_t)m);
        });
    });
    mgcw_static_for<6>([&](auto DD) MGCW_INL { /* only the residual planes that changed */
        constexpr int D = decltype(DD)::value;
        if (!(dirty & (1u << D))) return;
        w.lanes([&](int l) MGCW_INL {
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                w.st_stream(t_rcap + D * MGC_TV, K * 64 + l, r[D](l, K));
            });
        });
    });
    if (relabelled) {
        w.lanes([&](int l) MGCW_INL {
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                w.st(t_height, K * 64 + l, h(l, K));
            });
        });
    }
    w.mark(7); /* claims back, positions drawn, state / planes / labels stored */
    w.lanes([&](int l) MGCW_INL {
        /* outbox: plain stores -- the neighbour emptied these slots when it last absorbed, and it always runs (or
         * absorb_all does) between two of our discharges */
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const double ob = w.S.inbox[f][l];
            if (ob != 0.0) w.st(t_obox + f * MGC_TF, l, ob);
        }
        /* (the two lists as wave-uniform pointers selected per lane: indexing L.list[] with a per-lane list id would be a LOAD of
         * the pointer, and waiting for it would wait for every store above) */
        if (wk(l, 2)) (l == 6 ? list_self : list_nbr)[(int64_t)w.shard(L) * L.shard_cap + w.use_here(wk(l, 3))] = wk(l, 0);
        /* DIRTY only if a residual arc disappeared: otherwise no distance in the tile (or through it) can have changed */
        if (l == 7) L.status[tile] = ((uint32_t)st0(l, 0) & ~(MGC_ST_SINK | MGC_ST_EXCESS)) | (has_sink ? MGC_ST_SINK : 0u) | (saturated ? MGC_ST_DIRTY : 0u) | (has_exc ? MGC_ST_EXCESS : 0u);
    });
    w.mark(3); /* tail votes + stores */
}

template <int REP = 1, class W>
MGC_HD void mgcw_discharge_tile(W& w, const MgcLattice& L, int tile, uint32_t phase, int max_sweeps, int flags)
{
    if (L.status[tile] & MGC_ST_SINK) mgcw_discharge_impl<true, REP>(w, L, tile, phase, max_sweeps, flags | ((flags & MGCW_BFS_SINK) ? MGCW_BFS : 0));
    else mgcw_discharge_impl<false, REP>(w, L, tile, phase, max_sweeps, flags);
}

/* ---------------------------------------------------------------------------------------
 * Global relabel, one tile of one pass, by one wave: relax the tile's labels from their current values over the
 * residual masks with the current halo; wake the neighbours across every face where a lowered label could lower
 * theirs; record which faces support the tile's labels (incremental relabel).  Same contract as mgc_relabel_tile.
 * ------------------------------------------------------------------------------------- */
template <class W>
MGC_HD void mgcw_relabel_tile(W& w, const MgcLattice& L, int tile, uint32_t next_epoch, int next_list, bool first_pass)
{
    if (first_pass && (!(L.status[tile] & 2u) || !mgc_owned(L, tile))) return;
    typename W::template Reg<int, 8> m, h0, h;
    typename W::template Reg<int, 1> st0; /* the tile's status word (rewritten in the tail: fetched here, with everything else) */
    const int64_t base = (int64_t)tile * MGC_TV;
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    w.lanes([&](int l) MGCW_INL { /* one trip to HBM: masks, labels, label halo, status word */
        st0(l, 0) = (int)L.status[tile];
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            m(l, K) = w.ld(L.rmask + base, K * 64 + l);
            h0(l, K) = w.ld(L.height + base, K * 64 + l);
        });
        mgcw_load_halo(w, L, tile, l);
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            h(l, K) = h0(l, K);
            w.S.hs[mgcw_hs(l, K)] = h0(l, K);
        });
    });
    mgcw_relax(w, h, [&](int l, auto KK, auto DD) MGCW_INL -> bool {
        constexpr int K = decltype(KK)::value;
        constexpr int D = decltype(DD)::value;
        return ((m(l, K) >> D) & 1) != 0;
    });
    /* which faces saw a label drop that could lower the neighbour; which faces support a label; did any label come down?
     * Every lane collects its own bits over its eight slots, then one vote per bit */
    typename W::template Reg<int, 1> bits; /* 0..5: wake across face D, 6..11: face D supports a label, 12: lowered */
    w.lanes([&](int l) MGCW_INL {
        const int y = l >> 3, x = l & 7;
        int b = 0;
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            const int hm = h(l, K);
            if (hm < h0(l, K)) b |= 1 << 12;
            mgcw_static_for<6>([&](auto DD) MGCW_INL {
                constexpr int D = decltype(DD)::value;
                if constexpr (D == 4 && K != 0) return;
                if constexpr (D == 5 && K != 7) return;
                const bool on_face = D == 0 ? x == 0 : (D == 1 ? x == 7 : (D == 2 ? y == 0 : (D == 3 ? y == 7 : true)));
                if (!on_face) return;
                const int hv = mgcw_nbr_label<K, D>(w, h, l); /* the halo voxel behind the face */
                if (hm < MGC_HINF && ((m(l, K) >> D) & 1) && hv + 1 == hm) b |= 1 << (6 + D);
                /* wake the neighbour across a face only if its adjacent voxel could improve: labels only go down during a
                 * relabel, so a halo value is an upper bound of the neighbour's current label */
                if (hm < h0(l, K) && hm + 1 < hv) b |= 1 << D;
            });
        });
        bits(l, 0) = b;
    });
    uint32_t wake = 0, dep = 0;
    mgcw_static_for<6>([&](auto DD) MGCW_INL {
        constexpr int D = decltype(DD)::value;
        if (w.any([&](int l) MGCW_INL -> bool { return ((bits(l, 0) >> D) & 1) != 0; })) wake |= 1u << D;
        if (w.any([&](int l) MGCW_INL -> bool { return ((bits(l, 0) >> (6 + D)) & 1) != 0; })) dep |= 1u << D;
    });
    const bool lowered = w.any([&](int l) MGCW_INL -> bool { return ((bits(l, 0) >> 12) & 1) != 0; });
    w.lanes([&](int l) MGCW_INL { /* one block of global traffic: wake-ups + labels.  The claim of a neighbour (a returning atomic)
                                     goes out BEFORE the label stores: a wave's memory operations retire in issue order */
        int woken = -1;
        bool won = false;
        if (l < 6 && ((wake >> l) & 1u)) {
            const int nt = mgc_tile_nbr(L, tz, ty, tx, l);
            if (nt >= 0 && mgc_owned(L, nt)) {
                woken = nt;
                won = w.atomic_exch(&L.rstamp[nt], next_epoch) != next_epoch;
            }
        }
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            if (h(l, K) < h0(l, K)) w.st(L.height + base, K * 64 + l, h(l, K));
        });
        if (l == 6) L.status[tile] = ((uint32_t)st0(l, 0) & ~((63u << MGC_ST_DEP_SHIFT) | (lowered ? MGC_ST_ALLINF : 0u))) | (dep << MGC_ST_DEP_SHIFT);
        if (won) { /* first to queue it for the next pass (what mgc_enqueue does after its claim) */
            const int sh = w.shard(L);
            const int pos = w.atomic_add(mgc_counter(L, next_list, sh), 1);
            L.list[next_list][(int64_t)sh * L.shard_cap + pos] = woken;
        }
    });
}

/* ---------------------------------------------------------------------------------------
 * After a global relabel: does the tile hold excess that can still reach the sink?  One wave per tile, no LDS, no
 * barrier: sixteen loads per lane, one vote.  Same contract as mgc_activate_tile.
 * ------------------------------------------------------------------------------------- */
template <class W>
MGC_HD void mgcw_activate_tile(W& w, const MgcLattice& L, int tile, uint32_t phase, bool exact)
{
    if (!mgc_owned(L, tile) || (L.status[tile] & MGC_ST_ALLINF)) return; /* all labels INF: nothing can reach the sink */
    /* !exact: the status word alone decides.  MGC_ST_EXCESS says "at its last visit some voxel held excess under a finite
     * label"; a finite label can still turn INF at a later global relabel (never the other way round), so the word can only
     * err towards a visit too many -- which finds nothing to push, clears the bit, and is not repeated.  The voxel-level
     * test below reads 6 KiB per candidate tile; it runs when the candidates are few, i.e. near the end of a solve, where
     * a wrong "still active" would cost a whole extra round of relabel + phases. */
    if (!exact) {
        w.lanes([&](int l) MGCW_INL {
            if (l == 0) {
                int tz, ty, tx;
                mgc_tile_coords(L, tile, tz, ty, tx);
                const uint32_t target = phase + ((mgc_tile_colour(L, tz, ty, tx) ^ (int)(phase & 1u)) & 1);
                mgc_enqueue(w, L, (int)(target & 3u), L.stamp, target, tile);
                w.atomic_add(&L.count[6], 1);
            }
        });
        return;
    }
    typename W::template Reg<double, 8> e;
    typename W::template Reg<int, 8> h;
    const double* const t_excess = L.excess + (int64_t)tile * MGC_TV;
    const int32_t* const t_height = L.height + (int64_t)tile * MGC_TV;
    w.lanes([&](int l) MGCW_INL {
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            e(l, K) = w.ld(t_excess, K * 64 + l);
            h(l, K) = w.ld(t_height, K * 64 + l);
        });
    });
    const bool act = w.any([&](int l) MGCW_INL -> bool {
        bool a = false;
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            a = a || (e(l, K) > 0.0 && h(l, K) < MGC_HINF);
        });
        return a;
    });
    if (!act) return;
    w.lanes([&](int l) MGCW_INL {
        if (l == 0) {
            int tz, ty, tx;
            mgc_tile_coords(L, tile, tz, ty, tx);
            const uint32_t target = phase + ((mgc_tile_colour(L, tz, ty, tx) ^ (int)(phase & 1u)) & 1);
            mgc_enqueue(w, L, (int)(target & 3u), L.stamp, target, tile);
            w.atomic_add(&L.count[6], 1);
        }
    });
}

/* ---------------------------------------------------------------------------------------
 * Absorb-only visit (before a global relabel no flow may be in flight in an outbox while the residual masks are read),
 * one wave per tile.  Lane k is face cell k: for every face across which the neighbour left flow (its outbox flag), the
 * lane adds the slot's content to the excess and to the reverse residual of the voxel behind the cell and empties the
 * slot -- faces in ascending order, one after the other, so a voxel on an edge or a corner receives its two or three
 * contributions in the order (and with the roundings) of mgc_absorb_tile.  Same contract as mgc_absorb_tile.
 * ------------------------------------------------------------------------------------- */
template <class W>
MGC_HD void mgcw_absorb_tile(W& w, const MgcLattice& L, int tile)
{
    if (!mgc_owned(L, tile)) return;
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    double* const t_excess = L.excess + (int64_t)tile * MGC_TV;
    uint8_t* const t_rmask = L.rmask + (int64_t)tile * MGC_TV;
    bool got = false;
    mgcw_static_for<6>([&](auto FF) MGCW_INL {
        constexpr int F = decltype(FF)::value;
        const int nt = mgc_tile_nbr(L, tz, ty, tx, F);
        if (nt < 0 || !((L.oflags[nt] >> (F ^ 1)) & 1u)) return; /* wave-uniform */
        double* const slots = L.obox + ((int64_t)nt * 6 + (F ^ 1)) * MGC_TF;
        double* const t_plane = L.rcap + ((int64_t)tile * 6 + F) * MGC_TV;
        const bool any = w.any([&](int l) MGCW_INL -> bool {
            const double d = w.ld(slots, l);
            if (d != 0.0) {
                const int v = mgc_face_voxel(F, l);
                const double r = w.ld(t_plane, v) + d;
                w.st(t_excess, v, w.ld(t_excess, v) + d);
                w.st(t_plane, v, r);
                if (r > 0.0) w.st(t_rmask, v, (uint8_t)(w.ld(t_rmask, v) | (1u << F)));
                w.st(slots, l, 0.0);
            }
            return d != 0.0;
        });
        got = got || any;
        w.lanes([&](int l) MGCW_INL {
            if (l == 0) w.atomic_and(&L.oflags[nt], ~(1u << (F ^ 1)));
        });
    });
    if (got) {
        w.lanes([&](int l) MGCW_INL {
            if (l == 0) L.status[tile] |= MGC_ST_EXCESS; /* flow arrived: the tile may hold excess now */
        });
    }
}

#endif /* MGC_WAVE_OPS_INL */
