/*
 * mgc_tile_ops.inl -- the per-tile operations of the lattice max-flow solver, written once
 * against a small "block executor" concept X so that the very same source is
 *   (a) the body of the HIP kernels (X = GpuBlock, mgc_kernels.hip: 512 threads = one 8x8x8
 *       tile, par() = body + __syncthreads(), Reg<T> = a register), and
 *   (b) executed on the host by tests/hostsim (X = HostBlock: par() = a loop over the 512
 *       lanes, Reg<T> = an array) so the algorithm is parity-tested against the BK oracle in
 *       the CPU-only test tier.  The host executor is test infrastructure; it is not compiled
 *       into the product library.
 *
 * Replaces (reference): Graph::maxflow and its helpers, lib/maxflow/src/maxflow.cpp:119-604
 * (BK search trees) -- by a different algorithm with the same result definition:
 * Goldberg-Tarjan push-relabel in "region discharge" form (Delong & Boykov 2008) on 8x8x8
 * tiles with exact in-tile distance labels.  Labels come out identical because
 * what_segment() (graph.h:561-571) == "can reach the sink in the residual graph of a maximum
 * (pre)flow", which does not depend on the algorithm.
 *
 * X concept:
 *   X::Reg<T>            per-lane value, indexed with the lane id
 *   x.par(f)             run f(lane) for all 512 lanes, then barrier
 *   x.any(f)             barrier-OR of f(lane) over all lanes
 *   x.wpar(f)            run f(lane) for all lanes WITHOUT a barrier: f may only touch the lane's own registers / LDS slots
 *   x.wave_any(f)        OR of f(lane) over the 64 lanes of a wave (one z-layer of the tile); uniform per wave
 *   x.shift(dst, src, k) dst[lane] = src[lane + k] inside the wave, 0.0 at its ends (ds_bpermute on the GPU): the +-x / +-y
 *                        hand-offs of a push sweep need neither an LDS slot nor a barrier
 *   x.S                  MgcTileShared& (LDS)
 *   x.atomic_add/or/and/exch   device-scope atomics on global words
 *   x.shard(L)           which region of a work list this executor appends to (0 .. L.nshard - 1)
 *   x.async_to_lds / x.async_wait   HBM -> LDS copy without a register round trip (global_load_lds on gfx950)
 *   x.tile_labels(mask, out)   exact in-tile distance labels from scratch given the halo in x.S.hs: out[lane] and the
 *                              tile's own cells of x.S.hs (chaotic relaxation in LDS, mgc_tile_bfs below; the residual
 *                              mask is evaluated once)
 */
#ifndef MGC_TILE_OPS_INL
#define MGC_TILE_OPS_INL

#include <type_traits>

#include "mgc_common.h"

struct alignas(16) MgcTileShared {
    int32_t hs[1000];          /* 10x10x10 distance labels: the tile plus a one-voxel halo */
    double  out[2][MGC_TV];    /* per-direction push hand-off, double buffered             */
    double  r[6][MGC_TV];      /* residual n-link capacities of the tile being discharged  */
    int32_t nbr[8];            /* neighbour tile ids                                       */
    int32_t inflag[8];         /* neighbour f has flow for us in its outbox                */
    double  inbox[6][MGC_TF];  /* that flow, staged by the loading lanes                   */
    int32_t faceflag[8];
    int32_t depflag[8];        /* relabel: face f supports some label of the tile           */
    int32_t flag[2];
    int32_t satflag;           /* discharge: some arc (or sink link) of the tile was saturated */
    int32_t excflag;           /* discharge: some voxel still holds excess at store time      */
};

MGC_HD int mgc_hs_index(int z, int y, int x) { return (z + 1) * 100 + (y + 1) * 10 + (x + 1); }

/* step in hs[] / in the local index for direction d */
MGC_HD int mgc_hs_step(int d) { return d == 0 ? -1 : d == 1 ? 1 : d == 2 ? -10 : d == 3 ? 10 : d == 4 ? -100 : 100; }
MGC_HD int mgc_loc_step(int d) { return d == 0 ? -1 : d == 1 ? 1 : d == 2 ? -8 : d == 3 ? 8 : d == 4 ? -64 : 64; }

/* is the neighbour of local voxel (z,y,x) in direction d inside the same tile? */
MGC_HD bool mgc_inside(int d, int z, int y, int x)
{
    const int c = (d >> 1) == 0 ? x : ((d >> 1) == 1 ? y : z);
    return (d & 1) ? (c < MGC_T - 1) : (c > 0);
}

template <class X>
MGC_HD void mgc_enqueue(X& x, const MgcLattice& L, int listid, uint32_t* stamps, uint32_t epoch, int tile)
{
    if (!mgc_owned(L, tile)) return; /* ghost tiles are discharged / relabelled by the slab that owns them */
    if (x.atomic_exch(&stamps[tile], epoch) != epoch) {
        const int sh = x.shard(L); /* the appender's region of the list (MgcLattice::scount) */
        const int pos = x.atomic_add(mgc_counter(L, listid, sh), 1);
        L.list[listid][(int64_t)sh * L.shard_cap + pos] = tile;
    }
}

/* every lane: neighbour tile ids into LDS, face flags cleared.  Needs a barrier afterwards. */
template <class X>
MGC_HD void mgc_load_nbrs(X& x, const MgcLattice& L, int tile, int t)
{
    if (t < 6) {
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const int nt = mgc_tile_nbr(L, tz, ty, tx, t);
        x.S.nbr[t] = nt;
        x.S.inflag[t] = (nt >= 0 && L.obox) ? (int32_t)((L.oflags[nt] >> (t ^ 1)) & 1u) : 0;
        x.S.faceflag[t] = 0;
        x.S.depflag[t] = 0;
    }
    if (t < 2) x.S.flag[t] = 0;
    if (t == 2) x.S.satflag = 0;
    if (t == 3) x.S.excflag = 0;
}

/* halo labels: 6 faces x 64 voxels, read from the neighbour tiles' label arrays */
template <class X>
MGC_HD void mgc_load_halo(X& x, const MgcLattice& L, int t)
{
    if (t < 6 * MGC_TF) {
        const int f = t >> 6, k = t & 63;
        const int nt = x.S.nbr[f];
        const int mine = mgc_face_voxel(f, k);           /* my voxel on that face          */
        const int theirs = mgc_face_voxel(f ^ 1, k);     /* the voxel it touches next door */
        const int z = mine >> 6, y = (mine >> 3) & 7, xx = mine & 7;
        const int32_t h = nt < 0 ? MGC_HINF : L.height[(int64_t)nt * MGC_TV + theirs];
        x.S.hs[mgc_hs_index(z, y, xx) + mgc_hs_step(f)] = h;
    }
}

/* ONE trip to HBM for everything a tile needs from its six neighbours: lane k < 384 fetches the label of the
 * voxel its face cell touches AND the outbox slot the neighbour may have filled for it (slots nobody filled
 * hold 0.0, so they are fetched unconditionally instead of waiting for the flag word first).  Results go to
 * LDS (hs halo, inbox); filled slots are emptied.  Must run in the same par() as mgc_load_nbrs -- it computes
 * the neighbour ids itself. */
template <class X>
MGC_HD void mgc_load_halo_inbox(X& x, const MgcLattice& L, int tile, int t)
{
    if (t < 6 * MGC_TF) {
        const int f = t >> 6, k = t & 63;
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
        const int mine = mgc_face_voxel(f, k);
        const int z = mine >> 6, y = (mine >> 3) & 7, xx = mine & 7;
        int32_t h = MGC_HINF;
        double din = 0.0;
        if (nt >= 0) {
            h = L.height[(int64_t)nt * MGC_TV + mgc_face_voxel(f ^ 1, k)];
            double* slot = &L.obox[((int64_t)nt * 6 + (f ^ 1)) * MGC_TF + k];
            din = *slot;
            if (din != 0.0) *slot = 0.0;
        }
        x.S.hs[mgc_hs_index(z, y, xx) + mgc_hs_step(f)] = h;
        x.S.inbox[f][k] = din;
    }
}

/* one lane absorbs what the neighbour tiles pushed across its (up to three) faces:
 * e += delta, reverse residual += delta (the receiving half of a push, maxflow.cpp:268-271 analogue) */
template <class X, class RegD, class RAdd>
MGC_HD bool mgc_absorb_lane(X& x, const MgcLattice& L, int t, RegD& e, RAdd radd)
{
    const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
    bool got = false;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        if (mgc_inside(d, z, y, xx)) continue;
        if (!x.S.inflag[d]) continue;
        const int nt = x.S.nbr[d];
        const int k = mgc_face_index(d >> 1, z, y, xx);
        double* slot = &L.obox[((int64_t)nt * 6 + (d ^ 1)) * MGC_TF + k];
        const double delta = *slot;
        if (delta != 0.0) {
            e[t] += delta;
            radd(d, delta);
            *slot = 0.0;
            got = true;
        }
    }
    return got;
}

/* after a barrier: lanes 0..5 retire the inbox flags they consumed */
template <class X>
MGC_HD void mgc_clear_inbox_flags(X& x, const MgcLattice& L, int t)
{
    if (t < 6 && x.S.inflag[t]) x.atomic_and(&L.oflags[x.S.nbr[t]], ~(1u << (t ^ 1)));
}

/* ---------------------------------------------------------------------------------------
 * In-tile exact distance labels by chaotic relaxation from scratch:
 *   h(u) = 1 if u has residual capacity to the sink, else 1 + min h(v) over residual arcs u->v,
 * halo labels frozen.  `mask(t)` yields the residual bit mask of lane t.  Own labels must be
 * MGC_HINF (or any upper bound) on entry.  Values only decrease, so concurrent in-place
 * updates are benign and the fixpoint (exact distances given the halo) is unique.
 * ------------------------------------------------------------------------------------- */
template <class X, class MaskFn>
MGC_HD void mgc_tile_bfs(X& x, MaskFn mask)
{
    for (;;) {
        auto relax = [&](int t) -> bool {
            const int m = mask(t);
            if (!m) return false;
            const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
            const int me = mgc_hs_index(z, y, xx);
            int cand = (m & MGC_MASK_SINK) ? 1 : MGC_HINF;
            /* branch-free: all six neighbour labels are fetched back to back (one LDS latency) and masked afterwards; with
             * a branch per direction the compiler issues six DEPENDENT reads, each waiting for the previous one */
            int hv[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) hv[d] = x.S.hs[me + mgc_hs_step(d)];
            const int own = x.S.hs[me];
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const int c = ((m >> d) & 1) ? hv[d] + 1 : MGC_HINF;
                cand = c < cand ? c : cand;
            }
            if (cand < own) {
                x.S.hs[me] = cand;
                return true;
            }
            return false;
        };
        /* two plain relaxation rounds per convergence test: the OR-reduction costs more than a round.  (Measured on
         * MI355X: repeating the relaxation inside a wave between barriers -- 4 rounds, wave-level fence -- made the
         * labels section 23 % slower at 32 waves/CU: the extra LDS/VALU work outweighs the saved barriers.) */
        x.par([&](int t) { (void)relax(t); });
        x.par([&](int t) { (void)relax(t); });
        if (!x.any(relax)) break;
    }
}

/* ---------------------------------------------------------------------------------------
 * Global relabel, one tile of one pass: recompute the tile's labels from its residual mask
 * with the current halo; wake the neighbours across every face where a lowered label could lower theirs.
 * Passes repeat (driver) until no tile changes: exact distances to the sink, MGC_HINF for
 * voxels that cannot reach it -- the set the reference reads out with what_segment().
 * ------------------------------------------------------------------------------------- */
template <class X>
MGC_HD void mgc_relabel_tile(X& x, const MgcLattice& L, int tile, uint32_t next_epoch, int next_list, bool first_pass)
{
    /* first pass of a global relabel: every label is INF, so only tiles holding a sink arc can seed anything */
    if (first_pass && (!(L.status[tile] & 2u) || !mgc_owned(L, tile))) return;
    typename X::template Reg<int> m, h0, stw;
    const int64_t base = (int64_t)tile * MGC_TV;
    x.par([&](int t) { /* one trip to HBM: masks, own labels, label halo, the status word */
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        mgc_load_nbrs(x, L, tile, t);
        if (t == 6) stw[t] = (int)L.status[tile]; /* (rewritten below: fetched here, with everything else) */
        m[t] = L.rmask[base + t];
        h0[t] = L.height[base + t];
        x.S.hs[mgc_hs_index(z, y, xx)] = h0[t];
        if (t < 6 * MGC_TF) {
            const int f = t >> 6, k = t & 63;
            int tz, ty, tx;
            mgc_tile_coords(L, tile, tz, ty, tx);
            const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
            const int mine = mgc_face_voxel(f, k);
            x.S.hs[mgc_hs_index(mine >> 6, (mine >> 3) & 7, mine & 7) + mgc_hs_step(f)] =
                nt < 0 ? MGC_HINF : L.height[(int64_t)nt * MGC_TV + mgc_face_voxel(f ^ 1, k)];
        }
    });
    /* relax from the CURRENT labels, not from scratch: inside a global relabel labels only go down, so they are upper
     * bounds of the distances and the chaotic relaxation converges to the same exact values -- in a few rounds when the
     * tile is revisited because a neighbour improved (most visits), instead of a full in-tile BFS every time */
    mgc_tile_bfs(x, [&](int t) { return m[t]; });
    x.par([&](int t) { /* LDS only: which faces saw a label drop; which faces support a label (incremental relabel) */
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        const int me = mgc_hs_index(z, y, xx);
        const int hm = x.S.hs[me];
        if (hm < MGC_HINF) {
            for (int d = 0; d < 6; ++d)
                if (((m[t] >> d) & 1) && !mgc_inside(d, z, y, xx) && x.S.hs[me + mgc_hs_step(d)] + 1 == hm) x.S.depflag[d] = 1;
        }
        if (hm < h0[t]) {
            x.S.flag[0] = 1; /* some label of the tile came down: it is not "all INF" (any more) */
            /* wake the neighbour across a face only if its adjacent voxel could improve: labels only go down during a
             * relabel, so a halo value is an upper bound of the neighbour's current label and "hm + 1 >= halo" stays
             * true.  This drops the back-wakes (the tile the wave came from) and most side-wakes. */
            for (int d = 0; d < 6; ++d)
                if (!mgc_inside(d, z, y, xx) && hm + 1 < x.S.hs[me + mgc_hs_step(d)]) x.S.faceflag[d] = 1;
        }
    });
    x.par([&](int t) { /* one block of global traffic: wake-ups + labels.  The claim of a neighbour (a returning atomic) is issued
                          BEFORE the label stores: a thread's memory operations retire in issue order, behind the stores it would
                          wait for them to drain first */
        const int h = x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)];
        int wake = -1;
        bool won = false;
        if (t < 6 && x.S.faceflag[t] && x.S.nbr[t] >= 0 && mgc_owned(L, x.S.nbr[t])) {
            wake = x.S.nbr[t];
            won = x.atomic_exch(&L.rstamp[wake], next_epoch) != next_epoch;
        }
        if (h < h0[t]) L.height[base + t] = h;
        if (t == 6) {
            uint32_t dep = 0;
            for (int f = 0; f < 6; ++f) dep |= x.S.depflag[f] ? (1u << f) : 0u;
            L.status[tile] = ((uint32_t)stw[t] & ~((63u << MGC_ST_DEP_SHIFT) | (x.S.flag[0] ? MGC_ST_ALLINF : 0u))) | (dep << MGC_ST_DEP_SHIFT);
        }
        if (won) { /* first to queue it for the next pass (what mgc_enqueue does after its claim) */
            const int sh = x.shard(L);
            const int pos = x.atomic_add(mgc_counter(L, next_list, sh), 1);
            L.list[next_list][(int64_t)sh * L.shard_cap + pos] = wake;
        }
    });
}

/* ---------------------------------------------------------------------------------------
 * Absorb-only pass (run on every tile before a global relabel so that no flow is "in
 * flight" in an outbox while the residual masks are read).
 * ------------------------------------------------------------------------------------- */
template <class X>
MGC_HD void mgc_absorb_tile(X& x, const MgcLattice& L, int tile)
{
    if (!mgc_owned(L, tile)) return;
    typename X::template Reg<double> e, r[6];
    const int64_t base = (int64_t)tile * MGC_TV;
    x.par([&](int t) { mgc_load_nbrs(x, L, tile, t); });
    const bool pending = x.any([&](int t) -> bool { return t < 6 && x.S.inflag[t]; });
    if (!pending) return;
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        if (xx != 0 && xx != 7 && y != 0 && y != 7 && z != 0 && z != 7) return;
        e[t] = L.excess[base + t];
#pragma unroll
        for (int d = 0; d < 6; ++d) r[d][t] = L.rcap[((int64_t)tile * 6 + d) * MGC_TV + t];
        if (mgc_absorb_lane(x, L, t, e, [&](int d, double delta) { r[d][t] += delta; })) {
            L.excess[base + t] = e[t];
            int m = L.rmask[base + t];
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                L.rcap[((int64_t)tile * 6 + d) * MGC_TV + t] = r[d][t];
                if (r[d][t] > 0.0) m |= 1 << d;
            }
            L.rmask[base + t] = (uint8_t)m;
        }
    });
    x.par([&](int t) {
        mgc_clear_inbox_flags(x, L, t);
        if (t == 6) L.status[tile] |= MGC_ST_EXCESS; /* flow arrived: the tile may hold excess now */
    });
}

/* ---------------------------------------------------------------------------------------
 * After a global relabel: does the tile hold excess that can still reach the sink?
 * ------------------------------------------------------------------------------------- */
template <class X>
MGC_HD void mgc_activate_tile(X& x, const MgcLattice& L, int tile, uint32_t phase)
{
    if (!mgc_owned(L, tile) || (L.status[tile] & MGC_ST_ALLINF)) return; /* all labels INF: nothing can reach the sink */
    const int64_t base = (int64_t)tile * MGC_TV;
    const bool act = x.any([&](int t) -> bool { return L.excess[base + t] > 0.0 && L.height[base + t] < MGC_HINF; });
    x.par([&](int t) {
        if (t == 0 && act) {
            int tz, ty, tx;
            mgc_tile_coords(L, tile, tz, ty, tx);
            const uint32_t target = phase + ((mgc_tile_colour(L, tz, ty, tx) ^ (int)(phase & 1u)) & 1);
            mgc_enqueue(x, L, (int)(target & 3u), L.stamp, target, tile);
            x.atomic_add(&L.count[6], 1);
        }
    });
}

/* ---------------------------------------------------------------------------------------
 * Region discharge of one tile (colour phase `phase`; the six face neighbours are idle).
 *
 *   load state -> absorb inbox -> repeat { exact in-tile labels ; push sweeps } -> store.
 *
 * A sweep visits sink + 6 directions; in direction d every lane with excess pushes
 * min(excess, residual) along an admissible arc (label drop of exactly 1).  Direction by
 * direction each voxel receives from exactly one neighbour, so the receiving half is a plain
 * LDS hand-off: no atomics, fixed order of floating point operations (bit-reproducible).
 * Saturating pushes leave an exact 0.0 (x - x), which is what makes the residual graph --
 * hence the labels -- canonical.  Pushes over a tile face go to this tile's outbox and are
 * absorbed by the neighbour at the start of its next discharge.
 * ------------------------------------------------------------------------------------- */
template <class X>
MGC_HD void mgc_discharge_tile(X& x, const MgcLattice& L, int tile, uint32_t phase, int max_cycles, int max_sweeps)
{
    /* per lane: excess, sink residual, outbox accumulators, label, residual mask.  The six n-link residuals live
     * in LDS (x.S.r): that keeps the kernel under 64 VGPRs, i.e. 4 workgroups (32 waves) per CU, which is what
     * hides the HBM latency of the load / store phases of the neighbouring workgroups. */
    typename X::template Reg<double> e, snk, ob0, ob1, ob2;
    typename X::template Reg<int> hme, msk, lostarc; /* (lostarc: this voxel saturated an arc or its sink link) */
    /* per-tile (wave-uniform) base pointers + 32-bit lane offsets: SGPR-base addressing, fewer VGPRs */
    double* const t_excess = L.excess + (int64_t)tile * MGC_TV;
    double* const t_sink = L.sink + (int64_t)tile * MGC_TV;
    double* const t_rcap = L.rcap + (int64_t)tile * 6 * MGC_TV;
    double* const t_obox = L.obox + (int64_t)tile * 6 * MGC_TF;
    uint8_t* const t_rmask = L.rmask + (int64_t)tile * MGC_TV;
    int32_t* const t_height = L.height + (int64_t)tile * MGC_TV;

    /* was two dependent trips to HBM: (1) neighbour outbox flags + the tile's own state, issued together;
     * kept for reference: everything is fetched in ONE trip now */
    x.par([&](int t) {
        /* issue the state loads FIRST: the halo / inbox code below branches on loaded values, and loads placed
         * after such a branch would only be issued once the first batch has returned (a second trip to HBM) */
        /* the 24 KiB of residuals go HBM -> LDS directly (global_load_lds DMA on gfx950): no VGPR round trip, which
         * is what used to spill in this phase */
        x.async_to_lds(t, &x.S.r[0][0], t_rcap, 6 * MGC_TV * (int)sizeof(double));
        e[t] = t_excess[(unsigned)t];
        snk[t] = t_sink[(unsigned)t]; /* (meaningful only under MGC_ST_SINK: selected below, after the loads are on their way) */
        const uint32_t st_now = L.status[tile];
        mgc_load_nbrs(x, L, tile, t);
        mgc_load_halo_inbox(x, L, tile, t);
        ob0[t] = ob1[t] = ob2[t] = 0.0;
        lostarc[t] = 0;
        if (!(st_now & MGC_ST_SINK)) snk[t] = 0.0; /* the build writes the sink plane only where a tile has a sink link */
        x.async_wait(); /* the DMA must have landed before the barrier that ends this step */
    });
    x.par([&](int t) { /* absorb the staged inbox (LDS only): e += delta, reverse residual += delta, fixed face order */
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        for (int d = 0; d < 6; ++d) {
            if (mgc_inside(d, z, y, xx)) continue;
            const double delta = x.S.inbox[d][mgc_face_index(d >> 1, z, y, xx)];
            if (delta != 0.0) {
                e[t] += delta;
                x.S.r[d][t] += delta;
                /* on RADIAL labels flow that comes in marks the tile DIRTY: the push that sent it was admissible under the radial labels only,
                 * and the residual arc it opened back towards the sender may undercut the exact label kept aside for this voxel (see mgcw_discharge_impl) */
                if (max_cycles == -3) x.S.satflag = 1; /* (-3: a radial cycle that is not the first of its solve) */
            }
        }
        mgc_clear_inbox_flags(x, L, t);
    });
    x.mark(L, 0); /* load + absorb */

    bool active = false;
    int sweep_id = 0;
    /* max_cycles < 0: no exact in-tile labelling -- the stored labels are valid lower bounds (distances only grow) and the
     * local relabel at the end of every sweep raises the voxels that are stuck (what the one-wave-per-tile form does) */
    const bool stored_labels = max_cycles < 0;
    const bool any_saturation = max_cycles <= -2; /* the stored labels are RADIAL labels (mgc_dt_ops.inl), not distances: every saturated arc marks the tile DIRTY */
    if (stored_labels) max_cycles = 1;
    for (int cyc = 0; cyc < max_cycles; ++cyc) {
        if (stored_labels) {
            x.par([&](int t) {
                hme[t] = t_height[(unsigned)t];
                x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = hme[t];
            });
        } else {
            /* exact labels given the frozen halo */
            x.tile_labels([&](int t) {
                int m = snk[t] > 0.0 ? MGC_MASK_SINK : 0;
                for (int d = 0; d < 6; ++d) m |= (x.S.r[d][t] > 0.0) ? (1 << d) : 0;
                msk[t] = m;
                return m;
            }, hme);
        }
        active = x.any([&](int t) -> bool { return e[t] > 0.0 && hme[t] < MGC_HINF; });
        x.mark(L, 1); /* in-tile labels */
        if (!active) break;

        for (int sw = 0; sw < max_sweeps; ++sw, ++sweep_id) {
            const int fl = sweep_id & 1;
            /* One sweep = sink, then the six directions in turn; a voxel receives what its neighbour pushed in direction
             * d before it pushes in direction d + 1 (Gauss-Seidel order).  A wave holds one z-layer of the tile (lane =
             * (y, x)), so the +-x and +-y neighbours are lanes of the SAME wave: those four exchanges are lane shifts
             * (ds_bpermute), need no LDS slot and no workgroup barrier, and a wave without excess skips them outright.
             * Only the +-z exchanges cross waves (LDS slot + barrier): 2 barriers per sweep instead of 7. */
            typename X::template Reg<double> dl, din;
            auto push = [&](int t, int d) -> double { /* admissible push of lane t in direction d; returns the amount */
                /* both operands are fetched unconditionally and together: one LDS latency, not two dependent ones */
                const double rd = x.S.r[d][t];
                const int hn = x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7) + mgc_hs_step(d)];
                double delta = 0.0;
                if (e[t] > 0.0 && rd > 0.0 && hme[t] < MGC_HINF && hn == hme[t] - 1) {
                    delta = e[t] < rd ? e[t] : rd;
                    e[t] -= delta;
                    x.S.r[d][t] = rd - delta; /* saturating push: rd - rd == 0.0 exactly */
                    x.S.flag[fl] = 1;
                    if (delta == rd) {
                        if (stored_labels) lostarc[t] = 1; /* (looked at once, behind the sweeps) */
                        else x.S.satflag = 1;
                    }
                }
                return delta;
            };
            auto leave = [&](int t, int d, double delta) { /* flow that leaves the tile across face d */
                if (delta != 0.0) {
                    if ((d >> 1) == 0) ob0[t] += delta;
                    else if ((d >> 1) == 1) ob1[t] += delta;
                    else ob2[t] += delta;
                }
            };
            auto recv = [&](int t, int dp, double v) { /* what the neighbour pushed in direction dp arrives here */
                if (v != 0.0 && mgc_inside(dp ^ 1, t >> 6, (t >> 3) & 7, t & 7)) {
                    e[t] += v;
                    x.S.r[dp ^ 1][t] += v;
                }
            };
            auto inplane = [&](int t, int d) { /* push in direction d (0..3); dl = what stays inside the tile */
                const double delta = push(t, d);
                if (mgc_inside(d, t >> 6, (t >> 3) & 7, t & 7)) dl[t] = delta;
                else { dl[t] = 0.0; leave(t, d, delta); }
            };
            x.wpar([&](int t) { /* push to the sink first: always admissible (label 1 -> 0) */
                if (e[t] > 0.0 && snk[t] > 0.0) {
                    const double delta = e[t] < snk[t] ? e[t] : snk[t];
                    e[t] -= delta;
                    snk[t] -= delta;
                    x.S.flag[fl] = 1;
                    if (snk[t] == 0.0) {
                        if (stored_labels) lostarc[t] = 1;
                        else x.S.satflag = 1;
                    }
                }
                din[t] = 0.0;
            });
            if (x.wave_any([&](int t) -> bool { return e[t] > 0.0 && hme[t] < MGC_HINF; })) { /* uniform per wave */
                x.wpar([&](int t) { inplane(t, 0); });
                x.shift(din, dl, 1);  /* -x: from the lane at x + 1 */
                x.wpar([&](int t) { recv(t, 0, din[t]); inplane(t, 1); });
                x.shift(din, dl, -1); /* +x: from x - 1 */
                x.wpar([&](int t) { recv(t, 1, din[t]); inplane(t, 2); });
                x.shift(din, dl, 8);  /* -y: from y + 1 */
                x.wpar([&](int t) { recv(t, 2, din[t]); inplane(t, 3); });
                x.shift(din, dl, -8); /* +y: from y - 1 */
            }
            x.par([&](int t) {
                recv(t, 3, din[t]);
                const double delta = push(t, 4);
                if (mgc_inside(4, t >> 6, (t >> 3) & 7, t & 7)) x.S.out[0][t] = delta;
                else leave(t, 4, delta);
            });
            x.par([&](int t) {
                if (t == 0) x.S.flag[fl ^ 1] = 0; /* everybody has read it by now */
                if (mgc_inside(5, t >> 6, (t >> 3) & 7, t & 7)) recv(t, 4, x.S.out[0][t + MGC_TF]);
                const double delta = push(t, 5);
                if (mgc_inside(5, t >> 6, (t >> 3) & 7, t & 7)) x.S.out[1][t] = delta;
                else leave(t, 5, delta);
            });
            x.par([&](int t) {
                if (mgc_inside(4, t >> 6, (t >> 3) & 7, t & 7)) recv(t, 5, x.S.out[1][t - MGC_TF]);
            });
            /* local relabel (classic push-relabel step): a voxel that still holds excess and has no admissible arc left
             * rises to 1 + the lowest label behind a residual arc.  Labels stay valid lower bounds of the distance (no
             * push runs in this step; a neighbour's label read here is its old or its new one, both lower bounds), so the
             * next sweep can go on without recomputing the exact labels of the whole tile. */
            x.par([&](int t) {
                if (e[t] > 0.0 && hme[t] < MGC_HINF) {
                    const int me = mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7);
                    int cand = snk[t] > 0.0 ? 1 : MGC_HINF;
#pragma unroll
                    for (int d = 0; d < 6; ++d) { /* branch-free: twelve independent LDS reads */
                        const double rd = x.S.r[d][t];
                        const int hv = x.S.hs[me + mgc_hs_step(d)];
                        cand = (rd > 0.0 && hv < MGC_HINF && hv + 1 < cand) ? hv + 1 : cand;
                    }
                    if (cand > hme[t]) {
                        hme[t] = cand;
                        x.S.hs[me] = cand;
                        x.S.satflag = 1; /* a label rose: whoever stood on it has to be looked at */
                        if (cand < MGC_HINF) x.S.flag[fl] = 1; /* it can push again next sweep */
                    }
                }
            });
            x.mark(L, 2); /* one push sweep */
            if (!x.S.flag[fl]) break; /* uniform: written before the last barrier */
        }
    }
    if (active) {
        /* cycle budget exhausted: is there still something to do with the current labels? */
        active = x.any([&](int t) -> bool { return e[t] > 0.0 && hme[t] < MGC_HINF; });
    }
    /* DIRTY (the next global relabel recomputes the tile and whoever depends on it) iff a label rose, or a voxel that saturated an
     * arc has no residual arc one label down left: a voxel that keeps one of its supports keeps its distance.  A support in a
     * NEIGHBOUR tile counts only if the tile's support bits name that neighbour (then the tile turns suspect with it): the bits are
     * those of the tile's last relabel visit, and a neighbour's voxel may have come down to "one below" later in that relabel
     * without waking anybody (only a label that can IMPROVE a neighbour wakes it) -- a support nobody watches.  (With an exact
     * in-tile labelling per discharge the stored labels are not what the pushes followed: any saturation counts there.) */
    if (stored_labels) {
        x.par([&](int t) {
            if (lostarc[t] && hme[t] < MGC_HINF) {
                const int me = mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7);
                bool kept = snk[t] > 0.0; /* (a label of 1 stands on the sink link) */
                const uint32_t stk = L.status[tile]; /* (nobody else writes it during this launch) */
#pragma unroll
                for (int d = 0; d < 6; ++d) /* (a support in a neighbour tile counts only if the tile's support bits watch that neighbour: see below) */
                    kept = kept || (x.S.r[d][t] > 0.0 && x.S.hs[me + mgc_hs_step(d)] == hme[t] - 1 && (mgc_inside(d, t >> 6, (t >> 3) & 7, t & 7) || ((stk >> (MGC_ST_DEP_SHIFT + d)) & 1u)));
                if (!kept || any_saturation) x.S.satflag = 1;
            }
        });
    }
    const bool has_sink = x.any([&](int t) -> bool { return snk[t] > 0.0; });
    x.mark(L, 4); /* tail votes */

    /* which faces carry flow out of the tile (LDS only) ... */
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        if (ob0[t] != 0.0) x.S.faceflag[xx == 0 ? 0 : 1] = 1;
        if (ob1[t] != 0.0) x.S.faceflag[y == 0 ? 2 : 3] = 1;
        if (ob2[t] != 0.0) x.S.faceflag[z == 0 ? 4 : 5] = 1;
        if (e[t] > 0.0 && x.S.hs[mgc_hs_index(z, y, xx)] < MGC_HINF) x.S.excflag = 1; /* (excess under an INF label is dead for good) */
    });
    x.mark(L, 5); /* face flags */
    /* ... then ONE block of global stores: state, masks, labels, outbox, wake-ups */
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        t_excess[(unsigned)t] = e[t];
        t_sink[(unsigned)t] = snk[t];
        int m = snk[t] > 0.0 ? MGC_MASK_SINK : 0;
        for (int d = 0; d < 6; ++d) {
            const double rd = x.S.r[d][t];
            t_rcap[(unsigned)(d * MGC_TV + t)] = rd;
            m |= (rd > 0.0) ? (1 << d) : 0;
        }
        t_rmask[(unsigned)t] = (uint8_t)m;
        t_height[(unsigned)t] = x.S.hs[mgc_hs_index(z, y, xx)];
        /* plain stores: the neighbour emptied these slots when it last absorbed, and it always runs (or absorb_all
         * does) between two of our discharges */
        if (ob0[t] != 0.0) t_obox[(unsigned)((xx == 0 ? 0 : 1) * MGC_TF + mgc_face_index(0, z, y, xx))] = ob0[t];
        if (ob1[t] != 0.0) t_obox[(unsigned)((y == 0 ? 2 : 3) * MGC_TF + mgc_face_index(1, z, y, xx))] = ob1[t];
        if (ob2[t] != 0.0) t_obox[(unsigned)((z == 0 ? 4 : 5) * MGC_TF + mgc_face_index(2, z, y, xx))] = ob2[t];
        if (t < 6 && x.S.faceflag[t]) {
            x.atomic_or(&L.oflags[tile], 1u << t);
            mgc_enqueue(x, L, (int)((phase + 1) & 3u), L.stamp, phase + 1, x.S.nbr[t]);
        }
        if (t == 6 && active) mgc_enqueue(x, L, (int)((phase + 2) & 3u), L.stamp, phase + 2, tile);
        /* DIRTY only if a residual arc disappeared: otherwise no distance in the tile (or through it) can have changed */
        if (t == 7) L.status[tile] = (L.status[tile] & ~(MGC_ST_SINK | MGC_ST_EXCESS)) | (has_sink ? MGC_ST_SINK : 0u) | (x.S.satflag ? MGC_ST_DIRTY : 0u) | (x.S.excflag ? MGC_ST_EXCESS : 0u);
    });
    x.mark(L, 3); /* store */
}

/* ---------------------------------------------------------------------------------------
 * Incremental global relabel.  Distances to the sink never decrease, and a tile's exact labels stay exact as
 * long as a supporting path survives.  A tile is SUSPECT when it was discharged since the last global relabel
 * (arcs may have saturated) or when one of the faces that support its labels leads into a suspect tile
 * (transitive closure, tile-level, one thread per tile per pass).  Only suspect tiles are reset to INF and
 * recomputed; everything else (typically the whole sink side of the cut) keeps its labels.
 * ------------------------------------------------------------------------------------- */
MGC_HD bool mgc_suspect_tile(const MgcLattice& L, int tile)
{
    const uint32_t st = L.status[tile];
    if (st & MGC_ST_SUSPECT) return false;
    bool sus = (st & MGC_ST_DIRTY) != 0;
    if (!sus) {
        const uint32_t dep = (st >> MGC_ST_DEP_SHIFT) & 63u;
        if (dep) {
            int tz, ty, tx;
            mgc_tile_coords(L, tile, tz, ty, tx);
            for (int f = 0; f < 6 && !sus; ++f)
                if ((dep >> f) & 1u) {
                    const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
                    sus = nt >= 0 && (L.status[nt] & MGC_ST_SUSPECT);
                }
        }
    }
    if (sus) L.status[tile] = st | MGC_ST_SUSPECT;
    return sus;
}

/* A suspect tile is reset on BOTH sides of a slab border (the ghost mirrors the owner's flags): the ghost's labels are INF
 * from then on, so the owner's shadow of "what the neighbour holds" must say INF too -- otherwise a tile whose labels are
 * recomputed to their old values would never travel again and the neighbour would keep INF.  k = 0..63. */
MGC_HD void mgc_shadow_reset(const MgcLattice& L, int tile, int k)
{
    const int T = L.gy * L.gx, layer = tile / T, i = tile % T;
    if (L.hshadow[0] && L.tz_own_lo > 0 && layer == L.tz_own_lo) L.hshadow[0][(int64_t)i * MGC_TF + k] = MGC_HINF;
    if (L.hshadow[1] && L.tz_own_hi < L.gz && layer == L.tz_own_hi - 1) L.hshadow[1][(int64_t)i * MGC_TF + k] = MGC_HINF;
}

/* suspect tiles: labels := INF, flags retired; queued for the first relabel pass if a label can come from somewhere -- the
 * tile holds a sink link, or a face neighbour keeps labels (it is neither suspect nor all-INF; a neighbour that is being
 * reset at this moment shows one bit or the other, whichever side of its status store this read falls on), or lies in
 * another slab.  A reset tile in the middle of reset tiles would relax INF against INF on that visit; it is woken when
 * the label wave reaches a neighbour (mgc_relabel_tile wakes across a face on labels alone). */
template <class X>
MGC_HD void mgc_reset_suspect_tile(X& x, const MgcLattice& L, int tile, uint32_t epoch, int list)
{
    const uint32_t st = L.status[tile];
    if (st & MGC_ST_SUSPECT) {
        x.par([&](int t) {
            L.height[(int64_t)tile * MGC_TV + t] = MGC_HINF;
            if (t < MGC_TF) mgc_shadow_reset(L, tile, t);
            if (t == 0) {
                int tz, ty, tx;
                mgc_tile_coords(L, tile, tz, ty, tx);
                bool source = (st & MGC_ST_SINK) != 0;
                for (int f = 0; f < 6; ++f) {
                    const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
                    if (nt >= 0 && (!mgc_owned(L, nt) || !(L.status[nt] & (MGC_ST_SUSPECT | MGC_ST_ALLINF)))) source = true;
                }
                L.status[tile] = (st & ~(MGC_ST_SUSPECT | MGC_ST_DIRTY | (63u << MGC_ST_DEP_SHIFT))) | MGC_ST_ALLINF;
                if (source) mgc_enqueue(x, L, list, L.rstamp, epoch, tile);
            }
        });
    }
}

/* ---------------------------------------------------------------------------------------
 * Z-slab halo exchange (multi-GPU).  A slab boundary is an ordinary tile face whose
 * neighbour lives on another GPU: per border tile the sender ships the labels of its 64 face voxels
 * (+ its outbox across that face and the outbox flag in a discharge phase); the receiver unpacks them
 * into the GHOST tile that mirrors the sender's tile.  "side" 0 = lower slab boundary (face 4, -z),
 * 1 = upper (face 5, +z).
 *
 * Messages are COMPACTED: a 1024 x 1024 cross-section has 16 384 border tiles, and in a colour phase or a
 * relabel pass a few per cent of them change.  The sender keeps a shadow of the labels the neighbour last
 * received (L.hshadow); a tile only travels when its face labels differ from the shadow or its outbox
 * holds flow.  Layout for T = gy*gx tiles per layer:
 *     int32 slot1[T]  (0 = unchanged, else 1 + record index) ; int32 count ; pad to 16 B ; record[count]
 *     kind 0 (relabel passes):   record = int32 label[64]                                   (256 B)
 *     kind 1 (discharge phases): record = double flow[64] ; int32 label[64] ; int32 flag ; pad (784 B)
 *     kind 2 (suspect closure of an incremental relabel): int32 status[T]  (DIRTY | SUSPECT bits; dense, 4 B per tile)
 * A message is BOUNDED: the header plus L.halo_max_rec record slots, moved in ONE transfer whose size both sides know without
 * asking the device (mgc_halo_exchange).  A border tile that finds the message full keeps what it has to say -- its shadow
 * stays behind, its flow stays in the outbox -- gets slot1 = 0, counts in MGC_CNT_DEFERRED and goes out with the next
 * exchange; the schedules do not take "nothing woke up" for a fixpoint, nor start a global relabel, while that counter is
 * non-zero.  `count` is the number of tiles that WANTED a slot (it can exceed halo_max_rec; readers use slot1, not count).  The
 * sender zeroes `count` before packing.  (Dense, every exchange moved 772 B per border tile: 12.6 MB per side at 1024 x 1024.)
 * ------------------------------------------------------------------------------------- */
MGC_HD int64_t mgc_halo_off_count(const MgcLattice& L) { return (int64_t)L.gy * L.gx * 4; }
MGC_HD int64_t mgc_halo_off_rec(const MgcLattice& L) { return (mgc_halo_off_count(L) + 4 + 15) / 16 * 16; }
MGC_HD int64_t mgc_halo_rec_bytes(int kind) { return kind ? MGC_TF * 8 + MGC_TF * 4 + 16 : MGC_TF * 4; }

MGC_HD int64_t mgc_halo_bytes(const MgcLattice& L, int kind)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    return kind == 2 ? T * 4 : mgc_halo_off_rec(L) + T * mgc_halo_rec_bytes(kind); /* capacity: every tile changed */
}

/* i = tile index inside the layer; packs the OWNED border tile of `side` */
template <class X>
MGC_HD void mgc_halo_pack_tile(X& x, const MgcLattice& L, int side, int kind, int i, void* buf)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    const int layer = side ? L.tz_own_hi - 1 : L.tz_own_lo;
    const int tile = layer * (int)T + i;
    const int f = side ? 5 : 4;
    if (kind == 2) {
        x.par([&](int t) {
            if (t == 0) ((int32_t*)buf)[i] = (int32_t)(L.status[tile] & (MGC_ST_DIRTY | MGC_ST_SUSPECT));
        });
        return;
    }
    int32_t* slot1 = (int32_t*)buf;
    int32_t* count = (int32_t*)((char*)buf + mgc_halo_off_count(L));
    char* recs = (char*)buf + mgc_halo_off_rec(L);
    int32_t* shadow = L.hshadow[side] + (int64_t)i * MGC_TF;
    const uint32_t fl = kind ? ((L.oflags[tile] >> f) & 1u) : 0u; /* the outbox across the border holds flow */
    const bool changed = x.any([&](int t) -> bool {
        return fl || (t < MGC_TF && L.height[(int64_t)tile * MGC_TV + mgc_face_voxel(f, t)] != shadow[t]);
    });
    x.par([&](int t) {
        if (t == 0) {
            int sl = changed ? x.atomic_add(count, 1) : -1;
            if (sl >= L.halo_max_rec) { /* the message is full: this tile keeps its news for the next exchange */
                sl = -1;
                x.atomic_add(&L.count[MGC_CNT_DEFERRED], 1);
            }
            x.S.flag[0] = sl;
            slot1[i] = sl + 1;
        }
    });
    if (x.S.flag[0] < 0) { /* (uniform: written before the barrier that ended the step above) */
        x.par([&](int) {}); /* x.S.flag[0] is reused by the next tile of this block */
        return;
    }
    x.par([&](int t) {
        char* rec = recs + (int64_t)x.S.flag[0] * mgc_halo_rec_bytes(kind);
        double* flow = (double*)rec;
        int32_t* lab = kind ? (int32_t*)(rec + MGC_TF * 8) : (int32_t*)rec;
        if (t < MGC_TF) {
            const int32_t hcur = L.height[(int64_t)tile * MGC_TV + mgc_face_voxel(f, t)];
            lab[t] = hcur;
            shadow[t] = hcur; /* what the neighbour holds from now on */
            if (kind) {
                double* slot = &L.obox[((int64_t)tile * 6 + f) * MGC_TF + t];
                flow[t] = *slot;
                *slot = 0.0; /* the flow now travels in the message */
            }
        }
        if (kind && t == MGC_TF) {
            lab[MGC_TF] = (int32_t)fl;
            if (fl) x.atomic_and(&L.oflags[tile], ~(1u << f));
        }
    });
}

/* unpacks into the GHOST tile beyond `side`; wakes the owned tile next to it.
 * kind 1: `epoch` = the phase that just ran (the woken tile runs in phase + 1);
 * kind 0: `epoch` / `list` = stamp and list of the next relabel pass. */
template <class X>
MGC_HD void mgc_halo_unpack_tile(X& x, const MgcLattice& L, int side, int kind, int i, const void* buf, uint32_t epoch, int list)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    const int ghost_layer = side ? L.tz_own_hi : L.tz_own_lo - 1;
    const int own_layer = side ? L.tz_own_hi - 1 : L.tz_own_lo;
    const int ghost = ghost_layer * (int)T + i, own = own_layer * (int)T + i;
    const int f = side ? 4 : 5; /* the face of the SENDER's tile that touches us */
    if (kind == 2) { /* the ghost mirrors the owner's flags; a ghost that turns suspect keeps the closure going */
        x.par([&](int t) {
            if (t != 0) return;
            const uint32_t old = L.status[ghost], msg = (uint32_t)((const int32_t*)buf)[i];
            L.status[ghost] = (old & ~(MGC_ST_DIRTY | MGC_ST_SUSPECT)) | msg;
            if ((msg & MGC_ST_SUSPECT) && !(old & MGC_ST_SUSPECT)) L.count[MGC_CNT_CHANGED] = 1;
        });
        return;
    }
    const int sl = ((const int32_t*)buf)[i] - 1;
    if (sl < 0) return; /* the sender's tile is as the ghost already has it */
    const char* rec = (const char*)buf + mgc_halo_off_rec(L) + (int64_t)sl * mgc_halo_rec_bytes(kind);
    const double* flow = (const double*)rec;
    const int32_t* lab = kind ? (const int32_t*)(rec + MGC_TF * 8) : (const int32_t*)rec;
    const bool lowered = x.any([&](int t) -> bool {
        bool low = false;
        if (t < MGC_TF) {
            int32_t* hp = &L.height[(int64_t)ghost * MGC_TV + mgc_face_voxel(f, t)];
            const int32_t hn = lab[t];
            low = hn < *hp;
            *hp = hn;
            if (kind) {
                const double d = flow[t];
                if (d != 0.0) L.obox[((int64_t)ghost * 6 + f) * MGC_TF + t] += d;
            }
        }
        return low;
    });
    x.par([&](int t) {
        if (t != 0) return;
        if (kind) {
            if (lab[MGC_TF]) {
                x.atomic_or(&L.oflags[ghost], 1u << f);
                /* the owned tile absorbs this in the next phase of ITS colour: the borders are exchanged once per round of the two
                 * colours, so the message carries tiles of both (after every phase it would always be epoch + 1) */
                int tz, ty, tx;
                mgc_tile_coords(L, own, tz, ty, tx);
                const uint32_t target = epoch + 1 + ((uint32_t)(mgc_tile_colour(L, tz, ty, tx) ^ (int)((epoch + 1) & 1u)) & 1u);
                mgc_enqueue(x, L, (int)(target & 3u), L.stamp, target, own);
            }
        } else if (lowered) {
            mgc_enqueue(x, L, list, L.rstamp, epoch, own);
        }
    });
}

#endif /* MGC_TILE_OPS_INL */
