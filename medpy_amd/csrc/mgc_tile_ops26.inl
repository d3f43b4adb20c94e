/*
 * mgc_tile_ops26.inl -- tile operations of the lattice max-flow solver for the FULL neighbourhood
 * (26 neighbours in 3-D, 8 in 2-D).  Same executor concept and same algorithm as mgc_tile_ops.inl
 * (region discharge with exact in-tile labels + global relabel), single source for the HIP kernels and
 * the host simulator.
 *
 * No reference counterpart: MedPy's voxel graph supports only the 2*ndim neighbourhood
 * (reference medpy/graphcut/generate.py:44-49, energy_voxel.py:583).  The 26-neighbourhood is named by
 * BASELINE.json configs 3 and 5; its oracle is the reference BK core fed the edge list of
 * oracle/energy_numpy.py:boundary_weights_offsets.
 *
 * Differences from the 6-neighbourhood ops:
 *   - direction index d = 0..25 enumerates offsets (dz,dy,dx) in lexicographic order without the centre;
 *     opposite(d) = 25 - d;
 *   - tiles are coloured with EIGHT colours (parity of tz,ty,tx), so no two tiles that share a face, an
 *     edge or a corner run in the same phase.  A voxel of an idle tile is then adjacent to at most one
 *     running tile, so a push over a tile boundary updates the neighbour tile's excess / reverse residual
 *     directly in HBM -- direction by direction, one writer per target voxel per step, no atomics, fixed
 *     floating point order -- and there is no outbox;
 *   - rmask is 32 bits per voxel (26 arc bits + sink bit 26); discharge lists rotate over 16 slots.
 */
#ifndef MGC_TILE_OPS26_INL
#define MGC_TILE_OPS26_INL

#include "mgc_tile_ops.inl"

#define MGC26_NDIR 26
#define MGC26_INL __attribute__((always_inline))
#ifndef MGC26_COUNT_STEPS
#define MGC26_COUNT_STEPS(mask) /* host simulator: statistics of the direction masks */
#endif
#define MGC26_MASK_SINK (1u << 26)
/* counter / list layout of the 26-neighbourhood solver */
#define MGC26_NLIST 16        /* discharge lists 0..15 (target phase & 15)   */
#define MGC26_RL 16           /* relabel lists 16, 17                         */
#define MGC26_CNT_ACTIVE 18
#define MGC26_CNT_DIS 19
#define MGC26_CNT_REL 20

struct MgcTileShared26 {
    int32_t hs[1000];
    double  out[4][MGC_TV];  /* push hand-off: two directions per step, double buffered */
    int32_t nbr[27];      /* neighbour tile ids, index = (dz+1)*9 + (dy+1)*3 + (dx+1); 13 = self */
    int32_t nbrflag[27];
    int32_t nbrsettled[27]; /* relabel: the neighbour tile was MGC_ST_SETTLED when the visit began */
    int32_t depflag[27];  /* relabel: some label of the tile is supported by a voxel of that neighbour tile */
    int32_t flag[2];
    int32_t bfs_above[2]; /* mgc26_tile_bfs: the last round (of either parity) in which some voxel stood above 2 */
    int32_t satflag;      /* discharge: some arc (or sink link) of the tile was saturated               */
    uint32_t dirmask[2];  /* discharge: directions along which some active voxel can push in this sweep  */
    uint32_t pushmask;    /* discharge: directions along which a voxel of the tile DID push               */
    uint32_t pflag[MGC26_NDIR][8]; /* discharge: somebody pushed along direction d into z-layer k of the tile in the step before
                                      (cleared at the top of a sweep).  A z-layer is a wave: in the sparse phases of a solve
                                      most waves of most steps have nothing to receive and nothing to push, and skip both on
                                      two words instead of reading 128 hand-off slots */
};

/* status word of a tile, full neighbourhood: bits 0..5 as in mgc_common.h (SINK, DIRTY, SUSPECT, EXCESS, ALLINF), bits 6..31 =
 * the 26 neighbour tiles that support a label of this tile (incremental global relabel, see mgc_suspect_tile) */
#define MGC26_ST_DEP_SHIFT 6
#define MGC26_ST_DEP_MASK (~0u << MGC26_ST_DEP_SHIFT)
MGC_HD int mgc26_dep_bit(int ni) { return MGC26_ST_DEP_SHIFT + (ni < 13 ? ni : ni - 1); } /* ni = 0..26 without 13 */

/* LDS of the discharge kernel: half of the 26 residuals of every voxel live here (directions 13..25), the other half in
 * registers.  All 26 in registers need 256 VGPRs (2 waves/SIMD, one workgroup per CU, and still spill); 13 + 13 fits
 * 128 VGPRs and 66 KiB of LDS, i.e. two workgroups per CU. */
#ifndef MGC26_NREG
#define MGC26_NREG 13
#endif
struct MgcTileShared26D : MgcTileShared26 {
    alignas(16) double rl[MGC26_NDIR - MGC26_NREG > 0 ? MGC26_NDIR - MGC26_NREG : 1][MGC_TV];
};

MGC_HD void mgc26_offset(int d, int& dz, int& dy, int& dx)
{
    const int c = d < 13 ? d : d + 1;
    dz = c / 9 - 1;
    dy = (c / 3) % 3 - 1;
    dx = c % 3 - 1;
}

MGC_HD int mgc26_hs_step(int d)
{
    int dz, dy, dx;
    mgc26_offset(d, dz, dy, dx);
    return dz * 100 + dy * 10 + dx;
}

/* Does this tile's status word `st` vouch for the label behind direction (dz, dy, dx) of its voxel (z, y, x)?  It does for a voxel of the
 * tile itself (a tile whose labels move is DIRTY) and for a voxel of a neighbour tile whose support bit is set (the tile turns suspect with
 * that neighbour).  A label ONE DOWN behind a residual arc into any other neighbour tile is a support nobody watches -- the support bits are
 * those of the tile's last relabel visit; a tile that settled in the first pass of a relabel (MGC_ST_SETTLED) never looked over its borders, and
 * a neighbour may have come down to "one below" later without waking anybody -- so a voxel that saturates an arc does not keep its label
 * on the strength of it: its tile is DIRTY. */
MGC_HD bool mgc26_support_watched(uint32_t st, int z, int y, int xx, int dz, int dy, int dx)
{
    const int vz = z + dz, vy = y + dy, vx = xx + dx;
    const int oz = vz < 0 ? -1 : (vz > 7 ? 1 : 0), oy = vy < 0 ? -1 : (vy > 7 ? 1 : 0), ox = vx < 0 ? -1 : (vx > 7 ? 1 : 0);
    const int ni = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1);
    return ni == 13 || ((st >> mgc26_dep_bit(ni)) & 1u) != 0;
}

MGC_HD int mgc26_colour(const MgcLattice& L, int tz, int ty, int tx) { return (((tz + L.tz_global0) & 1) << 2) | ((ty & 1) << 1) | (tx & 1); }

/* every lane: 27 neighbour tile ids into LDS, flags cleared.  Needs a barrier afterwards. */
template <class X>
MGC_HD void mgc26_load_nbrs(X& x, const MgcLattice& L, int tile, int t)
{
    if (t < 27) {
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const int nz = tz + t / 9 - 1, ny = ty + (t / 3) % 3 - 1, nx = tx + t % 3 - 1;
        x.S.nbr[t] = (nz >= 0 && nz < L.gz && ny >= 0 && ny < L.gy && nx >= 0 && nx < L.gx) ? mgc_tile_id(L, nz, ny, nx) : -1;
        x.S.nbrflag[t] = 0;
        x.S.depflag[t] = 0;
    }
    if (t < 2) x.S.flag[t] = 0;
    if (t == 2) x.S.satflag = 0;
    if (t == 3) x.S.dirmask[0] = x.S.dirmask[1] = 0;
    if (t == 4) x.S.pushmask = 0;
}

/* the 488 halo cells of the 10x10x10 label block come from up to 26 neighbour tiles */
template <class X>
MGC_HD void mgc26_load_halo(X& x, const MgcLattice& L, int t)
{
    for (int k = t; k < 1000; k += MGC_TV) {
        const int z = k / 100 - 1, y = (k / 10) % 10 - 1, xx = k % 10 - 1;
        const int oz = z < 0 ? -1 : (z > 7 ? 1 : 0), oy = y < 0 ? -1 : (y > 7 ? 1 : 0), ox = xx < 0 ? -1 : (xx > 7 ? 1 : 0);
        if (oz == 0 && oy == 0 && ox == 0) continue;
        const int nt = x.S.nbr[(oz + 1) * 9 + (oy + 1) * 3 + (ox + 1)];
        x.S.hs[k] = nt < 0 ? MGC_HINF : L.height[(int64_t)nt * MGC_TV + mgc_local(z & 7, y & 7, xx & 7)];
    }
}

/* Chaotic relaxation of the tile's own cells of x.S.hs to the fixpoint, halo frozen.  A round that changed something is followed by
 * another one only while some voxel could still come down: a voxel with a sink link stands at 1, every other one at 2 or above, so
 * a tile whose voxels are all at 1 or 2 (or cut off: mask 0) is done whatever the round changed -- with a regional term that is
 * nine tiles in ten after ONE round (every voxel has a t-link; the few without a sink link stand next to one that has), and the
 * round that would only confirm it was a third of k26_relabel_all at 512^3.  sink_tile (uniform): the tile holds a sink link at all;
 * the others never get there and skip the bookkeeping. */
template <bool RULE, class X, class MaskFn>
MGC_HD bool mgc26_tile_bfs_rounds(X& x, MaskFn mask)
{
    for (int round = 0;; ++round) {
        /* "somebody is still above 2" travels beside the vote: whoever is writes the round's number into the slot of the round's
         * parity (read behind the vote's barrier; the next writer of that slot is two barriers away).  A stale number that happens
         * to match only costs a round. */
        const bool changed = x.any([&](int t) -> bool {
            const uint32_t m = mask(t);
            if (!m) return false;
            const int me = mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7);
            int cand = (m & MGC26_MASK_SINK) ? 1 : MGC_HINF;
            /* branch-free (see mgc_tile_bfs): 26 independent LDS reads in flight, masked afterwards */
            const int own = x.S.hs[me];
#pragma unroll
            for (int d = 0; d < MGC26_NDIR; ++d) {
                const int hv = x.S.hs[me + mgc26_hs_step(d)];
                const int c = ((m >> d) & 1u) ? hv + 1 : MGC_HINF;
                cand = c < cand ? c : cand;
            }
            if (RULE && (cand < own ? cand : own) > 2) x.S.bfs_above[round & 1] = round;
            if (cand < own) {
                x.S.hs[me] = cand;
                return true;
            }
            return false;
        });
        const bool settled = RULE && x.S.bfs_above[round & 1] != round;
        if (!changed || settled) return settled;
    }
}

template <class X, class MaskFn>
MGC_HD bool mgc26_tile_bfs(X& x, MaskFn mask, bool sink_tile) /* returns: every voxel with a residual arc is known to stand at 1 or 2 */
{
    /* (two instances of the rounds: the bookkeeping of the rule is 4 % of a round, and the relabels of a volume without a regional
     * term run nearly all their rounds in tiles without a sink link) */
    return sink_tile ? mgc26_tile_bfs_rounds<true>(x, mask) : mgc26_tile_bfs_rounds<false>(x, mask);
}

/* global relabel, one tile of one pass (see mgc_relabel_tile).
 * first_pass: the pass over ALL tiles that follows fill_heights_inf (the first relabel of a solve; every relabel without the incremental
 * closure).  Every label outside the tile is INF then -- or is being written by a neighbour in the same launch; reading INF for it is
 * just as valid (labels only come down during a relabel) and the same on every run -- so the halo is not fetched, no label of the
 * tile stands on a neighbour tile (no support bits), and a tile that found a finite label wakes ALL its neighbours for the second
 * pass, where every one of them sees the others' labels and the support bits are set for good.  (Until round 6 this pass fetched
 * 488 INFs per tile and walked the 26 directions of every border voxel to find that its neighbours are INF: 1.5 of the pass's 3.6 ms
 * at 512^3.) */
template <class X>
MGC_HD void mgc26_relabel_tile(X& x, const MgcLattice& L, int tile, uint32_t next_epoch, int next_list, bool first_pass)
{
    if (!mgc_owned(L, tile) || (first_pass && !(L.status[tile] & 2u)) || (L.status[tile] & MGC_ST_SETTLED)) return;
    typename X::template Reg<uint32_t> m;
    typename X::template Reg<int> h0;
    const int64_t base = (int64_t)tile * MGC_TV;
    const bool sink_tile = (L.status[tile] & MGC_ST_SINK) != 0;
    x.par([&](int t) { mgc26_load_nbrs(x, L, tile, t); });
    x.par([&](int t) {
        /* a neighbour that is settled already (MGC_ST_SETTLED) is never woken.  Asked here, with the visit's other loads, not in front of
         * the enqueue at its end (a dependent trip per visit: +3 % on the relabels of a volume without a regional term); a neighbour that
         * settles in the meantime is sent a pass it returns from at once */
        if (t < 27) x.S.nbrsettled[t] = (sink_tile && x.S.nbr[t] >= 0) ? (int32_t)(L.status[x.S.nbr[t]] & MGC_ST_SETTLED) : 0; /* (a tile without a sink link rarely has settled neighbours: it does not ask) */
        m[t] = L.rmask32[base + t];
        h0[t] = L.height[base + t];
        x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = h0[t];
        if (first_pass) {
            for (int k = t; k < 1000; k += MGC_TV) {
                const int z = k / 100 - 1, y = (k / 10) % 10 - 1, xx = k % 10 - 1;
                if (z < 0 || z > 7 || y < 0 || y > 7 || xx < 0 || xx > 7) x.S.hs[k] = MGC_HINF;
            }
        } else {
            mgc26_load_halo(x, L, t);
        }
    });
    const bool settled = mgc26_tile_bfs(x, [&](int t) { return m[t]; }, sink_tile);
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        const int me = mgc_hs_index(z, y, xx);
        const int h = x.S.hs[me];
        const bool border = z == 0 || z == 7 || y == 0 || y == 7 || xx == 0 || xx == 7;
        const bool lowered = h < h0[t];
        if (lowered) {
            L.height[base + t] = h;
            x.S.flag[0] = 1; /* not "all INF" (any more) */
        }
        if (first_pass) return; /* (flag[0] says whether the neighbours are woken, below) */
        /* (a voxel at 1 stands on its sink link: it has no support next door, and it wakes nobody unless it came down) */
        if (!border || h >= MGC_HINF || (h <= 1 && !lowered)) return;
#pragma unroll
        for (int d = 0; d < MGC26_NDIR; ++d) {
            int dz, dy, dx;
            mgc26_offset(d, dz, dy, dx);
            const int vz = z + dz, vy = y + dy, vx = xx + dx;
            const int oz = vz < 0 ? -1 : (vz > 7 ? 1 : 0), oy = vy < 0 ? -1 : (vy > 7 ? 1 : 0), ox = vx < 0 ? -1 : (vx > 7 ? 1 : 0);
            if (!(oz || oy || ox)) continue;
            const int ni = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1), hv = x.S.hs[me + mgc26_hs_step(d)];
            /* which neighbour tiles support a label of this tile (incremental relabel) */
            if (((m[t] >> d) & 1u) && hv + 1 == h) x.S.depflag[ni] = 1;
            /* wake a neighbour tile only if one of its voxels next to this one could improve (see mgc_relabel_tile) */
            if (lowered && h + 1 < hv) x.S.nbrflag[ni] = 1;
        }
    });
    x.par([&](int t) {
        if (t < 27 && t != 13 && (first_pass ? x.S.flag[0] : x.S.nbrflag[t]) && x.S.nbr[t] >= 0 && !x.S.nbrsettled[t])
            mgc_enqueue(x, L, next_list, L.rstamp, next_epoch, x.S.nbr[t]);
        if (t == 27) {
            uint32_t dep = 0;
            for (int ni = 0; ni < 27; ++ni)
                if (ni != 13 && x.S.depflag[ni]) dep |= 1u << mgc26_dep_bit(ni);
            L.status[tile] = (L.status[tile] & ~(MGC26_ST_DEP_MASK | (x.S.flag[0] ? MGC_ST_ALLINF : 0u))) | dep | (settled ? MGC_ST_SETTLED : 0u);
        }
    });
}

/* incremental global relabel, tile-level closure (see mgc_suspect_tile): 26 supporting neighbours instead of 6 */
MGC_HD bool mgc26_suspect_tile(const MgcLattice& L, int tile)
{
    const uint32_t st = L.status[tile];
    if (st & MGC_ST_SUSPECT) return false;
    bool sus = (st & MGC_ST_DIRTY) != 0;
    if (!sus && (st & MGC26_ST_DEP_MASK)) {
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        for (int ni = 0; ni < 27 && !sus; ++ni) {
            if (ni == 13 || !((st >> mgc26_dep_bit(ni)) & 1u)) continue;
            const int nz = tz + ni / 9 - 1, ny = ty + (ni / 3) % 3 - 1, nx = tx + ni % 3 - 1;
            if (nz < 0 || nz >= L.gz || ny < 0 || ny >= L.gy || nx < 0 || nx >= L.gx) continue;
            sus = (L.status[mgc_tile_id(L, nz, ny, nx)] & MGC_ST_SUSPECT) != 0;
        }
    }
    if (sus) L.status[tile] = st | MGC_ST_SUSPECT;
    return sus;
}

/* suspect tiles: labels := INF, queued for the first relabel pass; flags retired */
template <class X>
MGC_HD void mgc26_reset_suspect_tile(X& x, const MgcLattice& L, int tile, uint32_t epoch, int list)
{
    const uint32_t st = L.status[tile];
    if (st & MGC_ST_SUSPECT) {
        x.par([&](int t) {
            L.height[(int64_t)tile * MGC_TV + t] = MGC_HINF;
            if (t == 0) {
                L.status[tile] = (st & ~(MGC_ST_SUSPECT | MGC_ST_DIRTY | MGC_ST_SETTLED | MGC26_ST_DEP_MASK)) | MGC_ST_ALLINF;
                mgc_enqueue(x, L, list, L.rstamp, epoch, tile);
            }
        });
    }
}

/* returns whether the tile was queued; the CALLER counts the active tiles (one atomic per workgroup on the counter instead of one
 * per tile: with a regional term every tile is active after the first relabel, and 262 144 atomics on one address took 3 ms) */
template <class X>
MGC_HD bool mgc26_activate_tile(X& x, const MgcLattice& L, int tile, uint32_t phase)
{
    if (!mgc_owned(L, tile)) return false; /* a ghost tile's excess / rcap only accumulate what was pushed into it */
    if (L.status[tile] & MGC_ST_ALLINF) return false; /* no label of the tile is finite: nothing can flow */
    const int64_t base = (int64_t)tile * MGC_TV;
    const bool act = x.any([&](int t) -> bool { return L.excess[base + t] > 0.0 && L.height[base + t] < MGC_HINF; });
    x.par([&](int t) {
        if (t == 0 && act) {
            int tz, ty, tx;
            mgc_tile_coords(L, tile, tz, ty, tx);
            const uint32_t target = phase + (((uint32_t)mgc26_colour(L, tz, ty, tx) - phase) & 7u);
            mgc_enqueue(x, L, (int)(target & 15u), L.stamp, target, tile);
        }
    });
    return act;
}

/* region discharge of one tile in colour phase `phase` (all 26 neighbour tiles are idle).
 * NREG of the 26 residuals of a voxel live in registers, the rest in the LDS slots x.S.rl (own lane only):
 *   NREG = 13, 512 threads, 128 VGPRs  -- two workgroups per CU, but the 27 unrolled steps spill ~440 B per lane
 *                                          (measured: 4x the tile state goes to scratch and back per discharge);
 *   NREG = 26, 256 threads x 2 voxels  -- everything in registers under the 256-VGPR budget of two waves per SIMD,
 *                                          the same two tiles per CU in flight, no residuals in LDS, no spills. */
template <int NREG = MGC26_NREG, class X>
MGC_HD void mgc26_discharge_tile(X& x, const MgcLattice& L, int tile, uint32_t phase, int max_cycles, int max_sweeps)
{
    typename X::template Reg<double> e, snk, rr[NREG];
    /* residual of lane t in direction d: register for d < NREG, LDS slot for d >= NREG; d is a compile-time
     * constant wherever this is used (unrolled loops), so the choice folds away */
    auto R = [&](int d, int t) -> double& { return d < NREG ? rr[d < NREG ? d : 0][t] : x.S.rl[d >= NREG ? d - NREG : 0][t]; };
    typename X::template Reg<int> hme;
    typename X::template Reg<uint32_t> pushed; /* directions this voxel pushed along */
    typename X::template Reg<uint32_t> stw;    /* lane 28: the tile's status word, fetched with the state (nobody else writes it during this launch) */
    const int64_t base = (int64_t)tile * MGC_TV;

    x.par([&](int t) { mgc26_load_nbrs(x, L, tile, t); });
    x.par([&](int t) {
        pushed[t] = 0;
        stw[t] = 0;
        if (t == 28) stw[t] = L.status[tile];
        e[t] = L.excess[base + t];
        snk[t] = L.sink[base + t];
#pragma unroll
        for (int d = 0; d < MGC26_NDIR; ++d) R(d, t) = L.rcap[((int64_t)tile * MGC26_NDIR + d) * MGC_TV + t];
        mgc26_load_halo(x, L, t);
        /* hand-off slots are zero except between a push and its receive (the receiver clears what it took): a voxel that pushes
         * nothing writes nothing */
        x.S.out[0][t] = 0.0; x.S.out[1][t] = 0.0; x.S.out[2][t] = 0.0; x.S.out[3][t] = 0.0;
    });
    x.mark(L, 0); /* load */

    bool active = false;
    int sweep_id = 0;
    /* max_cycles < 0: no exact in-tile labelling at all -- the stored labels are valid lower bounds (distances only grow),
     * and the local relabel at the end of every sweep raises the voxels that are stuck; -max_cycles sweep blocks */
    /* max_cycles <= -1024: the visit runs on RADIAL labels (mgc_dt_ops.inl) -- stored labels, -(max_cycles + 1024) sweep blocks, and ANY
     * saturated arc marks the tile DIRTY (whether a voxel keeps "a residual arc one label down" says nothing about its distance when the
     * labels are not distances) */
    const bool sat_dirty = max_cycles <= -1024;
    if (sat_dirty) max_cycles += 1024;
    const bool stored_labels = max_cycles < 0;
    if (stored_labels) max_cycles = -max_cycles;
    for (int cyc = 0; cyc < max_cycles; ++cyc) {
        if (stored_labels) {
            if (cyc == 0) x.par([&](int t) { x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = L.height[base + t]; });
        } else {
            x.par([&](int t) { x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = MGC_HINF; });
            mgc26_tile_bfs(x, [&](int t) {
                uint32_t m = snk[t] > 0.0 ? MGC26_MASK_SINK : 0u;
#pragma unroll
                for (int d = 0; d < MGC26_NDIR; ++d) m |= (R(d, t) > 0.0) ? (1u << d) : 0u;
                return m;
            }, (L.status[tile] & 2u) != 0);
        }
        active = x.any([&](int t) -> bool {
            hme[t] = x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)];
            return e[t] > 0.0 && hme[t] < MGC_HINF;
        });
        if (!active) break;
        x.mark(L, 1); /* labels */

        for (int sw = 0; sw < max_sweeps; ++sw, ++sweep_id) {
            const int fl = sweep_id & 1;
            /* Along which directions can a voxel that holds excess NOW push?  Only those steps run: a step costs a barrier and
             * two LDS exchanges for all 512 voxels, and late sweeps move the excess of a few voxels along a few directions.
             * (A voxel that receives excess during the sweep pushes on in the steps that run; its other directions wait for
             * the next sweep -- any order of admissible pushes is a valid discharge.) */
            x.par([&](int t) {
                if (t < MGC26_NDIR * 8) (&x.S.pflag[0][0])[t] = 0; /* (the last receive of the sweep before lies behind a barrier) */
                if (e[t] > 0.0 && hme[t] < MGC_HINF) {
                    const int me = mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7);
                    uint32_t m = 0;
                    /* (every admissible direction, not only as many as the excess held now lasts for: with the pruned mask a step
                     * costs less but flow that arrives during the sweep waits for the next visit -- 1.40 M -> 1.92 M visits and
                     * 288 -> 327 ms at 512^3, profiles/r4_rejected_pruned_step_mask26.jsonl) */
#pragma unroll
                    for (int d = 0; d < MGC26_NDIR; ++d)
                        m |= (R(d, t) > 0.0 && x.S.hs[me + mgc26_hs_step(d)] == hme[t] - 1) ? (1u << d) : 0u;
                    if (m) x.atomic_or(&x.S.dirmask[fl], m);
                }
            });
#ifdef MGC26_ALL_STEPS /* A/B: every step runs */
            const uint32_t M = 0x3ffffffu;
#else
            const uint32_t M = x.uniform(x.S.dirmask[fl]);
#endif
            MGC26_COUNT_STEPS(M);
            x.mark(L, 5); /* which directions run */
            /* 14 steps: step p pushes along the OPPOSITE directions p and 25 - p (p < 13) after receiving what step p - 1
             * pushed.  Two directions share a barrier, and their LDS round trips overlap.  Opposite directions, because
             * pushes over the tile boundary update the idle neighbour tile in place, one writer per voxel and step: two
             * voxels of this tile reach the same outside voxel only along offsets that agree in the sign of the axis it lies
             * beyond, and o and -o agree in none.  (Nor can u push to v along o while v pushes to u along -o: a push goes
             * one label down.) */
            auto receive = [&](int t, auto dc, int buf) MGC26_INL {
                constexpr int d = decltype(dc)::value;
                int dz, dy, dx;
                mgc26_offset(d, dz, dy, dx);
                if (!x.S.pflag[d][t >> 6]) return; /* nobody pushed into this z-layer along d */
                const int sz = (t >> 6) - dz, sy = ((t >> 3) & 7) - dy, sx = (t & 7) - dx; /* the voxel that pushed towards me */
                if (sz >= 0 && sz < 8 && sy >= 0 && sy < 8 && sx >= 0 && sx < 8) {
                    const int src = mgc_local(sz, sy, sx);
                    const double din = x.S.out[buf][src];
                    if (din != 0.0) {
                        e[t] += din;
                        R(25 - d, t) += din;
                        x.S.out[buf][src] = 0.0; /* (one receiver per slot and step) */
                    }
                }
            };
            auto push = [&](int t, auto dc, int buf) MGC26_INL {
                constexpr int d = decltype(dc)::value;
                const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
                int dz, dy, dx;
                mgc26_offset(d, dz, dy, dx);
                const int vz = z + dz, vy = y + dy, vx = xx + dx;
                const bool inside = vz >= 0 && vz < 8 && vy >= 0 && vy < 8 && vx >= 0 && vx < 8;
                double delta = 0.0;
                if (e[t] > 0.0 && R(d, t) > 0.0 && hme[t] < MGC_HINF) {
                    const int hv = x.S.hs[mgc_hs_index(z, y, xx) + mgc26_hs_step(d)];
                    if (hv == hme[t] - 1) {
                        delta = e[t] < R(d, t) ? e[t] : R(d, t);
                        e[t] -= delta;
                        R(d, t) -= delta;
                        x.S.flag[fl] = 1;
                        if (R(d, t) == 0.0) {
                            if (stored_labels) pushed[t] |= 0x80000000u; /* (looked at once, behind the sweeps) */
                            else x.S.satflag = 1;
                        }
                        if (!((pushed[t] >> d) & 1u)) { /* (the lane's own copy spares the atomic on every further push) */
                            pushed[t] |= 1u << d;
                            x.atomic_or(&x.S.pushmask, 1u << d);
                        }
                    }
                }
                if (delta == 0.0) return;
                if (inside) {
                    x.S.out[buf][t] = delta;
                    x.S.pflag[d][vz] = 1;
                } else {
                    /* the target voxel lives in an idle neighbour tile and nobody else writes it in this step: update its
                     * excess and reverse residual in place */
                    const int ni = ((vz < 0 ? -1 : (vz > 7 ? 1 : 0)) + 1) * 9 + ((vy < 0 ? -1 : (vy > 7 ? 1 : 0)) + 1) * 3 +
                                   ((vx < 0 ? -1 : (vx > 7 ? 1 : 0)) + 1);
                    const int nt = x.S.nbr[ni];
                    const int lv = mgc_local(vz & 7, vy & 7, vx & 7);
                    x.gadd(&L.excess[(int64_t)nt * MGC_TV + lv], delta);
                    x.gadd(&L.rcap[((int64_t)nt * MGC26_NDIR + (25 - d)) * MGC_TV + lv], delta);
                    x.gor(&L.rmask32[(int64_t)nt * MGC_TV + lv], 1u << (25 - d));
                    x.S.nbrflag[ni] = 1;
                }
            };
            auto step = [&](auto pc) {
                constexpr int p = decltype(pc)::value;            /* 0..13 */
                constexpr int q = p > 0 ? p - 1 : 0;              /* the pair received in this step */
                constexpr int pp = p < 13 ? p : 0;                /* the pair pushed in this step   */
                constexpr uint32_t recv_bits = p > 0 ? ((1u << q) | (1u << (25 - q))) : 0u;
                constexpr uint32_t push_bits = p < 13 ? ((1u << pp) | (1u << (25 - pp))) : 0u;
                if (p > 0 && !(M & (recv_bits | push_bits))) return; /* nothing to receive, nobody pushes: no barrier either */
                x.par([&](int t) {
                    if (p == 0) {
                        if (t == 0) { /* the other buffers are free: their last readers are behind the barrier of the mask pass */
                            x.S.flag[fl ^ 1] = 0;
                            x.S.dirmask[fl ^ 1] = 0;
                        }
                        if (e[t] > 0.0 && snk[t] > 0.0) {
                            const double delta = e[t] < snk[t] ? e[t] : snk[t];
                            e[t] -= delta;
                            snk[t] -= delta;
                            x.S.flag[fl] = 1;
                            if (snk[t] == 0.0) {
                                if (stored_labels) pushed[t] |= 0x80000000u;
                                else x.S.satflag = 1;
                            }
                        }
                    } else {
                        if ((M >> q) & 1u) receive(t, std::integral_constant<int, q>{}, (q & 1) * 2);
                        if ((M >> (25 - q)) & 1u) receive(t, std::integral_constant<int, 25 - q>{}, (q & 1) * 2 + 1);
                    }
                    if (p < 13) {
                        if ((M >> pp) & 1u) push(t, std::integral_constant<int, pp>{}, (pp & 1) * 2);
                        if ((M >> (25 - pp)) & 1u) push(t, std::integral_constant<int, 25 - pp>{}, (pp & 1) * 2 + 1);
                    }
                });
            };
#define MGC26_STEP(n) step(std::integral_constant<int, n>{});
            MGC26_STEP(0) MGC26_STEP(1) MGC26_STEP(2) MGC26_STEP(3) MGC26_STEP(4) MGC26_STEP(5) MGC26_STEP(6) MGC26_STEP(7) MGC26_STEP(8)
            MGC26_STEP(9) MGC26_STEP(10) MGC26_STEP(11) MGC26_STEP(12) MGC26_STEP(13)
#undef MGC26_STEP
            x.mark(L, 6); /* the steps */
            /* local relabel of stuck active voxels (see mgc_discharge_tile): labels stay valid lower bounds */
            x.par([&](int t) {
                if (e[t] > 0.0 && hme[t] < MGC_HINF) {
                    const int me = mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7);
                    int cand = snk[t] > 0.0 ? 1 : MGC_HINF;
#pragma unroll
                    for (int d = 0; d < MGC26_NDIR; ++d)
                        if (R(d, t) > 0.0) {
                            const int hv = x.S.hs[me + mgc26_hs_step(d)];
                            cand = (hv < MGC_HINF && hv + 1 < cand) ? hv + 1 : cand;
                        }
                    if (cand > hme[t]) {
                        hme[t] = cand;
                        x.S.hs[me] = cand;
                        x.S.satflag = 1; /* a label rose: whoever stood on it has to be looked at */
                        if (cand < MGC_HINF) x.S.flag[fl] = 1;
                    }
                }
            });
            x.mark(L, 2); /* one sweep (what is left of it: the local relabel) */
            if (!x.S.flag[fl]) break;
        }
    }
    if (active) active = x.any([&](int t) -> bool { return e[t] > 0.0 && hme[t] < MGC_HINF; });
    /* DIRTY (the next global relabel recomputes the tile and whoever depends on it) iff a label rose, or a voxel that saturated an
     * arc has no residual arc one label down left that the tile watches (mgc26_support_watched).  A voxel that keeps one of its supports
     * keeps its distance: with 26 neighbours most do, and the tiles a small flow merely passes through stay clean.  (With an exact in-tile labelling per discharge the
     * stored labels are not what the pushes followed: any saturation counts there.) */
    if (stored_labels) {
        x.par([&](int t) {
            if ((pushed[t] >> 31) && hme[t] < MGC_HINF) {
                const int me = mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7);
                bool kept = false;
                if (!sat_dirty) {
                    kept = snk[t] > 0.0; /* (a label of 1 stands on the sink link) */
                    const uint32_t stk = L.status[tile]; /* (nobody else writes it during this launch) */
#pragma unroll
                    for (int d = 0; d < MGC26_NDIR; ++d) {
                        int dz, dy, dx;
                        mgc26_offset(d, dz, dy, dx);
                        kept = kept || (R(d, t) > 0.0 && x.S.hs[me + mgc26_hs_step(d)] == hme[t] - 1 && mgc26_support_watched(stk, t >> 6, (t >> 3) & 7, t & 7, dz, dy, dx));
                    }
                }
                if (!kept) x.S.satflag = 1;
            }
        });
    }
    const bool has_sink = x.any([&](int t) -> bool { return snk[t] > 0.0; });

    /* a residual plane changed iff somebody pushed along it or along its opposite (the receiver's reverse arc): the others --
     * typically 10 of 26 -- are not written back */
    /* wake-ups first, the write-back behind them: a returning atomic issued after the ~40 stores would wait for them to retire */
    x.par([&](int t) {
        /* wake-ups: lanes 0..26 the neighbours that received something, lane 13 (the centre) the tile itself when its budget ran
         * out with work left -- ONE claim and ONE position draw for all of them (two dependent trips; as separate steps the
         * tile's own wake-up waited for the neighbours' and the status word for both) */
        int wake = -1;
        uint32_t target = 0;
        if (t < 27 && t != 13 && x.S.nbrflag[t] && x.S.nbr[t] >= 0) {
            int tz, ty, tx;
            mgc_tile_coords(L, tile, tz, ty, tx);
            const int mine = mgc26_colour(L, tz, ty, tx);
            const int theirs = mgc26_colour(L, tz + t / 9 - 1, ty + (t / 3) % 3 - 1, tx + t % 3 - 1);
            target = phase + (uint32_t)((theirs - mine) & 7);
            wake = x.S.nbr[t];
            if (!mgc_owned(L, wake)) x.atomic_or(&L.oflags[wake], 1u); /* ghost: the halo exchange ships what it received */
        }
        if (t == 13 && active) { wake = tile; target = phase + 8; }
        if (wake >= 0) mgc_enqueue(x, L, (int)(target & 15u), L.stamp, target, wake);
        /* DIRTY only if a residual arc disappeared: otherwise no distance in the tile (or through it) can have changed */
        if (t == 28) L.status[tile] = (stw[t] & ~MGC_ST_SINK) | (has_sink ? MGC_ST_SINK : 0u) | (x.S.satflag ? MGC_ST_DIRTY : 0u);
    });
    const uint32_t PM = x.uniform(x.S.pushmask);
    x.par([&](int t) {
        MGC_STORE_STREAM(&L.excess[base + t], e[t]);
        MGC_STORE_STREAM(&L.sink[base + t], snk[t]);
        uint32_t m = snk[t] > 0.0 ? MGC26_MASK_SINK : 0u;
#pragma unroll
        for (int d = 0; d < MGC26_NDIR; ++d) {
            if (((PM >> d) | (PM >> (25 - d))) & 1u) MGC_STORE_STREAM(&L.rcap[((int64_t)tile * MGC26_NDIR + d) * MGC_TV + t], R(d, t));
            m |= (R(d, t) > 0.0) ? (1u << d) : 0u;
        }
        MGC_STORE_STREAM(&L.rmask32[base + t], m);
        L.height[base + t] = x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)];
    });
    x.mark(L, 3); /* store */
}

/* ---------------------------------------------------------------------------------------
 * Z-slab halo exchange, 26-neighbourhood.  There are no outboxes: a push over the slab border lands in the GHOST
 * tile's excess and reverse residuals, which are kept at zero otherwise (mgc_build) and therefore ACCUMULATE what the
 * neighbour slab has to add to its own border tile.  Per border tile the message carries the labels of the OWNED
 * border layer and, only for ghost tiles that received something, a 5 KiB record of the ghost's border voxel layer:
 * excess and the 9 residuals that point back over the border (dz = -1: directions 0..8 for the upper ghost,
 * dz = +1: directions 17..25 for the lower ghost).  Records are COMPACTED (slot = order of arrival), so the
 * transport moves the fixed header plus `count` records (mgc_halo_exchange) instead of 5 KiB for every tile:
 *     int32 label[T][64] ; int32 slot1[T] (0 = nothing, else 1 + record index) ; int32 count ; pad to 16 B ;
 *     double record[count][10][64]                                     (kind 1: discharge phases)
 *     int32 label[T][64]                                              (kind 0: relabel passes)
 * The sender zeroes `count` before packing.
 * ------------------------------------------------------------------------------------- */
#define MGC26_REC (10 * MGC_TF) /* doubles per record */

MGC_HD int64_t mgc26_halo_off_slot(const MgcLattice& L) { return (int64_t)L.gy * L.gx * MGC_TF * 4; }
MGC_HD int64_t mgc26_halo_off_count(const MgcLattice& L) { return mgc26_halo_off_slot(L) + (int64_t)L.gy * L.gx * 4; }
MGC_HD int64_t mgc26_halo_off_rec(const MgcLattice& L) { return (mgc26_halo_off_count(L) + 4 + 15) / 16 * 16; }

MGC_HD int64_t mgc26_halo_bytes(const MgcLattice& L, int kind)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    if (kind == 2) return T * 4; /* int32 status[T]: DIRTY | SUSPECT of the owned border tiles (suspect closure, as mgc_halo_bytes) */
    return kind == 1 ? mgc26_halo_off_rec(L) + T * MGC26_REC * 8 : T * MGC_TF * 4;
}

template <class X>
MGC_HD void mgc26_halo_pack_tile(X& x, const MgcLattice& L, int side, int kind, int i, void* buf)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    const int own = (side ? L.tz_own_hi - 1 : L.tz_own_lo) * (int)T + i;
    const int ghost = (side ? L.tz_own_hi : L.tz_own_lo - 1) * (int)T + i;
    const int f_own = side ? 5 : 4, f_ghost = side ? 4 : 5, dbase = side ? 0 : 17;
    if (kind == 2) {
        x.par([&](int t) {
            if (t == 0) ((int32_t*)buf)[i] = (int32_t)(L.status[own] & (MGC_ST_DIRTY | MGC_ST_SUSPECT));
        });
        return;
    }
    int32_t* lab = (int32_t*)buf;
    int32_t* slot1 = (int32_t*)((char*)buf + mgc26_halo_off_slot(L));
    int32_t* count = (int32_t*)((char*)buf + mgc26_halo_off_count(L));
    double* rec = (double*)((char*)buf + mgc26_halo_off_rec(L));
    x.par([&](int t) {
        if (t < MGC_TF) lab[(int64_t)i * MGC_TF + t] = L.height[(int64_t)own * MGC_TV + mgc_face_voxel(f_own, t)];
        if (kind && t == 0) {
            const uint32_t fl = L.oflags[ghost] & 1u;
            int sl = fl ? x.atomic_add(count, 1) : -1;
            if (sl >= L.halo_max_rec) { /* the message is full: what the ghost tile collected stays there until the next exchange */
                sl = -1;
                x.atomic_add(&L.count[MGC_CNT_DEFERRED], 1);
            }
            x.S.flag[0] = sl;
            slot1[i] = sl + 1;
            if (sl >= 0) L.oflags[ghost] = 0;
        }
    });
    if (!kind) return;
    x.par([&](int t) {
        const int sl = x.S.flag[0];
        if (sl < 0) return;
        for (int k = t; k < MGC26_REC; k += MGC_TV) {
            const int q = k >> 6, v = mgc_face_voxel(f_ghost, k & 63);
            double* src = q == 0 ? &L.excess[(int64_t)ghost * MGC_TV + v] : &L.rcap[((int64_t)ghost * MGC26_NDIR + dbase + q - 1) * MGC_TV + v];
            rec[(int64_t)sl * MGC26_REC + k] = *src;
            *src = 0.0; /* the flow now travels in the message */
        }
    });
    x.par([&](int) {}); /* x.S.flag[0] is reused by the next tile of this block */
}

template <class X>
MGC_HD void mgc26_halo_unpack_tile(X& x, const MgcLattice& L, int side, int kind, int i, const void* buf, uint32_t epoch, int list)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    const int own_layer = side ? L.tz_own_hi - 1 : L.tz_own_lo;
    const int own = own_layer * (int)T + i;
    const int ghost = (side ? L.tz_own_hi : L.tz_own_lo - 1) * (int)T + i;
    const int f_own = side ? 5 : 4, f_ghost = side ? 4 : 5, dbase = side ? 17 : 0; /* the sender packed ITS other side */
    if (kind == 2) { /* the ghost mirrors the owner's flags; a ghost that turns suspect keeps the closure going */
        x.par([&](int t) {
            if (t != 0) return;
            const uint32_t msg = (uint32_t)((const int32_t*)buf)[i], old = L.status[ghost];
            L.status[ghost] = (old & ~(MGC_ST_DIRTY | MGC_ST_SUSPECT)) | msg;
            if ((msg & MGC_ST_SUSPECT) && !(old & MGC_ST_SUSPECT)) L.count[MGC_CNT_CHANGED] = 1;
        });
        return;
    }
    const int32_t* lab = (const int32_t*)buf;
    const int32_t* slot1 = (const int32_t*)((const char*)buf + mgc26_halo_off_slot(L));
    const double* rec = (const double*)((const char*)buf + mgc26_halo_off_rec(L));
    const int sl = kind ? slot1[i] - 1 : -1;
    const bool lowered = x.any([&](int t) -> bool {
        bool low = false;
        if (t < MGC_TF) {
            int32_t* hp = &L.height[(int64_t)ghost * MGC_TV + mgc_face_voxel(f_ghost, t)];
            const int32_t hn = lab[(int64_t)i * MGC_TF + t];
            low = hn < *hp;
            *hp = hn;
            if (sl >= 0) {
                const double* r = rec + (int64_t)sl * MGC26_REC;
                const int v = mgc_face_voxel(f_own, t);
                const double de = r[t];
                if (de != 0.0) L.excess[(int64_t)own * MGC_TV + v] += de;
                uint32_t bits = 0;
                for (int q = 0; q < 9; ++q) {
                    const double d = r[(1 + q) * MGC_TF + t];
                    if (d != 0.0) {
                        L.rcap[((int64_t)own * MGC26_NDIR + dbase + q) * MGC_TV + v] += d;
                        bits |= 1u << (dbase + q);
                    }
                }
                if (bits) L.rmask32[(int64_t)own * MGC_TV + v] |= bits;
            }
        }
        return low;
    });
    x.par([&](int t) {
        if (kind) {
            if (t == 0 && sl >= 0) { /* the tile runs in the next phase of its colour */
                const int ty = i / L.gx, tx = i % L.gx;
                const uint32_t target = epoch + 1 + (((uint32_t)mgc26_colour(L, own_layer, ty, tx) - (epoch + 1)) & 7u);
                mgc_enqueue(x, L, (int)(target & 15u), L.stamp, target, own);
            }
        } else if (lowered && t < 9) { /* every owned tile that touches the ghost: face, edge and corner neighbours */
            const int ty = i / L.gx + t / 3 - 1, tx = i % L.gx + t % 3 - 1;
            if (ty >= 0 && ty < L.gy && tx >= 0 && tx < L.gx) mgc_enqueue(x, L, list, L.rstamp, epoch, mgc_tile_id(L, own_layer, ty, tx));
        }
    });
}

/* neighbourhood-agnostic entry points used by the kernels / the host simulator */
MGC_HD int64_t mgc_halo_bytes_nd(const MgcLattice& L, int kind) { return L.ndir == MGC26_NDIR ? mgc26_halo_bytes(L, kind) : mgc_halo_bytes(L, kind); }
/* compacted messages (header, then `count` records): which kinds, where the count sits, how long a record is */
MGC_HD bool mgc_halo_compact_nd(const MgcLattice& L, int kind) { return L.ndir == MGC26_NDIR ? kind == 1 : kind != 2; }
MGC_HD int64_t mgc_halo_off_count_nd(const MgcLattice& L) { return L.ndir == MGC26_NDIR ? mgc26_halo_off_count(L) : mgc_halo_off_count(L); }
MGC_HD int64_t mgc_halo_off_rec_nd(const MgcLattice& L) { return L.ndir == MGC26_NDIR ? mgc26_halo_off_rec(L) : mgc_halo_off_rec(L); }
MGC_HD int64_t mgc_halo_rec_bytes_nd(const MgcLattice& L, int kind) { return L.ndir == MGC26_NDIR ? (int64_t)MGC26_REC * 8 : mgc_halo_rec_bytes(kind); }

template <class X>
MGC_HD void mgc_halo_pack_nd(X& x, const MgcLattice& L, int side, int kind, int i, void* buf)
{
    if (L.ndir == MGC26_NDIR) mgc26_halo_pack_tile(x, L, side, kind, i, buf);
    else mgc_halo_pack_tile(x, L, side, kind, i, buf);
}

template <class X>
MGC_HD void mgc_halo_unpack_nd(X& x, const MgcLattice& L, int side, int kind, int i, const void* buf, uint32_t epoch, int list)
{
    if (L.ndir == MGC26_NDIR) mgc26_halo_unpack_tile(x, L, side, kind, i, buf, epoch, list);
    else mgc_halo_unpack_tile(x, L, side, kind, i, buf, epoch, list);
}

#endif /* MGC_TILE_OPS26_INL */
