/*
 * mgc_brick_ops.inl -- global-relabel pass over BRICKS of 2 x 2 x 2 tiles (16^3 voxels) instead of single tiles.
 *
 * A relabel pass is latency, not work: a tile visit is four or five dependent trips to HBM (list entry, state, claim,
 * position) around ~3 us of relaxation, and the label wave of a global relabel advances ONE tile per pass -- ~25 passes of
 * ~30 us per incremental relabel at 512^3, 8 ms per solve (profiles/README.md, round 3).  A workgroup that relaxes a brick
 * to its local fixpoint moves the wave two tiles per pass for the same number of trips: half the passes, each a little
 * longer.  Same fixpoint: labels only go down during a relabel and the relaxation is monotone, so the order in which
 * voxels are visited does not matter (mgc_relabel_tile's argument).
 *
 * Used for the INCREMENTAL global relabels of the 6-neighbourhood solver on a single handle (slabs keep the tile passes:
 * ownership is decided per tile layer there).  The work lists of these relabels hold brick ids, and so do the
 * de-duplication stamps (rstamp).  Per-tile results are what mgc_relabel_tile leaves: labels, the faces that support
 * them (status bits 8..13, incremental relabel) and the ALLINF flag.
 *
 * Executor X: as in mgc_tile_ops.inl, with 4096 lanes -- lane t = tile-in-brick (t >> 9: bit 2 = +z, bit 1 = +y, bit 0 = +x)
 * and voxel-in-tile (t & 511).  X::Reg<T> holds one value per lane; x.S is MgcBrickShared.
 */
#ifndef MGC_BRICK_OPS_INL
#define MGC_BRICK_OPS_INL

#include "mgc_tile_ops.inl"

#define MGC_BV 4096 /* voxels per brick */

struct alignas(16) MgcBrickShared {
    int32_t hs[18 * 18 * 18]; /* labels of the brick plus a one-voxel halo */
    int32_t tile[8];          /* the eight tiles (-1: beyond the grid)      */
    int32_t nbrick[8];        /* the six neighbour bricks (-1: none)        */
    int32_t wake[8];          /* face f saw a label drop the neighbour brick could use */
    int32_t dep[48];          /* [tile][face]: the face supports a label of the tile   */
    int32_t low[8];           /* some label of the tile came down                       */
    int32_t flag[2];
};

MGC_HD int mgcb_hs(int Z, int Y, int X) { return ((Z + 1) * 18 + (Y + 1)) * 18 + (X + 1); } /* brick coordinates -1 .. 16 */
MGC_HD int mgcb_step(int d) { return d == 0 ? -1 : d == 1 ? 1 : d == 2 ? -18 : d == 3 ? 18 : d == 4 ? -324 : 324; }

MGC_HD int mgc_brick_count(const MgcLattice& L) { return ((L.gz + 1) / 2) * ((L.gy + 1) / 2) * ((L.gx + 1) / 2); }
MGC_HD int mgc_brick_of_tile(const MgcLattice& L, int tile)
{
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    return ((tz >> 1) * ((L.gy + 1) / 2) + (ty >> 1)) * ((L.gx + 1) / 2) + (tx >> 1);
}

/* queue a brick for the next relabel pass (the stamps of these lists are indexed by brick) */
template <class X>
MGC_HD void mgc_enqueue_brick(X& x, const MgcLattice& L, int listid, uint32_t epoch, int brick)
{
    if (x.atomic_exch(&L.rstamp[brick], epoch) != epoch) {
        const int sh = x.shard(L);
        const int pos = x.atomic_add(mgc_counter(L, listid, sh), 1);
        L.list[listid][(int64_t)sh * L.shard_cap + pos] = brick;
    }
}

template <class X>
MGC_HD void mgc_relabel_brick(X& x, const MgcLattice& L, int brick, uint32_t next_epoch, int next_list)
{
    typename X::template Reg<int> m, h0;
    typename X::template Reg<int> stw; /* lane 8 + k: status word of tile k */
    const int bxn = (L.gx + 1) / 2, byn = (L.gy + 1) / 2;
    const int bx = brick % bxn, by = (brick / bxn) % byn, bz = brick / (bxn * byn);
    x.par([&](int t) { /* which tiles, which neighbour bricks */
        if (t < 8) {
            const int tz = bz * 2 + (t >> 2), ty = by * 2 + ((t >> 1) & 1), tx = bx * 2 + (t & 1);
            x.S.tile[t] = (tz < L.gz && ty < L.gy && tx < L.gx) ? mgc_tile_id(L, tz, ty, tx) : -1;
            x.S.low[t] = 0;
            x.S.wake[t] = 0;
            const int nz = bz + (t == 4 ? -1 : (t == 5 ? 1 : 0)), ny = by + (t == 2 ? -1 : (t == 3 ? 1 : 0)), nx = bx + (t == 0 ? -1 : (t == 1 ? 1 : 0));
            x.S.nbrick[t] = (t < 6 && nz >= 0 && nz < (L.gz + 1) / 2 && ny >= 0 && ny < byn && nx >= 0 && nx < bxn) ? (nz * byn + ny) * bxn + nx : -1;
        }
        if (t < 48) x.S.dep[t] = 0;
        if (t < 2) x.S.flag[t] = 0;
    });
    x.par([&](int t) { /* one trip to HBM: masks, labels, label halo, status words */
        const int k = t >> 9, loc = t & 511;
        const int tile = x.S.tile[k];
        const int Z = (k >> 2) * 8 + (loc >> 6), Y = ((k >> 1) & 1) * 8 + ((loc >> 3) & 7), XX = (k & 1) * 8 + (loc & 7);
        m[t] = 0;
        h0[t] = MGC_HINF;
        if (tile >= 0) {
            m[t] = L.rmask[(int64_t)tile * MGC_TV + loc];
            h0[t] = L.height[(int64_t)tile * MGC_TV + loc];
        }
        x.S.hs[mgcb_hs(Z, Y, XX)] = h0[t];
        if (t < 6 * 256) { /* halo: the voxel layer beyond brick face f */
            const int f = t >> 8, u = (t >> 4) & 15, v = t & 15;
            const int a = f >> 1, w = (f & 1) ? 16 : -1;
            const int hz = a == 2 ? w : u, hy = a == 1 ? w : (a == 2 ? u : v), hx = a == 0 ? w : v; /* a = 0: (u, v) = (z, y); 1: (z, x); 2: (y, x) */
            const int64_t gz = (int64_t)bz * 16 + hz, gy = (int64_t)by * 16 + hy, gx = (int64_t)bx * 16 + hx;
            int32_t hv = MGC_HINF;
            if (gz >= 0 && gy >= 0 && gx >= 0 && (gz >> 3) < L.gz && (gy >> 3) < L.gy && (gx >> 3) < L.gx)
                hv = L.height[(int64_t)mgc_tile_id(L, (int)(gz >> 3), (int)(gy >> 3), (int)(gx >> 3)) * MGC_TV + mgc_local((int)(gz & 7), (int)(gy & 7), (int)(gx & 7))];
            x.S.hs[mgcb_hs(hz, hy, hx)] = hv;
        }
        stw[t] = 0;
        if (t >= 8 && t < 16 && x.S.tile[t - 8] >= 0) stw[t] = (int)L.status[x.S.tile[t - 8]];
    });
    /* relax from the CURRENT labels to the fixpoint given the frozen halo (mgc_tile_bfs on 16^3 voxels) */
    for (;;) {
        auto relax = [&](int t) -> bool {
            const int mm = m[t];
            if (!mm) return false;
            const int k = t >> 9, loc = t & 511;
            const int me = mgcb_hs((k >> 2) * 8 + (loc >> 6), ((k >> 1) & 1) * 8 + ((loc >> 3) & 7), (k & 1) * 8 + (loc & 7));
            int cand = (mm & MGC_MASK_SINK) ? 1 : MGC_HINF;
            int hv[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) hv[d] = x.S.hs[me + mgcb_step(d)];
            const int own = x.S.hs[me];
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const int c = ((mm >> d) & 1) ? hv[d] + 1 : MGC_HINF;
                cand = c < cand ? c : cand;
            }
            if (cand < own) {
                x.S.hs[me] = cand;
                return true;
            }
            return false;
        };
        x.par([&](int t) { (void)relax(t); });
        x.par([&](int t) { (void)relax(t); });
        if (!x.any(relax)) break;
    }
    x.par([&](int t) { /* LDS only: which tile faces support a label, which tiles came down, which brick faces should wake the neighbour */
        const int k = t >> 9, loc = t & 511;
        const int z = loc >> 6, y = (loc >> 3) & 7, xx = loc & 7;
        const int Z = (k >> 2) * 8 + z, Y = ((k >> 1) & 1) * 8 + y, XX = (k & 1) * 8 + xx;
        const int me = mgcb_hs(Z, Y, XX);
        const int hm = x.S.hs[me];
        if (hm < MGC_HINF) {
            for (int d = 0; d < 6; ++d)
                if (((m[t] >> d) & 1) && !mgc_inside(d, z, y, xx) && x.S.hs[me + mgcb_step(d)] + 1 == hm) x.S.dep[k * 6 + d] = 1;
        }
        if (hm < h0[t]) {
            x.S.low[k] = 1;
            /* wake the neighbour brick across a face only if its adjacent voxel could improve (labels only go down during a
             * relabel: the halo value is an upper bound of the neighbour's current label) */
            for (int d = 0; d < 6; ++d) {
                const int c = (d >> 1) == 0 ? XX : ((d >> 1) == 1 ? Y : Z);
                if (((d & 1) ? c == 15 : c == 0) && hm + 1 < x.S.hs[me + mgcb_step(d)]) x.S.wake[d] = 1;
            }
        }
    });
    x.par([&](int t) { /* one block of global traffic: wake-ups (their claims first), labels, status words */
        const int k = t >> 9, loc = t & 511;
        int wake = -1;
        bool won = false;
        if (t < 6 && x.S.wake[t] && x.S.nbrick[t] >= 0) {
            wake = x.S.nbrick[t];
            won = x.atomic_exch(&L.rstamp[wake], next_epoch) != next_epoch;
        }
        const int tile = x.S.tile[k];
        const int h = x.S.hs[mgcb_hs((k >> 2) * 8 + (loc >> 6), ((k >> 1) & 1) * 8 + ((loc >> 3) & 7), (k & 1) * 8 + (loc & 7))];
        if (tile >= 0 && h < h0[t]) L.height[(int64_t)tile * MGC_TV + loc] = h;
        if (t >= 8 && t < 16 && x.S.tile[t - 8] >= 0) {
            const int kk = t - 8;
            uint32_t dep = 0;
            for (int f = 0; f < 6; ++f) dep |= x.S.dep[kk * 6 + f] ? (1u << f) : 0u;
            L.status[x.S.tile[kk]] = ((uint32_t)stw[t] & ~((63u << MGC_ST_DEP_SHIFT) | (x.S.low[kk] ? MGC_ST_ALLINF : 0u))) | (dep << MGC_ST_DEP_SHIFT);
        }
        if (won) {
            const int sh = x.shard(L);
            const int pos = x.atomic_add(mgc_counter(L, next_list, sh), 1);
            L.list[next_list][(int64_t)sh * L.shard_cap + pos] = wake;
        }
    });
}

#endif /* MGC_BRICK_OPS_INL */
