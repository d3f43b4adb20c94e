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
 *   x.S                  MgcTileShared& (LDS)
 *   x.atomic_add/or/and/exch   device-scope atomics on global words
 */
#ifndef MGC_TILE_OPS_INL
#define MGC_TILE_OPS_INL

#include <type_traits>

#include "mgc_common.h"

struct MgcTileShared {
    int32_t hs[1000];          /* 10x10x10 distance labels: the tile plus a one-voxel halo */
    double  out[2][MGC_TV];    /* per-direction push hand-off, double buffered             */
    int32_t nbr[8];            /* neighbour tile ids                                       */
    int32_t faceflag[8];
    int32_t flag[2];
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
        const int pos = x.atomic_add(&L.count[listid], 1);
        L.list[listid][pos] = tile;
    }
}

/* every lane: neighbour tile ids into LDS, face flags cleared.  Needs a barrier afterwards. */
template <class X>
MGC_HD void mgc_load_nbrs(X& x, const MgcLattice& L, int tile, int t)
{
    if (t < 6) {
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        x.S.nbr[t] = mgc_tile_nbr(L, tz, ty, tx, t);
        x.S.faceflag[t] = 0;
    }
    if (t < 2) x.S.flag[t] = 0;
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

/* one lane absorbs what the neighbour tiles pushed across its (up to three) faces:
 * e += delta, reverse residual += delta (the receiving half of a push, maxflow.cpp:268-271 analogue) */
template <class X, class RegD>
MGC_HD bool mgc_absorb_lane(X& x, const MgcLattice& L, int t, RegD& e, RegD (&r)[6])
{
    const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
    bool got = false;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        if (mgc_inside(d, z, y, xx)) continue;
        const int nt = x.S.nbr[d];
        if (nt < 0) continue;
        if (!((L.oflags[nt] >> (d ^ 1)) & 1u)) continue;
        const int k = mgc_face_index(d >> 1, z, y, xx);
        double* slot = &L.obox[((int64_t)nt * 6 + (d ^ 1)) * MGC_TF + k];
        const double delta = *slot;
        if (delta != 0.0) {
            e[t] += delta;
            r[d][t] += delta;
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
    if (t < 6) {
        const int nt = x.S.nbr[t];
        if (nt >= 0 && ((L.oflags[nt] >> (t ^ 1)) & 1u)) x.atomic_and(&L.oflags[nt], ~(1u << (t ^ 1)));
    }
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
        const bool changed = x.any([&](int t) -> bool {
            const int m = mask(t);
            if (!m) return false;
            const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
            const int me = mgc_hs_index(z, y, xx);
            int cand = (m & MGC_MASK_SINK) ? 1 : MGC_HINF;
#pragma unroll
            for (int d = 0; d < 6; ++d)
                if ((m >> d) & 1) {
                    const int hv = x.S.hs[me + mgc_hs_step(d)] + 1;
                    cand = hv < cand ? hv : cand;
                }
            if (cand < x.S.hs[me]) {
                x.S.hs[me] = cand;
                return true;
            }
            return false;
        });
        if (!changed) break;
    }
}

/* ---------------------------------------------------------------------------------------
 * Global relabel, one tile of one pass: recompute the tile's labels from its residual mask
 * with the current halo; wake the neighbours across every face whose labels went down.
 * Passes repeat (driver) until no tile changes: exact distances to the sink, MGC_HINF for
 * voxels that cannot reach it -- the set the reference reads out with what_segment().
 * ------------------------------------------------------------------------------------- */
template <class X>
MGC_HD void mgc_relabel_tile(X& x, const MgcLattice& L, int tile, uint32_t next_epoch, int next_list, bool first_pass)
{
    /* first pass of a global relabel: every label is INF, so only tiles holding a sink arc can seed anything */
    if (first_pass && (!(L.status[tile] & 2u) || !mgc_owned(L, tile))) return;
    typename X::template Reg<int> m, h0;
    const int64_t base = (int64_t)tile * MGC_TV;
    x.par([&](int t) { mgc_load_nbrs(x, L, tile, t); });
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        m[t] = L.rmask[base + t];
        h0[t] = L.height[base + t];
        x.S.hs[mgc_hs_index(z, y, xx)] = h0[t];
        mgc_load_halo(x, L, t);
    });
    mgc_tile_bfs(x, [&](int t) { return m[t]; });
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        const int h = x.S.hs[mgc_hs_index(z, y, xx)];
        if (h < h0[t]) {
            L.height[base + t] = h;
            if (xx == 0) x.S.faceflag[0] = 1;
            if (xx == MGC_T - 1) x.S.faceflag[1] = 1;
            if (y == 0) x.S.faceflag[2] = 1;
            if (y == MGC_T - 1) x.S.faceflag[3] = 1;
            if (z == 0) x.S.faceflag[4] = 1;
            if (z == MGC_T - 1) x.S.faceflag[5] = 1;
        }
    });
    x.par([&](int t) {
        if (t < 6 && x.S.faceflag[t] && x.S.nbr[t] >= 0) mgc_enqueue(x, L, next_list, L.rstamp, next_epoch, x.S.nbr[t]);
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
    const bool pending = x.any([&](int t) -> bool {
        if (t >= 6) return false;
        const int nt = x.S.nbr[t];
        return nt >= 0 && ((L.oflags[nt] >> (t ^ 1)) & 1u);
    });
    if (!pending) return;
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        if (xx != 0 && xx != 7 && y != 0 && y != 7 && z != 0 && z != 7) return;
        e[t] = L.excess[base + t];
#pragma unroll
        for (int d = 0; d < 6; ++d) r[d][t] = L.rcap[((int64_t)tile * 6 + d) * MGC_TV + t];
        if (mgc_absorb_lane(x, L, t, e, r)) {
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
    x.par([&](int t) { mgc_clear_inbox_flags(x, L, t); });
}

/* ---------------------------------------------------------------------------------------
 * After a global relabel: does the tile hold excess that can still reach the sink?
 * ------------------------------------------------------------------------------------- */
template <class X>
MGC_HD void mgc_activate_tile(X& x, const MgcLattice& L, int tile, uint32_t phase)
{
    if (!mgc_owned(L, tile)) return;
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
    typename X::template Reg<double> e, snk, r[6], ob[3];
    typename X::template Reg<int> hme;
    const int64_t base = (int64_t)tile * MGC_TV;

    x.par([&](int t) { mgc_load_nbrs(x, L, tile, t); });
    x.par([&](int t) {
        e[t] = L.excess[base + t];
        snk[t] = L.sink[base + t];
#pragma unroll
        for (int d = 0; d < 6; ++d) r[d][t] = L.rcap[((int64_t)tile * 6 + d) * MGC_TV + t];
        ob[0][t] = ob[1][t] = ob[2][t] = 0.0;
        mgc_load_halo(x, L, t);
        mgc_absorb_lane(x, L, t, e, r);
    });
    x.par([&](int t) { mgc_clear_inbox_flags(x, L, t); });

    bool active = false;
    int sweep_id = 0;
    for (int cyc = 0; cyc < max_cycles; ++cyc) {
        /* exact labels given the frozen halo */
        x.par([&](int t) { x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = MGC_HINF; });
        mgc_tile_bfs(x, [&](int t) {
            int m = snk[t] > 0.0 ? MGC_MASK_SINK : 0;
#pragma unroll
            for (int d = 0; d < 6; ++d) m |= (r[d][t] > 0.0) ? (1 << d) : 0;
            return m;
        });
        active = x.any([&](int t) -> bool {
            hme[t] = x.S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)];
            return e[t] > 0.0 && hme[t] < MGC_HINF;
        });
        if (!active) break;

        for (int sw = 0; sw < max_sweeps; ++sw, ++sweep_id) {
            const int fl = sweep_id & 1;
            /* 7 steps: step s pushes along direction s (s < 6) after receiving direction s-1 */
            auto step = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                x.par([&](int t) {
                    const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
                    if (s == 0) {
                        /* push to the sink first: always admissible (label 1 -> 0) */
                        if (e[t] > 0.0 && snk[t] > 0.0) {
                            const double delta = e[t] < snk[t] ? e[t] : snk[t];
                            e[t] -= delta;
                            snk[t] -= delta;
                            x.S.flag[fl] = 1;
                        }
                    } else {
                        if (s == 1 && t == 0) x.S.flag[fl ^ 1] = 0; /* everybody has read it by now */
                        /* receive what the neighbour pushed in direction s-1 */
                        constexpr int dp = s > 0 ? s - 1 : 0;
                        if (mgc_inside(dp ^ 1, z, y, xx)) {
                            const double din = x.S.out[dp & 1][t - mgc_loc_step(dp)];
                            if (din != 0.0) {
                                e[t] += din;
                                r[dp ^ 1][t] += din;
                            }
                        }
                    }
                    if (s < 6) {
                        constexpr int d = s < 6 ? s : 0;
                        double delta = 0.0;
                        if (e[t] > 0.0 && r[d][t] > 0.0 && hme[t] < MGC_HINF) {
                            const int hv = x.S.hs[mgc_hs_index(z, y, xx) + mgc_hs_step(d)];
                            if (hv == hme[t] - 1) {
                                delta = e[t] < r[d][t] ? e[t] : r[d][t];
                                e[t] -= delta;
                                r[d][t] -= delta;
                                x.S.flag[fl] = 1;
                            }
                        }
                        if (mgc_inside(d, z, y, xx)) {
                            x.S.out[d & 1][t] = delta;
                        } else if (delta != 0.0) {
                            ob[d >> 1][t] += delta;
                        }
                    }
                });
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            if (!x.S.flag[fl]) break; /* uniform: written before the last barrier */
        }
    }
    if (active) {
        /* cycle budget exhausted: is there still something to do with the current labels? */
        active = x.any([&](int t) -> bool { return e[t] > 0.0 && hme[t] < MGC_HINF; });
    }
    const bool has_sink = x.any([&](int t) -> bool { return snk[t] > 0.0; });

    /* store */
    x.par([&](int t) {
        const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
        L.excess[base + t] = e[t];
        L.sink[base + t] = snk[t];
        int m = snk[t] > 0.0 ? MGC_MASK_SINK : 0;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            L.rcap[((int64_t)tile * 6 + d) * MGC_TV + t] = r[d][t];
            m |= (r[d][t] > 0.0) ? (1 << d) : 0;
        }
        L.rmask[base + t] = (uint8_t)m;
        L.height[base + t] = x.S.hs[mgc_hs_index(z, y, xx)];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (ob[a][t] != 0.0) {
                const int c = a == 0 ? xx : (a == 1 ? y : z);
                const int f = 2 * a + (c == 0 ? 0 : 1);
                L.obox[((int64_t)tile * 6 + f) * MGC_TF + mgc_face_index(a, z, y, xx)] += ob[a][t];
                x.S.faceflag[f] = 1;
            }
        }
    });
    x.par([&](int t) {
        if (t < 6 && x.S.faceflag[t]) {
            x.atomic_or(&L.oflags[tile], 1u << t);
            mgc_enqueue(x, L, (int)((phase + 1) & 3u), L.stamp, phase + 1, x.S.nbr[t]);
        }
        if (t == 6 && active) mgc_enqueue(x, L, (int)((phase + 2) & 3u), L.stamp, phase + 2, tile);
        if (t == 7) L.status[tile] = (L.status[tile] & ~2u) | (has_sink ? 2u : 0u);
    });
}

/* ---------------------------------------------------------------------------------------
 * Z-slab halo exchange (multi-GPU).  A slab boundary is an ordinary tile face whose
 * neighbour lives on another GPU: the sender packs, per border tile, the labels of its 64 face
 * voxels, its outbox across that face and the outbox flag; the receiver unpacks them into the
 * GHOST tile that mirrors the sender's tile.  "side" 0 = lower slab boundary (face 4, -z),
 * 1 = upper (face 5, +z).  Buffer layout for T = gy*gx tiles per layer:
 *     double flow[T][64] ; int32 label[T][64] ; int32 flag[T]          (kind 1: discharge phases)
 *     int32 label[T][64]                                              (kind 0: relabel passes)
 * ------------------------------------------------------------------------------------- */
MGC_HD int64_t mgc_halo_bytes(const MgcLattice& L, int kind)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    return kind ? T * (MGC_TF * 8 + MGC_TF * 4 + 4) : T * MGC_TF * 4;
}

/* i = tile index inside the layer; packs the OWNED border tile of `side` */
template <class X>
MGC_HD void mgc_halo_pack_tile(X& x, const MgcLattice& L, int side, int kind, int i, void* buf)
{
    const int64_t T = (int64_t)L.gy * L.gx;
    const int layer = side ? L.tz_own_hi - 1 : L.tz_own_lo;
    const int tile = layer * (int)T + i;
    const int f = side ? 5 : 4;
    double* flow = (double*)buf;
    int32_t* lab = kind ? (int32_t*)((char*)buf + T * MGC_TF * 8) : (int32_t*)buf;
    int32_t* flg = (int32_t*)((char*)buf + T * MGC_TF * 12);
    x.par([&](int t) {
        if (t < MGC_TF) {
            lab[(int64_t)i * MGC_TF + t] = L.height[(int64_t)tile * MGC_TV + mgc_face_voxel(f, t)];
            if (kind) {
                double* slot = &L.obox[((int64_t)tile * 6 + f) * MGC_TF + t];
                flow[(int64_t)i * MGC_TF + t] = *slot;
                *slot = 0.0; /* the flow now travels in the message */
            }
        }
        if (kind && t == MGC_TF) {
            const uint32_t fl = (L.oflags[tile] >> f) & 1u;
            flg[i] = (int32_t)fl;
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
    const double* flow = (const double*)buf;
    const int32_t* lab = kind ? (const int32_t*)((const char*)buf + T * MGC_TF * 8) : (const int32_t*)buf;
    const int32_t* flg = (const int32_t*)((const char*)buf + T * MGC_TF * 12);
    const bool lowered = x.any([&](int t) -> bool {
        bool low = false;
        if (t < MGC_TF) {
            int32_t* hp = &L.height[(int64_t)ghost * MGC_TV + mgc_face_voxel(f, t)];
            const int32_t hn = lab[(int64_t)i * MGC_TF + t];
            low = hn < *hp;
            *hp = hn;
            if (kind) {
                const double d = flow[(int64_t)i * MGC_TF + t];
                if (d != 0.0) L.obox[((int64_t)ghost * 6 + f) * MGC_TF + t] += d;
            }
        }
        return low;
    });
    x.par([&](int t) {
        if (t != 0) return;
        if (kind) {
            if (flg[i]) {
                x.atomic_or(&L.oflags[ghost], 1u << f);
                mgc_enqueue(x, L, (int)((epoch + 1) & 3u), L.stamp, epoch + 1, own);
            }
        } else if (lowered) {
            mgc_enqueue(x, L, list, L.rstamp, epoch, own);
        }
    });
}

#endif /* MGC_TILE_OPS_INL */
