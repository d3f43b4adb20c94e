/*
 * mgc_wave_ops26.inl -- ONE WAVE PER TILE form of the region discharge of the FULL neighbourhood (26 in 3-D, 8 in 2-D).
 * Same state in HBM, same schedule, same contract as mgc26_discharge_tile (mgc_tile_ops26.inl); what changes is where a
 * tile lives while it is discharged:
 *
 *   a wave64 owns a whole 8x8x8 tile, lane = (y, x), and every lane keeps its z-COLUMN of eight voxels in registers:
 *   excess and ALL 26 residual planes = 216 doubles = 432 registers.  gfx950 gives a wave that runs alone on its SIMD the
 *   whole 512-entry register file (256 VGPRs + 256 accumulator registers, one file): a tile then needs no LDS for its
 *   capacities at all and a CU keeps FOUR tiles in flight (the 512-thread form: 13 planes in 74 KB of LDS, two tiles per
 *   CU, 128 VGPRs and scratch).  Labels, residual masks, sink links and the per-sweep push masks live in LDS (26 KB per
 *   wave), so the per-slot passes over them are run-time loops whose code exists once.
 *
 *   - a push along (dz, dy, dx) inside the tile is a lane shift by 8 dy + dx plus a static register index K + dz: no
 *     barrier anywhere, every decision is a wave-uniform mask;
 *   - per sweep ONE pass over the labels decides, per voxel that holds excess, along which residual arcs it may push
 *     (mask `cand`, 26 + 1 bits); its OR over the wave says which of the 27 x 8 (direction, slot) steps run at all --
 *     the steps themselves are straight-line code behind scalar bit tests;
 *   - pushes over the tile boundary update the idle neighbour tile IN PLACE as in the 512-thread form (eight tile
 *     colours: one running tile next to any idle voxel), but as fire-and-forget memory atomics issued from a per-slot
 *     run-time loop (the code of the address arithmetic exists once, and nobody waits for a returning load);
 *   - the local relabel that ends a sweep is a second run-time pass over the slots that still hold excess.
 *
 * Order inside a sweep: labels are frozen while pushes run (pass A reads them, the steps push, pass R raises the
 * labels of voxels that still hold excess), so every push goes exactly one label down and the labels stay valid lower
 * bounds of the distance to the sink, as in mgc26_discharge_tile.
 *
 * No reference counterpart for the neighbourhood (reference medpy/graphcut/generate.py:44-49); replaces what
 * Graph::maxflow (reference lib/maxflow/src/maxflow.cpp:472-604) does on the graph.
 *
 * Executor concept: the wave executor of mgc_wave_ops.inl plus
 *   w.wave_or(f)              OR of f(l) over the 64 lanes (wave-uniform)
 *   w.uput(reg, k, v) / w.uget(reg, k)   a wave-uniform word per k = 0..63 kept in ONE register (lane k holds word k)
 *   w.lds_and(p, v) / w.lds_or(p, v)     *p &= v / *p |= v on an LDS word of the lane's own, nobody waits for it
 *   W::RegA<N>                N doubles per lane with init(l, k, v) (first value) / get(l, k) / set(l, k, v): accumulator registers on the GPU
 *   w.pin(x)                  x is materialised in a register at this point (GPU; nothing on the host)
 *   w.gadd(p, v) / w.gor(p, v)           *p += v (f64) / *p |= v on a global word no other wave touches in this launch:
 *                                        memory atomics without return on the GPU, plain updates on the host
 */
#ifndef MGC_WAVE_OPS26_INL
#define MGC_WAVE_OPS26_INL

#include "mgc_tile_ops26.inl"
#include "mgc_wave_ops.inl"

#define MGCW26_ALL_SLOTS 1 /* discharge flag, see pass A */
#define MGCW26_SAT_DIRTY 2 /* discharge flag: the visit runs on radial labels (mgc_dt_ops.inl) -- any saturated arc marks the tile DIRTY */
#ifndef MGCW26_NLDS
#define MGCW26_NLDS 3
#endif
struct alignas(16) MgcWaveShared26 {
    int32_t  hs[1000];              /* 10x10x10 distance labels: the tile plus a one-voxel halo (all 26 neighbour tiles)      */
    uint32_t m[MGC_TV];             /* residual masks of the tile (rmask32), kept current as arcs saturate / reappear         */
    uint32_t cand[MGC_TV];          /* this sweep: residual AND admissible directions of the voxels that hold excess          */
    double   snk[MGC_TV];           /* residual sink links                                                                     */
    double   out[MGC26_NDIR][MGCW_LANES]; /* flow the slot being processed pushed over the tile boundary, per direction        */
    double   rl[MGCW26_NLDS > 0 ? MGCW26_NLDS : 1][MGC_TV]; /* the residual planes that do not fit the register file (mgcw26_lds_plane)   */
};

/* Where the 26 residual planes of a tile live while it is discharged.  432 registers of tile state leave no room for anything
 * else in a 512-entry file, and the compiler, left alone, gives EVERY plane a spill slot (256 accumulator registers hold 128
 * doubles, the rest went to scratch memory).  So the homes are spelled out:
 *   0  ordinary vector registers (8 planes, 128 registers, next to the excess and the temporaries),
 *   1  accumulator registers, moved with v_accvgpr_read / v_accvgpr_write around every use (15 planes = 240 of 256),
 *   2  LDS, a word of the lane's own: ds_read_b64 / ds_write_b64 per use (MGCW26_NLDS = 3 planes; a wave has 40 KB of LDS to
 *      itself, four waves per CU).
 * Which plane goes where is arbitrary (every plane is used as often as its opposite). */
MGC_HD constexpr int mgcw26_lds_plane(int d) { return d == 12 ? 0 : (d == 13 ? 1 : (d == 4 ? 2 : -1)); }
MGC_HD constexpr int mgcw26_plane_home(int d) { return mgcw26_lds_plane(d) >= 0 && mgcw26_lds_plane(d) < MGCW26_NLDS ? 2 : ((d < 4 || d > 21) ? 0 : 1); }

/* does the neighbour of lane l = (y, x) at in-plane offset (dy, dx) lie inside the tile's 8 x 8 cross-section? */
MGC_HD bool mgcw26_in_xy(int l, int dy, int dx)
{
    const int y = l >> 3, x = l & 7;
    return (dx < 0 ? x > 0 : (dx > 0 ? x < 7 : true)) && (dy < 0 ? y > 0 : (dy > 0 ? y < 7 : true));
}

/* neighbour tile at tile offset (oz, oy, ox), or -1 outside the grid */
MGC_HD int mgcw26_nbr_tile(const MgcLattice& L, int tz, int ty, int tx, int oz, int oy, int ox)
{
    const int nz = tz + oz, ny = ty + oy, nx = tx + ox;
    return (nz >= 0 && nz < L.gz && ny >= 0 && ny < L.gy && nx >= 0 && nx < L.gx) ? mgc_tile_id(L, nz, ny, nx) : -1;
}

/* The 488 halo cells of the label block in eight batches of 64 lanes: batches 0..5 the six faces (cell = lane), batch 6
 * the edges 0..7, batch 7 the edges 8..11 (lanes 0..31) and the eight corners (lanes 32..39).  Edge e: free axis a = e >> 2
 * (0: x, 1: y, 2: z), the two other axes at their low / high end by the bits of e & 3.
 * Returns the cell's index in hs[] (or -1: no cell for this lane) and where its label lives (tile, local voxel). */
MGC_HD int mgcw26_halo_cell(const MgcLattice& L, int tz, int ty, int tx, int batch, int l, int& nt, int& loc)
{
    int oz = 0, oy = 0, ox = 0, i = l & 7, j = (l >> 3) & 7;
    if (batch < 6) {
        const int a = batch >> 1, s = (batch & 1) ? 1 : -1;
        if (a == 0) ox = s; else if (a == 1) oy = s; else oz = s;
    } else {
        const int sel = batch == 6 ? (l >> 3) : (l < 32 ? 8 + (l >> 3) : (l < 40 ? 12 : -1));
        if (sel < 0) { nt = -1; loc = 0; return -1; }
        if (sel == 12) { oz = (l & 4) ? 1 : -1; oy = (l & 2) ? 1 : -1; ox = (l & 1) ? 1 : -1; }
        else {
            const int a = sel >> 2, s0 = (sel & 1) ? 1 : -1, s1 = (sel & 2) ? 1 : -1;
            if (a == 0) { oz = s0; oy = s1; } else if (a == 1) { oz = s0; ox = s1; } else { oy = s0; ox = s1; }
        }
    }
    /* coordinates of the cell relative to the tile: an axis at offset 0 runs over the face / edge (faces: two free axes
     * (j, i) in (z, y, x) order; edges: one, i) */
    int cz, cy, cx;
    if (batch < 6) {
        const int a = batch >> 1;
        cz = a == 2 ? (oz < 0 ? -1 : 8) : j;
        cy = a == 1 ? (oy < 0 ? -1 : 8) : (a == 2 ? j : i);
        cx = a == 0 ? (ox < 0 ? -1 : 8) : i;
    } else {
        cz = oz ? (oz < 0 ? -1 : 8) : i;
        cy = oy ? (oy < 0 ? -1 : 8) : i;
        cx = ox ? (ox < 0 ? -1 : 8) : i;
    }
    nt = mgcw26_nbr_tile(L, tz, ty, tx, oz, oy, ox);
    loc = mgc_local(cz & 7, cy & 7, cx & 7);
    return mgc_hs_index(cz, cy, cx);
}

/* ---------------------------------------------------------------------------------------
 * Region discharge of one tile by one wave, full neighbourhood, stored labels (valid lower bounds; see the header).
 * ------------------------------------------------------------------------------------- */
template <class W>
MGC_HD void mgcw26_discharge_tile(W& w, const MgcLattice& L, int tile, uint32_t phase, int max_sweeps, int max_passes = 2, int max_raises = 1, int flags = 0)
{
    typename W::template Reg<double, 8> e;
    typename W::template Reg<double, 8> r[MGC26_NDIR];   /* planes at home 0 (the others' slots are never touched: no registers) */
    typename W::template RegA<8> ra[MGC26_NDIR];          /* planes at home 1 */
    typename W::template Reg<double, 1> stay, outv, din;
    /* residual of (lane, slot K) along D: a register, or an LDS word of the lane's own (mgcw26_lds_plane) */
    auto RGET = [&](auto DD, auto KK, int l) MGCW_INL -> double {
        constexpr int D = decltype(DD)::value, K = decltype(KK)::value, H = mgcw26_plane_home(D);
        if constexpr (H == 2) return w.S.rl[mgcw26_lds_plane(D)][K * 64 + l];
        else if constexpr (H == 1) return ra[D].get(l, K);
        else return r[D](l, K);
    };
    auto RINIT = [&](auto DD, auto KK, int l, double v) MGCW_INL { /* the value loaded from HBM */
        constexpr int D = decltype(DD)::value, K = decltype(KK)::value, H = mgcw26_plane_home(D);
        if constexpr (H == 2) w.S.rl[mgcw26_lds_plane(D)][K * 64 + l] = v;
        else if constexpr (H == 1) ra[D].init(l, K, v);
        else r[D](l, K) = v;
    };
    auto RSET = [&](auto DD, auto KK, int l, double v) MGCW_INL {
        constexpr int D = decltype(DD)::value, K = decltype(KK)::value, H = mgcw26_plane_home(D);
        if constexpr (H == 2) w.S.rl[mgcw26_lds_plane(D)][K * 64 + l] = v;
        else if constexpr (H == 1) ra[D].set(l, K, v);
        else r[D](l, K) = v;
    };
    typename W::template Reg<int, 1> cnd, epos, satl, nbm, cmv, newh, ob;
    typename W::template Reg<int, 8> m8; /* the residual masks of the lane's column: updated in registers while the steps run, in LDS (w.S.m) for the passes over the labels */

    double* const t_excess = L.excess + (int64_t)tile * MGC_TV;
    double* const t_sink = L.sink + (int64_t)tile * MGC_TV;
    double* const t_rcap = L.rcap + (int64_t)tile * MGC26_NDIR * MGC_TV;
    uint32_t* const t_rmask = L.rmask32 + (int64_t)tile * MGC_TV;
    int32_t* const t_height = L.height + (int64_t)tile * MGC_TV;
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    const uint32_t st0 = L.status[tile]; /* nobody else writes it during this launch */
    const bool SINK = (st0 & MGC_ST_SINK) != 0;

    /* ---- one trip to HBM: the label halo (issued first: it comes back first), then the tile's own state ---- */
    {
        typename W::template Reg<int, 8> hv, hi;
        w.lanes([&](int l) MGCW_INL {
            mgcw_static_for<8>([&](auto BB) MGCW_INL { /* (branch-free: a label outside the grid is MGC_HINF by a select, not by a skipped load) */
                constexpr int B = decltype(BB)::value;
                int nt, loc;
                hi(l, B) = mgcw26_halo_cell(L, tz, ty, tx, B, l, nt, loc);
                const int v = L.height[(int64_t)(nt >= 0 ? nt : tile) * MGC_TV + loc];
                hv(l, B) = (hi(l, B) >= 0 && nt >= 0) ? v : MGC_HINF;
            });
            /* (1) what goes to LDS: masks, labels, sink links, the residual planes that live there -- loaded into temporaries,
             * stored to LDS behind the loads of (2) */
            typename W::template Reg<int, 8> tm, th;
            typename W::template Reg<double, 8> ts, tp[MGCW26_NLDS > 0 ? MGCW26_NLDS : 1];
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                tm(l, K) = (int)w.ld(t_rmask, K * 64 + l);
                th(l, K) = w.ld(t_height, K * 64 + l);
                ts(l, K) = 0.0;
                mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    if constexpr (mgcw26_plane_home(D) == 2) tp[mgcw26_lds_plane(D)](l, K) = w.ld(t_rcap + D * MGC_TV, K * 64 + l);
                });
            });
            if (SINK) mgcw_static_for<8>([&](auto KK) MGCW_INL { constexpr int K = decltype(KK)::value; ts(l, K) = w.ld(t_sink, K * 64 + l); });
            /* (2) excess and the planes in ordinary registers: straight into their registers */
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                e(l, K) = w.ld(t_excess, K * 64 + l);
                mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    if constexpr (mgcw26_plane_home(D) == 0) RINIT(DD, KK, l, w.ld(t_rcap + D * MGC_TV, K * 64 + l));
                });
            });
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                w.S.m[K * 64 + l] = (uint32_t)tm(l, K);
                m8(l, K) = tm(l, K);
                w.S.hs[mgcw_hs(l, K)] = th(l, K);
                w.S.snk[K * 64 + l] = ts(l, K);
                mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    if constexpr (mgcw26_plane_home(D) == 2) RINIT(DD, KK, l, tp[mgcw26_lds_plane(D)](l, K));
                });
            });
            mgcw_static_for<8>([&](auto BB) MGCW_INL {
                constexpr int B = decltype(BB)::value;
                if (hi(l, B) >= 0) w.S.hs[hi(l, B)] = hv(l, B);
            });
            /* (3) the planes in accumulator registers: a load lands in an ordinary register and is moved; the moves are volatile
             * statements no load crosses, so the loads of a batch are WRITTEN before the moves of the batch before it -- two
             * planes per batch, two batches in flight.  (A load followed by its move, plane by plane, was one trip to HBM per
             * value: 140 k cycles per tile.) */
            {
                typename W::template Reg<double, 16> tb[2];
                auto plane_of = [](int i) constexpr -> int { return i < 7 ? 5 + i : 14 + (i - 7); }; /* the 15 planes at home 1 */
                auto issue = [&](auto BB) MGCW_INL {
                    constexpr int B = decltype(BB)::value;
                    mgcw_static_for<2>([&](auto JJ) MGCW_INL {
                        constexpr int J = decltype(JJ)::value, I = 2 * B + J;
                        if constexpr (I < 15) {
                            constexpr int D = plane_of(I);
                            static_assert(mgcw26_plane_home(D) == 1, "plane_of lists the accumulator planes");
                            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                                constexpr int K = decltype(KK)::value;
                                tb[B & 1](l, J * 8 + K) = w.ld(t_rcap + D * MGC_TV, K * 64 + l);
                            });
                        }
                    });
                };
                auto commit = [&](auto BB) MGCW_INL {
                    constexpr int B = decltype(BB)::value;
                    mgcw_static_for<2>([&](auto JJ) MGCW_INL {
                        constexpr int J = decltype(JJ)::value, I = 2 * B + J;
                        if constexpr (I < 15) {
                            constexpr int D = plane_of(I);
                            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                                constexpr int K = decltype(KK)::value;
                                ra[D].init(l, K, tb[B & 1](l, J * 8 + K));
                            });
                        }
                    });
                };
                issue(std::integral_constant<int, 0>{});
                mgcw_static_for<8>([&](auto BB) MGCW_INL {
                    constexpr int B = decltype(BB)::value;
                    if constexpr (B + 1 < 8) issue(std::integral_constant<int, B + 1>{});
                    commit(BB);
                });
            }
            satl(l, 0) = 0;
            nbm(l, 0) = 0;
            cmv(l, 0) = 0;
        });
    }
    w.mark(0); /* load */
#if defined(MGCW26_CHECK_HALO) /* host check: the label block equals the one the 512-thread form builds */
    for (int k = 0; k < 1000; ++k) {
        const int z = k / 100 - 1, y = (k / 10) % 10 - 1, xx = k % 10 - 1;
        const int oz = z < 0 ? -1 : (z > 7 ? 1 : 0), oy = y < 0 ? -1 : (y > 7 ? 1 : 0), ox = xx < 0 ? -1 : (xx > 7 ? 1 : 0);
        const int nt = (oz || oy || ox) ? mgcw26_nbr_tile(L, tz, ty, tx, oz, oy, ox) : tile;
        const int want = nt < 0 ? MGC_HINF : L.height[(int64_t)nt * MGC_TV + mgc_local(z & 7, y & 7, xx & 7)];
        if (w.S.hs[k] != want) { fprintf(stderr, "halo cell %d (%d %d %d): %d, want %d\n", k, z, y, xx, w.S.hs[k], want); abort(); }
    }
#endif

    uint32_t dirty = 0;       /* bit D: residual plane D changed (somebody pushed along D, or received along 25 - D) */
    bool relabelled = false;  /* some label of the tile changed */
    bool active = false;      /* excess under a finite label is left when the sweep budget runs out */

    /* one (slot K, direction D) step: every lane whose mask allows it pushes min(excess, residual) along D; what stays inside
     * the tile is handed to lane + 8 dy + dx, slot K + dz; what leaves it is parked in LDS for the flush of the slot */
    uint32_t OUT = 0; /* directions with parked outflow of the slot being processed */
    bool moved = false; /* some step of the current pass over the slots ran */
    bool moved_slot = false; /* ... of the slot just processed */
    auto step = [&](auto KK, auto DD) MGCW_INL {
        constexpr int K = decltype(KK)::value;
        constexpr int D = decltype(DD)::value;
        constexpr int C = D < 13 ? D : D + 1;
        constexpr int dz = C / 9 - 1, dy = (C / 3) % 3 - 1, dx = C % 3 - 1;
        constexpr int K2 = K + dz;
        constexpr bool z_in = K2 >= 0 && K2 < 8;
        dirty |= (1u << D) | (1u << (25 - D));
        w.lanes([&](int l) MGCW_INL {
            const bool can = (((uint32_t)cnd(l, 0) >> D) & 1u) != 0;
            const double rd = RGET(DD, KK, l);
            const double delta = can ? fmin(e(l, K), rd) : 0.0; /* (the mask bit says rd > 0; excess may have gone elsewhere: 0.0) */
            e(l, K) -= delta;
            RSET(DD, KK, l, rd - delta); /* saturating push: rd - rd == 0.0 exactly */
            const bool sat = can && delta == rd;
            satl(l, 0) |= sat ? (1 << K) : 0;
            m8(l, K) &= sat ? (int)~(1u << D) : -1;
            const bool inside = z_in && mgcw26_in_xy(l, dy, dx);
            stay(l, 0) = inside ? delta : 0.0;
            outv(l, 0) = inside ? 0.0 : delta;
        });
        if constexpr (z_in) {
            if constexpr (dy == 0 && dx == 0) w.lanes([&](int l) MGCW_INL { din(l, 0) = stay(l, 0); });
            else if constexpr (dy == 0) w.shift_x(din, stay, -dx); /* stay is 0.0 on the lanes at the end of an x-row */
            else w.shift(din, stay, -(dy * 8 + dx));               /* ... and on every lane whose target lies outside the cross-section */
            w.lanes([&](int l) MGCW_INL { /* what the neighbour pushed arrives: the reverse residual grows */
                if constexpr (z_in) {
                    const double d = din(l, 0);
                    constexpr int KR = K2 < 0 ? 0 : (K2 > 7 ? 7 : K2);
                    constexpr std::integral_constant<int, 25 - D> DR{};
                    constexpr std::integral_constant<int, KR> KRC{};
                    e(l, KR) += d;
                    RSET(DR, KRC, l, RGET(DR, KRC, l) + d);
                    m8(l, KR) |= d > 0.0 ? (int)(1u << (25 - D)) : 0;
                }
            });
        }
        /* what leaves the tile is parked unconditionally (a store nobody waits for) and flagged per lane: one OR over the wave per
         * slot says which directions the flush has to look at (a vote per step was a compare -> branch round trip 200 times a sweep) */
        if constexpr (!z_in || dy != 0 || dx != 0) {
            w.lanes([&](int l) MGCW_INL {
                w.S.out[D][l] = outv(l, 0);
                ob(l, 0) |= outv(l, 0) != 0.0 ? (int)(1u << D) : 0;
            });
        }
        w.mark(5); /* one step */
    };
    auto slot = [&](auto KK) MGCW_INL {
        constexpr int K = decltype(KK)::value;
        w.lanes([&](int l) MGCW_INL { cnd(l, 0) = (int)w.S.cand[K * 64 + l] & m8(l, K); ob(l, 0) = 0; }); /* (arcs saturated since pass A are out) */
        /* the steps of this slot that run: the directions along which a voxel that holds excess NOW can push (what arrived from
         * the slots below during this sweep moves on at once); in a step that runs, every voxel pushes that can */
        const uint32_t CK = w.wave_or([&](int l) MGCW_INL -> uint32_t { return e(l, K) > 0.0 ? (uint32_t)cnd(l, 0) : 0u; });
        moved_slot = CK != 0;
        if (!CK) return;
        moved = true;
        if (CK & MGC26_MASK_SINK) { /* push to the sink first: always admissible (label 1 -> 0) */
            w.lanes([&](int l) MGCW_INL {
                const bool can = (((uint32_t)cnd(l, 0) >> 26) & 1u) != 0;
                const double sk = w.S.snk[K * 64 + l];
                const double delta = can ? fmin(e(l, K), sk) : 0.0;
                e(l, K) -= delta;
                w.S.snk[K * 64 + l] = sk - delta;
                const bool sat = can && delta == sk;
                satl(l, 0) |= sat ? (1 << K) : 0;
                m8(l, K) &= sat ? (int)~MGC26_MASK_SINK : -1;
            });
        }
        mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
            constexpr int D = decltype(DD)::value;
            if (CK & (1u << D)) step(KK, DD);
        });
    };

#ifdef MGCW26_DEBUG_NOSWEEP
    max_sweeps = 0;
#endif
    for (int sw = 0; sw < max_sweeps; ++sw) {
        /* ---- pass A: per voxel that holds excess, the residual arcs that go one label down ---- */
        w.lanes([&](int l) MGCW_INL {
            int m = 0;
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                m |= e(l, K) > 0.0 ? (1 << K) : 0;
            });
            epos(l, 0) = m;
        });
        uint32_t anyc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int K = 0; K < 8; ++K) {
            /* MGCW26_ALL_SLOTS: masks also for the slots that hold no excess yet (flow that arrives there during the sweep moves on) */
            if (!(flags & MGCW26_ALL_SLOTS) && !w.any([&](int l) MGCW_INL -> bool { return ((epos(l, 0) >> K) & 1) && w.S.hs[mgcw_hs(l, K)] < MGC_HINF; })) {
                w.uput(cmv, K, 0u);
                continue;
            }
            w.lanes([&](int l) MGCW_INL {
                const int me = mgcw_hs(l, K);
                const int h = w.S.hs[me];
                const bool act = ((epos(l, 0) >> K) & 1) && h < MGC_HINF;
                uint32_t adm = MGC26_MASK_SINK; /* (a voxel with a residual sink link has label 1) */
                mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    adm |= (w.S.hs[me + mgc26_hs_step(D)] == h - 1) ? (1u << D) : 0u;
                });
                /* (for every voxel under a finite label, not only the ones that hold excess now: flow that arrives during the sweep moves on) */
                cnd(l, 0) = h < MGC_HINF ? (int)(adm & w.S.m[K * 64 + l]) : 0;
                w.S.cand[K * 64 + l] = (uint32_t)cnd(l, 0);
                if (!act) cnd(l, 0) = 0;
            });
            const uint32_t CK = w.wave_or([&](int l) MGCW_INL -> uint32_t { return (uint32_t)cnd(l, 0); });
            w.uput(cmv, K, 1u);
            anyc |= CK;
        }
        w.mark(1); /* pass A */

        /* ---- the steps that somebody can take, slot by slot; then what left the tile.  (Static slots: a run-time loop around a
         * switch over the slot makes every register of the tile a phi of nine paths, and the register allocator answers with
         * copies of the whole state.) ---- */
        /* Labels are frozen until pass R, so the masks of pass A stay good: the steps are repeated (same code) while they move
         * something -- flow that a later slot handed DOWN, or that arrived behind a direction's turn, moves on without another
         * pass over the labels.  (One pass per sweep moved flow half as far per sweep as the 512-thread form does.) */
        for (int pass = 0; anyc && pass < max_passes; ++pass) {
            moved = false;
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                if (!w.uget(cmv, K)) return; /* no voxel of the slot held excess when the sweep began: its masks were not computed */
                slot(KK);
                OUT = moved_slot ? w.wave_or([&](int l) MGCW_INL -> uint32_t { return (uint32_t)ob(l, 0); }) : 0u;
                /* flush: the target voxel lives in an idle neighbour tile and nobody else writes it in this launch: its excess,
                 * reverse residual and mask are updated in place, direction by direction in a fixed order */
                while (OUT) {
                    const int D = __builtin_ctz(OUT);
                    OUT &= OUT - 1;
                    int dz, dy, dx;
                    mgc26_offset(D, dz, dy, dx);
                    w.lanes([&](int l) MGCW_INL {
                        const double d = w.S.out[D][l];
                        if (d != 0.0) {
                            const int vz = K + dz, vy = (l >> 3) + dy, vx = (l & 7) + dx;
                            const int oz = vz < 0 ? -1 : (vz > 7 ? 1 : 0), oy = vy < 0 ? -1 : (vy > 7 ? 1 : 0), ox = vx < 0 ? -1 : (vx > 7 ? 1 : 0);
                            const int64_t nt = (int64_t)tile + ((int64_t)oz * L.gy + oy) * L.gx + ox; /* exists: the arc does */
                            const int lv = mgc_local(vz & 7, vy & 7, vx & 7);
                            w.gadd(L.excess + nt * MGC_TV + lv, d);
                            w.gadd(L.rcap + (nt * MGC26_NDIR + (25 - D)) * MGC_TV + lv, d);
                            w.gor(L.rmask32 + nt * MGC_TV + lv, 1u << (25 - D));
                            nbm(l, 0) |= 1 << ((oz + 1) * 9 + (oy + 1) * 3 + (ox + 1));
                        }
                    });
                }
                w.mark(6); /* a slot's flush (and the votes between its last step and here) */
            });
            if (!moved) break;
        }
        if (anyc) w.lanes([&](int l) MGCW_INL { mgcw_static_for<8>([&](auto KK) MGCW_INL { w.S.m[decltype(KK)::value * 64 + l] = (uint32_t)m8(l, decltype(KK)::value); }); });
        w.mark(2); /* (the rest of the passes over the steps) */

        /* ---- pass R, the local relabel: a voxel that still holds excess rises to 1 + the lowest label behind a residual arc
         * (no change while one of them is still admissible).  Slot by slot; inside a slot all lanes read before any writes ---- */
        w.lanes([&](int l) MGCW_INL {
            int m = 0;
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                m |= e(l, K) > 0.0 ? (1 << K) : 0;
            });
            epos(l, 0) = m;
        });
        bool raised = false;
        for (int rr = 0; rr < max_raises; ++rr) { /* (a region of stuck voxels rises by one label per round) */
        bool raised_now = false;
        active = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int K = 0; K < 8; ++K) {
            if (!w.any([&](int l) MGCW_INL -> bool { return ((epos(l, 0) >> K) & 1) && w.S.hs[mgcw_hs(l, K)] < MGC_HINF; })) continue;
            w.lanes([&](int l) MGCW_INL {
                const int me = mgcw_hs(l, K);
                const int h = w.S.hs[me];
                const bool act = ((epos(l, 0) >> K) & 1) && h < MGC_HINF;
                const uint32_t m = w.S.m[K * 64 + l];
                int c = (m & MGC26_MASK_SINK) ? 1 : MGC_HINF;
                mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    const int hv = w.S.hs[me + mgc26_hs_step(D)];
                    const int cd = ((m >> D) & 1u) ? hv + 1 : MGC_HINF; /* hv == MGC_HINF gives a value above every label */
                    c = cd < c ? cd : c;
                });
                c = c < MGC_HINF ? c : MGC_HINF;
                newh(l, 0) = (act && c > h) ? c : h;
                cnd(l, 0) = (act && c > h) ? 1 : ((act && c < MGC_HINF) ? 2 : 0); /* 1: rises, 2: keeps an admissible arc */
            });
            w.lanes([&](int l) MGCW_INL { w.S.hs[mgcw_hs(l, K)] = newh(l, 0); });
            if (w.any([&](int l) MGCW_INL -> bool { return cnd(l, 0) == 1; })) raised_now = true;
            if (w.any([&](int l) MGCW_INL -> bool { return cnd(l, 0) == 2 || (cnd(l, 0) == 1 && newh(l, 0) < MGC_HINF); })) active = true;
        }
        raised = raised || raised_now;
        if (!raised_now) break;
        }
        relabelled = relabelled || raised;
        w.mark(3); /* pass R */
        if (!active) break; /* nothing left that could move */
        if (!anyc && !raised) break; /* (cannot happen: a voxel with excess and no admissible arc rises) */
    }

    /* ---- tail: wake-ups first (two dependent returning atomics per woken tile), the write-back behind them ---- */
    const uint32_t NB = w.wave_or([&](int l) MGCW_INL -> uint32_t { return (uint32_t)nbm(l, 0); });
    /* DIRTY (the tile's labels may no longer be exact distances; the next global relabel recomputes it and whoever depends on it)
     * iff a label rose, or a voxel that saturated an arc has no residual arc one label down left.  A voxel that keeps one of its
     * supports keeps its distance: with 26 neighbours most do, and the tiles a small flow merely passes through stay clean. */
    bool saturated = relabelled;
    if (!saturated && (flags & MGCW26_SAT_DIRTY)) saturated = w.any([&](int l) MGCW_INL -> bool { return satl(l, 0) != 0; });
    if (!saturated && w.any([&](int l) MGCW_INL -> bool { return satl(l, 0) != 0; })) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int K = 0; K < 8 && !saturated; ++K) {
            if (!w.any([&](int l) MGCW_INL -> bool { return ((satl(l, 0) >> K) & 1) != 0; })) continue;
            saturated = w.any([&](int l) MGCW_INL -> bool {
                const int me = mgcw_hs(l, K);
                const int h = w.S.hs[me];
                const uint32_t m = w.S.m[K * 64 + l];
                bool kept = (m & MGC26_MASK_SINK) != 0; /* (a label of 1 stands on the sink link) */
                mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL {
                    constexpr int D = decltype(DD)::value;
                    constexpr int C = D < 13 ? D : D + 1;
                    kept = kept || (((m >> D) & 1u) && w.S.hs[me + mgc26_hs_step(D)] == h - 1 && mgc26_support_watched(st0, K, l >> 3, l & 7, C / 9 - 1, (C / 3) % 3 - 1, C % 3 - 1));
                });
                return ((satl(l, 0) >> K) & 1) && h < MGC_HINF && !kept;
            });
        }
    }
    bool has_sink = false;
    if (SINK) {
        has_sink = w.any([&](int l) MGCW_INL -> bool {
            uint32_t m = 0;
            mgcw_static_for<8>([&](auto KK) MGCW_INL { m |= (uint32_t)m8(l, decltype(KK)::value); });
            return (m & MGC26_MASK_SINK) != 0;
        });
    }
    w.fresh();
    w.lanes([&](int l) MGCW_INL {
        int wake = -1;
        uint32_t target = 0;
        if (l < 27 && l != 13 && ((NB >> l) & 1u)) {
            const int oz = l / 9 - 1, oy = (l / 3) % 3 - 1, ox = l % 3 - 1;
            const int mine = mgc26_colour(L, tz, ty, tx);
            const int theirs = mgc26_colour(L, tz + oz, ty + oy, tx + ox);
            target = phase + (uint32_t)((theirs - mine) & 7);
            wake = mgc_tile_id(L, tz + oz, ty + oy, tx + ox);
            if (!mgc_owned(L, wake)) w.atomic_or(&L.oflags[wake], 1u); /* ghost: the halo exchange ships what it received */
        }
        if (l == 13 && active) { wake = tile; target = phase + 8; }
        if (wake >= 0) mgc_enqueue(w, L, (int)(target & 15u), L.stamp, target, wake);
        /* DIRTY only if a residual arc disappeared: otherwise no distance in the tile (or through it) can have changed */
        if (l == 28) L.status[tile] = (st0 & ~MGC_ST_SINK) | (has_sink ? MGC_ST_SINK : 0u) | (saturated ? MGC_ST_DIRTY : 0u);
    });
    w.lanes([&](int l) MGCW_INL {
        mgcw_static_for<8>([&](auto KK) MGCW_INL {
            constexpr int K = decltype(KK)::value;
            w.st_stream(t_excess, K * 64 + l, e(l, K)); /* (streaming stores for the tile's own state, as in mgc_wave_ops.inl) */
            w.st_stream(t_rmask, K * 64 + l, (uint32_t)m8(l, K));
            if (SINK) w.st_stream(t_sink, K * 64 + l, w.S.snk[K * 64 + l]);
            if (relabelled) w.st(t_height, K * 64 + l, w.S.hs[mgcw_hs(l, K)]);
        });
    });
    mgcw_static_for<MGC26_NDIR>([&](auto DD) MGCW_INL { /* only the residual planes that changed */
        constexpr int D = decltype(DD)::value;
        if (!(dirty & (1u << D))) return;
        w.lanes([&](int l) MGCW_INL {
            mgcw_static_for<8>([&](auto KK) MGCW_INL {
                constexpr int K = decltype(KK)::value;
                w.st_stream(t_rcap + D * MGC_TV, K * 64 + l, RGET(DD, KK, l));
            });
        });
    });
    w.mark(4); /* tail */
}

#endif /* MGC_WAVE_OPS26_INL */
