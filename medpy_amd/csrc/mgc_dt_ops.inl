/*
 * mgc_dt_ops.inl -- the FIRST global relabel of a solve as a separable distance transform.
 *
 * Right after mgc_build every n-link inside the volume is residual (all eight boundary terms floor their weights at
 * DBL_MIN > 0, energy_voxel.py:113,188,236,300,...; k_build counts the tiles where that does not hold -- NaN weights of a
 * *_linear term on a constant image, no boundary term at all -- in MGC_CNT_NOT_FULL).  The residual graph is then the
 * full 6-connected box lattice, and the distance label the relaxation passes converge to,
 *      h(v) = 1 + min over sink-linked voxels s of |v - s|_1          (MGC_HINF when no voxel has a sink link),
 * is an L1 distance transform: three axis passes (forward + backward min-plus scans), 6 streaming kernels over the
 * volume instead of ~70 dependent tile-relaxation passes whose label wave advances one tile per launch.
 *
 * Replaces (reference): nothing one to one -- BK grows its sink tree breadth first from the terminal
 * (maxflow.cpp:119-156, 506-559); this is the same breadth-first distance on the unsaturated lattice.
 *
 * Work unit = one TILE LINE (all tiles with the same two cross coordinates) by one wave: lane = one of the 8 x 8 voxel
 * lines through those tiles, walking them front to back with its running minimum in a register.  Intermediate distances
 * are uint16 (the caller checks D0 + D1 + D2 < 65535), the last pass widens to the int32 labels.
 * Written against the wave executor of mgc_wave_ops.inl (W::lanes), so tests/hostsim runs this very source.
 */
#ifndef MGC_DT_OPS_INL
#define MGC_DT_OPS_INL

#include "mgc_wave_ops.inl"

#define MGC_DT_INF 65535

/* local voxel index of position i along AXIS (0 = x, 1 = y, 2 = z) on the line of lane l */
template <int AXIS>
MGC_HD int mgc_dt_loc(int l, int i)
{
    return AXIS == 0 ? l * 8 + i : (AXIS == 1 ? (l >> 3) * 64 + i * 8 + (l & 7) : i * 64 + l);
}

/* tile of line `line` (index over the two cross tile coordinates) at position a along AXIS */
template <int AXIS>
MGC_HD int mgc_dt_tile(const MgcLattice& L, int line, int a)
{
    if (AXIS == 0) return line * L.gx + a;                                  /* line = tz * gy + ty */
    if (AXIS == 1) return ((line / L.gx) * L.gy + a) * L.gx + line % L.gx;  /* line = tz * gx + tx */
    return a * (L.gy * L.gx) + line;                                        /* line = ty * gx + tx */
}

template <int AXIS>
MGC_HD int mgc_dt_lines(const MgcLattice& L) { return AXIS == 0 ? L.gz * L.gy : (AXIS == 1 ? L.gz * L.gx : L.gy * L.gx); }

/* label steps that an L1 distance of `dist` from the source is worth.  6-neighbourhood: one per hop.  Full neighbourhood: an arc changes the L1
 * distance by up to three (a corner neighbour), so floor(dist / 3) is what stays 1-Lipschitz along EVERY one of the 26 arcs -- the labelling
 * min(exact, max(1, C - floor(ds / 3))) is valid for the same reason C - ds is in the 6-neighbourhood, and the L1 transform is the separable
 * one (a Chebyshev distance is not).  Away from the source there is always a corner neighbour three L1 steps further out: one label down. */
MGC_HD int mgc_radial_steps(const MgcLattice& L, int dist) { return L.ndir == 26 ? dist / 3 : dist; }

/* One scan of one tile line.  SEED 1: `in` is the residual mask (bit 6 = sink link) and the scan starts the transform
 * towards the SINK; SEED 2: `in` is the excess plane (f64) and the scan starts the transform away from the SOURCE (voxels
 * that hold excess, mgc_dt_lower_tile); SEED 0: `in` holds uint16 distances.  BWD: back to front.  FINAL: `out` is the
 * int32 label array (MGC_HINF for "no seed anywhere" and for the padding voxels of a partial tile), otherwise uint16. */
template <int AXIS, bool BWD, int SEED, int FINAL, class W> /* FINAL 2: the last scan of the distance FROM THE SOURCE -- uint16 out as for 0, and the labels
                                                               in L.height lowered to max(1, C - (distance - 1)) on the way (mgc_dt_lower_tile without a pass of its own) */
MGC_HD void mgc_dt_scan_line(W& w, const MgcLattice& L, int line, const void* in, void* out, int c_min = 0, int32_t* hout = nullptr,
                             const uint16_t* carry_in = nullptr, uint16_t* carry_out = nullptr, int carry_plane = -1)
{   /* hout (FINAL 2): where the lowered labels go -- EVERY label, lowered or not, so that the caller can swap the two arrays instead of copying
     * the exact labels aside first (HipDevT::radial_begin); nullptr: in place, only what changed.
     * Z-SLABS (AXIS 2 only; MgcSlabGroup::first_relabel_dt): the lattice is one slab of a taller volume and the scan continues the
     * neighbour slab's -- carry_in[y * dx + x] = the scan value of the plane in front of this slab's first plane (the scan's own
     * direction), or nullptr at the end of the volume; carry_out (if not nullptr) receives the value of local plane `carry_plane`:
     * what the NEXT slab's scan starts from.  A pipeline over the slabs, one uint16 plane per border and direction. */
    int C = MGC_HINF;
    if (FINAL == 2) { C = L.count[mgc_cnt_radial_c(L)]; if (C < c_min) C = MGC_HINF; }
    const int na = AXIS == 0 ? L.gx : (AXIS == 1 ? L.gy : L.gz);
    const int64_t len = AXIS == 0 ? L.dx : (AXIS == 1 ? L.dy : L.dz);
    w.lanes([&](int l) MGCW_INL {
        /* global cross coordinates of this lane's line: is it inside the volume at all? */
        int64_t cu, cv, du, dv; /* (u, v) = the two cross axes in (slow, fast) order */
        if (AXIS == 0) { cu = (int64_t)(line / L.gy) * 8 + (l >> 3); cv = (int64_t)(line % L.gy) * 8 + (l & 7); du = L.dz; dv = L.dy; }
        else if (AXIS == 1) { cu = (int64_t)(line / L.gx) * 8 + (l >> 3); cv = (int64_t)(line % L.gx) * 8 + (l & 7); du = L.dz; dv = L.dx; }
        else { cu = (int64_t)(line / L.gx) * 8 + (l >> 3); cv = (int64_t)(line % L.gx) * 8 + (l & 7); du = L.dy; dv = L.dx; }
        const bool live = cu < du && cv < dv;
        int carry = MGC_DT_INF;
        if (AXIS == 2 && carry_in && live) carry = carry_in[cu * dv + cv];
        /* tiles in groups of four: all loads of a group are issued before the first value is needed (a wave has 4 KiB in
         * flight; the scan itself is a register chain) */
        for (int s0 = 0; s0 < na; s0 += 4) {
            int v[4][8];
            bool src[4] = {true, true, true, true};
            if (SEED == 2) { /* excess as built sits in the tiles k_build marked (MGC_ST_SOURCE, what mgc_dt_cmin_tile goes by): the others' 4 KiB stay where they are */
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int s = s0 + g < na ? s0 + g : na - 1;
                    src[g] = mgc_source_tile(L, mgc_dt_tile<AXIS>(L, line, BWD ? na - 1 - s : s)); /* (uniform over the wave) */
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int s = s0 + g < na ? s0 + g : na - 1; /* (a short last group re-reads the last tile) */
                const int a = BWD ? na - 1 - s : s;
                const int64_t base = (int64_t)mgc_dt_tile<AXIS>(L, line, a) * MGC_TV;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int loc = mgc_dt_loc<AXIS>(l, i);
                    if (SEED == 1) v[g][i] = (((const uint8_t*)in)[base + loc] & MGC_MASK_SINK) ? 1 : MGC_DT_INF;
                    else if (SEED == 2) v[g][i] = src[g] && ((const double*)in)[base + loc] > 0.0 ? 1 : MGC_DT_INF;
                    else v[g][i] = ((const uint16_t*)in)[base + loc];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (s0 + g >= na) break;
                const int a = BWD ? na - 1 - (s0 + g) : s0 + g;
                const int64_t base = (int64_t)mgc_dt_tile<AXIS>(L, line, a) * MGC_TV;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = BWD ? 7 - k : k;
                    const bool inside = live && (int64_t)a * 8 + i < len;
                    const int c1 = carry < MGC_DT_INF ? carry + 1 : MGC_DT_INF;
                    int val = v[g][i] < c1 ? v[g][i] : c1;
                    if (!inside) val = MGC_DT_INF; /* padding: never a seed, never a relay (the scan has left the volume) */
                    else carry = val;
                    v[g][i] = val;
                    if (AXIS == 2 && carry_out && inside && a * 8 + i == carry_plane) carry_out[cu * dv + cv] = (uint16_t)val;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int loc = mgc_dt_loc<AXIS>(l, i);
                    if (FINAL == 1) ((int32_t*)out)[base + loc] = v[g][i] < MGC_DT_INF ? v[g][i] : MGC_HINF;
                    else ((uint16_t*)out)[base + loc] = (uint16_t)v[g][i];
                    if (FINAL == 2 && (hout || (C < MGC_HINF && v[g][i] < MGC_DT_INF))) {
                        const int hv = L.height[base + loc];
                        int nv = hv;
                        if (C < MGC_HINF && v[g][i] < MGC_DT_INF) {
                            int gl = C - mgc_radial_steps(L, v[g][i] - 1);
                            gl = gl < 1 ? 1 : gl;
                            if (hv < MGC_HINF && gl < hv) nv = gl;
                        }
                        if (hout) hout[base + loc] = nv;
                        else if (nv != hv) L.height[base + loc] = nv;
                    }
                }
            }
        }
    });
}

/* After the transform: what a relabel pass leaves behind besides the labels -- per tile the faces through which a label is
 * supported from next door (incremental relabel, mgc_suspect_tile) and the ALLINF flag.  One wave per tile; every arc
 * inside the volume is residual, so "face f supports a label" = some voxel on f is one above its neighbour across f. */
template <class W>
MGC_HD void mgc_dt_finish_tile(W& w, const MgcLattice& L, int tile)
{
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    const int32_t* const t_height = L.height + (int64_t)tile * MGC_TV;
    /* one trip to HBM: per face the lane's own face voxel and the voxel it touches next door */
    typename W::template Reg<int, 6> own, oth;
    w.lanes([&](int l) MGCW_INL {
        mgcw_static_for<6>([&](auto FF) MGCW_INL {
            constexpr int F = decltype(FF)::value;
            const int nt = mgc_tile_nbr(L, tz, ty, tx, F);
            own(l, F) = w.ld(t_height, mgc_face_voxel(F, l));
            oth(l, F) = MGC_HINF;
            if (nt >= 0) oth(l, F) = w.ld(L.height + (int64_t)nt * MGC_TV, mgc_face_voxel(F ^ 1, l));
        });
    });
    uint32_t dep = 0;
    mgcw_static_for<6>([&](auto FF) MGCW_INL {
        constexpr int F = decltype(FF)::value;
        if (w.any([&](int l) MGCW_INL -> bool { return own(l, F) < MGC_HINF && oth(l, F) + 1 == own(l, F); })) dep |= 1u << F;
    });
    /* the voxel (0, 0, 0) of a tile is always inside the volume; it lies on face 0 */
    const bool finite = w.any([&](int l) MGCW_INL -> bool { return own(l, 0) < MGC_HINF; });
    w.lanes([&](int l) MGCW_INL {
        if (l == 0) L.status[tile] = (L.status[tile] & ~((63u << MGC_ST_DEP_SHIFT) | MGC_ST_ALLINF)) | (dep << MGC_ST_DEP_SHIFT) | (finite ? 0u : MGC_ST_ALLINF);
    });
}

/* Z-slabs: after a transform that every slab ran over its ghost layers too, both sides of a border hold the same labels without
 * any having travelled -- the shadow of "what the neighbour holds" (mgc_halo_pack_tile) is brought in line.  i = tile index inside
 * the layer; one wave per border tile and side. */
template <class W>
MGC_HD void mgc_shadow_sync_tile(W& w, const MgcLattice& L, int side, int i, const int32_t* height)
{
    if (!L.hshadow[side] || (side == 0 ? L.tz_own_lo == 0 : L.tz_own_hi == L.gz)) return;
    const int layer = side ? L.tz_own_hi - 1 : L.tz_own_lo, f = side ? 5 : 4;
    const int64_t tile = (int64_t)layer * L.gy * L.gx + i;
    w.lanes([&](int l) MGCW_INL { L.hshadow[side][(int64_t)i * MGC_TF + l] = height[tile * MGC_TV + mgc_face_voxel(f, l)]; });
}

/* ---------------------------------------------------------------------------------------------------------------------
 * RADIAL LABELS for the flood phase of a solve (round 5).
 *
 * Exact distance labels send the excess of a source along the SHORTEST paths to the sink -- on a marker-seeded volume
 * (a compact source, the sink on the volume's faces) six axis-aligned beams.  Where the minimum cut is a closed surface of
 * weak arcs around the source (the usual picture: an object with a strong edge) the beams saturate the part of the surface
 * they hit, and the rest is saturated by excess that spills sideways row by row as local relabels lift it: the front moves
 * a few voxels per tile visit, and a 512^3 volume needs nine cycles of sixteen colour phases to close the surface.
 *
 * Any labelling d with d(u) <= d(v) + 1 on residual arcs and d <= 1 on sink-linked voxels is VALID for push-relabel
 * (Goldberg & Tarjan 1988; the exact distances are merely the largest valid labelling).  With ds(u) = the lattice (L1)
 * distance of u from the nearest voxel that holds excess, h(u) = C - ds(u) is 1-Lipschitz along EVERY lattice arc, residual
 * or not, so
 *                      d(u) = min( exact(u), max(1, C - ds(u)) )
 * is valid for any constant C, at any time of a solve.  With C = the hop length of the shortest source -> sink path the
 * second term is the smaller one wherever a voxel does not lie on such a path: every step AWAY from the source is one
 * label down, and the excess floods outwards in all directions at a tile per colour phase until it meets arcs it cannot
 * pass.  The schedule (mgc_driver.inl) keeps the labels radial while excess of the source can still reach the sink and goes back
 * to the exact labels once the source is sealed in.  512^3 headline volume in the host simulator: 152 -> 56 colour phases,
 * 881 k -> 616 k tile discharges, 377 -> 113 relabel passes; labels unchanged (the maximum preflow differs, the set of voxels
 * that can reach the sink does not).
 * ------------------------------------------------------------------------------------------------------------------- */

/* C = hop length of the shortest source -> sink path = the smallest EXACT label a voxel that holds excess carries (the labels count
 * the hop into the sink, as the distance from the source counts its seed): known as soon as the transform towards the sink is done,
 * from the few tiles that hold a source link -- so that the last scan of the transform away from the source can lower the labels
 * on its way.  One wave per tile; atomicMin into counter slot MGC_CNT_RADIAL_C (the host presets it to MGC_HINF). */
template <class W>
MGC_HD void mgc_dt_cmin_tile(W& w, const MgcLattice& L, int tile)
{
    if (!mgc_source_tile(L, tile)) return;
    typename W::template Reg<int, 1> best;
    w.lanes([&](int l) MGCW_INL {
        int b = MGC_HINF;
        for (int k = 0; k < 8; ++k) {
            const int64_t i = (int64_t)tile * MGC_TV + k * 64 + l;
            const int hv = L.height[i];
            if (L.excess[i] > 0.0 && hv < b) b = hv;
        }
        best(l, 0) = b;
    });
    w.lanes([&](int l) MGCW_INL {
        if (best(l, 0) < MGC_HINF) w.atomic_min(&L.count[mgc_cnt_radial_c(L)], best(l, 0));
    });
}

/* labels of one tile lowered to max(1, C - (ds - 1)) where that is below what the tile holds; C is read from the counter
 * block (no host round trip between the transform and this pass); nothing happens below `c_min` (sources next to sinks:
 * the exact labels are short already, and a radial field of that height guides nothing) */
template <class W>
MGC_HD void mgc_dt_lower_tile(W& w, const MgcLattice& L, int tile, const uint16_t* ds, int c_min)
{
    const int C = L.count[mgc_cnt_radial_c(L)];
    if (C >= MGC_HINF || C < c_min) return;
    w.lanes([&](int l) MGCW_INL {
        for (int k = 0; k < 8; ++k) {
            const int64_t i = (int64_t)tile * MGC_TV + k * 64 + l;
            const int hv = L.height[i];
            const int d = (int)ds[i];
            if (hv >= MGC_HINF || d >= MGC_DT_INF) continue;
            int g = C - mgc_radial_steps(L, d - 1);
            g = g < 1 ? 1 : g;
            if (g < hv) L.height[i] = g;
        }
    });
}

/* does excess of a SOURCE still stand under a finite label?  (tiles whose status says "held a source link when the graph was
 * built": MGC_ST_SOURCE)  Counts such tiles into MGC_CNT_SOURCE_OPEN. */
template <class W>
MGC_HD void mgc_source_open_tile(W& w, const MgcLattice& L, int tile)
{
    if (!mgc_source_tile(L, tile) || (L.status[tile] & MGC_ST_ALLINF) || !mgc_owned(L, tile)) return; /* (a ghost tile's excess is as built for ever: its owner answers) */
    const bool open = w.any([&](int l) MGCW_INL -> bool {
        bool o = false;
        for (int k = 0; k < 8; ++k) {
            const int64_t i = (int64_t)tile * MGC_TV + k * 64 + l;
            o = o || (L.excess[i] > 0.0 && L.height[i] < MGC_HINF);
        }
        return o;
    });
    w.lanes([&](int l) MGCW_INL {
        if (l == 0 && open) w.atomic_add(&L.count[mgc_cnt_source_open(L)], 1);
    });
}

#endif /* MGC_DT_OPS_INL */
