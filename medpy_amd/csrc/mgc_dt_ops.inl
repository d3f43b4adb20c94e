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

/* One scan of one tile line.  SEED: `in` is the residual mask (bit 6 = sink link) and the scan starts the transform;
 * otherwise `in` holds uint16 distances.  BWD: back to front.  FINAL: `out` is the int32 label array (MGC_HINF for
 * "no sink anywhere" and for the padding voxels of a partial tile), otherwise uint16. */
template <int AXIS, bool BWD, bool SEED, bool FINAL, class W>
MGC_HD void mgc_dt_scan_line(W& w, const MgcLattice& L, int line, const void* in, void* out)
{
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
        /* tiles in groups of four: all loads of a group are issued before the first value is needed (a wave has 4 KiB in
         * flight; the scan itself is a register chain) */
        for (int s0 = 0; s0 < na; s0 += 4) {
            int v[4][8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int s = s0 + g < na ? s0 + g : na - 1; /* (a short last group re-reads the last tile) */
                const int a = BWD ? na - 1 - s : s;
                const int64_t base = (int64_t)mgc_dt_tile<AXIS>(L, line, a) * MGC_TV;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int loc = mgc_dt_loc<AXIS>(l, i);
                    if (SEED) v[g][i] = (((const uint8_t*)in)[base + loc] & MGC_MASK_SINK) ? 1 : MGC_DT_INF;
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
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int loc = mgc_dt_loc<AXIS>(l, i);
                    if (FINAL) ((int32_t*)out)[base + loc] = v[g][i] < MGC_DT_INF ? v[g][i] : MGC_HINF;
                    else ((uint16_t*)out)[base + loc] = (uint16_t)v[g][i];
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

#endif /* MGC_DT_OPS_INL */
