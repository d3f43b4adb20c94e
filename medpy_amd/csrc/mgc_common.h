/*
 * mgc_common.h -- lattice layout shared by the HIP kernels, the host driver and the
 * host-side simulator used by the CPU tests (tests/hostsim).
 *
 * The reference keeps an explicit adjacency-list graph (lib/maxflow/src/graph.h:290-315:
 * 48 B per node + 2 x 32 B per edge = 240 B/voxel at 6-connectivity).  Here the voxel
 * lattice is implicit: per voxel six residual capacities (f64), excess, residual sink
 * capacity and a distance label, stored TILE-MAJOR: the volume is cut into 8x8x8 tiles and
 * every per-voxel field of one tile is contiguous (512 elements), so a workgroup loads its
 * whole tile with perfectly coalesced 4 KiB streams and keeps it in registers/LDS for many
 * push/relabel sweeps.
 *
 * Axis naming: the logical array has shape (D0, D1, D2) in C order (D2 fastest), node id =
 * (i0*D1 + i1)*D2 + i2 as the reference numbers voxels (energy_voxel.py:667-677).
 * Internally x = axis 2, y = axis 1, z = axis 0.
 */
#ifndef MGC_COMMON_H
#define MGC_COMMON_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MGC_HD __host__ __device__ __forceinline__
#else
#define MGC_HD inline
#endif

#define MGC_T 8                 /* tile edge                              */
#define MGC_TV 512              /* voxels per tile                        */
#define MGC_TF 64               /* voxels per tile face                   */
#define MGC_HINF 0x3f3f3f3f     /* "cannot reach the sink"; memset-able   */

/* direction d: 0 = -x, 1 = +x, 2 = -y, 3 = +y, 4 = -z, 5 = +z ; opposite = d ^ 1 ; axis = d >> 1 */
#define MGC_NDIR 6
#define MGC_MASK_SINK 0x40      /* rmask bit 6: residual capacity to the sink > 0 */
#define MGC_NCOUNT 32
#define MGC_ST_SETTLED 1u      /* (full neighbourhood) every voxel of the tile that has a residual arc at all stands at label 1 or 2, i.e. as low as a voxel
                                  with / without a sink link can ever stand: no relabel pass can lower anything here, so a pass that is sent to the tile
                                  returns at once and nobody needs to send one.  Set by the relabel visit that found it so (together with the support
                                  bits, which stay valid: labels only come down during a relabel); cleared wherever labels are reset (fill_heights_inf,
                                  reset of the suspect tiles -- a discharge that raises a label makes its tile DIRTY, hence suspect) */
#define MGC_ST_SINK 2u
#define MGC_ST_DIRTY 4u
#define MGC_ST_SUSPECT 8u
#define MGC_ST_EXCESS 16u      /* some voxel of the tile holds excess under a finite label, as of build / absorb / its last discharge */
#define MGC_ST_ALLINF 32u      /* every label of the tile is MGC_HINF: set when an incremental relabel resets the tile, cleared when a
                                  relabel pass lowers one of its labels (a clear bit promises nothing) */
#define MGC_ST_SOURCE 64u      /* (6-neighbourhood) the tile held a source link (tr_cap > 0) when the graph was built: where the schedule looks for
                                  "can excess of the source still reach the sink?" (mgc_source_open_tile) */
/* A "wall": a tile in which at least MGC_WALL_VOXELS voxels hold a WEAK forward n-link -- the patch of a closed surface of weak arcs (an
 * intensity edge under an exponential / power term: exp(-44) ~ 1e-19 on the headline volume) that crosses the tile.  Weak = g(.) below
 * MGC_WALL_WEIGHT, judged BEFORE the division by the voxel spacing: every boundary term of the reference takes values in (0, 1]
 * (energy_voxel.py:99-114, 174-189, 226-236, 290-300, 337-345, 399-407, 444-452, 506-514), so the threshold is relative to the largest
 * weight the term can produce and does not move with the units of the spacing.  A surface of arcs that weak SEALS: what crosses it is
 * nothing next to the excess behind it, and the flood phase on radial labels (mgc_dt_ops.inl) is what saturates it fastest.  An edge
 * under a linear or division term is never weaker than ~1 / (1 + range / sigma): it leaks, the cut lies elsewhere or carries a
 * flow of the order of the weights, and exact labels are the better guide (measured: profiles/r6_radial_on_off.jsonl).  Isolated weak
 * arcs of a noisy image do not qualify (~1 per tile).  mgc_build counts the wall tiles. */
#define MGC_WALL_WEIGHT 9.313225746154785e-10 /* 2^-30 */
#define MGC_WALL_VOXELS 24                    /* of the 64+ pairs a surface cuts in a tile it crosses (a sealing wall: nearly all of them weak; the edge of the
                                                 weak-contrast volume: 3 %, a dozen tiles of 262 144 reached 12 and flipped the rule) */
#define MGC_ST_DEP_SHIFT 8
/* counter slots no layout uses as a work list (6-neighbourhood: lists 0..7, totals 8 / 9; 26-neighbourhood: lists 0..17,
 * totals 18..20; tickets of the wave kernels 24..27) */
#define MGC_CNT_SINK_TILES 13  /* (6-neighbourhood) k_build: tiles that hold a sink link */
#define MGC_CNT_WALL_TILES 12  /* (6-neighbourhood) k_build: tiles a surface of weak arcs passes through (MGC_WALL_*), read by mgc_build */
#define MGC_CNT_RADIAL_C 14    /* (6-neighbourhood) hop length of the shortest source -> sink path (mgc_dt_cmin_tile) or MGC_HINF */
#define MGC_CNT_SOURCE_OPEN 15 /* (6-neighbourhood) source tiles whose excess still stands under a finite label (mgc_source_open_tile) */
/* (the full neighbourhood's lists take slots 0 .. 17: its two radial-label counters live in the ticket slots of the wave relabel kernel, which it has not) */
#define MGC26_CNT_RADIAL_C 26
#define MGC26_CNT_SOURCE_OPEN 27
#define MGC_CNT_CHANGED 21     /* suspect-closure pass changed something */
#define MGC_CNT_FILTER 22      /* length of the scratch list the tile filters fill (absorb / relabel seeding / suspect reset) */
#define MGC_CNT_FILTER_ACT 23  /* ... of the activation filter */
/* the filters alternate between two slots each: the kernel that consumes one list clears the slot the next filter will count
 * into (no launch in between just to clear a word) */
#define MGC_CNT_FILTER_B 30
#define MGC_CNT_FILTER_ACT_B 31
#define MGC_CNT_DEFERRED 28    /* (during a solve; k_build's MGC_CNT_NOT_FULL is read before it starts) border tiles a halo pack had to leave
                                  for the next exchange because the message was full (MgcLattice::halo_max_rec) */
#define MGC_CNT_WAVE_TILES 29  /* running total of the tiles k_discharge_w visited (count[8] pools both discharge kernels) */
#define MGC_CNT_NOT_FULL 28    /* k_build: tiles holding an n-link inside the volume that is not residual (0: the first global relabel is a distance transform) */

/* *p = v as a STREAMING store on the GPU: the write-back of a tile's own state by a discharge (excess, residual planes, masks) is not read again before the
 * caches have turned over many times; written through them it only pushes out what the visits in flight are about to read (measured in round 6:
 * profiles/r6_ab_nontemporal.jsonl).  Labels and outboxes, which neighbours read in the very next phase, stay ordinary stores. */
#if defined(__HIP_DEVICE_COMPILE__)
#define MGC_STORE_STREAM(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MGC_STORE_STREAM(p, v) (*(p) = (v))
#endif

struct MgcLattice {
    /* logical volume */
    int64_t dz, dy, dx;       /* D0, D1, D2                                         */
    int64_t nvox;
    /* tile grid */
    int gz, gy, gx;           /* ceil(D / 8)                                        */
    int ntiles;
    /* Z-slab decomposition (multi-GPU): this lattice is one slab of a taller volume.  Tile layers
       [tz_own_lo, tz_own_hi) are owned; a layer below / above is a GHOST layer mirroring the
       neighbour slab's border tiles (their labels and their outboxes towards us).  Single GPU:
       tz_own_lo = 0, tz_own_hi = gz, tz_global0 = 0. */
    int tz_own_lo, tz_own_hi;
    int tz_global0;           /* global tile-layer index of local layer 0 (checkerboard colour) */
    /* per-tile state, tile-major */
    double*   rcap;           /* [ntiles][6][512] residual n-link capacities        */
    double*   cap0;           /* [ntiles][6][512] capacities as built (cut value, getters) */
    double*   excess;         /* [ntiles][512]                                      */
    double*   sink;           /* [ntiles][512] residual capacity voxel -> sink      */
    int32_t*  height;         /* [ntiles][512] distance label, MGC_HINF = unreachable */
    uint8_t*  rmask;          /* [ntiles][512] bit d: rcap[d] > 0 ; bit 6: sink > 0 */
    uint32_t* rmask32;        /* 26-neighbourhood: [ntiles][512] bit d (0..25): rcap[d] > 0 ; bit 26: sink > 0 */
    int       ndir;           /* 6 or 26: rcap / cap0 hold ndir planes per tile      */
    double*   obox;           /* [ntiles][6][64] flow pushed across face f, not yet absorbed by the neighbour */
    uint32_t* oflags;         /* [ntiles] bit f: obox[f] holds something            */
    /* work lists: [0],[1] = discharge lists of colour 0 / 1 being consumed; [2],[3] = being produced;
       [4],[5] = relabel list consumed / produced */
    int32_t*  list[20];       /* 26-neighbourhood: [0..15] discharge lists (target phase & 15), [16],[17] relabel.  Every list is
                                 cut into `nshard` regions of `shard_cap` entries (see scount) */
    int32_t*  count;          /* [MGC_NCOUNT] device resident: [0..5] list lengths, [6] active tiles found by the last
                                 activation, [8]/[9] running totals of tiles discharged / relabelled            */
    /* List lengths, optionally SHARDED.  A list counter is the hottest word of the solver (every tile visit appends a few
       neighbours, a relabel pass appends ~9 000 tiles in 30 us, and one address takes ~88 returning atomics per microsecond,
       MI355X_MICROARCH.md "dequeue"), so the layout allows the length of list / counter slot c to be the SUM of nshard words
       scount[c * nshard + s] -- see MGC_NSHARD for what that measured.  With several regions an appender uses the
       shard its workgroup id selects (spread over the XCDs) and writes into region s of the list,
       list[l][s * shard_cap + pos].  Consumers read the nshard words once (MgcListView) and walk the regions as one
       sequence.  The host simulator runs with nshard = 1 and scount = count: the plain layout. */
    int32_t*  scount;         /* [MGC_NCOUNT][nshard] */
    int       nshard;         /* 1 or MGC_NSHARD */
    int       shard_cap;      /* entries per region = ntiles (a list holds a tile at most once) */
    uint32_t* stamp;          /* [ntiles] de-duplication stamp for list appends (discharge) */
    uint32_t* rstamp;         /* [ntiles] same for the relabel lists                */
    uint32_t* status;         /* [ntiles] bit1 (2): the tile holds a residual arc to the sink; bit2 (4): DIRTY = discharged
                                 since the last global relabel; bit3 (8): SUSPECT (labels must be recomputed);
                                 bits 8..13: faces through which the tile's labels are supported by a neighbour */
    int       halo_max_rec;   /* slabs: records one compacted border message may carry.  A message is the fixed header plus this many
                                 record slots, sent in ONE transfer whose size both sides know without asking the device; a
                                 border tile that finds the message full keeps what it has to say (labels: the shadow stays
                                 behind; flow: it stays in the outbox / ghost tile) and goes out with the next exchange.
                                 MGC_CNT_DEFERRED counts those, and the schedule does not take "nothing woke up" for a
                                 fixpoint, nor start a global relabel, while any are left (slab.py, mgc_solve_slab) */
    int32_t*  hshadow[2];     /* slabs, 6-neighbourhood: [gy*gx][64] labels of the owned border layer (lower / upper) as the neighbour
                                 slab last received them -- a border tile only travels when it differs from this (or holds flow) */
    unsigned long long* prof; /* optional [16] cycle accumulators of the discharge sections (development aid) or NULL */
    const uint8_t* tsrc;      /* full neighbourhood: [ntiles] bit 0 = the tile held a source link when the graph was built (what status bit MGC_ST_SOURCE says in
                                 the 6-neighbourhood, whose status words have room for it), or NULL */
};

/* Measured on MI355X (round 3): 16 regions per list do NOT pay -- 512^3 sphere 38.6 ms with 16, 37.6 ms with 1; 26-neighbourhood
 * 256^3 82.8 vs 76.7 ms: the appends of a launch are spread over its whole duration and over two lists, the extra trip for
 * the region lengths costs more than the queueing it avoids.  One region is the default; the machinery stays (tests and
 * tools can build with -DMGC_NSHARD=16 and switch with the list_shards parameter). */
#ifndef MGC_NSHARD
#define MGC_NSHARD 1
#endif

MGC_HD int mgc_cnt_radial_c(const MgcLattice& L) { return L.ndir == 26 ? MGC26_CNT_RADIAL_C : MGC_CNT_RADIAL_C; }
MGC_HD int mgc_cnt_source_open(const MgcLattice& L) { return L.ndir == 26 ? MGC26_CNT_SOURCE_OPEN : MGC_CNT_SOURCE_OPEN; }

MGC_HD int32_t* mgc_counter(const MgcLattice& L, int c, int shard) { return L.scount + c * L.nshard + shard; }

/* the regions of one list as one sequence: pre[s] = entries in the regions before s (pre[k] = n for k >= nshard) */
struct MgcListView {
    int n;
    int pre[MGC_NSHARD + 1];
};

MGC_HD int mgc_list_view(const MgcLattice& L, int c, MgcListView& v)
{
    int acc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < MGC_NSHARD; ++s) {
        v.pre[s] = acc;
        if (s < L.nshard) acc += L.scount[c * L.nshard + s];
    }
    v.pre[MGC_NSHARD] = acc;
    return v.n = acc;
}

/* entry i (0 <= i < v.n) of list l whose length lives in the counter slot the view was taken from */
MGC_HD int mgc_list_at(const MgcLattice& L, int l, const MgcListView& v, int i)
{
    int s = 0, base = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 1; k < MGC_NSHARD; ++k)
        if (i >= v.pre[k] && v.pre[k + 1] > v.pre[k]) { s = k; base = v.pre[k]; }
    return L.list[l][(int64_t)s * L.shard_cap + (i - base)];
}

/* did the tile hold a source link (tr_cap > 0) when the graph was built? */
MGC_HD bool mgc_source_tile(const MgcLattice& L, int tile) { return L.tsrc ? (L.tsrc[tile] & 1u) != 0 : (L.status[tile] & MGC_ST_SOURCE) != 0u; }

MGC_HD int mgc_tile_id(const MgcLattice& L, int tz, int ty, int tx) { return (tz * L.gy + ty) * L.gx + tx; }

MGC_HD void mgc_tile_coords(const MgcLattice& L, int tile, int& tz, int& ty, int& tx)
{
    tx = tile % L.gx;
    int r = tile / L.gx;
    ty = r % L.gy;
    tz = r / L.gy;
}

/* neighbour tile across face f, or -1 outside the grid */
MGC_HD int mgc_tile_nbr(const MgcLattice& L, int tz, int ty, int tx, int f)
{
    switch (f) {
    case 0: return tx > 0 ? mgc_tile_id(L, tz, ty, tx - 1) : -1;
    case 1: return tx + 1 < L.gx ? mgc_tile_id(L, tz, ty, tx + 1) : -1;
    case 2: return ty > 0 ? mgc_tile_id(L, tz, ty - 1, tx) : -1;
    case 3: return ty + 1 < L.gy ? mgc_tile_id(L, tz, ty + 1, tx) : -1;
    case 4: return tz > 0 ? mgc_tile_id(L, tz - 1, ty, tx) : -1;
    default: return tz + 1 < L.gz ? mgc_tile_id(L, tz + 1, ty, tx) : -1;
    }
}

MGC_HD int mgc_tile_colour(const MgcLattice& L, int tz, int ty, int tx) { return (tz + L.tz_global0 + ty + tx) & 1; }

/* is the tile owned by this slab (as opposed to a ghost mirror of the neighbour slab's tile)? */
MGC_HD bool mgc_owned(const MgcLattice& L, int tile)
{
    const int tz = tile / (L.gy * L.gx);
    return tz >= L.tz_own_lo && tz < L.tz_own_hi;
}

/* local voxel index inside a tile */
MGC_HD int mgc_local(int z, int y, int x) { return (z * MGC_T + y) * MGC_T + x; }

/* index of local voxel (z,y,x) on the face normal to axis a (a = 0:x, 1:y, 2:z) */
MGC_HD int mgc_face_index(int a, int z, int y, int x)
{
    return a == 0 ? (z * MGC_T + y) : (a == 1 ? (z * MGC_T + x) : (y * MGC_T + x));
}

/* local index of the k-th voxel of the face of direction f (the layer touching that face) */
MGC_HD int mgc_face_voxel(int f, int k)
{
    const int a = f >> 1, u = k >> 3, v = k & 7, w = (f & 1) ? (MGC_T - 1) : 0;
    return a == 0 ? mgc_local(u, v, w) : (a == 1 ? mgc_local(u, w, v) : mgc_local(w, u, v));
}

/* C-order node id -> (tile, local) and back */
MGC_HD void mgc_node_to_tile(const MgcLattice& L, int64_t id, int& tile, int& loc)
{
    const int64_t x = id % L.dx;
    const int64_t r = id / L.dx;
    const int64_t y = r % L.dy;
    const int64_t z = r / L.dy;
    tile = mgc_tile_id(L, (int)(z >> 3), (int)(y >> 3), (int)(x >> 3));
    loc = mgc_local((int)(z & 7), (int)(y & 7), (int)(x & 7));
}

/* Z-slab split of a volume with D0 planes over `nranks` slabs at tile-layer granularity (shared by the
 * HIP library and the host simulator so both cut the volume identically). */
struct MgcSlabSpec {
    int rank, nranks;
    int own_lo, own_hi;     /* local tile layers owned: [own_lo, own_hi)            */
    int tz_global0;         /* global tile-layer index of local layer 0             */
    int64_t plane0, plane1; /* global plane range held locally (ghost planes incl.) */
    int64_t own0, own1;     /* global plane range owned                             */
};

static inline int mgc_slab_spec(int64_t d0, int rank, int nranks, MgcSlabSpec* sp)
{
    const int64_t GZ = (d0 + 7) / 8;
    if (nranks < 1 || rank < 0 || rank >= nranks || nranks > GZ) return 1;
    const int64_t t0 = rank * GZ / nranks, t1 = (rank + 1) * GZ / nranks;
    const int has_lo = rank > 0, has_hi = rank + 1 < nranks;
    sp->rank = rank; sp->nranks = nranks;
    sp->own_lo = has_lo; sp->own_hi = has_lo + (int)(t1 - t0);
    sp->tz_global0 = (int)(t0 - has_lo);
    sp->plane0 = (t0 - has_lo) * 8;
    sp->plane1 = (t1 + has_hi) * 8 < d0 ? (t1 + has_hi) * 8 : d0;
    sp->own0 = t0 * 8;
    sp->own1 = t1 * 8 < d0 ? t1 * 8 : d0;
    return 0;
}

#endif /* MGC_COMMON_H */
