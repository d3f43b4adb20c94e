/*
 * medpy_hip.h -- C ABI of libmedpyhip.so: MI355X (gfx950) voxel graph-cut.
 *
 * This is the drop-in boundary for the hot path
 *     medpy.graphcut.graph_from_voxels  (reference medpy/graphcut/generate.py:33-174)
 *   + medpy.graphcut.energy_voxel.*     (reference medpy/graphcut/energy_voxel.py:33-664)
 *   + maxflow.GraphDouble               (reference lib/maxflow/src/wrapper.cpp:59-89 binding of
 *                                        lib/maxflow/src/graph.h / maxflow.cpp)
 * Plain C types only; a handle owns all device memory; every entry point returns an
 * status code (never exit()s, unlike graph.cpp:22,71,95).  The Python host layer
 * (medpy_amd/graphcut) binds these with ctypes and presents the reference's API;
 * INTEGRATION.md shows the binding a MedPy maintainer would add.
 *
 * Node ids are C-order flat indices of the logical array shape, as in the reference
 * (energy_voxel.py:667-677, generate.py:170-172).  All host arrays handed over must be
 * C-contiguous; they are copied into HBM inside the call and may be freed on return.
 */
#ifndef MEDPY_HIP_H
#define MEDPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgc_graph* mgc_handle;

typedef enum mgc_status {
    MGC_OK = 0,
    MGC_ERR_INVALID = 1,     /* bad argument (shape, dtype, id out of range ...)          */
    MGC_ERR_NO_DEVICE = 2,   /* no usable gfx950 device: the library never falls back     */
    MGC_ERR_HIP = 3,         /* a HIP runtime call failed; see mgc_last_error             */
    MGC_ERR_OOM = 4,
    MGC_ERR_STATE = 5,       /* call order violated (e.g. maxflow before build)           */
    MGC_ERR_UNSUPPORTED = 6, /* valid request outside the implemented path                */
    MGC_ERR_NOT_CONVERGED = 7
} mgc_status;

/* boundary terms, reference energy_voxel.py:68-516 */
typedef enum mgc_term {
    MGC_TERM_NONE = 0,
    MGC_TERM_DIFFERENCE_LINEAR = 1,      /* energy_voxel.py:117-191 */
    MGC_TERM_DIFFERENCE_EXPONENTIAL = 2, /* energy_voxel.py:241-302 */
    MGC_TERM_DIFFERENCE_DIVISION = 3,    /* energy_voxel.py:350-409 */
    MGC_TERM_DIFFERENCE_POWER = 4,       /* energy_voxel.py:455-516 */
    MGC_TERM_MAXIMUM_LINEAR = 5,         /* energy_voxel.py:68-114  */
    MGC_TERM_MAXIMUM_EXPONENTIAL = 6,    /* energy_voxel.py:194-238 */
    MGC_TERM_MAXIMUM_DIVISION = 7,       /* energy_voxel.py:305-347 (uses the difference skeleton, :347) */
    MGC_TERM_MAXIMUM_POWER = 8           /* energy_voxel.py:412-452 */
} mgc_term;

typedef enum mgc_dtype {
    MGC_U8 = 0, MGC_I8 = 1, MGC_U16 = 2, MGC_I16 = 3, MGC_U32 = 4, MGC_I32 = 5,
    MGC_U64 = 6, MGC_I64 = 7, MGC_F32 = 8, MGC_F64 = 9
} mgc_dtype;

/* termtype of the reference, lib/maxflow/src/graph.h:57-61 */
enum { MGC_SOURCE = 0, MGC_SINK = 1 };

typedef struct mgc_stats {
    double  build_ms;          /* device time of the last mgc_build                        */
    double  solve_ms;          /* device time of the last mgc_maxflow                      */
    double  discharge_ms;      /* ... of which tile-discharge kernels (HIP events)         */
    double  relabel_ms;        /* ... of which global-relabel kernels                      */
    int64_t discharge_launches;
    int64_t relabel_launches;
    int64_t discharge_tiles;   /* tile discharges executed                                 */
    int64_t relabel_tiles;     /* tile relabels executed                                   */
    int64_t global_relabels;
    int64_t phases;
    int64_t ntiles;
    int64_t nvox;
    int64_t device_bytes;      /* HBM held by the handle                                   */
    int64_t reserved[3];       /* [0]: counter read-backs (host syncs) of the solve; [1]: cycles of colour phases that ran on radial labels */
    /* the dominant kernel by itself: discharge_ms / _launches / _tiles pool the one-wave-per-tile kernel (k_discharge_w) and the
     * workgroup-per-tile kernel that takes the short lists; these three are k_discharge_w alone */
    double  discharge_wave_ms;
    int64_t discharge_wave_launches;
    int64_t discharge_wave_tiles;
    int64_t timing_stride;     /* every n-th solver launch of a kind carries a HIP event pair; the _ms are their mean x launches */
} mgc_stats;

/* Invariants of a maximum preflow, checked on the device (mgc_validate).  The reference has the same idea as a debugging
 * aid: Graph::test_consistency, lib/maxflow/src/maxflow.cpp:610-682.  It is the only check available for volumes no CPU
 * oracle can reach (BASELINE.json configs 4 and 5).  Counts are over the OWNED voxels of the handle (a slab: its planes);
 * everything must be zero, the two errors of the order of rounding, and the two flow values equal -- summed over the
 * ranks first when the volume is cut into slabs. */
typedef struct mgc_validation {
    int64_t voxels;                 /* owned voxels looked at                                                     */
    int64_t negative_values;        /* a residual, an excess or a sink link below zero                            */
    int64_t active_excess;          /* excess on a voxel that can still reach the sink: the preflow is not maximum */
    int64_t residual_arcs_across;   /* residual arc from a voxel that cannot reach the sink to one that can       */
    int64_t sink_links_across;      /* residual sink link on a voxel labelled "cannot reach the sink"             */
    int64_t pair_violations;        /* rcap(u,v) + rcap(v,u) != cap(u,v) + cap(v,u) beyond rounding               */
    int64_t node_violations;        /* from the source != excess + into the sink + net outflow beyond rounding    */
    int64_t pending_outbox;         /* flow pushed across a tile face and not yet absorbed (6-neighbourhood)      */
    int64_t reserved[4];
    double  max_pair_error;         /* largest relative error of the two conservation checks                      */
    double  max_node_error;
    double  flow_into_sink;         /* sum over the owned voxels of (sink link as built - residual sink link)     */
    double  cut_capacity;           /* capacity of the cut the labels define, this handle's part (without the constant) */
    double  flow_constant;          /* the part add_tweights folds into the flow (graph.h:416-425), this handle's  */
    double  sink_capacity_used;     /* sum of the built sink links that carry flow.  flow_into_sink is a sum of differences
                                       (65535 - residual) and only known to about 1e-13 of this; cut_capacity is exact */
    double  reserved_d[2];
} mgc_validation;

/* number of usable devices (0 => every other call fails with MGC_ERR_NO_DEVICE) */
int mgc_device_count(int* count);
/* free / total HBM of a device in bytes (hipMemGetInfo): callers that size work by memory -- tests that need tens of GB skip instead
 * of failing on a smaller device; no reference counterpart (the reference malloc()s and exit(1)s, graph.cpp:19-23) */
int mgc_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes);

/* Replaces GCGraph.__init__ -> GraphDouble(nodes, edges) + add_node (graph.py:294-308,
 * graph.cpp:12-31).  ndim 1..3; connectivity = 2*ndim (the only neighbourhood the reference supports,
 * generate.py:44-49) or 3^ndim - 1 (full neighbourhood: 8 in 2-D, 26 in 3-D -- an extension named by
 * BASELINE.json configs 3 and 5, weights = the same g(.) on every offset, spacing = Euclidean offset length). */
int mgc_create(int ndim, const int64_t* shape, int connectivity, int device, mgc_handle* out);
int mgc_destroy(mgc_handle h);
const char* mgc_last_error(mgc_handle h); /* h may be NULL: error of the last failed mgc_create */
/* Device memory of destroyed handles is kept in a per-device pool (blocks >= 1 MiB, by exact size, up to MEDPY_HIP_POOL_MB --
 * default an eighth of the device's memory; 0 switches the pool off) and handed to the next handle that asks for the same sizes:
 * the reference allocates per graph (graph.cpp:12-31, one graph per volume in bin/medpy_graphcut_voxel.py:163-182), and a 13 GB
 * handle per 512^3 volume is 0.5 - 0.8 s of hipMalloc / hipFree on some boxes.  An allocation that fails empties the pool first.
 * mgc_pool_trim returns everything to the driver; mgc_pool_info reports what is idle and how often the pool served a request. */
int mgc_pool_trim(int device);
int mgc_pool_info(int device, int64_t* idle_bytes, int64_t* hits, int64_t* misses);

/* Replaces boundary_<term>(graph, (image, sigma, spacing)) -> __skeleton_base
 * (energy_voxel.py:611-664).  `image` has the handle's shape.  p0: sigma for
 * exponential/division/power (for exponential pass sigma; the library squares it the way
 * math.pow does), unused for linear (the intensity range is reduced on the device).
 * spacing: ndim doubles or NULL (False). */
int mgc_set_boundary(mgc_handle h, int term, const void* image, int dtype, double sigma, const double* spacing);

/* Integer-valued images (CT, MR: uint8 / uint16 / int16, or floats that hold integers): the exponential and power terms
 * depend on the two intensities only through d = |I_p - I_q| (difference terms) or max(|I_p|, |I_q|) (maximum terms), an
 * integer below `n`.  table[d] = the boundary function of d as the REFERENCE evaluates it on the host -- NumPy's exp / pow,
 * energy_voxel.py:226-236, 290-300, 444-452, 506-513, floored at sys.float_info.min -- replaces the device's own exp / pow
 * (OCML, <= 2 ulp away): the n-link weights are then BIT-IDENTICAL to the reference's on such images.  The spacing division
 * still happens on the device (IEEE division).  Call after mgc_set_boundary (which forgets a table set earlier); n = 0
 * forgets it; n <= 65536; an intensity pair beyond the table falls back to the device's own evaluation. */
int mgc_set_boundary_lut(mgc_handle h, const double* table, int64_t n);

/* Replaces regional_probability_map(graph, (probability_map, alpha)) (energy_voxel.py:33-65)
 * -> set_tweights_all (graph.py:551-552).  dtype MGC_F32 or MGC_F64: products are evaluated in
 * that dtype, as NumPy does for the reference. */
int mgc_set_regional_probability(mgc_handle h, const void* probability_map, int dtype, double alpha);

/* Replaces set_source_nodes / set_sink_nodes over marker masks (generate.py:169-172,
 * graph.py:310-380): nonzero fg -> add_tweights(i, 65535, 0), nonzero bg -> add_tweights(i, 0, 65535). */
int mgc_set_markers(mgc_handle h, const uint8_t* fg, const uint8_t* bg);

/* After mgc_maxflow (or the slab driver's last step): see mgc_validation.  Also works on a graph whose solve was cut
 * short (MGC_ERR_NOT_CONVERGED): it then reports the excess that is still active.  MGC_ERR_STATE before the first solve
 * step of a build: the distance labels it reads do not exist yet. */
int mgc_validate(mgc_handle h, mgc_validation* out);

/* The *_linear terms divide by the intensity range of the image (energy_voxel.py:101: max |I|; 174-176:
 * |max - min| in the image's dtype).  A single handle measures it itself.  A SLAB only holds its own planes, so the
 * caller reduces the local triples {min, max, max|.|} over the ranks (min, max, max) and hands the global one back
 * before mgc_build; building a *_linear slab without it fails with MGC_ERR_STATE.  NULL forgets a range set earlier;
 * mgc_set_boundary does so too. */
int mgc_get_image_range(mgc_handle h, double* out3);
int mgc_set_image_range(mgc_handle h, const double* in3);

/* Plug-in path (user supplied energy callables drive GCGraph.set_nweight / set_tweight,
 * graph.py:382-440, 466-498).  Edges must join lattice neighbours (checked here: MGC_ERR_UNSUPPORTED
 * names the first edge that does not; arbitrary graphs go to msg_*); capacities accumulate like
 * sum_edge (graph.h:457-480): an edge given several times adds up IN CALL ORDER on top of the boundary
 * term's weight, the same floating point additions as the reference.  One batch per build (a later
 * batch replaces one that a build already applied); the batch stays with the handle, so a rebuild
 * applies it again.  t-weights: tr[n] = merged residual per node, flow_const = the
 * part add_tweights folds into the flow (graph.h:416-425); applied before the markers. */
int mgc_add_edges(mgc_handle h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev);
int mgc_set_tweights_merged(mgc_handle h, const double* tr, double flow_const);

/* Runs the n-link / t-link kernels: the residual graph is now resident in HBM. */
int mgc_build(mgc_handle h);

/* Energy read-back for parity tests: axis weights in the layout of
 * `neighbourhood_intensity_term` (energy_voxel.py:644-658); tr_cap per node (Graph::get_trcap). */
int mgc_get_nweights(mgc_handle h, int axis, double* out);
int mgc_get_tweights(mgc_handle h, double* out);
/* weight of the arc (p, p + offset) for every voxel p (handle shape), NaN where p + offset is outside; offset has
 * ndim components in {-1,0,1}.  The read-back used for the full (8 / 26) neighbourhood. */
int mgc_get_nweights_offset(mgc_handle h, const int* offset, double* out);
int mgc_get_edge(mgc_handle h, int64_t i, int64_t j, double* out); /* Graph::get_edge, graph.h:482-498 */

/* Replaces GraphDouble.maxflow() (maxflow.cpp:472-604).  flow = capacity of the minimum cut
 * found (equals the reference's return value up to summation order, ~1e-12 relative). */
int mgc_maxflow(mgc_handle h, double* flow);

/* Replaces the per-voxel what_segment loop of bin/medpy_graphcut_voxel.py:177-181:
 * out[i] = 0 if SINK == what_segment(i) else 1. */
int mgc_labels(mgc_handle h, uint8_t* out);
int mgc_what_segment(mgc_handle h, int64_t i, int* segment); /* Graph::what_segment, graph.h:561-571 */

int mgc_get_node_num(mgc_handle h, int64_t* n);
int mgc_set_param(mgc_handle h, const char* name, int64_t value); /* solver schedule knobs: the table in DESIGN.md section 3 */
int mgc_get_stats(mgc_handle h, mgc_stats* out);
/* development aid (mgc_set_param "profile_sections" 1): out16[0..3] = shader cycles of workgroup lane 0 spent in
 * load+absorb / in-tile labels / push sweeps / store of k_discharge, out16[8..11] = how many such sections */
int mgc_get_profile(mgc_handle h, uint64_t* out16);

/* ------------------------------------------------------------------------------------------
 * Z-slab decomposition across the GPUs of one node (no reference counterpart: the reference is
 * single-process; its only splitter, wrapper.py:72-204, is approximate and label-based).
 * One handle per slab.  The slab owns whole tile layers (8 voxel planes) of axis 0 and mirrors
 * one ghost tile layer per neighbour; mgc_set_* take the LOCAL sub-arrays (planes
 * info[0]..info[1] of the global arrays, ghost planes included).  The solve is mgc_solve_slabs (below): the single handle's
 * schedule with the packed borders (labels, outbox flow, suspect flags) exchanged at its hook points -- between the slabs' own
 * buffers when all slabs are handles of one process, over RCCL / xGMI inside the library (mgc_comm_init) when every rank holds
 * one, or through the caller's callbacks on host buffers (development transports).  mgc_solver_op / mgc_halo_pack / mgc_halo_unpack
 * issue single launches and single messages (profiling tools, the transport tests); no schedule is driven through them any more.
 * ---------------------------------------------------------------------------------------- */
enum {
    MGC_OP_ABSORB_ALL = 0,   /* -                                          */
    MGC_OP_FILL_INF = 1,     /* -                                          */
    MGC_OP_ZERO_COUNT = 2,   /* a0 = counter index                          */
    MGC_OP_RELABEL_ALL = 3,  /* a0 = next epoch, a1 = next list             */
    MGC_OP_RELABEL_LIST = 4, /* a0 = list, a1 = next epoch, a2 = next list  */
    MGC_OP_ACTIVATE = 5,     /* a0 = phase                                  */
    MGC_OP_DISCHARGE = 6,    /* a0 = list, a1 = phase, a2 = max cycles, a3 = max sweeps */
    MGC_OP_SUSPECT_PASS = 7, /* one pass of the tile-level suspect closure (sets counter MGC_CNT_CHANGED = 21 when something changed) */
    MGC_OP_RESET_SUSPECT = 8 /* a0 = next epoch, a1 = next list: suspect tiles -> labels INF, queued for relabelling */
};
int mgc_create_slab(int ndim, const int64_t* global_shape, int connectivity, int device, int rank, int nranks, mgc_handle* out);
/* info[0..1] = local plane range [first, last) in the global volume (ghost planes included), info[2..3] = owned
 * plane range, info[4] / info[5] = has a lower / upper neighbour, info[6] = tiles per layer */
int mgc_slab_info(mgc_handle h, int64_t* info8);
int mgc_solver_op(mgc_handle h, int op, int64_t a0, int64_t a1, int64_t a2, int64_t a3);
int mgc_read_counts(mgc_handle h, int32_t* out32); /* 32 counters */
int mgc_halo_bytes(mgc_handle h, int kind, int64_t* bytes);
/* side 0 = lower / 1 = upper slab boundary; kind 0 = labels (relabel pass), 1 = labels + outbox flow (phase),
   2 = DIRTY / SUSPECT flags of the border tiles (suspect closure of an incremental relabel) */
int mgc_halo_pack(mgc_handle h, int side, int kind, void* buf, int buf_on_device);
int mgc_halo_unpack(mgc_handle h, int side, int kind, const void* buf, int buf_on_device, uint32_t epoch, int list);
/* after the slab driver has converged: labels of the local planes + this slab's part of the cut capacity */
int mgc_finish(mgc_handle h, double* flow_partial);

/* Native transport: RCCL over xGMI (librccl is dlopen()ed on first use, single-GPU users never need it).
 * Rank 0 obtains a 128-byte id (mgc_comm_unique_id) that the launcher broadcasts out of band (bench.py: a private directory of
 * files, medpy_amd/rendezvous.py -- no PyTorch anywhere in the package); every rank then calls mgc_comm_init on its slab handle.  mgc_halo_exchange =
 * pack both borders -> grouped ncclSend/ncclRecv with rank-1 / rank+1 -> unpack, all ordered on the handle's
 * stream (no host synchronisation).  mgc_allreduce_counts sums the 32 solver counters over all ranks
 * (ncclAllReduce) and returns them: the termination / fixpoint tests of the distributed schedule. */
int mgc_comm_unique_id(uint8_t* id128);
int mgc_comm_init(mgc_handle h, const uint8_t* id128);
int mgc_halo_exchange(mgc_handle h, int kind, uint32_t epoch, int list);

/* What a distributed solve did (global numbers; per-kernel times and this slab's own counts are in mgc_get_stats afterwards, as after
 * mgc_maxflow).  The host looks at the device only where every rank has to take the same decision (reduced counters). */
typedef struct mgc_slab_stats {
    int64_t outer;            /* global relabels                                            */
    int64_t relabel_passes;
    int64_t phases;           /* colour phases                                              */
    int64_t exchanges;        /* border exchanges (each ONE grouped send/receive per neighbour) */
    int64_t reductions;       /* counter all-reduces = points where the host waits for the device */
    int64_t converged;
    int64_t discharge_tiles;  /* global                                                     */
    int64_t relabel_tiles;    /* global                                                     */
    int64_t deferred_drains;  /* extra exchanges because a border message was full          */
    int64_t reserved[7];      /* [0]: cycles of colour phases that ran on radial labels */
} mgc_slab_stats;
int mgc_solve_slab(mgc_handle h, mgc_slab_stats* out);

/* ONE entry point for every way a volume's slabs can be laid out (round 6; the schedule is the single handle's own, mgc_solve in
 * medpy_amd/csrc/mgc_driver.inl, with the borders exchanged at its hook points -- first relabel by distance transform carried
 * across the slab borders, flood phase on radial labels, incremental relabels):
 *   n == the number of slabs of the volume: ALL slabs are handles of this process on one device (time-multiplexed on one GPU: how
 *       the multi-GPU schedule is exercised and measured where there is one GPU); `t` is ignored;
 *   n == 1, mgc_comm_init was called: this rank's slab, borders and reductions over RCCL / xGMI (what bench.py --gpus N runs);
 *   n == 1, `t` given: borders and reductions through the caller's callbacks on HOST buffers (development transports: gloo,
 *       a directory of files).
 * Replaces the serial loop of Graph::maxflow (maxflow.cpp:472-604) for a volume cut into Z-slabs. */
typedef struct mgc_transport {
    void* ctx;
    /* both borders at once: send_lo / recv_lo with rank - 1, send_hi / recv_hi with rank + 1 (a pair is NULL where the volume ends), nbytes each */
    int (*exchange)(void* ctx, const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, int64_t nbytes);
    int (*allreduce)(void* ctx, int64_t* v, int n, int op); /* in place over all ranks; op 0 = sum, 1 = min */
    /* point to point with the neighbour on `side` (0: rank - 1, 1: rank + 1): the carry planes of the distance transforms, a pipeline over the ranks */
    int (*send)(void* ctx, int side, const void* buf, int64_t nbytes);
    int (*recv)(void* ctx, int side, void* buf, int64_t nbytes);
} mgc_transport;
int mgc_solve_slabs(mgc_handle* slabs, int n, const mgc_transport* t, mgc_slab_stats* out);
int mgc_allreduce_counts(mgc_handle h, int64_t* out32);


/* ======================================================================================
 * Sparse graphs ("msg"): everything that is not a 1-D..3-D voxel lattice -- the region graph of
 * graph_from_labels (reference medpy/graphcut/generate.py:177-338) with the terms of energy_label.py:33-404, voxel
 * graphs of more than three dimensions, and graphs assembled edge by edge through GCGraph.set_nweight
 * (graph.py:382-440).  One handle = one maxflow.GraphDouble (wrapper.cpp:59-89).
 * ==================================================================================== */
typedef struct msg_graph* msg_handle;

/* region terms, reference energy_label.py */
typedef enum msg_label_term {
    MSG_LABEL_STAWIASKI = 1,            /* energy_label.py:123-214; image = gradient image                         */
    MSG_LABEL_STAWIASKI_DIRECTED = 2,   /* energy_label.py:217-353; param = directedness (sign picks the direction) */
    MSG_LABEL_DIFFERENCE_OF_MEANS = 3   /* energy_label.py:33-120;  image = original image                          */
} msg_label_term;

typedef struct msg_stats {
    double  build_ms;        /* edge list -> CSR residual graph (sort, duplicate sums, reverse index) */
    double  solve_ms;
    int64_t rounds;          /* push + gather rounds                                               */
    int64_t global_relabels;
    int64_t relabel_passes;
    int64_t nodes;
    int64_t arcs;            /* distinct directed arcs                                             */
    int64_t edges_added;     /* sum_edge calls represented in the edge list                        */
    int64_t reserved[4];
} msg_stats;

/* GraphDouble(nodes, edges) + add_node(nodes) (graph.py:294-308, graph.cpp:12-31) */
int msg_create(int64_t nodes, int device, msg_handle* out);
int msg_destroy(msg_handle h);
const char* msg_last_error(msg_handle h);
int msg_set_param(msg_handle h, const char* name, int64_t value);

/* n calls of GCGraph.set_nweight(i, j, cap, rev) -> Graph::sum_edge (graph.py:382-440, graph.h:457-480): appended in
 * order, repeated (i, j) are added up in that order when the graph is solved */
int msg_add_edges(msg_handle h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev);
/* the n-links a voxel boundary term adds for an image of ANY number of axes (energy_voxel.py:611-664), generated
 * in HBM in the reference's order; term = mgc_term, spacing NULL = False */
int msg_add_lattice_edges(msg_handle h, int term, int ndim, const int64_t* shape, const void* image, int dtype, double sigma,
                          const double* spacing);
/* the n-links a region term adds for a label image with labels 1..nodes (energy_label.py); labels int64, C-order */
int msg_add_label_edges(msg_handle h, int term, int ndim, const int64_t* shape, const int64_t* labels, const void* image, int dtype,
                        double param);
/* per-region sums of `values` (and voxel counts) for labels 1..nregions: scipy.ndimage.mean / numpy.sum over a region
 * (energy_label.py:88, 394-397); accumulate_f32 = keep a float32 accumulator as numpy.sum does for float32 maps */
int msg_region_sums(int device, int64_t n, const int64_t* labels, const void* values, int dtype, int accumulate_f32, int64_t nregions,
                    double* sums, int64_t* counts);
/* merged t-links: tr[i] = tr_cap after all add_tweights calls, flow_const = what they added to the flow (graph.h:416-425) */
int msg_set_tweights_merged(msg_handle h, const double* tr, double flow_const);
/* GraphDouble.maxflow / what_segment / get_edge (pythongraph.h:20-21, graph.h:482-498, 561-571) */
int msg_maxflow(msg_handle h, double* flow);
int msg_labels(msg_handle h, uint8_t* out);
int msg_what_segment(msg_handle h, int64_t i, int* segment);
int msg_get_edge(msg_handle h, int64_t i, int64_t j, double* cap);
int msg_get_counts(msg_handle h, int64_t* nodes, int64_t* edges_added, int64_t* arcs);
/* every distinct arc of the graph as built (tail, head, capacity), sorted by (tail, head): bulk counterpart of get_edge,
 * used by the DIMACS writer (reference medpy/graphcut/write.py:29-76); arrays hold msg_get_counts(..., &arcs) entries */
int msg_get_arcs(msg_handle h, int32_t* tail, int32_t* head, double* cap);
int msg_get_stats(msg_handle h, msg_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* MEDPY_HIP_H */
