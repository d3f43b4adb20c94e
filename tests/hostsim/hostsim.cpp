/*
 * tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE ONLY (CPU test tier, `-m "not gpu"`).
 *
 * Executes the solver's single-source tile operations (medpy_amd/csrc/mgc_tile_ops.inl) and
 * schedule (mgc_driver.inl) on the host: a "block" is a loop over its 512 lanes, a kernel
 * launch is a loop over tiles.  This lets the algorithm -- including the multi-GPU Z-slab
 * protocol, with slabs living in different processes talking over gloo -- be parity-tested
 * against the BK oracle without a GPU.  It is NOT a fallback: the product library
 * (libmedpyhip.so) is built from mgc_kernels.hip only, contains none of this, and fails loudly
 * without a GPU.
 *
 * The handle API mirrors the slab part of include/medpy_hip.h (mgc_create_slab, mgc_slab_info,
 * mgc_solver_op, mgc_read_counts, mgc_halo_*, mgc_finish) so that medpy_amd/slab.py drives
 * either backend with the same code.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../medpy_amd/csrc/mgc_tile_ops.inl"
static long long g26_sweeps, g26_dirs; /* direction masks of the 26-neighbourhood discharge: sweeps, directions that ran */
#define MGC26_COUNT_STEPS(mask) (g26_sweeps++, g26_dirs += __builtin_popcount(mask))
#include "../../medpy_amd/csrc/mgc_tile_ops26.inl"
#include "../../medpy_amd/csrc/mgc_wave_ops.inl"
#include "../../medpy_amd/csrc/mgc_wave_ops26.inl"
#include "../../medpy_amd/csrc/mgc_dt_ops.inl"
#include "../../medpy_amd/csrc/mgc_brick_ops.inl"
#include "../../medpy_amd/csrc/mgc_driver.inl"
#include <cstdio>

/* work-profile counters (development aid, read with hostsim_prof): [id] = number of mark(id) calls (0 load, 1 labels, 2 sweep,
 * 3 store, ...); [16]/[17] = waves that voted "active" / waves asked; [18] relaxation rounds of mgc_tile_bfs (par calls inside) */
static int64_t g_prof[64];
static std::vector<int32_t> g_tile_discharges;

template <class SH>
struct HostBlockT {
    template <class T>
    struct Reg {
        T v[MGC_TV];
        T& operator[](int t) { return v[t]; }
    };
    SH& S;
    explicit HostBlockT(SH& s) : S(s) {}
    template <class F>
    void par(F f)
    {
        for (int t = 0; t < MGC_TV; ++t) f(t);
    }
    template <class F>
    bool any(F f)
    {
        bool r = false;
        for (int t = 0; t < MGC_TV; ++t) r |= (bool)f(t);
        return r;
    }
    template <class F>
    void wpar(F f)
    {
        for (int t = 0; t < MGC_TV; ++t) f(t);
    }
    template <class F>
    bool wave_any(F f) /* the GPU votes per wave and skips idle waves; running them all is equivalent (idle = no-op) */
    {
        bool r = false;
        for (int w = 0; w < MGC_TV / 64; ++w) {
            bool rw = false;
            for (int t = w * 64; t < w * 64 + 64; ++t) rw |= (bool)f(t);
            g_prof[16] += rw;
            g_prof[17]++;
            r |= rw;
        }
        return r;
    }
    void shift(Reg<double>& dst, Reg<double>& src, int delta)
    {
        for (int t = 0; t < MGC_TV; ++t) {
            const int from = (t & 63) + delta;
            dst[t] = (from >= 0 && from < 64) ? src[(t & ~63) + from] : 0.0;
        }
    }
    int shard(const MgcLattice&) const { return 0; } /* one region per list: the plain layout (MgcLattice::scount) */
    int atomic_add(int32_t* p, int v) { int o = *p; *p += v; return o; }
    uint32_t atomic_exch(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
    void atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
    uint32_t uniform(uint32_t v) const { return v; }
    void atomic_and(uint32_t* p, uint32_t v) { *p &= v; }
    void mark(const MgcLattice&, int id) { g_prof[id & 15]++; }
    void gadd(double* p, double v) { *p += v; }
    void gor(uint32_t* p, uint32_t v) { *p |= v; }
    void wave_fence() {}
    /* exact in-tile labels: reference implementation = chaotic relaxation from scratch (mgc_tile_bfs) */
    template <class MaskFn, class RegI>
    void tile_labels(MaskFn mask, RegI& out)
    {
        int m[MGC_TV];
        for (int t = 0; t < MGC_TV; ++t) {
            S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = MGC_HINF;
            m[t] = mask(t);
        }
        mgc_tile_bfs(*this, [&](int t) { return m[t]; });
        for (int t = 0; t < MGC_TV; ++t) out[t] = S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)];
    }
    void async_to_lds(int t, void* dst, const void* src, int bytes) { if (t == 0) memcpy(dst, src, (size_t)bytes); }
    void async_wait() {}
};
/* host form of the wave executor of mgc_wave_ops.inl: a "wave" is a loop over its 64 lanes */
template <class SH>
struct HostWaveT {
    template <class T, int N>
    struct Reg {
        T v[MGCW_LANES][N];
        T& operator()(int l, int k) { return v[l][k]; }
    };
    template <int N>
    struct RegA { /* (GPU: accumulator registers) */
        double v[MGCW_LANES][N];
        void init(int l, int k, double x) { v[l][k] = x; }
        void set(int l, int k, double x) { v[l][k] = x; }
        double get(int l, int k) const { return v[l][k]; }
    };
    SH& S;
    explicit HostWaveT(SH& s) : S(s) {}
    /* ---- mgc_wave_ops26.inl ---- */
    template <class F>
    uint32_t wave_or(F f)
    {
        uint32_t r = 0;
        for (int l = 0; l < MGCW_LANES; ++l) r |= (uint32_t)f(l);
        return r;
    }
    void uput(Reg<int, 1>& store, int k, uint32_t v) { store(k, 0) = (int)v; }
    uint32_t uget(Reg<int, 1>& store, int k) { return (uint32_t)store(k, 0); }
    void lds_and(uint32_t* p, uint32_t v) { *p &= v; }
    void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
    void gadd(double* p, double v) { *p += v; }
    void gor(uint32_t* p, uint32_t v) { *p |= v; }
    void pin(double&) {}
    void load_batch_end() {}
    template <class F>
    void lanes(F f)
    {
        g_prof[20]++;
        for (int l = 0; l < MGCW_LANES; ++l) f(l);
    }
    template <class F>
    bool any(F f)
    {
        g_prof[21]++;
        bool r = false;
        for (int l = 0; l < MGCW_LANES; ++l) r |= (bool)f(l);
        return r;
    }
    void shift(Reg<double, 1>& dst, Reg<double, 1>& src, int delta)
    {
        g_prof[22]++;
        for (int l = 0; l < MGCW_LANES; ++l) {
            const int from = l + delta;
            dst(l, 0) = (from >= 0 && from < MGCW_LANES) ? src(from, 0) : 0.0;
        }
    }
    /* the x-row contract of shift_x is checked here: nothing may arrive from another row of eight */
    void shift_x(Reg<double, 1>& dst, Reg<double, 1>& src, int delta)
    {
        g_prof[22]++;
        for (int l = 0; l < MGCW_LANES; ++l) {
            const int from = l + delta;
            const bool same_row = from >= 0 && from < MGCW_LANES && (from >> 3) == (l >> 3);
            if (!same_row && from >= 0 && from < MGCW_LANES && src(from, 0) != 0.0) abort();
            dst(l, 0) = same_row ? src(from, 0) : 0.0;
        }
    }
    static constexpr int kPrefetch = -1;
    void hint_begin() {}
    void ticket_issue(const MgcLattice&) {}
    int hint_end(const MgcLattice&) { return -1; }
    void prefetch(const void*, int) {}
    int shard(const MgcLattice&) const { return 0; } /* one region per list: the plain layout (MgcLattice::scount) */
    int atomic_add(int32_t* p, int v) { int o = *p; *p += v; return o; }
    uint32_t atomic_exch(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
    void atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
    void atomic_min(int32_t* p, int v) { if (v < *p) *p = v; }
    uint32_t uniform(uint32_t v) const { return v; }
    void atomic_and(uint32_t* p, uint32_t v) { *p &= v; }
    void mark(int id) { g_prof[id & 15]++; }
    template <class T> T ld(const T* p, int l) { g_prof[50] += (int64_t)sizeof(T); return p[l]; } /* bytes the wave forms move through w.ld / w.st: g_prof[50] read, [51] written */
    template <class T> void st(T* p, int l, T v) { g_prof[51] += (int64_t)sizeof(T); p[l] = v; }
    template <class T> void st_stream(T* p, int l, T v) { st(p, l, v); }
    void fresh() {}
    double fmin_pos(double a, double b) { return a < b ? a : b; }
    int use_here(int v) { return v; }
};
typedef HostWaveT<MgcWaveShared> HostWave;
typedef HostWaveT<MgcWaveShared26> HostWave26;
/* executor of the brick operations (mgc_brick_ops.inl): 4096 lanes, run one after another */
struct HostBrick {
    template <class T>
    struct Reg {
        T v[MGC_BV];
        T& operator[](int t) { return v[t]; }
    };
    MgcBrickShared& S;
    explicit HostBrick(MgcBrickShared& s) : S(s) {}
    template <class F>
    void par(F f)
    {
        for (int t = 0; t < MGC_BV; ++t) f(t);
    }
    template <class F>
    bool any(F f)
    {
        bool r = false;
        for (int t = 0; t < MGC_BV; ++t) r |= (bool)f(t);
        return r;
    }
    int shard(const MgcLattice&) const { return 0; }
    int atomic_add(int32_t* p, int v) { int o = *p; *p += v; return o; }
    uint32_t atomic_exch(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
};
static int g_bricks = 0; /* incremental relabels run their passes over bricks of 2 x 2 x 2 tiles (hostsim_set_bricks; the library's relabel_bricks, off by default) */

/* which form of the two hot tile operations the simulator runs: bit 0 = wave discharge, bit 1 = wave relabel,
 * bit 2 = the wave discharge computes exact in-tile labels first (MGCW_BFS); bit 3 = only in tiles holding a sink link (MGCW_BFS_SINK);
 * bit 4 = the 26-neighbourhood discharge runs one wave per tile (mgc_wave_ops26.inl) */
static int g_wave_mode = 0;
static int g_w26_passes = 2, g_w26_raises = 1, g_w26_flags = 0; /* hostsim_set_w26: step passes / relabel rounds per sweep of the 26-neighbourhood wave discharge */
static int g_act_exact_max = 4096; /* hostsim_set_act_exact: see mgcw_activate_tile */
static FILE* g_trace = NULL; /* one line per wave-form discharge: phase, tile, sweeps (hostsim_trace; tools/sim_launch_model.py) */
static int g_check_exact = 0; /* hostsim_set_check_exact: labels after a global relabel vs exact distances (g_prof[40], [41]) */
static int g_repeat_steps = 1; /* hostsim_set_repeat: which launches run the repeated in-plane steps (bit 0 exact labels, bit 1 radial labels), as the library's repeat_steps */
static int g_use_dt = 1; /* the first global relabel may be a distance transform (hostsim_set_dt) */

typedef HostBlockT<MgcTileShared> HostBlock;
typedef HostBlockT<MgcTileShared26> HostBlock26;
typedef HostBlockT<MgcTileShared26D> HostBlock26D;


/* ---- what MgcSlabGroup / MgcXchg (mgc_driver.inl) ask of a slab, on host memory: the same schedule and the same transport logic as the
 * library's run in the CPU test tier ---- */
template <class D, class BlockT>
struct HostSlabOps {
    D& self() { return *static_cast<D*>(this); }
    const D& self() const { return *static_cast<const D*>(this); }
    int64_t gd0 = 0; /* planes of the whole volume */
    std::vector<char> xbuf[4]; /* send lo, recv lo, send hi, recv hi */
    std::vector<uint16_t> carry[2], carry_recv[2];
    bool multi() const { return false; }
    void exchange(int, uint32_t, int) {}
    const MgcLattice& lattice() const { return self().L; }
    bool has_lower() const { return self().L.tz_own_lo > 0; }
    bool has_upper() const { return self().L.tz_own_hi < self().L.gz; }
    bool needs_carry(int dir) const { return dir == 0 ? self().spec.plane0 > 0 : self().spec.plane1 < gd0; }
    bool sends_carry(int dir) const { return dir == 0 ? (has_upper() && self().spec.own1 - 9 >= self().spec.plane0) : (has_lower() && self().spec.own0 + 8 < gd0); }
    int carry_plane(int dir) const { return (int)(dir == 0 ? self().spec.own1 - 9 - self().spec.plane0 : self().spec.own0 + 8 - self().spec.plane0); }
    int64_t carry_bytes() const { return (int64_t)self().L.dy * self().L.dx * (int64_t)sizeof(uint16_t); }
    uint16_t* carry_buf(int dir) { carry[dir].resize((size_t)(self().L.dy * self().L.dx)); return carry[dir].data(); }
    uint16_t* carry_recv_buf(int dir) { carry_recv[dir].resize((size_t)(self().L.dy * self().L.dx)); return carry_recv[dir].data(); }
    int64_t halo_msg_bytes(int kind) const
    {
        const MgcLattice& L = self().L;
        return mgc_halo_compact_nd(L, kind) ? mgc_halo_off_rec_nd(L) + (int64_t)L.halo_max_rec * mgc_halo_rec_bytes_nd(L, kind) : mgc_halo_bytes_nd(L, kind);
    }
    void* buf(int i)
    {
        const MgcLattice& L = self().L;
        int64_t n = mgc_halo_bytes_nd(L, 0);
        for (int k = 1; k < 3; ++k) n = mgc_halo_bytes_nd(L, k) > n ? mgc_halo_bytes_nd(L, k) : n;
        if ((int64_t)xbuf[i].size() < n) xbuf[i].assign((size_t)n, 0);
        return xbuf[i].data();
    }
    void* recv_buf(int side) { return buf(2 * side + 1); }
    void* halo_pack(int side, int kind)
    {
        MgcLattice& L = self().L;
        BlockT x(self().S);
        void* const dst = buf(2 * side);
        if (mgc_halo_compact_nd(L, kind)) memset((char*)dst + mgc_halo_off_count_nd(L), 0, 4);
        for (int i = 0; i < L.gy * L.gx; ++i) mgc_halo_pack_nd(x, L, side, kind, i, dst);
        return dst;
    }
    void halo_unpack(int side, int kind, const void* b, uint32_t epoch, int list)
    {
        MgcLattice& L = self().L;
        BlockT x(self().S);
        for (int i = 0; i < L.gy * L.gx; ++i) mgc_halo_unpack_nd(x, L, side, kind, i, b, epoch, list);
    }
    void to_host(void* host, const void* b, int64_t n) { memcpy(host, b, (size_t)n); }
    void from_host(void* b, const void* host, int64_t n) { memcpy(b, host, (size_t)n); }
    bool has_comm() const { return false; }
    int native_exchange(int, uint32_t, int) { return 1; }
    int native_allreduce(int64_t*, int, int) { return 1; }
    int native_send(int, const void*, int64_t) { return 1; }
    int native_recv(int, void*, int64_t) { return 1; }
    int count_get(int i) { return self().L.count[i]; }
    void count_set(int i, int v) { self().L.count[i] = v; }
};

struct HostDev : HostSlabOps<HostDev, HostBlockT<MgcTileShared>> {
    MgcLattice L;
    MgcTileShared S;
    MgcWaveShared WS;
    MgcSlabSpec spec;
    std::vector<double> rcap, excess, sink, obox;
    std::vector<int32_t> height, lists, count;
    std::vector<uint8_t> rmask;
    std::vector<uint32_t> oflags, stamp, rstamp, status;
    std::vector<int32_t> hshadow[2];
    void fill_heights_inf()
    {
        for (int64_t i = 0; i < (int64_t)L.ntiles * MGC_TV; ++i) L.height[i] = MGC_HINF;
        for (int t = 0; t < L.ntiles; ++t) L.status[t] = (L.status[t] & ~MGC_ST_SETTLED) | MGC_ST_ALLINF; /* until a relabel lowers a label */
        for (int sd = 0; sd < 2; ++sd) std::fill(hshadow[sd].begin(), hshadow[sd].end(), (int32_t)MGC_HINF); /* what the neighbours hold now */
    }
    void zero_count(int i) { L.count[i] = 0; }
    void range_push(const char*) {}
    void range_pop() {}
    void read_counts(int* out) { memcpy(out, L.count, MGC_NCOUNT * sizeof(int)); }
    void absorb_all()
    {
        HostBlock x(S);
        HostWave w(WS);
        for (int t = 0; t < L.ntiles; ++t) {
            if (g_wave_mode & 1) mgcw_absorb_tile(w, L, t);
            else mgc_absorb_tile(x, L, t);
        }
    }
    bool brick_mode = false; /* the passes of the relabel in progress run over bricks (set by reset_suspect, as in the library) */
    MgcBrickShared BS;
    void relabel_all(uint32_t epoch, int next)
    {
        HostBlock x(S);
        HostWave w(WS);
        brick_mode = false;
        for (int t = 0; t < L.ntiles; ++t) {
            if (L.status[t] & 2u) L.count[9]++;
            if (g_wave_mode & 2) mgcw_relabel_tile(w, L, t, epoch, next, true);
            else mgc_relabel_tile(x, L, t, epoch, next, true);
        }
    }
    /* the first global relabel as a distance transform (mgc_dt_ops.inl) under the library's condition: every n-link of
     * the volume residual, one handle (no slabs) */
    bool first_relabel_dt()
    {
        if (!g_use_dt || spec.nranks > 1 || labels_valid || L.dz + L.dy + L.dx >= MGC_DT_INF - 8) return false;
        for (int64_t z = 0; z < L.dz; ++z)
            for (int64_t y = 0; y < L.dy; ++y)
                for (int64_t x = 0; x < L.dx; ++x) {
                    int tile, loc;
                    mgc_node_to_tile(L, (z * L.dy + y) * L.dx + x, tile, loc);
                    const uint32_t need = (x > 0 ? 1u : 0u) | (x + 1 < L.dx ? 2u : 0u) | (y > 0 ? 4u : 0u) | (y + 1 < L.dy ? 8u : 0u) |
                                          (z > 0 ? 16u : 0u) | (z + 1 < L.dz ? 32u : 0u);
                    if ((rmask[(int64_t)tile * MGC_TV + loc] & need) != need) return false;
                }
        std::vector<uint16_t> T((size_t)L.ntiles * MGC_TV);
        HostWave w(WS);
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, false, 1, 0>(w, L, i, L.rmask, T.data());
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, true, 0, 0>(w, L, i, T.data(), T.data());
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, false, 0, 0>(w, L, i, T.data(), T.data());
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, true, 0, 0>(w, L, i, T.data(), T.data());
        for (int i = 0; i < L.gy * L.gx; ++i) mgc_dt_scan_line<2, false, 0, 0>(w, L, i, T.data(), T.data());
        for (int i = 0; i < L.gy * L.gx; ++i) mgc_dt_scan_line<2, true, 0, 1>(w, L, i, T.data(), L.height);
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_finish_tile(w, L, t);
        L.count[9] += L.ntiles;
        g_prof[28]++;
        return true;
    }
    bool radial_after_passes() const { return false; } /* (radial labels only on top of a first relabel by transform, as in the library) */
    /* ---- the transforms of a slab, scan by scan (MgcSlabGroup::transform) ---- */
    bool all_residual_local() const
    {
        for (int64_t z = 0; z < L.dz; ++z)
            for (int64_t y = 0; y < L.dy; ++y)
                for (int64_t x = 0; x < L.dx; ++x) {
                    int tile, loc;
                    mgc_node_to_tile(L, (z * L.dy + y) * L.dx + x, tile, loc);
                    const uint32_t need = (x > 0 ? 1u : 0u) | (x + 1 < L.dx ? 2u : 0u) | (y > 0 ? 4u : 0u) | (y + 1 < L.dy ? 8u : 0u) |
                                          (z > 0 ? 16u : 0u) | (z + 1 < L.dz ? 32u : 0u);
                    if ((rmask[(int64_t)tile * MGC_TV + loc] & need) != need) return false;
                }
        return true;
    }
    std::vector<uint16_t> dt16;
    uint16_t* dt_cur = nullptr;
    bool labels_valid = false;
    bool dt_applicable() { return g_use_dt && !labels_valid && gd0 + L.dy + L.dx < MGC_DT_INF - 8 && all_residual_local(); }
    void dt_scans_xy(int seed)
    {
        HostWave w(WS);
        labels_valid = true;
        std::vector<uint16_t>& A = seed == 1 ? dt16 : ds16;
        A.assign((size_t)L.ntiles * MGC_TV, 0);
        uint16_t* const T = dt_cur = A.data();
        if (seed == 1) for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, false, 1, 0>(w, L, i, L.rmask, T);
        else for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, false, 2, 0>(w, L, i, L.excess, T);
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, true, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, false, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, true, 0, 0>(w, L, i, T, T);
    }
    void dt_scan_z(bool bwd, int final_kind, int c_min, const uint16_t* cin, bool want_out)
    {
        HostWave w(WS);
        uint16_t* const T = dt_cur;
        uint16_t* const cout = want_out ? carry_buf(bwd ? 1 : 0) : nullptr;
        const int plane = want_out ? carry_plane(bwd ? 1 : 0) : -1;
        for (int i = 0; i < L.gy * L.gx; ++i) {
            if (!bwd) mgc_dt_scan_line<2, false, 0, 0>(w, L, i, T, T, 0, nullptr, cin, cout, plane);
            else if (final_kind == 1) mgc_dt_scan_line<2, true, 0, 1>(w, L, i, T, L.height, 0, nullptr, cin, cout, plane);
            else if (final_kind == 2) mgc_dt_scan_line<2, true, 0, 2>(w, L, i, T, T, c_min, hexact.data(), cin, cout, plane);
            else mgc_dt_scan_line<2, true, 0, 0>(w, L, i, T, T, 0, nullptr, cin, cout, plane);
        }
        if (bwd && final_kind == 2) lowered = L.count[mgc_cnt_radial_c(L)] < MGC_HINF && L.count[mgc_cnt_radial_c(L)] >= c_min;
    }
    void dt_finish()
    {
        HostWave w(WS);
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_finish_tile(w, L, t);
        L.count[9] += L.ntiles;
        g_prof[28]++;
    }
    void shadow_sync()
    {
        HostWave w(WS);
        for (int side = 0; side < 2; ++side)
            for (int i = 0; i < L.gy * L.gx; ++i) mgc_shadow_sync_tile(w, L, side, i, L.height);
    }
    bool radial_prepare() { hexact.assign((size_t)L.ntiles * MGC_TV, 0); return true; }
    void radial_cmin()
    {
        HostWave w(WS);
        L.count[mgc_cnt_radial_c(L)] = MGC_HINF;
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_cmin_tile(w, L, t);
    }
    void radial_swap() { std::swap(height, hexact); L.height = height.data(); }
    /* ---- radial labels of the flood phase (mgc_dt_ops.inl), the library's HipDevT ops on host arrays ---- */
    std::vector<uint16_t> ds16;      /* 1 + L1 distance from the nearest voxel that held excess when the solve began */
    std::vector<int32_t> hexact;     /* the exact labels of the last global relabel */
    bool radial_on = false, lowered = false;
    int radial_cycle_no = 0;
    void set_radial(bool on) { radial_on = on; if (on) radial_cycle_no++; }
    bool radial_begin(int c_min)
    {
        HostWave w(WS);
        ds16.assign((size_t)L.ntiles * MGC_TV, 0);
        uint16_t* const T = ds16.data();
        L.count[mgc_cnt_radial_c(L)] = MGC_HINF;
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_cmin_tile(w, L, t); /* C from the exact labels of the source voxels */
        hexact.assign((size_t)L.ntiles * MGC_TV, 0); /* (filled by the last scan: the library's HipDevT::radial_begin, array for array) */
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, false, 2, 0>(w, L, i, L.excess, T);
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, true, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, false, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, true, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gy * L.gx; ++i) mgc_dt_scan_line<2, false, 0, 0>(w, L, i, T, T);
        lowered = L.count[mgc_cnt_radial_c(L)] < MGC_HINF && L.count[mgc_cnt_radial_c(L)] >= c_min;
        /* ... and the labels lowered on the way, into the OTHER array -- every label, lowered or not -- and the two trade places: the exact labels
         * are never copied aside */
        for (int i = 0; i < L.gy * L.gx; ++i) mgc_dt_scan_line<2, true, 0, 2>(w, L, i, T, T, c_min, hexact.data());
        std::swap(height, hexact);
        L.height = height.data();
        return true;
    }
    void radial_save_exact() { hexact.assign(L.height, L.height + (size_t)L.ntiles * MGC_TV); }
    void radial_restore_exact() { std::swap(height, hexact); L.height = height.data(); lowered = false; } /* (what the flood made of the radial labels is dropped as a whole) */
    void radial_lower(int c_min)
    {
        HostWave w(WS);
        lowered = L.count[mgc_cnt_radial_c(L)] < MGC_HINF && L.count[mgc_cnt_radial_c(L)] >= c_min; /* (what mgc_dt_lower_tile decides by) */
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_lower_tile(w, L, t, ds16.data(), c_min);
    }
    void source_open()
    {
        HostWave w(WS);
        if (getenv("HOSTSIM_TRACE")) { /* development aid: how saturated are the tiles the flood touched? */
            int64_t dirty = 0, arcs = 0, gone = 0, exc_tiles = 0, exc_clean = 0;
            for (int t = 0; t < L.ntiles; ++t) {
                const bool d = (L.status[t] & MGC_ST_DIRTY) != 0, e = (L.status[t] & MGC_ST_EXCESS) != 0;
                exc_tiles += e; exc_clean += e && !d;
                if (!e) continue;
                dirty++;
                for (int v = 0; v < MGC_TV; ++v) { arcs += 6; gone += 6 - __builtin_popcount(rmask[(int64_t)t * MGC_TV + v] & 63); }
            }
            fprintf(stderr, "[sim] dirty tiles %lld: %.1f%% of their arcs are saturated; tiles with excess %lld, of them clean %lld\n", (long long)dirty, 100.0 * gone / (arcs ? arcs : 1), (long long)exc_tiles, (long long)exc_clean);
        }
        for (int t = 0; t < L.ntiles; ++t) mgc_source_open_tile(w, L, t);
    }
    void relabel_list(int lst, uint32_t epoch, int next, int zero_list = -1)
    {
        HostBlock x(S);
        HostWave w(WS);
        const int n = L.count[lst];
        if (zero_list >= 0) L.count[zero_list] = 0;
        if (brick_mode) {
            HostBrick b(BS);
            L.count[9] += 4 * n;
            g_prof[29] += n;
            for (int i = 0; i < n; ++i) mgc_relabel_brick(b, L, L.list[lst][i], epoch, next);
            return;
        }
        L.count[9] += n;
        for (int i = 0; i < n; ++i) {
            if (g_wave_mode & 2) mgcw_relabel_tile(w, L, L.list[lst][i], epoch, next, false);
            else mgc_relabel_tile(x, L, L.list[lst][i], epoch, next, false);
        }
    }
    /* hostsim_set_check_exact: after every global relabel the labels are compared with the exact distances to the sink in the
     * residual graph (plain relaxation over the whole volume; everything pushed has been absorbed by then): g_prof[40] relabels
     * checked, g_prof[41] voxels that differ.  (A single device only: a slab would need its neighbours' residuals.) */
    void check_exact_labels()
    {
        const int64_t N = L.nvox;
        std::vector<int32_t> dist((size_t)N, MGC_HINF), tl((size_t)N), lc((size_t)N);
        for (int64_t id = 0; id < N; ++id) {
            int tile, loc;
            mgc_node_to_tile(L, id, tile, loc);
            tl[id] = tile; lc[id] = loc;
            if ((L.status[tile] & MGC_ST_SINK) && L.sink[(int64_t)tile * MGC_TV + loc] > 0.0) dist[id] = 1;
        }
        static const int off[6][3] = {{0, 0, -1}, {0, 0, 1}, {0, -1, 0}, {0, 1, 0}, {-1, 0, 0}, {1, 0, 0}}; /* (dz, dy, dx) of direction d */
        for (bool changed = true; changed;) {
            changed = false;
            for (int64_t id = 0; id < N; ++id) {
                const int64_t x0 = id % L.dx, y0 = (id / L.dx) % L.dy, z0 = id / (L.dx * L.dy);
                int32_t best = dist[id];
                for (int d = 0; d < 6; ++d) {
                    if (!(L.rcap[((int64_t)tl[id] * 6 + d) * MGC_TV + lc[id]] > 0.0)) continue;
                    const int64_t z = z0 + off[d][0], y = y0 + off[d][1], xx = x0 + off[d][2];
                    if (z < 0 || z >= L.dz || y < 0 || y >= L.dy || xx < 0 || xx >= L.dx) continue;
                    const int32_t hv = dist[(z * L.dy + y) * L.dx + xx];
                    if (hv < MGC_HINF && hv + 1 < best) best = hv + 1;
                }
                if (best < dist[id]) { dist[id] = best; changed = true; }
            }
        }
        g_prof[40]++;
        /* (right after radial_begin the labels in L.height are the radial ones: the exact labels are the copy kept aside) */
        const int32_t* const lab = lowered ? hexact.data() : L.height;
        for (int64_t id = 0; id < N; ++id) g_prof[41] += lab[(int64_t)tl[id] * MGC_TV + lc[id]] != dist[id];
    }
    /* work profile: what does an incremental relabel change?  labels of the tiles it reset, before and after */
    std::vector<std::pair<int, std::vector<int32_t> > > reset_snapshot;
    std::vector<char> weak0;
    void activate_all(uint32_t phase)
    {
        for (auto& sn : reset_snapshot) {
            const int32_t* h = &L.height[(int64_t)sn.first * MGC_TV];
            int changed = 0, face_changed = 0;
            for (int t = 0; t < MGC_TV; ++t) {
                if (h[t] == sn.second[t]) continue;
                changed++;
                const int z = t >> 6, y = (t >> 3) & 7, xx = t & 7;
                if (z == 0 || z == 7 || y == 0 || y == 7 || xx == 0 || xx == 7) face_changed++;
            }
            g_prof[24]++;                       /* tiles reset                                        */
            g_prof[25] += changed == 0;         /* ... whose labels came back as they were            */
            g_prof[26] += face_changed == 0;    /* ... whose face voxels (what neighbours see) did    */
            g_prof[27] += changed;              /* voxels whose label changed                          */
        }
        reset_snapshot.clear();
        if (g_check_exact) check_exact_labels();
        if (getenv("HOSTSIM_WEAK")) { /* development aid: how many of the weak arcs (capacity as loaded < 1e-9) are saturated by now, how many voxels hold excess */
            int64_t weak = 0, sat = 0, exc = 0, excfin = 0;
            if (weak0.empty()) { weak0.resize(rcap.size()); for (size_t i = 0; i < rcap.size(); ++i) weak0[i] = rcap[i] > 0.0 && rcap[i] < 1e-9; }
            for (size_t i = 0; i < rcap.size(); ++i) if (weak0[i]) { weak++; sat += rcap[i] == 0.0; }
            for (size_t i = 0; i < excess.size(); ++i) if (excess[i] > 0.0) { exc++; excfin += height[i] < MGC_HINF; }
            fprintf(stderr, "[sim] weak arcs %lld, saturated %lld (%.1f%%); voxels with excess %lld, with a finite label %lld\n", (long long)weak, (long long)sat, 100.0 * sat / (weak ? weak : 1), (long long)exc, (long long)excfin);
        }
        HostBlock x(S);
        HostWave w(WS);
        int candidates = 0; /* what the library's tile filter would list */
        for (int t = 0; t < L.ntiles; ++t)
            candidates += mgc_owned(L, t) && (L.status[t] & (MGC_ST_EXCESS | MGC_ST_ALLINF)) == MGC_ST_EXCESS;
        const bool exact = candidates <= g_act_exact_max;
        for (int t = 0; t < L.ntiles; ++t) {
            if (g_wave_mode & 1) {
                if ((L.status[t] & (MGC_ST_EXCESS | MGC_ST_ALLINF)) == MGC_ST_EXCESS) mgcw_activate_tile(w, L, t, phase, exact);
            } else {
                mgc_activate_tile(x, L, t, phase);
            }
        }
    }
    int suspect_batch() const { return 8; }
    void suspect_pass()
    {
        for (int t = 0; t < L.ntiles; ++t)
            if (mgc_suspect_tile(L, t)) L.count[MGC_CNT_CHANGED] = 1;
    }
    void reset_suspect(uint32_t epoch, int list)
    {
        HostBlock x(S);
        brick_mode = g_bricks && spec.nranks == 1;
        for (int t = 0; t < L.ntiles; ++t) {
            if (!g_tile_discharges.empty() && (L.status[t] & MGC_ST_SUSPECT) && mgc_owned(L, t))
                reset_snapshot.emplace_back(t, std::vector<int32_t>(&L.height[(int64_t)t * MGC_TV], &L.height[(int64_t)t * MGC_TV] + MGC_TV));
            if (brick_mode) { /* what k_reset_suspect does with bricks = 1 */
                if (L.status[t] & MGC_ST_SUSPECT) {
                    for (int v = 0; v < MGC_TV; ++v) L.height[(int64_t)t * MGC_TV + v] = MGC_HINF;
                    L.status[t] = (L.status[t] & ~(MGC_ST_SUSPECT | MGC_ST_DIRTY | (63u << MGC_ST_DEP_SHIFT))) | MGC_ST_ALLINF;
                    mgc_enqueue_brick(x, L, list, epoch, mgc_brick_of_tile(L, t));
                }
            } else {
                mgc_reset_suspect_tile(x, L, t, epoch, list);
            }
        }
    }
    void discharge(int lst, uint32_t phase, int cycles, int sweeps)
    {
        HostBlock x(S);
        const int n = L.count[lst];
        L.count[8] += n;
        HostWave w(WS);
        if (getenv("HOSTSIM_DUMP")) { /* development aid: labels, excess and "a weak arc of this voxel is saturated" of the middle z-plane, before every phase */
            char path[512];
            snprintf(path, sizeof(path), "%s/phase_%04u.bin", getenv("HOSTSIM_DUMP"), phase);
            FILE* f = fopen(path, "wb");
            if (f) {
                const int64_t z = L.dz / 2;
                for (int64_t y = 0; y < L.dy; ++y)
                    for (int64_t xx = 0; xx < L.dx; ++xx) {
                        int tile, loc;
                        mgc_node_to_tile(L, (z * L.dy + y) * L.dx + xx, tile, loc);
                        const int64_t i = (int64_t)tile * MGC_TV + loc;
                        double rec[3] = {(double)height[i], excess[i], 0.0};
                        for (int d = 0; d < 6; ++d) if (!weak0.empty() && weak0[((int64_t)tile * 6 + d) * MGC_TV + loc] && rcap[((int64_t)tile * 6 + d) * MGC_TV + loc] == 0.0) rec[2] += 1.0;
                        fwrite(rec, sizeof(double), 3, f);
                    }
                fclose(f);
            }
        }
        if (getenv("HOSTSIM_WEAK") && atoi(getenv("HOSTSIM_WEAK")) > 1 && !weak0.empty()) {
            int64_t sat = 0;
            for (size_t i = 0; i < rcap.size(); ++i) sat += weak0[i] && rcap[i] == 0.0;
            /* voxels that hold excess under a finite label AND a residual arc one label down (could push now), split by whether their tile is queued */
            int64_t can_q = 0, can_nq = 0, obox_pending = 0;
            std::vector<char> queued(L.ntiles, 0);
            for (int i = 0; i < n; ++i) queued[L.list[lst][i]] = 1;
            for (int i = 0; i < L.count[(lst + 1) & 3]; ++i) queued[L.list[(lst + 1) & 3][i]] = 1;
            for (int t = 0; t < L.ntiles; ++t) {
                int tz, ty, tx;
                mgc_tile_coords(L, t, tz, ty, tx);
                for (int f = 0; f < 6; ++f) for (int k = 0; k < MGC_TF; ++k) obox_pending += L.obox[((int64_t)t * 6 + f) * MGC_TF + k] != 0.0;
                for (int v = 0; v < MGC_TV; ++v) {
                    const int64_t g = (int64_t)t * MGC_TV + v;
                    if (!(L.excess[g] > 0.0) || L.height[g] >= MGC_HINF) continue;
                    const int z = v >> 6, y = (v >> 3) & 7, xx = v & 7;
                    bool can = false;
                    for (int d = 0; d < 6 && !can; ++d) {
                        if (!(L.rcap[((int64_t)t * 6 + d) * MGC_TV + v] > 0.0)) continue;
                        int nz = z + (d == 4 ? -1 : d == 5 ? 1 : 0), ny = y + (d == 2 ? -1 : d == 3 ? 1 : 0), nx = xx + (d == 0 ? -1 : d == 1 ? 1 : 0);
                        int nt = t;
                        if (nz < 0 || nz > 7 || ny < 0 || ny > 7 || nx < 0 || nx > 7) { nt = mgc_tile_nbr(L, tz, ty, tx, d); nz &= 7; ny &= 7; nx &= 7; }
                        if (nt < 0) continue;
                        can = L.height[(int64_t)nt * MGC_TV + nz * 64 + ny * 8 + nx] == L.height[g] - 1;
                    }
                    if (can) (queued[t] ? can_q : can_nq)++;
                }
            }
            fprintf(stderr, "[sim]   phase %u: %d tiles, weak arcs saturated so far %lld; voxels that could push: %lld in queued tiles, %lld elsewhere; outbox cells pending %lld\n", phase, n, (long long)sat, (long long)can_q, (long long)can_nq, (long long)obox_pending);
        }
        for (int i = 0; i < n; ++i) {
            if ((int)g_tile_discharges.size() == L.ntiles) g_tile_discharges[L.list[lst][i]]++;
            if (g_wave_mode & 1) {
                const int64_t sweeps_before = g_prof[2];
                const int dflags = ((g_wave_mode & 4) ? MGCW_BFS : 0) | ((g_wave_mode & 8) ? MGCW_BFS_SINK : 0) | (radial_on ? MGCW_SAT_DIRTY : 0) | ((radial_on && radial_cycle_no > 1) ? MGCW_INFLOW_DIRTY : 0);
                if (g_repeat_steps & (radial_on ? 2 : 1)) mgcw_discharge_tile<MGCW_REPEAT_MAX>(w, L, L.list[lst][i], phase, sweeps, dflags);
                else mgcw_discharge_tile<1>(w, L, L.list[lst][i], phase, sweeps, dflags);
                if (g_trace) fprintf(g_trace, "%u %d %d\n", phase, L.list[lst][i], (int)(g_prof[2] - sweeps_before));
                g_prof[3]++;
            } else {
                mgc_discharge_tile(x, L, L.list[lst][i], phase, radial_on && cycles < 0 ? (radial_cycle_no > 1 ? -3 : -2) : cycles, sweeps);
            }
        }
    }

    void init(int64_t d0, int64_t d1, int64_t d2, const MgcSlabSpec* sp)
    {
        memset(&L, 0, sizeof(L));
        L.dz = d0; L.dy = d1; L.dx = d2;
        L.nvox = d0 * d1 * d2;
        L.gz = (int)((d0 + 7) / 8); L.gy = (int)((d1 + 7) / 8); L.gx = (int)((d2 + 7) / 8);
        L.ntiles = L.gz * L.gy * L.gx;
        L.tz_own_lo = 0; L.tz_own_hi = L.gz; L.tz_global0 = 0; L.ndir = 6;
        if (sp) { spec = *sp; L.tz_own_lo = sp->own_lo; L.tz_own_hi = sp->own_hi; L.tz_global0 = sp->tz_global0; }
        else { memset(&spec, 0, sizeof(spec)); spec.nranks = 1; spec.own_hi = L.gz; spec.plane1 = spec.own1 = d0; }
        const int64_t nt = L.ntiles;
        rcap.assign(nt * 6 * MGC_TV, 0.0); excess.assign(nt * MGC_TV, 0.0); sink.assign(nt * MGC_TV, 0.0);
        obox.assign(nt * 6 * MGC_TF, 0.0); height.assign(nt * MGC_TV, MGC_HINF); lists.assign(8 * nt, 0);
        count.assign(MGC_NCOUNT, 0); rmask.assign(nt * MGC_TV, 0);
        oflags.assign(nt, 0); stamp.assign(nt, 0); rstamp.assign(nt, 0); status.assign(nt, 0);
        L.rcap = rcap.data(); L.cap0 = NULL; L.excess = excess.data(); L.sink = sink.data(); L.height = height.data();
        L.rmask = rmask.data(); L.obox = obox.data(); L.oflags = oflags.data();
        for (int i = 0; i < 8; ++i) L.list[i] = lists.data() + i * nt;
        L.count = count.data(); L.stamp = stamp.data(); L.rstamp = rstamp.data(); L.status = status.data();
        L.scount = L.count; L.nshard = 1; L.shard_cap = (int)nt;
        L.halo_max_rec = L.gy * L.gx; /* a border message holds every border tile unless a test says otherwise (hostsim_set_halo_max) */
        for (int sd = 0; sd < 2; ++sd) {
            hshadow[sd].assign((size_t)L.gy * L.gx * MGC_TF, (int32_t)MGC_HINF);
            L.hshadow[sd] = hshadow[sd].data();
        }
    }

    /* w[a] = forward n-link capacities along array axis a of the LOCAL volume in the oracle's per-axis layout
     * (oracle/energy_numpy.py:boundary_weights); trcap = merged t-link residual per local voxel */
    void load(const double* w0, const double* w1, const double* w2, const double* trcap)
    {
        const int64_t D0 = L.dz, D1 = L.dy, D2 = L.dx;
        for (int64_t z = 0; z < D0; ++z)
            for (int64_t y = 0; y < D1; ++y)
                for (int64_t x = 0; x < D2; ++x) {
                    int tile, loc;
                    const int64_t id = (z * D1 + y) * D2 + x;
                    mgc_node_to_tile(L, id, tile, loc);
                    double r[6] = {0, 0, 0, 0, 0, 0};
                    if (x > 0) r[0] = w2[(z * D1 + y) * (D2 - 1) + (x - 1)];
                    if (x < D2 - 1) r[1] = w2[(z * D1 + y) * (D2 - 1) + x];
                    if (y > 0) r[2] = w1[(z * (D1 - 1) + (y - 1)) * D2 + x];
                    if (y < D1 - 1) r[3] = w1[(z * (D1 - 1) + y) * D2 + x];
                    if (z > 0) r[4] = w0[((z - 1) * D1 + y) * D2 + x];
                    if (z < D0 - 1) r[5] = w0[(z * D1 + y) * D2 + x];
                    int m = 0;
                    for (int d = 0; d < 6; ++d) {
                        rcap[((int64_t)tile * 6 + d) * MGC_TV + loc] = r[d];
                        if (r[d] > 0.0) m |= 1 << d;
                    }
                    const double tr = trcap[id];
                    excess[(int64_t)tile * MGC_TV + loc] = tr > 0 ? tr : 0.0;
                    if (tr > 0) status[tile] |= MGC_ST_EXCESS | MGC_ST_SOURCE;
                    sink[(int64_t)tile * MGC_TV + loc] = tr < 0 ? -tr : 0.0;
                    if (tr < 0) { m |= MGC_MASK_SINK; status[tile] |= 2u; }
                    rmask[(int64_t)tile * MGC_TV + loc] = (uint8_t)m;
                }
    }

    void labels(uint8_t* out)
    {
        for (int64_t id = 0; id < L.nvox; ++id) {
            int tile, loc;
            mgc_node_to_tile(L, id, tile, loc);
            out[id] = height[(int64_t)tile * MGC_TV + loc] < MGC_HINF ? 0 : 1;
        }
    }
};

extern "C" {

void hostsim_set_wave_mode(int mode) { g_wave_mode = mode; }
void hostsim_set_w26(int passes, int raises, int flags) { g_w26_passes = passes; g_w26_raises = raises; g_w26_flags = flags; }
void hostsim_set_dt(int on) { g_use_dt = on; }
void hostsim_set_repeat(int bits) { g_repeat_steps = bits; }
void hostsim_set_check_exact(int on) { g_check_exact = on; }
void hostsim_trace(const char* path) { if (g_trace) fclose(g_trace); g_trace = path && *path ? fopen(path, "w") : NULL; }
void hostsim_set_bricks(int on) { g_bricks = on; }
/* record slots of a compacted border message (MgcLattice::halo_max_rec); which: 6 or 26 */
void hostsim_set_halo_max(void* h, int which, int n);
void hostsim_set_act_exact(int n) { g_act_exact_max = n; }

/* work-profile read-out: copies and clears the counters; tiles != NULL with ntiles > 0 arms / returns the per-tile discharge counts */
void hostsim_prof(int64_t* out, int32_t* tiles, int ntiles)
{
    memcpy(out, g_prof, sizeof(g_prof));
    memset(g_prof, 0, sizeof(g_prof));
    if (tiles && (int)g_tile_discharges.size() == ntiles) memcpy(tiles, g_tile_discharges.data(), (size_t)ntiles * 4);
    g_tile_discharges.assign(ntiles > 0 ? ntiles : 0, 0);
}

/* ---- handle API (mirrors the slab part of include/medpy_hip.h) ---- */
void* hostsim_create(const int64_t* gshape, int rank, int nranks)
{
    HostDev* d = new HostDev();
    d->gd0 = gshape[0];
    if (nranks <= 1) {
        d->init(gshape[0], gshape[1], gshape[2], NULL);
    } else {
        MgcSlabSpec sp;
        if (mgc_slab_spec(gshape[0], rank, nranks, &sp)) { delete d; return NULL; }
        d->init(sp.plane1 - sp.plane0, gshape[1], gshape[2], &sp);
    }
    return d;
}

void hostsim_destroy(void* h) { delete (HostDev*)h; }

int hostsim_slab_info(void* h, int64_t* info)
{
    HostDev* d = (HostDev*)h;
    info[0] = d->spec.plane0; info[1] = d->spec.plane1; info[2] = d->spec.own0; info[3] = d->spec.own1;
    info[4] = d->L.tz_own_lo > 0; info[5] = d->L.tz_own_hi < d->L.gz; info[6] = (int64_t)d->L.gy * d->L.gx; info[7] = d->spec.nranks;
    return 0;
}

int hostsim_load(void* h, const double* w0, const double* w1, const double* w2, const double* trcap)
{
    ((HostDev*)h)->load(w0, w1, w2, trcap);
    return 0;
}

int hostsim_solver_op(void* h, int op, int64_t a0, int64_t a1, int64_t a2, int64_t a3)
{
    HostDev* d = (HostDev*)h;
    switch (op) {
    case 0: d->absorb_all(); break;
    case 1: d->fill_heights_inf(); break;
    case 2: d->zero_count((int)a0); break;
    case 3: d->relabel_all((uint32_t)a0, (int)a1); break;
    case 4: d->relabel_list((int)a0, (uint32_t)a1, (int)a2); break;
    case 5: d->activate_all((uint32_t)a0); break;
    case 6: d->discharge((int)a0, (uint32_t)a1, (int)a2, (int)a3); break;
    case 7: d->suspect_pass(); break;
    case 8: d->reset_suspect((uint32_t)a0, (int)a1); break;
    default: return 1;
    }
    return 0;
}

int hostsim_read_counts(void* h, int32_t* out) { ((HostDev*)h)->read_counts(out); return 0; }

int hostsim_halo_bytes(void* h, int kind, int64_t* bytes) { *bytes = mgc_halo_bytes(((HostDev*)h)->L, kind); return 0; }

int hostsim_halo_pack(void* h, int side, int kind, void* buf, int on_device)
{
    (void)on_device;
    HostDev* d = (HostDev*)h;
    HostBlock x(d->S);
    const int T = d->L.gy * d->L.gx;
    if (kind != 2) memset((char*)buf + mgc_halo_off_count(d->L), 0, 4);
    for (int i = 0; i < T; ++i) mgc_halo_pack_tile(x, d->L, side, kind, i, buf);
    return 0;
}

int hostsim_halo_unpack(void* h, int side, int kind, const void* buf, int on_device, uint32_t epoch, int list)
{
    (void)on_device;
    HostDev* d = (HostDev*)h;
    HostBlock x(d->S);
    const int T = d->L.gy * d->L.gx;
    for (int i = 0; i < T; ++i) mgc_halo_unpack_tile(x, d->L, side, kind, i, buf, epoch, list);
    return 0;
}

/* labels of the LOCAL planes (ghost planes included; the caller slices the owned range) */
int hostsim_labels(void* h, uint8_t* out) { ((HostDev*)h)->labels(out); return 0; }
void hostsim_set_halo_max(void* h, int which, int n);

/* ---- one-call convenience: single slab, the C++ schedule of mgc_driver.inl ---- */
int hostsim_solve(const int64_t* shape, const double* w0, const double* w1, const double* w2, const double* trcap,
                  int rounds, int cycles, int sweeps, int max_outer, uint8_t* labels_out, int64_t* stats_out)
{
    const int incremental = max_outer >= 0;
    if (max_outer < 0) max_outer = -max_outer - 1; /* negative: from-scratch relabels only (A/B tests) */
    HostDev* d = (HostDev*)hostsim_create(shape, 0, 1);
    d->load(w0, w1, w2, trcap);
    MgcSolveParams P = mgc_default_params();
    if (rounds > 0) P.rounds_per_relabel = rounds;
    if (cycles > 0) P.max_cycles = cycles;
    if (sweeps > 0) P.max_sweeps = sweeps;
    if (max_outer > 0) P.max_outer = max_outer;
    P.incremental_relabel = incremental;
    if (getenv("HOSTSIM_TRACE")) P.trace = atoi(getenv("HOSTSIM_TRACE")); /* one stderr line per global relabel (tools/sim_workprofile.py) */
    if (getenv("HOSTSIM_ADAPTIVE")) P.adaptive_rounds = atoi(getenv("HOSTSIM_ADAPTIVE"));
    if (getenv("HOSTSIM_RADIAL")) P.radial = atoi(getenv("HOSTSIM_RADIAL"));
    if (getenv("HOSTSIM_RADIAL_ROUNDS0")) P.radial_rounds0 = atoi(getenv("HOSTSIM_RADIAL_ROUNDS0"));
    if (getenv("HOSTSIM_RADIAL_MIN_C")) P.radial_min_c = atoi(getenv("HOSTSIM_RADIAL_MIN_C"));
    if (getenv("HOSTSIM_RADIAL_BUDGET")) P.radial_budget_x16 = atoi(getenv("HOSTSIM_RADIAL_BUDGET"));
    MgcSolveStats st;
    const int rc = mgc_solve(*d, d->L, P, st);
    memcpy(stats_out, &st, sizeof(st));
    d->labels(labels_out);
    delete d;
    return rc;
}

/* the first global relabel of a solve alone (use_dt: as a distance transform when the graph allows it, else by relaxation
 * passes): distance labels (tile-major) and status words out; returns 1 when the transform ran */
int hostsim_first_relabel(const int64_t* shape, const double* w0, const double* w1, const double* w2, const double* trcap,
                          int use_dt, int32_t* heights_out, uint32_t* status_out)
{
    HostDev* d = (HostDev*)hostsim_create(shape, 0, 1);
    d->load(w0, w1, w2, trcap);
    const int keep = g_use_dt;
    g_use_dt = use_dt & 1;
    MgcSolveParams P = mgc_default_params();
    P.max_outer = 1; /* one global relabel + the activation, then the rounds of colour phases: stop before them */
    P.rounds_per_relabel = 0;
    P.radial = (use_dt & 2) ? 1 : 0; /* bit 1: the radial labels of the flood phase on top of the transform (mgc_dt_ops.inl) */
    P.radial_budget_x16 = 0; /* (no colour rounds on the radial labels either) */
    if (use_dt & 4) P.radial_min_c = 1;
    MgcSolveStats st;
    g_prof[28] = 0;
    (void)mgc_solve(*d, d->L, P, st);
    const int ran = (int)g_prof[28];
    g_prof[28] = 0;
    g_use_dt = keep;
    memcpy(heights_out, d->height.data(), d->height.size() * sizeof(int32_t));
    memcpy(status_out, d->status.data(), d->status.size() * sizeof(uint32_t));
    delete d;
    return ran;
}

} /* extern "C" */

/* ------------------------------------------------------------------------------------------
 * 26-neighbourhood (mgc_tile_ops26.inl): one-call solve for the CPU tests
 * ---------------------------------------------------------------------------------------- */
struct HostDev26 : HostSlabOps<HostDev26, HostBlockT<MgcTileShared26D>> {
    MgcLattice L;
    MgcTileShared26D S;
    /* No transform TOWARDS the sink in the full neighbourhood (a Chebyshev distance is not separable): the first relabel runs as passes.
     * The radial labels of the flood phase only need the L1 transform AWAY from the source (mgc_radial_steps, mgc_dt_ops.inl): the scans
     * of the 6-neighbourhood on this lattice's excess plane, the library's HipDevT<true> ops on host arrays. */
    bool dt_applicable() { return false; }
    bool radial_after_passes() const { return true; }
    MgcWaveShared WSr;
    std::vector<uint16_t> ds16;
    std::vector<int32_t> hexact;
    std::vector<uint8_t> tsrc;
    uint16_t* dt_cur = nullptr;
    bool radial_on = false, lowered = false;
    int radial_cycle_no = 0;
    void set_radial(bool on) { radial_on = on; if (on) radial_cycle_no++; }
    void dt_scans_xy(int)
    {
        HostWave w(WSr);
        ds16.assign((size_t)L.ntiles * MGC_TV, 0);
        uint16_t* const T = dt_cur = ds16.data();
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, false, 2, 0>(w, L, i, L.excess, T);
        for (int i = 0; i < L.gz * L.gy; ++i) mgc_dt_scan_line<0, true, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, false, 0, 0>(w, L, i, T, T);
        for (int i = 0; i < L.gz * L.gx; ++i) mgc_dt_scan_line<1, true, 0, 0>(w, L, i, T, T);
    }
    void dt_scan_z(bool bwd, int final_kind, int c_min, const uint16_t* cin, bool want_out)
    {
        HostWave w(WSr);
        uint16_t* const T = dt_cur;
        uint16_t* const cout = want_out ? carry_buf(bwd ? 1 : 0) : nullptr;
        const int plane = want_out ? carry_plane(bwd ? 1 : 0) : -1;
        for (int i = 0; i < L.gy * L.gx; ++i) {
            if (!bwd) mgc_dt_scan_line<2, false, 0, 0>(w, L, i, T, T, 0, nullptr, cin, cout, plane);
            else if (final_kind == 2) mgc_dt_scan_line<2, true, 0, 2>(w, L, i, T, T, c_min, hexact.data(), cin, cout, plane);
            else mgc_dt_scan_line<2, true, 0, 0>(w, L, i, T, T, 0, nullptr, cin, cout, plane);
        }
        if (bwd && final_kind == 2) lowered = L.count[mgc_cnt_radial_c(L)] < MGC_HINF && L.count[mgc_cnt_radial_c(L)] >= c_min;
    }
    void dt_finish() {}
    void shadow_sync() {}
    bool radial_prepare() { if (spec.nranks > 1) return false; hexact.assign((size_t)L.ntiles * MGC_TV, 0); return true; }
    void radial_cmin()
    {
        HostWave w(WSr);
        L.count[mgc_cnt_radial_c(L)] = MGC_HINF;
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_cmin_tile(w, L, t);
    }
    void radial_swap() { std::swap(height, hexact); L.height = height.data(); }
    std::vector<double> rcap, excess, sink;
    std::vector<int32_t> height, lists, count;
    std::vector<uint32_t> rmask32, stamp, rstamp, status;
    bool first_relabel_dt() { return false; }
    bool radial_begin(int c_min)
    {
        if (getenv("HOSTSIM_RADIAL26") && !atoi(getenv("HOSTSIM_RADIAL26"))) return false;
        if (spec.nranks > 1) return false; /* (a slab's ghost tiles hold no excess to seed the transform from: single handles only) */
        radial_prepare();
        radial_cmin();
        dt_scans_xy(2);
        dt_scan_z(false, 0, 0, nullptr, false);
        dt_scan_z(true, 2, c_min, nullptr, false);
        radial_swap();
        return true;
    }
    void radial_save_exact() { hexact.assign(L.height, L.height + (size_t)L.ntiles * MGC_TV); }
    void radial_restore_exact() { std::swap(height, hexact); L.height = height.data(); lowered = false; }
    void radial_lower(int c_min)
    {
        HostWave w(WSr);
        lowered = L.count[mgc_cnt_radial_c(L)] < MGC_HINF && L.count[mgc_cnt_radial_c(L)] >= c_min;
        for (int t = 0; t < L.ntiles; ++t) mgc_dt_lower_tile(w, L, t, ds16.data(), c_min);
    }
    void source_open()
    {
        HostWave w(WSr);
        for (int t = 0; t < L.ntiles; ++t) mgc_source_open_tile(w, L, t);
    }
    void fill_heights_inf()
    {
        for (int64_t i = 0; i < (int64_t)L.ntiles * MGC_TV; ++i) L.height[i] = MGC_HINF;
        for (int t = 0; t < L.ntiles; ++t) L.status[t] = (L.status[t] & ~MGC_ST_SETTLED) | MGC_ST_ALLINF; /* until a relabel lowers a label */
    }
    void range_push(const char*) {}
    void range_pop() {}
    void zero_count(int i) { L.count[i] = 0; }
    void read_counts(int* out) { memcpy(out, L.count, MGC_NCOUNT * sizeof(int)); }
    void absorb_all() {}
    int suspect_batch() const { return 8; }
    void suspect_pass()
    {
        for (int t = 0; t < L.ntiles; ++t)
            if (mgc26_suspect_tile(L, t)) L.count[MGC_CNT_CHANGED] = 1;
    }
    void reset_suspect(uint32_t epoch, int list)
    {
        HostBlock26 x(S);
        if (getenv("HOSTSIM_TRACE26")) {
            int sus = 0, dirty = 0;
            for (int t = 0; t < L.ntiles; ++t) { sus += (L.status[t] & MGC_ST_SUSPECT) != 0; dirty += (L.status[t] & MGC_ST_DIRTY) != 0; }
            fprintf(stderr, "\n[relabel: %d of %d tiles suspect, %d dirty] ", sus, L.ntiles, dirty);
            int hd[16] = {0}, hs[16] = {0}, ha[16] = {0};
            for (int t = 0; t < L.ntiles; ++t) {
                int tx = t % L.gx, ty = (t / L.gx) % L.gy, tz = t / (L.gx * L.gy);
                int sh = std::min(std::min(std::min(tx, L.gx - 1 - tx), std::min(ty, L.gy - 1 - ty)), std::min(tz, L.gz - 1 - tz));
                ha[sh]++; hd[sh] += (L.status[t] & MGC_ST_DIRTY) != 0; hs[sh] += (L.status[t] & MGC_ST_SUSPECT) != 0;
            }
            for (int i = 0; i < 16 && ha[i]; ++i) fprintf(stderr, "{shell %d: %d tiles, %d dirty, %d suspect} ", i, ha[i], hd[i], hs[i]);
        }
        for (int t = 0; t < L.ntiles; ++t) mgc26_reset_suspect_tile(x, L, t, epoch, list);
    }
    void relabel_all(uint32_t epoch, int next)
    {
        HostBlock26 x(S);
        for (int t = 0; t < L.ntiles; ++t) {
            if (L.status[t] & 2u) L.count[MGC26_CNT_REL]++;
            mgc26_relabel_tile(x, L, t, epoch, next, true);
        }
    }
    void relabel_list(int lst, uint32_t epoch, int next, int = -1)
    {
        HostBlock26 x(S);
        const int n = L.count[lst];
        L.count[MGC26_CNT_REL] += n;
        for (int i = 0; i < n; ++i) mgc26_relabel_tile(x, L, L.list[lst][i], epoch, next, false);
    }
    /* hostsim_set_check_exact: after every global relabel the labels are compared with the exact distances to the sink in the
     * residual graph (plain relaxation over the whole volume): g_prof[40] relabels checked, g_prof[41] voxels that differ.
     * (A single device only: a slab would need its neighbours' residuals.) */
    void check_exact_labels()
    {
        const int64_t N = L.nvox;
        std::vector<int32_t> dist((size_t)N, MGC_HINF), tl((size_t)N), lc((size_t)N);
        for (int64_t id = 0; id < N; ++id) {
            int tile, loc;
            mgc_node_to_tile(L, id, tile, loc);
            tl[id] = tile; lc[id] = loc;
            if (L.sink[(int64_t)tile * MGC_TV + loc] > 0.0) dist[id] = 1;
        }
        for (bool changed = true; changed;) {
            changed = false;
            for (int64_t id = 0; id < N; ++id) {
                const int64_t x0 = id % L.dx, y0 = (id / L.dx) % L.dy, z0 = id / (L.dx * L.dy);
                int32_t best = dist[id];
                for (int d = 0; d < 26; ++d) {
                    if (!(L.rcap[((int64_t)tl[id] * 26 + d) * MGC_TV + lc[id]] > 0.0)) continue;
                    int dz, dy, dx;
                    mgc26_offset(d, dz, dy, dx);
                    const int64_t z = z0 + dz, y = y0 + dy, xx = x0 + dx;
                    if (z < 0 || z >= L.dz || y < 0 || y >= L.dy || xx < 0 || xx >= L.dx) continue;
                    const int32_t hv = dist[(z * L.dy + y) * L.dx + xx];
                    if (hv < MGC_HINF && hv + 1 < best) best = hv + 1;
                }
                if (best < dist[id]) { dist[id] = best; changed = true; }
            }
        }
        g_prof[40]++;
        for (int64_t id = 0; id < N; ++id) g_prof[41] += L.height[(int64_t)tl[id] * MGC_TV + lc[id]] != dist[id];
    }
    void activate_all(uint32_t phase)
    {
        if (g_check_exact) check_exact_labels();
        HostBlock26 x(S);
        for (int t = 0; t < L.ntiles; ++t)
            if (mgc26_activate_tile(x, L, t, phase)) L.count[MGC26_CNT_ACTIVE]++;
    }
    MgcWaveShared26 WS;
    void discharge(int lst, uint32_t phase, int cycles, int sweeps)
    {
        HostBlock26D x(S);
        HostWave26 w(WS);
        const int n = L.count[lst];
        L.count[MGC26_CNT_DIS] += n;
        if (getenv("HOSTSIM_TRACE26")) fprintf(stderr, "%d ", n);
        for (int i = 0; i < n; ++i) {
            if ((g_wave_mode & 16) && cycles < 0) mgcw26_discharge_tile(w, L, L.list[lst][i], phase, sweeps, g_w26_passes, g_w26_raises, g_w26_flags | (radial_on ? MGCW26_SAT_DIRTY : 0)); /* one wave per tile, stored labels */
            else mgc26_discharge_tile(x, L, L.list[lst][i], phase, radial_on && cycles < 0 ? cycles - 1024 : cycles, sweeps);
        }
    }
    std::vector<uint32_t> oflags;
    MgcSlabSpec spec;

    void init(int64_t d0, int64_t d1, int64_t d2, const MgcSlabSpec* sp)
    {
        memset(&L, 0, sizeof(L));
        L.dz = d0; L.dy = d1; L.dx = d2;
        L.nvox = d0 * d1 * d2;
        L.gz = (int)((d0 + 7) / 8); L.gy = (int)((d1 + 7) / 8); L.gx = (int)((d2 + 7) / 8);
        L.ntiles = L.gz * L.gy * L.gx;
        L.tz_own_lo = 0; L.tz_own_hi = L.gz; L.tz_global0 = 0; L.ndir = 26;
        if (sp) { spec = *sp; L.tz_own_lo = sp->own_lo; L.tz_own_hi = sp->own_hi; L.tz_global0 = sp->tz_global0; }
        else { memset(&spec, 0, sizeof(spec)); spec.nranks = 1; spec.own_hi = L.gz; spec.plane1 = spec.own1 = d0; }
        const int64_t nt = L.ntiles;
        rcap.assign(nt * 26 * MGC_TV, 0.0); excess.assign(nt * MGC_TV, 0.0); sink.assign(nt * MGC_TV, 0.0);
        height.assign(nt * MGC_TV, MGC_HINF); lists.assign(18 * nt, 0); count.assign(MGC_NCOUNT, 0);
        rmask32.assign(nt * MGC_TV, 0); stamp.assign(nt, 0); rstamp.assign(nt, 0); status.assign(nt, 0); oflags.assign(nt, 0);
        tsrc.assign(nt, 0);
        L.rcap = rcap.data(); L.excess = excess.data(); L.sink = sink.data(); L.height = height.data();
        L.rmask32 = rmask32.data(); L.oflags = oflags.data(); L.tsrc = tsrc.data();
        for (int i = 0; i < 18; ++i) L.list[i] = lists.data() + i * nt;
        L.count = count.data(); L.stamp = stamp.data(); L.rstamp = rstamp.data(); L.status = status.data();
        L.scount = L.count; L.nshard = 1; L.shard_cap = (int)nt;
        L.halo_max_rec = L.gy * L.gx; /* a border message holds every border tile unless a test says otherwise (hostsim_set_halo_max) */
    }

    /* w[d*N + id] = capacity of the arc from LOCAL voxel id in direction d (0..25, mgc26_offset order), 0 where there
     * is no neighbour; trcap[N].  Ghost tiles keep zero excess / residuals: they accumulate pushes over the border. */
    void load(const double* w, const double* trcap)
    {
        const int64_t N = L.nvox;
        for (int64_t id = 0; id < N; ++id) {
            int tile, loc;
            mgc_node_to_tile(L, id, tile, loc);
            if (!mgc_owned(L, tile)) continue;
            uint32_t m = 0;
            for (int dir = 0; dir < 26; ++dir) {
                const double c = w[(int64_t)dir * N + id];
                rcap[((int64_t)tile * 26 + dir) * MGC_TV + loc] = c;
                if (c > 0.0) m |= 1u << dir;
            }
            const double tr = trcap[id];
            excess[(int64_t)tile * MGC_TV + loc] = tr > 0 ? tr : 0.0;
            if (tr > 0) tsrc[tile] |= 1u;
            sink[(int64_t)tile * MGC_TV + loc] = tr < 0 ? -tr : 0.0;
            if (tr < 0) { m |= MGC26_MASK_SINK; status[tile] |= 2u; }
            rmask32[(int64_t)tile * MGC_TV + loc] = m;
        }
    }

    void labels(uint8_t* out)
    {
        for (int64_t id = 0; id < L.nvox; ++id) {
            int tile, loc;
            mgc_node_to_tile(L, id, tile, loc);
            out[id] = height[(int64_t)tile * MGC_TV + loc] < MGC_HINF ? 0 : 1;
        }
    }
};

extern "C" {

int hostsim_solve26(const int64_t* shape, const double* w, const double* trcap, int rounds, int cycles, int sweeps, int max_outer,
                    uint8_t* labels_out, int64_t* stats_out)
{
    HostDev26* d = new HostDev26();
    d->init(shape[0], shape[1], shape[2], NULL);
    d->load(w, trcap);
    MgcSolveParams P = mgc_default_params(26);
    if (rounds > 0) P.rounds_per_relabel = rounds;
    if (cycles != 0) P.max_cycles = cycles; /* < 0: stored labels instead of the exact in-tile labelling */
    if (sweeps > 0) P.max_sweeps = sweeps;
    if (max_outer > 0) P.max_outer = max_outer;
    if (getenv("HOSTSIM_RADIAL")) P.radial = atoi(getenv("HOSTSIM_RADIAL"));
    if (getenv("HOSTSIM_RADIAL_BUDGET")) P.radial_budget_x16 = atoi(getenv("HOSTSIM_RADIAL_BUDGET"));
    if (getenv("HOSTSIM_TRACE")) P.trace = atoi(getenv("HOSTSIM_TRACE"));
    MgcSolveStats st;
    const int rc = mgc_solve(*d, d->L, P, st, mgc_layout26());
    memcpy(stats_out, &st, sizeof(st));
    d->labels(labels_out);
    delete d;
    if (getenv("HOSTSIM_PROF26")) fprintf(stderr, "26-neighbourhood discharge: %lld sweeps, %.2f of 26 directions per sweep; wave form: %lld sweeps (pass A), %lld visits\n", g26_sweeps, g26_sweeps ? (double)g26_dirs / g26_sweeps : 0.0, (long long)g_prof[1], (long long)g_prof[0]);
    g26_sweeps = g26_dirs = 0;
    g_prof[0] = g_prof[1] = 0;
    return rc;
}

/* ---- handle API of a 26-neighbourhood slab (same calls as the 6-neighbourhood one above) ---- */
void* hostsim26_create(const int64_t* gshape, int rank, int nranks)
{
    HostDev26* d = new HostDev26();
    d->gd0 = gshape[0];
    if (nranks <= 1) {
        d->init(gshape[0], gshape[1], gshape[2], NULL);
    } else {
        MgcSlabSpec sp;
        if (mgc_slab_spec(gshape[0], rank, nranks, &sp)) { delete d; return NULL; }
        d->init(sp.plane1 - sp.plane0, gshape[1], gshape[2], &sp);
    }
    return d;
}

void hostsim26_destroy(void* h) { delete (HostDev26*)h; }

int hostsim26_slab_info(void* h, int64_t* info)
{
    HostDev26* d = (HostDev26*)h;
    info[0] = d->spec.plane0; info[1] = d->spec.plane1; info[2] = d->spec.own0; info[3] = d->spec.own1;
    info[4] = d->L.tz_own_lo > 0; info[5] = d->L.tz_own_hi < d->L.gz; info[6] = (int64_t)d->L.gy * d->L.gx; info[7] = d->spec.nranks;
    return 0;
}

int hostsim26_load(void* h, const double* w, const double* trcap) { ((HostDev26*)h)->load(w, trcap); return 0; }

int hostsim26_solver_op(void* h, int op, int64_t a0, int64_t a1, int64_t a2, int64_t a3)
{
    HostDev26* d = (HostDev26*)h;
    switch (op) {
    case 0: d->absorb_all(); break;
    case 1: d->fill_heights_inf(); break;
    case 2: d->zero_count((int)a0); break;
    case 3: d->relabel_all((uint32_t)a0, (int)a1); break;
    case 4: d->relabel_list((int)a0, (uint32_t)a1, (int)a2); break;
    case 5: d->activate_all((uint32_t)a0); break;
    case 6: d->discharge((int)a0, (uint32_t)a1, (int)a2, (int)a3); break;
    case 7: d->suspect_pass(); break;
    case 8: d->reset_suspect((uint32_t)a0, (int)a1); break;
    default: return 1;
    }
    return 0;
}

int hostsim26_read_counts(void* h, int32_t* out) { ((HostDev26*)h)->read_counts(out); return 0; }
int hostsim26_halo_bytes(void* h, int kind, int64_t* bytes) { *bytes = mgc_halo_bytes_nd(((HostDev26*)h)->L, kind); return 0; }

int hostsim26_halo_pack(void* h, int side, int kind, void* buf, int on_device)
{
    (void)on_device;
    HostDev26* d = (HostDev26*)h;
    HostBlock26 x(d->S);
    const int T = d->L.gy * d->L.gx;
    if (kind == 1) memset((char*)buf + mgc26_halo_off_count(d->L), 0, 4);
    for (int i = 0; i < T; ++i) mgc_halo_pack_nd(x, d->L, side, kind, i, buf);
    return 0;
}

int hostsim26_halo_unpack(void* h, int side, int kind, const void* buf, int on_device, uint32_t epoch, int list)
{
    (void)on_device;
    HostDev26* d = (HostDev26*)h;
    HostBlock26 x(d->S);
    const int T = d->L.gy * d->L.gx;
    for (int i = 0; i < T; ++i) mgc_halo_unpack_nd(x, d->L, side, kind, i, buf, epoch, list);
    return 0;
}

int hostsim26_labels(void* h, uint8_t* out) { ((HostDev26*)h)->labels(out); return 0; }

} /* extern "C" */

/* The slabs of one volume through mgc_solve (mgc_driver.inl) -- the schedule the library runs for them (mgc_solve_slabs), with the same
 * group and transport code: n = all slabs of the volume (handles of this process), or n = 1 with the callbacks of a host transport.
 * params[0..7]: rounds_per_relabel, max_cycles, max_sweeps, max_outer, incremental_relabel, exchange_passes, radial (-1: the default), exchange_rounds
 * (0 / as noted: the default).  stats_out[0..15]: MgcSolveStats, then exchanges, reductions. */
template <class Dev>
static int hostsim_solve_slabs_on(void** handles, int n, const mgc_transport* cb, const int64_t* params, const MgcLayout lay, int ndir, int64_t* stats_out)
{
    std::vector<Dev*> ptr((size_t)n);
    for (int i = 0; i < n; ++i) ptr[(size_t)i] = (Dev*)handles[i];
    const bool all_local = n == (ptr[0]->spec.nranks > 0 ? ptr[0]->spec.nranks : 1);
    MgcXchg<Dev> x(ptr, cb, all_local);
    MgcSlabGroup<Dev, MgcXchg<Dev>> group(ptr, x);
    MgcSolveParams P = mgc_default_params(ndir);
    if (params[0] > 0) P.rounds_per_relabel = (int)params[0];
    if (params[1] != 0) P.max_cycles = (int)params[1];
    if (params[2] > 0) P.max_sweeps = (int)params[2];
    if (params[3] > 0) P.max_outer = (int)params[3];
    P.incremental_relabel = params[4] != 0;
    if (params[5] > 0) P.exchange_passes = (int)params[5];
    if (params[6] >= 0) P.radial = (int)params[6];
    if (params[7] > 0) P.exchange_rounds = (int)params[7];
    if (getenv("HOSTSIM_TRACE")) P.trace = atoi(getenv("HOSTSIM_TRACE"));
    if (getenv("HOSTSIM_RADIAL_MIN_C")) P.radial_min_c = atoi(getenv("HOSTSIM_RADIAL_MIN_C"));
    MgcSolveStats st;
    const int rc = mgc_solve(group, ptr[0]->L, P, st, lay);
    memset(stats_out, 0, 16 * sizeof(int64_t));
    memcpy(stats_out, &st, sizeof(st) < 12 * sizeof(int64_t) ? sizeof(st) : 12 * sizeof(int64_t));
    stats_out[12] = group.exchanges;
    stats_out[13] = group.reductions;
    stats_out[14] = x.error;
    return x.error ? 100 + x.error : rc;
}

extern "C" int hostsim_solve_slabs(void** handles, int n, int ndir, const mgc_transport* cb, const int64_t* params, int64_t* stats_out)
{
    if (ndir == 26) return hostsim_solve_slabs_on<HostDev26>(handles, n, cb, params, mgc_layout26(), 26, stats_out);
    return hostsim_solve_slabs_on<HostDev>(handles, n, cb, params, mgc_layout6(), 6, stats_out);
}

extern "C" void hostsim_set_halo_max(void* h, int which, int n)
{
    MgcLattice& L = which == 26 ? ((HostDev26*)h)->L : ((HostDev*)h)->L;
    const int T = L.gy * L.gx;
    L.halo_max_rec = n < 1 ? 1 : (n < T ? n : T);
}
