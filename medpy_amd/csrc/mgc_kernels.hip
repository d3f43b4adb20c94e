/*
 * mgc_kernels.hip -- HIP (gfx950 / CDNA4) kernels and the C ABI of libmedpyhip.so.
 *
 * Written for MI355X only: wave64, 8x8x8-voxel tiles = 512-thread workgroups (8 waves, one
 * z-slice of the tile per wave), tile-major f64 state so every global access of the solver is
 * a contiguous 4 KiB stream, LDS for the label halo and the push hand-off, device-resident
 * work lists so only active tiles cost anything.  No MFMA: this is HBM/LDS-bound index and
 * f64 add/min work.
 *
 * Kernels
 *   k_minmax        intensity range for the *_linear terms (energy_voxel.py:101,174-176)
 *   k_build         n-link weights g(|Ip-Iq|) / g(max) for the 6 neighbours of every voxel
 *                   (energy_voxel.py:611-664) + merged t-links (energy_voxel.py:61-65,
 *                   graph.h:416-425, generate.py:169-172) -> tile-major residual graph
 *   k_absorb / k_relabel_all / k_relabel_list / k_activate / k_discharge
 *                   the solver (bodies: mgc_tile_ops.inl; schedule: mgc_driver.inl)
 *   k_labels        what_segment read-out (graph.h:561-571, bin/medpy_graphcut_voxel.py:177-181)
 *   k_cut_value     capacity of the cut = the value maxflow() returns
 *   k_get_*         energy read-back for the parity tests
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> /* types only: the library itself is dlopen()ed on first use */

#include <dlfcn.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/medpy_hip.h"
#include "mgc_tile_ops.inl"
#include "mgc_tile_ops26.inl"
#include "mgc_wave_ops.inl"
#include "mgc_wave_ops26.inl"
#include "mgc_dt_ops.inl"
#include "mgc_brick_ops.inl"
#include "mgc_terms.h"
#include "mgc_driver.inl"

#define MGC_MARKER_MAX 65535.0              /* GCGraph.MAX, graph.py:288-291 */

#ifndef MGCW_DPP
#define MGCW_DPP 1      /* +-x hand-offs of the wave discharge as DPP row shifts (0: ds_bpermute like +-y) */
#endif
#ifndef MGCW_RUNAHEAD
#define MGCW_RUNAHEAD 0 /* 1: a discharging wave draws the ticket for its NEXT tile while it still works on the current one (and may
                           prefetch it, MGCW_PREFETCH).  Measured on MI355X at 512^3: hides three dependent memory trips per visit
                           (ticket, list entry, status word) but every wave then sits on two tiles, and with ~3 visits per wave
                           and launch the last tiles of a phase start a whole visit late: 197 us per launch instead of 165.  Off. */
#endif
#ifndef MGCW_PREFETCH
#define MGCW_PREFETCH 0 /* (with MGCW_RUNAHEAD) 1: the wave requests the excess / labels / masks of its next tile while it sweeps; 2: the
                           residual planes too.  Measured: no gain -- a visit is bound by dependent trips, not by bytes */
#endif

/* ======================================================================================
 * block executor for the single-source tile operations
 * ==================================================================================== */
template <class SH, bool LAUNDER_EVERY_STEP = false>
struct GpuBlockT {
    template <class T>
    struct Reg {
        T v;
        __device__ __forceinline__ T& operator[](int) { return v; }
    };
    SH& S;
    int tid;
    __device__ __forceinline__ explicit GpuBlockT(SH& s) : S(s), tid((int)threadIdx.x) {}
    /* Call at the top of every iteration of a kernel's tile loop: makes the lane id opaque to the optimiser, so
     * lane-dependent addresses are recomputed per tile instead of being hoisted out of the loop and kept alive
     * (that hoisting cost ~40 VGPRs, i.e. scratch spills written once per workgroup: 400 MB per launch) */
    __device__ __forceinline__ void new_tile()
    {
        tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
    }
    /* MGC_LAUNDER_EVERY_STEP: additionally launder the lane id at every step (fewer spills, 44 vs 100 B/lane, but
     * measured 3 % slower on MI355X: 115 vs 112 ms at 512^3) */
    __device__ __forceinline__ int lane() const
    {
        int t = tid;
        /* LAUNDER_EVERY_STEP: the lane id is opaque at every step, so nothing derived from it (the 26 x {inside predicate,
         * LDS address, halo offset} of the full neighbourhood) is hoisted out of the sweep loop and kept alive */
        if (LAUNDER_EVERY_STEP) asm volatile("" : "+v"(t));
        return t;
    }
    template <class F>
    __device__ __forceinline__ void par(F f)
    {
        f(lane());
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ bool any(F f)
    {
        return __syncthreads_or((int)f(lane())) != 0;
    }
    /* wave-local step: touches only the lane's own registers / LDS slots, no workgroup barrier */
    template <class F>
    __device__ __forceinline__ void wpar(F f)
    {
        f(lane());
    }
    /* vote inside the wave (one z-layer of the tile): uniform per wave, may differ between waves */
    template <class F>
    __device__ __forceinline__ bool wave_any(F f)
    {
        return __any((int)f(lane())) != 0;
    }
    /* dst[t] = src[t + delta] for lanes of the same wave (same z-layer), 0.0 at the wave's ends: ds_bpermute, no LDS
     * storage, no barrier.  Collective: call it from wave-uniform control flow. */
    __device__ __forceinline__ void shift(Reg<double>& dst, Reg<double>& src, int delta)
    {
        const int from = (int)(threadIdx.x & 63u) + delta;
        const double v = __shfl(src.v, from & 63, 64);
        dst.v = (from >= 0 && from < 64) ? v : 0.0;
    }
    __device__ __forceinline__ int atomic_add(int32_t* p, int v) { return atomicAdd(p, v); }
    __device__ __forceinline__ uint32_t atomic_exch(uint32_t* p, uint32_t v) { return atomicExch(p, v); }
    __device__ __forceinline__ void atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
    /* the region of a work list this workgroup appends to (MgcLattice::scount): workgroup ids go round the XCDs */
    __device__ __forceinline__ int shard(const MgcLattice& L) const { return (int)(blockIdx.x & (unsigned)(L.nshard - 1)); }
    /* a value every lane of the workgroup holds alike (read from LDS after a barrier): scalar for the branches on it */
    __device__ __forceinline__ uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ void atomic_and(uint32_t* p, uint32_t v) { atomicAnd(p, v); }
    /* ---- exact in-tile labels (executor primitive, see mgc_tile_ops.inl) ----
     * Chaotic relaxation in LDS (mgc_tile_bfs).  A bit-parallel level-synchronous BFS (lane = (y,x) column, bit z of
     * a byte = voxel, four ds_bpermute + two shifts per level) was implemented and verified label-for-label against
     * this form on MI355X, but measured SLOWER: 18.3k cycles per BFS when every wave ran it redundantly, 20.9k with
     * one wave per tile, vs 15.2k for the relaxation -- ~40 dependent levels x (bpermute latency + a lone wave's
     * issue rate) cost more than ~24 barrier rounds.  Removed again; the numbers are kept in profiles/README.md. */
    template <class MaskFn, class RegI>
    __device__ __forceinline__ void tile_labels(MaskFn mask, RegI& out)
    {
        int m = 0; /* the residual mask is evaluated once, not once per relaxation round */
        par([&](int t) {
            S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)] = MGC_HINF;
            m = mask(t);
        });
        mgc_tile_bfs(*this, [&](int) { return m; });
        const int t = lane(); /* own cell, last written by this lane before the barrier that ended the relaxation */
        out[t] = S.hs[mgc_hs_index(t >> 6, (t >> 3) & 7, t & 7)];
    }
    /* HBM -> LDS copy without a VGPR round trip: every wave moves 1 KiB chunks (64 lanes x 16 B) with
     * global_load_lds_dwordx4; `bytes` must be a multiple of 1024 and both pointers 16-byte aligned.  Tracked by
     * vmcnt: async_wait() before the barrier that publishes the data. */
    __device__ __forceinline__ void async_to_lds(int, void* lds_dst, const void* gsrc, int bytes)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int c = wave; c < bytes / 1024; c += MGC_TV / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)gsrc + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)((char*)lds_dst + c * 1024), 16, 0, 0);
    }
    __device__ __forceinline__ void async_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    /* order this wave's LDS accesses (lockstep execution makes them visible to its own lanes) */
    __device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    /* development aid: cycles spent since the previous mark go to section `id` (count in id + 8).  Accumulated in
     * lane 0's registers and flushed once per workgroup (flush_marks), so the probe does not perturb the kernel. */
    /* += / |= on global memory as atomics WITHOUT return: the value is not needed, so nothing waits for the trip to HBM (a load +
     * add + store in the middle of a step cost the whole workgroup that trip at its next barrier).  One writer per address and
     * step, and an f64 atomic add is the IEEE sum (tools/probes/atomic_f64_probe.hip): the result is what += gives. */
    __device__ __forceinline__ void gadd(double* p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void gor(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    unsigned long long last = 0;
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void mark(const MgcLattice& L, int id)
    {
        if (L.prof && threadIdx.x == 0) {
            const unsigned long long now = clock64();
            if (last) { acc[id] += now - last; cnt[id]++; }
            last = now;
        }
    }
    __device__ __forceinline__ void flush_marks(const MgcLattice& L)
    {
        if (L.prof && threadIdx.x == 0)
            for (int i = 0; i < 8; ++i)
                if (cnt[i]) { atomicAdd(&L.prof[i], acc[i]); atomicAdd(&L.prof[i + 8], (unsigned long long)cnt[i]); }
    }
};
typedef GpuBlockT<MgcTileShared> GpuBlock;
typedef GpuBlockT<MgcTileShared26> GpuBlock26;
#ifndef MGC26_LAUNDER
#define MGC26_LAUNDER true
#endif
typedef GpuBlockT<MgcTileShared26D, MGC26_LAUNDER> GpuBlock26D;


/* ======================================================================================
 * wave executor for the one-wave-per-tile operations (mgc_wave_ops.inl): a workgroup IS one wave64,
 * Reg<T, N> = N registers, votes = ballots, no workgroup barrier anywhere
 * ==================================================================================== */
template <class SH> /* MgcWaveShared (discharge: labels, inbox, sink links) or MgcWaveSharedR (relabel: labels only) */
struct GpuWaveT {
    template <class T, int N>
    struct Reg {
        T v[N];
        __device__ __forceinline__ T& operator()(int, int k) { return v[k]; }
    };
    /* N doubles per lane that live in ACCUMULATOR registers (a wave that runs alone on its SIMD owns 256 of them on top of its
     * 256 vector registers): moved with v_accvgpr_read / v_accvgpr_write around every use.  The "a" constraints keep the value
     * in the accumulator file between the statements; they are not volatile, so unused moves disappear. */
    template <int N>
    struct RegA {
        int lo[N], hi[N];
        __device__ __forceinline__ double get(int, int k) const
        {
            int l, h;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo[k])); /* volatile: a read is cheaper than keeping its result alive, */
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi[k])); /* which is what common-subexpression elimination would do      */
            return __hiloint2double(h, l);
        }
        __device__ __forceinline__ void init(int, int k, double v) /* the first value (nothing is read) */
        {
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo[k]) : "v"(__double2loint(v))); /* volatile: stays where it is written (the */
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi[k]) : "v"(__double2hiint(v))); /* optimiser sank these into the first uses)  */
        }
        /* "+a": the new value is tied to the register of the old one, so a value that is updated on one side of a branch needs no
         * copy where the paths join (with "=a" every update is a new value that the allocator has to merge back: the accumulator
         * file overflowed into scratch memory) */
        __device__ __forceinline__ void set(int, int k, double v)
        {
            asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(lo[k]) : "v"(__double2loint(v)));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(hi[k]) : "v"(__double2hiint(v)));
        }
    };
    SH& S;
    int lane;
    __device__ __forceinline__ explicit GpuWaveT(SH& s) : S(s), lane((int)threadIdx.x) {}
    /* top of every tile: the lane id becomes opaque to the optimiser, so lane-dependent addresses and masks are recomputed
     * per tile (a few VALU ops) instead of being hoisted out of the tile loop and kept alive in registers that the
     * discharge needs for its state */
    __device__ __forceinline__ void new_tile()
    {
        lane = (int)threadIdx.x;
        asm volatile("" : "+v"(lane));
        lane &= 63; /* the range stays known: constant parts of an index fold into the instruction's offset field */
    }
    /* orders this wave's LDS accesses for the compiler; the hardware executes one wave's LDS instructions in order */
    __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    template <class F>
    __device__ __forceinline__ void lanes(F f)
    {
        f(lane);
        fence();
    }
    template <class F>
    __device__ __forceinline__ bool any(F f)
    {
        return __ballot((int)f(lane)) != 0ull;
    }
    __device__ __forceinline__ void shift(Reg<double, 1>& dst, Reg<double, 1>& src, int delta)
    {
        const int from = lane + delta;
        const double v = __shfl(src.v[0], from & 63, 64);
        dst.v[0] = (from >= 0 && from < 64) ? v : 0.0;
    }
    /* delta = +-1, src 0.0 wherever the source lane lies in another row of eight: a DPP shift inside the 16-lane rows of the
     * wave (row_shl / row_shr, zero fill at the row ends) moves both halves of the double in two VALU instructions; what
     * crosses from one row of eight into the other inside a 16-lane row is one of those zeros */
    __device__ __forceinline__ void shift_x(Reg<double, 1>& dst, Reg<double, 1>& src, int delta)
    {
#if MGCW_DPP
        const int lo = __double2loint(src.v[0]), hi = __double2hiint(src.v[0]);
        int rlo, rhi;
        if (delta > 0) { /* dst(l) = src(l + 1): row_shl:1 */
            rlo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);
            rhi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, true);
        } else { /* dst(l) = src(l - 1): row_shr:1 */
            rlo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);
            rhi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
        }
        dst.v[0] = __hiloint2double(rhi, rlo);
#else
        shift(dst, src, delta);
#endif
    }
    /* ---- running ahead of the tile loop (mgc_wave_ops.inl: W::kPrefetch) ---- */
    static constexpr int kPrefetch = MGCW_RUNAHEAD ? MGCW_PREFETCH : -1;
    int32_t* pf = nullptr;          /* 1 KiB of LDS the prefetch DMA lands in (never read) */
    const int32_t* nlist = nullptr; /* the list being consumed */
    int tkv = 0, lsv = -1;          /* in flight: the ticket just drawn (lane 0), the list entry it points at */
    /* The list being consumed is cut into regions (MgcLattice::scount); a ticket is an index into their concatenation.  The
     * region lengths sit in lanes 0 .. nshard - 1 of one register for the whole launch (nobody appends to a list while it is
     * consumed).  The ticket word itself stays ONE word: its atomic is issued when a visit starts and its value is only
     * needed in the middle of the visit, so the queueing on it is hidden; what had to be spread out are the appends. */
    const MgcLattice* lat = nullptr;
    int lens = 0;
    __device__ __forceinline__ void list_begin(const MgcLattice& L, int lst)
    {
        lat = &L;
        nlist = L.list[lst];
        lens = 0;
        if ((int)threadIdx.x < L.nshard) lens = *mgc_counter(L, lst, (int)threadIdx.x);
    }
    /* entry i of the concatenated regions, or -1 beyond the end */
    __device__ __forceinline__ int entry_load(int i) const
    {
        int sh = 0, off = i, found = 0;
#pragma unroll
        for (int k = 0; k < MGC_NSHARD; ++k) {
            const int len = __builtin_amdgcn_readlane(lens, k);
            if (!found) {
                if (off < len) { sh = k; found = 1; }
                else off -= len;
            }
        }
        int v = -1;
        if (found) v = nlist[(int64_t)sh * lat->shard_cap + off];
        return v;
    }
    int tk = 0; /* ticket word of this launch */
    __device__ __forceinline__ void ticket_issue(const MgcLattice& L)
    {
        tkv = 0;
        if (threadIdx.x == 0) tkv = atomicAdd(&L.count[tk], 1);
    }
    __device__ __forceinline__ void hint_begin()
    {
        lsv = entry_load((int)gridDim.x + __builtin_amdgcn_readfirstlane(tkv));
    }
    int nst = 0; /* in flight: the status word of that tile */
    int next_tile = -1;
    __device__ __forceinline__ int hint_end(const MgcLattice& L)
    {
        next_tile = __builtin_amdgcn_readfirstlane(lsv);
        nst = 0;
        if (next_tile >= 0) nst = (int)L.status[next_tile];
        return next_tile;
    }
    /* [p, p + bytes) -> L2 / Infinity Cache by way of LDS-DMA loads that nobody waits for (bytes a multiple of 256) */
    __device__ __forceinline__ void prefetch(const void* p, int bytes)
    {
        const char* g = (const char*)p;
        int c = 0;
        for (; c + 1024 <= bytes; c += 1024)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c + lane * 16), (__attribute__((address_space(3))) void*)pf, 16, 0, 0);
        for (; c + 256 <= bytes; c += 256)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c + lane * 4), (__attribute__((address_space(3))) void*)pf, 4, 0, 0);
    }
    __device__ __forceinline__ int atomic_add(int32_t* p, int v) { return atomicAdd(p, v); }
    __device__ __forceinline__ uint32_t atomic_exch(uint32_t* p, uint32_t v) { return atomicExch(p, v); }
    __device__ __forceinline__ void atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
    __device__ __forceinline__ void atomic_min(int32_t* p, int v) { atomicMin(p, v); }
    __device__ __forceinline__ int shard(const MgcLattice& L) const { return (int)(blockIdx.x & (unsigned)(L.nshard - 1)); }
    /* a value every lane of the workgroup holds alike (read from LDS after a barrier): scalar for the branches on it */
    __device__ __forceinline__ uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ void atomic_and(uint32_t* p, uint32_t v) { atomicAnd(p, v); }
    /* ---- mgc_wave_ops26.inl ---- */
    /* OR over the wave: four DPP row shifts leave the OR of every 16-lane row in its last lane, four v_readlane collect them */
    template <class F>
    __device__ __forceinline__ uint32_t wave_or(F f)
    {
        int v = (int)f(lane);
        v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); /* row_shr:1 */
        v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); /* row_shr:2 */
        v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); /* row_shr:4 */
        v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); /* row_shr:8 */
        return (uint32_t)(__builtin_amdgcn_readlane(v, 15) | __builtin_amdgcn_readlane(v, 31) | __builtin_amdgcn_readlane(v, 47) | __builtin_amdgcn_readlane(v, 63));
    }
    /* a wave-uniform word per k, all of them in one register: lane k holds word k */
    __device__ __forceinline__ void uput(Reg<int, 1>& store, int k, uint32_t v) { store.v[0] = lane == k ? (int)v : store.v[0]; }
    __device__ __forceinline__ uint32_t uget(Reg<int, 1>& store, int k) { return (uint32_t)__builtin_amdgcn_readlane(store.v[0], k); }
    /* LDS read-modify-write of a word only this lane touches: ds_and / ds_or without return, nobody waits */
    __device__ __forceinline__ void lds_and(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    /* global word no other wave touches during the launch: memory atomics without return (one wave's atomics on one address
     * are performed in the order it issued them) */
    __device__ __forceinline__ void gadd(double* p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }
    /* the loads issued so far are not moved below this point, nor later ones above it */
    __device__ __forceinline__ void load_batch_end() { __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void gor(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#if defined(MGCW_PROFILE) /* development build (tools/gpu_sections.py): cycles per section of a tile discharge, per wave.  The accumulators
                           * live in LDS (prof_lds: [0] the previous mark, [1..8] cycles, [9..16] counts), not in registers: the build
                           * must keep the register budget of the kernel it measures */
    unsigned long long* prof_lds = nullptr;
    __device__ __forceinline__ void mark(int id)
    {
        if (threadIdx.x == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            const unsigned long long last = prof_lds[0];
            if (last) { prof_lds[1 + (id & 7)] += now - last; prof_lds[9 + (id & 7)] += 1; }
            prof_lds[0] = now;
        }
    }
    __device__ __forceinline__ void flush_marks(const MgcLattice& L)
    {
        if (L.prof && threadIdx.x == 0)
            for (int i = 0; i < 8; ++i)
                if (prof_lds[9 + i]) { atomicAdd(&L.prof[i], prof_lds[1 + i]); atomicAdd(&L.prof[i + 8], prof_lds[9 + i]); }
    }
#else
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush_marks(const MgcLattice&) {}
#endif
    __device__ __forceinline__ void fresh() { asm volatile("" : "+v"(lane)); lane &= 63; }
    /* min(a, b) of two doubles that are never NaNs: ONE v_min_f64 (fmin() costs two canonicalising v_max_f64 in front of it under IEEE mode, three
     * instructions in the innermost step of the discharge).  Call it in front of a select, not inside its arm: the compiler branches around the asm there */
    __device__ __forceinline__ double fmin_pos(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    /* v unchanged, but nothing computed from it may move above this point, nor may memory operations cross it: pins the
     * FIRST USE of a value still on its way back from memory (and with it the wait for it) to where the code says */
    __device__ __forceinline__ int use_here(int v) { asm volatile("" : "+v"(v) : : "memory"); return v; }
    /* p[l], p wave-uniform: SGPR base + zero-extended 32-bit byte offset, the addressing mode of global_load / global_store */
    template <class T>
    __device__ __forceinline__ T ld(const T* p, int l) { return *(const T*)((const char*)p + (unsigned)(l * (int)sizeof(T))); }
    template <class T>
    __device__ __forceinline__ void st(T* p, int l, T v) { *(T*)((char*)p + (unsigned)(l * (int)sizeof(T))) = v; }
    /* the same as a STREAMING store: what a visit writes back of a tile's own state (excess, residual planes, masks) is not read again before the caches
     * have turned over many times -- the next visit of the tile is two colour phases away at the earliest -- and written through them it only pushes
     * out what the visits in flight are about to read (512^3 headline 18.5 -> 18.1 ms, weak contrast 60.2 -> 59.3; labels and outboxes, which the
     * neighbours read in the very next phase, stay ordinary stores; streaming LOADS of the state on top: no gain) */
    template <class T>
    __device__ __forceinline__ void st_stream(T* p, int l, T v) { __builtin_nontemporal_store(v, (T*)((char*)p + (unsigned)(l * (int)sizeof(T)))); }
};
typedef GpuWaveT<MgcWaveShared> GpuWave;
struct alignas(16) MgcWaveSharedR { int32_t hs[1000]; }; /* what a relabel visit touches of MgcWaveShared: 4 KB, 32 waves per CU */
typedef GpuWaveT<MgcWaveSharedR> GpuWaveR;

/* Work distribution of the wave kernels: a wave takes list position blockIdx.x first, then draws further positions from
 * a ticket counter (tiles differ a lot in cost, a static stride leaves a tail).  Two ticket slots alternate from launch
 * to launch (the host flips `tk`): a launch clears the slot the NEXT launch will use, which nobody touches meanwhile,
 * so no memset, no fence and no exit counter sit between two launches. */
#define MGC_CNT_TICKET_DIS 24 /* and 25 */
#define MGC_CNT_TICKET_REL 26 /* and 27 */
__device__ __forceinline__ int mgcw_next_ticket(const MgcLattice& L, int tk)
{
    int i = 0;
    if (threadIdx.x == 0) i = atomicAdd(&L.count[tk], 1);
    return (int)gridDim.x + __builtin_amdgcn_readfirstlane(i);
}

/* workgroup 0 clears the words of counter slot c (a list the schedule is done with, or the next launch's tickets) */
__device__ __forceinline__ void mgc_clear_counter(const MgcLattice& L, int c)
{
    if (c >= 0 && blockIdx.x == 0 && (int)threadIdx.x < L.nshard) *mgc_counter(L, c, (int)threadIdx.x) = 0;
}

#ifndef MGCW_DISCHARGE_WAVES
#define MGCW_DISCHARGE_WAVES 2 /* waves per SIMD the register allocator leaves room for: 256 VGPRs each.  Rounds 2 - 5 (150 - 360 registers spilled at three):
                                  24.0 ms of discharge kernels per step at 2, 31.8 at 3, 57 at 4.  Round 6 built the kernel WITHOUT a byte of scratch at 168
                                  registers (tools/experiments/r6_three_waves_lazy_planes.patch: outflow collected in LDS, the exact in-tile labelling as an
                                  instance of its own, the inbox absorbed slot by slot, nothing kept for all eight slots) and measured it: 11.3 - 11.6 ms at three
                                  waves against 11.4 - 12.1 at two -- the memory system is saturated at ~48 visits per microsecond whatever the number of waves
                                  in flight (profiles/r6_resident_waves_curve.jsonl, r6_discharge_sections_{2,3}_waves.txt: a sweep takes 7.1 k cycles at two
                                  and 7.2 k at three, the memory phases of a visit 42 k against 96 k) -- while the leaner code is 12 % slower per visit where
                                  nothing queues (16.6 -> 18.6 us with one wave per CU): 256^3 4.4 -> 4.7 ms, 128^3 2.4 -> 2.6.  Two waves stay. */
#endif
template <int REP> /* in-plane push steps per (slot, direction) and sweep: 1, or MGCW_REPEAT_MAX (mgc_wave_ops.inl) */
__global__ __launch_bounds__(MGCW_LANES) __attribute__((amdgpu_waves_per_eu(MGCW_DISCHARGE_WAVES, MGCW_DISCHARGE_WAVES)))
void k_discharge_w(MgcLattice L, int lst, uint32_t phase, int sweeps, int flags, int tk, int zero_idx, int stagger)
{
    __shared__ MgcWaveShared S;
    GpuWave w(S);
    if (blockIdx.x == 0) { /* running totals: tiles discharged */
        MgcListView view;
        const int n = mgc_list_view(L, lst, view);
        if (threadIdx.x == 0 && n) { atomicAdd(&L.count[8], n); atomicAdd(&L.count[MGC_CNT_WAVE_TILES], n); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) L.count[tk ^ 1] = 0; /* the next launch's ticket word */
    mgc_clear_counter(L, zero_idx); /* the list the previous phase consumed */
    /* development knob: the second half of the grid (the second wave of every SIMD) starts late, so that the two waves of a
     * SIMD do not sit in their load / store phases at the same moments (units of ~8 000 shader cycles) */
    if (stagger > 0 && blockIdx.x >= gridDim.x / 2)
        for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(127);
    __shared__ int32_t pf[256];
    w.pf = pf;
#if defined(MGCW_PROFILE)
    __shared__ unsigned long long prof_lds[17];
    if (threadIdx.x < 17) prof_lds[threadIdx.x] = 0;
    w.prof_lds = prof_lds;
#endif
    w.list_begin(L, lst);
    w.tk = tk;
    int tile = __builtin_amdgcn_readfirstlane(w.entry_load((int)blockIdx.x)), st = 0; /* first visit: no ticket */
    if (tile >= 0) st = (int)L.status[tile];
    while (tile >= 0) {
        w.new_tile();
        w.mark(1); /* between two tiles */
        /* the tile id is wave-uniform: keep it (and every base address derived from it) in SGPRs */
        if (__builtin_amdgcn_readfirstlane(st) & (int)MGC_ST_SINK) mgcw_discharge_impl<true, REP>(w, L, tile, phase, sweeps, flags | ((flags & MGCW_BFS_SINK) ? MGCW_BFS : 0));
        else mgcw_discharge_impl<false, REP>(w, L, tile, phase, sweeps, flags);
#if MGCW_RUNAHEAD
        tile = w.next_tile; /* resolved inside the visit (hint_begin / hint_end in mgcw_discharge_impl) */
        st = w.nst;
#else
        w.ticket_issue(L); /* a free wave takes the next unclaimed tile: ticket -> list entry -> status word */
        w.hint_begin();
        tile = w.hint_end(L);
        st = w.nst;
#endif
    }
    if (MGCW_RUNAHEAD && MGCW_PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* (nothing may still be on its way into this wave's LDS when it ends) */
    w.flush_marks(L);
}

/* global-relabel pass over a list, one wave per tile; first = the seeding pass of a from-scratch relabel over the
 * filter's scratch list (`cnt` = its counter), else `cnt` = lst */
__global__ __launch_bounds__(MGCW_LANES) void k_relabel_w(MgcLattice L, int lst, int cnt, uint32_t epoch, int next_list, int zero_list, int first, int tk)
{
    __shared__ MgcWaveSharedR S;
    GpuWaveR w(S);
    /* the wave's first list entry is fetched together with the list length (see k_relabel_v); positions are strided, not
     * ticketed: a pass is rarely deeper than the resident waves, and a ticket is one more trip and one more hot word */
    const int spec = L.nshard == 1 && (int)blockIdx.x < L.shard_cap ? L.list[lst][blockIdx.x] : 0;
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[9], n);
    (void)tk;
    mgc_clear_counter(L, zero_list); /* consumed by the previous pass; the next pass appends to it */
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        w.new_tile();
        const int tile = (i == (int)blockIdx.x && L.nshard == 1) ? spec : mgc_list_at(L, lst, view, i);
        mgcw_relabel_tile(w, L, __builtin_amdgcn_readfirstlane(tile), epoch, next_list, first != 0);
    }
}


/* ---- 26-neighbourhood, one wave per tile (mgc_wave_ops26.inl): the wave runs ALONE on its SIMD and owns the whole 512-entry
 * register file (excess + 26 residual planes of a z-column per lane = 432 registers); four tiles in flight per CU ---- */
typedef GpuWaveT<MgcWaveShared26> GpuWave26;
__global__ __launch_bounds__(MGCW_LANES) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k26_discharge_w(MgcLattice L, int lst, uint32_t phase, int sweeps, int passes, int raises, int flags, int tk, int zero_idx)
{
    __shared__ MgcWaveShared26 S;
    GpuWave26 w(S);
    mgc_clear_counter(L, zero_idx); /* the list the previous phase consumed (nobody appends to it for the next seven phases) */
    if (blockIdx.x == 0) {
        MgcListView view;
        const int n = mgc_list_view(L, lst, view);
        if (threadIdx.x == 0 && n) atomicAdd(&L.count[MGC26_CNT_DIS], n);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) L.count[tk ^ 1] = 0; /* the next launch's ticket word */
#if defined(MGCW_PROFILE)
    __shared__ unsigned long long prof_lds[17];
    if (threadIdx.x < 17) prof_lds[threadIdx.x] = 0;
    w.prof_lds = prof_lds;
#endif
    w.list_begin(L, lst);
    w.tk = tk;
    int tile = __builtin_amdgcn_readfirstlane(w.entry_load((int)blockIdx.x)); /* first visit: no ticket */
    while (tile >= 0) {
        w.new_tile();
        w.mark(7); /* between two tiles */
        mgcw26_discharge_tile(w, L, tile, phase, sweeps, passes, raises, flags);
        w.ticket_issue(L); /* a free wave takes the next unclaimed tile: ticket -> list entry */
        w.hint_begin();
        tile = __builtin_amdgcn_readfirstlane(w.lsv);
    }
    w.flush_marks(L);
}

/* ======================================================================================
 * block executor with V voxels per thread (512 / V threads per tile): the global-relabel passes.
 * A relabel visit is short (masks + labels + halo in, a few relaxation rounds, labels out) and there are only a few
 * thousand tiles per pass: with 256 threads a CU keeps 8 tiles in flight instead of 4, every barrier joins 4 waves
 * instead of 8 and the scalar bookkeeping of a tile is paid by half as many waves.
 * ==================================================================================== */
struct alignas(16) MgcTileSharedR { /* what the relabel / activate operations touch of MgcTileShared (4 KiB instead of 40) */
    int32_t hs[1000];
    int32_t nbr[8], inflag[8], faceflag[8], depflag[8], flag[2], satflag, excflag;
};

template <int V, class SH = MgcTileSharedR, bool LAUNDER_EVERY_STEP = false, int TOTAL = MGC_TV> /* TOTAL lanes (voxels) per workgroup: a tile, or a brick of eight */
struct GpuBlockV {
    static constexpr int NT = TOTAL / V;
    static constexpr int LOG_NT = NT == 512 ? 9 : (NT == 256 ? 8 : 7);
    template <class T>
    struct Reg {
        T v[V];
        /* t = thread + k * NT with thread < NT: the slot index k = t >> LOG_NT folds to a constant (the lane id is masked
         * to its range after it was made opaque, see new_tile) */
        __device__ __forceinline__ T& operator[](int t) { return v[t >> LOG_NT]; }
    };
    SH& S;
    int tid;
    __device__ __forceinline__ explicit GpuBlockV(SH& s) : S(s), tid((int)threadIdx.x) {}
    __device__ __forceinline__ void new_tile()
    {
        tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        tid &= NT - 1;
    }
    __device__ __forceinline__ int lane() const
    {
        int t = tid;
        if (LAUNDER_EVERY_STEP) { asm volatile("" : "+v"(t)); t &= NT - 1; }
        return t;
    }
    template <class F>
    __device__ __forceinline__ void par(F f)
    {
        const int t0 = lane();
#pragma unroll
        for (int k = 0; k < V; ++k) f(t0 + k * NT);
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ bool any(F f)
    {
        const int t0 = lane();
        int r = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) r |= (int)f(t0 + k * NT);
        return __syncthreads_or(r) != 0;
    }
    __device__ __forceinline__ int atomic_add(int32_t* p, int v) { return atomicAdd(p, v); }
    __device__ __forceinline__ uint32_t atomic_exch(uint32_t* p, uint32_t v) { return atomicExch(p, v); }
    __device__ __forceinline__ void mark(const MgcLattice&, int) {}
    __device__ __forceinline__ void gadd(double* p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void gor(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ int shard(const MgcLattice& L) const { return (int)(blockIdx.x & (unsigned)(L.nshard - 1)); }
    __device__ __forceinline__ void atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
    /* a value every lane of the workgroup holds alike (read from LDS after a barrier): scalar for the branches on it */
    __device__ __forceinline__ uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ void atomic_and(uint32_t* p, uint32_t v) { atomicAnd(p, v); }
};

#ifndef MGC_RELABEL_V
#define MGC_RELABEL_V 2 /* voxels per thread of the relabel passes */
#endif
typedef GpuBlockV<MGC_RELABEL_V> GpuBlockR;

/* one kernel for the three kinds of relabel pass: over a list (`cnt` = index of its length: the list's own counter, or the
 * scratch counter of the tile filter for the seeding pass `first`), clearing `zero_list` for the pass after next */
__global__ __launch_bounds__(MGC_TV / MGC_RELABEL_V) void k_relabel_v(MgcLattice L, int lst, int cnt, uint32_t epoch, int next_list, int zero_list, int first)
{
    __shared__ MgcTileSharedR S;
    GpuBlockR x(S);
    /* the workgroup's first list entry is fetched together with the list length, not after it (one dependent trip less per
     * visit: a pass is five of them around ~3 us of relaxation); an index beyond the length reads a stale entry that is not used */
    const int spec = L.nshard == 1 && (int)blockIdx.x < L.shard_cap ? L.list[lst][blockIdx.x] : 0;
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[9], n);
    mgc_clear_counter(L, zero_list); /* consumed by the previous pass; the next pass appends to it */
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_relabel_tile(x, L, (i == (int)blockIdx.x && L.nshard == 1) ? spec : mgc_list_at(L, lst, view, i), epoch, next_list, first != 0);
        __syncthreads();
    }
}

/* global-relabel pass over a list of BRICKS (2 x 2 x 2 tiles, mgc_brick_ops.inl): 512 threads, eight voxels each */
typedef GpuBlockV<8, MgcBrickShared, false, MGC_BV> GpuBlockB;
__global__ __launch_bounds__(MGC_TV) void k_relabel_b(MgcLattice L, int lst, uint32_t epoch, int next_list, int zero_list)
{
    __shared__ MgcBrickShared S;
    GpuBlockB x(S);
    MgcListView view;
    const int n = mgc_list_view(L, lst, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[9], 4 * n); /* (in tile visits, for the schedule's cost estimate: a brick visit ~ four) */
    mgc_clear_counter(L, zero_list); /* consumed by the previous pass; the next pass appends to it */
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_relabel_brick(x, L, mgc_list_at(L, lst, view, i), epoch, next_list);
        __syncthreads();
    }
}

/* ---- 26-neighbourhood solver kernels (bodies: mgc_tile_ops26.inl) ---- */
struct MgcTileShared26V : MgcTileShared26 { /* all 26 residuals in registers: rl is never addressed (NREG = 26) */
    double rl[1][1];
};
/* region discharge with two voxels per thread and all 26 residuals of both in registers (see mgc26_discharge_tile) */
__global__ __launch_bounds__(MGC_TV / 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k26_discharge_v(MgcLattice L, int lst, uint32_t phase, int cycles, int sweeps, int zero_idx)
{
    __shared__ MgcTileShared26V S;
    GpuBlockV<2, MgcTileShared26V, true> x(S);
    mgc_clear_counter(L, zero_idx);
    MgcListView view;
    const int n = mgc_list_view(L, lst, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[MGC26_CNT_DIS], n);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc26_discharge_tile<26>(x, L, mgc_list_at(L, lst, view, i), phase, cycles, sweeps);
        __syncthreads();
    }
}

#ifndef MGC26_RELABEL_WAVES
#define MGC26_RELABEL_WAVES 8 /* 64 VGPRs, no scratch: four workgroups per CU hide the barrier per relaxation round (config 3: relabel 16.2 -> 11.6 ms; 6 waves: 13.0) */
#endif
__global__ __launch_bounds__(MGC_TV, MGC26_RELABEL_WAVES) void k26_relabel_all(MgcLattice L, uint32_t epoch, int next_list)
{
    __shared__ MgcTileShared26 S;
    GpuBlock26 x(S);
    int visited = 0;
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        x.new_tile();
        visited += (L.status[tile] >> 1) & 1u;
        mgc26_relabel_tile(x, L, tile, epoch, next_list, true);
        __syncthreads();
    }
    if (threadIdx.x == 0 && visited) atomicAdd(&L.count[MGC26_CNT_REL], visited);
}

__global__ __launch_bounds__(MGC_TV, MGC26_RELABEL_WAVES) void k26_relabel_list(MgcLattice L, int lst, uint32_t epoch, int next_list)
{
    __shared__ MgcTileShared26 S;
    GpuBlock26 x(S);
    MgcListView view;
    const int n = mgc_list_view(L, lst, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[MGC26_CNT_REL], n);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc26_relabel_tile(x, L, mgc_list_at(L, lst, view, i), epoch, next_list, false);
        __syncthreads();
    }
}

__global__ __launch_bounds__(MGC_TV) void k26_activate(MgcLattice L, uint32_t phase)
{
    __shared__ MgcTileShared26 S;
    GpuBlock26 x(S);
    int nact = 0;
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        x.new_tile();
        nact += mgc26_activate_tile(x, L, tile, phase) ? 1 : 0;
        __syncthreads();
    }
    if (threadIdx.x == 0 && nact) atomicAdd(&L.count[MGC26_CNT_ACTIVE], nact);
}

/* the same, one WAVE per tile (four tiles per workgroup): sixteen independent loads per lane and one vote instead of a barrier
 * per tile -- the pass streams 12 bytes per voxel of every tile that is not all-INF (k26_activate: 1.7 ms at 512^3, twice per solve) */
__global__ __launch_bounds__(256) void k26_activate_w(MgcLattice L, uint32_t phase)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int nact = 0;
    for (int tile = (int)blockIdx.x * 4 + wv; tile < L.ntiles; tile += (int)gridDim.x * 4) {
        if (!mgc_owned(L, tile) || (L.status[tile] & MGC_ST_ALLINF)) continue; /* (wave-uniform) */
        const double* const e = L.excess + (int64_t)tile * MGC_TV;
        const int32_t* const h = L.height + (int64_t)tile * MGC_TV;
        bool a = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) a = a || (e[k * 64 + lane] > 0.0 && h[k * 64 + lane] < MGC_HINF);
        if (__ballot(a) != 0ull) {
            if (lane == 0) {
                int tz, ty, tx;
                mgc_tile_coords(L, tile, tz, ty, tx);
                const uint32_t target = phase + (((uint32_t)mgc26_colour(L, tz, ty, tx) - phase) & 7u);
                if (atomicExch(&L.stamp[tile], target) != target) { /* (mgc_enqueue) */
                    const int pos = atomicAdd(mgc_counter(L, (int)(target & 15u), 0), 1);
                    L.list[target & 15u][pos] = tile;
                }
            }
            nact++;
        }
    }
    if (lane == 0 && nact) atomicAdd(&L.count[MGC26_CNT_ACTIVE], nact);
}

#ifndef MGC26_DISCHARGE_WAVES
#define MGC26_DISCHARGE_WAVES 4 /* waves per SIMD the register allocator leaves room for: 128 VGPRs, 2 workgroups per CU
                                   (13 of the 26 residuals live in LDS, see MgcTileShared26D) */
#endif
__global__ __launch_bounds__(MGC_TV, MGC26_DISCHARGE_WAVES) void k26_discharge(MgcLattice L, int lst, uint32_t phase, int cycles, int sweeps, int zero_idx)
{
    __shared__ MgcTileShared26D S;
    GpuBlock26D x(S);
    mgc_clear_counter(L, zero_idx); /* the list the previous phase consumed (nobody appends to it for the next seven phases) */
    MgcListView view;
    const int n = mgc_list_view(L, lst, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[MGC26_CNT_DIS], n);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        if (L.prof && threadIdx.x == 0) x.last = clock64();
        mgc26_discharge_tile(x, L, mgc_list_at(L, lst, view, i), phase, cycles, sweeps);
        __syncthreads();
    }
    x.flush_marks(L);
}

__global__ __launch_bounds__(MGC_TV) void k_absorb(MgcLattice L)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        x.new_tile();
        mgc_absorb_tile(x, L, tile);
        __syncthreads();
    }
}

__global__ __launch_bounds__(MGC_TV) void k_relabel_all(MgcLattice L, uint32_t epoch, int next_list)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    int visited = 0;
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        x.new_tile();
        visited += (L.status[tile] >> 1) & 1u;
        mgc_relabel_tile(x, L, tile, epoch, next_list, true);
        __syncthreads();
    }
    if (threadIdx.x == 0 && visited) atomicAdd(&L.count[9], visited);
}

__global__ __launch_bounds__(MGC_TV) void k_relabel_list(MgcLattice L, int lst, uint32_t epoch, int next_list, int zero_list)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    MgcListView view;
    const int n = mgc_list_view(L, lst, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[9], n);
    mgc_clear_counter(L, zero_list); /* consumed by the previous pass; the next pass appends to it */
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_relabel_tile(x, L, mgc_list_at(L, lst, view, i), epoch, next_list, false);
        __syncthreads();
    }
}

__global__ void k_zero_counts(MgcLattice L, uint32_t mask)
{
    const int c = (int)threadIdx.x / MGC_NSHARD, sh = (int)threadIdx.x % MGC_NSHARD; /* MGC_NCOUNT * MGC_NSHARD threads */
    if (!((mask >> c) & 1u)) return;
    if (sh == 0) L.count[c] = 0;
    if (sh < L.nshard) *mgc_counter(L, c, sh) = 0;
}

__global__ void k_status_or(MgcLattice L, uint32_t bits, uint32_t clear)
{
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile < L.ntiles) L.status[tile] = (L.status[tile] & ~clear) | bits;
}

/* incremental global relabel: tile-level suspect closure (one thread per tile) and reset of the suspect tiles.
 * A workgroup owns a BRICK of 8x8x8 tiles and iterates the closure inside it to a fixpoint before it leaves (the flags
 * only ever get set, so racing with the neighbour bricks is benign: what they add is picked up by the next pass).  The
 * closure then advances a brick per launch instead of a tile per launch: ~30 launches per 512^3 step instead of 200. */
template <bool FULL> /* FULL: 26 supporting neighbour tiles (mgc26_suspect_tile) */
__global__ __launch_bounds__(MGC_TV) void k_suspect_pass(MgcLattice L)
{
    const int bxn = (L.gx + 7) / 8, byn = (L.gy + 7) / 8, bzn = (L.gz + 7) / 8;
    const int t = threadIdx.x;
    bool any = false;
    for (int b = blockIdx.x; b < bxn * byn * bzn; b += gridDim.x) {
        const int bx = b % bxn, by = (b / bxn) % byn, bz = b / (bxn * byn);
        const int tz = bz * 8 + (t >> 6), ty = by * 8 + ((t >> 3) & 7), tx = bx * 8 + (t & 7);
        const bool in = tz < L.gz && ty < L.gy && tx < L.gx;
        const int tile = in ? mgc_tile_id(L, tz, ty, tx) : 0;
        for (int it = 0; it < 24; ++it) { /* a brick is at most 22 steps across */
            const bool ch = in && (FULL ? mgc26_suspect_tile(L, tile) : mgc_suspect_tile(L, tile));
            any |= ch;
            __threadfence_block();
            if (!__syncthreads_or(ch)) break;
        }
    }
    if (any) L.count[MGC_CNT_CHANGED] = 1;
}

template <bool FULL>
__global__ __launch_bounds__(MGC_TV) void k_reset_suspect(MgcLattice L, uint32_t epoch, int list, int bricks)
{
    /* a workgroup scans the status words of 512 consecutive tiles (one per lane), then resets the few that are suspect:
     * launching a 512-lane tile operation per tile just to test one flag cost 320 us per global relabel at 512^3 */
    __shared__ MgcTileShared S;
    __shared__ int sel[MGC_TV];
    __shared__ int nsel;
    GpuBlock x(S);
    for (int base = blockIdx.x * MGC_TV; base < L.ntiles; base += gridDim.x * MGC_TV) {
        if (threadIdx.x == 0) nsel = 0;
        __syncthreads();
        const int mine = base + (int)threadIdx.x;
        if (mine < L.ntiles && (L.status[mine] & MGC_ST_SUSPECT)) sel[atomicAdd(&nsel, 1)] = mine;
        __syncthreads();
        const int n = nsel;
        /* same effect as mgc_reset_suspect_tile per selected tile, without a barrier per tile: all lanes stream INF over the
         * selected tiles' labels, then one lane per tile retires the flags and queues the tile */
        for (int i = 0; i < n; ++i) {
            L.height[(int64_t)sel[i] * MGC_TV + threadIdx.x] = MGC_HINF;
            if (threadIdx.x < MGC_TF) mgc_shadow_reset(L, sel[i], (int)threadIdx.x);
        }
        if ((int)threadIdx.x < n) {
            const int tile = sel[threadIdx.x];
            L.status[tile] = (L.status[tile] & ~(MGC_ST_SUSPECT | MGC_ST_DIRTY | MGC_ST_SETTLED | (FULL ? MGC26_ST_DEP_MASK : (63u << MGC_ST_DEP_SHIFT)))) | MGC_ST_ALLINF;
            if (bricks) mgc_enqueue_brick(x, L, list, epoch, mgc_brick_of_tile(L, tile)); /* the passes of this relabel run over bricks (k_relabel_b) */
            else mgc_enqueue(x, L, list, L.rstamp, epoch, tile);
        }
        __syncthreads();
    }
}

/* Tile-level filters: ONE THREAD PER TILE decides whether the tile needs the 512-lane operation at all and, if so,
 * appends it to a scratch list (one atomic per wave).  mode 0: a neighbour left flow in its outbox for this tile;
 * mode 1: the tile holds excess (status bit); mode 2: the tile is suspect; mode 3: the tile holds an arc to the sink. */
__global__ void k_filter(MgcLattice L, int mode, int list, int cnt)
{
    for (int base = blockIdx.x * blockDim.x; base < L.ntiles; base += gridDim.x * blockDim.x) { /* uniform per block */
        const int tile = base + (int)threadIdx.x;
        bool take = false;
        if (tile < L.ntiles && mgc_owned(L, tile)) {
            if (mode == 0) {
                int tz, ty, tx;
                mgc_tile_coords(L, tile, tz, ty, tx);
                for (int f = 0; f < 6 && !take; ++f) {
                    const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
                    take = nt >= 0 && ((L.oflags[nt] >> (f ^ 1)) & 1u);
                }
            } else if (mode == 1) {
                take = (L.status[tile] & (MGC_ST_EXCESS | MGC_ST_ALLINF)) == MGC_ST_EXCESS; /* excess that may reach the sink */
            } else if (mode == 3) {
                take = (L.status[tile] & MGC_ST_SINK) != 0;
            } else {
                take = (L.status[tile] & MGC_ST_SUSPECT) != 0;
            }
        }
        const unsigned long long m = __ballot(take);
        if (m) {
            int pos = 0;
            const int sh = (int)(blockIdx.x & (unsigned)(L.nshard - 1));
            if ((threadIdx.x & 63) == 0) pos = atomicAdd(mgc_counter(L, cnt, sh), __popcll(m));
            pos = __shfl(pos, 0);
            if (take) L.list[list][(int64_t)sh * L.shard_cap + pos + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = tile;
        }
    }
}

/* first pass of a from-scratch global relabel over the tiles the filter found to hold a sink arc (the only seeds) */
__global__ __launch_bounds__(MGC_TV) void k_relabel_first_list(MgcLattice L, int list, int cnt, uint32_t epoch, int next_list)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[9], n);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_relabel_tile(x, L, mgc_list_at(L, list, view, i), epoch, next_list, true);
        __syncthreads();
    }
}

/* the same, one wave per tile (four tiles per workgroup in flight) */
__global__ __launch_bounds__(256) void k_absorb_w(MgcLattice L, int list, int cnt, int clear_cnt)
{
    __shared__ MgcWaveShared S; /* (not touched: the executor wants one) */
    GpuWave w(S);
    mgc_clear_counter(L, clear_cnt); /* the slot the next tile filter counts into (HipDevT::fslot) */
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    for (int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); i < n; i += (int)gridDim.x * 4) {
        w.new_tile();
        mgcw_absorb_tile(w, L, __builtin_amdgcn_readfirstlane(mgc_list_at(L, list, view, i)));
    }
}

__global__ __launch_bounds__(MGC_TV) void k_absorb_list(MgcLattice L, int list, int cnt, int clear_cnt)
{
    mgc_clear_counter(L, clear_cnt); /* the slot the next tile filter counts into (HipDevT::fslot) */
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_absorb_tile(x, L, mgc_list_at(L, list, view, i));
        __syncthreads();
    }
}

__global__ __launch_bounds__(MGC_TV) void k_activate_list(MgcLattice L, int list, int cnt, uint32_t phase)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_activate_tile(x, L, mgc_list_at(L, list, view, i), phase);
        __syncthreads();
    }
}

/* first global relabel as a distance transform (mgc_dt_ops.inl): one scan of every tile line along AXIS, one wave per line */
template <int AXIS, bool BWD, int SEED, int FINAL> /* SEED 1: from the sink links (rmask), 2: from the voxels that hold excess (the radial labels), 0: a later pass */
__global__ __launch_bounds__(256) void k_dt_scan(MgcLattice L, const void* in, void* out, int c_min, int32_t* hout, const uint16_t* carry_in, uint16_t* carry_out, int carry_plane)
{
    __shared__ MgcWaveShared S; /* (not touched: the executor wants one) */
    GpuWave w(S);
    const int n = mgc_dt_lines<AXIS>(L);
    for (int line = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); line < n; line += (int)gridDim.x * 4) {
        w.new_tile();
        mgc_dt_scan_line<AXIS, BWD, SEED, FINAL>(w, L, __builtin_amdgcn_readfirstlane(line), in, out, c_min, hout, carry_in, carry_out, carry_plane);
    }
}

/* Z-slabs: the shadows of the border labels after a transform that ran on both sides of every border (mgc_shadow_sync_tile) */
__global__ __launch_bounds__(256) void k_shadow_sync(MgcLattice L)
{
    __shared__ MgcWaveShared S;
    GpuWave w(S);
    const int T = L.gy * L.gx;
    for (int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); i < 2 * T; i += (int)gridDim.x * 4) {
        w.new_tile();
        mgc_shadow_sync_tile(w, L, i / T, __builtin_amdgcn_readfirstlane(i % T), L.height);
    }
}

__global__ __launch_bounds__(256) void k_dt_finish(MgcLattice L)
{
    __shared__ MgcWaveShared S;
    GpuWave w(S);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&L.count[9], L.ntiles); /* every tile was labelled once */
    for (int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); tile < L.ntiles; tile += (int)gridDim.x * 4) {
        w.new_tile();
        mgc_dt_finish_tile(w, L, __builtin_amdgcn_readfirstlane(tile));
    }
}

/* radial labels of the flood phase (mgc_dt_ops.inl): C = hops of the shortest source -> sink path; the labels lowered to
 * max(1, C - distance from the source); "does excess of the source still stand under a finite label?" -- one wave per tile */
__global__ __launch_bounds__(256) void k_dt_cmin(MgcLattice L)
{
    __shared__ MgcWaveShared S;
    GpuWave w(S);
    for (int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); tile < L.ntiles; tile += (int)gridDim.x * 4) {
        w.new_tile();
        mgc_dt_cmin_tile(w, L, __builtin_amdgcn_readfirstlane(tile));
    }
}

__global__ __launch_bounds__(256) void k_dt_lower(MgcLattice L, const uint16_t* ds, int c_min)
{
    __shared__ MgcWaveShared S;
    GpuWave w(S);
    for (int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); tile < L.ntiles; tile += (int)gridDim.x * 4) {
        w.new_tile();
        mgc_dt_lower_tile(w, L, __builtin_amdgcn_readfirstlane(tile), ds, c_min);
    }
}

__global__ __launch_bounds__(256) void k_source_open(MgcLattice L)
{
    __shared__ MgcWaveShared S;
    GpuWave w(S);
    for (int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); tile < L.ntiles; tile += (int)gridDim.x * 4) {
        w.new_tile();
        mgc_source_open_tile(w, L, __builtin_amdgcn_readfirstlane(tile));
    }
}

/* activation over the filter's list, one wave per tile (four tiles per 256-thread workgroup in flight) */
__global__ __launch_bounds__(256) void k_activate_w(MgcLattice L, int list, int cnt, uint32_t phase, int exact_max, int clear_cnt)
{
    mgc_clear_counter(L, clear_cnt); /* the slot the next activation filter counts into (HipDevT::fslot) */
    __shared__ MgcWaveShared S; /* (not touched: the executor wants one) */
    GpuWave w(S);
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    if (n <= exact_max) { /* few candidates: look at their voxels (mgcw_activate_tile) */
        for (int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); i < n; i += (int)gridDim.x * 4) {
            w.new_tile();
            mgcw_activate_tile(w, L, __builtin_amdgcn_readfirstlane(mgc_list_at(L, list, view, i)), phase, true);
        }
        return;
    }
    /* many candidates: their status words decide (the filter checked them: owned, excess under a finite label, not ALLINF).
     * One THREAD per candidate, one atomic per wave and colour -- one per tile on the two list words cost 140 us at 512^3. */
    const int sh = (int)(blockIdx.x & (unsigned)(L.nshard - 1)), lane = (int)(threadIdx.x & 63);
    for (int base = (int)blockIdx.x * 256; base < n; base += (int)gridDim.x * 256) {
        const int i = base + (int)threadIdx.x;
        const bool take = i < n;
        const int tile = take ? mgc_list_at(L, list, view, i) : 0;
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const uint32_t target = phase + ((mgc_tile_colour(L, tz, ty, tx) ^ (int)(phase & 1u)) & 1);
        for (uint32_t tg = phase; tg <= phase + 1; ++tg) {
            const unsigned long long m = __ballot(take && target == tg);
            if (!m) continue;
            int pos = 0;
            if (lane == 0) pos = atomicAdd(mgc_counter(L, (int)(tg & 3u), sh), __popcll(m));
            pos = __shfl(pos, 0);
            if (take && target == tg) {
                L.list[tg & 3u][(int64_t)sh * L.shard_cap + pos + __popcll(m & ((1ull << lane) - 1ull))] = tile;
                L.stamp[tile] = tg; /* (each candidate is listed once: no exchange needed to keep it unique) */
            }
        }
        const unsigned long long all = __ballot(take);
        if (all && lane == 0) atomicAdd(&L.count[6], __popcll(all));
    }
}

__global__ __launch_bounds__(MGC_TV) void k_reset_suspect_list(MgcLattice L, int list, int cnt, uint32_t epoch, int out_list)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        mgc_reset_suspect_tile(x, L, mgc_list_at(L, list, view, i), epoch, out_list);
        __syncthreads();
    }
}

__global__ __launch_bounds__(MGC_TV) void k_activate(MgcLattice L, uint32_t phase)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        x.new_tile();
        mgc_activate_tile(x, L, tile, phase);
        __syncthreads();
    }
}

#ifndef MGC_DISCHARGE_WAVES
#define MGC_DISCHARGE_WAVES 8 /* waves per SIMD the register allocator must leave room for: 4 workgroups per CU
                                 (measured on MI355X: 151 ms vs 179 ms at 512^3 despite the spills) */
#endif
__global__ __launch_bounds__(MGC_TV, MGC_DISCHARGE_WAVES) void k_discharge(MgcLattice L, int lst, uint32_t phase, int cycles, int sweeps, int zero_idx)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    MgcListView view;
    const int n = mgc_list_view(L, lst, view);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&L.count[8], n);
    mgc_clear_counter(L, zero_idx); /* the list the previous phase consumed */
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        x.new_tile();
        if (L.prof && threadIdx.x == 0) x.last = clock64();
        mgc_discharge_tile(x, L, mgc_list_at(L, lst, view, i), phase, cycles, sweeps);
        __syncthreads();
    }
    x.flush_marks(L);
}

__global__ __launch_bounds__(MGC_TV) void k_halo_pack(MgcLattice L, int side, int kind, void* buf)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    const int T = L.gy * L.gx;
    for (int i = blockIdx.x; i < T; i += gridDim.x) mgc_halo_pack_nd(x, L, side, kind, i, buf);
}

__global__ __launch_bounds__(MGC_TV) void k_halo_unpack(MgcLattice L, int side, int kind, const void* buf, uint32_t epoch, int list)
{
    __shared__ MgcTileShared S;
    GpuBlock x(S);
    const int T = L.gy * L.gx;
    for (int i = blockIdx.x; i < T; i += gridDim.x) mgc_halo_unpack_nd(x, L, side, kind, i, buf, epoch, list);
}

/* ======================================================================================
 * graph construction
 * ==================================================================================== */
struct MgcBuildArgs {
    const void* image;   /* C-order, handle shape, any mgc_dtype; NULL when term == NONE */
    int img_dtype;
    int term;
    double p0;           /* linear: M ; exponential: sigma^2 ; division / power: sigma */
    double inv_axis[3];  /* divisor per axis x,y,z (spacing) -- only used when has_spacing */
    double div26[26];    /* 26-neighbourhood: Euclidean length of (offset * spacing) per direction */
    int has_spacing;
    const void* prob;    /* regional probability map or NULL */
    int prob_dtype;
    double alpha;
    const uint8_t* fg;   /* marker masks or NULL */
    const uint8_t* bg;
    const double* tr_in; /* merged explicit t-links (plug-in path) or NULL */
    double* tr0;         /* out: merged tr_cap per voxel, tile-major */
    double* fpart;       /* out: per-tile partial of the flow constant */
    uint8_t* tflags;     /* out: per tile, bit 0: some voxel has a source link (tr0 > 0), bit 1: a sink link (tr0 < 0) -- as built */
    const double* lut;   /* mgc_set_boundary_lut: the term by table for integer-valued images, or NULL */
    int lut_n;
    int prepush;         /* 26-neighbourhood: settle source -> u -> v -> sink paths inside a tile while its weights are in registers (k_build_prepush26) */
};

/* Graph::add_tweights, graph.h:416-425 */
__device__ __forceinline__ void mgc_add_tweights(double& tr, double& fconst, double cs, double ck)
{
    const double delta = tr;
    if (delta > 0) cs += delta;
    else           ck -= delta;
    fconst += (cs < ck) ? cs : ck;
    tr = cs - ck;
}

/* deterministic block sum of one double per lane (fixed tree), result valid in lane 0 */
__device__ __forceinline__ double mgc_block_sum(double v, double* scratch)
{
    const int t = threadIdx.x;
    scratch[t] = v;
    __syncthreads();
    for (int s = MGC_TV / 2; s > 0; s >>= 1) {
        if (t < s) scratch[t] += scratch[t + s];
        __syncthreads();
    }
    return scratch[0];
}

/* A kernel's own arguments, read AGAIN where they are used.  The twenty-odd pointers of k_build's two argument structs are loop invariants: the
 * optimiser loads them once in front of the tile loop and keeps them -- more scalar registers than the wave has, so they went to and fro between
 * SGPRs and lanes of a VGPR around every use (65 - 70 spilled registers, ~150 v_readlane / v_writelane per tile).  A pointer to the kernarg segment
 * that went through an empty asm statement is opaque: what is loaded through it is loaded HERE (s_load from the scalar cache, a few per tile) and
 * lives only as long as it is used.  OFF = byte offset of the argument in the segment. */
#define MGC_KARG __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const MGC_KARG T* mgc_kernarg_again(int off)
{
    const MGC_KARG char* p = (const MGC_KARG char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const MGC_KARG T*)(p + off);
}
#define MGC_BUILD_ARGS_OFFSET ((int)((sizeof(MgcLattice) + 7) & ~(size_t)7)) /* k_build(MgcLattice, MgcBuildArgs) */

/* TERM: the boundary term as a compile-time constant (the kernel dispatches once), so g(.) is straight-line code */
/* TABLE false: an instance without the term-by-table of integer-valued images (mgc_set_boundary_lut).  Not a micro-saving: a table load's wait sits
 * where the table path and the arithmetic path of g(.) merge, so every tile runs into it, table or not -- and on this hardware it is a wait for
 * ALL of the wave's memory operations in flight, the previous tile's stores and the next tile's image block included. */
template <bool FULL, int TERM, bool PRE6 = false, bool TABLE = true> /* FULL: 26-neighbourhood; PRE6: the 6-neighbourhood instance with the pre-push (graphs with a regional term) */
__device__ __forceinline__ void k_build_tiles(const MgcLattice& L, const MgcBuildArgs& A, double* img, double* scratch, double* wf, int* tflag_lds, double* pre_lds)
{
    const int t_lane = threadIdx.x;
    const bool take_abs = (TERM == MGC_TERM_MAXIMUM_LINEAR || TERM == MGC_TERM_MAXIMUM_EXPONENTIAL || TERM == MGC_TERM_MAXIMUM_POWER);
    /* XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Workgroup b therefore
     * works inside the b % 8-th eighth of the tile range, consecutive workgroups of one XCD on consecutive tiles, so the
     * image rows two x-neighbour tiles share (and the halo planes of y / z neighbours) are fetched by ONE L2 instead of
     * by up to eight.  (gridDim.x is a multiple of 8 or smaller than 8.) */
    const int nx = gridDim.x >= 8 ? 8 : 1;
    const int chunk = (L.ntiles + nx - 1) / nx, stride = (int)gridDim.x / nx;
    int sink_tiles = 0; /* (thread 0) tiles of this workgroup that hold a sink link */
    int wall_tiles = 0; /* (thread 0) ... that a surface of weak arcs passes through (MGC_WALL_*) */
    /* float32 images (the usual case): the 10 x 10 x 10 block of a tile is asked for ONE TILE AHEAD -- right behind the first barrier of the tile
     * before -- and put into LDS behind that tile's stores.  A wave's vector memory operations retire in issue order: a tile that begins by
     * waiting for its own image loads waits for every store of the tile before to be acknowledged (diagnostic builds, round 5: the kernel
     * without its image loads 2.4 ms instead of 3.15).  Loaded this way the block has the whole evaluation of g(.) to arrive in. */
    const bool ahead = TERM != MGC_TERM_NONE && A.img_dtype == MGC_F32; /* (uniform; every path is past its reads of the block at the end of its tile: barriers behind the weights) */
    float raw_a = 0.f, raw_b = 0.f;
    auto fetch_f32 = [&](int tile_n, int tl) __attribute__((always_inline)) {
        int nz, ny, nxx;
        mgc_tile_coords(L, tile_n, nz, ny, nxx);
        const int64_t bz = (int64_t)nz * 8, by = (int64_t)ny * 8, bx = (int64_t)nxx * 8;
        const float* const im = (const float*)mgc_kernarg_again<MgcBuildArgs>(MGC_BUILD_ARGS_OFFSET)->image;
        raw_a = raw_b = 0.f;
        {
            const int k = tl;
            const int64_t gz = bz + k / 100 - 1, gy = by + (k / 10) % 10 - 1, gx = bx + k % 10 - 1;
            if (gz >= 0 && gz < L.dz && gy >= 0 && gy < L.dy && gx >= 0 && gx < L.dx) raw_a = im[(gz * L.dy + gy) * L.dx + gx];
        }
        if (tl + MGC_TV < 1000) {
            const int k = tl + MGC_TV;
            const int64_t gz = bz + k / 100 - 1, gy = by + (k / 10) % 10 - 1, gx = bx + k % 10 - 1;
            if (gz >= 0 && gz < L.dz && gy >= 0 && gy < L.dy && gx >= 0 && gx < L.dx) raw_b = im[(gz * L.dy + gy) * L.dx + gx];
        }
    };
    auto stage_f32 = [&](int tl) __attribute__((always_inline)) {
        img[tl] = take_abs ? fabs((double)raw_a) : (double)raw_a;
        if (tl + MGC_TV < 1000) img[tl + MGC_TV] = take_abs ? fabs((double)raw_b) : (double)raw_b;
    };
    if (ahead) { /* the first tile's block */
        const int first = ((int)blockIdx.x % nx) * chunk + (int)blockIdx.x / nx;
        if ((int)blockIdx.x / nx < chunk && first < L.ntiles) {
            fetch_f32(first, t_lane);
            stage_f32(t_lane);
        }
    }
    for (int idx = (int)blockIdx.x / nx; idx < chunk; idx += stride) {
        const int tile = ((int)blockIdx.x % nx) * chunk + idx;
        if (tile >= L.ntiles) break;
        /* the lane id is made opaque once per tile: everything derived from it (LDS slots of the weights, halo indices, the
         * lane's coordinates) is then recomputed per tile -- a dozen integer operations -- instead of being hoisted out of the
         * tile loop, where those values lived across the whole body and were spilled under the kernel's register cap (round 4:
         * 64 B of scratch per lane for the exponential term, 140 - 144 B for the power terms, reloaded nine times per tile) */
        int t = t_lane;
        asm volatile("" : "+v"(t));
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const int64_t z0 = (int64_t)tz * 8, y0 = (int64_t)ty * 8, x0 = (int64_t)tx * 8;
        /* the marker bytes of this lane's voxel are asked for HERE, together with the image tile: they are needed behind the weights,
         * and fetched there they were a second full memory latency per tile (two dependent trips to HBM where one does) */
        uint32_t pre_fg = 0, pre_bg = 0; /* (whole registers: see the use in tlinks()) */
        {
            const int plz = t >> 6, ply = (t >> 3) & 7, plx = t & 7;
            const int64_t pz = z0 + plz, py = y0 + ply, px = x0 + plx;
            if (pz < L.dz && py < L.dy && px < L.dx) {
                const int64_t pid = (pz * L.dy + py) * L.dx + px;
                const MGC_KARG MgcBuildArgs* const Am = mgc_kernarg_again<MgcBuildArgs>(MGC_BUILD_ARGS_OFFSET);
                if (Am->fg) pre_fg = Am->fg[pid];
                if (Am->bg) pre_bg = Am->bg[pid];
            }
        }
        if (TERM != MGC_TERM_NONE && !ahead) {
            for (int k = t; k < 1000; k += MGC_TV) {
                const int64_t gz = z0 + k / 100 - 1, gy = y0 + (k / 10) % 10 - 1, gx = x0 + k % 10 - 1;
                double v = 0.0;
                if (gz >= 0 && gz < L.dz && gy >= 0 && gy < L.dy && gx >= 0 && gx < L.dx)
                    v = mgc_load_as_double(A.image, A.img_dtype, (gz * L.dy + gy) * L.dx + gx, take_abs);
                img[k] = v;
            }
        }
        __syncthreads();
        const int next = tile + stride;
        const bool has_next = ahead && idx + stride < chunk && next < L.ntiles; /* (uniform) */
        if constexpr (!FULL && !PRE6) {
            if (has_next) fetch_f32(next, t); /* in flight until the end of this tile */
        }
        if (t == 0) *tflag_lds = 0; /* everybody is past the previous tile's look at it; this tile's votes come after the next barrier */
        const int lz = t >> 6, ly = (t >> 3) & 7, lx = t & 7;
        const int64_t gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
        const bool valid = gz < L.dz && gy < L.dy && gx < L.dx;
        const int64_t id = (gz * L.dy + gy) * L.dx + gx;
        const int me = mgc_hs_index(lz, ly, lx);
        /* t-links: regional term, then fg marker, then bg marker (generate.py:159-172); then which signs of t-link the tile holds and
         * whether a flow constant has to be summed: ONE barrier (the waves vote, one lane per wave ORs the result into an LDS
         * word) where three barrier-reductions stood.  The 26-neighbourhood path runs this BEFORE its weights (the pre-push needs the
         * t-links), the 6-neighbourhood path behind them (its weight hand-over barrier separates the reset of the vote word from the votes). */
        double tr = 0.0, fc = 0.0;
        int tbits = 0, weak_voxels = 0;
        bool weak_voxel = false; /* one of this voxel's three FORWARD n-links is weak (MGC_WALL_WEIGHT; 6-neighbourhood, markers only: set before the vote) */
        /* Two halves.  tlinks(): everything that WAITS for memory (the marker bytes asked for at the top of the tile, the explicit t-links,
         * the probability map) and the merge; vote(): ballots and the barrier.  The 6-neighbourhood path runs tlinks() in FRONT of its
         * plane stores: a wave's vector memory operations retire in issue order, so a wait for a marker byte placed behind the six plane
         * stores -- where the whole lambda stood until round 5 -- was a wait for those stores to be acknowledged, once per tile
         * (diagnostic builds: the kernel without its image loads, whose wait has the same effect on the tile before, ran 2.4 ms
         * instead of 3.15). */
        auto tlinks = [&]() __attribute__((always_inline)) {
            /* the bytes are looked at HERE and not where they were loaded: left alone the compiler turns "byte != 0" into a lane mask right
             * behind the load, and the wait that goes with it -- at the top of the tile -- is a wait for the stores of the tile before */
            asm volatile("" : "+v"(pre_fg), "+v"(pre_bg));
            const MGC_KARG MgcBuildArgs* const At = mgc_kernarg_again<MgcBuildArgs>(MGC_BUILD_ARGS_OFFSET);
            if (valid) {
                if (At->tr_in) tr = At->tr_in[id];
                if (At->prob) {
                    double cs, ck;
                    if (At->prob_dtype == MGC_F32) {
                        const float p = ((const float*)At->prob)[id], al = (float)At->alpha;
                        cs = (double)(p * al);
                        ck = (double)((1.0f - p) * al);
                    } else {
                        const double p = ((const double*)At->prob)[id];
                        cs = p * At->alpha;
                        ck = (1.0 - p) * At->alpha;
                    }
                    mgc_add_tweights(tr, fc, cs, ck);
                }
                if (pre_fg) mgc_add_tweights(tr, fc, MGC_MARKER_MAX, 0.0);
                if (pre_bg) mgc_add_tweights(tr, fc, 0.0, MGC_MARKER_MAX);
            }
        };
        auto vote = [&]() __attribute__((always_inline)) {
            const int bits = (__ballot(tr < 0.0) != 0ull ? 2 : 0) | (__ballot(tr > 0.0) != 0ull ? 1 : 0) | (__ballot(fc != 0.0) != 0ull ? 4 : 0);
            /* bits 8..: how many voxels of the tile hold a weak n-link (the same word, one atomic per wave: the count never carries into the vote bits) */
            const int weak_here = __popcll(__ballot(weak_voxel));
            if ((t & 63) == 0 && weak_here) atomicAdd(tflag_lds, weak_here << 8);
            if ((t & 63) == 0 && bits) atomicOr(tflag_lds, bits);
            __syncthreads();
            tbits = *tflag_lds;
            weak_voxels = tbits >> 8;
            tbits &= 7;
        };
        if constexpr (FULL || PRE6) {
            __syncthreads(); /* the reset of the vote word above, before the votes */
            tlinks();
            vote();
            if (has_next) fetch_f32(next, t); /* (behind the wait for the marker bytes: in front of it the block's whole latency would be waited for with them) */
        }
        const bool pre = (FULL || PRE6) && A.prepush && (tbits & 3) == 3; /* (uniform) the tile holds source links AND sink links */
        double exc_out = tr > 0.0 ? tr : 0.0, snk_out = tr < 0.0 ? -tr : 0.0; /* (the 6-neighbourhood path: after its tlinks_and_vote below) */
        uint32_t m = 0;
        if constexpr (!FULL) {
            /* Every n-link weight is evaluated ONCE, by the lower voxel of its pair (g(lower, upper), the operand order of the
             * reference's slices, energy_voxel.py:650-664), and handed to the upper voxel through LDS: wf[axis][c][u][v] =
             * weight of the pair (c - 1, c) along `axis`.  512 lanes evaluate their three forward pairs, 192 of them also the
             * pair that enters the tile through a lower face: 1728 evaluations of g per tile instead of 3072. */
            if (TERM != MGC_TERM_NONE) {
                auto pair_weight = [&](int axis, int lo, bool ok, bool own = true) -> double { /* lo = img[] index of the lower voxel; ok = both voxels exist */
                    if (!ok) return 0.0;
                    double w = mgc_boundary_g(TERM, img[lo], img[lo + (axis == 0 ? 1 : (axis == 1 ? 10 : 100))], A.p0, TABLE ? A.lut : nullptr, TABLE ? A.lut_n : 0);
                    /* a WEAK pair (MGC_WALL_*): judged on g(.) itself, before the division by the spacing -- all eight terms take values in
                     * (0, 1], so "below 2^-30" is relative to the largest weight the term can produce whatever the voxel spacing (round 5
                     * compared the divided weight: a volume with a spacing of 1e-3 or 1e3 moved the rule by that factor).  Every pair is looked at
                     * once, by its lower voxel -- the lanes that evaluate the pair entering through a lower face leave it to the tile next door */
                    if (own) weak_voxel = weak_voxel || (w > 0.0 && w < MGC_WALL_WEIGHT);
                    if (A.has_spacing) w = w / A.inv_axis[axis]; /* energy_voxel.py:657-658 */
                    return w;
                };
                /* one g(.) at a time: left to itself the compiler interleaves the three evaluations (three exp / pow bodies in
                 * flight) and spills under the kernel's register cap -- 64 B of scratch per lane for the exponential term,
                 * 144 B for the power terms (round 4's resource remarks); the issue slots the interleaving would fill belong
                 * to the other five waves of the SIMD anyway */
                wf[0 * 576 + (lx + 1) * 64 + lz * 8 + ly] = pair_weight(0, me, valid && gx + 1 < L.dx);
                asm volatile("" ::: "memory");
                wf[1 * 576 + (ly + 1) * 64 + lz * 8 + lx] = pair_weight(1, me, valid && gy + 1 < L.dy);
                asm volatile("" ::: "memory");
                wf[2 * 576 + (lz + 1) * 64 + ly * 8 + lx] = pair_weight(2, me, valid && gz + 1 < L.dz);
                asm volatile("" ::: "memory");
                if (t < 192) { /* the pair that enters the tile through its lower face along `axis`: (halo voxel, voxel 0) */
                    const int axis = t >> 6, u = (t >> 3) & 7, v = t & 7;
                    const int lo = axis == 0 ? mgc_hs_index(u, v, -1) : (axis == 1 ? mgc_hs_index(u, -1, v) : mgc_hs_index(-1, u, v));
                    const bool ok = axis == 0 ? (x0 > 0 && z0 + u < L.dz && y0 + v < L.dy)
                                              : (axis == 1 ? (y0 > 0 && z0 + u < L.dz && x0 + v < L.dx) : (z0 > 0 && y0 + u < L.dy && x0 + v < L.dx));
                    wf[axis * 576 + u * 8 + v] = pair_weight(axis, lo, ok, false);
                }
            }
            __syncthreads();
            if constexpr (PRE6) {
                /* the pre-push of the 26-neighbourhood path below, three pairs of opposite directions: voxels that hold excess push
                 * min(excess, weight, what the neighbour's sink link still takes) to their neighbours inside the tile, the receiver hands
                 * it on to the sink -- the first colour round of a graph with a regional term, settled while the weights are in LDS */
                double w6[6];
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                    w6[d] = 0.0;
                    if (TERM != MGC_TERM_NONE && valid) {
                        const int c = ((d >> 1) == 0 ? lx : ((d >> 1) == 1 ? ly : lz)) + (d & 1);
                        const int uv = (d >> 1) == 0 ? lz * 8 + ly : ((d >> 1) == 1 ? lz * 8 + lx : ly * 8 + lx);
                        w6[d] = wf[(d >> 1) * 576 + c * 64 + uv];
                    }
                    if (L.cap0) L.cap0[((int64_t)tile * 6 + d) * MGC_TV + t] = w6[d]; /* (as built) */
                }
                if (pre) {
                    double* const dfc = pre_lds; /* [512] what the voxels' sink links still take */
                    double e = tr > 0.0 ? tr : 0.0;
                    dfc[t] = tr < 0.0 ? -tr : 0.0;
                    __syncthreads();
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        double* const revA = pre_lds + MGC_TV * (1 + 2 * (ax & 1)); /* what arrived along the - direction (double-buffered over the pairs) */
                        double* const revB = revA + MGC_TV;                           /* ... along the + direction */
                        const int cl = ax == 0 ? lx : (ax == 1 ? ly : lz), st = ax == 0 ? 1 : (ax == 1 ? 8 : 64);
                        const bool inA = cl > 0, inB = cl < 7; /* my neighbour along - / + lies in the tile */
                        double pA = 0.0, pB = 0.0;
                        if (inA) {
                            if (e > 0.0 && w6[2 * ax] > 0.0) {
                                const double q = dfc[t - st];
                                pA = fmin(e, fmin(w6[2 * ax], q));
                                if (pA > 0.0) { dfc[t - st] = q - pA; e -= pA; }
                            }
                            revA[t - st] = pA;
                        }
                        __syncthreads();
                        if (inB) {
                            if (e > 0.0 && w6[2 * ax + 1] > 0.0) {
                                const double q = dfc[t + st];
                                pB = fmin(e, fmin(w6[2 * ax + 1], q));
                                if (pB > 0.0) { dfc[t + st] = q - pB; e -= pB; }
                            }
                            revB[t + st] = pB;
                        }
                        __syncthreads();
                        w6[2 * ax] = (w6[2 * ax] - pA) + (inA ? revB[t] : 0.0); /* my - neighbour pushed back to me along + */
                        w6[2 * ax + 1] = (w6[2 * ax + 1] - pB) + (inB ? revA[t] : 0.0);
                    }
                    __syncthreads();
                    exc_out = e;
                    snk_out = dfc[t];
                    __syncthreads(); /* (dfc is rewritten by the next tile) */
                }
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                    L.rcap[((int64_t)tile * 6 + d) * MGC_TV + t] = w6[d];
                    if (w6[d] > 0.0) m |= 1u << d;
                }
            } else {
            tlinks(); /* (in front of the stores: see above) */
            const MGC_KARG MgcLattice* const Lw = mgc_kernarg_again<MgcLattice>(0);
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                double w = 0.0;
                if (TERM != MGC_TERM_NONE && valid) {
                    const int c = ((d >> 1) == 0 ? lx : ((d >> 1) == 1 ? ly : lz)) + (d & 1);
                    const int uv = (d >> 1) == 0 ? lz * 8 + ly : ((d >> 1) == 1 ? lz * 8 + lx : ly * 8 + lx);
                    w = wf[(d >> 1) * 576 + c * 64 + uv];
                }
                /* (16-byte stores -- two x-neighbours per lane, fetched from wf -- were measured on MI355X: 4.9 ms instead of
                 * 4.4 ms for this kernel at 512^3; the kernel is bound by instruction issue, not by the store width) */
                const int64_t o = ((int64_t)tile * 6 + d) * MGC_TV + t;
                __builtin_nontemporal_store(w, &Lw->rcap[o]); /* 6.4 GB at 512^3 that nothing reads before the caches have turned over many times (2.78 -> 2.70 ms) */
                if (Lw->cap0) Lw->cap0[o] = w;
                if (w > 0.0) m |= 1u << d; /* NaN (0/0 of the linear terms on a constant image) is not residual */
            }
            }
        } else {
            /* full neighbourhood: same g(.) on all 26 offsets (oracle/energy_numpy.py:boundary_weights_offsets).  Unrolled: the
             * offsets, the LDS steps and the plane offsets of the stores are constants, and "is there a neighbour" is three bits
             * of a per-lane mask instead of six 64-bit comparisons per direction (this kernel is bound by instruction issue). */
            const uint32_t nb = (gz > 0 ? 1u : 0u) | (gz + 1 < L.dz ? 2u : 0u) | (gy > 0 ? 4u : 0u) | (gy + 1 < L.dy ? 8u : 0u) |
                                (gx > 0 ? 16u : 0u) | (gx + 1 < L.dx ? 32u : 0u);
            double* const plane0 = L.rcap + ((int64_t)tile * MGC26_NDIR) * MGC_TV + t;
            const double mine = TERM != MGC_TERM_NONE ? img[me] : 0.0;
            auto weight = [&](auto dc) __attribute__((always_inline)) -> double {
                constexpr int d = decltype(dc)::value;
                constexpr int c = d < 13 ? d : d + 1;
                constexpr int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
                constexpr uint32_t need = (dz < 0 ? 1u : (dz > 0 ? 2u : 0u)) | (dy < 0 ? 4u : (dy > 0 ? 8u : 0u)) | (dx < 0 ? 16u : (dx > 0 ? 32u : 0u));
                const bool has = valid && (nb & need) == need;
                double w = 0.0;
                if (has && TERM != MGC_TERM_NONE) {
                    const double other = img[me + dz * 100 + dy * 10 + dx];
                    const double* const lut = TABLE ? A.lut : nullptr;
                    const int lut_n = TABLE ? A.lut_n : 0;
                    w = d >= 13 ? mgc_boundary_g(TERM, mine, other, A.p0, lut, lut_n) : mgc_boundary_g(TERM, other, mine, A.p0, lut, lut_n); /* g(lower, upper) */
                    if (A.has_spacing) w = w / A.div26[d];
                }
                if (L.cap0) L.cap0[((int64_t)tile * MGC26_NDIR + d) * MGC_TV + t] = w; /* (as built) */
                return w;
            };
            /* the same for ANOTHER voxel (vz, vy, vx) of the tile (run-time coordinates; see `produce` in the pre-push) */
            auto weight_at = [&](auto dc, int vz, int vy, int vx) __attribute__((always_inline)) -> double {
                constexpr int d = decltype(dc)::value;
                constexpr int c = d < 13 ? d : d + 1;
                constexpr int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
                constexpr uint32_t need = (dz < 0 ? 1u : (dz > 0 ? 2u : 0u)) | (dy < 0 ? 4u : (dy > 0 ? 8u : 0u)) | (dx < 0 ? 16u : (dx > 0 ? 32u : 0u));
                const int64_t qz = z0 + vz, qy = y0 + vy, qx = x0 + vx;
                const uint32_t nbq = (qz > 0 ? 1u : 0u) | (qz + 1 < L.dz ? 2u : 0u) | (qy > 0 ? 4u : 0u) | (qy + 1 < L.dy ? 8u : 0u) |
                                     (qx > 0 ? 16u : 0u) | (qx + 1 < L.dx ? 32u : 0u);
                const bool has = qz < L.dz && qy < L.dy && qx < L.dx && (nbq & need) == need;
                double w = 0.0;
                if (has && TERM != MGC_TERM_NONE) {
                    const int mq = mgc_hs_index(vz, vy, vx);
                    const double a = img[mq], b = img[mq + dz * 100 + dy * 10 + dx];
                    const double* const lut = TABLE ? A.lut : nullptr;
                    const int lut_n = TABLE ? A.lut_n : 0;
                    w = d >= 13 ? mgc_boundary_g(TERM, a, b, A.p0, lut, lut_n) : mgc_boundary_g(TERM, b, a, A.p0, lut, lut_n); /* g(lower, upper) */
                    if (A.has_spacing) w = w / A.div26[d];
                }
                return w;
            };
            if (pre) {
                /* ---- PRE-PUSH (no reference counterpart; part of the solver, not of the energy terms): with a regional term every
                 * tile holds voxels with a source link next to voxels with a sink link, and the first colour round of the solve
                 * would visit EVERY tile (120 KB in, up to 120 KB out per visit) to move flow one voxel over.  Here, while the
                 * tile's weights are in registers anyway, every voxel that holds excess pushes min(excess, weight, what the
                 * neighbour's sink link still takes) to its neighbours INSIDE the tile, direction by direction (one writer per
                 * target voxel and step: fixed order of f64 operations), and the receiver hands it straight on to the sink.  The result is a valid
                 * preflow (flow on u -> v -> sink paths only; no voxel gains excess); what it leaves is what the solve starts
                 * from.  Directions go in opposite pairs (d, 25 - d): the residual of d at a voxel changes by what it pushed
                 * along d and by what its neighbour pushed back along 25 - d, so both planes of a pair are stored together. ---- */
                double* const dfc = pre_lds;                 /* [512] what the voxels' sink links still take */
                double e = tr > 0.0 ? tr : 0.0;
                dfc[t] = tr < 0.0 ? -tr : 0.0;
                /* Every arc pair is evaluated ONCE, by its lower end: the weight of (v, dA) -- dA points to the LOWER neighbour -- is the weight of
                 * (v + o(dA), dB = 25 - dA), which that neighbour computes anyway.  Until round 6 every voxel evaluated all 26 (twice the exp / pow of
                 * the 13 pairs: 5.4 ms of instruction issue at 512^3).  produce(i): every voxel evaluates ITS dB-weight of pair i and leaves it in the
                 * exchange buffer at the slot of the voxel above (the one whose dA-weight it is); the voxels whose lower neighbour lies in ANOTHER tile
                 * (one, two or three boundary layers of the tile: 64 ... 169 of 512) get theirs evaluated directly, by the first so many threads of the
                 * workgroup -- whole waves, not the boundary lanes of every wave.  Pair i + 1 is produced while pair i is pushed (the two barriers of a
                 * pair are the ones the hand-off needs: none is added), into the buffer of the other parity. */
                double* const wx = pre_lds + 5 * MGC_TV; /* [2][512] */
                double wB_carry = 0.0;
                auto produce = [&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int dA = i, dB = 25 - i;
                    constexpr int dz = dA / 9 - 1, dy = (dA / 3) % 3 - 1, dx = dA % 3 - 1;
                    double* const buf = wx + (i & 1) * MGC_TV;
                    wB_carry = weight(std::integral_constant<int, dB>{});
                    { /* my dB-weight is the dA-weight of the voxel at t - o(dA) */
                        const int zb = lz - dz, yb = ly - dy, xb = lx - dx;
                        if (zb >= 0 && zb < 8 && yb >= 0 && yb < 8 && xb >= 0 && xb < 8) buf[t - (dz * 64 + dy * 8 + dx)] = wB_carry;
                    }
                    /* the voxels v with v + o(dA) outside the tile: layer z == zf (dz != 0), then y == yf of the other layers (dy != 0), then x == xf of what is left */
                    constexpr int zf = dz < 0 ? 0 : 7, yf = dy < 0 ? 0 : 7, xf = dx < 0 ? 0 : 7;
                    constexpr int zs = dz != 0 ? 7 : 8, ys = dy != 0 ? 7 : 8;
                    constexpr int nZ = dz != 0 ? 64 : 0, nY = dy != 0 ? zs * 8 : 0, nX = dx != 0 ? zs * ys : 0;
                    if (t < nZ + nY + nX) {
                        int vz, vy, vx;
                        if (t < nZ) { vz = zf; vy = t >> 3; vx = t & 7; }
                        else if (t < nZ + nY) {
                            const int k = t - nZ, zi = k >> 3;
                            vz = zi + ((dz != 0 && zi >= zf) ? 1 : 0); vy = yf; vx = k & 7;
                        } else {
                            const int k = t - nZ - nY, zi = k / ys, yi = k % ys;
                            vz = zi + ((dz != 0 && zi >= zf) ? 1 : 0); vy = yi + ((dy != 0 && yi >= yf) ? 1 : 0); vx = xf;
                        }
                        buf[vz * 64 + vy * 8 + vx] = weight_at(std::integral_constant<int, dA>{}, vz, vy, vx);
                    }
                };
                produce(std::integral_constant<int, 0>{});
                __syncthreads();
                mgcw_static_for<13>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int dA = i, dB = 25 - i;
                    constexpr int c = dA; /* dA < 13 */
                    constexpr int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
                    double* const revA = pre_lds + MGC_TV * (1 + 2 * (i & 1)); /* [512] what arrived along dA (double-buffered over the pairs) */
                    double* const revB = revA + MGC_TV;                          /* ... along dB */
                    double wA = wx[(i & 1) * MGC_TV + t], wB = wB_carry;
                    if (L.cap0) L.cap0[((int64_t)tile * MGC26_NDIR + dA) * MGC_TV + t] = wA; /* (as built) */
                    if constexpr (i < 12) produce(std::integral_constant<int, i + 1>{}); /* (read at the top of the next pair, two barriers from here) */
                    const int za = lz + dz, ya = ly + dy, xa = lx + dx, zb = lz - dz, yb = ly - dy, xb = lx - dx;
                    const bool inA = za >= 0 && za < 8 && ya >= 0 && ya < 8 && xa >= 0 && xa < 8; /* my neighbour along dA lies in the tile */
                    const bool inB = zb >= 0 && zb < 8 && yb >= 0 && yb < 8 && xb >= 0 && xb < 8;
                    const int vA = t + dz * 64 + dy * 8 + dx, vB = t - (dz * 64 + dy * 8 + dx);
                    double pA = 0.0, pB = 0.0;
                    if (inA) {
                        if (e > 0.0 && wA > 0.0) {
                            const double q = dfc[vA];
                            pA = fmin(e, fmin(wA, q));
                            if (pA > 0.0) { dfc[vA] = q - pA; e -= pA; }
                        }
                        revA[vA] = pA;
                    }
                    __syncthreads();
                    if (inB) {
                        if (e > 0.0 && wB > 0.0) {
                            const double q = dfc[vB];
                            pB = fmin(e, fmin(wB, q));
                            if (pB > 0.0) { dfc[vB] = q - pB; e -= pB; }
                        }
                        revB[vB] = pB;
                    }
                    __syncthreads();
                    wA = (wA - pA) + (inA ? revB[t] : 0.0); /* my neighbour along dA pushed back to me along dB */
                    wB = (wB - pB) + (inB ? revA[t] : 0.0);
                    __builtin_nontemporal_store(wA, &plane0[(int64_t)dA * MGC_TV]);
                    __builtin_nontemporal_store(wB, &plane0[(int64_t)dB * MGC_TV]);
                    if (wA > 0.0) m |= 1u << dA;
                    if (wB > 0.0) m |= 1u << dB;
                });
                __syncthreads();
                exc_out = e;
                snk_out = dfc[t];
                __syncthreads(); /* (dfc is rewritten by the next tile) */
            } else {
                mgcw_static_for<MGC26_NDIR>([&](auto dc) __attribute__((always_inline)) {
                    constexpr int d = decltype(dc)::value;
                    const double w = weight(dc);
                    __builtin_nontemporal_store(w, &plane0[(int64_t)d * MGC_TV]);
                    if (w > 0.0) m |= 1u << d;
                    if (d % 4 == 3) asm volatile("" ::: "memory"); /* four weights in flight, not 26: 26 keep 151 VGPRs alive (one workgroup per CU) */
                });
            }
        }
        if constexpr (!FULL && !PRE6) vote();
        const int64_t v = (int64_t)tile * MGC_TV + t;
        const int any_sink = tbits & 2, any_exc = tbits & 1;
        if constexpr (!FULL && !PRE6) { exc_out = tr > 0.0 ? tr : 0.0; snk_out = tr < 0.0 ? -tr : 0.0; }
        const MGC_KARG MgcLattice* const Ls = mgc_kernarg_again<MgcLattice>(0);
        const MGC_KARG MgcBuildArgs* const As = mgc_kernarg_again<MgcBuildArgs>(MGC_BUILD_ARGS_OFFSET);
        __builtin_nontemporal_store(exc_out, &Ls->excess[v]);
        if constexpr (!FULL) {
            /* 6-neighbourhood: the merged t-links and the residual sink links of a tile are only READ where the tile holds a t-link
             * of the sign in question (A.tflags, status bit MGC_ST_SINK: k_discharge_w, k_cut_value6, k_validate ...), so they are only
             * written there: 16 of the 79 bytes per voxel this kernel writes, for 98 % of the tiles of a marker-seeded volume */
            if (tbits & 3) As->tr0[v] = tr;
            if (any_sink) Ls->sink[v] = snk_out; /* (what a pre-push left of it) */
        } else {
            As->tr0[v] = tr;       /* (as built: the cut value and the invariant check start from the merged t-link) */
            Ls->sink[v] = snk_out; /* (what the pre-push left of it) */
        }
        if constexpr (!FULL) {
            /* is every n-link of the volume residual?  (then the first global relabel is a distance transform, mgc_dt_ops.inl) */
            const uint32_t need = (gx > 0 ? 1u : 0u) | (gx + 1 < L.dx ? 2u : 0u) | (gy > 0 ? 4u : 0u) | (gy + 1 < L.dy ? 8u : 0u) |
                                  (gz > 0 ? 16u : 0u) | (gz + 1 < L.dz ? 32u : 0u);
            if (__ballot(valid && (m & need) != need) != 0ull && (t & 63) == 0) atomicAdd(&Ls->count[MGC_CNT_NOT_FULL], 1);
            if (snk_out > 0.0) m |= MGC_MASK_SINK;
            Ls->rmask[v] = (uint8_t)m;
        } else {
            if (snk_out > 0.0) m |= MGC26_MASK_SINK;
            Ls->rmask32[v] = m;
        }
        /* (labels are not initialised here: every solve starts by filling them, mgc_driver.inl) */
        if (!FULL && t < 6 * MGC_TF / 2) /* 192 lanes x 16 bytes clear the 6x64 outbox */
            *(double2*)(Ls->obox + (int64_t)tile * 6 * MGC_TF + t * 2) = make_double2(0.0, 0.0);
        if (t == 0) {
            Ls->oflags[tile] = 0;
            Ls->stamp[tile] = 0;
            Ls->rstamp[tile] = 0;
            Ls->status[tile] = (any_sink ? MGC_ST_SINK : 0u) | (any_exc ? MGC_ST_EXCESS : 0u) | ((!FULL && (tbits & 1)) ? MGC_ST_SOURCE : 0u);
            As->tflags[tile] = (uint8_t)((any_exc ? 1 : 0) | (any_sink ? 2 : 0));
            sink_tiles += any_sink ? 1 : 0;
            wall_tiles += weak_voxels >= MGC_WALL_VOXELS ? 1 : 0;
        }
        /* flow constant: only voxels whose t-links were merged more than once contribute (regional term + marker, fg and bg
         * marker on one voxel): most tiles skip the ten barriers of the tree sum */
        if (tbits & 4) { /* uniform */
            const double s = mgc_block_sum(fc, scratch);
            if (t == 0) As->fpart[tile] = mgc_owned(L, tile) ? s : 0.0;
            __syncthreads();
        } else if (t == 0) {
            As->fpart[tile] = 0.0;
        }
        if constexpr (FULL) __syncthreads(); /* the weights above read the image tile in LDS, the next tile's load overwrites it (the
                                                6-neighbourhood path has its vote barrier behind the weights) */
        if (has_next) stage_f32(t); /* everybody is past this tile's reads of the block; this tile's stores are still on their way */
    }
    if (!FULL && t_lane == 0 && sink_tiles) atomicAdd(&L.count[MGC_CNT_SINK_TILES], sink_tiles); /* once per workgroup (mgc_build: exact_sink_tiles) */
    if (!FULL && t_lane == 0 && wall_tiles) atomicAdd(&L.count[MGC_CNT_WALL_TILES], wall_tiles);
}

/* One kernel per (neighbourhood, boundary term): g(.) is straight-line code, and every instance gets the registers ITS term
 * needs (one kernel with a switch is allocated for the power terms: 151 VGPRs, one workgroup per CU for all nine). */
#ifndef MGC_BUILD_WAVES6
#define MGC_BUILD_WAVES6 6 /* three workgroups per CU: 80 VGPRs (73 used, no scratch, for the exponential term; the power terms -- pow() is the largest g(.) -- get
                              two waves less and 128); measured 3.74 ms vs 4.28 ms at 4 (and 4.35 at 5) for 512^3 */
#endif
#ifndef MGC_BUILD_WAVES26
#define MGC_BUILD_WAVES26 4 /* two workgroups per CU: 128 VGPRs */
#endif
template <bool FULL, int TERM, bool PRE6 = false, bool TABLE = true> /* FULL: 26-neighbourhood; PRE6: 6-neighbourhood with the pre-push (regional term) */
__global__ __launch_bounds__(MGC_TV, FULL ? MGC_BUILD_WAVES26 : (PRE6 ? 4 : ((TERM == MGC_TERM_DIFFERENCE_POWER || TERM == MGC_TERM_MAXIMUM_POWER) ? MGC_BUILD_WAVES6 - 2 : MGC_BUILD_WAVES6))) void k_build(MgcLattice L, MgcBuildArgs A)
{
    __shared__ double img[1000]; /* 10x10x10: tile + one-voxel halo, already |.|'d for the maximum terms */
    __shared__ double scratch[MGC_TV];
    __shared__ double wf[FULL ? 1 : 3 * 576]; /* 6-neighbourhood: the forward n-link weights of the tile and its lower faces */
    __shared__ int tflag;
    __shared__ double pre_lds[FULL ? 7 * MGC_TV : (PRE6 ? 5 * MGC_TV : 1)]; /* pre-push: residual sink capacities + two double-buffered hand-off planes (+ full neighbourhood: the weight exchange, two buffers) */
    k_build_tiles<FULL, TERM, PRE6, TABLE>(L, A, img, scratch, wf, &tflag, pre_lds);
}

template <bool FULL>
static void mgc_launch_build(int term, int grid, hipStream_t stream, const MgcLattice& L, const MgcBuildArgs& A)
{
    const bool pre6 = !FULL && A.prepush && A.prob; /* a regional term on the 6-neighbourhood: the instance with the pre-push */
    if (!A.lut) { /* no table set: the instances without one (only these four terms are ever evaluated by table: for the others they would be the same code twice) */
        switch (term) {
#define MGC_BUILD_CASE_NT(T) case T: \
            if (pre6) hipLaunchKernelGGL((k_build<false, T, true, false>), dim3(grid), dim3(MGC_TV), 0, stream, L, A); \
            else hipLaunchKernelGGL((k_build<FULL, T, false, false>), dim3(grid), dim3(MGC_TV), 0, stream, L, A); \
            return;
        MGC_BUILD_CASE_NT(MGC_TERM_DIFFERENCE_EXPONENTIAL)
        MGC_BUILD_CASE_NT(MGC_TERM_DIFFERENCE_POWER)
        MGC_BUILD_CASE_NT(MGC_TERM_MAXIMUM_EXPONENTIAL)
        MGC_BUILD_CASE_NT(MGC_TERM_MAXIMUM_POWER)
#undef MGC_BUILD_CASE_NT
        default: break;
        }
    }
    if (pre6) {
        switch (term) {
#define MGC_BUILD_CASE6(T) case T: hipLaunchKernelGGL((k_build<false, T, true>), dim3(grid), dim3(MGC_TV), 0, stream, L, A); return;
        MGC_BUILD_CASE6(MGC_TERM_NONE)
        MGC_BUILD_CASE6(MGC_TERM_DIFFERENCE_LINEAR)
        MGC_BUILD_CASE6(MGC_TERM_DIFFERENCE_EXPONENTIAL)
        MGC_BUILD_CASE6(MGC_TERM_DIFFERENCE_DIVISION)
        MGC_BUILD_CASE6(MGC_TERM_DIFFERENCE_POWER)
        MGC_BUILD_CASE6(MGC_TERM_MAXIMUM_LINEAR)
        MGC_BUILD_CASE6(MGC_TERM_MAXIMUM_EXPONENTIAL)
        MGC_BUILD_CASE6(MGC_TERM_MAXIMUM_DIVISION)
        default: hipLaunchKernelGGL((k_build<false, MGC_TERM_MAXIMUM_POWER, true>), dim3(grid), dim3(MGC_TV), 0, stream, L, A); return;
#undef MGC_BUILD_CASE6
        }
    }
    switch (term) {
#define MGC_BUILD_CASE(T) case T: hipLaunchKernelGGL((k_build<FULL, T>), dim3(grid), dim3(MGC_TV), 0, stream, L, A); break;
    MGC_BUILD_CASE(MGC_TERM_NONE)
    MGC_BUILD_CASE(MGC_TERM_DIFFERENCE_LINEAR)
    MGC_BUILD_CASE(MGC_TERM_DIFFERENCE_EXPONENTIAL)
    MGC_BUILD_CASE(MGC_TERM_DIFFERENCE_DIVISION)
    MGC_BUILD_CASE(MGC_TERM_DIFFERENCE_POWER)
    MGC_BUILD_CASE(MGC_TERM_MAXIMUM_LINEAR)
    MGC_BUILD_CASE(MGC_TERM_MAXIMUM_EXPONENTIAL)
    MGC_BUILD_CASE(MGC_TERM_MAXIMUM_DIVISION)
    default: hipLaunchKernelGGL((k_build<FULL, MGC_TERM_MAXIMUM_POWER>), dim3(grid), dim3(MGC_TV), 0, stream, L, A); break;
#undef MGC_BUILD_CASE
    }
}


/* direction index of the arc i -> j (node ids) in a lattice with `ndir` neighbours, or -1 */
MGC_HD int mgc_arc_direction(const MgcLattice& L, int64_t i, int64_t j)
{
    const int64_t xi = i % L.dx, yi = (i / L.dx) % L.dy, zi = i / (L.dx * L.dy);
    const int64_t xj = j % L.dx, yj = (j / L.dx) % L.dy, zj = j / (L.dx * L.dy);
    const int64_t dz = zj - zi, dy = yj - yi, dx = xj - xi;
    if (dz < -1 || dz > 1 || dy < -1 || dy > 1 || dx < -1 || dx > 1 || (!dz && !dy && !dx)) return -1;
    if (L.ndir == 6) {
        if ((dz != 0) + (dy != 0) + (dx != 0) != 1) return -1;
        return dx ? (dx > 0 ? 1 : 0) : (dy ? (dy > 0 ? 3 : 2) : (dz > 0 ? 5 : 4));
    }
    const int c = (int)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1));
    return c < 13 ? c : c - 1;
}

/* plug-in path: explicit lattice edges accumulate like Graph::sum_edge (graph.h:457-480): every arc slot receives its
 * contributions ONE AFTER THE OTHER in call order, on top of the weight k_build left there -- the same floating point
 * additions in the same order as the reference, so repeated edges give bit-identical capacities (an atomicAdd per
 * contribution would add them in a non-deterministic order).  The host sorted the contributions by slot (stable);
 * run r covers contributions [run[r], run[r + 1]). */
__global__ void k_add_edges(MgcLattice L, int64_t n_runs, const int64_t* slot, const double* val, const int64_t* run)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = slot[run[r]];
        double v = L.rcap[o];
        for (int64_t k = run[r]; k < run[r + 1]; ++k) v += val[k];
        L.rcap[o] = v;
        L.cap0[o] = v;
    }
}

/* refresh rmask after explicit edges were added */
__global__ __launch_bounds__(MGC_TV) void k_refresh_mask(MgcLattice L)
{
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        const int t = threadIdx.x;
        const bool snk = (L.ndir != 6 || (L.status[tile] & MGC_ST_SINK)) && L.sink[(int64_t)tile * MGC_TV + t] > 0.0; /* (the plane is only written where a sink link exists) */
        uint32_t m = 0;
        for (int d = 0; d < L.ndir; ++d)
            if (L.rcap[((int64_t)tile * L.ndir + d) * MGC_TV + t] > 0.0) m |= 1u << d;
        if (L.ndir == 6) L.rmask[(int64_t)tile * MGC_TV + t] = (uint8_t)(m | (snk ? MGC_MASK_SINK : 0));
        else L.rmask32[(int64_t)tile * MGC_TV + t] = m | (snk ? MGC26_MASK_SINK : 0u);
    }
}

/* ======================================================================================
 * read-out
 * ==================================================================================== */
__global__ void k_labels(MgcLattice L, uint8_t* out)
{
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < L.nvox; id += (int64_t)gridDim.x * blockDim.x) {
        int tile, loc;
        mgc_node_to_tile(L, id, tile, loc);
        out[id] = L.height[(int64_t)tile * MGC_TV + loc] < MGC_HINF ? 0 : 1;
    }
}

/* Capacity, as built, of the arc that leaves voxel (gz, gy, gx) -- lane t of `tile` -- in direction d.  The built
 * capacities are only materialised (L.cap0) when explicit edges were added on top of the boundary term; otherwise they
 * are a pure function of the image and are re-evaluated here with exactly the operations of k_build (48 resp. 208 B per
 * voxel less to write and to hold). */
__device__ __forceinline__ double mgc_built_capacity(const MgcLattice& L, const MgcBuildArgs& A, int tile, int t, int64_t gz, int64_t gy, int64_t gx, int d)
{
    if (L.cap0) return L.cap0[((int64_t)tile * L.ndir + d) * MGC_TV + t];
    if (A.term == MGC_TERM_NONE) return 0.0;
    const bool take_abs = (A.term == MGC_TERM_MAXIMUM_LINEAR || A.term == MGC_TERM_MAXIMUM_EXPONENTIAL || A.term == MGC_TERM_MAXIMUM_POWER);
    int dz, dy, dx;
    bool fwd;
    if (L.ndir == 6) {
        dz = (d >> 1) == 2 ? ((d & 1) ? 1 : -1) : 0;
        dy = (d >> 1) == 1 ? ((d & 1) ? 1 : -1) : 0;
        dx = (d >> 1) == 0 ? ((d & 1) ? 1 : -1) : 0;
        fwd = (d & 1) != 0;
    } else {
        mgc26_offset(d, dz, dy, dx);
        fwd = d >= 13;
    }
    const double me = mgc_load_as_double(A.image, A.img_dtype, (gz * L.dy + gy) * L.dx + gx, take_abs);
    const double nb = mgc_load_as_double(A.image, A.img_dtype, ((gz + dz) * L.dy + (gy + dy)) * L.dx + (gx + dx), take_abs);
    double w = mgc_boundary_g(A.term, fwd ? me : nb, fwd ? nb : me, A.p0, A.lut, A.lut_n); /* (lower voxel, upper voxel) like the reference slices */
    if (A.has_spacing) w = w / (L.ndir == 6 ? A.inv_axis[d >> 1] : A.div26[d]);
    return w;
}

/* capacity of the cut (S = label 1, T = label 0) from the capacities as built */
__global__ __launch_bounds__(MGC_TV) void k_cut_value(MgcLattice L, MgcBuildArgs A, const double* tr0, const uint8_t* labels, double* part)
{
    __shared__ double scratch[MGC_TV];
    const int t = threadIdx.x;
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const int lz = t >> 6, ly = (t >> 3) & 7, lx = t & 7;
        const int64_t gz = (int64_t)tz * 8 + lz, gy = (int64_t)ty * 8 + ly, gx = (int64_t)tx * 8 + lx;
        double s = 0.0;
        if (gz < L.dz && gy < L.dy && gx < L.dx && mgc_owned(L, tile)) {
            const int64_t id = (gz * L.dy + gy) * L.dx + gx;
            const double tr = tr0[(int64_t)tile * MGC_TV + t];
            if (labels[id]) { /* source side: pays its sink link and every n-link into T */
                if (tr < 0.0) s += -tr;
                if (L.ndir == 6) {
                    const int64_t step[6] = {-1, 1, -L.dx, L.dx, -L.dx * L.dy, L.dx * L.dy};
#pragma unroll
                    for (int d = 0; d < 6; ++d) {
                        const int64_t c = (d >> 1) == 0 ? gx : ((d >> 1) == 1 ? gy : gz);
                        const int64_t lim = (d >> 1) == 0 ? L.dx : ((d >> 1) == 1 ? L.dy : L.dz);
                        const bool has = (d & 1) ? (c + 1 < lim) : (c > 0);
                        if (has && !labels[id + step[d]]) s += mgc_built_capacity(L, A, tile, t, gz, gy, gx, d);
                    }
                } else {
                    for (int d = 0; d < MGC26_NDIR; ++d) {
                        int dz, dy, dx;
                        mgc26_offset(d, dz, dy, dx);
                        const int64_t nz = gz + dz, ny = gy + dy, nx = gx + dx;
                        if (nz >= 0 && nz < L.dz && ny >= 0 && ny < L.dy && nx >= 0 && nx < L.dx && !labels[(nz * L.dy + ny) * L.dx + nx])
                            s += mgc_built_capacity(L, A, tile, t, gz, gy, gx, d);
                    }
                }
            } else if (tr > 0.0) { /* sink side: pays its source link */
                s += tr;
            }
        }
        /* only tiles the cut passes through (or that hold t-links on the paying side) contribute: the others skip the ten
         * barriers of the tree sum (the sum of zeros is the same 0.0) */
        if (__syncthreads_or(s != 0.0)) {
            const double tot = mgc_block_sum(s, scratch);
            if (t == 0) part[tile] = tot;
            __syncthreads();
        } else if (t == 0) {
            part[tile] = 0.0;
        }
    }
}

/* the form of k_cut_value for graphs with a regional term (and every 26-neighbourhood graph), one WAVE per tile (four tiles per
 * workgroup, no barrier).  With a regional term EVERY
 * voxel pays a t-link, so no tile can be skipped; but only the tiles the cut passes through pay n-links.  A tile whose label
 * summary (tsum, k_labels8) says "all on one side" and whose 26 neighbour tiles say the same pays t-links only: one coalesced
 * read of the merged t-links -- no label volume, no 26 byte loads per voxel.  Every other tile takes the general path.
 * The sum of a tile has a fixed order (a lane adds its eight voxels in z order, then a shuffle tree over the lanes); the
 * workgroup form spent its time in ten barriers per tile behind one dependent load (2.1 ms at 512^3). */
/* the general path of k_cut_value26 for ONE tile, by one wave.  The tile's 10 x 10 x 10 block of label bytes goes to LDS once (100 rows of
 * ten bytes, two rows per lane; 1 = source side or outside the volume: nothing is paid towards it), and so does the image block (as doubles,
 * |.|'d for the maximum terms): read from the volumes per voxel and direction it was 208 scattered byte loads and up to 416 dependent image
 * loads per lane and tile.  The n-links a z-layer of the tile pays are first LISTED -- (voxel, direction) pairs, a lane's pairs in ascending
 * direction, the lanes in order -- then evaluated 64 at a time, one pair per lane, and then summed up by their owners in the order of the
 * list: a tile the cut passes through pays along ~8 % of its 13 312 (voxel, direction) pairs, and evaluating g(.) under the mask of whoever
 * pays along direction d in layer z was 208 evaluations per wave for ~11 waves' worth of pairs (config 3: 1.3 of the 1.5 ms of the cut value).
 * The additions of a lane are the same in the same order as without the list (sink link, then the directions in ascending order, layer by
 * layer; a shuffle tree over the lanes): the value is bit for bit the same.  A layer with more than MGC_CUT_PAIRS pairs is evaluated in place.
 * Returns the tile's sum in lane 0. */
#ifndef MGC_CUT_PAIRS
#define MGC_CUT_PAIRS 512 /* (a library built with 16 runs the GPU tests through the in-place path of crowded layers as well: profiles/r6_cut_pairs_overflow_path.txt) */
#endif
struct MgcCutLds {
    uint8_t lb[1000];           /* label bytes of the block  */
    uint16_t pair[MGC_CUT_PAIRS]; /* (lane << 5) | direction   */
    double ib[1000];            /* image block               */
    double w[MGC_CUT_PAIRS];    /* the pairs' capacities     */
};
__device__ __forceinline__ void mgc_dir_offset(int ndir, int d, int& dz, int& dy, int& dx)
{
    if (ndir == 6) {
        dz = (d >> 1) == 2 ? ((d & 1) ? 1 : -1) : 0;
        dy = (d >> 1) == 1 ? ((d & 1) ? 1 : -1) : 0;
        dx = (d >> 1) == 0 ? ((d & 1) ? 1 : -1) : 0;
    } else {
        mgc26_offset(d, dz, dy, dx);
    }
}
__device__ __forceinline__ double mgc_cut_tile_general(const MgcLattice& L, const MgcBuildArgs& A, const double* tr0, const uint8_t* labels, MgcCutLds& W, int tile, int lane)
{
    int tz, ty, tx;
    mgc_tile_coords(L, tile, tz, ty, tx);
    uint8_t* const lb = W.lb;
    double* const ib = W.ib;
    const bool from_image = !L.cap0 && A.term != MGC_TERM_NONE;
    if (from_image) {
        const bool take_abs = (A.term == MGC_TERM_MAXIMUM_LINEAR || A.term == MGC_TERM_MAXIMUM_EXPONENTIAL || A.term == MGC_TERM_MAXIMUM_POWER);
        for (int k = lane; k < 1000; k += 64) {
            const int64_t rz = (int64_t)tz * 8 + k / 100 - 1, ry = (int64_t)ty * 8 + (k / 10) % 10 - 1, rx = (int64_t)tx * 8 + k % 10 - 1;
            double v = 0.0;
            if (rz >= 0 && rz < L.dz && ry >= 0 && ry < L.dy && rx >= 0 && rx < L.dx) v = mgc_load_as_double(A.image, A.img_dtype, (rz * L.dy + ry) * L.dx + rx, take_abs);
            ib[k] = v;
        }
    }
    const bool has_tlinks = L.ndir != 6 || A.tflags[tile] != 0; /* (6-neighbourhood: k_build writes tr0 only for tiles that hold a t-link) */
    const int ly = lane >> 3, lx = lane & 7;
    const int64_t gy = (int64_t)ty * 8 + ly, gx = (int64_t)tx * 8 + lx;
    for (int row = lane; row < 100; row += 64) {
        const int64_t rz = (int64_t)tz * 8 + row / 10 - 1, ry = (int64_t)ty * 8 + row % 10 - 1;
        const bool row_in = rz >= 0 && rz < L.dz && ry >= 0 && ry < L.dy;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int64_t rx = (int64_t)tx * 8 + k - 1;
            uint8_t v = 1;
            if (row_in && rx >= 0 && rx < L.dx) v = labels[(rz * L.dy + ry) * L.dx + rx];
            lb[row * 10 + k] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    /* the capacity of the arc (voxel l2 of layer lz, direction d) as built: mgc_built_capacity with the two voxels from the block */
    auto capacity = [&](int lz, int l2, int d) -> double {
        int dz, dy, dx;
        mgc_dir_offset(L.ndir, d, dz, dy, dx);
        if (from_image) {
            const int m2 = ((lz + 1) * 10 + ((l2 >> 3) + 1)) * 10 + (l2 & 7) + 1;
            const double a = ib[m2], b = ib[m2 + (dz * 10 + dy) * 10 + dx];
            const bool fwd = L.ndir == 6 ? (d & 1) != 0 : d >= 13;
            double w = mgc_boundary_g(A.term, fwd ? a : b, fwd ? b : a, A.p0, A.lut, A.lut_n); /* (lower voxel, upper voxel) like the reference slices */
            if (A.has_spacing) w = w / (L.ndir == 6 ? A.inv_axis[d >> 1] : A.div26[d]);
            return w;
        }
        return mgc_built_capacity(L, A, tile, lz * 64 + l2, (int64_t)tz * 8 + lz, (int64_t)ty * 8 + (l2 >> 3), (int64_t)tx * 8 + (l2 & 7), d);
    };
    double s = 0.0;
    for (int lz = 0; lz < 8; ++lz) {
        const int64_t gz = (int64_t)tz * 8 + lz;
        const int t = lz * 64 + lane;
        const bool valid = gz < L.dz && gy < L.dy && gx < L.dx;
        const double tr = (valid && has_tlinks) ? tr0[(int64_t)tile * MGC_TV + t] : 0.0;
        const int me = ((lz + 1) * 10 + (ly + 1)) * 10 + lx + 1;
        const bool source_side = valid && lb[me] != 0;
        uint32_t pays = 0; /* the directions along which this voxel pays an n-link: into T (a neighbour outside the volume reads 1) */
        if (source_side) {
            if (L.ndir == MGC26_NDIR) { /* (constant offsets: 26 byte reads with immediate addresses) */
                mgcw_static_for<MGC26_NDIR>([&](auto dc) __attribute__((always_inline)) {
                    constexpr int d = decltype(dc)::value;
                    constexpr int c = d < 13 ? d : d + 1;
                    if (!lb[me + ((c / 9 - 1) * 10 + ((c / 3) % 3 - 1)) * 10 + (c % 3 - 1)]) pays |= 1u << d;
                });
            } else {
                for (int d = 0; d < L.ndir; ++d) {
                    int dz, dy, dx;
                    mgc_dir_offset(L.ndir, d, dz, dy, dx);
                    if (!lb[me + (dz * 10 + dy) * 10 + dx]) pays |= 1u << d;
                }
            }
        }
        if (__ballot(pays != 0u) == 0ull) { /* (uniform) a layer the cut does not pass through */
            if (source_side) { if (tr < 0.0) s += -tr; }
            else if (valid && tr > 0.0) s += tr;
            continue;
        }
        const int mine = __popc(pays);
        int before = mine; /* inclusive scan over the lanes */
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(before, off, 64);
            if (lane >= off) before += v;
        }
        const int total = __shfl(before, 63, 64);
        before -= mine;
        if (total <= MGC_CUT_PAIRS) {
            {
                uint32_t m = pays;
                for (int j = before; m; ++j) {
                    const int d = __builtin_ctz(m);
                    m &= m - 1u;
                    W.pair[j] = (uint16_t)((lane << 5) | d);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < total; i += 64) {
                const int pr = W.pair[i];
                W.w[i] = capacity(lz, pr >> 5, pr & 31);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (source_side) { /* pays its sink link and every n-link into T */
                if (tr < 0.0) s += -tr;
                for (int j = before; j < before + mine; ++j) s += W.w[j];
            } else if (valid && tr > 0.0) { /* sink side: pays its source link */
                s += tr;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* (the next layer rewrites the list) */
            __builtin_amdgcn_wave_barrier();
        } else {
            if (source_side) {
                if (tr < 0.0) s += -tr;
                for (uint32_t m = pays; m; m &= m - 1u) s += capacity(lz, lane, __builtin_ctz(m));
            } else if (valid && tr > 0.0) {
                s += tr;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* (the next tile of this wave rewrites the blocks) */
    __builtin_amdgcn_wave_barrier();
    return s;
}

/* second launch of the two-launch form: the marked tiles (-1.0).  Wave g of W looks at the tiles g, g + W, g + 2 W, ... -- 64 of them at a
 * time, a lane each -- and takes the marked ones in turn.  W is not a multiple of the tile grid's row or layer length (mgc_launch_cut_value26),
 * so a run of marked tiles along any axis -- where the cut runs parallel to it for a while -- is spread over as many waves as it is long:
 * a wave per 64 CONSECUTIVE tiles left whole rows of the sphere's caps to single waves (config 3: 1.30 ms for the ~5 % of the tiles that are
 * marked, 0.25 ms for all the others in the first launch). */
__device__ __forceinline__ void k_cut_value26_marked(const MgcLattice& L, const MgcBuildArgs& A, const double* tr0, const uint8_t* labels, double* part)
{
    __shared__ MgcCutLds lds[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t W = (int64_t)gridDim.x * 4, g = (int64_t)blockIdx.x * 4 + wv;
    for (int64_t base = 0; base < L.ntiles; base += W * 64) {
        const int64_t mine = base + g + (int64_t)lane * W;
        unsigned long long todo = __ballot(mine < L.ntiles && part[mine] == -1.0);
        while (todo) {
            const int tile = (int)(base + g + (int64_t)__builtin_ctzll(todo) * W);
            todo &= todo - 1ull;
            const double s = mgc_cut_tile_general(L, A, tr0, labels, lds[wv], tile, lane);
            if (lane == 0) part[tile] = s;
        }
    }
}

template <int MODE> /* 0: every tile in one launch (no label summaries); 1: the tiles that pay t-links only, the others are marked; 2: the marked tiles */
__global__ __launch_bounds__(256) void k_cut_value26(MgcLattice L, MgcBuildArgs A, const double* tr0, const uint8_t* labels, const uint8_t* tsum, double* part)
{
    if constexpr (MODE == 2) {
        k_cut_value26_marked(L, A, tr0, labels, part);
        return;
    }
    __shared__ MgcCutLds lds[MODE == 0 ? 4 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int tile = (int)blockIdx.x * 4 + wv; tile < L.ntiles; tile += (int)gridDim.x * 4) {
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        if (!mgc_owned(L, tile)) { /* (wave-uniform) */
            if (lane == 0) part[tile] = 0.0;
            continue;
        }
        const int side = tsum ? (int)tsum[tile] : 2;
        bool differs = side > 1; /* is any of the 27 tiles around (and including) this one not wholly on `side`? */
        if (lane < 27 && !differs) {
            const int nz = tz + lane / 9 - 1, ny = ty + (lane / 3) % 3 - 1, nx = tx + lane % 3 - 1;
            if (nz >= 0 && nz < L.gz && ny >= 0 && ny < L.gy && nx >= 0 && nx < L.gx) {
                const int nt = mgc_tile_id(L, nz, ny, nx);
                differs = !mgc_owned(L, nt) || (int)tsum[nt] != side;
            }
        }
        const bool general = __ballot(differs) != 0ull;
        if (MODE == 1 && general) { /* (wave-uniform) left to the second launch: a capacity is never negative */
            if (lane == 0) part[tile] = -1.0;
            continue;
        }
        if constexpr (MODE == 0) {
            if (general) {
                const double sg = mgc_cut_tile_general(L, A, tr0, labels, lds[MODE == 0 ? wv : 0], tile, lane);
                if (lane == 0) part[tile] = sg;
                continue;
            }
        }
        const bool has_tlinks = L.ndir != 6 || A.tflags[tile] != 0; /* (6-neighbourhood: k_build writes tr0 only for tiles that hold a t-link) */
        const int ly = lane >> 3, lx = lane & 7;
        const int64_t gy = (int64_t)ty * 8 + ly, gx = (int64_t)tx * 8 + lx;
        double s = 0.0;
#pragma unroll
        for (int lz = 0; lz < 8; ++lz) {
            const int64_t gz = (int64_t)tz * 8 + lz;
            const int t = lz * 64 + lane;
            if (!(gz < L.dz && gy < L.dy && gx < L.dx)) continue;
            const double tr = has_tlinks ? tr0[(int64_t)tile * MGC_TV + t] : 0.0;
            if (side == 1) { if (tr < 0.0) s += -tr; } /* source side: pays its sink link (no n-link leaves the 27 tiles' common side) */
            else if (tr > 0.0) s += tr;                /* sink side: pays its source link */
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) part[tile] = s;
    }
}

/* 6-neighbourhood form, two kernels.  k_cut_filter: one THREAD per tile decides from the label summaries k_labels8 left
 * (tsum; NULL: no summaries, every owned tile is looked at) and from the signs of t-link k_build saw (A.tflags) whether the
 * tile can contribute to the cut at all; the few that can (the tiles the cut passes through, the marker tiles) go to a
 * list, the others get their zero here.  k_cut_value6: per listed tile, labels straight from the tile-major distance
 * labels (own tile + the six face layers next door in LDS), the merged t-links only where a paying sign exists.  Same
 * additions in the same order as k_cut_value: the value is bit-identical. */
__global__ void k_cut_filter(MgcLattice L, MgcBuildArgs A, const uint8_t* tsum, double* part, int list, int cnt)
{
    for (int base = blockIdx.x * blockDim.x; base < L.ntiles; base += gridDim.x * blockDim.x) { /* uniform per block */
        const int tile = base + (int)threadIdx.x;
        bool take = false;
        if (tile < L.ntiles) {
            const bool owned = mgc_owned(L, tile);
            take = owned;
            if (owned && tsum) {
                const uint32_t tf = A.tflags[tile], sm = tsum[tile];
                if (sm == 0) take = (tf & 1u) != 0; /* all on the sink side: only source links are paid */
                else if (sm == 1 && !(tf & 2u)) {  /* all on the source side, no sink link: pays n-links into a neighbour tile at most */
                    int tz, ty, tx;
                    mgc_tile_coords(L, tile, tz, ty, tx);
                    take = false;
                    for (int f = 0; f < 6; ++f) {
                        const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
                        if (nt >= 0 && !(mgc_owned(L, nt) && tsum[nt] == 1)) take = true; /* (a ghost tile only mirrors one voxel layer) */
                    }
                }
            }
            if (!take) part[tile] = 0.0;
        }
        const unsigned long long m = __ballot(take);
        if (m) {
            const int sh = (int)(blockIdx.x & (unsigned)(L.nshard - 1));
            int pos = 0;
            if ((threadIdx.x & 63) == 0) pos = atomicAdd(mgc_counter(L, cnt, sh), __popcll(m));
            pos = __shfl(pos, 0);
            if (take) L.list[list][(int64_t)sh * L.shard_cap + pos + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = tile;
        }
    }
}

__global__ __launch_bounds__(MGC_TV) void k_cut_value6(MgcLattice L, MgcBuildArgs A, const double* tr0, int list, int cnt, double* part, int clear_cnt)
{
    mgc_clear_counter(L, clear_cnt); /* the slot the next tile filter counts into (HipDevT::fslot) */
    __shared__ double scratch[MGC_TV];
    __shared__ int32_t hs[1000];
    const int t = threadIdx.x;
    MgcListView view;
    const int n = mgc_list_view(L, cnt, view);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int tile = mgc_list_at(L, list, view, i);
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const uint32_t tf = A.tflags[tile];
        const int lz = t >> 6, ly = (t >> 3) & 7, lx = t & 7;
        const int me = mgc_hs_index(lz, ly, lx);
        const int32_t hme = L.height[(int64_t)tile * MGC_TV + t];
        hs[me] = hme;
        if (t < 6 * MGC_TF) {
            const int f = t >> 6, k = t & 63;
            const int nt = mgc_tile_nbr(L, tz, ty, tx, f);
            const int mine = mgc_face_voxel(f, k);
            hs[mgc_hs_index(mine >> 6, (mine >> 3) & 7, mine & 7) + mgc_hs_step(f)] =
                nt < 0 ? MGC_HINF : L.height[(int64_t)nt * MGC_TV + mgc_face_voxel(f ^ 1, k)];
        }
        __syncthreads();
        const int64_t gz = (int64_t)tz * 8 + lz, gy = (int64_t)ty * 8 + ly, gx = (int64_t)tx * 8 + lx;
        double s = 0.0;
        if (gz < L.dz && gy < L.dy && gx < L.dx) {
            if (hme >= MGC_HINF) { /* source side: pays its sink link and every n-link into T */
                if (tf & 2u) {
                    const double tr = tr0[(int64_t)tile * MGC_TV + t];
                    if (tr < 0.0) s += -tr;
                }
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                    const int64_t c = (d >> 1) == 0 ? gx : ((d >> 1) == 1 ? gy : gz);
                    const int64_t lim = (d >> 1) == 0 ? L.dx : ((d >> 1) == 1 ? L.dy : L.dz);
                    const bool has = (d & 1) ? (c + 1 < lim) : (c > 0);
                    if (has && hs[me + mgc_hs_step(d)] < MGC_HINF) s += mgc_built_capacity(L, A, tile, t, gz, gy, gx, d);
                }
            } else if (tf & 1u) { /* sink side: pays its source link */
                const double tr = tr0[(int64_t)tile * MGC_TV + t];
                if (tr > 0.0) s += tr;
            }
        }
        if (__syncthreads_or(s != 0.0)) {
            const double tot = mgc_block_sum(s, scratch);
            if (t == 0) part[tile] = tot;
            __syncthreads();
        } else if (t == 0) {
            part[tile] = 0.0;
        }
    }
}

/* label read-out for rows that are whole runs of eight voxels (D2 a multiple of 8).  One wave per GROUP of eight tiles along x:
 * lane (z, y) turns the 32 bytes of its row in each of the eight tiles into 8 label bytes -- 64 consecutive bytes of the
 * C-order output, written as whole lines -- and the wave leaves a one-byte summary per tile for k_cut_filter
 * (0: every voxel on the sink side, 1: every voxel on the source side, 2: both). */
__global__ __launch_bounds__(256) void k_labels8(MgcLattice L, uint8_t* out, uint8_t* tsum)
{
    const int lane = threadIdx.x & 63;
    const int gxg = (L.gx + 7) / 8, ngroups = L.gz * L.gy * gxg;
    for (int g = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); g < ngroups; g += (int)gridDim.x * 4) {
        const int tx0 = (g % gxg) * 8, ty = (g / gxg) % L.gy, tz = g / (gxg * L.gy);
        const int64_t z = (int64_t)tz * 8 + (lane >> 3), y = (int64_t)ty * 8 + (lane & 7);
        const bool inside = z < L.dz && y < L.dy; /* (rows are whole: x never leaves the volume) */
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = 0ull;
            if (tx0 + k < L.gx) {
                const int4* hp = (const int4*)(L.height + (int64_t)mgc_tile_id(L, tz, ty, tx0 + k) * MGC_TV + lane * 8);
                const int4 a = hp[0], b = hp[1];
                v[k] = (a.x < MGC_HINF ? 0ull : 1ull) | (a.y < MGC_HINF ? 0ull : 1ull << 8) | (a.z < MGC_HINF ? 0ull : 1ull << 16) |
                       (a.w < MGC_HINF ? 0ull : 1ull << 24) | (b.x < MGC_HINF ? 0ull : 1ull << 32) | (b.y < MGC_HINF ? 0ull : 1ull << 40) |
                       (b.z < MGC_HINF ? 0ull : 1ull << 48) | (b.w < MGC_HINF ? 0ull : 1ull << 56);
            }
        }
        unsigned long long* const row = (unsigned long long*)(out + (z * L.dy + y) * L.dx + (int64_t)tx0 * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (tx0 + k >= L.gx) break; /* uniform */
            if (inside) row[k] = v[k];
            const bool any1 = __ballot(inside && v[k] != 0ull) != 0ull, any0 = __ballot(inside && v[k] != 0x0101010101010101ull) != 0ull;
            if (lane == 0) tsum[mgc_tile_id(L, tz, ty, tx0 + k)] = (uint8_t)(any1 ? (any0 ? 2 : 1) : 0);
        }
    }
}

/* ======================================================================================
 * invariants of a maximum preflow (mgc_validate; the reference's Graph::test_consistency, maxflow.cpp:610-682, in spirit)
 * ==================================================================================== */
struct MgcValidateOut {
    unsigned long long cnt[8]; /* voxels, negative, active excess, residual arcs across, sink links across, pair, node, pending outbox */
    unsigned long long max_pair_bits, max_node_bits; /* non-negative doubles compare like their bit patterns */
    double sink_cap_used; /* sum of the built sink links that carry flow: scale of the rounding in flow_into_sink */
};

__device__ __forceinline__ int mgc_label_of(const MgcLattice& L, int64_t gz, int64_t gy, int64_t gx)
{
    const int tile = mgc_tile_id(L, (int)(gz >> 3), (int)(gy >> 3), (int)(gx >> 3));
    return L.height[(int64_t)tile * MGC_TV + mgc_local((int)(gz & 7), (int)(gy & 7), (int)(gx & 7))] < MGC_HINF ? 0 : 1;
}

__global__ __launch_bounds__(MGC_TV) void k_validate(MgcLattice L, MgcBuildArgs A, const double* tr0, double* part, MgcValidateOut* out)
{
    __shared__ double scratch[MGC_TV];
    __shared__ unsigned long long sc[8];
    __shared__ unsigned long long smax[2];
    const int t = threadIdx.x;
    const double TOL = 1e-9;
    for (int tile = blockIdx.x; tile < L.ntiles; tile += gridDim.x) {
        if (t < 8) sc[t] = 0;
        if (t < 2) smax[t] = 0;
        __syncthreads();
        int tz, ty, tx;
        mgc_tile_coords(L, tile, tz, ty, tx);
        const int lz = t >> 6, ly = (t >> 3) & 7, lx = t & 7;
        const int64_t gz = (int64_t)tz * 8 + lz, gy = (int64_t)ty * 8 + ly, gx = (int64_t)tx * 8 + lx;
        double into_sink = 0.0, cap_used = 0.0;
        const bool owned = mgc_owned(L, tile);
        if (owned && gz < L.dz && gy < L.dy && gx < L.dx) {
            const int64_t v = (int64_t)tile * MGC_TV + t;
            /* (6-neighbourhood: k_build writes these two planes only for tiles that hold a t-link of the sign) */
            const uint32_t tfl = L.ndir == 6 ? A.tflags[tile] : 3u;
            const double e = L.excess[v], sk = (tfl & 2u) ? L.sink[v] : 0.0, tr = tfl ? tr0[v] : 0.0;
            const int lab = L.height[v] < MGC_HINF ? 0 : 1;
            const double src0 = tr > 0.0 ? tr : 0.0, snk0 = tr < 0.0 ? -tr : 0.0;
            unsigned neg = (e < 0.0) | (sk < 0.0);
            unsigned across = 0;
            double outflow = 0.0, scale = fmax(fmax(src0, snk0), fmax(e, 1e-300));
            into_sink = snk0 - sk;
            if (into_sink != 0.0) cap_used = snk0;
            for (int d = 0; d < L.ndir; ++d) {
                int dz, dy, dx;
                if (L.ndir == 6) {
                    dz = (d >> 1) == 2 ? ((d & 1) ? 1 : -1) : 0;
                    dy = (d >> 1) == 1 ? ((d & 1) ? 1 : -1) : 0;
                    dx = (d >> 1) == 0 ? ((d & 1) ? 1 : -1) : 0;
                } else {
                    mgc26_offset(d, dz, dy, dx);
                }
                const double r = L.rcap[((int64_t)tile * L.ndir + d) * MGC_TV + t];
                neg |= r < 0.0;
                const int64_t nz = gz + dz, ny = gy + dy, nx = gx + dx;
                if (nz < 0 || nz >= L.dz || ny < 0 || ny >= L.dy || nx < 0 || nx >= L.dx) continue;
                const double c = mgc_built_capacity(L, A, tile, t, gz, gy, gx, d);
                outflow += c - r;
                scale = fmax(scale, fmax(c, r));
                if (lab == 1 && r > 0.0 && mgc_label_of(L, nz, ny, nx) == 0) across++;
                /* the arc pair conserves its total capacity; each pair once (from its lower end), both ends owned */
                const bool fwd = L.ndir == 6 ? (d & 1) != 0 : d >= 13;
                const int nt = mgc_tile_id(L, (int)(nz >> 3), (int)(ny >> 3), (int)(nx >> 3));
                if (fwd && mgc_owned(L, nt)) {
                    const int nl = mgc_local((int)(nz & 7), (int)(ny & 7), (int)(nx & 7));
                    const int dr = L.ndir == 6 ? (d ^ 1) : (25 - d);
                    const double rr = L.rcap[((int64_t)nt * L.ndir + dr) * MGC_TV + nl];
                    const double cr = L.cap0 ? L.cap0[((int64_t)nt * L.ndir + dr) * MGC_TV + nl] : c; /* the built-in terms are symmetric */
                    const double err = fabs((r + rr) - (c + cr)) / fmax(c + cr, 1e-300);
                    if (err == err) { /* (NaN capacities of a constant image under a *_linear term: nothing to conserve) */
                        atomicMax(&smax[0], (unsigned long long)__double_as_longlong(err));
                        if (err > TOL) atomicAdd(&sc[5], 1ull);
                    }
                }
            }
            /* what came from the source = what is still here + what went into the sink + what left along the n-links */
            const double nerr = fabs(src0 - e - into_sink - outflow) / scale;
            if (nerr == nerr) {
                atomicMax(&smax[1], (unsigned long long)__double_as_longlong(nerr));
                if (nerr > TOL) atomicAdd(&sc[6], 1ull);
            }
            atomicAdd(&sc[0], 1ull);
            if (neg) atomicAdd(&sc[1], 1ull);
            if (lab == 0 && e > 0.0) atomicAdd(&sc[2], 1ull);
            if (across) atomicAdd(&sc[3], (unsigned long long)across);
            if (lab == 1 && sk > 0.0) atomicAdd(&sc[4], 1ull);
        }
        if (L.obox && t < 6 * MGC_TF && L.obox[(int64_t)tile * 6 * MGC_TF + t] != 0.0) atomicAdd(&sc[7], 1ull);
        const double tot = mgc_block_sum(into_sink, scratch);
        if (t == 0) part[tile] = tot;
        __syncthreads();
        const double used = mgc_block_sum(cap_used, scratch); /* one atomic per tile, not one per voxel with a sink link */
        if (t == 0 && used != 0.0) atomicAdd(&out->sink_cap_used, used);
        __syncthreads();
        if (t < 8 && sc[t]) atomicAdd(&out->cnt[t], sc[t]);
        if (t == 8 && smax[0]) atomicMax(&out->max_pair_bits, smax[0]);
        if (t == 9 && smax[1]) atomicMax(&out->max_node_bits, smax[1]);
        __syncthreads();
    }
}

/* fixed-order sum of n partials by one block */
/* first stage for long vectors: block b adds the contiguous chunk [b * len, (b + 1) * len) in a fixed order */
__global__ __launch_bounds__(256) void k_sum_chunks(const double* part, int64_t n, int64_t len, double* out)
{
    __shared__ double sm[256];
    double s = 0.0;
    const int64_t a = (int64_t)blockIdx.x * len, b = a + len < n ? a + len : n;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) s += part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(MGC_TV) void k_sum_partials(const double* part, int64_t n, double* out)
{
    __shared__ double scratch[MGC_TV];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += MGC_TV) s += part[i];
    const double tot = mgc_block_sum(s, scratch);
    if (threadIdx.x == 0) *out = tot;
}

__global__ void k_get_nweights(MgcLattice L, MgcBuildArgs A, int axis /* array axis 0..2 */, double* out)
{
    const int64_t sh[3] = {L.dz, L.dy, L.dx};
    int64_t osh[3] = {sh[0], sh[1], sh[2]};
    osh[axis] -= 1;
    const int64_t n = osh[0] * osh[1] * osh[2];
    const int d = L.ndir == 6 ? (axis == 2 ? 1 : (axis == 1 ? 3 : 5)) : (axis == 2 ? 13 : (axis == 1 ? 15 : 21)); /* forward direction of that axis */
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = k % osh[2], y = (k / osh[2]) % osh[1], z = k / (osh[2] * osh[1]);
        int tile, loc;
        mgc_node_to_tile(L, (z * L.dy + y) * L.dx + x, tile, loc);
        out[k] = mgc_built_capacity(L, A, tile, loc, z, y, x, d);
    }
}

/* weight of the arc (p, p + offset) for every voxel p, NaN where p + offset is outside (parity read-back) */
__global__ void k_get_nweights_offset(MgcLattice L, MgcBuildArgs A, int dz, int dy, int dx, double* out)
{
    int d;
    if (L.ndir == 6) d = dx ? (dx > 0 ? 1 : 0) : (dy ? (dy > 0 ? 3 : 2) : (dz > 0 ? 5 : 4));
    else { const int c = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1); d = c < 13 ? c : c - 1; }
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < L.nvox; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = id % L.dx, y = (id / L.dx) % L.dy, z = id / (L.dx * L.dy);
        const int64_t nz = z + dz, ny = y + dy, nx = x + dx;
        double w = NAN;
        if (nz >= 0 && nz < L.dz && ny >= 0 && ny < L.dy && nx >= 0 && nx < L.dx) {
            int tile, loc;
            mgc_node_to_tile(L, id, tile, loc);
            w = mgc_built_capacity(L, A, tile, loc, z, y, x, d);
        }
        out[id] = w;
    }
}

__global__ void k_untile_f64(MgcLattice L, const double* tiled, const uint8_t* tflags, double* out)
{
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < L.nvox; id += (int64_t)gridDim.x * blockDim.x) {
        int tile, loc;
        mgc_node_to_tile(L, id, tile, loc);
        /* (6-neighbourhood: k_build leaves the plane of a tile without t-links unwritten) */
        out[id] = (L.ndir != 6 || tflags[tile]) ? tiled[(int64_t)tile * MGC_TV + loc] : 0.0;
    }
}

/* ======================================================================================
 * host side: handle, device policy, C ABI
 * ==================================================================================== */

/* roctx ranges around the stretches of a solve (tracing), for rocprofv3 --marker-trace: opt-in with
 * MEDPY_HIP_ROCTX=1, resolved lazily from the profiler's own library so that nothing is linked or loaded otherwise */
struct MgcRoctx {
    bool tried = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    void resolve()
    {
        tried = true;
        const char* on = getenv("MEDPY_HIP_ROCTX");
        if (!on || !*on || *on == '0') return;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!dl) continue;
            *(void**)(&push) = dlsym(dl, "roctxRangePushA");
            *(void**)(&pop) = dlsym(dl, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
static MgcRoctx g_roctx;
static inline void mgc_range_push(const char* name)
{
    if (!g_roctx.tried) g_roctx.resolve();
    if (g_roctx.push) (void)g_roctx.push(name);
}
static inline void mgc_range_pop()
{
    if (g_roctx.pop) (void)g_roctx.pop();
}
struct MgcRange { /* scope guard */
    explicit MgcRange(const char* name) { mgc_range_push(name); }
    ~MgcRange() { mgc_range_pop(); }
};
static std::string g_create_error;

struct mgc_graph {
    int device = 0;
    int ndim = 0;
    int64_t shape[3] = {1, 1, 1}; /* padded to 3-D with leading 1s */
    int64_t nvox = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_pool; /* per-launch timing */
    /* launches issued one by one through mgc_solver_op (a schedule driven from outside, medpy_amd/slab.py): an event pair around
     * every discharge / relabel launch since the last mgc_build, resolved by mgc_finish into the same mgc_stats fields the
     * library's own schedules fill */
    struct OpSpan { int a, b, kind; }; /* kind 0: k_discharge_w, 1: relabel pass, 3: k_discharge (short list) */
    std::vector<hipEvent_t> op_ev;
    std::vector<OpSpan> op_spans;
    MgcLattice L{};
    /* inputs resident in HBM */
    void* d_image = nullptr; int img_dtype = 0; int term = MGC_TERM_NONE; double sigma = 0; double spacing[3] = {1, 1, 1};
    int has_spacing = 0;
    void* d_prob = nullptr; int prob_dtype = 0; double alpha = 0;
    void* d_lut = nullptr; int lut_n = 0; /* mgc_set_boundary_lut: the boundary function by table (doubles) */
    uint8_t* d_fg = nullptr; uint8_t* d_bg = nullptr;
    double* d_tr_in = nullptr; double flow_const_in = 0;
    /* pending explicit edges (host copy kept until build) */
    MgcBuildArgs build_args{}; /* of the last mgc_build: the built capacities are re-evaluated from it (mgc_built_capacity) */
    /* explicit lattice edges (plug-in path), kept until the next build: 2n arc contributions sorted by arc slot, call order
       preserved inside a slot (Graph::sum_edge adds repeated edges in call order, graph.h:457-480) */
    int64_t n_edges = 0, n_runs = 0; int64_t* d_eslot = nullptr; double* d_eval = nullptr; int64_t* d_erun = nullptr;
    bool range_set = false; double range[3] = {0, 0, 0}; /* global min / max / max|.| of the image (mgc_set_image_range) */
    bool edges_applied = false; /* the stored batch went into the last build (a new mgc_add_edges replaces it) */
    std::map<const void*, size_t> buf_cap; /* capacity of the buffers mgc_upload manages, keyed by the owning field */
    /* outputs / scratch */
    uint8_t* d_tsum = nullptr;   /* per tile: on which side of the cut its voxels lie (k_labels8) */
    uint8_t* d_tflags = nullptr; /* per tile: which signs of t-link k_build saw (MgcBuildArgs::tflags) */
    double* d_tr0 = nullptr; double* d_part = nullptr; double* d_part2 = nullptr; double* d_scalar = nullptr; uint8_t* d_labels = nullptr;
    int32_t* h_count = nullptr; /* pinned */
    double* h_scalar = nullptr; /* pinned */
    uint8_t* h_labels = nullptr; bool labels_on_host = false;
    bool built = false, solved = false;
    bool labels_valid = false; /* the distance labels belong to this build (set by the first label fill of a solve, cleared by mgc_build) */
    void* d_vout = nullptr;    /* MgcValidateOut of mgc_validate */
    uint16_t* d_dt16 = nullptr; /* scratch of the distance-transform relabel (uint16 per voxel, tile-major), allocated on first use */
    uint16_t* d_ds16 = nullptr; /* radial labels: 1 + L1 distance from the nearest voxel that held excess when the solve began (allocated on first use) */
    int32_t* d_hexact = nullptr; /* radial labels: the exact labels of the last global relabel, kept aside while the labels in L.height are the radial ones */
    bool radial_on = false;      /* the discharges run on radial labels (every saturation marks the tile DIRTY) */
    int radial_cycle_no = 0;     /* radial cycles of the current solve so far: from the second on, flow taken in marks a tile DIRTY too (MGCW_INFLOW_DIRTY) */
    bool all_residual = false; /* k_build found every n-link inside the volume residual */
    int exact_sink_tiles = 1;  /* k_discharge_w: exact in-tile labels per visit for the tiles that hold a sink link (MGCW_BFS_SINK; parameter
                                  exact_sink_tiles): 0 never, 2 always, 1 when most tiles of the volume hold one (markers scattered over the
                                  volume: 512^3 tie-heavy volume 1081 -> 790 ms; sink links only on the faces, as in the headline volume: the
                                  few visits of those tiles cost 0.7 ms of 36 more with it) */
    int sink_tiles = 0;        /* tiles holding a sink link, as built */
    int wall_tiles = 0;        /* tiles a surface of weak arcs passes through, as built (MGC_WALL_*) */
    int radial_min_walls = 16; /* parameter radial = 2 (the default): radial labels only when the graph holds at least this many wall tiles */
    int sink_sweeps = 8;       /* sweep budget of a visit while exact_sink_tiles = 1 has switched the exact labelling on (parameter sink_sweeps) */
    int use_bricks = 0;        /* incremental global relabels run their passes over bricks of 2 x 2 x 2 tiles (parameter relabel_bricks).  Measured on MI355X
                                  at 512^3: a third fewer passes (318 -> 202 launches) but 67 us instead of 31 us per pass -- 118 VGPRs allow two
                                  workgroups per CU, and a brick relaxes eight times the voxels over twice the rounds: 40.6 vs 36.3 ms per step,
                                  hard 85 vs 69 ms.  Off. */
    bool brick_mode = false;   /* ... the relabel in progress does */
    int use_dt = 1;            /* first global relabel as a distance transform when all_residual (parameter first_relabel_dt) */
    int rank = 0, nranks = 1;
    int64_t plane0 = 0, plane1 = 0, own0 = 0, own1 = 0; /* global plane ranges of a slab */
    int64_t gd0 = 0;                  /* planes of the WHOLE volume (a slab: of the volume it was cut from) */
    uint16_t* d_carry[2] = {nullptr, nullptr};    /* slabs: the plane this slab's upward / downward z-scan of a distance transform hands to the next slab */
    uint16_t* d_carry_in[2] = {nullptr, nullptr}; /* ... and where the plane of the slab before arrives when it lives in another process */
    uint16_t* dt_cur = nullptr;       /* scratch array of the transform in progress (d_dt16 / d_ds16) */
    void* d_halo = nullptr; int64_t halo_cap = 0;
    ncclComm_t comm = nullptr; void* d_xchg[4] = {nullptr, nullptr, nullptr, nullptr}; int64_t xchg_cap = 0; /* send lo, recv lo, send hi, recv hi */
    int64_t* d_cnt64 = nullptr;
    double flow_const = 0.0, flow = 0.0;
    MgcSolveParams params = mgc_default_params();
    int grid_cap = 4096;
    int grid26_dis = 16384;        /* workgroups of a k26_discharge launch (0: grid_cap): two tiles per workgroup at 512^3, 62.0 ms of discharges per config-3 step vs 63.0 at 4096 and 65.7 at 512 */
    int wave_kernels = 9;  /* bit0: region discharge, bit1: global-relabel passes run one wave per tile (mgc_wave_ops.inl);
                              bit2: the wave discharge starts from exact in-tile labels (MGCW_BFS); bit3: relabel passes with
                              MGC_RELABEL_V voxels per thread (k_relabel_v) instead of one (k_relabel_list) */
    int wave_grid_dis = 0, wave_grid_rel = 0; /* persistent grids of the wave kernels (waves resident on the device) */
    int tk_dis = MGC_CNT_TICKET_DIS, tk_rel = MGC_CNT_TICKET_REL; /* ticket slot of the next wave launch (alternates) */
    int est_phase_tiles = 1 << 30; /* length of the discharge lists at the last counter read-back */
    int sweeps_sparse26 = 5;       /* 26-neighbourhood: sweep budget of a discharge while fewer than 20 % of a colour's tiles are active (8 until incremental relabels got cheap; 512^3 markers only 285 ms at 8, 265 at 5 and 4, 287 at 3: profiles/r4_sched26_sparse_sweeps.jsonl) */
    int wave_grid26 = 0;           /* persistent grid of k26_discharge_w: one wave per SIMD (it needs the whole register file) */
    bool rounds_set = false, sparse26_set = false, wave_set = false;
    bool w26_auto = false; /* (during a solve) k26_discharge_w chosen by mgc_maxflow for a pre-pushed graph */ /* the caller chose rounds_per_relabel / sweeps_sparse26 (mgc_set_param): no automatic choice */
    int prepush = 1;               /* k_build (26-neighbourhood): settle source -> u -> v -> sink paths inside a tile while its weights are in registers (parameter prepush) */
    int w26_passes = 2, w26_raises = 1, w26_flags = 0; /* k26_discharge_w: passes over the steps / relabel rounds per sweep, MGCW26_* flags */
    int activate_exact_max = 4096; /* activation looks at the voxels of its candidate tiles only when there are at most this many (mgcw_activate_tile) */
    int wave_stagger = 0;          /* development knob of k_discharge_w (see there) */
    /* k_discharge_w<MGCW_REPEAT_MAX> (in-plane push steps repeated within a sweep, mgc_wave_ops.inl) -- parameter repeat_steps: bit 0 = for
     * graphs solved on exact labels throughout (no walls: weak contrast, integer-valued images, markers everywhere), bit 1 = also for graphs
     * whose flood runs on radial labels.  Measured on MI355X (profiles/r6_ab_repeat_policy.jsonl): 512^3 ct 73.4 -> 52.1 ms, hard 63.4 -> 59.7,
     * ties 706 -> 666; the headline volume 18.2 -> 18.0 at 512^3 but 4.40 -> 4.83 ms at 256^3 (every pushing direction pays a second vote,
     * and there the launches are a visit or two deep): on for the first kind, off for the second.  Bit 2 = graphs of the second kind ONCE THEIR
     * FLOOD IS OVER (the rounds on exact labels behind it are thin flows again: the last holes, the leak), in volumes of at least
     * repeat_flood_min_tiles tiles.  Measured and left OFF (profiles/r6_ab_repeat_after_flood*.jsonl): one object in the volume gains -- 512^3
     * 18.05 -> 17.85 ms, 640^3 34.8 -> 34.0 -- but 384^3 and 256^3 lose 1 - 8 %, and the 2048 x 1024 x 1024 volume of sixteen objects loses 6 %
     * on one handle (549 -> 585 ms) and 4 % on eight slabs */
    int repeat_steps = 1;
    int repeat_min_tiles = 0;      /* ... only for launches of at least this many tiles (parameter repeat_min_tiles) */
    int repeat_flood_min_tiles = 200000; /* ... (bit 2) tiles of the volume: 512^3 = 262 144, 384^3 = 110 592 */
    bool repeat_now = false;       /* (during a solve) */
    int wave_min_tiles = 512;      /* shorter lists are discharged by the workgroup-per-tile kernel (measured: 128^3 4.8 -> 3.4 ms, 256^3 10.8 -> 10.5 ms,
                                      512^3 unchanged; 1024 costs 512^3 8 % more discharges) */
    uint32_t zero_mask = 0; /* counters to clear before the next launch (HipDevT::flush_zero) */
    int filt[2] = {0, 0};   /* which slot of a filter's pair is in use next (HipDevT::fslot) */
    int pending_zero = -1; /* list counter the schedule asked to clear right after a discharge: the next discharge kernel clears
                              it (it neither reads nor appends to that list), any other operation flushes it with a memset first */
    int use_filters = 3; /* bit0 absorb, bit1 activate, bit2 reset-suspect go through the tile-level filter.  Bit2 is off:
                            measured on MI355X it doubles the number of global relabels (cause not understood yet) */
    mgc_stats stats{};
    int64_t device_bytes = 0;
    std::string err;
    bool timing = true;
    int timing_stride = 7; /* HIP event pairs around every n-th solver launch of a kind (see HipDevT::time_begin).  3 until round 5: with ~140 solver launches
                              per 512^3 step left, a pair around every third cost 0.65 ms of 19.6 (kernel_timing 0: 18.9 ms, stride 7: 19.0) */
    uint32_t timing_offset = 0; /* which residue of the stride is timed rotates from solve to solve: the launches of a solve have a shape (the flood grows, the
                                   leak shrinks), and a fixed residue samples the same few launches every time -- at stride 7 that read 8 % low on the headline
                                   volume; over seven steps every launch is timed once */
};

static int mgc_fail(mgc_handle h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

#define MGC_HIP(h, call)                                                                                       \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return mgc_fail(h, e_ == hipErrorOutOfMemory ? MGC_ERR_OOM : MGC_ERR_HIP, "%s failed: %s (%s:%d)", \
                            #call, hipGetErrorString(e_), __FILE__, __LINE__);                                 \
    } while (0)


/* ---- device memory of a handle comes from a small per-device POOL ----
 * A MedPy user calls graph_from_voxels once per volume (bin/medpy_graphcut_voxel.py:163-182), and the reference allocates per graph
 * (graph.cpp:12-31).  Here a 512^3 handle is 13 GB: hipMalloc maps it page by page and hipFree unmaps it -- 0.5 - 0.8 s per volume on
 * some boxes of the pool (BENCH_r05: api_end_to_end 57 ... 845 ms, every other call slow), thirty times the solve.  Blocks of at
 * least MGC_POOL_MIN bytes that a handle gives back are therefore kept (per device, by exact size: the next handle of the same shape
 * asks for exactly these sizes) up to a byte budget -- MEDPY_HIP_POOL_MB, default an eighth of the device's memory -- and handed
 * out again without a trip to the driver.  What does not fit the budget is freed as before; an allocation that fails empties the
 * pool and tries again, so the pool never turns a volume that fits the device into MGC_ERR_OOM.  mgc_pool_trim() empties it. */
#define MGC_POOL_MIN ((size_t)1 << 20)
struct MgcPool {
    std::mutex lock;
    std::multimap<size_t, void*> idle[16];  /* per device: size -> block */
    std::map<void*, std::pair<size_t, int>> live; /* blocks handed out through mgc_dmalloc: size, device */
    size_t idle_bytes[16] = {0};
    long long budget[16];
    long long hits = 0, misses = 0;
    MgcPool() { for (auto& b : budget) b = -1; }
    long long budget_of(int dev)
    {
        if (budget[dev] < 0) {
            const char* e = getenv("MEDPY_HIP_POOL_MB");
            if (e && *e) budget[dev] = atoll(e) << 20;
            else {
                size_t f = 0, t = 0;
                budget[dev] = hipMemGetInfo(&f, &t) == hipSuccess ? (long long)(t / 8) : 0;
            }
        }
        return budget[dev];
    }
    void trim(int dev) /* (lock held) */
    {
        for (auto& kv : idle[dev]) (void)hipFree(kv.second);
        idle[dev].clear();
        idle_bytes[dev] = 0;
    }
};
static MgcPool g_pool;

static hipError_t mgc_dmalloc(void** p, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    std::lock_guard<std::mutex> guard(g_pool.lock);
    if (bytes >= MGC_POOL_MIN) {
        auto it = g_pool.idle[dev].find(bytes);
        if (it != g_pool.idle[dev].end()) {
            *p = it->second;
            g_pool.idle[dev].erase(it);
            g_pool.idle_bytes[dev] -= bytes;
            g_pool.live[*p] = {bytes, dev};
            g_pool.hits++;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && g_pool.idle_bytes[dev]) { /* the pool gives way before the caller sees an error */
        (void)hipGetLastError();
        g_pool.trim(dev);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) { g_pool.live[*p] = {bytes, dev}; g_pool.misses++; }
    return e;
}

static hipError_t mgc_dfree(void* p)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> guard(g_pool.lock);
    auto it = g_pool.live.find(p);
    if (it == g_pool.live.end()) return hipFree(p);
    const size_t bytes = it->second.first;
    const int dev = it->second.second;
    g_pool.live.erase(it);
    if (bytes >= MGC_POOL_MIN && (long long)(g_pool.idle_bytes[dev] + bytes) <= g_pool.budget_of(dev)) {
        /* hipFree waits for the device; a block that goes back into the pool must be just as quiet before another handle (another
         * stream) is given it */
        (void)hipDeviceSynchronize();
        g_pool.idle[dev].insert({bytes, p});
        g_pool.idle_bytes[dev] += bytes;
        return hipSuccess;
    }
    return hipFree(p);
}

template <class T>
static int mgc_alloc(mgc_handle h, T** p, int64_t count)
{
    MGC_HIP(h, mgc_dmalloc((void**)p, (size_t)count * sizeof(T)));
    h->device_bytes += count * (int64_t)sizeof(T);
    return MGC_OK;
}

/* the counter block as the host sees it: slot c = its plain word + its shard words (a slot is used one way or the other) */
static void mgc_fold_counts(mgc_handle h)
{
    const int ns = h->L.nshard;
    for (int c = 0; c < MGC_NCOUNT; ++c)
        for (int sh = 0; sh < ns; ++sh) h->h_count[c] += h->h_count[MGC_NCOUNT + c * ns + sh];
}

/* device policy for mgc_solve(): one kernel launch per call, in-order on the handle's stream.
 * List lengths live on the device, so launches use a fixed persistent-style grid and never wait
 * for the host; per-kernel time comes from HIP event pairs recorded on the launch stream and
 * resolved after the solve (no synchronisation inside the timed region). */
template <bool FULL> /* FULL: 26-neighbourhood kernels */
struct HipDevT {
    mgc_handle h;
    hipError_t first_error = hipSuccess;
    float discharge_ms = 0.f, relabel_ms = 0.f, once_ms = 0.f;
    int64_t discharge_launches = 0, relabel_launches = 0, readbacks = 0;
    int last_discharged = -1; /* list consumed by the discharge launched last (see pending_zero) */
    int suspect_batch() const { return 4; } /* closure passes between two looks at the "changed" flag: a pass settles a brick (a pass is 5 - 12 us, a look a
                                               stream drain of ~25 us: 2 passes per look until round 5) */
    bool labels_inexact() const { return false; }
    void range_push(const char* name) { mgc_range_push(name); } /* roctx range around a stretch of the schedule (mgc_driver.inl) */
    void range_pop() { mgc_range_pop(); }
    struct Span { int a, b, kind; };
    std::vector<Span> spans;
    void check(hipError_t e) { if (e != hipSuccess && first_error == hipSuccess) first_error = e; }
    int grid(int64_t n) const { return (int)(n < 1 ? 1 : (n < h->grid_cap ? n : h->grid_cap)); }
    void fill_heights_inf()
    {
        flush_zero();
        h->labels_valid = true;
        check(hipMemsetAsync(h->L.height, 0x3f, (size_t)h->L.ntiles * MGC_TV * sizeof(int32_t), h->stream));
        hipLaunchKernelGGL(k_status_or, dim3((h->L.ntiles + 255) / 256), dim3(256), 0, h->stream, h->L, (uint32_t)MGC_ST_ALLINF, (uint32_t)MGC_ST_SETTLED); /* until a relabel lowers a label */
        for (int sd = 0; sd < 2; ++sd) /* the neighbour slabs fill their ghost layers too: the shadows of what they hold follow */
            if (h->L.hshadow[sd]) check(hipMemsetAsync(h->L.hshadow[sd], 0x3f, (size_t)h->L.gy * h->L.gx * MGC_TF * sizeof(int32_t), h->stream));
    }
    /* Counter clears are collected in a bit mask and go out as ONE one-thread kernel in front of the next launch that
     * is not itself a clear (a 512^3 solve asked for ~160 four-byte memsets, each a fill kernel of its own). */
    void flush_zero()
    {
        if (h->pending_zero >= 0) h->zero_mask |= 1u << h->pending_zero;
        h->pending_zero = -1;
        if (h->zero_mask) {
            hipLaunchKernelGGL(k_zero_counts, dim3(1), dim3(MGC_NCOUNT * MGC_NSHARD), 0, h->stream, h->L, h->zero_mask);
            check(hipGetLastError());
        }
        h->zero_mask = 0;
    }
    void zero_count(int i)
    {
        if (i == last_discharged) { flush_zero(); h->pending_zero = i; last_discharged = -1; return; }
        if (h->pending_zero >= 0) { h->zero_mask |= 1u << h->pending_zero; h->pending_zero = -1; }
        h->zero_mask |= 1u << i;
    }
    void read_counts(int* out)
    {
        flush_zero();
        check(hipMemcpyAsync(h->h_count, h->L.count, MGC_NCOUNT * (1 + MGC_NSHARD) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        check(hipStreamSynchronize(h->stream));
        mgc_fold_counts(h);
        memcpy(out, h->h_count, MGC_NCOUNT * sizeof(int32_t));
        readbacks++;
        if (!FULL) { /* how long are the discharge lists at the moment?  (picks the form of the discharge kernel) */
            int m = out[6] / 2; /* tiles the last activation found, two colours */
            for (int i = 0; i < 4; ++i) m = out[i] > m ? out[i] : m;
            h->est_phase_tiles = m;
        } else { /* (picks the sweep budget of a discharge) */
            int m = out[MGC26_CNT_ACTIVE] / 8; /* eight colours */
            for (int i = 0; i < MGC26_NLIST; ++i) m = out[i] > m ? out[i] : m;
            h->est_phase_tiles = m;
        }
    }
    int filter_grid() const { const int g = (h->L.ntiles + 255) / 256; return g < 1024 ? g : 1024; }
    /* the counter slot a tile filter counts into now / next time (which = 0: absorb, relabel seeding, suspect reset, cut value;
     * 1: activation).  The slot is clear: the consumer of the filter's previous list cleared it (or filter_done did). */
    int fslot(int which) const { return which == 0 ? (h->filt[0] ? MGC_CNT_FILTER_B : MGC_CNT_FILTER) : (h->filt[1] ? MGC_CNT_FILTER_ACT_B : MGC_CNT_FILTER_ACT); }
    int fnext(int which) const { return which == 0 ? (h->filt[0] ? MGC_CNT_FILTER : MGC_CNT_FILTER_B) : (h->filt[1] ? MGC_CNT_FILTER_ACT : MGC_CNT_FILTER_ACT_B); }
    void filter_done(int which, bool consumer_cleared_next)
    {
        if (!consumer_cleared_next) zero_count(fnext(which));
        h->filt[which] ^= 1;
    }
    void absorb_all()
    {
        flush_zero();
        if constexpr (FULL) return; /* no outboxes: neighbours are updated in place */
        else if (!(h->use_filters & 1)) { hipLaunchKernelGGL(k_absorb, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L); check(hipGetLastError()); }
        else {
            /* one thread per tile finds the tiles with a pending inbox; only those get a workgroup */
            flush_zero();
            hipLaunchKernelGGL(k_filter, dim3(filter_grid()), dim3(256), 0, h->stream, h->L, 0, 6, fslot(0));
            if (h->wave_kernels & 1) hipLaunchKernelGGL(k_absorb_w, dim3(grid((h->L.ntiles + 3) / 4)), dim3(256), 0, h->stream, h->L, 6, fslot(0), fnext(0));
            else hipLaunchKernelGGL(k_absorb_list, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, 6, fslot(0), fnext(0));
            filter_done(0, true);
            check(hipGetLastError());
        }
    }
    void relabel_all(uint32_t epoch, int next)
    {
        flush_zero();
        h->brick_mode = false; /* a from-scratch relabel runs over tiles */
        const int id = time_begin(1);
        if constexpr (FULL) hipLaunchKernelGGL(k26_relabel_all, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, epoch, next);
        else if (!(h->use_filters & 1)) hipLaunchKernelGGL(k_relabel_all, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, epoch, next);
        else { /* one thread per tile finds the seeds (tiles with an arc to the sink); only those get a workgroup */
            flush_zero();
            hipLaunchKernelGGL(k_filter, dim3(filter_grid()), dim3(256), 0, h->stream, h->L, 3, 6, fslot(0));
            if (h->wave_kernels & 8) { hipLaunchKernelGGL(k_relabel_v, dim3(grid(h->L.ntiles)), dim3(MGC_TV / MGC_RELABEL_V), 0, h->stream, h->L, 6, fslot(0), epoch, next, fnext(0), 1); filter_done(0, true); }
            else if (h->wave_kernels & 2) { hipLaunchKernelGGL(k_relabel_w, dim3(h->wave_grid_rel), dim3(MGCW_LANES), 0, h->stream, h->L, 6, fslot(0), epoch, next, -1, 1, h->tk_rel); h->tk_rel ^= 1; filter_done(0, false); }
            else { hipLaunchKernelGGL(k_relabel_first_list, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, 6, fslot(0), epoch, next); filter_done(0, false); }
        }
        check(hipGetLastError());
        time_end(id);
        relabel_launches++;
    }
    /* the first global relabel of a solve as a distance transform: labels, label supports and ALLINF flags as the relabel
     * passes would leave them (mgc_dt_ops.inl).  false: not applicable to this graph, run the passes. */
    bool first_relabel_dt()
    {
        /* only on a graph as built: a solve that runs again on its own residual graph (after MGC_ERR_NOT_CONVERGED, without a rebuild) has
         * saturated arcs, and all_residual is what k_build counted (labels_valid: some solve of this build filled the labels already) */
        if (FULL || !h->use_dt || !h->all_residual || h->labels_valid || h->nranks > 1) return false; /* (slabs: MgcSlabGroup::first_relabel_dt carries the scans across the borders) */
        if (!h->d_dt16) {
            if (mgc_dmalloc((void**)&h->d_dt16, (size_t)h->L.ntiles * MGC_TV * sizeof(uint16_t)) != hipSuccess) { (void)hipGetLastError(); return false; }
            h->device_bytes += (int64_t)h->L.ntiles * MGC_TV * (int64_t)sizeof(uint16_t);
        }
        flush_zero();
        h->labels_valid = true;
        const int id = time_begin(2);
        const MgcLattice& L = h->L;
        void* const T = h->d_dt16;
        const dim3 blk(256);
        auto g = [&](int lines) { return dim3(grid((lines + 3) / 4)); };
        hipLaunchKernelGGL((k_dt_scan<0, false, 1, 0>), g(L.gz * L.gy), blk, 0, h->stream, L, (const void*)L.rmask, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<0, true, 0, 0>), g(L.gz * L.gy), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<1, false, 0, 0>), g(L.gz * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<1, true, 0, 0>), g(L.gz * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<2, false, 0, 0>), g(L.gy * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<2, true, 0, 1>), g(L.gy * L.gx), blk, 0, h->stream, L, (const void*)T, (void*)L.height, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL(k_dt_finish, g(L.ntiles), blk, 0, h->stream, L);
        check(hipGetLastError());
        time_end(id);
        relabel_launches += 7;
        return true;
    }
    /* ---- radial labels of the flood phase (mgc_dt_ops.inl; the schedule: mgc_driver.inl) ---- */
    void set_radial(bool on) { h->radial_on = on; if (on) h->radial_cycle_no++; } /* (radial_cycle_no: reset by whoever starts a solve) */
    bool radial_begin(int c_min)
    {
        if (FULL) return false;
        const size_t nv = (size_t)h->L.ntiles * MGC_TV;
        if (!h->d_ds16) {
            if (mgc_dmalloc((void**)&h->d_ds16, nv * sizeof(uint16_t)) != hipSuccess) { (void)hipGetLastError(); return false; }
            h->device_bytes += (int64_t)(nv * sizeof(uint16_t));
        }
        if (!h->d_hexact) {
            if (mgc_dmalloc((void**)&h->d_hexact, nv * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); return false; }
            h->device_bytes += (int64_t)(nv * sizeof(int32_t));
        }
        flush_zero();
        const int id = time_begin(2);
        const MgcLattice& L = h->L;
        void* const T = h->d_ds16;
        const dim3 blk(256);
        auto g = [&](int lines) { return dim3(grid((lines + 3) / 4)); };
        check(hipMemsetAsync(L.count + mgc_cnt_radial_c(L), 0x3f, sizeof(int32_t), h->stream)); /* MGC_HINF */
        hipLaunchKernelGGL(k_dt_cmin, g(L.ntiles), blk, 0, h->stream, L); /* C from the exact labels of the source voxels */
        hipLaunchKernelGGL((k_dt_scan<0, false, 2, 0>), g(L.gz * L.gy), blk, 0, h->stream, L, (const void*)L.excess, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<0, true, 0, 0>), g(L.gz * L.gy), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<1, false, 0, 0>), g(L.gz * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<1, true, 0, 0>), g(L.gz * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<2, false, 0, 0>), g(L.gy * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        /* ... and the labels lowered on the way, INTO THE OTHER ARRAY: the last scan reads the exact labels anyway, so it writes every label
         * (lowered or not) to the array that keeps the exact ones aside and the two trade places -- no copy of 4 bytes per voxel to keep them
         * aside here (0.2 ms at 512^3), none to bring them back (radial_restore_exact) */
        hipLaunchKernelGGL((k_dt_scan<2, true, 0, 2>), g(L.gy * L.gx), blk, 0, h->stream, L, (const void*)T, T, c_min, h->d_hexact, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        std::swap(h->L.height, h->d_hexact); /* (kernels take the lattice by value at launch: everything from here on sees the lowered labels) */
        check(hipGetLastError());
        time_end(id);
        relabel_launches += 8;
        return true;
    }
    void radial_save_exact()
    {
        flush_zero();
        check(hipMemcpyAsync(h->d_hexact, h->L.height, (size_t)h->L.ntiles * MGC_TV * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    }
    void radial_restore_exact()
    {
        flush_zero();
        std::swap(h->L.height, h->d_hexact); /* what the flood made of the radial labels is dropped as a whole: the arrays trade places again */
    }
    void radial_lower(int c_min)
    {
        flush_zero();
        const int id = time_begin(2);
        hipLaunchKernelGGL(k_dt_lower, dim3(grid((h->L.ntiles + 3) / 4)), dim3(256), 0, h->stream, h->L, (const uint16_t*)h->d_ds16, c_min);
        check(hipGetLastError());
        time_end(id);
        relabel_launches++;
    }
    void source_open()
    {
        flush_zero();
        hipLaunchKernelGGL(k_source_open, dim3(grid((h->L.ntiles + 3) / 4)), dim3(256), 0, h->stream, h->L);
        check(hipGetLastError());
    }
    void relabel_list(int lst, uint32_t epoch, int next, int zero_list = -1)
    {
        flush_zero();
        const int id = time_begin(1);
        if (!FULL && h->brick_mode) hipLaunchKernelGGL(k_relabel_b, dim3(grid(mgc_brick_count(h->L))), dim3(MGC_TV), 0, h->stream, h->L, lst, epoch, next, zero_list);
        else if constexpr (FULL) hipLaunchKernelGGL(k26_relabel_list, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, lst, epoch, next);
        else if (h->wave_kernels & 8) hipLaunchKernelGGL(k_relabel_v, dim3(grid(h->L.ntiles)), dim3(MGC_TV / MGC_RELABEL_V), 0, h->stream, h->L, lst, lst, epoch, next, zero_list, 0);
        else if (h->wave_kernels & 2) { hipLaunchKernelGGL(k_relabel_w, dim3(h->wave_grid_rel), dim3(MGCW_LANES), 0, h->stream, h->L, lst, lst, epoch, next, zero_list, 0, h->tk_rel); h->tk_rel ^= 1; }
        else hipLaunchKernelGGL(k_relabel_list, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, lst, epoch, next, zero_list);
        check(hipGetLastError());
        time_end(id);
        relabel_launches++;
    }
    void suspect_pass()
    {
        flush_zero();
        const int nb = ((h->L.gx + 7) / 8) * ((h->L.gy + 7) / 8) * ((h->L.gz + 7) / 8);
        hipLaunchKernelGGL(k_suspect_pass<FULL>, dim3(nb < 4096 ? nb : 4096), dim3(MGC_TV), 0, h->stream, h->L);
        check(hipGetLastError());
    }
    void reset_suspect(uint32_t epoch, int list)
    {
        flush_zero();
        const int id = time_begin(1);
        /* the passes of an incremental relabel run over bricks of 2 x 2 x 2 tiles (mgc_brick_ops.inl): single handle, 6-neighbourhood */
        h->brick_mode = !FULL && h->use_bricks && h->nranks == 1 && !(h->use_filters & 4);
        if (FULL || !(h->use_filters & 4)) hipLaunchKernelGGL(k_reset_suspect<FULL>, dim3(grid((h->L.ntiles + MGC_TV - 1) / MGC_TV)), dim3(MGC_TV), 0, h->stream, h->L, epoch, list, h->brick_mode ? 1 : 0);
        else {
            flush_zero();
            hipLaunchKernelGGL(k_filter, dim3(filter_grid()), dim3(256), 0, h->stream, h->L, 2, 6, fslot(0));
            hipLaunchKernelGGL(k_reset_suspect_list, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, 6, fslot(0), epoch, list);
            filter_done(0, false);
        }
        check(hipGetLastError());
        time_end(id);
        relabel_launches++;
    }
    void activate_all(uint32_t phase)
    {
        flush_zero();
        if constexpr (FULL) {
            if (h->L.nshard == 1 && (h->use_filters & 2)) hipLaunchKernelGGL(k26_activate_w, dim3(grid((h->L.ntiles + 3) / 4)), dim3(256), 0, h->stream, h->L, phase);
            else hipLaunchKernelGGL(k26_activate, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, phase);
        }
        else if (!(h->use_filters & 2)) hipLaunchKernelGGL(k_activate, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, phase);
        else {
            /* only tiles whose status says "holds excess" are examined voxel by voxel */
            flush_zero();
            hipLaunchKernelGGL(k_filter, dim3(filter_grid()), dim3(256), 0, h->stream, h->L, 1, 7, fslot(1));
            if (h->wave_kernels & 1) { hipLaunchKernelGGL(k_activate_w, dim3(grid((h->L.ntiles + 3) / 4)), dim3(256), 0, h->stream, h->L, 7, fslot(1), phase, h->activate_exact_max, fnext(1)); filter_done(1, true); }
            else { hipLaunchKernelGGL(k_activate_list, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, 7, fslot(1), phase); filter_done(1, false); }
        }
        check(hipGetLastError());
    }
    void discharge(int lst, uint32_t phase, int cycles, int sweeps)
    {
        const int zero_idx = h->pending_zero; /* cleared inside the kernel: nothing sits between two colour phases */
        h->pending_zero = -1;
        flush_zero(); /* (whatever else is waiting to be cleared) */
        const bool wave_form = !FULL && (h->wave_kernels & 1) && h->est_phase_tiles >= h->wave_min_tiles;
        const int id = time_begin(FULL || wave_form ? 0 : 3);
        if constexpr (FULL) {
            /* Busy phases (a regional term: 40 % of the tiles hold excess) are paced by their heaviest tiles, and excess cannot
             * leave a tile before its neighbours run: three sweeps per visit.  Sparse phases (a front of active tiles) are
             * paced by launches and relabels: let a tile work longer.  Measured at 512^3: config 3 102.8 ms at 3 sweeps, 129.8 at
             * 6; markers only 494 ms at 3, 431 at 8. */
            if (h->sweeps_sparse26 > 0 && (int64_t)h->est_phase_tiles * 40 < h->L.ntiles) sweeps = h->sweeps_sparse26;
            if (((h->wave_kernels & 32) || h->w26_auto) && cycles < 0) { /* one wave per tile, the tile in registers (stored labels only) */
                hipLaunchKernelGGL(k26_discharge_w, dim3(h->wave_grid26), dim3(MGCW_LANES), 0, h->stream, h->L, lst, phase, sweeps, h->w26_auto ? 1 : h->w26_passes, h->w26_raises, h->w26_flags, h->tk_dis, zero_idx);
                h->tk_dis ^= 1;
            }
            else if (h->wave_kernels & 16) hipLaunchKernelGGL(k26_discharge_v, dim3(grid(h->L.ntiles)), dim3(MGC_TV / 2), 0, h->stream, h->L, lst, phase, cycles, sweeps, zero_idx);
            else hipLaunchKernelGGL(k26_discharge, dim3(h->grid26_dis > 0 ? (h->grid26_dis < h->L.ntiles ? h->grid26_dis : h->L.ntiles) : grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, lst, phase, cycles, sweeps, zero_idx);
        }
        /* one wave per tile has the higher throughput (2 048 tiles in flight, fewer instructions per tile), eight waves per tile
         * the shorter latency (36 us against 60 us for one tile): short lists -- small volumes, the tail of a solve -- are a
         * single tile deep per launch and go to the workgroup form.  Both forms keep the same state in HBM. */
        else if (wave_form) {
            const bool exact_sink = h->exact_sink_tiles == 2 || (h->exact_sink_tiles == 1 && 2 * (int64_t)h->sink_tiles > h->L.ntiles);
            /* a visit that starts from exact in-tile labels needs fewer sweeps to move what it can (tie-heavy 512^3: 796 ms at 12, 731 at 8, 788 at 6) */
            if (exact_sink && h->exact_sink_tiles == 1 && sweeps > h->sink_sweeps) sweeps = h->sink_sweeps;
            const int dflags = ((h->wave_kernels & 4) ? MGCW_BFS : 0) | (exact_sink ? MGCW_BFS_SINK : 0) | (h->radial_on ? MGCW_SAT_DIRTY : 0) | ((h->radial_on && h->radial_cycle_no > 1) ? MGCW_INFLOW_DIRTY : 0);
            /* repeated in-plane steps (mgcw_discharge_impl<.., MGCW_REPEAT_MAX>): decided per SOLVE (repeat_now, mgc_maxflow / mgc_solve_slabs) */
            const bool rep = (h->repeat_now && h->est_phase_tiles >= h->repeat_min_tiles) ||
                             ((h->repeat_steps & 4) && h->radial_cycle_no > 0 && !h->radial_on && h->L.ntiles >= h->repeat_flood_min_tiles); /* (bit 2: the flood is over) */
            if (rep) hipLaunchKernelGGL(k_discharge_w<MGCW_REPEAT_MAX>, dim3(h->wave_grid_dis), dim3(MGCW_LANES), 0, h->stream, h->L, lst, phase, sweeps, dflags, h->tk_dis, zero_idx, h->wave_stagger);
            else hipLaunchKernelGGL(k_discharge_w<1>, dim3(h->wave_grid_dis), dim3(MGCW_LANES), 0, h->stream, h->L, lst, phase, sweeps, dflags, h->tk_dis, zero_idx, h->wave_stagger);
            h->tk_dis ^= 1;
        }
        else hipLaunchKernelGGL(k_discharge, dim3(grid(h->L.ntiles)), dim3(MGC_TV), 0, h->stream, h->L, lst, phase,
                                (h->wave_kernels & 1) && !(h->wave_kernels & 4) ? (h->radial_on ? (h->radial_cycle_no > 1 ? -3 : -2) : -1) : cycles, sweeps, zero_idx); /* same labelling policy as the wave form (-2: radial labels, any saturation marks the tile) */
        check(hipGetLastError());
        time_end(id);
        discharge_launches++;
        last_discharged = lst;
    }
    /* ---- one slab of a volume, as MgcSlabGroup / MgcXchg (mgc_driver.inl) drive it ---- */
    bool multi() const { return false; }
    void exchange(int, uint32_t, int) {}
    const MgcLattice& lattice() const { return h->L; }
    bool radial_after_passes() const { return FULL && h->nranks == 1; } /* the full neighbourhood: radial labels on top of a first relabel by passes (single handles) */
    bool has_lower() const { return h->L.tz_own_lo > 0; }
    bool has_upper() const { return h->L.tz_own_hi < h->L.gz; }
    bool needs_carry(int dir) const { return dir == 0 ? h->plane0 > 0 : h->plane1 < h->gd0; }
    bool sends_carry(int dir) const { return dir == 0 ? (has_upper() && h->own1 - 9 >= h->plane0) : (has_lower() && h->own0 + 8 < h->gd0); }
    int carry_plane(int dir) const { return (int)(dir == 0 ? h->own1 - 9 - h->plane0 : h->own0 + 8 - h->plane0); }
    int64_t carry_bytes() const { return (int64_t)h->L.dy * h->L.dx * (int64_t)sizeof(uint16_t); }
    uint16_t* carry_buf(int dir) { return h->d_carry[dir]; }
    uint16_t* carry_recv_buf(int dir) { return h->d_carry_in[dir]; }
    bool ensure(void** p, size_t bytes)
    {
        if (*p) return true;
        if (mgc_dmalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
        h->device_bytes += (int64_t)bytes;
        return true;
    }
    bool dt_applicable()
    {
        if (FULL || !h->use_dt || !h->all_residual || h->labels_valid) return false;
        bool ok = ensure((void**)&h->d_dt16, (size_t)h->L.ntiles * MGC_TV * sizeof(uint16_t));
        for (int k = 0; k < 2 && ok; ++k) ok = ensure((void**)&h->d_carry[k], (size_t)carry_bytes()) && ensure((void**)&h->d_carry_in[k], (size_t)carry_bytes());
        return ok;
    }
    dim3 dt_grid(int lines) const { return dim3(grid((lines + 3) / 4)); }
    int dt_span = -1;
    void dt_scans_xy(int seed)
    {
        flush_zero();
        h->labels_valid = true;
        dt_span = time_begin(2);
        const MgcLattice& L = h->L;
        void* const T = h->dt_cur = seed == 1 ? h->d_dt16 : h->d_ds16;
        const dim3 blk(256);
        if (seed == 1) hipLaunchKernelGGL((k_dt_scan<0, false, 1, 0>), dt_grid(L.gz * L.gy), blk, 0, h->stream, L, (const void*)L.rmask, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        else hipLaunchKernelGGL((k_dt_scan<0, false, 2, 0>), dt_grid(L.gz * L.gy), blk, 0, h->stream, L, (const void*)L.excess, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<0, true, 0, 0>), dt_grid(L.gz * L.gy), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<1, false, 0, 0>), dt_grid(L.gz * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        hipLaunchKernelGGL((k_dt_scan<1, true, 0, 0>), dt_grid(L.gz * L.gx), blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)nullptr, -1);
        check(hipGetLastError());
        relabel_launches += 4;
    }
    /* final_kind 0: an intermediate scan; 1: the scan that writes the labels (transform towards the sink); 2: the scan that lowers the
     * labels into the other array (transform away from the source) */
    void dt_scan_z(bool bwd, int final_kind, int c_min, const uint16_t* cin, bool want_out)
    {
        const MgcLattice& L = h->L;
        void* const T = h->dt_cur;
        const dim3 blk(256), g = dt_grid(L.gy * L.gx);
        uint16_t* const cout = want_out ? h->d_carry[bwd ? 1 : 0] : nullptr;
        const int plane = want_out ? carry_plane(bwd ? 1 : 0) : -1;
        if (!bwd) hipLaunchKernelGGL((k_dt_scan<2, false, 0, 0>), g, blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, cin, cout, plane);
        else if (final_kind == 1) hipLaunchKernelGGL((k_dt_scan<2, true, 0, 1>), g, blk, 0, h->stream, L, (const void*)T, (void*)L.height, 0, (int32_t*)nullptr, cin, cout, plane);
        else if (final_kind == 2) hipLaunchKernelGGL((k_dt_scan<2, true, 0, 2>), g, blk, 0, h->stream, L, (const void*)T, T, c_min, h->d_hexact, cin, cout, plane);
        else hipLaunchKernelGGL((k_dt_scan<2, true, 0, 0>), g, blk, 0, h->stream, L, (const void*)T, T, 0, (int32_t*)nullptr, cin, cout, plane);
        check(hipGetLastError());
        relabel_launches++;
        if (bwd) { time_end(dt_span); dt_span = -1; }
    }
    void dt_finish()
    {
        hipLaunchKernelGGL(k_dt_finish, dt_grid(h->L.ntiles), dim3(256), 0, h->stream, h->L);
        check(hipGetLastError());
        relabel_launches++;
    }
    void shadow_sync()
    {
        if (FULL || (!has_lower() && !has_upper())) return;
        hipLaunchKernelGGL(k_shadow_sync, dt_grid(2 * h->L.gy * h->L.gx), dim3(256), 0, h->stream, h->L);
        check(hipGetLastError());
    }
    bool radial_prepare()
    {
        if (FULL) return false;
        const size_t nv = (size_t)h->L.ntiles * MGC_TV;
        return ensure((void**)&h->d_ds16, nv * sizeof(uint16_t)) && ensure((void**)&h->d_hexact, nv * sizeof(int32_t));
    }
    void radial_cmin()
    {
        flush_zero();
        check(hipMemsetAsync(h->L.count + mgc_cnt_radial_c(h->L), 0x3f, sizeof(int32_t), h->stream)); /* MGC_HINF */
        hipLaunchKernelGGL(k_dt_cmin, dt_grid(h->L.ntiles), dim3(256), 0, h->stream, h->L);
        check(hipGetLastError());
    }
    int count_get(int i)
    {
        int v = 0;
        check(hipMemcpyAsync(&v, h->L.count + i, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        check(hipStreamSynchronize(h->stream));
        readbacks++;
        return v;
    }
    void count_set(int i, int v) { check(hipMemsetD32Async((hipDeviceptr_t)(h->L.count + i), v, 1, h->stream)); }
    void radial_swap() { std::swap(h->L.height, h->d_hexact); }
    /* border messages: a fixed size per kind that both sides know without asking the device (mgc_halo_exchange) */
    int64_t halo_msg_bytes(int kind) const
    {
        return mgc_halo_compact_nd(h->L, kind) ? mgc_halo_off_rec_nd(h->L) + (int64_t)h->L.halo_max_rec * mgc_halo_rec_bytes_nd(h->L, kind) : mgc_halo_bytes_nd(h->L, kind);
    }
    bool xchg_ready()
    {
        int64_t bytes = mgc_halo_bytes_nd(h->L, 1);
        if (mgc_halo_bytes_nd(h->L, 0) > bytes) bytes = mgc_halo_bytes_nd(h->L, 0);
        if (h->xchg_cap >= bytes) return true;
        for (int i = 0; i < 4; ++i) {
            if (h->d_xchg[i]) (void)mgc_dfree(h->d_xchg[i]);
            h->d_xchg[i] = nullptr;
            if (mgc_dmalloc(&h->d_xchg[i], (size_t)bytes) != hipSuccess) { check(hipErrorOutOfMemory); return false; }
        }
        h->xchg_cap = bytes;
        return true;
    }
    void* recv_buf(int side) { return xchg_ready() ? h->d_xchg[2 * side + 1] : nullptr; }
    void* halo_pack(int side, int kind)
    {
        flush_zero();
        if (!xchg_ready()) return nullptr;
        void* const dst = h->d_xchg[2 * side];
        const int T = h->L.gy * h->L.gx;
        if (mgc_halo_compact_nd(h->L, kind)) check(hipMemsetAsync((char*)dst + mgc_halo_off_count_nd(h->L), 0, 4, h->stream));
        hipLaunchKernelGGL(k_halo_pack, dim3(T < 2048 ? T : 2048), dim3(MGC_TV), 0, h->stream, h->L, side, kind, dst);
        check(hipGetLastError());
        return dst;
    }
    void halo_unpack(int side, int kind, const void* buf, uint32_t epoch, int list)
    {
        if (!buf) return;
        flush_zero();
        const int T = h->L.gy * h->L.gx;
        hipLaunchKernelGGL(k_halo_unpack, dim3(T < 2048 ? T : 2048), dim3(MGC_TV), 0, h->stream, h->L, side, kind, buf, epoch, list);
        check(hipGetLastError());
    }
    void to_host(void* host, const void* buf, int64_t n)
    {
        check(hipMemcpyAsync(host, buf, (size_t)n, hipMemcpyDeviceToHost, h->stream));
        check(hipStreamSynchronize(h->stream));
    }
    void from_host(void* buf, const void* host, int64_t n)
    {
        check(hipMemcpyAsync(buf, host, (size_t)n, hipMemcpyHostToDevice, h->stream));
        check(hipStreamSynchronize(h->stream)); /* (the host buffer is the caller's again on return) */
    }
    bool has_comm() const { return h->comm != nullptr; }
    int native_exchange(int kind, uint32_t epoch, int list);
    int native_allreduce(int64_t* v, int n, int op);
    int native_send(int side, const void* buf, int64_t n);
    int native_recv(int side, void* buf, int64_t n);

    int64_t timed[4] = {0, 0, 0, 0}, seen[4] = {0, 0, 0, 0}; /* launches with an event pair / launches, per kind: 0 = k_discharge_w (26-neighbourhood:
                                                                 k26_discharge), 1 = relabel passes, 2 = one-off stretches, 3 = k_discharge (short lists) */
    float block_ms = 0.f;
    int time_begin(int kind)
    {
        if (!h->timing) return -1;
        /* every `timing_stride`-th launch of a kind carries a pair of HIP events (an event is a barrier packet in the queue:
         * a pair around each of the ~450 solver launches of a 512^3 step costs 3 ms of its 47); the kernel time of the
         * kind is the mean of the timed launches times the number of launches.  An odd stride samples both tile colours. */
        if (kind != 2) { /* (kind 2 = a one-off stretch, always timed, never extrapolated: the distance-transform relabel) */
            if (((seen[kind]++ + h->timing_offset) % (uint32_t)h->timing_stride) != 0) return -1;
            timed[kind]++;
        }
        const size_t need = 2 * spans.size() + 2;
        while (h->ev_pool.size() < need) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            h->ev_pool.push_back(e);
        }
        Span sp{(int)(2 * spans.size()), (int)(2 * spans.size() + 1), kind};
        check(hipEventRecord(h->ev_pool[sp.a], h->stream));
        spans.push_back(sp);
        return (int)spans.size() - 1;
    }
    void time_end(int id)
    {
        if (id >= 0) check(hipEventRecord(h->ev_pool[spans[id].b], h->stream));
    }
    void resolve_timing() /* after the stream has drained */
    {
        for (const Span& sp : spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev_pool[sp.a], h->ev_pool[sp.b]) == hipSuccess) (sp.kind == 0 ? discharge_ms : (sp.kind == 1 ? relabel_ms : (sp.kind == 2 ? once_ms : block_ms))) += ms;
        }
        if (timed[0]) discharge_ms *= (float)seen[0] / (float)timed[0];
        if (timed[1]) relabel_ms *= (float)seen[1] / (float)timed[1];
        if (timed[3]) block_ms *= (float)seen[3] / (float)timed[3];
        relabel_ms += once_ms;
    }
};
typedef HipDevT<false> HipDev;
typedef HipDevT<true> HipDev26;

/* C entry points that launch kernels or read the counters outside a HipDevT: clears that are still pending go out first */
static void mgc_flush_zero(mgc_handle h)
{
    HipDev dev;
    dev.h = h;
    dev.flush_zero();
}

/* one step of the solver schedule (mgc_solver_op), on the kernels of either neighbourhood */
template <class Dev>
static int mgc_solver_op_on(mgc_handle h, int op, int64_t a0, int64_t a1, int64_t a2, int64_t a3)
{
    Dev dev;
    dev.h = h;
    const bool timing = h->timing;
    h->timing = false; /* (the per-launch events of a HipDevT live as long as it does: here the handle keeps the pairs, see op_spans) */
    int span_kind = -1;
    if (op == MGC_OP_DISCHARGE) span_kind = (h->L.ndir != 6 || ((h->wave_kernels & 1) && h->est_phase_tiles >= h->wave_min_tiles)) ? 0 : 3;
    else if (op == MGC_OP_RELABEL_LIST || op == MGC_OP_RELABEL_ALL || op == MGC_OP_RESET_SUSPECT) span_kind = 1;
    int ev_a = -1;
    if (timing && span_kind >= 0) {
        dev.flush_zero(); /* (a pending counter clear is not part of the launch that is timed) */
        const size_t need = 2 * h->op_spans.size() + 2;
        while (h->op_ev.size() < need) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            h->op_ev.push_back(e);
        }
        if (h->op_ev.size() >= need) {
            ev_a = (int)(2 * h->op_spans.size());
            (void)hipEventRecord(h->op_ev[ev_a], h->stream);
        }
    }
    switch (op) {
    case MGC_OP_ABSORB_ALL: dev.absorb_all(); break;
    case MGC_OP_FILL_INF: dev.fill_heights_inf(); break;
    case MGC_OP_ZERO_COUNT:
        if (a0 < 0 || a0 >= MGC_NCOUNT) { h->timing = timing; return mgc_fail(h, MGC_ERR_INVALID, "counter index"); }
        dev.zero_count((int)a0);
        break;
    case MGC_OP_RELABEL_ALL: dev.relabel_all((uint32_t)a0, (int)a1); break;
    case MGC_OP_RELABEL_LIST: dev.relabel_list((int)a0, (uint32_t)a1, (int)a2); break;
    case MGC_OP_ACTIVATE: dev.activate_all((uint32_t)a0); break;
    case MGC_OP_DISCHARGE: dev.discharge((int)a0, (uint32_t)a1, (int)a2, (int)a3); break;
    case MGC_OP_SUSPECT_PASS: dev.suspect_pass(); break;
    case MGC_OP_RESET_SUSPECT: dev.reset_suspect((uint32_t)a0, (int)a1); break;
    default: h->timing = timing; return mgc_fail(h, MGC_ERR_INVALID, "unknown solver op %d", op);
    }
    if (ev_a >= 0) {
        (void)hipEventRecord(h->op_ev[ev_a + 1], h->stream);
        h->op_spans.push_back({ev_a, ev_a + 1, span_kind});
    }
    h->timing = timing;
    h->solved = false;
    if (dev.first_error != hipSuccess) return mgc_fail(h, MGC_ERR_HIP, "solver op %d: %s", op, hipGetErrorString(dev.first_error));
    return MGC_OK;
}


/* fixed-order sum of the per-tile partials in h->d_part: two stages for long vectors (one 512-lane block walking 262 144
 * doubles took 0.2 ms) */
static void mgc_sum_partials(mgc_handle h, int64_t n, double* out)
{
    if (n > 8192 && h->d_part2) {
        const int64_t len = (n + 255) / 256;
        hipLaunchKernelGGL(k_sum_chunks, dim3(256), dim3(256), 0, h->stream, (const double*)h->d_part, n, len, h->d_part2);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(MGC_TV), 0, h->stream, (const double*)h->d_part2, (int64_t)256, out);
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(MGC_TV), 0, h->stream, (const double*)h->d_part, n, out);
    }
}

/* k_cut_value26: with label summaries two launches (the t-link stream at full occupancy, then the tiles the cut passes through), without
 * them every tile takes the general path in one */
static void mgc_launch_cut_value26(mgc_handle h, int grid, bool rows8)
{
    MgcLattice& L = h->L;
    if (!rows8) {
        hipLaunchKernelGGL(k_cut_value26<0>, dim3((grid + 3) / 4 < 8192 ? (grid + 3) / 4 : 8192), dim3(256), 0, h->stream, L, h->build_args, (const double*)h->d_tr0, (const uint8_t*)h->d_labels, (const uint8_t*)nullptr, h->d_part);
        return;
    }
    const int wgs = (L.ntiles + 3) / 4;
    hipLaunchKernelGGL(k_cut_value26<1>, dim3(wgs < 16384 ? wgs : 16384), dim3(256), 0, h->stream, L, h->build_args, (const double*)h->d_tr0, (const uint8_t*)h->d_labels, (const uint8_t*)h->d_tsum, h->d_part);
    int wg2 = ((L.ntiles + 127) / 128) | 1; /* W = 4 wg2 waves look at ~32 tiles each; odd: W is no multiple of a power-of-two row or layer of tiles */
    if (wg2 > 16385) wg2 = 16385;
    hipLaunchKernelGGL(k_cut_value26<2>, dim3(wg2), dim3(256), 0, h->stream, L, h->build_args, (const double*)h->d_tr0, (const uint8_t*)h->d_labels, (const uint8_t*)h->d_tsum, h->d_part);
}

/* read-out: the label volume (C order, 0 = SINK side) and the capacity of the cut those labels define, summed into
 * h->d_scalar[slot] */
static int mgc_launch_readout(mgc_handle h, int slot, hipEvent_t after_labels)
{
    MgcRange range_("read-out: labels + cut value");
    MgcLattice& L = h->L;
    const bool rows8 = L.dx % 8 == 0;
    if (rows8) hipLaunchKernelGGL(k_labels8, dim3(2048), dim3(256), 0, h->stream, L, h->d_labels, h->d_tsum);
    else hipLaunchKernelGGL(k_labels, dim3(2048), dim3(256), 0, h->stream, L, h->d_labels);
    MGC_HIP(h, hipGetLastError());
    if (after_labels) MGC_HIP(h, hipEventRecord(after_labels, h->stream));
    const int grid = L.ntiles < h->grid_cap * 4 ? L.ntiles : h->grid_cap * 4;
    if (L.ndir == 6 && h->d_prob && h->d_tr0 && !L.cap0 && L.nshard == 1) {
        /* a regional term: every tile pays t-links (the tile filter below would list them all, and k_cut_value6 sums a tile behind ten
         * barriers); 6-neighbourhood tiles write tr0 only where they hold t-links -- with a regional term that is everywhere */
        mgc_launch_cut_value26(h, grid, rows8);
    }
    else if (L.ndir == 6) {
        const int fg = (L.ntiles + 255) / 256;
        HipDev dev;
        dev.h = h;
        dev.flush_zero();
        hipLaunchKernelGGL(k_cut_filter, dim3(fg < 1024 ? fg : 1024), dim3(256), 0, h->stream, L, h->build_args, rows8 ? (const uint8_t*)h->d_tsum : (const uint8_t*)nullptr, h->d_part, 6, dev.fslot(0));
        hipLaunchKernelGGL(k_cut_value6, dim3(grid), dim3(MGC_TV), 0, h->stream, L, h->build_args, (const double*)h->d_tr0, 6, dev.fslot(0), h->d_part, dev.fnext(0));
        dev.filter_done(0, true);
    }
    else if (L.ndir == MGC26_NDIR) mgc_launch_cut_value26(h, grid, rows8);
    else hipLaunchKernelGGL(k_cut_value, dim3(grid), dim3(MGC_TV), 0, h->stream, L, h->build_args, (const double*)h->d_tr0, (const uint8_t*)h->d_labels, h->d_part);
    MGC_HIP(h, hipGetLastError());
    mgc_sum_partials(h, (int64_t)L.ntiles, h->d_scalar + slot);
    MGC_HIP(h, hipGetLastError());
    return MGC_OK;
}

extern "C" {

int mgc_device_count(int* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (count) *count = n;
    return MGC_OK;
}

int mgc_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes)
{
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); return MGC_ERR_NO_DEVICE; }
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return MGC_OK;
}

const char* mgc_last_error(mgc_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mgc_pool_trim(int device)
{
    if (device < 0 || device > 15) return MGC_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return MGC_ERR_NO_DEVICE; }
    std::lock_guard<std::mutex> guard(g_pool.lock);
    g_pool.trim(device);
    return MGC_OK;
}

int mgc_pool_info(int device, int64_t* idle_bytes, int64_t* hits, int64_t* misses)
{
    if (device < 0 || device > 15) return MGC_ERR_INVALID;
    std::lock_guard<std::mutex> guard(g_pool.lock);
    if (idle_bytes) *idle_bytes = (int64_t)g_pool.idle_bytes[device];
    if (hits) *hits = g_pool.hits;
    if (misses) *misses = g_pool.misses;
    return MGC_OK;
}

static int mgc_create_impl(int ndim, const int64_t* shape, int connectivity, int device, const MgcSlabSpec* slab, mgc_handle* out, int64_t gd0 = 0)
{
    if (!out) return mgc_fail(nullptr, MGC_ERR_INVALID, "mgc_create: out is NULL");
    *out = nullptr;
    if (ndim < 1 || ndim > 3 || !shape) return mgc_fail(nullptr, MGC_ERR_UNSUPPORTED, "mgc_create: ndim must be 1..3 (got %d)", ndim);
    int full = 1;
    for (int k = 0; k < ndim; ++k) full *= 3;
    full -= 1; /* 2, 8, 26 */
    if (connectivity != 2 * ndim && connectivity != full)
        return mgc_fail(nullptr, MGC_ERR_UNSUPPORTED, "mgc_create: connectivity %d not implemented (2*ndim = %d is the reference's, %d the full neighbourhood)",
                        connectivity, 2 * ndim, full);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return mgc_fail(nullptr, MGC_ERR_NO_DEVICE, "no HIP device: libmedpyhip has no CPU fallback");
    if (device < 0 || device >= ndev) return mgc_fail(nullptr, MGC_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    mgc_handle h = new (std::nothrow) mgc_graph();
    if (!h) return mgc_fail(nullptr, MGC_ERR_OOM, "host allocation failed");
    h->device = device;
    h->ndim = ndim;
    int64_t n = 1;
    for (int k = 0; k < ndim; ++k) {
        if (shape[k] < 1) { delete h; return mgc_fail(nullptr, MGC_ERR_INVALID, "shape[%d] = %lld", k, (long long)shape[k]); }
        h->shape[3 - ndim + k] = shape[k];
        n *= shape[k];
    }
    h->nvox = n;
    MgcLattice& L = h->L;
    L.dz = h->shape[0]; L.dy = h->shape[1]; L.dx = h->shape[2];
    L.nvox = n;
    const int64_t gz = (L.dz + 7) / 8, gy = (L.dy + 7) / 8, gx = (L.dx + 7) / 8;
    if (gz * gy * gx > 0x3fffffff) { delete h; return mgc_fail(nullptr, MGC_ERR_UNSUPPORTED, "volume too large for 32-bit tile ids"); }
    L.gz = (int)gz; L.gy = (int)gy; L.gx = (int)gx;
    L.ntiles = (int)(gz * gy * gx);
    L.tz_own_lo = 0; L.tz_own_hi = L.gz; L.tz_global0 = 0;
    L.ndir = (connectivity == 2 * ndim) ? 6 : 26;
    h->params = mgc_default_params(L.ndir);
    /* (until round 6 the full neighbourhood timed EVERY launch -- "few, long launches": with the relabels of round 6 a config-3 step is ~100 launches of
     * 0.3 ms on average and an event pair costs 10 us: 0.6 of 32.9 ms, 7 of 234 ms without the regional term, 0.6 of 8.3 ms at 256^3.  Stride 7 like the
     * 6-neighbourhood: coprime to the eight colours, the timed residue rotates from solve to solve.) */
    if (slab) {
        L.tz_own_lo = slab->own_lo; L.tz_own_hi = slab->own_hi; L.tz_global0 = slab->tz_global0;
        h->rank = slab->rank; h->nranks = slab->nranks;
        h->plane0 = slab->plane0; h->plane1 = slab->plane1; h->own0 = slab->own0; h->own1 = slab->own1;
    } else {
        h->plane0 = h->own0 = 0; h->plane1 = h->own1 = L.dz;
    }
    h->gd0 = gd0 > 0 ? gd0 : L.dz;
    *out = h; /* from here on errors are reported through the handle; caller destroys it */
    MGC_HIP(h, hipSetDevice(device));
    MGC_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; ++i) MGC_HIP(h, hipEventCreate(&h->ev[i]));
    const int64_t nt = L.ntiles, nv = nt * MGC_TV;
    int rc;
    if ((rc = mgc_alloc(h, &L.rcap, nv * L.ndir))) return rc;
    L.cap0 = nullptr; /* only materialised when explicit edges are added (mgc_build) */
    if ((rc = mgc_alloc(h, &L.excess, nv))) return rc;
    if ((rc = mgc_alloc(h, &L.sink, nv))) return rc;
    if ((rc = mgc_alloc(h, &L.height, nv))) return rc;
    if (L.ndir == 6) {
        if ((rc = mgc_alloc(h, &L.rmask, nv))) return rc;
        if ((rc = mgc_alloc(h, &L.obox, nt * 6 * MGC_TF))) return rc;
    } else {
        if ((rc = mgc_alloc(h, &L.rmask32, nv))) return rc;
    }
    if ((rc = mgc_alloc(h, &L.oflags, nt))) return rc;
    if (slab && L.ndir == 6) { /* labels of the border layers as the neighbour slabs last received them */
        for (int sd = 0; sd < 2; ++sd) {
            if ((rc = mgc_alloc(h, &L.hshadow[sd], (int64_t)L.gy * L.gx * MGC_TF))) return rc;
            MGC_HIP(h, hipMemsetAsync(L.hshadow[sd], 0x3f, (size_t)L.gy * L.gx * MGC_TF * sizeof(int32_t), h->stream));
        }
    }
    L.nshard = 1; /* regions per work list (MgcLattice::scount): one, unless list_shards asks for MGC_NSHARD */
    { /* record slots of a compacted border message (MgcLattice::halo_max_rec): an eighth of the border tiles, at least 64 */
        const int64_t T = gy * gx;
        L.halo_max_rec = (int)(T / 8 > 64 ? T / 8 : (T < 64 ? T : 64));
    }
    L.shard_cap = (int)nt;
    for (int i = 0; i < (L.ndir == 6 ? 8 : 18); ++i)
        if ((rc = mgc_alloc(h, &L.list[i], nt * MGC_NSHARD))) return rc;
    if ((rc = mgc_alloc(h, &L.count, (int64_t)MGC_NCOUNT * (1 + MGC_NSHARD)))) return rc; /* count[MGC_NCOUNT] | scount[MGC_NCOUNT][MGC_NSHARD] */
    L.scount = L.count + MGC_NCOUNT;
    if ((rc = mgc_alloc(h, &L.stamp, nt))) return rc;
    if ((rc = mgc_alloc(h, &L.rstamp, nt))) return rc;
    if ((rc = mgc_alloc(h, &L.status, nt))) return rc;
    if ((rc = mgc_alloc(h, &h->d_tr0, nv))) return rc;
    if ((rc = mgc_alloc(h, &h->d_tflags, nt))) return rc;
    if ((rc = mgc_alloc(h, &h->d_tsum, nt))) return rc;
    if ((rc = mgc_alloc(h, &h->d_part, nt > 4096 ? nt : (int64_t)4096))) return rc;
    if ((rc = mgc_alloc(h, &h->d_part2, (int64_t)256))) return rc;
    if ((rc = mgc_alloc(h, &h->d_scalar, (int64_t)8))) return rc;
    if ((rc = mgc_alloc(h, &h->d_labels, n))) return rc;
    MGC_HIP(h, hipHostMalloc((void**)&h->h_count, MGC_NCOUNT * (1 + MGC_NSHARD) * sizeof(int32_t), hipHostMallocDefault));
    MGC_HIP(h, hipHostMalloc((void**)&h->h_scalar, 8 * sizeof(double), hipHostMallocDefault));
    MGC_HIP(h, hipMemsetAsync(L.count, 0, MGC_NCOUNT * (1 + MGC_NSHARD) * sizeof(int32_t), h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    if (const char* wv = getenv("MEDPY_HIP_WAVE")) { h->wave_kernels = atoi(wv); h->wave_set = true; } /* development aid: A/B the kernel forms */
    { /* persistent grids of the wave kernels: as many waves as the device keeps resident */
        hipDeviceProp_t prop;
        MGC_HIP(h, hipGetDeviceProperties(&prop, device));
        int per_cu_dis = 0, per_cu_rel = 0;
        MGC_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_dis, k_discharge_w<1>, MGCW_LANES, 0));
        MGC_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_rel, k_relabel_w, MGCW_LANES, 0));
        h->wave_grid_dis = prop.multiProcessorCount * (per_cu_dis > 0 ? per_cu_dis : 8);
        h->wave_grid_rel = prop.multiProcessorCount * (per_cu_rel > 0 ? per_cu_rel : 16);
        int per_cu_26 = 0;
        MGC_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_26, k26_discharge_w, MGCW_LANES, 0));
        h->wave_grid26 = prop.multiProcessorCount * (per_cu_26 > 0 ? per_cu_26 : 4);
    }
    return MGC_OK;
}

int mgc_create(int ndim, const int64_t* shape, int connectivity, int device, mgc_handle* out)
{
    return mgc_create_impl(ndim, shape, connectivity, device, nullptr, out);
}

int mgc_create_slab(int ndim, const int64_t* gshape, int connectivity, int device, int rank, int nranks, mgc_handle* out)
{
    if (!out) return mgc_fail(nullptr, MGC_ERR_INVALID, "mgc_create_slab: out is NULL");
    *out = nullptr;
    if (ndim != 3 || !gshape) return mgc_fail(nullptr, MGC_ERR_UNSUPPORTED, "mgc_create_slab: slabs need a 3-D volume");
    if (nranks < 1 || rank < 0 || rank >= nranks) return mgc_fail(nullptr, MGC_ERR_INVALID, "bad rank %d of %d", rank, nranks);
    MgcSlabSpec sp;
    if (mgc_slab_spec(gshape[0], rank, nranks, &sp)) return mgc_fail(nullptr, MGC_ERR_INVALID, "cannot cut %lld planes into %d slabs", (long long)gshape[0], nranks);
    const int64_t lshape[3] = {sp.plane1 - sp.plane0, gshape[1], gshape[2]};
    return mgc_create_impl(3, lshape, connectivity, device, &sp, out, gshape[0]);
}

int mgc_slab_info(mgc_handle h, int64_t* info)
{
    if (!h || !info) return MGC_ERR_INVALID;
    info[0] = h->plane0; info[1] = h->plane1; info[2] = h->own0; info[3] = h->own1;
    info[4] = h->L.tz_own_lo > 0; info[5] = h->L.tz_own_hi < h->L.gz; info[6] = (int64_t)h->L.gy * h->L.gx; info[7] = h->nranks;
    return MGC_OK;
}

int mgc_solver_op(mgc_handle h, int op, int64_t a0, int64_t a1, int64_t a2, int64_t a3)
{
    if (!h) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_solver_op before mgc_build");
    MGC_HIP(h, hipSetDevice(h->device));
    if (h->L.ndir == MGC26_NDIR) return mgc_solver_op_on<HipDevT<true>>(h, op, a0, a1, a2, a3);
    return mgc_solver_op_on<HipDevT<false>>(h, op, a0, a1, a2, a3);
}

int mgc_read_counts(mgc_handle h, int32_t* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    MGC_HIP(h, hipMemcpyAsync(h->h_count, h->L.count, MGC_NCOUNT * (1 + MGC_NSHARD) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    mgc_fold_counts(h);
    memcpy(out, h->h_count, MGC_NCOUNT * sizeof(int32_t));
    return MGC_OK;
}

int mgc_halo_bytes(mgc_handle h, int kind, int64_t* bytes)
{
    if (!h || !bytes) return MGC_ERR_INVALID;
    *bytes = mgc_halo_bytes_nd(h->L, kind);
    return MGC_OK;
}

static int mgc_halo_staging(mgc_handle h, int64_t bytes)
{
    if (h->halo_cap >= bytes) return MGC_OK;
    if (h->d_halo) (void)mgc_dfree(h->d_halo);
    h->d_halo = nullptr;
    MGC_HIP(h, mgc_dmalloc(&h->d_halo, (size_t)bytes));
    h->halo_cap = bytes;
    return MGC_OK;
}

int mgc_halo_pack(mgc_handle h, int side, int kind, void* buf, int buf_on_device)
{
    if (!h || !buf || side < 0 || side > 1) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_halo_pack before mgc_build");
    if (kind < 0 || kind > 2) return mgc_fail(h, MGC_ERR_INVALID, "mgc_halo_pack: no halo kind %d", kind);
    if (side == 0 ? h->L.tz_own_lo == 0 : h->L.tz_own_hi == h->L.gz) return mgc_fail(h, MGC_ERR_INVALID, "no neighbour slab on side %d", side);
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    const int64_t bytes = mgc_halo_bytes_nd(h->L, kind);
    void* dst = buf;
    if (!buf_on_device) {
        const int rc = mgc_halo_staging(h, bytes);
        if (rc) return rc;
        dst = h->d_halo;
    }
    const int T = h->L.gy * h->L.gx;
    const bool compact = mgc_halo_compact_nd(h->L, kind);
    if (compact) MGC_HIP(h, hipMemsetAsync((char*)dst + mgc_halo_off_count_nd(h->L), 0, 4, h->stream));
    hipLaunchKernelGGL(k_halo_pack, dim3(T < 2048 ? T : 2048), dim3(MGC_TV), 0, h->stream, h->L, side, kind, dst);
    MGC_HIP(h, hipGetLastError());
    if (!buf_on_device) {
        int64_t used = bytes;
        if (compact) { /* header first, then only the records that were filled */
            const int64_t off = mgc_halo_off_rec_nd(h->L);
            MGC_HIP(h, hipMemcpyAsync(buf, dst, (size_t)off, hipMemcpyDeviceToHost, h->stream));
            MGC_HIP(h, hipStreamSynchronize(h->stream));
            int32_t cnt = 0;
            memcpy(&cnt, (const char*)buf + mgc_halo_off_count_nd(h->L), 4);
            if (cnt > h->L.halo_max_rec) cnt = h->L.halo_max_rec; /* (tiles beyond that were deferred, not packed) */
            memcpy((char*)buf + mgc_halo_off_count_nd(h->L), &cnt, 4);
            used = (int64_t)cnt * mgc_halo_rec_bytes_nd(h->L, kind);
            if (used) MGC_HIP(h, hipMemcpyAsync((char*)buf + off, (const char*)dst + off, (size_t)used, hipMemcpyDeviceToHost, h->stream));
        } else {
            MGC_HIP(h, hipMemcpyAsync(buf, dst, (size_t)used, hipMemcpyDeviceToHost, h->stream));
        }
    }
    MGC_HIP(h, hipStreamSynchronize(h->stream)); /* the transport runs on the caller's stream / thread */
    return MGC_OK;
}

int mgc_halo_unpack(mgc_handle h, int side, int kind, const void* buf, int buf_on_device, uint32_t epoch, int list)
{
    if (!h || !buf || side < 0 || side > 1) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_halo_unpack before mgc_build");
    if (kind < 0 || kind > 2) return mgc_fail(h, MGC_ERR_INVALID, "mgc_halo_unpack: no halo kind %d", kind);
    if (side == 0 ? h->L.tz_own_lo == 0 : h->L.tz_own_hi == h->L.gz) return mgc_fail(h, MGC_ERR_INVALID, "no neighbour slab on side %d", side);
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    const int64_t bytes = mgc_halo_bytes_nd(h->L, kind);
    const void* src = buf;
    if (!buf_on_device) {
        const int rc = mgc_halo_staging(h, bytes);
        if (rc) return rc;
        int64_t used = bytes;
        if (mgc_halo_compact_nd(h->L, kind)) { /* the header says how many records follow it */
            int32_t cnt = 0;
            memcpy(&cnt, (const char*)buf + mgc_halo_off_count_nd(h->L), 4);
            if (cnt < 0 || cnt > h->L.halo_max_rec) return mgc_fail(h, MGC_ERR_INVALID, "mgc_halo_unpack: record count %d out of range (message holds %d)", (int)cnt, h->L.halo_max_rec);
            used = mgc_halo_off_rec_nd(h->L) + (int64_t)cnt * mgc_halo_rec_bytes_nd(h->L, kind);
        }
        MGC_HIP(h, hipMemcpyAsync(h->d_halo, buf, (size_t)used, hipMemcpyHostToDevice, h->stream));
        src = h->d_halo;
    }
    const int T = h->L.gy * h->L.gx;
    hipLaunchKernelGGL(k_halo_unpack, dim3(T < 2048 ? T : 2048), dim3(MGC_TV), 0, h->stream, h->L, side, kind, src, epoch, list);
    MGC_HIP(h, hipGetLastError());
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    return MGC_OK;
}

/* ---- RCCL, resolved lazily from the system ROCm (same HIP runtime this library is linked to) ---- */
struct MgcRccl {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (dl) return true;
        /* the RCCL that belongs to the HIP runtime this library is linked against comes first: it sits next to
         * libamdhip64 (a bare "librccl.so" could resolve to a copy some host application already loaded) */
        std::string sibling;
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            sibling = info.dli_fname;
            const size_t slash = sibling.rfind('/');
            sibling = (slash == std::string::npos ? std::string() : sibling.substr(0, slash + 1)) + "librccl.so.1";
        }
        const char* forced = getenv("MEDPY_HIP_RCCL"); /* explicit path: site installs, and the in-process mock of the tests */
        const char* names[] = {forced ? forced : "", sibling.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
        for (const char* n : names)
            /* DEEPBIND: librccl must bind to the HIP runtime it is linked against (the system one this library
             * uses too), not to another copy a host application may have put in the global scope (PyTorch wheels
             * bundle their own libamdhip64) */
            if (*n && (dl = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND))) break;
        if (!dl) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
#define MGC_SYM(field, name) do { *(void**)(&field) = dlsym(dl, name); if (!field) { err = std::string("librccl lacks ") + name; dl = nullptr; return false; } } while (0)
        MGC_SYM(GetUniqueId, "ncclGetUniqueId"); MGC_SYM(CommInitRank, "ncclCommInitRank"); MGC_SYM(CommDestroy, "ncclCommDestroy");
        MGC_SYM(Send, "ncclSend"); MGC_SYM(Recv, "ncclRecv"); MGC_SYM(AllReduce, "ncclAllReduce");
        MGC_SYM(GroupStart, "ncclGroupStart"); MGC_SYM(GroupEnd, "ncclGroupEnd"); MGC_SYM(GetErrorString, "ncclGetErrorString");
#undef MGC_SYM
        return true;
    }
};
static MgcRccl g_rccl;

#define MGC_NCCL(h, call)                                                                                          \
    do {                                                                                                           \
        ncclResult_t r_ = (call);                                                                                  \
        if (r_ != ncclSuccess) return mgc_fail(h, MGC_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)

int mgc_comm_unique_id(uint8_t* id128)
{
    if (!id128) return MGC_ERR_INVALID;
    if (!g_rccl.load()) return mgc_fail(nullptr, MGC_ERR_UNSUPPORTED, "%s", g_rccl.err.c_str());
    ncclUniqueId id;
    MGC_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
    return MGC_OK;
}

int mgc_comm_init(mgc_handle h, const uint8_t* id128)
{
    if (!h || !id128) return MGC_ERR_INVALID;
    if (!g_rccl.load()) return mgc_fail(h, MGC_ERR_UNSUPPORTED, "%s", g_rccl.err.c_str());
    MGC_HIP(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    MGC_NCCL(h, g_rccl.CommInitRank(&h->comm, h->nranks, id, h->rank));
    if (!h->d_cnt64) MGC_HIP(h, mgc_dmalloc((void**)&h->d_cnt64, 2 * MGC_NCOUNT * sizeof(int64_t)));
    return MGC_OK;
}

__global__ void k_widen_counts(MgcLattice L, int64_t* out)
{
    if (threadIdx.x < MGC_NCOUNT) {
        int64_t v = L.count[threadIdx.x];
        for (int sh = 0; sh < L.nshard; ++sh) v += *mgc_counter(L, (int)threadIdx.x, sh);
        out[threadIdx.x] = v;
    }
}

int mgc_allreduce_counts(mgc_handle h, int64_t* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    if (!h->comm) return mgc_fail(h, MGC_ERR_STATE, "mgc_allreduce_counts before mgc_comm_init");
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    hipLaunchKernelGGL(k_widen_counts, dim3(1), dim3(64), 0, h->stream, h->L, h->d_cnt64);
    MGC_HIP(h, hipGetLastError());
    MGC_NCCL(h, g_rccl.AllReduce(h->d_cnt64, h->d_cnt64 + MGC_NCOUNT, MGC_NCOUNT, ncclInt64, ncclSum, h->comm, h->stream));
    MGC_HIP(h, hipMemcpyAsync(out, h->d_cnt64 + MGC_NCOUNT, MGC_NCOUNT * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    return MGC_OK;
}

int mgc_halo_exchange(mgc_handle h, int kind, uint32_t epoch, int list)
{
    if (!h) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_halo_exchange before mgc_build");
    if (!h->comm) return mgc_fail(h, MGC_ERR_STATE, "mgc_halo_exchange before mgc_comm_init");
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    if (kind < 0 || kind > 2) return mgc_fail(h, MGC_ERR_INVALID, "mgc_halo_exchange: kind must be 0, 1 or 2");
    int64_t bytes = mgc_halo_bytes_nd(h->L, 1); /* capacity of the largest kind; reused for all */
    if (mgc_halo_bytes_nd(h->L, 0) > bytes) bytes = mgc_halo_bytes_nd(h->L, 0);
    if (h->xchg_cap < bytes) {
        for (int i = 0; i < 4; ++i) {
            if (h->d_xchg[i]) (void)mgc_dfree(h->d_xchg[i]);
            h->d_xchg[i] = nullptr;
            MGC_HIP(h, mgc_dmalloc(&h->d_xchg[i], (size_t)bytes));
        }
        h->xchg_cap = bytes;
    }
    /* compacted kinds: the fixed header plus halo_max_rec record slots, in ONE grouped transfer whose size both sides know
     * without asking the device (round 2 moved the header, read the two record counts back on the host -- a stream drain per
     * exchange -- and then moved exactly that many records in a second group).  Tiles that found the message full were
     * deferred by the pack kernel (MGC_CNT_DEFERRED). */
    const bool compact = mgc_halo_compact_nd(h->L, kind);
    const int64_t nb = compact ? mgc_halo_off_rec_nd(h->L) + (int64_t)h->L.halo_max_rec * mgc_halo_rec_bytes_nd(h->L, kind) : mgc_halo_bytes_nd(h->L, kind);
    const bool has[2] = {h->L.tz_own_lo > 0, h->L.tz_own_hi < h->L.gz};
    const int peer[2] = {h->rank - 1, h->rank + 1};
    const int T = h->L.gy * h->L.gx, grid = T < 2048 ? T : 2048;
    for (int side = 0; side < 2; ++side)
        if (has[side]) {
            if (compact) MGC_HIP(h, hipMemsetAsync((char*)h->d_xchg[2 * side] + mgc_halo_off_count_nd(h->L), 0, 4, h->stream));
            hipLaunchKernelGGL(k_halo_pack, dim3(grid), dim3(MGC_TV), 0, h->stream, h->L, side, kind, h->d_xchg[2 * side]);
            MGC_HIP(h, hipGetLastError());
        }
    MGC_NCCL(h, g_rccl.GroupStart());
    for (int side = 0; side < 2; ++side)
        if (has[side]) {
            MGC_NCCL(h, g_rccl.Send(h->d_xchg[2 * side], (size_t)nb, ncclUint8, peer[side], h->comm, h->stream));
            MGC_NCCL(h, g_rccl.Recv(h->d_xchg[2 * side + 1], (size_t)nb, ncclUint8, peer[side], h->comm, h->stream));
        }
    MGC_NCCL(h, g_rccl.GroupEnd());
    for (int side = 0; side < 2; ++side)
        if (has[side]) {
            hipLaunchKernelGGL(k_halo_unpack, dim3(grid), dim3(MGC_TV), 0, h->stream, h->L, side, kind, (const void*)h->d_xchg[2 * side + 1], epoch, list);
            MGC_HIP(h, hipGetLastError());
        }
    return MGC_OK;
}

} /* extern "C" */

/* ---- the native channel of a slab whose neighbours live in other processes: RCCL over xGMI, stream-ordered ---- */
template <bool FULL>
int HipDevT<FULL>::native_exchange(int kind, uint32_t epoch, int list) { return mgc_halo_exchange(h, kind, epoch, list); }

template <bool FULL>
int HipDevT<FULL>::native_allreduce(int64_t* v, int n, int op)
{
    if (n > MGC_NCOUNT) return mgc_fail(h, MGC_ERR_INVALID, "native_allreduce: %d values", n);
    flush_zero();
    MGC_HIP(h, hipMemcpyAsync(h->d_cnt64, v, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    MGC_NCCL(h, g_rccl.AllReduce(h->d_cnt64, h->d_cnt64 + MGC_NCOUNT, (size_t)n, ncclInt64, op == 1 ? ncclMin : ncclSum, h->comm, h->stream));
    MGC_HIP(h, hipMemcpyAsync(v, h->d_cnt64 + MGC_NCOUNT, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    return MGC_OK;
}

template <bool FULL>
int HipDevT<FULL>::native_send(int side, const void* buf, int64_t n)
{
    MGC_NCCL(h, g_rccl.Send(buf, (size_t)n, ncclUint8, h->rank + (side ? 1 : -1), h->comm, h->stream));
    return MGC_OK;
}

template <bool FULL>
int HipDevT<FULL>::native_recv(int side, void* buf, int64_t n)
{
    MGC_NCCL(h, g_rccl.Recv(buf, (size_t)n, ncclUint8, h->rank + (side ? 1 : -1), h->comm, h->stream));
    return MGC_OK;
}

/* what mgc_get_stats reports for one slab after a group solve (its own launches and tiles; the global totals are in mgc_slab_stats) */
template <class Dev>
static void mgc_slab_handle_stats(mgc_handle h, Dev& dev, const MgcLayout lay, const MgcSolveStats& st)
{
    int cnt[MGC_NCOUNT];
    dev.resolve_timing();
    h->stats.discharge_ms = dev.discharge_ms + dev.block_ms;
    h->stats.discharge_wave_ms = dev.discharge_ms;
    h->stats.discharge_wave_launches = dev.seen[0];
    h->stats.timing_stride = h->timing ? h->timing_stride : 0;
    h->stats.relabel_ms = dev.relabel_ms;
    h->stats.discharge_launches = dev.discharge_launches;
    h->stats.relabel_launches = dev.relabel_launches;
    h->stats.reserved[0] = dev.readbacks;
    h->stats.reserved[1] = st.radial_cycles;
    h->stats.reserved[2] = h->wall_tiles;
    dev.read_counts(cnt);
    h->stats.discharge_tiles = cnt[lay.cnt_dis];
    h->stats.relabel_tiles = cnt[lay.cnt_rel];
    h->stats.discharge_wave_tiles = h->L.ndir == 6 ? cnt[MGC_CNT_WAVE_TILES] : cnt[lay.cnt_dis];
    h->stats.global_relabels = st.outer;
    h->stats.phases = st.phases;
}

/* mgc_solve (mgc_driver.inl) over the slabs of one volume: the single handle's schedule with the borders exchanged at its hook points */
template <class Dev>
static int mgc_solve_slabs_on(mgc_handle* hs, int n, const mgc_transport* cb, const MgcLayout lay, mgc_slab_stats* out)
{
    mgc_handle h0 = hs[0];
    const bool all_local = n == h0->nranks;
    std::vector<Dev> devs((size_t)n);
    std::vector<Dev*> ptr((size_t)n);
    std::vector<hipStream_t> own_stream((size_t)n);
    for (int i = 0; i < n; ++i) {
        devs[(size_t)i].h = hs[i];
        ptr[(size_t)i] = &devs[(size_t)i];
        /* the slabs of one device share ONE stream for the length of the solve: their launches are in order without an event between them,
         * and a slab's kernels run alone on the device (what is measured per slab is what a device of its own would take) */
        own_stream[(size_t)i] = hs[i]->stream;
        hs[i]->stream = h0->stream;
        hs[i]->timing_offset++;
        hs[i]->solved = false;
    }
    MgcXchg<Dev> x(ptr, cb, all_local);
    MgcSlabGroup<Dev, MgcXchg<Dev>> group(ptr, x);
    MgcSolveParams P = h0->params;
    if (h0->L.ndir == 6 && P.radial == 2) { /* wall tiles of the WHOLE volume decide (mgc_maxflow's rule) */
        int64_t walls = 0;
        for (int i = 0; i < n; ++i) walls += hs[i]->wall_tiles;
        x.allreduce(&walls, 1, 0);
        P.radial = walls >= h0->radial_min_walls ? 1 : 0;
    }
    for (int i = 0; i < n; ++i) { hs[i]->repeat_now = (h0->repeat_steps & (P.radial ? 2 : 1)) != 0; hs[i]->radial_cycle_no = 0; }
    MgcSolveStats st;
    const int rc = mgc_solve(group, h0->L, P, st, lay);
    hipError_t first = hipSuccess;
    for (int i = 0; i < n; ++i) {
        mgc_flush_zero(hs[i]);
        if (devs[(size_t)i].first_error != hipSuccess && first == hipSuccess) first = devs[(size_t)i].first_error;
    }
    if (first == hipSuccess) first = hipStreamSynchronize(h0->stream);
    for (int i = 0; i < n; ++i) {
        if (first == hipSuccess) mgc_slab_handle_stats(hs[i], devs[(size_t)i], lay, st);
        hs[i]->stream = own_stream[(size_t)i];
    }
    if (first != hipSuccess) return mgc_fail(h0, MGC_ERR_HIP, "slab solver: HIP error %s", hipGetErrorString(first));
    if (x.error) return mgc_fail(h0, x.error, "slab solver: the transport failed (%s)", h0->err.c_str());
    if (out) {
        mgc_slab_stats so{};
        so.outer = st.outer; so.relabel_passes = st.relabel_passes; so.phases = st.phases; so.exchanges = group.exchanges; so.reductions = group.reductions;
        so.converged = st.converged; so.discharge_tiles = st.discharge_tiles; so.relabel_tiles = st.relabel_tiles; so.deferred_drains = st.deferred_drains;
        so.reserved[0] = st.radial_cycles;
        *out = so;
    }
    if (rc) return mgc_fail(h0, MGC_ERR_NOT_CONVERGED, "slab solver did not converge within %d global relabels", P.max_outer);
    return MGC_OK;
}

extern "C" {

int mgc_solve_slabs(mgc_handle* hs, int n, const mgc_transport* t, mgc_slab_stats* out)
{
    if (!hs || n < 1 || !hs[0]) return MGC_ERR_INVALID;
    mgc_handle h0 = hs[0];
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) return MGC_ERR_INVALID;
        if (!hs[i]->built) return mgc_fail(h0, MGC_ERR_STATE, "mgc_solve_slabs before mgc_build (slab %d)", i);
        if (hs[i]->device != h0->device || hs[i]->L.ndir != h0->L.ndir || hs[i]->nranks != h0->nranks || hs[i]->gd0 != h0->gd0 || hs[i]->L.dy != h0->L.dy || hs[i]->L.dx != h0->L.dx)
            return mgc_fail(h0, MGC_ERR_INVALID, "mgc_solve_slabs: the local slabs are cut from one volume and live on one device");
        if (n > 1 && hs[i]->rank != i) return mgc_fail(h0, MGC_ERR_INVALID, "mgc_solve_slabs: all slabs of the volume, in rank order");
    }
    if (n > 1 && n != h0->nranks) return mgc_fail(h0, MGC_ERR_INVALID, "mgc_solve_slabs: %d local slabs of a volume cut into %d (either all of them or this rank's one)", n, h0->nranks);
    if (n == 1 && h0->nranks > 1) {
        if (!h0->comm && !(t && t->exchange && t->allreduce && t->send && t->recv))
            return mgc_fail(h0, MGC_ERR_STATE, "mgc_solve_slabs: this slab has neighbours in other processes: call mgc_comm_init first (RCCL), or pass the four callbacks of a host transport");
    }
    MGC_HIP(h0, hipSetDevice(h0->device));
    MgcRange range_("mgc_solve_slabs");
    float ms = 0.f;
    MGC_HIP(h0, hipEventRecord(h0->ev[0], h0->stream));
    const int rc = h0->L.ndir == MGC26_NDIR ? mgc_solve_slabs_on<HipDev26>(hs, n, t, mgc_layout26(), out) : mgc_solve_slabs_on<HipDev>(hs, n, t, mgc_layout6(), out);
    if (rc == MGC_OK) {
        MGC_HIP(h0, hipEventRecord(h0->ev[1], h0->stream));
        MGC_HIP(h0, hipStreamSynchronize(h0->stream));
        MGC_HIP(h0, hipEventElapsedTime(&ms, h0->ev[0], h0->ev[1]));
        for (int i = 0; i < n; ++i) hs[i]->stats.solve_ms = ms;
    }
    return rc;
}

/* this rank's slab (the entry point of rounds 3 - 5; the same schedule since round 6) */
int mgc_solve_slab(mgc_handle h, mgc_slab_stats* out) { return mgc_solve_slabs(&h, 1, nullptr, out); }

int mgc_destroy(mgc_handle h)
{
    if (!h) return MGC_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    MgcLattice& L = h->L;
    void* ptrs[] = {L.rcap, L.cap0, L.excess, L.sink, L.height, L.rmask, L.rmask32, L.obox, L.oflags, L.list[0], L.list[1], L.list[2],
                    L.list[3], L.list[4], L.list[5], L.list[6], L.list[7], L.list[8], L.list[9], L.list[10], L.list[11], L.list[12],
                    L.list[13], L.list[14], L.list[15], L.list[16], L.list[17], L.count, L.stamp, L.rstamp, L.status, h->d_tr0, h->d_part, h->d_part2, h->d_scalar,
                    h->d_labels, h->d_tflags, h->d_tsum, h->d_image, h->d_lut, h->d_prob, h->d_fg, h->d_bg, h->d_tr_in, h->d_eslot, h->d_eval, h->d_erun, L.hshadow[0], L.hshadow[1], h->d_vout, h->d_dt16, h->d_ds16, h->d_hexact, h->d_halo, h->d_xchg[0], h->d_xchg[1], h->d_xchg[2], h->d_xchg[3], h->d_cnt64, h->d_carry[0], h->d_carry[1], h->d_carry_in[0], h->d_carry_in[1]};
    for (void* p : ptrs)
        if (p) (void)mgc_dfree(p);
    if (h->h_count) (void)hipHostFree(h->h_count);
    if (h->h_scalar) (void)hipHostFree(h->h_scalar);
    free(h->h_labels);
    for (int i = 0; i < 4; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->op_ev) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MGC_OK;
}

/* host -> device copy into a buffer owned by the handle; the buffer grows when a later call needs more (a second image of
 * a wider dtype, a larger batch of explicit edges) */
/* Host <-> HBM through pinned staging, several chunks in flight.  The caller's arrays are pageable (NumPy): a plain hipMemcpy of
 * such memory is staged by the runtime through ONE bounce buffer by ONE thread (measured round 4: 805 MB in 53 ms = 15 GB/s up, 134 MB in
 * 28 ms = 4.7 GB/s down).  Here MGC_STAGE_THREADS host threads fill / drain their own pinned chunks and the DMA engine moves chunk k while
 * the threads copy chunk k + 1: the PCIe link, not a memcpy, sets the pace.  Blocks until the whole transfer is done (the boundary's
 * contract: the caller's buffer may be freed on return).  Falls back to the plain copy for small transfers or if no pinned memory is to be had. */
#ifndef MGC_STAGE_THREADS
#define MGC_STAGE_THREADS 4    /* device -> host */
#define MGC_STAGE_THREADS_UP 4 /* host -> device.  tools/gpu_transfer_probe.py on the GPU box, 805 MB up / 134 MB down, warm: the runtime's own
                                  staging 22.7 ms (35 GB/s) / 22.5 ms; 2 threads 16.6 / 20.5; 4 threads 16.0 ms (50 GB/s) / 19.2; 8: 16.4 / 19.6 */
#endif
#define MGC_STAGE_CHUNK ((size_t)8 << 20)
struct MgcStage { /* one per process and device: pinning memory costs tens of milliseconds, a handle per volume must not pay it again */
    std::mutex lock;
    void* pinned = nullptr;
    hipStream_t stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool failed = false;
};
static MgcStage g_stage[16];

static hipError_t mgc_staged_copy(mgc_handle h, void* dev, void* host, size_t bytes, bool to_device)
{
    if (bytes < 4 * MGC_STAGE_CHUNK) {
        hipError_t e = to_device ? hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream) : hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream);
        return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
    }
    static const int env_threads = getenv("MEDPY_HIP_STAGE_THREADS") ? atoi(getenv("MEDPY_HIP_STAGE_THREADS")) : -1; /* 0: the runtime's own staging; n: n threads (at most 8) */
    const int want = env_threads >= 0 ? env_threads : (to_device ? MGC_STAGE_THREADS_UP : MGC_STAGE_THREADS);
    if (want == 0) {
        hipError_t e = to_device ? hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream) : hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream);
        return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
    }
    const int NT = want > 8 ? 8 : want, SLOTS = 2; /* two chunks per thread: one being filled / drained by the host, one on the wire */
    MgcStage& G = g_stage[h->device & 15];
    std::lock_guard<std::mutex> guard(G.lock); /* (one staged transfer per device at a time: they would share the link anyway) */
    if (!G.pinned && !G.failed) {
        if (hipHostMalloc(&G.pinned, (size_t)8 * SLOTS * MGC_STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            G.pinned = nullptr;
            G.failed = true;
        } else {
            for (int i = 0; i < 8; ++i) if (hipStreamCreateWithFlags(&G.stream[i], hipStreamNonBlocking) != hipSuccess) G.stream[i] = nullptr;
        }
    }
    if (!G.pinned) {
        hipError_t e = to_device ? hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, h->stream) : hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream);
        return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
    }
    hipError_t first = hipStreamSynchronize(h->stream); /* what the transfer reads / overwrites is settled */
    if (first != hipSuccess) return first;
    const size_t nchunks = (bytes + MGC_STAGE_CHUNK - 1) / MGC_STAGE_CHUNK;
    std::vector<std::thread> th;
    std::vector<hipError_t> err((size_t)NT, hipSuccess);
    const int device = h->device;
    for (int t = 0; t < NT; ++t) {
        th.emplace_back([=, &err, &G]() {
            (void)hipSetDevice(device);
            hipStream_t st = G.stream[t] ? G.stream[t] : h->stream;
            char* const base = (char*)G.pinned + (size_t)t * SLOTS * MGC_STAGE_CHUNK;
            hipEvent_t ev[SLOTS];
            for (int k = 0; k < SLOTS; ++k) if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { err[t] = hipErrorUnknown; return; }
            bool used[SLOTS] = {false, false};
            size_t pend_off[SLOTS] = {0, 0}, pend_len[SLOTS] = {0, 0};
            int slot = 0;
            for (size_t c = (size_t)t; c < nchunks && err[t] == hipSuccess; c += (size_t)NT, slot ^= 1) {
                const size_t off = c * MGC_STAGE_CHUNK, len = bytes - off < MGC_STAGE_CHUNK ? bytes - off : MGC_STAGE_CHUNK;
                char* const pin = base + (size_t)slot * MGC_STAGE_CHUNK;
                if (used[slot]) { /* the slot's previous transfer must be through before it is reused */
                    hipError_t e = hipEventSynchronize(ev[slot]);
                    if (e != hipSuccess) { err[t] = e; break; }
                    if (!to_device) memcpy((char*)host + pend_off[slot], pin, pend_len[slot]);
                }
                hipError_t e;
                if (to_device) {
                    memcpy(pin, (const char*)host + off, len);
                    e = hipMemcpyAsync((char*)dev + off, pin, len, hipMemcpyHostToDevice, st);
                } else {
                    e = hipMemcpyAsync(pin, (const char*)dev + off, len, hipMemcpyDeviceToHost, st);
                    pend_off[slot] = off; pend_len[slot] = len;
                }
                if (e == hipSuccess) e = hipEventRecord(ev[slot], st);
                if (e != hipSuccess) { err[t] = e; break; }
                used[slot] = true;
            }
            for (int k = 0; k < SLOTS; ++k) {
                if (used[k]) {
                    hipError_t e = hipEventSynchronize(ev[k]);
                    if (e != hipSuccess && err[t] == hipSuccess) err[t] = e;
                    if (!to_device && e == hipSuccess) memcpy((char*)host + pend_off[k], base + (size_t)k * MGC_STAGE_CHUNK, pend_len[k]);
                }
                (void)hipEventDestroy(ev[k]);
            }
        });
    }
    for (auto& x : th) x.join();
    for (hipError_t e : err) if (e != hipSuccess) return e;
    return hipSuccess;
}

static int mgc_upload(mgc_handle h, void** dst, const void* src, size_t bytes)
{
    size_t& cap = h->buf_cap[(const void*)dst];
    if (*dst && cap < bytes) {
        MGC_HIP(h, hipStreamSynchronize(h->stream));
        MGC_HIP(h, mgc_dfree(*dst));
        h->device_bytes -= (int64_t)cap;
        *dst = nullptr;
        cap = 0;
    }
    if (!*dst) {
        MGC_HIP(h, mgc_dmalloc(dst, bytes));
        h->device_bytes += (int64_t)bytes;
        cap = bytes;
    }
    MGC_HIP(h, mgc_staged_copy(h, *dst, const_cast<void*>(src), bytes, true));
    return MGC_OK;
}

int mgc_set_boundary(mgc_handle h, int term, const void* image, int dtype, double sigma, const double* spacing)
{
    if (!h) return MGC_ERR_INVALID;
    if (term < MGC_TERM_NONE || term > MGC_TERM_MAXIMUM_POWER) return mgc_fail(h, MGC_ERR_INVALID, "unknown boundary term %d", term);
    MGC_HIP(h, hipSetDevice(h->device));
    h->term = term;
    h->lut_n = 0; /* (a table belongs to one image / term / sigma) */
    h->range_set = false; /* a range handed over for the previous image does not describe this one */
    h->built = h->solved = false;
    if (term == MGC_TERM_NONE) return MGC_OK;
    const size_t es = mgc_dtype_size(dtype);
    if (!image || !es) return mgc_fail(h, MGC_ERR_INVALID, "mgc_set_boundary: image NULL or bad dtype %d", dtype);
    if (h->d_image && h->img_dtype != dtype) { (void)mgc_dfree(h->d_image); h->d_image = nullptr; }
    h->img_dtype = dtype;
    h->sigma = sigma;
    h->has_spacing = spacing ? 1 : 0;
    for (int k = 0; k < 3; ++k) h->spacing[k] = 1.0;
    if (spacing)
        for (int k = 0; k < h->ndim; ++k) h->spacing[3 - h->ndim + k] = spacing[k];
    return mgc_upload(h, &h->d_image, image, (size_t)h->nvox * es);
}

int mgc_set_boundary_lut(mgc_handle h, const double* table, int64_t n)
{
    if (!h) return MGC_ERR_INVALID;
    if (n < 0 || n > 65536 || (n > 0 && !table)) return mgc_fail(h, MGC_ERR_INVALID, "mgc_set_boundary_lut: 0 <= n <= 65536 entries");
    MGC_HIP(h, hipSetDevice(h->device));
    h->built = h->solved = false;
    h->lut_n = 0;
    if (n == 0) return MGC_OK;
    const int term = h->term;
    if (term != MGC_TERM_DIFFERENCE_EXPONENTIAL && term != MGC_TERM_MAXIMUM_EXPONENTIAL && term != MGC_TERM_DIFFERENCE_POWER && term != MGC_TERM_MAXIMUM_POWER)
        return mgc_fail(h, MGC_ERR_STATE, "mgc_set_boundary_lut: only the exponential and power terms are evaluated by table (the others are IEEE-basic arithmetic)");
    const int rc = mgc_upload(h, &h->d_lut, table, (size_t)n * sizeof(double));
    if (rc != MGC_OK) return rc;
    h->lut_n = (int)n;
    return MGC_OK;
}

int mgc_set_regional_probability(mgc_handle h, const void* pm, int dtype, double alpha)
{
    if (!h) return MGC_ERR_INVALID;
    if (!pm || (dtype != MGC_F32 && dtype != MGC_F64)) return mgc_fail(h, MGC_ERR_INVALID, "probability map must be float32 or float64");
    MGC_HIP(h, hipSetDevice(h->device));
    if (h->d_prob && h->prob_dtype != dtype) { (void)mgc_dfree(h->d_prob); h->d_prob = nullptr; }
    h->prob_dtype = dtype;
    h->alpha = alpha;
    h->built = h->solved = false;
    return mgc_upload(h, &h->d_prob, pm, (size_t)h->nvox * mgc_dtype_size(dtype));
}

int mgc_set_markers(mgc_handle h, const uint8_t* fg, const uint8_t* bg)
{
    if (!h) return MGC_ERR_INVALID;
    MGC_HIP(h, hipSetDevice(h->device));
    h->built = h->solved = false;
    int rc = MGC_OK;
    if (fg) rc = mgc_upload(h, (void**)&h->d_fg, fg, (size_t)h->nvox);
    else if (h->d_fg) { (void)mgc_dfree(h->d_fg); h->d_fg = nullptr; }
    if (rc) return rc;
    if (bg) rc = mgc_upload(h, (void**)&h->d_bg, bg, (size_t)h->nvox);
    else if (h->d_bg) { (void)mgc_dfree(h->d_bg); h->d_bg = nullptr; }
    return rc;
}

int mgc_set_tweights_merged(mgc_handle h, const double* tr, double flow_const)
{
    if (!h || !tr) return MGC_ERR_INVALID;
    MGC_HIP(h, hipSetDevice(h->device));
    h->flow_const_in = flow_const;
    h->built = h->solved = false;
    return mgc_upload(h, (void**)&h->d_tr_in, tr, (size_t)h->nvox * sizeof(double));
}

int mgc_validate(mgc_handle h, mgc_validation* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_validate before mgc_build");
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    MgcLattice& L = h->L;
    memset(out, 0, sizeof(*out));
    if (!h->labels_valid) return mgc_fail(h, MGC_ERR_STATE, "mgc_validate before a solve: the distance labels of this build were never computed");
    if (!h->d_vout) MGC_HIP(h, mgc_dmalloc((void**)&h->d_vout, sizeof(MgcValidateOut)));
    MgcValidateOut* const d_out = (MgcValidateOut*)h->d_vout;
    MGC_HIP(h, hipMemsetAsync(d_out, 0, sizeof(MgcValidateOut), h->stream));
    const int grid = L.ntiles < h->grid_cap * 4 ? L.ntiles : h->grid_cap * 4;
    hipLaunchKernelGGL(k_validate, dim3(grid), dim3(MGC_TV), 0, h->stream, L, h->build_args, (const double*)h->d_tr0, h->d_part, d_out);
    MGC_HIP(h, hipGetLastError());
    mgc_sum_partials(h, (int64_t)L.ntiles, h->d_scalar + 2);
    /* the capacity of the cut the current labels define (the labels are read from the distance labels, as k_labels does) */
    { const int rc = mgc_launch_readout(h, 3, nullptr); if (rc) return rc; }
    MgcValidateOut ho;
    MGC_HIP(h, hipMemcpyAsync(&ho, d_out, sizeof(ho), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipMemcpyAsync(h->h_scalar, h->d_scalar, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    h->labels_on_host = false;
    out->voxels = (int64_t)ho.cnt[0]; out->negative_values = (int64_t)ho.cnt[1]; out->active_excess = (int64_t)ho.cnt[2];
    out->residual_arcs_across = (int64_t)ho.cnt[3]; out->sink_links_across = (int64_t)ho.cnt[4];
    out->pair_violations = (int64_t)ho.cnt[5]; out->node_violations = (int64_t)ho.cnt[6]; out->pending_outbox = (int64_t)ho.cnt[7];
    memcpy(&out->max_pair_error, &ho.max_pair_bits, sizeof(double));
    memcpy(&out->max_node_error, &ho.max_node_bits, sizeof(double));
    out->flow_into_sink = h->h_scalar[2];
    out->cut_capacity = h->h_scalar[3];
    out->flow_constant = h->flow_const;
    out->sink_capacity_used = ho.sink_cap_used;
    return MGC_OK;
}

int mgc_get_image_range(mgc_handle h, double* out3)
{
    if (!h || !out3) return MGC_ERR_INVALID;
    if (!h->d_image) return mgc_fail(h, MGC_ERR_STATE, "mgc_get_image_range before mgc_set_boundary");
    MGC_HIP(h, hipSetDevice(h->device));
    const hipError_t e = mgc_image_range((const void*)h->d_image, h->img_dtype, h->nvox, h->d_part, h->stream, &out3[0], &out3[1], &out3[2]);
    MGC_HIP(h, e);
    return MGC_OK;
}

int mgc_set_image_range(mgc_handle h, const double* in3)
{
    if (!h) return MGC_ERR_INVALID;
    h->range_set = in3 != nullptr;
    if (in3) { h->range[0] = in3[0]; h->range[1] = in3[1]; h->range[2] = in3[2]; }
    h->built = h->solved = false;
    return MGC_OK;
}

int mgc_add_edges(mgc_handle h, int64_t n, const int64_t* i, const int64_t* j, const double* cap, const double* rev)
{
    if (!h || n < 0 || (n && (!i || !j || !cap || !rev))) return MGC_ERR_INVALID;
    if (h->n_edges && !h->edges_applied) return mgc_fail(h, MGC_ERR_UNSUPPORTED, "mgc_add_edges: one batch per build (concatenate on the host)");
    if (h->n_edges && h->edges_applied) h->built = h->solved = false; /* the built graph still holds the batch that is dropped here */
    h->n_edges = h->n_runs = 0; /* a batch that a build already applied is replaced */
    h->edges_applied = false;
    if (n == 0) return MGC_OK;
    MGC_HIP(h, hipSetDevice(h->device));
    const MgcLattice& L = h->L;
    /* every edge feeds two arc slots: i -> j gets cap, j -> i gets rev (sum_edge, graph.h:457-480) */
    std::vector<int64_t> slot((size_t)(2 * n));
    std::vector<double> val((size_t)(2 * n));
    for (int64_t k = 0; k < n; ++k) {
        if (i[k] < 0 || j[k] < 0 || i[k] >= L.nvox || j[k] >= L.nvox)
            return mgc_fail(h, MGC_ERR_INVALID, "mgc_add_edges: edge %lld joins nodes %lld and %lld outside 0..%lld", (long long)k, (long long)i[k], (long long)j[k], (long long)L.nvox - 1);
        const int d = mgc_arc_direction(L, i[k], j[k]);
        if (d < 0)
            return mgc_fail(h, MGC_ERR_UNSUPPORTED, "mgc_add_edges: edge %lld (%lld, %lld) does not join lattice neighbours (use the sparse-graph solver, msg_*)",
                            (long long)k, (long long)i[k], (long long)j[k]);
        const int dr = L.ndir == 6 ? (d ^ 1) : (25 - d);
        int ti, li, tj, lj;
        mgc_node_to_tile(L, i[k], ti, li);
        mgc_node_to_tile(L, j[k], tj, lj);
        slot[(size_t)(2 * k)] = ((int64_t)ti * L.ndir + d) * MGC_TV + li;
        val[(size_t)(2 * k)] = cap[k];
        slot[(size_t)(2 * k + 1)] = ((int64_t)tj * L.ndir + dr) * MGC_TV + lj;
        val[(size_t)(2 * k + 1)] = rev[k];
    }
    std::vector<int64_t> order((size_t)(2 * n));
    for (size_t k = 0; k < order.size(); ++k) order[k] = (int64_t)k;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return slot[(size_t)a] < slot[(size_t)b]; });
    std::vector<int64_t> sslot(order.size()), run;
    std::vector<double> sval(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        sslot[k] = slot[(size_t)order[k]];
        sval[k] = val[(size_t)order[k]];
        if (k == 0 || sslot[k] != sslot[k - 1]) run.push_back((int64_t)k);
    }
    const int64_t n_runs = (int64_t)run.size();
    run.push_back((int64_t)order.size());
    int rc;
    if ((rc = mgc_upload(h, (void**)&h->d_eslot, sslot.data(), sslot.size() * sizeof(int64_t)))) return rc;
    if ((rc = mgc_upload(h, (void**)&h->d_eval, sval.data(), sval.size() * sizeof(double)))) return rc;
    if ((rc = mgc_upload(h, (void**)&h->d_erun, run.data(), run.size() * sizeof(int64_t)))) return rc;
    h->n_edges = n;
    h->n_runs = n_runs;
    h->built = h->solved = false;
    return MGC_OK;
}

int mgc_build(mgc_handle h)
{
    if (!h) return MGC_ERR_INVALID;
    MGC_HIP(h, hipSetDevice(h->device));
    MgcRange range_("mgc_build");
    MgcLattice& L = h->L;
    MgcBuildArgs A{};
    A.image = h->d_image; A.img_dtype = h->img_dtype; A.term = h->d_image ? h->term : MGC_TERM_NONE;
    MGC_HIP(h, hipEventRecord(h->ev[0], h->stream));
    A.p0 = h->sigma;
    if (A.term == MGC_TERM_DIFFERENCE_EXPONENTIAL || A.term == MGC_TERM_MAXIMUM_EXPONENTIAL) A.p0 = pow(h->sigma, 2); /* math.pow(sigma, 2) */
    if (A.term == MGC_TERM_DIFFERENCE_LINEAR || A.term == MGC_TERM_MAXIMUM_LINEAR) {
        double mn, mx, ma;
        if (h->range_set) {
            mn = h->range[0]; mx = h->range[1]; ma = h->range[2];
        } else if (h->nranks > 1) {
            /* a slab only sees its own planes: normalising by the local range would give the two sides of a slab border
             * different capacities for the same arc */
            return mgc_fail(h, MGC_ERR_STATE, "the *_linear terms normalise by the intensity range of the WHOLE volume: reduce "
                                              "mgc_get_image_range over the ranks and hand the result to mgc_set_image_range before mgc_build");
        } else {
            const hipError_t e = mgc_image_range((const void*)h->d_image, h->img_dtype, h->nvox, h->d_part, h->stream, &mn, &mx, &ma);
            MGC_HIP(h, e);
        }
        A.p0 = (A.term == MGC_TERM_MAXIMUM_LINEAR) ? ma : mgc_range_in_dtype(mn, mx, h->img_dtype); /* energy_voxel.py:101 / 174-176 */
    }
    A.has_spacing = h->has_spacing;
    A.inv_axis[0] = h->spacing[2]; A.inv_axis[1] = h->spacing[1]; A.inv_axis[2] = h->spacing[0];
    for (int d = 0; d < 26; ++d) {
        int dz, dy, dx;
        mgc26_offset(d, dz, dy, dx);
        /* math.sqrt(sum((o_k * s_k) ** 2)) in array-axis order, as oracle/energy_numpy.py:boundary_weights_offsets */
        double acc = 0.0;
        const int off[3] = {dz, dy, dx};
        for (int k = 3 - h->ndim; k < 3; ++k) acc += (off[k] * h->spacing[k]) * (off[k] * h->spacing[k]);
        A.div26[d] = sqrt(acc);
    }
    A.prob = h->d_prob; A.prob_dtype = h->prob_dtype; A.alpha = h->alpha;
    A.prepush = h->prepush;
    A.lut = h->lut_n > 0 ? (const double*)h->d_lut : nullptr; A.lut_n = h->lut_n;
    A.fg = h->d_fg; A.bg = h->d_bg; A.tr_in = h->d_tr_in;
    A.tr0 = h->d_tr0; A.fpart = h->d_part; A.tflags = h->d_tflags;
    if (h->n_edges && !L.cap0) { /* explicit edges change capacities that the image no longer determines */
        const int rc = mgc_alloc(h, &L.cap0, (int64_t)L.ntiles * MGC_TV * L.ndir);
        if (rc) return rc;
    }
    h->build_args = A;
    const int grid = L.ntiles < h->grid_cap * 4 ? L.ntiles : h->grid_cap * 4;
    const int bgrid = grid >= 8 ? grid / 8 * 8 : grid; /* k_build deals tiles to XCDs: multiple of 8 */
    MGC_HIP(h, hipMemsetAsync(L.count, 0, MGC_NCOUNT * (1 + MGC_NSHARD) * sizeof(int32_t), h->stream)); /* (k_build counts in MGC_CNT_NOT_FULL) */
    h->zero_mask = 0;
    h->pending_zero = -1;
    h->filt[0] = h->filt[1] = 0;
    if (L.ndir == 6) mgc_launch_build<false>(A.term, bgrid, h->stream, L, A);
    else mgc_launch_build<true>(A.term, bgrid, h->stream, L, A);
    MGC_HIP(h, hipGetLastError());
    mgc_sum_partials(h, (int64_t)L.ntiles, h->d_scalar);
    MGC_HIP(h, hipGetLastError());
    if (h->n_edges) {
        hipLaunchKernelGGL(k_add_edges, dim3(1024), dim3(256), 0, h->stream, L, h->n_runs, (const int64_t*)h->d_eslot,
                           (const double*)h->d_eval, (const int64_t*)h->d_erun);
        MGC_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(k_refresh_mask, dim3(grid), dim3(MGC_TV), 0, h->stream, L);
        MGC_HIP(h, hipGetLastError());
    }
    if (L.ndir == MGC26_NDIR) {
        /* slabs: a ghost tile's excess / residuals only collect what is pushed over the border (mgc26_halo_pack_tile) */
        const int64_t T = (int64_t)L.gy * L.gx;
        for (int side = 0; side < 2; ++side) {
            if (side == 0 ? L.tz_own_lo == 0 : L.tz_own_hi == L.gz) continue;
            const int64_t t0 = (side ? L.tz_own_hi : L.tz_own_lo - 1) * T;
            MGC_HIP(h, hipMemsetAsync(L.excess + t0 * MGC_TV, 0, (size_t)T * MGC_TV * sizeof(double), h->stream));
            MGC_HIP(h, hipMemsetAsync(L.rcap + t0 * MGC26_NDIR * MGC_TV, 0, (size_t)T * MGC26_NDIR * MGC_TV * sizeof(double), h->stream));
        }
    }
    MGC_HIP(h, hipEventRecord(h->ev[1], h->stream));
    MGC_HIP(h, hipMemcpyAsync(h->h_scalar, h->d_scalar, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipMemcpyAsync(h->h_count, L.count, MGC_NCOUNT * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    /* every n-link of the volume residual (and nothing added on top that the mask refresh could have changed): the first
     * global relabel of the solve is a distance transform (mgc_dt_ops.inl) */
    h->sink_tiles = L.ndir == 6 ? h->h_count[MGC_CNT_SINK_TILES] : 0;
    h->wall_tiles = L.ndir == 6 ? h->h_count[MGC_CNT_WALL_TILES] : 0;
    /* the two build counters are read: their slots (MGC_CNT_NOT_FULL is MGC_CNT_DEFERRED during a solve) are cleared with the next
     * batch of counter clears, whichever schedule drives the solve */
    h->zero_mask |= (1u << MGC_CNT_NOT_FULL) | (1u << MGC_CNT_SINK_TILES) | (1u << MGC_CNT_WALL_TILES);
    h->all_residual = L.ndir == 6 && A.term != MGC_TERM_NONE && h->h_count[MGC_CNT_NOT_FULL] == 0 && !h->n_edges &&
                      h->gd0 + L.dy + L.dx < MGC_DT_INF - 8; /* (a slab: of ITS planes; the slab group asks every slab, MgcSlabGroup::first_relabel_dt) */
    float ms = 0.f;
    MGC_HIP(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    h->stats.build_ms = ms;
    h->flow_const = h->h_scalar[0] + (h->d_tr_in ? h->flow_const_in : 0.0);
    /* the explicit edges stay with the handle: a rebuild (new markers, another sigma) applies them again; mgc_add_edges
     * after a build replaces them (one batch per build) */
    h->edges_applied = h->n_edges != 0;
    h->built = true;
    h->solved = false;
    h->labels_valid = false;
    h->op_spans.clear(); /* (launch-by-launch timing: the pairs of the solve before) */
    h->labels_on_host = false;
    return MGC_OK;
}

int mgc_maxflow(mgc_handle h, double* flow)
{
    if (!h) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_maxflow before mgc_build");
    if (h->nranks > 1 && !h->solved) return mgc_fail(h, MGC_ERR_STATE, "a slab of a multi-GPU volume is solved by the slab driver (mgc_solver_op / mgc_finish)");
    MGC_HIP(h, hipSetDevice(h->device));
    MgcLattice& L = h->L;
    if (!h->solved) {
        MgcRange range_("mgc_maxflow");
        MgcSolveStats st;
        MGC_HIP(h, hipEventRecord(h->ev[0], h->stream));
        HipDev dev;
        HipDev26 dev26;
        dev.h = dev26.h = h;
        h->timing_offset++;
        int rc;
        if (L.ndir == 6) {
            MgcSolveParams P = h->params;
            /* radial = 2: the flood phase runs on radial labels where there are walls to flood against (a closed surface of weak arcs
             * takes many cycles of exact labels to saturate: 35.9 -> 23.7 ms on the headline volume); on a weak-contrast volume, where
             * what leaves the source mostly reaches the sink, exact labels are the better guide (66 vs 104 ms at 512^3) */
            if (P.radial == 2) P.radial = h->wall_tiles >= h->radial_min_walls ? 1 : 0;
            h->repeat_now = (h->repeat_steps & (P.radial ? 2 : 1)) != 0;
            h->radial_cycle_no = 0;
            if (h->prepush && h->d_prob && !h->rounds_set) P.rounds_per_relabel = 2; /* (a pre-pushed graph, see the 26-neighbourhood branch; 512^3 + regional map: 19.0 ms at 3, 17.8 at 2, 20.1 at 4; without the pre-push 21.4) */
            rc = mgc_solve(dev, L, P, st);
        } else {
            /* A graph whose source -> u -> v -> sink paths k_build settled (regional term + pre-push) starts with a sixth of its tiles
             * active and most of their excess enclosed: it pays to look at the labels again after three colour rounds instead of six,
             * and the long visits of sparse phases (sweeps_sparse26) only shuffle excess that the next relabel declares dead.  Measured
             * at 512^3 (BASELINE config 3, profiles/r4_prepush_schedule.jsonl): 107.9 ms with the general defaults, 51.6 ms with these.
             * Only where the caller set neither knob. */
            MgcSolveParams P = h->params;
            const int sparse_before = h->sweeps_sparse26;
            if (h->prepush && h->d_prob) {
                if (!h->rounds_set) P.rounds_per_relabel = 3;
                if (!h->sparse26_set) h->sweeps_sparse26 = P.max_sweeps;
                /* ... and what is left to discharge are the heavy tiles (every voxel holds excess): there the one-wave-per-tile kernel,
                 * with ONE pass over the steps per sweep, is the faster form (config 3: 14.8 vs 19.6 ms of discharges; on graphs that
                 * were not pre-pushed it is the slower one, 97 vs 61 ms) */
                h->w26_auto = !h->wave_set && P.max_cycles < 0;
            }
            rc = mgc_solve(dev26, L, P, st, mgc_layout26());
            h->sweeps_sparse26 = sparse_before;
            h->w26_auto = false;
            dev.first_error = dev26.first_error;
            dev.spans.clear();
            for (const auto& sp : dev26.spans) dev.spans.push_back({sp.a, sp.b, sp.kind});
            dev.discharge_launches = dev26.discharge_launches; dev.relabel_launches = dev26.relabel_launches; dev.readbacks = dev26.readbacks;
            for (int k = 0; k < 4; ++k) { dev.timed[k] = dev26.timed[k]; dev.seen[k] = dev26.seen[k]; }
        }
        mgc_flush_zero(h);
        if (dev.first_error != hipSuccess)
            return mgc_fail(h, MGC_ERR_HIP, "solver: HIP error %s", hipGetErrorString(dev.first_error));
        if (rc) { /* the work counters of the truncated run stay readable (mgc_get_stats) */
            (void)hipStreamSynchronize(h->stream);
            dev.resolve_timing();
            h->stats.discharge_ms = dev.discharge_ms + dev.block_ms; h->stats.relabel_ms = dev.relabel_ms;
            h->stats.discharge_launches = dev.discharge_launches; h->stats.relabel_launches = dev.relabel_launches;
            h->stats.discharge_tiles = st.discharge_tiles; h->stats.relabel_tiles = st.relabel_tiles;
            h->stats.global_relabels = st.outer; h->stats.phases = st.phases;
            return mgc_fail(h, MGC_ERR_NOT_CONVERGED, "solver did not converge within %d global relabels", h->params.max_outer);
        }
        /* read-out: labels, then the capacity of the cut they define */
        { const int rc2 = mgc_launch_readout(h, 1, h->ev[1]); if (rc2) return rc2; }
        MGC_HIP(h, hipMemcpyAsync(h->h_scalar, h->d_scalar, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        MGC_HIP(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        MGC_HIP(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
        h->stats.solve_ms = ms;
        dev.resolve_timing();
        h->stats.discharge_ms = dev.discharge_ms + dev.block_ms;
        h->stats.discharge_wave_ms = dev.discharge_ms;
        h->stats.discharge_wave_launches = dev.seen[0];
        h->stats.timing_stride = h->timing ? h->timing_stride : 0;
        h->stats.relabel_ms = dev.relabel_ms;
        h->stats.discharge_launches = dev.discharge_launches;
        h->stats.relabel_launches = dev.relabel_launches;
        h->stats.reserved[0] = dev.readbacks;
        h->stats.reserved[1] = st.radial_cycles;
        h->stats.reserved[2] = h->wall_tiles;
        h->stats.discharge_tiles = st.discharge_tiles;
        h->stats.discharge_wave_tiles = L.ndir == 6 ? h->h_count[MGC_CNT_WAVE_TILES] : st.discharge_tiles; /* (as of the solve's last counter read-back) */
        h->stats.relabel_tiles = st.relabel_tiles;
        h->stats.global_relabels = st.outer;
        h->stats.phases = st.phases;
        h->flow = h->flow_const + h->h_scalar[1];
        h->solved = true;
        h->labels_on_host = false;
    }
    if (flow) *flow = h->flow;
    return MGC_OK;
}

int mgc_finish(mgc_handle h, double* flow_partial)
{
    if (!h) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_finish before mgc_build");
    MGC_HIP(h, hipSetDevice(h->device));
    mgc_flush_zero(h);
    MgcLattice& L = h->L;
    { const int rc = mgc_launch_readout(h, 1, nullptr); if (rc) return rc; }
    MGC_HIP(h, hipMemcpyAsync(h->h_scalar, h->d_scalar, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (!h->op_spans.empty()) MGC_HIP(h, hipMemcpyAsync(h->h_count, L.count, MGC_NCOUNT * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    if (!h->op_spans.empty()) { /* the solve was driven launch by launch (mgc_solver_op): its kernel times */
        double ms_kind[4] = {0.0, 0.0, 0.0, 0.0};
        int64_t n_kind[4] = {0, 0, 0, 0};
        for (const auto& sp : h->op_spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->op_ev[sp.a], h->op_ev[sp.b]) == hipSuccess) { ms_kind[sp.kind] += ms; n_kind[sp.kind]++; }
        }
        h->stats.discharge_wave_ms = ms_kind[0];
        h->stats.discharge_wave_launches = n_kind[0];
        h->stats.discharge_ms = ms_kind[0] + ms_kind[3];
        h->stats.discharge_launches = n_kind[0] + n_kind[3];
        h->stats.relabel_ms = ms_kind[1];
        h->stats.relabel_launches = n_kind[1];
        h->stats.timing_stride = 1;
        h->stats.discharge_tiles = h->h_count[L.ndir == 6 ? 8 : MGC26_CNT_DIS];
        h->stats.discharge_wave_tiles = L.ndir == 6 ? h->h_count[MGC_CNT_WAVE_TILES] : h->stats.discharge_tiles;
        h->op_spans.clear();
    }
    h->flow = h->flow_const + h->h_scalar[1];
    h->solved = true;
    h->labels_on_host = false;
    if (flow_partial) *flow_partial = h->flow;
    return MGC_OK;
}

int mgc_labels(mgc_handle h, uint8_t* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    if (!h->solved) return mgc_fail(h, MGC_ERR_STATE, "mgc_labels before mgc_maxflow");
    MGC_HIP(h, hipSetDevice(h->device));
    MGC_HIP(h, mgc_staged_copy(h, h->d_labels, out, (size_t)h->nvox, false));
    return MGC_OK;
}

int mgc_what_segment(mgc_handle h, int64_t i, int* segment)
{
    if (!h || !segment) return MGC_ERR_INVALID;
    if (i < 0 || i >= h->nvox) return mgc_fail(h, MGC_ERR_INVALID, "node id %lld out of range", (long long)i);
    if (!h->solved) { /* before maxflow() every node is free -> default segment SOURCE (graph.h:561-571) */
        *segment = MGC_SOURCE;
        return MGC_OK;
    }
    if (!h->labels_on_host) {
        if (!h->h_labels) h->h_labels = (uint8_t*)malloc((size_t)h->nvox);
        if (!h->h_labels) return mgc_fail(h, MGC_ERR_OOM, "host allocation failed");
        const int rc = mgc_labels(h, h->h_labels);
        if (rc) return rc;
        h->labels_on_host = true;
    }
    *segment = h->h_labels[i] ? MGC_SOURCE : MGC_SINK;
    return MGC_OK;
}

int mgc_get_nweights(mgc_handle h, int axis, double* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_get_nweights before mgc_build");
    if (axis < 0 || axis >= h->ndim) return mgc_fail(h, MGC_ERR_INVALID, "axis %d out of range", axis);
    MGC_HIP(h, hipSetDevice(h->device));
    const int a3 = 3 - h->ndim + axis;
    int64_t osh[3] = {h->shape[0], h->shape[1], h->shape[2]};
    osh[a3] -= 1;
    const int64_t n = osh[0] * osh[1] * osh[2];
    if (n <= 0) return MGC_OK;
    double* d = nullptr;
    MGC_HIP(h, mgc_dmalloc((void**)&d, (size_t)n * sizeof(double)));
    hipLaunchKernelGGL(k_get_nweights, dim3(1024), dim3(256), 0, h->stream, h->L, h->build_args, a3, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)mgc_dfree(d);
    MGC_HIP(h, e);
    return MGC_OK;
}

int mgc_get_nweights_offset(mgc_handle h, const int* offset, double* out)
{
    if (!h || !out || !offset) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_get_nweights_offset before mgc_build");
    int o[3] = {0, 0, 0};
    for (int k = 0; k < h->ndim; ++k) o[3 - h->ndim + k] = offset[k];
    const int nz = (o[0] != 0) + (o[1] != 0) + (o[2] != 0);
    for (int k = 0; k < 3; ++k)
        if (o[k] < -1 || o[k] > 1) return mgc_fail(h, MGC_ERR_INVALID, "offset components must be -1, 0 or 1");
    if (nz == 0 || (h->L.ndir == 6 && nz != 1)) return mgc_fail(h, MGC_ERR_INVALID, "offset is not a neighbour of this lattice");
    MGC_HIP(h, hipSetDevice(h->device));
    double* d = nullptr;
    MGC_HIP(h, mgc_dmalloc((void**)&d, (size_t)h->nvox * sizeof(double)));
    hipLaunchKernelGGL(k_get_nweights_offset, dim3(1024), dim3(256), 0, h->stream, h->L, h->build_args, o[0], o[1], o[2], d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)h->nvox * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)mgc_dfree(d);
    MGC_HIP(h, e);
    return MGC_OK;
}

int mgc_get_tweights(mgc_handle h, double* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_get_tweights before mgc_build");
    MGC_HIP(h, hipSetDevice(h->device));
    double* d = nullptr;
    MGC_HIP(h, mgc_dmalloc((void**)&d, (size_t)h->nvox * sizeof(double)));
    hipLaunchKernelGGL(k_untile_f64, dim3(1024), dim3(256), 0, h->stream, h->L, (const double*)h->d_tr0, (const uint8_t*)h->d_tflags, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)h->nvox * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)mgc_dfree(d);
    MGC_HIP(h, e);
    return MGC_OK;
}

int mgc_get_edge(mgc_handle h, int64_t i, int64_t j, double* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    if (!h->built) return mgc_fail(h, MGC_ERR_STATE, "mgc_get_edge before mgc_build");
    if (i < 0 || j < 0 || i >= h->nvox || j >= h->nvox || i == j) return mgc_fail(h, MGC_ERR_INVALID, "bad node pair");
    MGC_HIP(h, hipSetDevice(h->device));
    const MgcLattice& L = h->L;
    const int d = mgc_arc_direction(L, i, j);
    *out = 0.0; /* no such arc: get_edge returns 0 (graph.h:497) */
    if (d < 0) return MGC_OK;
    int tile, loc;
    mgc_node_to_tile(L, i, tile, loc);
    /* residual after maxflow() like Graph::get_edge (graph.h:482-498); before it the residuals ARE the built capacities */
    const double* src = L.rcap + ((int64_t)tile * L.ndir + d) * MGC_TV + loc;
    MGC_HIP(h, hipMemcpyAsync(h->h_scalar + 6, src, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MGC_HIP(h, hipStreamSynchronize(h->stream));
    *out = h->h_scalar[6];
    return MGC_OK;
}

int mgc_get_node_num(mgc_handle h, int64_t* n)
{
    if (!h || !n) return MGC_ERR_INVALID;
    *n = h->nvox;
    return MGC_OK;
}

int mgc_set_param(mgc_handle h, const char* name, int64_t value)
{
    if (!h || !name) return MGC_ERR_INVALID;
    if (!strcmp(name, "rounds_per_relabel") && value > 0) { h->params.rounds_per_relabel = (int)value; h->rounds_set = true; }
    else if (!strcmp(name, "max_cycles") && value != 0) h->params.max_cycles = (int)value; /* < 0 (26-neighbourhood): stored labels, no in-tile BFS */
    else if (!strcmp(name, "max_sweeps") && value > 0) h->params.max_sweeps = (int)value;
    else if (!strcmp(name, "max_outer") && value > 0) h->params.max_outer = (int)value;
    else if (!strcmp(name, "grid_cap") && value > 0) h->grid_cap = (int)value;
    else if (!strcmp(name, "grid26_dis") && value >= 0) h->grid26_dis = (int)value;
    else if (!strcmp(name, "relabel_batch") && value > 0) h->params.relabel_batch = (int)value;
    else if (!strcmp(name, "check_rounds") && value > 0) h->params.check_rounds = (int)value;
    else if (!strcmp(name, "stop_below") && value >= 0) h->params.stop_below = (int)value;
    else if (!strcmp(name, "incremental_relabel")) h->params.incremental_relabel = value != 0;
    else if (!strcmp(name, "trace")) h->params.trace = value != 0;
    else if (!strcmp(name, "adaptive_rounds") && value >= 0) h->params.adaptive_rounds = (int)value; /* 0 = off, k = threshold */
    else if (!strcmp(name, "radial") && value >= 0 && value <= 2) h->params.radial = (int)value;   /* flood phase on radial labels (mgc_dt_ops.inl): 0 never, 1 always, 2 when the graph holds walls */
    else if (!strcmp(name, "radial_min_walls") && value >= 0) h->radial_min_walls = (int)value;
    else if (!strcmp(name, "radial_budget_x16") && value >= 1) h->params.radial_budget_x16 = (int)value;
    else if (!strcmp(name, "radial_min_c") && value >= 1) h->params.radial_min_c = (int)value;
    else if (!strcmp(name, "radial_rounds0") && value >= 0) h->params.radial_rounds0 = (int)value; /* 0: one radial cycle of the whole budget */
    else if (!strcmp(name, "use_filters")) h->use_filters = (int)value;
    else if (!strcmp(name, "wave_kernels")) { h->wave_kernels = (int)value; h->wave_set = true; }
    else if (!strcmp(name, "wave_min_tiles") && value >= 0) h->wave_min_tiles = (int)value;
    else if (!strcmp(name, "sweeps_sparse26") && value >= 0) { h->sweeps_sparse26 = (int)value; h->sparse26_set = true; }
    else if (!strcmp(name, "wave_grid_dis") && value > 0) h->wave_grid_dis = (int)value;
    else if (!strcmp(name, "wave_grid26") && value > 0) h->wave_grid26 = (int)value;
    else if (!strcmp(name, "prepush")) h->prepush = value != 0;
    else if ((!strcmp(name, "exchange_passes") || !strcmp(name, "relabel_exchange_every")) && value > 0) h->params.exchange_passes = (int)value; /* slabs: relabel passes between two exchanges of the border labels */
    else if (!strcmp(name, "exchange_rounds") && value > 0) h->params.exchange_rounds = (int)value; /* slabs (6-neighbourhood): colour rounds between two exchanges of labels + outbox flow */
    else if (!strcmp(name, "w26_passes") && value > 0) h->w26_passes = (int)value;
    else if (!strcmp(name, "w26_raises") && value > 0) h->w26_raises = (int)value;
    else if (!strcmp(name, "w26_flags") && value >= 0) h->w26_flags = (int)value;
    else if (!strcmp(name, "wave_grid_rel") && value > 0) h->wave_grid_rel = (int)value;
    else if (!strcmp(name, "first_relabel_dt")) h->use_dt = value != 0;
    else if (!strcmp(name, "relabel_bricks")) h->use_bricks = value != 0;
    else if (!strcmp(name, "exact_sink_tiles") && value >= 0 && value <= 2) h->exact_sink_tiles = (int)value;
    else if (!strcmp(name, "sink_sweeps") && value > 0) h->sink_sweeps = (int)value;
    else if (!strcmp(name, "halo_max_records") && value >= 1) { /* record slots of a border message (all slabs of a volume alike!) */
        const int64_t T = (int64_t)h->L.gy * h->L.gx;
        h->L.halo_max_rec = (int)(value < T ? value : T);
    }
    else if (!strcmp(name, "wave_stagger") && value >= 0) h->wave_stagger = (int)value;
    else if (!strcmp(name, "repeat_steps") && value >= 0 && value <= 7) h->repeat_steps = (int)value;
    else if (!strcmp(name, "repeat_min_tiles") && value >= 0) h->repeat_min_tiles = (int)value;
    else if (!strcmp(name, "repeat_flood_min_tiles") && value >= 0) h->repeat_flood_min_tiles = (int)value;
    else if (!strcmp(name, "list_shards") && (value == 1 || value == MGC_NSHARD)) { /* regions per work list (MgcLattice::scount); between solves only */
        mgc_flush_zero(h);
        MGC_HIP(h, hipMemsetAsync(h->L.count, 0, MGC_NCOUNT * (1 + MGC_NSHARD) * sizeof(int32_t), h->stream));
        h->L.nshard = (int)value;
        h->filt[0] = h->filt[1] = 0;
        h->solved = false;
    }
    else if (!strcmp(name, "activate_exact_max") && value >= 0) h->activate_exact_max = (int)value;
    else if (!strcmp(name, "kernel_timing")) h->timing = value != 0;
    else if (!strcmp(name, "timing_stride") && value > 0) h->timing_stride = (int)value;
    else if (!strcmp(name, "profile_sections")) {
        if (value && !h->L.prof) {
            MGC_HIP(h, mgc_dmalloc((void**)&h->L.prof, 16 * sizeof(unsigned long long)));
        }
        if (h->L.prof) MGC_HIP(h, hipMemset(h->L.prof, 0, 16 * sizeof(unsigned long long)));
        if (!value && h->L.prof) { (void)mgc_dfree(h->L.prof); h->L.prof = nullptr; }
    }
    else return mgc_fail(h, MGC_ERR_INVALID, "unknown or invalid parameter %s=%lld", name, (long long)value);
    return MGC_OK;
}

int mgc_get_profile(mgc_handle h, uint64_t* out16)
{
    if (!h || !out16) return MGC_ERR_INVALID;
    memset(out16, 0, 16 * sizeof(uint64_t));
    if (!h->L.prof) return MGC_OK;
    MGC_HIP(h, hipSetDevice(h->device));
    MGC_HIP(h, hipMemcpy(out16, h->L.prof, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MGC_OK;
}

int mgc_get_stats(mgc_handle h, mgc_stats* out)
{
    if (!h || !out) return MGC_ERR_INVALID;
    h->stats.ntiles = h->L.ntiles;
    h->stats.nvox = h->nvox;
    h->stats.device_bytes = h->device_bytes;
    *out = h->stats;
    return MGC_OK;
}

} /* extern "C" */
