/*
 * tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE ONLY (CPU test tier, `-m "not gpu"`).
 *
 * Executes the solver's single-source tile operations (medpy_amd/csrc/mgc_tile_ops.inl) and
 * schedule (mgc_driver.inl) on the host: a "block" is a loop over its 512 lanes, a kernel
 * launch is a loop over tiles.  This lets the algorithm be parity-tested against the BK
 * oracle without a GPU.  It is NOT a fallback: the product library (libmedpyhip.so) is built
 * from mgc_kernels.hip only, contains none of this, and fails loudly without a GPU.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../medpy_amd/csrc/mgc_tile_ops.inl"
#include "../../medpy_amd/csrc/mgc_driver.inl"

struct HostBlock {
    template <class T>
    struct Reg {
        T v[MGC_TV];
        T& operator[](int t) { return v[t]; }
    };
    MgcTileShared& S;
    explicit HostBlock(MgcTileShared& s) : S(s) {}
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
    int atomic_add(int32_t* p, int v) { int o = *p; *p += v; return o; }
    uint32_t atomic_exch(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
    void atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
    void atomic_and(uint32_t* p, uint32_t v) { *p &= v; }
};

struct HostDev {
    MgcLattice L;
    MgcTileShared S;
    void fill_heights_inf() { for (int64_t i = 0; i < (int64_t)L.ntiles * MGC_TV; ++i) L.height[i] = MGC_HINF; }
    void zero_count(int i) { L.count[i] = 0; }
    void read_counts(int* out) { memcpy(out, L.count, MGC_NCOUNT * sizeof(int)); }
    void absorb_all() { HostBlock x(S); for (int t = 0; t < L.ntiles; ++t) mgc_absorb_tile(x, L, t); }
    void relabel_all(uint32_t epoch, int next)
    {
        HostBlock x(S);
        for (int t = 0; t < L.ntiles; ++t) {
            if (L.status[t] & 2u) L.count[9]++;
            mgc_relabel_tile(x, L, t, epoch, next, true);
        }
    }
    void relabel_list(int lst, uint32_t epoch, int next)
    {
        HostBlock x(S);
        const int n = L.count[lst];
        L.count[9] += n;
        for (int i = 0; i < n; ++i) mgc_relabel_tile(x, L, L.list[lst][i], epoch, next, false);
    }
    void activate_all(uint32_t phase) { HostBlock x(S); for (int t = 0; t < L.ntiles; ++t) mgc_activate_tile(x, L, t, phase); }
    void discharge(int lst, uint32_t phase, int cycles, int sweeps)
    {
        HostBlock x(S);
        const int n = L.count[lst];
        L.count[8] += n;
        for (int i = 0; i < n; ++i) mgc_discharge_tile(x, L, L.list[lst][i], phase, cycles, sweeps);
    }
};

extern "C" {

/*
 * shape[3] = (D0, D1, D2); w[a] = forward n-link capacities along array axis a in the oracle's
 * per-axis layout (oracle/energy_numpy.py:boundary_weights, i.e. energy_voxel.py:644-658);
 * trcap[N] = merged t-link residual (graph.h:416-425).  labels_out[N]: 0 = sink side, 1 otherwise.
 * stats_out[8] = MgcSolveStats.  Returns 0 when converged.
 */
int hostsim_solve(const int64_t* shape, const double* w0, const double* w1, const double* w2, const double* trcap,
                  int rounds, int cycles, int sweeps, int max_outer, uint8_t* labels_out, int64_t* stats_out)
{
    HostDev dev;
    MgcLattice& L = dev.L;
    memset(&L, 0, sizeof(L));
    L.dz = shape[0]; L.dy = shape[1]; L.dx = shape[2];
    L.nvox = L.dz * L.dy * L.dx;
    L.gz = (int)((L.dz + 7) / 8); L.gy = (int)((L.dy + 7) / 8); L.gx = (int)((L.dx + 7) / 8);
    L.ntiles = L.gz * L.gy * L.gx;
    const int64_t nt = L.ntiles;
    std::vector<double> rcap(nt * 6 * MGC_TV, 0.0), excess(nt * MGC_TV, 0.0), sink(nt * MGC_TV, 0.0), obox(nt * 6 * MGC_TF, 0.0);
    std::vector<int32_t> height(nt * MGC_TV, MGC_HINF), lists(6 * nt, 0), count(MGC_NCOUNT, 0);
    std::vector<uint8_t> rmask(nt * MGC_TV, 0);
    std::vector<uint32_t> oflags(nt, 0), stamp(nt, 0), rstamp(nt, 0), status(nt, 0);
    L.rcap = rcap.data(); L.cap0 = NULL; L.excess = excess.data(); L.sink = sink.data(); L.height = height.data();
    L.rmask = rmask.data(); L.obox = obox.data(); L.oflags = oflags.data();
    for (int i = 0; i < 6; ++i) L.list[i] = lists.data() + i * nt;
    L.count = count.data(); L.stamp = stamp.data(); L.rstamp = rstamp.data(); L.status = status.data();

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
                sink[(int64_t)tile * MGC_TV + loc] = tr < 0 ? -tr : 0.0;
                if (tr < 0) { m |= MGC_MASK_SINK; status[tile] |= 2u; }
                rmask[(int64_t)tile * MGC_TV + loc] = (uint8_t)m;
            }

    MgcSolveParams P = mgc_default_params();
    if (rounds > 0) P.rounds_per_relabel = rounds;
    if (cycles > 0) P.max_cycles = cycles;
    if (sweeps > 0) P.max_sweeps = sweeps;
    if (max_outer > 0) P.max_outer = max_outer;
    MgcSolveStats st;
    const int rc = mgc_solve(dev, L, P, st);
    memcpy(stats_out, &st, sizeof(st));
    for (int64_t id = 0; id < L.nvox; ++id) {
        int tile, loc;
        mgc_node_to_tile(L, id, tile, loc);
        labels_out[id] = height[(int64_t)tile * MGC_TV + loc] < MGC_HINF ? 0 : 1;
    }
    return rc;
}

} /* extern "C" */
