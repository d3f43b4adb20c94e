// development aid: is the tile load/store phase TLB-bound?  SoA (8 big arrays) vs AoS (one record per tile)
// hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o /tmp/tlb_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static const int REC = 8 * 4096 + 2048 + 512 + 3072 + 1408; // 39808 -> pad to 40960
__global__ __launch_bounds__(512) void k_soa(double* rc, double* ex, double* sk, int* hg, unsigned char* rm, const int* list, int n)
{
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const long tile = list[i];
        const int t = threadIdx.x;
        double acc = ex[tile * 512 + t] + sk[tile * 512 + t];
        double r[6];
        for (int d = 0; d < 6; ++d) r[d] = rc[(tile * 6 + d) * 512 + t];
        for (int d = 0; d < 6; ++d) acc += r[d];
        __syncthreads();
        ex[tile * 512 + t] = acc; sk[tile * 512 + t] = acc * 0.5;
        for (int d = 0; d < 6; ++d) rc[(tile * 6 + d) * 512 + t] = r[d] + 1.0;
        hg[tile * 512 + t] = (int)acc; rm[tile * 512 + t] = (unsigned char)t;
        __syncthreads();
    }
}
__global__ __launch_bounds__(512) void k_aos(char* arena, const int* list, int n)
{
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        char* rec = arena + (long)list[i] * 40960;
        double* rc = (double*)rec; double* ex = (double*)(rec + 24576); double* sk = (double*)(rec + 28672);
        int* hg = (int*)(rec + 32768); unsigned char* rm = (unsigned char*)(rec + 34816);
        const int t = threadIdx.x;
        double acc = ex[t] + sk[t];
        double r[6];
        for (int d = 0; d < 6; ++d) r[d] = rc[d * 512 + t];
        for (int d = 0; d < 6; ++d) acc += r[d];
        __syncthreads();
        ex[t] = acc; sk[t] = acc * 0.5;
        for (int d = 0; d < 6; ++d) rc[d * 512 + t] = r[d] + 1.0;
        hg[t] = (int)acc; rm[t] = (unsigned char)t;
        __syncthreads();
    }
}
int main()
{
    const long NT = 262144; const int n = 7000;
    double *rc, *ex, *sk; int* hg; unsigned char* rm; char* arena; int* dl;
    CK(hipMalloc(&rc, NT * 6 * 4096)); CK(hipMalloc(&ex, NT * 4096)); CK(hipMalloc(&sk, NT * 4096)); CK(hipMalloc(&hg, NT * 2048)); CK(hipMalloc(&rm, NT * 512));
    CK(hipMalloc(&arena, NT * 40960)); CK(hipMalloc(&dl, n * sizeof(int)));
    CK(hipMemset(rc, 0, NT * 6 * 4096)); CK(hipMemset(ex, 0, NT * 4096)); CK(hipMemset(sk, 0, NT * 4096)); CK(hipMemset(arena, 0, NT * 40960));
    std::vector<int> list(n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) { // 0: random tiles, 1: contiguous tiles, 2: random but sorted
        srand(1);
        for (int i = 0; i < n; ++i) list[i] = mode == 1 ? 100000 + i : (int)(((long)rand() * 7919 + rand()) % NT);
        if (mode == 2) std::sort(list.begin(), list.end());
        CK(hipMemcpy(dl, list.data(), n * sizeof(int), hipMemcpyHostToDevice));
        for (int lay = 0; lay < 2; ++lay)
            for (int grid : {1024, 4096}) {
                float best = 1e9, first = 0;
                for (int rep = 0; rep < 5; ++rep) {
                    if (mode == 0) { /* fresh random tiles every repetition: cold TLB / caches */
                        for (int i = 0; i < n; ++i) list[i] = (int)(((long)rand() * 7919 + rand()) % NT);
                        CK(hipMemcpy(dl, list.data(), n * sizeof(int), hipMemcpyHostToDevice));
                    }
                    CK(hipEventRecord(a));
                    if (lay == 0) hipLaunchKernelGGL(k_soa, dim3(grid), dim3(512), 0, 0, rc, ex, sk, hg, rm, dl, n);
                    else hipLaunchKernelGGL(k_aos, dim3(grid), dim3(512), 0, 0, arena, dl, n);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; if (rep == 0) first = ms;
                }
                const double bytes = (double)n * (2 * 8 * 4096 + 2048 + 512);
                printf("mode %d (%s) layout %s grid %4d: best %.3f ms  %.1f GB/s   first %.3f ms\n", mode, mode == 0 ? "random-fresh" : mode == 1 ? "contiguous" : "sorted",
                       lay ? "AoS" : "SoA", grid, best, bytes / best / 1e6, first);
            }
    }
    return 0;
}
