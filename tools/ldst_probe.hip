// development aid: the memory phases of k_discharge_w in isolation.  One wave per tile loads 8 planes of 4 KiB (8-byte or
// 16-byte per lane), spins for `work` cycles (the sweeps), stores 8 planes; 2048-wave persistent grid over `n` tiles from a
// 262144-tile arena.  hipcc --offload-arch=gfx950 -O3 tools/ldst_probe.hip -o /tmp/ldst_probe && /tmp/ldst_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int W> /* W = doubles per lane per access: 1 or 2 */
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe(double* arena, const int* list, int n, int* ticket, int work)
{
    const int l = threadIdx.x;
    for (int i = blockIdx.x; i < n;) {
        double* t = arena + (long)__builtin_amdgcn_readfirstlane(list[i]) * 4096;
        double v[64];
        if (W == 1) {
#pragma unroll
            for (int k = 0; k < 64; ++k) v[k] = t[k * 64 + l];
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) { const double2 d = *(const double2*)(t + k * 128 + l * 2); v[2 * k] = d.x; v[2 * k + 1] = d.y; }
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) acc += v[k];
        const long t0 = __builtin_readcyclecounter();
        while ((long)__builtin_readcyclecounter() - t0 < work) acc += 1e-300;
        if (W == 1) {
#pragma unroll
            for (int k = 0; k < 64; ++k) t[k * 64 + l] = v[k] + acc;
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) *(double2*)(t + k * 128 + l * 2) = make_double2(v[2 * k] + acc, v[2 * k + 1] + acc);
        }
        int nx = 0;
        if (l == 0) nx = atomicAdd(ticket, 1);
        i = (int)gridDim.x + __builtin_amdgcn_readfirstlane(nx);
    }
}
int main(int argc, char** argv)
{
    const int ntiles = 262144, n = argc > 1 ? atoi(argv[1]) : 5375, launches = 150;
    double* arena; int *list, *ticket;
    CK(hipMalloc(&arena, (size_t)ntiles * 4096 * 8)); CK(hipMemset(arena, 0, (size_t)ntiles * 4096 * 8));
    std::vector<int> h(n); srand(1); for (int i = 0; i < n; ++i) h[i] = (int)(((long)rand() * 7919 + i * 48) % ntiles);
    CK(hipMalloc(&list, n * 4)); CK(hipMemcpy(list, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMalloc(&ticket, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int work : {0, 20000, 50000})
        for (int w = 1; w <= 2; ++w) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(a));
                for (int k = 0; k < launches; ++k) {
                    CK(hipMemsetAsync(ticket, 0, 4));
                    if (w == 1) hipLaunchKernelGGL(k_probe<1>, dim3(2048), dim3(64), 0, 0, arena, list, n, ticket, work);
                    else hipLaunchKernelGGL(k_probe<2>, dim3(2048), dim3(64), 0, 0, arena, list, n, ticket, work);
                }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            const double bytes = (double)launches * n * 4096 * 8 * 2;
            printf("work %6d cycles, %2d B/lane: %.3f ms per launch of %d tiles, %.2f TB/s (load + store of 64 KiB per tile)\n", work, 8 * w,
                   best / launches, n, bytes / (best * 1e-3) / 1e12);
        }
    return 0;
}
