// Does a global f64 atomic add (no return) keep subnormals, and is it the IEEE sum?  (k26_discharge_w updates the idle neighbour
// tile's excess / residual with such atomics.)  hipcc --offload-arch=gfx950 -O2 atomic_f64_probe.hip -o atomic_f64_probe && ./atomic_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
__global__ void k(double* x, const double* add, int n)
{
    const int i = threadIdx.x;
    if (i < n) (void)__hip_atomic_fetch_add(&x[i], add[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main()
{
    const int n = 6;
    double x[n] = {0.0, 1e-310, 2.2250738585072014e-308, 1.0, 0.1, 4e-324};
    double a[n] = {1e-310, 1e-310, 1e-310, 1e-17, 0.2, 4e-324};
    double want[n], got[n];
    for (int i = 0; i < n; ++i) want[i] = x[i] + a[i];
    double *dx, *da;
    hipMalloc(&dx, sizeof(x)); hipMalloc(&da, sizeof(a));
    hipMemcpy(dx, x, sizeof(x), hipMemcpyHostToDevice); hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, da, n);
    hipMemcpy(got, dx, sizeof(x), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        uint64_t g, w; memcpy(&g, &got[i], 8); memcpy(&w, &want[i], 8);
        printf("%g + %g = %.17g (want %.17g) %s\n", x[i], a[i], got[i], want[i], g == w ? "ok" : "DIFFERENT");
        bad += g != w;
    }
    printf("atomic f64 add: %s\n", bad ? "NOT IEEE-identical" : "IEEE-identical incl. subnormals");
    return bad;
}
