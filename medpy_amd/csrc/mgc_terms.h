/*
 * mgc_terms.h -- arithmetic of the voxel boundary terms (reference medpy/graphcut/energy_voxel.py:68-664), shared by
 * the lattice builder (mgc_kernels.hip:k_build) and the n-D edge generator of the sparse-graph path (msg_sparse.hip).
 */
#ifndef MGC_TERMS_H
#define MGC_TERMS_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "../../include/medpy_hip.h"

#define MGC_DBL_MIN 2.2250738585072014e-308 /* sys.float_info.min, energy_voxel.py:113,...,513 */

__device__ __forceinline__ double mgc_load_as_double(const void* p, int dtype, int64_t i, bool take_abs)
{
    double v;
    switch (dtype) {
    case MGC_U8: v = (double)((const uint8_t*)p)[i]; break;
    case MGC_I8: { int x = ((const int8_t*)p)[i]; v = (double)(take_abs && x < 0 ? -x : x); take_abs = false; } break;
    case MGC_U16: v = (double)((const uint16_t*)p)[i]; break;
    case MGC_I16: { int x = ((const int16_t*)p)[i]; v = (double)(take_abs && x < 0 ? -x : x); take_abs = false; } break;
    case MGC_U32: v = (double)((const uint32_t*)p)[i]; break;
    case MGC_I32: { int64_t x = ((const int32_t*)p)[i]; v = (double)(take_abs && x < 0 ? -x : x); take_abs = false; } break;
    case MGC_U64: v = (double)((const uint64_t*)p)[i]; break;
    case MGC_I64: { int64_t x = ((const int64_t*)p)[i]; v = (double)(take_abs && x < 0 ? -x : x); take_abs = false; } break;
    case MGC_F32: v = (double)((const float*)p)[i]; break;
    default: v = ((const double*)p)[i]; break;
    }
    return take_abs ? fabs(v) : v;
}

/* g(.) of the eight boundary terms, operation for operation as NumPy evaluates them in the
 * reference (energy_voxel.py:103-114, 226-236, 337-345, 444-452 and the difference twins). */
__device__ __forceinline__ double mgc_boundary_g(int term, double a, double b, double p0, const double* lut = nullptr, int lut_n = 0)
{
    const bool use_max = (term == MGC_TERM_MAXIMUM_LINEAR || term == MGC_TERM_MAXIMUM_EXPONENTIAL ||
                          term == MGC_TERM_MAXIMUM_POWER); /* MAXIMUM_DIVISION: difference skeleton, :347 */
    double x = use_max ? fmax(a, b) : fabs(a - b);
    if (lut) { /* integer-valued image: the host's own evaluation of the term, by table (mgc_set_boundary_lut) */
        const int i = (int)x;
        if (x >= 0.0 && x < (double)lut_n && (double)i == x) return lut[i];
    }
    switch (term) {
    case MGC_TERM_DIFFERENCE_LINEAR:
    case MGC_TERM_MAXIMUM_LINEAR:
        x = x / p0;
        x = 1.0 - x;
        if (x == 0.0) x = MGC_DBL_MIN;
        return x;
    case MGC_TERM_DIFFERENCE_EXPONENTIAL:
    case MGC_TERM_MAXIMUM_EXPONENTIAL:
        x = x * x;   /* numpy.power(x, 2) */
        x = x / p0;  /* /= math.pow(sigma, 2) */
        x = -x;      /* *= -1 */
        x = exp(x);
        if (x <= 0.0) x = MGC_DBL_MIN;
        return x;
    case MGC_TERM_DIFFERENCE_DIVISION:
    case MGC_TERM_MAXIMUM_DIVISION:
        x = x / p0;
        x = 1.0 / (x + 1.0);
        if (x <= 0.0) x = MGC_DBL_MIN;
        return x;
    default: /* power */
        x = 1.0 / (x + 1.0);
        x = pow(x, p0);
        if (x <= 0.0) x = MGC_DBL_MIN;
        return x;
    }
}

static size_t mgc_dtype_size(int dt)
{
    switch (dt) {
    case MGC_U8: case MGC_I8: return 1;
    case MGC_U16: case MGC_I16: return 2;
    case MGC_U32: case MGC_I32: case MGC_F32: return 4;
    case MGC_U64: case MGC_I64: case MGC_F64: return 8;
    default: return 0;
    }
}

/* float(abs(image.max() - image.min())) evaluated in the image's own dtype, as NumPy scalars do
 * (energy_voxel.py:174-176): float32 subtracts in float32, integers wrap at their width. */
static double mgc_range_in_dtype(double mn, double mx, int dtype)
{
    switch (dtype) {
    case MGC_F32: return (double)fabsf((float)mx - (float)mn);
    case MGC_I8:  { int8_t d = (int8_t)((int64_t)mx - (int64_t)mn); d = (int8_t)(d < 0 ? -d : d); return (double)d; }
    case MGC_I16: { int16_t d = (int16_t)((int64_t)mx - (int64_t)mn); d = (int16_t)(d < 0 ? -d : d); return (double)d; }
    case MGC_I32: { int32_t d = (int32_t)((int64_t)mx - (int64_t)mn); d = (int32_t)(d < 0 ? -(int64_t)d : d); return (double)d; }
    default: return fabs(mx - mn); /* unsigned: max >= min never wraps; f64; 64-bit ints via double */
    }
}


/* min / max / max|.| of the image, one partial triple per block (deterministic), finished on the host */
static __global__ __launch_bounds__(256) void k_minmax(const void* image, int dtype, int64_t n, double* part)
{
    __shared__ double smin[256], smax[256], sabs[256];
    double mn = INFINITY, mx = -INFINITY, ma = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = mgc_load_as_double(image, dtype, i, false);
        mn = fmin(mn, v);
        mx = fmax(mx, v);
        ma = fmax(ma, fabs(v));
    }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx; sabs[threadIdx.x] = ma;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
            sabs[threadIdx.x] = fmax(sabs[threadIdx.x], sabs[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = smin[0];
        part[blockIdx.x * 3 + 1] = smax[0];
        part[blockIdx.x * 3 + 2] = sabs[0];
    }
}

/* min, max and max|.| of a device image; `scratch` holds at least 3072 doubles */
static inline hipError_t mgc_image_range(const void* image, int dtype, int64_t n, double* scratch, hipStream_t stream, double* mn_out,
                                         double* mx_out, double* ma_out)
{
    const int nb = 1024;
    hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, stream, image, dtype, n, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    double* hp = (double*)malloc(nb * 3 * sizeof(double));
    if (!hp) return hipErrorOutOfMemory;
    e = hipMemcpyAsync(hp, scratch, nb * 3 * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    double mn = INFINITY, mx = -INFINITY, ma = 0.0;
    for (int b = 0; b < nb; ++b) { mn = fmin(mn, hp[3 * b]); mx = fmax(mx, hp[3 * b + 1]); ma = fmax(ma, hp[3 * b + 2]); }
    free(hp);
    *mn_out = mn; *mx_out = mx; *ma_out = ma;
    return e;
}

#endif /* MGC_TERMS_H */
