// Small device utilities of the C ABI that keep host-side glue free of array arithmetic:
// the mixture-of-Gaussians moments of tree_gps.py:116-131 and a strided device copy.
#include "common.h"

namespace bark {
namespace {

// partial[0][c] = sum_b mu[b][c],  partial[1][c] = sum_b (var[b][c] + mu[b][c]^2)   (b ascending: reproducible)
__global__ void mixture_partial_kernel(const double *__restrict__ mu, const double *__restrict__ var, int64_t B, int64_t C,
                                       double *__restrict__ partial) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        const double m = mu[b * C + c];
        s1 += m;
        s2 += var[b * C + c] + m * m;
    }
    partial[c] = s1;
    partial[C + c] = s2;
}

// tree_gps.py:128-130: mean = S1 / total, var = S2 / total - mean^2
__global__ void mixture_finish_kernel(const double *__restrict__ partial, double total, int64_t C, double *__restrict__ mean,
                                      double *__restrict__ var) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = partial[c] / total;
    mean[c] = m;
    var[c] = partial[C + c] / total - m * m;
}

}  // namespace
}  // namespace bark

using namespace bark;

extern "C" {

int bark_mixture_partial_hip(const double *mu, const double *var, int64_t B, int64_t C, double *partial, void *stream) {
    error_buffer()[0] = 0;
    if (!mu || !var || !partial || B < 1 || C < 1) return fail(BARK_ERR_ARG, "bark_mixture_partial_hip: bad argument");
    hipLaunchKernelGGL(mixture_partial_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mu,
                       var, B, C, partial);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int bark_mixture_finish_hip(const double *partial, double total, int64_t C, double *mean, double *var, void *stream) {
    error_buffer()[0] = 0;
    if (!partial || !mean || !var || C < 1 || !(total > 0.0)) return fail(BARK_ERR_ARG, "bark_mixture_finish_hip: bad argument");
    hipLaunchKernelGGL(mixture_finish_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       partial, total, C, mean, var);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int bark_copy2d_hip(double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int64_t cols, void *stream) {
    error_buffer()[0] = 0;
    if (!dst || !src || rows < 1 || cols < 1 || ldd < cols || lds < cols) return fail(BARK_ERR_ARG, "bark_copy2d_hip: bad argument");
    BARK_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(double), src, (size_t)lds * sizeof(double), (size_t)cols * sizeof(double),
                                    (size_t)rows, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return BARK_OK;
}

}  // extern "C"
