// Calibration microbenchmark (not part of the product): back-to-back v_mfma_f64_16x16x4_f64 from
// registers, 16 independent accumulators per wave, W waves per SIMD on every CU.  Prints the
// sustained TFLOP/s, i.e. the ceiling the Cholesky panel kernel can be priced against on THIS chip
// at the clock it holds under fp64 matrix load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void spin(double *out, int iters, double a0, double b0) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        int grid = 256 * wgs_per_cu;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        spin<<<grid, 256>>>(out, 100, 1.0, 0.5);
        (void)hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            spin<<<grid, 256>>>(out, iters, 1.0, 0.5);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 /*waves*/ * iters * 16.0 * 2048.0;
            printf("waves/SIMD=%d rep=%d  %.3f ms  %.2f TFLOP/s\n", wgs_per_cu, rep, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
