// Power-model microbenchmark (not part of the product): what a Cholesky row-kernel k-tile costs in ENERGY, piece by piece.
// Every mode runs the row kernel's inner loop shape — 4 waves x 64 v_mfma_f64_16x16x4_f64 per k-tile, 16 accumulators —
// on RANDOM operands (MFMA power depends on the data: tools/mfma_f64_peak.hip multiplies near-constant values), two
// workgroups per CU on every CU:
//   mode 0   operands in registers
//   mode 1   + the 32 ds_read_b64 per wave and k-tile from a (static) LDS stage image, as mma_stage reads them
//   mode 2   + the k-tile's 32 KiB staged by global_load_lds from a window that stays in L2 (every workgroup reads the same 1 MiB)
//   mode 3   + ... from HBM (every workgroup streams its own region)
//   mode 4   mode 3 with the per-k-tile barrier + vmcnt(0) wait of the real pipeline
//   mode 5   mode 4 with the A rows from the L2 window and only the B rows from HBM — what row_kernel does (a block row's tiles share
//            their A panel through the XCD's L2, every tile streams its own B panel)
//   mode 7   mode 5 with the B rows 4112 doubles apart — the row stride of the N = 4096 workspace (every 1 KiB row in a DRAM page
//            of its own) — instead of back to back
//   mode 6   mode 5 with the B region shared by a PAIR of workgroups of one XCD (ids i, i + 8): the traffic of two block rows
//            computed together (tiles (j, i) and (j+1, i) need the same B panel)
// Run under tools/power_probe.py: energy per flop = (socket power - idle power) / TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/power_model.hip -o /tmp/power_model && /tmp/power_model <mode> [seconds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 128, BK = 16, LDS_LD = NB + 16, STAGE = 2 * BK * LDS_LD, THREADS = 256;
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

__device__ __forceinline__ double rnd(unsigned &s) {  // uniform in (-1, 1)
    s = s * 1664525u + 1013904223u;
    return (double)(int)s * (1.0 / 2147483648.0);
}

template <int MODE>
__global__ __launch_bounds__(THREADS, 2) void ktile_loop(double *out, const double *src, long wg_stride, int ktiles_per_pass, int passes) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4, wr = wave >> 1, wc = wave & 1;
    unsigned seed = 1234567u + tid * 7919u + blockIdx.x * 104729u;
    for (int e = tid; e < 2 * STAGE; e += THREADS) lds[e] = rnd(seed);
    __syncthreads();
    f64x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) acc[i][k] = (f64x4){0, 0, 0, 0};
    double ra[4], rb[4];
    for (int i = 0; i < 4; ++i) ra[i] = rnd(seed), rb[i] = rnd(seed);
    // mode 6: workgroups i and i + 8 (same XCD, dispatched together) share a region
    const double *mine = src + (size_t)(MODE == 6 ? ((blockIdx.x >> 4) * 8 + (blockIdx.x & 7) + 1) : blockIdx.x + (MODE == 5 || MODE == 7 ? 1 : 0)) * wg_stride;
    for (int ps = 0; ps < passes; ++ps)
        for (int kt = 0; kt < ktiles_per_pass; ++kt) {
            const double *st = lds + (kt & 1) * STAGE;
            if (MODE >= 2) {  // stage_dma: wave w moves rows w, w+4, w+8, w+12 of both operands of the NEXT k-tile
                double *dst = lds + ((kt + 1) & 1) * STAGE;
                const double *a = mine + ((size_t)(MODE == 2 ? (kt & 31) : kt) * 2 * BK + wave) * NB + lane * 2;  // mode 2: a 1 MiB window
                const double *al = MODE >= 5 ? src + ((size_t)(kt & 31) * 2 * BK + wave) * NB + lane * 2 : a;        // modes 5, 6: A from the L2 window
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    __builtin_amdgcn_global_load_lds((glb_ptr_t *)(al + (size_t)(4 * p) * NB), (lds_ptr_t *)(dst + (wave + 4 * p) * LDS_LD), 16, 0, 0);
                    const double *bsrc = MODE == 7 ? mine + ((size_t)kt * BK + wave + 4 * p) * 4112 + lane * 2 : a + (size_t)(BK + 4 * p) * NB;
                    __builtin_amdgcn_global_load_lds((glb_ptr_t *)bsrc, (lds_ptr_t *)(dst + (BK + wave + 4 * p) * LDS_LD), 16, 0, 0);
                }
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a0, a1, a2, a3, b0, b1, b2, b3;
                if (MODE >= 1) {
                    const double *ar = st + (kk * 4 + lk) * LDS_LD + wr * 64 + lr;
                    const double *br = st + BK * LDS_LD + (kk * 4 + lk) * LDS_LD + wc * 64 + lr;
                    a0 = ar[0], a1 = ar[16], a2 = ar[32], a3 = ar[48];
                    b0 = br[0], b1 = br[16], b2 = br[32], b3 = br[48];
                } else {  // registers; rotate so consecutive MFMAs see different values
                    a0 = ra[kk], a1 = ra[(kk + 1) & 3], a2 = ra[(kk + 2) & 3], a3 = ra[(kk + 3) & 3];
                    b0 = rb[(kk + 1) & 3], b1 = rb[(kk + 2) & 3], b2 = rb[(kk + 3) & 3], b3 = rb[kk];
                }
#define M(mt, nt, av, bv) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[mt][nt], 0, 0, 0)
                M(0, 0, a0, b0); M(0, 1, a0, b1); M(0, 2, a0, b2); M(0, 3, a0, b3);
                M(1, 0, a1, b0); M(1, 1, a1, b1); M(1, 2, a1, b2); M(1, 3, a1, b3);
                M(2, 0, a2, b0); M(2, 1, a2, b1); M(2, 2, a2, b2); M(2, 3, a2, b3);
                M(3, 0, a3, b0); M(3, 1, a3, b1); M(3, 2, a3, b2); M(3, 3, a3, b3);
#undef M
            }
            if (MODE >= 4) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s = 0;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) s += acc[i][k][0] + acc[i][k][1] + acc[i][k][2] + acc[i][k][3];
    out[(size_t)blockIdx.x * THREADS + tid] = s;
}

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    const int grid = 512, ktiles = 128;  // one pass = a K = 2048 panel pair: 128 k-tiles x 32 KiB = 4 MiB per workgroup
    const size_t pass_doubles = mode == 7 ? (size_t)ktiles * BK * 4112 : (size_t)ktiles * 2 * BK * NB;
    // mode 2: every workgroup reads the first 1 MiB of one region; modes 3, 4: its own 4 MiB region of a 2 GiB buffer
    const bool hbm = mode >= 3;
    const size_t regions = hbm ? grid + 1 : 1;  // region 0 doubles as the L2 window of modes 5, 6
    double *src = nullptr, *out = nullptr;
    CK(hipMalloc(&src, regions * pass_doubles * sizeof(double)));
    CK(hipMalloc(&out, (size_t)grid * THREADS * sizeof(double)));
    {
        std::vector<double> h(pass_doubles);
        unsigned s = 99u;
        for (auto &v : h) {
            s = s * 1664525u + 1013904223u;
            v = (double)(int)s * (1.0 / 2147483648.0);
        }
        for (size_t r = 0; r < regions; ++r) CK(hipMemcpy(src + r * pass_doubles, h.data(), pass_doubles * sizeof(double), hipMemcpyHostToDevice));
    }
    const size_t lds_bytes = (size_t)2 * STAGE * sizeof(double);
    void (*fn)(double *, const double *, long, int, int) =
        mode == 0 ? ktile_loop<0> : mode == 1 ? ktile_loop<1> : mode == 2 ? ktile_loop<2> : mode == 3 ? ktile_loop<3> : mode == 4 ? ktile_loop<4> : mode == 5 ? ktile_loop<5> : mode == 6 ? ktile_loop<6> : ktile_loop<7>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const long wg_stride = hbm ? (long)pass_doubles : 0;
    const int passes = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), lds_bytes, 0, out, src, wg_stride, ktiles, 1);
    CK(hipDeviceSynchronize());
    double total_ms = 0, flops = 0;
    int launches = 0;
    while (total_ms < seconds * 1e3) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), lds_bytes, 0, out, src, wg_stride, ktiles, passes);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        launches += 8;
        flops += 8.0 * grid * 4 * (double)passes * ktiles * 64 * 2048.0;
    }
    const double bytes = mode >= 2 ? (double)launches * grid * passes * ktiles * 32768.0 : 0.0;  // staged into LDS; from HBM: all of it (3, 4), half (5), a quarter (6)
    printf("mode %d: %.2f TFLOP/s, staged %.2f TB/s, %.1f ms per launch\n", mode, flops / total_ms / 1e9, bytes / total_ms / 1e9, total_ms / launches);
    return 0;
}
