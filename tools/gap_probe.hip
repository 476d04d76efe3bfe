// Launch-gap probe (MI355X): what sits between two dependent kernels of one stream costs how much?
//   hipcc --offload-arch=gfx950 -O3 -o tools/gap_probe tools/gap_probe.hip && tools/gap_probe
// Every case runs REPS dependent "busy" kernels (one workgroup spinning ~BUSY_US on s_memrealtime) on stream A and
// reports (elapsed / REPS - busy) = the gap per boundary.  Cases differ in what the host enqueues between two of them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void busy(unsigned long long ticks, int *sink) {  // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (ticks == 0xFFFFFFFFull) *sink = 1;
}
// publishes step `v` in *flag when it starts (agent scope), then spins like busy
__global__ void busy_publish(unsigned long long ticks, int *flag, int v) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
// spins like busy, then polls *flag >= v (bounded) before it ends
__global__ void busy_then_poll(unsigned long long ticks, const int *flag, int v, int *timeouts) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        if (spins >= (1 << 22)) atomicAdd(timeouts, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
// one workgroup: waits until *flag >= v (bounded), then ends — the "gate" of an off-chain stream
__global__ void gate(const int *flag, int v, int *timeouts) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        if (spins >= (1 << 22)) atomicAdd(timeouts, 1);
    }
}
__global__ void setflag(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main() {
    const int REPS = 64;
    const double BUSY_US = 20.0;
    const unsigned long long ticks = (unsigned long long)(BUSY_US * 100.0);
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    int *flags;
    CK(hipMalloc(&flags, 64 * sizeof(int)));
    std::vector<hipEvent_t> ev(4 * REPS + 8);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    auto report = [&](const char *name, float ms) { printf("%-78s gap %6.2f us per boundary\n", name, ms * 1e3 / REPS - BUSY_US); };
    for (int round = 0; round < 2; ++round) {  // round 0 warms everything up
        float ms;
        // (a) kernels only
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < REPS; ++i) hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, A, ticks, flags);
        CK(hipEventRecord(t1, A));
        CK(hipStreamSynchronize(A));
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (round) report("(a) kernel, kernel, ...", ms);
        // (b) an event record between them (nobody waits for it)
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < REPS; ++i) {
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, A, ticks, flags);
            CK(hipEventRecord(ev[i], A));
        }
        CK(hipEventRecord(t1, A));
        CK(hipStreamSynchronize(A));
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (round) report("(b) kernel, record, kernel, ...", ms);
        // (c) a wait for an event of stream B that completed long ago
        hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, B, flags + 1, 1);
        CK(hipEventRecord(ev[REPS], B));
        CK(hipStreamSynchronize(B));
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < REPS; ++i) {
            CK(hipStreamWaitEvent(A, ev[REPS], 0));
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, A, ticks, flags);
        }
        CK(hipEventRecord(t1, A));
        CK(hipStreamSynchronize(A));
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (round) report("(c) wait(old event of B), kernel, ...", ms);
        // (d) record on A, B waits and runs a short kernel (off chain); A continues with kernels only
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < REPS; ++i) {
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, A, ticks, flags);
            CK(hipEventRecord(ev[i], A));
            CK(hipStreamWaitEvent(B, ev[i], 0));
            hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, B, flags + 2, i);
        }
        CK(hipEventRecord(t1, A));
        CK(hipStreamSynchronize(A));
        CK(hipStreamSynchronize(B));
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (round) report("(d) kernel, record [B waits, short kernel on B], kernel, ...", ms);
        // (e) the round-2 pattern: A: kernel -> record; B: wait, short kernel, record; A: wait -> kernel
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < REPS; ++i) {
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, A, ticks, flags);
            CK(hipEventRecord(ev[2 * i], A));
            CK(hipStreamWaitEvent(B, ev[2 * i], 0));
            hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, B, flags + 2, i);
            CK(hipEventRecord(ev[2 * i + 1], B));
            CK(hipStreamWaitEvent(A, ev[2 * i + 1], 0));
        }
        CK(hipEventRecord(t1, A));
        CK(hipStreamSynchronize(A));
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (round) report("(e) kernel, record, [B: wait, short kernel, record], wait, kernel  (B's kernel ON the chain)", ms);
        // (f) device flags: A = kernels only (each publishes its step at its start and polls B's flag at its end);
        //     B = gate(A reached step i) -> short kernel -> setflag(i + 1): no event anywhere
        CK(hipMemsetAsync(flags, 0, 64 * sizeof(int), A));
        CK(hipStreamSynchronize(A));
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < REPS; ++i) {
            hipLaunchKernelGGL(busy_publish, dim3(1), dim3(64), 0, A, 0ull, flags + 8, i + 1);        // "diag(j) start": publish
            hipLaunchKernelGGL(busy_then_poll, dim3(1), dim3(64), 0, A, ticks, flags + 16, i + 1, flags + 24);  // ends once B's step i+1 is done
            hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, B, flags + 8, i + 1, flags + 24);
            hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, B, flags + 16, i + 1);
        }
        CK(hipEventRecord(t1, A));
        CK(hipStreamSynchronize(A));
        CK(hipStreamSynchronize(B));
        CK(hipEventElapsedTime(&ms, t0, t1));
        int timeouts = 0;
        CK(hipMemcpy(&timeouts, flags + 24, sizeof(int), hipMemcpyDeviceToHost));
        if (round) {
            report("(f) A: [publish kernel, kernel that polls B's flag]; B: [gate kernel, setflag]  (2 boundaries)", ms);
            printf("    (f) spin timeouts: %d\n", timeouts);
        }
        // (g) as (f), but B's hand-backs are stream memory operations of the command processor instead of one-lane kernels:
        //     g1: setflag -> hipStreamWriteValue32;  g2: also gate -> hipStreamWaitValue32 (>=)
        for (int mode = 1; mode <= 2; ++mode) {
            CK(hipMemsetAsync(flags, 0, 64 * sizeof(int), A));
            CK(hipStreamSynchronize(A));
            CK(hipEventRecord(t0, A));
            hipError_t err = hipSuccess;
            for (int i = 0; i < REPS && err == hipSuccess; ++i) {
                hipLaunchKernelGGL(busy_publish, dim3(1), dim3(64), 0, A, 0ull, flags + 8, i + 1);
                hipLaunchKernelGGL(busy_then_poll, dim3(1), dim3(64), 0, A, ticks, flags + 16, i + 1, flags + 24);
                if (mode == 2)
                    err = hipStreamWaitValue32(B, flags + 8, i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
                else
                    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, B, flags + 8, i + 1, flags + 24);
                if (err == hipSuccess) err = hipStreamWriteValue32(B, flags + 16, i + 1, 0);
            }
            if (err != hipSuccess) {
                printf("(g%d) stream memory op refused: %s\n", mode, hipGetErrorString(err));
                hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, B, flags + 16, REPS + 1);  // let A drain
                CK(hipStreamSynchronize(A));
                CK(hipStreamSynchronize(B));
                continue;
            }
            CK(hipEventRecord(t1, A));
            CK(hipStreamSynchronize(A));
            CK(hipStreamSynchronize(B));
            CK(hipEventElapsedTime(&ms, t0, t1));
            int to2 = 0;
            CK(hipMemcpy(&to2, flags + 24, sizeof(int), hipMemcpyDeviceToHost));
            if (round) {
                report(mode == 1 ? "(g1) as (f) with hipStreamWriteValue32 in place of the setflag kernel  (2 boundaries)"
                                 : "(g2) ... and hipStreamWaitValue32 in place of the gate kernel  (2 boundaries)", ms);
                printf("    (g%d) spin timeouts: %d\n", mode, to2);
            }
        }
    }
    return 0;
}
