// What does this ROCm do when stream capture ends while a stream that was forked into the capture has not been joined back?
// (Round 3 saw `Fatal Python error: Segmentation fault` in torch's capture_end on a build whose sweep left a helper stream
// forked.)  Each case runs in a child process so that a crash is reported, not suffered.
//   hipcc --offload-arch=gfx950 -O2 tools/capture_unjoined_probe.hip -o /tmp/capture_probe && /tmp/capture_probe
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

__global__ void touch(int *p) { atomicAdd(p, 1); }

static int run_case(int which) {
    int *d;
    hipStream_t a, b;
    hipEvent_t fork_ev, join_ev;
    if (hipMalloc(&d, 4) != hipSuccess) return 90;
    (void)hipMemset(d, 0, 4);
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&join_ev, hipEventDisableTiming);
    hipError_t e = hipStreamBeginCapture(a, hipStreamCaptureModeGlobal);
    if (e != hipSuccess) return 91;
    touch<<<1, 1, 0, a>>>(d);
    (void)hipEventRecord(fork_ev, a);
    (void)hipStreamWaitEvent(b, fork_ev, 0);  // b joins the capture
    touch<<<1, 1, 0, b>>>(d);
    if (which == 1) {  // joined: the healthy pattern
        (void)hipEventRecord(join_ev, b);
        (void)hipStreamWaitEvent(a, join_ev, 0);
    }
    touch<<<1, 1, 0, a>>>(d);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(a, &g);
    printf("  case %d (%s): hipStreamEndCapture -> %d (%s), graph %s\n", which, which == 1 ? "joined" : "helper stream left forked", (int)e,
           hipGetErrorString(e), g ? "non-null" : "null");
    fflush(stdout);
    if (e == hipSuccess && g) {
        hipGraphExec_t ex = nullptr;
        e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        printf("  case %d: hipGraphInstantiate -> %d (%s)\n", which, (int)e, hipGetErrorString(e));
        if (e == hipSuccess) {
            e = hipGraphLaunch(ex, a);
            hipError_t s = hipStreamSynchronize(a);
            int h = -1;
            (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            printf("  case %d: launch -> %d, sync -> %d, counter %d\n", which, (int)e, (int)s, h);
        }
    }
    // the helper stream after an unjoined end of capture: still usable?
    e = hipStreamSynchronize(b);
    printf("  case %d: helper stream afterwards: synchronize -> %d (%s)\n", which, (int)e, hipGetErrorString(e));
    fflush(stdout);
    return 0;
}

int main() {
    for (int which = 0; which < 2; ++which) {
        fflush(stdout);
        const pid_t pid = fork();  // before any HIP call in this process: the child initialises its own runtime
        if (pid == 0) _exit(run_case(which));
        int status = 0;
        waitpid(pid, &status, 0);
        if (WIFSIGNALED(status))
            printf("case %d: child KILLED by signal %d (%s)\n", which, WTERMSIG(status), WTERMSIG(status) == 11 ? "SIGSEGV" : "other");
        else
            printf("case %d: child exited with %d\n", which, WEXITSTATUS(status));
    }
    return 0;
}
