#!/usr/bin/env python3
"""HISTORIC: the text anchors below match diag_kernel as it was up to the middle of round 3 (git show 6b1c09b:bark_amd/csrc/chol.hip);
against the current sources use tools/ab/make_stamps2.py / stamps2.py (coarse stamps, which do not perturb the kernel).
Tuning build with s_memtime stamps in diag_kernel -> tools/ab/stamps.so (never the product; the product sources carry
no stamp code: this script copies bark_amd/csrc to /tmp, inserts the stamps by text anchors, and builds).
   python tools/ab/make_stamps.py && (on the GPU box) python tools/ab/stamps.py"""
import os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "bark_amd", "csrc"), "/tmp/stamps_src/bark_amd/csrc"
shutil.rmtree("/tmp/stamps_src", ignore_errors=True)
os.makedirs(dst)
for f in os.listdir(src):
    if f.endswith((".hip", ".cpp", ".h")):
        shutil.copy(os.path.join(src, f), dst)
os.makedirs("/tmp/stamps_src/include")
shutil.copy(os.path.join(ROOT, "include", "bark_hip.h"), "/tmp/stamps_src/include")
p = os.path.join(dst, "chol.hip")
s = open(p).read()

def once(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old)
    s = s.replace(old, new)

once("namespace {\n\ntypedef double f64x4", "__device__ unsigned long long g_stamps[64];\n#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_readcyclecounter(); } while (0)\n#define STAMPW0(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && threadIdx.x < 64) g_stamps[i] = __builtin_readcyclecounter(); } while (0)\nnamespace {\n\ntypedef double f64x4")
once("    const int info_in = tid == 0 ? p.info[b] : 0;\n", "    const int info_in = tid == 0 ? p.info[b] : 0;\n    STAMP(0);\n")
once("    __syncthreads();\n\n    // --- blocked Cholesky D = U'U and X = U^-1", "    __syncthreads();\n    STAMP(2);\n\n    // --- blocked Cholesky D = U'U and X = U^-1")
once("        run_factor16(E, 0);\n    }\n    __syncthreads();\n", "        run_factor16(E, 0);\n    }\n    __syncthreads();\n    STAMP(7);\n")
once("        double *dblk = S + blk_off(kb, kb);  // W_kk\n", "        double *dblk = S + blk_off(kb, kb);  // W_kk\n        if (kb == 3) STAMP(10);\n")
once("        __syncthreads();\n        if (wave_u == 0) {\n            if (kb + 1 < NSB) {\n                // D[kb+1,kb+1]", "        __syncthreads();\n        if (kb == 3) STAMP(11);\n        if (wave_u == 0) {\n            if (kb + 1 < NSB) {\n                // D[kb+1,kb+1]")
once("                run_factor16(dnext - uu, kb + 1);\n", "                if (kb == 3) STAMP(16);\n                run_factor16(dnext - uu, kb + 1);\n                if (kb == 3) STAMP(12);\n")
once("        }\n        __syncthreads();\n    }\n    if (wave_u == 0) {  // log|D_jj| / 2", "        }\n        __syncthreads();\n        if (kb == 3) STAMP(13);\n    }\n    STAMP(3);\n    if (wave_u == 0) {  // log|D_jj| / 2")
once("    // --- W_j out, sub-block by sub-block", "    STAMP(4);\n    // --- W_j out, sub-block by sub-block")
once("    // --- z_j = W_j' y_j ; quad += |z_j|^2", "    STAMP(5);\n    // --- z_j = W_j' y_j ; quad += |z_j|^2")
once("    if (tid == 0) {  // wave 0 ran factor16: its logsum / bad are the matrix's\n", "    STAMP(6);\n    if (tid == 0) {  // wave 0 ran factor16: its logsum / bad are the matrix's\n")
# inside block4<0> of the kb == 3 call: stamps 20.. (only the first block of each call; the last call before reading wins)
once("    const double rd0 = pivot(P00, 0);\n", "    if (Q == 0) STAMPW0(20);\n    const double rd0 = pivot(P00, 0);\n")
once("    piv[Q][0] = P00;  // the frozen pivots", "    if (Q == 0) STAMPW0(21);\n    piv[Q][0] = P00;  // the frozen pivots")
once("    f[Q] = g == 0 ? S0 : g == 1 ? S1 : g == 2 ? S2 : S3;\n", "    f[Q] = g == 0 ? S0 : g == 1 ? S1 : g == 2 ? S2 : S3;\n    if (Q == 0) STAMPW0(22);\n    if (Q == 1) STAMPW0(23);\n    if (Q == 2) STAMPW0(24);\n    if (Q == 3) STAMPW0(25);\n")
once("    block4<0>(e, f, piv, c, g, base_index, bad);\n", "    STAMPW0(19);\n    block4<0>(e, f, piv, c, g, base_index, bad);\n    STAMPW0(26);\n")
once("        blk[c * SB + (g + 4 * v)] = f[v] * rs;\n    }\n}", "        blk[c * SB + (g + 4 * v)] = f[v] * rs;\n    }\n    STAMPW0(27);\n}")
s += '\nextern "C" int bark_debug_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64); }\n'
open(p, "w").write(s)
objs = []
procs = []
for f in sorted(os.listdir(dst)):
    if f.endswith((".hip", ".cpp")):
        o = f"/tmp/stamps_{f}.o"
        objs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                       "-Wno-unused-function", "-x", "hip", "-c", os.path.join(dst, f), "-o", o]))
for pr in procs:
    assert pr.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-pthread", "-o", os.path.join(ROOT, "tools/ab/stamps.so")] + objs)
print("built tools/ab/stamps.so")
