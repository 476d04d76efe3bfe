#!/usr/bin/env python3
"""Tuning build with a few coarse cycle stamps in diag_kernel<false> -> tools/ab/stamps2.so (never the product: the sources are
copied to /tmp and the stamps inserted by text anchors).   python tools/ab/make_stamps2.py && (GPU box) python tools/ab/stamps2.py"""
import os, shutil, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "bark_amd", "csrc"), "/tmp/stamps2_src/bark_amd/csrc"
shutil.rmtree("/tmp/stamps2_src", ignore_errors=True)
os.makedirs(dst)
for f in os.listdir(src):
    if f.endswith((".hip", ".cpp", ".h")):
        shutil.copy(os.path.join(src, f), dst)
os.makedirs("/tmp/stamps2_src/include")
shutil.copy(os.path.join(ROOT, "include", "bark_hip.h"), "/tmp/stamps2_src/include")
p = os.path.join(dst, "chol.hip")
s = open(p).read()

def once(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old)
    s = s.replace(old, new)

once("namespace {\n\ntypedef double f64x4", "__device__ unsigned long long g_stamps[64];\n#define STAMP(i) do { if (!ONE && blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_readcyclecounter(); } while (0)\nnamespace {\n\ntypedef double f64x4")
once("    const int info_in = (tid == 0 && !one) ? p.info[b] : 0;\n", "    const int info_in = (tid == 0 && !one) ? p.info[b] : 0;\n    STAMP(0);\n")
once("    __syncthreads();\n\n    // --- blocked Cholesky D = U'U and X = U^-1", "    __syncthreads();\n    STAMP(1);\n\n    // --- blocked Cholesky D = U'U and X = U^-1")
once("        if (wave_u == 0) factor16(S + blk_off(0, 0), lane, 0, pacc, bad);\n        __syncthreads();\n    }\n", "        if (wave_u == 0) factor16(S + blk_off(0, 0), lane, 0, pacc, bad);\n        __syncthreads();\n    }\n    STAMP(2);\n")
once("            double *dblk = S + blk_off(kb, kb);  // W_kk\n            phase_b(kb, dblk);\n            __syncthreads();\n            if (kb + 1 < nsb) {\n", "            double *dblk = S + blk_off(kb, kb);  // W_kk\n            STAMP(8 + kb);\n            phase_b(kb, dblk);\n            __syncthreads();\n            STAMP(16 + kb);\n            if (kb + 1 < nsb) {\n")
once("                factor16(dst, lane, (kb + 1) * SB, pacc, bad);\n", "                factor16(dst, lane, (kb + 1) * SB, pacc, bad);\n                STAMP(24 + kb);\n")
once("    {  // last column of X", "    STAMP(3);\n    {  // last column of X")
once("    // --- W_j out, sub-block by sub-block", "    STAMP(4);\n    // --- W_j out, sub-block by sub-block")
once("    if (want_g)  // workgroup-uniform", "    STAMP(5);\n    if (want_g)  // workgroup-uniform")
once("    // --- z_j = W_j' y_j ; quad += |z_j|^2", "    STAMP(6);\n    // --- z_j = W_j' y_j ; quad += |z_j|^2")
once("    if (tid == 0 && one) {  // finish_mll_kernel's arithmetic", "    STAMP(7);\n    if (tid == 0 && one) {  // finish_mll_kernel's arithmetic")
s += '\nextern "C" int bark_debug_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64); }\n'
open(p, "w").write(s)
objs, procs = [], []
for f in sorted(os.listdir(dst)):
    if f.endswith((".hip", ".cpp")):
        o = f"/tmp/stamps2_{f}.o"
        objs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                       "-Wno-unused-function", "-x", "hip", "-c", os.path.join(dst, f), "-o", o]))
for pr in procs:
    assert pr.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-pthread", "-o", os.path.join(ROOT, "tools/ab/stamps2.so")] + objs + ["-ldl"])
print("built tools/ab/stamps2.so")
