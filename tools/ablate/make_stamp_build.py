#!/usr/bin/env python3
"""Diagnostic build of the persistent K-contiguous kernel with s_memtime stamps: a COPY of csrc/bsp_kc.hip gets stamp
statements inserted by text substitution (the product source carries none), is compiled and linked with the product
objects into tools/ablate/libsnerf_hip_stamp.so.  Per tile (SIN forward launches, debug buffer = the otherwise unused
colsum pointer): tile start | k-loop | drain + barrier | next-tile requests | epilogue, time in the in-loop waits, time in
the strip read-back + stores; per workgroup: cycles, 100 MHz ticks (-> clock), tiles done.
Run:  python tools/ablate/make_stamp_build.py && gpurun -- 'SNERF_LIB_PATH=$PWD/tools/ablate/libsnerf_hip_stamp.so STAMP_OUT=gpurun_out/x.npy
      python tools/bsp_kernel_bench.py 3 stamp && python tools/ablate/stamp_kc.py gpurun_out/x.npy'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "semantic-nerf-for-satellite-data_amd", "csrc")
t = open(os.path.join(SRC, "bsp_kc.hip")).read()

def sub(old, new):
    global t
    assert old in t, old[:60]
    t = t.replace(old, new)

sub("  bool first = true;\n  for (int it = 0;; ++it) {",
    "  bool first = true;\n  unsigned long long st_w2 = 0, st_wall = 0;\n  const unsigned long long R0 = __builtin_amdgcn_s_memrealtime(), C0 = __builtin_amdgcn_s_memtime();\n"
    "  for (int it = 0;; ++it) {\n    const unsigned long long T0 = __builtin_amdgcn_s_memtime();\n    st_w2 = 0; st_wall = 0;")
sub("      if (s >= 2) wait_b(bc);",
    "      { const unsigned long long c0 = __builtin_amdgcn_s_memtime(); if (s >= 2) wait_b(bc); st_wall += __builtin_amdgcn_s_memtime() - c0; }")
sub("    __builtin_amdgcn_s_setprio(2);\n    for (int s = 0;", "    const unsigned long long T1 = __builtin_amdgcn_s_memtime();\n    __builtin_amdgcn_s_setprio(2);\n    for (int s = 0;")
sub("    wait_vm<0>();        // rejected requests behind the last stage write zeros into the ring: drain before re-using it\n",
    "    const unsigned long long T2a = __builtin_amdgcn_s_memtime();\n    wait_vm<0>();\n")
sub("    // ---- this tile's coordinates for the epilogue; then the next tile's operands are requested",
    "    const unsigned long long T2 = __builtin_amdgcn_s_memtime();\n    // ---- this tile's coordinates for the epilogue; then the next tile's operands are requested")
sub("    // ---- epilogue.  Lane l: point pt", "    const unsigned long long T3 = __builtin_amdgcn_s_memtime();\n    st_w2 = 0;\n    // ---- epilogue.  Lane l: point pt")
sub("            strip_put(gg, phi[gg], plo[gg]);\n          }\n          strip_flush(mi, nj);\n          keep_planes(phi, plo);\n        }\n        if (SIGNS",
    "            strip_put(gg, phi[gg], plo[gg]);\n          }\n          { __builtin_amdgcn_sched_barrier(0); const unsigned long long f0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);\n"
    "          strip_flush(mi, nj);\n          __builtin_amdgcn_sched_barrier(0); st_w2 += __builtin_amdgcn_s_memtime() - f0; __builtin_amdgcn_sched_barrier(0); }\n          keep_planes(phi, plo);\n        }\n        if (SIGNS")
sub("    if (!more) break;\n    vb = vbn;",
    "    { const unsigned long long T4 = __builtin_amdgcn_s_memtime();\n      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(kargs()->colsum);\n"
    "      if (ONEPASS && dbg != nullptr && t == 0) { dbg += 8 * (size_t)vb; dbg[0] = T0; dbg[1] = T1 - T0; dbg[2] = T2a - T1; dbg[3] = T2 - T2a; dbg[4] = T3 - T2; dbg[5] = T4 - T3; dbg[6] = st_wall; dbg[7] = st_w2;\n"
    "        if (!more) { unsigned long long* w = reinterpret_cast<unsigned long long*>(kargs()->colsum) + 8 * 4096 + 4 * (size_t)blockIdx.x; w[0] = C0; w[1] = T4 - C0; w[2] = __builtin_amdgcn_s_memrealtime() - R0; w[3] = it + 1; } } }\n"
    "    if (!more) break;\n    vb = vbn;")
tmp = os.path.join(SRC, "bsp_kc_stamp_tmp.hip")
open(tmp, "w").write(t)
try:
    subprocess.check_call(["make", "-C", SRC, "-j6"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + SRC, "-Wno-unused-function", "-Wno-pass-failed",
                           "-fno-slp-vectorize", "-c", tmp, "-o", "/tmp/bsp_kc_stamp.o"])
finally:
    os.remove(tmp)
objs = ["legacy/gemm.o", "legacy/gemm_x6.o", "profile.o", "/tmp/bsp_kc_stamp.o", "bsp_gemm.o", "bsp_aux.o", "bsp_pass.o", "aux_kernels.o", "composite.o", "loss.o", "optim.o", "api.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "tools", "ablate", "libsnerf_hip_stamp.so")] + objs, cwd=SRC)
print("built tools/ablate/libsnerf_hip_stamp.so")
