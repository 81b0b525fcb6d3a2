#!/usr/bin/env python3
"""Static instruction census of a gemm_kc_kernel instantiation's assembly: the k-loop (between s_setprio 2 / 0) and the rest of the
tile (epilogue + tile prologue), by class -- the yardstick for epilogue work per output element (a wave-tile is 128 x 64 = 8192
elements on 64 lanes = 128 elements per lane, so VALU per element = epilogue VALU / 128).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -fno-slp-vectorize -S --cuda-device-only bsp_kc.hip -o /tmp/bsp_kc.s
    python tools/asm_epilogue_stats.py /tmp/bsp_kc.s [substring of the mangled name ...]
"""
import collections
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(line)


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op in ("v_sin_f32", "v_sqrt_f32", "v_rcp_f32", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_cos_f32"):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("s_waitcnt") or op == "s_nop" or op == "s_barrier":
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def census(body):
    region, counts, ops = "pre", {"pre": collections.Counter(), "loop": collections.Counter(), "post": collections.Counter()}, collections.Counter()
    for line in body:
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_setprio":
            region = "loop" if "2" in t.split()[1] else "post"
            continue
        counts[region][klass(op)] += 1
        if region == "post" and klass(op) in ("valu", "valu_trans"):
            ops[op] += 1
    return counts, ops


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if "gemm_kc_kernel" not in name or (pats and not all(p in name for p in pats)):
            continue
        m = re.search(r"gemm_kc_kernelILi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)ELi(\d)E", name)
        tag = "PL=%s ACT=%s AUX=%s COLSUM=%s SIGNS=%s SINM=%s DIAG=%s NDOT=%s" % m.groups() if m else name
        c, ops = census(body)
        ep = c["post"]
        print(f"{tag}")
        for r in ("pre", "loop", "post"):
            print(f"  {r:5s}", dict(c[r]))
        print(f"  epilogue VALU per element: {(ep['valu'] + ep['valu_trans']) / 128:.2f}  (transcendental {ep['valu_trans'] / 128:.2f})")
        print("  top epilogue VALU ops:", ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))


if __name__ == "__main__":
    main()
