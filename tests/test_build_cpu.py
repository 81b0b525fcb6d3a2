"""Build-time checks on the generated gfx950 code (CPU only: hipcc cross-compiles; skipped where hipcc is absent): register hazards
the compiler does not model (test_no_vgpr_hazards_in_any_kernel) and the asynchronous atomic below.

bsp_kc.hip draws a workgroup's next tile with an ASYNCHRONOUS returning atomic written as an asm statement: the result register is an
ordinary "=v" output, so the compiler believes it is defined the moment the statement ends and would be free to copy or spill it
before the data has arrived.  What makes the code valid is that nothing touches the register until the
s_waitcnt that retires the atomic -- this test holds every instantiation of the kernel to that, in the assembly the product
build produces: the register may not appear in any instruction between the atomic and the vmcnt(0) drain behind the k-loop."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semantic-nerf-for-satellite-data_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _touches(line: str, reg: int) -> bool:
    for m in re.finditer(r"\bv(\d+)\b", line):
        if int(m.group(1)) == reg:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        if int(m.group(1)) <= reg <= int(m.group(2)):
            return True
    return False


SRCS = ("profile", "bsp_kc", "bsp_trunk", "bsp_gemm", "bsp_aux", "bsp_pass", "aux_kernels", "composite", "loss", "optim", "api")
NO_SLP = ("bsp_kc", "bsp_trunk", "bsp_gemm")     # csrc/Makefile: CXXFLAGS += -fno-slp-vectorize for these


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """every source of the library compiled to gfx950 assembly with the product build's flags (csrc/Makefile), four at a time"""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("asm")
    def cmd(name):
        return [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed",
                "-Wno-unused-command-line-argument", "-I" + CSRC] + (["-fno-slp-vectorize"] if name in NO_SLP else []) + \
               ["--cuda-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", str(out / (name + ".s"))]
    pending, running = list(SRCS), []
    while pending or running:
        while pending and len(running) < 4:
            n = pending.pop(0)
            running.append((n, subprocess.Popen(cmd(n))))
        n, pr = running.pop(0)
        assert pr.wait(timeout=900) == 0, f"hipcc -S {n}.hip failed"
    return {n: (out / (n + ".s")).read_text().splitlines() for n in SRCS}


def test_no_vgpr_hazards_in_any_kernel(device_asm):
    """tools/check_vgpr_hazards.py over the whole library (scan 3, round 5: no VALU instruction reads a transcendental's result in
    the next issue slot -- an asm v_bfi behind a v_sqrt did, and half of the derivative epilogue's rows were wrong): no VALU write into the data registers of a wide store within two wait
    states (the round-4 corruption: csrc/bsp_dev.h store_data_guard), no ds_read returning into the data registers of an unretired
    wide ds_write (the round-2 one: csrc/bsp_kc.hip keep_planes).  Neither is interlocked by the hardware or known to the compiler;
    both came and went with register allocation, i.e. with unrelated edits -- hence a check on the generated code."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_vgpr_hazards", os.path.join(ROOT, "tools", "check_vgpr_hazards.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    n_store = 0
    for name, lines in device_asm.items():
        n_store += sum(1 for ln in lines if re.search(r"\b(buffer|global)_store_dwordx4\b", ln))
        hits = chk.scan_store(lines)
        assert not hits, f"{name}.hip: store data overwritten too early: {hits[:3]}"
        hits = chk.scan_lds(lines)
        assert not hits, f"{name}.hip: ds_read into unretired ds_write data: {hits[:3]}"
        hits = chk.scan_trans(lines)
        assert not hits, f"{name}.hip: a VALU instruction reads a transcendental's result without a wait state: {hits[:3]}"
    assert n_store > 100       # the scan saw the plane stores at all
    # the scanner does flag the pattern it was written for
    bad = ["_Zk:", "buffer_store_dwordx4 v[90:93], v122, s[36:39], s20 offen nt", "v_or_b32_e32 v90, 32, v182", "s_endpgm"]
    assert len(chk.scan_store(bad)) == 1
    ok = ["_Zk:", "buffer_store_dwordx4 v[90:93], v122, s[36:39], s20 offen nt", "s_nop 1", "v_or_b32_e32 v90, 32, v182", "s_endpgm"]
    assert not chk.scan_store(ok)
    bad = ["_Zk:", "ds_write_b128 v1, v[4:7]", "ds_read_b128 v[6:9], v2", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert len(chk.scan_lds(bad)) == 1
    bad = ["_Zk:", "v_sqrt_f32_e64 v7, |v3|", "v_bfi_b32 v9, s4, v7, v8", "s_endpgm"]       # the round-5 corruption
    assert len(chk.scan_trans(bad)) == 1
    ok = ["_Zk:", "v_sqrt_f32_e64 v7, |v3|", "s_nop 0", "v_bfi_b32 v9, s4, v7, v8", "v_sin_f32_e32 v1, v2", "v_sqrt_f32_e32 v3, v1", "s_endpgm"]
    assert not chk.scan_trans(ok)


def test_tile_counter_atomic_result_is_untouched_until_its_wait(device_asm):
    lines = device_asm["bsp_kc"]
    atomics = [i for i, ln in enumerate(lines) if "global_atomic_add" in ln and " sc0" in ln]
    assert len(atomics) >= 18, len(atomics)          # one per instantiation (2 plane counts x 9 epilogue variants)
    for i in atomics:
        m = re.search(r"global_atomic_add\s+v(\d+),", lines[i])
        assert m, lines[i]
        dest = int(m.group(1))
        waited = False
        for j in range(i + 1, min(i + 6000, len(lines))):
            ln = lines[j].split(";")[0]
            if "s_waitcnt" in ln and re.search(r"vmcnt\(0\)", ln):     # the drain behind the k-loop (the counted waits inside the loop do
                waited = True                                          #  retire the atomic earlier, but which of them is reached first depends on
                break                                                  #  the trip count: the register must stay untouched through the whole loop)
            if ln.strip().startswith("s_endpgm"):
                break
            assert not _touches(ln, dest), f"v{dest} (tile-counter atomic of line {i + 1}) is touched at line {j + 1} before the k-loop's drain: {ln.strip()}"
        assert waited, f"no vmcnt(0) drain behind the atomic of line {i + 1}"


def test_trunk_tile_counter_atomic_is_untouched_through_its_k_loop(device_asm):
    """bsp_trunk.hip draws the next tile the same way, in front of layer 1's k-loop, and reads the result right behind the barrier that
    ends the loop: in the generated code nothing may name the register between the atomic and that barrier, and the loop's counted
    waits (vmcnt(4): everything but the youngest sub-step's weight requests has landed) must lie in between."""
    lines = device_asm["bsp_trunk"]
    atomics = [i for i, ln in enumerate(lines) if "global_atomic_add" in ln and " sc0" in ln]
    assert len(atomics) == 3, len(atomics)           # training, inference + feats, inference
    for i in atomics:
        dest = int(re.search(r"global_atomic_add\s+v(\d+),", lines[i]).group(1))
        counted = 0
        for j in range(i + 1, len(lines)):
            ln = lines[j].split(";")[0]
            if "s_barrier" in ln:
                break
            assert not ln.strip().startswith("s_endpgm")
            counted += bool(re.search(r"s_waitcnt\s+vmcnt\(4\)", ln))
            assert not _touches(ln, dest), f"v{dest} (tile-counter atomic of line {i + 1}) is touched at line {j + 1} inside the k-loop: {ln.strip()}"
        assert counted >= 4, counted


def test_hand_counted_kernels_do_not_spill(device_asm):
    """gemm_kc / gemm_dw / trunk kernels count their vector-memory requests by hand (s_waitcnt vmcnt(N) between asm loads): a spilled
    register is a scratch load or store the count does not know -- and whether the compiler spills came and went with unrelated edits
    (bsp_trunk.hip: a compile-time layer count instead of a run-time one spilled 445 registers).  Every instantiation: no spills, no scratch."""
    n = 0
    for name in ("bsp_kc", "bsp_trunk", "bsp_gemm"):
        text = "\n".join(device_asm[name])
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
            kname, scratch, spills = m.group(1), int(m.group(2)), int(m.group(3))
            if "gemm_kc_kernel" in kname or "trunk_kernel" in kname or "gemm_dw_kernel" in kname or "gemm_kcn_kernel" in kname:
                assert scratch == 0 and spills == 0, (kname, scratch, spills)
                n += 1
    assert n >= 30, n
