#!/usr/bin/env python3
"""Builds code-rag_amd/lib/libcoderag_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python code-rag_amd/build.py [--force]

One object per .hip file (compiled in parallel), then one shared library.  No CPU fallback is
built: the library refuses devices that are not gfx950.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libcoderag_hip.so")
SOURCES = ["crag_scan_gemv.hip", "crag_scan_q16.hip", "crag_scan_qsplit.hip", "crag_scan_qlong.hip",
           "crag_select.hip", "crag_capi.hip"]
INCLUDES = {"crag_scan_q16.hip": ["crag_q16_kloop.inc"],
            "crag_scan_qsplit.hip": ["crag_qsplit_kloop.inc"]}
ISA_CHECKED = ("crag_scan_q16.hip", "crag_scan_qsplit.hip", "crag_scan_gemv.hip", "crag_scan_qlong.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


_REG = re.compile(r"\b([av])(?:(\d+)\b|\[(\d+):(\d+)\])")


def check_fixed_register_isa(asm_path, kernel_re, n_kernels, hidden_vgpr_from, live_vgprs=None):
    """crag_scan_q16.hip and crag_scan_qsplit.hip keep their query fragments in fixed registers the compiler
    is not told about (gen_q16_kloop.py / gen_qsplit_kloop.py: all AGPRs, and for q16 at KS = 48 also
    v[128:255]).  The generated statements list them as clobbers, so nothing of the compiler's lives there
    ACROSS a statement; this check proves on the emitted ISA that no compiler-generated instruction touches
    them BETWEEN statements either.  hidden_vgpr_from(match) -> first hidden VGPR of that kernel, or None.
    live_vgprs = (lo, hi): VGPRs that stay live from one statement to the next (crag_scan_q16.hip: the next tile's first
    LDS reads are in flight into them), so compiler code may not name them either."""
    kernel, in_asm, bad, seen, vlim = None, False, [], set(), None
    for n, line in enumerate(open(asm_path), 1):
        m = re.match(kernel_re, line)
        if m:
            kernel, in_asm, vlim = m.group(1), False, hidden_vgpr_from(m)
            seen.add(kernel)
            continue
        if kernel is None:
            continue
        if "s_endpgm" in line:
            kernel = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm:
            code = line.split(";")[0]
            for f, one, lo, hi in _REG.findall(code):
                top = int(one) if one else int(hi)
                low = int(one) if one else int(lo)
                if f == "a" or (vlim is not None and top >= vlim) or \
                        (f == "v" and live_vgprs is not None and top >= live_vgprs[0] and low <= live_vgprs[1]):
                    bad.append(f"{asm_path}:{n}: {line.strip()}")
    if len(seen) != n_kernels:
        raise RuntimeError(f"check_fixed_register_isa: expected {n_kernels} kernels in {asm_path}, found {len(seen)}")
    if bad:
        raise RuntimeError("check_fixed_register_isa: compiler code touches the fixed fragment registers:\n" + "\n".join(bad[:20]))


def check_q16_isa(asm_path):
    check_fixed_register_isa(asm_path, r"^(_Z\S*k_scan_q16ILi(\d+)ELi\d+ELb[01]E\S*):", 12,
                             lambda m: 128 if int(m.group(2)) == 48 else None, live_vgprs=(64, 95))


def check_qsplit_isa(asm_path):
    check_fixed_register_isa(asm_path, r"^(_Z\S*k_scan_qsplitILi\d+ELb[01]E\S*):", 3, lambda m: None)


def check_streaming_isa(asm_path, kernel_re, n_kernels, min_rolling_wait, swap_nop=None):
    """Guards against the two code-generation faults round 2 found in the GEMV, on the emitted ISA
    of every variant of a streaming kernel:
      * no FLAT loads of row data (a pointer that lost its address space: a pending FLAT load turns every later wait
        into vmcnt(0), so the wave drains all its loads instead of keeping them rolling);
      * at least one counted wait `s_waitcnt vmcnt(N)` with N >= min_rolling_wait (the loads do stay in flight);
      * swap_nop: every v_permlane{16,32}_swap is directly preceded by `s_nop <swap_nop>` -- hipcc inserts no hazard
        nops inside an asm statement, and a v_dot2c result needs 3 wait states before the swap reads it (4 before it
        overwrites it); with 2 the last row of a tile got a stale accumulator."""
    kernel, seen, prev, roll, bad = None, set(), "", {}, []
    for n, line in enumerate(open(asm_path), 1):
        m = re.match(kernel_re, line)
        if m:
            kernel = m.group(1)
            seen.add(kernel)
            roll[kernel] = 0
            continue
        if kernel is None:
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        if "s_endpgm" in code:
            kernel = None
            continue
        if code.startswith("flat_load_dwordx"):
            bad.append(f"{asm_path}:{n}: FLAT load in {kernel}: {code}")
        w = re.match(r"s_waitcnt vmcnt\((\d+)\)", code)
        if w:
            roll[kernel] = max(roll[kernel], int(w.group(1)))
        if swap_nop is not None and re.match(r"v_permlane(16|32)_swap", code) and prev != f"s_nop {swap_nop}":
            bad.append(f"{asm_path}:{n}: swap without `s_nop {swap_nop}` in front ({prev!r}) in {kernel}")
        prev = code
    if len(seen) != n_kernels:
        raise RuntimeError(f"check_streaming_isa: expected {n_kernels} kernels in {asm_path}, found {len(seen)}")
    bad += [f"{asm_path}: {k}: deepest counted wait is vmcnt({v}) < {min_rolling_wait}: loads are not kept in flight"
            for k, v in roll.items() if v < min_rolling_wait]
    if bad:
        raise RuntimeError("check_streaming_isa:\n" + "\n".join(bad[:20]))


def check_gemv_isa(asm_path):
    # 2 dtypes x 7 row widths x 4 query widths; variants with few loads per super-tile still wait with vmcnt >= 3
    check_streaming_isa(asm_path, r"^(_Z\S*k_scan_gemvILi\d+ELi\d+ELi\d+E\S*):", 56, 3, swap_nop=4)


def check_qlong_isa(asm_path):
    # 2 dtypes x 4 tile widths x 2 modes x 2 load policies.  Two K steps of loads in flight:
    # 4 + TN / 32 per thread and step, so at least vmcnt(5) behind the prefetch of the next step
    check_streaming_isa(asm_path, r"^(_Z\S*k_scan_qlongILi\d+ELi\d+ELb[01]ELi\d+E\S*):", 32, 5)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(SRC, "crag_common.h"), os.path.join(HERE, "..", "include", "coderag_hip.h")]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(SRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        deps = [src] + headers + [os.path.join(SRC, i) for i in INCLUDES.get(s, [])]
        checked = s in ISA_CHECKED  # device ISA of the same sources and flags, for the fixed-register checks
        asm = os.path.join(OBJ, s.replace(".hip", ".s"))
        if force or not _newer(obj, deps):
            jobs.append(["hipcc"] + FLAGS + ["-c", src, "-o", obj])
            if checked:
                jobs.append(["hipcc"] + FLAGS + ["--cuda-device-only", "-S", src, "-o", asm])
        elif checked and not _newer(asm, deps):
            # the object is current but the listing that vouches for it is missing or older than a source: a check
            # of a stale listing proves nothing about the object that ships
            jobs.append(["hipcc"] + FLAGS + ["--cuda-device-only", "-S", src, "-o", asm])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    for s in ISA_CHECKED:  # fail loudly rather than check a listing that does not belong to the object
        src, asm, obj = os.path.join(SRC, s), os.path.join(OBJ, s.replace(".hip", ".s")), os.path.join(OBJ, s.replace(".hip", ".o"))
        deps = [src] + headers + [os.path.join(SRC, i) for i in INCLUDES.get(s, [])]
        if not os.path.exists(asm) or any(os.path.getmtime(d) > os.path.getmtime(asm) for d in deps):
            raise RuntimeError(f"{asm} is missing or older than its sources: the fixed-register check would vouch for other code than {obj}")
    check_q16_isa(os.path.join(OBJ, "crag_scan_q16.s"))
    check_qsplit_isa(os.path.join(OBJ, "crag_scan_qsplit.s"))
    check_gemv_isa(os.path.join(OBJ, "crag_scan_gemv.s"))
    check_qlong_isa(os.path.join(OBJ, "crag_scan_qlong.s"))
    if jobs or not os.path.exists(LIB):
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
