#!/usr/bin/env python3
"""Builds code-rag_amd/lib/libcoderag_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python code-rag_amd/build.py [--force]

One object per .hip file (compiled in parallel), then one shared library.  No CPU fallback is
built: the library refuses devices that are not gfx950.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libcoderag_hip.so")
SOURCES = ["crag_scan_gemv.hip", "crag_scan_mfma.hip", "crag_scan_qreg.hip", "crag_select.hip", "crag_capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


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
        if force or not _newer(obj, [src] + headers):
            jobs.append(["hipcc"] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
