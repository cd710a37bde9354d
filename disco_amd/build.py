"""Builds libdisco_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

The library is built IN-TREE (``disco_amd/csrc/libdisco_hip.so``) so that the binary travels with
the source tree to the GPU box; nothing is JIT-compiled at import time.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
SOURCES = ["disco_spmm.hip", "disco_stream.hip", "disco_schedule.hip", "disco_norm.hip", "disco_scorer.hip", "disco_build.hip", "disco_gather.hip", "disco_locality.hip"]
HEADERS = ["disco_internal.h", "disco_spmm_common.h", os.path.join(ROOT, "include", "disco_hip.h")]
ARCH = "gfx950"


def lib_path(name: str = "libdisco_hip.so") -> str:
    return os.path.join(CSRC, name)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest_source_mtime() -> float:
    paths = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    paths += [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    paths.append(os.path.abspath(__file__))
    return max(os.path.getmtime(p) for p in paths if os.path.exists(p))


def build_lib(force: bool = False, extra_flags=(), name: str = "libdisco_hip.so", verbose: bool = False) -> str:
    out = lib_path(name)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest_source_mtime():
        return out
    hipcc = _hipcc()
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-ffp-contract=off", "-Wall", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
    flags += list(extra_flags)
    objdir = os.path.join(CSRC, "build", name.replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
