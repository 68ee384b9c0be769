"""Build the HIP/C++ shared library in-tree: ferreus_rbf_rs_amd/libferreus_bbfmm_hip.so.

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box
with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libferreus_bbfmm_hip.so")

HOST_SOURCES = ["tree.cpp", "operators.cpp", "fmm_tree.cpp", "capi.cpp", "solver.cpp", "ddm.cpp", "ddm_solver.cpp", "schwarz.cpp"]
HIP_SOURCES = ["device.hip", "ddm_kernels.hip", "targets.hip"]
HEADERS = ["morton.hpp", "tree.hpp", "parallel.hpp", "kernels.hpp", "operators.hpp", "device.hpp",
           "fmm_tree.hpp", "targets.hpp", "ddm.hpp", "ddm_solver.hpp", os.path.join(ROOT, "include", "ferreus_bbfmm_hip.h")]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    common = ["-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
              "-Wall", "-Wno-unused-result"]
    objs = []
    for src in HOST_SOURCES + HIP_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        objs.append(obj)
        if force or _newer(obj, [path] + hdrs):
            cmd = [hipcc] + common + ["--offload-arch=gfx950", "-c", path, "-o", obj]
            if src.endswith(".cpp"):
                # host translation units still see the HIP runtime API (hip_runtime.h)
                cmd += ["-x", "hip"] if False else []
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(obj):
                os.remove(obj)  # never link a stale object after a failed compile
            subprocess.check_call(cmd)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
