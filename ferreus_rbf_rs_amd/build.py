"""Build the HIP/C++ shared library in-tree: ferreus_rbf_rs_amd/libferreus_bbfmm_hip.so.

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box
with the repository snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libferreus_bbfmm_hip.so")

HOST_SOURCES = ["tree.cpp", "operators.cpp", "fmm_tree.cpp", "fmm_m2l_tables.cpp", "fmm_plans.cpp", "device_group.cpp", "capi.cpp", "solver.cpp", "ddm.cpp", "ddm_solver.cpp", "schwarz.cpp"]
HIP_SOURCES = ["device.hip", "ddm_kernels.hip", "targets.hip", "schwarz_kernels.hip", "tree_device.hip", "tree_lists_device.hip"]
HEADERS = ["morton.hpp", "tree.hpp", "parallel.hpp", "kernels.hpp", "operators.hpp", "device.hpp",
           "fmm_tree.hpp", "fmm_tree_impl.hpp", "device_group.hpp", "targets.hpp", "ddm.hpp", "ddm_solver.hpp", "ddm_monomials.hpp", "schwarz_kernels.hpp", "tree_device.hpp", os.path.join(ROOT, "include", "ferreus_bbfmm_hip.h")]


def _digest(paths: list[str], extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def _file_digest(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def _stale(target: str, stamp: str, digest: str) -> bool:
    """Content-based (not mtime-based): a snapshot copied to another machine keeps its objects.  The stamp holds
    the digest of the inputs AND of the built file, so a stamp that no longer belongs to the file next to it
    (restored by a checkout, copied alone) never passes for "up to date"."""
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        want = f.read().split()
    return len(want) != 2 or want[0] != digest or want[1] != _file_digest(target)


def _write_stamp(target: str, stamp: str, digest: str) -> None:
    with open(stamp, "w") as f:
        f.write(digest + " " + _file_digest(target))


_COMPILER_ID = None


def _compiler_id(hipcc: str) -> str:
    """`hipcc --version` (a toolchain upgrade invalidates the objects; the path of the checkout does not)."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            _COMPILER_ID = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                          timeout=120).stdout.decode(errors="replace")
        except (OSError, subprocess.SubprocessError):
            _COMPILER_ID = "unknown"
    return _COMPILER_ID


# Host-side sanitizer builds (CPU only -- never on the GPU box, whose pool refuses GPU sanitizers): the HOST code of
# every translation unit instrumented, device code compiled as always.  A second library beside the product's, loaded
# instead of it when FERREUS_BBFMM_HIP_LIB names it (scripts/sanitize_host.sh preloads the matching runtime).
SANITIZERS = {
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
    "tsan": ["-fsanitize=thread"],
}


def sanitized_lib(kind: str) -> str:
    return os.path.join(HERE, f"libferreus_bbfmm_hip_{kind}.so")


def sanitizer_runtime(kind: str, hipcc: str = None) -> str:
    """The shared runtime to LD_PRELOAD into the Python process (clang's, matching the compiler that built the library)."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    name = {"asan": "libclang_rt.asan-x86_64.so", "tsan": "libclang_rt.tsan-x86_64.so"}[kind]
    out = subprocess.run([hipcc, f"-print-file-name={name}"], stdout=subprocess.PIPE, timeout=120).stdout.decode().strip()
    if not os.path.isabs(out) or not os.path.exists(out):
        raise FileNotFoundError(f"{name}: not found next to {hipcc}")
    return out


def build(force: bool = False, verbose: bool = False, sanitize: str = None) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    global OBJ, LIB
    if sanitize:
        if sanitize not in SANITIZERS:
            raise ValueError(f"unknown sanitizer {sanitize!r}: one of {sorted(SANITIZERS)}")
        obj_dir, lib_path = os.path.join(HERE, f"_obj_{sanitize}"), sanitized_lib(sanitize)
    else:
        obj_dir, lib_path = OBJ, LIB
    return _build(hipcc, obj_dir, lib_path, force, verbose, sanitize)


def _build(hipcc: str, OBJ: str, LIB: str, force: bool, verbose: bool, sanitize: str) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    flags = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "--offload-arch=gfx950"]  # path-independent
    if sanitize:   # -fno-gpu-sanitize: host code only; -shared-libsan: the runtime comes from the preloaded shared object
        flags = ["-O1", "-g", "-fno-omit-frame-pointer"] + SANITIZERS[sanitize] + ["-fno-gpu-sanitize", "-shared-libsan"] + flags[1:]
    common = flags[:-1] + ["-I", CSRC, "-I", os.path.join(ROOT, "include")]
    flag_id = " ".join(flags) + "\n" + _compiler_id(hipcc)
    objs, digests, todo = [], [], []
    for src in HOST_SOURCES + HIP_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        stamp = obj + ".sha256"
        cmd = [hipcc] + common + ["--offload-arch=gfx950", "-c", path, "-o", obj]
        digest = _digest([path] + hdrs, flag_id)  # file names + contents, flags, compiler: nothing about where the tree lies
        objs.append(obj)
        digests.append(digest)
        if force or _stale(obj, stamp, digest):
            todo.append((cmd, obj, stamp, digest))

    def compile_one(job):
        cmd, obj, stamp, digest = job
        if verbose:
            print(" ".join(cmd), flush=True)
        for f in (obj, stamp):
            if os.path.exists(f):
                os.remove(f)  # never link a stale object after a failed compile
        subprocess.check_call(cmd)
        _write_stamp(obj, stamp, digest)

    if todo:  # independent translation units: compile side by side (device.hip alone takes over a minute)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(compile_one, todo))
    lib_digest = hashlib.sha256("".join(digests).encode()).hexdigest()
    lib_stamp = os.path.join(OBJ, "lib.sha256")
    if force or _stale(LIB, lib_stamp, lib_digest):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread", "-ldl"]
        if sanitize:
            cmd[1:1] = SANITIZERS[sanitize] + ["-fno-gpu-sanitize", "-shared-libsan"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        _write_stamp(LIB, lib_stamp, lib_digest)
    return LIB


if __name__ == "__main__":
    kind = None
    for a in sys.argv[1:]:
        if a == "--sanitize":
            kind = "asan"
        elif a.startswith("--sanitize="):
            kind = a.split("=", 1)[1]
    print(build(force="--force" in sys.argv, verbose=True, sanitize=kind))
