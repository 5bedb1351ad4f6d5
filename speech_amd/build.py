"""Builds speech_amd/libctn_hip.so (gfx950) from csrc/*.hip with hipcc.

    python -m speech_amd.build [--force] [--verbose]

The objects are compiled one source at a time (parallel), then linked into a
single shared library that exports the C ABI declared in include/ctn.h.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libctn_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-fno-gpu-rdc"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ctn.h"))
    srcs = sources()
    objs = [os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not force and not _stale(obj, [src] + headers):
            return None
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on {}:\n{}\n{}".format(src, r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        logs = list(ex.map(compile_one, zip(srcs, objs)))
    if verbose:
        for log in logs:
            if log:
                print(log)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n{}\n{}".format(r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
