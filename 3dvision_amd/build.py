"""Build lib3dvision_hip.so (hand-written HIP for gfx950 + the extern "C" dispatch layer).

hipcc cross-compiles without a GPU.  The library is built IN-TREE (3dvision_amd/lib3dvision_hip.so)
so that it travels to the GPU box with the repo snapshot.

Flags that matter for parity:
  -ffp-contract=off       no FMA contraction: distances, transforms and the small solvers are
                          evaluated with the same rounding steps as the reference's x86-64 build
  -fno-slp-vectorize      hipcc otherwise packs the distance math into v_pk_*_f32, which on
                          gfx950 runs at the scalar-f32 rate and costs extra SGPR moves (measured)
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE sqrt/div (default on, stated explicitly)
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(SRC, "_obj")
LIB = os.path.join(HERE, "lib3dvision_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", SRC] + os.environ.get("TDV_HIPCC_FLAGS", "").split()   # tuning builds (-DNAME=value)


def _sources():
    return sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "tdv_hip.h")]
    if not _stale(obj, [os.path.join(SRC, src)] + headers + [os.path.abspath(__file__)]):
        return obj, ""
    r = subprocess.run([HIPCC] + FLAGS + ["-c", os.path.join(SRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
    return obj, r.stderr


def build(verbose=False, jobs=6):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    if verbose:
        for _, err in results:
            if err.strip():
                sys.stderr.write(err)
    # drop objects of sources that no longer exist
    for f in os.listdir(OBJ):
        if f.endswith(".o") and os.path.join(OBJ, f) not in objs:
            os.remove(os.path.join(OBJ, f))
    if _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
