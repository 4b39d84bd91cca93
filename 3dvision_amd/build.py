"""Build lib3dvision_hip.so (hand-written HIP for gfx950 + the extern "C" dispatch layer) and, beside it, lib3dvision_hip_study.so:
the same sources with -DTDV_STUDY, which keeps the A/B variants that lost their measurement and the tuning knobs (csrc/tdv_internal.hpp:
study_env).  The study library is test / tools infrastructure: the package loads it only when TDV_LIB_VARIANT=study.

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
OBJ_STUDY = os.path.join(SRC, "_obj_study")
LIB = os.path.join(HERE, "lib3dvision_hip.so")
LIB_STUDY = os.path.join(HERE, "lib3dvision_hip_study.so")
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


def _compile(job):
    src, obj_dir, extra = job
    obj = os.path.join(obj_dir, src[:-4] + ".o")
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "tdv_hip.h")]
    if not _stale(obj, [os.path.join(SRC, src)] + headers + [os.path.abspath(__file__)]):
        return obj, ""
    r = subprocess.run([HIPCC] + FLAGS + extra + ["-c", os.path.join(SRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
    return obj, r.stderr


def _link(lib, objs):
    if _stale(lib, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lz", "-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)


def build(verbose=False, jobs=6, study=True):
    """Returns the path of the product library; study=True also builds lib3dvision_hip_study.so (-DTDV_STUDY)."""
    variants = [(OBJ, [], LIB)] + ([(OBJ_STUDY, ["-DTDV_STUDY"], LIB_STUDY)] if study else [])
    srcs = _sources()
    for d, _, _ in variants:
        os.makedirs(d, exist_ok=True)
    work = [(s, d, extra) for d, extra, _ in variants for s in srcs]
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, work))
    if verbose:
        for _, err in results:
            if err.strip():
                sys.stderr.write(err)
    for d, _, lib in variants:
        objs = [o for o, _ in results if os.path.dirname(o) == d]
        for f in os.listdir(d):          # drop objects of sources that no longer exist
            if f.endswith(".o") and os.path.join(d, f) not in objs:
                os.remove(os.path.join(d, f))
        _link(lib, objs)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
