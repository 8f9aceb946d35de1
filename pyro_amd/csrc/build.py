"""Build libpyro_amd.so (the C-ABI shared library declared in include/pyro_amd.h) in-tree.

    python -m pyro_amd.csrc.build [--force]

hipcc cross-compiles for gfx950 without a GPU present. The .so is written to
pyro_amd/lib/libpyro_amd.so (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpyro_amd.so")
OBJ_DIR = os.path.join(HERE, "build")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file additions.  glm.hip: the matrix-core GLM kernels interleave f32 element-wise math with bf16
# MFMAs; packed f32 instructions (which the SLP vectoriser forms from adjacent scalar operations) do
# not overlap the MFMAs on gfx950 and cost 2-4x a plain instruction next to them
# (tools/probes/issue_probe.hip), so that file is built without SLP vectorisation.
FILE_FLAGS = {"glm.hip": ["-fno-slp-vectorize"]}
if os.environ.get("PA_CHAIN_LATENCY_PROBE"):          # developer probe (tools/chain_stamps.py)
    FILE_FLAGS["chain.hip"] = ["-DPA_CHAIN_LATENCY_PROBE"]


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "pyro_amd.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    spath = os.path.join(HERE, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(spath), hdr_mtime)):
        return obj, False
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", spath, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, res.stdout, res.stderr))
    return obj, True


def build_variant(suffix, extra_flags, verbose=False):
    """A second library next to the shipped one, built with extra compiler flags (developer A/B
    runs: ``PYRO_AMD_LIB=pyro_amd/lib/libpyro_amd_<suffix>.so python bench.py``)."""
    global OBJ_DIR, LIB_PATH, FLAGS
    keep = (OBJ_DIR, LIB_PATH, FLAGS)
    OBJ_DIR = os.path.join(HERE, "build_" + suffix)
    LIB_PATH = os.path.join(LIB_DIR, "libpyro_amd_%s.so" % suffix)
    FLAGS = FLAGS + list(extra_flags)
    try:
        return build_library(force=False, verbose=verbose)
    finally:
        OBJ_DIR, LIB_PATH, FLAGS = keep


def build_library(force=False, verbose=False):
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB_PATH):
            return LIB_PATH  # GPU box without a toolchain: use the shipped binary
        raise RuntimeError("hipcc not found at %s and no prebuilt %s" % (HIPCC, LIB_PATH))
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    hdr_mtime = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr_mtime), sources()))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or force or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + \
            ["-L/opt/rocm/lib", "-lhiprtc"]          # (csrc/rtc.hip: run-time compiled element-wise kernels)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
        if verbose:
            print("linked", LIB_PATH)
    return LIB_PATH


TORCH_LIB_PATH = os.path.join(LIB_DIR, "libpyro_amd_torch.so")


def build_torch_ops(force=False, verbose=False):
    """libpyro_amd_torch.so: the TORCH_LIBRARY registration of csrc/torch_ops.cpp (host-only C++ over
    the C-ABI; g++ against torch's headers, linked to libpyro_amd.so next to it)."""
    src = os.path.join(HERE, "torch_ops.cpp")
    cxx = os.environ.get("CXX", "g++")
    import shutil
    if shutil.which(cxx) is None:
        if os.path.exists(TORCH_LIB_PATH):
            return TORCH_LIB_PATH
        raise RuntimeError("no C++ compiler (%s) and no prebuilt %s" % (cxx, TORCH_LIB_PATH))
    deps = max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "pyro_amd.h")))
    if not force and os.path.exists(TORCH_LIB_PATH) and os.path.getmtime(TORCH_LIB_PATH) >= deps:
        return TORCH_LIB_PATH
    import torch
    from torch.utils import cpp_extension
    tlib = cpp_extension.library_paths()[0]
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in cpp_extension.include_paths()]
    cmd += ["-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), src, "-o", TORCH_LIB_PATH,
            "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-L" + LIB_DIR, "-lpyro_amd",
            "-Wl,-rpath,$ORIGIN"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("torch_ops.cpp failed to build:\n%s\n%s" % (res.stdout, res.stderr))
    if verbose:
        print("linked", TORCH_LIB_PATH)
    return TORCH_LIB_PATH


if __name__ == "__main__":
    if "--variant" in sys.argv:          # --variant <suffix> <flag> [<flag> ...]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
        print(build_torch_ops(force="--force" in sys.argv, verbose=True))
