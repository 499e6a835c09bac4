"""Builds smoothxg_amd/csrc/libsxgpoa.so for gfx950 with hipcc (in-tree, travels with gpurun)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libsxgpoa.so")
SOURCES = ["sxg_poa.hip"]
DEPS = SOURCES + ["poa_dp.hip.h", "poa_dp16.hip.h", "poa_band16.hip.h", "poa_graph_dev.h", "poa_bgraph_dev.h", "poa_types.h",
                  os.path.join("..", "..", "include", "sxg_poa.h")]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO] + \
          os.environ.get("SXG_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]   # (RCCL is dlopen-ed on first use)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


SMOOTH_SO = os.path.join(CSRC, "libsxgsmooth.so")
SMOOTH_DEPS = ["sxg_smooth.cpp", os.path.join("..", "..", "include", "sxg_smooth.h"),
               os.path.join("..", "..", "include", "sxg_poa.h")]


def build_smooth(force=False, verbose=False):
    """Host-side rows (collection, block graphs, lacing, GFA): plain g++, no HIP."""
    if not force and os.path.exists(SMOOTH_SO) and \
            all(os.path.getmtime(os.path.join(CSRC, d)) <= os.path.getmtime(SMOOTH_SO) for d in SMOOTH_DEPS):
        return SMOOTH_SO
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wall", "-o", SMOOTH_SO,
           os.path.join(CSRC, "sxg_smooth.cpp")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SMOOTH_SO


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_smooth(force=True, verbose=True)
