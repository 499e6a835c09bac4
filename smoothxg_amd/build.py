"""Builds smoothxg_amd/csrc/libsxgpoa.so for gfx950 with hipcc (in-tree, travels with gpurun)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libsxgpoa.so")
# sxg_poa.hip: host side of the C ABI; kern_part*.hip: the kernel classes in parts (poa_kern_tables.hip.h), one translation
# unit each so that they compile side by side
SOURCES = ["sxg_poa.hip"] + ["kern_part%d.hip" % k for k in range(1, 10)]
DEPS = SOURCES + ["poa_kernels.hip.h", "poa_kern_tables.hip.h", "poa_dp.hip.h", "poa_dp16.hip.h", "poa_band16.hip.h", "poa_graph_dev.h",
                  "poa_bgraph_dev.h", "poa_types.h", os.path.join("..", "..", "include", "sxg_poa.h")]
OBJ_DIR = os.path.join(CSRC, "build")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 on every translation unit (in parallel: SXG_BUILD_JOBS, default = CPUs), then one link."""
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("SXG_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = max(1, int(os.environ.get("SXG_BUILD_JOBS", str(os.cpu_count() or 1))))
    objs, running, pending = [], [], list(SOURCES)
    failed = None
    while pending or running:
        while pending and len(running) < jobs and failed is None:
            src = pending.pop(0)
            obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
            cmd = [hipcc] + flags + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((subprocess.Popen(cmd), cmd))
            objs.append(obj)
        if failed is not None:
            pending = []
        if not running:   # (fewer jobs than sources and the only running one failed: nothing left to wait for)
            break
        proc, cmd = running.pop(0)
        if proc.wait() != 0 and failed is None:
            failed = cmd
    if failed is not None:
        raise subprocess.CalledProcessError(1, failed)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"]   # (RCCL is dlopen-ed on first use)
    if verbose:
        print(" ".join(cmd), flush=True)
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
