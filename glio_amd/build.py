"""Builds glio_amd/lib/libglio_hip.so from the hand-written HIP sources (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libglio_hip.so")
SOURCES = ["lidar_kernels.hip", "factor_kernels.hip", "solver_kernels.hip", "assoc_kernels.hip", "batch_kernels.hip", "batch_solve_kernels.hip", "batch_tr_kernels.hip", "localmap_kernels.hip", "eval_kernels.hip", "capi.hip"]
HEADERS = [os.path.join(CSRC, "glio_device.h"), os.path.join(CSRC, "k3_device.h"), os.path.join(CSRC, "batch_device.h"), os.path.join(HERE, "..", "include", "glio_hip.h"), os.path.join(HERE, "..", "include", "glio_types.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: cross-compiles without a GPU (seconds per file)."""
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
           ] + (["-DGLIO_DEV_STAMPS"] if os.environ.get("GLIO_DEV_STAMPS") == "1" else []) + os.environ.get("GLIO_EXTRA_DEFS", "").split() + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
