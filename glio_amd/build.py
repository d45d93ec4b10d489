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
    """hipcc --offload-arch=gfx950: cross-compiles without a GPU.  Every source becomes its own object (recompiled only when
    it or a header changed, all stale ones in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(os.path.dirname(LIB), "obj")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"] + \
        (["-DGLIO_DEV_STAMPS"] if os.environ.get("GLIO_DEV_STAMPS") == "1" else []) + os.environ.get("GLIO_EXTRA_DEFS", "").split()
    tag = os.path.join(objdir, "flags.txt")
    flag_str = " ".join(flags)
    if os.path.exists(LIB) and os.path.exists(tag) and open(tag).read() != flag_str:
        force = True                  # built with other defines (e.g. GLIO_DEV_STAMPS): everything again
    if not force and not _stale():
        return LIB
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in HEADERS)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append(["hipcc"] + flags + ["-c", src, "-o", obj])
    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    open(tag, "w").write(flag_str)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in __import__("sys").argv, verbose=True))
