"""Flat binary window file for glio_amd/host/host_demo (the C++ call sequence of the hot path)."""
import ctypes as C
import os
import subprocess

import numpy as np

from .. import ctypes_types as T
from .. import synth

HERE = os.path.dirname(os.path.abspath(__file__))
DEMO = os.path.join(HERE, "host_demo")
_ABI_HEADERS = [os.path.join(HERE, "..", "..", "include", h) for h in ("glio_hip.h", "glio_types.h")]      # a changed struct must rebuild the demos


def build_demo(force=False):
    src = [os.path.join(HERE, "host_demo.cpp"), os.path.join(HERE, "glio_backend.hpp")] + _ABI_HEADERS
    if force or not os.path.exists(DEMO) or any(os.path.getmtime(s) > os.path.getmtime(DEMO) for s in src):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", src[0], "-I" + os.path.join(HERE, "..", "..", "include"),
                               "-L" + os.path.join(HERE, "..", "lib"), "-lglio_hip", "-Wl,-rpath,$ORIGIN/../lib", "-o", DEMO])
    return DEMO


def write_window(path, win):
    """opts | n_map n_imu 0 0 | map | trans quat speed_bias | preints | per slot: n, scan"""
    with open(path, "wb") as f:
        f.write(bytes(win.opts))
        f.write(np.array([len(win.map_pts), len(win.preints), 0, 0], np.int32).tobytes())
        f.write(np.ascontiguousarray(win.map_pts, np.float32).tobytes())
        f.write(win.init.trans.tobytes()); f.write(win.init.quat.tobytes()); f.write(win.init.speed_bias.tobytes())
        arr = T.preint_array(len(win.preints))
        for k, p in enumerate(win.preints):
            synth.fill_preint(arr[k], p)
        if win.preints:
            f.write(bytes(arr)[:C.sizeof(T.GlioPreint) * len(win.preints)])
        for s in range(win.W):
            f.write(np.array([len(win.scans[s])], np.int32).tobytes())
            f.write(np.ascontiguousarray(win.scans[s], np.float32).tobytes())


def run_demo(path):
    out = subprocess.run([build_demo(), path], capture_output=True, text=True, check=True).stdout.splitlines()
    head = out[0].split()
    info = dict(kept=int(head[1]), iterations=int(head[3]), termination=int(head[5]), initial_cost=float(head[7]), final_cost=float(head[9]))
    rows = np.array([[float(x) for x in ln.split()[2:]] for ln in out[1:] if ln.startswith("kf ")])
    for ln in out[1:]:
        if ln.startswith("prior "):
            p = ln.split()
            info["prior"] = dict(n=int(p[1]), n_blocks=int(p[2]), jac_fro2=float(p[3]), res2=float(p[4]))
        if ln.startswith("resident "):
            p = ln.split()
            info["resident"] = dict(kept=int(p[1]), iterations=int(p[2]), final_cost=float(p[3]))
    return info, rows[:, :3], rows[:, 3:]


DEMO_BATCH = os.path.join(HERE, "host_demo_batch")


def build_demo_batch(force=False):
    """The C++ sharded batch stage (links librccl: ncclAllReduce between linearise and step)."""
    src = [os.path.join(HERE, "host_demo_batch.cpp"), os.path.join(HERE, "glio_batch_backend.hpp")] + _ABI_HEADERS
    if force or not os.path.exists(DEMO_BATCH) or any(os.path.getmtime(s) > os.path.getmtime(DEMO_BATCH) for s in src):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", src[0], "-I" + os.path.join(HERE, "..", "..", "include"),
                               "-I/opt/rocm/include", "-L" + os.path.join(HERE, "..", "lib"), "-lglio_hip", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-lpthread",
                               "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib", "-o", DEMO_BATCH])
    return DEMO_BATCH


def write_batch_problem(path, K, band, iterations, poses, ci, cj, cp, nc, score, full=None):
    """full = (odo [K][7], search_range, frame, dd list): appends the data of the full pose problem (see host_demo_batch.cpp)."""
    with open(path, "wb") as f:
        f.write(np.array([K, band, iterations, 1 if full else 0], np.int32).tobytes())
        f.write(np.array([len(ci)], np.int64).tobytes())
        f.write(np.ascontiguousarray(poses, np.float64).tobytes())
        f.write(np.ascontiguousarray(ci, np.int32).tobytes()); f.write(np.ascontiguousarray(cj, np.int32).tobytes())
        f.write(np.ascontiguousarray(cp, np.float32).tobytes()); f.write(np.ascontiguousarray(nc, np.float64).tobytes())
        f.write(np.ascontiguousarray(score, np.float64).tobytes())
        if full:
            odo, search_range, frame, dd = full
            f.write(np.ascontiguousarray(odo, np.float64).tobytes())
            f.write(np.array([search_range, len(dd)], np.int32).tobytes())
            f.write(bytes(frame))
            for d in dd:
                f.write(bytes(d))


def write_batch_assoc_problem(path, K, band, iterations, poses, odo, search_range, frame, dd, clouds, max_points):
    """The association mode of host_demo_batch.cpp (header word 3 = 2): no constraints, the keyframe clouds instead."""
    with open(path, "wb") as f:
        f.write(np.array([K, band, iterations, 2], np.int32).tobytes())
        f.write(np.array([0], np.int64).tobytes())
        f.write(np.ascontiguousarray(poses, np.float64).tobytes())
        f.write(np.ascontiguousarray(odo, np.float64).tobytes())
        f.write(np.array([search_range, len(dd), max_points, 0], np.int32).tobytes())
        f.write(bytes(frame))
        for d in dd:
            f.write(bytes(d))
        for c in clouds:
            c = np.ascontiguousarray(c, np.float32)
            f.write(np.array([len(c)], np.int32).tobytes()); f.write(c.tobytes())


def run_demo_batch(path, iterations=None, env=None):
    cmd = [build_demo_batch(), path] + ([str(iterations)] if iterations is not None else [])
    out = subprocess.run(cmd, capture_output=True, text=True, check=True, env=env).stdout.splitlines()
    # (RCCL prints its own version banner on stdout: pick our lines by their first word)
    head = next(ln for ln in out if ln.startswith("batch ")).split()
    info = {head[i]: head[i + 1] for i in range(1, len(head) - 1, 2)}
    hist = [float(x) for x in next(ln for ln in out if ln.split()[:1] == ["cost"]).split()[1:]]
    rows = np.array([[float(x) for x in ln.split()[2:]] for ln in out if ln.startswith("kf ")])
    info["raw"] = [ln for ln in out if not ln.startswith("kf ")]
    for ln in out:
        if ln.startswith("assoc "):
            w = ln.split()
            info["assoc"] = {w[i]: float(w[i + 1]) for i in range(1, len(w) - 1, 2)}
    info["rounds"] = []
    for ln in out:
        if ln.startswith("round "):
            w = ln.split()
            info["rounds"].append({w[i]: float(w[i + 1]) for i in range(1, len(w) - 1, 2)})
    return info, hist, rows


DEMO_STREAM = os.path.join(HERE, "host_demo_stream")


def build_demo_stream(force=False):
    """The C++ moving-stream keyframe cycle (host_demo_stream.cpp): what bench.py times as keyframe_pipeline_cpp."""
    src = [os.path.join(HERE, "host_demo_stream.cpp"), os.path.join(HERE, "glio_backend.hpp"), os.path.join(HERE, "glio_batch_backend.hpp")] + _ABI_HEADERS
    if force or not os.path.exists(DEMO_STREAM) or any(os.path.getmtime(s) > os.path.getmtime(DEMO_STREAM) for s in src):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", src[0], "-I" + os.path.join(HERE, "..", "..", "include"),
                               "-L" + os.path.join(HERE, "..", "lib"), "-lglio_hip", "-Wl,-rpath,$ORIGIN/../lib", "-o", DEMO_STREAM])
    return DEMO_STREAM


def write_stream(path, long, wins, W, n_keyframes, pts, lm_width=50, leaf=0.4, batch_res_num=0):
    """The moving stream of bench.py's keyframe_stream as a flat file: opts | n_keyframes pts lm_width 0 | leaf tlb[3] | the W + n_keyframes scans |
    ground-truth q [.][4], t [.][3] (the poses the local map is pushed with) | per window j = 0..n_keyframes: n_ddt n_preint n_dd n_dop, init trans quat
    speed_bias rcv_ddt, preints, GNSS frame, DD factors, Doppler factors."""
    opts = wins[0].opts
    with open(path, "wb") as f:
        f.write(bytes(opts))
        f.write(np.array([n_keyframes, pts, lm_width, batch_res_num], np.int32).tobytes())      # (batch_feature_res_num of the per-keyframe batch association; 0 = the yaml's 25)
        f.write(np.array([leaf] + list(opts.t_lb), np.float32).tobytes())
        for j in range(W + n_keyframes):
            f.write(np.ascontiguousarray(long.scans[j], np.float32).tobytes())
        f.write(np.ascontiguousarray(long.gt.quat[:W + n_keyframes], np.float64).tobytes())
        f.write(np.ascontiguousarray(long.gt.trans[:W + n_keyframes], np.float64).tobytes())
        for win in wins:
            st = win.init
            f.write(np.array([st.n_ddt, len(win.preints), len(win.dd), len(win.dop)], np.int32).tobytes())
            f.write(np.ascontiguousarray(st.trans, np.float64).tobytes()); f.write(np.ascontiguousarray(st.quat, np.float64).tobytes())
            f.write(np.ascontiguousarray(st.speed_bias, np.float64).tobytes()); f.write(np.ascontiguousarray(np.asarray(st.rcv_ddt)[:st.n_ddt], np.float64).tobytes())
            arr = T.preint_array(len(win.preints))
            for k, p in enumerate(win.preints):
                synth.fill_preint(arr[k], p)
            if win.preints:
                f.write(bytes(arr)[:C.sizeof(T.GlioPreint) * len(win.preints)])
            f.write(bytes(win.frame) if win.frame is not None else bytes(C.sizeof(T.GlioGnssFrame)))
            for d in win.dd:
                f.write(bytes(d))
            for d in win.dop:
                f.write(bytes(d))


def run_demo_stream(path, device=0, env=None, search_range=6, defer=False, feature_res_num=0, draws=None, timed=None, per_slot=False, sleep_ms=0, sleep_at=0, stream_draws=True, prepare_early=True, ahead=False, map_ahead=False):
    """feature_res_num > 0: featureSelection behind every slot's search (Estimator.cpp:2223); draws: file of uint64 both hosts draw from (sliding.TableRng);
    timed: only the last `timed` keyframes enter the time averages"""
    import json
    cmd = [build_demo_stream(), path, str(device), str(search_range), str(int(defer))]
    if feature_res_num:
        cmd.append(f"res={int(feature_res_num)}")
    if draws:
        cmd.append(f"draws={draws}")
    if timed:
        cmd.append(f"timed={int(timed)}")
    if per_slot:
        cmd.append("per_slot=1")
    if sleep_ms:
        cmd += [f"sleep_ms={int(sleep_ms)}", f"sleep_at={int(sleep_at)}"]
    cmd.append("stream_draws=%d" % (1 if stream_draws else 0))
    if not prepare_early:
        cmd.append("prepare_early=0")
    if ahead:
        cmd.append("ahead=1")
    if map_ahead:
        cmd.append("map_ahead=1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError("host_demo_stream failed (%d): %s" % (r.returncode, (r.stderr or r.stdout)[-600:]))
    return json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("{")))


DEMO_ODOMETRY = os.path.join(HERE, "host_demo_odometry")


def build_demo_odometry(force=False):
    """The C++ front end (host_demo_odometry.cpp over glio::ScanToMapOdometry)."""
    src = [os.path.join(HERE, "host_demo_odometry.cpp"), os.path.join(HERE, "glio_backend.hpp")] + _ABI_HEADERS
    if force or not os.path.exists(DEMO_ODOMETRY) or any(os.path.getmtime(s) > os.path.getmtime(DEMO_ODOMETRY) for s in src):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", src[0], "-I" + os.path.join(HERE, "..", "..", "include"),
                               "-L" + os.path.join(HERE, "..", "lib"), "-lglio_hip", "-Wl,-rpath,$ORIGIN/../lib", "-o", DEMO_ODOMETRY])
    return DEMO_ODOMETRY


def write_odometry_stream(path, opts, scans, scan_match_cnt=1):
    """opts | n_scans scan_match_cnt 0 0 | per scan: n, points xyzi (the downsampled surf cloud of each LiDAR frame, body frame)"""
    with open(path, "wb") as f:
        f.write(bytes(opts))
        f.write(np.array([len(scans), scan_match_cnt, 0, 0], np.int32).tobytes())
        for sc in scans:
            sc = np.ascontiguousarray(sc, np.float32)
            f.write(np.array([len(sc)], np.int32).tobytes()); f.write(sc.tobytes())


def run_demo_odometry(path, device=0, env=None):
    """-> (poses [n][7] q then t, per-scan dicts, {"ms_per_scan": ...})"""
    import json
    r = subprocess.run([build_demo_odometry(), path, str(device)], capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError("host_demo_odometry failed (%d): %s" % (r.returncode, (r.stderr or r.stdout)[-600:]))
    poses, rows = [], []
    for ln in r.stdout.splitlines():
        if ln.startswith("pose "):
            w = ln.split()
            poses.append([float(x) for x in w[2:9]])
            rows.append({"rounds": int(w[9]), "kept": int(w[10]), "iterations": int(w[11]), "final_cost": float(w[12]), "map_points": int(w[13])})
    info = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("{")))
    return np.array(poses), rows, info
