"""The C++ host mirror (glio_amd/host/glio_backend.hpp): builds with plain g++ against the C-ABI (CPU),
and on the GPU reproduces the Python/ctypes call sequence bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from glio_amd import synth
from glio_amd.host import window_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_builds_without_hip_headers():
    demo = window_io.build_demo(force=True)
    assert os.path.exists(demo)
    hdr = open(os.path.join(os.path.dirname(demo), "glio_backend.hpp")).read()
    incs = [ln for ln in hdr.splitlines() if ln.startswith("#include")]
    assert incs and not any(("hip" in i and "glio_hip.h" not in i) or "torch" in i or "ceres" in i or "Eigen" in i for i in incs)
    ldd = subprocess.run(["ldd", demo], capture_output=True, text=True).stdout
    assert "libglio_hip.so" in ldd


@pytest.mark.gpu
def test_cpp_sequence_matches_python_sequence(tmp_path):
    from glio_amd import capi
    win = synth.make_window(W=4, pts_per_scan=3000, seed=synth.SEED_BASE + 31)
    path = str(tmp_path / "win.bin")
    window_io.write_window(path, win)
    info, trans, quat = window_io.run_demo(path)
    ctx = capi.Context(win.opts)
    ctx.set_map(win.map_pts)
    kept = 0
    for s in range(win.W):
        q2, t2 = capi.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s])
        kept += ctx.associate(s, win.scans[s], q2, t2)
    ctx.load_window(win, None, use_gnss=False, use_prior=False)
    sol, summ = ctx.solve(win.init)
    assert info["kept"] == kept and info["iterations"] == summ.iterations
    assert np.isclose(info["final_cost"], summ.final_cost, rtol=1e-12)
    assert np.abs(trans - sol.trans).max() < 1e-12
    qs = sol.quat * np.where(sol.quat[:, :1] < 0, -1.0, 1.0)
    assert np.abs(quat - qs / np.linalg.norm(qs, axis=1, keepdims=True)).max() < 1e-12
    st = sol.copy(); st.quat = qs
    out = ctx.marginalize(st)
    assert info["prior"]["n"] == out["n"] and info["prior"]["n_blocks"] == len(out["blk_slot"])
    assert np.isclose(info["prior"]["jac_fro2"], (out["lin_jac"] ** 2).sum(), rtol=1e-9)
    assert np.isclose(info["prior"]["res2"], out["lin_res"] @ out["lin_res"], rtol=1e-7, atol=1e-12)
    # the resident form (C++: findCorrespondingSurfFeaturesWindow / marginalizeAndKeep / solve) against the same calls here
    poses = [capi.lidar_pose(win.opts, st.quat[s], st.trans[s]) for s in range(win.W)]
    counts = ctx.associate_window(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
    ctx.marginalize_keep(st)
    sol2, summ2 = ctx.solve(st)
    assert info["resident"]["kept"] == int(np.sum(counts)) and info["resident"]["iterations"] == summ2.iterations
    assert np.isclose(info["resident"]["final_cost"], summ2.final_cost, rtol=1e-12)
    ctx.close()


def test_batch_host_builds_against_rccl_without_hip_in_the_backend_header():
    """The sharded batch stage as a C++ program: the backend header sees only the C-ABI; the demo links librccl for the
    ncclAllReduce between linearise and step."""
    demo = window_io.build_demo_batch(force=True)
    hdr = open(os.path.join(os.path.dirname(demo), "glio_batch_backend.hpp")).read()
    incs = [ln for ln in hdr.splitlines() if ln.startswith("#include")]
    assert incs and not any(("hip" in i and "glio_hip.h" not in i) or "rccl" in i or "torch" in i for i in incs)
    ldd = subprocess.run(["ldd", demo], capture_output=True, text=True).stdout
    assert "libglio_hip.so" in ldd and "librccl" in ldd
    src = open(os.path.join(os.path.dirname(demo), "host_demo_batch.cpp")).read()
    assert "ncclAllReduce(" in src and "ncclCommInitRank(" in src


@pytest.mark.gpu
def test_cpp_batch_stage_with_rccl_allreduce_matches_python(tmp_path):
    """host_demo_batch (C++: shard, linearise, ncclAllReduce over a real RCCL communicator -- one rank on this one-GPU box --
    banded solve, damped Gauss-Newton loop) against glio_amd.batch.lm_solve on the same problem: same cost history, same poses."""
    from glio_amd import batch
    K, band, iters = 64, 6, 4
    gt, init = batch.make_poses(K, seed=17)
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 300, band, seed=17, device="cuda:0")
    path = str(tmp_path / "batch.bin")
    window_io.write_batch_problem(path, K, band, iters, init, ci, cj, cp.cpu().numpy(), nc.cpu().numpy(), score.cpu().numpy())
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    info, hist, rows = window_io.run_demo_batch(path, iters, env=env)
    assert int(info["world"]) == 1 and int(info["allreduces"]) == iters + 1
    assert abs(float(info["allreduce_MB_each"]) - batch.hg_size(K, band) * 8 / 1e6) < 1e-3
    st = batch.BatchStage(K, band, len(ci)); st.set_constraints(ci, cj, cp, nc, score)
    bufs = [st.new_hg(), st.new_hg()]; flip = [0]

    def lin(p):
        flip[0] ^= 1
        st.linearize(p, bufs[flip[0]])
        return bufs[flip[0]], float(bufs[flip[0]][-1].item())
    poses, hist_py = batch.lm_solve(lin, st.step, init, iterations=iters)
    assert np.allclose(hist, hist_py, rtol=1e-13)
    assert np.abs(rows - poses).max() < 1e-12
    st.close()


@pytest.mark.gpu
def test_cpp_full_batch_problem_rounds_match_python(tmp_path):
    """host_demo_batch on the full pose problem (plane constraints + delta_q + DD pseudoranges; BatchBackend::solveRounds: the 4
    DDpsr_threshold rounds, glio_batch_solve_tr with ncclAllReduce as the hook) against glio_amd.batch.solve_batch_rounds."""
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    K, band, iters, sr = 48, 6, 12, 3
    gt, init = batch.make_poses(K, seed=19, perturb=(0.08, 0.004))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 200, band, seed=19)
    odo = gt.copy(); odo[:, :3] += np.random.default_rng(19).normal(0, 0.02, (K, 3))
    odo[5, 3:] *= -1.0                                        # a keyframe stored with w < 0: q_i is unified, q_j is not
    dd, frame = batch.make_batch_gnss(gt, seed=19)
    path = str(tmp_path / "full.bin")
    window_io.write_batch_problem(path, K, band, iters, init, ci, cj, cp.numpy(), nc.numpy(), score.numpy(), full=(odo, sr, frame, dd))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    info, _, rows = window_io.run_demo_batch(path, iters, env=env)
    st = batch.BatchStage(K, band, len(ci)); st.set_constraints(ci, cj, cp.numpy(), nc.numpy(), score.numpy())
    poses, hist = batch.solve_batch_rounds(st, init, odo, sr, dd, frame, opts=T.batch_tr_opts(iters))
    assert len(info["rounds"]) == 4
    for a, b in zip(info["rounds"], hist):
        assert int(a["iterations"]) == b["iterations"] and int(a["termination"]) == b["termination"]
        assert np.isclose(a["final_cost"], b["final_cost"], rtol=1e-12)
    assert int(info["allreduces"]) == 0          # one rank: the library calls no collective at all (the hook is for world > 1)
    assert np.abs(rows - poses).max() < 1e-12
    st.close()


@pytest.mark.gpu
def test_cpp_keyframe_stream_equals_the_python_driver(tmp_path):
    """host_demo_stream (C++17 over glio_backend.hpp): the moving-stream keyframe cycle -- slide, new scan, local map from the resident scan, asynchronous
    window association, factor tables, solve, marginalize-and-keep -- gives the same iterations, the same kept correspondences and the same solved
    translations as the Python driver of the same C entry points (same library, same inputs)."""
    import numpy as np
    from glio_amd import capi, synth
    from glio_amd.capi import lidar_pose
    from glio_amd.host import window_io
    W, pts, NK = 5, 4096, 3
    long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 21)
    wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
    opts = wins[0].opts
    opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
    opts.max_map_points = 1 << 16
    path = str(tmp_path / "stream.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    got = window_io.run_demo_stream(path, search_range=2, stream_draws=False)
    # (the selection on the association's stream from raw draws made before the counts exist: the same pairs found, the same 25 held per pair)
    got_sd = window_io.run_demo_stream(path, search_range=2, stream_draws=True)
    assert got_sd["batch_records_found"] == got["batch_records_found"] and got_sd["batch_records_held"] == got["batch_records_held"] and got_sd["iterations"] == got["iterations"]
    # (the next keyframe's cloud sent during the call, beside an asynchronous marginalization -- glio_set_scan_ahead, glio_marginalize_keep_async / _finish: the
    #  same stream, the same numbers)
    got_ah = window_io.run_demo_stream(path, search_range=2, ahead=True)
    got_ma = window_io.run_demo_stream(path, search_range=2, ahead=True, map_ahead=True)      # ... and the next call's local map built during the tail too
    for key in ("iterations", "correspondences_kept", "batch_records_found", "batch_records_held", "last_trans", "last_quat", "trans_checksum", "map_points"):
        assert got_ah[key] == got_sd[key], key
        assert got_ma[key] == got_sd[key], key
    ctx = capi.Context(opts)
    ctx.localmap_config(50, 0.4, pts)
    tlb = np.array(opts.t_lb, np.float32)
    for j in range(W - 1):
        c = long.scans[j].copy(); c[:, :3] -= tlb
        ctx.localmap_push(np.ascontiguousarray(c), long.gt.quat[j], long.gt.trans[j])
    for s in range(W - 1):
        ctx.set_scan(s + 1, long.scans[s])
    ctx.set_prior(None)
    iters, kept, checksum = [], [], 0.0
    sol = None
    # batchFeatureAssociation at the end of every keyframe call (search_range 2 here so that the 8-keyframe stream reaches it; selection 25 per pair)
    from glio_amd import batch, sliding
    ba = batch.BatchAssociation(W + NK, pts, (NK + 2) * 4 * pts)
    kba = sliding.KeyframeBatchAssociation(ba, search_range=2, feature_res_num=25, rng=np.random.default_rng(1))
    kf_poses = np.c_[long.gt.trans, long.gt.quat][:W + NK].copy()
    for j in range(W - 1):
        c = long.scans[j].copy(); c[:, :3] -= tlb
        ba.set_frame(j, np.ascontiguousarray(c))
    bfound, bheld = [], []
    for j in range(NK + 1):
        win = wins[j]
        state = win.init.copy()
        if j > 0:
            state.trans[:-1], state.quat[:-1], state.speed_bias[:-1] = sol.trans[1:], sol.quat[1:], sol.speed_bias[1:]
        new = j + W - 1
        ctx.slide_window(); ctx.set_scan(W - 1, long.scans[new])
        ctx.localmap_push_scan(W - 1, tlb, long.gt.quat[new], long.gt.trans[new]); ctx.localmap_build()
        poses = [lidar_pose(opts, state.quat[s], state.trans[s]) for s in range(W)]
        ctx.associate_window_async(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
        ctx.set_imu(win.preints); ctx.set_gnss(win.frame, win.dd, win.dop)
        counts = ctx.associate_window_counts()
        sol, summ = ctx.solve(state)
        usol = sliding.unify_quaternions(sol.copy())
        kf_poses[j:j + W, :3] = usol.trans; kf_poses[j:j + W, 3:] = usol.quat
        ba.set_frame_from_scan(new, ctx, W - 1, tlb)
        kba.enqueue(new + 1, kf_poses)
        ctx.marginalize_keep(sol)
        found = kba.finish()
        if j > 0:
            iters.append(int(summ.iterations)); kept.append(int(np.sum(counts))); checksum += float(np.sum(sol.trans))
            bfound.append(int(np.sum(found))); bheld.append(int(ba.total))
    ctx.close(); ba.close()
    assert got["iterations"] == iters and got["correspondences_kept"] == kept
    assert abs(got["trans_checksum"] - checksum) <= 1e-9 * abs(checksum)
    # the pair searches find the same records in both hosts (the 25 kept per pair are each host's own random draws: same counts)
    assert got["batch_records_found"] == bfound and sum(bfound) > 4 * 1000 and got["batch_records_held"] == bheld and bheld[-1] == 25 * len(kba.counts)


@pytest.mark.gpu
def test_cpp_batch_association_and_rounds_match_python(tmp_path):
    """host_demo_batch in association mode: glio::BatchAssociationBackend + glio::RoundsAssociation (one resident association object, its arrays laid out
    [interior | front ends | back ends], the ends truncated and re-searched every round) as the `reassociate` hook of BatchBackend::solveRounds, against the
    Python driver (batch.RoundsAssociation: three association objects + device copies, batch.solve_batch_rounds): the same constraint count, the same
    rounds (iterations, termination, costs), the same poses."""
    from glio_amd import batch, synth
    from glio_amd import ctypes_types as T
    K, sr, pts, iters = 12, 2, 3000, 10
    band = 2 * sr
    win = synth.make_window(W=K, pts_per_scan=pts, seed=synth.SEED_BASE + 53, perturb=(0.03, 0.2, 0.0), scan_radius=14.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    clouds = []
    for s in range(K):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        clouds.append(np.ascontiguousarray(c))
    gt = np.c_[win.gt.trans, win.gt.quat]
    init = np.c_[win.init.trans, win.init.quat]
    odo = gt.copy(); odo[:, :3] += np.random.default_rng(53).normal(0, 0.02, (K, 3))
    dd, frame = batch.make_batch_gnss(gt, seed=53)
    path = str(tmp_path / "assoc.bin")
    window_io.write_batch_assoc_problem(path, K, band, iters, init, odo, sr, frame, dd, clouds, 4096)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    info, _, rows = window_io.run_demo_batch(path, iters, env=env)
    ci, _ = batch.pair_list(K, sr)
    st = batch.BatchStage(K, band, len(ci) * 4096)
    ra = batch.RoundsAssociation(st, clouds, sr, 4096)
    ra.start(init)
    poses, hist = batch.solve_batch_rounds(st, init, odo, sr, dd, frame, reassociate=ra, opts=T.batch_tr_opts(iters))
    assert int(info["assoc"]["pairs"]) == len(ci) and int(info["assoc"]["constraints"]) == ra.n_constraints > 10000
    assert len(info["rounds"]) == 4
    for a, b in zip(info["rounds"], hist):
        assert int(a["iterations"]) == b["iterations"] and int(a["termination"]) == b["termination"]
        assert np.isclose(a["final_cost"], b["final_cost"], rtol=1e-8)      # (the two hosts lay the end regions out at different offsets: other summation order)
    assert np.abs(rows - poses).max() < 1e-8
    ra.close(); st.close()


@pytest.mark.gpu
def test_released_configuration_cpp_equals_python_equals_oracle(tmp_path):
    """The configuration the reference ships (config_urban_hk.yaml:60-104: slide_window_width 5, feature_res_num 100, random_select) replayed from C++:
    glio::SlidingWindowBackend::featureSelection behind every slot's search (Estimator.cpp:2222-2223) with the draws of a generator both hosts share
    (sliding.TableRng / host_demo_stream draws=) -- same iterations, same residual counts (100 per slot), same solved poses as the Python driver; and the
    last keyframe's solve, on the correspondences the selection left and the prior the device's marginalization produced, equals the oracle's."""
    import numpy as np
    from glio_amd import capi, sliding, synth
    from glio_amd.capi import lidar_pose
    from glio_amd.host import window_io
    from oracle import pyoracle as po
    W, pts, NK, RES = 5, 4096, 3, 100
    long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=False, with_prior=False, seed=synth.SEED_BASE + 23)
    wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
    opts = wins[0].opts
    opts.max_map_points = 1 << 16
    path, dpath = str(tmp_path / "stream.bin"), str(tmp_path / "draws.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    table = np.random.default_rng(5).integers(0, 2 ** 62, 4096, dtype=np.uint64)
    table.tofile(dpath)
    got = window_io.run_demo_stream(path, search_range=6, feature_res_num=RES, draws=dpath)      # (search_range 6: the 8-keyframe stream never reaches the batch association)
    # the C++ host selects the whole window in one call (glio_select_correspondences_window); W per-slot calls keep the same records
    got_ps = window_io.run_demo_stream(path, search_range=6, feature_res_num=RES, draws=dpath, per_slot=True)
    assert got_ps["last_trans"] == got["last_trans"] and got_ps["last_quat"] == got["last_quat"] and got_ps["iterations"] == got["iterations"]
    rng = sliding.TableRng(table)
    ctx = capi.Context(opts)
    ctx.localmap_config(50, 0.4, pts)
    tlb = np.array(opts.t_lb, np.float32)
    for j in range(W - 1):
        c = long.scans[j].copy(); c[:, :3] -= tlb
        ctx.localmap_push(np.ascontiguousarray(c), long.gt.quat[j], long.gt.trans[j])
    for s in range(W - 1):
        ctx.set_scan(s + 1, long.scans[s])
    ctx.set_prior(None)
    iters, kept, checksum, sol, prior = [], [], 0.0, None, None
    for j in range(NK + 1):
        win = wins[j]
        state = win.init.copy()
        if j > 0:
            state.trans[:-1], state.quat[:-1], state.speed_bias[:-1] = sol.trans[1:], sol.quat[1:], sol.speed_bias[1:]
        new = j + W - 1
        ctx.slide_window(); ctx.set_scan(W - 1, long.scans[new])
        ctx.localmap_push_scan(W - 1, tlb, long.gt.quat[new], long.gt.trans[new]); ctx.localmap_build()
        poses = [lidar_pose(opts, state.quat[s], state.trans[s]) for s in range(W)]
        ctx.associate_window_async(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
        ctx.set_imu(win.preints); ctx.set_gnss(win.frame, win.dd, win.dop)
        counts = list(ctx.associate_window_counts())
        assert min(counts) > 10 * RES                                   # the selection has something to select from
        if j % 2:                                                       # the Python host alternates between the two forms
            counts = sliding.feature_selection_window(ctx, counts, RES, rng)
        else:
            for s in range(W):
                counts[s] = sliding.feature_selection(ctx, s, counts[s], RES, rng)
        assert counts == [RES] * W
        corr = [ctx.get_correspondences(s) for s in range(W)]
        sol, summ = ctx.solve(state)
        if j == NK:                                                     # the oracle on the same buffers: selected correspondences, the device's prior
            ow = synth.sub_window(long, j, W); ow.prior = prior
            so, summ_o = po.Problem(ow, corr).solve(state)
            assert summ_o.iterations == summ.iterations
            assert np.linalg.norm(sol.trans - so.trans, axis=1).max() < 1e-9 and np.abs(sol.quat - so.quat).max() < 1e-10
        prior = ctx.marginalize(sol) if j == NK - 1 else None          # (read back for the oracle's side of the last keyframe)
        usol = sliding.unify_quaternions(sol.copy())
        ctx.marginalize_keep(sol)
        if j > 0:
            iters.append(int(summ.iterations)); kept.append(int(np.sum(counts))); checksum += float(np.sum(sol.trans))
    ctx.close()
    assert got["feature_res_num"] == RES and got["iterations"] == iters and got["correspondences_kept"] == kept == [RES * W] * NK
    assert abs(got["trans_checksum"] - checksum) <= 1e-12 * abs(checksum)
    assert np.array_equal(np.array(got["last_trans"]).reshape(W, 3), sol.trans)          # same library, same inputs, same draws: the same bits
    assert np.array_equal(np.array(got["last_quat"]).reshape(W, 4), usol.quat)


def test_feature_selection_draws_cpp_equals_python(tmp_path):
    """glio::featureSelectionDraws against sliding.feature_selection_draws on a shared table of numbers: the early return (count - 1 < feature_res_num keeps the
    set whole, quirk Q9), random_select = false (empties the slot), and draws without repetition in the order drawn."""
    import subprocess
    import numpy as np
    from glio_amd import sliding
    here = os.path.join(ROOT, "glio_amd", "host")
    src = tmp_path / "fs.cpp"
    src.write_text('#include "glio_backend.hpp"\n#include <cstdio>\n#include <cstdlib>\n'
                   'int main(int argc, char** argv) { const long count = atol(argv[1]); const int res = atoi(argv[2]); const bool rs = atoi(argv[3]) != 0; unsigned long long k = 0;\n'
                   '  std::vector<unsigned long long> t; for (int a = 4; a < argc; ++a) t.push_back(strtoull(argv[a], nullptr, 10));\n'
                   '  std::vector<int32_t> kept; const bool changed = glio::featureSelectionDraws(count, res, [&](uint64_t n) -> uint64_t { return t[k++ % t.size()] % n; }, rs, kept);\n'
                   '  printf("%d", changed ? 1 : 0); for (int32_t v : kept) printf(" %d", v); printf("\\n"); return 0; }\n')
    exe = str(tmp_path / "fs")
    subprocess.check_call(["g++", "-std=c++14", "-O1", str(src), "-I" + here, "-I" + os.path.join(ROOT, "include"), "-o", exe])
    table = np.random.default_rng(9).integers(0, 2 ** 62, 300, dtype=np.uint64)
    for count, res, rs in ((5000, 100, 1), (101, 100, 1), (100, 100, 1), (0, 100, 1), (102, 100, 1), (4096, 25, 1), (4096, 100, 0), (60, 100, 0)):
        out = subprocess.run([exe, str(count), str(res), str(rs)] + [str(int(v)) for v in table], capture_output=True, text=True, check=True).stdout.split()
        want = sliding.feature_selection_draws(count, res, sliding.TableRng(table), random_select=bool(rs))
        if want is None:
            assert out == ["0"], (count, res, rs, out)
        else:
            assert out[0] == "1" and [int(v) for v in out[1:]] == want.tolist(), (count, res, rs)
            assert len(set(want.tolist())) == len(want) and (len(want) == 0 or (0 <= want.min() and want.max() < count))
