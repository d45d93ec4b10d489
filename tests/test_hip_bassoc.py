"""GPU parity of the batch association: glio_bassoc_* vs the oracle's restatement of
findGlobalCorrespondingSurfFeaturesAdd_Batch (reference GLIO/src/Estimator.cpp:3808-3892).  Bit-exact: the kept set,
its order, the float point, the fp64 [local normal | centroid] record and the score are compared with ==."""
import numpy as np
import pytest

from glio_amd import batch, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    """8 keyframes of the synthetic street scene with slightly wrong poses (as pose_info_keyframe would hold)."""
    win = synth.make_window(W=8, pts_per_scan=4000, seed=synth.SEED_BASE + 51, perturb=(0.03, 0.2, 0.0), scan_radius=14.0, map_density=1.0)
    o = win.opts
    tlb = np.array(o.t_lb)
    scans = []
    for s in range(win.W):                       # the batch factor applies no extrinsic (quirk Q10): body-frame clouds
        sc = win.scans[s].copy()
        sc[:, :3] -= tlb.astype(np.float32)
        scans.append(np.ascontiguousarray(sc))
    poses = np.c_[win.init.trans, win.init.quat]
    return scans, poses


def test_pairs_match_oracle_bit_exact(frames):
    from oracle import pyoracle as po
    scans, poses = frames
    K = len(scans)
    ci, cj = batch.pair_list(K, 2)
    ba = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ba.set_frame(k, scans[k])
    counts, total = ba.run(poses, ci, cj)
    cp, nc, sc = ba.read()
    assert total == counts.sum() and total > 2000, "the synthetic scene must produce constraints"
    first = 0
    for p in range(len(ci)):
        ocp, onc, osc, _ = po.associate_pair(scans[ci[p]], poses[ci[p]], scans[cj[p]], poses[cj[p]])
        n = counts[p]
        assert n == len(osc), f"pair ({ci[p]},{cj[p]}): kept {n} vs oracle {len(osc)}"
        assert np.array_equal(cp[first:first + n], ocp)
        assert np.array_equal(nc[first:first + n], onc)
        assert np.array_equal(sc[first:first + n], osc)
        first += n
    # records are what BinaryLidarPlaneNormFactor expects: unit normals, weights in (0.3, 1]
    assert np.allclose(np.linalg.norm(nc[:, :3], axis=1), 1.0, atol=1e-12)
    assert sc.min() > 2.5 * 0.3 and sc.max() <= 2.5
    ba.close()


def test_empty_frame_and_repeatability(frames):
    scans, poses = frames
    K = len(scans)
    ci, cj = batch.pair_list(K, 1)
    ba = batch.BatchAssociation(K, 4096, 200000)
    for k in range(K):
        ba.set_frame(k, scans[k] if k != 3 else scans[k][:0])
    c1, t1 = ba.run(poses, ci, cj)
    r1 = [a.copy() for a in ba.read()]
    assert all(c1[p] == 0 for p in range(len(ci)) if ci[p] == 3 or cj[p] == 3)
    c2, t2 = ba.run(poses, ci, cj)
    r2 = ba.read()
    assert t1 == t2 and np.array_equal(c1, c2) and all(np.array_equal(a, b) for a, b in zip(r1, r2))
    ba.close()


def test_association_feeds_batch_stage(frames):
    """End to end on the device: associate -> K8 linearise; H, g, cost equal the oracle's on the oracle's constraints."""
    from oracle import pyoracle as po
    scans, poses = frames
    K, rng = len(scans), 2
    ci, cj = batch.pair_list(K, rng)
    ba = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ba.set_frame(k, scans[k])
    counts, total = ba.run(poses, ci, cj)
    st = batch.BatchStage(K, 2 * rng, total)
    ba.feed(st)
    Hg = st.new_hg()
    st.linearize(poses, Hg)
    Hb, g, cost = batch.unpack_hg(Hg.cpu().numpy(), K, 2 * rng)
    cp, nc, sc = ba.read()
    cci = np.repeat(ci, counts); ccj = np.repeat(cj, counts)
    Ho, go, co = po.batch_linearize(K, 2 * rng, poses, cci, ccj, cp, nc, sc)
    assert abs(cost - co) <= 1e-10 * abs(co)
    assert np.linalg.norm(g - go) <= 1e-10 * np.linalg.norm(go)
    assert np.linalg.norm(Hb - Ho) <= 1e-10 * np.linalg.norm(Ho)
    st.close(); ba.close()


def test_batch_feature_selection_gathers_on_device():
    """globalFeatureSelectionAdd_Batch (Estimator.cpp:4057-4116): at most batch_feature_res_num = 25 records per keyframe pair,
    drawn without repetition from all but the last record; the gather runs on the device and the compacted arrays feed K8."""
    from glio_amd import batch, synth
    K, pts, rng_s = 5, 900, 2
    win = synth.make_window(W=K, pts_per_scan=pts, seed=synth.SEED_BASE + 61, perturb=(0.03, 0.2, 0.0), scan_radius=14.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    poses = np.c_[win.init.trans, win.init.quat]
    ci, cj = batch.pair_list(K, rng_s)
    ba = batch.BatchAssociation(K, pts, len(ci) * pts)
    for k in range(K):
        sc = win.scans[k].copy(); sc[:, :3] -= tlb
        ba.set_frame(k, sc)
    counts, total = ba.run(poses, ci, cj)
    counts = counts.copy()
    assert total > 0 and counts.max() > 25
    before = [a.copy() for a in ba.read()]
    src = ba.select(25, np.random.default_rng(3))
    offs = np.concatenate([[0], np.cumsum(counts)])
    assert all(c == min(25, n) if n > 25 else c == n for c, n in zip(ba.pair_count, counts))
    after = ba.read()
    assert all(np.array_equal(a, b[src]) for a, b in zip(after, before))
    k0 = 0
    for p, n in enumerate(counts):                              # per pair: distinct indices of that pair, never its last record when drawn
        s = src[k0:k0 + ba.pair_count[p]] - offs[p]
        k0 += ba.pair_count[p]
        assert len(set(s.tolist())) == len(s) and (s >= 0).all() and (s < n).all()
        if n > 25:
            assert (s < n - 1).all()
    st = batch.BatchStage(K, 2 * rng_s, max(ba.total, 1))
    ba.feed(st)
    Hg = st.new_hg(); st.linearize(poses, Hg)
    assert float(Hg[-1].item()) > 0
    st.close(); ba.close()


def test_rounds_with_reassociation_follow_the_oracle(frames):
    """The measured form of optimizeBatch's outer loop: the end keyframes are re-searched in every DDpsr_threshold round at the
    current poses, the interior keeps its stored constraints (Estimator.cpp:3004-3076) -- device (RoundsAssociation + the
    trust-region solve) against the oracle doing the same with associate_pair + orc_batch2_solve."""
    from glio_amd import ctypes_types as T
    from oracle import pyoracle as po
    scans, poses0 = frames
    K, sr = len(scans), 2
    band = 2 * sr
    rng = np.random.default_rng(7)
    odo = poses0.copy(); odo[:, :3] += rng.normal(0, 0.01, (K, 3))
    st = batch.BatchStage(K, band, 400000)
    ra = batch.RoundsAssociation(st, scans, sr, 4096)
    ra.start(poses0)
    opts = T.batch_tr_opts(max_iterations=10)
    thresholds = batch.DDPSR_THRESHOLDS[:3]
    poses, hist = batch.solve_batch_rounds(st, poses0, odo, sr, [], None, reassociate=ra, opts=opts, thresholds=thresholds)
    assert ra.runs == 3 + 2 * len(thresholds)
    # ---- the oracle: interior pairs once at poses0, end pairs again at the start of every round
    ci, cj = batch.pair_list(K, sr)
    ends = (ci < sr) | (ci > K - 1 - sr)

    def assoc(pairs, at):
        out = {}
        for a, b in pairs:
            out[(a, b)] = po.associate_pair(scans[a], at[a], scans[b], at[b])[:3]
        return out

    stored = assoc([(a, b) for a, b, e in zip(ci, cj, ends) if not e], poses0)
    ref = poses0.copy()
    dq = batch.delta_q_pairs(odo, sr)
    for _ in thresholds:
        cur = dict(stored)
        cur.update(assoc([(a, b) for a, b, e in zip(ci, cj, ends) if e], ref))
        cis, cjs, cps, ncs, scs = [], [], [], [], []
        for a, b in zip(ci, cj):
            cp, nc, sc = cur[(a, b)]
            cis.append(np.full(len(sc), a, np.int32)); cjs.append(np.full(len(sc), b, np.int32)); cps.append(cp); ncs.append(nc); scs.append(sc)
        P = po.BatchProblem(K, band, np.concatenate(cis), np.concatenate(cjs), np.concatenate(cps), np.concatenate(ncs), np.concatenate(scs), dq=dq)
        ref, summ = P.solve(ref, opts)
    assert np.abs(poses - ref).max() < 1e-7
    assert hist[-1]["final_cost"] <= hist[0]["initial_cost"]
    ra.close(); st.close()


def test_keyframe_by_keyframe_accumulation_equals_one_run(frames):
    """batchFeatureAssociation (Estimator.cpp:3413-3432) is called once per keyframe and ADDS the 2 search_range pairs of keyframe size - search_range - 1 to
    gl_vec_surf_*: the records accumulated call after call (sliding.KeyframeBatchAssociation: asynchronous runs appended behind the resident set, the
    frame taken from a sliding-window context's resident scan) are, bit for bit, the records of ONE glio_bassoc_run over the same pairs and poses."""
    from glio_amd import capi, sliding
    scans, poses = frames
    K, sr = len(scans), 2
    ba = batch.BatchAssociation(K, 4096, 400000)
    kba = sliding.KeyframeBatchAssociation(ba, search_range=sr)
    o = synth.default_opts(2, pts=4096, map_pts=64)
    ctx = capi.Context(o)
    zero = np.zeros(3, np.float32)
    for size in range(1, K + 1):
        k = size - 1
        if k % 2:                                  # every other keyframe comes from a context's resident scan (ring slot), the rest from the host
            ctx.slide_window(); ctx.set_scan(1, scans[k])
            ba.set_frame_from_scan(k, ctx, 1, zero)
        else:
            ba.set_frame(k, scans[k])
        want = sliding.KeyframeBatchAssociation.pairs_of(size, sr)
        cnt = kba.step(size, poses)
        assert (want is None) == (len(cnt) == 0)
    assert len(kba.pair_ci) == 2 * sr * (K - 2 * sr) and ba.total == sum(kba.counts) > 1000
    got = [a.copy() for a in ba.read(0, ba.total)]
    ref = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ref.set_frame(k, scans[k])
    counts, total = ref.run(poses, np.array(kba.pair_ci, np.int32), np.array(kba.pair_cj, np.int32))
    assert total == ba.total and counts.tolist() == kba.counts
    for a, b in zip(got, ref.read()):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    # globalFeatureSelectionAdd_Batch on the tail only: earlier keyframes' records stay, each new pair keeps res_num draws (never its last record)
    ba.reset()
    ksel = sliding.KeyframeBatchAssociation(ba, search_range=sr, feature_res_num=25, rng=np.random.default_rng(3))
    for size in range(2 * sr, K + 1):
        ksel.step(size, poses)
    assert all(c == 25 for c in ksel.counts) and ba.total == 25 * len(ksel.counts)
    cp, nc, sc = ba.read(0, ba.total)
    full_cp = got[0]
    fs = {r.tobytes() for r in full_cp}
    assert all(r.tobytes() in fs for r in cp)
    ctx.close(); ba.close(); ref.close()


def test_prepared_run_equals_plain_run(frames):
    """glio_bassoc_prepare_async (descriptors sent, tables cleared before the poses exist) followed by a run over the same pairs; by a run over OTHER pairs
    (the preparation does not apply: the run clears for itself); a preparation that is never used; a cloud replaced between preparation and run (same size:
    still applies; other size: does not) -- the records always equal those of a fresh object's plain run, bit for bit."""
    scans, poses = frames
    K = len(scans)
    pa = (np.array([2, 2, 3, 3], np.int32), np.array([0, 1, 1, 4], np.int32))
    pb = (np.array([1, 4, 4], np.int32), np.array([3, 2, 5], np.int32))

    def plain(pairs, sc):
        ref = batch.BatchAssociation(K, 4096, 400000)
        for k in range(K):
            ref.set_frame(k, sc[k])
        counts, total = ref.run(poses, *pairs)
        out = [a.copy() for a in ref.read()]
        return counts.tolist(), total, out

    ba = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ba.set_frame(k, scans[k])
    alt = [s.copy() for s in scans]
    alt[1] = scans[1][::-1].copy()                         # the same points in another order: same size
    short = [s.copy() for s in scans]
    short[1] = scans[1][: len(scans[1]) // 2].copy()      # another size
    cases = [(pa, pa, scans, None), (pa, pb, scans, None), (pb, pa, scans, None), (pa, pa, alt, 1), (pa, pa, short, 1)]
    for prep, run, sc, replaced in cases:
        for k in range(K):
            ba.set_frame(k, scans[k])
        ba.prepare(*prep)
        if replaced is not None:
            ba.set_frame(replaced, sc[replaced])
        counts, total = ba.run(poses, *run)
        want = plain(run, sc)
        assert counts.tolist() == want[0] and total == want[1] and total > 100
        for a, b in zip(ba.read(), want[2]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    ba.prepare(*pa)                                        # prepared, never run: nothing to clean up


def test_an_asynchronous_runs_counts_survive_side_calls(frames):
    """Advisor finding of round 5: every side entry point (set_frame*, prepare, select_range, read ...) used to CONSUME a pending asynchronous run -- its
    per-pair counts were dropped and a later glio_bassoc_finish returned zeros; its overflow surfaced from the unrelated call.  Now side calls only wait
    for the stream: the counts and the overflow of a run go to glio_bassoc_finish and to nobody else."""
    from glio_amd import capi
    scans, poses = frames
    K = len(scans)
    pairs = (np.array([2, 2, 3], np.int32), np.array([0, 1, 4], np.int32))
    ref = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ref.set_frame(k, scans[k])
    want, want_total = ref.run(poses, *pairs)
    ba = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ba.set_frame(k, scans[k])
    ba.reset()
    assert ba.run_append(poses, *pairs, wait=False) is None
    ba.set_frame(7, scans[7])                              # side calls between the run and its collection: a frame no pair uses ...
    ba.prepare(np.array([5], np.int32), np.array([6], np.int32))      # ... a preparation ...
    ba.read(0, 10)                                         # ... and a read-back
    cnt, total = ba.finish()
    assert cnt.tolist() == want.tolist() and total == want_total and total > 100
    cnt2, total2 = ba.finish()                             # collected once: a second finish reports the total and no counts
    assert total2 == want_total
    # overflow: an object too small for the pairs; the error belongs to finish(), not to the side call in between
    small = batch.BatchAssociation(K, 4096, 50)
    for k in range(K):
        small.set_frame(k, scans[k])
    small.reset()
    small.run_append(poses, *pairs, wait=False)
    small.set_frame(7, scans[7])                           # must not raise
    with pytest.raises(capi.GlioError):
        small.finish()
    small.finish()                                         # reported once
    # ... and an uncollected overflow is not lost when the next run replaces the counts
    small.reset()
    small.run_append(poses, *pairs, wait=False)
    with pytest.raises(capi.GlioError):
        small.run_append(poses, *pairs, wait=False)
    ba.close(); ref.close(); small.close()


def test_on_stream_selection_equals_the_host_rule(frames):
    """glio_bassoc_select_tail_draws_async: globalFeatureSelectionAdd_Batch on the association's stream, behind the searches of an asynchronous run, from raw
    64-bit draws the caller made before the counts existed.  The records it leaves equal what the host rule (glio::batchSelectionDraws /
    batch.batch_selection_draws with rand_below(n) = raw mod n, applied to the read-back records) leaves, bit for bit: pairs with at most res_num records keep
    everything, the others the first res_num of a shuffle that never draws the last record; earlier records of the object stay; the total is updated."""
    scans, poses = frames
    K = len(scans)
    pairs1 = (np.array([2, 2], np.int32), np.array([0, 1], np.int32))
    pairs2 = (np.array([3, 3, 4, 5], np.int32), np.array([1, 4, 2, 7], np.int32))
    res = 25
    rng = np.random.default_rng(77)
    raws = rng.integers(0, 2 ** 62, len(pairs2[0]) * res, dtype=np.uint64)
    raws[res:2 * res] = 0                                   # a pair whose every draw is "position i itself"
    # reference: plain runs, read back, the rule in numpy
    ref = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ref.set_frame(k, scans[k])
    ref.reset()
    c1, t1 = ref.run_append(poses, *pairs1)
    c2, t2 = ref.run_append(poses, *pairs2)
    cp, nc, sc = [a.copy() for a in ref.read(0, t2)]
    keep = list(range(t1))
    off = t1
    want_kept = []
    for p, cnt in enumerate(c2.tolist()):
        if cnt <= res:
            sel = list(range(cnt))
        else:
            pos = {}
            sel = []
            for i in range(min(res, cnt - 1)):
                j = i + int(raws[p * res + i]) % (cnt - 1 - i)
                vi, vj = pos.get(i, i), pos.get(j, j)
                pos[j] = vi
                sel.append(vj)
        keep += [off + v for v in sel]; want_kept.append(len(sel)); off += cnt
    keep = np.array(keep, np.int64)
    assert min(c2) > res + 1                                # the rule has something to cut
    # the object under test: the second run asynchronous, the selection behind it on the stream
    ba = batch.BatchAssociation(K, 4096, 400000)
    for k in range(K):
        ba.set_frame(k, scans[k])
    ba.reset()
    ba.run_append(poses, *pairs1)
    assert ba.run_append(poses, *pairs2, wait=False) is None
    ba.select_tail_draws(res, raws)
    found, total = ba.finish()
    assert found.tolist() == c2.tolist() and total == t1 + sum(want_kept) == len(keep)
    got = ba.read(0, total)
    for a, b in zip(got, (cp[keep], nc[keep], sc[keep])):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    # a pair that holds fewer than res_num records keeps them all: res_num above every count
    ba.reset()
    ba.run_append(poses, *pairs2, wait=False)
    ba.select_tail_draws(64, rng.integers(0, 2 ** 62, len(pairs2[0]) * 64, dtype=np.uint64))
    found2, total2 = ba.finish()
    assert found2.tolist() == c2.tolist() and total2 == sum(min(64, c - 1) if c > 64 else c for c in c2.tolist())
    # the keyframe driver with device draws books res_num per pair and holds exactly that
    from glio_amd import sliding
    ba.reset()
    kd = sliding.KeyframeBatchAssociation(ba, search_range=2, feature_res_num=res, rng=np.random.default_rng(5), device_draws=True)
    for size in range(4, K + 1):
        kd.step(size, poses)
    assert all(c == res for c in kd.counts) and ba.total == res * len(kd.counts) > 0
    ba.close(); ref.close()


def test_tables_in_the_frames_own_frames_give_the_same_records(frames):
    """Local mode (runs with few pairs per search frame: a keyframe call's 2 search_range pairs): a keyframe's voxel hash is built once from its LOCAL cloud, a run
    re-poses the table's points and groups each pair's queries by their cell in the search frame's own frame.  Against the global mode (every run hashes its
    search frames at their poses), bit for bit: the same pairs at three sets of poses one after the other (the tables survive, the points are re-posed), a cloud
    replaced in between (its table is rebuilt), a search frame with an empty cloud, a quaternion that is not of unit length (the run takes the global mode by itself)."""
    from glio_amd import capi
    scans, poses = frames
    K = len(scans)
    lib = capi.load()
    ci = np.array([5, 5, 5, 5, 6, 6, 6], np.int32); cj = np.array([1, 2, 3, 4, 2, 3, 7], np.int32)      # 7 pairs over 6 search frames: local where allowed
    rng = np.random.default_rng(11)

    def jitter(p, s):
        q = p.copy()
        q[:, :3] += rng.normal(0, s, (K, 3))
        dq = np.c_[np.ones(K), rng.normal(0, s * 0.2, (K, 3))]
        w0, v0 = q[:, 3:4].copy(), q[:, 4:7].copy()
        w1, v1 = dq[:, 0:1], dq[:, 1:4]
        qq = np.c_[w0 * w1 - (v0 * v1).sum(1, keepdims=True), w0 * v1 + w1 * v0 + np.cross(v0, v1)]
        q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        return q

    pose_sets = [poses, jitter(poses, 0.05), jitter(poses, 0.3)]
    scaled = poses.copy(); scaled[2, 3:7] *= 1.0005                           # not a unit quaternion: the conjugate does not invert it
    alt = scans[3][::-1].copy()
    out = {}
    try:
        for mode in (0, 1):
            lib.glio_debug_set_bassoc_local(mode)
            ba = batch.BatchAssociation(K, 4096, 400000)
            for k in range(K):
                ba.set_frame(k, scans[k])
            got = []
            for P in pose_sets:
                counts, total = ba.run(P, ci, cj)
                got.append((counts.tolist(), [a.copy() for a in ba.read()]))
            ba.set_frame(3, alt)                                               # another cloud (same points, another order) in a search frame
            counts, total = ba.run(pose_sets[1], ci, cj); got.append((counts.tolist(), [a.copy() for a in ba.read()]))
            ba.set_frame(2, scans[2][:0])                                      # an empty search frame
            counts, total = ba.run(pose_sets[1], ci, cj); got.append((counts.tolist(), [a.copy() for a in ba.read()]))
            ba.set_frame(2, scans[2])
            counts, total = ba.run(scaled, ci, cj); got.append((counts.tolist(), [a.copy() for a in ba.read()]))
            counts, total = ba.run(poses, ci, cj); got.append((counts.tolist(), [a.copy() for a in ba.read()]))
            out[mode] = got
            ba.close()
    finally:
        lib.glio_debug_set_bassoc_local(1)
    assert sum(out[0][0][0]) > 500 and out[0][0][0] != out[0][2][0]            # the poses matter
    for step, (g0, g1) in enumerate(zip(out[0], out[1])):
        assert g0[0] == g1[0], step
        for a, b in zip(g0[1], g1[1]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), step
