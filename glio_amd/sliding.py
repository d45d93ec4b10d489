"""Streaming driver: the per-keyframe call sequence of `Estimator::optimizeSlidingWindowWithLandMark`
(reference GLIO/src/Estimator.cpp:2046-2736) over a stream of keyframes, written against a backend with the
methods of `capi.Context` (set_map / associate / set_imu / set_prior / set_gnss / solve / marginalize).

Per window: local map -> (:2056), state arrays (:2100-2127), prior (:2153-2158), IMU edges (:2182-2192),
correspondences per slot from the pre-solve poses (:2198-2248, once per solve: quirk Q3), solve (:2424-2433),
quaternion sign unification (:2439-2457), marginalization of slot 0 with the kept blocks renamed s -> s-1
(:2462-2607).  The window then slides by one keyframe; the newest slot starts from the caller's prediction
(the reference takes it from IMU propagation, outside this path).
"""
import numpy as np

from . import capi
from . import ctypes_types as T


def unify_quaternions(state):
    """Estimator.cpp:2439-2457: q and -q are the same rotation; the reference keeps w >= 0."""
    flip = state.quat[:, 0] < 0
    state.quat[flip] *= -1.0
    return state


def feature_selection_draws(count, feature_res_num, rng, random_select=True):
    """The index sequence `featureSelection` (Estimator.cpp:3894-3992) keeps, for `count` correspondences: nothing
    changes when count - 1 < feature_res_num (early return :3906-3909, quirk Q9); otherwise feature_res_num draws, each
    uniform over the records still left (the reference builds a no-repeat random array and takes its last element,
    :3948-3957, then erases the record, :3964-3978); with random_select == false the set is emptied (:3945,3981-3987).
    Returns None for "keep all", else the kept original indices in draw order."""
    if count < 1 or count - 1 < feature_res_num:
        return None
    if not random_select:
        return np.zeros(0, np.int32)
    # the d-th draw picks the k-th record STILL LEFT, k uniform below count - d; its original index is k advanced past every removed index <= it
    # (the same as popping from the list of remaining records, without building a list of `count` entries per slot; glio::featureSelectionDraws
    # in glio_backend.hpp is this loop in C++: same generator in, same indices out)
    gone, out = [], []
    for d in range(feature_res_num):
        v = int(rng.integers(0, int(count) - d))
        pos = 0
        while pos < len(gone) and gone[pos] <= v:
            v += 1; pos += 1
        gone.insert(pos, v)
        out.append(v)
    return np.asarray(out, np.int32)


class TableRng:
    """A generator that both hosts can share: integers(lo, hi) = lo + table[k++] mod (hi - lo) over a table of 63-bit numbers (written to a file for
    host_demo_stream's `draws=`).  Not a statistical generator -- the means by which a test makes the C++ and the Python host draw the same records."""

    def __init__(self, table):
        self.table, self.k = np.ascontiguousarray(table, np.uint64), 0

    def integers(self, lo, hi):
        v = int(self.table[self.k % len(self.table)]); self.k += 1
        lo, hi = int(lo), int(hi)
        return lo + v % (hi - lo)


def feature_selection(backend, slot, count, feature_res_num, rng, random_select=True):
    sel = feature_selection_draws(count, feature_res_num, rng, random_select)
    if sel is None:
        return count
    backend.select_correspondences(slot, sel)
    return len(sel)


def feature_selection_window(backend, counts, feature_res_num, rng, random_select=True):
    """featureSelection for every slot of the window, the draws in slot order (as W calls of feature_selection would make them), ONE device call.
    Returns the residual counts."""
    sels = [feature_selection_draws(int(c), feature_res_num, rng, random_select) for c in counts]
    if any(sel is not None for sel in sels):
        backend.select_correspondences_window(sels)
    return [int(c) if sel is None else len(sel) for c, sel in zip(counts, sels)]


MIN_MAP_POINTS = 50        # `if (surf_local_map_ds->points.size() > 50)` guards the correspondence search, Estimator.cpp:2221,2244


class SlidingWindowDriver:
    def __init__(self, backend, opts, lidar_pose=capi.lidar_pose):
        self.be, self.opts, self.W = backend, opts, opts.window
        self.lidar_pose = lidar_pose
        self.prior = None
        self.state = None
        self.first = 0                   # index (into the keyframe stream) of slot 0
        self.history = []

    def start(self, init_states):
        """init_states: WindowState of the first full window."""
        self.state = init_states.copy()
        self.state.n_ddt = 0
        self.first = 0

    def step(self, map_pts, scans, preints):
        """One optimizeSlidingWindowWithLandMark() call.  scans[s] / preints[s] are those of window slot s
        (preints[s] links slots s and s+1).  Returns (solved state, summary, correspondence counts)."""
        be, W = self.be, self.W
        be.set_map(map_pts)
        counts = []
        for s in range(W):
            if len(map_pts) <= MIN_MAP_POINTS:          # Estimator.cpp:2221: "Not enough feature points from the map" -- no LiDAR factors
                be.set_correspondences(s, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), np.zeros(0))
                counts.append(0)
                continue
            q2, t2 = self.lidar_pose(self.opts, self.state.quat[s], self.state.trans[s])
            counts.append(be.associate(s, scans[s], q2, t2))
        be.set_imu(preints)
        be.set_prior(self.prior)
        be.set_gnss(None, [], [])
        sol, summ = be.solve(self.state)
        unify_quaternions(sol)
        self.prior = be.marginalize(sol)
        self.state = sol
        self.history.append((self.first, sol.copy(), summ, counts))
        return sol, summ, counts

    def slide(self, new_trans, new_quat, new_speed_bias):
        """Drop slot 0, shift the rest down, append the prediction of the new keyframe."""
        st = self.state
        st.trans[:-1] = st.trans[1:].copy(); st.quat[:-1] = st.quat[1:].copy(); st.speed_bias[:-1] = st.speed_bias[1:].copy()
        st.trans[-1], st.quat[-1], st.speed_bias[-1] = new_trans, new_quat, new_speed_bias
        self.first += 1


class ResidentSlidingWindow:
    """The same per-keyframe sequence with everything kept on the device between keyframes (capi.Context only):
    scans slide with `glio_slide_window`, only the NEW keyframe's scan is uploaded, all slots are associated in one call,
    and the marginalization result stays resident as the next prior (`glio_marginalize_keep`)."""

    def __init__(self, ctx, opts, lidar_pose=capi.lidar_pose):
        self.ctx, self.opts, self.W = ctx, opts, opts.window
        self.lidar_pose = lidar_pose
        self.state = None
        self.first = 0
        self._have_scans = False

    def start(self, init_states):
        self.state = init_states.copy()
        self.state.n_ddt = 0
        self.ctx.set_prior(None)
        self.ctx.set_gnss(None, [], [])

    def step(self, map_pts, scans, preints):
        ctx, W = self.ctx, self.W
        ctx.set_map(map_pts)
        if not self._have_scans:
            for s in range(W):
                ctx.set_scan(s, scans[s])
            self._have_scans = True
        else:
            ctx.slide_window()
            ctx.set_scan(W - 1, scans[W - 1])
        poses = [self.lidar_pose(self.opts, self.state.quat[s], self.state.trans[s]) for s in range(W)]
        if len(map_pts) <= MIN_MAP_POINTS:              # Estimator.cpp:2221
            for s in range(W):
                ctx.set_correspondences(s, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), np.zeros(0))
            counts = [0] * W
        else:
            counts = ctx.associate_window(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
        ctx.set_imu(preints)
        sol, summ = ctx.solve(self.state)
        unify_quaternions(sol)
        ctx.marginalize_keep(sol)
        self.state = sol
        return sol, summ, [int(c) for c in counts]

    def slide(self, new_trans, new_quat, new_speed_bias):
        st = self.state
        st.trans[:-1] = st.trans[1:].copy(); st.quat[:-1] = st.quat[1:].copy(); st.speed_bias[:-1] = st.speed_bias[1:].copy()
        st.trans[-1], st.quat[-1], st.speed_bias[-1] = new_trans, new_quat, new_speed_bias
        self.first += 1


class KeyframeBatchAssociation:
    """batchFeatureAssociation() (Estimator.cpp:3413-3432), the call that ends every optimizeSlidingWindowWithLandMark (:2733): once the stream holds
    2 search_range keyframes, the keyframe idx = size - search_range - 1 is matched against its 2 search_range neighbours at the CURRENT poses
    (findGlobalCorrespondingSurfFeaturesAdd_Batch :3808-3892: 12 hash builds + 12 pair searches on the device) and -- with a generator --
    globalFeatureSelectionAdd_Batch (:4057-4116) keeps batch_feature_res_num records per pair.  The records accumulate in `ba` (a batch.BatchAssociation
    sized for the stream), pair major, exactly what BatchAssociation.run over the same pairs and poses gives; `pairs` / `counts` list them in call order."""

    def __init__(self, ba, search_range=6, feature_res_num=None, rng=None, device_draws=False):
        """device_draws: the selection's raw draws (res_num uint64 per pair, from `rng`) travel with the enqueue and the selection runs on the association's stream
        behind the searches (glio_bassoc_select_tail_draws_async) instead of after a host round trip at finish()."""
        self.ba, self.sr, self.res_num, self.rng = ba, search_range, feature_res_num, rng
        self.device_draws = bool(device_draws) and feature_res_num is not None and rng is not None and 1 <= feature_res_num <= 64
        self._on_stream = False
        self.pair_ci, self.pair_cj, self.counts = [], [], []
        self._enq = None

    @staticmethod
    def pairs_of(size, search_range):
        """(idx, [j ...]) of the call made when the stream holds `size` keyframes, or None (Estimator.cpp:3414-3416)."""
        idx = size - search_range - 1
        if size < 2 * search_range or idx < search_range:
            return None
        return idx, [j for j in range(idx - search_range, idx + search_range + 1) if j != idx]

    def prepare(self, size):
        """Optional, before the poses exist (before the solve of the same keyframe call): the pairs follow from the keyframe count alone; their search frames'
        build descriptors go to the device and their hash tables are cleared now (glio_bassoc_prepare_async), enqueue() then starts with the clouds' transform."""
        pr = self.pairs_of(size, self.sr)
        if pr is not None:
            idx, js = pr
            self.ba.prepare(np.full(len(js), idx, np.int32), np.asarray(js, np.int32))

    def enqueue(self, size, poses):
        """poses [K][7] = t, q of every keyframe slot of `ba`.  Enqueues the searches on the association's own stream and returns."""
        pr = self.pairs_of(size, self.sr)
        if pr is None:
            self._enq = None
            return 0
        idx, js = pr
        ci, cj = np.full(len(js), idx, np.int32), np.asarray(js, np.int32)
        self._first = self.ba.total
        self.ba.run_append(poses, ci, cj, wait=False)
        self._enq = (ci, cj)
        self._on_stream = False
        if self.device_draws:
            raws = np.array([int(self.rng.integers(0, 2 ** 62)) for _ in range(len(js) * self.res_num)], np.uint64)
            self.ba.select_tail_draws(self.res_num, raws)
            self._on_stream = True
        return len(js)

    def finish(self):
        """Waits for the enqueued call, applies the selection, books the pairs (`counts`: what was kept).  Returns the per-pair counts FOUND."""
        if self._enq is None:
            return np.zeros(0, np.int64)
        ci, cj = self._enq
        self._enq = None
        cnt, total = self.ba.finish()
        found = cnt.copy()
        if self._on_stream:            # the device applied the rule: everything of a pair with at most res_num records, else res_num (never the last record)
            cnt = np.array([c if c <= self.res_num else min(self.res_num, c - 1) for c in found.tolist()], np.int64)
        elif self.res_num is not None and self.rng is not None:
            from .batch import batch_selection_draws
            offs = self._first + np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            keep, kept = [], cnt.copy()
            for p in range(len(cnt)):
                d = batch_selection_draws(int(cnt[p]), self.res_num, self.rng)
                sel = np.arange(int(cnt[p]), dtype=np.int64) if d is None else d
                keep.append(offs[p] + sel); kept[p] = len(sel)
            src = np.concatenate(keep) if keep else np.zeros(0, np.int64)
            if len(src) != total - self._first:
                self.ba.select_range(self._first, src, total)
            cnt = kept
        self.pair_ci += ci.tolist(); self.pair_cj += cj.tolist(); self.counts += [int(c) for c in cnt]
        return found

    def step(self, size, poses):
        self.enqueue(size, poses)
        return self.finish()
