"""Frame-loop harness: this build's counterpart of the tracked-frame branch of
Tracking::TrackImage (reference modules/tracking/tracking.cc:72-112): DataAssociation (LK,
:304-307) -> CameraPoseEstimation (motion-model seed + pose-only solve, :309-319) ->
CameraPoseAndDeformationEstimation (:321-333) -> PointReuse (:394-506) -> KeyFrameInsertion
cadence (:336-392) -> SetLastFrame.  It only orchestrates: every numerical step is a call through a
*backend* -- `GpuBackend` = the C ABI of libnrs_hip.so through ctypes (one context for the main
tracker, one for the two-level reuse tracker); the tests plug the oracle in behind the same calls.

Keyframes extract new Shi-Tomasi features (tracking.cc:350-372): they enter the frame
as TRACKED observations without a map point and are followed by LK from then on.  Not reproduced (out of
the hot-path scope): map initialisation, the mapping thread (triangulation of those features,
UpdateTriangulatedPoints, BA on keyframes), visualisation.

Poses are Sophus::SE3f in the reference: unit quaternion + translation in float32, and so is the
motion-model algebra here (`se3f_*`)."""
import numpy as np

F32 = np.float32
TRACKED_WITH_3D, TRACKED, JUST_TRIANGULATED, BAD = 0, 1, 2, 3     # utilities/landmark_status.h:23-30


# ---- SE3f (xyzw quaternion, translation), float32 arithmetic ---------------------------------
def quat_mul_f(a, b):
    ax, ay, az, aw = [F32(v) for v in a]
    bx, by, bz, bw = [F32(v) for v in b]
    q = np.array([aw * bx + ax * bw + ay * bz - az * by,
                  aw * by - ax * bz + ay * bw + az * bx,
                  aw * bz + ax * by - ay * bx + az * bw,
                  aw * bw - ax * bx - ay * by - az * bz], F32)
    return (q / F32(np.sqrt(np.dot(q, q)))).astype(F32)


def quat_rot_f(q, v):
    x, y, z, w = [F32(c) for c in q]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], F32)
    return (R @ np.asarray(v, F32)).astype(F32)


def se3f_mul(a, b):
    return quat_mul_f(a[0], b[0]), (np.asarray(a[1], F32) + quat_rot_f(a[0], b[1])).astype(F32)


def se3f_act(a, X):
    """a * X for many points, float32, one fixed elementwise expression per coordinate."""
    x, y, z, w = [F32(c) for c in a[0]]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], F32)
    X = np.asarray(X, F32).reshape(-1, 3)
    t = np.asarray(a[1], F32)
    return np.stack([R[r, 0] * X[:, 0] + R[r, 1] * X[:, 1] + R[r, 2] * X[:, 2] + t[r] for r in range(3)], 1).astype(F32)


def se3f_inv(a):
    qi = np.array([-a[0][0], -a[0][1], -a[0][2], a[0][3]], F32)
    return qi, (-quat_rot_f(qi, a[1])).astype(F32)


def project_f32(model, prm, p):
    """CameraModel::Project in float32 (calibration/pin_hole.cc:27-33, kannala_brandt_8.cc:34-51); the
    KB8 trigonometry as defined in include/nrs.h (double function rounded to float)."""
    prm = np.asarray(prm, F32)
    p = np.asarray(p, F32).reshape(-1, 3)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    if model == 0:
        return np.stack([prm[0] * x / z + prm[2], prm[1] * y / z + prm[3]], 1).astype(F32)
    r2 = x * x + y * y
    th = np.arctan2(np.sqrt(r2).astype(np.float64), z.astype(np.float64)).astype(F32)
    psi = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(F32)
    th2 = th * th
    th3 = th * th2
    th5 = th3 * th2
    th7 = th5 * th2
    th9 = th7 * th2
    r = th + prm[4] * th3 + prm[5] * th5 + prm[6] * th7 + prm[7] * th9
    return np.stack([prm[0] * r * np.cos(psi.astype(np.float64)).astype(F32) + prm[2],
                     prm[1] * r * np.sin(psi.astype(np.float64)).astype(F32) + prm[3]], 1).astype(F32)


class GpuBackend:
    """The product path: two nrs contexts (each owns one LucasKanadeTracker state)."""

    def __init__(self, nrs, model, prm, klt_opts, dense_graph=False, cap_per_point=64, direct_solve=0, n_nodes=0):
        """n_nodes > 0: EMBEDDED-DEFORMATION mode (include/nrs.h nrs_track_deform_solve_embedded; needs the dense graph): n_nodes map points
        (farthest-point sampling on the initial map, nrs_skin_select_nodes) carry the deformation vertices for the whole sequence, every
        other tracked point is skinned to <= 11 of them in the pose-and-deformation solve of each frame (tracking.cc:321-333's call)"""
        self.nrs = nrs
        self.dense, self.cap, self.rg = dense_graph or n_nodes > 0, cap_per_point, None
        self.n_nodes, self.node_flag = n_nodes, None
        self.cam = nrs.make_camera(model, prm)
        self.ctx = nrs.Context(direct_solve=direct_solve)      # (nrs_options.direct_solve: the linear solver of the pose-and-deformation solve)
        self.ctx_reuse = nrs.Context()
        self.klt_opts = klt_opts
        self.ctx.klt_configure(klt_opts["win"], klt_opts["max_level"], klt_opts["max_iters"], klt_opts["epsilon"], klt_opts["min_eig"])

    # main tracker
    def klt_set_reference(self, im, pts):
        self.ctx.klt_set_reference(im, pts)

    def klt_track(self, im, pts, status, min_ssim):
        xy, st, good, _ = self.ctx.klt_track(im, pts, status, initial_flow=True, min_ssim=min_ssim)
        return xy, st

    def klt_get_templates(self, n):
        return self.ctx.klt_get_templates(0, n)

    def klt_insert_template(self, t):
        self.ctx.klt_insert_template(t)

    def klt_insert_templates(self, ts):                  # (one call, one upload: PointReuse's new slots of a frame)
        self.ctx.klt_insert_templates(ts)

    # The map's photometric information stays on the device (include/nrs.h nrs_klt_archive_templates): archived by map point id at
    # keyframes, inserted from there when a point is reused.  (NRS_FRAME_LOOP_HOST_TEMPLATES=1: through the host, as before round 5.)
    def archive_templates(self, slots, mps):
        self.ctx.klt_archive_templates(slots, mps)

    def insert_archived(self, mps, xy):
        self.ctx.klt_insert_archived(self.ctx, mps, xy)

    def reuse_track_archived(self, im, pts, mps, min_ssim):
        o = self.klt_opts
        self.ctx_reuse.klt_clear()
        self.ctx_reuse.klt_configure(o["win"], 1, o["max_iters"], o["epsilon"], o["min_eig"])
        self.ctx_reuse.klt_insert_archived(self.ctx, mps, pts)
        xy, st, good, _ = self.ctx_reuse.klt_track(im, pts, np.zeros(len(pts), np.int32), initial_flow=True, min_ssim=min_ssim)
        return xy, st

    # the tracker PointReuse builds for its candidates (maxLevel 1, tracking.cc:422-424)
    def reuse_track(self, im, pts, templates, min_ssim):
        o = self.klt_opts
        self.ctx_reuse.klt_clear()
        self.ctx_reuse.klt_configure(o["win"], 1, o["max_iters"], o["epsilon"], o["min_eig"])
        self.ctx_reuse.klt_insert_templates([dict(t, xy=np.asarray(p, F32)) for p, t in zip(pts, templates)])
        xy, st, good, _ = self.ctx_reuse.klt_track(im, pts, np.zeros(len(pts), np.int32), initial_flow=True, min_ssim=min_ssim)
        return xy, st

    def extract_features(self, im, held_xy, mask=None):
        xy, ids, _ = self.ctx.shi_extract(im, held_xy, mask)
        return xy, ids

    def pose_only(self, uv, X, q, t):
        q2, t2, _ = self.ctx.pose_only_solve(self.cam, uv, X, q, t)
        return q2, t2

    # dense_graph: the map's RegularizationGraph at the reference's density -- every pair of initial map points connected
    # (modules/map/map.cc:148-166) -- resident on the device; otherwise the caller's flat graph
    def make_graph(self, graph, X0):
        if not self.dense:
            return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}
        ids = np.arange(len(X0), dtype=np.int32)
        self.rg = self.nrs.RGraph(self.ctx, len(X0), graph["sigma"], graph["stretch_th"])
        self.rg.add_edges(np.asarray(X0, F32), ids, ids)
        if self.n_nodes > 0:                                       # the node set of the sequence: fixed map points
            nodes = self.ctx.skin_select_nodes(np.asarray(X0, F32), min(self.n_nodes, len(X0)), np.ones(len(X0), bool))
            self.node_flag = np.zeros(len(X0), np.uint8)
            self.node_flag[nodes] = 1
        return self.rg

    def track_deform(self, graph, map_pos, f_map, f_status, f_uv, f_pos, q, t, scale):
        self.last_trace = self.nrs.Trace(1024)
        if self.node_flag is not None:
            r = self.ctx.track_deform_solve_embedded(self.cam, graph, map_pos, f_map, f_status, f_uv, f_pos, self.node_flag[np.asarray(f_map)], q, t, scale,
                                                     self.last_trace, max(self.cap, 128))
            r["graph"] = graph
            return r
        if self.dense:
            r = self.ctx.track_deform_solve_rg(self.cam, graph, map_pos, f_map, f_status, f_uv, f_pos, q, t, scale, self.last_trace, self.cap)
            r["graph"] = graph                                     # updated in place on the device
            return r
        return self.ctx.track_deform_solve(self.cam, graph, map_pos, f_map, f_status, f_uv, f_pos, q, t, scale, self.last_trace)

    def close(self):
        if self.rg is not None:
            self.rg.close()
        self.ctx.close()
        self.ctx_reuse.close()


class FrameLoop:
    """State of Tracking + the slice of Map / Frame it touches, on flat arrays."""

    def __init__(self, backend, project_f32, wh, scale, kp0, X0, graph, pose_q, pose_t, im0,
                 klt_min_ssim=0.7, images_to_insert_keyframe=5, extract_on_keyframes=True):
        self.b, self.project, self.wh, self.scale = backend, project_f32, wh, float(scale)
        n = len(kp0)
        # current frame: slot i observes map point map_index[i]
        self.kp = np.asarray(kp0, F32).copy()
        self.pos = np.asarray(X0, F32).copy()
        self.status = np.zeros(n, np.int32)
        self.map_index = np.arange(n, dtype=np.int32)
        # map
        self.map_pos = np.asarray(X0, F32).copy()                  # MapPoint::GetLastWorldPosition
        # the caller's flat graph, or (a backend that keeps one) the all-pairs graph of the map
        self.graph = backend.make_graph(graph, X0) if hasattr(backend, "make_graph") else \
            {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}
        self.pose = (np.asarray(pose_q, F32), np.asarray(pose_t, F32))
        self.last_pose = self.pose
        self.motion = (np.array([0, 0, 0, 1], F32), np.zeros(3, F32))
        self.min_ssim, self.kf_every, self.since_kf = klt_min_ssim, images_to_insert_keyframe, 0
        self.extract = extract_on_keyframes
        # initial keyframe: klt reference + photometric information of every map point (tracking.cc:201-209)
        self.b.klt_set_reference(im0, self.kp)
        import os
        self.dev_templates = hasattr(self.b, "archive_templates") and not os.environ.get("NRS_FRAME_LOOP_HOST_TEMPLATES")
        if self.dev_templates:
            self.templates = None
            self.b.archive_templates(np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32))
        else:
            self.templates = self.b.klt_get_templates(n)
        self.log = []

    # ---- tracking.cc:72-112 (tracked branch)
    def track_image(self, im):
        lost = self.track_camera_and_deformation(im)
        reused = self.point_reuse(im, lost)
        n3d = int((self.status == TRACKED_WITH_3D).sum())
        kf = False
        if n3d >= 10:
            kf = self.keyframe_insertion(im)
            self.last_pose = self.pose
        self.log.append(dict(pose_q=self.pose[0].copy(), pose_t=self.pose[1].copy(), lost=sorted(int(x) for x in lost),
                             reused=reused, n_tracked=n3d, keyframe=kf, n_2d=int((self.status == TRACKED).sum()),
                             kp_2d=self.kp[self.map_index < 0].copy(),
                             status_by_map=self._status_by_map(), pos_by_map=self._pos_by_map()))
        return n3d >= 10

    def _status_by_map(self):
        s = np.full(len(self.map_pos), -1, np.int32)
        m = self.map_index >= 0                                    # slots without a map point: extracted 2D features
        s[self.map_index[m]] = self.status[m]
        return s

    def _pos_by_map(self):
        p = np.zeros((len(self.map_pos), 3), F32)
        m = self.map_index >= 0
        p[self.map_index[m]] = self.pos[m]
        return p

    # ---- tracking.cc:291-333
    def track_camera_and_deformation(self, im):
        self.kp, self.status = self.b.klt_track(im, self.kp, self.status, self.min_ssim)
        self.pose = se3f_mul(self.motion, self.pose)               # motion-model seed
        m = self.status == TRACKED_WITH_3D
        q, t = self.b.pose_only(self.kp[m], self.pos[m], self.pose[0].astype(np.float64), self.pose[1].astype(np.float64))
        self.pose = (np.asarray(q, np.float64).astype(F32), np.asarray(t, np.float64).astype(F32))
        mm = self.map_index >= 0                                   # the optimisation walks Frame::IndexToMapPointId (OPT:212-236)
        r = self.b.track_deform(self.graph, self.map_pos, self.map_index[mm], self.status[mm], self.kp[mm], self.pos[mm],
                                self.pose[0].astype(np.float64), self.pose[1].astype(np.float64), self.scale)
        self.pose = (np.asarray(r["pose_q"], np.float64).astype(F32), np.asarray(r["pose_t"], np.float64).astype(F32))
        self.pos[mm], self.status[mm] = np.asarray(r["f_pos"], F32), np.asarray(r["f_status"], np.int32)
        self.map_pos, self.graph = np.asarray(r["map_pos"], F32), r["graph"]
        self.motion = se3f_mul(self.pose, se3f_inv(self.last_pose))
        return set(int(x) for x in r["lost"])

    # ---- tracking.cc:394-506
    def point_reuse(self, im, lost):
        w, h = self.wh
        in_frame = np.full(len(self.map_pos), -1, np.int64)
        has_mp = self.map_index >= 0
        in_frame[self.map_index[has_mp]] = np.nonzero(has_mp)[0]
        pc = se3f_act(self.pose, self.map_pos)
        uv = self.project(pc) if len(pc) else np.zeros((0, 2), F32)
        inside = (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
        # candidates: the points lost in this frame and the map points the frame does not hold (tracking.cc:404-420), in front of the camera and
        # inside the image -- one pass over the map, ascending map index (the reference walks a std::set)
        present = np.zeros(len(self.map_pos), bool)
        held = np.nonzero(in_frame >= 0)[0]
        st_held = self.status[in_frame[held]]
        present[held] = (st_held == TRACKED_WITH_3D) | (st_held == JUST_TRIANGULATED)
        is_cand = np.zeros(len(self.map_pos), bool)
        if lost:
            is_cand[np.fromiter((int(x) for x in lost), np.int64, len(lost))] = True
        if len(pc):
            is_cand |= ~present & (pc[:, 2] >= 0) & inside
            is_cand &= inside & ~np.isnan(uv).any(1)
        cand = np.nonzero(is_cand)[0]
        if not len(cand):
            return 0
        seeds = uv[cand].astype(F32)
        if self.dev_templates:
            xy, st = self.b.reuse_track_archived(im, seeds, cand.astype(np.int32), 0.75)
        else:
            xy, st = self.b.reuse_track(im, seeds, [self.templates[int(mp)] for mp in cand], 0.75)
        # accepted: tracked, and within sqrt(5.99) px of the projection (fp32 arithmetic, element by element as the scalar form)
        xy = np.asarray(xy)
        ex = uv[cand, 0].astype(F32) - xy[:, 0].astype(F32)
        ey = uv[cand, 1].astype(F32) - xy[:, 1].astype(F32)
        ok = (np.asarray(st) == TRACKED_WITH_3D) & ~(ex * ex + ey * ey > F32(5.99))
        slot = in_frame[cand]
        upd = ok & (slot >= 0)                          # the frame holds a slot for the point: it is refreshed
        if upd.any():
            self.kp[slot[upd]] = xy[upd]
            self.pos[slot[upd]] = self.map_pos[cand[upd]]
            self.status[slot[upd]] = TRACKED_WITH_3D
        new = ok & (slot < 0)                           # candidates that enter the frame as new slots, in candidate order (one append below)
        new_k, new_mp = [int(k) for k in np.nonzero(new)[0]], [int(mp) for mp in cand[new]]
        reused = int(ok.sum())
        if new_mp:                                      # (the slots and their photometric information, appended in the loop's order)
            nk, nm = np.asarray(new_k), np.asarray(new_mp)
            self.kp = np.vstack([self.kp, xy[nk]]).astype(F32)
            self.pos = np.vstack([self.pos, self.map_pos[nm]]).astype(F32)
            self.status = np.concatenate([self.status, np.full(len(nm), TRACKED_WITH_3D, np.int32)]).astype(np.int32)
            self.map_index = np.concatenate([self.map_index, nm]).astype(np.int32)
            if self.dev_templates:
                self.b.insert_archived(nm.astype(np.int32), xy[nk].astype(F32))
                return reused
            tpl = [dict(self.templates[mp], xy=xy[k].astype(F32)) for k, mp in zip(new_k, new_mp)]
            if hasattr(self.b, "klt_insert_templates"):
                self.b.klt_insert_templates(tpl)
            else:
                for t in tpl:
                    self.b.klt_insert_template(t)
        return reused

    # ---- tracking.cc:336-392
    def keyframe_insertion(self, im):
        if self.since_kf < self.kf_every:
            self.since_kf += 1
            return False
        self.since_kf = 0
        if self.extract:
            # ExtractFeaturesInFrame (tracking.cc:374-382): the extractor is told the keypoints the frame
            # holds (TRACKED_WITH_3D and TRACKED, in slot order); new corners become TRACKED observations
            held = (self.status == TRACKED_WITH_3D) | (self.status == TRACKED)
            xy, _ = self.b.extract_features(im, self.kp[held])
            k = len(xy)
            self.kp = np.vstack([self.kp, xy]).astype(F32)
            self.pos = np.vstack([self.pos, np.zeros((k, 3), F32)]).astype(F32)
            self.status = np.concatenate([self.status, np.full(k, TRACKED, np.int32)]).astype(np.int32)
            self.map_index = np.concatenate([self.map_index, np.full(k, -1, np.int32)]).astype(np.int32)
        # KeyFrame(frame) + Frame::SetFromKeyFrame (keyframe.cc:26-55, frame.cc:47-77): the slots with 3D, then
        # the TRACKED ones; everything else leaves the frame
        order = np.concatenate([np.nonzero(self.status == TRACKED_WITH_3D)[0], np.nonzero(self.status == TRACKED)[0]])
        self.kp, self.pos, self.status, self.map_index = self.kp[order], self.pos[order], self.status[order], self.map_index[order]
        self.pos[self.status == TRACKED] = 0
        self.map_index[self.status == TRACKED] = -1               # only the 3D slots keep their map point (frame.cc:56-62)
        self.b.klt_set_reference(im, self.kp)
        if self.dev_templates:                                     # Frame::MapPointIdToIndex: slots that have a map point
            slots = np.nonzero(self.map_index >= 0)[0].astype(np.int32)
            self.b.archive_templates(slots, self.map_index[slots].astype(np.int32))
            return True
        tpl = self.b.klt_get_templates(len(self.map_index))
        for i, mp in enumerate(self.map_index):
            if mp >= 0:
                self.templates[mp] = tpl[i]
        return True
