"""Oracle behind the frame-loop harness (test infrastructure only): the same backend calls as
nrs_frame_loop.GpuBackend, answered by the CPU restatements (nrs_oracle.py, lk_oracle.py).
LucasKanadeTracker::GetPhotometricInformationOfPoint / InsertPhotometricInformation
(reference lucas_kanade_tracker.cc:590-620) are restated here on the LK oracle's lists."""
import numpy as np

import lk_oracle as LK
import nrs_oracle as O
import shi_oracle as SH

F32 = np.float32


def _get_template(lk, idx):
    nl = lk.max_level + 1
    gray, grad = np.zeros((nl, 21, 21), np.int16), np.zeros((nl, 21, 21, 2), np.int16)
    mean, valid = np.zeros((nl, 2), F32), np.zeros(nl, np.uint8)
    for level in range(nl):
        mean[level] = (lk.meanI[level][idx], lk.meanI2[level][idx])
        if lk.Iref[level][idx] is not None:
            gray[level], grad[level], valid[level] = lk.Iref[level][idx], lk.Idref[level][idx], 1
    return dict(xy=lk.prev[idx].copy(), gray=gray, grad=grad, mean=mean, valid=valid)


def _insert_template(lk, t):
    nl = lk.max_level + 1
    if len(lk.meanI) == 0:
        lk.meanI = [np.zeros(0, F32) for _ in range(nl)]
        lk.meanI2 = [np.zeros(0, F32) for _ in range(nl)]
        lk.Iref = [[] for _ in range(nl)]
        lk.Idref = [[] for _ in range(nl)]
        lk.n_levels = nl
    lk.prev = np.vstack([lk.prev, np.asarray(t["xy"], F32)[None]]).astype(F32)
    for level in range(nl):
        lk.meanI[level] = np.append(lk.meanI[level], F32(t["mean"][level][0])).astype(F32)
        lk.meanI2[level] = np.append(lk.meanI2[level], F32(t["mean"][level][1])).astype(F32)
        ok = bool(t["valid"][level])
        lk.Iref[level] = list(lk.Iref[level]) + [np.asarray(t["gray"][level], np.int16).copy() if ok else None]
        lk.Idref[level] = list(lk.Idref[level]) + [np.asarray(t["grad"][level], np.int16).copy() if ok else None]


class OracleBackend:
    def __init__(self, model, prm, klt_opts, dense_graph=False, n_nodes=0):
        """n_nodes > 0: the embedded-deformation mode (embedded_oracle.track_deform_solve_embedded) on the dense graph, nodes chosen once on the
        initial map by skin_oracle.select_nodes -- the twin of nrs_frame_loop.GpuBackend(n_nodes=...)"""
        self.model, self.prm, self.o, self.dense = model, prm, klt_opts, dense_graph or n_nodes > 0
        self.n_nodes, self.node_flag = n_nodes, None
        self.lk = LK.LucasKanadeOracle(klt_opts["win"], klt_opts["max_level"], klt_opts["max_iters"], klt_opts["epsilon"], klt_opts["min_eig"])

    def klt_set_reference(self, im, pts):
        self.lk.set_reference(im, pts)

    def klt_track(self, im, pts, status, min_ssim):
        xy, st, good, _ = self.lk.track(im, np.asarray(pts, F32).copy(), status, initial_flow=True, min_ssim=min_ssim)
        return xy, st

    def klt_get_templates(self, n):
        return [_get_template(self.lk, i) for i in range(n)]

    def klt_insert_template(self, t):
        _insert_template(self.lk, t)

    def reuse_track(self, im, pts, templates, min_ssim):
        o = self.o
        lk = LK.LucasKanadeOracle(o["win"], 1, o["max_iters"], o["epsilon"], o["min_eig"])
        for p, t in zip(pts, templates):
            _insert_template(lk, dict(t, xy=np.asarray(p, F32)))
        xy, st, good, _ = lk.track(im, np.asarray(pts, F32).copy(), np.zeros(len(pts), np.int32), initial_flow=True, min_ssim=min_ssim)
        return xy, st

    def extract_features(self, im, held_xy, mask=None):
        if not hasattr(self, "shi"):
            self.shi = SH.ShiTomasi(5)
        return self.shi.extract(im, held_xy, mask)

    def pose_only(self, uv, X, q, t):
        q2, t2, _ = O.pose_only_solve(self.model, self.prm, uv, X, q, t)
        return q2, t2

    def make_graph(self, graph, X0):
        if not self.dense:
            return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}
        import rgraph_oracle as RG                                  # all pairs (modules/map/map.cc:148-166)
        g = RG.DenseGraph(len(X0), graph["sigma"], graph["stretch_th"])
        ids = np.arange(len(X0))
        g.add_edges(np.asarray(X0, F32), ids, ids)
        if self.n_nodes > 0:
            import skin_oracle as K
            self.node_flag = np.zeros(len(X0), np.uint8)
            self.node_flag[K.select_nodes(np.asarray(X0, F32), min(self.n_nodes, len(X0)))] = 1
        return g

    def track_deform(self, graph, map_pos, f_map, f_status, f_uv, f_pos, q, t, scale):
        if self.node_flag is not None:
            import embedded_oracle as E
            return E.track_deform_solve_embedded(self.model, self.prm, graph, map_pos, f_map, f_status, f_uv, f_pos, self.node_flag[np.asarray(f_map)], q, t, scale)
        return O.track_deform_solve(self.model, self.prm, graph, map_pos, f_map, f_status, f_uv, f_pos, q, t, scale)

    def close(self):
        pass
