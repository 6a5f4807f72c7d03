"""Oracle of the EMBEDDED-DEFORMATION mode of the pose-and-deformation solve (TEST INFRASTRUCTURE ONLY).

SURVEY.md 8(d), C2: "M farthest-point-sampled nodes; other points interpolate from <= 11 nodes with normalised weights".  The
reference has no such estimator: its deformation graph has one vertex per map point, and the only skinning it contains is the
second stage of CameraPoseAndDeformationOptimization, where a point that is not optimised follows <= 11 optimised graph
neighbours (modules/optimization/g2o_optimization.cc:476-553, spatial_regularizer_fixed.cc:32-43).  This file generalises the
reference function (as restated in nrs_oracle.track_deform_solve, OPT:148-557) in the one way the wording allows:

  * NODES are the optimised points that carry a free deformation delta_m (LandmarkVertex, 3 dof); the regularisers of
    OPT:255-335 are built between nodes only (a point's GetEdges walk accepts nodes; other optimised points are passed over);
  * every other optimised point i is SKINNED: delta_i = sum_k omega_ik delta_{n_ik} over the nodes its own GetEdges walk accepts
    (the same walk: stop after more than 10 accepted or at the first BAD connection), omega_ik = w_ik / sum_k w_ik with the
    connection weights w (float32, summed and divided in float64); its ReprojectionErrorWithDeformation edge
    (reprojection_error_with_deformation.cc:37-68) keeps its residual, information and Huber kernel, and its Jacobian with
    respect to node k is omega_ik times the reference's 2 x 3 block -- the observations of the skinned points constrain the
    nodes and the pose;
  * rounds, inlier levels, the IQR rejection, status / position write-back and the graph update treat all optimised points
    alike (a skinned point has no vertex to fix); stage 2 lets the lost points follow their optimised neighbours, nodes and
    skinned points (the latter as constants).

With every optimised point a node this IS nrs_oracle.track_deform_solve, statement for statement: tests/test_oracle_embedded_cpu.py
holds the two equal to the last bit, which is the only pin this mode can have ("parity unpinned" beyond M = N)."""
import numpy as np

import nrs_oracle as O
from nrs_oracle import F32

MAX_NODES = 11


class SkinnedReprojEdges(O.EdgeGroup):
    """ReprojectionErrorWithDeformation whose point is X0 + sum_k omega_k pts[node_k] (<= 11 nodes, omega = 0 pads)."""
    dim = 2

    def __init__(self, uv, X0, nodes, omega, info, delta):
        n = len(uv)
        super().__init__(n, info, delta)
        self.uv = np.asarray(uv, np.float64).reshape(n, 2)
        self.X0 = np.asarray(X0, np.float64).reshape(n, 3)
        self.nodes = np.asarray(nodes, np.int64).reshape(n, MAX_NODES)
        self.omega = np.asarray(omega, np.float64).reshape(n, MAX_NODES)
        self.slots = [('pose', np.zeros(n, np.int64))] + [('pt', self.nodes[:, k].copy()) for k in range(MAX_NODES)]

    def deformation(self, G, idx):
        d = np.zeros((len(idx), 3))
        for k in range(MAX_NODES):                                 # (sequential over the nodes)
            d += self.omega[idx, k, None] * G.pts[self.nodes[idx, k]]
        return d

    def _cam(self, G, idx):
        return O._reproj_core(G, G.pose_q[0], G.pose_t[0], self.X0[idx] + self.deformation(G, idx))

    def residual(self, G, idx):
        p = self._cam(G, idx)
        return self.uv[idx] - O.project_f32(G.cam_model, G.cam_prm, p.astype(F32)).astype(np.float64)

    def jacobians(self, G, idx):
        p = self._cam(G, idx)
        Jp = -O.projection_jacobian_f32(G.cam_model, G.cam_prm, p.astype(F32)).astype(np.float64)
        Jl = np.einsum('nij,jk->nik', Jp, O.quat_to_R(G.pose_q[0]))
        return [np.einsum('nij,njk->nik', Jp, O._expmap_jac(p))] + [self.omega[idx, k, None, None] * Jl for k in range(MAX_NODES)]


def track_deform_solve_embedded(cam_model, cam_prm, graph, map_pos, f_map, f_status, f_uv, f_pos, f_node, pose_q, pose_t, scale,
                                trace=None, solver=O.solve_spd):
    """f_node[i] != 0: frame landmark i is a node.  Everything else as nrs_oracle.track_deform_solve (graph: a flat dict, copied,
    or a rgraph_oracle.DenseGraph, updated in place)."""
    if isinstance(graph, dict):
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}

        def get_edges_of(p):
            return [(o, g["e_w"][e], g["e_d0"][e], g["e_status"][e]) for o, e in O.graph_get_edges(g, p)]

        def update_vertex_of(p, pos):
            return O.graph_update_vertex_flat(g, p, pos)
    else:
        g = graph

        def get_edges_of(p):
            js, w, d0, st = g.get_edges(p)
            return list(zip(js.tolist(), w, d0, st.tolist()))

        def update_vertex_of(p, pos):
            return g.update_vertex(pos, p)
    map_pos = np.array(map_pos, F32)
    f_status = np.array(f_status, np.int32)
    f_pos = np.array(f_pos, F32)
    f_map = np.asarray(f_map, np.int64)
    f_uv = np.asarray(f_uv, F32)
    f_node = np.asarray(f_node).astype(bool)
    n_map = len(map_pos)
    map_to_frame = -np.ones(n_map, np.int64)
    map_to_frame[f_map[f_map >= 0]] = np.where(f_map >= 0)[0]
    opt_f = np.where((f_status == O.TRACKED_WITH_3D) & (f_map >= 0))[0]
    N = len(opt_f)
    ids = f_map[opt_f]
    id_to_idx = -np.ones(n_map, np.int64)
    id_to_idx[ids] = np.arange(N)
    is_node = f_node[opt_f]
    node_of = -np.ones(N, np.int64)                                 # optimised point -> vertex (node) index
    node_of[is_node] = np.arange(int(is_node.sum()))
    M = int(is_node.sum())
    node_idx = np.where(is_node)[0]                                # vertex -> optimised point
    X0 = f_pos[opt_f].astype(np.float64)
    q0 = O.quat_normalize(np.asarray(pose_q, np.float64))
    t0 = np.asarray(pose_t, np.float64).copy()
    info_sp = O.info_spatial(scale)

    # ---- edge construction: OPT:224-337 between nodes, the same walk binds a skinned point to its nodes
    reg = [dict() for _ in range(N)]
    dm_i, dm_j, dm_w, sp_d0 = [], [], [], []
    sk_nodes = np.zeros((N, MAX_NODES), np.int64)
    sk_omega = np.zeros((N, MAX_NODES))
    sk_cnt = np.zeros(N, np.int64)
    lost = set()
    for idx in range(N):
        n_reg = 0
        wsum = []
        for other, e_w, e_d0, e_st in get_edges_of(int(ids[idx])):
            if n_reg > O.REGULARIZERS_PER_POINT or e_st == O.GRAPH_BAD:
                break
            fo = map_to_frame[other]
            if fo < 0 or f_status[fo] != O.TRACKED_WITH_3D:
                if fo >= 0 and f_status[fo] != O.JUST_TRIANGULATED:
                    lost.add(other)
                continue
            io = int(id_to_idx[other])
            if not is_node[io]:
                continue                                            # an optimised point without a vertex: passed over
            if is_node[idx]:
                if io in reg[idx]:
                    continue
                k = len(dm_i)
                dm_i.append(int(node_of[idx])); dm_j.append(int(node_of[io])); dm_w.append(e_w); sp_d0.append(e_d0)
                reg[idx][io] = k
                reg[io][idx] = k
            else:
                sk_nodes[idx, n_reg] = node_of[io]
                wsum.append(float(np.float64(F32(e_w))))
            n_reg += 1
        if not is_node[idx] and wsum:
            sk_cnt[idx] = len(wsum)
            tot = 0.0
            for w in wsum:                                         # (sequential: the summation order is part of the statement)
                tot += w
            sk_omega[idx, :len(wsum)] = np.asarray(wsum) / tot
    E = len(dm_i)
    sk_idx = np.where(~is_node & (sk_cnt > 0))[0]                   # skinned points (an optimised non-node that met no node stays put)
    S = len(sk_idx)
    G = O.Graph(cam_model, cam_prm, [q0], [t0], np.zeros((M, 3)))
    rep = O.ReprojEdges('deform', f_uv[opt_f[node_idx]], np.zeros(M, np.int64), np.arange(M), X0[node_idx], float(O.INFO_REPROJ), O.TH2)
    dmp = O.DamperDeformEdges(dm_i, dm_j, np.asarray(dm_w, F32).astype(np.float64), info_sp, O.TH3)
    spr = O.SpringDeformEdges(dm_i, dm_j, np.asarray(sp_d0, F32).astype(np.float64),
                              X0[node_idx][np.asarray(dm_i, np.int64)] if E else np.zeros((0, 3)),
                              X0[node_idx][np.asarray(dm_j, np.int64)] if E else np.zeros((0, 3)),
                              float(O.INFO_POSITION), O.TH3)
    G.groups += [rep, dmp, spr]
    skn = None
    if S:
        skn = SkinnedReprojEdges(f_uv[opt_f[sk_idx]], X0[sk_idx], sk_nodes[sk_idx], sk_omega[sk_idx], float(O.INFO_REPROJ), O.TH2)
        G.groups.append(skn)
    inl = np.ones(N, bool)
    for rnd in range(2):                                            # OPT:338-395
        G.pose_q[0], G.pose_t[0] = q0.copy(), t0.copy()
        G.pts[:] = 0
        if G.initialize(0):
            tr = None if trace is None else []
            O.lm_optimize(G, 10, tr, solver)
            if trace is not None:
                trace.append(tr)
        rep.err[:] = rep.residual(G, np.arange(M))
        chi = rep.chi2().astype(F32)
        for v in range(M):
            idx = int(node_idx[v])
            out = bool(chi[v] > O.TH2_SQ)
            inl[idx] = not out
            rep.level[v] = 1 if out else 0
            for io, k in reg[idx].items():
                dmp.level[k] = 1 if out else 0
            for io, k in reg[idx].items():
                dmp.err[k] = dmp.residual(G, np.array([k]))[0]
                dmp.level[k] = 1 if dmp.chi2()[k] > float(O.TH3_SQ) else 0
        if S:
            skn.err[:] = skn.residual(G, np.arange(S))
            chs = skn.chi2().astype(F32)
            out = chs > O.TH2_SQ
            inl[sk_idx] = ~out
            skn.level[:] = np.where(out, 1, 0)
    pose_q_out, pose_t_out = G.pose_q[0].copy(), G.pose_t[0].copy()
    # ---- OPT:401-455 over all optimised points (a skinned point's deformation is its interpolated one)
    delta64 = np.zeros((N, 3))
    delta64[node_idx] = G.pts[:M]
    if S:
        delta64[sk_idx] = skn.deformation(G, np.arange(S))
    delta = delta64.astype(F32)
    mag = np.sqrt((delta[:, 0] * delta[:, 0] + delta[:, 1] * delta[:, 1] + delta[:, 2] * delta[:, 2]).astype(F32)).astype(F32)
    srt = np.sort(mag)
    q1 = srt[int(F32(N) * F32(0.25))]
    q3 = srt[int(F32(N) * F32(0.75))]
    th = F32(1.5) * (q3 - q1)
    chi_all = np.zeros(N, F32)
    rep.err[:] = rep.residual(G, np.arange(M))
    chi_all[node_idx] = rep.chi2().astype(F32)
    if S:
        skn.err[:] = skn.residual(G, np.arange(S))
        chi_all[sk_idx] = skn.chi2().astype(F32)
    for idx in range(N):
        fi = opt_f[idx]
        if chi_all[idx] > O.TH2_SQ:
            inl[idx] = False
            f_status[fi] = O.TRACKED
        if mag[idx] >= q3 + th:
            f_status[fi] = O.TRACKED
            continue
        if is_node[idx]:
            G.pt_fixed[node_of[idx]] = True
        cur = delta[idx] + f_pos[fi]
        f_pos[fi] = cur
        map_pos[ids[idx]] = cur
    median = float(np.partition(mag.copy(), N // 2)[N // 2])
    for idx in range(N):                                            # graph update OPT:457-474
        if not inl[idx]:
            continue
        good = update_vertex_of(int(ids[idx]), map_pos)
        if good < O.REGULARIZERS_PER_POINT * 0.5:
            f_status[opt_f[idx]] = O.BAD
    res = dict(pose_q=pose_q_out, pose_t=pose_t_out, f_pos=f_pos, f_status=f_status, map_pos=map_pos, graph=g, median=median,
               inliers=inl, delta=delta64.copy(), lost=[], n_edges=E, n_nodes=M, n_skinned=S)
    if not lost:
        return res
    # ---- stage 2 OPT:476-553: the lost points follow their optimised neighbours; vertices = nodes, then the skinned points as
    # constants (their interpolated deformation), then the lost points
    lost_sorted = sorted(lost)
    L = len(lost_sorted)
    vert_of = -np.ones(N, np.int64)
    vert_of[node_idx] = np.arange(M)
    other_idx = np.where(~is_node)[0]
    vert_of[other_idx] = M + np.arange(len(other_idx))
    nv = M + len(other_idx)
    G.pts = np.vstack([G.pts[:M], delta64[other_idx], np.zeros((L, 3))])
    G.M = nv + L
    G.pt_fixed = np.concatenate([G.pt_fixed[:M], np.ones(len(other_idx), bool), np.zeros(L, bool)])
    ui, uj, uw = [], [], []
    for li, lid in enumerate(lost_sorted):
        n_reg = 0
        for other, e_w, e_d0, e_st in get_edges_of(lid):
            if n_reg > 10:
                break
            if id_to_idx[other] < 0:
                continue
            ui.append(nv + li); uj.append(int(vert_of[id_to_idx[other]])); uw.append(e_w)
            n_reg += 1
    G.groups.append(O.DamperFixedEdges(ui, uj, np.asarray(uw, F32).astype(np.float64), info_sp, O.TH3))
    if S:
        skn.level[:] = 1                                            # the skinned observations take part in the two rounds only
    G.pose_fixed[0] = True
    if G.initialize(0):
        tr = None if trace is None else []
        O.lm_optimize(G, 10, tr, solver)
        if trace is not None:
            trace.append(tr)
    for li, lid in enumerate(lost_sorted):
        map_pos[lid] = G.pts[nv + li].astype(F32) + map_pos[lid]
    res.update(lost=lost_sorted, map_pos=map_pos)
    return res


# ----------------------------------------------------------------------------------------------------------------------------
# N2b: the embedded form of LocalDeformableBundleAdjustment (OPT:880-1161) -- BASELINE configs[1] as written, "5k map points x
# 500 deformation-graph nodes x 20 keyframes".  The window's VERTICES are the keyframe copies of the NODES (LandmarkVertex per
# (keyframe, node), OPT:985-1004) and the poses; springs (PositionRegularizer, OPT:1031-1072) and dampers (SpatialRegularizer,
# OPT:1076-1132) are built between node copies only, by the reference's own walks with every non-node passed over; every other
# observed point is SKINNED in its keyframe to the <= 11 node copies its walk accepts:
#       x = X0 + sum_k omega_k (x_{n_k} - x0_{n_k}),   omega = w / sum w  (float32 connection weights, summed in float64)
# (X0, x0: the estimates the window starts from), and its ReprojectionError edge (reprojection_error.cc:32-64: residual,
# information 1 / 0.5^2, Huber sqrt(5.99)) constrains those node copies and the keyframe pose -- Jacobian omega_k x the
# reference's 2 x 3 block.  With every point a node the lists are dba_build's and the solve is nrs_oracle.dba_solve, bit for bit
# (tests/test_oracle_embedded_cpu.py): the only pin this mode can have; beyond M = N it is "parity unpinned".
# ----------------------------------------------------------------------------------------------------------------------------
def dba_build_embedded(kf_points, is_node, nbr_rowptr, nbr_col, nbr_w, nbr_d0, nbr_status):
    """kf_points / nbr_* as nrs_oracle.dba_build; is_node[p] != 0: map point p is a node.  Observation o = position in the
    concatenation of kf_points (= landmark index of the plain form).  Returns the node copies (lm_obs: their observation index,
    kf-major; lm_kf, lm_pt), springs / dampers over node-copy indices in reference insertion order, and the skinned observations
    (sk_obs, sk_kf, sk_node [n x 11] node-copy indices with -1 pads, sk_omega [n x 11])."""
    is_node = np.asarray(is_node).astype(bool)
    K = len(kf_points)
    lm_obs, lm_kf, lm_pt = [], [], []
    inserted, observed = [], []
    o = 0
    for k, pts in enumerate(kf_points):
        d = {}
        ob = set()
        for p in pts:
            p = int(p)
            ob.add(p)
            if is_node[p]:
                d[p] = len(lm_kf)
                lm_obs.append(o); lm_kf.append(k); lm_pt.append(p)
            o += 1
        inserted.append(d)
        observed.append(ob)
    sp_i, sp_j, sp_d0, dm, dm_w = [], [], [], [], []
    sk_obs, sk_kf, sk_node, sk_omega = [], [], [], []
    spring_seen, damper_seen = set(), set()
    o = 0
    for k, pts in enumerate(kf_points):
        cur = inserted[k]
        nxt = inserted[k + 1] if k + 1 < K else None
        for p in pts:
            p = int(p)
            lo, hi = nbr_rowptr[p], nbr_rowptr[p + 1]
            if not is_node[p]:
                nodes, ws = [], []
                n_reg = 0
                for e in range(lo, hi):                            # the walk of OPT:1035-1072, nodes accepted, others passed over
                    if n_reg > O.REGULARIZERS_PER_POINT or nbr_status[e] == O.GRAPH_BAD:
                        break
                    q = int(nbr_col[e])
                    if q not in cur:                               # (not observed by the keyframe, or not a node)
                        continue
                    nodes.append(cur[q]); ws.append(float(np.float64(F32(nbr_w[e]))))
                    n_reg += 1
                if nodes:
                    tot = 0.0
                    for w in ws:
                        tot += w
                    sk_obs.append(o); sk_kf.append(k)
                    sk_node.append(nodes + [-1] * (MAX_NODES - len(nodes)))
                    sk_omega.append([w / tot for w in ws] + [0.0] * (MAX_NODES - len(ws)))
                o += 1
                continue
            l = cur[p]
            n_reg = 0
            for e in range(lo, hi):
                if n_reg > O.REGULARIZERS_PER_POINT or nbr_status[e] == O.GRAPH_BAD:
                    break
                q = int(nbr_col[e])
                if q not in cur:
                    continue
                key = (min(p, q), max(p, q), k)
                if key in spring_seen:
                    n_reg += 1
                    continue
                spring_seen.add(key)
                sp_i.append(l); sp_j.append(cur[q]); sp_d0.append(nbr_d0[e])
                n_reg += 1
            if nxt is not None and p in nxt:
                ln = nxt[p]
                n_reg = 0
                for e in range(lo, hi):
                    if n_reg > O.REGULARIZERS_PER_POINT or nbr_status[e] == O.GRAPH_BAD:
                        break
                    q = int(nbr_col[e])
                    if q not in cur or q not in nxt:
                        continue
                    key = (min(p, q), max(p, q), k)
                    if key in damper_seen:
                        n_reg += 1
                        continue
                    damper_seen.add(key)
                    dm.append((l, cur[q], ln, nxt[q])); dm_w.append(nbr_w[e])
                    n_reg += 1
            o += 1
    return dict(lm_obs=np.array(lm_obs, np.int32), lm_kf=np.array(lm_kf, np.int32), lm_pt=np.array(lm_pt, np.int32),
                sp_ij=np.array(list(zip(sp_i, sp_j)), np.int32).reshape(-1, 2), sp_d0=np.array(sp_d0, F32),
                dm_idx=np.array(dm, np.int32).reshape(-1, 4), dm_w=np.array(dm_w, F32),
                sk_obs=np.array(sk_obs, np.int32), sk_kf=np.array(sk_kf, np.int32),
                sk_node=np.array(sk_node, np.int32).reshape(-1, MAX_NODES), sk_omega=np.array(sk_omega, np.float64).reshape(-1, MAX_NODES))


class SkinnedBAReprojEdges(O.EdgeGroup):
    """ReprojectionError (reprojection_error.cc:32-64) of a point without a vertex: x = X0 + sum_k omega_k (pts[node_k] - P0[node_k])."""
    dim = 2

    def __init__(self, uv, pose_idx, X0, nodes, omega, P0, info, delta):
        n = len(uv)
        super().__init__(n, info, delta)
        self.uv = np.asarray(uv, np.float64).reshape(n, 2)
        self.pose_idx = np.asarray(pose_idx, np.int64)
        self.X0 = np.asarray(X0, np.float64).reshape(n, 3)
        nodes = np.asarray(nodes, np.int64).reshape(n, MAX_NODES)
        self.omega = np.where(nodes >= 0, np.asarray(omega, np.float64).reshape(n, MAX_NODES), 0.0)
        self.nodes = np.where(nodes >= 0, nodes, 0)
        self.P0 = np.asarray(P0, np.float64)
        self.slots = [('pose', self.pose_idx)] + [('pt', self.nodes[:, k].copy()) for k in range(MAX_NODES)]

    def world(self, G, idx):
        d = np.zeros((len(idx), 3))
        for k in range(MAX_NODES):                                 # (sequential over the nodes)
            nk = self.nodes[idx, k]
            d += self.omega[idx, k, None] * (G.pts[nk] - self.P0[nk])
        return self.X0[idx] + d

    def _cam(self, G, idx):
        k = self.pose_idx[idx]
        return O._reproj_core(G, G.pose_q[k], G.pose_t[k], self.world(G, idx))

    def residual(self, G, idx):
        p = self._cam(G, idx)
        return self.uv[idx] - O.project_f32(G.cam_model, G.cam_prm, p.astype(F32)).astype(np.float64)

    def jacobians(self, G, idx):
        p = self._cam(G, idx)
        Jp = -O.projection_jacobian_f32(G.cam_model, G.cam_prm, p.astype(F32)).astype(np.float64)
        k = self.pose_idx[idx]
        uk = np.unique(k)
        R = np.stack([O.quat_to_R(G.pose_q[kk]) for kk in uk])
        Rn = R[np.searchsorted(uk, k)]
        Jl = np.einsum('nij,njk->nik', Jp, Rn)
        return [np.einsum('nij,njk->nik', Jp, O._expmap_jac(p))] + [self.omega[idx, q, None, None] * Jl for q in range(MAX_NODES)]


def dba_graph_embedded(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0, dm_idx, dm_w,
                       sk_kf, sk_uv, sk_xyz, sk_node, sk_omega, scale):
    """lm_*: the node copies (vertices); sk_*: the skinned observations (sk_xyz: their positions at the start = X0)."""
    G = O.dba_graph(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0, dm_idx, dm_w, scale)
    skn = None
    if len(sk_kf):
        skn = SkinnedBAReprojEdges(np.asarray(sk_uv, F32), sk_kf, np.asarray(sk_xyz, F32).astype(np.float64), sk_node, sk_omega,
                                   G.pts.copy(), float(O.INFO_REPROJ), O.TH2)
        G.groups.append(skn)
    return G, skn


def dba_solve_embedded(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0, dm_idx, dm_w,
                       sk_kf, sk_uv, sk_xyz, sk_node, sk_omega, scale, iters=5, trace=None, solver=O.solve_spd):
    """optimize(iters) on the embedded window.  Returns poses (q, t), node copies fp64, skinned positions fp64, LM iterations."""
    G, skn = dba_graph_embedded(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0, dm_idx, dm_w,
                                sk_kf, sk_uv, sk_xyz, sk_node, sk_omega, scale)
    G.initialize(0)
    n_it = O.lm_optimize(G, iters, trace, solver)
    sk_out = skn.world(G, np.arange(skn.n)) if skn is not None else np.zeros((0, 3))
    return G.pose_q.copy(), G.pose_t.copy(), G.pts.copy(), sk_out, n_it
