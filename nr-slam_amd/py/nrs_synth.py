"""Seeded synthetic problems for the NR-SLAM hot path (BASELINE.json configs).

The build's own generator: scene layout follows the recipe of SURVEY.md 8(d) -- a smooth
endoscopic-like surface, an arc of keyframe poses looking at its centroid, a low-frequency
per-keyframe deformation field, noisy observations with a few gross outliers, and a
deformation graph whose edges carry {weight, first_distance, status} exactly like
RegularizationGraph::Edge (reference modules/map/regularization_graph.h:49-59).

Everything the C-ABI consumes is produced here as flat arrays; nothing in this file depends on
the oracle.
"""
import numpy as np
from scipy.spatial import cKDTree

F32 = np.float32

PINHOLE, KB8 = 0, 1
HAMLYN_PINHOLE = np.array([766.380279, 766.380279, 304.8638076782227, 258.3343734741211, 0, 0, 0, 0], F32)
ENDOMAPPER_KB8 = np.array([358.6052, 358.7408, 367.6783, 276.3991,
                           -0.1389272, -0.001239606, 0.0009125824, -4.071615e-05], F32)

GRAPH_NEUTRAL, GRAPH_BAD = 2, 3

CONFIGS = {
    # name: (n_points, n_keyframes, seed, camera)
    "C2": (5000, 20, 1, PINHOLE),
    "C3": (10000, 50, 2, PINHOLE),
    "C4": (50000, 200, 3, PINHOLE),
    "C5": (100000, 200, 4, KB8),
}


def _project(model, prm, p):
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    if model == PINHOLE:
        return np.stack([prm[0] * x / z + prm[2], prm[1] * y / z + prm[3]], 1)
    r = np.sqrt(x * x + y * y)
    th = np.arctan2(r, z)
    psi = np.arctan2(y, x)
    rd = th + prm[4] * th ** 3 + prm[5] * th ** 5 + prm[6] * th ** 7 + prm[7] * th ** 9
    return np.stack([prm[0] * rd * np.cos(psi) + prm[2], prm[1] * rd * np.sin(psi) + prm[3]], 1)


def _look_at(C, target):
    z = target - C
    z /= np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])            # rows = camera axes in world  => p_cam = R (X - C)
    return R, -R @ C


def _R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _small_rot(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def surface_points(n, rng, model):
    """Rest shape in mm."""
    if model == PINHOLE:
        x = rng.uniform(-22, 22, n)
        y = rng.uniform(-17, 17, n)
        z = 60 + 8 * np.sin(x / 15) * np.cos(y / 12)
        nrm = np.stack([-(8 / 15) * np.cos(x / 15) * np.cos(y / 12),
                        (8 / 12) * np.sin(x / 15) * np.sin(y / 12), np.ones(n)], 1)
    else:
        # colon-like tube of radius 15 mm around the optical axis, theta in ~10..80 deg
        ang = rng.uniform(0, 2 * np.pi, n)
        z = rng.uniform(3.0, 80.0, n)
        rad = 15 + 1.5 * np.sin(3 * ang) * np.cos(z / 9)
        x, y = rad * np.cos(ang), rad * np.sin(ang)
        nrm = np.stack([-np.cos(ang), -np.sin(ang), np.zeros(n)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.stack([x, y, z], 1), nrm


def _weight(d, sigma):
    """InterpolationWeight (reference geometry_toolbox.cc:26-28): float argument, exp in double,
    result rounded to float (include/nrs.h nrs_graph)."""
    d = np.asarray(d, F32)
    sigma = F32(sigma)
    arg = (-(d * d) / (F32(2) * sigma * sigma)).astype(F32)
    return np.exp(arg.astype(np.float64)).astype(F32)


def build_graph(X, sigma, knn=16, stretch_th=1.1):
    """Deformation graph over points X (fp32, map units) in the flat wire form of
    RegularizationGraph (reference regularization_graph.h:49-59,79,89):
      * undirected edges e: e_i < e_j, first_distance d0, max/min distance, weight, status
      * raw CSR per point, neighbours in ascending index order (= btree_map ID order), each
        directed entry naming its undirected edge (eid).
    Candidate edges = union of k-nearest-neighbour pairs whose initial weight reaches min_weight
    (the reference starts from all pairs, map.cc:148-166; pairs below min_weight can never be
    returned by GetEdges because weights only decrease, regularization_graph.cc:113)."""
    X = np.asarray(X, F32)
    n = len(X)
    sigma = F32(sigma)
    min_w = _weight(F32(float(sigma) * 1.5), sigma)
    k = min(knn + 1, n)
    tree = cKDTree(X.astype(np.float64))
    _, nn = tree.query(X.astype(np.float64), k=k)
    src = np.repeat(np.arange(n), k - 1)
    dst = nn[:, 1:].ravel()
    a, b = np.minimum(src, dst), np.maximum(src, dst)
    keep = a != b
    pairs = np.unique(np.stack([a[keep], b[keep]], 1), axis=0)
    rel = X[pairs[:, 1]] - X[pairs[:, 0]]
    d = np.sqrt((rel[:, 0] * rel[:, 0] + rel[:, 1] * rel[:, 1] + rel[:, 2] * rel[:, 2]).astype(F32)).astype(F32)
    w = _weight(d, sigma)
    ok = w >= min_w
    pairs, d, w = pairs[ok], d[ok], w[ok]
    ne = len(pairs)
    row = np.concatenate([pairs[:, 0], pairs[:, 1]])
    col = np.concatenate([pairs[:, 1], pairs[:, 0]])
    eid = np.concatenate([np.arange(ne), np.arange(ne)])
    order = np.lexsort((col, row))
    row, col, eid = row[order], col[order], eid[order]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, row + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    g = dict(n=n, rowptr=rowptr, col=col.astype(np.int32), eid=eid.astype(np.int32),
             e_ij=pairs.astype(np.int32), e_d0=d.copy(), e_w=w.copy(), e_max=d.copy(), e_min=d.copy(),
             e_status=np.full(ne, GRAPH_NEUTRAL, np.int32), min_w=float(min_w), sigma=float(sigma),
             stretch_th=float(stretch_th))
    g.update(ordered_neighbours(g))
    return g


def ordered_neighbours(g):
    """NumPy twin of GetEdges (reference regularization_graph.cc:61-87) on the flat graph: per
    point, neighbours sorted by (status asc, weight desc, index asc) and cut at the first weight
    below min_weight.  Used to build inputs; the product's version is nrs_graph_select_neighbours."""
    rowptr, col, eid = g["rowptr"], g["col"], g["eid"]
    n = len(rowptr) - 1
    row = np.repeat(np.arange(n), np.diff(rowptr))
    w = g["e_w"][eid]
    st = g["e_status"][eid]
    order = np.lexsort((col, -w.astype(np.float64), st, row))
    r2, c2, e2, w2 = row[order], col[order], eid[order], w[order]
    low = w2 < np.float32(g["min_w"])
    # cut every row at its first low-weight entry
    pos = np.arange(len(r2)) - rowptr[r2]
    first_low = np.full(n, np.iinfo(np.int64).max)
    np.minimum.at(first_low, r2[low], pos[low])
    keep = pos < first_low[r2]
    r2, c2, e2 = r2[keep], c2[keep], e2[keep]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r2 + 1, 1)
    return dict(o_rowptr=np.cumsum(rp).astype(np.int32), o_col=c2.astype(np.int32), o_eid=e2.astype(np.int32))


def ordered_view(g):
    """Ordered neighbour lists with per-entry weight / first_distance / status (input of
    nrs_dba_build_edges and of the oracle's dba_build)."""
    e = g["o_eid"]
    return dict(rowptr=g["o_rowptr"], col=g["o_col"], w=g["e_w"][e], d0=g["e_d0"][e], status=g["e_status"][e])


def make_scene(n_points, n_kf, seed, model=PINHOLE, dropout=0.05, outlier_frac=0.02,
               img_wh=(640, 480), knn=16):
    """Full synthetic sequence: rest shape, per-keyframe truth, observations, graph.
    Positions are in *map units* (mm * scale, scale = 3/median depth: reference
    tracking.cc:156-157)."""
    rng = np.random.default_rng(seed)
    prm = HAMLYN_PINHOLE if model == PINHOLE else ENDOMAPPER_KB8
    if model == KB8:
        img_wh = (736, 552)
    Xmm, nrm = surface_points(n_points, rng, model)
    centroid = Xmm.mean(0)
    if model == PINHOLE:
        depth0 = Xmm[:, 2]
    else:
        depth0 = Xmm[:, 2]
    scale = F32(3.0) / F32(np.median(depth0))
    sigma = F32(3.0) * F32(np.std(depth0.astype(F32))) * scale      # tracking.cc:159-160,200
    poses_R, poses_t = [], []
    phis = np.linspace(-1.0, 1.0, n_kf) if n_kf > 1 else np.array([0.0])
    for phi in phis:
        if model == PINHOLE:
            C = np.array([6.0 * phi, 2.0 * np.sin(np.pi * phi), 0.0])
            R, t = _look_at(C, centroid)
        else:
            C = np.array([1.5 * phi, 1.0 * np.sin(np.pi * phi), 2.0 * phi])
            R = _small_rot(np.array([0.03 * phi, -0.04 * phi, 0.02 * phi]))
            t = -R @ C
        poses_R.append(R)
        poses_t.append(t)
    # per-keyframe true positions (mm): low-frequency deformation along the normal
    obs = []
    Xk_true = []
    A = 1.5
    for k in range(n_kf):
        amp = A * np.sin(2 * np.pi * k / 20.0) * (0.6 + 0.4 * np.sin(Xmm[:, 0] / 11.0 + 0.3) * np.cos(Xmm[:, 1] / 9.0))
        Xk = Xmm + (amp + rng.normal(0, 0.05, n_points))[:, None] * nrm
        Xk_true.append(Xk)
        pc = Xk @ poses_R[k].T + poses_t[k]
        uv = _project(model, prm.astype(np.float64), pc)
        vis = (pc[:, 2] > 1.0) & (uv[:, 0] > 12) & (uv[:, 0] < img_wh[0] - 12) & (uv[:, 1] > 12) & (uv[:, 1] < img_wh[1] - 12)
        vis &= rng.uniform(size=n_points) > dropout
        uv = uv + rng.normal(0, 0.5, uv.shape)
        out = rng.uniform(size=n_points) < outlier_frac
        uv[out] = np.stack([rng.uniform(12, img_wh[0] - 12, out.sum()), rng.uniform(12, img_wh[1] - 12, out.sum())], 1)
        obs.append((np.where(vis)[0].astype(np.int32), uv[vis].astype(F32)))
    s = float(scale)
    X0 = (Xmm * s).astype(F32)
    graph = build_graph(X0, sigma, knn)
    return dict(model=model, prm=prm, scale=float(scale), sigma=float(sigma), X0=X0,
                Xk_true=[(x * s) for x in Xk_true], poses_R=poses_R,
                poses_t=[t * s for t in poses_t], obs=obs, graph=graph, rng=rng, img_wh=img_wh)


def make_dba_problem(name_or_n, n_kf=None, seed=None, model=PINHOLE, pose_noise=(0.002, 0.01),
                     lm_noise=0.004, **kw):
    """Flat input of nrs_dba_* for one windowed deformable BA (reference
    LocalDeformableBundleAdjustment, g2o_optimization.cc:880-1161), with the window cap lifted
    to n_kf keyframes.  Edge lists are NOT built here: the product's host builder
    (nrs_dba_build_edges) and the oracle's dba_build both derive them from (kf_points, graph)."""
    if isinstance(name_or_n, str):
        n_points, n_kf, seed, model = CONFIGS[name_or_n]
    else:
        n_points = name_or_n
    sc = make_scene(n_points, n_kf, seed, model, **kw)
    rng = sc["rng"]
    poses_q, poses_t = [], []
    for k in range(n_kf):
        dR = _small_rot(rng.normal(0, pose_noise[0], 3))
        R = dR @ sc["poses_R"][k]
        t = dR @ sc["poses_t"][k] + rng.normal(0, pose_noise[1], 3)
        q = _R_to_quat(R).astype(F32).astype(np.float64)          # boundary hands over SE3f
        poses_q.append(q)
        poses_t.append(t.astype(F32).astype(np.float64))
    kf_points = [o[0] for o in sc["obs"]]
    lm_uv = np.concatenate([o[1] for o in sc["obs"]]).astype(F32)
    lm_xyz = np.concatenate([sc["Xk_true"][k][kf_points[k]] + rng.normal(0, lm_noise, (len(kf_points[k]), 3))
                             for k in range(n_kf)]).astype(F32)
    lm_kf = np.concatenate([np.full(len(kf_points[k]), k, np.int32) for k in range(n_kf)])
    lm_pt = np.concatenate(kf_points).astype(np.int32)
    return dict(model=sc["model"], prm=sc["prm"], scale=sc["scale"], n_kf=n_kf, n_points=n_points,
                poses_q=np.array(poses_q), poses_t=np.array(poses_t), kf_points=kf_points,
                lm_xyz=lm_xyz, lm_kf=lm_kf, lm_pt=lm_pt, lm_uv=lm_uv, graph=sc["graph"],
                nbr=ordered_view(sc["graph"]), scene=sc)


def make_tracking_problem(n_points, seed, model=PINHOLE, frame=3, lost_frac=0.03, **kw):
    """One tracked frame for nrs_pose_only_solve / nrs_track_deform_solve (reference
    CameraPoseOptimization / CameraPoseAndDeformationOptimization, g2o_optimization.cc:50-557):
    previous positions X0 (= last estimate), current observations, pose seed = previous pose."""
    sc = make_scene(n_points, frame + 1, seed, model, **kw)
    rng = sc["rng"]
    idx, uv = sc["obs"][frame]
    prev = sc["Xk_true"][frame - 1] + rng.normal(0, 0.002, (n_points, 3))
    R = _small_rot(rng.normal(0, 0.003, 3)) @ sc["poses_R"][frame - 1]
    t = sc["poses_t"][frame - 1] + rng.normal(0, 0.01, 3)
    status = np.full(n_points, 3, np.int32)             # BAD unless observed
    status[idx] = 0                                     # TRACKED_WITH_3D
    lost = idx[rng.uniform(size=len(idx)) < lost_frac]
    status[lost] = 1                                    # TRACKED (no 3D this frame) -> "lost" neighbours
    uv_full = np.zeros((n_points, 2), F32)
    uv_full[idx] = uv
    return dict(model=sc["model"], prm=sc["prm"], scale=sc["scale"], n_points=n_points,
                status=status, uv=uv_full, X_prev=prev.astype(F32),
                pose_q=_R_to_quat(R).astype(F32).astype(np.float64),
                pose_t=t.astype(F32).astype(np.float64), graph=sc["graph"], scene=sc)


def make_lk_sequence(n_points=500, seed=5, wh=(640, 480), flow_px=6.0):
    """Synthetic LK pair (SURVEY.md 8d "LK"): band-limited noise texture, second frame = smooth warp
    (<= flow_px) with gain 0.8-1.2 and bias +-10; points at high-gradient locations."""
    rng = np.random.default_rng(seed)
    w, h = wh
    tex = np.zeros((h + 64, w + 64), np.float64)
    for sc, amp in ((4, 28.0), (8, 36.0), (16, 30.0), (32, 22.0)):
        gh, gw = (h + 64) // sc + 3, (w + 64) // sc + 3
        g = rng.normal(0, 1, (gh, gw))
        yy = np.arange(h + 64) / sc
        xx = np.arange(w + 64) / sc
        y0, x0 = np.floor(yy).astype(int), np.floor(xx).astype(int)
        fy, fx = (yy - y0)[:, None], (xx - x0)[None, :]
        fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
        tex += amp * ((g[np.ix_(y0, x0)] * (1 - fx) + g[np.ix_(y0, x0 + 1)] * fx) * (1 - fy)
                      + (g[np.ix_(y0 + 1, x0)] * (1 - fx) + g[np.ix_(y0 + 1, x0 + 1)] * fx) * fy)
    tex = 120 + tex

    def render(flow_fn, gain, bias):
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
        dx, dy = flow_fn(xs, ys)
        sx, sy = xs - dx + 32, ys - dy + 32                 # backward warp
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        fx, fy = sx - x0, sy - y0
        x0 = np.clip(x0, 0, w + 62)
        y0 = np.clip(y0, 0, h + 62)
        v = (tex[y0, x0] * (1 - fx) + tex[y0, x0 + 1] * fx) * (1 - fy) + (tex[y0 + 1, x0] * (1 - fx) + tex[y0 + 1, x0 + 1] * fx) * fy
        return np.clip(np.rint(gain * v + bias), 0, 255).astype(np.uint8)

    zero = lambda xs, ys: (np.zeros_like(xs), np.zeros_like(ys))
    ax, ay = rng.uniform(0.5, 1.0) * flow_px, rng.uniform(0.5, 1.0) * flow_px
    flow = lambda xs, ys: (ax * np.sin(xs / 90.0 + 0.3) * np.cos(ys / 70.0), ay * np.cos(xs / 80.0) * np.sin(ys / 60.0 + 0.5))
    im0 = render(zero, 1.0, 0.0)
    im1 = render(flow, rng.uniform(0.8, 1.2), rng.uniform(-10, 10))
    # corners: strongest gradient-energy responses on a coarse grid, away from the border
    g = im0.astype(np.float64)
    gx = np.zeros_like(g); gy = np.zeros_like(g)
    gx[:, 1:-1] = g[:, 2:] - g[:, :-2]
    gy[1:-1, :] = g[2:, :] - g[:-2, :]
    e = gx * gx + gy * gy
    pts = []
    cell = max(8, int(np.sqrt(w * h / max(1, n_points)) * 0.9))
    for y in range(30, h - 30 - cell, cell):
        for x in range(30, w - 30 - cell, cell):
            blk = e[y:y + cell, x:x + cell]
            k = np.unravel_index(np.argmax(blk), blk.shape)
            pts.append((x + k[1] + rng.uniform(-0.4, 0.4), y + k[0] + rng.uniform(-0.4, 0.4)))
    pts = np.array(pts, F32)
    if len(pts) > n_points:
        pts = pts[rng.choice(len(pts), n_points, replace=False)]
    fx, fy = flow(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64))
    truth = pts + np.stack([fx, fy], 1).astype(F32)
    return dict(im0=im0, im1=im1, pts=pts, truth=truth)


def _texture(h, w, rng, margin=32):
    tex = np.zeros((h + 2 * margin, w + 2 * margin), np.float64)
    for sc, amp in ((4, 28.0), (8, 36.0), (16, 30.0), (32, 22.0)):
        gh, gw = (h + 2 * margin) // sc + 3, (w + 2 * margin) // sc + 3
        g = rng.normal(0, 1, (gh, gw))
        yy = np.arange(h + 2 * margin) / sc
        xx = np.arange(w + 2 * margin) / sc
        y0, x0 = np.floor(yy).astype(int), np.floor(xx).astype(int)
        fy, fx = (yy - y0)[:, None], (xx - x0)[None, :]
        fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
        tex += amp * ((g[np.ix_(y0, x0)] * (1 - fx) + g[np.ix_(y0, x0 + 1)] * fx) * (1 - fy)
                      + (g[np.ix_(y0 + 1, x0)] * (1 - fx) + g[np.ix_(y0 + 1, x0 + 1)] * fx) * fy)
    return 120 + tex


def make_frame_sequence(n_points=400, n_frames=6, seed=9, model=PINHOLE, step=0.012, occluders=2):
    """A short consistent monocular sequence for the frame-loop harness: a deforming
    surface seen by a slowly moving camera.  Frame 0 is the initialised map (keyframe): 3D positions
    (truth + noise, map units), keypoints, regularisation graph; frames 1.. are images only.  Images
    are one texture warped by the dense flow that interpolates the projected point motions, so the LK
    tracker sees what the geometry says."""
    from scipy.interpolate import griddata
    rng = np.random.default_rng(seed)
    prm = HAMLYN_PINHOLE if model == PINHOLE else ENDOMAPPER_KB8
    w, h = (640, 480) if model == PINHOLE else (736, 552)
    Xmm, nrm = surface_points(n_points, rng, model)
    centroid = Xmm.mean(0)
    scale = F32(3.0) / F32(np.median(Xmm[:, 2]))
    sigma = F32(3.0) * F32(np.std(Xmm[:, 2].astype(F32))) * scale
    s = float(scale)
    poses_R, poses_t, Xt, uvt = [], [], [], []
    for f in range(n_frames):
        phi = step * f
        if model == PINHOLE:
            R, t = _look_at(np.array([6.0 * phi, 2.0 * np.sin(np.pi * phi), 0.0]), centroid)
        else:
            R = _small_rot(np.array([0.03 * phi, -0.04 * phi, 0.02 * phi]))
            t = -R @ np.array([1.5 * phi, 1.0 * np.sin(np.pi * phi), 2.0 * phi])
        amp = 1.2 * np.sin(2 * np.pi * f / 40.0) * (0.6 + 0.4 * np.sin(Xmm[:, 0] / 11.0 + 0.3) * np.cos(Xmm[:, 1] / 9.0))
        X = Xmm + amp[:, None] * nrm
        pc = X @ R.T + t
        poses_R.append(R); poses_t.append(t * s); Xt.append(X * s)
        uvt.append(_project(model, prm.astype(np.float64), pc))
    uv0 = uvt[0]
    vis = (uv0[:, 0] > 30) & (uv0[:, 0] < w - 30) & (uv0[:, 1] > 30) & (uv0[:, 1] < h - 30)
    idx = np.where(vis)[0]
    tex = _texture(h, w, rng)
    occl = [np.clip(np.rint(_texture(46, 58, rng, 0)), 0, 255).astype(np.uint8) for _ in range(occluders)]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    images = []
    for f in range(n_frames):
        if f == 0:
            sx, sy = xs, ys
        else:
            # backward warp: displacement known at the points' positions in frame f, interpolated densely
            d = uv0[idx] - uvt[f][idx]
            pts = uvt[f][idx]
            gx = griddata(pts, d[:, 0], (xs, ys), method="linear")
            gy = griddata(pts, d[:, 1], (xs, ys), method="linear")
            nx = griddata(pts, d[:, 0], (xs, ys), method="nearest")
            ny = griddata(pts, d[:, 1], (xs, ys), method="nearest")
            gx = np.where(np.isnan(gx), nx, gx); gy = np.where(np.isnan(gy), ny, gy)
            sx, sy = xs + gx, ys + gy
        sx2, sy2 = np.clip(sx + 32, 0, w + 62.0), np.clip(sy + 32, 0, h + 62.0)
        x0, y0 = np.floor(sx2).astype(int), np.floor(sy2).astype(int)
        x0, y0 = np.minimum(x0, w + 62), np.minimum(y0, h + 62)
        fx, fy = sx2 - x0, sy2 - y0
        v = (tex[y0, x0] * (1 - fx) + tex[y0, x0 + 1] * fx) * (1 - fy) + (tex[y0 + 1, x0] * (1 - fx) + tex[y0 + 1, x0 + 1] * fx) * fy
        img = np.clip(np.rint(v), 0, 255).astype(np.uint8)
        # moving occluders (from frame 2 on): points under them fail the tracker / its SSIM gate, are
        # re-found by the point-reuse step once the patch has moved on
        for k in range(occluders if f >= 2 else 0):
            cx = int(w * (0.25 + 0.5 * k / max(1, occluders)) + 40 * (f - 2)) % w
            cy = int(h * (0.35 + 0.2 * k))
            ph, pw = 46, 58
            y1, x1 = max(0, cy - ph // 2), max(0, cx - pw // 2)
            patch = occl[k][:max(0, min(ph, h - y1)), :max(0, min(pw, w - x1))]
            img[y1:y1 + patch.shape[0], x1:x1 + patch.shape[1]] = patch
        images.append(img)
    X0 = (Xt[0][idx] + rng.normal(0, 0.002, (len(idx), 3))).astype(F32)
    graph = build_graph(X0, sigma, 16)
    return dict(model=model, prm=prm, scale=float(scale), wh=(w, h), images=images, n_points=len(idx),
                kp0=uv0[idx].astype(F32), X0=X0, graph=graph,
                pose_q=[_R_to_quat(R).astype(F32) for R in poses_R], pose_t=[t.astype(F32) for t in poses_t],
                uv_true=[u[idx].astype(F32) for u in uvt], X_true=[x[idx] for x in Xt])


def edge_checksum(e):
    """order-sensitive 64-bit checksum of a BA edge list (sp_ij, sp_d0, dm_idx, dm_w): golden files carry it so that
    a trace is only ever compared on the edge list it was computed from"""
    import zlib
    h = 0
    for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w"):
        a = np.ascontiguousarray(e[key])
        a = a.astype(np.int32) if a.dtype.kind == "i" else a.astype(np.float32)
        h = (h * 1000003 + zlib.crc32(a.tobytes())) & 0xFFFFFFFFFFFF
    return int(h)


def make_temporal_buffer(n_frames=12, seed=3, model=PINHOLE, spacing=26.0, cand_frac=0.3, lm_noise=0.004, kp_noise=0.25,
                         baseline=0.35):
    """A flat TemporalBuffer (reference modules/map/temporal_buffer.h:43-63) for DeformableTriangulation: n_frames
    snapshots of a deforming surface seen by a translating camera.  Features sit on a jittered image grid (the
    extractor's non-maximum suppression keeps features apart; GetClosestMapPointsToFeature rejects a candidate with a
    3D neighbour closer than 20 px): (1 - cand_frac) of them are map points (landmark position in every snapshot they
    are seen in, TRACKED_WITH_3D in the last one), the rest are 2D-only candidates with tracks of varying length.
    Returns the dict nrs.triangulate_batch / oracle/triang_oracle.py read, plus `cand` (ids) and `truth` (world
    positions of the candidates in the last frame)."""
    rng = np.random.default_rng(seed)
    prm = HAMLYN_PINHOLE if model == PINHOLE else ENDOMAPPER_KB8
    w, h = (640, 480) if model == PINHOLE else (736, 552)
    gx, gy = np.meshgrid(np.arange(40, w - 40, spacing), np.arange(40, h - 40, spacing))
    uv0 = np.stack([gx.ravel(), gy.ravel()], 1) + rng.uniform(-2.0, 2.0, (gx.size, 2))
    n = len(uv0)
    # back-project the grid onto the surface z = 60 + 8 sin(x/15) cos(y/12) (two fixed-point steps are plenty)
    P = prm.astype(np.float64)
    z = np.full(n, 60.0)
    for _ in range(3):
        if model == PINHOLE:
            x, y = (uv0[:, 0] - P[2]) / P[0] * z, (uv0[:, 1] - P[3]) / P[1] * z
        else:
            th = np.hypot((uv0[:, 0] - P[2]) / P[0], (uv0[:, 1] - P[3]) / P[1])
            psi = np.arctan2((uv0[:, 1] - P[3]) / P[1], (uv0[:, 0] - P[2]) / P[0])
            x, y = z * np.tan(th) * np.cos(psi), z * np.tan(th) * np.sin(psi)
        z = 60.0 + 8.0 * np.sin(x / 15.0) * np.cos(y / 12.0)
    Xmm = np.stack([x, y, z], 1)
    nrm = np.array([0.0, 0.0, 1.0])
    scale = 3.0 / np.median(z)
    is_cand = rng.uniform(size=n) < cand_frac
    poses, has_kp, kp_xy, has_lm, lm_xyz = [], np.zeros((n_frames, n), bool), np.zeros((n_frames, n, 2), F32), np.zeros((n_frames, n), bool), np.zeros((n_frames, n, 3), F32)
    start = np.where(is_cand, rng.integers(0, n_frames - 2, n), 0)           # candidates appear at different times
    truth = np.zeros((n, 3))
    for f in range(n_frames):
        R = _small_rot(np.array([0.002 * f, -0.003 * f, 0.001 * f]))
        C = np.array([baseline * f, 0.12 * baseline * f, 0.0])              # camera centre (mm)
        t = -R @ C
        amp = 0.8 * np.sin(2 * np.pi * f / 30.0) * (0.6 + 0.4 * np.sin(Xmm[:, 0] / 11.0 + 0.3) * np.cos(Xmm[:, 1] / 9.0))
        X = Xmm + amp[:, None] * nrm
        uv = _project(model, P, X @ R.T + t) + rng.normal(0, kp_noise, (n, 2))
        q = _rot_to_quat(R)
        poses.append(np.concatenate([q, t * scale]))
        seen = (f >= start) & (uv[:, 0] > 12) & (uv[:, 0] < w - 12) & (uv[:, 1] > 12) & (uv[:, 1] < h - 12)
        has_kp[f] = seen
        kp_xy[f] = uv.astype(F32)
        has_lm[f] = seen & ~is_cand & (rng.uniform(size=n) > 0.03)          # a few map points drop out of a snapshot
        lm_xyz[f] = (X * scale + rng.normal(0, lm_noise, (n, 3))).astype(F32)
        truth = X * scale
    status = np.where(is_cand, 1, 0).astype(np.int32)                        # TRACKED (2D) / TRACKED_WITH_3D
    status[~has_kp[-1]] = 3
    cand = np.where(is_cand & has_kp[-1])[0].astype(np.int32)
    return dict(n_frames=n_frames, poses=np.array(poses, F32), has_kp=has_kp, kp_xy=kp_xy, has_lm=has_lm, lm_xyz=lm_xyz,
                status=status, cand=cand, truth=truth[cand].astype(F32), model=model, prm=prm, scale=F32(scale))


def _rot_to_quat(R):
    """unit quaternion (x y z w) of a rotation matrix, w >= 0"""
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def pick_nodes(points_xyz, n_nodes, start=0):
    """Farthest-point sampling of n_nodes map points (float64, ties to the lowest index): the node set of the synthetic
    embedded-deformation windows.  Returns a uint8 flag per map point."""
    X = np.asarray(points_xyz, np.float64)
    flag = np.zeros(len(X), np.uint8)
    if n_nodes >= len(X):
        flag[:] = 1
        return flag
    d = np.full(len(X), np.inf)
    cur = start
    for _ in range(n_nodes):
        flag[cur] = 1
        d = np.minimum(d, np.sum((X - X[cur]) ** 2, axis=1))
        d[flag == 1] = -1.0
        cur = int(np.argmax(d))
    return flag


def node_lists(X, sigma, flag, k=24):
    """Ordered neighbour lists for the EMBEDDED windows: per map point the k nearest NODES (itself excluded) whose weight reaches
    min_weight, by (weight desc, index asc), status NEUTRAL -- the prefix a GetEdges list of the reference's all-pairs graph
    (map.cc:148-166, regularization_graph.cc:61-87) shows once everything but the nodes is passed over; the walks of the embedded
    mode accept <= 11 of them.  Same wire form as ordered_view()."""
    X = np.asarray(X, F32)
    sigma = F32(sigma)
    nodes = np.where(np.asarray(flag) != 0)[0]
    min_w = _weight(F32(float(sigma) * 1.5), sigma)
    kk = min(k + 1, len(nodes))
    tree = cKDTree(X[nodes].astype(np.float64))
    _, nn = tree.query(X.astype(np.float64), k=kk)
    nn = nn.reshape(len(X), kk)
    rowptr = np.zeros(len(X) + 1, np.int32)
    col, w, d0 = [], [], []
    for p in range(len(X)):
        c = nodes[nn[p]]
        c = c[c != p]
        rel = X[c] - X[p]
        d = np.sqrt((rel[:, 0] * rel[:, 0] + rel[:, 1] * rel[:, 1] + rel[:, 2] * rel[:, 2]).astype(F32)).astype(F32)
        ww = _weight(d, sigma)
        keep = ww >= min_w
        c, d, ww = c[keep], d[keep], ww[keep]
        o = np.lexsort((c, -ww.astype(np.float64)))
        col.append(c[o]); w.append(ww[o]); d0.append(d[o])
        rowptr[p + 1] = rowptr[p] + len(o)
    col = np.concatenate(col).astype(np.int32)
    return dict(rowptr=rowptr, col=col, w=np.concatenate(w).astype(F32), d0=np.concatenate(d0).astype(F32), status=np.full(len(col), GRAPH_NEUTRAL, np.int32))


def embedded_problem(p, n_nodes, k=24):
    """node flags (farthest-point sampling on the rest shape) and the node lists of a window p (make_dba_problem)"""
    flag = pick_nodes(p["scene"]["X0"], n_nodes)
    return flag, node_lists(p["scene"]["X0"], p["scene"]["sigma"], flag, k)


def embedded_window(p, e):
    """Inputs of nrs_dba_*_embedded from a plain window p (make_dba_problem) and the embedded edge lists e (dba_build_embedded /
    nrs_dba_build_edges_embedded: node copies lm_obs, skinned observations sk_obs as indices into the window's observations)."""
    lo, so = e["lm_obs"], e["sk_obs"]
    return dict(lm_xyz=p["lm_xyz"][lo], lm_kf=p["lm_kf"][lo], lm_uv=p["lm_uv"][lo],
                sk_kf=p["lm_kf"][so], sk_uv=p["lm_uv"][so], sk_xyz=p["lm_xyz"][so])


def nd_block_system(n, seed=3, knn=11):
    """An SPD block system with the structure of a2's single-frame problem (input of nrs_debug_nd_solve): n points on a surface, 3 x 3
    couplings to the knn nearest neighbours, two `last` blocks (the halves of the pose) coupled to every point, block-diagonally dominant.
    Returns pos, last, pairs, Dn, Vp, bn (no dense copy: usable at 5k points)."""
    rng = np.random.default_rng(seed)
    pts = np.c_[rng.uniform(-20, 20, n), rng.uniform(-15, 15, n), 60 + rng.normal(0, 1, n)]
    _, nn = cKDTree(pts[:, :2]).query(pts[:, :2], k=min(knn + 1, n))
    a = np.repeat(np.arange(n), nn.shape[1] - 1)
    b = nn[:, 1:].ravel()
    pr = np.unique(np.stack([np.minimum(a, b), np.maximum(a, b)], 1), axis=0)
    pr = pr[pr[:, 0] != pr[:, 1]]
    pp = np.array([(n + h, i) if (i + h) % 2 else (i, n + h) for i in range(n) for h in range(2)] + [(n, n + 1)], np.int64)
    pairs = np.r_[pr, pp].astype(np.int32)
    N = n + 2
    Vp = rng.normal(0, 1, (len(pairs), 3, 3)) * 0.3
    rs = np.zeros((N, 3))
    np.add.at(rs, pairs[:, 0], np.abs(Vp).sum(2))
    np.add.at(rs, pairs[:, 1], np.abs(Vp).sum(1))
    S_ = rng.normal(0, 0.2, (N, 3, 3))
    Dn = np.einsum('nij,nkj->nik', S_, S_) + np.eye(3)[None] * (rs.max(1)[:, None, None] + 0.5)
    pos = np.r_[pts, np.zeros((2, 3))]
    last = np.zeros(N, np.uint8)
    last[n:] = 1
    return pos, last, pairs, Dn, Vp, rng.normal(0, 1, (N, 3))
