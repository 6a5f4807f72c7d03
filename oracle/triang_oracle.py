"""CPU restatement of DeformableTriangulation (TEST INFRASTRUCTURE ONLY).

Follows modules/optimization/g2o_optimization.cc:559-814 with
  TemporalBuffer::GetFeatureTrack / GetClosestMapPointsToFeature / GetLandmarkPosition  modules/map/temporal_buffer.cc:97-204
  TriangulateMidPoint, RaysParallax, SquaredReprojectionError                           modules/utilities/geometry_toolbox.cc:30-79
  ReprojectionErrorOnlyDeformation (NO analytic Jacobian: g2o differentiates numerically,
    delta = 1e-9 central, through the fp32 projection)                                   modules/optimization/reprojection_error_only_deformation.cc:32-38,
                                                                                         third_party/g2o/g2o/core/base_fixed_sized_edge.hpp:159-200
  SpatialRegularizerWithObservation (Jacobian as written: +-w I, rotations ignored)      modules/optimization/spatial_regularizer_with_observation.cc:33-52
  PinHole / KannalaBrandt8 Unproject                                                     modules/calibration/pin_hole.cc:33-38, kannala_brandt_8.cc:53-85
  Sophus SE3f algebra (float)                                                            third_party/Sophus/sophus/so3.hpp:246-248,358-395, se3.hpp:222-225
and the LM of oracle/nrs_oracle.py (lm_optimize).  The flat "temporal buffer" it reads is the wire form the product's
nrs_triangulate_batch takes (include/nrs.h).  Parity unpinned (no reference vector exists for this function); the
numeric Jacobian is quantisation noise by construction, so the product is held to this restatement on
status codes, and on the triangulated point within a tolerance, not on iterates."""
import numpy as np

import nrs_oracle as O

F32 = np.float32
OK, E_CLOSE, E_REPROJ1, E_REPROJ2, E_PARALLAX, E_NO_NEIGHBOUR, E_NEG_DEPTH, E_EMPTY, E_BAD_NEIGHBOURS, E_BAD_ERROR, E_SHORT = range(11)
STATUS_TEXT = ["ok", "Feature too close to other ones.", "High reprojection error at first camera.",
               "High reprojection error at second camera.", "Low parallax.", "Found no neighbours in a temporal point.",
               "Negative initial depth.", "Optimization is empty.", "Triangulation has to many bad neighbors.",
               "Triangulation has to much error.", "Short track"]


# ------------------------------------------------------------------------------------------------ Sophus SE3f in float
def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F32)


def so3_mul_point(q, p):            # so3.hpp:388-395
    qv = q[:3]
    uv = _cross(qv, p)
    uv = (uv + uv).astype(F32)
    return (p + q[3] * uv + _cross(qv, uv)).astype(F32)


def se3_mul_point(T, p):
    return (so3_mul_point(T[:4], np.asarray(p, F32)) + T[4:]).astype(F32)


def se3_inverse(T):                 # se3.hpp:222-225
    qi = np.array([-T[0], -T[1], -T[2], T[3]], F32)
    return np.concatenate([qi, so3_mul_point(qi, (T[4:] * F32(-1)).astype(F32))]).astype(F32)


def quat_mul_f32(a, b):             # so3.hpp QuaternionProduct
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                     a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
                     a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0],
                     a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]], F32)


def se3_mul(A, B):
    q = quat_mul_f32(A[:4], B[:4])
    return np.concatenate([q, (so3_mul_point(A[:4], B[4:]) + A[4:]).astype(F32)]).astype(F32)


def quat_to_R_f32(q):               # Eigen toRotationMatrix, float
    x, y, z, w = q
    tx, ty, tz = F32(2) * x, F32(2) * y, F32(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[F32(1) - (tyy + tzz), txy - twz, txz + twy], [txy + twz, F32(1) - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, F32(1) - (txx + tyy)]], F32)


def _norm(v):
    return F32(np.sqrt(F32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))


def _normalized(v):
    return (v / _norm(v)).astype(F32)


# ------------------------------------------------------------------------------------------------ camera, fp32
def unproject_f32(model, prm, u, v):
    prm = np.asarray(prm, F32)
    x, y = (F32(u) - prm[2]) / prm[0], (F32(v) - prm[3]) / prm[1]
    if model == O.PINHOLE:
        return np.array([x, y, F32(1)], F32)
    k0, k1, k2, k3 = prm[4:8]
    theta_d = F32(np.sqrt(F32(x * x + y * y)))
    th = F32(0)
    if theta_d > F32(1e-8):
        theta = theta_d
        for _ in range(10):
            t2 = theta * theta
            t4 = t2 * t2
            t6 = t4 * t2
            t8 = t4 * t4
            a, b, c, d = k0 * t2, k1 * t4, k2 * t6, k3 * t8
            fix = (theta * (F32(1) + a + b + c + d) - theta_d) / (F32(1) + F32(3) * a + F32(5) * b + F32(7) * c + F32(9) * d)
            theta = F32(theta - fix)
            if abs(fix) < F32(1e-6):
                break
        th = theta
    # sin / cos of a float through double (std::sin(float) promotes where the reference mixes; fixed convention)
    s, c = F32(np.sin(np.float64(th))), F32(np.cos(np.float64(th)))
    return np.array([s * x / theta_d, s * y / theta_d, c], F32)


def project_pt(model, prm, p):
    return O.project_f32(model, prm, np.asarray(p, F32)[None, :])[0]


# ------------------------------------------------------------------------------------------------ geometry_toolbox.cc
def triangulate_mid_point(ray_1, ray_2, T1, T2):                  # :45-79
    f0_hat, f1_hat = _normalized(ray_1), _normalized(ray_2)
    T10 = se3_mul(T2, se3_inverse(T1))
    t = T10[4:]
    R = quat_to_R_f32(T10[:4])
    # R * f0_hat: row dot products in float, left to right
    Rf0 = np.array([F32(F32(R[i, 0] * f0_hat[0] + R[i, 1] * f0_hat[1]) + R[i, 2] * f0_hat[2]) for i in range(3)], F32)
    p, q, r = _cross(Rf0, f1_hat), _cross(Rf0, t), _cross(f1_hat, t)
    x1 = (_norm(q) / (_norm(q) + _norm(r)) * (t + _norm(r) / _norm(p) * (Rf0 + f1_hat))).astype(F32)
    return se3_mul_point(se3_inverse(T2), x1)


def rays_parallax(a, b):                                          # :36-43
    dot = F32(F32(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])
    c = dot / (_norm(a) * _norm(b))
    return F32(np.arccos(np.float64(min(c, F32(1)))))


# ------------------------------------------------------------------------------------------------ edges for the oracle's LM
class ReprojNumeric(O.EdgeGroup):
    """ReprojectionErrorOnlyDeformation: r = z - Project(x), x in the camera frame; Jacobian = g2o's central difference"""
    dim = 2

    def __init__(self, uv, info):
        n = len(uv)
        super().__init__(n, info, None)
        self.uv = np.asarray(uv, np.float64)
        self.slots = [('pt', np.arange(n))]

    def _res(self, G, x, idx):
        return self.uv[idx] - O.project_f32(G.cam_model, G.cam_prm, x.astype(F32)).astype(np.float64)

    def residual(self, G, idx):
        return self._res(G, G.pts[idx], idx)

    def jacobians(self, G, idx):
        delta = 1e-9
        J = np.zeros((len(idx), 2, 3))
        for d in range(3):
            xp, xm = G.pts[idx].copy(), G.pts[idx].copy()
            xp[:, d] += delta
            xm[:, d] += -delta
            J[:, :, d] = (1 / (2 * delta)) * (self._res(G, xp, idx) - self._res(G, xm, idx))
        return [J]


class SpatialObs(O.EdgeGroup):
    """SpatialRegularizerWithObservation: r = w (flow - (T_wc_next x_next - T_wc_cur x_cur)); J = (+w I, -w I) as written"""
    dim = 3

    def __init__(self, a, b, flow, Twc_q, Twc_t, info):
        super().__init__(len(a), info, None)
        self.a, self.b = np.asarray(a, np.int64), np.asarray(b, np.int64)
        self.flow = np.asarray(flow, np.float64).reshape(-1, 3)
        self.q, self.t = Twc_q, Twc_t            # per vertex: world_transform_camera (double, from the float inverse)
        self.slots = [('pt', self.a), ('pt', self.b)]

    def _world(self, G, v):
        q = self.q[v]
        x = G.pts[v]
        uv = 2.0 * np.cross(q[:, :3], x)
        return x + q[:, 3:4] * uv + np.cross(q[:, :3], uv) + self.t[v]

    def residual(self, G, idx):
        return self.flow[idx] - (self._world(G, self.b[idx]) - self._world(G, self.a[idx]))

    def jacobians(self, G, idx):
        I = np.tile(np.eye(3)[None], (len(idx), 1, 1))
        return [I, -I]


# ------------------------------------------------------------------------------------------------ temporal buffer (flat)
def closest_map_points(tb, cand, num_neighbors=10, min_d=20, max_d=500):
    """GetClosestMapPointsToFeature (temporal_buffer.cc:97-141) on the last snapshot: None = "too close" """
    last = tb["n_frames"] - 1
    kp = tb["kp_xy"][last, cand]
    out = []
    for j in np.where(tb["has_kp"][last] & (tb["status"] == 0))[0]:
        if j == cand:
            continue
        dx, dy = np.float64(kp[0] - tb["kp_xy"][last, j, 0]), np.float64(kp[1] - tb["kp_xy"][last, j, 1])     # cv::norm(Point2f): double sqrt
        d = F32(np.sqrt(dx * dx + dy * dy))
        if d > max_d:
            continue
        if d < min_d:
            return None
        out.append((d, int(j)))
    out.sort()
    return [j for _, j in out[:num_neighbors + 1]]                 # `size() > num_neighbors` breaks: 11 entries


def deformable_triangulation(tb, cand, model, prm, min_track=5, trace=None):
    """returns (status, xyz float32[3]); tb: dict(n_frames, poses[F,7] SE3f camera_transform_world (qx qy qz qw tx ty tz),
    has_kp[F,n], kp_xy[F,n,2], has_lm[F,n], lm_xyz[F,n,3], status[n] of the last snapshot)"""
    nb = closest_map_points(tb, cand)
    if not nb:
        return E_CLOSE, np.zeros(3, F32)
    frames = [f for f in range(tb["n_frames"]) if tb["has_kp"][f, cand]]
    if len(frames) < min_track:                                    # Mapping::LandmarkTriangulation (mapping.cc:88-110): TrackLenght >= 5
        return E_SHORT, np.zeros(3, F32)
    first, lastf = frames[0], frames[-1]
    P = tb["poses"].astype(F32)
    kpc, kpp = tb["kp_xy"][first, cand], tb["kp_xy"][lastf, cand]
    cur_ray, prev_ray = _normalized(unproject_f32(model, prm, *kpc)), _normalized(unproject_f32(model, prm, *kpp))
    Tc, Tp = P[first], P[lastf]
    X = triangulate_mid_point(prev_ray, cur_ray, Tp, Tc)
    for T, kp, err in ((Tc, kpc, E_REPROJ1), (Tp, kpp, E_REPROJ2)):
        uv = project_pt(model, prm, se3_mul_point(T, X))
        ex, ey = F32(kp[0]) - uv[0], F32(kp[1]) - uv[1]
        if float(F32(ex * ex + ey * ey)) > 5.991:
            return err, np.zeros(3, F32)
    n1, n2 = (X - se3_inverse(Tc)[4:]).astype(F32), (X - se3_inverse(Tp)[4:]).astype(F32)
    if float(rays_parallax(n1, n2)) < 0.0025 * 5.0:
        return E_PARALLAX, np.zeros(3, F32)
    seeds = []
    for f in frames:
        depth, n = F32(0), 0
        for j in nb:
            if tb["has_lm"][f, j]:
                depth = F32(depth + se3_mul_point(P[f], tb["lm_xyz"][f, j])[2])
                n += 1
        if n == 0:
            return E_NO_NEIGHBOUR, np.zeros(3, F32)
        depth = F32(depth / F32(n))
        if depth < 0:
            return E_NEG_DEPTH, np.zeros(3, F32)
        seeds.append((unproject_f32(model, prm, *tb["kp_xy"][f, cand]) * depth).astype(F32).astype(np.float64))
    V = len(frames)
    ea, eb, flow = [], [], []
    for ia in range(V):
        for ib in range(ia + 1, V):
            for j in nb:
                if tb["has_lm"][frames[ia], j] and tb["has_lm"][frames[ib], j] and tb["has_lm"][first, j]:
                    ea.append(ia)
                    eb.append(ib)
                    flow.append((tb["lm_xyz"][frames[ib], j] - tb["lm_xyz"][frames[ia], j]).astype(F32))
    if V == 0:
        return E_EMPTY, np.zeros(3, F32)
    Twc = np.array([se3_inverse(P[f]) for f in frames]).astype(np.float64)      # SE3Quat(inverse().unit_quaternion().cast<double>(), ...)
    Twc_q = np.array([O.quat_normalize(q) for q in Twc[:, :4]])                  # SE3Quat constructor normalises (se3quat.h:56-58)
    G = O.Graph(model, prm, np.zeros((0, 4)), np.zeros((0, 3)), np.array(seeds))
    G.groups.append(ReprojNumeric([tb["kp_xy"][f, cand] for f in frames], 1.0 / (0.5 * 0.5)))
    reg = SpatialObs(ea, eb, np.array(flow).reshape(-1, 3), Twc_q, Twc[:, 4:], float(F32(1.0) / (F32(0.1) * F32(0.1))))
    G.groups.append(reg)
    G.initialize(0)
    O.lm_optimize(G, 10, trace)
    if reg.n:
        r = reg.residual(G, np.arange(reg.n))
        bad = int(np.sum(reg.info * np.sum(r * r, axis=1) > float(F32(7.815))))
        if F32(bad) / F32(reg.n) > F32(0.5):
            return E_BAD_NEIGHBOURS, np.zeros(3, F32)
    # (regularization_terms.size() == 0: 0 / 0 = NaN > 0.5 is false)
    rr = G.groups[0].residual(G, np.arange(V))
    nbad = int(np.sum(G.groups[0].info * np.sum(rr * rr, axis=1) > 5.99 * 10))
    if F32(nbad) / F32(V) > F32(0.5):
        return E_BAD_ERROR, np.zeros(3, F32)
    depth = F32(G.pts[V - 1, 2])
    un = unproject_f32(model, prm, *tb["kp_xy"][lastf, cand])
    un = (un / un[2]).astype(F32)
    return OK, se3_mul_point(se3_inverse(P[lastf]), (un * depth).astype(F32))
