"""CPU oracle for the NR-SLAM optimisation hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy/SciPy *restatement* of the reference algorithm, written from
the reference sources (cited per function as file:line under /root/reference).
It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it.  The product (nr-slam_amd/) never
does.

Parity pinning: the reference cannot be compiled in this image (Eigen3 and
OpenCV are absent), so the only reference-owned vectors that pin
this oracle are g2o's 36x36 known-answer linear system
(third_party/g2o/unit_test/solver/sparse_system_helper.cpp:52-149,255,298) and
the Huber-derivative property (unit_test/general/robust_kernel_tests.cpp:104-119);
both are checked in tests/test_oracle_pins.py.  Residuals, Jacobians, LM iterates
and neighbour selection are "parity unpinned": the goldens under tests/golden/
are this restatement's own output.

Numeric conventions restated from the reference:
  * solver state, residuals, Hessian: fp64 (g2o number_t = double)
  * projection and projection Jacobian: fp32 (calibration/camera_model.h:89-95,131-137)
  * thresholds / information values are fp32 constants widened to fp64
    (modules/optimization/g2o_optimization.cc:63-64,197-210,958-973)
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

F32 = np.float32

# ----------------------------------------------------------------------------
# constants (g2o_optimization.cc:63-64,195-210,958-973) -- float arithmetic
# ----------------------------------------------------------------------------
TH2_SQ = F32(5.99)
TH2 = F32(np.sqrt(TH2_SQ))            # sqrt(float) -> float
TH3_SQ = F32(0.584)
TH3 = F32(np.sqrt(TH3_SQ))
INFO_REPROJ = F32(1.0) / (F32(0.5) * F32(0.5))
INFO_POSITION = F32(1.0) / (F32(0.1) * F32(0.1))
K_SPRING = float(F32(1.1))
REGULARIZERS_PER_POINT = 10

PINHOLE, KB8 = 0, 1


def info_spatial(scale):
    """g2o_optimization.cc:209-210 / 972-973: sigma = (float)(0.1 * scale)."""
    sigma = F32(0.1 * float(F32(scale)))
    return float(F32(1.0) / (sigma * sigma))


# ----------------------------------------------------------------------------
# camera models, fp32 (calibration/pin_hole.cc:27-49, kannala_brandt_8.cc:34-51,87-116)
# ----------------------------------------------------------------------------
def project_f32(model, prm, p):
    """p: (n,3) float32 -> (n,2) float32."""
    prm = np.asarray(prm, F32)
    p = np.asarray(p, F32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    fx, fy, cx, cy = prm[0], prm[1], prm[2], prm[3]
    if model == PINHOLE:
        u = fx * x / z + cx
        v = fy * y / z + cy
    else:
        k0, k1, k2, k3 = prm[4], prm[5], prm[6], prm[7]
        r2 = x * x + y * y
        # atan2f / cosf / sinf are defined as the double function rounded to float (the value a
        # correctly rounded float routine returns); same definition on the device
        th = np.arctan2(np.sqrt(r2).astype(np.float64), z.astype(np.float64)).astype(F32)
        psi = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(F32)
        th2 = th * th
        th3 = th * th2
        th5 = th3 * th2
        th7 = th5 * th2
        th9 = th7 * th2
        r = th + k0 * th3 + k1 * th5 + k2 * th7 + k3 * th9
        u = fx * r * np.cos(psi.astype(np.float64)).astype(F32) + cx
        v = fy * r * np.sin(psi.astype(np.float64)).astype(F32) + cy
    return np.stack([u, v], axis=1).astype(F32)


def projection_jacobian_f32(model, prm, p):
    """p: (n,3) float32 -> (n,2,3) float32."""
    prm = np.asarray(prm, F32)
    p = np.asarray(p, F32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    fx, fy = prm[0], prm[1]
    J = np.zeros((p.shape[0], 2, 3), F32)
    if model == PINHOLE:
        J[:, 0, 0] = fx / z
        J[:, 0, 2] = -fx * x / (z * z)
        J[:, 1, 1] = fy / z
        J[:, 1, 2] = -fy * y / (z * z)
    else:
        k0, k1, k2, k3 = prm[4], prm[5], prm[6], prm[7]
        x2, y2, z2 = x * x, y * y, z * z
        r2 = x2 + y2
        r = np.sqrt(r2).astype(F32)
        r3 = r2 * r
        th = np.arctan2(r.astype(np.float64), z.astype(np.float64)).astype(F32)
        th2 = th * th
        th3 = th2 * th
        th4 = th2 * th2
        th5 = th4 * th
        th6 = th2 * th4
        th7 = th6 * th
        th8 = th4 * th4
        th9 = th8 * th
        f = th + th3 * k0 + th5 * k1 + th7 * k2 + th9 * k3
        fd = F32(1) + F32(3) * k0 * th2 + F32(5) * k1 * th4 + F32(7) * k2 * th6 + F32(9) * k3 * th8
        J[:, 0, 0] = fx * (fd * z * x2 / (r2 * (r2 + z2)) + f * y2 / r3)
        J[:, 0, 1] = fx * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3)
        J[:, 0, 2] = -fx * fd * x / (r2 + z2)
        J[:, 1, 0] = fy * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3)
        J[:, 1, 1] = fy * (fd * z * y2 / (r2 * (r2 + z2)) + f * x2 / r3)
        J[:, 1, 2] = -fy * fd * y / (r2 + z2)
    return J


# ----------------------------------------------------------------------------
# SE(3) as (unit quaternion xyzw, translation), fp64
# (third_party/g2o/g2o/types/slam3d/se3quat.h:96-102,201-229,250-255;
#  Eigen Quaternion semantics restated from their documented formulas)
# ----------------------------------------------------------------------------
def quat_normalize(q):
    q = np.array(q, np.float64)
    if q[3] < 0:
        q = -q
    return q / np.sqrt(np.dot(q, q))


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz])


def quat_to_R(q):
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def R_to_quat(R):
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def quat_rotate(q, v):
    """Eigen QuaternionBase::_transformVector: v + w*2(qv x v) + qv x 2(qv x v). v: (...,3)."""
    qv = q[:3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[3] * uv + np.cross(qv, uv)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def se3_exp(upd):
    """SE3Quat::exp, se3quat.h:201-229. upd = [omega(3), upsilon(3)]."""
    omega = np.asarray(upd[:3], np.float64)
    ups = np.asarray(upd[3:6], np.float64)
    theta = np.sqrt(np.dot(omega, omega))
    Om = skew(omega)
    Om2 = Om @ Om
    I = np.eye(3)
    if theta < 0.00001:
        R = I + Om + 0.5 * Om2
        V = I + 0.5 * Om + (1.0 / 6.0) * Om2
    else:
        R = I + np.sin(theta) / theta * Om + (1 - np.cos(theta)) / (theta * theta) * Om2
        V = I + (1 - np.cos(theta)) / (theta * theta) * Om + (theta - np.sin(theta)) / (theta ** 3) * Om2
    return quat_normalize(R_to_quat(R)), V @ ups


def se3_mul(qa, ta, qb, tb):
    """SE3Quat::operator*, se3quat.h:96-102."""
    t = ta + quat_rotate(qa, tb)
    q = quat_normalize(quat_mul(qa, qb))
    return q, t


def pose_oplus(q, t, upd):
    """VertexSE3Expmap::oplusImpl, types/sba/vertex_se3_expmap.cpp:48-51."""
    qe, te = se3_exp(upd)
    return se3_mul(qe, te, q, t)


# ----------------------------------------------------------------------------
# Huber (third_party/g2o/g2o/core/robust_kernel_impl.cpp:60-74)
# ----------------------------------------------------------------------------
def huber(e, delta):
    """returns rho(e), rho'(e) element-wise; delta None -> no kernel."""
    e = np.asarray(e, np.float64)
    if delta is None:
        return e.copy(), np.ones_like(e)
    d = float(delta)
    dsqr = d * d
    inl = e <= dsqr
    sq = np.sqrt(np.where(inl, 1.0, e))
    rho0 = np.where(inl, e, 2 * sq * d - dsqr)
    rho1 = np.where(inl, 1.0, d / sq)
    return rho0, rho1


# ----------------------------------------------------------------------------
# generic g2o-like graph: poses (6 dof) then points (3 dof), vectorised edge groups
# ----------------------------------------------------------------------------
class EdgeGroup:
    """A homogeneous set of edges.  Subclasses define residual() and jacobians().

    slots: list of (kind, index-array) with kind 'pose' or 'pt'.
    """
    dim = 0

    def __init__(self, n, info, delta):
        self.n = n
        self.info = float(info)
        self.delta = None if delta is None else float(delta)
        self.level = np.zeros(n, np.int32)
        self.err = np.zeros((n, self.dim))
        self.active = np.zeros(n, bool)
        self.slots = []

    def chi2(self):
        """BaseEdge::chi2 = err^T Omega err using the *stored* error."""
        return self.info * np.sum(self.err * self.err, axis=1)


class Graph:
    def __init__(self, cam_model, cam_prm, pose_q, pose_t, pts):
        self.cam_model = cam_model
        self.cam_prm = np.asarray(cam_prm, F32)
        self.pose_q = np.array(pose_q, np.float64).reshape(-1, 4)
        self.pose_t = np.array(pose_t, np.float64).reshape(-1, 3)
        self.pts = np.array(pts, np.float64).reshape(-1, 3)
        self.K = self.pose_q.shape[0]
        self.M = self.pts.shape[0]
        self.pose_fixed = np.zeros(self.K, bool)
        self.pt_fixed = np.zeros(self.M, bool)
        self.groups = []
        self.x = None

    # -- SparseOptimizer::initializeOptimization (sparse_optimizer.cpp:203-285)
    def initialize(self, level=0):
        pose_act = np.zeros(self.K, bool)
        pt_act = np.zeros(self.M, bool)
        for g in self.groups:
            lev_ok = (g.level == level) if level >= 0 else np.ones(g.n, bool)
            all_fixed = np.ones(g.n, bool)
            for kind, idx in g.slots:
                fx = self.pose_fixed[idx] if kind == 'pose' else self.pt_fixed[idx]
                all_fixed &= fx
            g.active = lev_ok & ~all_fixed
            for kind, idx in g.slots:
                if kind == 'pose':
                    pose_act[idx[g.active]] = True
                else:
                    pt_act[idx[g.active]] = True
        # buildIndexMapping: non-fixed active vertices in id order (poses first)
        self.pose_off = -np.ones(self.K, np.int64)
        self.pt_off = -np.ones(self.M, np.int64)
        pa = np.where(pose_act & ~self.pose_fixed)[0]
        self.pose_off[pa] = 6 * np.arange(len(pa))
        qa = np.where(pt_act & ~self.pt_fixed)[0]
        self.pt_off[qa] = 6 * len(pa) + 3 * np.arange(len(qa))
        self.ndim = 6 * len(pa) + 3 * len(qa)
        self.act_poses, self.act_pts = pa, qa
        self.x = np.zeros(self.ndim)
        return self.ndim > 0

    def compute_active_errors(self):
        for g in self.groups:
            idx = np.where(g.active)[0]
            if len(idx):
                g.err[idx] = g.residual(self, idx)

    def active_robust_chi2(self):
        chi = 0.0
        for g in self.groups:
            idx = np.where(g.active)[0]
            if len(idx):
                e = g.info * np.sum(g.err[idx] ** 2, axis=1)
                chi += float(np.sum(huber(e, g.delta)[0]))
        return chi

    # -- BlockSolver::buildSystem (block_solver.hpp:495-562) + constructQuadraticForm
    def build_system(self):
        n = self.ndim
        rows, cols, vals = [], [], []
        b = np.zeros(n)
        for g in self.groups:
            idx = np.where(g.active)[0]
            if not len(idx):
                continue
            r = g.err[idx]
            e = g.info * np.sum(r * r, axis=1)
            _, rho1 = huber(e, g.delta)
            w = rho1 * g.info                                  # robustInformation = rho' * Omega
            Js = g.jacobians(self, idx)
            offs = []
            for (kind, vi) in g.slots:
                o = (self.pose_off if kind == 'pose' else self.pt_off)[vi[idx]]
                offs.append(o)
            for a, Ja in enumerate(Js):
                oa = offs[a]
                ma = oa >= 0
                if not ma.any():
                    continue
                da = Ja.shape[2]
                # b_a += J_a^T (-rho' Omega r)
                ba = -np.einsum('nij,ni->nj', Ja, r) * w[:, None]
                np.add.at(b, (oa[ma, None] + np.arange(da)[None, :]).ravel(), ba[ma].ravel())
                for c, Jc in enumerate(Js):
                    oc = offs[c]
                    m = ma & (oc >= 0)
                    if not m.any():
                        continue
                    dc = Jc.shape[2]
                    blk = np.einsum('nki,nkj->nij', Ja[m], Jc[m]) * w[m, None, None]
                    rr = oa[m, None, None] + np.arange(da)[None, :, None] + np.zeros((1, 1, dc), np.int64)
                    cc = oc[m, None, None] + np.arange(dc)[None, None, :] + np.zeros((1, da, 1), np.int64)
                    rows.append(rr.ravel())
                    cols.append(cc.ravel())
                    vals.append(blk.ravel())
        if rows:
            H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                              shape=(n, n)).tocsc()
        else:
            H = sp.csc_matrix((n, n))
        return H, b

    # -- vertex push/pop/update (sparse_optimizer.cpp:457-470)
    def push(self):
        self._bak = (self.pose_q.copy(), self.pose_t.copy(), self.pts.copy())

    def pop(self):
        self.pose_q, self.pose_t, self.pts = self._bak

    def update(self, x):
        for k in self.act_poses:
            o = self.pose_off[k]
            self.pose_q[k], self.pose_t[k] = pose_oplus(self.pose_q[k], self.pose_t[k], x[o:o + 6])
        if len(self.act_pts):
            o = self.pt_off[self.act_pts]
            self.pts[self.act_pts] += x[o[:, None] + np.arange(3)[None, :]]


def solve_spd(A, b, x_prev):
    """LinearSolverEigen/Dense::solve: SPD solve; (False, stale x) when not positive definite
    (linear_solver_eigen.h:92-136, linear_solver_dense.h:56-104)."""
    n = A.shape[0]
    if n <= 4500:
        Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
        try:
            L = np.linalg.cholesky(Ad)
        except np.linalg.LinAlgError:
            return False, x_prev
        y = np.linalg.solve(L, b)      # small systems: plain dense triangular solves
        return True, np.linalg.solve(L.T, y)
    lu = spla.splu(sp.csc_matrix(A), permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0,
                   options=dict(SymmetricMode=True))
    if not (np.all(lu.U.diagonal() > 0) and np.array_equal(lu.perm_r, lu.perm_c)):
        return False, x_prev
    return True, lu.solve(b)


def lm_optimize(G, iterations, trace=None, solver=solve_spd):
    """SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
    (sparse_optimizer.cpp:392-455, optimization_algorithm_levenberg.cpp:57-174)."""
    lam, ni = -1.0, 2.0
    done_iters = 0
    if G.ndim == 0:
        return -1
    x = np.zeros(G.ndim)
    for it in range(iterations):
        G.compute_active_errors()
        chi = G.active_robust_chi2()
        H, b = G.build_system()
        if it == 0:
            lam = 1e-5 * float(np.max(np.abs(H.diagonal()))) if G.ndim else 0.0
            ni = 2.0
        rho, qmax = 0.0, 0
        while True:
            G.push()
            ok, x = solver(H + lam * sp.identity(G.ndim, format='csc'), b, x)
            G.update(x)
            G.compute_active_errors()
            temp = G.active_robust_chi2() if ok else np.finfo(np.float64).max
            scale = float(np.dot(x, lam * x + b)) + 1e-3
            rho = (chi - temp) / scale
            accepted = bool(rho > 0 and np.isfinite(temp))
            if trace is not None:
                trace.append(dict(iter=it, trial=qmax, lam=lam, chi=chi, chi_new=temp, rho=rho,
                                  accepted=accepted, ok=ok))
            if accepted:
                alpha = 1.0 - (2 * rho - 1) ** 3
                alpha = min(alpha, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                chi = temp
            else:
                lam *= ni
                ni *= 2
                G.pop()
                if not np.isfinite(lam):
                    break
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        done_iters += 1
        if qmax == 10 or rho == 0 or not np.isfinite(lam):
            break
    return done_iters


# ----------------------------------------------------------------------------
# edge types (modules/optimization/*.cc)
# ----------------------------------------------------------------------------
def _reproj_core(G, q, t, xw):
    """p = R x + t (fp64); fp32 projection + Jacobian. q,t: (n,4),(n,3) or single."""
    if q.ndim == 1:
        p = quat_rotate(q, xw) + t
    else:
        qv = q[:, :3]
        uv = 2.0 * np.cross(qv, xw)
        p = xw + q[:, 3:4] * uv + np.cross(qv, uv) + t
    return p


def _expmap_jac(p):
    """[-[p]x | I] as written in reprojection_error.cc:57-60 (rotation cols 0-2)."""
    n = p.shape[0]
    E = np.zeros((n, 3, 6))
    E[:, 0, 1] = p[:, 2]
    E[:, 0, 2] = -p[:, 1]
    E[:, 1, 0] = -p[:, 2]
    E[:, 1, 2] = p[:, 0]
    E[:, 2, 0] = p[:, 1]
    E[:, 2, 1] = -p[:, 0]
    E[:, 0, 3] = E[:, 1, 4] = E[:, 2, 5] = 1.0
    return E


class ReprojEdges(EdgeGroup):
    """ReprojectionError / ...OnlyPose / ...WithDeformation
    (reprojection_error.cc:32-64, reprojection_error_only_pose.cc:50-75,
     reprojection_error_with_deformation.cc:37-68).

    mode 'pose': point constant X0;  'deform': x = X0 + pts[pt_idx];  'ba': x = pts[pt_idx].
    """
    dim = 2

    def __init__(self, mode, uv, pose_idx, pt_idx, X0, info, delta):
        n = len(uv)
        super().__init__(n, info, delta)
        self.mode = mode
        self.uv = np.asarray(uv, np.float64).reshape(n, 2)
        self.pose_idx = np.asarray(pose_idx, np.int64)
        self.pt_idx = None if pt_idx is None else np.asarray(pt_idx, np.int64)
        self.X0 = None if X0 is None else np.asarray(X0, np.float64).reshape(n, 3)
        self.slots = [('pose', self.pose_idx)]
        if mode != 'pose':
            self.slots.append(('pt', self.pt_idx))

    def _world(self, G, idx):
        if self.mode == 'pose':
            return self.X0[idx]
        if self.mode == 'deform':
            return G.pts[self.pt_idx[idx]] + self.X0[idx]
        return G.pts[self.pt_idx[idx]]

    def _cam(self, G, idx):
        k = self.pose_idx[idx]
        return _reproj_core(G, G.pose_q[k], G.pose_t[k], self._world(G, idx))

    def residual(self, G, idx):
        p = self._cam(G, idx)
        return self.uv[idx] - project_f32(G.cam_model, G.cam_prm, p.astype(F32)).astype(np.float64)

    def jacobians(self, G, idx):
        p = self._cam(G, idx)
        Jp = -projection_jacobian_f32(G.cam_model, G.cam_prm, p.astype(F32)).astype(np.float64)
        Js = [np.einsum('nij,njk->nik', Jp, _expmap_jac(p))]
        if self.mode != 'pose':
            k = self.pose_idx[idx]
            R = np.stack([quat_to_R(G.pose_q[kk]) for kk in np.unique(k)])
            lut = {kk: i for i, kk in enumerate(np.unique(k))}
            Rn = R[[lut[kk] for kk in k]]
            Js.append(np.einsum('nij,njk->nik', Jp, Rn))
        return Js


class DamperDeformEdges(EdgeGroup):
    """SpatialRegularizerWithDeformation (spatial_regularizer_with_deformation.cc:36-49)."""
    dim = 3

    def __init__(self, i, j, w, info, delta):
        super().__init__(len(i), info, delta)
        self.i, self.j = np.asarray(i, np.int64), np.asarray(j, np.int64)
        self.w = np.asarray(w, np.float64)
        self.slots = [('pt', self.i), ('pt', self.j)]

    def residual(self, G, idx):
        return self.w[idx, None] * (G.pts[self.i[idx]] - G.pts[self.j[idx]])

    def jacobians(self, G, idx):
        I = np.eye(3)[None] * self.w[idx, None, None]
        return [I, -I]


class SpringDeformEdges(EdgeGroup):
    """PositionRegularizerWithDeformation (position_regularizer_with_deformation.cc:31-57)."""
    dim = 1

    def __init__(self, i, j, d0, Xi, Xj, info, delta, k=K_SPRING):
        super().__init__(len(i), info, delta)
        self.i, self.j = np.asarray(i, np.int64), np.asarray(j, np.int64)
        self.d0 = np.asarray(d0, np.float64)
        self.Xi, self.Xj = np.asarray(Xi, np.float64), np.asarray(Xj, np.float64)
        self.k = k
        self.slots = [('pt', self.i), ('pt', self.j)]

    def _v(self, G, idx):
        return (self.Xi[idx] + G.pts[self.i[idx]]) - (self.Xj[idx] + G.pts[self.j[idx]])

    def residual(self, G, idx):
        d = np.linalg.norm(self._v(G, idx), axis=1)
        return (self.k * (d - self.d0[idx]) / self.d0[idx])[:, None]

    def jacobians(self, G, idx):
        v = self._v(G, idx)
        d = np.linalg.norm(v, axis=1)
        a = self.k / (2 * self.d0[idx] * d)
        J = (a[:, None] * (2 * v))[:, None, :]
        return [J, -J]


class SpringBAEdges(EdgeGroup):
    """PositionRegularizer, BA form, Jacobian *as written* (position_regularizer.cc:32-61)."""
    dim = 1

    def __init__(self, i, j, d0, info, k=K_SPRING):
        super().__init__(len(i), info, None)
        self.i, self.j = np.asarray(i, np.int64), np.asarray(j, np.int64)
        self.d0 = np.asarray(d0, np.float64)
        self.k = k
        self.slots = [('pt', self.i), ('pt', self.j)]

    def residual(self, G, idx):
        d = np.linalg.norm(G.pts[self.i[idx]] - G.pts[self.j[idx]], axis=1)
        return (self.k * (d - self.d0[idx]) / self.d0[idx])[:, None]

    def jacobians(self, G, idx):
        v = G.pts[self.i[idx]] - G.pts[self.j[idx]]
        d = np.linalg.norm(v, axis=1)
        c = (self.k / self.d0[idx]) * (1.0 / np.sqrt(d))
        J = (c[:, None] * (2.0 * v))[:, None, :]
        return [J, -J]


class DamperBAEdges(EdgeGroup):
    """SpatialRegularizer, 4 vertices (1c,2c,1n,2n) (spatial_regularizer.cc:32-59)."""
    dim = 3

    def __init__(self, idx4, w, info, delta):
        idx4 = np.asarray(idx4, np.int64).reshape(-1, 4)
        super().__init__(len(idx4), info, delta)
        self.v = idx4
        self.w = np.asarray(w, np.float64)
        self.slots = [('pt', idx4[:, 0]), ('pt', idx4[:, 1]), ('pt', idx4[:, 2]), ('pt', idx4[:, 3])]

    def residual(self, G, idx):
        v = self.v[idx]
        P = G.pts
        return self.w[idx, None] * ((P[v[:, 2]] - P[v[:, 0]]) - (P[v[:, 3]] - P[v[:, 1]]))

    def jacobians(self, G, idx):
        I = np.eye(3)[None] * self.w[idx, None, None]
        return [-I, I, I, -I]


class DamperFixedEdges(EdgeGroup):
    """SpatialRegularizerFixed: unary, other end read live through a raw pointer
    (spatial_regularizer_fixed.cc:32-43, g2o_optimization.cc:512-529)."""
    dim = 3

    def __init__(self, i, j_fixed, w, info, delta):
        super().__init__(len(i), info, delta)
        self.i, self.j = np.asarray(i, np.int64), np.asarray(j_fixed, np.int64)
        self.w = np.asarray(w, np.float64)
        self.slots = [('pt', self.i)]

    def residual(self, G, idx):
        return self.w[idx, None] * (G.pts[self.i[idx]] - G.pts[self.j[idx]])

    def jacobians(self, G, idx):
        return [np.eye(3)[None] * self.w[idx, None, None]]


# ----------------------------------------------------------------------------
# a1: CameraPoseOptimization (g2o_optimization.cc:50-146)
# ----------------------------------------------------------------------------
def pose_only_solve(cam_model, cam_prm, uv, X, pose_q, pose_t, trace=None):
    """uv (n,2) f32, X (n,3) f32, pose (q xyzw, t) as handed over by the boundary.
    Returns pose_q, pose_t (fp64), inlier mask (bool)."""
    uv = np.asarray(uv, F32)
    X = np.asarray(X, F32)
    n = len(uv)
    q0 = quat_normalize(np.asarray(pose_q, np.float64))
    t0 = np.asarray(pose_t, np.float64).copy()
    G = Graph(cam_model, cam_prm, [q0], [t0], np.zeros((0, 3)))
    e = ReprojEdges('pose', uv, np.zeros(n, np.int64), None, X.astype(np.float64), 1.0, TH2)
    G.groups.append(e)
    inl = np.ones(n, bool)
    for rnd in range(3):
        G.pose_q[0], G.pose_t[0] = q0.copy(), t0.copy()
        if G.initialize(0):
            tr = None if trace is None else []
            lm_optimize(G, 10, tr)
            if trace is not None:
                trace.append(tr)
        # :115-140 -- inliers keep the error stored by the last computeActiveErrors
        out = np.where(~inl)[0]
        if len(out):
            e.err[out] = e.residual(G, out)
        chi = e.chi2().astype(F32)
        inl = ~(chi > TH2_SQ)
        e.level[:] = np.where(inl, 0, 1)
        if rnd == 2:
            e.delta = None
    return G.pose_q[0].copy(), G.pose_t[0].copy(), inl


# ----------------------------------------------------------------------------
# a19: RegularizationGraph::GetEdges + caller filters
# (regularization_graph.cc:61-87, geometry_toolbox.cc:26-28, g2o_optimization.cc:255-279)
# ----------------------------------------------------------------------------
GRAPH_BAD = 3          # Status enum VERIFIED,NEIGHBOR,NEUTRAL,BAD (regularization_graph.h:42-47)
GRAPH_NEUTRAL = 2


def interpolation_weight(d, sigma):
    """InterpolationWeight (geometry_toolbox.cc:26-28): float argument, exp evaluated in double and
    rounded to float (what a correctly rounded expf returns; libm expf implementations differ in
    the last ulp, so the build fixes this definition on both sides -- DESIGN.md "weights")."""
    d = np.asarray(d, F32)
    sigma = F32(sigma)
    arg = (-(d * d) / (F32(2) * sigma * sigma)).astype(F32)
    return np.exp(arg.astype(np.float64)).astype(F32)


def min_weight(sigma):
    """regularization_graph.cc:28-36: InterpolationWeight(sigma*1.5 (double->float), sigma)."""
    return interpolation_weight(F32(float(F32(sigma)) * 1.5), sigma)


def get_edges(ids, weight, status, min_w):
    """GetEdges for one vertex: candidates given in ascending-id order.  Sort by (status asc,
    weight desc); ties keep id order (std::sort leaves ties unspecified -- documented choice);
    cut at first weight < min_weight.  Returns positions into the candidate arrays."""
    order = np.lexsort((np.arange(len(ids)), -np.asarray(weight, np.float64), np.asarray(status)))
    out = []
    for p in order:
        if weight[p] < min_w:
            break
        out.append(p)
    return np.array(out, np.int64)


# ----------------------------------------------------------------------------
# a20: UpdateVertex / UpdateConnection (regularization_graph.cc:89-146)
# ----------------------------------------------------------------------------
def graph_update_vertex(pos_i, pos_others, max_d, min_d, status, sigma, stretch_th):
    """fp32 throughout.  Returns new (max_d, min_d, weight, status, good_count)."""
    pi = np.asarray(pos_i, F32)
    po = np.asarray(pos_others, F32)
    d = np.sqrt(np.sum((pi[None, :] - po) ** 2, axis=1, dtype=F32)).astype(F32)
    max_d = np.maximum(np.asarray(max_d, F32), d)
    min_d = np.minimum(np.asarray(min_d, F32), d)
    w = interpolation_weight(max_d, sigma)
    bad = np.abs((max_d - min_d) / min_d) > F32(stretch_th)
    status = np.where(bad, GRAPH_BAD, status)
    return max_d, min_d, w, status, int(np.sum(~bad))


# ----------------------------------------------------------------------------
# a3: LocalDeformableBundleAdjustment (g2o_optimization.cc:880-1161)
# ----------------------------------------------------------------------------
def dba_build(kf_points, nbr_rowptr, nbr_col, nbr_w, nbr_d0, nbr_status):
    """Edge construction of OPT:927-1137 on flattened inputs.

    kf_points: list (oldest -> newest keyframe) of int arrays: map-point index of every
               TRACKED_WITH_3D observation, in keyframe index order.
    nbr_*:     ordered neighbour CSR per map point = the output of GetEdges (a19).
    Returns dict with lm_kf, lm_pt (landmark = one per (kf, point), kf-major),
    springs (i, j, d0) and dampers (1c, 2c, 1n, 2n, w) in reference insertion order.
    """
    K = len(kf_points)
    lm_kf, lm_pt = [], []
    inserted = []                                  # inserted_landmarks[kf][mappoint] = landmark index
    for k, pts in enumerate(kf_points):
        d = {}
        for p in pts:
            d[int(p)] = len(lm_kf)
            lm_kf.append(k)
            lm_pt.append(int(p))
        inserted.append(d)
    sp_i, sp_j, sp_d0 = [], [], []
    dm, dm_w = [], []
    spring_seen, damper_seen = set(), set()
    for k, pts in enumerate(kf_points):
        cur = inserted[k]
        nxt = inserted[k + 1] if k + 1 < K else None
        for p in pts:
            p = int(p)
            l = cur[p]
            lo, hi = nbr_rowptr[p], nbr_rowptr[p + 1]
            n_reg = 0
            for e in range(lo, hi):
                if n_reg > REGULARIZERS_PER_POINT or nbr_status[e] == GRAPH_BAD:
                    break
                o = int(nbr_col[e])
                if o not in cur:
                    continue
                key = (min(p, o), max(p, o), k)
                if key in spring_seen:
                    n_reg += 1
                    continue
                spring_seen.add(key)
                sp_i.append(l)
                sp_j.append(cur[o])
                sp_d0.append(nbr_d0[e])
                n_reg += 1
            if nxt is not None:
                if p not in nxt:
                    continue
                ln = nxt[p]
                n_reg = 0
                for e in range(lo, hi):
                    if n_reg > REGULARIZERS_PER_POINT or nbr_status[e] == GRAPH_BAD:
                        break
                    o = int(nbr_col[e])
                    if o not in cur or o not in nxt:
                        continue
                    key = (min(p, o), max(p, o), k)
                    if key in damper_seen:
                        n_reg += 1
                        continue
                    damper_seen.add(key)
                    dm.append((l, cur[o], ln, nxt[o]))
                    dm_w.append(nbr_w[e])
                    n_reg += 1
    return dict(lm_kf=np.array(lm_kf, np.int32), lm_pt=np.array(lm_pt, np.int32),
                sp_ij=np.array(list(zip(sp_i, sp_j)), np.int32).reshape(-1, 2),
                sp_d0=np.array(sp_d0, F32),
                dm_idx=np.array(dm, np.int32).reshape(-1, 4), dm_w=np.array(dm_w, F32))


def dba_graph(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0,
              dm_idx, dm_w, scale):
    """Assemble the BA graph on flat arrays (OPT:1006-1136 edge parameters)."""
    pq = np.array([quat_normalize(q) for q in np.asarray(poses_q, np.float64)])
    G = Graph(cam_model, cam_prm, pq, np.asarray(poses_t, np.float64),
              np.asarray(lm_xyz, F32).astype(np.float64))
    n = len(lm_kf)
    G.groups.append(ReprojEdges('ba', np.asarray(lm_uv, F32), lm_kf, np.arange(n), None,
                                float(INFO_REPROJ), TH2))
    sp_ij = np.asarray(sp_ij).reshape(-1, 2)
    G.groups.append(SpringBAEdges(sp_ij[:, 0], sp_ij[:, 1], np.asarray(sp_d0, F32).astype(np.float64),
                                  float(INFO_POSITION)))
    G.groups.append(DamperBAEdges(dm_idx, np.asarray(dm_w, F32).astype(np.float64),
                                  info_spatial(scale), TH3))
    return G


def dba_solve(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0,
              dm_idx, dm_w, scale, iters=5, trace=None, solver=solve_spd):
    """optimize(5) on the BA graph (OPT:1141-1143).  Returns poses (q,t fp64), landmarks fp64."""
    G = dba_graph(cam_model, cam_prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0,
                  dm_idx, dm_w, scale)
    G.initialize(0)
    n_it = lm_optimize(G, iters, trace, solver)
    return G.pose_q.copy(), G.pose_t.copy(), G.pts.copy(), n_it


# ----------------------------------------------------------------------------
# flat RegularizationGraph helpers (regularization_graph.cc:61-146)
# ----------------------------------------------------------------------------
def graph_get_edges(g, p):
    """GetEdges(p) on the flat graph: list of (other, eid) in reference order."""
    sl = slice(g["rowptr"][p], g["rowptr"][p + 1])
    e = g["eid"][sl]
    pos = get_edges(g["col"][sl], g["e_w"][e], g["e_status"][e], F32(g["min_w"]))
    return [(int(g["col"][sl][k]), int(e[k])) for k in pos]


def graph_update_vertex_flat(g, p, map_pos):
    """UpdateVertex(p): every edge of p is refreshed from the last world positions
    (regularization_graph.cc:89-146).  Returns the number of good connections."""
    sl = slice(g["rowptr"][p], g["rowptr"][p + 1])
    e = g["eid"][sl]
    if len(e) == 0:
        return 0
    mx, mn, w, st, good = graph_update_vertex(map_pos[p], map_pos[g["col"][sl]], g["e_max"][e], g["e_min"][e],
                                              g["e_status"][e], g["sigma"], g["stretch_th"])
    g["e_max"][e], g["e_min"][e], g["e_w"][e], g["e_status"][e] = mx, mn, w, st
    return good


# ----------------------------------------------------------------------------
# a2: CameraPoseAndDeformationOptimization (g2o_optimization.cc:148-557)
# ----------------------------------------------------------------------------
TRACKED_WITH_3D, TRACKED, JUST_TRIANGULATED, BAD = 0, 1, 2, 3


def track_deform_solve(cam_model, cam_prm, graph, map_pos, f_map, f_status, f_uv, f_pos, pose_q, pose_t,
                       scale, trace=None, solver=solve_spd):
    """Flat restatement.  graph / map_pos / f_status / f_pos are copied, the updated copies are
    returned.  f_map[i] = map-point index of frame landmark i (-1: none)."""
    if isinstance(graph, dict):
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}

        def get_edges_of(p):                       # GetEdges: (other, weight, first_distance, status) in the reference's order
            return [(o, g["e_w"][e], g["e_d0"][e], g["e_status"][e]) for o, e in graph_get_edges(g, p)]

        def update_vertex_of(p, pos):
            return graph_update_vertex_flat(g, p, pos)
    else:
        # a dense all-pairs graph object (oracle/rgraph_oracle.DenseGraph): updated IN PLACE, the caller passes a copy
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
    n_map = len(map_pos)
    map_to_frame = -np.ones(n_map, np.int64)
    map_to_frame[f_map[f_map >= 0]] = np.where(f_map >= 0)[0]
    opt_f = np.where((f_status == TRACKED_WITH_3D) & (f_map >= 0))[0]       # frame order (frame.cc:83-119)
    N = len(opt_f)
    ids = f_map[opt_f]
    id_to_idx = -np.ones(n_map, np.int64)
    id_to_idx[ids] = np.arange(N)
    X0 = f_pos[opt_f].astype(np.float64)
    q0 = quat_normalize(np.asarray(pose_q, np.float64))
    t0 = np.asarray(pose_t, np.float64).copy()
    info_sp = info_spatial(scale)

    # ---- edge construction OPT:224-337
    reg = [dict() for _ in range(N)]               # spatial_regularizers[idx][idx_other] = edge index
    dm_i, dm_j, dm_w, sp_d0 = [], [], [], []
    lost = set()
    for idx in range(N):
        n_reg = 0
        for other, e_w, e_d0, e_st in get_edges_of(int(ids[idx])):
            if n_reg > REGULARIZERS_PER_POINT or e_st == GRAPH_BAD:
                break
            fo = map_to_frame[other]
            if fo < 0 or f_status[fo] != TRACKED_WITH_3D:
                if fo >= 0 and f_status[fo] != JUST_TRIANGULATED:
                    lost.add(other)
                continue
            io = int(id_to_idx[other])
            if io in reg[idx]:
                continue
            k = len(dm_i)
            dm_i.append(idx); dm_j.append(io); dm_w.append(e_w); sp_d0.append(e_d0)
            reg[idx][io] = k
            reg[io][idx] = k
            n_reg += 1
    E = len(dm_i)
    G = Graph(cam_model, cam_prm, [q0], [t0], np.zeros((N, 3)))
    rep = ReprojEdges('deform', f_uv[opt_f], np.zeros(N, np.int64), np.arange(N), X0, float(INFO_REPROJ), TH2)
    dmp = DamperDeformEdges(dm_i, dm_j, np.asarray(dm_w, F32).astype(np.float64), info_sp, TH3)
    spr = SpringDeformEdges(dm_i, dm_j, np.asarray(sp_d0, F32).astype(np.float64),
                            X0[np.asarray(dm_i, np.int64)] if E else np.zeros((0, 3)),
                            X0[np.asarray(dm_j, np.int64)] if E else np.zeros((0, 3)),
                            float(INFO_POSITION), TH3)
    G.groups += [rep, dmp, spr]
    inl = np.ones(N, bool)
    for rnd in range(2):                                                   # OPT:338-395
        G.pose_q[0], G.pose_t[0] = q0.copy(), t0.copy()
        G.pts[:] = 0
        if G.initialize(0):
            tr = None if trace is None else []
            lm_optimize(G, 10, tr, solver)
            if trace is not None:
                trace.append(tr)
        rep.err[:] = rep.residual(G, np.arange(N))
        chi = rep.chi2().astype(F32)
        for idx in range(N):
            out = bool(chi[idx] > TH2_SQ)
            inl[idx] = not out
            rep.level[idx] = 1 if out else 0
            for io, k in reg[idx].items():
                dmp.level[k] = 1 if out else 0
            for io, k in reg[idx].items():
                dmp.err[k] = dmp.residual(G, np.array([k]))[0]
                dmp.level[k] = 1 if dmp.chi2()[k] > float(TH3_SQ) else 0
    pose_q_out, pose_t_out = G.pose_q[0].copy(), G.pose_t[0].copy()
    # ---- OPT:401-455
    delta = G.pts[:N].astype(F32)
    mag = np.sqrt((delta[:, 0] * delta[:, 0] + delta[:, 1] * delta[:, 1] + delta[:, 2] * delta[:, 2]).astype(F32)).astype(F32)
    srt = np.sort(mag)
    q1 = srt[int(F32(N) * F32(0.25))]
    q3 = srt[int(F32(N) * F32(0.75))]
    th = F32(1.5) * (q3 - q1)
    rep.err[:] = rep.residual(G, np.arange(N))
    chi = rep.chi2().astype(F32)
    for idx in range(N):
        fi = opt_f[idx]
        if chi[idx] > TH2_SQ:
            inl[idx] = False
            f_status[fi] = TRACKED
        if mag[idx] >= q3 + th:
            f_status[fi] = TRACKED
            continue
        G.pt_fixed[idx] = True
        cur = delta[idx] + f_pos[fi]
        f_pos[fi] = cur
        map_pos[ids[idx]] = cur
    median = float(np.partition(mag.copy(), N // 2)[N // 2])
    # ---- graph update OPT:457-474
    for idx in range(N):
        if not inl[idx]:
            continue
        good = update_vertex_of(int(ids[idx]), map_pos)
        if good < REGULARIZERS_PER_POINT * 0.5:
            f_status[opt_f[idx]] = BAD
    res = dict(pose_q=pose_q_out, pose_t=pose_t_out, f_pos=f_pos, f_status=f_status, map_pos=map_pos, graph=g,
               median=median, inliers=inl, delta=G.pts[:N].copy(), lost=[], n_edges=E)
    if not lost:
        return res
    # ---- stage 2 OPT:476-553
    lost_sorted = sorted(lost)
    L = len(lost_sorted)
    G.pts = np.vstack([G.pts, np.zeros((L, 3))])
    G.M = N + L
    G.pt_fixed = np.concatenate([G.pt_fixed, np.zeros(L, bool)])
    ui, uj, uw = [], [], []
    for li, lid in enumerate(lost_sorted):
        n_reg = 0
        for other, e_w, e_d0, e_st in get_edges_of(lid):
            if n_reg > 10:
                break
            if id_to_idx[other] < 0:
                continue
            ui.append(N + li); uj.append(int(id_to_idx[other])); uw.append(e_w)
            n_reg += 1
    G.groups.append(DamperFixedEdges(ui, uj, np.asarray(uw, F32).astype(np.float64), info_sp, TH3))
    G.pose_fixed[0] = True
    if G.initialize(0):
        tr = None if trace is None else []
        lm_optimize(G, 10, tr, solver)
        if trace is not None:
            trace.append(tr)
    for li, lid in enumerate(lost_sorted):
        map_pos[lid] = G.pts[N + li].astype(F32) + map_pos[lid]
    res.update(lost=lost_sorted, lost_delta=G.pts[N:].copy(), delta_final=G.pts[:N].copy(), map_pos=map_pos)
    return res
