"""ctypes binding of oracle/_build/libnrs_cpu.so (oracle/nrs_cpu.cpp) -- TEST INFRASTRUCTURE ONLY: the C++ CPU
restatement of the deformable BA, used by tests/ and by bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Stats(C.Structure):
    _fields_ = [("t_total", C.c_double), ("t_linearize", C.c_double), ("t_analyze", C.c_double), ("t_factor", C.c_double),
                ("t_solve", C.c_double), ("t_errors", C.c_double), ("t_structure", C.c_double), ("chol_flops", C.c_double),
                ("chol_blocks", C.c_int64), ("h_blocks", C.c_int64), ("n_factor", C.c_int32), ("n_pcg_iters", C.c_int32),
                ("n_trials", C.c_int32), ("n_iters", C.c_int32), ("threads", C.c_int32), ("unknowns", C.c_int32)]


class Trial(C.Structure):
    _fields_ = [("iter", C.c_int32), ("trial", C.c_int32), ("accepted", C.c_int32), ("ok", C.c_int32), ("inner", C.c_int32),
                ("lam", C.c_double), ("chi", C.c_double), ("chi_new", C.c_double), ("rho", C.c_double)]


def build(native=False):
    """make -C oracle; native=True: a -march=native copy for timing on THIS host (oracle/_build/native)."""
    args = ["make", "-s", "-C", HERE]
    if native:
        args += ["ARCH=native", "OUT=_build/native"]
    subprocess.check_call(args)
    return os.path.join(HERE, "_build", "native" if native else "", "libnrs_cpu.so")


def load(native=False):
    path = os.path.join(HERE, "_build", "native" if native else "", "libnrs_cpu.so")
    if not os.path.exists(path):
        path = build(native)
    return C.CDLL(path)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def block_cholesky_solve(nb, br, bc, bv, rhs, lam=0.0, ordering=1, lib=None):
    lib = lib or load()
    br, bc = np.ascontiguousarray(br, np.int32), np.ascontiguousarray(bc, np.int32)
    bv, rhs = np.ascontiguousarray(bv, np.float64), np.ascontiguousarray(rhs, np.float64)
    x = np.zeros(3 * nb)
    rc = lib.nrs_cpu_block_cholesky_solve(C.c_int32(nb), C.c_int32(len(br)), _p(br, C.c_int32), _p(bc, C.c_int32), _p(bv, C.c_double),
                                          _p(rhs, C.c_double), C.c_double(lam), C.c_int32(ordering), _p(x, C.c_double))
    return rc == 0, x


def dba_solve(model, prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0, dm_idx, dm_w, scale, iters=5,
              solver=0, pcg_rtol=1e-10, threads=1, max_trials=0, lib=None):
    """returns (poses_q, poses_t, landmarks fp64, trace list, stats dict); solver 0 = sparse Cholesky, 1 = PCG"""
    lib = lib or load()
    p8 = np.zeros(8, np.float32)
    p8[:len(prm)] = np.asarray(prm, np.float32)
    qt = np.ascontiguousarray(np.concatenate([np.asarray(poses_q, np.float64), np.asarray(poses_t, np.float64)], 1))
    xyz = np.ascontiguousarray(lm_xyz, np.float32).copy()
    kf, uv = np.ascontiguousarray(lm_kf, np.int32), np.ascontiguousarray(lm_uv, np.float32)
    sp, d0 = np.ascontiguousarray(sp_ij, np.int32).reshape(-1, 2), np.ascontiguousarray(sp_d0, np.float32)
    dm, dw = np.ascontiguousarray(dm_idx, np.int32).reshape(-1, 4), np.ascontiguousarray(dm_w, np.float32)
    tr = (Trial * 256)()
    ntr = C.c_int32(0)
    x64 = np.zeros((len(xyz), 3), np.float64)
    st = Stats()
    rc = lib.nrs_cpu_dba_solve(C.c_int32(int(model)), _p(p8, C.c_float), C.c_int32(len(qt)), _p(qt, C.c_double), C.c_int32(len(xyz)),
                               _p(xyz, C.c_float), _p(kf, C.c_int32), _p(uv, C.c_float), C.c_int32(len(sp)), _p(sp, C.c_int32),
                               _p(d0, C.c_float), C.c_int32(len(dm)), _p(dm, C.c_int32), _p(dw, C.c_float), C.c_float(scale),
                               C.c_int32(iters), C.c_int32(solver), C.c_double(pcg_rtol), C.c_int32(threads), C.c_int32(max_trials),
                               tr, C.c_int32(256), C.byref(ntr), _p(x64, C.c_double), C.byref(st))
    if rc != 0:
        raise RuntimeError("nrs_cpu_dba_solve failed: %d" % rc)
    trace = [dict(iter=t.iter, trial=t.trial, accepted=bool(t.accepted), ok=bool(t.ok), inner=t.inner, lam=t.lam, chi=t.chi,
                  chi_new=t.chi_new, rho=t.rho) for t in tr[:min(ntr.value, 256)]]
    stats = {k: getattr(st, k) for k, _ in Stats._fields_}
    return qt[:, :4].copy(), qt[:, 4:].copy(), x64, trace, stats


def dba_solve_embedded(model, prm, poses_q, poses_t, lm_xyz, lm_kf, lm_uv, sp_ij, sp_d0, dm_idx, dm_w, sk_kf, sk_uv, sk_xyz, sk_node, sk_omega,
                       scale, iters=5, solver=0, pcg_rtol=1e-10, threads=1, max_trials=0, lib=None):
    """the embedded form of the window (oracle/embedded_oracle.py dba_solve_embedded): returns (poses_q, poses_t, node copies fp64,
    skinned positions fp64, trace list, stats dict)"""
    lib = lib or load()
    p8 = np.zeros(8, np.float32)
    p8[:len(prm)] = np.asarray(prm, np.float32)
    qt = np.ascontiguousarray(np.concatenate([np.asarray(poses_q, np.float64), np.asarray(poses_t, np.float64)], 1))
    xyz = np.ascontiguousarray(lm_xyz, np.float32).copy()
    kf, uv = np.ascontiguousarray(lm_kf, np.int32), np.ascontiguousarray(lm_uv, np.float32)
    sp, d0 = np.ascontiguousarray(sp_ij, np.int32).reshape(-1, 2), np.ascontiguousarray(sp_d0, np.float32)
    dm, dw = np.ascontiguousarray(dm_idx, np.int32).reshape(-1, 4), np.ascontiguousarray(dm_w, np.float32)
    skf, suv, sx = np.ascontiguousarray(sk_kf, np.int32), np.ascontiguousarray(sk_uv, np.float32), np.ascontiguousarray(sk_xyz, np.float32)
    snode, som = np.ascontiguousarray(sk_node, np.int32).reshape(-1, 11), np.ascontiguousarray(sk_omega, np.float64).reshape(-1, 11)
    tr = (Trial * 256)()
    ntr = C.c_int32(0)
    x64, s64 = np.zeros((len(xyz), 3), np.float64), np.zeros((len(skf), 3), np.float64)
    st = Stats()
    rc = lib.nrs_cpu_dba_solve_embedded(C.c_int32(int(model)), _p(p8, C.c_float), C.c_int32(len(qt)), _p(qt, C.c_double), C.c_int32(len(xyz)),
                                        _p(xyz, C.c_float), _p(kf, C.c_int32), _p(uv, C.c_float), C.c_int32(len(sp)), _p(sp, C.c_int32),
                                        _p(d0, C.c_float), C.c_int32(len(dm)), _p(dm, C.c_int32), _p(dw, C.c_float),
                                        C.c_int32(len(skf)), _p(skf, C.c_int32), _p(suv, C.c_float), _p(sx, C.c_float), _p(snode, C.c_int32), _p(som, C.c_double),
                                        C.c_float(scale), C.c_int32(iters), C.c_int32(solver), C.c_double(pcg_rtol), C.c_int32(threads), C.c_int32(max_trials),
                                        tr, C.c_int32(256), C.byref(ntr), _p(x64, C.c_double), _p(s64, C.c_double), C.byref(st))
    if rc != 0:
        raise RuntimeError("nrs_cpu_dba_solve_embedded failed: %d" % rc)
    trace = [dict(iter=t.iter, trial=t.trial, accepted=bool(t.accepted), ok=bool(t.ok), inner=t.inner, lam=t.lam, chi=t.chi,
                  chi_new=t.chi_new, rho=t.rho) for t in tr[:min(ntr.value, 256)]]
    return qt[:, :4].copy(), qt[:, 4:].copy(), x64, s64, trace, {k: getattr(st, k) for k, _ in Stats._fields_}


def max_threads(lib=None):
    return (lib or load()).nrs_cpu_max_threads()


# ---- per-frame solves (oracle/nrs_cpu_track.hpp): a1 CameraPoseOptimization, a2 CameraPoseAndDeformationOptimization -------
class TStats(C.Structure):
    _fields_ = [("t_total", C.c_double), ("t_graph", C.c_double), ("t_structure", C.c_double), ("t_factor", C.c_double),
                ("t_solve", C.c_double), ("t_linearize", C.c_double), ("chol_flops", C.c_double), ("n_factor", C.c_int32),
                ("n_trials", C.c_int32), ("n_iters", C.c_int32), ("unknowns_max", C.c_int32)]


def _trace_list(tr, n):
    return [dict(round=t.iter // 100, iter=t.iter % 100, trial=t.trial, accepted=bool(t.accepted), ok=bool(t.ok), lam=t.lam, chi=t.chi,
                 chi_new=t.chi_new, rho=t.rho) for t in tr[:min(n, len(tr))]]


def pose_only_solve(model, prm, uv, X, pose_q, pose_t, lib=None):
    """returns (pose_q, pose_t, inlier mask, trace, stats)"""
    lib = lib or load()
    p8 = np.zeros(8, np.float32)
    p8[:len(prm)] = np.asarray(prm, np.float32)
    uv, X = np.ascontiguousarray(uv, np.float32), np.ascontiguousarray(X, np.float32)
    qt = np.ascontiguousarray(np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)]))
    inl = np.zeros(len(uv), np.uint8)
    tr = (Trial * 512)()
    ntr, st = C.c_int32(0), TStats()
    rc = lib.nrs_cpu_pose_only_solve(C.c_int32(int(model)), _p(p8, C.c_float), C.c_int32(len(uv)), _p(uv, C.c_float), _p(X, C.c_float),
                                     _p(qt, C.c_double), _p(inl, C.c_uint8), tr, C.c_int32(512), C.byref(ntr), C.byref(st))
    if rc != 0:
        raise RuntimeError("nrs_cpu_pose_only_solve failed: %d" % rc)
    return qt[:4].copy(), qt[4:].copy(), inl.astype(bool), _trace_list(tr, ntr.value), {k: getattr(st, k) for k, _ in TStats._fields_}


def track_deform_solve(model, prm, graph, map_pos, f_map, f_status, f_uv, f_pos, pose_q, pose_t, scale, lib=None):
    """the flat-graph form of oracle/nrs_oracle.track_deform_solve; the graph dict is copied, the updated copy returned"""
    lib = lib or load()
    p8 = np.zeros(8, np.float32)
    p8[:len(prm)] = np.asarray(prm, np.float32)
    g = {k: (np.ascontiguousarray(v).copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}
    for k in ("rowptr", "col", "eid", "e_status"):
        g[k] = np.ascontiguousarray(g[k], np.int32)
    for k in ("e_w", "e_d0", "e_max", "e_min"):
        g[k] = np.ascontiguousarray(g[k], np.float32)
    map_pos = np.ascontiguousarray(map_pos, np.float32).copy()
    f_map = np.ascontiguousarray(f_map, np.int32)
    f_status = np.ascontiguousarray(f_status, np.int32).copy()
    f_uv = np.ascontiguousarray(f_uv, np.float32)
    f_pos = np.ascontiguousarray(f_pos, np.float32).copy()
    qt = np.ascontiguousarray(np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)]))
    n_points = len(map_pos)
    lost = np.zeros(n_points, np.int32)
    med, nl = C.c_float(0), C.c_int32(0)
    tr = (Trial * 1024)()
    ntr, st = C.c_int32(0), TStats()
    rc = lib.nrs_cpu_track_deform_solve(C.c_int32(int(model)), _p(p8, C.c_float), C.c_int32(n_points), _p(g["rowptr"], C.c_int32),
                                        _p(g["col"], C.c_int32), _p(g["eid"], C.c_int32), _p(g["e_w"], C.c_float), _p(g["e_d0"], C.c_float),
                                        _p(g["e_max"], C.c_float), _p(g["e_min"], C.c_float), _p(g["e_status"], C.c_int32),
                                        C.c_float(float(g["sigma"])), C.c_float(float(g["stretch_th"])), _p(map_pos, C.c_float),
                                        C.c_int32(len(f_map)), _p(f_map, C.c_int32), _p(f_status, C.c_int32), _p(f_uv, C.c_float),
                                        _p(f_pos, C.c_float), _p(qt, C.c_double), C.c_float(float(scale)), C.byref(med), C.byref(nl), _p(lost, C.c_int32),
                                        tr, C.c_int32(1024), C.byref(ntr), C.byref(st))
    if rc != 0:
        raise RuntimeError("nrs_cpu_track_deform_solve failed: %d" % rc)
    return dict(pose_q=qt[:4].copy(), pose_t=qt[4:].copy(), f_pos=f_pos, f_status=f_status, map_pos=map_pos, graph=g, median=float(med.value),
                lost=lost[:nl.value].tolist(), trace=_trace_list(tr, ntr.value), stats={k: getattr(st, k) for k, _ in TStats._fields_})


def track_deform_solve_embedded(model, prm, graph, map_pos, f_map, f_status, f_uv, f_pos, f_node, pose_q, pose_t, scale, lib=None):
    """the embedded-deformation form on the flat graph (oracle/embedded_oracle.track_deform_solve_embedded): f_node[i] != 0 marks the nodes"""
    lib = lib or load()
    p8 = np.zeros(8, np.float32)
    p8[:len(prm)] = np.asarray(prm, np.float32)
    g = {k: (np.ascontiguousarray(v).copy() if isinstance(v, np.ndarray) else v) for k, v in graph.items()}
    for k in ("rowptr", "col", "eid", "e_status"):
        g[k] = np.ascontiguousarray(g[k], np.int32)
    for k in ("e_w", "e_d0", "e_max", "e_min"):
        g[k] = np.ascontiguousarray(g[k], np.float32)
    map_pos = np.ascontiguousarray(map_pos, np.float32).copy()
    f_map = np.ascontiguousarray(f_map, np.int32)
    f_status = np.ascontiguousarray(f_status, np.int32).copy()
    f_uv = np.ascontiguousarray(f_uv, np.float32)
    f_pos = np.ascontiguousarray(f_pos, np.float32).copy()
    f_node = np.ascontiguousarray(np.asarray(f_node) != 0, np.uint8)
    qt = np.ascontiguousarray(np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)]))
    n_points = len(map_pos)
    lost = np.zeros(n_points, np.int32)
    med, nl, nn, ns = C.c_float(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    tr = (Trial * 1024)()
    ntr, st = C.c_int32(0), TStats()
    rc = lib.nrs_cpu_track_deform_solve_embedded(C.c_int32(int(model)), _p(p8, C.c_float), C.c_int32(n_points), _p(g["rowptr"], C.c_int32),
                                                 _p(g["col"], C.c_int32), _p(g["eid"], C.c_int32), _p(g["e_w"], C.c_float), _p(g["e_d0"], C.c_float),
                                                 _p(g["e_max"], C.c_float), _p(g["e_min"], C.c_float), _p(g["e_status"], C.c_int32),
                                                 C.c_float(float(g["sigma"])), C.c_float(float(g["stretch_th"])), _p(map_pos, C.c_float),
                                                 C.c_int32(len(f_map)), _p(f_map, C.c_int32), _p(f_status, C.c_int32), _p(f_uv, C.c_float),
                                                 _p(f_pos, C.c_float), _p(f_node, C.c_uint8), _p(qt, C.c_double), C.c_float(float(scale)), C.byref(med), C.byref(nl),
                                                 _p(lost, C.c_int32), C.byref(nn), C.byref(ns), tr, C.c_int32(1024), C.byref(ntr), C.byref(st))
    if rc != 0:
        raise RuntimeError("nrs_cpu_track_deform_solve_embedded failed: %d" % rc)
    return dict(pose_q=qt[:4].copy(), pose_t=qt[4:].copy(), f_pos=f_pos, f_status=f_status, map_pos=map_pos, graph=g, median=float(med.value),
                lost=lost[:nl.value].tolist(), n_nodes=nn.value, n_skinned=ns.value, trace=_trace_list(tr, ntr.value),
                stats={k: getattr(st, k) for k, _ in TStats._fields_})


class LucasKanadeCpp:
    """oracle/nrs_cpu_lk.hpp behind the interface of lk_oracle.LucasKanadeOracle (no mask)"""

    def __init__(self, win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4, lib=None):
        self.lib = lib or load()
        self.lib.nrs_cpu_lk_create.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.nrs_cpu_lk_create(C.c_int32(win), C.c_int32(max_level), C.c_int32(max_iters), C.c_float(epsilon), C.c_float(min_eig)))
        self.n = 0

    def close(self):
        if self.h:
            self.lib.nrs_cpu_lk_destroy(self.h)
            self.h = None

    def set_reference(self, img, pts):
        img = np.ascontiguousarray(img, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        self.n = len(pts)
        self.lib.nrs_cpu_lk_set_reference(self.h, _p(img, C.c_uint8), C.c_int32(img.shape[1]), C.c_int32(img.shape[0]), C.c_int32(img.strides[0]),
                                          C.c_int32(self.n), _p(pts, C.c_float))

    def track(self, img, pts, status, initial_flow=True, min_ssim=0.7):
        img = np.ascontiguousarray(img, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2).copy()
        status = np.ascontiguousarray(status, np.int32).copy()
        assert len(pts) == self.n
        good = C.c_int32(0)
        ssim = np.zeros(self.n, np.float32)
        self.lib.nrs_cpu_lk_track(self.h, _p(img, C.c_uint8), C.c_int32(img.shape[1]), C.c_int32(img.shape[0]), C.c_int32(img.strides[0]),
                                  _p(pts, C.c_float), _p(status, C.c_int32), C.c_int32(1 if initial_flow else 0), C.c_float(min_ssim), C.byref(good),
                                  _p(ssim, C.c_float))
        return pts, status, good.value, ssim


def nd_plan_check(pos, last, pairs, lib=None):
    """oracle/nd_host.cpp nrs_cpu_nd_plan_check: the structural invariants of the nested-dissection plan (0 = all hold)"""
    lib = lib or load()
    pos = np.ascontiguousarray(pos, np.float64)
    last = None if last is None else np.ascontiguousarray(last, np.uint8)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    return int(lib.nrs_cpu_nd_plan_check(C.c_int32(len(pos)), _p(pos, C.c_double), _p(last, C.c_uint8), C.c_int32(len(pairs)), _p(pairs, C.c_int32)))


def nd_solve(pos, last, pairs, Dn, Vp, bn, lam=0.0, lib=None):
    """oracle/nd_host.cpp: the nested-dissection plan of nr-slam_amd/csrc/nrs_nd_plan.hpp + its host reference solve of
    (A + lam I) x = b (A: diagonal blocks Dn [n,3,3], pair blocks Vp [p,3,3] with rows = pairs[:,0]'s components).
    Returns (ok, x [n,3], stats dict)."""
    lib = lib or load()
    pos = np.ascontiguousarray(pos, np.float64)
    n = len(pos)
    last = None if last is None else np.ascontiguousarray(last, np.uint8)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    Dn, Vp, bn = (np.ascontiguousarray(a, np.float64) for a in (Dn, Vp, bn))
    x = np.zeros((n, 3))
    st = np.zeros(8, np.int64)
    rc = lib.nrs_cpu_nd_solve(C.c_int32(n), _p(pos, C.c_double), _p(last, C.c_uint8), C.c_int32(len(pairs)), _p(pairs, C.c_int32),
                              _p(Dn, C.c_double), _p(Vp, C.c_double), _p(bn, C.c_double), C.c_double(lam), _p(x, C.c_double), _p(st, C.c_int64))
    if rc not in (0, -6):
        raise RuntimeError("nrs_cpu_nd_solve: the plan could not be built (rc %d)" % rc)
    keys = ("fronts", "levels", "max_s", "max_b", "L_doubles", "U_doubles", "flops", "workgroups")
    return rc == 0, x, dict(zip(keys, st.tolist()))
