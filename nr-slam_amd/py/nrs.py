"""ctypes binding of libnrs_hip.so (include/nrs.h) for the tests and bench.py.

Thin by design: every function maps 1:1 onto a C-ABI entry point, takes/returns NumPy arrays and
raises NrsError (with nrs_last_error) on a non-zero status.  There is no fallback: if the
shared library is missing the import fails; if no HIP device is usable nrs_create fails.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRS_LIB") or os.path.join(os.path.dirname(_HERE), "libnrs_hip.so")

OK = 0
STATUS_NAMES = {0: "NRS_OK", -1: "NRS_ERR_INVALID", -2: "NRS_ERR_NO_DEVICE", -3: "NRS_ERR_HIP",
                -4: "NRS_ERR_ALLOC", -5: "NRS_ERR_STATE", -6: "NRS_ERR_NUMERIC", -7: "NRS_ERR_COMM"}
COMM_ID_BYTES = 128

# every symbol include/nrs.h declares (tests check that the library exports all of them)
SYMBOLS = ["nrs_create", "nrs_options_init", "nrs_destroy", "nrs_last_error", "nrs_device_name", "nrs_get_profile",
           "nrs_reset_profile", "nrs_stream", "nrs_pose_only_solve", "nrs_dba_build_edges",
           "nrs_dba_solve", "nrs_dba_upload", "nrs_dba_build_edges_embedded", "nrs_dba_upload_embedded", "nrs_dba_download_skinned", "nrs_dba_solve_embedded", "nrs_dba_reset", "nrs_dba_optimize",
           "nrs_dba_download", "nrs_dba_residuals", "nrs_dba_gradient", "nrs_dba_pack_hash", "nrs_dba_solve_window", "nrs_dba_window_edges", "nrs_debug_pcg_solve", "nrs_debug_kft", "nrs_debug_set", "nrs_debug_nd_solve", "nrs_debug_nd_cache_stats", "nrs_track_deform_solve_embedded",
           "nrs_graph_select_neighbours", "nrs_graph_update", "nrs_track_deform_solve",
           "nrs_klt_configure", "nrs_klt_clear", "nrs_klt_num_points", "nrs_klt_set_reference",
           "nrs_klt_track", "nrs_klt_get_template", "nrs_klt_insert_template", "nrs_klt_get_templates",
           "nrs_klt_insert_templates", "nrs_klt_archive_templates", "nrs_klt_insert_archived",
           "nrs_shi_configure", "nrs_shi_extract", "nrs_shi_buffers",
           "nrs_comm_unique_id", "nrs_comm_init_rccl", "nrs_comm_rank", "nrs_shard_plan",
           "nrs_local_group_create", "nrs_local_group_destroy", "nrs_comm_init_local",
           "nrs_rgraph_create", "nrs_rgraph_destroy", "nrs_rgraph_set_sigma", "nrs_rgraph_min_weight", "nrs_rgraph_add_edges",
           "nrs_rgraph_update", "nrs_rgraph_get_edges", "nrs_rgraph_edge", "nrs_rgraph_rows", "nrs_triangulate_batch", "nrs_track_deform_solve_rg",
           "nrs_skin_select_nodes", "nrs_dba_stats"]


class NrsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (STATUS_NAMES.get(code, "?"), code, msg))
        self.code = code


class Camera(C.Structure):
    _fields_ = [("model", C.c_int32), ("params", C.c_float * 8)]


# ---- debug / A-B switches (include/nrs.h "Debug switches"): a context reads the NRS_* environment ONCE when it is created; afterwards they are
# changed through nrs_debug_set only.  debug_set() here applies a switch to every live Context of this process and to the ones created later.
import weakref
_DEBUG = {}
_LIVE = weakref.WeakSet()


def debug_set(name, value):
    """value None: unset.  (tests / probes: the replacement of os.environ[name] = value for a library that no longer reads the environment per call)"""
    if value is None:
        _DEBUG.pop(name, None)
    else:
        _DEBUG[name] = str(value)
    for c in list(_LIVE):
        c.debug_set(name, value)


def debug_clear():
    for k in list(_DEBUG):
        debug_set(k, None)


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("struct_size", C.c_uint32), ("pcg_rtol", C.c_double), ("pcg_max_iters", C.c_int32),
                ("pcg_batch", C.c_int32), ("profile", C.c_int32), ("exact_trials", C.c_int32), ("direct_solve", C.c_int32),
                ("embedded_solver", C.c_int32)]


class LmTrial(C.Structure):
    _fields_ = [("round", C.c_int32), ("iter", C.c_int32), ("trial", C.c_int32),
                ("accepted", C.c_int32), ("solver_ok", C.c_int32), ("inner_iters", C.c_int32),
                ("early_rejected", C.c_int32), ("reserved", C.c_int32),
                ("lam", C.c_double), ("chi2", C.c_double), ("chi2_new", C.c_double),
                ("rho", C.c_double)]


class LmTrace(C.Structure):
    _fields_ = [("trials", C.POINTER(LmTrial)), ("capacity", C.c_int32), ("count", C.c_int32),
                ("iterations", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("rowptr", C.POINTER(C.c_int32)), ("col", C.POINTER(C.c_int32)),
                ("eid", C.POINTER(C.c_int32)), ("n_edges", C.c_int32), ("e_w", C.POINTER(C.c_float)),
                ("e_d0", C.POINTER(C.c_float)), ("e_max", C.POINTER(C.c_float)), ("e_min", C.POINTER(C.c_float)),
                ("e_status", C.POINTER(C.c_int32)), ("sigma", C.c_float), ("stretch_th", C.c_float)]


class KltConfig(C.Structure):
    _fields_ = [("win_size", C.c_int32), ("max_level", C.c_int32), ("max_iters", C.c_int32),
                ("epsilon", C.c_float), ("min_eig_threshold", C.c_float)]


class Profile(C.Structure):
    _fields_ = [("linearize_ms", C.c_double), ("linearize_launches", C.c_int64),
                ("spmv_ms", C.c_double), ("spmv_launches", C.c_int64),
                ("vec_ms", C.c_double), ("vec_launches", C.c_int64),
                ("update_ms", C.c_double), ("update_launches", C.c_int64)]


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError("libnrs_hip.so not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C nr-slam_amd`): %s" % path)
    lib = C.CDLL(path)
    lib.nrs_last_error.restype = C.c_char_p
    lib.nrs_stream.restype = C.c_void_p
    lib.nrs_destroy.restype = None
    lib.nrs_local_group_destroy.restype = None
    lib.nrs_rgraph_destroy.restype = None
    lib.nrs_rgraph_destroy.argtypes = [C.c_void_p]
    lib.nrs_rgraph_min_weight.restype = C.c_float
    lib.nrs_rgraph_min_weight.argtypes = [C.c_void_p]
    lib.nrs_local_group_destroy.argtypes = [C.c_void_p]
    lib.nrs_comm_init_local.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    return lib


def comm_unique_id(lib=None):
    """RCCL unique id (bytes) made on rank 0; the caller broadcasts it (torch.distributed, MPI, a file ...)."""
    lib = lib or load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib.nrs_comm_unique_id(buf, C.c_int32(COMM_ID_BYTES))
    if rc != OK:
        raise NrsError(rc, "nrs_comm_unique_id (is librccl loadable?)")
    return bytes(buf)


def shard_plan(n_kf, lm_kf, world, lib=None):
    """Keyframe ranges of a sharded BA window (host only)."""
    lib = lib or load_library()
    lm_kf = _i32(lm_kf)
    kb = np.zeros(world + 1, np.int32)
    rc = lib.nrs_shard_plan(C.c_int32(n_kf), C.c_int32(len(lm_kf)), _p(lm_kf, C.c_int32), C.c_int32(world), _p(kb, C.c_int32))
    if rc != OK:
        raise NrsError(rc, "nrs_shard_plan")
    return kb


class LocalGroup:
    """Test harness: `world` ranks as threads of this process, contexts on the same GPU (include/nrs.h)."""
    def __init__(self, world, lib=None):
        self.lib = lib or load_library()
        self.world = world
        self.h = C.c_void_p()
        rc = self.lib.nrs_local_group_create(C.c_int32(world), C.byref(self.h))
        if rc != OK:
            raise NrsError(rc, "nrs_local_group_create")

    def close(self):
        if self.h:
            self.lib.nrs_local_group_destroy(self.h)
            self.h = C.c_void_p()


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, np.int32)


def make_camera(model, params):
    cam = Camera()
    cam.model = int(model)
    prm = np.zeros(8, np.float32)
    prm[:len(params)] = np.asarray(params, np.float32)
    for i in range(8):
        cam.params[i] = float(prm[i])
    return cam


class GraphArrays:
    """Owns contiguous copies of a flat graph dict (nrs_synth.build_graph) and the nrs_graph view."""

    def __init__(self, g):
        self.rowptr, self.col, self.eid = _i32(g["rowptr"]).copy(), _i32(g["col"]).copy(), _i32(g["eid"]).copy()
        self.e_w, self.e_d0 = _f32(g["e_w"]).copy(), _f32(g["e_d0"]).copy()
        self.e_max, self.e_min = _f32(g["e_max"]).copy(), _f32(g["e_min"]).copy()
        self.e_status = _i32(g["e_status"]).copy()
        self.c = Graph(len(self.rowptr) - 1, _p(self.rowptr, C.c_int32), _p(self.col, C.c_int32),
                       _p(self.eid, C.c_int32), len(self.e_w), _p(self.e_w, C.c_float), _p(self.e_d0, C.c_float),
                       _p(self.e_max, C.c_float), _p(self.e_min, C.c_float), _p(self.e_status, C.c_int32),
                       float(g["sigma"]), float(g["stretch_th"]))

    def as_dict(self, g):
        out = dict(g)
        out.update(e_w=self.e_w.copy(), e_max=self.e_max.copy(), e_min=self.e_min.copy(), e_status=self.e_status.copy())
        return out


class Trace:
    def __init__(self, capacity=512):
        self.buf = (LmTrial * capacity)()
        self.c = LmTrace(C.cast(self.buf, C.POINTER(LmTrial)), capacity, 0, 0)

    @property
    def trials(self):
        n = min(self.c.count, self.c.capacity)
        return [dict(round=t.round, iter=t.iter, trial=t.trial, accepted=bool(t.accepted),
                     ok=bool(t.solver_ok), inner=t.inner_iters, early=bool(t.early_rejected), lam=t.lam, chi=t.chi2,
                     chi_new=t.chi2_new, rho=t.rho) for t in self.buf[:n]]

    @property
    def iterations(self):
        return self.c.iterations


def dba_build_edges(kf_points, graph, lib=None):
    """Host-side (no GPU) edge construction, reference g2o_optimization.cc:927-1137."""
    lib = lib or load_library()
    n_kf = len(kf_points)
    kf_rowptr = np.zeros(n_kf + 1, np.int32)
    kf_rowptr[1:] = np.cumsum([len(k) for k in kf_points])
    kf_pt = _i32(np.concatenate(kf_points)) if n_kf else np.zeros(0, np.int32)
    n_points = len(graph["rowptr"]) - 1
    rp, col, w, d0, st = (_i32(graph["rowptr"]), _i32(graph["col"]), _f32(graph["w"]),
                          _f32(graph["d0"]), _i32(graph["status"]))
    ns, nd = C.c_int32(0), C.c_int32(0)
    args = [C.c_int32(n_kf), _p(kf_rowptr, C.c_int32), _p(kf_pt, C.c_int32), C.c_int32(n_points),
            _p(rp, C.c_int32), _p(col, C.c_int32), _p(w, C.c_float), _p(d0, C.c_float), _p(st, C.c_int32)]
    rc = lib.nrs_dba_build_edges(*args, C.byref(ns), None, None, C.byref(nd), None, None)
    if rc != OK:
        raise NrsError(rc, "nrs_dba_build_edges (count)")
    sp_ij = np.zeros((ns.value, 2), np.int32)
    sp_d0 = np.zeros(ns.value, np.float32)
    dm_idx = np.zeros((nd.value, 4), np.int32)
    dm_w = np.zeros(nd.value, np.float32)
    rc = lib.nrs_dba_build_edges(*args, C.byref(ns), _p(sp_ij, C.c_int32), _p(sp_d0, C.c_float),
                                 C.byref(nd), _p(dm_idx, C.c_int32), _p(dm_w, C.c_float))
    if rc != OK:
        raise NrsError(rc, "nrs_dba_build_edges (fill)")
    return dict(sp_ij=sp_ij, sp_d0=sp_d0, dm_idx=dm_idx, dm_w=dm_w)


def dba_build_edges_embedded(kf_points, is_node, graph, lib=None):
    """Host-side edge construction of the EMBEDDED window (include/nrs.h nrs_dba_build_edges_embedded, N2b): node copies, springs /
    dampers between them, skinned observations with their node copies and normalised weights."""
    lib = lib or load_library()
    n_kf = len(kf_points)
    kf_rowptr = np.zeros(n_kf + 1, np.int32)
    kf_rowptr[1:] = np.cumsum([len(k) for k in kf_points])
    kf_pt = _i32(np.concatenate(kf_points)) if n_kf else np.zeros(0, np.int32)
    n_points = len(graph["rowptr"]) - 1
    node = np.ascontiguousarray(is_node, np.uint8)
    assert len(node) == n_points
    rp, col, w, d0, st = (_i32(graph["rowptr"]), _i32(graph["col"]), _f32(graph["w"]), _f32(graph["d0"]), _i32(graph["status"]))
    nl, ns, nd, nk = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    args = [C.c_int32(n_kf), _p(kf_rowptr, C.c_int32), _p(kf_pt, C.c_int32), C.c_int32(n_points), _p(node, C.c_uint8),
            _p(rp, C.c_int32), _p(col, C.c_int32), _p(w, C.c_float), _p(d0, C.c_float), _p(st, C.c_int32)]
    rc = lib.nrs_dba_build_edges_embedded(*args, C.byref(nl), None, C.byref(ns), None, None, C.byref(nd), None, None, C.byref(nk), None, None, None)
    if rc != OK:
        raise NrsError(rc, "nrs_dba_build_edges_embedded (count)")
    lm_obs = np.zeros(nl.value, np.int32)
    sp_ij, sp_d0 = np.zeros((ns.value, 2), np.int32), np.zeros(ns.value, np.float32)
    dm_idx, dm_w = np.zeros((nd.value, 4), np.int32), np.zeros(nd.value, np.float32)
    sk_obs, sk_node, sk_omega = np.zeros(nk.value, np.int32), np.zeros((nk.value, 11), np.int32), np.zeros((nk.value, 11), np.float64)
    rc = lib.nrs_dba_build_edges_embedded(*args, C.byref(nl), _p(lm_obs, C.c_int32), C.byref(ns), _p(sp_ij, C.c_int32), _p(sp_d0, C.c_float),
                                          C.byref(nd), _p(dm_idx, C.c_int32), _p(dm_w, C.c_float), C.byref(nk), _p(sk_obs, C.c_int32),
                                          _p(sk_node, C.c_int32), _p(sk_omega, C.c_double))
    if rc != OK:
        raise NrsError(rc, "nrs_dba_build_edges_embedded (fill)")
    return dict(lm_obs=lm_obs, sp_ij=sp_ij, sp_d0=sp_d0, dm_idx=dm_idx, dm_w=dm_w, sk_obs=sk_obs, sk_node=sk_node, sk_omega=sk_omega)


class RGraph:
    """nrs_rgraph_* (include/nrs.h): the dense, device-resident RegularizationGraph of a context"""

    def __init__(self, ctx, capacity, sigma, stretch_th=1.1):
        self.ctx, self.lib, self.cap = ctx, ctx.lib, capacity
        self.h = C.c_void_p()
        ctx._chk(self.lib.nrs_rgraph_create(ctx.h, C.c_int32(capacity), C.c_float(sigma), C.c_float(stretch_th), C.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.nrs_rgraph_destroy(self.h)
            self.h = C.c_void_p()

    def _pos(self, pos):
        pos = _f32(pos).reshape(-1, 3)
        assert len(pos) == self.cap
        return pos

    def set_sigma(self, sigma):
        self.ctx._chk(self.lib.nrs_rgraph_set_sigma(self.h, C.c_float(sigma)))

    def min_weight(self):
        return float(self.lib.nrs_rgraph_min_weight(self.h))

    def add_edges(self, pos, new_ids, other_ids):
        pos, a, b = self._pos(pos), _i32(new_ids), _i32(other_ids)
        self.ctx._chk(self.lib.nrs_rgraph_add_edges(self.h, _p(pos, C.c_float), C.c_int32(len(a)), _p(a, C.c_int32), C.c_int32(len(b)), _p(b, C.c_int32)))

    def update(self, pos, ids):
        pos, ids = self._pos(pos), _i32(ids)
        good = np.zeros(len(ids), np.int32)
        self.ctx._chk(self.lib.nrs_rgraph_update(self.h, _p(pos, C.c_float), C.c_int32(len(ids)), _p(ids, C.c_int32), _p(good, C.c_int32)))
        return good

    def get_edges(self, ids, cap_per_point=256):
        ids = _i32(ids)
        n = len(ids)
        cnt = np.zeros(n, np.int32)
        col, st = np.zeros((n, cap_per_point), np.int32), np.zeros((n, cap_per_point), np.int32)
        w, d0 = np.zeros((n, cap_per_point), np.float32), np.zeros((n, cap_per_point), np.float32)
        self.ctx._chk(self.lib.nrs_rgraph_get_edges(self.h, C.c_int32(n), _p(ids, C.c_int32), C.c_int32(cap_per_point), _p(cnt, C.c_int32),
                                                    _p(col, C.c_int32), _p(w, C.c_float), _p(d0, C.c_float), _p(st, C.c_int32)))
        return cnt, col, w, d0, st

    def edge(self, i, j):
        out = (C.c_float * 4)()
        st = C.c_int32(0)
        self.ctx._chk(self.lib.nrs_rgraph_edge(self.h, C.c_int32(i), C.c_int32(j), out, C.byref(st)))
        return dict(w=out[0], d0=out[1], max=out[2], min=out[3], status=st.value)

    def rows(self, ids):
        ids = _i32(ids)
        n = len(ids)
        mx, mn, d0 = (np.zeros((n, self.cap), np.float32) for _ in range(3))
        st = np.zeros((n, self.cap), np.uint8)
        self.ctx._chk(self.lib.nrs_rgraph_rows(self.h, C.c_int32(n), _p(ids, C.c_int32), _p(mx, C.c_float), _p(mn, C.c_float), _p(d0, C.c_float),
                                               _p(st, C.c_uint8)))
        return mx, mn, d0, st


class Context:
    def __init__(self, device=-1, pcg_rtol=0.0, pcg_max_iters=0, pcg_batch=0, profile=0, exact_trials=0, direct_solve=0, embedded_solver=0):
        self.lib = load_library()
        opt = Options(device, C.sizeof(Options), pcg_rtol, pcg_max_iters, pcg_batch, profile, exact_trials, direct_solve, embedded_solver)
        self.h = C.c_void_p()
        rc = self.lib.nrs_create(C.byref(self.h), C.byref(opt))
        if rc != OK:
            raise NrsError(rc, "nrs_create failed (no usable HIP device?)")
        _LIVE.add(self)
        for k, v in _DEBUG.items():
            self.debug_set(k, v)

    def close(self):
        if self.h:
            self.lib.nrs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise NrsError(rc, self.lib.nrs_last_error(self.h).decode())

    # ---- f3: Shi-Tomasi extraction
    def shi_configure(self, nms_window=5):
        self._chk(self.lib.nrs_shi_configure(self.h, C.c_int32(nms_window)))

    def shi_extract(self, img, prev_xy=None, mask=None, capacity=65536):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        prev = None if prev_xy is None or len(prev_xy) == 0 else _f32(np.asarray(prev_xy).reshape(-1, 2))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        xy = np.zeros((capacity, 2), np.float32)
        ids = np.zeros(capacity, np.int32)
        n = C.c_int32(0)
        self._chk(self.lib.nrs_shi_extract(self.h, _p(img, C.c_uint8), C.c_int32(w), C.c_int32(h), C.c_int32(w),
                                           _p(m, C.c_uint8), C.c_int32(w), C.c_int32(0 if prev is None else len(prev)),
                                           _p(prev, C.c_float), C.c_int32(capacity), _p(xy, C.c_float), _p(ids, C.c_int32),
                                           C.byref(n)))
        self._shi_shape = (h, w)
        k = min(n.value, capacity)
        return xy[:k].copy(), ids[:k].copy(), n.value

    def shi_buffers(self):
        h, w = self._shi_shape
        sc = np.zeros((h, w), np.float32)
        xg = np.zeros((h, w), np.int16)
        yg = np.zeros((h, w), np.int16)
        self._chk(self.lib.nrs_shi_buffers(self.h, _p(sc, C.c_float), _p(xg, C.c_int16), _p(yg, C.c_int16)))
        return sc, xg, yg

    def comm_init_rccl(self, world, rank, uid):
        buf = (C.c_uint8 * len(uid)).from_buffer_copy(uid)
        self._chk(self.lib.nrs_comm_init_rccl(self.h, C.c_int32(world), C.c_int32(rank), buf, C.c_int32(len(uid))))

    def comm_init_local(self, group, rank):
        self._chk(self.lib.nrs_comm_init_local(self.h, group.h, C.c_int32(rank)))

    def comm_rank(self):
        r, w = C.c_int32(0), C.c_int32(1)
        self._chk(self.lib.nrs_comm_rank(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._chk(self.lib.nrs_device_name(self.h, buf, 256))
        return buf.value.decode()

    def profile(self):
        p = Profile()
        self._chk(self.lib.nrs_get_profile(self.h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def reset_profile(self):
        self._chk(self.lib.nrs_reset_profile(self.h))

    # ---- a1
    def pose_only_solve(self, cam, uv, X, pose_q, pose_t, trace=None):
        uv, X = _f32(uv).reshape(-1, 2), _f32(X).reshape(-1, 3)
        n = len(uv)
        qt = np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)])
        inl = np.zeros(n, np.uint8)
        self._chk(self.lib.nrs_pose_only_solve(self.h, C.byref(cam), C.c_int32(n), _p(uv, C.c_float),
                                               _p(X, C.c_float), _p(qt, C.c_double), _p(inl, C.c_uint8),
                                               C.byref(trace.c) if trace else None))
        return qt[:4].copy(), qt[4:].copy(), inl.astype(bool)

    # ---- f2
    def triangulate_batch(self, cam, tb, cand_ids, min_track=5, debug=False):
        """tb: flat temporal buffer dict (nrs_synth.make_temporal_buffer); returns (status[n], xyz[n,3][, debug[n,4]])"""
        F, n = tb["has_kp"].shape
        poses = _f32(tb["poses"]).reshape(F, 7)
        has_kp, has_lm = np.ascontiguousarray(tb["has_kp"], np.uint8), np.ascontiguousarray(tb["has_lm"], np.uint8)
        kp, lm = _f32(tb["kp_xy"]).reshape(F, n, 2), _f32(tb["lm_xyz"]).reshape(F, n, 3)
        st, cand = _i32(tb["status"]), _i32(cand_ids)
        o_st, o_xyz = np.zeros(len(cand), np.int32), np.zeros((len(cand), 3), np.float32)
        dbg = np.zeros((len(cand), 4), np.float64) if debug else None
        self._chk(self.lib.nrs_triangulate_batch(self.h, C.byref(cam), C.c_int32(F), _p(poses, C.c_float), C.c_int32(n), _p(has_kp, C.c_uint8),
                                                 _p(kp, C.c_float), _p(has_lm, C.c_uint8), _p(lm, C.c_float), _p(st, C.c_int32), C.c_int32(len(cand)),
                                                 _p(cand, C.c_int32), C.c_int32(min_track), _p(o_st, C.c_int32), _p(o_xyz, C.c_float),
                                                 _p(dbg, C.c_double)))
        return (o_st, o_xyz, dbg) if debug else (o_st, o_xyz)

    # ---- a19 / a20
    def graph_select_neighbours(self, g):
        ga = GraphArrays(g)
        n, nnz = len(ga.rowptr) - 1, len(ga.col)
        rp, oc, oe = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz, np.int32)
        self._chk(self.lib.nrs_graph_select_neighbours(self.h, C.byref(ga.c), _p(rp, C.c_int32), _p(oc, C.c_int32),
                                                       _p(oe, C.c_int32)))
        return rp, oc[:rp[-1]].copy(), oe[:rp[-1]].copy()

    def graph_update(self, g, pos, ids):
        ga = GraphArrays(g)
        pos = _f32(pos).reshape(-1, 3)
        ids = _i32(ids)
        good = np.zeros(len(ids), np.int32)
        self._chk(self.lib.nrs_graph_update(self.h, C.byref(ga.c), _p(pos, C.c_float), C.c_int32(len(ids)),
                                            _p(ids, C.c_int32), _p(good, C.c_int32)))
        return ga.as_dict(g), good

    # ---- a2
    def track_deform_solve(self, cam, g, map_pos, f_map, f_status, f_uv, f_pos, pose_q, pose_t, scale, trace=None):
        ga = GraphArrays(g)
        map_pos = _f32(map_pos).reshape(-1, 3).copy()
        f_map, f_status = _i32(f_map), _i32(f_status).copy()
        f_uv, f_pos = _f32(f_uv).reshape(-1, 2), _f32(f_pos).reshape(-1, 3).copy()
        qt = np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)])
        med = C.c_float(0)
        n_lost = C.c_int32(0)
        lost = np.zeros(len(map_pos), np.int32)
        self._chk(self.lib.nrs_track_deform_solve(
            self.h, C.byref(cam), C.byref(ga.c), _p(map_pos, C.c_float), C.c_int32(len(f_map)), _p(f_map, C.c_int32),
            _p(f_status, C.c_int32), _p(f_uv, C.c_float), _p(f_pos, C.c_float), _p(qt, C.c_double), C.c_float(scale),
            C.byref(med), C.byref(n_lost), _p(lost, C.c_int32), C.byref(trace.c) if trace else None))
        return dict(pose_q=qt[:4].copy(), pose_t=qt[4:].copy(), f_pos=f_pos, f_status=f_status, map_pos=map_pos,
                    graph=ga.as_dict(g), median=float(med.value), lost=lost[:n_lost.value].tolist())

    def track_deform_solve_rg(self, cam, rg, map_pos, f_map, f_status, f_uv, f_pos, pose_q, pose_t, scale, trace=None, cap_per_point=128):
        """a2 on the device-resident dense graph `rg` (nrs.RGraph of this context); the graph is updated in place"""
        map_pos = _f32(map_pos).reshape(-1, 3).copy()
        assert len(map_pos) == rg.cap
        f_map, f_status = _i32(f_map), _i32(f_status).copy()
        f_uv, f_pos = _f32(f_uv).reshape(-1, 2), _f32(f_pos).reshape(-1, 3).copy()
        qt = np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)])
        med, n_lost = C.c_float(0), C.c_int32(0)
        lost = np.zeros(len(map_pos), np.int32)
        self._chk(self.lib.nrs_track_deform_solve_rg(
            self.h, C.byref(cam), rg.h, C.c_int32(rg.cap), C.c_int32(cap_per_point), _p(map_pos, C.c_float), C.c_int32(len(f_map)),
            _p(f_map, C.c_int32), _p(f_status, C.c_int32), _p(f_uv, C.c_float), _p(f_pos, C.c_float), _p(qt, C.c_double), C.c_float(scale),
            C.byref(med), C.byref(n_lost), _p(lost, C.c_int32), C.byref(trace.c) if trace else None))
        return dict(pose_q=qt[:4].copy(), pose_t=qt[4:].copy(), f_pos=f_pos, f_status=f_status, map_pos=map_pos,
                    median=float(med.value), lost=lost[:n_lost.value].tolist())

    def track_deform_solve_embedded(self, cam, rg, map_pos, f_map, f_status, f_uv, f_pos, f_node, pose_q, pose_t, scale, trace=None, cap_per_point=128):
        """N2 (include/nrs.h nrs_track_deform_solve_embedded): a2 with a node set; the other optimised landmarks are skinned to their nodes"""
        map_pos = _f32(map_pos).reshape(-1, 3).copy()
        assert len(map_pos) == rg.cap
        f_map, f_status = _i32(f_map), _i32(f_status).copy()
        f_uv, f_pos = _f32(f_uv).reshape(-1, 2), _f32(f_pos).reshape(-1, 3).copy()
        f_node = np.ascontiguousarray(f_node, np.uint8)
        assert len(f_node) == len(f_map)
        qt = np.concatenate([np.asarray(pose_q, np.float64), np.asarray(pose_t, np.float64)])
        med, n_lost = C.c_float(0), C.c_int32(0)
        lost = np.zeros(len(map_pos), np.int32)
        self._chk(self.lib.nrs_track_deform_solve_embedded(
            self.h, C.byref(cam), rg.h, C.c_int32(rg.cap), C.c_int32(cap_per_point), _p(map_pos, C.c_float), C.c_int32(len(f_map)),
            _p(f_map, C.c_int32), _p(f_status, C.c_int32), _p(f_uv, C.c_float), _p(f_pos, C.c_float), _p(f_node, C.c_uint8), _p(qt, C.c_double),
            C.c_float(scale), C.byref(med), C.byref(n_lost), _p(lost, C.c_int32), C.byref(trace.c) if trace else None))
        return dict(pose_q=qt[:4].copy(), pose_t=qt[4:].copy(), f_pos=f_pos, f_status=f_status, map_pos=map_pos,
                    median=float(med.value), lost=lost[:n_lost.value].tolist())

    def dba_pack_hash(self):
        """checksums of the packed arrays of the resident BA problem (include/nrs.h nrs_dba_pack_hash); [21] = 1 if device-built"""
        out = (C.c_uint64 * 24)()
        self._chk(self.lib.nrs_dba_pack_hash(self.h, out))
        return list(out)

    def dba_stats(self):
        """sizes of the resident BA problem on this rank (include/nrs.h nrs_dba_stats)"""
        st = (C.c_int64 * 5)()
        self._chk(self.lib.nrs_dba_stats(self.h, st))
        return dict(rows=st[0], packed_rows=st[1], spring_slots=st[2], damper_slots=st[3], device_bytes=st[4])

    # ---- N2: skinned mode
    def skin_select_nodes(self, pos, n_nodes, eligible=None):
        """farthest point sampling of n_nodes graph nodes among the eligible points (pick order)"""
        pos = _f32(pos).reshape(-1, 3)
        el = None if eligible is None else np.ascontiguousarray(eligible, np.uint8)
        out = np.zeros(n_nodes, np.int32)
        self._chk(self.lib.nrs_skin_select_nodes(self.h, C.c_int32(len(pos)), _p(pos, C.c_float), _p(el, C.c_uint8) if el is not None else None,
                                                 C.c_int32(n_nodes), _p(out, C.c_int32)))
        return out

    # ---- a21-a23
    def klt_configure(self, win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4):
        cfg = KltConfig(win, max_level, max_iters, epsilon, min_eig)
        self._klt_levels = max_level + 1
        self._chk(self.lib.nrs_klt_configure(self.h, C.byref(cfg)))

    def klt_clear(self):
        self._chk(self.lib.nrs_klt_clear(self.h))

    def klt_num_points(self):
        return self.lib.nrs_klt_num_points(self.h)

    def klt_set_reference(self, img, xy, mask=None):
        img = np.ascontiguousarray(img, np.uint8)
        xy = _f32(xy).reshape(-1, 2)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._chk(self.lib.nrs_klt_set_reference(self.h, _p(img, C.c_uint8), C.c_int32(img.shape[1]), C.c_int32(img.shape[0]),
                                                 C.c_int32(img.strides[0]), _p(m, C.c_uint8), C.c_int32(len(xy)), _p(xy, C.c_float)))

    def klt_track(self, img, xy, status, initial_flow=True, min_ssim=0.7):
        img = np.ascontiguousarray(img, np.uint8)
        xy = _f32(xy).reshape(-1, 2).copy()
        st = _i32(status).copy()
        good = C.c_int32(0)
        ssim = np.full(len(xy), np.nan, np.float32)
        self._chk(self.lib.nrs_klt_track(self.h, _p(img, C.c_uint8), C.c_int32(img.shape[1]), C.c_int32(img.shape[0]),
                                         C.c_int32(img.strides[0]), C.c_int32(len(xy)), _p(xy, C.c_float), _p(st, C.c_int32),
                                         C.c_int32(1 if initial_flow else 0), C.c_float(min_ssim), C.byref(good), _p(ssim, C.c_float)))
        return xy, st, good.value, ssim

    def klt_get_template(self, idx):
        L = getattr(self, "_klt_levels", 5)
        xy = np.zeros(2, np.float32)
        gray, grad = np.zeros((L, 21, 21), np.int16), np.zeros((L, 21, 21, 2), np.int16)
        mean, valid = np.zeros((L, 2), np.float32), np.zeros(L, np.uint8)
        self._chk(self.lib.nrs_klt_get_template(self.h, C.c_int32(idx), _p(xy, C.c_float), _p(gray, C.c_int16),
                                                _p(grad, C.c_int16), _p(mean, C.c_float), _p(valid, C.c_uint8)))
        return dict(xy=xy, gray=gray, grad=grad, mean=mean, valid=valid)

    def klt_insert_template(self, t):
        xy, gray, grad = _f32(t["xy"]), np.ascontiguousarray(t["gray"], np.int16), np.ascontiguousarray(t["grad"], np.int16)
        mean, valid = _f32(t["mean"]), np.ascontiguousarray(t["valid"], np.uint8)
        self._chk(self.lib.nrs_klt_insert_template(self.h, _p(xy, C.c_float), _p(gray, C.c_int16), _p(grad, C.c_int16),
                                                   _p(mean, C.c_float), _p(valid, C.c_uint8)))

    def klt_get_templates(self, first, count):
        """list of `count` template dicts (same layout as klt_get_template), one device read."""
        L = getattr(self, "_klt_levels", 5)
        xy = np.zeros((count, 2), np.float32)
        gray, grad = np.zeros((count, L, 21, 21), np.int16), np.zeros((count, L, 21, 21, 2), np.int16)
        mean, valid = np.zeros((count, L, 2), np.float32), np.zeros((count, L), np.uint8)
        self._chk(self.lib.nrs_klt_get_templates(self.h, C.c_int32(first), C.c_int32(count), _p(xy, C.c_float), _p(gray, C.c_int16),
                                                 _p(grad, C.c_int16), _p(mean, C.c_float), _p(valid, C.c_uint8)))
        return [dict(xy=xy[i], gray=gray[i], grad=grad[i], mean=mean[i], valid=valid[i]) for i in range(count)]

    def klt_archive_templates(self, slots, keys):
        """copy the templates of tracker slots into this context's device-side archive under `keys` (include/nrs.h nrs_klt_archive_templates)"""
        slots, keys = _i32(slots), _i32(keys)
        assert len(slots) == len(keys)
        self._chk(self.lib.nrs_klt_archive_templates(self.h, C.c_int32(len(slots)), _p(slots, C.c_int32), _p(keys, C.c_int32)))

    def klt_insert_archived(self, src, keys, xy):
        """append the archive entries `keys` of context `src` to this context's tracker at positions xy (nrs_klt_insert_archived)"""
        keys, xy = _i32(keys), _f32(xy).reshape(-1, 2)
        assert len(keys) == len(xy)
        self._chk(self.lib.nrs_klt_insert_archived(self.h, src.h, C.c_int32(len(keys)), _p(keys, C.c_int32), _p(xy, C.c_float)))

    def klt_insert_templates(self, ts):
        """append several templates (each as returned by klt_get_template; levels beyond this tracker's are ignored)."""
        L = getattr(self, "_klt_levels", 5)
        if not ts:
            return
        xy = np.ascontiguousarray(np.stack([_f32(t["xy"]) for t in ts]), np.float32)
        gray = np.ascontiguousarray(np.stack([np.asarray(t["gray"], np.int16)[:L] for t in ts]))
        grad = np.ascontiguousarray(np.stack([np.asarray(t["grad"], np.int16)[:L] for t in ts]))
        mean = np.ascontiguousarray(np.stack([_f32(t["mean"])[:L] for t in ts]), np.float32)
        valid = np.ascontiguousarray(np.stack([np.asarray(t["valid"], np.uint8)[:L] for t in ts]))
        self._chk(self.lib.nrs_klt_insert_templates(self.h, C.c_int32(len(ts)), _p(xy, C.c_float), _p(gray, C.c_int16),
                                                    _p(grad, C.c_int16), _p(mean, C.c_float), _p(valid, C.c_uint8)))

    # ---- a3
    def _dba_args(self, cam, poses_qt, lm_xyz, lm_kf, lm_uv, edges, scale):
        self._keep = (np.ascontiguousarray(poses_qt, np.float64).reshape(-1, 7), _f32(lm_xyz).reshape(-1, 3),
                      _i32(lm_kf), _f32(lm_uv).reshape(-1, 2), _i32(edges["sp_ij"]).reshape(-1, 2),
                      _f32(edges["sp_d0"]), _i32(edges["dm_idx"]).reshape(-1, 4), _f32(edges["dm_w"]))
        pq, xyz, kf, uv, sp, d0, dm, dw = self._keep
        return [self.h, C.byref(cam), C.c_int32(len(pq)), _p(pq, C.c_double), C.c_int32(len(xyz)),
                _p(xyz, C.c_float), _p(kf, C.c_int32), _p(uv, C.c_float), C.c_int32(len(sp)),
                _p(sp, C.c_int32), _p(d0, C.c_float), C.c_int32(len(dm)), _p(dm, C.c_int32),
                _p(dw, C.c_float), C.c_float(scale)]

    def dba_solve(self, cam, poses_qt, lm_xyz, lm_kf, lm_uv, edges, scale, iters=5, trace=None):
        args = self._dba_args(cam, np.array(poses_qt, np.float64), np.array(lm_xyz, np.float32),
                              lm_kf, lm_uv, edges, scale)
        self._chk(self.lib.nrs_dba_solve(*args, C.c_int32(iters), C.byref(trace.c) if trace else None))
        return self._keep[0].copy(), self._keep[1].copy()

    def dba_solve_window(self, cam, poses_qt, kf_points, lm_xyz, lm_uv, graph, scale, iters=5, trace=None):
        """LocalDeformableBundleAdjustment in one call (include/nrs.h nrs_dba_solve_window): edge construction included"""
        n_kf = len(kf_points)
        kf_rowptr = np.zeros(n_kf + 1, np.int32)
        kf_rowptr[1:] = np.cumsum([len(k) for k in kf_points])
        kf_pt = _i32(np.concatenate(kf_points))
        pq = np.ascontiguousarray(np.array(poses_qt, np.float64)).reshape(-1, 7)
        xyz = np.array(lm_xyz, np.float32).reshape(-1, 3).copy()
        uv = _f32(lm_uv).reshape(-1, 2)
        rp, col, w, d0, st = (_i32(graph["rowptr"]), _i32(graph["col"]), _f32(graph["w"]), _f32(graph["d0"]), _i32(graph["status"]))
        self._chk(self.lib.nrs_dba_solve_window(self.h, C.byref(cam), C.c_int32(n_kf), _p(pq, C.c_double), _p(kf_rowptr, C.c_int32), _p(kf_pt, C.c_int32),
                                                _p(xyz, C.c_float), _p(uv, C.c_float), C.c_int32(len(rp) - 1), _p(rp, C.c_int32), _p(col, C.c_int32),
                                                _p(w, C.c_float), _p(d0, C.c_float), _p(st, C.c_int32), C.c_float(scale), C.c_int32(iters),
                                                C.byref(trace.c) if trace else None))
        return pq, xyz

    def dba_window_edges(self):
        """edge lists of the resident window as the device built them (None if they were built on the host)"""
        ns, nd = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.nrs_dba_window_edges(self.h, C.byref(ns), None, None, C.byref(nd), None, None))
        sp_ij, sp_d0 = np.zeros((ns.value, 2), np.int32), np.zeros(ns.value, np.float32)
        dm_idx, dm_w = np.zeros((nd.value, 4), np.int32), np.zeros(nd.value, np.float32)
        rc = self.lib.nrs_dba_window_edges(self.h, C.byref(ns), _p(sp_ij, C.c_int32), _p(sp_d0, C.c_float), C.byref(nd), _p(dm_idx, C.c_int32), _p(dm_w, C.c_float))
        if rc == -5:
            return None
        self._chk(rc)
        return dict(sp_ij=sp_ij, sp_d0=sp_d0, dm_idx=dm_idx, dm_w=dm_w)

    def dba_upload(self, cam, poses_qt, lm_xyz, lm_kf, lm_uv, edges, scale):
        args = self._dba_args(cam, poses_qt, lm_xyz, lm_kf, lm_uv, edges, scale)
        self._n_kf, self._n_lm = len(self._keep[0]), len(self._keep[1])
        self._n_sp, self._n_dm = len(self._keep[4]), len(self._keep[6])
        self._chk(self.lib.nrs_dba_upload(*args))

    # ---- N2b: the embedded window (nrs_dba_*_embedded): w = nrs_synth.embedded_window(p, e), e = dba_build_edges_embedded(...)
    def _dba_args_embedded(self, cam, poses_qt, w, e, scale):
        args = self._dba_args(cam, poses_qt, w["lm_xyz"], w["lm_kf"], w["lm_uv"], e, scale)
        self._keep_sk = (_i32(w["sk_kf"]), _f32(w["sk_uv"]).reshape(-1, 2), np.array(w["sk_xyz"], np.float32).reshape(-1, 3).copy(),
                         _i32(e["sk_node"]).reshape(-1, 11), np.ascontiguousarray(e["sk_omega"], np.float64).reshape(-1, 11))
        kf, uv, xyz, node, om = self._keep_sk
        self._n_skin = len(kf)
        return args[:-1] + [C.c_int32(len(kf)), _p(kf, C.c_int32), _p(uv, C.c_float), _p(xyz, C.c_float), _p(node, C.c_int32), _p(om, C.c_double), args[-1]]

    def dba_upload_embedded(self, cam, poses_qt, w, e, scale):
        args = self._dba_args_embedded(cam, poses_qt, w, e, scale)
        self._n_kf, self._n_lm = len(self._keep[0]), len(self._keep[1])
        self._n_sp, self._n_dm = len(self._keep[4]), len(self._keep[6])
        self._chk(self.lib.nrs_dba_upload_embedded(*args))

    def dba_download_skinned(self):
        xyz = np.zeros((self._n_skin, 3), np.float64)
        self._chk(self.lib.nrs_dba_download_skinned(self.h, _p(xyz, C.c_double)))
        return xyz

    def dba_solve_embedded(self, cam, poses_qt, w, e, scale, iters=5, trace=None):
        """one shot; returns (poses_qt, node copies fp32, skinned points fp32)"""
        args = self._dba_args_embedded(cam, np.array(poses_qt, np.float64), dict(w, lm_xyz=np.array(w["lm_xyz"], np.float32)), e, scale)
        self._chk(self.lib.nrs_dba_solve_embedded(*args, C.c_int32(iters), C.byref(trace.c) if trace else None))
        return self._keep[0].copy(), self._keep[1].copy(), self._keep_sk[2].copy()

    def dba_reset(self):
        self._chk(self.lib.nrs_dba_reset(self.h))

    def dba_optimize(self, iters=5, trace=None):
        self._chk(self.lib.nrs_dba_optimize(self.h, C.c_int32(iters), C.byref(trace.c) if trace else None))

    def dba_download(self):
        pq = np.zeros((self._n_kf, 7), np.float64)
        xyz = np.zeros((self._n_lm, 3), np.float64)
        self._chk(self.lib.nrs_dba_download(self.h, _p(pq, C.c_double), _p(xyz, C.c_double)))
        return pq, xyz

    def dba_residuals(self):
        rr = np.zeros((self._n_lm, 2), np.float64)
        rs = np.zeros(self._n_sp, np.float64)
        rd = np.zeros((self._n_dm, 3), np.float64)
        self._chk(self.lib.nrs_dba_residuals(self.h, _p(rr, C.c_double), _p(rs, C.c_double), _p(rd, C.c_double)))
        return rr, rs, rd

    def debug_pcg_solve(self, Hpp21, bp, D6, Hpl18, bl, lam=0.0):
        """include/nrs.h nrs_debug_pcg_solve: returns (x[6 + 3n], PCG iterations)."""
        D6 = np.ascontiguousarray(D6, np.float64).reshape(-1, 6)
        n = len(D6)
        Hpp21, bp = np.ascontiguousarray(Hpp21, np.float64), np.ascontiguousarray(bp, np.float64)
        Hpl18 = np.ascontiguousarray(Hpl18, np.float64).reshape(n, 18)
        bl = np.ascontiguousarray(bl, np.float64).reshape(n, 3)
        x = np.zeros(6 + 3 * n, np.float64)
        it = C.c_int32(0)
        self._chk(self.lib.nrs_debug_pcg_solve(self.h, C.c_int32(n), _p(Hpp21, C.c_double), _p(bp, C.c_double), _p(D6, C.c_double),
                                               _p(Hpl18, C.c_double), _p(bl, C.c_double), C.c_double(lam), _p(x, C.c_double), C.byref(it)))
        return x, it.value

    def nd_cache_stats(self):
        """include/nrs.h nrs_debug_nd_cache_stats: (problems that reused a cached plan, plans built)"""
        out = np.zeros(2, np.int64)
        self._chk(self.lib.nrs_debug_nd_cache_stats(self.h, _p(out, C.c_int64)))
        return int(out[0]), int(out[1])

    def debug_nd_solve(self, pos, last, pairs, Dn, Vp, bn, lam=0.0, repeats=0):
        """include/nrs.h nrs_debug_nd_solve: the direct (nested-dissection) solver's device kernels on an explicit block system.
        Returns (ok, x [n,3], stats dict, ms per solve)."""
        pos = np.ascontiguousarray(pos, np.float64)
        n = len(pos)
        last = None if last is None else np.ascontiguousarray(last, np.uint8)
        pairs = _i32(np.asarray(pairs).reshape(-1, 2))
        Dn, Vp, bn = (np.ascontiguousarray(a, np.float64) for a in (Dn, Vp, bn))
        x = np.zeros((n, 3))
        st = np.zeros(8, np.int64)
        ms = C.c_double(0)
        rc = self.lib.nrs_debug_nd_solve(self.h, C.c_int32(n), _p(pos, C.c_double), None if last is None else _p(last, C.c_uint8), C.c_int32(len(pairs)),
                                         _p(pairs, C.c_int32), _p(Dn, C.c_double), _p(Vp, C.c_double), _p(bn, C.c_double), C.c_double(lam),
                                         C.c_int32(repeats), _p(x, C.c_double), _p(st, C.c_int64), C.byref(ms))
        if rc not in (0, -6):
            self._chk(rc)
        keys = ("fronts", "levels", "max_s", "max_b", "L_doubles", "U_doubles", "flops", "workgroups")
        return rc == 0, x, dict(zip(keys, st.tolist())), ms.value

    def debug_set(self, name, value):
        """a debug / A-B switch of THIS context (include/nrs.h nrs_debug_set); value None: unset"""
        if self.h:
            self._chk(self.lib.nrs_debug_set(self.h, name.encode(), None if value is None else str(value).encode()))

    def debug_kft_info(self):
        """the keyframe-block factorisation of the resident embedded window: dict(on, K, ld, nb, m, mib, nf[K], np[K])"""
        o = np.zeros(6 + 2 * max(1, self._n_kf), np.int32)
        self._chk(self.lib.nrs_debug_kft(self.h, C.c_double(0.0), C.c_int32(0), C.c_int32(0), None, None, _p(o, C.c_int32)))
        K = self._n_kf
        return dict(on=bool(o[0]), K=int(o[1]), ld=int(o[2]), nb=int(o[3]), m=int(o[4]), mib=int(o[5]), nf=o[6:6 + K].copy(), np=o[6 + K:6 + 2 * K].copy())

    def debug_kft_block(self, lam, k, coupling=False):
        ld = self.debug_kft_info()["ld"]
        a = np.zeros((ld, ld), np.float64)
        self._chk(self.lib.nrs_debug_kft(self.h, C.c_double(lam), C.c_int32(2 if coupling else 1), C.c_int32(k), None, _p(a, C.c_double), None))
        return a

    def debug_kft_apply(self, lam, r):
        r = np.ascontiguousarray(r, np.float64)
        u = np.zeros_like(r)
        self._chk(self.lib.nrs_debug_kft(self.h, C.c_double(lam), C.c_int32(3), C.c_int32(0), _p(r, C.c_double), _p(u, C.c_double), None))
        return u

    def debug_kft_index(self):
        o = np.zeros((self._n_lm, 2), np.int32)
        self._chk(self.lib.nrs_debug_kft(self.h, C.c_double(0.0), C.c_int32(4), C.c_int32(0), None, None, _p(o, C.c_int32)))
        return o

    def dba_gradient(self):
        n = 6 * self._n_kf + 3 * self._n_lm
        b = np.zeros(n, np.float64)
        d = np.zeros(n, np.float64)
        self._chk(self.lib.nrs_dba_gradient(self.h, _p(b, C.c_double), _p(d, C.c_double)))
        return b, d


def skinned_status(f_status, f_map, nodes):
    """Skinned mode (include/nrs.h, N2): the nodes keep TRACKED_WITH_3D (0), every other TRACKED_WITH_3D point of the frame
    becomes TRACKED (1: in the frame, no 3D) and is carried by stage 2 of the pose-and-deformation solve"""
    st = np.array(f_status, np.int32)
    nodes = np.asarray(nodes, np.int64)
    is_node = np.zeros(int(max(np.max(f_map), nodes.max() if nodes.size else -1)) + 1, bool)
    is_node[nodes] = True
    fm = np.asarray(f_map)
    demote = (st == 0) & (fm >= 0) & ~is_node[np.maximum(fm, 0)]
    st[demote] = 1
    return st
