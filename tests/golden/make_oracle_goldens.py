"""Generates the small golden fixtures under tests/golden/ from the oracle (oracle/*.py).

These are the restatement's OWN outputs (the reference holds no fixture for this path and cannot be
built here): they pin the oracle against drift and give the GPU tests a second,
file-based target.  Inputs are stored next to the expected outputs so the files are self-contained.
    python tests/golden/make_oracle_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))

import lk_oracle as LK  # noqa: E402
import shi_oracle as SH  # noqa: E402
import nrs_oracle as O  # noqa: E402
import nrs_synth as S  # noqa: E402


def pose_only():
    tp = S.make_tracking_problem(120, 101)
    m = tp["status"] == 0
    q, t, inl = O.pose_only_solve(tp["model"], tp["prm"], tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"])
    np.savez_compressed(os.path.join(HERE, "pose_only_120.npz"), model=tp["model"], prm=tp["prm"], uv=tp["uv"][m],
                        X=tp["X_prev"][m], pose_q=tp["pose_q"], pose_t=tp["pose_t"], out_q=q, out_t=t, out_inlier=inl)


def dba():
    p = S.make_dba_problem(90, 3, 102)
    nb = p["nbr"]
    e = O.dba_build(p["kf_points"], nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
    tr = []
    q, t, pts, nit = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                 e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, tr)
    kf_rowptr = np.concatenate([[0], np.cumsum([len(k) for k in p["kf_points"]])]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "dba_90x3.npz"), model=p["model"], prm=p["prm"], scale=p["scale"],
                        poses_q=p["poses_q"], poses_t=p["poses_t"], lm_xyz=p["lm_xyz"], lm_kf=p["lm_kf"], lm_uv=p["lm_uv"],
                        kf_rowptr=kf_rowptr, kf_pt=np.concatenate(p["kf_points"]).astype(np.int32),
                        nbr_rowptr=nb["rowptr"], nbr_col=nb["col"], nbr_w=nb["w"], nbr_d0=nb["d0"], nbr_status=nb["status"],
                        sp_ij=e["sp_ij"], sp_d0=e["sp_d0"], dm_idx=e["dm_idx"], dm_w=e["dm_w"],
                        out_q=q, out_t=t, out_pts=pts, out_iters=nit,
                        out_accepted=np.array([x["accepted"] for x in tr]), out_chi=np.array([x["chi"] for x in tr]),
                        out_chi_new=np.array([x["chi_new"] for x in tr]), out_lam=np.array([x["lam"] for x in tr]))


def track():
    tp = S.make_tracking_problem(150, 103)
    n = tp["n_points"]
    g = tp["graph"]
    r = O.track_deform_solve(tp["model"], tp["prm"], g, tp["X_prev"], np.arange(n), tp["status"], tp["uv"], tp["X_prev"],
                             tp["pose_q"], tp["pose_t"], tp["scale"])
    keys = ("rowptr", "col", "eid", "e_w", "e_d0", "e_max", "e_min", "e_status", "e_ij")
    np.savez_compressed(os.path.join(HERE, "track_150.npz"), model=tp["model"], prm=tp["prm"], scale=tp["scale"],
                        sigma=g["sigma"], stretch_th=g["stretch_th"], min_w=g["min_w"], status=tp["status"], uv=tp["uv"],
                        X_prev=tp["X_prev"], pose_q=tp["pose_q"], pose_t=tp["pose_t"],
                        **{"g_" + k: g[k] for k in keys},
                        out_q=r["pose_q"], out_t=r["pose_t"], out_f_pos=r["f_pos"], out_f_status=r["f_status"],
                        out_map_pos=r["map_pos"], out_lost=np.array(r["lost"], np.int32), out_median=r["median"],
                        out_e_status=r["graph"]["e_status"], out_e_w=r["graph"]["e_w"], out_e_max=r["graph"]["e_max"],
                        out_e_min=r["graph"]["e_min"])


def lk():
    sq = S.make_lk_sequence(60, 104, wh=(240, 180), flow_px=4.0)
    lkt = LK.LucasKanadeOracle(max_level=2)
    lkt.set_reference(sq["im0"], sq["pts"])
    st = np.zeros(len(sq["pts"]), np.int32)
    xy, st2, good, ssim = lkt.track(sq["im1"], sq["pts"] + np.float32(0.5), st)
    pyr = LK.build_pyramid(sq["im0"], 2)
    np.savez_compressed(os.path.join(HERE, "lk_240x180.npz"), im0=sq["im0"], im1=sq["im1"], pts=sq["pts"],
                        level1=pyr[1].img, level2=pyr[2].img, deriv1=LK.scharr_deriv(pyr[1].img),
                        tpl_gray0=np.stack([lkt.Iref[0][i] for i in range(len(sq["pts"]))]),
                        tpl_mean0=np.stack([[lkt.meanI[0][i], lkt.meanI2[0][i]] for i in range(len(sq["pts"]))]),
                        out_xy=xy, out_status=st2, out_good=good, out_ssim=ssim)


def shi():
    """Two successive Extract calls of one extractor (its buffers persist): literal single-pass restatement."""
    sq = S.make_lk_sequence(10, 105, wh=(160, 120), flow_px=3.0)
    mask = np.ones((120, 160), np.uint8)
    mask[:, :12] = 0
    mask[100:, :] = 0
    ex = SH.ShiTomasi(5)
    xy0, id0 = ex.extract(sq["im0"], None, mask, literal=True)
    keep = xy0[::3]                                        # the frame keeps a third of them; the rest may be found again
    xy1, id1 = ex.extract(sq["im1"], keep, mask, literal=True)
    np.savez_compressed(os.path.join(HERE, "shi_160x120.npz"), im0=sq["im0"], im1=sq["im1"], mask=mask, nms=5,
                        out_xy0=xy0, out_id0=id0, prev1=keep, out_xy1=xy1, out_id1=id1,
                        out_scores1=ex.scores, out_xg1=ex.Xg, out_yg1=ex.Yg)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        pose_only()
        dba()
        track()
        lk()
        shi()
    for f in sorted(os.listdir(HERE)):
        print("%8d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))
