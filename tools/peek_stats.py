"""Statistics behind the early-rejection thresholds: NRS_PEEK_DEBUG=1 makes the engine print the gain
ratio seen at every milestone (1e-1 .. 1e-4) of every LM trial and never reject early; this tool runs
many seeded problems (the oracle_sweep / early_reject_sweep generators, small ill-conditioned windows
included) and reports, per milestone, the most negative estimate among trials whose FINAL gain ratio is
>= 0 (those must never be rejected) and the largest |estimate - final|."""
import sys, os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
    import numpy as np, nrs, nrs_synth as S
    ctx = nrs.Context()
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    for seed in range(lo, hi):
        rng = np.random.default_rng(100 + seed)
        model = S.PINHOLE if seed % 3 else S.KB8
        n, k = int(rng.integers(60, 260)), int(rng.integers(1, 5))
        p = S.make_dba_problem(n, k, 3000 + seed, model)
        e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
        cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
        print("[problem] BA small %d" % seed, file=sys.stderr, flush=True)
        ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 8)
        n = int(rng.integers(100, 500))
        tp = S.make_tracking_problem(n, 4000 + seed, model)
        cam = nrs.make_camera(tp["model"], tp["prm"]); fm = np.arange(n, dtype=np.int32)
        print("[problem] a2 small %d" % seed, file=sys.stderr, flush=True)
        ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
        rng = np.random.default_rng(seed)
        n, k = int(rng.integers(150, 1500)), int(rng.integers(2, 7))
        p = S.make_dba_problem(n, k, 1000 + seed, model)
        e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
        cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
        print("[problem] BA %d" % seed, file=sys.stderr, flush=True)
        ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 8)
        n = int(rng.integers(300, 3000))
        tp = S.make_tracking_problem(n, 2000 + seed, model)
        cam = nrs.make_camera(tp["model"], tp["prm"]); fm = np.arange(n, dtype=np.int32)
        print("[problem] a2 %d" % seed, file=sys.stderr, flush=True)
        ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
    sys.exit(0)
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 40
env = dict(os.environ, NRS_PEEK_DEBUG="1")
r = subprocess.run([sys.executable, __file__, "--child", "0", str(nseed)], env=env, capture_output=True, text=True)
cur, prob, rows = {}, "", []
for l in r.stderr.splitlines():
    if l.startswith("[problem]"):
        prob = l[10:]; continue
    m = re.match(r"\[peek\] it (\d+) trial (\d+) lvl (\d) pit (\d+) rho (\S+) relinc (\S+)", l)
    if m:
        cur[int(m.group(3))] = (float(m.group(5)), float(m.group(6))); continue
    m = re.match(r"\[peek\] it (\d+) trial (\d+) FINAL pit (\d+) rho (\S+)", l)
    if m:
        rows.append((prob, int(m.group(1)), int(m.group(2)), float(m.group(4)), dict(cur))); cur = {}
print("trials", len(rows))
for L in (1, 2, 3, 4):
    GUARD = 1e-5                                   # PEEK_MIN_REL_INCREASE: looks below it never reject
    have = [(r[3], r[4][L][0] if r[4][L][1] > GUARD else 1e9, r) for r in rows if L in r[4] and abs(r[3]) < 1e6]
    if not have: continue
    pos = [x for x in have if x[0] >= 0]
    worst = min(pos, key=lambda x: x[1])
    err = max([x for x in have if x[1] < 1e8], key=lambda x: abs(x[1] - x[0]))
    print("level %d: %d looks; most negative estimate of an eventually non-negative trial: %.3f (final %.3f, %s it %d trial %d); "
          "largest |estimate - final|: %.3f (estimate %.3f, final %.3f, %s)" % (L, len(have), worst[1], worst[0], worst[2][0], worst[2][1], worst[2][2],
          abs(err[1] - err[0]), err[1], err[0], err[2][0]))
    for thr in (-0.03, -0.1, -0.25, -0.5, -1.0, -2.0, -3.0, -5.0):
        print("     threshold %.2f would wrongly reject %d of %d non-negative trials, and catch %d of %d negative ones" % (
            thr, sum(1 for x in pos if x[1] < thr), len(pos), sum(1 for x in have if x[0] < 0 and x[1] < thr), sum(1 for x in have if x[0] < 0)))

# the most dangerous looks: final >= 0 but a very negative estimate (with its relative chi2 increase)
for L in (1, 2):
    danger = sorted([(r[4][L][0], r[3], r[4][L][1], r[0], r[1], r[2]) for r in rows if L in r[4] and r[3] >= 0 and r[4][L][1] > 1e-5])[:8]
    print("level %d, worst estimates of eventually non-negative trials (estimate, final, rel chi2 increase, problem, it, trial):" % L)
    for d_ in danger:
        print("    %.3f  %.3f  %.2e  %s it %d trial %d" % d_)
    neg = sorted([r[4][L][1] for r in rows if L in r[4] and r[3] < -1 and r[4][L][0] < -1])
    if neg:
        print("    rel chi2 increase of clearly rejected trials (final < -1, estimate < -1): min %.2e, 5%% %.2e, median %.2e" % (neg[0], neg[len(neg) // 20], neg[len(neg) // 2]))
