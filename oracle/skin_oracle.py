"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): CPU restatement of the node
selection of the skinned mode (include/nrs.h "N2"; SURVEY.md 8d C2: "M farthest-point-sampled nodes").  The reference has
no node set of its own (modules/map/regularization_graph.h:34-96: every map point is a vertex), so this pins the
product's kernel to a plain statement of farthest point sampling, not to reference output: parity unpinned for the
selection; the skinned SOLVE is the reference's stage 2 (g2o_optimization.cc:476-553) and is held to
oracle/nrs_oracle.track_deform_solve."""
import numpy as np

F32 = np.float32


def select_nodes(pos, n_nodes, eligible=None):
    """pick 0 = lowest eligible index; pick k = eligible point with the largest fp32 squared distance
    ((dx*dx + dy*dy) + dz*dz) to the picks so far, ties to the lowest index"""
    pos = np.asarray(pos, F32).reshape(-1, 3)
    n = len(pos)
    ok = np.ones(n, bool) if eligible is None else np.asarray(eligible).astype(bool)
    if ok.sum() < n_nodes:
        raise ValueError("fewer eligible points than nodes")
    mind = np.where(ok, F32(np.inf), F32(-1)).astype(F32)
    pick = int(np.argmax(ok))
    out = []
    for k in range(n_nodes):
        out.append(pick)
        if k + 1 == n_nodes:
            break
        d = pos - pos[pick]
        d2 = ((d[:, 0] * d[:, 0]).astype(F32) + (d[:, 1] * d[:, 1]).astype(F32)).astype(F32)
        d2 = (d2 + (d[:, 2] * d[:, 2]).astype(F32)).astype(F32)
        upd = mind >= 0
        mind[upd] = np.minimum(mind[upd], d2[upd])
        mind[pick] = F32(-1)
        pick = int(np.argmax(mind))                                  # first maximum = lowest index
    return np.asarray(out, np.int32)
