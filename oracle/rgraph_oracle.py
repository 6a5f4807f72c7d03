"""CPU restatement of RegularizationGraph at the reference's density (TEST INFRASTRUCTURE ONLY).

Follows modules/map/regularization_graph.cc:27-146 (constructor / SetSigma, AddEdge, GetEdges with EdgeComparator,
UpdateConnection, UpdateVertex), the all-pairs initialisation of Map::InitializeRegularizationGraph
(modules/map/map.cc:148-166) and the graph growth of Mapping (modules/mapping/mapping.cc:240-256).  The reference
keeps btree_map<ID, btree_map<ID, shared_ptr<Edge>>>; here the shared edge of (i, j) is one cell of symmetric
N x N arrays, which is the same state.  fp32 throughout, weights through oracle/nrs_oracle.interpolation_weight.
Parity unpinned (the reference holds no vector for the graph); the product's nrs_rgraph_* is held to this."""
import numpy as np

import nrs_oracle as O

F32 = np.float32
NONE = 255


class DenseGraph:
    def __init__(self, capacity, sigma, stretch_th):
        self.n = capacity
        self.stretch_th = F32(stretch_th)
        self.maxd = np.zeros((capacity, capacity), F32)
        self.mind = np.zeros((capacity, capacity), F32)
        self.d0 = np.zeros((capacity, capacity), F32)
        self.st = np.full((capacity, capacity), NONE, np.uint8)
        self.set_sigma(sigma)

    def set_sigma(self, sigma):                      # :27-36
        self.sigma = F32(sigma)
        self.min_w = O.min_weight(self.sigma)

    @staticmethod
    def _dist(pos, i, js):
        d = pos[i][None, :] - pos[js]
        return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F32)).astype(F32)

    def add_edges(self, pos, new_ids, other_ids):    # AddEdge :38-55 for every (new, other), new != other
        pos = np.asarray(pos, F32)
        other_ids = np.asarray(other_ids)
        for i in new_ids:
            js = other_ids[other_ids != i]
            d = self._dist(pos, i, js)
            for a in (self.maxd, self.mind, self.d0):
                a[i, js] = d
                a[js, i] = d
            self.st[i, js] = O.GRAPH_NEUTRAL
            self.st[js, i] = O.GRAPH_NEUTRAL

    def update_vertex(self, pos, i):                 # UpdateVertex :130-146 + UpdateConnection :89-128
        pos = np.asarray(pos, F32)
        js = np.where(self.st[i] != NONE)[0]
        d = self._dist(pos, i, js)
        mx = np.maximum(self.maxd[i, js], d)
        mn = np.minimum(self.mind[i, js], d)
        bad = np.abs((mx - mn) / mn) > self.stretch_th
        self.maxd[i, js] = mx
        self.maxd[js, i] = mx
        self.mind[i, js] = mn
        self.mind[js, i] = mn
        s = np.where(bad, O.GRAPH_BAD, self.st[i, js]).astype(np.uint8)
        self.st[i, js] = s
        self.st[js, i] = s
        return int(np.sum(~bad))

    def get_edges(self, i):                          # GetEdges :71-87: (other, weight, first_distance, status) in order
        js = np.where(self.st[i] != NONE)[0]         # btree_map order = ascending id
        w = O.interpolation_weight(self.maxd[i, js], self.sigma)
        pos = O.get_edges(js, w, self.st[i, js].astype(np.int64), self.min_w)
        return js[pos], w[pos], self.d0[i, js][pos], self.st[i, js][pos].astype(np.int64)


class LiteralGraph:
    """the same operations one pair at a time on dictionaries, as the reference's containers do them (small N only):
    holds DenseGraph's vectorised forms to the literal walk in tests/test_oracle_rgraph_cpu.py"""

    def __init__(self, sigma, stretch_th):
        self.g = {}
        self.stretch_th = F32(stretch_th)
        self.sigma = F32(sigma)
        self.min_w = O.min_weight(self.sigma)

    def add_edge(self, a, b, rel):
        rel = np.asarray(rel, F32)
        d = F32(np.sqrt(F32(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2])))
        e = dict(d0=d, w=O.interpolation_weight(d, self.sigma), st=O.GRAPH_NEUTRAL, mx=d, mn=d)
        self.g.setdefault(a, {})[b] = e
        self.g.setdefault(b, {})[a] = e

    def update_vertex(self, pos, i):
        good = 0
        for o in sorted(self.g[i]):
            e = self.g[i][o]
            dl = np.asarray(pos[i], F32) - np.asarray(pos[o], F32)
            d = F32(np.sqrt(F32(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2])))
            if d > e["mx"]:
                e["mx"] = d
            if d < e["mn"]:
                e["mn"] = d
            e["w"] = O.interpolation_weight(e["mx"], self.sigma)
            if abs(F32((e["mx"] - e["mn"]) / e["mn"])) > self.stretch_th:
                e["st"] = O.GRAPH_BAD
            else:
                good += 1
        return good

    def get_edges(self, i):
        items = sorted(self.g[i].items())
        # EdgeComparator :61-69 (status asc, weight desc); ties: ascending id (documented choice)
        items.sort(key=lambda kv: (kv[1]["st"], -float(kv[1]["w"]), kv[0]))
        out = []
        for o, e in items:
            if e["w"] < self.min_w:
                break
            out.append((o, e["w"], e["d0"], e["st"]))
        return out
