"""oracle/orc_deformation.py -- TEST INFRASTRUCTURE ONLY: a dense numpy restatement of the reference's deformation-graph optimiser
(EF/Deformation.cpp:68-220, EF/Utils/DeformationGraph.cpp) for checking instancefusion_amd/host/ifx_deformation.hpp.  Everything is f64 and
dense here (Jacobian by the chain rule on explicit matrices, normal equations by numpy.linalg.solve, rotations re-orthonormalised by SVD), so
the two agree to rounding, not bit for bit.  Parity status: unpinned by the reference (it holds no tests or fixtures for this part)."""
import numpy as np

K = 4
W_REG, W_CON = 10.0, 100.0


class Graph:
    def __init__(self, xyzt):
        a = np.asarray(xyzt, np.float64).reshape(-1, 4)
        self.g = a[:, :3].copy()                       # node positions
        self.time = a[:, 3].astype(np.int64)
        n = len(a)
        self.R = np.tile(np.eye(3), (n, 1, 1))
        self.t = np.zeros((n, 3))
        self.nb = []                                    # DeformationGraph.cpp:256-292
        for i in range(n):
            if i < K // 2:
                self.nb.append([j for j in range(K + 1) if j != i])
            elif i < n - K // 2:
                self.nb.append([v for d in range(K // 2) for v in (i - (d + 1), i + (d + 1))])
            else:
                self.nb.append([j for j in range(n - (K + 1), n) if j != i])

    # DeformationGraph.cpp:294-409 (and :137-254 for poses)
    def weigh(self, p, time):
        n = len(self.g)
        lo, hi = 0, n - 1
        mid = (lo + hi) // 2
        while hi >= lo:
            mid = (lo + hi) // 2
            if self.time[mid] < time:
                lo = mid + 1
            elif self.time[mid] > time:
                hi = mid - 1
            else:
                break
        lo = min(lo, n - 1)
        hic = max(hi, 0)
        d = lambda i: abs(int(self.time[i]) - int(time))
        if d(lo) <= d(mid) and d(lo) <= d(hic):
            found = lo
        elif d(mid) <= d(lo) and d(mid) <= d(hic):
            found = mid
        else:
            found = hic
        cand = list(range(found, max(found - 20, -1), -1))
        if len(cand) < 20:
            cand += list(range(found + 1, n))[: 20 - len(cand)]
        p32 = np.asarray(p, np.float32)
        dist = np.array([np.sqrt(((self.g[j].astype(np.float32) - p32) ** 2).sum(dtype=np.float32)) for j in cand], np.float32)
        order = np.argsort(dist, kind="stable")
        dmax = float(dist[order[K]])
        w = np.array([(1.0 - float(dist[order[j]]) / dmax) ** 2 for j in range(K)])
        w /= w.sum()
        nodes = np.array([cand[order[j]] for j in range(K)])
        o = np.argsort(nodes)
        return nodes[o], w[o]

    def deform(self, wm, v):
        nodes, w = wm
        return sum(w[i] * (self.R[nodes[i]] @ (v - self.g[nodes[i]]) + self.g[nodes[i]] + self.t[nodes[i]]) for i in range(K))


def constrain(xyzt, cons, poses, pose_times, fern_match, relax, last_deform_time):
    """cons: list of dicts(src, target, src_time, target_time, relative, pin).  Returns dict(ok, raw_graph, poses, error, mean_cons, new_rel)."""
    G = Graph(xyzt)
    n = len(G.g)
    pool, ptime = [], []
    for c in cons:
        pool.append(np.asarray(c["src"], np.float64)); ptime.append(c["src_time"]); c["sid"] = len(pool) - 1
        if c["relative"]:
            pool.append(np.asarray(c["target"], np.float64)); ptime.append(c["target_time"]); c["tid"] = len(pool) - 1
    wmap = [G.weigh(pool[i], ptime[i]) for i in range(len(pool))]

    def mean_cons():
        e = [np.linalg.norm(G.deform(wmap[c["sid"]], pool[c["sid"]]) - np.asarray(c["target"], np.float64)) for c in cons if not c["relative"]]
        return sum(e) / len(cons)

    out = dict(ok=False, raw_graph=None, poses=[p.copy() for p in poses], error=0.0, mean_cons=mean_cons(), new_rel=[])
    optimised = True
    ldt = 0 if (fern_match or relax) else last_deform_time
    if fern_match and out["mean_cons"] < 0.06:
        optimised = False
    else:
        enabled = G.time > ldt
        col = np.full(n, -1)
        col[enabled] = np.arange(enabled.sum()) * 12
        ncol = int(enabled.sum()) * 12

        def linearise():
            rows, res = [], []

            def new_row():
                rows.append(np.zeros(ncol)); return rows[-1]

            for j in range(n):                                   # rotation
                if not enabled[j]:
                    continue
                R = G.R[j]
                for a, b in ((0, 1), (0, 2), (1, 2)):
                    r = new_row(); res.append(R[:, a] @ R[:, b])
                    r[col[j] + 3 * a: col[j] + 3 * a + 3] += R[:, b]; r[col[j] + 3 * b: col[j] + 3 * b + 3] += R[:, a]
                for a in range(3):
                    r = new_row(); res.append(R[:, a] @ R[:, a] - 1.0)
                    r[col[j] + 3 * a: col[j] + 3 * a + 3] += 2 * R[:, a]
            s = np.sqrt(W_REG)
            for j in range(n):                                   # regularisation
                for m in G.nb[j]:
                    if not (enabled[j] or enabled[m]):
                        continue
                    d = G.g[m] - G.g[j]
                    v = G.R[j] @ d + G.g[j] + G.t[j] - (G.g[m] + G.t[m])
                    for x in range(3):
                        r = new_row(); res.append(v[x] * s)
                        if enabled[j]:
                            for c in range(3):
                                r[col[j] + 3 * c + x] += d[c] * s    # d/dR(x, c): column-major variable 3c + x
                            r[col[j] + 9 + x] += s
                        if enabled[m]:
                            r[col[m] + 9 + x] -= s
            s = np.sqrt(W_CON)
            for c in cons:                                       # constraints
                maps = [(wmap[c["sid"]], pool[c["sid"]], 1.0)]
                if c["relative"]:
                    maps.append((wmap[c["tid"]], pool[c["tid"]], -1.0))
                if not any(enabled[nd] for (wm, _, _) in maps for nd in wm[0]):
                    continue
                p = G.deform(*maps[0][:2])
                q = G.deform(*maps[1][:2]) if c["relative"] else np.asarray(c["target"], np.float64)
                for x in range(3):
                    r = new_row(); res.append((p[x] - q[x]) * s)
                    for (nodes, w), v, sign in maps:
                        for i in range(K):
                            nd = nodes[i]
                            if enabled[nd]:
                                for cc in range(3):
                                    r[col[nd] + 3 * cc + x] += sign * s * w[i] * (v[cc] - G.g[nd][cc])
                                r[col[nd] + 9 + x] += sign * s * w[i]
            return np.array(rows), np.array(res)

        J, r = linearise()
        err = float(r @ r)
        last = err
        for it in range(1, 4):
            delta = np.linalg.solve(J.T @ J, -J.T @ r)
            z = 0
            for j in range(n):
                if enabled[j]:
                    G.R[j] += delta[z:z + 9].reshape(3, 3).T      # variables are the column-major rotation
                    G.t[j] += delta[z + 9:z + 12]
                    z += 12
            J, r = linearise()
            err = float(r @ r)
            if err > last or np.linalg.norm(delta) < 1e-2 or err < 1e-3 or abs(err - last) < 1e-5 * err or (it == 1 and fern_match and err > 10.0):
                break
            last = err
        out["error"] = err
        out["mean_cons"] = mean_cons()
    if (not fern_match) or (optimised and out["mean_cons"] < 0.0003 and out["error"] < 0.12):
        for i, P in enumerate(out["poses"]):                      # DeformationGraph.cpp:106-135
            t = P[:3, 3].astype(np.float64)
            wm = G.weigh(t, pose_times[i])
            newp = G.deform(wm, t)
            Rb = sum(wm[1][k] * G.R[wm[0][k]] for k in range(K))
            U, _, Vt = np.linalg.svd(Rb @ P[:3, :3].astype(np.float64))
            P[:3, :3] = U @ Vt; P[:3, 3] = newp
        moved = [G.deform(wmap[i], pool[i]) for i in range(len(pool))]
        if not fern_match:
            out["new_rel"] = [(moved[c["sid"]], np.asarray(c["target"], np.float64), c["src_time"], c["target_time"]) for c in cons if not c["relative"] and not c["pin"]]
        raw = np.zeros((n, 16))
        raw[:, :3] = G.g
        raw[:, 3:12] = np.array([G.R[j].T.reshape(9) for j in range(n)])   # column-major
        raw[:, 12:15] = G.t
        raw[:, 15] = G.time
        out["raw_graph"] = raw
        out["ok"] = True
    return out
