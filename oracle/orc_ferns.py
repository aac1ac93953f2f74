"""oracle/orc_ferns.py -- TEST INFRASTRUCTURE ONLY (see orc.h): a plain restatement of the host maths of the reference's fern data base
(EF/Ferns.cpp) for checking instancefusion_amd/host/ifx_ferns.hpp.  The fern table (positions and thresholds, EF/Ferns.cpp:65-81, drawn from
a generator the reference seeds with time(0)) is an input here; the tracker (RGBDOdometry, :558-592) is an input function.

Parity status: unpinned by the reference (no tests or fixtures for ferns there); this file and the C++ class are two independent
restatements of the same source lines and are compared with each other."""
import numpy as np

BAD = 255


class Frame:
    def __init__(self):
        self.codes = None
        self.good = 0
        self.id = 0
        self.pose = None
        self.src_time = 0
        self.rgb = self.verts = self.norms = None


class Ferns:
    def __init__(self, table, width, height, max_depth, photo_thresh, fx, fy, cx, cy, min_gap=300):
        self.table = np.asarray(table, np.int64)           # n x (x, y, r, g, b, d)
        self.num = len(self.table)
        self.w, self.h = width, height                     # fern resolution (full / 8)
        self.max_depth, self.photo_thresh = max_depth, np.float32(photo_thresh)
        self.fx, self.fy, self.cx, self.cy = (np.float32(v) for v in (fx, fy, cx, cy))   # already divided by 8
        self.ids = [[[] for _ in range(16)] for _ in range(self.num)]
        self.frames = []
        self.last_closest = -1
        self.min_gap = min_gap

    # EF/Ferns.cpp:108-131 / :204-227
    def _encode(self, img, verts):
        codes = np.full(self.num, BAD, np.uint8)
        co = np.zeros(len(self.frames), np.int64)
        good = 0
        for i, (x, y, r, g, b, d) in enumerate(self.table):
            z = verts[y, x, 2]
            if z > 0:
                pix = img[y, x]
                code = (int(pix[0] > r) << 3) | (int(pix[1] > g) << 2) | (int(pix[2] > b) << 1) | int(int(np.float32(z) * np.float32(1000.0)) > d)
                good += 1
                for j in self.ids[i][code]:
                    co[j] += 1
                codes[i] = code
        return codes, good, co

    # EF/Ferns.cpp:83-172
    def add_frame(self, img, verts, norms, pose, src_time, threshold):
        codes, good, co = self._encode(img, verts)
        minimum = np.float32(np.finfo(np.float32).max)
        if good > 0:
            for i, f in enumerate(self.frames):
                max_co = np.float32(min(good, f.good))
                dissim = np.float32(max_co - np.float32(co[i])) / max_co
                if dissim < minimum:
                    minimum = dissim
        if (minimum > np.float32(threshold) or len(self.frames) == 0) and good > 0:
            f = Frame()
            f.codes, f.good, f.id, f.pose, f.src_time = codes, good, len(self.frames), np.array(pose, np.float32), src_time
            f.rgb, f.verts, f.norms = img.copy(), verts.copy(), norms.copy()
            for i in range(self.num):
                if codes[i] != BAD:
                    self.ids[i][codes[i]].append(f.id)
            self.frames.append(f)
            return True
        return False

    # EF/Ferns.cpp:791-808
    def _block_hd_aware(self, c1, c2):
        both = (c1 != BAD) & (c2 != BAD)
        count = int(both.sum())
        val = np.float32((c1[both] == c2[both]).sum())
        return val / np.float32(count) if count else np.float32(np.nan)

    # EF/Ferns.cpp:702-775
    def _photometric(self, verts, img, est, fern_pose, fern_rgb):
        inv = np.eye(4, dtype=np.float32)
        R, t = fern_pose[:3, :3], fern_pose[:3, 3]
        for r in range(3):
            inv[r, :3] = R[:, r]
            inv[r, 3] = -(R[0, r] * t[0] + R[1, r] * t[1] + R[2, r] * t[2])
        diff = np.zeros((4, 4), np.float32)
        for r in range(4):
            for c in range(4):
                s = np.float32(0)
                for k in range(4):
                    s = np.float32(s + inv[r, k] * est[k, c])
                diff[r, c] = s
        invfx, invfy = np.float32(1.0) / self.fx, np.float32(1.0) / self.fy
        total, count = np.float32(0), 0
        for (x, y, _r, _g, _b, _d) in self.table:
            v = verts[y, x]
            if v[2] > 0 and int(v[2] * np.float32(1000.0)) < self.max_depth:
                w = [np.float32(np.float32(np.float32(diff[r, 0] * v[0] + diff[r, 1] * v[1]) + diff[r, 2] * v[2]) + diff[r, 3]) for r in range(3)]
                with np.errstate(all="ignore"):
                    fx_ = np.float32(w[0] * (np.float32(1) / invfx) / w[2] + self.cx)
                    fy_ = np.float32(w[1] * (np.float32(1) / invfy) / w[2] + self.cy)
                if not (np.isfinite(fx_) and np.isfinite(fy_)):
                    continue
                px, py = int(fx_), int(fy_)                                # float -> int truncates towards zero
                if 0 <= px < self.w and 0 <= py < self.h and fern_rgb[py, px].any():
                    total = np.float32(total + np.abs(fern_rgb[py, px].astype(np.int64) - img[y, x].astype(np.int64)).sum())
                    count += 1
        with np.errstate(all="ignore"):
            return np.float32(total) / np.float32(count)

    # EF/Ferns.cpp:174-700 (the instance-aware branch :225-556 is behind `if(false)` there)
    def find_frame(self, curr_pose, img, verts, norms, time, lost, tracker):
        self.last_closest = -1
        codes, good, co = self._encode(img, verts)
        minimum, min_id = np.float32(np.finfo(np.float32).max), -1
        with np.errstate(all="ignore"):
            for i, f in enumerate(self.frames):
                max_co = np.float32(min(good, f.good))
                dissim = np.float32(max_co - np.float32(co[i])) / max_co
                if dissim < minimum and time - f.src_time > self.min_gap:
                    minimum, min_id = dissim, i
        est = np.eye(4, dtype=np.float32)
        out = dict(candidate=min_id, dissim=minimum, photo=np.float32(0), constraints=[])
        if min_id != -1 and self._block_hd_aware(codes, self.frames[min_id].codes) > 0.3:
            f = self.frames[min_id]
            est, icp_err, icp_count = tracker(f.verts, f.norms, verts, norms, f.pose.copy())
            photo = self._photometric(verts, img, est, f.pose, f.rgb)
            out["photo"] = photo
            if icp_err < 0.0003 and icp_count > (1400 if lost else 2400) and photo < self.photo_thresh:
                self.last_closest = min_id
                step = max(self.num // 50, 1)
                for i in range(0, self.num, step):
                    x, y = self.table[i, 0], self.table[i, 1]
                    v = verts[y, x]
                    if v[2] > 0 and int(v[2] * np.float32(1000.0)) < self.max_depth:
                        def mul(m):
                            return np.array([np.float32(np.float32(np.float32(m[r, 0] * v[0] + m[r, 1] * v[1]) + m[r, 2] * v[2]) + m[r, 3] * np.float32(1)) for r in range(4)], np.float32)
                        out["constraints"].append((mul(curr_pose), mul(est)))
        out["est"] = est
        out["closest"] = self.last_closest
        return out
