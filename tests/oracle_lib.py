"""ctypes binding of the CPU oracle (oracle/liborc.so).  Test infrastructure only: nothing under
instancefusion_amd/ imports this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")


class OrcConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("time_delta", C.c_int32), ("confidence", C.c_float), ("depth_cut", C.c_float),
        ("max_depth_processed", C.c_float), ("icp_weight", C.c_float),
        ("pyramid", C.c_int32), ("fast_odom", C.c_int32), ("so3", C.c_int32),
        ("max_surfels", C.c_int32), ("device", C.c_int32), ("n_ranks", C.c_int32), ("rank", C.c_int32),
    ]


def default_config(w=640, h=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, max_surfels=1 << 20, **kw):
    d = dict(width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy, time_delta=200, confidence=10.0, depth_cut=12.0,
             max_depth_processed=20.0, icp_weight=10.0, pyramid=1, fast_odom=0, so3=1, max_surfels=max_surfels,
             device=0, n_ranks=1, rank=0)
    d.update(kw)
    return d


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORC_DIR, "liborc.so"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORC_DIR, "liborc.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        L = _lib
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_process_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_map_count.argtypes = [C.c_void_p]
        L.orc_tick.argtypes = [C.c_void_p]
        L.orc_map_download.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_map_upload.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.orc_set_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_image.restype = C.c_void_p
        L.orc_image.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_predict_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_combined_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.orc_clean.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_render_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_should_segment.argtypes = [C.c_void_p, C.c_int]
        L.orc_process_segmentation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_labels.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_instance_table.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_vote_encode.restype = C.c_float
        L.orc_vote_encode.argtypes = [C.c_int, C.c_int]
        L.orc_vote_decode.argtypes = [C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_tracker_create.restype = C.c_void_p
        L.orc_tracker_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_tracker_destroy.argtypes = [C.c_void_p]
        L.orc_tracker_init_first_rgb.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_tracker_init_model.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_tracker_init_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.orc_tracker_run.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_tracker_buffer.restype = C.c_void_p
        L.orc_tracker_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    return _lib


def usable_cores():
    """Cores this process may actually use: the scheduler affinity capped by the cgroup's CPU quota (a GPU box shows 256 logical
    CPUs and grants 16: 256 OpenMP threads on 16 CPUs run 30x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def set_threads(n):
    """OpenMP threads the oracle's parallel loops use from now on (results do not depend on it)."""
    lib().orc_set_threads.argtypes = [C.c_int]
    lib().orc_set_threads(int(n))


def ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


_IMG_SPECS = {
    "ids_after": (np.int32, 1), "ids_tmp": (np.int32, 1), "index": (np.uint32, 1), "index_vc": (np.float32, 4),
    "index_ct": (np.float32, 4), "index_nr": (np.float32, 4), "pred_vertex": (np.float32, 4),
    "pred_normal": (np.float32, 4), "pred_image": (np.uint8, 4), "pred_inst": (np.uint8, 4), "pred_time": (np.uint16, 1),
    "fill_vertex": (np.float32, 4), "fill_normal": (np.float32, 4), "fill_image": (np.uint8, 4),
    "old_vertex": (np.float32, 4), "old_normal": (np.float32, 4), "old_image": (np.uint8, 4), "old_time": (np.uint16, 1),
    "depth_filtered": (np.uint16, 1), "depth_metric": (np.float32, 1), "depth_metric_filtered": (np.float32, 1),
}


class Oracle:
    """The whole-pipeline oracle object (orc_t)."""

    def __init__(self, **cfg):
        self.cfgd = default_config(**cfg)
        self.cfg = OrcConfig(**self.cfgd)
        self.L = lib()
        self.h = self.L.orc_create(C.byref(self.cfg))
        self.w_, self.h_ = self.cfgd["width"], self.cfgd["height"]

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, rgb, depth, ts=0, in_pose=None, weight_mult=1.0, bootstrap=False):
        if bootstrap:
            self.L.orc_set_bootstrap.argtypes = [C.c_void_p, C.c_int]
            self.L.orc_set_bootstrap(self.h, 1)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        out = np.zeros(16, np.float32)
        ip = None if in_pose is None else np.ascontiguousarray(in_pose, np.float32).reshape(16)
        self.L.orc_process_frame(self.h, ptr(rgb), ptr(depth), ts, ptr(ip), weight_mult, ptr(out))
        return out.reshape(4, 4)

    @property
    def count(self):
        return self.L.orc_map_count(self.h)

    @property
    def tick(self):
        return self.L.orc_tick(self.h)

    def download(self):
        n = self.count
        d = dict(pc=np.zeros((n, 4), np.float32), nr=np.zeros((n, 4), np.float32), col=np.zeros((n, 2), np.float32),
                 tm=np.zeros((n, 2), np.float32), ic=np.zeros((n, 4), np.float32), votes=np.zeros((n, 48), np.float32))
        self.L.orc_map_download(self.h, ptr(d["pc"]), ptr(d["nr"]), ptr(d["col"]), ptr(d["tm"]), ptr(d["ic"]), ptr(d["votes"]))
        return d

    def upload(self, m):
        n = m["pc"].shape[0]
        a = {k: np.ascontiguousarray(m[k], np.float32) for k in ("pc", "nr", "col", "tm", "ic", "votes")}
        self.L.orc_map_upload(self.h, n, ptr(a["pc"]), ptr(a["nr"]), ptr(a["col"]), ptr(a["tm"]), ptr(a["ic"]), ptr(a["votes"]))

    def tracker_diag(self):
        out = np.zeros(8, np.float32)
        self.L.orc_tracker_diag.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_tracker_diag(self.h, ptr(out))
        return out

    def stage_ms(self, reset=False):
        """wall-clock per stage since the last reset: track (preprocessing + tracker) | fuse (map passes) | instance"""
        out = (C.c_double * 3)()
        self.L.orc_stage_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_stage_ms(self.h, out, int(reset))
        return dict(track=out[0], fuse=out[1], instance=out[2])

    def get_pose(self):
        out = np.zeros(16, np.float32)
        self.L.orc_get_pose.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_get_pose(self.h, ptr(out))
        return out.reshape(4, 4)

    def set_pose(self, pose, tick):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_set_pose(self.h, ptr(p), tick)

    def image(self, name):
        dt, ch = _IMG_SPECS[name]
        p = self.L.orc_image(self.h, name.encode())
        n = self.w_ * self.h_ * ch
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
        a = np.frombuffer(buf, dtype=dt).copy()
        return a.reshape(self.h_, self.w_, ch) if ch > 1 else a.reshape(self.h_, self.w_)

    # ---- local loop-closure detection (EF/ElasticFusion.cpp:453-566)
    def set_loop_closure(self, enable=True, count_thresh=35000, err_thresh=5e-5, cov_thresh=1e-5):
        self.L.orc_set_loop_closure.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        self.L.orc_set_loop_closure(self.h, int(enable), int(count_thresh), err_thresh, cov_thresh)

    def loop_closure_diag(self):
        out = np.zeros(24, np.float32)
        self.L.orc_loop_closure_diag.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_loop_closure_diag(self.h, ptr(out))
        return dict(ran=bool(out[0]), inactive_pixels=int(out[1]), icp_error=float(out[2]), icp_count=float(out[3]), cov_ok=bool(out[4]),
                    accepted=bool(out[5]), est_pose=out[6:22].reshape(4, 4).copy(), cov_max=float(out[22]), candidates=int(out[23]))

    # ---- deformation hooks (orc_deform.c)
    def set_loop_closure_callback(self, fn):
        """fn(oracle, lc_dict) is called inside process_frame when a candidate is accepted (None removes it)."""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_void_p)
        self.L.orc_set_loop_closure_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if fn is None:
            self._cb = None
            self.L.orc_set_loop_closure_callback(self.h, None, None)
            return

        def tramp(_h, lc, _user):
            fn(self, np.ctypeslib.as_array(lc, shape=(24,)).copy())
            return 0

        self._cb = CB(tramp)
        self.L.orc_set_loop_closure_callback(self.h, C.cast(self._cb, C.c_void_p), None)

    def sample_graph_model(self, max_n=4096):
        out = np.zeros((max_n, 4), np.float32)
        self.L.orc_sample_graph_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        n = self.L.orc_sample_graph_model(self.h, ptr(out), max_n)
        return out[:n].copy()

    def loop_closure_constraints(self, max_n=4096):
        src, dst, tm = np.zeros((max_n, 3), np.float32), np.zeros((max_n, 3), np.float32), np.zeros(max_n, np.int32)
        self.L.orc_loop_closure_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        n = self.L.orc_loop_closure_constraints(self.h, ptr(src), ptr(dst), ptr(tm), max_n)
        return src[:n].copy(), dst[:n].copy(), tm[:n].copy()

    def set_deformation(self, graph16, is_fern=False):
        g = np.ascontiguousarray(graph16, np.float32).reshape(-1, 16)
        self.L.orc_set_deformation.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.L.orc_set_deformation(self.h, ptr(g), g.shape[0], int(is_fern))

    def set_fern_callback(self, fn):
        """fn(oracle) -> truthy when a graph was produced (EF/ElasticFusion.cpp:457-514)"""
        self.L.orc_set_fern_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if fn is None:
            self._fcb = None
            self.L.orc_set_fern_callback(self.h, None, None)
            return
        self._fcb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(lambda _h, _u: 1 if fn(self) else 0)
        self.L.orc_set_fern_callback(self.h, C.cast(self._fcb, C.c_void_p), None)

    def fern_frame(self):
        rw, rh = self.w_ // 8, self.h_ // 8
        img, inst = np.zeros((rh, rw, 3), np.uint8), np.zeros((rh, rw, 3), np.uint8)
        v, n = np.zeros((rh, rw, 4), np.float32), np.zeros((rh, rw, 4), np.float32)
        self.L.orc_fern_frame.argtypes = [C.c_void_p] * 5
        self.L.orc_fern_frame(self.h, ptr(img), ptr(v), ptr(n), ptr(inst))
        return img, v, n, inst

    def adopt_pose(self, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_adopt_pose.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_adopt_pose(self.h, ptr(p))

    def adopt_estimated_pose(self):
        self.L.orc_adopt_estimated_pose.argtypes = [C.c_void_p]
        self.L.orc_adopt_estimated_pose(self.h)

    def predict_indices(self, pose, time):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_predict_indices(self.h, ptr(p), time)

    def combined_predict(self, pose, time, max_time):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_stage_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.L.orc_stage_predict(self.h, ptr(p), time, max_time)   # combinedPredict + FillIn, what ifx_combined_predict resolves in one kernel

    def fuse(self, pose, time, weighting):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_fuse(self.h, ptr(p), time, weighting)

    def clean(self, pose, time):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_clean(self.h, ptr(p), time)

    def render_ids(self, pose, mode=0):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self.L.orc_render_ids(self.h, ptr(p), mode)
        return self.image("ids_tmp")

    def set_frame(self, rgb, depth):
        """upload + preprocess only (map, pose and tick untouched)"""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        self.L.orc_set_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.orc_set_frame(self.h, ptr(rgb), ptr(depth))

    def set_ids_after(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self.L.orc_set_ids_after.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_set_ids_after(self.h, ptr(ids))

    def should_segment(self, frame):
        return bool(self.L.orc_should_segment(self.h, frame))

    def process_segmentation(self, rgb, depth, masks, class_ids, frame, flags=0):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        masks = np.ascontiguousarray(masks, np.uint8)
        cls = np.ascontiguousarray(class_ids, np.int32)
        return self.L.orc_process_segmentation(self.h, ptr(rgb), ptr(depth), ptr(masks), ptr(cls), masks.shape[0], frame, flags)

    def knn_vote(self, with_neighbours=False):
        nbr = np.full((max(self.count, 1), 10), -1, np.int32) if with_neighbours else None
        self.L.orc_knn_vote.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_knn_vote(self.h, ptr(nbr) if with_neighbours else None)
        return nbr[: self.count] if with_neighbours else None

    def mask_geometric_filter(self, model_depth, masks, ori, unavailable=None):
        depth = np.ascontiguousarray(model_depth, np.uint16)
        masks = np.ascontiguousarray(masks, np.uint8).copy()
        ori = np.ascontiguousarray(ori, np.uint8)
        un = np.zeros(masks.shape[0], np.uint8) if unavailable is None else np.ascontiguousarray(unavailable, np.uint8).copy()
        self.L.orc_test_mask_geometric_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.L.orc_test_mask_geometric_filter(self.h, ptr(depth), ptr(masks), ptr(ori), masks.shape[0], ptr(un))
        return masks, un

    # ---- superpixel refinement stages (orc_slic.c)
    def slic_segment(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        seg = np.zeros((self.cfg.height, self.cfg.width), np.int32)
        self.L.orc_slic_segment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        n = self.L.orc_slic_segment(self.h, ptr(rgb), ptr(seg))
        return seg, n

    def merge_superpixels(self, depth, seg):
        depth = np.ascontiguousarray(depth, np.uint16)
        seg = np.ascontiguousarray(seg, np.int32).copy()
        fin = np.zeros_like(seg)
        spn = seg.size // 256
        info = np.zeros((spn, 30), np.float32)
        self.L.orc_merge_superpixels.argtypes = [C.c_void_p] * 5
        self.L.orc_merge_superpixels(self.h, ptr(depth), ptr(seg), ptr(fin), ptr(info))
        return seg, fin, info

    def mask_superpixel_filter(self, fin, masks):
        fin = np.ascontiguousarray(fin, np.int32)
        masks = np.ascontiguousarray(masks, np.uint8).copy()
        self.L.orc_mask_superpixel_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_mask_superpixel_filter(self.h, ptr(fin), ptr(masks), masks.shape[0])
        return masks

    def set_instance_gt(self, gt):
        self.L.orc_set_instance_gt.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_set_instance_gt(self.h, None if gt is None else ptr(np.ascontiguousarray(gt, np.uint8)))

    def precision_recall(self):
        a, b, c = np.zeros(96, np.int32), np.zeros(256, np.int32), np.zeros((256, 96), np.int32)
        self.L.orc_precision_recall.argtypes = [C.c_void_p] * 4
        self.L.orc_precision_recall(self.h, ptr(a), ptr(b), ptr(c))
        return a, b, c

    def render_project_map(self):
        out = np.zeros((self.h_, self.w_, 4), np.float32)
        self.L.orc_render_project_map.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_render_project_map(self.h, ptr(out))
        return out

    def map_bounding_boxes(self, bbox_type=True, ratio=1000000.0):
        boxes, gn, gc, im, gv = np.zeros((96, 6), np.float32), np.zeros(3, np.float32), np.zeros((4, 4), np.float32), np.zeros((96, 4, 4), np.float32), np.zeros(648, np.int32)
        self.L.orc_map_bounding_boxes.argtypes = [C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 5
        self.L.orc_map_bounding_boxes(self.h, int(bool(bbox_type)), ratio, ptr(boxes), ptr(gn), ptr(gc), ptr(im), ptr(gv))
        return boxes, gn, gc, im, gv

    def instance_point_cloud(self, inst=-1, bbox_type=True, max_records=1 << 20):
        counts = np.zeros(96, np.int32)
        out = np.zeros((max(max_records, 1), 10), np.float32)
        self.L.orc_instance_point_cloud.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        n = self.L.orc_instance_point_cloud(self.h, int(bool(bbox_type)), ptr(counts), int(inst), ptr(out), max_records)
        return counts, out[:n]

    def labels(self):
        out = np.zeros(self.count, np.int32)
        self.L.orc_labels(self.h, ptr(out))
        return out

    def instance_table(self):
        out = np.zeros(96, np.int32)
        self.L.orc_instance_table(self.h, ptr(out))
        return out
