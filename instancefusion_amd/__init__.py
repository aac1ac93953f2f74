"""instancefusion_amd -- host-side Python mirror of the reference's interface over libifx.so.

The product is the C-ABI shared library ``instancefusion_amd/libifx.so`` (HIP kernels for gfx950,
declared in ``include/ifx_c_api.h``).  This package only binds it with ctypes and mirrors the
reference's classes for the hot path:

* :class:`ElasticFusion`  -- ``ElasticFusion::processFrame`` / ``ElasticFusionInterface``
  (elasticfusionpublic/Core/src/ElasticFusion.h:75-82, src/map_interface/ElasticFusionInterface.h:55-130)
* :class:`InstanceFusion` -- ``InstanceFusion::whetherDoSegmentation`` / ``ProcessSegmentation``
  (src/Core/InstanceFusion.h:72-107) with the Mask-RCNN bridge replaced by replayed masks.

There is no CPU fallback: importing works anywhere (so the symbols can be checked), but creating a
handle without a MI355X raises :class:`IfxError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IFX_LIB") or os.path.join(_HERE, "libifx.so")   # IFX_LIB: an experimental build of the same library (tools/bench_variants.sh)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ifx_c_api.h")

NUM_INSTANCES = 96
VOTE_FLOATS = 48


class IfxError(RuntimeError):
    pass


class IfxConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("time_delta", C.c_int32), ("confidence", C.c_float), ("depth_cut", C.c_float),
        ("max_depth_processed", C.c_float), ("icp_weight", C.c_float),
        ("pyramid", C.c_int32), ("fast_odom", C.c_int32), ("so3", C.c_int32),
        ("max_surfels", C.c_int32), ("device", C.c_int32), ("n_ranks", C.c_int32), ("rank", C.c_int32),
    ]


class Pyramids(C.Structure):
    """ifx_pyramids: device pointers of the caller's output buffers, three levels each (0 = skipped)"""
    _fields_ = [(n, C.c_void_p * 3) for n in ("depth", "vmap_curr", "nmap_curr", "next_img", "didx", "didy", "vmap_g_prev", "nmap_g_prev", "last_depth", "last_img", "cloud")]


class SoaView(C.Structure):
    _fields_ = [("count", C.c_int32), ("capacity", C.c_int32), ("d_pos_conf", C.c_void_p), ("d_norm_rad", C.c_void_p),
                ("d_color", C.c_void_p), ("d_times", C.c_void_p), ("d_img_corr", C.c_void_p), ("d_votes", C.c_void_p)]


def default_config(w=640, h=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, max_surfels=1 << 20, **kw):
    """The constants the reference runs with (src/map_interface/ElasticFusionInterface.cpp:43-45, src/main.cpp:46-47)."""
    d = dict(width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy, time_delta=200, confidence=10.0, depth_cut=12.0,
             max_depth_processed=20.0, icp_weight=10.0, pyramid=1, fast_odom=0, so3=1, max_surfels=max_surfels,
             device=0, n_ranks=1, rank=0)
    d.update(kw)
    return d


_lib = None

_P = C.c_void_p
_SIGS = {
    "ifx_create": (C.c_int, [C.POINTER(IfxConfig), C.POINTER(_P)]),
    "ifx_destroy": (None, [_P]),
    "ifx_last_error": (C.c_char_p, [_P]),
    "ifx_global_error": (C.c_char_p, []),
    "ifx_process_frame": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_float, _P]),
    "ifx_process_frame_ex": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, C.c_float, C.c_int, _P]),
    "ifx_enqueue_frame_device": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_float]),
    "ifx_set_shard": (C.c_int, [_P, C.c_int, C.c_int]),
    "ifx_sharded_frame_phase": (C.c_int, [_P, C.c_int, _P, _P]),
    "ifx_key_images": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "ifx_stream_handles": (C.c_int, [_P, _P, _P]),
    "ifx_owner_frame_phase": (C.c_int, [_P, C.c_int, _P, _P]),
    "ifx_owner_exchange": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int]),
    "ifx_owner_of": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "ifx_owner_segmentation_begin": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "ifx_owner_segmentation_resume": (C.c_int, [_P]),
    "ifx_owner_ids_begin": (C.c_int, [_P]),
    "ifx_owner_ids_resume": (C.c_int, [_P]),
    "ifx_owner_knn_export": (C.c_int, [_P, _P, _P, _P]),
    "ifx_owner_knn_vote": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "ifx_owner_predict_phase": (C.c_int, [_P, C.c_int]),
    "ifx_camera_count": (C.c_int, [_P, C.c_int]),
    "ifx_camera_select": (C.c_int, [_P, C.c_int]),
    "ifx_owner_set_frame_pose": (C.c_int, [_P, _P]),
    "ifx_owner_set_tracking_rank": (C.c_int, [_P, C.c_int]),
    "ifx_owner_track_ahead": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "ifx_comm_unique_id": (C.c_int, [_P]),
    "ifx_owner_init_comm": (C.c_int, [_P, _P]),
    "ifx_owner_set_comm": (C.c_int, [_P, _P]),
    "ifx_owner_process_frame_device": (C.c_int, [_P, _P, _P, C.c_int64]),
    "ifx_owner_process_frame": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "ifx_owner_predict": (C.c_int, [_P]),
    "ifx_owner_process_segmentation": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "ifx_owner_knn_vote_colour": (C.c_int, [_P]),
    "ifx_owner_exchange_stats": (C.c_int, [_P, _P, C.c_int]),
    "ifx_owner_comm_ranks": (C.c_int, [_P]),
    "ifx_map_seq": (C.c_int, [_P, _P, C.c_int]),
    "ifx_prefetch_frame_device": (C.c_int, [_P, _P, _P]),
    "ifx_hint_next_frame_device": (C.c_int, [_P, _P, _P]),
    "ifx_hint_next_frame": (C.c_int, [_P, _P, _P]),
    "ifx_lookahead_stats": (C.c_int, [_P, _P, C.c_int]),
    "ifx_view_list_stats": (C.c_int, [_P, _P]),
    "ifx_map_bounding_boxes": (C.c_int, [_P, C.c_int, C.c_float, _P, _P, _P, _P, _P]),
    "ifx_instance_point_cloud": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int]),
    "ifx_sync": (C.c_int, [_P]),
    "ifx_get_pose": (C.c_int, [_P, _P]),
    "ifx_tick": (C.c_int, [_P]),
    "ifx_trajectory": (C.c_int, [_P, _P, C.c_int]),
    "ifx_tracker_diag": (C.c_int, [_P, _P]),
    "ifx_tracker_fallbacks": (C.c_int, [_P]),
    "ifx_build_pyramids": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(Pyramids)]),
    "ifx_tracker_range_exceeded": (C.c_int, [_P]),
    "ifx_hot_records_stale": (C.c_int, [_P]),
    "ifx_set_loop_closure": (C.c_int, [_P, C.c_int, C.c_int, C.c_float, C.c_float]),
    "ifx_loop_closure_diag": (C.c_int, [_P, _P]),
    "ifx_set_loop_closure_callback": (C.c_int, [_P, _P, _P]),
    "ifx_sample_graph_model": (C.c_int, [_P, _P, C.c_int]),
    "ifx_loop_closure_constraints": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "ifx_set_deformation": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "ifx_adopt_estimated_pose": (C.c_int, [_P]),
    "ifx_set_fern_callback": (C.c_int, [_P, _P, _P]),
    "ifx_adopt_pose": (C.c_int, [_P, _P]),
    "ifx_fern_frame": (C.c_int, [_P, _P, _P, _P, _P]),
    "ifx_fern_frame_async": (C.c_int, [_P]),
    "ifx_fern_frame_fetch": (C.c_int, [_P, _P, _P, _P, _P]),
    "ifx_track_maps": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ifx_map_view": (C.c_int, [_P, C.POINTER(SoaView)]),
    "ifx_map_count": (C.c_int, [_P]),
    "ifx_map_slots": (C.c_int, [_P]),
    "ifx_map_download": (C.c_int, [_P, C.c_int] + [_P] * 6),
    "ifx_map_upload": (C.c_int, [_P, C.c_int] + [_P] * 6),
    "ifx_set_pose": (C.c_int, [_P, _P, C.c_int]),
    "ifx_compact": (C.c_int, [_P]),
    "ifx_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "ifx_ids_after": (_P, [_P]),
    "ifx_image_download": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "ifx_predict_indices": (C.c_int, [_P, _P, C.c_int]),
    "ifx_combined_predict": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "ifx_fuse": (C.c_int, [_P, _P, C.c_int, C.c_float]),
    "ifx_clean": (C.c_int, [_P, _P, C.c_int]),
    "ifx_render_ids": (C.c_int, [_P, _P, C.c_int]),
    "ifx_set_frame": (C.c_int, [_P, _P, _P]),
    "ifx_icp_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, C.c_float, C.c_float, C.c_int, C.c_int, _P]),
    "ifx_rgb_residual": (C.c_int, [_P, C.c_float, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_int, C.c_int, _P, _P]),
    "ifx_rgb_step": (C.c_int, [_P, _P, C.c_float, _P, C.c_float, C.c_float, _P, _P, C.c_float, C.c_int, C.c_int, _P]),
    "ifx_so3_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "ifx_track_pair": (C.c_int, [_P] * 9),
    "ifx_tracker_buffer_download": (C.c_int, [_P, C.c_char_p, C.c_int, _P, C.c_int64]),
    "ifx_should_segment": (C.c_int, [_P, C.c_int]),
    "ifx_process_segmentation": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "ifx_labels": (C.c_int, [_P, _P, C.c_int]),
    "ifx_render_project_map": (C.c_int, [_P, _P, _P]),
    "ifx_set_instance_gt": (C.c_int, [_P, _P]),
    "ifx_precision_recall": (C.c_int, [_P, _P, _P, _P]),
    "ifx_instance_table": (C.c_int, [_P, _P]),
    "ifx_loop_closure_instance_table": (C.c_int, [_P, _P]),
    "ifx_mask_clean_overlap": (C.c_int, [_P, _P, C.c_int]),
    "ifx_mask_geometric_filter": (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    "ifx_knn_vote_colour": (C.c_int, [_P, _P, C.c_int]),
    "ifx_slic_segment": (C.c_int, [_P, _P, _P]),
    "ifx_merge_superpixels": (C.c_int, [_P, _P, _P, _P, _P]),
    "ifx_mask_superpixel_filter": (C.c_int, [_P, _P, _P, C.c_int]),
    "ifx_stage_ms": (C.c_int, [_P, _P, C.c_int]),
    "ifx_superpixel_ahead_stats": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "ifx_kernel_ms": (C.c_int, [_P, C.c_char_p, _P, _P]),
}


def lib():
    """Loads libifx.so (built by ``__graft_entry__.build()`` / ``make -C instancefusion_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IfxError(f"{LIB_PATH} is missing: build it with `make -C instancefusion_amd/csrc` "
                           "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)  # raises AttributeError when a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


_IMG_SPECS = {
    "ids_after": (np.int32, 1), "ids_tmp": (np.int32, 1), "index": (np.uint32, 1), "index_vc": (np.float32, 4),
    "index_ct": (np.float32, 4), "index_nr": (np.float32, 4), "pred_vertex": (np.float32, 4),
    "pred_normal": (np.float32, 4), "pred_image": (np.uint8, 4), "pred_inst": (np.uint8, 4), "pred_time": (np.uint16, 1),
    "fill_vertex": (np.float32, 4), "fill_normal": (np.float32, 4), "fill_image": (np.uint8, 4),
    "depth_filtered": (np.uint16, 1), "depth_metric": (np.float32, 1), "depth_metric_filtered": (np.float32, 1),
    "old_vertex": (np.float32, 4), "old_normal": (np.float32, 4), "old_image": (np.uint8, 4), "old_time": (np.uint16, 1),
    "act_vertex": (np.float32, 4), "act_normal": (np.float32, 4), "act_image": (np.uint8, 4),
}
_TRK_SPECS = {
    "vmap_curr": (np.float32, 3, True), "nmap_curr": (np.float32, 3, True), "vmap_prev": (np.float32, 3, True),
    "nmap_prev": (np.float32, 3, True), "last_depth": (np.float32, 1, False), "next_depth": (np.float32, 1, False),
    "last_img": (np.uint8, 1, False), "next_img": (np.uint8, 1, False), "lastnext_img": (np.uint8, 1, False),
    "didx": (np.int16, 1, False), "didy": (np.int16, 1, False), "cloud": (np.float32, 3, False),
    "depth_tmp": (np.uint16, 1, False),
}
CORRES_DTYPE = np.dtype([("zx", np.int16), ("zy", np.int16), ("diff", np.float32)])


class ElasticFusion:
    """Mirror of ``ElasticFusion`` / ``ElasticFusionInterface`` for the hot path."""

    def __init__(self, **cfg):
        self.cfgd = default_config(**cfg)
        self.cfg = IfxConfig(**self.cfgd)
        self.L = lib()
        self.w, self.h = self.cfgd["width"], self.cfgd["height"]
        hp = _P()
        r = self.L.ifx_create(C.byref(self.cfg), C.byref(hp))
        if r != 0:
            raise IfxError(f"ifx_create failed ({r}): {self.L.ifx_global_error().decode()}")
        self.handle = hp

    # -- life cycle
    def close(self):
        if getattr(self, "handle", None):
            self.L.ifx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r, what):
        if r < 0:
            raise IfxError(f"{what} failed ({r}): {self.L.ifx_last_error(self.handle).decode()}")
        return r

    def set_option(self, name, value):
        self._chk(self.L.ifx_set_option(self.handle, name.encode(), int(value)), "ifx_set_option")

    # -- frame entry (ElasticFusion::processFrame)
    def set_instance_gt(self, gt):
        """instanceGT of ElasticFusion::processFrame for the frames that follow (H x W uint8, None: off)."""
        self._chk(self.L.ifx_set_instance_gt(self.handle, None if gt is None else _ptr(np.ascontiguousarray(gt, np.uint8))), "ifx_set_instance_gt")

    def processFrame(self, rgb, depth, timestamp=0, smallInstanceTable=None, instanceGT=None, inPose=None, weightMultiplier=1.0, bootstrap=False):
        """ElasticFusion::processFrame (EF/ElasticFusion.h:75-82), same argument order and meaning."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        assert rgb.size == self.w * self.h * 3 and depth.size == self.w * self.h
        if instanceGT is not None:
            self.set_instance_gt(instanceGT)
        out = np.zeros(16, np.float32)
        ip = None if inPose is None else np.ascontiguousarray(inPose, np.float32).reshape(16)
        tab = None if smallInstanceTable is None else np.ascontiguousarray(smallInstanceTable, np.int32)
        self._chk(self.L.ifx_process_frame_ex(self.handle, _ptr(rgb), _ptr(depth), int(timestamp), _ptr(tab), _ptr(ip), float(weightMultiplier), int(bool(bootstrap)), _ptr(out)),
                  "ifx_process_frame_ex")
        return out.reshape(4, 4)

    process_frame = processFrame

    def hint_next_frame(self, rgb, depth):
        """Announce the frame after the one about to be processed, host arrays (see ifx_hint_next_frame): pass the SAME arrays to the next processFrame."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        assert rgb.size == self.w * self.h * 3 and depth.size == self.w * self.h
        self._chk(self.L.ifx_hint_next_frame(self.handle, _ptr(rgb), _ptr(depth)), "ifx_hint_next_frame")

    def enqueue_frame_device(self, d_rgb_ptr: int, d_depth_ptr: int, timestamp=0):
        self._chk(self.L.ifx_enqueue_frame_device(self.handle, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr), int(timestamp), None, 1.0),
                  "ifx_enqueue_frame_device")

    def prefetch_frame_device(self, d_rgb_ptr: int, d_depth_ptr: int):
        """One-frame look-ahead: frame side of the NEXT frame on the side stream (see ifx_prefetch_frame_device)."""
        self._chk(self.L.ifx_prefetch_frame_device(self.handle, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_prefetch_frame_device")

    def hint_next_frame_device(self, d_rgb_ptr: int, d_depth_ptr: int):
        """Announce the frame after the one about to be enqueued (see ifx_hint_next_frame_device)."""
        self._chk(self.L.ifx_hint_next_frame_device(self.handle, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_hint_next_frame_device")

    def camera_count(self, k):
        self._chk(self.L.ifx_camera_count(self.handle, int(k)), "ifx_camera_count")

    def camera_select(self, c):
        self._chk(self.L.ifx_camera_select(self.handle, int(c)), "ifx_camera_select")

    def owner_set_frame_pose(self, pose):
        p = None if pose is None else np.ascontiguousarray(pose, np.float32).reshape(16)
        self._chk(self.L.ifx_owner_set_frame_pose(self.handle, _ptr(p)), "ifx_owner_set_frame_pose")

    def owner_track_ahead(self, cam, tracking_rank, d_rgb_ptr=0, d_depth_ptr=0):
        """ifx_owner_track_ahead: rank `tracking_rank` runs the tracker of camera cam's NEXT frame now, from the camera's parked context (cam = -1: frames served that way so far)"""
        return self._chk(self.L.ifx_owner_track_ahead(self.handle, int(cam), int(tracking_rank), C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_owner_track_ahead")

    def owner_frame_phase(self, phase, d_rgb_ptr=0, d_depth_ptr=0):
        """ifx_owner_frame_phase: one phase of a sharded map's frame (the exchanges between the phases are the caller's, or ifx_comm.hip's)"""
        return self._chk(self.L.ifx_owner_frame_phase(self.handle, int(phase), C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_owner_frame_phase")

    def owner_set_tracking_rank(self, r):
        self._chk(self.L.ifx_owner_set_tracking_rank(self.handle, int(r)), "ifx_owner_set_tracking_rank")

    def view_list_stats(self):
        out = np.zeros(4, np.int32)
        self._chk(self.L.ifx_view_list_stats(self.handle, _ptr(out)), "ifx_view_list_stats")
        return dict(window=int(out[0]), outside=int(out[1]), scans=int(out[2]), age=int(out[3]))

    def sync(self):
        self._chk(self.L.ifx_sync(self.handle), "ifx_sync")

    def getCurrPose(self):
        out = np.zeros(16, np.float32)
        self._chk(self.L.ifx_get_pose(self.handle, _ptr(out)), "ifx_get_pose")
        return out.reshape(4, 4)

    @property
    def tick(self):
        return self.L.ifx_tick(self.handle)

    def trajectory(self, max_frames=1 << 16):
        out = np.zeros((max_frames, 16), np.float32)
        n = self._chk(self.L.ifx_trajectory(self.handle, _ptr(out), max_frames), "ifx_trajectory")
        return out[:n].reshape(n, 4, 4).copy()

    def tracker_diag(self):
        out = np.zeros(8, np.float32)
        self._chk(self.L.ifx_tracker_diag(self.handle, _ptr(out)), "ifx_tracker_diag")
        return out

    def tracker_fallbacks(self):
        """pyramid levels the persistent Gauss-Newton kernel handed to its one-workgroup fallback so far (0 in a healthy run)"""
        return self._chk(self.L.ifx_tracker_fallbacks(self.handle), "ifx_tracker_fallbacks")

    def tracker_range_exceeded(self):
        """reductions whose totals left the exact range of the tracker's sums so far (0: every pose is order-independent and equals the oracle's)"""
        return self._chk(self.L.ifx_tracker_range_exceeded(self.handle), "ifx_tracker_range_exceeded")

    # -- local loop-closure detection (closeLoops, countThresh, errThresh, covThresh of the reference constructor)
    def set_loop_closure(self, enable=True, count_thresh=35000, err_thresh=5e-5, cov_thresh=1e-5):
        self._chk(self.L.ifx_set_loop_closure(self.handle, int(enable), int(count_thresh), err_thresh, cov_thresh), "ifx_set_loop_closure")

    def loop_closure_diag(self):
        out = np.zeros(24, np.float32)
        self._chk(self.L.ifx_loop_closure_diag(self.handle, _ptr(out)), "ifx_loop_closure_diag")
        return dict(ran=bool(out[0]), inactive_pixels=int(out[1]), icp_error=float(out[2]), icp_count=float(out[3]), cov_ok=bool(out[4]),
                    accepted=bool(out[5]), est_pose=out[6:22].reshape(4, 4).copy(), cov_max=float(out[22]), candidates=int(out[23]))

    # -- hooks of the deformation an accepted candidate triggers (the graph optimisation itself is the caller's)
    def set_loop_closure_callback(self, fn):
        """fn(ef, lc24) runs inside processFrame when the frame's candidate was accepted (None removes it)."""
        if fn is None:
            self._lc_cb = None
            self._chk(self.L.ifx_set_loop_closure_callback(self.handle, None, None), "ifx_set_loop_closure_callback")
            return

        def tramp(_h, lc, _user):
            try:
                fn(self, np.ctypeslib.as_array(lc, shape=(24,)).copy())
                return 0
            except Exception:      # never unwind through the C frames
                import traceback

                traceback.print_exc()
                return -1

        self._lc_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_void_p)(tramp)
        self._chk(self.L.ifx_set_loop_closure_callback(self.handle, C.cast(self._lc_cb, C.c_void_p), None), "ifx_set_loop_closure_callback")

    def sample_graph_model(self, max_n=4096):
        out = np.zeros((max_n, 4), np.float32)
        n = self._chk(self.L.ifx_sample_graph_model(self.handle, _ptr(out), max_n), "ifx_sample_graph_model")
        return out[:n].copy()

    def loop_closure_constraints(self, max_n=4096):
        src, dst, tm = np.zeros((max_n, 3), np.float32), np.zeros((max_n, 3), np.float32), np.zeros(max_n, np.int32)
        n = self._chk(self.L.ifx_loop_closure_constraints(self.handle, _ptr(src), _ptr(dst), _ptr(tm), max_n), "ifx_loop_closure_constraints")
        return src[:n].copy(), dst[:n].copy(), tm[:n].copy()

    def set_deformation(self, graph16, is_fern=False):
        g = np.ascontiguousarray(graph16, np.float32).reshape(-1, 16)
        self._chk(self.L.ifx_set_deformation(self.handle, _ptr(g), g.shape[0], int(is_fern)), "ifx_set_deformation")

    def adopt_estimated_pose(self):
        self._chk(self.L.ifx_adopt_estimated_pose(self.handle), "ifx_adopt_estimated_pose")

    # -- GPU contacts of the fern data base (EF/Ferns.cpp)
    def set_fern_callback(self, fn):
        """fn(ef) -> truthy when a graph was produced; runs inside processFrame every frame after predict() at the tracked pose
        (Ferns::findFrame + global deformation, EF/ElasticFusion.cpp:457-514).  None removes it."""
        if fn is None:
            self._fern_cb = None
            self._chk(self.L.ifx_set_fern_callback(self.handle, None, None), "ifx_set_fern_callback")
            return

        def tramp(_h, _user):
            try:
                return 1 if fn(self) else 0
            except Exception:      # never unwind through the C frames
                import traceback

                traceback.print_exc()
                return -1

        self._fern_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(tramp)
        self._chk(self.L.ifx_set_fern_callback(self.handle, C.cast(self._fern_cb, C.c_void_p), None), "ifx_set_fern_callback")

    def adopt_pose(self, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self._chk(self.L.ifx_adopt_pose(self.handle, _ptr(p)), "ifx_adopt_pose")

    def fern_frame(self):
        """fill-in image / vertex / normal and instance render at (w/8) x (h/8): what Ferns::addFrame / findFrame read back"""
        rw, rh = self.w // 8, self.h // 8
        img, inst = np.zeros((rh, rw, 3), np.uint8), np.zeros((rh, rw, 3), np.uint8)
        v, n = np.zeros((rh, rw, 4), np.float32), np.zeros((rh, rw, 4), np.float32)
        self._chk(self.L.ifx_fern_frame(self.handle, _ptr(img), _ptr(v), _ptr(n), _ptr(inst)), "ifx_fern_frame")
        return img, v, n, inst

    def fern_frame_async(self):
        self._chk(self.L.ifx_fern_frame_async(self.handle), "ifx_fern_frame_async")

    def fern_frame_fetch(self):
        rw, rh = self.w // 8, self.h // 8
        img, inst = np.zeros((rh, rw, 3), np.uint8), np.zeros((rh, rw, 3), np.uint8)
        v, n = np.zeros((rh, rw, 4), np.float32), np.zeros((rh, rw, 4), np.float32)
        self._chk(self.L.ifx_fern_frame_fetch(self.handle, _ptr(img), _ptr(v), _ptr(n), _ptr(inst)), "ifx_fern_frame_fetch")
        return img, v, n, inst

    def track_maps(self, model_v4, model_n4, cur_v4, cur_n4, pose, model_rgba=None, cur_rgba=None):
        a = [np.ascontiguousarray(x, np.float32) for x in (model_v4, model_n4, cur_v4, cur_n4)]
        im = [None if x is None else np.ascontiguousarray(x, np.uint8) for x in (model_rgba, cur_rgba)]
        p = np.ascontiguousarray(pose, np.float32).reshape(16).copy()
        diag = np.zeros(8, np.float32)
        self._chk(self.L.ifx_track_maps(self.handle, _ptr(a[0]), _ptr(a[1]), _ptr(im[0]), _ptr(a[2]), _ptr(a[3]), _ptr(im[1]), _ptr(p), _ptr(diag)), "ifx_track_maps")
        return p.reshape(4, 4), diag

    # -- map access (getMapSurfelCount / getMapSurfelsGpu / id textures)
    def getMapSurfelCount(self):
        return self._chk(self.L.ifx_map_count(self.handle), "ifx_map_count")

    count = property(getMapSurfelCount)

    @property
    def slots(self):
        return self._chk(self.L.ifx_map_slots(self.handle), "ifx_map_slots")

    def map_view(self):
        v = SoaView()
        self._chk(self.L.ifx_map_view(self.handle, C.byref(v)), "ifx_map_view")
        return v

    def hot_records_stale(self):
        """slots whose gathered copy differed from the store when a frame was about to read it (option hot_verify; 0 unless a kept map_view pointer was written past a frame call)"""
        return self._chk(self.L.ifx_hot_records_stale(self.handle), "ifx_hot_records_stale")

    def download(self, fields=None):
        """the live surfels in map order; `fields`: a subset of (pc, nr, col, tm, ic, votes) -- a 50M-surfel map is 12.8 GB on the host, 9.6 of them votes"""
        n = self.getMapSurfelCount()
        width = dict(pc=4, nr=4, col=2, tm=2, ic=4, votes=48)
        want = tuple(width) if fields is None else tuple(fields)
        d = {k: np.zeros((n, width[k]), np.float32) for k in want}
        args = [(_ptr(d[k]) if k in d else C.c_void_p(None)) for k in ("pc", "nr", "col", "tm", "ic", "votes")]
        m = self._chk(self.L.ifx_map_download(self.handle, n, *args), "ifx_map_download")
        assert m == n
        return d

    def upload(self, m):
        n = m["pc"].shape[0]
        a = {k: np.ascontiguousarray(m[k], np.float32) for k in ("pc", "nr", "col", "tm", "ic", "votes")}
        self._chk(self.L.ifx_map_upload(self.handle, n, _ptr(a["pc"]), _ptr(a["nr"]), _ptr(a["col"]), _ptr(a["tm"]), _ptr(a["ic"]), _ptr(a["votes"])),
                  "ifx_map_upload")

    def set_pose(self, pose, tick):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self._chk(self.L.ifx_set_pose(self.handle, _ptr(p), int(tick)), "ifx_set_pose")

    def seq(self):
        """creation numbers of the live surfels in download() order (a sharded map's shards merge by them into the unsharded map)"""
        n = self.count
        out = np.zeros(max(n, 1), np.uint32)
        m = self._chk(self.L.ifx_map_seq(self.handle, _ptr(out), n), "ifx_map_seq")
        return out[:m]

    def compact(self):
        self._chk(self.L.ifx_compact(self.handle), "ifx_compact")

    def image(self, name):
        dt, ch = _IMG_SPECS[name]
        a = np.zeros((self.h, self.w, ch) if ch > 1 else (self.h, self.w), dt)
        self._chk(self.L.ifx_image_download(self.handle, name.encode(), _ptr(a), a.nbytes), "ifx_image_download")
        return a

    def getSurfelIdsAfterFusion(self):
        return self.image("ids_after")

    # -- map stage API
    def _pose(self, pose):
        return np.ascontiguousarray(pose, np.float32).reshape(16)

    def set_frame(self, rgb, depth):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        self._chk(self.L.ifx_set_frame(self.handle, _ptr(rgb), _ptr(depth)), "ifx_set_frame")

    def predict_indices(self, pose, time):
        self._chk(self.L.ifx_predict_indices(self.handle, _ptr(self._pose(pose)), int(time)), "ifx_predict_indices")

    def combined_predict(self, pose, time, max_time):
        self._chk(self.L.ifx_combined_predict(self.handle, _ptr(self._pose(pose)), int(time), int(max_time)), "ifx_combined_predict")

    def fuse(self, pose, time, weighting):
        self._chk(self.L.ifx_fuse(self.handle, _ptr(self._pose(pose)), int(time), float(weighting)), "ifx_fuse")

    def clean(self, pose, time):
        self._chk(self.L.ifx_clean(self.handle, _ptr(self._pose(pose)), int(time)), "ifx_clean")

    def render_ids(self, pose, mode=0):
        self._chk(self.L.ifx_render_ids(self.handle, _ptr(self._pose(pose)), int(mode)), "ifx_render_ids")
        return self.image("ids_tmp")

    # -- tracker
    def track_pair(self, model_v4, model_n4, model_rgba, prev_rgb, depth_filtered, rgb, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(16).copy()
        diag = np.zeros(8, np.float32)
        args = [np.ascontiguousarray(model_v4, np.float32), np.ascontiguousarray(model_n4, np.float32), np.ascontiguousarray(model_rgba, np.uint8),
                None if prev_rgb is None else np.ascontiguousarray(prev_rgb, np.uint8), np.ascontiguousarray(depth_filtered, np.uint16),
                np.ascontiguousarray(rgb, np.uint8)]
        self._chk(self.L.ifx_track_pair(self.handle, *[_ptr(a) for a in args], _ptr(p), _ptr(diag)), "ifx_track_pair")
        return p.reshape(4, 4), diag

    def build_pyramids(self, d_depth_filtered=0, d_rgb=0, d_model_v4=0, d_model_n4=0, d_model_rgba=0, model_pose=None, **out_ptrs):
        """ifx_build_pyramids: device pointers in; out_ptrs: name -> three device pointers (one per level) of caller-allocated dense buffers"""
        pyr = Pyramids()
        for name, ptrs in out_ptrs.items():
            getattr(pyr, name)[:] = [int(x) for x in ptrs]
        pose = None if model_pose is None else np.ascontiguousarray(model_pose, np.float32).reshape(16)
        self._chk(self.L.ifx_build_pyramids(self.handle, C.c_void_p(d_depth_filtered or None), C.c_void_p(d_rgb or None), C.c_void_p(d_model_v4 or None),
                                            C.c_void_p(d_model_n4 or None), C.c_void_p(d_model_rgba or None), None if pose is None else _ptr(pose), C.byref(pyr)), "ifx_build_pyramids")

    def tracker_buffer(self, name, level, m2m=False):
        if m2m:
            dt, ch, planar = _TRK_SPECS[name]
            w, h = self.w >> level, self.h >> level
            a = np.zeros((ch, h, w) if planar else ((h, w, ch) if ch > 1 else (h, w)), dt)
            self._chk(self.L.ifx_tracker_buffer_download(self.handle, ("m2m:" + name).encode(), level, _ptr(a), a.nbytes), "ifx_tracker_buffer_download")
            return a
        return self._tracker_buffer(name, level)

    def _tracker_buffer(self, name, level):
        w, h = self.w >> level, self.h >> level
        if name == "corres":
            a = np.zeros((h, w), CORRES_DTYPE)
        else:
            dt, ch, planar = _TRK_SPECS[name]
            a = np.zeros((ch, h, w) if planar else ((h, w, ch) if ch > 1 else (h, w)), dt)
        self._chk(self.L.ifx_tracker_buffer_download(self.handle, name.encode(), level, _ptr(a), a.nbytes), "ifx_tracker_buffer_download")
        return a

    # -- measurement
    def stage_ms(self, reset=False):
        out = np.zeros(4, np.float32)
        self._chk(self.L.ifx_stage_ms(self.handle, _ptr(out), int(reset)), "ifx_stage_ms")
        return dict(track=float(out[0]), fuse=float(out[1]), instance=float(out[2]), preprocess=float(out[3]))

    def lookahead_stats(self, reset=False):
        """ifx_lookahead_stats: frames that found their frame side done / their tracker run ahead / came through ifx_hint_next_frame."""
        out = np.zeros(3, np.int32)
        self._chk(self.L.ifx_lookahead_stats(self.handle, _ptr(out), int(reset)), "ifx_lookahead_stats")
        return dict(side_prepared=int(out[0]), tracked_ahead=int(out[1]), host_hinted=int(out[2]))

    def superpixel_ahead_stats(self, reset=False):
        """Superpixels run ahead of segmentation calls on the side stream (ifx_superpixel_ahead_stats): device ms, runs enqueued, runs a call used."""
        ms = C.c_float(0)
        runs = C.c_int32(0)
        used = C.c_int32(0)
        self._chk(self.L.ifx_superpixel_ahead_stats(self.handle, C.byref(ms), C.byref(runs), C.byref(used), int(reset)), "ifx_superpixel_ahead_stats")
        return dict(ms=float(ms.value), runs=int(runs.value), used=int(used.value))

    def kernel_ms(self, name):
        avg = C.c_float(0)
        n = C.c_int(0)
        self._chk(self.L.ifx_kernel_ms(self.handle, name.encode(), C.byref(avg), C.byref(n)), "ifx_kernel_ms")
        return float(avg.value), int(n.value)


class InstanceFusion:
    """Mirror of the ``InstanceFusion`` class for the hot path; masks come from the caller (replay)."""

    def __init__(self, ef: ElasticFusion):
        self.ef = ef
        self.L = ef.L

    def whetherDoSegmentation(self, frame):
        return bool(self.ef._chk(self.L.ifx_should_segment(self.ef.handle, int(frame)), "ifx_should_segment"))

    def ProcessSegmentation(self, rgb, depth, masks, class_ids, frame, isflann=False, superpixels=False):
        """InstanceFusion::ProcessSegmentation.  rgb = depth = None: the frame most recently processed (still resident on the device) instead of host copies."""
        rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
        depth = None if depth is None else np.ascontiguousarray(depth, np.uint16)
        masks = np.ascontiguousarray(masks, np.uint8)
        cls = np.ascontiguousarray(class_ids, np.int32)
        flags = (1 if isflann else 0) | (2 if superpixels else 0)
        self.ef._chk(self.L.ifx_process_segmentation(self.ef.handle, _ptr(rgb), _ptr(depth), _ptr(masks), _ptr(cls), int(masks.shape[0]), int(frame), flags),
                     "ifx_process_segmentation")

    def labels(self):
        n = self.ef.getMapSurfelCount()
        out = np.zeros(max(n, 1), np.int32)
        m = self.ef._chk(self.L.ifx_labels(self.ef.handle, _ptr(out), n), "ifx_labels")
        return out[:m]

    def precision_recall(self):
        """computePrecisionAndRecall: surfels per instance, per ground-truth id, per (gt, instance)."""
        a, b, c = np.zeros(96, np.int32), np.zeros(256, np.int32), np.zeros((256, 96), np.int32)
        self.ef._chk(self.L.ifx_precision_recall(self.ef.handle, _ptr(a), _ptr(b), _ptr(c)), "ifx_precision_recall")
        return a, b, c

    def renderProjectMap(self):
        """InstanceFusion::renderProjectMap: instance colour under every pixel (H x W x 4 float32)."""
        out = np.zeros((self.ef.h, self.ef.w, 4), np.float32)
        self.ef._chk(self.L.ifx_render_project_map(self.ef.handle, _ptr(out), None), "ifx_render_project_map")
        return out

    def getInstanceTable(self):
        out = np.zeros(96, np.int32)
        self.ef._chk(self.L.ifx_instance_table(self.ef.handle, _ptr(out)), "ifx_instance_table")
        return out

    def getLoopClosureInstanceTable(self):
        out = np.zeros(96 * 5, np.int32)
        self.ef._chk(self.L.ifx_loop_closure_instance_table(self.ef.handle, _ptr(out)), "ifx_loop_closure_instance_table")
        return out.reshape(96, 5)

    # superpixel refinement stages (IF/Core/InstanceFusion_superpixel.cpp)
    def gSLICrInterface(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        seg = np.zeros(rgb.shape[:2], np.int32)
        n = self.ef._chk(self.L.ifx_slic_segment(self.ef.handle, _ptr(rgb), _ptr(seg)), "ifx_slic_segment")
        return seg, n

    def mergeSuperPixel(self, depth, seg):
        depth = np.ascontiguousarray(depth, np.uint16)
        seg = np.ascontiguousarray(seg, np.int32).copy()
        fin = np.zeros_like(seg)
        info = np.zeros((seg.size // 256, 30), np.float32)
        self.ef._chk(self.L.ifx_merge_superpixels(self.ef.handle, _ptr(depth), _ptr(seg), _ptr(fin), _ptr(info)), "ifx_merge_superpixels")
        return seg, fin, info

    def maskSuperPixelFilter_OverSeg(self, fin, masks):
        fin = np.ascontiguousarray(fin, np.int32)
        masks = np.ascontiguousarray(masks, np.uint8).copy()
        self.ef._chk(self.L.ifx_mask_superpixel_filter(self.ef.handle, _ptr(fin), _ptr(masks), int(masks.shape[0])), "ifx_mask_superpixel_filter")
        return masks

    def maskGeometricFilter(self, model_depth, masks, ori, unavailable=None):
        depth = np.ascontiguousarray(model_depth, np.uint16)
        masks = np.ascontiguousarray(masks, np.uint8).copy()
        ori = np.ascontiguousarray(ori, np.uint8)
        un = np.zeros(masks.shape[0], np.uint8) if unavailable is None else np.ascontiguousarray(unavailable, np.uint8).copy()
        self.ef._chk(self.L.ifx_mask_geometric_filter(self.ef.handle, _ptr(depth), _ptr(masks), _ptr(ori), int(masks.shape[0]), _ptr(un)), "ifx_mask_geometric_filter")
        return masks, un

    def computeMapBoundingBox(self, bboxType=True, ratio=1000000.0):
        """InstanceFusion::computeMapBoundingBox: (boxes 96x6, ground normal, ground frame, instance frames, 648 ground votes)"""
        boxes, gn, gc, im, gv = np.zeros((96, 6), np.float32), np.zeros(3, np.float32), np.zeros((4, 4), np.float32), np.zeros((96, 4, 4), np.float32), np.zeros(648, np.int32)
        self.ef._chk(self.L.ifx_map_bounding_boxes(self.ef.handle, int(bool(bboxType)), float(ratio), _ptr(boxes), _ptr(gn), _ptr(gc), _ptr(im), _ptr(gv)), "ifx_map_bounding_boxes")
        return boxes, gn, gc, im, gv

    def getInstancePointCloud(self, inst=-1, bboxType=True, max_records=1 << 20):
        """InstanceFusion::getInstancePointCloud: (surfels per instance, records of `inst` {slot, xyz, normal, rgb})"""
        counts = np.zeros(96, np.int32)
        out = np.zeros((max_records if inst >= 0 else 1, 10), np.float32)
        n = self.ef._chk(self.L.ifx_instance_point_cloud(self.ef.handle, int(bool(bboxType)), _ptr(counts), int(inst), _ptr(out), max_records if inst >= 0 else 0), "ifx_instance_point_cloud")
        return counts, out[:n]

    def flannKnnVoteSurfelMap(self, with_neighbours=False):
        n = self.ef.slots
        nbr = np.full((max(n, 1), 10), -1, np.int32) if with_neighbours else None
        self.ef._chk(self.L.ifx_knn_vote_colour(self.ef.handle, _ptr(nbr) if with_neighbours else None, n if with_neighbours else 0), "ifx_knn_vote_colour")
        return nbr[:n] if with_neighbours else None

    def maskCleanOverlap(self, masks):
        masks = np.ascontiguousarray(masks, np.uint8).copy()
        self.ef._chk(self.L.ifx_mask_clean_overlap(self.ef.handle, _ptr(masks), int(masks.shape[0])), "ifx_mask_clean_overlap")
        return masks
