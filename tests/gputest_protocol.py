"""The reference's GPUTest protocol (elasticfusionpublic/GPUTest/src/GPUTest.cpp:69-129,247-283) on the
shipped RGB-D pair: model maps from frame 1 built on the CPU, frame 2 tracked against them with
getIncrementalTransformation(rgbOnly=false, icpWeight=10, pyramid, fastOdom=false, so3=true).
Shared by the golden generator, the CPU tests (oracle) and the GPU tests (libifx through the C-ABI)."""
from __future__ import annotations

import ctypes as C

import numpy as np

HALF_K = dict(fx=264.0, fy=264.0, cx=160.0, cy=120.0)  # 528/528/320/240 after the 2x sub-sampling


def load_vertices(d1: np.ndarray, fx, fy, cx, cy):
    """GPUTest.cpp loadVertices: depth/5000 m, forward-difference normals, valid only with 4 valid neighbours."""
    h, w = d1.shape
    z = d1.astype(np.float32) / np.float32(5000.0)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    P = np.stack([(u - np.float32(cx)) * z * np.float32(1.0 / fx), (v - np.float32(cy)) * z * np.float32(1.0 / fy), z], -1).astype(np.float32)
    V = np.zeros((h, w, 4), np.float32)
    N = np.zeros((h, w, 4), np.float32)
    V[..., 3] = 1
    N[..., 3] = 1
    valid = np.zeros((h, w), bool)
    c = d1[1:-1, 1:-1] > 0
    valid[1:-1, 1:-1] = c & (d1[2:, 1:-1] > 0) & (d1[1:-1, 2:] > 0) & (d1[:-2, 1:-1] > 0) & (d1[1:-1, :-2] > 0)
    dx = np.zeros_like(P)
    dy = np.zeros_like(P)
    dx[:, :-1] = P[:, 1:] - P[:, :-1]
    dy[:-1, :] = P[1:, :] - P[:-1, :]
    n = np.cross(dx, dy)
    n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-30)
    V[valid, :3] = P[valid]
    N[valid, :3] = n[valid].astype(np.float32)
    return np.ascontiguousarray(V), np.ascontiguousarray(N)


def protocol_inputs(c1, d1, c2, d2, K=HALF_K):
    V, N = load_vertices(d1, **K)
    rgba = np.concatenate([c1, np.full(c1.shape[:2] + (1,), 255, np.uint8)], -1).copy()
    depth_mm = (d2 // 5).astype(np.uint16)  # GPUTest.cpp loadDepth: /= 5
    return V, N, rgba, np.ascontiguousarray(c1), depth_mm, np.ascontiguousarray(c2)


def run_oracle_protocol(c1, d1, c2, d2, K=HALF_K):
    import oracle_lib as ol

    L = ol.lib()
    h, w = d1.shape
    V, N, rgba, prev, depth_mm, rgb = protocol_inputs(c1, d1, c2, d2, K)
    out = {}
    for tag, pyramid in (("single", 0), ("pyr", 1)):
        t = L.orc_tracker_create(w, h, K["fx"], K["fy"], K["cx"], K["cy"])
        L.orc_tracker_init_first_rgb(t, ol.ptr(prev))
        pose = np.eye(4, dtype=np.float32).reshape(16).copy()
        L.orc_tracker_init_model(t, ol.ptr(V), ol.ptr(N), ol.ptr(rgba), ol.ptr(pose))
        L.orc_tracker_init_frame(t, ol.ptr(depth_mm), ol.ptr(rgb), 20.0)
        diag = np.zeros(8, np.float32)
        L.orc_tracker_run(t, ol.ptr(pose), 10.0, pyramid, 0, 1, ol.ptr(diag))
        out[f"pose_{tag}"] = pose.reshape(4, 4).copy()
        out[f"diag_{tag}"] = diag.copy()
        for nm in ("last_icp29", "last_rgb29"):
            p = L.orc_tracker_buffer(t, nm.encode(), 0)
            out[f"{nm}_{tag}"] = np.frombuffer((C.c_float * 29).from_address(p), np.float32).copy()
        L.orc_tracker_destroy(t)
    return out
