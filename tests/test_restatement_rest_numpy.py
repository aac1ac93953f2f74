"""Second restatements of the rows that were a single reading after round 3: a2 bilateral filter + metric depth, a3 the frame / model
pyramid kernels, a7 the SO(3) row, a8 the Gauss-Newton host step (solve, exponential map, composition), a16 the cadence sums, a21 the superpixel merge.
Each is written here in plain numpy / Python from the reference's own source lines (cited per function) WITHOUT consulting oracle/*.c, and compared with the C
oracle on the reference's RGB-D pair (the only real data it ships).  As in test_restatement_numpy.py this does not pin the oracle to the reference -- nothing
can here -- it removes common-mode reading errors between the oracle and the HIP path, which share an author.  No GPU needed."""
import ctypes as C

import numpy as np

from conftest import SMALL  # noqa: F401  (path set-up)
from gputest_protocol import HALF_K, protocol_inputs
from test_restatement_numpy import _buf, _tracker

F = C.c_float
P = C.c_void_p


def _pair(gputest_pair):
    V, N, rgba, prev, depth_mm, rgb = protocol_inputs(*gputest_pair)
    h, w = depth_mm.shape
    return depth_mm, rgb, prev, w, h


# ------------------------------------------------------------------------------------------------ a2
def test_bilateral_and_metric_from_the_shaders(orc, gputest_pair):
    """EF/Shaders/depth_bilateral.frag:31-76: value outside [300, maxD * 1000] -> 0; else the 13 x 13 window clipped to the image, weight = exp(-(space2 *
    0.024691358 + color2 * 0.000555556)) over ALL texels of the window (zeros included), output round(sum1 / sum2).  A tap (cx, cy) is READ at
    texture(gSampler, vec2(float(cx) / cols, float(cy) / rows)) -- the corner of its texel -- with GL_NEAREST, i.e. texel floor(u * size) in f32: cx itself, or cx - 1
    for the few indices whose quotient times the size falls just below cx (the reference's shader, executed, does exactly this: tests/golden/gl_map_passes.npz).
    depth_metric.frag:29-40: the same gate, value / 1000."""
    depth, _, _, w, h = _pair(gputest_pair)
    L = orc.lib()
    maxD = 12.0
    d = depth.astype(np.float64)
    f = np.float32
    read_col = np.array([min(max(int(np.floor(f(f(f(c_) / f(w)) * f(w)))), 0), w - 1) for c_ in range(w)])     # the texel a tap at column c_ reads
    read_row = np.array([min(max(int(np.floor(f(f(f(r_) / f(h)) * f(h)))), 0), h - 1) for r_ in range(h)])
    s1 = np.zeros((h, w)); s2 = np.zeros((h, w))
    for dy in range(-6, 7):
        for dx in range(-6, 7):
            ys, xs = np.mgrid[0:h, 0:w]
            cy, cx = ys + dy, xs + dx
            ok = (cy >= 0) & (cy < h) & (cx >= 0) & (cx < w)
            t = d[read_row[np.clip(cy, 0, h - 1)], read_col[np.clip(cx, 0, w - 1)]]
            wgt = np.where(ok, np.exp(-((dx * dx + dy * dy) * 0.024691358 + (d - t) ** 2 * 0.000555556)), 0.0)
            s1 += t * wgt; s2 += wgt
    gate = (depth > int(maxD * 1000.0)) | (depth < 300)
    want = np.where(gate, 0, np.rint(s1 / np.maximum(s2, 1e-300))).astype(np.int64)
    got = np.zeros((h, w), np.uint16)
    L.orc_bilateral.argtypes = [P, P, C.c_int, C.c_int, F]
    L.orc_bilateral(orc.ptr(np.ascontiguousarray(depth)), orc.ptr(got), w, h, maxD)
    diff = np.abs(got.astype(np.int64) - want)
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (diff.max(), (diff != 0).mean())   # f32 sums against f64: a rounding tie now and then
    assert np.array_equal(got == 0, want == 0)
    m = np.zeros((h, w), np.float32)
    L.orc_metric.argtypes = [P, P, C.c_int, C.c_int, F]
    L.orc_metric(orc.ptr(np.ascontiguousarray(depth)), orc.ptr(m), w, h, maxD)
    assert np.array_equal(m, np.where(gate, 0.0, depth.astype(np.float32) / np.float32(1000.0)).astype(np.float32))


# ------------------------------------------------------------------------------------------------ a3
def _pyrdown_u16(src):
    """pyrDownGaussKernel, EF/Cuda/cudafuncs.cu:57-91: 5 x 5 window around (2x, 2y) clipped to the image, weights {0.375, 0.25, 0.0625} per axis, only taps
    within 3 * 30 of the centre value, dst = int(sum / wall)."""
    sh, sw = src.shape
    dh, dw = sh // 2, sw // 2
    out = np.zeros((dh, dw), np.uint16)
    wts = np.array([0.375, 0.25, 0.0625], np.float32)
    s = src.astype(np.int64)
    for y in range(dh):
        for x in range(dw):
            c = s[2 * y, 2 * x]
            sm = np.float32(0); wall = np.float32(0)
            for yi in range(max(0, 2 * y - 2) - 2 * y, min(sh, 2 * y + 3) - 2 * y):
                for xi in range(max(0, 2 * x - 2) - 2 * x, min(sw, 2 * x + 3) - 2 * x):
                    v = s[2 * y + yi, 2 * x + xi]
                    if abs(v - c) < 90:
                        wgt = wts[abs(xi)] * wts[abs(yi)]
                        sm = np.float32(sm + np.float32(v) * wts[abs(xi)] * wts[abs(yi)]); wall = np.float32(wall + wgt)
            out[y, x] = int(sm / wall)
    return out


def _gauss_down(src, test, cast):
    """pyrDownKernelGaussF / pyrDownKernelIntensityGauss, cudafuncs.cu:332-361, :470-497: window [2x - 2, min(2x + 3, cols - 1)) x likewise, weight table indexed from
    the window's END ((ty - cy - 1) * 5 + (tx - cx - 1)), integer count of the weights, dst = sum / count."""
    g = np.array([1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1], np.float32)
    sh, sw = src.shape
    dh, dw = sh // 2, sw // 2
    out = np.zeros((dh, dw), src.dtype)
    for y in range(dh):
        for x in range(dw):
            tx, ty = min(2 * x + 3, sw - 1), min(2 * y + 3, sh - 1)
            sm = np.float32(0); cnt = 0
            for cy in range(max(0, 2 * y - 2), ty):
                for cx in range(max(0, 2 * x - 2), tx):
                    v = src[cy, cx]
                    if test(v):
                        k = g[(ty - cy - 1) * 5 + (tx - cx - 1)]
                        sm = np.float32(sm + np.float32(v) * k); cnt += int(k)
            with np.errstate(all="ignore"):
                out[y, x] = cast(np.float32(sm) / np.float32(cnt))
    return out


def test_pyramid_kernels_from_the_cuda_source(orc, gputest_pair):
    """The frame side's kernels on the reference pair, level 0 -> 1 (and the model side's float variants on the resulting maps)."""
    depth, rgb, _, w, h = _pair(gputest_pair)
    L = orc.lib()
    fx, fy, cx, cy = (HALF_K[k] for k in ("fx", "fy", "cx", "cy"))
    # depth pyramid
    got = np.zeros((h // 2, w // 2), np.uint16)
    L.orc_pyrdown_u16.argtypes = [P, C.c_int, C.c_int, P]
    L.orc_pyrdown_u16(orc.ptr(np.ascontiguousarray(depth)), w, h, orc.ptr(got))
    assert np.array_equal(got, _pyrdown_u16(depth))
    # vertex map: computeVmapKernel, cudafuncs.cu:109-134 (z = depth / 1000; z != 0 and z < cutoff -> (z (u - cx) / fx, z (v - cy) / fy, z), else NaN in x)
    vm = np.zeros((3, h, w), np.float32)
    L.orc_vmap.argtypes = [P, C.c_int, C.c_int, F, F, F, F, F, P]
    L.orc_vmap(orc.ptr(np.ascontiguousarray(depth)), w, h, fx, fy, cx, cy, 20.0, orc.ptr(vm))
    z = depth.astype(np.float32) / np.float32(1000.0)
    vs, us = np.mgrid[0:h, 0:w].astype(np.float32)
    ok = (z != 0) & (z < 20.0)
    ifx_, ify_ = np.float32(1.0) / np.float32(fx), np.float32(1.0) / np.float32(fy)
    wantx = np.where(ok, z * (us - np.float32(cx)) * ifx_, np.nan).astype(np.float32)
    assert np.array_equal(np.isnan(vm[0]), ~ok)
    assert np.array_equal(vm[0][ok], wantx[ok]) and np.array_equal(vm[1][ok], (z * (vs - np.float32(cy)) * ify_)[ok]) and np.array_equal(vm[2][ok], z[ok])
    # normal map: computeNmapKernel, :151-188 (forward differences, NaN on the last row / column and wherever one of the three vertices is NaN)
    nm = np.zeros((3, h, w), np.float32)
    L.orc_nmap.argtypes = [P, C.c_int, C.c_int, P]
    L.orc_nmap(orc.ptr(vm), w, h, orc.ptr(nm))
    v = np.where(np.isnan(vm[0])[None], np.nan, vm).astype(np.float64)
    v00 = v[:, :-1, :-1]; v01 = v[:, :-1, 1:]; v10 = v[:, 1:, :-1]
    valid = ~(np.isnan(v00[0]) | np.isnan(v01[0]) | np.isnan(v10[0]))
    cr = np.cross((v01 - v00).transpose(1, 2, 0), (v10 - v00).transpose(1, 2, 0))
    with np.errstate(all="ignore"):
        n = cr / np.linalg.norm(cr, axis=2, keepdims=True)
    full = np.zeros((h, w), bool); full[:-1, :-1] = valid
    assert np.array_equal(~np.isnan(nm[0]), full)
    assert np.nanmax(np.abs(nm[:, :-1, :-1].transpose(1, 2, 0)[valid] - n[valid])) < 2e-4
    # resizeMap<false / true>, :365-411: 2 x 2 mean, NaN if any of the four x components is NaN, optionally normalised
    for normalize, src in ((0, vm), (1, nm)):
        src = src.copy()
        src[1:][:, np.isnan(src[0])] = np.nan   # (the y, z planes of an invalid pixel hold whatever was there: make them defined for the comparison)
        out = np.zeros((3, h // 2, w // 2), np.float32)
        L.orc_resize_map.argtypes = [P, C.c_int, C.c_int, P, C.c_int]
        L.orc_resize_map(orc.ptr(src), w, h, orc.ptr(out), normalize)
        s64 = src.astype(np.float64)
        q = [s64[:, 0::2, 0::2], s64[:, 0::2, 1::2], s64[:, 1::2, 0::2], s64[:, 1::2, 1::2]]
        bad = np.isnan(q[0][0]) | np.isnan(q[1][0]) | np.isnan(q[2][0]) | np.isnan(q[3][0])
        mean = (q[0] + q[1] + q[2] + q[3]) / 4
        if normalize:
            with np.errstate(all="ignore"):
                mean = mean / np.linalg.norm(mean, axis=0, keepdims=True)
        assert np.array_equal(np.isnan(out[0]), bad)
        assert np.nanmax(np.abs(out[:, ~bad] - mean[:, ~bad])) < 1e-5
    # intensity pyramid + Sobel on the colour image's intensity (bgr2Intensity: (r * 0.114 + g * 0.299 + b * 0.587) per :550-566 is exercised through the tracker buffers)
    t, _, _ = _tracker(orc, gputest_pair)
    i0 = _buf(orc, t, "next_img", 0, np.uint8, (h, w))
    i1 = _buf(orc, t, "next_img", 1, np.uint8, (h // 2, w // 2))
    assert np.array_equal(i1, _gauss_down(i0, lambda v_: v_ > 0, lambda f_: np.uint8(int(f_)) if np.isfinite(f_) else np.uint8(0)))
    d0 = _buf(orc, t, "last_depth", 0, np.float32, (h, w))
    d1 = _buf(orc, t, "last_depth", 1, np.float32, (h // 2, w // 2))
    want = _gauss_down(d0, lambda v_: not np.isnan(v_), lambda f_: np.float32(f_))
    assert np.array_equal(np.isnan(d1), np.isnan(want)) and np.array_equal(d1[~np.isnan(d1)], want[~np.isnan(want)])
    # applyKernel, :583-607: the 3 x 3 Sobel whose coefficient index counts DOWN from 8 over the in-bounds taps only (misaligned on the border), float -> short
    gx = np.array([0.52201, 0.0, -0.52201, 0.79451, -0.0, -0.79451, 0.52201, 0.0, -0.52201], np.float32)
    gy = np.array([0.52201, 0.79451, 0.52201, 0.0, 0.0, 0.0, -0.52201, -0.79451, -0.52201], np.float32)
    dx = np.zeros((h, w), np.int16); dy = np.zeros((h, w), np.int16)   # (computeDerivativeImages runs inside getIncrementalTransformation, :287-293: called here as a stage)
    L.orc_sobel.argtypes = [P, C.c_int, C.c_int, P, P]
    L.orc_sobel(orc.ptr(np.ascontiguousarray(i0)), w, h, orc.ptr(dx), orc.ptr(dy))
    rng = np.random.RandomState(3)
    pts = [(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (0, 7), (9, 0)] + [(int(rng.randint(h)), int(rng.randint(w))) for _ in range(4000)]
    for (y, x) in pts:
        ax = np.float32(0); ay = np.float32(0); k = 8
        for j in range(max(y - 1, 0), min(y + 1, h - 1) + 1):
            for i in range(max(x - 1, 0), min(x + 1, w - 1) + 1):
                ax = np.float32(ax + np.float32(i0[j, i]) * gx[k]); ay = np.float32(ay + np.float32(i0[j, i]) * gy[k]); k -= 1
        assert (int(ax), int(ay)) == (int(dx[y, x]), int(dy[y, x])), (y, x)
    L.orc_tracker_destroy(t)


# ------------------------------------------------------------------------------------------------ a7
def test_so3_rows_from_the_formulas(orc, gputest_pair):
    """SO3Reduction::getProducts, EF/Cuda/reduce.cu:972-1055: warp (x, y, 1) by imageBasis, round to the pixel, both pixels at least one from the border; the gradient of an
    image is ((back + actu) / 2 - (fore + actu) / 2) per axis, averaged over the two images; point = kinv (x, y, 1); leftProduct per the reference's Jacobian;
    row = [leftProduct x point, -(next(warped) - last(x, y))]; the ten products + the count."""
    _, _, _, w, h = _pair(gputest_pair)
    L = orc.lib()
    t, _, _ = _tracker(orc, gputest_pair)
    l2 = _buf(orc, t, "lastnext_img", 2, np.uint8, (h // 4, w // 4)).astype(np.float64)
    n2 = _buf(orc, t, "next_img", 2, np.uint8, (h // 4, w // 4)).astype(np.float64)
    hh, ww = l2.shape
    fx, fy, cx, cy = (HALF_K[k] / 4.0 for k in ("fx", "fy", "cx", "cy"))
    Kk = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    ang = np.array([0.004, -0.006, 0.003])
    th = np.linalg.norm(ang); kx = ang / th
    Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    ib = (Kk @ R @ np.linalg.inv(Kk)).astype(np.float32).astype(np.float64)     # (the kernel receives f32 matrices)
    kinv = np.linalg.inv(Kk).astype(np.float32).astype(np.float64)
    krlr = (Kk @ R).astype(np.float32).astype(np.float64)
    ys, xs = np.mgrid[0:hh, 0:ww].astype(np.float64)
    up = np.stack([xs, ys, np.ones_like(xs)], -1)
    wp = up @ ib.T
    wx = np.rint(wp[..., 0] / wp[..., 2]).astype(np.int64); wy = np.rint(wp[..., 1] / wp[..., 2]).astype(np.int64)
    found = (wx >= 1) & (wx < ww - 1) & (wy >= 1) & (wy < hh - 1) & (xs >= 1) & (xs < ww - 1) & (ys >= 1) & (ys < hh - 1)
    wxc, wyc = np.clip(wx, 1, ww - 2), np.clip(wy, 1, hh - 2)
    xc, yc = np.clip(xs.astype(np.int64), 1, ww - 2), np.clip(ys.astype(np.int64), 1, hh - 2)

    def grad(img, x, y):
        a = img[y, x]
        return (img[y, x - 1] + a) / 2 - (img[y, x + 1] + a) / 2, (img[y - 1, x] + a) / 2 - (img[y + 1, x] + a) / 2

    gnx, gny = grad(n2, wxc, wyc)
    glx, gly = grad(l2, xc, yc)
    gx, gy = (gnx + glx) / 2, (gny + gly) / 2
    pt = up @ kinv.T
    z2 = pt[..., 2] ** 2
    a, b, c = krlr[0]; d, e, f = krlr[1]; g, hq, iq = krlr[2]
    lp = np.stack([((pt[..., 2] * (d * gy + a * gx)) - gy * g * ys - gx * g * xs) / z2,
                   ((pt[..., 2] * (e * gy + b * gx)) - gy * hq * ys - gx * hq * xs) / z2,
                   ((pt[..., 2] * (f * gy + c * gx)) - gy * iq * ys - gx * iq * xs) / z2], -1)
    jr = np.cross(lp, pt)
    r3 = -(n2[wyc, wxc] - l2[yc, xc])
    rows = np.concatenate([jr, r3[..., None]], -1)[found]
    want = [np.sum(rows[:, i] * rows[:, j]) for i in range(3) for j in range(i, 4)] + [np.sum(rows[:, 3] ** 2), float(found.sum())]
    got = np.zeros(11, np.float32)
    L.orc_so3_step.argtypes = [P] * 5 + [C.c_int, C.c_int, P]
    f32 = lambda m_: np.ascontiguousarray(m_, np.float32)
    L.orc_so3_step(orc.ptr(np.ascontiguousarray(l2.astype(np.uint8))), orc.ptr(np.ascontiguousarray(n2.astype(np.uint8))), orc.ptr(f32(ib)), orc.ptr(f32(kinv)), orc.ptr(f32(krlr)),
                   ww, hh, orc.ptr(got))
    want = np.array(want)
    assert got[10] == want[10] and want[10] > 1000
    assert np.all(np.abs(got[:10] - want[:10]) <= 2e-4 * np.maximum(1.0, np.abs(want[:10]))), (got, want)
    L.orc_tracker_destroy(t)


# ------------------------------------------------------------------------------------------------ a8
def _rodrigues(v):
    """OdometryProvider::rodrigues, EF/Utils/OdometryProvider.h:35-71"""
    th = np.linalg.norm(v)
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    r = v / th
    rrt = np.outer(r, r)
    rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * rrt + np.sin(th) * rx


def test_gauss_newton_host_step_from_the_reference_loop(orc, gputest_pair):
    """EF/Utils/RGBDOdometry.cpp:384-603 driven from numpy: per iteration Rt = resultRt^-1, KRK^-1 and Kt for the residual pass, sigma = sqrt(count) by the reference's
    precedence (:461), A = A_rgb + w^2 A_icp, b = b_rgb + w b_icp in f64, x = A \\ b, resultRt = [rodrigues(x[3:]) | x[:3]] resultRt (OdometryProvider.h:73-93),
    current = prev * rgbOdom^-1 in f32 -- the oracle only supplies the reductions (its stage functions, cross-checked row by row elsewhere).  Single scale, three
    iterations (fastOdom), no SO(3): the pose equals orc_tracker_run's to f32 rounding."""
    L = orc.lib()
    t, w, h = _tracker(orc, gputest_pair)
    fx, fy, cx, cy = (HALF_K[k] for k in ("fx", "fy", "cx", "cy"))
    g = lambda name, dt, shp: _buf(orc, t, name, 0, dt, shp)
    vc, nc, vp, nprev = (np.ascontiguousarray(g(n_, np.float32, (3, h, w))) for n_ in ("vmap_curr", "nmap_curr", "vmap_prev", "nmap_prev"))
    last_d, next_d = np.ascontiguousarray(g("last_depth", np.float32, (h, w))), np.ascontiguousarray(g("next_depth", np.float32, (h, w)))
    last_i, next_i = np.ascontiguousarray(g("last_img", np.uint8, (h, w))), np.ascontiguousarray(g("next_img", np.uint8, (h, w)))
    didx = np.zeros((h, w), np.int16); didy = np.zeros((h, w), np.int16)   # computeDerivativeImages(nextImage), RGBDOdometry.cpp:287-293, and projectToPointCloud(lastDepth), :407
    L.orc_sobel.argtypes = [P, C.c_int, C.c_int, P, P]
    L.orc_sobel(orc.ptr(next_i), w, h, orc.ptr(didx), orc.ptr(didy))
    cloud = np.zeros((h, w, 3), np.float32)
    L.orc_project_cloud.argtypes = [P, C.c_int, C.c_int, F, F, F, F, P]
    L.orc_project_cloud(orc.ptr(last_d), w, h, fx, fy, cx, cy, orc.ptr(cloud))
    Rprev = np.eye(3, dtype=np.float32); tprev = np.zeros(3, np.float32)
    Rcurr, tcurr = Rprev.copy(), tprev.copy()
    Rprev_inv = np.linalg.inv(Rprev).astype(np.float32)
    Kk = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    resultRt = np.eye(4)
    wgt = 10.0
    corres = np.zeros((h * w, 4), np.int32)     # orc_dataterm: short zero_x, zero_y, one_x, one_y; float diff; int valid = 16 bytes
    L.orc_rgb_residual.argtypes = [F, P, P, P, P, P, P, P, F, P, P, C.c_int, C.c_int, P, P]
    L.orc_icp_step.argtypes = [P] * 6 + [F] * 4 + [P] * 2 + [F] * 2 + [C.c_int] * 2 + [P]
    L.orc_rgb_step.argtypes = [P, F, P, F, F, P, P, F, C.c_int, C.c_int, P]
    for _ in range(3):
        Rt = np.linalg.inv(resultRt)
        krk = np.ascontiguousarray((Kk @ Rt[:3, :3] @ np.linalg.inv(Kk)).astype(np.float32))
        kt = np.ascontiguousarray((Kk @ Rt[:3, 3]).astype(np.float32))
        cnt = C.c_int(0); sig = C.c_int(0)
        L.orc_rgb_residual(float(5.0 ** 2 / (1.0 / 8.0) ** 2), orc.ptr(didx), orc.ptr(didy), orc.ptr(last_d), orc.ptr(next_d), orc.ptr(last_i), orc.ptr(next_i), orc.ptr(corres),
                           0.07, orc.ptr(kt), orc.ptr(krk), w, h, C.byref(cnt), C.byref(sig))
        q = np.float32(sig.value) / np.float32(cnt.value) if cnt.value else np.float32(np.nan)
        sigma_val = float(np.sqrt(np.float32(1 if q == 0 else cnt.value)))                  # std::sqrt((float)sigma / rgbSize == 0 ? 1 : rgbSize)
        icp29 = np.zeros(29, np.float32); rgb29 = np.zeros(29, np.float32)
        L.orc_icp_step(orc.ptr(np.ascontiguousarray(Rcurr)), orc.ptr(np.ascontiguousarray(tcurr)), orc.ptr(vc), orc.ptr(nc), orc.ptr(Rprev_inv), orc.ptr(tprev), fx, fy, cx, cy,
                       orc.ptr(vp), orc.ptr(nprev), 0.10, float(np.sin(20.0 * 3.14159254 / 180.0)), w, h, orc.ptr(icp29))
        L.orc_rgb_step(orc.ptr(corres), sigma_val, orc.ptr(cloud), fx, fy, orc.ptr(didx), orc.ptr(didy), 1.0 / 8.0, w, h, orc.ptr(rgb29))

        def system(o):
            A = np.zeros((6, 6)); b = np.zeros(6); k = 0
            for i in range(6):
                for j in range(i, 7):
                    if j == 6:
                        b[i] = o[k]
                    else:
                        A[i, j] = A[j, i] = o[k]
                    k += 1
            return A, b

        Ai, bi = system(icp29.astype(np.float64)); Ar, br = system(rgb29.astype(np.float64))
        x = np.linalg.solve(Ar + wgt * wgt * Ai, br + wgt * bi)
        upd = np.eye(4); upd[:3, :3] = _rodrigues(x[3:]); upd[:3, 3] = x[:3]
        resultRt = upd @ resultRt
        odom = np.eye(4, dtype=np.float32); odom[:3, :3] = resultRt[:3, :3].astype(np.float32); odom[:3, 3] = resultRt[:3, 3].astype(np.float32)
        cur = np.eye(4, dtype=np.float32); cur[:3, :3] = Rprev; cur[:3, 3] = tprev
        cur = (cur @ np.linalg.inv(odom.astype(np.float64)).astype(np.float32)).astype(np.float32)
        Rcurr, tcurr = np.ascontiguousarray(cur[:3, :3]), np.ascontiguousarray(cur[:3, 3])
    pose = np.eye(4, dtype=np.float32).reshape(16).copy()
    diag = np.zeros(8, np.float32)
    L.orc_tracker_run.argtypes = [P, P, F, C.c_int, C.c_int, C.c_int, P]
    L.orc_tracker_run(t, orc.ptr(pose), 10.0, 0, 1, 0, orc.ptr(diag))
    pose = pose.reshape(4, 4)
    assert np.linalg.norm(pose[:3, 3]) > 1e-3                                           # the pair really moves
    assert np.abs(pose[:3, :3] - Rcurr).max() < 2e-6 and np.abs(pose[:3, 3] - tcurr).max() < 2e-6, (pose, Rcurr, tcurr)
    L.orc_tracker_destroy(t)


# ------------------------------------------------------------------------------------------------ a16
def test_cadence_decision_from_the_source(orc, small_stream):
    """InstanceFusion::whetherDoSegmentation + checkProjectDepthAndInstanceKernel (IF/Core/InstanceFusion.cpp:192-238, InstanceFusionCuda.cu:737-760): over every 10th pixel
    of every 10th row of the id image, count[0] += the 96 decoded vote counters of the surfel under the pixel (ids 1 .. n - 1), count[1] += 1 where there is none;
    slow cadence (> 45 frames since the last call) when count[0] > w / 10 * h / 10 * 0.48 * 30 or count[1] < w / 10 * h / 10 * 0.2, else fast (> 2 frames);
    lastSegFrameID starts at -1 and moves only when the answer is yes.  Replayed in Python on the oracle's own id image and votes, before and after votes exist."""
    from instancefusion_amd import synth

    st = small_stream
    W, H = SMALL["w"], SMALL["h"]
    o = orc.Oracle(**SMALL, max_surfels=400000)
    for i in range(6):
        po = o.process_frame(st["rgb"][i], st["depth"][i])
    state = {"last": -1}

    def decide(frame):
        ids = o.image("ids_after").astype(np.int64)
        votes = o.download()["votes"]
        n = votes.shape[0]
        c0 = c1 = 0
        for y in range(0, H, 10):
            for x in range(0, W, 10):
                sid = int(ids[y, x])
                if 0 < sid < n:
                    v = np.trunc(votes[sid]).astype(np.int64)
                    a = (((v >> 16) & 0xFFFF) ^ 0x8000) - 0x8000          # short(int(f) >> 16 & 0xFFFF), short(int(f) & 0xFFFF)
                    b = ((v & 0xFFFF) ^ 0x8000) - 0x8000
                    c0 += int(a.sum() + b.sum())
                else:
                    c1 += 1
        t1 = c0 > (W // 10 * H // 10 * 0.48 * 30)
        t2 = c1 < (W // 10 * H // 10 * 0.2)
        gap = 45 if (t1 or t2) else 2
        if frame - state["last"] > gap:
            state["last"] = frame
            return True, (c0, c1, t1, t2)
        return False, (c0, c1, t1, t2)

    got, want, seen_fast, seen_slow = [], [], False, False
    masks, cls = synth.canned_masks(st["obj"][5], st["scene"])
    for f in range(0, 60):
        if f == 20:                                  # until here: a young map, nothing stable, the id image all but empty -- the fast cadence; from here on a stable map that carries votes: the slow one
            m = o.download(); m["pc"][:, 3] = 20.0
            o.upload(m); o.set_pose(po, o.tick)
            o.process_frame(st["rgb"][5], st["depth"][5], in_pose=po)
            for rep in range(12):
                o.process_segmentation(st["rgb"][5], st["depth"][5], masks, cls, 1000 + rep)
        w_, info = decide(f)
        want.append(w_); got.append(bool(o.should_segment(f)))
        seen_fast |= not (info[2] or info[3]); seen_slow |= info[2] or info[3]
    assert got == want, (got, want)
    assert seen_fast and seen_slow and sum(want) >= 5
    o.close()


# ------------------------------------------------------------------------------------------------ a21
def _sp_merge_restated(depth, seg_in, w, h, spn, cam):
    """mergeSuperPixel, IF/Core/InstanceFusion_superpixel.cpp:45-160, with the kernels of IF/Core/InstanceFusionCuda.cu:144-735 and connectSuperPixel (:269-426), as plain
    numpy / Python.  Where the reference is order-dependent the documented deterministic reading is taken ("deviations": exact sums, neighbour list =
    the ascending set of distinct 4-neighbours, first 11; a -1 slot is skipped in the re-clustering)."""
    f32 = np.float32
    d = depth.astype(np.int64)
    # depthMapGaussianfilterKernel (:144-168): all eight neighbours present; weights 4 / 2 / 1 over the non-zero ones; integer division
    nz = d != 0
    inner = np.zeros((h, w), bool); inner[1:-1, 1:-1] = True

    def sh(a, dy, dx):
        out = np.zeros_like(a)
        ys = slice(max(0, dy), h + min(0, dy)); yd = slice(max(0, -dy), h + min(0, -dy))
        xs = slice(max(0, dx), w + min(0, dx)); xd = slice(max(0, -dx), w + min(0, -dx))
        out[yd, xd] = a[ys, xs]
        return out

    def all8(m):
        ok = inner.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy or dx:
                    ok &= sh(m, dy, dx)
        return ok

    ok8 = all8(nz)
    sm = np.zeros((h, w), np.int64); n = np.zeros((h, w), np.int64)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            wgt = 4 if (dy == 0 and dx == 0) else (2 if (dy == 0 or dx == 0) else 1)
            v = sh(d, dy, dx)
            sm += np.where(v != 0, wgt * v, 0); n += np.where(v != 0, wgt, 0)
    dg = np.where(ok8 & (n > 0), sm // np.maximum(n, 1), 0).astype(np.int64)
    # getVertex (:183-189): z = float(d) / 1186; ((x - cx) * z) * (1 / fx)
    xs_, ys_ = np.meshgrid(np.arange(w), np.arange(h))
    z = dg.astype(f32) / f32(1186.0)

    def vertex(zz, xx, yy):
        return np.stack([(xx.astype(f32) - f32(cam[0])) * zz * f32(cam[2]), (yy.astype(f32) - f32(cam[1])) * zz * f32(cam[3]), zz], -1).astype(f32)

    V = vertex(z, xs_, ys_)
    pos = np.where((dg != 0)[..., None], V, f32(0)).astype(f32)
    # getNormal (:211-273): five cross products of one-sided / central differences, weights 4, 2, 2, 2, 2, normalised; only where all eight neighbours of the FILTERED depth exist
    okn = all8(dg != 0)
    vxf, vxb, vyf, vyb = (np.roll(V, -1, 1), np.roll(V, 1, 1), np.roll(V, -1, 0), np.roll(V, 1, 0))

    def ncross(left, right, up, down):
        dx_, dy_ = (left - right).astype(f32), (up - down).astype(f32)
        return np.stack([dx_[..., 1] * dy_[..., 2] - dx_[..., 2] * dy_[..., 1], dx_[..., 2] * dy_[..., 0] - dx_[..., 0] * dy_[..., 2], dx_[..., 0] * dy_[..., 1] - dx_[..., 1] * dy_[..., 0]], -1).astype(f32)

    s_ = (ncross(vxb, vxf, vyb, vyf) * f32(4)).astype(f32)
    for (l_, r_, u_, dn_) in ((vxb, V, vyb, V), (V, vxf, vyb, V), (vxb, V, V, vyf), (V, vxf, V, vyf)):
        s_ = (s_ + ncross(l_, r_, u_, dn_) * f32(2)).astype(f32)
    with np.errstate(all="ignore"):
        ln = np.sqrt((s_[..., 0] * s_[..., 0] + s_[..., 1] * s_[..., 1] + s_[..., 2] * s_[..., 2]).astype(f32)).astype(f32)
        nor = np.where(okn[..., None], (s_ / ln[..., None]).astype(f32), f32(0)).astype(f32)
    # kernel A (:351-371)
    seg = seg_in.astype(np.int64).copy()
    pn = (pos.astype(np.float64) ** 2).sum(-1) + (nor.astype(np.float64) ** 2).sum(-1)
    with np.errstate(all="ignore"):
        seg[~(pn.astype(f32) >= f32(0.01)) | (seg >= spn) | (seg < 0)] = -1   # (pnTest < 0.01, or NaN: the reference's `<` is false for NaN -- the oracle drops non-finite pixels)
    seg[np.isnan(pn)] = -1
    # kernel B (:373-440)
    info = np.zeros((spn, 30), np.float64)
    info[:, 18:29] = -1; info[:, 17] = 11
    valid = seg >= 0
    ids = seg[valid]
    info[:, 0] = np.bincount(ids, minlength=spn)
    for c in range(3):
        info[:, 1 + c] = np.bincount(ids, weights=pos[..., c][valid].astype(np.float64), minlength=spn)
        info[:, 4 + c] = np.bincount(ids, weights=nor[..., c][valid].astype(np.float64), minlength=spn)
    info[:, 13] = np.bincount(ids, weights=dg[valid].astype(np.float64), minlength=spn)
    nb = [set() for _ in range(spn)]
    for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        a = seg[1:-1, 1:-1]; b = seg[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx]
        m = (a >= 0) & (b >= 0) & (a != b)
        for i_, j_ in zip(a[m].tolist(), b[m].tolist()):
            nb[i_].add(j_)
    for i_ in range(spn):
        lst = sorted(nb[i_])[:11]
        info[i_, 18:18 + len(lst)] = lst
    info = info.astype(f32)

    def averages(I):   # kernels C / E (:496-527, :623-657)
        for i_ in range(spn):
            t = int(I[i_, 0])
            if t != 0:
                I[i_, 7:10] = I[i_, 1:4] / f32(t)
                nx, ny, nz_ = I[i_, 4:7]
                with np.errstate(all="ignore"):
                    ln_ = np.sqrt(f32(nx * nx + ny * ny + nz_ * nz_))
                    I[i_, 10:13] = I[i_, 4:7] / ln_
                I[i_, 14] = I[i_, 13] / f32(t)

    averages(info)
    # kernel D (:530-621): every pixel looks at its superpixel's neighbours and itself; nearest by |plane distance| + centre distance; re-cluster
    avg_pos, avg_nor, avg_dep = info[:, 7:10].astype(np.float64), info[:, 10:13].astype(np.float64), info[:, 14].astype(np.float64)
    P_ = pos.reshape(-1, 3).astype(np.float64); Nn = nor.reshape(-1, 3).astype(np.float64)
    sflat = seg.reshape(-1)
    pix = np.nonzero(sflat >= 0)[0]
    cand = np.concatenate([info[sflat[pix], 18:29].astype(np.int64), sflat[pix][:, None]], 1)      # eleven neighbours, then itself
    best_d = np.full(len(pix), 999999.9); best_n = np.full(len(pix), 999999.9); best_id = sflat[pix].copy()
    for c in range(12):
        idt = cand[:, c]
        okc = idt >= 0
        it_ = np.where(okc, idt, 0)
        A = avg_nor[it_]; B = avg_pos[it_] - P_[pix]
        with np.errstate(all="ignore"):
            la = np.sqrt((A * A).sum(1)); lb = np.sqrt((B * B).sum(1))
            dist = np.abs((A * B).sum(1) / la) + lb
        dn_ = ((np.abs(A - Nn[pix])) ** 2).sum(1)
        take = okc & (dist < best_d)
        best_d = np.where(take, dist, best_d); best_n = np.where(take, dn_, best_n); best_id = np.where(take, idt, best_id)
    thr = (0.026 * avg_dep[best_id] - 4.0) / 1186.0
    best_id = np.where(best_d > 2 * thr, -1, best_id)
    I2 = info.astype(np.float64)
    keep = best_id >= 0
    I2[:, 15] += np.bincount(best_id[keep], weights=(best_d[keep] ** 2), minlength=spn)
    I2[:, 16] += np.bincount(best_id[keep], weights=best_n[keep], minlength=spn)
    mv = best_id != sflat[pix]
    for sign, tgt in ((-1.0, sflat[pix][mv]), (1.0, best_id[mv])):
        sel = tgt >= 0
        tt = tgt[sel]
        I2[:, 0] += sign * np.bincount(tt, minlength=spn)
        for c in range(3):
            I2[:, 1 + c] += sign * np.bincount(tt, weights=P_[pix][mv][sel][:, c], minlength=spn)
            I2[:, 4 + c] += sign * np.bincount(tt, weights=Nn[pix][mv][sel][:, c], minlength=spn)
        I2[:, 13] += sign * np.bincount(tt, weights=dg.reshape(-1)[pix][mv][sel].astype(np.float64), minlength=spn)
    seg2 = sflat.copy(); seg2[pix] = best_id
    info = I2.astype(f32)
    for i_ in range(spn):   # kernel E's deviations, then the averages again
        t = int(info[i_, 0])
        if t != 0:
            info[i_, 15] = np.sqrt(f32(info[i_, 15] / f32(t))); info[i_, 16] = np.sqrt(f32(info[i_, 16] / f32(t)))
    averages(info)
    # connectSuperPixel (:269-426), sequential as written
    I = info.astype(np.float64)
    I[:, 29] = -1
    for i_ in range(spn):
        for j_ in range(int(I[i_, 17])):
            b_ = int(I[i_, 18 + j_])
            if b_ == -1:
                continue
            va = I[i_, 10:13]; vb = I[i_, 7:10] - I[b_, 7:10]
            with np.errstate(all="ignore"):
                dist_term = abs(float(va @ vb) / np.sqrt(float(va @ va))) + np.sqrt(float(vb @ vb))
            nor_term = 0.1 * np.sqrt(float(((np.abs(I[i_, 10:13] - I[b_, 10:13])) ** 2).sum()))
            th_a = (0.026 * I[i_, 14] - 4.0) / 1186.0 + 2 * I[i_, 15]
            th_b = (0.026 * I[b_, 14] - 4.0) / 1186.0 + 2 * I[b_, 15]
            fin_test = dist_term + nor_term
            if fin_test > th_a or fin_test > th_b:
                I[i_, 18 + j_] = -1
                for k_ in range(int(I[b_, 17])):
                    if int(I[b_, 18 + k_]) == i_:
                        I[b_, 18 + k_] = -1
                        break
    for i_ in range(spn):
        final_id = i_ if I[i_, 29] == -1 else int(I[i_, 29])
        stack = [i_]
        while stack:
            tg = stack.pop()
            if I[tg, 29] != -1:
                continue
            I[tg, 29] = final_id
            for j_ in range(int(I[tg, 17])):
                c_ = int(I[tg, 18 + j_])
                if c_ == -1 or I[c_, 29] != -1:
                    continue
                stack.append(c_)
    final_of = I[:, 29].astype(np.int64)
    fin = np.where(seg2 >= 0, final_of[np.maximum(seg2, 0)], seg2).reshape(h, w)
    return seg2.reshape(h, w), fin, I


def test_superpixel_merge_from_the_source(orc, gputest_pair):
    """a21 on the reference's RGB-D pair: the oracle's SLIC labels (pinned to the reference's own code elsewhere) through mergeSuperPixel -- depth Gaussian, position /
    normal maps, the re-clustering statistics (kernels A-E), connectSuperPixel, the final ids -- and maskSuperPixelFilter_OverSeg (:651-710), restated above from the
    source.  Region ids are integers decided by float comparisons of sums: the two readings agree on all but a handful of threshold ties (f64 sums here, exact
    fixed-point sums in the oracle), the per-superpixel statistics to float rounding."""
    depth, rgb, _, w, h = _pair(gputest_pair)
    o = orc.Oracle(w=w, h=h, fx=HALF_K["fx"], fy=HALF_K["fy"], cx=HALF_K["cx"], cy=HALF_K["cy"], max_surfels=100000)
    seg0, nsp = o.slic_segment(rgb)
    spn = (w * h) // 256
    seg_o, fin_o, info_o = o.merge_superpixels(depth, seg0)
    cam = (np.float32(HALF_K["cx"]), np.float32(HALF_K["cy"]), np.float32(1.0 / HALF_K["fx"]), np.float32(1.0 / HALF_K["fy"]))
    seg_r, fin_r, info_r = _sp_merge_restated(depth, seg0, w, h, spn, cam)
    assert (seg_o >= 0).mean() > 0.5 and len(np.unique(fin_o)) > 5                      # the pair has valid depth and more than a region or two
    assert np.array_equal(seg_o < 0, seg_r < 0) or ((seg_o < 0) != (seg_r < 0)).mean() < 1e-3
    assert (seg_o != seg_r).mean() < 2e-3, (seg_o != seg_r).mean()                      # re-clustered labels
    used = info_o[:, 0] > 50
    assert np.abs(info_o[used, 0] - info_r[used, 0]).max() <= 0.02 * info_o[used, 0].max()
    assert np.allclose(info_o[used, 7:15], info_r[used, 7:15], rtol=2e-3, atol=2e-3)    # averages: position, normal, depth
    # regions: the same partition up to the few ties -- compared as a partition (ids are representatives, and a tie can rename a whole region)
    both = (fin_o >= 0) & (fin_r >= 0)
    pairs = np.stack([fin_o[both], fin_r[both]], 1)
    uniq, cnt = np.unique(pairs, axis=0, return_counts=True)
    best_for_o = {}
    for (a_, b_), c_ in zip(uniq.tolist(), cnt.tolist()):
        if c_ > best_for_o.get(a_, (None, 0))[1]:
            best_for_o[a_] = (b_, c_)
    agree = sum(c_ for (a_, b_), c_ in zip(uniq.tolist(), cnt.tolist()) if best_for_o[a_][0] == b_)
    assert agree / both.sum() > 0.98, agree / both.sum()
    # maskSuperPixelFilter_OverSeg on the ORACLE's regions (so that the filter itself is what is compared): a mask keeps a region iff > 75 % of the region lies inside it
    rng = np.random.RandomState(5)
    masks = np.zeros((3, h, w), np.uint8)
    masks[0, 40:160, 60:200] = 255; masks[1, 100:220, 150:300] = 255; masks[2] = (rng.rand(h, w) < 0.5) * 255
    got = o.mask_superpixel_filter(fin_o, masks)
    want = np.zeros_like(masks)
    for rid in np.unique(fin_o[fin_o >= 0]):
        if rid >= spn:
            continue
        reg = fin_o == rid
        n_ = int(reg.sum())
        for m_ in range(3):
            if np.float32(int((masks[m_][reg] != 0).sum()) * np.float32(1.0) / np.float32(n_)) > 0.75:
                want[m_][reg] = 255
    assert np.array_equal(got, want)
    o.close()
