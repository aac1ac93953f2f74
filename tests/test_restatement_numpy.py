"""A SECOND, independently written restatement of the stages the reference cannot pin: the tracker's normal-equation rows
(ICP, photometric residual + row), the assembly of the Gauss-Newton system and the measurement model of the fusion shaders, written
in vectorised f64 numpy straight from the formulas of SURVEY.md Appendix C (C.2, C.8, C.9, C.10) -- not from oracle/*.c -- and
compared with the C oracle on the only real data the reference ships, the GPUTest RGB-D pair.  It does not pin the oracle to the
reference (nothing can, here); it removes common-mode reading errors between the oracle and the HIP path, which share an author.
No GPU needed."""
import ctypes as C

import numpy as np

from conftest import SMALL  # noqa: F401  (path set-up)
from gputest_protocol import HALF_K, protocol_inputs


def _buf(orc, t, name, level, dtype, shape):
    p = orc.lib().orc_tracker_buffer(t, name.encode(), level)
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_char * n).from_address(p), dtype).reshape(shape).copy()


def _tracker(orc, gputest_pair):
    L = orc.lib()
    h, w = gputest_pair[1].shape
    V, N, rgba, prev, depth_mm, rgb = protocol_inputs(*gputest_pair)
    t = L.orc_tracker_create(w, h, HALF_K["fx"], HALF_K["fy"], HALF_K["cx"], HALF_K["cy"])
    L.orc_tracker_init_first_rgb(t, orc.ptr(prev))
    p0 = np.eye(4, dtype=np.float32).reshape(16).copy()
    L.orc_tracker_init_model(t, orc.ptr(V), orc.ptr(N), orc.ptr(rgba), orc.ptr(p0))
    L.orc_tracker_init_frame(t, orc.ptr(depth_mm), orc.ptr(rgb), 20.0)
    return t, w, h


def _sums29(J, r, found):
    """the 27 upper-triangle products of [J | r] in the reference's order, then sum r^2 and the count"""
    rows = np.concatenate([J, r[:, None]], 1)[found]
    out = []
    for i in range(6):
        for j in range(i, 7):
            out.append(np.sum(rows[:, i] * rows[:, j]))
    out.append(np.sum(rows[:, 6] ** 2))
    out.append(float(found.sum()))
    return np.array(out)


def test_icp_rows_from_the_formulas(orc, gputest_pair):
    """C.8: s = Rc v + tc, s' = Rp^-1 (s - tp), pixel = rn(K s'), gates (in bounds, s'_z >= 0, |n_c^g x n_d| < sin 20, |d - s| <= 0.10),
    J = [n', s' x n'], r = n' . (s' - d') with primes rotated by Rp^-1."""
    L = orc.lib()
    t, w, h = _tracker(orc, gputest_pair)
    fx, fy, cx, cy = (HALF_K[k] for k in ("fx", "fy", "cx", "cy"))
    vc = _buf(orc, t, "vmap_curr", 0, np.float32, (3, h, w)).astype(np.float64).reshape(3, -1).T
    nc = _buf(orc, t, "nmap_curr", 0, np.float32, (3, h, w)).astype(np.float64).reshape(3, -1).T
    vp = _buf(orc, t, "vmap_prev", 0, np.float32, (3, h, w)).astype(np.float64).reshape(3, -1)
    npv = _buf(orc, t, "nmap_prev", 0, np.float32, (3, h, w)).astype(np.float64).reshape(3, -1)
    ang = 0.012
    Rc = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    tc = np.array([0.003, -0.004, 0.002])
    Rp, tp = np.eye(3), np.zeros(3)
    s = vc @ Rc.T + tc
    sp = (s - tp) @ np.linalg.inv(Rp).T
    with np.errstate(all="ignore"):
        u = np.rint(sp[:, 0] * fx / sp[:, 2] + cx)
        v = np.rint(sp[:, 1] * fy / sp[:, 2] + cy)
    ok = np.isfinite(vc[:, 0]) & np.isfinite(u) & np.isfinite(v) & (u >= 0) & (v >= 0) & (u < w) & (v < h) & (sp[:, 2] >= 0)
    idx = np.where(ok, v * w + u, 0).astype(np.int64)
    d = vp[:, idx].T
    nd = npv[:, idx].T
    ncg = nc @ Rc.T
    sine = np.linalg.norm(np.cross(ncg, nd), axis=1)
    dist = np.linalg.norm(d - s, axis=1)
    found = ok & (sine < np.sin(np.deg2rad(20.0))) & (dist <= 0.10) & np.isfinite(nc[:, 0]) & np.isfinite(nd[:, 0])
    Ri = np.linalg.inv(Rp)
    n_ = nd @ Ri.T
    d_ = (d - tp) @ Ri.T
    J = np.concatenate([n_, np.cross(sp, n_)], 1)
    r = np.sum(n_ * (sp - d_), axis=1)
    want = _sums29(np.nan_to_num(J), np.nan_to_num(r), found)
    got = np.zeros(29, np.float32)
    L.orc_icp_step.argtypes = [C.c_void_p] * 6 + [C.c_float] * 4 + [C.c_void_p] * 2 + [C.c_float] * 2 + [C.c_int] * 2 + [C.c_void_p]
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    vcb, ncb = _buf(orc, t, "vmap_curr", 0, np.float32, (3, h, w)), _buf(orc, t, "nmap_curr", 0, np.float32, (3, h, w))
    vpb, npb = _buf(orc, t, "vmap_prev", 0, np.float32, (3, h, w)), _buf(orc, t, "nmap_prev", 0, np.float32, (3, h, w))
    L.orc_icp_step(orc.ptr(f32(Rc).reshape(9)), orc.ptr(f32(tc)), orc.ptr(vcb), orc.ptr(ncb), orc.ptr(f32(np.eye(3)).reshape(9)), orc.ptr(f32(tp)),
                   fx, fy, cx, cy, orc.ptr(vpb), orc.ptr(npb), 0.10, np.float32(np.sin(20.0 * 3.14159254 / 180.0)), w, h, orc.ptr(got))
    assert want[28] > 5000 and abs(got[28] - want[28]) <= 0.002 * want[28]        # the inlier sets agree up to threshold ties (f32 vs f64 evaluation)
    scale = np.sqrt(np.abs(np.outer(np.r_[want[[0, 7, 13, 18, 22, 25]], want[27]], np.r_[want[[0, 7, 13, 18, 22, 25]], want[27]])))
    k = 0
    for i in range(6):
        for j in range(i, 7):
            assert abs(got[k] - want[k]) <= 3e-3 * scale[i, j] + 1e-9, (i, j, got[k], want[k])
            k += 1
    assert abs(got[27] - want[27]) <= 3e-3 * want[27]
    L.orc_tracker_destroy(t)


def test_photometric_rows_from_the_formulas(orc, gputest_pair):
    """C.9: for next-image pixels with gradient^2 >= (g_min / sobelScale)^2 and a fully non-zero 4x4 neighbourhood, warp with
    d1 (K R K^-1 (x, y, 1)) + K t, require d0 > 0, |z' - d0| <= 0.07, last intensity != 0 -> diff = I_next - I_last; row with
    w = 1 / (sigma + |diff|), X the model point: v0 = w s dIdx fx / Xz, v1 = w s dIdy fy / Xz, v2 = -(v0 Xx + v1 Xy) / Xz,
    J = [v0, v1, v2, -Xz v1 + Xy v2, Xz v0 - Xx v2, -Xy v0 + Xx v1], r = -w diff."""
    L = orc.lib()
    t, w, h = _tracker(orc, gputest_pair)
    fx, fy, cx, cy = (HALF_K[k] for k in ("fx", "fy", "cx", "cy"))
    nxt = _buf(orc, t, "next_img", 0, np.uint8, (h, w))
    lst = _buf(orc, t, "last_img", 0, np.uint8, (h, w))
    ld = _buf(orc, t, "last_depth", 0, np.float32, (h, w))
    didx, didy = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
    L.orc_sobel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_sobel(orc.ptr(nxt), w, h, orc.ptr(didx), orc.ptr(didy))
    cloud = np.zeros((h, w, 3), np.float32)
    L.orc_project_cloud.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p]
    L.orc_project_cloud(orc.ptr(ld), w, h, fx, fy, cx, cy, orc.ptr(cloud))
    ang = 0.008
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tt = np.array([0.004, -0.002, 0.003])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    krk = (K @ R.T @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ (-R.T @ tt)).astype(np.float32)
    # --- numpy
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    border = 16
    inside = (yy >= border) & (yy < h - border) & (xx >= border) & (xx < w - border) & (xx < w - 5) & (yy < h - 1)
    nz = (nxt > 0)
    full = np.ones_like(nz)
    for a in range(-2, 2):
        for b in range(-2, 2):
            full &= np.roll(np.roll(nz, -a, 0), -b, 1)
    g2 = didx.astype(np.float64) ** 2 + didy.astype(np.float64) ** 2
    min_scale = 5.0 ** 2 / (1 / 8.0) ** 2
    d1 = ld.astype(np.float64)            # nextDepth == lastDepth in the frame-to-model tracker (reference quirk)
    kk = krk.astype(np.float64)
    ktd = kt.astype(np.float64)
    with np.errstate(all="ignore"):
        tz = d1 * (kk[2, 0] * xx + kk[2, 1] * yy + kk[2, 2]) + ktd[2]
        u0 = np.rint((d1 * (kk[0, 0] * xx + kk[0, 1] * yy + kk[0, 2]) + ktd[0]) / tz)
        v0 = np.rint((d1 * (kk[1, 0] * xx + kk[1, 1] * yy + kk[1, 2]) + ktd[1]) / tz)
    cand = inside & full & (g2 >= min_scale) & np.isfinite(d1) & np.isfinite(u0) & np.isfinite(v0) & (u0 >= 0) & (v0 >= 0) & (u0 < w) & (v0 < h)
    ui, vi = np.where(cand, u0, 0).astype(np.int64), np.where(cand, v0, 0).astype(np.int64)
    d0 = ld[vi, ui].astype(np.float64)
    li = lst[vi, ui]
    valid = cand & (d0 > 0) & (np.abs(tz - d0) <= 0.07) & (li != 0)
    diff = nxt.astype(np.float64) - li.astype(np.float64)
    cnt_np, sig_np = int(valid.sum()), int(np.sum(np.floor(diff[valid] ** 2)))
    # --- oracle
    dt_ref = np.zeros((h, w), np.dtype([("zx", np.int16), ("zy", np.int16), ("ox", np.int16), ("oy", np.int16), ("diff", np.float32), ("valid", np.int32)]))
    cnt, sig = C.c_int(), C.c_int()
    L.orc_rgb_residual.argtypes = [C.c_float] + [C.c_void_p] * 7 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_rgb_residual(min_scale, orc.ptr(didx), orc.ptr(didy), orc.ptr(ld), orc.ptr(ld), orc.ptr(lst), orc.ptr(nxt), orc.ptr(dt_ref), 0.07,
                       orc.ptr(kt), orc.ptr(np.ascontiguousarray(krk.reshape(9))), w, h, C.byref(cnt), C.byref(sig))
    assert cnt.value > 3000 and abs(cnt.value - cnt_np) <= 0.003 * cnt_np and abs(sig.value - sig_np) <= 0.01 * sig_np
    agree = (dt_ref["valid"] != 0) & valid
    assert agree.sum() >= 0.995 * cnt_np
    assert np.array_equal(dt_ref["zx"][agree], ui[agree]) and np.array_equal(dt_ref["zy"][agree], vi[agree]) and np.array_equal(dt_ref["diff"][agree], diff[agree].astype(np.float32))
    # --- the row, on the oracle's own correspondences (so that both sums run over the same set)
    ov = dt_ref["valid"] != 0
    sigma = float(np.sqrt(cnt.value))
    X = cloud[dt_ref["zy"].astype(np.int64), dt_ref["zx"].astype(np.int64)].astype(np.float64)
    wgt = 1.0 / (sigma + np.abs(dt_ref["diff"].astype(np.float64)))
    s = 1 / 8.0
    a0 = wgt * s * didx * fx / X[..., 2]
    a1 = wgt * s * didy * fy / X[..., 2]
    a2 = -(a0 * X[..., 0] + a1 * X[..., 1]) / X[..., 2]
    J = np.stack([a0, a1, a2, -X[..., 2] * a1 + X[..., 1] * a2, X[..., 2] * a0 - X[..., 0] * a2, -X[..., 1] * a0 + X[..., 0] * a1], -1)
    r = -wgt * dt_ref["diff"]
    want = _sums29(np.nan_to_num(J.reshape(-1, 6)), np.nan_to_num(r.reshape(-1)), ov.reshape(-1))
    got = np.zeros(29, np.float32)
    L.orc_rgb_step.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orc_rgb_step(orc.ptr(dt_ref), sigma, orc.ptr(cloud), fx, fy, orc.ptr(didx), orc.ptr(didy), 0.125, w, h, orc.ptr(got))
    assert got[28] == want[28]
    diag = np.r_[want[[0, 7, 13, 18, 22, 25]], want[27]]
    k = 0
    for i in range(6):
        for j in range(i, 7):
            assert abs(got[k] - want[k]) <= 1e-4 * np.sqrt(diag[i] * diag[j]) + 1e-6, (i, j, got[k], want[k])   # f32 rows + grid-valued terms against f64 rows
            k += 1
    L.orc_tracker_destroy(t)


def test_gauss_newton_system_from_the_formulas(orc, gputest_pair):
    """C.10: A = A_rgb + 100 A_icp, b = b_rgb + 10 b_icp from the two 29-sum vectors of the last iteration; the covariance the reference
    reads (lastA^-1) against numpy."""
    L = orc.lib()
    t, w, h = _tracker(orc, gputest_pair)
    pose = np.eye(4, dtype=np.float32).reshape(16).copy()
    L.orc_tracker_run(t, orc.ptr(pose), 10.0, 1, 0, 1, None)
    icp = np.frombuffer((C.c_float * 29).from_address(L.orc_tracker_buffer(t, b"last_icp29", 0)), np.float32).astype(np.float64)
    rgb = np.frombuffer((C.c_float * 29).from_address(L.orc_tracker_buffer(t, b"last_rgb29", 0)), np.float32).astype(np.float64)
    lastA = np.frombuffer((C.c_double * 36).from_address(L.orc_tracker_buffer(t, b"lastA", 0)), np.float64).reshape(6, 6)
    lastb = np.frombuffer((C.c_double * 6).from_address(L.orc_tracker_buffer(t, b"lastb", 0)), np.float64)

    def unpack(v):
        A, b = np.zeros((6, 6)), np.zeros(6)
        k = 0
        for i in range(6):
            for j in range(i, 7):
                if j == 6:
                    b[i] = v[k]
                else:
                    A[i, j] = A[j, i] = v[k]
                k += 1
        return A, b

    Ai, bi = unpack(icp)
    Ar, br = unpack(rgb)
    assert np.allclose(lastA, Ar + 100.0 * Ai, rtol=1e-12) and np.allclose(lastb, br + 10.0 * bi, rtol=1e-12)
    cov = np.zeros(36)
    L.orc_tracker_covariance.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_tracker_covariance(t, orc.ptr(cov))
    assert np.allclose(cov.reshape(6, 6), np.linalg.inv(lastA), rtol=1e-8)
    x = np.linalg.solve(lastA, lastb)                        # the last increment is small: the run has converged on this pair
    assert np.linalg.norm(x[:3]) < 5e-4 and np.linalg.norm(x[3:]) < 5e-4
    P = pose.reshape(4, 4).astype(np.float64)
    assert np.allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-5)
    L.orc_tracker_destroy(t)


def test_measurement_model_from_the_formulas(orc):
    """C.2 (data.vert / vertex_feedback.vert): v = backprojection of the RAW metric depth at (i + 0.5, j + 0.5); normal from the FILTERED
    depth by central differences normalize(cross((v_xb + v)/2 - (v_xf + v)/2, (v_yb + v)/2 - (v_yf + v)/2)); radius min(2 rho, rho / |n_z|),
    rho = sqrt(2) z / ((fx + fy) / 2); confidence exp(-(d / 400)^2 / 0.72); colour packed 24 bit.  Compared with the map the oracle builds
    from its first frame (dense initialisation, column-major order)."""
    from instancefusion_amd import synth

    W, H = SMALL["w"], SMALL["h"]
    fx, fy, cx, cy = SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"]
    st = synth.make_stream(1, W, H, fx, fy, cx, cy, noise=True)
    o = orc.Oracle(**SMALL, max_surfels=200000)
    o.process_frame(st["rgb"][0], st["depth"][0])
    m = o.download()
    zr = o.image("depth_metric").astype(np.float64)
    zf = o.image("depth_metric_filtered").astype(np.float64)
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    x, y = ii + 0.5, jj + 0.5

    def vert(z, dx=0, dy=0):
        zz = z[np.clip(jj + dy, 0, H - 1), np.clip(ii + dx, 0, W - 1)]
        return np.stack([(x + dx - cx) * zz / fx, (y + dy - cy) * zz / fy, zz], -1)

    v = vert(zr)
    vf = vert(zf)
    delx = (vert(zf, -1, 0) + vf) / 2 - (vert(zf, 1, 0) + vf) / 2
    dely = (vert(zf, 0, -1) + vf) / 2 - (vert(zf, 0, 1) + vf) / 2
    n = np.cross(delx, dely)
    with np.errstate(all="ignore"):
        n = n / np.linalg.norm(n, axis=-1, keepdims=True)
        rho = np.sqrt(2.0) * vf[..., 2] / ((fx + fy) / 2)
        rad = np.minimum(2 * rho, rho / np.abs(n[..., 2]))
    conf = np.exp(-((np.hypot(x - cx, y - cy) / 400.0) ** 2) / 0.72)
    keep = (zr > 0) & (zr <= 20) & (zf > 0) & (zf <= 20)
    order = np.argsort((ii * H + jj)[keep], kind="stable")           # column-major pixel order (EF/GlobalModel.cpp:103-112)
    assert m["pc"].shape[0] == int(keep.sum())
    assert np.allclose(m["pc"][:, :3], v[keep][order], rtol=2e-6, atol=1e-7)
    assert np.allclose(m["pc"][:, 3], conf[keep][order], rtol=1e-5)
    ok = np.isfinite(n[keep][order]).all(1)
    assert ok.mean() > 0.99
    assert np.allclose(m["nr"][ok, :3], n[keep][order][ok], atol=2e-4)
    assert np.allclose(m["nr"][ok, 3], rad[keep][order][ok], rtol=2e-3)
    rgbp = st["rgb"][0].astype(np.int64)
    packed = (rgbp[..., 0] << 16) + (rgbp[..., 1] << 8) + rgbp[..., 2]
    assert np.array_equal(m["col"][:, 0], packed[keep][order].astype(np.float32))
    o.close()
