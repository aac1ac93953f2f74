"""The map passes against the reference's OWN GLSL, executed (tests/golden/gl_map_passes.npz).

The golden vectors are outputs of the UNMODIFIED shader files of /root/reference/elasticfusionpublic/Core/src/Shaders -- depth_bilateral.frag, depth_metric.frag,
index_map.vert/.frag, data.vert/.geom/.frag, update.vert, copy_unstable.vert/.geom, surfel_ids.vert/.geom/.frag, splat.vert + combo_splat.frag, fill_*.frag --
run on Mesa's software rasteriser through a window-less GL 4.5 context (oracle/gl, tools/make_golden_gl.py; the generator and what is and is not the reference's in
it are described there).  One frame's map stage on a 20 811-surfel map at 160x120: preprocessing, index map, association + fusion, index map again, clean,
surfel-id render, splat prediction + fill-in.  Every stage of the generator ran the reference's shader on the ORACLE's input state for that stage, so a stage is
compared on identical inputs and a disagreement does not leak into the next stage.

Both the CPU oracle (this file's `-m "not gpu"` half) and the HIP path through its C-ABI stage calls (`-m gpu`) are held to the same agreement floors.  What the
floors mean: per-element arithmetic (vertex maps, fused positions, normals) agrees to float rounding (llvmpipe evaluates the shaders with LLVM's x86 float code, the
oracle without FMA contraction: last-bit differences); WHICH element lands where agrees up to the freedom an OpenGL implementation has -- sub-pixel snapping of
point positions (1/256 px in llvmpipe), the 24-bit depth buffer against f32 keys, nearest-texel selection for coordinates that sit exactly on a texel edge.  The
measured agreement is printed by the generator and asserted here with a small margin."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "gl_map_passes.npz")
MAP_KEYS = ("pc", "nr", "col", "tm", "ic", "votes")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


class Stages:
    """One frame's map stage through the stage API of either implementation (same names on both but for three)."""

    def __init__(self, impl, kind, gold):
        self.x, self.kind, self.g = impl, kind, gold
        self.pose, self.t = gold["pose"], int(gold["time"])

    def image(self, name):
        return self.x.image(name)

    def set_frame(self):
        self.x.set_frame(self.g["rgb"], self.g["depth"])

    def index(self):
        self.x.predict_indices(self.pose, self.t)
        return {k: self.image(k) for k in ("index", "index_vc", "index_nr")}

    def fuse(self):
        self.x.fuse(self.pose, self.t, 1.0)

    def clean(self):
        self.x.predict_indices(self.pose, self.t)      # (EF/ElasticFusion.cpp:662: the index map of the post-fuse map is the clean pass's input)
        self.x.clean(self.pose, self.t)

    def ids(self):
        r = self.x.render_ids(self.pose, 0)
        return r if self.kind == "oracle" else self.image("ids_tmp")

    def predict(self):
        self.x.combined_predict(self.pose, self.t, self.t)
        return {k: self.image(k) for k in ("pred_vertex", "pred_normal", "pred_image", "pred_time", "fill_vertex", "fill_normal", "fill_image")}


def survivors_of(before_pc, after_pc):
    keep, j = [], 0
    for i in range(before_pc.shape[0]):
        if j < after_pc.shape[0] and np.array_equal(before_pc[i, :3], after_pc[j, :3]):
            keep.append(i); j += 1
    return np.array(keep, np.int64), j


def run_and_compare(s, gold):
    """Returns the agreement figures (and asserts the floors)."""
    g = gold
    n = g["map_pc"].shape[0]
    t = np.float32(int(g["time"]))
    out = {}
    # ---- a2
    s.set_frame()
    df = np.abs(s.image("depth_filtered").astype(np.int32) - g["gl_depth_filtered"].astype(np.int32))
    out["a2 bilateral: pixels equal %"] = float((df == 0).mean() * 100)
    out["a2 bilateral: max difference mm"] = int(df.max())
    assert np.array_equal(s.image("depth_metric"), g["gl_depth_metric"])                       # depth_metric.frag: exact
    # ---- a10
    im = s.index()
    same = im["index"].astype(np.uint32) == g["gl_pre_index"]
    both = same & (g["gl_pre_index"] > 0)
    out["a10 index map: ids equal %"] = float(same.mean() * 100)
    out["a10 index map: max |vertex diff| where ids agree"] = float(np.abs(im["index_vc"][both] - g["gl_pre_index_vc"][both]).max())
    out["a10 index map: max |normal diff| where ids agree"] = float(np.abs(im["index_nr"][both] - g["gl_pre_index_nr"][both]).max())
    # ---- a11 + a12
    s.fuse()
    mf = s.x.download()
    mine = (mf["tm"][:, 1] == t) & (g["map_tm"][:, 1] != t)
    theirs = g["gl_fuse_updated"]
    out["a11 association: surfels matched by both / by either %"] = float((mine & theirs).sum() / max((mine | theirs).sum(), 1) * 100)
    b = mine & theirs
    d = np.abs(mf["pc"][b, :3] - g["gl_fused_pc"][b, :3]).max(axis=1)
    out["a12 fusion: positions within 1e-5 m where both matched %"] = float((d < 1e-5).mean() * 100)
    out["a12 fusion: confidences within 1e-5 where both matched %"] = float((np.abs(mf["pc"][b, 3] - g["gl_fused_pc"][b, 3]) < 1e-5).mean() * 100)
    dn = np.abs(mf["nr"][b] - g["gl_fused_nr"][b]).max(axis=1)
    out["a12 fusion: normals + radii within 1e-5 where both matched %"] = float((dn < 1e-5).mean() * 100)
    out["a12 fusion: colours equal where both matched %"] = float((mf["col"][b, 0] == g["gl_fused_col"][b, 0]).mean() * 100)
    un = ~mine & ~theirs
    assert np.array_equal(mf["pc"][un], g["gl_fused_pc"][un]) and np.array_equal(mf["nr"][un], g["gl_fused_nr"][un])          # untouched surfels pass through bit for bit
    # ---- a13
    s.clean()
    mc = s.x.download()
    keep, nk = survivors_of(mf["pc"], mc["pc"])
    theirs_keep = g["gl_clean_kept"].astype(np.int64)
    inter = np.intersect1d(keep, theirs_keep).size
    out["a13 clean: survivors common / union %"] = float(inter / np.union1d(keep, theirs_keep).size * 100)
    out["a13 clean: removed here / by the reference's shader"] = (int(n - nk), int(n - theirs_keep.size))
    out["a13 clean: new surfels here / reference"] = (int(mc["pc"].shape[0] - nk), int(g["gl_clean_new_pc"].shape[0]))
    # new surfels: matched by their creating pixel (imgCorr.xy)
    mine_new = {(int(np.floor(r[0])), int(np.floor(r[1]))): k for k, r in enumerate(mc["ic"][nk:])}      # (the creating pixel: vImgCorr.xy = texcoord * size, data.vert:208-209)
    hit, worst = 0, 0.0
    for k, r in enumerate(g["gl_clean_new_ic"]):
        j = mine_new.get((int(np.floor(r[0])), int(np.floor(r[1]))))
        if j is not None:
            hit += 1
            worst = max(worst, float(np.abs(mc["pc"][nk + j] - g["gl_clean_new_pc"][k]).max()), float(np.abs(mc["nr"][nk + j] - g["gl_clean_new_nr"][k]).max()))
    out["a13 clean: new surfels of the reference also created here %"] = float(hit / max(g["gl_clean_new_pc"].shape[0], 1) * 100)
    out["a11 new surfels: max |diff| of position / normal / radius where both created"] = worst
    # ---- a14
    ids = s.ids()
    out["a14 surfel ids: pixels equal %"] = float((ids == g["gl_ids"]).mean() * 100)
    out["a14 surfel ids: coverage differs (pixels)"] = int(((ids > 0) != (g["gl_ids"] > 0)).sum())
    # ---- a9
    p = s.predict()
    cov, cov_g = p["pred_vertex"][..., 2] != 0, g["gl_pred_vertex"][..., 2] != 0
    out["a9 splat: coverage differs (pixels)"] = int((cov != cov_g).sum())
    b = cov & cov_g
    dz = np.abs(p["pred_vertex"][..., :3] - g["gl_pred_vertex"][..., :3]).max(axis=-1)[b]
    out["a9 splat: vertices within 1e-5 m %"] = float((dz < 1e-5).mean() * 100)
    out["a9 splat: vertices within 1 mm %"] = float((dz < 1e-3).mean() * 100)
    out["a9 splat: normals within 1e-5 %"] = float((np.abs(p["pred_normal"][..., :3] - g["gl_pred_normal"][..., :3]).max(axis=-1)[b] < 1e-5).mean() * 100)
    out["a9 splat: colours equal %"] = float((p["pred_image"][b][:, :3] == g["gl_pred_image"][b][:, :3]).all(axis=1).mean() * 100)
    out["a9 splat: init times equal %"] = float((p["pred_time"][b] == g["gl_pred_time"][b]).mean() * 100)
    fin = np.isfinite(p["fill_normal"]).all(axis=-1) & np.isfinite(g["gl_fill_normal"]).all(axis=-1)
    out["a9 fill-in: vertices within 1e-5 m %"] = float((np.abs(p["fill_vertex"][..., :3] - g["gl_fill_vertex"][..., :3]).max(axis=-1) < 1e-5).mean() * 100)
    out["a9 fill-in: normals within 1e-5 %"] = float((np.abs(p["fill_normal"][..., :3] - g["gl_fill_normal"][..., :3]).max(axis=-1)[fin] < 1e-5).mean() * 100)
    out["a9 fill-in: image equal %"] = float((p["fill_image"][..., :3] == g["gl_fill_image"][..., :3]).all(axis=-1).mean() * 100)
    return out


# agreement floors: the measured values (profiles/r06_gl_agreement.txt) less a margin for the rounding of the figures.  What is NOT at 100 % and why:
#   a2   0.04 % of the pixels differ by 1 mm: GLSL's exp against the shared deterministic expf (<= 2 ulp), decided at a rounding boundary of the filtered depth;
#   a14  1 % of the pixels: the reference draws a surfel's id as a screen-space quad of two triangles with AFFINE texture coordinates (surfel_ids.geom: w = 1) and keeps
#        the unit disc of those; here (DESIGN.md "deviations") coverage is the ray-disc intersection through the pixel centre -- equal but for the outermost fraction of a
#        pixel of every disc's rim, where a neighbouring surfel shows instead;
#   a9   0.02 % of the pixels: two surfels within the 24-bit depth buffer's resolution of each other (ties fall to the first drawn there, to the nearer f32 depth here).
FLOORS = {
    "a2 bilateral: pixels equal %": 99.9,
    "a10 index map: ids equal %": 99.99,
    "a11 association: surfels matched by both / by either %": 99.9,
    "a12 fusion: positions within 1e-5 m where both matched %": 99.9,
    "a12 fusion: confidences within 1e-5 where both matched %": 99.9,
    "a12 fusion: normals + radii within 1e-5 where both matched %": 99.9,
    "a12 fusion: colours equal where both matched %": 99.9,
    "a13 clean: survivors common / union %": 99.99,
    "a13 clean: new surfels of the reference also created here %": 99.0,
    "a14 surfel ids: pixels equal %": 98.8,
    "a9 splat: vertices within 1e-5 m %": 99.9,
    "a9 splat: vertices within 1 mm %": 99.9,
    "a9 splat: normals within 1e-5 %": 99.9,
    "a9 splat: colours equal %": 99.9,
    "a9 splat: init times equal %": 99.9,
    "a9 fill-in: vertices within 1e-5 m %": 99.9,
    "a9 fill-in: normals within 1e-5 %": 99.9,
    "a9 fill-in: image equal %": 99.9,
}
CEILINGS = {"a2 bilateral: max difference mm": 1, "a10 index map: max |vertex diff| where ids agree": 1e-5, "a10 index map: max |normal diff| where ids agree": 1e-6,
            "a14 surfel ids: coverage differs (pixels)": 10, "a9 splat: coverage differs (pixels)": 2, "a11 new surfels: max |diff| of position / normal / radius where both created": 1e-5}


def check(out, who):
    print(f"\n{who} against the reference's GLSL run on Mesa llvmpipe:")
    for k, v in out.items():
        print(f"   {k:78s} {v}")
    for k, lo in FLOORS.items():
        assert out[k] >= lo, (k, out[k], lo)
    for k, hi in CEILINGS.items():
        assert out[k] <= hi, (k, out[k], hi)


def _handle(mod, cls, gold, **kw):
    W, H = int(gold["width"]), int(gold["height"])
    fx, fy, cx, cy = map(float, gold["K"])
    x = cls(w=W, h=H, fx=fx, fy=fy, cx=cx, cy=cy, max_surfels=200000, confidence=float(gold["confidence"]), **kw)
    m = {k: gold["map_" + k] for k in MAP_KEYS}
    x.upload(m)
    x.set_pose(gold["pose"], int(gold["time"]))
    return x


def deformation_and_compare(x, gold):
    """f-3: a deformation graph applied by the clean pass (copy_unstable.vert:176-330: binary search of the node times, the k = 4 nearest of 20 nodes, weighted rigid
    transforms, renormalised normal; stable surfels in front of the synthesised INACTIVE depth are re-activated) -- the survivors' positions, normals and last-seen times
    against the reference's shader on the same map (time window 6 frames: the map has inactive surfels) and the same 70-node graph."""
    g = gold
    t = int(g["d_time"])
    x.predict_indices(g["d_pose"], t)
    x.set_deformation(g["d_graph"], False)
    x.clean(g["d_pose"], t)
    m = x.download()
    out = {"f-3 deformation: survivors here / reference": (int(m["pc"].shape[0]), int(g["gl_d_kept"].shape[0]))}
    assert m["pc"].shape[0] == g["gl_d_kept"].shape[0]
    k = g["gl_d_kept"].astype(np.int64)
    assert np.array_equal(m["pc"][:, 3], g["d_map_pc"][k, 3]) and np.array_equal(m["tm"][:, 0], g["d_map_tm"][k, 0])        # the same surfels survive, in order
    out["f-3 deformation: positions within 1e-5 m %"] = float((np.abs(m["pc"][:, :3] - g["gl_d_pc"][:, :3]).max(axis=1) < 1e-5).mean() * 100)
    out["f-3 deformation: normals within 1e-4 %"] = float((np.abs(m["nr"][:, :3] - g["gl_d_nr"][:, :3]).max(axis=1) < 1e-4).mean() * 100)
    out["f-3 deformation: last-seen times equal %"] = float((m["tm"][:, 1] == g["gl_d_tm"][:, 1]).mean() * 100)
    out["f-3 deformation: largest displacement m"] = float(np.abs(m["pc"][:, :3] - g["d_map_pc"][k, :3]).max())
    out["f-3 deformation: surfels re-activated"] = int(((m["tm"][:, 1] == t) & (g["d_map_tm"][k, 1] != t)).sum())
    assert out["f-3 deformation: positions within 1e-5 m %"] >= 99.9 and out["f-3 deformation: normals within 1e-4 %"] >= 99.9 and out["f-3 deformation: last-seen times equal %"] >= 99.9
    assert out["f-3 deformation: largest displacement m"] > 0.02 and out["f-3 deformation: surfels re-activated"] > 100
    return out


def _handle_d(cls, gold, **kw):
    W, H = int(gold["width"]), int(gold["height"])
    fx, fy, cx, cy = map(float, gold["K"])
    x = cls(w=W, h=H, fx=fx, fy=fy, cx=cx, cy=cy, max_surfels=200000, confidence=float(gold["confidence"]), time_delta=int(gold["d_time_delta"]), **kw)
    n = gold["d_map_pc"].shape[0]
    x.upload(dict(pc=gold["d_map_pc"], nr=gold["d_map_nr"], col=gold["d_map_col"], tm=gold["d_map_tm"], ic=np.zeros((n, 4), np.float32), votes=np.zeros((n, 48), np.float32)))
    x.set_pose(gold["d_pose"], int(gold["d_time"]))
    return x


def test_oracle_against_the_reference_shaders(gold, orc):
    orc.set_threads(orc.usable_cores())
    o = _handle(orc, orc.Oracle, gold)
    check(run_and_compare(Stages(o, "oracle", gold), gold), "CPU oracle")
    o.close()
    od = _handle_d(orc.Oracle, gold)
    od.set_loop_closure(True)      # (allocates the INACTIVE prediction's buffers, where a deforming clean synthesises its depth image)
    for k, v in deformation_and_compare(od, gold).items():
        print(f"   {k:78s} {v}")
    od.close()


@pytest.mark.gpu
def test_hip_path_against_the_reference_shaders(gold):
    import instancefusion_amd as ifx

    ifx.lib()
    g = _handle(ifx, ifx.ElasticFusion, gold)
    g.set_option("compact_every_frame", 1)      # (slot numbers = map indices: the id images name surfels as the reference does)
    check(run_and_compare(Stages(g, "hip", gold), gold), "HIP path (C-ABI stage calls)")
    g.close()
    gd = _handle_d(ifx.ElasticFusion, gold)
    gd.set_option("compact_every_frame", 1)
    gd.set_loop_closure(True)
    for k, v in deformation_and_compare(gd, gold).items():
        print(f"   {k:78s} {v}")
    gd.close()
