#!/usr/bin/env python3
"""Golden vectors of the reference's OWN GLSL for the map passes, produced by running the shader files of
/root/reference/elasticfusionpublic/Core/src/Shaders UNMODIFIED on Mesa's software rasteriser (oracle/gl: a window-less GL 4.5 core context through the DRI swrast
loader interface; `make -C oracle ref` builds oracle/_ref/libdri_ctx.so).  Run in the development container only (the reference tree is absent on the GPU box):

    python tools/make_golden_gl.py            # writes tests/golden/gl_*.npz and prints how the CPU oracle compares

What is the reference's and what is not: every formula that decides a result -- culls, projections, the association window and its gates, the fusion update, the clean
rules, the splat intersection, the id quads, the bilateral weights -- is executed from the reference's shader text.  The GL calls AROUND the shaders (buffers, attribute
pointers, textures and their filters, framebuffer attachments, depth test, transform feedback, draw calls) restate the reference's host code and cite it; they are not
its code (Pangolin / GLEW / CUDA interop do not build here).  Inputs: the seeded synthetic stream at 160x120 run through the CPU oracle for a few frames (a map with
stable and unstable surfels, a pose, the next frame).  Outputs are compared by tests/test_gl_golden.py with the oracle (CPU suite) and the HIP path (GPU suite)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "gl"))
SHADERS = "/root/reference/elasticfusionpublic/Core/src/Shaders"
OUT = os.path.join(ROOT, "tests", "golden")

import glmini as G  # noqa: E402

W, H = 160, 120
K = dict(fx=132.0, fy=132.0, cx=80.0, cy=60.0)
CONF, MAX_DEPTH, TIME_DELTA = 3.0, 20.0, 200
VSIZE = 256                  # Vertex::SIZE, EF/Shaders/Vertex.cpp:46: sixteen vec4
TEXDIM = 1536                # GlobalModel::TEXTURE_DIMENSION, EF/GlobalModel.cpp:22


def pack_vbo(m):
    """the reference's vertex record (EF/Shaders/Vertex.cpp:20-46): position + confidence | colour, instance colour, initTime, timestamp | normal + radius | imgCorr | 12 vote vec4"""
    n = m["pc"].shape[0]
    v = np.zeros((n, 64), np.float32)
    v[:, 0:4] = m["pc"]
    v[:, 4:6] = m["col"]
    v[:, 6:8] = m["tm"]
    v[:, 8:12] = m["nr"]
    v[:, 12:16] = m["ic"]
    v[:, 16:64] = m["votes"]
    return v


def unpack_vbo(v):
    return dict(pc=v[:, 0:4].copy(), col=v[:, 4:6].copy(), tm=v[:, 6:8].copy(), nr=v[:, 8:12].copy(), ic=v[:, 12:16].copy(), votes=v[:, 16:64].copy())


class RefGL:
    """The reference's GL objects for the map passes, with its shader files."""

    def __init__(self):
        self.gl = gl = G.GL(W, H)
        print("GL:", *gl.version())
        gl.glEnable(G.GL_DEPTH_TEST)      # Gui::preCall, IF/gui/Gui.cpp:211-215 (in force from the second frame on)
        gl.glDepthFunc(G.GL_LESS)
        f4 = lambda: gl.tex2d(W, H, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT)
        # IndexMap::IndexMap, EF/IndexMap.cpp:23-39, 143-147
        self.index_prog = gl.program(SHADERS, "index_map.vert", "index_map.frag")
        self.index_tex = gl.tex2d(W, H, G.GL_R32UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_INT)
        self.vert_conf_tex, self.color_time_tex, self.norm_rad_tex = f4(), f4(), f4()
        self.index_fbo = gl.framebuffer(W, H, [self.index_tex, self.vert_conf_tex, self.color_time_tex, self.norm_rad_tex])

    # IndexMap::predictIndices, EF/IndexMap.cpp:221-279
    def predict_indices(self, vbo, n, pose, time):
        gl = self.gl
        gl.begin_pass(self.index_fbo, W, H, "ufff")
        gl.uniforms(self.index_prog, t_inv=np.linalg.inv(pose.astype(np.float32)), cam=[K["cx"], K["cy"], K["fx"], K["fy"]], maxDepth=MAX_DEPTH, cols=float(W), rows=float(H),
                    time=int(time), timeDelta=int(TIME_DELTA))
        gl.attribs(vbo, 3, VSIZE)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.attribs_off(3)
        gl.end_pass()
        return dict(index=gl.read_tex(self.index_tex, W, H, G.GL_RED_INTEGER, G.GL_UNSIGNED_INT, np.uint32, 1),
                    index_vc=gl.read_tex(self.vert_conf_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4),
                    index_ct=gl.read_tex(self.color_time_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4),
                    index_nr=gl.read_tex(self.norm_rad_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4))


def main():
    import oracle_lib as ol
    from instancefusion_amd import synth

    ol.build()
    NF = 8
    st = synth.make_stream(NF + 1, W, H, noise=True, **K)
    o = ol.Oracle(w=W, h=H, max_surfels=200000, confidence=CONF, **K)
    pose = None
    for i in range(NF):
        pose = o.process_frame(st["rgb"][i], st["depth"][i])
    m = o.download()
    tick = o.tick
    n = m["pc"].shape[0]
    print(f"map: {n} surfels, {(m['pc'][:, 3] >= CONF).sum()} stable, tick {tick}")
    ref = RefGL()
    gl = ref.gl
    vbo = gl.buffer(pack_vbo(m))
    gold = dict(map_pc=m["pc"], map_col=m["col"], map_tm=m["tm"], map_nr=m["nr"], map_ic=m["ic"], pose=pose.astype(np.float32), tick=np.int32(tick))
    # ---- a10 index map
    gi = ref.predict_indices(vbo, n, pose, tick)
    o.predict_indices(pose, tick)
    oi = {k: o.image(k) for k in ("index", "index_vc", "index_ct", "index_nr")}
    same = gi["index"] == oi["index"].astype(np.uint32)
    print(f"index map: ids equal on {same.mean() * 100:.3f} % of the pixels ({(~same).sum()} differ; GL draws {int((gi['index'] > 0).sum())}, oracle {int((oi['index'] > 0).sum())})")
    both = same & (gi["index"] > 0)
    for k in ("index_vc", "index_ct", "index_nr"):
        d = np.abs(gi[k][both] - oi[k][both])
        print(f"   {k}: max |diff| where the ids agree {d.max():.3e}, bit-equal on {(d == 0).all(axis=1).mean() * 100:.2f} %")
    gl.close()


if __name__ == "__main__":
    main()
