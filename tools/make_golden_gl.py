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

W, H = [int(v) for v in os.environ.get("IFX_GL_SIZE", "160x120").split("x")]      # (IFX_GL_SIZE=320x240 / 640x480: one-off runs at the other resolutions; their output is kept
K = dict(fx=132.0 * W / 160, fy=132.0 * W / 160, cx=W / 2.0, cy=H / 2.0)               #  under profiles/, the committed golden file is the 160x120 one)
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
        f4 = lambda w=W, h=H: gl.tex2d(w, h, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT)
        rgba8 = lambda: gl.tex2d(W, H, G.GL_RGBA8, G.GL_RGBA, G.GL_UNSIGNED_BYTE)
        names16 = ["vPosition0", "vColor0", "vNormRad0", "vImgCorr0"] + ["vInstInfo%s0" % c for c in "ABCDEFGHIJKL"]      # EF/GlobalModel.cpp:127-145
        # IndexMap::IndexMap, EF/IndexMap.cpp:23-39, 143-147
        self.index_prog = gl.program(SHADERS, "index_map.vert", "index_map.frag")
        self.index_tex = gl.tex2d(W, H, G.GL_R32UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_INT)
        self.vert_conf_tex, self.color_time_tex, self.norm_rad_tex = f4(), f4(), f4()
        self.index_fbo = gl.framebuffer(W, H, [self.index_tex, self.vert_conf_tex, self.color_time_tex, self.norm_rad_tex])
        # surfel ids, EF/IndexMap.cpp:40-49, 184-186
        self.ids_prog = gl.program(SHADERS, "surfel_ids.vert", "surfel_ids.frag", "surfel_ids.geom")
        self.ids_tex = gl.tex2d(W, H, G.GL_R32I, G.GL_RED_INTEGER, G.GL_INT)
        self.ids_fbo = gl.framebuffer(W, H, [self.ids_tex])
        # combined prediction, EF/IndexMap.cpp:83-111, 164-169
        self.combo_prog = gl.program(SHADERS, "splat.vert", "combo_splat.frag")
        self.image_tex, self.vertex_tex, self.normal_tex, self.inst_tex = rgba8(), f4(), f4(), rgba8()
        self.time_tex = gl.tex2d(W, H, G.GL_R16UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT)
        self.combo_fbo = gl.framebuffer(W, H, [self.image_tex, self.vertex_tex, self.normal_tex, self.time_tex, self.inst_tex])
        # GlobalModel::GlobalModel, EF/GlobalModel.cpp:27-176: data / update / copy_unstable programs with their transform-feedback varyings, the three update maps
        self.data_prog = gl.program(SHADERS, "data.vert", "data.frag", "data.geom", feedback=names16)
        self.update_prog = gl.program(SHADERS, "update.vert", feedback=names16)
        self.unstable_prog = gl.program(SHADERS, "copy_unstable.vert", None, "copy_unstable.geom", feedback=names16 + ["gl_NextBuffer", "deleted_id"])
        self.upd_vc, self.upd_ct, self.upd_nr = f4(TEXDIM, TEXDIM), f4(TEXDIM, TEXDIM), f4(TEXDIM, TEXDIM)
        self.upd_fbo = gl.framebuffer(TEXDIM, TEXDIM, [self.upd_vc, self.upd_ct, self.upd_nr])
        uv = np.zeros((W, H, 2), np.float32)      # the uvo buffer, EF/GlobalModel.cpp:103-119: COLUMN-major pixel order, texel centres
        for i in range(W):
            for j in range(H):
                uv[i, j, 0] = np.float32(np.float64(np.float32(i) / np.float32(W)) + 1.0 / (2 * np.float64(np.float32(W))))
                uv[i, j, 1] = np.float32(np.float64(np.float32(j) / np.float32(H)) + 1.0 / (2 * np.float64(np.float32(H))))
        self.uvo = gl.buffer(uv, usage=G.GL_STATIC_DRAW)
        self.new_unstable = gl.buffer(nbytes=W * H * VSIZE)
        self.count_query, self.delete_query = gl.query(), gl.query()
        # IndexMap::synthesizeDepth (EF/IndexMap.cpp:66-82, 161-162) and the deformation nodes (EF/GlobalModel.cpp:44: a 16384 x 1 float texture, 16 floats per node)
        self.depth_prog = gl.program(SHADERS, "splat.vert", "depth_splat.frag")
        self.syn_depth_tex = gl.tex2d(W, H, G.GL_R32F, G.GL_RED, G.GL_FLOAT)
        self.syn_depth_fbo = gl.framebuffer(W, H, [self.syn_depth_tex])
        self.node_tex = gl.tex2d(16384, 1, G.GL_R32F, G.GL_RED, G.GL_FLOAT, np.zeros((1, 16384), np.float32))
        self.dummy_f = gl.tex2d(4, 4, G.GL_R32F, G.GL_RED, G.GL_FLOAT, np.zeros((4, 4), np.float32))
        # preprocessing, EF/ElasticFusion.cpp:216-229 (ComputePack): one point -> quad.geom -> fragment shader
        self.filter_prog = gl.program(SHADERS, "empty.vert", "depth_bilateral.frag", "quad.geom")
        self.metric_prog = gl.program(SHADERS, "empty.vert", "depth_metric.frag", "quad.geom")
        self.depth_raw_tex = gl.tex2d(W, H, G.GL_R16UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT)
        self.depth_filt_tex = gl.tex2d(W, H, G.GL_R16UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT)
        self.dm_tex = gl.tex2d(W, H, G.GL_R32F, G.GL_RED, G.GL_FLOAT)
        self.dmf_tex = gl.tex2d(W, H, G.GL_R32F, G.GL_RED, G.GL_FLOAT)
        self.filter_fbo = gl.framebuffer(W, H, [self.depth_filt_tex])
        self.dm_fbo = gl.framebuffer(W, H, [self.dm_tex])
        self.dmf_fbo = gl.framebuffer(W, H, [self.dmf_tex])
        self.rgb_tex = gl.tex2d(W, H, G.GL_RGBA8, G.GL_RGB, G.GL_UNSIGNED_BYTE, linear=True)     # EF/ElasticFusion.cpp:174-180: draw = true -> linear sampling
        # FillIn, EF/Shaders/FillIn.cpp:21-58
        self.fill_progs = {}
        for k, frag in (("vertex", "fill_vertex.frag"), ("normal", "fill_normal.frag"), ("image", "fill_rgb.frag")):
            try:
                self.fill_progs[k] = gl.program(SHADERS, "empty.vert", frag, "quad.geom")
            except RuntimeError as e:
                print(f"   ({frag} does not compile in a core context and is left out: {str(e).strip().splitlines()[-1][:140]})")
        self.fill_v_tex, self.fill_n_tex, self.fill_i_tex = f4(), f4(), rgba8()
        self.fill_v_fbo, self.fill_n_fbo, self.fill_i_fbo = gl.framebuffer(W, H, [self.fill_v_tex]), gl.framebuffer(W, H, [self.fill_n_tex]), gl.framebuffer(W, H, [self.fill_i_tex])

    def _upload(self, tex, w, h, internal, fmt, typ, data):
        gl = self.gl
        data = np.ascontiguousarray(data)
        gl.glBindTexture(G.GL_TEXTURE_2D, tex)
        gl.glTexImage2D(G.GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, typ, data.ctypes.data)
        gl.glBindTexture(G.GL_TEXTURE_2D, 0)

    def cam(self, inv=False):
        return [K["cx"], K["cy"], (1.0 / K["fx"]) if inv else K["fx"], (1.0 / K["fy"]) if inv else K["fy"]]

    # ElasticFusion::filterDepth / metriciseDepth, EF/ElasticFusion.cpp:765-784 (ComputePack::compute, EF/Shaders/ComputePack.cpp:42-73)
    def preprocess(self, depth_mm, depth_cut):
        gl = self.gl
        self._upload(self.depth_raw_tex, W, H, G.GL_R16UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT, depth_mm)

        def compute(prog, fbo, src, kind, **un):
            gl.bind_textures([src])
            gl.begin_pass(fbo, W, H, kind)
            gl.uniforms(prog, **un)
            gl.glDrawArrays(G.GL_POINTS, 0, 1)
            gl.end_pass()

        compute(self.filter_prog, self.filter_fbo, self.depth_raw_tex, "u", cols=float(W), rows=float(H), maxD=float(depth_cut))
        compute(self.metric_prog, self.dm_fbo, self.depth_raw_tex, "f", maxD=float(depth_cut))
        compute(self.metric_prog, self.dmf_fbo, self.depth_filt_tex, "f", maxD=float(depth_cut))
        return dict(depth_filtered=gl.read_tex(self.depth_filt_tex, W, H, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT, np.uint16, 1),
                    depth_metric=gl.read_tex(self.dm_tex, W, H, G.GL_RED, G.GL_FLOAT, np.float32, 1),
                    depth_metric_filtered=gl.read_tex(self.dmf_tex, W, H, G.GL_RED, G.GL_FLOAT, np.float32, 1))

    # IndexMap::predictIndices, EF/IndexMap.cpp:221-279
    def predict_indices(self, vbo, n, pose, time, time_delta=TIME_DELTA):
        gl = self.gl
        gl.begin_pass(self.index_fbo, W, H, "ufff")
        gl.uniforms(self.index_prog, t_inv=np.linalg.inv(pose.astype(np.float32)), cam=self.cam(), maxDepth=MAX_DEPTH, cols=float(W), rows=float(H),
                    time=int(time), timeDelta=int(time_delta))
        gl.attribs(vbo, 3, VSIZE)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.attribs_off(3)
        gl.end_pass()
        return dict(index=gl.read_tex(self.index_tex, W, H, G.GL_RED_INTEGER, G.GL_UNSIGNED_INT, np.uint32, 1),
                    index_vc=gl.read_tex(self.vert_conf_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4),
                    index_ct=gl.read_tex(self.color_time_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4),
                    index_nr=gl.read_tex(self.norm_rad_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4))

    def set_index_images(self, im):
        """the four index-map textures from given images (the oracle's: a later pass is then compared on IDENTICAL inputs)"""
        self._upload(self.index_tex, W, H, G.GL_R32UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_INT, im["index"].astype(np.uint32))
        self._upload(self.vert_conf_tex, W, H, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT, im["index_vc"])
        self._upload(self.color_time_tex, W, H, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT, im["index_ct"])
        self._upload(self.norm_rad_tex, W, H, G.GL_RGBA32F, G.GL_RGBA, G.GL_FLOAT, im["index_nr"])

    # IndexMap::renderSurfelIds (GENERAL), EF/IndexMap.cpp:315-465
    def render_ids(self, vbo, n, pose, time):
        gl = self.gl
        gl.begin_pass(self.ids_fbo, W, H, "i")
        gl.uniforms(self.ids_prog, t_inv=np.linalg.inv(pose.astype(np.float32)), cam=self.cam(), maxDepth=MAX_DEPTH, cols=float(W), rows=float(H), time=int(time),
                    timeDelta=int(TIME_DELTA), conf=float(CONF))
        gl.attribs(vbo, 3, VSIZE)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.attribs_off(3)
        gl.end_pass()
        return gl.read_tex(self.ids_tex, W, H, G.GL_RED_INTEGER, G.GL_INT, np.int32, 1)

    # IndexMap::combinedPredict (ACTIVE), EF/IndexMap.cpp:468-574
    def combined_predict(self, vbo, n, pose, time, max_time):
        gl = self.gl
        gl.glEnable(G.GL_PROGRAM_POINT_SIZE)      # (GL_POINT_SPRITE of :477 is always on in a core context)
        gl.begin_pass(self.combo_fbo, W, H, "fffuf")
        gl.uniforms(self.combo_prog, t_inv=np.linalg.inv(pose.astype(np.float32)), cam=self.cam(), maxDepth=MAX_DEPTH, confThreshold=float(CONF), cols=float(W), rows=float(H),
                    time=int(time), maxTime=int(max_time), timeDelta=int(TIME_DELTA))
        gl.attribs(vbo, 3, VSIZE)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.attribs_off(3)
        gl.end_pass()
        gl.glDisable(G.GL_PROGRAM_POINT_SIZE)
        return dict(pred_image=gl.read_tex(self.image_tex, W, H, G.GL_RGBA, G.GL_UNSIGNED_BYTE, np.uint8, 4),
                    pred_vertex=gl.read_tex(self.vertex_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4),
                    pred_normal=gl.read_tex(self.normal_tex, W, H, G.GL_RGBA, G.GL_FLOAT, np.float32, 4),
                    pred_time=gl.read_tex(self.time_tex, W, H, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT, np.uint16, 1),
                    pred_inst=gl.read_tex(self.inst_tex, W, H, G.GL_RGBA, G.GL_UNSIGNED_BYTE, np.uint8, 4))

    # FillIn::vertex / normal / image, EF/Shaders/FillIn.cpp:64-195 (existing = the ACTIVE prediction's textures of the call above)
    def fill_in(self, rgb, depth_filtered, lost=False):
        gl = self.gl
        self._upload(self.depth_filt_tex, W, H, G.GL_R16UI, G.GL_RED_INTEGER, G.GL_UNSIGNED_SHORT, depth_filtered)
        self._upload(self.rgb_tex, W, H, G.GL_RGBA8, G.GL_RGB, G.GL_UNSIGNED_BYTE, rgb)
        out = {}
        for k, fbo, existing, raw, tex, fmt, typ, dt in (("vertex", self.fill_v_fbo, self.vertex_tex, self.depth_filt_tex, self.fill_v_tex, G.GL_RGBA, G.GL_FLOAT, np.float32),
                                                         ("normal", self.fill_n_fbo, self.normal_tex, self.depth_filt_tex, self.fill_n_tex, G.GL_RGBA, G.GL_FLOAT, np.float32),
                                                         ("image", self.fill_i_fbo, self.image_tex, self.rgb_tex, self.fill_i_tex, G.GL_RGBA, G.GL_UNSIGNED_BYTE, np.uint8)):
            if k not in self.fill_progs:
                continue
            gl.bind_textures([existing, raw])
            gl.begin_pass(fbo, W, H, "f")
            gl.uniforms(self.fill_progs[k], eSampler=0, rSampler=1, passthrough=int(lost), cam=self.cam(inv=True), cols=float(W), rows=float(H))
            gl.glDrawArrays(G.GL_POINTS, 0, 1)
            gl.end_pass()
            out["fill_" + k] = gl.read_tex(tex, W, H, fmt, typ, dt, 4)
        return out

    # GlobalModel::fuse, EF/GlobalModel.cpp:459-698: the data pass (association: update maps + new unstable surfels) and the update pass
    def fuse(self, vbo, n, pose, time, rgb, dm, dmf, weighting, frame_id=0):
        gl = self.gl
        self._upload(self.rgb_tex, W, H, G.GL_RGBA8, G.GL_RGB, G.GL_UNSIGNED_BYTE, rgb)
        self._upload(self.dm_tex, W, H, G.GL_R32F, G.GL_RED, G.GL_FLOAT, dm)
        self._upload(self.dmf_tex, W, H, G.GL_R32F, G.GL_RED, G.GL_FLOAT, dmf)
        gl.begin_pass(self.upd_fbo, TEXDIM, TEXDIM, "fff")
        gl.uniforms(self.data_prog, cSampler=0, drSampler=1, drfSampler=2, indexSampler=3, vertConfSampler=4, colorTimeSampler=5, normRadSampler=6, instgtSampler=7,
                    time=float(time), weighting=float(weighting), cam=self.cam(inv=True), cols=float(W), rows=float(H), scale=1.0, texDim=float(TEXDIM), pose=pose.astype(np.float32),
                    maxDepth=MAX_DEPTH, frameID=int(frame_id), hasInstanceGroundTruth=0)
        gl.glBindBuffer(G.GL_ARRAY_BUFFER, self.uvo)
        gl.glEnableVertexAttribArray(0)
        gl.glVertexAttribPointer(0, 2, G.GL_FLOAT, G.GL_FALSE, 0, None)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, self.new_unstable)
        gl.bind_textures([self.rgb_tex, self.dm_tex, self.dmf_tex, self.index_tex, self.vert_conf_tex, self.color_time_tex, self.norm_rad_tex])
        gl.glBeginQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, self.count_query)
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glDrawArrays(G.GL_POINTS, 0, W * H)
        gl.glEndTransformFeedback()
        gl.glEndQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN)
        gl.attribs_off(1)
        gl.end_pass()
        n_rec = gl.query_result(self.count_query)
        records = gl.read_buffer(self.new_unstable, n_rec * VSIZE).reshape(n_rec, 64)
        upd_ct = gl.read_tex(self.upd_ct, TEXDIM, TEXDIM, G.GL_RGBA, G.GL_FLOAT, np.float32, 4).reshape(-1, 4)
        # update pass: every surfel looks at its texel of the update maps
        out = gl.buffer(nbytes=max(n, 1) * VSIZE)
        gl.uniforms(self.update_prog, vertSamp=0, colorSamp=1, normSamp=2, texDim=float(TEXDIM), time=int(time))
        gl.attribs(vbo, 16, VSIZE)
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, out)
        gl.bind_textures([self.upd_vc, self.upd_ct, self.upd_nr])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.glEndTransformFeedback()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        gl.attribs_off(16)
        gl.end_pass()
        fused = gl.read_buffer(out, n * VSIZE).reshape(n, 64)
        return records, fused, out, n_rec, (upd_ct[:n, 3] == -1)

    # IndexMap::synthesizeDepth, EF/IndexMap.cpp:576-648, as ElasticFusion::processFrame calls it before a deforming clean (EF/ElasticFusion.cpp:667-676): time = tick,
    # maxTime = tick - timeDelta, timeDelta = 65535
    def synthesize_depth(self, vbo, n, pose, time, time_delta):
        gl = self.gl
        gl.glEnable(G.GL_PROGRAM_POINT_SIZE)
        gl.begin_pass(self.syn_depth_fbo, W, H, "f")
        gl.uniforms(self.depth_prog, t_inv=np.linalg.inv(pose.astype(np.float32)), cam=self.cam(), maxDepth=MAX_DEPTH, confThreshold=float(CONF), cols=float(W), rows=float(H),
                    time=int(time), maxTime=int(time - time_delta), timeDelta=65535)
        gl.attribs(vbo, 3, VSIZE)
        gl.glDrawArrays(G.GL_POINTS, 0, n)
        gl.attribs_off(3)
        gl.end_pass()
        gl.glDisable(G.GL_PROGRAM_POINT_SIZE)
        return gl.read_tex(self.syn_depth_tex, W, H, G.GL_RED, G.GL_FLOAT, np.float32, 1)

    # GlobalModel::clean, EF/GlobalModel.cpp:700-925 (no deformation graph): survivors of the map, then the frame's new unstable surfels, through copy_unstable.vert / .geom
    def clean(self, vbo, n, n_rec, pose, time, graph=None, time_delta=TIME_DELTA, is_fern=False):
        gl = self.gl
        nodes = 0
        if graph is not None:      # EF/GlobalModel.cpp:713-720: glTexSubImage2D of the raw graph into the node texture
            g16 = np.ascontiguousarray(graph, np.float32).reshape(-1)
            nodes = g16.size // 16
            row = np.zeros((1, 16384), np.float32)
            row[0, : g16.size] = g16
            self._upload(self.node_tex, 16384, 1, G.GL_R32F, G.GL_RED, G.GL_FLOAT, row)
        out = gl.buffer(nbytes=(n + n_rec + 1) * VSIZE)
        ids = gl.buffer(nbytes=(n + 1) * 4)
        gl.uniforms(self.unstable_prog, time=int(time), confThreshold=float(CONF), scale=1.0, indexSampler=0, vertConfSampler=1, colorTimeSampler=2, normRadSampler=3, nodeSampler=4,
                    depthSampler=5, nodes=float(nodes), nodeCols=16384.0, timeDelta=int(time_delta), maxDepth=MAX_DEPTH, isFern=int(is_fern), t_inv=np.linalg.inv(pose.astype(np.float32)), cam=self.cam(),
                    cols=float(W), rows=float(H), isNew=1)      # (isNew: what the previous frame's clean left it at, EF/GlobalModel.cpp:851)
        gl.attribs(vbo, 16, VSIZE)
        gl.glEnable(G.GL_RASTERIZER_DISCARD)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 0, out)
        gl.glBindBufferBase(G.GL_TRANSFORM_FEEDBACK_BUFFER, 1, ids)
        gl.bind_textures([self.index_tex, self.vert_conf_tex, self.color_time_tex, self.norm_rad_tex, self.node_tex if nodes else self.dummy_f, self.syn_depth_tex if nodes else self.dummy_f])
        gl.glBeginTransformFeedback(G.GL_POINTS)
        gl.glBeginQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, self.count_query)
        gl.glDrawArrays(G.GL_POINTS, 0, n)                                   # survivors -> stream 0
        gl.glFinish()
        gl.uniforms(self.unstable_prog, _bound=True, isNew=0)
        gl.glBeginQueryIndexed(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, 1, self.delete_query)
        gl.glDrawArrays(G.GL_POINTS, 0, n)                                   # their vertex ids -> stream 1
        gl.glFinish()
        gl.glEndQueryIndexed(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, 1)
        n_ids = gl.query_result(self.delete_query)
        gl.uniforms(self.unstable_prog, _bound=True, isNew=1)
        gl.attribs(self.new_unstable, 16, VSIZE)
        gl.glDrawArrays(G.GL_POINTS, 0, n_rec)                               # the frame's new unstable surfels
        gl.glEndQuery(G.GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN)
        count = gl.query_result(self.count_query)
        gl.glEndTransformFeedback()
        gl.glDisable(G.GL_RASTERIZER_DISCARD)
        gl.attribs_off(16)
        gl.end_pass()
        cleaned = gl.read_buffer(out, count * VSIZE).reshape(count, 64)
        kept = gl.read_buffer(ids, n_ids * 4, np.int32)
        return cleaned, kept, out, count


def rel(a, b):
    return float(np.abs(a - b).max()) if a.size else 0.0


def report(name, **kw):
    print(f"{name:34s} " + "  ".join(f"{k} {v:.4g}" if isinstance(v, float) else f"{k} {v}" for k, v in kw.items()))


def survivors_of(before, after_pc):
    """indices of the rows of `before` that appear, in order, at the head of `after` (the oracle compacts in order: EF/Shaders/copy_unstable.geom emits survivors in vertex order)"""
    keep, j = [], 0
    for i in range(before.shape[0]):
        if j < after_pc.shape[0] and np.array_equal(before[i, :3], after_pc[j, :3]):
            keep.append(i); j += 1
    return np.array(keep, np.int64), j


def survivors_of_deformed(before, after):
    """survivors of a clean pass that also MOVED them (deformation): rows are matched by their untouched columns -- confidence, radius, colour, initTime -- in order"""
    key_b = np.stack([before["pc"][:, 3], before["nr"][:, 3], before["col"][:, 0], before["tm"][:, 0], before["ic"][:, 0], before["ic"][:, 1]], 1)
    key_a = np.stack([after["pc"][:, 3], after["nr"][:, 3], after["col"][:, 0], after["tm"][:, 0], after["ic"][:, 0], after["ic"][:, 1]], 1)
    keep, j = [], 0
    for i in range(key_b.shape[0]):
        if j < key_a.shape[0] and np.array_equal(key_b[i], key_a[j]):
            keep.append(i); j += 1
    return np.array(keep, np.int64), j


def main():
    import oracle_lib as ol
    from instancefusion_amd import synth

    ol.build()
    NF = 16
    seed, motion = int(os.environ.get("IFX_GL_SEED", "0")), os.environ.get("IFX_GL_MOTION", "")      # (other scenes / camera motions: one-off runs, nothing written)
    if seed or motion:
        scene = synth.Scene(seed or 1)
        st = synth.make_stream_from_poses(synth.trajectory_profile(motion or "nominal", NF + 1, seed or 1), scene, W, H, noise_seed=(seed or 1) + 1, **K)
        print(f"scene seed {seed or 1}, motion {motion or 'nominal'}")
    else:
        st = synth.make_stream(NF + 1, W, H, noise=True, **K)
    o = ol.Oracle(w=W, h=H, max_surfels=max(200000, W * H * 8), confidence=CONF, **K)
    o2 = ol.Oracle(w=W, h=H, max_surfels=max(200000, W * H * 8), confidence=CONF, **K)      # the same frames plus the next one: its pose for that frame
    for i in range(NF):
        o.process_frame(st["rgb"][i], st["depth"][i]); o2.process_frame(st["rgb"][i], st["depth"][i])
    rgb, depth = st["rgb"][NF], st["depth"][NF]
    pose = o2.process_frame(rgb, depth).astype(np.float32)
    o2.close()
    m = o.download()
    t = int(o.tick)
    n = m["pc"].shape[0]
    print(f"map: {n} surfels, {(m['pc'][:, 3] > CONF).sum()} stable, frame time {t}")
    ref = RefGL()
    gl = ref.gl
    gold = dict(width=np.int32(W), height=np.int32(H), K=np.array([K["fx"], K["fy"], K["cx"], K["cy"]], np.float32), confidence=np.float32(CONF), time=np.int32(t),
                pose=pose, rgb=rgb, depth=depth, map_pc=m["pc"], map_nr=m["nr"], map_col=m["col"], map_tm=m["tm"], map_ic=m["ic"], map_votes=m["votes"])      # (votes: the reference's -1 words of first-frame surfels, 0 elsewhere: they compress to nothing)
    # ---- a2: bilateral filter + metric depth (depth_bilateral.frag, depth_metric.frag)
    o.set_frame(rgb, depth)
    gp = ref.preprocess(depth, 12.0)
    op = {k: o.image(k) for k in ("depth_filtered", "depth_metric", "depth_metric_filtered")}
    df = np.abs(gp["depth_filtered"].astype(np.int32) - op["depth_filtered"].astype(np.int32))
    report("a2 bilateral (u16 mm)", equal_pct=float((df == 0).mean() * 100), max_diff_mm=int(df.max()), valid=int((op["depth_filtered"] > 0).sum()))
    report("a2 metric raw (f32 m)", equal_pct=float((gp["depth_metric"] == op["depth_metric"]).mean() * 100), max_diff=rel(gp["depth_metric"], op["depth_metric"]))
    for k, v in gp.items():
        gold["gl_" + k] = v
    # ---- a10: index map of the pre-fuse map (index_map.vert / .frag)
    vbo = gl.buffer(pack_vbo(m))
    gi = ref.predict_indices(vbo, n, pose, t)
    o.predict_indices(pose, t)
    oi = {k: o.image(k) for k in ("index", "index_vc", "index_ct", "index_nr")}

    def index_report(tag, gi, oi):
        same = gi["index"] == oi["index"].astype(np.uint32)
        both = same & (gi["index"] > 0)
        report(tag, ids_equal_pct=float(same.mean() * 100), differ=int((~same).sum()), gl_drawn=int((gi["index"] > 0).sum()), oracle_drawn=int((oi["index"] > 0).sum()),
               vc_max=rel(gi["index_vc"][both], oi["index_vc"][both]), nr_max=rel(gi["index_nr"][both], oi["index_nr"][both]), ct_max=rel(gi["index_ct"][both], oi["index_ct"][both]))

    index_report("a10 index map (pre-fuse)", gi, oi)
    for k, v in gi.items():
        gold["gl_pre_" + k] = v
    # ---- a11 + a12: association and fusion (data.vert / .geom / .frag, update.vert) on the ORACLE's index map and metric depths
    ref.set_index_images(oi)
    records, fused, fused_buf, n_rec, upd_flag = ref.fuse(vbo, n, pose, t, rgb, op["depth_metric"], op["depth_metric_filtered"], 1.0)
    o.fuse(pose, t, 1.0)
    mf = o.download()
    g_upd = fused[:, 7] == np.float32(t)
    o_upd = (mf["tm"][:, 1] == np.float32(t)) & (m["tm"][:, 1] != np.float32(t))
    g_upd &= (m["tm"][:, 1] != np.float32(t))
    fu = unpack_vbo(fused)
    both = g_upd & o_upd
    report("a11 association (matched surfels)", gl=int(g_upd.sum()), oracle=int(o_upd.sum()), both=int(both.sum()), only_gl=int((g_upd & ~o_upd).sum()), only_oracle=int((o_upd & ~g_upd).sum()))
    report("a12 fusion update (both matched)", pc_max=rel(fu["pc"][both], mf["pc"][both]), nr_max=rel(fu["nr"][both], mf["nr"][both]),
           colour_equal_pct=float((fu["col"][both, 0] == mf["col"][both, 0]).mean() * 100), untouched_equal=bool(np.array_equal(fu["pc"][~g_upd & ~o_upd], mf["pc"][~g_upd & ~o_upd])))
    g_new = records[records[:, 7] == -2.0]
    report("a11 new unstable surfels", gl=int(g_new.shape[0]), matched_records=int((records[:, 7] == -1.0).sum()))
    gold.update(gl_fuse_updated=g_upd, gl_fused_pc=fu["pc"], gl_fused_nr=fu["nr"], gl_fused_col=fu["col"], gl_fused_tm=fu["tm"], gl_new_records=g_new[:, :16].copy())
    # ---- a10 again on the post-fuse map, then a13: clean (copy_unstable.vert / .geom) on the ORACLE's post-fuse map and its index map
    vbo_f = gl.buffer(pack_vbo(mf))
    gi2 = ref.predict_indices(vbo_f, n, pose, t)
    o.predict_indices(pose, t)
    oi2 = {k: o.image(k) for k in ("index", "index_vc", "index_ct", "index_nr")}
    index_report("a10 index map (post-fuse)", gi2, oi2)
    gold["gl_post_index"] = gi2["index"]
    ref.set_index_images(oi2)
    cleaned, kept, cleaned_buf, count = ref.clean(vbo_f, n, n_rec, pose, t)
    o.clean(pose, t)
    mc = o.download()
    o_keep, o_nk = survivors_of(mf["pc"], mc["pc"])
    report("a13 clean: survivors of the map", gl=int(kept.shape[0]), oracle=int(o_nk), same_set=bool(np.array_equal(np.sort(kept), o_keep)),
           only_gl=int(np.setdiff1d(kept, o_keep).size), only_oracle=int(np.setdiff1d(o_keep, kept).size))
    cu = unpack_vbo(cleaned)
    nk = int(kept.shape[0])
    report("a13 clean: appended new surfels", gl=int(count - nk), oracle=int(mc["pc"].shape[0] - o_nk),
           pc_max=rel(cu["pc"][nk:], mc["pc"][o_nk:]) if count - nk == mc["pc"].shape[0] - o_nk else float("nan"),
           nr_max=rel(cu["nr"][nk:], mc["nr"][o_nk:]) if count - nk == mc["pc"].shape[0] - o_nk else float("nan"))
    gold.update(gl_clean_kept=kept.astype(np.int32), gl_clean_new_pc=cu["pc"][nk:], gl_clean_new_nr=cu["nr"][nk:], gl_clean_new_col=cu["col"][nk:], gl_clean_new_tm=cu["tm"][nk:],
                gl_clean_new_ic=cu["ic"][nk:])
    # ---- a14: surfel ids, a9: splat prediction + fill-in, on the ORACLE's post-clean map
    nc = mc["pc"].shape[0]
    vbo_c = gl.buffer(pack_vbo(mc))
    g_ids = ref.render_ids(vbo_c, nc, pose, t)
    o_ids = o.render_ids(pose, 0)
    same = g_ids == o_ids
    report("a14 surfel ids", equal_pct=float(same.mean() * 100), differ=int((~same).sum()), gl_drawn=int((g_ids > 0).sum()), oracle_drawn=int((o_ids > 0).sum()),
           coverage_differs=int(((g_ids > 0) != (o_ids > 0)).sum()))
    gold["gl_ids"] = g_ids
    gpred = ref.combined_predict(vbo_c, nc, pose, t, t)
    o.combined_predict(pose, t, t)
    opred = {k: o.image(k) for k in ("pred_vertex", "pred_normal", "pred_image", "pred_time", "pred_inst", "fill_vertex", "fill_normal", "fill_image")}
    g_cov, o_cov = gpred["pred_vertex"][..., 2] != 0, opred["pred_vertex"][..., 2] != 0
    both = g_cov & o_cov
    dz = np.abs(gpred["pred_vertex"][..., 2] - opred["pred_vertex"][..., 2])
    report("a9 splat prediction", gl_drawn=int(g_cov.sum()), oracle_drawn=int(o_cov.sum()), coverage_differs=int((g_cov != o_cov).sum()), z_within_1e_5_pct=float((dz[both] < 1e-5).mean() * 100),
           z_within_1mm_pct=float((dz[both] < 1e-3).mean() * 100), time_equal_pct=float((gpred["pred_time"][both] == opred["pred_time"][both]).mean() * 100),
           image_equal_pct=float((gpred["pred_image"][both][:, :3] == opred["pred_image"][both][:, :3]).all(axis=1).mean() * 100))
    for k, v in gpred.items():
        gold["gl_" + k] = v
    gfill = ref.fill_in(rgb, op["depth_filtered"])
    for k, v in gfill.items():
        ov = opred[k]
        if v.dtype == np.uint8:
            report("a9 " + k, equal_pct=float((v[..., :3] == ov[..., :3]).all(axis=-1).mean() * 100))
        else:
            fin = np.isfinite(v).all(axis=-1) & np.isfinite(ov).all(axis=-1)
            d = np.abs(v[..., :3] - ov[..., :3]).max(axis=-1)
            report("a9 " + k, within_1e_5_pct=float((d[fin] < 1e-5).mean() * 100), within_1e_3_pct=float((d[fin] < 1e-3).mean() * 100), nonfinite_differs=int((np.isfinite(v).all(axis=-1) != np.isfinite(ov).all(axis=-1)).sum()))
        gold["gl_" + k] = v
    # ---- f-3: the deformation graph applied by the clean pass (copy_unstable.vert:176-330) + IndexMap::synthesizeDepth (splat.vert, depth_splat.frag).  A map with INACTIVE
    # surfels (time window of 6 frames over a moving camera), a synthetic graph: every 300th surfel a node, ordered by time, a small rotation + translation each.
    TD = 6
    od = ol.Oracle(w=W, h=H, max_surfels=max(200000, W * H * 8), confidence=CONF, time_delta=TD, **K)
    pd = None
    for i in range(NF):
        pd = od.process_frame(st["rgb"][i], st["depth"][i]).astype(np.float32)
    md = od.download()
    td = int(od.tick)
    nd = md["pc"].shape[0]
    rng = np.random.RandomState(5)
    pick = np.arange(0, nd, 300)
    pick = pick[np.argsort(md["tm"][pick, 0], kind="stable")]
    graph = np.zeros((pick.size, 16), np.float32)
    for k_, s_ in enumerate(pick):
        ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.03, 0.03)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        graph[k_, 0:3] = md["pc"][s_, :3]
        graph[k_, 3:12] = R.T.reshape(9)                     # column-major, as Eigen stores the node's rotation (EF/Deformation.cpp:195-197)
        graph[k_, 12:15] = rng.uniform(-0.01, 0.01, 3)
        graph[k_, 15] = md["tm"][s_, 0]
    vbo_d = gl.buffer(pack_vbo(md))
    od.predict_indices(pd, td)
    oid = {k: od.image(k) for k in ("index", "index_vc", "index_ct", "index_nr")}
    ref.set_index_images(oid)
    g_depth = ref.synthesize_depth(vbo_d, nd, pd, td, TD)
    cleaned_d, kept_d, _, count_d = ref.clean(vbo_d, nd, 0, pd, td, graph=graph, time_delta=TD)
    od.set_loop_closure(True)             # (allocates the buffers of the INACTIVE prediction, into which the oracle synthesises the depth image of a deforming clean)
    od.set_deformation(graph, False)
    od.clean(pd, td)
    mdc = od.download()
    cd = unpack_vbo(cleaned_d)
    o_keep_d, o_nk_d = survivors_of_deformed(md, mdc)
    same_set = np.array_equal(np.sort(kept_d), o_keep_d)
    dpos = np.abs(cd["pc"][:, :3] - mdc["pc"][:, :3]).max(axis=1) if same_set else np.array([np.nan])
    dnrm = np.abs(cd["nr"][:, :3] - mdc["nr"][:, :3]).max(axis=1) if same_set else np.array([np.nan])
    moved = np.abs(mdc["pc"][:, :3] - md["pc"][o_keep_d, :3]).max(axis=1) if same_set else np.array([np.nan])
    report("f-3 deformation in the clean pass", nodes=int(pick.size), gl_survivors=int(kept_d.shape[0]), oracle_survivors=int(o_nk_d), same_set=bool(same_set), inactive_depth_px=int((g_depth > 0).sum()),
           moved_max_m=float(moved.max()), pos_within_1e_5_pct=float((dpos < 1e-5).mean() * 100), pos_max=float(dpos.max()), normal_within_1e_4_pct=float((dnrm < 1e-4).mean() * 100),
           last_time_equal_pct=float((cd["tm"][:, 1] == mdc["tm"][:, 1]).mean() * 100) if same_set else float("nan"),
           reactivated_gl=int(((cd["tm"][:, 1] == td) & (md["tm"][kept_d, 1] != td)).sum()) if same_set else -1)
    gold.update(d_time_delta=np.int32(TD), d_time=np.int32(td), d_pose=pd, d_map_pc=md["pc"], d_map_nr=md["nr"], d_map_col=md["col"], d_map_tm=md["tm"], d_graph=graph,
                gl_d_kept=kept_d.astype(np.int32), gl_d_pc=cd["pc"], gl_d_nr=cd["nr"], gl_d_tm=cd["tm"], gl_d_depth=g_depth)
    od.close()
    gl.close()
    if (W, H) != (160, 120) or seed or motion:
        print("(not the golden configuration: nothing written)")
        return
    out = os.path.join(OUT, "gl_map_passes.npz")
    np.savez_compressed(out, **gold)
    print(f"wrote {out}: {os.path.getsize(out) / 1e6:.2f} MB, {len(gold)} arrays")


if __name__ == "__main__":
    main()
