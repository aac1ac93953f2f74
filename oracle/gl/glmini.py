"""TEST INFRASTRUCTURE -- a minimal ctypes binding of the OpenGL 3.3 / 4.x core entry points the golden generator needs (tools/make_golden_gl.py), on the window-less
context of oracle/gl/dri_ctx.c (Mesa llvmpipe through the DRI swrast loader interface).  PyOpenGL is not in the image; the ~70 functions below are bound by name from
libglapi.  Nothing of the product imports this.

The helpers mirror what the reference's host code does around its shaders -- pangolin::GlSlProgram::AddShaderFromFile with its `#include` expansion
(EF/Shaders/Shaders.h:71-116), GPUTexture (EF/GPUTexture.cpp:21-66: NEAREST filtering except the linear RGB texture, CLAMP_TO_EDGE... see tex2d), pangolin::GlFramebuffer
with a GlRenderBuffer depth attachment (GL_DEPTH_COMPONENT24) -- with core-profile formats in place of the legacy ones the reference names (GL_R32UI for
GL_LUMINANCE32UI_EXT, GL_R32F for GL_LUMINANCE32F_ARB ...): same bits per texel, same shader-visible values."""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "_ref", "libdri_ctx.so")

# ---- enums (GL/glcorearb.h)
GL_FALSE, GL_TRUE = 0, 1
GL_POINTS = 0x0000
GL_DEPTH_BUFFER_BIT, GL_COLOR_BUFFER_BIT = 0x0100, 0x4000
GL_LESS = 0x0201
GL_DEPTH_TEST = 0x0B71
GL_TEXTURE_2D = 0x0DE1
GL_UNSIGNED_BYTE, GL_UNSIGNED_SHORT, GL_INT, GL_UNSIGNED_INT, GL_FLOAT = 0x1401, 0x1403, 0x1404, 0x1405, 0x1406
GL_RED, GL_RGB, GL_RGBA = 0x1903, 0x1907, 0x1908
GL_RED_INTEGER = 0x8D94
GL_NEAREST, GL_LINEAR = 0x2600, 0x2601
GL_TEXTURE_MAG_FILTER, GL_TEXTURE_MIN_FILTER, GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T = 0x2800, 0x2801, 0x2802, 0x2803
GL_CLAMP_TO_EDGE, GL_REPEAT = 0x812F, 0x2901
GL_RGBA8, GL_RGBA32F, GL_R32F, GL_R32UI, GL_R32I, GL_R16UI, GL_R8UI = 0x8058, 0x8814, 0x822E, 0x8236, 0x8235, 0x8234, 0x8232
GL_DEPTH_COMPONENT24 = 0x81A6
GL_TEXTURE0 = 0x84C0
GL_ARRAY_BUFFER = 0x8892
GL_STREAM_DRAW, GL_STATIC_DRAW = 0x88E0, 0x88E4
GL_FRAGMENT_SHADER, GL_VERTEX_SHADER, GL_GEOMETRY_SHADER = 0x8B30, 0x8B31, 0x8DD9
GL_COMPILE_STATUS, GL_LINK_STATUS, GL_INFO_LOG_LENGTH = 0x8B81, 0x8B82, 0x8B84
GL_FRAMEBUFFER, GL_RENDERBUFFER = 0x8D40, 0x8D41
GL_COLOR_ATTACHMENT0, GL_DEPTH_ATTACHMENT = 0x8CE0, 0x8D00
GL_FRAMEBUFFER_COMPLETE = 0x8CD5
GL_RASTERIZER_DISCARD = 0x8C89
GL_TRANSFORM_FEEDBACK_BUFFER = 0x8C8E
GL_INTERLEAVED_ATTRIBS = 0x8C8C
GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN = 0x8C88
GL_QUERY_RESULT = 0x8866
GL_PROGRAM_POINT_SIZE = 0x8642
GL_COLOR = 0x1800
GL_PACK_ALIGNMENT, GL_UNPACK_ALIGNMENT = 0x0D05, 0x0CF5

_v, _i, _u, _f, _p, _sz = None, C.c_int, C.c_uint, C.c_float, C.c_void_p, C.c_ssize_t
_SIGS = {
    "glGetError": (_u, []), "glGetString": (C.c_char_p, [_u]), "glFinish": (_v, []), "glEnable": (_v, [_u]), "glDisable": (_v, [_u]), "glDepthFunc": (_v, [_u]),
    "glViewport": (_v, [_i, _i, _i, _i]), "glClearColor": (_v, [_f, _f, _f, _f]), "glClear": (_v, [_u]), "glPointSize": (_v, [_f]), "glPixelStorei": (_v, [_u, _i]),
    "glCreateShader": (_u, [_u]), "glShaderSource": (_v, [_u, _i, C.POINTER(C.c_char_p), C.POINTER(_i)]), "glCompileShader": (_v, [_u]),
    "glGetShaderiv": (_v, [_u, _u, C.POINTER(_i)]), "glGetShaderInfoLog": (_v, [_u, _i, C.POINTER(_i), C.c_char_p]),
    "glCreateProgram": (_u, []), "glAttachShader": (_v, [_u, _u]), "glLinkProgram": (_v, [_u]), "glUseProgram": (_v, [_u]),
    "glGetProgramiv": (_v, [_u, _u, C.POINTER(_i)]), "glGetProgramInfoLog": (_v, [_u, _i, C.POINTER(_i), C.c_char_p]),
    "glTransformFeedbackVaryings": (_v, [_u, _i, C.POINTER(C.c_char_p), _u]),
    "glGetUniformLocation": (_i, [_u, C.c_char_p]), "glUniform1i": (_v, [_i, _i]), "glUniform1f": (_v, [_i, _f]), "glUniform2f": (_v, [_i, _f, _f]),
    "glUniform3f": (_v, [_i, _f, _f, _f]), "glUniform4f": (_v, [_i, _f, _f, _f, _f]), "glUniformMatrix4fv": (_v, [_i, _i, C.c_ubyte, _p]),
    "glGenBuffers": (_v, [_i, C.POINTER(_u)]), "glBindBuffer": (_v, [_u, _u]), "glBufferData": (_v, [_u, _sz, _p, _u]), "glGetBufferSubData": (_v, [_u, _sz, _sz, _p]),
    "glDeleteBuffers": (_v, [_i, C.POINTER(_u)]),
    "glGenVertexArrays": (_v, [_i, C.POINTER(_u)]), "glBindVertexArray": (_v, [_u]), "glEnableVertexAttribArray": (_v, [_u]), "glDisableVertexAttribArray": (_v, [_u]),
    "glVertexAttribPointer": (_v, [_u, _i, _u, C.c_ubyte, _i, _p]),
    "glGenTextures": (_v, [_i, C.POINTER(_u)]), "glBindTexture": (_v, [_u, _u]), "glTexImage2D": (_v, [_u, _i, _i, _i, _i, _i, _u, _u, _p]),
    "glTexParameteri": (_v, [_u, _u, _i]), "glActiveTexture": (_v, [_u]), "glGetTexImage": (_v, [_u, _i, _u, _u, _p]), "glDeleteTextures": (_v, [_i, C.POINTER(_u)]),
    "glGenFramebuffers": (_v, [_i, C.POINTER(_u)]), "glBindFramebuffer": (_v, [_u, _u]), "glFramebufferTexture2D": (_v, [_u, _u, _u, _u, _i]),
    "glGenRenderbuffers": (_v, [_i, C.POINTER(_u)]), "glBindRenderbuffer": (_v, [_u, _u]), "glRenderbufferStorage": (_v, [_u, _u, _i, _i]),
    "glFramebufferRenderbuffer": (_v, [_u, _u, _u, _u]), "glDrawBuffers": (_v, [_i, C.POINTER(_u)]), "glCheckFramebufferStatus": (_u, [_u]),
    "glDrawArrays": (_v, [_u, _i, _i]), "glBindBufferBase": (_v, [_u, _u, _u]), "glBeginTransformFeedback": (_v, [_u]), "glEndTransformFeedback": (_v, []),
    "glGenQueries": (_v, [_i, C.POINTER(_u)]), "glBeginQuery": (_v, [_u, _u]), "glEndQuery": (_v, [_u]), "glBeginQueryIndexed": (_v, [_u, _u, _u]),
    "glEndQueryIndexed": (_v, [_u, _u]), "glGetQueryObjectuiv": (_v, [_u, _u, C.POINTER(_u)]),
    "glClearBufferiv": (_v, [_u, _i, C.POINTER(_i)]), "glClearBufferuiv": (_v, [_u, _i, C.POINTER(_u)]), "glClearBufferfv": (_v, [_u, _i, C.POINTER(_f)]),
}


class GL:
    """The context (current on the creating thread) and the bound entry points: gl.glDrawArrays(...)."""

    def __init__(self, width=640, height=480):
        self.lib = C.CDLL(LIB)
        self.lib.dri_ctx_create.argtypes = [_i, _i, C.c_char_p, _i]
        self.lib.dri_ctx_proc.restype = _p
        self.lib.dri_ctx_proc.argtypes = [C.c_char_p]
        err = C.create_string_buffer(256)
        if self.lib.dri_ctx_create(width, height, err, 256) != 0:
            raise RuntimeError("no GL context: " + err.value.decode())
        for name, (res, args) in _SIGS.items():
            addr = self.lib.dri_ctx_proc(name.encode())
            if not addr:
                raise RuntimeError("the driver has no " + name)
            setattr(self, name, C.CFUNCTYPE(res, *args)(addr))
        vao = _u(0)
        self.glGenVertexArrays(1, C.byref(vao))
        self.glBindVertexArray(vao)          # core profile: attribute state lives in a vertex array object (the reference runs a compatibility context without one)
        self.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        self.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)

    def close(self):
        self.lib.dri_ctx_destroy()

    def check(self, what=""):
        e = self.glGetError()
        if e:
            raise RuntimeError(f"GL error 0x{e:04x} {what}")

    def version(self):
        return self.glGetString(0x1F02).decode(), self.glGetString(0x1F01).decode()

    # ---- programs: the reference's shader FILES, as they lie (pangolin expands `#include "x"` textually from the shader directory)
    @staticmethod
    def load_source(shader_dir, name):
        def expand(path, depth=0):
            out = []
            for line in open(path).read().split("\n"):
                m = re.match(r'\s*#include\s+"([^"]+)"', line)
                if m and depth < 8:
                    out.append(expand(os.path.join(shader_dir, m.group(1)), depth + 1))
                else:
                    out.append(line)
            return "\n".join(out)

        return expand(os.path.join(shader_dir, name))

    def program(self, shader_dir, vert, frag=None, geom=None, feedback=None):
        prog = self.glCreateProgram()
        for kind, name in ((GL_VERTEX_SHADER, vert), (GL_GEOMETRY_SHADER, geom), (GL_FRAGMENT_SHADER, frag)):
            if not name:
                continue
            sh = self.glCreateShader(kind)
            src = C.c_char_p(self.load_source(shader_dir, name).encode())
            self.glShaderSource(sh, 1, C.byref(src), None)
            self.glCompileShader(sh)
            ok = _i(0)
            self.glGetShaderiv(sh, GL_COMPILE_STATUS, C.byref(ok))
            if not ok.value:
                log = C.create_string_buffer(8192)
                self.glGetShaderInfoLog(sh, 8192, None, log)
                raise RuntimeError(f"{name}: {log.value.decode()}")
            self.glAttachShader(prog, sh)
        if feedback:
            arr = (C.c_char_p * len(feedback))(*[s.encode() for s in feedback])
            self.glTransformFeedbackVaryings(prog, len(feedback), arr, GL_INTERLEAVED_ATTRIBS)
        self.glLinkProgram(prog)
        ok = _i(0)
        self.glGetProgramiv(prog, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            self.glGetProgramInfoLog(prog, 8192, None, log)
            raise RuntimeError(f"link {vert}: {log.value.decode()}")
        return prog

    def uniforms(self, prog, _bound=False, **kw):
        """Shader::setUniform (EF/Shaders/Shaders.h:38-67): int / float / vec2-4 / mat4 (column-major, as Eigen stores it: transpose = false).  _bound: the program is
        current already (glUseProgram is an error while transform feedback is active: the reference sets `isNew` between the draws of one feedback session)"""
        if not _bound:
            self.glUseProgram(prog)
        for name, v in kw.items():
            loc = self.glGetUniformLocation(prog, name.encode())
            if loc < 0:
                continue      # (a uniform the compiler removed: glUniform on -1 is a no-op in GL as well)
            if isinstance(v, (bool, int, np.integer)):
                self.glUniform1i(loc, int(v))
            elif isinstance(v, (float, np.floating)):
                self.glUniform1f(loc, float(v))
            else:
                a = np.asarray(v, np.float32)
                if a.shape == (4, 4):
                    m = np.ascontiguousarray(a.T)     # row-major numpy -> column-major GL
                    self.glUniformMatrix4fv(loc, 1, GL_FALSE, m.ctypes.data)
                elif a.size == 2:
                    self.glUniform2f(loc, *map(float, a))
                elif a.size == 3:
                    self.glUniform3f(loc, *map(float, a))
                elif a.size == 4:
                    self.glUniform4f(loc, *map(float, a.reshape(4)))
                else:
                    raise ValueError(name)

    # ---- buffers
    def buffer(self, data=None, nbytes=0, usage=GL_STREAM_DRAW):
        b = _u(0)
        self.glGenBuffers(1, C.byref(b))
        self.glBindBuffer(GL_ARRAY_BUFFER, b)
        if data is not None:
            data = np.ascontiguousarray(data)
            self.glBufferData(GL_ARRAY_BUFFER, data.nbytes, data.ctypes.data, usage)
        else:
            self.glBufferData(GL_ARRAY_BUFFER, nbytes, None, usage)
        self.glBindBuffer(GL_ARRAY_BUFFER, 0)
        return b.value

    def read_buffer(self, buf, nbytes, dtype=np.float32):
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype)
        self.glBindBuffer(GL_ARRAY_BUFFER, buf)
        self.glGetBufferSubData(GL_ARRAY_BUFFER, 0, nbytes, out.ctypes.data)
        self.glBindBuffer(GL_ARRAY_BUFFER, 0)
        return out

    def attribs(self, buf, n_vec4, stride, first=0):
        """n_vec4 consecutive vec4 attributes (locations first ...) at their offsets in an interleaved record of `stride` bytes"""
        self.glBindBuffer(GL_ARRAY_BUFFER, buf)
        for k in range(n_vec4):
            self.glEnableVertexAttribArray(first + k)
            self.glVertexAttribPointer(first + k, 4, GL_FLOAT, GL_FALSE, stride, C.c_void_p(16 * (first + k)))

    def attribs_off(self, n):
        for k in range(n):
            self.glDisableVertexAttribArray(k)
        self.glBindBuffer(GL_ARRAY_BUFFER, 0)

    # ---- textures (GPUTexture, EF/GPUTexture.cpp:32-39 -> pangolin::GlTexture, an un-vendored dependency: NEAREST min / mag filters, LINEAR where `draw` is set -- the RGB
    # texture --, and CLAMP_TO_EDGE in both directions as pangolin sets them)
    def tex2d(self, w, h, internal, fmt, typ, data=None, linear=False):
        t = _u(0)
        self.glGenTextures(1, C.byref(t))
        self.glBindTexture(GL_TEXTURE_2D, t)
        ptr = None
        if data is not None:
            data = np.ascontiguousarray(data)
            ptr = data.ctypes.data
        self.glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, typ, ptr)
        f = GL_LINEAR if linear else GL_NEAREST
        self.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, f)
        self.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, f)
        self.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE)
        self.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE)
        self.glBindTexture(GL_TEXTURE_2D, 0)
        self.check("tex2d")
        return t.value

    def read_tex(self, tex, w, h, fmt, typ, dtype, channels):
        out = np.zeros((h, w, channels) if channels > 1 else (h, w), dtype)
        self.glBindTexture(GL_TEXTURE_2D, tex)
        self.glGetTexImage(GL_TEXTURE_2D, 0, fmt, typ, out.ctypes.data)
        self.glBindTexture(GL_TEXTURE_2D, 0)
        self.check("read_tex")
        return out

    def bind_textures(self, texs):
        for k, t in enumerate(texs):
            self.glActiveTexture(GL_TEXTURE0 + k)
            self.glBindTexture(GL_TEXTURE_2D, t)
        self.glActiveTexture(GL_TEXTURE0)

    # ---- framebuffers (pangolin::GlFramebuffer + GlRenderBuffer: colour attachments in AttachColour order, a 24-bit depth renderbuffer)
    def framebuffer(self, w, h, colour_texs):
        f = _u(0)
        self.glGenFramebuffers(1, C.byref(f))
        self.glBindFramebuffer(GL_FRAMEBUFFER, f)
        for k, t in enumerate(colour_texs):
            self.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + k, GL_TEXTURE_2D, t, 0)
        r = _u(0)
        self.glGenRenderbuffers(1, C.byref(r))
        self.glBindRenderbuffer(GL_RENDERBUFFER, r)
        self.glRenderbufferStorage(GL_RENDERBUFFER, GL_DEPTH_COMPONENT24, w, h)
        self.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, r)
        bufs = (_u * len(colour_texs))(*[GL_COLOR_ATTACHMENT0 + k for k in range(len(colour_texs))])
        self.glDrawBuffers(len(colour_texs), bufs)
        st = self.glCheckFramebufferStatus(GL_FRAMEBUFFER)
        if st != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError(f"framebuffer incomplete 0x{st:04x}")
        self.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        return f.value

    def begin_pass(self, fbo, w, h, kinds):
        """Bind, viewport, glClearColor(0,0,0,0) + glClear(COLOR | DEPTH) as every pass of the reference does; `kinds`: per attachment 'f' / 'i' / 'u' (an integer attachment
        is cleared with glClearBuffer*: glClear's float colour is undefined for it in the core profile)"""
        self.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        self.glViewport(0, 0, w, h)
        self.glClearColor(0.0, 0.0, 0.0, 0.0)
        self.glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)
        for k, kind in enumerate(kinds):
            if kind == "i":
                z = (_i * 4)(0, 0, 0, 0)
                self.glClearBufferiv(GL_COLOR, k, z)
            elif kind == "u":
                z = (_u * 4)(0, 0, 0, 0)
                self.glClearBufferuiv(GL_COLOR, k, z)

    def end_pass(self):
        self.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        self.glUseProgram(0)
        self.glFinish()
        self.check("end_pass")

    def query(self):
        q = _u(0)
        self.glGenQueries(1, C.byref(q))
        return q.value

    def query_result(self, q):
        r = _u(0)
        self.glGetQueryObjectuiv(q, GL_QUERY_RESULT, C.byref(r))
        return int(r.value)
