/* TEST INFRASTRUCTURE -- oracle/gl: a GL 3.3 core context on Mesa's software rasteriser WITHOUT a window system, through the DRI "swrast" loader interface
 * (/usr/include/GL/internal/dri_interface.h, mesa-common-dev 23.2.1): swrast_dri.so is dlopen()ed, __driDriverGetExtensions_swrast gives the core and swrast
 * extensions, the screen is created with a loader extension whose drawable is a block of host memory (no X server, EGL or OSMesa in the image), and the GL entry
 * points come from libglapi's _glapi_get_proc_address.  Used by oracle/gl/ref_shaders.c to run the reference's UNMODIFIED GLSL (read from /root/reference at run
 * time, in this container only) and write its outputs as golden fixtures.  Nothing of the product links or loads this. */
#ifndef ORACLE_GL_DRI_CTX_H
#define ORACLE_GL_DRI_CTX_H
#ifdef __cplusplus
extern "C" {
#endif
/* returns 0 on success; the context is current on the calling thread afterwards */
int dri_ctx_create(int width, int height, char* err, int err_len);
void dri_ctx_destroy(void);
/* GL entry point by name (NULL if the driver has none) */
void* dri_ctx_proc(const char* name);
#ifdef __cplusplus
}
#endif
#endif
