/* see dri_ctx.h */
#include "dri_ctx.h"

#include <GL/gl.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void* g_drv;
static void* g_glapi;
static const __DRIcoreExtension* g_core;
static const __DRIswrastExtension* g_swrast;
static __DRIscreen* g_screen;
static __DRIcontext* g_ctx;
static __DRIdrawable* g_draw;
static const __DRIconfig** g_configs;
static int g_w, g_h;
static char* g_fb;

/* the loader side: the "window" is host memory */
static void ld_get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* priv) { (void)d; (void)priv; *x = 0; *y = 0; *w = g_w; *h = g_h; }
static void ld_put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* priv) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv; }
static void ld_get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* priv) { (void)d; (void)x; (void)y; (void)priv; memset(data, 0, (size_t)w * h * 4); }
static void ld_put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* priv) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)priv; }
static void ld_get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* priv) { (void)d; (void)x; (void)y; (void)w; (void)priv; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
    .putImage2 = ld_put_image2,
    .getImage2 = ld_get_image2,
};
static const __DRIextension* g_loader_exts[] = {&g_loader.base, NULL};

static int fail(char* err, int n, const char* what)
{
    if (err && n > 0) snprintf(err, (size_t)n, "%s", what);
    return -1;
}

int dri_ctx_create(int width, int height, char* err, int err_len)
{
    g_w = width; g_h = height;
    const char* paths[] = {"/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", "swrast_dri.so", NULL};
    for (int i = 0; paths[i] && !g_drv; i++) g_drv = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_drv) return fail(err, err_len, dlerror());
    g_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!g_glapi) return fail(err, err_len, "libglapi.so.0 not loadable");
    const __DRIextension** (*get_exts)(void) = (const __DRIextension** (*)(void))dlsym(g_drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) return fail(err, err_len, "__driDriverGetExtensions_swrast not exported");
    const __DRIextension** exts = get_exts();
    for (int i = 0; exts && exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!g_core || !g_swrast) return fail(err, err_len, "the driver lacks DRI_Core / DRI_SWRast");
    if (g_swrast->base.version < 4) return fail(err, err_len, "DRI_SWRast older than version 4");
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &g_configs, NULL);
    if (!g_screen) return fail(err, err_len, "createNewScreen2 failed");
    /* any RGBA8 config will do: everything is rendered into framebuffer objects */
    const __DRIconfig* cfg = NULL;
    for (int i = 0; g_configs[i]; i++) {
        unsigned int r = 0, a = 0, db = 0, depth = 0;
        g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
        g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_DEPTH_SIZE, &depth);
        if (r == 8 && a == 8 && !db && depth >= 24) { cfg = g_configs[i]; break; }
        if (!cfg && r == 8) cfg = g_configs[i];
    }
    if (!cfg) return fail(err, err_len, "no RGBA8 config");
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    unsigned cerr = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &cerr, NULL);
    if (!g_ctx) {   /* compatibility profile as a second try */
        const uint32_t attribs2[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 0};
        g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL, cfg, NULL, 2, attribs2, &cerr, NULL);
    }
    if (!g_ctx) { char b[96]; snprintf(b, sizeof b, "createContextAttribs failed (error %u)", cerr); return fail(err, err_len, b); }
    g_draw = g_swrast->createNewDrawable(g_screen, cfg, NULL);
    if (!g_draw) return fail(err, err_len, "createNewDrawable failed");
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) return fail(err, err_len, "bindContext failed");
    return 0;
}

void dri_ctx_destroy(void)
{
    if (g_ctx) { g_core->unbindContext(g_ctx); g_core->destroyContext(g_ctx); g_ctx = NULL; }
    if (g_draw) { g_core->destroyDrawable(g_draw); g_draw = NULL; }
    if (g_screen) { g_core->destroyScreen(g_screen); g_screen = NULL; }
    free(g_fb); g_fb = NULL;
}

void* dri_ctx_proc(const char* name)
{
    void* (*gpa)(const char*) = (void* (*)(const char*))dlsym(g_glapi, "_glapi_get_proc_address");
    return gpa ? gpa(name) : NULL;
}

#ifdef DRI_CTX_MAIN
int main(void)
{
    char err[256] = "";
    if (dri_ctx_create(640, 480, err, sizeof err)) { fprintf(stderr, "no context: %s\n", err); return 1; }
    const GLubyte* (*getString)(GLenum) = (const GLubyte* (*)(GLenum))dri_ctx_proc("glGetString");
    printf("GL_VERSION  %s\nGL_RENDERER %s\nGLSL        %s\n", getString(GL_VERSION), getString(GL_RENDERER), getString(0x8B8C));
    dri_ctx_destroy();
    return 0;
}
#endif
