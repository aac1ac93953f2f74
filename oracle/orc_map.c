/*
 * oracle/orc_map.c -- CPU restatement of the surfel-map half of the path (SURVEY.md 8a rows a1,
 * a9-a15) and of ElasticFusion::processFrame's orchestration.  TEST INFRASTRUCTURE ONLY (orc.h).
 *
 * The reference implements these stages as OpenGL passes (transform feedback + rasterisation).  Since round 6 its UNMODIFIED shaders are executed on Mesa's software
 * rasteriser (oracle/gl/, tools/make_golden_gl.py) and this file is held to their outputs (tests/test_gl_golden.py).  The rules in place of the GL rasteriser, the same
 * the HIP kernels implement:
 *   - 1-px points (index map): the pixel of the projected coordinate snapped to 1/256 px, lower pixel edges inclusive (point_pixel); nearest z wins, ties -> lowest id.
 *   - splats / id discs: a pixel is covered iff the ray through its centre (i+0.5, j+0.5) hits the
 *     surfel disc (EF/Shaders/combo_splat.frag:39-52); nearest intersection z wins, ties -> lowest id.
 *   - association / clean windows: the shaders' FLOAT loop as IEEE f32 runs it (window_taps: four taps per axis, five where the accumulated step falls short of the
 *     bound), each tap read at texel floor(u * size) clamped to the image (EF/Shaders/data.vert:137-153, copy_unstable.vert:110-151, IndexMap::FACTOR = 1).
 */
#define _POSIX_C_SOURCE 199309L
#include <time.h>
#include "orc.h"
#include "orc_math.h"
#include "orc_internal.h"
#include <stdlib.h>
#include <stdio.h>
#include <omp.h>

/* ------------------------------------------------------------------ shared GLSL helpers */

/* EF/Shaders/surfels.glsl:19-34; cam.z = 1/fx, cam.w = 1/fy as uploaded by the callers */
static float get_radius(float depth, float norm_z, float inv_fx, float inv_fy)
{
    float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}

/* EF/Shaders/surfels.glsl:36-46 */
static float confidence_fn(float x, float y, float cx, float cy, float weighting)
{
    const float maxRadDist = 400, twoSigmaSquared = 0.72f;
    float dx = x - cx, dy = y - cy;
    float radialDist = sqrtf(dx * dx + dy * dy) / maxRadDist;
    return ifx_expf((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}

/* EF/Shaders/color.glsl:19-34 */
float orc_encode_color(float r, float g, float b)
{
    int rgb = (int)roundf(r * 255.0f);
    rgb = (rgb << 8) + (int)roundf(g * 255.0f);
    rgb = (rgb << 8) + (int)roundf(b * 255.0f);
    return (float)rgb;
}
void orc_decode_color(float c, float* out3)
{
    int ic = orc_f2i_rz(c);
    out3[0] = (float)(ic >> 16 & 0xFF) / 255.0f;
    out3[1] = (float)(ic >> 8 & 0xFF) / 255.0f;
    out3[2] = (float)(ic & 0xFF) / 255.0f;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* float-depth texel fetch with clamp-to-edge */
static inline float tex_f(const float* img, int w, int h, int x, int y)
{
    return img[clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}

/* geometry.glsl:21-25 (float sampler): x,y are the half-pixel-centred coordinates */
static v3 get_vertex_f(const float* depth, int w, int h, int px, int py, float x, float y, float cx, float cy, float ifx, float ify)
{
    float z = tex_f(depth, w, h, px, py);
    return v3m((x - cx) * z * ifx, (y - cy) * z * ify, z);
}

/* geometry.glsl:28-40 central differences */
static v3 get_normal_f(const float* depth, int w, int h, int px, int py, float x, float y, v3 vp, float cx, float cy, float ifx, float ify)
{
    v3 xf = get_vertex_f(depth, w, h, px + 1, py, x + 1, y, cx, cy, ifx, ify);
    v3 xb = get_vertex_f(depth, w, h, px - 1, py, x - 1, y, cx, cy, ifx, ify);
    v3 yf = get_vertex_f(depth, w, h, px, py + 1, x, y + 1, cx, cy, ifx, ify);
    v3 yb = get_vertex_f(depth, w, h, px, py - 1, x, y - 1, cx, cy, ifx, ify);
    v3 del_x = v3sub(v3scale(v3add(xb, vp), 0.5f), v3scale(v3add(xf, vp), 0.5f));
    v3 del_y = v3sub(v3scale(v3add(yb, vp), 0.5f), v3scale(v3add(yf, vp), 0.5f));
    return v3normalized(v3cross(del_x, del_y));
}

void orc_pose_inverse(const float* p, float* o)
{
    /* rigid inverse (the reference calls Eigen's general pose.inverse(); equal to rounding) */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[i * 4 + j] = p[j * 4 + i];
    for (int i = 0; i < 3; i++) o[i * 4 + 3] = -(o[i * 4] * p[3] + o[i * 4 + 1] * p[7] + o[i * 4 + 2] * p[11]);
    o[12] = o[13] = o[14] = 0; o[15] = 1;
}
static inline v3 xf_point(const float* m, v3 p)
{
    return v3m(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
               m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static inline v3 xf_dir(const float* m, v3 p)
{
    return v3m(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z, m[8] * p.x + m[9] * p.y + m[10] * p.z);
}

/* ------------------------------------------------------------------ object */

orc_t* orc_create(const orc_config* cfg)
{
    orc_t* o = (orc_t*)calloc(1, sizeof(*o));
    o->cfg = *cfg;
    o->w = cfg->width; o->h = cfg->height; o->P = o->w * o->h;
    o->cap = cfg->max_surfels;
    o->tick = 1;
    for (int i = 0; i < 16; i++) o->pose[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    size_t P = (size_t)o->P, C = (size_t)o->cap;
    o->pc = (float*)calloc(C * 4, 4); o->nr = (float*)calloc(C * 4, 4); o->col = (float*)calloc(C * 2, 4);
    o->tm = (float*)calloc(C * 2, 4); o->ic = (float*)calloc(C * 4, 4); o->votes = (float*)calloc(C * ORC_VOTE_FLOATS, 4);
    o->rgb = (uint8_t*)calloc(P * 3, 1);
    o->depth_raw = (uint16_t*)calloc(P, 2); o->depth_filt = (uint16_t*)calloc(P, 2);
    o->dm = (float*)calloc(P, 4); o->dmf = (float*)calloc(P, 4);
    o->index_id = (uint32_t*)calloc(P, 4); o->index_z = (float*)calloc(P, 4);
    o->index_vc = (float*)calloc(P * 4, 4); o->index_ct = (float*)calloc(P * 4, 4); o->index_nr = (float*)calloc(P * 4, 4);
    o->pred_vertex = (float*)calloc(P * 4, 4); o->pred_normal = (float*)calloc(P * 4, 4);
    o->pred_image = (uint8_t*)calloc(P * 4, 1); o->pred_inst = (uint8_t*)calloc(P * 4, 1); o->pred_time = (uint16_t*)calloc(P, 2);
    o->fill_vertex = (float*)calloc(P * 4, 4); o->fill_normal = (float*)calloc(P * 4, 4); o->fill_image = (uint8_t*)calloc(P * 4, 1);
    o->ids_after = (int32_t*)calloc(P, 4); o->ids_tmp = (int32_t*)calloc(P, 4);
    o->zbuf = (float*)calloc(P, 4);
    o->trk = orc_tracker_create(o->w, o->h, cfg->fx, cfg->fy, cfg->cx, cfg->cy);
    orc_instance_init(o);
    return o;
}

void orc_destroy(orc_t* o)
{
    if (!o) return;
    free(o->pc); free(o->nr); free(o->col); free(o->tm); free(o->ic); free(o->votes); free(o->rgb);
    free(o->depth_raw); free(o->depth_filt); free(o->dm); free(o->dmf); free(o->index_id); free(o->index_z);
    free(o->index_vc); free(o->index_ct); free(o->index_nr); free(o->pred_vertex); free(o->pred_normal);
    free(o->pred_image); free(o->pred_inst); free(o->pred_time); free(o->fill_vertex); free(o->fill_normal);
    free(o->fill_image); free(o->ids_after); free(o->ids_tmp); free(o->zbuf); free(o->newbuf); free(o->updbuf);
    orc_tracker_destroy(o->trk);
    orc_tracker_destroy(o->m2m);
    free(o->graph); free(o->inst_gt);
    free(o->old_vertex); free(o->old_normal); free(o->old_image); free(o->old_inst); free(o->old_time);
    orc_instance_free(o);
    free(o);
}

int orc_map_count(orc_t* o) { return o->n; }
void orc_get_pose(orc_t* o, float* out16) { memcpy(out16, o->pose, 64); }
/* instanceGT argument of ElasticFusion::processFrame (EF/ElasticFusion.cpp:285-286): H x W bytes, kept until replaced; NULL switches it off */
void orc_set_instance_gt(orc_t* o, const uint8_t* gt)
{
    free(o->inst_gt);
    o->inst_gt = NULL;
    if (gt) { o->inst_gt = (uint8_t*)malloc((size_t)o->P); memcpy(o->inst_gt, gt, (size_t)o->P); }
}
int orc_tick(orc_t* o) { return o->tick; }
/* lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count of the last tracked frame */
void orc_tracker_diag(orc_t* o, float* out8) { memcpy(out8, o->diag, 32); }
/* the `bootstrap` argument of processFrame for the NEXT orc_process_frame call (needs in_pose16) */
void orc_set_bootstrap(orc_t* o, int on) { o->bootstrap_next = on; }
double orc_now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
/* wall-clock per stage since the last reset: track (preprocessing + tracker) | map passes | instance layer */
void orc_stage_ms(orc_t* o, double* out3, int reset)
{
    for (int k = 0; k < 3; k++) { out3[k] = o->stage_ms[k]; if (reset) o->stage_ms[k] = 0; }
}

void orc_map_download(orc_t* o, float* pc, float* nr, float* col, float* tm, float* ic, float* votes)
{
    size_t n = (size_t)o->n;
    if (pc) memcpy(pc, o->pc, n * 16);
    if (nr) memcpy(nr, o->nr, n * 16);
    if (col) memcpy(col, o->col, n * 8);
    if (tm) memcpy(tm, o->tm, n * 8);
    if (ic) memcpy(ic, o->ic, n * 16);
    if (votes) memcpy(votes, o->votes, n * ORC_VOTE_FLOATS * 4);
}

void orc_map_upload(orc_t* o, int n, const float* pc, const float* nr, const float* col,
                    const float* tm, const float* ic, const float* votes)
{
    if (n > o->cap) n = o->cap;
    o->n = n;
    memcpy(o->pc, pc, (size_t)n * 16);
    memcpy(o->nr, nr, (size_t)n * 16);
    memcpy(o->col, col, (size_t)n * 8);
    memcpy(o->tm, tm, (size_t)n * 8);
    if (ic) memcpy(o->ic, ic, (size_t)n * 16); else memset(o->ic, 0, (size_t)n * 16);
    if (votes) memcpy(o->votes, votes, (size_t)n * ORC_VOTE_FLOATS * 4); else memset(o->votes, 0, (size_t)n * ORC_VOTE_FLOATS * 4);
}

void orc_set_pose(orc_t* o, const float* pose16, int tick)
{
    memcpy(o->pose, pose16, 64);
    o->tick = tick;
}

const void* orc_image(orc_t* o, const char* name)
{
    if (!strcmp(name, "ids_after")) return o->ids_after;
    if (!strcmp(name, "ids_tmp")) return o->ids_tmp;
    if (!strcmp(name, "index")) return o->index_id;
    if (!strcmp(name, "index_vc")) return o->index_vc;
    if (!strcmp(name, "index_ct")) return o->index_ct;
    if (!strcmp(name, "index_nr")) return o->index_nr;
    if (!strcmp(name, "pred_vertex")) return o->pred_vertex;
    if (!strcmp(name, "pred_normal")) return o->pred_normal;
    if (!strcmp(name, "pred_image")) return o->pred_image;
    if (!strcmp(name, "pred_inst")) return o->pred_inst;
    if (!strcmp(name, "pred_time")) return o->pred_time;
    if (!strcmp(name, "fill_vertex")) return o->fill_vertex;
    if (!strcmp(name, "fill_normal")) return o->fill_normal;
    if (!strcmp(name, "fill_image")) return o->fill_image;
    if (!strcmp(name, "old_vertex")) return o->old_vertex;
    if (!strcmp(name, "old_normal")) return o->old_normal;
    if (!strcmp(name, "old_image")) return o->old_image;
    if (!strcmp(name, "old_time")) return o->old_time;
    if (!strcmp(name, "depth_filtered")) return o->depth_filt;
    if (!strcmp(name, "depth_metric")) return o->dm;
    if (!strcmp(name, "depth_metric_filtered")) return o->dmf;
    return 0;
}

/* ------------------------------------------------------------------ first frame (a15)
 * vertex_feedback.vert:41-74 + .geom:35-45 + init_unstable.vert:45-67.  The reference binds the
 * 48-B feedback records with a 256-B stride; the INTENDED dense initialisation is
 * implemented: one surfel per pixel whose raw AND filtered depth are valid, column-major order. */
static void init_first_frame(orc_t* o)
{
    int w = o->w, h = o->h;
    float cx = o->cfg.cx, cy = o->cfg.cy, ifx = 1.0f / o->cfg.fx, ify = 1.0f / o->cfg.fy;
    float maxDepth = o->cfg.max_depth_processed;
    int n = 0;
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            if (n >= o->cap) break;
            float x = (float)i + 0.5f, y = (float)j + 0.5f;
            v3 vp = get_vertex_f(o->dm, w, h, i, j, x, y, cx, cy, ifx, ify);
            v3 vpf = get_vertex_f(o->dmf, w, h, i, j, x, y, cx, cy, ifx, ify);
            if (vp.z <= 0 || vp.z > maxDepth || vpf.z <= 0 || vpf.z > maxDepth) continue;
            v3 nl = get_normal_f(o->dmf, w, h, i, j, x, y, vpf, cx, cy, ifx, ify);
            float* pc = &o->pc[n * 4];
            pc[0] = vp.x; pc[1] = vp.y; pc[2] = vp.z; pc[3] = confidence_fn(x, y, cx, cy, 1.0f);
            const uint8_t* c = &o->rgb[(j * w + i) * 3];
            o->col[n * 2] = orc_encode_color(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f);
            o->col[n * 2 + 1] = 0;
            o->tm[n * 2] = 1; /* init_unstable.vert:49 */
            o->tm[n * 2 + 1] = (float)o->tick;
            float* nr = &o->nr[n * 4];
            nr[0] = nl.x; nr[1] = nl.y; nr[2] = nl.z; nr[3] = get_radius(vpf.z, nl.z, ifx, ify);
            for (int k = 0; k < 4; k++) o->ic[n * 4 + k] = -1.0f;
            for (int k = 0; k < ORC_VOTE_FLOATS; k++) o->votes[(size_t)n * ORC_VOTE_FLOATS + k] = -1.0f;
            n++;
        }
    o->n = n;
}

/* ------------------------------------------------------------------ parallel z-buffers that keep the draw order
 * Every render of the map is "nearest z wins, on equal z the surfel drawn first (lowest index)" (GL_LESS + draw order).  The sequential loops
 * of round 2 got that from `z < zbuf` over i = 0, 1, ...; here thread t draws the contiguous index range [t n / T, (t + 1) n / T) into a z-buffer of its own
 * (depth + index per pixel), and the buffers are merged per pixel in thread order with the same strict `<` -- lower threads hold lower indices, so the winner is
 * the one the sequential loop picks, for any number of threads.  The winner's outputs are then written by a per-pixel resolve. */
typedef struct { int T; size_t P; float* z; int* id; } zbufs_t;
/* the buffers are kept between renders (one set per process, grown on demand: a render of 640x480 on 64 threads would otherwise allocate, fill and free
 * 157 MB several times per frame, which is the cpu_baseline's time, not the algorithm's); the oracle renders one image at a time */
static float* zb_keep_z = NULL; static int* zb_keep_id = NULL; static size_t zb_keep_n = 0;
static zbufs_t zb_open(size_t P)
{
    zbufs_t b;
    b.T = omp_get_max_threads(); b.P = P;
    if (b.T > 64) b.T = 64;
    const size_t need = (size_t)b.T * P;
    if (need > zb_keep_n) {
        free(zb_keep_z); free(zb_keep_id);
        zb_keep_z = (float*)malloc(need * sizeof(float)); zb_keep_id = (int*)malloc(need * sizeof(int)); zb_keep_n = need;
    }
    b.z = zb_keep_z; b.id = zb_keep_id;
#pragma omp parallel for schedule(static)
    for (long k = 0; k < (long)need; k++) { b.z[k] = INFINITY; b.id[k] = -1; }
    return b;
}
/* merged winner per pixel -> z_out / id_out (-1: nothing drawn) */
static void zb_merge(zbufs_t* b, float* z_out, int* id_out)
{
#pragma omp parallel for schedule(static)
    for (long k = 0; k < (long)b->P; k++) {
        float z = INFINITY; int id = -1;
        for (int t = 0; t < b->T; t++) { const float zt = b->z[(size_t)t * b->P + k]; if (zt < z) { z = zt; id = b->id[(size_t)t * b->P + k]; } }
        z_out[k] = z; id_out[k] = id;
    }
}
/* range t of T is drawn by ONE thread of the team, whatever size the team came out (OMP_DYNAMIC, thread limits, nested regions may deliver fewer than T
 * threads): ZB_FOR_RANGES walks t = thread, thread + team size, ... so that no range is ever left undrawn */
#define ZB_FOR_RANGES(t, T) for (int t = omp_get_thread_num(), zb_step_ = omp_get_num_threads(); t < (T); t += zb_step_)
static inline void zb_range(long n, int T, int t, long* lo, long* hi) { *lo = n * t / T; *hi = n * (t + 1) / T; }

/* ------------------------------------------------------------------ index map (a10)
 * index_map.vert:40-66 / index_map.frag:33-40 under GL_LESS. */
/* The pixel a 1-pixel GL point at window coordinate u lands on: the position is snapped to the rasteriser's sub-pixel grid (8 bits: GL_SUBPIXEL_BITS of Mesa llvmpipe and of
 * the NVIDIA GPUs the reference ran on), the point is the 1x1 square around it, and a pixel is produced when its centre lies in that square, the lower edge included and the
 * upper one not (top-left rule): floor(u) -- except that a point within 1/512 px above a pixel edge belongs to the pixel BELOW the edge.  Measured on the reference's
 * index_map shaders (tools/make_golden_gl.py: 47 of 15 697 points; rounds 1-5 had floor(u)). */
static inline int point_pixel(float u) { return (int)floorf((rintf(u * 256.0f) - 1.0f) / 256.0f); }

void orc_predict_indices(orc_t* o, const float* pose, int time)
{
    int w = o->w, h = o->h;
    float tinv[16];
    orc_pose_inverse(pose, tinv);
    float cx = o->cfg.cx, cy = o->cfg.cy, fx = o->cfg.fx, fy = o->cfg.fy;
    float maxDepth = o->cfg.max_depth_processed;
    zbufs_t zb = zb_open((size_t)o->P);
#pragma omp parallel num_threads(zb.T)
    {
        ZB_FOR_RANGES(t, zb.T) {
        long lo, hi;
        zb_range(o->n, zb.T, t, &lo, &hi);
        float* tz = zb.z + (size_t)t * zb.P; int* ti = zb.id + (size_t)t * zb.P;
        for (long i = lo; i < hi; i++) {
            v3 p = xf_point(tinv, v3m(o->pc[i * 4], o->pc[i * 4 + 1], o->pc[i * 4 + 2]));
            if (p.z > maxDepth || p.z < 0 || (float)time - o->tm[i * 2 + 1] > (float)o->cfg.time_delta) continue;
            float u = ((fx * p.x) / p.z) + cx, v = ((fy * p.y) / p.z) + cy;
            if (!(u >= 0 && u < (float)w && v >= 0 && v < (float)h)) continue;
            const int px = point_pixel(u), py = point_pixel(v);
            if (px < 0 || py < 0) continue;
            int k = py * w + px;
            if (p.z < tz[k]) { tz[k] = p.z; ti[k] = (int)i; }
        }
        }
    }
    int* win = (int*)malloc((size_t)o->P * sizeof(int));
    zb_merge(&zb, o->index_z, win);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < o->P; k++) {
        float* vc = &o->index_vc[k * 4]; float* ct = &o->index_ct[k * 4]; float* nr = &o->index_nr[k * 4];
        const int i = win[k];
        if (i < 0) { o->index_id[k] = 0; for (int c = 0; c < 4; c++) { vc[c] = 0; ct[c] = 0; nr[c] = 0; } continue; }
        v3 p = xf_point(tinv, v3m(o->pc[i * 4], o->pc[i * 4 + 1], o->pc[i * 4 + 2]));
        o->index_id[k] = (uint32_t)i;
        vc[0] = p.x; vc[1] = p.y; vc[2] = p.z; vc[3] = o->pc[i * 4 + 3];
        ct[0] = o->col[i * 2]; ct[1] = o->col[i * 2 + 1]; ct[2] = o->tm[i * 2]; ct[3] = o->tm[i * 2 + 1];
        v3 nn = v3normalized(xf_dir(tinv, v3m(o->nr[i * 4], o->nr[i * 4 + 1], o->nr[i * 4 + 2])));
        nr[0] = nn.x; nr[1] = nn.y; nr[2] = nn.z; nr[3] = o->nr[i * 4 + 3];
    }
    free(win);
}

/* ------------------------------------------------------------------ disc rasteriser
 * combo_splat.frag:39-52.  Calls emit(k, z) for every covered pixel in [x0,x1]x[y0,y1]. */
typedef struct { v3 q, n; float r2; } disc_t;
static inline int disc_hit(const disc_t* d, float px, float py, float cx, float cy, float fx, float fy, float* zout)
{
    /* combo_splat.frag:39-46 normalises the ray first; the intersection l*(q.n)/(l.n) does not depend
     * on |l|, so the (shared oracle / HIP) rule uses the un-normalised ray ((px-cx)/fx, (py-cy)/fy, 1) */
    v3 l = v3m((px - cx) * (1.0f / fx), (py - cy) * (1.0f / fy), 1.0f);
    float s = v3dot(d->q, d->n) / v3dot(l, d->n);
    v3 cp = v3scale(l, s);
    v3 df = v3sub(cp, d->q);
    if (!(v3dot(df, df) <= d->r2)) return 0;
    *zout = cp.z;
    return 1;
}

/* splat.vert:55-92: in-plane extremes and their image-space bounding box */
static void disc_extent(v3 q, v3 n, float r, float cx, float cy, float fx, float fy, float* xs, float* ys, float* minz)
{
    v3 x1 = v3scale(v3normalized(v3m(n.y - n.z, -n.x, n.x)), r * 1.41421356f);
    v3 y1 = v3cross(n, x1);
    v3 pts[4] = {v3add(q, x1), v3add(q, y1), v3sub(q, y1), v3sub(q, x1)};
    xs[0] = ys[0] = INFINITY; xs[1] = ys[1] = -INFINITY; *minz = INFINITY;
    for (int k = 0; k < 4; k++) {
        float u = ((fx * pts[k].x) / pts[k].z) + cx, v = ((fy * pts[k].y) / pts[k].z) + cy;
        xs[0] = fminf(xs[0], u); xs[1] = fmaxf(xs[1], u);
        ys[0] = fminf(ys[0], v); ys[1] = fmaxf(ys[1], v);
        *minz = fminf(*minz, pts[k].z);
    }
}

#define ORC_MAX_SPRITE 512.0f

/* ------------------------------------------------------------------ splat prediction (a9)
 * IndexMap::combinedPredict, EF/IndexMap.cpp:468-574 + splat.vert + combo_splat.frag, ACTIVE when
 * time == max_time == tick; INACTIVE when time = 0, max_time = tick - timeDelta. */
void orc_combined_predict(orc_t* o, const float* pose, int time, int max_time)
{
    int w = o->w, h = o->h;
    float tinv[16];
    orc_pose_inverse(pose, tinv);
    float cx = o->cfg.cx, cy = o->cfg.cy, fx = o->cfg.fx, fy = o->cfg.fy;
    float maxDepth = o->cfg.max_depth_processed, thr = o->cfg.confidence;
    zbufs_t zb = zb_open((size_t)o->P);
#pragma omp parallel num_threads(zb.T)
    {
        ZB_FOR_RANGES(t, zb.T) {
        long lo, hi;
        zb_range(o->n, zb.T, t, &lo, &hi);
        float* tz = zb.z + (size_t)t * zb.P; int* ti = zb.id + (size_t)t * zb.P;
        for (long i = lo; i < hi; i++) {
            v3 q = xf_point(tinv, v3m(o->pc[i * 4], o->pc[i * 4 + 1], o->pc[i * 4 + 2]));
            float conf = o->pc[i * 4 + 3], lastT = o->tm[i * 2 + 1];
            if (q.z > maxDepth || q.z < 0 || conf < thr || (float)time - lastT > (float)o->cfg.time_delta || lastT > (float)max_time) continue;
            float u = ((fx * q.x) / q.z) + cx, v = ((fy * q.y) / q.z) + cy;
            if (!(u >= 0 && u <= (float)w && v >= 0 && v <= (float)h)) continue; /* GL clips points by centre */
            v3 n = v3normalized(xf_dir(tinv, v3m(o->nr[i * 4], o->nr[i * 4 + 1], o->nr[i * 4 + 2])));
            float r = o->nr[i * 4 + 3];
            float xs[2], ys[2], minz;
            disc_extent(q, n, r, cx, cy, fx, fy, xs, ys, &minz);
            float s = fmaxf(fabsf(xs[1] - xs[0]), fabsf(ys[1] - ys[0]));
            if (!(s == s)) continue;
            s = fminf(fmaxf(s, 1.0f), ORC_MAX_SPRITE);
            int x0 = clampi((int)ceilf(u - s * 0.5f - 0.5f), 0, w - 1), x1 = clampi((int)floorf(u + s * 0.5f - 0.5f), 0, w - 1);
            int y0 = clampi((int)ceilf(v - s * 0.5f - 0.5f), 0, h - 1), y1 = clampi((int)floorf(v + s * 0.5f - 0.5f), 0, h - 1);
            disc_t d = {q, n, r * r};
            for (int py = y0; py <= y1; py++)
                for (int px = x0; px <= x1; px++) {
                    float z;
                    if (!disc_hit(&d, (float)px + 0.5f, (float)py + 0.5f, cx, cy, fx, fy, &z)) continue;
                    if (!(z >= -maxDepth && z <= maxDepth)) continue; /* gl_FragDepth in [0,1] */
                    int k = py * w + px;
                    if (z < tz[k]) { tz[k] = z; ti[k] = (int)i; }
                }
        }
        }
    }
    int* win = (int*)malloc((size_t)o->P * sizeof(int));
    zb_merge(&zb, o->zbuf, win);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < o->P; k++) {
        float* vo = &o->pred_vertex[k * 4]; float* no = &o->pred_normal[k * 4];
        const int i = win[k];
        if (i < 0) {
            for (int c = 0; c < 4; c++) { vo[c] = 0; no[c] = 0; o->pred_image[k * 4 + c] = 0; o->pred_inst[k * 4 + c] = 0; }
            o->pred_time[k] = 0;
            continue;
        }
        const int px = k % w, py = k / w;
        const float z = o->zbuf[k], fpx = (float)px + 0.5f, fpy = (float)py + 0.5f;
        v3 n = v3normalized(xf_dir(tinv, v3m(o->nr[i * 4], o->nr[i * 4 + 1], o->nr[i * 4 + 2])));
        vo[0] = (fpx - cx) * z * (1.f / fx); vo[1] = (fpy - cy) * z * (1.f / fy); vo[2] = z; vo[3] = o->pc[i * 4 + 3];
        no[0] = n.x; no[1] = n.y; no[2] = n.z; no[3] = o->nr[i * 4 + 3];
        float c3[3];
        orc_decode_color(o->col[i * 2], c3);
        for (int c = 0; c < 3; c++) o->pred_image[k * 4 + c] = (uint8_t)(int)roundf(c3[c] * 255.0f);
        o->pred_image[k * 4 + 3] = 255;
        orc_decode_color(o->col[i * 2 + 1], c3);
        for (int c = 0; c < 3; c++) o->pred_inst[k * 4 + c] = (uint8_t)(int)roundf(c3[c] * 255.0f);
        o->pred_inst[k * 4 + 3] = 255;
        o->pred_time[k] = (uint16_t)(uint32_t)o->tm[i * 2];
    }
    free(win);
}

/* ------------------------------------------------------------------ fill-in
 * EF/Shaders/FillIn.cpp:65-195 with fill_rgb/vertex/normal.frag (passthrough = 0). */
static void fill_in(orc_t* o)
{
    int w = o->w, h = o->h;
    float cx = o->cfg.cx, cy = o->cfg.cy, ifx = 1.0f / o->cfg.fx, ify = 1.0f / o->cfg.fy;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int k = y * w + x;
            /* image */
            const uint8_t* s = &o->pred_image[k * 4];
            if ((int)s[0] + (int)s[1] + (int)s[2] == 0) {
                for (int c = 0; c < 3; c++) o->fill_image[k * 4 + c] = o->rgb[k * 3 + c];
                o->fill_image[k * 4 + 3] = 255;
            } else memcpy(&o->fill_image[k * 4], s, 4);
            /* vertex: fill_vertex.frag:40-52, integer pixel coordinates, filtered u16 depth */
            if (o->pred_vertex[k * 4 + 2] == 0) {
                float z = (float)o->depth_filt[k] / 1000.0f;
                float* vo = &o->fill_vertex[k * 4];
                vo[0] = ((float)x - cx) * z * ifx; vo[1] = ((float)y - cy) * z * ify; vo[2] = z; vo[3] = 1;
            } else memcpy(&o->fill_vertex[k * 4], &o->pred_vertex[k * 4], 16);
            /* normal: geometry.glsl:45-60 forward differences on the u16 depth, clamp to edge */
            if (o->pred_normal[k * 4 + 2] == 0) {
                float z = (float)o->depth_filt[k] / 1000.0f;
                v3 vp = v3m(((float)x - cx) * z * ifx, ((float)y - cy) * z * ify, z);
                int xr = clampi(x + 1, 0, w - 1), yd = clampi(y + 1, 0, h - 1);
                float zx = (float)o->depth_filt[y * w + xr] / 1000.0f, zy = (float)o->depth_filt[yd * w + x] / 1000.0f;
                v3 vx = v3m(((float)(x + 1) - cx) * zx * ifx, ((float)y - cy) * zx * ify, zx);
                v3 vy = v3m(((float)x - cx) * zy * ifx, ((float)(y + 1) - cy) * zy * ify, zy);
                v3 nn = v3normalized(v3cross(v3sub(vx, vp), v3sub(vy, vp)));
                float* no = &o->fill_normal[k * 4];
                no[0] = nn.x; no[1] = nn.y; no[2] = nn.z; no[3] = 1;
            } else memcpy(&o->fill_normal[k * 4], &o->pred_normal[k * 4], 16);
        }
}

/* ElasticFusion::denseEnough, EF/ElasticFusion.cpp:252-267 on a (w/20 x h/20) nearest resample */
static int dense_enough(orc_t* o)
{
    int rw = o->w / 20, rh = o->h / 20, sum = 0;
    for (int j = 0; j < rh; j++)
        for (int i = 0; i < rw; i++) {
            int sx = (i * o->w + o->w / 2) / rw, sy = (j * o->h + o->h / 2) / rh;
            const uint8_t* s = &o->pred_image[(sy * o->w + sx) * 4];
            sum += s[0] > 0 && s[1] > 0 && s[2] > 0;
        }
    return (float)sum / (float)(rw * rh) > 0.75f;
}

/* ------------------------------------------------------------------ fuse (a11, a12) */
typedef orc_meas meas_t;

/* The window loop of data.vert:151-153 and copy_unstable.vert:110-112 AS THE SHADER TEXT EVALUATES IT in IEEE f32:
 *     for (float i = c - (scale * step * windowMultiplier); i < c + (scale * step * windowMultiplier); i += step)   with step = (1 / (size * scale)) * 0.5, scale = 1
 * and the texel a tap reads: nearest filtering = floor(u * size), clamped to the edge (GL 4.5 section 8.14.2; GPUTexture: GL_NEAREST).  In exact arithmetic the loop
 * makes 4 trips, at -1, -1/2, 0, +1/2 texels from c; in f32 the accumulated `i += step` falls short of the bound in a fraction of the cases (6 % of the projected
 * positions at 160 columns, 53 % at 320, 26 % at 640) and a FIFTH tap, one texel beyond c, is taken -- and a tap that sits on a texel edge (the +-1/2 taps of data.vert,
 * whose c is a texel centre) goes to whichever side the f32 product falls.  Pinned by running the reference's shaders (tests/golden/gl_map_passes.npz,
 * tools/make_golden_gl.py): rounds 1-5 assumed the 4 exact taps.  c: the normalised coordinate (x / cols, or the texcoord attribute); returns the number of taps. */
#define ORC_MAX_TAPS 5   /* the span is four steps wide: rounding can add a trip, never two (lo + 5 step is a whole step beyond hi), and never takes one away */
static int window_taps(float c, float size, int n, int* tex)
{
    const float scale = 1.0f, wm = 2.0f;
    const float step = (1.0f / (size * scale)) * 0.5f;
    const float lo = c - (scale * step * wm), hi = c + (scale * step * wm);
    int k = 0;
    for (float i = lo; i < hi; i += step) {
        if (k >= ORC_MAX_TAPS) { fprintf(stderr, "orc: window loop with more than %d taps (c = %.9g, size = %g)\n", ORC_MAX_TAPS, (double)c, (double)size); abort(); }
        tex[k++] = clampi((int)floorf(i * size), 0, n - 1);
    }
    return k;
}
/* the texcoord attribute of pixel column / row i: the uvo buffer of GlobalModel (EF/GlobalModel.cpp:103-119), float(i) / size + 1.0 / (2 * size) evaluated in double and stored as float */
static float uvo_coord(int i, int size) { return (float)((double)((float)i / (float)size) + 1.0 / (2 * (double)(float)size)); }

/* test hooks (tests/test_oracle_cpu.py pins the three rules against their numpy statement from the shader text) */
int orc_test_window_taps(float c, float size, int n, int* tex) { return window_taps(c, size, n, tex); }
float orc_test_uvo_coord(int i, int size) { return uvo_coord(i, size); }
int orc_test_point_pixel(float u) { return point_pixel(u); }

/* data.vert:94-241 for the pixel (i,j) */
static void associate_pixel(orc_t* o, const float* pose, int time, float weighting, int i, int j, meas_t* m)
{
    int w = o->w, h = o->h;
    float cx = o->cfg.cx, cy = o->cfg.cy, ifx = 1.0f / o->cfg.fx, ify = 1.0f / o->cfg.fy;
    float maxDepth = o->cfg.max_depth_processed;
    float x = (float)i + 0.5f, y = (float)j + 0.5f;
    m->kind = 0; m->target = 0;
    if (!(i % 2 == time % 2 && j % 2 == time % 2)) return;
    v3 vl = get_vertex_f(o->dm, w, h, i, j, x, y, cx, cy, ifx, ify);
    /* checkNeighbours :71-90 */
    if (tex_f(o->dm, w, h, i - 1, j) == 0 || tex_f(o->dm, w, h, i, j - 1) == 0 || tex_f(o->dm, w, h, i + 1, j) == 0 || tex_f(o->dm, w, h, i, j + 1) == 0) return;
    if (!(vl.z > 0 && vl.z <= maxDepth)) return;
    v3 vg = xf_point(pose, vl);
    v3 vf = get_vertex_f(o->dmf, w, h, i, j, x, y, cx, cy, ifx, ify);
    v3 nl = get_normal_f(o->dmf, w, h, i, j, x, y, vf, cx, cy, ifx, ify);
    v3 ng = xf_dir(pose, nl);
    m->pc[0] = vg.x; m->pc[1] = vg.y; m->pc[2] = vg.z; m->pc[3] = confidence_fn(x, y, cx, cy, weighting);
    const uint8_t* c = &o->rgb[(j * w + i) * 3];
    m->col0 = orc_encode_color(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f);
    m->nr[0] = ng.x; m->nr[1] = ng.y; m->nr[2] = ng.z; m->nr[3] = get_radius(vf.z, nl.z, ifx, ify);
    m->ic[0] = x; m->ic[1] = y; m->ic[2] = (float)time;
    m->ic[3] = o->inst_gt ? (float)o->inst_gt[j * w + i] : -2.0f;   /* data.vert:215-228: the ground-truth instance id of the creating pixel */

    float xl = (x - cx) * ifx, yl = (y - cy) * ify;
    float lambda = sqrtf(xl * xl + yl * yl + 1);
    v3 ray = v3m(xl, yl, 1);
    float rayLen = v3norm(ray);
    float bestDist = 1000;
    uint32_t best = 0;
    int counter = 0;
    int txs[ORC_MAX_TAPS], tys[ORC_MAX_TAPS];
    const int ntx = window_taps(uvo_coord(i, w), (float)w, w, txs), nty = window_taps(uvo_coord(j, h), (float)h, h, tys);
    for (int a = 0; a < ntx; a++)
        for (int b = 0; b < nty; b++) { /* x outer, y inner as in data.vert:151-153 */
            int tx = txs[a], ty = tys[b];
            int k = ty * w + tx;
            uint32_t cur = o->index_id[k];
            if (cur > 0u) {
                const float* vc = &o->index_vc[k * 4];
                if (fabsf((vc[2] * lambda) - (vl.z * lambda)) < 0.05f) {
                    float dist = v3norm(v3cross(ray, v3m(vc[0], vc[1], vc[2]))) / rayLen;
                    const float* nrm = &o->index_nr[k * 4];
                    v3 nn = v3m(nrm[0], nrm[1], nrm[2]);
                    /* acos(c) < 0.5  <=>  c > cos(0.5)  (monotone; avoids a transcendental) */
                    float cang = v3dot(nn, nl) / (v3norm(nn) * v3norm(nl));
                    if (dist < bestDist && (fabsf(nrm[2]) < 0.75f || cang > 0.87758256189f)) {
                        counter++; bestDist = dist; best = cur;
                    }
                }
            }
        }
    if (counter > 0) { m->kind = 1; m->target = best; }
    else m->kind = 2;
}

/* update.vert:55-141 applied in place */
static void apply_update(orc_t* o, uint32_t id, const meas_t* m, int time)
{
    float* pc = &o->pc[id * 4]; float* nr = &o->nr[id * 4];
    float c_k = pc[3], a = m->pc[3];
    if (m->nr[3] < (1.0f + 0.5f) * nr[3]) {
        for (int k = 0; k < 3; k++) pc[k] = ((c_k * pc[k]) + (a * m->pc[k])) / (c_k + a);
        pc[3] = c_k + a;
        float oc[3], nc[3], av[3];
        orc_decode_color(o->col[id * 2], oc);
        orc_decode_color(m->col0, nc);
        for (int k = 0; k < 3; k++) av[k] = ((c_k * oc[k]) + (a * nc[k])) / (c_k + a);
        o->col[id * 2] = orc_encode_color(av[0], av[1], av[2]);
        o->tm[id * 2 + 1] = (float)time;
        float t4[4];
        for (int k = 0; k < 4; k++) t4[k] = ((c_k * nr[k]) + (a * m->nr[k])) / (c_k + a);
        v3 nn = v3normalized(v3m(t4[0], t4[1], t4[2]));
        nr[0] = nn.x; nr[1] = nn.y; nr[2] = nn.z; nr[3] = t4[3];
    } else {
        pc[3] = c_k + a;
        o->tm[id * 2 + 1] = (float)time;
    }
}

/* ------------------------------------------------------------------ clean test (a13)
 * copy_unstable.vert:103-174.  rec: pc, nr, tm of the candidate.  Returns keep flag and the
 * resolved lastTime. */
static int clean_test(orc_t* o, const float* tinv, int time, const float* pc, const float* nr, float initT, float* lastT)
{
    int w = o->w, h = o->h;
    float cx = o->cfg.cx, cy = o->cfg.cy, fx = o->cfg.fx, fy = o->cfg.fy, thr = o->cfg.confidence;
    int timeDelta = o->cfg.time_delta;
    int test = 1;
    v3 lp = xf_point(tinv, v3m(pc[0], pc[1], pc[2]));
    float x = ((fx * lp.x) / lp.z) + cx, y = ((fy * lp.y) / lp.z) + cy;
    v3 ln = v3normalized(xf_dir(tinv, v3m(nr[0], nr[1], nr[2])));
    int count = 0, zCount = 0;
    float wv = *lastT;
    if ((float)time - wv < (float)timeDelta && lp.z > 0 && x > 0 && y > 0 && x < (float)w && y < (float)h) {
        int txs[ORC_MAX_TAPS], tys[ORC_MAX_TAPS];
        const int ntx = window_taps(x / (float)w, (float)w, w, txs), nty = window_taps(y / (float)h, (float)h, h, tys);
        for (int a = 0; a < ntx; a++)
            for (int b = 0; b < nty; b++) {
                int tx = txs[a], ty = tys[b];
                int k = ty * w + tx;
                if (o->index_id[k] > 0u) {
                    const float* vc = &o->index_vc[k * 4];
                    const float* ct = &o->index_ct[k * 4];
                    float dx = vc[0] - lp.x, dy = vc[1] - lp.y;
                    if (ct[2] < initT && vc[3] > thr && vc[2] > lp.z && vc[2] - lp.z < 0.01f && sqrtf(dx * dx + dy * dy) < nr[3] * 1.4f) count++;
                    if (ct[3] == (float)time && vc[3] > thr && vc[2] > lp.z && vc[2] - lp.z > 0.01f && fabsf(ln.z) > 0.85f) zCount++;
                }
            }
    }
    if (count > 8 || zCount > 4) test = 0;
    if (wv == -2) wv = (float)time;
    if (wv == -1 || (((float)time - wv) > 20 && pc[3] < thr)) test = 0;
    if (wv > 0 && (float)time - wv > (float)timeDelta) test = 1;
    *lastT = wv;
    return test;
}

/* GlobalModel::fuse, EF/GlobalModel.cpp:459-698 (data pass + update pass).  New unstable surfels
 * are kept in o->newbuf (column-major pixel order) for orc_clean to append. */
void orc_fuse(orc_t* o, const float* pose, int time, float weighting)
{
    int w = o->w, h = o->h;
    if (!o->newbuf) o->newbuf = (orc_meas*)calloc((size_t)o->P, sizeof(orc_meas));
    if (!o->updbuf) o->updbuf = (orc_meas*)calloc((size_t)o->P, sizeof(orc_meas));
    o->n_new = 0;
    o->n_upd = 0;
    uint8_t* touched = (uint8_t*)calloc((size_t)o->n + 1, 1);
    /* the association of a pixel reads the frame and the index map only: all pixels in parallel (the active ones: one in four), then the scan in column-major
     * order that decides who owns an update texel and in which order the new surfels are appended */
    const int par = time % 2, aw = (w - par + 1) / 2, ah = (h - par + 1) / 2;
    meas_t* all = (meas_t*)malloc((size_t)aw * ah * sizeof(meas_t));
#pragma omp parallel for schedule(static)
    for (int ai = 0; ai < aw; ai++)
        for (int aj = 0; aj < ah; aj++) associate_pixel(o, pose, time, weighting, 2 * ai + par, 2 * aj + par, &all[(size_t)ai * ah + aj]);
    for (int ai = 0; ai < aw; ai++)
        for (int aj = 0; aj < ah; aj++) {
            const meas_t* m = &all[(size_t)ai * ah + aj];
            if (m->kind == 1) {
                /* every update point is drawn at z = 0 under GL_LESS: the first
                 * pixel in column-major order that targets a surfel owns its update texel */
                if (m->target < (uint32_t)o->n && !touched[m->target]) {
                    touched[m->target] = 1;
                    o->updbuf[o->n_upd++] = *m;
                }
            } else if (m->kind == 2) o->newbuf[o->n_new++] = *m;
        }
    free(all);
    /* each update reads and writes only its own surfel (one update per surfel), so applying after the scan, in any order, equals update.vert */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < o->n_upd; k++) apply_update(o, o->updbuf[k].target, &o->updbuf[k], time);
    free(touched);
}

/* GlobalModel::clean, EF/GlobalModel.cpp:700-925 + copy_unstable.vert/.geom: order-preserving
 * compaction of the survivors followed by the surviving new unstable surfels.  Requires the index
 * map of the post-fuse state (orc_predict_indices).  No deformation graph (out of scope). */
void orc_clean(orc_t* o, const float* pose, int time)
{
    float tinv[16];
    orc_pose_inverse(pose, tinv);
    int m = 0;
    const float* depth = NULL;
    if (o->graph_nodes > 0 && !o->graph_is_fern) {
        /* IndexMap::synthesizeDepth (EF/ElasticFusion.cpp:667-676, EF/IndexMap.cpp:576-648): splat.vert with time = tick, maxTime = tick - timeDelta,
         * timeDelta = 65535 culls exactly what the INACTIVE prediction culls, and depth_splat.frag writes the z of the same ray-disc intersection:
         * the depth image is the z channel of an INACTIVE prediction of the post-fuse map at the (adopted) pose.  Rendered into the old* images. */
        float *pv = o->pred_vertex, *pn = o->pred_normal;
        uint8_t *pi = o->pred_image, *ps = o->pred_inst;
        uint16_t* pt = o->pred_time;
        o->pred_vertex = o->old_vertex; o->pred_normal = o->old_normal; o->pred_image = o->old_image; o->pred_inst = o->old_inst; o->pred_time = o->old_time;
        orc_combined_predict(o, pose, 0, time - o->cfg.time_delta);
        o->pred_vertex = pv; o->pred_normal = pn; o->pred_image = pi; o->pred_inst = ps; o->pred_time = pt;
        float* d = o->zbuf;   /* free between passes */
        for (int k = 0; k < o->P; k++) d[k] = o->old_vertex[(size_t)k * 4 + 2];
        depth = d;
    }
    /* the stability test of a surfel reads its own record and the index map only: all surfels in parallel, then the order-preserving compaction */
    uint8_t* keepf = (uint8_t*)malloc((size_t)o->n + 1);
    float* lastTs = (float*)malloc(((size_t)o->n + 1) * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < o->n; i++) {
        float lt = o->tm[i * 2 + 1];
        keepf[i] = (uint8_t)clean_test(o, tinv, time, &o->pc[i * 4], &o->nr[i * 4], o->tm[i * 2], &lt);
        lastTs[i] = lt;
    }
    for (int i = 0; i < o->n; i++) {
        float lastT = lastTs[i];
        if (!keepf[i]) continue;
        if (o->graph_nodes > 0 && o->tm[i * 2] != (float)time)   /* copy_unstable.vert:178-181: survivors not created this frame */
            orc_deform_surfel(o->graph, o->graph_nodes, &o->pc[i * 4], &o->nr[i * 4], o->tm[i * 2], &lastT, time, o->cfg.confidence, o->graph_is_fern, tinv, depth,
                              o->w, o->h, o->cfg.cx, o->cfg.cy, o->cfg.fx, o->cfg.fy, o->cfg.max_depth_processed);
        if (m != i) {
            memcpy(&o->pc[m * 4], &o->pc[i * 4], 16);
            memcpy(&o->nr[m * 4], &o->nr[i * 4], 16);
            memcpy(&o->col[m * 2], &o->col[i * 2], 8);
            memcpy(&o->tm[m * 2], &o->tm[i * 2], 8);
            memcpy(&o->ic[m * 4], &o->ic[i * 4], 16);
            memcpy(&o->votes[(size_t)m * ORC_VOTE_FLOATS], &o->votes[(size_t)i * ORC_VOTE_FLOATS], ORC_VOTE_FLOATS * 4);
        }
        o->tm[m * 2 + 1] = lastT;
        m++;
    }
    free(keepf); free(lastTs);
    for (int k = 0; k < o->n_new && m < o->cap; k++) {
        const orc_meas* s = &o->newbuf[k];
        float lastT = -2;
        int keep = clean_test(o, tinv, time, s->pc, s->nr, (float)time, &lastT);
        if (!keep) continue;
        memcpy(&o->pc[m * 4], s->pc, 16);
        memcpy(&o->nr[m * 4], s->nr, 16);
        o->col[m * 2] = s->col0; o->col[m * 2 + 1] = 0;
        o->tm[m * 2] = (float)time; o->tm[m * 2 + 1] = lastT;
        memcpy(&o->ic[m * 4], s->ic, 16);
        memset(&o->votes[(size_t)m * ORC_VOTE_FLOATS], 0, ORC_VOTE_FLOATS * 4);
        m++;
    }
    o->n = m;
    o->n_new = 0;
    if (o->graph_nodes > 0) orc_set_deformation(o, NULL, 0, 0);   /* rawGraph lives for one frame (EF/ElasticFusion.cpp:482) */
}

/* IndexMap::renderSurfelIds, EF/IndexMap.cpp:315-465 + surfel_ids.vert/.geom/.frag and
 * instance_surfel_ids.vert:44-70.  Disc coverage by ray-disc intersection (DESIGN.md rule) in
 * place of the screen-space quad + unit-disc discard.  Result in o->ids_tmp, 0 = empty. */
void orc_render_ids(orc_t* o, const float* pose, int mode)
{
    int w = o->w, h = o->h;
    float tinv[16];
    orc_pose_inverse(pose, tinv);
    float cx = o->cfg.cx, cy = o->cfg.cy, fx = o->cfg.fx, fy = o->cfg.fy;
    float maxDepth = o->cfg.max_depth_processed, thr = o->cfg.confidence;
    zbufs_t zb = zb_open((size_t)o->P);
#pragma omp parallel num_threads(zb.T)
    {
        ZB_FOR_RANGES(t, zb.T) {
        long lo, hi;
        zb_range(o->n, zb.T, t, &lo, &hi);
        float* tz = zb.z + (size_t)t * zb.P; int* ti = zb.id + (size_t)t * zb.P;
        for (long i = lo; i < hi; i++) {
            if (!(o->pc[i * 4 + 3] > thr)) continue;
            if (mode == 1) {
                const float* v = &o->votes[(size_t)i * ORC_VOTE_FLOATS];
                int alleq = 1;
                for (int k = 4; k < ORC_VOTE_FLOATS; k++) if (v[k] != v[k & 3]) { alleq = 0; break; }
                if (alleq) continue;
            }
            v3 q = xf_point(tinv, v3m(o->pc[i * 4], o->pc[i * 4 + 1], o->pc[i * 4 + 2]));
            if (!(q.z / maxDepth > 0.01f)) continue;
            v3 n = v3normalized(xf_dir(tinv, v3m(o->nr[i * 4], o->nr[i * 4 + 1], o->nr[i * 4 + 2])));
            float r = o->nr[i * 4 + 3];
            float xs[2], ys[2], minz;
            disc_extent(q, n, r, cx, cy, fx, fy, xs, ys, &minz);
            if (!(minz > 0) || !(xs[0] == xs[0]) || !(ys[0] == ys[0])) continue;
            if (xs[1] - xs[0] > ORC_MAX_SPRITE || ys[1] - ys[0] > ORC_MAX_SPRITE) continue;
            if (xs[1] < 0 || ys[1] < 0 || xs[0] > (float)w || ys[0] > (float)h) continue;
            int x0 = clampi((int)ceilf(xs[0] - 0.5f), 0, w - 1), x1 = clampi((int)floorf(xs[1] - 0.5f), 0, w - 1);
            int y0 = clampi((int)ceilf(ys[0] - 0.5f), 0, h - 1), y1 = clampi((int)floorf(ys[1] - 0.5f), 0, h - 1);
            disc_t d = {q, n, r * r};
            for (int py = y0; py <= y1; py++)
                for (int px = x0; px <= x1; px++) {
                    float z;
                    if (!disc_hit(&d, (float)px + 0.5f, (float)py + 0.5f, cx, cy, fx, fy, &z)) continue;
                    if (!(z > 0 && z <= maxDepth)) continue;
                    int k = py * w + px;
                    if (z < tz[k]) { tz[k] = z; ti[k] = (int)i; }
                }
        }
        }
    }
    int* win = (int*)malloc((size_t)o->P * sizeof(int));
    zb_merge(&zb, o->zbuf, win);
    for (int k = 0; k < o->P; k++) o->ids_tmp[k] = win[k] < 0 ? 0 : win[k];
    free(win);
}

/* rodrigues2, EF/ElasticFusion.cpp:1183-1228.  The SVD re-orthonormalisation of the (already
 * orthonormal to rounding) relative rotation is omitted. */
static void rodrigues2(const float* R, float* out3)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = ((double)(R[0] + R[4] + R[8]) - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0) rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0.0);
            t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    out3[0] = (float)rx; out3[1] = (float)ry; out3[2] = (float)rz;
}

/* Stage entry used by the tests and the bench after a map upload: the combined prediction at an explicit pose FOLLOWED by the fill-in
 * (EF/ElasticFusion.cpp:729-763 does both: the tracker takes the fill-in images whenever the prediction is not dense enough).  The HIP
 * library's ifx_combined_predict resolves the fill-in in the same kernel; without it here the two trackers would start from
 * different models on sparse maps. */
static void fill_in(orc_t* o);
void orc_stage_predict(orc_t* o, const float* pose, int time, int max_time)
{
    orc_combined_predict(o, pose, time, max_time);
    fill_in(o);
}

/* ElasticFusion::predict, EF/ElasticFusion.cpp:729-763 */
static void predict(orc_t* o)
{
    orc_combined_predict(o, o->pose, o->tick, o->tick);
    fill_in(o);
}

void orc_set_loop_closure(orc_t* o, int enable, int count_thresh, float err_thresh, float cov_thresh)
{
    o->lc_enable = enable; o->lc_count_thresh = count_thresh; o->lc_err_thresh = err_thresh; o->lc_cov_thresh = cov_thresh;
    if (enable && !o->m2m) {
        size_t P = (size_t)o->P;
        o->old_vertex = (float*)calloc(P * 4, 4); o->old_normal = (float*)calloc(P * 4, 4);
        o->old_image = (uint8_t*)calloc(P * 4, 1); o->old_inst = (uint8_t*)calloc(P * 4, 1); o->old_time = (uint16_t*)calloc(P, 2);
        o->m2m = orc_tracker_create(o->w, o->h, o->cfg.fx, o->cfg.fy, o->cfg.cx, o->cfg.cy);
    }
}
void orc_loop_closure_diag(orc_t* o, float* out24) { memcpy(out24, o->lc, sizeof(o->lc)); }

/* EF/ElasticFusion.cpp:453-566 with ferns.lastClosest == -1 (no fern data base): predict(); INACTIVE prediction; model-to-model
 * tracking; the three gates.  An accepted candidate is only reported (the deformation graph is not restated). */
static void loop_closure_local(orc_t* o)
{
    predict(o);                                                         /* :453 */
    if (o->fern_cb && o->fern_cb(o, o->fern_user) > 0) {                /* :457-514; matched to a fern and deformed: :516 skips the rest */
        const float cand = o->lc[23];
        memset(o->lc, 0, sizeof(o->lc));
        o->lc[23] = cand;
        memcpy(&o->lc[6], o->pose, 64);
        return;
    }
    float *pv = o->pred_vertex, *pn = o->pred_normal;                   /* :519-526: same shader, other framebuffer */
    uint8_t *pi = o->pred_image, *ps = o->pred_inst;
    uint16_t* pt = o->pred_time;
    o->pred_vertex = o->old_vertex; o->pred_normal = o->old_normal; o->pred_image = o->old_image; o->pred_inst = o->old_inst; o->pred_time = o->old_time;
    orc_combined_predict(o, o->pose, 0, o->tick - o->cfg.time_delta);
    o->pred_vertex = pv; o->pred_normal = pn; o->pred_image = pi; o->pred_inst = ps; o->pred_time = pt;
    int n_old = 0;
    for (int k = 0; k < o->P; k++) n_old += (o->old_vertex[(size_t)k * 4 + 2] != 0);
    const float cand = o->lc[23];
    memset(o->lc, 0, sizeof(o->lc));
    o->lc[1] = (float)n_old; o->lc[23] = cand;
    memcpy(&o->lc[6], o->pose, 64);
    if (n_old == 0) return;   /* every reduction would be empty: count 0 fails the gate whatever the rest computes */
    orc_tracker_init_model(o->m2m, o->old_vertex, o->old_normal, o->old_image, o->pose);      /* :530-531 */
    orc_tracker_init_frame_maps(o->m2m, o->pred_vertex, o->pred_normal, o->pred_image);        /* :533-534 */
    float est[16], diag[8];
    memcpy(est, o->pose, 64);
    orc_tracker_run(o->m2m, est, 10.0f, o->cfg.pyramid, o->cfg.fast_odom, 0, diag);          /* :539-545 */
    double cov[36];
    orc_tracker_covariance(o->m2m, cov);
    int cov_ok = 1;
    double cmax = 0;
    for (int i = 0; i < 6; i++) {
        if (cov[i * 6 + i] > o->lc_cov_thresh) { cov_ok = 0; }
        if (!(cov[i * 6 + i] <= cmax)) cmax = cov[i * 6 + i];
    }
    const int accept = cov_ok && diag[1] > (float)o->lc_count_thresh && diag[0] < o->lc_err_thresh;   /* :566 */
    o->lc[0] = 1; o->lc[2] = diag[0]; o->lc[3] = diag[1]; o->lc[4] = (float)cov_ok; o->lc[5] = (float)accept;
    memcpy(&o->lc[6], est, 64);
    o->lc[22] = (float)cmax;
    o->lc_candidates += accept;
    o->lc[23] = (float)o->lc_candidates;
    if (accept && o->lc_cb) o->lc_cb(o, o->lc, o->lc_user);   /* :566-613: constraints, graph optimisation (caller), rawGraph, currPose = estPose */
}
void orc_set_loop_closure_callback(orc_t* o, orc_lc_callback cb, void* user) { o->lc_cb = cb; o->lc_user = user; }
void orc_set_fern_callback(orc_t* o, orc_fern_callback cb, void* user) { o->fern_cb = cb; o->fern_user = user; }
void orc_adopt_pose(orc_t* o, const float* pose16) { memcpy(o->pose, pose16, 64); }
/* Resize::image / Resize::vertex (EF/Shaders/Resize.cpp, resize.frag: nearest sample at the centre of each target texel) */
int orc_fern_frame(orc_t* o, uint8_t* img_rgb, float* verts4, float* norms4, uint8_t* inst_rgb)
{
    const int rw = o->w / 8, rh = o->h / 8;
    for (int j = 0; j < rh; j++)
        for (int i = 0; i < rw; i++) {
            const int t = j * rw + i, k = ((j * o->h + o->h / 2) / rh) * o->w + (i * o->w + o->w / 2) / rw;
            memcpy(&img_rgb[t * 3], &o->fill_image[k * 4], 3);
            memcpy(&inst_rgb[t * 3], &o->pred_inst[k * 4], 3);
            memcpy(&verts4[t * 4], &o->fill_vertex[k * 4], 16);
            memcpy(&norms4[t * 4], &o->fill_normal[k * 4], 16);
        }
    return rw * rh;
}

/* ElasticFusion::processFrame, EF/ElasticFusion.cpp:269-720.  Of the loop-closure block (:450-617) the local detection is
 * restated (loop_closure_local, when enabled) with the two places where the caller's host code runs (fern lookup, deformation:
 * orc_set_fern_callback / orc_set_loop_closure_callback; the fern data base itself is restated in orc_ferns.py); without the
 * detection the predict() of :453, whose only consumers are those stages, is not executed. */
int orc_process_frame(orc_t* o, const uint8_t* rgb, const uint16_t* depth, int64_t ts,
                      const float* in_pose16, float weight_mult, float* out_pose16)
{
    (void)ts;
    int P = o->P;
    const double t_begin = orc_now_ms();
    double t_tracked = t_begin;
    memcpy(o->rgb, rgb, (size_t)P * 3);
    memcpy(o->depth_raw, depth, (size_t)P * 2);
    orc_bilateral(o->depth_raw, o->depth_filt, o->w, o->h, o->cfg.depth_cut);
    orc_metric(o->depth_raw, o->dm, o->w, o->h, o->cfg.depth_cut);
    orc_metric(o->depth_filt, o->dmf, o->w, o->h, o->cfg.depth_cut);
    if (o->tick == 1) {
        init_first_frame(o);
        orc_tracker_init_first_rgb(o->trk, rgb);
    } else {
        float lastPose[16];
        memcpy(lastPose, o->pose, 64);
        const int boot = o->bootstrap_next && in_pose16;
        o->bootstrap_next = 0;
        if (!in_pose16 || boot) {
            int fill = !dense_enough(o);
            orc_tracker_init_model(o->trk, fill ? o->fill_vertex : o->pred_vertex, fill ? o->fill_normal : o->pred_normal,
                                   fill ? o->fill_image : o->pred_image, o->pose);
            orc_tracker_init_frame(o->trk, o->depth_filt, rgb, o->cfg.max_depth_processed);
            if (boot) {   /* EF/ElasticFusion.cpp:352-356: currPose = currPose * inPose after the model maps were placed */
                float g[16];
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        float s = 0;
                        for (int k = 0; k < 4; k++) s += o->pose[r * 4 + k] * in_pose16[k * 4 + c];
                        g[r * 4 + c] = s;
                    }
                memcpy(o->pose, g, 64);
            }
            orc_tracker_run(o->trk, o->pose, o->cfg.icp_weight, o->cfg.pyramid, o->cfg.fast_odom, o->cfg.so3, o->diag);
        } else memcpy(o->pose, in_pose16, 64);
        /* velocity weighting :433-449 */
        float inv[16], diff[16];
        orc_pose_inverse(o->pose, inv);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float s = 0;
                for (int k = 0; k < 4; k++) s += inv[r * 4 + k] * lastPose[k * 4 + c];
                diff[r * 4 + c] = s;
            }
        float R3[9] = {diff[0], diff[1], diff[2], diff[4], diff[5], diff[6], diff[8], diff[9], diff[10]};
        float rv[3];
        rodrigues2(R3, rv);
        float tn = sqrtf(diff[3] * diff[3] + diff[7] * diff[7] + diff[11] * diff[11]);
        float rn = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
        float weighting = fmaxf(tn, rn);
        const float largest = 0.01f, minWeight = 0.5f;
        if (weighting > largest) weighting = largest;
        weighting = fmaxf(1.0f - (weighting / largest), minWeight) * weight_mult;
        o->last_weighting = weighting;
        t_tracked = orc_now_ms();
        if (o->lc_enable) loop_closure_local(o);

        orc_predict_indices(o, o->pose, o->tick);
        orc_fuse(o, o->pose, o->tick, weighting);
        orc_predict_indices(o, o->pose, o->tick);
        orc_clean(o, o->pose, o->tick);
        orc_render_ids(o, o->pose, 0);
        memcpy(o->ids_after, o->ids_tmp, (size_t)P * 4);
    }
    predict(o);
    {
        const double t_end = orc_now_ms();
        o->stage_ms[0] += t_tracked - t_begin;
        o->stage_ms[1] += t_end - t_tracked;
    }
    if (out_pose16) memcpy(out_pose16, o->pose, 64);
    o->tick++;
    return 0;
}
