/* oracle/orc_slic.c -- superpixel refinement of the masks.
 * TEST INFRASTRUCTURE ONLY (see orc.h): CPU restatement of
 *   IF/Core/InstanceFusion_superpixel.cpp:713-772   gSLICrInterface   (-> orc_slic_segment)
 *   IF/gSLICr/gSLICr_Lib/engines/gSLICr_seg_engine.cpp + gSLICr_seg_engine_GPU.cu + gSLICr_seg_engine_shared.h
 *   IF/Core/InstanceFusion_superpixel.cpp:40-225    mergeSuperPixel   (-> orc_merge_superpixels)
 *   IF/Core/InstanceFusionCuda.cu:41-61,141-180,180-732  depth gaussian, pos/normal maps, getSuperPixelInfo kernels 0,A-E
 *   IF/Core/InstanceFusion_superpixel.cpp:227-400   connectSuperPixel
 *   IF/Core/InstanceFusion_superpixel.cpp:651-710   maskSuperPixelFilter_OverSeg (-> orc_mask_superpixel_filter)
 *
 * SLIC: the per-pixel maths and the summation ORDER of the cluster update (16x16 blocks, tree of
 * strides 128..1, then the 9 blocks of a centre in sequence) are the reference's, so this file is
 * bit-identical to a build of the reference's own gSLICr_seg_engine_shared.h (oracle/_ref, see
 * oracle/ref_slic.cpp and tests/test_oracle_slic.py).
 *
 * Superpixel statistics: the reference accumulates with float atomicAdd (order = hardware
 * scheduling) and fills the neighbour lists with an unsynchronised read-modify-write; both are
 * made deterministic here and in the HIP path:
 *   - sums are exact 64-bit fixed point (2^-32 units), converted to float once;
 *   - a superpixel's neighbour list is the set of its distinct 4-neighbours in ascending id order,
 *     truncated to 11 entries, padded with -1;
 *   - non-finite contributions are dropped from the sums.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"
#include "orc_internal.h"

#define SPX 16 /* my_settings.spixel_size, IF/Core/InstanceFusion.cpp:446; BLOCK_DIM of gSLICr is 16 as well */

/* ---------------------------------------------------------------- SLIC */
typedef struct { float cx, cy, col[3]; int id, n; } sp_center;

/* rgb2xyz, gSLICr_seg_engine_shared.h:10-19.  The channel shuffle of gSLICrInterface + imageCV2SLIC
 * (BGR2RGB swap, then .b=[0] .g=[1] .r=[2]) leaves pix.x = c0, pix.z = c2, so "_b" is channel 0. */
static void slic_cvt(const uint8_t* rgb, float* xyz, int P)
{
    for (int i = 0; i < P; i++) {
        float b = (float)rgb[i * 3] * 0.0039216f, g = (float)rgb[i * 3 + 1] * 0.0039216f, r = (float)rgb[i * 3 + 2] * 0.0039216f;
        xyz[i * 3 + 0] = r * 0.412453f + g * 0.357580f + b * 0.180423f;
        xyz[i * 3 + 1] = r * 0.212671f + g * 0.715160f + b * 0.072169f;
        xyz[i * 3 + 2] = r * 0.019334f + g * 0.119193f + b * 0.950227f;
    }
}

/* find_center_association_shared + compute_slic_distance, gSLICr_seg_engine_shared.h:84-124 */
static void slic_assoc(const float* xyz, const sp_center* c, int* seg, int mw, int mh, int w, int h, float weight, float nxy, float ncol)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int gx = x / SPX, gy = y / SPX, minidx = -1;
            float dist = 999999.9999f;
            const float* p = &xyz[(y * w + x) * 3];
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++) {
                    int qx = gx + j, qy = gy + i;
                    if (qx < 0 || qy < 0 || qx >= mw || qy >= mh) continue;
                    const sp_center* q = &c[qy * mw + qx];
                    float dcol = (p[0] - q->col[0]) * (p[0] - q->col[0]) + (p[1] - q->col[1]) * (p[1] - q->col[1]) + (p[2] - q->col[2]) * (p[2] - q->col[2]);
                    float dxy = ((float)x - q->cx) * ((float)x - q->cx) + ((float)y - q->cy) * ((float)y - q->cy);
                    float d = sqrtf(dcol * ncol + weight * dxy * nxy);
                    if (d < dist) { dist = d; minidx = q->id; }
                }
            if (minidx >= 0) seg[y * w + x] = minidx;
        }
}

/* Update_Cluster_Center_device + finalize_reduction_result_shared, gSLICr_seg_engine_GPU.cu:203-290,
 * gSLICr_seg_engine_shared.h:143-166: 9 blocks of 16x16 window pixels per centre, each reduced by
 * the stride-128..1 tree, the 9 partials added in block order. */
static void slic_update(const float* xyz, const int* seg, sp_center* c, int mw, int mh, int w, int h)
{
    float s[5][256];
    int cnt[256];
    for (int gy = 0; gy < mh; gy++)
        for (int gx = 0; gx < mw; gx++) {
            int id = gy * mw + gx, n = 0;
            float acc[5] = {0, 0, 0, 0, 0};
            for (int z = 0; z < 9; z++) {
                int bx = z % 3, by = z / 3;
                for (int t = 0; t < 256; t++) {
                    int xi = gx * SPX - SPX + bx * 16 + (t & 15), yi = gy * SPX - SPX + by * 16 + (t >> 4);
                    int hit = xi >= 0 && xi < w && yi >= 0 && yi < h && seg[yi * w + xi] == id;
                    cnt[t] = hit;
                    s[0][t] = hit ? xyz[(yi * w + xi) * 3] : 0.f;
                    s[1][t] = hit ? xyz[(yi * w + xi) * 3 + 1] : 0.f;
                    s[2][t] = hit ? xyz[(yi * w + xi) * 3 + 2] : 0.f;
                    s[3][t] = hit ? (float)xi : 0.f;
                    s[4][t] = hit ? (float)yi : 0.f;
                }
                for (int st = 128; st >= 1; st >>= 1)
                    for (int t = 0; t < st; t++) {
                        for (int k = 0; k < 5; k++) s[k][t] += s[k][t + st];
                        cnt[t] += cnt[t + st];
                    }
                for (int k = 0; k < 5; k++) acc[k] += s[k][0];
                n += cnt[0];
            }
            sp_center* q = &c[id];
            q->n = n;
            if (n != 0) {
                q->cx = acc[3] / (float)n; q->cy = acc[4] / (float)n;
                q->col[0] = acc[0] / (float)n; q->col[1] = acc[1] / (float)n; q->col[2] = acc[2] / (float)n;
            } else {
                q->cx = q->cy = q->col[0] = q->col[1] = q->col[2] = 0.f;
            }
        }
}

/* supress_local_lable, gSLICr_seg_engine_shared.h:168-195 */
static void slic_enforce(const int* in, int* out, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int cl = in[y * w + x];
            if (x <= 1 || y <= 1 || x >= w - 2 || y >= h - 2) { out[y * w + x] = cl; continue; }
            int diff = 0, dl = -1;
            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++) {
                    int nl = in[(y + j) * w + x + i];
                    if (nl != cl) { dl = nl; diff++; }
                }
            out[y * w + x] = diff >= 16 ? dl : cl;
        }
}

/* gSLICrInterface -> core_engine::Process_Frame -> seg_engine::Perform_Segmentation
 * (gSLICr_seg_engine.cpp:33-50) with the settings of IF/Core/InstanceFusion.cpp:441-451:
 * spixel_size 16, coh_weight 0.6, 5 iterations, XYZ, enforce connectivity. */
int orc_slic_segment(orc_t* o, const uint8_t* rgb, int32_t* seg)
{
    int w = o->w, h = o->h, P = w * h, mw = w / SPX, mh = h / SPX;
    if (mw < 1 || mh < 1) return -1;
    float* xyz = (float*)malloc((size_t)P * 3 * sizeof(float));
    sp_center* c = (sp_center*)calloc((size_t)mw * mh, sizeof(sp_center));
    int* tmp = (int*)malloc((size_t)P * sizeof(int));
    slic_cvt(rgb, xyz, P);
    memset(seg, 0, (size_t)P * sizeof(int)); /* ORUtils images start zeroed; pixels right of the last full cell always find a centre */
    /* init_cluster_centers_shared :71-82 */
    for (int y = 0; y < mh; y++)
        for (int x = 0; x < mw; x++) {
            int ix = x * SPX + SPX / 2, iy = y * SPX + SPX / 2;
            ix = ix >= w ? (x * SPX + w) / 2 : ix;
            iy = iy >= h ? (y * SPX + h) / 2 : iy;
            sp_center* q = &c[y * mw + x];
            q->id = y * mw + x; q->cx = (float)ix; q->cy = (float)iy; q->n = 0;
            memcpy(q->col, &xyz[(iy * w + ix) * 3], 3 * sizeof(float));
        }
    float nxy = 1.0f / (1.4242f * SPX), ncol = 5.0f / 1.7321f; /* seg_engine_GPU ctor :40-56 */
    ncol *= ncol; nxy *= nxy;
    slic_assoc(xyz, c, seg, mw, mh, w, h, 0.6f, nxy, ncol);
    for (int it = 0; it < 5; it++) {
        slic_update(xyz, seg, c, mw, mh, w, h);
        slic_assoc(xyz, c, seg, mw, mh, w, h, 0.6f, nxy, ncol);
    }
    slic_enforce(seg, tmp, w, h);
    slic_enforce(tmp, seg, w, h);
    free(xyz); free(c); free(tmp);
    return mw * mh;
}

/* ---------------------------------------------------------------- merge */
enum { SPI_SIZE = 30, SPI_PNUM = 0, SPI_POS_S = 1, SPI_NOR_S = 4, SPI_POS_A = 7, SPI_NOR_A = 10, SPI_DEPTH_SUM = 13, SPI_DEPTH_AVG = 14,
       SPI_DIST_DEV = 15, SPI_NOR_DEV = 16, SPI_CONNECT_N = 17, SPI_NP_FIRST = 18, SPI_NP_MAX = 11, SPI_FINAL = 29 };

static inline int64_t sp_fx(float v)
{
    if (!(fabsf(v) < 1.0e6f)) return 0;
    return (int64_t)llrint((double)v * 4294967296.0);
}
static inline float sp_unfx(int64_t s) { return (float)((double)s * (1.0 / 4294967296.0)); }

/* checkNeighbours, IF/Core/InstanceFusionCuda.cu:41-61 (the centre itself is not tested) */
static int check_nb(const uint16_t* m, int x, int y, int w, int h)
{
    if (x + 1 >= w || x - 1 < 0 || y + 1 >= h || y - 1 < 0) return 0;
    return m[y * w + x + 1] && m[y * w + x - 1] && m[(y + 1) * w + x] && m[(y - 1) * w + x] && m[(y + 1) * w + x + 1] && m[(y + 1) * w + x - 1] &&
           m[(y - 1) * w + x + 1] && m[(y - 1) * w + x - 1];
}

/* getVertex :180-187: integer pixel coordinates, depth / 1186 */
static void sp_vertex(const uint16_t* d, int x, int y, int w, const float* cam, float* v)
{
    float z = (float)d[y * w + x] / 1186.0f;
    v[0] = ((float)x - cam[0]) * z * cam[2];
    v[1] = ((float)y - cam[1]) * z * cam[3];
    v[2] = z;
}
static void sp_cross(const float* l, const float* r, const float* u, const float* dn, float* a)
{
    float dx[3] = {l[0] - r[0], l[1] - r[1], l[2] - r[2]}, dy[3] = {u[0] - dn[0], u[1] - dn[1], u[2] - dn[2]};
    a[0] = dx[1] * dy[2] - dx[2] * dy[1];
    a[1] = dx[2] * dy[0] - dx[0] * dy[2];
    a[2] = dx[0] * dy[1] - dx[1] * dy[0];
}
/* getNormal :205-248: 4x the central-difference cross + 2x each of the four one-sided ones */
static void sp_normal(const uint16_t* d, int x, int y, int w, const float* cam, float* n)
{
    float c[3], xf[3], xb[3], yf[3], yb[3], t[3], s[3];
    sp_vertex(d, x, y, w, cam, c);
    sp_vertex(d, x + 1, y, w, cam, xf);
    sp_vertex(d, x - 1, y, w, cam, xb);
    sp_vertex(d, x, y + 1, w, cam, yf);
    sp_vertex(d, x, y - 1, w, cam, yb);
    sp_cross(xb, xf, yb, yf, t);
    for (int k = 0; k < 3; k++) s[k] = t[k] * 4;
    sp_cross(xb, c, yb, c, t);
    for (int k = 0; k < 3; k++) s[k] += t[k] * 2;
    sp_cross(c, xf, yb, c, t);
    for (int k = 0; k < 3; k++) s[k] += t[k] * 2;
    sp_cross(xb, c, c, yf, t);
    for (int k = 0; k < 3; k++) s[k] += t[k] * 2;
    sp_cross(c, xf, c, yf, t);
    for (int k = 0; k < 3; k++) s[k] += t[k] * 2;
    float len = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (int k = 0; k < 3; k++) n[k] = s[k] / len;
}

typedef struct { int64_t n, pos[3], nor[3], depth, dist, ndev; } sp_sums;

static void sp_averages(const sp_sums* s, float* info)
{
    int t = (int)s->n;
    info[SPI_PNUM] = (float)t;
    for (int k = 0; k < 3; k++) { info[SPI_POS_S + k] = sp_unfx(s->pos[k]); info[SPI_NOR_S + k] = sp_unfx(s->nor[k]); }
    info[SPI_DEPTH_SUM] = (float)s->depth;
    if (t != 0) { /* kernels C / E :492-516, :612-640 */
        for (int k = 0; k < 3; k++) info[SPI_POS_A + k] = info[SPI_POS_S + k] / (float)t;
        float nx = info[SPI_NOR_S], ny = info[SPI_NOR_S + 1], nz = info[SPI_NOR_S + 2];
        float len = sqrtf(nx * nx + ny * ny + nz * nz);
        for (int k = 0; k < 3; k++) info[SPI_NOR_A + k] = info[SPI_NOR_S + k] / len;
        info[SPI_DEPTH_AVG] = info[SPI_DEPTH_SUM] / (float)t;
    }
}

/* connectSuperPixel, IF/Core/InstanceFusion_superpixel.cpp:227-400 (double where the reference's
 * literals promote to double) */
static void sp_connect(int spn, float* info)
{
    for (int i = 0; i < spn; i++) {
        float* A = &info[i * SPI_SIZE];
        A[SPI_FINAL] = -1;
        int cn = (int)A[SPI_CONNECT_N];
        for (int j = 0; j < cn; j++) {
            int ib = (int)A[SPI_NP_FIRST + j];
            if (ib == -1) continue;
            float* B = &info[ib * SPI_SIZE];
            float va[3] = {A[SPI_NOR_A], A[SPI_NOR_A + 1], A[SPI_NOR_A + 2]};
            float vb[3] = {A[SPI_POS_A] - B[SPI_POS_A], A[SPI_POS_A + 1] - B[SPI_POS_A + 1], A[SPI_POS_A + 2] - B[SPI_POS_A + 2]};
            float lenA = sqrtf(va[0] * va[0] + va[1] * va[1] + va[2] * va[2]);
            float lenB = sqrtf(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
            float dot = va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2];
            float distTerm = (float)((double)fabsf(dot / lenA) + 1.0 * (double)lenB);
            float tA1 = (float)(1 * ((0.026 * (double)A[SPI_DEPTH_AVG] - (double)4.0f) / (double)1186.0f));
            float tB1 = (float)(1 * ((0.026 * (double)B[SPI_DEPTH_AVG] - (double)4.0f) / (double)1186.0f));
            float tA2 = 2 * A[SPI_DIST_DEV], tB2 = 2 * B[SPI_DIST_DEV];
            float d1 = fabsf(A[SPI_NOR_A] - B[SPI_NOR_A]), d2 = fabsf(A[SPI_NOR_A + 1] - B[SPI_NOR_A + 1]), d3 = fabsf(A[SPI_NOR_A + 2] - B[SPI_NOR_A + 2]);
            float norTerm = (float)(0.1 * (double)sqrtf(d1 * d1 + d2 * d2 + d3 * d3));
            float tA3 = 0 * A[SPI_NOR_DEV], tB3 = 0 * B[SPI_NOR_DEV];
            float fin = distTerm + norTerm, TA = tA1 + tA2 + tA3, TB = tB1 + tB2 + tB3;
            if (fin > TA || fin > TB) {
                A[SPI_NP_FIRST + j] = -1;
                int cnb = (int)B[SPI_CONNECT_N];
                for (int k = 0; k < cnb; k++)
                    if (B[SPI_NP_FIRST + k] == (float)i) { B[SPI_NP_FIRST + k] = -1; break; }
            }
        }
    }
    int* stack = (int*)malloc((size_t)spn * (SPI_NP_MAX + 1) * sizeof(int));
    int p = 0;
    for (int i = 0; i < spn; i++) {
        int fid = info[i * SPI_SIZE + SPI_FINAL] == -1 ? i : (int)info[i * SPI_SIZE + SPI_FINAL];
        stack[p++] = i;
        while (p > 0) {
            int tg = stack[--p];
            if (info[tg * SPI_SIZE + SPI_FINAL] != -1) continue;
            info[tg * SPI_SIZE + SPI_FINAL] = (float)fid;
            int cn = (int)info[tg * SPI_SIZE + SPI_CONNECT_N];
            for (int j = 0; j < cn; j++) {
                int q = (int)info[tg * SPI_SIZE + SPI_NP_FIRST + j];
                if (q == -1 || info[q * SPI_SIZE + SPI_FINAL] != -1) continue;
                stack[p++] = q;
            }
        }
    }
    free(stack);
}

/* mergeSuperPixel.  seg: in = SLIC labels, out = re-clustered labels (-1 = invalid), as the reference
 * copies back at :118.  final_out: merged region id per pixel.  info_out: spn*30 floats (or NULL). */
int orc_merge_superpixels(orc_t* o, const uint16_t* depth, int32_t* seg, int32_t* final_out, float* info_out)
{
    int w = o->w, h = o->h, P = w * h, spn = P / (SPX * SPX);
    float cam[4] = {o->cfg.cx, o->cfg.cy, (float)(1.0 / (double)o->cfg.fx), (float)(1.0 / (double)o->cfg.fy)};
    uint16_t* dg = (uint16_t*)calloc((size_t)P, 2);
    float* pos = (float*)calloc((size_t)P * 3, 4);
    float* nor = (float*)calloc((size_t)P * 3, 4);
    float* info = (float*)calloc((size_t)spn * SPI_SIZE, 4);
    sp_sums* s1 = (sp_sums*)calloc((size_t)spn, sizeof(sp_sums));
    sp_sums* s2 = (sp_sums*)calloc((size_t)spn, sizeof(sp_sums));
    uint8_t* adj = (uint8_t*)calloc((size_t)spn * spn, 1);

    /* depthMapGaussianfilterKernel :141-168 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (!check_nb(depth, x, y, w, h)) continue;
            static const int wt[3][3] = {{1, 2, 1}, {2, 4, 2}, {1, 2, 1}};
            int sum = 0, n = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    int d = depth[(y + j) * w + x + i];
                    if (d) { n += wt[j + 1][i + 1]; sum += wt[j + 1][i + 1] * d; }
                }
            if (n) dg[y * w + x] = (uint16_t)(sum / n);
        }
    /* getPosMapFromDepthKernel :250-263, getNormalMapFromDepthKernel :276-290 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (dg[y * w + x]) sp_vertex(dg, x, y, w, cam, &pos[(y * w + x) * 3]);
            if (check_nb(dg, x, y, w, h)) sp_normal(dg, x, y, w, cam, &nor[(y * w + x) * 3]);
        }
    /* kernel 0 :304-318 */
    for (int i = 0; i < spn; i++) {
        info[i * SPI_SIZE + SPI_CONNECT_N] = SPI_NP_MAX;
        for (int k = 0; k < SPI_NP_MAX; k++) info[i * SPI_SIZE + SPI_NP_FIRST + k] = -1;
    }
    /* kernel A :320-340 */
    for (int k = 0; k < P; k++) {
        float t = 0;
        t += pos[k * 3] * pos[k * 3]; t += pos[k * 3 + 1] * pos[k * 3 + 1]; t += pos[k * 3 + 2] * pos[k * 3 + 2];
        t += nor[k * 3] * nor[k * 3]; t += nor[k * 3 + 1] * nor[k * 3 + 1]; t += nor[k * 3 + 2] * nor[k * 3 + 2];
        if ((double)t < 0.01 || seg[k] >= spn || seg[k] < 0) seg[k] = -1;
    }
    /* kernel B :342-400: sums + adjacency */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int k = y * w + x, id = seg[k];
            if (id < 0) continue;
            sp_sums* s = &s1[id];
            s->n++;
            for (int c = 0; c < 3; c++) { s->pos[c] += sp_fx(pos[k * 3 + c]); s->nor[c] += sp_fx(nor[k * 3 + c]); }
            s->depth += dg[k];
            if (x == 0 || x == w - 1 || y == 0 || y == h - 1) continue;
            const int nb[4] = {k + w, k - w, k + 1, k - 1};
            for (int i = 0; i < 4; i++) {
                int q = seg[nb[i]];
                if (q >= 0 && q != id) adj[(size_t)id * spn + q] = 1;
            }
        }
    for (int i = 0; i < spn; i++) {
        int c = 0;
        for (int q = 0; q < spn && c < SPI_NP_MAX; q++)
            if (adj[(size_t)i * spn + q]) info[i * SPI_SIZE + SPI_NP_FIRST + c++] = (float)q;
    }
    /* kernel C */
    for (int i = 0; i < spn; i++) sp_averages(&s1[i], &info[i * SPI_SIZE]);
    /* kernel D :518-610 */
    for (int k = 0; k < P; k++) {
        int id = seg[k];
        if (id < 0) continue;
        float minDist = 999999.9f, minNor = 999999.9f;
        int minID = id;
        for (int i = 0; i <= SPI_NP_MAX; i++) {
            int it = i == SPI_NP_MAX ? id : (int)info[id * SPI_SIZE + SPI_NP_FIRST + i];
            if (it < 0) continue; /* the reference never sees -1 here unless the superpixel has no neighbour at all (out-of-bounds read there) */
            const float* T = &info[it * SPI_SIZE];
            float va[3] = {T[SPI_NOR_A], T[SPI_NOR_A + 1], T[SPI_NOR_A + 2]};
            float vb[3] = {T[SPI_POS_A] - pos[k * 3], T[SPI_POS_A + 1] - pos[k * 3 + 1], T[SPI_POS_A + 2] - pos[k * 3 + 2]};
            float lenA = sqrtf(va[0] * va[0] + va[1] * va[1] + va[2] * va[2]);
            float lenB = sqrtf(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
            float dot = va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2];
            float dist = (float)((double)fabsf(dot / lenA) + 1.0 * (double)lenB);
            float d1 = fabsf(va[0] - nor[k * 3]), d2 = fabsf(va[1] - nor[k * 3 + 1]), d3 = fabsf(va[2] - nor[k * 3 + 2]);
            float dn = d1 * d1 + d2 * d2 + d3 * d3;
            if (dist < minDist) { minNor = dn; minDist = dist; minID = it; }
        }
        float thr = (float)((0.026 * (double)info[minID * SPI_SIZE + SPI_DEPTH_AVG] - (double)4.0f) / (double)1186.0f);
        if (minDist > 2 * thr) minID = -1;
        seg[k] = minID;
        if (minID != -1) {
            sp_sums* s = &s2[minID];
            s->n++;
            for (int c = 0; c < 3; c++) { s->pos[c] += sp_fx(pos[k * 3 + c]); s->nor[c] += sp_fx(nor[k * 3 + c]); }
            s->depth += dg[k];
            s->dist += sp_fx(minDist * minDist);
            s->ndev += sp_fx(minNor);
        }
    }
    /* kernel E :612-640: a superpixel left without pixels keeps its first-pass averages */
    for (int i = 0; i < spn; i++) {
        float* I = &info[i * SPI_SIZE];
        sp_averages(&s2[i], I);
        I[SPI_DIST_DEV] = sp_unfx(s2[i].dist);
        I[SPI_NOR_DEV] = sp_unfx(s2[i].ndev);
        if (s2[i].n != 0) {
            I[SPI_DIST_DEV] = sqrtf(I[SPI_DIST_DEV] / (float)(int)s2[i].n);
            I[SPI_NOR_DEV] = sqrtf(I[SPI_NOR_DEV] / (float)(int)s2[i].n);
        }
    }
    sp_connect(spn, info);
    /* getFinalSuperPiexlKernel :690-705 */
    for (int k = 0; k < P; k++) final_out[k] = seg[k] < 0 ? seg[k] : (int)info[seg[k] * SPI_SIZE + SPI_FINAL];
    if (info_out) memcpy(info_out, info, (size_t)spn * SPI_SIZE * 4);
    free(dg); free(pos); free(nor); free(info); free(s1); free(s2); free(adj);
    return spn;
}

/* maskSuperPixelFilter_OverSeg, IF/Core/InstanceFusion_superpixel.cpp:651-710 */
void orc_mask_superpixel_filter(orc_t* o, const int32_t* fin, uint8_t* masks, int nm)
{
    int P = o->P, spn = P / (SPX * SPX);
    int* num = (int*)calloc((size_t)(nm + 1) * spn, sizeof(int));
    for (int k = 0; k < P; k++) {
        int id = fin[k];
        if (id >= spn || id < 0) continue;
        num[nm * spn + id]++;
        for (int i = 0; i < nm; i++) if (masks[(size_t)i * P + k]) num[i * spn + id]++;
    }
    for (int k = 0; k < P; k++) {
        int id = fin[k];
        for (int i = 0; i < nm; i++) {
            if (id >= spn || id < 0) { masks[(size_t)i * P + k] = 0; continue; }
            float test = (float)num[i * spn + id] * 1.0f / (float)num[nm * spn + id];
            masks[(size_t)i * P + k] = (double)test > 0.75 ? 255 : 0;
        }
    }
    free(num);
}

/* steps -1_1 .. -1_3 of processInstance, IF/Core/InstanceFusion.cpp:722-738 */
int orc_superpixel_refine(orc_t* o, const uint8_t* rgb, const uint16_t* depth, uint8_t* masks, int nm, int frame)
{
    (void)frame;
    int P = o->P;
    int32_t* seg = (int32_t*)malloc((size_t)P * 4);
    int32_t* fin = (int32_t*)malloc((size_t)P * 4);
    int r = orc_slic_segment(o, rgb, seg);
    if (r > 0) {
        orc_merge_superpixels(o, depth, seg, fin, NULL);
        orc_mask_superpixel_filter(o, fin, masks, nm);
    }
    free(seg); free(fin);
    return r > 0 ? 0 : -1;
}
