/*
 * oracle/orc_track.c -- CPU restatement of the tracking half of the path.
 * TEST INFRASTRUCTURE ONLY (see orc.h).  Every function cites the reference lines it restates.
 */
#include "orc.h"
#include "orc_math.h"
#include <stdlib.h>
#include <stdio.h>

/* test hook (orc_set_sum_order): 0 = rows added top to bottom, 1 = bottom to top, 2 = every row partial rounded to f32 first (the precision
 * of the HIP path's per-block partial rows), 3 = the reference's own f32 tree with its GTX 1080 launch table (ref_tree_sum below).  Same arithmetic, another summation order / rounding: used to
 * measure how far two equally valid executions of the algorithm drift apart (tests/test_oracle_cpu.py). */
int orc_sum_reverse = 0;
void orc_set_sum_order(int reverse) { orc_sum_reverse = reverse; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ======================================================================= preprocessing (a2) */

/* The texel a tap of depth_bilateral.frag:59-61 reads: the shader samples at texture(gSampler, vec2(float(cx) / cols, float(cy) / rows)) -- the CORNER of texel (cx, cy), not
 * its centre -- with GL_NEAREST, i.e. texel floor(u * size) (GL 4.5 section 8.14.2): cx itself, unless the f32 quotient times the size falls just below cx, in which case
 * the tap reads texel cx - 1.  At 640 columns no column does; at 480 rows seven rows do (63, 125, 126, 127, 250, 252, 254).  Pinned by executing the reference's shader
 * (tests/golden/gl_map_passes.npz); rounds 1-5 took texel cx for every tap. */
static inline int bilateral_tap(int c, int n)
{
    int t = (int)floorf(((float)c / (float)n) * (float)n);
    return t < 0 ? 0 : (t > n - 1 ? n - 1 : t);
}
int orc_test_bilateral_tap(int c, int n) { return bilateral_tap(c, n); }
/* EF/Shaders/depth_bilateral.frag:32-75. */
void orc_bilateral(const uint16_t* in, uint16_t* out, int w, int h, float maxD)
{
    const float sigma_space2_inv_half = 0.024691358f;
    const float sigma_color2_inv_half = 0.000555556f;
    const int R = 6, D = R * 2 + 1;
    const uint32_t maxv = (uint32_t)(maxD * 1000.0f);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t value = in[y * w + x];
            if (value > maxv || value < 300u) { out[y * w + x] = 0; continue; }
            int tx = imin(x - D / 2 + D, w), ty = imin(y - D / 2 + D, h);
            float sum1 = 0, sum2 = 0;
            for (int cy = imax(y - D / 2, 0); cy < ty; ++cy)
                for (int cx = imax(x - D / 2, 0); cx < tx; ++cx) {
                    uint32_t tmp = in[bilateral_tap(cy, h) * w + bilateral_tap(cx, w)];
                    float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) +
                                   ((float)y - (float)cy) * ((float)y - (float)cy);
                    float color2 = ((float)value - (float)tmp) * ((float)value - (float)tmp);
                    float weight = ifx_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
                    sum1 += (float)tmp * weight;
                    sum2 += weight;
                }
            out[y * w + x] = (uint16_t)(uint32_t)roundf(sum1 / sum2);
        }
}

/* EF/Shaders/depth_metric.frag:30-39 */
void orc_metric(const uint16_t* in, float* out, int w, int h, float maxD)
{
    const uint32_t maxv = (uint32_t)(maxD * 1000.0f);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < w * h; i++) {
        uint32_t v = in[i];
        out[i] = (v > maxv || v < 300u) ? 0.0f : (float)v / 1000.0f;
    }
}

/* ======================================================================= pyramid kernels (a3) */

/* pyrDownGaussKernel, EF/Cuda/cudafuncs.cu:57-94 (sigma_color = 30) */
void orc_pyrdown_u16(const uint16_t* src, int sw, int sh, uint16_t* dst)
{
    const int D = 5;
    const float sigma_color = 30;
    const float weights[3] = {0.375f, 0.25f, 0.0625f};
    int dw = sw / 2, dh = sh / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int center = src[(2 * y) * sw + 2 * x];
            int x_mi = imax(0, 2 * x - D / 2) - 2 * x, y_mi = imax(0, 2 * y - D / 2) - 2 * y;
            int x_ma = imin(sw, 2 * x - D / 2 + D) - 2 * x, y_ma = imin(sh, 2 * y - D / 2 + D) - 2 * y;
            float sum = 0, wall = 0;
            for (int yi = y_mi; yi < y_ma; ++yi)
                for (int xi = x_mi; xi < x_ma; ++xi) {
                    int val = src[(2 * y + yi) * sw + 2 * x + xi];
                    if ((float)abs(val - center) < 3 * sigma_color) {
                        sum += val * weights[abs(xi)] * weights[abs(yi)];
                        wall += weights[abs(xi)] * weights[abs(yi)];
                    }
                }
            dst[y * dw + x] = (uint16_t)(int)(sum / wall);
        }
}

/* computeVmapKernel, EF/Cuda/cudafuncs.cu:109-133 */
void orc_vmap(const uint16_t* depth, int w, int h, float fx, float fy, float cx, float cy,
              float cutoff, float* vmap)
{
    float fx_inv = 1.f / fx, fy_inv = 1.f / fy;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < h; v++)
        for (int u = 0; u < w; u++) {
            float z = depth[v * w + u] / 1000.f;
            if (z != 0 && z < cutoff) {
                vmap[v * w + u] = z * (u - cx) * fx_inv;
                vmap[(v + h) * w + u] = z * (v - cy) * fy_inv;
                vmap[(v + 2 * h) * w + u] = z;
            } else {
                vmap[v * w + u] = orc_qnan();
                /* y/z planes are left untouched by the reference; the oracle writes NaN so that
                 * comparisons are well defined (consumers only test the x plane). */
                vmap[(v + h) * w + u] = orc_qnan();
                vmap[(v + 2 * h) * w + u] = orc_qnan();
            }
        }
}

/* computeNmapKernel, EF/Cuda/cudafuncs.cu:151-188 */
void orc_nmap(const float* vmap, int w, int h, float* nmap)
{
#pragma omp parallel for schedule(static)
    for (int v = 0; v < h; v++)
        for (int u = 0; u < w; u++) {
            float* nx = &nmap[v * w + u];
            float* ny = &nmap[(v + h) * w + u];
            float* nz = &nmap[(v + 2 * h) * w + u];
            *nx = *ny = *nz = orc_qnan();
            if (u == w - 1 || v == h - 1) continue;
            v3 v00, v01, v10;
            v00.x = vmap[v * w + u];
            v01.x = vmap[v * w + u + 1];
            v10.x = vmap[(v + 1) * w + u];
            if (!isnan(v00.x) && !isnan(v01.x) && !isnan(v10.x)) {
                v00.y = vmap[(v + h) * w + u];
                v01.y = vmap[(v + h) * w + u + 1];
                v10.y = vmap[(v + 1 + h) * w + u];
                v00.z = vmap[(v + 2 * h) * w + u];
                v01.z = vmap[(v + 2 * h) * w + u + 1];
                v10.z = vmap[(v + 1 + 2 * h) * w + u];
                v3 r = v3normalized(v3cross(v3sub(v01, v00), v3sub(v10, v00)));
                *nx = r.x; *ny = r.y; *nz = r.z;
            }
        }
}

/* copyMapsKernel, EF/Cuda/cudafuncs.cu:270-310 */
void orc_copy_maps(const float* v4, const float* n4, int w, int h, float* vmap, float* nmap)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* vs = &v4[(y * w + x) * 4];
            const float* ns = &n4[(y * w + x) * 4];
            int ok = !(vs[2] == 0);
            for (int c = 0; c < 3; c++) {
                vmap[(y + c * h) * w + x] = ok ? vs[c] : orc_qnan();
                nmap[(y + c * h) * w + x] = ok ? ns[c] : orc_qnan();
            }
        }
}

/* resizeMapKernel<normalize>, EF/Cuda/cudafuncs.cu:365-416 */
void orc_resize_map(const float* in, int sw, int sh, float* out, int normalize)
{
    int dw = sw / 2, dh = sh / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int xs = x * 2, ys = y * 2;
            float x00 = in[ys * sw + xs], x01 = in[ys * sw + xs + 1];
            float x10 = in[(ys + 1) * sw + xs], x11 = in[(ys + 1) * sw + xs + 1];
            if (isnan(x00) || isnan(x01) || isnan(x10) || isnan(x11)) {
                out[y * dw + x] = orc_qnan();
                out[(y + dh) * dw + x] = orc_qnan();
                out[(y + 2 * dh) * dw + x] = orc_qnan();
                continue;
            }
            v3 n;
            n.x = (x00 + x01 + x10 + x11) / 4;
            const float* py = in + sh * sw;
            n.y = (py[ys * sw + xs] + py[ys * sw + xs + 1] + py[(ys + 1) * sw + xs] + py[(ys + 1) * sw + xs + 1]) / 4;
            const float* pz = in + 2 * sh * sw;
            n.z = (pz[ys * sw + xs] + pz[ys * sw + xs + 1] + pz[(ys + 1) * sw + xs] + pz[(ys + 1) * sw + xs + 1]) / 4;
            if (normalize) n = v3normalized(n);
            out[y * dw + x] = n.x;
            out[(y + dh) * dw + x] = n.y;
            out[(y + 2 * dh) * dw + x] = n.z;
        }
}

/* tranformMapsKernel (in place), EF/Cuda/cudafuncs.cu:206-248 */
void orc_transform_maps(float* vmap, float* nmap, int w, int h, const float* R, const float* t)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float* px[3] = {&vmap[y * w + x], &vmap[(y + h) * w + x], &vmap[(y + 2 * h) * w + x]};
            if (!isnan(*px[0])) {
                v3 d = v3add(m33mul(R, v3m(*px[0], *px[1], *px[2])), v3m(t[0], t[1], t[2]));
                *px[0] = d.x; *px[1] = d.y; *px[2] = d.z;
            } else { *px[1] = *px[2] = orc_qnan(); }
            float* pn[3] = {&nmap[y * w + x], &nmap[(y + h) * w + x], &nmap[(y + 2 * h) * w + x]};
            if (!isnan(*pn[0])) {
                v3 d = m33mul(R, v3m(*pn[0], *pn[1], *pn[2]));
                *pn[0] = d.x; *pn[1] = d.y; *pn[2] = d.z;
            } else { *pn[1] = *pn[2] = orc_qnan(); }
        }
}

/* verticesToDepthKernel, EF/Cuda/cudafuncs.cu:526-537 */
void orc_vertices_to_depth(const float* v4, int w, int h, float cutoff, float* d)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < w * h; i++) {
        float z = v4[i * 4 + 2];
        d[i] = (z > cutoff || z <= 0) ? orc_qnan() : z;
    }
}

static const float kGauss25[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};

/* pyrDownKernelGaussF, EF/Cuda/cudafuncs.cu:332-363 (int count of float weights reproduced) */
void orc_pyrdown_gauss_f(const float* src, int sw, int sh, float* dst)
{
    const int D = 5;
    int dw = sw / 2, dh = sh / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int tx = imin(2 * x - D / 2 + D, sw - 1), ty = imin(2 * y - D / 2 + D, sh - 1);
            float sum = 0;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    float s = src[cy * sw + cx];
                    if (!isnan(s)) {
                        float g = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += s * g;
                        count += g;
                    }
                }
            dst[y * dw + x] = (float)(sum / (float)count);
        }
}

/* pyrDownKernelIntensityGauss, EF/Cuda/cudafuncs.cu:470-500 */
void orc_pyrdown_gauss_u8(const uint8_t* src, int sw, int sh, uint8_t* dst)
{
    const int D = 5;
    int dw = sw / 2, dh = sh / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int tx = imin(2 * x - D / 2 + D, sw - 1), ty = imin(2 * y - D / 2 + D, sh - 1);
            float sum = 0;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    if (src[cy * sw + cx] > 0) {
                        float g = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += src[cy * sw + cx] * g;
                        count += g;
                    }
                }
            /* float -> uchar; count==0 gives NaN which CUDA converts to 0 */
            dst[y * dw + x] = count ? (uint8_t)orc_f2i_rz(sum / (float)count) : 0;
        }
}

/* bgr2IntensityKernel, EF/Cuda/cudafuncs.cu:550-563.  The texture is RGBA8 uploaded from RGB, so
 * .x=R .y=G .z=B and the "BGR" weights are applied to RGB order, as in the reference. */
void orc_rgb_to_intensity(const uint8_t* rgb, int stride, int n, uint8_t* dst)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const uint8_t* s = rgb + (size_t)i * stride;
        int value = (int)((float)s[0] * 0.114f + (float)s[1] * 0.299f + (float)s[2] * 0.587f);
        dst[i] = (uint8_t)value;
    }
}

/* applyKernel, EF/Cuda/cudafuncs.cu:583-607 with the coefficient tables of :615-621 */
void orc_sobel(const uint8_t* img, int w, int h, int16_t* dx, int16_t* dy)
{
    static const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    static const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float dxVal = 0, dyVal = 0;
            int k = 8;
            for (int j = imax(y - 1, 0); j <= imin(y + 1, h - 1); j++)
                for (int i = imax(x - 1, 0); i <= imin(x + 1, w - 1); i++) {
                    dxVal += (float)img[j * w + i] * gsx[k];
                    dyVal += (float)img[j * w + i] * gsy[k];
                    --k;
                }
            dx[y * w + x] = (int16_t)dxVal;
            dy[y * w + x] = (int16_t)dyVal;
        }
}

/* projectPointsKernel, EF/Cuda/cudafuncs.cu:641-659 */
void orc_project_cloud(const float* depth, int w, int h, float fx, float fy, float cx, float cy,
                       float* cloud3)
{
    float invFx = 1.0f / fx, invFy = 1.0f / fy;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float z = depth[y * w + x];
            cloud3[(y * w + x) * 3 + 0] = (float)((x - cx) * z * invFx);
            cloud3[(y * w + x) * 3 + 1] = (float)((y - cy) * z * invFy);
            cloud3[(y * w + x) * 3 + 2] = z;
        }
}

/* ======================================================================= reductions (a4-a7) */

static void accum_products7(const float* row, int found, double* acc29, const int* e)
{
    int s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) {
            /* JtJJtrSE3: the f32 product of the reference (EF/Cuda/types.cuh:101-152), rounded to the entry's grid, summed exactly */
            acc29[s++] += orc_quant(row[i] * row[j], e[i] + e[j] - ORC_EXACT_TERM_BITS);
        }
    /* order: aa..ag, bb..bg, cc..cg, dd..dg, ee,ef,eg, ff,fg  = 27 ; then residual, inliers */
    acc29[27] += orc_quant(row[6] * row[6], 2 * e[6] - ORC_EXACT_TERM_BITS);
    acc29[28] += found ? 1.0 : 0.0;
}

/* ---- mode 3 of orc_set_sum_order: the sums as THE REFERENCE builds them -- f32 products, f32 additions, in the tree of its kernels with the launch
 * table's GTX 1080 row (EF/Utils/GPUConfig.h:123-126: icpStep 128 x 160, rgbStep 160 x 80, so3Step 160 x 80).  Thread (b, t) of a <<<B, T>>> launch adds
 * the products of the pixels b T + t, + B T, ... one after the other (EF/Cuda/reduce.cu:397-402); blockReduceSum (:133-165) folds a warp by shuffles
 * (offsets 16, 8, 4, 2, 1), parks one partial per warp in shared memory and folds those in warp 0; reduceSum<<<1, 512>>> (:167-185) does the same over the
 * block partials.  Not what parity is asserted against (the reference's tree depends on the GPU model it runs on): it exists to MEASURE
 * how far the reference's arithmetic is from the exact sums over a trajectory (tests/test_oracle_cpu.py::test_reference_shaped_f32_tree_gap). */
static void ref_warp_fold(float v[32][29], int K)
{
    for (int off = 16; off >= 1; off >>= 1) {
        float nv[32][29];
        for (int l = 0; l < 32; l++)
            for (int k = 0; k < K; k++) nv[l][k] = v[l][k] + v[l + off < 32 ? l + off : l][k];   /* __shfl_down: a lane past the end reads itself */
        memcpy(v, nv, sizeof(nv));
    }
}
static void ref_block_reduce(const float* thread_vals /* [T][K] */, int T, int K, float* out /* [K] */)
{
    float shared[32][29];
    memset(shared, 0, sizeof(shared));
    const int nw = T / 32;
    for (int wp = 0; wp < nw; wp++) {
        float v[32][29];
        for (int l = 0; l < 32; l++)
            for (int k = 0; k < K; k++) v[l][k] = thread_vals[(size_t)(wp * 32 + l) * K + k];
        ref_warp_fold(v, K);
        for (int k = 0; k < K; k++) shared[wp][k] = v[0][k];
    }
    float v[32][29];
    for (int l = 0; l < 32; l++)
        for (int k = 0; k < K; k++) v[l][k] = l < nw ? shared[l][k] : 0.0f;
    ref_warp_fold(v, K);
    for (int k = 0; k < K; k++) out[k] = v[0][k];
}
/* prods: [N][K] f32 products of every pixel in the kernel's linear order */
static void ref_tree_sum(const float* prods, int N, int K, int T, int B, float* out)
{
    float* part = (float*)calloc((size_t)B * K, sizeof(float));
    float* tv = (float*)malloc((size_t)T * K * sizeof(float));
    for (int b = 0; b < B; b++) {
        for (int t = 0; t < T; t++) {
            float* sum = &tv[(size_t)t * K];
            for (int k = 0; k < K; k++) sum[k] = 0.0f;
            for (long i = (long)b * T + t; i < N; i += (long)B * T)
                for (int k = 0; k < K; k++) sum[k] += prods[(size_t)i * K + k];
        }
        ref_block_reduce(tv, T, K, &part[(size_t)b * K]);
    }
    free(tv);
    float* tv2 = (float*)calloc((size_t)512 * K, sizeof(float));      /* reduceSum<<<1, MAX_THREADS = 512>>>(sum, out, blocks) */
    for (int t = 0; t < 512; t++)
        for (int i = t; i < B; i += 512)
            for (int k = 0; k < K; k++) tv2[(size_t)t * K + k] += part[(size_t)i * K + k];
    ref_block_reduce(tv2, 512, K, out);
    free(tv2); free(part);
}
static void ref_products7(const float* row, int found, float* p29)
{
    int s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) p29[s++] = row[i] * row[j];
    p29[27] = row[6] * row[6];
    p29[28] = found ? 1.0f : 0.0f;
}

/* ICPReduction::search/getProducts, EF/Cuda/reduce.cu:282-387; sums of :397-402 (+ reduceSum) */
void orc_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr,
                  const float* nmap_curr, const float* Rprev_inv, const float* tprev, float fx,
                  float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                  float dist_thres, float angle_thres, int w, int h, float* out29)
{
    /* one f64 partial sum per image row, rows added in order: the same result for any number of OpenMP threads */
    double (*racc)[29] = (double (*)[29])calloc((size_t)h, sizeof(double[29]));
    float* prods = orc_sum_reverse == 3 ? (float*)malloc((size_t)w * h * 29 * sizeof(float)) : NULL;
    v3 tc = v3m(tcurr[0], tcurr[1], tcurr[2]), tp = v3m(tprev[0], tprev[1], tprev[2]);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double* acc = racc[y];
            float row[7] = {0, 0, 0, 0, 0, 0, 0};
            int found = 0;
            v3 vcurr = v3m(vmap_curr[y * w + x], vmap_curr[(y + h) * w + x], vmap_curr[(y + 2 * h) * w + x]);
            if (!isnan(vcurr.x)) {
                v3 vcurr_g = v3add(m33mul(Rcurr, vcurr), tc);
                v3 vcurr_cp = m33mul(Rprev_inv, v3sub(vcurr_g, tp));
                int ux = orc_f2i_rn(vcurr_cp.x * fx / vcurr_cp.z + cx);
                int uy = orc_f2i_rn(vcurr_cp.y * fy / vcurr_cp.z + cy);
                if (!(ux < 0 || uy < 0 || ux >= w || uy >= h || vcurr_cp.z < 0)) {
                    v3 vprev_g = v3m(vmap_g_prev[uy * w + ux], vmap_g_prev[(uy + h) * w + ux], vmap_g_prev[(uy + 2 * h) * w + ux]);
                    v3 ncurr = v3m(nmap_curr[y * w + x], nmap_curr[(y + h) * w + x], nmap_curr[(y + 2 * h) * w + x]);
                    v3 ncurr_g = m33mul(Rcurr, ncurr);
                    v3 nprev_g = v3m(nmap_g_prev[uy * w + ux], nmap_g_prev[(uy + h) * w + ux], nmap_g_prev[(uy + 2 * h) * w + ux]);
                    float dist = v3norm(v3sub(vprev_g, vcurr_g));
                    float sine = v3norm(v3cross(ncurr_g, nprev_g));
                    found = (sine < angle_thres && dist <= dist_thres && !isnan(ncurr.x) && !isnan(nprev_g.x));
                    if (found) {
                        v3 s_cp = m33mul(Rprev_inv, v3sub(vcurr_g, tp));
                        v3 d_cp = m33mul(Rprev_inv, v3sub(vprev_g, tp));
                        v3 n_cp = m33mul(Rprev_inv, nprev_g);
                        v3 c = v3cross(s_cp, n_cp);
                        row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
                        row[3] = c.x; row[4] = c.y; row[5] = c.z;
                        row[6] = v3dot(n_cp, v3sub(s_cp, d_cp));
                    }
                }
            }
            accum_products7(row, found, acc, ORC_E_ICP);
            if (prods) ref_products7(row, found, &prods[(size_t)(y * w + x) * 29]);
        }
    double tot[29];
    for (int i = 0; i < 29; i++) tot[i] = 0;
    for (int yy = 0; yy < h; yy++) {
        const int y = orc_sum_reverse == 1 ? h - 1 - yy : yy;
        for (int i = 0; i < 29; i++) tot[i] += orc_sum_reverse == 2 ? (double)(float)racc[y][i] : racc[y][i];
    }
    for (int i = 0; i < 29; i++) out29[i] = (float)tot[i];
    if (prods) { ref_tree_sum(prods, w * h, 29, 128, 160, out29); free(prods); }   /* icpStepMap["GeForce GTX 1080"] */
    free(racc);
}

/* RGBResidual::getProducts, EF/Cuda/reduce.cu:768-842 */
void orc_rgb_residual(float min_scale, const int16_t* didx, const int16_t* didy,
                      const float* last_depth, const float* next_depth, const uint8_t* last_img,
                      const uint8_t* next_img, orc_dataterm* corres, float max_depth_delta,
                      const float* kt, const float* krkinv, int w, int h, int* count, int* sigma)
{
    const int border = 16;
    int cnt = 0, sig = 0;
#pragma omp parallel for schedule(static) reduction(+ : cnt, sig)
    for (int i = 0; i < h; i++)
        for (int j0 = 0; j0 < w; j0++) {
            orc_dataterm c;
            memset(&c, 0, sizeof(c));
            if (i >= border && i < h - border && j0 >= border && j0 < w - border && j0 < w - 5 && i < h - 1) {
                int valid = 1;
                for (int u = imax(i - 2, 0); u < imin(i + 2, h); u++)
                    for (int v = imax(j0 - 2, 0); v < imin(j0 + 2, w); v++)
                        valid = valid && (next_img[u * w + v] > 0);
                if (valid) {
                    short valx = didx[i * w + j0], valy = didy[i * w + j0];
                    float mTwo = (float)((valx * valx) + (valy * valy));
                    if (mTwo >= min_scale) {
                        int y = i, x = j0;
                        float d1 = next_depth[y * w + x];
                        if (!isnan(d1)) {
                            float td1 = (float)(d1 * (krkinv[6] * x + krkinv[7] * y + krkinv[8]) + kt[2]);
                            int u0 = orc_f2i_rn((d1 * (krkinv[0] * x + krkinv[1] * y + krkinv[2]) + kt[0]) / td1);
                            int v0 = orc_f2i_rn((d1 * (krkinv[3] * x + krkinv[4] * y + krkinv[5]) + kt[1]) / td1);
                            if (u0 >= 0 && v0 >= 0 && u0 < w && v0 < h) {
                                float d0 = last_depth[v0 * w + u0];
                                if (d0 > 0 && fabsf(td1 - d0) <= max_depth_delta && last_img[v0 * w + u0] != 0) {
                                    c.zero_x = (int16_t)u0; c.zero_y = (int16_t)v0;
                                    c.one_x = (int16_t)x; c.one_y = (int16_t)y;
                                    c.diff = (float)next_img[y * w + x] - (float)last_img[v0 * w + u0];
                                    c.valid = 1;
                                    cnt += 1;
                                    sig += (int)(c.diff * c.diff);
                                }
                            }
                        }
                    }
                }
            }
            corres[i * w + j0] = c;
        }
    *count = cnt;
    *sigma = sig;
}

/* RGBReduction::getProducts, EF/Cuda/reduce.cu:512-595 */
void orc_rgb_step(const orc_dataterm* corres, float sigma, const float* cloud3, float fx, float fy,
                  const int16_t* didx, const int16_t* didy, float sobel_scale, int w, int h,
                  float* out29)
{
    double (*racc)[29] = (double (*)[29])calloc((size_t)h, sizeof(double[29]));   /* per-row partial sums, as orc_icp_step */
    float* prods = orc_sum_reverse == 3 ? (float*)malloc((size_t)w * h * 29 * sizeof(float)) : NULL;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
    for (int k = y * w; k < (y + 1) * w; k++) {   /* a row belongs to one thread: its partial sum is built in pixel order */
        double* acc = racc[y];
        const orc_dataterm* c = &corres[k];
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        if (c->valid) {
            float wgt = sigma + fabsf(c->diff);
            wgt = wgt > 1.19209290E-07F ? 1.0f / wgt : 1.0f;
            if (sigma == -1) wgt = 1;
            row[6] = -wgt * c->diff;
            const float* cp = &cloud3[(c->zero_y * w + c->zero_x) * 3];
            float invz = (float)(1.0 / cp[2]);
            float dI_dx = wgt * sobel_scale * didx[c->one_y * w + c->one_x];
            float dI_dy = wgt * sobel_scale * didy[c->one_y * w + c->one_x];
            float v0 = dI_dx * fx * invz;
            float v1 = dI_dy * fy * invz;
            float v2 = -(v0 * cp[0] + v1 * cp[1]) * invz;
            row[0] = v0; row[1] = v1; row[2] = v2;
            row[3] = -cp[2] * v1 + cp[1] * v2;
            row[4] = cp[2] * v0 - cp[0] * v2;
            row[5] = -cp[1] * v0 + cp[0] * v1;
        }
        accum_products7(row, c->valid, acc, ORC_E_RGB);
        if (prods) ref_products7(row, c->valid, &prods[(size_t)k * 29]);
    }
    double tot[29];
    for (int i = 0; i < 29; i++) tot[i] = 0;
    for (int yy = 0; yy < h; yy++) {
        const int y = orc_sum_reverse == 1 ? h - 1 - yy : yy;
        for (int i = 0; i < 29; i++) tot[i] += orc_sum_reverse == 2 ? (double)(float)racc[y][i] : racc[y][i];
    }
    for (int i = 0; i < 29; i++) out29[i] = (float)tot[i];
    if (prods) { ref_tree_sum(prods, w * h, 29, 160, 80, out29); free(prods); }   /* rgbStepMap["GeForce GTX 1080"] */
    free(racc);
}

/* SO3Reduction::getProducts, EF/Cuda/reduce.cu:954-1055 */
void orc_so3_step(const uint8_t* last_img, const uint8_t* next_img, const float* ib,
                  const float* kinv, const float* krlr, int w, int h, float* out11)
{
    double (*racc)[11] = (double (*)[11])calloc((size_t)h, sizeof(double[11]));   /* per-row partial sums, as orc_icp_step */
    float* prods = orc_sum_reverse == 3 ? (float*)malloc((size_t)w * h * 11 * sizeof(float)) : NULL;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double* acc = racc[y];
            v3 up = v3m((float)x, (float)y, 1.0f);
            v3 wp = m33mul(ib, up);
            int wx = orc_f2i_rn(wp.x / wp.z), wy = orc_f2i_rn(wp.y / wp.z);
            int found = (wx >= 1 && wx < w - 1 && wy >= 1 && wy < h - 1 && x >= 1 && x < w - 1 && y >= 1 && y < h - 1);
            float row[4] = {0, 0, 0, 0};
            if (found) {
#define GRADX(img, px, py) ((((float)img[(py) * w + (px) - 1] + (float)img[(py) * w + (px)]) / 2.0f) - (((float)img[(py) * w + (px) + 1] + (float)img[(py) * w + (px)]) / 2.0f))
#define GRADY(img, px, py) ((((float)img[((py) - 1) * w + (px)] + (float)img[(py) * w + (px)]) / 2.0f) - (((float)img[((py) + 1) * w + (px)] + (float)img[(py) * w + (px)]) / 2.0f))
                float gx = (GRADX(next_img, wx, wy) + GRADX(last_img, x, y)) / 2.0f;
                float gy = (GRADY(next_img, wx, wy) + GRADY(last_img, x, y)) / 2.0f;
                v3 pt = m33mul(kinv, up);
                float z2 = pt.z * pt.z;
                float a = krlr[0], b = krlr[1], c = krlr[2], d = krlr[3], e = krlr[4], f = krlr[5], g = krlr[6], hh = krlr[7], ii = krlr[8];
                v3 lp = v3m(((pt.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
                            ((pt.z * (e * gy + b * gx)) - (gy * hh * y) - (gx * hh * x)) / z2,
                            ((pt.z * (f * gy + c * gx)) - (gy * ii * y) - (gx * ii * x)) / z2);
                v3 jr = v3cross(lp, pt);
                row[0] = jr.x; row[1] = jr.y; row[2] = jr.z;
                row[3] = -((float)next_img[wy * w + wx] - (float)last_img[y * w + x]);
            }
            int s = 0;
            for (int i = 0; i < 3; i++)
                for (int j = i; j < 4; j++) acc[s++] += orc_quant(row[i] * row[j], ORC_E_SO3[i] + ORC_E_SO3[j] - ORC_SO3_TERM_BITS);
            acc[9] += orc_quant(row[3] * row[3], 2 * ORC_E_SO3[3] - ORC_SO3_TERM_BITS);
            acc[10] += found ? 1.0 : 0.0;
            if (prods) {   /* JtJJtrSO3, EF/Cuda/types.cuh:154-181 */
                float* pp = &prods[(size_t)(y * w + x) * 11];
                int q = 0;
                for (int i = 0; i < 3; i++)
                    for (int j = i; j < 4; j++) pp[q++] = row[i] * row[j];
                pp[9] = row[3] * row[3];
                pp[10] = found ? 1.0f : 0.0f;
            }
        }
    double tot[11];
    for (int i = 0; i < 11; i++) tot[i] = 0;
    for (int yy = 0; yy < h; yy++) {
        const int y = orc_sum_reverse == 1 ? h - 1 - yy : yy;
        for (int i = 0; i < 11; i++) tot[i] += orc_sum_reverse == 2 ? (double)(float)racc[y][i] : racc[y][i];
    }
    for (int i = 0; i < 11; i++) out11[i] = (float)tot[i];
    if (prods) { ref_tree_sum(prods, w * h, 11, 160, 80, out11); free(prods); }   /* so3StepMap["GeForce GTX 1080"] */
    free(racc);
}

/* ======================================================================= tracker object (a3,a8) */

struct orc_tracker {
    int w, h;
    float fx, fy, cx, cy;
    int lw[ORC_NUM_PYRS], lh[ORC_NUM_PYRS];
    uint16_t* depth_tmp[ORC_NUM_PYRS];
    float *vmap_curr[ORC_NUM_PYRS], *nmap_curr[ORC_NUM_PYRS];
    float *vmap_prev[ORC_NUM_PYRS], *nmap_prev[ORC_NUM_PYRS];
    float* last_depth[ORC_NUM_PYRS];
    float* next_depth[ORC_NUM_PYRS];
    uint8_t *last_img[ORC_NUM_PYRS], *next_img[ORC_NUM_PYRS], *lastnext_img[ORC_NUM_PYRS];
    int16_t *didx[ORC_NUM_PYRS], *didy[ORC_NUM_PYRS];
    float* cloud[ORC_NUM_PYRS];
    orc_dataterm* corres[ORC_NUM_PYRS];
    float* vmaps_tmp; /* float4 */
    double lastA[36], lastb[6];
    float last_icp29[29], last_rgb29[29];
    int last_sigma, last_count;
};

orc_tracker* orc_tracker_create(int w, int h, float fx, float fy, float cx, float cy)
{
    orc_tracker* t = (orc_tracker*)calloc(1, sizeof(*t));
    t->w = w; t->h = h; t->fx = fx; t->fy = fy; t->cx = cx; t->cy = cy;
    for (int i = 0; i < ORC_NUM_PYRS; i++) {
        int lw = w >> i, lh = h >> i, n = lw * lh;
        t->lw[i] = lw; t->lh[i] = lh;
        t->depth_tmp[i] = (uint16_t*)calloc(n, 2);
        t->vmap_curr[i] = (float*)calloc(3 * n, 4);
        t->nmap_curr[i] = (float*)calloc(3 * n, 4);
        t->vmap_prev[i] = (float*)calloc(3 * n, 4);
        t->nmap_prev[i] = (float*)calloc(3 * n, 4);
        t->last_depth[i] = (float*)calloc(n, 4);
        t->next_depth[i] = (float*)calloc(n, 4);
        t->last_img[i] = (uint8_t*)calloc(n, 1);
        t->next_img[i] = (uint8_t*)calloc(n, 1);
        t->lastnext_img[i] = (uint8_t*)calloc(n, 1);
        t->didx[i] = (int16_t*)calloc(n, 2);
        t->didy[i] = (int16_t*)calloc(n, 2);
        t->cloud[i] = (float*)calloc(3 * n, 4);
        t->corres[i] = (orc_dataterm*)calloc(n, sizeof(orc_dataterm));
    }
    t->vmaps_tmp = (float*)calloc((size_t)w * h * 4, 4);
    return t;
}

void orc_tracker_destroy(orc_tracker* t)
{
    if (!t) return;
    for (int i = 0; i < ORC_NUM_PYRS; i++) {
        free(t->depth_tmp[i]); free(t->vmap_curr[i]); free(t->nmap_curr[i]); free(t->vmap_prev[i]);
        free(t->nmap_prev[i]); free(t->last_depth[i]); free(t->next_depth[i]); free(t->last_img[i]);
        free(t->next_img[i]); free(t->lastnext_img[i]); free(t->didx[i]); free(t->didy[i]);
        free(t->cloud[i]); free(t->corres[i]);
    }
    free(t->vmaps_tmp);
    free(t);
}

/* RGBDOdometry::initFirstRGB, EF/Utils/RGBDOdometry.cpp:249-265 */
void orc_tracker_init_first_rgb(orc_tracker* t, const uint8_t* rgb)
{
    orc_rgb_to_intensity(rgb, 3, t->w * t->h, t->lastnext_img[0]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; i++)
        orc_pyrdown_gauss_u8(t->lastnext_img[i], t->lw[i], t->lh[i], t->lastnext_img[i + 1]);
}

/* populateRGBDData, EF/Utils/RGBDOdometry.cpp:208-235 */
static void populate_rgbd(orc_tracker* t, const uint8_t* img, int stride, float** depths, uint8_t** images)
{
    orc_vertices_to_depth(t->vmaps_tmp, t->w, t->h, 6.0f /* maxDepthRGB :37 */, depths[0]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; i++) orc_pyrdown_gauss_f(depths[i], t->lw[i], t->lh[i], depths[i + 1]);
    orc_rgb_to_intensity(img, stride, t->w * t->h, images[0]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; i++) orc_pyrdown_gauss_u8(images[i], t->lw[i], t->lh[i], images[i + 1]);
}

/* initICPModel + initRGBModel, EF/Utils/RGBDOdometry.cpp:169-206, 237-241 */
void orc_tracker_init_model(orc_tracker* t, const float* model_v4, const float* model_n4,
                            const uint8_t* model_rgba, const float* pose)
{
    memcpy(t->vmaps_tmp, model_v4, (size_t)t->w * t->h * 16);
    orc_copy_maps(model_v4, model_n4, t->w, t->h, t->vmap_prev[0], t->nmap_prev[0]);
    for (int i = 1; i < ORC_NUM_PYRS; i++) {
        orc_resize_map(t->vmap_prev[i - 1], t->lw[i - 1], t->lh[i - 1], t->vmap_prev[i], 0);
        orc_resize_map(t->nmap_prev[i - 1], t->lw[i - 1], t->lh[i - 1], t->nmap_prev[i], 1);
    }
    float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    float tv[3] = {pose[3], pose[7], pose[11]};
    for (int i = 0; i < ORC_NUM_PYRS; i++) orc_transform_maps(t->vmap_prev[i], t->nmap_prev[i], t->lw[i], t->lh[i], R, tv);
    populate_rgbd(t, model_rgba, 4, t->last_depth, t->last_img);
}

/* initICP(filteredDepth) + initRGB, EF/Utils/RGBDOdometry.cpp:118-142, 243-247.
 * NOTE (reference behaviour): initRGB re-reads vmaps_tmp, which still holds the MODEL vertices
 * written by initICPModel, so nextDepth == lastDepth. */
void orc_tracker_init_frame(orc_tracker* t, const uint16_t* depth_filtered, const uint8_t* rgb,
                            float depth_cutoff)
{
    memcpy(t->depth_tmp[0], depth_filtered, (size_t)t->w * t->h * 2);
    for (int i = 1; i < ORC_NUM_PYRS; i++) orc_pyrdown_u16(t->depth_tmp[i - 1], t->lw[i - 1], t->lh[i - 1], t->depth_tmp[i]);
    for (int i = 0; i < ORC_NUM_PYRS; i++) {
        float div = (float)(1 << i);
        orc_vmap(t->depth_tmp[i], t->lw[i], t->lh[i], t->fx / div, t->fy / div, t->cx / div, t->cy / div, depth_cutoff, t->vmap_curr[i]);
        orc_nmap(t->vmap_curr[i], t->lw[i], t->lh[i], t->nmap_curr[i]);
    }
    populate_rgbd(t, rgb, 3, t->next_depth, t->next_img);
}

/* initICP(predictedVertices, predictedNormals, depthCutoff) + initRGB(predictedImage), EF/Utils/RGBDOdometry.cpp:144-167, 243-247.
 * Unlike the depth-image variant this one DOES refresh vmaps_tmp, so nextDepth is the depth of these vertices. */
void orc_tracker_init_frame_maps(orc_tracker* t, const float* v4, const float* n4, const uint8_t* rgba)
{
    memcpy(t->vmaps_tmp, v4, (size_t)t->w * t->h * 16);
    orc_copy_maps(v4, n4, t->w, t->h, t->vmap_curr[0], t->nmap_curr[0]);
    for (int i = 1; i < ORC_NUM_PYRS; i++) {
        orc_resize_map(t->vmap_curr[i - 1], t->lw[i - 1], t->lh[i - 1], t->vmap_curr[i], 0);
        orc_resize_map(t->nmap_curr[i - 1], t->lw[i - 1], t->lh[i - 1], t->nmap_curr[i], 1);
    }
    populate_rgbd(t, rgba, 4, t->next_depth, t->next_img);
}

/* getCovariance, EF/Utils/RGBDOdometry.cpp:605-608: lastA.cast<double>().lu().inverse() -- Gauss-Jordan with partial
 * pivoting (the same elimination order as the partial-pivot LU).  A singular matrix gives non-finite entries. */
void orc_tracker_covariance(orc_tracker* t, double* cov)
{
    double a[6][12];
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) { a[r][c] = t->lastA[r * 6 + c]; a[r][6 + c] = (r == c) ? 1.0 : 0.0; }
    for (int k = 0; k < 6; k++) {
        int piv = k;
        for (int r = k + 1; r < 6; r++) if (fabs(a[r][k]) > fabs(a[piv][k])) piv = r;
        if (piv != k) for (int c = 0; c < 12; c++) { double tmp = a[k][c]; a[k][c] = a[piv][c]; a[piv][c] = tmp; }
        double d = 1.0 / a[k][k];
        for (int c = 0; c < 12; c++) a[k][c] *= d;
        for (int r = 0; r < 6; r++) {
            if (r == k) continue;
            double f = a[r][k];
            for (int c = 0; c < 12; c++) a[r][c] -= f * a[k][c];
        }
    }
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) cov[r * 6 + c] = a[r][6 + c];
}

const void* orc_tracker_buffer(orc_tracker* t, const char* name, int l)
{
    if (!strcmp(name, "vmap_curr")) return t->vmap_curr[l];
    if (!strcmp(name, "nmap_curr")) return t->nmap_curr[l];
    if (!strcmp(name, "vmap_prev")) return t->vmap_prev[l];
    if (!strcmp(name, "nmap_prev")) return t->nmap_prev[l];
    if (!strcmp(name, "last_depth")) return t->last_depth[l];
    if (!strcmp(name, "next_depth")) return t->next_depth[l];
    if (!strcmp(name, "last_img")) return t->last_img[l];
    if (!strcmp(name, "next_img")) return t->next_img[l];
    if (!strcmp(name, "lastnext_img")) return t->lastnext_img[l];
    if (!strcmp(name, "didx")) return t->didx[l];
    if (!strcmp(name, "didy")) return t->didy[l];
    if (!strcmp(name, "cloud")) return t->cloud[l];
    if (!strcmp(name, "corres")) return t->corres[l];
    if (!strcmp(name, "depth_tmp")) return t->depth_tmp[l];
    if (!strcmp(name, "lastA")) return t->lastA;
    if (!strcmp(name, "lastb")) return t->lastb;
    if (!strcmp(name, "last_icp29")) return t->last_icp29;
    if (!strcmp(name, "last_rgb29")) return t->last_rgb29;
    return 0;
}

static void unpack29(const float* hd, float* A, float* b)
{
    int shift = 0; /* EF/Cuda/reduce.cu:475-486 */
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            float value = hd[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
}

static void k_matrix_d(float fx, float fy, float cx, float cy, double* K, double* Kinv)
{
    for (int i = 0; i < 9; i++) K[i] = Kinv[i] = 0;
    K[0] = fx; K[4] = fy; K[2] = cx; K[5] = cy; K[8] = 1;
    /* inverse of an upper-triangular intrinsics matrix */
    Kinv[0] = 1.0 / K[0]; Kinv[4] = 1.0 / K[4];
    Kinv[2] = -K[2] / K[0]; Kinv[5] = -K[5] / K[4]; Kinv[8] = 1;
}

/* RGBDOdometry::getIncrementalTransformation, EF/Utils/RGBDOdometry.cpp:267-603 (rgbOnly=false) */
void orc_tracker_run(orc_tracker* t, float* pose, float icp_weight, int pyramid, int fast_odom,
                     int so3, float* diag)
{
    int icp = icp_weight > 0;
    int rgb = icp_weight < 100;
    float Rprev[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    float tprev[3] = {pose[3], pose[7], pose[11]};
    float Rcurr[9], tcurr[3];
    memcpy(Rcurr, Rprev, sizeof(Rcurr));
    memcpy(tcurr, tprev, sizeof(tcurr));
    float lastICPError = 0, lastICPCount = 0, lastRGBError = 0, lastRGBCount = 0, lastSO3Error = 0, lastSO3Count = 0;

    if (rgb)
        for (int i = 0; i < ORC_NUM_PYRS; i++) orc_sobel(t->next_img[i], t->lw[i], t->lh[i], t->didx[i], t->didy[i]);

    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (so3) { /* :294-382 */
        int L = 2;
        float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double K[9], Kinv[9];
        float div = (float)(1 << L);
        k_matrix_d(t->fx / div, t->fy / div, t->cx / div, t->cy / div, K, Kinv);
        float lastError = FLT_MAX / 2, lastCount = FLT_MAX / 2;
        double lastResultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int it = 0; it < 10; it++) {
            double H[9], KR[9];
            matmul_d(3, K, resultR, KR);
            matmul_d(3, KR, Kinv, H);
            float ib[9], kinvf[9], krlr[9];
            for (int k = 0; k < 9; k++) { ib[k] = (float)H[k]; kinvf[k] = (float)Kinv[k]; krlr[k] = (float)KR[k]; }
            float o[11];
            orc_so3_step(t->lastnext_img[L], t->next_img[L], ib, kinvf, krlr, t->lw[L], t->lh[L], o);
            float jtj[9], jtr[3];
            int shift = 0; /* EF/Cuda/reduce.cu:1126-1137 */
            for (int i = 0; i < 3; ++i)
                for (int j = i; j < 4; ++j) {
                    float v = o[shift++];
                    if (j == 3) jtr[i] = v; else jtj[j * 3 + i] = jtj[i * 3 + j] = v;
                }
            lastSO3Error = sqrtf(o[9]) / o[10];
            lastSO3Count = o[10];
            if (lastSO3Error < lastError && lastCount == lastSO3Count) break;
            else if (lastSO3Error > lastError + 0.001) {
                lastSO3Error = lastError; lastSO3Count = lastCount;
                memcpy(resultR, lastResultR, sizeof(resultR));
                break;
            }
            lastError = lastSO3Error; lastCount = lastSO3Count;
            memcpy(lastResultR, resultR, sizeof(resultR));
            float delta[3];
            ldlt_solve_f(3, jtj, jtr, delta);
            double dd[3] = {delta[0], delta[1], delta[2]}, ru[9];
            rodrigues_d(dd, ru);
            float ruf[9], nr[9];
            for (int k = 0; k < 9; k++) ruf[k] = (float)ru[k];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) nr[i * 3 + j] = ruf[i * 3] * R_lr[j] + ruf[i * 3 + 1] * R_lr[3 + j] + ruf[i * 3 + 2] * R_lr[6 + j];
            memcpy(R_lr, nr, sizeof(nr));
            for (int k = 0; k < 9; k++) resultR[k] = R_lr[k];
        }
    }

    int iterations[3] = {fast_odom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0};
    float Rprev_inv[9];
    inv33_f(Rprev, Rprev_inv);
    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (so3)
        for (int x = 0; x < 3; x++)
            for (int y = 0; y < 3; y++) resultRt[x * 4 + y] = resultR[x * 3 + y];

    for (int i = ORC_NUM_PYRS - 1; i >= 0; i--) {
        float div = (float)(1 << i);
        float lfx = t->fx / div, lfy = t->fy / div, lcx = t->cx / div, lcy = t->cy / div;
        int lw = t->lw[i], lh = t->lh[i];
        if (rgb) orc_project_cloud(t->last_depth[i], lw, lh, lfx, lfy, lcx, lcy, t->cloud[i]);
        double K[9], Kinv[9];
        k_matrix_d(lfx, lfy, lcx, lcy, K, Kinv);
        lastRGBError = FLT_MAX;
        static const float minGrad[3] = {5, 3, 1};
        const double sobelScale = 1.0 / pow(2.0, 3);
        for (int j = 0; j < iterations[i]; j++) {
            double Rt[16];
            rigid_inv_d(resultRt, Rt);
            double R3[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
            double KR[9], KRK[9];
            matmul_d(3, K, R3, KR);
            matmul_d(3, KR, Kinv, KRK);
            float krk[9];
            for (int k = 0; k < 9; k++) krk[k] = (float)KRK[k];
            double tt[3] = {Rt[3], Rt[7], Rt[11]};
            float kt[3];
            for (int r = 0; r < 3; r++) kt[r] = (float)(K[r * 3] * tt[0] + K[r * 3 + 1] * tt[1] + K[r * 3 + 2] * tt[2]);
            int sigma = 0, rgbSize = 0;
            if (rgb)
                orc_rgb_residual((float)(pow(minGrad[i], 2.0) / pow(sobelScale, 2.0)), t->didx[i], t->didy[i], t->last_depth[i],
                                 t->next_depth[i], t->last_img[i], t->next_img[i], t->corres[i], 0.07f, kt, krk, lw, lh, &rgbSize, &sigma);
            /* :461 precedence quirk: ((float)sigma / rgbSize == 0) ? 1 : rgbSize, then sqrt */
            float q = (float)sigma / (float)rgbSize;
            float sigmaVal = (float)sqrt((double)((q == 0) ? 1 : rgbSize));
            float rgbError = (float)(sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));
            lastRGBError = rgbError;
            lastRGBCount = (float)rgbSize;
            t->last_sigma = sigma; t->last_count = rgbSize;

            float A_icp[36] = {0}, b_icp[6] = {0}, A_rgb[36] = {0}, b_rgb[6] = {0};
            if (icp) {
                float o[29];
                orc_icp_step(Rcurr, tcurr, t->vmap_curr[i], t->nmap_curr[i], Rprev_inv, tprev, lfx, lfy, lcx, lcy, t->vmap_prev[i],
                             t->nmap_prev[i], 0.10f, sinf(20.f * 3.14159254f / 180.f), lw, lh, o);
                memcpy(t->last_icp29, o, sizeof(o));
                unpack29(o, A_icp, b_icp);
                lastICPError = sqrtf(o[27]) / o[28];
                lastICPCount = o[28];
            }
            if (rgb) {
                float o[29];
                orc_rgb_step(t->corres[i], sigmaVal, t->cloud[i], lfx, lfy, t->didx[i], t->didy[i], (float)sobelScale, lw, lh, o);
                memcpy(t->last_rgb29, o, sizeof(o));
                unpack29(o, A_rgb, b_rgb);
            }
            double result[6];
            if (icp && rgb) {
                double wgt = icp_weight;
                for (int k = 0; k < 36; k++) t->lastA[k] = (double)A_rgb[k] + wgt * wgt * (double)A_icp[k];
                for (int k = 0; k < 6; k++) t->lastb[k] = (double)b_rgb[k] + wgt * (double)b_icp[k];
            } else if (icp) {
                for (int k = 0; k < 36; k++) t->lastA[k] = A_icp[k];
                for (int k = 0; k < 6; k++) t->lastb[k] = b_icp[k];
            } else {
                for (int k = 0; k < 36; k++) t->lastA[k] = A_rgb[k];
                for (int k = 0; k < 6; k++) t->lastb[k] = b_rgb[k];
            }
            ldlt_solve_d(6, t->lastA, t->lastb, result);

            /* computeUpdateSE3, EF/Utils/OdometryProvider.h:73-93 */
            double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rr[9];
            rodrigues_d(&result[3], Rr);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) upd[r * 4 + c] = Rr[r * 3 + c];
            upd[3] = result[0]; upd[7] = result[1]; upd[11] = result[2];
            matmul_d(4, upd, resultRt, resultRt);
            float oR[9], ot[3];
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) oR[r * 3 + c] = (float)resultRt[r * 4 + c];
                ot[r] = (float)resultRt[r * 4 + 3];
            }
            /* currentT = [Rprev|tprev] * rgbOdom^-1 (float Isometry), :575-583 */
            float iR[9], it[3];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) iR[r * 3 + c] = oR[c * 3 + r];
            for (int r = 0; r < 3; r++) it[r] = -(iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1] + iR[r * 3 + 2] * ot[2]);
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++)
                    Rcurr[r * 3 + c] = Rprev[r * 3] * iR[c] + Rprev[r * 3 + 1] * iR[3 + c] + Rprev[r * 3 + 2] * iR[6 + c];
                tcurr[r] = Rprev[r * 3] * it[0] + Rprev[r * 3 + 1] * it[1] + Rprev[r * 3 + 2] * it[2] + tprev[r];
            }
        }
    }

    if (rgb) {
        v3 d = v3m(tcurr[0] - tprev[0], tcurr[1] - tprev[1], tcurr[2] - tprev[2]);
        if (v3norm(d) > 0.3f) { memcpy(Rcurr, Rprev, sizeof(Rcurr)); memcpy(tcurr, tprev, sizeof(tcurr)); }
    }
    if (so3)
        for (int i = 0; i < ORC_NUM_PYRS; i++) { uint8_t* tmp = t->lastnext_img[i]; t->lastnext_img[i] = t->next_img[i]; t->next_img[i] = tmp; }

    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) pose[r * 4 + c] = Rcurr[r * 3 + c];
        pose[r * 4 + 3] = tcurr[r];
    }
    pose[12] = pose[13] = pose[14] = 0; pose[15] = 1;
    if (diag) {
        diag[0] = lastICPError; diag[1] = lastICPCount; diag[2] = lastRGBError; diag[3] = lastRGBCount;
        diag[4] = lastSO3Error; diag[5] = lastSO3Count; diag[6] = 0; diag[7] = 0;
    }
}
