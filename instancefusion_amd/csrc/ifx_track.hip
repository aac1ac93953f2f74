// ifx_track.hip -- tracking half of the path as HIP kernels for gfx950.
//
// Design (MI355X-first, not a translation of EF/Cuda/*.cu):
//  * the whole Gauss-Newton loop runs on the device: every reduction kernel leaves per-block
//    partial sums in HBM, and a one-block "solve" kernel sums them in a fixed order (double),
//    solves the 6x6 system (pivoted LDLT in double) and writes the next iteration's matrices into
//    DevState.  The host enqueues the fixed schedule {4,5,10} and never reads anything back
//    (the reference does 3 blocking readbacks per iteration, EF/Cuda/reduce.cu:469-473,660-664,931).
//  * reductions are wave64: 64-lane __shfl_down trees, one LDS slot per wave, fixed order.
//  * the 13x13 bilateral stages its depth tile through LDS.
#include "ifx_ctx.h"
#include <math.h>
#include <string.h>

// ======================================================================= preprocessing (a2)

// EF/Shaders/depth_bilateral.frag:32-75 + depth_metric.frag:30-39 (raw and filtered) in one pass.
#define BIL_R 6
#define BIL_BX 32
#define BIL_BY 8
// The 13 x 13 window of one pixel: rows in a loop, the thirteen taps of a row unrolled -- their LDS reads leave together, and the taps go through the exponential two at a
// time with the packed f32 instructions (ifx_expf2_nonpos: 36 M -> ~21 M VALU wave-instructions per frame; the kernel was 85 % VALU-bound, profiles/r06_c_pmc_bound.json).
// Every tap goes through the scalar form's operations in the scalar form's order, and the two sums take their terms in tap order: results are bit-identical.
template <bool SHIFTED, bool XCLIP>
__device__ __forceinline__ void bil_window(const uint16_t (*tile)[BIL_BX + 2 * BIL_R + 3], const unsigned char* s_mapx, const unsigned char* s_mapy, int x, int y, int bx, int by, int tx1,
                                           int ty1, unsigned int value, float& sum1, float& sum2)
{
    const float ss = 0.024691358f, sc = 0.000555556f;
    constexpr int D = BIL_R * 2 + 1;
    const float fv = (float)value;
    for (int cy = max(y - D / 2, 0); cy < ty1; ++cy) {
        const int ry = SHIFTED ? (int)s_mapy[cy - by + BIL_R + 1] : cy - by + BIL_R + 1;
        const float dy = (float)y - (float)cy, dy2 = dy * dy;
        float tv[D];
#pragma unroll
        for (int q = 0; q < D; q++) {
            const int col = x - D / 2 + q - bx + BIL_R + 1;   // (inside the tile for every q: the tile reaches BIL_R + 1 columns to the left and BIL_R to the right of the block)
            tv[q] = (float)tile[ry][SHIFTED ? (int)s_mapx[col] : col];
        }
#pragma unroll
        for (int q = 0; q < D; q += 2) {
            const int cx = x - D / 2 + q;
            const bool ok0 = !XCLIP || (cx >= 0 && cx < tx1), ok1 = q + 1 < D && (!XCLIP || (cx + 1 >= 0 && cx + 1 < tx1));
            ifx_v2f dxv, t2;
            dxv.x = (float)x - (float)cx; dxv.y = (float)x - (float)(cx + 1);
            t2.x = tv[q]; t2.y = q + 1 < D ? tv[q + 1] : 0.f;
            const ifx_v2f space2 = dxv * dxv + dy2;
            const ifx_v2f dv = fv - t2;
            const ifx_v2f color2 = dv * dv;
            const ifx_v2f wgt = ifx_expf2_nonpos(-(space2 * ss + color2 * sc));
            if (ok0) { sum1 += t2.x * wgt.x; sum2 += wgt.x; }
            if (ok1) { sum1 += t2.y * wgt.y; sum2 += wgt.y; }
        }
    }
}
__global__ __launch_bounds__(BIL_BX* BIL_BY) void k_bilateral_metric(const uint16_t* __restrict__ in, uint16_t* __restrict__ filt,
                                                                      float* __restrict__ dm, float* __restrict__ dmf, int w, int h, float maxD)
{
    // (one more row and column on the low side than the window: a tap of the shader samples the CORNER of its texel and may read the texel before it)
    __shared__ uint16_t tile[BIL_BY + 2 * BIL_R + 1][BIL_BX + 2 * BIL_R + 3];
    // the texel a tap reads (depth_bilateral.frag:59-61: texture(gSampler, vec2(float(cx) / cols, float(cy) / rows)) with GL_NEAREST = floor(u * size)): cx, or cx - 1 where the
    // f32 quotient times the size falls just below cx (oracle/orc_track.c bilateral_tap: no column of a 640-wide image, seven rows of a 480-high one) -- as tile coordinates
    __shared__ unsigned char s_mapx[BIL_BX + 2 * BIL_R + 1], s_mapy[BIL_BY + 2 * BIL_R + 1];
    __shared__ int s_shifted;
    const int bx = blockIdx.x * BIL_BX, by = blockIdx.y * BIL_BY;
    const int tid = threadIdx.y * BIL_BX + threadIdx.x;
    const int TW = BIL_BX + 2 * BIL_R + 1, TH = BIL_BY + 2 * BIL_R + 1;
    if (tid == 0) s_shifted = 0;
    __syncthreads();
    for (int i = tid; i < TW * TH; i += BIL_BX * BIL_BY) {
        int ty = i / TW, tx = i - ty * TW;
        int gx = bx + tx - BIL_R - 1, gy = by + ty - BIL_R - 1;
        uint16_t v = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) v = in[gy * w + gx];
        tile[ty][tx] = v;
    }
    if (tid < TW) {
        const int gx = bx + tid - BIL_R - 1;
        int t = tid;
        if (gx >= 0 && gx < w) t = max(min(max((int)floorf(((float)gx / (float)w) * (float)w), 0), w - 1) - (bx - BIL_R - 1), 0);
        s_mapx[tid] = (unsigned char)t;
        if (t != tid) s_shifted = 1;
    }
    if (tid >= 64 && tid - 64 < TH) {
        const int k = tid - 64, gy = by + k - BIL_R - 1;
        int t = k;
        if (gy >= 0 && gy < h) t = max(min(max((int)floorf(((float)gy / (float)h) * (float)h), 0), h - 1) - (by - BIL_R - 1), 0);
        s_mapy[k] = (unsigned char)t;
        if (t != k) s_shifted = 1;
    }
    __syncthreads();
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= w || y >= h) return;
    const unsigned int maxv = (unsigned int)(maxD * 1000.0f);
    unsigned int value = tile[threadIdx.y + BIL_R + 1][threadIdx.x + BIL_R + 1];
    unsigned int outv = 0;
    if (!(value > maxv || value < 300u)) {
        const float ss = 0.024691358f, sc = 0.000555556f;
        const int D = BIL_R * 2 + 1;
        int tx1 = min(x - D / 2 + D, w), ty1 = min(y - D / 2 + D, h);
        float sum1 = 0, sum2 = 0;
        // Block-uniform cases as template parameters (no branch inside the tap loop): SHIFTED -- a column or row of this tile whose corner tap reads the texel before it;
        // XCLIP -- the block touches the left or right image border, where a window is clipped (taps outside the image are skipped, not read as zero).
        const bool shifted = s_shifted != 0, xclip = bx < BIL_R || bx + BIL_BX + BIL_R > w;
        if (shifted) { if (xclip) bil_window<true, true>(tile, s_mapx, s_mapy, x, y, bx, by, tx1, ty1, value, sum1, sum2); else bil_window<true, false>(tile, s_mapx, s_mapy, x, y, bx, by, tx1, ty1, value, sum1, sum2); }
        else { if (xclip) bil_window<false, true>(tile, s_mapx, s_mapy, x, y, bx, by, tx1, ty1, value, sum1, sum2); else bil_window<false, false>(tile, s_mapx, s_mapy, x, y, bx, by, tx1, ty1, value, sum1, sum2); }
        outv = (unsigned int)roundf(sum1 / sum2);
        outv &= 0xFFFFu;
    }
    filt[y * w + x] = (uint16_t)outv;
    dm[y * w + x] = (value > maxv || value < 300u) ? 0.0f : (float)value / 1000.0f;
    dmf[y * w + x] = (outv > maxv || outv < 300u) ? 0.0f : (float)outv / 1000.0f;
}

int ifx_preprocess(ifx* h)
{
    dim3 block(BIL_BX, BIL_BY), grid(cdiv(h->w, BIL_BX), cdiv(h->h, BIL_BY));
    LAUNCH(h, "bilateral_metric", grid, block, k_bilateral_metric, h->depth_raw, h->depth_filt, h->dm, h->dmf, h->w, h->h, h->cfg.depth_cut);
    return IFX_OK;
}

// ======================================================================= pyramid kernels (a3)


// computeVmapKernel + computeNmapKernel fused (EF/Cuda/cudafuncs.cu:109-133,151-188): the three
// vertices a normal needs are recomputed from the depth image instead of re-read from the vmap.
__device__ inline bool vert_from_depth(const uint16_t* depth, int w, int u, int v, float fx_inv, float fy_inv, float cx, float cy, float cutoff, v3& o)
{
    float z = depth[v * w + u] / 1000.f;
    if (z != 0 && z < cutoff) {
        o = v3m(z * (u - cx) * fx_inv, z * (v - cy) * fy_inv, z);
        return true;
    }
    return false;
}

// bgr2IntensityKernel, EF/Cuda/cudafuncs.cu:550-563 (weights applied to R,G,B order as the reference does)
__global__ void k_intensity(const uint8_t* __restrict__ src, int stride, int n, uint8_t* __restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* s = src + (size_t)i * stride;
    int value = (int)((float)s[0] * 0.114f + (float)s[1] * 0.299f + (float)s[2] * 0.587f);
    dst[i] = (uint8_t)value;
}

__constant__ float c_gauss25[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};


// pyrDownKernelIntensityGauss, EF/Cuda/cudafuncs.cu:470-500
__global__ void k_pyrdown_gauss_u8(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst)
{
    int dw = sw / 2, dh = sh / 2;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int D = 5;
    int tx = min(2 * x - D / 2 + D, sw - 1), ty = min(2 * y - D / 2 + D, sh - 1);
    float sum = 0;
    int count = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            int s = src[cy * sw + cx];
            if (s > 0) {
                float g = c_gauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                sum += s * g;
                count += (int)g;
            }
        }
    dst[y * dw + x] = count ? (uint8_t)f2i_rz(sum / (float)count) : (uint8_t)0;
}







// ---- frame side in 4 launches (was 12): intensity of level 0; one launch per coarser level for BOTH pyr-downs (depth: the
// bilateral-like pyrDownGaussKernel, intensity: the 5x5 Gaussian); one launch for the vertex/normal maps and the Sobel
// gradients of all three levels (each needs its level's complete depth / intensity image, nothing of another level).
__global__ void k_frame_down(const uint16_t* __restrict__ dsrc, const uint8_t* __restrict__ isrc, int sw, int sh, uint16_t* __restrict__ ddst, uint8_t* __restrict__ idst)
{
    const int dw = sw / 2, dh = sh / 2;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int D = 5;
    {   // pyrDownGaussKernel, EF/Cuda/cudafuncs.cu:57-94
        const float sigma_color = 30;
        const float weights[3] = {0.375f, 0.25f, 0.0625f};
        int center = dsrc[(2 * y) * sw + 2 * x];
        int x_mi = max(0, 2 * x - D / 2) - 2 * x, y_mi = max(0, 2 * y - D / 2) - 2 * y;
        int x_ma = min(sw, 2 * x - D / 2 + D) - 2 * x, y_ma = min(sh, 2 * y - D / 2 + D) - 2 * y;
        float sum = 0, wall = 0;
        for (int yi = y_mi; yi < y_ma; ++yi)
            for (int xi = x_mi; xi < x_ma; ++xi) {
                int val = dsrc[(2 * y + yi) * sw + 2 * x + xi];
                if ((float)abs(val - center) < 3 * sigma_color) {
                    sum += val * weights[abs(xi)] * weights[abs(yi)];
                    wall += weights[abs(xi)] * weights[abs(yi)];
                }
            }
        ddst[y * dw + x] = (uint16_t)(int)(sum / wall);
    }
    {   // pyrDownKernelIntensityGauss, EF/Cuda/cudafuncs.cu:470-500
        int tx = min(2 * x - D / 2 + D, sw - 1), ty = min(2 * y - D / 2 + D, sh - 1);
        float sum = 0;
        int count = 0;
        for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
            for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
                int sv = isrc[cy * sw + cx];
                if (sv > 0) {
                    float g = c_gauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                    sum += sv * g;
                    count += (int)g;
                }
            }
        idst[y * dw + x] = count ? (uint8_t)f2i_rz(sum / (float)count) : (uint8_t)0;
    }
}
struct FrameLevel { const uint16_t* depth; const uint8_t* img; float *vmap, *nmap; int16_t *dx, *dy; int w, h, tiles_x, first_block; float fx_inv, fy_inv, cx, cy; };
struct FrameLevels { FrameLevel l[IFX_NUM_PYRS]; float cutoff; };
__global__ void __launch_bounds__(256) k_frame_maps(FrameLevels a)
{
    int lv = 0;
#pragma unroll
    for (int q = 1; q < IFX_NUM_PYRS; q++) if ((int)blockIdx.x >= a.l[q].first_block) lv = q;
    const FrameLevel L = a.l[lv];
    const int b = blockIdx.x - L.first_block, u = (b % L.tiles_x) * 32 + (threadIdx.x & 31), v = (b / L.tiles_x) * 8 + (threadIdx.x >> 5);
    const int w = L.w, h = L.h;
    if (u >= w || v >= h) return;
    {   // computeVmapKernel + computeNmapKernel, EF/Cuda/cudafuncs.cu:109-133,151-188
        const float qn = qnan_f();
        v3 v00;
        bool ok00 = vert_from_depth(L.depth, w, u, v, L.fx_inv, L.fy_inv, L.cx, L.cy, a.cutoff, v00);
        L.vmap[v * w + u] = ok00 ? v00.x : qn;
        L.vmap[(v + h) * w + u] = ok00 ? v00.y : qn;
        L.vmap[(v + 2 * h) * w + u] = ok00 ? v00.z : qn;
        v3 r = v3m(qn, qn, qn);
        if (!(u == w - 1 || v == h - 1) && ok00) {
            v3 v01, v10;
            bool ok01 = vert_from_depth(L.depth, w, u + 1, v, L.fx_inv, L.fy_inv, L.cx, L.cy, a.cutoff, v01);
            bool ok10 = vert_from_depth(L.depth, w, u, v + 1, L.fx_inv, L.fy_inv, L.cx, L.cy, a.cutoff, v10);
            if (ok01 && ok10) r = normalized(cross(v01 - v00, v10 - v00));
        }
        L.nmap[v * w + u] = r.x;
        L.nmap[(v + h) * w + u] = r.y;
        L.nmap[(v + 2 * h) * w + u] = r.z;
    }
    {   // applyKernel (Sobel), EF/Cuda/cudafuncs.cu:583-607
        const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
        const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
        float dxVal = 0, dyVal = 0;
        int k = 8;
        for (int j = max(v - 1, 0); j <= min(v + 1, h - 1); j++)
            for (int i = max(u - 1, 0); i <= min(u + 1, w - 1); i++) {
                dxVal += (float)L.img[j * w + i] * gsx[k];
                dyVal += (float)L.img[j * w + i] * gsy[k];
                --k;
            }
        L.dx[v * w + u] = (int16_t)dxVal;
        L.dy[v * w + u] = (int16_t)dyVal;
    }
}

// ---- model side in one launch per pyramid level.  Level 0: copyMaps (EF/Cuda/cudafuncs.cu:270-310) + verticesToDepth (:526-537) + intensity,
// the global transform (tranformMaps :206-248) and the point cloud of the photometric step (projectPoints :641-659) are all
// per-pixel, so they chain through registers.  Level i > 0: the 2x2 resize of both maps, the two 5x5 Gaussian
// pyr-downs (depth, intensity), then transform + cloud of the SAME output pixel.  16 launches -> 3; a dependent
// launch costs ~4.7 us on this GPU whatever its size.
struct ModelOut {
    float *vcam, *ncam, *depth;     // camera-frame maps (input of the next level) and model depth of this level
    uint8_t* img;
    float *vprev, *nprev, *cloud;   // global-frame maps for ICP, point cloud for the RGB step (nullptr: skipped)
    float invFx, invFy, cx, cy;
};
__device__ __forceinline__ void model_tail(const DevState* __restrict__ st, int x, int y, int w, int h, v3 vs, v3 ns, float z, const ModelOut& o)
{
    if (o.vprev) {   // nullptr: the maps stay in the camera frame (frame side of the model-to-model tracker)
        const float* P = st->pose;
        const float qn = qnan_f();
        v3 vd = v3m(qn, qn, qn);
        if (!(vs.x != vs.x)) vd = xf_dir(P, vs) + v3m(P[3], P[7], P[11]);
        o.vprev[y * w + x] = vd.x; o.vprev[(y + h) * w + x] = vd.y; o.vprev[(y + 2 * h) * w + x] = vd.z;
        v3 nd = v3m(qn, qn, qn);
        if (!(ns.x != ns.x)) nd = xf_dir(P, ns);
        o.nprev[y * w + x] = nd.x; o.nprev[(y + h) * w + x] = nd.y; o.nprev[(y + 2 * h) * w + x] = nd.z;
    }
    if (o.cloud) {
        o.cloud[(y * w + x) * 3 + 0] = (float)((x - o.cx) * z * o.invFx);
        o.cloud[(y * w + x) * 3 + 1] = (float)((y - o.cy) * z * o.invFy);
        o.cloud[(y * w + x) * 3 + 2] = z;
    }
}
__global__ void k_model_l0(const DevState* __restrict__ st, const float* __restrict__ pv, const float* __restrict__ pn, const uint8_t* __restrict__ pi, const float* __restrict__ fv,
                           const float* __restrict__ fn, const uint8_t* __restrict__ fi, int w, int h, float cutoff, ModelOut o)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const bool fill = fv && !st->dense_enough;
    const float4 v = reinterpret_cast<const float4*>(fill ? fv : pv)[y * w + x];
    const float4 n = reinterpret_cast<const float4*>(fill ? fn : pn)[y * w + x];
    const uint8_t* s = (fill ? fi : pi) + (size_t)(y * w + x) * 4;
    const float qn = qnan_f();
    bool ok = !(v.z == 0);
    v3 vs = v3m(ok ? v.x : qn, ok ? v.y : qn, ok ? v.z : qn), ns = v3m(ok ? n.x : qn, ok ? n.y : qn, ok ? n.z : qn);
    o.vcam[y * w + x] = vs.x; o.vcam[(y + h) * w + x] = vs.y; o.vcam[(y + 2 * h) * w + x] = vs.z;
    o.ncam[y * w + x] = ns.x; o.ncam[(y + h) * w + x] = ns.y; o.ncam[(y + 2 * h) * w + x] = ns.z;
    const float z = (v.z > cutoff || v.z <= 0) ? qn : v.z;
    o.depth[y * w + x] = z;
    o.img[y * w + x] = (uint8_t)(int)((float)s[0] * 0.114f + (float)s[1] * 0.299f + (float)s[2] * 0.587f);
    model_tail(st, x, y, w, h, vs, ns, z, o);
}
// The start of the tracker run (gn_begin_dev: one thread's worth of work that only needs the pose and the SO(3) result) can ride on the last launch
// of the model side instead of being a launch of its own (k_track_gn_begin, 4.8 us): `gb.enabled` on the frame tracker's tracked-ahead path.
struct GnBegin { DevState* st; const DevState* ss; int so3; float fx, fy, cx, cy; int keep_last, enabled, persist; };
__device__ void gn_begin_dev(DevState* st, const DevState* __restrict__ ss, int so3, float fx, float fy, float cx, float cy, int keep_last);
// hand-off words of the persistent level kernel (option gn_persist): both parities of the accumulator rows and residual totals, the barrier words
__device__ __forceinline__ void gn_persist_reset(DevState* st, int lt, int nt)
{
    for (int k = lt; k < 2 * 2 * IFX_ACC_REPL * IFX_ACC_STRIDE; k += nt) st->gn_acc2[k] = 0.0;
    if (lt < 32) st->gn_res2[lt] = 0;
    for (int k = lt; k < 4 * 32 * 16; k += nt) st->gn_bar[k] = 0u;
    if (lt == 0) { st->gn_abort = 0; st->gn_done_seq = 0; }
}
__global__ void k_model_down(const DevState* __restrict__ st, const float* __restrict__ vin, const float* __restrict__ nin, const float* __restrict__ din, const uint8_t* __restrict__ iin,
                             int sw, int sh, ModelOut o, GnBegin gb)
{
    if (gb.enabled && blockIdx.x == 0 && blockIdx.y == 0) {   // (writes tracker state only; this launch reads the pose and the dense flag)
        const int lt = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
        if (gb.persist) gn_persist_reset(gb.st, lt, nt);
        if (lt == 0) gn_begin_dev(gb.st, gb.ss, gb.so3, gb.fx, gb.fy, gb.cx, gb.cy, gb.keep_last);
    }
    const int dw = sw / 2, dh = sh / 2;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const float qn = qnan_f();
    const int xs = x * 2, ys = y * 2;
    v3 res[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {   // resizeMapKernel<normalize>, EF/Cuda/cudafuncs.cu:365-416
        const float* in = m ? nin : vin;
        float* out = m ? o.ncam : o.vcam;
        float x00 = in[ys * sw + xs], x01 = in[ys * sw + xs + 1], x10 = in[(ys + 1) * sw + xs], x11 = in[(ys + 1) * sw + xs + 1];
        v3 n = v3m(qn, qn, qn);
        if (!((x00 != x00) || (x01 != x01) || (x10 != x10) || (x11 != x11))) {
            n.x = (x00 + x01 + x10 + x11) / 4;
            const float* py = in + sh * sw;
            n.y = (py[ys * sw + xs] + py[ys * sw + xs + 1] + py[(ys + 1) * sw + xs] + py[(ys + 1) * sw + xs + 1]) / 4;
            const float* pz = in + 2 * sh * sw;
            n.z = (pz[ys * sw + xs] + pz[ys * sw + xs + 1] + pz[(ys + 1) * sw + xs] + pz[(ys + 1) * sw + xs + 1]) / 4;
            if (m) n = normalized(n);
        }
        out[y * dw + x] = n.x; out[(y + dh) * dw + x] = n.y; out[(y + 2 * dh) * dw + x] = n.z;
        res[m] = n;
    }
    const int D = 5;
    const int tx = min(2 * x - D / 2 + D, sw - 1), ty = min(2 * y - D / 2 + D, sh - 1);
    float sumf = 0, sumi = 0;
    int cntf = 0, cnti = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const float g = c_gauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
            const float sf = din[cy * sw + cx];       // pyrDownKernelGaussF :332-363
            if (!(sf != sf)) { sumf += sf * g; cntf += (int)g; }
            const int si = iin[cy * sw + cx];         // pyrDownKernelIntensityGauss :470-500
            if (si > 0) { sumi += si * g; cnti += (int)g; }
        }
    const float z = (float)(sumf / (float)cntf);
    o.depth[y * dw + x] = z;
    o.img[y * dw + x] = cnti ? (uint8_t)f2i_rz(sumi / (float)cnti) : (uint8_t)0;
    model_tail(st, x, y, dw, dh, res[0], res[1], z, o);
}

// ---- the model side of the frame tracker in ONE launch: levels 0, 1 and 2 of k_model_l0 / k_model_down for an image whose width is a multiple
// of 32 and height a multiple of 16.  A block owns an 8 x 4 tile of level 2, i.e. 16 x 8 of level 1 and 32 x 16 of level 0; what the two 5 x 5
// Gaussian pyr-downs read beyond the tile (level-1 region 19 x 11, level-0 region 41 x 25: depth and intensity only) is recomputed into LDS
// from the prediction images, the 2 x 2 map resizes chain through LDS.  Same per-pixel arithmetic, same
// clipped windows: the pyramids are bit-identical to the three launches (tests/test_gpu_parity.py::test_tracker_gputest_pair compares every
// buffer with the oracle).  Three dependent launches (6.7 + 9.8 + 9.8 us) become one.
#define MP_R0W 41
#define MP_R0H 25
#define MP_R1W 19
#define MP_R1H 11
template <typename FD, typename FI>
__device__ __forceinline__ void pyr_gauss5(int x, int y, int sw, int sh, FD depth_at, FI img_at, float& z, uint8_t& lum)
{
    const int D = 5;
    const int tx = min(2 * x - D / 2 + D, sw - 1), ty = min(2 * y - D / 2 + D, sh - 1);
    float sumf = 0, sumi = 0;
    int cntf = 0, cnti = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const float g = c_gauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
            const float sf = depth_at(cx, cy);      // pyrDownKernelGaussF :332-363
            if (!(sf != sf)) { sumf += sf * g; cntf += (int)g; }
            const int si = img_at(cx, cy);          // pyrDownKernelIntensityGauss :470-500
            if (si > 0) { sumi += si * g; cnti += (int)g; }
        }
    z = (float)(sumf / (float)cntf);
    lum = cnti ? (uint8_t)f2i_rz(sumi / (float)cnti) : (uint8_t)0;
}
__device__ __forceinline__ v3 resize4(float a00, float a01, float a10, float a11, float b00, float b01, float b10, float b11, float c00, float c01, float c10, float c11, bool norm_it)
{
    const float qn = qnan_f();
    v3 n = v3m(qn, qn, qn);
    if (!((a00 != a00) || (a01 != a01) || (a10 != a10) || (a11 != a11))) {   // resizeMapKernel<normalize>, EF/Cuda/cudafuncs.cu:365-416 (the test looks at the x plane only)
        n.x = (a00 + a01 + a10 + a11) / 4;
        n.y = (b00 + b01 + b10 + b11) / 4;
        n.z = (c00 + c01 + c10 + c11) / 4;
        if (norm_it) n = normalized(n);
    }
    return n;
}
struct ModelOut3 { ModelOut l[3]; };
#ifdef IFX_EXPERIMENTS
__global__ __launch_bounds__(256) void k_model_pyr3(const DevState* __restrict__ st, const float* __restrict__ pv, const float* __restrict__ pn, const uint8_t* __restrict__ pi,
                                                    const float* __restrict__ fv, const float* __restrict__ fn, const uint8_t* __restrict__ fi, int w, int h, float cutoff, ModelOut3 o)
{
    __shared__ float s_d0[MP_R0H][MP_R0W];
    __shared__ uint8_t s_i0[MP_R0H][MP_R0W];
    __shared__ float s_d1[MP_R1H][MP_R1W];
    __shared__ uint8_t s_i1[MP_R1H][MP_R1W];
    __shared__ float s_v0[6][16][33];  // level-0 vertex (0..2) and normal (3..5) of the tile (padded rows)
    __shared__ float s_v1[6][8][16];   // level-1 vertex (0..2) and normal (3..5) of the tile
    const int tid = threadIdx.x;
    const int w1 = w / 2, h1 = h / 2, w2 = w / 4, h2 = h / 4;
    const int x2_0 = blockIdx.x * 8, y2_0 = blockIdx.y * 4, x1_0 = 2 * x2_0, y1_0 = 2 * y2_0, x0_0 = 2 * x1_0, y0_0 = 2 * y1_0;
    const int r1x = x1_0 - 2, r1y = y1_0 - 2;   // origin of the level-1 region
    const int r0x = x0_0 - 6, r0y = y0_0 - 6;   // origin of the level-0 region
    const bool fill = fv && !st->dense_enough;
    const float* sv = fill ? fv : pv;
    const float* sn = fill ? fn : pn;
    const uint8_t* si = fill ? fi : pi;
    const float qn = qnan_f();
    // ---- level 0, depth and intensity of the region (k_model_l0's z and luminance)
    for (int t = tid; t < MP_R0W * MP_R0H; t += 256) {
        const int ly = t / MP_R0W, lx = t - ly * MP_R0W, x = r0x + lx, y = r0y + ly;
        float z = qn;
        uint8_t lum = 0;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const float vz = sv[(size_t)(y * w + x) * 4 + 2];
            const uint8_t* s = si + (size_t)(y * w + x) * 4;
            z = (vz > cutoff || vz <= 0) ? qn : vz;
            lum = (uint8_t)(int)((float)s[0] * 0.114f + (float)s[1] * 0.299f + (float)s[2] * 0.587f);
        }
        s_d0[ly][lx] = z;
        s_i0[ly][lx] = lum;
    }
    __syncthreads();
    // ---- level 0 of the tile: two pixels per thread, row-major (coalesced loads and stores); vertex and normal also go to LDS for the 2 x 2 resize
    auto d0_at = [&](int cx, int cy) { return s_d0[cy - r0y][cx - r0x]; };
    auto i0_at = [&](int cx, int cy) { return (int)s_i0[cy - r0y][cx - r0x]; };
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int t = tid + u * 256, lx = t & 31, ly = t >> 5, x = x0_0 + lx, y = y0_0 + ly;
        const float4 v = reinterpret_cast<const float4*>(sv)[y * w + x];
        const float4 n = reinterpret_cast<const float4*>(sn)[y * w + x];
        const bool ok = !(v.z == 0);
        const v3 vs = v3m(ok ? v.x : qn, ok ? v.y : qn, ok ? v.z : qn), ns = v3m(ok ? n.x : qn, ok ? n.y : qn, ok ? n.z : qn);
        const ModelOut& o0 = o.l[0];
        o0.vcam[y * w + x] = vs.x; o0.vcam[(y + h) * w + x] = vs.y; o0.vcam[(y + 2 * h) * w + x] = vs.z;
        o0.ncam[y * w + x] = ns.x; o0.ncam[(y + h) * w + x] = ns.y; o0.ncam[(y + 2 * h) * w + x] = ns.z;
        const float z = s_d0[y - r0y][x - r0x];
        o0.depth[y * w + x] = z;
        o0.img[y * w + x] = s_i0[y - r0y][x - r0x];
        model_tail(st, x, y, w, h, vs, ns, z, o0);
        s_v0[0][ly][lx] = vs.x; s_v0[1][ly][lx] = vs.y; s_v0[2][ly][lx] = vs.z;
        s_v0[3][ly][lx] = ns.x; s_v0[4][ly][lx] = ns.y; s_v0[5][ly][lx] = ns.z;
    }
    __syncthreads();
    // ---- threads 0..127: level-1 pixel (lx1, ly1) of the tile; 128..255: level-1 depth / intensity of the halo
    if (tid < 128) {
        const int lx1 = tid & 15, ly1 = tid >> 4, x1 = x1_0 + lx1, y1 = y1_0 + ly1;
        const int ax = 2 * lx1, ay = 2 * ly1;
        const v3 v1 = resize4(s_v0[0][ay][ax], s_v0[0][ay][ax + 1], s_v0[0][ay + 1][ax], s_v0[0][ay + 1][ax + 1], s_v0[1][ay][ax], s_v0[1][ay][ax + 1], s_v0[1][ay + 1][ax], s_v0[1][ay + 1][ax + 1],
                              s_v0[2][ay][ax], s_v0[2][ay][ax + 1], s_v0[2][ay + 1][ax], s_v0[2][ay + 1][ax + 1], false);
        const v3 n1 = resize4(s_v0[3][ay][ax], s_v0[3][ay][ax + 1], s_v0[3][ay + 1][ax], s_v0[3][ay + 1][ax + 1], s_v0[4][ay][ax], s_v0[4][ay][ax + 1], s_v0[4][ay + 1][ax], s_v0[4][ay + 1][ax + 1],
                              s_v0[5][ay][ax], s_v0[5][ay][ax + 1], s_v0[5][ay + 1][ax], s_v0[5][ay + 1][ax + 1], true);
        float z1;
        uint8_t l1;
        pyr_gauss5(x1, y1, w, h, d0_at, i0_at, z1, l1);
        const ModelOut& o1 = o.l[1];
        o1.vcam[y1 * w1 + x1] = v1.x; o1.vcam[(y1 + h1) * w1 + x1] = v1.y; o1.vcam[(y1 + 2 * h1) * w1 + x1] = v1.z;
        o1.ncam[y1 * w1 + x1] = n1.x; o1.ncam[(y1 + h1) * w1 + x1] = n1.y; o1.ncam[(y1 + 2 * h1) * w1 + x1] = n1.z;
        o1.depth[y1 * w1 + x1] = z1;
        o1.img[y1 * w1 + x1] = l1;
        model_tail(st, x1, y1, w1, h1, v1, n1, z1, o1);
        s_d1[y1 - r1y][x1 - r1x] = z1;
        s_i1[y1 - r1y][x1 - r1x] = l1;
        s_v1[0][ly1][lx1] = v1.x; s_v1[1][ly1][lx1] = v1.y; s_v1[2][ly1][lx1] = v1.z;
        s_v1[3][ly1][lx1] = n1.x; s_v1[4][ly1][lx1] = n1.y; s_v1[5][ly1][lx1] = n1.z;
    } else {
        // the 19 x 11 level-1 region minus the 16 x 8 tile: 81 pixels
        for (int t = tid - 128; t < MP_R1W * MP_R1H; t += 128) {
            const int ly = t / MP_R1W, lx = t - ly * MP_R1W, x1 = r1x + lx, y1 = r1y + ly;
            if (lx >= 2 && lx < 18 && ly >= 2 && ly < 10) continue;   // the tile itself (threads 0..127)
            float z1 = qn;
            uint8_t l1 = 0;
            if (x1 >= 0 && x1 < w1 && y1 >= 0 && y1 < h1) pyr_gauss5(x1, y1, w, h, d0_at, i0_at, z1, l1);
            s_d1[ly][lx] = z1;
            s_i1[ly][lx] = l1;
        }
    }
    __syncthreads();
    // ---- level 2: 32 pixels
    if (tid < 32) {
        const int lx2 = tid & 7, ly2 = tid >> 3, x2 = x2_0 + lx2, y2 = y2_0 + ly2;
        const int ax = 2 * lx2, ay = 2 * ly2;
        const v3 v2 = resize4(s_v1[0][ay][ax], s_v1[0][ay][ax + 1], s_v1[0][ay + 1][ax], s_v1[0][ay + 1][ax + 1], s_v1[1][ay][ax], s_v1[1][ay][ax + 1], s_v1[1][ay + 1][ax], s_v1[1][ay + 1][ax + 1],
                              s_v1[2][ay][ax], s_v1[2][ay][ax + 1], s_v1[2][ay + 1][ax], s_v1[2][ay + 1][ax + 1], false);
        const v3 n2 = resize4(s_v1[3][ay][ax], s_v1[3][ay][ax + 1], s_v1[3][ay + 1][ax], s_v1[3][ay + 1][ax + 1], s_v1[4][ay][ax], s_v1[4][ay][ax + 1], s_v1[4][ay + 1][ax], s_v1[4][ay + 1][ax + 1],
                              s_v1[5][ay][ax], s_v1[5][ay][ax + 1], s_v1[5][ay + 1][ax], s_v1[5][ay + 1][ax + 1], true);
        float z2;
        uint8_t l2;
        pyr_gauss5(x2, y2, w1, h1, [&](int cx, int cy) { return s_d1[cy - r1y][cx - r1x]; }, [&](int cx, int cy) { return (int)s_i1[cy - r1y][cx - r1x]; }, z2, l2);
        const ModelOut& o2 = o.l[2];
        o2.vcam[y2 * w2 + x2] = v2.x; o2.vcam[(y2 + h2) * w2 + x2] = v2.y; o2.vcam[(y2 + 2 * h2) * w2 + x2] = v2.z;
        o2.ncam[y2 * w2 + x2] = n2.x; o2.ncam[(y2 + h2) * w2 + x2] = n2.y; o2.ncam[(y2 + 2 * h2) * w2 + x2] = n2.z;
        o2.depth[y2 * w2 + x2] = z2;
        o2.img[y2 * w2 + x2] = l2;
        model_tail(st, x2, y2, w2, h2, v2, n2, z2, o2);
    }
}
#endif   // IFX_EXPERIMENTS

// ======================================================================= reductions (a4-a7)

#define RED_THREADS 256
#define RED_WAVES (RED_THREADS / 64)

// Exact block sum of NV doubles per thread (grid-valued terms, see ifx_dev.h): quads by DPP, 64 quad sums per value through
// LDS, eight partial sums per value, then ONE f64 atomic add per value and block into replica `rep` of the global accumulator
// row.  Every addition is exact, so neither the tree shape nor the arrival order of the atomics matters.
__device__ __forceinline__ double dpp_quad_sum_d(double v)
{
    {   // quad_perm [1,0,3,2]
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
        v += __hiloint2double(hi, lo);
    }
    {   // quad_perm [2,3,0,1]
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xF, 0xF, true);
        v += __hiloint2double(hi, lo);
    }
    return v;
}
template <int NV>
__device__ __forceinline__ void block_sum_exact(const double* acc, double* __restrict__ gacc, int rep)
{
    __shared__ double lds[NV][RED_THREADS / 4 + 2];
    __shared__ double part[NV][8];
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const double q = dpp_quad_sum_d(acc[k]);
        if ((t & 3) == 0) lds[k][t >> 2] = q;
    }
    __syncthreads();
    if (t < NV * 8) {
        const int k = t >> 3, p = t & 7;
        double s = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) s += lds[k][p * 8 + ((j + p) & 7)];
        part[k][p] = s;
    }
    __syncthreads();
    if (t < NV) {
        double s = 0;
#pragma unroll
        for (int p = 0; p < 8; p++) s += part[t][p];
        if (s != 0.0) unsafeAtomicAdd(&gacc[rep * IFX_ACC_STRIDE + t], s);   // global_atomic_add_f64, no return
    }
}
// the accumulator rows of one quantity, read by the block that finishes last (agent-scope loads: the adds were performed at the memory side)
__device__ __forceinline__ double acc_total(const double* gacc, int k)
{
    double s = 0;
#pragma unroll
    for (int r = 0; r < IFX_ACC_REPL; r++) s += __hip_atomic_load(&gacc[r * IFX_ACC_STRIDE + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
}
__device__ __forceinline__ void acc_clear(double* gacc, int k)
{
#pragma unroll
    for (int r = 0; r < IFX_ACC_REPL; r++) __hip_atomic_store(&gacc[r * IFX_ACC_STRIDE + k], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Run-time guard of the exact-sum contract (ifx_dev.h): the sums are exact -- hence independent of the order the atomics arrive in, and equal to the oracle's -- while every
// PARTIAL sum stays below 2^53 grid units.  That is decided by the seven diagonal totals alone: a diagonal sum D_i = sum row_i^2 has non-negative terms, so its partial sums are
// below its total; and every partial sum of an off-diagonal entry is at most sum |row_i row_j| <= sqrt(D_i D_j) (Cauchy-Schwarz), whose limit 2^(E_i + E_j + 21) is the
// geometric mean of the two diagonal limits 2^(2 E_i + 21).  So: all seven diagonal totals below HALF their limit (the half pays for the rounding of the terms to the grid)
// => every addition of all 28 sums was exact.  Whoever reads the totals of an iteration checks its diagonal entry and counts a violation in DevState::range_exceeded
// (ifx_tracker_range_exceeded): 0 on every sequence of the test suite and of bench.py; a near-range frame of saturated edges drives it (tests/test_gpu_parity.py).
// (integer arithmetic on packed constants only: a runtime-indexed local array, or a switch over doubles, put every launch of the tracker on scratch memory)
__host__ __device__ constexpr unsigned long long range_pack7(const int (&e)[7], int bias)
{
    unsigned long long p = 0;
    for (int i = 0; i < 7; i++) p |= (unsigned long long)(unsigned int)(2 * e[i] + bias + 64) << (8 * i);   // biased by 64: a byte each
    return p;
}
__device__ __forceinline__ bool range_exceeded7(int kind, int k, double total)
{
    constexpr int EI[7] = IFX_E_ICP, ER[7] = IFX_E_RGB;
    constexpr unsigned long long PI = range_pack7(EI, 52 - IFX_EXACT_TERM_BITS), PR = range_pack7(ER, 52 - IFX_EXACT_TERM_BITS);
    constexpr unsigned int DIAG = (1u << 0) | (1u << 7) | (1u << 13) | (1u << 18) | (1u << 22) | (1u << 25) | (1u << 27);   // positions of the squares in the 29-vector
    if (k > 27 || !((DIAG >> k) & 1u)) return false;
    const int i = __popc(DIAG & ((1u << k) - 1u));
    const int ex = (int)(((kind ? PR : PI) >> (8 * i)) & 0xFFull) - 64;
    const double lim = __hiloint2double((1023 + ex) << 20, 0);   // 2^ex
    return !(total < lim);   // (a NaN total counts)
}
__device__ __forceinline__ bool range_exceeded_so3(int k, double total)
{
    constexpr int ES[4] = IFX_E_SO3;
    constexpr int B = 52 - IFX_SO3_TERM_BITS;
    constexpr unsigned int PS = (unsigned int)(2 * ES[0] + B + 64) | ((unsigned int)(2 * ES[1] + B + 64) << 8) | ((unsigned int)(2 * ES[2] + B + 64) << 16) | ((unsigned int)(2 * ES[3] + B + 64) << 24);
    constexpr unsigned int DIAG = (1u << 0) | (1u << 4) | (1u << 7) | (1u << 9);
    if (k > 9 || !((DIAG >> k) & 1u)) return false;
    const int i = __popc(DIAG & ((1u << k) - 1u));
    const int ex = (int)((PS >> (8 * i)) & 0xFFu) - 64;
    return !(total < __hiloint2double((1023 + ex) << 20, 0));
}
__device__ __forceinline__ void range_note(const DevState* st, bool icp_bad, bool rgb_bad)
{
    if (icp_bad || rgb_bad) atomicAdd(&const_cast<DevState*>(st)->range_exceeded, (icp_bad ? 1 : 0) + (rgb_bad ? 1 : 0));
}

// KIND 0: ICP row, 1: photometric row
template <int KIND>
__device__ __forceinline__ void products7(const float* row, bool found, double* acc)
{
    constexpr int E[2][7] = {IFX_E_ICP, IFX_E_RGB};
    int s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 7; j++) acc[s++] += ifx_quant(row[i] * row[j], ifx_magic(E[KIND][i] + E[KIND][j] - IFX_EXACT_TERM_BITS));
    acc[27] += ifx_quant(row[6] * row[6], ifx_magic(2 * E[KIND][6] - IFX_EXACT_TERM_BITS));
    acc[28] += found ? 1.0 : 0.0;
}

// Every reduction body handles IT pixels per thread in four straight-line stages -- coalesced
// loads of all IT pixels, projection, dependent gathers of all IT pixels, arithmetic -- so that a
// thread has all of its memory requests in flight together.  (The first version used a rolled
// grid-stride loop: 4-5 serial iterations of two dependent round trips each made even the 19 200-px
// level take 11 us.)  Pixel u of a thread is base + u * blockDim.x, i.e. coalesced across lanes.
// Pixels per thread and loop round.  One for the ICP / residual passes: with four (the first tuning) a level-0 launch at 640x480 took 16.0 us, with one
// 13.3 us -- the four-stage body already keeps a thread's requests in flight together, and the smaller register footprint lets more waves hide each
// other's gathers (1004 -> 1057 frames/s).  The photometric step, whose 8-byte records stream, keeps four.
#ifndef RED_IT
#define RED_IT 1
#endif
#ifndef RED_IT_RGB
#define RED_IT_RGB 4
#endif
#ifdef IFX_STAMPS
__device__ long long g_dbg2[8];
#define g_ts ts_local
__shared__ long long s_dbg_blk[4];
#endif

// ======================================================================= Gauss-Newton: the solve in the NEXT launch's prologue (option gn_prologue, default)
// Round 3's iteration was: launch A (ICP sums + residual pass) -> launch B (photometric sums; the block that draws the last ticket reads the 2 x 29 totals, solves the
// 6x6 system, stores the pose) -> launch A' (loads the pose) ...  The serial tail of B -- drain, ticket round trip, totals round trip, solve, stores -- and the pose loads of
// A' were all on the critical path of every iteration.  Here B just sums; EVERY block of A' fetches the totals of the previous iteration itself (the same 58 agent-scope
// loads the last block did), rebuilds the combined system and solves it on its lane 0 -- the same totals give the same bits in every block -- while the loads of its
// first pixels, which do not depend on the pose, are already in flight.  No ticket, no last block, no pose round trip through memory; the run's LAST iteration keeps the
// last-block form (it also ends the run: pose write-back, velocity weighting, view-list decision).  Sums, residual totals and the running increment are double-buffered
// by iteration parity (DevState::gnp_*): iteration j writes parity j & 1, launch B of iteration j + 1 clears it after every block of launch A of j + 1 has read it.
struct KInv { double i0, i4, i2, i5; };
struct GnPro {
    int k;                        // position of this launch's iteration in the run's two-launch tail; k >= 1: iteration k - 1 is solved in the prologue
    int icp, rgb;
    float icp_weight;
    float nfx, nfy, ncx, ncy;     // intrinsics of the level THIS launch runs at: the warp matrices the solve emits are for this launch's own residual pass
    KInv ki;
};
__device__ __forceinline__ void gn_serial_lane(const double* s_sys, int icp, double* RRt, const float* Rp, const float* tp, float nfx, float nfy, float ncx, float ncy, const KInv& ki,
                                               float* Rc, float* tc, float* krk, float* kt);
// returns the pose of the iteration about to run, in LDS: Rcurr 9, tcurr 3, krkinv 9, kt 3.  Call it with the block's first pose-independent loads already issued.
__device__ __forceinline__ const float* gn_prologue(const DevState* __restrict__ st, const GnPro& g, double* __restrict__ rrt_out)
{
    __shared__ double s_psys[27];
    __shared__ float s_ppose[24];
    const int prev = (g.k - 1) & 1;
    const double* const icp_acc = st->gnp_acc + (size_t)(prev * 2) * IFX_ACC_REPL * IFX_ACC_STRIDE;
    const double* const rgb_acc = icp_acc + IFX_ACC_REPL * IFX_ACC_STRIDE;
    if (threadIdx.x < 27) {
        const int k = threadIdx.x;
        const double ti = acc_total(icp_acc, k), tr = acc_total(rgb_acc, k);
        const float oi = g.icp ? (float)ti : 0.f, orr = g.rgb ? (float)tr : 0.f;   // rounded to f32 as the reference's reductions deliver them
        // lastA = A_rgb + w*w*A_icp, lastb = b_rgb + w*b_icp (EF/Utils/RGBDOdometry.cpp:547-565); column 6 of a row is its b entry
        const double wgt = g.icp_weight;
        const double wa = wgt * wgt, wb = wgt;
        const bool is_b = (k == 6) | (k == 12) | (k == 17) | (k == 21) | (k == 24) | (k == 26);
        const double vi = (double)oi, vr = (double)orr;
        s_psys[k] = (g.icp && g.rgb) ? vr + (is_b ? wb : wa) * vi : (g.icp ? vi : vr);
    }
    // the increment after iteration k - 2 (the run's start value for k == 1) and the run's constants: uniform addresses, fetched beside the totals
    const double* const rsrc = g.k == 1 ? st->resultRt : st->gnp_RRt[g.k & 1];
    double RRt[16];
    float Rp[9], tp[3];
#pragma unroll
    for (int k = 0; k < 12; k++) RRt[k] = rsrc[k];
    RRt[12] = 0.0; RRt[13] = 0.0; RRt[14] = 0.0; RRt[15] = 1.0;
#pragma unroll
    for (int k = 0; k < 9; k++) Rp[k] = st->Rprev[k];
#pragma unroll
    for (int k = 0; k < 3; k++) tp[k] = st->tprev[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        float Rc[9], tc[3], krk[9], kt[3];
#ifdef GN_SOLVE_TWICE   // measurement build (tools/ab.sh -b "ifx_track:-DGN_SOLVE_TWICE"): the serial part a second time, its result folded in as an exact zero -- what the solve costs the frame
        {
            double RRt2[16];
            float Rc2[9], tc2[3], krk2[9], kt2[3];
#pragma unroll
            for (int k = 0; k < 16; k++) RRt2[k] = RRt[k];
            gn_serial_lane(s_psys, g.icp, RRt2, Rp, tp, g.nfx, g.nfy, g.ncx, g.ncy, g.ki, Rc2, tc2, krk2, kt2);
            asm volatile("" : "+v"(Rc2[0]));
            if (Rc2[0] > 3.0e38f) s_psys[0] += 1.0;   // (never true; keeps the first run alive and in front of the second)
        }
#endif
        gn_serial_lane(s_psys, g.icp, RRt, Rp, tp, g.nfx, g.nfy, g.ncx, g.ncy, g.ki, Rc, tc, krk, kt);
#pragma unroll
        for (int k = 0; k < 9; k++) { s_ppose[k] = Rc[k]; s_ppose[12 + k] = krk[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { s_ppose[9 + k] = tc[k]; s_ppose[21 + k] = kt[k]; }
        if (rrt_out) {
#pragma unroll
            for (int k = 0; k < 12; k++) rrt_out[k] = RRt[k];
        }
    }
    __syncthreads();
    return s_ppose;
}

struct IcpArgs { float Rcurr[9], tcurr[3], Rprev_inv[9], tprev[3]; };
// ICPReduction, EF/Cuda/reduce.cu:257-411.  Rcurr/tcurr/Rprev_inv/tprev come from DevState (or from
// explicit arguments for the stage API when st == nullptr).
template <bool WT = false, bool FROM_STATE = false, bool PRO = false>
__device__ __forceinline__ void icp_body(int bid, int nblk, const DevState* __restrict__ st, const IcpArgs& ex, const float* __restrict__ vmap_curr,
                                         const float* __restrict__ nmap_curr, const float* __restrict__ vmap_prev, const float* __restrict__ nmap_prev, float fx,
                                         float fy, float cx, float cy, float distThres, float angleThres, int w, int h, double* __restrict__ gacc,
                                         const GnPro* pro = nullptr, double* __restrict__ rrt_out = nullptr)
{
    const int N = w * h;
    // One pixel per thread and round, software-pipelined: the coalesced loads of the NEXT round are issued right behind this round's gathers, so a
    // round costs one memory round trip instead of two (a level-0 block runs four rounds).  Issue order matters: vector loads return in order,
    // so the gathers go first and the prefetch rides behind them.
    const int stride = nblk * RED_THREADS;
    int i = bid * RED_THREADS + threadIdx.x;
    v3 vcurr, ncurr;
    {   // the first round's own-pixel loads do not depend on the pose: in flight before the pose is even known (PRO: under the prologue's solve)
        const int ii = i < N ? i : 0;
        vcurr = v3m(vmap_curr[ii], vmap_curr[ii + N], vmap_curr[ii + 2 * N]);
        ncurr = v3m(nmap_curr[ii], nmap_curr[ii + N], nmap_curr[ii + 2 * N]);
    }
    const bool from_state = FROM_STATE || st != nullptr;   // (FROM_STATE: known at compile time -- no per-element select, the loads leave in one batch)
    const float* Rc = from_state ? st->Rcurr : ex.Rcurr;
    const float* tcp = from_state ? st->tcurr : ex.tcurr;
    const float* Rpi = from_state ? st->Rprev_inv : ex.Rprev_inv;
    const float* tpp = from_state ? st->tprev : ex.tprev;
    if (PRO) { const float* sp = gn_prologue(st, *pro, rrt_out); Rc = sp; tcp = sp + 9; }   // the pose this iteration runs at: solved here, by every block alike
    float Rcurr[9], Rprev_inv[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rcurr[k] = Rc[k]; Rprev_inv[k] = Rpi[k]; }
    const v3 tc = v3m(tcp[0], tcp[1], tcp[2]), tp = v3m(tpp[0], tpp[1], tpp[2]);
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.0;
    while (i < N) {
        // stage 2: projection into the model frame
        const v3 vcurr_g = mulp(Rcurr, vcurr) + tc;
        const v3 vcurr_cp = mulp(Rprev_inv, vcurr_g - tp);
        const int ux = f2i_rn(vcurr_cp.x * fx / vcurr_cp.z + cx);
        const int uy = f2i_rn(vcurr_cp.y * fy / vcurr_cp.z + cy);
        const bool inb = !(vcurr.x != vcurr.x) && !(ux < 0 || uy < 0 || ux >= w || uy >= h || vcurr_cp.z < 0);
        const int j = inb ? uy * w + ux : 0;
        // stage 3: gathers
        const v3 vprev = v3m(vmap_prev[j], vmap_prev[j + N], vmap_prev[j + 2 * N]);
        const v3 nprev = v3m(nmap_prev[j], nmap_prev[j + N], nmap_prev[j + 2 * N]);
        // next round's coalesced loads (a thread in its last round re-reads its own pixel: no branch around the loads)
        const int inext = i + stride;
        const int ip = inext < N ? inext : i;
        const v3 vnext = v3m(vmap_curr[ip], vmap_curr[ip + N], vmap_curr[ip + 2 * N]);
        const v3 nnext = v3m(nmap_curr[ip], nmap_curr[ip + N], nmap_curr[ip + 2 * N]);
        // stage 4: row of the normal equations
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        bool found = false;
        if (inb) {
            v3 ncurr_g = mulp(Rcurr, ncurr);
            float dist = norm(vprev - vcurr_g);
            float sine = norm(cross(ncurr_g, nprev));
            found = (sine < angleThres && dist <= distThres && !(ncurr.x != ncurr.x) && !(nprev.x != nprev.x));
            if (found) {
                v3 s_cp = mulp(Rprev_inv, vcurr_g - tp);
                v3 d_cp = mulp(Rprev_inv, vprev - tp);
                v3 n_cp = mulp(Rprev_inv, nprev);
                v3 c = cross(s_cp, n_cp);
                row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
                row[3] = c.x; row[4] = c.y; row[5] = c.z;
                row[6] = dot(n_cp, s_cp - d_cp);
            }
        }
        products7<0>(row, found, acc);
        vcurr = vnext; ncurr = nnext; i = inext;
    }
    block_sum_exact<29>(acc, gacc, bid % IFX_ACC_REPL);
}
// The north star's "LDS-staged depth / normal tiles for the ICP reduction", as an option (ifx_set_option("icp_lds", 1); level 0 only):
// a block owns a 64 x 16 pixel tile, stages the model's vertex and normal maps of the tile plus an 8-pixel halo in LDS (80 x 32 x 24 B =
// 60 KB, coalesced row loads) and takes a correspondence from LDS when it falls inside, from global memory otherwise.  Same values, same
// exact sums: bit-identical results.  MEASURED: slower than the plain gathers -- a frame's motion is a few pixels, so
// neighbouring pixels already gather neighbouring model texels through L1 / L2, and the staging reads 2.5x the bytes the gathers touch.
#define LT_W 64
#define LT_H 16
#define LT_HALO 8
#define LT_SW (LT_W + 2 * LT_HALO)
#define LT_SH (LT_H + 2 * LT_HALO)
#ifdef IFX_EXPERIMENTS
__device__ __forceinline__ void icp_body_lds(int bid, const DevState* __restrict__ st, const float* __restrict__ vmap_curr, const float* __restrict__ nmap_curr,
                                             const float* __restrict__ vmap_prev, const float* __restrict__ nmap_prev, float fx, float fy, float cx, float cy,
                                             float distThres, float angleThres, int w, int h, double* __restrict__ gacc)
{
    __shared__ float s_m[6][LT_SH][LT_SW];
    float Rcurr[9], Rprev_inv[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rcurr[k] = st->Rcurr[k]; Rprev_inv[k] = st->Rprev_inv[k]; }
    const v3 tc = v3m(st->tcurr[0], st->tcurr[1], st->tcurr[2]), tp = v3m(st->tprev[0], st->tprev[1], st->tprev[2]);
    const int N = w * h, tiles_x = (w + LT_W - 1) / LT_W;
    const int x0 = (bid % tiles_x) * LT_W, y0 = (bid / tiles_x) * LT_H;
    for (int idx = threadIdx.x; idx < LT_SW * LT_SH; idx += RED_THREADS) {
        const int ly = idx / LT_SW, lx = idx - ly * LT_SW, gx = x0 - LT_HALO + lx, gy = y0 - LT_HALO + ly;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            const int g = gy * w + gx;
#pragma unroll
            for (int q = 0; q < 3; q++) { s_m[q][ly][lx] = vmap_prev[g + q * N]; s_m[3 + q][ly][lx] = nmap_prev[g + q * N]; }
        }
    }
    __syncthreads();
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.0;
    const int tx = threadIdx.x & (LT_W - 1), ty0 = threadIdx.x >> 6;   // 256 threads = 64 columns x 4 rows, 4 rounds of rows
#pragma unroll
    for (int u = 0; u < LT_H / 4; u++) {
        const int px = x0 + tx, py = y0 + ty0 + 4 * u;
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        bool found = false;
        if (px < w && py < h) {
            const int i = py * w + px;
            const v3 vcurr = v3m(vmap_curr[i], vmap_curr[i + N], vmap_curr[i + 2 * N]), ncurr = v3m(nmap_curr[i], nmap_curr[i + N], nmap_curr[i + 2 * N]);
            const v3 vcurr_g = mulp(Rcurr, vcurr) + tc;
            const v3 vcurr_cp = mulp(Rprev_inv, vcurr_g - tp);
            const int ux = f2i_rn(vcurr_cp.x * fx / vcurr_cp.z + cx), uy = f2i_rn(vcurr_cp.y * fy / vcurr_cp.z + cy);
            const bool inb = !(vcurr.x != vcurr.x) && !(ux < 0 || uy < 0 || ux >= w || uy >= h || vcurr_cp.z < 0);
            if (inb) {
                v3 vprev, nprev;
                const int lx = ux - x0 + LT_HALO, ly = uy - y0 + LT_HALO;
                if (lx >= 0 && lx < LT_SW && ly >= 0 && ly < LT_SH) {
                    vprev = v3m(s_m[0][ly][lx], s_m[1][ly][lx], s_m[2][ly][lx]);
                    nprev = v3m(s_m[3][ly][lx], s_m[4][ly][lx], s_m[5][ly][lx]);
                } else {
                    const int j = uy * w + ux;
                    vprev = v3m(vmap_prev[j], vmap_prev[j + N], vmap_prev[j + 2 * N]);
                    nprev = v3m(nmap_prev[j], nmap_prev[j + N], nmap_prev[j + 2 * N]);
                }
                const v3 ncurr_g = mulp(Rcurr, ncurr);
                const float dist = norm(vprev - vcurr_g), sine = norm(cross(ncurr_g, nprev));
                found = (sine < angleThres && dist <= distThres && !(ncurr.x != ncurr.x) && !(nprev.x != nprev.x));
                if (found) {
                    const v3 s_cp = mulp(Rprev_inv, vcurr_g - tp), d_cp = mulp(Rprev_inv, vprev - tp), n_cp = mulp(Rprev_inv, nprev), c = cross(s_cp, n_cp);
                    row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z; row[3] = c.x; row[4] = c.y; row[5] = c.z;
                    row[6] = dot(n_cp, s_cp - d_cp);
                }
            }
        }
        products7<0>(row, found, acc);
    }
    block_sum_exact<29>(acc, gacc, bid % IFX_ACC_REPL);
}
#endif   // IFX_EXPERIMENTS

__global__ __launch_bounds__(RED_THREADS) void k_icp(const DevState* __restrict__ st, IcpArgs ex, const float* __restrict__ vmap_curr,
                                                     const float* __restrict__ nmap_curr, const float* __restrict__ vmap_prev,
                                                     const float* __restrict__ nmap_prev, float fx, float fy, float cx, float cy, float distThres,
                                                     float angleThres, int w, int h, double* __restrict__ gacc)
{
    icp_body(blockIdx.x, gridDim.x, st, ex, vmap_curr, nmap_curr, vmap_prev, nmap_prev, fx, fy, cx, cy, distThres, angleThres, w, h, gacc);
}

// 8-byte correspondence record (the reference's DataTerm is 16 B, EF/Cuda/types.cuh:75-81: `one`
// is the pixel's own coordinate and `valid` is folded into zx >= 0)
struct Corres8 { short zx, zy; float diff; };

// RGBResidual, EF/Cuda/reduce.cu:739-863
struct ResArgs { float krkinv[9], kt[3]; };
template <bool FROM_STATE = false, bool PRO = false>
__device__ __forceinline__ void residual_body(int bid, int nblk, const DevState* __restrict__ st, const ResArgs& ex, float minScale, const int16_t* __restrict__ dIdx,
                                              const int16_t* __restrict__ dIdy, const float* __restrict__ lastDepth, const float* __restrict__ nextDepth,
                                              const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage, Corres8* __restrict__ corres,
                                              float maxDepthDelta, int w, int h, int* __restrict__ partials, int* __restrict__ res_total = nullptr,
                                              const GnPro* pro = nullptr, double* __restrict__ rrt_out = nullptr)
{
    const int border = 16;
    const int N = w * h;
    // stage 1 of a round: own-pixel tests (4x4 non-zero block, gradient gate) and depth -- nothing of it depends on the pose
    struct Own { bool cand[RED_IT]; int kidx[RED_IT]; float d1[RED_IT]; uint8_t ni[RED_IT]; };
    auto own_pixels = [&](int base, Own& o) {
#pragma unroll
        for (int u = 0; u < RED_IT; u++) {
            int k = base + u * RED_THREADS;
            o.kidx[u] = k;
            bool in = k < N;
            int kk2 = in ? k : 0;
            int i = kk2 / w, j0 = kk2 - i * w;
            bool ok = in && i >= border && i < h - border && j0 >= border && j0 < w - border && j0 < w - 5 && i < h - 1;
            // inside the 16-px border the 4x4 block [i-2,i+2) x [j0-2,j0+2) is always in the image
            int ci = ok ? i : 16, cj = ok ? j0 : 16;
            // "all 16 pixels non-zero": four unaligned 4-byte loads issued together and a has-zero-byte test per row (the byte-by-byte
            // form compiled to 16 dependent loads, each behind the previous one's branch: ~2 us of every residual launch)
            bool valid = true;
#pragma unroll
            for (int a = -2; a < 2; a++) {
                uint32_t r4;
                __builtin_memcpy(&r4, nextImage + (ci + a) * w + cj - 2, 4);
                valid = valid & (((r4 - 0x01010101u) & ~r4 & 0x80808080u) == 0u);
            }
            short valx = dIdx[kk2], valy = dIdy[kk2];
            float mTwo = (float)((valx * valx) + (valy * valy));
            o.d1[u] = nextDepth[kk2];
            o.ni[u] = nextImage[kk2];
            o.cand[u] = ok & valid & (mTwo >= minScale) & !(o.d1[u] != o.d1[u]);   // (bitwise: no load of this stage may hide behind a branch)
        }
    };
    const int base0 = bid * (RED_THREADS * RED_IT) + threadIdx.x, stride = nblk * RED_THREADS * RED_IT;
    Own own;
    own_pixels(base0, own);   // in flight before the warp matrices are known (PRO: under the prologue's solve)
    const bool from_state = FROM_STATE || st != nullptr;
    const float* kk = from_state ? st->krkinv : ex.krkinv;
    const float* ktp = from_state ? st->kt : ex.kt;
    if (PRO) { const float* sp = gn_prologue(st, *pro, rrt_out); kk = sp + 12; ktp = sp + 21; }
    float krk[9];
#pragma unroll
    for (int k = 0; k < 9; k++) krk[k] = kk[k];
    const float kt0 = ktp[0], kt1 = ktp[1], kt2 = ktp[2];
    int cnt = 0, sig = 0;
    for (int base = base0; base < N; base += stride) {
        if (base != base0) own_pixels(base, own);
        int gj[RED_IT];
        float td1[RED_IT];
#pragma unroll
        for (int u = 0; u < RED_IT; u++) {   // stage 2: warp into the last image
            int i = own.kidx[u] / w, j0 = own.kidx[u] - i * w;
            int y = i, x = j0;
            td1[u] = (float)(own.d1[u] * (krk[6] * x + krk[7] * y + krk[8]) + kt2);
            int u0 = f2i_rn((own.d1[u] * (krk[0] * x + krk[1] * y + krk[2]) + kt0) / td1[u]);
            int v0 = f2i_rn((own.d1[u] * (krk[3] * x + krk[4] * y + krk[5]) + kt1) / td1[u]);
            own.cand[u] = own.cand[u] & ((u0 >= 0) & (v0 >= 0) & (u0 < w) & (v0 < h));
            gj[u] = own.cand[u] ? v0 * w + u0 : 0;
        }
        float d0[RED_IT];
        uint8_t li[RED_IT];
#pragma unroll
        for (int u = 0; u < RED_IT; u++) { d0[u] = lastDepth[gj[u]]; li[u] = lastImage[gj[u]]; }   // stage 3: gathers
#pragma unroll
        for (int u = 0; u < RED_IT; u++) {   // stage 4
            Corres8 c;
            const bool hit = own.cand[u] & (d0[u] > 0) & (fabsf(td1[u] - d0[u]) <= maxDepthDelta) & (li[u] != 0);
            const int v0 = gj[u] / w, u0 = gj[u] - v0 * w;
            const float diff = (float)own.ni[u] - (float)li[u];
            c.zx = hit ? (short)u0 : (short)-1; c.zy = hit ? (short)v0 : (short)-1;
            c.diff = hit ? diff : 0.f;
            cnt += hit ? 1 : 0;
            sig += hit ? (int)(diff * diff) : 0;
            if (own.kidx[u] < N) corres[own.kidx[u]] = c;
        }
    }
    __shared__ int lds[RED_WAVES][2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    cnt = wave_sum_i(cnt);
    sig = wave_sum_i(sig);
    if (lane == 0) { lds[wid][0] = cnt; lds[wid][1] = sig; }
    __syncthreads();
    if (threadIdx.x < 2) {
        int s = 0;
        for (int wv = 0; wv < RED_WAVES; wv++) s += lds[wv][threadIdx.x];
        if (!FROM_STATE) partials[bid * 2 + threadIdx.x] = s;
        // grand totals by integer atomics (exact in any order): the next launch reads two ints instead of reducing rows
        if (res_total && s) atomicAdd(&res_total[threadIdx.x], s);
    }
}
__global__ __launch_bounds__(RED_THREADS) void k_rgb_residual(const DevState* __restrict__ st, ResArgs ex, float minScale, const int16_t* __restrict__ dIdx,
                                                              const int16_t* __restrict__ dIdy, const float* __restrict__ lastDepth,
                                                              const float* __restrict__ nextDepth, const uint8_t* __restrict__ lastImage,
                                                              const uint8_t* __restrict__ nextImage, Corres8* __restrict__ corres, float maxDepthDelta,
                                                              int w, int h, int* __restrict__ partials)
{
    residual_body(blockIdx.x, gridDim.x, st, ex, minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, corres, maxDepthDelta, w, h, partials);
}

// One launch for the two independent reductions of a Gauss-Newton iteration: blocks [0, nb_icp) run the
// ICP reduction, blocks [nb_icp, nb_icp + nb_res) the photometric residual pass.
struct PairArgs {
    const float *vmap_curr, *nmap_curr, *vmap_prev, *nmap_prev;
    float fx, fy, cx, cy, distThres, angleThres;
    float minScale, maxDepthDelta;
    const int16_t *dIdx, *dIdy;
    const float *lastDepth, *nextDepth;   // the same image in the frame-to-model tracker (reference quirk), two images model-to-model
    const uint8_t *lastImage, *nextImage;
    Corres8* corres;
    int w, h, nb_icp, nb_res;
    int check_skip;
    int lds_tiles;   // ICP half on 64 x 16 tiles with the model maps staged in LDS (option icp_lds, level 0)
};
// All kernel-argument words a launch needs are pulled into SGPRs in the entry block (one scalar round trip).  Left to itself the compiler sinks
// each group of s_loads into the branch that uses it: four to five dependent scalar round trips in front of the first vector load of a
// latency-bound launch.
#define IFX_PIN_S(x) asm volatile("" ::"s"(x))
template <bool LDS_TILES, bool CHECK_SKIP, bool PRO = false>
__global__ __launch_bounds__(RED_THREADS, 4) void k_icp_residual(const DevState* __restrict__ st, int nb_icp, int w, int h, double* __restrict__ gacc, int* __restrict__ gres, PairArgs a, GnPro g,
                                                              double* __restrict__ rrt_store)
{
    // `st`, `nb_icp`, `w`, `h` and the two hand-off pointers (DevState::gn_acc / gn_res of `st`: separate arguments, so that the state itself stays
    // read-only here and its fields come through the scalar cache) arrive preloaded: the branch below is decided without a load, and each half then fetches its argument words and its
    // state fields in ONE scalar round trip (they used to be four to five dependent ones in front of the first vector load)
    __builtin_assume(st != nullptr);
    if (CHECK_SKIP && st->skip) return;   // model-to-model instance only: the frame-to-model tracker pays no dependent load for it
    if ((int)blockIdx.x < nb_icp) {
#ifdef IFX_EXPERIMENTS
        if (LDS_TILES) { icp_body_lds(blockIdx.x, st, a.vmap_curr, a.nmap_curr, a.vmap_prev, a.nmap_prev, a.fx, a.fy, a.cx, a.cy, a.distThres, a.angleThres, w, h, gacc); return; }
#endif
        IcpArgs ia;   // unused when st != nullptr
        // PRO: `g` and `rrt_store` (DevState::gnp_RRt of this iteration's parity: block 0 publishes the increment it solved for the launch after the next)
        icp_body<false, true, PRO>(blockIdx.x, nb_icp, st, ia, a.vmap_curr, a.nmap_curr, a.vmap_prev, a.nmap_prev, a.fx, a.fy, a.cx, a.cy, a.distThres, a.angleThres, w, h, gacc, &g,
                                   blockIdx.x == 0 ? rrt_store : nullptr);
    } else {
        ResArgs ra;
        residual_body<true, PRO>(blockIdx.x - nb_icp, a.nb_res, st, ra, a.minScale, a.dIdx, a.dIdy, a.lastDepth, a.nextDepth, a.lastImage, a.nextImage, a.corres, a.maxDepthDelta, w,
                                 h, nullptr, gres, &g, blockIdx.x == 0 ? rrt_store : nullptr);
    }
}

#ifdef IFX_EXPERIMENTS
// The two reductions of k_icp_residual on the SAME pixels of one thread: PX pixels per thread, no loop.  Every coalesced load of the thread -- the ICP's
// vertex / normal and the residual pass's window, gradients, depth, intensity -- leaves in one batch, every gather (model vertex / normal, warped depth
// and intensity) in a second one: two memory round trips for the whole launch, whatever the level, where the split form runs an ICP block through three
// pipelined rounds at level 0 and needs 1 656 blocks (more than fit the GPU at once).  Same rows, same exact sums.
template <int PX, bool CHECK_SKIP>
__global__ __launch_bounds__(RED_THREADS) void k_icp_residual_px(const DevState* __restrict__ st, int nb, int w, int h, double* __restrict__ gacc, int* __restrict__ gres, PairArgs a)
{
    __builtin_assume(st != nullptr);
    if (CHECK_SKIP && st->skip) return;
    const int N = w * h, bid = blockIdx.x, tid = threadIdx.x;
    float Rcurr[9], Rprev_inv[9], krk[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rcurr[k] = st->Rcurr[k]; Rprev_inv[k] = st->Rprev_inv[k]; krk[k] = st->krkinv[k]; }
    const v3 tc = v3m(st->tcurr[0], st->tcurr[1], st->tcurr[2]), tp = v3m(st->tprev[0], st->tprev[1], st->tprev[2]);
    const float kt0 = st->kt[0], kt1 = st->kt[1], kt2 = st->kt[2];
    const int border = 16;
    // ---- batch 1: everything that is addressed by the pixel itself
    v3 vcurr[PX], ncurr[PX];
    float d1[PX], nif[PX];
    short gx[PX], gy[PX];
    uint32_t r4[PX][4];
    bool okb[PX], inp[PX];
    int xy[PX], pix[PX];
#pragma unroll
    for (int u = 0; u < PX; u++) {
        const int p = (u * nb + bid) * RED_THREADS + tid;
        inp[u] = p < N;
        const int pp = inp[u] ? p : 0;
        pix[u] = p;
        const int i = pp / w, j0 = pp - i * w;
        xy[u] = (i << 16) | j0;
        vcurr[u] = v3m(a.vmap_curr[pp], a.vmap_curr[pp + N], a.vmap_curr[pp + 2 * N]);
        ncurr[u] = v3m(a.nmap_curr[pp], a.nmap_curr[pp + N], a.nmap_curr[pp + 2 * N]);
        okb[u] = inp[u] && i >= border && i < h - border && j0 >= border && j0 < w - border && j0 < w - 5 && i < h - 1;
        const int ci = okb[u] ? i : 16, cj = okb[u] ? j0 : 16;
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_memcpy(&r4[u][q], a.nextImage + (ci + q - 2) * w + cj - 2, 4);
        gx[u] = a.dIdx[pp]; gy[u] = a.dIdy[pp];
        d1[u] = a.nextDepth[pp];
        nif[u] = (float)a.nextImage[pp];
    }
    // ---- projections, then batch 2: every gather
    v3 vcurr_g[PX], vprev[PX], nprev[PX];
    bool inb[PX], cand[PX];
    int gj[PX];
    float td1[PX], d0[PX], lif[PX];
#pragma unroll
    for (int u = 0; u < PX; u++) {
        if (!inp[u]) vcurr[u].x = qnan_f();
        vcurr_g[u] = mulp(Rcurr, vcurr[u]) + tc;
        const v3 vcurr_cp = mulp(Rprev_inv, vcurr_g[u] - tp);
        const int ux = f2i_rn(vcurr_cp.x * a.fx / vcurr_cp.z + a.cx);
        const int uy = f2i_rn(vcurr_cp.y * a.fy / vcurr_cp.z + a.cy);
        inb[u] = !(vcurr[u].x != vcurr[u].x) && !(ux < 0 || uy < 0 || ux >= w || uy >= h || vcurr_cp.z < 0);
        const int j = inb[u] ? uy * w + ux : 0;
        bool valid = true;
#pragma unroll
        for (int q = 0; q < 4; q++) valid = valid & (((r4[u][q] - 0x01010101u) & ~r4[u][q] & 0x80808080u) == 0u);
        const float mTwo = (float)((gx[u] * gx[u]) + (gy[u] * gy[u]));
        const bool c0 = okb[u] & valid & (mTwo >= a.minScale) & !(d1[u] != d1[u]);
        const int y = xy[u] >> 16, x = xy[u] & 0xFFFF;
        td1[u] = (float)(d1[u] * (krk[6] * x + krk[7] * y + krk[8]) + kt2);
        const int u0 = f2i_rn((d1[u] * (krk[0] * x + krk[1] * y + krk[2]) + kt0) / td1[u]);
        const int v0 = f2i_rn((d1[u] * (krk[3] * x + krk[4] * y + krk[5]) + kt1) / td1[u]);
        cand[u] = c0 & ((u0 >= 0) & (v0 >= 0) & (u0 < w) & (v0 < h));
        gj[u] = cand[u] ? v0 * w + u0 : 0;
        vprev[u] = v3m(a.vmap_prev[j], a.vmap_prev[j + N], a.vmap_prev[j + 2 * N]);
        nprev[u] = v3m(a.nmap_prev[j], a.nmap_prev[j + N], a.nmap_prev[j + 2 * N]);
        d0[u] = a.lastDepth[gj[u]];
        lif[u] = (float)a.lastImage[gj[u]];
    }
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.0;
    int cnt = 0, sig = 0;
#pragma unroll
    for (int u = 0; u < PX; u++) {
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        bool found = false;
        if (inb[u]) {
            const v3 ncurr_g = mulp(Rcurr, ncurr[u]);
            const float dist = norm(vprev[u] - vcurr_g[u]);
            const float sine = norm(cross(ncurr_g, nprev[u]));
            found = (sine < a.angleThres && dist <= a.distThres && !(ncurr[u].x != ncurr[u].x) && !(nprev[u].x != nprev[u].x));
            if (found) {
                const v3 s_cp = mulp(Rprev_inv, vcurr_g[u] - tp);
                const v3 d_cp = mulp(Rprev_inv, vprev[u] - tp);
                const v3 n_cp = mulp(Rprev_inv, nprev[u]);
                const v3 c = cross(s_cp, n_cp);
                row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
                row[3] = c.x; row[4] = c.y; row[5] = c.z;
                row[6] = dot(n_cp, s_cp - d_cp);
            }
        }
        products7<0>(row, found, acc);
        const bool hit = cand[u] & (d0[u] > 0) & (fabsf(td1[u] - d0[u]) <= a.maxDepthDelta) & (lif[u] != 0.f);
        const int v0 = gj[u] / w, u0 = gj[u] - v0 * w;
        const float diff = nif[u] - lif[u];
        Corres8 c8;
        c8.zx = hit ? (short)u0 : (short)-1; c8.zy = hit ? (short)v0 : (short)-1;
        c8.diff = hit ? diff : 0.f;
        cnt += hit ? 1 : 0;
        sig += hit ? (int)(diff * diff) : 0;
        if (inp[u]) a.corres[pix[u]] = c8;
    }
    block_sum_exact<29>(acc, gacc, bid % IFX_ACC_REPL);
    __shared__ int lds2[RED_WAVES][2];
    const int lane = tid & 63, wid = tid >> 6;
    cnt = wave_sum_i(cnt);
    sig = wave_sum_i(sig);
    if (lane == 0) { lds2[wid][0] = cnt; lds2[wid][1] = sig; }
    __syncthreads();
    if (tid < 2) {
        int s2 = 0;
        for (int wv = 0; wv < RED_WAVES; wv++) s2 += lds2[wv][tid];
        if (s2) atomicAdd(&gres[tid], s2);
    }
}
#endif   // IFX_EXPERIMENTS

// RGBReduction, EF/Cuda/reduce.cu:494-619.  sigma is either explicit (stage API) or derived from the
// residual pass's block partials with the reference's precedence quirk (EF/Utils/RGBDOdometry.cpp:461).
__device__ __forceinline__ void rgb_step_body(int bid, int nblk, const Corres8* __restrict__ corres, float sigma_explicit, const int* __restrict__ res_partials,
                                              int res_blocks, const float* __restrict__ cloud, float fx, float fy, const int16_t* __restrict__ dIdx,
                                              const int16_t* __restrict__ dIdy, float sobelScale, int w, int h, double* __restrict__ gacc, const int* __restrict__ res_total = nullptr)
{
#ifdef IFX_STAMPS
    long long ts_local[3]; long long ts_start = clock64();
#endif
    const int N = w * h;
    // stage 1 loads are issued before the sigma reduction so that both latencies overlap
    const int base0 = bid * (RED_THREADS * RED_IT_RGB) + threadIdx.x;
    // first round's records and gradients: requested before sigma is worked out, so that both latencies overlap
    Corres8 c[RED_IT_RGB];
    short gx[RED_IT_RGB], gy[RED_IT_RGB];
#pragma unroll
    for (int u = 0; u < RED_IT_RGB; u++) {
        const int k = base0 + u * RED_THREADS;
        const bool in = k < N;
        const int kk = in ? k : 0;
        c[u] = corres[kk];
        gx[u] = dIdx[kk]; gy[u] = dIdy[kk];
        if (!in) c[u].zx = -1;
    }
    float sigma = sigma_explicit;
    if (res_total) {
        int cnt = res_total[0], sg = res_total[1];
        float q = (float)sg / (float)cnt;
        sigma = (float)sqrt((double)((q == 0) ? 1 : cnt));
    } else if (res_partials) {
        __shared__ int s_cs[2];
        if (threadIdx.x < 64) {
            int cnt = 0, sg = 0;
            for (int b0 = threadIdx.x; b0 < res_blocks; b0 += 64 * 4) {
                int c[4], g[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { int b = b0 + 64 * u; bool in = b < res_blocks; c[u] = in ? res_partials[2 * b] : 0; g[u] = in ? res_partials[2 * b + 1] : 0; }
#pragma unroll
                for (int u = 0; u < 4; u++) { cnt += c[u]; sg += g[u]; }
            }
            cnt = wave_sum_i(cnt);
            sg = wave_sum_i(sg);
            if (threadIdx.x == 0) { s_cs[0] = cnt; s_cs[1] = sg; }
        }
        __syncthreads();
        int cnt = s_cs[0], sg = s_cs[1];
        float q = (float)sg / (float)cnt;
        sigma = (float)sqrt((double)((q == 0) ? 1 : cnt));
    }
#ifdef IFX_STAMPS
    g_ts[0] = clock64();
#endif
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.0;
    // software-pipelined like the ICP rounds: the records and gradients of the NEXT round are requested right behind this round's cloud gathers
    const int stride = nblk * RED_THREADS * RED_IT_RGB;
    for (int base = base0; base < N; base += stride) {
        float X[RED_IT_RGB], Y[RED_IT_RGB], Z[RED_IT_RGB];
#pragma unroll
        for (int u = 0; u < RED_IT_RGB; u++) {
            int g = (c[u].zx >= 0) ? ((int)c[u].zy * w + (int)c[u].zx) * 3 : 0;
            X[u] = cloud[g]; Y[u] = cloud[g + 1]; Z[u] = cloud[g + 2];
        }
        Corres8 cn[RED_IT_RGB];
        short gxn[RED_IT_RGB], gyn[RED_IT_RGB];
#pragma unroll
        for (int u = 0; u < RED_IT_RGB; u++) {   // (a thread past its last round reads record 0 and drops it: no branch around the loads)
            const int k = base + stride + u * RED_THREADS;
            const bool in = k < N;
            const int kk = in ? k : 0;
            cn[u] = corres[kk];
            gxn[u] = dIdx[kk]; gyn[u] = dIdy[kk];
            if (!in) cn[u].zx = -1;
        }
#pragma unroll
        for (int u = 0; u < RED_IT_RGB; u++) {
            float row[7] = {0, 0, 0, 0, 0, 0, 0};
            bool found = c[u].zx >= 0;
            if (found) {
                float wgt = sigma + fabsf(c[u].diff);
                wgt = wgt > 1.19209290E-07F ? 1.0f / wgt : 1.0f;
                if (sigma == -1) wgt = 1;
                row[6] = -wgt * c[u].diff;
                float invz = (float)(1.0 / Z[u]);
                float dI_dx = wgt * sobelScale * gx[u];
                float dI_dy = wgt * sobelScale * gy[u];
                float v0 = dI_dx * fx * invz;
                float v1 = dI_dy * fy * invz;
                float v2 = -(v0 * X[u] + v1 * Y[u]) * invz;
                row[0] = v0; row[1] = v1; row[2] = v2;
                row[3] = -Z[u] * v1 + Y[u] * v2;
                row[4] = Z[u] * v0 - X[u] * v2;
                row[5] = -Y[u] * v0 + X[u] * v1;
            }
            products7<1>(row, found, acc);
        }
#pragma unroll
        for (int u = 0; u < RED_IT_RGB; u++) { c[u] = cn[u]; gx[u] = gxn[u]; gy[u] = gyn[u]; }
    }
#ifdef IFX_STAMPS
    g_ts[1] = clock64();
#endif
    block_sum_exact<29>(acc, gacc, bid % IFX_ACC_REPL);
#ifdef IFX_STAMPS
    g_ts[2] = clock64();
    if (threadIdx.x == 0) { s_dbg_blk[0] = ts_local[0] - ts_start; s_dbg_blk[1] = ts_local[1] - ts_local[0]; s_dbg_blk[2] = ts_local[2] - ts_local[1]; }
#endif
}
__global__ __launch_bounds__(RED_THREADS) void k_rgb_step(const Corres8* __restrict__ corres, float sigma_explicit, const int* __restrict__ res_partials,
                                                          int res_blocks, const float* __restrict__ cloud, float fx, float fy,
                                                          const int16_t* __restrict__ dIdx, const int16_t* __restrict__ dIdy, float sobelScale, int w, int h,
                                                          double* __restrict__ gacc)
{
    rgb_step_body(blockIdx.x, gridDim.x, corres, sigma_explicit, res_partials, res_blocks, cloud, fx, fy, dIdx, dIdy, sobelScale, w, h, gacc);
}

// SO3Reduction, EF/Cuda/reduce.cu:938-1076
struct So3Args { float ib[9], kinv[9], krlr[9]; };
__device__ inline float gradx(const uint8_t* img, int w, int px, int py)
{
    return (((float)img[py * w + px - 1] + (float)img[py * w + px]) / 2.0f) - (((float)img[py * w + px + 1] + (float)img[py * w + px]) / 2.0f);
}
__device__ inline float grady(const uint8_t* img, int w, int px, int py)
{
    return (((float)img[(py - 1) * w + px] + (float)img[py * w + px]) / 2.0f) - (((float)img[(py + 1) * w + px] + (float)img[py * w + px]) / 2.0f);
}
__device__ __forceinline__ void so3_body(int bid, int nblk, const DevState* __restrict__ st, const So3Args& ex, const uint8_t* __restrict__ lastImage,
                                         const uint8_t* __restrict__ nextImage, int w, int h, double* __restrict__ gacc)
{
    const float* ibp = st ? st->imageBasis : ex.ib;
    const float* kip = st ? st->kinv : ex.kinv;
    const float* krp = st ? st->krlr : ex.krlr;
    float ib[9], kinv[9], krlr[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { ib[k] = ibp[k]; kinv[k] = kip[k]; krlr[k] = krp[k]; }
    double acc[11];
#pragma unroll
    for (int k = 0; k < 11; k++) acc[k] = 0.0;
    constexpr int E[4] = IFX_E_SO3;
    const int N = w * h;
    for (int k = bid * RED_THREADS + threadIdx.x; k < N; k += RED_THREADS * nblk) {
        int y = k / w, x = k - y * w;
        v3 up = v3m((float)x, (float)y, 1.0f);
        v3 wp = mulp(ib, up);
        int wx = f2i_rn(wp.x / wp.z), wy = f2i_rn(wp.y / wp.z);
        bool found = (wx >= 1 && wx < w - 1 && wy >= 1 && wy < h - 1 && x >= 1 && x < w - 1 && y >= 1 && y < h - 1);
        float row[4] = {0, 0, 0, 0};
        if (found) {
            float gx = (gradx(nextImage, w, wx, wy) + gradx(lastImage, w, x, y)) / 2.0f;
            float gy = (grady(nextImage, w, wx, wy) + grady(lastImage, w, x, y)) / 2.0f;
            v3 pt = mulp(kinv, up);
            float z2 = pt.z * pt.z;
            float a = krlr[0], b = krlr[1], c = krlr[2], d = krlr[3], e = krlr[4], f = krlr[5], g = krlr[6], hh = krlr[7], ii = krlr[8];
            v3 lp = v3m(((pt.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
                        ((pt.z * (e * gy + b * gx)) - (gy * hh * y) - (gx * hh * x)) / z2,
                        ((pt.z * (f * gy + c * gx)) - (gy * ii * y) - (gx * ii * x)) / z2);
            v3 jr = cross(lp, pt);
            row[0] = jr.x; row[1] = jr.y; row[2] = jr.z;
            row[3] = -((float)nextImage[wy * w + wx] - (float)lastImage[k]);
        }
        int s = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 4; j++) acc[s++] += ifx_quant(row[i] * row[j], ifx_magic(E[i] + E[j] - IFX_SO3_TERM_BITS));
        acc[9] += ifx_quant(row[3] * row[3], ifx_magic(2 * E[3] - IFX_SO3_TERM_BITS));
        acc[10] += found ? 1.0 : 0.0;
    }
    block_sum_exact<11>(acc, gacc, bid % IFX_ACC_REPL);
}
__global__ __launch_bounds__(RED_THREADS) void k_so3(const DevState* __restrict__ st, So3Args ex, const uint8_t* __restrict__ lastImage,
                                                     const uint8_t* __restrict__ nextImage, int w, int h, double* __restrict__ gacc)
{
    if (st && st->so3_done) return;   // block-uniform early exit once the host-free loop has converged
    so3_body(blockIdx.x, gridDim.x, st, ex, lastImage, nextImage, w, h, gacc);
}

// ======================================================================= device-side solve (a8)

// Pivoted LDLT (Eigen's LDLT as used by EF/Utils/RGBDOdometry.cpp:368,552).  Everything is
// compile-time unrolled with predicated swaps so that the matrix lives in registers: a
// runtime-indexed private array goes to scratch memory and made the first version of the solve
// kernel take 24 us (profiles/archive/r01_a_*).
template <typename T, int N>
__device__ __forceinline__ void ldlt_solve_n(const T* Ain, const T* bin, T* x, T tiny)
{
    T A[N * N], y[N], bb[N];
    int p[N];
#pragma unroll
    for (int i = 0; i < N * N; i++) A[i] = Ain[i];
#pragma unroll
    for (int i = 0; i < N; i++) { p[i] = i; bb[i] = bin[i]; }
#pragma unroll
    for (int k = 0; k < N; k++) {
        int piv = k;
        T big = A[k * N + k] < 0 ? -A[k * N + k] : A[k * N + k];
#pragma unroll
        for (int i = k + 1; i < N; i++) {
            T v = A[i * N + i] < 0 ? -A[i * N + i] : A[i * N + i];
            if (v > big) { big = v; piv = i; }
        }
#pragma unroll
        for (int i = k + 1; i < N; i++) {   // symmetric swap k <-> piv, predicated (no dynamic indexing)
            const bool sw = (i == piv);
#pragma unroll
            for (int j = 0; j < N; j++) { T a = A[k * N + j], b = A[i * N + j]; A[k * N + j] = sw ? b : a; A[i * N + j] = sw ? a : b; }
#pragma unroll
            for (int j = 0; j < N; j++) { T a = A[j * N + k], b = A[j * N + i]; A[j * N + k] = sw ? b : a; A[j * N + i] = sw ? a : b; }
            int pa = p[k], pb = p[i];
            p[k] = sw ? pb : pa; p[i] = sw ? pa : pb;
        }
        const T d = A[k * N + k];
        const bool zero = (big <= (T)0);
#pragma unroll
        for (int i = k + 1; i < N; i++) A[i * N + k] = zero ? (T)0 : A[i * N + k] / d;
#pragma unroll
        for (int i = k + 1; i < N; i++)
#pragma unroll
            for (int j = k + 1; j < N; j++) A[i * N + j] = zero ? A[i * N + j] : A[i * N + j] - A[i * N + k] * d * A[j * N + k];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        T v = 0;
#pragma unroll
        for (int j = 0; j < N; j++) v = (p[i] == j) ? bb[j] : v;
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < i; j++) y[i] -= A[i * N + j] * y[j];
#pragma unroll
    for (int i = 0; i < N; i++) {
        T d = A[i * N + i];
        T ad = d < 0 ? -d : d;
        y[i] = (ad > tiny) ? y[i] / d : (T)0;
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--)
#pragma unroll
        for (int j = i + 1; j < N; j++) y[i] -= A[j * N + i] * y[j];
#pragma unroll
    for (int j = 0; j < N; j++) {
        T v = 0;
#pragma unroll
        for (int i = 0; i < N; i++) v = (p[i] == j) ? y[i] : v;
        x[j] = v;
    }
}

// Unpivoted LDLT for the (symmetric positive semi-definite) 6x6 normal equations: the serial solve
// sits on the critical path of every Gauss-Newton iteration, and the predicated pivot swaps of the
// Eigen-style version cost ~1500 extra 64-bit selects on one lane.  For SPD input the two agree to
// f64 rounding; a zero pivot gives a zero component, as Eigen's LDLT does (all-zero system when no
// correspondence survives).
template <typename T, int N>
__device__ __forceinline__ void ldlt_nopivot(const T* Ain, const T* bin, T* x, T tiny)
{
    T A[N * N], y[N], inv[N];
#pragma unroll
    for (int i = 0; i < N * N; i++) A[i] = Ain[i];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const T d = A[k * N + k];
        const T ad = d < 0 ? -d : d;
        const bool zero = !(ad > tiny);
        inv[k] = zero ? (T)0 : (T)1 / d;      // one division per pivot; a zero pivot gives a zero component
#pragma unroll
        for (int i = k + 1; i < N; i++) A[i * N + k] = A[i * N + k] * inv[k];
#pragma unroll
        for (int i = k + 1; i < N; i++)
#pragma unroll
            for (int j = k + 1; j <= i; j++) {
                T v = A[i * N + j] - A[i * N + k] * d * A[j * N + k];
                A[i * N + j] = v;
                A[j * N + i] = v;
            }
    }
#pragma unroll
    for (int i = 0; i < N; i++) y[i] = bin[i];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < i; j++) y[i] -= A[i * N + j] * y[j];
#pragma unroll
    for (int i = 0; i < N; i++) y[i] = y[i] * inv[i];
#pragma unroll
    for (int i = N - 1; i >= 0; i--)
#pragma unroll
        for (int j = i + 1; j < N; j++) y[i] -= A[j * N + i] * y[j];
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = y[i];
}

// sin and cos of a small angle by Taylor series in f64 (|x| < 0.5: truncation < 1e-19); the libm
// routines carry a Payne-Hanek path and cost several microseconds on a lone lane.
__device__ __forceinline__ void sincos_small(double x, double* s, double* c)
{
    if (!(fabs(x) < 0.5)) { *s = sin(x); *c = cos(x); return; }
    const double x2 = x * x;
    double ps = 1.0 / 6227020800.0;          // 1/13!
    ps = ps * x2 - 1.0 / 39916800.0;         // 1/11!
    ps = ps * x2 + 1.0 / 362880.0;           // 1/9!
    ps = ps * x2 - 1.0 / 5040.0;
    ps = ps * x2 + 1.0 / 120.0;
    ps = ps * x2 - 1.0 / 6.0;
    ps = ps * x2 + 1.0;
    *s = ps * x;
    double pc = 1.0 / 87178291200.0;         // 1/14!
    pc = pc * x2 - 1.0 / 479001600.0;        // 1/12!
    pc = pc * x2 + 1.0 / 3628800.0;          // 1/10!
    pc = pc * x2 - 1.0 / 40320.0;
    pc = pc * x2 + 1.0 / 720.0;
    pc = pc * x2 - 1.0 / 24.0;
    pc = pc * x2 + 0.5;
    *c = 1.0 - pc * x2;
}

// OdometryProvider::rodrigues, EF/Utils/OdometryProvider.h:35-71
__device__ void rodrigues_d(const double* src, double* R)
{
    double rx = src[0], ry = src[1], rz = src[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (theta >= DBL_EPSILON) {
        double c, s;
        sincos_small(theta, &s, &c);
        double c1 = 1.0 - c;
        double it = theta ? 1.0 / theta : 0.0;
        rx *= it; ry *= it; rz *= it;
        double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx_[k];
    }
}

template <int N>
__device__ __forceinline__ void matmul_dn(const double* A, const double* B, double* C)
{
    double T[N * N];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < N; k++) s += A[i * N + k] * B[k * N + j];
            T[i * N + j] = s;
        }
#pragma unroll
    for (int i = 0; i < N * N; i++) C[i] = T[i];
}
#define matmul_d(n, A, B, C) matmul_dn<n>(A, B, C)

__device__ void k_matrix_d(float fx, float fy, float cx, float cy, double* K, double* Kinv)
{
    for (int i = 0; i < 9; i++) K[i] = Kinv[i] = 0;
    K[0] = fx; K[4] = fy; K[2] = cx; K[5] = cy; K[8] = 1;
    Kinv[0] = 1.0 / K[0]; Kinv[4] = 1.0 / K[4];
    Kinv[2] = -K[2] / K[0]; Kinv[5] = -K[5] / K[4]; Kinv[8] = 1;
}

// krkinv / kt of the next residual pass from resultRt (EF/Utils/RGBDOdometry.cpp:424-434), all in registers
__device__ __forceinline__ void warp_from(const double* M, float fx, float fy, float cx, float cy, float* krk, float* kt)
{
    // K = [fx 0 cx; 0 fy cy; 0 0 1], Kinv = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1].  The products below are K R Kinv and K t with the
    // terms that multiply K's / Kinv's zeros left out, in the order a dense 3x3 product adds the others: the same values (x + 0 = x)
    // in half the f64 instructions of this lone-lane path.
    const double K0 = fx, K2 = cx, K4 = fy, K5 = cy;
    const double I0 = 1.0 / K0, I4 = 1.0 / K4, I2 = -K2 / K0, I5 = -K5 / K4;
    double R3[9], tt[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R3[i * 3 + j] = M[j * 4 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) tt[i] = -(R3[i * 3 + 0] * M[3] + R3[i * 3 + 1] * M[7] + R3[i * 3 + 2] * M[11]);
    double KR[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        KR[j] = K0 * R3[j] + K2 * R3[6 + j];
        KR[3 + j] = K4 * R3[3 + j] + K5 * R3[6 + j];
        KR[6 + j] = R3[6 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        krk[i * 3 + 0] = (float)(KR[i * 3] * I0);
        krk[i * 3 + 1] = (float)(KR[i * 3 + 1] * I4);
        krk[i * 3 + 2] = (float)((KR[i * 3] * I2 + KR[i * 3 + 1] * I5) + KR[i * 3 + 2]);
    }
    kt[0] = (float)(K0 * tt[0] + K2 * tt[2]);
    kt[1] = (float)(K4 * tt[1] + K5 * tt[2]);
    kt[2] = (float)tt[2];
}
__device__ void set_warp_matrices(DevState* st, float fx, float fy, float cx, float cy)
{
    double M[16];
    float krk[9], kt[3];
    for (int k = 0; k < 16; k++) M[k] = st->resultRt[k];
    warp_from(M, fx, fy, cx, cy, krk, kt);
    for (int k = 0; k < 9; k++) st->krkinv[k] = krk[k];
    for (int k = 0; k < 3; k++) st->kt[k] = kt[k];
}

__device__ void set_so3_matrices(DevState* st, float fx, float fy, float cx, float cy)
{
    double K[9], Kinv[9], KR[9], H[9];
    k_matrix_d(fx, fy, cx, cy, K, Kinv);
    matmul_d(3, K, st->resultR, KR);
    matmul_d(3, KR, Kinv, H);
    for (int k = 0; k < 9; k++) { st->imageBasis[k] = (float)H[k]; st->kinv[k] = (float)Kinv[k]; st->krlr[k] = (float)KR[k]; }
}

// start of a tracker run (model side): Rprev/tprev from the current pose (:278-311, :388-403), then the seed of the
// Gauss-Newton loop from the SO(3) result held in the slot's shadow state (:392-403)
__global__ void k_track_gn_begin(DevState* st, const DevState* __restrict__ ss, int so3, float fx, float fy, float cx, float cy, int keep_last, int persist)
{
    if (persist) gn_persist_reset(st, threadIdx.x, blockDim.x);
    if (threadIdx.x != 0) return;
    gn_begin_dev(st, ss, so3, fx, fy, cx, cy, keep_last);
}
// start of a tracker run (EF/Utils/RGBDOdometry.cpp:384-434): previous pose, its inverse, the SO(3) result as the first estimate, the first warp matrices
__device__ void gn_begin_dev(DevState* st, const DevState* __restrict__ ss, int so3, float fx, float fy, float cx, float cy, int keep_last)
{
    // Everything this lane reads is loaded ahead of its first store and the values then stay in registers: `st` aliases itself, so reading the state back between the
    // stores made this a chain of ~40 L2 round trips in one block of the launch it rides on (k_model_down of the coarsest level).
    float P[16];
#pragma unroll
    for (int k = 0; k < 16; k++) P[k] = st->pose[k];
    float so3_err = 0.f, so3_cnt = 0.f;
    double sR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (so3) {
        so3_err = ss->lastSO3Error; so3_cnt = ss->lastSO3Count;
#pragma unroll
        for (int k = 0; k < 9; k++) sR[k] = ss->resultR[k];
    }
    if (!keep_last) {   // bootstrap: lastPose is the pose before the guess was applied (k_bootstrap_pose)
#pragma unroll
        for (int k = 0; k < 16; k++) st->last_pose[k] = P[k];
    }
    float m[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) { m[r * 3 + c] = P[r * 4 + c]; st->Rprev[r * 3 + c] = st->Rcurr[r * 3 + c] = m[r * 3 + c]; }
        st->tprev[r] = st->tcurr[r] = P[r * 4 + 3];
    }
    {   // general 3x3 inverse, as Eigen's Matrix3f::inverse (:388)
        float* o = st->Rprev_inv;
        float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
        float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
        float id = 1.0f / det;
        o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
        o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    }
    st->lastICPError = 0; st->lastICPCount = 0; st->lastRGBError = 0; st->lastRGBCount = 0;
    st->lastSO3Error = so3_err; st->lastSO3Count = so3_cnt;
    double M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (so3) {
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int y = 0; y < 3; y++) M[x * 4 + y] = sR[x * 3 + y];
    }
#pragma unroll
    for (int k = 0; k < 16; k++) st->resultRt[k] = M[k];
    float krk[9], kt[3];
    warp_from(M, fx, fy, cx, cy, krk, kt);   // (set_warp_matrices on the registers)
#pragma unroll
    for (int k = 0; k < 9; k++) st->krkinv[k] = krk[k];
#pragma unroll
    for (int k = 0; k < 3; k++) st->kt[k] = kt[k];
}

// start of the SO(3) pre-alignment (frame side; `st` is the slot's shadow state): identity increments (:313-330)
__global__ void k_so3_begin(DevState* st, float fx2, float fy2, float cx2, float cy2)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 9; k++) { st->resultR[k] = st->lastResultR[k] = (k % 4 == 0) ? 1.0 : 0.0; st->R_lr[k] = (k % 4 == 0) ? 1.f : 0.f; }
    st->so3_lastError = FLT_MAX / 2; st->so3_lastCount = FLT_MAX / 2;
    st->so3_done = 0;
    st->lastSO3Error = 0; st->lastSO3Count = 0;
    set_so3_matrices(st, fx2, fy2, cx2, cy2);
}

__device__ __forceinline__ void so3_update_scalar(DevState* st, const float* o, float fx2, float fy2, float cx2, float cy2);
// one SO(3) iteration's host logic, EF/Utils/RGBDOdometry.cpp:348-380: the 11 exact totals are read from the accumulator rows
// (and the rows cleared for the next launch), lane 0 then runs the scalar logic
__device__ __forceinline__ void so3_update_wave(DevState* st, double* __restrict__ gacc, float fx2, float fy2, float cx2, float cy2)
{
    const int lane = threadIdx.x & 63;
    double v = 0;
    if (lane < 11) { v = acc_total(gacc, lane); acc_clear(gacc, lane); if (range_exceeded_so3(lane, v)) atomicAdd(&st->range_exceeded, 1); }
    float o[11];
#pragma unroll
    for (int k = 0; k < 11; k++) o[k] = (float)__shfl(v, k, 64);
    if (lane != 0) return;
    so3_update_scalar(st, o, fx2, fy2, cx2, cy2);
}
__device__ __forceinline__ void so3_update_scalar(DevState* st, const float* o, float fx2, float fy2, float cx2, float cy2)
{
    float jtj[9], jtr[3];
    jtj[0] = o[0]; jtj[1] = jtj[3] = o[1]; jtj[2] = jtj[6] = o[2]; jtr[0] = o[3];
    jtj[4] = o[4]; jtj[5] = jtj[7] = o[5]; jtr[1] = o[6];
    jtj[8] = o[7]; jtr[2] = o[8];
    float err = sqrtf(o[9]) / o[10], cnt = o[10];
    st->lastSO3Error = err; st->lastSO3Count = cnt;
    if (err < st->so3_lastError && st->so3_lastCount == cnt) { st->so3_done = 1; return; }
    else if (err > st->so3_lastError + 0.001) {
        st->lastSO3Error = st->so3_lastError; st->lastSO3Count = st->so3_lastCount;
        for (int k = 0; k < 9; k++) st->resultR[k] = st->lastResultR[k];
        st->so3_done = 1;
        return;
    }
    st->so3_lastError = err; st->so3_lastCount = cnt;
    for (int k = 0; k < 9; k++) st->lastResultR[k] = st->resultR[k];
    float delta[3];
    ldlt_solve_n<float, 3>(jtj, jtr, delta, (float)(1.0 / FLT_MAX));
    double dd[3] = {delta[0], delta[1], delta[2]}, ru[9];
    rodrigues_d(dd, ru);
    float ruf[9], nr[9], Rl[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { ruf[k] = (float)ru[k]; Rl[k] = st->R_lr[k]; }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) nr[i * 3 + j] = ruf[i * 3] * Rl[j] + ruf[i * 3 + 1] * Rl[3 + j] + ruf[i * 3 + 2] * Rl[6 + j];
#pragma unroll
    for (int k = 0; k < 9; k++) { st->R_lr[k] = nr[k]; st->resultR[k] = nr[k]; }
    set_so3_matrices(st, fx2, fy2, cx2, cy2);
}

// SO(3) reduction + (last block) update in one launch; same hand-off as k_rgb_step_solve.
__global__ __launch_bounds__(RED_THREADS) void k_so3_fused(DevState* st, const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage, int w, int h,
                                                           double* __restrict__ gacc, int nb, unsigned int* ticket, float fx2, float fy2, float cx2, float cy2)
{
    __shared__ int s_last;
    if (st->so3_done) return;
    So3Args ex;
    so3_body(blockIdx.x, nb, st, ex, lastImage, nextImage, w, h, gacc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {   // this block's atomic adds were performed at the memory side: every wave drained vmcnt before the barrier
        unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (unsigned int)(nb - 1));
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the totals are read by agent-scope loads: no acquire needed)
    __syncthreads();
    if (threadIdx.x < 64) so3_update_wave(st, gacc, fx2, fy2, cx2, cy2);
}

// rodrigues2, EF/ElasticFusion.cpp:1183-1228 (without the SVD re-orthonormalisation)
__device__ void rodrigues2(const float* R, float* out3)
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

// end of the tracker run (:587-603) + pose write-back + velocity weighting (EF/ElasticFusion.cpp:425-449).
// The run's result (Rc, tc) and the pose it started from (Rp, tp) come in registers -- the lane that solved the last iteration holds them -- and everything else this reads
// (the last frame's pose, the inputs of the view-list decision) is loaded ahead of the first store: `st` aliases itself, so copying state to state element by element was
// a chain of ~30 L2 round trips in the last launch of every frame.
__device__ __forceinline__ void track_end_vals(DevState* st, int rgb, int tracked, const float* Rc_in, const float* tc_in, const float* Rp, const float* tp, float weight_mult,
                                               int commit, unsigned int* lctr)
{
    float LP[16], P0[16], A[16];
    int valid = 0, age = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) LP[k] = st->last_pose[k];
    if (!tracked) {
#pragma unroll
        for (int k = 0; k < 16; k++) P0[k] = st->pose[k];
    }
    const bool decide = commit && lctr;
    if (decide) {
#pragma unroll
        for (int k = 0; k < 16; k++) A[k] = st->vl_pose[k];
        valid = st->vl_valid; age = st->vl_age;
    }
    float pose[16], pinv[16];
    if (tracked) {
        float Rc[9], tc[3];
#pragma unroll
        for (int k = 0; k < 9; k++) Rc[k] = Rc_in[k];
#pragma unroll
        for (int k = 0; k < 3; k++) tc[k] = tc_in[k];
        if (rgb) {
            v3 d = v3m(tc[0] - tp[0], tc[1] - tp[1], tc[2] - tp[2]);
            if (norm(d) > 0.3f) {
#pragma unroll
                for (int k = 0; k < 9; k++) { Rc[k] = Rp[k]; st->Rcurr[k] = Rp[k]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { tc[k] = tp[k]; st->tcurr[k] = tp[k]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int c = 0; c < 3; c++) pose[r * 4 + c] = Rc[r * 3 + c];
            pose[r * 4 + 3] = tc[r];
        }
        pose[12] = pose[13] = pose[14] = 0.f; pose[15] = 1.f;
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) pose[k] = P0[k];
    }
    pose_inverse(pose, pinv);
    float* o_pose = commit ? st->pose : st->spec_pose;
    float* o_inv = commit ? st->pose_inv : st->spec_pose_inv;
#pragma unroll
    for (int k = 0; k < 16; k++) { o_pose[k] = pose[k]; o_inv[k] = pinv[k]; }
    float diff[16];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float sacc = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) sacc += pinv[r * 4 + k] * LP[k * 4 + c];
            diff[r * 4 + c] = sacc;
        }
    float R3[9] = {diff[0], diff[1], diff[2], diff[4], diff[5], diff[6], diff[8], diff[9], diff[10]};
    float rv[3];
    rodrigues2(R3, rv);
    float tn = sqrtf(diff[3] * diff[3] + diff[7] * diff[7] + diff[11] * diff[11]);
    float rn = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    float weighting = fmaxf(tn, rn);
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    const float wgt = fmaxf(1.0f - (weighting / largest), minWeight) * weight_mult;
    if (commit) st->weighting = wgt; else st->spec_weighting = wgt;
    if (decide) vlist_decide_core(st, A, pose, valid, age);   // the frame's pose is final: does the cached view list still cover it?
}
// (the stand-alone launch, and the ends of runs that were not tracked: the run's result is read from the state)
__device__ void track_end_dev(DevState* st, int rgb, int tracked, float weight_mult, int commit, unsigned int* lctr)
{
    float Rc[9], tc[3], Rp[9], tp[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rc[k] = st->Rcurr[k]; Rp[k] = st->Rprev[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { tc[k] = st->tcurr[k]; tp[k] = st->tprev[k]; }
    track_end_vals(st, rgb, tracked, Rc, tc, Rp, tp, weight_mult, commit, lctr);
}
__global__ void k_track_end(DevState* st, int rgb, int tracked, float weight_mult, int commit, unsigned int* lctr)
{
    if (threadIdx.x != 0) return;
    track_end_dev(st, rgb, tracked, weight_mult, commit, lctr);
}

// K^-1 entries of the level the next iteration runs at, as the lone lane would compute them (1/fx, 1/fy, -cx/fx, -cy/fy in f64): computed on the host
// -- the same IEEE divisions -- so that four f64 divisions leave the serial part of every iteration
static inline KInv kinv_of(float fx, float fy, float cx, float cy)
{
    const double K0 = fx, K2 = cx, K4 = fy, K5 = cy;
    KInv k;
    k.i0 = 1.0 / K0; k.i4 = 1.0 / K4; k.i2 = -K2 / K0; k.i5 = -K5 / K4;
    return k;
}
__device__ __forceinline__ void warp_from_k(const double* M, float fx, float fy, float cx, float cy, const KInv& ki, float* krk, float* kt)
{
    const double K0 = fx, K2 = cx, K4 = fy, K5 = cy;
    const double I0 = ki.i0, I4 = ki.i4, I2 = ki.i2, I5 = ki.i5;
    double R3[9], tt[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R3[i * 3 + j] = M[j * 4 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) tt[i] = -(R3[i * 3 + 0] * M[3] + R3[i * 3 + 1] * M[7] + R3[i * 3 + 2] * M[11]);
    double KR[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        KR[j] = K0 * R3[j] + K2 * R3[6 + j];
        KR[3 + j] = K4 * R3[3 + j] + K5 * R3[6 + j];
        KR[6 + j] = R3[6 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        krk[i * 3 + 0] = (float)(KR[i * 3] * I0);
        krk[i * 3 + 1] = (float)(KR[i * 3 + 1] * I4);
        krk[i * 3 + 2] = (float)((KR[i * 3] * I2 + KR[i * 3 + 1] * I5) + KR[i * 3 + 2]);
    }
    kt[0] = (float)(K0 * tt[0] + K2 * tt[2]);
    kt[1] = (float)(K4 * tt[1] + K5 * tt[2]);
    kt[2] = (float)tt[2];
}

// The serial lane of one Gauss-Newton iteration: the combined 6x6 system (27 doubles in LDS, layout of the reference's 29-vector) -> LDLT -> SE(3)
// update of resultRt (RRt, in/out) -> current pose (Rc, tc) and the warp matrices of the next residual pass (krk, kt).
// EF/Utils/RGBDOdometry.cpp:552-583, OdometryProvider.h:73-93.  Shared by the two-launch form (gn_solve_block) and the persistent level kernel.
__device__ __forceinline__ void gn_serial_lane(const double* s_sys, int icp, double* RRt, const float* Rp, const float* tp, float nfx, float nfy, float ncx, float ncy, const KInv& ki,
                                               float* Rc, float* tc, float* krk, float* kt)
{
    double lA[36], lb[6];
    {
        int shift = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 7; ++j) {
                const double v = s_sys[shift++];
                if (j == 6) lb[i] = v;
                else { lA[j * 6 + i] = v; lA[i * 6 + j] = v; }
            }
    }
    double result[6];
    // With the ICP term the system is well conditioned and the unpivoted factorisation agrees with Eigen's pivoted LDLT to
    // ~1e-7 in the pose; the photometric term alone can be close to singular along unobservable directions, where the
    // pivoting decides the answer: that configuration takes the pivoted path the reference takes (EF/Utils/RGBDOdometry.cpp:552).
    if (icp) ldlt_nopivot<double, 6>(lA, lb, result, 1.0 / DBL_MAX);
    else ldlt_solve_n<double, 6>(lA, lb, result, 1.0 / DBL_MAX);
    // computeUpdateSE3, EF/Utils/OdometryProvider.h:73-93
    double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rr[9];
    rodrigues_d(&result[3], Rr);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) upd[r * 4 + c] = Rr[r * 3 + c];
    upd[3] = result[0]; upd[7] = result[1]; upd[11] = result[2];
    {   // upd * RRt for two rigid transforms (last rows 0 0 0 1): the terms a dense 4x4 product would multiply by those zeros are left out
        double N[12];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int c = 0; c < 3; c++) N[r * 4 + c] = (upd[r * 4] * RRt[c] + upd[r * 4 + 1] * RRt[4 + c]) + upd[r * 4 + 2] * RRt[8 + c];
            N[r * 4 + 3] = ((upd[r * 4] * RRt[3] + upd[r * 4 + 1] * RRt[7]) + upd[r * 4 + 2] * RRt[11]) + upd[r * 4 + 3];
        }
#pragma unroll
        for (int k = 0; k < 12; k++) RRt[k] = N[k];
    }
    float oR[9], ot[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) oR[r * 3 + c] = (float)RRt[r * 4 + c];
        ot[r] = (float)RRt[r * 4 + 3];
    }
    float iR[9], it[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) iR[r * 3 + c] = oR[c * 3 + r];
#pragma unroll
    for (int r = 0; r < 3; r++) it[r] = -(iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1] + iR[r * 3 + 2] * ot[2]);
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) Rc[r * 3 + c] = Rp[r * 3] * iR[c] + Rp[r * 3 + 1] * iR[3 + c] + Rp[r * 3 + 2] * iR[6 + c];
        tc[r] = Rp[r * 3] * it[0] + Rp[r * 3 + 1] * it[1] + Rp[r * 3 + 2] * it[2] + tp[r];
    }
    warp_from_k(RRt, nfx, nfy, ncx, ncy, ki, krk, kt);   // intrinsics of the level the NEXT iteration runs at
}

// one Gauss-Newton iteration's host logic, EF/Utils/RGBDOdometry.cpp:461-583, run by the block that finishes last.
// What is serial -- the 6x6 factorisation, the pose update, the next warp matrices -- runs on lane 0; everything that is not sits on other lanes:
//   threads 0..28   fetch (and clear) the ICP and photometric total of "their" element, round both to f32 as the reference's reductions deliver
//                   them, and assemble the element of the combined system lastA / lastb (:547-565) -- lane 0 then reads 27 finished doubles;
//   wave 1          the diagnostics (error norms, counts; on the run's last iteration also lastA / lastb / the 2 x 29 sums: 130 stores).
// In-kernel stamps of the previous form (everything on lane 0): 8.6k cycles serial; see DESIGN.md section 6.
__device__ __forceinline__ void gn_solve_block(DevState* st, double* __restrict__ icp_acc, double* __restrict__ rgb_acc,
                           const int* __restrict__ res_partials, int res_blocks, int icp, int rgb, float icp_weight, float nfx, float nfy, float ncx, float ncy, const KInv& ki,
                           int* __restrict__ res_total = nullptr, int final_iter = 1, int end_run = 0, float weight_mult = 1.f, int commit = 1, unsigned int* lctr = nullptr,
                           const double* __restrict__ rrt_src = nullptr)
{
    __shared__ double s_sys[27];      // combined system: index = position in the reference's 29-vector (upper triangle of A row by row, b in column 6)
    __shared__ float s_oi[29], s_or[29];
    __shared__ int s_res[2];
    // the pose state the serial part starts from: written by the previous launch, so it can be fetched together with the totals
    // (one memory round trip instead of two on the critical path of every iteration)
    double RRt[16];
    float Rp[9], tp[3];
    if (rrt_src) {   // (option gn_prologue: the increment of the iteration before lives in DevState::gnp_RRt, twelve entries)
#pragma unroll
        for (int k = 0; k < 12; k++) RRt[k] = rrt_src[k];
        RRt[12] = 0.0; RRt[13] = 0.0; RRt[14] = 0.0; RRt[15] = 1.0;
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) RRt[k] = st->resultRt[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) Rp[k] = st->Rprev[k];
#pragma unroll
    for (int k = 0; k < 3; k++) tp[k] = st->tprev[k];
    if (threadIdx.x < 29) {
        const int k = threadIdx.x;
        const double ti = acc_total(icp_acc, k), tr = acc_total(rgb_acc, k);
        acc_clear(icp_acc, k);
        acc_clear(rgb_acc, k);
        range_note(st, icp && range_exceeded7(0, k, ti), rgb && range_exceeded7(1, k, tr));
        const float oi = icp ? (float)ti : 0.f, orr = rgb ? (float)tr : 0.f;
        s_oi[k] = oi; s_or[k] = orr;
        if (k < 27) {
            // lastA = A_rgb + w*w*A_icp, lastb = b_rgb + w*b_icp (EF/Utils/RGBDOdometry.cpp:547-565); column 6 of a row is its b entry
            const double wgt = icp_weight;
            const double wa = wgt * wgt, wb = wgt;
            const bool is_b = (k == 6) | (k == 12) | (k == 17) | (k == 21) | (k == 24) | (k == 26);
            const double vi = (double)oi, vr = (double)orr;
            s_sys[k] = (icp && rgb) ? vr + (is_b ? wb : wa) * vi : (icp ? vi : vr);
        }
    }
    if (res_total) {   // totals accumulated by the residual pass; re-armed (zeroed) for the next iteration
        if (threadIdx.x == 64) {
            s_res[0] = __hip_atomic_load(&res_total[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_res[1] = __hip_atomic_load(&res_total[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&res_total[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&res_total[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (threadIdx.x < 2) s_res[threadIdx.x] = 0;
        __syncthreads();
        int c0 = 0, c1 = 0;
        for (int b = threadIdx.x; b < res_blocks; b += 256) {
            const int2 r0 = reinterpret_cast<const int2*>(res_partials)[b];
            c0 += r0.x; c1 += r0.y;
        }
        c0 = wave_sum_i(c0);
        c1 = wave_sum_i(c1);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&s_res[0], c0); atomicAdd(&s_res[1], c1); }
    }
#ifdef IFX_STAMPS
    long long ts_a = clock64(), ts_b = ts_a;
#endif
    __syncthreads();
#ifdef IFX_STAMPS
    long long ts_c = clock64();
#endif
    if (threadIdx.x >= 64 && threadIdx.x < 128) {   // ---- wave 1: diagnostics, off the serial lane
        const int t = threadIdx.x - 64;
        if (t == 0) {
            const int rgbSize = rgb ? s_res[0] : 0, sigma = rgb ? s_res[1] : 0;
            st->rgb_count = rgbSize; st->rgb_sigma = sigma;
            st->lastRGBError = (float)(sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));
            st->lastRGBCount = (float)rgbSize;
            if (icp) { st->lastICPError = sqrtf(s_oi[27]) / s_oi[28]; st->lastICPCount = s_oi[28]; }
        }
        if (final_iter) {   // lastA / lastb / the 2 x 29 sums describe the run's LAST iteration (getCovariance, diagnostics)
            if (t < 29) { st->icp29[t] = s_oi[t]; st->rgb29[t] = s_or[t]; }
            if (t < 36) {
                const int i = t / 6, j = t - 6 * i, lo = i < j ? i : j, hi = i < j ? j : i;
                st->lastA[t] = s_sys[lo * 7 - (lo * (lo - 1)) / 2 + (hi - lo)];
            }
            if (t < 6) st->lastb[t] = s_sys[t * 7 - (t * (t - 1)) / 2 + (6 - t)];
        }
        return;
    }
    if (threadIdx.x != 0) return;
    // ---- serial part on one lane.  Every input is in registers or LDS and every output is stored once at the end.
    float Rc[9], tc[3], krk[9], kt[3];
    gn_serial_lane(s_sys, icp, RRt, Rp, tp, nfx, nfy, ncx, ncy, ki, Rc, tc, krk, kt);
    // ---- stores
#pragma unroll
    for (int k = 0; k < 16; k++) st->resultRt[k] = RRt[k];
#pragma unroll
    for (int k = 0; k < 9; k++) { st->Rcurr[k] = Rc[k]; st->krkinv[k] = krk[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { st->tcurr[k] = tc[k]; st->kt[k] = kt[k]; }
    // the run's last iteration also ends the run (:587-603, pose write-back, velocity weighting, view-list decision): the same lane, one launch less per frame
    if (end_run) track_end_vals(st, rgb, 1, Rc, tc, Rp, tp, weight_mult, commit, lctr);
#ifdef IFX_STAMPS
    { long long ts_d = clock64(); st->dbg[4] += ts_d - ts_c; st->dbg[6] += ts_b - ts_a; st->dbg[7] += ts_c - ts_b; st->dbg[3] += ts_a; }
#endif
}


// Photometric reduction + (in the block that finishes last) the final sums and the 6x6 solve: the
// second and last launch of a Gauss-Newton iteration.  Hand-off between blocks follows the
// agent-scope release / acquire recipe of cdna_hip_programming.md section 6 G16 (counter form):
// plain partial stores -> s_waitcnt vmcnt(0) -> barrier -> release fence -> ticket; the block that
// draws the last ticket acquires and reads every partial.
struct StepArgs {
    const Corres8* corres;
    const float* cloud;
    float fx, fy, sobelScale;
    const int16_t *dIdx, *dIdy;
    int w, h, nb, nb_icp, nb_res;
    int icp, rgb;
    float icp_weight, nfx, nfy, ncx, ncy;
    KInv ki;   // K^-1 of (nfx, nfy, ncx, ncy)
    int check_skip;
    int final_iter;   // last Gauss-Newton iteration of the run: leaves lastA / lastb / the sums in DevState
    int end_run, commit;   // ... and ends the run in the same lane (track_end_dev)
    float weight_mult;
    unsigned int* lctr;
    int pro, pro_k;        // option gn_prologue: 0 = round 3's form; 1 = sums only (the next launch's prologue solves); 2 = the run's last iteration (last-block form on the parity buffers).  pro_k: position in the two-launch tail
};
template <bool CHECK_SKIP>
__global__ __launch_bounds__(RED_THREADS) void k_rgb_step_solve(DevState* st, int nb, int rgb, int w, int h, StepArgs a)
{
    __builtin_assume(st != nullptr);
    if (CHECK_SKIP && st->skip) return;   // (model-to-model instance only) uniform over the grid: the last-block ticket stays armed
    // hand-off words and accumulator rows: behind the state pointer, which arrives preloaded.  gn_prologue: the buffers of this iteration's parity
    const int par = a.pro_k & 1;
    int* const res_total = a.pro ? st->gnp_res + par * 16 : st->gn_res;
    unsigned int* const ticket = &st->gn_ticket;
    double* const icp_acc = a.pro ? st->gnp_acc + (size_t)(par * 2) * IFX_ACC_REPL * IFX_ACC_STRIDE : st->gn_acc;
    double* const rgb_acc = icp_acc + IFX_ACC_REPL * IFX_ACC_STRIDE;
    __shared__ int s_last;
    if (a.pro && blockIdx.x == 0) {   // the other parity was last read by the prologues of this iteration's first launch: clean again for the next iteration's sums
        double* const other = st->gnp_acc + (size_t)((1 - par) * 2) * IFX_ACC_REPL * IFX_ACC_STRIDE;
        if (threadIdx.x < 64) {   // run-time guard of the exact sums (range_exceeded7), on the totals of the iteration before, here where they are cleared: the prologues that
            const int kind = threadIdx.x >> 5, k = threadIdx.x & 31;   // consumed them sit in a launch that has no register to spare (it spilled with the test in it)
            if (k < 28 && range_exceeded7(kind, k, acc_total(other + (size_t)kind * IFX_ACC_REPL * IFX_ACC_STRIDE, k))) atomicAdd(&st->range_exceeded, 1);
        }
        __syncthreads();
        __hip_atomic_store(&other[threadIdx.x], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 2 x REPL x STRIDE = 256 doubles = RED_THREADS
        if (threadIdx.x < 2) __hip_atomic_store(&st->gnp_res[(1 - par) * 16 + threadIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef IFX_STAMPS
    long long t0 = clock64();
#endif
    if (rgb) rgb_step_body(blockIdx.x, nb, a.corres, 0.f, nullptr, 0, a.cloud, a.fx, a.fy, a.dIdx, a.dIdy, a.sobelScale, w, h, rgb_acc, res_total);
    if (a.pro == 1) return;   // sums only: every block of the next launch reads the totals and solves (gn_prologue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef IFX_STAMPS
    long long t1 = clock64();
#endif
    if (threadIdx.x == 0) {   // this block's atomic adds were performed at the memory side: every wave drained vmcnt before the barrier
        unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (unsigned int)(nb - 1));
    }
    __syncthreads();
    if (!s_last) return;
    // Everything the other blocks of this launch produced reaches this block through agent-scope atomics and agent-scope
    // (sc1) loads -- the accumulator rows, the residual totals, the ticket -- which are performed at / served from the
    // memory side: no acquire (buffer_inv, ~1.7 us) is needed.
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch (stream order)
#ifdef IFX_STAMPS
    long long t2 = clock64();
#endif
    gn_solve_block(st, icp_acc, rgb_acc, nullptr, 0, a.icp, rgb, a.icp_weight, a.nfx, a.nfy, a.ncx, a.ncy, a.ki, res_total, a.final_iter, a.end_run, a.weight_mult, a.commit, a.lctr,
                   (a.pro && a.pro_k > 0) ? st->gnp_RRt[1 - par] : (const double*)nullptr);
#ifdef IFX_STAMPS
    if (threadIdx.x == 0) { long long t3 = clock64(); st->dbg[0] += t2 - t0; st->dbg[1] += t3 - t2; st->dbg[2] += 1; st->dbg[5] += t1 - t0; st->dbg[3] -= t2; g_dbg2[0] += s_dbg_blk[0]; g_dbg2[1] += s_dbg_blk[1]; g_dbg2[2] += s_dbg_blk[2]; }
#endif
}


// ---- option own_track_rows (a sharded map whose library holds the communicator; SURVEY.md 8e-i: "tracking is a sum over independent pixels -> shard the image, exchange =
// all-reduce of the 29 + 29 sums"): rank r of G runs the two REDUCTIONS of an iteration over the pixel blocks b with b mod G == r (virtual block ids: the bodies are the ones
// above, untouched), the accumulator rows are SUM-all-reduced in f64 -- exact sums of grid-valued terms: any order, any partition gives the same bits -- and one workgroup
// solves.  The residual pass (it writes the correspondence records the photometric step reads back, by other blocks) stays whole on every rank.  Three launches and two
// collectives per iteration instead of two launches: priced in DESIGN.md section 7 -- a measured loss at every BASELINE size, which is why the tracker stays replicated.
__global__ __launch_bounds__(RED_THREADS, 4) void k_icp_residual_rows(const DevState* __restrict__ st, int nb_icp, int w, int h, double* __restrict__ gacc, int* __restrict__ gres, PairArgs a,
                                                                      int srank, int sn)
{
    __builtin_assume(st != nullptr);
    if ((int)blockIdx.x < nb_icp) {
        IcpArgs ia;   // unused when st != nullptr
        icp_body<false, true, false>(blockIdx.x * sn + srank, nb_icp * sn, st, ia, a.vmap_curr, a.nmap_curr, a.vmap_prev, a.nmap_prev, a.fx, a.fy, a.cx, a.cy, a.distThres, a.angleThres, w, h, gacc);
    } else {
        ResArgs ra;
        residual_body<true, false>(blockIdx.x - nb_icp, a.nb_res, st, ra, a.minScale, a.dIdx, a.dIdy, a.lastDepth, a.nextDepth, a.lastImage, a.nextImage, a.corres, a.maxDepthDelta, w, h, nullptr, gres);
    }
}
__global__ __launch_bounds__(RED_THREADS) void k_rgb_step_rows(DevState* st, int nb, int w, int h, StepArgs a, int srank, int sn)
{
    __builtin_assume(st != nullptr);
    rgb_step_body(blockIdx.x * sn + srank, nb * sn, a.corres, 0.f, nullptr, 0, a.cloud, a.fx, a.fy, a.dIdx, a.dIdy, a.sobelScale, w, h, st->gn_acc + IFX_ACC_REPL * IFX_ACC_STRIDE, st->gn_res);
}
__global__ __launch_bounds__(RED_THREADS) void k_gn_solve_rows(DevState* st, StepArgs a)
{
    __builtin_assume(st != nullptr);
    gn_solve_block(st, st->gn_acc, st->gn_acc + IFX_ACC_REPL * IFX_ACC_STRIDE, nullptr, 0, a.icp, a.rgb, a.icp_weight, a.nfx, a.nfy, a.ncx, a.ncy, a.ki, st->gn_res, a.final_iter, a.end_run, a.weight_mult,
                   a.commit, a.lctr, (const double*)nullptr);
}

// (Round 2 measured it slower than the two-launch form at levels 0 and 1 -- the meetings of 300-1200 blocks cost more than launch boundaries -- and FASTER at the
// coarsest level, 75 blocks: 12.5 against 13.9 us per iteration.  Round 3 uses it there: option gn_persist is a bit per pyramid level, default 4.)
// ======================================================================= persistent Gauss-Newton level
// ALL iterations of one pyramid level in ONE launch.  The two-launch form above pays per iteration two launch boundaries, two argument / state
// prologues, a round trip of the correspondence records through memory and the last block's ticket: ~14 us even at 160 x 120, where the
// arithmetic is nothing.  Here a thread keeps "its" pixels for the whole level:
//   * everything that depends only on the frame -- vertex, normal, depth, intensity, gradients, the 4x4 validity test -- is fetched ONCE into
//     registers; an iteration only gathers from the model maps (one memory round trip: ICP correspondence, residual taps and the point-cloud
//     entry of the photometric step are all requested together);
//   * the correspondence record never leaves the thread (the residual pass and the step work on the same pixel);
//   * the blocks meet at two grid barriers per iteration (after the ICP sums + residual totals, after the photometric sums): one agent-scope
//     atomic per block and a polling load -- no data crosses blocks except through the exact accumulator rows (memory-side atomics, read back
//     with agent-scope loads), so no cache write-back / invalidate is needed anywhere;
//   * every block then solves the 6x6 system itself (the same totals give the same bits): no broadcast of the pose, no third meeting.
// Accumulator rows and residual totals are double-buffered by iteration parity; block 0 clears a buffer one barrier after its last reader.
// The grid is sized to be co-resident (ifx_tracker_init queries the occupancy); a barrier that would spin for seconds gives up and raises
// DevState::gn_timeout instead of hanging the GPU.
// Same per-pixel arithmetic and the same exact sums as k_icp_residual / k_rgb_step_solve: bit-identical poses (tests/test_gpu_parity.py).
struct LevelArgs {
    const float *vmap_curr, *nmap_curr, *vmap_prev, *nmap_prev;
    const int16_t *dIdx, *dIdy;
    const float *lastDepth, *nextDepth;
    const uint8_t *lastImage, *nextImage;
    const float* cloud;
    Corres8* corres;              // scratch of the one-workgroup fallback (the level's correspondence records)
    float fx, fy, cx, cy, distThres, angleThres, minScale, maxDepthDelta, sobelScale;
    int w, h, iters, icp, rgb;
    float icp_weight;
    float nfx, nfy, ncx, ncy;     // intrinsics of the level the iteration AFTER this level's last one runs at
    KInv ki_same, ki_next;
    int acc_base;                 // iterations of the run before this level (parity of the double buffers continues across levels)
    int level;                    // barrier word
    int final_level;              // the run's last level: its last iteration leaves the diagnostics and ends the run
    int commit;
    float weight_mult;
    unsigned int* lctr;
};

// The level on ONE workgroup (enqueued behind every k_gn_level launch; returns at its first instruction unless a meeting of that launch failed): the bodies of the two-launch form with a grid of one, the sums through the single-buffer
// accumulator rows (DevState::gn_acc / gn_res, which nothing else uses while this kernel runs), agent-scope fences between the phases (the block reads back what it
// wrote: correspondence records, residual totals).  Same rows, same exact sums, same solve: the level's result is the one the meetings would have produced.
__global__ __launch_bounds__(RED_THREADS) void k_gn_level_solo(DevState* st, LevelArgs a)
{
    if (st->gn_done_seq == a.level + 1) return;   // the meetings happened (every healthy launch): nothing to do
    __shared__ float s_pose[24];
    __shared__ double s_RRt[16];
    __shared__ double s_sys[27];
    __shared__ float s_oi[29], s_or[29];
    __shared__ int s_res[2];
    const int tid = threadIdx.x;
    float Rp[9], tpv[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Rp[k] = st->Rprev[k];
#pragma unroll
    for (int k = 0; k < 3; k++) tpv[k] = st->tprev[k];
    if (tid < 9) { s_pose[tid] = st->Rcurr[tid]; s_pose[12 + tid] = st->krkinv[tid]; }
    if (tid < 3) { s_pose[9 + tid] = st->tcurr[tid]; s_pose[21 + tid] = st->kt[tid]; }
    if (tid < 16) s_RRt[tid] = st->resultRt[tid];
    if (tid == 0) atomicAdd(&st->gn_timeout, 1);
    __syncthreads();
    double* const icp_acc = st->gn_acc, * const rgb_acc = st->gn_acc + IFX_ACC_REPL * IFX_ACC_STRIDE;
    int* const gres = st->gn_res;
    Corres8* const corres = a.corres;
    for (int it = 0; it < a.iters; it++) {
        IcpArgs ia;
        ResArgs ra;
#pragma unroll
        for (int k = 0; k < 9; k++) { ia.Rcurr[k] = s_pose[k]; ia.Rprev_inv[k] = st->Rprev_inv[k]; ra.krkinv[k] = s_pose[12 + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { ia.tcurr[k] = s_pose[9 + k]; ia.tprev[k] = tpv[k]; ra.kt[k] = s_pose[21 + k]; }
        icp_body<false, false>(0, 1, nullptr, ia, a.vmap_curr, a.nmap_curr, a.vmap_prev, a.nmap_prev, a.fx, a.fy, a.cx, a.cy, a.distThres, a.angleThres, a.w, a.h, icp_acc);
        residual_body<false>(0, 1, nullptr, ra, a.minScale, a.dIdx, a.dIdy, a.lastDepth, a.nextDepth, a.lastImage, a.nextImage, corres, a.maxDepthDelta, a.w, a.h, st->gn_pad, gres);
        __threadfence();
        __syncthreads();
        rgb_step_body(0, 1, corres, 0.f, nullptr, 0, a.cloud, a.fx, a.fy, a.dIdx, a.dIdy, a.sobelScale, a.w, a.h, rgb_acc, gres);
        __threadfence();
        __syncthreads();
        const bool last_it = it == a.iters - 1;
        if (tid < 29) {
            const double ti = acc_total(icp_acc, tid), tr = acc_total(rgb_acc, tid);
            acc_clear(icp_acc, tid);
            acc_clear(rgb_acc, tid);
            range_note(st, range_exceeded7(0, tid, ti), range_exceeded7(1, tid, tr));
            const float oi = (float)ti, orr = (float)tr;
            s_oi[tid] = oi; s_or[tid] = orr;
            if (tid < 27) {
                const double wgt = a.icp_weight;
                const double wa = wgt * wgt, wb = wgt;
                const bool is_b = (tid == 6) | (tid == 12) | (tid == 17) | (tid == 21) | (tid == 24) | (tid == 26);
                s_sys[tid] = (double)orr + (is_b ? wb : wa) * (double)oi;
            }
        }
        if (tid == 64) {
            s_res[0] = __hip_atomic_load(&gres[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_res[1] = __hip_atomic_load(&gres[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gres[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gres[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (last_it && a.final_level && tid >= 64 && tid < 128) {   // diagnostics of the run's last iteration
            const int t = tid - 64;
            if (t == 0) {
                const int rgbSize = s_res[0], sigma = s_res[1];
                st->rgb_count = rgbSize; st->rgb_sigma = sigma;
                st->lastRGBError = (float)(sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));
                st->lastRGBCount = (float)rgbSize;
                st->lastICPError = sqrtf(s_oi[27]) / s_oi[28]; st->lastICPCount = s_oi[28];
            }
            if (t < 29) { st->icp29[t] = s_oi[t]; st->rgb29[t] = s_or[t]; }
            if (t < 36) {
                const int i = t / 6, j = t - 6 * i, lo = i < j ? i : j, hi = i < j ? j : i;
                st->lastA[t] = s_sys[lo * 7 - (lo * (lo - 1)) / 2 + (hi - lo)];
            }
            if (t < 6) st->lastb[t] = s_sys[t * 7 - (t * (t - 1)) / 2 + (6 - t)];
        }
        if (tid == 0) {
            double RRt[16];
#pragma unroll
            for (int k = 0; k < 16; k++) RRt[k] = s_RRt[k];
            float Rc[9], tcn[3], krkn[9], ktn[3];
            if (last_it) gn_serial_lane(s_sys, 1, RRt, Rp, tpv, a.nfx, a.nfy, a.ncx, a.ncy, a.ki_next, Rc, tcn, krkn, ktn);
            else gn_serial_lane(s_sys, 1, RRt, Rp, tpv, a.fx, a.fy, a.cx, a.cy, a.ki_same, Rc, tcn, krkn, ktn);
#pragma unroll
            for (int k = 0; k < 12; k++) s_RRt[k] = RRt[k];
#pragma unroll
            for (int k = 0; k < 9; k++) { s_pose[k] = Rc[k]; s_pose[12 + k] = krkn[k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) { s_pose[9 + k] = tcn[k]; s_pose[21 + k] = ktn[k]; }
            if (last_it) {
#pragma unroll
                for (int k = 0; k < 16; k++) st->resultRt[k] = RRt[k];
#pragma unroll
                for (int k = 0; k < 9; k++) { st->Rcurr[k] = Rc[k]; st->krkinv[k] = krkn[k]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { st->tcurr[k] = tcn[k]; st->kt[k] = ktn[k]; }
                if (a.final_level) track_end_vals(st, 1, 1, Rc, tcn, Rp, tpv, a.weight_mult, a.commit, a.lctr);
                st->gn_done_seq = a.level + 1;
            }
        }
        __syncthreads();
    }
}

// Grid barrier.  Memory-side atomics on ONE line serialise at ~8 ns each, and so do the polling loads: 400 blocks arriving at and polling one word
// cost ~12 us per meeting.  Here a block arrives (fire and forget) at one of GN_BAR_SUB counters, each on a line of its own, and wave 0 polls
// all of them with one vector load (lane s reads counter s): a meeting is one atomic + one or two polling round trips.
#define GN_BAR_SUB 32
// Returns false when the meeting did not happen: this block gave up after `limit` polls (it then raises the run's abort word, which every other block polls with
// the same vector load -- lane GN_BAR_SUB -- so that they leave at once instead of spinning out their own limits), or another block had already given up.  The caller
// leaves the level; the one-workgroup kernel behind this launch re-runs it (k_gn_level_solo).  Co-residency of the grid is what a meeting needs: the launch is sized from the occupancy query with an
// eighth to spare, but other streams and other handles share the GPU, so a grid that does not fit must cost time, never the pose.
__device__ __forceinline__ bool gn_grid_barrier(unsigned int* bar, int k, int nb, DevState* st, int seq)
{
    __shared__ int s_bar_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's accumulator atomics have been performed at the memory side
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool absent = st->gn_fault != 0 && st->gn_fault == seq && blockIdx.x == 1;   // (test hook: a block that never arrives)
        if (lane == 0 && !absent) __hip_atomic_fetch_add(&bar[((int)blockIdx.x % GN_BAR_SUB) * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // sub-counter s is fed by the blocks with bid % GN_BAR_SUB == s
        const unsigned int mine = lane < GN_BAR_SUB ? (unsigned int)((nb - lane + GN_BAR_SUB - 1) / GN_BAR_SUB) : 0u;
        const unsigned int target = (unsigned int)k * mine;
        const int limit = st->gn_spin_limit > 0 ? st->gn_spin_limit : (1 << 21);
        const unsigned int* const word = lane < GN_BAR_SUB ? &bar[lane * 16] : (const unsigned int*)&st->gn_abort;
        int spin = 0;
        bool ok = true;
        for (;;) {
            const unsigned int v = lane <= GN_BAR_SUB ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__any(lane == GN_BAR_SUB && v != 0u)) { ok = false; break; }                 // somebody gave up
            if (__all(lane >= GN_BAR_SUB || v >= target)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spin > limit) {
                if (lane == 0) __hip_atomic_store(&st->gn_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        if (lane == 0) s_bar_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return s_bar_ok != 0;
}

template <int PX>
__global__ __launch_bounds__(RED_THREADS) void k_gn_level(DevState* st, LevelArgs a)
{
    // (ICP + photometric term, the reference's default: any other weighting takes the two-launch form, which also carries the pivoted solve)
    constexpr int ICP = 1, RGB = 1;
    const int N = a.w * a.h, nb = gridDim.x, bid = blockIdx.x, tid = threadIdx.x;
    __shared__ float s_pose[24];      // Rcurr 9, tcurr 3, krkinv 9, kt 3 of the iteration about to run
    __shared__ double s_RRt[16];
    __shared__ double s_sys[27];
    __shared__ float s_oi[29], s_or[29];
    __shared__ int s_res[2];
    __shared__ int s_cs[RED_WAVES][2];
    // ---- state of the run (constant over the level) and the pose the level starts from
    float Rprev_inv[9], Rp[9], tpv[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rprev_inv[k] = st->Rprev_inv[k]; Rp[k] = st->Rprev[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) tpv[k] = st->tprev[k];
    const v3 tp = v3m(tpv[0], tpv[1], tpv[2]);
    if (tid < 9) { s_pose[tid] = st->Rcurr[tid]; s_pose[12 + tid] = st->krkinv[tid]; }
    if (tid < 3) { s_pose[9 + tid] = st->tcurr[tid]; s_pose[21 + tid] = st->kt[tid]; }
    if (tid < 16) s_RRt[tid] = st->resultRt[tid];
    // ---- per-pixel constants: pixel u of a thread is (u * nb + bid) * 256 + tid (coalesced across lanes)
    v3 vcurr[PX], ncurr[PX];
    float d1[PX], nif[PX];
    short gx[PX], gy[PX];
    bool cand0[PX];
    int xy[PX];
    const int border = 16;
#pragma unroll
    for (int u = 0; u < PX; u++) {
        const int p = (u * nb + bid) * RED_THREADS + tid;
        const bool in = p < N;
        const int pp = in ? p : 0;
        const int i = pp / a.w, j0 = pp - i * a.w;
        xy[u] = (i << 16) | j0;
        vcurr[u] = v3m(a.vmap_curr[pp], a.vmap_curr[pp + N], a.vmap_curr[pp + 2 * N]);
        ncurr[u] = v3m(a.nmap_curr[pp], a.nmap_curr[pp + N], a.nmap_curr[pp + 2 * N]);
        if (!in) vcurr[u].x = qnan_f();
        // RGBResidual's own-pixel tests (EF/Cuda/reduce.cu:739-790): 16-px border, 4x4 non-zero block, gradient gate, valid depth
        const bool ok = in && i >= border && i < a.h - border && j0 >= border && j0 < a.w - border && j0 < a.w - 5 && i < a.h - 1;
        const int ci = ok ? i : 16, cj = ok ? j0 : 16;
        bool valid = true;
#pragma unroll
        for (int q = -2; q < 2; q++) {
            uint32_t r4;
            __builtin_memcpy(&r4, a.nextImage + (ci + q) * a.w + cj - 2, 4);
            valid = valid & (((r4 - 0x01010101u) & ~r4 & 0x80808080u) == 0u);
        }
        gx[u] = a.dIdx[pp]; gy[u] = a.dIdy[pp];
        const float mTwo = (float)((gx[u] * gx[u]) + (gy[u] * gy[u]));
        d1[u] = a.nextDepth[pp];
        nif[u] = (float)a.nextImage[pp];
        cand0[u] = ok & valid & (mTwo >= a.minScale) & !(d1[u] != d1[u]);
    }
    __syncthreads();
    unsigned int* bar = &st->gn_bar[a.level * GN_BAR_SUB * 16];
    bool failed = false;
    for (int it = 0; it < a.iters; it++) {
        const int par = (a.acc_base + it) & 1;
        double* const icp_acc = st->gn_acc2 + (size_t)(par * 2 + 0) * IFX_ACC_REPL * IFX_ACC_STRIDE;
        double* const rgb_acc = st->gn_acc2 + (size_t)(par * 2 + 1) * IFX_ACC_REPL * IFX_ACC_STRIDE;
        int* const gres = st->gn_res2 + par * 16;
        float Rcurr[9], krk[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { Rcurr[k] = s_pose[k]; krk[k] = s_pose[12 + k]; }
        const v3 tc = v3m(s_pose[9], s_pose[10], s_pose[11]);
        const float kt0 = s_pose[21], kt1 = s_pose[22], kt2 = s_pose[23];
        // ---- phase A: projections, then every gather of the iteration in one batch
        v3 vcurr_g[PX], vprev[PX], nprev[PX];
        bool inb[PX], cand[PX];
        int gj[PX];
        float td1[PX], d0[PX], lif[PX], X[PX], Y[PX], Z[PX];
#pragma unroll
        for (int u = 0; u < PX; u++) {
            // ICPReduction::search, EF/Cuda/reduce.cu:257-300
            vcurr_g[u] = mulp(Rcurr, vcurr[u]) + tc;
            const v3 vcurr_cp = mulp(Rprev_inv, vcurr_g[u] - tp);
            const int ux = f2i_rn(vcurr_cp.x * a.fx / vcurr_cp.z + a.cx);
            const int uy = f2i_rn(vcurr_cp.y * a.fy / vcurr_cp.z + a.cy);
            inb[u] = !(vcurr[u].x != vcurr[u].x) && !(ux < 0 || uy < 0 || ux >= a.w || uy >= a.h || vcurr_cp.z < 0);
            const int j = inb[u] ? uy * a.w + ux : 0;
            // RGBResidual warp, EF/Cuda/reduce.cu:791-810
            const int y = xy[u] >> 16, x = xy[u] & 0xFFFF;
            td1[u] = (float)(d1[u] * (krk[6] * x + krk[7] * y + krk[8]) + kt2);
            const int u0 = f2i_rn((d1[u] * (krk[0] * x + krk[1] * y + krk[2]) + kt0) / td1[u]);
            const int v0 = f2i_rn((d1[u] * (krk[3] * x + krk[4] * y + krk[5]) + kt1) / td1[u]);
            cand[u] = cand0[u] & ((u0 >= 0) & (v0 >= 0) & (u0 < a.w) & (v0 < a.h));
            gj[u] = cand[u] ? v0 * a.w + u0 : 0;
            vprev[u] = v3m(a.vmap_prev[j], a.vmap_prev[j + N], a.vmap_prev[j + 2 * N]);
            nprev[u] = v3m(a.nmap_prev[j], a.nmap_prev[j + N], a.nmap_prev[j + 2 * N]);
            d0[u] = a.lastDepth[gj[u]];
            lif[u] = (float)a.lastImage[gj[u]];
            const int g3 = gj[u] * 3;   // the step's point-cloud entry lives at the correspondence: known before the taps come back
            X[u] = a.cloud[g3]; Y[u] = a.cloud[g3 + 1]; Z[u] = a.cloud[g3 + 2];
        }
        double acc[29];
#pragma unroll
        for (int k = 0; k < 29; k++) acc[k] = 0.0;
        int cnt = 0, sig = 0;
        bool hit[PX];
        float diff[PX];
#pragma unroll
        for (int u = 0; u < PX; u++) {
            float row[7] = {0, 0, 0, 0, 0, 0, 0};
            bool found = false;
            if (ICP && inb[u]) {   // EF/Cuda/reduce.cu:302-387
                const v3 ncurr_g = mulp(Rcurr, ncurr[u]);
                const float dist = norm(vprev[u] - vcurr_g[u]);
                const float sine = norm(cross(ncurr_g, nprev[u]));
                found = (sine < a.angleThres && dist <= a.distThres && !(ncurr[u].x != ncurr[u].x) && !(nprev[u].x != nprev[u].x));
                if (found) {
                    const v3 s_cp = mulp(Rprev_inv, vcurr_g[u] - tp);
                    const v3 d_cp = mulp(Rprev_inv, vprev[u] - tp);
                    const v3 n_cp = mulp(Rprev_inv, nprev[u]);
                    const v3 c = cross(s_cp, n_cp);
                    row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
                    row[3] = c.x; row[4] = c.y; row[5] = c.z;
                    row[6] = dot(n_cp, s_cp - d_cp);
                }
            }
            if (ICP) products7<0>(row, found, acc);
            // EF/Cuda/reduce.cu:811-842
            hit[u] = RGB && (cand[u] & (d0[u] > 0) & (fabsf(td1[u] - d0[u]) <= a.maxDepthDelta) & (lif[u] != 0.f));
            diff[u] = nif[u] - lif[u];
            cnt += hit[u] ? 1 : 0;
            sig += hit[u] ? (int)(diff[u] * diff[u]) : 0;
        }
        if (ICP) block_sum_exact<29>(acc, icp_acc, bid % IFX_ACC_REPL);
        {
            const int lane = tid & 63, wid = tid >> 6;
            cnt = wave_sum_i(cnt);
            sig = wave_sum_i(sig);
            if (lane == 0) { s_cs[wid][0] = cnt; s_cs[wid][1] = sig; }
            __syncthreads();
            if (tid < 2) {
                int s2 = 0;
                for (int wv = 0; wv < RED_WAVES; wv++) s2 += s_cs[wv][tid];
                if (s2) atomicAdd(&gres[tid], s2);
            }
        }
        if (!gn_grid_barrier(bar, 2 * it + 1, nb, st, a.acc_base * 2 + 2 * it + 1)) { failed = true; break; }
        if (bid == 0) {   // the other parity's buffers were last read in the solve of the previous iteration: every block is past it now
            double* const other = st->gn_acc2 + (size_t)((1 - par) * 2) * IFX_ACC_REPL * IFX_ACC_STRIDE;
            __hip_atomic_store(&other[tid], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 2 x REPL x STRIDE = 256 doubles
            if (tid < 2) __hip_atomic_store(&st->gn_res2[(1 - par) * 16 + tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- phase B: the photometric rows (RGBReduction, EF/Cuda/reduce.cu:494-619) from registers
        const int rcnt = __hip_atomic_load(&gres[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), rsig = __hip_atomic_load(&gres[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (RGB) {
            const float q = (float)rsig / (float)rcnt;
            const float sigma = (float)sqrt((double)((q == 0) ? 1 : rcnt));   // the reference's precedence quirk (EF/Utils/RGBDOdometry.cpp:461)
#pragma unroll
            for (int k = 0; k < 29; k++) acc[k] = 0.0;
#pragma unroll
            for (int u = 0; u < PX; u++) {
                float row[7] = {0, 0, 0, 0, 0, 0, 0};
                const bool found = hit[u];
                if (found) {
                    float wgt = sigma + fabsf(diff[u]);
                    wgt = wgt > 1.19209290E-07F ? 1.0f / wgt : 1.0f;
                    if (sigma == -1) wgt = 1;
                    row[6] = -wgt * diff[u];
                    const float invz = (float)(1.0 / Z[u]);
                    const float dI_dx = wgt * a.sobelScale * gx[u];
                    const float dI_dy = wgt * a.sobelScale * gy[u];
                    const float v0 = dI_dx * a.fx * invz;
                    const float v1 = dI_dy * a.fy * invz;
                    const float v2 = -(v0 * X[u] + v1 * Y[u]) * invz;
                    row[0] = v0; row[1] = v1; row[2] = v2;
                    row[3] = -Z[u] * v1 + Y[u] * v2;
                    row[4] = Z[u] * v0 - X[u] * v2;
                    row[5] = -Y[u] * v0 + X[u] * v1;
                }
                products7<1>(row, found, acc);
            }
            block_sum_exact<29>(acc, rgb_acc, bid % IFX_ACC_REPL);
        }
        if (!gn_grid_barrier(bar, 2 * it + 2, nb, st, a.acc_base * 2 + 2 * it + 2)) { failed = true; break; }
        // ---- every block: totals -> combined system -> solve -> the next iteration's pose in LDS
        const bool last_it = it == a.iters - 1;
        if (tid < 29) {
            const double ti = acc_total(icp_acc, tid), tr = acc_total(rgb_acc, tid);
            if (bid == 0) range_note(st, ICP && range_exceeded7(0, tid, ti), RGB && range_exceeded7(1, tid, tr));
            const float oi = ICP ? (float)ti : 0.f, orr = RGB ? (float)tr : 0.f;
            s_oi[tid] = oi; s_or[tid] = orr;
            if (tid < 27) {
                const double wgt = a.icp_weight;
                const double wa = wgt * wgt, wb = wgt;
                const bool is_b = (tid == 6) | (tid == 12) | (tid == 17) | (tid == 21) | (tid == 24) | (tid == 26);
                const double vi = (double)oi, vr = (double)orr;
                s_sys[tid] = (ICP != 0 && RGB != 0) ? vr + (is_b ? wb : wa) * vi : (ICP ? vi : vr);
            }
        }
        if (tid == 64) { s_res[0] = rcnt; s_res[1] = rsig; }
        __syncthreads();
        if (bid == 0 && last_it && a.final_level && tid >= 64 && tid < 128) {   // diagnostics of the run's last iteration (block 0, off the serial lane)
            const int t = tid - 64;
            if (t == 0) {
                const int rgbSize = RGB ? s_res[0] : 0, sigma = RGB ? s_res[1] : 0;
                st->rgb_count = rgbSize; st->rgb_sigma = sigma;
                st->lastRGBError = (float)(sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));
                st->lastRGBCount = (float)rgbSize;
                if (ICP) { st->lastICPError = sqrtf(s_oi[27]) / s_oi[28]; st->lastICPCount = s_oi[28]; }
            }
            if (t < 29) { st->icp29[t] = s_oi[t]; st->rgb29[t] = s_or[t]; }
            if (t < 36) {
                const int i = t / 6, j = t - 6 * i, lo = i < j ? i : j, hi = i < j ? j : i;
                st->lastA[t] = s_sys[lo * 7 - (lo * (lo - 1)) / 2 + (hi - lo)];
            }
            if (t < 6) st->lastb[t] = s_sys[t * 7 - (t * (t - 1)) / 2 + (6 - t)];
        }
        if (tid == 0 && !(last_it && bid != 0)) {   // (the level's last solve is only block 0's to publish)
            double RRt[16];
#pragma unroll
            for (int k = 0; k < 16; k++) RRt[k] = s_RRt[k];
            float Rc[9], tcn[3], krkn[9], ktn[3];
            if (last_it) gn_serial_lane(s_sys, ICP, RRt, Rp, tpv, a.nfx, a.nfy, a.ncx, a.ncy, a.ki_next, Rc, tcn, krkn, ktn);
            else gn_serial_lane(s_sys, ICP, RRt, Rp, tpv, a.fx, a.fy, a.cx, a.cy, a.ki_same, Rc, tcn, krkn, ktn);
#pragma unroll
            for (int k = 0; k < 12; k++) s_RRt[k] = RRt[k];
#pragma unroll
            for (int k = 0; k < 9; k++) { s_pose[k] = Rc[k]; s_pose[12 + k] = krkn[k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) { s_pose[9 + k] = tcn[k]; s_pose[21 + k] = ktn[k]; }
            if (bid == 0 && last_it) {   // the level's result for whatever runs next (the next level's launch, the end of the run, diagnostics)
#pragma unroll
                for (int k = 0; k < 16; k++) st->resultRt[k] = RRt[k];
#pragma unroll
                for (int k = 0; k < 9; k++) { st->Rcurr[k] = Rc[k]; st->krkinv[k] = krkn[k]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { st->tcurr[k] = tcn[k]; st->kt[k] = ktn[k]; }
                if (a.final_level) track_end_vals(st, RGB, 1, Rc, tcn, Rp, tpv, a.weight_mult, a.commit, a.lctr);
                st->gn_done_seq = a.level + 1;
            }
        }
        __syncthreads();
    }
    // A meeting failed: nothing of this level was published (only block 0 publishes, and only behind the level's last meeting, when it also marks the level done).
    // Every block leaves; the one-workgroup kernel enqueued behind this launch (k_gn_level_solo) finds the level not done and runs it again from the state the level
    // started at -- slowly, exactly (the sums are order-independent), and inside the same frame.
    (void)failed;
}


// publishes the result of a tracker run that was enqueued before its frame (k_track_end with commit = 0)
__global__ void k_commit_pose(DevState* st, unsigned int* lctr)
{
    // Every load of the launch is issued ahead of its first store: `st` aliases itself, so the one-lane copy loop was ~35 chained L2 round trips (7.4 us on the
    // main stream of every frame of the tracked-ahead path).  Lanes 0..32 carry the copy, lane 0 the inputs of the view-list decision.
    (void)lctr;
    const int t = threadIdx.x;
    float cp = 0.f;
    if (t < 16) cp = st->spec_pose[t]; else if (t < 32) cp = st->spec_pose_inv[t - 16]; else if (t == 32) cp = st->spec_weighting;
    float A[16], B[16];
    int valid = 0, age = 0;
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) { A[k] = st->vl_pose[k]; B[k] = st->spec_pose[k]; }
        valid = st->vl_valid; age = st->vl_age;
    }
    if (t < 16) st->pose[t] = cp; else if (t < 32) st->pose_inv[t - 16] = cp; else if (t == 32) st->weighting = cp;
    if (t != 0) return;
    vlist_decide_core(st, A, B, valid, age);
}


// ======================================================================= host drivers

static const dim3 B2(32, 8);
static inline dim3 G2(int w, int h) { return dim3(cdiv(w, 32), cdiv(h, 8)); }

int ifx_alloc_tracker(ifx* h)
{
    Pyr& p = h->pyr;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        p.w[i] = h->w >> i; p.h[i] = h->h >> i;
        size_t n = (size_t)p.w[i] * p.h[i];
        for (int q = 0; q < 2; q++) {   // frame side, double-buffered (FrameSlot)
            FrameSlot& f = h->slot[q];
            HIPCHK(h, hipMalloc(&f.depth_tmp[i], n * 2));
            HIPCHK(h, hipMalloc(&f.vmap_curr[i], n * 12)); HIPCHK(h, hipMalloc(&f.nmap_curr[i], n * 12));
            HIPCHK(h, hipMalloc(&f.next_img[i], n)); HIPCHK(h, hipMemset(f.next_img[i], 0, n));
            HIPCHK(h, hipMalloc(&f.didx[i], n * 2)); HIPCHK(h, hipMalloc(&f.didy[i], n * 2));
        }
        HIPCHK(h, hipMalloc(&p.vmap_cam[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_cam[i], n * 12));
        HIPCHK(h, hipMalloc(&p.vmap_prev[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_prev[i], n * 12));
        HIPCHK(h, hipMalloc(&p.last_depth[i], n * 4));
        HIPCHK(h, hipMalloc(&p.last_img[i], n));
        HIPCHK(h, hipMalloc(&p.cloud[i], n * 12));
        HIPCHK(h, hipMalloc(&p.corres[i], n * 8));
    }
    const int maxb = 1024;
    HIPCHK(h, hipMalloc(&p.acc, (3 * IFX_ACC_REPL * IFX_ACC_STRIDE * sizeof(double))));
    HIPCHK(h, hipMemset(p.acc, 0, (3 * IFX_ACC_REPL * IFX_ACC_STRIDE * sizeof(double))));
    {   // grids of the persistent level kernel must be co-resident: blocks per CU from the runtime's occupancy calculator x the CU count
        int cus = 0, dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int occ[4] = {0, 0, 0, 0};
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[0], k_gn_level<1>, RED_THREADS, 0);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[1], k_gn_level<2>, RED_THREADS, 0);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[2], k_gn_level<3>, RED_THREADS, 0);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[3], k_gn_level<4>, RED_THREADS, 0);
        for (int q = 0; q < 4; q++) h->gn_max_blocks[q] = std::max(0, occ[q]) * std::max(0, cus);
        int occ_ir = 0;   // one round of residency of the two-launch form's first launch (its residual half is capped to what the ICP half leaves of it: ifx_tracker_run)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_ir, k_icp_residual<false, false, true>, RED_THREADS, 0) == hipSuccess && occ_ir > 0 && cus > 0) h->icp_resident_blocks = occ_ir * cus;
    }
    h->res_rows = std::max(maxb, cdiv(h->P, RED_THREADS) + 1);   // the residual pass runs one block per 256 pixels
    HIPCHK(h, hipMalloc(&h->res_partials, (size_t)h->res_rows * 2 * 4));
    HIPCHK(h, hipMalloc(&h->d_out29, 64 * 4));
    HIPCHK(h, hipMalloc(&h->d_ticket, 512));
    HIPCHK(h, hipMemset(h->d_ticket, 0, 512));
    p.res_partials = h->res_partials; p.ticket = h->d_ticket;
    for (int q = 0; q < 2; q++) {
        HIPCHK(h, hipMalloc(&h->slot[q].so3, sizeof(DevState)));
        HIPCHK(h, hipMemset(h->slot[q].so3, 0, sizeof(DevState)));
    }
    ifx_bind_slot(h, 0);
    return IFX_OK;
}

// makes slot s the current frame: the members the rest of the code reads (h->rgb, h->dm, pyr.vmap_curr, ...) are
// aliases of the slot's buffers; the previous image pyramid of the SO(3) step is the other slot's
void ifx_bind_slot(ifx* h, int s)
{
    FrameSlot& f = h->slot[s];
    Pyr& p = h->pyr;
    h->cur_slot = s;
    h->rgb = f.rgb; h->depth_raw = f.depth_raw; h->depth_filt = f.depth_filt; h->dm = f.dm; h->dmf = f.dmf;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        p.depth_tmp[i] = (i == 0) ? f.depth_filt : f.depth_tmp[i]; p.vmap_curr[i] = f.vmap_curr[i]; p.nmap_curr[i] = f.nmap_curr[i];
        p.next_img[i] = f.next_img[i]; p.didx[i] = f.didx[i]; p.didy[i] = f.didy[i];
        p.lastnext_img[i] = h->slot[s < 2 ? (s ^ 1) : 0].next_img[i];   // (slots 3.., a camera's run-ahead frame: the caller points it at the camera's parked pyramid)
    }
}

static void free_m2m(ifx* h);
void ifx_free_tracker(ifx* h)
{
    Pyr& p = h->pyr;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        for (size_t q = 0; q < h->slot.size(); q++) {
            FrameSlot& f = h->slot[q];
            hipFree(f.depth_tmp[i]); hipFree(f.vmap_curr[i]); hipFree(f.nmap_curr[i]); hipFree(f.next_img[i]); hipFree(f.didx[i]); hipFree(f.didy[i]);
        }
        hipFree(p.vmap_cam[i]); hipFree(p.nmap_cam[i]); hipFree(p.vmap_prev[i]); hipFree(p.nmap_prev[i]); hipFree(p.last_depth[i]); hipFree(p.last_img[i]);
        hipFree(p.cloud[i]); hipFree(p.corres[i]);
    }
    for (size_t q = 0; q < h->slot.size(); q++) hipFree(h->slot[q].so3);
    if (h->d_cam_trk) {
        Pyr& cp = h->cam_pyr;
        for (int i = 0; i < IFX_NUM_PYRS; i++) { hipFree(cp.vmap_cam[i]); hipFree(cp.nmap_cam[i]); hipFree(cp.vmap_prev[i]); hipFree(cp.nmap_prev[i]); hipFree(cp.last_depth[i]); hipFree(cp.last_img[i]); hipFree(cp.cloud[i]); hipFree(cp.corres[i]); }
        hipFree(h->d_cam_trk); hipFree(h->cam_so3_acc); hipFree(h->cam_so3_ticket);
        h->d_cam_trk = nullptr;
    }
    free_m2m(h);
    hipFree(h->d_graph); hipFree(h->d_sample); hipFree(h->d_cons); hipFree(h->d_project); hipFree(h->d_fern); hipFree(h->d_inst_gt);
    if (h->h_fern) hipHostFree(h->h_fern);
    if (h->ev_fern) hipEventDestroy(h->ev_fern);
    hipFree(h->pyr.acc); hipFree(h->res_partials); hipFree(h->d_out29); hipFree(h->d_ticket);
}

static inline int red_blocks(ifx* h, int n, int it = RED_IT)
{
    int b = cdiv(n, RED_THREADS * it);
    // cap on the blocks of a reduction launch (their partial rows are summed by the last block): 304 at 640x480 (one 1024-pixel chunk per block at
    // level 0, flat beyond), one block per 2048 pixels on larger images (1280x960: 608 blocks, tracker 1.33 -> 1.18 ms)
    const int cap = h->opt_icp_blocks > 0 ? h->opt_icp_blocks : std::max(456, std::min(1024, h->P / 2048));
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return b;
}

// RGBDOdometry::initFirstRGB, EF/Utils/RGBDOdometry.cpp:249-265
int ifx_tracker_init_first(ifx* h)
{
    Pyr& p = h->pyr;
    LAUNCH(h, "intensity", dim3(cdiv(h->P, 256)), dim3(256), k_intensity, h->rgb, 3, h->P, p.lastnext_img[0]);
    for (int i = 0; i + 1 < IFX_NUM_PYRS; i++)
        LAUNCH(h, "pyrdown_u8", G2(p.w[i + 1], p.h[i + 1]), B2, k_pyrdown_gauss_u8, p.lastnext_img[i], p.w[i], p.h[i], p.lastnext_img[i + 1]);
    return IFX_OK;
}

// model side: initICPModel + initRGBModel (EF/Utils/RGBDOdometry.cpp:169-206,237-241)
static inline GnBegin gb_none() { GnBegin g; g.st = nullptr; g.ss = nullptr; g.so3 = 0; g.fx = g.fy = g.cx = g.cy = 0.f; g.keep_last = 0; g.enabled = 0; g.persist = 0; return g; }
static void tracker_init_model(ifx* h, DevState* st, Pyr& p, float icp_weight, const float* pv, const float* pn, const uint8_t* pi, const float* fv, const float* fn, const uint8_t* fi, const GnBegin* gbp = nullptr)
{
    const GnBegin gb_off = gb_none();
    const ifx_config& c = h->cfg;
    const int rgb = icp_weight < 100;
    const int iterations[3] = {c.fast_odom ? 3 : 10, c.pyramid ? 5 : 0, c.pyramid ? 4 : 0};
    ModelOut3 o3;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        const float div = (float)(1 << i);
        ModelOut& o = o3.l[i];
        o.vcam = p.vmap_cam[i]; o.ncam = p.nmap_cam[i]; o.depth = p.last_depth[i]; o.img = p.last_img[i];
        o.vprev = p.vmap_prev[i]; o.nprev = p.nmap_prev[i]; o.cloud = (rgb && iterations[i] > 0) ? p.cloud[i] : nullptr;
        o.invFx = 1.0f / (c.fx / div); o.invFy = 1.0f / (c.fy / div); o.cx = c.cx / div; o.cy = c.cy / div;
    }
#ifdef IFX_EXPERIMENTS
    if (h->opt_model_fused && IFX_NUM_PYRS == 3 && h->w % 32 == 0 && h->h % 16 == 0 && fv) {   // the three levels in one launch
        LAUNCH(h, "model_pyr3", dim3(h->w / 32, h->h / 16), dim3(256), k_model_pyr3, st, pv, pn, pi, fv, fn, fi, h->w, h->h, 6.0f, o3);
        return;
    }
#endif
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        const ModelOut& o = o3.l[i];
        if (i == 0) LAUNCH(h, "model_l0", G2(h->w, h->h), B2, k_model_l0, st, pv, pn, pi, fv, fn, fi, h->w, h->h, 6.0f, o);
        else LAUNCH(h, "model_down", G2(p.w[i], p.h[i]), B2, k_model_down, st, p.vmap_cam[i - 1], p.nmap_cam[i - 1], p.last_depth[i - 1], p.last_img[i - 1], p.w[i - 1], p.h[i - 1], o, (i == IFX_NUM_PYRS - 1 && gbp) ? *gbp : gb_off);
    }
}

// frame side of the model-to-model tracker: initICP(predictedVertices, predictedNormals) + initRGB(predictedImage)
// (EF/Utils/RGBDOdometry.cpp:144-167, 243-247) -- copyMaps, verticesToDepth, intensity at level 0, then resizeVMap / resizeNMap and the
// two Gaussian pyr-downs per level: the model-side kernels with the camera-frame maps as their only output -- plus the Sobel
// images of :287-293.
__global__ void k_sobel(const uint8_t* __restrict__ img, int w, int h, int16_t* __restrict__ dx, int16_t* __restrict__ dy)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y * blockDim.y + threadIdx.y;
    if (u >= w || v >= h) return;
    // applyKernel (Sobel), EF/Cuda/cudafuncs.cu:583-607
    const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    float dxVal = 0, dyVal = 0;
    int k = 8;
    for (int j = max(v - 1, 0); j <= min(v + 1, h - 1); j++)
        for (int i = max(u - 1, 0); i <= min(u + 1, w - 1); i++) {
            dxVal += (float)img[j * w + i] * gsx[k];
            dyVal += (float)img[j * w + i] * gsy[k];
            --k;
        }
    dx[v * w + u] = (int16_t)dxVal;
    dy[v * w + u] = (int16_t)dyVal;
}
static void tracker_init_frame_maps(ifx* h, DevState* st, Pyr& p, const float* pv, const float* pn, const uint8_t* pi)
{
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        ModelOut o;
        o.vcam = p.vmap_curr[i]; o.ncam = p.nmap_curr[i]; o.depth = p.next_depth[i]; o.img = p.next_img[i];
        o.vprev = nullptr; o.nprev = nullptr; o.cloud = nullptr;
        o.invFx = 0; o.invFy = 0; o.cx = 0; o.cy = 0;
        if (i == 0) LAUNCH(h, "model_l0", G2(h->w, h->h), B2, k_model_l0, st, pv, pn, pi, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, h->w, h->h, 6.0f, o);
        else LAUNCH(h, "model_down", G2(p.w[i], p.h[i]), B2, k_model_down, st, p.vmap_curr[i - 1], p.nmap_curr[i - 1], p.next_depth[i - 1], p.next_img[i - 1], p.w[i - 1], p.h[i - 1], o, gb_none());
        LAUNCH(h, "sobel", G2(p.w[i], p.h[i]), B2, k_sobel, p.next_img[i], p.w[i], p.h[i], p.didx[i], p.didy[i]);
    }
}

// frame side: initICP(filteredDepth) + initRGB (EF/Utils/RGBDOdometry.cpp:118-142,243-247); nextDepth
// aliases lastDepth because initRGB re-reads the model's vmaps_tmp (reference behaviour).
static void tracker_init_frame(ifx* h, const uint16_t* depth_filt, const uint8_t* rgb)
{
    Pyr& p = h->pyr;
    (void)depth_filt;   // level 0 of the depth pyramid IS the filtered depth (ifx_bind_slot aliases it; the reference copies it)
    LAUNCH(h, "intensity", dim3(cdiv(h->P, 256)), dim3(256), k_intensity, rgb, 3, h->P, p.next_img[0]);
    for (int i = 1; i < IFX_NUM_PYRS; i++)
        LAUNCH(h, "frame_down", G2(p.w[i], p.h[i]), B2, k_frame_down, p.depth_tmp[i - 1], p.next_img[i - 1], p.w[i - 1], p.h[i - 1], p.depth_tmp[i], p.next_img[i]);
    FrameLevels fl;
    int blocks = 0;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        float div = (float)(1 << i);
        float fx = h->cfg.fx / div, fy = h->cfg.fy / div;
        FrameLevel& L = fl.l[i];
        L.depth = p.depth_tmp[i]; L.img = p.next_img[i]; L.vmap = p.vmap_curr[i]; L.nmap = p.nmap_curr[i]; L.dx = p.didx[i]; L.dy = p.didy[i];
        L.w = p.w[i]; L.h = p.h[i]; L.tiles_x = cdiv(p.w[i], 32); L.first_block = blocks;
        L.fx_inv = 1.f / fx; L.fy_inv = 1.f / fy; L.cx = h->cfg.cx / div; L.cy = h->cfg.cy / div;
        blocks += L.tiles_x * cdiv(p.h[i], 8);
    }
    fl.cutoff = h->cfg.max_depth_processed;
    LAUNCH(h, "frame_maps", dim3(blocks), dim3(256), k_frame_maps, fl);
}

int ifx_comm_ready(ifx* h);
int ifx_comm_allreduce_f64(ifx* h, double* d_ptr, int n);   // SUM over the ranks, in place, on the handle's CURRENT stream (ifx_comm.hip)
static inline const char* icp_name_rows(int level) { static const char* const n[3] = {"icp_residual_rows@L0", "icp_residual_rows@L1", "icp_residual_rows@L2"}; return n[level]; }
// getIncrementalTransformation, EF/Utils/RGBDOdometry.cpp:267-603, enqueued without any readback
// `st` / `p`: the state and pyramids of the tracker instance (frame-to-model: h->d_state / h->pyr; model-to-model: h->d_m2m / h->m2m)
static void tracker_run(ifx* h, DevState* st, Pyr& p, float icp_weight, int so3, float weight_mult, int commit, bool frame_tracker, int keep_last = 0)
{
    const ifx_config& c = h->cfg;
    const int icp = icp_weight > 0, rgb = icp_weight < 100;
    int iterations[3] = {c.fast_odom ? 3 : 10, c.pyramid ? 5 : 0, c.pyramid ? 4 : 0};
    int first = -1;
    for (int i = IFX_NUM_PYRS - 1; i >= 0; i--) if (iterations[i] > 0) { first = i; break; }
    if (frame_tracker && h->gn_begin_folded) h->gn_begin_folded = 0;   // the model side's last launch already did it (ifx_tracker_model_side)
    else {
        const float div = (float)(1 << (first < 0 ? 0 : first));
        LAUNCH(h, "track_gn_begin", dim3(1), dim3(64), k_track_gn_begin, st, h->slot[h->cur_slot].so3, so3, c.fx / div, c.fy / div, c.cx / div, c.cy / div, keep_last, h->opt_gn_persist ? 1 : 0);
    }
    static const float minGrad[3] = {5, 3, 1};
    const double sobelScale = 1.0 / 8.0;
    bool ended = false;
    // which levels run in the persistent kernel, and whether the two-launch iterations form the run's tail (then their solves move into the next launch's prologue: gn_prologue)
    int persist_q[IFX_NUM_PYRS], persist_nb[IFX_NUM_PYRS];
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        persist_q[i] = -1; persist_nb[i] = 0;
        bool lds_lvl = false;
#ifdef IFX_EXPERIMENTS
        lds_lvl = h->opt_icp_lds && i == 0 && frame_tracker;
#endif
        if (frame_tracker && (h->opt_gn_persist & (1 << i)) && iterations[i] > 0 && !lds_lvl && icp && rgb) {
            static const int pxs[4] = {1, 2, 3, 4};
            const int n = p.w[i] * p.h[i];
            for (int t = 0; t < 4 && persist_q[i] < 0; t++) {   // (fewer, fatter blocks at the finer levels -- 4 pixels per thread -- were tried for cheaper meetings: slower)
                const int need = cdiv(n, RED_THREADS * pxs[t]);
                if (need <= h->gn_max_blocks[t] * 7 / 8 && need <= h->opt_gn_persist_blocks) { persist_q[i] = t; persist_nb[i] = need; }   // (co-residency is what the barriers need; an eighth of the slots stays free for whatever shares the GPU)
            }
        }
    }
    if (frame_tracker && h->own && h->opt_own_track_rows && ifx_comm_ready(h) && h->own_track_rank < 0) {
        // the reductions over this rank's blocks + all-reduce + one-workgroup solve (k_icp_residual_rows above).  opt_own_track_rows_emulate = G > 1 (test switch): this ONE rank
        // plays G in turn into the same accumulator rows -- the partition's cover, without a second GPU
        const int Ge = h->opt_own_track_rows_emulate > 1 ? h->opt_own_track_rows_emulate : 0;
        const int sn = Ge ? Ge : h->own_g, r_lo = Ge ? 0 : h->cfg.rank, r_hi = Ge ? Ge : h->cfg.rank + 1;
        double* const gacc = (double*)((char*)st + offsetof(DevState, gn_acc));
        int* const gres = (int*)((char*)st + offsetof(DevState, gn_res));
        const int row = IFX_ACC_REPL * IFX_ACC_STRIDE;
        for (int i = IFX_NUM_PYRS - 1; i >= 0; i--) {
            if (i == 0) ifx_enqueue_hinted_frame_side(h);
            const float div = (float)(1 << i);
            const float fx = c.fx / div, fy = c.fy / div, cx = c.cx / div, cy = c.cy / div;
            const int lw = p.w[i], lh = p.h[i], n = lw * lh;
            const int nbv = cdiv(red_blocks(h, n), sn), nbr = std::min(cdiv(n, RED_THREADS * RED_IT), h->res_rows);
            const int nb_rgb_v = cdiv(std::min(red_blocks(h, n, RED_IT_RGB), h->opt_rgb_blocks > 0 ? h->opt_rgb_blocks : 192), sn);
            int nl = i - 1;
            while (nl >= 0 && iterations[nl] == 0) nl--;
            if (nl < 0) nl = 0;
            const float ld = (float)(1 << nl);
            PairArgs pa;
            pa.vmap_curr = p.vmap_curr[i]; pa.nmap_curr = p.nmap_curr[i]; pa.vmap_prev = p.vmap_prev[i]; pa.nmap_prev = p.nmap_prev[i];
            pa.fx = fx; pa.fy = fy; pa.cx = cx; pa.cy = cy; pa.distThres = 0.10f; pa.angleThres = sinf(20.f * 3.14159254f / 180.f);
            pa.minScale = (float)(pow(minGrad[i], 2.0) / pow(sobelScale, 2.0)); pa.maxDepthDelta = 0.07f;
            pa.dIdx = p.didx[i]; pa.dIdy = p.didy[i]; pa.lastDepth = p.last_depth[i]; pa.nextDepth = p.next_depth[i] ? p.next_depth[i] : p.last_depth[i]; pa.lastImage = p.last_img[i]; pa.nextImage = p.next_img[i];
            pa.lds_tiles = 0; pa.corres = (Corres8*)p.corres[i]; pa.w = lw; pa.h = lh; pa.nb_icp = icp ? nbv : 0; pa.nb_res = rgb ? nbr : 0; pa.check_skip = 0;
            for (int j = 0; j < iterations[i]; j++) {
                const float nd = (j == iterations[i] - 1) ? ld : div;
                for (int r = r_lo; r < r_hi; r++)   // (the residual pass once: with the first of the launches)
                    LAUNCH(h, icp_name_rows(i), dim3(pa.nb_icp + (r == r_lo ? pa.nb_res : 0)), dim3(RED_THREADS), k_icp_residual_rows, (const DevState*)st, pa.nb_icp, lw, lh, gacc, gres, pa, r, sn);
                if (icp && h->track_rc == IFX_OK) h->track_rc = ifx_comm_allreduce_f64(h, gacc, row);
                StepArgs sa2;
                sa2.corres = (const Corres8*)p.corres[i]; sa2.cloud = p.cloud[i]; sa2.fx = fx; sa2.fy = fy; sa2.sobelScale = (float)sobelScale;
                sa2.dIdx = p.didx[i]; sa2.dIdy = p.didy[i]; sa2.w = lw; sa2.h = lh; sa2.nb = nb_rgb_v; sa2.nb_icp = pa.nb_icp; sa2.nb_res = pa.nb_res;
                sa2.icp = icp; sa2.rgb = rgb; sa2.icp_weight = icp_weight; sa2.nfx = c.fx / nd; sa2.nfy = c.fy / nd; sa2.ncx = c.cx / nd; sa2.ncy = c.cy / nd; sa2.ki = kinv_of(sa2.nfx, sa2.nfy, sa2.ncx, sa2.ncy);
                sa2.check_skip = 0;
                bool later = false;
                for (int q = i - 1; q >= 0; q--) later = later || iterations[q] > 0;
                sa2.final_iter = (j == iterations[i] - 1 && !later) ? 1 : 0;
                sa2.end_run = sa2.final_iter; sa2.commit = commit; sa2.weight_mult = weight_mult; sa2.lctr = h->d_list_ctr; sa2.pro = 0; sa2.pro_k = 0;
                ended = ended || sa2.end_run;
                if (rgb) {
                    for (int r = r_lo; r < r_hi; r++) LAUNCH(h, "rgb_step_rows", dim3(nb_rgb_v), dim3(RED_THREADS), k_rgb_step_rows, st, nb_rgb_v, lw, lh, sa2, r, sn);
                    if (h->track_rc == IFX_OK) h->track_rc = ifx_comm_allreduce_f64(h, gacc + row, row);   // (a failed collective: the run is enqueued to its end, the frame call reports it)
                }
                LAUNCH(h, "gn_solve_rows", dim3(1), dim3(RED_THREADS), k_gn_solve_rows, st, sa2);
            }
        }
        if (!ended) LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, st, rgb, 1, weight_mult, commit, h->d_list_ctr);
        return;
    }
    bool pro = h->opt_gn_prologue != 0;
    // The prologue solve is repeated by every block of a launch: free while the grid is one wave of blocks, 4-6 us per launch at 4800 + 304 blocks (level 0 of a
    // 1280x960 frame: 516 against 540 frames/s, profiles/r04_z2_*).  n_pro: the leading iterations of the tail (coarse levels first) whose launches stay under
    // opt_gn_prologue_blocks; the last of them is solved in the last-block form (StepArgs::pro = 2) and leaves the pose in the state, where round 3's form of the
    // finer levels reads it.
    int n_pro = 0;
    {
        bool seen_two_launch = false, small = true;
        for (int i = IFX_NUM_PYRS - 1; i >= 0; i--) {
            if (iterations[i] <= 0) continue;
            if (persist_q[i] >= 0) { if (seen_two_launch) pro = false; }   // a persistent level BEHIND two-launch iterations reads the pose from the state: round 3's form keeps it there
            else {
                seen_two_launch = true;
                const int n = p.w[i] * p.h[i];
                const int blocks = (icp ? red_blocks(h, n) : 0) + (rgb ? std::min(std::min(cdiv(n, RED_THREADS * RED_IT), h->res_rows), h->opt_res_blocks > 0 ? h->opt_res_blocks : (1 << 30)) : 0);
                small = small && blocks <= h->opt_gn_prologue_blocks;
                if (small) n_pro += iterations[i];
            }
        }
#ifdef IFX_EXPERIMENTS
        if (h->opt_icp_lds || h->opt_icp_px) pro = false;
#endif
        if (n_pro < 2) pro = false;   // (a chain of one has nothing to hand over)
    }
    int persist_iters = 0;
    int tail_k = 0;   // position of the next two-launch iteration in the tail
    char* const stb = (char*)st;
    auto gnp_acc_of = [&](int par) { return (double*)(stb + offsetof(DevState, gnp_acc)) + (size_t)(par * 2) * IFX_ACC_REPL * IFX_ACC_STRIDE; };
    auto gnp_res_of = [&](int par) { return (int*)(stb + offsetof(DevState, gnp_res)) + par * 16; };
    auto gnp_rrt_of = [&](int par) { return (double*)(stb + offsetof(DevState, gnp_RRt)) + par * 16; };
    // per-kernel timing (option kernel_timing) is kept per pyramid level: ifx_kernel_ms("icp_residual") sums the levels, "icp_residual@L0" is level 0 alone (bench.py's per-level roofline)
    static const char* const icp_name[3] = {"icp_residual@L0", "icp_residual@L1", "icp_residual@L2"};
    static const char* const rgb_name[3] = {"rgb_step_solve@L0", "rgb_step_solve@L1", "rgb_step_solve@L2"};
    static_assert(IFX_NUM_PYRS == 3, "level names");
    for (int i = IFX_NUM_PYRS - 1; i >= 0; i--) {
        // The coarse levels are on the queue: the host is now ahead of the GPU by ~20 latency-bound launches, and the
        // finest level keeps the GPU mostly idle for another ~0.3 ms -- the place to slip in the next frame's image-only work.
        if (i == 0 && frame_tracker) ifx_enqueue_hinted_frame_side(h);
        float div = (float)(1 << i);
        float fx = c.fx / div, fy = c.fy / div, cx = c.cx / div, cy = c.cy / div;
        int lw = p.w[i], lh = p.h[i], n = lw * lh, nb = red_blocks(h, n), nb_rgb = std::min(red_blocks(h, n, RED_IT_RGB), h->opt_rgb_blocks > 0 ? h->opt_rgb_blocks : 192);   // the photometric step's launch is dominated by the last block's hand-off and solve: fewer blocks, fewer partial rows (64: 16.9, 192: 15.3, 304: 16.1, 608: 18.5 us per launch)
        // intrinsics of the level the iteration after this level's last one runs at (for the warp matrices the solve emits)
        int nl = i - 1;
        while (nl >= 0 && iterations[nl] == 0) nl--;
        if (nl < 0) nl = 0;
        const float ld = (float)(1 << nl);
        PairArgs pa;
        pa.vmap_curr = p.vmap_curr[i]; pa.nmap_curr = p.nmap_curr[i]; pa.vmap_prev = p.vmap_prev[i]; pa.nmap_prev = p.nmap_prev[i];
        pa.fx = fx; pa.fy = fy; pa.cx = cx; pa.cy = cy; pa.distThres = 0.10f; pa.angleThres = sinf(20.f * 3.14159254f / 180.f);
        pa.minScale = (float)(pow(minGrad[i], 2.0) / pow(sobelScale, 2.0)); pa.maxDepthDelta = 0.07f;
        pa.dIdx = p.didx[i]; pa.dIdy = p.didy[i]; pa.lastDepth = p.last_depth[i]; pa.nextDepth = p.next_depth[i] ? p.next_depth[i] : p.last_depth[i]; pa.lastImage = p.last_img[i]; pa.nextImage = p.next_img[i];
        // The residual half of the launch is the slower one and scales with its blocks (its totals go through integer atomics, it has no partial rows for the
        // last block to sum): one pixel per thread, no loop -- 152 blocks 17.5 us, 304 blocks 13.3 us, 1200 blocks 11.3 us per launch at 640x480 (1053 -> 1102 frames/s).
#ifdef IFX_EXPERIMENTS
        const bool lds_tiles = h->opt_icp_lds && i == 0 && frame_tracker;
#else
        const bool lds_tiles = false;
#endif
        // One round of residency: the launch's blocks (ICP half + residual half) must all be on the GPU at once.  At 640x480 level 0 they were 456 + 1024 = 1480 against the 1024 the GPU holds
        // (118 VGPRs: 4 blocks per CU), so a third of them started when the first ones left -- and paid the launch's start-up chain (kernel arguments, the previous iteration's totals and
        // 6x6 solve in every block's prologue, the first loads) a second time: 17.2 -> 14.5 us per level-0 launch with the residual half capped to what is left of one round
        // (profiles/r05_ai_ab_tracker_blocks.txt: 1514 -> 1561 frames/s).  The halves loop over their pixels anyway (grid-stride); opt_res_blocks > 0 overrides.
        const int nbi = lds_tiles ? cdiv(lw, LT_W) * cdiv(lh, LT_H) : nb;
        const int res_round = std::max(128, h->icp_resident_blocks - (icp ? nbi : 0));
        const int nbr = std::min(std::min(cdiv(n, RED_THREADS * RED_IT), h->res_rows), h->opt_res_blocks > 0 ? h->opt_res_blocks : res_round);
        pa.lds_tiles = lds_tiles ? 1 : 0;
        pa.corres = (Corres8*)p.corres[i]; pa.w = lw; pa.h = lh; pa.nb_icp = icp ? nbi : 0; pa.nb_res = rgb ? nbr : 0;
        double* const gacc = (double*)((char*)st + offsetof(DevState, gn_acc));
        int* const gres = (int*)((char*)st + offsetof(DevState, gn_res));
        pa.check_skip = frame_tracker ? 0 : 1;   // (accumulator rows, residual totals, ticket: DevState::gn_acc / gn_res / gn_ticket of `st`)
        if (persist_q[i] >= 0) {   // all iterations of the level in one persistent launch
            const int q = persist_q[i], nbp = persist_nb[i];
            {
                LevelArgs la;
                la.vmap_curr = pa.vmap_curr; la.nmap_curr = pa.nmap_curr; la.vmap_prev = pa.vmap_prev; la.nmap_prev = pa.nmap_prev;
                la.dIdx = pa.dIdx; la.dIdy = pa.dIdy; la.lastDepth = pa.lastDepth; la.nextDepth = pa.nextDepth; la.lastImage = pa.lastImage; la.nextImage = pa.nextImage;
                la.cloud = p.cloud[i]; la.corres = (Corres8*)p.corres[i];
                la.fx = fx; la.fy = fy; la.cx = cx; la.cy = cy; la.distThres = pa.distThres; la.angleThres = pa.angleThres; la.minScale = pa.minScale; la.maxDepthDelta = pa.maxDepthDelta;
                la.sobelScale = (float)sobelScale;
                la.w = lw; la.h = lh; la.iters = iterations[i]; la.icp = icp; la.rgb = rgb; la.icp_weight = icp_weight;
                la.nfx = c.fx / ld; la.nfy = c.fy / ld; la.ncx = c.cx / ld; la.ncy = c.cy / ld;
                la.ki_same = kinv_of(fx, fy, cx, cy); la.ki_next = kinv_of(la.nfx, la.nfy, la.ncx, la.ncy);
                la.acc_base = persist_iters; la.level = i;   // (parity of the persistent kernel's double buffers: continuous over ITS launches, whatever two-launch levels lie between them)
                persist_iters += iterations[i];
                bool later = false;
                for (int q2 = i - 1; q2 >= 0; q2--) later = later || iterations[q2] > 0;
                la.final_level = later ? 0 : 1; la.commit = commit; la.weight_mult = weight_mult; la.lctr = frame_tracker ? h->d_list_ctr : (unsigned int*)nullptr;
                if (q == 0) LAUNCH(h, "gn_level", dim3(nbp), dim3(RED_THREADS), k_gn_level<1>, st, la);
                else if (q == 1) LAUNCH(h, "gn_level", dim3(nbp), dim3(RED_THREADS), k_gn_level<2>, st, la);
                else if (q == 2) LAUNCH(h, "gn_level", dim3(nbp), dim3(RED_THREADS), k_gn_level<3>, st, la);
                else LAUNCH(h, "gn_level", dim3(nbp), dim3(RED_THREADS), k_gn_level<4>, st, la);
                LAUNCH(h, "gn_level_solo", dim3(1), dim3(RED_THREADS), k_gn_level_solo, st, la);   // the safety net: a no-op unless a meeting of the launch above did not happen
                ended = ended || la.final_level;
                continue;
            }
        }
        // both reductions on the same pixels of one thread (k_icp_residual_px); option bits: 1 = at level 0, 2 = at levels 1 and 2, 4 = one pixel per thread at level 0 too
#ifdef IFX_EXPERIMENTS
        const bool px_form = frame_tracker && icp && rgb && !lds_tiles && (i == 0 ? (h->opt_icp_px & 1) : (h->opt_icp_px & 2));
        const bool px_two = !(h->opt_icp_px & 4);
#endif
        for (int j = 0; j < iterations[i]; j++) {
            const float nd = (j == iterations[i] - 1) ? ld : div;
#ifdef IFX_EXPERIMENTS
            if (px_form && px_two && n > 150000) LAUNCH(h, icp_name[i], dim3(cdiv(n, RED_THREADS * 2)), dim3(RED_THREADS), (k_icp_residual_px<2, false>), st, cdiv(n, RED_THREADS * 2), pa.w, pa.h, gacc, gres, pa);
            else if (px_form) LAUNCH(h, icp_name[i], dim3(cdiv(n, RED_THREADS)), dim3(RED_THREADS), (k_icp_residual_px<1, false>), st, cdiv(n, RED_THREADS), pa.w, pa.h, gacc, gres, pa);
            else if (pa.lds_tiles) { GnPro g0; memset(&g0, 0, sizeof(g0)); LAUNCH(h, icp_name[i], dim3(pa.nb_icp + pa.nb_res), dim3(RED_THREADS), (k_icp_residual<true, false, false>), st, pa.nb_icp, pa.w, pa.h, gacc, gres, pa, g0, (double*)nullptr); }   // (its 60 KB of LDS would cost the plain kernel its occupancy: a kernel of its own)
            else
#endif
            {
                // gn_prologue: iteration tail_k sums into parity tail_k & 1; from the tail's second iteration on, every block first solves the iteration before
                GnPro gp;
                gp.k = tail_k; gp.icp = icp; gp.rgb = rgb; gp.icp_weight = icp_weight; gp.nfx = fx; gp.nfy = fy; gp.ncx = cx; gp.ncy = cy; gp.ki = kinv_of(fx, fy, cx, cy);
                const bool it_pro = pro && tail_k < n_pro;
                double* const ga = it_pro ? gnp_acc_of(tail_k & 1) : gacc;
                int* const gr = it_pro ? gnp_res_of(tail_k & 1) : gres;
                double* const rrt_store = gnp_rrt_of((tail_k + 1) & 1);   // the increment after iteration tail_k - 1
                const dim3 grid(pa.nb_icp + pa.nb_res);
                if (it_pro && tail_k > 0) {
                    if (frame_tracker) LAUNCH(h, icp_name[i], grid, dim3(RED_THREADS), (k_icp_residual<false, false, true>), st, pa.nb_icp, pa.w, pa.h, ga, gr, pa, gp, rrt_store);
                    else LAUNCH(h, icp_name[i], grid, dim3(RED_THREADS), (k_icp_residual<false, true, true>), st, pa.nb_icp, pa.w, pa.h, ga, gr, pa, gp, rrt_store);
                } else {
                    if (frame_tracker) LAUNCH(h, icp_name[i], grid, dim3(RED_THREADS), (k_icp_residual<false, false, false>), st, pa.nb_icp, pa.w, pa.h, ga, gr, pa, gp, rrt_store);
                    else LAUNCH(h, icp_name[i], grid, dim3(RED_THREADS), (k_icp_residual<false, true, false>), st, pa.nb_icp, pa.w, pa.h, ga, gr, pa, gp, rrt_store);
                }
            }
            StepArgs sa2;
            sa2.corres = (const Corres8*)p.corres[i]; sa2.cloud = p.cloud[i]; sa2.fx = fx; sa2.fy = fy; sa2.sobelScale = (float)sobelScale;
            sa2.dIdx = p.didx[i]; sa2.dIdy = p.didy[i]; sa2.w = lw; sa2.h = lh; sa2.nb = nb_rgb; sa2.nb_icp = nbi; sa2.nb_res = nbr;
            sa2.icp = icp; sa2.rgb = rgb; sa2.icp_weight = icp_weight; sa2.nfx = c.fx / nd; sa2.nfy = c.fy / nd; sa2.ncx = c.cx / nd; sa2.ncy = c.cy / nd; sa2.ki = kinv_of(sa2.nfx, sa2.nfy, sa2.ncx, sa2.ncy);
            sa2.check_skip = frame_tracker ? 0 : 1;
            {   // is this the run's last iteration?  (no level below this one iterates)
                bool later = false;
                for (int q = i - 1; q >= 0; q--) later = later || iterations[q] > 0;
                sa2.final_iter = (j == iterations[i] - 1 && !later) ? 1 : 0;
            }
            sa2.end_run = sa2.final_iter; sa2.commit = commit; sa2.weight_mult = weight_mult; sa2.lctr = frame_tracker ? h->d_list_ctr : (unsigned int*)nullptr;
            sa2.pro = (pro && tail_k < n_pro) ? (tail_k == n_pro - 1 ? 2 : 1) : 0; sa2.pro_k = tail_k;   // (the chain's last iteration: last-block form; it is the run's last too unless finer levels were too large for the prologue)
            tail_k++;
            ended = ended || sa2.end_run;
            if (frame_tracker) LAUNCH(h, rgb_name[i], dim3(nb_rgb), dim3(RED_THREADS), k_rgb_step_solve<false>, st, sa2.nb, sa2.rgb, sa2.w, sa2.h, sa2);
            else LAUNCH(h, rgb_name[i], dim3(nb_rgb), dim3(RED_THREADS), k_rgb_step_solve<true>, st, sa2.nb, sa2.rgb, sa2.w, sa2.h, sa2);
        }
    }
    if (!ended)   // (no iteration ran at all: every level has zero iterations)
        LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, st, rgb, 1, weight_mult, commit, frame_tracker ? h->d_list_ctr : (unsigned int*)nullptr);
}

// frame side of the tracker for the bound slot: frame pyramids, then (unless this is the first frame, which only
// needs its intensity pyramid as the next frame's "previous image") the SO(3) pre-alignment against the other
// slot's intensity pyramid, EF/Utils/RGBDOdometry.cpp:313-386: up to 10 reduction + update launches, each exiting at
// once after convergence (device flag)
int ifx_tracker_frame_side(ifx* h, int first, double* so3_acc, unsigned int* so3_ticket)
{
    Pyr& p = h->pyr;
    const ifx_config& c = h->cfg;
    if (first) {
        LAUNCH(h, "intensity", dim3(cdiv(h->P, 256)), dim3(256), k_intensity, h->rgb, 3, h->P, p.next_img[0]);
        for (int i = 0; i + 1 < IFX_NUM_PYRS; i++)
            LAUNCH(h, "pyrdown_u8", G2(p.w[i + 1], p.h[i + 1]), B2, k_pyrdown_gauss_u8, p.next_img[i], p.w[i], p.h[i], p.next_img[i + 1]);
        return IFX_OK;
    }
    tracker_init_frame(h, h->depth_filt, h->rgb);
    if (c.so3) {
        const float d2 = 4.f;
        const int L = 2, n = p.w[L] * p.h[L], nb = cdiv(n, RED_THREADS);
        DevState* ss = h->slot[h->cur_slot].so3;
        LAUNCH(h, "so3_begin", dim3(1), dim3(64), k_so3_begin, ss, c.fx / d2, c.fy / d2, c.cx / d2, c.cy / d2);
        for (int it = 0; it < 10; it++)
            LAUNCH(h, "so3_fused", dim3(nb), dim3(RED_THREADS), k_so3_fused, ss, p.lastnext_img[L], p.next_img[L], p.w[L], p.h[L], so3_acc ? so3_acc : p.acc + 2 * IFX_ACC_REPL * IFX_ACC_STRIDE, nb,
                   so3_ticket ? so3_ticket : h->d_ticket + 4, c.fx / d2, c.fy / d2, c.cx / d2, c.cy / d2);
    }
    return IFX_OK;
}

int ifx_tracker_model_side(ifx* h, int fold_begin)
{
    // fold_begin: the caller runs the frame tracker right behind this (nothing in between touches the pose, and the frame side of the slot is
    // ready): the start of the run rides on the model side's last launch
    const ifx_config& c = h->cfg;
    const bool fold = fold_begin && !h->opt_model_fused && c.pyramid && IFX_NUM_PYRS == 3;
    GnBegin gb = gb_none();
    if (fold) {
        const float div = (float)(1 << (IFX_NUM_PYRS - 1));   // the run starts at the coarsest level (pyramid on: every level iterates)
        gb.st = h->d_state; gb.ss = h->slot[h->cur_slot].so3; gb.so3 = c.so3; gb.fx = c.fx / div; gb.fy = c.fy / div; gb.cx = c.cx / div; gb.cy = c.cy / div;
        gb.keep_last = 0; gb.enabled = 1; gb.persist = h->opt_gn_persist ? 1 : 0;
    }
    tracker_init_model(h, h->d_state, h->pyr, h->cfg.icp_weight, h->pred_vertex, h->pred_normal, h->pred_image, h->fill_vertex, h->fill_normal, h->fill_image, fold ? &gb : nullptr);
    h->gn_begin_folded = fold ? 1 : 0;
    return IFX_OK;
}

// The tracker of a PARKED camera's next frame on a tracker instance of its own (K streams over a sharded map, camera k tracked by rank k: ifx_owner_track_ahead).  The
// caller has bound frame slot 2 (its frame side is computed by ifx_tracker_frame_side with this instance's SO(3) accumulators) and set h->cur to the stream the run goes to.
// Inputs: the camera's parked prediction / fill-in / pose block; output: the instance's pose block, committed into the live state when the frame's turn comes.
static int cam_trk_alloc(ifx* h)
{
    if (h->d_cam_trk) return IFX_OK;
    Pyr& p = h->cam_pyr;
    HIPCHK(h, hipMalloc(&h->d_cam_trk, sizeof(DevState)));
    HIPCHK(h, hipMemset(h->d_cam_trk, 0, sizeof(DevState)));
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        p.w[i] = h->w >> i; p.h[i] = h->h >> i;
        const size_t n = (size_t)p.w[i] * p.h[i];
        HIPCHK(h, hipMalloc(&p.vmap_cam[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_cam[i], n * 12));
        HIPCHK(h, hipMalloc(&p.vmap_prev[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_prev[i], n * 12));
        HIPCHK(h, hipMalloc(&p.last_depth[i], n * 4));
        HIPCHK(h, hipMalloc(&p.last_img[i], n));
        HIPCHK(h, hipMalloc(&p.cloud[i], n * 12));
        HIPCHK(h, hipMalloc(&p.corres[i], n * 8));
    }
    HIPCHK(h, hipMalloc(&h->cam_so3_acc, IFX_ACC_REPL * IFX_ACC_STRIDE * sizeof(double)));
    HIPCHK(h, hipMemset(h->cam_so3_acc, 0, IFX_ACC_REPL * IFX_ACC_STRIDE * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->cam_so3_ticket, 64));
    HIPCHK(h, hipMemset(h->cam_so3_ticket, 0, 64));
    return IFX_OK;
}
// the frame slot of camera `cam`'s run ahead (index 3 + cam): its own, because the frame that takes the run also takes the slot -- raw images, filtered depth, frame
// pyramids -- instead of computing its frame side a second time, and other cameras' runs come in between
static int cam_slot_alloc(ifx* h, int cam)
{
    const size_t idx = 3 + (size_t)cam;
    if (h->slot.size() <= idx) h->slot.resize(idx + 1);
    FrameSlot& f = h->slot[idx];
    if (f.rgb) return IFX_OK;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        const size_t n = (size_t)(h->w >> i) * (h->h >> i);
        HIPCHK(h, hipMalloc(&f.depth_tmp[i], n * 2));
        HIPCHK(h, hipMalloc(&f.vmap_curr[i], n * 12)); HIPCHK(h, hipMalloc(&f.nmap_curr[i], n * 12));
        HIPCHK(h, hipMalloc(&f.next_img[i], n)); HIPCHK(h, hipMemset(f.next_img[i], 0, n));
        HIPCHK(h, hipMalloc(&f.didx[i], n * 2)); HIPCHK(h, hipMalloc(&f.didy[i], n * 2));
    }
    const size_t P = (size_t)h->P;
    HIPCHK(h, hipMalloc(&f.rgb, P * 3)); HIPCHK(h, hipMalloc(&f.depth_raw, P * 2)); HIPCHK(h, hipMalloc(&f.depth_filt, P * 2)); HIPCHK(h, hipMalloc(&f.dm, P * 4)); HIPCHK(h, hipMalloc(&f.dmf, P * 4));
    HIPCHK(h, hipMalloc(&f.so3, sizeof(DevState)));
    HIPCHK(h, hipMemset(f.so3, 0, sizeof(DevState)));
    return IFX_OK;
}
int ifx_tracker_camera_ahead(ifx* h, int cam, const uint8_t* d_rgb, const uint16_t* d_depth)
{
    int r = cam_trk_alloc(h);
    if (r) return r;
    if ((r = cam_slot_alloc(h, cam))) return r;
    CamCtx& cc = h->cams[(size_t)cam];
    DevState* st = h->d_cam_trk;
    Pyr& p = h->cam_pyr;
    const int bound = h->cur_slot, cs = 3 + cam;
    // the run's frame side -- image-only work -- on the side stream when the caller prepared one (h->cam_side_stream: it already waits for the parking of the
    // camera's context): it then overlaps the tracker of the run before it on the third stream, which waits for "frame side done" below
    hipStream_t const trk_stream = h->cur;
    if (h->cam_side_stream) h->cur = h->cam_side_stream;
    HIPCHK(h, hipMemcpyAsync(h->slot[cs].rgb, d_rgb, (size_t)h->P * 3, hipMemcpyDeviceToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->slot[cs].depth_raw, d_depth, (size_t)h->P * 2, hipMemcpyDeviceToDevice, h->cur));
    ifx_bind_slot(h, cs);
    for (int i = 0; i < IFX_NUM_PYRS; i++) h->pyr.lastnext_img[i] = cc.img[i];   // the "previous image" of the SO(3) step: the camera's last frame, parked with its context
    ifx_preprocess(h);
    ifx_tracker_frame_side(h, 0, h->cam_so3_acc, h->cam_so3_ticket);
    if (h->cam_side_stream) {
        if (!h->ev_cam_side) HIPCHK(h, hipEventCreateWithFlags(&h->ev_cam_side, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_cam_side, h->cam_side_stream));
        HIPCHK(h, hipStreamWaitEvent(trk_stream, h->ev_cam_side, 0));
        h->cur = trk_stream;
    }
    for (int i = 0; i < IFX_NUM_PYRS; i++) {   // the instance tracks the frame in the camera's slot
        p.vmap_curr[i] = h->pyr.vmap_curr[i]; p.nmap_curr[i] = h->pyr.nmap_curr[i]; p.next_img[i] = h->pyr.next_img[i]; p.didx[i] = h->pyr.didx[i]; p.didy[i] = h->pyr.didy[i];
        p.depth_tmp[i] = h->pyr.depth_tmp[i]; p.lastnext_img[i] = cc.img[i];
    }
    HIPCHK(h, hipMemcpyAsync((void*)st, cc.state, IFX_CAM_STATE_BYTES, hipMemcpyDeviceToDevice, h->cur));   // the camera's pose block: the run starts from ITS pose
    // the camera's parked prediction block has the layout of the live one
    const uint8_t* pb = cc.pred;
    const float* pv = (const float*)pb;
    const float* pn = (const float*)(pb + ((const uint8_t*)h->pred_normal - (const uint8_t*)h->pred_vertex));
    const uint8_t* pi = pb + ((const uint8_t*)h->pred_image - (const uint8_t*)h->pred_vertex);
    tracker_init_model(h, st, p, h->cfg.icp_weight, pv, pn, pi, cc.fill_v, cc.fill_n, cc.fill_i);
    tracker_run(h, st, p, h->cfg.icp_weight, h->cfg.so3, 1.0f, 1, false);   // (commit = 1 into the INSTANCE's pose block; no view-list decision: that belongs to the frame's commit)
    ifx_bind_slot(h, bound);
    return IFX_OK;
}

int ifx_tracker_run_frame(ifx* h, int commit, int keep_last)
{
    h->track_rc = IFX_OK;
    tracker_run(h, h->d_state, h->pyr, h->cfg.icp_weight, h->cfg.so3, 1.0f, commit, true, keep_last);
    return h->track_rc;   // (option own_track_rows: a collective of the run that could not be enqueued; IFX_OK otherwise)
}

// bootstrap (EF/ElasticFusion.cpp:352-356): currPose = currPose * inPose as the tracker's initial guess; lastPose keeps the pose before it
__global__ void k_bootstrap_pose(DevState* st, const float* __restrict__ in16)
{
    if (threadIdx.x != 0) return;
    float o[16];
    for (int k = 0; k < 16; k++) st->last_pose[k] = st->pose[k];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += st->pose[r * 4 + k] * in16[k * 4 + c];
            o[r * 4 + c] = s;
        }
    for (int k = 0; k < 16; k++) st->pose[k] = o[k];
}
int ifx_tracker_bootstrap_pose(ifx* h, const float* d_in_pose16)
{
    LAUNCH(h, "bootstrap_pose", dim3(1), dim3(64), k_bootstrap_pose, h->d_state, d_in_pose16);
    return IFX_OK;
}

// ---- local loop-closure detection, tracker side (EF/ElasticFusion.cpp:528-566)
// start of the detection of a frame: the model-to-model state takes the pose just tracked; its pixel counter is cleared
__global__ void k_m2m_begin(const DevState* __restrict__ st, DevState* __restrict__ m)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) { m->pose[k] = st->pose[k]; m->pose_inv[k] = st->pose_inv[k]; }
    m->dense_enough = 1; m->count = 0; m->skip = 0;
}
// after the INACTIVE render: its covered pixels are counted (one atomic per block); nothing old in view -> every reduction would be empty
// (count 0 fails the gate of :566 whatever else is computed) and the tracker kernels return at once
__global__ __launch_bounds__(256) void k_m2m_count(DevState* __restrict__ m, const float4* __restrict__ old_vertex, int P)
{
    __shared__ int lds[4];
    int c = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P; k += blockDim.x * gridDim.x) c += (old_vertex[k].z != 0);
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { const int t = lds[0] + lds[1] + lds[2] + lds[3]; if (t) atomicAdd(&m->count, t); }
}
__global__ void k_m2m_arm(DevState* __restrict__ m)
{
    if (threadIdx.x == 0) m->skip = (m->count == 0);
}
// getCovariance (EF/Utils/RGBDOdometry.cpp:605-608: lastA.lu().inverse(), here Gauss-Jordan with partial pivoting in f64) and the gates of
// EF/ElasticFusion.cpp:547-566.  The verdict goes to the main state; the deformation an accepted candidate triggers in the reference is
// not part of this path: candidates are counted and reported.
__global__ void k_m2m_decide(DevState* __restrict__ st, const DevState* __restrict__ m, int count_thresh, float err_thresh, float cov_thresh, float* __restrict__ host_lc)
{
    if (threadIdx.x != 0) return;
    float* lc = st->lc;
    for (int k = 0; k < 23; k++) lc[k] = 0.f;
    lc[1] = (float)m->count;
    for (int k = 0; k < 16; k++) lc[6 + k] = st->pose[k];
    if (!m->skip) {
        // every index below is a compile-time constant after unrolling (the pivot row is swapped in by predicated exchanges), so the
        // 6 x 12 tableau lives in registers: 50 us -> a few us for this one-thread kernel on the critical path of the frame result
        double a[6][12];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) { a[r][c] = m->lastA[r * 6 + c]; a[r][6 + c] = (r == c) ? 1.0 : 0.0; }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int piv = k;
            double best = fabs(a[k][k]);
#pragma unroll
            for (int r = k + 1; r < 6; r++) { const double v = fabs(a[r][k]); if (v > best) { best = v; piv = r; } }
#pragma unroll
            for (int r = k + 1; r < 6; r++)
                if (piv == r) {
#pragma unroll
                    for (int c = 0; c < 12; c++) { const double tmp = a[k][c]; a[k][c] = a[r][c]; a[r][c] = tmp; }
                }
            const double d = 1.0 / a[k][k];
#pragma unroll
            for (int c = 0; c < 12; c++) a[k][c] *= d;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                if (r == k) continue;
                const double f = a[r][k];
#pragma unroll
                for (int c = 0; c < 12; c++) a[r][c] -= f * a[k][c];
            }
        }
        int cov_ok = 1;
        double cmax = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (a[i][6 + i] > (double)cov_thresh) cov_ok = 0;
            if (!(a[i][6 + i] <= cmax)) cmax = a[i][6 + i];
        }
        const int accept = cov_ok && m->lastICPCount > (float)count_thresh && m->lastICPError < err_thresh;
        lc[0] = 1.f; lc[2] = m->lastICPError; lc[3] = m->lastICPCount; lc[4] = (float)cov_ok; lc[5] = (float)accept;
        for (int k = 0; k < 16; k++) lc[6 + k] = m->pose[k];
        lc[22] = (float)cmax;
        st->lc_candidates += accept;
    }
    lc[23] = (float)st->lc_candidates;
    for (int k = 0; k < 24; k++) host_lc[k] = lc[k];
}

int ifx_tracker_alloc_m2m(ifx* h)
{
    if (h->d_m2m) return IFX_OK;
    Pyr& p = h->m2m;
    const size_t P = (size_t)h->P;
    HIPCHK(h, hipMalloc(&h->d_m2m, sizeof(DevState)));
    HIPCHK(h, hipMemset(h->d_m2m, 0, sizeof(DevState)));
    // the two renders of the detection are ONE allocation, [act_vertex | act_normal | act_image | act_inst | act_time | pad][old_* likewise]: on a spatially
    // sharded map the owners' winners of both travel in one collective (ifx_owner_exchange(h, 301, ...))
    h->lc_half = ((P * 42 + 15) / 16) * 16;
    HIPCHK(h, hipMalloc(&h->act_vertex, 2 * h->lc_half));
    HIPCHK(h, hipMemset(h->act_vertex, 0, 2 * h->lc_half));
    h->act_normal = h->act_vertex + 4 * P; h->act_image = (uint8_t*)(h->act_normal + 4 * P); h->act_inst = h->act_image + 4 * P; h->act_time = (uint16_t*)(h->act_inst + 4 * P);
    h->old_vertex = (float*)((uint8_t*)h->act_vertex + h->lc_half);
    h->old_normal = h->old_vertex + 4 * P; h->old_image = (uint8_t*)(h->old_normal + 4 * P); h->old_inst = h->old_image + 4 * P; h->old_time = (uint16_t*)(h->old_inst + 4 * P);
    HIPCHK(h, hipHostMalloc((void**)&h->h_lc, 24 * 4, hipHostMallocDefault));
    memset(h->h_lc, 0, 24 * 4);
    const int maxb = 1024;
    HIPCHK(h, hipMalloc(&p.acc, (3 * IFX_ACC_REPL * IFX_ACC_STRIDE * sizeof(double)))); HIPCHK(h, hipMemset(p.acc, 0, (3 * IFX_ACC_REPL * IFX_ACC_STRIDE * sizeof(double)))); HIPCHK(h, hipMalloc(&p.res_partials, (size_t)h->res_rows * 2 * 4));
    HIPCHK(h, hipMalloc(&p.ticket, 512));
    HIPCHK(h, hipMemset(p.ticket, 0, 512));
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        p.w[i] = h->w >> i; p.h[i] = h->h >> i;
        const size_t n = (size_t)p.w[i] * p.h[i];
        p.depth_tmp[i] = nullptr; p.lastnext_img[i] = nullptr;
        HIPCHK(h, hipMalloc(&p.vmap_curr[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_curr[i], n * 12));
        HIPCHK(h, hipMalloc(&p.vmap_cam[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_cam[i], n * 12));
        HIPCHK(h, hipMalloc(&p.vmap_prev[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_prev[i], n * 12));
        HIPCHK(h, hipMalloc(&p.last_depth[i], n * 4)); HIPCHK(h, hipMalloc(&p.next_depth[i], n * 4));
        HIPCHK(h, hipMalloc(&p.last_img[i], n)); HIPCHK(h, hipMalloc(&p.next_img[i], n));
        HIPCHK(h, hipMalloc(&p.didx[i], n * 2)); HIPCHK(h, hipMalloc(&p.didy[i], n * 2));
        HIPCHK(h, hipMalloc(&p.cloud[i], n * 12)); HIPCHK(h, hipMalloc(&p.corres[i], n * 8));
    }
    return IFX_OK;
}
static void free_m2m(ifx* h)
{
    if (!h->d_m2m) return;
    Pyr& p = h->m2m;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        hipFree(p.vmap_curr[i]); hipFree(p.nmap_curr[i]); hipFree(p.vmap_cam[i]); hipFree(p.nmap_cam[i]); hipFree(p.vmap_prev[i]); hipFree(p.nmap_prev[i]);
        hipFree(p.last_depth[i]); hipFree(p.next_depth[i]); hipFree(p.last_img[i]); hipFree(p.next_img[i]); hipFree(p.didx[i]); hipFree(p.didy[i]);
        hipFree(p.cloud[i]); hipFree(p.corres[i]);
    }
    hipFree(h->d_m2m); hipFree(h->act_vertex);   // (act_* and old_* are one allocation)
    hipFree(p.acc); hipFree(p.res_partials); hipFree(p.ticket);
    if (h->h_lc) hipHostFree(h->h_lc);
    h->d_m2m = nullptr;
}
int ifx_tracker_m2m_begin(ifx* h)
{
    LAUNCH(h, "m2m_begin", dim3(1), dim3(64), k_m2m_begin, (const DevState*)h->d_state, h->d_m2m);
    return IFX_OK;
}
// modelToModel.initICPModel / initRGBModel (old render, global frame at currPose), initICP / initRGB (active render), then
// getIncrementalTransformation(trans, rot, false, 10, pyramid, fastOdom, false) (EF/ElasticFusion.cpp:528-545) and the gates
int ifx_tracker_loop_closure(ifx* h)
{
    DevState* m = h->d_m2m;
    LAUNCH(h, "m2m_count", dim3(cdiv(h->P, 4096)), dim3(256), k_m2m_count, m, (const float4*)h->old_vertex, h->P);
    LAUNCH(h, "m2m_arm", dim3(1), dim3(64), k_m2m_arm, m);
    tracker_init_model(h, m, h->m2m, 10.0f, h->old_vertex, h->old_normal, h->old_image, nullptr, nullptr, nullptr);
    tracker_init_frame_maps(h, m, h->m2m, h->act_vertex, h->act_normal, h->act_image);
    tracker_run(h, m, h->m2m, 10.0f, 0, 1.0f, 1, false);
    LAUNCH(h, "m2m_decide", dim3(1), dim3(64), k_m2m_decide, h->d_state, (const DevState*)m, h->lc_count_thresh, h->lc_err_thresh, h->lc_cov_thresh, h->h_lc);
    return IFX_OK;
}

// ---- Ferns (EF/Ferns.cpp) GPU contact no. 2: RGBDOdometry on two small renders (a stored keyframe against the current frame, Ferns.cpp:558-592),
// and in general the texture-initialised tracker (initICPModel / initRGBModel + initICP(vertices, normals) / initRGB + getIncrementalTransformation)
// as a stage: host float4 maps in, estimate out.  Runs on the model-to-model instance of this handle with the handle's own configuration
// (resolution, intrinsics, icp_weight, pyramid, fast_odom; no SO(3)): for ferns the caller creates a handle at width/8 x height/8 with
// icp_weight 100, pyramid 0 -- what Ferns::findFrame passes.
__global__ void k_m2m_prepare(DevState* m, const float* __restrict__ pose16)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) m->pose[k] = pose16[k];
    pose_inverse(m->pose, m->pose_inv);
    m->dense_enough = 1; m->count = 0; m->skip = 0;
}
extern "C" int ifx_track_maps(ifx_t* h, const float* model_v4, const float* model_n4, const uint8_t* model_rgba, const float* cur_v4, const float* cur_n4,
                              const uint8_t* cur_rgba, float* pose16, float* diag8)
{
    if (!h || !model_v4 || !model_n4 || !cur_v4 || !cur_n4 || !pose16) return IFX_E_INVALID;
    int r = ifx_tracker_alloc_m2m(h);
    if (r) return r;
    if (h->stream_c) { HIPCHK(h, hipStreamSynchronize(h->stream_c)); h->lc_pending = 0; }
    ifx_drop_tracked(h);
    const size_t P = (size_t)h->P;
    HIPCHK(h, hipMemcpyAsync(h->old_vertex, model_v4, P * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->old_normal, model_n4, P * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->act_vertex, cur_v4, P * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->act_normal, cur_n4, P * 16, hipMemcpyHostToDevice, h->stream));
    if (model_rgba) HIPCHK(h, hipMemcpyAsync(h->old_image, model_rgba, P * 4, hipMemcpyHostToDevice, h->stream));
    else HIPCHK(h, hipMemsetAsync(h->old_image, 0, P * 4, h->stream));
    if (cur_rgba) HIPCHK(h, hipMemcpyAsync(h->act_image, cur_rgba, P * 4, hipMemcpyHostToDevice, h->stream));
    else HIPCHK(h, hipMemsetAsync(h->act_image, 0, P * 4, h->stream));
    float* slot = h->d_scratch + 4 * 16;
    HIPCHK(h, hipMemcpyAsync(slot, pose16, 64, hipMemcpyHostToDevice, h->stream));
    DevState* m = h->d_m2m;
    LAUNCH(h, "m2m_prepare", dim3(1), dim3(64), k_m2m_prepare, m, (const float*)slot);
    tracker_init_model(h, m, h->m2m, h->cfg.icp_weight, h->old_vertex, h->old_normal, h->old_image, nullptr, nullptr, nullptr);
    tracker_init_frame_maps(h, m, h->m2m, h->act_vertex, h->act_normal, h->act_image);
    tracker_run(h, m, h->m2m, h->cfg.icp_weight, 0, 1.0f, 1, false);
    DevState hs;
    HIPCHK(h, hipMemcpyAsync(&hs, m, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(pose16, hs.pose, 64);
    if (diag8) {
        diag8[0] = hs.lastICPError; diag8[1] = hs.lastICPCount; diag8[2] = hs.lastRGBError; diag8[3] = hs.lastRGBCount;
        diag8[4] = 0; diag8[5] = 0; diag8[6] = 0; diag8[7] = 0;
    }
    return IFX_OK;
}

int ifx_tracker_commit(ifx* h)
{
    LAUNCH(h, "commit_pose", dim3(1), dim3(64), k_commit_pose, h->d_state, h->d_list_ctr);
    return IFX_OK;
}

// pose supplied by the caller instead of tracking (inPose != NULL, EF/ElasticFusion.cpp:428-431)
__global__ void k_set_pose(DevState* st, const float* pose16)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) { st->last_pose[k] = st->pose[k]; }
    for (int k = 0; k < 16; k++) st->pose[k] = pose16[k];
}
int ifx_tracker_external_pose(ifx* h, const float* d_pose16, float weight_mult)
{
    LAUNCH(h, "set_pose", dim3(1), dim3(64), k_set_pose, h->d_state, d_pose16);
    LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, h->d_state, 0, 0, weight_mult, 1, h->d_list_ctr);
    return IFX_OK;
}
int ifx_tracker_set_weight(ifx* h, float weight_mult)
{
    // re-evaluates the velocity weighting with the caller's multiplier (weightMultiplier argument)
    LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, h->d_state, 0, 0, weight_mult, 1, (unsigned int*)nullptr);
    return IFX_OK;
}

// ======================================================================= stage API (unit parity)

// stage API: the exact totals of one quantity (0 icp, 1 rgb, 2 so3) from the accumulator rows, which are cleared again
static double* stage_acc(ifx* h, int which) { return h->pyr.acc + (size_t)which * IFX_ACC_REPL * IFX_ACC_STRIDE; }
static int final_sum(ifx* h, int which, int nv, float* out_host)
{
    double hp[IFX_ACC_REPL * IFX_ACC_STRIDE];
    HIPCHK(h, hipMemcpyAsync(hp, stage_acc(h, which), sizeof(hp), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemsetAsync(stage_acc(h, which), 0, sizeof(hp), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int r = 0; r < IFX_ACC_REPL; r++) s += hp[r * IFX_ACC_STRIDE + k];
        out_host[k] = (float)s;
    }
    return IFX_OK;
}

extern "C" int ifx_icp_step(ifx_t* h, const float* Rcurr9, const float* tcurr3, const float* d_vmap_curr, const float* d_nmap_curr, const float* Rprev_inv9,
                            const float* tprev3, float fx, float fy, float cx, float cy, const float* d_vmap_g_prev, const float* d_nmap_g_prev, float dist_thres,
                            float angle_thres, int w, int hgt, float* out29_host)
{
    if (!h || !out29_host) return IFX_E_INVALID;
    ifx_drop_tracked(h);   // (stage call: it rewrites what a run enqueued ahead reads)
    IcpArgs ia;
    memcpy(ia.Rcurr, Rcurr9, 36); memcpy(ia.tcurr, tcurr3, 12); memcpy(ia.Rprev_inv, Rprev_inv9, 36); memcpy(ia.tprev, tprev3, 12);
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "icp", dim3(nb), dim3(RED_THREADS), k_icp, (const DevState*)nullptr, ia, d_vmap_curr, d_nmap_curr, d_vmap_g_prev, d_nmap_g_prev, fx, fy, cx, cy, dist_thres,
           angle_thres, w, hgt, stage_acc(h, 0));
    return final_sum(h, 0, 29, out29_host);
}

extern "C" int ifx_rgb_residual(ifx_t* h, float min_scale, const int16_t* d_didx, const int16_t* d_didy, const float* d_last_depth, const float* d_next_depth,
                                const uint8_t* d_last_img, const uint8_t* d_next_img, void* d_corres8, float max_depth_delta, const float* kt3, const float* krkinv9,
                                int w, int hgt, int* count_host, int* sigma_host)
{
    if (!h) return IFX_E_INVALID;
    ifx_drop_tracked(h);   // (stage call: it rewrites what a run enqueued ahead reads)
    ResArgs ra;
    memcpy(ra.krkinv, krkinv9, 36); memcpy(ra.kt, kt3, 12);
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "rgb_residual", dim3(nb), dim3(RED_THREADS), k_rgb_residual, (const DevState*)nullptr, ra, min_scale, d_didx, d_didy, d_last_depth, d_next_depth, d_last_img,
           d_next_img, (Corres8*)d_corres8, max_depth_delta, w, hgt, h->res_partials);
    std::vector<int> hp((size_t)nb * 2);
    HIPCHK(h, hipMemcpyAsync(hp.data(), h->res_partials, hp.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int c = 0, s = 0;
    for (int b = 0; b < nb; b++) { c += hp[2 * b]; s += hp[2 * b + 1]; }
    if (count_host) *count_host = c;
    if (sigma_host) *sigma_host = s;
    return IFX_OK;
}

extern "C" int ifx_rgb_step(ifx_t* h, const void* d_corres8, float sigma, const float* d_cloud3, float fx, float fy, const int16_t* d_didx, const int16_t* d_didy,
                            float sobel_scale, int w, int hgt, float* out29_host)
{
    if (!h || !out29_host) return IFX_E_INVALID;
    ifx_drop_tracked(h);   // (stage call: it rewrites what a run enqueued ahead reads)
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "rgb_step", dim3(nb), dim3(RED_THREADS), k_rgb_step, (const Corres8*)d_corres8, sigma, (const int*)nullptr, 0, d_cloud3, fx, fy, d_didx, d_didy, sobel_scale, w,
           hgt, stage_acc(h, 1));
    return final_sum(h, 1, 29, out29_host);
}

extern "C" int ifx_so3_step(ifx_t* h, const uint8_t* d_last_img, const uint8_t* d_next_img, const float* image_basis9, const float* kinv9, const float* krlr9, int w,
                            int hgt, float* out11_host)
{
    if (!h || !out11_host) return IFX_E_INVALID;
    ifx_drop_tracked(h);   // (stage call: it rewrites what a run enqueued ahead reads)
    So3Args sa;
    memcpy(sa.ib, image_basis9, 36); memcpy(sa.kinv, kinv9, 36); memcpy(sa.krlr, krlr9, 36);
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "so3", dim3(nb), dim3(RED_THREADS), k_so3, (const DevState*)nullptr, sa, d_last_img, d_next_img, w, hgt, stage_acc(h, 2));
    return final_sum(h, 2, 11, out11_host);
}

__global__ void k_write_pose(DevState* st, const float* p)
{
    if (threadIdx.x == 0) for (int k = 0; k < 16; k++) st->pose[k] = p[k];
}
__global__ void k_set_dense(DevState* st, int v) { if (threadIdx.x == 0) st->dense_enough = v; }

extern "C" int ifx_track_pair(ifx_t* h, const float* model_v4, const float* model_n4, const uint8_t* model_rgba, const uint8_t* prev_rgb, const uint16_t* depth_filtered,
                              const uint8_t* rgb, float* pose16, float* diag8)
{
    if (!h || !pose16) return IFX_E_INVALID;
    ifx_drop_tracked(h);   // (stage call: it rewrites what a run enqueued ahead reads)
    size_t P = (size_t)h->P;
    HIPCHK(h, hipMemcpyAsync(h->pred_vertex, model_v4, P * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->pred_normal, model_n4, P * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->pred_image, model_rgba, P * 4, hipMemcpyHostToDevice, h->stream));
    if (prev_rgb) {
        HIPCHK(h, hipMemcpyAsync(h->rgb, prev_rgb, P * 3, hipMemcpyHostToDevice, h->stream));
        ifx_tracker_init_first(h);
    }
    HIPCHK(h, hipMemcpyAsync(h->rgb, rgb, P * 3, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->depth_filt, depth_filtered, P * 2, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_scratch + 6 * 16, pose16, 64, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "write_pose", dim3(1), dim3(64), k_write_pose, h->d_state, h->d_scratch + 6 * 16);
    LAUNCH(h, "set_dense", dim3(1), dim3(64), k_set_dense, h->d_state, 1);
    tracker_init_model(h, h->d_state, h->pyr, h->cfg.icp_weight, h->pred_vertex, h->pred_normal, h->pred_image, h->fill_vertex, h->fill_normal, h->fill_image);
    ifx_tracker_frame_side(h, 0);
    tracker_run(h, h->d_state, h->pyr, h->cfg.icp_weight, h->cfg.so3, 1.0f, 1, true);
    DevState hs;
    HIPCHK(h, hipMemcpyAsync(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(pose16, hs.pose, 64);
    if (diag8) {
        diag8[0] = hs.lastICPError; diag8[1] = hs.lastICPCount; diag8[2] = hs.lastRGBError; diag8[3] = hs.lastRGBCount;
        diag8[4] = hs.lastSO3Error; diag8[5] = hs.lastSO3Count; diag8[6] = hs.weighting; diag8[7] = 0;
    }
    return IFX_OK;
}

// Stage entry for the pyramid builders (SURVEY 8(b) "tracker stage API"): createVMap / createNMap / pyrDown / pyrDownGaussF / pyrDownUcharGauss / resizeVMap / resizeNMap /
// tranformMaps / verticesToDepth / imageBGRToIntensity / computeDerivativeImages / projectToPointCloud (EF/Cuda/cudafuncs.cuh:64-183) as RGBDOdometry::initICP / initRGB /
// initICPModel / initRGBModel chain them (EF/Utils/RGBDOdometry.cpp:118-247, 287-293).  Device pointers in, device pointers out; the kernels are the ones a frame runs
// (k_intensity, k_frame_down, k_frame_maps; k_model_l0, k_model_down), on the handle's own pyramid buffers, and what the caller asked for is copied out device to device.
extern "C" int ifx_build_pyramids(ifx_t* h, const uint16_t* d_depth_filtered, const uint8_t* d_rgb, const float* d_model_v4, const float* d_model_n4, const uint8_t* d_model_rgba,
                                  const float* model_pose16, ifx_pyramids* out)
{
    if (!h || !out || (!d_depth_filtered != !d_rgb) || (!d_model_v4 != !d_model_n4) || (!d_model_v4 != !d_model_rgba) || (d_model_v4 && !model_pose16) || (!d_rgb && !d_model_v4)) return IFX_E_INVALID;
    ifx_drop_tracked(h);   // (stage call: it rewrites what a run enqueued ahead reads)
    const size_t P = (size_t)h->P;
    Pyr& p = h->pyr;
    auto give = [&](void* dst, const void* src, size_t bytes) { return (!dst || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream) == hipSuccess) ? 0 : 1; };
    int bad = 0;
    if (d_rgb) {
        HIPCHK(h, hipMemcpyAsync(h->rgb, d_rgb, P * 3, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->depth_filt, d_depth_filtered, P * 2, hipMemcpyDeviceToDevice, h->stream));
        tracker_init_frame(h, h->depth_filt, h->rgb);
        for (int l = 0; l < IFX_NUM_PYRS; l++) {
            const size_t n = (size_t)p.w[l] * p.h[l];
            bad += give(out->depth[l], p.depth_tmp[l], n * 2) + give(out->vmap_curr[l], p.vmap_curr[l], n * 12) + give(out->nmap_curr[l], p.nmap_curr[l], n * 12) +
                   give(out->next_img[l], p.next_img[l], n) + give(out->didx[l], p.didx[l], n * 2) + give(out->didy[l], p.didy[l], n * 2);
        }
    }
    if (d_model_v4) {
        HIPCHK(h, hipMemcpyAsync(h->pred_vertex, d_model_v4, P * 16, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->pred_normal, d_model_n4, P * 16, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->pred_image, d_model_rgba, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_scratch + 6 * 16, model_pose16, 64, hipMemcpyHostToDevice, h->stream));
        LAUNCH(h, "write_pose", dim3(1), dim3(64), k_write_pose, h->d_state, h->d_scratch + 6 * 16);
        LAUNCH(h, "set_dense", dim3(1), dim3(64), k_set_dense, h->d_state, 1);   // (the prediction as given: no fill-in substitution)
        tracker_init_model(h, h->d_state, h->pyr, 10.0f /* with the photometric term: the point clouds are built */, h->pred_vertex, h->pred_normal, h->pred_image, h->fill_vertex, h->fill_normal,
                           h->fill_image);
        const int iterations[3] = {h->cfg.fast_odom ? 3 : 10, h->cfg.pyramid ? 5 : 0, h->cfg.pyramid ? 4 : 0};
        for (int l = 0; l < IFX_NUM_PYRS; l++) {
            const size_t n = (size_t)p.w[l] * p.h[l];
            bad += give(out->vmap_g_prev[l], p.vmap_prev[l], n * 12) + give(out->nmap_g_prev[l], p.nmap_prev[l], n * 12) + give(out->last_depth[l], p.last_depth[l], n * 4) +
                   give(out->last_img[l], p.last_img[l], n);
            if (out->cloud[l]) {
                if (iterations[l] <= 0) { h->err = "ifx_build_pyramids: the handle's configuration runs no iteration at level " + std::to_string(l) + " (pyramid = 0): no point cloud there"; return IFX_E_STATE; }
                bad += give(out->cloud[l], p.cloud[l], n * 12);
            }
        }
    }
    if (bad) { (void)hipGetLastError(); h->err = "ifx_build_pyramids: a device-to-device copy into the caller's buffers failed"; return IFX_E_HIP; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}

extern "C" int ifx_tracker_buffer_download(ifx_t* h, const char* name, int l, void* out, int64_t max_bytes)
{
    if (!h || !name || l < 0 || l >= IFX_NUM_PYRS) return IFX_E_INVALID;
    std::string s(name);
    const bool second = s.rfind("m2m:", 0) == 0;   // "m2m:<name>": the model-to-model instance (loop-closure detection, ifx_track_maps)
    if (second) {
        if (!h->d_m2m) { h->err = "the model-to-model tracker was never used"; return IFX_E_STATE; }
        if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
        s = s.substr(4);
    }
    Pyr& p = second ? h->m2m : h->pyr;
    size_t n = (size_t)p.w[l] * p.h[l];
    const void* src = nullptr;
    size_t bytes = 0;
    if (s == "vmap_curr") { src = p.vmap_curr[l]; bytes = n * 12; }
    else if (s == "nmap_curr") { src = p.nmap_curr[l]; bytes = n * 12; }
    else if (s == "vmap_prev") { src = p.vmap_prev[l]; bytes = n * 12; }
    else if (s == "nmap_prev") { src = p.nmap_prev[l]; bytes = n * 12; }
    else if (s == "last_depth") { src = p.last_depth[l]; bytes = n * 4; }
    else if (s == "next_depth") { src = p.next_depth[l] ? p.next_depth[l] : p.last_depth[l]; bytes = n * 4; }
    else if (s == "last_img") { src = p.last_img[l]; bytes = n; }
    else if (s == "next_img") { src = p.next_img[l]; bytes = n; }
    else if (s == "lastnext_img") { src = p.lastnext_img[l]; bytes = n; }
    else if (s == "didx") { src = p.didx[l]; bytes = n * 2; }
    else if (s == "didy") { src = p.didy[l]; bytes = n * 2; }
    else if (s == "cloud") { src = p.cloud[l]; bytes = n * 12; }
    else if (s == "corres") { src = p.corres[l]; bytes = n * 8; }
    else if (s == "depth_tmp") { src = p.depth_tmp[l]; bytes = n * 2; }
    else { h->err = "unknown tracker buffer " + s; return IFX_E_INVALID; }
    if ((int64_t)bytes > max_bytes) return IFX_E_INVALID;
    HIPCHK(h, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return (int)bytes;
}

#ifdef IFX_STAMPS
void ifx_debug_copy2(long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg2), 64); }
#endif
