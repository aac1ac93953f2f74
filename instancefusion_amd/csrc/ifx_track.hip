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
__global__ __launch_bounds__(BIL_BX* BIL_BY) void k_bilateral_metric(const uint16_t* __restrict__ in, uint16_t* __restrict__ filt,
                                                                      float* __restrict__ dm, float* __restrict__ dmf, int w, int h, float maxD)
{
    __shared__ uint16_t tile[BIL_BY + 2 * BIL_R][BIL_BX + 2 * BIL_R + 2];
    const int bx = blockIdx.x * BIL_BX, by = blockIdx.y * BIL_BY;
    const int tid = threadIdx.y * BIL_BX + threadIdx.x;
    const int TW = BIL_BX + 2 * BIL_R, TH = BIL_BY + 2 * BIL_R;
    for (int i = tid; i < TW * TH; i += BIL_BX * BIL_BY) {
        int ty = i / TW, tx = i - ty * TW;
        int gx = bx + tx - BIL_R, gy = by + ty - BIL_R;
        uint16_t v = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) v = in[gy * w + gx];
        tile[ty][tx] = v;
    }
    __syncthreads();
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= w || y >= h) return;
    const unsigned int maxv = (unsigned int)(maxD * 1000.0f);
    unsigned int value = tile[threadIdx.y + BIL_R][threadIdx.x + BIL_R];
    unsigned int outv = 0;
    if (!(value > maxv || value < 300u)) {
        const float ss = 0.024691358f, sc = 0.000555556f;
        const int D = BIL_R * 2 + 1;
        int tx1 = min(x - D / 2 + D, w), ty1 = min(y - D / 2 + D, h);
        float sum1 = 0, sum2 = 0;
        for (int cy = max(y - D / 2, 0); cy < ty1; ++cy)
            for (int cx = max(x - D / 2, 0); cx < tx1; ++cx) {
                unsigned int tmp = tile[cy - by + BIL_R][cx - bx + BIL_R];
                float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) + ((float)y - (float)cy) * ((float)y - (float)cy);
                float color2 = ((float)value - (float)tmp) * ((float)value - (float)tmp);
                float weight = ifx_expf(-(space2 * ss + color2 * sc));
                sum1 += (float)tmp * weight;
                sum2 += weight;
            }
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

// pyrDownGaussKernel, EF/Cuda/cudafuncs.cu:57-94
__global__ void k_pyrdown_u16(const uint16_t* __restrict__ src, int sw, int sh, uint16_t* __restrict__ dst)
{
    int dw = sw / 2, dh = sh / 2;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int D = 5;
    const float sigma_color = 30;
    const float weights[3] = {0.375f, 0.25f, 0.0625f};
    int center = src[(2 * y) * sw + 2 * x];
    int x_mi = max(0, 2 * x - D / 2) - 2 * x, y_mi = max(0, 2 * y - D / 2) - 2 * y;
    int x_ma = min(sw, 2 * x - D / 2 + D) - 2 * x, y_ma = min(sh, 2 * y - D / 2 + D) - 2 * y;
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
__global__ void k_vmap_nmap(const uint16_t* __restrict__ depth, int w, int h, float fx_inv, float fy_inv, float cx, float cy, float cutoff,
                            float* __restrict__ vmap, float* __restrict__ nmap)
{
    int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y * blockDim.y + threadIdx.y;
    if (u >= w || v >= h) return;
    const float qn = qnan_f();
    v3 v00;
    bool ok00 = vert_from_depth(depth, w, u, v, fx_inv, fy_inv, cx, cy, cutoff, v00);
    vmap[v * w + u] = ok00 ? v00.x : qn;
    vmap[(v + h) * w + u] = ok00 ? v00.y : qn;
    vmap[(v + 2 * h) * w + u] = ok00 ? v00.z : qn;
    v3 r = v3m(qn, qn, qn);
    if (!(u == w - 1 || v == h - 1) && ok00) {
        v3 v01, v10;
        bool ok01 = vert_from_depth(depth, w, u + 1, v, fx_inv, fy_inv, cx, cy, cutoff, v01);
        bool ok10 = vert_from_depth(depth, w, u, v + 1, fx_inv, fy_inv, cx, cy, cutoff, v10);
        if (ok01 && ok10) r = normalized(cross(v01 - v00, v10 - v00));
    }
    nmap[v * w + u] = r.x;
    nmap[(v + h) * w + u] = r.y;
    nmap[(v + 2 * h) * w + u] = r.z;
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

// pyrDownKernelGaussF, EF/Cuda/cudafuncs.cu:332-363
__global__ void k_pyrdown_gauss_f(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst)
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
            float s = src[cy * sw + cx];
            if (!(s != s)) {
                float g = c_gauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                sum += s * g;
                count += (int)g;
            }
        }
    dst[y * dw + x] = (float)(sum / (float)count);
}

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

// applyKernel, EF/Cuda/cudafuncs.cu:583-607
__global__ void k_sobel(const uint8_t* __restrict__ img, int w, int h, int16_t* __restrict__ dx, int16_t* __restrict__ dy)
{
    const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float dxVal = 0, dyVal = 0;
    int k = 8;
    for (int j = max(y - 1, 0); j <= min(y + 1, h - 1); j++)
        for (int i = max(x - 1, 0); i <= min(x + 1, w - 1); i++) {
            dxVal += (float)img[j * w + i] * gsx[k];
            dyVal += (float)img[j * w + i] * gsy[k];
            --k;
        }
    dx[y * w + x] = (int16_t)dxVal;
    dy[y * w + x] = (int16_t)dyVal;
}

// copyMapsKernel (EF/Cuda/cudafuncs.cu:270-310) + verticesToDepthKernel (:526-537) + intensity of the
// model image, reading either the prediction or the fill-in maps according to DevState::dense_enough
// (EF/ElasticFusion.cpp:337-346).
__global__ void k_model_level0(const DevState* __restrict__ st, const float* __restrict__ pv, const float* __restrict__ pn, const uint8_t* __restrict__ pi,
                               const float* __restrict__ fv, const float* __restrict__ fn, const uint8_t* __restrict__ fi, int w, int h,
                               float* __restrict__ vmap, float* __restrict__ nmap, float* __restrict__ depth, uint8_t* __restrict__ img, float cutoff)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const bool fill = !st->dense_enough;
    const float4 v = reinterpret_cast<const float4*>(fill ? fv : pv)[y * w + x];
    const float4 n = reinterpret_cast<const float4*>(fill ? fn : pn)[y * w + x];
    const uint8_t* s = (fill ? fi : pi) + (size_t)(y * w + x) * 4;
    const float qn = qnan_f();
    bool ok = !(v.z == 0);
    vmap[y * w + x] = ok ? v.x : qn;
    vmap[(y + h) * w + x] = ok ? v.y : qn;
    vmap[(y + 2 * h) * w + x] = ok ? v.z : qn;
    nmap[y * w + x] = ok ? n.x : qn;
    nmap[(y + h) * w + x] = ok ? n.y : qn;
    nmap[(y + 2 * h) * w + x] = ok ? n.z : qn;
    depth[y * w + x] = (v.z > cutoff || v.z <= 0) ? qn : v.z;
    img[y * w + x] = (uint8_t)(int)((float)s[0] * 0.114f + (float)s[1] * 0.299f + (float)s[2] * 0.587f);
}

// resizeMapKernel<normalize>, EF/Cuda/cudafuncs.cu:365-416, both maps in one launch
__global__ void k_resize_maps(const float* __restrict__ vin, const float* __restrict__ nin, int sw, int sh, float* __restrict__ vout, float* __restrict__ nout)
{
    int dw = sw / 2, dh = sh / 2;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const float qn = qnan_f();
    int xs = x * 2, ys = y * 2;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const float* in = m ? nin : vin;
        float* out = m ? nout : vout;
        float x00 = in[ys * sw + xs], x01 = in[ys * sw + xs + 1], x10 = in[(ys + 1) * sw + xs], x11 = in[(ys + 1) * sw + xs + 1];
        if ((x00 != x00) || (x01 != x01) || (x10 != x10) || (x11 != x11)) {
            out[y * dw + x] = qn; out[(y + dh) * dw + x] = qn; out[(y + 2 * dh) * dw + x] = qn;
            continue;
        }
        v3 n;
        n.x = (x00 + x01 + x10 + x11) / 4;
        const float* py = in + sh * sw;
        n.y = (py[ys * sw + xs] + py[ys * sw + xs + 1] + py[(ys + 1) * sw + xs] + py[(ys + 1) * sw + xs + 1]) / 4;
        const float* pz = in + 2 * sh * sw;
        n.z = (pz[ys * sw + xs] + pz[ys * sw + xs + 1] + pz[(ys + 1) * sw + xs] + pz[(ys + 1) * sw + xs + 1]) / 4;
        if (m) n = normalized(n);
        out[y * dw + x] = n.x; out[(y + dh) * dw + x] = n.y; out[(y + 2 * dh) * dw + x] = n.z;
    }
}

// tranformMapsKernel, EF/Cuda/cudafuncs.cu:206-248 (camera-frame maps -> global), out of place
__global__ void k_transform_maps(const DevState* __restrict__ st, const float* __restrict__ vsrc, const float* __restrict__ nsrc, int w, int h,
                                 float* __restrict__ vdst, float* __restrict__ ndst)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float* P = st->pose;
    const float qn = qnan_f();
    v3 vs = v3m(vsrc[y * w + x], vsrc[(y + h) * w + x], vsrc[(y + 2 * h) * w + x]);
    v3 vd = v3m(qn, qn, qn);
    if (!(vs.x != vs.x)) vd = xf_dir(P, vs) + v3m(P[3], P[7], P[11]);
    vdst[y * w + x] = vd.x; vdst[(y + h) * w + x] = vd.y; vdst[(y + 2 * h) * w + x] = vd.z;
    v3 ns = v3m(nsrc[y * w + x], nsrc[(y + h) * w + x], nsrc[(y + 2 * h) * w + x]);
    v3 nd = v3m(qn, qn, qn);
    if (!(ns.x != ns.x)) nd = xf_dir(P, ns);
    ndst[y * w + x] = nd.x; ndst[(y + h) * w + x] = nd.y; ndst[(y + 2 * h) * w + x] = nd.z;
}

// projectPointsKernel, EF/Cuda/cudafuncs.cu:641-659
__global__ void k_project_cloud(const float* __restrict__ depth, int w, int h, float invFx, float invFy, float cx, float cy, float* __restrict__ cloud)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float z = depth[y * w + x];
    cloud[(y * w + x) * 3 + 0] = (float)((x - cx) * z * invFx);
    cloud[(y * w + x) * 3 + 1] = (float)((y - cy) * z * invFy);
    cloud[(y * w + x) * 3 + 2] = z;
}

// ======================================================================= reductions (a4-a7)

#define RED_THREADS 256
#define RED_WAVES (RED_THREADS / 64)

// Block reduction of NV floats per thread: wave64 shuffle tree, one LDS row per wave, the first
// NV threads add the rows in wave order and store the block partial.
template <int NV>
__device__ inline void block_reduce_store(float* acc, float* __restrict__ out /* [NV] for this block */)
{
    __shared__ float lds[RED_WAVES][NV];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        float v = wave_sum(acc[k]);
        if (lane == 0) lds[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0;
#pragma unroll
        for (int wv = 0; wv < RED_WAVES; wv++) s += lds[wv][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

__device__ inline void products7(const float* row, bool found, float* acc)
{
    int s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 7; j++) acc[s++] += row[i] * row[j];
    acc[27] += row[6] * row[6];
    acc[28] += found ? 1.0f : 0.0f;
}

// ICPReduction, EF/Cuda/reduce.cu:257-411.  Rcurr/tcurr/Rprev_inv/tprev come from DevState (or from
// explicit arguments for the stage API when st == nullptr).
struct IcpArgs { float Rcurr[9], tcurr[3], Rprev_inv[9], tprev[3]; };
__global__ __launch_bounds__(RED_THREADS) void k_icp(const DevState* __restrict__ st, IcpArgs ex, const float* __restrict__ vmap_curr,
                                                     const float* __restrict__ nmap_curr, const float* __restrict__ vmap_prev,
                                                     const float* __restrict__ nmap_prev, float fx, float fy, float cx, float cy, float distThres,
                                                     float angleThres, int w, int h, float* __restrict__ partials)
{
    const float* Rc = st ? st->Rcurr : ex.Rcurr;
    const float* tcp = st ? st->tcurr : ex.tcurr;
    const float* Rpi = st ? st->Rprev_inv : ex.Rprev_inv;
    const float* tpp = st ? st->tprev : ex.tprev;
    float Rcurr[9], Rprev_inv[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rcurr[k] = Rc[k]; Rprev_inv[k] = Rpi[k]; }
    const v3 tc = v3m(tcp[0], tcp[1], tcp[2]), tp = v3m(tpp[0], tpp[1], tpp[2]);
    float acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.f;
    const int N = w * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
        int y = i / w, x = i - y * w;
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        bool found = false;
        v3 vcurr = v3m(vmap_curr[i], vmap_curr[i + N], vmap_curr[i + 2 * N]);
        if (!(vcurr.x != vcurr.x)) {
            v3 vcurr_g = mulp(Rcurr, vcurr) + tc;
            v3 vcurr_cp = mulp(Rprev_inv, vcurr_g - tp);
            int ux = f2i_rn(vcurr_cp.x * fx / vcurr_cp.z + cx);
            int uy = f2i_rn(vcurr_cp.y * fy / vcurr_cp.z + cy);
            if (!(ux < 0 || uy < 0 || ux >= w || uy >= h || vcurr_cp.z < 0)) {
                int j = uy * w + ux;
                v3 vprev_g = v3m(vmap_prev[j], vmap_prev[j + N], vmap_prev[j + 2 * N]);
                v3 ncurr = v3m(nmap_curr[i], nmap_curr[i + N], nmap_curr[i + 2 * N]);
                v3 ncurr_g = mulp(Rcurr, ncurr);
                v3 nprev_g = v3m(nmap_prev[j], nmap_prev[j + N], nmap_prev[j + 2 * N]);
                float dist = norm(vprev_g - vcurr_g);
                float sine = norm(cross(ncurr_g, nprev_g));
                found = (sine < angleThres && dist <= distThres && !(ncurr.x != ncurr.x) && !(nprev_g.x != nprev_g.x));
                if (found) {
                    v3 s_cp = mulp(Rprev_inv, vcurr_g - tp);
                    v3 d_cp = mulp(Rprev_inv, vprev_g - tp);
                    v3 n_cp = mulp(Rprev_inv, nprev_g);
                    v3 c = cross(s_cp, n_cp);
                    row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
                    row[3] = c.x; row[4] = c.y; row[5] = c.z;
                    row[6] = dot(n_cp, s_cp - d_cp);
                }
            }
        }
        (void)x;
        products7(row, found, acc);
    }
    block_reduce_store<29>(acc, partials + (size_t)blockIdx.x * 32);
}

// 8-byte correspondence record (the reference's DataTerm is 16 B, EF/Cuda/types.cuh:75-81: `one`
// is the pixel's own coordinate and `valid` is folded into zx >= 0)
struct Corres8 { short zx, zy; float diff; };

// RGBResidual, EF/Cuda/reduce.cu:739-863
struct ResArgs { float krkinv[9], kt[3]; };
__global__ __launch_bounds__(RED_THREADS) void k_rgb_residual(const DevState* __restrict__ st, ResArgs ex, float minScale, const int16_t* __restrict__ dIdx,
                                                              const int16_t* __restrict__ dIdy, const float* __restrict__ lastDepth,
                                                              const float* __restrict__ nextDepth, const uint8_t* __restrict__ lastImage,
                                                              const uint8_t* __restrict__ nextImage, Corres8* __restrict__ corres, float maxDepthDelta,
                                                              int w, int h, int* __restrict__ partials)
{
    const float* kk = st ? st->krkinv : ex.krkinv;
    const float* ktp = st ? st->kt : ex.kt;
    float krk[9];
#pragma unroll
    for (int k = 0; k < 9; k++) krk[k] = kk[k];
    const float kt0 = ktp[0], kt1 = ktp[1], kt2 = ktp[2];
    const int border = 16;
    const int N = w * h;
    int cnt = 0, sig = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += blockDim.x * gridDim.x) {
        int i = k / w, j0 = k - i * w;
        Corres8 c;
        c.zx = -1; c.zy = -1; c.diff = 0.f;
        if (i >= border && i < h - border && j0 >= border && j0 < w - border && j0 < w - 5 && i < h - 1) {
            bool valid = true;
            for (int u = max(i - 2, 0); u < min(i + 2, h); u++)
                for (int v = max(j0 - 2, 0); v < min(j0 + 2, w); v++) valid = valid && (nextImage[u * w + v] > 0);
            if (valid) {
                short valx = dIdx[k], valy = dIdy[k];
                float mTwo = (float)((valx * valx) + (valy * valy));
                if (mTwo >= minScale) {
                    int y = i, x = j0;
                    float d1 = nextDepth[k];
                    if (!(d1 != d1)) {
                        float td1 = (float)(d1 * (krk[6] * x + krk[7] * y + krk[8]) + kt2);
                        int u0 = f2i_rn((d1 * (krk[0] * x + krk[1] * y + krk[2]) + kt0) / td1);
                        int v0 = f2i_rn((d1 * (krk[3] * x + krk[4] * y + krk[5]) + kt1) / td1);
                        if (u0 >= 0 && v0 >= 0 && u0 < w && v0 < h) {
                            float d0 = lastDepth[v0 * w + u0];
                            uint8_t li = lastImage[v0 * w + u0];
                            if (d0 > 0 && fabsf(td1 - d0) <= maxDepthDelta && li != 0) {
                                c.zx = (short)u0; c.zy = (short)v0;
                                c.diff = (float)nextImage[k] - (float)li;
                                cnt += 1;
                                sig += (int)(c.diff * c.diff);
                            }
                        }
                    }
                }
            }
        }
        corres[k] = c;
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
        partials[blockIdx.x * 2 + threadIdx.x] = s;
    }
}

// RGBReduction, EF/Cuda/reduce.cu:494-619.  sigma is either explicit (stage API) or derived from the
// residual pass's block partials with the reference's precedence quirk (EF/Utils/RGBDOdometry.cpp:461).
__global__ __launch_bounds__(RED_THREADS) void k_rgb_step(const Corres8* __restrict__ corres, float sigma_explicit, const int* __restrict__ res_partials,
                                                          int res_blocks, const float* __restrict__ cloud, float fx, float fy,
                                                          const int16_t* __restrict__ dIdx, const int16_t* __restrict__ dIdy, float sobelScale, int w, int h,
                                                          float* __restrict__ partials)
{
    float sigma = sigma_explicit;
    if (res_partials) {
        __shared__ int s_cs[2];
        if (threadIdx.x < 64) {
            int cnt = 0, sg = 0;
            for (int b = threadIdx.x; b < res_blocks; b += 64) { cnt += res_partials[2 * b]; sg += res_partials[2 * b + 1]; }
            cnt = wave_sum_i(cnt);
            sg = wave_sum_i(sg);
            if (threadIdx.x == 0) { s_cs[0] = cnt; s_cs[1] = sg; }
        }
        __syncthreads();
        int cnt = s_cs[0], sg = s_cs[1];
        float q = (float)sg / (float)cnt;
        sigma = (float)sqrt((double)((q == 0) ? 1 : cnt));
    }
    float acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.f;
    const int N = w * h;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += blockDim.x * gridDim.x) {
        Corres8 c = corres[k];
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        bool found = c.zx >= 0;
        if (found) {
            float wgt = sigma + fabsf(c.diff);
            wgt = wgt > 1.19209290E-07F ? 1.0f / wgt : 1.0f;
            if (sigma == -1) wgt = 1;
            row[6] = -wgt * c.diff;
            const float* cp = &cloud[(c.zy * w + c.zx) * 3];
            float X = cp[0], Y = cp[1], Z = cp[2];
            float invz = (float)(1.0 / Z);
            float dI_dx = wgt * sobelScale * dIdx[k];
            float dI_dy = wgt * sobelScale * dIdy[k];
            float v0 = dI_dx * fx * invz;
            float v1 = dI_dy * fy * invz;
            float v2 = -(v0 * X + v1 * Y) * invz;
            row[0] = v0; row[1] = v1; row[2] = v2;
            row[3] = -Z * v1 + Y * v2;
            row[4] = Z * v0 - X * v2;
            row[5] = -Y * v0 + X * v1;
        }
        products7(row, found, acc);
    }
    block_reduce_store<29>(acc, partials + (size_t)blockIdx.x * 32);
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
__global__ __launch_bounds__(RED_THREADS) void k_so3(const DevState* __restrict__ st, So3Args ex, const uint8_t* __restrict__ lastImage,
                                                     const uint8_t* __restrict__ nextImage, int w, int h, float* __restrict__ partials)
{
    if (st && st->so3_done) return;   // wave-uniform early exit once the host-free loop has converged
    const float* ibp = st ? st->imageBasis : ex.ib;
    const float* kip = st ? st->kinv : ex.kinv;
    const float* krp = st ? st->krlr : ex.krlr;
    float ib[9], kinv[9], krlr[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { ib[k] = ibp[k]; kinv[k] = kip[k]; krlr[k] = krp[k]; }
    float acc[11];
#pragma unroll
    for (int k = 0; k < 11; k++) acc[k] = 0.f;
    const int N = w * h;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += blockDim.x * gridDim.x) {
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
            for (int j = i; j < 4; j++) acc[s++] += row[i] * row[j];
        acc[9] += row[3] * row[3];
        acc[10] += found ? 1.0f : 0.0f;
    }
    block_reduce_store<11>(acc, partials + (size_t)blockIdx.x * 12);
}

// ======================================================================= device-side solve (a8)

template <typename T>
__device__ void ldlt_solve(int n, const T* Ain, const T* bin, T* x, T tiny)
{
    T A[36], y[6];
    int p[6];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int i = 0; i < n; i++) p[i] = i;
    for (int k = 0; k < n; k++) {
        int piv = k;
        T big = A[k * n + k] < 0 ? -A[k * n + k] : A[k * n + k];
        for (int i = k + 1; i < n; i++) {
            T v = A[i * n + i] < 0 ? -A[i * n + i] : A[i * n + i];
            if (v > big) { big = v; piv = i; }
        }
        if (piv != k) {
            for (int j = 0; j < n; j++) { T t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            for (int j = 0; j < n; j++) { T t = A[j * n + k]; A[j * n + k] = A[j * n + piv]; A[j * n + piv] = t; }
            int t = p[k]; p[k] = p[piv]; p[piv] = t;
        }
        T d = A[k * n + k];
        if (big <= (T)0) {
            for (int i = k + 1; i < n; i++) A[i * n + k] = 0;
            continue;
        }
        for (int i = k + 1; i < n; i++) A[i * n + k] = A[i * n + k] / d;
        for (int i = k + 1; i < n; i++)
            for (int j = k + 1; j < n; j++) A[i * n + j] -= A[i * n + k] * d * A[j * n + k];
    }
    for (int i = 0; i < n; i++) y[i] = bin[p[i]];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) y[i] -= A[i * n + j] * y[j];
    for (int i = 0; i < n; i++) {
        T d = A[i * n + i];
        T ad = d < 0 ? -d : d;
        y[i] = (ad > tiny) ? y[i] / d : (T)0;
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = i + 1; j < n; j++) y[i] -= A[j * n + i] * y[j];
    for (int i = 0; i < n; i++) x[p[i]] = y[i];
}

// OdometryProvider::rodrigues, EF/Utils/OdometryProvider.h:35-71
__device__ void rodrigues_d(const double* src, double* R)
{
    double rx = src[0], ry = src[1], rz = src[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (theta >= DBL_EPSILON) {
        double c = cos(theta), s = sin(theta), c1 = 1.0 - c;
        double it = theta ? 1.0 / theta : 0.0;
        rx *= it; ry *= it; rz *= it;
        double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx_[k];
    }
}

__device__ void matmul_d(int n, const double* A, const double* B, double* C)
{
    double T[16];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * n + j];
            T[i * n + j] = s;
        }
    for (int i = 0; i < n * n; i++) C[i] = T[i];
}

__device__ void k_matrix_d(float fx, float fy, float cx, float cy, double* K, double* Kinv)
{
    for (int i = 0; i < 9; i++) K[i] = Kinv[i] = 0;
    K[0] = fx; K[4] = fy; K[2] = cx; K[5] = cy; K[8] = 1;
    Kinv[0] = 1.0 / K[0]; Kinv[4] = 1.0 / K[4];
    Kinv[2] = -K[2] / K[0]; Kinv[5] = -K[5] / K[4]; Kinv[8] = 1;
}

// writes krkinv / kt of the next residual pass from resultRt (EF/Utils/RGBDOdometry.cpp:424-434)
__device__ void set_warp_matrices(DevState* st, float fx, float fy, float cx, float cy)
{
    double K[9], Kinv[9], Rt[16];
    k_matrix_d(fx, fy, cx, cy, K, Kinv);
    const double* M = st->resultRt;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 4 + j] = M[j * 4 + i];
    for (int i = 0; i < 3; i++) Rt[i * 4 + 3] = -(Rt[i * 4 + 0] * M[3] + Rt[i * 4 + 1] * M[7] + Rt[i * 4 + 2] * M[11]);
    double R3[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
    double KR[9], KRK[9];
    matmul_d(3, K, R3, KR);
    matmul_d(3, KR, Kinv, KRK);
    for (int k = 0; k < 9; k++) st->krkinv[k] = (float)KRK[k];
    double tt[3] = {Rt[3], Rt[7], Rt[11]};
    for (int r = 0; r < 3; r++) st->kt[r] = (float)(K[r * 3] * tt[0] + K[r * 3 + 1] * tt[1] + K[r * 3 + 2] * tt[2]);
}

__device__ void set_so3_matrices(DevState* st, float fx, float fy, float cx, float cy)
{
    double K[9], Kinv[9], KR[9], H[9];
    k_matrix_d(fx, fy, cx, cy, K, Kinv);
    matmul_d(3, K, st->resultR, KR);
    matmul_d(3, KR, Kinv, H);
    for (int k = 0; k < 9; k++) { st->imageBasis[k] = (float)H[k]; st->kinv[k] = (float)Kinv[k]; st->krlr[k] = (float)KR[k]; }
}

// start of a tracker run: Rprev/tprev from the current pose, identity increments (:278-311, :388-403)
__global__ void k_track_begin(DevState* st, int so3, float fx2, float fy2, float cx2, float cy2)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) st->last_pose[k] = st->pose[k];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) st->Rprev[r * 3 + c] = st->Rcurr[r * 3 + c] = st->pose[r * 4 + c];
        st->tprev[r] = st->tcurr[r] = st->pose[r * 4 + 3];
    }
    {   // general 3x3 inverse, as Eigen's Matrix3f::inverse (:388)
        const float* m = st->Rprev;
        float* o = st->Rprev_inv;
        float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
        float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
        float id = 1.0f / det;
        o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
        o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    }
    for (int k = 0; k < 9; k++) { st->resultR[k] = st->lastResultR[k] = (k % 4 == 0) ? 1.0 : 0.0; st->R_lr[k] = (k % 4 == 0) ? 1.f : 0.f; }
    st->so3_lastError = FLT_MAX / 2; st->so3_lastCount = FLT_MAX / 2;
    st->so3_done = so3 ? 0 : 1;
    st->lastSO3Error = 0; st->lastSO3Count = 0; st->lastICPError = 0; st->lastICPCount = 0; st->lastRGBError = 0; st->lastRGBCount = 0;
    if (so3) set_so3_matrices(st, fx2, fy2, cx2, cy2);
}

// one SO(3) iteration's host logic, EF/Utils/RGBDOdometry.cpp:348-380, on one wave
__global__ void k_so3_update(DevState* st, const float* __restrict__ partials, int blocks, float fx2, float fy2, float cx2, float cy2)
{
    if (st->so3_done) return;
    __shared__ double sums[11];
    const int lane = threadIdx.x;
    for (int k = 0; k < 11; k++) {
        double v = 0;
        for (int b = lane; b < blocks; b += 64) v += (double)partials[b * 12 + k];
        v = wave_sum_d(v);
        if (lane == 0) sums[k] = v;
    }
    __syncthreads();
    if (lane != 0) return;
    float o[11];
    for (int k = 0; k < 11; k++) o[k] = (float)sums[k];
    float jtj[9], jtr[3];
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            float v = o[shift++];
            if (j == 3) jtr[i] = v; else jtj[j * 3 + i] = jtj[i * 3 + j] = v;
        }
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
    ldlt_solve<float>(3, jtj, jtr, delta, (float)(1.0 / FLT_MAX));
    double dd[3] = {delta[0], delta[1], delta[2]}, ru[9];
    rodrigues_d(dd, ru);
    float ruf[9], nr[9];
    for (int k = 0; k < 9; k++) ruf[k] = (float)ru[k];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) nr[i * 3 + j] = ruf[i * 3] * st->R_lr[j] + ruf[i * 3 + 1] * st->R_lr[3 + j] + ruf[i * 3 + 2] * st->R_lr[6 + j];
    for (int k = 0; k < 9; k++) { st->R_lr[k] = nr[k]; st->resultR[k] = nr[k]; }
    set_so3_matrices(st, fx2, fy2, cx2, cy2);
}

// after the SO(3) loop: seed resultRt and the first warp matrices (:392-403)
__global__ void k_gn_begin(DevState* st, int so3, float fx, float fy, float cx, float cy)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) st->resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (so3)
        for (int x = 0; x < 3; x++)
            for (int y = 0; y < 3; y++) st->resultRt[x * 4 + y] = st->resultR[x * 3 + y];
    set_warp_matrices(st, fx, fy, cx, cy);
}

// one Gauss-Newton iteration's host logic, EF/Utils/RGBDOdometry.cpp:461-583 (icp && rgb branch
// selected by the flags), on one block: fixed-order double sums of the block partials, 6x6 pivoted
// LDLT in double, SE(3) update, next warp matrices.
__global__ void k_gn_solve(DevState* st, const float* __restrict__ icp_partials, int icp_blocks, const float* __restrict__ rgb_partials, int rgb_blocks,
                           const int* __restrict__ res_partials, int res_blocks, int icp, int rgb, float icp_weight, float nfx, float nfy, float ncx, float ncy)
{
    __shared__ double s_icp[29], s_rgb[29];
    __shared__ int s_res[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = wid; k < 58; k += nw) {
        const float* src = k < 29 ? icp_partials : rgb_partials;
        int nb = k < 29 ? icp_blocks : rgb_blocks, kk = k < 29 ? k : k - 29;
        double v = 0;
        for (int b = lane; b < nb; b += 64) v += (double)src[b * 32 + kk];
        v = wave_sum_d(v);
        if (lane == 0) { if (k < 29) s_icp[kk] = v; else s_rgb[kk] = v; }
    }
    if (wid == 0) {
        for (int k = 0; k < 2; k++) {
            int v = 0;
            for (int b = lane; b < res_blocks; b += 64) v += res_partials[b * 2 + k];
            v = wave_sum_i(v);
            if (lane == 0) s_res[k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float oi[29], orr[29];
    for (int k = 0; k < 29; k++) { oi[k] = icp ? (float)s_icp[k] : 0.f; orr[k] = rgb ? (float)s_rgb[k] : 0.f; st->icp29[k] = oi[k]; st->rgb29[k] = orr[k]; }
    int rgbSize = rgb ? s_res[0] : 0, sigma = rgb ? s_res[1] : 0;
    st->rgb_count = rgbSize; st->rgb_sigma = sigma;
    st->lastRGBError = (float)(sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));
    st->lastRGBCount = (float)rgbSize;
    if (icp) { st->lastICPError = sqrtf(oi[27]) / oi[28]; st->lastICPCount = oi[28]; }
    float A_icp[36], b_icp[6], A_rgb[36], b_rgb[6];
    {
        int shift = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 7; ++j) {
                float vi = oi[shift], vr = orr[shift];
                shift++;
                if (j == 6) { b_icp[i] = vi; b_rgb[i] = vr; }
                else { A_icp[j * 6 + i] = A_icp[i * 6 + j] = vi; A_rgb[j * 6 + i] = A_rgb[i * 6 + j] = vr; }
            }
    }
    double* lastA = st->lastA;
    double* lastb = st->lastb;
    if (icp && rgb) {
        double wgt = icp_weight;
        for (int k = 0; k < 36; k++) lastA[k] = (double)A_rgb[k] + wgt * wgt * (double)A_icp[k];
        for (int k = 0; k < 6; k++) lastb[k] = (double)b_rgb[k] + wgt * (double)b_icp[k];
    } else if (icp) {
        for (int k = 0; k < 36; k++) lastA[k] = A_icp[k];
        for (int k = 0; k < 6; k++) lastb[k] = b_icp[k];
    } else {
        for (int k = 0; k < 36; k++) lastA[k] = A_rgb[k];
        for (int k = 0; k < 6; k++) lastb[k] = b_rgb[k];
    }
    double result[6];
    ldlt_solve<double>(6, lastA, lastb, result, 1.0 / DBL_MAX);
    // computeUpdateSE3, EF/Utils/OdometryProvider.h:73-93
    double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rr[9];
    rodrigues_d(&result[3], Rr);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) upd[r * 4 + c] = Rr[r * 3 + c];
    upd[3] = result[0]; upd[7] = result[1]; upd[11] = result[2];
    matmul_d(4, upd, st->resultRt, st->resultRt);
    float oR[9], ot[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) oR[r * 3 + c] = (float)st->resultRt[r * 4 + c];
        ot[r] = (float)st->resultRt[r * 4 + 3];
    }
    float iR[9], it[3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) iR[r * 3 + c] = oR[c * 3 + r];
    for (int r = 0; r < 3; r++) it[r] = -(iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1] + iR[r * 3 + 2] * ot[2]);
    const float* Rp = st->Rprev;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) st->Rcurr[r * 3 + c] = Rp[r * 3] * iR[c] + Rp[r * 3 + 1] * iR[3 + c] + Rp[r * 3 + 2] * iR[6 + c];
        st->tcurr[r] = Rp[r * 3] * it[0] + Rp[r * 3 + 1] * it[1] + Rp[r * 3 + 2] * it[2] + st->tprev[r];
    }
    set_warp_matrices(st, nfx, nfy, ncx, ncy);   // intrinsics of the level the NEXT iteration runs at
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

// end of the tracker run (:587-603) + pose write-back + velocity weighting (EF/ElasticFusion.cpp:425-449)
__global__ void k_track_end(DevState* st, int rgb, int tracked, float weight_mult)
{
    if (threadIdx.x != 0) return;
    if (tracked) {
        if (rgb) {
            v3 d = v3m(st->tcurr[0] - st->tprev[0], st->tcurr[1] - st->tprev[1], st->tcurr[2] - st->tprev[2]);
            if (norm(d) > 0.3f) {
                for (int k = 0; k < 9; k++) st->Rcurr[k] = st->Rprev[k];
                for (int k = 0; k < 3; k++) st->tcurr[k] = st->tprev[k];
            }
        }
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) st->pose[r * 4 + c] = st->Rcurr[r * 3 + c];
            st->pose[r * 4 + 3] = st->tcurr[r];
        }
        st->pose[12] = st->pose[13] = st->pose[14] = 0.f; st->pose[15] = 1.f;
    }
    pose_inverse(st->pose, st->pose_inv);
    float diff[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += st->pose_inv[r * 4 + k] * st->last_pose[k * 4 + c];
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
    st->weighting = fmaxf(1.0f - (weighting / largest), minWeight) * weight_mult;
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
        HIPCHK(h, hipMalloc(&p.depth_tmp[i], n * 2));
        HIPCHK(h, hipMalloc(&p.vmap_curr[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_curr[i], n * 12));
        HIPCHK(h, hipMalloc(&p.vmap_cam[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_cam[i], n * 12));
        HIPCHK(h, hipMalloc(&p.vmap_prev[i], n * 12)); HIPCHK(h, hipMalloc(&p.nmap_prev[i], n * 12));
        HIPCHK(h, hipMalloc(&p.last_depth[i], n * 4));
        HIPCHK(h, hipMalloc(&p.last_img[i], n)); HIPCHK(h, hipMalloc(&p.next_img[i], n)); HIPCHK(h, hipMalloc(&p.lastnext_img[i], n));
        HIPCHK(h, hipMemset(p.lastnext_img[i], 0, n)); HIPCHK(h, hipMemset(p.next_img[i], 0, n));
        HIPCHK(h, hipMalloc(&p.didx[i], n * 2)); HIPCHK(h, hipMalloc(&p.didy[i], n * 2));
        HIPCHK(h, hipMalloc(&p.cloud[i], n * 12));
        HIPCHK(h, hipMalloc(&p.corres[i], n * 8));
    }
    const int maxb = 1024;
    HIPCHK(h, hipMalloc(&h->icp_partials, maxb * 32 * 4));
    HIPCHK(h, hipMalloc(&h->rgb_partials, maxb * 32 * 4));
    HIPCHK(h, hipMalloc(&h->res_partials, maxb * 2 * 4));
    HIPCHK(h, hipMalloc(&h->so3_partials, maxb * 12 * 4));
    HIPCHK(h, hipMalloc(&h->d_out29, 64 * 4));
    return IFX_OK;
}

void ifx_free_tracker(ifx* h)
{
    Pyr& p = h->pyr;
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        hipFree(p.depth_tmp[i]); hipFree(p.vmap_curr[i]); hipFree(p.nmap_curr[i]); hipFree(p.vmap_cam[i]); hipFree(p.nmap_cam[i]);
        hipFree(p.vmap_prev[i]); hipFree(p.nmap_prev[i]); hipFree(p.last_depth[i]); hipFree(p.last_img[i]); hipFree(p.next_img[i]);
        hipFree(p.lastnext_img[i]); hipFree(p.didx[i]); hipFree(p.didy[i]); hipFree(p.cloud[i]); hipFree(p.corres[i]);
    }
    hipFree(h->icp_partials); hipFree(h->rgb_partials); hipFree(h->res_partials); hipFree(h->so3_partials); hipFree(h->d_out29);
}

static inline int red_blocks(ifx* h, int n)
{
    int b = cdiv(n, RED_THREADS * 4);
    if (b > h->opt_icp_blocks) b = h->opt_icp_blocks;
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
static void tracker_init_model(ifx* h, const float* pv, const float* pn, const uint8_t* pi, const float* fv, const float* fn, const uint8_t* fi)
{
    Pyr& p = h->pyr;
    LAUNCH(h, "model_level0", G2(h->w, h->h), B2, k_model_level0, h->d_state, pv, pn, pi, fv, fn, fi, h->w, h->h, p.vmap_cam[0], p.nmap_cam[0],
           p.last_depth[0], p.last_img[0], 6.0f);
    for (int i = 1; i < IFX_NUM_PYRS; i++) {
        LAUNCH(h, "resize_maps", G2(p.w[i], p.h[i]), B2, k_resize_maps, p.vmap_cam[i - 1], p.nmap_cam[i - 1], p.w[i - 1], p.h[i - 1], p.vmap_cam[i], p.nmap_cam[i]);
        LAUNCH(h, "pyrdown_f", G2(p.w[i], p.h[i]), B2, k_pyrdown_gauss_f, p.last_depth[i - 1], p.w[i - 1], p.h[i - 1], p.last_depth[i]);
        LAUNCH(h, "pyrdown_u8", G2(p.w[i], p.h[i]), B2, k_pyrdown_gauss_u8, p.last_img[i - 1], p.w[i - 1], p.h[i - 1], p.last_img[i]);
    }
    for (int i = 0; i < IFX_NUM_PYRS; i++)
        LAUNCH(h, "transform_maps", G2(p.w[i], p.h[i]), B2, k_transform_maps, h->d_state, p.vmap_cam[i], p.nmap_cam[i], p.w[i], p.h[i], p.vmap_prev[i], p.nmap_prev[i]);
}

// frame side: initICP(filteredDepth) + initRGB (EF/Utils/RGBDOdometry.cpp:118-142,243-247); nextDepth
// aliases lastDepth because initRGB re-reads the model's vmaps_tmp (reference behaviour).
static void tracker_init_frame(ifx* h, const uint16_t* depth_filt, const uint8_t* rgb)
{
    Pyr& p = h->pyr;
    hipMemcpyAsync(p.depth_tmp[0], depth_filt, (size_t)h->P * 2, hipMemcpyDeviceToDevice, h->stream);
    LAUNCH(h, "intensity", dim3(cdiv(h->P, 256)), dim3(256), k_intensity, rgb, 3, h->P, p.next_img[0]);
    for (int i = 1; i < IFX_NUM_PYRS; i++) {
        LAUNCH(h, "pyrdown_u16", G2(p.w[i], p.h[i]), B2, k_pyrdown_u16, p.depth_tmp[i - 1], p.w[i - 1], p.h[i - 1], p.depth_tmp[i]);
        LAUNCH(h, "pyrdown_u8", G2(p.w[i], p.h[i]), B2, k_pyrdown_gauss_u8, p.next_img[i - 1], p.w[i - 1], p.h[i - 1], p.next_img[i]);
    }
    for (int i = 0; i < IFX_NUM_PYRS; i++) {
        float div = (float)(1 << i);
        float fx = h->cfg.fx / div, fy = h->cfg.fy / div, cx = h->cfg.cx / div, cy = h->cfg.cy / div;
        LAUNCH(h, "vmap_nmap", G2(p.w[i], p.h[i]), B2, k_vmap_nmap, p.depth_tmp[i], p.w[i], p.h[i], 1.f / fx, 1.f / fy, cx, cy, h->cfg.max_depth_processed,
               p.vmap_curr[i], p.nmap_curr[i]);
        LAUNCH(h, "sobel", G2(p.w[i], p.h[i]), B2, k_sobel, p.next_img[i], p.w[i], p.h[i], p.didx[i], p.didy[i]);
    }
}

// getIncrementalTransformation, EF/Utils/RGBDOdometry.cpp:267-603, enqueued without any readback
static void tracker_run(ifx* h, float weight_mult)
{
    Pyr& p = h->pyr;
    const ifx_config& c = h->cfg;
    const int icp = c.icp_weight > 0, rgb = c.icp_weight < 100, so3 = c.so3;
    const float d2 = 4.f;
    LAUNCH(h, "track_begin", dim3(1), dim3(64), k_track_begin, h->d_state, so3, c.fx / d2, c.fy / d2, c.cx / d2, c.cy / d2);
    IcpArgs ia; ResArgs ra; So3Args sa;
    memset(&ia, 0, sizeof(ia)); memset(&ra, 0, sizeof(ra)); memset(&sa, 0, sizeof(sa));
    if (so3) {
        int L = 2, n = p.w[L] * p.h[L], nb = red_blocks(h, n);
        for (int it = 0; it < 10; it++) {
            LAUNCH(h, "so3", dim3(nb), dim3(RED_THREADS), k_so3, h->d_state, sa, p.lastnext_img[L], p.next_img[L], p.w[L], p.h[L], h->so3_partials);
            LAUNCH(h, "so3_update", dim3(1), dim3(64), k_so3_update, h->d_state, h->so3_partials, nb, c.fx / d2, c.fy / d2, c.cx / d2, c.cy / d2);
        }
    }
    int iterations[3] = {c.fast_odom ? 3 : 10, c.pyramid ? 5 : 0, c.pyramid ? 4 : 0};
    int first = -1;
    for (int i = IFX_NUM_PYRS - 1; i >= 0; i--) if (iterations[i] > 0) { first = i; break; }
    {
        float div = (float)(1 << (first < 0 ? 0 : first));
        LAUNCH(h, "gn_begin", dim3(1), dim3(64), k_gn_begin, h->d_state, so3, c.fx / div, c.fy / div, c.cx / div, c.cy / div);
    }
    static const float minGrad[3] = {5, 3, 1};
    const double sobelScale = 1.0 / 8.0;
    for (int i = IFX_NUM_PYRS - 1; i >= 0; i--) {
        float div = (float)(1 << i);
        float fx = c.fx / div, fy = c.fy / div, cx = c.cx / div, cy = c.cy / div;
        int lw = p.w[i], lh = p.h[i], n = lw * lh, nb = red_blocks(h, n);
        if (rgb && iterations[i] > 0)
            LAUNCH(h, "project_cloud", G2(lw, lh), B2, k_project_cloud, p.last_depth[i], lw, lh, 1.0f / fx, 1.0f / fy, cx, cy, p.cloud[i]);
        for (int j = 0; j < iterations[i]; j++) {
            // intrinsics of the level the next iteration runs at (for the warp matrices the solve emits)
            int nl = i;
            if (j == iterations[i] - 1) { nl = i - 1; while (nl >= 0 && iterations[nl] == 0) nl--; if (nl < 0) nl = 0; }
            float nd = (float)(1 << nl);
            if (rgb)
                LAUNCH(h, "rgb_residual", dim3(nb), dim3(RED_THREADS), k_rgb_residual, h->d_state, ra, (float)(pow(minGrad[i], 2.0) / pow(sobelScale, 2.0)), p.didx[i],
                       p.didy[i], p.last_depth[i], p.last_depth[i], p.last_img[i], p.next_img[i], (Corres8*)p.corres[i], 0.07f, lw, lh, h->res_partials);
            if (icp)
                LAUNCH(h, "icp", dim3(nb), dim3(RED_THREADS), k_icp, h->d_state, ia, p.vmap_curr[i], p.nmap_curr[i], p.vmap_prev[i], p.nmap_prev[i], fx, fy, cx, cy, 0.10f,
                       sinf(20.f * 3.14159254f / 180.f), lw, lh, h->icp_partials);
            if (rgb)
                LAUNCH(h, "rgb_step", dim3(nb), dim3(RED_THREADS), k_rgb_step, (const Corres8*)p.corres[i], 0.f, h->res_partials, nb, p.cloud[i], fx, fy, p.didx[i], p.didy[i],
                       (float)sobelScale, lw, lh, h->rgb_partials);
            LAUNCH(h, "gn_solve", dim3(1), dim3(256), k_gn_solve, h->d_state, h->icp_partials, nb, h->rgb_partials, nb, h->res_partials, nb, icp, rgb, c.icp_weight,
                   c.fx / nd, c.fy / nd, c.cx / nd, c.cy / nd);
        }
    }
    LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, h->d_state, rgb, 1, weight_mult);
    if (so3)
        for (int i = 0; i < IFX_NUM_PYRS; i++) std::swap(p.lastnext_img[i], p.next_img[i]);
}

int ifx_tracker_run_frame(ifx* h)
{
    tracker_init_model(h, h->pred_vertex, h->pred_normal, h->pred_image, h->fill_vertex, h->fill_normal, h->fill_image);
    tracker_init_frame(h, h->depth_filt, h->rgb);
    tracker_run(h, 1.0f);
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
    LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, h->d_state, 0, 0, weight_mult);
    return IFX_OK;
}
int ifx_tracker_set_weight(ifx* h, float weight_mult)
{
    // re-evaluates the velocity weighting with the caller's multiplier (weightMultiplier argument)
    LAUNCH(h, "track_end", dim3(1), dim3(64), k_track_end, h->d_state, 0, 0, weight_mult);
    return IFX_OK;
}

// ======================================================================= stage API (unit parity)

static int final_sum(ifx* h, const float* d_partials, int stride, int nv, int nb, float* out_host)
{
    std::vector<float> hp((size_t)nb * stride);
    HIPCHK(h, hipMemcpyAsync(hp.data(), d_partials, hp.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int b = 0; b < nb; b++) s += (double)hp[(size_t)b * stride + k];
        out_host[k] = (float)s;
    }
    return IFX_OK;
}

extern "C" int ifx_icp_step(ifx_t* h, const float* Rcurr9, const float* tcurr3, const float* d_vmap_curr, const float* d_nmap_curr, const float* Rprev_inv9,
                            const float* tprev3, float fx, float fy, float cx, float cy, const float* d_vmap_g_prev, const float* d_nmap_g_prev, float dist_thres,
                            float angle_thres, int w, int hgt, float* out29_host)
{
    if (!h || !out29_host) return IFX_E_INVALID;
    IcpArgs ia;
    memcpy(ia.Rcurr, Rcurr9, 36); memcpy(ia.tcurr, tcurr3, 12); memcpy(ia.Rprev_inv, Rprev_inv9, 36); memcpy(ia.tprev, tprev3, 12);
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "icp", dim3(nb), dim3(RED_THREADS), k_icp, (const DevState*)nullptr, ia, d_vmap_curr, d_nmap_curr, d_vmap_g_prev, d_nmap_g_prev, fx, fy, cx, cy, dist_thres,
           angle_thres, w, hgt, h->icp_partials);
    return final_sum(h, h->icp_partials, 32, 29, nb, out29_host);
}

extern "C" int ifx_rgb_residual(ifx_t* h, float min_scale, const int16_t* d_didx, const int16_t* d_didy, const float* d_last_depth, const float* d_next_depth,
                                const uint8_t* d_last_img, const uint8_t* d_next_img, void* d_corres8, float max_depth_delta, const float* kt3, const float* krkinv9,
                                int w, int hgt, int* count_host, int* sigma_host)
{
    if (!h) return IFX_E_INVALID;
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
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "rgb_step", dim3(nb), dim3(RED_THREADS), k_rgb_step, (const Corres8*)d_corres8, sigma, (const int*)nullptr, 0, d_cloud3, fx, fy, d_didx, d_didy, sobel_scale, w,
           hgt, h->rgb_partials);
    return final_sum(h, h->rgb_partials, 32, 29, nb, out29_host);
}

extern "C" int ifx_so3_step(ifx_t* h, const uint8_t* d_last_img, const uint8_t* d_next_img, const float* image_basis9, const float* kinv9, const float* krlr9, int w,
                            int hgt, float* out11_host)
{
    if (!h || !out11_host) return IFX_E_INVALID;
    So3Args sa;
    memcpy(sa.ib, image_basis9, 36); memcpy(sa.kinv, kinv9, 36); memcpy(sa.krlr, krlr9, 36);
    int nb = red_blocks(h, w * hgt);
    LAUNCH(h, "so3", dim3(nb), dim3(RED_THREADS), k_so3, (const DevState*)nullptr, sa, d_last_img, d_next_img, w, hgt, h->so3_partials);
    return final_sum(h, h->so3_partials, 12, 11, nb, out11_host);
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
    HIPCHK(h, hipMemcpyAsync(h->d_traj, pose16, 64, hipMemcpyHostToDevice, h->stream));   // scratch slot 0 of the log
    LAUNCH(h, "write_pose", dim3(1), dim3(64), k_write_pose, h->d_state, h->d_traj);
    LAUNCH(h, "set_dense", dim3(1), dim3(64), k_set_dense, h->d_state, 1);
    tracker_init_model(h, h->pred_vertex, h->pred_normal, h->pred_image, h->fill_vertex, h->fill_normal, h->fill_image);
    tracker_init_frame(h, h->depth_filt, h->rgb);
    tracker_run(h, 1.0f);
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

extern "C" int ifx_tracker_buffer_download(ifx_t* h, const char* name, int l, void* out, int64_t max_bytes)
{
    if (!h || !name || l < 0 || l >= IFX_NUM_PYRS) return IFX_E_INVALID;
    Pyr& p = h->pyr;
    size_t n = (size_t)p.w[l] * p.h[l];
    const void* src = nullptr;
    size_t bytes = 0;
    std::string s(name);
    if (s == "vmap_curr") { src = p.vmap_curr[l]; bytes = n * 12; }
    else if (s == "nmap_curr") { src = p.nmap_curr[l]; bytes = n * 12; }
    else if (s == "vmap_prev") { src = p.vmap_prev[l]; bytes = n * 12; }
    else if (s == "nmap_prev") { src = p.nmap_prev[l]; bytes = n * 12; }
    else if (s == "last_depth" || s == "next_depth") { src = p.last_depth[l]; bytes = n * 4; }
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
