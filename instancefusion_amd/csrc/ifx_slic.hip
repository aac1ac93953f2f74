// ifx_slic.hip -- superpixel refinement of the instance masks.
//
//   ifx_slic_segment            <- InstanceFusion::gSLICrInterface          IF/Core/InstanceFusion_superpixel.cpp:713-772
//                                  gSLICr seg_engine (5 iterations, XYZ)    IF/gSLICr/gSLICr_Lib/engines/gSLICr_seg_engine{.cpp,_GPU.cu,_shared.h}
//   ifx_merge_superpixels       <- InstanceFusion::mergeSuperPixel          IF/Core/InstanceFusion_superpixel.cpp:40-225
//                                  depth gaussian / pos / normal maps       IF/Core/InstanceFusionCuda.cu:141-300
//                                  getSuperPixelInfoCuda kernels 0, A-E     IF/Core/InstanceFusionCuda.cu:304-675
//                                  connectSuperPixel (host graph pass)      IF/Core/InstanceFusion_superpixel.cpp:227-400
//   ifx_mask_superpixel_filter  <- maskSuperPixelFilter_OverSeg             IF/Core/InstanceFusion_superpixel.cpp:651-710
//
// Layout: everything per pixel is a flat [P] array (float4 for xyz colour, position and normal so a
// pixel is one 16-B load); per-superpixel statistics are 10 x int64 fixed-point sums (2^-32 units):
// the reference's float atomicAdd order depends on scheduling, exact integer atomics do not, so the
// result is deterministic and identical to the CPU oracle.  Neighbour sets are a spn x spn bit matrix
// (atomicOr), read back in ascending order (first 11).  The cluster update of SLIC keeps the reference's
// summation order (16x16 blocks, stride-128..1 tree, 9 blocks in sequence) so labels match a CPU build
// of the reference's own per-pixel functions bit for bit.
#include "ifx_ctx.h"
#include "ifx_dev.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int SPX = 16;      // my_settings.spixel_size (IF/Core/InstanceFusion.cpp:446) == gSLICr BLOCK_DIM
constexpr int NB_MAX = 11;   // SPI_NP_MAX
constexpr int NSUM = 10;     // n, pos xyz, normal xyz, depth, dist^2, normal deviation
enum { SPI_SIZE = 30, SPI_PNUM = 0, SPI_POS_S = 1, SPI_NOR_S = 4, SPI_POS_A = 7, SPI_NOR_A = 10, SPI_DEPTH_SUM = 13, SPI_DEPTH_AVG = 14,
       SPI_DIST_DEV = 15, SPI_NOR_DEV = 16, SPI_CONNECT_N = 17, SPI_NP_FIRST = 18, SPI_FINAL = 29 };

struct SlicBuf {
    int P = 0, spn = 0, mw = 0, mh = 0, adj_words = 0;
    uint8_t* rgb = nullptr;
    uint16_t *depth = nullptr, *dg = nullptr;
    const uint8_t* cur_rgb = nullptr;      // the frame the stages read: b->rgb / b->depth after a copy, or the frame slot's own images (the resident frame of a call)
    const uint16_t* cur_depth = nullptr;
    float4 *xyz = nullptr, *pos = nullptr, *nor = nullptr;
    float4* ccol = nullptr;   // [spn] centre colour
    float2* cxy = nullptr;    // [spn] centre position
    int *seg = nullptr, *tmp = nullptr, *fin = nullptr;
    long long *sum1 = nullptr, *sum2 = nullptr;   // [spn][NSUM]
    unsigned int* adj = nullptr;                  // [spn][adj_words]
    float* info = nullptr;                        // [spn][30]
    int* final_of = nullptr;                      // [spn]
    int* num = nullptr; size_t num_cap = 0;       // [(nm+1)][spn]
    std::vector<float> h_info;
    std::vector<int> h_final;
};

// ---------------------------------------------------------------- SLIC
// rgb2xyz (gSLICr_seg_engine_shared.h:10-19); after gSLICrInterface's channel shuffle "_b" is channel 0
__global__ void k_slic_cvt(const uint8_t* __restrict__ rgb, float4* __restrict__ xyz, int P)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float b = (float)rgb[i * 3] * 0.0039216f, g = (float)rgb[i * 3 + 1] * 0.0039216f, r = (float)rgb[i * 3 + 2] * 0.0039216f;
    float4 o;
    o.x = r * 0.412453f + g * 0.357580f + b * 0.180423f;
    o.y = r * 0.212671f + g * 0.715160f + b * 0.072169f;
    o.z = r * 0.019334f + g * 0.119193f + b * 0.950227f;
    o.w = 0.f;
    xyz[i] = o;
}

// init_cluster_centers_shared :71-82
__global__ void k_slic_init(const float4* __restrict__ xyz, float4* ccol, float2* cxy, int mw, int mh, int w, int h)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mw * mh) return;
    int x = i % mw, y = i / mw;
    int ix = x * SPX + SPX / 2, iy = y * SPX + SPX / 2;
    ix = ix >= w ? (x * SPX + w) / 2 : ix;
    iy = iy >= h ? (y * SPX + h) / 2 : iy;
    cxy[i] = make_float2((float)ix, (float)iy);
    ccol[i] = xyz[iy * w + ix];
}

// find_center_association_shared :93-124.  One 16x16 block == one grid cell: its 9 candidate centres go to LDS.
__global__ void __launch_bounds__(256) k_slic_assoc(const float4* __restrict__ xyz, const float4* __restrict__ ccol, const float2* __restrict__ cxy, int* __restrict__ seg,
                                                    int mw, int mh, int w, int h, float weight, float nxy, float ncol)
{
    __shared__ float4 s_col[9];
    __shared__ float2 s_xy[9];
    __shared__ int s_id[9];
    int gx = blockIdx.x, gy = blockIdx.y, t = threadIdx.y * 16 + threadIdx.x;
    if (t < 9) {
        int qx = gx + (t % 3) - 1, qy = gy + (t / 3) - 1;
        int ok = qx >= 0 && qy >= 0 && qx < mw && qy < mh;
        s_id[t] = ok ? qy * mw + qx : -1;
        if (ok) { s_col[t] = ccol[qy * mw + qx]; s_xy[t] = cxy[qy * mw + qx]; }
    }
    __syncthreads();
    int x = gx * 16 + threadIdx.x, y = gy * 16 + threadIdx.y;
    if (x >= w || y >= h) return;
    float4 p = xyz[y * w + x];
    int minidx = -1;
    float dist = 999999.9999f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (s_id[k] < 0) continue;
        float4 c = s_col[k];
        float2 q = s_xy[k];
        float dcol = (p.x - c.x) * (p.x - c.x) + (p.y - c.y) * (p.y - c.y) + (p.z - c.z) * (p.z - c.z);
        float dxy = ((float)x - q.x) * ((float)x - q.x) + ((float)y - q.y) * ((float)y - q.y);
        float d = sqrtf(dcol * ncol + weight * dxy * nxy);
        if (d < dist) { dist = d; minidx = s_id[k]; }
    }
    if (minidx >= 0) seg[y * w + x] = minidx;
}

// Update_Cluster_Center_device + finalize_reduction_result_shared (gSLICr_seg_engine_GPU.cu:203-290,
// _shared.h:143-166): one workgroup per centre; the nine 16x16 window blocks are reduced by the same
// stride-128..1 tree and added in block order.
__global__ void __launch_bounds__(256) k_slic_update(const float4* __restrict__ xyz, const int* __restrict__ seg, float4* ccol, float2* cxy, int mw, int w, int h)
{
    __shared__ float s[6][256];
    int id = blockIdx.x, gx = id % mw, gy = id / mw, t = threadIdx.x;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int n = 0;
    for (int z = 0; z < 9; z++) {
        int xi = gx * SPX - SPX + (z % 3) * 16 + (t & 15), yi = gy * SPX - SPX + (z / 3) * 16 + (t >> 4);
        bool hit = xi >= 0 && xi < w && yi >= 0 && yi < h && seg[yi * w + xi] == id;
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int c = hit ? 1 : 0;
        if (hit) {
            float4 p = xyz[yi * w + xi];
            v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = (float)xi; v[4] = (float)yi;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; k++) s[k][t] = v[k];
        s[5][t] = __int_as_float(c);
        __syncthreads();
        if (t < 128) {
#pragma unroll
            for (int k = 0; k < 5; k++) { v[k] += s[k][t + 128]; s[k][t] = v[k]; }
            c += __float_as_int(s[5][t + 128]); s[5][t] = __int_as_float(c);
        }
        __syncthreads();
        if (t < 64) {
#pragma unroll
            for (int k = 0; k < 5; k++) v[k] += s[k][t + 64];
            c += __float_as_int(s[5][t + 64]);
#pragma unroll
            for (int st = 32; st >= 1; st >>= 1) {
#pragma unroll
                for (int k = 0; k < 5; k++) v[k] += __shfl_down(v[k], st);
                c += __shfl_down(c, st);
            }
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < 5; k++) acc[k] += v[k];
                n += c;
            }
        }
    }
    if (t == 0) {
        if (n != 0) {
            cxy[id] = make_float2(acc[3] / (float)n, acc[4] / (float)n);
            ccol[id] = make_float4(acc[0] / (float)n, acc[1] / (float)n, acc[2] / (float)n, 0.f);
        } else {
            cxy[id] = make_float2(0.f, 0.f);
            ccol[id] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// supress_local_lable :168-195
__global__ void k_slic_enforce(const int* __restrict__ in, int* __restrict__ out, int w, int h)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    int cl = in[y * w + x];
    if (x <= 1 || y <= 1 || x >= w - 2 || y >= h - 2) { out[y * w + x] = cl; return; }
    int diff = 0, dl = -1;
#pragma unroll
    for (int j = -2; j <= 2; j++)
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            int nl = in[(y + j) * w + x + i];
            if (nl != cl) { dl = nl; diff++; }
        }
    out[y * w + x] = diff >= 16 ? dl : cl;
}

// ---------------------------------------------------------------- merge
__device__ __forceinline__ long long sp_fx(float v)
{
    if (!(fabsf(v) < 1.0e6f)) return 0;
    return __double2ll_rn((double)v * 4294967296.0);
}
__device__ __host__ __forceinline__ float sp_unfx(long long s) { return (float)((double)s * (1.0 / 4294967296.0)); }

// checkNeighbours (InstanceFusionCuda.cu:41-61): the centre itself is not tested
__device__ __forceinline__ bool check_nb(const uint16_t* __restrict__ m, int x, int y, int w, int h)
{
    if (x + 1 >= w || x - 1 < 0 || y + 1 >= h || y - 1 < 0) return false;
    return m[y * w + x + 1] && m[y * w + x - 1] && m[(y + 1) * w + x] && m[(y - 1) * w + x] && m[(y + 1) * w + x + 1] && m[(y + 1) * w + x - 1] && m[(y - 1) * w + x + 1] &&
           m[(y - 1) * w + x - 1];
}

// depthMapGaussianfilterKernel :141-168
__global__ void k_sp_gauss(const uint16_t* __restrict__ d, uint16_t* __restrict__ dg, int w, int h)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    uint16_t out = 0;
    if (check_nb(d, x, y, w, h)) {
        int sum = 0, n = 0;
#pragma unroll
        for (int j = -1; j <= 1; j++)
#pragma unroll
            for (int i = -1; i <= 1; i++) {
                int wt = (j == 0 ? 2 : 1) * (i == 0 ? 2 : 1);
                int v = d[(y + j) * w + x + i];
                if (v) { n += wt; sum += wt * v; }
            }
        if (n) out = (uint16_t)(sum / n);
    }
    dg[y * w + x] = out;
}

struct V3 { float x, y, z; };
// getVertex :180-187
__device__ __forceinline__ V3 sp_vertex(const uint16_t* __restrict__ d, int x, int y, int w, float4 cam)
{
    float z = (float)d[y * w + x] / 1186.0f;
    V3 v;
    v.x = ((float)x - cam.x) * z * cam.z;
    v.y = ((float)y - cam.y) * z * cam.w;
    v.z = z;
    return v;
}
__device__ __forceinline__ V3 sp_cross(V3 l, V3 r, V3 u, V3 dn)
{
    V3 dx = {l.x - r.x, l.y - r.y, l.z - r.z}, dy = {u.x - dn.x, u.y - dn.y, u.z - dn.z}, a;
    a.x = dx.y * dy.z - dx.z * dy.y;
    a.y = dx.z * dy.x - dx.x * dy.z;
    a.z = dx.x * dy.y - dx.y * dy.x;
    return a;
}

// getPosMapFromDepthKernel :250-263, getNormalMapFromDepthKernel + getNormal :205-248,276-290 and kernel A :320-340
__global__ void k_sp_posnor(const uint16_t* __restrict__ dg, float4 cam, int w, int h, int spn, float4* __restrict__ pos, float4* __restrict__ nor, int* __restrict__ seg)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    int k = y * w + x;
    V3 p = {0.f, 0.f, 0.f}, nn = {0.f, 0.f, 0.f};
    if (dg[k]) p = sp_vertex(dg, x, y, w, cam);
    if (check_nb(dg, x, y, w, h)) {
        V3 c = sp_vertex(dg, x, y, w, cam), xf = sp_vertex(dg, x + 1, y, w, cam), xb = sp_vertex(dg, x - 1, y, w, cam), yf = sp_vertex(dg, x, y + 1, w, cam),
           yb = sp_vertex(dg, x, y - 1, w, cam);
        V3 t = sp_cross(xb, xf, yb, yf), s;
        s.x = t.x * 4; s.y = t.y * 4; s.z = t.z * 4;
        t = sp_cross(xb, c, yb, c);  s.x += t.x * 2; s.y += t.y * 2; s.z += t.z * 2;
        t = sp_cross(c, xf, yb, c);  s.x += t.x * 2; s.y += t.y * 2; s.z += t.z * 2;
        t = sp_cross(xb, c, c, yf);  s.x += t.x * 2; s.y += t.y * 2; s.z += t.z * 2;
        t = sp_cross(c, xf, c, yf);  s.x += t.x * 2; s.y += t.y * 2; s.z += t.z * 2;
        float len = sqrtf(s.x * s.x + s.y * s.y + s.z * s.z);
        nn.x = s.x / len; nn.y = s.y / len; nn.z = s.z / len;
    }
    pos[k] = make_float4(p.x, p.y, p.z, 0.f);
    nor[k] = make_float4(nn.x, nn.y, nn.z, 0.f);
    float t = 0.f;
    t += p.x * p.x; t += p.y * p.y; t += p.z * p.z;
    t += nn.x * nn.x; t += nn.y * nn.y; t += nn.z * nn.z;
    int id = seg[k];
    if ((double)t < 0.01 || id >= spn || id < 0) seg[k] = -1;
}

__device__ __forceinline__ long long wave_sum_ll(long long x)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// Adds v[0..N) of every lane to sums[id][off..off+N).  The lanes of a wave cover a 16x4 pixel patch, i.e. one
// to three superpixels: lanes are grouped by id, each group is reduced in registers and its leader issues the
// N atomics (integer sums: any grouping gives the same result).  Every lane of the wave must call this.
template <int N>
__device__ __forceinline__ void group_add(long long* sums, int id, const long long (&v)[N], int off)
{
    const int lane = __lane_id();
    unsigned long long todo = __ballot(id >= 0);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int lid = __shfl(id, leader);
        bool mine = id == lid;
        todo &= ~__ballot(mine);
#pragma unroll
        for (int k = 0; k < N; k++) {
            long long x = wave_sum_ll(mine ? v[k] : 0ll);
            if (lane == leader) atomicAdd((unsigned long long*)(sums + (size_t)lid * NSUM + off + k), (unsigned long long)x);
        }
    }
}

// kernel B :342-400: first sums + neighbour relation.  Block = 16x16 pixels, wave = 16x4.
__global__ void __launch_bounds__(256) k_sp_sums(const int* __restrict__ seg, const uint16_t* __restrict__ dg, const float4* __restrict__ pos, const float4* __restrict__ nor, int w, int h,
                                                 long long* sum1, unsigned int* adj, int adj_words)
{
    int x = blockIdx.x * 16 + threadIdx.x, y = blockIdx.y * 16 + threadIdx.y;
    bool inside = x < w && y < h;
    int k = inside ? y * w + x : 0, id = inside ? seg[k] : -1;
    long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (id >= 0) {
        float4 p = pos[k], n = nor[k];
        v[0] = 1; v[1] = sp_fx(p.x); v[2] = sp_fx(p.y); v[3] = sp_fx(p.z); v[4] = sp_fx(n.x); v[5] = sp_fx(n.y); v[6] = sp_fx(n.z); v[7] = dg[k];
    }
    group_add<8>(sum1, id, v, 0);
    if (id < 0 || x == 0 || x == w - 1 || y == 0 || y == h - 1) return;
    const int nb[4] = {k + w, k - w, k + 1, k - 1};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int q = seg[nb[i]];
        if (q >= 0 && q != id) {
            unsigned int* a = adj + (size_t)id * adj_words + (q >> 5);
            unsigned int bit = 1u << (q & 31);
            if (!(__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(a, bit);
        }
    }
}

__device__ __forceinline__ void sp_averages(const long long* s, float* I)
{
    int t = (int)s[0];
    float ps[3] = {sp_unfx(s[1]), sp_unfx(s[2]), sp_unfx(s[3])}, ns[3] = {sp_unfx(s[4]), sp_unfx(s[5]), sp_unfx(s[6])};
    float ds = (float)s[7];
    I[SPI_PNUM] = (float)t;
#pragma unroll
    for (int k = 0; k < 3; k++) { I[SPI_POS_S + k] = ps[k]; I[SPI_NOR_S + k] = ns[k]; }
    I[SPI_DEPTH_SUM] = ds;
    if (t != 0) {
        float len = sqrtf(ns[0] * ns[0] + ns[1] * ns[1] + ns[2] * ns[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) { I[SPI_POS_A + k] = ps[k] / (float)t; I[SPI_NOR_A + k] = ns[k] / len; }
        I[SPI_DEPTH_AVG] = ds / (float)t;
    }
}

// kernel 0 + neighbour lists (ascending, first 11) + kernel C :304-318, :492-516
__global__ void k_sp_first_avg(const long long* __restrict__ sum1, const unsigned int* __restrict__ adj, int adj_words, int spn, float* __restrict__ info)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= spn) return;
    float* I = info + (size_t)i * SPI_SIZE;
    for (int k = 0; k < SPI_SIZE; k++) I[k] = 0.f;
    I[SPI_CONNECT_N] = (float)NB_MAX;
    int c = 0;
    for (int wd = 0; wd < adj_words; wd++) {
        unsigned int m = adj[(size_t)i * adj_words + wd];
        while (m && c < NB_MAX) {
            int b = __ffs(m) - 1;
            m &= m - 1;
            I[SPI_NP_FIRST + c++] = (float)(wd * 32 + b);
        }
    }
    for (; c < NB_MAX; c++) I[SPI_NP_FIRST + c] = -1.f;
    sp_averages(sum1 + (size_t)i * NSUM, I);
}

// kernel D :518-610: every pixel re-clusters to the nearest of its superpixel and that superpixel's neighbours
__global__ void __launch_bounds__(256) k_sp_recluster(int* __restrict__ seg, const uint16_t* __restrict__ dg, const float4* __restrict__ pos, const float4* __restrict__ nor, int w, int h,
                                                      const float* __restrict__ info, long long* sum2)
{
    int x = blockIdx.x * 16 + threadIdx.x, y = blockIdx.y * 16 + threadIdx.y;
    bool inside = x < w && y < h;
    int k = inside ? y * w + x : 0, id = inside ? seg[k] : -1;
    int minID = -1;
    long long v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (id >= 0) {
        float4 p = pos[k], n = nor[k];
        float minDist = 999999.9f, minNor = 999999.9f;
        minID = id;
        for (int i = 0; i <= NB_MAX; i++) {
            int it = i == NB_MAX ? id : (int)info[(size_t)id * SPI_SIZE + SPI_NP_FIRST + i];
            if (it < 0) continue;
            const float* T = info + (size_t)it * SPI_SIZE;
            float va0 = T[SPI_NOR_A], va1 = T[SPI_NOR_A + 1], va2 = T[SPI_NOR_A + 2];
            float vb0 = T[SPI_POS_A] - p.x, vb1 = T[SPI_POS_A + 1] - p.y, vb2 = T[SPI_POS_A + 2] - p.z;
            float lenA = sqrtf(va0 * va0 + va1 * va1 + va2 * va2);
            float lenB = sqrtf(vb0 * vb0 + vb1 * vb1 + vb2 * vb2);
            float dot = va0 * vb0 + va1 * vb1 + va2 * vb2;
            float dist = (float)((double)fabsf(dot / lenA) + 1.0 * (double)lenB);
            float d1 = fabsf(va0 - n.x), d2 = fabsf(va1 - n.y), d3 = fabsf(va2 - n.z);
            float dn = d1 * d1 + d2 * d2 + d3 * d3;
            if (dist < minDist) { minNor = dn; minDist = dist; minID = it; }
        }
        float thr = (float)((0.026 * (double)info[(size_t)minID * SPI_SIZE + SPI_DEPTH_AVG] - (double)4.0f) / (double)1186.0f);
        if (minDist > 2 * thr) minID = -1;
        seg[k] = minID;
        if (minID != -1) {
            v[0] = 1; v[1] = sp_fx(p.x); v[2] = sp_fx(p.y); v[3] = sp_fx(p.z); v[4] = sp_fx(n.x); v[5] = sp_fx(n.y); v[6] = sp_fx(n.z); v[7] = dg[k];
            v[8] = sp_fx(minDist * minDist); v[9] = sp_fx(minNor);
        }
    }
    group_add<10>(sum2, minID, v, 0);
}

// kernel E :612-640 (a superpixel left without pixels keeps its first-pass averages)
__global__ void k_sp_second_avg(const long long* __restrict__ sum2, int spn, float* __restrict__ info)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= spn) return;
    float* I = info + (size_t)i * SPI_SIZE;
    const long long* s = sum2 + (size_t)i * NSUM;
    sp_averages(s, I);
    float dd = sp_unfx(s[8]), nd = sp_unfx(s[9]);
    int t = (int)s[0];
    if (t != 0) { dd = sqrtf(dd / (float)t); nd = sqrtf(nd / (float)t); }
    I[SPI_DIST_DEV] = dd;
    I[SPI_NOR_DEV] = nd;
}

// getFinalSuperPiexlKernel :690-705
__global__ void k_sp_final(const int* __restrict__ seg, const int* __restrict__ final_of, int* __restrict__ fin, int P)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    int id = seg[k];
    fin[k] = id < 0 ? id : final_of[id];
}

// maskSuperPixelFilter_OverSeg, first loop: pixels per region and per (mask, region).  Counted per KEY first
// (key = re-clustered superpixel in the fused path: <= a few waves share a counter, whereas a merged region can
// span most of the image), lanes grouped by key like group_add; k_sp_count_regions then folds keys into regions.
__global__ void __launch_bounds__(256) k_sp_count(const int* __restrict__ key, const uint8_t* __restrict__ masks, int nm, int w, int h, int spn, int* __restrict__ num)
{
    int x = blockIdx.x * 16 + threadIdx.x, y = blockIdx.y * 16 + threadIdx.y;
    bool inside = x < w && y < h;
    const size_t P = (size_t)w * h;
    int k = inside ? y * w + x : 0, id = inside ? key[k] : -1;
    if (id >= spn) id = -1;
    const int lane = __lane_id();
    for (int c = 0; c < nm; c += 32) {
        unsigned int bits = 0;
        if (id >= 0)
            for (int i = c; i < min(nm, c + 32); i++) bits |= (masks[(size_t)i * P + k] ? 1u : 0u) << (i - c);
        unsigned long long todo = __ballot(id >= 0);
        while (todo) {
            int leader = __ffsll((long long)todo) - 1;
            int lid = __shfl(id, leader);
            bool mine = id == lid;
            unsigned long long grp = __ballot(mine);
            todo &= ~grp;
            if (c == 0 && lane == leader) atomicAdd(&num[(size_t)nm * spn + lid], __popcll(grp));
            for (int i = c; i < min(nm, c + 32); i++) {
                unsigned long long b = __ballot(mine && ((bits >> (i - c)) & 1u));
                if (lane == leader && b) atomicAdd(&num[(size_t)i * spn + lid], __popcll(b));
            }
        }
    }
}

// one workgroup per counter row (mask i, or the totals): key counts -> region counts through final_of
__global__ void __launch_bounds__(256) k_sp_count_regions(const int* __restrict__ num_key, const int* __restrict__ final_of, int spn, int* __restrict__ num)
{
    extern __shared__ int s_cnt[];
    const int row = blockIdx.x;
    for (int s = threadIdx.x; s < spn; s += 256) s_cnt[s] = 0;
    __syncthreads();
    for (int s = threadIdx.x; s < spn; s += 256) {
        int c = num_key[(size_t)row * spn + s];
        if (c) atomicAdd(&s_cnt[final_of[s]], c);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < spn; s += 256) num[(size_t)row * spn + s] = s_cnt[s];
}

// second loop: a mask keeps a region when it covers more than 75 % of it
__global__ void k_sp_filter(const int* __restrict__ fin, uint8_t* __restrict__ masks, int nm, int P, int spn, const int* __restrict__ num)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    int id = fin[k];
    bool valid = id >= 0 && id < spn;
    int n = valid ? num[(size_t)nm * spn + id] : 1;
    for (int i = 0; i < nm; i++) {
        uint8_t o = 0;
        if (valid) {
            float test = (float)num[(size_t)i * spn + id] * 1.0f / (float)n;
            o = (double)test > 0.75 ? 255 : 0;
        }
        masks[(size_t)i * P + k] = o;
    }
}

// connectSuperPixel (IF/Core/InstanceFusion_superpixel.cpp:227-400), host: ~1200 nodes with <= 11 edges each.
// First pass drops the edges whose plane-distance + normal term exceeds either end's threshold, second
// pass labels the connected components by their lowest member.
//
// On the device in ONE block (the reference does both passes on the host, and so did round 2: a read-back, ~13 k sequential edge tests and an upload in the
// middle of every segmentation call).  The reference's two sequential passes have order-free equivalents:
//  * pass 1 visits a = 0, 1, ... and, when the directed test of (a -> b) fails, deletes b from a's list AND a from b's list.  An entry (a -> b) therefore
//    survives iff test(a -> b) passes and (a is not in b's list or test(b -> a) passes): every thread evaluates its own node's directed tests, a barrier, then
//    the symmetric look-up -- on the ORIGINAL lists, which is what the sequential pass reads too (a deleted entry is never re-tested).
//  * pass 2 labels, for a = 0, 1, ..., everything reachable from a through still unlabelled nodes with a.  By induction on a this is
//    label(t) = min { a : t reachable from a along surviving DIRECTED entries } (a node on a path from the minimal such a to t cannot have been labelled
//    earlier, or t would be reachable from something smaller): min-propagation along the entries with pointer jumping, to the fixpoint.
__device__ __forceinline__ bool sp_edge_fails(const float* __restrict__ A, const float* __restrict__ B)
{
    const float nx = A[SPI_NOR_A], ny = A[SPI_NOR_A + 1], nz = A[SPI_NOR_A + 2];
    const float ex = A[SPI_POS_A] - B[SPI_POS_A], ey = A[SPI_POS_A + 1] - B[SPI_POS_A + 1], ez = A[SPI_POS_A + 2] - B[SPI_POS_A + 2];
    const float ln = sqrtf(nx * nx + ny * ny + nz * nz), le = sqrtf(ex * ex + ey * ey + ez * ez);
    const float dot = nx * ex + ny * ey + nz * ez;
    const float dist_term = (float)((double)fabsf(dot / ln) + 1.0 * (double)le);
    const float thrA = (float)(1 * ((0.026 * (double)A[SPI_DEPTH_AVG] - (double)4.0f) / (double)1186.0f));
    const float thrB = (float)(1 * ((0.026 * (double)B[SPI_DEPTH_AVG] - (double)4.0f) / (double)1186.0f));
    const float devA = 2 * A[SPI_DIST_DEV], devB = 2 * B[SPI_DIST_DEV];
    const float q1 = fabsf(nx - B[SPI_NOR_A]), q2 = fabsf(ny - B[SPI_NOR_A + 1]), q3 = fabsf(nz - B[SPI_NOR_A + 2]);
    const float nor_term = (float)(0.1 * (double)sqrtf(q1 * q1 + q2 * q2 + q3 * q3));
    const float zA = 0 * A[SPI_NOR_DEV], zB = 0 * B[SPI_NOR_DEV];
    const float test = dist_term + nor_term, limA = thrA + devA + zA, limB = thrB + devB + zB;
    return test > limA || test > limB;
}
// The directed tests of all entries, one thread per (node, list position), over the whole chip: ~13 k tests of twenty table floats and double arithmetic each were
// the long part of the one-block pass (78 us).  etab[a * NB_MAX + j] = neighbour | 0x40000000 when the test a -> neighbour passes; -1: no entry.
__global__ void k_sp_edges(int spn, const float* __restrict__ info, int* __restrict__ etab)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= spn * NB_MAX) return;
    const int a = t / NB_MAX, j = t - a * NB_MAX;
    const float* A = info + (size_t)a * SPI_SIZE;
    int e = -1;
    if (j < (int)A[SPI_CONNECT_N]) {
        const int b = (int)A[SPI_NP_FIRST + j];
        e = b;
        if (b != -1 && !sp_edge_fails(A, info + (size_t)b * SPI_SIZE)) e |= 0x40000000;
    }
    etab[t] = e;
}
// (first version: every round of the propagation re-read the lists from the table in global memory, 30-float records, one dependent load after the other:
// 150 us for 1200 nodes.  The lists now live in LDS as ints for the whole kernel, filled from k_sp_edges' table.)
__global__ __launch_bounds__(1024) void k_sp_connect(int spn, float* __restrict__ info, int* __restrict__ final_of, const int* __restrict__ etab)
{
    extern __shared__ int sm[];
    int* label = sm;                                   // [spn]
    unsigned int* pass = (unsigned int*)(sm + spn);    // [spn] bit j: the directed test of entry j passed
    int* nbr = sm + 2 * spn;                           // [spn][NB_MAX] the lists (-1: no entry / deleted)
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int t = tid; t < spn * NB_MAX; t += nt) nbr[t] = etab[t];   // (entry | pass flag; split below)
    __syncthreads();
    for (int a = tid; a < spn; a += nt) {
        unsigned int m = 0;
        for (int j = 0; j < NB_MAX; j++) {
            const int e = nbr[a * NB_MAX + j];
            if (e < 0) continue;   // (-1: no entry; a stored -1.f entry comes through as -1 too)
            if (e & 0x40000000) m |= 1u << j;
            nbr[a * NB_MAX + j] = e & 0x3FFFFFFF;
        }
        pass[a] = m;
        label[a] = a;
    }
    __syncthreads();
    unsigned int my_alive[2] = {0u, 0u};               // spn <= 2 * 1024 (ifx_slic.hip sizes the grid of superpixels from the image; checked by the launcher)
    for (int a = tid, q = 0; a < spn; a += nt, q++) {
        unsigned int m = pass[a];
        for (int j = 0; j < NB_MAX; j++) {
            if (!(m & (1u << j))) continue;
            const int b = nbr[a * NB_MAX + j];
            for (int k = 0; k < NB_MAX; k++)
                if (nbr[b * NB_MAX + k] == a) { if (!(pass[b] & (1u << k))) m &= ~(1u << j); break; }   // (the first entry equal to a, as the sequential deletion)
        }
        my_alive[q] = m;
    }
    __syncthreads();   // every original list has been read: the deletions may be written now
    for (int a = tid, q = 0; a < spn; a += nt, q++) {
        float* A = info + (size_t)a * SPI_SIZE;
        const unsigned int m = my_alive[q];
        for (int j = 0; j < NB_MAX; j++)
            if (!(m & (1u << j)) && nbr[a * NB_MAX + j] != -1) { nbr[a * NB_MAX + j] = -1; A[SPI_NP_FIRST + j] = -1.f; }
    }
    __syncthreads();
    // Nodes joined by entries in BOTH directions reach each other, so they end with one label: those classes first, by union-find on the label array (the larger
    // root is hung under the smaller, atomicMin, retried until it sticks), then every node points at its class's smallest member.  The directed propagation below
    // starts from that valid state ("my label is a node that reaches me") and has only the one-way entries left to honour: 2-3 rounds where the plain propagation
    // needed one per step of the longest chain (65 -> ~25 us for 1200 nodes).
    for (int a = tid; a < spn; a += nt) {
        for (int j = 0; j < NB_MAX; j++) {
            const int b = nbr[a * NB_MAX + j];
            if (b < 0 || b < a) continue;                       // (each mutual pair once, from its smaller end)
            bool back = false;
            for (int k = 0; k < NB_MAX; k++) back = back || nbr[b * NB_MAX + k] == a;
            if (!back) continue;
            int ra = a, rb = b;
            for (;;) {
                { int p_ = ((volatile int*)label)[ra]; while (p_ != ra) { ra = p_; p_ = ((volatile int*)label)[ra]; } }
                { int p_ = ((volatile int*)label)[rb]; while (p_ != rb) { rb = p_; p_ = ((volatile int*)label)[rb]; } }
                if (ra == rb) break;
                if (ra > rb) { const int s_ = ra; ra = rb; rb = s_; }
                const int old = atomicMin(&label[rb], ra);
                if (old == rb) break;
                rb = old;
            }
        }
    }
    __syncthreads();
    for (int a = tid; a < spn; a += nt) {
        int r = a, p_ = ((volatile int*)label)[r];
        while (p_ != r) { r = p_; p_ = ((volatile int*)label)[r]; }
        if (r != a) label[a] = r;   // (a concurrent reader walking through a sees either pointer: both lead to r)
    }
    __syncthreads();
    for (int round = 0; round < 4 * 1024; round++) {   // (the bound only guards against a corrupt table)
        int any = 0;
        for (int a = tid; a < spn; a += nt) {
            const int la = label[a];
            for (int j = 0; j < NB_MAX; j++) {
                const int b = nbr[a * NB_MAX + j];
                if (b >= 0 && atomicMin(&label[b], la) > la) any = 1;
            }
        }
        __syncthreads();
        for (int a = tid; a < spn; a += nt) {           // pointer jump: my label's label reaches me too
            const int l = label[a], g = label[l];
            if (g < l) { atomicMin(&label[a], g); any = 1; }
        }
        if (!__syncthreads_or(any)) break;
    }
    for (int a = tid; a < spn; a += nt) {
        info[(size_t)a * SPI_SIZE + SPI_FINAL] = (float)label[a];
        final_of[a] = label[a];
    }
}

// the same pass for images with more superpixels than the lists-in-LDS version holds (1280 x 960: 4800): labels and flags in LDS, the lists read from the table
__global__ __launch_bounds__(1024) void k_sp_connect_big(int spn, float* __restrict__ info, int* __restrict__ final_of)
{
    extern __shared__ int sm[];
    int* label = sm;                                   // [spn]
    unsigned int* pass = (unsigned int*)(sm + spn);    // [spn] bit j: the directed test of entry j passed
    unsigned int* alive = pass + spn;                  // [spn] bit j: entry j survives pass 1
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int a = tid; a < spn; a += nt) {
        const float* A = info + (size_t)a * SPI_SIZE;
        const int na = (int)A[SPI_CONNECT_N];
        unsigned int m = 0;
        for (int j = 0; j < na; j++) {
            const int b = (int)A[SPI_NP_FIRST + j];
            if (b == -1) continue;
            if (!sp_edge_fails(A, info + (size_t)b * SPI_SIZE)) m |= 1u << j;
        }
        pass[a] = m;
        label[a] = a;
    }
    __syncthreads();
    for (int a = tid; a < spn; a += nt) {
        const float* A = info + (size_t)a * SPI_SIZE;
        const int na = (int)A[SPI_CONNECT_N];
        unsigned int m = pass[a];
        for (int j = 0; j < na; j++) {
            if (!(m & (1u << j))) continue;
            const int b = (int)A[SPI_NP_FIRST + j];
            const float* B = info + (size_t)b * SPI_SIZE;
            const int nb = (int)B[SPI_CONNECT_N];
            for (int k = 0; k < nb; k++)
                if (B[SPI_NP_FIRST + k] == (float)a) { if (!(pass[b] & (1u << k))) m &= ~(1u << j); break; }   // (the first entry equal to a, as the sequential deletion)
        }
        alive[a] = m;
    }
    __syncthreads();   // every original list has been read: the deletions may be written now
    for (int a = tid; a < spn; a += nt) {
        float* A = info + (size_t)a * SPI_SIZE;
        const int na = (int)A[SPI_CONNECT_N];
        const unsigned int m = alive[a];
        for (int j = 0; j < na; j++)
            if (!(m & (1u << j)) && A[SPI_NP_FIRST + j] != -1.f) A[SPI_NP_FIRST + j] = -1.f;
    }
    __syncthreads();
    for (int round = 0; round < 4 * 1024; round++) {   // (the fixpoint arrives within ~log(diameter) rounds; the bound only guards against a corrupt table)
        int any = 0;
        for (int a = tid; a < spn; a += nt) {
            const float* A = info + (size_t)a * SPI_SIZE;
            const int na = (int)A[SPI_CONNECT_N], la = label[a];
            const unsigned int m = alive[a];
            for (int j = 0; j < na; j++) {
                if (!(m & (1u << j))) continue;
                const int b = (int)A[SPI_NP_FIRST + j];
                if (atomicMin(&label[b], la) > la) any = 1;
            }
        }
        __syncthreads();
        for (int a = tid; a < spn; a += nt) {           // pointer jump: my label's label reaches me too
            const int l = label[a], g = label[l];
            if (g < l) { atomicMin(&label[a], g); any = 1; }
        }
        if (!__syncthreads_or(any)) break;
    }
    for (int a = tid; a < spn; a += nt) {
        info[(size_t)a * SPI_SIZE + SPI_FINAL] = (float)label[a];
        final_of[a] = label[a];
    }
}

template <typename T>
int dev_alloc(ifx* h, T** p, size_t n)
{
    HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
    return IFX_OK;
}

int slic_buffers(ifx* h, SlicBuf** out, bool ahead = false)
{
    // superpixels run ahead for a frame whose call never came (the hint was only a hint): the side stream may still be writing the buffers -- whoever uses them next
    // on another stream queues behind that run
    if (!ahead && h->slic_ahead_busy) { HIPCHK(h, hipStreamWaitEvent(h->cur, h->ev_slic_ahead, 0)); h->slic_ahead_tick = -1; h->slic_ahead_busy = 0; }
    if (h->slic) { *out = (SlicBuf*)h->slic; return IFX_OK; }
    if (h->w < SPX || h->h < SPX) { h->err = "image smaller than one superpixel"; return IFX_E_INVALID; }
    SlicBuf* b = new SlicBuf();
    h->slic = b;
    b->P = h->P; b->mw = h->w / SPX; b->mh = h->h / SPX; b->spn = h->P / (SPX * SPX);
    if (b->mw * b->mh > b->spn) b->spn = b->mw * b->mh;
    b->adj_words = cdiv(b->spn, 32);
    const size_t P = b->P, S = b->spn;
    int r = 0;
    r |= dev_alloc(h, &b->rgb, P * 3); r |= dev_alloc(h, &b->depth, P); r |= dev_alloc(h, &b->dg, P);
    r |= dev_alloc(h, &b->xyz, P); r |= dev_alloc(h, &b->pos, P); r |= dev_alloc(h, &b->nor, P);
    r |= dev_alloc(h, &b->ccol, S); r |= dev_alloc(h, &b->cxy, S);
    r |= dev_alloc(h, &b->seg, P); r |= dev_alloc(h, &b->tmp, P); r |= dev_alloc(h, &b->fin, P);
    r |= dev_alloc(h, &b->sum1, S * NSUM * 2); b->sum2 = b->sum1 + S * NSUM;
    r |= dev_alloc(h, &b->adj, S * b->adj_words);
    r |= dev_alloc(h, &b->info, S * SPI_SIZE); r |= dev_alloc(h, &b->final_of, S * (1 + NB_MAX));   // [spn] + k_sp_edges' table [spn][NB_MAX]
    if (r) return IFX_E_HIP;
    b->h_info.resize(S * SPI_SIZE);
    *out = b;
    return IFX_OK;
}

// device stages; inputs are already in b->rgb / b->depth / b->seg
int slic_run(ifx* h, SlicBuf* b)
{
    const int w = h->w, hh = h->h, P = b->P, S = b->mw * b->mh;
    float nxy = 1.0f / (1.4242f * SPX), ncol = 5.0f / 1.7321f;   // seg_engine_GPU ctor :40-56 (XYZ)
    ncol *= ncol; nxy *= nxy;
    dim3 cells(cdiv(w, 16), cdiv(hh, 16)), tile(16, 16);
    HIPCHK(h, hipMemsetAsync(b->seg, 0, (size_t)P * 4, h->cur));
    LAUNCH(h, "slic_cvt", dim3(cdiv(P, 256)), dim3(256), k_slic_cvt, b->cur_rgb ? b->cur_rgb : (const uint8_t*)b->rgb, b->xyz, P);
    LAUNCH(h, "slic_init", dim3(cdiv(S, 256)), dim3(256), k_slic_init, b->xyz, b->ccol, b->cxy, b->mw, b->mh, w, hh);
    LAUNCH(h, "slic_assoc", cells, tile, k_slic_assoc, b->xyz, b->ccol, b->cxy, b->seg, b->mw, b->mh, w, hh, 0.6f, nxy, ncol);
    for (int it = 0; it < 5; it++) {   // my_settings.no_iters
        LAUNCH(h, "slic_update", dim3(S), dim3(256), k_slic_update, b->xyz, b->seg, b->ccol, b->cxy, b->mw, w, hh);
        LAUNCH(h, "slic_assoc", cells, tile, k_slic_assoc, b->xyz, b->ccol, b->cxy, b->seg, b->mw, b->mh, w, hh, 0.6f, nxy, ncol);
    }
    LAUNCH(h, "slic_enforce", cells, tile, k_slic_enforce, b->seg, b->tmp, w, hh);
    LAUNCH(h, "slic_enforce", cells, tile, k_slic_enforce, b->tmp, b->seg, w, hh);
    return IFX_OK;
}

int merge_run(ifx* h, SlicBuf* b)
{
    const int w = h->w, hh = h->h, P = b->P, S = b->spn;
    float4 cam = make_float4(h->cfg.cx, h->cfg.cy, (float)(1.0 / (double)h->cfg.fx), (float)(1.0 / (double)h->cfg.fy));
    dim3 cells(cdiv(w, 64), cdiv(hh, 4)), tile(64, 4);
    HIPCHK(h, hipMemsetAsync(b->sum1, 0, (size_t)S * NSUM * 2 * 8, h->cur));
    HIPCHK(h, hipMemsetAsync(b->adj, 0, (size_t)S * b->adj_words * 4, h->cur));
    LAUNCH(h, "sp_gauss", cells, tile, k_sp_gauss, b->cur_depth ? b->cur_depth : (const uint16_t*)b->depth, b->dg, w, hh);
    LAUNCH(h, "sp_posnor", cells, tile, k_sp_posnor, b->dg, cam, w, hh, S, b->pos, b->nor, b->seg);
    LAUNCH(h, "sp_sums", dim3(cdiv(w, 16), cdiv(hh, 16)), dim3(16, 16), k_sp_sums, b->seg, b->dg, b->pos, b->nor, w, hh, b->sum1, b->adj, b->adj_words);
    LAUNCH(h, "sp_first_avg", dim3(cdiv(S, 64)), dim3(64), k_sp_first_avg, b->sum1, b->adj, b->adj_words, S, b->info);
    LAUNCH(h, "sp_recluster", dim3(cdiv(w, 16), cdiv(hh, 16)), dim3(16, 16), k_sp_recluster, b->seg, b->dg, b->pos, b->nor, w, hh, b->info, b->sum2);
    LAUNCH(h, "sp_second_avg", dim3(cdiv(S, 64)), dim3(64), k_sp_second_avg, b->sum2, S, b->info);
    // connectSuperPixel on the device: no read-back inside a call
    if (S <= 1200) {
        LAUNCH(h, "sp_edges", dim3(cdiv(S * NB_MAX, 256)), dim3(256), k_sp_edges, S, (const float*)b->info, b->final_of + S);
        LAUNCH_SMEM(h, "sp_connect", dim3(1), dim3(1024), (size_t)S * (2 + NB_MAX) * 4, k_sp_connect, S, b->info, b->final_of, (const int*)(b->final_of + S));
    }
    else if (S <= 5000) LAUNCH_SMEM(h, "sp_connect", dim3(1), dim3(1024), (size_t)S * 12, k_sp_connect_big, S, b->info, b->final_of);
    else { h->err = "too many superpixels for the one-block connect pass"; return IFX_E_INVALID; }
    LAUNCH(h, "sp_final", dim3(cdiv(P, 256)), dim3(256), k_sp_final, b->seg, b->final_of, b->fin, P);
    return IFX_OK;
}

// masks live in `d_masks` ([nm][P] on the device) and are rewritten in place.  by_superpixel: count per
// re-clustered superpixel (b->seg) and fold through final_of (fused path); otherwise count b->fin directly.
// the counters of the region filter, cleared: nothing of the masks in it, so the device-scheduled call enqueues it before the masks are staged (`prepared`)
static int filter_prepare(ifx* h, SlicBuf* b, int nm)
{
    const int S = b->spn;
    size_t need = (size_t)(nm + 1) * S;
    if (need > b->num_cap) {
        if (b->num) hipFree(b->num);
        b->num = nullptr; b->num_cap = 0;
        HIPCHK(h, hipMalloc((void**)&b->num, need * 2 * 4));
        b->num_cap = need;
    }
    HIPCHK(h, hipMemsetAsync(b->num + need, 0, need * 4, h->cur));
    return IFX_OK;
}
int filter_run(ifx* h, SlicBuf* b, uint8_t* d_masks, int nm, bool by_superpixel, bool prepared = false)
{
    const int P = b->P, S = b->spn;
    size_t need = (size_t)(nm + 1) * S;
    if (by_superpixel && !prepared) { int r = filter_prepare(h, b, nm); if (r) return r; }
    if (need > b->num_cap) {
        if (b->num) hipFree(b->num);
        b->num = nullptr; b->num_cap = 0;
        HIPCHK(h, hipMalloc((void**)&b->num, need * 2 * 4));
        b->num_cap = need;
    }
    int* num_key = b->num + need;
    dim3 cells(cdiv(h->w, 16), cdiv(h->h, 16)), tile(16, 16);
    if (by_superpixel) {
        LAUNCH(h, "sp_count", cells, tile, k_sp_count, b->seg, d_masks, nm, h->w, h->h, S, num_key);
        hipEvent_t ea_ = nullptr;
        if (h->opt_kernel_timing) ifx_ktime_begin(h, "sp_count_regions", &ea_);
        hipLaunchKernelGGL(k_sp_count_regions, dim3(nm + 1), dim3(256), (size_t)S * 4, h->cur, num_key, b->final_of, S, b->num);
        if (h->opt_kernel_timing) ifx_ktime_end(h, "sp_count_regions", ea_);
    } else {
        HIPCHK(h, hipMemsetAsync(b->num, 0, need * 4, h->cur));
        LAUNCH(h, "sp_count", cells, tile, k_sp_count, b->fin, d_masks, nm, h->w, h->h, S, b->num);
    }
    LAUNCH(h, "sp_filter", dim3(cdiv(P, 256)), dim3(256), k_sp_filter, b->fin, d_masks, nm, P, S, b->num);
    return IFX_OK;
}

}  // namespace

void ifx_slic_free(ifx* h)
{
    SlicBuf* b = (SlicBuf*)h->slic;
    if (!b) return;
    void* ptrs[] = {b->rgb, b->depth, b->dg, b->xyz, b->pos, b->nor, b->ccol, b->cxy, b->seg, b->tmp, b->fin, b->sum1, b->adj, b->info, b->final_of, b->num};
    for (void* p : ptrs) if (p) hipFree(p);
    delete b;
    h->slic = nullptr;
}

// steps -1_1 .. -1_3 of processInstance (IF/Core/InstanceFusion.cpp:722-738): the masks in h->d_masks ([nm][P], already
// through clean-overlap) are refined in place on the device
// The frame of a segmentation call into the superpixel buffers.  rgb == depth == NULL: the frame most recently processed -- the call belongs to it, and its
// raw images are still in their frame slot (the slot is reused two frames later) -- is copied on the device; otherwise the caller's images go through the
// handle's pinned staging (free here: ifx_process_frame, its other user, synchronises before it returns; a copy from pageable memory would make the call
// wait for everything queued in front of it).
static int slic_load_frame(ifx* h, SlicBuf* b, const uint8_t* rgb, const uint16_t* depth)
{
    const size_t P = b->P;
    if (!rgb && !depth) {
        if (h->tick < 2) { h->err = "superpixel refinement of the resident frame: no frame has been processed yet"; return IFX_E_STATE; }
        const FrameSlot& f = h->slot[(size_t)h->last_frame_slot];
        b->cur_rgb = f.rgb; b->cur_depth = f.depth_raw;   // read in place: the slot is not reused before the frame after next, and the call is synchronous
        return IFX_OK;
    }
    if (!rgb || !depth) { h->err = "superpixel refinement needs the RGB and depth frame (or neither: the resident frame)"; return IFX_E_INVALID; }
    std::memcpy(h->rgb_stage, rgb, P * 3);
    std::memcpy(h->depth_stage, depth, P * 2);
    HIPCHK(h, hipMemcpyAsync(b->rgb, h->rgb_stage, P * 3, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(b->depth, h->depth_stage, P * 2, hipMemcpyHostToDevice, h->cur));
    b->cur_rgb = nullptr; b->cur_depth = nullptr;
    return IFX_OK;
}

// In two halves for the device-scheduled call: the superpixels and their merge need the frame only, so they are on the queue before the host starts copying the
// masks into pinned memory (2.4 MB for eight masks: 0.1 ms during which the device used to wait); the region filter follows the masks.
int ifx_superpixel_begin(ifx* h, const uint8_t* rgb, const uint16_t* depth)
{
    SlicBuf* b;
    if (!rgb && !depth && h->tick >= 2 && h->slic_ahead_tick == h->tick - 1) {   // the resident frame's superpixels ran ahead on the side stream: the call queues behind them
        HIPCHK(h, hipStreamWaitEvent(h->cur, h->ev_slic_ahead, 0));
        h->slic_ahead_tick = -1;
        h->slic_ahead_busy = 0;
        h->slic_ahead_used++;
        return IFX_OK;
    }
    int r = slic_buffers(h, &b);
    if (r) return r;
    if ((r = slic_load_frame(h, b, rgb, depth))) return r;
    if ((r = slic_run(h, b))) return r;
    return merge_run(h, b);
}
// Look-ahead of a segmentation call (ifx_should_segment: "not this frame, but the cadence says the next one").  SLIC and the superpixel merge read the frame only
// -- 27 dispatches, about half of a call -- and the announced next frame's raw images are already in their frame slot (copied there by its frame side, on the side
// stream): the same stream runs them now, under the next frame's tracker and map passes, and the call that comes waits for one event instead.  The same kernels on
// the same images: nothing about the result changes.  A call that does not come leaves the run unused (slic_buffers orders the next user behind it).
int ifx_superpixel_ahead(ifx* h)
{
    if (!h->opt_slic_ahead || !h->opt_two_streams || !h->stream_b || h->own) return IFX_OK;
    const FrameSlot& f = h->slot[h->tick & 1];
    if (f.for_tick != h->tick || h->slic_ahead_tick == h->tick) return IFX_OK;   // no frame announced ahead (or done already)
    if (h->tick >= 2 && h->slic_ahead_tick == h->tick - 1) return IFX_OK;        // the run for the frame just processed may still be claimed by that frame's call: the buffers are its
    SlicBuf* b;
    int r = slic_buffers(h, &b, true);   // (an unused earlier run: this one queues behind it on the same stream anyway; calls are synchronous, so the main stream holds no claim on the buffers)
    if (r) return r;
    if (!h->ev_slic_ahead) HIPCHK(h, hipEventCreateWithFlags(&h->ev_slic_ahead, hipEventDisableTiming));
    hipStream_t keep = h->cur;
    h->cur = h->stream_b;
    hipEvent_t ta = nullptr;
    ta = ifx_event_get(h); hipEventRecord(ta, h->cur);   // (always timed, like the calls themselves: two events per run)
    b->cur_rgb = f.rgb; b->cur_depth = f.depth_raw;   // behind the slot's copy-in on the same stream; the slot is reused two frames on, by the same stream
    if (!(r = slic_run(h, b))) r = merge_run(h, b);
    if (ta) { hipEvent_t tb = ifx_event_get(h); hipEventRecord(tb, h->cur); h->stage_pending.push_back({4, {ta, tb}}); }
    hipEventRecord(h->ev_slic_ahead, h->stream_b);
    h->cur = keep;
    if (r) return r;
    h->slic_ahead_tick = h->tick;
    h->slic_ahead_busy = 1;
    h->slic_ahead_runs++;
    return IFX_OK;
}
int ifx_superpixel_filter(ifx* h, int nm, bool prepared)
{
    SlicBuf* b;
    int r = slic_buffers(h, &b);
    if (r) return r;
    return filter_run(h, b, h->d_masks, nm, true, prepared);
}
int ifx_superpixel_filter_prepare(ifx* h, int nm)
{
    SlicBuf* b;
    int r = slic_buffers(h, &b);
    if (r) return r;
    return filter_prepare(h, b, nm);
}

int ifx_superpixel_refine(ifx* h, const uint8_t* rgb, const uint16_t* depth, int nm, int frame)
{
    (void)frame;
    SlicBuf* b;
    int r = slic_buffers(h, &b);
    if (r) return r;
    if ((r = slic_load_frame(h, b, rgb, depth))) return r;
    if ((r = slic_run(h, b))) return r;
    if ((r = merge_run(h, b))) return r;
    return filter_run(h, b, h->d_masks, nm, true);
}

// ---------------------------------------------------------------- C-ABI stage entry points
extern "C" int ifx_slic_segment(ifx_t* h, const uint8_t* rgb, int32_t* seg_out)
{
    if (!h || !rgb || !seg_out) return IFX_E_INVALID;
    SlicBuf* b;
    int r = slic_buffers(h, &b);
    if (r) return r;
    b->cur_rgb = nullptr;
    HIPCHK(h, hipMemcpyAsync(b->rgb, rgb, (size_t)b->P * 3, hipMemcpyHostToDevice, h->cur));
    if ((r = slic_run(h, b))) return r;
    HIPCHK(h, hipMemcpyAsync(seg_out, b->seg, (size_t)b->P * 4, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return b->mw * b->mh;
}

extern "C" int ifx_merge_superpixels(ifx_t* h, const uint16_t* depth, int32_t* seg_inout, int32_t* final_out, float* info_out)
{
    if (!h || !depth || !seg_inout || !final_out) return IFX_E_INVALID;
    SlicBuf* b;
    int r = slic_buffers(h, &b);
    if (r) return r;
    const size_t P = b->P;
    b->cur_depth = nullptr;
    HIPCHK(h, hipMemcpyAsync(b->depth, depth, P * 2, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(b->seg, seg_inout, P * 4, hipMemcpyHostToDevice, h->cur));
    if ((r = merge_run(h, b))) return r;
    HIPCHK(h, hipMemcpyAsync(seg_inout, b->seg, P * 4, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipMemcpyAsync(final_out, b->fin, P * 4, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipMemcpyAsync(b->h_info.data(), b->info, (size_t)b->spn * SPI_SIZE * 4, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    if (info_out) std::memcpy(info_out, b->h_info.data(), (size_t)b->spn * SPI_SIZE * 4);
    return b->spn;
}

extern "C" int ifx_mask_superpixel_filter(ifx_t* h, const int32_t* final_ids, uint8_t* masks, int nm)
{
    if (!h || !final_ids || nm < 0 || (nm > 0 && !masks)) return IFX_E_INVALID;
    if (nm == 0) return IFX_OK;
    SlicBuf* b;
    int r = slic_buffers(h, &b);
    if (r) return r;
    const size_t P = b->P;
    if ((r = ifx_ensure_masks(h, (size_t)nm * P))) return r;
    HIPCHK(h, hipMemcpyAsync(b->fin, final_ids, P * 4, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->d_masks, masks, (size_t)nm * P, hipMemcpyHostToDevice, h->cur));
    if ((r = filter_run(h, b, h->d_masks, nm, false))) return r;
    HIPCHK(h, hipMemcpyAsync(masks, h->d_masks, (size_t)nm * P, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}
