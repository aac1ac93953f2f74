// ifx_instance.hip -- instance layer of the path for gfx950.
//
// Host flow mirrors InstanceFusion::processInstance (IF/Core/InstanceFusion.cpp:655-1067) with the
// Mask-RCNN call replaced by the caller's pre-computed masks.  Device work: mask clean-up, per-pixel
// vote gathering + bounding boxes (fused; the reference materialises a 119 MB [97][H][W] int image
// first), model depth image, the vote update and the per-surfel arg-max scan over the planar vote
// store (12 fully coalesced 16-B loads per lane).  The two CPU passes of the reference (box IoU
// matching, depth flood fill) stay on the host, as in the reference.
#include <chrono>
#include "ifx_ctx.h"
#include <string.h>
#include <vector>
#include <algorithm>
#include <cmath>

#define DEAD_TIME (-1.0e9f)
#define NI IFX_NUM_INSTANCES

int ifx_superpixel_refine(ifx* h, const uint8_t* rgb, const uint16_t* depth, int nm, int frame);
int ifx_superpixel_begin(ifx* h, const uint8_t* rgb, const uint16_t* depth);
int ifx_superpixel_filter(ifx* h, int nm, bool prepared = false);
int ifx_superpixel_filter_prepare(ifx* h, int nm);
int ifx_superpixel_ahead(ifx* h);

// maskCleanOverlapKernel, IF/Core/InstanceFusionCuda.cu:118-131
__global__ void k_mask_clean_overlap(uint8_t* __restrict__ masks, int nm, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    int flag = 0;
    for (int m = nm - 1; m >= 0; m--) {
        size_t a = (size_t)m * P + p;
        uint8_t v = masks[a];
        if (flag && v) masks[a] = 0;
        if (!flag && v) flag = 1;
    }
}
// the same from the masks as they arrived (`ori`, kept: "BAK ORI MASK") into the working copy -- the copy and the clean in one pass -- and the per-mask verdict
// bytes of the call cleared on the way (the device-side call: two dispatches less)
__global__ void k_mask_clean_overlap_from(const uint8_t* __restrict__ ori, uint8_t* __restrict__ masks, int nm, int P, uint8_t* __restrict__ unavail)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (unavail && p < nm) unavail[p] = 0;
    if (p >= P) return;
    int flag = 0;
    for (int m = nm - 1; m >= 0; m--) {
        size_t a = (size_t)m * P + p;
        const uint8_t v = ori[a];
        masks[a] = (flag && v) ? (uint8_t)0 : v;
        if (!flag && v) flag = 1;
    }
}

// checkProjectDepthAndInstanceKernel, IF/Core/InstanceFusionCuda.cu:736-760
__global__ void k_check_project(const DevState* __restrict__ st, const int32_t* __restrict__ ids, const float4* __restrict__ votes, int cap, int w, int h, int downsample,
                                int* __restrict__ counts)
{
    int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y * blockDim.y + threadIdx.y;
    int x = gx * downsample, y = gy * downsample;
    if (x >= w || y >= h) return;
    int id = ids[y * w + x];
    if (id > 0 && id < st->count) {
        int s = 0;
        for (int q = 0; q < 12; q++) {
            float4 v = VOTE4(votes, id, q);
            int a, b;
            vote_decode(v.x, a, b); s += a + b;
            vote_decode(v.y, a, b); s += a + b;
            vote_decode(v.z, a, b); s += a + b;
            vote_decode(v.w, a, b); s += a + b;
        }
        atomicAdd(&counts[0], s);
    } else atomicAdd(&counts[1], 1);
}

// getProjectInstanceListKernel + computeProjectBoundingBoxKernel fused,
// IF/Core/InstanceFusionCuda.cu:781-807, 909-952.  bbox layout: [96][4] project boxes then [nm][4] mask boxes.
__global__ void k_init_bbox(int* __restrict__ bbox, int n, int w, int h)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 4) return;
    int t = i & 3;
    bbox[i] = t == 0 ? w + 1 : t == 1 ? -1 : t == 2 ? h + 1 : -1;
}
// extend a box {minX, maxX, minY, maxY}: almost every pixel lies inside the box already, so look (L2, no contention)
// before the atomic; a stale look can only cause a redundant atomic, never skip a needed one
__device__ __forceinline__ void bbox_extend(int* b, int x, int y)
{
    if (x < __hip_atomic_load(&b[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&b[0], x);
    if (x > __hip_atomic_load(&b[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&b[1], x);
    if (y < __hip_atomic_load(&b[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&b[2], y);
    if (y > __hip_atomic_load(&b[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&b[3], y);
}
// The pixels of a wave (2 rows x 32 columns of the 32 x 8 block) that extend the same box are reduced in the wave first: their ballot IS the reduction -- lowest /
// highest set bit of each row's half give min / max x, which halves are non-empty min / max y -- so a box costs a few scalar operations, one look and at most four
// atomics per wave (first four memory-side looks per pixel and box, then four shuffle reductions of six steps per box; the twelve vote floats4 and the mask bytes of a
// pixel are fetched in one batch each).  min / max commute: same boxes.
__device__ __forceinline__ void bbox_extend_ballot_lds(int* b, unsigned long long bal, int xb, int yb)
{
    const unsigned int lo = (unsigned int)bal, hi = (unsigned int)(bal >> 32);
    const int x0 = xb + min(lo ? __ffs(lo) - 1 : 32, hi ? __ffs(hi) - 1 : 32), x1 = xb + max(lo ? 31 - __clz(lo) : -1, hi ? 31 - __clz(hi) : -1);
    const int y0 = yb + (lo ? 0 : 1), y1 = yb + (hi ? 1 : 0);
    atomicMin(&b[0], x0); atomicMax(&b[1], x1); atomicMin(&b[2], y0); atomicMax(&b[3], y1);
}
// a block's box into the global one: almost every box lies inside it already, so look (L2) before the atomic; a stale look can only cause a redundant atomic
__device__ __forceinline__ void bbox_extend_box(int* b, const int* sb)
{
    if (sb[0] < __hip_atomic_load(&b[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&b[0], sb[0]);
    if (sb[1] > __hip_atomic_load(&b[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&b[1], sb[1]);
    if (sb[2] < __hip_atomic_load(&b[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&b[2], sb[2]);
    if (sb[3] > __hip_atomic_load(&b[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&b[3], sb[3]);
}
#define PB_ROWS 32   // k_project_bbox: blocks of 32 x PB_ROWS pixels
// MODE 0: instance boxes and mask boxes (the reference's two kernels as one).  The device-scheduled call splits them: MODE 1, the instance boxes -- the twelve vote
// float4 of every pixel's surfel, the heavy half, and nothing of the masks -- is on the queue BEFORE the host copies the masks into pinned memory (and writes the model
// depth under every pixel on the way: getProjectDepthMapKernel reads the same id); MODE 2, the mask boxes, follows the masks and needs one vote float4 per pixel.
template <int MODE>
__global__ void __launch_bounds__(32 * PB_ROWS) k_project_bbox(const DevState* __restrict__ st, const int32_t* __restrict__ ids, const float4* __restrict__ votes, int cap, const uint8_t* __restrict__ masks,
                               int nm, int w, int h, int* __restrict__ bbox, IdMap im, const float4* __restrict__ pc = nullptr, uint16_t* __restrict__ pdm = nullptr)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lane = (threadIdx.y * blockDim.x + threadIdx.x) & 63;
    const int xb = blockIdx.x * 32, yb = blockIdx.y * blockDim.y + (threadIdx.y & ~1);   // origin of this wave's 32 x 2 pixels (blocks of 32 x PB_ROWS)
    // Boxes of the block first, in LDS: a wave that went to the global box itself made every box a hot spot of one L2 channel (4800 waves x a handful of boxes x four
    // looks at the same few lines: most of the kernel's 85 us); now a block of 16 waves settles in LDS and sends one look per box it touched.
    __shared__ int s_box[NI + 8][4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
    for (int t = tid; t < (NI + 8) * 4; t += nthreads) s_box[t >> 2][t & 3] = (t & 1) ? (int)0x80000000 : 0x7fffffff;
    __syncthreads();
    const bool inside = x < w && y < h;
    const int P = w * h, k = inside ? y * w + x : 0;
    const int gid = inside ? ids[k] : 0;
    const int id = idmap_slot(im, st->count, gid);   // (sharded map: the pixels whose surfel another rank owns extend that rank's partial boxes)
    bool has = inside && id >= 0;
    int maxNum = 0, maxID = -1, first = 0;
    if (MODE == 1 && pdm && inside) {   // getProjectDepthMapKernel (k_project_depth) on the way
        uint16_t o = 0;
        if (id >= 0) {
            const float4 p = pc[id];
            const float dx = st->pose[3] - p.x, dy = st->pose[7] - p.y, dz = st->pose[11] - p.z;
            o = (uint16_t)(sqrtf(dx * dx + dy * dy + dz * dz) * 1186);
        }
        pdm[k] = o;
    }
    if (MODE == 2) {
        if (has) { int b_; vote_decode(VOTE4(votes, id, 0).x, first, b_); }
    } else if (has) {
        float4 v[12];
#pragma unroll
        for (int q = 0; q < 12; q++) v[q] = VOTE4(votes, id, q);
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const float f[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                int a, b;
                vote_decode(f[t], a, b);
                if (q == 0 && t == 0) first = a;
                if (a > maxNum) { maxNum = a; maxID = (q * 4 + t) * 2; }
                if (b > maxNum) { maxNum = b; maxID = (q * 4 + t) * 2 + 1; }
            }
        }
    }
    if (first == -1) has = false;   // instanceProjectMap[y*width+x] != -1 test (:915)
    // boxes of the projected instances: one reduction per distinct arg-max id in the wave (a wave rarely sees more than two or three)
    const int key = (MODE != 2 && has && maxID != -1) ? maxID : -1;
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int pick = __shfl(key, leader, 64);
        const unsigned long long grp = __ballot(key == pick);
        if (lane == leader) bbox_extend_ballot_lds(s_box[pick], grp, xb, yb);
        todo &= ~grp;
    }
    // boxes of the masks
    for (int m0 = 0; MODE != 1 && m0 < nm; m0 += 8) {
        uint8_t mb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) mb[u] = (has && m0 + u < nm) ? masks[(size_t)(m0 + u) * P + k] : (uint8_t)0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long bal = __ballot(mb[u] > 0);
            if (bal && lane == u) bbox_extend_ballot_lds(s_box[NI + u], bal, xb, yb);
        }
        __syncthreads();
        if (tid < 8 && s_box[NI + tid][1] >= s_box[NI + tid][0]) bbox_extend_box(&bbox[(NI + m0 + tid) * 4], s_box[NI + tid]);
        if (m0 + 8 < nm) {   // (uniform) another chunk of masks follows: re-arm their eight boxes
            __syncthreads();
            if (tid < 32) s_box[NI + (tid >> 2)][tid & 3] = (tid & 1) ? (int)0x80000000 : 0x7fffffff;
            __syncthreads();
        }
    }
    __syncthreads();
    if (MODE != 2 && tid < NI && s_box[tid][1] >= s_box[tid][0]) bbox_extend_box(&bbox[tid * 4], s_box[tid]);
}

// getProjectDepthMapKernel, IF/Core/InstanceFusionCuda.cu:977-996
__global__ void k_project_depth(const DevState* __restrict__ st, const int32_t* __restrict__ ids, const float4* __restrict__ pc, int P, int ratio, uint16_t* __restrict__ pdm, IdMap im)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int id = idmap_slot(im, st->count, ids[k]);
    uint16_t o = 0;
    if (id >= 0) {
        float4 p = pc[id];
        float dx = st->pose[3] - p.x, dy = st->pose[7] - p.y, dz = st->pose[11] - p.z;
        o = (uint16_t)(sqrtf(dx * dx + dy * dy + dz * dz) * ratio);
    }
    pdm[k] = o;
}

// updateSurfelMapInstanceKernel, IF/Core/InstanceFusionCuda.cu:1100-1139 (deleteNum = -1).  Like the
// reference this is a non-atomic read-modify-write: several pixels of one mask can see the same surfel;
// an atomicCAS loop makes the result independent of scheduling (every pixel's increment lands), which
// is also what the sequential oracle computes.
__global__ void k_vote_update(const DevState* __restrict__ st, const int32_t* __restrict__ ids, const uint8_t* __restrict__ mask, int P, int cap, int instanceID, int inc,
                              float* __restrict__ votes, IdMap im)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    if (!(mask[k] > 0)) return;
    const int id = idmap_slot(im, st->count, ids[k]);
    if (id < 0) return;
    int fi = instanceID / 2, p = instanceID % 2;
    // planar float4 store: float fi of surfel id lives in plane fi/4, component fi%4
    unsigned int* addr = (unsigned int*)&VOTEF(votes, id, fi);
    unsigned int old = *addr, assumed;
    do {
        assumed = old;
        int a, b;
        vote_decode(__uint_as_float(assumed), a, b);
        if (p == 0) a += inc; else b += inc;
        if (a >= 65535) a = 65535;
        if (b >= 65535) b = 65535;
        old = atomicCAS(addr, assumed, __float_as_uint(vote_encode(a, b)));
    } while (old != assumed);
}

// countAndColourSurfelMapKernel, IF/Core/InstanceFusionCuda.cu:1158-1200: the label scan.  Per surfel
// 192 B of votes read as 12 coalesced float4 plane loads + 8 B colour RMW + 4 B label.
// countAndColourSurfelMapKernel restricted to what a segmentation call can have changed.  A call changes votes only of the surfels under the
// id image (k_vote_update), so labels and colours of all other surfels are what the previous scan left -- except the surfels created since then,
// which the full scan would move from "no colour" (0) to the default colour: pass A does that from 16 bytes per slot, pass B redoes the arg-max
// for the surfel under every pixel (192 bytes each, <= P of them) exactly as the full scan would.  5 M surfels: 83 MB + <= 59 MB instead of
// 1.1 GB.  Anything that rewrites votes wholesale (upload, table eviction) sets ifx::labels_stale_all and the next call scans everything.
// `gate` (all three scan kernels): the first two words of the device-side call's control block (SegCtl: ff_incomplete, evict_at).  A call that cannot finish on the
// device -- the flood fill needs more relaxations, the table is full at some mask -- must not colour anything yet: colours are assigned ONCE (c.y == 0 or the
// default), and an instance registered before an eviction may be gone after it.  The scan then runs once, behind the host-driven tail, as in the reference.
#define SCAN_GATED(gate) ((gate) && ((gate)[0] != 0 || (gate)[1] >= 0))
__global__ __launch_bounds__(256) void k_colour_default(const DevState* __restrict__ st, const float2* __restrict__ tm, float2* __restrict__ col, int32_t* __restrict__ labels,
                                                        const int* __restrict__ gate)
{
    if (SCAN_GATED(gate)) return;
    const float defaultColor = 7434609;
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const float2 c = col[i];
        const bool live = tm[i].y > DEAD_TIME;
        if (c.y == 0 && live) col[i] = make_float2(c.x, defaultColor);   // (votes untouched since creation: no label, the default colour)
        if (!live && labels[i] != -1) labels[i] = -1;                    // a tombstone carries no label, as in the full scan (visible through ifx_owner_knn_export until a compaction)
    }
}
__global__ __launch_bounds__(256) void k_count_colour_px(const DevState* __restrict__ st, const int32_t* __restrict__ ids, int P, const float4* __restrict__ votes, int cap,
                                                         const float2* __restrict__ tm, float2* __restrict__ col, const float* __restrict__ inst_color, int32_t* __restrict__ labels,
                                                         IdMap im, const int* __restrict__ gate)
{
    if (SCAN_GATED(gate)) return;
    const float defaultColor = 7434609;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int i = idmap_slot(im, st->count, ids[k]);
    if (i < 0) return;
    float4 v[12];
#pragma unroll
    for (int q = 0; q < 12; q++) v[q] = VOTE4(votes, i, q);
    int best = -1, bestCount = 0;
#pragma unroll
    for (int q = 0; q < 12; q++) {
        float f[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int a, b;
            vote_decode(f[t], a, b);
            if (bestCount < a) { bestCount = a; best = (q * 4 + t) * 2; }
            if (bestCount < b) { bestCount = b; best = (q * 4 + t) * 2 + 1; }
        }
    }
    if (tm[i].y <= DEAD_TIME) { labels[i] = -1; return; }
    labels[i] = best;          // (several pixels can show the same surfel: they all write the same values)
    float2 c = col[i];
    if (c.y == 0 || c.y == defaultColor) {
        c.y = (best != -1) ? inst_color[best] : defaultColor;
        col[i] = c;
    }
}
__global__ __launch_bounds__(256) void k_count_colour(const DevState* __restrict__ st, const float4* __restrict__ votes, int cap, const float2* __restrict__ tm,
                                                      float2* __restrict__ col, const float* __restrict__ inst_color, int32_t* __restrict__ labels, const int* __restrict__ gate)
{
    if (SCAN_GATED(gate)) return;
    // the whole map: four lanes per surfel, each instruction of a wave reads 16 x 64 contiguous bytes of the 192-byte records; the arg-max of the four quarters is
    // merged "larger count, then smaller index" = the sequential first-maximum rule
    (void)cap;
    const float defaultColor = 7434609;
    const int n = st->count, r = threadIdx.x & 3;
    const int per = blockDim.x >> 2;
    for (int i0 = blockIdx.x * per; i0 < n; i0 += per * gridDim.x) {   // (uniform trip count per block: the shuffles below see whole groups)
        const int i = i0 + (threadIdx.x >> 2);
        int best = -1, bestCount = 0;
        if (i < n) {
            float4 v[3];
#pragma unroll
            for (int j = 0; j < 3; j++) v[j] = VOTE4(votes, i, j * 4 + r);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int q = j * 4 + r;
                float f[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    int a, b;
                    vote_decode(f[t], a, b);
                    if (bestCount < a) { bestCount = a; best = (q * 4 + t) * 2; }
                    if (bestCount < b) { bestCount = b; best = (q * 4 + t) * 2 + 1; }
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) {
            const int oc = __shfl_xor(bestCount, o, 64), ob = __shfl_xor(best, o, 64);
            if (oc > bestCount || (oc == bestCount && oc > 0 && ob < best)) { bestCount = oc; best = ob; }
        }
        if (i >= n || r != 0) continue;
        if (tm[i].y <= DEAD_TIME) { labels[i] = -1; continue; }
        labels[i] = best;
        float2 c = col[i];
        if (c.y == 0 || c.y == defaultColor) {
            c.y = (best != -1) ? inst_color[best] : defaultColor;
            col[i] = c;
        }
    }
}

// computeMaxCountInMapKernel, IF/Core/InstanceFusionCuda.cu:1012-1036 (per-wave pre-reduction instead
// of 192 global atomics per surfel)
__global__ __launch_bounds__(256) void k_max_count(const DevState* __restrict__ st, const float4* __restrict__ votes, int cap, const float2* __restrict__ tm,
                                                   int* __restrict__ maxv, int* __restrict__ sumv)
{
    (void)cap;
    __shared__ int smax[NI], ssum[NI];
    for (int t = threadIdx.x; t < NI; t += blockDim.x) { smax[t] = 0; ssum[t] = 0; }
    __syncthreads();
    const size_t n4 = (size_t)st->count * 12;   // one thread per float4 of the records: consecutive threads, consecutive bytes
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (size_t)blockDim.x * gridDim.x) {
        const int i = (int)(g / 12), q = (int)(g - (size_t)i * 12);
        if (tm[i].y <= DEAD_TIME) continue;
        const float4 v = votes[g];
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int a, b;
            const int k = (q * 4 + t) * 2;
            vote_decode(f[t], a, b);
            if (a > 0) atomicMax(&smax[k], a);
            if (b > 0) atomicMax(&smax[k + 1], b);
            if (a) atomicAdd(&ssum[k], a);
            if (b) atomicAdd(&ssum[k + 1], b);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NI; t += blockDim.x) {
        if (smax[t]) atomicMax(&maxv[t], smax[t]);
        if (ssum[t]) atomicAdd(&sumv[t], ssum[t]);
    }
}

// cleanInstanceTableMapKernel, IF/Core/InstanceFusionCuda.cu:1057-1084
__global__ void k_clean_table(const DevState* __restrict__ st, float* __restrict__ votes, int cap, const int* __restrict__ clean_list)
{
    (void)cap;
    const size_t nf = (size_t)st->count * IFX_VF;   // one thread per float of the records
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < nf; g += (size_t)blockDim.x * gridDim.x) {
        const int fi = (int)(g % IFX_VF);
        const int c1 = clean_list[2 * fi], c2 = clean_list[2 * fi + 1];
        if (!c1 && !c2) continue;
        int a, b;
        vote_decode(votes[g], a, b);
        if (c1) a = 0;
        if (c2) b = 0;
        votes[g] = vote_encode(a, b);
    }
}

// ------------------------------------------------------------------ host side
int ifx_alloc_instance(ifx* h)
{
    uint32_t s = 0x1F5u;
    for (int i = 0; i < NI; i++) {
        h->inst_class[i] = -1;
        // createInstanceTable (IF/Core/InstanceTable.cpp:11-35) draws colours from unseeded rand();
        // a fixed LCG is used instead -- the colours are arbitrary in the reference too.
        int c[3];
        for (int k = 0; k < 3; k++) { s = s * 1664525u + 1013904223u; c[k] = (int)((s >> 8) % 255u); }
        h->inst_color[i] = (float)((c[0] << 16) + (c[1] << 8) + c[2]);
    }
    HIPCHK(h, hipMalloc(&h->d_inst_color, NI * 4));
    HIPCHK(h, hipMemcpy(h->d_inst_color, h->inst_color, NI * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&h->d_pdm, (size_t)h->P * 2));
    HIPCHK(h, hipMalloc(&h->d_bbox, (NI + 256) * 16));
    HIPCHK(h, hipMalloc(&h->d_inst_stats, NI * 2 * 4));
    HIPCHK(h, hipMalloc(&h->d_clean_list, NI * 4));
    return IFX_OK;
}
void ifx_free_instance(ifx* h)
{
    hipFree(h->d_inst_color); hipFree(h->d_masks); hipFree(h->d_masks_ori); hipFree(h->d_unavail); hipFree(h->d_ff_label); hipFree(h->d_pdm); hipFree(h->d_bbox); hipFree(h->d_inst_stats); hipFree(h->d_clean_list);
    hipFree(h->d_segctl);
    if (h->h_segctl) hipHostFree(h->h_segctl);
    if (h->h_masks_stage) hipHostFree(h->h_masks_stage);
}

int ifx_ensure_masks(ifx* h, size_t bytes)
{
    if (bytes <= h->masks_cap) return IFX_OK;
    if (h->d_masks) hipFree(h->d_masks);
    if (h->d_masks_ori) hipFree(h->d_masks_ori);
    h->d_masks = nullptr; h->d_masks_ori = nullptr; h->masks_cap = 0;
    HIPCHK(h, hipMalloc(&h->d_masks, bytes));
    HIPCHK(h, hipMalloc(&h->d_masks_ori, bytes));
    if (!h->d_unavail) HIPCHK(h, hipMalloc(&h->d_unavail, 512));
    h->masks_cap = bytes;
    return IFX_OK;
}

extern "C" int ifx_mask_clean_overlap(ifx_t* h, uint8_t* masks, int n)
{
    if (!h || !masks || n < 0) return IFX_E_INVALID;
    if (n == 0) return IFX_OK;
    size_t bytes = (size_t)n * h->P;
    int r = ifx_ensure_masks(h, bytes);
    if (r) return r;
    HIPCHK(h, hipMemcpyAsync(h->d_masks, masks, bytes, hipMemcpyHostToDevice, h->cur));
    LAUNCH(h, "mask_clean_overlap", dim3(cdiv(h->P, 256)), dim3(256), k_mask_clean_overlap, h->d_masks, n, h->P);
    HIPCHK(h, hipMemcpyAsync(masks, h->d_masks, bytes, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}

// whetherDoSegmentation, IF/Core/InstanceFusion.cpp:192-238
static int mask_geometric_filter_device(ifx* h, const uint16_t* d_depth, uint8_t* d_masks, const uint8_t* d_ori, int nm, uint8_t* d_unavail, int fixed_rounds = 0, bool resume = false,
                                        bool verdict_by_caller = false);
// maskGeometricFilter as a stage (host buffers): depth = model depth under the camera (u16, 1186 units per metre as
// getProjectDepthMap produces it), masks in/out, ori = masks before clean-overlap, unavailable in/out
extern "C" int ifx_mask_geometric_filter(ifx_t* h, const uint16_t* depth, uint8_t* masks, const uint8_t* ori, int n, uint8_t* unavailable)
{
    if (!h || !depth || n < 0 || n > 256 || (n > 0 && (!masks || !ori || !unavailable))) return IFX_E_INVALID;
    if (n == 0) return IFX_OK;
    const size_t P = h->P, bytes = (size_t)n * P;
    int r = ifx_ensure_masks(h, bytes);
    if (r) return r;
    HIPCHK(h, hipMemcpyAsync(h->d_pdm, depth, P * 2, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->d_masks, masks, bytes, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->d_masks_ori, ori, bytes, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->d_unavail, unavailable, n, hipMemcpyHostToDevice, h->cur));
    if ((r = mask_geometric_filter_device(h, h->d_pdm, h->d_masks, h->d_masks_ori, n, h->d_unavail))) return r;
    HIPCHK(h, hipMemcpyAsync(masks, h->d_masks, bytes, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipMemcpyAsync(unavailable, h->d_unavail, n, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}

extern "C" int ifx_should_segment(ifx_t* h, int frame)
{
    if (!h) return IFX_E_INVALID;
    const int downsample = 10, fixedL = 2, fixedH = 45;
    int count[2];
    if (h->seg_counts_valid) {
        // the frame that just ran accumulated the two sums while it rendered ids_after (k_raster_finish) and left them in the
        // pinned frame result: nothing to launch, only the frame to wait for (not the stream: the next frame's tracker may
        // already be queued behind it)
        HIPCHK(h, hipEventSynchronize(h->ev_result));
        // (the one place an asynchronous host looks at every frame's result: the sticky failures are reported here too, not only by ifx_sync)
        if (h->h_result->overflow) { h->err = "surfel store capacity exceeded"; return IFX_E_CAPACITY; }
        count[0] = h->h_result->seg_counts[0]; count[1] = h->h_result->seg_counts[1];
    } else {
        int* cnt = h->d_inst_stats;
        HIPCHK(h, hipMemsetAsync(cnt, 0, 8, h->cur));
        int gw = cdiv(h->w, downsample), gh = cdiv(h->h, downsample);
        LAUNCH(h, "check_project", dim3(cdiv(gw, 16), cdiv(gh, 16)), dim3(16, 16), k_check_project, h->d_state, h->ids_after, (const float4*)h->votes, h->cap, h->w, h->h, downsample, cnt);
        HIPCHK(h, hipMemcpyAsync(count, cnt, 8, hipMemcpyDeviceToHost, h->cur));
        HIPCHK(h, hipStreamSynchronize(h->cur));
    }
    int w = h->w, hh = h->h;
    bool test1 = count[0] > (w / downsample * hh / downsample * 0.48 * 30);
    bool test2 = count[1] < (w / downsample * hh / downsample * 0.2);
    const int gap = (test1 || test2) ? fixedH : fixedL;
    if (frame - h->last_seg_frame > gap) { h->last_seg_frame = frame; return 1; }
    // Not this frame -- but if the cadence says "the next one", that frame draws the WHOLE id image while it rasterises its prediction (one pass over the same lists:
    // ~20 us) instead of leaving the rest of the image to the call (a walk of its own over the view lists + a resolve at the head of the call: ~80 us).  A hint only:
    // the decision is taken again on the next frame's own sums, and a call that finds the sparse image renders the rest as before (ifx_ids_ensure).
    const bool next = frame + 1 - h->last_seg_frame > gap;
    if (next) h->ids_full_hint = 1;
    if (next || h->opt_slic_ahead == 2) {   // ... and its superpixels (frame-only work, half of a call) start now on the side stream, if the frame is announced
        int r = ifx_superpixel_ahead(h);    // (option value 2, for tests: for every announced frame, whatever the cadence says)
        if (r) return r;
    }
    return 0;
}

// getDepthThreshold, IF/Core/InstanceFusion.h:170-176
__device__ __forceinline__ float depth_threshold_dev(int depth)
{
    float t = 0.074f * depth - 246.0f;
    t = fmaxf(50.0f, t);
    t = fminf(420.0f, t);
    return t;
}

// filterAreaCompute + maskGeometricFilter, IF/Core/InstanceFusion.cpp:470-593, on the device.
// The reference floods each mask on the CPU: pixels (interior, mask set, model depth valid) are visited in row-major
// order, an unvisited one seeds a breadth-first fill along edges  now -> neighbour  that exist when
// |depth[now] - depth[neighbour]| < threshold(depth[now])  (the threshold of the SOURCE pixel, so an edge can be
// one-way beyond 4 m), and the seed's sequence number is the region id.  A filled region is closed under forward
// reachability, hence a pixel ends up in the region of the EARLIEST pixel (row-major) that reaches it:
//     label(p) = min { q : q ~> p }.
// That fixpoint is reached by monotone min-propagation along the directed edges in any order: 32x32 tiles relax
// in LDS until nothing changes, one pointer jump (label <- label[label], valid by transitivity) per launch carries
// labels across tiles, and launches repeat until a launch changes nothing.
#define FF_T 32
#define FF_SLOTS 64
struct FFArgs {
    const uint16_t* depth; uint8_t* masks; const uint8_t* ori; const uint8_t* skip;   // skip[m] != 0: mask left alone
    int* label; int* cnt; int* meta;   // meta[m*32 + 0] oriPoints, [1] kept count, [2] finalPoints, [4..24) kept region ids
    int* changed;                      // [it] set when relaxation launch `it` lowered a label (host-driven rounds use slot 0 only)
    const int* gate;                   // fixed schedule: the tail kernels run only when *gate == 0 (the last scheduled relaxation changed nothing: fixpoint); nullptr: always
    uint8_t* tile_active;              // [nm][tiles]: the tile holds a labelled pixel (a tile without one has nothing to relax, whatever its halo says: labels only ever move between mask pixels)
    int nm, w, h;
};

__global__ void k_ff_prep(FFArgs a, const uint8_t* __restrict__ unavail, uint8_t* __restrict__ skip, int n_tiles)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles) a.tile_active[t] = 0;
    if (t < a.nm * 32) a.meta[t] = 0;
    if (t < FF_SLOTS) a.changed[t] = 0;
    if (t < a.nm) skip[t] = unavail[t];
}
__global__ void k_ff_init(FFArgs a)
{
    const int P = a.w * a.h, k = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    bool in = k < P && !a.skip[m];
    int x = in ? k % a.w : 0, y = in ? k / a.w : 0;
    bool interior = in && x >= 1 && x < a.w - 1 && y >= 1 && y < a.h - 1;
    bool o = interior && a.ori[(size_t)m * P + k];
    if (k < P) {
        bool v = interior && a.masks[(size_t)m * P + k] && a.depth[k];
        a.label[(size_t)m * P + k] = v ? k : -1;
        a.cnt[(size_t)m * P + k] = 0;
        if (v) a.tile_active[(size_t)m * ((a.w + FF_T - 1) / FF_T) * ((a.h + FF_T - 1) / FF_T) + (y / FF_T) * ((a.w + FF_T - 1) / FF_T) + x / FF_T] = 1;   // (same value from every writer)
    }
    unsigned long long b = __ballot(o);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&a.meta[m * 32], __popcll(b));
}

// `it`: index of this launch in a fixed schedule.  A launch whose predecessor changed nothing has nothing to do either (that predecessor re-checked every edge:
// the fixpoint) and returns at its first instruction, leaving its own flag clear -- so the host can enqueue the whole schedule without looking.
//
// Inside a tile the relaxation runs as SWEEPS (round 3): a thread owns a row (then a column) of the 34 x 34 window and carries the minimum along it in one
// sequential pass -- left to right, right to left, top to bottom, bottom to top, halo cells as sources -- so a label crosses the whole tile in one pass instead of
// one pixel per barrier-separated iteration (the first version: up to 128 iterations of a 256-thread block, 140 us for the first launch of a call; now a handful of
// rounds of four passes on one wave).  Same fixpoint: min-propagation along the same directed edges (the threshold is the SOURCE pixel's), in another order.
__device__ __forceinline__ bool ff_pull(int& l, int dp_unused, int ql, int dq, int dp)
{
    (void)dp_unused;
    if (ql < 0 || ql >= l) return false;
    if ((float)abs(dq - dp) < depth_threshold_dev(dq)) { l = ql; return true; }   // edge q -> p, threshold of the source q
    return false;
}
__global__ void __launch_bounds__(256) k_ff_relax(FFArgs a, int it)
{
    __shared__ int s_lab[FF_T + 2][FF_T + 3];          // (+1 column: rows and columns land on different banks)
    __shared__ unsigned short s_d[FF_T + 2][FF_T + 4];
    if (it > 0 && a.changed[it - 1] == 0) return;
    const int P = a.w * a.h, m = blockIdx.z;
    if (a.skip[m] || !a.tile_active[((size_t)m * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) return;   // (most (mask, tile) pairs: the masks cover a fraction of the image)
    int* lab = a.label + (size_t)m * P;
    const int x0 = blockIdx.x * FF_T - 1, y0 = blockIdx.y * FF_T - 1, tid = threadIdx.x;
    {   // the window: all loads of a thread in one batch, then the pointer jumps of the tile's own pixels in a second one (two round trips, not ten)
        constexpr int NL = ((FF_T + 2) * (FF_T + 2) + 255) / 256;
        int l[NL], d[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int t = tid + q * 256, ly = t / (FF_T + 2), lx = t - ly * (FF_T + 2), x = x0 + lx, y = y0 + ly;
            const bool in = t < (FF_T + 2) * (FF_T + 2) && x >= 0 && x < a.w && y >= 0 && y < a.h;
            l[q] = in ? lab[y * a.w + x] : -1;
            d[q] = in ? a.depth[y * a.w + x] : 0;
        }
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int t = tid + q * 256, ly = t / (FF_T + 2), lx = t - ly * (FF_T + 2);
            const bool interior = lx >= 1 && lx <= FF_T && ly >= 1 && ly <= FF_T;
            const int g = (interior && l[q] >= 0) ? lab[l[q]] : -1;   // one pointer jump: my label reaches me, so does ITS label
            if (g >= 0 && g < l[q]) l[q] = g;
        }
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int t = tid + q * 256, ly = t / (FF_T + 2), lx = t - ly * (FF_T + 2);
            if (t < (FF_T + 2) * (FF_T + 2)) { s_lab[ly][lx] = l[q]; s_d[ly][lx] = (unsigned short)d[q]; }
        }
    }
    __syncthreads();
    // A thread pulls its line into registers, sweeps it forth and back there (no LDS latency inside the dependent chain) and writes back what changed.
    const int line = (tid & 31) + 1;   // the row / column of this thread: 1 .. FF_T (one half-wave sweeps; the whole block of 256 loads and stores)
    for (int round = 0; round < 4 * FF_T; round++) {
        int any = 0;
#pragma unroll 1
        for (int dir = 0; dir < 2; dir++) {   // 0: rows, 1: columns
            if (tid < 32) {
                int l[FF_T + 2], d[FF_T + 2];
#pragma unroll
                for (int x = 0; x < FF_T + 2; x++) { l[x] = dir ? s_lab[x][line] : s_lab[line][x]; d[x] = dir ? s_d[x][line] : s_d[line][x]; }
                unsigned int ch = 0;
#pragma unroll
                for (int x = 1; x <= FF_T; x++)
                    if (l[x] >= 0 && l[x - 1] >= 0 && l[x - 1] < l[x] && (float)abs(d[x - 1] - d[x]) < depth_threshold_dev(d[x - 1])) { l[x] = l[x - 1]; ch |= 1u << (x - 1); }
#pragma unroll
                for (int x = FF_T; x >= 1; x--)
                    if (l[x] >= 0 && l[x + 1] >= 0 && l[x + 1] < l[x] && (float)abs(d[x + 1] - d[x]) < depth_threshold_dev(d[x + 1])) { l[x] = l[x + 1]; ch |= 1u << (x - 1); }
                if (ch) {
                    any = 1;
#pragma unroll
                    for (int x = 1; x <= FF_T; x++)
                        if (ch & (1u << (x - 1))) { if (dir) s_lab[x][line] = l[x]; else s_lab[line][x] = l[x]; }
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(any)) break;
    }
    int dirty = 0;
    for (int t = tid; t < FF_T * FF_T; t += 256) {
        const int ly = t / FF_T + 1, lx = t - (ly - 1) * FF_T + 1, x = x0 + lx, y = y0 + ly;
        if (x < a.w && y < a.h) {
            const int l = s_lab[ly][lx];
            if (l >= 0 && l != lab[y * a.w + x]) { lab[y * a.w + x] = l; dirty = 1; }
        }
    }
    if (__syncthreads_or(dirty) && tid == 0) a.changed[it] = 1;
}

// Union-find start of the fill (round 3).  Pixels joined by edges that exist in BOTH directions reach each other, so they end with the same label, whatever else
// happens: their components can be merged with the lock-free union-find of connected-component labelling instead of walking labels across the image one tile
// border per launch.  k_ff_tile_uf leaves every pixel pointing at the smallest pixel of its component INSIDE its tile (a forest of depth one, roots point at
// themselves); k_ff_merge unites the trees across every tile border edge (the larger root is hung under the smaller: atomicMin, retried until it sticks);
// k_ff_flatten points every pixel at its root = the smallest pixel of the whole component.  Every label is still "a pixel that reaches me", so the directed
// relaxation that follows starts from a valid state and -- where every edge is two-way (model depth under 4 m: one threshold) -- finds nothing left to do.
// Tile-local half: union-find in LDS over the two-way edges INSIDE a 32 x 32 tile (right and lower neighbour of every pixel: each edge once), then every pixel
// points at the smallest pixel of its component within the tile (row-major order inside a tile is the global order restricted to it).  No rounds, no halo: the
// edges across tile borders are k_ff_merge's.  (First version: the relaxation kernel restricted to two-way edges, sweeps to the local fixpoint, 64 us a call; this one 32.)
__device__ __forceinline__ int ff_find_lds(const volatile int* par, int x)
{
    int p = par[x];
    while (p != x) { x = p; p = par[x]; }
    return x;
}
__global__ void __launch_bounds__(256) k_ff_tile_uf(FFArgs a)
{
    __shared__ int par[FF_T * FF_T];
    __shared__ unsigned short dep[FF_T * FF_T];
    const int P = a.w * a.h, m = blockIdx.z;
    if (a.skip[m] || !a.tile_active[((size_t)m * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) return;
    int* lab = a.label + (size_t)m * P;
    const int x0 = blockIdx.x * FF_T, y0 = blockIdx.y * FF_T, tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < FF_T * FF_T / 256; q++) {
        const int t = tid + q * 256, ly = t / FF_T, lx = t - ly * FF_T, x = x0 + lx, y = y0 + ly;
        const bool in = x < a.w && y < a.h;
        par[t] = (in && lab[y * a.w + x] >= 0) ? t : -1;
        dep[t] = in ? a.depth[y * a.w + x] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FF_T * FF_T / 256; q++) {
        const int t = tid + q * 256, ly = t / FF_T, lx = t - ly * FF_T;
        if (par[t] < 0) continue;
        const int dp = dep[t];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (e == 0 ? lx + 1 >= FF_T : ly + 1 >= FF_T) continue;
            const int u = e == 0 ? t + 1 : t + FF_T;
            if (par[u] < 0) continue;
            const int dq = dep[u];
            const float dd = (float)abs(dp - dq);
            if (!(dd < depth_threshold_dev(dp) && dd < depth_threshold_dev(dq))) continue;
            int ra = t, rb = u;
            for (;;) {
                ra = ff_find_lds(par, ra); rb = ff_find_lds(par, rb);
                if (ra == rb) break;
                if (ra > rb) { const int s_ = ra; ra = rb; rb = s_; }
                const int old = atomicMin(&par[rb], ra);
                if (old == rb) break;
                rb = old;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FF_T * FF_T / 256; q++) {
        const int t = tid + q * 256, ly = t / FF_T, lx = t - ly * FF_T;
        if (par[t] < 0) continue;
        const int r = ff_find_lds(par, t), ry = r / FF_T, rx = r - ry * FF_T;
        if (r != t) lab[(y0 + ly) * a.w + x0 + lx] = (y0 + ry) * a.w + x0 + rx;
    }
}
__device__ __forceinline__ int ff_find(const int* lab, int x)
{
    int p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}
__global__ void k_ff_merge(FFArgs a)
{
    const int m = blockIdx.y;
    if (a.skip[m]) return;
    const int nbx = (a.w - 1) / FF_T, nby = (a.h - 1) / FF_T;   // vertical / horizontal tile borders inside the image
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nv = nbx * a.h, nh = nby * a.w;
    if (t >= nv + nh) return;
    int p, q;
    if (t < nv) { const int b = t / a.h, y = t - b * a.h, x = (b + 1) * FF_T; p = y * a.w + x - 1; q = p + 1; }
    else { const int u = t - nv, b = u / a.w, x = u - b * a.w, y = (b + 1) * FF_T; p = (y - 1) * a.w + x; q = p + a.w; }
    int* lab = a.label + (size_t)m * a.w * a.h;
    if (lab[p] < 0 || lab[q] < 0) return;
    const int dp = a.depth[p], dq = a.depth[q];
    const float dd = (float)abs(dp - dq);
    if (!(dd < depth_threshold_dev(dp) && dd < depth_threshold_dev(dq))) return;
    int ra = p, rb = q;
    for (;;) {
        ra = ff_find(lab, ra); rb = ff_find(lab, rb);
        if (ra == rb) break;
        if (ra > rb) { const int s_ = ra; ra = rb; rb = s_; }
        const int old = atomicMin(lab + rb, ra);
        if (old == rb) break;   // rb was a root and now hangs under ra
        rb = old;               // somebody hung it elsewhere first: unite with that tree
    }
}
__global__ void k_ff_flatten(FFArgs a)
{
    const int P = a.w * a.h, k = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (k >= P || a.skip[m]) return;
    int* lab = a.label + (size_t)m * P;
    const int l = lab[k];
    if (l < 0) return;
    const int r = ff_find(lab, l);
    if (r != l) __hip_atomic_store(lab + k, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// region sizes: one counter per root pixel; lanes of a wave grouped by label (regions are large)
__global__ void k_ff_count(FFArgs a)
{
    if (a.gate && *a.gate) return;
    const int P = a.w * a.h, k = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    int l = (k < P && !a.skip[m]) ? a.label[(size_t)m * P + k] : -1;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(l >= 0);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int ll = __shfl(l, leader);
        unsigned long long grp = __ballot(l == ll);
        todo &= ~grp;
        if (lane == leader) atomicAdd(&a.cnt[(size_t)m * P + ll], __popcll(grp));
    }
}
// regions holding more than a quarter of the original mask are kept (IF/Core/InstanceFusion.cpp:560)
__global__ void k_ff_select(FFArgs a)
{
    if (a.gate && *a.gate) return;
    const int P = a.w * a.h, k = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (k >= P || a.skip[m] || a.label[(size_t)m * P + k] != k) return;
    float points = (float)a.cnt[(size_t)m * P + k], oriPoints = (float)a.meta[m * 32];
    if (points / oriPoints > 0.25f) {
        int slot = atomicAdd(&a.meta[m * 32 + 1], 1);
        if (slot < 20) a.meta[m * 32 + 4 + slot] = k;
    }
}
__global__ void k_ff_apply(FFArgs a)
{
    if (a.gate && *a.gate) return;
    const int P = a.w * a.h, k = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    bool keep = false;
    if (k < P && !a.skip[m]) {
        const int l = a.label[(size_t)m * P + k], n = min(a.meta[m * 32 + 1], 20);
        for (int j = 0; j < n; j++) keep = keep || (l >= 0 && l == a.meta[m * 32 + 4 + j]);
        a.masks[(size_t)m * P + k] = keep ? 255 : 0;
    }
    unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&a.meta[m * 32 + 2], __popcll(b));
}
// a mask that lost more than 35 % of its pixels is unusable (:590)
__global__ void k_ff_verdict(FFArgs a, uint8_t* unavailable)
{
    if (a.gate && *a.gate) return;
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.nm || a.skip[m]) return;
    float finalPoints = (float)a.meta[m * 32 + 2], oriPoints = (float)a.meta[m * 32];
    if (finalPoints / oriPoints < 0.65f) unavailable[m] = 1;
}

// d_unavail: [nm] bytes in/out (set entries are skipped, as `continue` at :520); masks / ori: device, [nm][P]
// fixed_rounds == 0: relaxations until a launch changes nothing, the host looking every 7 launches (stage API, sharded calls, slow path).
// fixed_rounds  > 0: exactly that many relaxation launches are enqueued -- the ones behind the fixpoint return at once -- and the tail (region sizes, selection,
//   verdict) only runs when the last one found nothing to change; otherwise *ff_gate() stays set and the caller finishes with resume = true.  No host look at all.
static FFArgs ff_args(ifx* h, const uint16_t* d_depth, uint8_t* d_masks, const uint8_t* d_ori, int nm)
{
    FFArgs a;
    a.depth = d_depth; a.masks = d_masks; a.ori = d_ori; a.nm = nm; a.w = h->w; a.h = h->h;
    a.label = h->d_ff_label; a.cnt = h->d_ff_label + h->ff_cap; a.meta = a.cnt + h->ff_cap; a.changed = a.meta + 256 * 32;
    uint8_t* d_skip = (uint8_t*)(a.changed + FF_SLOTS);
    a.skip = d_skip;
    a.tile_active = d_skip + 256;
    a.gate = nullptr;
    return a;
}
// verdict_by_caller: the last step (a mask that lost more than 35 % is unusable) is left to the kernel the caller launches next (k_seg_register: one dispatch less on the call's chain)
static int mask_geometric_filter_device(ifx* h, const uint16_t* d_depth, uint8_t* d_masks, const uint8_t* d_ori, int nm, uint8_t* d_unavail, int fixed_rounds, bool resume, bool verdict_by_caller)
{
    const int P = h->P;
    size_t need = (size_t)nm * P;
    if (need > h->ff_cap) {
        if (h->d_ff_label) hipFree(h->d_ff_label);
        h->d_ff_label = nullptr; h->ff_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_ff_label, need * 2 * 4 + (size_t)256 * 32 * 4 + FF_SLOTS * 4 + 256 + (size_t)256 * cdiv(h->w, FF_T) * cdiv(h->h, FF_T)));
        h->ff_cap = need;
    }
    FFArgs a = ff_args(h, d_depth, d_masks, d_ori, nm);
    uint8_t* d_skip = (uint8_t*)a.skip;
    dim3 per_px(cdiv(P, 256), nm);
    dim3 tiles(cdiv(h->w, FF_T), cdiv(h->h, FF_T), nm);
    if (!resume) {
        // one launch re-arms the fill's bookkeeping (tile flags, per-mask counters, the schedule's "changed" words) and freezes who is skipped (the verdict must not
        // change that mid-way): three fills and a copy, four dispatches of a latency-bound chain, before
        LAUNCH(h, "ff_prep", dim3(cdiv(std::max(nm * cdiv(h->w, FF_T) * cdiv(h->h, FF_T), nm * 32), 256)), dim3(256), k_ff_prep, a, (const uint8_t*)d_unavail, d_skip,
               nm * cdiv(h->w, FF_T) * cdiv(h->h, FF_T));
        LAUNCH(h, "ff_init", per_px, dim3(256), k_ff_init, a);
        if (h->opt_ff_union) {   // the two-way edges by union-find: three launches for what took the relaxation a launch per tile border crossed
            const int pairs = ((h->w - 1) / FF_T) * h->h + ((h->h - 1) / FF_T) * h->w;
            LAUNCH(h, "ff_local", tiles, dim3(256), k_ff_tile_uf, a);
            if (pairs > 0) LAUNCH(h, "ff_merge", dim3(cdiv(pairs, 256), nm), dim3(256), k_ff_merge, a);
            LAUNCH(h, "ff_flatten", per_px, dim3(256), k_ff_flatten, a);
        }
    }
    if (fixed_rounds > 0 && !resume) {
        if (fixed_rounds > FF_SLOTS) fixed_rounds = FF_SLOTS;
        for (int it = 0; it < fixed_rounds; it++) LAUNCH(h, "ff_relax", tiles, dim3(256), k_ff_relax, a, it);
        a.gate = a.changed + (fixed_rounds - 1);
    } else {
        for (int round = 0; round < 64; round++) {
            HIPCHK(h, hipMemsetAsync(a.changed, 0, 4, h->cur));
            for (int it = 0; it < 6; it++) LAUNCH(h, "ff_relax", tiles, dim3(256), k_ff_relax, a, 0);
            // the last launch of the batch decides: it re-checks every edge, so "no change" there is the fixpoint
            HIPCHK(h, hipMemsetAsync(a.changed, 0, 4, h->cur));
            LAUNCH(h, "ff_relax", tiles, dim3(256), k_ff_relax, a, 0);
            int changed = 0;
            HIPCHK(h, hipMemcpyAsync(&changed, a.changed, 4, hipMemcpyDeviceToHost, h->cur));
            HIPCHK(h, hipStreamSynchronize(h->cur));
            if (!changed) break;
        }
    }
    LAUNCH(h, "ff_count", per_px, dim3(256), k_ff_count, a);
    LAUNCH(h, "ff_select", per_px, dim3(256), k_ff_select, a);
    LAUNCH(h, "ff_apply", per_px, dim3(256), k_ff_apply, a);
    if (!verdict_by_caller) LAUNCH(h, "ff_verdict", dim3(cdiv(nm, 64)), dim3(64), k_ff_verdict, a, d_unavail);
    return IFX_OK;
}

// computeCompareMap, IF/Core/InstanceFusion.cpp:595-651
static void compare_map(ifx* h, const int* maskBBox, const int* projBBox, const int32_t* class_ids, int nm, std::vector<uint8_t>& unavailable, std::vector<int>& cmp)
{
    for (int m = 0; m < nm; m++) {
        int minX_m = maskBBox[m * 4], maxX_m = maskBBox[m * 4 + 1], minY_m = maskBBox[m * 4 + 2], maxY_m = maskBBox[m * 4 + 3];
        if (maxX_m <= minX_m || maxY_m <= minY_m || unavailable[m]) { unavailable[m] = 1; continue; }
        int best = -1;
        for (int q = 0; q < NI; q++) {
            if (h->inst_class[q] == -1 || class_ids[m] != h->inst_class[q]) continue;
            int minX_i = projBBox[q * 4], maxX_i = projBBox[q * 4 + 1], minY_i = projBBox[q * 4 + 2], maxY_i = projBBox[q * 4 + 3];
            if (maxX_i <= minX_i || maxY_i <= minY_i) continue;
            float IW = (float)(std::min(maxX_i, maxX_m) - std::max(minX_i, minX_m));
            float IH = (float)(std::min(maxY_i, maxY_m) - std::max(minY_i, minY_m));
            if (IW <= 0 || IH <= 0) continue;
            float I = IW * IH;
            float U = (float)(((maxX_i - minX_i) * (maxY_i - minY_i)) + ((maxX_m - minX_m) * (maxY_m - minY_m))) - I;
            if (I / U > 0.5f) best = q;   // last match wins (:641-645)
        }
        if (best > 0) cmp[best + m * NI] = 1;   // instance 0 can never match (:649)
    }
}

static int first_not_used(ifx* h)
{
    for (int i = 0; i < NI; i++) if (h->inst_class[i] == -1) return i;
    return -1;
}

static int run_bboxes(ifx* h, int nm, std::vector<int>& bbox)
{
    LAUNCH(h, "init_bbox", dim3(cdiv((NI + nm) * 4, 256)), dim3(256), k_init_bbox, h->d_bbox, NI + nm, h->w, h->h);
    LAUNCH(h, "project_bbox", dim3(cdiv(h->w, 32), cdiv(h->h, PB_ROWS)), dim3(32, PB_ROWS), k_project_bbox<0>, h->d_state, h->ids_after, (const float4*)h->votes, h->cap, h->d_masks, nm, h->w, h->h,
           h->d_bbox, ifx_idmap(h));
    bbox.resize((size_t)(NI + nm) * 4);
    HIPCHK(h, hipMemcpyAsync(bbox.data(), h->d_bbox, bbox.size() * 4, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}

// ------------------------------------------------------------------ a segmentation call on a spatially sharded map
// Every rank runs the call on the same masks and the same (replicated) id image; what depends on a surfel's votes or position is computed by the
// rank that owns the surfel and merged at three kinds of exchange points (ifx_owner_exchange(h, 200, ...) names the buffers and the operation):
//   1  the projected boxes           partial min / max per instance (maxima negated: one MIN over 32-bit words)
//   2  the model depth under the camera (flood fill input)   disjoint supports: SUM
//   3  per-instance max / sum of the vote counters, when the instance table is full and its twenty weakest entries are evicted
//      (followed by another exchange of kind 1: the boxes after the eviction)
// The mask pipeline (overlap cleaning, superpixels, flood fill), the compare map and the instance table are image-space / host work and run
// replicated; vote updates and the label scan touch owned surfels only.  ifx_owner_segmentation_begin returns 1 while an exchange is pending
// (then: exchange, ifx_owner_segmentation_resume), 0 when the call is complete.  The kNN smoothing (flags & 1) needs every rank's positions and
// is not offered here.
__global__ void k_bbox_flip(int* __restrict__ bbox, int nboxes)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nboxes * 4 && (k & 1)) bbox[k] = -bbox[k];   // {minX, maxX, minY, maxY}: entries 1 and 3 are maxima
}
static int oseg_launch_bboxes(ifx* h)
{
    const int nm = h->oseg_nm;
    LAUNCH(h, "init_bbox", dim3(cdiv((NI + nm) * 4, 256)), dim3(256), k_init_bbox, h->d_bbox, NI + nm, h->w, h->h);
    LAUNCH(h, "project_bbox", dim3(cdiv(h->w, 32), cdiv(h->h, PB_ROWS)), dim3(32, PB_ROWS), k_project_bbox<0>, h->d_state, h->ids_after, (const float4*)h->votes, h->cap, h->d_masks, nm, h->w, h->h,
           h->d_bbox, ifx_idmap(h));
    LAUNCH(h, "bbox_flip", dim3(cdiv((NI + nm) * 4, 256)), dim3(256), k_bbox_flip, h->d_bbox, NI + nm);
    h->oseg_pending = 1;
    return 1;
}
static int oseg_read_bboxes(ifx* h)
{
    const int nm = h->oseg_nm;
    LAUNCH(h, "bbox_flip", dim3(cdiv((NI + nm) * 4, 256)), dim3(256), k_bbox_flip, h->d_bbox, NI + nm);
    h->oseg_bbox.resize((size_t)(NI + nm) * 4);
    HIPCHK(h, hipMemcpyAsync(h->oseg_bbox.data(), h->d_bbox, h->oseg_bbox.size() * 4, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    std::fill(h->oseg_cmp.begin(), h->oseg_cmp.end(), 0);
    compare_map(h, &h->oseg_bbox[NI * 4], &h->oseg_bbox[0], h->oseg_class.data(), nm, h->oseg_unavail, h->oseg_cmp);
    return IFX_OK;
}
// the loop over the masks (step 3) from mask oseg_m on; stops (returns 1) when the table is full and the eviction statistics must be merged
static int oseg_mask_loop(ifx* h, bool after_eviction)
{
    const int nm = h->oseg_nm, P = h->P;
    std::vector<int>& cmp = h->oseg_cmp;
    for (int m = h->oseg_m; m < nm; m++) {
        const bool resumed = after_eviction && m == h->oseg_m;   // this mask asked for room before the eviction: it goes on where the unsharded loop does (no second look at `exist`)
        bool need = resumed;
        if (!resumed) {
            bool exist = false;
            for (int q = 0; q < NI; q++) if (cmp[q + m * NI] == 1) { exist = true; break; }
            need = !exist && !h->oseg_unavail[m];
        }
        if (need) {
            int empty = first_not_used(h);
            if (empty == -1 && !resumed) {
                h->clean_times++;
                HIPCHK(h, hipMemsetAsync(h->d_inst_stats, 0, NI * 2 * 4, h->cur));
                LAUNCH(h, "max_count", dim3(1024), dim3(256), k_max_count, h->d_state, (const float4*)h->votes, h->cap, (const float2*)h->tm, h->d_inst_stats, h->d_inst_stats + NI);
                h->oseg_m = m; h->oseg_state = 3; h->oseg_pending = 3;
                return 1;
            }
            if (empty >= 0) { h->inst_class[empty] = h->oseg_class[m]; cmp[empty + m * NI] = 1; }
        }
        for (int q = 0; q < NI; q++)
            if (cmp[q + m * NI] == 1)
                LAUNCH(h, "vote_update", dim3(cdiv(P, 256)), dim3(256), k_vote_update, h->d_state, h->ids_after, h->d_masks + (size_t)m * P, P, h->cap, q, m + 1, h->votes, ifx_idmap(h));
    }
    // step 4: labels of the owned surfels
    LAUNCH(h, "count_colour", dim3(2048), dim3(256), k_count_colour, h->d_state, (const float4*)h->votes, h->cap, (const float2*)h->tm, (float2*)h->col, h->d_inst_color, h->labels, (const int*)nullptr);
    hipEvent_t eb = ifx_event_get(h);
    hipEventRecord(eb, h->cur);
    h->stage_pending.push_back({2, {h->oseg_ev, eb}});
    h->oseg_ev = nullptr;
    HIPCHK(h, hipStreamSynchronize(h->cur));
    h->oseg_state = 0; h->oseg_pending = 0;
    return 0;
}
extern "C" int ifx_owner_segmentation_begin(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame, int flags)
{
    if (!h || nm < 0 || (nm > 0 && (!masks_in || !class_ids))) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_segmentation_begin: the handle was not created with n_ranks > 1"; return IFX_E_STATE; }
    if (flags & 1) { h->err = "kNN smoothing is not offered on a sharded map (it needs every rank's positions)"; return IFX_E_INVALID; }
    if (h->oseg_state) { h->err = "a segmentation call is already in flight"; return IFX_E_STATE; }
    if (nm > 256) { h->err = "too many masks"; return IFX_E_INVALID; }
    ifx_vlist_reap(h);
    h->seg_counts_valid = 0;
    h->oseg_ev = ifx_event_get(h);
    hipEventRecord(h->oseg_ev, h->cur);
    HIPCHK(h, hipStreamSynchronize(h->cur));
    if (nm == 0) { h->event_pool.push_back(h->oseg_ev); h->oseg_ev = nullptr; return 0; }   // (an empty shard still takes part: the other ranks wait for its boxes)
    const int P = h->P;
    const size_t mbytes = (size_t)nm * P;
    int r = ifx_ensure_masks(h, mbytes);
    if (r) return r;
    h->oseg_nm = nm; h->oseg_m = 0; h->oseg_flags = flags;
    h->oseg_unavail.assign(nm, 0);
    h->oseg_cmp.assign((size_t)nm * NI, 0);
    h->oseg_class.assign(class_ids, class_ids + nm);
    HIPCHK(h, hipMemcpyAsync(h->d_masks_ori, masks_in, mbytes, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->d_masks, h->d_masks_ori, mbytes, hipMemcpyDeviceToDevice, h->cur));
    LAUNCH(h, "mask_clean_overlap", dim3(cdiv(P, 256)), dim3(256), k_mask_clean_overlap, h->d_masks, nm, P);
    if (flags & 2) {
        r = ifx_superpixel_refine(h, rgb, depth, nm, frame);
        if (r) return r;
    }
    r = ifx_owner_ids_begin_impl(h);   // option own_lazy_ids: the frame exchanged the sampled lattice only -- the call's first exchange point is the whole id image's keys
    if (r == 1) { h->oseg_state = 6; h->oseg_pending = 0; return 1; }
    if (r < 0) { h->oseg_state = 0; h->oseg_pending = 0; return r; }
    h->oseg_state = 1;
    r = oseg_launch_bboxes(h);
    if (r < 0) { h->oseg_state = 0; h->oseg_pending = 0; }
    return r;
}
static int oseg_resume(ifx* h);
extern "C" int ifx_owner_segmentation_resume(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    if (!h->oseg_state) { h->err = "no segmentation call in flight"; return IFX_E_STATE; }
    const int r = oseg_resume(h);
    if (r < 0) {   // a failed call is over: the next _begin starts from scratch
        h->oseg_state = 0; h->oseg_pending = 0;
        if (h->oseg_ev) { h->event_pool.push_back(h->oseg_ev); h->oseg_ev = nullptr; }
    }
    return r;
}
static int oseg_resume(ifx* h)
{
    const int nm = h->oseg_nm, P = h->P;
    int r;
    switch (h->oseg_state) {
    case 6:   // the id image's keys merged -> the image, then the boxes
        if ((r = ifx_owner_ids_resume_impl(h))) return r;
        h->oseg_state = 1;
        return oseg_launch_bboxes(h);
    case 1:   // boxes merged -> compare map; model depth of the owned surfels
        if ((r = oseg_read_bboxes(h))) return r;
        LAUNCH(h, "project_depth", dim3(cdiv(P, 256)), dim3(256), k_project_depth, h->d_state, h->ids_after, (const float4*)h->pc, P, 1186, h->d_pdm, ifx_idmap(h));
        h->oseg_state = 2; h->oseg_pending = 2;
        return 1;
    case 2:   // model depth merged -> flood fill (replicated), then the masks
        HIPCHK(h, hipMemcpyAsync(h->d_unavail, h->oseg_unavail.data(), nm, hipMemcpyHostToDevice, h->cur));
        if ((r = mask_geometric_filter_device(h, h->d_pdm, h->d_masks, h->d_masks_ori, nm, h->d_unavail))) return r;
        HIPCHK(h, hipMemcpyAsync(h->oseg_unavail.data(), h->d_unavail, nm, hipMemcpyDeviceToHost, h->cur));
        HIPCHK(h, hipStreamSynchronize(h->cur));
        h->oseg_state = 5;
        return oseg_mask_loop(h, false);
    case 3: {   // eviction statistics merged -> getInstanceTableCleanList (IF/Core/InstanceTable.cpp:185-224), clean, the boxes again
        int stats[NI * 2];
        HIPCHK(h, hipMemcpyAsync(stats, h->d_inst_stats, sizeof(stats), hipMemcpyDeviceToHost, h->cur));
        HIPCHK(h, hipStreamSynchronize(h->cur));
        int* maxv = stats; int* sumv = stats + NI;
        int order[NI], cl[NI];
        for (int i = 0; i < NI; i++) order[i] = i;
        for (int i = 0; i < NI; i++)
            for (int j = i + 1; j < NI; j++)
                if ((float)sumv[j] < (float)sumv[i]) { std::swap(maxv[i], maxv[j]); std::swap(order[i], order[j]); std::swap(sumv[i], sumv[j]); }
        for (int i = 0; i < NI; i++) cl[i] = 0;
        for (int i = 0; i < 20; i++) cl[order[i]] = 1;
        for (int q = 0; q < NI; q++) if (cl[q] == 1) h->inst_class[q] = -1;
        HIPCHK(h, hipMemcpyAsync(h->d_clean_list, cl, sizeof(cl), hipMemcpyHostToDevice, h->cur));
        LAUNCH(h, "clean_table", dim3(1024), dim3(256), k_clean_table, h->d_state, h->votes, h->cap, h->d_clean_list);
        HIPCHK(h, hipStreamSynchronize(h->cur));
        h->oseg_state = 4;
        return oseg_launch_bboxes(h);
    }
    case 4:   // boxes after the eviction merged -> compare map again, the mask that asked for room goes on
        if ((r = oseg_read_bboxes(h))) return r;
        h->oseg_state = 5;
        return oseg_mask_loop(h, true);
    default: break;
    }
    h->err = "ifx_owner_segmentation_resume: bad state";
    return IFX_E_STATE;
}

static int process_segmentation_host(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame, int flags);
static int process_segmentation_device(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame, int flags);
extern "C" int ifx_process_segmentation(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame, int flags)
{
    if (!h || nm < 0 || (nm > 0 && (!masks_in || !class_ids))) return IFX_E_INVALID;
    if (nm > 256) { h->err = "too many masks"; return IFX_E_INVALID; }
    // The next frame's tracker is already queued on the main stream (enqueue_frame, "tracked ahead") and touches nothing this call does: the call's ~60 short
    // launches then go to the handle's third stream (the loop-closure tracker's: four streams is what the runtime's hardware queues hold) and run beside the tracker's 170 instead of behind them.  The call ends with the host waiting for its stream,
    // so whatever the caller enqueues next is ordered behind it as before.
    const bool aside = h->opt_seg_aside && h->stream_c && h->opt_two_streams && h->opt_seg_device && !h->own && h->tracked_ahead == h->tick && h->cur == h->stream &&
                       !h->lc_pending;   // (the third stream is busy with a loop-closure tracker: behind the frame tracker on the main stream is the shorter wait)
    if (aside) {
        if (h->ev_result) HIPCHK(h, hipStreamWaitEvent(h->stream_c, h->ev_result, 0));   // behind the frame the call belongs to
        h->cur = h->stream_c;
    }
    const int r = h->opt_seg_device ? process_segmentation_device(h, rgb, depth, masks_in, class_ids, nm, frame, flags)
                                    : process_segmentation_host(h, rgb, depth, masks_in, class_ids, nm, frame, flags);
    if (aside) { hipStreamSynchronize(h->stream_c); h->cur = h->stream; }
    // a call that failed part-way may have updated votes without the label scan that follows them: the incremental scan of the next call assumes
    // that votes outside its own id image are unchanged since the last scan, so the next call scans everything
    if (r != IFX_OK) h->labels_stale_all = 1;
    return r;
}

// step 3 of processInstance (IF/Core/InstanceFusion.cpp:955-1040) from mask m_start on, driven by the host: registration, the eviction of a full table
// (computeMaxCountInMap, getInstanceTableCleanList, cleanInstanceTableMap, boxes and compare map again), one vote launch per (mask, instance)
static int seg_host_mask_loop(ifx* h, int nm, const int32_t* class_ids, int m_start, std::vector<int>& cmp, std::vector<uint8_t>& unavailable, std::vector<int>& bbox)
{
    const int P = h->P;
    int r;
    for (int m = m_start; m < nm; m++) {
        bool exist = false;
        for (int q = 0; q < NI; q++) if (cmp[q + m * NI] == 1) { exist = true; break; }
        if (!exist && !unavailable[m]) {
            int empty = first_not_used(h);
            if (empty == -1) {
                h->clean_times++;
                HIPCHK(h, hipMemsetAsync(h->d_inst_stats, 0, NI * 2 * 4, h->cur));
                LAUNCH(h, "max_count", dim3(1024), dim3(256), k_max_count, h->d_state, (const float4*)h->votes, h->cap, (const float2*)h->tm, h->d_inst_stats, h->d_inst_stats + NI);
                int stats[NI * 2];
                HIPCHK(h, hipMemcpyAsync(stats, h->d_inst_stats, sizeof(stats), hipMemcpyDeviceToHost, h->cur));
                HIPCHK(h, hipStreamSynchronize(h->cur));
                // getInstanceTableCleanList, IF/Core/InstanceTable.cpp:185-224
                int* maxv = stats; int* sumv = stats + NI;
                int order[NI], cl[NI];
                for (int i = 0; i < NI; i++) order[i] = i;
                for (int i = 0; i < NI; i++)
                    for (int j = i + 1; j < NI; j++)
                        if ((float)sumv[j] < (float)sumv[i]) { std::swap(maxv[i], maxv[j]); std::swap(order[i], order[j]); std::swap(sumv[i], sumv[j]); }
                for (int i = 0; i < NI; i++) cl[i] = 0;
                for (int i = 0; i < 20; i++) cl[order[i]] = 1;
                for (int q = 0; q < NI; q++) if (cl[q] == 1) h->inst_class[q] = -1;
                HIPCHK(h, hipMemcpyAsync(h->d_clean_list, cl, sizeof(cl), hipMemcpyHostToDevice, h->cur));
                LAUNCH(h, "clean_table", dim3(1024), dim3(256), k_clean_table, h->d_state, h->votes, h->cap, h->d_clean_list);
                h->labels_stale_all = 1;   // votes of every surfel that carried an evicted instance changed
                HIPCHK(h, hipStreamSynchronize(h->cur));
                r = run_bboxes(h, nm, bbox);
                if (r) return r;
                std::fill(cmp.begin(), cmp.end(), 0);
                compare_map(h, &bbox[NI * 4], &bbox[0], class_ids, nm, unavailable, cmp);
                empty = first_not_used(h);
            }
            if (empty >= 0) { h->inst_class[empty] = class_ids[m]; cmp[empty + m * NI] = 1; }
        }
        for (int q = 0; q < NI; q++)
            if (cmp[q + m * NI] == 1)
                LAUNCH(h, "vote_update", dim3(cdiv(P, 256)), dim3(256), k_vote_update, h->d_state, h->ids_after, h->d_masks + (size_t)m * P, P, h->cap, q, m + 1, h->votes, ifx_idmap(h));
    }
    return IFX_OK;
}
// step 4: the label scan (countAndColourSurfelMap) -- restricted to what a call can have changed unless the votes were rewritten wholesale
// `gate`: device pointer to (ff_incomplete, evict_at) of the device-side call, or null.  Returns 1 when the scan enqueued was the full one.
// default_done: the caller already ran k_colour_default in this call (it reads nothing a call changes -- a surfel whose votes the call then touches is recoloured by
// k_count_colour_px, which treats the default colour as "none yet" -- so the device-scheduled call enqueues it before the masks are staged)
static int seg_label_scan(ifx* h, const int* gate = nullptr, bool default_done = false)
{
    const int P = h->P;
    if (h->opt_labels_incremental && !h->labels_stale_all) {
        if (!default_done) LAUNCH(h, "colour_default", dim3(2048), dim3(256), k_colour_default, h->d_state, (const float2*)h->tm, (float2*)h->col, h->labels, gate);
        LAUNCH(h, "count_colour_px", dim3(cdiv(P, 256)), dim3(256), k_count_colour_px, h->d_state, h->ids_after, P, (const float4*)h->votes, h->cap, (const float2*)h->tm, (float2*)h->col,
               h->d_inst_color, h->labels, ifx_idmap(h), gate);
        return 0;
    }
    LAUNCH(h, "count_colour", dim3(2048), dim3(256), k_count_colour, h->d_state, (const float4*)h->votes, h->cap, (const float2*)h->tm, (float2*)h->col, h->d_inst_color, h->labels, gate);
    h->labels_stale_all = 0;
    return 1;
}

static int process_segmentation_host(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame, int flags)
{
    ifx_ids_ensure(h);   // the call reads the id image under every mask pixel
    // Only the kNN smoothing looks at surfels the id image does not show: a slot outside the cached view list that has outlived the age rule
    // (ifx_map.hip "View list") is unstable, so it was never in an id image, carries no votes and takes part in nothing else of this call.
    if (flags & 1) ifx_vlist_reap(h);
    h->seg_counts_valid = 0;
    hipEvent_t ea = ifx_event_get(h);
    hipEventRecord(ea, h->cur);
    HIPCHK(h, hipStreamSynchronize(h->cur));
    int n = 0;
    HIPCHK(h, hipMemcpy(&n, &h->d_state->count, sizeof(int), hipMemcpyDeviceToHost));
    if (nm == 0 || n == 0) { h->event_pool.push_back(ea); return IFX_OK; }
    const int P = h->P;
    size_t mbytes = (size_t)nm * P;
    std::vector<uint8_t> unavailable(nm, 0);
    int r = ifx_ensure_masks(h, mbytes);
    if (r) return r;
    // the masks stay on the device from here on: d_masks_ori = "BAK ORI MASK" (:705-706), d_masks = working copy
    HIPCHK(h, hipMemcpyAsync(h->d_masks_ori, masks_in, mbytes, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipMemcpyAsync(h->d_masks, h->d_masks_ori, mbytes, hipMemcpyDeviceToDevice, h->cur));
    // step 0_1
    LAUNCH(h, "mask_clean_overlap", dim3(cdiv(P, 256)), dim3(256), k_mask_clean_overlap, h->d_masks, nm, P);
    // steps -1_1 .. -1_3 (superpixel refinement)
    if (flags & 2) {
        r = ifx_superpixel_refine(h, rgb, depth, nm, frame);
        if (r) return r;
    }
    // steps 1, 2
    std::vector<int> bbox;
    r = run_bboxes(h, nm, bbox);
    if (r) return r;
    std::vector<int> cmp((size_t)nm * NI, 0);
    compare_map(h, &bbox[NI * 4], &bbox[0], class_ids, nm, unavailable, cmp);
    // step 3_0: model depth under the camera, then the flood fill of every usable mask (device)
    LAUNCH(h, "project_depth", dim3(cdiv(P, 256)), dim3(256), k_project_depth, h->d_state, h->ids_after, (const float4*)h->pc, P, 1186, h->d_pdm, ifx_idmap(h));
    HIPCHK(h, hipMemcpyAsync(h->d_unavail, unavailable.data(), nm, hipMemcpyHostToDevice, h->cur));
    r = mask_geometric_filter_device(h, h->d_pdm, h->d_masks, h->d_masks_ori, nm, h->d_unavail);
    if (r) return r;
    HIPCHK(h, hipMemcpyAsync(unavailable.data(), h->d_unavail, nm, hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    // step 3
    r = seg_host_mask_loop(h, nm, class_ids, 0, cmp, unavailable, bbox);
    if (r) return r;
    // step 4
    seg_label_scan(h);
    // flannKnnVoteSurfelMap (isflann, :1051)
    if (flags & 1) { r = ifx_knn_vote(h, nullptr); if (r) return r; }
    hipEvent_t eb = ifx_event_get(h);
    hipEventRecord(eb, h->cur);
    h->stage_pending.push_back({2, {ea, eb}});
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}

// ---- the same call without the host in the middle of it (default; ifx_set_option("seg_device", 0) restores the host-driven schedule above).
// Round 2's call cost 1.4-1.8 ms -- two frames -- of which the kernels were ~0.8: the host waited for the stream at entry (behind the NEXT frame's tracker, enqueued
// ahead), copied the masks from pageable memory, read the superpixel table back for connectSuperPixel, read the boxes back for computeCompareMap, looked at the
// flood fill every 7 relaxations, read the verdicts back, and decided registration mask by mask.  Here: pinned staging, connectSuperPixel on the device
// (ifx_slic.hip: k_sp_connect), compare map and registration in two tiny kernels on a device copy of the instance table, a FIXED schedule of flood-fill
// relaxations (the ones behind the fixpoint return at once), one vote launch per mask that looks its target up on the device -- and ONE synchronisation at the end,
// which brings back the table, the verdicts and two flags: "the fill did not converge inside its schedule" and "the table was full at mask m".  Both are rare and
// are finished the old way from exactly that point (results identical either way: tests/test_gpu_parity.py::test_segmentation_device_schedule_equals_host_schedule).
struct SegCtl { int ff_incomplete, evict_at, nm, pad; int inst_class[NI]; int cls[256]; int best[256]; int target[256]; };
// computeCompareMap (IF/Core/InstanceFusion.cpp:595-651): one thread per mask; the LAST instance over the threshold wins, instance 0 never matches
__global__ void k_seg_compare(SegCtl* __restrict__ s, const int* __restrict__ bbox, uint8_t* __restrict__ unavailable)
{
    __shared__ int s_cls[NI], s_box[NI * 4];
    for (int t = threadIdx.x; t < NI; t += blockDim.x) s_cls[t] = s->inst_class[t];
    for (int t = threadIdx.x; t < NI * 4; t += blockDim.x) s_box[t] = bbox[t];
    __syncthreads();
    // a wave per mask, its lanes over the instances: "the last instance over the threshold" is the largest matching index (one mask per thread walked the
    // 96 instances one after the other: 13 us on the call's chain)
    const int lane = threadIdx.x & 63, nm = s->nm;
    for (int m = threadIdx.x >> 6; m < nm; m += blockDim.x >> 6) {
        const int* mb = bbox + (NI + m) * 4;
        const int minX_m = mb[0], maxX_m = mb[1], minY_m = mb[2], maxY_m = mb[3];
        if (maxX_m <= minX_m || maxY_m <= minY_m || unavailable[m]) {   // (wave-uniform)
            if (lane == 0) { s->target[m] = -1; unavailable[m] = 1; s->best[m] = -1; }
            continue;
        }
        const int cls = s->cls[m];
        int best = -1;
        for (int q = lane; q < NI; q += 64) {
            if (s_cls[q] == -1 || cls != s_cls[q]) continue;
            const int minX_i = s_box[q * 4], maxX_i = s_box[q * 4 + 1], minY_i = s_box[q * 4 + 2], maxY_i = s_box[q * 4 + 3];
            if (maxX_i <= minX_i || maxY_i <= minY_i) continue;
            const float IW = (float)(min(maxX_i, maxX_m) - max(minX_i, minX_m));
            const float IH = (float)(min(maxY_i, maxY_m) - max(minY_i, minY_m));
            if (IW <= 0 || IH <= 0) continue;
            const float I = IW * IH;
            const float U = (float)(((maxX_i - minX_i) * (maxY_i - minY_i)) + ((maxX_m - minX_m) * (maxY_m - minY_m))) - I;
            if (I / U > 0.5f) best = q;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
        if (lane == 0) { s->target[m] = -1; s->best[m] = best > 0 ? best : -1; }
    }
}
// the registration decisions of the mask loop (:955-1010), sequential as in the reference: a mask without a match that is still usable takes the first free
// slot of the table; when there is none the loop stops there (evict_at) and the host evicts (rare: 96 slots).  The table and the per-mask inputs are staged in LDS
// by the whole block; one lane then walks the masks.
// ff_meta / ff_skip (optional): the flood fill's per-mask counters -- the fill's verdict (k_ff_verdict: a mask that kept less than 65 % of its pixels is unusable) is
// taken here, while the per-mask inputs are staged, instead of in a launch of its own.
__global__ void k_seg_register(SegCtl* __restrict__ s, uint8_t* __restrict__ unavailable, const int* __restrict__ ff_gate, const int* __restrict__ ff_meta, const uint8_t* __restrict__ ff_skip)
{
    __shared__ int s_cls[NI], s_best[256], s_mcls[256], s_tgt[256];
    __shared__ unsigned char s_un[256];
    const int nm = s->nm;
    const int gate = ff_gate ? *ff_gate : 0;
    for (int t = threadIdx.x; t < NI; t += blockDim.x) s_cls[t] = s->inst_class[t];
    for (int t = threadIdx.x; t < nm; t += blockDim.x) {
        unsigned char un = unavailable[t];
        if (ff_meta && !gate && !ff_skip[t]) {
            const float finalPoints = (float)ff_meta[t * 32 + 2], oriPoints = (float)ff_meta[t * 32];
            if (finalPoints / oriPoints < 0.65f) { un = 1; unavailable[t] = 1; }
        }
        s_best[t] = s->best[t]; s_mcls[t] = s->cls[t]; s_un[t] = un; s_tgt[t] = -1;
    }
    __syncthreads();
    __shared__ int s_evict;
    if (threadIdx.x == 0) {
        s_evict = -1;
        if (!gate) {
            int next_free = 0;
            for (int m = 0; m < nm; m++) {
                int t = s_best[m];
                if (t < 0 && !s_un[m]) {
                    while (next_free < NI && s_cls[next_free] != -1) next_free++;   // the first free slot: slots only fill up during the loop, so the scan never goes back
                    if (next_free >= NI) { s_evict = m; break; }
                    s_cls[next_free] = s_mcls[m];
                    t = next_free;
                }
                s_tgt[m] = t;
            }
        }
        s->evict_at = s_evict;
        s->ff_incomplete = gate ? 1 : 0;
    }
    __syncthreads();
    if (gate) return;
    for (int t = threadIdx.x; t < NI; t += blockDim.x) s->inst_class[t] = s_cls[t];
    for (int t = threadIdx.x; t < nm; t += blockDim.x) s->target[t] = s_tgt[t];
}
// updateSurfelMapInstance (IF/Core/InstanceFusionCuda.cu:1100-1150) for every mask with the instance the device chose for it; masks without one (or behind the
// point where the host takes over) are passed over.  All masks in one launch: the update is a saturating add of a positive increment, so the masks' updates of a surfel commute (any order ends at
// min(65535, count + sum of increments)) and a pixel can serve every mask it lies in at once
// ONE launch per mask, in mask order (m_only >= 0), as the reference runs updateSurfelMapInstanceKernel once per mask (IF/Core/InstanceFusion.cpp:986-1000): the packed counters are lossy
// while a word's low half is negative (a surfel of the FIRST frame starts with -1 in every word, init_unstable.vert): each read-modify-write of the high half then loses 1 to
// the borrow of the low one, until a vote for the low half's instance makes it non-negative.  How many votes a surfel keeps therefore depends on WHICH MASK comes first --
// within a mask every update goes to the same half with the same weight and commutes, across masks it does not.  All masks in one launch (round 4: 18 us against ~3 us per
// mask here) let the arrival order of the atomics decide: 3 of 373 297 surfels differed from the oracle on a young 640x480 map (tests/test_gpu_sweep.py, round 5).
__global__ void k_vote_update_all(const DevState* __restrict__ st, const int32_t* __restrict__ ids, const uint8_t* __restrict__ masks, int P, int cap, const SegCtl* __restrict__ s, int nm,
                                  float* __restrict__ votes, IdMap im, int m_only)
{
    if (s->ff_incomplete) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int last = s->evict_at >= 0 ? min(nm, s->evict_at) : nm;
    if (m_only >= 0) {
        if (m_only >= last || !(masks[(size_t)m_only * P + k] > 0)) return;
        const int instanceID = s->target[m_only];
        if (instanceID < 0) return;
        const int id = idmap_slot(im, st->count, ids[k]);
        if (id < 0) return;
        const int fi = instanceID / 2, p = instanceID % 2, inc = m_only + 1;
        unsigned int* addr = (unsigned int*)&VOTEF(votes, id, fi);
        unsigned int old = *addr, assumed;
        do {
            assumed = old;
            int a, b;
            vote_decode(__uint_as_float(assumed), a, b);
            if (p == 0) a += inc; else b += inc;
            if (a >= 65535) a = 65535;
            if (b >= 65535) b = 65535;
            old = atomicCAS(addr, assumed, __float_as_uint(vote_encode(a, b)));
        } while (old != assumed);
        return;
    }
    unsigned int in = 0;   // (nm <= 256: eight words would cover it; the masks of a call are few -- handled 32 at a time)
    int id = -2;
    for (int m0 = 0; m0 < last; m0 += 32) {
        in = 0;
        const int mc = min(32, last - m0);
        for (int j = 0; j < mc; j++) in |= (masks[(size_t)(m0 + j) * P + k] > 0 ? 1u : 0u) << j;
        while (in) {
            const int j = __ffs(in) - 1;
            in &= in - 1;
            const int m = m0 + j, instanceID = s->target[m];
            if (instanceID < 0) continue;
            if (id == -2) id = idmap_slot(im, st->count, ids[k]);
            if (id < 0) return;
            const int fi = instanceID / 2, p = instanceID % 2, inc = m + 1;
            unsigned int* addr = (unsigned int*)&VOTEF(votes, id, fi);
            unsigned int old = *addr, assumed;
            do {
                assumed = old;
                int a, b;
                vote_decode(__uint_as_float(assumed), a, b);
                if (p == 0) a += inc; else b += inc;
                if (a >= 65535) a = 65535;
                if (b >= 65535) b = 65535;
                old = atomicCAS(addr, assumed, __float_as_uint(vote_encode(a, b)));
            } while (old != assumed);
        }
    }
}

static int seg_ensure_ctl(ifx* h, size_t mask_bytes)
{
    if (!h->d_segctl) {
        HIPCHK(h, hipMalloc(&h->d_segctl, sizeof(SegCtl)));
        HIPCHK(h, hipHostMalloc((void**)&h->h_segctl, sizeof(SegCtl) + 256, hipHostMallocDefault));
    }
    if (mask_bytes > h->h_masks_cap) {
        if (h->h_masks_stage) hipHostFree(h->h_masks_stage);
        h->h_masks_stage = nullptr; h->h_masks_cap = 0;
        HIPCHK(h, hipHostMalloc((void**)&h->h_masks_stage, mask_bytes, hipHostMallocDefault));
        h->h_masks_cap = mask_bytes;
    }
    return IFX_OK;
}

static int process_segmentation_device(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame, int flags)
{
    static const bool trace = getenv("IFX_SEG_TRACE") != nullptr;   // diagnostic: where the host is inside a call (us since entry, to stderr)
    const auto t_in = std::chrono::steady_clock::now();
    auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count(); };
    double t_res = 0, t_ids = 0, t_stage = 0, t_enq = 0, t_sync = 0;
    if (flags & 1) ifx_vlist_reap(h);
    // the frame's result, not the stream: the next frame's tracker may already be queued behind it, and nothing here has to wait for that
    int n = 0;
    if (h->seg_counts_valid && h->ev_result) { HIPCHK(h, hipEventSynchronize(h->ev_result)); n = h->h_result->count; }
    else { HIPCHK(h, hipStreamSynchronize(h->cur)); HIPCHK(h, hipMemcpy(&n, &h->d_state->count, sizeof(int), hipMemcpyDeviceToHost)); }
    h->seg_counts_valid = 0;
    if (nm == 0 || n == 0) return IFX_OK;
    t_res = us();
    ifx_ids_ensure(h);   // the call reads the id image under every mask pixel: the whole image, if the frame rendered only the sampled lattice
    t_ids = us();
    const int P = h->P;
    const size_t mbytes = (size_t)nm * P;
    int r = ifx_ensure_masks(h, mbytes);
    if (r) return r;
    if ((r = seg_ensure_ctl(h, mbytes))) return r;
    SegCtl* hc = (SegCtl*)h->h_segctl;
    hipEvent_t ea = ifx_event_get(h);
    hipEventRecord(ea, h->cur);
    if (flags & 2) {   // superpixels and their merge: on the queue before the masks are even copied (they need the frame only)
        if ((r = ifx_superpixel_begin(h, rgb, depth))) { h->event_pool.push_back(ea); return r; }
    }
    // what needs the map and the id image but not the masks -- the boxes of the projected instances (the twelve vote float4 under every pixel: the heavy half of
    // getProjectInstanceList / computeProjectBoundingBox) and the model depth under every pixel -- runs while the host copies the masks into pinned memory
    LAUNCH(h, "init_bbox", dim3(cdiv((NI + nm) * 4, 256)), dim3(256), k_init_bbox, h->d_bbox, NI + nm, h->w, h->h);
    LAUNCH(h, "project_bbox_inst", dim3(cdiv(h->w, 32), cdiv(h->h, PB_ROWS)), dim3(32, PB_ROWS), k_project_bbox<1>, h->d_state, h->ids_after, (const float4*)h->votes, h->cap, (const uint8_t*)nullptr, nm,
           h->w, h->h, h->d_bbox, ifx_idmap(h), (const float4*)h->pc, h->d_pdm);
    const bool default_early = h->opt_labels_incremental && !h->labels_stale_all;
    if (default_early) LAUNCH(h, "colour_default", dim3(2048), dim3(256), k_colour_default, h->d_state, (const float2*)h->tm, (float2*)h->col, h->labels, (const int*)nullptr);
    if (flags & 2) { if ((r = ifx_superpixel_filter_prepare(h, nm))) return r; }
    memcpy(h->h_masks_stage, masks_in, mbytes);
    hc->ff_incomplete = 0; hc->evict_at = -1; hc->nm = nm; hc->pad = 0;
    for (int i = 0; i < NI; i++) hc->inst_class[i] = h->inst_class[i];
    for (int m = 0; m < 256; m++) { hc->cls[m] = m < nm ? class_ids[m] : -1; hc->best[m] = -1; hc->target[m] = -1; }
    t_stage = us();
    HIPCHK(h, hipMemcpyAsync(h->d_masks_ori, h->h_masks_stage, mbytes, hipMemcpyHostToDevice, h->cur));   // pinned: a true asynchronous copy
    HIPCHK(h, hipMemcpyAsync(h->d_segctl, hc, sizeof(SegCtl), hipMemcpyHostToDevice, h->cur));
    LAUNCH(h, "mask_clean_overlap", dim3(cdiv(P, 256)), dim3(256), k_mask_clean_overlap_from, (const uint8_t*)h->d_masks_ori, h->d_masks, nm, P, h->d_unavail);
    if (flags & 2) {
        r = ifx_superpixel_filter(h, nm, true);
        if (r) return r;
    }
    SegCtl* dc = (SegCtl*)h->d_segctl;
    LAUNCH(h, "project_bbox_mask", dim3(cdiv(h->w, 32), cdiv(h->h, PB_ROWS)), dim3(32, PB_ROWS), k_project_bbox<2>, h->d_state, h->ids_after, (const float4*)h->votes, h->cap, h->d_masks, nm, h->w, h->h,
           h->d_bbox, ifx_idmap(h));
    LAUNCH(h, "seg_compare", dim3(1), dim3(256), k_seg_compare, dc, (const int*)h->d_bbox, h->d_unavail);
    const int rounds = h->opt_ff_rounds > 0 ? h->opt_ff_rounds : (h->opt_ff_union ? 4 : 24);   // (after the union-find start the first relaxation normally finds the fixpoint: the rest are spares for one-way edges)
    r = mask_geometric_filter_device(h, h->d_pdm, h->d_masks, h->d_masks_ori, nm, h->d_unavail, rounds, false, true);
    if (r) return r;
    const FFArgs fa = ff_args(h, h->d_pdm, h->d_masks, h->d_masks_ori, nm);
    const int* gate = fa.changed + (std::min(rounds, FF_SLOTS) - 1);
    LAUNCH(h, "seg_register", dim3(1), dim3(64), k_seg_register, dc, h->d_unavail, gate, (const int*)fa.meta, fa.skip);
    for (int m_ = h->opt_vote_per_mask ? 0 : -1; m_ < (h->opt_vote_per_mask ? nm : 0); m_++)   // (mask order is part of the result: see k_vote_update_all)
            LAUNCH(h, "vote_update", dim3(cdiv(P, 256)), dim3(256), k_vote_update_all, h->d_state, h->ids_after, (const uint8_t*)h->d_masks, P, h->cap, (const SegCtl*)dc, nm, h->votes, ifx_idmap(h), m_);
    // the scan kernels look at the control block themselves: when the call has to be finished by the host (fill incomplete / table full) they return at once and
    // the ONE scan of the call runs behind the host-driven tail, after every mask and the eviction -- colours are assigned once, so an early scan would be visible
    const int full_scan = seg_label_scan(h, (const int*)dc, default_early);
    uint8_t* h_un = (uint8_t*)h->h_segctl + sizeof(SegCtl);
    HIPCHK(h, hipMemcpyAsync(hc, dc, sizeof(SegCtl), hipMemcpyDeviceToHost, h->cur));
    HIPCHK(h, hipMemcpyAsync(h_un, h->d_unavail, nm, hipMemcpyDeviceToHost, h->cur));
    t_enq = us();
    HIPCHK(h, hipStreamSynchronize(h->cur));
    t_sync = us();
    if (trace) fprintf(stderr, "seg call: result %.0f  ids %.0f  staged %.0f  enqueued %.0f  synced %.0f us (nm %d)\n", t_res, t_ids, t_stage, t_enq, t_sync, nm);
    if ((hc->ff_incomplete || hc->evict_at >= 0) && full_scan) h->labels_stale_all = 1;   // the gated scan did not run: whatever made it a full one still holds
    if (hc->ff_incomplete) {   // the fill needs more relaxations than the schedule holds: finish it with the host looking, then the tail again (nothing was voted yet)
        r = mask_geometric_filter_device(h, h->d_pdm, h->d_masks, h->d_masks_ori, nm, h->d_unavail, 0, true);
        if (r) return r;
        LAUNCH(h, "seg_register", dim3(1), dim3(64), k_seg_register, dc, h->d_unavail, (const int*)nullptr, (const int*)nullptr, (const uint8_t*)nullptr);
        for (int m_ = h->opt_vote_per_mask ? 0 : -1; m_ < (h->opt_vote_per_mask ? nm : 0); m_++)   // (mask order is part of the result: see k_vote_update_all)
            LAUNCH(h, "vote_update", dim3(cdiv(P, 256)), dim3(256), k_vote_update_all, h->d_state, h->ids_after, (const uint8_t*)h->d_masks, P, h->cap, (const SegCtl*)dc, nm, h->votes, ifx_idmap(h), m_);
        const int full2 = seg_label_scan(h, (const int*)dc);   // (gated again: the table may turn out full at some mask)
        HIPCHK(h, hipMemcpyAsync(hc, dc, sizeof(SegCtl), hipMemcpyDeviceToHost, h->cur));
        HIPCHK(h, hipMemcpyAsync(h_un, h->d_unavail, nm, hipMemcpyDeviceToHost, h->cur));
        HIPCHK(h, hipStreamSynchronize(h->cur));
        if (hc->evict_at >= 0 && full2) h->labels_stale_all = 1;
    }
    for (int i = 0; i < NI; i++) h->inst_class[i] = hc->inst_class[i];
    if (hc->evict_at >= 0) {   // the table is full at this mask: the reference evicts its twenty weakest instances and goes on -- from here the host-driven loop
        std::vector<uint8_t> unavailable(h_un, h_un + nm);
        std::vector<int> cmp((size_t)nm * NI, 0), bbox;
        for (int m = 0; m < nm; m++) if (hc->best[m] > 0) cmp[hc->best[m] + m * NI] = 1;
        r = seg_host_mask_loop(h, nm, class_ids, hc->evict_at, cmp, unavailable, bbox);
        if (r) return r;
        seg_label_scan(h);
    }
    if (flags & 1) { r = ifx_knn_vote(h, nullptr); if (r) return r; }
    hipEvent_t eb = ifx_event_get(h);
    hipEventRecord(eb, h->cur);
    h->stage_pending.push_back({2, {ea, eb}});
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}

__global__ void k_gather_labels(const DevState* __restrict__ st, const float2* __restrict__ tm, const int32_t* __restrict__ labels, const int* __restrict__ rank, int cap,
                                int32_t* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || i >= st->count) return;
    if (tm[i].y > DEAD_TIME) out[rank[i]] = labels[i];
}
__global__ void k_alive_flags2(const DevState* __restrict__ st, const float2* __restrict__ tm, int* __restrict__ flags, int cap)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    flags[i] = (i < st->count && tm[i].y > DEAD_TIME) ? 1 : 0;
}

extern "C" int ifx_labels(ifx_t* h, int32_t* out, int max_n)
{
    if (h) ifx_vlist_reap(h);   // whole-map consumer: nothing outside the cached view list may outlive the age rule (ifx_map.hip "View list")
    if (!h || !out) return IFX_E_INVALID;
    // labels of the live surfels in map order: rank = exclusive scan of the alive flags
    LAUNCH(h, "alive_flags", dim3(cdiv(h->cap, 256)), dim3(256), k_alive_flags2, h->d_state, (const float2*)h->tm, h->scan_flags, h->cap);
    ifx_scan_exclusive(h, h->scan_flags, h->cap, h->scan_out, &h->d_state->seg_counts[1]);
    LAUNCH(h, "gather_labels", dim3(cdiv(h->cap, 256)), dim3(256), k_gather_labels, h->d_state, (const float2*)h->tm, h->labels, h->scan_out, h->cap, h->labels2);
    DevState hs;
    HIPCHK(h, hipStreamSynchronize(h->cur));
    HIPCHK(h, hipMemcpy(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost));
    int n = std::min(hs.seg_counts[1], max_n);
    HIPCHK(h, hipMemcpy(out, h->labels2, (size_t)n * 4, hipMemcpyDeviceToHost));
    return n;
}

// ---- instance ground truth (ScanNet evaluation of the reference)
// instanceGT argument of ElasticFusion::processFrame (EF/ElasticFusion.cpp:273-291): surfels created from now on remember the id under their pixel
extern "C" int ifx_set_instance_gt(ifx_t* h, const uint8_t* gt_hw)
{
    if (!h) return IFX_E_INVALID;
    if (!gt_hw) { h->inst_gt_on = 0; return IFX_OK; }
    if (!h->d_inst_gt) HIPCHK(h, hipMalloc(&h->d_inst_gt, (size_t)h->P));
    HIPCHK(h, hipMemcpyAsync(h->d_inst_gt, gt_hw, (size_t)h->P, hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));   // gt_hw is the caller's
    h->inst_gt_on = 1;
    ifx_drop_tracked(h);
    return IFX_OK;
}
// computePrecisionAndRecallKernel, IF/Core/InstanceFusionCuda.cu:2085-2114
__global__ void k_precision_recall(const DevState* __restrict__ st, const float2* __restrict__ col, const float2* __restrict__ tm, const float4* __restrict__ ic,
                                   const float* __restrict__ inst_color, int* __restrict__ inst_num, int* __restrict__ gt_num, int* __restrict__ inst_gt_map)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= st->count || !(tm[s].y > -1.0e8f)) return;   // tombstones are not part of the map
    const float cc = col[s].y;
    int inst = -1;
    for (int i = 0; i < IFX_NUM_INSTANCES; i++) if (cc == inst_color[i]) { inst = i; break; }
    if (inst >= 0) atomicAdd(&inst_num[inst], 1);
    const int gt = (int)ic[s].w;
    if (gt >= 0 && gt < 256) atomicAdd(&gt_num[gt], 1);
    if (inst >= 0 && gt >= 0 && gt < 256) atomicAdd(&inst_gt_map[gt * IFX_NUM_INSTANCES + inst], 1);
}
extern "C" int ifx_precision_recall(ifx_t* h, int32_t* inst_num96, int32_t* gt_num256, int32_t* inst_gt_map)
{
    if (h) ifx_vlist_reap(h);   // whole-map consumer: nothing outside the cached view list may outlive the age rule (ifx_map.hip "View list")
    if (!h || !inst_num96 || !gt_num256 || !inst_gt_map) return IFX_E_INVALID;
    const size_t n = IFX_NUM_INSTANCES + 256 + 256 * IFX_NUM_INSTANCES;
    int* d = nullptr;
    HIPCHK(h, hipMalloc(&d, n * 4));
    hipMemsetAsync(d, 0, n * 4, h->cur);
    hipMemcpyAsync(h->d_inst_color, h->inst_color, sizeof(h->inst_color), hipMemcpyHostToDevice, h->cur);
    LAUNCH(h, "precision_recall", dim3(cdiv(h->cap, 256)), dim3(256), k_precision_recall, (const DevState*)h->d_state, (const float2*)h->col, (const float2*)h->tm,
           (const float4*)h->ic, (const float*)h->d_inst_color, d, d + IFX_NUM_INSTANCES, d + IFX_NUM_INSTANCES + 256);
    std::vector<int32_t> host(n);
    hipError_t e = hipMemcpyAsync(host.data(), d, n * 4, hipMemcpyDeviceToHost, h->cur);
    if (e == hipSuccess) e = hipStreamSynchronize(h->cur);
    hipFree(d);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return IFX_E_HIP; }
    memcpy(inst_num96, host.data(), IFX_NUM_INSTANCES * 4);
    memcpy(gt_num256, host.data() + IFX_NUM_INSTANCES, 256 * 4);
    memcpy(inst_gt_map, host.data() + IFX_NUM_INSTANCES + 256, (size_t)256 * IFX_NUM_INSTANCES * 4);
    return IFX_OK;
}

// renderProjectFrameKernel, IF/Core/InstanceFusionCuda.cu:1432-1498 (InstanceFusion::renderProjectMap without the boxes drawn on the host)
__global__ void k_render_project(const DevState* __restrict__ st, const int32_t* __restrict__ ids, const float2* __restrict__ col, int P, float4* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int id = ids[k];
    float4 c = make_float4(0.f, 0.f, 0.f, 1.f);
    if (id > 0 && id < st->count) {
        const int v = (int)col[id].y;
        c.x = (float)((v >> 16) & 0xFF) / 255.0f; c.y = (float)((v >> 8) & 0xFF) / 255.0f; c.z = (float)(v & 0xFF) / 255.0f;
    }
    out[k] = c;
}
extern "C" int ifx_render_project_map(ifx_t* h, float* out_rgba, float* d_out_rgba)
{
    if (!h || (!out_rgba && !d_out_rgba)) return IFX_E_INVALID;
    ifx_ids_ensure(h);
    float* dst = d_out_rgba;
    if (!dst) {
        if (!h->d_project) HIPCHK(h, hipMalloc(&h->d_project, (size_t)h->P * 16));
        dst = h->d_project;
    }
    LAUNCH(h, "render_project", dim3(cdiv(h->P, 256)), dim3(256), k_render_project, (const DevState*)h->d_state, (const int32_t*)h->ids_after, (const float2*)h->col, h->P, (float4*)dst);
    if (out_rgba) {
        HIPCHK(h, hipMemcpyAsync(out_rgba, dst, (size_t)h->P * 16, hipMemcpyDeviceToHost, h->cur));
        HIPCHK(h, hipStreamSynchronize(h->cur));
    }
    return IFX_OK;
}

extern "C" int ifx_instance_table(ifx_t* h, int32_t* out96)
{
    if (!h || !out96) return IFX_E_INVALID;
    memcpy(out96, h->inst_class, sizeof(h->inst_class));
    return IFX_OK;
}

// getLoopClosureInstanceTable, IF/Core/InstanceTable.cpp:98-121
extern "C" int ifx_loop_closure_instance_table(ifx_t* h, int32_t* out)
{
    if (!h || !out) return IFX_E_INVALID;
    for (int i = 0; i < NI; i++) {
        int c = (int)h->inst_color[i];
        out[5 * i + 0] = (c >> 16) & 0xFF; out[5 * i + 1] = (c >> 8) & 0xFF; out[5 * i + 2] = c & 0xFF;
        out[5 * i + 3] = h->inst_class[i];
        out[5 * i + 4] = i;
    }
    return IFX_OK;
}

// =========================================================================== f-4: 3-D boxes and per-instance point clouds
// InstanceFusion::computeMapBoundingBox / getInstancePointCloud (IF/Core/InstanceFusion.cpp:1261-1590) with their kernels
// (IF/Core/InstanceFusionCuda.cu:1555-2083): the reference's display / export branch.  A surfel belongs to the instance whose table colour
// its instance colour equals (first match); its normal votes for the cells of an 18 x 36 longitude-latitude grid of the unit sphere whose
// centre lies within the cell's vertex-to-centre distance; the cell with most votes over the whole map gives the ground normal, every
// instance's own votes its heading around that normal; boxes are min / max of the integer-scaled coordinates in the ground (bbox_type 1)
// or instance (0) frame.  The reference evaluates the 648 cell centres with cos / sin inside the kernel for every surfel; here the table
// is built once on the host (the same libm the oracle uses, so the votes are comparable count for count) and staged in LDS.
#define BB_SEG 18
#define BB_CELLS (BB_SEG * BB_SEG * 2)
struct BBCell { float x, y, z, R; };

static void bb_cell_table(BBCell* t)
{
    const float pi = 3.1415926f;
    int p = 0;
    for (int i = -BB_SEG / 2; i < BB_SEG / 2; i++) {
        const float theta_v = i * pi / BB_SEG, d_v = cosf(theta_v), y_v = sinf(theta_v);
        const float theta_m = (i + 0.5f) * pi / BB_SEG, d_m = cosf(theta_m), y_m = sinf(theta_m);
        for (int j = 0; j < 2 * BB_SEG; j++) {
            const float alpha_v = j * pi / BB_SEG, x_v = d_v * cosf(alpha_v), z_v = d_v * sinf(alpha_v);
            const float alpha_m = (j + 0.5f) * pi / BB_SEG, x_m = d_m * cosf(alpha_m), z_m = d_m * sinf(alpha_m);
            const float dx = x_v - x_m, dy = y_v - y_m, dz = z_v - z_m;
            t[p].x = x_m; t[p].y = y_m; t[p].z = z_m; t[p].R = sqrtf(dx * dx + dy * dy + dz * dz);
            p++;
        }
    }
}
__device__ __forceinline__ int instance_of_colour(float cc, const float* __restrict__ inst_color)
{
    for (int i = 0; i < IFX_NUM_INSTANCES; i++) if (cc == inst_color[i]) return i;
    return -1;
}
// testAllSurfelNormalVoteKernel, IF/Core/InstanceFusionCuda.cu:1555-1637
__global__ __launch_bounds__(256) void k_normal_vote(const DevState* __restrict__ st, const float4* __restrict__ nr, const float2* __restrict__ col, const float2* __restrict__ tm,
                                                     const float* __restrict__ inst_color, const BBCell* __restrict__ cells, int* __restrict__ ground_vote, int* __restrict__ inst_vote)
{
    __shared__ BBCell s_cell[BB_CELLS];
    __shared__ int s_vote[BB_CELLS];
    __shared__ float s_col[IFX_NUM_INSTANCES];
    for (int k = threadIdx.x; k < BB_CELLS; k += blockDim.x) { s_cell[k] = cells[k]; s_vote[k] = 0; }
    for (int k = threadIdx.x; k < IFX_NUM_INSTANCES; k += blockDim.x) s_col[k] = inst_color[k];
    __syncthreads();
    const int n = st->count;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += blockDim.x * gridDim.x) {
        if (!(tm[id].y > DEAD_TIME)) continue;
        const int inst = instance_of_colour(col[id].y, s_col);
        const float4 n4 = nr[id];
        float nx = n4.x, ny = -n4.y, nz = -n4.z;   // "surfels normal in (Map) is inconsistent with (World)", :1574
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        for (int p = 0; p < BB_CELLS; p++) {
            const float dx = nx - s_cell[p].x, dy = ny - s_cell[p].y, dz = nz - s_cell[p].z;
            if (sqrtf(dx * dx + dy * dy + dz * dz) < s_cell[p].R) {
                atomicAdd(&s_vote[p], 1);
                if (inst != -1) atomicAdd(&inst_vote[BB_CELLS * inst + p], 1);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < BB_CELLS; k += blockDim.x) if (s_vote[k]) atomicAdd(&ground_vote[k], s_vote[k]);
}
// testAllSurfelFindBBoxKernel, IF/Core/InstanceFusionCuda.cu:1831-1901 (including its comparison of the frame coordinates with the WORLD ones)
__global__ __launch_bounds__(256) void k_find_bbox(const DevState* __restrict__ st, const float4* __restrict__ pc, const float2* __restrict__ col, const float2* __restrict__ tm,
                                                   const float* __restrict__ inst_color, float ratio, const float* __restrict__ gc_inv, const float* __restrict__ inst_inv,
                                                   int bbox_type, int* __restrict__ box)
{
    __shared__ float s_col[IFX_NUM_INSTANCES];
    for (int k = threadIdx.x; k < IFX_NUM_INSTANCES; k += blockDim.x) s_col[k] = inst_color[k];
    __syncthreads();
    const int n = st->count;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += blockDim.x * gridDim.x) {
        if (!(tm[id].y > DEAD_TIME)) continue;
        const int inst = instance_of_colour(col[id].y, s_col);
        if (inst == -1) continue;
        const float4 p = pc[id];
        const float* M = bbox_type ? gc_inv : inst_inv + 16 * inst;
        const float gx = M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3] * 1.0f;
        const float gy = M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7] * 1.0f;
        const float gz = M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11] * 1.0f;
        const int vx = f2i_rz(gx * ratio), vy = f2i_rz(gy * ratio), vz = f2i_rz(gz * ratio);
        int minX = vx, minY = vy, minZ = vz, maxX = vx, maxY = vy, maxZ = vz;
        if (minX > p.x * ratio) minX--;
        if (minY > p.y * ratio) minY--;
        if (minZ > p.z * ratio) minZ--;
        if (maxX < p.x * ratio) maxX++;
        if (maxY < p.y * ratio) maxY++;
        if (maxZ < p.z * ratio) maxZ++;
        atomicMin(&box[6 * inst + 0], minX); atomicMin(&box[6 * inst + 2], minY); atomicMin(&box[6 * inst + 4], minZ);
        atomicMax(&box[6 * inst + 1], maxX); atomicMax(&box[6 * inst + 3], maxY); atomicMax(&box[6 * inst + 5], maxZ);
    }
}

// ---- the small per-instance part (setGroundandInstanceCoordinateKernel, :1675-1813: 97 threads in the reference) on the host
static void bb_normalize(float* v) { const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; }
static void bb_cross(const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
static void bb_rodrigues(float angle, const float* v, const float* k, float* r)
{
    const float c = cosf(angle), s = sinf(angle), kv = v[0] * k[0] + v[1] * k[1] + v[2] * k[2];
    const float cr[3] = {v[1] * k[2] - v[2] * k[1], v[2] * k[0] - v[0] * k[2], v[0] * k[1] - v[1] * k[0]};
    for (int q = 0; q < 3; q++) r[q] = c * v[q] + (1 - c) * kv * k[q] + s * cr[q];
}
static void bb_matrix(const float* bx, const float* by, const float* bz, float* m)
{
    m[0] = bx[0]; m[1] = by[0]; m[2] = bz[0]; m[3] = 0;
    m[4] = bx[1]; m[5] = by[1]; m[6] = bz[1]; m[7] = 0;
    m[8] = bx[2]; m[9] = by[2]; m[10] = bz[2]; m[11] = 0;
    m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1;
}
static void bb_frames(const float* ground_normal, const int* inst_vote, float* gc, float* instm)
{
    const float pi = 3.1415926f;
    float by[3] = {ground_normal[0], ground_normal[1], ground_normal[2]}, bx[3], bz[3];
    bb_normalize(by);
    {
        const float setZ[3] = {0, 0, 1};
        bb_cross(setZ, by, bx); bb_normalize(bx);
        bb_cross(by, bx, bz); bb_normalize(bz);
        bb_matrix(bx, by, bz, gc);
    }
    const float step = pi / BB_SEG;
    for (int id = 0; id < IFX_NUM_INSTANCES; id++) {
        const int* nv = inst_vote + id * BB_CELLS;
        int cross_vote[2 * BB_SEG] = {0};
        const float oriZ[3] = {0, 0, -1};
        float oriX[3];
        bb_cross(by, oriZ, oriX); bb_normalize(oriX);
        for (int i = 0; i < BB_CELLS; i++) {
            if (nv[i] == 0) continue;
            const int a = i / (2 * BB_SEG) - BB_SEG / 2, b = i % (2 * BB_SEG) - 1;
            const float theta = (a + 0.5f) * pi / BB_SEG, alpha = (b + 0.5f) * pi / BB_SEG, d = cosf(theta);
            float tz[3] = {d * cosf(alpha), sinf(theta), d * sinf(alpha)};
            bb_normalize(tz);
            const float cs = by[0] * tz[0] + by[1] * tz[1] + by[2] * tz[2];
            if (cs > 0.525f || -cs > 0.525f) continue;
            float best = 999999.9f;
            int best_j = -1;
            for (int j = 0; j < 2 * BB_SEG; j++) {
                float rv[3];
                bb_rodrigues(j * step, oriX, by, rv); bb_normalize(rv);
                const float dx = rv[0] - tz[0], dy = rv[1] - tz[1], dz = rv[2] - tz[2], dist = sqrtf(dx * dx + dy * dy + dz * dz);
                if (dist < best) { best_j = j; best = dist; }
            }
            cross_vote[best_j] += nv[i];
        }
        int vmax = 0, vid = 0;
        for (int i = 0; i < 2 * BB_SEG; i++) if (cross_vote[i] > vmax) { vmax = cross_vote[i]; vid = i; }
        float rv[3];
        bb_rodrigues(vid * step, oriX, by, rv); bb_normalize(rv);
        bb_cross(rv, by, bx); bb_normalize(bx);
        bb_cross(by, bx, bz); bb_normalize(bz);
        bb_matrix(bx, by, bz, instm + 16 * id);
    }
}
// inverse of [B 0; 0 1] with B orthonormal up to rounding: general 3x3 inverse in f64 (Eigen's Matrix4f::inverse in the reference)
static void bb_inverse(const float* m, float* o)
{
    const double a[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
    const double inv[9] = {c00 * id, (a[2] * a[7] - a[1] * a[8]) * id, (a[1] * a[5] - a[2] * a[4]) * id, c01 * id, (a[0] * a[8] - a[2] * a[6]) * id,
                           (a[2] * a[3] - a[0] * a[5]) * id, c02 * id, (a[1] * a[6] - a[0] * a[7]) * id, (a[0] * a[4] - a[1] * a[3]) * id};
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) o[r * 4 + c] = (float)inv[r * 3 + c]; o[r * 4 + 3] = 0; }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

struct BBState { std::vector<float> gc_inv, inst_inv; };
static int bb_compute(ifx* h, float* ground_normal3, float* gc16, float* inst16, int32_t* ground_votes, float* d_inv /* [16 + 96*16] device, out */)
{
    BBCell cells[BB_CELLS];
    bb_cell_table(cells);
    const size_t nv = (size_t)BB_CELLS * (1 + IFX_NUM_INSTANCES);
    int* d_vote = nullptr;
    BBCell* d_cells = nullptr;
    HIPCHK(h, hipMalloc(&d_vote, nv * 4));
    if (hipMalloc(&d_cells, sizeof(cells)) != hipSuccess) { hipFree(d_vote); h->err = "hipMalloc failed"; return IFX_E_HIP; }
    hipMemsetAsync(d_vote, 0, nv * 4, h->cur);
    hipMemcpyAsync(d_cells, cells, sizeof(cells), hipMemcpyHostToDevice, h->cur);
    hipMemcpyAsync(h->d_inst_color, h->inst_color, sizeof(h->inst_color), hipMemcpyHostToDevice, h->cur);
    LAUNCH(h, "normal_vote", dim3(2048), dim3(256), k_normal_vote, (const DevState*)h->d_state, (const float4*)h->nr, (const float2*)h->col, (const float2*)h->tm,
           (const float*)h->d_inst_color, (const BBCell*)d_cells, d_vote, d_vote + BB_CELLS);
    std::vector<int> votes(nv);
    hipError_t e = hipMemcpyAsync(votes.data(), d_vote, nv * 4, hipMemcpyDeviceToHost, h->cur);
    if (e == hipSuccess) e = hipStreamSynchronize(h->cur);
    hipFree(d_vote); hipFree(d_cells);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return IFX_E_HIP; }
    // ground normal: the cell with most votes (IF/Core/InstanceFusion.cpp:1289-1328, including its `- 1` on the longitude index)
    float gn[3] = {0, -1, 0};
    int vmax = -1, vid = -1;
    for (int i = 0; i < BB_CELLS; i++) if (votes[i] > vmax) { vmax = votes[i]; vid = i; }
    if (vid != -1) {
        const float pi = 3.1415926f;
        const int i = vid / (2 * BB_SEG) - BB_SEG / 2, j = vid % (2 * BB_SEG) - 1;
        const float theta = (i + 0.5f) * pi / BB_SEG, alpha = (j + 0.5f) * pi / BB_SEG, d = cosf(theta);
        gn[0] = d * cosf(alpha); gn[1] = sinf(theta); gn[2] = d * sinf(alpha);
        bb_normalize(gn);
    }
    float gc[16], instm[16 * IFX_NUM_INSTANCES], inv[16 + 16 * IFX_NUM_INSTANCES];
    bb_frames(gn, votes.data() + BB_CELLS, gc, instm);
    bb_inverse(gc, inv);
    for (int i = 0; i < IFX_NUM_INSTANCES; i++) bb_inverse(instm + 16 * i, inv + 16 + 16 * i);
    if (ground_normal3) memcpy(ground_normal3, gn, 12);
    if (gc16) memcpy(gc16, gc, 64);
    if (inst16) memcpy(inst16, instm, sizeof(instm));
    if (ground_votes) memcpy(ground_votes, votes.data(), BB_CELLS * 4);
    HIPCHK(h, hipMemcpyAsync(d_inv, inv, sizeof(inv), hipMemcpyHostToDevice, h->cur));
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}

extern "C" int ifx_map_bounding_boxes(ifx_t* h, int bbox_type, float ratio, float* boxes96x6, float* ground_normal3, float* gc_matrix16, float* inst_matrix96x16, int32_t* ground_votes648)
{
    if (!h || !boxes96x6) return IFX_E_INVALID;
    ifx_vlist_reap(h);
    float* d_inv = nullptr;
    int* d_box = nullptr;
    HIPCHK(h, hipMalloc(&d_inv, (16 + 16 * IFX_NUM_INSTANCES) * 4));
    int r = bb_compute(h, ground_normal3, gc_matrix16, inst_matrix96x16, ground_votes648, d_inv);
    if (r) { hipFree(d_inv); return r; }
    if (hipMalloc(&d_box, IFX_NUM_INSTANCES * 6 * 4) != hipSuccess) { hipFree(d_inv); h->err = "hipMalloc failed"; return IFX_E_HIP; }
    int init[IFX_NUM_INSTANCES * 6];
    for (int i = 0; i < IFX_NUM_INSTANCES; i++) { init[i * 6] = init[i * 6 + 2] = init[i * 6 + 4] = 999999999; init[i * 6 + 1] = init[i * 6 + 3] = init[i * 6 + 5] = -999999999; }
    hipMemcpyAsync(d_box, init, sizeof(init), hipMemcpyHostToDevice, h->cur);
    LAUNCH(h, "find_bbox", dim3(2048), dim3(256), k_find_bbox, (const DevState*)h->d_state, (const float4*)h->pc, (const float2*)h->col, (const float2*)h->tm,
           (const float*)h->d_inst_color, ratio, (const float*)d_inv, (const float*)(d_inv + 16), bbox_type ? 1 : 0, d_box);
    hipError_t e = hipMemcpyAsync(init, d_box, sizeof(init), hipMemcpyDeviceToHost, h->cur);
    if (e == hipSuccess) e = hipStreamSynchronize(h->cur);
    hipFree(d_box); hipFree(d_inv);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return IFX_E_HIP; }
    for (int k = 0; k < IFX_NUM_INSTANCES * 6; k++) boxes96x6[k] = init[k] / ratio;   // :1424-1430
    return IFX_OK;
}

// mapCountInstanceByInstColorKernel + getSurfelToInstanceBufferKernel, IF/Core/InstanceFusionCuda.cu:1920-2066: the surfels of one instance
// as records {slot, position in the box frame, normal in the box frame (incl. the reference's homogeneous 1 on the direction), r, g, b}.
// The reference fills its buffers in the order the atomics land; here in slot order (flags, scan, scatter).
__global__ void k_inst_flags(const DevState* __restrict__ st, const float2* __restrict__ col, const float2* __restrict__ tm, const float* __restrict__ inst_color, int want, int cap,
                             int* __restrict__ flags, int* __restrict__ counts)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= cap) return;
    int f = 0;
    if (id < st->count && tm[id].y > DEAD_TIME) {
        const int inst = instance_of_colour(col[id].y, inst_color);
        if (inst != -1) atomicAdd(&counts[inst], 1);
        f = inst == want;
    }
    flags[id] = f;
}
__global__ void k_inst_records(const int* __restrict__ flags, const int* __restrict__ rank, int cap, int max_rec, const float4* __restrict__ pc, const float4* __restrict__ nr,
                               const float2* __restrict__ col, const float* __restrict__ M, float* __restrict__ out)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= cap || !flags[id]) return;
    const int k = rank[id];
    if (k >= max_rec) return;
    const float4 p = pc[id], n4 = nr[id];
    float nx = n4.x, ny = -n4.y, nz = -n4.z;
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    nx = nx / len; ny = ny / len; nz = nz / len;
    float* o = out + (size_t)k * 10;
    o[0] = (float)id;
    o[1] = M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3] * 1.0f;
    o[2] = M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7] * 1.0f;
    o[3] = M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11] * 1.0f;
    o[4] = M[0] * nx + M[1] * ny + M[2] * nz + M[3] * 1.0f;
    o[5] = M[4] * nx + M[5] * ny + M[6] * nz + M[7] * 1.0f;
    o[6] = M[8] * nx + M[9] * ny + M[10] * nz + M[11] * 1.0f;
    const int c = f2i_rz(col[id].x);
    o[7] = (float)(c >> 16 & 0xFF) / 255.0f; o[8] = (float)(c >> 8 & 0xFF) / 255.0f; o[9] = (float)(c & 0xFF) / 255.0f;
}
extern "C" int ifx_instance_point_cloud(ifx_t* h, int bbox_type, int32_t* counts96, int inst, float* out10, int max_records)
{
    if (!h || !counts96 || inst < -1 || inst >= IFX_NUM_INSTANCES || (inst >= 0 && (!out10 || max_records <= 0))) return IFX_E_INVALID;
    ifx_vlist_reap(h);
    float* d_inv = nullptr;
    HIPCHK(h, hipMalloc(&d_inv, (16 + 16 * IFX_NUM_INSTANCES) * 4));
    int r = bb_compute(h, nullptr, nullptr, nullptr, nullptr, d_inv);
    if (r) { hipFree(d_inv); return r; }
    int* d_cnt = h->d_inst_stats;
    hipMemsetAsync(d_cnt, 0, IFX_NUM_INSTANCES * 4, h->cur);
    const int n = h->cap;
    LAUNCH(h, "inst_flags", dim3(cdiv(n, 256)), dim3(256), k_inst_flags, (const DevState*)h->d_state, (const float2*)h->col, (const float2*)h->tm, (const float*)h->d_inst_color, inst, n,
           h->scan_flags, d_cnt);
    int written = 0;
    float* d_out = nullptr;
    if (inst >= 0) {
        ifx_scan_exclusive(h, h->scan_flags, n, h->scan_out, &h->d_state->seg_counts[1]);
        if (hipMalloc(&d_out, (size_t)max_records * 40) != hipSuccess) { hipFree(d_inv); h->err = "hipMalloc failed"; return IFX_E_HIP; }
        LAUNCH(h, "inst_records", dim3(cdiv(n, 256)), dim3(256), k_inst_records, (const int*)h->scan_flags, (const int*)h->scan_out, n, max_records, (const float4*)h->pc,
               (const float4*)h->nr, (const float2*)h->col, (const float*)(bbox_type ? d_inv : d_inv + 16 + 16 * inst), d_out);
    }
    hipError_t e = hipMemcpyAsync(counts96, d_cnt, IFX_NUM_INSTANCES * 4, hipMemcpyDeviceToHost, h->cur);
    if (e == hipSuccess) e = hipStreamSynchronize(h->cur);
    if (e == hipSuccess && inst >= 0) {
        written = std::min(counts96[inst], max_records);
        if (written > 0) e = hipMemcpy(out10, d_out, (size_t)written * 40, hipMemcpyDeviceToHost);
    }
    hipFree(d_out); hipFree(d_inv);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return IFX_E_HIP; }
    return written;
}
