// ifx_map.hip -- surfel-map half of the path as HIP kernels for gfx950.
//
// The reference runs these stages as OpenGL transform-feedback / rasterisation passes over a
// 256-B-per-surfel AoS vertex buffer (EF/GlobalModel.cpp, EF/IndexMap.cpp, EF/Shaders/*).  Here:
//  * the map is struct-of-arrays in HBM (pos+conf 16 B, normal+radius 16 B, colours 8 B, times 8 B,
//    imgCorr 16 B, votes 12 x 16 B planar), so a projection pass streams 24-40 B per surfel with
//    fully coalesced 8/16-B lane loads instead of 256 B;
//  * rasterisation is a 64-bit atomicMin on (depth bits << 32 | surfel id) per covered pixel into an
//    L2-resident key image, resolved by a per-pixel pass (nearest z wins, ties -> lowest id);
//  * fusion updates matched surfels in place (first pixel in column-major order owns a surfel);
//  * deletion leaves tombstones (conf = -1, lastTime = -1e9) and ifx_compact removes them with an
//    order-preserving scan + scatter when asked or when too many accumulate, instead of moving the
//    whole map through a second buffer every frame.
#include <climits>
#include "ifx_ctx.h"
#include <string.h>

#define DEAD_TIME (-1.0e9f)
#define ASSOC_NONE 0xFFFFFFFFu
#define ASSOC_NEW 0xFFFFFFFEu
#define MAP_THREADS 256
#ifndef MAP_BLOCKS
#define MAP_BLOCKS 2048
#endif
#ifndef LIST_BLOCKS
#define LIST_BLOCKS 1024   // grid of the work-list consumers (k_raster_list, k_clean_list, k_tile_count / fill): a multiple of the list segments
#endif

// ------------------------------------------------------------------ shared GLSL helpers
// EF/Shaders/surfels.glsl:19-34
__device__ inline float get_radius(float depth, float norm_z, float inv_fx, float inv_fy)
{
    float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}
// EF/Shaders/surfels.glsl:36-46
__device__ inline float confidence_fn(float x, float y, float cx, float cy, float weighting)
{
    const float maxRadDist = 400, twoSigmaSquared = 0.72f;
    float dx = x - cx, dy = y - cy;
    float radialDist = sqrtf(dx * dx + dy * dy) / maxRadDist;
    return ifx_expf((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}
__device__ inline float tex_f(const float* img, int w, int h, int x, int y) { return img[clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)]; }
// geometry.glsl:21-25
__device__ inline v3 get_vertex_f(const float* depth, int w, int h, int px, int py, float x, float y, float cx, float cy, float ifx_, float ify_)
{
    float z = tex_f(depth, w, h, px, py);
    return v3m((x - cx) * z * ifx_, (y - cy) * z * ify_, z);
}
// geometry.glsl:28-40
__device__ inline v3 get_normal_f(const float* depth, int w, int h, int px, int py, float x, float y, v3 vp, float cx, float cy, float ifx_, float ify_)
{
    v3 xf = get_vertex_f(depth, w, h, px + 1, py, x + 1, y, cx, cy, ifx_, ify_);
    v3 xb = get_vertex_f(depth, w, h, px - 1, py, x - 1, y, cx, cy, ifx_, ify_);
    v3 yf = get_vertex_f(depth, w, h, px, py + 1, x, y + 1, cx, cy, ifx_, ify_);
    v3 yb = get_vertex_f(depth, w, h, px, py - 1, x, y - 1, cx, cy, ifx_, ify_);
    v3 del_x = ((xb + vp) * 0.5f) - ((xf + vp) * 0.5f);
    v3 del_y = ((yb + vp) * 0.5f) - ((yf + vp) * 0.5f);
    return normalized(cross(del_x, del_y));
}

// Work lists are kept in LIST_SEGS independent segments, each with its own length counter on a cache line of its own: a block appends the survivors
// of chunk c to segment c % LIST_SEGS.  One returning atomic per chunk on a SINGLE counter costs ~8 ns each once ~1300 blocks hit it at the same
// time (doubling the chunks per launch slowed the cull kernels from 42 to 52 us); spread over 8 addresses they no longer queue up: 42 -> 34 us.
// A consumer block works on segment blockIdx % LIST_SEGS (the segments hold interleaved chunks, so they are equally long up to one chunk).
#define LIST_SEGS IFX_LIST_SEGS
#define LIST_CTR_STRIDE IFX_LIST_CTR_STRIDE   // uints between counters: 128 B
struct Cam { float fx, fy, cx, cy; int w, h; float maxDepth, conf; int timeDelta; int srank, sn; unsigned int seg_cap; unsigned int* lctr; const uint32_t* seq; int own_n, own_rank; const int* first_live; int raw_slots; int age_epoch; };   // age_epoch: first clean pass that saw the store as it is (age_rule_gone; fills the struct's tail padding)   // seq / own_n / own_rank: spatially sharded map (this handle stores the surfels it owns; ids in keys and images are creation numbers)   // srank / sn: this rank's slice of the slots in the projection passes (sharded mode); seg_cap / lctr: capacity of one list segment, the counters [3 lists][LIST_SEGS]
// Loads of the surfel store by the passes that touch a surfel ONCE per frame (the scan, the list walkers' gathers): with IFX_NT they carry the non-temporal hint, so that the
// ~100-450 MB of lines they pull through per frame do not evict the tracker's working set (pyramids, prediction: tens of MB) from L2 / the Infinity Cache.
#ifdef IFX_NT
typedef float ifx_f4v __attribute__((ext_vector_type(4)));
typedef float ifx_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_once(const float4* p) { const ifx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ifx_f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float2 ld_once(const float2* p) { const ifx_f2v v = __builtin_nontemporal_load(reinterpret_cast<const ifx_f2v*>(p)); return make_float2(v.x, v.y); }
#else
template <typename T> __device__ __forceinline__ T ld_once(const T* p) { return *p; }
#endif
__device__ __forceinline__ unsigned int* list_ctr(const Cam& c, int list, int seg) { return c.lctr + (list * LIST_SEGS + seg) * LIST_CTR_STRIDE; }
// Spatially sharded map: the id a surfel carries in keys / id images is its creation number (the same on every rank; ascending in slot order,
// so "lowest id wins" is the single-GPU tie-break), and a rank finds the slot of an id it owns by binary search -- -1: another rank's surfel.
// Unsharded map: the id is the slot -- except that id 0 means "no surfel" in every image (index_map.vert:51, `surfel_id > 0` in the instance kernels: the reference's surfel 0
// occludes like any other but can never be associated, counted in a clean window, voted for).  The reference compacts every frame, so "surfel 0" is always the FIRST LIVE
// surfel of the map; with tombstones the first live surfel may sit at a later slot: it is drawn as id 0 (DevState::first_live, kept by k_append_scan / k_vlist_offsets /
// the compaction), and id 0 resolves back to that slot.  With compact_every_frame first_live is always 0: the identity.
// fl = DevState::first_live, read ONCE at the top of every kernel that names surfels (FIRST_LIVE(c): before the kernel's first store, where a uniform load goes through
// the scalar cache).  As `*c.first_live` at the point of use -- behind stores and atomics -- it compiled to a vector load of one address by every wave: ~2 ns each at the
// one L2 channel that holds the line, 11 us of a 300 k-thread launch (k_splat_resolve 40 -> 29 us, profiles/r05_*).
// Spatially sharded map: ids are creation numbers, and the reference's "surfel 0" is the live surfel with the LOWEST creation number on any rank.  Every rank publishes
// the lowest live creation number of its shard (k_own_first_live, behind the two places a frame removes surfels: the view-list scan of phase 0, the clean of phase 4), the
// word sits right behind the key images of exchanges 0 / 2 and 4 and is MIN-reduced WITH them (ifx_owner_exchange: one word more in the same collective), and
// Cam::first_live points at it there: fl is then a creation number, and the resolves turn that id into 0 where they write id images (k_index_resolve,
// splat_resolve_body) -- attributes and occlusion are untouched, as on one GPU.
#define FIRST_LIVE(c) (*(c).first_live)
__device__ __forceinline__ void own_first_live(const DevState* __restrict__ st, const float2* __restrict__ tm, const uint32_t* __restrict__ seq, unsigned long long* __restrict__ out)
{
    int f = st->first_live;
    const int n = st->count;
    while (f < n && !(tm[f].y > DEAD_TIME)) f++;
    *out = f < n ? (unsigned long long)seq[f] : ~0ull;   // (no live surfel here: the identity of the MIN; as an int -1, which no creation number reaches -- k_append_scan stops at 0xFFF00000)
}
__global__ void k_own_first_live(const DevState* __restrict__ st, const float2* __restrict__ tm, const uint32_t* __restrict__ seq, unsigned long long* __restrict__ out) { own_first_live(st, tm, seq, out); }

// ---- "hot" records (round 5, option hot_records): position + confidence, normal + radius and times of a slot in ONE 64-byte record.  The store stays struct-of-arrays (the
// scans stream 24 B per slot, the C API hands out the arrays), but at random map order every field a list walker or a resolve GATHERS is a 128-byte line fetch of its own:
// three for the clean + raster walk (322 MB of HBM traffic for 1.1 M entries, profiles/r05_an_pmc_traffic.json), two to four for the resolves and the fusion update.  The
// frame path's gathers read the copy -- one line per surfel -- and its three writers (fusion update, tombstones of the walk and of the scan's age rule, append) write both;
// anything else that writes the store (upload, compaction, deformation, the per-pass paths, a sharded map's phases, ifx_map_view handing out pointers) just marks the copy
// stale on the host (ifx::hot_valid) and the next frame rebuilds it in one streaming launch.
struct alignas(64) Hot { float4 pc, nr; float2 tm; float2 pad0; float4 pad1; };
static_assert(sizeof(Hot) == 64, "one record per 64 bytes");
__global__ void k_hot_rebuild(const DevState* __restrict__ st, const float4* __restrict__ pc, const float4* __restrict__ nr, const float2* __restrict__ tm, Hot* __restrict__ hot)
{
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) { Hot r; r.pc = pc[i]; r.nr = nr[i]; r.tm = tm[i]; r.pad0 = make_float2(0.f, 0.f); r.pad1 = make_float4(0.f, 0.f, 0.f, 0.f); hot[i] = r; }
}
// option hot_verify (debug): the copy a frame is about to trust against the store; a slot that differs was written through a pointer the caller kept past the next frame call
// (include/ifx_c_api.h, ifx_map_view) -- counted in DevState::hot_stale and repaired, so that one violation is reported once and does not snowball
__global__ void k_hot_verify(DevState* st, const float4* __restrict__ pc, const float4* __restrict__ nr, const float2* __restrict__ tm, Hot* __restrict__ hot)
{
    const int n = st->count;
    int bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const float4 a = pc[i], b = nr[i]; const float2 t = tm[i];
        const Hot r = hot[i];
        const bool same = __float_as_uint(a.x) == __float_as_uint(r.pc.x) && __float_as_uint(a.y) == __float_as_uint(r.pc.y) && __float_as_uint(a.z) == __float_as_uint(r.pc.z) &&
                          __float_as_uint(a.w) == __float_as_uint(r.pc.w) && __float_as_uint(b.x) == __float_as_uint(r.nr.x) && __float_as_uint(b.y) == __float_as_uint(r.nr.y) &&
                          __float_as_uint(b.z) == __float_as_uint(r.nr.z) && __float_as_uint(b.w) == __float_as_uint(r.nr.w) && __float_as_uint(t.x) == __float_as_uint(r.tm.x) &&
                          __float_as_uint(t.y) == __float_as_uint(r.tm.y);
        if (!same) { bad++; Hot q = r; q.pc = a; q.nr = b; q.tm = t; hot[i] = q; }
    }
    bad = wave_sum_i(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&st->hot_stale, bad);
}
__device__ __forceinline__ unsigned int key_id(const Cam& c, unsigned int i, int fl) { return c.own_n > 0 ? c.seq[i] : ((c.raw_slots || i != (unsigned int)fl) ? i : 0u); }
__device__ __forceinline__ int local_slot(const Cam& c, int count, unsigned int id, int fl)
{
    if (c.own_n <= 0) return id == 0u ? fl : (int)id;
    int lo = 0, hi = count - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned int v = c.seq[mid];
        if (v == id) return mid;
        if (v < id) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
// Sharded map, frame path: the projections of a rank draw SLOTS into their key images, as on one GPU (no gather of the creation number per drawn surfel);
// before the keys travel k_own_translate swaps the slot of every pixel's local winner for its creation number and keeps the slot in an image of its own.
// After the exchange the rank owns a pixel's winner exactly when its local winner carries the winning creation number -- one gather instead of the
// binary search over the shard's creation numbers (23 dependent loads at 5 M slots: k_associate 80 -> 14 us, k_index_resolve 29 -> 9, k_splat_resolve 39 -> 22).
__device__ __forceinline__ int own_slot_of(const Cam& c, int slot, unsigned int id) { return (slot >= 0 && c.seq[slot] == id) ? slot : -1; }
__global__ void k_own_translate(unsigned long long* __restrict__ keys, int P, const uint32_t* __restrict__ seq, int32_t* __restrict__ slot_img,
                                const DevState* __restrict__ st = nullptr, const float2* __restrict__ tm = nullptr, unsigned long long* __restrict__ gfl_out = nullptr)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && gfl_out) own_first_live(st, tm, seq, gfl_out);   // (the word that travels with these keys)
    if (k >= P) return;
    const unsigned long long key = keys[k];
    if (key == IFX_KEY_EMPTY) { slot_img[k] = -1; return; }
    const unsigned int slot = (unsigned int)(key & 0xFFFFFFFFull);
    slot_img[k] = (int32_t)slot;
    keys[k] = (key & 0xFFFFFFFF00000000ull) | (unsigned long long)seq[slot];
}
// ---- option own_lazy_ids: the id keys of the sampled lattice travel packed behind key_splat.  ONE workgroup each: the packed run [0, L] overlaps the pixels it is
// gathered from / scattered to, so every source is read before the first destination is written (registers across a __syncthreads).
#define OWN_LAT_DS 10
#define OWN_LAT_PER 16
int ifx_own_lattice(const ifx* h) { return cdiv(h->w, OWN_LAT_DS) * cdiv(h->h, OWN_LAT_DS); }
// key_ids[pixel of lattice point j] -> dst[j], the "surfel 0" word -> dst[L]   (dst = key_ids itself, or a staging array)
__global__ __launch_bounds__(1024) void k_own_ids_pack(const unsigned long long* __restrict__ key_ids, int w, int hh, const unsigned long long* __restrict__ word, unsigned long long* dst)
{
    const int lw = (w + OWN_LAT_DS - 1) / OWN_LAT_DS, L = lw * ((hh + OWN_LAT_DS - 1) / OWN_LAT_DS);
    unsigned long long v[OWN_LAT_PER];
#pragma unroll
    for (int u = 0; u < OWN_LAT_PER; u++) {
        const int j = threadIdx.x + u * 1024;
        v[u] = IFX_KEY_EMPTY;
        if (j < L) v[u] = key_ids[(size_t)((j / lw) * OWN_LAT_DS) * w + (j % lw) * OWN_LAT_DS];
        else if (j == L) v[u] = *word;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < OWN_LAT_PER; u++) {
        const int j = threadIdx.x + u * 1024;
        if (j <= L) dst[j] = v[u];
    }
}
// the reverse, behind the exchange: key_ids[0 .. L] -> the lattice pixels (everything else of the run: empty again), the word -> its place behind key_ids
__global__ __launch_bounds__(1024) void k_own_ids_unpack(unsigned long long* key_ids, int w, int hh, unsigned long long* __restrict__ word)
{
    const int lw = (w + OWN_LAT_DS - 1) / OWN_LAT_DS, L = lw * ((hh + OWN_LAT_DS - 1) / OWN_LAT_DS);
    unsigned long long v[OWN_LAT_PER];
#pragma unroll
    for (int u = 0; u < OWN_LAT_PER; u++) {
        const int j = threadIdx.x + u * 1024;
        v[u] = IFX_KEY_EMPTY;
        if (j <= L) { v[u] = key_ids[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < OWN_LAT_PER; u++) {
        const int j = threadIdx.x + u * 1024;
        if (j <= L) key_ids[j] = IFX_KEY_EMPTY;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < OWN_LAT_PER; u++) {
        const int j = threadIdx.x + u * 1024;
        if (j < L) key_ids[(size_t)((j / lw) * OWN_LAT_DS) * w + (j % lw) * OWN_LAT_DS] = v[u];
        else if (j == L) *word = v[u];
    }
}
// the whole id image of a sharded map from its exchanged keys (ifx_owner_ids_resume): creation numbers, the reference's "surfel 0" as 0 (as the frame's own resolve writes them)
__global__ void k_own_ids_resolve(unsigned long long* __restrict__ keys, int P, int32_t* __restrict__ ids, const int* __restrict__ first_live)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const unsigned long long key = keys[k];
    keys[k] = IFX_KEY_EMPTY;
    int id = (key == IFX_KEY_EMPTY) ? 0 : (int32_t)(key & 0xFFFFFFFFull);
    if (id == *first_live) id = 0;
    ids[k] = id;
}
static Cam make_cam(ifx* h)
{
    Cam c;
    c.fx = h->cfg.fx; c.fy = h->cfg.fy; c.cx = h->cfg.cx; c.cy = h->cfg.cy; c.w = h->w; c.h = h->h;
    c.maxDepth = h->cfg.max_depth_processed; c.conf = h->cfg.confidence; c.timeDelta = h->cfg.time_delta;
    c.srank = h->shard_rank; c.sn = h->shard_n > 0 ? h->shard_n : 1;
    c.seg_cap = h->list_seg_cap; c.lctr = h->d_list_ctr;
    c.seq = h->seq; c.own_n = h->own ? h->own_g : 0; c.own_rank = h->own ? h->cfg.rank : 0;
    c.first_live = (h->own && h->gfl_splat) ? (const int*)h->gfl_splat : &h->d_state->first_live; c.raw_slots = 0;   // (sharded map: the low word of the reduced lowest live creation number)
    c.age_epoch = h->age_epoch;
    return c;
}

// ------------------------------------------------------------------ exclusive scan of int flags
#define SCAN_ITEMS 8
#define SCAN_TILE (MAP_THREADS * SCAN_ITEMS)
__global__ __launch_bounds__(MAP_THREADS) void k_scan_reduce(const int* __restrict__ flags, int n, int* __restrict__ block_sums)
{
    __shared__ int lds[MAP_THREADS / 64];
    int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += flags[base + k];
    s = wave_sum_i(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < MAP_THREADS / 64; k++) t += lds[k]; block_sums[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_scan_sums(int* __restrict__ block_sums, int nb, int* __restrict__ total)
{
    __shared__ int lds[1024];
    int per = (nb + 1023) / 1024, lo = threadIdx.x * per, s = 0;
    for (int k = 0; k < per; k++) if (lo + k < nb) s += block_sums[lo + k];
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (threadIdx.x >= off) ? lds[threadIdx.x - off] : 0;
        __syncthreads();
        lds[threadIdx.x] += v;
        __syncthreads();
    }
    int run = lds[threadIdx.x] - s;
    for (int k = 0; k < per; k++) if (lo + k < nb) { int v = block_sums[lo + k]; block_sums[lo + k] = run; run += v; }
    if (threadIdx.x == 1023 && total) *total = lds[1023];
}
__global__ __launch_bounds__(MAP_THREADS) void k_scan_final(const int* __restrict__ flags, int n, const int* __restrict__ block_sums, int* __restrict__ out)
{
    __shared__ int lds[MAP_THREADS];
    int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, s = 0;
    int f[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { f[k] = (base + k < n) ? flags[base + k] : 0; s += f[k]; }
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < MAP_THREADS; off <<= 1) {
        int v = (threadIdx.x >= off) ? lds[threadIdx.x - off] : 0;
        __syncthreads();
        lds[threadIdx.x] += v;
        __syncthreads();
    }
    int run = lds[threadIdx.x] - s + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) { out[base + k] = run; run += f[k]; }
}
int ifx_scan_exclusive(ifx* h, const int* d_flags, int n, int* d_out, int* d_total)
{
    if (n <= 0) { if (d_total) hipMemsetAsync(d_total, 0, 4, h->cur); return IFX_OK; }
    int nb = cdiv(n, SCAN_TILE);
    LAUNCH(h, "scan_reduce", dim3(nb), dim3(MAP_THREADS), k_scan_reduce, d_flags, n, h->scan_block);
    LAUNCH(h, "scan_sums", dim3(1), dim3(1024), k_scan_sums, h->scan_block, nb, d_total);
    LAUNCH(h, "scan_final", dim3(nb), dim3(MAP_THREADS), k_scan_final, d_flags, n, h->scan_block, d_out);
    return IFX_OK;
}

// ------------------------------------------------------------------ first frame (a15)
// vertex_feedback.vert:41-74 + init_unstable.vert:45-67, intended dense initialisation,
// column-major pixel order (EF/GlobalModel.cpp:103-112).
__global__ void k_init_flags(const float* __restrict__ dm, const float* __restrict__ dmf, Cam c, int* __restrict__ flags)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c.w * c.h) return;
    int i = k / c.h, j = k - i * c.h;   // k is the column-major order index
    float z = dm[j * c.w + i], zf = dmf[j * c.w + i];
    flags[k] = !(z <= 0 || z > c.maxDepth || zf <= 0 || zf > c.maxDepth);
}
__global__ void k_init_scatter(DevState* st, const float* __restrict__ dm, const float* __restrict__ dmf, const uint8_t* __restrict__ rgb, Cam c, int tick,
                               const int* __restrict__ flags, const int* __restrict__ rank, int cap, float4* __restrict__ pc, float4* __restrict__ nr,
                               float2* __restrict__ col, float2* __restrict__ tm, float4* __restrict__ ic, float4* __restrict__ votes, uint32_t* __restrict__ seq)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c.w * c.h || !flags[k]) return;
    int n = rank[k];
    if (n >= cap) { st->overflow = 1; return; }
    int i = k / c.h, j = k - i * c.h;
    float ifx_ = 1.0f / c.fx, ify_ = 1.0f / c.fy;
    float x = (float)i + 0.5f, y = (float)j + 0.5f;
    v3 vp = get_vertex_f(dm, c.w, c.h, i, j, x, y, c.cx, c.cy, ifx_, ify_);
    v3 vpf = get_vertex_f(dmf, c.w, c.h, i, j, x, y, c.cx, c.cy, ifx_, ify_);
    v3 nl = get_normal_f(dmf, c.w, c.h, i, j, x, y, vpf, c.cx, c.cy, ifx_, ify_);
    pc[n] = make_float4(vp.x, vp.y, vp.z, confidence_fn(x, y, c.cx, c.cy, 1.0f));
    const uint8_t* cc = &rgb[(j * c.w + i) * 3];
    col[n] = make_float2(encode_color(cc[0] / 255.0f, cc[1] / 255.0f, cc[2] / 255.0f), 0.f);
    tm[n] = make_float2(1.f, (float)tick);
    const float rad = get_radius(vpf.z, nl.z, ifx_, ify_);
    nr[n] = make_float4(nl.x, nl.y, nl.z, rad);
    if (rad > 0.f && rad < 1e30f && __float_as_uint(rad) > st->r_max_bits) atomicMax(&st->r_max_bits, __float_as_uint(rad));
    ic[n] = make_float4(-1.f, -1.f, -1.f, -1.f);
    for (int q = 0; q < 12; q++) VOTE4(votes, n, q) = make_float4(-1.f, -1.f, -1.f, -1.f);
    seq[n] = (uint32_t)n;
}
__global__ void k_init_count(DevState* st, const int* total, int cap)
{
    if (threadIdx.x == 0) { int t = *total; st->count = t < cap ? t : cap; st->n_dead = 0; st->n_new = st->count; st->vl_valid = 0; st->next_seq = (unsigned int)st->count; }
}

int ifx_map_init_first(ifx* h)
{
    h->hot_valid = 0;
    Cam c = make_cam(h);
    LAUNCH(h, "init_flags", dim3(cdiv(h->P, 256)), dim3(256), k_init_flags, h->dm, h->dmf, c, h->scan_flags);
    ifx_scan_exclusive(h, h->scan_flags, h->P, h->scan_out, &h->d_state->seg_counts[0]);
    LAUNCH(h, "init_scatter", dim3(cdiv(h->P, 256)), dim3(256), k_init_scatter, h->d_state, h->dm, h->dmf, h->rgb, c, h->tick, h->scan_flags, h->scan_out, h->cap,
           (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, (float4*)h->ic, (float4*)h->votes, h->seq);
    LAUNCH(h, "init_count", dim3(1), dim3(64), k_init_count, h->d_state, &h->d_state->seg_counts[0], h->cap);
    return IFX_OK;
}

// 64-bit atomicMin on a key image.  (A plain read first to skip atomics that cannot win was measured
// and is slower: the no-return atomic is fire-and-forget, the read is a dependent round trip.)
__device__ __forceinline__ void key_min(unsigned long long* __restrict__ addr, unsigned long long key) { atomicMin(addr, key); }

// ------------------------------------------------------------------ index map (a10)
// index_map.vert:40-66: per surfel 24 B (pos+conf 16, times 8), one 64-bit atomicMin per visible surfel.
__global__ __launch_bounds__(MAP_THREADS) void k_index_project(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc,
                                                               const float2* __restrict__ tm, Cam c, int time, unsigned long long* __restrict__ keys)
{
    const int fl = FIRST_LIVE(c);
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    const int n = st->count;
    const int lo = (int)((long long)n * c.srank / c.sn), hi = (int)((long long)n * (c.srank + 1) / c.sn);   // this rank's slice (all slots when not sharded)
    for (int i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += blockDim.x * gridDim.x) {
        float lastT = tm[i].y;
        if ((float)time - lastT > (float)c.timeDelta) continue;   // inactive (or tombstone): 8 B, position not loaded
        float4 p4 = pc[i];
        v3 p = xf_point(T, v3m(p4.x, p4.y, p4.z));
        if (p.z > c.maxDepth || p.z < 0) continue;
        float u = ((c.fx * p.x) / p.z) + c.cx, v = ((c.fy * p.y) / p.z) + c.cy;
        if (!(u >= 0 && u < (float)c.w && v >= 0 && v < (float)c.h)) continue;
        const int px = point_pixel(u), py = point_pixel(v);
        if (px < 0 || py < 0) continue;
        key_min(&keys[py * c.w + px], make_key(p.z, key_id(c, (unsigned int)i, fl)));
    }
}
// index_map.frag:33-40: gathers the winner's attributes; also re-arms the key image for the next pass
__global__ void k_index_resolve(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, unsigned long long* __restrict__ keys, const float4* __restrict__ pc,
                                const float4* __restrict__ nr, const float2* __restrict__ col, const float2* __restrict__ tm, int P, uint32_t* __restrict__ index_id,
                                float4* __restrict__ vc, float4* __restrict__ ct, float4* __restrict__ nrm, int time, float conf_thr, float4* __restrict__ tap, Cam c,
                                const int32_t* __restrict__ own_slot = nullptr, const Hot* __restrict__ hot = nullptr)
{
    const int fl = FIRST_LIVE(c);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    unsigned long long key = keys[k];
    keys[k] = IFX_KEY_EMPTY;
    // index_id == nullptr: the caller only needs the tap record (the index map of the clean pass): the winner's normal and
    // colour are not fetched and the 68 B of attribute images are not written
    if (key == IFX_KEY_EMPTY) {
        if (index_id) { index_id[k] = 0; vc[k] = make_float4(0, 0, 0, 0); nrm[k] = make_float4(0, 0, 0, 0); if (ct) ct[k] = make_float4(0, 0, 0, 0); }
        if (tap) tap[k] = make_float4(0, 0, 0, 0);
        return;
    }
    const float* T = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    unsigned int id = (unsigned int)(key & 0xFFFFFFFFull);
    const int li = own_slot ? own_slot_of(c, own_slot[k], id) : local_slot(c, st->count, id, fl);
    if (c.own_n > 0 && id == (unsigned int)fl) id = 0;   // sharded map: the reference's "surfel 0" (FIRST_LIVE) -- found by its creation number, named 0
    if (li < 0) {   // sharded map: another rank's surfel won this pixel -- that rank writes its attributes, this one zeros (the images are summed bitwise across ranks)
        if (index_id) { index_id[k] = id; vc[k] = make_float4(0, 0, 0, 0); nrm[k] = make_float4(0, 0, 0, 0); if (ct) ct[k] = make_float4(0, 0, 0, 0); }
        if (tap) tap[k] = make_float4(0, 0, 0, 0);
        return;
    }
    float4 p4 = hot ? hot[li].pc : pc[li];
    float2 t2 = make_float2(0.f, 0.f);
    if (ct || tap) t2 = hot ? hot[li].tm : tm[li];
    float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (index_id) n4 = hot ? hot[li].nr : nr[li];   // (with the position and the times: one round trip for the winner's record)
    v3 p = xf_point(T, v3m(p4.x, p4.y, p4.z));
    if (index_id) {
        v3 nn = normalized(xf_dir(T, v3m(n4.x, n4.y, n4.z)));
        index_id[k] = id;
        vc[k] = make_float4(p.x, p.y, p.z, p4.w);
        if (ct) { float2 c2 = col[li]; ct[k] = make_float4(c2.x, c2.y, t2.x, t2.y); }
        nrm[k] = make_float4(nn.x, nn.y, nn.z, n4.w);
    }
    // 16-B record for the clean window taps: (x, y, z, initTime) with two flags in the (otherwise positive)
    // signs: z < 0 <=> updated this frame (colorTime.w == time), w > 0 <=> stable (vertConf.w > confThreshold)
    if (tap) tap[k] = (id > 0u && p.z > 0.f) ? make_float4(p.x, p.y, (t2.y == (float)time) ? -p.z : p.z, (p4.w > conf_thr) ? t2.x : -t2.x) : make_float4(0, 0, 0, 0);
}

// for_association: the frame path, where k_associate is the only consumer (it reads ids, positions and normals): the colour /
// time image is not produced and the winner's colour and times are not fetched
static void index_pass(ifx* h, const float* d_pose_inv, int time, bool for_association = false, int part = 0, const int32_t* own_slot = nullptr)
{
    Cam c = make_cam(h);
    if (part == 0) { c.srank = 0; c.sn = 1; }   // a whole pass (stage API, re-render after a compaction) is never sliced
    if (part != 2) LAUNCH(h, "index_project", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_index_project, h->d_state, d_pose_inv, (const float4*)h->pc, (const float2*)h->tm, c, time, h->key_index);
    if (part != 1) LAUNCH(h, "index_resolve", dim3(cdiv(h->P, 256)), dim3(256), k_index_resolve, h->d_state, d_pose_inv, h->key_index, (const float4*)h->pc, (const float4*)h->nr,
           (const float2*)h->col, (const float2*)h->tm, h->P, h->index_id, (float4*)h->index_vc, for_association ? (float4*)nullptr : (float4*)h->index_ct, (float4*)h->index_nr, time,
           h->cfg.confidence, (float4*)nullptr, c, own_slot);
}

// ------------------------------------------------------------------ disc rasteriser (a9, a14)
struct Disc { v3 q, n; float r2; };
// combo_splat.frag:39-52
__device__ inline bool disc_hit(const Disc& d, float px, float py, const Cam& c, float& z)
{
    // un-normalised ray: l*(q.n)/(l.n) does not depend on |l| (see oracle/orc_map.c disc_hit)
    v3 l = v3m((px - c.cx) * (1.0f / c.fx), (py - c.cy) * (1.0f / c.fy), 1.0f);
    float s = dot(d.q, d.n) / dot(l, d.n);
    v3 cp = l * s;
    v3 df = cp - d.q;
    if (!(dot(df, df) <= d.r2)) return false;
    z = cp.z;
    return true;
}
// splat.vert:55-92
__device__ inline void disc_extent(v3 q, v3 n, float r, const Cam& c, float* xs, float* ys, float& minz)
{
    v3 x1 = normalized(v3m(n.y - n.z, -n.x, n.x)) * (r * 1.41421356f);
    v3 y1 = cross(n, x1);
    v3 pts[4] = {q + x1, q + y1, q - y1, q - x1};
    xs[0] = ys[0] = INFINITY; xs[1] = ys[1] = -INFINITY; minz = INFINITY;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float u = ((c.fx * pts[k].x) / pts[k].z) + c.cx, v = ((c.fy * pts[k].y) / pts[k].z) + c.cy;
        xs[0] = fminf(xs[0], u); xs[1] = fmaxf(xs[1], u);
        ys[0] = fminf(ys[0], v); ys[1] = fmaxf(ys[1], v);
        minz = fminf(minz, pts[k].z);
    }
}

// MODE 0: combinedPredict splat (splat.vert culls, sprite extent, GL point clipping by centre)
// MODE 1: renderSurfelIds GENERAL (surfel_ids.vert/.geom culls, quad extent)
// MODE 2: renderSurfelIds INSTANCECOMPARE (instance_surfel_ids.vert:44-70: skips surfels whose 12 vote vec4 are equal)
template <int MODE>
__global__ __launch_bounds__(MAP_THREADS) void k_raster(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc,
                                                        const float4* __restrict__ nr, const float2* __restrict__ tm, const float4* __restrict__ votes, int cap, Cam c,
                                                        int time, int maxTime, unsigned long long* __restrict__ keys)
{
    const int fl = FIRST_LIVE(c);
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        float4 p4 = pc[i];
        if (MODE == 0) { if (p4.w < c.conf) continue; }
        else { if (!(p4.w > c.conf)) continue; }
        v3 q = xf_point(T, v3m(p4.x, p4.y, p4.z));
        if (MODE == 0) {
            float lastT = tm[i].y;
            if (q.z > c.maxDepth || q.z < 0 || (float)time - lastT > (float)c.timeDelta || lastT > (float)maxTime) continue;
        } else {
            if (!(q.z / c.maxDepth > 0.01f)) continue;
        }
        float u = ((c.fx * q.x) / q.z) + c.cx, v = ((c.fy * q.y) / q.z) + c.cy;
        if (MODE == 0) { if (!(u >= 0 && u <= (float)c.w && v >= 0 && v <= (float)c.h)) continue; }
        if (MODE == 2) {
            // every vote vec4 compared with the first one (vInstInfoB..L == vInstInfoA)
            float4 a = VOTE4(votes, i, 0);
            bool alleq = true;
            for (int k = 1; k < 12 && alleq; k++) {
                float4 b = VOTE4(votes, i, k);
                alleq = (b.x == a.x) && (b.y == a.y) && (b.z == a.z) && (b.w == a.w);
            }
            if (alleq) continue;
        }
        float4 n4 = nr[i];
        v3 nn = normalized(xf_dir(T, v3m(n4.x, n4.y, n4.z)));
        float r = n4.w;
        float xs[2], ys[2], minz;
        disc_extent(q, nn, r, c, xs, ys, minz);
        int x0, x1, y0, y1;
        if (MODE == 0) {
            float s = fmaxf(fabsf(xs[1] - xs[0]), fabsf(ys[1] - ys[0]));
            if (!(s == s)) continue;
            s = fminf(fmaxf(s, 1.0f), IFX_MAX_SPRITE);
            x0 = clampi((int)ceilf(u - s * 0.5f - 0.5f), 0, c.w - 1); x1 = clampi((int)floorf(u + s * 0.5f - 0.5f), 0, c.w - 1);
            y0 = clampi((int)ceilf(v - s * 0.5f - 0.5f), 0, c.h - 1); y1 = clampi((int)floorf(v + s * 0.5f - 0.5f), 0, c.h - 1);
        } else {
            if (!(minz > 0) || !(xs[0] == xs[0]) || !(ys[0] == ys[0])) continue;
            if (xs[1] - xs[0] > IFX_MAX_SPRITE || ys[1] - ys[0] > IFX_MAX_SPRITE) continue;
            if (xs[1] < 0 || ys[1] < 0 || xs[0] > (float)c.w || ys[0] > (float)c.h) continue;
            x0 = clampi((int)ceilf(xs[0] - 0.5f), 0, c.w - 1); x1 = clampi((int)floorf(xs[1] - 0.5f), 0, c.w - 1);
            y0 = clampi((int)ceilf(ys[0] - 0.5f), 0, c.h - 1); y1 = clampi((int)floorf(ys[1] - 0.5f), 0, c.h - 1);
        }
        Disc d;
        d.q = q; d.n = nn; d.r2 = r * r;
        for (int py = y0; py <= y1; py++)
            for (int px = x0; px <= x1; px++) {
                float z;
                if (!disc_hit(d, (float)px + 0.5f, (float)py + 0.5f, c, z)) continue;
                if (MODE == 0) { if (!(z >= -c.maxDepth && z <= c.maxDepth)) continue; }
                else { if (!(z > 0 && z <= c.maxDepth)) continue; }
                atomicMin(&keys[py * c.w + px], make_key(z, key_id(c, (unsigned int)i, fl)));
            }
    }
}

__device__ inline float key_depth(unsigned long long key)
{
    unsigned int b = (unsigned int)(key >> 32);
    unsigned int u = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
    return __uint_as_float(u);
}

__global__ void k_ids_resolve(unsigned long long* __restrict__ keys, int P, int32_t* __restrict__ ids)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    unsigned long long key = keys[k];
    keys[k] = IFX_KEY_EMPTY;
    ids[k] = (key == IFX_KEY_EMPTY) ? 0 : (int32_t)(key & 0xFFFFFFFFull);
}

// End-of-pass sums folded into the resolve (view-list frames of the unsharded map; k_raster_finish's job everywhere else): the thread of a dense-test sample
// or of a lattice pixel of whetherDoSegmentation already holds the colour / the id the sums ask about, and the twelve vote gathers of the 1 % lattice threads
// hide under the kernel instead of being a 12 us launch of their own.  Block totals go to sixteen partials (DevState::fold_acc), k_frame_result folds them.
struct FinishFold {
    int* acc;             // null: off
    int* total;           // null: the same launch also writes the frame result (FrameOut) and is told the total itself
    const float4* votes;
    int cap, ds, rw, rh;
};
// The frame result written by the resolve's last block (the view-list frame path: the resolve is the frame's last launch, and what k_frame_result would read is
// complete when its last block is): one dispatch less on every frame's chain.
struct FrameOut {
    FrameResult* out;     // null: off
    float* traj;
    int fold_total;
};

#define NEW_PER_BLOCK 1024   // pixels (in append order) per block of the new-surfel flags pass

// combo_splat.frag:54-66 outputs for the winner of each pixel, fused with FillIn
// (fill_rgb/vertex/normal.frag, EF/Shaders/FillIn.cpp:65-195, passthrough = 0).
__device__ __forceinline__ void splat_resolve_body(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, unsigned long long* __restrict__ keys, const float4* __restrict__ pc,
                                const float4* __restrict__ nr, const float2* __restrict__ col, const float2* __restrict__ tm, Cam c, const uint8_t* __restrict__ rgb,
                                const uint16_t* __restrict__ depth_filt, float4* __restrict__ pv, float4* __restrict__ pn, uchar4* __restrict__ pimg,
                                uchar4* __restrict__ pinst, uint16_t* __restrict__ ptime, float4* __restrict__ fv, float4* __restrict__ fn, uchar4* __restrict__ fimg,
                                unsigned long long* __restrict__ id_keys, unsigned long long* __restrict__ both_keys, int32_t* __restrict__ ids_out,
                                int* __restrict__ n_valid, FinishFold fold, float* __restrict__ pconf, int ids_step = 1, const int32_t* __restrict__ own_slot = nullptr,
                                const int fl = 0, const bool raw_ids = false, const Hot* __restrict__ hot = nullptr)   // fl: FIRST_LIVE(c) from the top of the kernel; raw_ids: the walk that drew these keys named every surfel by its slot (option clean_raster): "surfel 0" is named here
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= c.w || y >= c.h) return;
    int k = y * c.w + x;
    // everything addressed by the pixel itself leaves in one batch: the three keys and, for the fill-in, the frame's colour and filtered depth
    // (one after the other, each behind the previous test, they were five dependent round trips)
    unsigned long long key = keys[k];
    // (sparse id render: only the lattice pixels were drawn, the other pixels' id keys are empty and stay so, and their entries of the id image are not looked at
    // before ifx_ids_ensure redraws the whole image)
    if (ids_step > 1 && (x % ids_step != 0 || y % ids_step != 0)) ids_out = nullptr;
    const unsigned long long bk0 = ids_out ? both_keys[k] : IFX_KEY_EMPTY, ik0 = ids_out ? id_keys[k] : IFX_KEY_EMPTY;
    uint8_t f_r = 0, f_g = 0, f_b = 0;
    uint16_t f_d = 0, f_dx = 0, f_dy = 0;
    if (fv) {
        f_r = rgb[k * 3]; f_g = rgb[k * 3 + 1]; f_b = rgb[k * 3 + 2];
        f_d = depth_filt[k];
        f_dx = depth_filt[y * c.w + clampi(x + 1, 0, c.w - 1)];
        f_dy = depth_filt[clampi(y + 1, 0, c.h - 1) * c.w + x];
    }
    asm volatile("" ::"v"((unsigned int)key), "v"((unsigned int)bk0), "v"((unsigned int)ik0), "v"((unsigned int)f_r), "v"((unsigned int)f_d), "v"((unsigned int)f_dx), "v"((unsigned int)f_dy));
    keys[k] = IFX_KEY_EMPTY;
    int fold_id = 0;
    if (ids_out) {   // k_ids_resolve of the id render that shared the raster pass; `both` holds the pixels common to the two renders
        const unsigned long long bk = bk0;
        both_keys[k] = IFX_KEY_EMPTY;
        unsigned long long ik = ik0;
        id_keys[k] = IFX_KEY_EMPTY;
        ik = ik < bk ? ik : bk;
        key = key < bk ? key : bk;
        fold_id = (ik == IFX_KEY_EMPTY) ? 0 : (int32_t)(ik & 0xFFFFFFFFull);
        if ((raw_ids || c.own_n > 0) && fold_id == fl) fold_id = 0;   // (clean_raster: the walk drew slot numbers; the append behind it settled which slot is the reference's "surfel 0")
        ids_out[k] = fold_id;
    }
    float4 vo = make_float4(0, 0, 0, 0), no = make_float4(0, 0, 0, 0);
    uchar4 io = make_uchar4(0, 0, 0, 0), so = make_uchar4(0, 0, 0, 0);
    uint16_t to = 0;
    int li = -1;
    if (key != IFX_KEY_EMPTY) {
        li = raw_ids ? (int)(unsigned int)(key & 0xFFFFFFFFull)
                     : (own_slot ? own_slot_of(c, own_slot[k], (unsigned int)(key & 0xFFFFFFFFull)) : local_slot(c, st->count, (unsigned int)(key & 0xFFFFFFFFull), fl));
        if (li < 0) {   // sharded map: the winner's rank writes this pixel, the others leave zeros (summed bitwise across ranks) ...
            if (pconf) {   // ... except the vertex of the frame's prediction, which every rank rebuilds from the key it holds (the confidence comes from the owner through pconf)
                const float z = key_depth(key);
                vo = make_float4(((float)x + 0.5f - c.cx) * z * (1.f / c.fx), ((float)y + 0.5f - c.cy) * z * (1.f / c.fy), z, 0.f);
            }
            key = IFX_KEY_EMPTY;
        }
    }
    if (key != IFX_KEY_EMPTY) {
        const float* T = pose_inv_ex ? pose_inv_ex : st->pose_inv;
        unsigned int id = (unsigned int)li;
        float z = key_depth(key);
        float4 p4 = hot ? hot[id].pc : pc[id], n4 = hot ? hot[id].nr : nr[id];
        float2 c2 = col[id], t2 = hot ? hot[id].tm : tm[id];
        v3 nn = normalized(xf_dir(T, v3m(n4.x, n4.y, n4.z)));
        float fpx = (float)x + 0.5f, fpy = (float)y + 0.5f;
        vo = make_float4((fpx - c.cx) * z * (1.f / c.fx), (fpy - c.cy) * z * (1.f / c.fy), z, p4.w);
        no = make_float4(nn.x, nn.y, nn.z, n4.w);
        float c3[3];
        decode_color(c2.x, c3);
        io = make_uchar4((uint8_t)(int)roundf(c3[0] * 255.0f), (uint8_t)(int)roundf(c3[1] * 255.0f), (uint8_t)(int)roundf(c3[2] * 255.0f), 255);
        decode_color(c2.y, c3);
        so = make_uchar4((uint8_t)(int)roundf(c3[0] * 255.0f), (uint8_t)(int)roundf(c3[1] * 255.0f), (uint8_t)(int)roundf(c3[2] * 255.0f), 255);
        to = (uint16_t)(unsigned int)t2.x;
    }
    if (pconf) { pconf[k] = vo.w; vo.w = 0.f; }   // (owner: the confidence travels apart; fill_in puts it back after the exchange)
    pv[k] = vo; pn[k] = no; pimg[k] = io; pinst[k] = so; ptime[k] = to;
    (void)n_valid;
    if (fv) {   // (no fill-in for the loop-closure renders: EF/ElasticFusion.cpp:519-534 reads the raw old textures)  -- before the end-of-pass sums: what it needs is dead when their twelve vote gathers are in flight
    // fill-in
    float ifx_ = 1.0f / c.fx, ify_ = 1.0f / c.fy;
    if ((int)io.x + (int)io.y + (int)io.z == 0) fimg[k] = make_uchar4(f_r, f_g, f_b, 255);
    else fimg[k] = io;
    float zc = (float)f_d / 1000.0f;
    if (vo.z == 0) fv[k] = make_float4(((float)x - c.cx) * zc * ifx_, ((float)y - c.cy) * zc * ify_, zc, 1.f);
    else fv[k] = vo;
    if (no.z == 0) {
        v3 vp = v3m(((float)x - c.cx) * zc * ifx_, ((float)y - c.cy) * zc * ify_, zc);
        float zx = (float)f_dx / 1000.0f, zy = (float)f_dy / 1000.0f;
        v3 vx = v3m(((float)(x + 1) - c.cx) * zx * ifx_, ((float)y - c.cy) * zx * ify_, zx);
        v3 vy = v3m(((float)x - c.cx) * zy * ifx_, ((float)(y + 1) - c.cy) * zy * ify_, zy);
        v3 nn = normalized(cross(vx - vp, vy - vp));
        fn[k] = make_float4(nn.x, nn.y, nn.z, 1.f);
    } else fn[k] = no;
    }
    if (fold.acc) {   // (uniform)
        __shared__ int s_f[3];
        const int lt = threadIdx.y * blockDim.x + threadIdx.x;
        if (lt < 3) s_f[lt] = 0;
        __syncthreads();
        {   // ElasticFusion::denseEnough, EF/ElasticFusion.cpp:252-267: the samples of the (w/20 x h/20) nearest resample
            const int i = x * fold.rw / c.w, j = y * fold.rh / c.h;
            if ((i * c.w + c.w / 2) / fold.rw == x && (j * c.h + c.h / 2) / fold.rh == y && io.x > 0 && io.y > 0 && io.z > 0) atomicAdd(&s_f[2], 1);
        }
        if (ids_out && x % fold.ds == 0 && y % fold.ds == 0) {   // checkProjectDepthAndInstanceKernel, IF/Core/InstanceFusionCuda.cu:736-760
            const int gid = fold_id;
            if (gid > 0 && gid < st->count) {
                float4 v[12];
#pragma unroll
                for (int q = 0; q < 12; q++) v[q] = VOTE4(fold.votes, gid, q);
                int mass = 0;
#pragma unroll
                for (int q = 0; q < 12; q++) {
                    int a, b;
                    vote_decode(v[q].x, a, b); mass += a + b;
                    vote_decode(v[q].y, a, b); mass += a + b;
                    vote_decode(v[q].z, a, b); mass += a + b;
                    vote_decode(v[q].w, a, b); mass += a + b;
                }
                if (mass) atomicAdd(&s_f[0], mass);
            } else atomicAdd(&s_f[1], 1);
        }
        __syncthreads();
        if (lt < 3 && s_f[lt]) atomicAdd(fold.acc + (blockIdx.x & 15) * 4 + lt, s_f[lt]);
        if (fold.total && lt == 0 && blockIdx.x == 0 && blockIdx.y == 0) *fold.total = fold.rw * fold.rh;
    }
}

// Cam::raw_slots == 2 (option clean_raster): the keys were drawn by the fused clean + raster walk, which names every surfel by its slot -- it cannot know which slot is the reference's
// "surfel 0" while it is still removing surfels; k_append_scan, between the walk and this launch, settles first_live, and the id image gets its 0 here.
__global__ void k_splat_resolve(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, unsigned long long* __restrict__ keys, const float4* __restrict__ pc,
                                const float4* __restrict__ nr, const float2* __restrict__ col, const float2* __restrict__ tm, Cam c, const uint8_t* __restrict__ rgb,
                                const uint16_t* __restrict__ depth_filt, float4* __restrict__ pv, float4* __restrict__ pn, uchar4* __restrict__ pimg,
                                uchar4* __restrict__ pinst, uint16_t* __restrict__ ptime, float4* __restrict__ fv, float4* __restrict__ fn, uchar4* __restrict__ fimg,
                                unsigned long long* __restrict__ id_keys, unsigned long long* __restrict__ both_keys, int32_t* __restrict__ ids_out,
                                int* __restrict__ n_valid, FinishFold fold, float* __restrict__ pconf = nullptr, int ids_step = 1, const int32_t* __restrict__ own_slot = nullptr,
                                FrameOut fo = FrameOut{nullptr, nullptr, 0}, const Hot* __restrict__ hot = nullptr)
{
    const int fl = FIRST_LIVE(c);
    splat_resolve_body(st, pose_inv_ex, keys, pc, nr, col, tm, c, rgb, depth_filt, pv, pn, pimg, pinst, ptime, fv, fn, fimg, id_keys, both_keys, ids_out, n_valid, fold, pconf, ids_step, own_slot, fl, c.raw_slots == 2, hot);
    if (!fo.out) return;
    // the block that finishes last writes the frame result: every block's fold atomics are performed (vmcnt drained) before its ticket
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DevState* stw = const_cast<DevState*>(st);
    const int lt = threadIdx.y * blockDim.x + threadIdx.x;
    if (lt == 0) {
        const unsigned int t = __hip_atomic_fetch_add(&stw->result_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x * gridDim.y - 1u);
    }
    __syncthreads();
    if (!s_last) return;
    if (lt == 0) __hip_atomic_store(&stw->result_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lt < 64) frame_result_wave(stw, fo.out, fo.traj, fo.fold_total, lt);
}
// the two renders of the loop-closure detection (ACTIVE into the act* images, INACTIVE into the old* images) resolved by one launch: blockIdx.z picks the render
struct ResolveTarget { unsigned long long* keys; float4 *pv, *pn; uchar4 *pimg, *pinst; uint16_t* ptime; };
__global__ void k_splat_resolve_pair(const DevState* __restrict__ st, const float4* __restrict__ pc, const float4* __restrict__ nr, const float2* __restrict__ col, const float2* __restrict__ tm,
                                     Cam c, const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth_filt, ResolveTarget t0, ResolveTarget t1)
{
    const int fl = FIRST_LIVE(c);
    const ResolveTarget t = blockIdx.z ? t1 : t0;
    splat_resolve_body(st, nullptr, t.keys, pc, nr, col, tm, c, rgb, depth_filt, t.pv, t.pn, t.pimg, t.pinst, t.ptime, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, FinishFold(),
                       nullptr, 1, nullptr, fl);
}

// ElasticFusion::denseEnough, EF/ElasticFusion.cpp:252-267 on the (w/20 x h/20) nearest resample
// End of a raster pass in one launch: block 0 = the dense-enough test (EF/ElasticFusion.cpp:252-267) and the re-arm of the work list;
// blocks 1.. = checkProjectDepthAndInstanceKernel (IF/Core/InstanceFusionCuda.cu:736-760) over the id image this pass
// rendered, accumulated for k_frame_result, so that whetherDoSegmentation needs no launch of its own.
// Sharded map: the vote mass goes to `mass_out` (the tail of the prediction block, summed across the ranks with it) and the launch of the next
// phase takes the total back through `mass_in`.
__global__ void k_raster_finish(DevState* st, const uchar4* __restrict__ pimg, int w, int h, int do_dense, const int32_t* __restrict__ ids, const float4* __restrict__ votes, int cap,
                                int downsample, unsigned int* __restrict__ lctr, IdMap im, int* __restrict__ mass_out = nullptr, int* __restrict__ mass_in = nullptr,
                                const int32_t* __restrict__ own_slot = nullptr)
{
    if (blockIdx.x == 0) {
        if (mass_in && threadIdx.x == 0) { st->seg_acc[0] = mass_in[0]; mass_in[0] = 0; }
        if (do_dense) {
            __shared__ int lds[4];
            int rw = w / 20, rh = h / 20, cnt = 0;
            for (int t = threadIdx.x; t < rw * rh; t += blockDim.x) {
                int j = t / rw, i = t - j * rw;
                int sx = (i * w + w / 2) / rw, sy = (j * h + h / 2) / rh;
                uchar4 s = pimg[sy * w + sx];
                cnt += (s.x > 0 && s.y > 0 && s.z > 0);
            }
            cnt = wave_sum_i(cnt);
            if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = cnt;
            __syncthreads();
            if (threadIdx.x == 0) {
                int sum = lds[0] + lds[1] + lds[2] + lds[3];
                st->dense_enough = ((float)sum / (float)(rw * rh) > 0.75f) ? 1 : 0;
            }
        }
        if (threadIdx.x < LIST_SEGS) lctr[threadIdx.x * LIST_CTR_STRIDE] = 0;   // re-arm the raster work list (list 0)
        return;
    }
    const int gw = (w + downsample - 1) / downsample, gh = (h + downsample - 1) / downsample;
    const int t = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
    int mass = 0, empty = 0;
    if (t < gw * gh) {
        const int gy = t / gw, gx = t - gy * gw, x = gx * downsample, y = gy * downsample;
        if (x < w && y < h) {
            const int gid = ids[y * w + x];
            // sharded map: the vote mass of a pixel is counted by the rank that owns its surfel (summed across ranks afterwards)
            int id;
            if (own_slot) { const int sl = own_slot[y * w + x]; id = (gid != 0 && sl >= 0 && im.seq[sl] == (uint32_t)gid) ? sl : -1; }
            else id = idmap_slot(im, st->count, gid);
            if (im.own_n > 0 ? gid > 0 : id >= 0) {
              if (id >= 0) {
                float4 v[12];   // all twelve planes in flight together (one after the other they were twelve HBM round trips: 13 us for this little kernel)
#pragma unroll
                for (int q = 0; q < 12; q++) v[q] = VOTE4(votes, id, q);
#pragma unroll
                for (int q = 0; q < 12; q++) {
                    int a, b;
                    vote_decode(v[q].x, a, b); mass += a + b;
                    vote_decode(v[q].y, a, b); mass += a + b;
                    vote_decode(v[q].z, a, b); mass += a + b;
                    vote_decode(v[q].w, a, b); mass += a + b;
                }
              }
            } else empty = 1;
        }
    }
    mass = wave_sum_i(mass);
    empty = wave_sum_i(empty);
    if ((threadIdx.x & 63) == 0) {
        if (mass) atomicAdd(mass_out ? mass_out : &st->seg_acc[0], mass);
        if (empty) atomicAdd(&st->seg_acc[1], empty);
    }
}

__device__ inline int clean_test(const float* T, const Cam& c, int time, float4 p4, float4 n4, float initT, float& lastT, const float4* __restrict__ tap);

// ------------------------------------------------------------------ work-list passes
// A map pass over N slots is split in two: (1) a pure streaming cull that reads 24 B per slot
// (pos+conf, times), decides with a conservative frustum test and appends the few survivors to a
// work list (wave ballot -> one atomicAdd per wave), and (2) a dense pass over the list that reads
// normal+radius and does the expensive part (disc rasterisation / window taps) with every lane
// busy.  The one-kernel versions above ran at 0.6-1.1 TB/s because the heavy path diverged inside
// waves of mostly-culled surfels (profiles/archive/r01_a: the one-kernel raster 205 us, the one-kernel clean 206 us for 5.6M slots).
#define LIST_SPLAT 0x40000000u
#define LIST_IDS 0x80000000u
#define LIST_IDX 0x3FFFFFFFu
#define LIST_DUAL 0x1u   // only in the `want` argument of the cull: two splat renders (active / inactive window) from one scan

// Block-aggregated list append: survivors of a 4096-slot chunk are collected in LDS (wave ballot ->
// one LDS atomic per wave) and flushed with ONE global atomicAdd per block and chunk.  (One global
// returning atomic per wave on a single counter saturates at ~88 per microsecond -- 87k waves cost 1 ms.)
#ifndef CHUNK_ROUNDS
#define CHUNK_ROUNDS 16
#endif
#define CHUNK_SLOTS (MAP_THREADS * CHUNK_ROUNDS)
static_assert(CHUNK_SLOTS == 4096, "ifx_create sizes the work-list segments for 4096-slot chunks (list_seg_cap)");
// Per-chunk append without an LDS staging buffer: every round reserves wave-contiguous positions in a
// block-local counter (one LDS atomic per wave), the block reserves its range of the global list with
// ONE global atomic, and each thread then writes its survivors straight to list[base + position].
struct BlockCount { unsigned int n; unsigned int base; };
struct BlockCount2 { unsigned int n[2]; unsigned int base[2]; };
__device__ __forceinline__ unsigned int bcount_reserve(BlockCount& L, bool pred)
{
    unsigned long long m = __ballot(pred);
    if (m == 0ull) return 0u;
    const int lane = threadIdx.x & 63;
    unsigned int base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(&L.n, (unsigned int)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}
// two mutually exclusive predicates, two lists, one pass
__device__ __forceinline__ unsigned int bcount2_reserve(BlockCount2& L, bool a, bool b)
{
    unsigned long long ma = __ballot(a), mb = __ballot(b);
    if ((ma | mb) == 0ull) return 0u;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned int ba = 0, bb = 0;
    if (lane == 0) { if (ma) ba = atomicAdd(&L.n[0], (unsigned int)__popcll(ma)); if (mb) bb = atomicAdd(&L.n[1], (unsigned int)__popcll(mb)); }
    ba = __shfl(ba, 0, 64); bb = __shfl(bb, 0, 64);
    return a ? ba + __popcll(ma & below) : bb + __popcll(mb & below);
}
// all threads; after it L.base holds the block's range start in the global list and L.n is re-armed
__device__ __forceinline__ unsigned int bcount_commit(BlockCount& L, unsigned int* counter)
{
    __syncthreads();
    if (threadIdx.x == 0) { L.base = L.n ? atomicAdd(counter, L.n) : 0u; L.n = 0; }
    __syncthreads();
    return L.base;
}

// Conservative "can this surfel's disc touch the image?" test from its centre alone.  Every point of the
// disc (and of the quad around it) lies within reach = r_max*sqrt(2) of the centre in 3-D, so its
// projection moves by at most f*reach*(1 + |x|/z)/(z - reach) pixels; r_max bounds every radius ever
// stored.  Approximate reciprocals are fine here: the test only has to be conservative (2 px + 0.5 % slack);
// survivors are re-tested exactly in phase 2.
__device__ __forceinline__ bool may_touch_image(float reach, v3 q, const Cam& c)
{
    const float zn = q.z - reach;
    if (!(zn > 1e-3f)) return true;
    const float rz = __builtin_amdgcn_rcpf(q.z), rzn = __builtin_amdgcn_rcpf(zn) * 1.005f;
    const float ax = c.fx * q.x * rz, ay = c.fy * q.y * rz;          // u - cx, v - cy (approx.)
    const float mx = reach * (c.fx + fabsf(ax)) * rzn + 2.0f, my = reach * (c.fy + fabsf(ay)) * rzn + 2.0f;
    const float u = ax + c.cx, v = ay + c.cy;
    return !(u + mx < 0.f || v + my < 0.f || u - mx > (float)c.w || v - my > (float)c.h) || !(u == u) || !(v == v);
}

// phase 1 of the post-clean pass: candidates for the splat prediction and / or the id render
__global__ __launch_bounds__(MAP_THREADS) void k_cull_raster(DevState* st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc,
                                                             const float2* __restrict__ tm, Cam c, int time, int maxTime, unsigned int want,
                                                             unsigned int* __restrict__ list, unsigned int* __restrict__ zero_buf, int zero_n)
{
    if (zero_buf && blockIdx.x == gridDim.x - 1)   // the tiled rasteriser's per-tile counts, for the count pass that follows (an idle block: the grid exceeds the chunks)
        for (int k = threadIdx.x; k < zero_n; k += blockDim.x) zero_buf[k] = 0;
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    __shared__ BlockCount L;
    if (threadIdx.x == 0) L.n = 0;
    __syncthreads();
    const int n = st->count;
    const int lo = (int)((long long)n * c.srank / c.sn), hi = (int)((long long)n * (c.srank + 1) / c.sn);   // this rank's slice (all slots when not sharded)
    const float reach = __uint_as_float(st->r_max_bits) * 1.41421356f * 1.001f;
    for (int chunk = lo / CHUNK_SLOTS + blockIdx.x; chunk * CHUNK_SLOTS < hi; chunk += gridDim.x) {
        unsigned int val[CHUNK_ROUNDS], pos[CHUNK_ROUNDS];
#pragma unroll
        for (int r = 0; r < CHUNK_ROUNDS; r++) {
            int i = chunk * CHUNK_SLOTS + r * MAP_THREADS + threadIdx.x;
            unsigned int flags = 0;
            if (i >= lo && i < hi) {
                float4 p4 = pc[i];
                float lastT = tm[i].y;
                // cheapest tests first: unstable surfels are on neither list, nor is anything behind the camera
                if (!(p4.w < c.conf)) {
                    v3 q = xf_point(T, v3m(p4.x, p4.y, p4.z));
                    if (q.z > 0.f && may_touch_image(reach, q, c)) {
                        if ((want & LIST_IDS) && (p4.w > c.conf) && (q.z / c.maxDepth > 0.01f)) flags |= LIST_IDS;
                        if ((want & LIST_SPLAT) && !(q.z > c.maxDepth)) {
                            // LIST_DUAL (loop-closure renders): the ACTIVE prediction (time, maxTime) and the INACTIVE one (time = 0, maxTime = time - timeDelta:
                            // last seen at or before that) in one scan -- the second flag travels in the LIST_IDS bit, which that pass does not use otherwise
                            const bool act = !((float)time - lastT > (float)c.timeDelta || lastT > (float)maxTime);
                            const bool old = (want & LIST_DUAL) && !(0.f - lastT > (float)c.timeDelta || lastT > (float)(time - c.timeDelta));
                            if (act || old) {
                                float u = ((c.fx * q.x) / q.z) + c.cx, v = ((c.fy * q.y) / q.z) + c.cy;   // exact: GL clips points by their centre
                                if (u >= 0 && u <= (float)c.w && v >= 0 && v <= (float)c.h) flags |= (act ? LIST_SPLAT : 0u) | (old ? LIST_IDS : 0u);
                            }
                        }
                    }
                }
            }
            val[r] = flags ? ((unsigned int)i | flags) : 0u;
            pos[r] = bcount_reserve(L, flags != 0);
        }
        const int seg = chunk % LIST_SEGS;
        const unsigned int base = seg * c.seg_cap + bcount_commit(L, list_ctr(c, 0, seg));
#pragma unroll
        for (int r = 0; r < CHUNK_ROUNDS; r++)
            if (val[r]) list[base + pos[r]] = val[r];
    }
}

// Screen-space geometry of one listed surfel (splat.vert:55-92): camera-frame centre and normal, the projected extent of its quad, the point-sprite box
// of the splat render.  Shared by the global-atomic rasteriser (k_raster_list) and the tiled one (k_tile_*), so that both draw the same pixels.
struct SurfGeo { v3 q, nn; float r, u, v, xs[2], ys[2], minz; bool do_s; int sx0, sx1, sy0, sy1; };
__device__ __forceinline__ void surfel_geo_cam(v3 q, v3 nn, float r, unsigned int e, const Cam& c, SurfGeo& G);
__device__ __forceinline__ void surfel_geo(const float* T, float4 p4, float4 n4, unsigned int e, const Cam& c, SurfGeo& G)
{
    surfel_geo_cam(xf_point(T, v3m(p4.x, p4.y, p4.z)), normalized(xf_dir(T, v3m(n4.x, n4.y, n4.z))), n4.w, e, c, G);
}
// the same from the camera-frame centre / normal (the tiled rasteriser keeps them per list entry)
__device__ __forceinline__ void surfel_geo_cam(v3 q, v3 nn, float r, unsigned int e, const Cam& c, SurfGeo& G)
{
    G.q = q;
    G.nn = nn;
    G.r = r;
    disc_extent(G.q, G.nn, G.r, c, G.xs, G.ys, G.minz);
    G.u = ((c.fx * G.q.x) / G.q.z) + c.cx; G.v = ((c.fy * G.q.y) / G.q.z) + c.cy;
    // splat region: GL point sprite (splat.vert:75-92)
    G.do_s = (e & LIST_SPLAT) != 0;
    G.sx0 = 0; G.sx1 = -1; G.sy0 = 0; G.sy1 = -1;
    if (G.do_s) {
        float s = fmaxf(fabsf(G.xs[1] - G.xs[0]), fabsf(G.ys[1] - G.ys[0]));
        if (!(s == s)) G.do_s = false;
        else {
            s = fminf(fmaxf(s, 1.0f), IFX_MAX_SPRITE);
            G.sx0 = clampi((int)ceilf(G.u - s * 0.5f - 0.5f), 0, c.w - 1); G.sx1 = clampi((int)floorf(G.u + s * 0.5f - 0.5f), 0, c.w - 1);
            G.sy0 = clampi((int)ceilf(G.v - s * 0.5f - 0.5f), 0, c.h - 1); G.sy1 = clampi((int)floorf(G.v + s * 0.5f - 0.5f), 0, c.h - 1);
        }
    }
}
// id region of the id render: bounding box of the quad (surfel_ids.geom:49-82); false = nothing to draw
__device__ __forceinline__ bool surfel_id_box(const SurfGeo& G, unsigned int e, const Cam& c, int& ix0, int& ix1, int& iy0, int& iy1)
{
    ix0 = 0; ix1 = -1; iy0 = 0; iy1 = -1;
    if (!(e & LIST_IDS)) return false;
    if (!(G.minz > 0) || !(G.xs[0] == G.xs[0]) || !(G.ys[0] == G.ys[0]) || G.xs[1] - G.xs[0] > IFX_MAX_SPRITE || G.ys[1] - G.ys[0] > IFX_MAX_SPRITE || G.xs[1] < 0 || G.ys[1] < 0 ||
        G.xs[0] > (float)c.w || G.ys[0] > (float)c.h)
        return false;
    ix0 = clampi((int)ceilf(G.xs[0] - 0.5f), 0, c.w - 1); ix1 = clampi((int)floorf(G.xs[1] - 0.5f), 0, c.w - 1);
    iy0 = clampi((int)ceilf(G.ys[0] - 0.5f), 0, c.h - 1); iy1 = clampi((int)floorf(G.ys[1] - 0.5f), 0, c.h - 1);
    return true;
}

// phase 2: disc rasterisation of the listed surfels into the splat and / or id key images
__global__ __launch_bounds__(MAP_THREADS) void k_raster_list(DevState* st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc,
                                                             const float4* __restrict__ nr, Cam c, const unsigned int* __restrict__ list,
                                                             unsigned long long* __restrict__ key_splat, unsigned long long* __restrict__ key_ids,
                                                             unsigned long long* __restrict__ key_both, int dual, const int* __restrict__ gate)
{
    const int fl = FIRST_LIVE(c);
    if (gate && !*gate) return;   // fallback launch behind the tiled rasteriser: only when its pair buffer overflowed
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    // block b works on segment b % LIST_SEGS (the segments hold interleaved chunks, so they are equally long up to one chunk): one length to read, no search
    const int seg = blockIdx.x % LIST_SEGS;
    const unsigned int n = *list_ctr(c, 0, seg);
    const unsigned int* __restrict__ seg_list = list + (size_t)seg * c.seg_cap;
    for (unsigned int t = (blockIdx.x / LIST_SEGS) * blockDim.x + threadIdx.x; t < n; t += blockDim.x * (gridDim.x / LIST_SEGS)) {
        const unsigned int e = seg_list[t];
        const unsigned int i = e & LIST_IDX, kid = key_id(c, i, fl);
        SurfGeo G;
        surfel_geo(T, pc[i], nr[i], e, c, G);
        const v3 q = G.q, nn = G.nn;
        const float r = G.r, u = G.u, v = G.v;
        const float* xs = G.xs;
        const float* ys = G.ys;
        const float minz = G.minz;
        bool do_s = G.do_s;
        int sx0 = G.sx0, sx1 = G.sx1, sy0 = G.sy0, sy1 = G.sy1;
        if (dual) {   // two splat renders: the LIST_IDS bit marks the INACTIVE one, same sprite region and depth rule, its own key image
            if (!(e & LIST_SPLAT)) {   // (do_s was computed for the SPLAT bit only)
                float s2 = fmaxf(fabsf(xs[1] - xs[0]), fabsf(ys[1] - ys[0]));
                if (!(s2 == s2)) continue;
                s2 = fminf(fmaxf(s2, 1.0f), IFX_MAX_SPRITE);
                sx0 = clampi((int)ceilf(u - s2 * 0.5f - 0.5f), 0, c.w - 1); sx1 = clampi((int)floorf(u + s2 * 0.5f - 0.5f), 0, c.w - 1);
                sy0 = clampi((int)ceilf(v - s2 * 0.5f - 0.5f), 0, c.h - 1); sy1 = clampi((int)floorf(v + s2 * 0.5f - 0.5f), 0, c.h - 1);
            } else if (!do_s) continue;
            const bool to_act = (e & LIST_SPLAT) != 0, to_old = (e & LIST_IDS) != 0;
            Disc dd;
            dd.q = q; dd.n = nn; dd.r2 = r * r;
            for (int py = sy0; py <= sy1; py++)
                for (int px = sx0; px <= sx1; px++) {
                    float z;
                    if (!disc_hit(dd, (float)px + 0.5f, (float)py + 0.5f, c, z)) continue;
                    if (!(z >= -c.maxDepth && z <= c.maxDepth)) continue;
                    if (to_act) key_min(&key_splat[py * c.w + px], make_key(z, kid));
                    if (to_old) key_min(&key_ids[py * c.w + px], make_key(z, kid));
                }
            continue;
        }
        int ix0, ix1, iy0, iy1;
        bool do_i = surfel_id_box(G, e, c, ix0, ix1, iy0, iy1);
        if (!do_s && !do_i) continue;
        if (!do_s) { sx0 = ix0; sx1 = ix1; sy0 = iy0; sy1 = iy1; }
        if (!do_i) { ix0 = sx0; ix1 = sx1; iy0 = sy0; iy1 = sy1; }
        const int x0 = min(sx0, ix0), x1 = max(sx1, ix1), y0 = min(sy0, iy0), y1 = max(sy1, iy1);
        Disc d;
        d.q = q; d.n = nn; d.r2 = r * r;
        for (int py = y0; py <= y1; py++)
            for (int px = x0; px <= x1; px++) {
                float z;
                if (!disc_hit(d, (float)px + 0.5f, (float)py + 0.5f, c, z)) continue;
                // The two renders share most pixels of most surfels and the kernel is bound by its atomics (half of them off:
                // 127 -> 56 us), so a pixel covered in both goes to a third image once; the resolve takes min(own, both).
                const bool in_s = do_s && px >= sx0 && px <= sx1 && py >= sy0 && py <= sy1 && (z >= -c.maxDepth && z <= c.maxDepth);
                const bool in_i = do_i && px >= ix0 && px <= ix1 && py >= iy0 && py <= iy1 && (z > 0 && z <= c.maxDepth);
                if (in_s && in_i) key_min(&key_both[py * c.w + px], make_key(z, kid));
                else if (in_s) key_min(&key_splat[py * c.w + px], make_key(z, kid));
                else if (in_i) key_min(&key_ids[py * c.w + px], make_key(z, kid));
            }
    }
}

// ------------------------------------------------------------------ tiled rasteriser
// k_raster_list is bound by its global 64-bit atomics (one per covered pixel and render; 105 us at VGA / 5M surfels, 0.9 ms at 1280x960 / 20M).  The
// tiled path bins the listed surfels by the 32x32-pixel tiles their discs touch and lets one workgroup per tile resolve its three key tiles
// (splat / ids / both) in LDS -- LDS atomics instead of L2 atomics, each key written to HBM once:
//   k_tile_count : per list entry the tile box of its disc (kept in tile_box) and a per-tile count -- block-local histogram in LDS, then one global add
//                  per block and non-empty tile (a direct global add per (entry, tile) would queue thousands of atomics on each of a few hundred words)
//   k_tile_scan  : exclusive scan of the tile counts (one block)
//   k_tile_fill  : the (tile, entry) pairs, grouped by tile: block-local ranks from an LDS histogram + one returning global add per block and tile
//   k_tile_raster: one block per tile; identical coverage / depth rules as k_raster_list (same surfel_geo), atomicMin on LDS keys, tiles stored whole
// More pairs than the pair buffer holds (a camera inside a cloud of huge discs): the scan raises a flag, the tile kernels return and k_raster_list,
// launched behind them with that flag as its gate, draws the frame with global atomics as before.
#define TILE 32
#define TILE_LOG 5
#define TILE_MAX 4096   // tiles per image the LDS histograms are sized for (2048 x 2048 pixels)
#define TILE_CHUNK 2048   // pairs one rasterising block draws; a tile with more pairs is shared by several blocks, which merge through atomicMin
struct TileRec { float qx, qy, qz, nx, ny, nz, r; unsigned int e; };   // camera-frame geometry of a list entry: the draw pass gathers one 32-B record instead of two 16-B ones and repeats no transform
struct TileArgs { unsigned int *tile_n, *tile_off, *tile_fill, *blk_off, *tile_box, *pairs; TileRec* recs; int* overflow; unsigned int pair_cap; int tw, th; };

__device__ __forceinline__ unsigned int tile_box_of(const float* T, const float4* __restrict__ pc, const float4* __restrict__ nr, unsigned int e, const Cam& c, TileRec& rec)
{
    SurfGeo G;
    const unsigned int i = e & LIST_IDX;
    surfel_geo(T, pc[i], nr[i], e, c, G);
    rec.qx = G.q.x; rec.qy = G.q.y; rec.qz = G.q.z; rec.nx = G.nn.x; rec.ny = G.nn.y; rec.nz = G.nn.z; rec.r = G.r; rec.e = e;
    int ix0, ix1, iy0, iy1;
    const bool do_i = surfel_id_box(G, e, c, ix0, ix1, iy0, iy1);
    if (!G.do_s && !do_i) return 0xFFFFFFFFu;
    int x0 = G.do_s ? G.sx0 : ix0, x1 = G.do_s ? G.sx1 : ix1, y0 = G.do_s ? G.sy0 : iy0, y1 = G.do_s ? G.sy1 : iy1;
    if (do_i) { x0 = min(x0, ix0); x1 = max(x1, ix1); y0 = min(y0, iy0); y1 = max(y1, iy1); }
    if (x1 < x0 || y1 < y0) return 0xFFFFFFFFu;
    return (unsigned int)(x0 >> TILE_LOG) | ((unsigned int)(y0 >> TILE_LOG) << 8) | ((unsigned int)(x1 >> TILE_LOG) << 16) | ((unsigned int)(y1 >> TILE_LOG) << 24);
}

__global__ __launch_bounds__(MAP_THREADS) void k_tile_count(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc,
                                                            const float4* __restrict__ nr, Cam c, const unsigned int* __restrict__ list, TileArgs ta)
{
    __shared__ unsigned int hist[TILE_MAX];
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    const int ntiles = ta.tw * ta.th;
    for (int k = threadIdx.x; k < ntiles; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const int seg = blockIdx.x % LIST_SEGS;
    const unsigned int n = *list_ctr(c, 0, seg);
    const unsigned int* __restrict__ seg_list = list + (size_t)seg * c.seg_cap;
    unsigned int* __restrict__ seg_box = ta.tile_box + (size_t)seg * c.seg_cap;
    TileRec* __restrict__ seg_rec = ta.recs + (size_t)seg * c.seg_cap;
    for (unsigned int t = (blockIdx.x / LIST_SEGS) * blockDim.x + threadIdx.x; t < n; t += blockDim.x * (gridDim.x / LIST_SEGS)) {
        TileRec rec;
        const unsigned int box = tile_box_of(T, pc, nr, seg_list[t], c, rec);
        seg_box[t] = box;
        if (box != 0xFFFFFFFFu) *reinterpret_cast<float4*>(&seg_rec[t]) = make_float4(rec.qx, rec.qy, rec.qz, rec.nx), *(reinterpret_cast<float4*>(&seg_rec[t]) + 1) = make_float4(rec.ny, rec.nz, rec.r, __uint_as_float(rec.e));
        if (box == 0xFFFFFFFFu) continue;
        const int tx0 = box & 255, ty0 = (box >> 8) & 255, tx1 = (box >> 16) & 255, ty1 = box >> 24;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&hist[ty * ta.tw + tx], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ntiles; k += blockDim.x)
        if (hist[k]) atomicAdd(&ta.tile_n[k], hist[k]);
}

__global__ __launch_bounds__(256) void k_tile_scan(TileArgs ta)
{
    __shared__ unsigned int part[256], bpart[256];
    const int ntiles = ta.tw * ta.th, per = (ntiles + 255) / 256, lo = threadIdx.x * per, hi = min(lo + per, ntiles);
    unsigned int sum = 0, bsum = 0;
    for (int k = lo; k < hi; k++) { const unsigned int v = ta.tile_n[k]; sum += v; bsum += (v + TILE_CHUNK - 1) / TILE_CHUNK; }
    part[threadIdx.x] = sum; bpart[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0, brun = 0;
        for (int k = 0; k < 256; k++) { const unsigned int v = part[k], b = bpart[k]; part[k] = run; bpart[k] = brun; run += v; brun += b; }
        *ta.overflow = run > ta.pair_cap ? 1 : 0;
        ta.blk_off[ntiles] = brun;   // number of rasterising blocks with work
    }
    __syncthreads();
    unsigned int run = part[threadIdx.x], brun = bpart[threadIdx.x];
    for (int k = lo; k < hi; k++) {
        const unsigned int v = ta.tile_n[k];
        ta.tile_off[k] = run; run += v;
        ta.blk_off[k] = brun; brun += (v + TILE_CHUNK - 1) / TILE_CHUNK;
        ta.tile_fill[k] = 0;
    }
}

__global__ __launch_bounds__(MAP_THREADS) void k_tile_fill(Cam c, const unsigned int* __restrict__ list, TileArgs ta)
{
    __shared__ unsigned int hist[TILE_MAX];   // phase 1: this block's count per tile; phase 2: its base inside the tile's range; phase 3: running rank
    __shared__ unsigned int base[TILE_MAX];
    if (*ta.overflow) return;
    const int ntiles = ta.tw * ta.th;
    for (int k = threadIdx.x; k < ntiles; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const int seg = blockIdx.x % LIST_SEGS;
    const unsigned int n = *list_ctr(c, 0, seg);
    const unsigned int* __restrict__ seg_list = list + (size_t)seg * c.seg_cap;
    const unsigned int* __restrict__ seg_box = ta.tile_box + (size_t)seg * c.seg_cap;
    const unsigned int t0 = (blockIdx.x / LIST_SEGS) * blockDim.x + threadIdx.x, stride = blockDim.x * (gridDim.x / LIST_SEGS);
    for (unsigned int t = t0; t < n; t += stride) {
        const unsigned int box = seg_box[t];
        if (box == 0xFFFFFFFFu) continue;
        const int tx0 = box & 255, ty0 = (box >> 8) & 255, tx1 = (box >> 16) & 255, ty1 = box >> 24;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&hist[ty * ta.tw + tx], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ntiles; k += blockDim.x) {
        const unsigned int cnt = hist[k];
        base[k] = ta.tile_off[k] + (cnt ? atomicAdd(&ta.tile_fill[k], cnt) : 0u);
        hist[k] = 0;
    }
    __syncthreads();
    for (unsigned int t = t0; t < n; t += stride) {
        const unsigned int box = seg_box[t];
        if (box == 0xFFFFFFFFu) continue;
        const unsigned int ri = (unsigned int)((size_t)seg * c.seg_cap + t);   // index of the entry's geometry record
        const int tx0 = box & 255, ty0 = (box >> 8) & 255, tx1 = (box >> 16) & 255, ty1 = box >> 24;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int k = ty * ta.tw + tx;
                ta.pairs[base[k] + atomicAdd(&hist[k], 1u)] = ri;
            }
    }
}

#define TILE_THREADS 512
__device__ __forceinline__ void tile_draw(float4 ra, float4 rb, const Cam& c, const int fl, int bx0, int by0, int bx1, int by1,
                                          unsigned long long* ks, unsigned long long* ki, unsigned long long* kb)
{
    const unsigned int e = __float_as_uint(rb.w), i = e & LIST_IDX;
    SurfGeo G;
    surfel_geo_cam(v3m(ra.x, ra.y, ra.z), v3m(ra.w, rb.x, rb.y), rb.z, e, c, G);
    int sx0 = G.sx0, sx1 = G.sx1, sy0 = G.sy0, sy1 = G.sy1, ix0, ix1, iy0, iy1;
    const bool do_s = G.do_s, do_i = surfel_id_box(G, e, c, ix0, ix1, iy0, iy1);
    if (!do_s && !do_i) return;
    if (!do_s) { sx0 = ix0; sx1 = ix1; sy0 = iy0; sy1 = iy1; }
    if (!do_i) { ix0 = sx0; ix1 = sx1; iy0 = sy0; iy1 = sy1; }
    const int x0 = max(min(sx0, ix0), bx0), x1 = min(max(sx1, ix1), bx1), y0 = max(min(sy0, iy0), by0), y1 = min(max(sy1, iy1), by1);
    Disc d;
    d.q = G.q; d.n = G.nn; d.r2 = G.r * G.r;
    for (int py = y0; py <= y1; py++)
        for (int px = x0; px <= x1; px++) {
            float z;
            if (!disc_hit(d, (float)px + 0.5f, (float)py + 0.5f, c, z)) continue;
            const bool in_s = do_s && px >= sx0 && px <= sx1 && py >= sy0 && py <= sy1 && (z >= -c.maxDepth && z <= c.maxDepth);
            const bool in_i = do_i && px >= ix0 && px <= ix1 && py >= iy0 && py <= iy1 && (z > 0 && z <= c.maxDepth);
            const int k = (py - by0) * TILE + (px - bx0);
            if (in_s && in_i) atomicMin(&kb[k], make_key(z, key_id(c, i, fl)));
            else if (in_s) atomicMin(&ks[k], make_key(z, key_id(c, i, fl)));
            else if (in_i) atomicMin(&ki[k], make_key(z, key_id(c, i, fl)));
        }
}
__global__ __launch_bounds__(TILE_THREADS) void k_tile_raster(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc,
                                                               const float4* __restrict__ nr, Cam c, TileArgs ta, unsigned long long* __restrict__ key_splat,
                                                               unsigned long long* __restrict__ key_ids, unsigned long long* __restrict__ key_both)
{
    __shared__ unsigned long long ks[TILE * TILE], ki[TILE * TILE], kb[TILE * TILE];
    const int fl = FIRST_LIVE(c);
    if (*ta.overflow) return;
    const int ntiles = ta.tw * ta.th;
    if (blockIdx.x >= ta.blk_off[ntiles]) return;   // the grid is sized for the worst case
    int lo = 0, hi = ntiles - 1;                     // last tile whose first block is <= blockIdx.x (empty tiles own no block: skip over equal offsets)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ta.blk_off[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int tile = lo, tx = tile % ta.tw, ty = tile / ta.tw;
    const unsigned int part = blockIdx.x - ta.blk_off[tile], nt = ta.tile_n[tile], nblk = (nt + TILE_CHUNK - 1) / TILE_CHUNK;
    const unsigned int off = ta.tile_off[tile] + part * TILE_CHUNK, n = min(nt - part * TILE_CHUNK, (unsigned int)TILE_CHUNK);
    for (int k = threadIdx.x; k < TILE * TILE; k += blockDim.x) { ks[k] = IFX_KEY_EMPTY; ki[k] = IFX_KEY_EMPTY; kb[k] = IFX_KEY_EMPTY; }
    __syncthreads();
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    const int bx0 = tx * TILE, by0 = ty * TILE, bx1 = bx0 + TILE - 1, by1 = by0 + TILE - 1;
    // four pairs per thread and round: their eight gathers are in flight together
    for (unsigned int p = threadIdx.x; p < n; p += 4 * blockDim.x) {
        unsigned int ri[4];
        float4 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const unsigned int q = p + u * blockDim.x; ri[u] = q < n ? ta.pairs[off + q] : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4* r4 = reinterpret_cast<const float4*>(&ta.recs[ri[u] == 0xFFFFFFFFu ? 0u : ri[u]]);
            ra[u] = r4[0]; rb[u] = r4[1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (ri[u] != 0xFFFFFFFFu) tile_draw(ra[u], rb[u], c, fl, bx0, by0, bx1, by1, ks, ki, kb);
    }
    __syncthreads();
    // the key images are all-empty between passes (the resolve clears what it reads): only winners are written -- plainly by a tile's only block,
    // through atomicMin when several blocks share the tile
    for (int k = threadIdx.x; k < TILE * TILE; k += blockDim.x) {
        const int px = bx0 + (k & (TILE - 1)), py = by0 + (k >> TILE_LOG);
        if (px < c.w && py < c.h) {
            const int g = py * c.w + px;
            const unsigned long long a = ks[k], b = ki[k], d = kb[k];
            if (nblk == 1) {
                if (a != IFX_KEY_EMPTY) key_splat[g] = a;
                if (b != IFX_KEY_EMPTY) key_ids[g] = b;
                if (d != IFX_KEY_EMPTY) key_both[g] = d;
            } else {
                if (a != IFX_KEY_EMPTY) atomicMin(&key_splat[g], a);
                if (b != IFX_KEY_EMPTY) atomicMin(&key_ids[g], b);
                if (d != IFX_KEY_EMPTY) atomicMin(&key_both[g], d);
            }
        }
    }
}

// phase 1 of the post-fuse pass: index-map projection (index_map.vert) + the cheap half of the clean
// rules (copy_unstable.vert:160-172) + the list of surfels that need the window test
__global__ __launch_bounds__(MAP_THREADS) void k_cull_clean(DevState* st, const float* __restrict__ pose_inv_ex, const float4* __restrict__ pc, const float2* __restrict__ tm,
                                                            Cam c, int time, unsigned long long* __restrict__ keys, unsigned int* __restrict__ list_cand,
                                                            unsigned int* __restrict__ list_kill)
{
    const int fl = FIRST_LIVE(c);
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    __shared__ BlockCount2 L2;
    if (threadIdx.x == 0) { L2.n[0] = 0; L2.n[1] = 0; }
    __syncthreads();
    const int n = st->count;
    const int lo = (int)((long long)n * c.srank / c.sn), hi = (int)((long long)n * (c.srank + 1) / c.sn);   // slice for the index projection only: the lists cover every slot
    for (int chunk = blockIdx.x; chunk * CHUNK_SLOTS < n; chunk += gridDim.x) {
        unsigned int pos[CHUNK_ROUNDS];
        unsigned int cmask = 0, kmask = 0;
#pragma unroll
        for (int r = 0; r < CHUNK_ROUNDS; r++) {
            int i = chunk * CHUNK_SLOTS + r * MAP_THREADS + threadIdx.x;
            bool cand = false, kill = false;
            if (i < n) {
                float2 t = tm[i];
                const float wv = t.y;
                // Inactive surfels (outside the time window, lastTime > 0) are neither projected
                // (index_map.vert:47) nor window-tested (copy_unstable.vert:107) and the age rule keeps them
                // (:171): they cost 8 B, their position is not even loaded.
                if (wv > DEAD_TIME && !(wv > 0.f && (float)time - wv > (float)c.timeDelta)) {
                    float4 p4 = pc[i];
                    if (!((float)time - wv > (float)c.timeDelta)) {
                        v3 p = xf_point(T, v3m(p4.x, p4.y, p4.z));
                        if (p.z > 0.f) {
                            float u = ((c.fx * p.x) / p.z) + c.cx, v = ((c.fy * p.y) / p.z) + c.cy;
                            if (!(p.z > c.maxDepth) && (u >= 0 && u < (float)c.w && v >= 0 && v < (float)c.h) && i >= lo && i < hi) {
                                const int px_ = point_pixel(u), py_ = point_pixel(v);
                                if (px_ >= 0 && py_ >= 0) key_min(&keys[py_ * c.w + px_], make_key(p.z, key_id(c, (unsigned int)i, fl)));
                            }
                            cand = ((float)time - wv < (float)c.timeDelta && u > 0 && v > 0 && u < (float)c.w && v < (float)c.h);
                        }
                    }
                    if (!cand) {   // count = zCount = 0: only the stability / age rules apply
                        int test = 1;
                        if (wv == -1 || (((float)time - wv) > 20 && p4.w < c.conf)) test = 0;
                        if (wv > 0 && (float)time - wv > (float)c.timeDelta) test = 1;
                        kill = !test;
                    }
                }
            }
            // a slot is on at most one of the two lists, so one position register serves both
            pos[r] = bcount2_reserve(L2, cand, kill);
            cmask |= cand ? (1u << r) : 0u;
            kmask |= kill ? (1u << r) : 0u;
        }
        __syncthreads();
        const int seg = chunk % LIST_SEGS;
        if (threadIdx.x < 2) { L2.base[threadIdx.x] = L2.n[threadIdx.x] ? atomicAdd(list_ctr(c, 1 + threadIdx.x, seg), L2.n[threadIdx.x]) : 0u; L2.n[threadIdx.x] = 0; }
        __syncthreads();
        const unsigned int bc = seg * c.seg_cap + L2.base[0], bk = seg * c.seg_cap + L2.base[1];
#pragma unroll
        for (int r = 0; r < CHUNK_ROUNDS; r++) {
            unsigned int i = (unsigned int)(chunk * CHUNK_SLOTS + r * MAP_THREADS + threadIdx.x);
            if (cmask & (1u << r)) list_cand[bc + pos[r]] = i;
            if (kmask & (1u << r)) list_kill[bk + pos[r]] = i;
        }
    }
}

// phase 2: window test (copy_unstable.vert:103-157) for the listed surfels, after the index map is resolved
__global__ __launch_bounds__(MAP_THREADS) void k_clean_list(DevState* st, const float* __restrict__ pose_inv_ex, Cam c, int time, float4* __restrict__ pc,
                                                            const float4* __restrict__ nr, float2* __restrict__ tm, const float4* __restrict__ tap,
                                                            const unsigned int* __restrict__ list_cand, const unsigned int* __restrict__ list_kill)
{
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    const int seg = blockIdx.x % LIST_SEGS;   // as k_raster_list: one segment of both lists per block
    const unsigned int nc = *list_ctr(c, 1, seg), nk = *list_ctr(c, 2, seg);
    list_cand += (size_t)seg * c.seg_cap; list_kill += (size_t)seg * c.seg_cap;
    int dead = 0;
    for (unsigned int t = (blockIdx.x / LIST_SEGS) * blockDim.x + threadIdx.x; t < nc + nk; t += blockDim.x * (gridDim.x / LIST_SEGS)) {
        bool del;
        unsigned int i;
        float2 tt;
        float4 p4;
        if (t < nc) {
            i = list_cand[t];
            tt = tm[i];
            p4 = pc[i];
            float lastT = tt.y;
            del = !clean_test(T, c, time, p4, nr[i], tt.x, lastT, tap);
        } else {
            i = list_kill[t - nc];
            tt = tm[i];
            p4 = pc[i];
            del = true;
        }
        if (del) {
            p4.w = -1.0f;
            pc[i] = p4;
            tm[i] = make_float2(tt.x, DEAD_TIME);
            dead++;
        }
    }
    dead = wave_sum_i(dead);
    if ((threadIdx.x & 63) == 0 && dead) atomicAdd(&st->n_dead, dead);
}


// The age rule of the clean pass (copy_unstable.vert:154-172: `lastTime == -1`, or unstable and unseen for more than 20 frames -> removed; kept whatever else says once
// `lastTime > 0 && time - lastTime > timeDelta`) for a slot NO view list holds.  The reference evaluates it at every frame; here such a slot is looked at when a scan or
// the walk of first_live comes by.  Nothing updates an unlisted slot, so whether one of the per-frame evaluations of [epoch, t] has removed it has a closed form: the
// rule fails on the frames T with 20 < T - lastTime (and, for lastTime > 0, T - lastTime <= timeDelta: from there on the slot is exempt FOR GOOD -- one evaluation at a
// late t would see the exemption and keep what the reference removed at age 21).  `epoch` = the first clean pass that saw the store in this state (0 for
// a map that grew here; the first frame after an upload: an uploaded surfel meets the rule from there on, whatever its times say about the frames before).
__device__ __forceinline__ bool age_rule_gone(float wv, float conf, int t, const Cam& c)
{
    if (t < c.age_epoch) return false;
    if (wv == -1.f) return true;
    if (!(conf < c.conf)) return false;
    float T = (float)t;
    if (wv > 0.f) {
        float Tc = floorf(wv + (float)c.timeDelta);          // the last frame at which the slot is not yet exempt
        if (Tc - wv > (float)c.timeDelta) Tc -= 1.f;
        T = fminf(T, Tc);
        if (T < (float)c.age_epoch) return false;
    }
    return T - wv > 20.f;
}
// ------------------------------------------------------------------ view list (frame path)
// The three culls of a frame (index map before the fusion, index map + clean after it, raster after the clean) all
// look at the map from the frame's pose, and consecutive frames look from almost the same pose.  k_cull_frame therefore
// streams the store ONCE (24 B per slot: times, position + confidence -- the id render has no time window, so every
// live position is needed) and keeps every slot whose disc can touch the image from ANY pose within VL_ROT of rotation and
// VL_TRANS of translation of the scan pose: the view list (list 3).  The passes of the following frames walk that list
// instead of the store -- new surfels are appended to it as they are created (k_append_scan) -- until the tracked pose
// leaves the margin, the list is VL_MAX_AGE frames old or the store was renumbered / moved (compaction, upload,
// deformation): the decision is taken on the device when the frame's pose is committed (vlist_decide), and a scan kernel that
// has nothing to do returns at its first instruction.  Results are those of the per-pass culls bit for bit: the list is a
// superset, every pass re-tests its entries exactly.
// Slots outside the list cannot be seen, matched or updated while it is valid; the one thing the clean pass would still do
// to them is the age rule (copy_unstable.vert:166: unstable and not seen for 20 frames -> removed).  Every scan applies it,
// as of the last clean pass, to every live slot before it builds the list (age_rule_gone).  Between two scans such a slot
// may outlive its deadline by a few frames, invisible to every pass -- but for one thing: it can be the map's first live
// slot, "surfel 0", so k_append_scan evaluates the same rule where it moves first_live on.  ifx_vlist_reap forces a scan
// before anything that looks at the whole store (count, download, compaction, labels, kNN, segmentation statistics).
__global__ void k_vlist_decide(DevState* st, unsigned int* lctr, int force)
{
    if (threadIdx.x != 0) return;
    if (force) st->vl_valid = 0;
    vlist_decide(st, lctr);
}

// can the disc of a surfel at camera-frame position q (scan pose) touch the image from a pose within the margins?
// Every point of the disc lies within `reach` of q; a point of the moved frustum lies within VL_TRANS + VL_ROT * |x| of the
// scan frustum; so q is within D = reach + VL_TRANS + VL_ROT * (|q| + reach + VL_TRANS) of the scan frustum, hence within D of
// each of its bounding half-spaces (conservative; 1 % + 1 cm slack for the arithmetic).
struct VlPlanes { float lx, lz, rx, rz, ty, tz, by, bz; };   // unit outward normals of the four side planes: (lx, 0, lz), (rx, 0, rz), (0, ty, tz), (0, by, bz)
__device__ __forceinline__ bool near_frustum(v3 q, float reach, const VlPlanes& P, float maxDepth)
{
    const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
    const float D = (reach + VL_TRANS + VL_ROT * (len + reach + VL_TRANS)) * 1.01f + 0.01f;
    if (!(len == len)) return true;
    return !(P.lx * q.x + P.lz * q.z > D || P.rx * q.x + P.rz * q.z > D || P.ty * q.y + P.tz * q.z > D || P.by * q.y + P.bz * q.z > D || -q.z > D || q.z - maxDepth > D);
}
static VlPlanes make_planes(const Cam& c)
{
    VlPlanes P;
    // u >= 0  <=>  fx x + cx z >= 0 ; u <= w  <=>  fx x + (cx - w) z <= 0 ; same in v
    float n = sqrtf(c.fx * c.fx + c.cx * c.cx);
    P.lx = -c.fx / n; P.lz = -c.cx / n;
    n = sqrtf(c.fx * c.fx + (c.cx - c.w) * (c.cx - c.w));
    P.rx = c.fx / n; P.rz = (c.cx - c.w) / n;
    n = sqrtf(c.fy * c.fy + c.cy * c.cy);
    P.ty = -c.fy / n; P.tz = -c.cy / n;
    n = sqrtf(c.fy * c.fy + (c.cy - c.h) * (c.cy - c.h));
    P.by = c.fy / n; P.bz = (c.cy - c.h) / n;
    return P;
}

// Two lists: LIST_V, the slots inside the time window at the scan (everything the index maps, the clean pass and the splat render can
// ever touch while the list lives: a slot outside the window stays outside -- only a deformation re-activates surfels, and that voids
// the list), and LIST_VI, the stable slots outside it, which only the id render draws (it has no time window).  Unstable slots outside
// the window are frozen (nothing updates them) and invisible to every render: they are on neither list.
__global__ __launch_bounds__(MAP_THREADS) void k_cull_frame(DevState* st, const float4* __restrict__ pc_in, float4* __restrict__ pc_rw, float2* __restrict__ tm, Cam c, VlPlanes P,
                                                            int time, unsigned int* __restrict__ list, unsigned int* __restrict__ list_i, Hot* __restrict__ hot, int time_prev)
{
    if (!st->vl_scan) return;
    float Tm[16], T[12];
#pragma unroll
    for (int k = 0; k < 16; k++) Tm[k] = st->vl_pose[k];
    {   // rigid inverse of the scan pose
        float inv[16];
        pose_inverse(Tm, inv);
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = inv[k];
    }
    __shared__ BlockCount2 L2;
    if (threadIdx.x == 0) { L2.n[0] = 0; L2.n[1] = 0; }
    __syncthreads();
    const int n = st->count;
    const float reach = __uint_as_float(st->r_max_bits) * 1.41421356f * 1.001f;
    int dead = 0;
    // A chunk of this scan is 8 rounds (2048 slots).  All sixteen loads of a chunk -- position + confidence and times of eight slots per thread -- are
    // in flight before the first test.  (Round by round, times then position, each load behind the previous test's branch, a thread of the first
    // version strung 32 dependent round trips together: ~50 us per scan of 5.2 M slots, 2.5 TB/s.)
    constexpr int CF_ROUNDS = 8, CF_SLOTS = MAP_THREADS * CF_ROUNDS;
    for (int chunk = blockIdx.x; chunk * CF_SLOTS < n; chunk += gridDim.x) {
        unsigned int pos[CF_ROUNDS];
        unsigned int keep = 0, keep_i = 0;
        float4 p4s[CF_ROUNDS];
        float2 ts[CF_ROUNDS];
#pragma unroll
        for (int r = 0; r < CF_ROUNDS; r++) {
            const int i = chunk * CF_SLOTS + r * MAP_THREADS + threadIdx.x;
            const int ii = i < n ? i : n - 1;
            p4s[r] = ld_once(&pc_in[ii]);
            ts[r] = ld_once(&tm[ii]);
        }
        asm volatile("" ::"v"(p4s[0].x), "v"(p4s[1].x), "v"(p4s[2].x), "v"(p4s[3].x), "v"(p4s[4].x), "v"(p4s[5].x), "v"(p4s[6].x), "v"(p4s[7].x), "v"(ts[0].x), "v"(ts[1].x), "v"(ts[2].x),
                     "v"(ts[3].x), "v"(ts[4].x), "v"(ts[5].x), "v"(ts[6].x), "v"(ts[7].x));
#pragma unroll
        for (int r = 0; r < CF_ROUNDS; r++) {
            const int i = chunk * CF_SLOTS + r * MAP_THREADS + threadIdx.x;
            bool in = false, in_i = false;
            if (i < n) {
                const float4 p4 = p4s[r];
                const float2 t = ts[r];
                const float wv = t.y;
                // A slot that NO list held during the last frames has not met the clean pass's age rule since the scan that left it out -- and may come into THIS list: every
                // live slot first meets the rule as the clean passes up to the last one (time_prev) would have applied it (age_rule_gone; for a slot the previous list did hold,
                // that list's clean passes applied the same rule at the same times: nothing changes).  (Without it such a slot lived on into this frame's association and, at a
                // forced scan before a download, into the map: tests/test_gpu_sweep.py, seed 2238 "shake" -- lists that survive several frames AND a map older than 20 frames.)
                // The rule of the frame BEING ENQUEUED is not applied here: the reference removes such a surfel at the END of that frame, and until then it can still be the
                // map's first live surfel ("surfel 0", whose id reads 0).  A slot this list leaves out stays as it is -- invisible -- until the next scan finds it overdue;
                // k_append_scan's first_live steps over it from the frame on in which the rule removes it (tests/test_gpu_sweep.py::test_surfel_0_outside_the_view_lists).
                const bool overdue = wv > DEAD_TIME && time_prev >= 0 && age_rule_gone(wv, p4.w, time_prev, c);
                if (overdue) {
                    float4 q4 = p4;
                    q4.w = -1.0f;
                    pc_rw[i] = q4;
                    tm[i] = make_float2(t.x, DEAD_TIME);
                    if (hot) { hot[i].pc = q4; hot[i].tm = make_float2(t.x, DEAD_TIME); }
                    dead++;
                } else if (wv > DEAD_TIME) {
                    in = near_frustum(xf_point(T, v3m(p4.x, p4.y, p4.z)), reach, P, c.maxDepth);
                    if (in && wv > 0.f && (float)time - wv > (float)c.timeDelta) {   // outside the time window for good
                        in = false;
                        in_i = !(p4.w < c.conf);   // (>=: the INACTIVE splat of the loop-closure detection draws a surfel exactly AT the threshold, the id render tests > per entry)
                    } else if (!in && time_prev < 0) {   // (option overdue_rule 0, round 4's scan: the age rule of THIS frame's clean pass, applied now, to the slots the list leaves out)
                        int test = 1;
                        if (wv == -1 || (((float)time - wv) > 20 && p4.w < c.conf)) test = 0;
                        if (wv > 0 && (float)time - wv > (float)c.timeDelta) test = 1;
                        if (!test) {
                            float4 q4 = p4;
                            q4.w = -1.0f;
                            pc_rw[i] = q4;
                            tm[i] = make_float2(t.x, DEAD_TIME);
                            if (hot) { hot[i].pc = q4; hot[i].tm = make_float2(t.x, DEAD_TIME); }
                            dead++;
                        }
                    }
                }
            }
            pos[r] = bcount2_reserve(L2, in, in_i);
            keep |= in ? (1u << r) : 0u;
            keep_i |= in_i ? (1u << r) : 0u;
        }
        __syncthreads();
        const int seg = chunk % LIST_SEGS;
        if (threadIdx.x < 2) { L2.base[threadIdx.x] = L2.n[threadIdx.x] ? atomicAdd(list_ctr(c, 1 + threadIdx.x, seg), L2.n[threadIdx.x]) : 0u; L2.n[threadIdx.x] = 0; }
        __syncthreads();
        const unsigned int ba = seg * c.seg_cap + L2.base[0], bi = seg * c.seg_cap + L2.base[1];
#pragma unroll
        for (int r = 0; r < CF_ROUNDS; r++) {
            const unsigned int i = (unsigned int)(chunk * CF_SLOTS + r * MAP_THREADS + threadIdx.x);
            if (keep & (1u << r)) list[ba + pos[r]] = i;
            if (keep_i & (1u << r)) list_i[bi + pos[r]] = i;
        }
    }
    dead = wave_sum_i(dead);
    if ((threadIdx.x & 63) == 0 && dead) atomicAdd(&st->n_dead, dead);
}

// ---- the scan's raw output (8 segments per list, each with its own counter so that ~2000 appending blocks do not queue on one word) is
// concatenated into two flat lists, to which k_append_scan adds the new surfels.  (Sorting the lists by screen tile of the scan pose
// was tried -- a wave then works inside one 32x32-pixel neighbourhood -- and changed nothing: raster 79 -> 84 us, clean 64 -> 67 us,
// index 28 -> 29 us.  The list passes are bound by their GATHERS FROM THE STORE, one 64-B line per field and entry whatever the order of
// the list: 476 k window entries + 640 k outside ones out of 5.4 M slots; profiles/archive/r02_l_kernel_stats_tile_sorted_lists.csv.  Compact copies of
// the hot fields in list order, kept coherent through the fusion update / clean / append, were tried next: raster 79 -> 74 us, clean 64 -> 54,
// index 28 -> 23, but the fusion update 13 -> 23 and 11 us per frame for the copies: no net gain, removed; profiles/archive/r02_p_kernel_stats_compact_view_cache.csv.
// What is left in these passes is their atomics and the clean pass's taps.)
__global__ void k_vlist_offsets(DevState* st, Cam c, const float2* __restrict__ tm)
{
    if (!st->vl_scan || threadIdx.x != 0) return;
    for (int which = 0; which < 2; which++) {
        unsigned int run = 0;
        for (int seg = 0; seg < LIST_SEGS; seg++) {
            unsigned int* ctr = list_ctr(c, 1 + which, seg);
            const unsigned int n = *ctr;
            *ctr = 0;                                            // re-armed for the clean pass / the next scan
            st->vl_seg_n[which * LIST_SEGS + seg] = n;
            st->vl_seg_off[which * LIST_SEGS + seg] = run;
            run += n;
        }
        st->vl_n[which] = run;
    }
    {   // the scan applied the age rule to slots outside the list: the lowest live slot may have moved on
        int f = st->first_live;
        const int n = st->count;
        while (f < n && !(tm[f].y > DEAD_TIME)) f++;
        st->first_live = f;
    }
}
__global__ __launch_bounds__(MAP_THREADS) void k_vlist_concat(const DevState* __restrict__ st, Cam c, const unsigned int* __restrict__ raw_a, const unsigned int* __restrict__ raw_i,
                                                              unsigned int* __restrict__ flat_a, unsigned int* __restrict__ flat_i)
{
    if (!st->vl_scan) return;
    const int which = blockIdx.y, seg = blockIdx.x % LIST_SEGS;
    const unsigned int n = st->vl_seg_n[which * LIST_SEGS + seg], off = st->vl_seg_off[which * LIST_SEGS + seg];
    const unsigned int* __restrict__ raw = (which ? raw_i : raw_a) + (size_t)seg * c.seg_cap;
    unsigned int* __restrict__ flat = (which ? flat_i : flat_a) + off;
    for (unsigned int t = (blockIdx.x / LIST_SEGS) * blockDim.x + threadIdx.x; t < n; t += blockDim.x * (gridDim.x / LIST_SEGS)) flat[t] = raw[t];
}
// The two launches above in one (option vlist_one; measured equal, off): every block adds up the eight segment counters of its list itself (the scan's atomics were performed at the
// memory side: plain loads behind the kernel boundary see them), copies its share, and the block that finishes LAST re-arms the counters, publishes the lengths and moves
// first_live on -- one launch less on the frame's chain.
__global__ __launch_bounds__(MAP_THREADS) void k_vlist_flatten(DevState* st, Cam c, const float2* __restrict__ tm, const unsigned int* __restrict__ raw_a, const unsigned int* __restrict__ raw_i,
                                                               unsigned int* __restrict__ flat_a, unsigned int* __restrict__ flat_i)
{
    if (!st->vl_scan) return;
    const int which = blockIdx.y, seg = blockIdx.x % LIST_SEGS;
    unsigned int n = 0, off = 0;
#pragma unroll
    for (int s_ = 0; s_ < LIST_SEGS; s_++) {
        const unsigned int v = *list_ctr(c, 1 + which, s_);
        off += s_ < seg ? v : 0u;
        n = s_ == seg ? v : n;
    }
    const unsigned int* __restrict__ raw = (which ? raw_i : raw_a) + (size_t)seg * c.seg_cap;
    unsigned int* __restrict__ flat = (which ? flat_i : flat_a) + off;
    for (unsigned int t = (blockIdx.x / LIST_SEGS) * blockDim.x + threadIdx.x; t < n; t += blockDim.x * (gridDim.x / LIST_SEGS)) flat[t] = raw[t];
    // the last block: everybody has read the counters (their values bound the loops above) before the ticket
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(&st->append_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x * gridDim.y - 1u);
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __hip_atomic_store(&st->append_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int w_ = 0; w_ < 2; w_++) {
        unsigned int run = 0;
        for (int s_ = 0; s_ < LIST_SEGS; s_++) {
            unsigned int* ctr = list_ctr(c, 1 + w_, s_);
            const unsigned int v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the clean pass / the next scan
            st->vl_seg_n[w_ * LIST_SEGS + s_] = v;
            st->vl_seg_off[w_ * LIST_SEGS + s_] = run;
            run += v;
        }
        st->vl_n[w_] = run;
    }
    {   // the scan applied the age rule to slots outside the list: the lowest live slot may have moved on
        int f = st->first_live;
        const int nn = st->count;
        while (f < nn && !(tm[f].y > DEAD_TIME)) f++;
        st->first_live = f;
    }
}

// index-map projection (index_map.vert:40-66) of the view-list entries: the work of k_index_project on the slots that can be seen at all
__global__ __launch_bounds__(MAP_THREADS) void k_index_list(const DevState* __restrict__ st, const float4* __restrict__ pc, const float2* __restrict__ tm, Cam c, int time,
                                                            const unsigned int* __restrict__ list, unsigned long long* __restrict__ keys, const Hot* __restrict__ hot = nullptr)
{
    const int fl = FIRST_LIVE(c);
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->pose_inv[k];
    const unsigned int n = min(st->vl_n[0], c.seg_cap * LIST_SEGS);
    // Two entries per thread and round, every load of a stage issued before the first use: list entries, then times AND positions (a position is
    // fetched even when the time test will reject the entry -- in the time-window list that is the rare case), then the atomics.  One entry at a
    // time with the early exits in front of each load was three dependent round trips per entry and two rounds per thread.
#ifndef INDEX_U
#define INDEX_U 1
#endif
    constexpr int U = INDEX_U;
    const unsigned int stride = blockDim.x * gridDim.x;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += U * stride) {
        unsigned int i[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const unsigned int tt = t + u * stride; in[u] = tt < n; i[u] = list[in[u] ? tt : t]; }
        float lastT[U];
        float4 p4[U];
#pragma unroll
        for (int u = 0; u < U; u++) { lastT[u] = hot ? hot[i[u]].tm.y : ld_once(&tm[i[u]]).y; p4[u] = hot ? hot[i[u]].pc : ld_once(&pc[i[u]]); }   // (hot records: both from one line)
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" ::"v"(lastT[u]), "v"(p4[u].x), "v"(p4[u].y), "v"(p4[u].z));   // (keeps the compiler from sinking a load behind the previous entry's branches)
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!in[u] || (float)time - lastT[u] > (float)c.timeDelta) continue;   // inactive or tombstone
            const v3 p = xf_point(T, v3m(p4[u].x, p4[u].y, p4[u].z));
            if (p.z > c.maxDepth || p.z < 0) continue;
            const float uu = ((c.fx * p.x) / p.z) + c.cx, v = ((c.fy * p.y) / p.z) + c.cy;
            if (!(uu >= 0 && uu < (float)c.w && v >= 0 && v < (float)c.h)) continue;
            const int px_ = point_pixel(uu), py_ = point_pixel(v);
            if (px_ < 0 || py_ < 0) continue;
            key_min(&keys[py_ * c.w + px_], make_key(p.z, key_id(c, i[u], fl)));   // (sharded map: the creation number instead of the slot)
        }
    }
}

// clean (copy_unstable.vert:103-174) over the view list, after the post-fuse index map is resolved: the decisions of
// k_cull_clean + k_clean_list for the listed slots.  A block classifies 256 entries at a time; the ones that need the
// 16-tap window test are gathered in LDS and tested with every lane busy.
__global__ __launch_bounds__(MAP_THREADS) void k_clean_view(DevState* st, Cam c, int time, float4* __restrict__ pc, const float4* __restrict__ nr, float2* __restrict__ tm,
                                                            const float4* __restrict__ tap, const unsigned int* __restrict__ list)
{
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->pose_inv[k];
    __shared__ unsigned int s_cand[2 * MAP_THREADS];
    __shared__ unsigned int s_n;
    const unsigned int n = min(st->vl_n[0], c.seg_cap * LIST_SEGS);
    const unsigned int* __restrict__ seg_list = list;
    const unsigned int stride = blockDim.x * gridDim.x;
    int dead = 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (unsigned int t0 = blockIdx.x * blockDim.x; t0 < n; t0 += stride) {
        const unsigned int t = t0 + threadIdx.x;
        bool cand = false;
        unsigned int i = 0;
        if (t < n) {
            i = seg_list[t];
            const float2 tt = ld_once(&tm[i]);
            const float4 p4 = ld_once(&pc[i]);   // (fetched with the times: in the time-window list nearly every entry needs it, and behind the test it was a third dependent round trip)
            asm volatile("" ::"v"(tt.y), "v"(p4.x), "v"(p4.y), "v"(p4.z), "v"(p4.w));
            const float wv = tt.y;
            if (wv > DEAD_TIME && !(wv > 0.f && (float)time - wv > (float)c.timeDelta)) {   // live and not exempt by the time window
                if (!((float)time - wv > (float)c.timeDelta)) {
                    const v3 p = xf_point(T, v3m(p4.x, p4.y, p4.z));
                    if (p.z > 0.f) {
                        const float u = ((c.fx * p.x) / p.z) + c.cx, v = ((c.fy * p.y) / p.z) + c.cy;
                        cand = ((float)time - wv < (float)c.timeDelta && u > 0 && v > 0 && u < (float)c.w && v < (float)c.h);
                    }
                }
                if (!cand) {   // count = zCount = 0: only the stability / age rules apply
                    int test = 1;
                    if (wv == -1 || (((float)time - wv) > 20 && p4.w < c.conf)) test = 0;
                    if (wv > 0 && (float)time - wv > (float)c.timeDelta) test = 1;
                    if (!test) {
                        float4 q4 = p4;
                        q4.w = -1.0f;
                        pc[i] = q4;
                        tm[i] = make_float2(tt.x, DEAD_TIME);
                        dead++;
                    }
                }
            }
        }
        {   // gather the candidates of this round
            const unsigned long long m = __ballot(cand);
            unsigned int base = 0;
            const int lane = threadIdx.x & 63;
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) base = atomicAdd(&s_n, (unsigned int)__popcll(m));
                base = __shfl(base, leader, 64);
            }
            if (cand) s_cand[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        }
        __syncthreads();
        // drain in full rounds of 256; the remainder waits for the next round (or the end)
        const bool last_round = t0 + stride >= n;
        unsigned int have = s_n;
        unsigned int done = 0;
        while (have - done >= blockDim.x || (last_round && have > done)) {
            const unsigned int k = done + threadIdx.x;
            if (k < have) {
                const unsigned int j = s_cand[k];
                const float2 tt = tm[j];
                const float4 p4 = pc[j];
                float lastT = tt.y;
                if (!clean_test(T, c, time, p4, nr[j], tt.x, lastT, tap)) {
                    float4 q4 = p4;
                    q4.w = -1.0f;
                    pc[j] = q4;
                    tm[j] = make_float2(tt.x, DEAD_TIME);
                    dead++;
                }
            }
            done += blockDim.x;
            if (done > have) done = have;
        }
        __syncthreads();
        if (done > 0 && done < have) {   // move the remainder to the front
            const unsigned int rest = have - done;
            unsigned int v = 0;
            if (threadIdx.x < rest) v = s_cand[done + threadIdx.x];
            __syncthreads();
            if (threadIdx.x < rest) s_cand[threadIdx.x] = v;
        }
        if (threadIdx.x == 0) s_n = have - done;
        __syncthreads();
    }
    dead = wave_sum_i(dead);
    if ((threadIdx.x & 63) == 0 && dead) atomicAdd(&st->n_dead, dead);
}

// Splat prediction + id render over the view list: the culls of k_cull_raster and the coverage / depth rules of
// k_raster_list (same surfel_geo, same disc_hit: the images are bit-identical), with the pixel work of a wave FLATTENED:
// every lane first prepares one listed surfel (cull, camera-frame geometry, pixel box), the boxes' areas are prefix-summed
// across the wave, and the wave then walks the concatenation of all candidate pixels 64 at a time -- lane l of step s
// tests pixel 64 s + l of that sequence, whichever surfel it belongs to.  In k_raster_list a lane walks its own box, so a
// wave runs as long as its largest box and its 64 atomics of a step go to 64 unrelated places; here every lane has work
// in every step and neighbouring lanes hit neighbouring pixels of the same few surfels.
// earlyz: a plain (agent-scope) read of the key before the atomic; a key that is not below what the image already held
// at ANY earlier time cannot win (keys only decrease during the pass), so the atomic is dropped -- with ~17 discs over
// every pixel of the benchmark map most of them lose.
struct alignas(16) RvRec { float qx, qy, qz, nx, ny, nz, r2; unsigned int id; int x0, y0, bw, excl; int s01, s23, i01, i23; };
// LDSMIN: the depth test of a wave's 64 surfels is first played out in LDS.  The entries of a wave are neighbours in creation order, i.e. (mostly)
// neighbouring pixels of the frame that created them, so their discs overlap each other: ~3 candidates per touched pixel inside one wave.  A
// wave-private direct-mapped table (16 x 16 pixel window x 2 targets, tag = pixel and target image) takes the minimum per pixel with LDS atomics;
// a candidate whose slot holds another pixel goes to global memory as before; after the walk the occupied slots are flushed with one global
// atomic each.  min is order-independent: the images are bit-identical.
#define RV_SLOTS 512
// CLEAN (option clean_raster, the frame path's default): the clean pass (k_clean_view) and the end-of-frame raster walk the same list and gather the same records
// (position + confidence, times, normal + radius of ~0.4 M entries, one 64-B line per field and entry at random map order: 221 + 201 MB of HBM traffic per frame,
// profiles/r04_zz_final_pmc_traffic.json).  Here ONE walk does both: a lane runs the stability test of copy_unstable.vert:103-174 on its entry (k_clean_view's two
// branches, same arithmetic), tombstones it if it fails, and rasterises it from the same registers if it survives -- a surfel's raster depends on its OWN clean
// verdict only (tombstones keep the slots in place; a deformation never takes this path).  The new surfels of the frame are not in the walk: they are appended
// behind it (k_append_scan), and the caller takes this path only while a new surfel's confidence cannot reach the drawing threshold (splat.vert:56-65,
// surfel_ids.vert:45) -- the reference draws them for nothing.  The first nf_blocks blocks of the grid are k_new_flags_count's: they read the same tap image and
// touch nothing the walk touches (5.9 us of their own on the frame's chain before).
// "Surfel 0": which slot is the first LIVE one is not settled while the walk removes surfels, so it draws slot numbers; k_append_scan's last block advances
// first_live as ever, and k_splat_resolve turns that slot's id into 0 (ties between keys are decided the same way: the first live slot is the lowest id either way).
struct Hot;
struct CleanArgs { float4* pc_rw; float2* tm_rw; const float4* tap; int nf_blocks; const uint32_t* assoc; const float4* mpc; const float4* mnr; int* flags; int* block_counts; Hot* hot; };
__device__ __forceinline__ void new_flags_body(DevState* st, const float* __restrict__ pose_inv_ex, const Cam& c, int time, const uint32_t* __restrict__ assoc, const float4* __restrict__ mpc,
                                               const float4* __restrict__ mnr, const float4* __restrict__ tap, int* __restrict__ flags, int* __restrict__ block_counts, int bid);
#ifndef WALK_MIN_WAVES
#define WALK_MIN_WAVES 1
#endif
template <bool LDSMIN, bool CLEAN>
__global__ __launch_bounds__(MAP_THREADS, CLEAN ? WALK_MIN_WAVES : 1) void k_raster_view(DevState* st, const float4* __restrict__ pc, const float4* __restrict__ nr, const float2* __restrict__ tm, Cam c,
                                                             int time, int maxTime, unsigned int want, const unsigned int* __restrict__ list_a, const unsigned int* __restrict__ list_i,
                                                             unsigned long long* __restrict__ key_splat, unsigned long long* __restrict__ key_ids,
                                                             unsigned long long* __restrict__ key_both, int earlyz, int ids_step, CleanArgs ca, const DevState* __restrict__ stc)
{
    // stc == st: what the walk READS of the state -- the pose, the list lengths, the radius bound -- comes through a const restrict parameter of its own.  With the clean's stores in
    // the kernel, the same loads through `st` compile to vector loads of one address by every one of the 16 k waves (the scalar cache is only used for memory the kernel provably
    // does not write), and they queue at the one L2 channel that holds the line.
    const int fl = FIRST_LIVE(c);
    if (CLEAN && (int)blockIdx.x < ca.nf_blocks) { new_flags_body(st, stc->pose_inv, c, time, ca.assoc, ca.mpc, ca.mnr, ca.tap, ca.flags, ca.block_counts, (int)blockIdx.x); return; }
    const unsigned int rblk = CLEAN ? blockIdx.x - (unsigned int)ca.nf_blocks : blockIdx.x, rgrid = CLEAN ? gridDim.x - (unsigned int)ca.nf_blocks : gridDim.x;
    __shared__ RvRec recs[MAP_THREADS / 64][64];
    __shared__ unsigned int s_tag[LDSMIN ? MAP_THREADS / 64 : 1][LDSMIN ? RV_SLOTS : 1];
    __shared__ unsigned long long s_key[LDSMIN ? MAP_THREADS / 64 : 1][LDSMIN ? RV_SLOTS : 1];
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = stc->pose_inv[k];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (LDSMIN) {
#pragma unroll
        for (int q = 0; q < RV_SLOTS / 64; q++) { s_tag[wid][q * 64 + lane] = 0u; s_key[wid][q * 64 + lane] = ~0ull; }
    }
    // the two view lists, one after the other: [0, na) the time-window list, [na, na + ni) the stable slots outside the window (id render only)
    // want & LIST_DUAL (the loop-closure detection's two splat renders, as k_cull_raster / k_raster_list's dual mode): the ACTIVE prediction goes to key_splat, the INACTIVE
    // one -- last seen at or before time - timeDelta -- to key_ids with the SPLAT geometry and depth rule; every entry of both lists is classified by its own time stamp
    const bool dual = (want & LIST_DUAL) != 0;
    const unsigned int na = min(stc->vl_n[0], c.seg_cap * LIST_SEGS), n = na + (((want & LIST_IDS) || dual) ? min(stc->vl_n[1], c.seg_cap * LIST_SEGS) : 0u);
    const unsigned int* __restrict__ seg_a = list_a;
    const unsigned int* __restrict__ seg_i = list_i;
    const float reach = __uint_as_float(stc->r_max_bits) * 1.41421356f * 1.001f;
    const unsigned int stride = blockDim.x * rgrid;
    int dead = 0;
#ifndef WALK_PIPE
#define WALK_PIPE 1
#endif
    // WALK_PIPE: a wave walks several chunks of 64 entries, and the first two dependent round trips of a chunk -- the list entry, then position + times of that slot -- are
    // issued one and two chunks AHEAD (the entry of chunk k + 2 and the records of chunk k + 1 leave before chunk k is processed), so that only the normal / tap gathers and the
    // pixel walk of a chunk are exposed.  (One chunk per wave and 4396 blocks was the round-4 form: every chunk then paid its four round trips in a row.)
    const unsigned int tfirst = rblk * blockDim.x + wid * 64 + lane;
    auto entry_of = [&](unsigned int t_) -> unsigned int { return t_ < n ? (t_ < na ? seg_a[t_] : seg_i[t_ - na]) : 0u; };
    unsigned int i_cur = 0, i_nxt = 0;
    float4 p4_cur = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 tt_cur = make_float2(0.f, 0.f);
    const Hot* const hot = CLEAN ? ca.hot : (const Hot*)nullptr;   // (option hot_records: position, normal and times of a slot in one line)
    if (WALK_PIPE) {
        i_cur = entry_of(tfirst);
        i_nxt = entry_of(tfirst + stride);
        if (tfirst < n) { p4_cur = hot ? hot[i_cur].pc : ld_once(&pc[i_cur]); if (tfirst < na || dual) tt_cur = hot ? hot[i_cur].tm : ld_once(&tm[i_cur]); }
    }
    for (unsigned int t0 = rblk * blockDim.x + wid * 64; t0 < n; t0 += stride) {   // a wave owns 64 consecutive entries: no block barrier anywhere
        const unsigned int t = t0 + lane;
        int area = 0;
        RvRec R;
        R.bw = 1;
        unsigned int i_pre = 0;
        float4 p4_pre = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 tt_pre = make_float2(0.f, 0.f);
        if (WALK_PIPE) {   // take this chunk's prefetched values, send the next ones on their way
            i_pre = i_cur; p4_pre = p4_cur; tt_pre = tt_cur;
            const unsigned int tn = t + stride;
            i_cur = i_nxt;
            if (tn < n) {
                p4_cur = hot ? hot[i_nxt].pc : ld_once(&pc[i_nxt]);
                tt_cur = (tn < na || dual) ? (hot ? hot[i_nxt].tm : ld_once(&tm[i_nxt])) : make_float2(0.f, 0.f);   // (an entry of the stable list carries no times: 0, never the previous chunk's)
            }
            i_nxt = entry_of(tn + stride);
        }
        if (t < n) {
            const unsigned int i = WALK_PIPE ? i_pre : (t < na ? seg_a[t] : seg_i[t - na]);
            float4 p4 = WALK_PIPE ? p4_pre : (hot ? hot[i].pc : ld_once(&pc[i]));
            const float2 tt = WALK_PIPE ? tt_pre : ((t < na || dual) ? (hot ? hot[i].tm : ld_once(&tm[i])) : make_float2(0.f, 0.f));   // (with the position: one round trip for both; the id render has no time window: no load for the stable list)
            const float lastT = tt.y;
            asm volatile("" ::"v"(lastT), "v"(p4.x), "v"(p4.y), "v"(p4.z), "v"(p4.w));
            float4 n4c = make_float4(0.f, 0.f, 0.f, 0.f);
            bool have_n = false;
            if (CLEAN && t < na) {   // k_clean_view for this entry: the window test for the candidates, the stability / age rules for the rest
                const float wv = tt.y;
                if (wv > DEAD_TIME && !(wv > 0.f && (float)time - wv > (float)c.timeDelta)) {   // live and not exempt by the time window
                    bool cand = false;
                    if (!((float)time - wv > (float)c.timeDelta)) {
                        const v3 p = xf_point(T, v3m(p4.x, p4.y, p4.z));
                        if (p.z > 0.f) {
                            const float u = ((c.fx * p.x) / p.z) + c.cx, v = ((c.fy * p.y) / p.z) + c.cy;
                            cand = ((float)time - wv < (float)c.timeDelta && u > 0 && v > 0 && u < (float)c.w && v < (float)c.h);
                        }
                    }
                    int test = 1;
                    if (cand) {
                        n4c = hot ? hot[i].nr : ld_once(&nr[i]);
                        have_n = true;
                        float lt2 = tt.y;
                        test = clean_test(T, c, time, p4, n4c, tt.x, lt2, ca.tap);
                    } else {   // count = zCount = 0: only the stability / age rules apply
                        if (wv == -1 || (((float)time - wv) > 20 && p4.w < c.conf)) test = 0;
                        if (wv > 0 && (float)time - wv > (float)c.timeDelta) test = 1;
                    }
                    if (!test) {
                        p4.w = -1.0f;   // (a tombstone's confidence: the raster below leaves it out, as it would after k_clean_view)
                        ca.pc_rw[i] = p4;
                        ca.tm_rw[i] = make_float2(tt.x, DEAD_TIME);
                        if (ca.hot) { ca.hot[i].pc = p4; ca.hot[i].tm = make_float2(tt.x, DEAD_TIME); }
                        dead++;
                    }
                }
            }
            unsigned int flags = 0;
            if (!(p4.w < c.conf)) {   // tombstones carry confidence -1
                const v3 q = xf_point(T, v3m(p4.x, p4.y, p4.z));
                if (q.z > 0.f && may_touch_image(reach, q, c)) {
                    if (dual) {
                        if (!(q.z > c.maxDepth)) {
                            const bool act = !((float)time - lastT > (float)c.timeDelta || lastT > (float)maxTime);
                            const bool old = !(0.f - lastT > (float)c.timeDelta || lastT > (float)(time - c.timeDelta));
                            if (act || old) {
                                const float u = ((c.fx * q.x) / q.z) + c.cx, v = ((c.fy * q.y) / q.z) + c.cy;
                                if (u >= 0 && u <= (float)c.w && v >= 0 && v <= (float)c.h) flags |= (act ? LIST_SPLAT : 0u) | (old ? LIST_IDS : 0u);
                            }
                        }
                    } else {
                    if ((want & LIST_IDS) && (p4.w > c.conf) && (q.z / c.maxDepth > 0.01f)) flags |= LIST_IDS;
                    if ((want & LIST_SPLAT) && !(q.z > c.maxDepth) && !((float)time - lastT > (float)c.timeDelta || lastT > (float)maxTime)) {
                        const float u = ((c.fx * q.x) / q.z) + c.cx, v = ((c.fy * q.y) / q.z) + c.cy;   // exact: GL clips points by their centre
                        if (u >= 0 && u <= (float)c.w && v >= 0 && v <= (float)c.h) flags |= LIST_SPLAT;
                    }
                    }
                }
            }
            if (flags) {
                SurfGeo G;
                surfel_geo(T, p4, (CLEAN && have_n) ? n4c : (hot ? hot[i].nr : ld_once(&nr[i])), dual ? (i | LIST_SPLAT) : (i | flags), c, G);   // (dual: the sprite region for whichever render draws it)
                int sx0 = G.sx0, sx1 = G.sx1, sy0 = G.sy0, sy1 = G.sy1, ix0, ix1, iy0, iy1;
                bool do_i = dual ? (G.do_s && (flags & LIST_IDS)) : surfel_id_box(G, i | flags, c, ix0, ix1, iy0, iy1);
                const bool do_s = dual ? (G.do_s && (flags & LIST_SPLAT)) : G.do_s;
                if (dual) { ix0 = sx0; ix1 = sx1; iy0 = sy0; iy1 = sy1; }
                int lattice = 0;
                if (do_i && ids_step > 1) {   // sparse id render: only the pixels of the ids_step lattice (what whetherDoSegmentation samples); the full image is rendered on demand
                    const int lx0 = ((ix0 + ids_step - 1) / ids_step) * ids_step, lx1 = (ix1 / ids_step) * ids_step;
                    const int ly0 = ((iy0 + ids_step - 1) / ids_step) * ids_step, ly1 = (iy1 / ids_step) * ids_step;
                    if (lx0 > lx1 || ly0 > ly1) do_i = false;
                    else { ix0 = lx0; ix1 = lx1; iy0 = ly0; iy1 = ly1; lattice = !do_s; }   // an id-only entry walks the lattice points of its box, nothing else
                }
                if (do_s || do_i) {
                    // a render that does not draw this surfel gets an empty box (x1 < x0): no pixel passes its range test
                    if (!do_s) { sx0 = 1; sx1 = 0; sy0 = 1; sy1 = 0; }
                    if (!do_i) { ix0 = 1; ix1 = 0; iy0 = 1; iy1 = 0; }
                    const int x0 = do_s && do_i ? min(sx0, ix0) : (do_s ? sx0 : ix0), x1 = do_s && do_i ? max(sx1, ix1) : (do_s ? sx1 : ix1);
                    const int y0 = do_s && do_i ? min(sy0, iy0) : (do_s ? sy0 : iy0), y1 = do_s && do_i ? max(sy1, iy1) : (do_s ? sy1 : iy1);
                    if (x1 >= x0 && y1 >= y0) {
                        const int nx_ = lattice ? (x1 - x0) / ids_step + 1 : x1 - x0 + 1, ny_ = lattice ? (y1 - y0) / ids_step + 1 : y1 - y0 + 1;
                        area = nx_ * ny_;
                        R.qx = G.q.x; R.qy = G.q.y; R.qz = G.q.z; R.nx = G.nn.x; R.ny = G.nn.y; R.nz = G.nn.z; R.r2 = G.r * G.r; R.id = CLEAN ? i : key_id(c, i, fl);   // (CLEAN: slot numbers -- first_live is not settled while this walk removes surfels; k_splat_resolve names "surfel 0")
                        R.x0 = x0; R.y0 = y0; R.bw = nx_ | (lattice << 16);
                        R.s01 = (sx0 & 0xFFFF) | (sx1 << 16); R.s23 = (sy0 & 0xFFFF) | (sy1 << 16);
                        R.i01 = (ix0 & 0xFFFF) | (ix1 << 16); R.i23 = (iy0 & 0xFFFF) | (iy1 << 16);
                    }
                }
            }
        }
        int incl = area;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        const int excl = incl - area, total = __shfl(incl, 63, 64);
        if (total == 0) continue;
        R.excl = excl;
        if (area) recs[wid][lane] = R;
        // (single wave: LDS writes are ordered before the reads below by the wave's own program order + s_waitcnt)
        int jlo = 0;
        for (int g0 = 0; g0 < total; g0 += 64) {
            while (jlo < 63 && __builtin_amdgcn_readlane(incl, __builtin_amdgcn_readfirstlane(jlo)) <= g0) jlo++;
            const int g = g0 + lane;
            int mine = jlo;
            for (int j = jlo + 1; j < 64; j++) {
                const int ej = __builtin_amdgcn_readlane(excl, __builtin_amdgcn_readfirstlane(j));
                if (ej >= g0 + 64) break;
                if (g >= ej) mine = j;
            }
            if (g >= total) continue;
            const RvRec* rp = &recs[wid][mine];
            const float4 a = *reinterpret_cast<const float4*>(&rp->qx), b = *reinterpret_cast<const float4*>(&rp->ny);
            const int4 bx = *reinterpret_cast<const int4*>(&rp->x0), rg = *reinterpret_cast<const int4*>(&rp->s01);
            const int k = g - bx.w;
            const int bwv = bx.z & 0xFFFF, stp = (bx.z >> 16) ? ids_step : 1;
            const int row = (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)bwv)), col = k - row * bwv;   // exact for boxes up to 512 x 512 (IFX_MAX_SPRITE)
            const int px = bx.x + col * stp, py = bx.y + row * stp;
            Disc d;
            d.q = v3m(a.x, a.y, a.z); d.n = v3m(a.w, b.x, b.y); d.r2 = b.z;
            const unsigned int id = __float_as_uint(b.w);
            float z;
            if (!disc_hit(d, (float)px + 0.5f, (float)py + 0.5f, c, z)) continue;
            const int sx0 = (short)(rg.x & 0xFFFF), sx1 = rg.x >> 16, sy0 = (short)(rg.y & 0xFFFF), sy1 = rg.y >> 16;
            const int ix0 = (short)(rg.z & 0xFFFF), ix1 = rg.z >> 16, iy0 = (short)(rg.w & 0xFFFF), iy1 = rg.w >> 16;
            const bool in_s = px >= sx0 && px <= sx1 && py >= sy0 && py <= sy1 && (z >= -c.maxDepth && z <= c.maxDepth);
            const bool in_i = px >= ix0 && px <= ix1 && py >= iy0 && py <= iy1 && (dual ? (z >= -c.maxDepth && z <= c.maxDepth) : (z > 0 && z <= c.maxDepth)) &&
                              (ids_step <= 1 || (px % ids_step == 0 && py % ids_step == 0));
            if (!in_s && !in_i) continue;
            if (dual && in_s && in_i) {   // (last seen exactly timeDelta frames ago: in both renders; their resolves do not read key_both)
                key_min(key_splat + (py * c.w + px), make_key(z, id));
                key_min(key_ids + (py * c.w + px), make_key(z, id));
                continue;
            }
            const int tsel = in_s && in_i ? 0 : (in_s ? 1 : 2);
            unsigned long long* addr = (tsel == 0 ? key_both : (tsel == 1 ? key_splat : key_ids)) + (py * c.w + px);
            const unsigned long long key = make_key(z, id);
            if (LDSMIN) {
                const unsigned int tag = ((unsigned int)(py * c.w + px) << 2 | (unsigned int)tsel) + 1u;
                const int slot = ((py & 15) << 5) | ((px & 15) << 1) | (tsel != 0);
                const unsigned int old = atomicCAS(&s_tag[wid][slot], 0u, tag);
                if (old == 0u || old == tag) { atomicMin(&s_key[wid][slot], key); continue; }
            }
            if (earlyz && !(key < __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) continue;
            key_min(addr, key);
        }
        if (LDSMIN) {   // flush: one global atomic per touched (pixel, target); the table is left empty for the next round
#pragma unroll
            for (int q = 0; q < RV_SLOTS / 64; q++) {
                const int sl = q * 64 + lane;
                const unsigned int tag = s_tag[wid][sl];
                if (tag) {
                    const unsigned long long key = s_key[wid][sl];
                    const unsigned int pt = tag - 1u, tsel = pt & 3u, pix = pt >> 2;
                    key_min((tsel == 0 ? key_both : (tsel == 1 ? key_splat : key_ids)) + pix, key);
                    s_tag[wid][sl] = 0u; s_key[wid][sl] = ~0ull;
                }
            }
        }
    }
    if (CLEAN) {
        dead = wave_sum_i(dead);
        if (lane == 0 && dead) atomicAdd(&st->n_dead, dead);
    }
}

// splat prediction (want & LIST_SPLAT) and / or id render (want & LIST_IDS) in one cull + one dense raster pass
static void raster_pass(ifx* h, const float* d_pose_inv, int time, int maxTime, unsigned int want, int32_t* ids_out, bool frame_sums = false, int part = 0, int old_target = 0,
                        bool fold_finish = false, int resolve_ids_step = 1, bool raw_ids = false)
{
    Cam c = make_cam(h);
    if (part == 0) { c.srank = 0; c.sn = 1; }   // a whole pass (stage API, re-render after a compaction) is never sliced
    if (part != 2) {
        const int tw = cdiv(h->w, TILE), th = cdiv(h->h, TILE);
        const bool want_tiles = h->opt_raster_tiles < 0 ? (h->P >= 1000000) : (h->opt_raster_tiles != 0);
        if (want_tiles && !h->tile_recs && hipMalloc(&h->tile_recs, (size_t)h->list_seg_cap * LIST_SEGS * 32) != hipSuccess) h->tile_recs = nullptr;   // 32 B per list entry, on first use
        const bool tiled = want_tiles && tw * th <= TILE_MAX && tw <= 255 && th <= 255 && h->tile_pairs && h->tile_recs;
        LAUNCH(h, "cull_raster", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_cull_raster, h->d_state, d_pose_inv, (const float4*)h->pc, (const float2*)h->tm, c, time, maxTime, want,
               h->list_a, tiled ? h->tile_n : (unsigned int*)nullptr, tw * th);
        if (tiled) {
            TileArgs ta;
            ta.tile_n = h->tile_n; ta.tile_off = h->tile_n + TILE_MAX; ta.tile_fill = h->tile_n + 2 * TILE_MAX; ta.blk_off = h->tile_n + 3 * TILE_MAX; ta.overflow = (int*)(h->tile_n + 4 * TILE_MAX + 8);
            ta.tile_box = h->tile_box; ta.pairs = h->tile_pairs; ta.recs = (TileRec*)h->tile_recs; ta.pair_cap = h->tile_pair_cap; ta.tw = tw; ta.th = th;
            LAUNCH(h, "tile_count", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_tile_count, (const DevState*)h->d_state, d_pose_inv, (const float4*)h->pc, (const float4*)h->nr, c, h->list_a, ta);
            LAUNCH(h, "tile_scan", dim3(1), dim3(256), k_tile_scan, ta);
            LAUNCH(h, "tile_fill", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_tile_fill, c, h->list_a, ta);
            LAUNCH(h, "tile_raster", dim3(tw * th + h->tile_pair_cap / TILE_CHUNK), dim3(TILE_THREADS), k_tile_raster, (const DevState*)h->d_state, d_pose_inv, (const float4*)h->pc, (const float4*)h->nr, c, ta, h->key_splat,
                   h->key_ids, h->key_both);
            LAUNCH(h, "raster_list", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_raster_list, h->d_state, d_pose_inv, (const float4*)h->pc, (const float4*)h->nr, c, h->list_a, h->key_splat,
                   h->key_ids, h->key_both, 0, (const int*)ta.overflow);
        } else
            LAUNCH(h, "raster_list", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_raster_list, h->d_state, d_pose_inv, (const float4*)h->pc, (const float4*)h->nr, c, h->list_a, h->key_splat,
                   h->key_ids, h->key_both, 0, (const int*)nullptr);
    }
    if (part == 1) return;
    if ((want & LIST_SPLAT) && old_target) {   // loop-closure renders: 1 = INACTIVE prediction into the old* images (IndexMap::oldFrameBuffer, EF/IndexMap.cpp:480-483),
        const bool old = old_target == 1;       // 2 = the predict() of EF/ElasticFusion.cpp:453 into the act* images; no fill-in, no dense flag
        LAUNCH(h, old ? "splat_resolve_old" : "splat_resolve_act", dim3(cdiv(h->w, 32), cdiv(h->h, 8)), dim3(32, 8), k_splat_resolve, h->d_state, d_pose_inv, h->key_splat,
               (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->col, (const float2*)h->tm, c, h->rgb, h->depth_filt, (float4*)(old ? h->old_vertex : h->act_vertex),
               (float4*)(old ? h->old_normal : h->act_normal), (uchar4*)(old ? h->old_image : h->act_image), (uchar4*)(old ? h->old_inst : h->act_inst),
               old ? h->old_time : h->act_time, (float4*)nullptr, (float4*)nullptr, (uchar4*)nullptr, h->key_ids, h->key_both, (int32_t*)nullptr, (int*)nullptr, FinishFold());
        LAUNCH(h, "raster_finish", dim3(1), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, 0, h->ids_after, (const float4*)h->votes, h->cap, 10, h->d_list_ctr, ifx_idmap(h));
        return;
    }
    if (want & LIST_SPLAT) {
        FinishFold ff = FinishFold();
        if (fold_finish && frame_sums && !h->own && h->w >= 20 && h->h >= 20) {
            ff.acc = &h->d_state->fold_acc[0][0]; ff.total = &h->d_state->fold_total;
            ff.votes = (const float4*)h->votes; ff.cap = h->cap; ff.ds = 10; ff.rw = h->w / 20; ff.rh = h->h / 20;
        }
        FrameOut fo = FrameOut{nullptr, nullptr, 0};
        if (ff.acc && h->result_fold_traj && h->opt_fold_result) {   // this resolve is the frame's last launch (enqueue_frame asked): its last block writes the frame result
            fo.out = h->h_result; fo.traj = h->result_fold_traj; fo.fold_total = ff.rw * ff.rh;
            ff.total = nullptr;
            h->result_folded = 1;
        }
        if (raw_ids) {   // option clean_raster: the walk drew slot numbers
            c.raw_slots = 2;
            LAUNCH(h, "splat_resolve", dim3(cdiv(h->w, 32), cdiv(h->h, 8)), dim3(32, 8), k_splat_resolve, h->d_state, d_pose_inv, h->key_splat, (const float4*)h->pc,
                   (const float4*)h->nr, (const float2*)h->col, (const float2*)h->tm, c, h->rgb, h->depth_filt, (float4*)h->pred_vertex, (float4*)h->pred_normal,
                   (uchar4*)h->pred_image, (uchar4*)h->pred_inst, h->pred_time, (float4*)h->fill_vertex, (float4*)h->fill_normal, (uchar4*)h->fill_image, h->key_ids,
                   h->key_both, (want & LIST_IDS) ? ids_out : (int32_t*)nullptr, (int*)nullptr, ff, (float*)nullptr, resolve_ids_step, (const int32_t*)nullptr, fo, (const Hot*)h->frame_hot);
            if (ff.acc) return;
        } else
        LAUNCH(h, "splat_resolve", dim3(cdiv(h->w, 32), cdiv(h->h, 8)), dim3(32, 8), k_splat_resolve, h->d_state, d_pose_inv, h->key_splat, (const float4*)h->pc,
               (const float4*)h->nr, (const float2*)h->col, (const float2*)h->tm, c, h->rgb, h->depth_filt, (float4*)h->pred_vertex, (float4*)h->pred_normal,
               (uchar4*)h->pred_image, (uchar4*)h->pred_inst, h->pred_time, (float4*)h->fill_vertex, (float4*)h->fill_normal, (uchar4*)h->fill_image, h->key_ids,
               h->key_both, (want & LIST_IDS) ? ids_out : (int32_t*)nullptr, (int*)nullptr, ff, (float*)nullptr, resolve_ids_step, (const int32_t*)nullptr, fo);
        if (ff.acc) return;   // (the view-list pass leaves list 0 alone: nothing to re-arm)
    } else if (want & LIST_IDS) LAUNCH(h, "ids_resolve", dim3(cdiv(h->P, 256)), dim3(256), k_ids_resolve, h->key_ids, h->P, ids_out);
    // dense flag, list re-arm and (frame path: the id image is ids_after) the whetherDoSegmentation sums
    const int seg = frame_sums && (want & LIST_IDS);   // only the frame's own render feeds whetherDoSegmentation (not the re-render after a compaction)
    const int ds = 10, nseg = seg ? cdiv(cdiv(h->w, ds) * cdiv(h->h, ds), 256) : 0;
    LAUNCH(h, "raster_finish", dim3(1 + nseg), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, (want & LIST_SPLAT) ? 1 : 0, h->ids_after,
           (const float4*)h->votes, h->cap, ds, h->d_list_ctr, ifx_idmap(h));
}
static void splat_pass(ifx* h, const float* d_pose_inv, int time, int maxTime) { raster_pass(h, d_pose_inv, time, maxTime, LIST_SPLAT, nullptr); }

static void ids_pass(ifx* h, const float* d_pose_inv, int mode, int32_t* out)
{
    if (mode != 1) { raster_pass(h, d_pose_inv, 0, 0, LIST_IDS, out); return; }
    Cam c = make_cam(h);   // INSTANCECOMPARE also reads the 192 B of votes: one-kernel version
    LAUNCH(h, "ids_raster_inst", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_raster<2>, h->d_state, d_pose_inv, (const float4*)h->pc, (const float4*)h->nr,
           (const float2*)h->tm, (const float4*)h->votes, h->cap, c, 0, 0, h->key_ids);
    LAUNCH(h, "ids_resolve", dim3(cdiv(h->P, 256)), dim3(256), k_ids_resolve, h->key_ids, h->P, out);
}

// The frame path renders the id image on the sampled lattice only (ifx_map_predict); whoever needs the whole image -- a segmentation call, ifx_ids_after,
// a download, the display -- gets it here: the id render of the current map at the current pose, exactly what the frame would have drawn (the map and the pose
// do not change between the end of a frame and the start of the next).
int ifx_comm_ready(ifx* h);
int ifx_comm_exchange(ifx* h, int phase);
// sharded map: the id render of every shard at the current pose, its keys MIN-reduced over the ranks (ifx_owner_exchange(200)), then the image every rank reads
int ifx_owner_ids_begin_impl(ifx* h)
{
    if (h->ids_full_valid || !h->ids_sparse_frame) return 0;
    raster_pass(h, nullptr, 0, 0, LIST_IDS, nullptr, false, 1);   // local: all slots of the shard, creation numbers in the keys
    h->own_ids_pending = 1;
    return 1;
}
int ifx_owner_ids_resume_impl(ifx* h)
{
    if (!h->own_ids_pending) { h->err = "ifx_owner_ids_resume: no id render is waiting"; return IFX_E_STATE; }
    LAUNCH(h, "own_ids_resolve", dim3(cdiv(h->P, 256)), dim3(256), k_own_ids_resolve, h->key_ids, h->P, h->ids_after, (const int*)h->gfl_splat);
    h->own_ids_pending = 0;
    h->ids_full_valid = 1;
    return IFX_OK;
}
int ifx_ids_ensure(ifx* h)
{
    if (h->ids_full_valid || !h->ids_sparse_frame) return IFX_OK;
    if (h->own) {   // (option own_lazy_ids) every rank comes by here together: with the library's communicator the exchange is enqueued in place
        if (!ifx_comm_ready(h)) { h->err = "the id image of this sharded map holds the sampled lattice only (option own_lazy_ids): ifx_owner_ids_begin, the exchange of ifx_owner_exchange(200), ifx_owner_ids_resume first"; return IFX_E_STATE; }
        int r = ifx_owner_ids_begin_impl(h);
        if (r < 0) return r;
        if (r == 1) { if ((r = ifx_comm_exchange(h, 200))) return r; if ((r = ifx_owner_ids_resume_impl(h))) return r; }
        return IFX_OK;
    }
    if (h->ids_view_ok && !h->own) {   // nothing touched the store, the pose or the cached view list since the frame drew its lattice from it: the rest of the image from the same list
        Cam c = make_cam(h);
        c.srank = 0; c.sn = 1;
        LAUNCH(h, "raster_view_ids", dim3(h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS), dim3(MAP_THREADS), (k_raster_view<false, false>), h->d_state, (const float4*)h->pc, (const float4*)h->nr,
               (const float2*)h->tm, c, h->tick, h->tick, LIST_IDS, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, h->opt_raster_earlyz, 1, CleanArgs{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (const DevState*)h->d_state);
        LAUNCH(h, "ids_resolve", dim3(cdiv(h->P, 256)), dim3(256), k_ids_resolve, h->key_ids, h->P, h->ids_after);
    } else
        ids_pass(h, nullptr, 0, h->ids_after);   // (all slots, per-pass cull: unstable surfels -- the only ones the view list's age rule concerns -- are never drawn here)
    h->ids_full_valid = 1;
    return IFX_OK;
}

// ------------------------------------------------------------------ association + fusion (a11, a12)
// data.vert:94-241 for every pixel; the measurement is kept per pixel for the update / append passes
// akey != nullptr (spatially sharded map): the rank tests only the candidates it OWNS (it holds the frame and their attributes) and leaves, per measurement pixel,
// (distance bits << 32 | window position) of its best one; the element-wise minimum across the ranks is the winner of the replicated scan -- strict-less on the
// distance, the earlier window position on a tie -- and k_assoc_decode turns it back into the surfel: 2 B per pixel travel instead of the 32-byte attribute images.
__global__ void k_associate(const DevState* __restrict__ st, const float* __restrict__ pose_ex, float weighting_ex, const float* __restrict__ dm, const float* __restrict__ dmf,
                            const uint8_t* __restrict__ rgb, const uint32_t* __restrict__ index_id, const float4* __restrict__ index_vc,
                            const float4* __restrict__ index_nr, Cam c, int time, uint32_t* __restrict__ assoc, float4* __restrict__ mpc, float4* __restrict__ mnr,
                            float* __restrict__ mcol, uint32_t* __restrict__ upd_owner, const uint8_t* __restrict__ vis, unsigned long long* __restrict__ akey = nullptr)
{
    const int fl = FIRST_LIVE(c);
    // Only the pixels with i % 2 == j % 2 == time % 2 create measurements (data.vert:98): one thread per 2x2 block, so that
    // every lane of a wave works; the thread also marks the three silent pixels of its block.
    const int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y * blockDim.y + threadIdx.y;
    const int par = time % 2, i = 2 * bx + par, j = 2 * by + par;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int ii = 2 * bx + a, jj = 2 * by + b;
            if (ii < c.w && jj < c.h && !(ii == i && jj == j)) assoc[jj * c.w + ii] = ASSOC_NONE;
        }
    if (akey && bx < (c.w + 1) / 2 && by < (c.h + 1) / 2) akey[by * ((c.w + 1) / 2) + bx] = IFX_KEY_EMPTY;
    if (i >= c.w || j >= c.h) return;
    int k = j * c.w + i;
    uint32_t res = ASSOC_NONE;
    {
        const float* pose = pose_ex ? pose_ex : st->pose;
        float weighting = pose_ex ? weighting_ex : st->weighting;
        float ifx_ = 1.0f / c.fx, ify_ = 1.0f / c.fy;
        float x = (float)i + 0.5f, y = (float)j + 0.5f;
        // Every load of the pixel is issued up front -- the raw and filtered depth crosses, the colour, the index-map window -- and the reference's
        // tests then run on registers in the reference's order.  (Written test by test, each load sat behind the previous test's branch: up to
        // ~30 dependent round trips for a launch of only 300 blocks, 24 us.)
        // The window: half-texel steps from the centre of texel i - 1 reach {i-1, i, i+1} at most: 3 x 3 distinct texels, visited 16 to 25 times.  A repeated
        // visit never changes the running best (dist < bestDist is strict), so the nine first visits in the original (a, b) order decide.
        const int xs[3] = {clampi((int)floorf(x - 1.0f), 0, c.w - 1), clampi((int)floorf(x - 0.5f), 0, c.w - 1), clampi((int)floorf(x + 0.5f), 0, c.w - 1)};
        const int ys[3] = {clampi((int)floorf(y - 1.0f), 0, c.h - 1), clampi((int)floorf(y - 0.5f), 0, c.h - 1), clampi((int)floorf(y + 0.5f), 0, c.h - 1)};
        // WHICH of the three columns / rows the shader's window loop visits is decided by its f32 arithmetic (window_taps, ifx_dev.h): the taps run from the centre of texel
        // i - 1 in half-texel steps, so i - 1 and i are always among them, while i + 1 is reached only when the tap on the edge between i and i + 1 falls to the right of it or
        // the loop makes its fifth trip -- a property of the column (row) index alone, tabulated when the handle is created.  Visiting order and first-visit semantics are unchanged.
        const unsigned int vis_x = vis[i], vis_y = vis[c.w + j];   // (tabulated at create from window_taps: ifx_api.hip)
        uint32_t cur[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) cur[a * 3 + b] = index_id[ys[b] * c.w + xs[a]];
        const float d_c = tex_f(dm, c.w, c.h, i, j), d_l = tex_f(dm, c.w, c.h, i - 1, j), d_u = tex_f(dm, c.w, c.h, i, j - 1), d_r = tex_f(dm, c.w, c.h, i + 1, j),
                    d_d = tex_f(dm, c.w, c.h, i, j + 1);
        const float f_c = tex_f(dmf, c.w, c.h, i, j), f_l = tex_f(dmf, c.w, c.h, i - 1, j), f_u = tex_f(dmf, c.w, c.h, i, j - 1), f_r = tex_f(dmf, c.w, c.h, i + 1, j),
                    f_d = tex_f(dmf, c.w, c.h, i, j + 1);
        const uint8_t* cc = &rgb[k * 3];
        const uint8_t c0 = cc[0], c1 = cc[1], c2 = cc[2];
        float4 vcs[9], nrs[9];
#pragma unroll
        for (int q = 0; q < 9; q++) {   // the records of all nine texels, occupied or not (an empty texel's record is never looked at): no branch around a load
            const int a = q / 3, b = q - 3 * a;
            const int kk = ys[b] * c.w + xs[a];
            vcs[q] = index_vc[kk];
            nrs[q] = index_nr[kk];
        }
        asm volatile("" ::"v"(d_c), "v"(d_l), "v"(d_u), "v"(d_r), "v"(d_d), "v"(f_c), "v"(f_l), "v"(f_u), "v"(f_r), "v"(f_d), "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(cur[4]),
                     "v"(cur[5]), "v"(cur[6]), "v"(cur[7]), "v"(cur[8]));
        asm volatile("" ::"v"(vcs[0].x), "v"(vcs[1].x), "v"(vcs[2].x), "v"(vcs[3].x), "v"(vcs[4].x), "v"(vcs[5].x), "v"(vcs[6].x), "v"(vcs[7].x), "v"(vcs[8].x), "v"(nrs[0].x),
                     "v"(nrs[1].x), "v"(nrs[2].x), "v"(nrs[3].x), "v"(nrs[4].x), "v"(nrs[5].x), "v"(nrs[6].x), "v"(nrs[7].x), "v"(nrs[8].x));   // (all 18 in flight before the first test)
        v3 vl = v3m((x - c.cx) * d_c * ifx_, (y - c.cy) * d_c * ify_, d_c);
        bool nb = !(d_l == 0 || d_u == 0 || d_r == 0 || d_d == 0);
        if (nb && vl.z > 0 && vl.z <= c.maxDepth) {
            v3 vg = xf_point(pose, vl);
            v3 vf = v3m((x - c.cx) * f_c * ifx_, (y - c.cy) * f_c * ify_, f_c);
            v3 nl;
            {   // get_normal_f on the taps already in registers
                const v3 xf = v3m((x + 1 - c.cx) * f_r * ifx_, (y - c.cy) * f_r * ify_, f_r), xb = v3m((x - 1 - c.cx) * f_l * ifx_, (y - c.cy) * f_l * ify_, f_l);
                const v3 yf = v3m((x - c.cx) * f_d * ifx_, (y + 1 - c.cy) * f_d * ify_, f_d), yb = v3m((x - c.cx) * f_u * ifx_, (y - 1 - c.cy) * f_u * ify_, f_u);
                const v3 del_x = ((xb + vf) * 0.5f) - ((xf + vf) * 0.5f), del_y = ((yb + vf) * 0.5f) - ((yf + vf) * 0.5f);
                nl = normalized(cross(del_x, del_y));
            }
            v3 ng = xf_dir(pose, nl);
            mpc[k] = make_float4(vg.x, vg.y, vg.z, confidence_fn(x, y, c.cx, c.cy, weighting));
            mcol[k] = encode_color(c0 / 255.0f, c1 / 255.0f, c2 / 255.0f);
            mnr[k] = make_float4(ng.x, ng.y, ng.z, get_radius(vf.z, nl.z, ifx_, ify_));
            float xl = (x - c.cx) * ifx_, yl = (y - c.cy) * ify_;
            float lambda = sqrtf(xl * xl + yl * yl + 1);
            v3 ray = v3m(xl, yl, 1);
            float rayLen = norm(ray);
            float bestDist = 1000;
            uint32_t best = 0;
            int counter = 0, best_q = 0;
#pragma unroll
            for (int q = 0; q < 9; q++) {
                // (sharded map, akey: only the candidates this rank OWNS -- their attribute records are the ones k_index_resolve filled in; a foreign winner's record is
                // all zeros here, and a zero depth could never pass the 5 cm test against a measurement of at least 0.3 m anyway)
                if (!((vis_x >> (q / 3)) & (vis_y >> (q % 3)) & 1u)) continue;   // a texel the window loop of this column / row does not reach
                if (cur[q] > 0u && (!akey || (__float_as_uint(vcs[q].x) | __float_as_uint(vcs[q].y) | __float_as_uint(vcs[q].z) | __float_as_uint(vcs[q].w)) != 0u)) {
                    const float4 vc = vcs[q];
                    if (fabsf((vc.z * lambda) - (vl.z * lambda)) < 0.05f) {
                        float dist = norm(cross(ray, v3m(vc.x, vc.y, vc.z))) / rayLen;
                        const float4 nrm = nrs[q];
                        v3 nn = v3m(nrm.x, nrm.y, nrm.z);
                        float cang = dot(nn, nl) / (norm(nn) * norm(nl));
                        if (dist < bestDist && (fabsf(nrm.z) < 0.75f || cang > 0.87758256189f)) { counter++; bestDist = dist; best = cur[q]; best_q = q; }
                    }
                }
            }
            if (akey) {   // the verdict waits for the other ranks' candidates (k_assoc_decode): "new" unless somebody has one
                if (counter > 0) akey[by * ((c.w + 1) / 2) + bx] = ((unsigned long long)__float_as_uint(bestDist) << 32) | (unsigned int)best_q;   // (bestDist >= 0: its bits order like its value)
                res = ASSOC_NEW;
            } else if (counter > 0) {
                res = best;
                const int lb = local_slot(c, st->count, best, fl);
                if (lb >= 0) atomicMin(&upd_owner[lb], (uint32_t)(i * c.h + j));   // first pixel in column-major order owns the surfel (sharded map: only the surfel's rank keeps the score)
            } else res = ASSOC_NEW;
        }
    }
    assoc[k] = res;
}

// the exchanged minimum back into the association: window position -> texel -> surfel; the surfel's rank enters the first-pixel-wins arbitration
__global__ void k_assoc_decode(const DevState* __restrict__ st, const unsigned long long* __restrict__ akey, const uint32_t* __restrict__ index_id, Cam c, int time,
                               uint32_t* __restrict__ assoc, uint32_t* __restrict__ upd_owner, const int32_t* __restrict__ own_slot = nullptr, int32_t* __restrict__ assoc_slot = nullptr)
{
    const int fl = FIRST_LIVE(c);
    const int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y * blockDim.y + threadIdx.y;
    const int par = time % 2, i = 2 * bx + par, j = 2 * by + par;
    if (i >= c.w || j >= c.h) return;
    const unsigned long long key = akey[by * ((c.w + 1) / 2) + bx];
    if (key == IFX_KEY_EMPTY) return;
    const int q = (int)(key & 0xFu), a = q / 3, b = q - 3 * a;
    const float x = (float)i + 0.5f, y = (float)j + 0.5f;
    const int xs[3] = {clampi((int)floorf(x - 1.0f), 0, c.w - 1), clampi((int)floorf(x - 0.5f), 0, c.w - 1), clampi((int)floorf(x + 0.5f), 0, c.w - 1)};
    const int ys[3] = {clampi((int)floorf(y - 1.0f), 0, c.h - 1), clampi((int)floorf(y - 0.5f), 0, c.h - 1), clampi((int)floorf(y + 0.5f), 0, c.h - 1)};
    const uint32_t best = index_id[ys[b] * c.w + xs[a]];
    assoc[j * c.w + i] = best;
    const int lb = own_slot ? own_slot_of(c, own_slot[ys[b] * c.w + xs[a]], best) : local_slot(c, st->count, best, fl);
    if (assoc_slot) assoc_slot[j * c.w + i] = lb;   // (k_fuse_update's slot of the associated surfel: -1 = another rank applies the update)
    if (lb >= 0) atomicMin(&upd_owner[lb], (uint32_t)(i * c.h + j));
}

// update.vert:55-141 in place, by the owning pixel only
__global__ void k_fuse_update(DevState* __restrict__ st, const uint32_t* __restrict__ assoc, const float4* __restrict__ mpc, const float4* __restrict__ mnr,
                              const float* __restrict__ mcol, Cam c, int time, uint32_t* __restrict__ upd_owner, float4* __restrict__ pc, float4* __restrict__ nr,
                              float2* __restrict__ col, float2* __restrict__ tm, const int32_t* __restrict__ assoc_slot = nullptr, Hot* __restrict__ hot = nullptr)
{
    const int fl = FIRST_LIVE(c);
    const int par = time % 2, i = 2 * (blockIdx.x * blockDim.x + threadIdx.x) + par, j = 2 * (blockIdx.y * blockDim.y + threadIdx.y) + par;   // the pixels that can hold an association
    if (i >= c.w || j >= c.h) return;
    int k = j * c.w + i;
    const uint32_t gid = assoc[k];
    if (gid >= ASSOC_NEW) return;
    const int li = assoc_slot ? assoc_slot[k] : local_slot(c, st->count, gid, fl);   // sharded map: -1 = another rank's surfel (that rank applies the update)
    if (li < 0 || li >= st->count) return;
    const uint32_t id = (uint32_t)li;
    if (upd_owner[id] != (uint32_t)(i * c.h + j)) return;
    upd_owner[id] = 0xFFFFFFFFu;
    float4 p = hot ? hot[id].pc : pc[id], n = hot ? hot[id].nr : nr[id], mp = mpc[k], mn = mnr[k];
    const float2 cl0 = col[id], t0_ = hot ? hot[id].tm : tm[id];   // issued with the other loads (they used to follow them: two more dependent round trips).  Fetching all of it
    const float mc = mcol[k];                   // together with the ownership word was tried too: 13.9 -> 15.6 us (the losers' 80 bytes cost more than the round trip)
    float c_k = p.w, a = mp.w;
    if (mn.w < (1.0f + 0.5f) * n.w) {
        p.x = ((c_k * p.x) + (a * mp.x)) / (c_k + a);
        p.y = ((c_k * p.y) + (a * mp.y)) / (c_k + a);
        p.z = ((c_k * p.z) + (a * mp.z)) / (c_k + a);
        p.w = c_k + a;
        float2 cl = cl0;
        float oc[3], nc[3];
        decode_color(cl.x, oc);
        decode_color(mc, nc);
        cl.x = encode_color(((c_k * oc[0]) + (a * nc[0])) / (c_k + a), ((c_k * oc[1]) + (a * nc[1])) / (c_k + a), ((c_k * oc[2]) + (a * nc[2])) / (c_k + a));
        col[id] = cl;
        float t0 = ((c_k * n.x) + (a * mn.x)) / (c_k + a), t1 = ((c_k * n.y) + (a * mn.y)) / (c_k + a), t2 = ((c_k * n.z) + (a * mn.z)) / (c_k + a);
        float t3 = ((c_k * n.w) + (a * mn.w)) / (c_k + a);
        v3 nn = normalized(v3m(t0, t1, t2));
        nr[id] = make_float4(nn.x, nn.y, nn.z, t3);
        if (hot) hot[id].nr = make_float4(nn.x, nn.y, nn.z, t3);
        if (t3 > 0.f && t3 < 1e30f && __float_as_uint(t3) > st->r_max_bits) atomicMax(&st->r_max_bits, __float_as_uint(t3));
        pc[id] = p;
    } else {
        p.w = c_k + a;
        pc[id] = p;
    }
    float2 t = t0_;
    t.y = (float)time;
    tm[id] = t;
    if (hot) { hot[id].pc = p; hot[id].tm = t; }
}

// ------------------------------------------------------------------ clean (a13)
// copy_unstable.vert:103-174
__device__ inline int clean_test(const float* T, const Cam& c, int time, float4 p4, float4 n4, float initT, float& lastT, const float4* __restrict__ tap)
{
    int test = 1;
    v3 lp = xf_point(T, v3m(p4.x, p4.y, p4.z));
    float x = ((c.fx * lp.x) / lp.z) + c.cx, y = ((c.fy * lp.y) / lp.z) + c.cy;
    int count = 0, zCount = 0;
    float wv = lastT;
    if ((float)time - wv < (float)c.timeDelta && lp.z > 0 && x > 0 && y > 0 && x < (float)c.w && y < (float)c.h) {
        v3 ln = normalized(xf_dir(T, v3m(n4.x, n4.y, n4.z)));
        const bool flat = fabsf(ln.z) > 0.85f;
        // The window: the shader's float loop (window_taps: four or FIVE taps per axis, decided by its f32 arithmetic) around the projected position.
        int tx[IFX_MAX_TAPS], ty[IFX_MAX_TAPS];
        window_taps(x / (float)c.w, (float)c.w, c.w, tx);
        window_taps(y / (float)c.h, (float)c.h, c.h, ty);
        // The 16 to 25 taps visit (almost always) at most 3 x 3 DISTINCT texels: consecutive columns tx[0], tx[0] + 1, tx[0] + 2 (non-decreasing, clamped), a column being visited
        // as often as it appears among the taps -- so nine gathers, each counted with its multiplicity mx * my, give the window's two counts exactly (counter evidence,
        // profiles/r04_a_pmc_bound.json: the pass was bound by the address processing of its divergent gathers, not by its arithmetic).  A fourth column / row (a tap span that
        // straddles two texel edges by rounding) is handled by the slow path below.
        int mx[4], my[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            mx[q] = 0; my[q] = 0;
#pragma unroll
            for (int a = 0; a < IFX_MAX_TAPS; a++) { mx[q] += (tx[a] == tx[0] + q) ? 1 : 0; my[q] += (ty[a] == ty[0] + q) ? 1 : 0; }
        }
        if (mx[3] | my[3]) {   // rare: sixteen distinct texels, one at a time
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) {
                    const int wgt = mx[a] * my[b];
                    if (!wgt) continue;
                    const float4 e = tap[min(ty[0] + b, c.h - 1) * c.w + min(tx[0] + a, c.w - 1)];
                    if (e.z == 0.f) continue;
                    const float ez = fabsf(e.z);
                    const bool stable = e.w > 0.f, now = e.z < 0.f;
                    const float dx = e.x - lp.x, dy = e.y - lp.y;
                    if (fabsf(e.w) < initT && stable && ez > lp.z && ez - lp.z < 0.01f && sqrtf(dx * dx + dy * dy) < n4.w * 1.4f) count += wgt;
                    if (now && stable && ez > lp.z && ez - lp.z > 0.01f && flat) zCount += wgt;
                }
            mx[0] = mx[1] = mx[2] = 0;   // (nothing left for the fast path)
        }
        float4 t[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) t[a * 3 + b] = tap[min(ty[0] + b, c.h - 1) * c.w + min(tx[0] + a, c.w - 1)];   // all nine gathers in flight
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int wgt = mx[k / 3] * my[k % 3];
            const float4 e = t[k];
            if (wgt == 0 || e.z == 0.f) continue;           // a column / row the window does not visit; an empty texel (or surfel id 0)
            const float ez = fabsf(e.z);
            const bool stable = e.w > 0.f, now = e.z < 0.f;
            const float dx = e.x - lp.x, dy = e.y - lp.y;
            if (fabsf(e.w) < initT && stable && ez > lp.z && ez - lp.z < 0.01f && sqrtf(dx * dx + dy * dy) < n4.w * 1.4f) count += wgt;
            if (now && stable && ez > lp.z && ez - lp.z > 0.01f && flat) zCount += wgt;
        }
    }
    if (count > 8 || zCount > 4) test = 0;
    if (wv == -2) wv = (float)time;
    if (wv == -1 || (((float)time - wv) > 20 && p4.w < c.conf)) test = 0;
    if (wv > 0 && (float)time - wv > (float)c.timeDelta) test = 1;
    lastT = wv;
    return test;
}



// The append of the new surfels in two launches (was: flags, 3-kernel scan, scatter, count).  Launch 1 evaluates
// the stability test for the pixels that created a surfel, in column-major order (the order of the reference's
// transform-feedback stream), 4 consecutive order indices per thread, and leaves the flags and one count per
// 1024-pixel block; it also re-arms the clean work lists.  Launch 2: every block adds the counts of the blocks
// before it (<= 300 integers), scans its own flags and scatters; the block that finishes last publishes the
// new surfel count.
// flags: bit 0 = the new surfel survives its first clean test, bit 1 = ... and this rank owns it (spatially sharded map; always set otherwise).
// block_counts: [block][2] = survivors, owned survivors.
__device__ __forceinline__ void new_flags_body(DevState* st, const float* __restrict__ pose_inv_ex, const Cam& c, int time, const uint32_t* __restrict__ assoc, const float4* __restrict__ mpc,
                                               const float4* __restrict__ mnr, const float4* __restrict__ tap, int* __restrict__ flags, int* __restrict__ block_counts, int bid)
{
    __shared__ int lds[4][2];
    const int P = c.w * c.h, ord0 = bid * NEW_PER_BLOCK + threadIdx.x * 4;
    const float* Ti = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    float T[12];   // (fetched once, here: as a pointer handed to clean_test it was re-read -- as vector loads of one address -- inside every one of the four tests)
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Ti[k];
    int keep[4], cnt = 0, own = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int ord = ord0 + u;
        keep[u] = 0;
        if (ord < P) {
            const int i = ord / c.h, j = ord - i * c.h, k = j * c.w + i;
            if (assoc[k] == ASSOC_NEW) {
                float lastT = -2.f;
                const float4 m4 = mpc[k];
                if (clean_test(T, c, time, m4, mnr[k], (float)time, lastT, tap)) keep[u] = 1 | ((c.own_n <= 0 || ifx_owner_of_point(m4.x, m4.y, m4.z, c.own_n) == c.own_rank) ? 2 : 0);
            }
        }
        cnt += keep[u] & 1;
        own += keep[u] >> 1;
    }
    if (ord0 + 3 < P) *reinterpret_cast<int4*>(flags + ord0) = make_int4(keep[0], keep[1], keep[2], keep[3]);
    else
        for (int u = 0; u < 4; u++) if (ord0 + u < P) flags[ord0 + u] = keep[u];
    cnt = wave_sum_i(cnt);
    own = wave_sum_i(own);
    if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6][0] = cnt; lds[threadIdx.x >> 6][1] = own; }
    __syncthreads();
    if (threadIdx.x < 2) block_counts[bid * 2 + threadIdx.x] = lds[0][threadIdx.x] + lds[1][threadIdx.x] + lds[2][threadIdx.x] + lds[3][threadIdx.x];
    if (bid == 0 && threadIdx.x < 2 * LIST_SEGS) c.lctr[(LIST_SEGS + threadIdx.x) * LIST_CTR_STRIDE] = 0;   // lists 1, 2: k_clean_list, the launch before this one, was their last reader
    if (bid == 0 && threadIdx.x == 64) { st->app_count0 = st->count; st->app_seq0 = st->next_seq; st->app_vln0 = st->vl_n[0]; }   // what k_append_scan, the next launch, starts from (nothing in between changes them)
}
__global__ void __launch_bounds__(256) k_new_flags_count(DevState* st, const float* __restrict__ pose_inv_ex, Cam c, int time, const uint32_t* __restrict__ assoc,
                                                         const float4* __restrict__ mpc, const float4* __restrict__ mnr, const float4* __restrict__ tap, int* __restrict__ flags,
                                                         int* __restrict__ block_counts)
{
    new_flags_body(st, pose_inv_ex, c, time, assoc, mpc, mnr, tap, flags, block_counts, (int)blockIdx.x);
}

// Two ranks per surviving new surfel: its place in the frame's append order (-> creation number, the same on every rank) and its place among
// the ones this rank stores (-> slot).  Without sharding the two coincide.
__device__ __forceinline__ void append_scan_body(DevState* st, const Cam& c, int time, int tick, const int* __restrict__ flags, const int* __restrict__ block_counts, int nblocks,
                                                 const float4* __restrict__ mpc, const float4* __restrict__ mnr, const float* __restrict__ mcol, int cap, float4* __restrict__ pc,
                                                 float4* __restrict__ nr, float2* __restrict__ col, float2* __restrict__ tm, float4* __restrict__ ic, float4* __restrict__ votes,
                                                 const uint8_t* __restrict__ inst_gt, unsigned int* __restrict__ list_v, int32_t* __restrict__ labels, uint32_t* __restrict__ seq,
                                                 const int tid, const int bid, Hot* __restrict__ hot)
{
    // No grid-wide hand-off inside this launch (round 5; it had two: one returning atomic per block for the view-list positions and a last-block ticket + recount to publish the
    // new count -- 3 of its ~8 dependent round trips): the slot count, the creation number and the view list's length the frame started its append with are a SNAPSHOT left by the
    // flags pass (k_new_flags_count's block 0, the launch before this one), so nothing this launch publishes can be read by a block that starts late; every position follows from
    // the counts of the blocks before; and the block with the highest index, which knows the frame's total from its own prefix, publishes.
    __shared__ int s_wave[4][2], s_base[2];
    const int P = c.w * c.h, lane = tid & 63, wid = tid >> 6;
    const int count0 = st->app_count0;
    const unsigned int seq0 = st->app_seq0, vln0 = st->app_vln0;
    const unsigned int rmax0 = st->r_max_bits;   // (here, through the scalar cache: re-read per new surfel behind the stores below it was a vector load of one address, ~2 ns per wave and surfel at one L2 channel)
    // counts of the blocks before this one
    int beforeG = 0, beforeO = 0;
    for (int b = tid; b < bid; b += 256) { beforeG += block_counts[2 * b]; beforeO += block_counts[2 * b + 1]; }
    beforeG = wave_sum_i(beforeG);
    beforeO = wave_sum_i(beforeO);
    if (lane == 0) { s_wave[wid][0] = beforeG; s_wave[wid][1] = beforeO; }
    __syncthreads();
    if (tid < 2) s_base[tid] = s_wave[0][tid] + s_wave[1][tid] + s_wave[2][tid] + s_wave[3][tid];
    __syncthreads();
    // exclusive scans of this block's flags: 4 per thread, wave scan by shuffles, wave totals through LDS
    const int ord0 = bid * NEW_PER_BLOCK + tid * 4;
    int f[4] = {0, 0, 0, 0};
    if (ord0 + 3 < P) { int4 v = *reinterpret_cast<const int4*>(flags + ord0); f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    else
        for (int u = 0; u < 4; u++) if (ord0 + u < P) f[u] = flags[ord0 + u];
    const int mineG = (f[0] & 1) + (f[1] & 1) + (f[2] & 1) + (f[3] & 1), mineO = (f[0] >> 1) + (f[1] >> 1) + (f[2] >> 1) + (f[3] >> 1);
    int inclG = mineG, inclO = mineO;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int tg = __shfl_up(inclG, o), to = __shfl_up(inclO, o);
        if (lane >= o) { inclG += tg; inclO += to; }
    }
    __syncthreads();
    if (lane == 63) { s_wave[wid][0] = inclG; s_wave[wid][1] = inclO; }
    __syncthreads();
    int wave_offG = 0, wave_offO = 0;
    for (int q = 0; q < wid; q++) { wave_offG += s_wave[q][0]; wave_offO += s_wave[q][1]; }
    int rankG = s_base[0] + wave_offG + inclG - mineG, rankO = s_base[1] + wave_offO + inclO - mineO;
    // the new surfels join the cached view list (they were created from this frame's pixels, so they are in view): one reservation per block
    // Only surfels that are actually STORED get a list position: slots ascend with the owned rank, so the ones of this block that fit below
    // `cap` are its first `fit` -- a full map must not leave reserved-but-unwritten entries for the list walkers to dereference.
    const int blk_total = s_wave[0][1] + s_wave[1][1] + s_wave[2][1] + s_wave[3][1];
    const bool to_view = list_v && st->vl_valid;
    // (stored surfels are the first min(total, cap - count0) in append order: block b's are at view-list positions vln0 + owned-before-b ...)
    const unsigned int vbase = vln0 + (unsigned int)s_base[1];
    unsigned int vpos = vbase + (unsigned int)(wave_offO + inclO - mineO);
    bool over = false;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (!(f[u] & 1)) continue;
        const unsigned int sq = seq0 + (unsigned int)rankG++;
        if (!(f[u] & 2)) continue;   // another rank stores it
        const int ord = ord0 + u, i = ord / c.h, j = ord - i * c.h, k = j * c.w + i;
        const int n = count0 + rankO++;
        const unsigned int vp = vpos++;
        if (n >= cap) { over = true; continue; }
        if (to_view && vp < c.seg_cap * LIST_SEGS) list_v[vp] = (unsigned int)n;
        const float4 m4 = mpc[k], n4 = mnr[k];
        pc[n] = m4;
        nr[n] = n4;
        { const float rad = n4.w; if (rad > 0.f && rad < 1e30f && __float_as_uint(rad) > rmax0) atomicMax(&st->r_max_bits, __float_as_uint(rad)); }
        col[n] = make_float2(mcol[k], 0.f);
        tm[n] = make_float2((float)time, (float)time);
        if (hot) { Hot r; r.pc = m4; r.nr = n4; r.tm = make_float2((float)time, (float)time); r.pad0 = make_float2(0.f, 0.f); r.pad1 = make_float4(0.f, 0.f, 0.f, 0.f); hot[n] = r; }
        ic[n] = make_float4((float)i + 0.5f, (float)j + 0.5f, (float)tick, inst_gt ? (float)inst_gt[j * c.w + i] : -2.f);   // data.vert:215-228: ground-truth instance id of the creating pixel
        for (int q = 0; q < 12; q++) VOTE4(votes, n, q) = make_float4(0.f, 0.f, 0.f, 0.f);
        labels[n] = -1;   // no label until the next label scan (the slot may hold one from before a compaction)
        seq[n] = sq;
    }
    if (over) { st->overflow = 1; st->vl_valid = 0; }
    if (bid != nblocks - 1) return;
    if (tid == 0) {   // the frame's totals = what lies before this block + this block
        const int tg = s_base[0] + s_wave[0][0] + s_wave[1][0] + s_wave[2][0] + s_wave[3][0], to = s_base[1] + blk_total;
        int nc = count0 + to;
        bool ovf = false;
        if (nc > cap) { nc = cap; ovf = true; }
        if (seq0 + (unsigned int)tg < seq0 || seq0 + (unsigned int)tg > 0xFFF00000u) ovf = true;   // creation numbers are never renumbered: 2^32 of them is the life of a sharded map (reported as a full store)
        st->n_new = nc - count0;
        st->count = nc;
        {   // the clean pass of this frame is behind us: if it removed the reference's "surfel 0", the next live slot takes its place (the appended ones are alive by construction).
            // A slot that no view list holds was not visited by this clean pass: the age rule -- all that can apply to it -- is evaluated here (the next scan tombstones it);
            // for a slot the pass did visit and keep the rule says "kept" again.
            int f = st->first_live;
            while (f < count0) {
                const float wv = tm[f].y;
                if (wv > DEAD_TIME && !age_rule_gone(wv, pc[f].w, time, c)) break;
                f++;
            }
            st->first_live = f;
        }
        st->next_seq = seq0 + (unsigned int)tg;
        if (ovf) st->overflow = 1;
        if (to_view) {   // the view list grew by the surfels that were stored; a list that cannot hold them is void (rebuilt by the next frame's scan)
            const unsigned int vn = vln0 + (unsigned int)(nc - count0);
            if (vn > c.seg_cap * LIST_SEGS) st->vl_valid = 0; else st->vl_n[0] = vn;
        }
    }
}
__global__ void __launch_bounds__(256) k_append_scan(DevState* st, Cam c, int time, int tick, const int* __restrict__ flags, const int* __restrict__ block_counts, int nblocks,
                                                     const float4* __restrict__ mpc, const float4* __restrict__ mnr, const float* __restrict__ mcol, int cap, float4* __restrict__ pc,
                                                     float4* __restrict__ nr, float2* __restrict__ col, float2* __restrict__ tm, float4* __restrict__ ic, float4* __restrict__ votes,
                                                     const uint8_t* __restrict__ inst_gt, unsigned int* __restrict__ list_v, int32_t* __restrict__ labels, uint32_t* __restrict__ seq,
                                                     Hot* __restrict__ hot = nullptr)
{
    append_scan_body(st, c, time, tick, flags, block_counts, nblocks, mpc, mnr, mcol, cap, pc, nr, col, tm, ic, votes, inst_gt, list_v, labels, seq, (int)threadIdx.x, (int)blockIdx.x, hot);
}

// ------------------------------------------------------------------ tombstone compaction
__global__ void k_alive_flags(const DevState* __restrict__ st, const float2* __restrict__ tm, int* __restrict__ flags, int cap)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    flags[i] = (i < st->count && tm[i].y > DEAD_TIME) ? 1 : 0;
}
__global__ void k_compact_scatter(const int* __restrict__ flags, const int* __restrict__ rank, int cap, const float4* __restrict__ pc, const float4* __restrict__ nr,
                                  const float2* __restrict__ col, const float2* __restrict__ tm, const float4* __restrict__ ic, const float4* __restrict__ votes,
                                  const int32_t* __restrict__ labels, float4* __restrict__ pc2, float4* __restrict__ nr2, float2* __restrict__ col2,
                                  float2* __restrict__ tm2, float4* __restrict__ ic2, float4* __restrict__ votes2, int32_t* __restrict__ labels2,
                                  const uint32_t* __restrict__ seq, uint32_t* __restrict__ seq2)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || !flags[i]) return;
    int d = rank[i];
    pc2[d] = pc[i]; nr2[d] = nr[i]; col2[d] = col[i]; tm2[d] = tm[i]; ic2[d] = ic[i];
    if (labels2) labels2[d] = labels[i];
    seq2[d] = seq[i];
    for (int q = 0; q < 12; q++) VOTE4(votes2, d, q) = VOTE4(votes, i, q);
}
__global__ void k_compact_count(DevState* st, const int* total)
{
    if (threadIdx.x == 0) { st->count = *total; st->n_dead = 0; st->vl_valid = 0; st->first_live = 0; }   // slots renumbered: the view list is void
}

static void ids_pass(ifx* h, const float* d_pose_inv, int mode, int32_t* out);
int ifx_compact_enqueue(ifx* h, int refresh_ids)
{
    h->ids_view_ok = 0;
    h->hot_valid = 0;
    ifx_vlist_reap(h);   // slots the view list left out may have outlived the age rule: tombstone them before the live ranks are taken
    // alive flags over the host-known upper bound of slots; scan; scatter into the second buffer set; swap
    int n = h->cap;
    LAUNCH(h, "alive_flags", dim3(cdiv(n, 256)), dim3(256), k_alive_flags, h->d_state, (const float2*)h->tm, h->scan_flags, n);
    ifx_scan_exclusive(h, h->scan_flags, n, h->scan_out, &h->d_state->seg_counts[1]);
    LAUNCH(h, "compact_scatter", dim3(cdiv(n, 256)), dim3(256), k_compact_scatter, h->scan_flags, h->scan_out, n, (const float4*)h->pc, (const float4*)h->nr,
           (const float2*)h->col, (const float2*)h->tm, (const float4*)h->ic, (const float4*)h->votes, (const int32_t*)h->labels, (float4*)h->pc2, (float4*)h->nr2,
           (float2*)h->col2, (float2*)h->tm2, (float4*)h->ic2, (float4*)h->votes2, h->labels2, (const uint32_t*)h->seq, h->seq2);
    LAUNCH(h, "compact_count", dim3(1), dim3(64), k_compact_count, h->d_state, &h->d_state->seg_counts[1]);
    std::swap(h->pc, h->pc2); std::swap(h->nr, h->nr2); std::swap(h->col, h->col2); std::swap(h->tm, h->tm2); std::swap(h->ic, h->ic2); std::swap(h->votes, h->votes2); std::swap(h->labels, h->labels2); std::swap(h->seq, h->seq2);
    if (refresh_ids && !h->own) { ids_pass(h, nullptr, 0, h->ids_after); h->ids_full_valid = 1; }   // slot numbers changed: re-render the id image (a sharded map's id image holds creation numbers: nothing changed)
    return IFX_OK;
}

// ------------------------------------------------------------------ per-frame orchestration of the map stages
// part 0: the whole pass; spatially sharded map: 1 = the association among the candidates this rank owns (leaves h->assoc_key for the exchange),
// 2 = the exchanged verdicts decoded + the update of the owned surfels
static void fuse_pass(ifx* h, const float* d_pose, float weighting, int time, int part = 0, const int32_t* own_slot = nullptr, Hot* hot = nullptr)
{
    if (!hot) h->hot_valid = 0;   // (the update below writes the arrays only)
    Cam c = make_cam(h);
    dim3 b(32, 8), g(cdiv(cdiv(h->w, 2), 32), cdiv(cdiv(h->h, 2), 8));   // one thread per 2x2 pixel block
    if (part != 2)
        LAUNCH(h, "associate", g, b, k_associate, h->d_state, d_pose, weighting, h->dm, h->dmf, h->rgb, h->index_id, (const float4*)h->index_vc, (const float4*)h->index_nr, c, time,
               h->assoc_target, (float4*)h->meas_pc, (float4*)h->meas_nr, h->meas_col, h->upd_owner, (const uint8_t*)h->assoc_vis, part == 1 ? h->assoc_key : (unsigned long long*)nullptr);
    if (part == 1) return;
    const bool slots = part == 2 && own_slot && h->own_slot_img;
    if (part == 2) LAUNCH(h, "assoc_decode", g, b, k_assoc_decode, h->d_state, (const unsigned long long*)h->assoc_key, h->index_id, c, time, h->assoc_target, h->upd_owner,
                          slots ? own_slot : (const int32_t*)nullptr, slots ? h->own_slot_img + 3 * (size_t)h->P : (int32_t*)nullptr);
    LAUNCH(h, "fuse_update", g, b, k_fuse_update, h->d_state, h->assoc_target, (const float4*)h->meas_pc, (const float4*)h->meas_nr, h->meas_col, c, time, h->upd_owner,
           (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, slots ? (const int32_t*)(h->own_slot_img + 3 * (size_t)h->P) : (const int32_t*)nullptr, hot);
}

// ------------------------------------------------------------------ loop-closure hooks on the map
// Deformation-graph application of copy_unstable.vert:178-374 for every surviving surfel that was not created this frame: binary search of the
// node nearest in time, the 20 nodes around it in the (time-sorted) sequence, the 4 nearest of those in space with weights (1 - d / d5)^2,
// blended rigid motions; then (stable surfels, local loop closure) the time stamp is refreshed when the moved surfel lies in front of the
// re-rendered INACTIVE model depth (IndexMap::synthesizeDepth).  The graph (<= 1023 nodes x 64 B) is staged in LDS.
__global__ __launch_bounds__(256) void k_deform(const DevState* __restrict__ st, const float* __restrict__ pose_inv_ex, const float* __restrict__ graph, int nodes, int is_fern,
                                                 int time, Cam c, const float4* __restrict__ depth_v4, float4* __restrict__ pc, float4* __restrict__ nr, float2* __restrict__ tm)
{
    extern __shared__ float g[];
    for (int k = threadIdx.x; k < nodes * 16; k += blockDim.x) g[k] = graph[k];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<DevState*>(st)->vl_valid = 0;   // positions move: the cached view list is void
    const float* T = pose_inv_ex ? pose_inv_ex : st->pose_inv;
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        float2 t2 = tm[i];
        if (!(t2.y > DEAD_TIME) || t2.x == (float)time) continue;   // tombstone / created this frame (fused with the updated pose already)
        float4 p4 = pc[i], n4 = nr[i];
        const int K = 4, LOOK = 20;
        int nearNodes[LOOK];
        float nearDists[LOOK];
#pragma unroll
        for (int q = 0; q < LOOK; q++) { nearNodes[q] = -1; nearDists[q] = 16777216.0f; }
        const int poseTime = (int)t2.x;
        int foundIndex = 0, imin = 0, imax = nodes - 1, imid = (imin + imax) / 2;
        while (imax >= imin) {
            imid = (imin + imax) / 2;
            const int nodeTime = (int)g[imid * 16 + 15];
            if (nodeTime < poseTime) imin = imid + 1;
            else if (nodeTime > poseTime) imax = imid - 1;
            else break;
        }
        imin = min(imin, nodes - 1);
        const int cmax = imax < 0 ? 0 : imax;   // see oracle/orc_deform.c: an index clamp instead of the shader's out-of-range texel
        const int nodeMin = (int)g[imin * 16 + 15], nodeMid = (int)g[imid * 16 + 15], nodeMax = (int)g[cmax * 16 + 15];
        if (abs(nodeMin - poseTime) <= abs(nodeMid - poseTime) && abs(nodeMin - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imin;
        else if (abs(nodeMid - poseTime) <= abs(nodeMin - poseTime) && abs(nodeMid - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imid;
        else foundIndex = cmax;
        if (foundIndex == nodes) foundIndex = nodes - 1;
        int nearNodeIndex = 0, distanceBack = 0;
        const v3 p = v3m(p4.x, p4.y, p4.z);
        for (int j = foundIndex; j >= 0; j--) {
            const v3 d = p - v3m(g[j * 16], g[j * 16 + 1], g[j * 16 + 2]);
            nearNodes[nearNodeIndex] = j;
            nearDists[nearNodeIndex] = sqrtf(dot(d, d));
            nearNodeIndex++;
            if (++distanceBack == LOOK / 2) break;
        }
        for (int j = foundIndex + 1; j < nodes; j++) {
            const v3 d = p - v3m(g[j * 16], g[j * 16 + 1], g[j * 16 + 2]);
            nearNodes[nearNodeIndex] = j;
            nearDists[nearNodeIndex] = sqrtf(dot(d, d));
            nearNodeIndex++;
            if (++distanceBack == LOOK) break;
        }
        for (int a = 0; a < LOOK - 1; ++a)
            for (int b = a + 1; b < LOOK; ++b)
                if (nearDists[b] < nearDists[a]) {
                    const float t = nearDists[a]; nearDists[a] = nearDists[b]; nearDists[b] = t;
                    const int u = nearNodes[a]; nearNodes[a] = nearNodes[b]; nearNodes[b] = u;
                }
        const float dMax = nearDists[K];
        float wgt[K], weightSum = 0;
        for (int j = 0; j < K; j++) {
            const float* nd = &g[nearNodes[j] * 16];
            const v3 d = p - v3m(nd[0], nd[1], nd[2]);
            const float u = 1.0f - (sqrtf(dot(d, d)) / dMax);
            wgt[j] = u * u;
            weightSum += wgt[j];
        }
        for (int j = 0; j < K; j++) wgt[j] /= weightSum;
        v3 newPos = v3m(0, 0, 0), newNorm = v3m(0, 0, 0);
        for (int q = 0; q < K; q++) {
            const float* nd = &g[nearNodes[q] * 16];
            const v3 gp = v3m(nd[0], nd[1], nd[2]);
            const float R[9] = {nd[3], nd[6], nd[9], nd[4], nd[7], nd[10], nd[5], nd[8], nd[11]};   // mat3(column0, column1, column2), row-major here
            const v3 tr = v3m(nd[12], nd[13], nd[14]);
            const v3 qd = p - gp;
            const v3 rq = v3m(R[0] * qd.x + R[1] * qd.y + R[2] * qd.z, R[3] * qd.x + R[4] * qd.y + R[5] * qd.z, R[6] * qd.x + R[7] * qd.y + R[8] * qd.z);
            newPos = newPos + ((rq + gp) + tr) * wgt[q];
            // transpose(inverse(rotation)): general 3x3 inverse in cofactor form
            const float c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
            const float det = R[0] * c00 + R[1] * c01 + R[2] * c02;
            const float id = 1.0f / det;
            const float Ri[9] = {c00 * id, (R[2] * R[7] - R[1] * R[8]) * id, (R[1] * R[5] - R[2] * R[4]) * id, c01 * id, (R[0] * R[8] - R[2] * R[6]) * id,
                                 (R[2] * R[3] - R[0] * R[5]) * id, c02 * id, (R[1] * R[6] - R[0] * R[7]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
            const v3 nv = v3m(Ri[0] * n4.x + Ri[3] * n4.y + Ri[6] * n4.z, Ri[1] * n4.x + Ri[4] * n4.y + Ri[7] * n4.z, Ri[2] * n4.x + Ri[5] * n4.y + Ri[8] * n4.z);
            newNorm = newNorm + nv * wgt[q];
        }
        const v3 nn = normalized(newNorm);
        pc[i] = make_float4(newPos.x, newPos.y, newPos.z, p4.w);
        nr[i] = make_float4(nn.x, nn.y, nn.z, n4.w);
        if (p4.w > c.conf && !is_fern) {
            const v3 lp = xf_point(T, newPos);
            const float x = ((c.fx * lp.x) / lp.z) + c.cx, y = ((c.fy * lp.y) / lp.z) + c.cy;
            if (lp.z > 0 && lp.z < c.maxDepth && x > 0 && y > 0 && x < (float)c.w && y < (float)c.h) {
                const float cur = depth_v4[(int)floorf(y) * c.w + (int)floorf(x)].z;
                if (cur > 0.0f && lp.z < cur + 0.1f) tm[i] = make_float2(t2.x, (float)time);
            }
        }
    }
}

// Deformation::sampleGraphModel (sample.vert/.geom): the live surfel number 5000 k of the map order -> x, y, z, init time
__global__ void k_sample_graph(const DevState* __restrict__ st, const int* __restrict__ flags, const int* __restrict__ rank, const float4* __restrict__ pc,
                               const float2* __restrict__ tm, float4* __restrict__ out, int max_n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st->count || !flags[i]) return;
    const int r = rank[i];
    if (r % 5000 != 0 || r / 5000 >= max_n) return;
    const float4 p = pc[i];
    out[r / 5000] = make_float4(p.x, p.y, p.z, tm[i].x);
}

// the constraint samples of EF/ElasticFusion.cpp:568-598: ACTIVE vertex render and INACTIVE time render at the centres of a (w/20) x (h/20)
// grid; record = valid, time, worldRawPoint (currPose * v), worldModelPoint (estPose * v)
__global__ void k_cons_sample(const DevState* __restrict__ st, const float4* __restrict__ pred_vertex, const uint16_t* __restrict__ old_time, int w, int h, float maxDepth,
                              float* __restrict__ out8)
{
    const int rw = w / 20, rh = h / 20, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rw * rh) return;
    const int i = t / rh, j = t - i * rh;   // column by column, as the reference's loops
    const int sx = (i * w + w / 2) / rw, sy = (j * h + h / 2) / rh, k = sy * w + sx;
    const float4 v = pred_vertex[k];
    const int tt = old_time[k];
    float* o = out8 + (size_t)t * 8;
    const bool ok = v.z > 0 && v.z < maxDepth && tt > 0;
    o[0] = ok ? 1.f : 0.f; o[1] = (float)tt;
    const float* P = st->pose;
    const float* E = &st->lc[6];
    for (int r = 0; r < 3; r++) {
        o[2 + r] = P[r * 4] * v.x + P[r * 4 + 1] * v.y + P[r * 4 + 2] * v.z + P[r * 4 + 3] * 1.0f;
        o[5 + r] = E[r * 4] * v.x + E[r * 4 + 1] * v.y + E[r * 4 + 2] * v.z + E[r * 4 + 3] * 1.0f;
    }
}
// currPose = estPose (EF/ElasticFusion.cpp:606)
__global__ void k_adopt_est_pose(DevState* st)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) st->pose[k] = st->lc[6 + k];
    pose_inverse(st->pose, st->pose_inv);
    st->vl_valid = 0;   // another pose, and a deformation follows (positions move): the frame takes the per-pass culls, the list is rebuilt next frame
}

// the first clean pass that sees the store as an upload / a jump of the clock left it: from here on the age rule counts (age_rule_gone)
static inline void age_epoch_begin(ifx* h, int time) { if (h->age_epoch == INT_MAX) h->age_epoch = time; }
static void clean_pass(ifx* h, const float* d_pose_inv, int time, int part = 0)
{
    age_epoch_begin(h, time);
    h->hot_valid = 0;
    h->last_clean_time = time;
    Cam c = make_cam(h);
    if (part == 0) { c.srank = 0; c.sn = 1; }   // a whole pass (stage API, re-render after a compaction) is never sliced
    // A deformation graph was handed in for this clean (local loop closure): IndexMap::synthesizeDepth (EF/ElasticFusion.cpp:667-676) -- splat.vert
    // with time = tick, maxTime = tick - timeDelta, timeDelta = 65535 culls what the INACTIVE prediction culls and depth_splat.frag writes the z of
    // the same ray-disc intersection, so the depth image is the z channel of an INACTIVE prediction of the post-fuse map: rendered into old_vertex.
    const bool deform = h->graph_nodes > 0 && part != 1;
    if (deform && !h->graph_is_fern) raster_pass(h, d_pose_inv, 0, time - h->cfg.time_delta, LIST_SPLAT, nullptr, false, 0, 1);
    // index map of the post-fuse state (EF/ElasticFusion.cpp:662) fused with the clean cull
    if (part != 2) LAUNCH(h, "cull_clean", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_cull_clean, h->d_state, d_pose_inv, (const float4*)h->pc, (const float2*)h->tm, c, time, h->key_index,
           h->list_b, h->list_c);
    if (part == 1) return;
    LAUNCH(h, "index_resolve_taps", dim3(cdiv(h->P, 256)), dim3(256), k_index_resolve, h->d_state, d_pose_inv, h->key_index, (const float4*)h->pc, (const float4*)h->nr,
           (const float2*)h->col, (const float2*)h->tm, h->P, (uint32_t*)nullptr, (float4*)nullptr, (float4*)nullptr, (float4*)nullptr, time, h->cfg.confidence,
           (float4*)h->index_tap, c);
    LAUNCH(h, "clean_list", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_clean_list, h->d_state, d_pose_inv, c, time, (float4*)h->pc, (const float4*)h->nr, (float2*)h->tm,
           (const float4*)h->index_tap, h->list_b, h->list_c);
    if (deform) {
        LAUNCH_SMEM(h, "deform", dim3(MAP_BLOCKS), dim3(256), (size_t)h->graph_nodes * 64, k_deform, h->d_state, d_pose_inv, h->d_graph, h->graph_nodes, h->graph_is_fern, time, c,
                    (const float4*)h->old_vertex, (float4*)h->pc, (float4*)h->nr, (float2*)h->tm);
        h->graph_nodes = 0;   // rawGraph lives for one frame (EF/ElasticFusion.cpp:482)
    }
    const int nb_new = cdiv(h->P, NEW_PER_BLOCK);
    LAUNCH(h, "new_flags_count", dim3(nb_new), dim3(256), k_new_flags_count, h->d_state, d_pose_inv, c, time, h->assoc_target, (const float4*)h->meas_pc, (const float4*)h->meas_nr,
           (const float4*)h->index_tap, h->scan_flags, h->scan_block);
    LAUNCH(h, "append_scan", dim3(nb_new), dim3(256), k_append_scan, h->d_state, c, time, time, h->scan_flags, h->scan_block, nb_new, (const float4*)h->meas_pc,
           (const float4*)h->meas_nr, h->meas_col, h->cap, (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, (float4*)h->ic, (float4*)h->votes,
           h->inst_gt_on ? (const uint8_t*)h->d_inst_gt : (const uint8_t*)nullptr, h->own ? (unsigned int*)nullptr : h->list_v, h->labels, h->seq);
    // the new surfels were never associated: clear the arbitration words nobody reset (losing pixels)
}

// ---- frame path through the cached view list
__global__ void k_vlist_invalidate(DevState* st) { if (threadIdx.x == 0) st->vl_valid = 0; }
void hs_invalidate_view(ifx* h) { h->ids_view_ok = 0; LAUNCH(h, "vlist_invalidate", dim3(1), dim3(64), k_vlist_invalidate, h->d_state); }
static bool use_view_list(ifx* h) { return h->opt_vlist && h->shard_n <= 1 && !h->own && !h->opt_reference_passes && h->graph_nodes == 0 && !h->view_block && h->tick > 1; }
// the same path for the frames of a spatially sharded map (ifx_map_owner_phase): the lists are per-rank supersets of what the rank's shard can show, every pass re-tests its
// entries with the per-pass rule, keys carry creation numbers (key_id) -- the exactness argument is the unsharded one
static bool use_view_list_own(ifx* h) { return h->opt_vlist && h->own && !h->opt_reference_passes && h->graph_nodes == 0 && !h->view_block; }
static void view_scan(ifx* h, int time)
{
    Cam c = make_cam(h);
    c.srank = 0; c.sn = 1;
    // raw output in the clean pass's lists 1, 2 (free at this point of a frame and between frames), then concatenated into list_v / list_vi
    LAUNCH(h, "cull_frame", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_cull_frame, h->d_state, (const float4*)h->pc, (float4*)h->pc, (float2*)h->tm, c, make_planes(c), time, h->list_b, h->list_c,
           h->hot_valid ? (Hot*)h->hot : (Hot*)nullptr,   // (the age rule's tombstones go into the gathered copy too while it is valid)
           h->opt_overdue_rule ? std::max(0, std::min(time, h->last_clean_time)) : -1);   // (the last clean pass any list can have run: the age rule unlisted slots have outlived since)
    if (h->opt_vlist_one) {
        LAUNCH(h, "vlist_flatten", dim3(256, 2), dim3(MAP_THREADS), k_vlist_flatten, h->d_state, c, (const float2*)h->tm, h->list_b, h->list_c, h->list_v, h->list_vi);
        return;
    }
    LAUNCH(h, "vlist_offsets", dim3(1), dim3(64), k_vlist_offsets, h->d_state, c, (const float2*)h->tm);
    LAUNCH(h, "vlist_concat", dim3(256, 2), dim3(MAP_THREADS), k_vlist_concat, (const DevState*)h->d_state, c, h->list_b, h->list_c, h->list_v, h->list_vi);
}
// A forced scan at the current pose with the time of the last processed frame: every slot the list leaves out gets the age rule
// it may have outlived (see "View list"); cheap no-op when the view-list path never ran since the last scan of this kind.
int ifx_vlist_reap(ifx* h)
{
    if (!h->view_dirty) return IFX_OK;
    h->view_dirty = 0;
    h->ids_view_ok = 0;
    LAUNCH(h, "vlist_decide", dim3(1), dim3(64), k_vlist_decide, h->d_state, h->d_list_ctr, 1);
    view_scan(h, h->last_clean_time);
    return IFX_OK;
}
static void index_list_pass(ifx* h, int time, bool taps)
{
    Cam c = make_cam(h);
    c.srank = 0; c.sn = 1;
    LAUNCH(h, "index_list", dim3(h->opt_index_blocks > 0 ? h->opt_index_blocks : LIST_BLOCKS), dim3(MAP_THREADS), k_index_list, (const DevState*)h->d_state, (const float4*)h->pc, (const float2*)h->tm, c, time, h->list_v, h->key_index, (const Hot*)h->frame_hot);
    if (!taps)
        LAUNCH(h, "index_resolve", dim3(cdiv(h->P, 256)), dim3(256), k_index_resolve, h->d_state, (const float*)nullptr, h->key_index, (const float4*)h->pc, (const float4*)h->nr,
               (const float2*)h->col, (const float2*)h->tm, h->P, h->index_id, (float4*)h->index_vc, (float4*)nullptr, (float4*)h->index_nr, time, h->cfg.confidence, (float4*)nullptr, c,
               (const int32_t*)nullptr, (const Hot*)h->frame_hot);
    else
        LAUNCH(h, "index_resolve_taps", dim3(cdiv(h->P, 256)), dim3(256), k_index_resolve, h->d_state, (const float*)nullptr, h->key_index, (const float4*)h->pc, (const float4*)h->nr,
               (const float2*)h->col, (const float2*)h->tm, h->P, (uint32_t*)nullptr, (float4*)nullptr, (float4*)nullptr, (float4*)nullptr, time, h->cfg.confidence, (float4*)h->index_tap, c,
               (const int32_t*)nullptr, (const Hot*)h->frame_hot);
}

// the clean pass and the append of a view-list frame as launches of their own (k_clean_view ; k_new_flags_count ; k_append_scan)
static void view_clean_append(ifx* h, const Cam& c, int time)
{
    h->hot_valid = 0;
    LAUNCH(h, "clean_view", dim3(h->opt_clean_blocks > 0 ? h->opt_clean_blocks : 2 * LIST_BLOCKS), dim3(MAP_THREADS), k_clean_view, h->d_state, c, time, (float4*)h->pc, (const float4*)h->nr, (float2*)h->tm,
           (const float4*)h->index_tap, h->list_v);
    const int nb_new = cdiv(h->P, NEW_PER_BLOCK);
    LAUNCH(h, "new_flags_count", dim3(nb_new), dim3(256), k_new_flags_count, h->d_state, (const float*)nullptr, c, time, h->assoc_target, (const float4*)h->meas_pc,
           (const float4*)h->meas_nr, (const float4*)h->index_tap, h->scan_flags, h->scan_block);
    LAUNCH(h, "append_scan", dim3(nb_new), dim3(256), k_append_scan, h->d_state, c, time, time, h->scan_flags, h->scan_block, nb_new, (const float4*)h->meas_pc,
           (const float4*)h->meas_nr, h->meas_col, h->cap, (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, (float4*)h->ic, (float4*)h->votes,
           h->inst_gt_on ? (const uint8_t*)h->d_inst_gt : (const uint8_t*)nullptr, h->own ? (unsigned int*)nullptr : h->list_v, h->labels, h->seq);
}

// EF/ElasticFusion.cpp:620-694 without the loop-closure branches
int ifx_map_frame(ifx* h)
{
    h->clean_raster_pending = 0;
    h->view_frame = 0;
    h->ids_view_ok = 0;
    age_epoch_begin(h, h->tick);
    if (use_view_list(h)) {
        Cam c = make_cam(h);
        c.srank = 0; c.sn = 1;
        const int time = h->tick;
        // Option clean_raster (default): the clean, the new surfels' flags and the append are left to the end-of-frame prediction (ifx_map_predict): ONE walk of the
        // view list cleans and rasterises (+ the flags), then the append, then the resolve.  Only when that prediction will take the view-list raster, and while no new
        // surfel can be drawn in the frame that creates it: its confidence starts at most at max(1, weight multiplier) (confidence_fn, k_track_end), the renders draw
        // from the threshold on (splat.vert:56-65, surfel_ids.vert:45).
        const bool fused = h->opt_clean_raster && !h->opt_compact_every_frame && h->last_compact_tick != h->tick && h->opt_raster_tiles <= 0 && !h->opt_raster_lds &&
                           h->cfg.confidence > fmaxf(1.f, h->frame_weight_mult);
        // the gathered copy of the store ("hot records") serves the frames that take the fused path: rebuilt here when something outside the frame path wrote the store since
        Hot* hot = nullptr;
        if (fused && h->opt_hot) {
            if (!h->hot && hipMalloc(&h->hot, (size_t)h->cap * sizeof(Hot)) != hipSuccess) { h->hot = nullptr; (void)hipGetLastError(); }
            if (h->hot && !h->hot_valid) {
                LAUNCH(h, "hot_rebuild", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_hot_rebuild, (const DevState*)h->d_state, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->tm, (Hot*)h->hot);
                h->hot_valid = 1;
            } else if (h->hot && h->opt_hot_verify)
                LAUNCH(h, "hot_verify", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_hot_verify, h->d_state, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->tm, (Hot*)h->hot);
            hot = (Hot*)h->hot;
        }
        if (!hot) h->hot_valid = 0;
        h->frame_hot = hot;
        if (h->view_scan_tick != time) view_scan(h, time);   // rebuilds the list when vlist_decide asked for it, returns at once otherwise (the loop-closure renders may have taken it already)
        index_list_pass(h, time, false);        // predictIndices of the pre-fuse map (:620)
        fuse_pass(h, nullptr, 0.f, time, 0, nullptr, hot);
        index_list_pass(h, time, true);         // predictIndices of the post-fuse map (:662), resolved into the clean pass's tap records
        h->clean_raster_pending = fused ? 1 : 0;
        if (!fused) view_clean_append(h, c, time);
        h->view_frame = 1; h->view_dirty = 1; h->last_clean_time = time;
        if (h->opt_compact_every_frame) ifx_compact_enqueue(h, 0);   // (reaps first; the raster below then takes the per-pass cull: the list is void after a compaction)
        h->ids_pending = 1;
        return IFX_OK;
    }
    h->frame_hot = nullptr;
    if (h->opt_vlist && !h->own && h->shard_n <= 1) hs_invalidate_view(h);   // this frame runs no scan: a device-side "valid" must never describe a list the host did not build
    index_pass(h, nullptr, h->tick, true);
    fuse_pass(h, nullptr, 0.f, h->tick);
    if (h->opt_reference_passes) {   // renders nobody on the path consumes (EF/ElasticFusion.cpp:679-680); they need the post-fuse index map too
        index_pass(h, nullptr, h->tick);
        ids_pass(h, nullptr, 1, h->ids_tmp);
        ids_pass(h, nullptr, 0, h->ids_tmp);
    }
    clean_pass(h, nullptr, h->tick);   // includes the second predictIndices
    h->view_block = 0;
    if (h->opt_compact_every_frame) ifx_compact_enqueue(h, 0);
    h->ids_pending = 1;                // rendered together with the prediction (same map state, same pose)
    return IFX_OK;
}

// Loop-closure detection, map side (EF/ElasticFusion.cpp:453 and :519-526): predict() at the pose just tracked (pre-fusion map), then the
// INACTIVE prediction -- surfels last seen at or before tick - timeDelta (splat.vert:60 with time = 0, maxTime = tick - timeDelta) -- into
// the old* images.  The first render goes to images of its own (act*): pred_* belong to the frame-to-model tracker.
int ifx_map_predict_loop_closure(ifx* h)
{
    // both renders see the same map at the same pose and differ only in the time window: one scan of the store, one raster launch (two key images)
    Cam c = make_cam(h);
    c.srank = 0; c.sn = 1;
    const bool by_view = use_view_list(h) && h->opt_lc_view;
    if (by_view) {   // the frame's view lists hold every stable surfel in view, inside the time window or not: the scan this frame needs anyway, taken first
        view_scan(h, h->tick);
        h->view_scan_tick = h->tick;
        LAUNCH(h, "raster_view_lc", dim3(h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS), dim3(MAP_THREADS), (k_raster_view<false, false>), h->d_state, (const float4*)h->pc, (const float4*)h->nr,
               (const float2*)h->tm, c, h->tick, h->tick, LIST_SPLAT | LIST_DUAL, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, h->opt_raster_earlyz, 1, CleanArgs{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (const DevState*)h->d_state);
    } else {
    LAUNCH(h, "cull_raster", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_cull_raster, h->d_state, (const float*)nullptr, (const float4*)h->pc, (const float2*)h->tm, c, h->tick, h->tick,
           LIST_SPLAT | LIST_DUAL, h->list_a, (unsigned int*)nullptr, 0);
    LAUNCH(h, "raster_list", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_raster_list, h->d_state, (const float*)nullptr, (const float4*)h->pc, (const float4*)h->nr, c, h->list_a, h->key_splat,
           h->key_ids, h->key_both, 1, (const int*)nullptr);
    }
    {
        ResolveTarget ta, to;
        ta.keys = h->key_splat; ta.pv = (float4*)h->act_vertex; ta.pn = (float4*)h->act_normal; ta.pimg = (uchar4*)h->act_image; ta.pinst = (uchar4*)h->act_inst; ta.ptime = h->act_time;
        to.keys = h->key_ids; to.pv = (float4*)h->old_vertex; to.pn = (float4*)h->old_normal; to.pimg = (uchar4*)h->old_image; to.pinst = (uchar4*)h->old_inst; to.ptime = h->old_time;
        LAUNCH(h, "splat_resolve_lc", dim3(cdiv(h->w, 32), cdiv(h->h, 8), 2), dim3(32, 8), k_splat_resolve_pair, (const DevState*)h->d_state, (const float4*)h->pc, (const float4*)h->nr,
               (const float2*)h->col, (const float2*)h->tm, c, (const uint8_t*)h->rgb, (const uint16_t*)h->depth_filt, ta, to);
    }
    if (!by_view)   // (re-arms work list 0, which the view-list path does not touch)
        LAUNCH(h, "raster_finish", dim3(1), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, 0, h->ids_after, (const float4*)h->votes, h->cap, 10, h->d_list_ctr, ifx_idmap(h));
    return IFX_OK;
}

// ElasticFusion::predict, EF/ElasticFusion.cpp:729-763, fused with renderSurfelIds(GENERAL_AFTER) of :694
int ifx_map_predict(ifx* h)
{
    static const bool no_ids = getenv("IFX_EXPERIMENT_NO_IDS") != nullptr;   // measurement only: what the per-frame id render costs (the id image is then stale)
    const unsigned int want = LIST_SPLAT | ((h->ids_pending && !no_ids) ? LIST_IDS : 0u);
    // the tiled rasteriser only on request: at 1280x960 / 20 M surfels the view-list rasteriser takes 455 us where cull + bin + tile raster take 744 (profiles/archive/r02_o_1280_20m.txt)
    const bool tiles = h->opt_raster_tiles > 0;
    if (h->view_frame && !(h->opt_compact_every_frame || h->last_compact_tick == h->tick) && !tiles) {   // the frame built / checked the view list and nothing renumbered the store since
        Cam c = make_cam(h);
        c.srank = 0; c.sn = 1;
        // The id image has one consumer per frame -- whetherDoSegmentation's sums over every 10th pixel -- and a full-image consumer only when a segmentation call, a
        // download or the display asks for it: the frame renders the sampled lattice only (the id half of this pass: 71 -> 40 us), ifx_ids_ensure the rest on demand.
        const int ids_step = (h->opt_lazy_ids && !h->ids_full_hint && (want & LIST_IDS)) ? 10 : 1;
        h->ids_full_hint = 0;
        h->ids_full_valid = ids_step == 1;
        h->ids_sparse_frame = ids_step > 1;
        h->ids_view_ok = ids_step > 1;
        if (h->clean_raster_pending) {   // the frame's clean + new-surfel flags in the raster's walk, then the append, then the resolve (see ifx_map_frame)
            h->clean_raster_pending = 0;
            const int nb_new = cdiv(h->P, NEW_PER_BLOCK);
            CleanArgs ca;
            ca.pc_rw = (float4*)h->pc; ca.tm_rw = (float2*)h->tm; ca.tap = (const float4*)h->index_tap; ca.nf_blocks = nb_new; ca.assoc = h->assoc_target; ca.mpc = (const float4*)h->meas_pc;
            ca.mnr = (const float4*)h->meas_nr; ca.flags = h->scan_flags; ca.block_counts = h->scan_block; ca.hot = (Hot*)h->frame_hot;
            LAUNCH(h, "clean_raster_view", dim3(nb_new + (h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS)), dim3(MAP_THREADS), (k_raster_view<false, true>), h->d_state, (const float4*)h->pc, (const float4*)h->nr,
                   (const float2*)h->tm, c, h->tick, h->tick, want, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, h->opt_raster_earlyz, ids_step, ca, (const DevState*)h->d_state);
            LAUNCH(h, "append_scan", dim3(nb_new), dim3(256), k_append_scan, h->d_state, c, h->tick, h->tick, h->scan_flags, h->scan_block, nb_new, (const float4*)h->meas_pc,
                   (const float4*)h->meas_nr, h->meas_col, h->cap, (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, (float4*)h->ic, (float4*)h->votes,
                   h->inst_gt_on ? (const uint8_t*)h->d_inst_gt : (const uint8_t*)nullptr, h->list_v, h->labels, h->seq, (Hot*)h->frame_hot);
            raster_pass(h, nullptr, h->tick, h->tick, want, h->ids_after, true, 2, 0, h->opt_fold_finish != 0, ids_step, true);
        } else {
        if (h->opt_raster_lds)
            LAUNCH(h, "raster_view", dim3(h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS), dim3(MAP_THREADS), (k_raster_view<true, false>), h->d_state, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->tm, c, h->tick, h->tick,
                   want, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, h->opt_raster_earlyz, ids_step, CleanArgs{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (const DevState*)h->d_state);
        else
            LAUNCH(h, "raster_view", dim3(h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS), dim3(MAP_THREADS), (k_raster_view<false, false>), h->d_state, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->tm, c, h->tick, h->tick,
                   want, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, h->opt_raster_earlyz, ids_step, CleanArgs{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (const DevState*)h->d_state);
        raster_pass(h, nullptr, h->tick, h->tick, want, h->ids_after, true, 2, 0, h->opt_fold_finish != 0, ids_step);   // resolve + the end-of-pass sums in the same launch
        }
    } else {
        if (h->clean_raster_pending) {   // (the host-side conditions of the two functions are the same: never) -- the deferred passes as launches of their own
            h->clean_raster_pending = 0;
            Cam cc = make_cam(h);
            cc.srank = 0; cc.sn = 1;
            view_clean_append(h, cc, h->tick);
        }
        raster_pass(h, nullptr, h->tick, h->tick, want, h->ids_after, true);
        if (want & LIST_IDS) { h->ids_full_valid = 1; h->ids_sparse_frame = 0; }
        h->ids_view_ok = 0;
    }
    h->view_frame = 0;
    h->ids_pending = 0;
    return IFX_OK;
}

// ------------------------------------------------------------------ sharded projection
// Every rank keeps the full map replica and runs the whole frame deterministically, so the replicas stay bit-identical;
// only the projection passes (index map x2, splat + id raster) -- the passes that stream the whole surfel store with one
// atomic per visible surfel / covered pixel -- are cut by slot range (Cam::srank / sn), and the ranks exchange their key
// images by an element-wise unsigned minimum between the phases (min over disjoint slices == global min, ties included,
// because the key carries the slot id).  The exchange itself is the caller's: a RCCL all-reduce(MIN) over xGMI through
// torch.distributed in instancefusion_amd/sharded.py.
//   phase 0: index projection of the slice                         | exchange key_index
//   phase 1: resolve, association, fusion, clean cull + projection | exchange key_index
//   phase 2: tap resolve, clean, append, raster cull + raster      | exchange key_splat, key_ids, key_both
//   phase 3: splat / id resolve, finish
int ifx_map_sharded_phase(ifx* h, int phase, bool first_frame)
{
    h->hot_valid = 0;
    if (first_frame) {   // no map yet: the first-frame initialisation is replicated; only the prediction raster is sliced
        if (phase == 0) return ifx_map_init_first(h);
        if (phase == 2) raster_pass(h, nullptr, h->tick, h->tick, LIST_SPLAT, h->ids_after, true, 1);   // as ifx_map_predict on the first frame: no id render yet
        if (phase == 3) raster_pass(h, nullptr, h->tick, h->tick, LIST_SPLAT, h->ids_after, true, 2);
        return IFX_OK;
    }
    switch (phase) {
    case 0: index_pass(h, nullptr, h->tick, true, 1); break;
    case 1: index_pass(h, nullptr, h->tick, true, 2); fuse_pass(h, nullptr, 0.f, h->tick); clean_pass(h, nullptr, h->tick, 1); break;
    case 2: clean_pass(h, nullptr, h->tick, 2); raster_pass(h, nullptr, h->tick, h->tick, LIST_SPLAT | LIST_IDS, h->ids_after, true, 1); break;
    case 3: raster_pass(h, nullptr, h->tick, h->tick, LIST_SPLAT | LIST_IDS, h->ids_after, true, 2); break;
    default: return IFX_E_INVALID;
    }
    return IFX_OK;
}

// ------------------------------------------------------------------ spatially sharded map (ifx_config::n_ranks > 1)
// One process per GPU; this handle STORES only the surfels it owns (owner = Morton(8 cm voxel of the creation position) mod n_ranks,
// ifx_owner_of_point), i.e. 1 / n_ranks of the map.  Every rank is fed the same frame.  Per-surfel work -- the three projections,
// the fusion update, the clean pass -- runs on the local shard; per-pixel work -- tracking, association, the stability test of new
// surfels -- is replicated (P pixels against N / G surfels: cheaper than routing it).  Between the phases of a frame the ranks combine
//   * their key images by an element-wise unsigned 64-bit MIN: keys carry the CREATION NUMBER of a surfel instead of its slot, the same
//     number on every rank and ascending in map order, so the nearest surfel and the tie-break are those of one GPU;
//   * the attribute images of the winners (index map, clean taps, prediction) by a bitwise SUM over int32: a pixel is written by the one
//     rank that owns its winner (binary search of the creation number in the local store), all others contribute zeros -- the
//     cross-shard reprojection exchange of the north star, as an all-reduce over disjoint supports.
// New surfels need no routing: every rank computes the same append list and keeps the ones it owns, numbering all of them alike.
// Compaction is local and independent (ids are creation numbers, not slots).  The result is bit-identical to one GPU:
// tests/test_gpu_parity.py::test_owner_sharded_map_emulated.  Exchange volume per frame at W x H pixels: keys 5 x 8 B, attributes
// 32 + 16 + 42 B per pixel = 130 B/pixel (40 MB at 640x480) whatever the map size; DESIGN.md section 7 has the cost model.
__global__ void k_owner_flags(const DevState* __restrict__ st, const float4* __restrict__ pc, int n_ranks, int rank, int* __restrict__ flags, int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    int f = 0;
    if (i < st->count) { const float4 p = pc[i]; f = ifx_owner_of_point(p.x, p.y, p.z, n_ranks) == rank; }
    flags[i] = f;
}
__global__ void k_owner_count(DevState* st, const int* total)
{
    if (threadIdx.x == 0) { st->next_seq = (unsigned int)st->count; st->count = *total; st->n_dead = 0; }
}
// fill_rgb / fill_vertex / fill_normal.frag on the exchanged prediction (EF/Shaders/FillIn.cpp:65-195): the second half of k_splat_resolve
__global__ void k_fill_in(Cam c, const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth_filt, float4* __restrict__ pv, const float4* __restrict__ pn,
                          const uchar4* __restrict__ pimg, float4* __restrict__ fv, float4* __restrict__ fn, uchar4* __restrict__ fimg, const float* __restrict__ pconf)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= c.w || y >= c.h) return;
    const int k = y * c.w + x;
    float4 vo = pv[k];
    const float4 no = pn[k];
    if (pconf) { vo.w = pconf[k]; pv[k] = vo; }   // the winner's confidence, summed across the ranks apart from the vertex every rank rebuilt itself
    const uchar4 io = pimg[k];
    float ifx_ = 1.0f / c.fx, ify_ = 1.0f / c.fy;
    if ((int)io.x + (int)io.y + (int)io.z == 0) fimg[k] = make_uchar4(rgb[k * 3], rgb[k * 3 + 1], rgb[k * 3 + 2], 255);
    else fimg[k] = io;
    float zc = (float)depth_filt[k] / 1000.0f;
    if (vo.z == 0) fv[k] = make_float4(((float)x - c.cx) * zc * ifx_, ((float)y - c.cy) * zc * ify_, zc, 1.f);
    else fv[k] = vo;
    if (no.z == 0) {
        v3 vp = v3m(((float)x - c.cx) * zc * ifx_, ((float)y - c.cy) * zc * ify_, zc);
        int xr = clampi(x + 1, 0, c.w - 1), yd = clampi(y + 1, 0, c.h - 1);
        float zx = (float)depth_filt[y * c.w + xr] / 1000.0f, zy = (float)depth_filt[yd * c.w + x] / 1000.0f;
        v3 vx = v3m(((float)(x + 1) - c.cx) * zx * ifx_, ((float)y - c.cy) * zx * ify_, zx);
        v3 vy = v3m(((float)x - c.cx) * zy * ifx_, ((float)(y + 1) - c.cy) * zy * ify_, zy);
        v3 nn = normalized(cross(vx - vp, vy - vp));
        fn[k] = make_float4(nn.x, nn.y, nn.z, 1.f);
    } else fn[k] = no;
}

// first frame: the dense initialisation is computed by every rank, which then keeps its own surfels (creation numbers = the unsharded slots)
static void owner_filter(ifx* h)
{
    const int n = h->cap;
    LAUNCH(h, "owner_flags", dim3(cdiv(n, 256)), dim3(256), k_owner_flags, (const DevState*)h->d_state, (const float4*)h->pc, h->own_g, h->cfg.rank, h->scan_flags, n);
    ifx_scan_exclusive(h, h->scan_flags, n, h->scan_out, &h->d_state->seg_counts[1]);
    LAUNCH(h, "compact_scatter", dim3(cdiv(n, 256)), dim3(256), k_compact_scatter, h->scan_flags, h->scan_out, n, (const float4*)h->pc, (const float4*)h->nr,
           (const float2*)h->col, (const float2*)h->tm, (const float4*)h->ic, (const float4*)h->votes, (const int32_t*)h->labels, (float4*)h->pc2, (float4*)h->nr2,
           (float2*)h->col2, (float2*)h->tm2, (float4*)h->ic2, (float4*)h->votes2, h->labels2, (const uint32_t*)h->seq, h->seq2);
    LAUNCH(h, "owner_count", dim3(1), dim3(64), k_owner_count, h->d_state, &h->d_state->seg_counts[1]);
    std::swap(h->pc, h->pc2); std::swap(h->nr, h->nr2); std::swap(h->col, h->col2); std::swap(h->tm, h->tm2); std::swap(h->ic, h->ic2); std::swap(h->votes, h->votes2); std::swap(h->labels, h->labels2); std::swap(h->seq, h->seq2);
}

// The pixels a surfel covers in BOTH renders of a raster pass went to key_both (one atomic instead of two); before the keys travel they are folded back
// into the two renders' own images, so that an exchange point moves two key images instead of three (the resolve's min with key_both then finds it empty).
__global__ void k_merge_both(unsigned long long* __restrict__ ks, unsigned long long* __restrict__ ki, unsigned long long* __restrict__ kb, int P)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const unsigned long long b = kb[k];
    if (b == IFX_KEY_EMPTY) return;
    const unsigned long long a = ks[k], c = ki[k];
    ks[k] = a < b ? a : b;
    ki[k] = c < b ? c : b;
    kb[k] = IFX_KEY_EMPTY;
}

// k_merge_both and the k_own_translate of [key_splat | key_ids] in one launch (the frame path of a sharded map whose raster drew slots): a pixel's three keys are its own
__global__ void k_own_merge_translate(unsigned long long* __restrict__ ks, unsigned long long* __restrict__ ki, unsigned long long* __restrict__ kb, int P, const uint32_t* __restrict__ seq,
                                      int32_t* __restrict__ slot_s, int32_t* __restrict__ slot_i, const DevState* __restrict__ st, const float2* __restrict__ tm, unsigned long long* __restrict__ gfl_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && gfl_out) own_first_live(st, tm, seq, gfl_out);   // (the word that travels with these keys)
    if (k >= P) return;
    const unsigned long long b = kb[k];
    unsigned long long a = ks[k], c = ki[k];
    if (b != IFX_KEY_EMPTY) { a = a < b ? a : b; c = c < b ? c : b; kb[k] = IFX_KEY_EMPTY; }
    if (a == IFX_KEY_EMPTY) slot_s[k] = -1;
    else { const unsigned int s_ = (unsigned int)(a & 0xFFFFFFFFull); slot_s[k] = (int32_t)s_; ks[k] = (a & 0xFFFFFFFF00000000ull) | (unsigned long long)seq[s_]; }
    if (c == IFX_KEY_EMPTY) slot_i[k] = -1;
    else { const unsigned int s_ = (unsigned int)(c & 0xFFFFFFFFull); slot_i[k] = (int32_t)s_; ki[k] = (c & 0xFFFFFFFF00000000ull) | (unsigned long long)seq[s_]; }
}

// phase p of a frame of the sharded map; the buffers ifx_owner_exchange(p) lists are reduced across the ranks before phase p + 1 -- by the library itself
// on its communicator (ifx_comm.hip: ifx_owner_process_frame_device), or by the caller (the emulation tests).  phase 104..106: ElasticFusion::predict
// outside a frame (ifx_owner_predict_phase): phases 4..6 without the clean / append and without the whetherDoSegmentation sums.
#define OWN_FIRST_LIVE(h, out) do { if ((out) && (h)->opt_own_first_live) LAUNCH((h), "own_first_live", dim3(1), dim3(1), k_own_first_live, (const DevState*)(h)->d_state, (const float2*)(h)->tm, (const uint32_t*)(h)->seq, (out)); } while (0)
#define OWN_GFL(h, out) (const DevState*)(h)->d_state, (const float2*)(h)->tm, ((h)->opt_own_first_live ? (out) : (unsigned long long*)nullptr)
int ifx_map_owner_phase(ifx* h, int phase, bool first_frame)
{
    h->hot_valid = 0;
    Cam c = make_cam(h);
    c.srank = 0; c.sn = 1;
    if (h->gfl_index && phase >= 1 && phase <= 3) c.first_live = (const int*)h->gfl_index;   // (behind exchanges 0 / 2: the word that came with key_index; everywhere else the one behind [key_splat | key_ids])
    const int time = h->tick;
    const dim3 b2(32, 8), g2(cdiv(h->w, 32), cdiv(h->h, 8));
    const bool in_frame = phase < 100 || phase >= 300;
    if (phase >= 105 && phase < 300) phase -= 100;
    if (first_frame) {
        switch (phase) {
        case 0: ifx_map_init_first(h); owner_filter(h); break;
        case 4: h->own_ids_lat = 0; raster_pass(h, nullptr, time, time, LIST_SPLAT, h->ids_after, false, 1); break;
        case 5:
            LAUNCH(h, "splat_resolve", g2, b2, k_splat_resolve, h->d_state, (const float*)nullptr, h->key_splat, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->col,
                   (const float2*)h->tm, c, h->rgb, h->depth_filt, (float4*)h->pred_vertex, (float4*)h->pred_normal, (uchar4*)h->pred_image, (uchar4*)h->pred_inst, h->pred_time,
                   (float4*)nullptr, (float4*)nullptr, (uchar4*)nullptr, h->key_ids, h->key_both, (int32_t*)nullptr, (int*)nullptr, FinishFold(), h->pred_conf);
            break;
        case 6:
            LAUNCH(h, "fill_in", g2, b2, k_fill_in, c, (const uint8_t*)h->rgb, (const uint16_t*)h->depth_filt, (float4*)h->pred_vertex, (const float4*)h->pred_normal,
                   (const uchar4*)h->pred_image, (float4*)h->fill_vertex, (float4*)h->fill_normal, (uchar4*)h->fill_image, (const float*)h->pred_conf);
            LAUNCH(h, "raster_finish", dim3(1), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, 1, h->ids_after, (const float4*)h->votes, h->cap, 10, h->d_list_ctr, ifx_idmap(h),
                   (int*)nullptr, h->pred_tail);
            break;
        default: break;   // (incl. 7)
        }
        return IFX_OK;
    }
    switch (phase) {
    // ---- local loop-closure detection on the sharded map (EF/ElasticFusion.cpp:453-566), before the frame's map passes: predict() at the tracked pose and the
    // INACTIVE prediction from one scan of the local shard (as ifx_map_predict_loop_closure), the owners' winners of both renders, then the model-to-model
    // tracker replicated on the exchanged images (ifx_api.hip: owner_frame_phase)
    case 300:                                                                                               // local dual raster | [key_splat (ACTIVE) | key_ids (INACTIVE)]: MIN
        LAUNCH(h, "cull_raster", dim3(MAP_BLOCKS), dim3(MAP_THREADS), k_cull_raster, h->d_state, (const float*)nullptr, (const float4*)h->pc, (const float2*)h->tm, c, time, time,
               LIST_SPLAT | LIST_DUAL, h->list_a, (unsigned int*)nullptr, 0);
        LAUNCH(h, "raster_list", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_raster_list, h->d_state, (const float*)nullptr, (const float4*)h->pc, (const float4*)h->nr, c, h->list_a, h->key_splat,
               h->key_ids, h->key_both, 1, (const int*)nullptr);
        break;
    case 301:                                                                                               // owned winners of both renders | [act_* | old_*]: SUM
        for (int old = 0; old < 2; old++)
            LAUNCH(h, old ? "splat_resolve_old" : "splat_resolve_act", g2, b2, k_splat_resolve, h->d_state, (const float*)nullptr, old ? h->key_ids : h->key_splat, (const float4*)h->pc,
                   (const float4*)h->nr, (const float2*)h->col, (const float2*)h->tm, c, h->rgb, h->depth_filt, (float4*)(old ? h->old_vertex : h->act_vertex),
                   (float4*)(old ? h->old_normal : h->act_normal), (uchar4*)(old ? h->old_image : h->act_image), (uchar4*)(old ? h->old_inst : h->act_inst),
                   old ? h->old_time : h->act_time, (float4*)nullptr, (float4*)nullptr, (uchar4*)nullptr, h->key_ids, h->key_both, (int32_t*)nullptr, (int*)nullptr, FinishFold());
        LAUNCH(h, "raster_finish", dim3(1), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, 0, h->ids_after, (const float4*)h->votes, h->cap, 10, h->d_list_ctr, ifx_idmap(h),
               (int*)nullptr, (int*)nullptr);
        break;
    case 0:                                                                                                 // local projection | keys: MIN
        age_epoch_begin(h, time);
        c.age_epoch = h->age_epoch;
        h->view_frame = 0;
        h->ids_view_ok = 0;
        if (h->own_need_decide) {   // the pose came from the tracking rank (exchange 310): this rank has not yet asked whether its cached lists still cover it
            LAUNCH(h, "vlist_decide", dim3(1), dim3(64), k_vlist_decide, h->d_state, h->d_list_ctr, 0);
            h->own_need_decide = 0;
        }
        h->own_fast = 0; h->own_fast_raster = 0;
        if (use_view_list_own(h)) {
            if (h->view_scan_tick != time) view_scan(h, time);   // one scan of the shard when the lists are stale; returns at once otherwise
            h->view_frame = 1;
            if (!h->own_slot_img && hipMalloc(&h->own_slot_img, (size_t)h->P * 4 * sizeof(int32_t)) != hipSuccess) h->own_slot_img = nullptr;   // [index | splat | ids | association] slot images
            h->own_fast = h->own_slot_img != nullptr;
            Cam cl = c;
            if (h->own_fast) { cl.own_n = 0; cl.raw_slots = 1; }   // local keys carry slots; k_own_translate swaps in the creation numbers before they travel
            LAUNCH(h, "index_list", dim3(h->opt_index_blocks > 0 ? h->opt_index_blocks : LIST_BLOCKS), dim3(MAP_THREADS), k_index_list, (const DevState*)h->d_state, (const float4*)h->pc, (const float2*)h->tm, cl, time, h->list_v, h->key_index);
            if (h->own_fast) LAUNCH(h, "own_translate", dim3(cdiv(h->P, 256)), dim3(256), k_own_translate, h->key_index, h->P, (const uint32_t*)h->seq, h->own_slot_img, OWN_GFL(h, h->gfl_index));   // (the word: behind the view-list scan, whose age rule removes surfels)
            else OWN_FIRST_LIVE(h, h->gfl_index);
        } else {
            if (h->opt_vlist) hs_invalidate_view(h);   // no scan this frame: a device-side "valid" must never describe lists the host did not maintain
            index_pass(h, nullptr, time, true, 1);
            OWN_FIRST_LIVE(h, h->gfl_index);
        }
        break;
    case 1: index_pass(h, nullptr, time, true, 2, h->own_fast ? h->own_slot_img : nullptr); fuse_pass(h, nullptr, 0.f, time, 1); break;   // attributes of the winners this rank owns, association among them | assoc_key: MIN
    case 2:                                                                                                 // verdicts decoded, update (owned), post-fuse projection | keys: MIN
        fuse_pass(h, nullptr, 0.f, time, 2, h->own_fast ? h->own_slot_img : nullptr);
        if (h->view_frame) {
            Cam cl = c;
            if (h->own_fast) { cl.own_n = 0; cl.raw_slots = 1; }
            LAUNCH(h, "index_list", dim3(h->opt_index_blocks > 0 ? h->opt_index_blocks : LIST_BLOCKS), dim3(MAP_THREADS), k_index_list, (const DevState*)h->d_state, (const float4*)h->pc, (const float2*)h->tm, cl, time, h->list_v, h->key_index);
            if (h->own_fast) LAUNCH(h, "own_translate", dim3(cdiv(h->P, 256)), dim3(256), k_own_translate, h->key_index, h->P, (const uint32_t*)h->seq, h->own_slot_img);
        } else clean_pass(h, nullptr, time, 1);
        break;
    case 3:                                                                                                 // owned tap records | index_tap: SUM
        LAUNCH(h, "index_resolve_taps", dim3(cdiv(h->P, 256)), dim3(256), k_index_resolve, h->d_state, (const float*)nullptr, h->key_index, (const float4*)h->pc, (const float4*)h->nr,
               (const float2*)h->col, (const float2*)h->tm, h->P, (uint32_t*)nullptr, (float4*)nullptr, (float4*)nullptr, (float4*)nullptr, time, h->cfg.confidence,
               (float4*)h->index_tap, c, (h->own_fast && h->view_frame) ? (const int32_t*)h->own_slot_img : (const int32_t*)nullptr);
        break;
    case 4: {                                                                                               // clean (local), append (replicated list, owned kept), local raster | [key_splat | key_ids]: MIN
        const int nb_new = cdiv(h->P, NEW_PER_BLOCK);
        const bool lat = h->opt_own_lazy_ids != 0;   // (a replicated switch: every rank's exchange 4 has the same form, whichever way its local raster went)
        bool whole_drawn = false;
        h->own_ids_lat = lat;
        // ONE walk of the view list cleans and rasterises (+ the new surfels' flags), then the append -- the single-GPU frame's form (ifx_map_frame, option clean_raster), under the
        // same conditions: the list is this frame's, nothing renumbers the shard in between, and no new surfel can be drawn in the frame that creates it.  The walk draws slots
        // (k_own_translate swaps in the creation numbers); its new-surfel blocks keep the owner filter of the Cam.
        const bool fused = h->opt_clean_raster && h->view_frame && h->own_fast && !h->opt_compact_every_frame && h->last_compact_tick != h->tick && h->opt_raster_tiles <= 0 &&
                           h->cfg.confidence > fmaxf(1.f, h->frame_weight_mult);
        if (fused) {
            CleanArgs ca;
            ca.pc_rw = (float4*)h->pc; ca.tm_rw = (float2*)h->tm; ca.tap = (const float4*)h->index_tap; ca.nf_blocks = nb_new; ca.assoc = h->assoc_target; ca.mpc = (const float4*)h->meas_pc;
            ca.mnr = (const float4*)h->meas_nr; ca.flags = h->scan_flags; ca.block_counts = h->scan_block; ca.hot = nullptr;
            LAUNCH(h, "clean_raster_view", dim3(nb_new + (h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS)), dim3(MAP_THREADS), (k_raster_view<false, true>), h->d_state, (const float4*)h->pc, (const float4*)h->nr,
                   (const float2*)h->tm, c, time, time, LIST_SPLAT | LIST_IDS, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, 0, lat ? OWN_LAT_DS : 1, ca, (const DevState*)h->d_state);
            LAUNCH(h, "append_scan", dim3(nb_new), dim3(256), k_append_scan, h->d_state, c, time, time, h->scan_flags, h->scan_block, nb_new, (const float4*)h->meas_pc,
                   (const float4*)h->meas_nr, h->meas_col, h->cap, (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, (float4*)h->ic, (float4*)h->votes,
                   h->inst_gt_on ? (const uint8_t*)h->d_inst_gt : (const uint8_t*)nullptr, h->list_v, h->labels, h->seq);
            h->last_clean_time = time;
            h->view_dirty = 1;
            h->own_fast_raster = h->own_fast;
        } else {
        if (h->view_frame)
            LAUNCH(h, "clean_view", dim3(h->opt_clean_blocks > 0 ? h->opt_clean_blocks : 2 * LIST_BLOCKS), dim3(MAP_THREADS), k_clean_view, h->d_state, c, time, (float4*)h->pc, (const float4*)h->nr, (float2*)h->tm,
                   (const float4*)h->index_tap, h->list_v);
        else
        LAUNCH(h, "clean_list", dim3(LIST_BLOCKS), dim3(MAP_THREADS), k_clean_list, h->d_state, (const float*)nullptr, c, time, (float4*)h->pc, (const float4*)h->nr, (float2*)h->tm,
               (const float4*)h->index_tap, h->list_b, h->list_c);
        LAUNCH(h, "new_flags_count", dim3(nb_new), dim3(256), k_new_flags_count, h->d_state, (const float*)nullptr, c, time, h->assoc_target, (const float4*)h->meas_pc,
               (const float4*)h->meas_nr, (const float4*)h->index_tap, h->scan_flags, h->scan_block);
        LAUNCH(h, "append_scan", dim3(nb_new), dim3(256), k_append_scan, h->d_state, c, time, time, h->scan_flags, h->scan_block, nb_new, (const float4*)h->meas_pc,
               (const float4*)h->meas_nr, h->meas_col, h->cap, (float4*)h->pc, (float4*)h->nr, (float2*)h->col, (float2*)h->tm, (float4*)h->ic, (float4*)h->votes,
               h->inst_gt_on ? (const uint8_t*)h->d_inst_gt : (const uint8_t*)nullptr, h->view_frame ? h->list_v : (unsigned int*)nullptr, h->labels, h->seq);
        h->last_clean_time = time;
        if (h->view_frame) h->view_dirty = 1;
        if (h->opt_compact_every_frame) ifx_compact_enqueue(h, 0);
        if (h->view_frame && !(h->opt_compact_every_frame || h->last_compact_tick == h->tick)) {   // the lists were built / checked by this frame and nothing renumbered the shard since
            Cam cl = make_cam(h);   // (the store's arrays may have been swapped by a compaction earlier in this phase: taken afresh)
            cl.srank = 0; cl.sn = 1;
            if (h->own_fast) { cl.own_n = 0; cl.raw_slots = 1; }
            LAUNCH(h, "raster_view", dim3(h->opt_view_blocks > 0 ? h->opt_view_blocks : 4 * LIST_BLOCKS), dim3(MAP_THREADS), (k_raster_view<false, false>), h->d_state, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->tm, cl, time, time,
                   LIST_SPLAT | LIST_IDS, h->list_v, h->list_vi, h->key_splat, h->key_ids, h->key_both, 0, lat ? OWN_LAT_DS : 1, CleanArgs{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (const DevState*)h->d_state);   // (the whole id image travels with the splat keys; option own_lazy_ids: the sampled lattice)
            h->own_fast_raster = h->own_fast;
        } else {
            raster_pass(h, nullptr, time, time, LIST_SPLAT | LIST_IDS, h->ids_after, false, 1);
            whole_drawn = true;
        }
        }
        h->view_frame = 0;
        if (h->own_fast_raster) {   // key_both folded into the two renders' images and the slots swapped for creation numbers, pixel by pixel (the slot images of the two renders follow the index pass's)
            LAUNCH(h, "own_merge_translate", dim3(cdiv(h->P, 256)), dim3(256), k_own_merge_translate, h->key_splat, h->key_ids, h->key_both, h->P, (const uint32_t*)h->seq, h->own_slot_img + (size_t)h->P,
                   h->own_slot_img + 2 * (size_t)h->P, OWN_GFL(h, h->gfl_splat));   // (the word: behind the clean and the append)
        } else {
            LAUNCH(h, "merge_both", dim3(cdiv(h->P, 256)), dim3(256), k_merge_both, h->key_splat, h->key_ids, h->key_both, h->P);
            OWN_FIRST_LIVE(h, h->gfl_splat);
        }
        if (lat) {   // [key_splat | the lattice's id keys | word]: packed in front of key_ids
            if (!whole_drawn) LAUNCH(h, "own_ids_pack", dim3(1), dim3(1024), k_own_ids_pack, (const unsigned long long*)h->key_ids, h->w, h->h, (const unsigned long long*)h->gfl_splat, h->key_ids);
            else {   // the per-pass raster drew every pixel: the rest of the image must be empty again before the next frame draws into it
                const size_t L = (size_t)ifx_own_lattice(h);
                if (!h->own_lat_tmp && hipMalloc(&h->own_lat_tmp, (L + 1) * 8) != hipSuccess) { h->err = "hipMalloc (own_lat_tmp)"; return IFX_E_HIP; }
                LAUNCH(h, "own_ids_pack", dim3(1), dim3(1024), k_own_ids_pack, (const unsigned long long*)h->key_ids, h->w, h->h, (const unsigned long long*)h->gfl_splat, h->own_lat_tmp);
                HIPCHK(h, hipMemsetAsync(h->key_ids, 0xFF, (size_t)h->P * 8, h->cur));
                HIPCHK(h, hipMemcpyAsync(h->key_ids, h->own_lat_tmp, (L + 1) * 8, hipMemcpyDeviceToDevice, h->cur));
            }
        }
        break;
    }
    case 104:                                                                                               // ifx_owner_predict_phase: the local raster alone
        h->own_ids_lat = 0;
        raster_pass(h, nullptr, time, time, LIST_SPLAT | LIST_IDS, h->ids_after, false, 1);
        LAUNCH(h, "merge_both", dim3(cdiv(h->P, 256)), dim3(256), k_merge_both, h->key_splat, h->key_ids, h->key_both, h->P);
        OWN_FIRST_LIVE(h, h->gfl_splat);   // (an upload / a deformation may have come in between)
        break;
    case 5: {                                                                                               // owned winners of the prediction; ids_after = creation numbers, from the keys; vote mass of the owned surfels under it | [pred_* | tail]: SUM
        const int lat = h->own_ids_lat;
        if (lat) LAUNCH(h, "own_ids_unpack", dim3(1), dim3(1024), k_own_ids_unpack, h->key_ids, h->w, h->h, h->gfl_splat);
        h->ids_full_valid = !lat; h->ids_sparse_frame = lat; h->ids_view_ok = 0;
        LAUNCH(h, "splat_resolve", g2, b2, k_splat_resolve, h->d_state, (const float*)nullptr, h->key_splat, (const float4*)h->pc, (const float4*)h->nr, (const float2*)h->col,
               (const float2*)h->tm, c, h->rgb, h->depth_filt, (float4*)h->pred_vertex, (float4*)h->pred_normal, (uchar4*)h->pred_image, (uchar4*)h->pred_inst, h->pred_time,
               (float4*)nullptr, (float4*)nullptr, (uchar4*)nullptr, h->key_ids, h->key_both, h->ids_after, (int*)nullptr, FinishFold(), h->pred_conf, lat ? OWN_LAT_DS : 1,
               (in_frame && h->own_fast_raster) ? (const int32_t*)(h->own_slot_img + (size_t)h->P) : (const int32_t*)nullptr);
        if (in_frame) {   // whetherDoSegmentation sums: empty pixels replicated, vote mass by the owners -> the tail of the prediction block
            const int ds = 10, nseg = cdiv(cdiv(h->w, ds) * cdiv(h->h, ds), 256);
            LAUNCH(h, "raster_finish", dim3(1 + nseg), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, 0, h->ids_after, (const float4*)h->votes, h->cap, ds, h->d_list_ctr, ifx_idmap(h),
                   h->pred_tail, (int*)nullptr, h->own_fast_raster ? (const int32_t*)(h->own_slot_img + 2 * (size_t)h->P) : (const int32_t*)nullptr);
        }
        break;
    }
    case 6:                                                                                                 // fill-in and dense flag on the exchanged prediction (replicated); takes the summed vote mass
        LAUNCH(h, "fill_in", g2, b2, k_fill_in, c, (const uint8_t*)h->rgb, (const uint16_t*)h->depth_filt, (float4*)h->pred_vertex, (const float4*)h->pred_normal,
               (const uchar4*)h->pred_image, (float4*)h->fill_vertex, (float4*)h->fill_normal, (uchar4*)h->fill_image, (const float*)h->pred_conf);
        LAUNCH(h, "raster_finish", dim3(1), dim3(256), k_raster_finish, h->d_state, (const uchar4*)h->pred_image, h->w, h->h, 1, h->ids_after, (const float4*)h->votes, h->cap, 10, h->d_list_ctr, ifx_idmap(h),
               (int*)nullptr, in_frame ? h->pred_tail : (int*)nullptr);
        break;
    case 7: break;                                                                                          // (the frame result is published by ifx_owner_frame_phase)
    default: return IFX_E_INVALID;
    }
    return IFX_OK;
}

// ------------------------------------------------------------------ stage API with explicit poses
static int upload_pose(ifx* h, const float* pose16, float** d_pose, float** d_inv)
{
    float buf[32];
    memcpy(buf, pose16, 64);
    pose_inverse(pose16, buf + 16);
    float* slot = h->d_scratch;   // pose + inverse
    HIPCHK(h, hipMemcpyAsync(slot, buf, 128, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *d_pose = slot;
    *d_inv = slot + 16;
    return IFX_OK;
}

extern "C" int ifx_predict_indices(ifx_t* h, const float* pose16, int time)
{
    if (!h || !pose16) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    float *dp, *di;
    int r = upload_pose(h, pose16, &dp, &di);
    if (r) return r;
    index_pass(h, di, time);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}
extern "C" int ifx_combined_predict(ifx_t* h, const float* pose16, int time, int max_time)
{
    if (!h || !pose16) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    float *dp, *di;
    int r = upload_pose(h, pose16, &dp, &di);
    if (r) return r;
    splat_pass(h, di, time, max_time);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}
extern "C" int ifx_fuse(ifx_t* h, const float* pose16, int time, float weighting)
{
    if (!h || !pose16) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    h->seg_counts_valid = 0;
    float *dp, *di;
    int r = upload_pose(h, pose16, &dp, &di);
    if (r) return r;
    fuse_pass(h, dp, weighting, time);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}
extern "C" int ifx_clean(ifx_t* h, const float* pose16, int time)
{
    if (!h || !pose16) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    h->seg_counts_valid = 0;
    float *dp, *di;
    int r = upload_pose(h, pose16, &dp, &di);
    if (r) return r;
    clean_pass(h, di, time);
    if (h->opt_compact_every_frame) ifx_compact_enqueue(h, 0);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}
extern "C" int ifx_render_ids(ifx_t* h, const float* pose16, int mode)
{
    if (!h || !pose16) return IFX_E_INVALID;
    float *dp, *di;
    int r = upload_pose(h, pose16, &dp, &di);
    if (r) return r;
    ids_pass(h, di, mode, h->ids_tmp);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}


// ------------------------------------------------------------------ loop-closure hooks, C-ABI (include/ifx_c_api.h)
extern "C" int ifx_sample_graph_model(ifx_t* h, float* out_xyzt, int max_n)
{
    if (!h || !out_xyzt || max_n <= 0) return IFX_E_INVALID;
    ifx_vlist_reap(h);
    const int cap_s = h->cap / 5000 + 2;
    if (!h->d_sample) HIPCHK(h, hipMalloc(&h->d_sample, (size_t)cap_s * 16));
    if (max_n > cap_s) max_n = cap_s;
    const int n = h->cap;
    LAUNCH(h, "alive_flags", dim3(cdiv(n, 256)), dim3(256), k_alive_flags, h->d_state, (const float2*)h->tm, h->scan_flags, n);
    ifx_scan_exclusive(h, h->scan_flags, n, h->scan_out, &h->d_state->seg_counts[1]);
    LAUNCH(h, "sample_graph", dim3(cdiv(n, 256)), dim3(256), k_sample_graph, h->d_state, h->scan_flags, h->scan_out, (const float4*)h->pc, (const float2*)h->tm,
           (float4*)h->d_sample, max_n);
    int live = 0;
    HIPCHK(h, hipMemcpyAsync(&live, &h->d_state->seg_counts[1], 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int m = (live + 4999) / 5000;
    if (m > max_n) m = max_n;
    if (m > 0) HIPCHK(h, hipMemcpy(out_xyzt, h->d_sample, (size_t)m * 16, hipMemcpyDeviceToHost));
    return m;
}

extern "C" int ifx_loop_closure_constraints(ifx_t* h, float* src3, float* dst3, int32_t* times, int max_n)
{
    if (!h || !src3 || !dst3 || !times) return IFX_E_INVALID;
    if (!h->d_m2m) { h->err = "loop-closure detection is not enabled"; return IFX_E_STATE; }
    if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
    const int rw = h->w / 20, rh = h->h / 20, ns = rw * rh;
    if (!h->d_cons) HIPCHK(h, hipMalloc(&h->d_cons, (size_t)ns * 32));
    LAUNCH(h, "cons_sample", dim3(cdiv(ns, 64)), dim3(64), k_cons_sample, (const DevState*)h->d_state, (const float4*)h->act_vertex, h->old_time, h->w, h->h,
           h->cfg.max_depth_processed, h->d_cons);
    std::vector<float> rec((size_t)ns * 8);
    HIPCHK(h, hipMemcpyAsync(rec.data(), h->d_cons, rec.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int m = 0;
    for (int t = 0; t < ns && m < max_n; t++) {
        const float* r = &rec[(size_t)t * 8];
        if (r[0] == 0.f) continue;
        times[m] = (int32_t)r[1];
        for (int k = 0; k < 3; k++) { src3[m * 3 + k] = r[2 + k]; dst3[m * 3 + k] = r[5 + k]; }
        m++;
    }
    return m;
}

// ---- Ferns (EF/Ferns.cpp) GPU contact no. 1: the four Resize passes of addFrame / findFrame (:95-98, :192-195) -- the fill-in image, vertex and
// normal maps and the instance render resampled to (w/8) x (h/8) (nearest texel of the sample centre) and read back
// fill = 0: img / vert / norm are the fill-in images; fill = 1: they are a raw prediction and the fill-in (fill_rgb/vertex/normal.frag, as in
// k_splat_resolve) is evaluated at the samples only
__global__ void k_fern_resize(const uchar4* __restrict__ img, const float4* __restrict__ vert, const float4* __restrict__ norm, const uchar4* __restrict__ inst, int w, int h,
                              uint8_t* __restrict__ o_img, float4* __restrict__ o_vert, float4* __restrict__ o_norm, uint8_t* __restrict__ o_inst, int fill, Cam c,
                              const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth_filt)
{
    const int rw = w / 8, rh = h / 8, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rw * rh) return;
    const int j = t / rw, i = t - j * rw;
    const int y = (j * h + h / 2) / rh, x = (i * w + w / 2) / rw;
    const int k = y * w + x;
    uchar4 cl = img[k];
    const uchar4 q = inst[k];
    float4 vo = vert[k], no = norm[k];
    if (fill) {
        const float ifx_ = 1.0f / c.fx, ify_ = 1.0f / c.fy;
        if ((int)cl.x + (int)cl.y + (int)cl.z == 0) cl = make_uchar4(rgb[k * 3], rgb[k * 3 + 1], rgb[k * 3 + 2], 255);
        const float zc = (float)depth_filt[k] / 1000.0f;
        if (no.z == 0) {
            v3 vp = v3m(((float)x - c.cx) * zc * ifx_, ((float)y - c.cy) * zc * ify_, zc);
            int xr = clampi(x + 1, 0, w - 1), yd = clampi(y + 1, 0, h - 1);
            float zx = (float)depth_filt[y * w + xr] / 1000.0f, zy = (float)depth_filt[yd * w + x] / 1000.0f;
            v3 vx = v3m(((float)(x + 1) - c.cx) * zx * ifx_, ((float)y - c.cy) * zx * ify_, zx);
            v3 vy = v3m(((float)x - c.cx) * zy * ifx_, ((float)(y + 1) - c.cy) * zy * ify_, zy);
            v3 nn = normalized(cross(vx - vp, vy - vp));
            no = make_float4(nn.x, nn.y, nn.z, 1.f);
        }
        if (vo.z == 0) vo = make_float4(((float)x - c.cx) * zc * ifx_, ((float)y - c.cy) * zc * ify_, zc, 1.f);
    }
    o_img[t * 3] = cl.x; o_img[t * 3 + 1] = cl.y; o_img[t * 3 + 2] = cl.z;
    o_inst[t * 3] = q.x; o_inst[t * 3 + 1] = q.y; o_inst[t * 3 + 2] = q.z;
    o_vert[t] = vo;
    o_norm[t] = no;
}
extern "C" int ifx_fern_frame(ifx_t* h, uint8_t* img_rgb, float* verts4, float* norms4, uint8_t* inst_rgb)
{
    if (!h || !img_rgb || !verts4 || !norms4 || !inst_rgb) return IFX_E_INVALID;
    const int n = (h->w / 8) * (h->h / 8);
    if (!h->d_fern) HIPCHK(h, hipMalloc(&h->d_fern, (size_t)n * (3 + 16 + 16 + 3) + 64));
    uint8_t* base = (uint8_t*)h->d_fern;
    float4* dv = (float4*)base;
    float4* dn = dv + n;
    uint8_t* di = (uint8_t*)(dn + n);
    uint8_t* ds = di + (size_t)n * 3;
    // the last predict(): inside the fern callback the one at the tracked pose (act* images, fill-in evaluated at the samples), else the end-of-frame one
    if (h->in_fern_cb)
        LAUNCH(h, "fern_resize", dim3(cdiv(n, 64)), dim3(64), k_fern_resize, (const uchar4*)h->act_image, (const float4*)h->act_vertex, (const float4*)h->act_normal,
               (const uchar4*)h->act_inst, h->w, h->h, di, dv, dn, ds, 1, make_cam(h), (const uint8_t*)h->rgb, (const uint16_t*)h->depth_filt);
    else
        LAUNCH(h, "fern_resize", dim3(cdiv(n, 64)), dim3(64), k_fern_resize, (const uchar4*)h->fill_image, (const float4*)h->fill_vertex, (const float4*)h->fill_normal,
               (const uchar4*)h->pred_inst, h->w, h->h, di, dv, dn, ds, 0, make_cam(h), (const uint8_t*)h->rgb, (const uint16_t*)h->depth_filt);
    HIPCHK(h, hipMemcpyAsync(verts4, dv, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(norms4, dn, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(img_rgb, di, (size_t)n * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(inst_rgb, ds, (size_t)n * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return n;
}

// The same read-back without stalling the caller: enqueued behind the frame, fetched when the host next has to wait for the stream anyway (the
// fern callback of the following frame) -- Ferns::addFrame only has to be done before the next findFrame.
extern "C" int ifx_fern_frame_async(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    if (h->in_fern_cb) { h->err = "ifx_fern_frame_async reads the end-of-frame prediction: not inside the fern callback"; return IFX_E_STATE; }
    const int n = (h->w / 8) * (h->h / 8);
    const size_t bytes = (size_t)n * (3 + 16 + 16 + 3);
    if (!h->d_fern) HIPCHK(h, hipMalloc(&h->d_fern, bytes + 64));
    if (!h->h_fern) { HIPCHK(h, hipHostMalloc((void**)&h->h_fern, bytes + 64)); HIPCHK(h, hipEventCreateWithFlags(&h->ev_fern, hipEventDisableTiming)); }
    uint8_t* base = (uint8_t*)h->d_fern;
    float4* dv = (float4*)base;
    float4* dn = dv + n;
    uint8_t* di = (uint8_t*)(dn + n);
    uint8_t* ds = di + (size_t)n * 3;
    LAUNCH(h, "fern_resize", dim3(cdiv(n, 64)), dim3(64), k_fern_resize, (const uchar4*)h->fill_image, (const float4*)h->fill_vertex, (const float4*)h->fill_normal,
           (const uchar4*)h->pred_inst, h->w, h->h, di, dv, dn, ds, 0, make_cam(h), (const uint8_t*)h->rgb, (const uint16_t*)h->depth_filt);
    HIPCHK(h, hipMemcpyAsync(h->h_fern, h->d_fern, bytes, hipMemcpyDeviceToHost, h->stream));   // one copy: the staging buffer has the layout of the pinned one
    HIPCHK(h, hipEventRecord(h->ev_fern, h->stream));
    h->fern_pending = 1;
    return n;
}
extern "C" int ifx_fern_frame_fetch(ifx_t* h, uint8_t* img_rgb, float* verts4, float* norms4, uint8_t* inst_rgb)
{
    if (!h || !img_rgb || !verts4 || !norms4 || !inst_rgb) return IFX_E_INVALID;
    if (!h->fern_pending) { h->err = "ifx_fern_frame_fetch: nothing was requested (ifx_fern_frame_async)"; return IFX_E_STATE; }
    HIPCHK(h, hipEventSynchronize(h->ev_fern));
    h->fern_pending = 0;
    const int n = (h->w / 8) * (h->h / 8);
    const uint8_t* p = h->h_fern;
    memcpy(verts4, p, (size_t)n * 16); p += (size_t)n * 16;
    memcpy(norms4, p, (size_t)n * 16); p += (size_t)n * 16;
    memcpy(img_rgb, p, (size_t)n * 3); p += (size_t)n * 3;
    memcpy(inst_rgb, p, (size_t)n * 3);
    return n;
}

extern "C" int ifx_set_deformation(ifx_t* h, const float* graph16, int n_nodes, int is_fern)
{
    if (!h || n_nodes < 0 || (n_nodes > 0 && !graph16)) return IFX_E_INVALID;
    if (n_nodes == 0) { h->graph_nodes = 0; return IFX_OK; }
    if (n_nodes < 4 || n_nodes >= 1024) { h->err = "a deformation graph has 4 .. 1023 nodes (GlobalModel::MAX_NODES, 4 neighbours per surfel)"; return IFX_E_INVALID; }
    for (int i = 1; i < n_nodes; i++)
        if (graph16[i * 16 + 15] < graph16[(i - 1) * 16 + 15]) { h->err = "deformation graph nodes must be sorted by time"; return IFX_E_INVALID; }
    if (!is_fern) {   // the time-stamp refresh samples the re-rendered INACTIVE depth: needs the old* images
        int r = ifx_tracker_alloc_m2m(h);
        if (r) return r;
    }
    if (h->stream_c) { HIPCHK(h, hipStreamSynchronize(h->stream_c)); h->lc_pending = 0; }   // a model-to-model run still reading the old* images
    if (!h->d_graph) HIPCHK(h, hipMalloc(&h->d_graph, 1024 * 64));
    HIPCHK(h, hipMemcpyAsync(h->d_graph, graph16, (size_t)n_nodes * 64, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));   // graph16 is the caller's
    h->graph_nodes = n_nodes; h->graph_is_fern = is_fern ? 1 : 0;
    ifx_drop_tracked(h);
    return IFX_OK;
}

struct Pose16 { float m[16]; };
__global__ void k_adopt_pose(DevState* st, Pose16 p)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 16; k++) st->pose[k] = p.m[k];
    pose_inverse(st->pose, st->pose_inv);
    st->vl_valid = 0;
}
extern "C" int ifx_adopt_pose(ifx_t* h, const float* pose16)
{
    if (!h || !pose16) return IFX_E_INVALID;
    if (!h->in_fern_cb) { h->err = "ifx_adopt_pose: only inside the fern callback (currPose = recoveryPose, EF/ElasticFusion.cpp:482,504)"; return IFX_E_STATE; }
    Pose16 p;
    memcpy(p.m, pose16, sizeof(p.m));
    ifx_drop_tracked(h);
    h->view_block = 1;
    LAUNCH(h, "adopt_pose", dim3(1), dim3(64), k_adopt_pose, h->d_state, p);
    return IFX_OK;
}

extern "C" int ifx_adopt_estimated_pose(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    if (!h->d_m2m) { h->err = "loop-closure detection is not enabled"; return IFX_E_STATE; }
    ifx_drop_tracked(h);
    h->view_block = 1;
    LAUNCH(h, "adopt_est_pose", dim3(1), dim3(64), k_adopt_est_pose, h->d_state);
    return IFX_OK;
}
