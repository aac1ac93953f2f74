// ifx_knn.hip -- k-nearest-neighbour smoothing of the instance colours.
//
//   ifx_knn_vote  <-  InstanceFusion::flannKnnVoteSurfelMap   IF/Core/InstanceFusion.cpp:1070-1163
//                     mapKnnVoteColourKernel                  IF/Core/InstanceFusionCuda.cu:1237-1340
//
// The reference builds a FLANN kd-tree over all surfel positions (REF/deps/flann-1.8.4, KDTreeCuda3dIndex,
// exact search, k = 10, the query set is the indexed set, so every surfel is its own first neighbour) and lets
// each surfel take the colour of the instance that most of its 10 neighbours are labelled with
// (bestIDInEachSurfel >= 0 only; first maximum; nothing changes when no neighbour is labelled).
//
// Here: exact k-NN on a uniform grid instead of a kd-tree.  Cells = bounding box of the live surfels cut into
// <= 256 cells along its longest side; counting sort by cell (histogram, exclusive scan, scatter); a query walks
// the cell shells around its own cell and stops as soon as its 10th candidate is closer than the inner border
// of the next shell, so the result is the exact k-NN set.  Ties in distance go to the lower map index (FLANN
// leaves that order unspecified); the oracle uses the same rule with a brute-force search.
#include "ifx_ctx.h"
#include "ifx_dev.h"
#include <algorithm>

#define DEAD_TIME (-1.0e9f)   // tombstone marker in times.y (ifx_map.hip)

namespace {

constexpr int KNN = 10;
constexpr int GRID_MAX = 256;

struct KnnGrid {        // device-resident, filled by k_knn_grid
    int lo_bits[3], hi_bits[3];   // ordered-int encodings of the bounding box (atomicMin / atomicMax)
    float lo[3], cell, inv_cell;
    int dim[3];
};

__device__ __forceinline__ int ord(float f) { int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float unord(int b) { return __int_as_float(b >= 0 ? b : b ^ 0x7FFFFFFF); }

__device__ __forceinline__ bool live(const DevState* st, const float2* tm, int i) { return i < st->count && tm[i].y > DEAD_TIME; }

__global__ void k_knn_bounds_init(KnnGrid* g)
{
    if (threadIdx.x < 3) { g->lo_bits[threadIdx.x] = 0x7FFFFFFF; g->hi_bits[threadIdx.x] = (int)0x80000000; }
}
__global__ void k_knn_bounds(const DevState* __restrict__ st, const float4* __restrict__ pc, const float2* __restrict__ tm, KnnGrid* g)
{
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        if (!(tm[i].y > DEAD_TIME)) continue;
        float4 p = pc[i];
        int v[3] = {ord(p.x), ord(p.y), ord(p.z)};
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], v[k]); hi[k] = max(hi[k], v[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o; o >>= 1) { lo[k] = min(lo[k], __shfl_xor(lo[k], o)); hi[k] = max(hi[k], __shfl_xor(hi[k], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&g->lo_bits[k], lo[k]); atomicMax(&g->hi_bits[k], hi[k]); }
    }
}
__global__ void k_knn_grid(KnnGrid* g)
{
    if (threadIdx.x != 0) return;
    float ext = 0.f;
    for (int k = 0; k < 3; k++) { g->lo[k] = unord(g->lo_bits[k]); ext = fmaxf(ext, unord(g->hi_bits[k]) - g->lo[k]); }
    if (!(ext > 0.f)) ext = 1.0f;
    g->cell = ext / (float)(GRID_MAX - 1);
    g->inv_cell = 1.0f / g->cell;
    for (int k = 0; k < 3; k++) {
        int d = (int)((unord(g->hi_bits[k]) - g->lo[k]) * g->inv_cell) + 1;
        g->dim[k] = min(max(d, 1), GRID_MAX);
    }
}
__device__ __forceinline__ void cell_of(const KnnGrid* g, float4 p, int* c)
{
    const float q[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = min(max((int)((q[k] - g->lo[k]) * g->inv_cell), 0), g->dim[k] - 1);
}
__global__ void k_knn_count(const DevState* __restrict__ st, const float4* __restrict__ pc, const float2* __restrict__ tm, const KnnGrid* __restrict__ g, int* __restrict__ cell_id,
                            int* __restrict__ counts)
{
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        int id = -1;
        if (tm[i].y > DEAD_TIME) {
            int c[3];
            cell_of(g, pc[i], c);
            id = (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
            atomicAdd(&counts[id], 1);
        }
        cell_id[i] = id;
    }
}
// scatter into cell order: slot index AND position (slot in the w lane), so that the search reads its candidates with one
// contiguous 16-B load each and neighbouring threads (= neighbouring cells) share cache lines
__global__ void k_knn_scatter(const DevState* __restrict__ st, const float4* __restrict__ pc, const int* __restrict__ cell_id, const int* __restrict__ start, int* __restrict__ fill,
                              float4* __restrict__ sorted)
{
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        int id = cell_id[i];
        if (id < 0) continue;
        float4 p = pc[i];
        p.w = __int_as_float(i);
        sorted[start[id] + atomicAdd(&fill[id], 1)] = p;
    }
}

// one query per thread: exact k-NN by growing cell shells, then the majority vote of the labelled neighbours
__global__ void __launch_bounds__(128) k_knn_vote(const DevState* __restrict__ st, const KnnGrid* __restrict__ g, const int* __restrict__ start, const int* __restrict__ counts,
                                                  const float4* __restrict__ sorted, const int* __restrict__ total, const int32_t* __restrict__ labels,
                                                  const float* __restrict__ inst_color, float2* __restrict__ col, int32_t* __restrict__ nbr_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;   // queries in cell order
    if (t >= *total) return;
    const float4 q = sorted[t];
    const int i = __float_as_int(q.w);
    int c[3];
    cell_of(g, q, c);
    const int dx = g->dim[0], dy = g->dim[1], dz = g->dim[2];
    float bd[KNN];
    int bi[KNN];
#pragma unroll
    for (int k = 0; k < KNN; k++) { bd[k] = INFINITY; bi[k] = 0x7FFFFFFF; }
    const int rmax = max(max(dx, dy), dz);
    for (int r = 0; r <= rmax; r++) {
        for (int z = max(c[2] - r, 0); z <= min(c[2] + r, dz - 1); z++)
            for (int y = max(c[1] - r, 0); y <= min(c[1] + r, dy - 1); y++) {
                const bool face = abs(z - c[2]) == r || abs(y - c[1]) == r;   // on a z / y face the whole x run is new, else only its two ends
                for (int x = max(c[0] - r, 0); x <= min(c[0] + r, dx - 1); x++) {
                    if (!face && abs(x - c[0]) != r) continue;
                    const int id = (z * dy + y) * dx + x, s0 = start[id], s1 = s0 + counts[id];
                    for (int s = s0; s < s1; s++) {
                        const float4 p = sorted[s];
                        const int j = __float_as_int(p.w);
                        const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
                        float d = (ex * ex + ey * ey) + ez * ez;
                        if (!(d < bd[KNN - 1] || (d == bd[KNN - 1] && j < bi[KNN - 1]))) continue;
                        int jj = j;   // insertion into the sorted top-10 by (distance, index); fully unrolled -> registers
#pragma unroll
                        for (int k = 0; k < KNN; k++) {
                            const bool before = d < bd[k] || (d == bd[k] && jj < bi[k]);
                            const float td = before ? bd[k] : d;
                            const int ti = before ? bi[k] : jj;
                            bd[k] = before ? d : bd[k];
                            bi[k] = before ? jj : bi[k];
                            d = td; jj = ti;
                        }
                    }
                }
            }
        // every point not visited yet lies outside the cube of (2r+1)^3 cells around the query's cell: farther than r cells
        const float safe = (float)r * g->cell * 0.99f;
        if (bi[KNN - 1] != 0x7FFFFFFF && bd[KNN - 1] <= safe * safe) break;
    }
    int lab[KNN];
#pragma unroll
    for (int k = 0; k < KNN; k++) {
        lab[k] = (bi[k] != 0x7FFFFFFF) ? labels[bi[k]] : -1;
        if (nbr_out) nbr_out[(size_t)i * KNN + k] = (bi[k] != 0x7FFFFFFF) ? bi[k] : -1;
    }
    int best = -1, bestCount = 0;   // first maximum over instance ids 0..95 == highest count, lowest id on ties
#pragma unroll
    for (int k = 0; k < KNN; k++) {
        if (lab[k] < 0) continue;
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < KNN; m++) cnt += (lab[m] == lab[k]);
        if (cnt > bestCount || (cnt == bestCount && lab[k] < best)) { bestCount = cnt; best = lab[k]; }
    }
    if (bestCount > 0) col[i].y = inst_color[best];
}

// ---- the same search over an EXPLICIT point set: the smoothing on a spatially sharded map (the ranks' exports, all-gathered).
// pts[i] = (x, y, z, creation number); x = NaN marks a dead slot.  Ties in distance go to the lower creation number -- the order of the
// unsharded map's indices (compaction preserves it), so the neighbour sets are the ones ifx_knn_vote finds on the unsharded map.
__global__ void k_knnx_export(const DevState* __restrict__ st, const float4* __restrict__ pc, const float2* __restrict__ tm, const uint32_t* __restrict__ seq,
                              const int32_t* __restrict__ labels, float4* __restrict__ pts, int32_t* __restrict__ lab)
{
    const int n = st->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        float4 p = pc[i];
        p.w = __uint_as_float(seq[i]);
        if (!(tm[i].y > DEAD_TIME)) p.x = __int_as_float(0x7FC00000);
        pts[i] = p;
        lab[i] = labels[i];
    }
}
__global__ void k_knnx_bounds(const float4* __restrict__ pts, int n, KnnGrid* g)
{
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const float4 p = pts[i];
        if (p.x != p.x) continue;
        int v[3] = {ord(p.x), ord(p.y), ord(p.z)};
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], v[k]); hi[k] = max(hi[k], v[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o; o >>= 1) { lo[k] = min(lo[k], __shfl_xor(lo[k], o)); hi[k] = max(hi[k], __shfl_xor(hi[k], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&g->lo_bits[k], lo[k]); atomicMax(&g->hi_bits[k], hi[k]); }
    }
}
__global__ void k_knnx_count(const float4* __restrict__ pts, int n, const KnnGrid* __restrict__ g, int* __restrict__ cell_id, int* __restrict__ counts)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        int id = -1;
        const float4 p = pts[i];
        if (!(p.x != p.x)) {
            int c[3];
            cell_of(g, p, c);
            id = (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
            atomicAdd(&counts[id], 1);
        }
        cell_id[i] = id;
    }
}
__global__ void k_knnx_scatter(const float4* __restrict__ pts, const int32_t* __restrict__ lab, int n, const int* __restrict__ cell_id, const int* __restrict__ start,
                               int* __restrict__ fill, float4* __restrict__ sorted, int2* __restrict__ aux)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const int id = cell_id[i];
        if (id < 0) continue;
        const int at = start[id] + atomicAdd(&fill[id], 1);
        sorted[at] = pts[i];
        aux[at] = make_int2(i, lab[i]);   // index in the gathered set, bestIDInEachSurfel
    }
}
__global__ void __launch_bounds__(128) k_knnx_vote(const KnnGrid* __restrict__ g, const int* __restrict__ start, const int* __restrict__ counts, const float4* __restrict__ sorted,
                                                   const int2* __restrict__ aux, const int* __restrict__ total, int own_lo, int own_n, const float* __restrict__ inst_color,
                                                   float2* __restrict__ col)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;   // queries in cell order; only the points this rank exported are queries
    if (t >= *total) return;
    const int gi = aux[t].x;
    if (gi < own_lo || gi >= own_lo + own_n) return;
    const float4 q = sorted[t];
    int c[3];
    cell_of(g, q, c);
    const int dx = g->dim[0], dy = g->dim[1], dz = g->dim[2];
    float bd[KNN];
    unsigned int bs[KNN];   // creation number (tie-break)
    int bl[KNN];            // label
#pragma unroll
    for (int k = 0; k < KNN; k++) { bd[k] = INFINITY; bs[k] = 0xFFFFFFFFu; bl[k] = -1; }
    const int rmax = max(max(dx, dy), dz);
    for (int r = 0; r <= rmax; r++) {
        for (int z = max(c[2] - r, 0); z <= min(c[2] + r, dz - 1); z++)
            for (int y = max(c[1] - r, 0); y <= min(c[1] + r, dy - 1); y++) {
                const bool face = abs(z - c[2]) == r || abs(y - c[1]) == r;
                for (int x = max(c[0] - r, 0); x <= min(c[0] + r, dx - 1); x++) {
                    if (!face && abs(x - c[0]) != r) continue;
                    const int id = (z * dy + y) * dx + x, s0 = start[id], s1 = s0 + counts[id];
                    for (int s = s0; s < s1; s++) {
                        const float4 p = sorted[s];
                        const unsigned int j = __float_as_uint(p.w);
                        const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
                        float d = (ex * ex + ey * ey) + ez * ez;
                        if (!(d < bd[KNN - 1] || (d == bd[KNN - 1] && j < bs[KNN - 1]))) continue;
                        unsigned int jj = j;
                        int ll = aux[s].y;
#pragma unroll
                        for (int k = 0; k < KNN; k++) {
                            const bool before = d < bd[k] || (d == bd[k] && jj < bs[k]);
                            const float td = before ? bd[k] : d;
                            const unsigned int ts = before ? bs[k] : jj;
                            const int tl = before ? bl[k] : ll;
                            bd[k] = before ? d : bd[k];
                            bs[k] = before ? jj : bs[k];
                            bl[k] = before ? ll : bl[k];
                            d = td; jj = ts; ll = tl;
                        }
                    }
                }
            }
        const float safe = (float)r * g->cell * 0.99f;
        if (bs[KNN - 1] != 0xFFFFFFFFu && bd[KNN - 1] <= safe * safe) break;
    }
    int best = -1, bestCount = 0;
#pragma unroll
    for (int k = 0; k < KNN; k++) {
        if (bs[k] == 0xFFFFFFFFu || bl[k] < 0) continue;
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < KNN; m++) cnt += (bs[m] != 0xFFFFFFFFu && bl[m] == bl[k]);
        if (cnt > bestCount || (cnt == bestCount && bl[k] < best)) { bestCount = cnt; best = bl[k]; }
    }
    if (bestCount > 0) col[gi - own_lo].y = inst_color[best];   // (the export is in slot order: export index = slot)
}

}  // namespace

void ifx_knn_free(ifx* h)
{
    if (h->d_knn) hipFree(h->d_knn);
    h->d_knn = nullptr; h->knn_cap = 0;
}
void ifx_knn_free_all(ifx* h)
{
    ifx_knn_free(h);
    if (h->d_kexp) hipFree(h->d_kexp);
    h->d_kexp = nullptr;
}

int ifx_scan_exclusive(ifx* h, const int* d_flags, int n, int* d_out, int* d_total);

// flags bit0 of ifx_process_segmentation; nbr_out (device, [slots][10], optional) receives the neighbour slots
int ifx_knn_vote(ifx* h, int32_t* d_nbr_out)
{
    const size_t cells = (size_t)GRID_MAX * GRID_MAX * GRID_MAX, cap = ((size_t)h->cap + 3) & ~(size_t)3;
    const size_t need = 64 + cells * 3 + cap * 5 + 16;   // grid header, counts / starts / fill, cell ids, positions in cell order (float4)
    if (need > h->knn_cap) {
        ifx_knn_free(h);
        HIPCHK(h, hipMalloc(&h->d_knn, need * 4));
        h->knn_cap = need;
    }
    KnnGrid* g = (KnnGrid*)h->d_knn;
    int* counts = h->d_knn + 64;
    int* starts = counts + cells;
    int* fill = starts + cells;
    int* cell_id = fill + cells;
    float4* sorted = (float4*)(cell_id + cap);   // 16-B aligned: header 64 ints, cells and cap are multiples of 4 after rounding below
    int* total = (int*)(sorted + cap);
    HIPCHK(h, hipMemsetAsync(counts, 0, cells * 4, h->cur));
    HIPCHK(h, hipMemsetAsync(fill, 0, cells * 4, h->cur));
    LAUNCH(h, "knn_bounds_init", dim3(1), dim3(64), k_knn_bounds_init, g);
    LAUNCH(h, "knn_bounds", dim3(1024), dim3(256), k_knn_bounds, h->d_state, (const float4*)h->pc, (const float2*)h->tm, g);
    LAUNCH(h, "knn_grid", dim3(1), dim3(64), k_knn_grid, g);
    LAUNCH(h, "knn_count", dim3(2048), dim3(256), k_knn_count, h->d_state, (const float4*)h->pc, (const float2*)h->tm, g, cell_id, counts);
    int r = ifx_scan_exclusive(h, counts, (int)cells, starts, total);
    if (r) return r;
    LAUNCH(h, "knn_scatter", dim3(2048), dim3(256), k_knn_scatter, h->d_state, (const float4*)h->pc, cell_id, starts, fill, sorted);
    LAUNCH(h, "knn_vote", dim3(cdiv(h->cap, 128)), dim3(128), k_knn_vote, h->d_state, g, starts, counts, sorted, total, h->labels, h->d_inst_color, (float2*)h->col, d_nbr_out);
    return IFX_OK;
}

// stage entry for tests: runs the smoothing on the current map / labels and returns the neighbour slots of the first
// `max_n` slots (host, [max_n][10], -1 = none / dead slot)
extern "C" int ifx_knn_vote_colour(ifx_t* h, int32_t* nbr_out, int max_n)
{
    if (h) ifx_vlist_reap(h);   // whole-map consumer: nothing outside the cached view list may outlive the age rule (ifx_map.hip "View list")
    if (!h || max_n < 0) return IFX_E_INVALID;
    int32_t* d_nbr = nullptr;
    if (nbr_out && max_n > 0) {
        HIPCHK(h, hipMalloc(&d_nbr, (size_t)h->cap * KNN * 4));
        HIPCHK(h, hipMemsetAsync(d_nbr, 0xFF, (size_t)h->cap * KNN * 4, h->cur));
    }
    int r = ifx_knn_vote(h, d_nbr);
    if (!r && d_nbr) r = hipMemcpyAsync(nbr_out, d_nbr, (size_t)std::min(max_n, h->cap) * KNN * 4, hipMemcpyDeviceToHost, h->cur) == hipSuccess ? IFX_OK : IFX_E_HIP;
    hipStreamSynchronize(h->cur);
    if (d_nbr) hipFree(d_nbr);
    return r;
}

// ---- flannKnnVoteSurfelMap on a spatially sharded map: export, all-gather by the caller, vote
extern "C" int ifx_owner_knn_export(ifx_t* h, void** d_points, void** d_labels, int* n)
{
    if (!h || !d_points || !d_labels || !n) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_knn_export: the handle was not created with n_ranks > 1"; return IFX_E_STATE; }
    ifx_vlist_reap(h);
    HIPCHK(h, hipStreamSynchronize(h->cur));
    int cnt = 0;
    HIPCHK(h, hipMemcpy(&cnt, &h->d_state->count, sizeof(int), hipMemcpyDeviceToHost));
    if (!h->d_kexp) HIPCHK(h, hipMalloc(&h->d_kexp, (size_t)h->cap * 20));
    float4* pts = (float4*)h->d_kexp;
    int32_t* lab = (int32_t*)(pts + h->cap);
    if (cnt > 0) LAUNCH(h, "knnx_export", dim3(1024), dim3(256), k_knnx_export, h->d_state, (const float4*)h->pc, (const float2*)h->tm, (const uint32_t*)h->seq, h->labels, pts, lab);
    HIPCHK(h, hipStreamSynchronize(h->cur));
    *d_points = pts; *d_labels = lab; *n = cnt;
    return IFX_OK;
}
extern "C" int ifx_owner_knn_vote(ifx_t* h, const void* d_all_points, const void* d_all_labels, int n_all, int own_offset)
{
    if (!h || !d_all_points || !d_all_labels || n_all < 0 || own_offset < 0) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_knn_vote: the handle was not created with n_ranks > 1"; return IFX_E_STATE; }
    int own_n = 0;
    HIPCHK(h, hipMemcpy(&own_n, &h->d_state->count, sizeof(int), hipMemcpyDeviceToHost));
    if (own_offset + own_n > n_all) { h->err = "ifx_owner_knn_vote: this rank's export does not fit the gathered set"; return IFX_E_INVALID; }
    if (n_all == 0) return IFX_OK;
    const size_t cells = (size_t)GRID_MAX * GRID_MAX * GRID_MAX, capx = ((size_t)n_all + 3) & ~(size_t)3;
    const size_t need = 64 + cells * 3 + capx * 7 + 16;   // grid header, counts / starts / fill, cell ids, positions (float4) and (index, label) pairs in cell order
    if (need > h->knn_cap) {
        ifx_knn_free(h);
        HIPCHK(h, hipMalloc(&h->d_knn, need * 4));
        h->knn_cap = need;
    }
    KnnGrid* g = (KnnGrid*)h->d_knn;
    int* counts = h->d_knn + 64;
    int* starts = counts + cells;
    int* fill = starts + cells;
    int* cell_id = fill + cells;
    float4* sorted = (float4*)(cell_id + capx);
    int2* aux = (int2*)(sorted + capx);
    int* total = (int*)(aux + capx);
    const float4* pts = (const float4*)d_all_points;
    const int32_t* lab = (const int32_t*)d_all_labels;
    HIPCHK(h, hipMemsetAsync(counts, 0, cells * 4, h->cur));
    HIPCHK(h, hipMemsetAsync(fill, 0, cells * 4, h->cur));
    LAUNCH(h, "knn_bounds_init", dim3(1), dim3(64), k_knn_bounds_init, g);
    LAUNCH(h, "knnx_bounds", dim3(1024), dim3(256), k_knnx_bounds, pts, n_all, g);
    LAUNCH(h, "knn_grid", dim3(1), dim3(64), k_knn_grid, g);
    LAUNCH(h, "knnx_count", dim3(2048), dim3(256), k_knnx_count, pts, n_all, g, cell_id, counts);
    int r = ifx_scan_exclusive(h, counts, (int)cells, starts, total);
    if (r) return r;
    LAUNCH(h, "knnx_scatter", dim3(2048), dim3(256), k_knnx_scatter, pts, lab, n_all, cell_id, starts, fill, sorted, aux);
    LAUNCH(h, "knnx_vote", dim3(cdiv(n_all, 128)), dim3(128), k_knnx_vote, g, starts, counts, sorted, aux, total, own_offset, own_n, h->d_inst_color, (float2*)h->col);
    HIPCHK(h, hipStreamSynchronize(h->cur));
    return IFX_OK;
}
