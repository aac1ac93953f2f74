// ifx_api.hip -- handle life cycle, frame orchestration and the data-movement half of the C-ABI.
#include <climits>
#include "ifx_ctx.h"
#include <string.h>
#include <stdio.h>
#include <algorithm>

int ifx_tracker_external_pose(ifx* h, const float* d_pose16, float weight_mult);
int ifx_tracker_set_weight(ifx* h, float weight_mult);
int ifx_compact_enqueue(ifx* h, int refresh_ids);

void hs_invalidate_view(ifx* h);
static std::string g_err;
extern "C" const char* ifx_global_error(void) { return g_err.c_str(); }
extern "C" const char* ifx_last_error(ifx_t* h) { return h ? h->err.c_str() : g_err.c_str(); }

// ------------------------------------------------------------------ kernel timing
hipEvent_t ifx_event_get(ifx* h)
{
    if (!h->event_pool.empty()) { hipEvent_t e = h->event_pool.back(); h->event_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void ifx_ktime_begin(ifx* h, const char* name, hipEvent_t* a)
{
    (void)name;
    *a = ifx_event_get(h);
    hipEventRecord(*a, h->cur);
}
void ifx_ktime_end(ifx* h, const char* name, hipEvent_t a)
{
    hipEvent_t b = ifx_event_get(h);
    hipEventRecord(b, h->cur);
    auto it = h->kname_id.find(name);
    int id;
    if (it == h->kname_id.end()) { id = (int)h->knames.size(); h->kname_id[name] = id; h->knames.push_back(name); h->ktimes.push_back(KernelTiming()); }
    else id = it->second;
    PendingEvent pe; pe.name_id = id; pe.a = a; pe.b = b;
    h->kpending.push_back(pe);
}
static void ktime_flush(ifx* h)
{
    for (auto& pe : h->kpending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            h->ktimes[pe.name_id].total_ms += ms; h->ktimes[pe.name_id].launches++;
            // k_cull_frame decides ON THE DEVICE whether it scans the store or returns at its first instruction (the cached view lists are still valid): the launches that
            // did scan are also kept under "cull_frame@scan" -- told apart by their duration (a launch that returns at once takes ~4 us, a scan of a million slots ~20)
            if (ms > 0.02f && h->knames[pe.name_id] == "cull_frame") {
                auto it = h->kname_id.find("cull_frame@scan");
                int id;
                if (it == h->kname_id.end()) { id = (int)h->knames.size(); h->kname_id["cull_frame@scan"] = id; h->knames.push_back("cull_frame@scan"); h->ktimes.push_back(KernelTiming()); }
                else id = it->second;
                h->ktimes[id].total_ms += ms; h->ktimes[id].launches++;
            }
        }
        h->event_pool.push_back(pe.a);
        h->event_pool.push_back(pe.b);
    }
    h->kpending.clear();
}
static void stage_flush(ifx* h)
{
    for (auto& sp : h->stage_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.second.first, sp.second.second) == hipSuccess) h->stage_ms[sp.first] += ms;
        h->event_pool.push_back(sp.second.first);
        h->event_pool.push_back(sp.second.second);
    }
    h->stage_pending.clear();
}
struct StageTimer {
    ifx* h; int id; hipEvent_t a;
    StageTimer(ifx* h_, int id_) : h(h_), id(id_), a(nullptr) { if (h->opt_stage_timing) { a = ifx_event_get(h); hipEventRecord(a, h->cur); } }
    ~StageTimer() { if (a) { hipEvent_t b = ifx_event_get(h); hipEventRecord(b, h->cur); h->stage_pending.push_back({id, {a, b}}); } }
};

// ------------------------------------------------------------------ create / destroy
#define ALLOC(ptr, bytes)                                                                          \
    do {                                                                                           \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                        \
        if (e_ != hipSuccess) { g_err = std::string("hipMalloc " #ptr ": ") + hipGetErrorString(e_); ifx_destroy(h); return IFX_E_HIP; } \
    } while (0)

extern "C" int ifx_set_option(ifx_t* h, const char* name, int value);
// the prediction block [pred_vertex | pred_conf | pred_normal | pred_image | pred_inst | pred_time | tail] at `base` (creation; a camera switch that swaps blocks)
static void ifx_pred_rebind(ifx* h, float* base)
{
    const size_t P = (size_t)h->P, conf_bytes = ((P * 4 + 15) / 16) * 16;
    h->pred_vertex = base; h->pred_conf = (float*)((uint8_t*)h->pred_vertex + P * 16); h->pred_normal = (float*)((uint8_t*)h->pred_conf + conf_bytes);
    h->pred_image = (uint8_t*)(h->pred_normal + 4 * P); h->pred_inst = h->pred_image + 4 * P;
    h->pred_time = (uint16_t*)(h->pred_inst + 4 * P); h->pred_tail = (int*)((uint8_t*)h->pred_vertex + h->pred_bytes - 16);
}
extern "C" int ifx_create(const ifx_config* cfg, ifx_t** out)
{
    if (!cfg || !out) { g_err = "null argument"; return IFX_E_INVALID; }
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width % 4 || cfg->height % 4 || cfg->max_surfels <= 0) {
        g_err = "width/height must be positive multiples of 4 (three pyramid levels) and max_surfels > 0";
        return IFX_E_INVALID;
    }
    if (cfg->n_ranks < -1 || cfg->n_ranks > 64 || (cfg->n_ranks > 1 && (cfg->rank < 0 || cfg->rank >= cfg->n_ranks)) || (cfg->n_ranks <= 1 && cfg->rank != 0)) {
        g_err = "n_ranks / rank: a spatially sharded map has 2..64 ranks and 0 <= rank < n_ranks (n_ranks 0 or 1: unsharded, rank 0; -1: the sharded path with a single rank)";
        return IFX_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_err = "no HIP device available: libifx.so has no CPU fallback"; return IFX_E_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_err = "device ordinal out of range"; return IFX_E_INVALID; }
    if (hipSetDevice(cfg->device) != hipSuccess) { g_err = "hipSetDevice failed"; return IFX_E_HIP; }
    ifx* h = new ifx();
    h->cfg = *cfg;
    h->w = cfg->width; h->h = cfg->height; h->P = h->w * h->h; h->cap = cfg->max_surfels;
    h->own = cfg->n_ranks > 1 || cfg->n_ranks == -1;   // -1: a world of one on the sharded path (creation-number ids, owner filter, exchange points) -- RCCL tests and `bench.py --sharded --gpus 1`
    h->own_g = cfg->n_ranks > 1 ? cfg->n_ranks : 1;
    size_t P = (size_t)h->P, C = (size_t)h->cap;
    {   // the main stream carries the latency-bound chain of the frame (tracker, map passes): highest priority; the side stream's image-only work
        // (bilateral filter, frame pyramids, SO(3)) fills in around it: lowest (IFX_STREAM_PRIORITIES=0 in the environment: both default)
        int lo = 0, hi = 0;
        const char* e = getenv("IFX_STREAM_PRIORITIES");
        const bool prio = !(e && e[0] == '0') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi;
        hipError_t r1 = prio ? hipStreamCreateWithPriority(&h->stream, hipStreamDefault, hi) : hipStreamCreate(&h->stream);
        hipError_t r2 = prio ? hipStreamCreateWithPriority(&h->stream_b, hipStreamDefault, lo) : hipStreamCreate(&h->stream_b);
        if (r1 != hipSuccess || r2 != hipSuccess) { g_err = "hipStreamCreate failed"; delete h; return IFX_E_HIP; }
        hipEventCreateWithFlags(&h->ev_gate, hipEventDisableTiming);
        // ONE more stream for everything that runs beside the main stream now and then (the model-to-model tracker of the loop-closure detection, a segmentation call):
        // the runtime multiplexes streams onto four hardware queues, the process's default stream has one, and a fifth stream shares a queue with another -- whose
        // barrier packets then hold it up (a stream of its own for the segmentation call cost closeLoops = true two thirds of its frame rate: 1031 -> 348 frames/s)
        hipError_t r3 = prio ? hipStreamCreateWithPriority(&h->stream_c, hipStreamDefault, hi) : hipStreamCreate(&h->stream_c);
        if (r3 != hipSuccess) { g_err = "hipStreamCreate failed"; ifx_destroy(h); return IFX_E_HIP; }
    }
    h->cur = h->stream;
    ALLOC(h->d_state, sizeof(DevState));
    hipMemset(h->d_state, 0, sizeof(DevState));
    if (hipHostMalloc((void**)&h->h_result, sizeof(FrameResult)) != hipSuccess) { g_err = "hipHostMalloc failed"; ifx_destroy(h); return IFX_E_HIP; }
    memset(h->h_result, 0, sizeof(FrameResult));
    if (hipHostMalloc((void**)&h->h_pose_early, 64) != hipSuccess) { g_err = "hipHostMalloc failed"; ifx_destroy(h); return IFX_E_HIP; }
    hipEventCreateWithFlags(&h->ev_pose_early, hipEventDisableTiming);
    ALLOC(h->d_traj, (size_t)h->max_traj * 64);
    ALLOC(h->d_scratch, 8 * 64);
    ALLOC(h->pc, C * 16); ALLOC(h->nr, C * 16); ALLOC(h->col, C * 8); ALLOC(h->tm, C * 8); ALLOC(h->ic, C * 16); ALLOC(h->votes, C * 192);
    ALLOC(h->pc2, C * 16); ALLOC(h->nr2, C * 16); ALLOC(h->col2, C * 8); ALLOC(h->tm2, C * 8); ALLOC(h->ic2, C * 16); ALLOC(h->votes2, C * 192);
    ALLOC(h->upd_owner, C * 4);
    // a chunk of 4096 slots (ifx_map.hip: MAP_THREADS x CHUNK_ROUNDS) appends to segment chunk % 8: a segment holds at most its share of the chunks
    // (+ P: a segment of the cached view list also takes its share of the surfels appended while the list lives, at most P / 4 per frame over 32 frames and 8 segments)
    h->list_seg_cap = (unsigned int)(((C + 4095) / 4096 / IFX_LIST_SEGS + 1) * 4096 + P);
    ALLOC(h->list_a, (size_t)h->list_seg_cap * IFX_LIST_SEGS * 4); ALLOC(h->list_b, (size_t)h->list_seg_cap * IFX_LIST_SEGS * 4); ALLOC(h->list_c, (size_t)h->list_seg_cap * IFX_LIST_SEGS * 4);
    if (const char* ev = getenv("IFX_RASTER_TILES")) h->opt_raster_tiles = atoi(ev);   // A/B switch for whole test runs
    if (h->own) h->opt_raster_tiles = 0;   // (the tiled rasteriser's pair records carry slots, the sharded map's keys creation numbers)
    h->tile_pair_cap = (unsigned int)std::max<size_t>(2 * C, (size_t)1 << 22);
    ALLOC(h->tile_n, 5 * 4096 * 4 + 64); ALLOC(h->tile_box, (size_t)h->list_seg_cap * IFX_LIST_SEGS * 4); ALLOC(h->tile_pairs, (size_t)h->tile_pair_cap * 4);
    hipMemset(h->tile_n, 0, 5 * 4096 * 4 + 64);
    ALLOC(h->d_list_ctr, 5 * IFX_LIST_SEGS * 32 * 4);
    hipMemset(h->d_list_ctr, 0, 5 * IFX_LIST_SEGS * 32 * 4);
    ALLOC(h->list_v, (size_t)h->list_seg_cap * IFX_LIST_SEGS * 4); ALLOC(h->list_vi, (size_t)h->list_seg_cap * IFX_LIST_SEGS * 4);
    hipMemset(h->upd_owner, 0xFF, C * 4);
    {   // the association window's texels per pixel column and row (data.vert:151-153 as its f32 loop runs: ifx_dev.h window_taps), a property of the index alone: tabulated
        ALLOC(h->assoc_vis, (size_t)(h->w + h->h));
        std::vector<uint8_t> vis((size_t)(h->w + h->h), 0);
        for (int axis = 0; axis < 2; axis++) {
            const int n = axis ? h->h : h->w;
            for (int i = 0; i < n; i++) {
                int tex[IFX_MAX_TAPS];
                window_taps(uvo_coord(i, n), (float)n, n, tex);
                const int t3[3] = {std::min(std::max(i - 1, 0), n - 1), i, std::min(i + 1, n - 1)};
                uint8_t m = 0;
                for (int q = 0; q < IFX_MAX_TAPS; q++)
                    for (int a = 0; a < 3; a++) if (tex[q] == t3[a]) m |= (uint8_t)(1u << a);
                vis[(size_t)(axis ? h->w : 0) + i] = m;
            }
        }
        hipMemcpy(h->assoc_vis, vis.data(), vis.size(), hipMemcpyHostToDevice);
    }
    ALLOC(h->labels, C * 4); ALLOC(h->labels2, C * 4);
    ALLOC(h->seq, C * 4); ALLOC(h->seq2, C * 4);
    hipMemset(h->labels, 0xFF, C * 4);
    size_t SN = std::max(C, P);
    ALLOC(h->scan_flags, SN * 4); ALLOC(h->scan_out, SN * 4); ALLOC(h->scan_block, (std::max(SN, (size_t)1 << 24) / 1024 + 8) * 4);   // also serves the 256^3-cell scan of the kNN grid
    for (int q = 0; q < 2; q++) {
        FrameSlot& f = h->slot[q];
        ALLOC(f.rgb, P * 3); ALLOC(f.depth_raw, P * 2); ALLOC(f.depth_filt, P * 2); ALLOC(f.dm, P * 4); ALLOC(f.dmf, P * 4);
        if (hipEventCreateWithFlags(&f.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&f.released, hipEventDisableTiming) != hipSuccess) {
            g_err = "hipEventCreate failed"; ifx_destroy(h); return IFX_E_HIP;
        }
    }
    hipHostMalloc((void**)&h->rgb_stage, P * 3);
    hipHostMalloc((void**)&h->depth_stage, P * 2);
    // Buffers that travel together between the ranks of a sharded map are ONE allocation each, so that an exchange point is one collective
    // (ifx_owner_exchange): [key_splat | key_ids | key_both], [index_vc | index_nr], [pred_vertex | pred_normal | pred_image | pred_inst | pred_time | tail].
    // Sharded map: one more word behind key_index and behind [key_splat | key_ids] -- the lowest live creation number (the reference's "surfel 0", ifx_map.hip FIRST_LIVE),
    // MIN-reduced with the keys in the same collective.  Hence key_both in FRONT of the pair: [key_both | key_splat | key_ids | word].
    ALLOC(h->key_index, P * 8 + 8 + IFX_KEY_SLACK); ALLOC(h->key_both, P * 8 * 3 + 8); h->key_splat = h->key_both + P; h->key_ids = h->key_both + 2 * P;
    hipMemset(h->key_index, 0xFF, P * 8 + 8 + IFX_KEY_SLACK); hipMemset(h->key_both, 0xFF, P * 8 * 3 + 8);
    if (h->own) { h->gfl_index = h->key_index + P; h->gfl_splat = h->key_both + 3 * P; }
    ALLOC(h->index_id, P * 4); ALLOC(h->index_vc, P * 16 * 2); h->index_nr = h->index_vc + 4 * P; ALLOC(h->index_ct, P * 16); ALLOC(h->index_tap, P * 16);
    // [pred_vertex | pred_conf | pred_normal | pred_image | pred_inst | pred_time | tail]: everything behind pred_vertex travels (sharded map).  The vertex itself does not: it is
    // a function of the pixel and of the winning key's depth, which every rank holds after the key exchange -- only its fourth component, the winner's confidence, is the
    // owner's to tell (pred_conf, 4 B per pixel instead of 16).  Tail: [0] vote mass of the owned surfels under the id image (whetherDoSegmentation), summed with the prediction.
    const size_t conf_bytes = ((P * 4 + 15) / 16) * 16;
    h->pred_bytes = P * 16 + conf_bytes + ((P * 26 + 15) / 16) * 16 + 16;
    ALLOC(h->pred_vertex, h->pred_bytes);
    ifx_pred_rebind(h, h->pred_vertex);
    hipMemset(h->pred_vertex, 0, h->pred_bytes);
    ALLOC(h->fill_vertex, P * 16); ALLOC(h->fill_normal, P * 16); ALLOC(h->fill_image, P * 4);
    ALLOC(h->ids_after, P * 4); ALLOC(h->ids_tmp, P * 4);
    hipMemset(h->ids_after, 0, P * 4); hipMemset(h->ids_tmp, 0, P * 4);
    hipMemset(h->pred_vertex, 0, P * 16); hipMemset(h->pred_normal, 0, P * 16); hipMemset(h->pred_image, 0, P * 4);
    hipMemset(h->index_id, 0, P * 4);
    { const size_t nm_ = (size_t)((h->w + 1) / 2) * ((h->h + 1) / 2); ALLOC(h->assoc_key, nm_ * 8); hipMemset(h->assoc_key, 0xFF, nm_ * 8); }
    ALLOC(h->assoc_target, P * 4); ALLOC(h->meas_pc, P * 16); ALLOC(h->meas_nr, P * 16); ALLOC(h->meas_col, P * 4);
    if (ifx_alloc_tracker(h) != IFX_OK) { g_err = h->err; ifx_destroy(h); return IFX_E_HIP; }
    if (ifx_alloc_instance(h) != IFX_OK) { g_err = h->err; ifx_destroy(h); return IFX_E_HIP; }
    // identity pose
    DevState hs;
    memset(&hs, 0, sizeof(hs));
    for (int k = 0; k < 16; k++) hs.pose[k] = hs.pose_inv[k] = hs.last_pose[k] = (k % 5 == 0) ? 1.f : 0.f;
    hs.weighting = 1.f;
    hipMemcpy(h->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice);
    for (int k = 0; k < 16; k++) h->h_result->pose[k] = hs.pose[k];
    if (hipDeviceSynchronize() != hipSuccess) { g_err = "device synchronize failed after allocation"; ifx_destroy(h); return IFX_E_HIP; }
    // From 1280x960 on the coarsest pyramid level (320x240: 75 blocks of four pixels per thread) runs its Gauss-Newton iterations in the persistent kernel: there
    // its meetings cost less than the launch boundaries they replace (profiles/r04_z2_*: 519 -> 546 frames/s at 1280x960 / 20M; at 640x480 the same level is within
    // 0.7 % either way and stays on the two-launch form).  A meeting that does not happen falls back inside the frame (k_gn_level_solo, ifx_tracker_fallbacks).
    if ((long long)cfg->width * cfg->height >= 1280LL * 960LL) h->opt_gn_persist = 4;
    *out = h;
    if (const char* e = getenv("IFX_OPTS")) {   // debugging aid: "name=value,name=value" applied to every handle at creation (bisecting an option without touching the caller)
        std::string all(e);
        size_t p0 = 0;
        while (p0 < all.size()) {
            size_t p1 = all.find(',', p0);
            if (p1 == std::string::npos) p1 = all.size();
            const std::string kv = all.substr(p0, p1 - p0);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) ifx_set_option(h, kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1));
            p0 = p1 + 1;
        }
    }
    return IFX_OK;
}

static void camera_free(ifx* h);
extern "C" void ifx_destroy(ifx_t* h)
{
    if (h && h->hot) { hipFree(h->hot); h->hot = nullptr; }
    if (!h) return;
    if (h->stream_c) hipStreamSynchronize(h->stream_c);
    if (h->stream_b) hipStreamSynchronize(h->stream_b);
    if (h->stream) hipStreamSynchronize(h->stream);
    ifx_comm_free(h);
    camera_free(h);
    if (h->own_slot_img) hipFree(h->own_slot_img);
    if (h->own_lat_tmp) hipFree(h->own_lat_tmp);
    ktime_flush(h);
    stage_flush(h);
    for (auto e : h->event_pool) hipEventDestroy(e);
    if (h->ev_cam_ahead) hipEventDestroy(h->ev_cam_ahead);
    if (h->ev_cam_side) hipEventDestroy(h->ev_cam_side);
    if (h->ev_slic_ahead) hipEventDestroy(h->ev_slic_ahead);
    if (h->ev_cam_parked) hipEventDestroy(h->ev_cam_parked);
    if (h->ev_lc_ready) hipEventDestroy(h->ev_lc_ready);
    if (h->ev_lc_done) hipEventDestroy(h->ev_lc_done);
    void* ptrs[] = {h->d_state, h->d_traj, h->d_scratch, h->pc, h->nr, h->col, h->tm, h->ic, h->votes, h->pc2, h->nr2, h->col2, h->tm2, h->ic2, h->votes2, h->upd_owner, h->assoc_vis, h->list_a, h->list_b, h->list_c, h->list_v, h->list_vi, h->d_list_ctr, h->tile_n, h->tile_box, h->tile_pairs, h->tile_recs, h->labels, h->seq, h->seq2,
                    h->labels2, h->scan_flags, h->scan_out, h->scan_block, h->slot[0].rgb, h->slot[0].depth_raw, h->slot[0].depth_filt, h->slot[0].dm, h->slot[0].dmf, h->slot[1].rgb, h->slot[1].depth_raw,
                    h->slot[1].depth_filt, h->slot[1].dm, h->slot[1].dmf, h->key_index, h->key_both,
                    h->index_id, h->index_vc, h->index_ct, h->index_tap, h->pred_vertex, h->fill_vertex,
                    h->fill_normal, h->fill_image, h->ids_after, h->ids_tmp, h->assoc_key, h->assoc_target, h->meas_pc, h->meas_nr, h->meas_col};
    for (void* p : ptrs) if (p) hipFree(p);
    for (size_t q = 2; q < h->slot.size(); q++) { FrameSlot& f2 = h->slot[q]; void* p2[] = {f2.rgb, f2.depth_raw, f2.depth_filt, f2.dm, f2.dmf}; for (void* p : p2) if (p) hipFree(p); }
    if (h->h_result) hipHostFree(h->h_result);
    if (h->h_pose_early) hipHostFree(h->h_pose_early);
    if (h->ev_pose_early) hipEventDestroy(h->ev_pose_early);
    for (int q_ = 0; q_ < 2; q_++) { if (h->hint_stage_rgb[q_]) hipHostFree(h->hint_stage_rgb[q_]); if (h->hint_stage_depth[q_]) hipHostFree(h->hint_stage_depth[q_]); }
    if (h->rgb_stage) hipHostFree(h->rgb_stage);
    if (h->depth_stage) hipHostFree(h->depth_stage);
    ifx_free_tracker(h);
    ifx_free_instance(h);
    ifx_slic_free(h);
    ifx_knn_free_all(h);
    for (int q = 0; q < 2; q++) { if (h->slot[q].ready) hipEventDestroy(h->slot[q].ready); if (h->slot[q].released) hipEventDestroy(h->slot[q].released); }
    if (h->ev_gate) hipEventDestroy(h->ev_gate);
    if (h->stream_c) hipStreamDestroy(h->stream_c);
    if (h->stream_b) hipStreamDestroy(h->stream_b);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int ifx_set_loop_closure(ifx_t* h, int enable, int count_thresh, float err_thresh, float cov_thresh)
{
    if (!h) return IFX_E_INVALID;
    if (h->hint_rgb || h->slot[h->tick & 1].for_tick == h->tick) { h->err = "loop-closure detection cannot change while a frame is announced ahead"; return IFX_E_STATE; }
    ifx_drop_tracked(h);
    if (h->stream_c) { HIPCHK(h, hipStreamSynchronize(h->stream_c)); h->lc_pending = 0; }
    if (enable) {
        int r = ifx_tracker_alloc_m2m(h);
        if (r) return r;
        if (!h->stream_c && hipStreamCreate(&h->stream_c) != hipSuccess) { h->err = "stream creation failed"; return IFX_E_HIP; }   // (one-stream handles)
        if (!h->ev_lc_ready) {
            if (hipEventCreateWithFlags(&h->ev_lc_ready, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_lc_done, hipEventDisableTiming) != hipSuccess) { h->err = "event creation failed"; return IFX_E_HIP; }
        }
    }
    h->lc_enable = enable ? 1 : 0; h->lc_count_thresh = count_thresh; h->lc_err_thresh = err_thresh; h->lc_cov_thresh = cov_thresh;
    return IFX_OK;
}
extern "C" int ifx_set_loop_closure_callback(ifx_t* h, ifx_loop_closure_cb cb, void* user)
{
    if (!h) return IFX_E_INVALID;
    h->lc_cb = cb; h->lc_user = user;
    return IFX_OK;
}
extern "C" int ifx_set_fern_callback(ifx_t* h, ifx_fern_cb cb, void* user)
{
    if (!h) return IFX_E_INVALID;
    if (cb && !h->lc_enable) { h->err = "the fern callback runs inside the loop-closure block: enable it first (ifx_set_loop_closure)"; return IFX_E_STATE; }
    h->fern_cb = cb; h->fern_user = user;
    return IFX_OK;
}
extern "C" int ifx_loop_closure_diag(ifx_t* h, float* out24)
{
    if (!h || !out24) return IFX_E_INVALID;
    if (!h->h_lc) { memset(out24, 0, 24 * 4); return IFX_OK; }   // detection never enabled
    if (h->ev_result) HIPCHK(h, hipEventSynchronize(h->ev_result));
    if (h->lc_event_valid) HIPCHK(h, hipEventSynchronize(h->ev_lc_done));   // the model-to-model tracker may outlive its frame
    memcpy(out24, h->h_lc, 24 * 4);
    return IFX_OK;
}

extern "C" int ifx_set_option(ifx_t* h, const char* name, int value)
{
    if (!h || !name) return IFX_E_INVALID;
    std::string s(name);
    if (s == "compact_every_frame") h->opt_compact_every_frame = value;
    else if (s == "kernel_timing") { hipStreamSynchronize(h->stream); ktime_flush(h); h->opt_kernel_timing = value; }
    else if (s == "reference_passes") { h->opt_reference_passes = value; ifx_vlist_reap(h); hs_invalidate_view(h); }   // the view-list path is off while it is set: no device-side "valid" may outlive the switch
    else if (s == "two_streams") h->opt_two_streams = value;
    else if (s == "stage_timing") h->opt_stage_timing = value;
    else if (s == "track_ahead") h->opt_track_ahead = value;
    else if (s == "slic_ahead") h->opt_slic_ahead = value;
    else if (s == "fold_result") h->opt_fold_result = value;
    else if (s == "clean_raster") h->opt_clean_raster = value;
    else if (s == "hot_records") { h->opt_hot = value; h->hot_valid = 0; }
    else if (s == "hot_verify") h->opt_hot_verify = value;
    else if (s == "side_late") h->opt_side_late = value;
    else if (s == "vote_per_mask") h->opt_vote_per_mask = value;
    else if (s == "own_first_live") h->opt_own_first_live = value;
    else if (s == "own_track_rows") {
        if (!h->own) { h->err = "own_track_rows: the handle was not created for a sharded map"; return IFX_E_STATE; }
        h->opt_own_track_rows = value;   // (takes effect while the library holds the communicator and no single rank tracks: ifx_owner_init_comm / _set_comm, ifx_owner_set_tracking_rank)
    }
    else if (s == "own_track_rows_emulate") h->opt_own_track_rows_emulate = value;
    else if (s == "own_key_rs") {
        if (!h->own) { h->err = "own_key_rs: the handle was not created for a sharded map"; return IFX_E_STATE; }
        if (h->own_g * 8 > IFX_KEY_SLACK) { h->err = "own_key_rs: more ranks than the key image's slack allows"; return IFX_E_INVALID; }
        h->opt_own_key_rs = value;
    }
    else if (s == "own_lazy_ids") {
        if (!h->own) { h->err = "own_lazy_ids: the handle was not created for a sharded map"; return IFX_E_STATE; }
        if (value && ifx_own_lattice(h) + 1 > 16 * 1024) { h->err = "own_lazy_ids: the image's id lattice is larger than the pack kernel's one workgroup holds (16 383 entries)"; return IFX_E_INVALID; }
        h->opt_own_lazy_ids = value;   // (takes effect with the next frame; a sparse image that is still around is completed on demand as before)
    }
    else if (s == "vlist_one") h->opt_vlist_one = value;
    else if (s == "overdue_rule") h->opt_overdue_rule = value;
    else if (s == "cam_swap") h->opt_cam_swap = value;
    else if (s == "cam_side") h->opt_cam_side = value;
    else if (s == "host_entry_async") h->opt_host_entry_async = value;
    else if (s == "gn_prologue_blocks") h->opt_gn_prologue_blocks = value;
    else if (s == "compact_divisor") h->opt_compact_divisor = value;
    else if (s == "icp_blocks") h->opt_icp_blocks = std::max(0, std::min(2048, value));
    else if (s == "view_blocks") h->opt_view_blocks = std::max(0, std::min(65536, value));
    else if (s == "clean_blocks") h->opt_clean_blocks = std::max(0, std::min(65536, value));
    else if (s == "index_blocks") h->opt_index_blocks = std::max(0, std::min(8192, value));
    else if (s == "res_blocks") h->opt_res_blocks = std::max(0, std::min(4096, value));
    else if (s == "icp_lds") {
#ifdef IFX_EXPERIMENTS
        h->opt_icp_lds = value;
#else
        if (value) { h->err = "icp_lds: a measured alternative that lost (DESIGN.md section 6); this library was built without -DIFX_EXPERIMENTS"; return IFX_E_STATE; }
#endif
    }
    else if (s == "rgb_blocks") h->opt_rgb_blocks = std::max(0, std::min(1024, value));
    else if (s == "raster_tiles") h->opt_raster_tiles = value;
    else if (s == "view_list") { h->opt_vlist = value; ifx_vlist_reap(h); hs_invalidate_view(h); }
    else if (s == "seg_aside") h->opt_seg_aside = value;
    else if (s == "pace") h->opt_pace = value;
    else if (s == "lc_view") h->opt_lc_view = value;
    else if (s == "side_gate") h->opt_side_gate = value;
    else if (s == "ff_union") h->opt_ff_union = value;
    else if (s == "fold_finish") h->opt_fold_finish = value;
    else if (s == "lazy_ids") { ifx_ids_ensure(h); h->opt_lazy_ids = value; }
    else if (s == "seg_device") h->opt_seg_device = value;
    else if (s == "ff_rounds") h->opt_ff_rounds = value;
    else if (s == "labels_incremental") { h->opt_labels_incremental = value; h->labels_stale_all = 1; }
    else if (s == "icp_px") {
#ifdef IFX_EXPERIMENTS
        h->opt_icp_px = value;
#else
        if (value) { h->err = "icp_px: a measured alternative that lost (DESIGN.md section 6); this library was built without -DIFX_EXPERIMENTS"; return IFX_E_STATE; }
#endif
    }
    else if (s == "model_fused") {
#ifdef IFX_EXPERIMENTS
        h->opt_model_fused = value;
#else
        if (value) { h->err = "model_fused: a measured alternative that lost (DESIGN.md section 6); this library was built without -DIFX_EXPERIMENTS"; return IFX_E_STATE; }
#endif
    }
    else if (s == "gn_persist_blocks") { if (value < 1) return IFX_E_INVALID; ifx_drop_tracked(h); h->opt_gn_persist_blocks = value; }
    else if (s == "gn_persist") {   // a bit per pyramid level: that level's Gauss-Newton iterations in one persistent launch (k_gn_level); default 4 = the coarsest level only
        if (value < 0 || value > 7) { h->err = "gn_persist is a mask of pyramid levels (0..7)"; return IFX_E_INVALID; }
        ifx_drop_tracked(h);
        h->opt_gn_persist = value;
    }
    else if (s == "gn_spin_limit" || s == "gn_fault") {   // test hooks of the persistent level kernel's fallback: polls before a meeting gives up (0: 2^21) / block 1 never arrives at meeting number `value`
        if (value < 0) return IFX_E_INVALID;
        ifx_drop_tracked(h);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy((char*)h->d_state + (s == "gn_fault" ? offsetof(DevState, gn_fault) : offsetof(DevState, gn_spin_limit)), &value, sizeof(int), hipMemcpyHostToDevice));
    }
    else if (s == "gn_prologue") { ifx_drop_tracked(h); h->opt_gn_prologue = value ? 1 : 0; }
    else if (s == "raster_lds") h->opt_raster_lds = value;
    else if (s == "raster_earlyz") h->opt_raster_earlyz = value;
    // ElasticFusion::setPyramid / setFastOdom / setSo3 / setIcpWeight (EF/ElasticFusion.h:153-176): tracker configuration from the next frame on;
    // refused while a frame is announced ahead (its image-only work may already be on the queue with the old configuration)
    else if (s == "pyramid" || s == "fast_odom" || s == "so3" || s == "icp_weight_x1000") {
        if (h->hint_rgb || h->slot[h->tick & 1].for_tick == h->tick) { h->err = "tracker options cannot change while a frame is announced ahead"; return IFX_E_STATE; }
        ifx_drop_tracked(h);
        if (s == "pyramid") h->cfg.pyramid = value ? 1 : 0;
        else if (s == "fast_odom") h->cfg.fast_odom = value ? 1 : 0;
        else if (s == "so3") h->cfg.so3 = value ? 1 : 0;
        else { if (value < 0 || value > 100000) { h->err = "icp_weight_x1000 must be in [0, 100000]"; return IFX_E_INVALID; } h->cfg.icp_weight = value / 1000.f; }
    }
    else { h->err = "unknown option " + s; return IFX_E_INVALID; }
    return IFX_OK;
}

// ------------------------------------------------------------------ frame orchestration
__global__ void k_frame_result(DevState* __restrict__ st, FrameResult* __restrict__ out, float* __restrict__ traj_slot)
{
    if (threadIdx.x < 64) frame_result_wave(st, out, traj_slot, 0, (int)threadIdx.x);   // (ifx_ctx.h: shared with the last block of k_splat_resolve)
}

// Frame side of frame `tick` into slot s (copy-in, bilateral + metric depth, frame pyramids, SO(3) pre-alignment).
// It depends only on the input images and on the previous frame's intensity pyramid, so it goes to the side
// stream: next to the model pyramid of its own frame, or -- when prefetched -- under the previous frame's
// tracking and map passes.  src_kind: 0 device pointers, 1 the pinned staging buffers.
static int enqueue_frame_side(ifx* h, int s, int tick, const uint8_t* rgb, const uint16_t* depth, int src_kind)
{
    FrameSlot& f = h->slot[s];
    const int bound = h->cur_slot;
    hipStream_t q = h->opt_two_streams ? h->stream_b : h->stream;
    if (q != h->stream) HIPCHK(h, hipStreamWaitEvent(q, f.released, 0));   // the frame that last used this slot (two frames ago) is done
    h->cur = q;
    ifx_bind_slot(h, s);
    hipMemcpyKind kind = src_kind ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    // depth first: the bilateral filter (the longest launch of the frame side, 65 us) needs nothing else.  Host-pointer entry (ifx_process_frame): the caller's colour image is
    // copied into the pinned staging buffer only NOW -- 0.9 MB of host memcpy that used to sit in front of the first transfer run under the depth transfer and the filter
    hipError_t e2 = hipMemcpyAsync(f.depth_raw, depth, (size_t)h->P * 2, kind, q), e1 = hipSuccess;
    if (e2 == hipSuccess) {
        StageTimer t(h, 3);
        ifx_preprocess(h);
        if (h->late_rgb_src && src_kind == 1) { memcpy(h->rgb_stage, h->late_rgb_src, (size_t)h->P * 3); h->late_rgb_src = nullptr; }
        e1 = hipMemcpyAsync(f.rgb, rgb, (size_t)h->P * 3, kind, q);
        if (e1 == hipSuccess) ifx_tracker_frame_side(h, tick == 1);
    }
    hipEventRecord(f.ready, q);
    f.for_tick = tick; f.src_rgb = rgb; f.src_depth = depth;
    h->cur = h->stream;
    ifx_bind_slot(h, bound);
    if (e1 != hipSuccess || e2 != hipSuccess) { h->err = "frame copy failed"; return IFX_E_HIP; }
    return IFX_OK;
}

// frame side of the frame announced by ifx_hint_next_frame_device: called from the middle of the tracker's enqueue
// (or at the end of the frame when nothing was tracked)
int ifx_enqueue_hinted_frame_side(ifx* h)
{
    if (!h->hint_rgb || !h->opt_two_streams) { h->hint_rgb = nullptr; return IFX_OK; }
    const uint8_t* r = h->hint_rgb; const uint16_t* d = h->hint_depth;
    const int kind = h->hint_kind;
    h->hint_rgb = nullptr; h->hint_depth = nullptr; h->hint_kind = 0;
    // Frame sides are ordered among themselves: each reads the image pyramid of the frame before it and all of them sum into the same SO(3) accumulators / ticket.
    // On the side stream that order is the stream's; but the current frame's side may have run on the MAIN stream (the sharded entry does that for a frame that was
    // not announced) -- its "slot ready" event orders the announced frame's side behind it (recorded on the side stream itself in the usual case: no wait at all).
    // Without it the two ran side by side once in a few hundred frames: a garbled SO(3) start for the current frame (tests: the rare failure of
    // test_owner_sharded_rccl_world_of_one_in_library at the frame after its host-pointer frame).
    if (h->stream_b && h->slot[h->tick & 1].ready) HIPCHK(h, hipStreamWaitEvent(h->stream_b, h->slot[h->tick & 1].ready, 0));
    int rr = enqueue_frame_side(h, (h->tick + 1) & 1, h->tick + 1, r, d, kind);
    if (!rr && kind == 1) h->prestaged_tick = h->tick + 1;   // (the frame's data is in its parity's staging pair: ifx_process_frame_ex takes the slot as it is)
    return rr;
}

// Local loop-closure detection of the frame just tracked (EF/ElasticFusion.cpp:453-566 with no fern match): predict() at the new pose,
// INACTIVE prediction, model-to-model tracking, covariance / count / error gates.  While tick - timeDelta < 1 no surfel the frames created
// can be old enough (lastTime >= 1), so unless a map was uploaded the whole block is skipped on the host -- exactly what the gates would
// decide on an empty render.
__global__ void k_lc_idle(DevState* st, float* __restrict__ host_lc)
{
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 23; k++) st->lc[k] = 0.f;
    for (int k = 0; k < 16; k++) st->lc[6 + k] = st->pose[k];
    st->lc[23] = (float)st->lc_candidates;
    for (int k = 0; k < 24; k++) host_lc[k] = st->lc[k];
}
// Part 1, before the map passes: the two renders.  With a callback (the caller deforms the map on an accepted candidate) the model-to-model
// tracker follows at once on the main stream and the host waits for the verdict; otherwise the tracker is deferred to part 2.
static int enqueue_loop_closure_renders(ifx* h)
{
    h->lc_deferred = 0;
    if (!h->fern_cb && !h->map_external && h->tick - h->cfg.time_delta < 1) {
        LAUNCH(h, "lc_idle", dim3(1), dim3(64), k_lc_idle, h->d_state, h->h_lc);
        return IFX_OK;
    }
    // the previous frame's model-to-model run may still be on its stream: it owns the act* / old* images and its state until it is done
    if (h->lc_pending) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_lc_done, 0)); h->lc_pending = 0; }
    {
        StageTimer t(h, 1);
        ifx_tracker_m2m_begin(h);
        ifx_map_predict_loop_closure(h);
    }
    if (h->fern_cb) {   // Ferns::findFrame and the global deformation (EF/ElasticFusion.cpp:457-514): host code of the caller, every frame, on the finished renders
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->in_fern_cb = 1;
        int r = h->fern_cb(h, h->fern_user);
        h->in_fern_cb = 0;
        if (r < 0) { h->err = "fern callback failed"; return r; }
        if (r > 0) {    // matched to a fern and deformed (rawGraph.size() > 0): no local loop closure this frame (:516)
            LAUNCH(h, "lc_idle", dim3(1), dim3(64), k_lc_idle, h->d_state, h->h_lc);
            return IFX_OK;
        }
    }
    if (!h->lc_cb) {
        if (h->opt_two_streams && h->stream_c) HIPCHK(h, hipEventRecord(h->ev_lc_ready, h->stream));
        h->lc_deferred = 1;
        return IFX_OK;
    }
    {
        StageTimer t(h, 0);
        int r = ifx_tracker_loop_closure(h);
        if (r) return r;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));   // the verdict is in pinned memory (k_m2m_decide)
    if (h->h_lc[5] != 0.f) {
        float lc[24];
        memcpy(lc, h->h_lc, sizeof(lc));
        int r = h->lc_cb(h, lc, h->lc_user);
        if (r < 0) { h->err = "loop-closure callback failed"; return r; }
    }
    return IFX_OK;
}
// Part 2, after the map passes of the frame are on the main stream: the model-to-model tracker reads only the two renders (images of their
// own) and its own state, and nothing of the frame consumes its verdict, so it goes to a stream of its own and runs under the map passes, the
// end-of-frame predict() and the next frame's tracker (both trackers are chains of small latency-bound launches).  Enqueued AFTER the map
// passes on the host as well: its ~50 launches would otherwise keep the main queue empty for their whole enqueue time.
static int enqueue_loop_closure_tracker(ifx* h)
{
    if (!h->lc_deferred) return IFX_OK;
    h->lc_deferred = 0;
    const bool aside = h->opt_two_streams && h->stream_c;
    if (aside) {
        HIPCHK(h, hipStreamWaitEvent(h->stream_c, h->ev_lc_ready, 0));
        h->cur = h->stream_c;
    }
    int r;
    {
        StageTimer t(h, 0);
        r = ifx_tracker_loop_closure(h);
    }
    if (aside) { hipEventRecord(h->ev_lc_done, h->stream_c); h->cur = h->stream; h->lc_pending = 1; h->lc_event_valid = 1; }
    return r;
}

// A frame that takes its pose from the caller never reaches RGBDOdometry::initRGB (EF/ElasticFusion.cpp:330-356: only the tracking branch calls it), so the tracker's "last next
// image" -- what the NEXT tracked frame's SO(3) pre-alignment compares itself with -- stays that of the last TRACKED frame.  Here every frame's side fills the intensity
// pyramid of its slot and the next frame reads its predecessor's slot: the held frame's slot gets the predecessor's pyramid back, on stream q behind this frame's side and in
// front of the next one's (which waits for the slot's "ready" event).  Found by tests/test_gpu_sweep.py: a tracked frame behind a held one was 5 mm off the oracle.
// Not with camera contexts (K streams, beyond the reference): a camera enters with its extrinsic pose, and its next frame is compared with THAT frame's image, not with the
// image another camera saw last.
static int hold_last_image(ifx* h, int s, hipStream_t q)
{
    if (!h->cams.empty() || s > 1) return IFX_OK;
    FrameSlot& f = h->slot[s];
    const FrameSlot& prev = h->slot[s ^ 1];
    for (int l = 0; l < IFX_NUM_PYRS; l++)
        HIPCHK(h, hipMemcpyAsync(f.next_img[l], prev.next_img[l], (size_t)h->pyr.w[l] * h->pyr.h[l], hipMemcpyDeviceToDevice, q));
    HIPCHK(h, hipEventRecord(f.ready, q));   // (whoever waits for this slot's frame side also waits for the copy)
    return IFX_OK;
}

// ElasticFusion::processFrame, EF/ElasticFusion.cpp:269-720, enqueued on the handle's streams.  Of the loop-closure block (:450-617)
// the local detection is implemented (ifx_set_loop_closure), the fern lookup and both deformations run in the caller's callbacks (the graph
// optimiser is host code of the reference); without the detection the first predict() of :453, whose only consumers are those stages, is not executed.
int ifx_tracker_bootstrap_pose(ifx* h, const float* d_in_pose16);
static int enqueue_frame(ifx* h, const uint8_t* rgb, const uint16_t* depth, int src_kind, const float* in_pose16, float weight_mult, int bootstrap = 0)
{
    if (h->own) { h->err = "a sharded map (n_ranks > 1) is driven through ifx_owner_frame_phase / ifx_owner_exchange"; return IFX_E_STATE; }
    if (bootstrap && !in_pose16) { h->err = "bootstrap needs inPose (EF/ElasticFusion.cpp:352-356)"; return IFX_E_INVALID; }
    // Bounded run-ahead: the previous frame's RESULT (its map passes; the tracker enqueued ahead of this frame is still behind it on the queue, so the device never
    // idles) before this frame goes onto the queues.  A host that enqueues without ever looking back piles frame sides, event waits and barrier packets several
    // frames deep, and the same workload runs a quarter slower (bench.py --no-instance: 1100 against 1460 frames/s; a host that asks ifx_should_segment every
    // frame waits exactly here anyway).  Option "pace" = 0 restores the unbounded enqueue.
    if (h->opt_pace && h->ev_result && h->tick > 1) HIPCHK(h, hipEventSynchronize(h->ev_result));
    const int s = h->tick & 1;
    FrameSlot& f = h->slot[s];
    const bool prepared = f.for_tick == h->tick && f.src_rgb == rgb && f.src_depth == depth && (src_kind == 0 || h->prestaged_tick == h->tick);   // (prestaged: ifx_process_frame_ex put the frame side of its staged frame on the side stream before it waited for the previous frame)
    // a tracker run enqueued behind the previous frame counts only for exactly this frame, tracked, with the default weight
    const bool tracked = prepared && h->tracked_ahead == h->tick && !in_pose16;
    h->n_side_prepared += prepared ? 1 : 0; h->n_tracked_ahead += tracked ? 1 : 0;
    ifx_drop_tracked(h);
    if (!prepared) {
        if (h->slic_ahead_tick == h->tick) h->slic_ahead_tick = -1;   // superpixels run ahead for a frame that was announced and did not come: not this frame's (the run stays "busy" until somebody queues behind it)
        int r = enqueue_frame_side(h, s, h->tick, rgb, depth, src_kind);
        if (r) return r;
    }
    f.for_tick = -1;
    ifx_bind_slot(h, s);
    h->last_frame_slot = s;
    if (h->tick == 1) {
        HIPCHK(h, hipStreamWaitEvent(h->stream, f.ready, 0));
        StageTimer t(h, 1);
        ifx_map_init_first(h);
    } else {
        {
            StageTimer t(h, 0);
            if (tracked) {
                ifx_tracker_commit(h);
                if (weight_mult != 1.0f) ifx_tracker_set_weight(h, weight_mult);
                if (h->opt_side_gate == 1 && h->ev_gate) { hipEventRecord(h->ev_gate, h->stream); hipStreamWaitEvent(h->stream_b, h->ev_gate, 0); }   // (experiment) not under the tracker's tail
                // (option side_late: the announced frame's side enqueued behind this frame's map passes instead -- measured slower, see ifx_ctx.h)
                if (h->opt_side_gate != 2 && !h->opt_side_late) {
                    int r = ifx_enqueue_hinted_frame_side(h);   // no tracker enqueue to hide it in: it runs under this frame's map passes
                    if (r) return r;
                }
            } else if (!in_pose16 || bootstrap) {
                ifx_tracker_model_side(h);                       // model pyramid: independent of the frame side
                HIPCHK(h, hipStreamWaitEvent(h->stream, f.ready, 0));
                if (bootstrap) {   // inPose is a guess, not a replacement: currPose = currPose * inPose AFTER the model maps were placed with the old pose (:334-356)
                    float* slot = h->d_scratch + 2 * 16;
                    HIPCHK(h, hipMemcpyAsync(slot, in_pose16, 64, hipMemcpyHostToDevice, h->stream));
                    ifx_tracker_bootstrap_pose(h, slot);
                }
                ifx_tracker_run_frame(h, 1, bootstrap);
                if (weight_mult != 1.0f) ifx_tracker_set_weight(h, weight_mult);
            } else {
                HIPCHK(h, hipStreamWaitEvent(h->stream, f.ready, 0));
                float* slot = h->d_scratch + 2 * 16;
                HIPCHK(h, hipMemcpyAsync(slot, in_pose16, 64, hipMemcpyHostToDevice, h->stream));
                ifx_tracker_external_pose(h, slot, weight_mult);
                hold_last_image(h, s, h->opt_two_streams ? h->stream_b : h->stream);
            }
        }
        if (h->want_early_pose && !h->lc_enable) {   // ifx_process_frame: the pose the call returns, behind the tracker and in front of the map passes
            HIPCHK(h, hipMemcpyAsync(h->h_pose_early, (const void*)h->d_state, 64, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipEventRecord(h->ev_pose_early, h->stream));
            h->early_pose_valid = 1;
        }
        if (h->lc_enable) {
            int r = enqueue_loop_closure_renders(h);
            if (r) return r;
        }
        {
            StageTimer t(h, 1);
            h->frame_weight_mult = weight_mult;
            h->result_fold_traj = h->d_traj + (size_t)(h->n_traj % h->max_traj) * 16;   // (ifx_map_frame asks whether the prediction's resolve will take the frame result along)
            ifx_map_frame(h);
            h->result_fold_traj = nullptr;
        }
        if (h->lc_enable) {
            int r = enqueue_loop_closure_tracker(h);
            if (r) return r;
        }
    }
    const int slot = h->n_traj % h->max_traj;
    {
        StageTimer t(h, 1);
        h->result_fold_traj = h->d_traj + (size_t)slot * 16;   // (the prediction's resolve is the frame's last launch on the view-list path: it can take the frame result along)
        h->result_folded = 0;
        ifx_map_predict(h);
        h->result_fold_traj = nullptr;
    }
    if (!h->result_folded) LAUNCH(h, "frame_result", dim3(1), dim3(64), k_frame_result, h->d_state, h->h_result, h->d_traj + (size_t)slot * 16);
    hipEventRecord(f.released, h->stream);   // one marker: the side stream waits for it before it reuses the slot,
    h->ev_result = f.released;               // the host before it reads the frame result
    {
        if (h->opt_side_gate == 2 && h->hint_rgb && h->opt_two_streams) hipStreamWaitEvent(h->stream_b, f.released, 0);   // (experiment) under the next tracker only
        int r = ifx_enqueue_hinted_frame_side(h);   // not consumed by the tracker (first frame, external pose)
        if (r) return r;
    }
    h->seg_counts_valid = 1;
    h->n_traj++;
    h->tick++;
    // Tracking of the announced next frame reads only its frame slot, the prediction just rendered and the pose: nothing
    // a segmentation call in between touches.  Enqueue it now, parked (DevState::spec_*), so that the GPU has ~0.6 ms of
    // work while the host waits for this frame's result to decide about segmentation.
    FrameSlot& nf = h->slot[h->tick & 1];
    if (h->opt_track_ahead && h->opt_two_streams && nf.for_tick == h->tick && h->tick > 1) {
        ifx_bind_slot(h, h->tick & 1);
        HIPCHK(h, hipStreamWaitEvent(h->stream, nf.ready, 0));
        {
            StageTimer t(h, 0);
            ifx_tracker_model_side(h, 1);
            ifx_tracker_run_frame(h, 0);
        }
        ifx_bind_slot(h, s);
        h->tracked_ahead = h->tick;
    }
    return IFX_OK;
}

// ---- camera contexts: K streams into one map (BASELINE configuration 5).  The single-GPU semantics of a FRAME SET -- one frame per camera -- is: the K frames
// processed one after the other, in camera order, on the one map; every camera tracks against the prediction rendered at the end of ITS last frame.
// A run ahead parks, and the frame that takes it commits, the pose block AND the tracker's diagnostics (lastICPError ... rgb_sigma: what FrameResult.diag and
// ifx_tracker_diag report) -- one launch for the two ranges, words of 4 bytes
#define IFX_DIAG_OFF offsetof(DevState, lastICPError)
#define IFX_DIAG_BYTES (offsetof(DevState, seg_counts) - offsetof(DevState, lastICPError))
#define IFX_AHEAD_BYTES (IFX_CAM_STATE_BYTES + IFX_DIAG_BYTES)
static_assert(IFX_DIAG_OFF % 4 == 0 && IFX_DIAG_BYTES % 4 == 0 && IFX_CAM_STATE_BYTES % 4 == 0, "word copies");
__global__ void k_ahead_block(uint32_t* __restrict__ dst_pose, uint32_t* __restrict__ dst_diag, const uint32_t* __restrict__ src_pose, const uint32_t* __restrict__ src_diag)
{
    for (int i = threadIdx.x; i < (int)(IFX_CAM_STATE_BYTES / 4); i += blockDim.x) dst_pose[i] = src_pose[i];
    for (int i = threadIdx.x; i < (int)(IFX_DIAG_BYTES / 4); i += blockDim.x) dst_diag[i] = src_diag[i];
}
static_assert(offsetof(DevState, count) == IFX_CAM_STATE_BYTES, "the pose block of DevState (pose, pose_inv, last_pose, weighting, dense_enough) is what a camera context parks");
static void camera_free(ifx* h)
{
    for (CamCtx& c : h->cams) {
        hipFree(c.state); hipFree(c.pred); hipFree(c.fill_v); hipFree(c.fill_n); hipFree(c.fill_i); hipFree(c.ids); hipFree(c.ahead_pose); if (c.ev_ahead) hipEventDestroy(c.ev_ahead);
        for (int l = 0; l < IFX_NUM_PYRS; l++) hipFree(c.img[l]);
    }
    h->cams.clear();
}
extern "C" int ifx_camera_count(ifx_t* h, int n_cameras)
{
    if (!h || n_cameras < 1 || n_cameras > 64) return IFX_E_INVALID;
    if (h->stream_c) hipStreamSynchronize(h->stream_c);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    camera_free(h);
    if (h->slot.size() < 3 + (size_t)n_cameras) h->slot.resize(3 + (size_t)n_cameras);   // (a camera's run-ahead slot is allocated on first use; sized here so that no reference into the vector moves later)
    h->cur_cam = 0;
    if (n_cameras == 1) return IFX_OK;
    const size_t P = (size_t)h->P;
    h->cams.resize((size_t)n_cameras);
    for (CamCtx& c : h->cams) {
        HIPCHK(h, hipMalloc(&c.state, IFX_CAM_STATE_BYTES)); HIPCHK(h, hipMalloc(&c.pred, h->pred_bytes));
        HIPCHK(h, hipMalloc(&c.fill_v, P * 16)); HIPCHK(h, hipMalloc(&c.fill_n, P * 16)); HIPCHK(h, hipMalloc(&c.fill_i, P * 4)); HIPCHK(h, hipMalloc(&c.ids, P * 4));
        for (int l = 0; l < IFX_NUM_PYRS; l++) HIPCHK(h, hipMalloc(&c.img[l], (size_t)(h->w >> l) * (h->h >> l)));
    }
    return IFX_OK;
}
// camera `cam` takes over the handle: the current camera's set is parked, cam's is brought in (a camera selected for the first time starts as a copy of the
// current one: give it its pose with ifx_set_pose / an external pose for its first frame).  Enqueue-only; between frames.
extern "C" int ifx_camera_select(ifx_t* h, int cam)
{
    if (!h || cam < 0 || (cam > 0 && (size_t)cam >= h->cams.size())) return IFX_E_INVALID;
    if (h->cams.empty() || cam == h->cur_cam) return IFX_OK;
    if (h->hint_rgb || h->slot[h->tick & 1].for_tick == h->tick) { h->err = "ifx_camera_select: a frame is announced ahead"; return IFX_E_STATE; }
    ifx_drop_tracked(h);
    if (h->housekeeping_due) {   // (option host_entry_async: the compaction decision of the frame just processed, while its camera is still the live one -- a compaction re-renders at the live pose)
        if (h->ev_result) HIPCHK(h, hipEventSynchronize(h->ev_result));
        ifx_housekeeping(h);
        h->housekeeping_due = 0;
    }
    if (h->lc_pending && h->stream_c) { HIPCHK(h, hipStreamSynchronize(h->stream_c)); h->lc_pending = 0; }
    ifx_vlist_reap(h); hs_invalidate_view(h);   // another pose: the cached view list is void (reaped first: no slot outlives the age rule unseen)
    const size_t P = (size_t)h->P;
    FrameSlot& prev = h->slot[(h->tick & 1) ^ 1];   // where the next frame side looks for its "previous image"
    FrameSlot& last = h->slot[(size_t)h->last_frame_slot];   // the slot of the frame just processed (a camera's run-ahead slot when the frame took its run)
    auto move = [&](CamCtx& c, bool save) -> int {
        const hipMemcpyKind k = hipMemcpyDeviceToDevice;
#define CAMCP(ctx_ptr, live_ptr, bytes) HIPCHK(h, save ? hipMemcpyAsync((ctx_ptr), (live_ptr), (bytes), k, h->stream) : hipMemcpyAsync((live_ptr), (ctx_ptr), (bytes), k, h->stream))
        CAMCP(c.state, (void*)h->d_state, IFX_CAM_STATE_BYTES);
        CAMCP(c.pred, (void*)h->pred_vertex, h->pred_bytes);
        CAMCP(c.fill_v, (void*)h->fill_vertex, P * 16); CAMCP(c.fill_n, (void*)h->fill_normal, P * 16); CAMCP(c.fill_i, (void*)h->fill_image, P * 4);
        CAMCP(c.ids, (void*)h->ids_after, P * 4);
        for (int l = 0; l < IFX_NUM_PYRS; l++) CAMCP(c.img[l], (void*)(save ? last : prev).next_img[l], (size_t)(h->w >> l) * (h->h >> l));
#undef CAMCP
        return IFX_OK;
    };
    int r = ifx_ids_ensure(h);   // a parked id image is a whole one (the map moves on under the other cameras)
    if (r) return r;
    if (h->cams[(size_t)cam].valid && h->opt_cam_swap) {
        // Both contexts exist: the big blocks -- prediction 46 B/px, fill-in 36, id image 4: 26 MB at 640x480 -- change hands by pointer (the parked context takes
        // the live buffers, the live pointers take the other camera's); only the pose block and the intensity pyramid (0.4 MB) are copied.  53 MB of device
        // copies per camera switch was 0.1 ms of every frame on every rank of a K-stream run.
        CamCtx& o = h->cams[(size_t)h->cur_cam];
        CamCtx& n = h->cams[(size_t)cam];
        HIPCHK(h, hipMemcpyAsync(o.state, (void*)h->d_state, IFX_CAM_STATE_BYTES, hipMemcpyDeviceToDevice, h->stream));
        for (int l = 0; l < IFX_NUM_PYRS; l++) HIPCHK(h, hipMemcpyAsync(o.img[l], last.next_img[l], (size_t)(h->w >> l) * (h->h >> l), hipMemcpyDeviceToDevice, h->stream));
        { uint8_t* t = o.pred; o.pred = (uint8_t*)h->pred_vertex; ifx_pred_rebind(h, (float*)n.pred); n.pred = t; }
        { float* t = o.fill_v; o.fill_v = h->fill_vertex; h->fill_vertex = n.fill_v; n.fill_v = t; }
        { float* t = o.fill_n; o.fill_n = h->fill_normal; h->fill_normal = n.fill_n; n.fill_n = t; }
        { uint8_t* t = o.fill_i; o.fill_i = h->fill_image; h->fill_image = n.fill_i; n.fill_i = t; }
        { int32_t* t = o.ids; o.ids = h->ids_after; h->ids_after = n.ids; n.ids = t; }
        HIPCHK(h, hipMemcpyAsync((void*)h->d_state, n.state, IFX_CAM_STATE_BYTES, hipMemcpyDeviceToDevice, h->stream));
        for (int l = 0; l < IFX_NUM_PYRS; l++) HIPCHK(h, hipMemcpyAsync(prev.next_img[l], n.img[l], (size_t)(h->w >> l) * (h->h >> l), hipMemcpyDeviceToDevice, h->stream));
        o.valid = 1;
    } else {
        r = move(h->cams[(size_t)h->cur_cam], true);
        if (r) return r;
        h->cams[(size_t)h->cur_cam].valid = 1;
        if (h->cams[(size_t)cam].valid) {
            r = move(h->cams[(size_t)cam], false);
            if (r) return r;
        }
    }
    h->cams[(size_t)h->cur_cam].pred_root = h->pred_root;   // (who holds the parked prediction complete travels with it)
    h->pred_root = h->cams[(size_t)cam].valid ? h->cams[(size_t)cam].pred_root : h->pred_root;   // (a camera selected for the first time starts as a copy of the current one)
    h->cur_cam = cam;
    h->seg_counts_valid = 0;
    h->last_frame_slot = (h->tick & 1) ^ 1;   // (what the next frame side reads as its previous image now lives there)
    return IFX_OK;
}
// sharded map: the next frame takes `pose16` instead of tracking (the in_pose argument of the unsharded entry points); NULL clears it
extern "C" int ifx_owner_set_frame_pose(ifx_t* h, const float* pose16)
{
    if (!h) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_set_frame_pose: the handle was not created for a sharded map"; return IFX_E_STATE; }
    h->own_frame_pose_set = pose16 ? 1 : 0;
    if (pose16) memcpy(h->own_frame_pose, pose16, 64);
    return IFX_OK;
}
// sharded map, K streams: only `rank` tracks the frames to come (stream k tracked on GPU k: no tracker collective); a frame then starts with
// phase 310 (frame side everywhere, tracker on `rank`) and the exchange ifx_owner_exchange(h, 310, ...) -- the pose block, broadcast from `rank` (op 4 | rank << 8).
// -1: every rank tracks (replicated; the default).
extern "C" int ifx_owner_set_tracking_rank(ifx_t* h, int rank)
{
    if (!h || rank < -1 || rank >= (h ? h->own_g : 1)) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_set_tracking_rank: the handle was not created for a sharded map"; return IFX_E_STATE; }
    h->own_track_rank = rank;
    return IFX_OK;
}

// K streams over a sharded map, camera `cam` tracked by rank `tracking_rank` only: that rank's tracker of camera cam's NEXT frame, enqueued NOW on the handle's third stream,
// so that it runs under the other cameras' map phases instead of at the head of the frame.  A camera tracks against the prediction rendered at the end of its own last frame,
// and that prediction -- with the fill-in images, the last frame's intensity pyramid and the pose block -- is parked in the camera's context from the moment another camera is
// selected until this camera's next frame: the run reads the parked copies (a tracker instance of its own: state, pyramids, frame slot), so nothing the frames in between
// do can reach it.  When the frame itself arrives (camera cam selected, ifx_owner_process_frame_device / ifx_owner_frame_phase with exactly these device pointers, rank
// tracking_rank tracking) the parked pose block is committed instead of a tracker run: same inputs, same arithmetic, same pose (tests: test_config5_tracker_runs_ahead).
// The other ranks: no-op.  cam = -1: returns the number of frames whose tracker was taken from a run ahead so far.
extern "C" int ifx_owner_track_ahead(ifx_t* h, int cam, int tracking_rank, const uint8_t* d_rgb, const uint16_t* d_depth)
{
    if (!h) return IFX_E_INVALID;
    if (cam == -1) return h->cam_ahead_used;
    if (!h->own) { h->err = "ifx_owner_track_ahead: the handle was not created for a sharded map"; return IFX_E_STATE; }
    if (cam < 0 || (size_t)cam >= h->cams.size() || !d_rgb || !d_depth || tracking_rank < 0 || tracking_rank >= h->own_g) return IFX_E_INVALID;
    {   // the run reads the camera's parked prediction: it must be complete on the rank that runs (every rank decides this alike, so nobody is left alone in a collective)
        const int root = (cam == h->cur_cam) ? h->pred_root : h->cams[(size_t)cam].pred_root;
        if (root >= 0 && root != tracking_rank) {
            h->err = "ifx_owner_track_ahead: camera " + std::to_string(cam) + "'s prediction was reduced to rank " + std::to_string(root) + " only; rank " + std::to_string(tracking_rank) +
                     " holds partial sums (run one frame of the camera with ifx_owner_set_tracking_rank(-1) to all-reduce it)";
            return IFX_E_STATE;
        }
    }
    if (tracking_rank != h->cfg.rank) return IFX_OK;
    if (cam == h->cur_cam || !h->cams[(size_t)cam].valid) { h->err = "ifx_owner_track_ahead: the camera's context must be parked (select another camera first)"; return IFX_E_STATE; }
    if (h->lc_enable || !h->stream_c) { h->err = "ifx_owner_track_ahead: not available with the loop-closure detection on / on a one-stream handle"; return IFX_E_STATE; }
    CamCtx& cc = h->cams[(size_t)cam];
    cc.ahead_valid = 0;
    if (!cc.ahead_pose) HIPCHK(h, hipMalloc(&cc.ahead_pose, IFX_AHEAD_BYTES));
    if (!cc.ev_ahead) HIPCHK(h, hipEventCreateWithFlags(&cc.ev_ahead, hipEventDisableTiming));
    if (!h->ev_cam_parked) {   // (first use: the instance's buffers and events)
        hipEventCreateWithFlags(&h->ev_cam_ahead, hipEventDisableTiming);
        hipEventCreateWithFlags(&h->ev_cam_parked, hipEventDisableTiming);
    }
    // behind everything the main stream holds so far: the parking of the camera's context by ifx_camera_select above all
    HIPCHK(h, hipEventRecord(h->ev_cam_parked, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream_c, h->ev_cam_parked, 0));
    h->cam_side_stream = (h->opt_cam_side && h->opt_two_streams && h->stream_b) ? h->stream_b : nullptr;
    if (h->cam_side_stream) HIPCHK(h, hipStreamWaitEvent(h->cam_side_stream, h->ev_cam_parked, 0));
    h->cur = h->stream_c;
    int r;
    {
        StageTimer t(h, 0);
        r = ifx_tracker_camera_ahead(h, cam, d_rgb, d_depth);
    }
    h->cam_side_stream = nullptr;
    // the run's pose block into the camera's own parking place: the tracker instance is free for the next camera's run (they queue on the third stream).  The event
    // is the camera's own: the frame that takes this run must not wait for runs enqueued after it (one shared event, re-recorded behind every run, made every frame
    // wait for the run enqueued just before it: no overlap at all)
    if (!r) LAUNCH(h, "ahead_block", dim3(1), dim3(256), k_ahead_block, (uint32_t*)cc.ahead_pose, (uint32_t*)((char*)cc.ahead_pose + IFX_CAM_STATE_BYTES),
                   (const uint32_t*)h->d_cam_trk, (const uint32_t*)((const char*)h->d_cam_trk + IFX_DIAG_OFF));
    hipEventRecord(cc.ev_ahead, h->stream_c);
    hipEventRecord(h->ev_cam_ahead, h->stream_c);
    h->cur = h->stream;
    if (r) return r;
    cc.ahead_valid = 1; cc.ahead_rgb = d_rgb; cc.ahead_depth = d_depth;
    return IFX_OK;
}

// ---- sharded projection (ifx_map_sharded_phase): one frame in four phases, the caller exchanging the key
// images (element-wise unsigned min across ranks) between them.  Every rank is fed the same frames and masks.
extern "C" int ifx_set_shard(ifx_t* h, int rank, int nranks)
{
    if (!h || nranks < 1 || rank < 0 || rank >= nranks) return IFX_E_INVALID;
    ifx_vlist_reap(h);
    hs_invalidate_view(h);
    h->shard_rank = rank; h->shard_n = nranks;
    ifx_drop_tracked(h); h->hint_rgb = nullptr;
    return IFX_OK;
}

extern "C" int ifx_key_images(ifx_t* h, void** key_index, void** key_splat, void** key_ids, void** key_both, int64_t* n_pixels)
{
    if (!h) return IFX_E_INVALID;
    if (key_index) *key_index = h->key_index;
    if (key_splat) *key_splat = h->key_splat;
    if (key_ids) *key_ids = h->key_ids;
    if (key_both) *key_both = h->key_both;
    if (n_pixels) *n_pixels = h->P;
    return IFX_OK;
}

// the handle's HIP streams (hipStream_t as void*), so that a host can order its own work -- e.g. the RCCL exchange of the
// sharded mode -- on them instead of synchronising
extern "C" int ifx_stream_handles(ifx_t* h, void** main_stream, void** side_stream)
{
    if (!h) return IFX_E_INVALID;
    if (main_stream) *main_stream = (void*)h->stream;
    if (side_stream) *side_stream = (void*)h->stream_b;
    return IFX_OK;
}

extern "C" int ifx_sharded_frame_phase(ifx_t* h, int phase, const uint8_t* d_rgb, const uint16_t* d_depth)
{
    if (h && h->lc_enable) { h->err = "loop-closure detection is not available in the sharded frame phases"; return IFX_E_STATE; }
    if (!h || phase < 0 || phase > 3) return IFX_E_INVALID;
    const bool first = h->tick == 1;
    const int s = h->tick & 1;
    FrameSlot& f = h->slot[s];
    if (phase == 0) {
        if (!d_rgb || !d_depth) return IFX_E_INVALID;
        ifx_drop_tracked(h);
        // The compaction decision must be the SAME on every rank (slot numbers travel inside the exchanged keys): it is taken from the
        // result of the previous frame, which is identical on all ranks -- once the host has actually waited for it.
        if (h->ev_result) HIPCHK(h, hipEventSynchronize(h->ev_result));
        ifx_housekeeping(h);
        const int two = h->opt_two_streams;
        h->opt_two_streams = 0;                     // the frame side runs inline on the main stream in this mode
        int r = enqueue_frame_side(h, s, h->tick, d_rgb, d_depth, 0);
        h->opt_two_streams = two;
        if (r) return r;
        f.for_tick = -1;
        ifx_bind_slot(h, s);
        h->last_frame_slot = s;
        if (!first) { ifx_tracker_model_side(h); ifx_tracker_run_frame(h); }
    }
    int r = ifx_map_sharded_phase(h, phase, first);
    if (r) return r;
    if (phase == 3) {
        const int slot = h->n_traj % h->max_traj;
        LAUNCH(h, "frame_result", dim3(1), dim3(64), k_frame_result, h->d_state, h->h_result, h->d_traj + (size_t)slot * 16);
        hipEventRecord(f.released, h->stream);
        h->ev_result = f.released;
        h->seg_counts_valid = 1;
        h->n_traj++;
        h->tick++;
    }
    return IFX_OK;
}

// ---- spatially sharded map (ifx_config::n_ranks > 1): one frame in eight phases; after phase p (0..6) the caller reduces the buffers
// ifx_owner_exchange(p) lists across the ranks (instancefusion_amd/sharded.py: RCCL all-reduce; tests: the same reduction by hand)
int ifx_map_owner_phase(ifx* h, int phase, bool first_frame);
static int owner_frame_phase(ifx* h, int phase, const uint8_t* d_rgb, const uint16_t* d_depth, int src_kind);
// the rank exchange 5 reduces the prediction to (only that rank may track this camera's next frame); -1: all-reduced, every rank holds it
static int owner_pred_root(const ifx* h) { return (h->own_track_rank >= 0 && h->own_tracked_tick == h->tick && h->own_g > 1) ? h->own_track_rank : -1; }
static bool owner_lc_due(ifx* h);
extern "C" int ifx_owner_frame_phase(ifx_t* h, int phase, const uint8_t* d_rgb, const uint16_t* d_depth) { return h ? owner_frame_phase(h, phase, d_rgb, d_depth, 0) : IFX_E_INVALID; }
// One call = one frame of the sharded map: the eight phases with their exchanges enqueued by the library on the handle's main stream (ifx_comm.hip), no
// host synchronisation (the frame result lands in pinned memory behind the last kernel, as on the unsharded path).
static int owner_process_frame(ifx* h, const uint8_t* rgb, const uint16_t* depth, int src_kind)
{
    if (!ifx_comm_ready(h)) { h->err = "no communicator: ifx_owner_init_comm / ifx_owner_set_comm first"; return IFX_E_STATE; }
    if (h->own_track_rank >= 0) {   // K streams: this frame is tracked by one rank, the pose block broadcast to the others
        int r = owner_frame_phase(h, 310, rgb, depth, src_kind);
        if (r) return r;
        if ((r = ifx_comm_exchange(h, 310))) return r;
    }
    if (owner_lc_due(h))   // the local loop-closure detection: two more phases in front of the frame, two more collectives (16 + 84 bytes per pixel)
        for (int phase = 300; phase < 302; phase++) {
            int r = owner_frame_phase(h, phase, rgb, depth, src_kind);
            if (r) return r;
            if ((r = ifx_comm_exchange(h, phase))) return r;
        }
    for (int phase = 0; phase < 8; phase++) {
        int r = owner_frame_phase(h, phase, rgb, depth, src_kind);
        if (r) return r;
        if (phase < 6) { StageTimer t(h, 1); r = ifx_comm_exchange(h, phase); if (r) return r; }   // (the exchanges are billed to the map stage: ifx_stage_ms)
    }
    return IFX_OK;
}
extern "C" int ifx_owner_process_frame_device(ifx_t* h, const uint8_t* d_rgb, const uint16_t* d_depth, int64_t timestamp)
{
    (void)timestamp;
    if (!h || !d_rgb || !d_depth) return IFX_E_INVALID;
    return owner_process_frame(h, d_rgb, d_depth, 0);
}
// ElasticFusion::processFrame's signature on the sharded map: host pointers, H2D inside the call, one host synchronisation, the pose back
extern "C" int ifx_owner_process_frame(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp, float* out_pose16)
{
    (void)timestamp;
    if (!h || !rgb || !depth) return IFX_E_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(h->rgb_stage, rgb, (size_t)h->P * 3);
    memcpy(h->depth_stage, depth, (size_t)h->P * 2);
    int r = owner_process_frame(h, h->rgb_stage, h->depth_stage, 1);
    if (r) return r;
    r = ifx_sync(h);
    if (out_pose16) memcpy(out_pose16, h->h_result->pose, 64);
    return r;
}
// ElasticFusion::predict on the sharded map outside a frame (after ifx_map_upload / ifx_set_pose), exchanges included
extern "C" int ifx_owner_predict(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    if (!ifx_comm_ready(h)) { h->err = "no communicator: ifx_owner_init_comm / ifx_owner_set_comm first"; return IFX_E_STATE; }
    for (int step = 0; step < 3; step++) {
        int r = ifx_owner_predict_phase(h, step);
        if (r) return r;
        if (step < 2) { r = ifx_comm_exchange(h, 4 + step); if (r) return r; }
    }
    return IFX_OK;
}
// is the local loop-closure detection of this frame due?  (the host-side skip of enqueue_loop_closure_renders: while tick - timeDelta < 1 nothing the frames created can be old enough)
static bool owner_lc_due(ifx* h) { return h->lc_enable && !(h->tick == 1 && h->n_traj == 0) && (h->map_external || h->tick - h->cfg.time_delta >= 1); }
static int owner_frame_phase(ifx* h, int phase, const uint8_t* d_rgb, const uint16_t* d_depth, int src_kind)
{
    if (!h || ((phase < 0 || phase > 7) && phase != 300 && phase != 301 && phase != 310)) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_frame_phase: the handle was not created with n_ranks > 1"; return IFX_E_STATE; }
    if (h->lc_enable && (h->lc_cb || h->fern_cb)) { h->err = "on a sharded map the loop-closure DETECTION is available; the deformation callbacks are not"; return IFX_E_STATE; }
    if (h->own_ids_pending || h->oseg_state) { h->err = "ifx_owner_frame_phase: an id render / a segmentation call of this sharded map is waiting for its exchange (ifx_owner_ids_resume / ifx_owner_segmentation_resume)"; return IFX_E_STATE; }
    const bool first = h->tick == 1 && h->n_traj == 0;
    const int s = h->tick & 1;
    FrameSlot& f = h->slot[s];
    // With the detection on, a frame starts with two extra phases (300, 301: the two renders of EF/ElasticFusion.cpp:453 / :519-526 at the tracked pose, pre-fusion map);
    // phase 300 then carries the frame side and the tracker, and phase 0 runs the model-to-model tracker on the exchanged renders before its index projection.
    const bool lc_due = owner_lc_due(h);
    if ((phase == 300 || phase == 301) && !lc_due) return IFX_OK;   // (nothing to render: the caller's exchange list for it is empty too)
    if (phase == 310 && h->own_track_rank < 0) return IFX_OK;   // every rank tracks: nothing to hand over
    const bool starts_frame = phase == 310 || ((phase == 300 || phase == 0) && h->own_tracked_tick != h->tick);
    if (starts_frame) {
        if (!d_rgb || !d_depth) return IFX_E_INVALID;
        // Bounded run-ahead, as on the unsharded path (enqueue_frame): the previous frame's RESULT before this frame goes onto the queues
        if (h->opt_pace && h->ev_result && h->tick > 1) HIPCHK(h, hipEventSynchronize(h->ev_result));
        const bool tracks = h->own_track_rank < 0 || h->own_track_rank == h->cfg.rank;
        if (!first && !h->own_frame_pose_set && h->pred_root >= 0 && h->pred_root != h->own_track_rank) {
            // the prediction this frame tracks against was reduced to ONE rank (exchange 5 with a tracking rank) and that rank is not (the only) one tracking now: the others hold
            // partial sums.  Every rank sees the same two numbers and refuses alike -- a pose tracked from a partial block would be broadcast to everybody without an error
            h->err = "the live camera's prediction was reduced to rank " + std::to_string(h->pred_root) + " only, and " +
                     (h->own_track_rank < 0 ? std::string("every rank") : "rank " + std::to_string(h->own_track_rank)) +
                     " is to track this frame: keep the tracking rank, or give this one frame its pose (ifx_owner_set_frame_pose) -- its own prediction is then all-reduced";
            return IFX_E_STATE;
        }
        // one-frame look-ahead (ifx_hint_next_frame_device before the previous frame): the frame side of this frame ran on the side stream under the previous frame's
        // phases, and its tracker -- which reads only its slot, the exchanged prediction and the pose -- was enqueued right behind that frame, its result parked
        const bool prepared = f.for_tick == h->tick && f.src_rgb == d_rgb && f.src_depth == d_depth && src_kind == 0;
        const bool tracked = prepared && h->tracked_ahead == h->tick && !h->own_frame_pose_set && tracks && !first;
        ifx_drop_tracked(h);
        ifx_housekeeping(h);                        // local and independent: ids are creation numbers, a compaction renumbers nothing the other ranks see
        // this camera's tracker ran ahead on the third stream, from the camera's parked context (ifx_owner_track_ahead), on exactly these images: its pose block is the
        // frame's, and so is its frame side -- raw images, filtered depth, pyramids sit in the camera's own slot: the frame binds it instead of computing them again
        const bool ahead = !first && !h->own_frame_pose_set && tracks && !h->cams.empty() && h->cams[(size_t)h->cur_cam].ahead_valid &&
                           h->cams[(size_t)h->cur_cam].ahead_rgb == (const void*)d_rgb && h->cams[(size_t)h->cur_cam].ahead_depth == (const void*)d_depth && src_kind == 0;
        int bound_slot = s;
        if (!ahead && !h->cams.empty() && h->cams[(size_t)h->cur_cam].ahead_valid && h->cams[(size_t)h->cur_cam].ev_ahead)   // a run this frame does not take may still be reading the camera's context
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->cams[(size_t)h->cur_cam].ev_ahead, 0));
        if (ahead) {
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->cams[(size_t)h->cur_cam].ev_ahead, 0));
            bound_slot = 3 + h->cur_cam;
        } else if (!prepared) {
            // (a frame announced wrongly: the side stream may still be running the frame side of what was announced, into this slot and the shared SO(3) sums)
            if (f.for_tick == h->tick && f.ready && h->opt_two_streams) HIPCHK(h, hipStreamWaitEvent(h->stream, f.ready, 0));
            if (h->last_frame_slot >= 3) {   // the frame before this one ran on a camera's run-ahead slot and no camera was selected since: its intensity pyramid is this frame's "previous image"
                for (int l = 0; l < IFX_NUM_PYRS; l++)
                    HIPCHK(h, hipMemcpyAsync(h->slot[(size_t)(s ^ 1)].next_img[l], h->slot[(size_t)h->last_frame_slot].next_img[l], (size_t)(h->w >> l) * (h->h >> l), hipMemcpyDeviceToDevice, h->stream));
            }
            const int two = h->opt_two_streams;
            h->opt_two_streams = 0;
            int r = enqueue_frame_side(h, s, h->tick, d_rgb, d_depth, src_kind);
            h->opt_two_streams = two;
            if (r) return r;
        } else HIPCHK(h, hipStreamWaitEvent(h->stream, f.ready, 0));
        f.for_tick = -1;
        ifx_bind_slot(h, bound_slot);
        h->last_frame_slot = bound_slot;
        h->own_need_decide = 0;
        if (!first && h->own_frame_pose_set) {   // an external pose replaces tracking (EF/ElasticFusion.cpp:357-360), on every rank alike
            float* slot = h->d_scratch + 2 * 16;
            HIPCHK(h, hipMemcpyAsync(slot, h->own_frame_pose, 64, hipMemcpyHostToDevice, h->stream));
            ifx_tracker_external_pose(h, slot, 1.0f);
            { int r = hold_last_image(h, bound_slot, h->stream); if (r) return r; }   // (the main stream is behind this frame's side: it waited for the slot above, or ran it itself)
        } else if (!first && tracks) {   // replicated (every rank holds the exchanged prediction), or on the one tracking rank
            StageTimer t(h, 0);
            if (ahead) {
                const char* ap = (const char*)h->cams[(size_t)h->cur_cam].ahead_pose;   // pose block + the run's diagnostics (the frame result reports THIS camera's tracker)
                LAUNCH(h, "ahead_block", dim3(1), dim3(256), k_ahead_block, (uint32_t*)h->d_state, (uint32_t*)((char*)h->d_state + IFX_DIAG_OFF), (const uint32_t*)ap,
                       (const uint32_t*)(ap + IFX_CAM_STATE_BYTES));
                h->own_need_decide = 1;   // (the view-list decision for the committed pose: phase 0)
                h->cam_ahead_used++;
            } else if (tracked) ifx_tracker_commit(h);
            else { ifx_tracker_model_side(h); const int tr_ = ifx_tracker_run_frame(h); if (tr_) return tr_; }
            if (!h->cams.empty()) h->cams[(size_t)h->cur_cam].ahead_valid = 0;   // (a run ahead is for the camera's NEXT frame: whatever this frame was, it is spent)
        } else if (!first) h->own_need_decide = 1;   // the pose arrives with exchange 310: the view-list decision follows it (phase 0)
        { int r = ifx_enqueue_hinted_frame_side(h); if (r) return r; }   // (a tracker enqueue consumed the hint already: no-op)
        h->own_frame_pose_set = 0;
        h->own_tracked_tick = h->tick;
        if (phase == 310) return IFX_OK;   // the pose block travels from the tracking rank next (exchange 310)
    }
    if (phase == 300 && h->own_tracked_tick == h->tick) {
        {
            if (h->lc_pending) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_lc_done, 0)); h->lc_pending = 0; }
            ifx_tracker_m2m_begin(h);
        }
    }
    if (phase == 0 && h->lc_enable && !first) {
        if (lc_due) { int r = ifx_tracker_loop_closure(h); if (r) return r; h->lc_event_valid = 0; }   // replicated on the exchanged act_* / old_* images: the same verdict on every rank
        else LAUNCH(h, "lc_idle", dim3(1), dim3(64), k_lc_idle, h->d_state, h->h_lc);
    }
    int r;
    {
        StageTimer t(h, 1);
        r = ifx_map_owner_phase(h, phase, first);
    }
    if (r) return r;
    if (phase == 5) h->pred_root = owner_pred_root(h);   // who will hold this prediction complete once exchange 5 has run: set where the phase is ENQUEUED (ifx_owner_exchange only describes buffers)
    if (phase == 7) {   // after the vote mass of phase 6 was summed across the ranks: the frame result every rank reads its whetherDoSegmentation decision from
        const int slot = h->n_traj % h->max_traj;
        LAUNCH(h, "frame_result", dim3(1), dim3(64), k_frame_result, h->d_state, h->h_result, h->d_traj + (size_t)slot * 16);
        hipEventRecord(f.released, h->stream);
        h->ev_result = f.released;
        h->seg_counts_valid = first ? 0 : 1;
        h->n_traj++;
        h->tick++;
        // the announced next frame: its tracker reads only its slot, the prediction just exchanged and the pose -- enqueued now, parked (DevState::spec_*), so that the
        // GPU has work while the host decides about segmentation (enqueue_frame does the same on the unsharded path).  Replicated tracking only: with camera contexts /
        // a tracking rank the frames of a set interleave cameras (ifx_owner_track_camera_ahead is that case's look-ahead)
        FrameSlot& nf = h->slot[h->tick & 1];
        if (h->opt_track_ahead && h->opt_two_streams && nf.for_tick == h->tick && h->tick > 1 && h->own_track_rank < 0 && !h->lc_enable && h->cams.empty()) {
            ifx_bind_slot(h, h->tick & 1);
            HIPCHK(h, hipStreamWaitEvent(h->stream, nf.ready, 0));
            {
                StageTimer t(h, 0);
                ifx_tracker_model_side(h, 1);
                const int tr_ = ifx_tracker_run_frame(h, 0);
                if (tr_) return tr_;
            }
            ifx_bind_slot(h, s);
            h->tracked_ahead = h->tick;
        }
    }
    return IFX_OK;
}
// ElasticFusion::predict on a sharded map outside a frame (after an upload / set_pose): step 0 = local raster (then exchange as after
// phase 4), step 1 = owned winners (then exchange as after phase 5), step 2 = fill-in
extern "C" int ifx_owner_predict_phase(ifx_t* h, int step)
{
    if (!h || step < 0 || step > 2) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_predict_phase: the handle was not created with n_ranks > 1"; return IFX_E_STATE; }
    ifx_drop_tracked(h);
    const int r = ifx_map_owner_phase(h, 104 + step, false);
    if (!r && step == 1) h->pred_root = owner_pred_root(h);   // (the exchange behind this step is exchange 5)
    return r;
}
// what to reduce after phase `phase`: ptrs[k] (device), bytes[k], ops[k] (0: element-wise minimum of unsigned 64-bit words, 1: sum of
// 32-bit words -- the supports are disjoint, so the sum is a bitwise merge); returns the number of buffers (0: nothing to exchange)
extern "C" int ifx_owner_exchange(ifx_t* h, int phase, void** ptrs, int64_t* bytes, int32_t* ops, int max_n)
{
    if (!h || !ptrs || !bytes || !ops) return IFX_E_INVALID;
    const bool first = h->tick == 1 && h->n_traj == 0;   // first frame in flight (phase 6 has not run yet); an uploaded map makes every frame a regular one
    const size_t P = (size_t)h->P;
    int n = 0;
    auto add = [&](void* p, size_t b, int op) { if (n < max_n) { ptrs[n] = p; bytes[n] = (int64_t)b; ops[n] = op; } n++; };
    switch (phase) {
    case 0: if (!first) add(h->key_index, P * 8 + 8, h->opt_own_key_rs ? 6 : 0); break;                  // + the lowest live creation number (the reference's "surfel 0", ifx_map.hip FIRST_LIVE)
    case 1: if (!first) add(h->assoc_key, (size_t)((h->w + 1) / 2) * ((h->h + 1) / 2) * 8, 0); break;   // the best owned candidate of every measurement pixel (distance | window position)
    case 2: if (!first) add(h->key_index, P * 8 + 8, h->opt_own_key_rs ? 6 : 0); break;                                    // (the word is the reduced one of exchange 0: a MIN of equal values)
    case 3: if (!first) add(h->index_tap, P * 16, 1); break;
    case 4:                                                                                         // [key_splat | key_ids | word] (key_both was folded into them by k_merge_both)
        if (h->own_ids_lat) add(h->key_splat, (P + (size_t)ifx_own_lattice(h)) * 8 + 8, 0);         // option own_lazy_ids: [key_splat | the id keys of the sampled lattice | word]
        else add(h->key_splat, P * 16 + 8, 0);
        break;
    case 5:   // [pred_conf | pred_normal | pred_image | pred_inst | pred_time | tail: vote mass of the owned surfels under the id image]
        if (owner_pred_root(h) >= 0) {
            // K streams, camera k tracked by rank k only: the prediction rendered at the end of camera k's frame has ONE consumer, rank k's tracker -- a reduction to that
            // rank (op 5 | root << 8: ncclReduce, (G - 1) / G of the block per link instead of the all-reduce's 2 (G - 1) / G); the 16-byte tail -- the vote mass every
            // rank's whetherDoSegmentation decision needs -- still goes to everybody.  On the other ranks the block holds their own partial sums and is never read.
            add(h->pred_conf, h->pred_bytes - (size_t)h->P * 16 - 16, 5 | (h->own_track_rank << 8));
            add(h->pred_tail, 16, 1);
        } else add(h->pred_conf, h->pred_bytes - (size_t)h->P * 16, 1);   // (who holds the block complete afterwards is remembered where phase 5 is enqueued: h->pred_root)
        break;
    case 6: break;
    case 310: if (h->own_track_rank >= 0 && !first) add((void*)h->d_state, IFX_CAM_STATE_BYTES, 4 | (h->own_track_rank << 8)); break;   // the tracked pose block, broadcast from the tracking rank
    case 300: if (owner_lc_due(h)) add(h->key_splat, P * 16, 0); break;                       // the detection's two renders: [key_splat (ACTIVE) | key_ids (INACTIVE)]
    case 301: if (owner_lc_due(h)) add(h->act_vertex, 2 * h->lc_half, 1); break;            // [act_* | old_*]: the owners' winners of both
    case 200:   // a segmentation call on a sharded map is waiting at an exchange point (ifx_owner_segmentation_begin / _resume), or an id render of the shards (ifx_owner_ids_begin)
        if (h->own_ids_pending) { add(h->key_ids, P * 8, 0); break; }                              // the whole id image's keys: MIN
        switch (h->oseg_pending) {
        case 1: add(h->d_bbox, (size_t)(96 + h->oseg_nm) * 4 * 4, 2); break;                       // boxes, maxima negated: MIN of 32-bit words
        case 2: add(h->d_pdm, P * 2, 1); break;                                                     // model depth under the camera: disjoint supports
        case 3: add(h->d_inst_stats, 96 * 4, 3); add(h->d_inst_stats + 96, 96 * 4, 1); break;      // per-instance maximum (MAX) and sum (SUM) of the vote counters
        default: break;
        }
        break;
    default: break;
    }
    return n;
}
// The whole id image of a sharded map whose frames draw the sampled lattice only (option own_lazy_ids), for callers that run the exchanges themselves:
// _begin enqueues the shard's id render and returns 1 (then: the exchange ifx_owner_exchange(h, 200, ...) lists, _resume) or 0 when the image is whole already.
extern "C" int ifx_owner_ids_begin(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    if (!h->own) { h->err = "ifx_owner_ids_begin: the handle was not created for a sharded map"; return IFX_E_STATE; }
    if (h->oseg_state) { h->err = "ifx_owner_ids_begin: a segmentation call is in flight"; return IFX_E_STATE; }
    return ifx_owner_ids_begin_impl(h);
}
extern "C" int ifx_owner_ids_resume(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    return ifx_owner_ids_resume_impl(h);
}
extern "C" int ifx_owner_of(const float* xyz, int n, int n_ranks, int32_t* out)
{
    if (!xyz || !out || n < 0) return IFX_E_INVALID;
    for (int i = 0; i < n; i++) out[i] = ifx_owner_of_point(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], n_ranks);
    return IFX_OK;
}
// creation numbers of the live surfels, in the order of ifx_map_download (which compacts): merging the shards of a sharded map by them
// gives the unsharded map
extern "C" int ifx_map_seq(ifx_t* h, uint32_t* out, int max_n)
{
    if (!h || !out) return IFX_E_INVALID;
    int r = ifx_compact(h);
    if (r) return r;
    DevState hs;
    HIPCHK(h, hipMemcpy(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost));
    const int n = std::min(hs.count, max_n);
    HIPCHK(h, hipMemcpy(out, h->seq, (size_t)n * 4, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int ifx_enqueue_frame_device(ifx_t* h, const uint8_t* d_rgb, const uint16_t* d_depth, int64_t timestamp, const float* in_pose16, float weight_mult)
{
    (void)timestamp;
    if (!h || !d_rgb || !d_depth) return IFX_E_INVALID;
    ifx_housekeeping(h);   // before this frame's map passes; a tracker run that is already queued does not care about slot numbers
    return enqueue_frame(h, d_rgb, d_depth, 0, in_pose16, weight_mult);
}

// One-frame look-ahead for streams whose next frame is already resident (log replay, the benchmark): the frame
// side of the NEXT frame is enqueued on the side stream now, so it runs under the current frame's tracking and
// map passes.  The next ifx_enqueue_frame_device call must pass the same pointers (otherwise the slot is
// simply recomputed).  Call it after enqueueing the current frame.
extern "C" int ifx_prefetch_frame_device(ifx_t* h, const uint8_t* d_rgb_next, const uint16_t* d_depth_next)
{
    if (!h || !d_rgb_next || !d_depth_next) return IFX_E_INVALID;
    if (h->tick == 1 || !h->opt_two_streams) return IFX_OK;   // nothing to overlap with
    return enqueue_frame_side(h, h->tick & 1, h->tick, d_rgb_next, d_depth_next, 0);
}

// Same look-ahead, announced BEFORE the current frame is enqueued: the library places the next frame's image-only
// work itself, behind the coarse pyramid levels of the current frame's tracker, where the GPU is least busy.
extern "C" int ifx_hint_next_frame_device(ifx_t* h, const uint8_t* d_rgb_next, const uint16_t* d_depth_next)
{
    if (!h || !d_rgb_next || !d_depth_next) return IFX_E_INVALID;
    h->hint_rgb = d_rgb_next; h->hint_depth = d_depth_next; h->hint_kind = 0;
    return IFX_OK;
}

// The same announcement with HOST pointers, for the reference-shaped entry (ifx_process_frame): a log reader or a camera queue that already holds the next frame hands it
// over BEFORE it calls ifx_process_frame for the current one.  The buffers are borrowed for this call only: the frame is copied into a pinned staging pair of its own
// (by frame parity) here, its transfer and image-only work go to the side stream from inside the current frame's enqueue, and its tracker is parked behind the current
// frame -- exactly the resident path's look-ahead, plus the transfer.  The next ifx_process_frame call must pass the SAME pointers (anything else: the announcement is
// ignored and the frame is staged and computed as ever).  Single-stream, unsharded handles.
extern "C" int ifx_hint_next_frame(ifx_t* h, const uint8_t* rgb_next, const uint16_t* depth_next)
{
    if (!h || !rgb_next || !depth_next) return IFX_E_INVALID;
    if (h->own || !h->cams.empty()) { h->err = "ifx_hint_next_frame: single-stream, unsharded handles (use ifx_hint_next_frame_device)"; return IFX_E_STATE; }
    if (!h->opt_two_streams || !h->stream_b) return IFX_OK;   // nothing to overlap with
    const int p = (h->tick + 1) & 1;   // (this pair's last transfer belongs to frame tick - 1, whose pose a call has returned: complete)
    if (!h->hint_stage_rgb[p] || !h->hint_stage_depth[p]) {   // the pair is allocated together or not at all (a half-allocated pair would be written through a null pointer by the next call)
        if (h->hint_stage_rgb[p]) { (void)hipHostFree(h->hint_stage_rgb[p]); h->hint_stage_rgb[p] = nullptr; }
        if (h->hint_stage_depth[p]) { (void)hipHostFree(h->hint_stage_depth[p]); h->hint_stage_depth[p] = nullptr; }
        uint8_t* sr = nullptr; uint16_t* sd = nullptr;
        if (hipHostMalloc((void**)&sr, (size_t)h->P * 3, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void**)&sd, (size_t)h->P * 2, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (sr) (void)hipHostFree(sr);
            h->err = "ifx_hint_next_frame: no pinned memory for the staging pair";
            return IFX_E_HIP;
        }
        h->hint_stage_rgb[p] = sr; h->hint_stage_depth[p] = sd;
    }
    memcpy(h->hint_stage_depth[p], depth_next, (size_t)h->P * 2);
    memcpy(h->hint_stage_rgb[p], rgb_next, (size_t)h->P * 3);
    h->hinted_src_rgb[p] = rgb_next; h->hinted_src_depth[p] = depth_next; h->hinted_tick[p] = h->tick + 1;
    h->hint_rgb = h->hint_stage_rgb[p]; h->hint_depth = h->hint_stage_depth[p]; h->hint_kind = 1;
    return IFX_OK;
}

extern "C" int ifx_view_list_stats(ifx_t* h, int32_t* out4)
{
    if (!h || !out4) return IFX_E_INVALID;
    DevState hs;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost));
    out4[0] = (int32_t)hs.vl_n[0]; out4[1] = (int32_t)hs.vl_n[1]; out4[2] = hs.vl_scans; out4[3] = hs.vl_age;
    return IFX_OK;
}

extern "C" int ifx_sync(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream_b));
    if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
    ktime_flush(h);
    stage_flush(h);
    if (h->h_result->overflow) { h->err = "surfel store capacity exceeded"; return IFX_E_CAPACITY; }
    return IFX_OK;
}
// Levels of the tracker's pyramid that the persistent kernel (option gn_persist) could not finish -- a meeting of its blocks did not happen: the grid was not co-resident --
// and that its block 0 re-ran alone inside the same frame, with the same result (k_gn_level / gn_level_solo).  A count since the handle was created, not an error:
// results are unaffected, such a frame is just slower; a host that sees it grow should set gn_persist to 0.  (No counterpart in the reference: diagnostics.)
extern "C" int ifx_tracker_fallbacks(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    int r = ifx_sync(h);
    return r ? r : h->h_result->gn_timeout;
}

// Run-time guard of the tracker's exact sums (ifx_track.hip range_exceeded7): how often, since ifx_create, a reduction's diagonal total was found beyond half the range in
// which every addition of the 29 (SO(3): 11) sums is exact -- i.e. how often "the sums do not depend on the order of the atomics, and equal the oracle's" was NOT guaranteed
// (frames of saturated edges at near range; the images this path sees in tests and bench.py are 2^7 below it).  0 = every pose so far is the pose of the fixed arithmetic.
// Summed over every tracker instance of the handle (frame tracker, its SO(3) pre-alignment in the frame slots, model-to-model tracker, camera run-ahead tracker).
// (No counterpart in the reference: its f32 tree sums have no exactness to lose.)  Waits for the frames in flight.
extern "C" int ifx_hot_records_stale(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int v = 0;
    HIPCHK(h, hipMemcpy(&v, &h->d_state->hot_stale, sizeof(int), hipMemcpyDeviceToHost));
    return v;
}
extern "C" int ifx_tracker_range_exceeded(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    int r = ifx_sync(h);
    if (r) return r;
    if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
    if (h->stream_b) HIPCHK(h, hipStreamSynchronize(h->stream_b));
    std::vector<const DevState*> sts;
    sts.push_back(h->d_state);
    if (h->d_m2m) sts.push_back(h->d_m2m);
    if (h->d_cam_trk) sts.push_back(h->d_cam_trk);
    for (const FrameSlot& f : h->slot) if (f.so3) sts.push_back(f.so3);
    long long total = 0;
    for (const DevState* d : sts) {
        int v = 0;
        HIPCHK(h, hipMemcpy(&v, (const char*)d + offsetof(DevState, range_exceeded), sizeof(int), hipMemcpyDeviceToHost));
        total += v;
    }
    return (int)std::min<long long>(total, 0x7FFFFFFF);
}

// End of a synchronous ifx_process_frame: the frame just enqueued is complete.  With the NEXT frame announced (ifx_hint_next_frame) its side and its parked tracker are on
// the queues behind this frame: the call waits for the frame's own end (the event the resident path paces itself with), not for the streams.
static int frame_sync(ifx* h)
{
    if (h->tracked_ahead == h->tick && h->ev_result && !h->lc_enable && !h->opt_kernel_timing && !h->opt_stage_timing) {
        HIPCHK(h, hipEventSynchronize(h->ev_result));
        if (h->h_result->overflow) { h->err = "surfel store capacity exceeded"; return IFX_E_CAPACITY; }
        return IFX_OK;
    }
    return ifx_sync(h);
}

extern "C" int ifx_process_frame(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp, const float* in_pose16, float weight_mult, float* out_pose16)
{
    return ifx_process_frame_ex(h, rgb, depth, timestamp, nullptr, in_pose16, weight_mult, 0, out_pose16);
}

// The full argument list of ElasticFusion::processFrame (EF/ElasticFusion.h:75-82).  inst_table (smallInstanceTable, 96 x 5) has one consumer in
// the reference, Ferns::findFrame (EF/ElasticFusion.cpp:468), i.e. host code above this boundary (host/ifx_ferns.hpp): accepted, not read.
extern "C" int ifx_process_frame_ex(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp, const int32_t* inst_table, const float* in_pose16,
                                    float weight_mult, int bootstrap, float* out_pose16)
{
    (void)timestamp; (void)inst_table;
    if (!h || !rgb || !depth) return IFX_E_INVALID;
    // caller buffers are borrowed for the call only (EF/ElasticFusion.cpp:280-281 copies them too).
    // The call returns when the frame's POSE is known -- read back right behind the tracker -- and the frame's map passes finish under whatever the caller does next
    // (typically: the next call's copy of its frame into the staging buffers, 0.1 ms during which the device used to idle).  Everything that looks at the map, the
    // images or the frame result afterwards waits for the frame as on the enqueue path (ifx_enqueue_frame_device has always returned earlier than this); a surfel store
    // that filled up is reported by the NEXT call (or by ifx_sync / any whole-map consumer).  Not with the loop-closure detection on (a deformation may still adopt
    // another pose) and not for the first frame.  OPT-IN (option "host_entry_async" 1): the compaction decision a synchronous call takes right behind its frame is
    // taken at the start of the NEXT ifx_process_frame instead (from the same numbers) -- equivalent for a host whose loop is ifx_process_frame after
    // ifx_process_frame, but a camera switch, an upload or a segmentation call in between would see the store before that compaction instead of after it.
    // a frame announced by ifx_hint_next_frame: its data sits in its parity's staging pair, its frame side is on the side stream since the previous frame's enqueue (and its
    // tracker parked behind that frame): nothing to stage, nothing to wait for here
    const int par = h->tick & 1;
    const bool pre = h->hinted_tick[par] == h->tick && h->hinted_src_rgb[par] == (const void*)rgb && h->hinted_src_depth[par] == (const void*)depth && h->prestaged_tick == h->tick &&
                     h->slot[par].for_tick == h->tick && h->slot[par].src_rgb == (const void*)h->hint_stage_rgb[par] && !h->own;
    h->hinted_tick[par] = -1;
    if (pre) {
        h->want_early_pose = 0;
        h->n_host_hinted++;
        if (h->housekeeping_due) {   // (option host_entry_async: the decision an early-returning call left for this one)
            if (h->ev_result) HIPCHK(h, hipEventSynchronize(h->ev_result));
            ifx_housekeeping(h);
            h->housekeeping_due = 0;
        }
        int r = enqueue_frame(h, h->hint_stage_rgb[par], h->hint_stage_depth[par], 1, in_pose16, weight_mult, bootstrap);
        if (r) return r;
        r = frame_sync(h);
        if (out_pose16) memcpy(out_pose16, h->h_result->pose, 64);
        if (r) return r;
        ifx_housekeeping(h);
        h->housekeeping_due = 0;
        return 0;
    }
    const bool can_early = h->opt_host_entry_async && !h->lc_enable && !h->in_fern_cb && h->tick > 1 && h->cams.empty() && !in_pose16;   // (plain tracked frames of a single stream)
    // (the staging buffers are free: their last copy to the device ran in front of a tracker whose pose a previous call has waited for -- or behind a full synchronisation)
    if (!can_early) HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(h->depth_stage, depth, (size_t)h->P * 2);
    h->late_rgb_src = rgb;   // (copied by enqueue_frame_side, behind the depth transfer and the bilateral filter's launch)
    bool early = false;
    if (can_early && h->opt_two_streams && h->stream_b && !h->hint_rgb && h->slot[h->tick & 1].for_tick != h->tick) {
        // this frame's copy-in and image-only work (0.2 ms) go to the side stream NOW, under the previous frame's map passes: they need nothing of that frame but the
        // intensity pyramid its own side left on this stream.  The wait below used to come first, and the device idled through the transfer and the filter.
        int r = enqueue_frame_side(h, h->tick & 1, h->tick, h->rgb_stage, h->depth_stage, 1);
        if (r) { h->late_rgb_src = nullptr; return r; }
        h->prestaged_tick = h->tick;
    }
    if (can_early) {
        // the frame before this one is complete from here on: its result is final, and the housekeeping decision a synchronous call would have taken right behind it
        // is taken now, from the same numbers, at the same place in the stream -- slot numbers do not depend on timing
        if (h->ev_result) HIPCHK(h, hipEventSynchronize(h->ev_result));
        if (h->housekeeping_due) { ifx_housekeeping(h); h->housekeeping_due = 0; }
        // near the capacity the frame itself must report a store that fills up: the synchronous form
        early = !h->h_result->overflow && (long long)h->h_result->count + 2LL * h->P <= (long long)h->cap;
    }
    h->want_early_pose = early ? 1 : 0;
    h->early_pose_valid = 0;
    int r = enqueue_frame(h, h->rgb_stage, h->depth_stage, 1, in_pose16, weight_mult, bootstrap);
    if (h->late_rgb_src) { h->late_rgb_src = nullptr; if (!r) { h->err = "ifx_process_frame: the frame side did not take the colour image (internal)"; r = IFX_E_STATE; } }
    h->want_early_pose = 0;
    if (r) return r;
    if (early && h->early_pose_valid) {
        HIPCHK(h, hipEventSynchronize(h->ev_pose_early));
        if (out_pose16) memcpy(out_pose16, h->h_pose_early, 64);
        h->housekeeping_due = 1;
        return 0;
    }
    r = frame_sync(h);
    if (out_pose16) memcpy(out_pose16, h->h_result->pose, 64);
    if (r) return r;
    ifx_housekeeping(h);
    h->housekeeping_due = 0;
    return 0;
}

// Tombstones pile up: when the last frame result the host has seen says that more than 1/8 of the
// slots are dead, or that the capacity gets tight, the map is compacted (order-preserving) and ids_after re-rendered.
// The decision uses the pinned frame result as it is -- possibly a frame or two old on the asynchronous path: it is a
// heuristic about WHEN, not about what; results do not depend on it (tests: test_full_size_properties, housekeeping mode
// of test_lookahead_equivalence).
int ifx_housekeeping(ifx* h)
{
    const FrameResult* r = h->h_result;
    if (h->opt_compact_every_frame || h->tick <= 2 || r->n_dead <= 0) return IFX_OK;
    if (r->n_dead > r->count / std::max(h->opt_compact_divisor, 1) || r->count > h->cap - h->P) {
        if (h->last_compact_tick == h->tick) return IFX_OK;
        h->last_compact_tick = h->tick;
        return ifx_compact_enqueue(h, 1);
    }
    return IFX_OK;
}

extern "C" int ifx_set_frame(ifx_t* h, const uint8_t* rgb, const uint16_t* depth)
{
    if (!h || !rgb || !depth) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    HIPCHK(h, hipMemcpyAsync(h->rgb, rgb, (size_t)h->P * 3, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->depth_raw, depth, (size_t)h->P * 2, hipMemcpyHostToDevice, h->stream));
    ifx_preprocess(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return IFX_OK;
}

extern "C" int ifx_get_pose(ifx_t* h, float* out)
{
    if (!h || !out) return IFX_E_INVALID;
    int r = ifx_sync(h);
    DevState hs;
    HIPCHK(h, hipMemcpy(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost));
    memcpy(out, hs.pose, 64);
    return r;
}
extern "C" int ifx_tick(ifx_t* h) { return h ? h->tick : IFX_E_INVALID; }

extern "C" int ifx_trajectory(ifx_t* h, float* out, int max_frames)
{
    if (!h || !out) return IFX_E_INVALID;
    // the log is a ring of max_traj frames: the LAST min(frames processed, max_traj, max_frames) poses, oldest first
    const int n = std::min(std::min(h->n_traj, max_frames), h->max_traj);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int first = (h->n_traj - n) % h->max_traj, head = std::min(n, h->max_traj - first);
    if (head > 0) HIPCHK(h, hipMemcpy(out, h->d_traj + (size_t)first * 16, (size_t)head * 64, hipMemcpyDeviceToHost));
    if (n > head) HIPCHK(h, hipMemcpy(out + (size_t)head * 16, h->d_traj, (size_t)(n - head) * 64, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int ifx_tracker_diag(ifx_t* h, float* diag8)
{
    if (!h || !diag8) return IFX_E_INVALID;
    int r = ifx_sync(h);
    memcpy(diag8, h->h_result->diag, 32);
    return r;
}

extern "C" int ifx_stage_ms(ifx_t* h, float* ms4, int reset)
{
    if (!h || !ms4) return IFX_E_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    stage_flush(h);
    for (int k = 0; k < 4; k++) ms4[k] = (float)h->stage_ms[k];
    if (reset) for (int k = 0; k < 4; k++) h->stage_ms[k] = 0;
    return IFX_OK;
}
// Superpixels run ahead of a segmentation call on the side stream (ifx_superpixel_ahead): how many runs, how many a call then used, and -- with stage timing on --
// the device time of the runs (ms, side stream; the "instance" entry of ifx_stage_ms is the main-stream span of the calls and does not contain it)
extern "C" int ifx_lookahead_stats(ifx_t* h, int32_t* out3, int reset)
{
    if (!h || !out3) return IFX_E_INVALID;
    out3[0] = h->n_side_prepared; out3[1] = h->n_tracked_ahead; out3[2] = h->n_host_hinted;
    if (reset) h->n_side_prepared = h->n_tracked_ahead = h->n_host_hinted = 0;
    return IFX_OK;
}

extern "C" int ifx_superpixel_ahead_stats(ifx_t* h, float* ms, int32_t* runs, int32_t* used, int reset)
{
    if (!h) return IFX_E_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->stream_b) HIPCHK(h, hipStreamSynchronize(h->stream_b));
    stage_flush(h);
    if (ms) *ms = (float)h->stage_ms[4];
    if (runs) *runs = h->slic_ahead_runs;
    if (used) *used = h->slic_ahead_used;
    if (reset) { h->stage_ms[4] = 0; h->slic_ahead_runs = 0; h->slic_ahead_used = 0; }
    return IFX_OK;
}

extern "C" int ifx_kernel_ms(ifx_t* h, const char* kernel, float* avg_ms, int* launches)
{
    if (!h || !kernel) return IFX_E_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    ktime_flush(h);
    std::string s(kernel);
    if (s == "__reset__") { for (auto& k : h->ktimes) k = KernelTiming(); return IFX_OK; }
    if (s == "__list__") {   // debugging aid: prints the table to stderr
        for (size_t i = 0; i < h->knames.size(); i++)
            fprintf(stderr, "%-20s launches %6d  avg %.4f ms  total %.3f ms\n", h->knames[i].c_str(), h->ktimes[i].launches,
                    h->ktimes[i].launches ? h->ktimes[i].total_ms / h->ktimes[i].launches : 0.0, h->ktimes[i].total_ms);
        return IFX_OK;
    }
    auto it = h->kname_id.find(s);
    if (it == h->kname_id.end()) {   // a family name: the sum over "name@..." (the tracker's launches are timed per pyramid level: icp_residual@L0 ...)
        double tot = 0; int n = 0;
        for (size_t i = 0; i < h->knames.size(); i++)
            if (h->knames[i].compare(0, s.size() + 1, s + "@") == 0) { tot += h->ktimes[i].total_ms; n += h->ktimes[i].launches; }
        if (avg_ms) *avg_ms = n ? (float)(tot / n) : 0.f;
        if (launches) *launches = n;
        return IFX_OK;
    }
    const KernelTiming& k = h->ktimes[it->second];
    if (avg_ms) *avg_ms = k.launches ? (float)(k.total_ms / k.launches) : 0.f;
    if (launches) *launches = k.launches;
    return IFX_OK;
}

// ------------------------------------------------------------------ map access
extern "C" int ifx_map_view(ifx_t* h, ifx_soa_view* out)
{
    if (h) h->hot_valid = 0;   // (the caller holds the arrays' addresses from here on: whatever it writes, the next frame rebuilds the gathered copy)
    if (!h || !out) return IFX_E_INVALID;
    int r = ifx_sync(h);
    DevState hs;
    HIPCHK(h, hipMemcpy(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost));
    out->count = hs.count; out->capacity = h->cap;
    out->d_pos_conf = h->pc; out->d_norm_rad = h->nr; out->d_color = h->col; out->d_times = h->tm; out->d_img_corr = h->ic; out->d_votes = h->votes;
    return r;
}
static int read_state(ifx* h, DevState* hs)
{
    ifx_vlist_reap(h);   // counts are read: nothing outside the view list may outlive the age rule
    // every stream that writes the state: the model-to-model tracker (stream_c) leaves its verdict in lc[] / lc_candidates of the main state
    if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
    if (h->stream_b) HIPCHK(h, hipStreamSynchronize(h->stream_b));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(hs, h->d_state, sizeof(*hs), hipMemcpyDeviceToHost));
    return IFX_OK;
}
extern "C" int ifx_map_slots(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    DevState hs;
    int r = read_state(h, &hs);
    return r ? r : hs.count;
}
extern "C" int ifx_map_count(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    DevState hs;
    int r = read_state(h, &hs);
    return r ? r : hs.count - hs.n_dead;
}
extern "C" int ifx_compact(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    h->seg_counts_valid = 0;
    ifx_compact_enqueue(h, 1);
    return ifx_sync(h);
}

extern "C" int ifx_map_download(ifx_t* h, int max_n, float* pc, float* nr, float* col, float* tm, float* ic, float* votes)
{
    if (!h) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    h->seg_counts_valid = 0;
    int r = ifx_compact(h);   // live surfels in map order
    if (r) return r;
    DevState hs;
    r = read_state(h, &hs);
    if (r) return r;
    int n = std::min(hs.count, max_n);
    if (pc) HIPCHK(h, hipMemcpy(pc, h->pc, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (nr) HIPCHK(h, hipMemcpy(nr, h->nr, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (col) HIPCHK(h, hipMemcpy(col, h->col, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (tm) HIPCHK(h, hipMemcpy(tm, h->tm, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (ic) HIPCHK(h, hipMemcpy(ic, h->ic, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (votes) HIPCHK(h, hipMemcpy(votes, h->votes, (size_t)n * IFX_VF * 4, hipMemcpyDeviceToHost));   // (the device layout is the API's: one 48-float record per surfel)
    return n;
}

extern "C" int ifx_map_upload(ifx_t* h, int n, const float* pc, const float* nr, const float* col, const float* tm, const float* ic, const float* votes)
{
    if (!h || n < 0 || !pc || !nr || !col || !tm) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    for (CamCtx& c_ : h->cams) {   // a run ahead that is dropped may still be reading the camera's parked buffers: whatever reuses them queues behind it
        if (c_.ahead_valid && c_.ev_ahead) hipStreamWaitEvent(h->stream, c_.ev_ahead, 0);
        c_.ahead_valid = 0;
    }
    h->housekeeping_due = 0;   // (a decision about the map that is being replaced)
    h->seg_counts_valid = 0;
    h->map_external = 1;
    h->hot_valid = 0;
    h->age_epoch = INT_MAX;   // the uploaded surfels meet the clean pass's age rule from the next clean pass on (ifx_map.hip age_rule_gone)
    // spatially sharded map: this rank keeps the rows it owns (owner = hash of the uploaded position); creation numbers = the rows' indices
    const int n_all = n;
    std::vector<uint32_t> keep;
    std::vector<float> f_pc, f_nr, f_col, f_tm, f_ic, f_votes;
    keep.reserve(h->own ? (size_t)n / h->own_g + 1024 : (size_t)n);
    for (int i = 0; i < n; i++)
        if (!h->own || ifx_owner_of_point(pc[(size_t)i * 4], pc[(size_t)i * 4 + 1], pc[(size_t)i * 4 + 2], h->own_g) == h->cfg.rank) keep.push_back((uint32_t)i);
    if (h->own) {
        const size_t m = keep.size();
        auto gather = [&](const float* src, int width, std::vector<float>& dst) {
            if (!src) return (const float*)nullptr;
            dst.resize(m * width);
            for (size_t k = 0; k < m; k++) memcpy(&dst[k * width], &src[(size_t)keep[k] * width], (size_t)width * 4);
            return (const float*)dst.data();
        };
        pc = gather(pc, 4, f_pc); nr = gather(nr, 4, f_nr); col = gather(col, 2, f_col); tm = gather(tm, 2, f_tm); ic = gather(ic, 4, f_ic); votes = gather(votes, 48, f_votes);
        n = (int)m;
    }
    if (n > h->cap) { h->err = "upload exceeds capacity"; return IFX_E_CAPACITY; }
    if (h->stream_c) { HIPCHK(h, hipStreamSynchronize(h->stream_c)); h->lc_pending = 0; }
    if (h->stream_b) HIPCHK(h, hipStreamSynchronize(h->stream_b));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->pc, pc, (size_t)n * 16, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->nr, nr, (size_t)n * 16, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->col, col, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->tm, tm, (size_t)n * 8, hipMemcpyHostToDevice));
    if (ic) HIPCHK(h, hipMemcpy(h->ic, ic, (size_t)n * 16, hipMemcpyHostToDevice));
    else HIPCHK(h, hipMemset(h->ic, 0, (size_t)n * 16));
    if (votes) HIPCHK(h, hipMemcpy(h->votes, votes, (size_t)n * IFX_VF * 4, hipMemcpyHostToDevice));
    else HIPCHK(h, hipMemset(h->votes, 0, (size_t)n * IFX_VF * 4));
    DevState hs;
    HIPCHK(h, hipMemcpy(&hs, h->d_state, sizeof(hs), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(h->seq, keep.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hs.count = n; hs.n_dead = 0; hs.n_new = 0; hs.overflow = 0; hs.vl_valid = 0; hs.next_seq = (unsigned int)n_all; hs.first_live = 0;
    h->view_dirty = 0;
    {
        float rmax = 0.f;
        for (int i = 0; i < n; i++) { float r = nr[(size_t)i * 4 + 3]; if (r == r && r > rmax && r < 1e30f) rmax = r; }
        unsigned int bits; memcpy(&bits, &rmax, 4);
        if (bits > hs.r_max_bits) hs.r_max_bits = bits;
    }
    HIPCHK(h, hipMemcpy(h->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(h->upd_owner, 0xFF, (size_t)h->cap * 4));
    HIPCHK(h, hipMemset(h->labels, 0xFF, (size_t)h->cap * 4));
    h->labels_stale_all = 1;   // uploaded votes: the next segmentation call scans the labels of the whole map
    return IFX_OK;
}

extern "C" int ifx_set_pose(ifx_t* h, const float* pose16, int tick)
{
    if (!h || !pose16) return IFX_E_INVALID;
    ifx_drop_tracked(h);
    for (CamCtx& c_ : h->cams) {   // a run ahead that is dropped may still be reading the camera's parked buffers: whatever reuses them queues behind it
        if (c_.ahead_valid && c_.ev_ahead) hipStreamWaitEvent(h->stream, c_.ev_ahead, 0);
        c_.ahead_valid = 0;
    }
    DevState hs;
    int r = read_state(h, &hs);
    if (r) return r;
    memcpy(hs.pose, pose16, 64);
    memcpy(hs.last_pose, pose16, 64);
    pose_inverse(hs.pose, hs.pose_inv);
    HIPCHK(h, hipMemcpy(h->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
    if (tick != h->tick) h->age_epoch = INT_MAX;   // the clock jumps: the age rule is evaluated per frame from the next clean pass on (read_state above reaped what was overdue)
    h->tick = tick;
    return IFX_OK;
}

extern "C" const int32_t* ifx_ids_after(ifx_t* h)
{
    if (!h) return nullptr;
    if (ifx_ids_ensure(h)) return nullptr;   // (enqueued on the handle's main stream, like the frame that precedes it)
    return h->ids_after;
}

extern "C" int ifx_image_download(ifx_t* h, const char* name, void* out, int64_t max_bytes)
{
    if (!h || !name || !out) return IFX_E_INVALID;
    std::string s(name);
    size_t P = (size_t)h->P, bytes = 0;
    const void* src = nullptr;
    if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
    if (s == "ids_after") { const int r_ = ifx_ids_ensure(h); if (r_) return r_; src = h->ids_after; bytes = P * 4; }
    else if (s == "ids_tmp") { src = h->ids_tmp; bytes = P * 4; }
    else if (s == "index") { src = h->index_id; bytes = P * 4; }
    else if (s == "index_vc") { src = h->index_vc; bytes = P * 16; }
    else if (s == "index_ct") { src = h->index_ct; bytes = P * 16; }
    else if (s == "index_nr") { src = h->index_nr; bytes = P * 16; }
    else if (s == "pred_vertex") { src = h->pred_vertex; bytes = P * 16; }
    else if (s == "pred_normal") { src = h->pred_normal; bytes = P * 16; }
    else if (s == "pred_image") { src = h->pred_image; bytes = P * 4; }
    else if (s == "pred_inst") { src = h->pred_inst; bytes = P * 4; }
    else if (s == "pred_time") { src = h->pred_time; bytes = P * 2; }
    else if (s == "old_vertex" && h->d_m2m) { src = h->old_vertex; bytes = P * 16; }
    else if (s == "old_normal" && h->d_m2m) { src = h->old_normal; bytes = P * 16; }
    else if (s == "old_image" && h->d_m2m) { src = h->old_image; bytes = P * 4; }
    else if (s == "old_time" && h->d_m2m) { src = h->old_time; bytes = P * 2; }
    else if (s == "act_vertex" && h->d_m2m) { src = h->act_vertex; bytes = P * 16; }
    else if (s == "act_normal" && h->d_m2m) { src = h->act_normal; bytes = P * 16; }
    else if (s == "act_image" && h->d_m2m) { src = h->act_image; bytes = P * 4; }
    else if (s == "fill_vertex") { src = h->fill_vertex; bytes = P * 16; }
    else if (s == "fill_normal") { src = h->fill_normal; bytes = P * 16; }
    else if (s == "fill_image") { src = h->fill_image; bytes = P * 4; }
    else if (s == "depth_filtered") { src = h->depth_filt; bytes = P * 2; }
    else if (s == "depth_metric") { src = h->dm; bytes = P * 4; }
    else if (s == "depth_metric_filtered") { src = h->dmf; bytes = P * 4; }
    else { h->err = "unknown image " + s; return IFX_E_INVALID; }
    if ((int64_t)bytes > max_bytes) { h->err = "buffer too small"; return IFX_E_INVALID; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return (int)bytes;
}

#ifdef IFX_STAMPS
void ifx_debug_copy2(long long* out);
// diagnostic builds only (make FLAGS+=-DIFX_STAMPS): accumulated in-kernel cycle stamps
extern "C" int ifx_debug_stamps(ifx_t* h, long long* out8, int reset)
{
    DevState hs;
    int r = read_state(h, &hs);
    if (r) return r;
    memcpy(out8, hs.dbg, sizeof(hs.dbg));
    ifx_debug_copy2(out8 + 8);
    if (reset) { memset(hs.dbg, 0, sizeof(hs.dbg)); hipMemcpy((char*)h->d_state + offsetof(DevState, dbg), hs.dbg, sizeof(hs.dbg), hipMemcpyHostToDevice); }
    return IFX_OK;
}
#endif
