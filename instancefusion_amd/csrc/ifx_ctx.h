// ifx_ctx.h -- host-side context of libifx.so (one handle = one GPU, one stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/ifx_c_api.h"
#include "ifx_dev.h"

#ifndef IFX_LIST_SEGS
#define IFX_LIST_SEGS 8   // segments of a work list, each with its own length counter (ifx_map.hip)
#endif

// ---- device-resident per-frame state: everything the kernels of one frame hand to each other
// without a host round trip (the reference reads back 60-125 times per frame).
struct DevState {
    // poses (row-major 4x4, camera-to-world)
    float pose[16];       // current pose (tracker output / fusion input)
    float pose_inv[16];
    float last_pose[16];
    float weighting;      // velocity weighting, EF/ElasticFusion.cpp:433-449
    int dense_enough;     // EF/ElasticFusion.cpp:252-267
    // map bookkeeping
    int count;            // slots in use (incl. tombstones)
    int n_dead;           // tombstones
    int n_new;            // new surfels appended by the last clean
    int overflow;         // set when an append hit the capacity
    unsigned int list_n[4];   // (unused: the work-list lengths live in ifx::d_list_ctr, one counter per list segment)
    unsigned int r_max_bits;  // float bits of an upper bound of every surfel radius ever stored (conservative frustum margin)
    // tracker state (RGBDOdometry::getIncrementalTransformation, EF/Utils/RGBDOdometry.cpp:267-603)
    float Rprev[9], tprev[3], Rprev_inv[9];
    float Rcurr[9], tcurr[3];
    double resultRt[16];
    float krkinv[9], kt[3];
    // so3 pre-alignment
    double resultR[9], lastResultR[9];
    float R_lr[9];
    float so3_lastError, so3_lastCount;
    int so3_done;
    float imageBasis[9], kinv[9], krlr[9];
    // diagnostics
    float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    double lastA[36], lastb[6];
    float icp29[29], rgb29[29];
    int rgb_count, rgb_sigma;
    // instance
    int seg_counts[2];
    int seg_acc[2];              // whetherDoSegmentation sums of the frame being finished (k_raster_finish -> k_frame_result)
    int fold_total;              // != 0: this frame's k_splat_resolve accumulated fold_acc itself (= the number of dense-test samples); k_frame_result folds and clears
    int fold_acc[16][4];         // partial sums by blockIdx.x & 15: vote mass, empty lattice pixels, lit dense-test samples, -
    int first_live;              // lowest live slot: the reference's "surfel 0" (drawn as id 0 = "no surfel" in every id-carrying image; ifx_map.hip key_id)
    unsigned int append_ticket;  // last-block ticket of k_vlist_flatten (k_append_scan, whose it was, publishes without one since round 5)
    int app_count0; unsigned int app_seq0, app_vln0;   // count / next_seq / vl_n[0] when the frame's new-surfel flags were taken (k_new_flags_count): k_append_scan's starting point
    unsigned int result_ticket;  // last-block ticket of k_splat_resolve when it also writes the frame result (FrameOut)
    unsigned int next_seq;       // creation number of the next new surfel (spatially sharded map: identical on every rank)
    int hot_stale;               // option hot_verify: slots whose gathered copy ("hot record") differed from the store when a frame was about to read it (ifx_hot_records_stale; 0 unless somebody wrote the store through a kept ifx_map_view pointer)
    int range_exceeded;          // run-time guard of the tracker's exact sums: diagonal totals found beyond half their exact range since the handle was created (ifx_track.hip range_exceeded7; ifx_tracker_range_exceeded)
    float spec_pose[16], spec_pose_inv[16], spec_weighting;   // result of a tracker run enqueued ahead of its frame (k_commit_pose publishes it)
    // local loop-closure detection (EF/ElasticFusion.cpp:453-566).  The model-to-model tracker has a DevState of its own (ifx::d_m2m):
    // there `count` counts the pixels of the INACTIVE render and `skip` is set when it is empty; the verdict lands in lc[] of the MAIN state.
    int skip;             // tracker kernels return at once (model-to-model run with nothing to align)
    float lc[24];         // ifx_loop_closure_diag layout
    int lc_candidates;
    // cached view list (ifx_map.hip, "View list"): slots that can touch the image from any pose near vl_pose
    float vl_pose[16];    // camera-to-world pose the list was built for
    int vl_valid;         // the list describes the store (cleared by compaction, upload, deformation, first frame)
    int vl_scan;          // decision for the frame being enqueued: 1 = k_cull_frame rebuilds the list, 0 = it returns at once
    int vl_age;           // frames since the last scan
    int vl_scans;         // scans so far (diagnostics)
    unsigned int vl_n[2]; // lengths of the two flat view lists: [0] inside the time window (grows with the appended surfels), [1] stable slots outside it
    unsigned int vl_seg_n[2 * IFX_LIST_SEGS], vl_seg_off[2 * IFX_LIST_SEGS];   // segment lengths / offsets of the scan's raw output (k_vlist_offsets, which re-arms the live counters)
    long long dbg[8];     // in-kernel cycle stamps (IFX_STAMPS builds only)
    // Gauss-Newton hand-off words of THIS tracker instance.  They live in the state so that a kernel reaches them through the state pointer, which
    // arrives preloaded with the wave (kernarg preload): no kernarg round trip in front of the first dependent load.  A line of their own: they are
    // the target of atomics while other fields are read through the scalar cache.
    alignas(64) int gn_res[2];   // residual totals (count, sigma) of the iteration in flight: written by k_icp_residual, consumed and re-armed by k_rgb_step_solve
    unsigned int gn_ticket;      // last-block ticket of k_rgb_step_solve
    int gn_pad[13];
    double gn_acc[2 * IFX_ACC_REPL * IFX_ACC_STRIDE];   // exact accumulator rows of the iteration in flight: [0] ICP, [1] photometric (layout of block_sum_exact)
    // persistent level kernel (k_gn_level): accumulator rows and residual totals double-buffered by iteration parity, one grid-barrier word per pyramid level
    double gn_acc2[2 * 2 * IFX_ACC_REPL * IFX_ACC_STRIDE];   // [parity][ICP | photometric]
    alignas(64) int gn_res2[2 * 16];                          // [parity] (count, sigma), a line each
    alignas(64) unsigned int gn_bar[4 * 32 * 16];             // [level][32 sub-counters] arrivals, a line each (gn_grid_barrier)
    int gn_timeout;                                           // levels the persistent kernel could not finish and block 0 re-ran alone (never in a healthy run; a count, not an error: ifx_tracker_fallbacks)
    int gn_abort;                                             // this run's persistent kernel gave up at a barrier: its blocks leave, k_gn_level_solo re-runs the level on one workgroup (reset at the start of every run)
    int gn_done_seq;                                          // level + 1 of the last persistent launch whose result was published (k_gn_level's block 0 / k_gn_level_solo)
    int gn_spin_limit, gn_fault;                              // test hooks (options gn_spin_limit / gn_fault): polls before a barrier gives up (0: 2^21); != 0: block 1 never arrives at barrier number gn_fault
    // "solve in the next launch's prologue" (option gn_prologue, ifx_track.hip): iteration j of a run's two-launch tail keeps its sums in parity j & 1 -- every block of
    // the NEXT launch reads the 2 x 29 totals, rebuilds the 6x6 system and solves it itself (no last-block ticket, no pose store / pose load between the two launches);
    // the running increment after iteration j lives in gnp_RRt[j & 1] (block 0 of the launch that solved it writes it; its readers are the launch after that)
    alignas(64) double gnp_acc[2 * 2 * IFX_ACC_REPL * IFX_ACC_STRIDE];   // [parity][ICP | photometric]
    alignas(64) int gnp_res[2 * 16];                                     // [parity] (count, sigma), a line each
    alignas(64) double gnp_RRt[2][16];
};

// Ids in the id images are slot indices on an unsharded map and creation numbers on a spatially sharded one (ifx_map.hip, key_id / local_slot):
// the instance layer, which goes from a pixel's id to the votes / position of that surfel, maps them through this.  -1: no surfel of THIS rank.
// On a sharded map the id is a creation number: an UNSIGNED 32-bit value carried in the int32 id image (0 = no surfel), so numbers from 2^31 on
// stay reachable; k_append_scan raises DevState::overflow before the numbering could wrap (include/ifx_c_api.h, "creation numbers").
struct IdMap { const uint32_t* seq; int own_n; };
__device__ __forceinline__ int idmap_slot(const IdMap& m, int count, int id)
{
    if (m.own_n <= 0) return (id > 0 && id < count) ? id : -1;
    if (id == 0) return -1;
    int lo = 0, hi = count - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned int v = m.seq[mid];
        if (v == (unsigned int)id) return mid;
        if (v < (unsigned int)id) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

struct FrameResult {   // copied to pinned host memory at the end of every frame
    float pose[16];
    float diag[8];
    int count, n_dead, n_new, overflow;
    int gn_timeout;      // DevState::gn_timeout: levels re-run by the persistent kernel's fallback so far (ifx_tracker_fallbacks)
    int seg_counts[2];   // checkProjectDepthAndInstance sums of this frame (vote mass under every 10th pixel, pixels without a surfel)
};

// The frame result (pinned host memory) + the trajectory slot, by one wave.  Shared by k_frame_result (a launch of its own) and by k_splat_resolve's last block (the
// view-list frame path: one dispatch less per frame).  folded_total > 0: the caller's own resolve accumulated DevState::fold_acc with agent-scope atomics in THIS
// launch -- they are read and re-armed the same way; 0: whatever an earlier launch left in fold_total / fold_acc (plain accesses, ordered by the launch boundary).
__device__ __forceinline__ void frame_result_wave(DevState* st, FrameResult* out, float* traj_slot, int folded_total, int lane)
{
    // view-list frames: the resolve of the prediction accumulated the end-of-pass sums (k_splat_resolve, FinishFold) in sixteen partials
    const int fold = folded_total > 0 ? folded_total : st->fold_total;
    int f_mass = 0, f_empty = 0, f_lit = 0;
    if (fold && lane < 16) {
        if (folded_total > 0) {
            f_mass = __hip_atomic_load(&st->fold_acc[lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_empty = __hip_atomic_load(&st->fold_acc[lane][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_lit = __hip_atomic_load(&st->fold_acc[lane][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->fold_acc[lane][0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->fold_acc[lane][1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->fold_acc[lane][2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            f_mass = st->fold_acc[lane][0]; f_empty = st->fold_acc[lane][1]; f_lit = st->fold_acc[lane][2];
            st->fold_acc[lane][0] = 0; st->fold_acc[lane][1] = 0; st->fold_acc[lane][2] = 0;
        }
    }
    f_mass = wave_sum_i(f_mass); f_empty = wave_sum_i(f_empty); f_lit = wave_sum_i(f_lit);
    if (lane != 0) return;
    if (fold) {
        st->seg_acc[0] += f_mass; st->seg_acc[1] += f_empty;
        st->dense_enough = ((float)f_lit / (float)fold > 0.75f) ? 1 : 0;   // EF/ElasticFusion.cpp:252-267
        if (folded_total <= 0) st->fold_total = 0;
    }
    out->seg_counts[0] = st->seg_acc[0]; out->seg_counts[1] = st->seg_acc[1];
    st->seg_acc[0] = 0; st->seg_acc[1] = 0;
    for (int k = 0; k < 16; k++) { out->pose[k] = st->pose[k]; traj_slot[k] = st->pose[k]; }
    out->diag[0] = st->lastICPError; out->diag[1] = st->lastICPCount; out->diag[2] = st->lastRGBError; out->diag[3] = st->lastRGBCount;
    out->diag[4] = st->lastSO3Error; out->diag[5] = st->lastSO3Count; out->diag[6] = st->weighting; out->diag[7] = st->dense_enough ? 0.f : 1.f;
    out->count = st->count; out->n_dead = st->n_dead; out->n_new = st->n_new; out->overflow = st->overflow;
    out->gn_timeout = st->gn_timeout;
}

struct Pyr {
    int w[IFX_NUM_PYRS], h[IFX_NUM_PYRS];
    uint16_t* depth_tmp[IFX_NUM_PYRS];
    float *vmap_curr[IFX_NUM_PYRS], *nmap_curr[IFX_NUM_PYRS];
    float *vmap_cam[IFX_NUM_PYRS], *nmap_cam[IFX_NUM_PYRS];   // model maps in the camera frame (before the global transform)
    float *vmap_prev[IFX_NUM_PYRS], *nmap_prev[IFX_NUM_PYRS];
    float* last_depth[IFX_NUM_PYRS];                          // == next_depth in the frame-to-model tracker (reference quirk)
    float* next_depth[IFX_NUM_PYRS] = {};                     // model-to-model tracker only (nullptr: last_depth)
    // reduction scratch of this tracker instance (the two instances can be on the GPU at the same time)
    double* acc = nullptr;                                    // exact accumulator rows of this tracker instance: [3 quantities: icp, rgb, so3][IFX_ACC_REPL replicas][IFX_ACC_STRIDE] f64, all zero between launches
    int* res_partials = nullptr;
    unsigned int* ticket = nullptr;                           // [0] last-block ticket of k_rgb_step_solve, [8..9] residual totals
    uint8_t *last_img[IFX_NUM_PYRS], *next_img[IFX_NUM_PYRS], *lastnext_img[IFX_NUM_PYRS];
    int16_t *didx[IFX_NUM_PYRS], *didy[IFX_NUM_PYRS];
    float* cloud[IFX_NUM_PYRS];
    void* corres[IFX_NUM_PYRS];                               // 8 B per pixel: short zx, zy; float diff
};

// Everything that depends only on the input frame (the "frame side" of a frame): copies of the images, the
// bilateral / metric depth, the frame pyramids and the SO(3) pre-alignment against the previous image.  Two
// slots alternate so that the frame side of frame k+1 can run on the side stream while frame k is tracked
// and fused (ifx_prefetch_frame_device), or at least next to the model pyramid of its own frame.
struct FrameSlot {
    uint8_t* rgb = nullptr;
    uint16_t *depth_raw = nullptr, *depth_filt = nullptr;
    float *dm = nullptr, *dmf = nullptr;
    uint16_t* depth_tmp[IFX_NUM_PYRS] = {};
    float *vmap_curr[IFX_NUM_PYRS] = {}, *nmap_curr[IFX_NUM_PYRS] = {};
    uint8_t* next_img[IFX_NUM_PYRS] = {};
    int16_t *didx[IFX_NUM_PYRS] = {}, *didy[IFX_NUM_PYRS] = {};
    DevState* so3 = nullptr;          // shadow state: only the SO(3) fields are used
    hipEvent_t ready = nullptr;       // recorded on the side stream when the slot is complete
    hipEvent_t released = nullptr;    // recorded on the main stream when the frame that used the slot is done
    const void *src_rgb = nullptr, *src_depth = nullptr;
    int for_tick = -1;                // frame the slot was prepared for
};


// ---- cached view list (ifx_map.hip "View list"): margins and the per-frame decision, shared by the kernels that commit a pose
#ifndef VL_ROT
#define VL_ROT 0.0523599f      // 3 degrees
#endif
#ifndef VL_TRANS
#define VL_TRANS 0.06f         // metres
#endif
#define VL_MAX_AGE 32
#define LIST_V 3
#define LIST_VI 4
#define IFX_LIST_CTR_STRIDE 32
#define IFX_KEY_SLACK 512       // bytes behind key_index | word that a reduce-scatter over up to 64 ranks may read (tiles of equal size)

// decision for the frame whose pose was just committed; one thread (k_track_end / k_commit_pose / pose adoption)
// (A = the pose the list was built for, B = the pose just committed: pointers into the state, or registers the caller loaded ahead of its stores)
__device__ __forceinline__ void vlist_decide_core(DevState* st, const float* A, const float* B, int valid, int age)
{
    bool ok = valid && age < VL_MAX_AGE;
    if (ok) {
        const float dx = B[3] - A[3], dy = B[7] - A[7], dz = B[11] - A[11];
        float tr = 0.f;   // trace(Ra^T Rb) = sum of the element-wise products
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) tr += A[r * 4 + c] * B[r * 4 + c];
        // 10 % slack on both margins for the rounding of this test itself
        ok = (dx * dx + dy * dy + dz * dz <= 0.81f * VL_TRANS * VL_TRANS) && ((tr - 1.0f) * 0.5f >= cosf(0.9f * VL_ROT)) && tr == tr;
    }
    if (ok) { st->vl_scan = 0; st->vl_age = age + 1; return; }
    st->vl_scan = 1; st->vl_age = 0; st->vl_valid = 1; st->vl_scans++;
#pragma unroll
    for (int k = 0; k < 16; k++) st->vl_pose[k] = B[k];
    st->vl_n[0] = 0; st->vl_n[1] = 0;
}
__device__ inline void vlist_decide(DevState* st, unsigned int* __restrict__ lctr)
{
    (void)lctr;
    vlist_decide_core(st, st->vl_pose, st->pose, st->vl_valid, st->vl_age);
}

// ---- camera contexts (BASELINE configuration 5: K streams into ONE map).  What a camera carries from its last frame to its next: the pose block of the
// state (pose, inverse, last pose, velocity weighting, dense flag), the model prediction it tracks against (+ fill-in), the intensity pyramid of its last
// frame (the "previous image" of the SO(3) pre-alignment) and the id image.  ifx_camera_select parks the current camera's set and brings another one in.
struct CamCtx {
    float* state = nullptr;            // IFX_CAM_STATE_BYTES of DevState from offset 0
    uint8_t* pred = nullptr;           // the prediction block (ifx::pred_bytes)
    float *fill_v = nullptr, *fill_n = nullptr;
    uint8_t* fill_i = nullptr;
    uint8_t* img[IFX_NUM_PYRS] = {};
    int32_t* ids = nullptr;
    int valid = 0;
    // a tracker run ahead for this camera's NEXT frame (ifx_owner_track_ahead): the pose block it produced, parked until the frame comes (one per camera: on one GPU
    // that tracks all K cameras, K runs are pending at any time)
    void* ahead_pose = nullptr;
    hipEvent_t ev_ahead = nullptr;     // behind this camera's run (its own event: a frame must not wait for a run enqueued after its own)
    int ahead_valid = 0;
    const void *ahead_rgb = nullptr, *ahead_depth = nullptr;
    int pred_root = -1;                // who holds this camera's parked prediction COMPLETE: -1 every rank (all-reduced), else the one rank exchange 5 reduced it to (ifx::pred_root while the camera is live)
};
#define IFX_CAM_STATE_BYTES 200        // pose[16], pose_inv[16], last_pose[16], weighting, dense_enough

struct KernelTiming { double total_ms = 0; int launches = 0; };
struct PendingEvent { int name_id; hipEvent_t a, b; };

struct ifx {
    ifx_config cfg;
    int w, h, P, cap;
    hipStream_t stream = nullptr;      // main stream: model side, tracking, map, instance layer
    hipStream_t stream_b = nullptr;    // side stream: frame side (FrameSlot)
    hipStream_t stream_c = nullptr;    // loop-closure detection: the model-to-model tracker, under the map passes of its frame
    hipEvent_t ev_lc_ready = nullptr, ev_lc_done = nullptr;
    int lc_pending = 0, lc_deferred = 0;
    hipStream_t cur = nullptr;         // stream LAUNCH enqueues on (== stream except while a frame side is enqueued)
    std::vector<FrameSlot> slot = std::vector<FrameSlot>(3);   // [0], [1] alternate per frame; [2]: unused; [3 + c]: the frame camera c's tracker runs ahead on (ifx_owner_track_ahead), allocated on
                                        // first use -- the frame then takes its frame side from there instead of computing it again
    int last_frame_slot = 0;            // the slot the frame most recently enqueued is bound to (its raw images, its intensity pyramid: the resident frame of a segmentation call, what a camera parks)
    int cur_slot = 0;
    std::vector<CamCtx> cams;           // camera contexts (ifx_camera_count); empty: the handle is its one camera
    int cur_cam = 0;
    // K streams over a sharded map, camera k tracked by rank k: rank k's tracker for camera k's NEXT frame runs on the third stream under the other cameras' map phases
    // (ifx_owner_track_ahead) -- a tracker instance of its own (state, model-side pyramids, frame slot 2, SO(3) accumulators), fed from the camera's PARKED context
    DevState* d_cam_trk = nullptr;
    Pyr cam_pyr;
    double* cam_so3_acc = nullptr; unsigned int* cam_so3_ticket = nullptr;
    hipEvent_t ev_cam_ahead = nullptr, ev_cam_parked = nullptr, ev_cam_side = nullptr;
    hipStream_t cam_side_stream = nullptr;   // set around ifx_tracker_camera_ahead: the stream its frame side goes to (null: the same stream as its tracker)
    int cam_ahead_used = 0;             // frames whose tracker was taken from a run ahead (diagnostics / tests)
    unsigned long long *gfl_index = nullptr, *gfl_splat = nullptr;   // sharded map: the word behind key_index / behind [key_splat | key_ids] that carries the lowest live creation number (this rank's before the exchange, every rank's behind it; ifx_map.hip FIRST_LIVE: the reference's "surfel 0")
    int32_t* own_slot_img = nullptr;    // sharded map, frame path: [4][P] slots of this rank's local winners (index map, splat, ids) and of the associated surfels (k_own_translate, ifx_map.hip)
    int own_fast = 0, own_fast_raster = 0;   // this frame's key images were drawn with slots and translated (index maps / the end-of-frame raster)
    // option own_lazy_ids (sharded map): the frame draws and exchanges the id keys of the lattice whetherDoSegmentation samples only -- exchange 4 is
    // [key_splat | lattice keys | word] = 8 P + 8 L + 8 bytes instead of 16 P + 8 -- and whoever needs the whole id image (a segmentation call, a download, a camera
    // that is parked) gets it from an id render + ONE key exchange of its own (ifx_owner_ids_begin / exchange 200 / _resume; inline when the library holds the communicator)
    int opt_own_lazy_ids = 0;
    // option own_key_rs (sharded map): the index keys of exchanges 0 and 2 have ONE kind of consumer -- k_index_resolve, which reads the winner's creation number -- so their
    // MIN runs as reduce-scatter + all-gather of the low words (ifx_owner_exchange op 6; ifx_comm.hip comm_keys_rs): 12 instead of 16 bytes per key and link direction
    int opt_own_key_rs = 0;
    // option own_track_rows (sharded map with the library's communicator): the tracker's two reductions run over this rank's share of the pixel blocks and
    // the 2 x 29 exact sums are all-reduced (ifx_track.hip k_icp_residual_rows).  _emulate = G: one rank plays G in turn (test switch)
    int opt_own_track_rows = 0, opt_own_track_rows_emulate = 0;
    int track_rc = 0;                   // result of the collectives the last tracker run enqueued (own_track_rows)
    int own_ids_lat = 0;                // the frame in flight exchanges the lattice form (set by phase 4, read by ifx_owner_exchange(4) and phase 5)
    int own_ids_pending = 0;            // an id render of the shard is waiting for its key exchange (ifx_owner_exchange(200))
    unsigned long long* own_lat_tmp = nullptr;   // L + 1 keys: staging of the lattice when the local raster drew the whole image (per-pass cull frames)
    int own_need_decide = 0;            // sharded map, one rank tracks: this rank received the frame's pose (exchange 310) and has not yet run the view-list decision for it
    int pred_root = -1;                 // sharded map: the rank the live camera's prediction was last reduced to (exchange 5 with a tracking rank: the other ranks hold partial sums), -1: all-reduced.
                                        // The same on every rank (it follows the host's call sequence), so every rank refuses alike a frame / run-ahead that would track from a partial block
    int own_track_rank = -1;            // sharded map: the one rank that tracks the frames to come (-1: every rank tracks, replicated); the others receive the pose block (exchange 310)
    float own_frame_pose[16]; int own_frame_pose_set = 0;   // sharded map: the next frame takes this pose instead of tracking (ifx_owner_set_frame_pose)
    int own = 0, own_g = 1;             // spatially sharded map (ifx_config::n_ranks > 1, or -1: a world of one): this handle stores the surfels it owns; own_g = number of ranks
    size_t pred_bytes = 0; int* pred_tail = nullptr;   // the prediction images are one allocation of pred_bytes (+ a 16-byte tail that travels with them on a sharded map)
    void* comm = nullptr;               // ifx_comm.hip: the RCCL communicator + exchange scratch of a sharded map (ifx_owner_init_comm / ifx_owner_set_comm)
    int shard_rank = 0, shard_n = 1;    // sharded projection: this rank's slice of the slots (ifx_set_shard)
    int opt_two_streams = 1;
    int opt_stage_timing = 0;           // HIP events around the stages of every frame (ifx_stage_ms); each record is a marker packet on the queue: ~4 % of the frame rate
    int opt_gn_prologue_blocks = 2048;  // gn_prologue only for launches of at most this many blocks (every block repeats the solve)
    int opt_cam_side = 1;               // a run-ahead tracker's frame side on the side stream, its tracker on the third (0: both on the third)
    int opt_cam_swap = 1;               // a camera switch between two existing contexts hands the prediction / fill-in / id blocks over by pointer instead of copying them
    void* hot = nullptr;                // [cap] 64-byte records (position + confidence | normal + radius | times): what the frame path's gathers read (ifx_map.hip "hot records")
    int hot_valid = 0;                  // the copy describes the store (its three per-frame writers write both; everything else that writes the store clears this)
    int opt_hot_verify = 0;             // option hot_verify (debug): every frame that trusts the gathered copy first compares it with the store (one streaming launch) and counts the slots that differ
    int opt_hot = 1;                    // option hot_records
    void* frame_hot = nullptr;          // the copy as this frame's map passes use it (null: the arrays)
    int opt_clean_raster = 1;           // view-list frames: ONE walk of the view list cleans and rasterises (k_raster_view<., true>, with k_new_flags_count's blocks in the same launch):
                                        // two launches and one set of gathers less on every frame's chain (ifx_map.hip, "CLEAN")
    int clean_raster_pending = 0;       // ifx_map_frame left the clean / new-surfel flags / append of this frame to ifx_map_predict's launches
    float frame_weight_mult = 1.f;      // weight multiplier of the frame being enqueued (bounds the confidence a new surfel can start with)
    int opt_fold_result = 1;            // view-list frames: the frame result is written by the last block of the frame's last launch (k_splat_resolve) instead of a launch of its own
    float* result_fold_traj = nullptr;  // set by enqueue_frame around ifx_map_predict: the trajectory slot of the frame being finished (null: nobody asked)
    int result_folded = 0;              // ifx_map_predict's answer: the resolve took the frame result along
    int opt_slic_ahead = 1;             // when the cadence says the announced next frame ends with a segmentation call: its superpixels + merge (frame-only work) go to the side stream now
    int opt_track_ahead = 1;            // with a hinted next frame: enqueue its tracker right behind the current frame, before the host decides about segmentation
    int tracked_ahead = 0;              // tick whose tracker is already on the queue (result parked in DevState::spec_*)
    int opt_side_gate = 0; hipEvent_t ev_gate = nullptr;   // (experiment) where the announced frame's image-only work may start: 0 at once, 1 behind the commit, 2 behind the frame
    int opt_overdue_rule = 1;           // a list rebuild first applies the age rule that slots no list held have outlived (k_cull_frame); 0: round 4's scan, which let such a slot into the new list alive (test switch)
    int opt_vlist_one = 0;              // the view list's segment offsets and its concatenation in ONE launch (k_vlist_flatten: the last block publishes) instead of two; measured equal (1504 against 1503 frames/s in the driver-shaped window, tools/ab_driver.sh: a launch that only finds out that it has nothing to do costs the chain next to nothing when the next launch is already queued): off
    int opt_own_first_live = 1;         // sharded map: the reference's "surfel 0" is the lowest live creation number of any rank (ifx_map.hip FIRST_LIVE); 0: round 4's rule -- creation number 0 for ever (test switch)
    int opt_vote_per_mask = 1;          // instance votes: one launch per mask, in mask order, as the reference (IF/Core/InstanceFusion.cpp:986-1000) -- the order is part of the result while a packed counter's low half is negative (ifx_instance.hip k_vote_update_all); 0: round 4's one launch over all masks (experiments only)
    int opt_side_late = 0;              // a frame whose tracker ran ahead enqueues the announced next frame's side behind its own map passes instead of in front of them (measured: 1490 against 1510 frames/s -- the frame side then runs beside the next tracker instead of beside this frame's map passes; off)
    int opt_pace = 1;                   // ifx_enqueue_frame_device waits for the previous frame's result before it enqueues (bounded run-ahead)
    int opt_ff_union = 1;               // flood fill of the masks: two-way edges merged by union-find before the directed relaxation (k_ff_merge)
    int opt_seg_aside = 1;              // a segmentation call that finds the next frame's tracker already queued on the main stream runs beside it on stream_c (the call is
                                        // synchronous for the host, so nothing has to join afterwards); ifx_instance / ifx_slic / ifx_knn enqueue on h->cur throughout
    hipEvent_t ev_result = nullptr;     // the `released` event of the slot of the last frame (recorded after k_frame_result)
    int n_side_prepared = 0, n_tracked_ahead = 0, n_host_hinted = 0;   // ifx_lookahead_stats: frames that found their frame side done / their tracker run / came through ifx_hint_next_frame
    int hint_kind = 0;                      // 0: device pointers (ifx_hint_next_frame_device), 1: the pinned staging pair of the announced frame's parity (ifx_hint_next_frame)
    uint8_t* hint_stage_rgb[2] = {nullptr, nullptr}; uint16_t* hint_stage_depth[2] = {nullptr, nullptr};   // ifx_hint_next_frame: pinned staging by frame parity (allocated at the first use)
    const void *hinted_src_rgb[2] = {nullptr, nullptr}, *hinted_src_depth[2] = {nullptr, nullptr};       // ... the caller's pointers of the announced frame, and
    int hinted_tick[2] = {-1, -1};                                                                        // ... the frame it was announced for
    const uint8_t* hint_rgb = nullptr;      // next frame announced by ifx_hint_next_frame_device, not enqueued yet
    const uint16_t* hint_depth = nullptr;
    std::string err;
    int tick = 1;
    int ids_pending = 0;
    int opt_fold_finish = 1;            // view-list frames: the end-of-pass sums (dense test, whetherDoSegmentation) ride in k_splat_resolve instead of a launch of their own
    int ids_view_ok = 0;                // the cached view list still describes store and pose of the frame that drew the sparse id image (ifx_ids_ensure may walk it)
    int view_scan_tick = -1;            // frame whose view-list scan is already on the queue (the loop-closure renders come before the map passes)
    int opt_lc_view = 1;                // loop-closure detection: its two renders from the view lists (one k_raster_view in dual mode) instead of a scan of the store + k_raster_list
    int opt_lazy_ids = 1;               // the frame renders the id image on the lattice whetherDoSegmentation samples; the whole image on demand (ifx_ids_ensure)
    int ids_full_valid = 1, ids_sparse_frame = 0;
    int ids_full_hint = 0;              // the cadence says the NEXT frame ends with a segmentation call (ifx_should_segment): that frame draws the whole id image itself
    // options
    int opt_compact_every_frame = 0;
    int last_compact_tick = -1;
    int opt_compact_divisor = 8;        // housekeeping: compact when tombstones exceed count / divisor (or capacity gets tight)
    int opt_kernel_timing = 0;
    int opt_reference_passes = 0;   // also run the id renders nobody consumes (EF/ElasticFusion.cpp:679-680)
    int opt_icp_px = 0;              // ICP and residual reductions on the same pixels of one thread, all loads in two batches (k_icp_residual_px; bits: 1 level 0, 2 levels 1-2, 4 one pixel per thread); measured slower: off
    int opt_model_fused = 0;         // model pyramid of the frame tracker in one launch (k_model_pyr3) when the image size allows; measured equal to the three launches (28 vs 27 us): off
    int opt_gn_persist_blocks = 128;  // ... and only while its grid has at most this many blocks: the meetings cost grows with the blocks (75 at 160 x 120: faster; 300: slower)
    int opt_gn_persist = 0;          // bit i: all Gauss-Newton iterations of pyramid level i in one persistent launch (k_gn_level) when its grid fits the GPU.  Round 3's default was 4 (the coarsest level,
                                     // +1.4 %); with the solve in the next launch's prologue the two-launch form is within 0.7 % of it (profiles/r04_b_ab_gn_prologue*.txt), and a spinning grid barrier does
                                     // not belong on the default frame path for that: off.  As an option it is safe: a meeting that does not happen costs time, never the pose (k_gn_level_solo)
    int gn_max_blocks[4] = {0, 0, 0, 0};   // co-resident blocks of k_gn_level<1 | 2 | 3 | 4>
    int opt_icp_lds = 0;             // level-0 ICP reduction on 64 x 16 tiles with the model maps staged in LDS (measured slower: DESIGN.md section 6)
    int opt_rgb_blocks = 0;          // cap on the blocks of the photometric step (0: 192)
    int opt_gn_prologue = 1;         // two-launch Gauss-Newton iterations: the 6x6 solve of iteration j runs in the prologue of EVERY block of iteration j + 1's first launch (no last-block
                                     // hand-off inside the photometric step's launch); 0: round 3's form, the last block of the photometric step solves and stores the pose
    // spatially sharded map: ifx_owner_segmentation_begin / _resume carry one segmentation call across its exchange points
    int oseg_state = 0, oseg_nm = 0, oseg_m = 0, oseg_flags = 0, oseg_pending = 0;
    std::vector<uint8_t> oseg_unavail;
    std::vector<int> oseg_cmp, oseg_bbox, oseg_class;
    hipEvent_t oseg_ev = nullptr;
    void* d_kexp = nullptr;            // ifx_owner_knn_export: [cap] float4 (x, y, z, creation number) + [cap] int32 labels
    int gn_begin_folded = 0;         // the start of the next frame-tracker run was done by the model side's last launch
    int labels_stale_all = 1;        // the next segmentation call re-scans the labels of ALL surfels (after create / upload / table eviction); otherwise only what the call can have changed
    int opt_labels_incremental = 1;
    int opt_raster_lds = 0;          // view raster: per-wave depth test in LDS before the global atomics (k_raster_view<true>)
    int opt_view_blocks = 0, opt_clean_blocks = 0, opt_index_blocks = 0;   // grids of the view-list kernels (0: LIST_BLOCKS)
    int icp_resident_blocks = 1024;   // blocks of k_icp_residual the GPU holds at once (occupancy query at tracker allocation: 4 per CU x 256 CUs on MI355X)
    int opt_res_blocks = 0;          // cap on the blocks of the residual half of k_icp_residual (0: one block per 256 pixels)
    int opt_icp_blocks = 0;          // cap on the blocks of a tracker reduction launch; 0 = by image size (ifx_track.hip red_blocks)
    int opt_raster_tiles = -1;       // tiled rasteriser (k_tile_*: key tiles resolved in LDS) instead of global atomics: 0 off, 1 on, -1 by image size (on from 1 Mpixel:
                                     // at 640x480 / 5M surfels the binning passes cost what the LDS tiles save, at 1280x960 / 20M the frame rate gains 12 %)
    // local loop-closure detection (ifx_set_loop_closure): second tracker instance + the INACTIVE prediction images
    int map_external = 0;               // a map was uploaded: surfel times are not bounded by the frames processed
    int lc_enable = 0, lc_count_thresh = 35000;
    float lc_err_thresh = 5e-5f, lc_cov_thresh = 1e-5f;
    DevState* d_m2m = nullptr;
    Pyr m2m;
    float *old_vertex = nullptr, *old_normal = nullptr;        // INACTIVE prediction (IndexMap::oldVertexTex() ...)
    uint8_t *old_image = nullptr, *old_inst = nullptr;
    uint16_t* old_time = nullptr;
    float *act_vertex = nullptr, *act_normal = nullptr;        // predict() of EF/ElasticFusion.cpp:453 (ACTIVE render at the tracked pose, pre-fusion map): kept apart from
    uint8_t *act_image = nullptr, *act_inst = nullptr;         // pred_*, which the end-of-frame predict() rewrites while the model-to-model tracker may still be reading
    uint16_t* act_time = nullptr;
    size_t lc_half = 0;                                        // bytes of one of the two render blocks (act_* / old_* are one allocation of 2 * lc_half)
    int own_tracked_tick = 0;                                  // sharded map with the detection on: the frame side + tracker of this tick ran in phase 300 already
    float* h_lc = nullptr;                                     // pinned: verdict of the last detection (ifx_loop_closure_diag)
    int lc_event_valid = 0;
    float* d_graph = nullptr;           // deformation graph handed in for the next clean (ifx_set_deformation): nodes x 16 floats
    int graph_nodes = 0, graph_is_fern = 0;
    uint8_t* d_inst_gt = nullptr;       // instanceGT of processFrame (H x W bytes) for the frames to come
    int inst_gt_on = 0;
    void* d_fern = nullptr;             // ifx_fern_frame scratch
    float* d_project = nullptr;         // ifx_render_project_map scratch (H x W float4)
    float *d_sample = nullptr, *d_cons = nullptr;   // scratch of ifx_sample_graph_model / ifx_loop_closure_constraints
    ifx_loop_closure_cb lc_cb = nullptr;
    void* lc_user = nullptr;
    ifx_fern_cb fern_cb = nullptr;      // global loop closure (Ferns::findFrame, EF/ElasticFusion.cpp:457-514): runs every frame after predict()
    void* fern_user = nullptr;
    uint8_t* h_fern = nullptr;          // pinned: ifx_fern_frame_async -> ifx_fern_frame_fetch
    hipEvent_t ev_fern = nullptr;
    int fern_pending = 0;
    int in_fern_cb = 0;                 // inside it the "last predict()" is the one at the tracked pose (act* images)
    // device state
    DevState* d_state = nullptr;
    FrameResult* h_result = nullptr;   // pinned
    // host-pointer entry (ifx_process_frame): the pose is read back right behind the tracker, the call returns with it and the frame's map passes finish under the
    // caller's next steps (its copy of the next frame into the staging buffers, typically)
    float* h_pose_early = nullptr;     // pinned, 16 floats
    hipEvent_t ev_pose_early = nullptr;
    int want_early_pose = 0, early_pose_valid = 0, housekeeping_due = 0;
    int prestaged_tick = -1;           // ifx_process_frame_ex (host_entry_async): the frame whose side was enqueued from the staging buffers before the call waited for its predecessor
    int opt_host_entry_async = 0;      // opt-in (see ifx_process_frame_ex): the deferred housekeeping decision must not be separated from its frame by other calls
    float* d_traj = nullptr;           // [max_traj][16] ring: frame f's pose lives in slot f % max_traj (ifx_trajectory returns the last max_traj frames)
    int max_traj = 1 << 16;
    float* d_scratch = nullptr;        // [8][16] poses uploaded by the stage API / inPose / ifx_track_maps (0-1 stage pose + inverse, 2 inPose, 4 track_maps, 6 track_pair)
    int n_traj = 0;
    // map (SoA)
    float *pc = nullptr, *nr = nullptr, *col = nullptr, *tm = nullptr, *ic = nullptr, *votes = nullptr;
    float *pc2 = nullptr, *nr2 = nullptr, *col2 = nullptr, *tm2 = nullptr, *ic2 = nullptr, *votes2 = nullptr; // compaction targets
    uint32_t* upd_owner = nullptr;     // [cap] first-pixel-wins arbitration of the fuse pass
    uint8_t* assoc_vis = nullptr;      // [w + h] which of the texels i - 1, i, i + 1 the association window of pixel column (row) i reads: bits 0..2 (window_taps of the column's texcoord, computed once at create)
    uint32_t *list_v = nullptr, *list_vi = nullptr;   // [8 x list_seg_cap] the cached view lists, flat (lengths: DevState::vl_n): inside / outside the time window
    int view_frame = 0;                 // the frame being enqueued went through the view list (its end-of-frame raster may too)
    int view_block = 0;                 // the pose was replaced after the view-list decision of this frame (pose adoption): the frame takes the per-pass culls
    int age_epoch = 0;                  // first clean pass that saw the store in its present state (ifx_map.hip age_rule_gone); INT_MAX: none yet since an upload / a jump of the clock
    int last_clean_time = 0;            // time of the last clean pass (the age rule a forced scan applies to the slots outside the list)
    int view_dirty = 0;                 // frames ran through the view list since the last forced scan: slots outside it may have outlived the age rule
    int opt_vlist = 1;                  // frame path through the cached view list (0: one cull per pass over all slots, the round-1 path)
    int opt_raster_earlyz = 0;          // view-list rasteriser: skip the atomic when a plain read of the key image already shows a nearer surfel (measured: 136 against 113 us -- the reads cost more than the dropped atomics save)
    uint32_t *list_a = nullptr, *list_b = nullptr, *list_c = nullptr;   // [8 segments x list_seg_cap] work lists (surfel index | flags << 30): raster candidates, clean candidates, kill list
    unsigned int *tile_n = nullptr, *tile_box = nullptr, *tile_pairs = nullptr;   // tiled rasteriser: [4 x TILE_MAX] counters / offsets / fill / flag, per-entry tile box, (tile, entry) pairs
    unsigned int tile_pair_cap = 0;
    void* tile_recs = nullptr;          // [list size] 32-B camera-frame geometry records of the listed surfels (allocated when the tiled path is first used)
    unsigned int* d_list_ctr = nullptr;   // [4 lists][8 segments] lengths, 128 B apart (raster, clean candidates, kill, view list)
    unsigned int list_seg_cap = 0;
    int32_t *labels = nullptr, *labels2 = nullptr;   // [cap] bestIDInEachSurfel per slot
    uint32_t *seq = nullptr, *seq2 = nullptr;        // [cap] creation number per slot (ascending; = the slot index of an unsharded, compacted map)
    int* scan_flags = nullptr;         // [max(cap,P)]
    int* scan_block = nullptr;
    int* scan_out = nullptr;
    // frame buffers
    uint8_t* rgb = nullptr;
    uint16_t *depth_raw = nullptr, *depth_filt = nullptr;
    float *dm = nullptr, *dmf = nullptr;
    uint8_t* rgb_stage = nullptr; uint16_t* depth_stage = nullptr; // pinned staging
    const uint8_t* late_rgb_src = nullptr;   // ifx_process_frame: the caller's colour image, copied into rgb_stage by the frame side behind the depth transfer
    // index map
    unsigned long long *key_index = nullptr, *key_splat = nullptr, *key_ids = nullptr;
    unsigned long long* key_both = nullptr;   // pixels a surfel covers in BOTH the splat and the id render: one atomic instead of two
    uint32_t* index_id = nullptr;
    float *index_vc = nullptr, *index_ct = nullptr, *index_nr = nullptr, *index_tap = nullptr;
    // predictions
    float *pred_vertex = nullptr, *pred_normal = nullptr, *pred_conf = nullptr;
    uint8_t *pred_image = nullptr, *pred_inst = nullptr;
    uint16_t* pred_time = nullptr;
    float *fill_vertex = nullptr, *fill_normal = nullptr;
    uint8_t* fill_image = nullptr;
    int32_t *ids_after = nullptr, *ids_tmp = nullptr;
    // association scratch (per pixel)
    uint32_t* assoc_target = nullptr;  // 0xFFFFFFFF none, 0xFFFFFFFE new, else surfel id
    unsigned long long* assoc_key = nullptr;   // [ceil(w/2) * ceil(h/2)] sharded map: best owned candidate per measurement pixel (k_associate -> exchange MIN -> k_assoc_decode)
    float *meas_pc = nullptr, *meas_nr = nullptr, *meas_col = nullptr;
    // tracker
    Pyr pyr;
    int* res_partials = nullptr;    // [blocks][2]
    float* d_out29 = nullptr;
    int res_rows = 1024;                // rows of res_partials (blocks of the residual pass)
    unsigned int* d_ticket = nullptr;   // last-block ticket of k_rgb_step_solve
    // instance layer
    int32_t inst_class[IFX_NUM_INSTANCES];
    float inst_color[IFX_NUM_INSTANCES];
    float* d_inst_color = nullptr;
    uint8_t *d_masks = nullptr, *d_masks_ori = nullptr, *d_unavail = nullptr; size_t masks_cap = 0;   // [nm][P] working masks, masks before clean-overlap, [nm] flags
    int* d_ff_label = nullptr; size_t ff_cap = 0;   // flood-fill labels + region counters + per-mask bookkeeping
    uint16_t* d_pdm = nullptr;
    int* d_bbox = nullptr;             // [96*4 + maxmasks*4]
    int* d_inst_stats = nullptr;       // [96*2]
    int* d_clean_list = nullptr;
    void* d_segctl = nullptr; void* h_segctl = nullptr;          // SegCtl of the device-side segmentation call + its pinned mirror (+ 256 verdict bytes)
    uint8_t* h_masks_stage = nullptr; size_t h_masks_cap = 0;    // pinned staging of the caller's masks
    int opt_seg_device = 1;            // segmentation call without the host in the middle (0: the host-driven schedule of round 2)
    int opt_ff_rounds = 0;             // relaxation launches of the flood fill's fixed schedule (0: 24)
    int last_seg_frame = -1;
    int seg_counts_valid = 0;          // h_result->seg_counts describe the current ids_after / votes
    int clean_times = 0;
    int* d_knn = nullptr; size_t knn_cap = 0;   // grid + sort buffers of the kNN smoothing (ifx_knn.hip), allocated on first use
    void* slic = nullptr;              // superpixel buffers (ifx_slic.hip), allocated on first use
    // timing
    hipEvent_t ev_stage[8];
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> stage_pending;
    double stage_ms[5] = {0, 0, 0, 0, 0};   // track, fuse, instance (main-stream span of the calls), preprocess (side stream), superpixels run ahead (side stream)
    // superpixels of the announced next frame, run ahead on the side stream (ifx_superpixel_ahead): the tick they belong to (-1: none), the event behind them
    int slic_ahead_busy = 0;            // a run may still be executing on the side stream (whoever uses the superpixel buffers next queues behind its event)
    int slic_ahead_tick = -1;
    hipEvent_t ev_slic_ahead = nullptr;
    int slic_ahead_runs = 0, slic_ahead_used = 0;
    std::map<std::string, int> kname_id;
    std::vector<std::string> knames;
    std::vector<KernelTiming> ktimes;
    std::vector<PendingEvent> kpending;
    std::vector<hipEvent_t> event_pool;
};
static inline IdMap ifx_idmap(const ifx* h) { IdMap m; m.seq = h->seq; m.own_n = h->own ? h->own_g : 0; return m; }

#define HIPCHK(h, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return IFX_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

// kernel launch with optional per-kernel event timing
hipEvent_t ifx_event_get(ifx* h);
void ifx_ktime_begin(ifx* h, const char* name, hipEvent_t* a);
void ifx_ktime_end(ifx* h, const char* name, hipEvent_t a);

// Forget a tracker run enqueued ahead (the caller is about to change something it read).
static inline void ifx_drop_tracked(ifx* h) { h->tracked_ahead = 0; h->ids_view_ok = 0; }

#define LAUNCH(h, name, grid, block, kernel, ...)                                                  \
    do {                                                                                           \
        hipEvent_t ea_ = nullptr;                                                                  \
        if ((h)->opt_kernel_timing) ifx_ktime_begin((h), name, &ea_);                              \
        hipLaunchKernelGGL(kernel, grid, block, 0, (h)->cur, __VA_ARGS__);                      \
        if ((h)->opt_kernel_timing) ifx_ktime_end((h), name, ea_);                                 \
    } while (0)

#define LAUNCH_SMEM(h, name, grid, block, smem, kernel, ...)                                      \
    do {                                                                                           \
        hipEvent_t ea_ = nullptr;                                                                  \
        if ((h)->opt_kernel_timing) ifx_ktime_begin((h), name, &ea_);                              \
        hipLaunchKernelGGL(kernel, grid, block, smem, (h)->cur, __VA_ARGS__);                   \
        if ((h)->opt_kernel_timing) ifx_ktime_end((h), name, ea_);                                 \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// stage entry points implemented across the .hip files
void ifx_slic_free(ifx* h);
void ifx_knn_free(ifx* h);
void ifx_knn_free_all(ifx* h);
int ifx_knn_vote(ifx* h, int32_t* d_nbr_out);
int ifx_ensure_masks(ifx* h, size_t bytes);
int ifx_preprocess(ifx* h);                                   // bilateral + metric
int ifx_tracker_init_first(ifx* h);
int ifx_tracker_run_frame(ifx* h, int commit = 1, int keep_last = 0);                            // model pyramid + GN loops (all on device); the frame side is in the slot
int ifx_tracker_commit(ifx* h);                                // publish the pose of a tracker run that was enqueued ahead
int ifx_tracker_model_side(ifx* h, int fold_begin = 0);                           // model pyramid from the prediction of the previous frame
int ifx_tracker_frame_side(ifx* h, int first, double* so3_acc = nullptr, unsigned int* so3_ticket = nullptr);   // frame pyramids + SO(3) pre-alignment of the bound slot
int ifx_tracker_camera_ahead(ifx* h, int cam, const uint8_t* d_rgb, const uint16_t* d_depth);                // tracker of a parked camera's next frame on the instance of its own (the frame is in slot 2, h->cur is the stream)
void ifx_bind_slot(ifx* h, int s);
int ifx_housekeeping(ifx* h);                                  // tombstone compaction decided from the last frame result
int ifx_enqueue_hinted_frame_side(ifx* h);                     // frame side of the announced next frame (no-op without a hint)
int ifx_map_init_first(ifx* h);
int ifx_map_frame(ifx* h);                                    // index -> fuse -> index -> clean -> ids
int ifx_ids_ensure(ifx* h);                                   // the whole id image, if the last frame rendered only the sampled lattice
int ifx_owner_ids_begin_impl(ifx* h);                         // sharded map: 1 = the shard's id render is enqueued and waits for its key exchange, 0 = the image is whole already
int ifx_owner_ids_resume_impl(ifx* h);
int ifx_own_lattice(const ifx* h);                            // entries of the id lattice (one per 10 x 10 pixels)
int ifx_vlist_reap(ifx* h);                                   // forced view-list scan: applies the age rule to the slots outside the list (before any whole-map consumer)
int ifx_map_sharded_phase(ifx* h, int phase, bool first_frame);
int ifx_map_predict_loop_closure(ifx* h);                     // predict() at the tracked pose + INACTIVE prediction (old* images)
int ifx_tracker_alloc_m2m(ifx* h);
int ifx_tracker_loop_closure(ifx* h);                         // model-to-model tracking + gates, after ifx_map_predict_loop_closure
int ifx_tracker_m2m_begin(ifx* h);
int ifx_map_predict(ifx* h);                                  // splat + fill-in + dense flag
int ifx_scan_exclusive(ifx* h, const int* d_flags, int n, int* d_out, int* d_total /*device ptr or null*/);
int ifx_alloc_tracker(ifx* h);
void ifx_free_tracker(ifx* h);
int ifx_alloc_instance(ifx* h);
// ifx_comm.hip: the collectives of a sharded map inside the library
int ifx_comm_exchange(ifx* h, int phase);
int ifx_comm_ready(ifx* h);
void ifx_comm_free(ifx* h);
void ifx_free_instance(ifx* h);
