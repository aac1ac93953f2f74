/*
 * include/ifx_c_api.h -- C-ABI of libifx.so, the MI355X-native (HIP, gfx950) implementation of
 * InstanceFusion's per-frame dense surfel pipeline.
 *
 * This header is the drop-in boundary.  Every entry point names the reference
 * interface it replaces.  Citation prefixes: EF/ = elasticfusionpublic/Core/src/, IF/ = src/ of the
 * reference tree.  Plain C types only: no HIP, torch or C++ types cross the boundary.  Pointers
 * named d_* are device (HBM) pointers of the handle's GPU, every other pointer is host memory.
 *
 * Error behaviour: every function that can fail returns 0 on success and a negative IFX_E_* code
 * on failure; ifx_last_error() gives the message.  Nothing calls exit() (the reference's
 * cudaSafeCall / gpuErrChk do: EF/Cuda/convenience.cuh:64-71, IF/Core/InstanceFusionCuda.cu:11-20).
 *
 * Threading: one ifx_t = one host thread + its own HIP streams (main, frame side, loop-closure tracker); the handle is not thread-safe.
 */
#ifndef IFX_C_API_H_
#define IFX_C_API_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFX_NUM_INSTANCES 96   /* IF/main.cpp:31-44 (instanceNum) */
#define IFX_VOTE_FLOATS 48     /* two int16 counters per float, IF/Core/InstanceFusionCuda.cu:22-39 */

enum {
    IFX_OK = 0,
    IFX_E_INVALID = -1,   /* bad argument */
    IFX_E_HIP = -2,       /* a HIP runtime call failed */
    IFX_E_CAPACITY = -3,  /* the surfel store is full */
    IFX_E_STATE = -4      /* call not valid in the current state */
};

/* Replaces the ElasticFusion constructor arguments (EF/ElasticFusion.h:47-62, values used by
 * IF/map_interface/ElasticFusionInterface.cpp:43-45) and the Resolution/Intrinsics singletons
 * (IF/main.cpp:46-47). */
typedef struct ifx_config {
    int32_t width, height;       /* 640 x 480 */
    float fx, fy, cx, cy;        /* 528, 528, 320, 240 */
    int32_t time_delta;          /* 200 */
    float confidence;            /* 10 */
    float depth_cut;             /* 12 m */
    float max_depth_processed;   /* 20 m (EF/ElasticFusion.cpp:73) */
    float icp_weight;            /* 10 */
    int32_t pyramid;             /* 1 */
    int32_t fast_odom;           /* 0 */
    int32_t so3;                 /* 1 */
    int32_t max_surfels;         /* capacity of the surfel store (reference: 1536^2, EF/GlobalModel.cpp:22-23) */
    int32_t device;              /* HIP device ordinal */
    int32_t n_ranks, rank;       /* spatially sharded map: this handle stores shard `rank` of `n_ranks` (0 or 1: the whole map; -1: the sharded path with one rank); see ifx_owner_frame_phase */
} ifx_config;

typedef struct ifx ifx_t;

/* ---- construction (replaces `new ElasticFusion(...)`, IF/map_interface/ElasticFusionInterface.cpp:43-45) */
int ifx_create(const ifx_config* cfg, ifx_t** out);
void ifx_destroy(ifx_t* h);
const char* ifx_last_error(ifx_t* h);
/* Global (handle-free) message for failures of ifx_create itself. */
const char* ifx_global_error(void);

/* ---- frame entry.  Replaces ElasticFusion::processFrame (EF/ElasticFusion.h:75-82,
 * EF/ElasticFusion.cpp:269-720) and ElasticFusionInterface::ProcessFrame
 * (IF/map_interface/ElasticFusionInterface.h:129-130).
 * rgb: H*W*3 u8 row-major, depth: H*W u16 millimetres (0 invalid).  in_pose16: NULL to track, or a
 * row-major 4x4 camera-to-world pose to use instead of tracking.  out_pose16 (may be NULL) receives
 * currPose.  Returns 0 ok, 1 lost (never in this configuration: reloc=false), <0 error.
 * The caller's buffers are copied during the call, which returns when the whole frame is done.
 * Option "host_entry_async" 1 (for a host whose loop is ifx_process_frame after ifx_process_frame -- a log replay without masks): the call returns when the frame's
 * POSE is known (it is read back right behind the tracker); the frame's map passes finish under the caller's next steps -- its copy of the next frame, typically
 * -- like the reference's GL work after processFrame has issued it.  Every accessor of the map, the images or the frame result waits for them, exactly as after
 * ifx_enqueue_frame_device; the housekeeping decision of the frame (compaction) is taken at the start of the next ifx_process_frame, from the same numbers; near
 * the capacity, with loop-closure detection on, with camera contexts, with an external pose and for the first frame the call synchronises fully. */
int ifx_process_frame(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp,
                      const float* in_pose16, float weight_mult, float* out_pose16);
/* The complete argument list of ElasticFusion::processFrame (EF/ElasticFusion.h:75-82).
 * inst_table: `smallInstanceTable` (96 x 5 ints, may be NULL).  Its only consumer in the reference is Ferns::findFrame
 *   (EF/ElasticFusion.cpp:468), host code that lives above this boundary (instancefusion_amd/host/ifx_ferns.hpp), so the
 *   library accepts it and does not read it -- which is why ifx_process_frame omits it.
 * bootstrap != 0: in_pose16 (required) is a GUESS, not a replacement: the model maps are placed with the current pose, then
 *   currPose = currPose * inPose is the tracker's initial estimate, and the velocity weighting compares the tracked pose with
 *   the pose before the guess (EF/ElasticFusion.cpp:330-356, :433). */
int ifx_process_frame_ex(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp, const int32_t* inst_table,
                         const float* in_pose16, float weight_mult, int bootstrap, float* out_pose16);
/* Same, with the frame already resident in HBM and no host synchronisation: the call only
 * enqueues work on the handle's stream.  Poses are appended to the device-side trajectory log. */
int ifx_enqueue_frame_device(ifx_t* h, const uint8_t* d_rgb, const uint16_t* d_depth,
                             int64_t timestamp, const float* in_pose16, float weight_mult);
/* Optional one-frame look-ahead for replayed streams (the reference's log readers know the next frame,
 * IF/utilities/RawLogReader.cpp:66-115): enqueues the part of the NEXT frame that depends only on its
 * images (copy, bilateral filter, frame pyramids, SO(3) pre-alignment against the current frame's
 * image) on a side stream, under the current frame's tracking and map passes.  Call it after
 * ifx_enqueue_frame_device for the current frame and pass the same pointers to the next
 * ifx_enqueue_frame_device; results are identical with or without it. */
int ifx_prefetch_frame_device(ifx_t* h, const uint8_t* d_rgb_next, const uint16_t* d_depth_next);
/* The same look-ahead announced BEFORE the current frame is enqueued: the next ifx_enqueue_frame_device
 * call places the announced frame's image-only work itself (behind the coarse pyramid levels of its own
 * tracker, where the GPU is least busy).  Preferred over ifx_prefetch_frame_device. */
int ifx_hint_next_frame_device(ifx_t* h, const uint8_t* d_rgb_next, const uint16_t* d_depth_next);
/* The same announcement with HOST pointers, for the reference-shaped entry: call it BEFORE ifx_process_frame of the current
 * frame, with the frame the reference's log reader would return next (IF/utilities/RawLogReader.cpp:66-115,
 * IF/main.cpp:108-307).  The buffers are borrowed for this call only (the frame is copied into pinned staging here); its
 * transfer and image-only work run on the side stream under the current frame, its tracker is parked behind the current
 * frame, and the next ifx_process_frame -- which must pass the SAME pointers; anything else and the announcement is simply
 * ignored -- finds all of that done.  Results are identical with or without it, PROVIDED the two buffers still hold at that
 * ifx_process_frame what they held here: the frame is matched by pointer identity and what was staged at the hint is what is
 * processed (a caller that refills the same buffers in between must not announce them).  Single-stream, unsharded handles. */
int ifx_hint_next_frame(ifx_t* h, const uint8_t* rgb_next, const uint16_t* depth_next);
/* ---- sharded projection for large maps.  Every rank (one process per GPU) holds the full map replica and is
 * fed the same frames and masks; the passes that stream the whole surfel store with one atomic per visible surfel (index map
 * x2, splat + id raster) only handle this rank's slice of the slots, and the ranks combine their key images between the four
 * phases of a frame by an element-wise UNSIGNED 64-bit minimum (RCCL all-reduce over xGMI; instancefusion_amd/sharded.py does
 * it through torch.distributed).  Exchange after phase 0 and after phase 1: key_index; after phase 2: key_splat, key_ids,
 * key_both; phase 3 needs none.  The replicas stay bit-identical, and identical to a single-GPU run. */
int ifx_set_shard(ifx_t* h, int rank, int nranks);
int ifx_sharded_frame_phase(ifx_t* h, int phase, const uint8_t* d_rgb, const uint16_t* d_depth);   /* pointers used by phase 0 */
int ifx_key_images(ifx_t* h, void** key_index, void** key_splat, void** key_ids, void** key_both, int64_t* n_pixels);
/* The handle's HIP streams (hipStream_t): work enqueued on the main stream between two phases (the key exchange) is
 * ordered with the phases without any host synchronisation. */
int ifx_stream_handles(ifx_t* h, void** main_stream, void** side_stream);
/* ---- spatially sharded map (BASELINE configurations 4 and 5): a handle created with ifx_config::n_ranks = G > 1 STORES only
 * the surfels it owns -- owner = Morton code of the 8 cm voxel of the position a surfel was created (or uploaded) at, mod G (ifx_owner_of) --
 * i.e. 1 / G of the map.  One process per GPU, every rank fed the same frame.  A frame is eight calls of ifx_owner_frame_phase (phase 0..7,
 * the image pointers are used by phase 0); after phase p (0..6) the caller reduces, across the ranks, the device buffers ifx_owner_exchange(p)
 * lists: ops 0 = element-wise MINIMUM of unsigned 64-bit words (key images: depth | creation number), ops 1 = SUM of 32-bit words
 * (attribute blocks with disjoint supports: the winner's rank writes a pixel, the others hold zeros); op 6 (option "own_key_rs", exchanges 0 and 2) = the MINIMUM of op 0 of which
 * only the low 32 bits of every word are wanted back -- the buffer must come back as (uint64) low word, all-ones words whole: any transport that computes op 0 and strips the
 * high words will do, the library's own runs a reduce-scatter and an all-gather of the low words (12 instead of 16 bytes per key and link direction).  The library does it itself once it holds a
 * communicator (ifx_owner_process_frame_device below); the phase / exchange pair stays for hosts with their own transport and for the emulation tests.  Poses, images and -- merged by ifx_map_seq -- the map equal the unsharded run bit for
 * bit.  Per-surfel work (projections, fusion update, clean, votes, label scan) is sharded, per-pixel work (tracking, association, the mask
 * pipeline) replicated; segmentation calls go through ifx_owner_segmentation_begin / _resume, the kNN smoothing through ifx_owner_knn_export /
 * _vote.  With the local loop-closure detection enabled (ifx_set_loop_closure; the deformation callbacks are not offered in this mode) a frame has two
 * more phases IN FRONT of phase 0 -- 300: frame side, tracker and the local ACTIVE + INACTIVE renders at the tracked pose; 301: the owners' winners of both --
 * each followed by the exchange ifx_owner_exchange(300 | 301) lists; phase 0 then runs the model-to-model tracker and the gates on the exchanged renders,
 * replicated (the same verdict on every rank, ifx_loop_closure_diag).  Both are no-ops while the detection is off or nothing can be inactive yet. */
int ifx_owner_frame_phase(ifx_t* h, int phase, const uint8_t* d_rgb, const uint16_t* d_depth);
/* ---- the same frame as ONE call, the collectives enqueued by the library itself (instancefusion_amd/csrc/ifx_comm.hip).  The reference has no
 * counterpart (one GPU, IF/main.cpp:75); BASELINE.json's north star asks for "RCCL all-reduce over xGMI ... all-to-all for cross-shard surfel
 * reprojection" with the host in C++.  The library holds a RCCL communicator (librccl.so.1 loaded with dlopen on first use: a single-GPU process
 * never maps it) and issues the exchange of every phase on the handle's main stream, between the kernels of two phases: no host round trip inside a
 * frame, six collectives per frame (u64 MIN of key images and of the association verdicts, int32 SUM of the winners' attribute blocks), 80 bytes per pixel.
 *   ifx_comm_unique_id(out128)       ncclGetUniqueId on one rank; the host hands the 128 bytes to the other ranks (MPI, a file, a socket, torch.distributed)
 *   ifx_owner_init_comm(h, id128)    ncclCommInitRank(n_ranks, id, rank) on the handle's device -- collective: every rank calls it
 *   ifx_owner_set_comm(h, comm)      adopt a ncclComm_t the host already owns (size / rank must match the handle's); NULL: back to caller-driven exchanges
 *   ifx_owner_process_frame_device   phases 0..7 + exchanges, device pointers, no synchronisation (poses: ifx_trajectory / ifx_get_pose)
 *   ifx_owner_process_frame          ElasticFusion::processFrame's shape (EF/ElasticFusion.h:75-82): host pointers, one synchronisation, currPose back
 *   ifx_owner_predict                ElasticFusion::predict outside a frame (after ifx_map_upload / ifx_set_pose)
 *   ifx_owner_process_segmentation   InstanceFusion::processInstance (IF/Core/InstanceFusion.cpp:655-1067): begin / resume with the exchanges inside; flags bit 0 also runs
 *   ifx_owner_knn_vote_colour        flannKnnVoteSurfelMap (:1070-1163): all-gather of every rank's slots (20 B each), exact 10-NN of the owned surfels
 *   ifx_owner_exchange_stats         out2 = collectives enqueued, bytes handed to them since the last reset (in all-reduce-equivalent bytes: a ring all-reduce of S bytes moves
 *                                    2 S (G - 1) / G per link direction; the reduce-scatter + all-gather pair of op 6 counts (8 + 4) / 2 bytes per key)
 * ifx_config::n_ranks = -1 creates a WORLD OF ONE on this path (creation-number ids, owner filter, every exchange point as a one-rank collective): what
 * `bench.py --sharded --gpus 1` and the single-GPU RCCL test run.
 * Creation numbers (the ids of a sharded map) are unsigned 32-bit and never renumbered: a handle reports IFX_E_CAPACITY once 2^32 - 2^20 of them have been
 * handed out (at most P / 4 per frame: > 50 000 frames at 640 x 480 in the worst case, millions in practice). */
/* ---- K streams into ONE map (BASELINE configuration 5; the reference has one stream, IF/main.cpp:75).  Semantics of a FRAME SET (one frame per camera): the K frames
 * are processed one after the other, in camera order, on the one map; a camera tracks against the prediction rendered at the end of ITS last frame.
 *   ifx_camera_count(h, K)        K camera contexts on this handle (pose block, prediction + fill-in, the last frame's intensity pyramid, id image)
 *   ifx_camera_select(h, c)       park the current camera's context, bring camera c's in (enqueue-only, between frames); a camera selected for the first time
 *                                 starts as a copy of the current one: give it its pose (ifx_set_pose) or an external pose for its first frame.  Between two
 *                                 cameras that both have a context the prediction / fill-in / id images change hands by pointer (no copies): device pointers
 *                                 obtained before the switch -- ifx_ids_after, ifx_owner_exchange lists -- are stale after it; ask again
 * On a spatially sharded map (every rank holds the K contexts and is fed all K streams):
 *   ifx_owner_set_frame_pose      the next frame takes this pose instead of tracking (the in_pose of the unsharded entry points)
 *   ifx_owner_set_tracking_rank   only this rank tracks the frames to come -- stream k on GPU k, no tracker collective; the others run the frame side,
 *                                 and the tracked pose block reaches them by a broadcast (phase 310 + ifx_owner_exchange(h, 310): op 4 | root << 8), which
 *                                 ifx_owner_process_frame_device issues itself.  -1: every rank tracks (replicated).  While a tracking rank is set the prediction
 *                                 rendered at the end of the frame has one consumer, that rank's tracker: exchange 5 lists the prediction block with op 5 | root << 8
 *                                 (int32 SUM to the root only: ncclReduce) and the 16-byte vote-mass tail with op 1 (to everybody); on the other ranks the
 *                                 prediction / fill-in images of that camera are partial and must not be read.
 * tests/test_gpu_parity.py::test_config5_two_streams_one_sharded_map: K = 2 cameras, G = 2 ranks, bit-identical to one GPU. */
int ifx_camera_count(ifx_t* h, int n_cameras);
int ifx_camera_select(ifx_t* h, int cam);
int ifx_owner_set_frame_pose(ifx_t* h, const float* pose16);
int ifx_owner_set_tracking_rank(ifx_t* h, int rank);
/* K streams, camera `cam` tracked by rank `tracking_rank` only: that rank enqueues the tracker of camera cam's NEXT frame now, on the handle's third stream, reading the
 * camera's PARKED context (prediction, fill-in, last intensity pyramid, pose block: final since the camera's last frame, untouched until its next) -- so rank k tracks camera k
 * under the other cameras' map phases instead of at the head of camera k's frame.  When that frame arrives with exactly these device pointers the parked pose block is
 * committed instead of a tracker run (same inputs and arithmetic: same pose).  Call it on every rank once camera cam's context is parked (another camera selected); the
 * other ranks return at once.  cam = -1: the number of frames whose tracker came from a run ahead so far.  (The reference has one stream and one GPU, IF/main.cpp:75.) */
int ifx_owner_track_ahead(ifx_t* h, int cam, int tracking_rank, const uint8_t* d_rgb, const uint16_t* d_depth);
int ifx_comm_unique_id(uint8_t* out128);
int ifx_owner_init_comm(ifx_t* h, const uint8_t* unique_id128);
int ifx_owner_set_comm(ifx_t* h, void* nccl_comm);
int ifx_owner_process_frame_device(ifx_t* h, const uint8_t* d_rgb, const uint16_t* d_depth, int64_t timestamp);
int ifx_owner_process_frame(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp, float* out_pose16);
int ifx_owner_predict(ifx_t* h);
int ifx_owner_process_segmentation(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks, const int32_t* class_ids, int nm, int frame, int flags);
int ifx_owner_knn_vote_colour(ifx_t* h);
int ifx_owner_exchange_stats(ifx_t* h, int64_t* out2, int reset);
/* ranks of the communicator the handle's collectives run on, as RCCL itself counts them (ncclCommCount of the communicator created by ifx_owner_init_comm or adopted
 * by ifx_owner_set_comm); 0: none yet.  What a benchmark line quotes as the number of GPUs its exchanges really crossed (no counterpart in the reference: one GPU). */
int ifx_owner_comm_ranks(ifx_t* h);
int ifx_owner_exchange(ifx_t* h, int phase, void** ptrs, int64_t* bytes, int32_t* ops, int max_n);
/* ElasticFusion::predict on the sharded map outside a frame (after ifx_map_upload / ifx_set_pose): step 0, exchange as after phase 4,
 * step 1, exchange as after phase 5, step 2. */
int ifx_owner_predict_phase(ifx_t* h, int step);
int ifx_owner_of(const float* xyz, int n, int n_ranks, int32_t* out);
/* InstanceFusion::processInstance (src/Core/InstanceFusion.cpp:655-1067) on a sharded map.  Every rank passes the same masks; what depends on a surfel's
 * votes or position is computed by its owner and merged at exchange points: _begin / _resume return 1 while one is pending -- reduce the buffers
 * ifx_owner_exchange(h, 200, ...) names (ops: 1 sum of 32-bit words, 2 minimum of signed 32-bit words, 3 maximum of signed 32-bit words), then call
 * _resume -- and 0 when the call is complete (labels of the owned surfels: ifx_labels).  flags: bit 1 superpixel refinement; the kNN smoothing (bit 0)
 * is a separate call on a sharded map (ifx_owner_knn_export / _vote).  The whetherDoSegmentation sums of a frame are complete after ifx_owner_frame_phase(h, 7, ...) (phase 6 leaves
 * the vote mass of the owned surfels in the buffer ifx_owner_exchange(h, 6, ...) names; phase 7 publishes the frame result). */
int ifx_owner_segmentation_begin(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks, const int32_t* class_ids, int nm, int frame, int flags);
int ifx_owner_segmentation_resume(ifx_t* h);
/* Option "own_lazy_ids" (ifx_set_option, sharded map only; no counterpart in the reference, whose id image -- IndexMap::renderSurfelIds, src/Core/InstanceFusion.cpp:402-466 --
 * never leaves one GPU): a frame draws and exchanges the id keys of the 10 x 10 lattice whetherDoSegmentation samples only, so exchange 4 carries
 * [splat keys | lattice keys | word] = 8 P + 8 ceil(w/10) ceil(h/10) + 8 bytes instead of 16 P + 8.  Whoever reads the whole id image completes it: a segmentation call does so
 * at an exchange point of its own in front of the others (nothing changes for the caller of _begin / _resume); ifx_image_download("ids_after"), ifx_ids_after and
 * ifx_camera_select do it in place when the library holds the communicator (ifx_owner_init_comm / _set_comm: every rank makes the same call), and return IFX_E_STATE
 * otherwise -- a caller that runs the exchanges itself calls ifx_owner_ids_begin on every rank first (1: reduce the buffer ifx_owner_exchange(h, 200, ...) names, op 0, then
 * ifx_owner_ids_resume; 0: the image is whole already). */
int ifx_owner_ids_begin(ifx_t* h);
int ifx_owner_ids_resume(ifx_t* h);
/* Option "own_track_rows" (sharded map, the library's communicator, no single tracking rank): the tracker's two reductions of an iteration (EF/Cuda/reduce.cu:257-490 ICP,
 * :494-678 photometric) run over this rank's share of the pixel blocks and the 2 x 29 exact sums are all-reduced in f64 on the handle's stream, one workgroup solves
 * (EF/Utils/RGBDOdometry.cpp:541-583).  Same poses bit for bit (the sums are exact); 38 small collectives more per frame.  Off by default: DESIGN.md section 7. */
/* InstanceFusion::flannKnnVoteSurfelMap (src/Core/InstanceFusion.cpp:1070-1163) on a sharded map: exact 10-NN over ALL surfels needs every rank's
 * positions.  ifx_owner_knn_export hands out this rank's slots as device arrays -- points[n] float4 (x, y, z, creation number; x = NaN: dead slot),
 * labels[n] int32 (bestIDInEachSurfel) --, the caller all-gathers both in rank order (16 + 4 bytes per slot, once per smoothing, i.e. every > 40
 * frames), and ifx_owner_knn_vote searches the gathered set for the surfels of this rank (own_offset = where this rank's export starts in it)
 * and recolours them.  Ties in distance go to the lower creation number: the neighbour sets of the unsharded map. */
int ifx_owner_knn_export(ifx_t* h, void** d_points, void** d_labels, int* n);
int ifx_owner_knn_vote(ifx_t* h, const void* d_all_points, const void* d_all_labels, int n_all, int own_offset);
/* creation numbers (uint32) of the live surfels in the order of ifx_map_download; returns the count */
int ifx_map_seq(ifx_t* h, uint32_t* out, int max_n);
/* ---- display / export branch of the instance layer.
 * ifx_map_bounding_boxes: InstanceFusion::computeMapBoundingBox (IF/Core/InstanceFusion.cpp:1261-1457; kernels testAllSurfelNormalVote,
 *   setGroundandInstanceCoordinate, testAllSurfelFindBBox, IF/Core/InstanceFusionCuda.cu:1555-1917): normal votes on an 18 x 36 sphere grid,
 *   ground normal = the cell with most votes, ground frame and one frame per instance (heading from the instance's own votes), boxes as
 *   min / max of the coordinates scaled by `ratio` (reference: 1e6) and truncated, in the ground frame (bbox_type 1) or the instance's
 *   (0); boxes96x6 = {minX, maxX, minY, maxY, minZ, maxZ} per instance, +-999999999 / ratio for an instance without surfels.  The optional
 *   outputs: ground normal (3), ground frame (16, row-major), instance frames (96 x 16), the 648 vote counts of the whole map.
 * ifx_instance_point_cloud: InstanceFusion::getInstancePointCloud (:1459-1590; mapCountInstanceByInstColor, getSurfelToInstanceBuffer,
 *   IF/Core/InstanceFusionCuda.cu:1920-2066): surfels per instance (counts96) and, for inst >= 0, its records {slot, x, y, z and normal in
 *   the box frame, r, g, b} (10 floats) in slot order; returns the number of records written. */
int ifx_map_bounding_boxes(ifx_t* h, int bbox_type, float ratio, float* boxes96x6, float* ground_normal3, float* gc_matrix16, float* inst_matrix96x16,
                           int32_t* ground_votes648);
int ifx_instance_point_cloud(ifx_t* h, int bbox_type, int32_t* counts96, int inst, float* out10, int max_records);
/* Diagnostics of the cached view list ("View list"): out4 = entries inside the time window, stable entries
 * outside it, scans of the store so far, frames since the last scan. */
int ifx_view_list_stats(ifx_t* h, int32_t* out4);
int ifx_sync(ifx_t* h);
/* getCurrPose(), EF/ElasticFusion.cpp:1346 / ElasticFusionInterface.h:112-115 (synchronises) */
int ifx_get_pose(ifx_t* h, float* out_pose16);
int ifx_tick(ifx_t* h);
/* Trajectory log: one pose per processed frame (ResultModel.freiburg, EF/ElasticFusion.cpp:99-136), kept on the device as a ring of
 * 65536 frames: returns the LAST min(frames processed, 65536, max_frames) poses, oldest first. */
int ifx_trajectory(ifx_t* h, float* out_poses16, int max_frames);
/* diag[8]: lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count,
 * velocity weighting, fill-in flag (EF/Utils/RGBDOdometry.h:65-70). */
int ifx_tracker_diag(ifx_t* h, float* diag8);
/* Pyramid levels that the persistent Gauss-Newton kernel (option gn_persist) could not finish because a meeting of its workgroups did not happen (the grid was
 * not co-resident: other work held the GPU) and that one workgroup re-ran alone inside the same frame -- same sums, same solve, same pose, only slower.  A count since
 * ifx_create, >= 0; negative: error.  Replaces nothing of the reference (its tracker reads every reduction back, EF/Utils/RGBDOdometry.cpp:461-583): a diagnostic
 * of this implementation's schedule.  Waits for the frames in flight. */
int ifx_tracker_fallbacks(ifx_t* h);
/* Run-time guard of the tracker's exact sums: the number of reductions, since ifx_create, whose diagonal totals left the range in which every addition of the normal-equation
 * sums is exact (2^53 units of the entry's grid; tested at half of it).  While it is 0 the sums -- and with them every pose -- are independent of the order in which blocks and
 * atomics arrive and equal to the CPU oracle's; a non-zero count names frames (saturated edges at near range, magnitudes ~2^7 above what real RGB-D data produces) whose poses
 * may differ from run to run in the last bit.  >= 0; negative: error.  Replaces nothing of the reference (its reductions are f32 trees whose shape depends on the GPU,
 * EF/Utils/GPUConfig.h:53-137: no order-independence to guard).  Waits for the frames in flight. */
int ifx_tracker_range_exceeded(ifx_t* h);

/* ---- local loop-closure DETECTION (the closeLoops / countThresh / errThresh / covThresh constructor arguments, EF/ElasticFusion.h:48-51;
 * EF/ElasticFusion.cpp:453-566 with no fern match).  When enabled every tracked frame also runs predict() at the new pose, the INACTIVE
 * prediction (surfels not seen for time_delta frames), the model-to-model tracker (RGBDOdometry modelToModel: active render against
 * inactive render, ICP weight 10, no SO(3)) and the gates covariance diagonal <= cov_thresh, lastICPCount > count_thresh, lastICPError <
 * err_thresh.  The reference then deforms the map (deformation graph) and adopts the estimated pose: the GPU half of that is provided by
 * the hooks below (ifx_set_loop_closure_callback ... ifx_adopt_estimated_pose), the graph optimiser is host code above this boundary
 * (instancefusion_amd/host/ifx_deformation.hpp).  Without a callback an accepted candidate is counted and reported and the frame
 * continues with the tracked pose; a run whose candidate count stays 0 is what the reference computes with closeLoops = true.
 * out24: 0 model-to-model ran (0: nothing inactive in view), 1 pixels of the inactive render, 2 lastICPError, 3 lastICPCount, 4 covOk,
 * 5 accepted, 6..21 estimated pose (row-major 4x4), 22 largest diagonal covariance entry, 23 candidates accepted so far. */
int ifx_set_loop_closure(ifx_t* h, int enable, int count_thresh, float err_thresh, float cov_thresh);
int ifx_loop_closure_diag(ifx_t* h, float* out24);
/* ---- hooks for the deformation that follows an accepted candidate (EF/ElasticFusion.cpp:566-613).  The graph OPTIMISATION is host code
 * of the reference (Deformation / DeformationGraph: Eigen + cholmod) and stays with the caller; the library provides what touched the GPU:
 * ifx_set_loop_closure_callback: cb runs inside ifx_process_frame / ifx_enqueue_frame_device right after the gates of a frame whose candidate
 *   was accepted (the frame waits for the verdict: one host synchronisation per frame while a callback is set).  From the callback:
 * ifx_sample_graph_model: Deformation::sampleGraphModel (EF/Deformation.cpp:224-337, sample.geom): x, y, z, init time of every 5000th surfel
 *   of the map order (the map state is the one the reference samples at the end of the previous frame); returns the number written.
 * ifx_loop_closure_constraints: the samples of :568-598 -- ACTIVE vertex render and INACTIVE time render on a (w/20) x (h/20) grid; per valid
 *   sample worldRawPoint = currPose * v (src3), worldModelPoint = estPose * v (dst3) and the surfel time (times); returns the count.
 * ifx_set_deformation: the `graph` argument of GlobalModel::clean (EF/GlobalModel.cpp:700-760): n_nodes x 16 floats (position 3, rotation 9
 *   column-major, translation 3, time), sorted by time, 4 <= n_nodes < 1024.  The NEXT clean (of this frame when called from the callback)
 *   applies it to every surviving surfel not created in that frame (copy_unstable.vert:178-374) and, for a local loop closure
 *   (is_fern = 0), refreshes the time stamp of moved stable surfels in front of the re-rendered INACTIVE depth (IndexMap::synthesizeDepth).
 * ifx_adopt_estimated_pose: currPose = estPose (:606). */
typedef int (*ifx_loop_closure_cb)(ifx_t* h, const float* lc24, void* user);
int ifx_set_loop_closure_callback(ifx_t* h, ifx_loop_closure_cb cb, void* user);
int ifx_sample_graph_model(ifx_t* h, float* out_xyzt, int max_n);
int ifx_loop_closure_constraints(ifx_t* h, float* src3, float* dst3, int32_t* times, int max_n);
int ifx_set_deformation(ifx_t* h, const float* graph16, int n_nodes, int is_fern);
int ifx_adopt_estimated_pose(ifx_t* h);

/* ---- the GPU contacts of the fern data base (EF/Ferns.cpp; codes, similarity search and keyframe store are host code in the reference and here:
 * instancefusion_amd/host/ifx_ferns.hpp is that class over the entry points below).
 * ifx_fern_frame: the four Resize passes of Ferns::addFrame / findFrame (:95-98, :192-195): fill-in image / vertex / normal and the instance render of the
 *   last predict(), resampled to (w/8) x (h/8) and read back (rgb: 3 bytes, maps: float4 per sample); returns the number of samples.
 * ifx_track_maps: the texture-initialised tracker as a stage -- initICPModel / initRGBModel(model maps, given in the frame of pose16) + initICP(vertices,
 *   normals) / initRGB(current maps) + getIncrementalTransformation starting at pose16 (EF/Ferns.cpp:558-592, EF/ElasticFusion.cpp:528-545).  Host
 *   float4 maps of the handle's resolution (RGBA8 images or NULL); uses the handle's configuration (icp_weight, pyramid, fast_odom; no SO(3)): for
 *   ferns create a handle with width/8, height/8, intrinsics/8, icp_weight 100, pyramid 0.  pose16 in: model pose = initial estimate; out: estimate.
 *   diag8: lastICPError, lastICPCount, lastRGBError, lastRGBCount, 0...
 * ifx_set_fern_callback: where the reference looks its fern data base up and deforms globally (EF/ElasticFusion.cpp:457-514): every frame after the
 *   first, with loop closure enabled, right after predict() at the tracked pose and before the local detection.  Inside the callback ifx_fern_frame
 *   delivers that predict() (what findFrame resamples), ifx_set_deformation(is_fern = 1) hands the optimised graph to this frame's clean and
 *   ifx_adopt_pose sets currPose = recoveryPose (:482, :504).  Return 1 when a graph was produced (rawGraph.size() > 0: the local detection of this
 *   frame is skipped, :516), 0 otherwise, < 0 to fail the frame.  A handle with this callback synchronises once per frame, as the reference does.
 *   Outside the callback ifx_fern_frame delivers the end-of-frame predict() (what Ferns::addFrame stores, :713-716). */
typedef int (*ifx_fern_cb)(ifx_t* h, void* user);
int ifx_set_fern_callback(ifx_t* h, ifx_fern_cb cb, void* user);
int ifx_adopt_pose(ifx_t* h, const float* pose16);
int ifx_fern_frame(ifx_t* h, uint8_t* img_rgb, float* verts4, float* norms4, uint8_t* inst_rgb);
/* the end-of-frame read-back (Ferns::addFrame) without a host stall: _async enqueues it behind the frame just processed, _fetch waits for it and hands
 * the four images over -- typically at the start of the next frame's fern callback, where the host has to wait for the stream anyway */
int ifx_fern_frame_async(ifx_t* h);
int ifx_fern_frame_fetch(ifx_t* h, uint8_t* img_rgb, float* verts4, float* norms4, uint8_t* inst_rgb);
int ifx_track_maps(ifx_t* h, const float* model_v4, const float* model_n4, const uint8_t* model_rgba, const float* cur_v4, const float* cur_n4,
                   const uint8_t* cur_rgba, float* pose16, float* diag8);

/* ---- map access (replaces getMapSurfelsGpu / getMapSurfelCount / id textures,
 * IF/map_interface/ElasticFusionInterface.h:55-120).  The store is struct-of-arrays; slots whose
 * surfel was deleted stay in place as tombstones until ifx_compact (DESIGN.md "Tombstones"). */
typedef struct ifx_soa_view {
    int32_t count;          /* slots in use (including tombstones) */
    int32_t capacity;
    float* d_pos_conf;      /* [capacity] float4: x,y,z, confidence          (vPosition)   */
    float* d_norm_rad;      /* [capacity] float4: nx,ny,nz, radius           (vNormRad)    */
    float* d_color;         /* [capacity] float2: packed rgb, packed inst.   (vColor.xy)   */
    float* d_times;         /* [capacity] float2: init time, last time       (vColor.zw)   */
    float* d_img_corr;      /* [capacity] float4                             (vImgCorr)    */
    float* d_votes;         /* [capacity][48]: vInstInfoA..L, one 192-byte record per slot */
} ifx_soa_view;
/* The pointers are MUTABLE, like the reference's getMapSurfelsGpu (its instance kernels write votes and colours through it), and a view is valid UNTIL THE NEXT FRAME
 * CALL on the handle: the frame path keeps a gathered copy of position / normal / times per slot ("hot records") that it rebuilds after every
 * ifx_map_view, so writes made between this call and the next frame are seen; writes through a pointer kept PAST a frame call are not (the list passes would read the
 * stale copy while the scans read the arrays) -- call ifx_map_view again before such writes.  ifx_set_option(h, "hot_verify", 1) makes every frame compare the copy with
 * the arrays first (one streaming pass, debug only); ifx_hot_records_stale returns how many slots it found differing (and repaired) since ifx_create: 0 for a caller
 * that keeps the rule. */
int ifx_map_view(ifx_t* h, ifx_soa_view* out);
int ifx_hot_records_stale(ifx_t* h);
int ifx_map_count(ifx_t* h);      /* live surfels (synchronises) */
int ifx_map_slots(ifx_t* h);      /* slots incl. tombstones (synchronises) */
/* Host copies of the live surfels in map order.  pc,nr,ic: float4 per surfel; col,tm: float2;
 * votes: 48 floats per surfel.  Any pointer may be NULL.  Returns the number of surfels written. */
int ifx_map_download(ifx_t* h, int max_n, float* pc, float* nr, float* col, float* tm, float* ic,
                     float* votes);
int ifx_map_upload(ifx_t* h, int n, const float* pc, const float* nr, const float* col,
                   const float* tm, const float* ic, const float* votes);
int ifx_set_pose(ifx_t* h, const float* pose16, int tick);
int ifx_compact(ifx_t* h);        /* order-preserving removal of tombstones */
/* Runtime options (name, value):
 *   "pyramid" / "fast_odom" / "so3" (0|1), "icp_weight_x1000" -- ElasticFusion::setPyramid / setFastOdom / setSo3 / setIcpWeight (EF/ElasticFusion.h:153-176), from the
 *                           next frame on; refused while a frame is announced ahead
 *   "reference_passes" 1  -- also run the BEFORE / INSTANCECOMPARE id renders of EF/ElasticFusion.cpp:679-680 (nobody on this path consumes them)
 *   "compact_every_frame" 1 -- remove tombstones after every clean (tests); "compact_divisor" d -- housekeeping compaction when tombstones > slots / d (default 8)
 *   "two_streams" 0       -- everything on one stream; "track_ahead" 0 -- do not enqueue the announced frame's tracker behind the current frame; "slic_ahead" 0 -- no superpixels ahead of a call
 *   "stage_timing" / "kernel_timing" 1 -- HIP-event records for ifx_stage_ms / ifx_kernel_ms (cost frame rate: off by default)
 *   "icp_blocks" n        -- cap on the blocks of a tracker reduction launch (0 = by image size)
 *   "raster_tiles" -1|0|1 -- tiled rasteriser (key tiles in LDS): by image size (on from 1 Mpixel) | off | on; results are identical either way
 *   "pace" 0              -- ifx_enqueue_frame_device normally waits for the PREVIOUS frame's result before it enqueues (the tracker announced ahead keeps the
 *                           device busy meanwhile); 0 = enqueue without looking back (a host that runs frames ahead measured 25 % slower)
 *   "gn_persist" mask     -- bit i: the Gauss-Newton iterations of pyramid level i in one persistent launch with grid barriers, while that level's grid has at
 *                           most "gn_persist_blocks" blocks (default 128).  Default 0 below 1280x960 (the two-launch form with the solve in the next launch's
 *                           prologue is within 0.7 % there and needs no co-resident grid), 4 = the coarsest level from 1280x960 on (+5 %).  A meeting of its
 *                           blocks that does not happen is re-run by one workgroup inside the frame (ifx_tracker_fallbacks counts them): slower, never wrong.
 *   "gn_prologue" 0       -- the 6x6 solve of an iteration by the last block of its second launch (round 3's form) instead of by every block of the next
 *                           iteration's first launch; "gn_prologue_blocks" n -- the prologue form only for launches of at most n blocks (default 2048)
 *   "fold_result" 0       -- the frame result by a launch of its own instead of the last block of the prediction's resolve
 *   "lazy_ids" 0          -- render the whole id image every frame (default: the lattice whetherDoSegmentation samples; the rest on demand)
 *   "fold_finish" 0, "seg_device" 0, "seg_aside" 0, "ff_union" 0, "ff_rounds" n -- the earlier forms of the end-of-frame sums and of the segmentation call's
 *                           schedule (host-driven / on the main stream / relaxation-only flood fill / length of the fixed relaxation schedule); identical results
 *   "clean_raster" 0, "hot_records" 0, "vlist_one" 1 -- round 5's map-pass forms off / on: the clean pass and the prediction's raster as ONE walk of the view list; the
 *                           gathered 64-byte copy of the hot fields; list offsets + concatenation in one launch (measured equal: off).  Identical results
 *   "host_entry_async" 1  -- ifx_process_frame returns when the frame's POSE is known (see there); "vote_per_mask" 0 -- the instance votes of a call in ONE launch over all
 *                           masks instead of one per mask in mask order (experiments only: the order is part of the reference's result while a packed counter's low half is negative)
 *   "overdue_rule" 0, "own_first_live" 0 -- test switches: a view-list rebuild without the age rule its newcomers have outlived / the sharded map's "surfel 0" fixed at
 *                           creation number 0 (round 4's behaviour of both: results then differ from the reference's in the cases tests/test_gpu_sweep.py and
 *                           test_owner_sharded_map_emulated hold) */
int ifx_set_option(ifx_t* h, const char* name, int value);

/* R32I surfel-id image after fusion (getSurfelIdsAfterFusionGpu, ElasticFusionInterface.h:90-102):
 * linear H*W int32 device buffer, 0 = empty. */
const int32_t* ifx_ids_after(ifx_t* h);
/* Host copy of an internal image (synchronises).  Names: "ids_after", "ids_tmp", "index",
 * "index_vc", "index_ct", "index_nr", "pred_vertex", "pred_normal", "pred_image", "pred_inst",
 * "pred_time", "fill_vertex", "fill_normal", "fill_image", "depth_filtered", "depth_metric",
 * "depth_metric_filtered", and with loop-closure detection on "old_vertex", "old_normal", "old_image", "old_time" (the
 * INACTIVE prediction, IndexMap::oldVertexTex() etc.) and "act_vertex", "act_normal", "act_image" (the predict() of
 * EF/ElasticFusion.cpp:453).  Returns bytes written or <0. */
int ifx_image_download(ifx_t* h, const char* name, void* out, int64_t max_bytes);

/* ---- map stage API (unit-parity surface; each replaces one GL pass of the reference) */
int ifx_predict_indices(ifx_t* h, const float* pose16, int time);                /* EF/IndexMap.cpp:221-279 */
int ifx_combined_predict(ifx_t* h, const float* pose16, int time, int max_time); /* EF/IndexMap.cpp:468-574 */
int ifx_fuse(ifx_t* h, const float* pose16, int time, float weighting);          /* EF/GlobalModel.cpp:459-698 */
int ifx_clean(ifx_t* h, const float* pose16, int time);                          /* EF/GlobalModel.cpp:700-925 */
int ifx_render_ids(ifx_t* h, const float* pose16, int mode);                     /* EF/IndexMap.cpp:315-465; result in "ids_tmp" */
int ifx_set_frame(ifx_t* h, const uint8_t* rgb, const uint16_t* depth);          /* upload + preprocess only */

/* ---- tracker stage API (replaces the blocking host launchers of EF/Cuda/cudafuncs.cuh:64-183).
 * All pointers are device pointers; planar maps are [3][h][w]; results go to device memory. */
int ifx_icp_step(ifx_t* h, const float* Rcurr9, const float* tcurr3, const float* d_vmap_curr,
                 const float* d_nmap_curr, const float* Rprev_inv9, const float* tprev3, float fx,
                 float fy, float cx, float cy, const float* d_vmap_g_prev, const float* d_nmap_g_prev,
                 float dist_thres, float angle_thres, int w, int hgt, float* out29_host);
int ifx_rgb_residual(ifx_t* h, float min_scale, const int16_t* d_didx, const int16_t* d_didy,
                     const float* d_last_depth, const float* d_next_depth, const uint8_t* d_last_img,
                     const uint8_t* d_next_img, void* d_corres8, float max_depth_delta,
                     const float* kt3, const float* krkinv9, int w, int hgt, int* count_host,
                     int* sigma_host);
int ifx_rgb_step(ifx_t* h, const void* d_corres8, float sigma, const float* d_cloud3, float fx,
                 float fy, const int16_t* d_didx, const int16_t* d_didy, float sobel_scale, int w,
                 int hgt, float* out29_host);
int ifx_so3_step(ifx_t* h, const uint8_t* d_last_img, const uint8_t* d_next_img,
                 const float* image_basis9, const float* kinv9, const float* krlr9, int w, int hgt,
                 float* out11_host);
/* The pyramid builders as ONE stage call: createVMap / createNMap / pyrDown / pyrDownGaussF / pyrDownUcharGauss / resizeVMap / resizeNMap / tranformMaps /
 * verticesToDepth / imageBGRToIntensity / computeDerivativeImages / projectToPointCloud (EF/Cuda/cudafuncs.cuh:64-183), chained as RGBDOdometry::initICP + initRGB
 * (the frame side: EF/Utils/RGBDOdometry.cpp:118-142, 243-247, 287-293) and initICPModel + initRGBModel (the model side: :169-206, 237-241) chain them.
 * In (device): the bilateral-filtered depth (u16 millimetres, DEPTH_FILTERED) and the RGB8 frame -- both or neither; the model prediction as float4 vertex / normal
 * maps in the camera frame of `model_pose16` (host, row-major camera-to-world) and its RGBA8 image -- all three or none.
 * Out (device, caller-allocated, DENSE: pitch = the level's width; the reference's DeviceArray2D are pitched, its kernels address them by row all the same): level l
 * is (width >> l) x (height >> l); planar maps are [3][h_l][w_l].  A NULL entry is skipped.  The handle's own pyramids are overwritten (it is a stage call, like
 * ifx_track_pair); the depth cut-off of createVMap is the handle's max_depth_processed (EF/ElasticFusion.cpp:73), that of verticesToDepth 6 m as in the reference.
 * Error behaviour: IFX_E_INVALID for a half-given input group, IFX_E_STATE for a point cloud at a level the handle's configuration never iterates on. */
typedef struct ifx_pyramids {
    /* frame side */
    uint16_t* depth[3];        /* depth_tmp: level 0 = the input, then pyrDown (5x5 Gaussian with the 3-sigma depth gate)          [h_l][w_l]    */
    float* vmap_curr[3];       /* createVMap                                                                                       [3][h_l][w_l] */
    float* nmap_curr[3];       /* createNMap                                                                                       [3][h_l][w_l] */
    uint8_t* next_img[3];      /* imageBGRToIntensity, pyrDownUcharGauss                                                           [h_l][w_l]    */
    int16_t* didx[3];          /* computeDerivativeImages (Sobel x)                                                                [h_l][w_l]    */
    int16_t* didy[3];          /* computeDerivativeImages (Sobel y)                                                                [h_l][w_l]    */
    /* model side */
    float* vmap_g_prev[3];     /* copyMaps, resizeVMap, tranformMaps: vertices in the GLOBAL frame                                 [3][h_l][w_l] */
    float* nmap_g_prev[3];     /* copyMaps, resizeNMap, tranformMaps                                                               [3][h_l][w_l] */
    float* last_depth[3];      /* verticesToDepth, pyrDownGaussF                                                                   [h_l][w_l]    */
    uint8_t* last_img[3];      /* intensity of the predicted image, pyrDownUcharGauss                                              [h_l][w_l]    */
    float* cloud[3];           /* projectToPointCloud                                                                              [h_l][w_l][3] */
} ifx_pyramids;
int ifx_build_pyramids(ifx_t* h, const uint16_t* d_depth_filtered, const uint8_t* d_rgb, const float* d_model_v4, const float* d_model_n4, const uint8_t* d_model_rgba,
                       const float* model_pose16, ifx_pyramids* out);
/* Whole tracker on explicit inputs (RGBDOdometry::initICPModel/initRGBModel/initICP/initRGB +
 * getIncrementalTransformation, EF/Utils/RGBDOdometry.cpp:118-603).  Host inputs. */
int ifx_track_pair(ifx_t* h, const float* model_v4, const float* model_n4, const uint8_t* model_rgba,
                   const uint8_t* prev_rgb, const uint16_t* depth_filtered, const uint8_t* rgb,
                   float* pose16_inout, float* diag8);
/* Host copy of a tracker pyramid buffer; names as in the reference's members
 * (EF/Utils/RGBDOdometry.h:80-121): "vmap_curr","nmap_curr","vmap_prev","nmap_prev","last_depth",
 * "last_img","next_img","lastnext_img","didx","didy","cloud","corres","depth_tmp". */
int ifx_tracker_buffer_download(ifx_t* h, const char* name, int level, void* out, int64_t max_bytes);

/* ---- instance layer (replaces InstanceFusion::whetherDoSegmentation / ProcessSegmentation /
 * processInstance, IF/Core/InstanceFusion.h:72-107, IF/Core/InstanceFusion.cpp:192-270,655-1067) */
int ifx_should_segment(ifx_t* h, int frame);
/* masks: n x H x W u8 (0/255), sorted by area descending (the contract of the Mask-RCNN bridge,
 * build/mask_ori.py:117); class_ids: n COCO indices.  flags bit0: kNN smoothing of the instance colours
 * (isflann: flannKnnVoteSurfelMap, IF/Core/InstanceFusion.cpp:1070-1163), bit1: superpixel refinement (needs rgb and depth of the frame: host pointers, as
 * InstanceFusion::ProcessSegmentation takes them -- or BOTH NULL = the frame most recently processed, whose raw images are still resident in their frame slot:
 * no staging copy, no upload). */
int ifx_process_segmentation(ifx_t* h, const uint8_t* rgb, const uint16_t* depth,
                             const uint8_t* masks, const int32_t* class_ids, int n, int frame,
                             int flags);
/* bestIDInEachSurfel (IF/Core/InstanceFusionCuda.cu:1158-1200) for the live surfels, map order. */
int ifx_labels(ifx_t* h, int32_t* out, int max_n);
/* InstanceFusion::renderProjectMap (IF/Core/InstanceFusion.cpp:1232-1252, renderProjectFrameKernel IF/Core/InstanceFusionCuda.cu:1432-1498): the
 * instance colour of the surfel under every pixel of the id image after fusion -- what getProjectColorMap_gpu() hands to the GUI: H x W x 4
 * floats (r, g, b in [0,1], alpha 1; black where no stable surfel is visible).  out_rgba: host buffer or NULL; d_out_rgba: device buffer or NULL
 * (enqueue only, no synchronisation).  The 2-D boxes the reference draws on top on the host are not drawn. */
int ifx_render_project_map(ifx_t* h, float* out_rgba, float* d_out_rgba);
/* ---- instance ground truth (the reference's ScanNet evaluation).
 * ifx_set_instance_gt: the instanceGT argument of ElasticFusion::processFrame (EF/ElasticFusion.h:79, EF/ElasticFusion.cpp:285-291): H x W bytes; surfels created by the
 *   frames that follow remember the id under the pixel that created them (vImgCorr.w, data.vert:215-228; -2 without ground truth).  NULL switches it off.
 * ifx_precision_recall: computePrecisionAndRecallKernel (IF/Core/InstanceFusionCuda.cu:2085-2114), the device half of InstanceFusion::evaluateAndSave
 *   (IF/Core/InstanceTable.cpp:336-370): surfels per instance (by instance colour), per ground-truth id, and per (ground-truth id, instance) pair. */
int ifx_set_instance_gt(ifx_t* h, const uint8_t* gt_hw);
int ifx_precision_recall(ifx_t* h, int32_t* inst_num96, int32_t* gt_num256, int32_t* inst_gt_map_256x96);
/* class id per instance slot, -1 = unused (getInstanceTable, IF/Core/InstanceFusion.h:87) */
int ifx_instance_table(ifx_t* h, int32_t* out96);
/* getLoopClosureInstanceTable, IF/Core/InstanceTable.cpp:98-121: int[96*5] = r,g,b,class,index */
int ifx_loop_closure_instance_table(ifx_t* h, int32_t* out480);
int ifx_mask_clean_overlap(ifx_t* h, uint8_t* masks, int n);  /* IF/Core/InstanceFusionCuda.cu:118-141 (host in/out) */
/* maskGeometricFilter + filterAreaCompute (IF/Core/InstanceFusion.cpp:470-593) as a stage, host buffers: depth = model
 * depth under the camera (u16 H x W, 1186 units per metre, what getProjectDepthMap :977-996 produces); masks n x H x W
 * in/out; ori = the masks before clean-overlap; unavailable n bytes in/out (set entries are skipped). */
int ifx_mask_geometric_filter(ifx_t* h, const uint16_t* depth, uint8_t* masks, const uint8_t* ori, int n, uint8_t* unavailable);

/* ---- superpixel refinement stages (the three calls of processInstance steps -1_1..-1_3,
 * IF/Core/InstanceFusion.cpp:722-738; ifx_process_segmentation runs them when flags bit1 is set).
 * All buffers are host memory, images are H x W row-major.
 * ifx_slic_segment: gSLICrInterface (IF/Core/InstanceFusion_superpixel.cpp:713-772; gSLICr with
 *   spixel_size 16, coh_weight 0.6, 5 iterations, XYZ, enforce connectivity): rgb H x W x 3 u8 ->
 *   superpixel label per pixel; returns the number of superpixels (> 0) or a negative error.
 * ifx_merge_superpixels: mergeSuperPixel (IF/Core/InstanceFusion_superpixel.cpp:40-225): seg is the
 *   SLIC labelling on entry and the re-clustered labelling (-1 = no geometry) on return, final_out the
 *   merged region id per pixel, info_out (optional) the spNum x 30 float table in the reference's SPI_*
 *   layout (:12-33).  Returns spNum.
 * ifx_mask_superpixel_filter: maskSuperPixelFilter_OverSeg (:651-710): a mask keeps a merged region
 *   iff it covers more than 75 % of it; masks n x H x W u8 rewritten in place. */
/* flannKnnVoteSurfelMap + mapKnnVoteColourKernel (IF/Core/InstanceFusion.cpp:1070-1163, IF/Core/InstanceFusionCuda.cu:1237-1340)
 * as a stage: every live surfel takes the colour of the instance most of its 10 nearest surfels (itself included) are
 * labelled with, using the labels of the last segmentation call.  nbr_out (optional, [max_n][10] int32, host) receives
 * the neighbour slots of the first max_n slots, -1 = none. */
int ifx_knn_vote_colour(ifx_t* h, int32_t* nbr_out, int max_n);
int ifx_slic_segment(ifx_t* h, const uint8_t* rgb, int32_t* seg_out);
int ifx_merge_superpixels(ifx_t* h, const uint16_t* depth, int32_t* seg_inout, int32_t* final_out, float* info_out);
int ifx_mask_superpixel_filter(ifx_t* h, const int32_t* final_ids, uint8_t* masks, int n);

/* ---- measurement hooks */
/* Per-stage GPU time of the frames processed since the last reset, from HIP events on the handle's
 * streams: ms[0]=track, ms[1]=fuse (all map passes), ms[2]=instance, ms[3]=frame side (bilateral, frame pyramids,
 * SO(3); runs concurrently with the others when a frame is announced ahead).  Stages 0, 1, 3 are recorded only while
 * ifx_set_option(h, "stage_timing", 1) is on: every event record is a marker packet on the queue and the eight of a
 * frame cost about 4 % of the frame rate. */
int ifx_stage_ms(ifx_t* h, float* ms4, int reset);
/* Superpixels run ahead of a segmentation call.  When ifx_should_segment answers "not this frame" but its cadence says the NEXT frame will end with a call, and
 * that frame is announced (ifx_hint_next_frame_device), SLIC + the superpixel merge of the announced frame -- work that reads the frame only, about half of a call's
 * dispatches (gSLICr + IF/Core/InstanceFusion.cpp:722-738 steps -1_1, -1_2) -- are enqueued on the side stream at once, under that frame's tracker and map passes;
 * ifx_process_segmentation(rgb = depth = NULL) of that frame then waits for one event instead of running them.  Same kernels, same images, same result.
 * ms: device time of the runs on the side stream (NOT part of ifx_stage_ms's "instance", which is the main-stream span of the calls); runs / used: how many were
 * enqueued and how many a call consumed (a hint that does not come true leaves its run unused).  Option "slic_ahead" 0 turns the look-ahead off. */
int ifx_superpixel_ahead_stats(ifx_t* h, float* ms, int32_t* runs, int32_t* used, int reset);
/* The one-frame look-ahead in numbers (diagnostics, tests; no counterpart in the reference): out3 = frames of this handle (unsharded entries) that found their frame
 * side already computed (ifx_prefetch_ / ifx_hint_next_frame*), frames whose tracker had run ahead, frames that came through ifx_hint_next_frame (host pointers). */
int ifx_lookahead_stats(ifx_t* h, int32_t* out3, int reset);
/* Average duration (ms) of the named kernel over its launches since the last reset, measured with
 * HIP events around each launch (enabled by ifx_set_option("kernel_timing",1)). */
int ifx_kernel_ms(ifx_t* h, const char* kernel, float* avg_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* IFX_C_API_H_ */
