/*
 * oracle/orc.h -- CPU restatement ("oracle") of InstanceFusion's per-frame dense surfel pipeline.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (instancefusion_amd/csrc, libifx.so) never
 * links, loads or calls anything in this directory.
 *
 * PARITY STATUS, by row of SURVEY.md section 8(a):
 *   PINNED to values the reference's own sources produced here:
 *     a2, a9-a14 (the GLSL map passes: bilateral / metric depth, index map, association, fusion update, clean, surfel ids, splat prediction + fill-in) -- the reference's
 *       UNMODIFIED shader files run on Mesa's software rasteriser through a window-less GL 4.5 context (oracle/gl/, tools/make_golden_gl.py ->
 *       tests/golden/gl_map_passes.npz, tests/test_gl_golden.py).  Agreement of this oracle with them on identical inputs: index map, association, fusion, clean 100 %
 *       of the elements (values to f32 rounding); bilateral 99.96 % of the pixels (1 mm: exp); splat prediction 99.98 %; surfel ids 99.0 % (a documented
 *       coverage rule).  The GL calls around the shaders restate the reference's host code (Pangolin / GLEW / CUDA interop do not build here);
 *       every formula that decides a result is the reference's own text, executed.
 *     f-3, GPU part (the deformation graph applied by the clean pass + synthesizeDepth): the same generator; survivors, positions (1.4e-6 m), normals and re-activated
 *       last-seen times equal the shader's.
 *     a20 (gSLICr: the reference's shared per-pixel maths, -DCOMPILE_WITHOUT_CUDA, oracle/_ref/libref_slic.so, tests/test_oracle_slic.py)
 *     f-2 (k-NN: the reference's vendored FLANN 1.8.4, oracle/ref_knn.cpp, tests/golden/knn_ref.npz)
 *   "PARITY UNPINNED" -- restated by hand from the sources, each function citing the file:line it follows, cross-checked by a second restatement in numpy
 *   (tests/test_restatement_*.py) but by no value the reference produced: the CUDA stages a3-a8 (pyramids, ICP / RGB / SO(3) reductions, Gauss-Newton step) and
 *   the instance layer a16-a19, a21-a23.  The reference has no CPU path, no tests and no golden vectors for them, and its CUDA sources do
 *   not build in this image without stand-ins for the CUDA toolkit.
 *
 * Citation prefixes:  EF/ = elasticfusionpublic/Core/src/   IF/ = src/   (under /root/reference)
 *
 * Conventions: images are row-major [y*w+x]; planar maps are [3][h][w] (x-plane, y-plane, z-plane)
 * as EF/Cuda/cudafuncs.cu:109-149; 4-channel maps are interleaved float4 per pixel.
 */
#ifndef ORC_H_
#define ORC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_PYRS 3
#define ORC_NUM_INST 96          /* IF/main.cpp:31-44 instanceNum */
#define ORC_VOTE_FLOATS 48       /* 96 int16 counters, two per float (IF/Core/InstanceFusionCuda.cu:22-39) */

/* Same field order as ifx_config in include/ifx_c_api.h (tests build both from one dict). */
typedef struct orc_config {
    int32_t width, height;
    float fx, fy, cx, cy;
    int32_t time_delta;          /* 200   IF/map_interface/ElasticFusionInterface.cpp:43-45 */
    float confidence;            /* 10 */
    float depth_cut;             /* 12 m  (preprocess gate) */
    float max_depth_processed;   /* 20 m  EF/ElasticFusion.cpp:73 */
    float icp_weight;            /* 10 */
    int32_t pyramid;             /* 1 */
    int32_t fast_odom;           /* 0 */
    int32_t so3;                 /* 1 */
    int32_t max_surfels;         /* capacity */
    int32_t device;              /* unused by the oracle */
    int32_t n_ranks, rank;       /* unused by the oracle */
} orc_config;

/* DataTerm, EF/Cuda/types.cuh:75-81 (reference layout, 16 B) */
typedef struct orc_dataterm {
    int16_t zero_x, zero_y;      /* pixel in the last (model) image */
    int16_t one_x, one_y;        /* pixel in the next (current) image */
    float diff;
    int32_t valid;
} orc_dataterm;

/* ---------------------------------------------------------------- preprocessing (a2) */
void orc_bilateral(const uint16_t* in, uint16_t* out, int w, int h, float maxD);
void orc_metric(const uint16_t* in, float* out, int w, int h, float maxD);

/* ---------------------------------------------------------------- pyramid kernels (a3) */
void orc_pyrdown_u16(const uint16_t* src, int sw, int sh, uint16_t* dst);
void orc_vmap(const uint16_t* depth, int w, int h, float fx, float fy, float cx, float cy,
              float cutoff, float* vmap);
void orc_nmap(const float* vmap, int w, int h, float* nmap);
void orc_copy_maps(const float* v4, const float* n4, int w, int h, float* vmap, float* nmap);
void orc_resize_map(const float* in, int sw, int sh, float* out, int normalize);
void orc_transform_maps(float* vmap, float* nmap, int w, int h, const float* R, const float* t);
void orc_vertices_to_depth(const float* v4, int w, int h, float cutoff, float* d);
void orc_pyrdown_gauss_f(const float* src, int sw, int sh, float* dst);
void orc_pyrdown_gauss_u8(const uint8_t* src, int sw, int sh, uint8_t* dst);
void orc_rgb_to_intensity(const uint8_t* rgb, int stride, int n, uint8_t* dst);
void orc_sobel(const uint8_t* img, int w, int h, int16_t* dx, int16_t* dy);
void orc_project_cloud(const float* depth, int w, int h, float fx, float fy, float cx, float cy,
                       float* cloud3);

/* ---------------------------------------------------------------- tracker reductions (a4-a7) */
void orc_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr,
                  const float* nmap_curr, const float* Rprev_inv, const float* tprev, float fx,
                  float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                  float dist_thres, float angle_thres, int w, int h, float* out29);
void orc_rgb_residual(float min_scale, const int16_t* didx, const int16_t* didy,
                      const float* last_depth, const float* next_depth, const uint8_t* last_img,
                      const uint8_t* next_img, orc_dataterm* corres, float max_depth_delta,
                      const float* kt, const float* krkinv, int w, int h, int* count, int* sigma);
void orc_rgb_step(const orc_dataterm* corres, float sigma, const float* cloud3, float fx, float fy,
                  const int16_t* didx, const int16_t* didy, float sobel_scale, int w, int h,
                  float* out29);
void orc_so3_step(const uint8_t* last_img, const uint8_t* next_img, const float* image_basis,
                  const float* kinv, const float* krlr, int w, int h, float* out11);

/* ---------------------------------------------------------------- whole tracker (a3-a8) */
typedef struct orc_tracker orc_tracker;
orc_tracker* orc_tracker_create(int w, int h, float fx, float fy, float cx, float cy);
void orc_tracker_destroy(orc_tracker*);
void orc_tracker_init_first_rgb(orc_tracker*, const uint8_t* rgb);
/* model_v4/model_n4: float4 maps (camera frame of pose), model_rgba: RGBA8 image */
void orc_tracker_init_model(orc_tracker*, const float* model_v4, const float* model_n4,
                            const uint8_t* model_rgba, const float* pose16);
void orc_tracker_init_frame(orc_tracker*, const uint16_t* depth_filtered, const uint8_t* rgb,
                            float depth_cutoff);
/* initICP(predictedVertices, predictedNormals) + initRGB (EF/Utils/RGBDOdometry.cpp:144-167, 243-247): the "current frame" of
 * the model-to-model tracker is a render of the map (float4 maps in the camera frame, RGBA8 image) */
void orc_tracker_init_frame_maps(orc_tracker*, const float* v4, const float* n4, const uint8_t* rgba);
/* getCovariance, EF/Utils/RGBDOdometry.cpp:605-608: inverse of the last normal matrix (row-major 6x6) */
void orc_tracker_covariance(orc_tracker*, double* cov36);
/* pose16 in/out (row-major 4x4, camera-to-world); diag[8]: icpErr,icpCount,rgbErr,rgbCount,so3Err,so3Count,0,0 */
void orc_tracker_run(orc_tracker*, float* pose16, float icp_weight, int pyramid, int fast_odom,
                     int so3, float* diag);
/* access to internals for stage-level parity tests; returns pointer owned by the tracker */
const void* orc_tracker_buffer(orc_tracker*, const char* name, int level);

/* ---------------------------------------------------------------- map (a9-a15) */
typedef struct orc_map orc_map;   /* AoS-free: separate arrays, reference order/compaction semantics */

/* ---------------------------------------------------------------- full pipeline object */
typedef struct orc orc_t;
orc_t* orc_create(const orc_config*);
void orc_destroy(orc_t*);
int orc_process_frame(orc_t*, const uint8_t* rgb, const uint16_t* depth, int64_t ts,
                      const float* in_pose16, float weight_mult, float* out_pose16);
int orc_map_count(orc_t*);
void orc_get_pose(orc_t*, float* out16);
void orc_set_instance_gt(orc_t*, const uint8_t* gt_hw);   /* instanceGT of processFrame: new surfels remember the id under their pixel (vImgCorr.w) */
int orc_tick(orc_t*);
/* f-4: computeMapBoundingBox / getInstancePointCloud (IF/Core/InstanceFusion.cpp:1261-1590) */
void orc_map_bounding_boxes(orc_t*, int bbox_type, float ratio, float* boxes96x6, float* ground_normal3, float* gc16, float* inst16, int32_t* ground_votes648);
int orc_instance_point_cloud(orc_t*, int bbox_type, int32_t* counts96, int inst, float* out10, int max_records);
void orc_tracker_diag(orc_t*, float* out8);
void orc_set_bootstrap(orc_t*, int on);
/* wall-clock per stage since the last reset (ms): track incl. preprocessing | map passes | instance layer */
void orc_stage_ms(orc_t*, double* out3, int reset);
/* copy out map fields; any pointer may be NULL.  pc,nr,ic: float4 per surfel; col,tm: float2; votes: 48 floats/surfel */
void orc_map_download(orc_t*, float* pc, float* nr, float* col, float* tm, float* ic, float* votes);
void orc_map_upload(orc_t*, int n, const float* pc, const float* nr, const float* col,
                    const float* tm, const float* ic, const float* votes);
void orc_set_pose(orc_t*, const float* pose16, int tick);
/* images: name in {"ids_after","index","pred_vertex","pred_normal","pred_image","pred_time",
 * "fill_vertex","fill_normal","fill_image","depth_filtered","depth_metric","depth_metric_filtered",
 * "old_vertex","old_normal","old_image","old_time"} */
const void* orc_image(orc_t*, const char* name);

/* ---- local loop-closure detection (EF/ElasticFusion.cpp:453-566 without ferns): predict() at the tracked pose, INACTIVE
 * prediction (surfels not seen for time_delta frames), model-to-model tracking of the active render against the inactive one,
 * covariance / count / error gates.  Constructor arguments countThresh, errThresh, covThresh (EF/ElasticFusion.h:48-50).
 * The deformation that follows an accepted candidate is NOT part of the oracle. */
void orc_set_loop_closure(orc_t*, int enable, int count_thresh, float err_thresh, float cov_thresh);
/* out[24]: 0 model-to-model ran (0: no inactive pixel in view), 1 valid pixels of the inactive render, 2 lastICPError,
 * 3 lastICPCount, 4 covOk, 5 accepted, 6..21 estPose (row-major), 22 largest diagonal covariance entry, 23 candidates so far */
void orc_loop_closure_diag(orc_t*, float* out24);
/* ---- device-side hooks of the deformation that follows an accepted candidate (orc_deform.c).  The callback runs inside
 * orc_process_frame right after the gates when a candidate was accepted (the place of EF/ElasticFusion.cpp:566-613); it may call
 * the four functions below.  The graph optimisation itself (EF/Utils/DeformationGraph.cpp) is the caller's. */
typedef int (*orc_lc_callback)(orc_t*, const float* lc24, void* user);
void orc_set_loop_closure_callback(orc_t*, orc_lc_callback cb, void* user);
/* the place of Ferns::findFrame and the global deformation (EF/ElasticFusion.cpp:457-514): every frame after predict() at the tracked pose; > 0 = a graph
 * was produced, the local detection is skipped (:516).  orc_fern_frame: the four Resize passes of EF/Ferns.cpp:95-98 / :192-195 on the last predict()
 * ((w/8) x (h/8) nearest samples of the fill-in image / vertex / normal and of the instance render).  orc_adopt_pose: currPose = recoveryPose (:482, :504). */
typedef int (*orc_fern_callback)(orc_t*, void* user);
void orc_set_fern_callback(orc_t*, orc_fern_callback cb, void* user);
int orc_fern_frame(orc_t*, uint8_t* img_rgb, float* verts4, float* norms4, uint8_t* inst_rgb);
void orc_adopt_pose(orc_t*, const float* pose16);
int orc_sample_graph_model(orc_t*, float* out_xyzt, int max_n);                                   /* Deformation::sampleGraphModel */
int orc_loop_closure_constraints(orc_t*, float* src3, float* dst3, int32_t* times, int max_n);   /* EF/ElasticFusion.cpp:568-598 */
void orc_set_deformation(orc_t*, const float* graph16, int n_nodes, int is_fern);                 /* `graph` of GlobalModel::clean, applied by the next clean */
void orc_adopt_estimated_pose(orc_t*);                                                            /* currPose = estPose, :606 */

/* stage-level map entry points operating on the object's map with an explicit pose/time */
void orc_predict_indices(orc_t*, const float* pose16, int time);
void orc_combined_predict(orc_t*, const float* pose16, int time, int max_time);
void orc_stage_predict(orc_t*, const float* pose16, int time, int max_time);   /* + fill-in, as ifx_combined_predict */
void orc_fuse(orc_t*, const float* pose16, int time, float weighting);
void orc_clean(orc_t*, const float* pose16, int time);
void orc_render_ids(orc_t*, const float* pose16, int mode /*0 general, 1 instance-compare*/);

/* ---------------------------------------------------------------- instance path (a16-a23) */
int orc_should_segment(orc_t*, int frame);
/* masks: n x H x W uint8 (0/255, area-descending); returns 0 ok */
int orc_process_segmentation(orc_t*, const uint8_t* rgb, const uint16_t* depth,
                             const uint8_t* masks, const int32_t* class_ids, int n, int frame,
                             int do_knn);
void orc_labels(orc_t*, int32_t* out);
void orc_precision_recall(orc_t*, int32_t* inst_num96, int32_t* gt_num256, int32_t* inst_gt_map_256x96);   /* computePrecisionAndRecallKernel */
void orc_render_project_map(orc_t*, float* out_rgba);   /* renderProjectFrameKernel: instance colour under every pixel, H x W x 4 floats */
void orc_instance_table(orc_t*, int32_t* class_of_instance /*96, -1 unused*/);

/* superpixel refinement (a20, a21; orc_slic.c).  orc_slic_segment: gSLICrInterface, returns the number of
 * superpixels; orc_merge_superpixels: mergeSuperPixel (seg in/out, final ids out, optional spn*30 info
 * table in the reference's SPI_* layout); orc_mask_superpixel_filter: maskSuperPixelFilter_OverSeg. */
int orc_slic_segment(orc_t*, const uint8_t* rgb, int32_t* seg);
int orc_merge_superpixels(orc_t*, const uint16_t* depth, int32_t* seg, int32_t* final_out, float* info_out);
void orc_mask_superpixel_filter(orc_t*, const int32_t* final_ids, uint8_t* masks, int n);

/* kNN smoothing of the instance colours (flannKnnVoteSurfelMap); nbr optional n x 10 */
void orc_knn_vote(orc_t*, int32_t* nbr);

/* stage-level instance helpers */
void orc_mask_clean_overlap(uint8_t* masks, int n, int w, int h);
float orc_vote_encode(int a, int b);
void orc_vote_decode(float f, int* a, int* b);

#ifdef __cplusplus
}
#endif
#endif
