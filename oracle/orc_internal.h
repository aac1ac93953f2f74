/* oracle/orc_internal.h -- private state of the CPU oracle (test infrastructure only, see orc.h). */
#ifndef ORC_INTERNAL_H_
#define ORC_INTERNAL_H_

#include "orc.h"

/* one measurement produced by the association pass (data.vert) */
typedef struct orc_meas {
    float pc[4];   /* world position, confidence a */
    float col0;    /* packed rgb */
    float nr[4];   /* world normal, radius */
    float ic[4];   /* imgCorr */
    int kind;      /* 0 nothing, 1 update of `target`, 2 new unstable surfel */
    uint32_t target;
} orc_meas;

struct orc {
    orc_config cfg;
    int w, h, P;
    int tick;
    float pose[16];
    /* map: reference order, compacted every frame like the transform-feedback passes */
    int n, cap;
    float *pc, *nr, *col, *tm, *ic, *votes;
    /* frame inputs / preprocess */
    uint8_t* rgb;
    uint16_t *depth_raw, *depth_filt;
    float *dm, *dmf;
    /* index map */
    uint32_t* index_id;
    float *index_z, *index_vc, *index_ct, *index_nr;
    /* predictions */
    float *pred_vertex, *pred_normal, *zbuf;
    uint8_t *pred_image, *pred_inst;
    uint16_t* pred_time;
    float *fill_vertex, *fill_normal;
    uint8_t* fill_image;
    int32_t *ids_after, *ids_tmp;
    /* fuse scratch */
    orc_meas *newbuf, *updbuf;
    int n_new, n_upd;
    orc_tracker* trk;
    float last_weighting;
    float diag[8];
    /* instance layer */
    int32_t inst_class[ORC_NUM_INST]; /* -1 = slot unused */
    float inst_color[ORC_NUM_INST];
    int32_t* labels;                  /* bestIDInEachSurfel, length cap */
    int last_seg_frame;
    int clean_times;
    /* local loop-closure detection (EF/ElasticFusion.cpp:453-566) */
    int lc_enable, lc_count_thresh, lc_candidates;
    float lc_err_thresh, lc_cov_thresh;
    float *old_vertex, *old_normal;
    uint8_t *old_image, *old_inst;
    uint16_t* old_time;
    orc_tracker* m2m;
    float lc[24];
    /* deformation graph handed in for the next clean (GlobalModel::clean's `graph` argument), orc_deform.c */
    float* graph;
    int graph_nodes, graph_is_fern;
    uint8_t* inst_gt;   /* instance ground truth of the frames to come (NULL: none) */
    orc_lc_callback lc_cb;
    void* lc_user;
    int bootstrap_next;          /* processFrame's `bootstrap` for the next frame (EF/ElasticFusion.cpp:334-356) */
    double stage_ms[3];          /* wall-clock of the stages (track incl. preprocessing | map passes | instance layer): the bench's CPU baseline reads them */
    orc_fern_callback fern_cb;   /* Ferns::findFrame + global deformation of the caller (EF/ElasticFusion.cpp:457-514) */
    void* fern_user;
};

void orc_deform_surfel(const float* g, int nodes, float* pc, float* nr, float initT, float* lastT, int time, float thr, int is_fern, const float* tinv,
                       const float* depth, int w, int h, float cx, float cy, float fx, float fy, float maxDepth);

float orc_encode_color(float r, float g, float b);
void orc_decode_color(float c, float* out3);
void orc_pose_inverse(const float* p, float* o);
void orc_instance_init(orc_t* o);
void orc_instance_free(orc_t* o);
double orc_now_ms(void);

#endif
