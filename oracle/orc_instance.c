/*
 * oracle/orc_instance.c -- CPU restatement of the instance layer of the path (SURVEY.md 8a rows
 * a16-a19, a22, a23).  TEST INFRASTRUCTURE ONLY (see orc.h).
 */
#include "orc.h"
#include "orc_math.h"
#include "orc_internal.h"
#include <stdlib.h>
#include <stdio.h>

#define NI ORC_NUM_INST

/* encode_Instance / decode{1,2}_Instance, IF/Core/InstanceFusionCuda.cu:22-39.  The arguments are
 * `short` in the reference, the int -> float conversion rounds to nearest even, and float -> int
 * truncates with CUDA's saturation. */
float orc_vote_encode(int a, int b)
{
    int16_t sa = (int16_t)a, sb = (int16_t)b;
    int32_t info = (int32_t)((uint32_t)(int32_t)sa << 16) + (int32_t)sb;
    return (float)info;
}
void orc_vote_decode(float f, int* a, int* b)
{
    int v = orc_f2i_rz(f);
    *a = (int16_t)((v >> 16) & 0xFFFF);
    *b = (int16_t)(v & 0xFFFF);
}

void orc_instance_init(orc_t* o)
{
    uint32_t s = 0x1F5u;
    for (int i = 0; i < NI; i++) {
        o->inst_class[i] = -1;
        /* createInstanceTable (IF/Core/InstanceTable.cpp:11-35) draws colours from unseeded rand();
         * a fixed LCG is used instead -- colours are arbitrary in the reference too. */
        int c[3];
        for (int k = 0; k < 3; k++) { s = s * 1664525u + 1013904223u; c[k] = (int)((s >> 8) % 255u); }
        o->inst_color[i] = (float)((c[0] << 16) + (c[1] << 8) + c[2]);
    }
    o->labels = (int32_t*)calloc((size_t)o->cap, 4);
    o->last_seg_frame = -1;
    o->clean_times = 0;
}
void orc_instance_free(orc_t* o) { free(o->labels); }

void orc_instance_table(orc_t* o, int32_t* out) { memcpy(out, o->inst_class, sizeof(o->inst_class)); }
void orc_labels(orc_t* o, int32_t* out) { memcpy(out, o->labels, (size_t)o->n * 4); }

/* computePrecisionAndRecallKernel, IF/Core/InstanceFusionCuda.cu:2085-2114 (InstanceFusion::evaluateAndSave, IF/Core/InstanceTable.cpp:336-370): per surfel the
 * instance whose table colour equals its instance colour (first match) and the ground-truth id kept in vImgCorr.w */
void orc_precision_recall(orc_t* o, int32_t* inst_num, int32_t* gt_num, int32_t* inst_gt_map)
{
    memset(inst_num, 0, ORC_NUM_INST * 4); memset(gt_num, 0, 256 * 4); memset(inst_gt_map, 0, 256 * ORC_NUM_INST * 4);
    for (int s = 0; s < o->n; s++) {
        int inst = -1;
        for (int i = 0; i < ORC_NUM_INST; i++) if (o->col[s * 2 + 1] == o->inst_color[i]) { inst = i; break; }
        if (inst >= 0) inst_num[inst]++;
        const int gt = (int)o->ic[s * 4 + 3];
        if (gt >= 0 && gt < 256) gt_num[gt]++;
        if (inst >= 0 && gt >= 0 && gt < 256) inst_gt_map[gt * ORC_NUM_INST + inst]++;
    }
}

/* InstanceFusion::renderProjectMap without the boxes: renderProjectFrameKernel, IF/Core/InstanceFusionCuda.cu:1432-1498 -- the instance colour of
 * the surfel under every pixel of the id image after fusion, RGBA float, black where no stable surfel is visible */
void orc_render_project_map(orc_t* o, float* out_rgba)
{
    for (int k = 0; k < o->P; k++) {
        const int id = o->ids_after[k];
        float* c = out_rgba + (size_t)k * 4;
        if (id > 0 && id < o->n) {
            const int v = (int)o->col[id * 2 + 1];
            c[0] = (float)((v >> 16) & 0xFF) / 255.0f; c[1] = (float)((v >> 8) & 0xFF) / 255.0f; c[2] = (float)(v & 0xFF) / 255.0f;
        } else c[0] = c[1] = c[2] = 0.0f;
        c[3] = 1.0f;
    }
}

/* maskCleanOverlapKernel, IF/Core/InstanceFusionCuda.cu:118-131 */
void orc_mask_clean_overlap(uint8_t* masks, int n, int w, int h)
{
    for (int p = 0; p < w * h; p++) {
        int flag = 0;
        for (int m = n - 1; m >= 0; m--) {
            if (flag) masks[(size_t)m * w * h + p] = 0;
            if (masks[(size_t)m * w * h + p]) flag = 1;
        }
    }
}

/* whetherDoSegmentation + checkProjectDepthAndInstanceKernel,
 * IF/Core/InstanceFusion.cpp:192-238, IF/Core/InstanceFusionCuda.cu:736-760 */
int orc_should_segment(orc_t* o, int frame)
{
    const int downsample = 10, fixedL = 2, fixedH = 45;
    int w = o->w, h = o->h;
    long long c0 = 0;
    int c1 = 0;
    for (int y = 0; y < h; y += downsample)
        for (int x = 0; x < w; x += downsample) {
            int id = o->ids_after[y * w + x];
            if (id > 0 && id < o->n) {
                for (int i = 0; i < ORC_VOTE_FLOATS; i++) {
                    int a, b;
                    orc_vote_decode(o->votes[(size_t)id * ORC_VOTE_FLOATS + i], &a, &b);
                    c0 += a; c0 += b;
                }
            } else c1++;
        }
    int count0 = (int)c0; /* int atomics wrap in the reference */
    int test1 = count0 > (w / downsample * h / downsample * 0.48 * 30);
    int test2 = c1 < (w / downsample * h / downsample * 0.2);
    if (test1 || test2) {
        if (frame - o->last_seg_frame > fixedH) { o->last_seg_frame = frame; return 1; }
        return 0;
    }
    if (frame - o->last_seg_frame > fixedL) { o->last_seg_frame = frame; return 1; }
    return 0;
}

/* getDepthThreshold, IF/Core/InstanceFusion.h:170-176 */
static float depth_threshold(int depth)
{
    float t = 0.074f * depth - 246.0f;
    t = fmaxf(50.0f, t);
    t = fminf(420.0f, t);
    return t;
}

/* filterAreaCompute + maskGeometricFilter, IF/Core/InstanceFusion.cpp:470-593 */
static void mask_geometric_filter(orc_t* o, const uint16_t* depth, uint8_t* masks, const uint8_t* ori, int nm, uint8_t* unavailable)
{
    int w = o->w, h = o->h, P = w * h;
    int* filterMap = (int*)malloc((size_t)P * 4);
    int* queue = (int*)malloc((size_t)P * 4 * 4 + 16);
    static const int StepX[4] = {0, 0, 1, -1}, StepY[4] = {1, -1, 0, 0};
    for (int i = 0; i < nm; i++) {
        if (unavailable[i]) continue;
        uint8_t* mask = masks + (size_t)i * P;
        const uint8_t* om = ori + (size_t)i * P;
        memset(filterMap, 0, (size_t)P * 4);
        float oriPoints = 0;
        for (int y = 1; y < h - 1; y++)
            for (int x = 1; x < w - 1; x++) {
                if (om[y * w + x]) oriPoints++;
                if (mask[y * w + x] && depth[y * w + x]) filterMap[y * w + x] = 1;
            }
        int areaFlag = 2, list[20], p = 0;
        for (int y = 1; y < h - 1; y++)
            for (int x = 1; x < w - 1; x++) {
                if (filterMap[y * w + x] != 1) continue;
                float points = 0;
                int front = 0, tail = 0;
                queue[front++] = y * w + x;
                while (front > tail) {
                    int now = queue[tail++];
                    int nx = now % w, ny = now / w;
                    if (filterMap[ny * w + nx] != 1) continue;
                    points++;
                    filterMap[ny * w + nx] = areaFlag;
                    for (int k = 0; k < 4; k++) {
                        int dx = nx + StepX[k], dy = ny + StepY[k];
                        float thr = depth_threshold(depth[ny * w + nx]);
                        if (filterMap[dy * w + dx] == 1 && (float)abs((int)depth[ny * w + nx] - (int)depth[dy * w + dx]) < thr) queue[front++] = dy * w + dx;
                    }
                }
                if (points / oriPoints > 0.25f /* eachGeoThreshold */) { if (p < 20) list[p++] = areaFlag; }
                areaFlag++;
            }
        float finalPoints = 0;
        for (int k = 0; k < P; k++) {
            int flag = 0;
            for (int j = 0; j < p; j++) if (filterMap[k] == list[j]) { flag = 1; break; }
            if (flag) { mask[k] = 255; finalPoints++; } else mask[k] = 0;
        }
        if (finalPoints / oriPoints < 0.65f /* totalGeoThreshold */) unavailable[i] = 1;
    }
    free(filterMap);
    free(queue);
}

/* getProjectInstanceList + computeProjectBoundingBox, IF/Core/InstanceFusionCuda.cu:781-974 */
static void project_bboxes(orc_t* o, const uint8_t* masks, int nm, int* maskBBox, int* projBBox)
{
    int w = o->w, h = o->h, P = w * h;
    for (int i = 0; i < NI; i++) { projBBox[i * 4] = w + 1; projBBox[i * 4 + 1] = -1; projBBox[i * 4 + 2] = h + 1; projBBox[i * 4 + 3] = -1; }
    for (int i = 0; i < nm; i++) { maskBBox[i * 4] = w + 1; maskBBox[i * 4 + 1] = -1; maskBBox[i * 4 + 2] = h + 1; maskBBox[i * 4 + 3] = -1; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int id = o->ids_after[y * w + x];
            if (!(id > 0 && id < o->n)) continue; /* projected counters are all -1 */
            int cnt[NI];
            for (int i = 0; i < ORC_VOTE_FLOATS; i++) orc_vote_decode(o->votes[(size_t)id * ORC_VOTE_FLOATS + i], &cnt[2 * i], &cnt[2 * i + 1]);
            if (cnt[0] == -1) continue; /* instanceProjectMap[y*width+x] != -1 test (:915) */
            int maxNum = 0, maxID = -1;
            for (int q = 0; q < NI; q++) if (cnt[q] > maxNum) { maxNum = cnt[q]; maxID = q; }
            if (maxID != -1) {
                int* b = &projBBox[maxID * 4];
                if (x < b[0]) b[0] = x; if (x > b[1]) b[1] = x; if (y < b[2]) b[2] = y; if (y > b[3]) b[3] = y;
            }
            for (int m = 0; m < nm; m++)
                if (masks[(size_t)m * P + y * w + x] > 0) {
                    int* b = &maskBBox[m * 4];
                    if (x < b[0]) b[0] = x; if (x > b[1]) b[1] = x; if (y < b[2]) b[2] = y; if (y > b[3]) b[3] = y;
                }
        }
}

/* computeCompareMap, IF/Core/InstanceFusion.cpp:595-651 (last match wins; instance 0 never matches) */
static void compare_map(orc_t* o, const int* maskBBox, const int* projBBox, const int32_t* class_ids, int nm, uint8_t* unavailable, int* cmp)
{
    for (int m = 0; m < nm; m++) {
        int minX_m = maskBBox[m * 4], maxX_m = maskBBox[m * 4 + 1], minY_m = maskBBox[m * 4 + 2], maxY_m = maskBBox[m * 4 + 3];
        if (maxX_m <= minX_m || maxY_m <= minY_m || unavailable[m]) { unavailable[m] = 1; continue; }
        int best = -1;
        for (int q = 0; q < NI; q++) {
            if (o->inst_class[q] == -1 || class_ids[m] != o->inst_class[q]) continue;
            int minX_i = projBBox[q * 4], maxX_i = projBBox[q * 4 + 1], minY_i = projBBox[q * 4 + 2], maxY_i = projBBox[q * 4 + 3];
            if (maxX_i <= minX_i || maxY_i <= minY_i) continue;
            float IW = (float)((maxX_i < maxX_m ? maxX_i : maxX_m) - (minX_i > minX_m ? minX_i : minX_m));
            float IH = (float)((maxY_i < maxY_m ? maxY_i : maxY_m) - (minY_i > minY_m ? minY_i : minY_m));
            if (IW <= 0 || IH <= 0) continue;
            float I = IW * IH;
            float U = (float)(((maxX_i - minX_i) * (maxY_i - minY_i)) + ((maxX_m - minX_m) * (maxY_m - minY_m))) - I;
            if (I / U > 0.5f /* compareThreshold */) best = q;
        }
        if (best > 0) cmp[best + m * NI] = 1;
    }
}

/* getInstanceTableCleanList, IF/Core/InstanceTable.cpp:185-224 (score = sum count, 20 lowest) */
static void clean_list(int* maxv, int* sumv, int* out)
{
    int order[NI];
    for (int i = 0; i < NI; i++) order[i] = i;
    for (int i = 0; i < NI; i++)
        for (int j = i + 1; j < NI; j++)
            if ((float)sumv[j] < (float)sumv[i]) {
                int t = maxv[j]; maxv[j] = maxv[i]; maxv[i] = t;
                t = order[j]; order[j] = order[i]; order[i] = t;
                t = sumv[j]; sumv[j] = sumv[i]; sumv[i] = t;
            }
    for (int i = 0; i < NI; i++) out[i] = 0;
    for (int i = 0; i < 20 /* cleanNum */; i++) out[order[i]] = 1;
}

static int first_not_used(orc_t* o)
{
    for (int i = 0; i < NI; i++) if (o->inst_class[i] == -1) return i;
    return -1;
}

/* InstanceFusion::processInstance, IF/Core/InstanceFusion.cpp:655-1067 with the Mask-RCNN call
 * replaced by the caller's pre-computed masks.  Superpixel refinement (steps -1_1..-1_3) is in
 * orc_slic.c and is applied when do_knn bit 1 (value 2) is set. */
int orc_superpixel_refine(orc_t* o, const uint8_t* rgb, const uint16_t* depth, uint8_t* masks, int nm, int frame);
void orc_knn_vote(orc_t* o, int32_t* nbr);

int orc_process_segmentation(orc_t* o, const uint8_t* rgb, const uint16_t* depth,
                             const uint8_t* masks_in, const int32_t* class_ids, int nm, int frame,
                             int flags)
{
    int w = o->w, h = o->h, P = w * h, n = o->n;
    if (nm == 0 || n == 0) return 0;
    const double t_begin = orc_now_ms();
    uint8_t* masks = (uint8_t*)malloc((size_t)nm * P);
    uint8_t* ori = (uint8_t*)malloc((size_t)nm * P);
    memcpy(masks, masks_in, (size_t)nm * P);
    memcpy(ori, masks_in, (size_t)nm * P); /* "BAK ORI MASK" :705-706, before clean-overlap */
    uint8_t* unavailable = (uint8_t*)calloc((size_t)nm, 1);
    orc_mask_clean_overlap(masks, nm, w, h);
    if (flags & 2) orc_superpixel_refine(o, rgb, depth, masks, nm, frame);

    int* maskBBox = (int*)malloc((size_t)nm * 16);
    int projBBox[NI * 4];
    int* cmp = (int*)calloc((size_t)nm * NI, 4);
    project_bboxes(o, masks, nm, maskBBox, projBBox);
    compare_map(o, maskBBox, projBBox, class_ids, nm, unavailable, cmp);

    /* getProjectDepthMapKernel, IF/Core/InstanceFusionCuda.cu:977-996 */
    uint16_t* pdm = (uint16_t*)calloc((size_t)P, 2);
    for (int k = 0; k < P; k++) {
        int id = o->ids_after[k];
        if (id > 0 && id < n) {
            float dx = o->pose[3] - o->pc[id * 4], dy = o->pose[7] - o->pc[id * 4 + 1], dz = o->pose[11] - o->pc[id * 4 + 2];
            pdm[k] = (uint16_t)(sqrtf(dx * dx + dy * dy + dz * dz) * 1186 /* depthRatio */);
        }
    }
    mask_geometric_filter(o, pdm, masks, ori, nm, unavailable);

    for (int m = 0; m < nm; m++) {
        int exist = 0;
        for (int q = 0; q < NI; q++) if (cmp[q + m * NI] == 1) { exist = 1; break; }
        if (!exist && !unavailable[m]) {
            int empty = first_not_used(o);
            if (empty == -1) {
                o->clean_times++;
                /* computeMaxCountInMapKernel :1012-1036 */
                int maxv[NI], sumv[NI], cl[NI];
                for (int q = 0; q < NI; q++) { maxv[q] = 0; sumv[q] = 0; }
                for (int i = 0; i < n; i++)
                    for (int k = 0; k < ORC_VOTE_FLOATS; k++) {
                        int a, b;
                        orc_vote_decode(o->votes[(size_t)i * ORC_VOTE_FLOATS + k], &a, &b);
                        if (a > maxv[2 * k]) maxv[2 * k] = a;
                        if (b > maxv[2 * k + 1]) maxv[2 * k + 1] = b;
                        if (a) sumv[2 * k] += a;
                        if (b) sumv[2 * k + 1] += b;
                    }
                clean_list(maxv, sumv, cl);
                for (int q = 0; q < NI; q++) if (cl[q] == 1) o->inst_class[q] = -1;
                /* cleanInstanceTableMapKernel :1057-1084 */
                for (int i = 0; i < n; i++)
                    for (int k = 0; k < ORC_VOTE_FLOATS; k++) {
                        if (!cl[2 * k] && !cl[2 * k + 1]) continue;
                        int a, b;
                        float* f = &o->votes[(size_t)i * ORC_VOTE_FLOATS + k];
                        orc_vote_decode(*f, &a, &b);
                        if (cl[2 * k]) a = 0;
                        if (cl[2 * k + 1]) b = 0;
                        *f = orc_vote_encode(a, b);
                    }
                project_bboxes(o, masks, nm, maskBBox, projBBox);
                memset(cmp, 0, (size_t)nm * NI * 4);
                compare_map(o, maskBBox, projBBox, class_ids, nm, unavailable, cmp);
                empty = first_not_used(o);
            }
            if (empty >= 0) { /* registerInstanceTable :231-235 */
                o->inst_class[empty] = class_ids[m];
                cmp[empty + m * NI] = 1;
            }
        }
        /* updateSurfelMapInstanceKernel :1100-1139, deleteNum = -1 */
        for (int q = 0; q < NI; q++) {
            if (cmp[q + m * NI] != 1) continue;
            const uint8_t* mask = masks + (size_t)m * P;
            for (int k = 0; k < P; k++) {
                if (!(mask[k] > 0)) continue;
                int id = o->ids_after[k];
                if (!(id > 0 && id < n)) continue;
                float* f = &o->votes[(size_t)id * ORC_VOTE_FLOATS + q / 2];
                int a, b;
                orc_vote_decode(*f, &a, &b);
                if (q % 2 == 0) a += (m + 1); else b += (m + 1);
                if (a >= 65535) a = 65535;
                if (b >= 65535) b = 65535;
                *f = orc_vote_encode(a, b);
            }
        }
    }

    /* countAndColourSurfelMapKernel :1158-1200 */
    const float defaultColor = 7434609;
    for (int i = 0; i < n; i++) {
        int best = -1, bestCount = 0;
        for (int k = 0; k < ORC_VOTE_FLOATS; k++) {
            int a, b;
            orc_vote_decode(o->votes[(size_t)i * ORC_VOTE_FLOATS + k], &a, &b);
            if (bestCount < a) { bestCount = a; best = 2 * k; }
            if (bestCount < b) { bestCount = b; best = 2 * k + 1; }
        }
        o->labels[i] = best;
        float* ic = &o->col[i * 2 + 1];
        if (*ic == 0 || *ic == defaultColor) *ic = (best != -1) ? o->inst_color[best] : defaultColor;
    }
    free(masks); free(ori); free(unavailable); free(maskBBox); free(cmp); free(pdm);
    if (flags & 1) orc_knn_vote(o, NULL); /* isflann, :1051 */
    o->stage_ms[2] += orc_now_ms() - t_begin;
    return 0;
}

/* flannKnnVoteSurfelMap + mapKnnVoteColourKernel, IF/Core/InstanceFusion.cpp:1070-1163, IF/Core/InstanceFusionCuda.cu:1237-1340:
 * every surfel takes the colour of the instance that most of its 10 nearest surfels (the query set is the indexed set, so
 * itself included) are labelled with; labels = bestIDInEachSurfel of the last count/colour pass; first maximum; no labelled
 * neighbour -> unchanged.  FLANN's exact kd-tree search (REF/deps/flann-1.8.4) is restated as a brute-force search; ties in
 * distance, which FLANN leaves unspecified, go to the lower index.  nbr (optional): n x 10 neighbour indices, -1 = none. */
void orc_knn_vote(orc_t* o, int32_t* nbr)
{
    const int n = o->n, K = 10;
    float* newcol = (float*)malloc((size_t)n * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        float bd[10];
        int bi[10];
        for (int k = 0; k < K; k++) { bd[k] = INFINITY; bi[k] = 0x7FFFFFFF; }
        const float qx = o->pc[i * 4], qy = o->pc[i * 4 + 1], qz = o->pc[i * 4 + 2];
        for (int j = 0; j < n; j++) {
            const float ex = o->pc[j * 4] - qx, ey = o->pc[j * 4 + 1] - qy, ez = o->pc[j * 4 + 2] - qz;
            float d = (ex * ex + ey * ey) + ez * ez;
            if (!(d < bd[K - 1] || (d == bd[K - 1] && j < bi[K - 1]))) continue;
            int jj = j;
            for (int k = 0; k < K; k++)
                if (d < bd[k] || (d == bd[k] && jj < bi[k])) { float td = bd[k]; int ti = bi[k]; bd[k] = d; bi[k] = jj; d = td; jj = ti; }
        }
        float temp[ORC_NUM_INST];
        for (int q = 0; q < ORC_NUM_INST; q++) temp[q] = 0;
        for (int k = 0; k < K; k++) {
            if (nbr) nbr[(size_t)i * K + k] = bi[k] == 0x7FFFFFFF ? -1 : bi[k];
            if (bi[k] == 0x7FFFFFFF) continue;
            if (o->labels[bi[k]] >= 0) temp[o->labels[bi[k]]]++;
        }
        int maxNum = -1, maxID = -1;
        for (int q = 0; q < ORC_NUM_INST; q++)
            if (temp[q] > maxNum) { maxID = q; maxNum = (int)temp[q]; }
        newcol[i] = maxNum > 0 ? o->inst_color[maxID] : o->col[i * 2 + 1];
    }
    for (int i = 0; i < n; i++) o->col[i * 2 + 1] = newcol[i];
    free(newcol);
}

/* test hook: the flood fill alone (depth = model depth map as getProjectDepthMap produces it) */
void orc_test_mask_geometric_filter(orc_t* o, const uint16_t* depth, uint8_t* masks, const uint8_t* ori, int nm, uint8_t* unavailable)
{
    mask_geometric_filter(o, depth, masks, ori, nm, unavailable);
}

/* =========================================================================== f-4: 3-D boxes and per-instance point clouds
 * InstanceFusion::computeMapBoundingBox / getInstancePointCloud (IF/Core/InstanceFusion.cpp:1261-1590) and their kernels
 * (IF/Core/InstanceFusionCuda.cu:1555-2083), restated sequentially: the cell loop of testAllSurfelNormalVoteKernel is evaluated as
 * written (cos / sin of every cell for every surfel would be the same values every time: they are computed once). */
#define BB_SEG 18
#define BB_CELLS (BB_SEG * BB_SEG * 2)
static int bb_instance_of(const orc_t* o, int s)
{
    for (int i = 0; i < ORC_NUM_INST; i++) if (o->col[s * 2 + 1] == o->inst_color[i]) return i;
    return -1;
}
static void bb_norm3(float* v) { const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; }
static void bb_cross3(const float* a, const float* b, float* r) { r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0]; }
/* rodriguesRotation, IF/Core/InstanceFusionCuda.cu:94-115 */
static void bb_rot(float angle, const float* v, const float* k, float* r)
{
    const float c = cosf(angle), s = sinf(angle), kv = v[0] * k[0] + v[1] * k[1] + v[2] * k[2];
    const float cr[3] = {v[1] * k[2] - v[2] * k[1], v[2] * k[0] - v[0] * k[2], v[0] * k[1] - v[1] * k[0]};
    for (int q = 0; q < 3; q++) r[q] = c * v[q] + (1 - c) * kv * k[q] + s * cr[q];
}
static void bb_set(const float* bx, const float* by, const float* bz, float* m)   /* matrixSetCoordinate, shift == 0 */
{
    m[0] = bx[0]; m[1] = by[0]; m[2] = bz[0]; m[3] = 0; m[4] = bx[1]; m[5] = by[1]; m[6] = bz[1]; m[7] = 0;
    m[8] = bx[2]; m[9] = by[2]; m[10] = bz[2]; m[11] = 0; m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1;
}
static void bb_inv(const float* m, float* o)
{
    const double a[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
    const double inv[9] = {c00 * id, (a[2] * a[7] - a[1] * a[8]) * id, (a[1] * a[5] - a[2] * a[4]) * id, c01 * id, (a[0] * a[8] - a[2] * a[6]) * id,
                           (a[2] * a[3] - a[0] * a[5]) * id, c02 * id, (a[1] * a[6] - a[0] * a[7]) * id, (a[0] * a[4] - a[1] * a[3]) * id};
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) o[r * 4 + c] = (float)inv[r * 3 + c]; o[r * 4 + 3] = 0; }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}
/* votes, ground normal, frames, their inverses: everything up to testAllSurfelFindBBox */
static void bb_prepare(orc_t* o, int* gvote, float* gn, float* gc, float* instm, float* gc_inv, float* inst_inv)
{
    const float pi = 3.1415926f;
    float cx[BB_CELLS], cy[BB_CELLS], cz[BB_CELLS], cR[BB_CELLS];
    int p = 0;
    for (int i = -BB_SEG / 2; i < BB_SEG / 2; i++) {
        const float tv = i * pi / BB_SEG, dv = cosf(tv), yv = sinf(tv), tmid = (i + 0.5f) * pi / BB_SEG, dm = cosf(tmid), ym = sinf(tmid);
        for (int j = 0; j < 2 * BB_SEG; j++) {
            const float av = j * pi / BB_SEG, xv = dv * cosf(av), zv = dv * sinf(av), am = (j + 0.5f) * pi / BB_SEG, xm = dm * cosf(am), zm = dm * sinf(am);
            const float dx = xv - xm, dy = yv - ym, dz = zv - zm;
            cx[p] = xm; cy[p] = ym; cz[p] = zm; cR[p] = sqrtf(dx * dx + dy * dy + dz * dz);
            p++;
        }
    }
    int* ivote = (int*)calloc((size_t)ORC_NUM_INST * BB_CELLS, sizeof(int));
    memset(gvote, 0, BB_CELLS * sizeof(int));
    for (int s = 0; s < o->n; s++) {
        const int inst = bb_instance_of(o, s);
        float nx = o->nr[s * 4], ny = -o->nr[s * 4 + 1], nz = -o->nr[s * 4 + 2];
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        for (int q = 0; q < BB_CELLS; q++) {
            const float dx = nx - cx[q], dy = ny - cy[q], dz = nz - cz[q];
            if (sqrtf(dx * dx + dy * dy + dz * dz) < cR[q]) { gvote[q]++; if (inst != -1) ivote[inst * BB_CELLS + q]++; }
        }
    }
    gn[0] = 0; gn[1] = -1; gn[2] = 0;
    int vmax = -1, vid = -1;
    for (int i = 0; i < BB_CELLS; i++) if (gvote[i] > vmax) { vmax = gvote[i]; vid = i; }
    if (vid != -1) {
        const int i = vid / (2 * BB_SEG) - BB_SEG / 2, j = vid % (2 * BB_SEG) - 1;
        const float theta = (i + 0.5f) * pi / BB_SEG, alpha = (j + 0.5f) * pi / BB_SEG, d = cosf(theta);
        gn[0] = d * cosf(alpha); gn[1] = sinf(theta); gn[2] = d * sinf(alpha);
        bb_norm3(gn);
    }
    float by[3] = {gn[0], gn[1], gn[2]}, bx[3], bz[3];
    bb_norm3(by);
    const float setZ[3] = {0, 0, 1};
    bb_cross3(setZ, by, bx); bb_norm3(bx);
    bb_cross3(by, bx, bz); bb_norm3(bz);
    bb_set(bx, by, bz, gc);
    const float step = pi / BB_SEG;
    for (int id = 0; id < ORC_NUM_INST; id++) {
        int crossv[2 * BB_SEG];
        memset(crossv, 0, sizeof(crossv));
        const float oriZ[3] = {0, 0, -1};
        float oriX[3];
        bb_cross3(by, oriZ, oriX); bb_norm3(oriX);
        for (int i = 0; i < BB_CELLS; i++) {
            const int v = ivote[id * BB_CELLS + i];
            if (v == 0) continue;
            const int a = i / (2 * BB_SEG) - BB_SEG / 2, b = i % (2 * BB_SEG) - 1;
            const float theta = (a + 0.5f) * pi / BB_SEG, alpha = (b + 0.5f) * pi / BB_SEG, d = cosf(theta);
            float tz[3] = {d * cosf(alpha), sinf(theta), d * sinf(alpha)};
            bb_norm3(tz);
            const float cs = by[0] * tz[0] + by[1] * tz[1] + by[2] * tz[2];
            if (cs > 0.525f || -cs > 0.525f) continue;
            float best = 999999.9f;
            int bj = -1;
            for (int j = 0; j < 2 * BB_SEG; j++) {
                float rv[3];
                bb_rot(j * step, oriX, by, rv); bb_norm3(rv);
                const float dx = rv[0] - tz[0], dy = rv[1] - tz[1], dz = rv[2] - tz[2], dist = sqrtf(dx * dx + dy * dy + dz * dz);
                if (dist < best) { bj = j; best = dist; }
            }
            crossv[bj] += v;
        }
        int m = 0, mid = 0;
        for (int i = 0; i < 2 * BB_SEG; i++) if (crossv[i] > m) { m = crossv[i]; mid = i; }
        float rv[3];
        bb_rot(mid * step, oriX, by, rv); bb_norm3(rv);
        bb_cross3(rv, by, bx); bb_norm3(bx);
        bb_cross3(by, bx, bz); bb_norm3(bz);
        bb_set(bx, by, bz, instm + 16 * id);
    }
    bb_inv(gc, gc_inv);
    for (int i = 0; i < ORC_NUM_INST; i++) bb_inv(instm + 16 * i, inst_inv + 16 * i);
    free(ivote);
}
void orc_map_bounding_boxes(orc_t* o, int bbox_type, float ratio, float* boxes, float* ground_normal3, float* gc16, float* inst16, int32_t* ground_votes)
{
    int gvote[BB_CELLS];
    float gn[3], gc[16], instm[16 * ORC_NUM_INST], gci[16], insti[16 * ORC_NUM_INST];
    bb_prepare(o, gvote, gn, gc, instm, gci, insti);
    int box[ORC_NUM_INST * 6];
    for (int i = 0; i < ORC_NUM_INST; i++) { box[i * 6] = box[i * 6 + 2] = box[i * 6 + 4] = 999999999; box[i * 6 + 1] = box[i * 6 + 3] = box[i * 6 + 5] = -999999999; }
    for (int s = 0; s < o->n; s++) {   /* testAllSurfelFindBBoxKernel :1831-1901 */
        const int inst = bb_instance_of(o, s);
        if (inst == -1) continue;
        const float* P = &o->pc[s * 4];
        const float* M = bbox_type ? gci : insti + 16 * inst;
        const float g[3] = {M[0] * P[0] + M[1] * P[1] + M[2] * P[2] + M[3] * 1.0f, M[4] * P[0] + M[5] * P[1] + M[6] * P[2] + M[7] * 1.0f,
                            M[8] * P[0] + M[9] * P[1] + M[10] * P[2] + M[11] * 1.0f};
        for (int a = 0; a < 3; a++) {
            const int v = orc_f2i_rz(g[a] * ratio);
            int mn = v, mx = v;
            if (mn > P[a] * ratio) mn--;   /* compares with the WORLD coordinate, as the reference does */
            if (mx < P[a] * ratio) mx++;
            if (mn < box[inst * 6 + 2 * a]) box[inst * 6 + 2 * a] = mn;
            if (mx > box[inst * 6 + 2 * a + 1]) box[inst * 6 + 2 * a + 1] = mx;
        }
    }
    for (int k = 0; k < ORC_NUM_INST * 6; k++) boxes[k] = box[k] / ratio;
    if (ground_normal3) memcpy(ground_normal3, gn, 12);
    if (gc16) memcpy(gc16, gc, 64);
    if (inst16) memcpy(inst16, instm, sizeof(instm));
    if (ground_votes) memcpy(ground_votes, gvote, sizeof(gvote));
}
/* mapCountInstanceByInstColor + getSurfelToInstanceBuffer (:1920-2066), records of one instance in map order */
int orc_instance_point_cloud(orc_t* o, int bbox_type, int32_t* counts, int inst, float* out10, int max_records)
{
    int gvote[BB_CELLS];
    float gn[3], gc[16], instm[16 * ORC_NUM_INST], gci[16], insti[16 * ORC_NUM_INST];
    bb_prepare(o, gvote, gn, gc, instm, gci, insti);
    memset(counts, 0, ORC_NUM_INST * sizeof(int32_t));
    int w = 0;
    for (int s = 0; s < o->n; s++) {
        const int q = bb_instance_of(o, s);
        if (q == -1) continue;
        counts[q]++;
        if (q != inst || w >= max_records) continue;
        const float* P = &o->pc[s * 4];
        const float* M = bbox_type ? gci : insti + 16 * inst;
        float nx = o->nr[s * 4], ny = -o->nr[s * 4 + 1], nz = -o->nr[s * 4 + 2];
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        float* r = out10 + (size_t)w * 10;
        r[0] = (float)s;
        r[1] = M[0] * P[0] + M[1] * P[1] + M[2] * P[2] + M[3] * 1.0f; r[2] = M[4] * P[0] + M[5] * P[1] + M[6] * P[2] + M[7] * 1.0f; r[3] = M[8] * P[0] + M[9] * P[1] + M[10] * P[2] + M[11] * 1.0f;
        r[4] = M[0] * nx + M[1] * ny + M[2] * nz + M[3] * 1.0f; r[5] = M[4] * nx + M[5] * ny + M[6] * nz + M[7] * 1.0f; r[6] = M[8] * nx + M[9] * ny + M[10] * nz + M[11] * 1.0f;
        const int c = orc_f2i_rz(o->col[s * 2]);
        r[7] = (float)(c >> 16 & 0xFF) / 255.0f; r[8] = (float)(c >> 8 & 0xFF) / 255.0f; r[9] = (float)(c & 0xFF) / 255.0f;
        w++;
    }
    return w;
}
