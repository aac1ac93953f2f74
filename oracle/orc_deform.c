/* oracle/orc_deform.c -- CPU restatement of the device-side loop-closure hooks of ElasticFusion (test infrastructure only, see orc.h):
 *   - Deformation::sampleGraphModel (EF/Deformation.cpp:224-337, sample.vert/.geom): every 5000th surfel -> x, y, z, init time
 *   - the constraint samples of EF/ElasticFusion.cpp:568-598 (resize.vertex / resize.time + the two world points per sample)
 *   - the deformation-graph application inside clean (copy_unstable.vert:178-374), incl. the re-rendered model depth of
 *     IndexMap::synthesizeDepth (EF/IndexMap.cpp:576-648, EF/ElasticFusion.cpp:667-676) that refreshes time stamps
 * The graph OPTIMISATION (EF/Utils/DeformationGraph.cpp, cholmod) is host code of the reference and is not restated: the graph is an input. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_internal.h"
#include "orc_math.h"

void orc_set_deformation(orc_t* o, const float* graph16, int n_nodes, int is_fern)
{
    free(o->graph);
    o->graph = NULL;
    o->graph_nodes = 0;
    if (n_nodes > 0) {
        o->graph = (float*)malloc((size_t)n_nodes * 64);
        memcpy(o->graph, graph16, (size_t)n_nodes * 64);
        o->graph_nodes = n_nodes;
    }
    o->graph_is_fern = is_fern;
}

/* sample.geom:47-53: id % 5000 == 0 (ids are positions in the compacted buffer) */
int orc_sample_graph_model(orc_t* o, float* out_xyzt, int max_n)
{
    int m = 0;
    for (int i = 0; i < o->n && m < max_n; i += 5000) {
        out_xyzt[m * 4 + 0] = o->pc[i * 4]; out_xyzt[m * 4 + 1] = o->pc[i * 4 + 1]; out_xyzt[m * 4 + 2] = o->pc[i * 4 + 2];
        out_xyzt[m * 4 + 3] = o->tm[i * 2];
        m++;
    }
    return m;
}

/* EF/ElasticFusion.cpp:568-598: the ACTIVE vertex render and the INACTIVE time render resampled to (w/20) x (h/20) (nearest texel of the
 * sample centre, the rule of dense_enough()), column by column; a sample with 0 < z < maxDepth and an old time stamp gives the pair
 * worldRawPoint = currPose * v, worldModelPoint = estPose * v. */
int orc_loop_closure_constraints(orc_t* o, float* src3, float* dst3, int32_t* times, int max_n)
{
    const int w = o->w, h = o->h, rw = w / 20, rh = h / 20;
    const float* est = &o->lc[6];
    int m = 0;
    for (int i = 0; i < rw; i++)
        for (int j = 0; j < rh; j++) {
            const int sx = (i * w + w / 2) / rw, sy = (j * h + h / 2) / rh, k = sy * w + sx;
            const float* v = &o->pred_vertex[(size_t)k * 4];
            const int t = o->old_time[k];
            if (!(v[2] > 0 && v[2] < o->cfg.max_depth_processed && t > 0)) continue;
            if (m >= max_n) return m;
            for (int r = 0; r < 3; r++) {
                src3[m * 3 + r] = o->pose[r * 4] * v[0] + o->pose[r * 4 + 1] * v[1] + o->pose[r * 4 + 2] * v[2] + o->pose[r * 4 + 3] * 1.0f;
                dst3[m * 3 + r] = est[r * 4] * v[0] + est[r * 4 + 1] * v[1] + est[r * 4 + 2] * v[2] + est[r * 4 + 3] * 1.0f;
            }
            times[m] = t;
            m++;
        }
    return m;
}

/* currPose = estPose, EF/ElasticFusion.cpp:606 */
void orc_adopt_estimated_pose(orc_t* o) { memcpy(o->pose, &o->lc[6], 64); }

/* copy_unstable.vert:178-374 for one surviving surfel.  g: nodes x 16 floats (position 3, rotation 9 column-major, translation 3, time). */
void orc_deform_surfel(const float* g, int nodes, float* pc, float* nr, float initT, float* lastT, int time, float thr, int is_fern, const float* tinv,
                       const float* depth, int w, int h, float cx, float cy, float fx, float fy, float maxDepth)
{
    enum { K = 4, LOOK = 20 };
    int nearNodes[LOOK];
    float nearDists[LOOK];
    for (int i = 0; i < LOOK; i++) { nearNodes[i] = -1; nearDists[i] = 16777216.0f; }
    const int poseTime = (int)initT;
    int foundIndex = 0, imin = 0, imax = nodes - 1, imid = (imin + imax) / 2;
    while (imax >= imin) {
        imid = (imin + imax) / 2;
        const int nodeTime = (int)g[imid * 16 + 15];
        if (nodeTime < poseTime) imin = imid + 1;
        else if (nodeTime > poseTime) imax = imid - 1;
        else break;
    }
    imin = imin < nodes - 1 ? imin : nodes - 1;
    /* imax can be -1 after the search: the shader then samples left of the texture (clamped to texel 0, the x of node 0); its |time
     * difference| only matters when it is the smallest, which the index clamp below makes harmless -- restated with an index clamp */
    const int cmax = imax < 0 ? 0 : imax;
    const int nodeMin = (int)g[imin * 16 + 15], nodeMid = (int)g[imid * 16 + 15], nodeMax = (int)g[cmax * 16 + 15];
    if (abs(nodeMin - poseTime) <= abs(nodeMid - poseTime) && abs(nodeMin - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imin;
    else if (abs(nodeMid - poseTime) <= abs(nodeMin - poseTime) && abs(nodeMid - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imid;
    else foundIndex = cmax;
    if (foundIndex == nodes) foundIndex = nodes - 1;
    int nearNodeIndex = 0, distanceBack = 0;
    const v3 p = v3m(pc[0], pc[1], pc[2]);
    for (int j = foundIndex; j >= 0; j--) {
        const v3 d = v3sub(p, v3m(g[j * 16], g[j * 16 + 1], g[j * 16 + 2]));
        nearNodes[nearNodeIndex] = j;
        nearDists[nearNodeIndex] = sqrtf(v3dot(d, d));
        nearNodeIndex++;
        if (++distanceBack == LOOK / 2) break;
    }
    for (int j = foundIndex + 1; j < nodes; j++) {
        const v3 d = v3sub(p, v3m(g[j * 16], g[j * 16 + 1], g[j * 16 + 2]));
        nearNodes[nearNodeIndex] = j;
        nearDists[nearNodeIndex] = sqrtf(v3dot(d, d));
        nearNodeIndex++;
        if (++distanceBack == LOOK) break;
    }
    for (int i = 0; i < LOOK - 1; ++i)
        for (int j = i + 1; j < LOOK; ++j)
            if (nearDists[j] < nearDists[i]) {
                const float t = nearDists[i]; nearDists[i] = nearDists[j]; nearDists[j] = t;
                const int t2 = nearNodes[i]; nearNodes[i] = nearNodes[j]; nearNodes[j] = t2;
            }
    const float dMax = nearDists[K];
    float wgt[K], weightSum = 0;
    for (int j = 0; j < K; j++) {
        const float* n = &g[nearNodes[j] * 16];
        const v3 d = v3sub(p, v3m(n[0], n[1], n[2]));
        const float u = 1.0f - (sqrtf(v3dot(d, d)) / dMax);
        wgt[j] = u * u;
        weightSum += wgt[j];
    }
    for (int j = 0; j < K; j++) wgt[j] /= weightSum;
    v3 newPos = v3m(0, 0, 0), newNorm = v3m(0, 0, 0);
    for (int i = 0; i < K; i++) {
        const float* n = &g[nearNodes[i] * 16];
        const v3 gp = v3m(n[0], n[1], n[2]);
        /* mat3(column0, column1, column2): row-major form R[r][c] = n[3 + c*3 + r] */
        const float R[9] = {n[3], n[6], n[9], n[4], n[7], n[10], n[5], n[8], n[11]};
        const v3 tr = v3m(n[12], n[13], n[14]);
        const v3 q = v3add(v3add(m33mul(R, v3sub(p, gp)), gp), tr);
        newPos = v3add(newPos, v3scale(q, wgt[i]));
        float Ri[9];
        inv33_f(R, Ri);
        const float RiT[9] = {Ri[0], Ri[3], Ri[6], Ri[1], Ri[4], Ri[7], Ri[2], Ri[5], Ri[8]};
        newNorm = v3add(newNorm, v3scale(m33mul(RiT, v3m(nr[0], nr[1], nr[2])), wgt[i]));
    }
    pc[0] = newPos.x; pc[1] = newPos.y; pc[2] = newPos.z;
    const v3 nn = v3normalized(newNorm);
    nr[0] = nn.x; nr[1] = nn.y; nr[2] = nn.z;
    if (pc[3] > thr && !is_fern) {
        const v3 lp = v3m(tinv[0] * newPos.x + tinv[1] * newPos.y + tinv[2] * newPos.z + tinv[3],
                          tinv[4] * newPos.x + tinv[5] * newPos.y + tinv[6] * newPos.z + tinv[7],
                          tinv[8] * newPos.x + tinv[9] * newPos.y + tinv[10] * newPos.z + tinv[11]);
        const float x = ((fx * lp.x) / lp.z) + cx, y = ((fy * lp.y) / lp.z) + cy;
        if (lp.z > 0 && lp.z < maxDepth && x > 0 && y > 0 && x < (float)w && y < (float)h) {
            const float cur = depth[(int)floorf(y) * w + (int)floorf(x)];
            if (cur > 0.0f && lp.z < cur + 0.1f) *lastT = (float)time;
        }
    }
}
