/* oracle/orc_testhooks.c -- exports of the oracle's internal helpers so tests can pin them against
 * numpy, plus a frame-feeding hook (test infrastructure only, see orc.h). */
#include "orc.h"
#include "orc_math.h"
#include "orc_internal.h"

float orc_test_expf(float x) { return ifx_expf(x); }
void orc_test_ldlt(int n, const double* A, const double* b, double* x) { ldlt_solve_d(n, A, b, x); }
void orc_test_ldlt_f(int n, const float* A, const float* b, float* x) { ldlt_solve_f(n, A, b, x); }
void orc_test_rodrigues(const double* v, double* R) { rodrigues_d(v, R); }
int orc_test_f2i_rn(float v) { return orc_f2i_rn(v); }

/* upload + preprocess only (EF/ElasticFusion.cpp:280-281,309-310); the map, pose and tick are untouched */
void orc_set_frame(orc_t* o, const uint8_t* rgb, const uint16_t* depth)
{
    memcpy(o->rgb, rgb, (size_t)o->P * 3);
    memcpy(o->depth_raw, depth, (size_t)o->P * 2);
    orc_bilateral(o->depth_raw, o->depth_filt, o->w, o->h, o->cfg.depth_cut);
    orc_metric(o->depth_raw, o->dm, o->w, o->h, o->cfg.depth_cut);
    orc_metric(o->depth_filt, o->dmf, o->w, o->h, o->cfg.depth_cut);
}

/* copies an id image into ids_after (lets tests drive the instance layer from a chosen id render) */
void orc_set_ids_after(orc_t* o, const int32_t* ids) { memcpy(o->ids_after, ids, (size_t)o->P * 4); }

/* number of OpenMP threads for the oracle's parallel loops (bench.py's cpu_baseline: one core / all cores) */
#include <omp.h>
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
