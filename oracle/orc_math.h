/* oracle/orc_math.h -- small vector / solver helpers for the CPU oracle (test infrastructure only). */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <limits.h>
#include <float.h>

typedef struct { float x, y, z; } v3;

/* ---- exact, order-independent sums of the tracker's normal equations (the arithmetic contract shared with the HIP path,
 * instancefusion_amd/csrc/ifx_dev.h): every f32 product that enters one of the 29 (11) sums is rounded to a fixed grid 2^g
 * (g = e_i + e_j - 32 for a product of row entries i and j; e = binary exponent of the entry's working range) and the sums run
 * in f64, where every partial sum is an integer number of grid units below 2^53: each addition is exact, so the total is the
 * same for any order (rows, OpenMP chunks; on the GPU threads, waves, blocks, atomics).  The reference's own sums are an f32
 * tree whose shape depends on a per-GPU launch table (EF/Utils/GPUConfig.h:53-137): there is no single reference order. */
static const int ORC_E_ICP[7] = {0, 0, 0, 4, 4, 4, -3};
static const int ORC_E_RGB[7] = {11, 11, 11, 13, 13, 13, 3};   /* measured RMS on the reference's RGB-D pair: v 6..33, r 0.14 */
static const int ORC_E_SO3[4] = {16, 16, 16, 8};             /* measured RMS: jr 650..1750, r 36 */
#define ORC_EXACT_TERM_BITS 32
#define ORC_SO3_TERM_BITS 38
/* (t + M) - M with M = 1.5 * 2^(52 + g) rounds t to a multiple of 2^g (nearest even) */
static inline double orc_quant(float p, int g) {
    const double M = ldexp(1.5, 52 + g);
    double t = (double)p;
    t = t + M;
    return t - M;
}

static inline v3 v3m(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3sub(v3 a, v3 b) { return v3m(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3add(v3 a, v3 b) { return v3m(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3scale(v3 a, float s) { return v3m(a.x * s, a.y * s, a.z * s); }
/* EF/Cuda/operators.cuh:67-80 */
static inline v3 v3cross(v3 a, v3 b) {
    return v3m(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v3norm(v3 a) { return sqrtf(v3dot(a, a)); }
/* normalized(): EF/Cuda/operators.cuh:82-86 uses rsqrtf; restated as 1/sqrt (tolerance-level) */
static inline v3 v3normalized(v3 a) {
    float rn = 1.0f / sqrtf(v3dot(a, a));
    return v3m(a.x * rn, a.y * rn, a.z * rn);
}
/* row-major 3x3 times vector: EF/Cuda/operators.cuh:88-91 */
static inline v3 m33mul(const float* m, v3 a) {
    return v3m(m[0] * a.x + m[1] * a.y + m[2] * a.z, m[3] * a.x + m[4] * a.y + m[5] * a.z,
               m[6] * a.x + m[7] * a.y + m[8] * a.z);
}

static inline float orc_qnan(void) {
    uint32_t u = 0x7fffffffu; /* CUDART_NAN_F as written in EF/Cuda/cudafuncs.cu:130 */
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* __float2int_rn with CUDA semantics (NaN -> 0, saturating) */
static inline int orc_f2i_rn(float v) {
    if (isnan(v)) return 0;
    if (v >= 2147483648.0f) return INT_MAX;
    if (v <= -2147483648.0f) return INT_MIN;
    return (int)nearbyintf(v);
}
/* float -> int truncation with CUDA semantics */
static inline int orc_f2i_rz(float v) {
    if (isnan(v)) return 0;
    if (v >= 2147483648.0f) return INT_MAX;
    if (v <= -2147483648.0f) return INT_MIN;
    return (int)v;
}


/* Deterministic expf for x <= 0 (Cody-Waite reduction + degree-7 Taylor/Horner, no FMA
 * contraction).  The same code is used by the CPU oracle and the HIP kernels so that the stages
 * that call exp (bilateral weights, surfel confidence) agree bit for bit; it differs from libm's /
 * GLSL's exp by at most 2 ulp. */
static inline float ifx_expf(float x)
{
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 0.0f) x = 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.0f / 5040.0f;
    p = p * r + 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    int e = (int)n + 127;
    union { unsigned int u; float f; } s;
    s.u = (unsigned int)e << 23;
    return p * s.f;
}

/* ---- dense helpers (row-major) */
static inline void matmul_d(int n, const double* A, const double* B, double* C) {
    double T[16];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * n + j];
            T[i * n + j] = s;
        }
    memcpy(C, T, sizeof(double) * n * n);
}

/* rigid 4x4 inverse in double (resultRt is always rigid: EF/Utils/RGBDOdometry.cpp:424) */
static inline void rigid_inv_d(const double* M, double* O) {
    double T[16];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 4 + j] = M[j * 4 + i];
    for (int i = 0; i < 3; i++)
        T[i * 4 + 3] = -(T[i * 4 + 0] * M[3] + T[i * 4 + 1] * M[7] + T[i * 4 + 2] * M[11]);
    T[12] = T[13] = T[14] = 0;
    T[15] = 1;
    memcpy(O, T, sizeof(T));
}

/* general 3x3 inverse (float), as Eigen's Matrix3f::inverse (cofactor form) */
static inline void inv33_f(const float* m, float* o) {
    float c00 = m[4] * m[8] - m[5] * m[7];
    float c01 = m[5] * m[6] - m[3] * m[8];
    float c02 = m[3] * m[7] - m[4] * m[6];
    float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    float id = 1.0f / det;
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* LDLT with symmetric diagonal pivoting, following Eigen's LDLT (used by
 * EF/Utils/RGBDOdometry.cpp:368,552): pivots that are exactly ~0 give a zero component. */
#define ORC_DEFINE_LDLT(NAME, T, TINY)                                                         \
    static inline void NAME(int n, const T* Ain, const T* bin, T* x) {                         \
        T A[36], y[6];                                                                         \
        int p[6];                                                                              \
        for (int i = 0; i < n * n; i++) A[i] = Ain[i];                                         \
        for (int i = 0; i < n; i++) { p[i] = i; }                                              \
        for (int k = 0; k < n; k++) {                                                          \
            int piv = k;                                                                       \
            T big = (T)fabs((double)A[k * n + k]);                                             \
            for (int i = k + 1; i < n; i++) {                                                  \
                T v = (T)fabs((double)A[i * n + i]);                                           \
                if (v > big) { big = v; piv = i; }                                             \
            }                                                                                  \
            if (piv != k) {                                                                    \
                for (int j = 0; j < n; j++) { T t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; } \
                for (int j = 0; j < n; j++) { T t = A[j * n + k]; A[j * n + k] = A[j * n + piv]; A[j * n + piv] = t; } \
                int t = p[k]; p[k] = p[piv]; p[piv] = t;                                       \
            }                                                                                  \
            T d = A[k * n + k];                                                                \
            if (big <= (T)0) {                                                                 \
                for (int i = k + 1; i < n; i++) A[i * n + k] = 0;                              \
                continue;                                                                      \
            }                                                                                  \
            for (int i = k + 1; i < n; i++) A[i * n + k] = A[i * n + k] / d;                   \
            for (int i = k + 1; i < n; i++)                                                    \
                for (int j = k + 1; j < n; j++) A[i * n + j] -= A[i * n + k] * d * A[j * n + k]; \
        }                                                                                      \
        for (int i = 0; i < n; i++) y[i] = bin[p[i]];                                          \
        for (int i = 0; i < n; i++)                                                            \
            for (int j = 0; j < i; j++) y[i] -= A[i * n + j] * y[j];                           \
        for (int i = 0; i < n; i++) {                                                          \
            T d = A[i * n + i];                                                                \
            y[i] = (fabs((double)d) > (double)(TINY)) ? y[i] / d : (T)0;                       \
        }                                                                                      \
        for (int i = n - 1; i >= 0; i--)                                                       \
            for (int j = i + 1; j < n; j++) y[i] -= A[j * n + i] * y[j];                       \
        for (int i = 0; i < n; i++) x[p[i]] = y[i];                                            \
    }

ORC_DEFINE_LDLT(ldlt_solve_d, double, 1.0 / DBL_MAX)
ORC_DEFINE_LDLT(ldlt_solve_f, float, 1.0 / FLT_MAX)

/* OdometryProvider::rodrigues, EF/Utils/OdometryProvider.h:35-71 (double, row-major 3x3) */
static inline void rodrigues_d(const double* src, double* R) {
    double rx = src[0], ry = src[1], rz = src[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (theta >= DBL_EPSILON) {
        double c = cos(theta), s = sin(theta), c1 = 1.0 - c;
        double it = theta ? 1.0 / theta : 0.0;
        rx *= it; ry *= it; rz *= it;
        double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx_[k];
    }
}

#endif
