// ifx_dev.h -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#define IFX_WAVE 64
#define IFX_NUM_PYRS 3
#define IFX_NI 96
#define IFX_VF 48
// The vote counters of a surfel are ONE 192-byte record (48 floats = 12 float4, the layout of the C API's [n][48] arrays): everything that goes from a pixel's id to
// its surfel's votes -- the projected boxes, the label scan under the id image, whetherDoSegmentation's vote mass -- is a gather, and a record is 1.5 cache lines where
// twelve planes were twelve (round 3; gfx950 moves a 128-byte line per 16-byte gather, profiles/archive/r03_b_pmc_calibration.json).
#define VOTE4(votes, id, q) (votes)[(size_t)(id) * 12 + (q)]
#define VOTEF(votes, id, fi) (votes)[(size_t)(id) * IFX_VF + (fi)]
#define IFX_MAX_SPRITE 512.0f

struct v3 { float x, y, z; };
__host__ __device__ inline v3 v3m(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ inline v3 operator-(v3 a, v3 b) { return v3m(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline v3 operator+(v3 a, v3 b) { return v3m(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline v3 operator*(v3 a, float s) { return v3m(a.x * s, a.y * s, a.z * s); }
__host__ __device__ inline v3 cross(v3 a, v3 b) { return v3m(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__host__ __device__ inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline float norm(v3 a) { return sqrtf(dot(a, a)); }
__device__ inline v3 normalized(v3 a) { float rn = 1.0f / sqrtf(dot(a, a)); return v3m(a.x * rn, a.y * rn, a.z * rn); }

// 3x3 row-major matrix passed by value in kernel arguments
struct m33 { float m[9]; };
__host__ __device__ inline v3 mul(const m33& M, v3 a)
{
    return v3m(M.m[0] * a.x + M.m[1] * a.y + M.m[2] * a.z, M.m[3] * a.x + M.m[4] * a.y + M.m[5] * a.z, M.m[6] * a.x + M.m[7] * a.y + M.m[8] * a.z);
}
__host__ __device__ inline v3 mulp(const float* M, v3 a)
{
    return v3m(M[0] * a.x + M[1] * a.y + M[2] * a.z, M[3] * a.x + M[4] * a.y + M[5] * a.z, M[6] * a.x + M[7] * a.y + M[8] * a.z);
}
// 4x4 row-major rigid transform applied to a point / direction
__host__ __device__ inline v3 xf_point(const float* m, v3 p)
{
    return v3m(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__host__ __device__ inline v3 xf_dir(const float* m, v3 p)
{
    return v3m(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z, m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
__host__ __device__ inline void pose_inverse(const float* p, float* o)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[i * 4 + j] = p[j * 4 + i];
    for (int i = 0; i < 3; i++) o[i * 4 + 3] = -(o[i * 4] * p[3] + o[i * 4 + 1] * p[7] + o[i * 4 + 2] * p[11]);
    o[12] = o[13] = o[14] = 0.f;
    o[15] = 1.f;
}

__device__ inline float qnan_f() { return __int_as_float(0x7fffffff); }

// __float2int_rn / truncation with the CUDA semantics the reference relies on (NaN -> 0, saturating)
__device__ inline int f2i_rn(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)rintf(v);
}
__device__ inline int f2i_rz(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
__host__ __device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }


// Deterministic expf for x <= 0 (Cody-Waite reduction + degree-7 Taylor/Horner, no FMA
// contraction).  The same code is used by the CPU oracle (oracle/orc_math.h) and the HIP kernels
// so that the stages that call exp (bilateral weights, surfel confidence) agree bit for bit; it
// differs from libm's / GLSL's exp by at most 2 ulp.
__host__ __device__ inline float ifx_expf(float x)
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

// The window loop of data.vert:151-153 and copy_unstable.vert:110-112 AS THE SHADER TEXT EVALUATES IT in IEEE f32:
//     for (float i = c - (scale * step * windowMultiplier); i < c + (scale * step * windowMultiplier); i += step)     step = (1 / (size * scale)) * 0.5, scale = 1
// and the texel each tap reads under GL_NEAREST: floor(u * size), clamped to the edge (GL 4.5 section 8.14.2).  In exact arithmetic the loop makes four trips, at -1, -1/2, 0,
// +1/2 texels from c; in f32 the accumulated `i += step` falls short of the bound in a fraction of the cases and a FIFTH tap, one texel beyond c, is taken -- and a tap that
// sits on a texel edge goes to whichever side the f32 product falls.  Pinned by executing the reference's shaders (tests/golden/gl_map_passes.npz, tools/make_golden_gl.py;
// the same function, statement for statement, in oracle/orc_map.c).  Returns the number of taps (4 or 5).
#define IFX_MAX_TAPS 5   // (the span is four steps wide: rounding can add a trip, never two, and never takes one away; the oracle aborts if a loop ever made a sixth)
__host__ __device__ __forceinline__ int window_taps(float c, float size, int n, int* tex)
{
    const float scale = 1.0f, wm = 2.0f;
    const float step = (1.0f / (size * scale)) * 0.5f;
    const float lo = c - (scale * step * wm), hi = c + (scale * step * wm);
    int k = 0;
    float i = lo;
#pragma unroll
    for (int it = 0; it < IFX_MAX_TAPS; it++) {   // (the float loop, unrolled with a predicate so that `tex` stays in registers)
        const bool in = i < hi;
        int t = (int)floorf(i * size);
        t = t < 0 ? 0 : (t > n - 1 ? n - 1 : t);
        tex[it] = in ? t : -1;
        k += in ? 1 : 0;
        i = in ? i + step : i;
    }
    return k;
}
// the texcoord attribute of pixel column / row i: the uvo buffer of GlobalModel (EF/GlobalModel.cpp:103-119), float(i) / size + 1.0 / (2 * size) evaluated in double, stored as float
__host__ __device__ __forceinline__ float uvo_coord(int i, int size) { return (float)((double)((float)i / (float)size) + 1.0 / (2 * (double)(float)size)); }
// The pixel a 1-pixel GL point at window coordinate u lands on: the position snaps to the rasteriser's sub-pixel grid (8 bits), the point is the 1x1 square around it, and a
// pixel is produced when its centre lies in that square, lower edge included, upper edge not: floor(u), except that a point within 1/512 px above a pixel edge belongs to
// the pixel below the edge (measured on the reference's index_map shaders: oracle/orc_map.c point_pixel).  -1: left of / above the image.
__device__ __forceinline__ int point_pixel(float u) { return (int)floorf((rintf(u * 256.0f) - 1.0f) / 256.0f); }

// ifx_expf on two arguments at once with the packed f32 instructions of CDNA3 / CDNA4 (v_pk_mul_f32, v_pk_add_f32: two IEEE operations per lane and issue slot, no fused
// multiply-add).  Operation for operation the scalar routine above -- each component's result is bit-identical to ifx_expf of that component -- for arguments that are
// finite and <= 0 (the bilateral weights); what the scalar routine's first two branches handle is folded into the final select.
typedef float ifx_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ifx_v2f ifx_expf2_nonpos(ifx_v2f x)
{
    ifx_v2f n;
    n.x = rintf(x.x * 1.44269504088896341f); n.y = rintf(x.y * 1.44269504088896341f);
    ifx_v2f r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    ifx_v2f p = (ifx_v2f)(1.0f / 5040.0f);
    p = p * r + 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    ifx_v2f s;
    s.x = __uint_as_float((unsigned int)(((int)n.x + 127) & 0xFF) << 23); s.y = __uint_as_float((unsigned int)(((int)n.y + 127) & 0xFF) << 23);   // (& 0xFF: an argument below -87 is selected away below)
    ifx_v2f e = p * s;
    e.x = (x.x > -87.0f) ? e.x : 0.0f;
    e.y = (x.y > -87.0f) ? e.y : 0.0f;
    return e;
}

// color.glsl:19-34
__device__ inline float encode_color(float r, float g, float b)
{
    int rgb = (int)roundf(r * 255.0f);
    rgb = (rgb << 8) + (int)roundf(g * 255.0f);
    rgb = (rgb << 8) + (int)roundf(b * 255.0f);
    return (float)rgb;
}
__device__ inline void decode_color(float c, float* o)
{
    int ic = f2i_rz(c);
    o[0] = (float)(ic >> 16 & 0xFF) / 255.0f;
    o[1] = (float)(ic >> 8 & 0xFF) / 255.0f;
    o[2] = (float)(ic & 0xFF) / 255.0f;
}

// vote counter packing, IF/Core/InstanceFusionCuda.cu:22-39 (short arguments, RNE int->float, trunc back)
__device__ inline float vote_encode(int a, int b)
{
    short sa = (short)a, sb = (short)b;
    int info = (int)((unsigned)(int)sa << 16) + (int)sb;
    return (float)info;
}
__device__ inline void vote_decode(float f, int& a, int& b)
{
    int v = f2i_rz(f);
    a = (short)((v >> 16) & 0xFFFF);
    b = (short)(v & 0xFFFF);
}

// ---- wave64 / block reductions ------------------------------------------------------------
// Full-wave sum with DPP (one VALU instruction per step, no LDS crossbar): quad swaps, row mirrors,
// then the gfx9 row broadcasts.  The total lands in lane 63 (wave_sum_last); wave_sum() moves it to
// every lane's view of lane 0 for the callers that want it there.  The first version used six
// __shfl_down steps (ds_bpermute): 29 values x 6 steps cost 13k cycles per block (profiles r01).
#define IFX_DPP_ADD(v, ctrl, rmask) ((v) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xF, true)))
__device__ __forceinline__ float wave_sum_last(float v)
{
    v = IFX_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    v = IFX_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    v = IFX_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
    v = IFX_DPP_ADD(v, 0x140, 0xF);   // row_mirror: every lane holds its 16-lane row sum
    v = IFX_DPP_ADD(v, 0x142, 0xA);   // row_bcast15 into rows 1 and 3
    v = IFX_DPP_ADD(v, 0x143, 0xC);   // row_bcast31 into rows 2 and 3: lane 63 holds the wave sum
    return v;
}
__device__ inline float wave_sum(float v)
{
    v = wave_sum_last(v);
    return __shfl(v, 63, 64);
}
__device__ inline double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---- exact, order-independent sums of the tracker's normal equations (DESIGN.md "Arithmetic contract").
// Every f32 product that enters one of the 29 (11) sums is first rounded to a fixed grid 2^g (g per entry class, chosen so
// that a term spans at most 32 bits above the grid for the magnitudes the tracker sees) and the sums run in f64: all partial
// sums are integers in units of the grid and stay below 2^53 of them, so every addition is exact and the total does not
// depend on the order -- thread, wave, block, atomic arrival or OpenMP chunk.  oracle/orc_math.h holds the same tables; the
// HIP path and the CPU oracle therefore produce bit-identical sums (the reference's own f32 tree depends on a per-GPU launch
// configuration, EF/Utils/GPUConfig.h:53-137, so there is no single reference order to reproduce).
__host__ __device__ constexpr double ifx_pow2(int e)
{
    double r = 1.0;
    for (int i = 0; i < (e < 0 ? -e : e); i++) r = e < 0 ? r * 0.5 : r * 2.0;
    return r;
}
// (t + M) - M rounds t to a multiple of 2^g (round to nearest even) while |t| < 2^(51 + g)
__host__ __device__ constexpr double ifx_magic(int g) { return 1.5 * ifx_pow2(52 + g); }
__host__ __device__ inline double ifx_quant(float p, double M)
{
    double t = (double)p;
    t = t + M;
    return t - M;
}
#define IFX_EXACT_TERM_BITS 32
// binary exponents of the row entries' working range: ICP row = (n', s' x n', r), RGB row = (v0, v1, v2, rotational part, r), SO(3) row = (jr, r)
#define IFX_E_ICP {0, 0, 0, 4, 4, 4, -3}
#define IFX_E_RGB {11, 11, 11, 13, 13, 13, 3}
#define IFX_E_SO3 {16, 16, 16, 8}
#define IFX_SO3_TERM_BITS 38
#define IFX_ACC_REPL 4       // replicas of a global accumulator row (same-address atomics queue up at the memory side)
#define IFX_ACC_STRIDE 32    // doubles per replica row (256 B: a row per cache line pair)

// ---- owner of a surfel in the spatially sharded map: a pure function of the position it was created at (or uploaded
// with): 8 cm voxel -> 30-bit Morton code -> mod n_ranks, so that neighbouring voxels land on different GPUs (load balance) and
// every rank can tell who owns a new surfel without asking.  instancefusion_amd/dist.py owner_of is the same function.
__host__ __device__ inline unsigned int ifx_part1by2_10(unsigned int v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__host__ __device__ inline int ifx_owner_of_point(float x, float y, float z, int n_ranks)
{
    if (n_ranks <= 1) return 0;
    const int vx = (int)floorf(x / 0.08f) + 512, vy = (int)floorf(y / 0.08f) + 512, vz = (int)floorf(z / 0.08f) + 512;
    const unsigned int code = ifx_part1by2_10((unsigned int)vx) | (ifx_part1by2_10((unsigned int)vy) << 1) | (ifx_part1by2_10((unsigned int)vz) << 2);
    return (int)(code % (unsigned int)n_ranks);
}

// order-preserving map float -> uint for atomicMin depth keys (depths are >= 0 here but be general)
__device__ inline unsigned int depth_bits(float z)
{
    unsigned int u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline unsigned long long make_key(float z, unsigned int id)
{
    return ((unsigned long long)depth_bits(z) << 32) | (unsigned long long)id;
}
#define IFX_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
