// tools/micro/pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the map passes.
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports exactly half of the bytes of a wide (16 B per lane) coalesced streaming read; "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel here reads a KNOWN number of
// bytes from buffers far larger than the 256 MiB Infinity Cache; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (tools/pmc_calib.sh) and
// divide the printed byte counts by the counter: that factor is what tools/pmc_summary.py applies per pattern.
//   k_cal_stream16   float4 per lane, coalesced                  (the guide's case: factor 2)
//   k_cal_stream8    float2 per lane, coalesced                  (times[])
//   k_cal_stream4    one dword per lane, coalesced               (view-list entries, id images)
//   k_cal_stream24   float4 + float2 of the same slot            (k_cull_frame's pattern)
//   k_cal_gather16   one 16-B record per lane at an ascending, ~9 % dense index list (the view lists: 0.45 M entries of 5.2 M slots)
//   k_cal_gather16r  one 16-B record per lane at random indices (every lane its own line)
// hipcc -O3 --offload-arch=gfx950 pmc_calib.hip -o pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_cal_stream16(const float4* __restrict__ a, size_t n, unsigned int* out)
{
    unsigned int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x) { const float4 v = a[i]; acc += (v.x > 1.f) + (v.w > 2.f); }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_stream8(const float2* __restrict__ a, size_t n, unsigned int* out)
{
    unsigned int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x) { const float2 v = a[i]; acc += (v.x > 1.f) + (v.y > 2.f); }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_stream4(const float* __restrict__ a, size_t n, unsigned int* out)
{
    unsigned int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x) acc += (a[i] > 1.f);
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_stream24(const float4* __restrict__ a, const float2* __restrict__ b, size_t n, unsigned int* out)
{
    unsigned int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x) { const float4 v = a[i]; const float2 t = b[i]; acc += (v.x > 1.f) + (t.y > 2.f); }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_gather16(const float4* __restrict__ a, const unsigned int* __restrict__ idx, size_t m, unsigned int* out)
{
    unsigned int acc = 0;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < m; t += (size_t)blockDim.x * gridDim.x) { const float4 v = a[idx[t]]; acc += (v.x > 1.f) + (v.w > 2.f); }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_gather16r(const float4* __restrict__ a, const unsigned int* __restrict__ idx, size_t m, unsigned int* out)
{
    unsigned int acc = 0;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < m; t += (size_t)blockDim.x * gridDim.x) { const float4 v = a[idx[t]]; acc += (v.x > 1.f) + (v.w > 2.f); }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}

int main()
{
    const size_t n = 48u << 20;   // 48 M records: 768 MB of float4, 384 MB of float2, 192 MB of floats -- each beyond the 256 MiB Infinity Cache
    float4* a; float2* b; float* c; unsigned int *idx, *idxr, *out;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 4 * 2)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 8));
    // ascending ~9 % dense list (one of every ~11.5 slots, jittered) and a random list of the same length
    std::vector<unsigned int> h, hr;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < n; i++) if (rnd() % 1000 < 87) h.push_back((unsigned int)i);
    const size_t m = h.size();
    hr.resize(m);
    for (size_t i = 0; i < m; i++) hr[i] = (unsigned int)(rnd() % n);
    CK(hipMalloc(&idx, m * 4)); CK(hipMalloc(&idxr, m * 4));
    CK(hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(idxr, hr.data(), m * 4, hipMemcpyHostToDevice));
    // distinct 64-B / 128-B lines the ascending list touches (what a perfect line-granular fetch would move)
    size_t l64 = 0, l128 = 0; unsigned int p64 = ~0u, p128 = ~0u;
    for (size_t i = 0; i < m; i++) { const unsigned int q64 = h[i] / 4, q128 = h[i] / 8; if (q64 != p64) { l64++; p64 = q64; } if (q128 != p128) { l128++; p128 = q128; } }
    const dim3 g(8192), t(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_cal_stream16, g, t, 0, 0, (const float4*)a, n, out);
        hipLaunchKernelGGL(k_cal_stream8, g, t, 0, 0, (const float2*)b, n, out);
        hipLaunchKernelGGL(k_cal_stream4, g, t, 0, 0, (const float*)c, 2 * n, out);
        hipLaunchKernelGGL(k_cal_stream24, g, t, 0, 0, (const float4*)a, (const float2*)b, n, out);
        hipLaunchKernelGGL(k_cal_gather16, g, t, 0, 0, (const float4*)a, (const unsigned int*)idx, m, out);
        hipLaunchKernelGGL(k_cal_gather16r, g, t, 0, 0, (const float4*)a, (const unsigned int*)idxr, m, out);
    }
    CK(hipDeviceSynchronize());
    printf("{\"k_cal_stream16\": %zu, \"k_cal_stream8\": %zu, \"k_cal_stream4\": %zu, \"k_cal_stream24\": %zu,\n", n * 16, n * 8, n * 8, n * 24);
    printf(" \"k_cal_gather16\": {\"entries\": %zu, \"algorithmic\": %zu, \"lines64\": %zu, \"lines128\": %zu},\n", m, m * 20, l64 * 64 + m * 4, l128 * 128 + m * 4);
    printf(" \"k_cal_gather16r\": {\"entries\": %zu, \"algorithmic\": %zu, \"lines64\": %zu, \"lines128\": %zu}}\n", m, m * 20, m * 64 + m * 4, m * 128 + m * 4);
    return 0;
}
