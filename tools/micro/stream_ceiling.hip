// tools/micro/stream_ceiling.hip -- how fast can ONE launch stream the 24 B/slot (float4 + float2) of the surfel store at the BASELINE size?
// The ceiling the cull kernels of ifx_map.hip are measured against.   hipcc -O3 --offload-arch=gfx950 stream_ceiling.hip -o stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// (a) chunked like k_cull_raster: one 4096-slot chunk per block iteration, ROUNDS loads per thread in flight
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_chunked(const float4* __restrict__ pc, const float2* __restrict__ tm, int n, unsigned int* __restrict__ out)
{
    unsigned int acc = 0;
    for (int chunk = blockIdx.x; chunk * 256 * ROUNDS < n; chunk += gridDim.x) {
        float4 p[ROUNDS];
        float2 t[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const int i = chunk * 256 * ROUNDS + r * 256 + threadIdx.x;
            p[r] = i < n ? pc[i] : make_float4(0, 0, 0, 0);
            t[r] = i < n ? tm[i] : make_float2(0, 0);
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) acc += (p[r].w > 10.f) + (t[r].y > 5.f) + (p[r].z > 0.f);
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;   // never: keeps the loads alive
}
// (b) plain grid-stride, one slot per thread per iteration
__global__ __launch_bounds__(256) void k_stride(const float4* __restrict__ pc, const float2* __restrict__ tm, int n, unsigned int* __restrict__ out)
{
    unsigned int acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const float4 p = pc[i];
        const float2 t = tm[i];
        acc += (p.w > 10.f) + (t.y > 5.f) + (p.z > 0.f);
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}

int main()
{
    const int n = 5178252;   // slots of the BASELINE run
    float4* pc; float2* tm; unsigned int* out;
    CK(hipMalloc(&pc, (size_t)n * 16)); CK(hipMalloc(&tm, (size_t)n * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemset(pc, 0, (size_t)n * 16)); CK(hipMemset(tm, 0, (size_t)n * 8));
    // something large in between so that nothing of the store stays in the 256 MB Infinity Cache between launches
    float* junk; const size_t junk_bytes = 1ull << 30;
    CK(hipMalloc(&junk, junk_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int flush_mode = 1;   // 0 none (store partly resident in the 256 MB Infinity Cache), 1 read 1 GB of something else, 2 write 1 GB
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0; const int reps = 20;
        for (int k = 0; k < reps; k++) {
            if (flush_mode == 2) hipMemsetAsync(junk, k, junk_bytes, 0);
            if (flush_mode == 1) hipLaunchKernelGGL(k_stride, dim3(16384), dim3(256), 0, 0, (const float4*)junk, (const float2*)(junk + (junk_bytes / 8)), (int)(junk_bytes / 32), out);
            hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
            sum += ms;
        }
        printf("%-28s avg %.1f us  best %.1f us  -> %.2f TB/s (avg)\n", name, 1000 * sum / reps, 1000 * best, (double)n * 24 / (sum / reps * 1e-3) / 1e12);
        return 0;
    };
    for (flush_mode = 0; flush_mode < 3; flush_mode++) {
    printf("--- flush mode %d (0 none, 1 read 1 GB before, 2 write 1 GB before)\n", flush_mode);
    run("chunked R16 2048 blocks", [&] { hipLaunchKernelGGL(k_chunked<16>, dim3(2048), dim3(256), 0, 0, pc, tm, n, out); });
    run("chunked R8 4096 blocks", [&] { hipLaunchKernelGGL(k_chunked<8>, dim3(4096), dim3(256), 0, 0, pc, tm, n, out); });
    run("chunked R4 8192 blocks", [&] { hipLaunchKernelGGL(k_chunked<4>, dim3(8192), dim3(256), 0, 0, pc, tm, n, out); });
    run("chunked R8 1024 blocks", [&] { hipLaunchKernelGGL(k_chunked<8>, dim3(1024), dim3(256), 0, 0, pc, tm, n, out); });
    run("chunked R4 2048 blocks", [&] { hipLaunchKernelGGL(k_chunked<4>, dim3(2048), dim3(256), 0, 0, pc, tm, n, out); });
    run("stride 2048 blocks", [&] { hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, pc, tm, n, out); });
    run("stride 8192 blocks", [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, pc, tm, n, out); });
    run("stride 20228 blocks (1/thread)", [&] { hipLaunchKernelGGL(k_stride, dim3((n + 255) / 256), dim3(256), 0, 0, pc, tm, n, out); });
    }
    // size sweep, cold (read flush): fixed cost + bytes / bandwidth
    flush_mode = 1;
    for (int m = n / 8; m <= n; m *= 2) {
        char nm[64]; snprintf(nm, sizeof(nm), "stride, %d slots (%.0f MB)", m, m * 24e-6);
        run(nm, [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, pc, tm, m, out); });
    }
    return 0;
}
