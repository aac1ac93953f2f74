// atomic_scope.hip -- what bounds the rasteriser's 64-bit atomicMin traffic on MI355X?
// N ops into a key image of P pixels (8 B each), per variant: scope (agent = device-coherent, executed at the memory side; workgroup = may
// execute in the issuing XCD's L2), address pattern (random over the image; random inside the XCD's own eighth; runs of 3 adjacent pixels).
// Timing only: the workgroup-scope variants over the whole image are not coherent across XCDs (results meaningless), the partitioned one is.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned int xcc_id() { unsigned int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }
__device__ __forceinline__ unsigned int hash(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SCOPE, int MODE, int WIDTH64>
__global__ __launch_bounds__(256) void k_atomics(unsigned long long* __restrict__ img, int P, int n_ops, unsigned int seed)
{
    const unsigned int xcc = xcc_id();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += blockDim.x * gridDim.x) {
        unsigned int hsh = hash(i * 2654435761u + seed);
        unsigned int pix;
        if (MODE == 0) pix = hsh % (unsigned)P;                                              // anywhere
        else if (MODE == 1) pix = (xcc & 7) * (P / 8) + hsh % (unsigned)(P / 8);             // inside this XCD's eighth of the image
        else pix = (hash((i / 3) * 2654435761u + seed) % (unsigned)(P - 3)) + (i % 3);       // runs of 3 adjacent pixels in adjacent lanes
        const unsigned long long key = ((unsigned long long)(hash(hsh) | 1u) << 32) | (unsigned)i;
        if (WIDTH64) {
            if (SCOPE == 0) __hip_atomic_fetch_min(&img[pix], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_min(&img[pix], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            unsigned int* p32 = reinterpret_cast<unsigned int*>(img) + pix;
            if (SCOPE == 0) __hip_atomic_fetch_min(p32, (unsigned int)(key >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_min(p32, (unsigned int)(key >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
__global__ void k_fill(unsigned long long* img, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) img[i] = ~0ull; }
__global__ void k_xcc(unsigned int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE, int MODE, int W64>
static float run(unsigned long long* img, int P, int n_ops, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(k_fill, dim3((P + 255) / 256), dim3(256), 0, 0, img, P);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_atomics<SCOPE, MODE, W64>), dim3(blocks), dim3(256), 0, 0, img, P, n_ops, 1234u + rep);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    return best * 1000.f;
}
int main()
{
    const int P = 640 * 480, n_ops = 5200000;
    unsigned long long* img;
    CHECK(hipMalloc(&img, (size_t)P * 8));
    unsigned int* d_x; CHECK(hipMalloc(&d_x, 64 * 4));
    hipLaunchKernelGGL(k_xcc, dim3(16), dim3(64), 0, 0, d_x);
    unsigned int hx[16]; CHECK(hipMemcpy(hx, d_x, 64, hipMemcpyDeviceToHost));
    printf("XCC id of blocks 0..15:"); for (int i = 0; i < 16; i++) printf(" %u", hx[i]); printf("\n");
    for (int blocks : {1024, 2048}) {
        printf("blocks %d, %d ops into %d pixels (us per launch, best of 5)\n", blocks, n_ops, P);
        printf("  u64 agent     random        %8.1f\n", run<0, 0, 1>(img, P, n_ops, blocks));
        printf("  u64 workgroup random        %8.1f   (not coherent across XCDs: timing only)\n", run<1, 0, 1>(img, P, n_ops, blocks));
        printf("  u64 agent     own eighth    %8.1f\n", run<0, 1, 1>(img, P, n_ops, blocks));
        printf("  u64 workgroup own eighth    %8.1f   (XCD-partitioned: coherent)\n", run<1, 1, 1>(img, P, n_ops, blocks));
        printf("  u64 agent     runs of 3     %8.1f\n", run<0, 2, 1>(img, P, n_ops, blocks));
        printf("  u64 workgroup runs of 3     %8.1f\n", run<1, 2, 1>(img, P, n_ops, blocks));
        printf("  u32 agent     random        %8.1f\n", run<0, 0, 0>(img, P, n_ops, blocks));
        printf("  u32 workgroup own eighth    %8.1f\n", run<1, 1, 0>(img, P, n_ops, blocks));
    }
    return 0;
}
