// tools/micro/launch_gap.hip -- what does a chain of DEPENDENT small kernels cost per link on this GPU, launched one by one into a stream and as one
// hipGraph?  The tracker of ifx_track.hip is such a chain (19 Gauss-Newton iterations x 2 launches per frame): this is the number
// that says whether a graph of a pyramid level would shorten it.          hipcc -O3 --offload-arch=gfx950 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one link: `blocks` workgroups each reduce a slice of `in` (work comparable to one tracker reduction at 640x480), the result feeds the next link
__global__ __launch_bounds__(256) void k_link(const float* __restrict__ in, float* __restrict__ out, int n, const float* __restrict__ prev, float* __restrict__ next)
{
    float acc = prev[0] * 1e-9f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += in[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;   // per-wave partials (no shared atomic)
    if (blockIdx.x == 0 && threadIdx.x == 0) next[0] = acc;
}

int main()
{
    const int n_full = 640 * 480 * 4, links = 38, reps = 200;
    float *in, *out, *chain;
    CK(hipMalloc(&in, n_full * 4)); CK(hipMalloc(&out, 4096 * 16)); CK(hipMalloc(&chain, (links + 1) * 4));
    CK(hipMemset(in, 0, n_full * 4)); CK(hipMemset(out, 0, 4096 * 16)); CK(hipMemset(chain, 0, (links + 1) * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int blocks : {1, 64, 304, 1200}) {   // 1 block over 4 floats = an empty link: the bare cost of a dependent launch
        const int n = blocks == 1 ? 4 : n_full;
        // (1) one by one
        for (int warm = 0; warm < 2; warm++) {
            CK(hipEventRecord(a, s));
            for (int r = 0; r < reps; r++)
                for (int l = 0; l < links; l++) hipLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, s, in, out, n, chain + l, chain + l + 1);
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
        }
        float ms_stream = 0;
        CK(hipEventElapsedTime(&ms_stream, a, b));
        // (2) the same chain captured once, launched `reps` times
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int l = 0; l < links; l++) hipLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, s, in, out, n, chain + l, chain + l + 1);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms_graph = 0;
        for (int warm = 0; warm < 2; warm++) {
            CK(hipEventRecord(a, s));
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
        }
        CK(hipEventElapsedTime(&ms_graph, a, b));
        printf("blocks %4d: stream %.2f us per link, graph %.2f us per link (%d links x %d repetitions)\n", blocks, ms_stream * 1000.f / (links * reps),
               ms_graph * 1000.f / (links * reps), links, reps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
