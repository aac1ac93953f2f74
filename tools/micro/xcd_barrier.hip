// tools/micro/xcd_barrier.hip -- what would one meeting of a PERSISTENT Gauss-Newton level cost if its workgroups all sat on ONE XCD, against a meeting across the eight XCDs
// and against the dependent kernel boundary the two-launch form pays?  ("tracker coarse levels in one XCD-local launch ... or a measured kill".)
// A meeting as the tracker would need it: every workgroup adds its 29 partial sums (f64, agent-scope atomics: exact sums commute) and then one arrival; everybody polls the
// arrival counter (relaxed agent-scope load) and reads the 29 totals.  No fence: nothing but atomically accumulated words is exchanged (each workgroup would keep its own pixels).
// Workgroups place themselves: each reads HW_REG_XCC_ID; in the one-XCD variant the FIRST arrival claims its XCD and only workgroups of that XCD take part (rank by ticket),
// the others leave -- no assumption about the dispatcher's placement (MI355X_MICROARCH.md, "Workgroup dispatch").      hipcc -O3 --offload-arch=gfx950 xcd_barrier.hip -o xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl { int claim; unsigned int ticket; unsigned int pad[14]; unsigned int arrive[16]; double acc[2][32]; long long cycles; int members; };

__global__ __launch_bounds__(256) void k_meet(Ctl* c, int want, int one_xcd, int meetings, int spin_limit)
{
    __shared__ int s_rank;
    if (threadIdx.x == 0) {
        int rank = -1;
        unsigned int xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xF;
        bool in = true;
        if (one_xcd) {
            int expected = -1;
            __hip_atomic_compare_exchange_strong(&c->claim, &expected, (int)xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            in = __hip_atomic_load(&c->claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)xcc;
        }
        if (in) { const unsigned int t = __hip_atomic_fetch_add(&c->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rank = t < (unsigned int)want ? (int)t : -1; }
        s_rank = rank;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    // (every member must be resident: `want` is at most the number of workgroup slots of one XCD, and the launch over-subscribes by 8x so that one XCD alone can fill it)
    long long t0 = 0;
    for (int m = 0; m < meetings; m++) {
        const int par = m & 1;
        if (m == 1 && rank == 0 && threadIdx.x == 0) t0 = wall_clock64();
        if (threadIdx.x < 29) __hip_atomic_fetch_add(&c->acc[par][threadIdx.x], 1.0 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&c->arrive[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = (unsigned int)want * (unsigned int)(m + 1);
            int spins = 0;
            while (__hip_atomic_load(&c->arrive[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < spin_limit) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        double v = 0;
        if (threadIdx.x < 29) v = __hip_atomic_load(&c->acc[par][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < 0) c->pad[0] = 1;   // (keeps the load)
    }
    if (rank == 0 && threadIdx.x == 0) { c->cycles = wall_clock64() - t0; c->members = (int)__hip_atomic_load(&c->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

int main()
{
    Ctl* d; CK(hipMalloc(&d, sizeof(Ctl)));
    int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const int meetings = 2001;
    printf("meeting = 29 f64 atomic adds + 1 arrival per workgroup, poll, read 29 totals; %d meetings, wall clock %d kHz\n", meetings - 1, rate_khz);
    for (int one_xcd : {1, 0})
        for (int want : {8, 16, 32, 64, 75, 128}) {
            Ctl h; memset(&h, 0, sizeof(h)); h.claim = -1;
            CK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
            const int grid = one_xcd ? want * 8 + 64 : want;   // round-robin placement gives one XCD an eighth of the grid; + slack
            hipLaunchKernelGGL(k_meet, dim3(grid), dim3(256), 0, 0, d, want, one_xcd, meetings, 1 << 22);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("%-9s %4d workgroups (tickets drawn %4d of a grid of %4d): %.2f us per meeting\n", one_xcd ? "one XCD" : "all XCDs", want, h.members, grid,
                   (double)h.cycles / rate_khz * 1e3 / (meetings - 1));
        }
    return 0;
}
