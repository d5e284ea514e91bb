// residency_census.hip -- how many single-wave workgroups are co-resident on the chip, by the
// registers they hold, against what hipOccupancyMaxActiveBlocksPerMultiprocessor answers.
// Every workgroup checks in and waits (bounded) until all of the grid has: a grid that is not
// co-resident never completes the roll call.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/residency_census.hip -o /tmp/census && /tmp/census
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VREG>
__global__ __launch_bounds__(64) void census(unsigned *count, unsigned *seen_all, unsigned n)
{
    if constexpr (VREG == 64)  asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (VREG == 96)  asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if constexpr (VREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (VREG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if constexpr (VREG == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool all = false;
        while (__builtin_amdgcn_s_memrealtime() - t0 < 2000000ull) {            // 20 ms
            if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) {
                all = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (all)
            __hip_atomic_fetch_add(seen_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int VREG>
void run(unsigned *d)
{
    int api = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census<VREG>, 64, 0);
    printf("vgprs %3d: API %2d per CU;", VREG, api);
    for (unsigned per_cu : {4u, 8u, 12u, 16u, 20u, 24u, 32u}) {
        const unsigned n = per_cu * 256u;
        (void)hipMemset(d, 0, 8);
        census<VREG><<<n, 64>>>(d, d + 1, n);
        (void)hipDeviceSynchronize();
        unsigned h[2];
        (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("  %u/CU: %s", per_cu, h[1] == n ? "resident" : "NOT");
    }
    printf("\n");
}

int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 8);
    run<64>(d);
    run<96>(d);
    run<128>(d);
    run<168>(d);
    run<256>(d);
    return 0;
}
