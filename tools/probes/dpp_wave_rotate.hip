#include <hip/hip_runtime.h>
__global__ void k(int *out)
{
    int v = threadIdx.x * 10;
    int r = __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false);   // wave_rol:1
    int s = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
    out[threadIdx.x] = r;
    out[64 + threadIdx.x] = s;
}
int main()
{
    int *d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("rol1: %d %d ... %d %d | shl1: %d %d ... %d %d\n", h[0], h[1], h[62], h[63], h[64], h[65], h[126], h[127]);
    return 0;
}
