// Probe (GPU box only): does `buffer_load_dwordx4 ... offen lds` write ZEROS into LDS for lanes whose offset is
// out of the descriptor's range, or leave LDS untouched?  Decides whether the conv DMA can use the hardware range
// check for padding instead of a zero page.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out, int nbytes) {
    __shared__ __attribute__((aligned(1024))) float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -5.f;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x >= 32 && threadIdx.x < 48) voff = 0xFFFFFF00u;  // far out of range
    if (threadIdx.x >= 48) voff = nbytes + (threadIdx.x - 48) * 16;   // just past the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    const int n = 256;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *g, *o;
    hipMalloc(&g, n * 4);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, n * 4);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    for (int lane : {0, 1, 31, 32, 40, 47, 48, 63}) printf("lane %2d -> lds[%3d..] = %g %g %g %g\n", lane, lane * 4, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
    printf("untouched tail lds[300] = %g\n", r[300]);
    return 0;
}
