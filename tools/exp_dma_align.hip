// exp_dma_align.hip — does global_load_lds_dwordx4 accept a source that is only 4- or 8-byte aligned?  (not product code)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_dma_align.hip -o tools/exp_dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int off) {
    __shared__ f32x4 buf[64];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 160 * lane + off),
                                     (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    const f32x4 v = buf[lane];
    for (int i = 0; i < 4; ++i) out[4 * lane + i] = v[i];
}
int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 160 * 4 + 64); hipMalloc(&out, 256 * 4);
    float h[64 * 160 + 16];
    for (int i = 0; i < 64 * 160 + 16; ++i) h[i] = (float)i;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    for (int off = 0; off < 8; ++off) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, off);
        hipError_t e = hipDeviceSynchronize();
        float o[256];
        hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (o[4 * l + i] != (float)(160 * l + off + i)) ++bad;
        printf("offset %d floats (%2d bytes): %s, %d wrong values (lane 1: %.0f %.0f %.0f %.0f, want %d ..)\n", off, 4 * off, hipGetErrorString(e), bad,
               o[4], o[5], o[6], o[7], 160 + off);
    }
    return 0;
}
