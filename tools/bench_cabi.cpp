// bench_cabi.cpp — the headline measurement through the C ABI only: no Python, no torch, no kernel headers.
// Links libnplda_hip.so and libamdhip64, fills device buffers with a hash-based generator, packs a random model and times
// nplda_score_pairs_f32 with HIP events on the launch stream.  It is what a C / C++ / Go-cgo consumer of the library
// would do; bench.py is the driver-facing twin (same kernel, torch only for buffers).
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/bench_cabi.cpp -Iinclude -Lneuralplda_amd -lnplda_hip \
//              -Wl,-rpath,$PWD/neuralplda_amd -o tools/bench_cabi
// run:   tools/bench_cabi [log2_pairs=20] [D=150] [steps=20]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nplda_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define NK(x) do { int r_ = (x); if (r_ != NPLDA_OK) { fprintf(stderr, "nplda error %d (%s) at line %d\n", r_, nplda_strerror(r_), __LINE__); return 1; } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        float u = 0.f;
        for (int k = 0; k < 4; ++k) u += (float)((h >> (16 * k)) & 0xffff) / 65536.0f;
        p[i] = (u - 2.0f) * 1.7320508f * scale;
    }
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20, D = argc > 2 ? atoi(argv[2]) : 150, steps = argc > 3 ? atoi(argv[3]) : 20;
    const int D0 = 512;
    const long long B = 1ll << lg;
    float *x1, *x2, *s, *W1, *b1, *W2, *b2, *Ps, *Q;
    void* packed;
    const size_t pbytes = nplda_packed_bytes(D0, D, D);
    if (pbytes == 0) { fprintf(stderr, "unsupported model\n"); return 1; }
    CK(hipMalloc(&x1, B * D0 * 4)); CK(hipMalloc(&x2, B * D0 * 4)); CK(hipMalloc(&s, B * 4)); CK(hipMalloc(&packed, pbytes));
    CK(hipMalloc(&W1, (size_t)D * D0 * 4)); CK(hipMalloc(&b1, D * 4)); CK(hipMalloc(&W2, (size_t)D * D * 4));
    CK(hipMalloc(&b2, D * 4)); CK(hipMalloc(&Ps, D * 4)); CK(hipMalloc(&Q, D * 4));
    fill<<<4096, 256>>>(x1, (size_t)B * D0, 1, 1.f); fill<<<4096, 256>>>(x2, (size_t)B * D0, 2, 1.f);
    fill<<<64, 256>>>(W1, (size_t)D * D0, 3, 0.05f); fill<<<1, 256>>>(b1, D, 4, 0.1f);
    fill<<<64, 256>>>(W2, (size_t)D * D, 5, 0.1f); fill<<<1, 256>>>(b2, D, 6, 0.1f);
    fill<<<1, 256>>>(Ps, D, 7, 0.5f); fill<<<1, 256>>>(Q, D, 8, 0.5f);
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    NK(nplda_pack_params_f32(W1, b1, W2, b2, Ps, Q, D0, D, D, packed, pbytes, st));
    for (int i = 0; i < 3; ++i) NK(nplda_score_pairs_f32(x1, x2, B, D0, packed, D0, D, D, s, st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < steps; ++i) NK(nplda_score_pairs_f32(x1, x2, B, D0, packed, D0, D, D, s, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= steps;
    std::vector<float> hs(1024);
    CK(hipMemcpy(hs.data(), s, hs.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0;
    for (float v : hs) cs += v;
    const double flops = 2.0 * (2.0 * D0 * D + 2.0 * D * D) + 8.0 * D;
    printf("{\"metric\": \"scored trial-pairs/sec (512-d xvec)\", \"value\": %.6e, \"unit\": \"pairs/s\", \"pairs\": %lld, \"D\": %d, "
           "\"kernel_ms\": %.4f, \"TFLOPs_algorithmic\": %.2f, \"frac_of_157.3\": %.4f, \"checksum_first_1024\": %.6f, "
           "\"via\": \"C ABI (libnplda_hip.so), no torch\"}\n",
           B / (ms * 1e-3), B, D, ms, B * flops / (ms * 1e-3) / 1e12, B * flops / (ms * 1e-3) / 1e12 / 157.3, cs);
    return 0;
}
