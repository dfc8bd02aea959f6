// exp_store_patterns.hip — what a CU's dwordx4 stores cost by lane -> address pattern (round 6: the embedding epilogue of
// nplda_fwd_mid.h issues ten 1 KB stores per wave in ~3 500 cycles; which part of that is the pattern?).
// One 256-thread block per CU; every wave writes `n` dwordx4 stores of 64 lanes (1 KB each) into its own rows of a
// (rows, 160)-float table, in a burst, then waits for them (s_waitcnt vmcnt(0)); cycles per burst by s_memtime of wave 0.
//   pattern 0: one instruction = 1 KB contiguous (1.6 rows)
//   pattern 1: one instruction = 16 rows x 64 B, lane 16 g + j -> row j, bytes 16 g .. (the MFMA accumulator layout)
//   pattern 2: one instruction = 16 rows x 64 B, lane L -> row L >> 2, bytes 16 (L & 3) .. (after a ds_bpermute)
//   pattern 3: one instruction = 8 rows x 128 B (whole lines)
//   pattern 4: one instruction = 4 rows x 256 B
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_store_patterns.hip -o tools/exp_store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool LOAD = false>
__global__ __launch_bounds__(256, 1) void store_kernel(float* out, long long rows_per_block, int groups, int gap_sleep,
                                                         unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const long long ldz = 160;
    float* base = out + (long long)blockIdx.x * rows_per_block * ldz;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    unsigned long long t_acc = 0;
    for (int grp = 0; grp < groups; ++grp) {
        float* tile = base + (long long)grp * 64 * ldz;  // 64 rows x 160 floats = 40 KB per group, 10 KB per wave
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        f32x4 ld[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            float* p;
            if (PAT == 0) {  // wave w: rows 16 w .. 16 w + 15 = 2 560 floats contiguous; instruction i: floats 256 i ..
                p = tile + (16 * wave) * ldz + 256 * i + 4 * lane;
            } else if (PAT == 1) {  // wave w: column block b = (w + 4 i) % 10 ... any 10 (row group, block) units of the tile
                // the kernel's units: wave w holds blocks w, w + 4 of all four row groups and two left-over units
                const int rg = i < 8 ? (i & 3) : 2 * (wave >> 1) + (i - 8), b = i < 8 ? wave + 4 * (i >> 2) : 8 + (wave & 1);
                p = tile + (16 * rg + j) * ldz + 16 * b + 4 * g;
            } else if (PAT == 2) {
                const int rg = i < 8 ? (i & 3) : 2 * (wave >> 1) + (i - 8), b = i < 8 ? wave + 4 * (i >> 2) : 8 + (wave & 1);
                p = tile + (16 * rg + (lane >> 2)) * ldz + 16 * b + 4 * (lane & 3);
            } else if (PAT == 3) {  // 8 rows x 128 B: unit (row octet, 32-float column block): 8 x 5 = 40 units, 10 per wave
                const int u = wave * 10 + i;
                p = tile + (8 * (u / 5) + (lane >> 3)) * ldz + 32 * (u % 5) + 4 * (lane & 7);
            } else {  // 4 rows x 256 B: 64 floats; 160 = 2.5 x 64 -> treat the tile as 16 x 640 floats
                const int u = wave * 10 + i;
                p = tile + (long long)(4 * (u / 10) + (lane >> 4)) * 640 + 64 * (u % 10) + 4 * (lane & 15);
            }
            if (LOAD) ld[i] = *reinterpret_cast<const f32x4*>(p);
            else *reinterpret_cast<f32x4*>(p) = v;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        if (LOAD) {
#pragma unroll
            for (int i = 0; i < 10; ++i) v += ld[i];
        }
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {
            cyc[2 * grp] = t1 - t0;
            cyc[2 * grp + 1] = t2 - t0;
        }
        t_acc += t2 - t0;
        for (int s = 0; s < gap_sleep; ++s) __builtin_amdgcn_s_sleep(16);  // ~1k cycles each: the next group's compute
    }
    if (t_acc == 1 || v[0] == 1.2345f) out[0] = v[1];
}

int main(int argc, char** argv) {
    const int groups = 12;
    const int gap = argc > 1 ? atoi(argv[1]) : 20;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    const long long rpb = 64LL * groups;
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)grid * rpb * 160 * 4));
    CK(hipMalloc(&cyc, 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"1 KB contiguous", "16 rows x 64 B, accumulator lanes", "16 rows x 64 B, 4 lanes per piece", "8 rows x 128 B",
                           "4 rows x 256 B"};
    const bool load = argc > 3;  // any third argument: the same patterns as LOADS (each group's 40 KB read once: HBM / L2 misses)
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 5; ++pat) {
            auto go = [&]() {
                if (load) {
                    switch (pat) {
                        case 0: store_kernel<0, true><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                        case 1: store_kernel<1, true><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                        case 2: store_kernel<2, true><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                        case 3: store_kernel<3, true><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                        default: store_kernel<4, true><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                    }
                    return;
                }
                switch (pat) {
                    case 0: store_kernel<0><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                    case 1: store_kernel<1><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                    case 2: store_kernel<2><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                    case 3: store_kernel<3><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                    default: store_kernel<4><<<grid, 256>>>(out, rpb, groups, gap, cyc); break;
                }
            };
            go(); go();
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 5; ++k) go();
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[64];
            CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
            unsigned long long si = 0, sd = 0;
            for (int gI = 2; gI < groups; ++gI) { si += h[2 * gI]; sd += h[2 * gI + 1]; }
            printf("%s pattern %d (%-34s): issue %6llu cyc, issue + drain %6llu cyc per 40 KB group of a CU (%.1f B/clk)   launch %.1f us\n",
                   load ? "LOAD " : "STORE", pat, names[pat], si / (groups - 2), sd / (groups - 2), 40960.0 / (sd / (double)(groups - 2)), ms * 200.0);
        }
    return 0;
}
