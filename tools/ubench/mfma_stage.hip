// Micro-benchmark: what limits a stream of v_mfma_f32_32x32x2_f32 fed by selects / LDS reads?
// One wave per SIMD (256 blocks x 256 threads, 100 KB LDS to force 1 block/CU), s_memtime
// around REP "stages" of 16 MFMAs.  Prints median cycles per stage for several variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
constexpr int REP = 256;

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ gin, float* gout, unsigned long long* cyc, int flag) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = gin[i & 1023];
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float b[8];
    for (int j = 0; j < 8; ++j) b[j] = gin[lane + 64 * j];
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = gin[lane + 64 * j + 512];
    const bool ok0 = (lane + flag) & 1, ok1 = (lane + flag) & 2;
    const float* s0 = lds + lane, * s1 = lds + lane + 32;
    float na[8];
    for (int j = 0; j < 8; ++j) na[j] = a[j];
    SB();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    SB();
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
        if (V == 0) {            // pure MFMA, operands live in registers
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
        } else if (V == 1) {     // select (v_cndmask) right before each MFMA pair
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = ok0 ? a[j] : 0.f, a1 = ok1 ? a[4 + j] : 0.f;
                acc[0] = MFMA(a0, b[j], acc[0]); acc[1] = MFMA(a0, b[4 + j], acc[1]);
                acc[2] = MFMA(a1, b[j], acc[2]); acc[3] = MFMA(a1, b[4 + j], acc[3]);
            }
        } else if (V == 2) {     // all 8 selects first, then the 16 MFMAs
            float m[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { m[j] = ok0 ? a[j] : 0.f; m[4 + j] = ok1 ? a[4 + j] : 0.f; }
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(m[j], b[j], acc[0]); acc[1] = MFMA(m[j], b[4 + j], acc[1]);
                acc[2] = MFMA(m[4 + j], b[j], acc[2]); acc[3] = MFMA(m[4 + j], b[4 + j], acc[3]);
            }
        } else if (V == 3) {     // like the kernel: LDS reads for the next stage, selects interleaved
#pragma unroll
            for (int j = 0; j < 4; ++j) { na[j] = s0[(2 * j) * 68 + (rep & 7) * 544]; na[4 + j] = s1[(2 * j) * 68 + (rep & 7) * 544]; }
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = ok0 ? a[j] : 0.f, a1 = ok1 ? a[4 + j] : 0.f;
                acc[0] = MFMA(a0, b[j], acc[0]); acc[1] = MFMA(a0, b[4 + j], acc[1]);
                acc[2] = MFMA(a1, b[j], acc[2]); acc[3] = MFMA(a1, b[4 + j], acc[3]);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = na[j];
        } else if (V == 4) {     // LDS reads for the next stage, NO selects (multiply-free path)
#pragma unroll
            for (int j = 0; j < 4; ++j) { na[j] = s0[(2 * j) * 68 + (rep & 7) * 544]; na[4 + j] = s1[(2 * j) * 68 + (rep & 7) * 544]; }
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = na[j];
        } else if (V == 7) {     // V3 with the loads spread into the MFMA gaps (sched_group_barrier)
#pragma unroll
            for (int j = 0; j < 4; ++j) { na[j] = s0[(2 * j) * 68 + (rep & 7) * 544]; na[4 + j] = s1[(2 * j) * 68 + (rep & 7) * 544]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = ok0 ? a[j] : 0.f, a1 = ok1 ? a[4 + j] : 0.f;
                acc[0] = MFMA(a0, b[j], acc[0]); acc[1] = MFMA(a0, b[4 + j], acc[1]);
                acc[2] = MFMA(a1, b[j], acc[2]); acc[3] = MFMA(a1, b[4 + j], acc[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);     // 2 VALU (selects)
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);     // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = na[j];
        } else if (V == 8) {     // V7 but one DS read per TWO MFMAs (8 reads over 16 MFMAs evenly)
#pragma unroll
            for (int j = 0; j < 4; ++j) { na[j] = s0[(2 * j) * 68 + (rep & 7) * 544]; na[4 + j] = s1[(2 * j) * 68 + (rep & 7) * 544]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = ok0 ? a[j] : 0.f, a1 = ok1 ? a[4 + j] : 0.f;
                acc[0] = MFMA(a0, b[j], acc[0]); acc[1] = MFMA(a0, b[4 + j], acc[1]);
                acc[2] = MFMA(a1, b[j], acc[2]); acc[3] = MFMA(a1, b[4 + j], acc[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = na[j];
        } else if (V == 9) {     // 16 LDS reads per stage (layer2 shape), one per MFMA gap, 2 acc
            float nb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { na[j] = s0[(2 * j) * 20 + (rep & 7) * 320]; nb[j] = s1[(2 * j) * 20 + (rep & 7) * 320 + 4096]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a0 = ok0 ? a[j] : 0.f, a1 = ok0 ? b[j] : 0.f;
                acc[0] = MFMA(a0, b[j], acc[0]); acc[1] = MFMA(a1, b[j], acc[1]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = na[j]; b[j] = nb[j] + b[j] * 0.0f; }
        } else if (V == 10) {    // same 16 reads per stage in a burst (reference for V9)
            float nb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { na[j] = s0[(2 * j) * 20 + (rep & 7) * 320]; nb[j] = s1[(2 * j) * 20 + (rep & 7) * 320 + 4096]; }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a0 = ok0 ? a[j] : 0.f, a1 = ok0 ? b[j] : 0.f;
                acc[0] = MFMA(a0, b[j], acc[0]); acc[1] = MFMA(a1, b[j], acc[1]);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = na[j]; b[j] = nb[j] + b[j] * 0.0f; }
        } else if (V == 11) {    // 2 x ds_read_b128 per stage (8 floats) instead of 8 x ds_read_b32
            const float4 q0 = *reinterpret_cast<const float4*>(lds + lane * 68 + (rep & 7) * 8);
            const float4 q1 = *reinterpret_cast<const float4*>(lds + lane * 68 + (rep & 7) * 8 + 4);
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
            SB();
            a[0] = q0.x; a[1] = q0.y; a[2] = q0.z; a[3] = q0.w; a[4] = q1.x; a[5] = q1.y; a[6] = q1.z; a[7] = q1.w;
        } else if (V == 12) {    // 4 x ds_read_b64 per stage
            float2 q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const float2*>(lds + lane * 66 + (rep & 7) * 8 + 2 * j);
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[2 * j] = q[j].x; a[2 * j + 1] = q[j].y; }
        } else if (V == 13) {    // 2 x global_load_dwordx4 per stage (weights), no LDS
            const float4 q0 = *reinterpret_cast<const float4*>(gin + ((rep * 64 + lane) & 2047) * 4);
            const float4 q1 = *reinterpret_cast<const float4*>(gin + ((rep * 64 + lane + 1024) & 2047) * 4);
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
            SB();
            b[0] = q0.x; b[1] = q0.y; b[2] = q0.z; b[3] = q0.w; b[4] = q1.x; b[5] = q1.y; b[6] = q1.z; b[7] = q1.w;
        } else if (V == 14) {    // 1 x ds_read_b32 per stage (cost of a single read)
            const float q0 = s0[(rep & 7) * 544];
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
            SB();
            a[0] = q0;
        } else if (V == 15) {    // 8 v_mov (register rotation) per stage, no loads
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[j], b[4 + j], acc[1]);
                acc[2] = MFMA(a[4 + j], b[j], acc[2]); acc[3] = MFMA(a[4 + j], b[4 + j], acc[3]);
            }
            SB();
#pragma unroll
            for (int j = 0; j < 8; ++j) { float t = a[j]; asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(t)); }
        } else if (V == 5) {     // 2 accumulators only (layer2 shape): 16 MFMAs alternate acc0/acc1
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[0] = MFMA(a[j], b[j], acc[0]); acc[1] = MFMA(a[(j + 1) & 7], b[j], acc[1]); }
        } else if (V == 6) {     // 16x16x4 would be another option; here: 1 accumulator chain (dependent)
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[0] = MFMA(a[j], b[j], acc[0]); acc[0] = MFMA(a[(j + 1) & 7], b[j], acc[0]); }
        }
    }
    SB();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    gout[blockIdx.x * 256 + threadIdx.x] = s + a[0];
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(const char* name, float* gin, float* gout, unsigned long long* cyc, int blocks, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int it = 0; it < 3; ++it) k<V><<<blocks, 256, lds>>>(gin, gout, cyc, 1);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s blocks %4d lds %6zu: median %8.1f cycles/stage (16 MFMA = 1024 ideal)\n", name, blocks, lds,
           (double)h[h.size() / 2] / REP);
}

int main() {
    float *gin, *gout; unsigned long long* cyc;
    hipMalloc(&gin, 1 << 16); hipMalloc(&gout, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 4 * 8);
    std::vector<float> h(1 << 14);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(gin, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int two = 0; two < 2; ++two) {
        const int blocks = two ? 512 : 256;
        const size_t lds = two ? 68 * 1024 : 100 * 1024;
        printf("---- %d wave(s) per SIMD\n", two + 1);
        run<0>("V0 pure MFMA (4 acc)", gin, gout, cyc, blocks, lds);
        run<1>("V1 select before each MFMA pair", gin, gout, cyc, blocks, lds);
        run<2>("V2 8 selects, then 16 MFMA", gin, gout, cyc, blocks, lds);
        run<3>("V3 LDS prefetch + interleaved selects", gin, gout, cyc, blocks, lds);
        run<4>("V4 LDS prefetch, no selects", gin, gout, cyc, blocks, lds);
        run<7>("V7 V3 + loads spread in MFMA gaps", gin, gout, cyc, blocks, lds);
        run<8>("V8 V3 + 1 DS read per 2 MFMA", gin, gout, cyc, blocks, lds);
        run<10>("V10 16 LDS reads burst (layer2 shape)", gin, gout, cyc, blocks, lds);
        run<9>("V9 16 LDS reads spread (layer2 shape)", gin, gout, cyc, blocks, lds);
        run<11>("V11 2 x ds_read_b128 per stage", gin, gout, cyc, blocks, lds);
        run<12>("V12 4 x ds_read_b64 per stage", gin, gout, cyc, blocks, lds);
        run<14>("V14 1 x ds_read_b32 per stage", gin, gout, cyc, blocks, lds);
        run<13>("V13 2 x global_load_dwordx4 per stage", gin, gout, cyc, blocks, lds);
        run<15>("V15 8 x v_mov after the MFMAs", gin, gout, cyc, blocks, lds);
        run<5>("V5 pure MFMA, 2 acc alternating", gin, gout, cyc, blocks, lds);
        run<6>("V6 pure MFMA, 1 acc dependent chain", gin, gout, cyc, blocks, lds);
    }
    return 0;
}
