// Micro-benchmark: how much of the fp32 matrix pipe do ONE and TWO wavefronts per SIMD reach - MFMAs alone, with the
// operand loads of the projector GEMM stage (6 x b128 per 32 MFMAs) between them, and with its normalisation VALU work.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_waves.hip -o tools/ubench/mfma_waves && tools/ubench/mfma_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int LOADS, int VALU>
__global__ __launch_bounds__(256, 2) void probe(const f32x4* __restrict__ src, float* out, int stages, unsigned mask) {
    extern __shared__ float lds[];
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* p = src + (size_t)(blockIdx.x & 63) * 4096 + wave * 1024 + lane;
    struct Stage { f32x4 v[6]; };
    Stage s0, s1, s2;                       // ring of 3: operands requested two stages ahead (as conv_nhwc_kernel<1, 4, true, 4>)
#pragma unroll
    for (int t = 0; t < 6; ++t) s0.v[t] = s1.v[t] = s2.v[t] = p[t * 64];
    float m0 = 0.5f, r0 = 1.25f;
    unsigned off = 0;
#define ISSUE(S)                                                                        \
    if (LOADS) {                                                                        \
        _Pragma("unroll") for (int t = 0; t < 6; ++t)                                   \
            if (t < LOADS) S.v[t] = p[((off + t * 64) & mask)];                         \
        off += 384;                                                                     \
    }
#define STAGE(SL, SM)                                                                   \
    ISSUE(SL)                                                                           \
    if (VALU) {                                                                         \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                   \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) SM.v[t][e] = (SM.v[t][e] - m0) * r0; \
    }                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[mt * 4 + t] = MFMA(SM.v[mt][j], SM.v[2 + t][j], acc[mt * 4 + t]); \
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                  \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int s = 0; s < stages; s += 3) {
        STAGE(s2, s0)
        STAGE(s0, s1)
        STAGE(s1, s2)
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    if (sum == 12345.678f) out[threadIdx.x] = sum + lds[0];
}

template <int LOADS, int VALU>
static void run(const char* what, const f32x4* src, float* out, int cus) {
    const int stages = 4095;
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const size_t lds = per_cu == 1 ? 100 * 1024 : 60 * 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(probe<LOADS, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            probe<LOADS, VALU><<<dim3(cus * per_cu), dim3(256), lds, 0>>>(src, out, stages, 0xfffu);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        // MFMAs per SIMD: per_cu waves x stages x 32, 64 pipe cycles each
        const double cyc = (double)per_cu * stages * 32 * 64;
        const double tf = (double)cus * per_cu * 4 * stages * 32 * (2.0 * 32 * 32 * 2) / (best * 1e-3) / 1e12;
        printf("%-44s %d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3; pipe cycles %.2f M -> %.0f MHz if the pipe were full)\n",
               what, per_cu, best, tf, tf / 157.3, cyc / 1e6, cyc / (best * 1e-3) / 1e6);
    }
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    f32x4* src; float* out;
    hipMalloc(&src, 64 * 4096 * sizeof(f32x4) + (1 << 20));
    hipMemset(src, 0, 64 * 4096 * sizeof(f32x4) + (1 << 20));
    hipMalloc(&out, 4096);
    run<0, 0>("MFMAs alone", src, out, cus);
    run<0, 1>("MFMAs + 8 VALU per 32", src, out, cus);
    run<2, 0>("MFMAs + 2 loads (b128) per 32", src, out, cus);
    run<6, 0>("MFMAs + 6 loads (b128) per 32", src, out, cus);
    run<6, 1>("MFMAs + 6 loads + 8 VALU per 32", src, out, cus);
    return 0;
}
