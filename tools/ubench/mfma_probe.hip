// mfma_probe.hip - what v_mfma_f32_32x32x16_bf16 computes, bit for bit: one instruction per wave on random bf16
// operands with a wide exponent spread; inputs and outputs are dumped for offline comparison with candidate models
// (tools/mfma_models.py).   hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip && ./mfma_probe out.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// per wave: A (32 rows x 16 k) bf16, B (16 k x 32 cols) bf16, C (32x32) f32 -> D (32x32) f32
__global__ void probe(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    const int lane = threadIdx.x, w = blockIdx.x;
    const uint16_t* a = A + (size_t)w * 512; const uint16_t* b = B + (size_t)w * 512;
    const float* c = C + (size_t)w * 1024; float* d = D + (size_t)w * 1024;
    uint4 av, bv;
    uint16_t ta[8], tb[8];
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 5) + j;
        ta[j] = a[(lane & 31) * 16 + k];           // A[row = lane&31][k]
        tb[j] = b[k * 32 + (lane & 31)];           // B[k][col = lane&31]
    }
    av.x = ta[0] | (ta[1] << 16); av.y = ta[2] | (ta[3] << 16); av.z = ta[4] | (ta[5] << 16); av.w = ta[6] | (ta[7] << 16);
    bv.x = tb[0] | (tb[1] << 16); bv.y = tb[2] | (tb[3] << 16); bv.z = tb[4] | (tb[5] << 16); bv.w = tb[6] | (tb[7] << 16);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        acc[r] = c[row * 32 + col];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        d[row * 32 + col] = acc[r];
    }
}
int main(int argc, char** argv) {
    const int W = 64;
    std::mt19937_64 rng(7);
    std::vector<uint16_t> A(W * 512), B(W * 512);
    std::vector<float> C(W * 1024), D(W * 1024);
    auto rbf = [&](int spread) -> uint16_t {       // random finite bf16 with exponent in [127-spread, 127+spread]
        const uint32_t sign = rng() & 1, exp = 127 - spread + (rng() % (2 * spread + 1)), man = rng() & 0x7F;
        return (uint16_t)((sign << 15) | (exp << 7) | man);
    };
    for (int w = 0; w < W; ++w) {
        const int spread = (w % 4 == 0) ? 0 : (w % 4 == 1) ? 3 : (w % 4 == 2) ? 12 : 30;
        for (int i = 0; i < 512; ++i) { A[w * 512 + i] = rbf(spread); B[w * 512 + i] = rbf(spread); }
        for (int i = 0; i < 1024; ++i) {
            const uint16_t h = rbf(spread); uint32_t u = ((uint32_t)h << 16) | (uint32_t)(rng() & 0xFFFF);
            float f; memcpy(&f, &u, 4);
            C[w * 1024 + i] = (w % 8 < 4) ? f : 0.0f;
        }
    }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    probe<<<W, 64>>>(dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    FILE* f = fopen(argc > 1 ? argv[1] : "mfma_probe.bin", "wb");
    const int hdr = W; fwrite(&hdr, 4, 1, f);
    fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, C.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("wrote %d waves, D[0]=%g, err=%s\n", W, D[0], hipGetErrorString(hipGetLastError()));
    return 0;
}
