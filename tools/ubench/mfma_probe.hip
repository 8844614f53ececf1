#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ inline __bf16 tobf(float f) { return (__bf16)f; }
// D = A(32x16) * B(16x32); A[i][k], B[k][j] given row-major in global; each lane assembles its fragment per the assumed map
__global__ void probe(const float *A, const float *B, float *D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = tobf(A[(l % 32) * 16 + 8 * (l / 32) + e]);
        b[e] = tobf(B[(8 * (l / 32) + e) * 32 + (l % 32)]);
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = c[r];
    }
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 2 + k * j) % 13 - 6);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) if (hD[i] != ref[i]) ++bad;
    printf("mfma 32x32x16 bf16 layout probe: %d mismatches of 1024\n", bad);
    return bad != 0;
}
