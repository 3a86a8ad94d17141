// Probe: operand layout of v_mfma_f32_16x16x32_bf16 on gfx950 (assumed: lane l holds A[l&15][8*(l>>4)+j],
// B[8*(l>>4)+j][l&15], j=0..7; D[4*(l>>4)+r][l&15]).  Asymmetric integer data, exact in bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ short f2bf(float x) { unsigned u = __float_as_uint(x); return (short)(u >> 16); }
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[i * 32 + 8 * g + j]); b[j] = f2bf(B[(8 * g + j) * 16 + i]); }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
int main() {
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k2 = 0; k2 < 32; ++k2) A[i * 32 + k2] = (float)((i * 7 + k2 * 3) % 13 - 6);
    for (int k2 = 0; k2 < 32; ++k2) for (int j = 0; j < 16; ++j) B[k2 * 16 + j] = (float)((k2 * 5 + j * 11) % 9 - 4);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k2 = 0; k2 < 32; ++k2) R[i * 16 + j] += A[i * 32 + k2] * B[k2 * 16 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
    printf("bf16 16x16x32 layout probe: %d mismatches of 256 (D[1][2]=%g ref %g)\n", bad, D[18], R[18]);
    return bad != 0;
}
