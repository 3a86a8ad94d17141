// OctbitMatMul on the GPU: dynamic u8 activation quantisation, s8 weights, int16-saturating pair
// sums, bias correction and rescale -- octbit/octbit_mat_mul_op.cc:90-181, bit-exact.
//   pass 1  per-row min/max                      (:92-99; the reference scans the whole matrix --
//           pass 1b folds the rows into one range unless per_row_scale is set)
//   pass 2  one workgroup per activation row: quantise the row into LDS (:105-124), then each thread
//           produces output columns n = tid, tid+256, ...: K/4 dword loads of Wq[n,:], two saturated
//           u8*s8 pair sums per dword folded into the four i32 lanes of the SSE accumulator (:147-170),
//           float lane sum, bias, scale (:172-179).
// Integer byte work, HBM/L2-bound on Wq; deliberately not reshaped onto MFMA because
// v_mfma_i32_*_i8 accumulates exactly in i32 and cannot reproduce the int16 saturation.
#include "kws_internal.h"

namespace kws {

__global__ void __launch_bounds__(256) octbit_range_kernel(const float* __restrict__ x, int K, float* __restrict__ ws) {
    const int a = blockIdx.x;
    const float* row = x + (size_t)a * K;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    for (int k = threadIdx.x; k < K; k += 256) { const float v = row[k]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off)); mx = fmaxf(mx, __shfl_down(mx, off)); }
    __shared__ float pmn[4], pmx[4];
    if ((threadIdx.x & 63) == 0) { pmn[threadIdx.x >> 6] = mn; pmx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 + 2 * a] = fminf(fminf(pmn[0], pmn[1]), fminf(pmn[2], pmn[3]));
        ws[3 + 2 * a] = fmaxf(fmaxf(pmx[0], pmx[1]), fmaxf(pmx[2], pmx[3]));
    }
}

__global__ void __launch_bounds__(256) octbit_fold_kernel(int A, float* __restrict__ ws) {
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    for (int a = threadIdx.x; a < A; a += 256) { mn = fminf(mn, ws[2 + 2 * a]); mx = fmaxf(mx, ws[3 + 2 * a]); }
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off)); mx = fmaxf(mx, __shfl_down(mx, off)); }
    __shared__ float pmn[4], pmx[4];
    if ((threadIdx.x & 63) == 0) { pmn[threadIdx.x >> 6] = mn; pmx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[0] = fminf(fminf(pmn[0], pmn[1]), fminf(pmn[2], pmn[3]));
        ws[1] = fmaxf(fmaxf(pmx[0], pmx[1]), fmaxf(pmx[2], pmx[3]));
    }
}

__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

__global__ void __launch_bounds__(256) octbit_mm_kernel(const float* __restrict__ x, const int8_t* __restrict__ Wq,
                                                        float scale_w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int K, int N, int per_row,
                                                        const float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char q[];
    const int a = blockIdx.x;
    const float mn = per_row ? ws[2 + 2 * a] : ws[0];
    const float mx = per_row ? ws[3 + 2 * a] : ws[1];
    const bool is_signed = mn < 0.f;
    const float bscale = is_signed ? fmaxf(-mn, mx) / 127.0f : mx / 254.0f;
    const float* row = x + (size_t)a * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        // C round() = half away from zero on the float quotient (:112,:121), then the quint8 cast
        const float r = roundf(row[k] / bscale);
        q[k] = (unsigned char)(int)(is_signed ? r + 127.0f : r);
    }
    __syncthreads();
    const float scale = scale_w * bscale;
    const uint32_t* q4 = reinterpret_cast<const uint32_t*>(q);
    for (int n = threadIdx.x; n < N; n += 256) {
        const uint32_t* w4 = reinterpret_cast<const uint32_t*>(Wq + (size_t)n * K);
        int lane0 = 0, lane1 = 0, lane2 = 0, lane3 = 0;
        for (int k4 = 0; k4 < K / 4; k4 += 2) {
            const uint32_t qa = q4[k4], qb = q4[k4 + 1];
            const uint32_t wa = w4[k4], wb = w4[k4 + 1];
            const int p0 = sat16((int)(qa & 0xff) * (int)(int8_t)(wa & 0xff) + (int)((qa >> 8) & 0xff) * (int)(int8_t)((wa >> 8) & 0xff));
            const int p1 = sat16((int)((qa >> 16) & 0xff) * (int)(int8_t)((wa >> 16) & 0xff) + (int)(qa >> 24) * (int)(int8_t)(wa >> 24));
            const int p2 = sat16((int)(qb & 0xff) * (int)(int8_t)(wb & 0xff) + (int)((qb >> 8) & 0xff) * (int)(int8_t)((wb >> 8) & 0xff));
            const int p3 = sat16((int)((qb >> 16) & 0xff) * (int)(int8_t)((wb >> 16) & 0xff) + (int)(qb >> 24) * (int)(int8_t)(wb >> 24));
            lane0 += p0; lane1 += p1; lane2 += p2; lane3 += p3;   // pair index mod 4 = SSE lane
        }
        float o = 0.f;
        o += (float)lane0; o += (float)lane1; o += (float)lane2; o += (float)lane3;
        if (is_signed) o -= bias[n];
        out[(size_t)a * N + n] = o * scale;
    }
}

hipError_t launch_octbit_matmul(const float* x, const int8_t* Wq, float scale_w, const float* bias, float* out,
                                int A, int K, int N, int per_row, float* range_ws, hipStream_t st) {
    hipLaunchKernelGGL(octbit_range_kernel, dim3(A), dim3(256), 0, st, x, K, range_ws);
    if (!per_row) hipLaunchKernelGGL(octbit_fold_kernel, dim3(1), dim3(256), 0, st, A, range_ws);
    hipLaunchKernelGGL(octbit_mm_kernel, dim3(A), dim3(256), (size_t)K, st, x, Wq, scale_w, bias, out, K, N, per_row,
                       range_ws);
    return hipGetLastError();
}

}  // namespace kws
