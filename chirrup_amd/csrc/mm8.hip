// mm8 (w8a16): y = x @ ((w + 0.5) * ry * rx + my + mx), uint8 weights, binary16 activations.
//
// Reference: kernel_mm_seq_fp16i8 / kernel_mm_one_fp16i8 and their tiled variants,
// scripts/test_mm8/rwkv_pip_operators.cu:59-97, :150-189, :205-558; formulas
// scripts/test_mm8/benchmark.py:114-118 (direct) and :167-179 (algebraic split).
//
// Two code paths:
//   mm8_seq  (B rows)  "direct" kernel: the as-coded expression, binary32 accumulate over j in
//            order, one rounding per operation -> bit-identical to oracle_mm8_seq.  A workgroup
//            owns 32 batch rows x 256 output columns; the dequantised weight is formed once
//            per (j,k) and reused for the 32 rows, x is broadcast from LDS.
//            TODO(round 2): MFMA path (u8 -> f16 in registers, v_mfma_f32_32x32x16_f16).
//   mm8_one  (GEMV)    split over j like the reference (24 slices in the reference, here
//            enough slices to fill the chip), binary32 atomicAdd into the caller-zeroed y.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;

constexpr int kBT = 32;    // batch rows per workgroup
constexpr int kKT = 256;   // output columns per workgroup (one per lane)
constexpr int kJT = 32;    // reduction tile staged in LDS

__global__ __launch_bounds__(kKT) void mm8_seq_direct_kernel(
    const int B, const int N, const int M, const f16 *__restrict__ x, const int x_stride,
    const uint8_t *__restrict__ w, const int w_stride, const f16 *__restrict__ mx,
    const f16 *__restrict__ rx, const f16 *__restrict__ my, const f16 *__restrict__ ry,
    f16 *__restrict__ y, const int y_stride) {
    __shared__ float sx[kJT][kBT];   // x tile, [j][i] so that a fixed j is one broadcast row
    __shared__ float sry[kJT], smy[kJT];
    const int tid = threadIdx.x;
    const int k = blockIdx.x * kKT + tid;
    const int i0 = blockIdx.y * kBT;
    const bool kin = k < M;
    const float rxk = kin ? (float)rx[k] : 0.f;
    const float mxk = kin ? (float)mx[k] : 0.f;

    float acc[kBT];
#pragma unroll
    for (int i = 0; i < kBT; i++) acc[i] = 0.f;

    for (int j0 = 0; j0 < N; j0 += kJT) {
        __syncthreads();
        // 256 lanes stage a 32(j) x 32(i) tile of x: lane -> (i = tid/8.., j = ..)
#pragma unroll
        for (int e = tid; e < kJT * kBT; e += kKT) {
            const int i = e / kJT, jj = e % kJT;
            const int gi = i0 + i, gj = j0 + jj;
            sx[jj][i] = (gi < B && gj < N) ? (float)x[(int64_t)gi * x_stride + gj] : 0.f;
        }
        if (tid < kJT) {
            const int gj = j0 + tid;
            sry[tid] = gj < N ? (float)ry[gj] : 0.f;
            smy[tid] = gj < N ? (float)my[gj] : 0.f;
        }
        __syncthreads();
        const int jn = (N - j0) < kJT ? (N - j0) : kJT;
        if (kin) {
            for (int jj = 0; jj < jn; jj++) {
                const float wq = (float)w[(int64_t)(j0 + jj) * w_stride + k];
                // ((w + 0.5) * rx * ry) + mx + my, left to right (operators.cu:76-79)
                const float dq = (wq + 0.5f) * rxk * sry[jj] + mxk + smy[jj];
#pragma unroll
                for (int i = 0; i < kBT; i++) acc[i] = acc[i] + sx[jj][i] * dq;
            }
        }
    }
    if (kin) {
#pragma unroll
        for (int i = 0; i < kBT; i++) {
            const int gi = i0 + i;
            if (gi < B) y[(int64_t)gi * y_stride + k] = (f16)acc[i];
        }
    }
}

constexpr int kOneCols = 4;      // columns per lane (one dword of u8)
constexpr int kOneThreads = 256;

__global__ __launch_bounds__(kOneThreads) void mm8_one_kernel(
    const int N, const int M, const int rows_per_block, const f16 *__restrict__ x,
    const uint8_t *__restrict__ w, const int w_stride, const f16 *__restrict__ mx,
    const f16 *__restrict__ rx, const f16 *__restrict__ my, const f16 *__restrict__ ry,
    float *__restrict__ y) {
    const int k0 = (blockIdx.x * kOneThreads + threadIdx.x) * kOneCols;
    const int j0 = blockIdx.y * rows_per_block;
    const int j1 = (j0 + rows_per_block) < N ? (j0 + rows_per_block) : N;
    if (k0 >= M) return;
    float rxk[kOneCols], mxk[kOneCols], acc[kOneCols];
#pragma unroll
    for (int c = 0; c < kOneCols; c++) {
        const bool in = k0 + c < M;
        rxk[c] = in ? (float)rx[k0 + c] : 0.f;
        mxk[c] = in ? (float)mx[k0 + c] : 0.f;
        acc[c] = 0.f;
    }
    const bool vec_ok = (k0 + kOneCols <= M) && ((w_stride & 3) == 0);
    for (int j = j0; j < j1; j++) {
        const float xj = (float)x[j], ryj = (float)ry[j], myj = (float)my[j];
        uint32_t q = 0;
        const uint8_t *wr = w + (int64_t)j * w_stride + k0;
        if (vec_ok) {
            q = *reinterpret_cast<const uint32_t *>(wr);
        } else {
#pragma unroll
            for (int c = 0; c < kOneCols; c++)
                if (k0 + c < M) q |= (uint32_t)wr[c] << (8 * c);
        }
#pragma unroll
        for (int c = 0; c < kOneCols; c++) {
            const float wq = (float)((q >> (8 * c)) & 0xffu);
            const float dq = (wq + 0.5f) * rxk[c] * ryj + mxk[c] + myj;
            acc[c] = acc[c] + xj * dq;
        }
    }
#pragma unroll
    for (int c = 0; c < kOneCols; c++)
        if (k0 + c < M) atomicAdd(&y[k0 + c], acc[c]);
}

}  // namespace

extern "C" int64_t mm8_seq_workspace_bytes(int B, int N, int M) {
    (void)B; (void)N; (void)M;
    return 0;  // the direct kernel needs none; the MFMA path will stage xs = x*ry here
}

extern "C" int mm8_seq(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                       const void *mx, const void *rx, const void *my, const void *ry, void *y,
                       int y_stride, void *workspace, void *stream) {
    (void)workspace;
    if (B <= 0 || N <= 0 || M <= 0 || x_stride < N || w_stride < M || y_stride < M) return CHIRRUP_E_SHAPE;
    if (!x || !w || !mx || !rx || !my || !ry || !y) return CHIRRUP_E_NULL;
    const dim3 grid((M + kKT - 1) / kKT, (B + kBT - 1) / kBT);
    if (grid.y > 65535u) return CHIRRUP_E_SHAPE;
    hipLaunchKernelGGL(mm8_seq_direct_kernel, grid, dim3(kKT), 0, static_cast<hipStream_t>(stream), B, N, M,
                       static_cast<const f16 *>(x), x_stride, static_cast<const uint8_t *>(w), w_stride,
                       static_cast<const f16 *>(mx), static_cast<const f16 *>(rx), static_cast<const f16 *>(my),
                       static_cast<const f16 *>(ry), static_cast<f16 *>(y), y_stride);
    return (int)hipGetLastError();
}

extern "C" int mm8_one(int N, int M, const void *x, const void *w, int w_stride, const void *mx,
                       const void *rx, const void *my, const void *ry, float *y, void *stream) {
    if (N <= 0 || M <= 0 || w_stride < M) return CHIRRUP_E_SHAPE;
    if (!x || !w || !mx || !rx || !my || !ry || !y) return CHIRRUP_E_NULL;
    if (reinterpret_cast<uintptr_t>(w) & 3) return CHIRRUP_E_ALIGN;
    const int col_blocks = (M + kOneThreads * kOneCols - 1) / (kOneThreads * kOneCols);
    // enough j-slices that col_blocks * slices >= ~1024 workgroups, at least 32 rows each
    int slices = (1024 + col_blocks - 1) / col_blocks;
    int rows = (N + slices - 1) / slices;
    if (rows < 32) rows = 32;
    slices = (N + rows - 1) / rows;
    if (slices > 65535) return CHIRRUP_E_SHAPE;
    hipLaunchKernelGGL(mm8_one_kernel, dim3(col_blocks, slices), dim3(kOneThreads), 0,
                       static_cast<hipStream_t>(stream), N, M, rows, static_cast<const f16 *>(x),
                       static_cast<const uint8_t *>(w), w_stride, static_cast<const f16 *>(mx),
                       static_cast<const f16 *>(rx), static_cast<const f16 *>(my), static_cast<const f16 *>(ry), y);
    return (int)hipGetLastError();
}
