// mm8 (w8a16): y = x @ ((w + 0.5) * ry * rx + my + mx), uint8 weights, binary16 activations.
//
// Reference: kernel_mm_seq_fp16i8 / kernel_mm_one_fp16i8 and their tiled variants,
// scripts/test_mm8/rwkv_pip_operators.cu:59-97, :150-189, :205-558; formulas
// scripts/test_mm8/benchmark.py:114-118 (direct) and :167-179 (algebraic split).
//
// Code paths:
//   mm8_seq / mm8_seq_opt (B rows; the reference's op names, rwkv_pip_wrapper.cpp:206-211, weights w [N][M] row-major):
//            the MFMA path.  The uint8 matrix is re-laid once per call into the K-contiguous 8-KiB tile images of the
//            ring GEMM (mm8_pack: a transposing copy through LDS, N*M bytes of workspace) and multiplied by mm8t_seq
//            (csrc/skinny_gemm.hip: u8 -> f16 in registers, v_mfma_f32_32x32x16_f16, the reference's algebraically
//            split form).  Callers with static weights pack once (mm8_pack) and call mm8t_seq(w_tiled = 1) themselves
//            -- chirrup_amd.ops.mm8_seq caches the packed matrix per weight tensor.  Shapes the tile images cannot
//            hold (N % 64, M % 128, 16-byte alignment) fall back to mm8_seq_direct.
//   mm8_seq_direct      the as-coded expression, binary32 accumulate over j in order, one rounding per operation
//            -> bit-identical to oracle_mm8_seq.  A workgroup owns 32 batch rows x 256 output columns; the
//            dequantised weight is formed once per (j,k) and reused for the 32 rows, x is broadcast from LDS.
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

// w [N][M] uint8 row-major  ->  the ring GEMM's tile images of wT [M][N]: tile (ng = m / 128, kb = j / 64) is 512 chunks
// of 16 B; chunk c holds, for output column m = 128 ng + (c >> 2), the 16 reduction indices j = 64 kb + 16 lc + (0..15)
// with lc = (c & 3) ^ (((c >> 2) >> 2) & 3)  (the XOR swizzle of skinny_gemm.hip's uint8 W image).
// One workgroup per tile: 64 rows x 128 B in (coalesced 16-B loads), transposed through LDS, 8 KiB out (coalesced).
__global__ __launch_bounds__(256) void mm8_pack_kernel(const int N, const int M, const uint8_t *__restrict__ w,
                                                       const int w_stride, uint8_t *__restrict__ out) {
    __shared__ uint32_t tile[64][33];                     // [j][m / 4], rows padded by one dword
    const int ng = blockIdx.x, kb = blockIdx.y, tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int j = p * 32 + (tid >> 3), c16 = tid & 7;
        const uint4 v = *reinterpret_cast<const uint4 *>(w + (int64_t)(kb * 64 + j) * w_stride + ng * 128 + c16 * 16);
        tile[j][c16 * 4 + 0] = v.x, tile[j][c16 * 4 + 1] = v.y, tile[j][c16 * 4 + 2] = v.z, tile[j][c16 * 4 + 3] = v.w;
    }
    __syncthreads();
    uint8_t *dst = out + ((int64_t)ng * (N / 64) + kb) * 8192;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int c = p * 256 + tid;
        const int nr = c >> 2, lc = (c & 3) ^ ((nr >> 2) & 3);
        const int sh = 8 * (nr & 3);
        uint32_t q[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) acc |= ((tile[lc * 16 + d * 4 + b][nr >> 2] >> sh) & 0xffu) << (8 * b);
            q[d] = acc;
        }
        *reinterpret_cast<uint4 *>(dst + c * 16) = make_uint4(q[0], q[1], q[2], q[3]);
    }
}

// uint8 weights wT [M_out][N_in] (K-contiguous; the 8-KiB tile images of mm8_pack / skinny_tile_weight_u8, or row-major with row
// stride w_stride)  ->  the dequantised matrix as binary16 [M_out][N_in] row-major: out[m][k] = fp16(((q + 0.5) * rx[m]) * ry[k] +
// mx[m] + my[k]), the as-coded dequantisation (rwkv_pip_operators.cu:76-79, left to right in binary32) rounded once.  What a
// chunked-prefill forward multiplies through the library GEMM: above 256 rows a product is MFMA-bound, the weights' bytes no longer
// matter, and one 40-us pass that rebuilds a layer's matrix into a reused scratch costs less than re-streaming uint8 tiles
// per 256-row block (round 3: 2500 rows took 10 x 52 us per ffn product against 290 us for the binary16 library call).
__global__ __launch_bounds__(256) void mm8_dequant_kernel(const int M, const int N, const uint8_t *__restrict__ wT, const int64_t w_stride,
                                                          const int tiled, const f16 *__restrict__ rx, const f16 *__restrict__ mx,
                                                          const f16 *__restrict__ ry, const f16 *__restrict__ my, f16 *__restrict__ out) {
    const int ng = blockIdx.x, kb = blockIdx.y, tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int c = p * 256 + tid;                                   // 16-byte chunk of the (128 x 64) tile
        const int nr = c >> 2;
        const int lc = tiled ? ((c & 3) ^ ((nr >> 2) & 3)) : (c & 3);  // logical chunk held at position c & 3 of row nr
        const int m = ng * 128 + nr, k0 = kb * 64 + lc * 16;
        if (m >= M) continue;
        const uint8_t *src = tiled ? wT + ((int64_t)ng * (N / 64) + kb) * 8192 + c * 16 : wT + (int64_t)m * w_stride + k0;
        const uint4 q4 = *reinterpret_cast<const uint4 *>(src);
        const uint32_t qs[4] = {q4.x, q4.y, q4.z, q4.w};
        const float rxm = (float)rx[m], mxm = (float)mx[m];
        f16 o[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float wq = (float)((qs[e >> 2] >> (8 * (e & 3))) & 0xffu);
            o[e] = (f16)((wq + 0.5f) * rxm * (float)ry[k0 + e] + mxm + (float)my[k0 + e]);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out + (int64_t)m * N + k0);
        dst[0] = *reinterpret_cast<const uint4 *>(o);
        dst[1] = *reinterpret_cast<const uint4 *>(o + 8);
    }
}

inline bool mfma_eligible(int N, int M, const void *x, int x_stride, const void *w, int w_stride, int y_stride) {
    return N % 64 == 0 && M % 128 == 0 && (w_stride & 15) == 0 && (x_stride & 7) == 0 && (y_stride & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

inline int64_t round256(int64_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t mm8_packed_bytes(int N, int M) {
    return (N > 0 && M > 0 && N % 64 == 0 && M % 128 == 0) ? (int64_t)N * M : 0;
}

extern "C" int mm8_pack(int N, int M, const void *w, int w_stride, void *packed, void *stream) {
    if (N <= 0 || M <= 0 || (N % 64) || (M % 128) || w_stride < M || (w_stride & 15)) return CHIRRUP_E_SHAPE;
    if (!w || !packed) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15)) return CHIRRUP_E_ALIGN;
    if (N / 64 > 65535) return CHIRRUP_E_SHAPE;
    hipLaunchKernelGGL(mm8_pack_kernel, dim3(M / 128, N / 64), dim3(256), 0, static_cast<hipStream_t>(stream), N, M,
                       static_cast<const uint8_t *>(w), w_stride, static_cast<uint8_t *>(packed));
    return (int)hipGetLastError();
}

extern "C" int mm8_dequant_f16(int M_out, int N_in, const void *wT, int64_t w_stride, int w_tiled, const void *rx, const void *mx,
                               const void *ry, const void *my, void *out, void *stream) {
    if (M_out <= 0 || N_in <= 0 || (N_in % 64) || (w_tiled && (M_out % 128)) || (!w_tiled && (w_stride < N_in || (w_stride & 15)))) return CHIRRUP_E_SHAPE;
    if (!wT || !rx || !mx || !ry || !my || !out) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(wT) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return CHIRRUP_E_ALIGN;
    if (N_in / 64 > 65535) return CHIRRUP_E_SHAPE;
    hipLaunchKernelGGL(mm8_dequant_kernel, dim3((M_out + 127) / 128, N_in / 64), dim3(256), 0, static_cast<hipStream_t>(stream), M_out, N_in,
                       static_cast<const uint8_t *>(wT), w_stride, w_tiled ? 1 : 0, static_cast<const f16 *>(rx), static_cast<const f16 *>(mx),
                       static_cast<const f16 *>(ry), static_cast<const f16 *>(my), static_cast<f16 *>(out));
    return (int)hipGetLastError();
}

extern "C" int64_t mm8_seq_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0 || (N % 64) || (M % 128)) return 0;      // the direct kernel needs none
    return round256((int64_t)N * M) + mm8t_workspace_bytes(B, N, M, 0) + 256;
}

extern "C" int mm8_seq_direct(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                              const void *mx, const void *rx, const void *my, const void *ry, void *y,
                              int y_stride, void *stream);

extern "C" int mm8_seq(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                       const void *mx, const void *rx, const void *my, const void *ry, void *y,
                       int y_stride, void *workspace, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || x_stride < N || w_stride < M || y_stride < M) return CHIRRUP_E_SHAPE;
    if (!x || !w || !mx || !rx || !my || !ry || !y) return CHIRRUP_E_NULL;
    if (!mfma_eligible(N, M, x, x_stride, w, w_stride, y_stride))
        return mm8_seq_direct(B, N, M, x, x_stride, w, w_stride, mx, rx, my, ry, y, y_stride, stream);
    if (!workspace) return CHIRRUP_E_NULL;
    unsigned char *ws = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    int rc = mm8_pack(N, M, w, w_stride, ws, stream);
    if (rc) return rc;
    return mm8t_seq(B, N, M, x, x_stride, ws, N, 1, mx, rx, my, ry, y, y_stride, 0, 0, ws + round256((int64_t)N * M), stream);
}

extern "C" int mm8_seq_opt(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                           const void *mx, const void *rx, const void *my, const void *ry, void *y,
                           int y_stride, void *workspace, void *stream) {
    return mm8_seq(B, N, M, x, x_stride, w, w_stride, mx, rx, my, ry, y, y_stride, workspace, stream);
}

extern "C" int mm8_seq_direct(int B, int N, int M, const void *x, int x_stride, const void *w, int w_stride,
                              const void *mx, const void *rx, const void *my, const void *ry, void *y,
                              int y_stride, void *stream) {
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
