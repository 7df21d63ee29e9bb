// Sparse binary16 vector x matrix for the bsz=1 channel-mix (relu^2 output x ffn.value).
//
// Reference: spvecmatmul_noindices (Albatross/cuda/rwkv7_state_fwd_fp16.cu:222-310) and the
// ROCm path rwkv_mm_sparsity_kernel (Albatross/rwkv_mm_op_triton.py:6-37).  Arithmetic
// follows the Triton path: products of binary16 values summed in binary32, rows whose vector
// entry is +-0 are never read, one rounding to binary16 at the end; like the CUDA path the
// result is ADDED into `out` (caller zeroes it).  Unlike the CUDA path no half atomics are
// used, so the result is deterministic.
//
// gfx950 layout: the matrix is [D][C] row-major, each kept row is a contiguous 2*C-byte
// stream.  A workgroup of 256 lanes takes a chunk of 64 vector entries, compacts the non-zero
// ones with one wave ballot, then streams only those rows with 16-B loads per lane (one
// 4-KiB span of the row per instruction, four rows in flight), accumulating 8 binary32 sums
// per lane per 2048-column tile.  Chunk partials go to a binary32 workspace [D/64][C]; a
// second pass adds them in chunk order.  HBM bytes ~ density * 2*D*C, the same saving the
// reference's skip gives.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kChunk = 64;       // vector entries per workgroup (one ballot)
constexpr int kThreads = 256;
constexpr int kMaxTiles = 8;     // 2048-column tiles per row -> C <= 16384

template <int TILES>
__global__ __launch_bounds__(kThreads) void spmv_partial_kernel(const int D, const int C,
                                                                 const f16 *__restrict__ vec,
                                                                 const f16 *__restrict__ mat,
                                                                 float *__restrict__ part) {
    __shared__ int nz_idx[kChunk];
    __shared__ float nz_val[kChunk];
    __shared__ int nz_count;
    const int tid = threadIdx.x;
    const int d0 = blockIdx.x * kChunk;

    if (tid < 64) {  // wave 0 compacts the chunk's non-zero entries, order preserved
        const int d = d0 + tid;
        const f16 x = d < D ? vec[d] : (f16)0.f;
        const bool nz = x != (f16)0.f;  // +-0 both compare equal to 0
        const unsigned long long m = __ballot(nz);
        if (nz) {
            const int pos = __popcll(m & ((1ull << tid) - 1ull));
            nz_idx[pos] = d;
            nz_val[pos] = (float)x;
        }
        if (tid == 0) nz_count = __popcll(m);
    }
    __syncthreads();
    const int n = nz_count;

    float acc[TILES][8];
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[t][e] = 0.f;

    for (int i = 0; i < n; i++) {
        const float xv = nz_val[i];
        const f16 *row = mat + (int64_t)nz_idx[i] * C;
#pragma unroll
        for (int t = 0; t < TILES; t++) {
            const int col = t * 2048 + tid * 8;
            if (col < C) {
                const f16x8 m8 = *reinterpret_cast<const f16x8 *>(row + col);
#pragma unroll
                for (int e = 0; e < 8; e++) acc[t][e] = acc[t][e] + xv * (float)m8[e];
            }
        }
    }
    float *dst = part + (int64_t)blockIdx.x * C;
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int col = t * 2048 + tid * 8;
        if (col < C) {
#pragma unroll
            for (int e = 0; e < 8; e++) dst[col + e] = acc[t][e];
        }
    }
}

__global__ __launch_bounds__(256) void spmv_reduce_kernel(const int nchunks, const int C,
                                                          const float *__restrict__ part,
                                                          f16 *__restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nchunks; k++) s = s + part[(int64_t)k * C + c];
    out[c] = (f16)((float)out[c] + s);
}

}  // namespace

extern "C" int64_t spmv_fp16_workspace_bytes(int D, int C) {
    if (D <= 0 || C <= 0) return 0;
    return (int64_t)((D + kChunk - 1) / kChunk) * C * (int64_t)sizeof(float);
}

extern "C" int spmv_fp16(int D, int C, const void *vec, const void *mat, void *out, void *workspace,
                         void *stream) {
    if (D <= 0 || C <= 0 || (C & 7) || C > 2048 * kMaxTiles) return CHIRRUP_E_SHAPE;
    if (!vec || !mat || !out || !workspace) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(mat) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return CHIRRUP_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunks = (D + kChunk - 1) / kChunk;
    const int tiles = (C + 2047) / 2048;
    const f16 *v = static_cast<const f16 *>(vec);
    const f16 *m = static_cast<const f16 *>(mat);
    float *p = static_cast<float *>(workspace);
#define LAUNCH(TT) hipLaunchKernelGGL(spmv_partial_kernel<TT>, dim3(nchunks), dim3(kThreads), 0, st, D, C, v, m, p)
    switch (tiles) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        default: LAUNCH(8); break;
    }
#undef LAUNCH
    hipLaunchKernelGGL(spmv_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, st, nchunks, C, p,
                       static_cast<f16 *>(out));
    return (int)hipGetLastError();
}
