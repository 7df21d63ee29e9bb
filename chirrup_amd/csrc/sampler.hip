// Penalties + greedy selection in ONE pass over the logits (rows of V binary16 values).
//
// Replaces, for rows decoded greedily, the chain at chirrup/worker.py:724-740:
//     occurrence[row] *= penalty_decay[row]                        (fp32 * fp16 -> fp32)
//     logits[row]     -= alpha_presence[row] + occurrence[row] * frequency_penalty[row]
//                        (fp32 arithmetic, result rounded back into the fp16 logits, in place)
//     token = sample(logits)   with temperature 0 -> (T=1, top_p=0): only the largest probability
//                              survives (chirrup/utils/samplers.py:195-197, :214-221) = arg-max
// and the B separate `.item()` host syncs by one int32 id per row in a device buffer.
// Ties between equal largest logits resolve to the LOWEST token id (the reference draws among the
// tied ids at random).
//
// One workgroup of 256 lanes per row; 8 logits (16 B), 8 occurrence and 8 alpha values (2 x 32 B
// each) per lane and step; occurrence / alpha / penalty vectors are addressed through the row's
// slot (slot_idx[row], or row when NULL) so that the worker's tables never move.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void penalize_argmax_kernel(
    const int V, f16 *__restrict__ logits, float *__restrict__ occurrence, const float *__restrict__ alpha,
    const f16 *__restrict__ decay, const f16 *__restrict__ freq, const int32_t *__restrict__ slot_idx,
    int32_t *__restrict__ ids, const int apply_penalty) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const int row = blockIdx.x;
    const int64_t slot = slot_idx ? (int64_t)slot_idx[row] : (int64_t)row;
    f16 *lg = logits + (int64_t)row * V;
    float *occ = occurrence + slot * V;
    const float *al = alpha + slot * V;
    const float dk = apply_penalty ? (float)decay[slot] : 1.f;
    const float fq = apply_penalty ? (float)freq[slot] : 0.f;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int c = threadIdx.x * 8; c < V; c += 256 * 8) {
        f16x8 l8 = *reinterpret_cast<const f16x8 *>(lg + c);
        if (apply_penalty) {
            f32x4 o0 = *reinterpret_cast<const f32x4 *>(occ + c), o1 = *reinterpret_cast<const f32x4 *>(occ + c + 4);
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(al + c), a1 = *reinterpret_cast<const f32x4 *>(al + c + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                o0[e] = o0[e] * dk;
                o1[e] = o1[e] * dk;
                l8[e] = (f16)((float)l8[e] - (a0[e] + o0[e] * fq));
                l8[e + 4] = (f16)((float)l8[e + 4] - (a1[e] + o1[e] * fq));
            }
            *reinterpret_cast<f32x4 *>(occ + c) = o0;
            *reinterpret_cast<f32x4 *>(occ + c + 4) = o1;
            *reinterpret_cast<f16x8 *>(lg + c) = l8;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = (float)l8[e];
            if (v > best) {           // strictly greater: the first (lowest) index of a tie wins
                best = v;
                best_i = c + e;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) {
            best = ov;
            best_i = oi;
        }
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_val[wid] = best;
        s_idx[wid] = best_i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; w++)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < best_i)) {
                best = s_val[w];
                best_i = s_idx[w];
            }
        ids[row] = best_i == 0x7fffffff ? 0 : best_i;     // all-NaN / all -inf row: id 0
    }
}

}  // namespace

extern "C" int rwkv7_penalize_argmax(int B, int V, void *logits, float *occurrence, const float *alpha_presence,
                                     const void *penalty_decay, const void *frequency_penalty,
                                     const int32_t *slot_idx, int32_t *ids, void *stream) {
    if (B <= 0 || V <= 0 || (V & 7)) return CHIRRUP_E_SHAPE;
    if (!logits || !ids) return CHIRRUP_E_NULL;
    const int pen = occurrence != nullptr;
    if (pen && (!alpha_presence || !penalty_decay || !frequency_penalty)) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(logits) & 15) || (reinterpret_cast<uintptr_t>(occurrence) & 15) ||
        (reinterpret_cast<uintptr_t>(alpha_presence) & 15))
        return CHIRRUP_E_ALIGN;
    hipLaunchKernelGGL(penalize_argmax_kernel, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), V,
                       static_cast<f16 *>(logits), occurrence, alpha_presence, static_cast<const f16 *>(penalty_decay),
                       static_cast<const f16 *>(frequency_penalty), slot_idx, ids, pen);
    return (int)hipGetLastError();
}
