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

#include <atomic>
#include <stdint.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kArgmaxThreads = 1024;       // 4x the loads in flight per row of 256 lanes (200 rows are 200 workgroups on 256 CUs): 46 -> 3x us at bsz 200
__global__ __launch_bounds__(kArgmaxThreads) void penalize_argmax_kernel(
    const int V, f16 *__restrict__ logits, float *__restrict__ occurrence, const float *__restrict__ alpha,
    const f16 *__restrict__ decay, const f16 *__restrict__ freq, const int32_t *__restrict__ slot_idx,
    int32_t *__restrict__ ids, const int apply_penalty) {
    __shared__ float s_val[kArgmaxThreads / 64];
    __shared__ int s_idx[kArgmaxThreads / 64];
    const int row = blockIdx.x;
    const int64_t slot = slot_idx ? (int64_t)slot_idx[row] : (int64_t)row;
    f16 *lg = logits + (int64_t)row * V;
    float *occ = occurrence + slot * V;
    const float *al = alpha + slot * V;
    const float dk = apply_penalty ? (float)decay[slot] : 1.f;
    const float fq = apply_penalty ? (float)freq[slot] : 0.f;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int c = threadIdx.x * 8; c < V; c += kArgmaxThreads * 8) {
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
        for (int w = 1; w < kArgmaxThreads / 64; w++)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < best_i)) {
                best = s_val[w];
                best_i = s_idx[w];
            }
        ids[row] = best_i == 0x7fffffff ? 0 : best_i;     // all-NaN / all -inf row: id 0
    }
}

// The same step when the worker knows WHICH entries of a slot's tables can be non-zero (round 4).  A request's occurrence / presence
// rows are zero except at the token ids it has sampled so far -- a few hundred of 65 536 -- yet the dense pass reads and writes
// 768 KB of tables per row and step (184 MB per step at bsz 200: 42 us).  The tables stay what they are (dense, exact, no cap);
// beside them every slot keeps the LIST of ids it has sampled (pen_list [n_slots][cap], pen_count [n_slots]; maintained by
// commit_sampled_kernel with a bitmap so that an id is listed once).  pen_count >= 0: only the listed entries are decayed and
// subtracted -- per element the dense kernel's arithmetic, and for every other element that arithmetic is the identity
// (occurrence 0 * decay = 0; fp16(float(logit) - (0 + 0 * freq)) = logit) -- then the arg-max pass reads the logits alone.
// pen_count < 0 (more distinct ids than cap): the dense pass, as before.  Bit-identical to penalize_argmax_kernel either way.
__global__ __launch_bounds__(kArgmaxThreads) void penalize_argmax_listed_kernel(
    const int V, f16 *__restrict__ logits, float *__restrict__ occurrence, const float *__restrict__ alpha,
    const f16 *__restrict__ decay, const f16 *__restrict__ freq, const int32_t *__restrict__ slot_idx,
    int32_t *__restrict__ ids, const int32_t *__restrict__ pen_list, const int32_t *__restrict__ pen_count, const int cap) {
    __shared__ float s_val[kArgmaxThreads / 64];
    __shared__ int s_idx[kArgmaxThreads / 64];
    const int row = blockIdx.x;
    const int64_t slot = slot_idx ? (int64_t)slot_idx[row] : (int64_t)row;
    f16 *lg = logits + (int64_t)row * V;
    float *occ = occurrence + slot * V;
    const float *al = alpha + slot * V;
    const float dk = (float)decay[slot], fq = (float)freq[slot];
    const int n = pen_count[slot];
    const bool dense = n < 0;
    if (!dense) {
        const int32_t *lst = pen_list + slot * cap;
        for (int i = threadIdx.x; i < n; i += kArgmaxThreads) {
            const int id = lst[i];
            const float o = occ[id] * dk;
            occ[id] = o;
            lg[id] = (f16)((float)lg[id] - (al[id] + o * fq));
        }
        __threadfence_block();
        __syncthreads();                               // the arg-max pass below reads what other lanes of this workgroup just wrote
    }
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int c = threadIdx.x * 8; c < V; c += kArgmaxThreads * 8) {
        f16x8 l8 = *reinterpret_cast<const f16x8 *>(lg + c);
        if (dense) {
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
            if (v > best) {
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
        for (int w = 1; w < kArgmaxThreads / 64; w++)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < best_i)) {
                best = s_val[w];
                best_i = s_idx[w];
            }
        ids[row] = best_i == 0x7fffffff ? 0 : best_i;
    }
}

}  // namespace

extern "C" int rwkv7_penalize_argmax_listed(int B, int V, void *logits, float *occurrence, const float *alpha_presence,
                                            const void *penalty_decay, const void *frequency_penalty, const int32_t *slot_idx,
                                            int32_t *ids, const int32_t *pen_list, const int32_t *pen_count, int cap, void *stream) {
    if (B <= 0 || V <= 0 || (V & 7) || cap <= 0) return CHIRRUP_E_SHAPE;
    if (!logits || !ids || !occurrence || !alpha_presence || !penalty_decay || !frequency_penalty || !pen_list || !pen_count) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(logits) & 15) || (reinterpret_cast<uintptr_t>(occurrence) & 15) ||
        (reinterpret_cast<uintptr_t>(alpha_presence) & 15))
        return CHIRRUP_E_ALIGN;
    hipLaunchKernelGGL(penalize_argmax_listed_kernel, dim3((unsigned)B), dim3(kArgmaxThreads), 0, static_cast<hipStream_t>(stream), V,
                       static_cast<f16 *>(logits), occurrence, alpha_presence, static_cast<const f16 *>(penalty_decay),
                       static_cast<const f16 *>(frequency_penalty), slot_idx, ids, pen_list, pen_count, cap);
    return (int)hipGetLastError();
}

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
    hipLaunchKernelGGL(penalize_argmax_kernel, dim3((unsigned)B), dim3(kArgmaxThreads), 0, static_cast<hipStream_t>(stream), V,
                       static_cast<f16 *>(logits), occurrence, alpha_presence, static_cast<const f16 *>(penalty_decay),
                       static_cast<const f16 *>(frequency_penalty), slot_idx, ids, pen);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Device-side consequences of the sampled ids (chirrup/worker.py:527-535; chirrup_amd/worker.py::_commit_sampled): the next
// decode input of the slot, occurrence[slot][id] += penalty_weight[id] (0 for the no-penalty ids), alpha_presence[slot][id] =
// presence[slot].  One lane per row: as torch ops (index_copy_, a gather, index_put_(accumulate=True) -- which sorts its
// indices --, an indexed assignment and their bounds-check kernels) this was ~12 eager launches behind every decode step.
namespace {
__global__ __launch_bounds__(256) void commit_sampled_kernel(const int n, const int V, const int32_t *__restrict__ ids,
                                                             const int32_t *__restrict__ slot_idx, int32_t *__restrict__ last_ids,
                                                             float *__restrict__ occurrence, const float *__restrict__ penalty_weight,
                                                             float *__restrict__ alpha, const float *__restrict__ presence,
                                                             const int64_t presence_stride, const int32_t *__restrict__ status_src,
                                                             int32_t *__restrict__ status_dst, int32_t *__restrict__ pen_list = nullptr,
                                                             int32_t *__restrict__ pen_count = nullptr, uint32_t *__restrict__ pen_bits = nullptr,
                                                             const int cap = 0) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    // the launch-status word of the step that produced these ids travels to the host behind them (one D2H copy for both)
    if (row == 0 && status_dst) *status_dst = status_src ? __hip_atomic_load(status_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (row >= n) return;
    const int id = ids[row];
    const int64_t slot = slot_idx ? (int64_t)slot_idx[row] : (int64_t)row;
    last_ids[slot] = id;
    if (id < 0 || id >= V) return;                     // (cannot be produced by the samplers; the tables are left alone)
    atomicAdd(occurrence + slot * V + id, penalty_weight[id]);     // (slots of a batch are distinct; the atomic keeps accumulate=True's meaning if not)
    alpha[slot * V + id] = presence[slot * presence_stride];
    if (pen_list) {
        // the slot's list of ids whose table entries may be non-zero (penalize_argmax_listed_kernel): each id once (bitmap), in
        // sampling order; more than cap distinct ids -> count = -1, the slot falls back to the dense pass until it is reset.
        // (atomicOr: two rows of ONE launch never share a slot in the worker; if a caller's do, the id may be listed twice --
        // then it would be decayed twice: callers with duplicate slots in a batch must not use the listed form.)
        uint32_t *w = pen_bits + slot * (V >> 5) + (id >> 5);
        const uint32_t bit = 1u << (id & 31);
        if (!(atomicOr(w, bit) & bit)) {
            const int n = pen_count[slot];
            if (n >= 0) {
                if (n < cap) pen_list[slot * cap + n] = id, pen_count[slot] = n + 1;
                else pen_count[slot] = -1;
            }
        }
    }
}
}  // namespace

extern "C" int rwkv7_commit_sampled(int n, int V, const int32_t *ids, const int32_t *slot_idx, int32_t *last_ids, float *occurrence,
                                    const float *penalty_weight, float *alpha_presence, const float *presence,
                                    int64_t presence_stride, const int32_t *status_src, int32_t *status_dst, void *stream) {
    if (n <= 0 || V <= 0 || presence_stride < 0) return CHIRRUP_E_SHAPE;
    if (!ids || !last_ids || !occurrence || !penalty_weight || !alpha_presence || !presence) return CHIRRUP_E_NULL;
    hipLaunchKernelGGL(commit_sampled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, V, ids,
                       slot_idx, last_ids, occurrence, penalty_weight, alpha_presence, presence, presence_stride, status_src, status_dst, nullptr, nullptr, nullptr, 0);
    return (int)hipGetLastError();
}

extern "C" int rwkv7_commit_sampled_listed(int n, int V, const int32_t *ids, const int32_t *slot_idx, int32_t *last_ids, float *occurrence,
                                           const float *penalty_weight, float *alpha_presence, const float *presence,
                                           int64_t presence_stride, const int32_t *status_src, int32_t *status_dst, int32_t *pen_list,
                                           int32_t *pen_count, uint32_t *pen_bits, int cap, void *stream) {
    if (n <= 0 || V <= 0 || (V & 31) || presence_stride < 0 || cap <= 0) return CHIRRUP_E_SHAPE;
    if (!ids || !last_ids || !occurrence || !penalty_weight || !alpha_presence || !presence || !pen_list || !pen_count || !pen_bits) return CHIRRUP_E_NULL;
    hipLaunchKernelGGL(commit_sampled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, V, ids,
                       slot_idx, last_ids, occurrence, penalty_weight, alpha_presence, presence, presence_stride, status_src, status_dst,
                       pen_list, pen_count, pen_bits, cap);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Sort-free top-p / top-k / temperature sampling for the rows that are NOT greedy.
//
// Semantics of sample_logits_rwkv_pip_compatible (chirrup/utils/samplers.py:171-255): probs =
// softmax(logits); cutoff = the probability at which the DESCENDING cumulative sum first reaches top_p,
// everything below it is dropped (ties at the cutoff stay); optional top-k; probs ** (1/T); one draw
// from the remaining mass.  The reference sorts all V probabilities per row; here the row's binary16
// logits sit in LDS (V <= 65536 -> 128 KiB) and the cutoff is found by a search over the bits of the
// 16-bit order-preserving keys of the logits (probability is monotone in the logit), two bits per pass
// over masses and packed keys held in registers (counts, for top-k).  The draw is an inverse-CDF walk in token
// order with a caller-supplied uniform number per row (torch's generator stays the source of
// randomness).  Differences from the reference: ties AT the top-k boundary are all kept, and the
// draw uses one uniform instead of torch.multinomial's stream -- same distribution, different ids.
namespace {

constexpr int kSampThreads = 1024;
constexpr int kSampRed = 3 * 16;               // one slot per (threshold, wave) of a block reduction

__device__ __forceinline__ unsigned key_of(f16 v) {        // larger value <=> larger key
    const unsigned short b = __builtin_bit_cast(unsigned short, v);
    return (b & 0x8000u) ? (unsigned)(unsigned short)~b : (unsigned)(b | 0x8000u);
}

__device__ __forceinline__ float block_reduce_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < kSampThreads / 64; i++) t += red[i];
    return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < kSampThreads / 64; i++) t = fmaxf(t, red[i]);
    return t;
}

// One packed pair of keys d = klo | khi << 16 against three rising thresholds p1 < p2 < p3 (P = p << 16): m_i += mass of the keys
// >= p_i.  v_cmpx narrows EXEC so that each (key, threshold) costs a compare and an add -- the compiler's own form is compare,
// select, add, and a CU issues only 64 lanes of VALU per clock, which is what bounds the cutoff search.
__device__ __forceinline__ void mass_ge3(unsigned d, float mlo, float mhi, unsigned p1, unsigned p2, unsigned p3, unsigned P1,
                                         unsigned P2, unsigned P3, float &m1, float &m2, float &m3) {
    unsigned t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_and_b32_e32 %[t], 0xffff, %[d]\n\t"
        "v_cmpx_le_u32_e32 %[p1], %[t]\n\t"
        "v_add_f32_e32 %[m1], %[m1], %[mlo]\n\t"
        "v_cmpx_le_u32_e32 %[p2], %[t]\n\t"
        "v_add_f32_e32 %[m2], %[m2], %[mlo]\n\t"
        "v_cmpx_le_u32_e32 %[p3], %[t]\n\t"
        "v_add_f32_e32 %[m3], %[m3], %[mlo]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_le_u32_e32 %[P1], %[d]\n\t"
        "v_add_f32_e32 %[m1], %[m1], %[mhi]\n\t"
        "v_cmpx_le_u32_e32 %[P2], %[d]\n\t"
        "v_add_f32_e32 %[m2], %[m2], %[mhi]\n\t"
        "v_cmpx_le_u32_e32 %[P3], %[d]\n\t"
        "v_add_f32_e32 %[m3], %[m3], %[mhi]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m1] "+v"(m1), [m2] "+v"(m2), [m3] "+v"(m3), [t] "=&v"(t), [sv] "=&s"(sv)
        : [d] "v"(d), [mlo] "v"(mlo), [mhi] "v"(mhi), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [P1] "s"(P1), [P2] "s"(P2), [P3] "s"(P3)
        : "vcc");
}
// the same for counts
__device__ __forceinline__ void count_ge3(unsigned d, unsigned p1, unsigned p2, unsigned p3, unsigned P1, unsigned P2, unsigned P3,
                                          unsigned &n1, unsigned &n2, unsigned &n3) {
    unsigned t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_and_b32_e32 %[t], 0xffff, %[d]\n\t"
        "v_cmpx_le_u32_e32 %[p1], %[t]\n\t"
        "v_add_u32_e32 %[n1], 1, %[n1]\n\t"
        "v_cmpx_le_u32_e32 %[p2], %[t]\n\t"
        "v_add_u32_e32 %[n2], 1, %[n2]\n\t"
        "v_cmpx_le_u32_e32 %[p3], %[t]\n\t"
        "v_add_u32_e32 %[n3], 1, %[n3]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_le_u32_e32 %[P1], %[d]\n\t"
        "v_add_u32_e32 %[n1], 1, %[n1]\n\t"
        "v_cmpx_le_u32_e32 %[P2], %[d]\n\t"
        "v_add_u32_e32 %[n2], 1, %[n2]\n\t"
        "v_cmpx_le_u32_e32 %[P3], %[d]\n\t"
        "v_add_u32_e32 %[n3], 1, %[n3]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [n1] "+v"(n1), [n2] "+v"(n2), [n3] "+v"(n3), [t] "=&v"(t), [sv] "=&s"(sv)
        : [d] "v"(d), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [P1] "s"(P1), [P2] "s"(P2), [P3] "s"(P3)
        : "vcc");
}

// SAMP_STOP (diagnostic builds only, `make ablate A=s<n> S=<n>`): the kernel ends after phase n -- 1 row load + max, 2 masses + Z,
// 3 top-p cutoff, 4 kept mass -- so that tools/sampler_bench.py under CHIRRUP_AMD_LIB gives the time up to there.
#ifndef SAMP_STOP
#define SAMP_STOP 0
#endif
#define SAMP_PHASE_END(n, v) if (SAMP_STOP == (n)) { if (tid == 0) ids[r] = (int32_t)(v); return; }
// SAMP_STOP 9: workgroup b reports the time of interval (b % 8) in 10 ns ticks instead of the token: 0 load + max, 1 masses + Z,
// 2 the compares of the eight cutoff passes, 3 their reductions, 4 top-k, 5 kept mass, 6 scan + draw, 7 everything
#define SAMP_STAMP(i) if (SAMP_STOP == 9) st[i] = wall_clock64();

__global__ __launch_bounds__(kSampThreads) void sample_topp_kernel(
    const int V, const f16 *__restrict__ logits, const int32_t *__restrict__ rows, const f16 *__restrict__ temperature,
    const f16 *__restrict__ top_p, const int32_t *__restrict__ top_k, const int32_t *__restrict__ slot_idx,
    const float *__restrict__ uniform, int32_t *__restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16 *row = reinterpret_cast<f16 *>(smem);                                   // V halves (padded to 8)
    float *redm = reinterpret_cast<float *>(smem + ((size_t)V * 2 + 15) / 16 * 16);    // [3][16] masses of a pass, per wave
    unsigned *redn = reinterpret_cast<unsigned *>(redm + kSampRed);              // [3][16] counts
    float *red = reinterpret_cast<float *>(redn + kSampRed);                     // [16]
    float *scal = red + 16;                                                      // small broadcast area
    const int tid = threadIdx.x;
    const int r = rows[blockIdx.x];
    const int slot = slot_idx ? slot_idx[r] : r;
    const float T = (float)temperature[slot];
    const float P = (float)top_p[slot];
    const int Kk = top_k[slot];
    const f16 *src = logits + (int64_t)r * V;
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_cmp = 0, t_red = 0;
    SAMP_STAMP(0)

    // ---- row -> LDS (for the token-order draw at the end) AND into registers: this lane's 64 tokens, eight groups of eight,
    //      group g = tid + 1024 it; max
    constexpr int kGroups = 65536 / kSampThreads / 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned kq[4 * kGroups];                                 // raw binary16 pairs first, order-preserving 16-bit keys after the max
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < kGroups; it++) {
        const int c = 8 * (tid + it * kSampThreads);
        if (c < V) {
            const f16x8 v = *reinterpret_cast<const f16x8 *>(src + c);
            *reinterpret_cast<f16x8 *>(row + c) = v;
#pragma unroll
            for (int e = 0; e < 8; e++) mx = fmaxf(mx, (float)v[e]);
            const u32x4 q = __builtin_bit_cast(u32x4, v);
#pragma unroll
            for (int e = 0; e < 4; e++) kq[4 * it + e] = q[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) kq[4 * it + e] = 0xfc00fc00u;          // -inf: mass 0, the smallest key
        }
    }
    mx = block_reduce_max(mx, red);
    const unsigned Kmax = key_of((f16)mx);             // the row's largest key (mx is one of its binary16 values)
    SAMP_PHASE_END(1, Kmax)
    SAMP_STAMP(1)
    float ms[8 * kGroups];                             // un-normalised mass per token
    float z = 0.f;
#pragma unroll
    for (int it = 0; it < kGroups; it++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned d = kq[4 * it + e];
            const f16 lo = __builtin_bit_cast(f16, (unsigned short)(d & 0xffffu)), hi = __builtin_bit_cast(f16, (unsigned short)(d >> 16));
            const float m0 = __expf((float)lo - mx), m1 = __expf((float)hi - mx);
            ms[8 * it + 2 * e] = m0, ms[8 * it + 2 * e + 1] = m1;
            z += m0 + m1;
            kq[4 * it + e] = 8 * (tid + it * kSampThreads) < V ? key_of(lo) | (key_of(hi) << 16) : 0u;   // past the row: key 0, below every threshold
        }
    }
    const float Z0 = block_reduce_sum(z, red);
    const float target = P * Z0;                 // in un-normalised mass
    SAMP_PHASE_END(2, target)
    SAMP_STAMP(2)

    // ---- the cutoffs, two key bits per pass: c_key = the largest OCCURRING key with mass{key >= c_key} >= target, k_key = the
    //      largest key with count{key >= k_key} >= K (top-k).  Eight passes over registers only, one block reduction each.
    //      (Rounds 1-2 built two-level histograms in LDS with atomics: with real logits most tokens fall into a few bins, the
    //      same-address atomics serialise -- 167 us for 200 rows; one bit per pass with keys re-derived from LDS: 125-142 us.)
    unsigned c_key = 0, k_key = 0;
    for (int bit = 14; bit >= 0; bit -= 2) {
        const unsigned p1 = c_key | (1u << bit), p2 = c_key | (2u << bit), p3 = c_key | (3u << bit);
        float m1 = 0.f, m2 = 0.f, m3 = 0.f;
        SAMP_STAMP(6)
#pragma unroll
        for (int it = 0; it < kGroups; it++) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                mass_ge3(kq[4 * it + e], ms[8 * it + 2 * e], ms[8 * it + 2 * e + 1], p1, p2, p3, p1 << 16, p2 << 16, p3 << 16, m1, m2, m3);
        }
        SAMP_STAMP(7)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            m1 += __shfl_xor(m1, o, 64), m2 += __shfl_xor(m2, o, 64), m3 += __shfl_xor(m3, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) redm[tid >> 6] = m1, redm[16 + (tid >> 6)] = m2, redm[32 + (tid >> 6)] = m3;
        __syncthreads();
        float M1 = 0.f, M2 = 0.f, M3 = 0.f;
        for (int w = 0; w < kSampThreads / 64; w++) M1 += redm[w], M2 += redm[16 + w], M3 += redm[32 + w];
        // mass{key >= t} falls as t grows: the largest candidate that still holds the target, among keys that occur (t <= Kmax;
        // top_p = 0 then ends at the row's largest key, i.e. greedy)
        if (M3 >= target && p3 <= Kmax) c_key = p3;
        else if (M2 >= target && p2 <= Kmax) c_key = p2;
        else if (M1 >= target && p1 <= Kmax) c_key = p1;
        if (SAMP_STOP == 9) t_cmp += st[7] - st[6], t_red += wall_clock64() - st[7];
    }
    SAMP_PHASE_END(3, c_key)
    SAMP_STAMP(3)
    if (Kk > 0) {                                // top-k: the same search over counts
        for (int bit = 14; bit >= 0; bit -= 2) {
            const unsigned q1 = k_key | (1u << bit), q2 = k_key | (2u << bit), q3 = k_key | (3u << bit);
            unsigned n1 = 0, n2 = 0, n3 = 0;
#pragma unroll
            for (int it = 0; it < kGroups; it++) {
#pragma unroll
                for (int e = 0; e < 4; e++) count_ge3(kq[4 * it + e], q1, q2, q3, q1 << 16, q2 << 16, q3 << 16, n1, n2, n3);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
                n1 += __shfl_xor(n1, o, 64), n2 += __shfl_xor(n2, o, 64), n3 += __shfl_xor(n3, o, 64);
            __syncthreads();
            if ((tid & 63) == 0) redn[tid >> 6] = n1, redn[16 + (tid >> 6)] = n2, redn[32 + (tid >> 6)] = n3;
            __syncthreads();
            unsigned N1 = 0, N2 = 0, N3 = 0;
            for (int w = 0; w < kSampThreads / 64; w++) N1 += redn[w], N2 += redn[16 + w], N3 += redn[32 + w];
            if (N3 >= (unsigned)Kk) k_key = q3;
            else if (N2 >= (unsigned)Kk) k_key = q2;
            else if (N1 >= (unsigned)Kk) k_key = q1;
        }
    }
    const unsigned thr = (Kk > 0 && k_key > c_key) ? k_key : c_key;
    SAMP_STAMP(4)

    // ---- kept mass after temperature, then inverse-CDF draw in token order
    // p^(1/T) with p = exp(v - mx) / Z0 is exp((v - mx - ln Z0) / T): one exponential per kept token (powf here cost 80 us per
    // launch at top_p 0.9)
    const float invT = 1.0f / T;
    const bool hot = T != 1.0f;
    const float lnZ = __logf(Z0);
    auto weight = [&](float v) -> float { return hot ? __expf((v - mx - lnZ) * invT) : __expf(v - mx) / Z0; };
    const int per = (V + kSampThreads - 1) / kSampThreads;       // contiguous chunk per thread
    const int c0 = tid * per, c1 = (c0 + per) < V ? (c0 + per) : V;
    float local = 0.f;
    if ((per & 7) == 0 && c0 + per <= V) {             // the usual vocabulary: 16-byte LDS reads (64 two-byte reads at a 128-byte lane
                                                       // stride are 32-way bank conflicts each)
        for (int c = c0; c < c1; c += 8) {
            const f16x8 v8 = *reinterpret_cast<const f16x8 *>(row + c);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (key_of(v8[e]) >= thr) local += weight((float)v8[e]);
            }
        }
    } else {
        for (int c = c0; c < c1; c++) {
            if (key_of(row[c]) >= thr) local += weight((float)row[c]);
        }
    }
    SAMP_PHASE_END(4, __shfl_xor(local, 1, 64))
    SAMP_STAMP(5)
    // exclusive scan of `local` over the 1024 threads
    float incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += t;
    }
    __syncthreads();
    if ((tid & 63) == 63) red[tid >> 6] = incl;
    __syncthreads();
    float wave_base = 0.f, total = 0.f;
    for (int i = 0; i < kSampThreads / 64; i++) { if (i < (tid >> 6)) wave_base += red[i]; total += red[i]; }
    const float before = wave_base + incl - local;
    const float want = uniform[blockIdx.x] * total;
    if (tid == 0) scal[4] = -1.f;
    __syncthreads();
    if (local > 0.f && want >= before && want < before + local) {
        float acc = before; int pick = -1;
        for (int c = c0; c < c1; c++) {
            if (key_of(row[c]) >= thr) {
                acc += weight((float)row[c]);
                pick = c;
                if (acc > want) break;
            }
        }
        scal[4] = (float)pick;           // exactly one thread's interval contains `want`
    }
    __syncthreads();
    if (tid == 0) {
        int pick = (int)scal[4];
        if (pick < 0) {                  // want == total (u -> 1) or rounding: take the last kept token
            for (int c = V - 1; c >= 0; c--) if (key_of(row[c]) >= thr) { pick = c; break; }
        }
        ids[r] = pick;
        if (SAMP_STOP == 9) {
            const unsigned long long t_end = wall_clock64();
            const unsigned long long d[8] = {st[1] - st[0], st[2] - st[1], t_cmp, t_red, st[4] - st[3], st[5] - st[4], t_end - st[5], t_end - st[0]};
            ids[r] = (int32_t)d[blockIdx.x & 7];
        }
    }
}

}  // namespace

extern "C" int rwkv7_sample_topp(int n_rows, int V, const void *logits, const int32_t *rows, const void *temperature,
                                 const void *top_p, const int32_t *top_k, const int32_t *slot_idx, const float *uniform,
                                 int32_t *ids, void *stream) {
    if (n_rows <= 0 || V <= 0 || (V & 7) || V > 65536) return CHIRRUP_E_SHAPE;
    if (!logits || !rows || !temperature || !top_p || !top_k || !uniform || !ids) return CHIRRUP_E_NULL;
    if (reinterpret_cast<uintptr_t>(logits) & 15) return CHIRRUP_E_ALIGN;
    const size_t lds = ((size_t)V * 2 + 15) / 16 * 16 + (size_t)kSampRed * 8 + 16 * 4 + 8 * 4;
    static std::atomic<bool> attr_set[32];    // per device: one engine process may drive several GPUs from several threads
    int dev = 0;                              // (the attribute call is idempotent; the flag only saves repeating it)
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 31].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sample_topp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev & 31].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(sample_topp_kernel, dim3((unsigned)n_rows), dim3(kSampThreads), lds, static_cast<hipStream_t>(stream), V,
                       static_cast<const f16 *>(logits), rows, static_cast<const f16 *>(temperature),
                       static_cast<const f16 *>(top_p), top_k, slot_idx, uniform, ids);
    return (int)hipGetLastError();
}
