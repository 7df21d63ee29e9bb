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
                                                             const int64_t presence_stride) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const int id = ids[row];
    const int64_t slot = slot_idx ? (int64_t)slot_idx[row] : (int64_t)row;
    last_ids[slot] = id;
    if (id < 0 || id >= V) return;                     // (cannot be produced by the samplers; the tables are left alone)
    atomicAdd(occurrence + slot * V + id, penalty_weight[id]);     // (slots of a batch are distinct; the atomic keeps accumulate=True's meaning if not)
    alpha[slot * V + id] = presence[slot * presence_stride];
}
}  // namespace

extern "C" int rwkv7_commit_sampled(int n, int V, const int32_t *ids, const int32_t *slot_idx, int32_t *last_ids, float *occurrence,
                                    const float *penalty_weight, float *alpha_presence, const float *presence,
                                    int64_t presence_stride, void *stream) {
    if (n <= 0 || V <= 0 || presence_stride < 0) return CHIRRUP_E_SHAPE;
    if (!ids || !last_ids || !occurrence || !penalty_weight || !alpha_presence || !presence) return CHIRRUP_E_NULL;
    hipLaunchKernelGGL(commit_sampled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, V, ids,
                       slot_idx, last_ids, occurrence, penalty_weight, alpha_presence, presence, presence_stride);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Sort-free top-p / top-k / temperature sampling for the rows that are NOT greedy.
//
// Semantics of sample_logits_rwkv_pip_compatible (chirrup/utils/samplers.py:171-255): probs =
// softmax(logits); cutoff = the probability at which the DESCENDING cumulative sum first reaches top_p,
// everything below it is dropped (ties at the cutoff stay); optional top-k; probs ** (1/T); one draw
// from the remaining mass.  The reference sorts all V probabilities per row; here the row's binary16
// logits sit in LDS (V <= 65536 -> 128 KiB) and the cutoff is found by a two-level radix search over
// the 16-bit order-preserving keys of the logits (probability is monotone in the logit), 256 bins of
// probability mass (and of counts, for top-k) per level.  The draw is an inverse-CDF walk in token
// order with a caller-supplied uniform number per row (torch's generator stays the source of
// randomness).  Differences from the reference: ties AT the top-k boundary are all kept, and the
// draw uses one uniform instead of torch.multinomial's stream -- same distribution, different ids.
namespace {

constexpr int kSampThreads = 1024;
constexpr int kHistReplicas = 8;

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

__global__ __launch_bounds__(kSampThreads) void sample_topp_kernel(
    const int V, const f16 *__restrict__ logits, const int32_t *__restrict__ rows, const f16 *__restrict__ temperature,
    const f16 *__restrict__ top_p, const int32_t *__restrict__ top_k, const int32_t *__restrict__ slot_idx,
    const float *__restrict__ uniform, int32_t *__restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16 *row = reinterpret_cast<f16 *>(smem);                                   // V halves (padded to 8)
    float *hmass = reinterpret_cast<float *>(smem + ((size_t)V * 2 + 15) / 16 * 16);   // [replicas][256]
    unsigned *hcnt = reinterpret_cast<unsigned *>(hmass + kHistReplicas * 256);  // [replicas][256]
    float *red = reinterpret_cast<float *>(hcnt + kHistReplicas * 256);          // [16]
    float *scal = red + 16;                                                      // small broadcast area
    const int tid = threadIdx.x;
    const int r = rows[blockIdx.x];
    const int slot = slot_idx ? slot_idx[r] : r;
    const float T = (float)temperature[slot];
    const float P = (float)top_p[slot];
    const int Kk = top_k[slot];
    const f16 *src = logits + (int64_t)r * V;

    // ---- row -> LDS, max
    float mx = -INFINITY;
    for (int c = tid * 8; c < V; c += kSampThreads * 8) {
        const f16x8 v = *reinterpret_cast<const f16x8 *>(src + c);
        *reinterpret_cast<f16x8 *>(row + c) = v;
#pragma unroll
        for (int e = 0; e < 8; e++) mx = fmaxf(mx, (float)v[e]);
    }
    mx = block_reduce_max(mx, red);
    // this lane's elements -- the token pairs {2 (tid + 1024 i), +1} -- as un-normalised mass in registers; their 16-bit
    // order-preserving keys are re-derived from the row in LDS in every pass (two per 4-byte read; 64 masses AND 64 keys do not fit
    // the 128 registers a lane of a 1024-lane workgroup has)
    constexpr int kPairs = 65536 / kSampThreads / 2;
    typedef f16 f16x2_t __attribute__((ext_vector_type(2)));
    float ms[2 * kPairs];
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < kPairs; i++) {
        const int c = 2 * (tid + i * kSampThreads);
        float m0 = 0.f, m1 = 0.f;
        if (c < V) {                                 // (V is even: checked by the entry point)
            const f16x2_t v = *reinterpret_cast<const f16x2_t *>(row + c);
            m0 = __expf((float)v[0] - mx), m1 = __expf((float)v[1] - mx);
        }
        ms[2 * i] = m0, ms[2 * i + 1] = m1;
        z += m0 + m1;
    }
    const float Z0 = block_reduce_sum(z, red);
    const float target = P * Z0;                 // in un-normalised mass

    // ---- the cutoffs, by bisection over the key bits: c_key = the largest key with mass{key >= c_key} >= target, k_key = the
    //      largest key with count{key >= k_key} >= K (top-k).  16 passes, one block reduction each.  (Rounds 1-2 built two-level
    //      histograms in LDS with atomics: with real logits most tokens fall into a few bins, the same-address atomics serialise,
    //      and the kernel took 167 us for 200 rows.)
    unsigned c_key = 0, k_key = 0;
    const bool use_k = Kk > 0;
    unsigned *redu = hcnt;                       // [16]
    for (int bit = 15; bit >= 0; bit--) {
        const unsigned cp = c_key | (1u << bit), ck = k_key | (1u << bit);
        float m = 0.f;
        unsigned n = 0;                              // low half: tokens with key >= ck (top-k); high half: ... with key >= cp
#pragma unroll
        for (int i = 0; i < kPairs; i++) {
            const int c = 2 * (tid + i * kSampThreads);
            if (c < V) {
                const f16x2_t v = *reinterpret_cast<const f16x2_t *>(row + c);
                const unsigned k0 = key_of(v[0]), k1 = key_of(v[1]);
                m += (k0 >= cp ? ms[2 * i] : 0.f) + (k1 >= cp ? ms[2 * i + 1] : 0.f);
                n += (k0 >= cp ? 0x10000u : 0u) + (k1 >= cp ? 0x10000u : 0u);
                if (use_k) n += (k0 >= ck ? 1u : 0u) + (k1 >= ck ? 1u : 0u);
            }
        }
        const unsigned any_p = __any((n >> 16) != 0);        // (a count of 65536 would overflow the half: only "any" is asked of it)
        n &= 0xffffu;                                          // per lane at most 64
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            m += __shfl_xor(m, o, 64);
            n += __shfl_xor(n, o, 64);
        }
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = m, redu[tid >> 6] = n | (any_p ? 0x80000000u : 0u);
        __syncthreads();
        float M = 0.f;
        unsigned Nn = 0, anyp = 0;
        for (int w = 0; w < kSampThreads / 64; w++) M += red[w], Nn += redu[w] & 0x7fffffffu, anyp |= redu[w] >> 31;
        if (M >= target && anyp) c_key = cp;         // ... among the keys that occur (top_p = 0: the row's largest key, i.e. greedy)
        if (use_k && Nn >= (unsigned)Kk) k_key = ck;
    }
    const unsigned thr = (Kk > 0 && k_key > c_key) ? k_key : c_key;

    // ---- kept mass after temperature, then inverse-CDF draw in token order
    const float invT = 1.0f / T;
    const bool hot = T != 1.0f;
    const int per = (V + kSampThreads - 1) / kSampThreads;       // contiguous chunk per thread
    const int c0 = tid * per, c1 = (c0 + per) < V ? (c0 + per) : V;
    float local = 0.f;
    for (int c = c0; c < c1; c++) {
        if (key_of(row[c]) >= thr) {
            float p = __expf((float)row[c] - mx) / Z0;
            local += hot ? powf(p, invT) : p;
        }
    }
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
                float p = __expf((float)row[c] - mx) / Z0;
                acc += hot ? powf(p, invT) : p;
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
    }
}

}  // namespace

extern "C" int rwkv7_sample_topp(int n_rows, int V, const void *logits, const int32_t *rows, const void *temperature,
                                 const void *top_p, const int32_t *top_k, const int32_t *slot_idx, const float *uniform,
                                 int32_t *ids, void *stream) {
    if (n_rows <= 0 || V <= 0 || (V & 7) || V > 65536) return CHIRRUP_E_SHAPE;
    if (!logits || !rows || !temperature || !top_p || !top_k || !uniform || !ids) return CHIRRUP_E_NULL;
    if (reinterpret_cast<uintptr_t>(logits) & 15) return CHIRRUP_E_ALIGN;
    const size_t lds = ((size_t)V * 2 + 15) / 16 * 16 + (size_t)kHistReplicas * 256 * 8 + 16 * 4 + 8 * 4;
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
