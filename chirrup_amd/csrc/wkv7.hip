// WKV7 recurrent state update for gfx950 (MI355X) -- hand-written, HBM-bound.
//
// What it computes: spec A1 of SURVEY.md section 8, i.e. the arithmetic of the reference's
// kernel_forward_w0_fp16_dither_seq / _one (Albatross/cuda/rwkv7_state_fwd_fp16.cu:26-167):
// binary16 storage and binary16 accumulate, two-lane (even j / odd j) summation order, one
// rounding per operation, no FMA contraction.  Results are bit-identical to oracle/oracle.c.
//
// How it is laid out for CDNA4 (not a translation of the reference's 64-thread CUDA block):
//   * one 64-lane wavefront per (slot, head); the workgroup IS the wave, so no barrier is ever
//     waited on and up to 18 independent waves per CU keep ~144 KiB of HBM traffic in flight;
//   * the 8 KiB head state goes HBM -> LDS with eight 1-KiB LDS-DMA instructions
//     (global_load_lds_dwordx4, no VGPR staging).  Each DMA reads 8 whole 128-B state rows
//     (fully coalesced); the 16-B chunk order inside a row is permuted on the per-lane SOURCE
//     address so that the row-per-lane ds_read_b128 that follows is bank-conflict free
//     (chunk c of row i sits at i*128 + ((c ^ ((i>>1)&7))<<4));
//   * lane i owns state row i in 32 packed-half VGPRs for the whole T loop; r,w~,k,a,b of
//     the head are written once to a 640-B LDS strip and read back as wave-uniform
//     (broadcast) ds_read_b128, which is what makes the per-row dot products "reductions
//     over a wavefront-shared vector" without a single cross-lane shuffle on the hot path;
//   * the updated rows return through the same swizzled LDS image and leave as eight
//     coalesced 1-KiB global_store_dwordx4.
// Algorithmic HBM bytes per (slot, layer, token): 270*C + 4.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

// Albatross/cuda/rwkv7_state_fwd_fp16.cu:20-22
constexpr float kTwoToNeg41 = 4.547473508864641e-13f;
constexpr float kNexpHalfLog2e = -0.8750387749145276f;
constexpr float kNlog2e = -1.4426950408889634f;
constexpr uint32_t kRo1 = 2654435769u;

#ifndef WKV7_LOAD_AUX
#define WKV7_LOAD_AUX 2       // aux bits of the LDS-DMA state loads: nt (state is touched once per step)
#endif
#ifndef WKV7_NT_STORE
#define WKV7_NT_STORE 1       // non-temporal stores for the updated state (A/B: tools/ab_wkv7.py, +15 % with both)
#endif
#ifndef WKV7_MIN_WAVES
#define WKV7_MIN_WAVES 1      // __launch_bounds__ second argument (waves per SIMD)
#endif
#ifndef WKV7_FUSED_WAVES
#define WKV7_FUSED_WAVES 4    // the fused decode forms: 4 waves per SIMD (the 8.8-KB strip per wave allows 4.5); MODE 2 takes 132 registers by itself
#endif
#ifndef WKV7_ENTRY_SUFFIX
#define WKV7_ENTRY_SUFFIX
#endif
#define WKV7_CAT2(a, b) a##b
#define WKV7_CAT(a, b) WKV7_CAT2(a, b)

constexpr int kStateBytes = 64 * 64 * 2;  // one head
constexpr int kVecBytes = 64 * 2;         // one 64-channel vector

#pragma clang fp contract(off)

// exp2f of the reference evaluated as the correctly rounded binary32 value (binary64 exp2,
// rounded once) so that host oracle and device agree bit for bit; costs two f64 exp2 per
// channel per token, which hides under the 16 KiB of HBM traffic per head.
__device__ __forceinline__ float exp2f_cr(float x) { return (float)exp2((double)x); }

// w~ = exp(-e^-0.5 * sigmoid(w)) - 1 + dither   (.cu:59), binary32, one rounding per op.
__device__ __forceinline__ f16 decay_term(f16 w_raw, float dither) {
    const float e1 = exp2f_cr(kNlog2e * (float)w_raw);
    const float q = kNexpHalfLog2e / (1.0f + e1);
    const float e2 = exp2f_cr(q);
    return (f16)((e2 - 1.0f) + dither);
}

__device__ __forceinline__ f16 hf(float x) { return (f16)x; }          // one rounding to binary16
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float wave_sum(float v) {                  // all 64 lanes get the total
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Extra operands of the fused time-mix form (MODE > 0): the gating chain of rwkv7.py:629-637 in front of
// the state update and the group-norm / bonus / gate chain of :647-649 behind it run inside this kernel.
struct TmixArgs {
    const f16 *a_pre;      // [B,T,C]  a0 + (xa@a1)@a2, before the sigmoid
    const f16 *vg_pre;     // [B,T,C]  v0 + (xv@v1)@v2, before the sigmoid (layers > 0), else nullptr
    const f16 *v_first;    // [B,T,C]  layer 0's v (layers > 0), else nullptr
    const f16 *g;          // [B,T,C]
    const f16 *k_k, *k_a, *r_k, *lnx_w, *lnx_b;   // [C]
    float eps;
    // mm8 (w8a16) att.output: the kernel's output o feeds a uint8 GEMM in the split form of scripts/test_mm8/benchmark.py:
    // 167-179, whose activation prologue is xs = binary16(o * ry) and the row sums {sum xs, sum o*my, sum o}.  With q_ry set the
    // kernel stores xs INSTEAD of o and this head's share of the three sums: q_S[row][head][3] (the consumer adds the H parts
    // in head order -- rwkv7_add_ln_mix_mm8, in_S_parts = H).
    const f16 *q_ry, *q_my;    // [C] or nullptr
    float *q_S;                // [B*T][H][3]
};

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// MODE 0: the operator of the reference (inputs a = -kk, b = kk*a given).  MODE 1 (MODE 2: + the mm8 prologue of TmixArgs): fused time-mix core:
// k, v are the RAW projections, a_/b_ are unused, y_ receives (group_norm(y) + bonus*v) * g.
// DECAYED (MODE 0 only): w_ already holds w~ (wkv7_decay_kernel below) -- for chunks of many tokens everything that does
// not depend on the state is computed by row-parallel kernels (rwkv7_tmix_mid, wkv7_decay, rwkv7_tmix_post) and the
// sequential loop keeps only the recurrence.  The loop is VALU-issue bound (one wave per head, ~530 instructions per
// token of which 320 are the packed-half multiplies and adds that the one-rounding-per-operation contract keeps
// apart): the two binary64 exp2 per channel and token of decay_term and the gating / group-norm arithmetic of MODE 1 are
// a third of it, and no load latency is exposed (eight tokens of prefetch instead of one: slower, the code no longer
// fits the instruction cache -- profiles/r02_prefill_B25_T100.txt).
template <int MODE, bool DECAYED = false>
__global__ __launch_bounds__(64, (MODE >= 1 && WKV7_FUSED_WAVES) ? WKV7_FUSED_WAVES : WKV7_MIN_WAVES) void wkv7_seq_kernel(
    const int T, const int C, const int H, f16 *__restrict__ state, const int64_t slot_stride,
    const int32_t *__restrict__ slot_idx, const f16 *__restrict__ r_, const f16 *__restrict__ w_,
    const f16 *__restrict__ k_, const f16 *__restrict__ v_, const f16 *__restrict__ a_,
    const f16 *__restrict__ b_, f16 *__restrict__ y_, const int32_t *__restrict__ elapsed_t, const TmixArgs tm) {
    // [0, 8192): swizzled state image; then r, w~, k, a, b strips of 128 B each.
    // (the chunk scan -- MODE 0, DECAYED -- keeps TWO strips: token t + 1's vectors are written while token t's are read)
    constexpr bool kTwoStrips = MODE == 0 && DECAYED;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kStateBytes + (kTwoStrips ? 10 : 5) * kVecBytes];

    const int bb = blockIdx.x / H;
    const int h = blockIdx.x - bb * H;
    const int lane = threadIdx.x;
    const int64_t slot = slot_idx ? (int64_t)slot_idx[bb] : (int64_t)bb;
    unsigned char *gS = reinterpret_cast<unsigned char *>(state + slot * slot_stride + (int64_t)h * 4096);

    // ---- state: HBM -> LDS, 8 x 1 KiB LDS-DMA, chunk order permuted on the source side ----
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int row = q * 8 + (lane >> 3);
        const int lch = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(gS + row * 128 + lch * 16), (lptr_t)(smem + q * 1024), 16, 0, WKV7_LOAD_AUX);
    }

    // ---- first timestep's vectors (lane j holds channel j of the head) ----
    int64_t o = (int64_t)bb * T * C + (int64_t)h * 64 + lane;
    f16 rj = r_[o], wj = w_[o], kj = k_[o], vi = v_[o];
    f16 aj, bj, gj = (f16)0.f, vgj = (f16)0.f, vfj = (f16)0.f;     // MODE 1: aj carries a_pre
    f16 p_kk = (f16)0.f, p_ka = (f16)0.f, p_rk = (f16)0.f, p_lw = (f16)0.f, p_lb = (f16)0.f, p_ry = (f16)0.f, p_my = (f16)0.f;
    if (MODE == 0) {
        aj = a_[o];
        bj = b_[o];
    } else {
        const int ch = h * 64 + lane;
        aj = tm.a_pre[o];
        bj = (f16)0.f;
        gj = tm.g[o];
        if (tm.v_first) { vgj = tm.vg_pre[o]; vfj = tm.v_first[o]; }
        p_kk = tm.k_k[ch]; p_ka = tm.k_a[ch]; p_rk = tm.r_k[ch]; p_lw = tm.lnx_w[ch]; p_lb = tm.lnx_b[ch];
        if (MODE == 2) { p_ry = tm.q_ry[ch]; p_my = tm.q_my[ch]; }
    }
    const int32_t et = elapsed_t[bb];

    // The DMA is older than every load above, and vmcnt retires in order, but make the
    // LDS image's readiness explicit rather than implied by the compiler's counted waits.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- row `lane` of the head into 32 packed-half registers (conflict-free b128 reads) ----
    f16x2 S[32];
    const int sw = (lane >> 1) & 7;
    unsigned char *myrow = smem + lane * 128;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const f16x8 q8 = *reinterpret_cast<const f16x8 *>(myrow + ((c ^ sw) << 4));
        S[4 * c + 0] = q8.s01;
        S[4 * c + 1] = q8.s23;
        S[4 * c + 2] = q8.s45;
        S[4 * c + 3] = q8.s67;
    }

    f16 *vec = reinterpret_cast<f16 *>(smem + kStateBytes);
    const f16x2 *R2 = reinterpret_cast<const f16x2 *>(smem + kStateBytes + 0 * kVecBytes);
    const f16x2 *W2 = reinterpret_cast<const f16x2 *>(smem + kStateBytes + 1 * kVecBytes);
    const f16x2 *K2 = reinterpret_cast<const f16x2 *>(smem + kStateBytes + 2 * kVecBytes);
    const f16x2 *A2 = reinterpret_cast<const f16x2 *>(smem + kStateBytes + 3 * kVecBytes);
    const f16x2 *B2 = reinterpret_cast<const f16x2 *>(smem + kStateBytes + 4 * kVecBytes);

    if constexpr (kTwoStrips) {
        // The chunk scan with few sequences is ONE wave per SIMD, and a lone wave exposes every latency of the token's chain
        // "vectors -> LDS strip -> broadcast reads -> two 32-long dependent sums" (1 x 256 tokens at 7.2B: 1.68 us per token, 65 % of the
        // chunk).  Two strips: token t + 1's vectors (in registers since the iteration before) are written while token t's are read,
        // and token t + 2's are requested -- the strip's write-to-read latency and the global loads leave the critical path.  Same
        // arithmetic on the same values in the same order: bit-exact with the one-strip loop (tests/test_wkv7_gpu.py).
        auto put = [&](int which, f16 r, f16 w, f16 k, f16 a, f16 b) {
            f16 *v5 = vec + which * 5 * 64;
            v5[0 * 64 + lane] = r, v5[1 * 64 + lane] = w, v5[2 * 64 + lane] = k, v5[3 * 64 + lane] = a, v5[4 * 64 + lane] = b;
        };
        put(0, rj, wj, kj, aj, bj);
        f16 v_cur = vi;
        f16 r1 = (f16)0.f, w1 = r1, k1 = r1, v1 = r1, a1 = r1, b1 = r1;      // token t + 1
        if (T > 1) {
            const int64_t o1 = o + C;
            r1 = r_[o1], w1 = w_[o1], k1 = k_[o1], v1 = v_[o1], a1 = a_[o1], b1 = b_[o1];
        }
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < T; t++) {
            f16 r2 = (f16)0.f, w2 = r2, k2 = r2, v2 = r2, a2 = r2, b2 = r2;  // token t + 2: requested now, used next iteration
            if (t + 2 < T) {
                const int64_t o2 = o + 2 * (int64_t)C;
                r2 = r_[o2], w2 = w_[o2], k2 = k_[o2], v2 = v_[o2], a2 = a_[o2], b2 = b_[o2];
            }
            // the whole strip of token t into registers in one burst of 40 reads (it was written an iteration ago: no store-to-load
            // wait), pinned in front of the arithmetic -- left to itself the scheduler trickles the reads in between the packed ops
            // with a counted wait in front of every group, and a lone wave pays each of those waits
            f16x2 pR[32], pW[32], pK[32], pA[32], pB[32];
            {
                const f16x8 *P8 = reinterpret_cast<const f16x8 *>(smem + kStateBytes + (t & 1) * 5 * kVecBytes);
                auto get = [&](f16x2 (&dst)[32], int which) {
#pragma unroll
                    for (int c8 = 0; c8 < 8; c8++) {
                        const f16x8 q8 = P8[which * 8 + c8];
                        dst[4 * c8 + 0] = q8.s01, dst[4 * c8 + 1] = q8.s23, dst[4 * c8 + 2] = q8.s45, dst[4 * c8 + 3] = q8.s67;
                    }
                };
                get(pA, 3), get(pW, 1), get(pK, 2), get(pB, 4), get(pR, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            f16x2 sa2 = {(f16)0.f, (f16)0.f};
#pragma unroll
            for (int p = 0; p < 32; p++) sa2 = sa2 + pA[p] * S[p];
            const f16 sa = sa2.x + sa2.y;
            const f16x2 sab = {sa, sa};
            const f16x2 vv2 = {v_cur, v_cur};
            if (t + 1 < T) put((t + 1) & 1, r1, w1, k1, a1, b1);             // (the other strip: nobody reads it in this iteration)
            f16x2 y2 = {(f16)0.f, (f16)0.f};
#pragma unroll
            for (int p = 0; p < 32; p++) {
                f16x2 sv = S[p];
                sv = sv + (sv * pW[p] + pK[p] * vv2 + sab * pB[p]);
                S[p] = sv;
                y2 = y2 + sv * pR[p];
            }
            y_[o] = y2.x + y2.y;
            __builtin_amdgcn_s_barrier();          // single-wave workgroup: orders the strips, never waits
            o += C;
            v_cur = v1;
            r1 = r2, w1 = w2, k1 = k2, v1 = v2, a1 = a2, b1 = b2;
        }
    } else
    for (int t = 0; t < T; t++) {
        // dither: int32 wrap-around multiply, int -> float, exact scale (.cu:23, :59)
        const float dither = kTwoToNeg41 * (float)(int32_t)(kRo1 * (uint32_t)(et + t));
        f16 k_in = kj, a_in = aj, b_in = bj, vv = vi;
        const f16 r_cur = rj, g_cur = gj;
        if (MODE >= 1) {
            // rwkv7.py:629-637, one rounding per torch op; the head's L2 norm is a wavefront reduction
            const float a = (float)hf(sigmoid_f((float)aj));
            const float kk_in = (float)hf((float)kj * (float)p_kk);
            float nrm = (float)hf(sqrtf(wave_sum(kk_in * kk_in)));
            nrm = nrm > 0.f ? nrm : 0.f;                    // clamp_min(1e-12) is 0 in binary16
            const float kk = (float)hf(kk_in / nrm);
            k_in = hf((float)kj * (float)hf(1.0f + (float)hf((float)hf(a - 1.0f) * (float)p_ka)));
            a_in = hf(-kk);
            b_in = hf(kk * a);
            if (tm.v_first) {
                const float gate = (float)hf(sigmoid_f((float)vgj));
                vv = hf((float)vi + (float)hf((float)hf((float)vfj - (float)vi) * gate));
            }
        }
        vec[0 * 64 + lane] = r_cur;
        vec[1 * 64 + lane] = DECAYED ? wj : decay_term(wj, dither);
        vec[2 * 64 + lane] = k_in;
        vec[3 * 64 + lane] = a_in;
        vec[4 * 64 + lane] = b_in;
        const int64_t o_cur = o;
        if (t + 1 < T) {  // prefetch the next timestep's vectors under this step's arithmetic
            o += C;
            rj = r_[o]; wj = w_[o]; kj = k_[o]; vi = v_[o];
            if (MODE == 0) {
                aj = a_[o]; bj = b_[o];
            } else {
                aj = tm.a_pre[o]; gj = tm.g[o];
                if (tm.v_first) { vgj = tm.vg_pre[o]; vfj = tm.v_first[o]; }
            }
        }
        __builtin_amdgcn_s_barrier();  // single-wave workgroup: orders the LDS strip, never waits

        // sa = sum_j a[j]*S[i][j]   (.cu:65-69)
        f16x2 sa2 = {(f16)0.f, (f16)0.f};
#pragma unroll
        for (int p = 0; p < 32; p++) sa2 = sa2 + A2[p] * S[p];
        const f16 sa = sa2.x + sa2.y;
        const f16x2 sab = {sa, sa};
        const f16x2 vv2 = {vv, vv};

        // S += S*w~ + k*v + sa*b ;  y = sum_j S[i][j]*r[j]   (.cu:72-81)
        f16x2 y2 = {(f16)0.f, (f16)0.f};
#pragma unroll
        for (int p = 0; p < 32; p++) {
            f16x2 s = S[p];
            s = s + (s * W2[p] + K2[p] * vv2 + sab * B2[p]);
            S[p] = s;
            y2 = y2 + s * R2[p];
        }
        const f16 yv = y2.x + y2.y;
        if (MODE == 0) {
            y_[o_cur] = yv;
        } else {
            // rwkv7.py:647-649: group_norm over the head (lane i holds y[i]), + (sum_j r*k*r_k) * v, * g
            const float yf = (float)yv;
            const float mean = wave_sum(yf) * (1.0f / 64.0f);
            const float dlt = yf - mean;
            const float rstd = 1.0f / sqrtf(wave_sum(dlt * dlt) * (1.0f / 64.0f) + tm.eps);
            const float gn = (float)hf(dlt * rstd * (float)p_lw + (float)p_lb);
            const float bonus = (float)hf(wave_sum((float)hf((float)hf((float)r_cur * (float)k_in) * (float)p_rk)));
            const f16 out = hf((float)hf(gn + (float)hf(bonus * (float)vv)) * (float)g_cur);
            if (MODE == 2) {                                // mm8 prologue of att.output (mm8_prep_kernel's arithmetic)
                const f16 xs = hf((float)out * (float)p_ry);
                const float s0 = wave_sum((float)xs), s1 = wave_sum((float)out * (float)p_my), s2 = wave_sum((float)out);
                y_[o_cur] = xs;
                if (lane == 0) {
                    float *dst = tm.q_S + (((int64_t)bb * T + t) * H + h) * 3;
                    dst[0] = s0, dst[1] = s1, dst[2] = s2;
                }
            } else {
                y_[o_cur] = out;
            }
        }
        __builtin_amdgcn_s_barrier();  // strip is rewritten next iteration
    }

    // ---- rows back to the swizzled image, then 8 coalesced 1-KiB stores ----
#pragma unroll
    for (int c = 0; c < 8; c++) {
        f16x8 q8;
        q8.s01 = S[4 * c + 0];
        q8.s23 = S[4 * c + 1];
        q8.s45 = S[4 * c + 2];
        q8.s67 = S[4 * c + 3];
        *reinterpret_cast<f16x8 *>(myrow + ((c ^ sw) << 4)) = q8;
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int row = q * 8 + (lane >> 3);
        const int lch = (lane & 7) ^ ((row >> 1) & 7);
        const f16x8 q8 = *reinterpret_cast<const f16x8 *>(smem + q * 1024 + lane * 16);
#if WKV7_NT_STORE
        __builtin_nontemporal_store(q8, reinterpret_cast<f16x8 *>(gS + row * 128 + lch * 16));
#else
        *reinterpret_cast<f16x8 *>(gS + row * 128 + lch * 16) = q8;
#endif
    }
}

int check_args(int B, int T, int C, int H, const void *state, const void *r, const void *w,
               const void *k, const void *v, const void *a, const void *b, const void *y,
               const int32_t *elapsed_t, int64_t slot_stride) {
    if (B <= 0 || T <= 0 || C <= 0 || H <= 0 || (int64_t)H * 64 != (int64_t)C) return CHIRRUP_E_SHAPE;
    if ((int64_t)B * H > 2147483647LL) return CHIRRUP_E_SHAPE;
    if (!state || !r || !w || !k || !v || !a || !b || !y || !elapsed_t) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(state) & 15) || (slot_stride & 7) || slot_stride < 0) return CHIRRUP_E_ALIGN;
    if (slot_stride != 0 && slot_stride < (int64_t)H * 4096) return CHIRRUP_E_SHAPE;
    return CHIRRUP_OK;
}

}  // namespace

extern "C" int WKV7_CAT(wkv7_fwd_seq, WKV7_ENTRY_SUFFIX)(int B, int T, int C, int H, void *state, const void *r, const void *w,
                            const void *k, const void *v, const void *a, const void *b, void *y,
                            const int32_t *elapsed_t, const int32_t *slot_idx, int64_t slot_stride,
                            void *stream) {
    const int rc = check_args(B, T, C, H, state, r, w, k, v, a, b, y, elapsed_t, slot_stride);
    if (rc != CHIRRUP_OK) return rc;
    if (slot_stride == 0) slot_stride = (int64_t)H * 4096;
    hipLaunchKernelGGL(wkv7_seq_kernel<0>, dim3((unsigned)(B * H)), dim3(64), 0, static_cast<hipStream_t>(stream), T, C, H,
                       static_cast<f16 *>(state), slot_stride, slot_idx, static_cast<const f16 *>(r),
                       static_cast<const f16 *>(w), static_cast<const f16 *>(k), static_cast<const f16 *>(v),
                       static_cast<const f16 *>(a), static_cast<const f16 *>(b), static_cast<f16 *>(y), elapsed_t, TmixArgs{});
    return (int)hipGetLastError();
}

#ifndef WKV7_VARIANT_BUILD
namespace {
// w~[b][t][c] = decay_term(w[b][t][c], dither(elapsed_t[b] + t)): the per-token decay of the WKV7 update (.cu:23, :59),
// the same device function the scan kernel evaluates in its loop, over all rows at once.
__global__ __launch_bounds__(256) void wkv7_decay_kernel(const int64_t nchunks, const int T, const int C, const f16 *__restrict__ w,
                                                         const int32_t *__restrict__ elapsed_t, f16 *__restrict__ w_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nchunks) return;
    const int64_t row = g * 8 / C;
    const int bb = (int)(row / T), t = (int)(row - (int64_t)bb * T);
    const float dither = kTwoToNeg41 * (float)(int32_t)(kRo1 * (uint32_t)(elapsed_t[bb] + t));
    const f16x8 wv = *reinterpret_cast<const f16x8 *>(w + g * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = decay_term(wv[e], dither);
    *reinterpret_cast<f16x8 *>(w_out + g * 8) = o;
}
}  // namespace

extern "C" int wkv7_decay(int B, int T, int C, const void *w, const int32_t *elapsed_t, void *w_out, void *stream) {
    if (B <= 0 || T <= 0 || C <= 0 || (C & 7)) return CHIRRUP_E_SHAPE;
    if (!w || !elapsed_t || !w_out) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(w_out) & 15)) return CHIRRUP_E_ALIGN;
    const int64_t nchunks = (int64_t)B * T * C / 8;
    hipLaunchKernelGGL(wkv7_decay_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       nchunks, T, C, static_cast<const f16 *>(w), elapsed_t, static_cast<f16 *>(w_out));
    return (int)hipGetLastError();
}

// wkv7_fwd_seq with the decay already applied to w (w = wkv7_decay(w_raw)): bit-identical results, the T loop without
// the transcendental work.
extern "C" int wkv7_fwd_seq_decayed(int B, int T, int C, int H, void *state, const void *r, const void *w_decayed, const void *k,
                                    const void *v, const void *a, const void *b, void *y, const int32_t *elapsed_t,
                                    const int32_t *slot_idx, int64_t slot_stride, void *stream) {
    const int rc = check_args(B, T, C, H, state, r, w_decayed, k, v, a, b, y, elapsed_t, slot_stride);
    if (rc != CHIRRUP_OK) return rc;
    if (slot_stride == 0) slot_stride = (int64_t)H * 4096;
    hipLaunchKernelGGL((wkv7_seq_kernel<0, true>), dim3((unsigned)(B * H)), dim3(64), 0, static_cast<hipStream_t>(stream), T, C, H,
                       static_cast<f16 *>(state), slot_stride, slot_idx, static_cast<const f16 *>(r),
                       static_cast<const f16 *>(w_decayed), static_cast<const f16 *>(k), static_cast<const f16 *>(v),
                       static_cast<const f16 *>(a), static_cast<const f16 *>(b), static_cast<f16 *>(y), elapsed_t, TmixArgs{});
    return (int)hipGetLastError();
}

// Fused time-mix core: gating (rwkv7.py:629-637) + WKV7 (:645) + group-norm / bonus / gate (:647-649).
extern "C" int rwkv7_tmix_wkv7_fused(int B, int T, int C, int H, void *state, const void *r, const void *w,
                                     const void *k, const void *v, const void *a_pre, const void *vg_pre,
                                     const void *v_first, const void *g, const void *k_k, const void *k_a,
                                     const void *r_k, const void *lnx_w, const void *lnx_b, float eps, void *out,
                                     const int32_t *elapsed_t, const int32_t *slot_idx, int64_t slot_stride,
                                     void *stream) {
    return rwkv7_tmix_wkv7_fused_mm8(B, T, C, H, state, r, w, k, v, a_pre, vg_pre, v_first, g, k_k, k_a, r_k, lnx_w, lnx_b, eps, out,
                                     elapsed_t, slot_idx, slot_stride, nullptr, nullptr, nullptr, stream);
}

// ... with the mm8 activation prologue of the GEMM that consumes `out` (att.output as uint8 weights): out receives
// xs = binary16(o * ry), S [B*T][H][3] this head's shares of {sum xs, sum o*my, sum o}.  ry == NULL: the plain form.
extern "C" int rwkv7_tmix_wkv7_fused_mm8(int B, int T, int C, int H, void *state, const void *r, const void *w,
                                         const void *k, const void *v, const void *a_pre, const void *vg_pre,
                                         const void *v_first, const void *g, const void *k_k, const void *k_a,
                                         const void *r_k, const void *lnx_w, const void *lnx_b, float eps, void *out,
                                         const int32_t *elapsed_t, const int32_t *slot_idx, int64_t slot_stride,
                                         const void *ry, const void *my, float *S, void *stream) {
    const int rc = check_args(B, T, C, H, state, r, w, k, v, a_pre, g, out, elapsed_t, slot_stride);
    if (rc != CHIRRUP_OK) return rc;
    if (!k_k || !k_a || !r_k || !lnx_w || !lnx_b) return CHIRRUP_E_NULL;
    if ((vg_pre == nullptr) != (v_first == nullptr)) return CHIRRUP_E_NULL;
    if (ry && (!my || !S)) return CHIRRUP_E_NULL;
    if (slot_stride == 0) slot_stride = (int64_t)H * 4096;
    TmixArgs tm{static_cast<const f16 *>(a_pre), static_cast<const f16 *>(vg_pre), static_cast<const f16 *>(v_first),
                static_cast<const f16 *>(g), static_cast<const f16 *>(k_k), static_cast<const f16 *>(k_a),
                static_cast<const f16 *>(r_k), static_cast<const f16 *>(lnx_w), static_cast<const f16 *>(lnx_b), eps,
                static_cast<const f16 *>(ry), static_cast<const f16 *>(my), S};
    // (the prologue is a template form of its own: as a run-time branch it cost the binary16 form 4 registers past 128, i.e. a wave per SIMD)
    if (ry)
        hipLaunchKernelGGL(wkv7_seq_kernel<2>, dim3((unsigned)(B * H)), dim3(64), 0, static_cast<hipStream_t>(stream), T, C, H,
                           static_cast<f16 *>(state), slot_stride, slot_idx, static_cast<const f16 *>(r),
                           static_cast<const f16 *>(w), static_cast<const f16 *>(k), static_cast<const f16 *>(v), nullptr, nullptr,
                           static_cast<f16 *>(out), elapsed_t, tm);
    else
        hipLaunchKernelGGL(wkv7_seq_kernel<1>, dim3((unsigned)(B * H)), dim3(64), 0, static_cast<hipStream_t>(stream), T, C, H,
                           static_cast<f16 *>(state), slot_stride, slot_idx, static_cast<const f16 *>(r),
                           static_cast<const f16 *>(w), static_cast<const f16 *>(k), static_cast<const f16 *>(v), nullptr, nullptr,
                           static_cast<f16 *>(out), elapsed_t, tm);
    return (int)hipGetLastError();
}
#endif

#ifndef WKV7_VARIANT_BUILD
extern "C" int wkv7_fwd_one(int B, int C, int H, void *state, const void *r, const void *w, const void *k,
                            const void *v, const void *a, const void *b, void *y, const int32_t *elapsed_t,
                            const int32_t *slot_idx, int64_t slot_stride, void *stream) {
    return wkv7_fwd_seq(B, 1, C, H, state, r, w, k, v, a, b, y, elapsed_t, slot_idx, slot_stride, stream);
}
#endif
