// Fused element-wise / normalisation kernels around the WKV7 update and the projection GEMMs.
//
// The reference runs this work as ~40 separate torch kernels per layer (SURVEY.md section 8a, A4/A5:
// "launch-bound").  Here each chain between two GEMMs is ONE kernel.  Arithmetic keeps the
// reference's eager semantics: every torch op of the chain rounds to binary16 once, reductions
// (layer-norm, group-norm, L2 norm, head sums) accumulate in binary32 and round once -- so results
// match the unfused torch-op path except for the summation order inside a reduction.
//
//   add_ln_mix   x_new = x (+ delta);  cur = LN(x_new);  dx = prev - cur;  out[m] = cur + dx*mix[m]
//                (rwkv7.py:523 + :621-623 with 6 mix vectors; :533 + :675-677 with 1; :548-550 with 0)
//   tmix_mid     a = sigmoid(.), kk = normalize(k*k_k) per head, k *= 1+(a-1)*k_a, kka = kk*a,
//                v += (v_first - v)*sigmoid(.)                        (rwkv7.py:629-637)
//   tmix_post    group_norm(y) + (sum_head r*k*r_k)*v, times g         (rwkv7.py:647-649)
//   relu_sq      relu(k)**2                                             (rwkv7.py:678)
//
// Layout: rows of C binary16 channels, 16-byte (8-channel) accesses per lane everywhere; a 64-wide
// head is 8 consecutive lanes, so head reductions are 3 DPP/shuffle steps inside a wavefront.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16 h(float x) { return (f16)x; }                // one rounding to binary16
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float head_sum(float v) {   // 8 consecutive lanes = one 64-channel head
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// One workgroup per row, ONE 8-channel chunk per lane (a lane with four chunks keeps a quarter of the loads in flight: step 8.20 vs
// 8.00 ms in round 1): TH = 256 lanes for C <= 2048, 1024 above (512 is built for A/B).  With 256 or 512 lanes a lane has 512 / 256
// registers to its name instead of 128, and the kernel spends them on latency (HOIST): the LN weights, the lerp coefficients and
// all eight split-K planes are requested before the first reduction instead of after it -- a row is a chain of dependent
// round trips (row + planes -> mean -> variance -> LN weights -> lerp coefficients -> stores).  Worth 0.5 % of a step at C = 2048
// and nothing at C = 4096 (the host picks; rwkv7_add_ln_mix_mm8).
constexpr int kLnMaxThreads = 1024;
constexpr int kLnMaxChunks = 1;

template <int TH>
__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TH / 64; i++) t += red[i];
    return t;
}

// three sums at once (the mm8 prologue's S0, S1, S2): one pair of barriers instead of three
template <int TH>
__device__ __forceinline__ void block_sum3(float &a, float &b, float &c, float *red3) {
    a = wave_sum(a), b = wave_sum(b), c = wave_sum(c);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red3[wid] = a, red3[TH / 64 + wid] = b, red3[2 * (TH / 64) + wid] = c;
    __syncthreads();
    a = b = c = 0.f;
#pragma unroll
    for (int i = 0; i < TH / 64; i++) a += red3[i], b += red3[TH / 64 + i], c += red3[2 * (TH / 64) + i];
}

// Load row (x + delta) as binary16, return its layer-norm in `out` (binary16 values held as float).
// If x_out != nullptr the summed row is stored there.
// x and x_out may alias (the in-place residual update of a decode step), so neither is __restrict__.
template <int TH>
__device__ __forceinline__ void ln_row(const f16 *x, const f16 *__restrict__ delta, f16 *x_out,
                                       const f16 *__restrict__ w, const f16 *__restrict__ b, int C, float eps,
                                       float (&out)[kLnMaxChunks][8], float *red, const float *__restrict__ dpart = nullptr,
                                       int dsplits = 0, int64_t dsplit_stride = 0, const f16 *__restrict__ q_rx = nullptr,
                                       const f16 *__restrict__ q_mx = nullptr, const float *__restrict__ q_S = nullptr,
                                       const int q_parts = 0, float *q_sh = nullptr, unsigned long long *st = nullptr) {
    auto stamp = [&](int i) {
        if (st && threadIdx.x == 0) st[i] = __builtin_amdgcn_s_memrealtime();
    };
    // q_S: this row's mm8 row sums [q_parts][3] in global memory; they are added up (one wave per sum) into q_sh[3] AFTER the
    // loads of the row and of its first four partial planes have been issued, so that their latency is not a stage of its own
    constexpr bool HOIST = TH < kLnMaxThreads;
    constexpr int PL = HOIST ? 8 : 4;                  // split-K planes requested together
    const int nchunk = C >> 3;
    float vals[kLnMaxChunks][8];
    float s = 0.f;
    bool q_ready = false;
    f16x8 wv0 = {}, bv0 = {};                          // (HOIST) the LN weights of this lane's chunk, requested with the row
#pragma unroll
    for (int q = 0; q < kLnMaxChunks; q++) {
        const int c = threadIdx.x + q * TH;
        const bool in_row = c < nchunk;
        {
            f16x8 xv = {};
            if (in_row) xv = *reinterpret_cast<const f16x8 *>(x + c * 8);
            if (HOIST && in_row) wv0 = *reinterpret_cast<const f16x8 *>(w + c * 8), bv0 = *reinterpret_cast<const f16x8 *>(b + c * 8);
            if (dpart) {            // delta = binary16(sum of split-K partials): the GEMM's reduce folded into this prologue
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // PL planes' loads are issued before their adds (in plane order): four with 1024 lanes (eight would cost 32 of the
                // 128 registers a lane has there), all eight otherwise
                for (int s0 = 0; s0 < dsplits; s0 += PL) {
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    f32x4_t p[PL][2];
#pragma unroll
                    for (int u = 0; u < PL; u++) {
                        if (in_row && s0 + u < dsplits) {
                            p[u][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(dpart + (s0 + u) * dsplit_stride + c * 8));
                            p[u][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(dpart + (s0 + u) * dsplit_stride + c * 8 + 4));
                        }
                    }
                    if (q_S && !q_ready) {             // (uniform: every lane of the workgroup gets here, in_row or not)
                        static_assert(TH >= 3 * 64, "one wave per sum");
                        if (threadIdx.x < 3 * 64) {
                            const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
                            float tsum = 0.f;
                            for (int pp = lane; pp < q_parts; pp += 64) tsum += q_S[pp * 3 + g];
                            tsum = wave_sum(tsum);
                            if (lane == 0) q_sh[g] = tsum;
                        }
                        __syncthreads();
                        q_ready = true;
                    }
#pragma unroll
                    for (int u = 0; u < PL; u++) {
                        if (in_row && s0 + u < dsplits) {
                            acc[0] += p[u][0].x; acc[1] += p[u][0].y; acc[2] += p[u][0].z; acc[3] += p[u][0].w;
                            acc[4] += p[u][1].x; acc[5] += p[u][1].y; acc[6] += p[u][1].z; acc[7] += p[u][1].w;
                        }
                    }
                }
                if (q_S && in_row) {  // the partials are the core of an mm8 product taken against 1024 + q (skinny_gemm.hip: cvt_u8x2):
                                      // y = rx*(core - 1024*S0 + 0.5*S0) + S1 + mx*S2 (benchmark.py:167-179)
                    const f16x8 rxv = *reinterpret_cast<const f16x8 *>(q_rx + c * 8);
                    const f16x8 mxv = *reinterpret_cast<const f16x8 *>(q_mx + c * 8);
                    const float s0 = q_sh[0], s1 = q_sh[1], s2 = q_sh[2];
#pragma unroll
                    for (int e = 0; e < 8; e++) acc[e] = (float)rxv[e] * (acc[e] - 1023.5f * s0) + s1 + (float)mxv[e] * s2;
                }
#pragma unroll
                for (int e = 0; e < 8; e++) xv[e] = h((float)xv[e] + (float)h(acc[e]));
            } else if (delta && in_row) {
                const f16x8 dv = *reinterpret_cast<const f16x8 *>(delta + c * 8);
#pragma unroll
                for (int e = 0; e < 8; e++) xv[e] = h((float)xv[e] + (float)dv[e]);
            }
            if (in_row) {
                if (x_out) *reinterpret_cast<f16x8 *>(x_out + c * 8) = xv;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    vals[q][e] = (float)xv[e];
                    s += vals[q][e];
                }
            }
        }
    }
    stamp(1);                                          // lane 0's row data has arrived (its sums are formed)
    const float mean = block_sum<TH>(s, red) / (float)C;
    stamp(2);
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < kLnMaxChunks; q++) {
        const int c = threadIdx.x + q * TH;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d = vals[q][e] - mean;
                s2 += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(block_sum<TH>(s2, red) / (float)C + eps);
    stamp(3);
#pragma unroll
    for (int q = 0; q < kLnMaxChunks; q++) {
        const int c = threadIdx.x + q * TH;
        if (c < nchunk) {
            const f16x8 wv = HOIST ? wv0 : *reinterpret_cast<const f16x8 *>(w + c * 8);
            const f16x8 bv = HOIST ? bv0 : *reinterpret_cast<const f16x8 *>(b + c * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) out[q][e] = (float)h((vals[q][e] - mean) * rstd * (float)wv[e] + (float)bv[e]);
        }
    }
}

// One workgroup per R consecutive (b, t) rows of a sequence.
// PROBE: the diagnostic instantiation that writes phase stamps (the stamp code costs the 1024-lane forms 4-12 registers and a
// spill: 11.7 -> 13.4 us for LN1 at 7.2B / bsz 200 while it sat in the product kernel behind a run-time null check).
// Q: the mm8 (w8a16) hooks of chirrup_mm8_fuse are live; the binary16 model's launches instantiate the kernel without them.
template <int NMIX, int TH, bool PROBE = false, bool Q = true>
__global__ __launch_bounds__(TH) void add_ln_mix_kernel(
    const int T, const int C, const f16 *x, const f16 *__restrict__ delta, f16 *x_out,
    const f16 *__restrict__ ln_w, const f16 *__restrict__ ln_b, const float eps, const f16 *__restrict__ prev_in,
    f16 *__restrict__ prev_out, const f16 *__restrict__ mix, f16 *__restrict__ out, const int64_t out_stride,
    const int32_t *__restrict__ slot_idx, const float *__restrict__ dpart, const int dsplits, const int64_t dsplit_stride,
    const chirrup_mm8_fuse fz_, const int R, unsigned long long *stamps_) {
    unsigned long long *const stamps = PROBE ? stamps_ : nullptr;
    const chirrup_mm8_fuse fz = Q ? fz_ : chirrup_mm8_fuse{};
    // diagnostic (rwkv7_ln_probe; tools/ln_timeline.py): 100-MHz stamps of a workgroup's phases, 8 per workgroup
    auto stamp = [&](int i) {
        if (stamps && threadIdx.x == 0) stamps[(int64_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    constexpr bool HOIST = TH < kLnMaxThreads;
    __shared__ float red[3 * (TH / 64)];
    // A workgroup takes R consecutive tokens of one sequence: the token shift of row t needs LN(row t-1), which is the row
    // this workgroup normalised one trip earlier -- only the first of its rows recomputes its predecessor (R = 1, one row per
    // workgroup, recomputes it for every row with t > 0: twice the loads and arithmetic at T > 1).
    const int nb = (T + R - 1) / R;
    const int bb = blockIdx.x / nb, t0 = (blockIdx.x - bb * nb) * R;
    const int64_t slot = slot_idx ? (int64_t)slot_idx[bb] : (int64_t)bb;   // row of the carry tables
    const int nchunk = C >> 3;
    float cur[kLnMaxChunks][8], prev[kLnMaxChunks][8];
    const f16 *q_rx = static_cast<const f16 *>(fz.in_rx), *q_mx = static_cast<const f16 *>(fz.in_mx);
    __shared__ float qsum[2][3];                      // mm8 row sums of this row and of its predecessor (ln_row adds the parts up)
    const f16 *p_ry = static_cast<const f16 *>(fz.out_ry), *p_my = static_cast<const f16 *>(fz.out_my);
    f16 *p_xs = static_cast<f16 *>(fz.out_xs);
    // token shift of the first row: the carried state for t == 0, else the previous row's LN, recomputed
    if (NMIX > 0) {
        if (t0 == 0) {
#pragma unroll
            for (int q = 0; q < kLnMaxChunks; q++) {
                const int c = threadIdx.x + q * TH;
                if (c < nchunk) {
                    const f16x8 pv = *reinterpret_cast<const f16x8 *>(prev_in + slot * C + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; e++) prev[q][e] = (float)pv[e];
                }
            }
        } else {
            const int64_t rp = ((int64_t)bb * T + t0 - 1) * C;
            ln_row<TH>(x + rp, delta ? delta + rp : nullptr, nullptr, ln_w, ln_b, C, eps, prev, red, dpart ? dpart + rp : nullptr, dsplits,
                   dsplit_stride, q_rx, q_mx, fz.in_S ? fz.in_S + ((int64_t)bb * T + t0 - 1) * fz.in_S_parts * 3 : nullptr,
                   fz.in_S_parts, qsum[1]);
        }
    }
    for (int r = 0; r < R; r++) {
        const int t = t0 + r;
        if (t >= T) break;
        const int row = bb * T + t;
        const int64_t ro = (int64_t)row * C;
        const float *q_row = fz.in_S ? fz.in_S + (int64_t)row * fz.in_S_parts * 3 : nullptr;
        f16x8 mvh[NMIX ? NMIX : 1];                   // (HOIST) the lerp coefficients of this lane's chunk, requested with the row
        if (HOIST && NMIX > 0 && (int)threadIdx.x < nchunk) {
#pragma unroll
            for (int m = 0; m < NMIX; m++) mvh[m] = *reinterpret_cast<const f16x8 *>(mix + (int64_t)m * C + threadIdx.x * 8);
        }
        ln_row<TH>(x + ro, delta ? delta + ro : nullptr, x_out ? x_out + ro : nullptr, ln_w, ln_b, C, eps, cur, red,
               dpart ? dpart + ro : nullptr, dsplits, dsplit_stride, q_rx, q_mx, q_row, fz.in_S_parts, qsum[0],
               stamps ? stamps + (int64_t)blockIdx.x * 8 : nullptr);
        stamp(4);                                      // normalised (the LN weights have arrived)
        if (NMIX == 0) {
#pragma unroll
            for (int q = 0; q < kLnMaxChunks; q++) {
                const int c = threadIdx.x + q * TH;
                if (c < nchunk) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = h(cur[q][e]);
                    *reinterpret_cast<f16x8 *>(out + ro + c * 8) = o;
                }
            }
            continue;
        }
        constexpr int NQ = NMIX >= 3 ? 3 : (NMIX == 1 ? 1 : 0);     // mix planes that can carry an mm8 prologue (r, k, v of six; the one of one)
        const int n_q = p_xs ? (fz.out_planes > 0 ? (fz.out_planes < NQ ? fz.out_planes : NQ) : (NQ ? 1 : 0)) : 0;
        float ps[NQ ? NQ : 1][3];
#pragma unroll
        for (int m = 0; m < (NQ ? NQ : 1); m++) ps[m][0] = ps[m][1] = ps[m][2] = 0.f;
#pragma unroll
        for (int q = 0; q < kLnMaxChunks; q++) {
            const int c = threadIdx.x + q * TH;
            if (c < nchunk) {
                float dx[8];
                f16x8 cv;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    dx[e] = (float)h(prev[q][e] - cur[q][e]);
                    cv[e] = h(cur[q][e]);
                }
                if (t == T - 1) *reinterpret_cast<f16x8 *>(prev_out + slot * C + c * 8) = cv;
#pragma unroll
                for (int m = 0; m < NMIX; m++) {
                    const f16x8 mv = HOIST ? mvh[m] : *reinterpret_cast<const f16x8 *>(mix + (int64_t)m * C + c * 8);
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = h(cur[q][e] + (float)h(dx[e] * (float)mv[e]));
                    *reinterpret_cast<f16x8 *>(out + (int64_t)m * out_stride + ro + c * 8) = o;
                    if (NQ > 0 && m < NQ && m < n_q) {  // mm8 activation prologue of the GEMM that consumes plane m (mm8_prep_kernel's arithmetic)
                        const f16x8 rv = *reinterpret_cast<const f16x8 *>(p_ry + (int64_t)m * C + c * 8);
                        const f16x8 yv = *reinterpret_cast<const f16x8 *>(p_my + (int64_t)m * C + c * 8);
                        f16x8 xs;
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            xs[e] = (f16)((float)o[e] * (float)rv[e]);
                            ps[m < NQ ? m : 0][0] += (float)xs[e];
                            ps[m < NQ ? m : 0][1] += (float)o[e] * (float)yv[e];
                            ps[m < NQ ? m : 0][2] += (float)o[e];
                        }
                        *reinterpret_cast<f16x8 *>(p_xs + (int64_t)m * out_stride + ro + c * 8) = xs;
                    }
                }
            }
        }
        if (NQ > 0) {
            const int64_t rows_total = out_stride / C;                 // S of plane m: out_S[m][rows][3]
#pragma unroll
            for (int m = 0; m < NQ; m++) {
                if (m < n_q) {                                         // (uniform over the workgroup)
                    block_sum3<TH>(ps[m][0], ps[m][1], ps[m][2], red);
                    if (threadIdx.x == 0) {
                        float *dst = fz.out_S + ((int64_t)m * rows_total + row) * 3;
                        dst[0] = ps[m][0], dst[1] = ps[m][1], dst[2] = ps[m][2];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kLnMaxChunks; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) prev[q][e] = cur[q][e];
        stamp(5);                                      // lerps formed, stores issued
    }
    if (stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[(int64_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();      // lane 0's stores acknowledged
    }
}

// One lane per 8 channels; 8 lanes per head.
__global__ __launch_bounds__(256) void tmix_mid_kernel(
    const int64_t nchunks, const int C, f16 *__restrict__ k, f16 *__restrict__ v, const f16 *__restrict__ a_pre,
    const f16 *__restrict__ vg_pre, const f16 *__restrict__ v_first, const f16 *__restrict__ k_k,
    const f16 *__restrict__ k_a, f16 *__restrict__ neg_kk, f16 *__restrict__ kka) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = g < nchunks;
    const int64_t off = (live ? g : 0) * 8;
    const int ch = (int)(off % C);
    const f16x8 kv = *reinterpret_cast<const f16x8 *>(k + off);
    const f16x8 ap = *reinterpret_cast<const f16x8 *>(a_pre + off);
    const f16x8 kkv = *reinterpret_cast<const f16x8 *>(k_k + ch);
    const f16x8 kav = *reinterpret_cast<const f16x8 *>(k_a + ch);
    float kk_in[8], ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        kk_in[e] = (float)h((float)kv[e] * (float)kkv[e]);
        ss += kk_in[e] * kk_in[e];
    }
    ss = head_sum(ss);
    float nrm = (float)h(sqrtf(ss));           // torch.norm -> binary16, clamp_min(1e-12 -> 0 in binary16)
    nrm = nrm > 0.f ? nrm : 0.f;
    f16x8 ko, nk, ka;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float a = (float)h(sigmoid_f((float)ap[e]));
        const float kk = (float)h(kk_in[e] / nrm);
        const float t1 = (float)h(a - 1.0f);
        const float t2 = (float)h(t1 * (float)kav[e]);
        const float t3 = (float)h(1.0f + t2);
        ko[e] = h((float)kv[e] * t3);
        nk[e] = h(-kk);
        ka[e] = h(kk * a);
    }
    if (live) {
        *reinterpret_cast<f16x8 *>(k + off) = ko;
        *reinterpret_cast<f16x8 *>(neg_kk + off) = nk;
        *reinterpret_cast<f16x8 *>(kka + off) = ka;
        if (v_first) {
            const f16x8 vv = *reinterpret_cast<const f16x8 *>(v + off);
            const f16x8 vf = *reinterpret_cast<const f16x8 *>(v_first + off);
            const f16x8 gp = *reinterpret_cast<const f16x8 *>(vg_pre + off);
            f16x8 vo;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float gate = (float)h(sigmoid_f((float)gp[e]));
                const float d = (float)h((float)vf[e] - (float)vv[e]);
                vo[e] = h((float)vv[e] + (float)h(d * gate));
            }
            *reinterpret_cast<f16x8 *>(v + off) = vo;
        }
    }
}

__global__ __launch_bounds__(256) void tmix_post_kernel(
    const int64_t nchunks, const int C, const f16 *__restrict__ y, const f16 *__restrict__ r, const f16 *__restrict__ k,
    const f16 *__restrict__ v, const f16 *__restrict__ g, const f16 *__restrict__ r_k, const f16 *__restrict__ lnx_w,
    const f16 *__restrict__ lnx_b, const float eps, f16 *__restrict__ out) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = gi < nchunks;
    const int64_t off = (live ? gi : 0) * 8;
    const int ch = (int)(off % C);
    const f16x8 yv = *reinterpret_cast<const f16x8 *>(y + off);
    const f16x8 rv = *reinterpret_cast<const f16x8 *>(r + off);
    const f16x8 kv = *reinterpret_cast<const f16x8 *>(k + off);
    const f16x8 rkv = *reinterpret_cast<const f16x8 *>(r_k + ch);
    float s = 0.f, bsum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        s += (float)yv[e];
        bsum += (float)h((float)h((float)rv[e] * (float)kv[e]) * (float)rkv[e]);
    }
    const float mean = head_sum(s) * (1.0f / 64.0f);
    const float bonus = (float)h(head_sum(bsum));
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float d = (float)yv[e] - mean;
        s2 += d * d;
    }
    const float rstd = 1.0f / sqrtf(head_sum(s2) * (1.0f / 64.0f) + eps);
    if (live) {
        const f16x8 vv = *reinterpret_cast<const f16x8 *>(v + off);
        const f16x8 gv = *reinterpret_cast<const f16x8 *>(g + off);
        const f16x8 wv = *reinterpret_cast<const f16x8 *>(lnx_w + ch);
        const f16x8 bv = *reinterpret_cast<const f16x8 *>(lnx_b + ch);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float gn = (float)h(((float)yv[e] - mean) * rstd * (float)wv[e] + (float)bv[e]);
            const float t = (float)h(gn + (float)h(bonus * (float)vv[e]));
            o[e] = h(t * (float)gv[e]);
        }
        *reinterpret_cast<f16x8 *>(out + off) = o;
    }
}

__global__ __launch_bounds__(256) void relu_sq_kernel(const int64_t nchunks, f16 *__restrict__ x) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= nchunks) return;
    f16x8 v = *reinterpret_cast<const f16x8 *>(x + gi * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float t = (float)v[e] > 0.f ? (float)v[e] : 0.f;
        v[e] = h(t * t);
    }
    *reinterpret_cast<f16x8 *>(x + gi * 8) = v;
}

// LoRA hidden activations, planes [v, w, a, g] of D columns each: w -> tanh, g -> sigmoid
// (rwkv7.py:626 torch.tanh(xw@w1), :630 torch.sigmoid(xg@g1)); a and v pass through (:629, :637).
__global__ __launch_bounds__(256) void lora_act_kernel(const int64_t plane_chunks, const int first_plane,
                                                       f16 *__restrict__ hbuf, const int nplanes) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int plane = blockIdx.y + first_plane;          // 0=v 1=w 2=a 3=g
    if (gi >= plane_chunks || (plane != 1 && plane != 3)) return;
    f16 *p = hbuf + ((int64_t)blockIdx.y * plane_chunks + gi) * 8;
    f16x8 v = *reinterpret_cast<const f16x8 *>(p);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = plane == 1 ? h(tanhf((float)v[e])) : h(sigmoid_f((float)v[e]));
    *reinterpret_cast<f16x8 *>(p) = v;
}

inline bool mis16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

// Diagnostic (tools/ln_timeline.py): while set, every add_ln_mix launch writes 8 100-MHz stamps per workgroup to buf (entry, row data
// arrived, mean, variance, normalised, stores issued, stores acknowledged); buf must hold 8 x the largest grid.  Process-wide.
static unsigned long long *g_ln_probe = nullptr;
extern "C" int rwkv7_ln_probe(void *buf) {
    g_ln_probe = static_cast<unsigned long long *>(buf);
    return 0;
}

extern "C" int rwkv7_add_ln_mix(int B, int T, int C, int n_mix, const void *x, const void *delta, void *x_out,
                                const void *ln_w, const void *ln_b, float eps, const void *prev_in, void *prev_out,
                                const void *mix, void *out, int64_t out_stride, const int32_t *slot_idx,
                                const float *delta_partials, int delta_splits, void *stream) {
    return rwkv7_add_ln_mix_mm8(B, T, C, n_mix, x, delta, x_out, ln_w, ln_b, eps, prev_in, prev_out, mix, out, out_stride, slot_idx,
                                delta_partials, delta_splits, nullptr, stream);
}

extern "C" int rwkv7_add_ln_mix_mm8(int B, int T, int C, int n_mix, const void *x, const void *delta, void *x_out,
                                    const void *ln_w, const void *ln_b, float eps, const void *prev_in, void *prev_out,
                                    const void *mix, void *out, int64_t out_stride, const int32_t *slot_idx,
                                    const float *delta_partials, int delta_splits, const chirrup_mm8_fuse *fuse, void *stream) {
    chirrup_mm8_fuse q{};
    if (fuse) q = *fuse;
    if (q.in_S && (!delta_partials || !q.in_rx || !q.in_mx)) return CHIRRUP_E_NULL;
    if (q.in_S && q.in_S_parts <= 0) q.in_S_parts = 1;
    if (q.out_xs && (n_mix == 0 || !q.out_ry || !q.out_my || !q.out_S)) return q.out_xs && n_mix == 0 ? CHIRRUP_E_UNSUPPORTED : CHIRRUP_E_NULL;
    if (q.out_xs && (q.out_planes < 0 || q.out_planes > (n_mix == 1 ? 1 : 3))) return CHIRRUP_E_SHAPE;
    if (mis16(q.in_rx) || mis16(q.in_mx) || mis16(q.out_ry) || mis16(q.out_my) || mis16(q.out_xs)) return CHIRRUP_E_ALIGN;
    if (B <= 0 || T <= 0 || C <= 0 || (C & 63) || C > kLnMaxThreads * kLnMaxChunks * 8) return CHIRRUP_E_SHAPE;
    if (!(n_mix == 0 || n_mix == 1 || n_mix == 6)) return CHIRRUP_E_UNSUPPORTED;
    if (!x || !ln_w || !ln_b || !out) return CHIRRUP_E_NULL;
    if (n_mix > 0 && (!prev_in || !prev_out || !mix)) return CHIRRUP_E_NULL;
    if (n_mix > 0 && T > 1 && prev_in == prev_out) return CHIRRUP_E_UNSUPPORTED;  // rows race on the carry
    // T > 1: the workgroup of row t recomputes LN(x[t-1] + delta[t-1]) for the token shift while the workgroup of row
    // t-1 stores x_out[t-1]; in place that is a cross-workgroup race (delta added twice when the store lands first)
    if (n_mix > 0 && T > 1 && x_out == x && (delta || delta_partials)) return CHIRRUP_E_UNSUPPORTED;
    if (delta_partials && (delta || delta_splits <= 0)) return CHIRRUP_E_UNSUPPORTED;
    if (mis16(delta_partials)) return CHIRRUP_E_ALIGN;
    if (mis16(x) || mis16(delta) || mis16(x_out) || mis16(ln_w) || mis16(ln_b) || mis16(prev_in) || mis16(prev_out) ||
        mis16(mix) || mis16(out) || (out_stride & 7))
        return CHIRRUP_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // rows per workgroup (see the kernel): 1 for short chunks; else enough that the grid is ONE round of the chip's
    // 512 resident 1024-lane workgroups (a second, mostly empty round costs as much as the first), at most 8
    int R = 1;
    if (T >= 8) {
        R = (int)(((int64_t)B * T + 511) / 512);
        R = R < 2 ? 2 : (R > 8 ? 8 : R);
        if (R > T) R = T;
    }
    static const int forced_th = [] { const char *e = getenv("CHIRRUP_LN_THREADS"); return e ? atoi(e) : 0; }();     // (A/B only: 1024 = round 2's kernel)
    // one 8-channel chunk per lane.  A/B on one box (gpurun_out/r3_ln_ab.log): 256 lanes with the hoisted loads win 0.5 % of the step at
    // C = 2048 (1.5B bsz 32: 1.806 -> 1.797 ms); 512 lanes at C = 4096 LOSE 0.5-1 % against 1024 lanes of which half idle (7.2B bsz 200
    // 6.75 -> 6.79, bsz 32 3.91 -> 3.93; 13.3B bsz 64 7.37 -> 7.42) -- wider rows keep the 1024-lane form
    const int TH = forced_th == 1024 ? 1024 : (forced_th == 512 && C <= 4096 ? 512 : (C <= 2048 ? 256 : 1024));
    const dim3 grid((unsigned)(B * ((T + R - 1) / R))), block(TH);
#define ARGS T, C, (const f16 *)x, (const f16 *)delta, (f16 *)x_out, (const f16 *)ln_w, (const f16 *)ln_b, eps, \
             (const f16 *)prev_in, (f16 *)prev_out, (const f16 *)mix, (f16 *)out, out_stride, slot_idx, delta_partials, \
             delta_splits, (int64_t)B * T * C, q, R, g_ln_probe
#define LN_GO(NM)                                                                                  \
    do {                                                                                           \
        if (g_ln_probe && TH == 1024) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 1024, true>), grid, block, 0, st, ARGS);  \
        else if (g_ln_probe && TH == 256) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 256, true>), grid, block, 0, st, ARGS);  \
        else if (!fuse && TH == 256) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 256, false, false>), grid, block, 0, st, ARGS);  \
        else if (!fuse && TH == 1024) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 1024, false, false>), grid, block, 0, st, ARGS);  \
        else if (!fuse && TH == 512) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 512, false, false>), grid, block, 0, st, ARGS);  \
        else if (TH == 256) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 256>), grid, block, 0, st, ARGS);  \
        else if (TH == 512) hipLaunchKernelGGL((add_ln_mix_kernel<NM, 512>), grid, block, 0, st, ARGS); \
        else hipLaunchKernelGGL((add_ln_mix_kernel<NM, 1024>), grid, block, 0, st, ARGS);           \
    } while (0)
    if (n_mix == 0) LN_GO(0);
    else if (n_mix == 1) LN_GO(1);
    else LN_GO(6);
#undef LN_GO
#undef ARGS
    return (int)hipGetLastError();
}

extern "C" int rwkv7_tmix_mid(int64_t rows, int C, void *k, void *v, const void *a_pre, const void *vg_pre,
                              const void *v_first, const void *k_k, const void *k_a, void *neg_kk, void *kka,
                              void *stream) {
    if (rows <= 0 || C <= 0 || (C & 63)) return CHIRRUP_E_SHAPE;
    if (!k || !v || !a_pre || !k_k || !k_a || !neg_kk || !kka) return CHIRRUP_E_NULL;
    if ((v_first == nullptr) != (vg_pre == nullptr)) return CHIRRUP_E_NULL;
    if (mis16(k) || mis16(v) || mis16(a_pre) || mis16(vg_pre) || mis16(v_first) || mis16(k_k) || mis16(k_a) ||
        mis16(neg_kk) || mis16(kka))
        return CHIRRUP_E_ALIGN;
    const int64_t nchunks = rows * (C / 8);
    hipLaunchKernelGGL(tmix_mid_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), nchunks, C, (f16 *)k, (f16 *)v, (const f16 *)a_pre,
                       (const f16 *)vg_pre, (const f16 *)v_first, (const f16 *)k_k, (const f16 *)k_a, (f16 *)neg_kk,
                       (f16 *)kka);
    return (int)hipGetLastError();
}

extern "C" int rwkv7_tmix_post(int64_t rows, int C, const void *y, const void *r, const void *k, const void *v,
                               const void *g, const void *r_k, const void *lnx_w, const void *lnx_b, float eps,
                               void *out, void *stream) {
    if (rows <= 0 || C <= 0 || (C & 63)) return CHIRRUP_E_SHAPE;
    if (!y || !r || !k || !v || !g || !r_k || !lnx_w || !lnx_b || !out) return CHIRRUP_E_NULL;
    if (mis16(y) || mis16(r) || mis16(k) || mis16(v) || mis16(g) || mis16(r_k) || mis16(lnx_w) || mis16(lnx_b) || mis16(out))
        return CHIRRUP_E_ALIGN;
    const int64_t nchunks = rows * (C / 8);
    hipLaunchKernelGGL(tmix_post_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), nchunks, C, (const f16 *)y, (const f16 *)r, (const f16 *)k,
                       (const f16 *)v, (const f16 *)g, (const f16 *)r_k, (const f16 *)lnx_w, (const f16 *)lnx_b, eps,
                       (f16 *)out);
    return (int)hipGetLastError();
}

extern "C" int rwkv7_relu_sq(int64_t n, void *x, void *stream) {
    if (n <= 0 || (n & 7)) return CHIRRUP_E_SHAPE;
    if (!x) return CHIRRUP_E_NULL;
    if (mis16(x)) return CHIRRUP_E_ALIGN;
    const int64_t nchunks = n / 8;
    hipLaunchKernelGGL(relu_sq_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), nchunks, (f16 *)x);
    return (int)hipGetLastError();
}

extern "C" int rwkv7_lora_act(int nplanes, int first_plane, int64_t plane_elems, void *hbuf, void *stream) {
    if (nplanes <= 0 || first_plane < 0 || first_plane + nplanes > 4 || plane_elems <= 0 || (plane_elems & 7))
        return CHIRRUP_E_SHAPE;
    if (!hbuf) return CHIRRUP_E_NULL;
    if (mis16(hbuf)) return CHIRRUP_E_ALIGN;
    const int64_t chunks = plane_elems / 8;
    hipLaunchKernelGGL(lora_act_kernel, dim3((unsigned)((chunks + 255) / 256), (unsigned)nplanes), dim3(256), 0,
                       static_cast<hipStream_t>(stream), chunks, first_plane, (f16 *)hbuf, nplanes);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The two ends of a decode step that were torch kernels (rwkv7.py:503-517, :561-563; chirrup/worker.py feeding the sampled id
// back): as `tok < 0`, an index into the fed-back ids, a dtype copy, `where`, the embedding gather and the zeroing of the
// launch-sync words they were SIX launches of ~4.8 us in front of every step, and `state[2] += T` over a slot list three behind it.
namespace {
// x[row] = emb[token]; token = tokens[row], or feedback[slot of the row's sequence] when tokens[row] < 0 (the id the sampler left
// there one step ago -- never seen by the host).  One workgroup per row, 16 B per lane per trip.  Workgroup 0 also zeroes
// `zero_words` (the launch-sync words of this stream: tile counters and time-mix hand-off words).
__global__ __launch_bounds__(256) void embed_rows_kernel(const int T, const int C, const int V, const f16 *__restrict__ emb,
                                                         const int64_t *__restrict__ tokens, const int32_t *__restrict__ slot_idx,
                                                         const int32_t *__restrict__ feedback, f16 *__restrict__ x,
                                                         int32_t *__restrict__ zero_words, const int n_zero,
                                                         const int32_t *__restrict__ elapsed_pool, int32_t *__restrict__ elapsed_rows) {
    const int row = blockIdx.x;
    if (row == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0;
    const int b = row / T;
    const int slot = slot_idx ? slot_idx[b] : b;
    if (elapsed_rows && row == b * T && threadIdx.x == 0) elapsed_rows[b] = elapsed_pool[slot];      // the slot table's counters, by row
    int64_t tok = tokens[row];
    if (tok < 0 && feedback) tok = feedback[slot];
    const bool ok = tok >= 0 && tok < V;               // (an id outside the table: a zero row instead of a wild read)
    const f16x8 *src = reinterpret_cast<const f16x8 *>(emb + (ok ? tok : 0) * (int64_t)C);
    f16x8 *dst = reinterpret_cast<f16x8 *>(x + (int64_t)row * C);
    for (int c = threadIdx.x; c < (C >> 3); c += 256) dst[c] = ok ? src[c] : f16x8{};
}

__global__ __launch_bounds__(256) void advance_elapsed_kernel(const int B, const int T, const int32_t *__restrict__ slot_idx,
                                                              int32_t *__restrict__ elapsed) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) elapsed[slot_idx ? slot_idx[b] : b] += T;
}
}  // namespace

extern "C" int rwkv7_embed_rows(int B, int T, int C, int V, const void *emb, const int64_t *tokens, const int32_t *slot_idx,
                                const int32_t *feedback, void *x, int32_t *zero_words, int n_zero, const int32_t *elapsed_pool,
                                int32_t *elapsed_rows, void *stream) {
    if (B <= 0 || T <= 0 || C <= 0 || (C & 7) || V <= 0 || n_zero < 0) return CHIRRUP_E_SHAPE;
    if (!emb || !tokens || !x || (n_zero && !zero_words) || (elapsed_rows && !elapsed_pool)) return CHIRRUP_E_NULL;
    if (mis16(emb) || mis16(x)) return CHIRRUP_E_ALIGN;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)(B * T)), dim3(256), 0, static_cast<hipStream_t>(stream), T, C, V,
                       static_cast<const f16 *>(emb), tokens, slot_idx, feedback, static_cast<f16 *>(x), zero_words, n_zero, elapsed_pool,
                       elapsed_rows);
    return (int)hipGetLastError();
}

extern "C" int rwkv7_advance_elapsed(int B, int T, const int32_t *slot_idx, int32_t *elapsed, void *stream) {
    if (B <= 0 || T <= 0) return CHIRRUP_E_SHAPE;
    if (!elapsed) return CHIRRUP_E_NULL;
    hipLaunchKernelGGL(advance_elapsed_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), B, T,
                       slot_idx, elapsed);
    return (int)hipGetLastError();
}

// dst[slot][:] = src[slot][:] for the slots of a batch (binary16 rows of C channels): the token-shift carry of a chunk of T > 1
// tokens goes to a side table first (rows of one launch race on the carry, see rwkv7_add_ln_mix) and is committed per slot
// afterwards -- as torch ops an index_select and an index_copy_ per commit, two commits per layer.
namespace {
__global__ __launch_bounds__(256) void copy_slot_rows_kernel(const int C, const int32_t *__restrict__ slot_idx, const f16 *__restrict__ src,
                                                             f16 *__restrict__ dst) {
    const int64_t off = (int64_t)slot_idx[blockIdx.x] * C;
    const f16x8 *s = reinterpret_cast<const f16x8 *>(src + off);
    f16x8 *d = reinterpret_cast<f16x8 *>(dst + off);
    for (int c = threadIdx.x; c < (C >> 3); c += 256) d[c] = s[c];
}
}  // namespace

extern "C" int rwkv7_copy_slot_rows(int B, int C, const int32_t *slot_idx, const void *src, void *dst, void *stream) {
    if (B <= 0 || C <= 0 || (C & 7)) return CHIRRUP_E_SHAPE;
    if (!slot_idx || !src || !dst) return CHIRRUP_E_NULL;
    if (mis16(src) || mis16(dst)) return CHIRRUP_E_ALIGN;
    hipLaunchKernelGGL(copy_slot_rows_kernel, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), C, slot_idx,
                       static_cast<const f16 *>(src), static_cast<f16 *>(dst));
    return (int)hipGetLastError();
}
