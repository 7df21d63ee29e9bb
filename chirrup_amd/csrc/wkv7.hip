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

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__global__ __launch_bounds__(64, WKV7_MIN_WAVES) void wkv7_seq_kernel(
    const int T, const int C, const int H, f16 *__restrict__ state, const int64_t slot_stride,
    const int32_t *__restrict__ slot_idx, const f16 *__restrict__ r_, const f16 *__restrict__ w_,
    const f16 *__restrict__ k_, const f16 *__restrict__ v_, const f16 *__restrict__ a_,
    const f16 *__restrict__ b_, f16 *__restrict__ y_, const int32_t *__restrict__ elapsed_t) {
    // [0, 8192): swizzled state image; then r, w~, k, a, b strips of 128 B each.
    __shared__ __attribute__((aligned(16))) unsigned char smem[kStateBytes + 5 * kVecBytes];

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
    f16 rj = r_[o], wj = w_[o], kj = k_[o], aj = a_[o], bj = b_[o], vi = v_[o];
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

    for (int t = 0; t < T; t++) {
        // dither: int32 wrap-around multiply, int -> float, exact scale (.cu:23, :59)
        const float dither = kTwoToNeg41 * (float)(int32_t)(kRo1 * (uint32_t)(et + t));
        vec[0 * 64 + lane] = rj;
        vec[1 * 64 + lane] = decay_term(wj, dither);
        vec[2 * 64 + lane] = kj;
        vec[3 * 64 + lane] = aj;
        vec[4 * 64 + lane] = bj;
        const f16 vv = vi;
        const int64_t o_cur = o;
        if (t + 1 < T) {  // prefetch the next timestep's vectors under this step's arithmetic
            o += C;
            rj = r_[o]; wj = w_[o]; kj = k_[o]; aj = a_[o]; bj = b_[o]; vi = v_[o];
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
        y_[o_cur] = y2.x + y2.y;
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
    hipLaunchKernelGGL(wkv7_seq_kernel, dim3((unsigned)(B * H)), dim3(64), 0, static_cast<hipStream_t>(stream), T, C, H,
                       static_cast<f16 *>(state), slot_stride, slot_idx, static_cast<const f16 *>(r),
                       static_cast<const f16 *>(w), static_cast<const f16 *>(k), static_cast<const f16 *>(v),
                       static_cast<const f16 *>(a), static_cast<const f16 *>(b), static_cast<f16 *>(y), elapsed_t);
    return (int)hipGetLastError();
}

#ifndef WKV7_VARIANT_BUILD
extern "C" int wkv7_fwd_one(int B, int C, int H, void *state, const void *r, const void *w, const void *k,
                            const void *v, const void *a, const void *b, void *y, const int32_t *elapsed_t,
                            const int32_t *slot_idx, int64_t slot_stride, void *stream) {
    return wkv7_fwd_seq(B, 1, C, H, state, r, w, k, v, a, b, y, elapsed_t, slot_idx, slot_stride, stream);
}
#endif
