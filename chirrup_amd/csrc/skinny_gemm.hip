// Skinny-M MFMA GEMMs for the decode step:  Y[M][N] = X[M][K] . W[N][K]^T   (M <= 256 token rows)
// with binary16 or uint8 (mm8) weights, binary32 accumulation.  One RWKV-7 layer at 1..256 token rows runs five launches of
// the ring kernel below: R/K/V + the four LoRA down-projections (grouped), the four LoRA up-projections (batched, per-
// problem reduction length), att.output, ffn.key, ffn.value (DESIGN.md sections 4-5 have the measurements).
//
// Why hand-written: at M = 200 the library GEMMs (hipBLASLt through torch) stream weights at 1.4-2.8 TB/s and every
// projection is its own launch (or needs a side stream, whose graph edges cost ~19 us per layer).  Here the batch is
// small enough that ONE workgroup holds all M rows of x for its K-block in LDS, and independent problems share a launch.
//
// skinny_gemm_ring_kernel (what ships, variant 3 of skinny_gemm_select):
//   * workgroup = 4 compute waves + 4 loader waves.  A 64-wide K-block of BOTH operands goes global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4) into a 3-slot ring: x image MT*32 rows x 128 B, W image 128 rows x 128 B (u8: 64 B),
//     both XOR-swizzled on the SOURCE address so that the ds_read_b128 of the MFMA fragments are conflict-free.  No
//     load of the main loop has a register destination, so one hand-placed `s_waitcnt vmcnt(n)` + raw `s_barrier`
//     per K-block keeps two whole stages in flight.  The loader waves exist because a wave is blocked while its
//     LDS-DMA instructions issue; the compute waves spend that time in MFMAs.
//   * compute wave w owns 32 rows of W: v_mfma_f32_32x32x16_f16 with A = W tile (32 n x 16 k), B = x^T (16 k x 32 m);
//     one W fragment feeds MT = ceil(M/32) MFMAs (u8 -> f16 by v_perm in registers).  __builtin_amdgcn_sched_barrier
//     pins "read the next MT x fragments, then MT MFMAs": left alone the scheduler sinks every ds_read to its MFMA.
//   * split-K over blockIdx.y with binary32 partials; the partials are reduced by skinny_reduce_kernel (bias, relu^2,
//     tanh / sigmoid of the LoRA planes, or the mm8 rank-1 corrections of scripts/test_mm8/benchmark.py:167-179) or by
//     the NEXT layer-norm kernel (rwkv7_add_ln_mix, delta_partials).
//   * batched (gridDim.z problems at uniform strides, optional per-problem K) and grouped launches (per-problem
//     operands, N and output stride; blockIdx.x runs over an exact tile list -- an empty workgroup would still have to
//     be given its 132 KiB of LDS before it could leave).
//   * workgroup -> tile order is XCD-aware (tile_of_block).
// skinny_gemm_kernel is the first, register-staged design (variant 0), kept for A/B: W streamed HBM -> VGPRs, x through a
// double-buffered LDS image.  Variants 1 and 2 are intermediate forms of the ring kernel (every wave loads; x / W loader
// roles).  K is consumed in a permuted order inside each 64-block (lane half h takes k = 32h + 8s + j at MFMA step s)
// -- the same permutation on both operands, so the dot products are unchanged.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int kKB = 64;          // K-block
constexpr int kThreads = 256;    // 4 waves
constexpr int kBN = 128;         // output columns per workgroup

enum { EPI_F16 = 0, EPI_PARTIAL = 1 };

// two uint8 -> two binary16, exact: bytes b0,b1 -> halves 0x6400|b = 1024 + b, minus 1024
__device__ __forceinline__ f16x2 cvt_u8x2(uint32_t packed, int sel_lo) {
    // v_perm_b32: selector bytes pick from {src0 (hi dword), src1 (lo dword)}; 0x64 bytes come from a constant
    const uint32_t magic = 0x64646464u;
    uint32_t r;
    if (sel_lo)  r = __builtin_amdgcn_perm(magic, packed, 0x04010400u);   // [b1,0x64 | b0,0x64] -> halves (b0),(b1)
    else         r = __builtin_amdgcn_perm(magic, packed, 0x04030402u);   // bytes 2,3
    f16x2 v = __builtin_bit_cast(f16x2, r);
    const f16x2 off = {(f16)1024.f, (f16)1024.f};
    return v - off;
}

// Raw weight bytes of one K-block for one lane (k = k0 + 32h .. +32): 64 B (f16) or 32 B (u8).
template <bool W8>
struct WRaw {
    u32x4 q[W8 ? 2 : 4];
};

// Workgroup -> (N-group, K-slice).  Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its
// own L2: give XCD j a contiguous run of the K-slice-major tile order, so the x K-slice a workgroup re-reads is shared
// by its L2 neighbours (ffn.value at bsz 200: x is 6.5 MB, a K-slice 0.8 MB; the L2 is 4 MB).
__device__ __forceinline__ void tile_of_block(int &ngroup, int &kslice, int &batch) {
    const int G = gridDim.x, GS = G * gridDim.y, total = GS * gridDim.z;
    const int L = blockIdx.x + G * blockIdx.y + GS * blockIdx.z;
    int v = (total & 7) ? L : (L & 7) * (total >> 3) + (L >> 3);
    batch = v / GS;
    v -= batch * GS;
    kslice = v / G;
    ngroup = v - kslice * G;
}

// element strides between the problems of a batched launch (gridDim.z problems; 0s for a single GEMM)
struct BatchStrides {
    int64_t x, w, y, bias;
    int k[8];          // per-problem reduction length (<= K, multiple of 64) or 0 = K: zero-padded tails are not streamed
    int tiled;         // W of every problem is in the tile-image layout (skinny_tile_weight)
    int relu_sq;       // EPI_F16 only: y = relu(binary16(x.w + bias))^2 in the epilogue (unsplit launches need no reduce)
};

// Problems that share only M, K, ldx, ldw and the split count (a "grouped" launch: R/K/V and the four LoRA
// down-projections of one RWKV-7 layer are seven such problems): everything else per problem.  used = 0: not grouped.
struct GroupTable {
    const f16 *X[8];
    const void *W[8];
    f16 *Y[8];
    const f16 *bias[8];
    float *part[8];
    int N[8], ldy[8], act[8];      // act: 0 none, 1 relu^2, 2 tanh, 3 sigmoid (reduce kernel)
    int tiled[8];                  // W of problem p is in the tile-image layout
    int first[9];                  // N-groups before problem p (gridDim.x = first[used]): no workgroup without a tile --
    int used;                      // an empty one would still wait for its 132 KiB of LDS before it could leave
};

template <int MT, bool W8, int EPI>
__global__ __launch_bounds__(kThreads) void skinny_gemm_kernel(
    const int M, const int N, const int K, const int k_slice, const f16 *__restrict__ X, const int ldx,
    const void *__restrict__ Wv, const int64_t ldw, f16 *__restrict__ Y, const int ldy,
    const f16 *__restrict__ bias, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x MT x 4 KiB
    constexpr int kTileBytes = MT * 32 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int ngroup, kslice, batch_unused;
    tile_of_block(ngroup, kslice, batch_unused);
    const int n0 = (ngroup * 4 + wave) * 32;
    const bool wave_live = n0 < N;
    const int k_begin = kslice * k_slice;
    const int k_end = (k_begin + k_slice) < K ? (k_begin + k_slice) : K;
    const int nkb = (k_end - k_begin) / kKB;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mt][i] = 0.f;

    // x K-block: MT 16-B chunks per lane.  Chunk g = i*256 + tid lands at LDS byte g*16 (linear,
    // conflict-free ds_write_b128); WHICH chunk of x that is is permuted on the source side so that
    // chunk c of row m sits at m*128 + ((c ^ ((m>>1)&7))<<4) for the B-fragment reads.
    const f16 *xsrc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const int g = i * kThreads + tid;
        int m = g >> 3;
        const int lc = (g & 7) ^ ((m >> 1) & 7);
        m = m < M ? m : M - 1;
        xsrc[i] = X + (int64_t)m * ldx + k_begin + lc * 8;
    }
    auto load_x = [&](int kb, u32x4 (&xr)[MT]) {
#pragma unroll
        for (int i = 0; i < MT; i++) xr[i] = *reinterpret_cast<const u32x4 *>(xsrc[i] + kb * kKB);
    };
    auto store_x = [&](int buf, const u32x4 (&xr)[MT]) {
#pragma unroll
        for (int i = 0; i < MT; i++) *reinterpret_cast<u32x4 *>(smem + buf * kTileBytes + (i * kThreads + tid) * 16) = xr[i];
    };
    const int n_row = (n0 + r) < N ? (n0 + r) : (N - 1);
    const unsigned char *wrow = static_cast<const unsigned char *>(Wv) + ((int64_t)n_row * ldw + k_begin + 32 * h) * (W8 ? 1 : 2);
    auto load_w = [&](int kb, WRaw<W8> &w) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(wrow + (int64_t)kb * kKB * (W8 ? 1 : 2));
#pragma unroll
        for (int i = 0; i < (W8 ? 2 : 4); i++) w.q[i] = p[i];
    };

    auto compute = [&](int buf, const WRaw<W8> &w) {
        f16x8 wf[4];
        if constexpr (W8) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const uint32_t lo = w.q[s >> 1][2 * (s & 1)], hi = w.q[s >> 1][2 * (s & 1) + 1];
                const f16x2 a = cvt_u8x2(lo, 1), b = cvt_u8x2(lo, 0), c = cvt_u8x2(hi, 1), d = cvt_u8x2(hi, 0);
                wf[s] = (f16x8){a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) wf[s] = __builtin_bit_cast(f16x8, w.q[s]);
        }
        const unsigned char *xt = smem + buf * kTileBytes;
        auto bfrag = [&](int s, int mt) {
            const int m = mt * 32 + r;
            return *reinterpret_cast<const f16x8 *>(xt + m * 128 + (((4 * h + s) ^ ((m >> 1) & 7)) << 4));
        };
        // B fragments one MFMA group ahead of their use (two register sets of MT fragments)
        f16x8 b0[MT], b1[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) b0[mt] = bfrag(0, mt);
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) b1[mt] = bfrag(s + 1, mt);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], b0[mt], acc[mt], 0, 0, 0);
            if (s + 2 < 4) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) b0[mt] = bfrag(s + 2, mt);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s + 1], b1[mt], acc[mt], 0, 0, 0);
        }
    };

    // Everything is visible to hipcc (no LDS-DMA, no asm): its counted vmcnt waits keep the W loads
    // of the next two K-blocks and the x loads of the next one in flight across __syncthreads()
    // (a plain s_barrier when no LDS-DMA is pending).
    WRaw<W8> w0, w1, w2;
    u32x4 xr[MT];
    if (nkb > 0) {
        load_x(0, xr);
        load_w(0, w0);
        if (nkb > 1) load_w(1, w1);
        store_x(0, xr);
    }
    __syncthreads();
#define SKINNY_STEP(KB, WCUR, WNEXT2)                                   \
    if ((KB) < nkb) {                                                   \
        const bool more = (KB) + 1 < nkb;                               \
        if (more) load_x((KB) + 1, xr);                                 \
        if ((KB) + 2 < nkb) load_w((KB) + 2, WNEXT2);                   \
        compute((KB) & 1, WCUR);                                        \
        if (more) store_x(((KB) + 1) & 1, xr);                          \
        __syncthreads();                                                \
    }
    for (int kb = 0; kb < nkb; kb += 3) {
        SKINNY_STEP(kb, w0, w2)
        SKINNY_STEP(kb + 1, w1, w0)
        SKINNY_STEP(kb + 2, w2, w1)
    }
#undef SKINNY_STEP

    if (!wave_live) return;
    // acc[mt][i]: m = mt*32 + (lane & 31), n = n0 + 8*(i>>2) + 4*h + (i&3)
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int m = mt * 32 + r;
        if (m >= M) continue;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int n = n0 + 8 * g + 4 * h;
            if (n >= N) continue;
            if (EPI == EPI_F16) {
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = acc[mt][4 * g + e];
                    if (bias) v += (float)bias[n + e];
                    o[e] = (f16)v;
                }
                *reinterpret_cast<f16x4 *>(Y + (int64_t)m * ldy + n) = o;
            } else {
                const f32x4 o = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(part + ((int64_t)kslice * M + m) * N + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ring variant (binary16 weights): BOTH operands reach LDS by LDS-DMA into a 3-deep ring, so no load
// of the main loop has a register destination -- hipcc inserts no vmcnt wait of its own, and the
// hand-placed counted waits keep two whole K-blocks (2 x 44 KiB at M = 200) in flight per workgroup
// across raw barriers.  Same tiles, swizzle and MFMA schedule as above; W fragments are read from
// the ring like the x fragments.
// s_waitcnt vmcnt(ahead * PER): leave the `ahead` youngest stages (PER LDS-DMA instructions each) in flight
template <int PER>
__device__ __forceinline__ void wait_stages_ahead(const int ahead) {
#define WAIT_CASE(A)                                                                  \
    case A:                                                                           \
        if constexpr ((A) * PER <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * PER) : "memory"); \
        break;
    switch (ahead) {
        WAIT_CASE(7) WAIT_CASE(6) WAIT_CASE(5) WAIT_CASE(4) WAIT_CASE(3) WAIT_CASE(2) WAIT_CASE(1)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef WAIT_CASE
}

// Both operands through LDS-DMA rings; four waves compute (32 W rows x MT*32 x rows each).  Who issues the loads (RM):
//   0: every compute wave loads its share of x and W; one ring of XD (== WD) slots.
//   1: waves 0-1 load x (L2-resident, XD slots), waves 2-3 load W (HBM, WD slots): vmcnt counts per wave and retires
//      in order, so only a wave that issues nothing but W loads can keep WD-1 weight stages in flight without also
//      waiting for the x stage issued after them.
//   2: four extra loader waves (one per SIMD, 512-thread workgroup) issue everything: a wave is blocked while its
//      LDS-DMA instructions issue (~0.15 us per 16 KiB), which the compute waves then spend in MFMAs instead.
template <int MT, bool W8, int EPI, int XD, int WD, int RM>
__global__ __launch_bounds__(RM == 2 ? 512 : kThreads) void skinny_gemm_ring_kernel(
    const int M, const int N, const int K, const int k_slice, const f16 *__restrict__ X, const int ldx,
    const void *__restrict__ Wv, const int64_t ldw, f16 *__restrict__ Y, const int ldy,
    const f16 *__restrict__ bias, float *__restrict__ part, const BatchStrides bs, const GroupTable gt) {
    static_assert(RM == 1 || XD == WD, "one ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kXBytes = MT * 32 * 128;            // x K-block image
    constexpr int kWBytes = kBN * (W8 ? 64 : 128);    // W K-block image: 128 rows x 64 k (binary16 or uint8)
    constexpr int kLanes = RM == 1 ? 128 : 256;       // lanes that load one operand
    constexpr int kXLoads = kXBytes / 16 / kLanes;    // LDS-DMA instructions per loading lane per stage
    constexpr int kWLoads = kWBytes / 16 / kLanes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool computes = wave < 4;
    const bool loads_x = RM == 0 || (RM == 1 ? wave < 2 : wave >= 4), loads_w = RM == 0 || (RM == 1 ? wave >= 2 : wave >= 4);
    const int lt = RM == 1 ? (tid & 127) : (tid & 255), lw = RM == 1 ? (wave & 1) : (wave & 3);
    int ngroup, kslice, batch;
    tile_of_block(ngroup, kslice, batch);
    int Np = N, ldyp = ldy;
    bool w_tiled = bs.tiled != 0;
    if (gt.used) {                                     // per-problem operands; blockIdx.x runs over all problems' N-groups
        batch = 0;
        while (batch + 1 < gt.used && ngroup >= gt.first[batch + 1]) batch++;
        ngroup -= gt.first[batch];
        X = gt.X[batch], Wv = gt.W[batch], Y = gt.Y[batch], bias = gt.bias[batch], part = gt.part[batch];
        Np = gt.N[batch], ldyp = gt.ldy[batch];
        w_tiled = gt.tiled[batch] != 0;
    } else {
        X += batch * bs.x;
        Wv = static_cast<const unsigned char *>(Wv) + batch * bs.w * (W8 ? 1 : 2);
        if (Y) Y += batch * bs.y;
        if (bias) bias += batch * bs.bias;
        if (part) part += (int64_t)batch * gridDim.y * M * N;
    }
    const int n_base = ngroup * kBN;
    const int n0 = n_base + wave * 32;
    const bool wave_live = computes && n0 < Np;
    const int k_begin = kslice * k_slice;
    const int Kz = (batch < 8 && bs.k[batch] > 0) ? bs.k[batch] : K;
    const int k_end = (k_begin + k_slice) < Kz ? (k_begin + k_slice) : Kz;
    const int nkb = k_end > k_begin ? (k_end - k_begin) / kKB : 0;
    unsigned char *const xring = smem, *const wring = smem + XD * kXBytes;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mt][i] = 0.f;

#ifndef SKINNY_EXP
#define SKINNY_EXP 0   // ingest experiments (tools/exp_skinny_ingest.py): 1/2 = W / x always from the slice's first K-block,
#endif                 // 4 = fragments fetched but no MFMA
    auto stage_x = [&](int kb) {
        const int k0 = k_begin + ((SKINNY_EXP & 2) ? 0 : kb * kKB);
        unsigned char *base = xring + (kb % XD) * kXBytes;
#pragma unroll
        for (int i = 0; i < kXLoads; i++) {
            const int g = i * kLanes + lt;
            int m = g >> 3;
            const int lc = (g & 7) ^ ((m >> 1) & 7);
            m = m < M ? m : M - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(X + (int64_t)m * ldx + k0 + lc * 8),
                                             (lptr_t)(base + (i * kLanes + lw * 64) * 16), 16, 0, 0);
        }
    };
    auto stage_w = [&](int kb) {                       // weights: streamed once -> nt
        const int k0 = k_begin + ((SKINNY_EXP & 1) ? 0 : kb * kKB);
        unsigned char *base = wring + (kb % WD) * kWBytes;
#pragma unroll
        for (int i = 0; i < kWLoads; i++) {
            const int g = i * kLanes + lt;
            if constexpr (W8) {                        // 64-B rows: 4 chunks per row, chunk position ^ ((row>>2)&3)
                const int nr = g >> 2;
                const int lc = (g & 3) ^ ((nr >> 2) & 3);
                int n = n_base + nr;
                n = n < Np ? n : Np - 1;
                const uint8_t *src = w_tiled ? static_cast<const uint8_t *>(Wv) + ((int64_t)ngroup * (K / kKB) + k0 / kKB) * (kBN * kKB) + g * 16
                                             : static_cast<const uint8_t *>(Wv) + (int64_t)n * ldw + k0 + lc * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * kLanes + lw * 64) * 16), 16, 0, 2);
            } else {
                const int nr = g >> 3;
                const int lc = (g & 7) ^ ((nr >> 1) & 7);
                int n = n_base + nr;
                n = n < Np ? n : Np - 1;
                // tile-image layout: the 16 KiB image of (N-group, K-block) is stored contiguously in LDS order, so a
                // wave-instruction reads 1 KiB of consecutive bytes instead of eight 128-B pieces of eight rows
                const f16 *src = w_tiled ? static_cast<const f16 *>(Wv) + ((int64_t)ngroup * (K / kKB) + k0 / kKB) * (kBN * kKB) + g * 8
                                         : static_cast<const f16 *>(Wv) + (int64_t)n * ldw + k0 + lc * 8;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * kLanes + lw * 64) * 16), 16, 0, 2);
            }
        }
    };
    auto compute = [&](int kb) {
        const unsigned char *xt = xring + (kb % XD) * kXBytes;
        const unsigned char *wt = wring + (kb % WD) * kWBytes;
        const int wr = wave * 32 + r;
        f16x8 wf[4];
        if constexpr (W8) {
            // lane (r,h) needs bytes 32h .. 32h+31 of its row: chunks 2h and 2h+1
            const int sw = (wr >> 2) & 3;
            const u32x4 q0 = *reinterpret_cast<const u32x4 *>(wt + wr * 64 + (((2 * h) ^ sw) << 4));
            const u32x4 q1 = *reinterpret_cast<const u32x4 *>(wt + wr * 64 + (((2 * h + 1) ^ sw) << 4));
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const uint32_t lo = s < 2 ? q0[2 * s] : q1[2 * (s - 2)];
                const uint32_t hi = s < 2 ? q0[2 * s + 1] : q1[2 * (s - 2) + 1];
                const f16x2 a = cvt_u8x2(lo, 1), b = cvt_u8x2(lo, 0), c = cvt_u8x2(hi, 1), d = cvt_u8x2(hi, 0);
                wf[s] = (f16x8){a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) wf[s] = *reinterpret_cast<const f16x8 *>(wt + wr * 128 + (((4 * h + s) ^ ((wr >> 1) & 7)) << 4));
        }
        auto bfrag = [&](int s, int mt) {
            const int m = mt * 32 + r;
            return *reinterpret_cast<const f16x8 *>(xt + m * 128 + (((4 * h + s) ^ ((m >> 1) & 7)) << 4));
        };
#if SKINNY_EXP & 4
#define MMA(A, B, C) asm volatile("" ::"v"(A), "v"(B))   // exp 4: fragments fetched, no MFMA
#else
#define MMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif
        // x fragments one MFMA group ahead of their use.  The sched_barriers pin that order: left alone, the scheduler
        // sinks every ds_read to just before its MFMA (fewer live registers) and each MFMA then waits a full LDS latency.
        f16x8 bq[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bq[0][mt] = bfrag(0, mt);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s + 1 < 4) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) bq[(s + 1) & 1][mt] = bfrag(s + 1, mt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) MMA(wf[s], bq[s & 1][mt], acc[mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef MMA
    };

    // at step kb the x queue holds stages kb .. kb+XD-2 and the W queue kb .. kb+WD-2, each in issue order
#pragma unroll
    for (int p = 0; p < (XD > WD ? XD : WD) - 1; p++) {
        if (loads_x && p < XD - 1 && p < nkb) stage_x(p);
        if (loads_w && p < WD - 1 && p < nkb) stage_w(p);
    }
    for (int kb = 0; kb < nkb; kb++) {
        const int left = nkb - 1 - kb;
        const int ax = left < XD - 2 ? left : XD - 2, aw = left < WD - 2 ? left : WD - 2;   // younger stages already issued
        if constexpr (RM == 0) wait_stages_ahead<kXLoads + kWLoads>(ax);
        else if constexpr (RM == 2) { if (!computes) wait_stages_ahead<kXLoads + kWLoads>(ax); }
        else if (loads_x) wait_stages_ahead<kXLoads>(ax);
        else wait_stages_ahead<kWLoads>(aw);
        asm volatile("s_barrier" ::: "memory");        // every wave's share landed; the slots restaged below are no longer read
        if (loads_x && kb + XD - 1 < nkb) stage_x(kb + XD - 1);
        if (loads_w && kb + WD - 1 < nkb) stage_w(kb + WD - 1);
        if (RM != 2 || computes) compute(kb);
    }

    // ---- epilogue: accumulators -> LDS (row-major, 16-B shift per row) -> stores of whole 512-B (f16: 256-B) row pieces.
    // Straight from the MFMA layout every store instruction would touch 32 rows x 32 B; measured on the decode step,
    // those scattered stores cost ~8 us per launch (SKINNY_EXP 64: 8.00 -> 6.62 ms/step without them).
    if (!computes) return;                             // loader waves are done; finished waves do not count in s_barrier
    if ((SKINNY_EXP & 64) && M > 1) {                  // exp 64: no epilogue stores (every accumulator stays live)
        float t = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) t += acc[mt][i];
        if (t == 12345.678f) part[0] = t;
        return;
    }
    constexpr int kLd = kBN + 4;                       // floats per staged row: rows shift by 16 B -> conflict-free b128 writes
    float *stage = reinterpret_cast<float *>(smem);    // the ring is no longer needed (MT*32 rows x 528 B <= its size)
    __syncthreads();                                   // every compute wave is past its last fragment read
    if (wave_live) {
        // acc[mt][i]: m = mt*32 + (lane & 31), n = n0 + 8*(i>>2) + 4*h + (i&3)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int m = mt * 32 + r;
            if (m >= M) continue;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 o = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(stage + m * kLd + wave * 32 + 8 * g + 4 * h) = o;
            }
        }
    }
    __syncthreads();
    {
        const int c4 = tid & 31;                       // 4 columns per lane, 32 lanes per row, 8 rows per pass
        const int n = n_base + 4 * c4;
        if (n < Np) {
            for (int m = tid >> 5; m < M; m += 8) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 4 * c4);
                if (EPI == EPI_F16) {
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float t = bias ? v[e] + (float)bias[n + e] : v[e];
                        if (bs.relu_sq) {
                            t = (float)(f16)t;               // relu(fp16(y))**2, rwkv7.py:678
                            t = t > 0.f ? t * t : 0.f;
                        }
                        o[e] = (f16)t;
                    }
                    *reinterpret_cast<f16x4 *>(Y + (int64_t)m * ldyp + n) = o;
                } else {
                    *reinterpret_cast<f32x4 *>(part + ((int64_t)kslice * M + m) * Np + n) = v;
                }
            }
        }
    }
}


// Sum the split-K partials and apply the epilogue.
//   mode 0: y = sum (+ bias[n]);  mode 1: y = relu(sum (+bias))^2;
//   mode 2 (mm8): y = rx[n]*(sum + 0.5*S[m][0]) + S[m][1] + mx[n]*S[m][2]      (benchmark.py:167-179)
//   mode 3: mm8 then relu^2
//   mode 4 + p: the RWKV-7 LoRA hidden planes (v, w, a, g), first problem = plane p: tanh on w, sigmoid on g
//               (rwkv7.py:626, :630), applied to the binary16-rounded sum like the reference's separate op
// blockIdx.y = problem of a batched launch (partials [Z][splits][M][N], Y / bias advance by y_bs / bias_bs).
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const int M, const int N, const int splits,
                                                            const float *__restrict__ part, const f16 *__restrict__ bias,
                                                            const f16 *__restrict__ rx, const f16 *__restrict__ mx,
                                                            const float *__restrict__ S, const int mode,
                                                            f16 *__restrict__ Y, int ldy, const int64_t y_bs = 0,
                                                            const int64_t bias_bs = 0, const GroupTable gt = GroupTable{}) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int z = blockIdx.y;
    int N_ = N;
    const bool mm8 = !gt.used && (mode == 2 || mode == 3);
    int act = (mode == 1 || mode == 3) ? 1 : 0;        // 0 none, 1 relu^2, 2 tanh, 3 sigmoid
    if (mode >= 4) act = (z + mode - 4) == 1 ? 2 : ((z + mode - 4) == 3 ? 3 : 0);
    if (gt.used) {
        N_ = gt.N[z], ldy = gt.ldy[z], part = gt.part[z], Y = gt.Y[z], bias = gt.bias[z];
        act = gt.act[z];
    }
    const int64_t total = (int64_t)M * N_ / 4;
    if (gi >= total) return;
    if (!gt.used) {
        part += (int64_t)z * splits * M * N;
        Y += z * y_bs;
        if (bias) bias += z * bias_bs;
    }
    const int m = (int)(gi / (N_ / 4)), n = (int)(gi % (N_ / 4)) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; k++) s += *reinterpret_cast<const f32x4 *>(part + ((int64_t)k * M + m) * N_ + n);
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float v = s[e];
        if (mm8) v = (float)rx[n + e] * (v + 0.5f * S[m * 3 + 0]) + S[m * 3 + 1] + (float)mx[n + e] * S[m * 3 + 2];
        else if (bias) v += (float)bias[n + e];
        if (act == 1) {
            v = (float)(f16)v;                       // relu(fp16(y))**2, rwkv7.py:678
            v = v > 0.f ? v * v : 0.f;
        } else if (act == 2) {
            v = tanhf((float)(f16)v);
        } else if (act == 3) {
            v = 1.f / (1.f + __expf(-(float)(f16)v));
        }
        o[e] = (f16)v;
    }
    *reinterpret_cast<f16x4 *>(Y + (int64_t)m * ldy + n) = o;
}

// mm8 activation prologue: xs = fp16(x * ry), S[m] = {sum xs, sum x*my, sum x}   (benchmark.py:167-173)
__global__ __launch_bounds__(256) void mm8_prep_kernel(const int K, const f16 *__restrict__ x, const int ldx,
                                                       const f16 *__restrict__ ry, const f16 *__restrict__ my,
                                                       f16 *__restrict__ xs, float *__restrict__ S) {
    __shared__ float red[3][4];
    const int m = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x * 8; c < K; c += 256 * 8) {
        const f16x8 xv = *reinterpret_cast<const f16x8 *>(x + (int64_t)m * ldx + c);
        const f16x8 rv = *reinterpret_cast<const f16x8 *>(ry + c);
        const f16x8 mv = *reinterpret_cast<const f16x8 *>(my + c);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = (f16)((float)xv[e] * (float)rv[e]);
            s0 += (float)o[e];
            s1 += (float)xv[e] * (float)mv[e];
            s2 += (float)xv[e];
        }
        *reinterpret_cast<f16x8 *>(xs + (int64_t)m * K + c) = o;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s0 += __shfl_xor(s0, o, 64);
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
        red[2][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) S[m * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

template <bool W8>
int launch(int MT, bool partial, dim3 grid, size_t lds, hipStream_t st, int M, int N, int K, int k_slice, const f16 *X,
           int ldx, const void *W, int64_t ldw, f16 *Y, int ldy, const f16 *bias, float *part) {
#define GO(MTV)                                                                                                              \
    do {                                                                                                                     \
        if (lds > 65536) {                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_gemm_kernel<MTV, W8, EPI_PARTIAL>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_gemm_kernel<MTV, W8, EPI_F16>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
        }                                                                                                                    \
        if (partial)                                                                                                         \
            hipLaunchKernelGGL((skinny_gemm_kernel<MTV, W8, EPI_PARTIAL>), grid, dim3(kThreads), lds, st, M, N, K, k_slice, X, \
                               ldx, W, ldw, Y, ldy, bias, part);                                                             \
        else                                                                                                                 \
            hipLaunchKernelGGL((skinny_gemm_kernel<MTV, W8, EPI_F16>), grid, dim3(kThreads), lds, st, M, N, K, k_slice, X,    \
                               ldx, W, ldw, Y, ldy, bias, part);                                                             \
    } while (0)
    switch (MT) {
        case 1: GO(1); break;
        case 2: GO(2); break;
        case 3: GO(3); break;
        case 4: GO(4); break;
        case 5: GO(5); break;
        case 6: GO(6); break;
        case 7: GO(7); break;
        default: GO(8); break;
    }
#undef GO
    return (int)hipGetLastError();
}

int g_mode = 3;             // 0: register-staged kernel; 1-3: LDS-DMA ring kernel variants (launch_ring_mode); 3 measured fastest

int pick_splits(int N, int K, int requested, int Z = 1) {
    if (requested > 0) {                               // a request is honoured as far as K allows: whole, equal K-blocks
        int s = requested < K / kKB ? requested : K / kKB;
        while (s > 1 && (K / kKB) % s) s--;
        return s < 1 ? 1 : s;
    }
    const int ngroups = Z * ((N + kBN - 1) / kBN);
    int s = (256 + ngroups - 1) / ngroups;             // aim at >= 256 workgroups
    const int max_s = K / 256 > 0 ? K / 256 : 1;       // keep >= 4 K-blocks per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    while (s > 1 && (K / kKB) % s) s--;                // slices of whole K-blocks, equal size
    return s;
}

}  // namespace

// MODE 1: one ring, every wave loads both operands; 2: x / W loader roles, x 2 slots / W 6 (u8: 8); 3: dedicated loader waves
template <bool W8, int EPI, int MODE>
int launch_ring_mode(int MT, dim3 grid, hipStream_t st, int M, int N, int K, int k_slice, const f16 *X, int ldx, const void *W,
                     int64_t ldw, f16 *Y, int ldy, const f16 *bias, float *part, BatchStrides bs, const GroupTable &gt) {
    constexpr int RM = MODE - 1;
    constexpr int XD = MODE == 2 ? 2 : (W8 ? 4 : 3);
    constexpr int WD = MODE == 2 ? (W8 ? 8 : 6) : XD;
    const size_t ring = (size_t)XD * (MT * 32 * 128) + (size_t)WD * (kBN * (W8 ? 64 : 128));
    const size_t stage = (size_t)MT * 32 * (kBN + 4) * sizeof(float);     // the epilogue's row-major staging area reuses the ring
    const size_t lds = ring > stage ? ring : stage;
#define GO(MTV)                                                                                                           \
    do {                                                                                                                  \
        auto kern = skinny_gemm_ring_kernel<MTV, W8, EPI, XD, WD, RM>;                                                    \
        static bool lds_limit_raised[32] = {};     /* per instantiation AND device (one engine process drives several */  \
        int dev_ = 0;                              /* GPUs); the call is idempotent, a race is harmless */                 \
        (void)hipGetDevice(&dev_);                                                                                        \
        if (!lds_limit_raised[dev_ & 31]) {                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      (int)lds);                                                                          \
            lds_limit_raised[dev_ & 31] = true;                                                                           \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(RM == 2 ? 512 : kThreads), lds, st, M, N, K, k_slice, X, ldx, W, ldw, Y, ldy, bias, part, bs, gt); \
    } while (0)
    switch (MT) {
        case 1: GO(1); break;
        case 2: GO(2); break;
        case 3: GO(3); break;
        case 4: GO(4); break;
        case 5: GO(5); break;
        case 6: GO(6); break;
        case 7: GO(7); break;
        default: GO(8); break;
    }
#undef GO
    return (int)hipGetLastError();
}

template <bool W8, int EPI>
int launch_ring(int MT, dim3 grid, hipStream_t st, int M, int N, int K, int k_slice, const f16 *X, int ldx, const void *W,
                int64_t ldw, f16 *Y, int ldy, const f16 *bias, float *part, BatchStrides bs = BatchStrides{},
                const GroupTable &gt = GroupTable{}) {
    switch (g_mode) {
        case 2: return launch_ring_mode<W8, EPI, 2>(MT, grid, st, M, N, K, k_slice, X, ldx, W, ldw, Y, ldy, bias, part, bs, gt);
        case 3: return launch_ring_mode<W8, EPI, 3>(MT, grid, st, M, N, K, k_slice, X, ldx, W, ldw, Y, ldy, bias, part, bs, gt);
        default: return launch_ring_mode<W8, EPI, 1>(MT, grid, st, M, N, K, k_slice, X, ldx, W, ldw, Y, ldy, bias, part, bs, gt);
    }
}

namespace {
// W [N][K] row-major -> tile images: tile (N-group g, K-block b) = 1024 chunks of 16 B in the order the ring kernel
// keeps them in LDS (row nr = c >> 3 at chunk position c & 7 holds logical chunk (c & 7) ^ ((nr >> 1) & 7)).
__global__ __launch_bounds__(256) void tile_weight_kernel(const int N, const int K, const f16 *__restrict__ W, const int64_t ldw,
                                                          f16 *__restrict__ Wt) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * K / 8;
    if (c >= total) return;
    const int g = (int)(c & 1023);
    const int64_t tile = c >> 10;
    const int kb = (int)(tile % (K / kKB)), ng = (int)(tile / (K / kKB));
    const int nr = g >> 3, lc = (g & 7) ^ ((nr >> 1) & 7);
    *reinterpret_cast<f16x8 *>(Wt + c * 8) = *reinterpret_cast<const f16x8 *>(W + ((int64_t)ng * kBN + nr) * ldw + kb * kKB + lc * 8);
}
// uint8 form: tile (g, b) = 512 chunks of 16 B; row nr = c >> 2 at chunk position c & 3 holds logical chunk (c & 3) ^ ((nr >> 2) & 3)
__global__ __launch_bounds__(256) void tile_weight_u8_kernel(const int N, const int K, const uint8_t *__restrict__ W, const int64_t ldw,
                                                             uint8_t *__restrict__ Wt) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * K / 16;
    if (c >= total) return;
    const int g = (int)(c & 511);
    const int64_t tile = c >> 9;
    const int kb = (int)(tile % (K / kKB)), ng = (int)(tile / (K / kKB));
    const int nr = g >> 2, lc = (g & 3) ^ ((nr >> 2) & 3);
    *reinterpret_cast<u32x4 *>(Wt + c * 16) = *reinterpret_cast<const u32x4 *>(W + ((int64_t)ng * kBN + nr) * ldw + kb * kKB + lc * 16);
}
}  // namespace

// The same for the uint8 (mm8) weights wT [M_out][N_in] of mm8t_seq (w_tiled = 1 there): 8-KiB tile images.
extern "C" int skinny_tile_weight_u8(int N, int K, const void *W, int64_t ldw, void *Wt, void *stream) {
    if (N <= 0 || K <= 0 || (N % kBN) || (K % kKB) || ldw < K || (ldw & 15)) return CHIRRUP_E_SHAPE;
    if (!W || !Wt) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(Wt) & 15)) return CHIRRUP_E_ALIGN;
    const int64_t total = (int64_t)N * K / 16;
    hipLaunchKernelGGL(tile_weight_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, K, static_cast<const uint8_t *>(W), ldw, static_cast<uint8_t *>(Wt));
    return (int)hipGetLastError();
}

// Re-lay a binary16 weight matrix W [N][K] (N % 128 == 0, K % 64 == 0) as contiguous 16-KiB tile images for the ring
// kernel (w_tiled = 1 in the GEMM calls).  Wt needs N*K elements; W and Wt must not overlap.
extern "C" int skinny_tile_weight(int N, int K, const void *W, int64_t ldw, void *Wt, void *stream) {
    if (N <= 0 || K <= 0 || (N % kBN) || (K % kKB) || ldw < K || (ldw & 7)) return CHIRRUP_E_SHAPE;
    if (!W || !Wt) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(Wt) & 15)) return CHIRRUP_E_ALIGN;
    const int64_t total = (int64_t)N * K / 8;
    hipLaunchKernelGGL(tile_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, K, static_cast<const f16 *>(W), ldw, static_cast<f16 *>(Wt));
    return (int)hipGetLastError();
}

extern "C" int64_t skinny_gemm_workspace_bytes(int M, int N, int K, int splits) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = pick_splits(N, K, splits);
    return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

// Y = act(X . W^T + bias);  W binary16 [N][K] (row stride ldw).  act: 0 none, 1 relu^2.
extern "C" int skinny_gemm_f16(int M, int N, int K, const void *X, int ldx, const void *W, int64_t ldw, int w_tiled,
                               const void *bias, void *Y, int ldy, int act, int splits, void *workspace, void *stream) {
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 3) || (K % kKB) || ldx < K || ldw < K || ldy < N || (ldx & 7) || (ldw & 7) || (ldy & 3))
        return CHIRRUP_E_SHAPE;
    if (w_tiled && ((N % kBN) || !g_mode)) return CHIRRUP_E_UNSUPPORTED;
    if (!X || !W || !Y) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(Y) & 7))
        return CHIRRUP_E_ALIGN;
    const int s = pick_splits(N, K, splits);
    const bool partial = s > 1 || (act != 0 && !(g_mode && act == 1));     // unsplit relu^2 runs in the ring kernel's epilogue
    if (partial && !workspace) return CHIRRUP_E_NULL;
    const int MT = (M + 31) / 32;
    const int k_slice = K / s;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + kBN - 1) / kBN, s);
    const size_t lds = (size_t)2 * MT * 32 * 128;
    int rc;
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    bs.relu_sq = (!partial && act == 1) ? 1 : 0;
    if (g_mode)
        rc = partial ? launch_ring<false, EPI_PARTIAL>(MT, grid, st, M, N, K, k_slice, (const f16 *)X, ldx, W, ldw, (f16 *)Y,
                                                       ldy, (const f16 *)bias, (float *)workspace, bs)
                     : launch_ring<false, EPI_F16>(MT, grid, st, M, N, K, k_slice, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                   (const f16 *)bias, (float *)workspace, bs);
    else
        rc = launch<false>(MT, partial, grid, lds, st, M, N, K, k_slice, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                           (const f16 *)bias, (float *)workspace);
    if (rc) return rc;
    if (partial) {
        const int64_t total = (int64_t)M * N / 4;
        hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, M, N, s,
                           (const float *)workspace, (const f16 *)bias, nullptr, nullptr, nullptr, act ? 1 : 0, (f16 *)Y, ldy);
        rc = (int)hipGetLastError();
    }
    return rc;
}

extern "C" int64_t skinny_gemm_batched_workspace_bytes(int Z, int M, int N, int K, int splits) {
    if (Z <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)Z * pick_splits(N, K, splits, Z) * M * N * (int64_t)sizeof(float);
}

// Z independent problems in ONE launch: Y[z] = act(X[z] . W[z]^T + bias[z]); operands of problem z start z * (their
// batch stride, in elements) after problem 0.  act: 0 none, 1 relu^2, 4 + p = LoRA hidden planes starting at plane p.
extern "C" int skinny_gemm_f16_batched(int Z, int M, int N, int K, const void *X, int ldx, int64_t x_bs, const void *W,
                                       int64_t ldw, int64_t w_bs, const void *bias, int64_t bias_bs, void *Y, int ldy,
                                       int64_t y_bs, int act, int splits, void *workspace, void *stream) {
    return skinny_gemm_f16_grouped(Z, M, N, K, nullptr, X, ldx, x_bs, W, ldw, w_bs, bias, bias_bs, Y, ldy, y_bs, act, splits,
                                   workspace, stream);
}

// As skinny_gemm_f16_batched, with a reduction length per problem: problem z multiplies only the first k_of[z] columns
// of X[z] and W[z] (k_of[z] <= K, a multiple of 64; k_of == NULL: K for all).  For operands that are zero-padded to a
// common K (RWKV-7's LoRA ranks: 96 / 128 / 128 / 480 packed as 512) the padding is then never read.  Z <= 8, and
// splits must be 1 when k_of is given.
extern "C" int skinny_gemm_f16_grouped(int Z, int M, int N, int K, const int *k_of, const void *X, int ldx, int64_t x_bs,
                                       const void *W, int64_t ldw, int64_t w_bs, const void *bias, int64_t bias_bs, void *Y,
                                       int ldy, int64_t y_bs, int act, int splits, void *workspace, void *stream) {
    if (Z <= 0 || Z > 65535 || M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 3) || (K % kKB) || ldx < K || ldw < K || ldy < N ||
        (ldx & 7) || (ldw & 7) || (ldy & 3) || (x_bs & 7) || (w_bs & 7) || (y_bs & 3) || act < 0 || act > 7 || act == 2 || act == 3)
        return CHIRRUP_E_SHAPE;
    if (!X || !W || !Y) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(Y) & 7))
        return CHIRRUP_E_ALIGN;
    if (!g_mode) return CHIRRUP_E_UNSUPPORTED;            // the register-staged variant has no batch dimension
    const int s = pick_splits(N, K, splits, Z);
    const bool partial = s > 1 || act != 0;
    if (partial && !workspace) return CHIRRUP_E_NULL;
    const int MT = (M + 31) / 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + kBN - 1) / kBN, s, Z);
    BatchStrides bs{};
    bs.x = x_bs, bs.w = w_bs, bs.y = y_bs, bs.bias = bias_bs;
    if (k_of) {
        if (Z > 8 || splits != 1) return CHIRRUP_E_UNSUPPORTED;
        for (int z = 0; z < Z; z++) {
            if (k_of[z] <= 0 || k_of[z] > K || (k_of[z] % kKB)) return CHIRRUP_E_SHAPE;
            bs.k[z] = k_of[z];
        }
    }
    int rc = partial ? launch_ring<false, EPI_PARTIAL>(MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                       (const f16 *)bias, (float *)workspace, bs)
                     : launch_ring<false, EPI_F16>(MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                   (const f16 *)bias, (float *)workspace, bs);
    if (rc) return rc;
    if (partial) {
        const int64_t total = (int64_t)M * N / 4;
        hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((total + 255) / 256), Z), dim3(256), 0, st, M, N, s,
                           (const float *)workspace, (const f16 *)bias, nullptr, nullptr, nullptr, act, (f16 *)Y, ldy, y_bs,
                           bias_bs);
        rc = (int)hipGetLastError();
    }
    return rc;
}

extern "C" int64_t skinny_gemm_group_workspace_bytes(int count, const chirrup_gemm_problem *problems, int M, int splits) {
    if (count <= 0 || count > 8 || !problems || M <= 0 || splits <= 0) return 0;
    int64_t b = 0;
    for (int i = 0; i < count; i++) b += ((int64_t)splits * M * problems[i].n * (int64_t)sizeof(float) + 255) / 256 * 256;
    return b;
}

// Up to 8 GEMMs that share M, K, the row strides of x and W and the split count, in ONE launch (+ one reduce launch):
// y_i = act_i(x_i . w_i^T + bias_i).  Always goes through split-K partials (splits >= 1) so that the activations run in
// the reduce kernel; workgroups are dealt over the largest problem's N-groups, the smaller problems' spare ones exit.
extern "C" int skinny_gemm_f16_group(int count, const chirrup_gemm_problem *problems, int M, int K, int ldx, int64_t ldw,
                                     int splits, void *workspace, void *stream) {
    if (count <= 0 || count > 8 || !problems) return CHIRRUP_E_SHAPE;
    if (M <= 0 || M > 256 || K <= 0 || (K % kKB) || ldx < K || ldw < K || (ldx & 7) || (ldw & 7) || splits <= 0 ||
        ((K / kKB) % splits))
        return CHIRRUP_E_SHAPE;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255)) return workspace ? CHIRRUP_E_ALIGN : CHIRRUP_E_NULL;
    if (!g_mode) return CHIRRUP_E_UNSUPPORTED;
    GroupTable gt{};
    gt.used = count;
    int max_n = 0;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    for (int i = 0; i < count; i++) {
        const chirrup_gemm_problem &q = problems[i];
        if (q.n <= 0 || (q.n & 3) || q.ldy < q.n || (q.ldy & 3) || q.act < 0 || q.act > 3) return CHIRRUP_E_SHAPE;
        if (!q.x || !q.w || !q.y) return CHIRRUP_E_NULL;
        if ((reinterpret_cast<uintptr_t>(q.x) & 15) || (reinterpret_cast<uintptr_t>(q.w) & 15) || (reinterpret_cast<uintptr_t>(q.y) & 7))
            return CHIRRUP_E_ALIGN;
        gt.X[i] = static_cast<const f16 *>(q.x), gt.W[i] = q.w, gt.Y[i] = static_cast<f16 *>(q.y);
        gt.bias[i] = static_cast<const f16 *>(q.bias), gt.N[i] = q.n, gt.ldy[i] = q.ldy, gt.act[i] = q.act;
        if (q.w_tiled && (q.n % kBN)) return CHIRRUP_E_UNSUPPORTED;
        gt.tiled[i] = q.w_tiled ? 1 : 0;
        gt.part[i] = reinterpret_cast<float *>(ws);
        ws += ((int64_t)splits * M * q.n * (int64_t)sizeof(float) + 255) / 256 * 256;
        max_n = q.n > max_n ? q.n : max_n;
        gt.first[i + 1] = gt.first[i] + (q.n + kBN - 1) / kBN;
    }
    const int MT = (M + 31) / 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(gt.first[count], splits, 1);
    int rc = launch_ring<false, EPI_PARTIAL>(MT, grid, st, M, max_n, K, K / splits, gt.X[0], ldx, gt.W[0], ldw, gt.Y[0], gt.ldy[0],
                                             nullptr, gt.part[0], BatchStrides{}, gt);
    if (rc) return rc;
    const int64_t total = (int64_t)M * max_n / 4;
    hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((total + 255) / 256), count), dim3(256), 0, st, M, max_n, splits,
                       (const float *)nullptr, (const f16 *)nullptr, nullptr, nullptr, nullptr, 0, (f16 *)nullptr, 0, (int64_t)0,
                       (int64_t)0, gt);
    return (int)hipGetLastError();
}

extern "C" int skinny_gemm_f16_partial(int M, int N, int K, const void *X, int ldx, const void *W, int64_t ldw, int w_tiled,
                                       int splits, float *partials, void *stream) {
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 3) || (K % kKB) || ldx < K || ldw < K || (ldx & 7) || (ldw & 7))
        return CHIRRUP_E_SHAPE;
    if (!X || !W || !partials) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(partials) & 15))
        return CHIRRUP_E_ALIGN;
    const int s = pick_splits(N, K, splits);
    const int MT = (M + 31) / 32;
    const dim3 grid((N + kBN - 1) / kBN, s);
    if (w_tiled && ((N % kBN) || !g_mode)) return CHIRRUP_E_UNSUPPORTED;
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    const int rc = launch_ring<false, EPI_PARTIAL>(MT, grid, static_cast<hipStream_t>(stream), M, N, K, K / s, (const f16 *)X, ldx,
                                                   W, ldw, nullptr, N, nullptr, partials, bs);
    return rc ? -1000 - rc : s;
}

// mm8 with K-contiguous ("packed", [M_out][N_in]) uint8 weights.  Same quantisation and formula as
// mm8_seq; evaluated in the split form: xs = fp16(x*ry) through MFMA, rank-1 corrections after.
// workspace layout: xs [B][N_in] f16 | S [B][3] f32 | partials [splits][B][M_out] f32
extern "C" int64_t mm8t_workspace_bytes(int B, int N_in, int M_out, int splits) {
    if (B <= 0 || N_in <= 0 || M_out <= 0) return 0;
    if (B > 256) B = 256;                                  // more rows are processed 256 at a time
    const int s = pick_splits(M_out, N_in, splits);
    int64_t b = (int64_t)B * N_in * 2;
    b = (b + 255) / 256 * 256;
    b += 256 * ((B * 3 * 4 + 255) / 256);
    b += (int64_t)s * B * M_out * 4;
    return b;
}

extern "C" int mm8t_seq(int B, int N_in, int M_out, const void *x, int x_stride, const void *wT, int64_t w_stride, int w_tiled,
                        const void *mx, const void *rx, const void *my, const void *ry, void *y, int y_stride, int act,
                        int splits, void *workspace, void *stream) {
    if (w_tiled && ((M_out % kBN) || !g_mode)) return CHIRRUP_E_UNSUPPORTED;
    if (B <= 0 || N_in <= 0 || M_out <= 0 || (M_out & 3) || (N_in % kKB) || x_stride < N_in || w_stride < N_in ||
        y_stride < M_out || (x_stride & 7) || (w_stride & 15) || (y_stride & 3))
        return CHIRRUP_E_SHAPE;
    if (!x || !wT || !mx || !rx || !my || !ry || !y || !workspace) return CHIRRUP_E_NULL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wT) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return CHIRRUP_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int s = pick_splits(M_out, N_in, splits);
    const int Bmax = B < 256 ? B : 256;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    f16 *xs = reinterpret_cast<f16 *>(ws);
    int64_t off = ((int64_t)Bmax * N_in * 2 + 255) / 256 * 256;
    float *S = reinterpret_cast<float *>(ws + off);
    off += 256 * ((Bmax * 3 * 4 + 255) / 256);
    float *part = reinterpret_cast<float *>(ws + off);
    // The kernel holds at most 256 activation rows per weight pass (8 accumulator tiles per wave): a longer batch
    // (chunked prefill) is cut into 256-row blocks that re-stream the weights; the blocks reuse the workspace in
    // stream order.
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int bn = (B - b0) < 256 ? (B - b0) : 256;
        const f16 *xb = static_cast<const f16 *>(x) + (int64_t)b0 * x_stride;
        f16 *yb = static_cast<f16 *>(y) + (int64_t)b0 * y_stride;
        hipLaunchKernelGGL(mm8_prep_kernel, dim3(bn), dim3(256), 0, st, N_in, xb, x_stride, (const f16 *)ry, (const f16 *)my,
                           xs, S);
        const int MT = (bn + 31) / 32;
        const dim3 grid((M_out + kBN - 1) / kBN, s);
        const size_t lds = (size_t)2 * MT * 32 * 128;
        BatchStrides bs{};
        bs.tiled = w_tiled ? 1 : 0;
        int rc = g_mode ? launch_ring<true, EPI_PARTIAL>(MT, grid, st, bn, M_out, N_in, N_in / s, xs, N_in, wT, w_stride, yb,
                                                         y_stride, nullptr, part, bs)
                        : launch<true>(MT, true, grid, lds, st, bn, M_out, N_in, N_in / s, xs, N_in, wT, w_stride, yb, y_stride,
                                       nullptr, part);
        if (rc) return rc;
        const int64_t total = (int64_t)bn * M_out / 4;
        hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bn, M_out, s, part,
                           nullptr, (const f16 *)rx, (const f16 *)mx, S, act ? 3 : 2, yb, y_stride);
        rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return 0;
}

extern "C" void skinny_gemm_select(int mode) { g_mode = mode < 0 || mode > 3 ? 3 : mode; }
