// Skinny-M MFMA GEMMs for the decode step:  Y[M][N] = X[M][K] . W[N][K]^T   (M <= 256 token rows)
// with binary16 or uint8 (mm8) weights, binary32 accumulation.  One RWKV-7 layer at 1..256 token rows runs five launches
// of these kernels: R/K/V + the four LoRA down-projections (grouped), the four LoRA up-projections (batched, per-
// problem reduction length), att.output, ffn.key, ffn.value (DESIGN.md sections 4-5 have the measurements).
//
// Why hand-written: at M = 200 the library GEMMs (hipBLASLt through torch) stream weights at 1.4-2.8 TB/s and every
// projection is its own launch.  Here the batch is small enough that ONE workgroup holds all M rows of x for its K-block
// in LDS, and independent problems share a launch.  HBM traffic equals the algorithmic bytes (profiles/r02_gemm_pmc_traffic
// .json, r02b_...: the x re-reads are L2 hits); what bounds the launches is POWER -- the chip runs the main loop at
// 1.36 GHz, 2.1 GHz with either the MFMAs or the loads alone (profiles/r02_gemm_experiments.txt) -- so the levers are
// doing less work and spending less time outside the main loop, not overlapping more.
//
// Both kernels: a 64-wide K-block of BOTH operands goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) into rings --
// x image MT*16 rows x 128 B, W image BN rows x 128 B (u8: 64 B), XOR-swizzled on the SOURCE address so that the
// ds_read_b128 of the MFMA fragments are conflict-free.  No load of the main loop has a register destination, so hand-
// placed `s_waitcnt vmcnt(n)` + one raw `s_barrier` per K-block keep whole stages in flight across the barrier.
// A compute wave owns 32 rows of W as two 16-row tiles: v_mfma_f32_16x16x32_f16 with A = W tile (16 n x 32 k), B = x^T
// (32 k x 16 m); one x fragment feeds two MFMAs, one W fragment MT = ceil(M/16) of them (u8 -> f16 by v_perm in
// registers).  16-row tiles: M = 200 pads to 208 rows instead of 224 (-7 % MFMAs, fragment reads and x bytes), and the
// 16x16x32 shape delivers more flops per joule than 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7).
//
// ring_gemm_kernel (BN = 128; every GEMM of a layer): 4 compute + 4 dedicated loader waves, one ring for both operands, as
//     deep as 160 KiB of LDS allows (3 slots at 13 x tiles, 5 at 7: ring_depth).
// wide_gemm_kernel (BN = 256; the head, N >= 32768): 8 compute waves, two per SIMD; waves 0-3 also issue the x loads of the
//     NEXT K-block (L2-resident, 2 slots) at the start of an iteration, waves 4-7 the W loads two K-blocks ahead (HBM,
//     `nt`, 3 slots) in the MIDDLE of theirs -- a wave is blocked while its LDS-DMA instructions issue, and this way its
//     SIMD partner has MFMAs to run meanwhile.  Per W byte a workgroup ingests 0.875 B of x instead of 1.75 B at BN = 128.
//
// Common: split-K over blockIdx.y with binary32 partials, reduced by skinny_reduce_kernel (bias, relu^2, tanh / sigmoid
// of the LoRA planes, or the mm8 rank-1 corrections of scripts/test_mm8/benchmark.py:167-179) or by the NEXT layer-norm
// kernel (rwkv7_add_ln_mix, delta_partials); unsplit launches apply bias / activation (EPI_F16) or the whole mm8
// epilogue incl. the next product's prologue (EPI_MM8) themselves; ROW HALVES (BatchStrides::row_halves): two workgroups
// per tile and K-slice, one per half of the rows -- twice the workgroups without more partial planes, W streamed by both
// from one XCD's L2; batched (gridDim.z problems at uniform strides, optional per-problem K) and grouped launches (per-
// problem operands, N and output stride; blockIdx.x runs over an exact tile list); workgroup -> tile order is XCD-aware
// and grids are padded to a multiple of 8 workgroups so that it stays so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/chirrup_amd.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int kKB = 64;          // K-block
constexpr int kTileRows = 128;   // rows of one tile image (skinny_tile_weight): both kernels stage whole 128-row tiles

// EPI_MM8 (uint8 weights, unsplit): the rank-1 corrections of the mm8 split form, relu^2 and the activation prologue of the
// NEXT mm8 product in the GEMM's own epilogue (store_staged_mm8) -- no partials, no reduce launch
// EPI_PAIR (binary16 weights, split 2..4, few rows): the last of a tile's workgroups to finish adds the others' partials and its
// own sums in slice order and applies bias / activation -- no reduce launch (store_staged_sc1 / combine_store)
enum { EPI_F16 = 0, EPI_PARTIAL = 1, EPI_MM8 = 2, EPI_PAIR = 3 };

// two uint8 -> two binary16 values 1024 + b (exact): bytes b0,b1 -> halves 0x6400|b.  ONE v_perm per pair and no
// subtraction: the matrix cores multiply by (1024 + q) and the constant is taken out again with the other rank-1 terms of
// the mm8 split form, core = sum_k xs*(1024 + q) - 1024*S0 (kU8Offset below; every product xs*(1024+q) is exact in
// binary32, the sums carry 3 more bits of magnitude than without the offset -- ~1e-5 relative on the result, far inside
// the binary16 output rounding).
constexpr float kU8Offset = 1024.f;
__device__ __forceinline__ f16x2 cvt_u8x2(uint32_t packed, int sel_lo) {
    // v_perm_b32: selector bytes pick from {src0 (hi dword), src1 (lo dword)}; 0x64 bytes come from a constant
    const uint32_t magic = 0x64646464u;
    uint32_t r;
    if (sel_lo)  r = __builtin_amdgcn_perm(magic, packed, 0x04010400u);   // [b1,0x64 | b0,0x64] -> halves (1024+b0),(1024+b1)
    else         r = __builtin_amdgcn_perm(magic, packed, 0x04030402u);   // bytes 2,3
    return __builtin_bit_cast(f16x2, r);
}

// Workgroup -> (N-group, K-slice).  Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its
// own L2: give XCD j a contiguous run of the K-slice-major tile order, so the x K-slice a workgroup re-reads is shared
// by its L2 neighbours (ffn.value at bsz 200: x is 6.5 MB, a K-slice 0.8 MB; the L2 is 4 MB).
// halves = 2 (row halves, BatchStrides::row_halves): the two workgroups of a tile's K-slice are neighbours in that order --
// on the same XCD when the grid divides by 8 and dispatched together, so the W tile both stream is fetched from HBM once.
__device__ __forceinline__ void tile_of_block(int &ngroup, int &kslice, int &batch, int &half, const int halves, const bool pair) {
    const int G = gridDim.x, GS = G * (gridDim.y / halves), total = G * gridDim.y * gridDim.z;
    const int L = blockIdx.x + G * blockIdx.y + G * gridDim.y * blockIdx.z;
    int v = (total & 7) ? L : (L & 7) * (total >> 3) + (L >> 3);
    half = 0;
    if (halves == 2) {
        half = v & 1;
        v >>= 1;
    }
    if (pair) {                                        // the K-slices of a tile as neighbours (same XCD, dispatched together)
        const int S = gridDim.y;
        kslice = v % S;
        v /= S;
        batch = v / G;
        ngroup = v - batch * G;
        return;
    }
    batch = v / GS;
    v -= batch * GS;
    kslice = v / G;
    ngroup = v - kslice * G;
}

// element strides between the problems of a batched launch (gridDim.z problems; 0s for a single GEMM)
struct Mm8Epilogue {
    const f16 *rx, *mx;          // [N] scales of this product's weights
    const float *S;              // [M][S_parts][3] row sums of this product's prologue
    const f16 *ry2, *my2;        // [N] scales of the next product (its K = this N); NULL: no prologue
    f16 *xs2;                    // [M][N]
    float *S2;                   // [M][S2_parts][3]: one partial row sum per 128-column tile
    int S_parts, S2_parts, act;
};

struct BatchStrides {
    int64_t x, w, y, bias;
    int k[8];          // per-problem reduction length (<= K, multiple of 64) or 0 = K: zero-padded tails are not streamed
    int tiled;         // W of every problem is in the tile-image layout (skinny_tile_weight)
    int relu_sq;       // EPI_F16: y = relu(binary16(x.w + bias))^2 in the epilogue
    Mm8Epilogue q8;    // EPI_MM8
    int *counters;     // EPI_PAIR: one zero-initialised int per tile; the second arriver leaves it zero again
    unsigned long long *clock;   // diagnostic (skinny_gemm_clock_probe): per workgroup {shader-clock ticks, 100-MHz ticks} of the main loop
    unsigned long long *timeline;   // ... and, behind the pairs, 4 absolute 100-MHz stamps per workgroup: kernel entry, loop start, loop end, epilogue done
    int min_lds;       // host side only: at least this much dynamic LDS per workgroup (EPI_PAIR with a small ring: one workgroup per CU)
    int row_halves;    // 1: two workgroups per tile and K-slice, rows [0, m0) and [m0, M) of x / y, m0 = 16 * MT of the launch
                       //    (gridDim.y = 2 x splits): twice the workgroups WITHOUT more partials -- W is streamed by both
                       //    (one HBM fetch when they run side by side on one XCD), and with half the x image per stage the
                       //    ring is deeper
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

// s_waitcnt vmcnt(ahead * PER): leave the `ahead` youngest stages (PER LDS-DMA instructions each) in flight.  The
// counter field holds 0..63: a count that does not fit waits for fewer outstanding loads, which is always safe.
template <int PER>
__device__ __forceinline__ void wait_stages_ahead(const int ahead) {
#define WAIT_CASE(A)                                                                                            \
    case A:                                                                                                     \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * PER <= 63 ? (A) * PER : (63 / PER) * PER) : "memory"); \
        break;
    switch (ahead) {
        WAIT_CASE(7) WAIT_CASE(6) WAIT_CASE(5) WAIT_CASE(4) WAIT_CASE(3) WAIT_CASE(2) WAIT_CASE(1)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef WAIT_CASE
}

// What one workgroup works on, resolved from the launch arguments (plain / batched / grouped launch).
struct Tile {
    const f16 *X;
    const void *W;
    f16 *Y;
    const f16 *bias;
    float *part;
    int Np, ldy, ngroup, kslice, batch;
    int M;             // rows of x / y this workgroup works on (the launch's M, or its part under row_halves)
    int rows0;         // ... and its first row
    int pair_id;       // EPI_PAIR: index of the tile's counter
    int act;           // EPI_F16: 0 none, 1 relu^2, 2 tanh, 3 sigmoid (as skinny_reduce_kernel)
    bool w_tiled;
};

template <bool W8, int EPI, int MT>
__device__ __forceinline__ Tile resolve_tile(const int N, const f16 *X, const void *Wv, f16 *Y, const int ldy, const f16 *bias,
                                             float *part, const int M, const int ldx, const BatchStrides &bs, const GroupTable &gt) {
    Tile t;
    int half;
    tile_of_block(t.ngroup, t.kslice, t.batch, half, bs.row_halves ? 2 : 1, EPI == EPI_PAIR);
    t.pair_id = t.batch * gridDim.x + t.ngroup;
    t.X = X, t.W = Wv, t.Y = Y, t.bias = bias, t.part = part, t.Np = N, t.ldy = ldy, t.w_tiled = bs.tiled != 0;
    t.M = M, t.act = bs.relu_sq ? 1 : 0;
    const int rows0 = half ? MT * 16 : 0;              // (row_halves launches have M > 16 MT)
    if (bs.row_halves) t.M = half ? M - MT * 16 : MT * 16;
    t.rows0 = rows0;
    if (gt.used) {                                     // per-problem operands; blockIdx.x runs over all problems' N-groups
        int b = 0;
        while (b + 1 < gt.used && t.ngroup >= gt.first[b + 1]) b++;
        t.batch = b;
        t.ngroup -= gt.first[b];
        t.X = gt.X[b], t.W = gt.W[b], t.Y = gt.Y[b], t.bias = gt.bias[b], t.part = gt.part[b];
        t.Np = gt.N[b], t.ldy = gt.ldy[b];
        t.w_tiled = gt.tiled[b] != 0;
        t.act = gt.act[b];
    } else {
        t.X += t.batch * bs.x;
        t.W = static_cast<const unsigned char *>(Wv) + t.batch * bs.w * (W8 ? 1 : 2);
        if (Y) t.Y += t.batch * bs.y;
        if (bias) t.bias += t.batch * bs.bias;
        if (part) t.part += (int64_t)t.batch * (gridDim.y >> (bs.row_halves ? 1 : 0)) * M * N;
    }
    t.X += (int64_t)rows0 * ldx;
    if (t.Y) t.Y += (int64_t)rows0 * t.ldy;
    if (t.part) t.part += (int64_t)rows0 * t.Np;
    return t;
}

// LDS-DMA loads of both kernels: `buffer_load_dwordx4 ... lds` through a buffer descriptor -- one 32-bit lane offset per
// operand for the whole kernel, everything that changes per instruction in the scalar offset (a compute wave carries
// 26-32 accumulator tiles and has no registers to spare for 64-bit addresses; hoisted 64-bit address arithmetic of the
// global_load_lds form spilled), and rows past the end of an operand read as zeros (descriptor bounds) instead of
// needing a clamp.  One instruction moves 1 KiB per wave: a "round" = 256 lanes (4 waves) = 4 KiB.
//   x round i   : rows 32 i .. 32 i + 31 of the K-block image; chunk g = 256 i + lt is row m = g >> 3, logical 16-B chunk
//                 (g & 7) ^ ((m >> 1) & 7) -- the XOR term does not depend on i, so round i is round 0 plus 32 rows.
//   W round     : 32 rows (uint8: 64 rows) of a 128-row tile image; tile-image weights are read linearly (the image is
//                 stored in LDS order: 1 KiB of consecutive memory per instruction), row-major rows like x
//                 (uint8: 4 chunks per 64-B row, chunk position ^ ((row >> 2) & 3)).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes < 0xffffffffll ? (int)bytes : (int)0xffffffff, 0x00020000);
}

template <bool W8>
struct Loader {
    static constexpr int kEl = W8 ? 1 : 2;                       // bytes per weight element
    static constexpr int kWTile = kTileRows * kKB * kEl;         // one 128-row tile image: 16 KiB (uint8: 8 KiB)
    static constexpr int kRoundsPerTile = kWTile / 4096;
    static constexpr int kRowsPerRound = W8 ? 64 : 32;
    __amdgpu_buffer_rsrc_t xsrc, wsrc;
    int xoff, woff, ldx, K, Np;
    int64_t ldw;
    bool w_tiled;

    __device__ __forceinline__ Loader(const Tile &t, int M, int ldx_, int64_t ldw_, int K_, int lt)
        : xsrc(make_rsrc(t.X, (int64_t)M * ldx_ * 2)),
          wsrc(make_rsrc(t.W, t.w_tiled ? (int64_t)t.Np * K_ * kEl : (int64_t)t.Np * ldw_ * kEl)),
          ldx(ldx_), K(K_), Np(t.Np), ldw(ldw_), w_tiled(t.w_tiled) {
        xoff = ((lt >> 3) * ldx_ + (((lt & 7) ^ ((lt >> 4) & 7)) << 3)) * 2;
        const int wrow = W8 ? (lt >> 2) : (lt >> 3);
        const int wlc = W8 ? ((lt & 3) ^ ((wrow >> 2) & 3)) : ((lt & 7) ^ ((wrow >> 1) & 7));
        woff = t.w_tiled ? lt * 16 : (int)((wrow * ldw_) * kEl + wlc * 16);
    }
    // dst: the wave's 1 KiB of the round's 4 KiB in LDS (wave-uniform)
    __device__ __forceinline__ void x_round(int i, int k0, unsigned char *dst) const {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lptr_t)dst, 16, xoff, (i * 32 * ldx + k0) * 2, 0, 0);
    }
    __device__ __forceinline__ void w_round(int tile_row0, int round, int k0, unsigned char *dst) const {   // streamed once -> nt
        unsigned soff;
        if (w_tiled) {
            if (tile_row0 >= Np) tile_row0 = 0;        // (a 256-wide group whose second tile does not exist re-reads a valid one)
            soff = (unsigned)(((tile_row0 / kTileRows) * (K / kKB) + k0 / kKB) * kWTile + round * 4096);
        } else {
            soff = (unsigned)(((int64_t)(tile_row0 + round * kRowsPerRound) * ldw + k0) * kEl);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lptr_t)dst, 16, woff, (int)soff, 0, 2);
    }
};

// MFMA tiling of one K-block for one compute wave (32 W rows x MT 16-row tiles of x):
//   v_mfma_f32_16x16x32_f16: A = W[16 n][32 k], B = x^T[32 k][16 m]; lane l = (c = l & 15, q = l >> 4) supplies
//   A[n = c][k = 8q .. 8q+7] and B[k = 8q .. 8q+7][m = c], i.e. one 16-B chunk of a row of either LDS image; it receives
//   D[n = 4q + i][m = c] in accumulator register i.
// W fragments of lane (c, q): rows row0 + 16 nt + c (nt = 0, 1) of the 128-row tile image at wt, k-step ks = 0, 1.
template <bool W8>
__device__ __forceinline__ void read_w_frags(const unsigned char *wt, const int row0, const int c, const int q, f16x8 (&wf)[2][2]) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int row = row0 + 16 * nt + c;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            if constexpr (W8) {
                // bytes 32 ks + 8q .. +8 of a 64-B row: 16-B chunk 2 ks + (q >> 1) (position ^ ((row >> 2) & 3)), half q & 1
                const u32x2 v = *reinterpret_cast<const u32x2 *>(wt + row * 64 + (((2 * ks + (q >> 1)) ^ ((row >> 2) & 3)) << 4) + (q & 1) * 8);
                const f16x2 a = cvt_u8x2(v[0], 1), b = cvt_u8x2(v[0], 0), cc = cvt_u8x2(v[1], 1), d = cvt_u8x2(v[1], 0);
                wf[nt][ks] = (f16x8){a.x, a.y, b.x, b.y, cc.x, cc.y, d.x, d.y};
            } else {
                wf[nt][ks] = *reinterpret_cast<const f16x8 *>(wt + row * 128 + (((4 * ks + q) ^ ((row >> 1) & 7)) << 4));
            }
        }
    }
}

// One K-block of MFMAs for one wave: acc[nt][mt] += W tile nt . x tile mt over 64 k.  The x fragments are fetched one
// group (half of the m-tiles of one k-step) ahead of the MFMAs that use them; the sched_barriers pin that order -- left
// alone, the scheduler sinks every ds_read to just before its MFMA (fewer live registers) and each MFMA then waits a full
// LDS latency.  `mid()` runs between the two k-steps (the wide kernel issues its W loads there).
// Diagnostic builds only (make ablate A=<bits>; tools/ablate_gemm.sh): RING_ABLATE bit 0 = no MFMAs (the fragments are still read),
// bit 1 = the loader waves issue no loads, bit 2 = no fragment reads (the MFMAs run on whatever the registers hold), bit 3 = the
// epilogue's stores non-temporal.
#ifndef RING_ABLATE
#define RING_ABLATE 0
#endif
#ifndef RING_INTERLEAVE
#define RING_INTERLEAVE 1
#endif
#ifndef RING_DIRECT_PARTIAL
#define RING_DIRECT_PARTIAL 0
#endif
template <int MT, typename Mid>
__device__ __forceinline__ void mma_kblock(const unsigned char *xt, const f16x8 (&wf)[2][2], const int c, const int q, f32x4 (&acc)[2][MT],
                                           Mid &&mid) {
    constexpr int HA = (MT + 1) / 2, HB = MT - HA;     // m-tiles of the two groups of a k-step
    // IL: the next group's fragment reads go out one behind every tile's two MFMAs (in their shadow) instead of as a burst in
    // front of the group -- a burst of 6-7 ds_read_b128 costs the wave ~100 issue cycles in which no MFMA starts.  Compute waves
    // alone (no loads, 200 rows): ffn.key 22.9 -> 21.2 us, ffn.value 19.4 -> 17.4, their uint8 forms 26.6 -> 24.8 and 21.4 -> 19.0;
    // launches: uint8 ffn.value -2.4 us, uint8 ffn.key -1.0, binary16 +-0 (ingest-bound); below 5 tiles it costs 0.8 % of a step
    constexpr bool IL = RING_INTERLEAVE && MT >= 5;
    auto xfrag = [&](int ks, int mt) {
        const int m = mt * 16 + c;
#if RING_ABLATE & 4
        return f16x8{(f16)m, (f16)ks, 0, 0, 0, 0, 0, 0};
#else
        return *reinterpret_cast<const f16x8 *>(xt + m * 128 + (((4 * ks + q) ^ ((m >> 1) & 7)) << 4));
#endif
    };
    f16x8 bq[2][HA];
#pragma unroll
    for (int j = 0; j < HA; j++) bq[0][j] = xfrag(0, j);
#pragma unroll
    for (int g = 0; g < 4; g++) {                      // group g: k-step g >> 1, m-tiles [first, first + count)
        const int ks = g >> 1, first = (g & 1) ? HA : 0, count = (g & 1) ? HB : HA;
        const int nks = (g + 1) >> 1, nfirst = ((g + 1) & 1) ? HA : 0, ncount = g + 1 < 4 ? (((g + 1) & 1) ? HB : HA) : 0;
        if constexpr (!IL) {
#pragma unroll
            for (int j = 0; j < ncount; j++) bq[(g + 1) & 1][j] = xfrag(nks, nfirst + j);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g == 2) {
            mid();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < count; j++) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
#if RING_ABLATE & 1
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                asm volatile("" ::"v"(__builtin_bit_cast(i32x4, wf[nt][ks])), "v"(__builtin_bit_cast(i32x4, bq[g & 1][j])));
#else
                acc[nt][first + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt][ks], bq[g & 1][j], acc[nt][first + j], 0, 0, 0);
#endif
            }
            if constexpr (IL) {
                __builtin_amdgcn_sched_barrier(0);
                if (j < ncount) bq[(g + 1) & 1][j] = xfrag(nks, nfirst + j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (IL) {
#pragma unroll
            for (int jj = count; jj < ncount; jj++) bq[(g + 1) & 1][jj] = xfrag(nks, nfirst + jj);     // (the next group is the larger one)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Epilogue shared by both kernels: `stage` holds rows m of 128 binary32 columns (kLd floats apart); the threads of the
// workgroup store them as whole 512-B (f16: 256-B) row pieces.  Straight from the MFMA layout every store instruction
// would touch 32 rows x 32 B (measured: ~8 us per launch of scattered stores).
constexpr int kLd = kTileRows + 4;                     // floats per staged row: rows shift by 16 B -> conflict-free b128 writes
__device__ __forceinline__ float apply_act(float v, const int act) {
    if (act == 1) {
        v = (float)(f16)v;                           // relu(fp16(y))**2, rwkv7.py:678
        return v > 0.f ? v * v : 0.f;
    }
    if (act == 2) return tanhf((float)(f16)v);       // the LoRA hidden planes (rwkv7.py:626, :630), as skinny_reduce_kernel
    if (act == 3) return 1.f / (1.f + __expf(-(float)(f16)v));
    return v;
}
// M: rows this workgroup holds; plane_rows: rows of one K-slice's partial plane (the launch's M)
template <int EPI, int THREADS>
__device__ __forceinline__ void store_staged(const float *stage, const int M, const int n_first, const Tile &t, const int kslice,
                                             const int plane_rows) {
    const int tid = threadIdx.x;
    if (EPI == EPI_F16 && !(t.Np & 7) && !(t.ldy & 7) && !((reinterpret_cast<uintptr_t>(t.Y) | reinterpret_cast<uintptr_t>(t.bias)) & 15)) {
        // 8 columns per lane, 16 lanes per row: 16-B stores (8-B stores run at 0.54-0.70x their rate, MI355X_MICROARCH.md)
        const int c8 = tid & 15;
        const int n = n_first + 8 * c8;
        if (n >= t.Np) return;
        f16x8 bv = {};
        if (t.bias) bv = *reinterpret_cast<const f16x8 *>(t.bias + n);      // (bias + n: 16-B aligned for an aligned bias)
        for (int m = tid >> 4; m < M; m += THREADS / 16) {
            const f32x4 va = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 8 * c8);
            const f32x4 vb = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 8 * c8 + 4);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float v = e < 4 ? va[e] : vb[e - 4];
                o[e] = (f16)apply_act(t.bias ? v + (float)bv[e] : v, t.act);
            }
#if RING_ABLATE & 8
            __builtin_nontemporal_store(o, reinterpret_cast<f16x8 *>(t.Y + (int64_t)m * t.ldy + n));
#else
            *reinterpret_cast<f16x8 *>(t.Y + (int64_t)m * t.ldy + n) = o;
#endif
        }
        return;
    }
    const int c4 = tid & 31;                           // 4 columns per lane, 32 lanes per row, THREADS/32 rows per pass
    const int n = n_first + 4 * c4;
    if (n >= t.Np) return;
    for (int m = tid >> 5; m < M; m += THREADS / 32) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 4 * c4);
        if (EPI == EPI_F16) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (f16)apply_act(t.bias ? v[e] + (float)t.bias[n + e] : v[e], t.act);
            *reinterpret_cast<f16x4 *>(t.Y + (int64_t)m * t.ldy + n) = o;
        } else {
#if RING_ABLATE & 8
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(t.part + ((int64_t)kslice * plane_rows + m) * t.Np + n));
#else
            *reinterpret_cast<f32x4 *>(t.part + ((int64_t)kslice * plane_rows + m) * t.Np + n) = v;
#endif
        }
    }
}

// EPI_MM8: `stage` holds the core sums sum_k xs*(1024 + q) of rows m x 128 columns.  Per element, the arithmetic of
// mm8_reduce_rows_kernel: y = rx*(core - 1023.5*S0) + S1 + mx*S2 (benchmark.py:174-179 for kernels that multiply by 1024 + q),
// relu(fp16(y))^2 (rwkv7.py:678), written to Y if given; and for the next mm8 product xs2 = fp16(y*ry2) and this tile's share
// of its row sums {sum xs2, sum y*my2, sum y} -> S2[row][tile] (the consumer adds the tiles up in order).  A row of the
// tile is handled by 16 lanes x 8 columns -- one DPP row, so the row sums are four row shifts at VALU rate, and the stores
// are 16 B wide (measured per launch at 7.2B / bsz 200: partial epilogue 39.6 us; this with a 32-lane ds_bpermute butterfly
// and 8-B stores 47.6, with DPP 44.5, with 16 lanes per row 42.0; the reduce launch it replaces: 7.8).
// lane i of each 16-lane row receives v of lane i + n (0 past the row's end)
__device__ __forceinline__ float dpp_row_shl(const float v, const int n) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (n) {
        case 8: r = __builtin_amdgcn_update_dpp(0, x, 0x108, 0xf, 0xf, true); break;
        case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0xf, true); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x102, 0xf, 0xf, true); break;
        default: r = __builtin_amdgcn_update_dpp(0, x, 0x101, 0xf, 0xf, true); break;
    }
    return __builtin_bit_cast(float, r);
}
template <int THREADS>
__device__ __forceinline__ void store_staged_mm8(const float *stage, const int M, const int n_first, const Tile &t, const Mm8Epilogue &e8) {
    const int tid = threadIdx.x;
    const int c8 = tid & 15;                           // 8 columns per lane: a row of the tile is one 16-lane DPP row
    const int n = n_first + 8 * c8;
    const bool live = n < t.Np;                        // (N % 4 == 0, and N % 8 == 0 is checked by the entry point)
    f16x8 rxv = {}, mxv = {}, ryv = {}, myv = {};
    if (live) {
        rxv = *reinterpret_cast<const f16x8 *>(e8.rx + n), mxv = *reinterpret_cast<const f16x8 *>(e8.mx + n);
        if (e8.xs2) ryv = *reinterpret_cast<const f16x8 *>(e8.ry2 + n), myv = *reinterpret_cast<const f16x8 *>(e8.my2 + n);
    }
    const int tile = n_first / kTileRows;
    // this product's row sums, parts added up once, into the padding columns of the staged rows (a load per row inside
    // the loop below is a memory latency per trip: +9 us per launch, measured)
    for (int m = tid; m < M; m += THREADS) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int p = 0; p < e8.S_parts; p++) {
            const float *sp = e8.S + ((int64_t)(t.rows0 + m) * e8.S_parts + p) * 3;
            s0 += sp[0], s1 += sp[1], s2 += sp[2];
        }
        float *pad = const_cast<float *>(stage) + m * kLd + kTileRows;
        pad[0] = s0, pad[1] = s1, pad[2] = s2;
    }
    __syncthreads();
    for (int m = tid >> 4; m < M; m += THREADS / 16) {
        const int row = t.rows0 + m;
        const float s0 = stage[m * kLd + kTileRows], s1 = stage[m * kLd + kTileRows + 1], s2 = stage[m * kLd + kTileRows + 2];
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        if (live) {
            const f32x4 va = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 8 * c8);
            const f32x4 vb = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 8 * c8 + 4);
            f16x8 o, xs;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float y = (float)rxv[e] * ((e < 4 ? va[e] : vb[e - 4]) - (kU8Offset - 0.5f) * s0) + s1 + (float)mxv[e] * s2;
                if (e8.act) {
                    y = (float)(f16)y;
                    y = y > 0.f ? y * y : 0.f;
                }
                o[e] = (f16)y;
                xs[e] = (f16)((float)o[e] * (float)ryv[e]);
                t0 += (float)xs[e];
                t1 += (float)o[e] * (float)myv[e];
                t2 += (float)o[e];
            }
            if (t.Y) *reinterpret_cast<f16x8 *>(t.Y + (int64_t)m * t.ldy + n) = o;      // (t.Y starts at this workgroup's first row)
            if (e8.xs2) *reinterpret_cast<f16x8 *>(e8.xs2 + (int64_t)row * t.Np + n) = xs;
        }
        if (e8.xs2) {
#pragma unroll
            for (int sh = 8; sh >= 1; sh >>= 1) {      // 16 lanes -> their first lane by DPP row shifts (VALU rate; a
                t0 += dpp_row_shl(t0, sh);              // ds_bpermute butterfly is dependent LDS round trips)
                t1 += dpp_row_shl(t1, sh);
                t2 += dpp_row_shl(t2, sh);
            }
            if (c8 == 0) {
                float *dst = e8.S2 + ((int64_t)row * e8.S2_parts + tile) * 3;
                dst[0] = t0, dst[1] = t1, dst[2] = t2;
            }
        }
    }
}

// EPI_PAIR hand-off of a partial between the two workgroups of a tile (MI355X_MICROARCH.md, inter-workgroup visibility, first
// row of the table of hand-offs without fences): every byte of the partial is stored write-through (`sc1`), every storing
// wave drains its stores, the workgroup's barrier, then ONE lane adds to the tile's counter (agent scope); the workgroup whose
// add came second loads the other partial with `sc1` loads only (they bypass this CU's L1; nothing of it can be in this XCD's
// L2 -- sc1 stores drop the line).  a + b is the same binary32 value in either order, so the result does not depend on which
// workgroup was last and equals the reduce launch's bit for bit.  Pays for small slabs only (M <= 32: 16 KB per slice and tile;
// at M = 200 the lone last arriver needs as long as the whole reduce launch, profiles/r02_gemm_experiments.txt sections 9, 11).
constexpr int kPairMaxSlices = 4;
template <int THREADS>
__device__ __forceinline__ void store_staged_sc1(const float *stage, const int M, const int n_first, const Tile &t, const int kslice) {
    const int tid = threadIdx.x;
    const int c4 = tid & 31;
    const int n = n_first + 4 * c4;
    if (n >= t.Np) return;
    const __amdgpu_buffer_rsrc_t dst = make_rsrc(t.part + (int64_t)kslice * M * t.Np, (int64_t)M * t.Np * 4);
    for (int m = tid >> 5; m < M; m += THREADS / 32) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + m * kLd + 4 * c4);
        __builtin_amdgcn_raw_buffer_store_b128(v, dst, (m * t.Np + n) * 4, 0, 16);        // aux 16 = sc1
    }
}
// last arriver: y = epilogue(sum of the K-slices' partials in slice order, its own from the staged sums, the others from memory)
template <int THREADS>
__device__ __forceinline__ void combine_store(const float *stage, const int M, const int n_first, const Tile &t, const int own_slice,
                                              const int slices) {
    const int tid = threadIdx.x;
    const int c4 = tid & 31;
    const int n = n_first + 4 * c4;
    if (n >= t.Np) return;
    constexpr int RP = THREADS / 32;                   // rows per pass
    const __amdgpu_buffer_rsrc_t src = make_rsrc(t.part, (int64_t)slices * M * t.Np * 4);
    f16x4 bv = {};
    if (t.bias) bv = *reinterpret_cast<const f16x4 *>(t.bias + n);
    for (int m0 = tid >> 5; m0 < M; m0 += RP * 2) {    // two rows x up to three other slices in flight per lane
        f32x4 oth[2][kPairMaxSlices];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int k = 0; k < kPairMaxSlices; k++)
                if (k < slices && k != own_slice)      // (rows past M: within the descriptor, never used)
                    oth[u][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, ((k * M + m0 + u * RP) * t.Np + n) * 4, 0, 16));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int m = m0 + u * RP;
            if (m >= M) break;
            const f32x4 own = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 4 * c4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};            // 0 + slice 0 + slice 1 + ...: the reduce kernel's order, also for signed zeros
#pragma unroll
            for (int k = 0; k < kPairMaxSlices; k++)
                if (k < slices) v += (k == own_slice) ? own : oth[u][k];
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (f16)apply_act(t.bias ? v[e] + (float)bv[e] : v[e], t.act);
            *reinterpret_cast<f16x4 *>(t.Y + (int64_t)m * t.ldy + n) = o;
        }
    }
}

// ... for a uint8 (mm8) product: the slices' sums are core sums sum_k xs*(1024 + q); the last arriver applies the rank-1
// corrections y = rx*(core - 1023.5*S0) + S1 + mx*S2 (store_staged_mm8's arithmetic; S [M][3], one part per row)
template <int THREADS>
__device__ __forceinline__ void combine_store_mm8(const float *stage, const int M, const int n_first, const Tile &t, const int own_slice,
                                                  const int slices, const f16 *rx, const f16 *mx, const float *S) {
    const int tid = threadIdx.x;
    const int c4 = tid & 31;
    const int n = n_first + 4 * c4;
    if (n >= t.Np) return;
    constexpr int RP = THREADS / 32;
    const __amdgpu_buffer_rsrc_t src = make_rsrc(t.part, (int64_t)slices * M * t.Np * 4);
    const f16x4 rxv = *reinterpret_cast<const f16x4 *>(rx + n), mxv = *reinterpret_cast<const f16x4 *>(mx + n);
    for (int m0 = tid >> 5; m0 < M; m0 += RP * 2) {
        f32x4 oth[2][kPairMaxSlices];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int k = 0; k < kPairMaxSlices; k++)
                if (k < slices && k != own_slice)
                    oth[u][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, ((k * M + m0 + u * RP) * t.Np + n) * 4, 0, 16));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int m = m0 + u * RP;
            if (m >= M) break;
            const f32x4 own = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 4 * c4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kPairMaxSlices; k++)
                if (k < slices) v += (k == own_slice) ? own : oth[u][k];
            const float *sp = S + (int64_t)(t.rows0 + m) * 3;
            const float s0 = sp[0], s1 = sp[1], s2 = sp[2];
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (f16)((float)rxv[e] * (v[e] - (kU8Offset - 0.5f) * s0) + s1 + (float)mxv[e] * s2);
            *reinterpret_cast<f16x4 *>(t.Y + (int64_t)m * t.ldy + n) = o;
        }
    }
}

// ... for a uint8 product whose epilogue is store_staged_mm8 (corrections, relu^2, the next product's prologue): the last
// arriver first makes its staged sums the sums of ALL slices (slice order, the others' by sc1 loads), then runs that epilogue
template <int THREADS>
__device__ __forceinline__ void combine_into_stage(float *stage, const int M, const int n_first, const Tile &t, const int own_slice, const int slices) {
    const int tid = threadIdx.x;
    const int c4 = tid & 31;
    const int n = n_first + 4 * c4;
    if (n >= t.Np) return;
    constexpr int RP = THREADS / 32;
    const __amdgpu_buffer_rsrc_t src = make_rsrc(t.part, (int64_t)slices * M * t.Np * 4);
    for (int m0 = tid >> 5; m0 < M; m0 += RP * 2) {
        f32x4 oth[2][kPairMaxSlices];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int k = 0; k < kPairMaxSlices; k++)
                if (k < slices && k != own_slice)
                    oth[u][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, ((k * M + m0 + u * RP) * t.Np + n) * 4, 0, 16));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int m = m0 + u * RP;
            if (m >= M) break;
            const f32x4 own = *reinterpret_cast<const f32x4 *>(stage + m * kLd + 4 * c4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};            // 0 + slice 0 + slice 1 + ...: mm8_reduce_rows' order
#pragma unroll
            for (int k = 0; k < kPairMaxSlices; k++)
                if (k < slices) v += (k == own_slice) ? own : oth[u][k];
            *reinterpret_cast<f32x4 *>(stage + m * kLd + 4 * c4) = v;
        }
    }
}

// accumulators of the wave that owns staged columns col0 .. col0+31: acc[nt][mt][i] is m = 16 mt + c, n = col0 + 16 nt + 4q + i
template <int MT>
__device__ __forceinline__ void stage_acc(float *stage, const f32x4 (&acc)[2][MT], const int M, const int col0, const int c, const int q) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int m = mt * 16 + c;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4 *>(stage + m * kLd + col0 + 16 * nt + 4 * q) = acc[nt][mt];
    }
}

// ------------------------------------------------------------------------------------------------
// BN = 128: four compute waves + four dedicated loader waves, one ring of D slots for both operands (each loader wave
// issues its quarter of the x and of the W stage and waits for it with one counted vmcnt).  MT = 16-row tiles of x.
// (Measured and dropped, profiles/r02_gemm_experiments.txt: x and W loader roles on two waves each with a deeper W ring
// -- 6x slower, a loader wave that has to wait for its whole x stage every K-block is the critical path.)
// Ring slots of the BN = 128 kernel: what fits in 160 KiB, at most 6 (at M = 200: 3, uint8 weights 4; half the rows: 5)
template <int MT, bool W8>
constexpr int ring_depth() {
    constexpr int stage = MT * 16 * 128 + kTileRows * (W8 ? 64 : 128);
    constexpr int d = (160 * 1024) / stage;
    return d > 6 ? 6 : d;
}

template <int MT, bool W8, int EPI>
__global__ __launch_bounds__(512) void ring_gemm_kernel(
    const int M, const int N, const int K, const int k_slice, const f16 *__restrict__ X, const int ldx,
    const void *__restrict__ Wv, const int64_t ldw, f16 *__restrict__ Y, const int ldy,
    const f16 *__restrict__ bias, float *__restrict__ part, const BatchStrides bs, const GroupTable gt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BN = 128, D = ring_depth<MT, W8>();
    constexpr int kXBytes = MT * 16 * 128;            // x K-block image
    constexpr int kWBytes = BN * (W8 ? 64 : 128);     // W K-block image: 128 rows x 64 k (binary16 or uint8)
    constexpr int kXRounds = (MT + 1) / 2;            // rounds of 256 lanes x 16 B = 32 rows; the last one is half a round when MT is odd
    constexpr int kWLoads = kWBytes / 16 / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const bool computes = wave < 4;
    const int lt = tid & 255, lw = wave & 3;
    const bool short_x = (MT & 1) && lw >= 2;         // this loader wave sits out the half round
    const unsigned long long t_entry = bs.timeline ? __builtin_amdgcn_s_memrealtime() : 0;
    const Tile t = resolve_tile<W8, EPI, MT>(N, X, Wv, Y, ldy, bias, part, M, ldx, bs, gt);
    const int n_base = t.ngroup * BN;
    if (n_base >= t.Np) return;                        // padding workgroup (launch_gemm rounds the grid up for the XCD map)
    const int n0 = n_base + wave * 32;
    const bool wave_live = computes && n0 < t.Np;
    const int k_begin = t.kslice * k_slice;
    const int Kz = (t.batch < 8 && bs.k[t.batch] > 0) ? bs.k[t.batch] : K;
    const int k_end = (k_begin + k_slice) < Kz ? (k_begin + k_slice) : Kz;
    const int nkb = k_end > k_begin ? (k_end - k_begin) / kKB : 0;
    unsigned char *const xring = smem, *const wring = smem + D * kXBytes;

    f32x4 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const Loader<W8> ld(t, t.M, ldx, ldw, K, lt);
    auto stage = [&](int kb) {
#if !(RING_ABLATE & 2)
        const int k0 = k_begin + kb * kKB;
        unsigned char *xb = xring + (kb % D) * kXBytes, *wb = wring + (kb % D) * kWBytes;
#pragma unroll
        for (int i = 0; i < kXRounds; i++) {
            if (i == kXRounds - 1 && short_x) break;
            ld.x_round(i, k0, xb + (i * 256 + lw * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < kWLoads; i++) ld.w_round(n_base, i, k0, wb + (i * 256 + lw * 64) * 16);
#endif
    };

    // at step kb the ring holds stages kb .. kb+D-2, in issue order
    if (!computes) {
#pragma unroll
        for (int p = 0; p < D - 1; p++)
            if (p < nkb) stage(p);
    }
    unsigned long long clk0 = 0, rt0 = 0;
    if (bs.clock) clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    // One loop per role, the same number of barriers in both: with the two roles as branches of ONE loop body the accumulators
    // were merged at the loop latch (the loader waves "carry" theirs through), and for every tile count but 7 the register
    // allocator paid for that merge with a copy of every accumulator register per K-block -- 103 v_mov per 52 MFMAs at 13
    // tiles, 412 of the compute waves' ~2000 cycles per K-block (profiles/r03_gemm_main_loop_bound.txt).
    if (!computes) {
        for (int kb = 0; kb < nkb; kb++) {
            const int left = nkb - 1 - kb, ahead = left < D - 2 ? left : D - 2;
            if (short_x) wait_stages_ahead<kXRounds - 1 + kWLoads>(ahead);
            else wait_stages_ahead<kXRounds + kWLoads>(ahead);
            asm volatile("s_barrier" ::: "memory");    // every loader's share landed; the slot restaged below is no longer read
            if (kb + D - 1 < nkb) stage(kb + D - 1);
        }
    } else {
        for (int kb = 0; kb < nkb; kb++) {
            asm volatile("s_barrier" ::: "memory");
            f16x8 wf[2][2];
#if RING_ABLATE & 4
            for (int a_ = 0; a_ < 2; a_++)
                for (int b_ = 0; b_ < 2; b_++) wf[a_][b_] = f16x8{(f16)kb, (f16)c, 0, 0, 0, 0, 0, 0};
#else
            read_w_frags<W8>(wring + (kb % D) * kWBytes, wave * 32, c, q, wf);
#endif
            mma_kblock<MT>(xring + (kb % D) * kXBytes, wf, c, q, acc, [] {});
        }
    }
    if (!computes) return;                             // loader waves are done; finished waves do not count in s_barrier
    if (bs.clock && tid == 0) {
        const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        bs.clock[2 * wg] = __builtin_amdgcn_s_memtime() - clk0, bs.clock[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        if (bs.timeline) bs.timeline[4 * wg] = t_entry, bs.timeline[4 * wg + 1] = rt0, bs.timeline[4 * wg + 2] = __builtin_amdgcn_s_memrealtime();
    }
    float *stg = reinterpret_cast<float *>(smem);      // the ring is no longer needed (MT*16 rows x 528 B <= its size)
#if RING_DIRECT_PARTIAL
    // A/B build (make ablate A=dp X=-DRING_DIRECT_PARTIAL=1): binary32 partial planes straight from the accumulators -- a lane holds 4
    // consecutive columns of a row, 4 lanes cover 64 B of it -- without the round trip through LDS and its two barriers
    if constexpr (EPI == EPI_PARTIAL && !W8) {
        if (wave_live) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int m = mt * 16 + c;
                if (m >= t.M) continue;
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const int n = n0 + 16 * nt + 4 * q;
                    if (n < t.Np) *reinterpret_cast<f32x4 *>(t.part + ((int64_t)t.kslice * M + m) * t.Np + n) = acc[nt][mt];
                }
            }
        }
        if (bs.timeline && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bs.timeline[4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 3] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
#endif
    __syncthreads();                                   // every compute wave is past its last fragment read
    if (wave_live) stage_acc<MT>(stg, acc, t.M, wave * 32, c, q);
    __syncthreads();
    if constexpr (EPI == EPI_MM8) {
        store_staged_mm8<256>(stg, t.M, n_base, t, bs.q8);
    } else if constexpr (EPI == EPI_PAIR) {
        // the "second" flag goes through the padding columns of the staging rows (a second __shared__ object would
        // de-pipeline the main loop, cdna_hip_programming.md)
        store_staged_sc1<256>(stg, t.M, n_base, t, t.kslice);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *const flag = reinterpret_cast<int *>(stg + kTileRows);
        if (tid == 0) {
            const int drawn = __hip_atomic_fetch_add(bs.counters + t.pair_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (int)gridDim.y - 1;
            if (drawn == last) __hip_atomic_store(bs.counters + t.pair_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            *flag = drawn == last;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the loads below the barrier)
        if (*flag) {
            if constexpr (W8) {                        // uint8 weights: the mm8 epilogue over the sums of all slices
                combine_into_stage<256>(stg, t.M, n_base, t, t.kslice, (int)gridDim.y);
                __syncthreads();                       // (flag is uniform over the workgroup's four compute waves)
                store_staged_mm8<256>(stg, t.M, n_base, t, bs.q8);
            } else {
                combine_store<256>(stg, t.M, n_base, t, t.kslice, (int)gridDim.y);
            }
        }
    } else {
        store_staged<EPI, 256>(stg, t.M, n_base, t, t.kslice, M);
    }
    if (bs.timeline && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bs.timeline[4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------------
// The time-mix launch: R/K/V and the WHOLE LoRA chain of a layer (four down-projections -> tanh / sigmoid -> four up-
// projections + bias) in ONE launch of 128-column tiles with two workgroups per tile over the two halves of the rows.
// Round 2 ran the up-projections as a launch of their own behind the grouped R/K/V + down-projection launch: 12 us per layer
// for 13 MB of traffic, all prologue / epilogue / launch boundary, while that grouped launch left 50 of the 256 CUs idle.
// Here the chain runs on those CUs BESIDE the R/K/V tiles, off the critical path:
//   workgroups [0, n_chain)   (lowest ids: dispatched first)
//       1. one K-slice of one down-projection tile (split `dsplits` ways so that the slice is short); the slice's binary32
//          slab goes out write-through (sc1), one lane draws the tile's ticket, the LAST slice to arrive adds the slabs in
//          slice order (its own from LDS; sc1 loads of the others: the fence-free hand-off of MI355X_MICROARCH.md, one
//          workgroup per CU), applies tanh / sigmoid, stores the binary16 hidden tile write-through and, after every storing
//          wave has drained and the workgroup's barrier, adds to done[half] (one counter for all problems of a half);
//       2. a contiguous share of the up-projection tiles: ONE lane polls done[half] (relaxed, s_sleep, bounded),
//          ONE agent-scope acquire, vmcnt(0), barrier, then plain LDS-DMA loads of the hidden rows (cdna_hip_programming.md
//          Guideline 16, recipe R1), the usual main loop and the bias epilogue;
//       3. a `finished` ticket; the last chain workgroup zeroes the counters for the next launch.
//   workgroups [n_chain, ...)  one R/K/V tile half each, exactly ring_gemm_kernel's EPI_F16 path; they never wait.
// No workgroup waits for a higher-numbered one that could be undispatched behind it: the chain's producers ARE the lowest ids,
// and every spin is bounded (STICKY status word set, outputs then undefined -- chirrup_amd's worker receives the word with every
// step's sampled ids: rwkv7_commit_sampled copies it behind them, the step that gave up is fatal).
struct ChainTable {
    const f16 *dX[4];            // down-projection inputs  [M][ldx]
    const f16 *dW[4];            // down-projection weights [dN[p]][K], row stride ldw (rows >= dN read as zeros)
    f16 *hid[4];                 // hidden planes [M][ld_hid]: problem p's first dN[p] columns are written
    const void *uW[4];           // up-projection weights: tile images of [up_N][up_Kimg], the first uK[p] columns multiplied
    const f16 *ubias[4];
    f16 *uY[4];                  // [M][up_ldy]
    int dN[4], dact[4], dfirst[5], uK[4];
    int n_lora, n_chain, dsplits, ld_hid, up_N, up_Kimg, up_ldy;
    const f16 *m_rx[4], *m_mx[4];   // uint8 main problems (chain_gemm_kernel<.., true>): scales of problem b and the row sums [M][3] of
    const float *m_S[4];            // its activation prologue -- the rank-1 corrections of the mm8 split form run in the tile's epilogue
    int warm;                    // 1: the chain workgroups touch their up-projection weights at the start (L2 warm-up)
    int halves;                  // 2: every tile as two workgroups over the two halves of the rows (M > 32); 1: whole rows
    int rkv_splits;              // halves == 1 only: K-slices of an R/K/V tile, reduced inside the launch by the last to arrive (EPI_PAIR)
    float *slab;                 // [down tile][half][slice][16 MT][128] binary32
    int *sync;                   // kChainTickets tickets, kChainDone done counters, `finished`, kChainPairs R/K/V tile tickets, then the status word
    int *status;                 // the sticky status word: the caller's own, or sync + kChainWords - 1
    int spin_limit;
    unsigned long long *stamps;  // diagnostic (skinny_gemm_clock_probe): 8 x 100-MHz time stamps per workgroup
};
constexpr int kChainTickets = 64, kChainDone = 8, kChainPairs = 512, kChainMaxSplits = 8;
// The status word is the LAST word: everything in front of it may be zeroed between launches (a captured decode graph does so at
// the head of every replay), the status word never is (round-3 advisor finding: it sat between `finished` and the pair tickets,
// inside the range every replay zeroed, so a step that gave up was erased by the next one).
constexpr int kChainWords = kChainTickets + kChainDone + 1 + kChainPairs + 1;

// one K-range of one 128-column tile: prologue, main loop; leaves the sums in `acc` and EVERY wave behind a barrier (the
// ring is free again).  Loader waves come back too (ring_gemm_kernel's leave at this point).  D: ring slots.
// chain_prologue: the loader waves' first D - 1 stages alone -- issued for the NEXT tile while the compute waves still store
// the previous one (the up-projection share keeps its staging area apart from the ring), then chain_mainloop(.., true).
template <int MT, int D, bool W8 = false>
__device__ __forceinline__ void chain_stage(const Loader<W8> &ld, const int kb, const int k_begin, const int n_base, unsigned char *smem) {
    constexpr int kXBytes = MT * 16 * 128, kWBytes = 128 * (W8 ? 64 : 128);
    constexpr int kXRounds = (MT + 1) / 2, kWLoads = kWBytes / 16 / 256;
    const int lw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3;
    const bool short_x = (MT & 1) && lw >= 2;
    const int k0 = k_begin + kb * kKB;
    unsigned char *xb = smem + (kb % D) * kXBytes, *wb = smem + D * kXBytes + (kb % D) * kWBytes;
#pragma unroll
    for (int i = 0; i < kXRounds; i++) {
        if (i == kXRounds - 1 && short_x) break;
        ld.x_round(i, k0, xb + (i * 256 + lw * 64) * 16);
    }
#pragma unroll
    for (int i = 0; i < kWLoads; i++) ld.w_round(n_base, i, k0, wb + (i * 256 + lw * 64) * 16);
}

template <int MT, int D>
__device__ __forceinline__ void chain_prologue(const Tile &t, const int ldx, const int64_t ldw, const int K_img, const int k_begin,
                                               const int nkb, const int n_base, unsigned char *smem) {
    const Loader<false> ld(t, t.M, ldx, ldw, K_img, threadIdx.x & 255);
#pragma unroll
    for (int p = 0; p < D - 1; p++)
        if (p < nkb) chain_stage<MT, D>(ld, p, k_begin, n_base, smem);
}

template <int MT, int D, bool W8 = false>
__device__ __forceinline__ void chain_mainloop(const Tile &t, const int ldx, const int64_t ldw, const int K_img, const int k_begin,
                                               const int nkb, const int n_base, unsigned char *smem, f32x4 (&acc)[2][MT],
                                               const bool prologue_issued = false) {
    constexpr int kXBytes = MT * 16 * 128, kWBytes = 128 * (W8 ? 64 : 128);
    constexpr int kXRounds = (MT + 1) / 2, kWLoads = kWBytes / 16 / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const bool computes = wave < 4;
    const int lt = tid & 255, lw = wave & 3;
    const bool short_x = (MT & 1) && lw >= 2;
    unsigned char *const xring = smem, *const wring = smem + D * kXBytes;
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const Loader<W8> ld(t, t.M, ldx, ldw, K_img, lt);
    if (!computes && !prologue_issued) {
#pragma unroll
        for (int p = 0; p < D - 1; p++)
            if (p < nkb) chain_stage<MT, D, W8>(ld, p, k_begin, n_base, smem);
    }
    if (!computes) {                                   // (one loop per role: see ring_gemm_kernel)
        for (int kb = 0; kb < nkb; kb++) {
            const int left = nkb - 1 - kb, ahead = left < D - 2 ? left : (D - 2 > 0 ? D - 2 : 0);
            if (short_x) wait_stages_ahead<kXRounds - 1 + kWLoads>(ahead);
            else wait_stages_ahead<kXRounds + kWLoads>(ahead);
            asm volatile("s_barrier" ::: "memory");
            if (kb + D - 1 < nkb) chain_stage<MT, D, W8>(ld, kb + D - 1, k_begin, n_base, smem);
        }
    } else {
        for (int kb = 0; kb < nkb; kb++) {
            asm volatile("s_barrier" ::: "memory");
            f16x8 wf[2][2];
            read_w_frags<W8>(wring + (kb % D) * kWBytes, wave * 32, c, q, wf);
            mma_kblock<MT>(xring + (kb % D) * kXBytes, wf, c, q, acc, [] {});
        }
    }
    __syncthreads();                                   // every wave: the last fragment reads are done, no LDS-DMA is pending
}

// ring slots of the up-projection share: what fits beside a staging area of its own (so that the loader waves can run ahead)
template <int MT>
constexpr int chain_up_depth() {
    constexpr int slot = MT * 16 * 128 + 128 * 128, stage = MT * 16 * kLd * 4;
    constexpr int d = (160 * 1024 - stage) / slot;
    return d > 3 ? 3 : (d < 2 ? 2 : d);
}

__device__ __forceinline__ bool chain_wait(int *word, const int want, const int limit, int *status) {
    for (int i = 0; i < limit; i++) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_fetch_or(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sticky: only ever OR-ed by launches
    return false;
}

template <int MT, bool W8M = false>
__global__ __launch_bounds__(512) void chain_gemm_kernel(const int M, const int K, const int ldx, const int64_t ldw,
                                                         const GroupTable gt, const ChainTable ct) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const bool computes = wave < 4;
    float *const stg = reinterpret_cast<float *>(smem);
    int *const flag = reinterpret_cast<int *>(stg + kTileRows);          // a padding column of staged row 0
    f32x4 acc[2][MT];
    const int L = blockIdx.x;
    const int m0 = MT * 16;                                               // rows of the first half

    if (L >= ct.n_chain) {
        // ---- an R/K/V tile: one half of its rows (halves == 2; ring_gemm_kernel's EPI_F16 path), or one of its K-slices (rkv_splits
        //      > 1: EPI_PAIR, the last slice to arrive reduces), or all of it; the tile order is XCD-aware as tile_of_block's
        const int Lr = L - ct.n_chain, total = (int)gridDim.x - ct.n_chain;
        int v = (total & 7) ? Lr : (Lr & 7) * (total >> 3) + (Lr >> 3);
        int half = 0, kslice = 0;
        if (ct.halves == 2) {
            half = v & 1;
            v >>= 1;
        } else if (ct.rkv_splits > 1) {
            kslice = v % ct.rkv_splits;
            v /= ct.rkv_splits;
        }
        if (v >= gt.first[gt.used]) return;
        int b = 0;
        while (b + 1 < gt.used && v >= gt.first[b + 1]) b++;
        Tile t;
        t.ngroup = v - gt.first[b], t.kslice = kslice, t.batch = b, t.pair_id = v;
        t.rows0 = half ? m0 : 0;
        t.M = ct.halves == 2 ? (half ? M - m0 : m0) : M;
        t.X = gt.X[b] + (int64_t)t.rows0 * ldx, t.W = gt.W[b], t.Y = gt.Y[b] + (int64_t)t.rows0 * gt.ldy[b];
        t.bias = gt.bias[b], t.part = gt.part[b], t.Np = gt.N[b], t.ldy = gt.ldy[b], t.act = gt.act[b], t.w_tiled = gt.tiled[b] != 0;
        const int n_base = t.ngroup * 128, k_slice = K / ct.rkv_splits;
        if (ct.stamps && tid == 0) ct.stamps[(int64_t)L * 8] = __builtin_amdgcn_s_memrealtime();
        chain_mainloop<MT, ring_depth<MT, W8M>(), W8M>(t, ldx, ldw, K, kslice * k_slice, k_slice / kKB, n_base, smem, acc);
        if (ct.stamps && tid == 0) ct.stamps[(int64_t)L * 8 + 1] = __builtin_amdgcn_s_memrealtime();
        if (!computes) return;
        if (n_base + wave * 32 < t.Np) stage_acc<MT>(stg, acc, t.M, wave * 32, c, q);
        __syncthreads();                               // (the four compute waves; finished waves do not count)
        if (W8M && ct.rkv_splits == 1) {               // uint8 weights: y = rx*(core - 1023.5*S0) + S1 + mx*S2 (ring_gemm_kernel's EPI_MM8)
            Mm8Epilogue e8{};
            e8.rx = ct.m_rx[b], e8.mx = ct.m_mx[b], e8.S = ct.m_S[b], e8.S_parts = 1;
            store_staged_mm8<256>(stg, t.M, n_base, t, e8);
        } else if (ct.rkv_splits > 1) {                       // ring_gemm_kernel's EPI_PAIR epilogue (same hand-off, same bits as a reduce launch)
            int *const pair = ct.sync + kChainTickets + kChainDone + 1 + t.pair_id;
            store_staged_sc1<256>(stg, t.M, n_base, t, t.kslice);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int drawn = __hip_atomic_fetch_add(pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (drawn == ct.rkv_splits - 1) __hip_atomic_store(pair, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = drawn == ct.rkv_splits - 1;
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (*flag) {
                if (W8M) combine_store_mm8<256>(stg, t.M, n_base, t, t.kslice, ct.rkv_splits, ct.m_rx[b], ct.m_mx[b], ct.m_S[b]);
                else combine_store<256>(stg, t.M, n_base, t, t.kslice, ct.rkv_splits);
            }
        } else {
            store_staged<EPI_F16, 256>(stg, t.M, n_base, t, 0, M);
        }
        if (ct.stamps && tid == 0) ct.stamps[(int64_t)L * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        return;
    }

    // ---- a chain workgroup
    int *const tickets = ct.sync, *const done = ct.sync + kChainTickets, *const finished = ct.sync + kChainTickets + kChainDone;
    int *const status = ct.status;
    const int n_dtiles = ct.dfirst[ct.n_lora];
    unsigned long long *const stamps = ct.stamps ? ct.stamps + (int64_t)L * 8 : nullptr;
    auto stamp = [&](int i) {
        if (stamps && tid == 0) stamps[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    // this workgroup's share of the up-projection tile halves: half-major, then column tile, problem fastest -- a contiguous share
    // lies in one half (two at most) and mixes the problems' lengths
    const int H = ct.halves;
    const int ctiles = (ct.up_N + 127) / 128, U = ct.n_lora * ctiles * H, per_half = ct.n_lora * ctiles;
    const int lo = (int)((int64_t)L * U / ct.n_chain), hi = (int)((int64_t)(L + 1) * U / ct.n_chain);
    // Warm THIS XCD's L2 with the up-projection weights this workgroup will stream later (a few hundred KB, contiguous runs of
    // a tile image): issued by the compute waves now, while they wait for the first stage anyway; nobody waits for the data.
    uint32_t warm = 0;
    if (computes && ct.warm) {
        for (int it = lo; it < hi; it++) {
            const int r = it % per_half, p = r % ct.n_lora, ctile = r / ct.n_lora;
            const unsigned char *w0 = static_cast<const unsigned char *>(ct.uW[p]) + (int64_t)ctile * (ct.up_Kimg / kKB) * (kTileRows * kKB * 2);
            const int lines = (ct.uK[p] / kKB) * (kTileRows * kKB * 2) / 128;
            for (int ln = tid; ln < lines; ln += 256) warm ^= *reinterpret_cast<const uint32_t *>(w0 + (int64_t)ln * 128);     // default policy: stays in L2
        }
    }
    bool ok = true;
    if (L < n_dtiles * H * ct.dsplits) {
        // 1. one K-slice of one down-projection tile (half)
        const int slice = L % ct.dsplits, th = L / ct.dsplits, half = th % H, dtile = th / H;
        int p = 0;
        while (p + 1 < ct.n_lora && dtile >= ct.dfirst[p + 1]) p++;
        Tile t;
        t.ngroup = dtile - ct.dfirst[p], t.kslice = slice, t.batch = p, t.pair_id = dtile * H + half;
        t.rows0 = half ? m0 : 0;
        t.M = H == 2 ? (half ? M - m0 : m0) : M;
        t.X = ct.dX[p] + (int64_t)t.rows0 * ldx, t.W = ct.dW[p], t.Y = ct.hid[p] + (int64_t)t.rows0 * ct.ld_hid;
        t.bias = nullptr, t.part = nullptr, t.Np = ct.dN[p], t.ldy = ct.ld_hid, t.act = ct.dact[p], t.w_tiled = false;
        const int n_base = t.ngroup * 128, k_slice = K / ct.dsplits;
        chain_mainloop<MT, ring_depth<MT, false>()>(t, ldx, ldw, K, slice * k_slice, k_slice / kKB, n_base, smem, acc);
        stamp(1);
        if (computes && n_base + wave * 32 < t.Np) stage_acc<MT>(stg, acc, t.M, wave * 32, c, q);
        __syncthreads();
        float *const slabs = ct.slab + (int64_t)t.pair_id * ct.dsplits * (m0 * 128);     // [slice][m0][128]
        const int c4 = tid & 31;
        const bool col_live = n_base + 4 * c4 < t.Np;
        if (computes && col_live) {                    // this slice's slab, write-through
            const __amdgpu_buffer_rsrc_t dst = make_rsrc(slabs + (int64_t)slice * (m0 * 128), (int64_t)m0 * 128 * 4);
            for (int m = tid >> 5; m < t.M; m += 8) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(stg + m * kLd + 4 * c4);
                __builtin_amdgcn_raw_buffer_store_b128(v, dst, (m * 128 + 4 * c4) * 4, 0, 16);      // aux 16 = sc1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {                                // arrive, then wait for the other K-slices of this tile half (bounded)
            const int drawn = __hip_atomic_fetch_add(tickets + t.pair_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = drawn == ct.dsplits - 1 || chain_wait(tickets + t.pair_id, ct.dsplits, ct.spin_limit, status);     // (the last arriver need not look again)
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the loads below the barrier)
        stamp(2);
        ok = *flag != 0;
        if (ok) {
            // EVERY slice combines its share of the rows (rows slice * rq ...), all eight waves, one row x 8 columns per lane:
            // hidden = act(binary16(sum of the slices in slice order)) -- its own slice from LDS, the others by sc1 loads, all of
            // them in flight before the first add (a lone last arriver took 15 us for the same 172 KB: per-CU load latency under
            // the R/K/V tiles' streaming is ~6 us a round) -- stored write-through, 16 B per lane
            const int c8 = tid & 15, n = n_base + 8 * c8;
            const int rq = (t.M + ct.dsplits - 1) / ct.dsplits, r_end = (slice + 1) * rq < t.M ? (slice + 1) * rq : t.M;
            if (n < t.Np) {
                const __amdgpu_buffer_rsrc_t src = make_rsrc(slabs, (int64_t)ct.dsplits * m0 * 128 * 4);
                const __amdgpu_buffer_rsrc_t out = make_rsrc(t.Y, ((int64_t)(t.M - 1) * ct.ld_hid + t.Np) * 2);
                for (int m = slice * rq + (tid >> 4); m < r_end; m += 32) {
                    f32x4 oth[kChainMaxSplits][2];
#pragma unroll
                    for (int k = 0; k < kChainMaxSplits; k++)
                        if (k < ct.dsplits && k != slice) {
#pragma unroll
                            for (int h = 0; h < 2; h++)
                                oth[k][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, ((k * m0 + m) * 128 + 8 * c8 + 4 * h) * 4, 0, 16));
                        }
                    f32x4 sum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int k = 0; k < kChainMaxSplits; k++)
                        if (k < ct.dsplits) {
#pragma unroll
                            for (int h = 0; h < 2; h++)
                                sum[h] += (k == slice) ? *reinterpret_cast<const f32x4 *>(stg + m * kLd + 8 * c8 + 4 * h) : oth[k][h];
                        }
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = (f16)apply_act(e < 4 ? sum[0][e] : sum[1][e - 4], t.act);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out, (m * ct.ld_hid + n) * 2, 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains ...
            __syncthreads();                                        // ... before ONE lane signals for all of them
            if (tid == 0) __hip_atomic_fetch_add(done + half, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // one counter per half, all problems
        }
        __syncthreads();                                            // the staging area is free again
        stamp(3);
    }
    // 2. the up-projection share; ONE wait on all of a half's counters and ONE acquire when the share first touches that half.
    //    The ring is shallower here (tiles of 2..8 K-blocks) and the staging area lies behind it, so the loader waves issue the
    //    NEXT tile's first stages while the compute waves still store this one.
    {
        constexpr int DU = chain_up_depth<MT>();
        float *const stgu = reinterpret_cast<float *>(smem + DU * (MT * 16 * 128 + 128 * 128));
        auto up_tile = [&](int it, Tile &t, int &nkb, int &n_base) {
            const int half = it / per_half, r = it % per_half, p = r % ct.n_lora, ctile = r / ct.n_lora;
            t.ngroup = ctile, t.kslice = 0, t.batch = p, t.pair_id = 0;
            t.rows0 = half ? m0 : 0;
            t.M = H == 2 ? (half ? M - m0 : m0) : M;
            t.X = ct.hid[p] + (int64_t)t.rows0 * ct.ld_hid, t.W = ct.uW[p], t.Y = ct.uY[p] + (int64_t)t.rows0 * ct.up_ldy;
            t.bias = ct.ubias[p], t.part = nullptr, t.Np = ct.up_N, t.ldy = ct.up_ldy, t.act = 0, t.w_tiled = true;
            nkb = ct.uK[p] / kKB, n_base = ctile * 128;
        };
        int have = 0;                                               // halves already acquired (bit per half)
        bool pre = false;                                           // the ring already holds the first stages of tile `it`
        for (int it = lo; it < hi && ok; it++) {
            const int half = it / per_half;
            if (!((have >> half) & 1)) {                            // (never with pre: a tile is only run ahead inside an acquired half)
                if (tid == 0) {
                    const bool got = chain_wait(done + half, n_dtiles * ct.dsplits, ct.spin_limit, status);     // (one round trip, not one per problem)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    *flag = got;
                }
                __syncthreads();
                ok = *flag != 0;
                __syncthreads();                                    // (flag lies in the ring: rewritten by the stages below)
                if (!ok) break;
                have |= 1 << half;
                stamp(4);
            }
            Tile t;
            int nkb, n_base;
            up_tile(it, t, nkb, n_base);
            chain_mainloop<MT, DU>(t, ct.ld_hid, ct.up_Kimg, ct.up_Kimg, 0, nkb, n_base, smem, acc, pre);
            pre = false;
            if (it + 1 < hi && ((have >> ((it + 1) / per_half)) & 1)) {
                pre = true;
                if (!computes) {
                    Tile tn;
                    int nkb_n, n_base_n;
                    up_tile(it + 1, tn, nkb_n, n_base_n);
                    chain_prologue<MT, DU>(tn, ct.ld_hid, ct.up_Kimg, ct.up_Kimg, 0, nkb_n, n_base_n, smem);
                }
            }
            if (computes && n_base + wave * 32 < t.Np) stage_acc<MT>(stgu, acc, t.M, wave * 32, c, q);
            __syncthreads();
            if (computes) store_staged<EPI_F16, 256>(stgu, t.M, n_base, t, 0, M);
            // (no barrier here: the next tile's main loop has at least two before anybody writes the staging area again)
            if (it == lo) stamp(5);
        }
        stamp(6);
    }
    if (warm == 0x9e3779b9u && (uint32_t)tid == 511u + warm) __hip_atomic_fetch_or(status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (never: keeps the warm-up loads)
    // 3. the last chain workgroup to finish leaves every counter at zero for the next launch
    stamp(7);
    if (tid == 0) {
        const int f = __hip_atomic_fetch_add(finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f == ct.n_chain - 1) {
            for (int i = 0; i < kChainTickets; i++) __hip_atomic_store(tickets + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < kChainDone; i++) __hip_atomic_store(done + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(finished, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BN = 256: eight compute waves (two per SIMD).  Waves 0-3 issue the x loads (next K-block, 2 slots) at the start of an
// iteration, waves 4-7 the W loads (WD-1 K-blocks ahead, WD slots) between the two k-steps of their MFMAs.
template <int MT, bool W8, int EPI>
__global__ __launch_bounds__(512) void wide_gemm_kernel(
    const int M, const int N, const int K, const int k_slice, const f16 *__restrict__ X, const int ldx,
    const void *__restrict__ Wv, const int64_t ldw, f16 *__restrict__ Y, const int ldy,
    const f16 *__restrict__ bias, float *__restrict__ part, const BatchStrides bs, const GroupTable gt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BN = 256, XD = 2, WD = W8 ? 5 : 3;
    constexpr int kXBytes = MT * 16 * 128;
    constexpr int kWTile = Loader<W8>::kWTile;         // one 128-row tile image: 16 KiB (uint8: 8 KiB)
    constexpr int kWBytes = 2 * kWTile;
    constexpr int kXRounds = (MT + 1) / 2;             // per x-loader wave: rounds of 4 waves x 1 KiB (32 rows); half a round last when MT is odd
    constexpr int kWLoads = kWBytes / 16 / 256;        // per W-loader wave: 8 (uint8: 4)
    constexpr int kPerTile = kWLoads / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const bool xrole = wave < 4;
    const int lt = tid & 255, lw = wave & 3;
    const bool short_x = (MT & 1) && lw >= 2;
    const Tile t = resolve_tile<W8, EPI, MT>(N, X, Wv, Y, ldy, bias, part, M, ldx, bs, gt);
    const int n_base = t.ngroup * BN;
    if (n_base >= t.Np) return;                        // padding workgroup (launch_gemm rounds the grid up for the XCD map)
    const int n0 = n_base + wave * 32;
    const bool wave_live = n0 < t.Np;
    const int k_begin = t.kslice * k_slice;
    const int Kz = (t.batch < 8 && bs.k[t.batch] > 0) ? bs.k[t.batch] : K;
    const int k_end = (k_begin + k_slice) < Kz ? (k_begin + k_slice) : Kz;
    const int nkb = k_end > k_begin ? (k_end - k_begin) / kKB : 0;
    unsigned char *const xring = smem, *const wring = smem + XD * kXBytes;

    const Loader<W8> ld(t, t.M, ldx, ldw, K, lt);

    f32x4 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage_x = [&](int kb) {
        const int k0 = k_begin + kb * kKB;
        unsigned char *xb = xring + (kb % XD) * kXBytes;
#pragma unroll
        for (int i = 0; i < kXRounds; i++) {
            if (i == kXRounds - 1 && short_x) break;
            ld.x_round(i, k0, xb + (i * 256 + lw * 64) * 16);
        }
    };
    auto stage_w = [&](int kb) {
        const int k0 = k_begin + kb * kKB;
        unsigned char *wb = wring + (kb % WD) * kWBytes;
#pragma unroll
        for (int i = 0; i < kWLoads; i++)
            ld.w_round(n_base + (i / kPerTile) * kTileRows, i % kPerTile, k0, wb + (i * 256 + lw * 64) * 16);
    };

    // at step kb: the x ring holds stage kb, the W ring stages kb .. kb+WD-2, each in issue order of its loader waves
    if (xrole) {
        if (nkb > 0) stage_x(0);
    } else {
#pragma unroll
        for (int p = 0; p < WD - 1; p++)
            if (p < nkb) stage_w(p);
    }
    for (int kb = 0; kb < nkb; kb++) {
        const int left = nkb - 1 - kb;
        if (xrole) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else wait_stages_ahead<kWLoads>(left < WD - 2 ? left : WD - 2);
        asm volatile("s_barrier" ::: "memory");        // stage kb of both operands landed; the slots restaged below are no longer read
        if (xrole && kb + 1 < nkb) stage_x(kb + 1);
        __builtin_amdgcn_sched_barrier(0);
        f16x8 wf[2][2];
        read_w_frags<W8>(wring + (kb % WD) * kWBytes + (wave >> 2) * kWTile, (wave & 3) * 32, c, q, wf);
        // W loaders: between the k-steps of their MFMAs, when the x loaders are back to computing
        mma_kblock<MT>(xring + (kb % XD) * kXBytes, wf, c, q, acc, [&] { if (!xrole && kb + WD - 1 < nkb) stage_w(kb + WD - 1); });
    }
    // ---- epilogue: two passes of 128 columns through a row-major LDS staging area (the rings are no longer needed)
    float *stg = reinterpret_cast<float *>(smem);
    __syncthreads();                                   // every wave is past its last fragment read (no LDS-DMA is pending)
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if ((wave >> 2) == half && wave_live) stage_acc<MT>(stg, acc, t.M, (wave & 3) * 32, c, q);
        __syncthreads();
        store_staged<EPI, 512>(stg, t.M, n_base + half * kTileRows, t, t.kslice, M);
        if (half == 0) __syncthreads();
    }
}

// Sum the split-K partials and apply the epilogue.
//   mode 0: y = sum (+ bias[n]);  mode 1: y = relu(sum (+bias))^2;
//   mode 2 (mm8): y = rx[n]*(sum - 1024*S[m][0] + 0.5*S[m][0]) + S[m][1] + mx[n]*S[m][2]      (benchmark.py:167-179; the
//                 u8 kernels multiply by 1024 + q, see cvt_u8x2)
//   mode 3: mm8 then relu^2
//   mode 4 + p: the RWKV-7 LoRA hidden planes (v, w, a, g), first problem = plane p: tanh on w, sigmoid on g
//               (rwkv7.py:626, :630), applied to the binary16-rounded sum like the reference's separate op
// blockIdx.y = problem of a batched launch (partials [Z][splits][M][N], Y / bias advance by y_bs / bias_bs).
// G: groups of 4 columns per lane -- 2 (one 16-B store per lane; 8-B stores run at 0.54-0.70x the rate) when every N, row
// stride and pointer of the launch allows it (launch_reduce).
template <int G>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const int M, const int N, const int splits,
                                                            const float *__restrict__ part, const f16 *__restrict__ bias,
                                                            const f16 *__restrict__ rx, const f16 *__restrict__ mx,
                                                            const float *__restrict__ S, const int mode,
                                                            f16 *__restrict__ Y, int ldy, const int64_t y_bs = 0,
                                                            const int64_t bias_bs = 0, const GroupTable gt = GroupTable{}) {
    constexpr int COLS = 4 * G;
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
    const int64_t total = (int64_t)M * N_ / COLS;
    if (gi >= total) return;
    if (!gt.used) {
        part += (int64_t)z * splits * M * N;
        Y += z * y_bs;
        if (bias) bias += z * bias_bs;
    }
    const int m = (int)(gi / (N_ / COLS)), n = (int)(gi % (N_ / COLS)) * COLS;
    f32x4 s[G];
#pragma unroll
    for (int g = 0; g < G; g++) s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < splits; k0 += 4) {           // four planes' loads in flight, added in plane order
        f32x4 pv[4][G];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int g = 0; g < G; g++)
                if (k0 + u < splits)
                    pv[u][g] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(part + ((int64_t)(k0 + u) * M + m) * N_ + n + 4 * g));
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int g = 0; g < G; g++)
                if (k0 + u < splits) s[g] += pv[u][g];
    }
    f16 o[COLS];
#pragma unroll
    for (int g = 0; g < G; g++) {
        f16x4 rxv = {}, mxv = {}, bv = {};
        if (mm8) rxv = *reinterpret_cast<const f16x4 *>(rx + n + 4 * g), mxv = *reinterpret_cast<const f16x4 *>(mx + n + 4 * g);
        else if (bias) bv = *reinterpret_cast<const f16x4 *>(bias + n + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float v = s[g][e];
            if (mm8) v = (float)rxv[e] * (v - (kU8Offset - 0.5f) * S[m * 3 + 0]) + S[m * 3 + 1] + (float)mxv[e] * S[m * 3 + 2];
            else if (bias) v += (float)bv[e];
            o[4 * g + e] = (f16)apply_act(v, act);
        }
    }
    if constexpr (G == 2) {
        *reinterpret_cast<f16x8 *>(Y + (int64_t)m * ldy + n) = (f16x8){o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
    } else {
        *reinterpret_cast<f16x4 *>(Y + (int64_t)m * ldy + n) = (f16x4){o[0], o[1], o[2], o[3]};
    }
}

// wide: every problem has N % 8 == 0, ldy % 8 == 0 and a 16-byte aligned Y (callers check)
int launch_reduce(bool wide, unsigned zdim, hipStream_t st, int M, int N, int s, const float *part, const f16 *bias, const f16 *rx,
                  const f16 *mx, const float *S, int mode, f16 *Y, int ldy, int64_t y_bs = 0, int64_t bias_bs = 0,
                  const GroupTable &gt = GroupTable{}) {
    const int64_t total = (int64_t)M * N / (wide ? 8 : 4);
    const dim3 grid((unsigned)((total + 255) / 256), zdim);
    if (wide) hipLaunchKernelGGL(skinny_reduce_kernel<2>, grid, dim3(256), 0, st, M, N, s, part, bias, rx, mx, S, mode, Y, ldy, y_bs, bias_bs, gt);
    else hipLaunchKernelGGL(skinny_reduce_kernel<1>, grid, dim3(256), 0, st, M, N, s, part, bias, rx, mx, S, mode, Y, ldy, y_bs, bias_bs, gt);
    return (int)hipGetLastError();
}
inline bool wide_ok(int N, int ldy, const void *Y, int64_t y_bs = 0) {
    return !(N & 7) && !(ldy & 7) && !(reinterpret_cast<uintptr_t>(Y) & 15) && !(y_bs & 7);
}

// mm8 activation prologue: xs = fp16(x * ry), S[m] = {sum xs, sum x*my, sum x}   (benchmark.py:167-173)
// xs_lo != NULL (mm8t_seq_exact): the product x*ry of two binary16 numbers has at most 22 significant bits, so it is EXACT in
// binary32 and splits exactly into two binary16 numbers, hi = fp16(p) and lo = fp16(p - hi) (lo below 2^-14 loses its last bits
// to the subnormal grid: <= 2^-25 absolute); S0 then sums hi + lo.  Two matrix-core passes (hi, lo) with binary32 accumulation
// reproduce the as-coded expression's arithmetic -- products exact, sums in binary32 -- instead of rounding xs to binary16.
__global__ __launch_bounds__(256) void mm8_prep_kernel(const int K, const f16 *__restrict__ x, const int ldx,
                                                       const f16 *__restrict__ ry, const f16 *__restrict__ my,
                                                       f16 *__restrict__ xs, float *__restrict__ S, f16 *__restrict__ xs_lo = nullptr) {
    __shared__ float red[3][4];
    const int m = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x * 8; c < K; c += 256 * 8) {
        const f16x8 xv = *reinterpret_cast<const f16x8 *>(x + (int64_t)m * ldx + c);
        const f16x8 rv = *reinterpret_cast<const f16x8 *>(ry + c);
        const f16x8 mv = *reinterpret_cast<const f16x8 *>(my + c);
        f16x8 o, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float p = (float)xv[e] * (float)rv[e];
            o[e] = (f16)p;
            lo[e] = (f16)(p - (float)o[e]);
            s0 += xs_lo ? (float)o[e] + (float)lo[e] : (float)o[e];
            s1 += (float)xv[e] * (float)mv[e];
            s2 += (float)xv[e];
        }
        *reinterpret_cast<f16x8 *>(xs + (int64_t)m * K + c) = o;
        if (xs_lo) *reinterpret_cast<f16x8 *>(xs_lo + (int64_t)m * K + c) = lo;
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

// mm8 between two GEMMs: y = rx*(sum of core partials - 1023.5*S0) + S1 + mx*S2 (the rank-1 corrections of benchmark.py:
// 174-179 for kernels that multiply by 1024 + q), optionally relu(fp16(y))^2 (rwkv7.py:678), written to Y if given -- and, if
// the consumer is another mm8 product, its activation prologue on the fly (xs2 = fp16(y*ry2); S2 = {sum xs2, sum y*my2,
// sum y}, mm8_prep_kernel's arithmetic), so that neither the reduce nor the next prologue is its own launch.
// Grid (row, part): a 256-lane workgroup covers 1024 columns of one row, one 16-B load per split and lane; the row sums
// come out as `parts` = ceil(N / 1024) partial sums per row, S2[row][part][3], which the consumers add up in part order
// (rwkv7_add_ln_mix_mm8: in_S_parts; this kernel: S_parts) -- deterministic, no atomics.
constexpr int kRowPart = 1024;
__global__ __launch_bounds__(256) void mm8_reduce_rows_kernel(const int N, const int splits, const int64_t split_stride,
                                                               const float *__restrict__ part, const f16 *__restrict__ rx,
                                                               const f16 *__restrict__ mx, const float *__restrict__ S, const int S_parts,
                                                               const int act, f16 *__restrict__ Y, const int ldy,
                                                               const f16 *__restrict__ ry2, const f16 *__restrict__ my2,
                                                               f16 *__restrict__ xs2, float *__restrict__ S2) {
    __shared__ float red[3][4], ssum[3];
    const int m = blockIdx.x, n = blockIdx.y * kRowPart + threadIdx.x * 4;
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int p = 0; p < S_parts; p++) t += S[(m * S_parts + p) * 3 + threadIdx.x];
        ssum[threadIdx.x] = t;
    }
    __syncthreads();
    const float s0 = ssum[0], s1 = ssum[1], s2 = ssum[2];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    if (n < N) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < splits; k++)
            v += __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(part + k * split_stride + (int64_t)m * N + n));
        const f16x4 rxv = *reinterpret_cast<const f16x4 *>(rx + n), mxv = *reinterpret_cast<const f16x4 *>(mx + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float y = (float)rxv[e] * (v[e] - (kU8Offset - 0.5f) * s0) + s1 + (float)mxv[e] * s2;
            if (act) {
                y = (float)(f16)y;
                y = y > 0.f ? y * y : 0.f;
            }
            o[e] = (f16)y;
        }
        if (Y) *reinterpret_cast<f16x4 *>(Y + (int64_t)m * ldy + n) = o;
        if (xs2) {
            const f16x4 ryv = *reinterpret_cast<const f16x4 *>(ry2 + n), myv = *reinterpret_cast<const f16x4 *>(my2 + n);
            f16x4 xs;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                xs[e] = (f16)((float)o[e] * (float)ryv[e]);
                t0 += (float)xs[e];
                t1 += (float)o[e] * (float)myv[e];
                t2 += (float)o[e];
            }
            *reinterpret_cast<f16x4 *>(xs2 + (int64_t)m * N + n) = xs;
        }
    }
    if (!xs2) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        t0 += __shfl_xor(t0, o, 64);
        t1 += __shfl_xor(t1, o, 64);
        t2 += __shfl_xor(t2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = t0, red[1][threadIdx.x >> 6] = t1, red[2][threadIdx.x >> 6] = t2;
    __syncthreads();
    if (threadIdx.x < 3)
        S2[(m * gridDim.y + blockIdx.y) * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// Output columns per workgroup.  256 wherever there are enough columns to fill the chip with 256-wide tiles at a sane
// split (every big matrix of the model); 128 for the narrow problems.  CHIRRUP_GEMM_BN=128|256 forces one kernel (A/B).
int choose_bn(int n_max) {
    static const int forced = [] {
        const char *e = getenv("CHIRRUP_GEMM_BN");
        const int v = e ? atoi(e) : 0;
        return (v == 128 || v == 256) ? v : 0;
    }();
    if (forced) return forced;
    return n_max >= 32768 ? 256 : 128;
}

// K-split factor: enough workgroups for the 256 CUs, slices of whole, equal K-blocks, at least four per slice.
int pick_splits(int bn, int N, int K, int requested, int Z = 1) {
    if (requested > 0) {                               // a request is honoured as far as K allows: whole, equal K-blocks
        int s = requested < K / kKB ? requested : K / kKB;
        while (s > 1 && (K / kKB) % s) s--;
        return s < 1 ? 1 : s;
    }
    const int ngroups = Z * ((N + bn - 1) / bn);
    int s = (256 + ngroups / 2) / ngroups;             // aim at ~256 workgroups (one per CU)
    const int max_s = K / 256 > 0 ? K / 256 : 1;       // keep >= 4 K-blocks per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    while (s > 1 && (K / kKB) % s) s--;                // slices of whole K-blocks, equal size
    return s;
}

size_t lds_bytes(int bn, int MT, bool w8) {
    const size_t x = (size_t)MT * 16 * 128, stage = (size_t)MT * 16 * kLd * sizeof(float);
    size_t ring;
    if (bn == 256) ring = 2 * x + (size_t)(w8 ? 5 : 3) * 2 * kTileRows * (w8 ? 64 : 128);
    else {
        const size_t st = x + (size_t)kTileRows * (w8 ? 64 : 128);
        size_t d = (160 * 1024) / st;                  // ring_depth<MT, W8>()
        ring = (d > 6 ? 6 : d) * st;
    }
    return ring > stage ? ring : stage;
}

}  // namespace

// Diagnostic: while set, every 128-column launch writes per workgroup the duration of its main loop in shader-clock ticks and in
// ticks of the constant 100-MHz counter (their ratio = the clock the CU ran at) to buf[2 * workgroup + {0, 1}]; buf must
// hold 2 * pairs values for the largest grid launched meanwhile.  Process-wide, not thread-safe: bench.py / tools only.
static unsigned long long *g_clock_probe = nullptr;
static int g_clock_pairs = 0;
extern "C" int skinny_gemm_clock_probe(void *buf, int pairs) {
    g_clock_probe = static_cast<unsigned long long *>(buf), g_clock_pairs = buf ? pairs : 0;
    return 0;
}

// Experiment (tools/exp_warm.py): while set, every 128-column launch is preceded by a launch of the same
// grid whose workgroups touch the first `stages` K-blocks of the W tile the GEMM workgroup with the same id will stream -- what a
// preceding kernel's idle lanes could do for the GEMM that follows it (same block id -> same XCD -> the lines wait in that XCD's
// L2).  stages + 100 * mode: mode 1 = the NEXT tile's bytes, 3 = a tile half the matrix away (right memory-side cache, wrong L2),
// 2 = no loads, `stages` x ~4 us of idling (the control that separates "warm" from "rested").
static int g_warm_stages = 0;
static unsigned *g_warm_sink = nullptr;
extern "C" int skinny_gemm_warm_probe(int stages, void *sink) {
    g_warm_stages = stages, g_warm_sink = static_cast<unsigned *>(sink);
    return 0;
}
template <bool W8, int EPI>
__global__ __launch_bounds__(256) void warm_gemm_kernel(const int N, const int K, const int k_slice, const f16 *X, const void *Wv, const int64_t ldw,
                                                        const BatchStrides bs, const GroupTable gt, int stages, unsigned *sink) {
    Tile t = resolve_tile<W8, EPI, 1>(N, X, Wv, nullptr, 0, nullptr, nullptr, 32, 0, bs, gt);
    if (t.ngroup * kTileRows >= t.Np) return;
    const int mode = stages / 100;
    stages %= 100;
    if (mode == 1) t.ngroup = (t.ngroup + 1) * kTileRows < t.Np ? t.ngroup + 1 : 0;
    if (mode == 3) {
        const int nt = (t.Np + kTileRows - 1) / kTileRows;
        t.ngroup = (t.ngroup + nt / 2 + 1) % nt;
    }
    if (mode == 2) {
        for (int i = 0; i < stages; i++) __builtin_amdgcn_s_sleep(127);     // ~4 us each
        return;
    }
    constexpr int kEl = W8 ? 1 : 2, kWTile = kTileRows * kKB * kEl;
    const int k0 = t.kslice * k_slice;
    const unsigned char *w = static_cast<const unsigned char *>(t.W);
    unsigned acc = 0;
    if (t.w_tiled) {
        const unsigned char *base = w + ((int64_t)t.ngroup * (K / kKB) + k0 / kKB) * kWTile;
        for (int ln = threadIdx.x; ln < stages * kWTile / 128; ln += 256) acc ^= *reinterpret_cast<const unsigned *>(base + (int64_t)ln * 128);
    } else {
        const int per_row = stages * kKB * kEl / 128;
        for (int ln = threadIdx.x; ln < kTileRows * per_row; ln += 256) {
            const int row = ln / per_row, c = ln % per_row;
            acc ^= *reinterpret_cast<const unsigned *>(w + ((int64_t)(t.ngroup * kTileRows + row) * ldw + k0) * kEl + c * 128);
        }
    }
    if (acc == 0x9e3779b9u && sink) *sink = acc;       // (keeps the loads)
}

template <bool W8, int EPI>
int launch_gemm(int bn, int MT, dim3 grid, hipStream_t st, int M, int N, int K, int k_slice, const f16 *X, int ldx, const void *W,
                int64_t ldw, f16 *Y, int ldy, const f16 *bias, float *part, BatchStrides bs = BatchStrides{},
                const GroupTable &gt = GroupTable{}) {
    size_t lds = lds_bytes(bn, MT, W8);
    if (lds < (size_t)bs.min_lds) lds = (size_t)bs.min_lds;
    // tile_of_block deals contiguous runs of tiles to the XCDs only when the workgroup count divides by 8: round the
    // N-group count up (the extra workgroups leave at once)
    while ((grid.x * grid.y * grid.z) & 7) grid.x++;
    if (g_warm_stages > 0 && bn == 128) {
        int stages = k_slice / kKB < g_warm_stages % 100 ? k_slice / kKB : g_warm_stages % 100;
        stages += g_warm_stages / 100 * 100;
        hipLaunchKernelGGL((warm_gemm_kernel<W8, EPI>), grid, dim3(256), 0, st, N, K, k_slice, X, W, ldw, bs, gt, stages, g_warm_sink);
    }
    bs.clock = (bn == 128 && g_clock_probe && (int)(grid.x * grid.y * grid.z) <= g_clock_pairs) ? g_clock_probe : nullptr;
    bs.timeline = (bs.clock && (int)(grid.x * grid.y * grid.z) * 3 <= g_clock_pairs) ? g_clock_probe + 2 * (grid.x * grid.y * grid.z) : nullptr;
#define GO_K(KERN, MTV)                                                                                                   \
    do {                                                                                                                  \
        auto kern = KERN<MTV, W8, EPI>;                                                                                   \
        static std::atomic<bool> lds_limit_raised[32];   /* per instantiation AND device (one engine process may drive */ \
        int dev_ = 0;                                    /* several GPUs); the call itself is idempotent */               \
        (void)hipGetDevice(&dev_);                                                                                        \
        if (!lds_limit_raised[dev_ & 31].load(std::memory_order_acquire)) {                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      160 * 1024);                                                                        \
            lds_limit_raised[dev_ & 31].store(true, std::memory_order_release);                                           \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, M, N, K, k_slice, X, ldx, W, ldw, Y, ldy, bias, part, bs, gt); \
    } while (0)
#define GO(MTV)                                                   \
    do {                                                          \
        if constexpr (EPI == EPI_MM8 || EPI == EPI_PAIR) {        \
            if (bn == 256) return CHIRRUP_E_UNSUPPORTED;          \
            GO_K(ring_gemm_kernel, MTV);                          \
        } else {                                                  \
            if (bn == 256) GO_K(wide_gemm_kernel, MTV);           \
            else GO_K(ring_gemm_kernel, MTV);                     \
        }                                                         \
    } while (0)
    switch (MT) {                                      // 16-row tiles of x: M <= 256
        case 1: GO(1); break;
        case 2: GO(2); break;
        case 3: GO(3); break;
        case 4: GO(4); break;
        case 5: GO(5); break;
        case 6: GO(6); break;
        case 7: GO(7); break;
        case 8: GO(8); break;
        case 9: GO(9); break;
        case 10: GO(10); break;
        case 11: GO(11); break;
        case 12: GO(12); break;
        case 13: GO(13); break;
        case 14: GO(14); break;
        case 15: GO(15); break;
        default: GO(16); break;
    }
#undef GO
#undef GO_K
    return (int)hipGetLastError();
}

namespace {
// W [N][K] row-major -> tile images: tile (N-group g, K-block b) = 1024 chunks of 16 B in the order the kernels
// keep them in LDS (row nr = c >> 3 at chunk position c & 7 holds logical chunk (c & 7) ^ ((nr >> 1) & 7)).
__global__ __launch_bounds__(256) void tile_weight_kernel(const int N, const int K, const f16 *__restrict__ W, const int64_t ldw,
                                                          f16 *__restrict__ Wt) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * K / 8;
    if (c >= total) return;
    const int g = (int)(c & 1023);
    const int64_t tile = c >> 10;
    const int kb = (int)(tile % (K / kKB)), ng = (int)(tile / (K / kKB));
    const int nr = g >> 3, lc = (g & 7) ^ ((nr >> 1) & 7);
    *reinterpret_cast<f16x8 *>(Wt + c * 8) = *reinterpret_cast<const f16x8 *>(W + ((int64_t)ng * kTileRows + nr) * ldw + kb * kKB + lc * 8);
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
    *reinterpret_cast<u32x4 *>(Wt + c * 16) = *reinterpret_cast<const u32x4 *>(W + ((int64_t)ng * kTileRows + nr) * ldw + kb * kKB + lc * 16);
}

// tile images -> W [N][K] row-major (the inverse of tile_weight_kernel; same chunk walk, source and destination swapped)
__global__ __launch_bounds__(256) void untile_weight_kernel(const int N, const int K, const f16 *__restrict__ Wt, f16 *__restrict__ W,
                                                            const int64_t ldw) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * K / 8;
    if (c >= total) return;
    const int g = (int)(c & 1023);
    const int64_t tile = c >> 10;
    const int kb = (int)(tile % (K / kKB)), ng = (int)(tile / (K / kKB));
    const int nr = g >> 3, lc = (g & 7) ^ ((nr >> 1) & 7);
    *reinterpret_cast<f16x8 *>(W + ((int64_t)ng * kTileRows + nr) * ldw + kb * kKB + lc * 8) = *reinterpret_cast<const f16x8 *>(Wt + c * 8);
}

inline bool mis16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
}  // namespace

// Tile images back to a row-major matrix W [N][K] (row stride ldw): for a model that keeps ONLY the tile images of its big
// matrices (chirrup_amd.rwkv7: keep_row_major=False) and needs a row-major operand for a library GEMM (prefill chunks > 256 rows).
extern "C" int skinny_untile_weight(int N, int K, const void *Wt, void *W, int64_t ldw, void *stream) {
    if (N <= 0 || K <= 0 || (N % kTileRows) || (K % kKB) || ldw < K || (ldw & 7)) return CHIRRUP_E_SHAPE;
    if (!W || !Wt) return CHIRRUP_E_NULL;
    if (mis16(W) || mis16(Wt)) return CHIRRUP_E_ALIGN;
    const int64_t total = (int64_t)N * K / 8;
    hipLaunchKernelGGL(untile_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, K, static_cast<const f16 *>(Wt), static_cast<f16 *>(W), ldw);
    return (int)hipGetLastError();
}

// The same for the uint8 (mm8) weights wT [M_out][N_in] of mm8t_seq (w_tiled = 1 there): 8-KiB tile images.
extern "C" int skinny_tile_weight_u8(int N, int K, const void *W, int64_t ldw, void *Wt, void *stream) {
    if (N <= 0 || K <= 0 || (N % kTileRows) || (K % kKB) || ldw < K || (ldw & 15)) return CHIRRUP_E_SHAPE;
    if (!W || !Wt) return CHIRRUP_E_NULL;
    if (mis16(W) || mis16(Wt)) return CHIRRUP_E_ALIGN;
    const int64_t total = (int64_t)N * K / 16;
    hipLaunchKernelGGL(tile_weight_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, K, static_cast<const uint8_t *>(W), ldw, static_cast<uint8_t *>(Wt));
    return (int)hipGetLastError();
}

// Re-lay a binary16 weight matrix W [N][K] (N % 128 == 0, K % 64 == 0) as contiguous 16-KiB tile images for the GEMM
// kernels (w_tiled = 1 in the GEMM calls).  Wt needs N*K elements; W and Wt must not overlap.
extern "C" int skinny_tile_weight(int N, int K, const void *W, int64_t ldw, void *Wt, void *stream) {
    if (N <= 0 || K <= 0 || (N % kTileRows) || (K % kKB) || ldw < K || (ldw & 7)) return CHIRRUP_E_SHAPE;
    if (!W || !Wt) return CHIRRUP_E_NULL;
    if (mis16(W) || mis16(Wt)) return CHIRRUP_E_ALIGN;
    const int64_t total = (int64_t)N * K / 8;
    hipLaunchKernelGGL(tile_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, K, static_cast<const f16 *>(W), ldw, static_cast<f16 *>(Wt));
    return (int)hipGetLastError();
}

// The split count a call with these arguments uses (splits = 0: the library's choice) -- for sizing partial buffers.
extern "C" int skinny_gemm_splits(int N, int K, int Z, int splits) {
    if (N <= 0 || K <= 0 || Z <= 0 || (K % kKB)) return 0;
    return pick_splits(choose_bn(N), N, K, splits, Z);
}

extern "C" int64_t skinny_gemm_workspace_bytes(int M, int N, int K, int splits) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = pick_splits(choose_bn(N), N, K, splits);
    return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

// Row halves (row_halves = 1 in the entry points below, honoured for M > 32 with the 128-column kernel): every tile and
// K-slice is worked on by TWO workgroups, one per half of the rows.  That doubles the workgroups without doubling the
// partial planes (or makes an unsplit launch fill the chip: no partials, no reduce launch at all); the price is that both
// stream the W tile -- neighbours in dispatch order on one XCD, so HBM sees it once.  With splits = 0 the split count
// is chosen for twice the tiles.
namespace {
inline bool use_halves(int row_halves, int M, int bn) { return row_halves && M > 32 && bn == 128; }
inline int tiles_of(int M, bool halves) { return halves ? ((M + 15) / 16 + 1) / 2 : (M + 15) / 16; }
}  // namespace

// tile_counters (skinny_gemm_f16, skinny_gemm_f16_group; may be NULL): skinny_gemm_pair_counters() ints, ZERO before their
// first use and used by no other launch that may run concurrently.  With them a launch of at most kPairMaxRows rows whose
// split count is 2..4 (and (splits - 1) * rows <= 96) needs no reduce launch (EPI_PAIR); the counters are zero again when it ends.
constexpr int kPairCounters = 4096, kPairMaxRows = 32;   // A/B at 7.2B: bsz 16 -3 %, 48 +-0, 64 +1 % (profiles/r02_gemm_experiments.txt section 11)
extern "C" int skinny_gemm_pair_counters(void) { return kPairCounters; }
namespace {
inline bool use_pair(int s, bool halves, int bn, int M, int tiles, const void *counters) {
    // (s - 1) slabs of M x 128 binary32 values are read by the last arriver: at most 48 KB.
    // The fence-free hand-off is the one-workgroup-per-CU form of MI355X_MICROARCH.md: hold that by construction -- a workgroup
    // of this launch must take more than half of the CU's 160 KiB of LDS (today 108-120 KiB at 1-2 x tiles; a shallower ring
    // would silently allow two per CU).
    const bool one_per_cu = lds_bytes(bn, (M + 15) / 16, false) > 80 * 1024;
    return s >= 2 && s <= kPairMaxSlices && (s - 1) * M <= 96 && !halves && bn == 128 && M <= kPairMaxRows && counters && one_per_cu &&
           tiles + 8 <= kPairCounters && !(reinterpret_cast<uintptr_t>(counters) & 3);
}
}  // namespace

// Y = act(X . W^T + bias);  W binary16 [N][K] (row stride ldw).  act: 0 none, 1 relu^2.
extern "C" int skinny_gemm_f16(int M, int N, int K, const void *X, int ldx, const void *W, int64_t ldw, int w_tiled,
                               const void *bias, void *Y, int ldy, int act, int splits, int row_halves, void *workspace,
                               void *tile_counters, void *stream) {
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 3) || (K % kKB) || ldx < K || ldw < K || ldy < N || (ldx & 7) || (ldw & 7) || (ldy & 3))
        return CHIRRUP_E_SHAPE;
    if (act < 0 || act > 1) return CHIRRUP_E_UNSUPPORTED;
    if (w_tiled && (N % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    if (!X || !W || !Y) return CHIRRUP_E_NULL;
    if (mis16(X) || mis16(W) || (reinterpret_cast<uintptr_t>(Y) & 7) || (reinterpret_cast<uintptr_t>(bias) & 7)) return CHIRRUP_E_ALIGN;
    const int bn = choose_bn(N);
    const bool halves = use_halves(row_halves, M, bn);
    const int s = pick_splits(bn, N, K, splits, halves ? 2 : 1);
    const bool partial = s > 1;                        // unsplit: bias and relu^2 run in the kernel's own epilogue
    if (partial && !workspace) return CHIRRUP_E_NULL;
    const int MT = tiles_of(M, halves);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + bn - 1) / bn, halves ? 2 * s : s);
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    bs.row_halves = halves ? 1 : 0;
    if (use_pair(s, halves, bn, M, (int)grid.x, tile_counters)) {
        bs.relu_sq = act == 1 ? 1 : 0;
        bs.counters = static_cast<int *>(tile_counters);
        return launch_gemm<false, EPI_PAIR>(bn, MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy, (const f16 *)bias,
                                            (float *)workspace, bs);
    }
    bs.relu_sq = (!partial && act == 1) ? 1 : 0;
    int rc = partial ? launch_gemm<false, EPI_PARTIAL>(bn, MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                       (const f16 *)bias, (float *)workspace, bs)
                     : launch_gemm<false, EPI_F16>(bn, MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                   (const f16 *)bias, (float *)workspace, bs);
    if (rc) return rc;
    if (partial) {
        rc = launch_reduce(wide_ok(N, ldy, Y), 1, st, M, N, s, (const float *)workspace, (const f16 *)bias, nullptr, nullptr, nullptr,
                           act ? 1 : 0, (f16 *)Y, ldy);
    }
    return rc;
}

extern "C" int64_t skinny_gemm_batched_workspace_bytes(int Z, int M, int N, int K, int splits) {
    if (Z <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)Z * pick_splits(choose_bn(N), N, K, splits, Z) * M * N * (int64_t)sizeof(float);
}

// Z independent problems in ONE launch: Y[z] = act(X[z] . W[z]^T + bias[z]); operands of problem z start z * (their
// batch stride, in elements) after problem 0.  act: 0 none, 1 relu^2, 4 + p = LoRA hidden planes starting at plane p.
extern "C" int skinny_gemm_f16_batched(int Z, int M, int N, int K, const void *X, int ldx, int64_t x_bs, const void *W,
                                       int64_t ldw, int64_t w_bs, const void *bias, int64_t bias_bs, void *Y, int ldy,
                                       int64_t y_bs, int act, int splits, void *workspace, void *stream) {
    return skinny_gemm_f16_grouped(Z, M, N, K, nullptr, X, ldx, x_bs, W, ldw, w_bs, 0, bias, bias_bs, Y, ldy, y_bs, act, splits, 0,
                                   workspace, stream);
}

// As skinny_gemm_f16_batched, with a reduction length per problem: problem z multiplies only the first k_of[z] columns
// of X[z] and W[z] (k_of[z] <= K, a multiple of 64; k_of == NULL: K for all).  For operands that are zero-padded to a
// common K (RWKV-7's LoRA ranks: 96 / 128 / 128 / 480 packed as 512) the padding is then never read.  Z <= 8, and
// splits must be 1 when k_of is given.  w_tiled: every W[z] is a tile image of the [N][K] matrix (skinny_tile_weight).
// row_halves = 1 (unsplit launches without activation only): each problem runs as two sets of workgroups over the upper and
// lower half of the rows -- twice the workgroups and a deeper operand ring for problems with few K-blocks.
extern "C" int skinny_gemm_f16_grouped(int Z, int M, int N, int K, const int *k_of, const void *X, int ldx, int64_t x_bs,
                                       const void *W, int64_t ldw, int64_t w_bs, int w_tiled, const void *bias, int64_t bias_bs,
                                       void *Y, int ldy, int64_t y_bs, int act, int splits, int row_halves, void *workspace,
                                       void *stream) {
    if (Z <= 0 || Z > 65535 || M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 3) || (K % kKB) || ldx < K || ldw < K || ldy < N ||
        (ldx & 7) || (ldw & 7) || (ldy & 3) || (x_bs & 7) || (w_bs & 7) || (y_bs & 3) || act < 0 || act > 7 || act == 2 || act == 3)
        return CHIRRUP_E_SHAPE;
    if (!X || !W || !Y) return CHIRRUP_E_NULL;
    if (mis16(X) || mis16(W) || (reinterpret_cast<uintptr_t>(Y) & 7) || (reinterpret_cast<uintptr_t>(bias) & 7) || (bias_bs & 3))
        return CHIRRUP_E_ALIGN;
    const int bn = choose_bn(N);
    const int s = pick_splits(bn, N, K, splits, Z);
    const bool partial = s > 1 || act != 0;
    if (partial && !workspace) return CHIRRUP_E_NULL;
    if (w_tiled && (N % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    const bool halves = !partial && use_halves(row_halves, M, bn);
    const int MT = tiles_of(M, halves);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + bn - 1) / bn, halves ? 2 * s : s, Z);
    BatchStrides bs{};
    bs.x = x_bs, bs.w = w_bs, bs.y = y_bs, bs.bias = bias_bs;
    bs.tiled = w_tiled ? 1 : 0;
    bs.row_halves = halves ? 1 : 0;
    if (k_of) {
        if (Z > 8 || splits != 1) return CHIRRUP_E_UNSUPPORTED;
        for (int z = 0; z < Z; z++) {
            if (k_of[z] <= 0 || k_of[z] > K || (k_of[z] % kKB)) return CHIRRUP_E_SHAPE;
            bs.k[z] = k_of[z];
        }
    }
    int rc = partial ? launch_gemm<false, EPI_PARTIAL>(bn, MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                       (const f16 *)bias, (float *)workspace, bs)
                     : launch_gemm<false, EPI_F16>(bn, MT, grid, st, M, N, K, K / s, (const f16 *)X, ldx, W, ldw, (f16 *)Y, ldy,
                                                   (const f16 *)bias, (float *)workspace, bs);
    if (rc) return rc;
    if (partial) {
        rc = launch_reduce(wide_ok(N, ldy, Y, y_bs), Z, st, M, N, s, (const float *)workspace, (const f16 *)bias, nullptr, nullptr, nullptr,
                           act, (f16 *)Y, ldy, y_bs, bias_bs);
    }
    return rc;
}

namespace {
int group_bn_and_splits(int count, const chirrup_gemm_problem *problems, int K, int splits, int &s_out, bool halves = false) {
    int max_n = 0, sum_groups = 0;
    for (int i = 0; i < count; i++) max_n = problems[i].n > max_n ? problems[i].n : max_n;
    const int bn = choose_bn(max_n);
    for (int i = 0; i < count; i++) sum_groups += (problems[i].n + bn - 1) / bn;
    if (splits > 0) {
        s_out = splits;
    } else {                                           // as pick_splits, over the launch's exact tile list
        if (halves) sum_groups *= 2;
        int s = (256 + sum_groups / 2) / sum_groups;
        const int max_s = K / 256 > 0 ? K / 256 : 1;
        s = s > max_s ? max_s : (s < 1 ? 1 : s);
        while (s > 1 && (K / kKB) % s) s--;
        s_out = s;
    }
    return bn;
}
}  // namespace

extern "C" int64_t skinny_gemm_group_workspace_bytes(int count, const chirrup_gemm_problem *problems, int M, int K, int splits) {
    if (count <= 0 || count > 8 || !problems || M <= 0 || K <= 0 || (K % kKB) || splits < 0) return 0;
    int s;
    group_bn_and_splits(count, problems, K, splits, s);
    int64_t b = 0;
    for (int i = 0; i < count; i++) b += ((int64_t)s * M * problems[i].n * (int64_t)sizeof(float) + 255) / 256 * 256;
    return b;
}

// Up to 8 GEMMs that share M, K, the row strides of x and W and the split count, in ONE launch: y_i = act_i(x_i . w_i^T +
// bias_i); blockIdx.x runs over the exact list of the problems' N-groups.  Unsplit (splits = 1, or the library's choice
// with row_halves): bias and activation run in the GEMM epilogue; split: through binary32 partials + one reduce launch.
extern "C" int skinny_gemm_f16_group(int count, const chirrup_gemm_problem *problems, int M, int K, int ldx, int64_t ldw,
                                     int splits, int row_halves, void *workspace, void *tile_counters, void *stream) {
    if (count <= 0 || count > 8 || !problems) return CHIRRUP_E_SHAPE;
    if (M <= 0 || M > 256 || K <= 0 || (K % kKB) || ldx < K || ldw < K || (ldx & 7) || (ldw & 7) || splits < 0 ||
        (splits > 0 && ((K / kKB) % splits)))
        return CHIRRUP_E_SHAPE;
    int s, s_plain;
    group_bn_and_splits(count, problems, K, splits, s_plain);
    const int bn = group_bn_and_splits(count, problems, K, splits, s, true);
    const bool halves = use_halves(row_halves, M, bn);
    if (!halves) s = s_plain;
    if (s > 1 && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255))) return workspace ? CHIRRUP_E_ALIGN : CHIRRUP_E_NULL;
    GroupTable gt{};
    gt.used = count;
    int max_n = 0;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    for (int i = 0; i < count; i++) {
        const chirrup_gemm_problem &q = problems[i];
        if (q.n <= 0 || (q.n & 3) || q.ldy < q.n || (q.ldy & 3) || q.act < 0 || q.act > 3) return CHIRRUP_E_SHAPE;
        if (!q.x || !q.w || !q.y) return CHIRRUP_E_NULL;
        if (mis16(q.x) || mis16(q.w) || (reinterpret_cast<uintptr_t>(q.y) & 7) || (reinterpret_cast<uintptr_t>(q.bias) & 7)) return CHIRRUP_E_ALIGN;
        gt.X[i] = static_cast<const f16 *>(q.x), gt.W[i] = q.w, gt.Y[i] = static_cast<f16 *>(q.y);
        gt.bias[i] = static_cast<const f16 *>(q.bias), gt.N[i] = q.n, gt.ldy[i] = q.ldy, gt.act[i] = q.act;
        if (q.w_tiled && (q.n % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
        gt.tiled[i] = q.w_tiled ? 1 : 0;
        if (s > 1) {
            gt.part[i] = reinterpret_cast<float *>(ws);
            ws += ((int64_t)s * M * q.n * (int64_t)sizeof(float) + 255) / 256 * 256;
        }
        max_n = q.n > max_n ? q.n : max_n;
        gt.first[i + 1] = gt.first[i] + (q.n + bn - 1) / bn;
    }
    const int MT = tiles_of(M, halves);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(gt.first[count], halves ? 2 * s : s, 1);
    BatchStrides bs{};
    bs.row_halves = halves ? 1 : 0;
    if (s == 1)
        return launch_gemm<false, EPI_F16>(bn, MT, grid, st, M, max_n, K, K, gt.X[0], ldx, gt.W[0], ldw, gt.Y[0], gt.ldy[0], nullptr,
                                           nullptr, bs, gt);
    if (use_pair(s, halves, bn, M, (int)grid.x, tile_counters)) {
        bs.counters = static_cast<int *>(tile_counters);
        return launch_gemm<false, EPI_PAIR>(bn, MT, grid, st, M, max_n, K, K / s, gt.X[0], ldx, gt.W[0], ldw, gt.Y[0], gt.ldy[0], nullptr,
                                            gt.part[0], bs, gt);
    }
    int rc = launch_gemm<false, EPI_PARTIAL>(bn, MT, grid, st, M, max_n, K, K / s, gt.X[0], ldx, gt.W[0], ldw, gt.Y[0], gt.ldy[0],
                                             nullptr, gt.part[0], bs, gt);
    if (rc) return rc;
    bool wide = true;
    for (int i = 0; i < count; i++) wide = wide && wide_ok(gt.N[i], gt.ldy[i], gt.Y[i]);
    return launch_reduce(wide, count, st, M, max_n, s, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, 0, gt);
}

// ------------------------------------------------------------------------------------------------
// R/K/V + the whole LoRA chain of one RWKV-7 layer in ONE launch (chain_gemm_kernel; rwkv7.py:625-630, :637).
//   main[i]  (i < n_main <= 4): y_i = x_i . w_i^T   (w tile-image, or row-major with row stride ldw)          [M][n_i]
//   lora[p]  (p < n_lora <= 4): hid_p = act_p(x_p . wd_p^T)  [M][n_p of ld_hid],  y_p = hid_p[:, :k_up_p] . wu_p^T + bias_p  [M][up_n]
// All x share M (1..256 rows; whole-row tiles need M <= 128), K and ldx; wd_p row-major [n_p][K] with row stride ldw (n_p % 64 == 0, rows past n_p of a
// 128-row tile read as zeros); wu_p tile images of [up_n][up_kimg] (k_up_p <= up_kimg, both % 64 == 0; up_n % 128 == 0).
// workspace: rwkv7_tmix_gemms_workspace_bytes(...) bytes of hipMalloc'ed memory; sync: rwkv7_tmix_sync_words() ints, ZERO before the
// first use, zero again after every completed launch (a launch that did not complete: zero them yourself); used by one
// launch at a time.  The STATUS word (`status`, an int32 of the caller's; NULL: sync[rwkv7_tmix_status_word()], the LAST sync word) is
// sticky: launches only ever OR into it -- non-zero after a launch whose bounded waits gave up (another tenant holding most of
// the chip for > spin budget): that launch's LoRA outputs are then undefined.  Words [0, rwkv7_tmix_status_word()) may be zeroed
// between launches at any time; the status word is cleared by its owner only.
extern "C" int rwkv7_tmix_sync_words(void) { return kChainWords; }
extern "C" int rwkv7_tmix_status_word(void) { return kChainWords - 1; }

namespace {
// K-slices per down-projection tile (half): as many as the idle CUs can take at once -- the chain's latency at <= 128 rows is
// what bounds the launch (tools/chain_stamps.py), and an 8-way slice streams 1/8 of K before the first hand-off
int chain_dsplits(int K, int down_tile_halves = 14, int spare_wgs = 64) {
    const int kb = K / kKB;
    for (int s = kChainMaxSplits; s > 1; s >>= 1)
        if (kb % s == 0 && kb / s >= 4 && down_tile_halves * s <= (spare_wgs > 64 ? spare_wgs : 64)) return s;
    return 1;
}
}  // namespace

namespace {
// how a time-mix launch is cut: two row halves per tile from 128 rows (the R/K/V tiles then run unsplit), whole rows below;
// at <= 32 rows the R/K/V tiles are split 2..4 ways over K and reduced inside the launch (as ring_gemm_kernel's EPI_PAIR)
void chain_plan(int M, int K, int main_tiles, int row_halves, int &halves, int &rkv_splits) {
    halves = (row_halves && M > 32) ? 2 : 1;
    rkv_splits = 1;
    if (halves == 1 && M <= 64) {                      // (64: two slices of a 96-tile layer, one 32-KB partial per other slice)
        int s = (192 + main_tiles / 2) / main_tiles;
        const int max_s = K / 256 > 0 ? K / 256 : 1;
        s = s > max_s ? max_s : s;
        s = s > kPairMaxSlices ? kPairMaxSlices : (s < 1 ? 1 : s);
        while (s > 1 && ((K / kKB) % s || (s - 1) * M > 96)) s--;
        rkv_splits = s;
    }
}
}  // namespace

extern "C" int64_t rwkv7_tmix_gemms_workspace_bytes(int M, int K, int n_main, const chirrup_gemm_problem *main_p, int n_lora,
                                                    const chirrup_lora_problem *lora, int row_halves) {
    if (M <= 0 || K <= 0 || (K % kKB) || n_lora <= 0 || n_lora > 4 || !lora || n_main <= 0 || n_main > 4 || !main_p) return 0;
    int tiles = 0, main_tiles = 0;
    for (int p = 0; p < n_lora; p++) tiles += (lora[p].n + kTileRows - 1) / kTileRows;
    for (int i = 0; i < n_main; i++) main_tiles += (main_p[i].n + kTileRows - 1) / kTileRows;
    int halves, rs;
    chain_plan(M, K, main_tiles, row_halves, halves, rs);
    const int MT = tiles_of(M, halves == 2);
    int64_t b = (int64_t)tiles * halves * kChainMaxSplits * (MT * 16) * kTileRows * (int64_t)sizeof(float);      // (upper bound: the split count depends on the free CUs)
    if (rs > 1)
        for (int i = 0; i < n_main; i++) b += ((int64_t)rs * M * main_p[i].n * (int64_t)sizeof(float) + 255) / 256 * 256;
    return b + 256;
}

namespace {
// compute units of the current device (256 on a whole MI355X; 32 on a CPX partition), cached per device ordinal
int device_cus() {
    static std::atomic<int> cus[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int n = cus[dev & 31].load(std::memory_order_acquire);
    if (n <= 0) {
        n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev & 31].store(n, std::memory_order_release);
    }
    return n;
}
}  // namespace

extern "C" int chirrup_device_cu_count(void) { return device_cus(); }

namespace {
// the launch behind rwkv7_tmix_gemms (binary16 main problems in gt) and rwkv7_tmix_gemms_mm8 (uint8: w8 = true, scales / row sums in m_*)
int tmix_launch(bool w8, int M, int K, int ldx, int64_t ldw, GroupTable &gt, const f16 *const *m_rx, const f16 *const *m_mx,
                const float *const *m_S, int n_lora, const chirrup_lora_problem *lora, int ld_hid, int up_n, int up_kimg, int up_ldy,
                int row_halves, void *workspace, void *sync, void *status, int spin_limit, void *stream) {
    const int n_main = gt.used;
    if (n_lora <= 0 || n_lora > 4 || !lora) return CHIRRUP_E_SHAPE;
    if (M <= 0 || M > 256 || K <= 0 || (K % kKB) || ldx < K || ldw < K || (ldx & 7) || (ldw & (w8 ? 15 : 7))) return CHIRRUP_E_SHAPE;
    if (up_n <= 0 || (up_n % kTileRows) || up_kimg <= 0 || (up_kimg % kKB) || up_ldy < up_n || (up_ldy & 3) || ld_hid < up_kimg || (ld_hid & 7))
        return CHIRRUP_E_SHAPE;
    if (!workspace || !sync) return CHIRRUP_E_NULL;
    if (mis16(workspace) || (reinterpret_cast<uintptr_t>(sync) & 15)) return CHIRRUP_E_ALIGN;
    ChainTable ct{};
    ct.n_lora = n_lora, ct.ld_hid = ld_hid, ct.up_N = up_n, ct.up_Kimg = up_kimg, ct.up_ldy = up_ldy;
    chain_plan(M, K, gt.first[n_main], row_halves, ct.halves, ct.rkv_splits);
    if (w8) {
        for (int i = 0; i < n_main; i++) ct.m_rx[i] = m_rx[i], ct.m_mx[i] = m_mx[i], ct.m_S[i] = m_S[i];
    }
    for (int p = 0; p < n_lora; p++) {
        const chirrup_lora_problem &q = lora[p];
        if (q.n <= 0 || (q.n % kKB) || q.n > ld_hid || q.k_up <= 0 || (q.k_up % kKB) || q.k_up > q.n || q.k_up > up_kimg || q.act < 0 || q.act > 3)
            return CHIRRUP_E_SHAPE;
        if (!q.x || !q.w || !q.hid || !q.w_up || !q.y) return CHIRRUP_E_NULL;
        if (mis16(q.x) || mis16(q.w) || mis16(q.hid) || mis16(q.w_up) || (reinterpret_cast<uintptr_t>(q.y) & 7) || (reinterpret_cast<uintptr_t>(q.bias) & 7))
            return CHIRRUP_E_ALIGN;
        ct.dX[p] = static_cast<const f16 *>(q.x), ct.dW[p] = static_cast<const f16 *>(q.w), ct.hid[p] = static_cast<f16 *>(q.hid);
        ct.uW[p] = q.w_up, ct.ubias[p] = static_cast<const f16 *>(q.bias), ct.uY[p] = static_cast<f16 *>(q.y);
        ct.dN[p] = q.n, ct.dact[p] = q.act, ct.uK[p] = q.k_up;
        ct.dfirst[p + 1] = ct.dfirst[p] + (q.n + kTileRows - 1) / kTileRows;
    }
    const int n_dtiles = ct.dfirst[n_lora];
    const int MT = tiles_of(M, ct.halves == 2);
    if (n_dtiles * ct.halves > kChainTickets || gt.first[n_main] > kChainPairs) return CHIRRUP_E_UNSUPPORTED;
    const int main_wgs = (gt.first[n_main] * ct.halves * ct.rkv_splits + 15) / 16 * 16;   // whole runs of the XCD-aware tile order
    const int n_cus = device_cus();                                  // (round-3 advisor finding: 256 was hard-coded)
    const int spare = n_cus > main_wgs ? (n_cus - main_wgs) / 8 * 8 : 0;      // every CU the R/K/V tiles leave idle
    ct.dsplits = chain_dsplits(K, n_dtiles * ct.halves, spare);
    // workspace: the down-projection slabs, then (split R/K/V) one run of partial planes per R/K/V problem
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    ct.slab = reinterpret_cast<float *>(ws);
    ws += ((int64_t)n_dtiles * ct.halves * ct.dsplits * (MT * 16) * kTileRows * (int64_t)sizeof(float) + 255) / 256 * 256;
    if (ct.rkv_splits > 1)
        for (int i = 0; i < n_main; i++) {
            gt.part[i] = reinterpret_cast<float *>(ws);
            ws += ((int64_t)ct.rkv_splits * M * gt.N[i] * (int64_t)sizeof(float) + 255) / 256 * 256;
        }
    ct.sync = static_cast<int *>(sync);
    ct.status = status ? static_cast<int *>(status) : ct.sync + kChainWords - 1;
    ct.n_chain = (n_dtiles * ct.halves * ct.dsplits + 7) / 8 * 8;   // the down-projection slices ... and every idle CU, for the up-projections
    static const int chain_max = [] { const char *e = getenv("CHIRRUP_CHAIN_MAX"); return e ? atoi(e) : 96; }();      // (tuning / A-B only)
    static const int chain_warm = [] { const char *e = getenv("CHIRRUP_CHAIN_WARM"); return e ? atoi(e) : 1; }();
    if (spare > ct.n_chain) ct.n_chain = spare < chain_max ? spare : (chain_max > ct.n_chain ? chain_max / 8 * 8 : ct.n_chain);
    // chain workgroups wait for each other (a tile's slices, the up-projection shares for the hidden tiles): all of them must be
    // resident at once, one per CU (160 KB of LDS each) -- on a CU-partitioned device they might not be: refuse, the caller has the
    // two-launch form (skinny_gemm_f16_group + the batched up-projection launch)
    if (ct.n_chain > n_cus) return CHIRRUP_E_UNSUPPORTED;
    ct.warm = chain_warm;
    ct.spin_limit = spin_limit > 0 ? spin_limit : 400000;          // x ~0.25 us of s_sleep: ~0.1 s
    const dim3 grid(ct.n_chain + main_wgs);
    ct.stamps = (g_clock_probe && g_clock_pairs >= (int)grid.x * 4) ? g_clock_probe : nullptr;
    const size_t lds = lds_bytes(128, MT, false);       // (>= the uint8 main tiles' ring: the chain workgroups stream binary16 weights)
    hipStream_t st = static_cast<hipStream_t>(stream);
#define CHAIN_GO_K(MTV, W8V)                                                                                                   \
    do {                                                                                                                       \
        auto kern = chain_gemm_kernel<MTV, W8V>;                                                                               \
        static std::atomic<bool> lds_limit_raised[32];                                                                         \
        int dev_ = 0;                                                                                                          \
        (void)hipGetDevice(&dev_);                                                                                             \
        if (!lds_limit_raised[dev_ & 31].load(std::memory_order_acquire)) {                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            lds_limit_raised[dev_ & 31].store(true, std::memory_order_release);                                                \
        }                                                                                                                      \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, M, K, ldx, ldw, gt, ct);                                            \
    } while (0)
#define CHAIN_GO(MTV)                     \
    do {                                  \
        if (w8) CHAIN_GO_K(MTV, true);    \
        else CHAIN_GO_K(MTV, false);      \
    } while (0)
    switch (MT) {
        case 1: CHAIN_GO(1); break;
        case 2: CHAIN_GO(2); break;
        case 3: CHAIN_GO(3); break;
        case 4: CHAIN_GO(4); break;
        case 5: CHAIN_GO(5); break;
        case 6: CHAIN_GO(6); break;
        case 7: CHAIN_GO(7); break;
        case 8: CHAIN_GO(8); break;
        default: return CHIRRUP_E_UNSUPPORTED;         // whole rows above 128: not used (row halves from 128 rows on)
    }
#undef CHAIN_GO
#undef CHAIN_GO_K
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int rwkv7_tmix_gemms(int M, int K, int ldx, int64_t ldw, int n_main, const chirrup_gemm_problem *main_p, int n_lora,
                                const chirrup_lora_problem *lora, int ld_hid, int up_n, int up_kimg, int up_ldy, int row_halves,
                                void *workspace, void *sync, void *status, int spin_limit, void *stream) {
    if (n_main <= 0 || n_main > 4 || !main_p) return CHIRRUP_E_SHAPE;
    GroupTable gt{};
    gt.used = n_main;
    for (int i = 0; i < n_main; i++) {
        const chirrup_gemm_problem &q = main_p[i];
        if (q.n <= 0 || (q.n & 3) || q.ldy < q.n || (q.ldy & 3) || q.act < 0 || q.act > 3) return CHIRRUP_E_SHAPE;
        if (!q.x || !q.w || !q.y) return CHIRRUP_E_NULL;
        if (mis16(q.x) || mis16(q.w) || (reinterpret_cast<uintptr_t>(q.y) & 7) || (reinterpret_cast<uintptr_t>(q.bias) & 7)) return CHIRRUP_E_ALIGN;
        if (q.w_tiled && (q.n % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
        gt.X[i] = static_cast<const f16 *>(q.x), gt.W[i] = q.w, gt.Y[i] = static_cast<f16 *>(q.y), gt.bias[i] = static_cast<const f16 *>(q.bias);
        gt.N[i] = q.n, gt.ldy[i] = q.ldy, gt.act[i] = q.act, gt.tiled[i] = q.w_tiled ? 1 : 0;
        gt.first[i + 1] = gt.first[i] + (q.n + kTileRows - 1) / kTileRows;
    }
    return tmix_launch(false, M, K, ldx, ldw, gt, nullptr, nullptr, nullptr, n_lora, lora, ld_hid, up_n, up_kimg, up_ldy, row_halves, workspace,
                       sync, status, spin_limit, stream);
}

// The same launch with uint8 (mm8, w8a16) main problems: y = mm8(x, w) in the reference's split form -- xs = the activation
// prologue binary16(x * ry) [M][ldx], S [M][3] its row sums {sum xs, sum x*my, sum x} (both written by rwkv7_add_ln_mix_mm8 with
// out_planes = 3), wT the K-contiguous uint8 weights [n][K] (tile images of skinny_tile_weight_u8 when w_tiled, else row stride
// ldw), rx / mx [n] the column scales; the rank-1 corrections run in each tile's epilogue (as mm8t_gemm_fused).  n % 8 == 0,
// ldy % 8 == 0, 16-byte aligned y / rx / mx.  The LoRA problems stay binary16.
extern "C" int rwkv7_tmix_gemms_mm8(int M, int K, int ldx, int64_t ldw, int n_main, const chirrup_mm8_problem *main_p, int n_lora,
                                    const chirrup_lora_problem *lora, int ld_hid, int up_n, int up_kimg, int up_ldy, int row_halves,
                                    void *workspace, void *sync, void *status, int spin_limit, void *stream) {
    if (n_main <= 0 || n_main > 4 || !main_p) return CHIRRUP_E_SHAPE;
    GroupTable gt{};
    gt.used = n_main;
    const f16 *rx[4], *mx[4];
    const float *S[4];
    for (int i = 0; i < n_main; i++) {
        const chirrup_mm8_problem &q = main_p[i];
        if (q.n <= 0 || (q.n & 7) || q.ldy < q.n || (q.ldy & 7)) return CHIRRUP_E_SHAPE;
        if (!q.xs || !q.w || !q.y || !q.rx || !q.mx || !q.S) return CHIRRUP_E_NULL;
        if (mis16(q.xs) || mis16(q.w) || mis16(q.y) || mis16(q.rx) || mis16(q.mx)) return CHIRRUP_E_ALIGN;
        if (q.w_tiled && (q.n % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
        gt.X[i] = static_cast<const f16 *>(q.xs), gt.W[i] = q.w, gt.Y[i] = static_cast<f16 *>(q.y), gt.bias[i] = nullptr;
        gt.N[i] = q.n, gt.ldy[i] = q.ldy, gt.act[i] = 0, gt.tiled[i] = q.w_tiled ? 1 : 0;
        gt.first[i + 1] = gt.first[i] + (q.n + kTileRows - 1) / kTileRows;
        rx[i] = static_cast<const f16 *>(q.rx), mx[i] = static_cast<const f16 *>(q.mx), S[i] = q.S;
    }
    return tmix_launch(true, M, K, ldx, ldw, gt, rx, mx, S, n_lora, lora, ld_hid, up_n, up_kimg, up_ldy, row_halves, workspace, sync,
                       status, spin_limit, stream);
}

extern "C" int skinny_gemm_f16_partial(int M, int N, int K, const void *X, int ldx, const void *W, int64_t ldw, int w_tiled,
                                       int splits, int row_halves, float *partials, void *stream) {
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 3) || (K % kKB) || ldx < K || ldw < K || (ldx & 7) || (ldw & 7))
        return CHIRRUP_E_SHAPE;
    if (!X || !W || !partials) return CHIRRUP_E_NULL;
    if (mis16(X) || mis16(W) || mis16(partials)) return CHIRRUP_E_ALIGN;
    if (w_tiled && (N % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    const int bn = choose_bn(N);
    const bool halves = use_halves(row_halves, M, bn);
    const int s = pick_splits(bn, N, K, splits, halves ? 2 : 1);
    const int MT = tiles_of(M, halves);
    const dim3 grid((N + bn - 1) / bn, halves ? 2 * s : s);
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    bs.row_halves = halves ? 1 : 0;
    const int rc = launch_gemm<false, EPI_PARTIAL>(bn, MT, grid, static_cast<hipStream_t>(stream), M, N, K, K / s, (const f16 *)X, ldx,
                                                   W, ldw, nullptr, N, nullptr, partials, bs);
    return rc ? -1000 - rc : s;
}

// mm8 with K-contiguous ("packed", [M_out][N_in]) uint8 weights.  Same quantisation and formula as
// mm8_seq; evaluated in the split form: xs = fp16(x*ry) through MFMA, rank-1 corrections after.
// workspace layout: xs [B][N_in] f16 | S [B][3] f32 | partials [splits][B][M_out] f32
extern "C" int64_t mm8t_workspace_bytes(int B, int N_in, int M_out, int splits) {
    if (B <= 0 || N_in <= 0 || M_out <= 0) return 0;
    if (B > 256) B = 256;                                  // more rows are processed 256 at a time
    const int s = pick_splits(choose_bn(M_out), M_out, N_in, splits);
    int64_t b = (int64_t)B * N_in * 2;
    b = (b + 255) / 256 * 256;
    b += 256 * ((B * 3 * 4 + 255) / 256);
    b += (int64_t)s * B * M_out * 4;
    return b;
}

extern "C" int mm8t_seq(int B, int N_in, int M_out, const void *x, int x_stride, const void *wT, int64_t w_stride, int w_tiled,
                        const void *mx, const void *rx, const void *my, const void *ry, void *y, int y_stride, int act,
                        int splits, void *workspace, void *stream) {
    if (w_tiled && (M_out % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    if (B <= 0 || N_in <= 0 || M_out <= 0 || (M_out & 3) || (N_in % kKB) || x_stride < N_in || w_stride < N_in ||
        y_stride < M_out || (x_stride & 7) || (w_stride & 15) || (y_stride & 3))
        return CHIRRUP_E_SHAPE;
    if (!x || !wT || !mx || !rx || !my || !ry || !y || !workspace) return CHIRRUP_E_NULL;
    if (mis16(x) || mis16(wT) || (reinterpret_cast<uintptr_t>(workspace) & 255)) return CHIRRUP_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(mx) & 7) || (reinterpret_cast<uintptr_t>(rx) & 7) || mis16(my) || mis16(ry)) return CHIRRUP_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int bnc = choose_bn(M_out);
    const int s = pick_splits(bnc, M_out, N_in, splits);
    const int Bmax = B < 256 ? B : 256;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    f16 *xs = reinterpret_cast<f16 *>(ws);
    int64_t off = ((int64_t)Bmax * N_in * 2 + 255) / 256 * 256;
    float *S = reinterpret_cast<float *>(ws + off);
    off += 256 * ((Bmax * 3 * 4 + 255) / 256);
    float *part = reinterpret_cast<float *>(ws + off);
    // The kernels hold at most 256 activation rows per weight pass (32 accumulator tiles per wave): a longer batch
    // (chunked prefill) is cut into 256-row blocks that re-stream the weights; the blocks reuse the workspace in
    // stream order.
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int bn = (B - b0) < 256 ? (B - b0) : 256;
        const f16 *xb = static_cast<const f16 *>(x) + (int64_t)b0 * x_stride;
        f16 *yb = static_cast<f16 *>(y) + (int64_t)b0 * y_stride;
        hipLaunchKernelGGL(mm8_prep_kernel, dim3(bn), dim3(256), 0, st, N_in, xb, x_stride, (const f16 *)ry, (const f16 *)my,
                           xs, S);
        const int MT = (bn + 15) / 16;
        const dim3 grid((M_out + bnc - 1) / bnc, s);
        BatchStrides bs{};
        bs.tiled = w_tiled ? 1 : 0;
        int rc = launch_gemm<true, EPI_PARTIAL>(bnc, MT, grid, st, bn, M_out, N_in, N_in / s, xs, N_in, wT, w_stride, yb, y_stride,
                                                nullptr, part, bs);
        if (rc) return rc;
        rc = launch_reduce(wide_ok(M_out, y_stride, yb), 1, st, bn, M_out, s, part, nullptr, (const f16 *)rx, (const f16 *)mx, S, act ? 3 : 2,
                           yb, y_stride);
        if (rc) return rc;
    }
    return 0;
}

// mm8t_seq with the arithmetic of the reference's mm8_seq (kernel_mm_seq_fp16i8, rwkv_pip_operators.cu:59-83: the as-coded
// expression, every product and sum in binary32) at matrix-core speed: x*ry is split exactly into two binary16 operands
// (mm8_prep_kernel, xs_lo) and multiplied in two passes whose binary32 partial planes the reduce adds with the rank-1
// corrections.  What still differs from the as-coded kernel is the ORDER of the binary32 sums (and the regrouping of
// (q + 0.5) * rx * ry + mx + my into the split form): ~1e-6 of the row scale, against 2e-3 for the one-pass split form that
// rounds xs to binary16 (the reference's own mm8_seq_opt does that too, rwkv_pip_wrapper.cpp:148-191).
// workspace layout: xs_hi [B][N_in] f16 | xs_lo [B][N_in] f16 | S [B][3] f32 | partials [2 * splits][B][M_out] f32
extern "C" int64_t mm8t_exact_workspace_bytes(int B, int N_in, int M_out, int splits) {
    if (B <= 0 || N_in <= 0 || M_out <= 0) return 0;
    if (B > 256) B = 256;
    const int s = pick_splits(choose_bn(M_out), M_out, N_in, splits);
    int64_t b = 2 * (((int64_t)B * N_in * 2 + 255) / 256 * 256);
    b += 256 * ((B * 3 * 4 + 255) / 256);
    b += (int64_t)2 * s * B * M_out * 4;
    return b;
}

extern "C" int mm8t_seq_exact(int B, int N_in, int M_out, const void *x, int x_stride, const void *wT, int64_t w_stride, int w_tiled,
                              const void *mx, const void *rx, const void *my, const void *ry, void *y, int y_stride, int act,
                              int splits, void *workspace, void *stream) {
    if (w_tiled && (M_out % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    if (B <= 0 || N_in <= 0 || M_out <= 0 || (M_out & 3) || (N_in % kKB) || x_stride < N_in || w_stride < N_in ||
        y_stride < M_out || (x_stride & 7) || (w_stride & 15) || (y_stride & 3))
        return CHIRRUP_E_SHAPE;
    if (!x || !wT || !mx || !rx || !my || !ry || !y || !workspace) return CHIRRUP_E_NULL;
    if (mis16(x) || mis16(wT) || (reinterpret_cast<uintptr_t>(workspace) & 255)) return CHIRRUP_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(mx) & 7) || (reinterpret_cast<uintptr_t>(rx) & 7) || mis16(my) || mis16(ry)) return CHIRRUP_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int bnc = choose_bn(M_out);
    const int s = pick_splits(bnc, M_out, N_in, splits);
    const int Bmax = B < 256 ? B : 256;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const int64_t xs_bytes = ((int64_t)Bmax * N_in * 2 + 255) / 256 * 256;
    f16 *xs_hi = reinterpret_cast<f16 *>(ws), *xs_lo = reinterpret_cast<f16 *>(ws + xs_bytes);
    int64_t off = 2 * xs_bytes;
    float *S = reinterpret_cast<float *>(ws + off);
    off += 256 * ((Bmax * 3 * 4 + 255) / 256);
    float *part = reinterpret_cast<float *>(ws + off);
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int bn = (B - b0) < 256 ? (B - b0) : 256;
        const f16 *xb = static_cast<const f16 *>(x) + (int64_t)b0 * x_stride;
        f16 *yb = static_cast<f16 *>(y) + (int64_t)b0 * y_stride;
        hipLaunchKernelGGL(mm8_prep_kernel, dim3(bn), dim3(256), 0, st, N_in, xb, x_stride, (const f16 *)ry, (const f16 *)my,
                           xs_hi, S, xs_lo);
        const int MT = (bn + 15) / 16;
        const dim3 grid((M_out + bnc - 1) / bnc, s);
        BatchStrides bs{};
        bs.tiled = w_tiled ? 1 : 0;
        int rc = launch_gemm<true, EPI_PARTIAL>(bnc, MT, grid, st, bn, M_out, N_in, N_in / s, xs_hi, N_in, wT, w_stride, yb, y_stride,
                                                nullptr, part, bs);
        if (rc) return rc;
        rc = launch_gemm<true, EPI_PARTIAL>(bnc, MT, grid, st, bn, M_out, N_in, N_in / s, xs_lo, N_in, wT, w_stride, yb, y_stride,
                                            nullptr, part + (int64_t)s * bn * M_out, bs);
        if (rc) return rc;
        rc = launch_reduce(wide_ok(M_out, y_stride, yb), 1, st, bn, M_out, 2 * s, part, nullptr, (const f16 *)rx, (const f16 *)mx, S, act ? 3 : 2,
                           yb, y_stride);
        if (rc) return rc;
    }
    return 0;
}

// The matrix product of mm8t_seq alone, for callers that fuse its prologue and its reduce into the neighbouring kernels
// (rwkv7_add_ln_mix_mm8, mm8_reduce_rows): xs = the prologue's output [B][N_in] binary16 (B <= 256), partials receive the
// fp32 core sums [splits][B][M_out].  Returns the split count used (> 0) or a negative error like skinny_gemm_f16_partial.
extern "C" int mm8t_gemm_partial(int B, int N_in, int M_out, const void *xs, int xs_stride, const void *wT, int64_t w_stride,
                                 int w_tiled, int splits, int row_halves, float *partials, void *stream) {
    if (w_tiled && (M_out % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    if (B <= 0 || B > 256 || N_in <= 0 || M_out <= 0 || (M_out & 3) || (N_in % kKB) || xs_stride < N_in || w_stride < N_in ||
        (xs_stride & 7) || (w_stride & 15))
        return CHIRRUP_E_SHAPE;
    if (!xs || !wT || !partials) return CHIRRUP_E_NULL;
    if (mis16(xs) || mis16(wT) || mis16(partials)) return CHIRRUP_E_ALIGN;
    const int bn = choose_bn(M_out);
    const bool halves = use_halves(row_halves, B, bn);
    const int s = pick_splits(bn, M_out, N_in, splits, halves ? 2 : 1);
    const dim3 grid((M_out + bn - 1) / bn, halves ? 2 * s : s);
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    bs.row_halves = halves ? 1 : 0;
    const int rc = launch_gemm<true, EPI_PARTIAL>(bn, tiles_of(B, halves), grid, static_cast<hipStream_t>(stream), B, M_out, N_in, N_in / s,
                                                  static_cast<const f16 *>(xs), xs_stride, wT, w_stride, nullptr, M_out, nullptr, partials, bs);
    return rc ? -1000 - rc : s;
}

// mm8t_gemm_partial + mm8_reduce_rows in ONE launch, for products whose N gives an unsplit launch enough workgroups
// (ffn.key: 128 tiles x 2 row halves): the core sums never leave the workgroup; corrections, relu^2 (act = 1), y (may be
// NULL) and the next product's prologue (xs2, S2; may be NULL) come out of the GEMM epilogue.  S: [B][S_parts][3];
// S2: [B][mm8_tile_parts(M_out)][3] -- one partial row sum per 128-column tile, which the consumer adds up in tile order
// (rwkv7_add_ln_mix_mm8: in_S_parts; mm8_reduce_rows / this function: S_parts).  M_out < 32768 (the 128-column kernel).
extern "C" int mm8_tile_parts(int M_out) { return M_out > 0 ? (M_out + kTileRows - 1) / kTileRows : 0; }

extern "C" int mm8t_gemm_fused(int B, int N_in, int M_out, const void *xs, int xs_stride, const void *wT, int64_t w_stride, int w_tiled,
                               const void *rx, const void *mx, const float *S, int S_parts, int act, void *y, int y_stride,
                               const void *ry2, const void *my2, void *xs2, float *S2, int row_halves, void *stream) {
    if (w_tiled && (M_out % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    if (B <= 0 || B > 256 || N_in <= 0 || M_out <= 0 || (M_out & 7) || (N_in % kKB) || xs_stride < N_in || w_stride < N_in ||
        (xs_stride & 7) || (w_stride & 15) || S_parts <= 0 || (y && (y_stride < M_out || (y_stride & 7))))
        return CHIRRUP_E_SHAPE;
    if (!xs || !wT || !rx || !mx || !S || (!y && !xs2)) return CHIRRUP_E_NULL;
    if (xs2 && (!ry2 || !my2 || !S2)) return CHIRRUP_E_NULL;
    if (mis16(xs) || mis16(wT) || mis16(y) || mis16(xs2) || mis16(rx) || mis16(mx) || mis16(ry2) || mis16(my2)) return CHIRRUP_E_ALIGN;
    const int bn = choose_bn(M_out);
    if (bn != 128) return CHIRRUP_E_UNSUPPORTED;
    const bool halves = use_halves(row_halves, B, bn);
    const dim3 grid((M_out + bn - 1) / bn, halves ? 2 : 1);
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    bs.row_halves = halves ? 1 : 0;
    bs.q8.rx = static_cast<const f16 *>(rx), bs.q8.mx = static_cast<const f16 *>(mx), bs.q8.S = S, bs.q8.S_parts = S_parts;
    bs.q8.ry2 = static_cast<const f16 *>(ry2), bs.q8.my2 = static_cast<const f16 *>(my2), bs.q8.xs2 = static_cast<f16 *>(xs2);
    bs.q8.S2 = S2, bs.q8.S2_parts = mm8_tile_parts(M_out), bs.q8.act = act ? 1 : 0;
    return launch_gemm<true, EPI_MM8>(bn, tiles_of(B, halves), grid, static_cast<hipStream_t>(stream), B, M_out, N_in, N_in,
                                      static_cast<const f16 *>(xs), xs_stride, wT, w_stride, static_cast<f16 *>(y), y ? y_stride : M_out,
                                      nullptr, nullptr, bs);
}

// mm8t_gemm_fused for few rows: the product is split `splits` ways over K (2..4; 0 = the library's choice) and the LAST of a
// tile's workgroups to finish adds the other slices' sums to its own (slice order) and runs the same epilogue -- no partials for a
// consumer, no mm8_reduce_rows launch.  partials: room for splits x B x M_out binary32 values (the slices' hand-off slabs);
// tile_counters: as skinny_gemm_f16 (skinny_gemm_pair_counters() ints, zero before first use).  Limits: B <= 64 and
// (splits - 1) * B <= 96; returns CHIRRUP_E_UNSUPPORTED where the in-launch reduction does not apply (callers then use
// mm8t_gemm_partial + mm8_reduce_rows).  The sums are those of mm8_reduce_rows at the same split count, bit for bit.
constexpr int kPairMaxRowsMm8 = 64;
extern "C" int mm8t_gemm_fused_split(int B, int N_in, int M_out, const void *xs, int xs_stride, const void *wT, int64_t w_stride, int w_tiled,
                                     const void *rx, const void *mx, const float *S, int S_parts, int act, void *y, int y_stride,
                                     const void *ry2, const void *my2, void *xs2, float *S2, int splits, float *partials,
                                     void *tile_counters, void *stream) {
    if (w_tiled && (M_out % kTileRows)) return CHIRRUP_E_UNSUPPORTED;
    if (B <= 0 || B > 256 || N_in <= 0 || M_out <= 0 || (M_out & 7) || (N_in % kKB) || xs_stride < N_in || w_stride < N_in ||
        (xs_stride & 7) || (w_stride & 15) || S_parts <= 0 || (y && (y_stride < M_out || (y_stride & 7))))
        return CHIRRUP_E_SHAPE;
    if (!xs || !wT || !rx || !mx || !S || (!y && !xs2) || !partials || !tile_counters) return CHIRRUP_E_NULL;
    if (xs2 && (!ry2 || !my2 || !S2)) return CHIRRUP_E_NULL;
    if (mis16(xs) || mis16(wT) || mis16(y) || mis16(xs2) || mis16(rx) || mis16(mx) || mis16(ry2) || mis16(my2) || mis16(partials) ||
        (reinterpret_cast<uintptr_t>(tile_counters) & 3))
        return CHIRRUP_E_ALIGN;
    const int bn = choose_bn(M_out);
    if (bn != 128) return CHIRRUP_E_UNSUPPORTED;
    const int s = pick_splits(bn, M_out, N_in, splits);
    const int tiles = (M_out + bn - 1) / bn;
    if (s < 2 || s > kPairMaxSlices || (s - 1) * B > 96 || B > kPairMaxRowsMm8 || tiles + 8 > kPairCounters) return CHIRRUP_E_UNSUPPORTED;
    const dim3 grid(tiles, s);
    BatchStrides bs{};
    bs.tiled = w_tiled ? 1 : 0;
    bs.counters = static_cast<int *>(tile_counters);
    bs.min_lds = 81 * 1024;                            // the fence-free hand-off is the one-workgroup-per-CU form (use_pair): the uint8 ring alone is 72 KiB at <= 32 rows
    bs.q8.rx = static_cast<const f16 *>(rx), bs.q8.mx = static_cast<const f16 *>(mx), bs.q8.S = S, bs.q8.S_parts = S_parts;
    bs.q8.ry2 = static_cast<const f16 *>(ry2), bs.q8.my2 = static_cast<const f16 *>(my2), bs.q8.xs2 = static_cast<f16 *>(xs2);
    bs.q8.S2 = S2, bs.q8.S2_parts = mm8_tile_parts(M_out), bs.q8.act = act ? 1 : 0;
    return launch_gemm<true, EPI_PAIR>(bn, tiles_of(B, false), grid, static_cast<hipStream_t>(stream), B, M_out, N_in, N_in / s,
                                       static_cast<const f16 *>(xs), xs_stride, wT, w_stride, static_cast<f16 *>(y), y ? y_stride : M_out,
                                       nullptr, partials, bs);
}

// Row-wise reduce of an mm8 product's partials with its rank-1 corrections (+ relu^2 when act = 1), writing y (may be
// NULL) and/or the activation prologue (xs2, S2 for scales ry2, my2) of the NEXT mm8 product.  M_out % 4 == 0.
// S is given as S_parts partial sums per row ([B][S_parts][3]); S2 is written as mm8_row_parts(M_out) partial sums per row.
extern "C" int mm8_row_parts(int M_out) { return M_out > 0 ? (M_out + kRowPart - 1) / kRowPart : 0; }

extern "C" int mm8_reduce_rows(int B, int M_out, int splits, const float *partials, const void *rx, const void *mx, const float *S,
                               int S_parts, int act, void *y, int y_stride, const void *ry2, const void *my2, void *xs2, float *S2,
                               void *stream) {
    if (B <= 0 || M_out <= 0 || (M_out & 3) || splits <= 0 || S_parts <= 0 || (y && (y_stride < M_out || (y_stride & 3)))) return CHIRRUP_E_SHAPE;
    if (!partials || !rx || !mx || !S || (!y && !xs2)) return CHIRRUP_E_NULL;
    if (xs2 && (!ry2 || !my2 || !S2)) return CHIRRUP_E_NULL;
    if (mis16(partials) || (reinterpret_cast<uintptr_t>(y) & 7) || (reinterpret_cast<uintptr_t>(xs2) & 7) ||
        (reinterpret_cast<uintptr_t>(rx) & 7) || (reinterpret_cast<uintptr_t>(mx) & 7) || (reinterpret_cast<uintptr_t>(ry2) & 7) ||
        (reinterpret_cast<uintptr_t>(my2) & 7))
        return CHIRRUP_E_ALIGN;
    hipLaunchKernelGGL(mm8_reduce_rows_kernel, dim3(B, mm8_row_parts(M_out)), dim3(256), 0, static_cast<hipStream_t>(stream), M_out, splits,
                       (int64_t)B * M_out, partials, static_cast<const f16 *>(rx), static_cast<const f16 *>(mx), S, S_parts, act,
                       static_cast<f16 *>(y), y_stride, static_cast<const f16 *>(ry2), static_cast<const f16 *>(my2),
                       static_cast<f16 *>(xs2), S2);
    return (int)hipGetLastError();
}
