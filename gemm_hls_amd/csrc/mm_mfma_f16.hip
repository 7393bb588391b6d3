// half (Multiply, Add) fast path for gfx950: C[N x M] = A[N x K] . B[K x M], row-major binary16
// in and out, on the f16 matrix-core instructions (v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16).
//
// Numerical contract: products of two binary16 values are exact in f32;
// they are accumulated in f32 by the matrix core and rounded to binary16 ONCE on store.  The
// reference's HLS kernel accumulates in half (and overflows to inf beyond K ~ 2000 on its own
// [1,10) inputs); the k-ordered kernel (MM_PATH_ORDERED) reproduces that behaviour exactly.
//
// Kernels in this file (what MM_PATH_AUTO can dispatch, plus one cross-check):
//   pingpong_16x16x32   default: 256 x 256 tile, ping-pong schedule, A in full-line double slabs, the 16x16x32
//                       instruction (K % 64 == 0, K >= 256, row-major A)
//   pingpong_32x32x16   the same tile / rings / protocol on the 32x32x16 instruction: the independently written
//                       cross-check of the default (f16_variant 100); also what round 2 shipped
//   pingpong_k32        32-deep slabs with 64-byte A rows (K % 32 == 0, K >= 128), row-major and K x N A
//   slab64              one barrier per 64-deep slab (K % 16 == 0; any N; 256 x 256, or 128 x 256 for small problems)
// The schedules and ablations these went through (lock step, early barrier, DMA cache policies, no-DMA / no-read power
// breakdown, the 384 x 256 tile) live in tools/lab/lab_mfma_f16.hip -> tools/lab/libmm_gemm_amd_lab.so.
//
// Common organisation (as mm_mfma_f32.hip: resident output tile, LDS ring fed by global_load_lds):
//   B fragment: B is K x M row-major, the operand wants consecutive k of ONE column, so the
//     LDS image stays row-major [k][256 cols] and the operand is gathered by two
//     ds_read_b64_tr_b16 (hardware 4 x 16 transpose: lane i of a 16-lane group receives column i
//     of the [4 k][16 col] block whose rows the group's lanes point at;
//     out[i][j] = in[4j + (i>>2)][i&3], profiles/r01_probe_ds_read_b64_tr_b16.txt).
//     A B k-row is 512 B, so the 4 rows of a block would share banks; the 16-B chunk index is
//     XORed with (k&3)<<2 on the DMA source side, which spreads the 4 rows over the 4 quadrants
//     of the 256-B bank row: each half-wave then reads 256 distinct bytes.
// Edges: N arbitrary, K % 16 == 0, M % 8 == 0 (reference contract for half: K % 32, M % 32).
#include <cstdlib>
#include <type_traits>

#include "mm_common.h"

namespace mm {
namespace {

using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short s4 __attribute__((vector_size(8)));
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int WM_, int WN_, int TM_, int BK_ = 64, int NS_ = 2>
struct GeoHT {
  static constexpr int WM = WM_, WN = WN_, NS = NS_;
  static constexpr int TM = TM_, TN = 4;
  static constexpr int NW = WM * WN, THREADS = NW * 64;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = BK_;  // 256 x 256 x BK
  static constexpr int CPR = BK * 2 / 16;                              // 16-B chunks per A row
  static constexpr int SWZ_SHIFT = (CPR == 4) ? 2 : 1;
  static_assert(BK == 32 || BK == 64, "BK");
  static constexpr int BROW = BN * 2, BCH = BROW / 16;                 // B k-row bytes / chunks
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BK * BROW;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int EPI_BYTES = NW * TM * 32 * 256;  // epilogue staging: a [TM*32][128] half slice per wave
  static constexpr int LDS_BYTES = NS * STAGE_BYTES > EPI_BYTES ? NS * STAGE_BYTES : EPI_BYTES;
  static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;
  static constexpr int LA = NA / NW, LB = NB / NW;
  static constexpr int KS = BK / 16;                                   // MFMA k-steps per slab
  static_assert(NA % NW == 0 && NB % NW == 0, "DMA split");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(BN == 256, "B swizzle / chunk math assumes 256-column slabs");
};
// slab64: workgroup 256 x 256 (or 128 x 256), wavefronts of 64 x 128 = 2 x 4 accumulators of 32 x 32 (32x32x16 instruction);
// K slab = 64 halves (an A row is 128 B = 8 chunks of 16 B, swizzled with (row>>1)&7); A fragment: ds_read_b128 =
// A[row = l&31][8 consecutive k at 8*(l>>5)] -- exactly the operand; one barrier per slab, fragments double-buffered.
using GeoH = GeoHT<4, 2, 2>;   // 256 x 256, 8 wavefronts of 64 x 128 (2 per SIMD)
using GeoHS = GeoHT<2, 2, 2>;            // 128 x 256, 4 wavefronts of 64 x 128: small / mid-size shapes
using GeoHXS = GeoHT<2, 2, 1>;           // 64 x 256, 4 wavefronts of 32 x 128: below a round of the 128 x 256 tile (round 3)

// LDS-DMA issued from inline asm.  hipcc waits vmcnt(0) before every ds_read_b64_tr_b16 that
// follows a __builtin_amdgcn_global_load_lds (the transpose-read builtin carries no alias
// information, so the pending-DMA hazard check is conservative), which would serialise the ring.
// An asm DMA is invisible to that bookkeeping; its completion is tracked by the kernel's own
// counted s_waitcnt vmcnt(N) + barrier.  M0 (LDS base of the DMA) is saved and restored inside the
// same statement because the compiler owns it.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}

__device__ __forceinline__ h8 join(s4 lo, s4 hi) {
  union { s4 s[2]; h8 h; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.h;
}

template <typename G, bool AT>
__global__ __launch_bounds__(G::THREADS) void mfma_f16_kernel(const _Float16 *__restrict__ A,
                                                                 const _Float16 *__restrict__ B,
                                                                 _Float16 *__restrict__ C, unsigned N, unsigned K,
                                                                 unsigned M, unsigned tiles_n, unsigned tiles_m, unsigned kBand) {
  constexpr int TM = G::TM, TN = G::TN, BK = G::BK, NS = G::NS, CPR = G::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wm = wave / G::WN, wn = wave % G::WN;
  const unsigned lo = lane & 31u, hi = lane >> 5;

  const unsigned nwg = tiles_n * tiles_m;
  const unsigned lin = xcd_remap(blockIdx.x, nwg);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  // ---- DMA sources ---------------------------------------------------------------------------
  size_t a_row_off[G::LA];
  unsigned a_kchunk[G::LA];
#pragma unroll
  for (int i = 0; i < G::LA; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    if (AT) {  // A stored K x N: the slab is [BK][BM] with the same quadrant swizzle as B's
      const unsigned kr = slot / (G::BM / 8), pc = slot % (G::BM / 8);
      a_kchunk[i] = kr;
      a_row_off[i] = min(row0 + (pc ^ ((kr & 3u) << 2)) * 8, N - 8);
    } else {
      const unsigned row = slot / CPR, pc = slot % CPR;
      a_kchunk[i] = pc ^ ((row >> G::SWZ_SHIFT) & (CPR - 1));
      a_row_off[i] = (size_t)min(row0 + row, N - 1) * K;
    }
  }
  unsigned b_krow[G::LB], b_col[G::LB];
#pragma unroll
  for (int i = 0; i < G::LB; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    const unsigned kr = slot / G::BCH, pc = slot % G::BCH;
    const unsigned lc = pc ^ ((kr & 3u) << 2);  // k0 is a multiple of 4, so (k0+kr)&3 == kr&3
    b_krow[i] = kr;
    b_col[i] = min(col0 + lc * 8, M - 8);
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  // part < 0: the whole slab; part 0 / 1: the even / odd DMA instructions of this wave.  In the
  // steady state a slab's refill is issued in two halves one k-step apart: the LDS write traffic
  // of the DMA competes with the fragment reads, and a burst of all 64 KiB right after the barrier
  // costs ~3.5 % (ablation: no refill at all would be +31 %, so this kernel is LDS-port bound).
  auto stage = [&](unsigned buf, unsigned k0, int part = -1) {
    const unsigned base = lds0 + buf * G::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      if (part >= 0 && (i & 1) != part) continue;
      const _Float16 *src = AT ? A + (size_t)min(k0 + a_kchunk[i], K - 1) * N + a_row_off[i]
                               : A + a_row_off[i] + min(k0 + a_kchunk[i] * 8, K - 8);
      dma16(src, base + (wave + G::NW * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < G::LB; ++i) {
      if (part >= 0 && (i & 1) != part) continue;
      const unsigned kr = min(k0 + b_krow[i], K - 1);
      dma16(B + (size_t)kr * M + b_col[i], base + G::A_BYTES + (wave + G::NW * i) * 1024);
    }
  };

  // ---- fragment addresses ----------------------------------------------------------------------
  // A: row = wm*64 + mi*32 + lo, chunk = (2*ks + hi) ^ swz(lo) = (2*ks) ^ (hi ^ swz)
  const unsigned a_swz = hi ^ ((lo >> G::SWZ_SHIFT) & (CPR - 1));
  const unsigned a_frag_base = (wm * TM * 32 + lo) * (BK * 2);
  // B (tr read): x = lane & 15, group column half gq = (lane >> 4) & 1, r = x >> 2 (k row in block)
  //   k = ks*16 + 8*hi + 4*h + r ; logical chunk = wn*16 + ni*4 + 2*gq + ((x&3)>>1) ; +8 B if x odd
  //   physical chunk = logical ^ (r << 2)  ->  ni' = ni ^ r
  const unsigned x = lane & 15u, gq = (lane >> 4) & 1u, r = x >> 2;
  const unsigned b_lane_base = G::A_BYTES + (8 * hi + r) * G::BROW + (wn * 16 + 2 * gq + ((x & 3u) >> 1)) * 16 + (x & 1u) * 8;
  unsigned b_ni_off[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) b_ni_off[ni] = b_lane_base + ((unsigned)ni ^ r) * 64;
  // K x N layout of A: same transpose-read gather as B, over the [k][BM] image (row bytes BM*2)
  unsigned at_mi_off[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
    at_mi_off[mi] = (8 * hi + r) * (G::BM * 2) + (((wm * TM * 4 + mi * 4) ^ (r << 2)) + 2 * gq + ((x & 3u) >> 1)) * 16 + (x & 1u) * 8;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (f32x16)0.0f;

  auto load_frags = [&](unsigned buf, int ks, h8 (&af)[TM], h8 (&bf)[TN]) {
    const char *base = smem + buf * G::STAGE_BYTES;
    const unsigned achunk = ((unsigned)(2 * ks) ^ a_swz) * 16;
    if (AT) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const char *p = base + at_mi_off[mi] + ks * 16 * (G::BM * 2);
        const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)p);
        const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)(p + 4 * G::BM * 2));
        af[mi] = join(v0, v1);
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[mi] = *(const h8 *)(base + a_frag_base + mi * 32 * (BK * 2) + achunk);
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const char *p = base + b_ni_off[ni] + ks * 16 * G::BROW;
      const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)p);
      const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)(p + 4 * G::BROW));
      bf[ni] = join(v0, v1);
    }
  };
  auto mfma_step = [&](const h8 (&af)[TM], const h8 (&bf)[TN]) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
  };

  const unsigned num_tiles = (K + BK - 1) / BK;
  constexpr int L = G::LA + G::LB;
#pragma unroll
  for (int s = 0; s < NS; ++s) stage(s, s * BK);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * L) : "memory");
  __builtin_amdgcn_s_barrier();

  h8 af0[TM], bf0[TN], af1[TM], bf1[TN];
  load_frags(0, 0, af0, bf0);

  const unsigned steady = num_tiles - 1;
  for (unsigned t = 0; t < steady; ++t) {
    const unsigned buf = t % NS;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      h8(&afc)[TM] = (ks & 1) ? af1 : af0;
      h8(&bfc)[TN] = (ks & 1) ? bf1 : bf0;
      h8(&afn)[TM] = (ks & 1) ? af0 : af1;
      h8(&bfn)[TN] = (ks & 1) ? bf0 : bf1;
      if (ks + 1 < G::KS) {
        if (ks == 0 && t > 0) stage((t + NS - 1) % NS, (t + NS - 1) * BK, 1);  // second half of the refill
        load_frags(buf, ks + 1, afn, bfn);
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        stage(buf, (t + NS) * BK, 0);  // first half of the refill of the slot just freed
        load_frags((t + 1) % NS, 0, afn, bfn);
      }
      mfma_step(afc, bfc);
    }
  }
  {
    const unsigned t = num_tiles - 1;
    const int steps = (int)((K - t * BK) / 16);
    for (int ks = 0; ks < steps; ++ks) {
      load_frags(t % NS, ks, af0, bf0);
      mfma_step(af0, bf0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing ring refills (clamped, unread)

  // ---- epilogue: one rounding f32 -> binary16, then through this wave's slice of the (now free)
  // LDS ring so that global stores are 16 B per lane and 256 contiguous bytes per row instead of
  // one half per lane (the MFMA result layout gives a lane ONE column of 16 rows).
  __builtin_amdgcn_s_barrier();  // every wave has finished reading the last slab
  {
    constexpr int ROWS = TM * 32;                       // rows of this wave's tile, 128 columns = 256 B each
    char *slice = smem + wave * (ROWS * 256);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const unsigned row = mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          *(_Float16 *)(slice + row * 256 + (ni * 32 + lo) * 2) = (_Float16)acc[mi][ni][rr];
        }
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < ROWS * 16 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 16, ch = c % 16;
      const u32x4 v = *(const u32x4 *)(slice + row * 256 + ch * 16);
      const unsigned grow = row0 + wm * ROWS + row, gcol = col0 + wn * 128 + ch * 8;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}


// =================================================================================================
// Ping-pong schedule (round 2): the 256 x 256 resident tile, organised so that the matrix pipe
// of every SIMD is fed by ONE of its two waves at a time while the other one does all of its memory
// work.  Why: with both waves of a SIMD in the same phase, each LDS-DMA instruction blocks its wave's
// in-order issue for 60-190 cycles (MI355X_MICROARCH.md: "LDS-DMA piece issue cost"), so right after every
// slab barrier all 8 waves sit in their DMA issue and the matrix pipes idle (ablation: no refill = +31 %).  Here
//   * k-slabs are 32 deep, 4-slab LDS ring (4 x 32 KiB), 3 slabs in flight;
//   * waves 0-3 (one per SIMD) and waves 4-7 (their SIMD partners) run the same code shifted by one
//     barrier: while group X executes the MFMAs of slab u ("compute segment", priority 1),
//     group Y reads its fragments of its next slab from LDS and issues its 4 DMA pieces of a slab
//     three ahead ("load segment"); one s_barrier per segment keeps the two groups in antiphase;
//   * 8 waves as 2 x 4, 128 x 64 per wave; fragments single-buffered (load and compute segments of
//     one wave never overlap -- the overlap comes from the partner wave);
//   * DMA sources are (uniform SGPR base) + (32-bit per-lane offset): the per-lane part never
//     changes, the base advances by one slab per segment with scalar adds: no vector address math
//     in the loop.
// LDS-DMA hand-over rules (cdna_hip_programming.md, 8-phase template): a slab is read one segment
// AFTER the counted vmcnt + barrier that retires it; a buffer is refilled only after a barrier that
// every reader passed with lgkmcnt(0) (replayed on the CPU by tests/test_schedules.py).
//
// pingpong_k32: 32-deep slabs for A and B (K % 32 == 0, the reference's own contract for half: 64-byte bus = 32
// elements, host/RunHardware.cpp:50-55), on the 16x16x32 instruction since round 3 (one slab = one MFMA k; the 32x32x16
// edition is in the lab); row-major A or K x N A (staged and gathered like B).
//   A slab image [256 rows][32 k]: 64-B rows, 16-B chunk index XORed with (-(row>>2))&3;
//   B slab image [32 k][256 cols]: chunk index ^ (k&3)<<2 ^ ((k>>3)&1)<<1.
struct GeoPP {
  static constexpr int BM = 256, BN = 256, BK = 32, NS = 4, THREADS = 512;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BK * BN * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES;  // 128 KiB; the epilogue needs 8 x 128 x 128 B = 128 KiB
  static constexpr int BROW = BN * 2;
};

// one LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to LDS at m0
#define MM_DMA_PIECE(vo, sb, la) "s_mov_b32 m0, " la "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " vo ", " sb "\n\t"

template <bool AT>  // AT: A stored K x N (MM_TRANSPOSED_A): the A slab is staged and gathered exactly like B's
__global__ __launch_bounds__(GeoPP::THREADS) void mfma_f16_pp_kernel(const _Float16 *__restrict__ A,
                                                                       const _Float16 *__restrict__ B,
                                                                       _Float16 *__restrict__ C, unsigned N, unsigned K,
                                                                       unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                       unsigned kBand) {
  using G = GeoPP;
  constexpr int RB = 8, NB = 4;  // 16-row / 16-column blocks of a wavefront's 128 x 64 part (16x16x32 instruction, round 3)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned group = wave >> 2;                 // waves w and w+4 share a SIMD
  const unsigned wq = wave & 3u;
  const unsigned wm = wq >> 1, wn = (wq & 1u) * 2 + group;  // 2 x 4 wave grid; partners sit side by side
  const unsigned l15 = lane & 15u, g = lane >> 4;

  const unsigned nwg = tiles_n * tiles_m;
  const unsigned lin = xcd_remap(blockIdx.x, nwg);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  // ---- DMA: per-lane offsets (constant) and uniform bases (advance with k) ----------------------
  //   row-major A: 16 pieces of 16 rows x 64 B, source chunk = pc ^ (-(row>>2))&3 (the swizzle under which the
  //     16x16 operand read -- row l&15, chunk l>>4 -- covers 16 distinct slots per ds_read_b128 service group);
  //   B, and a K x N A: 16 pieces of 2 k-rows x 512 B, source chunk = pb ^ (k&3)<<2 ^ ((k>>3)&1)<<1 (see pingpong_16x16x32)
  unsigned voff_a[2], voff_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned piece = wave + 8 * i;
    const unsigned row = piece * 16 + lane / 4, pc = lane % 4;
    const unsigned chunk = pc ^ ((0u - (row >> 2)) & 3u);
    const unsigned kr = piece * 2 + lane / 32, pb = lane % 32;
    const unsigned lc = pb ^ ((kr & 3u) << 2) ^ (((kr >> 3) & 1u) << 1);
    voff_a[i] = AT ? kr * N * 2 + (min(row0 + lc * 8, N - 8) - row0) * 2  // K x N: 2 k-rows x 256 tile rows, like B
                   : (min(row0 + row, N - 1) - row0) * K * 2 + chunk * 16;
    voff_b[i] = kr * M * 2 + (min(col0 + lc * 8, M - 8) - col0) * 2;
  }
  const char *a_base = (const char *)A + (AT ? (size_t)row0 * 2 : (size_t)row0 * K * 2);
  const char *b_base = (const char *)B + (size_t)col0 * 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK;
  auto issue = [&](unsigned slab, unsigned buf) {
    const unsigned sl = min(slab, U - 1);               // past the end: harmless re-fetch into a dead buffer
    const char *ap = a_base + (AT ? (size_t)sl * G::BK * N * 2 : (size_t)sl * (G::BK * 2));
    const char *bp = b_base + (size_t)sl * G::BK * M * 2;
    const unsigned la0 = lds0 + buf * G::STAGE_BYTES + wave * 1024, la1 = la0 + 8 * 1024;
    const unsigned lb0 = la0 + G::A_BYTES, lb1 = lb0 + 8 * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" MM_DMA_PIECE("%1", "%5", "%7") MM_DMA_PIECE("%2", "%5", "%8")
                     MM_DMA_PIECE("%3", "%6", "%9") MM_DMA_PIECE("%4", "%6", "%10") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_a[0]), "v"(voff_a[1]), "v"(voff_b[0]), "v"(voff_b[1]), "s"(ap), "s"(bp), "s"(la0), "s"(la1),
                   "s"(lb0), "s"(lb1)
                 : "memory");
  };

  // ---- fragment addresses (per lane, constant) ---------------------------------------------------
  // A (row-major): row = wm*128 + rb*16 + l15, chunk g, physical = g ^ (-(l15>>2))&3
  const unsigned a_off = (wm * 128 + l15) * (G::BK * 2) + (g ^ ((0u - (l15 >> 2)) & 3u)) * 16;
  // B (transpose read): k = 8*g + 4*h2 + r, r = l15>>2; 8-byte piece l15&3 of the block's 32 B (16 columns)
  const unsigned r = l15 >> 2, piece = l15 & 3u, xk = (r << 2) ^ ((g & 1u) << 1);
  unsigned b_off[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    b_off[nb] = G::A_BYTES + (8 * g + r) * G::BROW + ((wn * 8 + nb * 2 + (piece >> 1)) ^ xk) * 16 + (piece & 1u) * 8;
  // K x N A: the same gather over the [k][256 rows] image; a wavefront's 128 rows are chunks wm*16 .. +15, block rb = 2 chunks
  unsigned at_off[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
    at_off[rb] = (8 * g + r) * (G::BM * 2) + ((wm * 16 + rb * 2 + (piece >> 1)) ^ xk) * 16 + (piece & 1u) * 8;

  using f32x4 = __attribute__((ext_vector_type(4))) float;
  f32x4 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4)0.0f;

  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tr2 = [&](const char *p, unsigned row_bytes) {
    const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)p);
    const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)(p + 4 * row_bytes));
    return join(v0, v1);
  };
  // one slab (= one MFMA k): load segment | barrier | compute segment | barrier
  auto phase = [&](auto bufc, unsigned u) {
    constexpr int BUF = decltype(bufc)::value;
    const char *base = smem + BUF * G::STAGE_BYTES;
    h8 af[RB], bf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bf[nb] = tr2(base + b_off[nb], G::BROW);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      if (AT) af[rb] = tr2(base + at_off[rb], G::BM * 2);
      else af[rb] = *(const h8 *)(base + a_off + rb * 16 * (G::BK * 2));
    }
    issue(u + 3, (BUF + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");  // own pieces of slab u+1 landed; fragments in registers
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[rb], bf[nb], acc[rb][nb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    sync();
  };

  // ---- prologue: 3 slabs in flight, slab 0 published -------------------------------------------------
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  sync();
  const bool shifted = group == 1;
  if (shifted) sync();                                   // group 1 runs one segment behind group 0
  for (unsigned u = 0; u < U; u += 4) {
    phase(std::integral_constant<int, 0>{}, u);
    if (u + 1 < U) phase(std::integral_constant<int, 1>{}, u + 1);
    if (u + 2 < U) phase(std::integral_constant<int, 2>{}, u + 2);
    if (u + 3 < U) phase(std::integral_constant<int, 3>{}, u + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing (dead) refills must have landed before LDS is reused
  if (!shifted) sync();                                  // group 0 waits for group 1's last segment
  sync();

  // ---- epilogue: one rounding f32 -> binary16, staged through this wave's 16 KiB slice of the ring
  //      (C/D of the 16x16 form: column l15, rows 4*g + i)
  {
    char *slice = smem + wave * (128 * 128);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *(_Float16 *)(slice + (rb * 16 + 4 * g + i) * 128 + (nb * 16 + l15) * 2) = (_Float16)acc[rb][nb][i];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < 128 * 8 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 8, ch = c % 8;
      const u32x4 v = *(const u32x4 *)(slice + row * 128 + ch * 16);
      const unsigned grow = row0 + wm * 128 + row, gcol = col0 + wn * 64 + ch * 8;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Ping-pong with full-line A requests.  A 32-deep slab gives A rows of 64 bytes, i.e. TWO L2 requests per 128-byte
// line (one per slab); measured: 817 M vs 546 M L2 requests per 16384^3 launch at identical misses, on a kernel
// whose power budget goes into exactly that path.  Here A is staged in DOUBLE slabs
// [256 rows][64 k] (128-byte rows, one request per line, chunk index ^ (row>>1)&7), each double slab
// serving two consecutive segments; B stays in 32-deep slabs.  LDS: 3 A double slabs (96 KiB) +
// 4 B slabs (64 KiB) = all 160 KiB.  A wave still issues 4 DMA pieces per load segment: 2 of A
// (its half of double slab u/2 + 2) and 2 of B (slab u + 3); the counted vmcnt(8) and the barrier
// pairing are unchanged.  Requirements: K % 64 == 0, K >= 256.
// pingpong_32x32x16: this organisation on v_mfma_f32_32x32x16_f16 (4 x 2 accumulators of 32 x 32 per wavefront).
struct GeoPP2 {
  static constexpr int BM = 256, BN = 256, BK = 32, THREADS = 512;
  static constexpr int TM = 4, TN = 2;
  static constexpr int A2_BYTES = BM * 64 * 2, NA = 3;          // A double slab: 32 KiB, ring of 3
  static constexpr int B_BYTES = BK * BN * 2, NB = 4;           // B slab: 16 KiB, ring of 4
  static constexpr int B_REGION = NA * A2_BYTES;
  static constexpr int LDS_BYTES = NA * A2_BYTES + NB * B_BYTES;  // 163840 = the whole LDS of a CU
  static constexpr int BROW = BN * 2;
};

__global__ __launch_bounds__(GeoPP2::THREADS) void mfma_f16_pp2_kernel(const _Float16 *__restrict__ A,
                                                                         const _Float16 *__restrict__ B,
                                                                         _Float16 *__restrict__ C, unsigned N, unsigned K,
                                                                         unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                         unsigned kBand) {
  using G = GeoPP2;
  constexpr int TM = G::TM, TN = G::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned group = wave >> 2, wq = wave & 3u;
  const unsigned wm = wq >> 1, wn = (wq & 1u) * 2 + group;
  const unsigned lo = lane & 31u, hi = lane >> 5;

  const unsigned lin = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  // ---- DMA offsets: A double slab = 32 pieces of 8 rows x 128 B (4 per wave), B slab = 16 pieces of 2 k-rows
  unsigned voff_a[4], voff_b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = (wave + 8 * i) * 8 + lane / 8, pc = lane % 8;
    voff_a[i] = (min(row0 + row, N - 1) - row0) * K * 2 + (pc ^ ((row >> 1) & 7u)) * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned kr = (wave + 8 * i) * 2 + lane / 32, pb = lane % 32;
    const unsigned lc = pb ^ ((kr & 3u) << 2);
    voff_b[i] = kr * M * 2 + (min(col0 + lc * 8, M - 8) - col0) * 2;
  }
  const char *a_base = (const char *)A + (size_t)row0 * K * 2;
  const char *b_base = (const char *)B + (size_t)col0 * 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK, UD = U / 2;
  // half `h` (pieces wave+16h, wave+16h+8) of A double slab `ds` into A buffer `abuf`, and B slab `slab` into B buffer `bbuf`
#define MM_PP2_ISSUE(V0, V1, SB, L0, L1)                                                                   \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3"       \
               "\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" \
               : "=&s"(keep)                                                                               \
               : "v"(V0), "v"(V1), "s"(SB), "s"(L0), "s"(L1)                                               \
               : "memory")
  auto issue_a = [&](unsigned ds, unsigned abuf, int h) {
    const char *ap = a_base + (size_t)min(ds, UD - 1) * 128;
    const unsigned la0 = lds0 + abuf * G::A2_BYTES + (wave + 16 * h) * 1024, la1 = la0 + 8 * 1024;
    const unsigned v0 = h ? voff_a[2] : voff_a[0], v1 = h ? voff_a[3] : voff_a[1];
    unsigned keep;
    MM_PP2_ISSUE(v0, v1, ap, la0, la1);
  };
  auto issue_b = [&](unsigned slab, unsigned bbuf) {
    const char *bp = b_base + (size_t)min(slab, U - 1) * G::BK * M * 2;
    const unsigned lb0 = lds0 + G::B_REGION + bbuf * G::B_BYTES + wave * 1024, lb1 = lb0 + 8 * 1024;
    unsigned keep;
    MM_PP2_ISSUE(voff_b[0], voff_b[1], bp, lb0, lb1);
  };
#undef MM_PP2_ISSUE

  // ---- fragment addresses.  A: row = wm*128 + mi*32 + lo, logical chunk = 4*h + 2*ks + hi (h = slab parity),
  //      physical = logical ^ ((lo>>1)&7) = (4h | 2ks) ^ c with c = hi ^ ((lo>>1)&7)
  const unsigned ca = hi ^ ((lo >> 1) & 7u);
  const unsigned a_row_byte = (wm * 128 + lo) * 128;
  unsigned a_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a_off[j] = a_row_byte + ((unsigned)(2 * j) ^ ca) * 16;  // j = 2h + ks
  const unsigned x = lane & 15u, gq = (lane >> 4) & 1u, r = x >> 2;
  unsigned b_off[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const unsigned logical = wn * 8 + ni * 4 + 2 * gq + ((x & 3u) >> 1);
    b_off[ni] = G::B_REGION + (8 * hi + r) * G::BROW + (logical ^ (r << 2)) * 16 + (x & 1u) * 8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (f32x16)0.0f;

  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one slab u: BUF = u % 4 (B buffer, compile time), abuf = (u/2) % 3 (A buffer of the double slab being read)
  auto phase = [&](auto bufc, unsigned u, unsigned abuf) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr int H = BUF & 1;
    const char *abase = smem + abuf * G::A2_BYTES;
    const char *bbase = smem + BUF * G::B_BYTES;
    h8 af[TM][2], bf[TN][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const char *p = bbase + b_off[ni] + ks * 16 * G::BROW;
        const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)p);
        const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)(p + 4 * G::BROW));
        bf[ni][ks] = join(v0, v1);
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[mi][ks] = *(const h8 *)(abase + a_off[2 * H + ks] + mi * 32 * 128);
    }
    const unsigned abuf_fill = abuf >= 1 ? abuf - 1 : 2;           // (abuf + 2) % 3: the buffer of double slab u/2 - 1
    issue_a(u / 2 + 2, abuf_fill, H);
    issue_b(u + 3, (BUF + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][ks], bf[ni][ks], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    sync();
  };

  // ---- prologue = the issue order of virtual segments -4 .. -1, so that the steady-state vmcnt(8)
  //      ("everything issued two segments ago has landed") holds from the first segment on
  issue_a(0, 0, 0);
  issue_a(0, 0, 1);
  issue_b(0, 0);
  issue_a(1, 1, 0);
  issue_b(1, 1);
  issue_a(1, 1, 1);
  issue_b(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  sync();
  const bool shifted = group == 1;
  if (shifted) sync();
  unsigned ab = 0;  // (u / 2) % 3 at the top of the 4-slab body
  for (unsigned u = 0; u < U; u += 4) {
    const unsigned ab1 = ab == 2 ? 0 : ab + 1;
    phase(std::integral_constant<int, 0>{}, u, ab);
    phase(std::integral_constant<int, 1>{}, u + 1, ab);
    if (u + 2 < U) {                                            // U is even: slabs come in pairs
      phase(std::integral_constant<int, 2>{}, u + 2, ab1);
      phase(std::integral_constant<int, 3>{}, u + 3, ab1);
    }
    ab = ab1 == 2 ? 0 : ab1 + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!shifted) sync();
  sync();

  {
    char *slice = smem + wave * (128 * 128);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const unsigned row = mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          *(_Float16 *)(slice + row * 128 + (ni * 32 + lo) * 2) = (_Float16)acc[mi][ni][rr];
        }
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < 128 * 8 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 8, ch = c % 8;
      const u32x4 v = *(const u32x4 *)(slice + row * 128 + ch * 16);
      const unsigned grow = row0 + wm * 128 + row, gcol = col0 + wn * 64 + ch * 8;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// pingpong_16x16x32 (round 3, the default).  Same tile, LDS rings, DMA and segment
// protocol as pingpong_32x32x16; what changes is the matrix instruction: 16 x 16 outputs x 32 k (4 accumulator
// registers, 16 cycles) instead of 32 x 32 x 16 (16 registers, 32 cycles).  Both run at the same flop rate,
// but on random [1,10) operands the register-only loop of the 16x16x32 form holds 1.98 GHz = 2.06 PF where
// the 32x32x16 form holds 1.68 GHz = 1.76 PF (tools/probes/probe_mfma_power.hip,
// profiles/r03b_probe_mfma_power_by_shape_and_operand_order.txt): the kernel is power-limited, so the
// cheaper instruction is clock for everything else.  A wavefront's 128 x 64 block is 8 x 4 accumulators;
// per 32-deep slab it reads 8 A operands (ds_read_b128: row l&15, k = 8*(l>>4)..+7 -- one slab = one MFMA k)
// and 4 B operands (two ds_read_b64_tr_b16 each: lane group l>>4 gathers k = 8*(l>>4)..+7 of 16 columns):
// the same 16 LDS instructions and bytes as before, now for 32 MFMAs.
//   A image: unchanged ([256 rows][64 k] double slabs, chunk ^ (row>>1)&7): the four 16-lane service groups
//     of a ds_read_b128 still cover 16 distinct 16-B slots (tests/test_layouts.py).
//   B image [32 k][256 cols]: the two 16-lane groups of a half-wave now differ in k by 8 instead of in
//     column by 16, so the chunk index is XORed with ((k>>3)&1)<<1 on top of (k&3)<<2: the 8 k-rows a
//     half-wave touches fall into 8 different 32-byte octants of the 256-byte bank row.
// Accumulation order per output element: k ascending in steps of 32, inside an MFMA the hardware's order;
// results are within the same 1-ulp-of-binary16 bound as the 32x32x16 kernels (not bit-identical to them).
__global__ __launch_bounds__(GeoPP2::THREADS) void mfma_f16_pp2s_kernel(const _Float16 *__restrict__ A,
                                                                          const _Float16 *__restrict__ B,
                                                                          _Float16 *__restrict__ C, unsigned N, unsigned K,
                                                                          unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                          unsigned kBand) {
  using G = GeoPP2;
  constexpr int RB = 8, NB = 4;  // 16-row / 16-column blocks of a wavefront's 128 x 64 part
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned group = wave >> 2, wq = wave & 3u;
  const unsigned wm = wq >> 1, wn = (wq & 1u) * 2 + group;
  const unsigned l15 = lane & 15u, g = lane >> 4;

  const unsigned lin = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  unsigned voff_a[4], voff_b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = (wave + 8 * i) * 8 + lane / 8, pc = lane % 8;
    voff_a[i] = (min(row0 + row, N - 1) - row0) * K * 2 + (pc ^ ((row >> 1) & 7u)) * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned kr = (wave + 8 * i) * 2 + lane / 32, pb = lane % 32;
    const unsigned lc = pb ^ ((kr & 3u) << 2) ^ (((kr >> 3) & 1u) << 1);
    voff_b[i] = kr * M * 2 + (min(col0 + lc * 8, M - 8) - col0) * 2;
  }
  const char *a_base = (const char *)A + (size_t)row0 * K * 2;
  const char *b_base = (const char *)B + (size_t)col0 * 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK, UD = U / 2;
#define MM_PP2_ISSUE(V0, V1, SB, L0, L1)                                                                   \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3"       \
               "\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" \
               : "=&s"(keep)                                                                               \
               : "v"(V0), "v"(V1), "s"(SB), "s"(L0), "s"(L1)                                               \
               : "memory")
  auto issue_a = [&](unsigned ds, unsigned abuf, int h) {
    const char *ap = a_base + (size_t)min(ds, UD - 1) * 128;
    const unsigned la0 = lds0 + abuf * G::A2_BYTES + (wave + 16 * h) * 1024, la1 = la0 + 8 * 1024;
    const unsigned v0 = h ? voff_a[2] : voff_a[0], v1 = h ? voff_a[3] : voff_a[1];
    unsigned keep;
    MM_PP2_ISSUE(v0, v1, ap, la0, la1);
  };
  auto issue_b = [&](unsigned slab, unsigned bbuf) {
    const char *bp = b_base + (size_t)min(slab, U - 1) * G::BK * M * 2;
    const unsigned lb0 = lds0 + G::B_REGION + bbuf * G::B_BYTES + wave * 1024, lb1 = lb0 + 8 * 1024;
    unsigned keep;
    MM_PP2_ISSUE(voff_b[0], voff_b[1], bp, lb0, lb1);
  };
#undef MM_PP2_ISSUE

  // A: row = wm*128 + rb*16 + l15, logical chunk = 4*H + g (H = slab parity inside the double slab), physical = logical ^ (row>>1)&7
  const unsigned a_row_byte = (wm * 128 + l15) * 128;
  const unsigned a_off[2] = {a_row_byte + (g ^ (l15 >> 1)) * 16, a_row_byte + ((4u + g) ^ (l15 >> 1)) * 16};
  // B (transpose read): k = 8*g + 4*h2 + r, r = l15>>2; 8-byte piece l15&3 of the block's 32 B (16 columns)
  const unsigned r = l15 >> 2, piece = l15 & 3u;
  unsigned b_off[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const unsigned logical = wn * 8 + nb * 2 + (piece >> 1);
    b_off[nb] = G::B_REGION + (8 * g + r) * G::BROW + (logical ^ (r << 2) ^ ((g & 1u) << 1)) * 16 + (piece & 1u) * 8;
  }

  using f32x4 = __attribute__((ext_vector_type(4))) float;
  f32x4 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4)0.0f;

  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase = [&](auto bufc, unsigned u, unsigned abuf) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr int H = BUF & 1;
    const char *abase = smem + abuf * G::A2_BYTES;
    const char *bbase = smem + BUF * G::B_BYTES;
    h8 af[RB], bf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const char *p = bbase + b_off[nb];
      const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)p);
      const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lptr_t)(p + 4 * G::BROW));
      bf[nb] = join(v0, v1);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) af[rb] = *(const h8 *)(abase + a_off[H] + rb * 16 * 128);
    const unsigned abuf_fill = abuf >= 1 ? abuf - 1 : 2;
    issue_a(u / 2 + 2, abuf_fill, H);
    issue_b(u + 3, (BUF + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[rb], bf[nb], acc[rb][nb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    sync();
  };

  issue_a(0, 0, 0);
  issue_a(0, 0, 1);
  issue_b(0, 0);
  issue_a(1, 1, 0);
  issue_b(1, 1);
  issue_a(1, 1, 1);
  issue_b(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  sync();
  const bool shifted = group == 1;
  if (shifted) sync();
  unsigned ab = 0;
  for (unsigned u = 0; u < U; u += 4) {
    const unsigned ab1 = ab == 2 ? 0 : ab + 1;
    phase(std::integral_constant<int, 0>{}, u, ab);
    phase(std::integral_constant<int, 1>{}, u + 1, ab);
    if (u + 2 < U) {
      phase(std::integral_constant<int, 2>{}, u + 2, ab1);
      phase(std::integral_constant<int, 3>{}, u + 3, ab1);
    }
    ab = ab1 == 2 ? 0 : ab1 + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!shifted) sync();
  sync();

  {  // epilogue: C/D of the 16x16 form: column l15, rows 4*g + i
    char *slice = smem + wave * (128 * 128);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *(_Float16 *)(slice + (rb * 16 + 4 * g + i) * 128 + (nb * 16 + l15) * 2) = (_Float16)acc[rb][nb][i];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < 128 * 8 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 8, ch = c % 8;
      const u32x4 v = *(const u32x4 *)(slice + row * 128 + ch * 16);
      const unsigned grow = row0 + wm * 128 + row, gcol = col0 + wn * 64 + ch * 8;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}

#undef MM_DMA_PIECE

enum Kind { K_PP16, K_PP32, K_PPK32, K_PPK32_AT, K_SLAB64, K_SLAB64_AT, K_SLAB64_128, K_SLAB64_64, K_NONE };
const char *const kNames[] = {"mfma_f16_256x256_pingpong_16x16x32", "mfma_f16_256x256_pingpong_32x32x16",
                              "mfma_f16_256x256_pingpong_k32", "mfma_f16_256x256_pingpong_k32_KxN",
                              "mfma_f16_256x256x64_slab64", "mfma_f16_256x256x64_slab64_KxN", "mfma_f16_128x256x64_slab64",
                              "mfma_f16_64x256x64_slab64", "unsupported"};

}  // namespace

bool mfma_f16_serves(const Problem &p) {
  if (!(p.n >= 1 && p.m >= 8 && p.k >= 16 && p.m % 8 == 0 && p.k % 16 == 0)) return false;
  return !p.a_transposed || (p.n >= 8 && p.n % 8 == 0);
}

// The DMA addresses a tile's rows with 32-bit byte offsets from a uniform 64-bit base: 256 rows x K x 2 B, and 64 k-rows
// x M (or N, K x N layout) x 2 B, must stay below 4 GiB; longer rows are served by the slab64 kernel (64-bit addresses).
static bool pp_reach(const Problem &p) {
  return 256ull * (p.a_transposed ? 1ull : p.k) * 2ull < (1ull << 32) && 64ull * (p.m > p.n ? p.m : p.n) * 2ull < (1ull << 32);
}
static bool ppk32_serves(const Problem &p) {
  const bool shape = p.k % 32 == 0 && p.k >= 128 && p.m % 8 == 0 && p.m >= 8 && pp_reach(p);
  return p.a_transposed ? shape && p.n % 8 == 0 && p.n >= 8 : shape && p.n >= 1;
}
static bool pp64_serves(const Problem &p) { return !p.a_transposed && ppk32_serves(p) && p.k % 64 == 0 && p.k >= 256; }

int mfma_f16_tile(const Problem &p) {  // 0: 256x256, 4: 128x256, 5: 64x256
  static const TileCandidate cands[] = {{0, 256, 256, 1, 1.00}, {4, 128, 256, 1, 0.80}, {5, 64, 256, 1, 0.60}};
  return p.a_transposed ? 0 : pick_tile(cands, 3, p.n, p.m);
}

// The one place that decides which kernel a (problem, f16_variant knob) pair runs; mm_kernel_name and the launcher
// both go through it.  f16_variant: -1 the best the shape allows; 0 slab64; 4 / 5 slab64 on the 128 x 256 / 64 x 256 tile;
// 11 pingpong_k32; 100 pingpong_32x32x16; 200 pingpong_16x16x32.  A pinned kernel that cannot serve the shape falls
// through to the next one down (as the default does), so a knob never turns a servable problem into an error.
// K x N A (MM_TRANSPOSED_A) of a wide problem under the shape-adaptive pick: transposed into a stream-ordered workspace
// first (mm_transpose.hip), then the row-major default with its bits.  A pinned f16_variant keeps the K x N kernels.
static bool transposes_first(const Problem &p) {
  if (tuning(TUNE_F16_VARIANT) >= 0 || !transposes_first_small(p, 2)) return false;
  Problem q = p;
  q.a_transposed = false;
  return pp64_serves(q) && mfma_f16_tile(q) == 0;
}

static Kind resolve(const Problem &p) {
  if (!mfma_f16_serves(p)) return K_NONE;
  const int v = tuning(TUNE_F16_VARIANT);
  if (!(v < 0 || v == 0 || v == 4 || v == 5 || v == 11 || v == 100 || v == 200)) return K_NONE;   // lab ids are not in this library
  if (transposes_first(p)) return K_PP16;
  if (p.a_transposed) return (v != 0 && ppk32_serves(p)) ? K_PPK32_AT : K_SLAB64_AT;
  if (v == 5 || (v < 0 && mfma_f16_tile(p) == 5)) return K_SLAB64_64;
  if (v == 4 || (v < 0 && mfma_f16_tile(p) == 4)) return K_SLAB64_128;
  if (v == 0) return K_SLAB64;
  if ((v < 0 || v == 200) && pp64_serves(p)) return K_PP16;   // +7 % over pingpong_32x32x16 (profiles/r03c_*)
  if (v == 100 && pp64_serves(p)) return K_PP32;              // round 2's default: +2-4 % over pingpong_k32 (profiles/r02h_*)
  if (ppk32_serves(p)) return K_PPK32;
  return K_SLAB64;
}

const char *mfma_f16_name(const Problem &p) { return kNames[resolve(p)]; }

template <typename Kern>
static int launch_tile(hipStream_t s, const Problem &p, Kern kern, unsigned bm, unsigned bn, unsigned threads, int lds,
                       unsigned long long &configured) {
  const unsigned tiles_n = (p.n + bm - 1) / bm, tiles_m = (p.m + bn - 1) / bn;
  if (int e = ensure_dynamic_lds((const void *)kern, lds, configured)) return e;
  hipLaunchKernelGGL(kern, dim3(tiles_n * tiles_m), dim3(threads), lds, s, (const _Float16 *)p.a, (const _Float16 *)p.b,
                     (_Float16 *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, band_rows(bm, bn, 1));
  return (int)hipGetLastError();
}

// A K x N A served where it lies (no workspace): the ping-pong K x N kernel where its shape rules allow, else slab64's
static int launch_kxn_in_place(hipStream_t s, const Problem &p, unsigned long long (&cfg)[K_NONE]) {
  if (tuning(TUNE_F16_VARIANT) != 0 && ppk32_serves(p))
    return launch_tile(s, p, mfma_f16_pp_kernel<true>, 256, 256, GeoPP::THREADS, GeoPP::LDS_BYTES, cfg[K_PPK32_AT]);
  return launch_tile(s, p, mfma_f16_kernel<GeoH, true>, GeoH::BM, GeoH::BN, GeoH::THREADS, GeoH::LDS_BYTES, cfg[K_SLAB64_AT]);
}

int launch_mfma_f16(hipStream_t s, const Problem &p) {
  static unsigned long long cfg[K_NONE] = {};
  if (transposes_first(p)) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipMemPool_t pool = nullptr;
    if (int rc = workspace_pool(dev, &pool)) return rc;
    void *an = nullptr;
    if ((e = hipMallocFromPoolAsync(&an, (size_t)p.n * p.k * 2, pool, s)) != hipSuccess) {
      // no room for the N x K copy (up to 4 GiB next to a nearly full device): not an error -- the K x N kernels serve the
      // call without extra memory, as they did before the pre-pass existed; same contract, ~2-3 % slower
      (void)hipGetLastError();
      return launch_kxn_in_place(s, p, cfg);
    }
    int rc = launch_transpose_kxn(s, p.a, an, p.k, p.n, 2);
    if (rc == 0) {
      Problem q = p;
      q.a = an;
      q.a_transposed = false;
      rc = launch_mfma_f16(s, q);
    }
    const hipError_t f = hipFreeAsync(an, s);
    return rc ? rc : (int)f;
  }
  const Kind k = resolve(p);
  switch (k) {
    case K_PP16: return launch_tile(s, p, mfma_f16_pp2s_kernel, 256, 256, GeoPP2::THREADS, GeoPP2::LDS_BYTES, cfg[k]);
    case K_PP32: return launch_tile(s, p, mfma_f16_pp2_kernel, 256, 256, GeoPP2::THREADS, GeoPP2::LDS_BYTES, cfg[k]);
    case K_PPK32: return launch_tile(s, p, mfma_f16_pp_kernel<false>, 256, 256, GeoPP::THREADS, GeoPP::LDS_BYTES, cfg[k]);
    case K_PPK32_AT: return launch_tile(s, p, mfma_f16_pp_kernel<true>, 256, 256, GeoPP::THREADS, GeoPP::LDS_BYTES, cfg[k]);
    case K_SLAB64: return launch_tile(s, p, mfma_f16_kernel<GeoH, false>, GeoH::BM, GeoH::BN, GeoH::THREADS, GeoH::LDS_BYTES, cfg[k]);
    case K_SLAB64_AT: return launch_tile(s, p, mfma_f16_kernel<GeoH, true>, GeoH::BM, GeoH::BN, GeoH::THREADS, GeoH::LDS_BYTES, cfg[k]);
    case K_SLAB64_128: return launch_tile(s, p, mfma_f16_kernel<GeoHS, false>, GeoHS::BM, GeoHS::BN, GeoHS::THREADS, GeoHS::LDS_BYTES, cfg[k]);
    case K_SLAB64_64: return launch_tile(s, p, mfma_f16_kernel<GeoHXS, false>, GeoHXS::BM, GeoHXS::BN, GeoHXS::THREADS, GeoHXS::LDS_BYTES, cfg[k]);
    default: return kErrNotSupported;
  }
}

}  // namespace mm
