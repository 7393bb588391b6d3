// int8_t / uint8_t (Multiply, Add) fast path for gfx950 on the int8 matrix-core instructions
// (v_mfma_i32_16x16x64_i8 / v_mfma_i32_32x32x32_i8).
//
// Why the signed-int8 matrix core serves BOTH element types exactly: the reference's semiring on
// an 8-bit Data_t wraps every product and every sum to 8 bits (hlslib::op::Multiply/Add return
// Data_t), i.e. the result is sum_k a*b taken mod 2^8.  uint8 and int8 bit patterns are congruent
// mod 2^8, products and sums of congruent numbers stay congruent, and the i32 accumulator wraps mod
// 2^32 (a multiple of 2^8), so (int8)(i32 accumulator) is bit-identical to Naive
// (include/Utility.h:18-42) for int8_t and for uint8_t (the type the reference special-cases at
// CMakeLists.txt:46-47).  Checked against the oracle in tests/test_gpu_parity.py.
//
// Kernels in this file, as in mm_mfma_f16.hip: pingpong_16x16x64 (default: K % 128 == 0, K >= 512, row-major A),
// pingpong_32x32x32 (cross-check, i8_variant 100), pingpong_k64 (K % 64 == 0; row-major and K x N A), slab128 (K % 32 == 0).
// Lock-step ablations: tools/lab/lab_mfma_i8.hip.
// Organisation of slab128 as mm_mfma_f16.hip's slab64: 256 x 256 x 128(bytes) slabs, 8 wavefronts of 64 x 128, A operand
// by one ds_read_b128 (16 consecutive k of a row, rows swizzled with (row>>1)&7), B operand (16
// consecutive k of ONE column of the row-major B) by two ds_read_b64_tr_b8: lane i of a 16-lane
// group receives column i of the [8 k][16 col] block whose rows the group's lanes point at,
// out[i][j] = in[2j + (i>>3)][i&7] (profiles/r01_probe_ds_read_b64_tr_b8_and_mfma_i8.txt).  A B
// k-row is 256 B = one bank row, so the 16-B chunk index is XORed with (k&7)<<1 on the DMA source
// side: the 8 rows of a block then sit in 8 different chunk pairs and a half-wave reads 256
// distinct bytes.  Operand layout of the MFMA (same probe): lane l, byte b <-> k = 16*(l>>5) + b.
// Edges: N arbitrary, K % 32 == 0, M % 16 == 0 (reference contract for 1-byte types: K % 64,
// M % 64); a K x N A (N % 16 == 0) is gathered like B; other shapes go to the predicated kernels.
#include <cstdlib>
#include <type_traits>

#include "mm_common.h"

namespace mm {
namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
typedef int v2i __attribute__((vector_size(8)));
typedef __attribute__((address_space(3))) void *lptr_t;

template <int WM_, int WN_, int TM_>
struct GeoI8T {
  static constexpr int WM = WM_, WN = WN_, NS = 2, TM = TM_, TN = 4;
  static constexpr int NW = WM * WN, THREADS = NW * 64;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 128;   // BK in elements == bytes
  static_assert(BN == 256, "B swizzle / chunk math assumes 256-column slabs");
  static constexpr int CPR = 8;                                // 16-B chunks per A row
  static constexpr int BROW = BN, BCH = BROW / 16;             // B k-row bytes / chunks
  static constexpr int A_BYTES = BM * BK, B_BYTES = BK * BROW;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES, LDS_BYTES = NS * STAGE_BYTES;
  static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;
  static constexpr int LA = NA / NW, LB = NB / NW;
  static constexpr int KS = BK / 32;                           // MFMA k-steps per slab
};
using GeoI8 = GeoI8T<4, 2, 2>;    // 256 x 256, 8 wavefronts of 64 x 128
using GeoI8S = GeoI8T<2, 2, 1>;   // 64 x 256, 4 wavefronts of 32 x 128: problems below a round of the 256 x 256 tile (round 3; row-major A only)

// asm LDS-DMA (see mm_mfma_f16.hip: transpose-read builtins make hipcc drain builtin DMAs)
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

__device__ __forceinline__ i32x4 join(v2i lo, v2i hi) {
  i32x4 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
  return r;
}

template <typename G, bool AT>
__global__ __launch_bounds__(G::THREADS) void mfma_i8_kernel(const signed char *__restrict__ A,
                                                                 const signed char *__restrict__ B,
                                                                 signed char *__restrict__ C, unsigned N, unsigned K,
                                                                 unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                 unsigned kBand) {
  constexpr int TM = G::TM, TN = G::TN, BK = G::BK, NS = G::NS, CPR = G::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wm = wave / G::WN, wn = wave % G::WN;
  const unsigned lo = lane & 31u, hi = lane >> 5;

  const unsigned lin = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  // ---- DMA sources ---------------------------------------------------------------------------
  size_t a_row_off[G::LA];
  unsigned a_kchunk[G::LA];
#pragma unroll
  for (int i = 0; i < G::LA; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    if (AT) {  // A stored K x N: slab [BK][BM] bytes with the same chunk swizzle as B's
      const unsigned kr = slot / (G::BM / 16), pc = slot % (G::BM / 16);
      a_kchunk[i] = kr;
      a_row_off[i] = min(row0 + (pc ^ ((kr & 7u) << 1)) * 16, N - 16);
    } else {
      const unsigned row = slot / CPR, pc = slot % CPR;
      a_kchunk[i] = pc ^ ((row >> 1) & (CPR - 1));
      a_row_off[i] = (size_t)min(row0 + row, N - 1) * K;
    }
  }
  unsigned b_krow[G::LB], b_col[G::LB];
#pragma unroll
  for (int i = 0; i < G::LB; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    const unsigned kr = slot / G::BCH, pc = slot % G::BCH;
    b_krow[i] = kr;
    b_col[i] = min(col0 + (pc ^ ((kr & 7u) << 1)) * 16, M - 16);  // k0 % 8 == 0, so (k0+kr)&7 == kr&7
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  // part < 0: whole slab; part 0 / 1: even / odd DMA instructions (refill issued in two halves one
  // k-step apart, see mm_mfma_f16.hip)
  auto stage = [&](unsigned buf, unsigned k0, int part = -1) {
    const unsigned base = lds0 + buf * G::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      if (part >= 0 && (i & 1) != part) continue;
      dma16(AT ? A + (size_t)min(k0 + a_kchunk[i], K - 1) * N + a_row_off[i]
               : A + a_row_off[i] + min(k0 + a_kchunk[i] * 16, K - 16),
            base + (wave + G::NW * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < G::LB; ++i) {
      if (part >= 0 && (i & 1) != part) continue;
      dma16(B + (size_t)min(k0 + b_krow[i], K - 1) * M + b_col[i], base + G::A_BYTES + (wave + G::NW * i) * 1024);
    }
  };

  // ---- fragment addresses ----------------------------------------------------------------------
  const unsigned a_swz = hi ^ ((lo >> 1) & (CPR - 1));
  const unsigned a_frag_base = (wm * TM * 32 + lo) * BK;
  // B: source-lane role y = lane & 15 -> block row r = y >> 1, 8-byte half q = y & 1; gq = 16-col half
  const unsigned y = lane & 15u, gq = (lane >> 4) & 1u, r = y >> 1, q = y & 1u;
  const unsigned b_lane_base = G::A_BYTES + (16 * hi + r) * G::BROW + 8 * q;
  unsigned b_ni_off[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) b_ni_off[ni] = b_lane_base + ((((wn * 4 + ni) ^ r) * 2) + gq) * 16;
  // K x N layout of A: the same transpose-read gather over the [k][BM] image (256-byte k-rows)
  unsigned at_mi_off[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
    at_mi_off[mi] = (16 * hi + r) * G::BM + 8 * q + ((((wm * TM + mi) ^ r) * 2) + gq) * 16;

  i32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (i32x16)0;

  auto load_frags = [&](unsigned buf, int ks, i32x4 (&af)[TM], i32x4 (&bf)[TN]) {
    const char *base = smem + buf * G::STAGE_BYTES;
    const unsigned achunk = ((unsigned)(2 * ks) ^ a_swz) * 16;
    if (AT) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const char *p = base + at_mi_off[mi] + ks * 32 * G::BM;
        const v2i v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)p);
        const v2i v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)(p + 8 * G::BM));
        af[mi] = join(v0, v1);
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[mi] = *(const i32x4 *)(base + a_frag_base + mi * 32 * BK + achunk);
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const char *p = base + b_ni_off[ni] + ks * 32 * G::BROW;
      const v2i v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)p);
      const v2i v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)(p + 8 * G::BROW));
      bf[ni] = join(v0, v1);
    }
  };
  auto mfma_step = [&](const i32x4 (&af)[TM], const i32x4 (&bf)[TN]) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
  };

  const unsigned num_tiles = (K + BK - 1) / BK;
  constexpr int L = G::LA + G::LB;
#pragma unroll
  for (int s = 0; s < NS; ++s) stage(s, s * BK);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * L) : "memory");
  __builtin_amdgcn_s_barrier();

  i32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
  load_frags(0, 0, af0, bf0);
  const unsigned steady = num_tiles - 1;
  for (unsigned t = 0; t < steady; ++t) {
    const unsigned buf = t % NS;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      i32x4(&afc)[TM] = (ks & 1) ? af1 : af0;
      i32x4(&bfc)[TN] = (ks & 1) ? bf1 : bf0;
      i32x4(&afn)[TM] = (ks & 1) ? af0 : af1;
      i32x4(&bfn)[TN] = (ks & 1) ? bf0 : bf1;
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
    const int steps = (int)((K - t * BK) / 32);
    for (int ks = 0; ks < steps; ++ks) {
      load_frags(t % NS, ks, af0, bf0);
      mfma_step(af0, bf0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: low 8 bits of the i32 sums, through the wave's LDS slice, 16-B global stores ----
  __builtin_amdgcn_s_barrier();
  {
    constexpr int ROWS = TM * 32;
    char *slice = smem + wave * (ROWS * 128);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const unsigned row = mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          slice[row * 128 + ni * 32 + lo] = (char)acc[mi][ni][rr];
        }
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < ROWS * 8 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 8, ch = c % 8;
      const u32x4 v = *(const u32x4 *)(slice + row * 128 + ch * 16);
      const unsigned grow = row0 + wm * ROWS + row, gcol = col0 + wn * 128 + ch * 16;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}


// =================================================================================================
// Ping-pong schedule (round 2), the int8 twin of mfma_f16_pp_kernel (mm_mfma_f16.hip -- read the
// comment there): 64-byte-deep slabs (2 MFMA k-steps), 4-slab LDS ring of 32 KiB, waves 0-3 and 4-7
// one barrier apart so that one wave of every SIMD multiplies while its partner reads fragments and
// issues DMA; 8 waves as 2 x 4, 128 x 64 outputs per wave.  The A slab image is byte-for-byte the
// f16 kernel's ([256 rows][64 B], chunk ^ (row>>2)&3, fragment = chunk 2*ks + hi); the B slab is
// [64 k][256 B] with the transpose-read layout of the kernel above (chunk ^ (k&7)<<1).
// Requirements: K % 64 == 0 (the reference's contract for 1-byte types), M % 16 == 0, row-major A.
struct GeoI8PP {
  static constexpr int BM = 256, BN = 256, BK = 64, NS = 4, THREADS = 512;
  static constexpr int A_BYTES = BM * BK, B_BYTES = BK * BN, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES;
  static constexpr int BROW = BN;
};
#define MM_DMA_PIECE(vo, sb, la) "s_mov_b32 m0, " la "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " vo ", " sb "\n\t"

template <bool AT>  // AT: A stored K x N, staged and gathered like B
__global__ __launch_bounds__(GeoI8PP::THREADS) void mfma_i8_pp_kernel(const signed char *__restrict__ A,
                                                                        const signed char *__restrict__ B,
                                                                        signed char *__restrict__ C, unsigned N, unsigned K,
                                                                        unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                        unsigned kBand) {
  // pingpong_k64 on v_mfma_i32_16x16x64_i8 since round 3 (one 64-deep slab = one MFMA k; the 32x32x32 edition is in the lab)
  using G = GeoI8PP;
  constexpr int RB = 8, NB = 4;
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

  // DMA.  Row-major A: 16 pieces of 16 rows x 64 B, source chunk = pc ^ (-(row>>2))&3; B and a K x N A: 16 pieces of
  // 4 k-rows x 256 B, source chunk = pb ^ (((k&7)<<1) | ((k>>4)&1))  (see pingpong_16x16x64)
  unsigned voff_a[2], voff_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned piece = wave + 8 * i;
    const unsigned row = piece * 16 + lane / 4, pc = lane % 4;
    const unsigned kr = piece * 4 + lane / 16, pb = lane % 16;
    const unsigned lc = pb ^ (((kr & 7u) << 1) | ((kr >> 4) & 1u));
    voff_a[i] = AT ? kr * N + (min(row0 + lc * 16, N - 16) - row0)
                   : (min(row0 + row, N - 1) - row0) * K + (pc ^ ((0u - (row >> 2)) & 3u)) * 16;
    voff_b[i] = kr * M + (min(col0 + lc * 16, M - 16) - col0);
  }
  const char *a_base = (const char *)A + (AT ? (size_t)row0 : (size_t)row0 * K);
  const char *b_base = (const char *)B + col0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK;
  auto issue = [&](unsigned slab, unsigned buf) {
    const unsigned sl = min(slab, U - 1);
    const char *ap = a_base + (AT ? (size_t)sl * G::BK * N : (size_t)sl * G::BK);
    const char *bp = b_base + (size_t)sl * G::BK * M;
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

  // A (row-major): row = wm*128 + rb*16 + l15, chunk g (16 k bytes), physical = g ^ (-(l15>>2))&3
  const unsigned a_off = (wm * 128 + l15) * G::BK + (g ^ ((0u - (l15 >> 2)) & 3u)) * 16;
  // B (8-bit transpose read): block row r = l15>>1 (k = 16*g + 8*h2 + r), 8-byte half q = l15&1 of the 16 columns
  const unsigned r = l15 >> 1, q = l15 & 1u, xk = (r << 1) | (g & 1u);
  unsigned b_off[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b_off[nb] = G::A_BYTES + (16 * g + r) * G::BROW + 8 * q + ((wn * 4 + nb) ^ xk) * 16;
  unsigned at_off[RB];  // K x N A: the same gather over the [k][256 rows] image (this wavefront's 16-row chunks wm*8 .. +7)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) at_off[rb] = (16 * g + r) * G::BM + 8 * q + ((wm * 8 + rb) ^ xk) * 16;

  i32x4 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (i32x4)0;

  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tr2 = [&](const char *p, unsigned row_bytes) {
    const v2i v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)p);
    const v2i v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)(p + 8 * row_bytes));
    return join(v0, v1);
  };
  auto phase = [&](auto bufc, unsigned u) {
    constexpr int BUF = decltype(bufc)::value;
    const char *base = smem + BUF * G::STAGE_BYTES;
    i32x4 af[RB], bf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bf[nb] = tr2(base + b_off[nb], G::BROW);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      if (AT) af[rb] = tr2(base + at_off[rb], G::BM);
      else af[rb] = *(const i32x4 *)(base + a_off + rb * 16 * G::BK);
    }
    issue(u + 3, (BUF + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[rb][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rb], bf[nb], acc[rb][nb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    sync();
  };

  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  sync();
  const bool shifted = group == 1;
  if (shifted) sync();
  for (unsigned u = 0; u < U; u += 4) {
    phase(std::integral_constant<int, 0>{}, u);
    if (u + 1 < U) phase(std::integral_constant<int, 1>{}, u + 1);
    if (u + 2 < U) phase(std::integral_constant<int, 2>{}, u + 2);
    if (u + 3 < U) phase(std::integral_constant<int, 3>{}, u + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!shifted) sync();
  sync();

  // epilogue: low 8 bits of the i32 sums through this wave's 8 KiB slice, 16-B global stores (C/D: column l15, rows 4*g + i)
  {
    char *slice = smem + wave * (128 * 64);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) slice[(rb * 16 + 4 * g + i) * 64 + nb * 16 + l15] = (char)acc[rb][nb][i];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < 128 * 4 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 4, ch = c % 4;
      const u32x4 v = *(const u32x4 *)(slice + row * 64 + ch * 16);
      const unsigned grow = row0 + wm * 128 + row, gcol = col0 + wn * 64 + ch * 16;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}

// Ping-pong with full-line A requests (see mfma_f16_pp2_kernel in mm_mfma_f16.hip): A staged in
// double slabs [256 rows][128 B] (ring of 3 x 32 KiB, chunk ^ (row>>1)&7), B in 64-deep slabs (ring of
// 4 x 16 KiB); all 160 KiB of LDS.  Requirements: K % 128 == 0, K >= 512.
struct GeoI8PP2 {
  static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
  static constexpr int TM = 4, TN = 2;
  static constexpr int A2_BYTES = BM * 128, NA = 3, B_BYTES = BK * BN, NB = 4;
  static constexpr int B_REGION = NA * A2_BYTES, LDS_BYTES = NA * A2_BYTES + NB * B_BYTES;
  static constexpr int BROW = BN;
};

__global__ __launch_bounds__(GeoI8PP2::THREADS) void mfma_i8_pp2_kernel(const signed char *__restrict__ A,
                                                                         const signed char *__restrict__ B,
                                                                         signed char *__restrict__ C, unsigned N, unsigned K,
                                                                         unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                         unsigned kBand) {
  using G = GeoI8PP2;
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

  unsigned voff_a[4], voff_b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = (wave + 8 * i) * 8 + lane / 8, pc = lane % 8;      // 32 A pieces of 8 rows x 128 B
    voff_a[i] = (min(row0 + row, N - 1) - row0) * K + (pc ^ ((row >> 1) & 7u)) * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned kr = (wave + 8 * i) * 4 + lane / 16, pb = lane % 16;     // 16 B pieces of 4 k-rows x 256 B
    const unsigned lc = pb ^ ((kr & 7u) << 1);
    voff_b[i] = kr * M + (min(col0 + lc * 16, M - 16) - col0);
  }
  const char *a_base = (const char *)A + (size_t)row0 * K;
  const char *b_base = (const char *)B + col0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK, UD = U / 2;
  auto issue_a = [&](unsigned ds, unsigned abuf, int h) {
    const char *ap = a_base + (size_t)min(ds, UD - 1) * 128;
    const unsigned la0 = lds0 + abuf * G::A2_BYTES + (wave + 16 * h) * 1024, la1 = la0 + 8 * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" MM_DMA_PIECE("%1", "%3", "%4") MM_DMA_PIECE("%2", "%3", "%5") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(h ? voff_a[2] : voff_a[0]), "v"(h ? voff_a[3] : voff_a[1]), "s"(ap), "s"(la0), "s"(la1)
                 : "memory");
  };
  auto issue_b = [&](unsigned slab, unsigned bbuf) {
    const char *bp = b_base + (size_t)min(slab, U - 1) * G::BK * M;
    const unsigned lb0 = lds0 + G::B_REGION + bbuf * G::B_BYTES + wave * 1024, lb1 = lb0 + 8 * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" MM_DMA_PIECE("%1", "%3", "%4") MM_DMA_PIECE("%2", "%3", "%5") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_b[0]), "v"(voff_b[1]), "s"(bp), "s"(lb0), "s"(lb1)
                 : "memory");
  };

  const unsigned ca = hi ^ ((lo >> 1) & 7u);
  const unsigned a_row_byte = (wm * 128 + lo) * 128;
  unsigned a_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a_off[j] = a_row_byte + ((unsigned)(2 * j) ^ ca) * 16;  // j = 2*(slab parity) + ks
  const unsigned y = lane & 15u, gq = (lane >> 4) & 1u, r = y >> 1, q = y & 1u;
  unsigned b_off[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
    b_off[ni] = G::B_REGION + (16 * hi + r) * G::BROW + 8 * q + ((((wn * 2 + ni) ^ r) * 2) + gq) * 16;

  i32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (i32x16)0;

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
    i32x4 af[TM][2], bf[TN][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const char *p = bbase + b_off[ni] + ks * 32 * G::BROW;
        const v2i v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)p);
        const v2i v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)(p + 8 * G::BROW));
        bf[ni][ks] = join(v0, v1);
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[mi][ks] = *(const i32x4 *)(abase + a_off[2 * H + ks] + mi * 32 * 128);
    }
    issue_a(u / 2 + 2, abuf >= 1 ? abuf - 1 : 2, H);
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
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mi][ks], bf[ni][ks], acc[mi][ni], 0, 0, 0);
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

  {
    char *slice = smem + wave * (128 * 64);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const unsigned row = mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
          slice[row * 64 + ni * 32 + lo] = (char)acc[mi][ni][rr];
        }
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < 128 * 4 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 4, ch = c % 4;
      const u32x4 v = *(const u32x4 *)(slice + row * 64 + ch * 16);
      const unsigned grow = row0 + wm * 128 + row, gcol = col0 + wn * 64 + ch * 16;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}

// pingpong_16x16x64 (round 3, the default): same tile, rings, DMA and segment protocol as
// pingpong_32x32x32, the matrix instruction in its 16 x 16 x 64 form (4 accumulator registers, 16 cycles) -- on full-range random
// bytes the register-only loop of this form holds 1.97 GHz = 4.09 POp/s where the 32x32x32 form holds 1.67 GHz =
// 3.50 POp/s (profiles/r03b_probe_mfma_power_by_shape_and_operand_order.txt), and the kernel is power-limited.
// A wavefront's 128 x 64 block is 8 x 4 accumulators; a 64-deep slab is ONE MFMA k: 8 A operands (ds_read_b128:
// row l&15, k = 16*(l>>4)..+15) and 4 B operands (two ds_read_b64_tr_b8 each: lane group l>>4 gathers
// k = 16*(l>>4)..+15 of 16 columns).  B image [64 k][256 cols]: a half-wave's two lane groups differ in k by 16
// instead of in column by 16, so the chunk index is XORed with ((k&7)<<1) | ((k>>4)&1): the 16 k-rows a half-wave
// touches fall into the 16 different chunks of the 256-byte bank row.  Integer sums: bit-identical to every
// other schedule and to Naive.
__global__ __launch_bounds__(GeoI8PP2::THREADS) void mfma_i8_pp2s_kernel(const signed char *__restrict__ A,
                                                                          const signed char *__restrict__ B,
                                                                          signed char *__restrict__ C, unsigned N, unsigned K,
                                                                          unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                          unsigned kBand) {
  using G = GeoI8PP2;
  constexpr int RB = 8, NB = 4;
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
    voff_a[i] = (min(row0 + row, N - 1) - row0) * K + (pc ^ ((row >> 1) & 7u)) * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned kr = (wave + 8 * i) * 4 + lane / 16, pb = lane % 16;
    const unsigned lc = pb ^ (((kr & 7u) << 1) | ((kr >> 4) & 1u));
    voff_b[i] = kr * M + (min(col0 + lc * 16, M - 16) - col0);
  }
  const char *a_base = (const char *)A + (size_t)row0 * K;
  const char *b_base = (const char *)B + col0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK, UD = U / 2;
  auto issue_a = [&](unsigned ds, unsigned abuf, int h) {
    const char *ap = a_base + (size_t)min(ds, UD - 1) * 128;
    const unsigned la0 = lds0 + abuf * G::A2_BYTES + (wave + 16 * h) * 1024, la1 = la0 + 8 * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" MM_DMA_PIECE("%1", "%3", "%4") MM_DMA_PIECE("%2", "%3", "%5") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(h ? voff_a[2] : voff_a[0]), "v"(h ? voff_a[3] : voff_a[1]), "s"(ap), "s"(la0), "s"(la1)
                 : "memory");
  };
  auto issue_b = [&](unsigned slab, unsigned bbuf) {
    const char *bp = b_base + (size_t)min(slab, U - 1) * G::BK * M;
    const unsigned lb0 = lds0 + G::B_REGION + bbuf * G::B_BYTES + wave * 1024, lb1 = lb0 + 8 * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" MM_DMA_PIECE("%1", "%3", "%4") MM_DMA_PIECE("%2", "%3", "%5") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_b[0]), "v"(voff_b[1]), "s"(bp), "s"(lb0), "s"(lb1)
                 : "memory");
  };

  // A: row = wm*128 + rb*16 + l15, logical chunk 4*H + g, physical = logical ^ (row>>1)&7
  const unsigned a_row_byte = (wm * 128 + l15) * 128;
  const unsigned a_off[2] = {a_row_byte + (g ^ (l15 >> 1)) * 16, a_row_byte + ((4u + g) ^ (l15 >> 1)) * 16};
  // B (8-bit transpose read): block row r = l15>>1 (k = 16*g + 8*h2 + r), 8-byte half q = l15&1 of the 16 columns
  const unsigned r = l15 >> 1, q = l15 & 1u;
  unsigned b_off[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    b_off[nb] = G::B_REGION + (16 * g + r) * G::BROW + 8 * q + ((wn * 4 + nb) ^ ((r << 1) | (g & 1u))) * 16;

  i32x4 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (i32x4)0;

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
    i32x4 af[RB], bf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const char *p = bbase + b_off[nb];
      const v2i v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)p);
      const v2i v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lptr_t)(p + 8 * G::BROW));
      bf[nb] = join(v0, v1);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) af[rb] = *(const i32x4 *)(abase + a_off[H] + rb * 16 * 128);
    issue_a(u / 2 + 2, abuf >= 1 ? abuf - 1 : 2, H);
    issue_b(u + 3, (BUF + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[rb][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rb], bf[nb], acc[rb][nb], 0, 0, 0);
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

  {  // C/D of the 16x16 form: column l15, rows 4*g + i
    char *slice = smem + wave * (128 * 64);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) slice[(rb * 16 + 4 * g + i) * 64 + nb * 16 + l15] = (char)acc[rb][nb][i];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int it = 0; it < 128 * 4 / 64; ++it) {
      const unsigned c = it * 64 + lane, row = c / 4, ch = c % 4;
      const u32x4 v = *(const u32x4 *)(slice + row * 64 + ch * 16);
      const unsigned grow = row0 + wm * 128 + row, gcol = col0 + wn * 64 + ch * 16;
      if (grow < N && gcol < M) *(u32x4 *)(C + (size_t)grow * M + gcol) = v;
    }
  }
}
#undef MM_DMA_PIECE

enum Kind { K_PP16, K_PP32, K_PPK64, K_PPK64_AT, K_SLAB128, K_SLAB128_AT, K_SLAB128_64, K_NONE };
const char *const kNames[] = {"mfma_i8_256x256_pingpong_16x16x64", "mfma_i8_256x256_pingpong_32x32x32", "mfma_i8_256x256_pingpong_k64",
                              "mfma_i8_256x256_pingpong_k64_KxN", "mfma_i8_256x256x128_slab128", "mfma_i8_256x256x128_slab128_KxN",
                              "mfma_i8_64x256x128_slab128", "unsupported"};

}  // namespace

bool mfma_i8_serves(const Problem &p) {
  if (!(p.n >= 1 && p.m >= 16 && p.k >= 32 && p.m % 16 == 0 && p.k % 32 == 0)) return false;
  return !p.a_transposed || (p.n >= 16 && p.n % 16 == 0);
}

// 32-bit byte offsets inside a tile's rows (see mm_mfma_f16.hip): 256 rows x K B and 128 k-rows x max(M, N) B below 4 GiB
static bool pp_reach(const Problem &p) {
  return 256ull * (p.a_transposed ? 1ull : p.k) < (1ull << 32) && 128ull * (p.m > p.n ? p.m : p.n) < (1ull << 32);
}
static bool ppk64_serves(const Problem &p) {
  const bool shape = p.k % 64 == 0 && p.k >= 256 && p.m % 16 == 0 && p.m >= 16 && pp_reach(p);
  return p.a_transposed ? shape && p.n % 16 == 0 && p.n >= 16 : shape && p.n >= 1;
}
static bool pp128_serves(const Problem &p) { return !p.a_transposed && ppk64_serves(p) && p.k % 128 == 0 && p.k >= 512; }

// Below a round of 256 x 256 tiles the slab128 kernel's 64 x 256 tile gives four times the workgroups (1024^3: 64 instead
// of 16 for 256 CUs); efficiency relative to the ping-pong kernel fitted to profiles/r03y_i8_small_tile.txt.
static int mfma_i8_tile(const Problem &p) {  // 0: 256x256, 5: 64x256
  static const TileCandidate cands[] = {{0, 256, 256, 1, 1.00}, {5, 64, 256, 1, 0.45}};
  return p.a_transposed ? 0 : pick_tile(cands, 2, p.n, p.m);
}

// i8_variant: -1 the best the shape allows; 0 slab128; 5 slab128 on the 64 x 256 tile; 10 pingpong_k64; 100 pingpong_32x32x32;
// 200 pingpong_16x16x64 (one resolver for mm_kernel_name and the launcher; a pinned kernel that cannot serve the shape
// falls through).
// K x N A of a wide problem under the shape-adaptive pick: transposition pre-pass (mm_transpose.hip), then the row-major default
static bool transposes_first(const Problem &p) {
  if (tuning(TUNE_I8_VARIANT) >= 0 || !transposes_first_small(p, 1)) return false;
  Problem q = p;
  q.a_transposed = false;
  return pp128_serves(q) && mfma_i8_tile(q) == 0;
}

static Kind resolve(const Problem &p) {
  if (!mfma_i8_serves(p)) return K_NONE;
  const int v = tuning(TUNE_I8_VARIANT);
  if (!(v < 0 || v == 0 || v == 5 || v == 10 || v == 100 || v == 200)) return K_NONE;   // lab ids are not in this library
  if (transposes_first(p)) return K_PP16;
  if (p.a_transposed) return (v != 0 && ppk64_serves(p)) ? K_PPK64_AT : K_SLAB128_AT;
  if (v == 5 || (v < 0 && mfma_i8_tile(p) == 5)) return K_SLAB128_64;
  if (v == 0) return K_SLAB128;
  if ((v < 0 || v == 200) && pp128_serves(p)) return K_PP16;  // +7.6 % over pingpong_32x32x32 (profiles/r03e_*)
  if (v == 100 && pp128_serves(p)) return K_PP32;
  if (ppk64_serves(p)) return K_PPK64;
  return K_SLAB128;
}

const char *mfma_i8_name(const Problem &p) { return kNames[resolve(p)]; }

template <typename Kern>
static int launch_tile(hipStream_t s, const Problem &p, Kern kern, unsigned threads, int lds, unsigned long long &configured,
                       unsigned bm = 256) {
  const unsigned tiles_n = (p.n + bm - 1) / bm, tiles_m = (p.m + 255) / 256;
  if (int e = ensure_dynamic_lds((const void *)kern, lds, configured)) return e;
  hipLaunchKernelGGL(kern, dim3(tiles_n * tiles_m), dim3(threads), lds, s, (const signed char *)p.a, (const signed char *)p.b,
                     (signed char *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, band_rows(bm, 256, 1));
  return (int)hipGetLastError();
}

// A K x N A served where it lies (no workspace): the ping-pong K x N kernel where its shape rules allow, else slab128's
static int launch_kxn_in_place(hipStream_t s, const Problem &p, unsigned long long (&cfg)[K_NONE]) {
  if (tuning(TUNE_I8_VARIANT) != 0 && ppk64_serves(p))
    return launch_tile(s, p, mfma_i8_pp_kernel<true>, GeoI8PP::THREADS, GeoI8PP::LDS_BYTES, cfg[K_PPK64_AT]);
  return launch_tile(s, p, mfma_i8_kernel<GeoI8, true>, GeoI8::THREADS, GeoI8::LDS_BYTES, cfg[K_SLAB128_AT]);
}

int launch_mfma_i8(hipStream_t s, const Problem &p) {
  static unsigned long long cfg[K_NONE] = {};
  if (transposes_first(p)) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipMemPool_t pool = nullptr;
    if (int rc = workspace_pool(dev, &pool)) return rc;
    void *an = nullptr;
    if ((e = hipMallocFromPoolAsync(&an, (size_t)p.n * p.k, pool, s)) != hipSuccess) {
      // no room for the N x K copy (up to 4 GiB next to a nearly full device): not an error -- the K x N kernels serve the
      // call without extra memory, as they did before the pre-pass existed; same contract, ~2-3 % slower
      (void)hipGetLastError();
      return launch_kxn_in_place(s, p, cfg);
    }
    int rc = launch_transpose_kxn(s, p.a, an, p.k, p.n, 1);
    if (rc == 0) {
      Problem q = p;
      q.a = an;
      q.a_transposed = false;
      rc = launch_mfma_i8(s, q);
    }
    const hipError_t f = hipFreeAsync(an, s);
    return rc ? rc : (int)f;
  }
  const Kind k = resolve(p);
  switch (k) {
    case K_PP16: return launch_tile(s, p, mfma_i8_pp2s_kernel, GeoI8PP2::THREADS, GeoI8PP2::LDS_BYTES, cfg[k]);
    case K_PP32: return launch_tile(s, p, mfma_i8_pp2_kernel, GeoI8PP2::THREADS, GeoI8PP2::LDS_BYTES, cfg[k]);
    case K_PPK64: return launch_tile(s, p, mfma_i8_pp_kernel<false>, GeoI8PP::THREADS, GeoI8PP::LDS_BYTES, cfg[k]);
    case K_PPK64_AT: return launch_tile(s, p, mfma_i8_pp_kernel<true>, GeoI8PP::THREADS, GeoI8PP::LDS_BYTES, cfg[k]);
    case K_SLAB128: return launch_tile(s, p, mfma_i8_kernel<GeoI8, false>, GeoI8::THREADS, GeoI8::LDS_BYTES, cfg[k]);
    case K_SLAB128_AT: return launch_tile(s, p, mfma_i8_kernel<GeoI8, true>, GeoI8::THREADS, GeoI8::LDS_BYTES, cfg[k]);
    case K_SLAB128_64: return launch_tile(s, p, mfma_i8_kernel<GeoI8S, false>, GeoI8S::THREADS, GeoI8S::LDS_BYTES, cfg[k], GeoI8S::BM);
    default: return kErrNotSupported;
  }
}

}  // namespace mm
