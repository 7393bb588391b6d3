// fp32 (Multiply, Add) on the bf16 matrix cores: C = A x B with every fp32 operand split into three
// bf16 planes, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); each
// subtraction is exact in fp32, so the three planes carry all 24 significand bits), and the six
// products of weight >= 2^-16 kept:
//     a*b ~= a1*b3 + a3*b1 + a2*b2 + a1*b2 + a2*b1 + a1*b1        (dropped: a2*b3 + a3*b2 + a3*b3 <= 2^-25 |a*b|)
// Every bf16 x bf16 product is exact in the MFMA's fp32 datapath, so a product is represented to
// ~2^-25 relative -- finer than the 2^-24 rounding of an fp32 multiply -- and the accumulation is
// fp32, 16 k per instruction (v_mfma_f32_32x32x16_bf16) instead of the fp32 MFMA's 2.
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) peaks at 157.3 TFLOP/s; six bf16 MFMAs per 16 k do the
// same work in 6 x 32 cycles instead of 8 x 64 -- 2.67x at equal clock.
// This is MM_PATH_SPLIT: opt-in, same role as the reference's kernel (C = A x B, fp32 in, fp32 out,
// include/MatrixMultiplication.h:155-171), tolerance-checked like every fp32 fast path.
//
// Two steps per launch, both on the caller's stream:
//  1. pack: A and B are split AND re-tiled into the exact LDS image the GEMM kernel wants --
//       [256-row (A) / 256-column (B) block][16-deep k slab][plane 3][fragment 8][lane 64][8 bf16]
//     one fragment = the operand of one MFMA (32 rows x 16 k; lane l holds row l%32, k = 8*(l/32)..+7),
//     so a k slab of a block is ONE contiguous 24 KiB run: the GEMM kernel's LDS-DMA is a linear copy
//     (full cache lines, no swizzle arithmetic) and its fragment reads are lane-linear ds_read_b128
//     (conflict-free by construction, no transpose reads).  Rows / columns / k beyond the matrix are
//     written as zeros, so the GEMM kernel has no edge handling on loads and serves ANY N, K, M.
//     Cost: reads 4 B, writes 6 B per element of A and B at 4.5 TB/s -- O(N^2), 3.4 % of the launch at 16384^3.
//  2. GEMM: 256 x 256 tile per workgroup, 8 wavefronts (2 x 4, 128 x 64 each = 4 x 2 MFMA tiles),
//     ring of three 48 KiB stages (A slab + B slab) filled by global_load_lds_dwordx4 two stages
//     ahead, counted vmcnt.  Per stage a wavefront issues 48 MFMAs (1536 cycles) against 18 fragment
//     reads and 6 DMA pieces.  Shipped schedule: ping-pong -- a load segment and a compute segment per
//     stage, the two wavefronts of a SIMD one segment apart (pp_stage below).  Kept as variants: one
//     barrier per stage with double-buffered fragment registers, reads ahead of / interleaved with
//     the MFMA groups (stage below) -- the steps this kernel went through (HISTORY.md 3.3).
//
// The workspace (6 bytes per element of A and B, padded to whole blocks/slabs) is allocated
// stream-ordered (hipMallocAsync / hipFreeAsync on the launch stream): no hidden global state, safe
// for concurrent launches, and the pool keeps the memory between launches.
#include <mutex>
#include <type_traits>

#include "mm_common.h"

namespace mm {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Packed global layout (both operands, always): [256-row block][16-k slab][plane 3][fragment 8][lane 64][8 bf16]
struct Packed {
  static constexpr int FRAG_BYTES = 64 * 16;                 // one MFMA operand: 64 lanes x 8 bf16
  static constexpr int PLANE_BYTES = 8 * FRAG_BYTES;         // 256 rows x 16 k of one plane
  static constexpr int SLAB_BYTES = 3 * PLANE_BYTES;         // 24 KiB: one k slab of one block, three planes
};
// Tile geometries of the GEMM kernel.  FR = fragments (32-row groups) of a tile side; the LDS image of a stage is
// [A: plane][FR fragments] [B: plane][FR fragments], i.e. the packed layout restricted to the tile's fragments.
template <int FR_, int WM_, int WN_, int TM_>
struct GeoT {
  static constexpr int FR = FR_, BM = 32 * FR_, BN = 32 * FR_, WM = WM_, WN = WN_, THREADS = 64 * WM_ * WN_;
  static constexpr int TM = TM_, TN = 2, NS = 3;
  static_assert(WM * TM == FR && WN * TN == FR && WM * WN == FR, "one wavefront per fragment and per DMA piece");
  static constexpr int FRAG_BYTES = Packed::FRAG_BYTES;
  static constexpr int PLANE_BYTES = FR * FRAG_BYTES;        // in LDS
  static constexpr int SLAB_BYTES = 3 * PLANE_BYTES;
  static constexpr int STAGE_BYTES = 2 * SLAB_BYTES;         // A slab + B slab
  static constexpr int LDS_BYTES = NS * STAGE_BYTES;
  static constexpr int HALVES = 8 / FR;                      // tiles per packed 256-row block
};
using GeoS = GeoT<8, 2, 4, 4>;    // 256 x 256, 8 wavefronts of 128 x 64, 144 KiB LDS: one workgroup per CU
using GeoS128 = GeoT<4, 2, 2, 2>; // 128 x 128, 4 wavefronts of 64 x 64, 72 KiB LDS: two workgroups per CU (mid-size problems)

// ---- step 1: split + re-tile -----------------------------------------------------------------------
__device__ __forceinline__ void split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;        // exact
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;       // exact
  p3 = (__bf16)r2;
  if (!(__builtin_fabsf((float)p1) < __builtin_inff())) {  // inf / nan (or a finite x that rounds up to inf):
    p2 = (__bf16)0.0f;                                     // keep the value in plane 1 only, no inf - inf
    p3 = (__bf16)0.0f;
  }
}

// One thread = one lane of one fragment: 8 consecutive k of one row (ALONG_K: the source is
// contiguous along k, i.e. row-major A) or of one column (source contiguous along the row/column
// index: B, or A stored K x N).  `rows` = extent of the blocked dimension, `ld` = source leading
// dimension.  grid = (k slabs, blocks), 512 threads = 8 fragments x 64 lanes.
// PAIRED (the B operand): fragment f, lane column j stands for column (f/2)*64 + 2*j + f%2 of the block, so the two
// fragments a GEMM wavefront multiplies side by side interleave and each lane ends up owning 2 ADJACENT columns of C
// (8-byte accesses, 256 contiguous bytes per row per half-wavefront in the write-back).
template <bool ALONG_K, bool PAIRED>
__global__ __launch_bounds__(512) void split_pack_kernel(const float *__restrict__ src, char *__restrict__ dst,
                                                         unsigned rows, unsigned K, unsigned ld, unsigned slabs) {
  const unsigned slab = blockIdx.x, blk = blockIdx.y;
  const unsigned frag = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned r = blk * 256 + (PAIRED ? (frag >> 1) * 64 + 2 * (lane & 31u) + (frag & 1u) : frag * 32 + (lane & 31u));
  const unsigned k0 = slab * 16 + (lane >> 5) * 8;
  float x[8];
  if (ALONG_K) {
    if (r < rows && k0 + 8 <= K && (ld & 3u) == 0) {
      const f32x4 lo = *(const f32x4 *)(src + (size_t)r * ld + k0), hi = *(const f32x4 *)(src + (size_t)r * ld + k0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (r < rows && k0 + e < K) ? src[(size_t)r * ld + k0 + e] : 0.0f;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (r < rows && k0 + e < K) ? src[(size_t)(k0 + e) * ld + r] : 0.0f;
  }
  bf16x8 p1, p2, p3;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 a, b, c;
    split3(x[e], a, b, c);
    p1[e] = a; p2[e] = b; p3[e] = c;
  }
  char *out = dst + ((size_t)blk * slabs + slab) * Packed::SLAB_BYTES + frag * Packed::FRAG_BYTES + lane * 16;
  *(bf16x8 *)(out) = p1;
  *(bf16x8 *)(out + Packed::PLANE_BYTES) = p2;
  *(bf16x8 *)(out + 2 * Packed::PLANE_BYTES) = p3;
}

// a pointer every lane agrees on, in scalar registers (the DMA instruction takes its base from an SGPR pair)
__device__ __forceinline__ const char *uniform(const char *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char *)(((unsigned long long)hi << 32) | lo);
}

// ---- step 2: GEMM over the packed planes -----------------------------------------------------------
// VAR bit 0: s_setprio(1) around the MFMA groups; bit 1: no chunked flush (one chain over all of K).
// (The no-DMA / L2-resident-source ablations of the power breakdown live in tools/lab/lab_mfma_f32_split.hip.)
// bit 4: fragment reads interleaved one by one with the MFMAs (sched_group_barrier) instead of issued in front of them;
// bit 5: flush every 4128 k instead of every 8256; bit 6: ping-pong schedule (see pp_stage).
// TERMS: 6 (default) or 3 (a1b2 + a2b1 + a1b1 only:
// products to ~2^-16, the "three-pass" accuracy class; measurement knob, not dispatched by default).
template <int VAR, int TERMS, class G = GeoS>
__global__ __launch_bounds__(G::THREADS, 2)  // 2 wavefronts per SIMD in both geometries
void mfma_f32_split_kernel(const char *__restrict__ Ap, const char *__restrict__ Bp, float *__restrict__ C, unsigned N,
                           unsigned M, unsigned slabs, unsigned tiles_n, unsigned tiles_m, unsigned kBand) {
  constexpr int TM = G::TM, TN = G::TN;
  constexpr int NP = TERMS == 6 ? 3 : 2;  // planes read
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr bool PP = (VAR & 64) != 0;
  static_assert(!PP || G::THREADS == 512, "the ping-pong schedule pairs wavefronts w and w+4 of one workgroup");
  const unsigned group = wave >> 2;               // PP: wavefronts w and w+4 share a SIMD and run one segment apart
  const unsigned wm = PP ? (wave & 3u) >> 1 : wave / G::WN;
  const unsigned wn = PP ? (wave & 1u) * 2 + group : wave % G::WN;  // WM x WN wavefront grid
  const unsigned lo = lane & 31u, hi = lane >> 5;

  const unsigned lin = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned tile_r = band * kBand + within % rows_in_band, tile_c = within / rows_in_band;

  // ---- DMA: wavefront w moves fragment w of each plane of the tile's A slab and B slab (1 KiB pieces: source plane
  // stride 8 KiB in the packed layout, LDS plane stride FR KiB); a 128-row tile is one half of a packed 256-row block
  const unsigned voff0 = wave * 1024 + lane * 16, voff1 = voff0 + 8 * 1024, voff2 = voff0 + 16 * 1024;
  const char *a_base = Ap + (size_t)(tile_r / G::HALVES) * slabs * Packed::SLAB_BYTES + (tile_r % G::HALVES) * G::FR * Packed::FRAG_BYTES;
  const char *b_base = Bp + (size_t)(tile_c / G::HALVES) * slabs * Packed::SLAB_BYTES + (tile_c % G::HALVES) * G::FR * Packed::FRAG_BYTES;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  auto issue = [&](unsigned slab, unsigned buf) {
    const unsigned sl = min(slab, slabs - 1);  // past the end: harmless re-fetch into a dead buffer
    const char *ap = uniform(a_base + (size_t)sl * Packed::SLAB_BYTES), *bp = uniform(b_base + (size_t)sl * Packed::SLAB_BYTES);
    const unsigned la = lds0 + buf * G::STAGE_BYTES + wave * 1024, lb = la + G::SLAB_BYTES;
    unsigned keep;
#define MM_PIECE(vo, sb, la) "s_mov_b32 m0, " la "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " vo ", " sb "\n\t"
    if constexpr (NP == 3) {
      const unsigned la1 = la + G::PLANE_BYTES, la2 = la + 2 * G::PLANE_BYTES, lb1 = lb + G::PLANE_BYTES, lb2 = lb + 2 * G::PLANE_BYTES;
      asm volatile("s_mov_b32 %0, m0\n\t" MM_PIECE("%1", "%4", "%6") MM_PIECE("%2", "%4", "%7") MM_PIECE("%3", "%4", "%8")
                       MM_PIECE("%1", "%5", "%9") MM_PIECE("%2", "%5", "%10") MM_PIECE("%3", "%5", "%11") "s_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff0), "v"(voff1), "v"(voff2), "s"(ap), "s"(bp), "s"(la), "s"(la1), "s"(la2), "s"(lb), "s"(lb1),
                     "s"(lb2)
                   : "memory");
    } else {  // planes 1 and 2 only: 16 KiB per operand, wavefront w moves KiB w and w+8
      const unsigned la1 = la + G::PLANE_BYTES, lb1 = lb + G::PLANE_BYTES;
      asm volatile("s_mov_b32 %0, m0\n\t" MM_PIECE("%1", "%3", "%5") MM_PIECE("%2", "%3", "%6") MM_PIECE("%1", "%4", "%7")
                       MM_PIECE("%2", "%4", "%8") "s_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff0), "v"(voff1), "s"(ap), "s"(bp), "s"(la), "s"(la1), "s"(lb), "s"(lb1)
                   : "memory");
    }
#undef MM_PIECE
  };
  constexpr int PIECES = 2 * NP;  // DMA pieces per wavefront per stage

  // ---- fragment addresses: lane-linear inside a fragment
  const unsigned a_off = (wm * TM) * G::FRAG_BYTES + lane * 16;                    // + mi*FRAG + plane*PLANE
  const unsigned b_off = G::SLAB_BYTES + (wn * TN) * G::FRAG_BYTES + lane * 16;    // + ni*FRAG + plane*PLANE

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (f32x16)0.0f;

  bf16x8 af[2][3], bf[2][TN][3];  // [parity][plane]: A per row-tile, B per stage
  auto read_a = [&](const char *stage, int mi, bf16x8 (&dst)[3]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) dst[p] = *(const bf16x8 *)(stage + a_off + mi * G::FRAG_BYTES + p * G::PLANE_BYTES);
  };
  auto read_b = [&](const char *stage, bf16x8 (&dst)[TN][3]) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[ni][p] = *(const bf16x8 *)(stage + b_off + ni * G::FRAG_BYTES + p * G::PLANE_BYTES);
  };
  // the products of one row-tile, smallest weights first; consecutive MFMAs alternate between the two accumulators
  auto mac = [&](int mi, const bf16x8 (&a)[3], const bf16x8 (&b)[TN][3]) {
    constexpr int PA6[6] = {0, 2, 1, 0, 1, 0}, PB6[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PA3[3] = {0, 1, 0}, PB3[3] = {1, 0, 0};
    if ((VAR & 1) && !(VAR & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int pa = TERMS == 6 ? PA6[t] : PA3[t % 3], pb = TERMS == 6 ? PB6[t] : PB3[t % 3];
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[ni][pb], acc[mi][ni], 0, 0, 0);
      }
    if ((VAR & 1) && !(VAR & 16)) __builtin_amdgcn_s_setprio(0);
  };
  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // one stage s: BUF = s % 3 (LDS buffer), PAR = s % 2 (which B register set holds this stage's fragments)
  auto stage = [&](auto bufc, auto parc, unsigned s) {
    constexpr int BUF = decltype(bufc)::value, PAR = decltype(parc)::value;
    const char *cur = smem + BUF * G::STAGE_BYTES, *nxt = smem + ((BUF + 1) % 3) * G::STAGE_BYTES;
#pragma unroll
    for (int mi = 0; mi + 1 < TM; ++mi) {   // row-tile mi multiplies while row-tile mi+1's fragments arrive
      read_a(cur, mi + 1, af[(mi + 1) & 1]);
      mac(mi, af[mi & 1], bf[PAR]);
    }
    if constexpr (VAR & 16) {  // one fragment read between MFMAs instead of three ahead of a group: the 8 wavefronts of a
                               // workgroup run in step after a barrier, and 8 x 3 (or 8 x 9) simultaneous 1 KiB reads
                               // queue on the 128 B/clk LDS port while instruction issue -- MFMAs included -- waits
#pragma unroll
      for (int grp = 0; grp + 1 < TM; ++grp) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS - NP, 0);
      }
    }
    // own pieces of stage s+1 landed (stage s+2's may still fly); every read of stage s has returned
    if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();                                 // stage s+1 is published, buffer BUF is free
    issue(s + 3, BUF);
    read_b(nxt, bf[PAR ^ 1]);
    read_a(nxt, 0, af[0]);
    mac(TM - 1, af[(TM - 1) & 1], bf[PAR]);
    if constexpr (VAR & 16) {
      constexpr int READS = (TN + 1) * NP, MFMAS = 2 * TERMS;
#pragma unroll
      for (int i = 0; i < (READS < MFMAS ? READS : MFMAS); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (MFMAS > READS) __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - READS, 0);
      else if (READS > MFMAS) __builtin_amdgcn_sched_group_barrier(0x100, READS - MFMAS, 0);
    }
  };

  // Ping-pong form of a stage (VAR bit 6): a LOAD segment (all 18 fragments of the stage into registers, the 6 DMA
  // pieces of stage s+2, the counted wait) and a COMPUTE segment (48 MFMAs, nothing else), a barrier after each.  The
  // two wavefronts of a SIMD run one segment apart (group 1 starts one barrier late), so one is always in its compute
  // segment: the matrix core never waits for a barrier round trip, a DMA issue or a fragment read.  Ring of 3 still
  // suffices: stage s+2 goes into the buffer of stage s-1, whose last reader (group 1's load segment s-1) finished
  // before the barrier that opens group 0's load segment s (tests/test_schedules.py replays this).
  bf16x8 paf[TM][3];  // PP only: all A fragments of the stage (B fragments reuse bf[0])
  auto pp_stage = [&](auto bufc, unsigned s) {
    constexpr int BUF = decltype(bufc)::value;
    const char *cur = smem + BUF * G::STAGE_BYTES;
    read_b(cur, bf[0]);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) read_a(cur, mi, paf[mi]);
    issue(s + 2, (BUF + 2) % 3);
    if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) mac(mi, paf[mi], bf[0]);
    __builtin_amdgcn_s_setprio(0);
    sync();
  };

  // ---- prologue: three stages in flight, stage 0 published, its first fragments in registers
  const bool shifted = PP && group == 1;
  issue(0, 0);
  issue(1, 1);
  if constexpr (PP) {
    if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    sync();
    if (shifted) sync();  // group 1 runs one segment behind group 0
  } else {
    issue(2, 2);
    if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    sync();
    read_b(smem, bf[0]);
    read_a(smem, 0, af[0]);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // C (+)= accumulators.  Lane (lo, hi) owns columns 2*lo, 2*lo+1 of the wavefront's 64 (see PAIRED above) in rows
  // (rr&3) + 8*(rr>>2) + 4*hi of each 32-row tile.  Addressing: uniform 64-bit base of the wavefront's 128 x 64 block
  // + a 32-bit per-lane offset (128 rows x M x 4 B < 4 GiB).
  // "+=" reads C in batches of 8 rows (16 registers): the kernel has ~25 spare registers in the main loop, so the 64
  // row-pairs of a lane cannot all be in flight; 8 dependent round trips per write-back.  (No-return float atomics
  // need no registers at all but run at ~0.8 TB/s chip-wide in the L2 -- measured 80 us per round of tiles.)
  const unsigned row_w = tile_r * G::BM + wm * (TM * 32), col_w = tile_c * G::BN + wn * 64, col = col_w + 2 * lo;
  char *const c_wave = (char *)uniform((const char *)(C + (size_t)row_w * M + col_w));
  const bool pairs = (M & 1u) == 0 && ((size_t)C & 7u) == 0;
  auto writeback = [&](bool accumulate, auto finalc) {
    constexpr bool FINAL = decltype(finalc)::value;  // nothing else is live any more: 16 rows per batch, batches free to overlap
    constexpr int BATCH = FINAL ? 16 : 8;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    unsigned Mv = M;
    asm volatile("" : "+s"(Mv));  // opaque: keeps the 64 row offsets from being hoisted out of the chunk loop (they would
                                  // live across the main loop and spill)
    const unsigned lane_off = (4 * hi * Mv + 2 * lo) * 4u;
    const unsigned rows_left = N > row_w ? N - row_w : 0u;  // the lower wavefront row of a ragged tile may own no row at all
    // Interior wavefront blocks (every row and column in range, 8-byte accesses possible): straight-line code, no
    // per-lane predicate -- the predicated form below costs a branch and a full s_waitcnt per element, which
    // serialises the 64 loads of a read-modify-write into 64 round trips.
    const bool interior = pairs && row_w + TM * 32 <= N && col_w + 64 <= M;  // wavefront-uniform
    if (interior) {
      typedef __attribute__((address_space(1))) f32x2 *gpair_t;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int half = 0; half < 16 / BATCH; ++half) {
          f32x2 old[BATCH];
          if (accumulate) {
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
              const int rr = half * BATCH + j, lrow = mi * 32 + (rr & 3) + 8 * (rr >> 2);
              old[j] = *(gpair_t)(c_wave + (lane_off + (unsigned)lrow * Mv * 4u));
            }
          } else {
#pragma unroll
            for (int j = 0; j < BATCH; ++j) old[j] = f32x2{0.0f, 0.0f};
          }
#pragma unroll
          for (int j = 0; j < BATCH; ++j) {
            const int rr = half * BATCH + j, lrow = mi * 32 + (rr & 3) + 8 * (rr >> 2);
            *(gpair_t)(c_wave + (lane_off + (unsigned)lrow * Mv * 4u)) = f32x2{acc[mi][0][rr], acc[mi][1][rr]} + old[j];
          }
          if (!FINAL) __builtin_amdgcn_sched_barrier(0);
        }
      return;
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int half = 0; half < 16 / BATCH; ++half) {
        f32x2 old[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int rr = half * BATCH + j, lrow = mi * 32 + (rr & 3) + 8 * (rr >> 2);
          const float *src = (const float *)(c_wave + (lane_off + (unsigned)lrow * Mv * 4u));
          const bool row_ok = lrow + 4 * hi < rows_left;
          old[j] = f32x2{0.0f, 0.0f};
          if (accumulate && row_ok) {
            if (pairs) { if (col < M) old[j] = *(const f32x2 *)src; }
            else {
              if (col < M) old[j][0] = src[0];
              if (col + 1 < M) old[j][1] = src[1];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int rr = half * BATCH + j, lrow = mi * 32 + (rr & 3) + 8 * (rr >> 2);
          float *dst = (float *)(c_wave + (lane_off + (unsigned)lrow * Mv * 4u));
          const bool row_ok = lrow + 4 * hi < rows_left;
          const f32x2 v = f32x2{acc[mi][0][rr], acc[mi][1][rr]} + old[j];
          if (row_ok) {
            if (pairs) { if (col < M) *(f32x2 *)dst = v; }
            else {
              if (col < M) dst[0] = v[0];
              if (col + 1 < M) dst[1] = v[1];
            }
          }
        }
        if (!FINAL) __builtin_amdgcn_sched_barrier(0);
      }
  };

  // Chunked accumulation: every FC slabs (8256 k) the tile is added into C and the chains restart from zero, which
  // bounds the drift of a long chain of same-sign terms exactly as the fp32 kernel's flush does (mm_mfma_f32.hip:
  // one extra read + write of C per chunk, by the workgroup that owns the tile -- race-free and deterministic).
  // FC is a multiple of 6, so the stage -> (LDS buffer, register set) assignment carries across chunks.
  constexpr unsigned FC = (VAR & 2) ? ~0u / 2 / 6 * 6 : ((VAR & 32) ? 258 : 516);
  bool flushed = false;
  unsigned c0 = 0, cend = min(FC, slabs);
  for (;;) {
    for (unsigned s = c0; s < cend; s += 6) {
      if constexpr (PP) {
        pp_stage(I0{}, s);
        if (s + 1 < cend) pp_stage(I1{}, s + 1);
        if (s + 2 < cend) pp_stage(I2{}, s + 2);
        if (s + 3 < cend) pp_stage(I0{}, s + 3);
        if (s + 4 < cend) pp_stage(I1{}, s + 4);
        if (s + 5 < cend) pp_stage(I2{}, s + 5);
        continue;
      }
      stage(I0{}, I0{}, s);
      if (s + 1 < cend) stage(I1{}, I1{}, s + 1);
      if (s + 2 < cend) stage(I2{}, I0{}, s + 2);
      if (s + 3 < cend) stage(I0{}, I1{}, s + 3);
      if (s + 4 < cend) stage(I1{}, I0{}, s + 4);
      if (s + 5 < cend) stage(I2{}, I1{}, s + 5);
    }
    if (cend >= slabs) break;
    // the counted vmcnt waits of the stages must see DMA pieces only (stores may retire out of order with loads):
    // drain before and after the C traffic
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    writeback(flushed, std::false_type{});
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (f32x16)0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flushed = true;
    c0 = cend;
    cend = min(cend + FC, slabs);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (dead) refills land before the LDS is released
  if (PP && !shifted) sync();                         // group 0 executes as many barriers as group 1
  writeback(flushed, std::true_type{});
}

template <int VAR, int TERMS, class G = GeoS>
int launch_gemm(hipStream_t s, const char *ap, const char *bp, const Problem &p, unsigned slabs) {
  static unsigned long long configured = 0;
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_split_kernel<VAR, TERMS, G>, G::LDS_BYTES, configured)) return e;
  hipLaunchKernelGGL((mfma_f32_split_kernel<VAR, TERMS, G>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s, ap,
                     bp, (float *)p.c, p.n, p.m, slabs, tiles_n, tiles_m, band_rows());
  return (int)hipGetLastError();
}

}  // namespace

// (the write-back addresses up to 131 rows x M x 4 B from a wavefront's first row with 32-bit offsets: rows of C beyond
// 8 Mi floats are not served -- ADVICE r2)
bool mfma_f32_split_serves(const Problem &p) { return p.n > 0 && p.m > 0 && p.k > 0 && 132ull * p.m * 4ull < (1ull << 32); }

// 256 x 256 (ping-pong, one workgroup per CU) or 128 x 128 (two workgroups per CU) for problems whose 256-tiles
// would leave compute units idle; same estimator as the other families (mm_common.h).  variant bit 8 (256) pins 256,
// bit 9 (512) pins 128.
int mfma_f32_split_tile(const Problem &p, int variant) {
  if (variant > 0 && (variant & 256)) return 256;
  if (variant > 0 && (variant & 512)) return 128;
  static const TileCandidate cands[] = {{256, 256, 256, 1, 1.0}, {128, 128, 128, 2, 0.85}};
  return pick_tile(cands, 2, p.n, p.m);
}

size_t mfma_f32_split_workspace_bytes(const Problem &p) {
  const size_t slabs = (p.k + 15) / 16;
  return ((size_t)((p.n + 255) / 256) + (size_t)((p.m + 255) / 256)) * slabs * Packed::SLAB_BYTES;
}


// variant (split_variant knob): -1 / 0 = default (6 products, ping-pong schedule, flush every 8256 k); otherwise a bit
// mask: 1 first schedule of the round (one barrier per stage, fragment reads ahead of each MFMA group, s_setprio around
// the group), 2 three products (planes 1 and 2 only), 4 no flush, 8 plain hipMalloc/hipFree workspace (diagnosis),
// 64 flush every 4128 k, 128 one barrier per stage with the fragment reads interleaved between the MFMAs,
// 256 / 512 pin the 256 x 256 / 128 x 128 tile (default: by shape, mfma_f32_split_tile)
int launch_mfma_f32_split(hipStream_t s, const Problem &p, int variant) {
  const unsigned slabs = (p.k + 15) / 16;
  const unsigned blocks_a = (p.n + 255) / 256, blocks_b = (p.m + 255) / 256;
  const size_t a_bytes = (size_t)blocks_a * slabs * Packed::SLAB_BYTES, b_bytes = (size_t)blocks_b * slabs * Packed::SLAB_BYTES;
  // The workspace comes from the pool this library owns (mm_capi.hip: workspace_pool).
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipMemPool_t pool = nullptr;
  if (int rc = workspace_pool(dev, &pool)) return rc;
  char *ws = nullptr;
  const int v = variant < 0 ? 0 : variant;
  if (v & 48) return kErrNotSupported;   // (ablation ids of the lab build)
  const bool sync_alloc = v & 8;
  if (sync_alloc) e = hipMalloc((void **)&ws, a_bytes + b_bytes);
  else e = hipMallocFromPoolAsync((void **)&ws, a_bytes + b_bytes, pool, s);
  if (e != hipSuccess) return (int)e;
  const float *a = (const float *)p.a, *b = (const float *)p.b;
  (void)hipGetLastError();   // a stale error of the application's own calls must not be reported as this launch's
  if (p.a_transposed)
    hipLaunchKernelGGL((split_pack_kernel<false, false>), dim3(slabs, blocks_a), dim3(512), 0, s, a, ws, p.n, p.k, p.n, slabs);
  else
    hipLaunchKernelGGL((split_pack_kernel<true, false>), dim3(slabs, blocks_a), dim3(512), 0, s, a, ws, p.n, p.k, p.k, slabs);
  int rc = (int)hipGetLastError();
  if (rc == 0) {
    hipLaunchKernelGGL((split_pack_kernel<false, true>), dim3(slabs, blocks_b), dim3(512), 0, s, b, ws + a_bytes, p.m, p.k, p.m,
                       slabs);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) {
    const char *bp = ws + a_bytes;
    if (v & 1) rc = launch_gemm<1, 6>(s, ws, bp, p, slabs);
    else if ((v & 128) && (v & 4)) rc = launch_gemm<16 | 2, 6>(s, ws, bp, p, slabs);
    else if (v & 128) rc = launch_gemm<16, 6>(s, ws, bp, p, slabs);
    else if (v & 2) rc = launch_gemm<64, 3>(s, ws, bp, p, slabs);
    else if (v & 4) rc = launch_gemm<64 | 2, 6>(s, ws, bp, p, slabs);
    else if (v & 64) rc = launch_gemm<64 | 32, 6>(s, ws, bp, p, slabs);
    else if (mfma_f32_split_tile(p, v) == 128) rc = launch_gemm<16, 6, GeoS128>(s, ws, bp, p, slabs);
    else rc = launch_gemm<64, 6>(s, ws, bp, p, slabs);
  }
  hipError_t f;
  if (sync_alloc) { f = hipStreamSynchronize(s); (void)hipFree(ws); }
  else f = hipFreeAsync(ws, s);
  return rc ? rc : (int)f;
}

}  // namespace mm
