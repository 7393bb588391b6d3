// fp32 (Multiply, Add) fast path for gfx950: C[N x M] = A[N x K] . B[K x M], all row-major.
//
// This is the MI355X counterpart of the reference's ProcessingElement chain + ReadA/TransposeA/
// ReadB/FeedB/WriteC streaming (kernel/Compute.cpp:11-231, kernel/Memory.cpp:106-438): an output
// tile stays resident on chip for the whole K loop ("we do not tile K further",
// kernel/Compute.cpp:58-60) while K-slabs of the A row-panel and B column-panel stream past it.
// Here the resident tile lives in the accumulation registers of the wavefronts of one workgroup
// and the slabs are DMA'd HBM/L2 -> LDS (global_load_lds, 16 B per lane, no VGPR round trip).
//
// Geometry (template parameters):
//   workgroup = WM x WN wavefronts; each wavefront owns a (TM*32) x 128 block of C as TM x 4
//   v_mfma_f32_32x32x2_f32 accumulators; workgroup tile BM x BN = (WM*TM*32) x (WN*128);
//   K is consumed in slabs of BK floats through a 2-deep LDS ring.
//
// LDS images (both written lane-linearly by the DMA, so any permutation is applied to the
// per-lane SOURCE address):
//   A slab: [BM rows][BK] floats, a row = BK/4 16-byte chunks, chunk index XOR-swizzled with
//           (row >> log2(16/CPR)) so that the 16 lanes of each ds_read_b128 service group hit 16
//           distinct 16-B slots of the 256-B bank row (conflict-free, checked by enumeration in
//           tests/test_layouts.py).
//   B slab: [BK][BN] floats, untouched (a k-row is 1 KiB == one wave-level DMA instruction).
//
// Fragment trick: the MFMA wants A[i][k] / B[k][j] with (i or j) = lane&31 and k = lane>>5.
//   A: lane reads 16 B = A[row = lane&31][4 consecutive k, at k-offset 4*(lane>>5)] -> the 4
//      dwords feed 4 successive MFMAs whose k-pairs are (p, p+4), p = 0..3, of an 8-deep k-group.
//   B: lane reads 16 B = B[k = p + 4*(lane>>5)][4 consecutive columns 4*(lane&31)..+3] -> the 4
//      dwords feed the 4 column-accumulators, accumulator t holding columns 4*j+t.
//   So per 8 k's a wavefront issues TM + 4 ds_read_b128 for 16*TM MFMAs, and in the epilogue a
//   lane owns 4 CONSECUTIVE columns of a row -> one 16-byte store, 512 contiguous bytes per
//   half-wave.
// Accumulation order per output element: k-groups ascending, inside a group k = 0,4,1,5,2,6,3,7.
// Each MFMA is an exact-f32 fused multiply-add chain (one rounding per product).
//
// Bounding the summation chain (a single 16384-long f32 chain of positive products drifts to ~1e-5
// relative in the worst element) -- policy `Chain`:
//   FlushIntoC (shipped): every CHUNK slabs the workgroup adds its accumulators into its own C tile in
//     HBM (first chunk: plain store) and restarts them from zero.  The same workgroup owns the tile for
//     the whole launch, so the read-modify-write is race-free and deterministic; it costs one extra
//     read+write of C per chunk, mostly served by the Infinity Cache.  Worst error 2.7e-6 at K = 16384.
//   TwoLevel: every CHUNK slabs the MFMA accumulators are added into a second register set (costs the
//     registers of half a tile; kept as the independently written cross-check of FlushIntoC).
//   Single: one chain over all of K (cross-check geometry with compiler-placed fragment reads).
//
// What this file holds is what MM_PATH_AUTO can dispatch plus two independently scheduled cross-check
// geometries: the three big-tile geometries through one kernel body (tile_body), and around it the launch forms for
// problems that do not fill whole rounds of tiles -- the 64 x 64 geometry (below a round of 128 x 128 tiles; its own small
// kernel, same per-element arithmetic), split-K with an ordered reduce kernel (few tiles, long K), stream-K in teams, ONE
// kernel in which the last part of a cut tile to arrive gathers (between whole rounds; Combine::LastArriver -- the two-kernel
// fix-up form and the counter-ticket form are the cross-checks, f32_splitk 11 / 12), and a transposition pre-pass for a K x N A outside whole rounds
// of the K x N kernel.  Every one of them is deterministic; the whole-tile forms are bit-identical to one another.
// The schedules, ring depths and ablations this kernel went through (HISTORY.md 3.1) live in
// tools/lab/lab_mfma_f32.hip and are built into tools/lab/libmm_gemm_amd_lab.so, not into the product.
//
// Edges: N arbitrary (row indices clamped for loads, stores predicated); M % 4 == 0 (column
// chunks clamped / predicated); K % 8 == 0 (a partial last slab is consumed in 8-deep groups;
// the DMA of its unused part is clamped to valid addresses and never read).  Everything else is
// served by the predicated kernels (mm_valu_tile / mm_ordered).
#include <cstdlib>
#include <type_traits>

#include "mm_common.h"

namespace mm {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// How the summation chain of an output element is bounded (see the header).
enum class Chain { Single, TwoLevel, FlushIntoC };
// Who places the fragment reads of a k-group.
//   CompilerPlaced: "read the next group, multiply this one" as written; the machine scheduler sinks every read
//     down to its first use (ds_read x4 | s_waitcnt lgkmcnt(0) | 16 MFMAs, four times per slab).
//   Pipelined (shipped): the order is pinned -- 16 MFMAs | reads of the next group | 16 MFMAs; in a slab's last
//     group: 16 MFMAs | wait + barrier | DMA of slab t+2 and the reads of slab t+1's first group, one between
//     MFMAs | the remaining MFMAs -- so every fragment is requested >= 16 MFMAs (1024 cycles) before its first
//     use.  Bit-identical to CompilerPlaced (same MFMAs in the same order).
enum class Reads { CompilerPlaced, Pipelined };
// How a DMA piece gets its source address.
//   VectorAddress: the builtin form, a 64-bit address per lane (two VGPRs + v_lshl_add_u64 per slab).
//   ScalarBase (shipped): uniform 64-bit base in SGPRs + a constant 32-bit per-lane offset
//     (global_load_lds_dwordx4 v, s[base:base+1]): +2.5 % at 16384^3.  Needs K >= BK and tile rows within 4 GiB
//     of the base (sdma_fits); other problems take the VectorAddress twin of the same geometry.
enum class Dma { VectorAddress, ScalarBase };

template <int TM_, int WM_, int WN_, int BK_, Chain CHAIN_, int CHUNK_, Reads READS_, Dma DMA_>
struct Geo {
  static constexpr Chain CHAIN = CHAIN_;
  static constexpr Reads READS = READS_;
  static constexpr Dma DMA = DMA_;
  static constexpr int CHUNK = CHUNK_;               // slabs per chunk of the chain (TwoLevel / FlushIntoC)
  static constexpr int TM = TM_, WM = WM_, WN = WN_, BK = BK_, NS = 2;
  static constexpr int TN = 4;                       // 4 accumulators x 32 = 128 columns per wave
  static constexpr int NW = WM * WN;                 // wavefronts per workgroup
  static constexpr int THREADS = NW * 64;
  static constexpr int BM = WM * TM * 32, BN = WN * 128;
  static constexpr int CPR = BK / 4;                 // 16-B chunks per A row
  static constexpr int SWZ_SHIFT = (CPR == 4) ? 2 : 1;
  static constexpr int A_BYTES = BM * BK * 4, B_BYTES = BK * BN * 4;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES;
  // 4-wavefront geometries whose LDS and accumulators fit twice on a CU are compiled for 2 wavefronts per SIMD
  // (<= 256 VGPRs): two INDEPENDENT workgroups then share every SIMD, each with its own barriers
  static constexpr int MIN_WAVES =
      (NW == 4 && 2 * LDS_BYTES <= 160 * 1024 && (CHAIN_ == Chain::TwoLevel ? 2 : 1) * TM * TN * 16 <= 128) ? 2 : 1;
  static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;  // wave-level DMA instructions
  static constexpr int LA = NA / NW, LB = NB / NW;                // ... per wavefront
  static constexpr int KG = BK / 8;                               // 8-deep k-groups per slab
  static_assert(BK == 16 || BK == 32, "BK");
  static_assert(NA % NW == 0 && NB % NW == 0, "DMA instructions must split evenly over waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(DMA_ == Dma::VectorAddress || READS_ == Reads::Pipelined, "the scalar-base DMA pieces are placed by hand");
};

// One k-range of one output tile: A / B point at the first k of the range (row stride of A: lda), K = its length, the
// result goes to Cst[row * ldc + col] for rows < Nst, cols < Mst.  For a whole tile of C that is (C, M, N, M); a partial
// tile of a stream-K launch targets a 128 x 128 scratch slot instead (ldc = 128, no limits).
template <typename G, bool AT, bool AGENT_STORES = false>
__device__ __forceinline__ void tile_body(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ Cst,
                                          unsigned N, unsigned K, unsigned M, unsigned lda, unsigned ldc, unsigned Nst,
                                          unsigned Mst, unsigned row0, unsigned col0) {
  constexpr int TM = G::TM, TN = G::TN, BK = G::BK, NS = G::NS, CPR = G::CPR;
  constexpr bool TWO_LEVEL = G::CHAIN == Chain::TwoLevel, FLUSH = G::CHAIN == Chain::FlushIntoC;
  constexpr bool PIPELINED = G::READS == Reads::Pipelined, SDMA = G::DMA == Dma::ScalarBase;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // AGENT_STORES (a stream-K scratch slot, read by a workgroup on another XCD): the predicated stores go out
  // at agent scope (sc1: written through this XCD's L2 to the agent's coherence point), so that handing it over needs no
  // write-back of the whole L2 -- see mfma_f32_streamk_teams_kernel.  Inline asm, because the language has no 16-byte
  // scoped store; the compiler's hazard recogniser does not look inside it, so the wait state a store of more than 8 bytes
  // needs before its data registers are written again (the compiler inserts it for its own stores) is part of the statement.
  static_assert(!AGENT_STORES || G::TM == 1, "agent-scope stores are wired into the predicated write-back only");
  auto store_quad = [](f32x4 *dst, f32x4 v) {
    if constexpr (AGENT_STORES) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    else *dst = v;
  };

  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wm = wave / G::WN, wn = wave % G::WN;
  const unsigned lo = lane & 31u, hi = lane >> 5;

  // ---- per-lane DMA sources ----------------------------------------------------------------
  // A instruction ja covers LDS 16-B slots [ja*64, ja*64+64): slot -> (row, physical chunk).
  size_t a_row_off[G::LA];   // element offset of the (clamped) source row
  unsigned a_kchunk[G::LA];  // logical k-chunk (x4 floats) this lane fetches
#pragma unroll
  for (int i = 0; i < G::LA; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    if (AT) {
      // A stored K x N (MM_TRANSPOSED_A, kernel/Memory.cpp:205-228): the slab is [BK][BM] like B's,
      // a k-row of the tile is contiguous in memory; a_kchunk = k-row, a_row_off = column offset.
      a_kchunk[i] = slot / (G::BM / 4);
      a_row_off[i] = min(row0 + (slot % (G::BM / 4)) * 4, N - 4);
    } else {
      const unsigned row = slot / CPR, pc = slot % CPR;
      a_kchunk[i] = pc ^ ((row >> G::SWZ_SHIFT) & (CPR - 1));
      a_row_off[i] = (size_t)min(row0 + row, N - 1) * lda;
    }
  }
  unsigned b_krow[G::LB];
  unsigned b_col[G::LB];
#pragma unroll
  for (int i = 0; i < G::LB; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    b_krow[i] = slot / (G::BN / 4);
    b_col[i] = min(col0 + (slot % (G::BN / 4)) * 4, M - 4);
  }

  // Dma::VectorAddress: every source address is clamped per lane, so staging a slab index past the end is harmless
  auto stage = [&](unsigned buf, unsigned k0) {
    char *base = smem + buf * G::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      const float *src;
      if (AT) src = A + (size_t)min(k0 + a_kchunk[i], K - 1) * N + a_row_off[i];
      else src = A + a_row_off[i] + min(k0 + a_kchunk[i] * 4, K - 4);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (wave + G::NW * i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < G::LB; ++i) {
      const unsigned kr = min(k0 + b_krow[i], K - 1);
      const float *src = B + (size_t)kr * M + b_col[i];
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + G::A_BYTES + (wave + G::NW * i) * 1024), 16, 0, 0);
    }
  };

  // Dma::ScalarBase: a slab start past K - BK (beyond the end, or the partial last slab) is clamped to K - BK,
  // uniformly: a partial last slab then sits in the SECOND half of its buffer (see the last-slab loop).
  unsigned voa[G::LA], vob[G::LB];
  if (SDMA) {
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      const unsigned slot = (wave + G::NW * i) * 64 + lane, row = slot / CPR;
      if (AT) voa[i] = a_kchunk[i] * N * 4u + ((unsigned)a_row_off[i] - row0) * 4u;   // K x N: k-row, clamped column
      else voa[i] = (min(row0 + row, N - 1) - row0) * lda * 4u + a_kchunk[i] * 16u;
    }
#pragma unroll
    for (int i = 0; i < G::LB; ++i) vob[i] = b_krow[i] * M * 4u + (b_col[i] - col0) * 4u;
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  auto dma_piece_s = [&](unsigned buf, unsigned k0, int i) {
    const unsigned kc = min(k0, K - BK);
    const bool is_a = i < G::LA;
    const int j = is_a ? i : i - G::LA;
    unsigned long long base = !is_a ? (unsigned long long)(B + (size_t)kc * M + col0)
                              : AT ? (unsigned long long)(A + (size_t)kc * N + row0) : (unsigned long long)(A + (size_t)row0 * lda + kc);
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    const unsigned la = lds_base + buf * G::STAGE_BYTES + (is_a ? 0 : G::A_BYTES) + (wave + G::NW * j) * 1024;
    const unsigned vo = is_a ? voa[j] : vob[j];
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(base), "s"(la) : "memory");
  };
  auto stage_any = [&](unsigned buf, unsigned k0) {
    if constexpr (SDMA) {
#pragma unroll
      for (int i = 0; i < G::LA + G::LB; ++i) dma_piece_s(buf, k0, i);
    } else {
      stage(buf, k0);
    }
  };

  // ---- per-lane fragment addresses (bytes inside a stage) -------------------------------------
  // A: row = wm*TM*32 + mi*32 + lo; chunk = (2*kg + hi) ^ swz(row) = (2*kg) ^ (hi ^ swz(lo))
  const unsigned a_swz = hi ^ ((lo >> G::SWZ_SHIFT) & (CPR - 1));
  const unsigned a_frag_base = AT ? (4 * hi) * (G::BM * 4) + (wm * TM * 32 + TM * lo) * 4  // [k][row], TM rows per lane
                                  : (wm * TM * 32 + lo) * (BK * 4);
  // B: k = kg*8 + p + 4*hi; col = wn*128 + 4*lo
  const unsigned b_frag_base = G::A_BYTES + (4 * hi) * (G::BN * 4) + (wn * 128 + 4 * lo) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[mi][t] = (f32x16)0.0f;
  f32x16 master[TWO_LEVEL ? TM : 1][TWO_LEVEL ? TN : 1];
  if (TWO_LEVEL) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int t = 0; t < TN; ++t) master[mi][t] = (f32x16)0.0f;
  }

  auto load_frags = [&](unsigned buf, int kg, f32x4 (&af)[TM], f32x4 (&bf)[4]) {
    const char *base = smem + buf * G::STAGE_BYTES;
    const unsigned achunk = ((unsigned)(2 * kg) ^ a_swz) * 16;
    if (AT) {
      // lane reads TM consecutive rows of one k-row: row block mi then holds rows TM*i + mi
      using fvec = __attribute__((ext_vector_type(TM))) float;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const fvec v = *(const fvec *)(base + a_frag_base + (kg * 8 + p) * (G::BM * 4));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af[mi][p] = v[mi];
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
        af[mi] = *(const f32x4 *)(base + a_frag_base + mi * 32 * (BK * 4) + achunk);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
      bf[p] = *(const f32x4 *)(base + b_frag_base + (kg * 8 + p) * (G::BN * 4));
  };

  auto mfma_group = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int t = 0; t < TN; ++t)
          acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][p], bf[p][t], acc[mi][t], 0, 0, 0);
  };

  // Interior wavefront blocks (all rows and 128 columns in range): straight-line C (+)= values, BATCH rows in flight at
  // a time.  The predicated forms below cost a branch and a full s_waitcnt per row, which turns the 32 loads of a
  // read-modify-write into 32 dependent round trips.  Row offsets are 32-bit here: `interior32` (a kernel argument,
  // wavefront-uniform) is false for problems whose rows are so long that 64 rows x M x 4 B pass 4 GiB, and those take
  // the predicated forms with 64-bit offsets.
  typedef __attribute__((address_space(1))) f32x4 *gquad_t;
  const bool interior_block = col0 + wn * 128 + 128 <= Mst && row0 + wm * TM * 32 + TM * 32 <= Nst &&
                              (unsigned long long)(TM * 32) * ldc * 4ull < (1ull << 32);  // wavefront-uniform
  auto rmw_interior = [&](bool accumulate, auto batchc, auto value) {
    constexpr int BATCH = decltype(batchc)::value;
    unsigned Mv = ldc;
    asm volatile("" : "+s"(Mv));  // opaque: the row offsets must not be hoisted out of the chunk loop (they would stay
                                  // live across the main loop and cost it registers)
    char *base = (char *)(Cst + (size_t)(row0 + wm * TM * 32) * Mv + col0 + wn * 128);
    const unsigned lane_off = ((AT ? TM * 4 * hi : 4 * hi) * Mv + 4 * lo) * 4u;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += BATCH) {
        f32x4 old[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int r = r0 + j, ri = (r & 3) + 8 * (r >> 2), lrow = AT ? TM * ri + mi : mi * 32 + ri;
          old[j] = accumulate ? *(gquad_t)(base + (lane_off + (unsigned)lrow * Mv * 4u)) : (f32x4)0.0f;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int r = r0 + j, ri = (r & 3) + 8 * (r >> 2), lrow = AT ? TM * ri + mi : mi * 32 + ri;
          f32x4 v;
#pragma unroll
          for (int tt = 0; tt < TN; ++tt) v[tt] = value(mi, tt, r);
          *(gquad_t)(base + (lane_off + (unsigned)lrow * Mv * 4u)) = v + old[j];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // C (+)= accumulators; accumulators = 0   (Chain::FlushIntoC only)
  auto flush_tile = [&](bool accumulate) {
    if (PIPELINED && TM >= 2 && interior_block) {  // (the 32-row wavefront tile measured slower with it: 139.0 vs 141.1 TF)
      rmw_interior(accumulate, std::integral_constant<int, 4>{}, [&](int mi, int tt, int r) { return acc[mi][tt][r]; });
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int tt = 0; tt < TN; ++tt) acc[mi][tt] = (f32x16)0.0f;
      return;
    }
    const unsigned ccol = col0 + wn * 128 + 4 * lo;
    unsigned ldv = ldc;
    asm volatile("" : "+s"(ldv));   // opaque, as in rmw_interior: the 16 x TM row addresses (two registers each) must not be hoisted out
                                    // of the chunk loop -- they would stay live across the main loop; in the VectorAddress twin of the
                                    // 256 x 256 geometry that cost a spill
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned ri = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const unsigned row = row0 + wm * TM * 32 + (AT ? TM * ri + mi : mi * 32 + ri);
        if (ccol < Mst && row < Nst) {
          f32x4 *dst = (f32x4 *)(Cst + (size_t)row * ldv + ccol);
          f32x4 v;
#pragma unroll
          for (int tt = 0; tt < TN; ++tt) v[tt] = acc[mi][tt][r];
          if (accumulate) v += *dst;
          store_quad(dst, v);
        }
      }
#pragma unroll
      for (int tt = 0; tt < TN; ++tt) acc[mi][tt] = (f32x16)0.0f;
    }
  };

  const unsigned num_tiles = (K + BK - 1) / BK;   // slabs, the last one possibly partial
  constexpr int L = G::LA + G::LB;                // DMA instructions per wavefront per slab

  // ---- prologue: fill the whole ring (slabs 0..NS-1), wait for slab 0 ------------------------
  // Staging a slab index past the end is harmless (clamped addresses, a ring slot nobody reads again); this keeps
  // the steady state branch-free and the vmcnt immediates constant.
#pragma unroll
  for (int s = 0; s < NS; ++s) stage_any(s, s * BK);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * L) : "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 af0[TM], bf0[4], af1[TM], bf1[4];
  load_frags(0, 0, af0, bf0);

  // One full slab that HAS a successor.  8-deep k-groups, register double-buffered fragments
  // (set 0 for even groups, set 1 for odd; KG is even so the alternation carries across slabs).
  // The last group's MFMAs are issued after the barrier that publishes slab t+1 and after that
  // slab's first fragment reads, so barrier skew and LDS latency hide under them.
  // p-pairs [2h, 2h+1] of a k-group: the two halves of mfma_group, in the same accumulation order
  auto mfma_half = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[4], int h) {
#pragma unroll
    for (int p = 2 * h; p < 2 * h + 2; ++p)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int t = 0; t < TN; ++t)
          acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][p], bf[p][t], acc[mi][t], 0, 0, 0);
  };
  auto slab = [&](unsigned t) {
    const unsigned buf = t % NS;
#pragma unroll
    for (int kg = 0; kg < G::KG; ++kg) {
      f32x4(&afc)[TM] = (kg & 1) ? af1 : af0;
      f32x4(&bfc)[4] = (kg & 1) ? bf1 : bf0;
      f32x4(&afn)[TM] = (kg & 1) ? af0 : af1;
      f32x4(&bfn)[4] = (kg & 1) ? bf0 : bf1;
      if (PIPELINED) {
        constexpr int NM = 8 * TM, NR = TM + 4;   // MFMAs of a half group, fragment reads of a group
        constexpr bool EARLY = L + NR > NM;       // narrow wavefront tiles: the post-barrier interleave needs the whole group
        static_assert(L + NR <= 2 * NM, "a k-group has too few MFMAs to spread the DMA pieces and reads over");
        __builtin_amdgcn_sched_barrier(0);
        if (kg + 1 < G::KG) {
          mfma_half(afc, bfc, 0);
          __builtin_amdgcn_sched_barrier(0);
          load_frags(buf, kg + 1, afn, bfn);
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(afc, bfc, 1);
        } else {
          if (!EARLY) {
            mfma_half(afc, bfc, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SDMA) {
            // the DMA pieces are inline asm (the scheduler cannot classify them): one MFMA, one piece, by hand; then
            // the reads of the next slab's first group one-per-MFMA as in the builtin form.  EARLY geometries spread
            // this over the whole last group (the barrier came before its first half), the others over its second half.
            constexpr int TOTAL = (EARLY ? 2 : 1) * NM, P0 = EARLY ? 0 : 2;
            auto mfma_one = [&](int idx) {
              const int p = P0 + idx / (TM * TN), mi = (idx / TN) % TM, tt = idx % TN;
              acc[mi][tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afc[mi][p], bfc[p][tt], acc[mi][tt], 0, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < L; ++i) {
              mfma_one(i);
              dma_piece_s(buf, (t + NS) * BK, i);
              __builtin_amdgcn_sched_barrier(0);
            }
            load_frags((t + 1) % NS, 0, afn, bfn);
#pragma unroll
            for (int i = L; i < TOTAL; ++i) mfma_one(i);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // masks: MFMA 0x8, VMEM read 0x20, DS read 0x100
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TOTAL - L - NR, 0);
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          stage(buf, (t + NS) * BK);
          load_frags((t + 1) % NS, 0, afn, bfn);
          if (EARLY) mfma_half(afc, bfc, 0);
          mfma_half(afc, bfc, 1);
#pragma unroll
          for (int i = 0; i < L; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, (EARLY ? 2 : 1) * NM - L - NR, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      // Reads::CompilerPlaced
      if (kg + 1 < G::KG) {
        load_frags(buf, kg + 1, afn, bfn);
      } else {
        // Slab t+1 must have landed.  This wave's LDS reads of slab t are all in registers (lgkmcnt(0)), so after
        // the barrier its ring slot is free for slab t+NS.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        stage(buf, (t + NS) * BK);
        load_frags((t + 1) % NS, 0, afn, bfn);
      }
      mfma_group(afc, bfc);
    }
  };

  const unsigned steady = num_tiles - 1;  // slabs 0 .. num_tiles-2 are full and have a successor
  bool flushed = false;
  if (FLUSH) {
    for (unsigned t0 = 0; t0 < steady; t0 += G::CHUNK) {
      const unsigned tend = min(t0 + (unsigned)G::CHUNK, steady);
      for (unsigned t = t0; t < tend; ++t) slab(t);
      // a further chunk follows: C (+)= acc, restart the chain.  The rule is in k, not in slabs, so that every geometry
      // flushes at the same points whatever its BK: after each 4096 k that more than 32 k follow (32 = the deepest slab)
      if (tend < steady && K > tend * BK + 32) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flush_tile(flushed);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flushed = true;
      }
    }
  } else if (TWO_LEVEL) {
    for (unsigned t0 = 0; t0 < steady; t0 += G::CHUNK) {
      const unsigned tend = min(t0 + (unsigned)G::CHUNK, steady);
      for (unsigned t = t0; t < tend; ++t) slab(t);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int tt = 0; tt < TN; ++tt) {
          master[mi][tt] += acc[mi][tt];
          acc[mi][tt] = (f32x16)0.0f;
        }
    }
  } else {
    for (unsigned t = 0; t < steady; ++t) slab(t);
  }

  // ---- last slab (full or partial): plain group loop, nothing left to prefetch --------------
  {
    const unsigned t = num_tiles - 1;
    const int groups = (int)((K - t * BK) / 8);
    // scalar-base DMA fetched a partial last slab as the LAST BK k of the matrix: its k-groups start further in
    const int shift = SDMA ? G::KG - groups : 0;
    for (int kg = 0; kg < groups; ++kg) {
      load_frags(t % NS, kg + shift, af0, bf0);
      mfma_group(af0, bf0);
    }
  }

  // Trailing (clamped, never read) ring refills may still be in flight: drain them before this
  // wave can retire and its workgroup's LDS allocation can be handed to another workgroup.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- tile write: lane owns 4 consecutive columns of 16 rows per accumulator row-block --------
  auto write_tile = [&](bool accumulate) {
    if (PIPELINED && TM >= 2 && interior_block) {
      rmw_interior(accumulate, std::integral_constant<int, (TM * TN > 8 ? 4 : 8)>{}, [&](int mi, int tt, int r) {
        float x = acc[mi][tt][r];
        if (TWO_LEVEL) x += master[mi][tt][r];
        return x;
      });
      return;
    }
    const unsigned ccol = col0 + wn * 128 + 4 * lo;
    if (ccol < Mst) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned ri = (r & 3) + 8 * (r >> 2) + 4 * hi;
          const unsigned row = row0 + wm * TM * 32 + (AT ? TM * ri + mi : mi * 32 + ri);
          if (row < Nst) {
            f32x4 *dst = (f32x4 *)(Cst + (size_t)row * ldc + ccol);
            f32x4 v;
#pragma unroll
            for (int tt = 0; tt < TN; ++tt) {
              float x = acc[mi][tt][r];
              if (TWO_LEVEL) x += master[mi][tt][r];
              v[tt] = x;
            }
            if (accumulate) v += *dst;
            store_quad(dst, v);
          }
        }
      }
    }
  };
  if (FLUSH && flushed) write_tile(true); else write_tile(false);
}

template <typename G, bool AT>
__global__ __launch_bounds__(G::THREADS, G::MIN_WAVES) void mfma_f32_kernel(const float *__restrict__ A,
                                                              const float *__restrict__ B,
                                                              float *__restrict__ C, unsigned N,
                                                              unsigned K, unsigned M,
                                                              unsigned tiles_n, unsigned tiles_m,
                                                              unsigned kBand, unsigned kChunk, float *__restrict__ partials) {
  // ---- split-K launches (small problems, see mfma_f32_splitk): the grid holds `splits` copies of the tile grid; copy s
  //      multiplies the k range [s * kChunk, min(K, (s+1) * kChunk)) into its own N x M plane of `partials`, and a second
  //      kernel adds the planes in ascending s (deterministic).  lda = the row stride of A, which no longer equals K.
  const unsigned nwg = tiles_n * tiles_m;
  const unsigned lda = K;
  unsigned bid = blockIdx.x;
  if (kChunk) {   // wavefront-uniform
    const unsigned split = bid / nwg;
    bid -= split * nwg;
    const unsigned kbeg = split * kChunk;
    A += AT ? (size_t)kbeg * N : (size_t)kbeg;
    B += (size_t)kbeg * M;
    C = partials + (size_t)split * N * M;
    K = min(kChunk, K - kbeg);
  }
  // ---- workgroup -> output tile: XCD-contiguous chunks, then bands of kBand tile-rows ------------
  const unsigned lin = xcd_remap(bid, nwg);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned tile_row = band * kBand + within % rows_in_band;
  const unsigned tile_col = within / rows_in_band;
  tile_body<G, AT>(A, B, C, N, K, M, lda, M, N, M, tile_row * G::BM, tile_col * G::BN);
}

// ---- small problems (round 3): 64 x 64 tiles, one 32 x 32 accumulator per wavefront -------------------------------------
// Below a round of 128 x 128 tiles the chip has more SIMDs than the problem has wavefront tiles of 32 x 128 (1024^3: 256 for
// 1024 SIMDs), which split-K answers with partial planes and a second kernel.  This geometry instead makes the wavefront
// tile 32 x 32 -- a workgroup of 2 x 2 wavefronts owns 64 x 64 of C, two workgroups per CU (64 KiB of LDS each) --
// so that 1024^3 is 256 workgroups of whole K and no reduction.  The price: twice the L2 -> LDS bytes per flop of the
// 128 x 128 tile, and every MFMA of a wavefront accumulates into the same registers (a dependent chain: 94 % of the
// matrix-core rate with one wavefront per SIMD, 100 % with two -- tools/probes/probe_mfma_chain.hip).
// Arithmetic: per output element exactly the chain of tile_body -- k-groups of 8 ascending, inside a group the MFMA
// pairs (p, p + 4), accumulators flushed into C after every 4096 k that more than 32 k follow -- so the result has the
// bits of the other shipped geometries (tested).  Row-major A only.
//   A slab in LDS: [64 rows][8 chunks of 4 k], chunk index XOR (row >> 1) & 7 (the 16 lanes of a ds_read_b128 service
//     group -- 16 rows, one logical chunk -- hit 16 distinct 16-B slots of the 256-B bank row);
//   B slab in LDS: [32 k][16 chunks of 4 columns], chunk index XOR 8 for k & 4: the MFMA's B operand is one float per
//     lane, B[k = p + 4 * (lane >> 5)][column lane & 31], read with ds_read_b32 -- lanes 0-31 take 32 consecutive dwords
//     of k-row p, lanes 32-63 the same columns of k-row p + 4, which the XOR moves to the other 32 banks.
//   (Both enumerated in tests/test_layouts.py.)  Ring of four stages, one barrier per slab, the order of MFMAs, DMA pieces and
//   fragment reads pinned (below).
template <int NS_>
struct SmallT {
  static constexpr int BM = 64, BN = 64, BK = 32, NS = NS_, NW = 4, THREADS = 256;
  static constexpr int A_BYTES = BM * BK * 4, B_BYTES = BK * BN * 4, STAGE_BYTES = A_BYTES + B_BYTES, LDS_BYTES = NS * STAGE_BYTES;
  static constexpr int PER_CU = 160 * 1024 / LDS_BYTES > 3 ? 3 : 160 * 1024 / LDS_BYTES;
  static constexpr int CHUNK = 128;   // slabs per link of the chain: 4096 k
  static constexpr int L = 4;         // DMA instructions per wavefront per slab: 2 of A, 2 of B
};

using Small = SmallT<4>;

template <typename G>
__global__ __launch_bounds__(G::THREADS, G::PER_CU) void mfma_f32_small_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, unsigned N, unsigned K, unsigned M,
    unsigned tiles_n, unsigned tiles_m, unsigned kBand, unsigned kChunk, float *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned nwg = tiles_n * tiles_m;
  const unsigned lda = K;
  unsigned bid = blockIdx.x;
  if (kChunk) {   // split-K (few tiles, long K), as in mfma_f32_kernel: copy `split` of the tile grid multiplies its k range into its own plane
    const unsigned split = bid / nwg, kbeg = split * kChunk;
    bid -= split * nwg;
    A += kbeg;
    B += (size_t)kbeg * M;
    C = partials + (size_t)split * N * M;
    K = min(kChunk, K - kbeg);
  }
  const unsigned lin = xcd_remap(bid, nwg);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wm = wave >> 1, wn = wave & 1u, lo = lane & 31u, hi = lane >> 5;

  // per-lane DMA sources (clamped: a slab index past the end, a row past N or a column chunk past M fetch valid bytes
  // that are never multiplied into a stored element)
  size_t a_row_off[2];
  unsigned a_k[2], b_k[2], b_col[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned slot = (wave + 4 * i) * 64 + lane;
    const unsigned row = slot >> 3, pc = slot & 7u;
    a_k[i] = (pc ^ ((row >> 1) & 7u)) * 4;
    a_row_off[i] = (size_t)min(row0 + row, N - 1) * lda;
    const unsigned k = slot >> 4, pcb = slot & 15u;
    b_k[i] = k;
    b_col[i] = min(col0 + (pcb ^ ((k & 4u) << 1)) * 4, M - 4);
  }
  auto stage = [&](unsigned buf, unsigned k0) {
    char *base = smem + buf * G::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(A + a_row_off[i] + min(k0 + a_k[i], K - 4)), (lptr_t)(base + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(B + (size_t)min(k0 + b_k[i], K - 1) * M + b_col[i]),
                                       (lptr_t)(base + G::A_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
  };

  // Steady state (K >= 32): the same pieces with a wavefront-uniform 64-bit base in SGPRs and a constant 32-bit per-lane
  // offset (global_load_lds_dwordx4 v, s[base:base+1]) -- no address arithmetic on the vector ALU.  A slab start past K - 32
  // (beyond the end, or the partial last slab) is clamped to K - 32, uniformly: a partial last slab then sits at the END
  // of its stage (see the last-slab loop).  64 rows x K x 4 B and 32 k-rows x M x 4 B stay below 4 GiB (launcher).
  unsigned voa[2], vob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned slot = (wave + 4 * i) * 64 + lane;
    voa[i] = (min(row0 + (slot >> 3), N - 1) - row0) * lda * 4u + a_k[i] * 4u;
    vob[i] = b_k[i] * M * 4u + (b_col[i] - col0) * 4u;
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  auto dma_piece = [&](unsigned buf, unsigned k0, int i) {   // i = 0, 1: A; 2, 3: B
    const unsigned kc = min(k0, K - G::BK);
    const bool is_a = i < 2;
    const int j = is_a ? i : i - 2;
    unsigned long long base = is_a ? (unsigned long long)(A + (size_t)row0 * lda + kc) : (unsigned long long)(B + (size_t)kc * M + col0);
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    const unsigned la = lds_base + buf * G::STAGE_BYTES + (is_a ? 0 : G::A_BYTES) + (wave + 4 * j) * 1024;
    const unsigned vo = is_a ? voa[j] : vob[j];
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(base), "s"(la) : "memory");
  };

  // per-lane fragment addresses inside a stage
  const unsigned a_row = wm * 32 + lo;
  const unsigned a_base = a_row * (G::BK * 4), a_swz = hi ^ ((a_row >> 1) & 7u);               // chunk (2 kg + hi) ^ swz(row)
  const unsigned b_base = G::A_BYTES + (4 * hi) * (G::BN * 4) + ((((wn * 32 + lo) >> 2) ^ (8 * hi)) * 16) + (lo & 3u) * 4;

  f32x16 acc = (f32x16)0.0f;
  struct Frags { f32x4 a; float b[4]; };
  auto read_group = [&](unsigned buf, int kg, Frags &f) {
    const char *base = smem + buf * G::STAGE_BYTES;
    f.a = *(const f32x4 *)(base + a_base + (((unsigned)(2 * kg) ^ a_swz) * 16));
#pragma unroll
    for (int p = 0; p < 4; ++p) f.b[p] = *(const float *)(base + b_base + (kg * 8 + p) * (G::BN * 4));
  };
  auto mfma_group = [&](const Frags &f) {
#pragma unroll
    for (int p = 0; p < 4; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[p], f.b[p], acc, 0, 0, 0);
  };
  // C (+)= accumulator: lane owns column lo of 16 rows
  auto write_tile = [&](bool accumulate) {
    const unsigned col = col0 + wn * 32 + lo;
    if (col < M) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < N) {
          float *dst = C + (size_t)row * M + col;
          *dst = accumulate ? acc[r] + *dst : acc[r];
        }
      }
    }
  };

  const unsigned num_tiles = (K + G::BK - 1) / G::BK, steady = num_tiles - 1;
  const bool short_k = K < G::BK;   // a single partial slab: the per-lane clamped form, at the start of stage 0
  // prologue: slabs 0 .. NS-2 (slab NS-1 is requested piece by piece during slab 0); wait for slab 0
  if (short_k) {
#pragma unroll
    for (int s = 0; s < G::NS - 1; ++s) stage(s, s * G::BK);
  } else {
#pragma unroll
    for (int s = 0; s < G::NS - 1; ++s)
#pragma unroll
      for (int i = 0; i < G::L; ++i) dma_piece(s, s * G::BK, i);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::NS - 2) * G::L) : "memory");
  __builtin_amdgcn_s_barrier();

  // One slab that has a successor.  The wavefront's MFMAs are one dependent chain: the second MFMA of a group cannot issue
  // before the first has gone through the pipe, so that is where everything else goes.  After the first MFMA of group g:
  // DMA piece g of slab t+NS-1 (into the stage slab t-1 left at the last hand-over) and the fragment reads of the NEXT
  // group (by that group's first MFMA they are three MFMAs old: the `s_waitcnt lgkmcnt(0)` the compiler puts there is
  // free).  The slab hand-over -- wait for slab t+1 (all but the pieces of the NS-2 slabs requested after it), barrier,
  // first fragments of slab t+1 -- sits in the same place of the fourth group.
  Frags f0, f1;
  read_group(0, 0, f0);
  unsigned buf = 0;
  auto group = [&](const Frags &cur, auto &&meanwhile) {
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0], cur.b[0], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    meanwhile();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 1; p < 4; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[p], cur.b[p], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto slab = [&](unsigned t) {
    const unsigned next = (buf + 1) % G::NS, refill = (buf + G::NS - 1) % G::NS, kr = (t + G::NS - 1) * G::BK;
    group(f0, [&] { dma_piece(refill, kr, 0); read_group(buf, 1, f1); });
    group(f1, [&] { dma_piece(refill, kr, 1); read_group(buf, 2, f0); });
    group(f0, [&] { dma_piece(refill, kr, 2); read_group(buf, 3, f1); });
    group(f1, [&] {
      dma_piece(refill, kr, 3);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G::NS - 2) * G::L) : "memory");   // slab t+1 landed; this stage read out
      __builtin_amdgcn_s_barrier();
      read_group(next, 0, f0);
    });
    buf = next;
  };
  bool flushed = false;
  for (unsigned t0 = 0; t0 < steady; t0 += G::CHUNK) {
    const unsigned tend = min(t0 + (unsigned)G::CHUNK, steady);
    for (unsigned t = t0; t < tend; ++t) slab(t);
    if (tend < steady) {   // the flush rule of tile_body (with BK = 32, "more than 32 k follow" == a further steady slab)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      write_tile(flushed);
      acc = (f32x16)0.0f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flushed = true;
    }
  }
  {
    const int groups = (int)((K - steady * G::BK) / 8);   // the last slab, full or partial
    const int shift = short_k ? 0 : 4 - groups;            // fetched as the LAST 32 k of the matrix: its groups start further in
    for (int kg = 0; kg < groups; ++kg) {
      read_group(buf, kg + shift, f0);
      mfma_group(f0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // trailing (clamped, never read) refills
  write_tile(flushed);
}

// (launch_small: after the split-K reduce kernel, which it shares with launch_geo)

// ---- stream-K (round 3): problems of a few partial rounds of tiles ------------------------------------------------
// A launch runs in whole rounds of resident workgroups, so 2560^3 (400 tiles of 128 x 128 for 512 slots) or 3072^3 (576
// tiles: one full round and an eighth) leave a large part of the chip idle in their last round.  Here the job's
// (tile, 32-deep slab) units are dealt out to `gridDim.x` persistent workgroups in equal contiguous ranges; a workgroup
// walks its range tile by tile.  A tile whose whole K falls into one range is written to C directly; the head and tail
// tiles of a range are written to the workgroup's two 128 x 128 scratch slots, and a second kernel adds the slots of every
// split tile in ascending k -- no atomics, same bits run to run.  Needs K % BK == 0 (whole slabs) and row-major A.
__device__ __forceinline__ unsigned sk_range_begin(unsigned long long units, unsigned w, unsigned nwg) {
  return (unsigned)(units * w / nwg);
}

template <typename G>
__global__ __launch_bounds__(G::THREADS, G::MIN_WAVES) void mfma_f32_streamk_kernel(const float *__restrict__ A,
                                                                                      const float *__restrict__ B,
                                                                                      float *__restrict__ C, unsigned N, unsigned K,
                                                                                      unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                                      unsigned kBand, unsigned spt,
                                                                                      float *__restrict__ slots) {
  const unsigned nwg = gridDim.x;
  const unsigned w = xcd_remap(blockIdx.x, nwg);   // an XCD's workgroups own one contiguous stretch of the unit order
  const unsigned long long units = (unsigned long long)tiles_n * tiles_m * spt;
  const unsigned u0 = sk_range_begin(units, w, nwg), u1 = sk_range_begin(units, w + 1, nwg);
  const unsigned first_tile = u0 / spt;
  for (unsigned u = u0; u < u1;) {
    const unsigned tile = u / spt, s0 = u - tile * spt, s1 = min(spt, s0 + (u1 - u));
    const unsigned band = tile / (kBand * tiles_m), within = tile % (kBand * tiles_m);
    const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
    const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;
    const unsigned kbeg = s0 * G::BK, klen = (s1 - s0) * G::BK;
    const float *a = A + kbeg, *b = B + (size_t)kbeg * M;
    if (s0 == 0 && s1 == spt) {
      tile_body<G, false>(a, b, C, N, klen, M, K, M, N, M, row0, col0);
    } else {   // head (slot 0) or tail (slot 1) of this workgroup's range: into its scratch slot, addressed like a C with ldc = BN
      float *slot = slots + ((size_t)w * 2 + (tile != first_tile)) * (G::BM * G::BN);
      float *base = (float *)((unsigned long long)slot - ((unsigned long long)row0 * G::BN + col0) * sizeof(float));
      tile_body<G, false>(a, b, base, N, klen, M, K, G::BN, ~0u, ~0u, row0, col0);
    }
    u += s1 - s0;
    __syncthreads();   // the next segment's prologue overwrites the LDS ring
  }
}

// Stream-K in TEAMS (what the shape-adaptive rule and f32_splitk = 0 run; 9 = the single-range form above, kept as an
// independent cross-check).  Same idea of unit ranges, two changes.
//
// Teams.  With one range per workgroup, neighbouring workgroups sit at different k offsets of neighbouring tiles, so no two
// of them ever want the same A or B slab at the same time: the XCD's L2 shares nothing and the launch becomes fabric-bound
// (3968^3: 3.9 GB fetched against 0.93 GB for whole tiles, 133 TF; 4096^3, where the ranges happen to be whole tiles: 0.80
// GB, 146 TF -- profiles/r03w).  So the ranges go to TEAMS of sr x sc workgroups inside one XCD: the tile grid is cut into
// super-tiles of sr x sc tiles, a team walks its range of (super-tile, slab) units with workgroup `lane` of the team on
// tile `lane` of the super-tile, and all of a team are at the same slab at the same time -- each A slab is wanted by sc
// of them and each B slab by sr.  Per lane this is exactly the scheme above with `teams` ranges.  Tiles of a ragged
// super-tile that fall outside the matrix are skipped by their lane (the same lane in every team).
//
// The combine.  Every segment of a tile other than its lowest-k one is the FIRST segment of the workgroup that owns it (a
// range enters a tile from below only at its own start), and the lowest-k segment is the LAST thing its owner does.  A cut
// tile is finished as C = part(t0) + part(t0 + 1) + ... in ascending k, a fixed order, no atomics on data, in one of two
// ways that perform the same additions in the same order (bit-identical; tests compare them on every stream-K shape):
//
// Combine::LastArriver (what MM_PATH_AUTO runs): ONE kernel in which nobody waits.  Every part of a cut tile -- the lowest-k
// one included -- goes to a scratch slot (agent-scope stores: the reader may sit on another XCD) and raises its flag; right
// after raising it, the part's workgroup LOOKS at the flags of the tile's other parts (no loop): whoever finds them all
// raised is (one of) the last to arrive and adds the slots in ascending k into C.  Raising and looking are ordered like a
// sequentially consistent store and load (sc1 store, s_waitcnt vmcnt(0), sc1 loads: how this target implements seq_cst at
// agent scope), so of two parts finishing at the same moment at least one sees the other; if both do, both write the same
// bits to C -- the gather reads slots only, never C, so it is idempotent (which is why the lowest-k part may NOT live in C:
// a second gatherer would read the first one's sum).  No workgroup ever depends on another one being resident, so the form
// is sound next to any other launch of any process, on CU-masked streams, on partitions and in graphs.  A workgroup owns
// two slots and two flags: index 2w for its first segment (a higher part of some tile), 2w + 1 for its last one (the
// lowest-k part of another tile).  A flag is raised when it holds the launch's epoch (flags_alloc, mm_capi.hip): nothing is
// cleared between launches.  Visibility: slot stores sc1 + s_waitcnt vmcnt(0) + __syncthreads before the flag; on the
// gathering side an agent-scope acquire (buffer_inv sc1) + __syncthreads before plain loads (MI355X_MICROARCH.md,
// inter-workgroup visibility: the valid forms); tests/test_isa_contract.py pins the machine code.
//
// Combine::FixupKernel (f32_splitk 11, the cross-check): the lowest-k part of a cut tile goes to C, the other parts to their
// slots (plain stores, no flags), and streamk_teams_fixup_kernel below adds the slots on top of C.  Two kernels, no
// inter-workgroup communication at all; 0-3.5 % slower (profiles/r05f_*).
//
// (Rounds 3-4 ran a third form in which the owner of the lowest-k part WAITED inside the launch for the others' flags.  It
// needed every workgroup of the launch -- and of every other launch of its kind, in any process -- to be resident at once,
// which a library cannot guarantee; the last-arriver form is as fast without waiting, so the waiting form is retired:
// HISTORY.md, round 5.)
//
// Combine::Ticket (f32_splitk 12; round 6): the last-arriver idea written in the LANGUAGE's memory model instead of
// this target's -- the canonical last-block pattern.  Every part stores its slot with plain stores, every thread issues an
// agent-scope release fence, the workgroup meets at a barrier, and one thread takes a TICKET with a single agent-scope acq_rel
// read-modify-write on a COUNTER of the tile (not on data): the part that draws ticket parts - 1 knows -- by release / acquire
// through the counter's modification order -- that every other part's slot is visible to it, and gathers in ascending k.  Exactly
// one gatherer, no flag loads, no inline-asm stores, no s_waitcnt contract; the same additions in the same order, hence the same
// bits.  The counter word carries the launch's epoch in its upper 56 bits (a compare-exchange loop, lock-free: a failed exchange
// means another part got its ticket), so nothing is cleared between launches here either.
enum class Combine { FixupKernel, LastArriver, Ticket };

template <typename G, Combine MODE>
__global__ __launch_bounds__(G::THREADS, G::MIN_WAVES) void mfma_f32_streamk_teams_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, unsigned N, unsigned K, unsigned M,
    unsigned tiles_n, unsigned tiles_m, unsigned spt, unsigned sr, unsigned sc, unsigned teams_per_xcd, unsigned teams,
    float *__restrict__ slots, unsigned long long *__restrict__ flags, unsigned long long epoch) {
  __shared__ unsigned gather_here;   // LastArriver: thread 0's verdict for the workgroup
  const unsigned lanes = sr * sc, xcd = blockIdx.x % 8, place = blockIdx.x / 8;
  const unsigned team_in_xcd = place / lanes, lane = place % lanes;
  const unsigned team = team_in_xcd * 8 + xcd, w = team * lanes + lane;   // consecutive teams on consecutive XCDs: fewer teams than places still use all eight
  if (team_in_xcd >= teams_per_xcd || team >= teams) return;     // teams <= 8 x teams_per_xcd, <= units: no empty range
  const unsigned st_rows = (tiles_n + sr - 1) / sr, st_cols = (tiles_m + sc - 1) / sc;
  const unsigned long long units = (unsigned long long)st_rows * st_cols * spt;   // per lane; teams <= units
  const unsigned u0 = sk_range_begin(units, team, teams), u1 = sk_range_begin(units, team + 1, teams);
  for (unsigned u = u0; u < u1;) {
    const unsigned st = u / spt, s0 = u - st * spt, s1 = min(spt, s0 + (u1 - u));
    const unsigned tile_r = (st % st_rows) * sr + lane % sr, tile_c = (st / st_rows) * sc + lane / sr;
    u += s1 - s0;
    if (tile_r >= tiles_n || tile_c >= tiles_m) continue;        // uniform over the workgroup
    const unsigned row0 = tile_r * G::BM, col0 = tile_c * G::BN;
    const unsigned kbeg = s0 * G::BK, klen = (s1 - s0) * G::BK;
    const float *a = A + kbeg, *b = B + (size_t)kbeg * M;
    if (s0 == 0 && s1 == spt) {
      tile_body<G, false>(a, b, C, N, klen, M, K, M, N, M, row0, col0);
    } else if constexpr (MODE == Combine::Ticket) {
      const unsigned u_lo = st * spt, u_hi = u_lo + spt;
      unsigned t0 = team;
      while (t0 > 0 && sk_range_begin(units, t0, teams) > u_lo) --t0;
      unsigned t_end = team + 1;
      while (t_end < teams && sk_range_begin(units, t_end, teams) < u_hi) ++t_end;
      auto part_index = [&](unsigned o) { return 2u * (o * lanes + lane) + (o == t0 ? 1u : 0u); };
      float *slot = slots + (size_t)part_index(team) * (G::BM * G::BN);
      float *base = (float *)((unsigned long long)slot - ((unsigned long long)row0 * G::BN + col0) * sizeof(float));
      tile_body<G, false>(a, b, base, N, klen, M, K, G::BN, ~0u, ~0u, row0, col0);   // plain stores
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // every thread: its slot stores are ordered before whatever follows the barrier
      __syncthreads();
      if (threadIdx.x == 0) {
        // the tile's counter = the flag word of its lowest-k part (one cut tile per workgroup ends there): (epoch tag << 8) | parts arrived
        unsigned long long *counter = flags + part_index(t0);
        const unsigned long long tag = (epoch << 8);
        unsigned long long seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
        do {
          want = ((seen ^ tag) >> 8) == 0 ? seen + 1 : (tag | 1ull);
        } while (!__hip_atomic_compare_exchange_strong(counter, &seen, want, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        gather_here = (unsigned)(want & 0xffu) == t_end - t0;   // the last ticket of the tile
      }
      __syncthreads();
      if (gather_here) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        constexpr int QUADS = G::BM * G::BN / 4 / G::THREADS;
        f32x4 acc[QUADS];
        const f32x4 *first = (const f32x4 *)(slots + (size_t)part_index(t0) * (G::BM * G::BN));
#pragma unroll
        for (int i = 0; i < QUADS; ++i) acc[i] = first[i * G::THREADS + threadIdx.x];
        for (unsigned o = t0 + 1; o < t_end; ++o) {
          const f32x4 *src = (const f32x4 *)(slots + (size_t)part_index(o) * (G::BM * G::BN));
#pragma unroll
          for (int i = 0; i < QUADS; ++i) acc[i] += src[i * G::THREADS + threadIdx.x];
        }
#pragma unroll
        for (int i = 0; i < QUADS; ++i) {
          const unsigned q = i * G::THREADS + threadIdx.x, row = row0 + q / (G::BN / 4), col = col0 + (q % (G::BN / 4)) * 4;
          if (row < N && col < M) *(f32x4 *)(C + (size_t)row * M + col) = acc[i];
        }
      }
    } else if constexpr (MODE == Combine::LastArriver) {
      // a part of a cut tile.  The tile's parts belong to teams t0 (lowest k: that team's LAST segment, slot / flag 2w' + 1)
      // and t0 + 1 .. t_end - 1 (their FIRST segments, slot / flag 2w'), same lane.
      const unsigned u_lo = st * spt, u_hi = u_lo + spt;
      unsigned t0 = team;
      while (t0 > 0 && sk_range_begin(units, t0, teams) > u_lo) --t0;
      unsigned t_end = team + 1;
      while (t_end < teams && sk_range_begin(units, t_end, teams) < u_hi) ++t_end;
      auto part_index = [&](unsigned o) { return 2u * (o * lanes + lane) + (o == t0 ? 1u : 0u); };
      const unsigned mine = part_index(team);
      float *slot = slots + (size_t)mine * (G::BM * G::BN);
      float *base = (float *)((unsigned long long)slot - ((unsigned long long)row0 * G::BN + col0) * sizeof(float));
      tile_body<G, false, true>(a, b, base, N, klen, M, K, G::BN, ~0u, ~0u, row0, col0);   // slot stores at agent scope
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wavefront: its slot stores have reached the agent's coherence point
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(flags + mine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // raised = holds this launch's epoch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the flag is out before the others' flags are looked at (store -> load order)
        unsigned all = 1;
        for (unsigned o = t0; o < t_end; ++o)
          if (o != team && __hip_atomic_load(flags + part_index(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) all = 0;
        if (all) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        gather_here = all;
      }
      __syncthreads();
      if (gather_here) {   // (one of) the last to arrive: C = slot(t0) + slot(t0 + 1) + ... in ascending k, from slots only
        // one slot at a time, all of a thread's quads in flight together: a slot costs one trip to memory, not QUADS of them
        constexpr int QUADS = G::BM * G::BN / 4 / G::THREADS;
        f32x4 acc[QUADS];
        const f32x4 *first = (const f32x4 *)(slots + (size_t)part_index(t0) * (G::BM * G::BN));
#pragma unroll
        for (int i = 0; i < QUADS; ++i) acc[i] = first[i * G::THREADS + threadIdx.x];
        for (unsigned o = t0 + 1; o < t_end; ++o) {
          const f32x4 *src = (const f32x4 *)(slots + (size_t)part_index(o) * (G::BM * G::BN));
#pragma unroll
          for (int i = 0; i < QUADS; ++i) acc[i] += src[i * G::THREADS + threadIdx.x];
        }
#pragma unroll
        for (int i = 0; i < QUADS; ++i) {
          const unsigned q = i * G::THREADS + threadIdx.x, row = row0 + q / (G::BN / 4), col = col0 + (q % (G::BN / 4)) * 4;
          if (row < N && col < M) *(f32x4 *)(C + (size_t)row * M + col) = acc[i];
        }
      }
    } else if (s0 > 0) {   // FixupKernel, not the lowest-k segment: this workgroup's first segment -> its slot
      float *slot = slots + (size_t)w * (G::BM * G::BN);
      float *base = (float *)((unsigned long long)slot - ((unsigned long long)row0 * G::BN + col0) * sizeof(float));
      tile_body<G, false>(a, b, base, N, klen, M, K, G::BN, ~0u, ~0u, row0, col0);
    } else {               // FixupKernel, the lowest-k segment of a split tile: into C; the fix-up kernel adds the others
      tile_body<G, false>(a, b, C, N, klen, M, K, M, N, M, row0, col0);
    }
    __syncthreads();
  }
}

// Team shape for a tile grid: 4, 2 or 1 tiles a side, the largest that divides the grid exactly -- a ragged super-tile idles
// its outside lanes for whole segments and costs more than the sharing returns (profiles/r03w_f32_streamk_teams.txt: 2816^3,
// 22 x 22 tiles, 4 x 4 teams 113 TF against 128; 3584^3, 28 x 28, 140 against 137; 5120^3 146 against 142).  Wider teams
// (8 x 4, 8 x 8) measured no better than 4 x 4 where they divide.
struct TeamShape { unsigned sr, sc; };
static TeamShape streamk_team_shape(unsigned tiles_n, unsigned tiles_m) {
  auto side = [](unsigned t) { return t % 4 == 0 ? 4u : t % 2 == 0 ? 2u : 1u; };
  return {side(tiles_n), side(tiles_m)};
}

// The second kernel of the two-kernel teams form: one workgroup per tile of C.  A tile that one team's range holds whole
// was finished by the main kernel; for a cut tile C already holds the lowest-k part, and the slots of the teams that own
// the other parts (the next teams in range order, same lane) are added on top in ascending k -- the additions of the
// last-arriver form's gather, in its order, by its thread mapping.
template <int BM, int BN, int THREADS>
__global__ __launch_bounds__(THREADS) void streamk_teams_fixup_kernel(const float *__restrict__ slots, float *__restrict__ C, unsigned N,
                                                                       unsigned M, unsigned tiles_n, unsigned tiles_m, unsigned spt,
                                                                       unsigned sr, unsigned sc, unsigned teams) {
  const unsigned tile_r = blockIdx.x % tiles_n, tile_c = blockIdx.x / tiles_n;
  const unsigned lanes = sr * sc, st_rows = (tiles_n + sr - 1) / sr, st_cols = (tiles_m + sc - 1) / sc;
  const unsigned st = (tile_c / sc) * st_rows + tile_r / sr, lane = tile_r % sr + (tile_c % sc) * sr;
  const unsigned long long units = (unsigned long long)st_rows * st_cols * spt;
  const unsigned u_lo = st * spt, u_hi = u_lo + spt;
  unsigned t0 = (unsigned)((unsigned long long)u_lo * teams / units);     // the team whose range holds the tile's first unit
  while (t0 + 1 < teams && sk_range_begin(units, t0 + 1, teams) <= u_lo) ++t0;
  while (t0 > 0 && sk_range_begin(units, t0, teams) > u_lo) --t0;
  unsigned t_end = t0 + 1;
  while (t_end < teams && sk_range_begin(units, t_end, teams) < u_hi) ++t_end;
  if (t_end == t0 + 1) return;                                            // one range holds the whole tile
  const unsigned row0 = tile_r * BM, col0 = tile_c * BN;
  constexpr int QUADS = BM * BN / 4 / THREADS;
  f32x4 acc[QUADS];
#pragma unroll
  for (int i = 0; i < QUADS; ++i) {
    const unsigned q = i * THREADS + threadIdx.x, row = row0 + q / (BN / 4), col = col0 + (q % (BN / 4)) * 4;
    acc[i] = (row < N && col < M) ? *(const f32x4 *)(C + (size_t)row * M + col) : (f32x4)0.0f;
  }
  for (unsigned o = t0 + 1; o < t_end; ++o) {
    const f32x4 *src = (const f32x4 *)(slots + (size_t)(o * lanes + lane) * (BM * BN));
#pragma unroll
    for (int i = 0; i < QUADS; ++i) acc[i] += src[i * THREADS + threadIdx.x];
  }
#pragma unroll
  for (int i = 0; i < QUADS; ++i) {
    const unsigned q = i * THREADS + threadIdx.x, row = row0 + q / (BN / 4), col = col0 + (q % (BN / 4)) * 4;
    if (row < N && col < M) *(f32x4 *)(C + (size_t)row * M + col) = acc[i];
  }
}

// How a problem is dealt out to teams: shared by both forms, so that they cut the same tiles at the same k.
struct TeamPlan { unsigned tiles_n, tiles_m, spt, sr, sc, teams_per_xcd, teams; };
template <typename G>
static TeamPlan streamk_team_plan(const Problem &p) {
  TeamPlan t;
  t.tiles_n = (p.n + G::BM - 1) / G::BM; t.tiles_m = (p.m + G::BN - 1) / G::BN; t.spt = p.k / G::BK;
  const TeamShape ts = streamk_team_shape(t.tiles_n, t.tiles_m);
  t.sr = ts.sr; t.sc = ts.sc;
  t.teams_per_xcd = 64 / (ts.sr * ts.sc);   // 8 XCDs x 64 places
  const unsigned long long super_tiles = (unsigned long long)(t.tiles_n / ts.sr) * (t.tiles_m / ts.sc), units = super_tiles * t.spt;
  // every place gets a team when there is work for it: a team's range is at least 8 slabs (below that the fill and drain
  // of the rings outweigh the slabs -- the chunk floor of split-K) and a tile is cut at most 8 ways (the gather is serial
  // in whoever does it, and a part looks at no more than 7 siblings' flags)
  t.teams = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>({8ull * t.teams_per_xcd, units / 8, 8 * super_tiles}));
  return t;
}

// The two-kernel teams form (f32_splitk 11; the cross-check of the default): teams, slots, and the fix-up kernel.  No
// workgroup waits for another one either; the bits of the default form.
template <typename G>
int launch_streamk_teams(hipStream_t s, const Problem &p) {
  const TeamPlan t = streamk_team_plan<G>(p);
  const unsigned nwg = 512;
  static_assert(G::MIN_WAVES == 2, "two workgroups per CU: 64 places per XCD");
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_streamk_teams_kernel<G, Combine::FixupKernel>, G::LDS_BYTES, configured)) return e;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipMemPool_t pool = nullptr;
  if (int rc = workspace_pool(dev, &pool)) return rc;
  float *slots = nullptr;
  const size_t slot_bytes = (size_t)nwg * G::BM * G::BN * sizeof(float);
  if ((e = hipMallocFromPoolAsync((void **)&slots, slot_bytes, pool, s)) != hipSuccess) return (int)e;
  // debug_poison: NaN in every slot AND in C (pure output: a tile nobody finished then shows as NaN instead of as whatever the buffer held)
  if (tuning(TUNE_DEBUG_POISON) == 1 && ((e = hipMemsetAsync(slots, 0xFF, slot_bytes, s)) != hipSuccess ||
                                         (e = hipMemsetAsync(p.c, 0xFF, (size_t)p.n * p.m * sizeof(float), s)) != hipSuccess)) { (void)hipFreeAsync(slots, s); return (int)e; }
  (void)hipGetLastError();
  hipLaunchKernelGGL((mfma_f32_streamk_teams_kernel<G, Combine::FixupKernel>), dim3(nwg), dim3(G::THREADS), G::LDS_BYTES, s, (const float *)p.a,
                     (const float *)p.b, (float *)p.c, p.n, p.k, p.m, t.tiles_n, t.tiles_m, t.spt, t.sr, t.sc, t.teams_per_xcd, t.teams,
                     slots, (unsigned long long *)nullptr, 0ull);
  int rc = (int)hipGetLastError();
  if (rc == 0 && t.teams > 1) {
    hipLaunchKernelGGL((streamk_teams_fixup_kernel<G::BM, G::BN, G::THREADS>), dim3(t.tiles_n * t.tiles_m), dim3(G::THREADS), 0, s,
                       (const float *)slots, (float *)p.c, p.n, p.m, t.tiles_n, t.tiles_m, t.spt, t.sr, t.sc, t.teams);
    rc = (int)hipGetLastError();
  }
  const hipError_t f = hipFreeAsync(slots, s);
  return rc ? rc : (int)f;
}

// Stream-K as MM_PATH_AUTO runs it (f32_splitk 0): teams, two slots and two flags per workgroup, the last part to arrive
// gathers -- one kernel, no workgroup waits for another one (Combine::LastArriver above).  Flags hold the launch's epoch
// (flags_alloc, mm_capi.hip), so nothing is cleared between launches.
template <typename G, Combine MODE = Combine::LastArriver>
int launch_streamk_arrive(hipStream_t s, const Problem &p) {
  const TeamPlan t = streamk_team_plan<G>(p);
  const unsigned nwg = 512;
  static_assert(G::MIN_WAVES == 2, "two workgroups per CU: 64 places per XCD");
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_streamk_teams_kernel<G, MODE>, G::LDS_BYTES, configured)) return e;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipMemPool_t pool = nullptr;
  if (int rc = workspace_pool(dev, &pool)) return rc;
  float *slots = nullptr;
  void *flags = nullptr;
  unsigned long long epoch = 0;
  const size_t slot_bytes = (size_t)2 * nwg * G::BM * G::BN * sizeof(float);
  int rc = (int)hipMallocFromPoolAsync((void **)&slots, slot_bytes, pool, s);
  if (rc == 0 && tuning(TUNE_DEBUG_POISON) == 1) rc = (int)hipMemsetAsync(slots, 0xFF, slot_bytes, s);   // debug_poison: NaN in every slot ...
  if (rc == 0 && tuning(TUNE_DEBUG_POISON) == 1) rc = (int)hipMemsetAsync(p.c, 0xFF, (size_t)p.n * p.m * sizeof(float), s);   // ... and in C (pure output): an unfinished tile shows
  if (rc == 0) rc = flags_alloc(dev, s, (size_t)2 * nwg * sizeof(unsigned long long), &flags, &epoch);
  if (rc == 0) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((mfma_f32_streamk_teams_kernel<G, MODE>), dim3(nwg), dim3(G::THREADS), G::LDS_BYTES, s,
                       (const float *)p.a, (const float *)p.b, (float *)p.c, p.n, p.k, p.m, t.tiles_n, t.tiles_m, t.spt, t.sr, t.sc,
                       t.teams_per_xcd, t.teams, slots, (unsigned long long *)flags, epoch);
    rc = (int)hipGetLastError();
  }
  const hipError_t f1 = flags ? hipFreeAsync(flags, s) : hipSuccess, f2 = slots ? hipFreeAsync(slots, s) : hipSuccess;
  return rc ? rc : f1 != hipSuccess ? (int)f1 : (int)f2;
}

// C tile = sum of the scratch slots of the workgroups whose ranges cut it, ascending k.  One workgroup per tile; tiles that
// one range covered whole were written by the main kernel and are skipped.
template <int BM, int BN>
__global__ __launch_bounds__(256) void streamk_fixup_kernel(const float *__restrict__ slots, float *__restrict__ C, unsigned N,
                                                            unsigned M, unsigned tiles_n, unsigned tiles_m, unsigned kBand,
                                                            unsigned spt, unsigned nwg) {
  const unsigned tile = blockIdx.x;
  const unsigned long long units = (unsigned long long)tiles_n * tiles_m * spt;
  const unsigned u_lo = tile * spt, u_hi = u_lo + spt;
  unsigned w = (unsigned)((unsigned long long)u_lo * nwg / units);
  while (w + 1 < nwg && sk_range_begin(units, w + 1, nwg) <= u_lo) ++w;
  while (w > 0 && sk_range_begin(units, w, nwg) > u_lo) --w;
  if (sk_range_begin(units, w + 1, nwg) >= u_hi) return;   // one range holds the whole tile
  const unsigned band = tile / (kBand * tiles_m), within = tile % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * BM, col0 = (within / rows_in_band) * BN;
  constexpr int QUADS = BM * BN / 4 / 256;
  f32x4 acc[QUADS];
#pragma unroll
  for (int i = 0; i < QUADS; ++i) acc[i] = (f32x4)0.0f;
  for (; w < nwg && sk_range_begin(units, w, nwg) < u_hi; ++w) {
    if (sk_range_begin(units, w + 1, nwg) == sk_range_begin(units, w, nwg)) continue;   // an empty range (fewer units than workgroups) wrote nothing
    const unsigned first_tile = sk_range_begin(units, w, nwg) / spt;
    const f32x4 *slot = (const f32x4 *)(slots + ((size_t)w * 2 + (tile != first_tile)) * (BM * BN));
#pragma unroll
    for (int i = 0; i < QUADS; ++i) acc[i] += slot[i * 256 + threadIdx.x];
  }
#pragma unroll
  for (int i = 0; i < QUADS; ++i) {
    const unsigned q = i * 256 + threadIdx.x, row = row0 + q / (BN / 4), col = col0 + (q % (BN / 4)) * 4;
    if (row < N && col < M) *(f32x4 *)(C + (size_t)row * M + col) = acc[i];
  }
}

template <typename G>
int launch_streamk(hipStream_t s, const Problem &p) {
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN, spt = p.k / G::BK;
  const unsigned nwg = 256 * G::MIN_WAVES;
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_streamk_kernel<G>, G::LDS_BYTES, configured)) return e;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipMemPool_t pool = nullptr;
  if (int rc = workspace_pool(dev, &pool)) return rc;
  float *slots = nullptr;
  const size_t slot_bytes = (size_t)nwg * 2 * G::BM * G::BN * sizeof(float);
  if ((e = hipMallocFromPoolAsync((void **)&slots, slot_bytes, pool, s)) != hipSuccess) return (int)e;
  // debug_poison: NaN in every slot AND in C (pure output: a tile nobody finished then shows as NaN instead of as whatever the buffer held)
  if (tuning(TUNE_DEBUG_POISON) == 1 && ((e = hipMemsetAsync(slots, 0xFF, slot_bytes, s)) != hipSuccess ||
                                         (e = hipMemsetAsync(p.c, 0xFF, (size_t)p.n * p.m * sizeof(float), s)) != hipSuccess)) { (void)hipFreeAsync(slots, s); return (int)e; }
  (void)hipGetLastError();
  const unsigned kband = band_rows(G::BM, G::BN, G::MIN_WAVES);
  hipLaunchKernelGGL((mfma_f32_streamk_kernel<G>), dim3(nwg), dim3(G::THREADS), G::LDS_BYTES, s, (const float *)p.a, (const float *)p.b,
                     (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, kband, spt, slots);
  int rc = (int)hipGetLastError();
  if (rc == 0) {
    hipLaunchKernelGGL((streamk_fixup_kernel<G::BM, G::BN>), dim3(tiles_n * tiles_m), dim3(256), 0, s, (const float *)slots, (float *)p.c,
                       p.n, p.m, tiles_n, tiles_m, kband, spt, nwg);
    rc = (int)hipGetLastError();
  }
  const hipError_t f = hipFreeAsync(slots, s);
  return rc ? rc : (int)f;
}

// C = sum over s of partials[s], s ascending: the second kernel of a split-K launch (4 floats per thread)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ partials, float *__restrict__ C,
                                                            size_t quads, unsigned splits) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= quads) return;
  const f32x4 *src = (const f32x4 *)partials;
  f32x4 acc = src[i];
  for (unsigned s = 1; s < splits; ++s) acc += src[(size_t)s * quads + i];
  ((f32x4 *)C)[i] = acc;
}

template <typename G, bool AT = false>
int launch_geo(hipStream_t s, const Problem &p, unsigned splits = 1) {
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_kernel<G, AT>, G::LDS_BYTES, configured)) return e;
  (void)hipGetLastError();   // a stale error of the application's own calls must not be reported as this launch's
  if (splits <= 1) {
    hipLaunchKernelGGL((mfma_f32_kernel<G, AT>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s,
                       (const float *)p.a, (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m,
                       band_rows(G::BM, G::BN, G::MIN_WAVES), 0u, (float *)nullptr);
    return (int)hipGetLastError();
  }
  // split-K: `splits` planes of N x M partial sums from the library's stream-ordered pool, then the ordered reduction
  const unsigned chunk = ((p.k + splits - 1) / splits + 31u) & ~31u;   // whole slabs of either depth
  const size_t plane = (size_t)p.n * p.m;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipMemPool_t pool = nullptr;
  if (int rc = workspace_pool(dev, &pool)) return rc;
  float *ws = nullptr;
  if ((e = hipMallocFromPoolAsync((void **)&ws, plane * splits * sizeof(float), pool, s)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL((mfma_f32_kernel<G, AT>), dim3(tiles_n * tiles_m * splits), dim3(G::THREADS), G::LDS_BYTES, s,
                     (const float *)p.a, (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m,
                     band_rows(G::BM, G::BN, G::MIN_WAVES), chunk, ws);
  int rc = (int)hipGetLastError();
  if (rc == 0) {
    const size_t quads = plane / 4;   // M % 4 == 0
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, ws, (float *)p.c, quads, splits);
    rc = (int)hipGetLastError();
  }
  const hipError_t f = hipFreeAsync(ws, s);
  return rc ? rc : (int)f;
}

// The geometries of the product.  Every shipped geometry exists as a (ScalarBase, VectorAddress) pair with identical
// arithmetic; the launcher takes the first when the problem allows it.
//                       TM WM WN BK  chain            chunk  reads            DMA
using T128x256 = Geo<2, 2, 2, 16, Chain::FlushIntoC, 256, Reads::Pipelined, Dma::ScalarBase>;     // default for large problems: two
using T128x256v = Geo<2, 2, 2, 16, Chain::FlushIntoC, 256, Reads::Pipelined, Dma::VectorAddress>; //   4-wavefront workgroups per CU
using T256x256 = Geo<2, 4, 2, 16, Chain::FlushIntoC, 256, Reads::Pipelined, Dma::ScalarBase>;     // 8 wavefronts; also every K x N A
using T256x256v = Geo<2, 4, 2, 16, Chain::FlushIntoC, 256, Reads::Pipelined, Dma::VectorAddress>;
using T128x128 = Geo<1, 4, 1, 32, Chain::FlushIntoC, 128, Reads::Pipelined, Dma::ScalarBase>;     // mid-size problems, two per CU
using T128x128v = Geo<1, 4, 1, 32, Chain::FlushIntoC, 128, Reads::Pipelined, Dma::VectorAddress>;
// Cross-check geometries (never picked by MM_PATH_AUTO; f32_variant 0 / 3): written plainly, scheduled by the compiler,
// with a different way of bounding the chain -- what the shipped kernels are tested against on the device.
using X128x256x32_2lvl = Geo<2, 2, 2, 32, Chain::TwoLevel, 16, Reads::CompilerPlaced, Dma::VectorAddress>;
using X256x256_single = Geo<2, 4, 2, 16, Chain::Single, 1, Reads::CompilerPlaced, Dma::VectorAddress>;

struct VariantInfo { int id; const char *name; unsigned bm, bn, bk, waves; };
template <typename G> constexpr VariantInfo info_of(int id, const char *name) { return {id, name, G::BM, G::BN, G::BK, G::NW}; }
// f32_variant values (the numbers are the round-1/2 sweep ids the profiles/ logs cite)
constexpr VariantInfo kVariants[] = {
    info_of<T128x256>(33, "mfma_f32_128x256x16_w4x2_flush4096"),
    info_of<T256x256>(8, "mfma_f32_256x256x16_w8_flush4096"),
    info_of<T128x128>(35, "mfma_f32_128x128x32_w4x2_flush4096"),
    info_of<Small>(64, "mfma_f32_64x64x32_w4x2_flush4096"),
    info_of<X128x256x32_2lvl>(0, "mfma_f32_128x256x32_w4_2lvl"),
    info_of<X256x256_single>(3, "mfma_f32_256x256x16_w8"),
};

const VariantInfo *find_variant(int v) {
  for (const VariantInfo &i : kVariants)
    if (i.id == v) return &i;
  return nullptr;
}

}  // namespace

int mfma_f32_num_variants() { return (int)(sizeof(kVariants) / sizeof(kVariants[0])); }
int mfma_f32_variant_id(int index) { return index >= 0 && index < mfma_f32_num_variants() ? kVariants[index].id : -1; }

// Shape-adaptive geometry (variant < 0): the 256x256 tile is the most economical per CU, but a launch runs
// in whole rounds of resident workgroups, so mid-size problems lose up to a round to quantisation
// (6144^3: 576 tiles = 2.25 rounds of 256) and small ones leave CUs idle (2048^3: 64 tiles).  Pick
// the candidate with the smallest estimated time = (workgroups the busiest CU runs) x tile area /
// relative efficiency.
static const TileCandidate kAutoCands[] = {{33, 128, 256, 2, 1.00}, {8, 256, 256, 1, 0.991}, {35, 128, 128, 2, 0.993}};

// Stream-K (launch_streamk_arrive) against the best whole-tile launch, in pick_tile's units (tile area x workgroups the
// busiest CU runs one after the other; a full round of 512 tiles of 128 x 128 = 2).  Fitted to
// profiles/r03w_f32_streamk_ordered_sweep.txt (2176^3 ... 9216^3): the persistent workgroups run at the whole-tile
// kernel's rate, plus a quarter of a tile for the launch's one-off parts that the whole tiles of a multi-round launch hide
// behind each other -- the cold first slabs, the scratch write and gather of the cut tiles, all of C written at the same
// moment.  It pays where the last round of whole tiles would leave much of the chip idle: 2304^3 +30 %, 2944^3 +30 %,
// 3072^3 +20 %, 3584^3 +23 %, 4608^3 +15 %, 5120^3 +10 %, 5888^3 +6 %, 7680^3 +6 %; it is not taken where whole tiles fit
// (2816^3, 3456^3, 4096^3, 5376^3, 6144^3, 8192^3) nor below a full round's worth of tiles per two CUs (<= 256 tiles:
// split-K or the plain kernel, see mfma_f32_splitk).
static bool streamk_wins(const Problem &p) {
  if (p.a_transposed || p.k % 32 != 0 || p.k < 256 || p.n_total) return false;
  if ((unsigned long long)((p.n + 127) / 128) * ((p.m + 127) / 128) * (p.k / 32) >= (1ull << 31)) return false;   // units are counted in 32 bits
  const double tiles = (double)((p.n + 127) / 128) * ((p.m + 127) / 128);
  const unsigned tn = (p.n + 127) / 128, tm = (p.m + 127) / 128;
  const bool teams_4x4 = tn % 4 == 0 && tm % 4 == 0;               // streamk_team_shape: full sharing inside an XCD
  if (tiles <= 256 || tiles > (teams_4x4 ? 4096 : 2304)) return false;   // smaller teams turn fabric-bound as the job grows (54^2, 58^2, 62^2 tiles: -3 %)
  double whole = 0;
  pick_tile(kAutoCands, 3, p.n, p.m, &whole);
  const double sk = (2.0 * tiles / 512 + 0.25) * 128 * 128 / 0.993;
  return sk < 0.99 * whole;
}

template <typename G>
static int launch_small(hipStream_t s, const Problem &p, unsigned splits) {
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_small_kernel<G>, G::LDS_BYTES, configured)) return e;
  (void)hipGetLastError();
  const unsigned kband = band_rows(G::BM, G::BN, G::PER_CU);
  if (splits <= 1) {
    hipLaunchKernelGGL((mfma_f32_small_kernel<G>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s, (const float *)p.a,
                       (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, kband, 0u, (float *)nullptr);
    return (int)hipGetLastError();
  }
  // split-K, exactly as launch_geo: planes of partial sums from the library's pool, then the ordered reduction
  const unsigned chunk = ((p.k + splits - 1) / splits + 31u) & ~31u;
  if ((unsigned long long)(splits - 1) * chunk >= p.k) return kErrNotSupported;   // (the rule never asks for an empty chunk)
  const size_t plane = (size_t)p.n * p.m;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipMemPool_t pool = nullptr;
  if (int rc = workspace_pool(dev, &pool)) return rc;
  float *ws = nullptr;
  if ((e = hipMallocFromPoolAsync((void **)&ws, plane * splits * sizeof(float), pool, s)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL((mfma_f32_small_kernel<G>), dim3(tiles_n * tiles_m * splits), dim3(G::THREADS), G::LDS_BYTES, s, (const float *)p.a,
                     (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, kband, chunk, ws);
  int rc = (int)hipGetLastError();
  if (rc == 0) {
    const size_t quads = plane / 4;   // M % 4 == 0
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, ws, (float *)p.c, quads, splits);
    rc = (int)hipGetLastError();
  }
  const hipError_t f = hipFreeAsync(ws, s);
  return rc ? rc : (int)f;
}

// K chunks of the shape-adaptive split-K rule for `tiles` tiles of 128 x 128 (mfma_f32_splitk): when the tiles leave a quarter
// or more of the CUs without a workgroup, as many copies of the tile grid as fit the 512 places, at most 8, chunks >= 256 k
static unsigned auto_split_chunks(unsigned long long tiles, unsigned k) {
  unsigned s = tiles <= 192 ? (unsigned)(512 / tiles) : 1;   // 256 tiles already give every CU a workgroup (2048^3: 125 vs 112 TF split)
  if (s > 8) s = 8;
  while (s > 1 && k / s < 256) --s;
  return s;
}

// The 64 x 64 geometry (launch_small) addresses a tile's rows with 32-bit byte offsets from a 64-bit base.
static bool small_fits(const Problem &p) {
  return !p.a_transposed && 64ull * p.k * 4ull < (1ull << 32) && 32ull * p.m * 4ull < (1ull << 32);
}

// Below a round of 128 x 128 tiles: the 64 x 64 geometry (whole K, no second kernel) against the 128 x 128 one with its
// split-K rule, in microseconds, fitted to profiles/r03x_f32_small_tile_k_slope.txt and r03x_f32_small_shapes.txt:
//   64 x 64:   3.5 + slabs x 0.52 (one workgroup per CU: a lone dependent MFMA chain per SIMD) or x 0.90 (two per CU);
//              more than 512 tiles never (a second round: the 128 x 128 tile's half bytes per flop win);
//   128 x 128: 5 + slabs x 1.79 (one workgroup per CU) or x 3.49 (two); a split adds the partial planes' round trip and
//              the reduce kernel: + 13.5.
// 1024^3 105 vs 75 TF, 1024 x 512 x 1024 89 vs 40, 768^3 56 vs 34, 512^3 23 vs 12, 1280^3 98 vs 91; the 128 x 128 geometry with
// its split keeps 1536^3 (576 tiles of 64 x 64); few tiles and a long K split on this geometry too (small_split_chunks).  Decided on the whole job
// (Problem::n_total), like the split itself, so row slabs keep the bits of the one-launch result.
// The 64 x 64 geometry splits K too when its tiles leave half the CUs or more without a workgroup (<= 128 tiles): as many
// copies of the tile grid as fit one workgroup per CU, at most 8, chunks of at least 512 k, and only if the model says the
// second kernel pays (+ 13.5 us).  512 x 4096 x 512: 31 -> 81 TF (split-K on the 128 x 128 geometry: 53), 256 x 8192 x 256: 8 -> 40 (16).
static unsigned small_split_chunks(const Problem &p) {
  const unsigned rows = p.n_total ? p.n_total : p.n;
  const unsigned long long t64 = (unsigned long long)((rows + 63) / 64) * ((p.m + 63) / 64);
  if (tuning(TUNE_F32_SPLITK) == 1 || t64 > 128) return 1;
  unsigned s = (unsigned)(256 / t64);
  if (s > 8) s = 8;
  while (s > 1 && p.k / s < 512) --s;
  const double slabs = (p.k + 31) / 32;
  return (s > 1 && 13.5 + (slabs / s) * 0.52 < slabs * 0.52) ? s : 1;
}
static bool small_wins(const Problem &p) {
  if (!small_fits(p) || p.k < 32) return false;
  const unsigned rows = p.n_total ? p.n_total : p.n;
  const unsigned long long t64 = (unsigned long long)((rows + 63) / 64) * ((p.m + 63) / 64), t128 = (unsigned long long)((rows + 127) / 128) * ((p.m + 127) / 128);
  if (t64 > 512) return false;
  const double slabs = (p.k + 31) / 32;
  const unsigned ss = small_split_chunks(p);
  const double t_small = 3.5 + (ss > 1 ? 13.5 : 0.0) + (slabs / ss) * (t64 * ss <= 256 ? 0.52 : 0.90);
  const unsigned s = tuning(TUNE_F32_SPLITK) == 1 ? 1 : auto_split_chunks(t128, p.k);
  const double t_128 = (s > 1 ? 18.5 : 5.0) + (slabs / s) * (t128 * s <= 256 ? 1.79 : 3.49);
  return t_small < t_128;
}

int mfma_f32_auto_variant(const Problem &p) {
  // relative efficiencies at 16384^3 at the end of round 2, all three with the pinned schedule and scalar-base DMA:
  // 128x256 as two 4-wavefront workgroups per CU (33) 152.2 TF, 256x256 / 8 wavefronts (8) 150.8, 128x128x32 (35) 151.2
  // (profiles/r02z_f32_scalar_base_dma.log, r02z_f32_small_tile_scalar_base_dma.log); small and mid-size shapes:
  // r02z_f32_small_shapes_after_scalar_base_dma.log
  const int knob = tuning(TUNE_F32_SPLITK);
  if (knob < 0 && streamk_wins(p)) return 35;   // the 128 x 128 geometry is the one stream-K runs on
  const int v = pick_tile(kAutoCands, 3, p.n, p.m);
  if (v == 35 && (knob < 0 || knob == 1) && small_wins(p)) return 64;   // (a forced split or stream-K means the 128 x 128 geometry)
  // Round 4, decided by energy.  Where the problem is whole rounds of 256 x 256 tiles, the two big
  // geometries are within 1 % of each other in steady state (float 16384^3: 151.29 vs 150.72 TF, 20 launches back to back,
  // alternating in one process: profiles/r04c_f32_default_ab_steady_state.txt), but the 256 x 256 one draws 4.6 % less board
  // power for it (1135 W against 1188 W, 128.5 against 122.8 GFLOP/s/W: profiles/r04b_f32_energy_33_vs_8.txt) and pulls half
  // the bytes through the fabric (33.5 GB against 72 GB per launch) -- eight boards of that on the node this is meant
  // for.  Same bits either way (the shipped geometries issue the same MFMAs per output element in the same order).
  if (v == 33) {
    const unsigned long long t256 = (unsigned long long)((p.n + 255) / 256) * ((p.m + 255) / 256);
    if (t256 % 256 == 0 && t256 >= 4 * 256) return 8;
  }
  return v;
}

bool mfma_f32_serves(const Problem &p) {
  if (!(p.n >= 1 && p.m >= 4 && p.k >= 8 && p.m % 4 == 0 && p.k % 8 == 0)) return false;
  return !p.a_transposed || (p.n >= 4 && p.n % 4 == 0);  // K x N A is DMA'd in 16-B chunks along N
}

// K x N A (MM_TRANSPOSED_A) outside whole rounds of 256 x 256 tiles.  The K x N kernel exists in one geometry, so a problem
// that does not fill rounds of it pays dearly -- 1024^3 9 TF against 101 row-major, 2048^3 38 against 141, 6144^3 113
// against 150 (profiles/r03y_f32_transposed_a_sizes.txt).  Under the shape-adaptive pick such a problem is therefore
// transposed into a stream-ordered workspace first (N x K x 4 B, read once and written once: 2.5 % of the product at
// 6144^3, less above) and then runs whatever the row-major rules choose -- with their bits: the result equals the row-major
// call's.  Whole rounds (4096^3, 8192^3, 16384^3 ...) and a pinned f32_variant keep the K x N kernel as it is.
static bool transposes_first(const Problem &p, int variant_knob) {
  if (!p.a_transposed || variant_knob >= 0) return false;
  const unsigned long long tiles = (unsigned long long)((p.n + 255) / 256) * ((p.m + 255) / 256), rounds = (tiles + 255) / 256;
  if ((double)tiles >= 0.93 * (double)(rounds * 256)) return false;
  return (unsigned long long)p.n * p.k * sizeof(float) <= (2ull << 30);
}
static Problem as_row_major(const Problem &p) {
  Problem q = p;
  q.a_transposed = false;
  return q;
}

// dst[n][k] = src[k][n]: 64 x 64 tiles through LDS, 16-byte accesses on both sides (N % 4 == 0 and K % 8 == 0 here)
__global__ __launch_bounds__(256) void transpose_kxn_kernel(const float *__restrict__ src, float *__restrict__ dst, unsigned K, unsigned N) {
  __shared__ float tile[64][65];
  const unsigned blocks_n = (N + 63) / 64;   // 1-D grid: either dimension may exceed the 65535 of gridDim.y
  const unsigned k0 = (blockIdx.x / blocks_n) * 64, n0 = (blockIdx.x % blocks_n) * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned idx = threadIdx.x + 256 * i, r = idx >> 4, c = (idx & 15u) * 4;
    if (k0 + r < K && n0 + c < N) {
      const f32x4 v = *(const f32x4 *)(src + (size_t)(k0 + r) * N + n0 + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[r][c + j] = v[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned idx = threadIdx.x + 256 * i, r = idx >> 4, c = (idx & 15u) * 4;
    if (n0 + r < N && k0 + c < K) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = tile[c + j][r];
      *(f32x4 *)(dst + (size_t)(n0 + r) * K + k0 + c) = v;
    }
  }
}

// The one place that decides which geometry a (problem, f32_variant knob) pair runs: mm_kernel_name,
// mm_kernel_info and the launcher all go through it.  Returns the variant id, or -1 when unsupported.
int mfma_f32_resolve(const Problem &p, int variant) {
  if (!mfma_f32_serves(p)) return -1;
  if (p.a_transposed) {
    if (transposes_first(p, variant)) return mfma_f32_resolve(as_row_major(p), variant);
    return 8;                                          // K x N A: the 256 x 256 geometry, whatever the knob says
  }
  if (variant < 0) variant = mfma_f32_auto_variant(p);
  if (variant == 64 && !small_fits(p)) return -1;
  return find_variant(variant) ? variant : -1;
}

// Split-K for problems that cannot fill the chip with whole tiles (1024^3 ran at 33 TF -- 64 tiles of
// 128 x 128 for 512 workgroup slots).  Only the 128 x 128 geometry under the shape-adaptive pick splits: when its tiles
// leave a quarter or more of the CUs without a workgroup (<= 192 tiles), K is cut into S = min(slots / tiles, K / 256, 8) chunks, the S copies of the tile grid run
// side by side and a second kernel adds the S partial planes in ascending order -- deterministic, run to run and
// whatever the placement, but a different summation order than the unsplit kernel (as accurate or better: shorter
// chains).  f32_splitk: -1 this rule (and stream-K by streamk_wins), 0 stream-K (teams; the last part to arrive gathers), 1 neither, 2..8 that many chunks, 9 stream-K in
// single ranges with its fix-up kernel (cross-check), 11 the teams form with a fix-up kernel (cross-check): the bits of 0 (always only
// for row-major A and variant 35).  Row slabs of a bigger job (Problem::n_total) never take stream-K: its unit ranges would
// depend on the slab, and with them a row's summation order.
int mfma_f32_splitk(const Problem &p, int variant) {
  if (p.a_transposed) return transposes_first(p, tuning(TUNE_F32_VARIANT)) ? mfma_f32_splitk(as_row_major(p), variant) : 1;
  if (variant == 64) {   // the 64 x 64 geometry: its own rule under the shape-adaptive pick, 2..8 forced by the knob
    const int forced = tuning(TUNE_F32_SPLITK);
    unsigned s = forced >= 2 && forced <= 8 ? (unsigned)forced : tuning(TUNE_F32_VARIANT) >= 0 ? 1u : small_split_chunks(p);
    while (s > 1 && p.k / s < 64) --s;
    return (int)s;
  }
  if (variant != 35) return 1;
  const int knob = tuning(TUNE_F32_SPLITK);
  const unsigned long long sk_units = (unsigned long long)((p.n + 127) / 128) * ((p.m + 127) / 128) * (p.k / 32);   // the kernels count units in 32 bits
  if (knob == 0 || knob == 9 || knob == 11 || knob == 12) return (p.k % 32 == 0 && p.k >= 64 && sk_units < (1ull << 31)) ? knob : 1;   // stream-K forced: 0 = teams, last arriver gathers (the one auto takes), 11 = teams + fix-up kernel, 9 = single ranges + fix-up kernel
  const unsigned rows = p.n_total ? p.n_total : p.n;   // of the whole job (Problem::n_total)
  const unsigned long long tiles = (unsigned long long)((rows + 127) / 128) * ((p.m + 127) / 128);
  unsigned s;
  if (knob >= 1) s = (unsigned)knob;
  else if (tuning(TUNE_F32_VARIANT) >= 0) s = 1;          // a pinned geometry is run as pinned
  else if (streamk_wins(p)) return 0;                     // a few partial rounds of tiles: stream-K
  else return (int)auto_split_chunks(tiles, p.k);
  if (s > 8) s = 8;
  while (s > 1 && p.k / s < 256) --s;                     // chunks of at least 256 k
  return (int)s;
}

const char *mfma_f32_name(int v) {
  const VariantInfo *i = find_variant(v);
  return i ? i->name : "?";
}

void mfma_f32_geometry(int v, unsigned *bm, unsigned *bn, unsigned *bk, unsigned *waves) {
  const VariantInfo *i = find_variant(v);
  if (!i) i = find_variant(8);
  *bm = i->bm; *bn = i->bn; *bk = i->bk; *waves = i->waves;
}

// The scalar-base DMA addresses a tile's rows with 32-bit byte offsets from a 64-bit base: 256 rows x K x 4 B (and
// 32 k-rows x M x 4 B for B / a K x N A) must stay below 4 GiB; longer rows take the VectorAddress twin.
static bool sdma_fits(const Problem &p, unsigned bk) {
  const unsigned long long span = 256ull * (p.a_transposed ? 1ull : p.k) * 4ull, spanb = 32ull * (p.m > p.n ? p.m : p.n) * 4ull;
  return p.k >= bk && span < (1ull << 32) && spanb < (1ull << 32);
}

int launch_mfma_f32(hipStream_t s, const Problem &p, int variant) {
  const int v = mfma_f32_resolve(p, variant);
  if (v < 0) return kErrNotSupported;
  if (p.a_transposed) {
    if (!transposes_first(p, variant)) return sdma_fits(p, 16) ? launch_geo<T256x256, true>(s, p) : launch_geo<T256x256v, true>(s, p);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipMemPool_t pool = nullptr;
    if (int rc = workspace_pool(dev, &pool)) return rc;
    float *an = nullptr;
    if ((e = hipMallocFromPoolAsync((void **)&an, (size_t)p.n * p.k * sizeof(float), pool, s)) != hipSuccess) return (int)e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(transpose_kxn_kernel, dim3(((p.n + 63) / 64) * ((p.k + 63) / 64)), dim3(256), 0, s, (const float *)p.a, an, p.k, p.n);
    int rc = (int)hipGetLastError();
    if (rc == 0) {
      Problem q = as_row_major(p);
      q.a = an;
      rc = launch_mfma_f32(s, q, variant);
    }
    const hipError_t f = hipFreeAsync(an, s);
    return rc ? rc : (int)f;
  }
  switch (v) {
    case 33: return sdma_fits(p, 16) ? launch_geo<T128x256>(s, p) : launch_geo<T128x256v>(s, p);
    case 8: return sdma_fits(p, 16) ? launch_geo<T256x256>(s, p) : launch_geo<T256x256v>(s, p);
    case 35: {
      const int splits = mfma_f32_splitk(p, v);
      if (splits == 0) return sdma_fits(p, 32) ? launch_streamk_arrive<T128x128>(s, p) : launch_streamk_arrive<T128x128v>(s, p);
      if (splits == 11) return sdma_fits(p, 32) ? launch_streamk_teams<T128x128>(s, p) : launch_streamk_teams<T128x128v>(s, p);
      if (splits == 12) return sdma_fits(p, 32) ? launch_streamk_arrive<T128x128, Combine::Ticket>(s, p) : launch_streamk_arrive<T128x128v, Combine::Ticket>(s, p);
      if (splits == 9) return sdma_fits(p, 32) ? launch_streamk<T128x128>(s, p) : launch_streamk<T128x128v>(s, p);
      return sdma_fits(p, 32) ? launch_geo<T128x128>(s, p, (unsigned)splits) : launch_geo<T128x128v>(s, p, (unsigned)splits);
    }
    case 64: return launch_small<Small>(s, p, (unsigned)mfma_f32_splitk(p, v));
    case 0: return launch_geo<X128x256x32_2lvl>(s, p);
    case 3: return launch_geo<X256x256_single>(s, p);
  }
  return kErrNotSupported;
}

}  // namespace mm
