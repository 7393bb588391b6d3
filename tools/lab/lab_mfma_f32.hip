// LAB EDITION (tools/lab, built into libmm_gemm_amd_lab.so only): every schedule, ring depth and ablation the fp32
// kernel went through in rounds 1-2, kept so that the sweeps cited under profiles/ stay reproducible.  Variants 28-32
// skip work on purpose and return WRONG results (power / issue-cost breakdown); they need MM_ABLATIONS=1.  The product
// file is gemm_hls_amd/csrc/mm_mfma_f32.hip.
//
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
//   K is consumed in slabs of BK floats through an NS-deep LDS ring.
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
// Two-level accumulation (TWO_LEVEL): every KC k-slabs the MFMA accumulators are added into a
// second register set and restarted from zero.  A single 16384-long f32 chain of positive
// products drifts to ~1e-5 relative in the worst element; with 512-deep chunks
// the worst case stays < 1e-6, inside BASELINE.json's 1e-5 bar with margin.
//
// Chunked flush (FC > 0): the other way to bound the chain length, for geometries whose register
// file has no room for a second accumulator set (8 wavefronts of 64x128, 2 per SIMD): every FC
// slabs the workgroup adds its accumulators into its own C tile in HBM (first chunk: plain store)
// and restarts them from zero.  The same workgroup owns the tile for the whole launch, so the
// read-modify-write is race-free and deterministic; it costs one extra read+write of C per chunk
// (K/(FC*BK) - 1 times 2 x N*M*4 bytes, mostly served by the Infinity Cache).
//
// Edges: N arbitrary (row indices clamped for loads, stores predicated); M % 4 == 0 (column
// chunks clamped / predicated); K % 8 == 0 (a partial last slab is consumed in 8-deep groups;
// the DMA of its unused part is clamped to valid addresses and never read).  Everything else is
// served by the predicated kernels (mm_valu_tile / mm_ordered).
#include <cstdlib>
#include <type_traits>

#include "../../gemm_hls_amd/csrc/mm_common.h"

namespace mm {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int TM_, int WM_, int WN_, int BK_, int NS_, bool TWO_LEVEL_, int KC_, int FC_ = 0, bool STAGGER_ = false,
          bool PIN_ = false, int HINT_ = 0>
struct Geo {
  // HINT: sched_group_barrier shaping of a k-group (softer than PIN's sched_barrier fences):
  //   1: the group's 6 fragment reads first, then its MFMAs; after the slab barrier 4 MFMAs, the 4 DMA
  //      pieces, the reads, the remaining MFMAs.   2: reads in the middle of the group (8 MFMAs, 6 reads, rest).
  static constexpr int HINT = HINT_;
  // PIN: a scheduling fence after every group's fragment reads.  Without it hipcc sinks four of the six
  // ds_read_b128 of a k-group down to just before the slab barrier, where the s_waitcnt lgkmcnt(0)
  // that the barrier needs then exposes their full LDS latency once per slab.
  static constexpr bool PIN = PIN_;
  // STAGGER: the second half of the wavefronts (the SIMD partners of the first half) issue their
  // share of a slab's DMA one k-group later instead of right after the barrier.  An LDS-DMA
  // instruction blocks its wave's issue for 60-190 cycles; with both waves of a SIMD doing that at the
  // same moment the matrix pipe idles (the 5-6 % this kernel was missing); staggered, one partner
  // always has MFMAs to issue.  Needs NS == 2 (the pre-barrier wait is vmcnt(0), whatever the order).
  static constexpr bool STAGGER = STAGGER_;
  static constexpr int TM = TM_, WM = WM_, WN = WN_, BK = BK_, NS = NS_, KC = KC_;
  static constexpr int FC = FC_;  // > 0: flush the accumulators into C every FC slabs (see below)
  static constexpr bool TWO_LEVEL = TWO_LEVEL_;
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
  static constexpr int MIN_WAVES = (NW == 4 && 2 * LDS_BYTES <= 160 * 1024 && (TWO_LEVEL_ ? 2 : 1) * TM * TN * 16 <= 128) ? 2 : 1;
  static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;  // wave-level DMA instructions
  static constexpr int LA = NA / NW, LB = NB / NW;                // ... per wavefront
  static constexpr int KG = BK / 8;                               // 8-deep k-groups per slab
  static_assert(BK == 16 || BK == 32, "BK");
  static_assert(NA % NW == 0 && NB % NW == 0, "DMA instructions must split evenly over waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(!STAGGER || (NS == 2 && BK / 8 >= 2 && NW % 2 == 0), "stagger");
};

template <typename G, bool AT>
__global__ __launch_bounds__(G::THREADS, G::MIN_WAVES) void mfma_f32_kernel(const float *__restrict__ A,
                                                              const float *__restrict__ B,
                                                              float *__restrict__ C, unsigned N,
                                                              unsigned K, unsigned M,
                                                              unsigned tiles_n, unsigned tiles_m,
                                                              unsigned kBand) {
  constexpr int TM = G::TM, TN = G::TN, BK = G::BK, NS = G::NS, CPR = G::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wm = wave / G::WN, wn = wave % G::WN;
  const unsigned lo = lane & 31u, hi = lane >> 5;

  // ---- workgroup -> output tile: XCD-contiguous chunks, then bands of 8 tile-rows ----------
  const unsigned nwg = tiles_n * tiles_m;
  const unsigned lin = xcd_remap(blockIdx.x, nwg);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned tile_row = band * kBand + within % rows_in_band;
  const unsigned tile_col = within / rows_in_band;
  const unsigned row0 = tile_row * G::BM, col0 = tile_col * G::BN;

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
      a_row_off[i] = (size_t)min(row0 + row, N - 1) * K;
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

  auto stage = [&](unsigned buf, unsigned k0) {
    char *base = smem + buf * G::STAGE_BYTES;
    if (G::HINT == 8) asm volatile("s_mov_b64 exec, 0" ::: "memory");  // ablation: the DMA instructions issue with no lane active
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
    if (G::HINT == 8) asm volatile("s_mov_b64 exec, -1" ::: "memory");
  };

  // HINT 9: the DMA in its scalar-base form -- uniform 64-bit base in SGPRs + a constant 32-bit per-lane
  // offset, as the half kernels use it: one address VGPR per lane instead of two, no per-slab 64-bit VALU address
  // arithmetic.  A slab start past K - BK (beyond the end, or the partial last slab) is clamped to K - BK, uniformly:
  // a partial last slab then sits in the SECOND half of its buffer (see the last-slab loop).  Needs K >= BK.
  constexpr bool SDMA = G::HINT == 9;
  unsigned voa[G::LA], vob[G::LB];
  if (SDMA) {
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      const unsigned slot = (wave + G::NW * i) * 64 + lane, row = slot / CPR;
      if (AT) voa[i] = a_kchunk[i] * N * 4u + ((unsigned)a_row_off[i] - row0) * 4u;   // K x N: k-row, clamped column
      else voa[i] = (min(row0 + row, N - 1) - row0) * K * 4u + a_kchunk[i] * 16u;
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
                              : AT ? (unsigned long long)(A + (size_t)kc * N + row0) : (unsigned long long)(A + (size_t)row0 * K + kc);
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
  f32x16 master[G::TWO_LEVEL ? TM : 1][G::TWO_LEVEL ? TN : 1];
  if (G::TWO_LEVEL) {
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
  // read-modify-write into 32 dependent round trips.
  typedef __attribute__((address_space(1))) f32x4 *gquad_t;
  const bool interior_block = col0 + wn * 128 + 128 <= M && row0 + wm * TM * 32 + TM * 32 <= N;  // wavefront-uniform
  auto rmw_interior = [&](bool accumulate, auto batchc, auto value) {
    constexpr int BATCH = decltype(batchc)::value;
    unsigned Mv = M;
    asm volatile("" : "+s"(Mv));  // opaque: the row offsets must not be hoisted out of the chunk loop (they would stay
                                  // live across the main loop and cost it registers)
    char *base = (char *)(C + (size_t)(row0 + wm * TM * 32) * Mv + col0 + wn * 128);
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

  // C (+)= accumulators; accumulators = 0   (chunked flush, FC > 0 only)
  auto flush_tile = [&](bool accumulate) {
    if ((G::HINT == 3 || G::HINT == 9) && TM >= 2 && interior_block) {  // (the 32-row wavefront tile measured slower with it: 139.0 vs 141.1 TF)
      rmw_interior(accumulate, std::integral_constant<int, 4>{}, [&](int mi, int tt, int r) { return acc[mi][tt][r]; });
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int tt = 0; tt < TN; ++tt) acc[mi][tt] = (f32x16)0.0f;
      return;
    }
    const unsigned ccol = col0 + wn * 128 + 4 * lo;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned ri = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const unsigned row = row0 + wm * TM * 32 + (AT ? TM * ri + mi : mi * 32 + ri);
        if (ccol < M && row < N) {
          f32x4 *dst = (f32x4 *)(C + (size_t)row * M + ccol);
          f32x4 v;
#pragma unroll
          for (int tt = 0; tt < TN; ++tt) v[tt] = acc[mi][tt][r];
          if (accumulate) v += *dst;
          *dst = v;
        }
      }
#pragma unroll
      for (int tt = 0; tt < TN; ++tt) acc[mi][tt] = (f32x16)0.0f;
    }
  };

  const unsigned num_tiles = (K + BK - 1) / BK;   // slabs, the last one possibly partial
  constexpr int L = G::LA + G::LB;                // DMA instructions per wavefront per slab

  // ---- prologue: fill the whole ring (slabs 0..NS-1), wait for slab 0 ------------------------
  // stage() clamps every source address, so staging a slab index past the end is harmless
  // (it lands in a ring slot nobody reads again); this keeps the steady state branch-free and
  // the vmcnt immediates constant.
#pragma unroll
  for (int s = 0; s < NS; ++s) stage_any(s, s * BK);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * L) : "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 af0[TM], bf0[4], af1[TM], bf1[4];
  load_frags(0, 0, af0, bf0);
  const bool late = G::STAGGER && wave >= G::NW / 2;  // waves w and w + NW/2 share a SIMD

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
      if (G::PIN) {
        // Pinned order: [barrier, DMA] | first half of this group's MFMAs | the NEXT group's fragment
        // reads | second half.  The compiler's wait for a group's fragments then always sits behind
        // 16 MFMAs (1024 cycles) of cover instead of right behind the reads.
        if (kg + 1 == G::KG) {
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
          __builtin_amdgcn_s_barrier();
          if (!G::STAGGER || !late) stage(buf, (t + NS) * BK);
        } else if (G::STAGGER && kg == 0 && late && t > 0) {
          stage((t - 1) % NS, (t - 1 + NS) * BK);  // the refill the early half issued a k-group ago
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(afc, bfc, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kg + 1 < G::KG) load_frags(buf, kg + 1, afn, bfn);
        else load_frags((t + 1) % NS, 0, afn, bfn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(afc, bfc, 1);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (G::HINT >= 3) {
        // HINT 4..8 are ablations (timing only, wrong results): 4 no barrier; 5 no DMA; 6 no fragment reads; 7 neither;
        // 8 DMA instructions issued with an empty exec mask (issue cost without the memory traffic)
        constexpr bool DO_DMA = G::HINT != 5 && G::HINT != 7, DO_READS = G::HINT != 6 && G::HINT != 7;
        // (HINT 4 = the same without the workgroup barrier: an ablation that races by construction, timing only.)
        // Software-pipelined fragment reads with the barrier still covered by MFMAs.  The plain form below says
        // "read the next group's fragments, then multiply this group's", but the machine scheduler sinks every read
        // down to its first use: the shipped loop is  ds_read x4 | s_waitcnt lgkmcnt(0) | 16 MFMAs  four times per
        // slab (ISA of the default geometry), i.e. four exposed LDS round trips per slab with both wavefronts of a
        // SIMD in lock-step.  Here the order is pinned:
        //   not the slab's last group:  16 MFMAs | reads of the next group | 16 MFMAs
        //   the last group:             16 MFMAs | wait + barrier | DMA of slab t+NS and reads of slab t+1's first
        //                               group, one between MFMAs | the remaining MFMAs
        // so every fragment is requested >= 16 MFMAs (1024 cycles) before its first use.
        constexpr int NM = 8 * TM, NR = TM + 4;   // MFMAs of a half group, fragment reads of a group
        constexpr bool EARLY = L + NR > NM;       // narrow wavefront tiles: the post-barrier interleave needs the whole group
        static_assert(L + NR <= 2 * NM, "a k-group has too few MFMAs to spread the DMA pieces and reads over");
        __builtin_amdgcn_sched_barrier(0);
        if (kg + 1 < G::KG) {
          mfma_half(afc, bfc, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (DO_READS) load_frags(buf, kg + 1, afn, bfn);
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(afc, bfc, 1);
        } else {
          if (!EARLY) {
            mfma_half(afc, bfc, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
          if (G::HINT != 4) __builtin_amdgcn_s_barrier();
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
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TOTAL - L - NR, 0);
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          if (DO_DMA) stage(buf, (t + NS) * BK);
          if (DO_READS) load_frags((t + 1) % NS, 0, afn, bfn);
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
      if (kg + 1 < G::KG) {
        if (G::STAGGER && kg == 0 && late && t > 0) stage((t - 1) % NS, (t - 1 + NS) * BK);  // the refill E issued a k-group ago
        load_frags(buf, kg + 1, afn, bfn);
      } else {
        // Slab t+1 must have landed; slabs t+2 .. t+NS-1 may stay in flight across the barrier.
        // This wave's LDS reads of slab t are all in registers (lgkmcnt(0)), so after the
        // barrier its ring slot is free for slab t+NS.
        if (G::HINT) __builtin_amdgcn_sched_barrier(0);  // the wait + barrier must not float up into the previous group
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!G::STAGGER || !late) stage(buf, (t + NS) * BK);
        load_frags((t + 1) % NS, 0, afn, bfn);
      }
      mfma_group(afc, bfc);
      if (G::HINT == 1 || G::HINT == 2) {  // masks: MFMA 0x8, VMEM read 0x20, DS read 0x100
        constexpr int NM = 16 * TM, NR = TM + 4;
        if (kg + 1 < G::KG) {
          if (G::HINT == 1) {
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM - 8, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, L, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, G::HINT == 1 ? 0 : 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NM - (G::HINT == 1 ? 4 : 8), 0);
        }
      }
    }
  };

  const unsigned steady = num_tiles - 1;  // slabs 0 .. num_tiles-2 are full and have a successor
  bool flushed = false;
  if (G::FC > 0) {
    for (unsigned t0 = 0; t0 < steady; t0 += G::FC) {
      const unsigned tend = min(t0 + (unsigned)G::FC, steady);
      for (unsigned t = t0; t < tend; ++t) slab(t);
      if (tend < steady) {  // a further chunk follows: C (+)= acc, restart the chain
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flush_tile(flushed);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flushed = true;
      }
    }
  } else if (G::TWO_LEVEL) {
    for (unsigned t0 = 0; t0 < steady; t0 += G::KC) {
      const unsigned tend = min(t0 + (unsigned)G::KC, steady);
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
    if ((G::HINT == 3 || G::HINT == 9) && TM >= 2 && interior_block) {
      rmw_interior(accumulate, std::integral_constant<int, (TM * TN > 8 ? 4 : 8)>{}, [&](int mi, int tt, int r) {
        float x = acc[mi][tt][r];
        if (G::TWO_LEVEL) x += master[mi][tt][r];
        return x;
      });
      return;
    }
    const unsigned ccol = col0 + wn * 128 + 4 * lo;
    if (ccol < M) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned ri = (r & 3) + 8 * (r >> 2) + 4 * hi;
          const unsigned row = row0 + wm * TM * 32 + (AT ? TM * ri + mi : mi * 32 + ri);
          if (row < N) {
            f32x4 *dst = (f32x4 *)(C + (size_t)row * M + ccol);
            f32x4 v;
#pragma unroll
            for (int tt = 0; tt < TN; ++tt) {
              float x = acc[mi][tt][r];
              if (G::TWO_LEVEL) x += master[mi][tt][r];
              v[tt] = x;
            }
            if (accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  };
  if (G::FC > 0 && flushed) write_tile(true); else write_tile(false);
}

// =================================================================================================
// Ping-pong schedule for fp32 (round 2) -- the structure that took the half kernel from 67 % to 90 %
// MFMA utilisation (mm_mfma_f16.hip), with this file's fragment trick:
//   256 x 256 tile, 8 waves of 64 x 128 (2 x 4 accumulators), 16-deep slabs in a 4-slab LDS ring
//   (4 x 32 KiB), 3 slabs in flight; waves 0-3 and their SIMD partners 4-7 run the same code one
//   barrier apart: one group issues its 64 MFMAs of slab u (4096 cycles per SIMD) while the other
//   reads its 12 fragment vectors of its next slab and issues its 4 DMA pieces of slab u + 3.
//   A slab [256][16] floats: 64-B rows, chunk ^ (row>>2)&3; B slab [16][256] floats as in memory.
//   DMA: uniform SGPR base + constant per-lane 32-bit offset; counted vmcnt(8); a slab is read one
//   segment after the barrier that retires it, a buffer refilled only after a barrier every reader
//   passed with lgkmcnt(0).
// Accumulation order per output and the flush every 4096 k are those of the default kernel (V8):
// the two are bit-identical (tests).  Requirements: K % 16 == 0, K >= 64, M % 4 == 0, row-major A.
// MEASURED SLOWER than the default kernel (143.4 vs 145.7 TF at 16384^3, profiles/r02m_*): with
// 64-cycle MFMAs the compiler-scheduled one-barrier-per-slab stream already hides its DMA issue and
// LDS latency; the antiphase buys +1 % over its own lock-step ablation and pays more for the two
// barriers per slab.  Kept as variants 20 / 21 (tested) for that record; not dispatched by default.
struct GeoF32PP {
  static constexpr int BM = 256, BN = 256, BK = 16, THREADS = 512, TM = 2, TN = 4;
  static constexpr int A_BYTES = BM * BK * 4, B_BYTES = BK * BN * 4, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 4 * STAGE_BYTES;  // 128 KiB
  static constexpr int FLUSH_SLABS = 256;            // 4096 k
};
#define MM_DMA_PIECE(vo, sb, la) "s_mov_b32 m0, " la "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " vo ", " sb "\n\t"

template <int VAR>  // bit 1: lock-step ablation
__global__ __launch_bounds__(GeoF32PP::THREADS) void mfma_f32_pp_kernel(const float *__restrict__ A,
                                                                         const float *__restrict__ B,
                                                                         float *__restrict__ C, unsigned N, unsigned K,
                                                                         unsigned M, unsigned tiles_n, unsigned tiles_m,
                                                                         unsigned kBand) {
  using G = GeoF32PP;
  constexpr int TM = G::TM, TN = G::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned group = wave >> 2, wq = wave & 3u;       // waves w and w + 4 share a SIMD
  const unsigned wm = wq, wn = group;                      // 4 x 2 wave grid of 64 x 128 blocks
  const unsigned lo = lane & 31u, hi = lane >> 5;

  const unsigned lin = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  // DMA: 16 A pieces (16 rows x 64 B) + 16 B pieces (one k-row of 1 KiB) per slab, 2 + 2 per wave
  unsigned voff_a[2], voff_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned piece = wave + 8 * i;
    const unsigned row = piece * 16 + lane / 4, pc = lane % 4;
    voff_a[i] = (min(row0 + row, N - 1) - row0) * K * 4 + (pc ^ ((row >> 2) & 3u)) * 16;
    voff_b[i] = piece * M * 4 + (min(col0 + lane * 4, M - 4) - col0) * 4;
  }
  const char *a_base = (const char *)A + (size_t)row0 * K * 4;
  const char *b_base = (const char *)B + (size_t)col0 * 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned U = K / G::BK;
  auto issue = [&](unsigned slab, unsigned buf) {
    const unsigned sl = min(slab, U - 1);
    const char *ap = a_base + (size_t)sl * (G::BK * 4);
    const char *bp = b_base + (size_t)sl * G::BK * M * 4;
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

  // fragments.  A: row = wm*64 + mi*32 + lo, logical chunk = 2*kg + hi, physical = (2*kg) ^ (hi ^ (lo>>2)&3)
  const unsigned ca = hi ^ ((lo >> 2) & 3u);
  const unsigned a_row_byte = (wm * 64 + lo) * (G::BK * 4);
  const unsigned a_off[2] = {a_row_byte + ca * 16, a_row_byte + (ca ^ 2u) * 16};
  // B: k = kg*8 + p + 4*hi, 4 consecutive columns wn*128 + 4*lo
  const unsigned b_off = G::A_BYTES + (4 * hi) * (G::BN * 4) + (wn * 128 + 4 * lo) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[mi][t] = (f32x16)0.0f;

  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // C (+)= accumulators (lane owns 4 consecutive columns of 16 rows per row block); optionally restart them
  auto write_tile = [&](bool accumulate, bool restart) {
    const unsigned ccol = col0 + wn * 128 + 4 * lo;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned row = row0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (ccol < M && row < N) {
          f32x4 *dst = (f32x4 *)(C + (size_t)row * M + ccol);
          f32x4 v;
#pragma unroll
          for (int t = 0; t < TN; ++t) v[t] = acc[mi][t][r];
          if (accumulate) v += *dst;
          *dst = v;
        }
      }
      if (restart) {
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[mi][t] = (f32x16)0.0f;
      }
    }
  };
  bool flushed = false;
  auto phase = [&](auto bufc, unsigned u) {
    constexpr int BUF = decltype(bufc)::value;
    const char *base = smem + BUF * G::STAGE_BYTES;
    f32x4 af[2][TM], bf[2][4];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
      for (int p = 0; p < 4; ++p) bf[kg][p] = *(const f32x4 *)(base + b_off + (kg * 8 + p) * (G::BN * 4));
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[kg][mi] = *(const f32x4 *)(base + a_off[kg] + mi * 32 * (G::BK * 4));
    }
    issue(u + 3, (BUF + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int t = 0; t < TN; ++t)
            acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][mi][p], bf[kg][p][t], acc[mi][t], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    // bound the summation chain: every 4096 k this wave adds its accumulators into its own part of C
    // and restarts them (private to the wave: no barrier involved, the partner keeps the pipe busy)
    if ((u + 1) % G::FLUSH_SLABS == 0 && u + 2 < U) {  // same flush points as the default kernel (none before the last slab)
      write_tile(flushed, true);
      flushed = true;
    }
    sync();
  };

  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  sync();
  const bool shifted = !(VAR & 2) && group == 1;
  if (shifted) sync();
  for (unsigned u = 0; u < U; u += 4) {
    phase(std::integral_constant<int, 0>{}, u);
    if (u + 1 < U) phase(std::integral_constant<int, 1>{}, u + 1);
    if (u + 2 < U) phase(std::integral_constant<int, 2>{}, u + 2);
    if (u + 3 < U) phase(std::integral_constant<int, 3>{}, u + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!shifted && !(VAR & 2)) sync();
  sync();
  write_tile(flushed, false);
}
#undef MM_DMA_PIECE

bool mfma_f32_pp_serves_impl(const Problem &p) {
  return !p.a_transposed && p.k % 16 == 0 && p.k >= 64 && p.m % 4 == 0 && p.m >= 4 && p.n >= 1;
}

template <int VAR>
int launch_f32_pp(hipStream_t s, const Problem &p) {
  using G = GeoF32PP;
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_pp_kernel<VAR>, G::LDS_BYTES, configured)) return e;
  hipLaunchKernelGGL((mfma_f32_pp_kernel<VAR>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s,
                     (const float *)p.a, (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, band_rows());
  return (int)hipGetLastError();
}

template <typename G>
int launch_geo(hipStream_t s, const Problem &p) {
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_kernel<G, false>, G::LDS_BYTES, configured)) return e;
  hipLaunchKernelGGL((mfma_f32_kernel<G, false>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s,
                     (const float *)p.a, (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m,
                     band_rows(G::BM, G::BN, G::MIN_WAVES));
  return (int)hipGetLastError();
}

// A stored K x N: only the default geometry is instantiated for it.
template <typename G>
int launch_geo_at(hipStream_t s, const Problem &p) {
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  static unsigned long long configured = 0;
  if (int e = ensure_dynamic_lds((const void *)mfma_f32_kernel<G, true>, G::LDS_BYTES, configured)) return e;
  hipLaunchKernelGGL((mfma_f32_kernel<G, true>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s,
                     (const float *)p.a, (const float *)p.b, (float *)p.c, p.n, p.k, p.m, tiles_n, tiles_m, band_rows());
  return (int)hipGetLastError();
}

//                TM WM WN BK NS two-level KC(slabs)
using V0 = Geo<2, 2, 2, 32, 2, true, 16>;    // 128x256, 4 waves, 512-deep chunks
using V1 = Geo<2, 2, 2, 32, 2, false, 1>;    // same, single chain
using V2 = Geo<4, 2, 2, 16, 2, false, 1>;    // 256x256, 4 waves, 256 accumulators
using V3 = Geo<2, 4, 2, 16, 2, false, 1>;    // 256x256, 8 waves (2 per SIMD)
using V4 = Geo<2, 2, 2, 32, 3, true, 16>;    // V0 with a 3-deep ring and counted vmcnt
using V5 = Geo<2, 2, 2, 16, 2, true, 32>;    // V0 with BK = 16
using V6 = Geo<4, 2, 2, 16, 3, false, 1>;    // V2 with a 3-deep ring
using V7 = Geo<1, 4, 2, 32, 2, true, 16>;    // 128x256, 8 waves, 32x128 per wave
using V8 = Geo<2, 4, 2, 16, 2, false, 1, 256, false, false, 3>;  // 256x256x16, 8 waves, flush into C every 4096 k, fragment reads software-pipelined (HINT 3)
using V9 = Geo<2, 4, 2, 16, 2, false, 1, 128>;  // V3 + flush into C every 2048 k
using V10 = Geo<2, 4, 2, 32, 2, false, 1, 128>; // 256x256x32, 8 waves (144 KiB LDS), flush every 4096 k
using V11 = Geo<2, 4, 2, 16, 3, false, 1, 256>; // V8 with a 3-slab ring (96 KiB)
using V12 = Geo<2, 4, 2, 16, 4, false, 1, 256>; // V8 with a 4-slab ring (128 KiB)
using V13 = Geo<2, 2, 2, 32, 2, false, 1, 128>; // 128x256x32, 4 waves, flush every 4096 k (mid-size shapes)
using V14 = Geo<1, 4, 1, 32, 2, false, 1, 128>; // 128x128x32, 4 waves of 32x128, 64 KiB LDS: 2 workgroups per CU (small shapes)
using V15 = Geo<2, 2, 2, 16, 2, false, 1, 256>; // 128x256x16, 4 waves, 48 KiB LDS: 2 independent workgroups per CU
using V16 = Geo<2, 4, 2, 16, 2, false, 1, 256, true>;  // V8 + staggered DMA issue between SIMD partners
using V17 = Geo<2, 4, 2, 32, 2, false, 1, 128, true>;  // V10 (256x256x32) + stagger
using V18 = Geo<2, 4, 2, 16, 2, false, 1, 256, false, true>;  // V8 + pinned fragment reads
using V19 = Geo<2, 4, 2, 16, 2, false, 1, 256, true, true>;   // V8 + stagger + pinned fragment reads
using V22 = Geo<2, 4, 2, 16, 2, false, 1, 256, false, false, 1>;  // V8 + sched_group_barrier shaping, reads first
using V23 = Geo<2, 4, 2, 16, 2, false, 1, 256, false, false, 2>;  // V8 + shaping, reads mid-group
using V24 = Geo<2, 4, 2, 16, 2, false, 1, 256>;                     // V8 as first shipped: the compiler places the fragment reads (it sinks them to their uses)
using V25 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 3>;    // V3 (no flush) + the same
using V26 = Geo<2, 4, 2, 16, 3, false, 1, 0, false, false, 3>;    // V25 with a ring of 3 (two slabs of DMA in flight)
using V27 = Geo<2, 4, 2, 16, 4, false, 1, 0, false, false, 3>;    // V25 with a ring of 4
using V28 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 4>;    // ABLATION: V25 without the slab barrier (races; timing only)
using V29 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 5>;    // ABLATION: V25 without DMA in the main loop
using V30 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 6>;    // ABLATION: V25 without fragment reads in the main loop
using V31 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 7>;    // ABLATION: V25 with neither
using V32 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 8>;    // ABLATION: V25, DMA issued with exec = 0
using V33 = Geo<2, 2, 2, 16, 2, false, 1, 256, false, false, 3>;  // V15 (128x256x16, 4 waves, 2 workgroups per CU) + pipelined reads
using V34 = Geo<2, 2, 2, 16, 3, false, 1, 256, false, false, 3>;  // the same with a ring of 3 (72 KiB LDS)
using V35 = Geo<1, 4, 1, 32, 2, false, 1, 128, false, false, 3>;  // V14 (128x128x32, 2 workgroups per CU) + pipelined reads
using V8S = Geo<2, 4, 2, 16, 2, false, 1, 256, false, false, 9>; // V8 with the DMA in its scalar-base form: what variant 8 runs for K >= 16
using V36 = V8;                                                   // variant 36 pins the builtin (vector-address) DMA form of V8
using V37 = Geo<2, 4, 2, 16, 2, false, 1, 0, false, false, 9>;    // V25 (no flush) with scalar-base DMA
using V33S = Geo<2, 2, 2, 16, 2, false, 1, 256, false, false, 9>; // V33 with scalar-base DMA: what variant 33 runs for K >= 16
using V34S = Geo<2, 2, 2, 16, 3, false, 1, 256, false, false, 9>; // V33S with a ring of 3 (72 KiB LDS per workgroup)
using V35S = Geo<1, 4, 1, 32, 2, false, 1, 128, false, false, 9>; // V35 with scalar-base DMA: what variant 35 runs for K >= 32

}  // namespace

int mfma_f32_num_variants() { return 38; }  // ids 0..37 (28-32 are the ablations: refused unless MM_ABLATIONS=1)
int mfma_f32_variant_id(int index) { return index >= 0 && index < 38 ? index : -1; }

const char *mfma_f32_name(int v) {
  switch (v) {
    case 0: return "mfma_f32_128x256x32_w4_2lvl";
    case 1: return "mfma_f32_128x256x32_w4";
    case 2: return "mfma_f32_256x256x16_w4";
    case 3: return "mfma_f32_256x256x16_w8";
    case 4: return "mfma_f32_128x256x32_w4_2lvl_ns3";
    case 5: return "mfma_f32_128x256x16_w4_2lvl";
    case 6: return "mfma_f32_256x256x16_w4_ns3";
    case 7: return "mfma_f32_128x256x32_w8_2lvl";
    case 8: return "mfma_f32_256x256x16_w8_flush4096";
    case 9: return "mfma_f32_256x256x16_w8_flush2048";
    case 10: return "mfma_f32_256x256x32_w8_flush4096";
    case 11: return "mfma_f32_256x256x16_w8_flush4096_ns3";
    case 12: return "mfma_f32_256x256x16_w8_flush4096_ns4";
    case 13: return "mfma_f32_128x256x32_w4_flush4096";
    case 14: return "mfma_f32_128x128x32_w4_flush4096";
    case 15: return "mfma_f32_128x256x16_w4_flush4096_2percu";
    case 16: return "mfma_f32_256x256x16_w8_flush4096_stagger";
    case 17: return "mfma_f32_256x256x32_w8_flush4096_stagger";
    case 18: return "mfma_f32_256x256x16_w8_flush4096_pin";
    case 19: return "mfma_f32_256x256x16_w8_flush4096_stagger_pin";
    case 20: return "mfma_f32_256x256x16_w8_flush4096_pingpong";
    case 21: return "mfma_f32_256x256x16_w8_flush4096_pingpong_lockstep";
    case 22: return "mfma_f32_256x256x16_w8_flush4096_sgb1";
    case 23: return "mfma_f32_256x256x16_w8_flush4096_sgb2";
    case 24: return "mfma_f32_256x256x16_w8_flush4096_sunkreads";
    case 25: return "mfma_f32_256x256x16_w8_piperead";
    case 26: return "mfma_f32_256x256x16_w8_piperead_ns3";
    case 27: return "mfma_f32_256x256x16_w8_piperead_ns4";
    case 28: return "mfma_f32_256x256x16_w8_piperead_ABLATION_no_barrier";
    case 29: return "mfma_f32_256x256x16_w8_piperead_ABLATION_no_dma";
    case 30: return "mfma_f32_256x256x16_w8_piperead_ABLATION_no_reads";
    case 31: return "mfma_f32_256x256x16_w8_piperead_ABLATION_mfma_only";
    case 32: return "mfma_f32_256x256x16_w8_piperead_ABLATION_dma_exec0";
    case 33: return "mfma_f32_128x256x16_w4x2_flush4096";  // two independent 4-wavefront workgroups per CU
    case 34: return "mfma_f32_128x256x16_w4x2_flush4096_ns3";
    case 35: return "mfma_f32_128x128x32_w4x2_flush4096";
    case 36: return "mfma_f32_256x256x16_w8_flush4096_vdma";
    case 37: return "mfma_f32_256x256x16_w8_sdma";
  }
  return "?";
}

template <typename G> static void geo_of(unsigned *bm, unsigned *bn, unsigned *bk, unsigned *waves) {
  *bm = G::BM; *bn = G::BN; *bk = G::BK; *waves = G::NW;
}
void mfma_f32_geometry(int v, unsigned *bm, unsigned *bn, unsigned *bk, unsigned *waves) {
  switch (v) {
    case 0: return geo_of<V0>(bm, bn, bk, waves);
    case 1: return geo_of<V1>(bm, bn, bk, waves);
    case 2: return geo_of<V2>(bm, bn, bk, waves);
    case 3: return geo_of<V3>(bm, bn, bk, waves);
    case 4: return geo_of<V4>(bm, bn, bk, waves);
    case 5: return geo_of<V5>(bm, bn, bk, waves);
    case 6: return geo_of<V6>(bm, bn, bk, waves);
    case 7: return geo_of<V7>(bm, bn, bk, waves);
    case 9: return geo_of<V9>(bm, bn, bk, waves);
    case 10: return geo_of<V10>(bm, bn, bk, waves);
    case 11: return geo_of<V11>(bm, bn, bk, waves);
    case 12: return geo_of<V12>(bm, bn, bk, waves);
    case 13: return geo_of<V13>(bm, bn, bk, waves);
    case 14: return geo_of<V14>(bm, bn, bk, waves);
    case 15: return geo_of<V15>(bm, bn, bk, waves);
    case 16: return geo_of<V16>(bm, bn, bk, waves);
    case 17: return geo_of<V17>(bm, bn, bk, waves);
    case 18: return geo_of<V18>(bm, bn, bk, waves);
    case 19: return geo_of<V19>(bm, bn, bk, waves);
    case 33: return geo_of<V33>(bm, bn, bk, waves);
    case 34: return geo_of<V34>(bm, bn, bk, waves);
    case 35: return geo_of<V35>(bm, bn, bk, waves);
    case 36: return geo_of<V36>(bm, bn, bk, waves);
    case 37: return geo_of<V37>(bm, bn, bk, waves);
    default: return geo_of<V8>(bm, bn, bk, waves);
  }
}

// Shape-adaptive geometry (variant < 0): the 256x256 tile is the fastest per CU, but a launch runs
// in whole rounds of resident workgroups, so mid-size problems lose up to a round to quantisation
// (6144^3: 576 tiles = 2.25 rounds of 256) and small ones leave CUs idle (2048^3: 64 tiles).  Pick
// the candidate with the smallest estimated time = (workgroups the busiest CU runs) x tile area /
// relative efficiency.
int mfma_f32_auto_variant(const Problem &p) {
  // relative efficiencies at 16384^3 at the end of round 2, all three with the pinned schedule and scalar-base DMA:
  // 128x256 as two 4-wavefront workgroups per CU (33) 152.2 TF, 256x256 / 8 wavefronts (8) 150.8, 128x128x32 (35) 151.2
  // (profiles/r02z_f32_scalar_base_dma.log, r02z_f32_small_tile_scalar_base_dma.log); small and mid-size shapes:
  // r02z_f32_small_shapes_after_scalar_base_dma.log
  static const TileCandidate cands[] = {{33, 128, 256, 2, 1.00}, {8, 256, 256, 1, 0.991}, {35, 128, 128, 2, 0.993}};
  return pick_tile(cands, 3, p.n, p.m);
}

int mfma_f32_resolve(const Problem &p, int variant);
bool mfma_f32_serves(const Problem &p) {
  if (!(p.n >= 1 && p.m >= 4 && p.k >= 8 && p.m % 4 == 0 && p.k % 8 == 0)) return false;
  return !p.a_transposed || (p.n >= 4 && p.n % 4 == 0);  // K x N A is DMA'd in 16-B chunks along N
}

// The scalar-base DMA form addresses a tile's rows with 32-bit byte offsets from a 64-bit base: 256 rows x K x 4 B (and
// 32 k-rows x M x 4 B for B, K x N A) must stay below 4 GiB; absurdly long rows fall back to the vector-address kernels.
static bool sdma_fits(const Problem &p, unsigned bk) {
  const unsigned long long span = 256ull * (p.a_transposed ? 1ull : p.k) * 4ull, spanb = 32ull * (p.m > p.n ? p.m : p.n) * 4ull;
  return p.k >= bk && span < (1ull << 32) && spanb < (1ull << 32);
}

int mfma_f32_splitk(const Problem &, int) { return 1; }   // the lab edition never splits K

int mfma_f32_resolve(const Problem &p, int variant) {
  if (!mfma_f32_serves(p)) return -1;
  if (p.a_transposed) return 8;
  if (variant < 0) variant = mfma_f32_auto_variant(p);
  return variant <= 37 ? variant : -1;
}

int launch_mfma_f32(hipStream_t s, const Problem &p, int variant) {
  if (!mfma_f32_serves(p)) return kErrNotSupported;
  if (p.a_transposed) return sdma_fits(p, 16) ? launch_geo_at<V8S>(s, p) : launch_geo_at<V8>(s, p);
  if (variant < 0) variant = mfma_f32_auto_variant(p);
  switch (variant) {
    case 0: return launch_geo<V0>(s, p);
    case 1: return launch_geo<V1>(s, p);
    case 2: return launch_geo<V2>(s, p);
    case 3: return launch_geo<V3>(s, p);
    case 4: return launch_geo<V4>(s, p);
    case 5: return launch_geo<V5>(s, p);
    case 6: return launch_geo<V6>(s, p);
    case 7: return launch_geo<V7>(s, p);
    case 8: return sdma_fits(p, 16) ? launch_geo<V8S>(s, p) : launch_geo<V8>(s, p);
    case 9: return launch_geo<V9>(s, p);
    case 10: return launch_geo<V10>(s, p);
    case 11: return launch_geo<V11>(s, p);
    case 12: return launch_geo<V12>(s, p);
    case 13: return launch_geo<V13>(s, p);
    case 14: return launch_geo<V14>(s, p);
    case 15: return launch_geo<V15>(s, p);
    case 16: return launch_geo<V16>(s, p);
    case 17: return launch_geo<V17>(s, p);
    case 18: return launch_geo<V18>(s, p);
    case 19: return launch_geo<V19>(s, p);
    case 20: return mfma_f32_pp_serves_impl(p) ? launch_f32_pp<0>(s, p) : launch_geo<V8>(s, p);
    case 21: return mfma_f32_pp_serves_impl(p) ? launch_f32_pp<2>(s, p) : launch_geo<V8>(s, p);
    case 22: return launch_geo<V22>(s, p);
    case 23: return launch_geo<V23>(s, p);
    case 24: return launch_geo<V24>(s, p);
    case 25: return launch_geo<V25>(s, p);
    case 26: return launch_geo<V26>(s, p);
    case 27: return launch_geo<V27>(s, p);
    case 28: return tuning(TUNE_ABLATIONS) == 1 ? launch_geo<V28>(s, p) : kErrNotSupported;
    case 29: return tuning(TUNE_ABLATIONS) == 1 ? launch_geo<V29>(s, p) : kErrNotSupported;
    case 30: return tuning(TUNE_ABLATIONS) == 1 ? launch_geo<V30>(s, p) : kErrNotSupported;
    case 31: return tuning(TUNE_ABLATIONS) == 1 ? launch_geo<V31>(s, p) : kErrNotSupported;
    case 32: return tuning(TUNE_ABLATIONS) == 1 ? launch_geo<V32>(s, p) : kErrNotSupported;
    case 33: return sdma_fits(p, 16) ? launch_geo<V33S>(s, p) : launch_geo<V33>(s, p);
    case 34: return sdma_fits(p, 16) ? launch_geo<V34S>(s, p) : launch_geo<V34>(s, p);
    case 35: return sdma_fits(p, 32) ? launch_geo<V35S>(s, p) : launch_geo<V35>(s, p);
    case 36: return launch_geo<V36>(s, p);
    case 37: return sdma_fits(p, 16) ? launch_geo<V37>(s, p) : launch_geo<V25>(s, p);
  }
  return kErrNotSupported;
}

}  // namespace mm
