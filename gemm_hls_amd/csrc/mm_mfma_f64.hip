// fp64 (Multiply, Add) fast path for gfx950: C[N x M] = A[N x K] . B[K x M], row-major, on
// v_mfma_f64_16x16x4_f64.  Same organisation as the fp32 kernel (mm_mfma_f32.hip: resident output
// tile in accumulation registers for the whole K loop -- kernel/Compute.cpp:58-60 -- A row-panel /
// B column-panel k-slabs DMA'd to an LDS ring -- the role of kernel/Memory.cpp's ReadA/ReadB/FeedB),
// with the fragment shapes of the f64 instruction:
//   operands: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] (one f64 each);
//   result:   4 f64 per lane, column l&15, rows (l>>4) + 4*r   (NOT the f32 row map).
// Workgroup = WM x WN wavefronts, each owning 64 x 64 of C as 4 x 4 accumulators (128 registers).
// K slab = 16 doubles: an A row is 128 B = 8 chunks of 16 B, XOR-swizzled with (row>>1)&7 on the
// DMA source address so that every ds_read_b128 service group touches 16 distinct 16-B slots.
// Fragment reads (all ds_read_b128 = 2 doubles):
//   A: lane reads A[row = l&15][2 consecutive k at k-offset 2*(l>>4)] -> feeds 2 MFMAs of an
//      8-deep k-group, MFMA p using k = {2g+p : g = 0..3};
//   B: lane reads B[k = 2*(l>>4)+p][2 consecutive columns 2*(l&15)..+1] -> feeds 2 column
//      accumulators (accumulator t holds columns 2*j+t), so the epilogue stores 16 B per lane.
// Accumulation: one f64 fma chain per element (k order 0,2,4,6,1,3,5,7 inside each group).
// Edges: N arbitrary, M % 2 == 0, K % 8 == 0 (the reference's own contract for double is
// K % 8 == 0 and M % 8 == 0, host/RunHardware.cpp:50-61); other shapes -> predicated kernels.
#include <cstdlib>

#include "mm_common.h"

namespace mm {
namespace {

using f64x2 = __attribute__((ext_vector_type(2))) double;
using f64x4 = __attribute__((ext_vector_type(4))) double;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int WM_, int WN_, int NS_, bool PIPE_ = true, int TM_ = 4, int TP_ = 2, int MIN_WAVES_ = 2>
struct GeoD {
  static constexpr int WM = WM_, WN = WN_, NS = NS_;
  static constexpr bool PIPE = PIPE_;  // pinned, software-pipelined fragment reads (see the main loop)
  static constexpr int TM = TM_, TP = TP_;           // row tiles x pairs of column tiles of 16 (4 x 2: a 64 x 64 wavefront tile)
  static constexpr int MIN_WAVES = MIN_WAVES_;       // wavefronts per SIMD the kernel is compiled for
  static constexpr int NW = WM * WN, THREADS = NW * 64;
  static constexpr int WTM = TM * 16, WTN = TP * 32;  // the wavefront's tile
  static constexpr int BM = WM * WTM, BN = WN * WTN, BK = 16;
  static constexpr int CPR = 8;                      // 16-B chunks per A row (16 doubles)
  static constexpr int A_BYTES = BM * BK * 8, B_BYTES = BK * BN * 8;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES, LDS_BYTES = NS * STAGE_BYTES;
  static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;
  static constexpr int LA = NA / NW, LB = NB / NW;
  static constexpr int KG = BK / 8;
  static constexpr int BCH = BN / 2;                 // 16-B chunks per B k-row
  static_assert(NA % NW == 0 && NB % NW == 0, "DMA split");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <typename G, bool AT>
__global__ __launch_bounds__(G::THREADS, G::MIN_WAVES) void mfma_f64_kernel(  // 2 wavefronts per SIMD: <= 256 VGPRs, so that the
                                                                   // 4-wavefront geometry really fits twice on a CU
    const double *__restrict__ A,
                                                              const double *__restrict__ B,
                                                              double *__restrict__ C, unsigned N, unsigned K,
                                                              unsigned M, unsigned tiles_n, unsigned tiles_m, unsigned kBand) {
  constexpr int TM = G::TM, TP = G::TP, BK = G::BK, NS = G::NS, CPR = G::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wm = wave / G::WN, wn = wave % G::WN;
  const unsigned lo = lane & 15u, g4 = lane >> 4;

  const unsigned nwg = tiles_n * tiles_m;
  const unsigned lin = xcd_remap(blockIdx.x, nwg);
  const unsigned band = lin / (kBand * tiles_m), within = lin % (kBand * tiles_m);
  const unsigned rows_in_band = min(kBand, tiles_n - band * kBand);
  const unsigned row0 = (band * kBand + within % rows_in_band) * G::BM, col0 = (within / rows_in_band) * G::BN;

  size_t a_row_off[G::LA];
  unsigned a_kchunk[G::LA];
#pragma unroll
  for (int i = 0; i < G::LA; ++i) {
    const unsigned slot = (wave + G::NW * i) * 64 + lane;
    if (AT) {  // A stored K x N: slab is [BK][BM] like B's; a_kchunk = k-row, a_row_off = column offset
      a_kchunk[i] = slot / (G::BM / 2);
      a_row_off[i] = min(row0 + (slot % (G::BM / 2)) * 2, N - 2);
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
    b_krow[i] = slot / G::BCH;
    b_col[i] = min(col0 + (slot % G::BCH) * 2, M - 2);
  }
  auto stage = [&](unsigned buf, unsigned k0) {
    char *base = smem + buf * G::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      const double *src = AT ? A + (size_t)min(k0 + a_kchunk[i], K - 1) * N + a_row_off[i]
                             : A + a_row_off[i] + min(k0 + a_kchunk[i] * 2, K - 2);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (wave + G::NW * i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < G::LB; ++i) {
      const unsigned kr = min(k0 + b_krow[i], K - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(B + (size_t)kr * M + b_col[i]),
                                       (lptr_t)(base + G::A_BYTES + (wave + G::NW * i) * 1024), 16, 0, 0);
    }
  };

  // Scalar-base form of the same DMA (K >= BK): uniform 64-bit base in SGPRs + a constant 32-bit per-lane
  // offset -- one address VGPR per lane instead of two, no per-slab 64-bit VALU address arithmetic.  On the fp32 kernel
  // this removed most of the DMA instructions' issue cost (+2.5 %, mm_mfma_f32.hip).  A slab start past K - BK (beyond
  // the end, or the partial last slab) is clamped to K - BK uniformly: a partial last slab sits in the SECOND half of
  // its buffer (see the last-slab loop).  K < BK never reaches a PIPE geometry (launch_mfma_f64).
  constexpr bool SDMA = G::PIPE;
  unsigned voa[G::LA], vob[G::LB];
  if (SDMA) {
#pragma unroll
    for (int i = 0; i < G::LA; ++i) {
      const unsigned slot = (wave + G::NW * i) * 64 + lane, row = slot / CPR;
      if (AT) voa[i] = a_kchunk[i] * N * 8u + ((unsigned)a_row_off[i] - row0) * 8u;   // K x N: k-row, clamped column
      else voa[i] = (min(row0 + row, N - 1) - row0) * K * 8u + a_kchunk[i] * 16u;
    }
#pragma unroll
    for (int i = 0; i < G::LB; ++i) vob[i] = b_krow[i] * M * 8u + (b_col[i] - col0) * 8u;
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

  // A: row = wm*64 + mi*16 + lo; chunk = (4*kg + g4) ^ swz(lo)
  const unsigned a_swz = (lo >> 1) & (CPR - 1);
  // K x N layout: image [k][row]; a lane reads 2 consecutive rows of k-row 2*g4 + p, so row
  // tiles pair up: tile 2q+t holds rows q*32 + 2*i + t
  const unsigned a_frag_base = AT ? (2 * g4) * (G::BM * 8) + (wm * G::WTM + 2 * lo) * 8 : (wm * G::WTM + lo) * (BK * 8);
  // B: k = kg*8 + 2*g4 + p; col = wn*64 + pair*32 + 2*lo
  const unsigned b_frag_base = G::A_BYTES + (2 * g4) * (G::BN * 8) + (wn * G::WTN + 2 * lo) * 8;

  f64x4 acc[TM][TP][2];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int pr = 0; pr < TP; ++pr) {
      acc[mi][pr][0] = (f64x4)0.0;
      acc[mi][pr][1] = (f64x4)0.0;
    }

  auto load_frags = [&](unsigned buf, int kg, f64x2 (&af)[TM], f64x2 (&bf)[2][TP]) {
    const char *base = smem + buf * G::STAGE_BYTES;
    const unsigned achunk = (((unsigned)(4 * kg) + g4) ^ a_swz) * 16;
    if (AT) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < TM / 2; ++q) {
          const f64x2 v = *(const f64x2 *)(base + a_frag_base + (kg * 8 + p) * (G::BM * 8) + q * 32 * 8);
          af[2 * q][p] = v[0];
          af[2 * q + 1][p] = v[1];
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[mi] = *(const f64x2 *)(base + a_frag_base + mi * 16 * (BK * 8) + achunk);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int pr = 0; pr < TP; ++pr)
        bf[p][pr] = *(const f64x2 *)(base + b_frag_base + (kg * 8 + p) * (G::BN * 8) + pr * 32 * 8);
  };
  auto mfma_group = [&](const f64x2 (&af)[TM], const f64x2 (&bf)[2][TP]) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int pr = 0; pr < TP; ++pr) {
          acc[mi][pr][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi][p], bf[p][pr][0], acc[mi][pr][0], 0, 0, 0);
          acc[mi][pr][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi][p], bf[p][pr][1], acc[mi][pr][1], 0, 0, 0);
        }
  };

  const unsigned num_tiles = (K + BK - 1) / BK;
  constexpr int L = G::LA + G::LB;
  constexpr bool sdma = SDMA;          // the launcher sends K < BK to the non-PIPE geometry
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if constexpr (sdma) {
#pragma unroll
      for (int i = 0; i < L; ++i) dma_piece_s(s, s * BK, i);
    } else {
      stage(s, s * BK);
    }
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * L) : "memory");
  __builtin_amdgcn_s_barrier();

  f64x2 af0[TM], bf0[2][TP], af1[TM], bf1[2][TP];
  load_frags(0, 0, af0, bf0);

  // the two p-halves of a k-group, in mfma_group's order
  auto mfma_half = [&](const f64x2 (&af)[TM], const f64x2 (&bf)[2][TP], int p) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int pr = 0; pr < TP; ++pr) {
        acc[mi][pr][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi][p], bf[p][pr][0], acc[mi][pr][0], 0, 0, 0);
        acc[mi][pr][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi][p], bf[p][pr][1], acc[mi][pr][1], 0, 0, 0);
      }
  };

  // one full slab with a successor; KG == 2: group 0 from set 0, group 1 from set 1 (see f32 kernel)
  const unsigned steady = num_tiles - 1;
  for (unsigned t = 0; t < steady; ++t) {
    const unsigned buf = t % NS;
    if constexpr (G::PIPE) {
      // Pinned order (round 2).  Written plainly ("read the next group, multiply this one, barrier, refill, ...") the
      // machine scheduler moved BOTH groups' MFMAs behind the barrier: per slab the matrix core then waited for 8
      // fragment reads, the barrier and 6 DMA issues in a row (MfmaUtil 92 %).  Here every fragment is requested 16
      // MFMAs before its first use and the barrier sits between two MFMA halves:
      //   16 MFMAs | reads of group 1 | 16 MFMAs | 16 MFMAs | wait + barrier | (MFMA, DMA piece) x L,
      //   (MFMA, read of slab t+1's group 0) x 8, remaining MFMAs
      constexpr int NM = 4 * TM * TP / 2 * 2 / 2, NR = TM + 2 * TP;  // MFMAs per half group (16), reads per group (8)
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(af0, bf0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(buf, 1, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(af0, bf0, 1);
      mfma_half(af1, bf1, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      static_assert(L + NR <= NM, "post-barrier half group too short for the interleave");
      if constexpr (sdma) {
        // the DMA pieces are inline asm (the scheduler cannot classify them): one MFMA, one piece, by hand
        auto mfma_one = [&](int idx) {  // idx-th MFMA of mfma_half(af1, bf1, 1)
          const int mi = idx / (2 * TP), pr = (idx / 2) % TP, h = idx % 2;
          acc[mi][pr][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[mi][1], bf1[1][pr][h], acc[mi][pr][h], 0, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < L; ++i) {
          mfma_one(i);
          dma_piece_s(buf, (t + NS) * BK, i);
          __builtin_amdgcn_sched_barrier(0);
        }
        load_frags((t + 1) % NS, 0, af0, bf0);
#pragma unroll
        for (int i = L; i < NM; ++i) mfma_one(i);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - L - NR, 0);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      stage(buf, (t + NS) * BK);
      load_frags((t + 1) % NS, 0, af0, bf0);
      mfma_half(af1, bf1, 1);
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
      __builtin_amdgcn_sched_group_barrier(0x008, NM - L - NR, 0);
      __builtin_amdgcn_sched_barrier(0);
      continue;
    }
    load_frags(buf, 1, af1, bf1);
    mfma_group(af0, bf0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * L) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(buf, (t + NS) * BK);
    load_frags((t + 1) % NS, 0, af0, bf0);
    mfma_group(af1, bf1);
  }
  {
    const unsigned t = num_tiles - 1;
    const int groups = (int)((K - t * BK) / 8);
    const int shift = sdma ? G::KG - groups : 0;  // scalar-base DMA fetched a partial last slab as the LAST BK k
    for (int kg = 0; kg < groups; ++kg) {
      load_frags(t % NS, kg + shift, af0, bf0);
      mfma_group(af0, bf0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing ring refills (clamped, unread)

  // epilogue: lane owns 2 consecutive columns of rows g4 + 4*r
#pragma unroll
  for (int pr = 0; pr < TP; ++pr) {
    const unsigned ccol = col0 + wn * G::WTN + pr * 32 + 2 * lo;
    if (ccol >= M) continue;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned ri = g4 + 4 * r;
        const unsigned row = row0 + wm * G::WTM + (AT ? (mi >> 1) * 32 + 2 * ri + (mi & 1) : mi * 16 + ri);
        if (row < N) {
          f64x2 v;
          v[0] = acc[mi][pr][0][r];
          v[1] = acc[mi][pr][1][r];
          *(f64x2 *)(C + (size_t)row * M + ccol) = v;
        }
      }
  }
}

using D0 = GeoD<4, 2, 2>;  // 256 x 128 tile, 8 wavefronts (2 per SIMD), 96 KiB LDS
using D1 = GeoD<2, 2, 2>;  // 128 x 128 tile, 4 wavefronts, 64 KiB LDS: two workgroups per CU (small / mid-size shapes)
using D0R1 = GeoD<4, 2, 2, false>;  // the same tiles with the compiler-placed schedule they first shipped with (f64_variant 2 / 3)
using D1R1 = GeoD<2, 2, 2, false>;
// Problems below a round of 128 x 128 tiles (round 3): a 32 x 32 wavefront tile (2 x 1 x 2 accumulators), 64 x 64 per workgroup,
// 32 KiB of LDS, compiled for four wavefronts per SIMD.  Same fma chain per element as the others: identical bits.
using DS = GeoD<2, 2, 2, false, 2, 1, 4>;

}  // namespace

bool mfma_f64_serves(const Problem &p) {
  if (!(p.n >= 1 && p.m >= 2 && p.k >= 8 && p.m % 2 == 0 && p.k % 8 == 0)) return false;
  return !p.a_transposed || (p.n >= 2 && p.n % 2 == 0);
}

template <typename G>
static int launch_d(hipStream_t s, const Problem &p) {
  const unsigned tiles_n = (p.n + G::BM - 1) / G::BM, tiles_m = (p.m + G::BN - 1) / G::BN;
  static unsigned long long configured = 0, configured_at = 0;
  if (p.a_transposed) {
    if (int e = ensure_dynamic_lds((const void *)mfma_f64_kernel<G, true>, G::LDS_BYTES, configured_at)) return e;
    hipLaunchKernelGGL((mfma_f64_kernel<G, true>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s,
                       (const double *)p.a, (const double *)p.b, (double *)p.c, p.n, p.k, p.m, tiles_n, tiles_m,
                       band_rows(G::BM, G::BN, G::BM * G::BN <= 64 * 64 ? 4 : G::BM * G::BN <= 128 * 128 ? 2 : 1));
    return (int)hipGetLastError();
  }
  if (int e = ensure_dynamic_lds((const void *)mfma_f64_kernel<G, false>, G::LDS_BYTES, configured)) return e;
  hipLaunchKernelGGL((mfma_f64_kernel<G, false>), dim3(tiles_n * tiles_m), dim3(G::THREADS), G::LDS_BYTES, s,
                     (const double *)p.a, (const double *)p.b, (double *)p.c, p.n, p.k, p.m, tiles_n, tiles_m,
                       band_rows(G::BM, G::BN, G::BM * G::BN <= 64 * 64 ? 4 : G::BM * G::BN <= 128 * 128 ? 2 : 1));
  return (int)hipGetLastError();
}

int mfma_f64_tile(const Problem &p) {  // 0: 256x128, 1: 128x128, 4: 64x64
  const int v = tuning(TUNE_F64_VARIANT);
  if (v == 4) return 4;
  if (v >= 0) return v & 1;
  // measured (profiles/r02z_f64_pinned_schedule.log): with two workgroups per CU the small tile sustains the same
  // 74.6 TF as the large one at 16384^3; the large one is kept on ties (fewer, larger DMA streams per CU)
  // 64 x 64 (round 3): compiler-placed, four workgroups per CU; efficiency fitted to profiles/r03y_f64_small_tile.txt
  static const TileCandidate cands[] = {{0, 256, 128, 1, 1.00}, {1, 128, 128, 2, 0.995}, {4, 64, 64, 4, 0.90}};
  return pick_tile(cands, 3, p.n, p.m);
}

// One resolver for mm_kernel_name and the launcher: bit 0 = the 128 x 128 tile, bit 1 = the compiler-placed schedule
// with per-lane 64-bit DMA addresses (f64_variant 2 / 3, and every problem beyond the scalar-base DMA's reach).
static int resolve(const Problem &p) {
  if (!mfma_f64_serves(p) || tuning(TUNE_F64_VARIANT) > 4) return -1;   // f64_variant: -1 (by shape), 0 .. 4
  // scalar-base DMA: 32-bit byte offsets inside a tile's rows (256 rows x K x 8 B, 16 k-rows x M x 8 B) and K >= BK
  const bool sdma_fits = p.k >= 16 && 256ull * (p.a_transposed ? 1ull : p.k) * 8ull < (1ull << 32) &&
                         16ull * (p.m > p.n ? p.m : p.n) * 8ull < (1ull << 32);
  const int tile = mfma_f64_tile(p);
  if (tile == 4) return 4;   // its own (per-lane 64-bit) DMA addresses: no reach limit
  return tile | ((tuning(TUNE_F64_VARIANT) >= 2 || !sdma_fits) ? 2 : 0);
}

const char *mfma_f64_name(const Problem &p) {
  static const char *const names[] = {"mfma_f64_256x128x16_w8", "mfma_f64_128x128x16_w4x2", "mfma_f64_256x128x16_w8_compiler_placed",
                                      "mfma_f64_128x128x16_w4x2_compiler_placed", "mfma_f64_64x64x16_w4x4"};
  const int r = resolve(p);
  return r < 0 ? "unsupported" : names[r];
}

int launch_mfma_f64(hipStream_t s, const Problem &p) {
  switch (resolve(p)) {
    case 0: return launch_d<D0>(s, p);
    case 1: return launch_d<D1>(s, p);
    case 2: return launch_d<D0R1>(s, p);
    case 3: return launch_d<D1R1>(s, p);
    case 4: return launch_d<DS>(s, p);
  }
  return kErrNotSupported;
}

}  // namespace mm
