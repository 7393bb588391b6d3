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

// The kernels live in three include files (one translation unit, so that every launch form shares the one tile body and the
// machine code is what it was when this was a single 1 600-line file -- tools/lab/isa_fingerprint.py before and after the split):
#include "mm_mfma_f32_tile.inc"      // Geo, tile_body, mfma_f32_kernel
#include "mm_mfma_f32_small.inc"     // the 64 x 64 geometry
#include "mm_mfma_f32_streamk.inc"   // stream-K kernels, plans and launchers

// ---- launch forms: whole tiles, split-K, the shape-adaptive pick, the K x N pre-pass, dispatch --------------------------------
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
