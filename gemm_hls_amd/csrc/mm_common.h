// Shared device-side vocabulary of the gfx950 kernels: element types, the (map, reduce)
// operator functors and launch descriptors.  Mirrors what the reference gets from
// include/Config.h.in:15,34-35 (Data_t, OperatorMap, OperatorReduce) and from
// hlslib::op::{Add,Multiply,And,Min,Max} (third-party header, absent from the reference tree;
// semantics: Apply(a,b) and identity(), used at kernel/Compute.cpp:129,133 and
// include/Utility.h:29,37).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mm_gemm.h"

namespace mm {

using half_t = _Float16;

// ---- operators -------------------------------------------------------------------------------
template <typename T> struct Limits;
template <> struct Limits<float> {
  __host__ __device__ static constexpr float max() { return 3.40282346638528859812e+38f; }
  __host__ __device__ static constexpr float lowest() { return -3.40282346638528859812e+38f; }
};
template <> struct Limits<double> {
  __host__ __device__ static constexpr double max() { return 1.79769313486231570815e+308; }
  __host__ __device__ static constexpr double lowest() { return -1.79769313486231570815e+308; }
};
template <> struct Limits<half_t> {
  __host__ __device__ static constexpr half_t max() { return (half_t)65504.0f; }
  __host__ __device__ static constexpr half_t lowest() { return (half_t)-65504.0f; }
};
#define MM_INT_LIMITS(T, LO, HI)                                       \
  template <> struct Limits<T> {                                       \
    __host__ __device__ static constexpr T max() { return HI; }        \
    __host__ __device__ static constexpr T lowest() { return LO; }     \
  };
MM_INT_LIMITS(int8_t, INT8_MIN, INT8_MAX)
MM_INT_LIMITS(uint8_t, 0, UINT8_MAX)
MM_INT_LIMITS(int16_t, INT16_MIN, INT16_MAX)
MM_INT_LIMITS(uint16_t, 0, UINT16_MAX)
MM_INT_LIMITS(int32_t, INT32_MIN, INT32_MAX)
MM_INT_LIMITS(uint32_t, 0, UINT32_MAX)
MM_INT_LIMITS(int64_t, INT64_MIN, INT64_MAX)
MM_INT_LIMITS(uint64_t, 0, UINT64_MAX)
#undef MM_INT_LIMITS

template <int OP, typename T> struct Op;
template <typename T> struct Op<MM_OP_ADD, T> {
  __device__ static __forceinline__ T apply(T a, T b) { return (T)(a + b); }
  __host__ __device__ static constexpr T identity() { return (T)0; }
};
template <typename T> struct Op<MM_OP_MULTIPLY, T> {
  __device__ static __forceinline__ T apply(T a, T b) { return (T)(a * b); }
  __host__ __device__ static constexpr T identity() { return (T)1; }
};
template <typename T> struct Op<MM_OP_AND, T> {
  __device__ static __forceinline__ T apply(T a, T b) { return (T)((a != (T)0) && (b != (T)0)); }
  __host__ __device__ static constexpr T identity() { return (T)1; }
};
template <typename T> struct Op<MM_OP_MIN, T> {
  __device__ static __forceinline__ T apply(T a, T b) { return b < a ? b : a; }  // std::min
  __host__ __device__ static constexpr T identity() { return Limits<T>::max(); }
};
template <typename T> struct Op<MM_OP_MAX, T> {
  __device__ static __forceinline__ T apply(T a, T b) { return a < b ? b : a; }  // std::max
  __host__ __device__ static constexpr T identity() { return Limits<T>::lowest(); }
};

// ---- problem descriptor handed to every kernel launcher --------------------------------------
struct Problem {
  const void *a;   // N x K row-major (or K x N when a_transposed)
  const void *b;   // K x M row-major
  void *c;         // N x M row-major, pure output
  unsigned n, k, m;
  bool a_transposed;
  // rows of the whole job when this launch is one row slab of it (the host-pointer pipeline, the N-split over GPUs);
  // 0 = n.  Decisions that change the summation order (split-K) are taken on the whole job, so that a row's bits do
  // not depend on how the rows were dealt out.
  unsigned n_total = 0;
};

// Launchers (one translation unit each).  Return hipError_t as int; hipErrorNotSupported (801)
// means "this family does not serve this (config, shape)".
int launch_ordered(hipStream_t s, const mm_config_t &cfg, const Problem &p);
int launch_half_wide(hipStream_t s, const Problem &p);  // half (x,+), f32 accumulate, any shape
int launch_valu_tile(hipStream_t s, const mm_config_t &cfg, const Problem &p);
bool valu_tile_serves(const mm_config_t &cfg, const Problem &p);
// The same register-tiled kernels under the k-ORDERED contract ("ordered_tile"): floating-point types from the unit compiled
// with contraction off and Op<> to the letter (mm_valu_tile_fp_exact.hip); the integer types' kernels are bit-identical to
// Naive as they are.  Serves whenever valu_tile_serves() and the operands are 16-byte aligned.
int launch_valu_tile_exact(hipStream_t s, const mm_config_t &cfg, const Problem &p);
int launch_mfma_f32(hipStream_t s, const Problem &p, int variant);
int launch_mfma_f64(hipStream_t s, const Problem &p);
int launch_mfma_f16(hipStream_t s, const Problem &p);
int launch_mfma_i8(hipStream_t s, const Problem &p);
int launch_mfma_f32_split(hipStream_t s, const Problem &p, int variant);  // MM_PATH_SPLIT (mm_mfma_f32_split.hip)
bool mfma_f32_split_serves(const Problem &p);
size_t mfma_f32_split_workspace_bytes(const Problem &p);
int mfma_f32_split_tile(const Problem &p, int variant);  // 256 or 128
int workspace_pool(int device, hipMemPool_t *pool);      // the library-owned, stream-ordered workspace pool of `device` (mm_capi.hip)
int workspace_release(int device);                       // hands its cached memory back to the driver
// `bytes` of epoch flags for one stream-K launch on `stream` (free with hipFreeAsync), and the launch's epoch: mm_capi.hip
int flags_alloc(int device, hipStream_t stream, size_t bytes, void **flags, unsigned long long *epoch);
int device_compute_units(int device);                    // as reported by the device when the library initialised
int mfma_f32_splitk(const Problem &p, int variant);      // K chunks the fp32 MFMA launcher uses for (problem, resolved variant)
bool mfma_f32_serves(const Problem &p);
bool mfma_f64_serves(const Problem &p);
bool mfma_f16_serves(const Problem &p);
bool mfma_i8_serves(const Problem &p);
const char *mfma_f32_name(int variant);
void mfma_f32_geometry(int variant, unsigned *bm, unsigned *bn, unsigned *bk, unsigned *waves);
int mfma_f32_num_variants();
int mfma_f32_variant_id(int index);                      // the valid f32_variant values, 0 <= index < num_variants
int mfma_f32_resolve(const Problem &p, int variant);     // variant id a (problem, knob) pair runs, -1 = unsupported
const char *mfma_f16_name(const Problem &p);
const char *mfma_f64_name(const Problem &p);
const char *mfma_i8_name(const Problem &p);
int mfma_f32_auto_variant(const Problem &p);
int mfma_f64_tile(const Problem &p);  // 0: 256x128, 1: 128x128
int mfma_f16_tile(const Problem &p);  // 0: 256x256, 4: 128x256  // shape-adaptive pick (variant < 0)
// dst[n][k] = src[k][n] for 1- and 2-byte elements (mm_transpose.hip); N and K multiples of 16 bytes' worth of elements
int launch_transpose_kxn(hipStream_t s, const void *src, void *dst, unsigned K, unsigned N, unsigned elem_size);
bool transposes_first_small(const Problem &p, unsigned elem_size);   // K x N A of half / int8: pre-pass + the row-major default
int launch_fill(hipStream_t s, mm_dtype_t dtype, void *ptr, size_t elements, unsigned long long seed);

constexpr int kErrNotSupported = 801;  // hipErrorNotSupported

// Kernels that need more than 64 KiB of dynamic LDS must opt in once per (function, device).
// `mask` is a per-kernel static bitmask of devices already configured.
inline int ensure_dynamic_lds(const void *func, int bytes, unsigned long long &mask) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit) return 0;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  __atomic_fetch_or(&mask, bit, __ATOMIC_RELEASE);
  return 0;
}

// Tuning knobs (sweeps and ablations only; defaults are the measured best).  Each one is read
// ONCE from its environment variable when the library initialises and can be changed afterwards
// through mm_tuning_set() -- the launch path itself never calls getenv.  -1 = unset.
enum Tunable {
  TUNE_F32_VARIANT = 0,  // MM_F32_VARIANT  geometry of the fp32 MFMA kernel (mm_mfma_f32.hip)
  TUNE_F64_VARIANT,      // MM_F64_VARIANT
  TUNE_F16_VARIANT,      // MM_F16_VARIANT
  TUNE_I8_VARIANT,       // MM_I8_VARIANT
  TUNE_BAND_ROWS,        // MM_BAND_ROWS    tile-rows per rasterisation band
  TUNE_VALU_VARIANT,     // MM_VALU_VARIANT 0 = synchronous valu_tile kernel, else (default) the DMA-staged one
  TUNE_SPLIT_VARIANT,    // MM_SPLIT_VARIANT schedule / product count of the MM_PATH_SPLIT kernel (mm_mfma_f32_split.hip)
  TUNE_F32_SPLITK,       // MM_F32_SPLITK   fp32 MFMA path: -1 by shape (small problems only), 1 never, 2..8 that many K chunks
  TUNE_ABLATIONS,        // MM_ABLATIONS    1 = allow the variants that skip work on purpose (power breakdown
                         //                 measurements; they produce WRONG results and are refused otherwise)
  TUNE_DEBUG_POISON,     // MM_DEBUG_POISON 1 = fill scratch the kernels hand data through (stream-K slots) with NaN before
                         //                 every launch: a read of anything this launch did not write shows up in C
  TUNE_KXN_PREPASS_MIN_M,  // MM_KXN_PREPASS_MIN_M  half / int8 with a K x N A: M from which the transposition pre-pass is taken (-1: 6144)
  TUNE_MD_VIRTUAL_DEVICES, // MM_MD_VIRTUAL_DEVICES  G > 0: mm_gemm_multi_device accepts device_count <= G and deals its logical devices
                           //                 out over the physical ones round-robin (tests: every g > 0 branch on a 1-GPU box)
  TUNE_ORDERED_VARIANT,    // MM_ORDERED_VARIANT  0 = MM_PATH_ORDERED always runs the 64 x 64 kernel of mm_ordered.hip (the cross-check);
                           //                 else (default) the register-tiled k-ordered kernel wherever it serves -- same bits
  TUNE_HALF_CONTRACT,      // MM_HALF_CONTRACT  1 / "reference": half (Multiply, Add) under MM_PATH_AUTO keeps the REFERENCE's arithmetic
                           //                 (binary16 products and binary16 accumulation, k ascending: kernel/Compute.cpp:129-133) on the
                           //                 k-ordered tile kernel instead of the matrix cores' f32 accumulation; 0 / "wide" / unset: f32
  TUNE_COUNT
};
int tuning(Tunable t);  // mm_capi.hip

// Tile rasterisation: after the XCD remap, workgroups are ordered in bands of `band_rows` tile-rows
// (column-major inside a band), so the T = 32 x per_cu workgroups an XCD runs at once cover
// band_rows x (T / band_rows) tiles and share A row-panels / B column-panels in that XCD's L2.  What the XCD then
// pulls through the fabric per round is band_rows x BM rows of A plus (T / band_rows) x BN columns of B: the default is
// the power of two that minimises that sum (ties: the smaller band).  256 x 256 tiles, one per CU: 4 (measured on the
// fp32 kernel: fabric fetch 30 GB per 16384^3 launch vs 51 GB at 8 and 100 GB at 16, profiles/r01_band_rows_sweep.txt);
// 128-row tiles running two workgroups per CU: 8 (round 3: the 128 x 256 fp32 default fetched 50 GB with bands of 4).
inline unsigned band_rows(unsigned bm = 256, unsigned bn = 256, unsigned per_cu = 1) {
  const int v = tuning(TUNE_BAND_ROWS);
  if (v > 0) return (unsigned)v;
  const unsigned t = 32u * per_cu;
  unsigned best = 4, cost = ~0u;
  for (unsigned r = 1; r <= t; r *= 2) {
    const unsigned c = r * bm + (t / r) * bn;
    if (c < cost) { cost = c; best = r; }
  }
  return best;
}

// Shape-adaptive tile choice shared by the MFMA families.  A launch runs in rounds of resident
// workgroups (256 CUs x per_cu), so a big tile loses up to a round to quantisation on mid-size
// problems and leaves CUs idle on small ones.  Estimated time of a candidate ~ (workgroups the
// busiest CU runs) x tile area / relative efficiency; the smallest wins.
struct TileCandidate { int id; unsigned bm, bn, per_cu; double eff; };
inline int pick_tile(const TileCandidate *cands, int count, unsigned n, unsigned m, double *best_time = nullptr) {
  double best = 0;
  int pick = cands[0].id;
  for (int i = 0; i < count; ++i) {
    const TileCandidate &c = cands[i];
    const unsigned long long tiles = (unsigned long long)((n + c.bm - 1) / c.bm) * ((m + c.bn - 1) / c.bn);
    const unsigned long long slots = 256ull * c.per_cu, full = tiles / slots, rem = tiles % slots;
    const double t = ((double)full * c.per_cu + (double)((rem + 255) / 256)) * c.bm * c.bn / c.eff;
    if (best == 0 || t < best * 0.999) { best = t; pick = c.id; }
  }
  if (best_time) *best_time = best;
  return pick;
}

// XCD-aware remap of a 1-D workgroup id: the dispatcher places workgroup b on XCD b % 8
// (observed, used for speed only); give every XCD one contiguous chunk of the tile order so that
// tiles sharing A row-panels / B column-panels meet in the same private L2.  Bijective for any
// grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  constexpr unsigned kXcds = 8;
  const unsigned q = nwg / kXcds, r = nwg % kXcds;
  const unsigned xcd = bid % kXcds, slot = bid / kXcds;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

}  // namespace mm
