/*
 * mm_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference
 * (spcl/gemm_hls) semantics for the hot path C = A (map,reduce) B.
 *
 * Nothing under gemm_hls_amd/ (the product) may include, link or dlopen this
 * file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, as the checker.
 *
 * Parity pin: the reference holds no golden vectors on disk;
 * the pin is (seed 5, dist(1,10), A-then-B draw order, comparison rule) plus
 * the reference's own sources compiled into oracle/_ref (see oracle/Makefile,
 * oracle/hlslib_shim) and run against this restatement in
 * tests/test_oracle.py.  RNG golden draws: tests/golden/.
 *
 * What follows which reference lines:
 *   mm_oracle_fill      host/RunHardware.cpp:31-35,99-105 == test/TestSimulation.cpp:42-55,
 *                       include/MatrixMultiplication.h:14 (kSeed = 5)
 *   mm_oracle_naive     include/Utility.h:18-42 (Naive<Map,Reduce>, incl. MM_TRANSPOSED_A
 *                       indexing :31-35), identity() init :29
 *   mm_oracle_compare   test/TestSimulation.cpp:75-92 == host/RunHardware.cpp:208-225,
 *                       include/Utility.h:113-129 (make_signed)
 *   operators           hlslib/xilinx/Operators.h -- THIRD PARTY, ABSENT from /root/reference
 *                       (un-vendored submodule https://github.com/definelicht/hlslib.git,
 *                       commit unrecorded).  Restated from the published library:
 *                       op::Add      Apply a+b,        identity 0
 *                       op::Multiply Apply a*b,        identity 1
 *                       op::And      Apply a&&b,       identity true(1)
 *                       op::Min      Apply min(a,b),   identity numeric_limits<T>::max()
 *                       op::Max      Apply max(a,b),   identity numeric_limits<T>::lowest()
 *                       (call sites: kernel/Compute.cpp:129,133; include/Utility.h:29,37;
 *                       include/Config.h.in:34-35).
 *
 * Plain C99, no dependencies.  Build: see oracle/Makefile (-ffp-contract=off so
 * that multiply-then-add stays two roundings like the reference's g++ -O build).
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- enums: numerically identical to include/mm_gemm.h (checked in tests) ---- */
enum {
  MM_F32 = 0, MM_F64 = 1, MM_F16 = 2, MM_I8 = 3, MM_U8 = 4, MM_I16 = 5,
  MM_U16 = 6, MM_I32 = 7, MM_U32 = 8, MM_I64 = 9, MM_U64 = 10, MM_NUM_DTYPES
};
enum { MM_OP_ADD = 0, MM_OP_MULTIPLY = 1, MM_OP_AND = 2, MM_OP_MIN = 3, MM_OP_MAX = 4, MM_NUM_OPS };

static const size_t kDtypeSize[MM_NUM_DTYPES] = {4, 8, 2, 1, 1, 2, 2, 4, 4, 8, 8};

size_t mm_oracle_dtype_size(int dtype) {
  return (dtype >= 0 && dtype < MM_NUM_DTYPES) ? kDtypeSize[dtype] : 0;
}

/* ======================================================================== */
/* IEEE binary16 <-> double, round-to-nearest-even.  The reference's `half`  */
/* comes from Xilinx hls_half.h (absent); it is IEEE binary16 with RNE.      */
/* A sum/product of two binary16 values is exact in double, so computing in  */
/* double and rounding once gives the correctly rounded binary16 result.     */
/* ======================================================================== */
static double half_to_double(uint16_t h) {
  const int sign = h >> 15, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
  double v;
  if (exp == 0) v = ldexp((double)man, -24);
  else if (exp == 31) v = man ? NAN : INFINITY;
  else v = ldexp((double)(man | 0x400), exp - 25);
  return sign ? -v : v;
}

static uint16_t double_to_half(double d) {
  uint16_t sign = 0;
  if (signbit(d)) { sign = 0x8000; d = -d; }
  if (isnan(d)) return (uint16_t)(sign | 0x7e00);
  if (d >= 65520.0) return (uint16_t)(sign | 0x7c00); /* rounds to inf (65504 + half ulp) */
  if (d == 0.0) return sign;
  int e;
  (void)frexp(d, &e); /* d = f * 2^e, f in [0.5,1) */
  int unb = e - 1;    /* unbiased exponent of leading bit */
  if (unb < -14) unb = -14; /* subnormal: fixed quantum 2^-24 */
  /* quantum = 2^(unb-10); q = d / quantum in [0, 2048] */
  const double q = ldexp(d, 10 - unb);
  double r = nearbyint(q); /* default rounding mode = RNE */
  uint32_t m = (uint32_t)r;
  int bexp = unb + 15;
  if (unb == -14 && m < 0x400) return (uint16_t)(sign | m); /* subnormal / zero */
  if (m == 0x800) { m = 0x400; bexp += 1; }
  if (bexp >= 31) return (uint16_t)(sign | 0x7c00);
  return (uint16_t)(sign | (bexp << 10) | (m & 0x3ff));
}

uint16_t mm_oracle_double_to_half(double d) { return double_to_half(d); }
double mm_oracle_half_to_double(uint16_t h) { return half_to_double(h); }

/* ======================================================================== */
/* Input generator                                                           */
/* std::default_random_engine on libstdc++ == std::minstd_rand0:             */
/*   x <- 16807 * x mod 2147483647, min 1, max 2147483646, seed 5.           */
/* uniform_real_distribution<double>(1,10): libstdc++ generate_canonical     */
/*   <double,53>: R = 2147483646, log2R = 30, m = 2 draws,                   */
/*   u = ((x1-1) + (x2-1)*R) / R^2 ; value = u*(b-a) + a.                    */
/* uniform_int_distribution<unsigned long>(1,10): "downscaling" branch:      */
/*   scaling = 2147483645/10, past = 10*scaling, reject ret>=past,           */
/*   value = ret/scaling + 1.                                                */
/* (bits/random.tcc generate_canonical, bits/uniform_int_dist.h operator())  */
/* ======================================================================== */
typedef struct { uint64_t x; } minstd0_t;
static uint32_t minstd0_next(minstd0_t *g) {
  g->x = (g->x * 16807ull) % 2147483647ull;
  return (uint32_t)g->x;
}

static double draw_real_1_10(minstd0_t *g) {
  const double R = 2147483646.0; /* max - min + 1, exactly representable */
  double sum = 0.0, tmp = 1.0;
  for (int k = 0; k < 2; ++k) {
    sum += (double)(minstd0_next(g) - 1u) * tmp;
    tmp *= R;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret * (10.0 - 1.0) + 1.0;
}

static uint64_t draw_int_1_10(minstd0_t *g) {
  const uint64_t urngrange = 2147483646ull - 1ull;
  const uint64_t uerange = 10ull;
  const uint64_t scaling = urngrange / uerange;
  const uint64_t past = uerange * scaling;
  uint64_t ret;
  do { ret = (uint64_t)minstd0_next(g) - 1ull; } while (ret >= past);
  return ret / scaling + 1ull;
}

static void store_real(int dtype, void *dst, size_t i, double v) {
  switch (dtype) {
    case MM_F32: ((float *)dst)[i] = (float)v; break;
    case MM_F64: ((double *)dst)[i] = v; break;
    /* Data_t(double) for half: one rounding double -> binary16 */
    case MM_F16: ((uint16_t *)dst)[i] = double_to_half(v); break;
    default: break;
  }
}
static void store_int(int dtype, void *dst, size_t i, uint64_t v) {
  switch (dtype) {
    case MM_I8: ((int8_t *)dst)[i] = (int8_t)v; break;
    case MM_U8: ((uint8_t *)dst)[i] = (uint8_t)v; break;
    case MM_I16: ((int16_t *)dst)[i] = (int16_t)v; break;
    case MM_U16: ((uint16_t *)dst)[i] = (uint16_t)v; break;
    case MM_I32: ((int32_t *)dst)[i] = (int32_t)v; break;
    case MM_U32: ((uint32_t *)dst)[i] = (uint32_t)v; break;
    case MM_I64: ((int64_t *)dst)[i] = (int64_t)v; break;
    case MM_U64: ((uint64_t *)dst)[i] = v; break;
    default: break;
  }
}

/* NB std::is_integral<half> is false AND is_floating_point<half> is false, so the
 * reference's conditional picks the REAL distribution for half (RunHardware.cpp:32-35). */
static int dtype_is_integral(int dtype) { return dtype >= MM_I8 && dtype <= MM_U64; }

/* Fill A (count_a elements) THEN B (count_b elements) from ONE stream, seed 5. */
int mm_oracle_fill(int dtype, void *a, size_t count_a, void *b, size_t count_b) {
  if (dtype < 0 || dtype >= MM_NUM_DTYPES) return 1;
  minstd0_t g = {5};
  void *dst[2] = {a, b};
  size_t cnt[2] = {count_a, count_b};
  for (int which = 0; which < 2; ++which) {
    if (dtype_is_integral(dtype)) {
      for (size_t i = 0; i < cnt[which]; ++i) store_int(dtype, dst[which], i, draw_int_1_10(&g));
    } else {
      for (size_t i = 0; i < cnt[which]; ++i) store_real(dtype, dst[which], i, draw_real_1_10(&g));
    }
  }
  return 0;
}

/* Raw draws for golden-vector tests. */
void mm_oracle_draws_real(double *out, size_t n) {
  minstd0_t g = {5};
  for (size_t i = 0; i < n; ++i) out[i] = draw_real_1_10(&g);
}
void mm_oracle_draws_int(uint64_t *out, size_t n) {
  minstd0_t g = {5};
  for (size_t i = 0; i < n; ++i) out[i] = draw_int_1_10(&g);
}

/* ======================================================================== */
/* Naive<Map,Reduce> -- include/Utility.h:18-42.                             */
/* Loop nest is (n, k, m) with a row of accumulators instead of the          */
/* reference's (n, m, k): every output element still sees                    */
/*   acc = identity; for k = 0..K-1: acc = Reduce(acc, Map(A[n,k], B[k,m]))  */
/* in the same k order with the same two roundings, so results are           */
/* bit-identical to the reference's order; this form merely vectorises.      */
/* Rows are distributed over threads (independent outputs).                  */
/* ======================================================================== */
#define OP_APPLY(op, T, a, b)                                                  \
  ((op) == MM_OP_ADD ? (T)((a) + (b))                                          \
   : (op) == MM_OP_MULTIPLY ? (T)((a) * (b))                                   \
   : (op) == MM_OP_AND ? (T)((a) && (b))                                       \
   : (op) == MM_OP_MIN ? ((b) < (a) ? (b) : (a))                               \
                       : ((a) < (b) ? (b) : (a)))

typedef struct {
  int dtype, map, reduce, transposed_a;
  const void *a, *b;
  void *c;
  size_t n, k, m;
  size_t row_begin, row_end;
} naive_job_t;

#define DEFINE_NAIVE(NAME, T, IDENT_MIN, IDENT_MAX)                              \
  static void NAME(const naive_job_t *j) {                                       \
    const T *A = (const T *)j->a, *B = (const T *)j->b;                          \
    T *C = (T *)j->c;                                                            \
    const size_t K = j->k, M = j->m, N = j->n;                                   \
    const int map = j->map, red = j->reduce;                                     \
    T ident;                                                                     \
    switch (red) {                                                               \
      case MM_OP_ADD: ident = (T)0; break;                                       \
      case MM_OP_MULTIPLY: ident = (T)1; break;                                  \
      case MM_OP_AND: ident = (T)1; break;                                       \
      case MM_OP_MIN: ident = IDENT_MIN; break;                                  \
      default: ident = IDENT_MAX; break;                                         \
    }                                                                            \
    for (size_t n = j->row_begin; n < j->row_end; ++n) {                         \
      T *acc = C + n * M;                                                        \
      for (size_t m = 0; m < M; ++m) acc[m] = ident;                             \
      for (size_t k = 0; k < K; ++k) {                                           \
        const T av = j->transposed_a ? A[k * N + n] : A[n * K + k];              \
        const T *brow = B + k * M;                                               \
        if (map == MM_OP_MULTIPLY && red == MM_OP_ADD) {                         \
          for (size_t m = 0; m < M; ++m) acc[m] = (T)(acc[m] + (T)(av * brow[m]));\
        } else if (map == MM_OP_ADD && red == MM_OP_MIN) {                       \
          for (size_t m = 0; m < M; ++m) {                                       \
            const T s = (T)(av + brow[m]);                                       \
            acc[m] = s < acc[m] ? s : acc[m];                                    \
          }                                                                      \
        } else {                                                                 \
          for (size_t m = 0; m < M; ++m) {                                       \
            const T mapped = OP_APPLY(map, T, av, brow[m]);                      \
            acc[m] = OP_APPLY(red, T, acc[m], mapped);                           \
          }                                                                      \
        }                                                                        \
      }                                                                          \
    }                                                                            \
  }

DEFINE_NAIVE(naive_f32, float, 3.40282346638528859812e+38f, -3.40282346638528859812e+38f)
DEFINE_NAIVE(naive_f64, double, 1.79769313486231570815e+308, -1.79769313486231570815e+308)
DEFINE_NAIVE(naive_i8, int8_t, INT8_MAX, INT8_MIN)
DEFINE_NAIVE(naive_u8, uint8_t, UINT8_MAX, 0)
DEFINE_NAIVE(naive_i16, int16_t, INT16_MAX, INT16_MIN)
DEFINE_NAIVE(naive_u16, uint16_t, UINT16_MAX, 0)
DEFINE_NAIVE(naive_i32, int32_t, INT32_MAX, INT32_MIN)
DEFINE_NAIVE(naive_u32, uint32_t, UINT32_MAX, 0)
DEFINE_NAIVE(naive_i64, int64_t, INT64_MAX, INT64_MIN)
DEFINE_NAIVE(naive_u64, uint64_t, UINT64_MAX, 0)

/* binary16: every Apply is one correctly-rounded half operation (see above).
 * accumulate_mode 0: reference semantics -- accumulator is `half`, rounded after every op.
 * accumulate_mode 1: "wide" contract of the MI355X MFMA path -- products and sums kept in
 *                    double, ONE rounding to half at the end (the GPU kernel accumulates the
 *                    exact f16 products in f32; tests allow it 1 half-ulp against this). */
static double half_op(int op, double a, double b) {
  switch (op) {
    case MM_OP_ADD: return a + b;
    case MM_OP_MULTIPLY: return a * b;
    case MM_OP_AND: return (double)((a != 0.0) && (b != 0.0));
    case MM_OP_MIN: return b < a ? b : a;
    default: return a < b ? b : a;
  }
}
static void naive_f16(const naive_job_t *j, int wide) {
  const uint16_t *A = (const uint16_t *)j->a, *B = (const uint16_t *)j->b;
  uint16_t *C = (uint16_t *)j->c;
  const size_t K = j->k, M = j->m, N = j->n;
  double ident;
  switch (j->reduce) {
    case MM_OP_ADD: ident = 0; break;
    case MM_OP_MULTIPLY: ident = 1; break;
    case MM_OP_AND: ident = 1; break;
    case MM_OP_MIN: ident = 65504.0; break;
    default: ident = -65504.0; break;
  }
  double *acc = (double *)malloc(M * sizeof(double));
  double *brow = (double *)malloc(M * sizeof(double));
  for (size_t n = j->row_begin; n < j->row_end; ++n) {
    for (size_t m = 0; m < M; ++m) acc[m] = ident;
    for (size_t k = 0; k < K; ++k) {
      const double av = half_to_double(j->transposed_a ? A[k * N + n] : A[n * K + k]);
      for (size_t m = 0; m < M; ++m) brow[m] = half_to_double(B[k * M + m]);
      if (wide) {
        for (size_t m = 0; m < M; ++m) acc[m] = half_op(j->reduce, acc[m], half_op(j->map, av, brow[m]));
      } else {
        for (size_t m = 0; m < M; ++m) {
          const double mapped = half_to_double(double_to_half(half_op(j->map, av, brow[m])));
          acc[m] = half_to_double(double_to_half(half_op(j->reduce, acc[m], mapped)));
        }
      }
    }
    for (size_t m = 0; m < M; ++m) C[n * M + m] = double_to_half(acc[m]);
  }
  free(acc);
  free(brow);
}

static int g_f16_wide = 0;

static void *naive_thread(void *arg) {
  const naive_job_t *j = (const naive_job_t *)arg;
  switch (j->dtype) {
    case MM_F32: naive_f32(j); break;
    case MM_F64: naive_f64(j); break;
    case MM_F16: naive_f16(j, g_f16_wide); break;
    case MM_I8: naive_i8(j); break;
    case MM_U8: naive_u8(j); break;
    case MM_I16: naive_i16(j); break;
    case MM_U16: naive_u16(j); break;
    case MM_I32: naive_i32(j); break;
    case MM_U32: naive_u32(j); break;
    case MM_I64: naive_i64(j); break;
    case MM_U64: naive_u64(j); break;
    default: break;
  }
  return NULL;
}

static int naive_dispatch(int dtype, int map, int reduce, int transposed_a, const void *a,
                          const void *b, void *c, size_t n, size_t k, size_t m, int nthreads) {
  if (dtype < 0 || dtype >= MM_NUM_DTYPES || map < 0 || map >= MM_NUM_OPS || reduce < 0 ||
      reduce >= MM_NUM_OPS)
    return 1;
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  naive_job_t *jobs = (naive_job_t *)calloc((size_t)nthreads, sizeof(naive_job_t));
  pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; ++t) {
    naive_job_t *j = &jobs[t];
    j->dtype = dtype; j->map = map; j->reduce = reduce; j->transposed_a = transposed_a;
    j->a = a; j->b = b; j->c = c; j->n = n; j->k = k; j->m = m;
    j->row_begin = n * (size_t)t / (size_t)nthreads;
    j->row_end = n * (size_t)(t + 1) / (size_t)nthreads;
  }
  if (nthreads == 1) {
    naive_thread(&jobs[0]);
  } else {
    for (int t = 0; t < nthreads; ++t) pthread_create(&tids[t], NULL, naive_thread, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(tids[t], NULL);
  }
  free(jobs);
  free(tids);
  return 0;
}

/* Reference semantics (half accumulates in half). */
int mm_oracle_naive(int dtype, int map, int reduce, int transposed_a, const void *a, const void *b,
                    void *c, size_t n, size_t k, size_t m, int nthreads) {
  g_f16_wide = 0;
  return naive_dispatch(dtype, map, reduce, transposed_a, a, b, c, n, k, m, nthreads);
}

/* Same, but binary16 accumulates wide and rounds once (documented contract of the MFMA path). */
int mm_oracle_naive_wide(int dtype, int map, int reduce, int transposed_a, const void *a,
                         const void *b, void *c, size_t n, size_t k, size_t m, int nthreads) {
  g_f16_wide = 1;
  const int rc = naive_dispatch(dtype, map, reduce, transposed_a, a, b, c, n, k, m, nthreads);
  g_f16_wide = 0;
  return rc;
}

/* f32 inputs, (Multiply, Add) accumulated in double, result left in double: the
 * "exact" yardstick used to bound f32 error independently of any BLAS blocking. */
int mm_oracle_gemm_f32_in_f64(const float *a, const float *b, double *c, size_t n, size_t k,
                              size_t m, int transposed_a) {
  for (size_t i = 0; i < n; ++i) {
    double *acc = c + i * m;
    for (size_t j = 0; j < m; ++j) acc[j] = 0.0;
    for (size_t kk = 0; kk < k; ++kk) {
      const double av = transposed_a ? a[kk * n + i] : a[i * k + kk];
      const float *brow = b + kk * m;
      for (size_t j = 0; j < m; ++j) acc[j] += av * (double)brow[j];
    }
  }
  return 0;
}

/* ======================================================================== */
/* Comparison rule -- test/TestSimulation.cpp:75-92, host/RunHardware.cpp:208-225 */
/*   floating (float/double): mismatch iff |test-ref| / ref > tol  (tol 1e-3 there; */
/*                            BASELINE.json tightens f32 to 1e-5 -- caller passes it) */
/*   integral AND half:       mismatch iff test != ref                              */
/* Returns the number of mismatches; *first = row*m+col of the first one, or -1.    */
/* ======================================================================== */
long mm_oracle_compare(int dtype, const void *test, const void *ref, size_t n, size_t m,
                       double tol, long *first, double *max_rel) {
  long bad = 0;
  double worst = 0.0;
  if (first) *first = -1;
  const size_t total = n * m;
  for (size_t i = 0; i < total; ++i) {
    int mismatch;
    if (dtype == MM_F32 || dtype == MM_F64) {
      const double t = dtype == MM_F32 ? ((const float *)test)[i] : ((const double *)test)[i];
      const double r = dtype == MM_F32 ? ((const float *)ref)[i] : ((const double *)ref)[i];
      const double rel = fabs(t - r) / r; /* signed division, as the reference does */
      if (rel > worst) worst = rel;
      mismatch = !(rel <= tol);
    } else {
      const size_t sz = kDtypeSize[dtype];
      mismatch = memcmp((const char *)test + i * sz, (const char *)ref + i * sz, sz) != 0;
    }
    if (mismatch) {
      if (!bad && first) *first = (long)i;
      ++bad;
    }
  }
  if (max_rel) *max_rel = worst;
  return bad;
}
