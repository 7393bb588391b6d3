// Host-side reference computation and comparison for the host binaries: the role of the
// reference's include/Utility.h (Naive :18-42, CallBLAS :66-103, ReferenceImplementation :105-111,
// make_signed :113-129), written for this repo.  The BLAS is located at run time (dlopen) because
// this image ships no cblas.h; without one it falls back on the naive loop with the same warning
// the reference prints.
#pragma once
#include <dlfcn.h>

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "HostConfig.h"

namespace mmhost {

inline unsigned HostThreads() {
  if (const char *e = std::getenv("MM_HOST_THREADS")) return std::max(1, std::atoi(e));
  const unsigned n = std::thread::hardware_concurrency();
  return n ? n : 1;
}

template <typename F>
void ParallelRows(unsigned rows, F &&body) {
  const unsigned workers = std::min<unsigned>(HostThreads(), std::max(1u, rows));
  std::vector<std::thread> pool;
  for (unsigned w = 0; w < workers; ++w) {
    const unsigned begin = (unsigned)((size_t)rows * w / workers), end = (unsigned)((size_t)rows * (w + 1) / workers);
    pool.emplace_back([=, &body] { body(begin, end); });
  }
  for (auto &t : pool) t.join();
}

// Every output: acc = identity; k ascending: acc = Reduce(acc, Map(a, b)).  Loop nest is
// (row, k, column) over a row of accumulators so that it vectorises; the per-element operation
// sequence is exactly Naive's.  Rows go to host threads (independent outputs).
template <class Map, class Reduce, typename T>
void Naive(const T *a, const T *b, T *c, unsigned size_n, unsigned size_k, unsigned size_m, bool a_transposed = false) {
  ParallelRows(size_n, [=](unsigned r0, unsigned r1) {
    for (unsigned n = r0; n < r1; ++n) {
      T *acc = c + (size_t)n * size_m;
      for (unsigned m = 0; m < size_m; ++m) acc[m] = Reduce::identity();
      for (unsigned k = 0; k < size_k; ++k) {
        const T av = a_transposed ? a[(size_t)k * size_n + n] : a[(size_t)n * size_k + k];
        const T *brow = b + (size_t)k * size_m;
        for (unsigned m = 0; m < size_m; ++m) acc[m] = Reduce::Apply(acc[m], Map::Apply(av, brow[m]));
      }
    }
  });
}

// binary16 with a wide accumulator and ONE final rounding: the documented contract of the MFMA
// half path (the reference accumulates in half and overflows to inf beyond K ~ 2000 on its own
// [1,10) inputs).
inline void NaiveHalfWide(const half *a, const half *b, half *c, unsigned size_n, unsigned size_k, unsigned size_m,
                          bool a_transposed = false) {
  ParallelRows(size_n, [=](unsigned r0, unsigned r1) {
    std::vector<double> acc(size_m);
    for (unsigned n = r0; n < r1; ++n) {
      std::fill(acc.begin(), acc.end(), 0.0);
      for (unsigned k = 0; k < size_k; ++k) {
        const double av = (double)(a_transposed ? a[(size_t)k * size_n + n] : a[(size_t)n * size_k + k]);
        const half *brow = b + (size_t)k * size_m;
        for (unsigned m = 0; m < size_m; ++m) acc[m] += av * (double)brow[m];
      }
      for (unsigned m = 0; m < size_m; ++m) c[(size_t)n * size_m + m] = (half)acc[m];
    }
  });
}

// ---- BLAS by dlopen ---------------------------------------------------------------------------
struct Blas {
  using sgemm_t = void (*)(int, int, int, int, int, int, float, const float *, int, const float *, int, float, float *, int);
  using dgemm_t = void (*)(int, int, int, int, int, int, double, const double *, int, const double *, int, double, double *, int);
  sgemm_t sgemm = nullptr;
  dgemm_t dgemm = nullptr;
  std::string name;
  static Blas &Get() {
    static Blas blas = [] {
      Blas b;
#ifdef MM_DISABLE_BLAS_AT_BUILD  // -DMM_ENABLE_BLAS=OFF (CMakeLists.txt:14 in the reference)
      return b;
#endif
      if (std::getenv("MM_DISABLE_BLAS")) return b;
      struct Cand { const char *path, *s, *d; };
      std::vector<Cand> cands;
      if (const char *e = std::getenv("MM_BLAS_LIBRARY")) cands.push_back({e, "cblas_sgemm", "cblas_dgemm"});
      cands.push_back({"/opt/conda/lib/libmkl_rt.so", "cblas_sgemm", "cblas_dgemm"});
      cands.push_back({"libmkl_rt.so", "cblas_sgemm", "cblas_dgemm"});
      cands.push_back({"libopenblas.so", "cblas_sgemm", "cblas_dgemm"});
      cands.push_back({"libopenblas.so.0", "cblas_sgemm", "cblas_dgemm"});
      cands.push_back({"libcblas.so", "cblas_sgemm", "cblas_dgemm"});
      for (auto &cnd : cands) {
        if (void *h = dlopen(cnd.path, RTLD_NOW | RTLD_GLOBAL)) {
          b.sgemm = (sgemm_t)dlsym(h, cnd.s);
          b.dgemm = (dgemm_t)dlsym(h, cnd.d);
          if (b.sgemm && b.dgemm) { b.name = cnd.path; return b; }
          b.sgemm = nullptr; b.dgemm = nullptr;
        }
      }
      return b;
    }();
    return blas;
  }
};
constexpr int kCblasRowMajor = 101, kCblasNoTrans = 111, kCblasTrans = 112;

// ReferenceImplementation: BLAS for (Multiply, Add) on float/double, Naive otherwise
// (include/Utility.h:66-111).  `a_transposed`: A is K x N (MM_TRANSPOSED_A); the BLAS call then takes
// CblasTrans with lda = size_n -- the reference passes lda = size_k there (include/Utility.h:86-87,
// 99-100), which is only right when N == K; this is the corrected call.
// `half_accumulate`: for Data_t = half, use the reference's own semantics (Naive accumulating in
// binary16) instead of the wide-accumulate contract of the fast path: what RunHardware's hw_emu
// mode (MM_PATH_ORDERED) reproduces bit for bit.
template <typename T, class Map, class Reduce>
void ReferenceImplementation(const T *a, const T *b, T *c, unsigned size_n, unsigned size_k, unsigned size_m,
                             bool a_transposed = false, bool half_accumulate = false) {
  constexpr bool mul_add = Map::code == MM_OP_MULTIPLY && Reduce::code == MM_OP_ADD;
  const int trans_a = a_transposed ? kCblasTrans : kCblasNoTrans;
  const int lda = a_transposed ? size_n : size_k;
  if constexpr (mul_add && std::is_same<T, float>::value) {
    if (Blas::Get().sgemm) {
      std::cout << "Running BLAS...\n" << std::flush;
      Blas::Get().sgemm(kCblasRowMajor, trans_a, kCblasNoTrans, size_n, size_m, size_k, 1.0f, a, lda, b, size_m, 0.0f, c,
                        size_m);
      return;
    }
  }
  if constexpr (mul_add && std::is_same<T, double>::value) {
    if (Blas::Get().dgemm) {
      std::cout << "Running BLAS...\n" << std::flush;
      Blas::Get().dgemm(kCblasRowMajor, trans_a, kCblasNoTrans, size_n, size_m, size_k, 1.0, a, lda, b, size_m, 0.0, c,
                        size_m);
      return;
    }
  }
  if constexpr (mul_add && IsHalf<T>::value) {
    if (!half_accumulate) {
      std::cout << "Running wide-accumulate half reference (f32-accumulate contract of the MFMA path)...\n" << std::flush;
      NaiveHalfWide(a, b, c, size_n, size_k, size_m, a_transposed);
      return;
    }
    std::cout << "Running half-accumulating reference (the reference's own Naive semantics)...\n" << std::flush;
    Naive<Map, Reduce>(a, b, c, size_n, size_k, size_m, a_transposed);
    return;
  }
  std::cout << "WARNING: BLAS not available, so I'm falling back on a naive implementation. This will take a long time "
               "for large matrix sizes.\n"
            << std::flush;
  Naive<Map, Reduce>(a, b, c, size_n, size_k, size_m, a_transposed);
}

// Comparison rule of test/TestSimulation.cpp:75-92 / host/RunHardware.cpp:208-225:
// floating point -> |test - ref| / ref > tolerance; integral -> any difference.  `half` follows the
// floating rule here with a one-ulp-of-binary16 tolerance (the reference compares half exactly,
// which only a half-accumulating implementation can meet).
template <typename T> double DefaultTolerance() {
  if (const char *e = std::getenv("MM_VERIFY_TOLERANCE")) return std::atof(e);
  if (std::is_same<T, float>::value) return 1e-5;   // BASELINE.json north_star (reference: 1e-3)
  if (std::is_same<T, double>::value) return 1e-12;
  if (IsHalf<T>::value) return 9.8e-4;              // 2^-10
  return 0.0;
}

// returns true on success; prints the reference's mismatch line otherwise.  `exact`: compare bit
// patterns' values exactly whatever the type (the reference's rule for half, test/TestSimulation.cpp:81-85).
template <typename T>
bool Verify(const T *test, const T *ref, unsigned size_n, unsigned size_m, bool exact = false) {
  const double tol = exact ? 0.0 : DefaultTolerance<T>();
  for (size_t i = 0; i < size_n; ++i) {
    for (size_t j = 0; j < size_m; ++j) {
      const T tv = test[i * size_m + j], rv = ref[i * size_m + j];
      bool mismatch;
      if constexpr (std::is_integral<T>::value) {
        mismatch = tv != rv;
      } else {
        const double t = (double)tv, r = (double)rv;
        // |t - r| / |r|: the reference divides by the signed r (test/TestSimulation.cpp:84), which on its own all-positive
        // data is the same thing and on a negative reference value would flag every correct result
        mismatch = (t != r) && !(std::fabs(t - r) / std::fabs(r) <= tol);
      }
      if (mismatch) {
        if constexpr (std::is_integral<T>::value)
          std::cerr << "Mismatch at (" << i << ", " << j << "): " << (long)tv << " vs. " << (long)rv << "\n";
        else
          std::cerr << "Mismatch at (" << i << ", " << j << "): " << (double)tv << " vs. " << (double)rv << "\n";
        return false;
      }
    }
  }
  return true;
}

}  // namespace mmhost
