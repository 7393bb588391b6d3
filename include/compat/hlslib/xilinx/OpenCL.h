/*
 * hlslib/xilinx/OpenCL.h (include/compat) -- hlslib::ocl over the MI355X library.
 *
 * The reference's host talks to its device through hlslib's OpenCL wrapper (host/RunHardware.cpp:114-190):
 *
 *     hlslib::ocl::Context context;                                             :116
 *     auto program = context.MakeProgram("MatrixMultiplication_hw.xclbin");     :119
 *     auto aDevice = context.MakeBuffer<MemoryPackK_t, Access::read>(StorageType::DDR, 1, elements);   :122-138
 *     aDevice.CopyFromHost(aMem.cbegin());                                      :142-144
 *     auto kernel = program.MakeKernel("MatrixMultiplicationKernel", aDevice, bDevice, cDevice, N, K, M);   :148-154
 *     const auto elapsed = kernel.ExecuteTask();    // elapsed.first = seconds  :162,174-176
 *     cDevice.CopyToHost(cMem.begin());                                         :189
 *
 * This header gives exactly those names over include/mm_gemm.h, so that host/RunHardware.cpp, test/TestSimulation.cpp
 * and src/PrintSpecifications.cpp compile UNMODIFIED, from where they lie, with one extra include path
 * (-I<this repo>/include/compat -I<this repo>/include) and link against libmm_gemm_amd.so.  The mapping:
 *
 *     Context            mm_init; one HIP device (index 0 unless given)
 *     MakeProgram(path)  nothing to program: the gfx950 code objects live inside the library; the xclbin path is kept
 *                        for error messages only
 *     MakeBuffer<T, A>   mm_alloc / mm_free (RAII, move-only); the memory-bank arguments are accepted and ignored (one
 *                        HBM).  A read-only buffer is filled on the device with the reference generator's distribution
 *                        (uniform [1, 10)) when it is created: the reference runs `verify off` on uninitialised DDR
 *                        (RunHardware.cpp:99,140), which on a GPU could be NaNs or all zeros and change what is timed
 *     CopyFromHost/ToHost  mm_copy_to_device / mm_copy_to_host, whole buffer, blocking
 *     MakeKernel(name, a, b, c[, N, K, M])   binds mm_gemm_launch with the configuration of THIS build: Data_t,
 *                        OperatorMap, OperatorReduce from the build's generated Config.h (include/Config.h.in:15,34-35),
 *                        MM_TRANSPOSED_A -> a K x N A, and kSizeN/K/M when the build has static sizes (3-argument form)
 *     ExecuteTask()      one untimed launch the first time a Kernel object runs (code-object upload, LDS opt-in, clock ramp
 *                        -- per kernel, shape and device; C is pure output, kernel/Top.cpp never reads it, so running twice
 *                        changes nothing), then the timed launch; returns {seconds, seconds} measured with HIP events on the
 *                        launch stream
 *     XCL_EMULATION_MODE=hw_emu (set by RunHardware.cpp:76 for "hw_emu")   the k-ordered kernel (MM_PATH_ORDERED),
 *                        bit-identical to the reference's Naive; anything else: the fast path (MM_PATH_AUTO)
 *     -DMM_HALF_CONTRACT_REFERENCE (or MM_HALF_CONTRACT=reference in the environment)   a half build keeps the REFERENCE's
 *                        arithmetic under "hw" as well: binary16 products accumulated in binary16, k ascending
 *                        (kernel/Compute.cpp:129-133), on the register-tiled k-ordered kernel -- the result its hosts compare
 *                        with exactly (test/TestSimulation.cpp:80-85, host/RunHardware.cpp:214-218).  Default: the matrix
 *                        cores, f32 accumulation, one rounding (more accurate, not equal)
 *
 * Errors surface as hlslib::ocl::RuntimeError / ConfigurationError, both std::runtime_error, so the reference's one catch
 * block (RunHardware.cpp:192-196) prints them and returns 1.  No CPU fallback: without an MI355X the Context throws.
 */
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

#include "Config.h" /* the gemm_hls build's generated configuration (CMakeLists.txt:136): Data_t, OperatorMap, OperatorReduce */
#include "hlslib/xilinx/DataPack.h"
#ifndef MM_GEMM_NO_KERNEL_SYMBOL
#define MM_GEMM_NO_KERNEL_SYMBOL /* a gemm_hls build declares MatrixMultiplicationKernel itself, with its pack types
                                    (include/MatrixMultiplication.h:155-171): that declaration is the one in force */
#endif
#include "mm_gemm.h"

namespace hlslib {
namespace ocl {

class RuntimeError : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};
class ConfigurationError : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

enum class Access { read, write, readWrite };
enum class StorageType { DDR, HBM };
enum class MemoryBank { unspecified, bank0, bank1, bank2, bank3 }; /* hlslib's older bank form; ignored like the (type, index) pair */

/* std::vector allocator with a fixed alignment: include/Utility.h:48, host/RunHardware.cpp:94-97 (4096 bytes). */
template <typename T, std::size_t alignment>
class AlignedAllocator {
 public:
  using value_type = T;
  template <typename U>
  struct rebind {
    using other = AlignedAllocator<U, alignment>;
  };
  AlignedAllocator() noexcept = default;
  template <typename U>
  AlignedAllocator(AlignedAllocator<U, alignment> const &) noexcept {}
  T *allocate(std::size_t count) {
    const std::size_t bytes = (count * sizeof(T) + alignment - 1) / alignment * alignment;
    void *p = std::aligned_alloc(alignment, bytes ? bytes : alignment);
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, std::size_t) noexcept { std::free(p); }
  template <typename U>
  bool operator==(AlignedAllocator<U, alignment> const &) const noexcept { return true; }
  template <typename U>
  bool operator!=(AlignedAllocator<U, alignment> const &) const noexcept { return false; }
};

namespace detail {

inline void Check(int status, const char *what) {
  if (status != MM_OK) throw RuntimeError(std::string(what) + ": " + mm_last_error());
}

template <typename T> struct ElementOf { using type = T; };
template <typename T, int W> struct ElementOf<DataPack<T, W>> { using type = T; };

template <typename T> struct DTypeOf;
template <> struct DTypeOf<float> { static constexpr mm_dtype_t value = MM_DTYPE_F32; };
template <> struct DTypeOf<double> { static constexpr mm_dtype_t value = MM_DTYPE_F64; };
template <> struct DTypeOf<half> { static constexpr mm_dtype_t value = MM_DTYPE_F16; };
template <> struct DTypeOf<signed char> { static constexpr mm_dtype_t value = MM_DTYPE_I8; };
template <> struct DTypeOf<char> { static constexpr mm_dtype_t value = std::is_signed<char>::value ? MM_DTYPE_I8 : MM_DTYPE_U8; };
template <> struct DTypeOf<unsigned char> { static constexpr mm_dtype_t value = MM_DTYPE_U8; };
template <> struct DTypeOf<short> { static constexpr mm_dtype_t value = MM_DTYPE_I16; };
template <> struct DTypeOf<unsigned short> { static constexpr mm_dtype_t value = MM_DTYPE_U16; };
template <> struct DTypeOf<int> { static constexpr mm_dtype_t value = MM_DTYPE_I32; };
template <> struct DTypeOf<unsigned> { static constexpr mm_dtype_t value = MM_DTYPE_U32; };
template <> struct DTypeOf<long> { static constexpr mm_dtype_t value = MM_DTYPE_I64; };
template <> struct DTypeOf<unsigned long> { static constexpr mm_dtype_t value = MM_DTYPE_U64; };

/* the configuration this translation unit's gemm_hls build stands for */
inline mm_config_t BuildConfig() {
  mm_config_t cfg;
  cfg.dtype = DTypeOf<::Data_t>::value;
  cfg.map_op = ::OperatorMap::code;
  cfg.reduce_op = ::OperatorReduce::code;
  const char *mode = std::getenv("XCL_EMULATION_MODE");
  cfg.path = (mode && std::strcmp(mode, "hw_emu") == 0) ? MM_PATH_ORDERED : MM_PATH_AUTO;
#ifdef MM_TRANSPOSED_A
  cfg.layout_a = MM_A_TRANSPOSED;
#else
  cfg.layout_a = MM_A_ROW_MAJOR;
#endif
  return cfg;
}

/* seeds of the device-side fills: ONE counter for every Buffer instantiation, so that A and B never get the same stream
 * (a per-template static would start both MemoryPackK_t and MemoryPackM_t buffers at the same seed) */
inline unsigned long long NextFillSeed() {
  static unsigned long long seed = 0x5eed;
  return ++seed;
}

}  // namespace detail

template <typename T, Access access>
class Buffer {
 public:
  using Element_t = typename detail::ElementOf<T>::type;

  Buffer() = default;
  Buffer(int device, std::size_t elements) : device_(device), elements_(elements) {
    detail::Check(mm_alloc(device_, bytes(), &pointer_), "MakeBuffer (mm_alloc)");
    if (access == Access::read && elements_ > 0) {
      const int rc = mm_fill_device(device_, detail::DTypeOf<Element_t>::value, pointer_, bytes() / sizeof(Element_t), detail::NextFillSeed());
      if (rc != MM_OK) {
        const std::string message = std::string("MakeBuffer (mm_fill_device): ") + mm_last_error();
        (void)mm_free(device_, pointer_);
        pointer_ = nullptr;
        throw RuntimeError(message);
      }
    }
  }
  Buffer(Buffer const &) = delete;
  Buffer &operator=(Buffer const &) = delete;
  Buffer(Buffer &&other) noexcept { swap(other); }
  Buffer &operator=(Buffer &&other) noexcept {
    swap(other);
    return *this;
  }
  ~Buffer() {
    if (pointer_) (void)mm_free(device_, pointer_);
  }

  /* the whole buffer, from / to `nElements()` consecutive host elements starting at the iterator */
  template <typename Iterator>
  void CopyFromHost(Iterator source) {
    static_assert(sizeof(typename std::iterator_traits<Iterator>::value_type) == sizeof(T), "host element type differs from the buffer's");
    if (elements_) detail::Check(mm_copy_to_device(device_, pointer_, &*source, bytes()), "CopyFromHost");
  }
  template <typename Iterator>
  void CopyToHost(Iterator target) {
    static_assert(sizeof(typename std::iterator_traits<Iterator>::value_type) == sizeof(T), "host element type differs from the buffer's");
    if (elements_) detail::Check(mm_copy_to_host(device_, &*target, pointer_, bytes()), "CopyToHost");
  }

  std::size_t nElements() const { return elements_; }
  std::size_t bytes() const { return elements_ * sizeof(T); }
  void *devicePointer() const { return pointer_; }
  int device() const { return device_; }

 private:
  void swap(Buffer &other) noexcept {
    std::swap(device_, other.device_);
    std::swap(elements_, other.elements_);
    std::swap(pointer_, other.pointer_);
  }
  int device_ = 0;
  std::size_t elements_ = 0;
  void *pointer_ = nullptr;
};

class Kernel {
 public:
  Kernel(int device, const void *a, const void *b, void *c, unsigned size_n, unsigned size_k, unsigned size_m)
      : device_(device), a_(a), b_(b), c_(c), n_(size_n), k_(size_k), m_(size_m) {}

  /* {seconds, seconds}: hlslib reports (elapsed by host clock, elapsed by device profiling); RunHardware.cpp:174-180
   * divides the operation count by .first */
  std::pair<double, double> ExecuteTask() {
    const mm_config_t cfg = detail::BuildConfig();
    if (!warmed_up_) {
      detail::Check(mm_gemm_launch(device_, &cfg, a_, b_, c_, n_, k_, m_, nullptr), "ExecuteTask (first launch)");
      warmed_up_ = true;
    }
    double seconds = 0.0;
    detail::Check(mm_gemm_launch(device_, &cfg, a_, b_, c_, n_, k_, m_, &seconds), "ExecuteTask (mm_gemm_launch)");
    return {seconds, seconds};
  }

  /* what the library will run for this kernel object (tools and logs; not part of hlslib) */
  const char *Name() const {
    const mm_config_t cfg = detail::BuildConfig();
    return mm_kernel_name(&cfg, n_, k_, m_);
  }

 private:
  int device_;
  const void *a_, *b_;
  void *c_;
  unsigned n_, k_, m_;
  bool warmed_up_ = false;   /* per Kernel object: another shape / configuration / device pays its own first launch untimed */
};

class Program {
 public:
  Program(int device, std::string path) : device_(device), path_(std::move(path)) {}

  /* dynamic sizes: MakeKernel("MatrixMultiplicationKernel", a, b, c, N, K, M)   (RunHardware.cpp:152-154) */
  template <typename TA, Access AA, typename TB, Access AB, typename TC, Access AC>
  Kernel MakeKernel(std::string const &name, Buffer<TA, AA> &a, Buffer<TB, AB> &b, Buffer<TC, AC> &c, unsigned size_n,
                    unsigned size_k, unsigned size_m) {
    CheckName(name);
    const std::size_t es = sizeof(::Data_t);
    if (a.bytes() < (std::size_t)size_n * size_k * es || b.bytes() < (std::size_t)size_k * size_m * es ||
        c.bytes() < (std::size_t)size_n * size_m * es)
      throw ConfigurationError("MakeKernel: a buffer is smaller than the N x K / K x M / N x M matrix it is bound to");
    return Kernel(device_, a.devicePointer(), b.devicePointer(), c.devicePointer(), size_n, size_k, size_m);
  }

#ifndef MM_DYNAMIC_SIZES
  /* static sizes: MakeKernel("MatrixMultiplicationKernel", a, b, c) with kSizeN/K/M of the build   (RunHardware.cpp:149-150) */
  template <typename TA, Access AA, typename TB, Access AB, typename TC, Access AC>
  Kernel MakeKernel(std::string const &name, Buffer<TA, AA> &a, Buffer<TB, AB> &b, Buffer<TC, AC> &c) {
    return MakeKernel(name, a, b, c, (unsigned)::kSizeN, (unsigned)::kSizeK, (unsigned)::kSizeM);
  }
#endif

  std::string const &path() const { return path_; }

 private:
  void CheckName(std::string const &name) const {
    if (name != "MatrixMultiplicationKernel")
      throw ConfigurationError("MakeKernel: \"" + name + "\" is not in " + path_ +
                               " (this library provides MatrixMultiplicationKernel, kernel/Top.cpp:6-18)");
  }
  int device_;
  std::string path_;
};

class Context {
 public:
  Context() : Context(0) {}
  explicit Context(int device_index) : device_(device_index) {
    int count = 0;
    const int rc = mm_init(&count);
    if (rc != MM_OK) throw ConfigurationError(std::string("no usable device: ") + mm_last_error());
    if (device_ < 0 || device_ >= count)
      throw ConfigurationError("device index " + std::to_string(device_) + " out of range: " + std::to_string(count) + " MI355X visible");
#ifdef MM_HALF_CONTRACT_REFERENCE
    detail::Check(mm_tuning_set("half_contract", 1), "half_contract = reference");
#endif
  }
  /* hlslib's (vendor, device name) form: the names are Xilinx's and mean nothing here */
  Context(std::string const &, std::string const &) : Context(0) {}

  Program MakeProgram(std::string const &path) { return Program(device_, path); }

  template <typename T, Access access>
  Buffer<T, access> MakeBuffer(StorageType, int /* memory bank */, std::size_t elements) {
    return Buffer<T, access>(device_, elements);
  }
  template <typename T, Access access>
  Buffer<T, access> MakeBuffer(std::size_t elements) {
    return Buffer<T, access>(device_, elements);
  }
  template <typename T, Access access>
  Buffer<T, access> MakeBuffer(MemoryBank, std::size_t elements) {
    return Buffer<T, access>(device_, elements);
  }
  template <typename T, Access access, typename Iterator>
  Buffer<T, access> MakeBuffer(StorageType, int /* memory bank */, Iterator begin, Iterator end) {
    Buffer<T, access> buffer(device_, (std::size_t)std::distance(begin, end));
    buffer.CopyFromHost(begin);
    return buffer;
  }

  int device() const { return device_; }

 private:
  int device_;
};

}  // namespace ocl
}  // namespace hlslib
