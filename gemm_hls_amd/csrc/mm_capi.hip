// C ABI of libmm_gemm_amd.so (declared in include/mm_gemm.h): device management, dispatch to
// the gfx950 kernel families, the timed blocking launch that replaces hlslib's
// Kernel::ExecuteTask() (host/RunHardware.cpp:161-162), the N-split multi-device driver and the
// reference's own entry point MatrixMultiplicationKernel (kernel/Top.cpp:6-18).
// There is no CPU compute path in this file or anywhere in this library.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <unordered_map>
#include <string>
#include <vector>

#include "mm_common.h"

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? MM_ERR_NO_DEVICE : MM_ERR_HIP,
              "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

#define MM_HIP(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #call);  \
  } while (0)

// ---- process-wide state: initialised exactly once, read-only afterwards (safe for concurrent
// calls from any number of host threads; g_error above is per thread) -------------------------
std::once_flag g_init_once;
int g_device_count = 0;
int g_device_cus[64] = {};   // compute units each device reports (MI355X in SPX mode: 256; a CPX partition: 32)
int g_init_status = MM_ERR_NO_DEVICE;
char g_init_error[512] = "";

void init_once() {
  auto init_fail = [](const char *fmt, auto... args) {
    snprintf(g_init_error, sizeof(g_init_error), fmt, args...);
    g_device_count = 0;
    g_init_status = MM_ERR_NO_DEVICE;
  };
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return init_fail("hipGetDeviceCount: %s; this library needs an MI355X (gfx950), there is no CPU fallback",
                     e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  // device indices are HIP's, so every visible device must be usable (mask others out with
  // HIP_VISIBLE_DEVICES); a mixed box is reported, not half-used.
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, d);
    if (e != hipSuccess) return init_fail("hipGetDeviceProperties(%d): %s", d, hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return init_fail("device %d is %s; the kernels in this library are built for gfx950 only "
                       "(hide other devices with HIP_VISIBLE_DEVICES)", d, prop.gcnArchName);
    if (d < 64) g_device_cus[d] = prop.multiProcessorCount;
  }
  g_device_count = n;
  g_init_status = MM_OK;
}

int ensure_init() {
  std::call_once(g_init_once, init_once);
  return g_init_status == MM_OK ? MM_OK : fail(g_init_status, "%s", g_init_error);
}

std::mutex g_default_cfg_mutex;
mm_config_t g_default_cfg = {MM_DTYPE_F32, MM_OP_MULTIPLY, MM_OP_ADD, MM_PATH_AUTO, MM_A_ROW_MAJOR};
mm_config_t default_cfg() {
  std::lock_guard<std::mutex> lock(g_default_cfg_mutex);
  return g_default_cfg;
}

// ---- tuning knobs: environment read once, then only mm_tuning_set() changes them --------------
std::once_flag g_tuning_once;
std::atomic<int> g_tuning[mm::TUNE_COUNT];
const char *const kTuneName[mm::TUNE_COUNT] = {"f32_variant", "f64_variant", "f16_variant", "i8_variant", "band_rows",
                                               "valu_variant", "split_variant", "f32_splitk", "ablations", "debug_poison", "kxn_prepass_min_m",
                                               "md_virtual_devices", "ordered_variant", "half_contract"};
const char *const kTuneEnv[mm::TUNE_COUNT] = {"MM_F32_VARIANT", "MM_F64_VARIANT", "MM_F16_VARIANT", "MM_I8_VARIANT",
                                              "MM_BAND_ROWS", "MM_VALU_VARIANT", "MM_SPLIT_VARIANT", "MM_F32_SPLITK", "MM_ABLATIONS", "MM_DEBUG_POISON", "MM_KXN_PREPASS_MIN_M",
                                              "MM_MD_VIRTUAL_DEVICES", "MM_ORDERED_VARIANT", "MM_HALF_CONTRACT"};
void tuning_init() {
  for (int i = 0; i < mm::TUNE_COUNT; ++i) {
    const char *e = getenv(kTuneEnv[i]);
    int v = (e && *e) ? atoi(e) : -1;
    if (i == mm::TUNE_HALF_CONTRACT && e && *e) {   // the one knob with words for values: MM_HALF_CONTRACT=reference | wide
      if (strcmp(e, "reference") == 0) v = 1;
      else if (strcmp(e, "wide") == 0) v = 0;
    }
    g_tuning[i].store(v, std::memory_order_relaxed);
  }
}

int check_device(int device) {
  int rc = ensure_init();
  if (rc) return rc;
  if (device < 0 || device >= g_device_count)
    return fail(MM_ERR_BAD_ARGUMENT, "device %d out of range [0, %d)", device, g_device_count);
  return MM_OK;
}

bool valid_cfg(const mm_config_t *cfg) {
  return cfg && cfg->dtype >= MM_DTYPE_F32 && cfg->dtype <= MM_DTYPE_U64 && cfg->map_op >= MM_OP_ADD &&
         cfg->map_op <= MM_OP_MAX && cfg->reduce_op >= MM_OP_ADD && cfg->reduce_op <= MM_OP_MAX &&
         (cfg->path == MM_PATH_AUTO || cfg->path == MM_PATH_ORDERED || cfg->path == MM_PATH_SPLIT) &&
         (cfg->layout_a == MM_A_ROW_MAJOR || cfg->layout_a == MM_A_TRANSPOSED);
}

// fp32 geometry: 256x256x16, 8 wavefronts, accumulators flushed into C every 4096 k (146 TF at
// 16384^3, max rel err 2.7e-6 over the full matrix: profiles/r01_f32_precision_full_matrix.txt).
// The f32_variant knob selects another one for sweeps (tools/sweep.py); -1 = shape-adaptive choice
// among the flush-capable geometries (mm_mfma_f32.hip).
int f32_variant() { return mm::tuning(mm::TUNE_F32_VARIANT); }
int f32_variant_for(const mm::Problem &p) { return mm::mfma_f32_resolve(p, f32_variant()); }  // what the launcher runs; -1: none

enum Family { FAM_ORDERED, FAM_ORDERED_TILE, FAM_VALU_TILE, FAM_MFMA_F32, FAM_MFMA_F64, FAM_MFMA_F16, FAM_MFMA_I8, FAM_HALF_WIDE,
              FAM_F32_SPLIT, FAM_NONE };

// The family is a property of the JOB: a row slab of a bigger job (Problem::n_total) runs the family the whole job would
// run, so that a split never changes which arithmetic a row gets.  (Row-major slabs qualify for whatever the job
// qualifies for -- the families only ask n >= 1; a K x N A asks for N % 4 / 2 / 8 / 16 == 0, which tile-aligned slabs of
// an N that has it keep, the last one included.)
Family choose(const mm_config_t &cfg, const mm::Problem &slab) {
  mm::Problem p = slab;
  if (slab.n_total) { p.n = slab.n_total; p.n_total = 0; }
  const bool mul_add = cfg.map_op == MM_OP_MULTIPLY && cfg.reduce_op == MM_OP_ADD;
  // The k-ordered contract (Naive's: include/Utility.h:18-42) has two kernels with the same bits: the register-tiled one
  // wherever it serves, the fully predicated 64 x 64 one for everything else (and always under ordered_variant = 0).
  // half (Multiply, Add) under AUTO joins it when the process asked for the REFERENCE's half arithmetic (half_contract = 1:
  // binary16 accumulating in binary16, kernel/Compute.cpp:129-133 -- what its hosts compare with exactly).
  const bool reference_half = cfg.path == MM_PATH_AUTO && mul_add && cfg.dtype == MM_DTYPE_F16 && mm::tuning(mm::TUNE_HALF_CONTRACT) == 1;
  if (cfg.path == MM_PATH_ORDERED || reference_half)
    return mm::tuning(mm::TUNE_ORDERED_VARIANT) != 0 && mm::valu_tile_serves(cfg, p) ? FAM_ORDERED_TILE : FAM_ORDERED;
  if (cfg.path == MM_PATH_SPLIT)  // an explicit request is never re-routed: fp32 (x,+) or nothing
    return mul_add && cfg.dtype == MM_DTYPE_F32 && mm::mfma_f32_split_serves(p) ? FAM_F32_SPLIT : FAM_NONE;
  if (mul_add && cfg.dtype == MM_DTYPE_F32 && mm::mfma_f32_serves(p)) return FAM_MFMA_F32;
  if (mul_add && cfg.dtype == MM_DTYPE_F64 && mm::mfma_f64_serves(p)) return FAM_MFMA_F64;
  // half (x,+) keeps ONE numerical contract under AUTO (exact products, f32 accumulation, one
  // rounding): shapes the matrix-core kernel does not take run the wide-accumulate plain kernel,
  // never the half-accumulating ones (which would overflow to inf beyond K ~ 2000 on [1,10) data)
  if (mul_add && cfg.dtype == MM_DTYPE_F16) return mm::mfma_f16_serves(p) ? FAM_MFMA_F16 : FAM_HALF_WIDE;
  if (mul_add && (cfg.dtype == MM_DTYPE_I8 || cfg.dtype == MM_DTYPE_U8) && mm::mfma_i8_serves(p)) return FAM_MFMA_I8;
  return mm::valu_tile_serves(cfg, p) ? FAM_VALU_TILE : FAM_ORDERED;
}

// The fast families move 16 bytes per lane (global_load_lds_dwordx4, 16-B vector stores): their
// operands must be 16-B aligned, which every allocator gives (hipMalloc 256 B; the reference's
// host vectors are 4096-B aligned, include/Utility.h:48).  An offset view is refused loudly; the
// ordered path takes any element-aligned pointer.
bool aligned16(const mm::Problem &p) {
  return (((uintptr_t)p.a | (uintptr_t)p.b | (uintptr_t)p.c) & 15u) == 0;
}

int dispatch(hipStream_t s, const mm_config_t &cfg, const mm::Problem &p) {
  if (p.n == 0 || p.m == 0) return MM_OK;  // nothing to write
  Family fam = choose(cfg, p);
  if (fam == FAM_NONE)
    return fail(MM_ERR_UNSUPPORTED, "MM_PATH_SPLIT serves float (Multiply, Add) only (got dtype %d, map %d, reduce %d)",
                (int)cfg.dtype, (int)cfg.map_op, (int)cfg.reduce_op);
  if (fam == FAM_ORDERED_TILE && !aligned16(p)) fam = FAM_ORDERED;   // the k-ordered contract takes any element-aligned pointer
  if (fam != FAM_ORDERED && fam != FAM_HALF_WIDE && fam != FAM_F32_SPLIT && !aligned16(p))
    return fail(MM_ERR_BAD_ARGUMENT, "a, b and c must be 16-byte aligned for the fast path (got %p, %p, %p); "
                "use an aligned allocation or MM_PATH_ORDERED", p.a, p.b, p.c);
  // the launchers report hipGetLastError() after their launch: a stale (sticky-until-read) error of the application's own
  // earlier HIP calls on this thread must not be reported as this launch's
  (void)hipGetLastError();
  int e;
  switch (fam) {
    case FAM_MFMA_F32: e = mm::launch_mfma_f32(s, p, f32_variant()); break;
    case FAM_MFMA_F64: e = mm::launch_mfma_f64(s, p); break;
    case FAM_MFMA_F16: e = mm::launch_mfma_f16(s, p); break;
    case FAM_MFMA_I8: e = mm::launch_mfma_i8(s, p); break;
    case FAM_HALF_WIDE: e = mm::launch_half_wide(s, p); break;
    case FAM_F32_SPLIT: e = mm::launch_mfma_f32_split(s, p, mm::tuning(mm::TUNE_SPLIT_VARIANT)); break;
    case FAM_VALU_TILE:
      e = mm::launch_valu_tile(s, cfg, p);
      if (e == mm::kErrNotSupported) e = mm::launch_ordered(s, cfg, p);  // still the GPU
      break;
    case FAM_ORDERED_TILE:
      e = mm::launch_valu_tile_exact(s, cfg, p);
      if (e == mm::kErrNotSupported) e = mm::launch_ordered(s, cfg, p);  // the same bits
      break;
    default: e = mm::launch_ordered(s, cfg, p); break;
  }
  if (e == mm::kErrNotSupported)
    return fail(MM_ERR_UNSUPPORTED, "configuration (dtype %d, map %d, reduce %d) is not compiled into this library",
                (int)cfg.dtype, (int)cfg.map_op, (int)cfg.reduce_op);
  if (e != 0) return hip_fail((hipError_t)e, "kernel launch");
  return MM_OK;
}

int check_problem(const mm_config_t *cfg, const void *a, const void *b, void *c, unsigned n, unsigned k, unsigned m) {
  if (!valid_cfg(cfg)) return fail(MM_ERR_BAD_ARGUMENT, "invalid mm_config_t");
  if (k == 0) return fail(MM_ERR_BAD_ARGUMENT, "size_k must be positive");
  if (n && m && (!a || !b || !c)) return fail(MM_ERR_BAD_ARGUMENT, "null matrix pointer");
  return MM_OK;
}

// hipEvent_t with a destructor: no exit path of the timed launch can leak one
struct Event {
  hipEvent_t e = nullptr;
  ~Event() { if (e) (void)hipEventDestroy(e); }
};

}  // namespace

// One library-owned memory pool per device for stream-ordered workspace (the packed planes of MM_PATH_SPLIT, the
// partial tiles of the fp32 split-K launches): freed workspace stays cached between launches (release threshold
// "never") without touching the process's default pool, whose settings belong to the application.
// mm_release_workspace() trims it.
static hipMemPool_t g_workspace_pool[64] = {};
static std::mutex g_workspace_mu;
static int make_pool(int dev, hipMemPool_t &pool) {   // under g_workspace_mu
  if (pool) return 0;
  hipMemPoolProps props = {};
  props.allocType = hipMemAllocationTypePinned;
  props.handleTypes = hipMemHandleTypeNone;
  props.location.type = hipMemLocationTypeDevice;
  props.location.id = dev;
  hipError_t e = hipMemPoolCreate(&pool, &props);
  if (e != hipSuccess) { pool = nullptr; return (int)e; }
  unsigned long long keep = ~0ull;
  return (int)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
}
int mm::workspace_pool(int dev, hipMemPool_t *out) {
  std::lock_guard<std::mutex> lock(g_workspace_mu);
  hipMemPool_t &pool = g_workspace_pool[dev & 63];
  if (int rc = make_pool(dev, pool)) return rc;
  *out = pool;
  return 0;
}

// Flags of the stream-K launches (mm_mfma_f32_streamk.inc: launch_streamk_arrive).  A flag is "raised" when it holds the
// launch's EPOCH, a process-wide 64-bit count that no two launches share, so flags never have to be lowered -- provided the
// memory they live in has never held anything but zeros and epochs.  Hence a second pool of this library's own that
// serves nothing else, every block of it cleared (stream-ordered) the first time its address is handed out.  That saves
// the fill kernel and its dependency gap in front of every launch (~10 us: 4 % of a 2560^3 product).  While the stream is
// being captured into a graph the block is cleared every time: a replay repeats the epoch it was captured with.
static hipMemPool_t g_flags_pool[64] = {};
static std::unordered_map<unsigned long long, size_t> g_flags_seen[64];   // address -> bytes cleared there
static unsigned long long g_flags_reserved[64] = {};                    // the pool's reserved bytes when last looked at
// Epochs start from a random 64-bit base per process: what is left in recycled memory -- a small integer,
// an index buffer, an epoch of another process -- cannot plausibly equal a live epoch even if a block were ever missed.
static unsigned long long flags_epoch_base() {
  std::random_device rd;
  return (((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count()) | (1ull << 62);
}
static std::atomic<unsigned long long> g_flags_epoch{flags_epoch_base()};
int mm::flags_alloc(int dev, hipStream_t s, size_t bytes, void **out, unsigned long long *epoch) {
  std::lock_guard<std::mutex> lock(g_workspace_mu);
  hipMemPool_t &pool = g_flags_pool[dev & 63];
  if (int rc = make_pool(dev, pool)) return rc;
  hipError_t e = hipMallocFromPoolAsync(out, bytes, pool, s);
  if (e != hipSuccess) return (int)e;
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if ((e = hipStreamIsCapturing(s, &capture)) != hipSuccess) { (void)hipFreeAsync(*out, s); return (int)e; }
  *epoch = g_flags_epoch.fetch_add(1, std::memory_order_relaxed);
  // The pool mapped or unmapped memory since the last look (the runtime may trim a pool by itself when the device runs
  // out of memory): an address seen before may be backed by fresh physical memory now -- forget what was cleared.
  unsigned long long reserved = 0;
  if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &reserved) == hipSuccess && reserved != g_flags_reserved[dev & 63]) {
    g_flags_reserved[dev & 63] = reserved;
    g_flags_seen[dev & 63].clear();
  }
  // (a block handed out during capture is not recorded: whether the graph ever runs and clears it is not known here)
  const bool capturing = capture != hipStreamCaptureStatusNone;
  bool clear = capturing;
  if (!capturing) {
    size_t &cleared = g_flags_seen[dev & 63][(unsigned long long)(size_t)*out];   // 0 when the address is new
    if (cleared < bytes) { cleared = bytes; clear = true; }                        // new, or seen with fewer bytes
  }
  if (clear) {
    if ((e = hipMemsetAsync(*out, 0, bytes, s)) != hipSuccess) { (void)hipFreeAsync(*out, s); return (int)e; }
  }
  return 0;
}

int mm::device_compute_units(int dev) { return dev >= 0 && dev < 64 && g_device_cus[dev] > 0 ? g_device_cus[dev] : 256; }

int mm::workspace_release(int dev) {
  std::lock_guard<std::mutex> lock(g_workspace_mu);
  int rc = 0;
  if (hipMemPool_t pool = g_workspace_pool[dev & 63]) rc = (int)hipMemPoolTrimTo(pool, 0);
  if (hipMemPool_t pool = g_flags_pool[dev & 63]) {
    g_flags_seen[dev & 63].clear();   // what comes back from the driver after this is fresh memory again
    const int rf = (int)hipMemPoolTrimTo(pool, 0);
    if (!rc) rc = rf;
  }
  return rc;
}

int mm::tuning(mm::Tunable t) {
  std::call_once(g_tuning_once, tuning_init);
  return g_tuning[t].load(std::memory_order_relaxed);
}

static void kernel_info_for(const mm_config_t *cfg, const mm::Problem &p, mm_kernel_info_t *info);

extern "C" {

const char *mm_last_error(void) { return g_error; }

int mm_tuning_set(const char *name, int value) {
  std::call_once(g_tuning_once, tuning_init);
  for (int i = 0; name && i < mm::TUNE_COUNT; ++i)
    if (strcmp(name, kTuneName[i]) == 0) { g_tuning[i].store(value, std::memory_order_relaxed); return MM_OK; }
  return fail(MM_ERR_BAD_ARGUMENT, "unknown tuning knob '%s'", name ? name : "(null)");
}

int mm_tuning_get(const char *name, int *value) {
  for (int i = 0; name && value && i < mm::TUNE_COUNT; ++i)
    if (strcmp(name, kTuneName[i]) == 0) { *value = mm::tuning((mm::Tunable)i); return MM_OK; }
  return fail(MM_ERR_BAD_ARGUMENT, "unknown tuning knob '%s'", name ? name : "(null)");
}

size_t mm_dtype_size(mm_dtype_t dtype) {
  static const size_t sz[] = {4, 8, 2, 1, 1, 2, 2, 4, 4, 8, 8};
  return (dtype >= MM_DTYPE_F32 && dtype <= MM_DTYPE_U64) ? sz[dtype] : 0;
}

int mm_init(int *device_count) {
  int rc = ensure_init();
  if (device_count) *device_count = g_device_count;
  return rc;
}

int mm_alloc(int device, size_t bytes, void **device_ptr) {
  if (!device_ptr) return fail(MM_ERR_BAD_ARGUMENT, "device_ptr is null");
  int rc = check_device(device);
  if (rc) return rc;
  MM_HIP(hipSetDevice(device));
  MM_HIP(hipMalloc(device_ptr, bytes ? bytes : 1));
  return MM_OK;
}

int mm_free(int device, void *device_ptr) {
  int rc = check_device(device);
  if (rc) return rc;
  MM_HIP(hipSetDevice(device));
  MM_HIP(hipFree(device_ptr));
  return MM_OK;
}

int mm_release_workspace(int device) {
  int rc = check_device(device);
  if (rc) return rc;
  MM_HIP(hipSetDevice(device));
  MM_HIP(hipDeviceSynchronize());   // stream-ordered frees are complete: everything the pool holds is reclaimable
  MM_HIP((hipError_t)mm::workspace_release(device));
  return MM_OK;
}

int mm_device_pci_bus_id(int device, char *buffer, int length) {
  int rc = check_device(device);
  if (rc) return rc;
  if (!buffer || length < 13) return fail(MM_ERR_BAD_ARGUMENT, "mm_device_pci_bus_id needs a buffer of at least 13 bytes");
  MM_HIP(hipDeviceGetPCIBusId(buffer, length, device));
  return MM_OK;
}

int mm_copy_to_device(int device, void *dst, const void *src, size_t bytes) {
  int rc = check_device(device);
  if (rc) return rc;
  if (bytes && (!dst || !src)) return fail(MM_ERR_BAD_ARGUMENT, "null pointer in copy");
  MM_HIP(hipSetDevice(device));
  MM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return MM_OK;
}

int mm_copy_to_host(int device, void *dst, const void *src, size_t bytes) {
  int rc = check_device(device);
  if (rc) return rc;
  if (bytes && (!dst || !src)) return fail(MM_ERR_BAD_ARGUMENT, "null pointer in copy");
  MM_HIP(hipSetDevice(device));
  MM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return MM_OK;
}

int mm_fill_device(int device, mm_dtype_t dtype, void *ptr, size_t elements, unsigned long long seed) {
  int rc = check_device(device);
  if (rc) return rc;
  if (!mm_dtype_size(dtype) || (elements && !ptr)) return fail(MM_ERR_BAD_ARGUMENT, "bad fill arguments");
  MM_HIP(hipSetDevice(device));
  int e = mm::launch_fill(nullptr, dtype, ptr, elements, seed);
  if (e) return hip_fail((hipError_t)e, "fill launch");
  MM_HIP(hipDeviceSynchronize());
  return MM_OK;
}

int mm_gemm_enqueue(void *hip_stream, const mm_config_t *cfg, const void *a, const void *b, void *c, unsigned n,
                    unsigned k, unsigned m) {
  int rc = ensure_init();
  if (rc) return rc;
  rc = check_problem(cfg, a, b, c, n, k, m);
  if (rc) return rc;
  mm::Problem p{a, b, c, n, k, m, cfg->layout_a == MM_A_TRANSPOSED};
  return dispatch((hipStream_t)hip_stream, *cfg, p);
}

int mm_gemm_launch(int device, const mm_config_t *cfg, const void *a, const void *b, void *c, unsigned n, unsigned k,
                   unsigned m, double *elapsed_seconds) {
  int rc = check_device(device);
  if (rc) return rc;
  rc = check_problem(cfg, a, b, c, n, k, m);
  if (rc) return rc;
  MM_HIP(hipSetDevice(device));
  Event start, stop;
  MM_HIP(hipEventCreate(&start.e));
  MM_HIP(hipEventCreate(&stop.e));
  mm::Problem p{a, b, c, n, k, m, cfg->layout_a == MM_A_TRANSPOSED};
  MM_HIP(hipEventRecord(start.e, nullptr));
  rc = dispatch(nullptr, *cfg, p);
  if (rc != MM_OK) return rc;
  MM_HIP(hipEventRecord(stop.e, nullptr));
  MM_HIP(hipEventSynchronize(stop.e));
  if (elapsed_seconds) {
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, start.e, stop.e));
    *elapsed_seconds = 1e-3 * (double)ms;
  }
  return MM_OK;
}

// Rows per device of the N split: ceil(N / G) rounded up to whole tile rows of the kernel that will run on a slab of that
// height (mm_kernel_info's tile_n: 256 for the large fp32 default, 128 / 64 for the smaller geometries), so that only the last busy
// device owns a ragged tile row.  The one place this arithmetic lives: mm_row_slab() hands it to bench.py and partition.py.
static unsigned md_slab_rows(const mm_config_t &cfg, unsigned n, unsigned k, unsigned m, int device_count) {
  if (n == 0) return 0;
  const size_t base = ((size_t)n + device_count - 1) / device_count;
  // The tile height is a property of the kernel that runs on a slab of the FINAL height, which the rounding itself changes
  // (ADVICE r5): iterate to a fixed point -- tile of a slab of `rows` rows -> rows = ceil(N / G) rounded up to it -> ... -- and
  // if the geometry choice flips back and forth, round to the largest tile any candidate uses (256), which every smaller
  // one divides.  Depends on (configuration, shape, knobs) only: not on the current device, so that every rank of a
  // one-process-per-GPU job computes the same partition.
  auto tile_for = [&](unsigned rows) -> size_t {
    mm_kernel_info_t info = {};
    const mm::Problem p{nullptr, nullptr, nullptr, rows, k, m, cfg.layout_a == MM_A_TRANSPOSED, rows == n ? 0u : n};
    kernel_info_for(&cfg, p, &info);
    return info.tile_n ? info.tile_n : 128;
  };
  unsigned rows = (unsigned)std::min<size_t>(n, (base + 127) / 128 * 128);
  for (int it = 0; it < 4; ++it) {
    const size_t tile = tile_for(rows);
    const unsigned next = (unsigned)std::min<size_t>(n, (base + tile - 1) / tile * tile);
    if (next == rows) return rows;
    rows = next;
  }
  return (unsigned)std::min<size_t>(n, (base + 255) / 256 * 256);
}

int mm_row_slab(const mm_config_t *cfg, unsigned n, unsigned k, unsigned m, int device_count, int rank, unsigned *row0,
                unsigned *rows) {
  if (!valid_cfg(cfg) || !row0 || !rows) return fail(MM_ERR_BAD_ARGUMENT, "invalid arguments to mm_row_slab");
  if (device_count < 1 || rank < 0 || rank >= device_count)
    return fail(MM_ERR_BAD_ARGUMENT, "rank %d not in [0, device_count = %d)", rank, device_count);
  const unsigned slab = md_slab_rows(*cfg, n, k ? k : 1, m, device_count);
  *row0 = (unsigned)std::min<size_t>((size_t)rank * slab, n);
  *rows = std::min(slab, n - *row0);
  return MM_OK;
}

namespace {
// What one logical device of the N split owns; everything is released by the destructor, on every path.
struct SplitDevice {
  int phys = 0;                       // the HIP device this logical device runs on
  void *a = nullptr, *b = nullptr, *c = nullptr;
  unsigned row0 = 0, rows = 0;
  hipStream_t s = nullptr;
  hipEvent_t start = nullptr, stop = nullptr;   // around this device's launch, on its stream
  ~SplitDevice() {
    if (!a && !b && !c && !s && !start && !stop) return;
    (void)hipSetDevice(phys);
    if (s) (void)hipStreamSynchronize(s);
    if (start) (void)hipEventDestroy(start);
    if (stop) (void)hipEventDestroy(stop);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (c) (void)hipFree(c);
    if (s) (void)hipStreamDestroy(s);
  }
};
}  // namespace

int mm_gemm_multi_device(int device_count, const mm_config_t *cfg, const void *a_host, const void *b_host,
                         void *c_host, unsigned n, unsigned k, unsigned m, double *elapsed_seconds) {
  return mm_gemm_multi_device_timed(device_count, cfg, a_host, b_host, c_host, n, k, m, elapsed_seconds, nullptr, nullptr);
}

int mm_gemm_multi_device_timed(int device_count, const mm_config_t *cfg, const void *a_host, const void *b_host,
                               void *c_host, unsigned n, unsigned k, unsigned m, double *elapsed_seconds,
                               double *per_device_seconds, double *host_wall_seconds) {
  int rc = ensure_init();
  if (rc) return rc;
  rc = check_problem(cfg, a_host, b_host, c_host, n, k, m);
  if (rc) return rc;
  // MM_MD_VIRTUAL_DEVICES / "md_virtual_devices" = V > 0 (tests): up to V LOGICAL devices, dealt out over the physical ones
  // round-robin -- on a 1-GPU box all of them on device 0, each with its own stream, slabs and copy of B, the B fan-out
  // taking the peer-copy branch (device 0 -> device 0).  Every line below is the one a real G-GPU node runs.
  const int virt = mm::tuning(mm::TUNE_MD_VIRTUAL_DEVICES);
  const int limit = virt > 0 ? virt : g_device_count;
  if (device_count < 1 || device_count > limit)
    return fail(MM_ERR_BAD_ARGUMENT, "device_count %d not in [1, %d]", device_count, limit);
  if (per_device_seconds) std::fill(per_device_seconds, per_device_seconds + device_count, 0.0);
  if (host_wall_seconds) *host_wall_seconds = 0.0;
  if (n == 0 || m == 0) {   // an empty C: nothing to copy, launch or time (the single-device entry points do the same)
    if (elapsed_seconds) *elapsed_seconds = 0.0;
    return MM_OK;
  }
  const size_t es = mm_dtype_size(cfg->dtype);
  const bool kxn = cfg->layout_a == MM_A_TRANSPOSED;
  const unsigned slab = md_slab_rows(*cfg, n, k, m, device_count);
  std::vector<SplitDevice> devs(device_count);
#define MM_HIP_MD(call)                                       \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return hip_fail(e_, #call);         \
  } while (0)
  // independent row slabs: logical device g owns C[row0 : row0+rows, :] = A[row0 : row0+rows, :] . B
  // (kernel/Compute.cpp:53-60: no outer tile of C depends on another one)
  for (int g = 0; g < device_count; ++g) {
    SplitDevice &d = devs[g];
    d.phys = g % g_device_count;
    d.row0 = (unsigned)std::min<size_t>((size_t)g * slab, n);
    d.rows = std::min(slab, n - d.row0);
    MM_HIP_MD(hipSetDevice(d.phys));
    MM_HIP_MD(hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking));
    if (!d.rows) continue;          // a trailing device without rows takes no part (and no copy of B)
    MM_HIP_MD(hipEventCreate(&d.start));
    MM_HIP_MD(hipEventCreate(&d.stop));
    MM_HIP_MD(hipMalloc(&d.b, (size_t)k * m * es));
    MM_HIP_MD(hipMalloc(&d.a, (size_t)d.rows * k * es));
    MM_HIP_MD(hipMalloc(&d.c, (size_t)d.rows * m * es));
  }
  // B crosses PCIe ONCE (host -> device 0) and is then fanned out device 0 -> device g over xGMI
  // (every GPU has its own link to GPU 0, so the G-1 peer copies run concurrently), instead of G
  // pageable host copies through one root complex; the A slabs go up meanwhile.  A device that
  // cannot take the peer copy gets B from the host as before.
  Event b_on_dev0;
  MM_HIP_MD(hipSetDevice(devs[0].phys));
  MM_HIP_MD(hipEventCreateWithFlags(&b_on_dev0.e, hipEventDisableTiming));
  MM_HIP_MD(hipMemcpyAsync(devs[0].b, b_host, (size_t)k * m * es, hipMemcpyHostToDevice, devs[0].s));
  MM_HIP_MD(hipEventRecord(b_on_dev0.e, devs[0].s));
  for (int g = 0; g < device_count; ++g) {
    SplitDevice &d = devs[g];
    if (!d.rows) continue;
    MM_HIP_MD(hipSetDevice(d.phys));
    if (g > 0) {
      bool peer = d.phys == devs[0].phys;          // the same physical device (virtual devices): a peer copy onto itself
      if (!peer) {
        int can = 0;
        peer = hipDeviceCanAccessPeer(&can, d.phys, devs[0].phys) == hipSuccess && can;
        if (peer) {
          const hipError_t en = hipDeviceEnablePeerAccess(devs[0].phys, 0);
          peer = en == hipSuccess || en == hipErrorPeerAccessAlreadyEnabled;
        }
        (void)hipGetLastError();  // "already enabled" / "no peer" are not errors of this call
      }
      if (peer) {
        MM_HIP_MD(hipStreamWaitEvent(d.s, b_on_dev0.e, 0));
        MM_HIP_MD(hipMemcpyPeerAsync(d.b, d.phys, devs[0].b, devs[0].phys, (size_t)k * m * es, d.s));
      } else {
        MM_HIP_MD(hipMemcpyAsync(d.b, b_host, (size_t)k * m * es, hipMemcpyHostToDevice, d.s));
      }
    }
    if (!kxn) {   // rows [row0, row0 + rows) of a row-major A: one contiguous block
      MM_HIP_MD(hipMemcpyAsync(d.a, (const char *)a_host + (size_t)d.row0 * k * es, (size_t)d.rows * k * es,
                               hipMemcpyHostToDevice, d.s));
    } else {      // the same rows of a K x N A (MM_TRANSPOSED_A, kernel/Memory.cpp:205-261): COLUMNS [row0, row0 + rows) of every
                  // one of its K rows -> a dense K x rows matrix on the device
      MM_HIP_MD(hipMemcpy2DAsync(d.a, (size_t)d.rows * es, (const char *)a_host + (size_t)d.row0 * es, (size_t)n * es,
                                 (size_t)d.rows * es, k, hipMemcpyHostToDevice, d.s));
    }
  }
  for (SplitDevice &d : devs) { MM_HIP_MD(hipSetDevice(d.phys)); MM_HIP_MD(hipStreamSynchronize(d.s)); }
  // untimed warm-up pass of the same launch: the first dispatch on a device loads the code object,
  // opts the kernel into its LDS size and ramps the clocks; paying that outside the timed region
  // makes this figure comparable with mm_gemm_launch's (whose callers warm up the same way).
  // pass 1 is the timed region (copies excluded, as RunHardware.cpp:161-180 does).  Every device's launch is bracketed by HIP
  // events on ITS stream: the job's time is the MAX over devices of that kernel time -- what ExecuteTask() would
  // report on the slowest device -- and the per-device figures say which device that was.  The host clock from the first
  // dispatch to the last completion is kept as a cross-check (it adds G launch latencies and G stream synchronisations).
  double elapsed = 0.0, wall = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const auto t0 = std::chrono::steady_clock::now();
    for (SplitDevice &d : devs) {
      if (!d.rows) continue;
      MM_HIP_MD(hipSetDevice(d.phys));
      // a row slab of the n-row job (n_total 0: the slab IS the job -- the same launch as mm_gemm_launch's)
      const mm::Problem p{d.a, d.b, d.c, d.rows, k, m, kxn, d.rows == n ? 0u : n};
      MM_HIP_MD(hipEventRecord(d.start, d.s));
      rc = dispatch(d.s, *cfg, p);
      if (rc) return rc;
      MM_HIP_MD(hipEventRecord(d.stop, d.s));
    }
    for (SplitDevice &d : devs) { MM_HIP_MD(hipSetDevice(d.phys)); MM_HIP_MD(hipStreamSynchronize(d.s)); }
    wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  for (int g = 0; g < device_count; ++g) {
    SplitDevice &d = devs[g];
    if (!d.rows) continue;
    MM_HIP_MD(hipSetDevice(d.phys));
    float ms = 0.f;
    MM_HIP_MD(hipEventElapsedTime(&ms, d.start, d.stop));
    if (per_device_seconds) per_device_seconds[g] = 1e-3 * (double)ms;
    elapsed = std::max(elapsed, 1e-3 * (double)ms);
  }
  if (host_wall_seconds) *host_wall_seconds = wall;
  if (elapsed_seconds) *elapsed_seconds = elapsed;
  for (SplitDevice &d : devs) {
    if (!d.rows) continue;
    MM_HIP_MD(hipSetDevice(d.phys));
    MM_HIP_MD(hipMemcpy((char *)c_host + (size_t)d.row0 * m * es, d.c, (size_t)d.rows * m * es, hipMemcpyDeviceToHost));
  }
#undef MM_HIP_MD
  return MM_OK;
}

int mm_set_default_config(const mm_config_t *cfg) {
  if (!valid_cfg(cfg)) return fail(MM_ERR_BAD_ARGUMENT, "invalid mm_config_t");
  std::lock_guard<std::mutex> lock(g_default_cfg_mutex);
  g_default_cfg = *cfg;
  return MM_OK;
}

// Host-pointer entry point.  Rows of C are independent (kernel/Compute.cpp:53-60), so large problems
// are pipelined in row slabs: while slab i is multiplied on a non-blocking stream, slab i+1 of A is
// copied in and finished slabs of C are copied out, hiding most of the PCIe time behind the kernel
// (float 16384^3: 120 ms one-shot -> see profiles/).  The arithmetic of a slab is the arithmetic of
// the same rows in a single launch (tests: row-slab property), so results do not depend on slabbing.
static int mm_run_host_pointers(const mm_config_t &cfg, const void *a, const void *b, void *c, unsigned n, unsigned k,
                                unsigned m) {
  int rc = check_device(0);
  if (rc) return rc;
  rc = check_problem(&cfg, a, b, c, n, k, m);
  if (rc) return rc;
  if (n == 0 || m == 0) return MM_OK;
  const size_t es = mm_dtype_size(cfg.dtype);
  MM_HIP(hipSetDevice(0));
  void *ad = nullptr, *bd = nullptr, *cd = nullptr;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> done;
  auto cleanup = [&]() {
    for (hipEvent_t e : done) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    if (ad) (void)hipFree(ad);
    if (bd) (void)hipFree(bd);
    if (cd) (void)hipFree(cd);
  };
#define MM_HIP_HP(call)                                               \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) { cleanup(); return hip_fail(e_, #call); }  \
  } while (0)
  MM_HIP_HP(hipMalloc(&ad, (size_t)n * k * es));
  MM_HIP_HP(hipMalloc(&bd, (size_t)k * m * es));
  MM_HIP_HP(hipMalloc(&cd, (size_t)n * m * es));
  MM_HIP_HP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  MM_HIP_HP(hipMemcpy(bd, b, (size_t)k * m * es, hipMemcpyHostToDevice));
  // a K x N A is not row-sliceable; small problems are not worth slicing
  bool pipelined = cfg.layout_a == MM_A_ROW_MAJOR && (double)n * k * m >= 64.0 * 1024 * 1024 * 1024 && n >= 2048;
  if (pipelined) {   // a job that one launch would run as stream-K stays one launch: slabs could not reproduce its unit ranges
    const mm::Problem whole{nullptr, nullptr, nullptr, n, k, m, false};
    if (choose(cfg, whole) == FAM_MFMA_F32) {
      const int v = f32_variant_for(whole);
      if (v >= 0 && mm::mfma_f32_splitk(whole, v) == 0) pipelined = false;
    }
  }
  const unsigned slabs = pipelined ? 8 : 1;
  const unsigned slab_rows = ((n + slabs - 1) / slabs + 255) / 256 * 256;
  for (unsigned r0 = 0; r0 < n; r0 += slab_rows) {
    const unsigned rows = std::min(slab_rows, n - r0);
    const char *a_src = (const char *)a + (cfg.layout_a == MM_A_ROW_MAJOR ? (size_t)r0 * k * es : 0);
    char *a_dst = (char *)ad + (cfg.layout_a == MM_A_ROW_MAJOR ? (size_t)r0 * k * es : 0);
    MM_HIP_HP(hipMemcpy(a_dst, a_src, (size_t)(slabs == 1 ? n : rows) * k * es, hipMemcpyHostToDevice));
    mm::Problem p{a_dst, bd, (char *)cd + (size_t)r0 * m * es, slabs == 1 ? n : rows, k, m,
                  cfg.layout_a == MM_A_TRANSPOSED, slabs == 1 ? 0u : n};   // a row slab of the n-row job (or all of it)
    rc = dispatch(stream, cfg, p);
    if (rc) { cleanup(); return rc; }
    hipEvent_t e;
    MM_HIP_HP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    done.push_back(e);
    MM_HIP_HP(hipEventRecord(e, stream));
    if (slabs == 1) break;
  }
  size_t idx = 0;
  for (unsigned r0 = 0; r0 < n; r0 += slab_rows, ++idx) {
    const unsigned rows = slabs == 1 ? n : std::min(slab_rows, n - r0);
    MM_HIP_HP(hipEventSynchronize(done[idx]));
    MM_HIP_HP(hipMemcpy((char *)c + (size_t)r0 * m * es, (char *)cd + (size_t)r0 * m * es, (size_t)rows * m * es,
                        hipMemcpyDeviceToHost));
    if (slabs == 1) break;
  }
#undef MM_HIP_HP
  cleanup();
  return MM_OK;
}

int mm_gemm_host(const mm_config_t *cfg, const void *a, const void *b, void *c, unsigned n, unsigned k, unsigned m) {
  if (!valid_cfg(cfg)) return fail(MM_ERR_BAD_ARGUMENT, "invalid mm_config_t");
  return mm_run_host_pointers(*cfg, a, b, c, n, k, m);
}

void MatrixMultiplicationKernel(const void *a, const void *b, void *c, unsigned n, unsigned k, unsigned m) {
  const int rc = mm_run_host_pointers(default_cfg(), a, b, c, n, k, m);
  if (rc) {
    fprintf(stderr, "MatrixMultiplicationKernel failed: %s\n", mm_last_error());
    abort();  // the reference's symbol returns void; failing silently would fake a result
  }
}

int mm_config_supported(const mm_config_t *cfg) {
  if (!valid_cfg(cfg)) return 0;
  if (cfg->path == MM_PATH_SPLIT)
    return cfg->dtype == MM_DTYPE_F32 && cfg->map_op == MM_OP_MULTIPLY && cfg->reduce_op == MM_OP_ADD;
  return 1;
}

const char *mm_kernel_name(const mm_config_t *cfg, unsigned n, unsigned k, unsigned m) {
  if (!valid_cfg(cfg)) return "invalid";
  mm::Problem p{nullptr, nullptr, nullptr, n, k, m, cfg->layout_a == MM_A_TRANSPOSED};
  switch (choose(*cfg, p)) {
    // every family answers through the same resolver its launcher uses, so the name IS the kernel that runs
    case FAM_MFMA_F32: {
      const int v = f32_variant_for(p);
      if (v < 0) return "unsupported";
      static const char *const split_names[] = {nullptr, nullptr, "mfma_f32_128x128x32_w4x2_splitk2", "mfma_f32_128x128x32_w4x2_splitk3",
                                                "mfma_f32_128x128x32_w4x2_splitk4", "mfma_f32_128x128x32_w4x2_splitk5",
                                                "mfma_f32_128x128x32_w4x2_splitk6", "mfma_f32_128x128x32_w4x2_splitk7",
                                                "mfma_f32_128x128x32_w4x2_splitk8"};
      const int splits = mm::mfma_f32_splitk(p, v);
      if (v == 64 && splits > 1) {
        static const char *const small_split_names[] = {nullptr, nullptr, "mfma_f32_64x64x32_w4x2_splitk2", "mfma_f32_64x64x32_w4x2_splitk3",
                                                        "mfma_f32_64x64x32_w4x2_splitk4", "mfma_f32_64x64x32_w4x2_splitk5",
                                                        "mfma_f32_64x64x32_w4x2_splitk6", "mfma_f32_64x64x32_w4x2_splitk7",
                                                        "mfma_f32_64x64x32_w4x2_splitk8"};
        return small_split_names[splits];
      }
      if (splits == 0) return "mfma_f32_128x128x32_w4x2_streamk";            // teams, the last part to arrive gathers: what MM_PATH_AUTO runs
      if (splits == 9) return "mfma_f32_128x128x32_w4x2_streamk_fixup";      // single ranges + fix-up kernel (cross-check, its own bits)
      if (splits == 11) return "mfma_f32_128x128x32_w4x2_streamk_two_kernels";   // teams + fix-up kernel (cross-check); the bits of `streamk`
      if (splits == 12) return "mfma_f32_128x128x32_w4x2_streamk_ticket";        // teams, one counter ticket per part, the last ticket gathers; the bits of `streamk`
      return splits > 1 ? split_names[splits] : mm::mfma_f32_name(v);
    }
    case FAM_MFMA_F64: return mm::mfma_f64_name(p);
    case FAM_MFMA_F16: return mm::mfma_f16_name(p);
    case FAM_MFMA_I8: return mm::mfma_i8_name(p);
    case FAM_HALF_WIDE: return "ordered_wide_f16";
    case FAM_F32_SPLIT: return "mfma_f32_split_bf16x3";
    case FAM_NONE: return "unsupported";
    case FAM_VALU_TILE: return "valu_tile";
    case FAM_ORDERED_TILE: return "ordered_tile";   // k ascending, one accumulator, unfused: Naive's bits on 128 x 128 register tiles
    default: return "ordered";
  }
}

int mm_kernel_info(const mm_config_t *cfg, unsigned n, unsigned k, unsigned m, mm_kernel_info_t *info) {
  if (!valid_cfg(cfg) || !info) return fail(MM_ERR_BAD_ARGUMENT, "invalid arguments to mm_kernel_info");
  mm::Problem p{nullptr, nullptr, nullptr, n, k, m, cfg->layout_a == MM_A_TRANSPOSED};
  kernel_info_for(cfg, p, info);
  return MM_OK;
}

}  // extern "C"

static void kernel_info_for(const mm_config_t *cfg, const mm::Problem &p, mm_kernel_info_t *info) {
  mm_kernel_info_t r = {};
  // the CURRENT device's compute units once the library is initialised (not device 0's); the MI355X's 256 before
  int cur = 0;
  if (g_init_status != MM_OK || hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = 0; }
  r.compute_units = (unsigned)mm::device_compute_units(cur);
  r.max_clock_mhz = 2400.0;
  switch (choose(*cfg, p)) {
    case FAM_MFMA_F32:
      mm::mfma_f32_geometry(f32_variant_for(p), &r.tile_n, &r.tile_m, &r.tile_k, &r.wavefronts);
      r.inst_n = 32; r.inst_m = 32; r.inst_k = 2; r.ops_per_clk_per_cu = 256.0;   // 64 FLOP/clk/SIMD
      r.measured_issue_efficiency = 0.967;  // 152.2 TF of 157.3 (profiles/r02z_f32_scalar_base_dma.log)
      break;
    case FAM_MFMA_F64:
      switch (mm::mfma_f64_tile(p)) {
        case 4: r.tile_n = 64; r.tile_m = 64; r.wavefronts = 4; break;
        case 1: r.tile_n = 128; r.tile_m = 128; r.wavefronts = 4; break;
        default: r.tile_n = 256; r.tile_m = 128; r.wavefronts = 8; break;
      }
      r.tile_k = 16;
      r.inst_n = 16; r.inst_m = 16; r.inst_k = 4; r.ops_per_clk_per_cu = 128.0;
      r.measured_issue_efficiency = 0.97;  // pinned schedule + scalar-base DMA: 76.4 TF of 78.6 (profiles/r02z_f64_scalar_base_dma.log)
      break;
    case FAM_MFMA_F16: {
      const std::string name = mm::mfma_f16_name(p);   // geometry read off the resolved kernel's name
      const bool s16 = name.find("16x16x32") != std::string::npos, pp = name.find("pingpong") != std::string::npos;
      r.tile_n = name.find("_64x256") != std::string::npos ? 64 : name.find("128x256") != std::string::npos ? 128 : 256;
      r.tile_m = 256; r.wavefronts = r.tile_n == 256 ? 8 : 4;
      r.tile_k = pp ? 32 : 64;
      r.inst_n = r.inst_m = s16 ? 16 : 32; r.inst_k = s16 ? 32 : 16; r.ops_per_clk_per_cu = 4096.0;
      // ping-pong schedule: MfmaUtil 91.4 % (16x16x32, profiles/r03g_pmc_f16_32768_16x16x32.json) / 89.7 % (32x32x16) at
      // 32768^3; the chip is power-limited there and delivers ~1.5 GHz, so 0.91 x 2.4 GHz over-predicts wall throughput
      r.measured_issue_efficiency = pp ? (s16 ? 0.91 : 0.90) : 0.67;
      break;
    }
    case FAM_MFMA_I8: {
      const std::string name = mm::mfma_i8_name(p);
      const bool s16 = name.find("16x16x64") != std::string::npos, pp = name.find("pingpong") != std::string::npos;
      r.tile_n = name.find("_64x256") != std::string::npos ? 64 : 256; r.tile_m = 256; r.wavefronts = r.tile_n == 256 ? 8 : 4;
      r.tile_k = pp ? 64 : 128;
      r.inst_n = r.inst_m = s16 ? 16 : 32; r.inst_k = s16 ? 64 : 32; r.ops_per_clk_per_cu = 8192.0;
      r.measured_issue_efficiency = pp ? 0.92 : 0.68;  // profiles/r02i_pmc_i8_32768.json / r01_pmc_i8.json
      break;
    }
    case FAM_F32_SPLIT:
      {  // the 128 x 128 geometry runs only in the launcher's default branch (no schedule / product / flush bits set)
        const int sv = mm::tuning(mm::TUNE_SPLIT_VARIANT);
        const bool plain = sv <= 0 || (sv & (1 | 2 | 4 | 64 | 128)) == 0;
        r.tile_n = r.tile_m = plain ? (unsigned)mm::mfma_f32_split_tile(p, sv) : 256u;
      }
      r.tile_k = 16; r.wavefronts = r.tile_n == 256 ? 8 : 4;
      r.inst_n = 32; r.inst_m = 32; r.inst_k = 16;
      r.ops_per_clk_per_cu = 4096.0 / 6.0;  // six bf16 MFMAs per fp32 multiply-add block
      r.measured_issue_efficiency = 0.90;
      break;
    case FAM_ORDERED_TILE: {
      r.tile_n = 128; r.tile_m = 128; r.tile_k = (unsigned)(64 / mm_dtype_size(cfg->dtype)); r.wavefronts = mm_dtype_size(cfg->dtype) == 8 ? 8 : 4;
      r.inst_n = 1; r.inst_m = 64; r.inst_k = 1;
      // one instruction per operation (multiply and add are NOT fused here).  4-byte types: 2-cycle issue, 128 operations per
      // clock per CU; half: v_pk_mul_f16 / v_pk_add_f16 take two elements per lane but issue every 4 cycles (measured 4.30:
      // profiles/r06a_probe_valu_issue_rates_unfused_pairs.txt) -- the same 128; 8-byte types: 4-cycle issue, 64
      r.ops_per_clk_per_cu = mm_dtype_size(cfg->dtype) == 8 ? 64.0 : 128.0;
      r.measured_issue_efficiency = cfg->dtype == MM_DTYPE_F16 ? 0.94 : 0.74;   // half 74.4 / float 58.4 / double 28.9 TOp/s at 8192^3 (r06a)
      break;
    }
    case FAM_VALU_TILE: {
      r.tile_n = 128; r.tile_m = 128; r.tile_k = 16; r.wavefronts = 4;
      r.inst_n = 1; r.inst_m = 64; r.inst_k = 1;
      // 128 lane-instructions per clock per CU; a map+reduce step (2 ops) costs 2 instructions,
      // 1 when it fuses (float multiply-add), 1.5 for float min/max reductions (v_min3/v_max3)
      const bool fp = cfg->dtype == MM_DTYPE_F32 || cfg->dtype == MM_DTYPE_F16;
      const bool fma = cfg->map_op == MM_OP_MULTIPLY && cfg->reduce_op == MM_OP_ADD && (fp || cfg->dtype == MM_DTYPE_F64);
      const bool m3 = fp && (cfg->reduce_op == MM_OP_MIN || cfg->reduce_op == MM_OP_MAX);
      r.ops_per_clk_per_cu = fma ? 256.0 : (m3 ? 128.0 * 4.0 / 3.0 : 128.0);
      r.measured_issue_efficiency = 0.65;   // against 2-cycle issue; min/max instructions take 4.2 (profiles/r02_probe_valu_*)
      break;
    }
    default:
      r.tile_n = 64; r.tile_m = 64; r.tile_k = 16; r.wavefronts = 4;
      r.inst_n = 1; r.inst_m = 64; r.inst_k = 1; r.ops_per_clk_per_cu = 128.0;
      break;
  }
  *info = r;
}
