/*
 * mm_gemm.h -- C ABI of the MI355X (gfx950) implementation of gemm_hls's hot path
 *              C = A (map, reduce) B,  A: N x K, B: K x M, C: N x M, all row-major.
 *
 * Plain C, plain pointers and sizes.  One shared library (libmm_gemm_amd.so) exports all
 * of it.  Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  Unless stated otherwise a function returns 0 on success and a
 * non-zero status otherwise, with a human-readable message available from mm_last_error();
 * nothing throws across this boundary.  There is NO CPU fallback anywhere behind this ABI:
 * without a usable gfx950 device every compute call fails with MM_ERR_NO_DEVICE.
 */
#ifndef MM_GEMM_H
#define MM_GEMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element type == the reference's build-time MM_DATA_TYPE / Data_t
 * (CMakeLists.txt:17, include/Config.h.in:15; "half" and "uint8_t" special-cased at
 * CMakeLists.txt:40-50). */
typedef enum {
  MM_DTYPE_F32 = 0, /* float          */
  MM_DTYPE_F64 = 1, /* double         */
  MM_DTYPE_F16 = 2, /* half (IEEE binary16).  (Multiply, Add) under MM_PATH_AUTO: exact products,
                       f32 accumulation, ONE rounding on store, for EVERY shape (matrix cores when
                       K % 16 == 0 and M % 8 == 0, a plain wide-accumulate kernel otherwise);
                       MM_PATH_ORDERED accumulates in binary16 exactly like the reference
                       (kernel/Compute.cpp:129-133), and so does MM_PATH_AUTO once the process asked for
                       the REFERENCE's half contract: mm_tuning_set("half_contract", 1) /
                       MM_HALF_CONTRACT=reference -- the result the reference's hosts compare with
                       EXACTLY (test/TestSimulation.cpp:80-85, host/RunHardware.cpp:214-218) */
  MM_DTYPE_I8 = 3,  /* int8_t         */
  MM_DTYPE_U8 = 4,  /* uint8_t        */
  MM_DTYPE_I16 = 5, /* short          */
  MM_DTYPE_U16 = 6, /* unsigned short */
  MM_DTYPE_I32 = 7, /* int            */
  MM_DTYPE_U32 = 8, /* unsigned       */
  MM_DTYPE_I64 = 9, /* long           */
  MM_DTYPE_U64 = 10 /* unsigned long  */
} mm_dtype_t;

/* == the reference's MM_MAP_OP / MM_REDUCE_OP, i.e. hlslib::op::<Name><Data_t>
 * (CMakeLists.txt:33-34, include/Config.h.in:34-35; applied at kernel/Compute.cpp:129,133 and
 * include/Utility.h:29,37).  identity(): Add 0, Multiply 1, And 1, Min max(), Max lowest().
 * hlslib's Operators.h is an absent submodule of the reference tree, so these seeds are NOT pinned by a
 * reference file: Add / Multiply / And / Min are what any identity of those reductions must be; Max lowest()
 * is THIS LIBRARY'S CHOICE (the mathematical identity).  The published hlslib is believed to seed Max with
 * numeric_limits<T>::min() -- for floating types the smallest POSITIVE value -- which differs from lowest()
 * only when every mapped value of an output is negative; on the reference's own inputs ([1,10]) both give
 * the same bits, which is all that parity is pinned on. */
typedef enum {
  MM_OP_ADD = 0,
  MM_OP_MULTIPLY = 1,
  MM_OP_AND = 2,
  MM_OP_MIN = 3,
  MM_OP_MAX = 4
} mm_op_t;

/* Which kernel family serves a launch.
 *   MM_PATH_AUTO     the fast path for the configuration (MFMA for (Multiply,Add) on
 *                    float/half/double, register-tiled VALU otherwise); == RunHardware "hw".
 *   MM_PATH_ORDERED  the plain GPU kernel that evaluates every output as the reference's
 *                    Naive does -- acc = identity; for k ascending: acc = Reduce(acc, Map(a,b)),
 *                    multiply and add NOT fused -- bit-identical to include/Utility.h:18-42
 *                    for every dtype; == RunHardware "hw_emu" (a slower, independently written
 *                    execution of the same contract on the same device).  Two kernels with the same
 *                    bits serve it: "ordered_tile", 128 x 128 register tiles (K % 4 == 0, M % 4 == 0,
 *                    16-byte aligned operands; half 74, float 58, double 29 TOp/s at 8192^3), and
 *                    "ordered", fully predicated 64 x 64 tiles, for everything else ("ordered_variant"
 *                    = 0 forces it: the cross-check).
 *   MM_PATH_SPLIT    float (Multiply, Add) only, opt-in: fp32 operands split into three bf16
 *                    planes (all 24 significand bits), six bf16 matrix-core products per
 *                    element pair accumulated in fp32 -- each product good to ~2^-25, i.e. at
 *                    least as fine as an fp32 multiply; 1.7x the fp32 matrix-core peak.  Any
 *                    N, K, M and any element-aligned pointers; needs 6 bytes of stream-ordered
 *                    workspace per element of A and B (hipMallocAsync on the launch stream).
 *                    Inputs must be finite and below 2^127 in magnitude for the error bound
 *                    to hold (an inf/nan operand still gives non-finite results, but +inf may
 *                    become nan).  Any other configuration: MM_ERR_UNSUPPORTED.  RunHardware:
 *                    MM_PATH=split with "hw".  See mm_release_workspace(). */
typedef enum { MM_PATH_AUTO = 0, MM_PATH_ORDERED = 1, MM_PATH_SPLIT = 2 } mm_path_t;

/* Layout of A: row-major N x K (default) or K x N == the reference's MM_TRANSPOSED_A
 * (CMakeLists.txt:30, include/Utility.h:31-35, kernel/Memory.cpp:205-261). */
typedef enum { MM_A_ROW_MAJOR = 0, MM_A_TRANSPOSED = 1 } mm_layout_a_t;

enum {
  MM_OK = 0,
  MM_ERR_NO_DEVICE = 1,   /* no gfx950 device / HIP runtime unusable            */
  MM_ERR_BAD_ARGUMENT = 2,/* null pointer, unknown enum, device index out of range */
  MM_ERR_UNSUPPORTED = 3, /* (dtype, map, reduce) not compiled into this library */
  MM_ERR_HIP = 4          /* a HIP call failed; see mm_last_error()             */
};

typedef struct {
  mm_dtype_t dtype;
  mm_op_t map_op;
  mm_op_t reduce_op;
  mm_path_t path;
  mm_layout_a_t layout_a;
} mm_config_t;

/* ---- device management: replaces hlslib::ocl::Context / MakeBuffer / CopyFromHost /
 *      CopyToHost as used by host/RunHardware.cpp:116-145,187-190 ----------------------------- */

/* hlslib::ocl::Context context; (RunHardware.cpp:116-117).  Reports the number of usable
 * gfx950 devices. */
int mm_init(int *device_count);

/* context.MakeBuffer<...>(DDR bank, elements) (RunHardware.cpp:122-138).  The caller owns the
 * returned device pointer and releases it with mm_free. */
int mm_alloc(int device, size_t bytes, void **device_ptr);
int mm_free(int device, void *device_ptr);

/* buffer.CopyFromHost / CopyToHost (RunHardware.cpp:142-144,189).  Blocking. */
/* Stream-ordered workspace (MM_PATH_SPLIT: 6 bytes per element of A and B of the largest launch so far; the fp32
 * default path: the partial planes / scratch slots of its split-K and stream-K launches, a few MiB, and for a K x N A
 * that does not fill rounds of the K x N kernel's tiles its N x K copy, 4 bytes per element, at most 2 GiB) is cached
 * between launches in a memory pool this library owns -- never in the process's default pool.  This waits for the device and
 * hands that memory back to the driver; the next launch that needs workspace allocates again. */
int mm_release_workspace(int device);

/* The PCI address ("0000:c1:00.0") of HIP device `device`, for tools that must find the same physical GPU through
 * an interface that does not honour HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (ROCm SMI, sysfs): the runner's
 * power meter (the reference's PowerMeter role, host/RunHardware.cpp:156-172). */
int mm_device_pci_bus_id(int device, char *buffer, int length);

int mm_copy_to_device(int device, void *device_dst, const void *host_src, size_t bytes);
int mm_copy_to_host(int device, void *host_dst, const void *device_src, size_t bytes);

/* Fill a device buffer with the reference's input distribution (uniform in [1,10], values
 * drawn on the device, NOT the seed-5 host stream): for `verify off` timing runs, where the
 * reference leaves its device buffers uninitialised (RunHardware.cpp:99,140). */
int mm_fill_device(int device, mm_dtype_t dtype, void *device_ptr, size_t elements,
                   unsigned long long seed);

/* ---- the kernel: replaces program.MakeKernel("MatrixMultiplicationKernel", a, b, c, N, K, M)
 *      + kernel.ExecuteTask() (RunHardware.cpp:148-162).  C is pure output (kernel/Top.cpp
 *      never reads it).  Sizes follow the reference's contract (RunHardware.cpp:50-61): K and
 *      M multiples of the 64-byte bus in elements; other sizes are served by a slower,
 *      fully predicated kernel rather than rejected. ----------------------------------------- */

/* Blocking launch on `device`; *elapsed_seconds (may be NULL) receives the kernel time
 * measured with HIP events on the launch stream == ExecuteTask()'s elapsed.first.
 * MM_PATH_AUTO needs a, b and c 16-byte aligned (any allocator's result; the reference's host
 * vectors are 4096-byte aligned, include/Utility.h:48) and returns MM_ERR_BAD_ARGUMENT for an
 * offset view; MM_PATH_ORDERED takes any element-aligned pointer.
 * Thread safety: every entry point may be called concurrently from several host threads (on the
 * same or on different devices / streams); mm_last_error() is per thread. */
int mm_gemm_launch(int device, const mm_config_t *cfg, const void *a_dev, const void *b_dev,
                   void *c_dev, unsigned size_n, unsigned size_k, unsigned size_m,
                   double *elapsed_seconds);

/* Asynchronous launch on a caller-provided hipStream_t (passed as void*; NULL = the default
 * stream) of the CURRENT device.  No synchronisation, no timing: for callers that own streams,
 * events and graphs (bench.py, multi-stream pipelines). */
int mm_gemm_enqueue(void *hip_stream, const mm_config_t *cfg, const void *a_dev,
                    const void *b_dev, void *c_dev, unsigned size_n, unsigned size_k,
                    unsigned size_m);

/* One node, `device_count` GPUs, rows of C split into contiguous slabs (device g gets the rows mm_row_slab() names), B
 * replicated, no collective: every outer tile of C is independent (kernel/Compute.cpp:53-60, kernel/Memory.cpp:114-127,
 * 272-286, 367-391).  Host pointers in, host pointer out; copies are outside the timed region exactly as in
 * RunHardware.cpp:140-190.  *elapsed_seconds = the longest of the devices' kernel times, each measured with HIP events on that
 * device's stream (mm_gemm_multi_device_timed also hands out the per-device figures and the host clock).  New functionality
 * (the reference is single-device).  A K x N A (MM_A_TRANSPOSED == MM_TRANSPOSED_A, kernel/Memory.cpp:205-261)
 * is split along its columns (one strided copy per device).  With device_count = 1 (or N within one slab) the launch is
 * mm_gemm_launch's.  The kernel FAMILY and every decision that changes a row's summation order (split-K of small fp32
 * problems) are taken on the whole job, so a split gives the one-device bits -- except for the mid-size fp32 shapes a
 * single device runs as stream-K (one to two rounds of 128 x 128 tiles), which slabs run as whole tiles: same contract,
 * different summation order.
 * Tests only: "md_virtual_devices" = V (MM_MD_VIRTUAL_DEVICES) lets device_count go up to V LOGICAL devices dealt out over
 * the physical ones round-robin, so that every per-device branch runs on a 1-GPU box. */
int mm_gemm_multi_device(int device_count, const mm_config_t *cfg, const void *a_host,
                         const void *b_host, void *c_host, unsigned size_n, unsigned size_k,
                         unsigned size_m, double *elapsed_seconds);
/* The same call, saying where the time went ("max over devices of kernel time").  Every
 * device's launch is bracketed by HIP events on its own stream: *elapsed_seconds = the MAX over devices of that kernel time
 * (== what mm_gemm_multi_device reports), per_device_seconds[g] (device_count entries, may be NULL) = device g's own kernel
 * time, 0 for a trailing device that got no rows, *host_wall_seconds (may be NULL) = the host clock from the first dispatch
 * to the last completion, a cross-check that also contains the G launch latencies.  `MM_GPUS=G RunHardware.exe` prints them. */
int mm_gemm_multi_device_timed(int device_count, const mm_config_t *cfg, const void *a_host, const void *b_host,
                               void *c_host, unsigned size_n, unsigned size_k, unsigned size_m, double *elapsed_seconds,
                               double *per_device_seconds, double *host_wall_seconds);
/* The row partition mm_gemm_multi_device uses, for callers that drive one process per GPU themselves (bench.py):
 * slabs of ceil(N / G) rows rounded up to whole tile rows of the kernel that will run on them (mm_kernel_info's tile_n),
 * so only the last busy device owns a ragged tile row; trailing devices may get *rows = 0.  Pure arithmetic: works
 * without a device. */
int mm_row_slab(const mm_config_t *cfg, unsigned size_n, unsigned size_k, unsigned size_m, int device_count, int rank,
                unsigned *row0, unsigned *rows);

/* extern "C" void MatrixMultiplicationKernel(MemoryPackK_t const a[], MemoryPackM_t const b[],
 * MemoryPackM_t c[], unsigned size_n, unsigned size_k, unsigned size_m)
 * (include/MatrixMultiplication.h:155-171, kernel/Top.cpp:6-18; called directly by
 * test/TestSimulation.cpp:66-67).  Host pointers; DataPack arrays are layout-compatible with
 * plain row-major Data_t arrays (include/Utility.h:44-63).  Uses device 0 and the
 * configuration set by mm_set_default_config (initially float, Multiply, Add == the
 * reference's CMake defaults).  Errors are reported on stderr and abort(), because the
 * reference's symbol returns void. */
#ifndef MM_GEMM_NO_KERNEL_SYMBOL /* defined by translation units that declare the reference's typed
                                    or 3-pointer form themselves (host/KernelShim.cpp and its callers) */
void MatrixMultiplicationKernel(const void *a, const void *b, void *c, unsigned size_n,
                                unsigned size_k, unsigned size_m);
#endif
int mm_set_default_config(const mm_config_t *cfg);
/* The same host-pointer call with an explicit configuration and a status instead of abort().
 * This is what the build-time-configured kernel shim (gemm_hls_amd/host/KernelShim.cpp: one
 * library per MM_DATA_TYPE / MM_MAP_OP / MM_REDUCE_OP / MM_TRANSPOSED_A / MM_DYNAMIC_SIZES choice,
 * exporting the reference's exact symbol incl. the 3-pointer static-size form,
 * include/MatrixMultiplication.h:155-171) forwards to. */
int mm_gemm_host(const mm_config_t *cfg, const void *a, const void *b, void *c, unsigned size_n,
                 unsigned size_k, unsigned size_m);

/* ---- introspection ------------------------------------------------------------------------ */
size_t mm_dtype_size(mm_dtype_t dtype);
/* 1 if (dtype, map, reduce) is compiled in, else 0. */
int mm_config_supported(const mm_config_t *cfg);
/* Name of the kernel family that mm_gemm_launch would run for this problem (static string),
 * e.g. "mfma_f32_128x256x32", "valu_tile", "ordered". */
const char *mm_kernel_name(const mm_config_t *cfg, unsigned size_n, unsigned size_k, unsigned size_m);
/* Geometry of the kernel family that would serve this problem: the GPU counterpart of the
 * constants src/PrintSpecifications.cpp prints for the FPGA build (memory tile = the output tile a
 * workgroup keeps resident, compute tile = what one wavefront instruction computes). */
typedef struct {
  unsigned tile_n, tile_m, tile_k;     /* resident C tile (rows x cols) and k-slab per workgroup  */
  unsigned wavefronts;                 /* wavefronts per workgroup                               */
  unsigned inst_n, inst_m, inst_k;     /* shape of one matrix/vector instruction (1x64x1 = VALU) */
  double ops_per_clk_per_cu;           /* peak map+reduce operations per clock per compute unit  */
  unsigned compute_units;              /* 256 on MI355X                                          */
  double max_clock_mhz;                /* 2400 on MI355X                                         */
  double measured_issue_efficiency;    /* fraction of issue slots the family sustains at BASELINE
                                          size (rocprofv3 MfmaUtil, profiles/), 0 if not measured */
} mm_kernel_info_t;
int mm_kernel_info(const mm_config_t *cfg, unsigned size_n, unsigned size_k, unsigned size_m,
                   mm_kernel_info_t *info);
/* Tuning knobs for sweeps ("f32_variant", "f64_variant", "f16_variant", "i8_variant", "valu_variant", "split_variant",
 * "band_rows", "f32_splitk", "ordered_variant"; -1 = the library's own choice) and ONE contract knob, "half_contract"
 * (MM_HALF_CONTRACT = reference | wide; 1 = reference: half (Multiply, Add) under MM_PATH_AUTO is evaluated like the reference's
 * kernel -- binary16 products, binary16 accumulation, k ascending -- on the k-ordered tile kernel; unset / 0: f32 accumulation on
 * the matrix cores).  Kernel ids of this library: f32_variant 33 / 8 / 35 / 64 (the
 * geometries of the shape-adaptive pick) and 0 / 3 (cross-checks); f16_variant 200 / 100 / 11 / 0 / 4 / 5; i8_variant 200 /
 * 100 / 10 / 5 / 0; f64_variant 0-4 (4: the 64 x 64 geometry for small problems); f32_splitk 1 whole tiles, 2-8 K chunks,
 * 0 stream-K as MM_PATH_AUTO runs it: teams of workgroups, every part of a tile that a range boundary cuts goes to a scratch
 * slot and raises a flag, and the LAST part to arrive adds the slots in ascending k into C -- one kernel in which no workgroup
 * ever waits for another one, so it is sound next to anything else on the device (other streams, other processes), on
 * CU-masked streams, on partitions and in graphs; 11 the same with a small fix-up kernel doing the gather (cross-check, the
 * bits of 0); 12 the same again with the hand-over written as the canonical last-block pattern -- a release fence, ONE agent-scope acq_rel
 * read-modify-write per part on a per-tile counter, the last ticket gathers: correct by the language's memory model alone, the bits of 0,
 * 0.2-30 % slower (its release fence writes the L2 back; profiles/r06d_*), kept as the cross-check of the shipped flag protocol; 9 stream-K in
 * single ranges with its own fix-up kernel (cross-check, its own bits).
 * "debug_poison" = 1 fills the scratch that kernels hand partial tiles through, and C itself (pure output), with NaN before
 * every stream-K launch: a read of anything the launch did not write, or a tile nobody finished, then shows in C (tests only).  "md_virtual_devices": see mm_gemm_multi_device.  Any
 * other id is refused: the retired schedules and the work-skipping ablations of the measurement history exist only in the
 * lab build (tools/lab/libmm_gemm_amd_lab.so, built on request, where "ablations" = 1 unlocks the latter).  Each knob is
 * initialised ONCE from its environment variable (MM_F32_VARIANT, ...) and changed only through this call afterwards; the
 * launch path never reads the environment.  The reference's counterpart is the build-time tile knob
 * (CMakeLists.txt:18-20), which the host binaries honour through -DMM_MEMORY_TILE_SIZE_N / _M (host/HostConfig.h:
 * ApplyBuildTimeTile). */
int mm_tuning_set(const char *name, int value);
int mm_tuning_get(const char *name, int *value);
/* Message of the last failing call on this thread (static or thread-local storage). */
const char *mm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MM_GEMM_H */
