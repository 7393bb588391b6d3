// valu_tile instantiations for the k-ORDERED contract on the floating-point element types ("ordered_tile"): this unit is
// compiled with -ffp-contract=off (gemm_hls_amd/build.py, CMakeLists.txt) and with MM_VT_EXACT, so that every output is
//     acc = Reduce::identity(); for k ascending: acc = Reduce(acc, Map(a, b))      (include/Utility.h:18-42)
// with Map and Reduce two separately rounded operations in Data_t -- binary16 accumulating in binary16, which is what the
// reference's ProcessingElement does (kernel/Compute.cpp:129-133) and what its hosts compare half results with EXACTLY
// (test/TestSimulation.cpp:80-85, host/RunHardware.cpp:214-218).  Serves MM_PATH_ORDERED (RunHardware "hw_emu") and, under
// the half_contract = reference knob, half (Multiply, Add) launches of MM_PATH_AUTO.
#define MM_VT_EXACT 1
#pragma clang fp contract(off)   // belt and braces: the flag is also on this unit's compile line
#include "mm_valu_tile.inc"
namespace mm {
int launch_valu_tile_fp_exact(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.dtype) {
    case MM_DTYPE_F32: return vt_type<float>(s, cfg, p);
    case MM_DTYPE_F64: return vt_type<double>(s, cfg, p);
    case MM_DTYPE_F16: return vt_type<half_t>(s, cfg, p);
    default: return kErrNotSupported;
  }
}
}  // namespace mm
