// valu_tile instantiations: floating-point element types.
#include "mm_valu_tile.inc"
namespace mm {
int launch_valu_tile_fp(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.dtype) {
    case MM_DTYPE_F32: return vt_type<float>(s, cfg, p);
    case MM_DTYPE_F64: return vt_type<double>(s, cfg, p);
    case MM_DTYPE_F16: return vt_type<half_t>(s, cfg, p);
    default: return kErrNotSupported;
  }
}
}  // namespace mm
