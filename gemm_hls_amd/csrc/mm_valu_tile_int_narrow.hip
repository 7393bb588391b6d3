// valu_tile instantiations: 8- and 16-bit integers.
#include "mm_valu_tile.inc"
namespace mm {
int launch_valu_tile_int_narrow(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.dtype) {
    case MM_DTYPE_I8: return vt_type<int8_t>(s, cfg, p);
    case MM_DTYPE_U8: return vt_type<uint8_t>(s, cfg, p);
    case MM_DTYPE_I16: return vt_type<int16_t>(s, cfg, p);
    case MM_DTYPE_U16: return vt_type<uint16_t>(s, cfg, p);
    default: return kErrNotSupported;
  }
}
}  // namespace mm
