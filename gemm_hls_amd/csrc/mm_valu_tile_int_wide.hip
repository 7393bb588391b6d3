// valu_tile instantiations: 32- and 64-bit integers; and the family's dispatcher.
#include "mm_valu_tile.inc"
namespace mm {
int launch_valu_tile_fp(hipStream_t s, const mm_config_t &cfg, const Problem &p);
int launch_valu_tile_int_narrow(hipStream_t s, const mm_config_t &cfg, const Problem &p);
int launch_valu_tile_fp_exact(hipStream_t s, const mm_config_t &cfg, const Problem &p);   // mm_valu_tile_fp_exact.hip
bool valu_tile_serves(const mm_config_t &cfg, const Problem &p) {
  if (p.k % 4 != 0 || p.m % 4 != 0 || (p.a_transposed && p.n % 4 != 0)) return false;
  const bool map_ok = cfg.map_op == MM_OP_MULTIPLY || cfg.map_op == MM_OP_ADD || cfg.map_op == MM_OP_MIN || cfg.map_op == MM_OP_MAX;
  const bool red_ok = cfg.reduce_op == MM_OP_ADD || cfg.reduce_op == MM_OP_MIN || cfg.reduce_op == MM_OP_MAX;
  return map_ok && red_ok;
}

int launch_valu_tile_exact(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.dtype) {
    case MM_DTYPE_F32: case MM_DTYPE_F64: case MM_DTYPE_F16: return launch_valu_tile_fp_exact(s, cfg, p);
    default: return launch_valu_tile(s, cfg, p);   // integer algebras: k ascending with one accumulator IS Naive
  }
}

int launch_valu_tile(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.dtype) {
    case MM_DTYPE_F32: case MM_DTYPE_F64: case MM_DTYPE_F16: return launch_valu_tile_fp(s, cfg, p);
    case MM_DTYPE_I8: case MM_DTYPE_U8: case MM_DTYPE_I16: case MM_DTYPE_U16:
      return launch_valu_tile_int_narrow(s, cfg, p);
    case MM_DTYPE_I32: return vt_type<int32_t>(s, cfg, p);
    case MM_DTYPE_U32: return vt_type<uint32_t>(s, cfg, p);
    case MM_DTYPE_I64: return vt_type<int64_t>(s, cfg, p);
    case MM_DTYPE_U64: return vt_type<uint64_t>(s, cfg, p);
  }
  return kErrNotSupported;
}
}  // namespace mm
