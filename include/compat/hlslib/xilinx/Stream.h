/*
 * hlslib/xilinx/Stream.h (include/compat) -- the NAME hlslib::Stream<T>, which include/MatrixMultiplication.h:10-12 pulls
 * into the global namespace and include/Memory.h / include/Compute.h use in the declarations of the FPGA dataflow
 * functions (ReadA ... WriteC, ProcessingElement; arrays of streams appear as parameters, so the type is complete).
 * Host code never constructs one: on the MI355X those functions are the HIP kernels behind libmm_gemm_amd.so, not FIFOs
 * between processes -- so there is no Push / Pop here, and trying to build the FPGA kernel sources against this header
 * fails at compile time, which is the intent.  The standard headers below are the ones the original drags in and the
 * reference's sources rely on transitively (src/PrintSpecifications.cpp uses std::cout without including <iostream>).
 */
#pragma once
#include <cstddef>
#include <iostream>
#include <string>

namespace hlslib {
template <typename T, unsigned depth = 0>
class Stream {
 public:
  Stream() = delete;
};
}  // namespace hlslib
