// A gemm_hls-style build configuration written for this test (NOT the reference's generated file): the names the
// hlslib::ocl adapter reads -- Data_t, OperatorMap, OperatorReduce -- for an int (Multiply, Add) build with dynamic sizes.
#pragma once
#include "hlslib/xilinx/Operators.h"
using Data_t = int;
using OperatorMap = hlslib::op::Multiply<Data_t>;
using OperatorReduce = hlslib::op::Add<Data_t>;
