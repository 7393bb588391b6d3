/* hlslib/xilinx/Resource.h (include/compat) -- FPGA resource-binding pragmas (include/MatrixMultiplication.h:141-153)
 * mean nothing off the FPGA. */
#pragma once
#define HLSLIB_RESOURCE_PRAGMA(variable, resource)
