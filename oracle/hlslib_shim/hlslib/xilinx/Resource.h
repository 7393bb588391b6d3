// TEST-ONLY shim: resource pragmas mean nothing off-FPGA.
#pragma once
#define HLSLIB_RESOURCE_PRAGMA(var, res)
