#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
# Compiles the reference's OWN sources, unmodified and from where they lie under
# /root/reference, against oracle/hlslib_shim (hlslib itself is an absent submodule), into
# oracle/_ref/<config>/:
#     libmmkernel_ref.so   kernel/{Compute,Memory,Top}.cpp   -> extern "C" MatrixMultiplicationKernel
#     TestSimulation       test/TestSimulation.cpp linked against it (the reference's CTest binary)
# It does NOT run the reference's CMake (needs Vitis + hlslib, both absent: SURVEY.md 8c); it
# re-does the two things that CMake did for these files: configure_file(Config.h.in) and the
# -DMM_DYNAMIC_SIZES definition (CMakeLists.txt:96-99,136).  No reference source is copied
# into the repository; outputs are git-ignored build products.
#
# usage: [MM_REF_TRANSPOSED_A=1] build_ref.sh [DATA_TYPE [MAP_OP [REDUCE_OP [TILE_N TILE_M PAR_N PAR_M]]]]
#   MM_REF_TRANSPOSED_A=1  the reference's -DMM_TRANSPOSED_A build (CMakeLists.txt:30,100-103): `a` is K x N.
#                          Only its MM_CONVERT_A branch compiles (kernel/Memory.cpp:205-261), which
#                          Config.h.in:45-48 selects whenever sizeof(Data_t) != the N bus width -- true for every
#                          type at the default 64-byte bus.
set -euo pipefail
REF=${MM_REFERENCE_DIR:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
DT=${1:-float}; MAP=${2:-Multiply}; RED=${3:-Add}
TN=${4:-256}; TM=${5:-256}; PN=${6:-32}; PM=${7:-8}
BUS=64; TW=64
if [ ! -d "$REF/kernel" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) - using prebuilt oracle/_ref if any" >&2
  exit 0
fi
case "$DT" in
  float|int|unsigned) W=4;; double|long|"unsigned long") W=8;; short|"unsigned short"|half) W=2;;
  char|"unsigned char"|uint8_t|int8_t) W=1;; *) echo "unsupported type $DT" >&2; exit 1;;
esac
NAME="${DT// /_}_${MAP}_${RED}_${TN}x${TM}_${PN}x${PM}"
TRANSPOSED=""
if [ "${MM_REF_TRANSPOSED_A:-0}" = 1 ]; then NAME="${NAME}_transposedA"; TRANSPOSED="-DMM_TRANSPOSED_A"; fi
OUT="$HERE/_ref/$NAME"
mkdir -p "$OUT"
KW=$((W * PM))
sed -e "s/\${MM_DATA_WIDTH_\${MM_DATA_TYPE}}/$W/g" \
    -e "s/\${MM_DATA_TYPE}/$DT/g" \
    -e "s/\${MM_MEMORY_BUS_WIDTH_N}/$BUS/g" -e "s/\${MM_MEMORY_BUS_WIDTH_K}/$BUS/g" \
    -e "s/\${MM_MEMORY_BUS_WIDTH_M}/$BUS/g" \
    -e "s/\${MM_SIZE_N}/512/g" -e "s/\${MM_SIZE_K}/512/g" -e "s/\${MM_SIZE_M}/512/g" \
    -e "s/\${MM_MEMORY_TILE_SIZE_N}/$TN/g" -e "s/\${MM_MEMORY_TILE_SIZE_M}/$TM/g" \
    -e "s/\${MM_PARALLELISM_N}/$PN/g" -e "s/\${MM_PARALLELISM_M}/$PM/g" \
    -e "s/\${MM_GRANULARITY_N}/1/g" -e "s/\${MM_TRANSPOSE_WIDTH}/$TW/g" \
    -e "s/\${MM_CLOCK_INTERNAL}/300/g" -e "s#\${MM_GOLDEN_DIR}##g" \
    -e "s/\${MM_MAP_OP}/$MAP/g" -e "s/\${MM_REDUCE_OP}/$RED/g" \
    -e "s/\${MM_KERNEL_WIDTH_M}/$KW/g" \
    "$REF/include/Config.h.in" > "$OUT/Config.h"
CXX=${CXX:-g++}
FLAGS="-std=c++17 -O2 -fPIC -pthread -DMM_DYNAMIC_SIZES $TRANSPOSED -I$OUT -I$REF/include -I$HERE/hlslib_shim"
if [ "$DT" = half ]; then FLAGS="$FLAGS -DMM_HALF_PRECISION -I$HERE/hlslib_shim/hlslib/xilinx"; fi   # CMakeLists.txt:110-112; hls_half.h is a Vitis include-path header
$CXX $FLAGS -shared -o "$OUT/libmmkernel_ref.so" \
    "$REF/kernel/Compute.cpp" "$REF/kernel/Memory.cpp" "$REF/kernel/Top.cpp"
$CXX $FLAGS -o "$OUT/TestSimulation" "$REF/test/TestSimulation.cpp" \
    -L"$OUT" -lmmkernel_ref -Wl,-rpath,'$ORIGIN'
echo "$OUT"
