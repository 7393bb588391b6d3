#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds the reference's OWN host programs, unmodified and from where they lie under
/root/reference, against THIS repository's library -- the strongest form of the drop-in claim (missing 1):

    /root/reference/host/RunHardware.cpp        -> bin/ref_hosts/<config>/RunHardware.exe
    /root/reference/test/TestSimulation.cpp     -> bin/ref_hosts/<config>/TestSimulation.exe
    /root/reference/src/PrintSpecifications.cpp -> bin/ref_hosts/<config>/PrintSpecifications.exe

The only things added to the reference's compile line are one include path for the hlslib names its host uses
(include/compat: hlslib::ocl::{Context, Program, Kernel, Buffer, ...} over include/mm_gemm.h; hlslib itself is an absent
submodule of the reference) and the libraries to link: gemm_hls_amd/libmm_gemm_amd.so for RunHardware (which reaches the
device through hlslib::ocl only), and for TestSimulation the build-time configured kernel library bin/libmmkernel*.so in
the place of the reference's `mmkernel` target (CMakeLists.txt:138-150).  The reference's CMake is NOT run (it needs
Vitis); this script re-does what it did for these three files: configure_file(include/Config.h.in -> Config.h)
(CMakeLists.txt:136) and the definitions MM_DYNAMIC_SIZES / MM_TRANSPOSED_A / MM_HALF_PRECISION
(CMakeLists.txt:96-112).  No reference source is copied into the repository: the generated Config.h and the binaries go
to bin/ref_hosts/ (git-ignored; travels to the GPU box with the snapshot, where /root/reference does not exist).

usage: build_ref_hosts.py            (all configurations below; a no-op with a note when /root/reference is absent)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("MM_REFERENCE_DIR", "/root/reference")
OUT = os.path.join(ROOT, "bin", "ref_hosts")
CXX = os.environ.get("MM_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")   # half needs _Float16 on the host: clang, not g++ 11
WIDTH = {"float": 4, "int": 4, "unsigned": 4, "double": 8, "long": 8, "half": 2, "short": 2, "uint8_t": 1, "char": 1}

# The BLAS the reference's oracle is linked against in the *_blas configurations (its CMake: find_package(BLAS), -DMM_HAS_BLAS,
# CMakeLists.txt:75-85,151,205).  This image ships MKL's single dynamic library without headers: tests/ref_hosts/blas/cblas.h
# declares the two CBLAS entry points include/Utility.h:76-103 calls.
SYSTEM_LIB_DIR = "/usr/lib/x86_64-linux-gnu"
BLAS_LIBRARY = os.environ.get("MM_BLAS_LIBRARY", "/opt/conda/lib/libmkl_rt.so")

# name -> (MM_DATA_TYPE, MM_MAP_OP, MM_REDUCE_OP, transposed A, static sizes or None, kernel library for TestSimulation or None)
STATIC = (528, 512, 560)
CONFIGS = {
    "float": ("float", "Multiply", "Add", False, None, "libmmkernel.so"),
    # half: the reference's verification is EXACT for half (its `half` class is not std::is_floating_point), against a Naive
    # that accumulates in binary16 -- so its TestSimulation gets the kernel library over the k-ordered kernel
    "half": ("half", "Multiply", "Add", False, None, "libmmkernel_half_ordered.so"),
    # the same half build with -DMM_HALF_CONTRACT_REFERENCE (include/compat): "hw" keeps the reference's half arithmetic on the
    # k-ordered tile kernel, so the reference's own exact comparison holds for the timed kernel too
    "half_reference_contract": ("half", "Multiply", "Add", False, None, None),
    "float_transposedA": ("float", "Multiply", "Add", True, None, "libmmkernel_transposedA.so"),
    "float_static_%dx%dx%d" % STATIC: ("float", "Multiply", "Add", False, STATIC, "libmmkernel_static_float_%dx%dx%d.so" % STATIC),
    "float_Add_Min": ("float", "Add", "Min", False, None, None),
    "int": ("int", "Multiply", "Add", False, None, None),
    "double": ("double", "Multiply", "Add", False, None, None),
    # the reference's hosts WITH its BLAS oracle (-DMM_HAS_BLAS): "Running BLAS..." instead of the Naive fall-back, so they can
    # verify BASELINE C2 / C4 (16384^3) themselves.  (No transposed-A BLAS build: the reference passes lda = size_k with
    # CblasTrans, include/Utility.h:86-87,99-100 -- right only for N == K; a reference defect)
    "float_blas": ("float", "Multiply", "Add", False, None, "libmmkernel.so"),
    "double_blas": ("double", "Multiply", "Add", False, None, "libmmkernel_double.so"),
}


def configure_file(dtype, map_op, reduce_op, sizes, out_path):
    """CMake's configure_file on the reference's include/Config.h.in, with the reference's defaults (CMakeLists.txt:16-36)."""
    par_m = 8
    values = {
        "MM_DATA_TYPE": dtype, "MM_MEMORY_BUS_WIDTH_N": 64, "MM_MEMORY_BUS_WIDTH_K": 64, "MM_MEMORY_BUS_WIDTH_M": 64,
        "MM_SIZE_N": (sizes or (512, 512, 512))[0], "MM_SIZE_K": (sizes or (512, 512, 512))[1], "MM_SIZE_M": (sizes or (512, 512, 512))[2],
        "MM_MEMORY_TILE_SIZE_N": 256, "MM_MEMORY_TILE_SIZE_M": 256, "MM_PARALLELISM_N": 32, "MM_PARALLELISM_M": par_m,
        "MM_GRANULARITY_N": 1, "MM_TRANSPOSE_WIDTH": 64, "MM_CLOCK_INTERNAL": 300, "MM_GOLDEN_DIR": "",
        "MM_MAP_OP": map_op, "MM_REDUCE_OP": reduce_op, "MM_KERNEL_WIDTH_M": WIDTH[dtype] * par_m,
        "MM_DATA_WIDTH_" + dtype: WIDTH[dtype],
    }
    text = open(os.path.join(REF, "include", "Config.h.in")).read()
    for _ in range(2):   # ${MM_DATA_WIDTH_${MM_DATA_TYPE}} is nested
        text = re.sub(r"\$\{(\w+)\}", lambda mo: str(values[mo.group(1)]) if mo.group(1) in values else mo.group(0), text)
    left = re.findall(r"\$\{\w+\}", text)
    if left:
        raise SystemExit(f"Config.h.in names variables this recipe does not set: {sorted(set(left))}")
    with open(out_path, "w") as f:
        f.write(text)


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write("FAILED: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise SystemExit(1)


def build(verbose=True):
    if not os.path.isdir(os.path.join(REF, "host")):
        if verbose:
            print(f"build_ref_hosts: {REF} not present (GPU box?) - using prebuilt bin/ref_hosts if any")
        return []
    lib_dir = os.path.join(ROOT, "gemm_hls_amd")
    built = []
    for name, (dtype, map_op, reduce_op, transposed, sizes, kernel_lib) in CONFIGS.items():
        out = os.path.join(OUT, name)
        os.makedirs(out, exist_ok=True)
        configure_file(dtype, map_op, reduce_op, sizes, os.path.join(out, "Config.h"))
        flags = ["-std=c++17", "-O2", "-pthread", "-I" + out, "-I" + os.path.join(REF, "include"),
                 "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include")]
        flags += [] if sizes else ["-DMM_DYNAMIC_SIZES"]
        flags += ["-DMM_TRANSPOSED_A"] if transposed else []
        flags += ["-DMM_HALF_PRECISION"] if dtype == "half" else []
        flags += ["-DMM_HALF_CONTRACT_REFERENCE"] if name == "half_reference_contract" else []
        rpath = "-Wl,-rpath,$ORIGIN/../../../gemm_hls_amd:$ORIGIN/../.."
        device = ["-L" + lib_dir, "-lmm_gemm_amd", rpath]
        if name.endswith("_blas"):
            if not os.path.exists(BLAS_LIBRARY):
                if verbose:
                    print(f"build_ref_hosts: {BLAS_LIBRARY} not found - skipping {name} (set MM_BLAS_LIBRARY)")
                continue
            flags += ["-DMM_HAS_BLAS", "-I" + os.path.join(ROOT, "tests", "ref_hosts", "blas")]
            # the BLAS's directory goes LAST on the run path, behind the system's library directory: /opt/conda/lib also holds an
            # older libstdc++ than the one libmm_gemm_amd.so was linked against, which must not be the one that gets loaded
            device += [BLAS_LIBRARY, "-Wl,--disable-new-dtags", "-Wl,-rpath," + SYSTEM_LIB_DIR + ":" + os.path.dirname(BLAS_LIBRARY)]
        run([CXX] + flags + [os.path.join(REF, "host", "RunHardware.cpp"), "-o", os.path.join(out, "RunHardware.exe")] + device)
        built.append(os.path.join(out, "RunHardware.exe"))
        if kernel_lib:   # the reference's `mmkernel` role: first in link order, so its MatrixMultiplicationKernel is the one bound
            run([CXX] + flags + [os.path.join(REF, "test", "TestSimulation.cpp"), "-o", os.path.join(out, "TestSimulation.exe"),
                                 "-L" + os.path.join(ROOT, "bin"), "-l:" + kernel_lib] + device)
            built.append(os.path.join(out, "TestSimulation.exe"))
        if name == "float":
            run([CXX] + flags + [os.path.join(REF, "src", "PrintSpecifications.cpp"), "-o", os.path.join(out, "PrintSpecifications.exe")])
            built.append(os.path.join(out, "PrintSpecifications.exe"))
    if verbose:
        print(f"built {len(built)} reference host programs (unmodified sources from {REF}) in {OUT}")
    return built


if __name__ == "__main__":
    build()
