#!/usr/bin/env python3
"""Builds the product, in-tree (so the .so travels with the repo snapshot to the GPU box):

    gemm_hls_amd/libmm_gemm_amd.so   every HIP kernel family + the C ABI (include/mm_gemm.h)
    bin/RunHardware.exe              C++ host runner, default config (float, Multiply, Add)
    bin/RunHardware_<T>_<Map>_<Reduce>.exe  further build-time configurations, like the
                                     reference's one-binary-per-(MM_DATA_TYPE, MM_MAP_OP,
                                     MM_REDUCE_OP) model (CMakeLists.txt:16-34)
    bin/PrintSpecifications.exe ...  (when present)

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached by source mtime.
"""
import concurrent.futures
import os
import shlex
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
BIN = os.path.join(ROOT, "bin")
LIB = os.path.join(HERE, "libmm_gemm_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# --offload-compress: the device code objects are stored zstd-compressed (the generic-semiring families are 11 types x
# 25 operator pairs of kernels; 7.3 MB -> ~2 MB), the runtime inflates them when the library is loaded
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress"]
# per-file extra flags
EXTRA = {
    # the parity anchor must not contract a*b+c into an fma (reference Naive is unfused)
    "mm_ordered.hip": ["-ffp-contract=off"],
    # the register-tiled kernels under the same k-ordered, unfused contract ("ordered_tile")
    "mm_valu_tile_fp_exact.hip": ["-ffp-contract=off"],
}

HOST_CONFIGS = [  # (Data_t, MM_MAP_OP, MM_REDUCE_OP): BASELINE.json configs + an integer semiring
    ("float", "Multiply", "Add"),
    ("double", "Multiply", "Add"),
    ("half", "Multiply", "Add"),
    ("float", "Add", "Min"),
    ("int", "Multiply", "Add"),
    ("uint8_t", "Multiply", "Add"),
]


def newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write("FAILED: " + " ".join(shlex.quote(c) for c in cmd) + "\n" + r.stdout + r.stderr)
        raise SystemExit(1)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


LAB_DIR = os.path.join(ROOT, "tools", "lab")
LAB_LIB = os.path.join(LAB_DIR, "libmm_gemm_amd_lab.so")
LAB_REPLACES = {"lab_mfma_f32.hip": "mm_mfma_f32.o", "lab_mfma_f16.hip": "mm_mfma_f16.o", "lab_mfma_i8.hip": "mm_mfma_i8.o",
                "lab_mfma_f32_split.hip": "mm_mfma_f32_split.o"}


def build_lab(product_objs, headers, verbose=True):
    """tools/lab/libmm_gemm_amd_lab.so: the SAME C ABI with the four matrix-core translation units swapped for their
    lab editions (every schedule / ring depth / ablation the kernels went through, incl. the ones that skip work on
    purpose).  Measurement tooling only (MM_LIB=lab for tools/sweep.py and friends); nothing in the product links it."""
    if not os.path.isdir(LAB_DIR) or os.environ.get("MM_BUILD_LAB") != "1":
        return   # measurement tooling: built on request only (MM_BUILD_LAB=1), not by every build()
    os.makedirs(os.path.join(LAB_DIR, "_obj"), exist_ok=True)
    jobs, lab_objs = [], []
    for src, replaced in LAB_REPLACES.items():
        obj = os.path.join(LAB_DIR, "_obj", src[:-4] + ".o")
        lab_objs.append(obj)
        if newer(obj, [os.path.join(LAB_DIR, src)] + headers + [__file__]):
            jobs.append([HIPCC] + COMMON + ["-DMM_LAB_BUILD", "-c", os.path.join(LAB_DIR, src), "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(run, jobs))
    keep = [o for o in product_objs if os.path.basename(o) not in LAB_REPLACES.values()]
    if jobs or newer(LAB_LIB, keep + lab_objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "--offload-compress", "-o", LAB_LIB] + keep + lab_objs)
    if verbose:
        print(f"built {LAB_LIB} ({len(jobs)} lab objects recompiled)")


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(BIN, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(ROOT, "include", "mm_gemm.h"))
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = []
    objs = []
    for src in sources:
        obj = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(obj)
        if newer(obj, [os.path.join(CSRC, src)] + headers + [__file__]):
            jobs.append([HIPCC] + COMMON + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(run, jobs))
    if jobs or newer(LIB, objs):   # also after an object was rebuilt by hand
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "--offload-compress", "-o", LIB] + objs)
    if verbose:
        print(f"built {LIB} ({len(jobs)} objects recompiled)")
    build_lab(objs, headers, verbose)

    host_dir = os.path.join(HERE, "host")
    host_jobs = []
    shim_jobs = []
    host_hdrs = [os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "mm_gemm.h")]
    rpath = "-Wl,-rpath,$ORIGIN/../gemm_hls_amd:$ORIGIN"
    cxx = "/opt/rocm/lib/llvm/bin/clang++"  # amdclang++ (not g++): the host needs _Float16 for MM_DATA_TYPE=half
    common = [cxx, "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + host_dir]

    def defs(dt, mp, rd, sizes=None, transposed=False, power=False):
        d = [f"-DMM_DATA_TYPE={dt}", f"-DMM_MAP_OP={mp}", f"-DMM_REDUCE_OP={rd}"]
        d += ["-DMM_DYNAMIC_SIZES"] if sizes is None else [f"-DMM_SIZE_{ax}={v}" for ax, v in zip("NKM", sizes)]
        if transposed:
            d.append("-DMM_TRANSPOSED_A")
        if power:
            d.append("-DMM_POWER_METER")
        return d

    def exe(name, source, d, shim=None):
        src = os.path.join(host_dir, source)
        out = os.path.join(BIN, name)
        deps = [src, LIB] + host_hdrs + [__file__] + ([os.path.join(BIN, shim)] if shim else [])
        if os.path.exists(src) and newer(out, [p for p in deps if os.path.exists(p)] or deps):
            # the shim goes first: its MatrixMultiplicationKernel is the one the caller binds to
            link = (["-L" + BIN, "-l:" + shim] if shim else []) + ["-L" + HERE, "-lmm_gemm_amd", rpath, "-ldl"]
            host_jobs.append(common + d + [src, "-o", out] + link)

    def shim(name, d):
        """Build-time configured kernel library (the reference's `mmkernel`, CMakeLists.txt:138-146)."""
        src = os.path.join(host_dir, "KernelShim.cpp")
        out = os.path.join(BIN, name)
        if os.path.exists(src) and newer(out, [src, LIB] + host_hdrs + [__file__]):
            shim_jobs.append(common + d + ["-shared", "-fPIC", src, "-o", out, "-L" + HERE, "-lmm_gemm_amd",
                                           "-Wl,-rpath,$ORIGIN/../gemm_hls_amd"])

    for (dt, mp, rd) in HOST_CONFIGS:
        exe(f"RunHardware_{dt}_{mp}_{rd}.exe", "RunHardware.cpp", defs(dt, mp, rd))
        # MM_POWER_METER builds (CMakeLists.txt:13,212-214), what tools/benchmark.py drives
        exe(f"RunHardware_{dt}_{mp}_{rd}_power.exe", "RunHardware.cpp", defs(dt, mp, rd, power=True))
    dt, mp, rd = HOST_CONFIGS[0]
    exe("RunHardware.exe", "RunHardware.cpp", defs(dt, mp, rd))
    # static-size build (MM_DYNAMIC_SIZES=OFF, CMakeLists.txt:21-24): sizes baked in, argv = [mode] [verify]
    static = (528, 512, 560)
    tag = "x".join(map(str, static))
    exe(f"RunHardware_static_float_{tag}.exe", "RunHardware.cpp", defs(dt, mp, rd, static))
    # MM_TRANSPOSED_A builds (CMakeLists.txt:30,100-103): A generated / handed over / verified as K x N
    exe("RunHardware_transposedA_float.exe", "RunHardware.cpp", defs("float", "Multiply", "Add", transposed=True))
    exe("RunHardware_transposedA_int.exe", "RunHardware.cpp", defs("int", "Multiply", "Add", transposed=True))
    # a half build that keeps the reference's half arithmetic under "hw" (binary16 accumulating in binary16): timed and
    # verified -- exactly -- on the same kernel
    exe("RunHardware_half_reference_contract.exe", "RunHardware.cpp", defs("half", mp, rd) + ["-DMM_HALF_CONTRACT_REFERENCE"])
    exe("RunHardware_power.exe", "RunHardware.cpp", defs(dt, mp, rd, power=True))
    # the reference's tile knob (CMakeLists.txt:18-20): a build pinned to the 256 x 256 resident tile
    pin = ["-DMM_MEMORY_TILE_SIZE_N=256", "-DMM_MEMORY_TILE_SIZE_M=256"]
    exe("RunHardware_tile256x256.exe", "RunHardware.cpp", defs(dt, mp, rd) + pin)
    exe("PrintSpecifications_tile256x256.exe", "PrintSpecifications.cpp", defs(dt, mp, rd) + pin)
    # kernel shims + the CTest binary bound to them: dynamic, static (3-pointer symbol) and K x N A
    shim("libmmkernel.so", defs(dt, mp, rd))
    shim(f"libmmkernel_static_float_{tag}.so", defs(dt, mp, rd, static))
    shim("libmmkernel_transposedA.so", defs(dt, mp, rd, transposed=True))
    shim("libmmkernel_half.so", defs("half", mp, rd))
    shim("libmmkernel_double.so", defs("double", mp, rd))
    # the same library over the k-ordered kernel (MM_PATH_ORDERED): the bit-true "simulation" build.  The reference's own
    # test/TestSimulation.cpp compares half results EXACTLY with its binary16-accumulating Naive (its `half` is not
    # std::is_floating_point, test/TestSimulation.cpp:80-85) -- tests/ref_hosts links it against this one
    shim("libmmkernel_half_ordered.so", defs("half", mp, rd) + ["-DMM_DEFAULT_PATH=MM_PATH_ORDERED"])
    exe("TestSimulation.exe", "TestSimulation.cpp", defs(dt, mp, rd), shim="libmmkernel.so")
    exe(f"TestSimulation_static_float_{tag}.exe", "TestSimulation.cpp", defs(dt, mp, rd, static),
        shim=f"libmmkernel_static_float_{tag}.so")
    exe("TestSimulation_transposedA.exe", "TestSimulation.cpp", defs(dt, mp, rd, transposed=True),
        shim="libmmkernel_transposedA.so")
    exe("TestSimulation_half.exe", "TestSimulation.cpp", defs("half", mp, rd), shim="libmmkernel_half.so")
    exe("PrintSpecifications.exe", "PrintSpecifications.cpp", defs(dt, mp, rd))
    # hardware probes cited by profiles/ (stand-alone HIP programs, not part of the product)
    probe_dir = os.path.join(ROOT, "tools", "probes")
    if os.path.isdir(probe_dir):
        for f in sorted(os.listdir(probe_dir)):
            if f.endswith(".hip"):
                out = os.path.join(BIN, f[:-4])
                if newer(out, [os.path.join(probe_dir, f)]):
                    host_jobs.append([HIPCC, "--offload-arch=" + ARCH, "-O2", "-w",
                                      os.path.join(probe_dir, f), "-o", out])
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(run, shim_jobs))      # the shims first: the CTest binaries link against them
        list(ex.map(run, host_jobs))
    if verbose and host_jobs:
        print(f"built {len(host_jobs)} host binaries in {BIN}")
    return LIB


if __name__ == "__main__":
    build()
