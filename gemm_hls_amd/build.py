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
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file extra flags
EXTRA = {
    # the parity anchor must not contract a*b+c into an fma (reference Naive is unfused)
    "mm_ordered.hip": ["-ffp-contract=off"],
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
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print(f"built {LIB} ({len(jobs)} objects recompiled)")

    host_dir = os.path.join(HERE, "host")
    host_jobs = []
    host_hdrs = [os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "mm_gemm.h")]
    rpath = "-Wl,-rpath,$ORIGIN/../gemm_hls_amd"
    for (dt, mp, rd) in HOST_CONFIGS:
        names = [f"RunHardware_{dt}_{mp}_{rd}.exe"] + (["RunHardware.exe"] if (dt, mp, rd) == HOST_CONFIGS[0] else [])
        src = os.path.join(host_dir, "RunHardware.cpp")
        if not os.path.exists(src):
            break
        for name in names:
            out = os.path.join(BIN, name)
            if newer(out, [src, LIB] + host_hdrs + [__file__]):
                # amdclang++ (not g++): the host needs _Float16 for MM_DATA_TYPE=half
                host_jobs.append(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-pthread",
                                  f"-DMM_DATA_TYPE={dt}", f"-DMM_MAP_OP={mp}", f"-DMM_REDUCE_OP={rd}",
                                  "-DMM_DYNAMIC_SIZES", "-I" + os.path.join(ROOT, "include"), "-I" + host_dir,
                                  src, "-o", out, "-L" + HERE, "-lmm_gemm_amd", rpath, "-ldl"])
    # static-size build (MM_DYNAMIC_SIZES=OFF, CMakeLists.txt:21-24): sizes baked in, argv = [mode] [verify]
    src = os.path.join(host_dir, "RunHardware.cpp")
    out = os.path.join(BIN, "RunHardware_static_float_528x512x560.exe")
    if os.path.exists(src) and newer(out, [src, LIB] + host_hdrs + [__file__]):
        host_jobs.append(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-pthread", "-DMM_DATA_TYPE=float",
                          "-DMM_MAP_OP=Multiply", "-DMM_REDUCE_OP=Add", "-DMM_SIZE_N=528", "-DMM_SIZE_K=512",
                          "-DMM_SIZE_M=560", "-I" + os.path.join(ROOT, "include"), "-I" + host_dir, src, "-o", out,
                          "-L" + HERE, "-lmm_gemm_amd", rpath, "-ldl"])
    for extra in ("PrintSpecifications", "TestSimulation"):
        src = os.path.join(host_dir, extra + ".cpp")
        out = os.path.join(BIN, extra + ".exe")
        if os.path.exists(src) and newer(out, [src, LIB] + host_hdrs + [__file__]):
            host_jobs.append(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-pthread", "-DMM_DATA_TYPE=float",
                              "-DMM_MAP_OP=Multiply", "-DMM_REDUCE_OP=Add", "-DMM_DYNAMIC_SIZES",
                              "-I" + os.path.join(ROOT, "include"), "-I" + host_dir, src, "-o", out,
                              "-L" + HERE, "-lmm_gemm_amd", rpath, "-ldl"])
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
        list(ex.map(run, host_jobs))
    if verbose and host_jobs:
        print(f"built {len(host_jobs)} host binaries in {BIN}")
    return LIB


if __name__ == "__main__":
    build()
