#!/usr/bin/env python3
"""Regenerates the committed golden fixtures.  Run in the BUILD container (needs g++ and, for
the kernel fixtures, /root/reference so that oracle/build_ref.sh can compile the reference's
own kernel).  The GPU box only reads the fixtures.

  rng_golden.json           first 256 draws of the reference's input generator (libstdc++)
  ref_[transposedA_]<cfg>_<N>x<K>x<M>.npz
                            C computed by the REFERENCE'S OWN kernel sources
                            (kernel/{Compute,Memory,Top}.cpp via oracle/_ref) on the
                            reference's seeded inputs; A and B are not stored (regenerated from
                            the seed by the oracle's generator, whose draws rng_golden.json pins).
  ref_checksums.json        sha256 of C for the reference's CTest shape 513x528x528
                            (CMakeLists.txt:155-159) per config.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402


def main():
    exe = os.path.join(HERE, "rng_reference.bin")
    subprocess.run(["g++", "-O1", "-o", exe, os.path.join(HERE, "rng_reference.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    os.remove(exe)
    json.loads(out)
    with open(os.path.join(HERE, "rng_golden.json"), "w") as f:
        f.write(out)

    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True, stdout=subprocess.DEVNULL)
    small = [(37, 32, 48), (1, 16, 16), (300, 64, 272)]
    checks = {}
    # bus = 64 bytes -> K and M are multiples of 64 / sizeof(Data_t) elements (host/RunHardware.cpp:50-61), and K of the
    # transpose width as well (test/TestSimulation.cpp:28-33); the CTest shape is CMakeLists.txt:155-159's per width.
    # half: the reference's kernel accumulates in binary16 (Xilinx `half`; oracle/hlslib_shim/.../hls_half.h is
    # an IEEE binary16 with round-to-nearest-even) -- the semantics RunHardware's hw_emu mode reproduces.
    # MM_TRANSPOSED_A builds: `a` is K x N and N must be a multiple of the bus width too (SizeNMemory floors,
    # include/MatrixMultiplication.h:61-64) -- which is why the reference's own CTest shape (N = 513) FAILS for that
    # build (observed here: "Mismatch at (0, 0)"); the shapes below keep N % 16 == 0.
    shapes = {
        "half": ([(37, 32, 64), (1, 32, 32), (300, 64, 288)], (513, 544, 544)),
        "double": ([(37, 16, 24), (1, 8, 8), (300, 64, 264)], (513, 520, 520)),
        "uint8_t": ([(37, 64, 64), (1, 64, 64), (300, 128, 320)], (513, 576, 576)),
    }
    transposed_shapes = ([(48, 32, 48), (16, 16, 16), (304, 64, 272)], (528, 528, 528))
    configs = [("float", "Multiply", "Add", False), ("int", "Multiply", "Add", False), ("float", "Add", "Min", False),
               ("half", "Multiply", "Add", False), ("double", "Multiply", "Add", False), ("uint8_t", "Multiply", "Add", False),
               ("float", "Multiply", "Add", True), ("int", "Multiply", "Add", True)]
    for dtype, mp, rd, ta in configs:
        if not _oracle.ref_available(dtype, mp, rd, transposed_a=ta):
            print("skip (oracle/_ref not built):", dtype, mp, rd, "transposedA" if ta else "")
            continue
        cases, ctest = transposed_shapes if ta else shapes.get(dtype, (small, (513, 528, 528)))
        tag = "transposedA_" if ta else ""
        for (n, k, m) in cases:
            a, b = _oracle.fill(dtype, n, k, m, transposed_a=ta)
            c = _oracle.ref_kernel(dtype, mp, rd, a, b, transposed_a=ta)
            np.savez_compressed(os.path.join(HERE, f"ref_{tag}{dtype}_{mp}_{rd}_{n}x{k}x{m}.npz"), c=c,
                                a_sha256=hashlib.sha256(a.tobytes()).hexdigest(),
                                b_sha256=hashlib.sha256(b.tobytes()).hexdigest())
        n, k, m = ctest
        a, b = _oracle.fill(dtype, n, k, m, transposed_a=ta)
        c = _oracle.ref_kernel(dtype, mp, rd, a, b, transposed_a=ta)
        checks[f"{tag}{dtype}_{mp}_{rd}"] = {
            "shape": [n, k, m],
            "a_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "b_sha256": hashlib.sha256(b.tobytes()).hexdigest(),
            "c_sha256": hashlib.sha256(c.tobytes()).hexdigest(),
        }
    with open(os.path.join(HERE, "ref_checksums.json"), "w") as f:
        json.dump(checks, f, indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
