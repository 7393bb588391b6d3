#!/usr/bin/env python3
"""Fingerprints the gfx950 ISA of every kernel in a .hip file: compiles device-only to assembly, splits by kernel,
drops comments / directives, renames local labels by order of appearance and hashes what is left.  Used to show that a
source refactoring left the shipped kernels' machine code unchanged (names change with template parameters, bodies
must not).   python tools/lab/isa_fingerprint.py gemm_hls_amd/csrc/mm_mfma_f32.hip [-D...]"""
import hashlib
import re
import subprocess
import sys


def fingerprints(src, extra=()):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                          "--cuda-device-only", "-Wno-unused-function", *extra, src, "-o", "-"],
                         capture_output=True, text=True, check=True).stdout
    out = {}
    lines = asm.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*; @", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        body = []
        i += 1
        while i < len(lines) and "s_endpgm" not in lines[i]:
            t = lines[i].split(";")[0].rstrip()
            if t.strip() and not t.strip().startswith("."):
                body.append(t.strip())
            elif re.match(r"^\.LBB\d+_\d+:", t.strip()):
                body.append(t.strip())
            i += 1
        labels = {}
        norm = []
        for t in body:
            for lab in re.findall(r"\.LBB\d+_\d+", t):
                labels.setdefault(lab, f"L{len(labels)}")
            norm.append(re.sub(r"\.LBB\d+_\d+", lambda mm: labels[mm.group(0)], t))
        h = hashlib.sha256("\n".join(norm).encode()).hexdigest()[:16]
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out[demangled] = (h, len(norm), sum("v_mfma" in t for t in norm))
    return out


if __name__ == "__main__":
    for k, v in sorted(fingerprints(sys.argv[1], sys.argv[2:]).items()):
        print(v[0], v[1], v[2], k)
