"""Random shapes, data types, layouts and device counts through mm_gemm_multi_device over VIRTUAL devices (md_virtual_devices:
logical devices dealt out over the physical ones), every result bit for bit against the one-device call.  fp32 shapes that one
device runs as stream-K are counted but compared by value only (slabs run whole tiles there: documented exception).
  python tools/fuzz_multi_device.py [--cases 200] [--seed 1]"""
import argparse, collections, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemm_hls_amd as g
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=200); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
g.set_tuning("md_virtual_devices", 8)
TYPES = [("float", np.float32, 8, 4), ("double", np.float64, 8, 2), ("half", np.float16, 16, 8), ("uint8_t", np.uint8, 32, 16), ("int", np.int32, 4, 4)]
met = collections.Counter()
for i in range(args.cases):
    dtype, npdt, km, mm = TYPES[i % len(TYPES)]
    devices = int(rng.choice([2, 3, 4, 5, 8]))
    kxn = bool(rng.integers(0, 2))
    n = int(rng.integers(1, 5000))
    if kxn and rng.integers(0, 2):
        n = max(16, n // 16 * 16)                      # half of the K x N cases qualify for the matrix-core K x N kernels
    k = max(km, int(rng.integers(km, 1500)) // km * km)
    m = max(mm, int(rng.integers(mm, 3000)) // mm * mm)
    if np.issubdtype(npdt, np.integer):
        a = rng.integers(0, 11, size=(n, k)).astype(npdt); b = rng.integers(0, 11, size=(k, m)).astype(npdt)
    else:
        a = rng.uniform(1, 4, size=(n, k)).astype(npdt); b = rng.uniform(1, 4, size=(k, m)).astype(npdt)
    src = np.ascontiguousarray(a.T) if kxn else a
    name = g.kernel_name(g.make_config(dtype, transposed_a=kxn), n, k, m)
    c1, _ = g.matmul_host(src, b, dtype, devices=1, transposed_a=kxn)
    cg, _ = g.matmul_host(src, b, dtype, devices=devices, transposed_a=kxn)
    slabs = [g.row_slab(g.make_config(dtype, transposed_a=kxn), n, k, m, devices, r) for r in range(devices)]
    assert sum(r for _, r in slabs) == n
    if "streamk" in name:
        exact = a.astype(np.float64) @ b.astype(np.float64)
        assert np.max(np.abs(cg - exact) / exact) < 1e-5, (dtype, n, k, m, devices, kxn, name)
        met["stream-K on one device: slabs compared by value"] += 1
    else:
        assert np.array_equal(cg.view(np.uint8), c1.view(np.uint8)), (dtype, n, k, m, devices, kxn, name, slabs)
        met[("K x N " if kxn else "") + name] += 1
print(f"{args.cases} cases ok (devices in 2/3/4/5/8, row-major and K x N A), bitwise equal to one device")
for name, cnt in met.most_common():
    print(f"{cnt:6d}  {name}")
