import sys, numpy as np
sys.path.insert(0, "tools"); from _lib import g
import sweep
rng = np.random.default_rng(0)
for (n, k, m) in [(1024, 1024, 1024), (300, 2048, 272), (640, 512, 384), (2560, 2560, 2560), (129, 4096, 132), (1000, 96, 3000), (3072, 1056, 520)]:
    a = rng.uniform(-3, 10, (n, k)).astype(np.float32); b = rng.uniform(-3, 10, (k, m)).astype(np.float32)
    g.set_tuning("f32_variant", 35); g.set_tuning("f32_splitk", 0)
    name = g.kernel_name(g.make_config("float"), n, k, m)
    c1, _ = g.matmul_capi(a, b); c2, _ = g.matmul_capi(a, b)
    g.set_tuning("f32_splitk", 1)
    c0, _ = g.matmul_capi(a, b)
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    exact = a.astype(np.float64) @ b.astype(np.float64)
    print((n, k, m), name, "deterministic", np.array_equal(c1, c2), "err", float(np.max(np.abs(c1 - exact) / scale)), "vs unsplit", float(np.max(np.abs(c1 - c0) / scale)), flush=True)
for s in (1024, 1536, 1792, 2048, 2304, 2560, 2816, 3072, 3328, 3584, 4096, 5120, 6144):
    out = []
    for sk, var in ((-1, -1), (0, 35), (1, 35), (1, 33)):
        g.set_tuning("f32_splitk", sk); g.set_tuning("f32_variant", var)
        med, best = sweep.time_config("float", "Multiply", "Add", s, s, s, 7)
        out.append(round(2.0 * s ** 3 / med / 1e12, 1))
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    print(s, "auto / streamk / 128x128 plain / 128x256 plain TF:", out, flush=True)
