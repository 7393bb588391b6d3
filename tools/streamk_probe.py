"""Stream-K probe: correctness of the combine forms (f32_splitk 0 = teams, the last arriver gathers, 11 = teams + fix-up kernel,
9 = single ranges + fix-up kernel)
against float64 on ragged shapes, run-to-run bit identity, and their rates next to the whole-tile kernels."""
import sys, numpy as np
sys.path.insert(0, "tools"); from _lib import g
import sweep
rng = np.random.default_rng(0)
for (n, k, m) in [(1024, 1024, 1024), (300, 2048, 272), (640, 512, 384), (2560, 2560, 2560), (129, 4096, 132), (1000, 96, 3000), (3072, 1056, 520),
                  (2341, 2304, 2304), (4000, 64, 4000), (128, 32768, 128)]:
    a = rng.uniform(-3, 10, (n, k)).astype(np.float32); b = rng.uniform(-3, 10, (k, m)).astype(np.float32)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    exact = a.astype(np.float64) @ b.astype(np.float64)
    g.set_tuning("f32_variant", 35); g.set_tuning("f32_splitk", 1)
    c0, _ = g.matmul_capi(a, b)
    for knob in (0, 11, 9):
        g.set_tuning("f32_splitk", knob)
        name = g.kernel_name(g.make_config("float"), n, k, m)
        cs = [g.matmul_capi(a, b)[0] for _ in range(4)]
        print((n, k, m), name, "deterministic", all(np.array_equal(cs[0], c) for c in cs[1:]), "err", float(np.max(np.abs(cs[0] - exact) / scale)),
              "vs unsplit", float(np.max(np.abs(cs[0] - c0) / scale)), flush=True)
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
for s in (1024, 1536, 1792, 2048, 2304, 2560, 2816, 3072, 3328, 3584, 4096, 5120, 6144):
    out = []
    for sk, var in ((-1, -1), (9, 35), (0, 35), (1, 35), (1, 33)):
        g.set_tuning("f32_splitk", sk); g.set_tuning("f32_variant", var)
        med, best = sweep.time_config("float", "Multiply", "Add", s, s, s, 7)
        out.append(round(2.0 * s ** 3 / med / 1e12, 1))
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    print(s, "auto / streamk / streamk ordered / 128x128 plain / 128x256 plain TF:", out, flush=True)
