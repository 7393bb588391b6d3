"""Stream-K rate sweep: shape-adaptive pick / whole tiles only / teams, the last arriver gathers (f32_splitk 0, what auto runs) /
teams + fix-up kernel (11) / single ranges + fix-up kernel (9)."""
import sys
sys.path.insert(0, "tools"); from _lib import g
import sweep
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(2176, 4097, 128)) + list(range(4352, 6401, 256))
def rate(s, sk, var):
    g.set_tuning("f32_splitk", sk); g.set_tuning("f32_variant", var)
    try:
        med, best = sweep.time_config("float", "Multiply", "Add", s, s, s, 9)
        return round(2.0 * s ** 3 / med / 1e12, 1)
    finally:
        g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
for s in sizes:
    t = (s + 127) // 128
    name = g.kernel_name(g.make_config("float"), s, s, s)
    print(s, f"{t}x{t} tiles: auto", rate(s, -1, -1), "whole tiles", rate(s, 1, -1), "stream-K last-arriver", rate(s, 0, 35), "teams+fixup", rate(s, 11, 35), "single-range fix-up", rate(s, 9, 35), " auto =", name, flush=True)
