#!/usr/bin/env python3
"""The reference's `scripts/optimal_memory_tile_size.py`, re-targeted: there the output tile lives in
BRAM and the script finds the largest near-square TN x TM that fits a BRAM budget; here it lives in
the accumulation registers of one compute unit, and the budgets are the CU's register file and LDS.

  python tools/optimal_tile_size.py <data_size_bits> <size_n> <size_m>
         [--acc-bits 32] [--vgpr-kib 512] [--lds-kib 160] [--waves 8] [--mfma 32] [--slab-k 16] [--stages 2]
         [--address-regs 24] [--pow2]

Constraints (one workgroup per CU):
  registers  per wavefront: accumulators + double-buffered operand fragments (one 16-byte read = 4 registers per
             MFMA tile row and column of the wavefront's tile, twice) + addresses <= 512 / (wavefronts per SIMD)
  LDS        stages x (TN + TM) x slab_k x data_size <= LDS
  shape      TN, TM multiples of the MFMA tile times the wavefront grid
Among the feasible tiles the one with the least operand traffic per flop, (1/TN + 1/TM), wins; ties go
to the squarer one, then to the wavefront grid with the fewest fragment reads per MFMA.  Prints the reference script's two lines, then the traffic model of
src/PrintSpecifications.cpp:72-74 for that tile."""
import argparse
import math


def best_tile(data_bits, acc_bits, vgpr_kib, lds_kib, waves, mfma, slab_k, stages, address_regs, pow2=False):
    lanes = 64
    waves_per_simd = max(1, waves // 4)
    regs_per_wave = 512 // waves_per_simd                    # unified VGPR + AGPR budget of a wavefront
    vgpr_bytes = vgpr_kib * 1024
    best = None
    for wm in (1, 2, 4, 8):
        if waves % wm:
            continue
        wn = waves // wm
        for tm_tiles in range(1, 17):                        # MFMA tiles per wavefront along N
            for tn_tiles in range(1, 17):                    # ... along M
                acc_regs = tm_tiles * tn_tiles * (mfma * mfma // lanes) * (acc_bits // 32)
                fragment_regs = 2 * (tm_tiles + tn_tiles) * 4
                if acc_regs + fragment_regs + address_regs > regs_per_wave:
                    continue
                if waves * lanes * 4 * (acc_regs + fragment_regs + address_regs) > vgpr_bytes:
                    continue
                tn, tm = wm * tm_tiles * mfma, wn * tn_tiles * mfma
                if pow2 and (tn & (tn - 1) or tm & (tm - 1)):
                    continue
                if stages * (tn + tm) * slab_k * data_bits // 8 > lds_kib * 1024:
                    continue
                # then: fewest fragment reads per MFMA (squarest wavefront tile), then the taller tile (A rows stream in K-long runs)
                key = (1.0 / tn + 1.0 / tm, abs(tn - tm), (tm_tiles + tn_tiles) / (tm_tiles * tn_tiles), -tn)
                if best is None or key < best[0]:
                    best = (key, tn, tm, wm, wn, acc_regs)
    if best is None:
        raise ValueError("no feasible tile")
    return best[1:]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("data_size_bits", type=int)
    ap.add_argument("size_n", type=int)
    ap.add_argument("size_m", type=int)
    ap.add_argument("--acc-bits", type=int, default=0, help="accumulator width (default: max(32, data size))")
    ap.add_argument("--vgpr-kib", type=int, default=512)
    ap.add_argument("--lds-kib", type=int, default=160)
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--mfma", type=int, default=0, help="MFMA tile edge (default: 16 for 64-bit data, else 32)")
    ap.add_argument("--slab-k", type=int, default=16)
    ap.add_argument("--stages", type=int, default=2)
    ap.add_argument("--address-regs", type=int, default=24)
    ap.add_argument("--pow2", action="store_true", help="power-of-two tiles only (what the shipped kernels use)")
    a = ap.parse_args()
    acc_bits = a.acc_bits or max(32, a.data_size_bits)
    mfma = a.mfma or (16 if a.data_size_bits == 64 else 32)
    tn, tm, wm, wn, acc_regs = best_tile(a.data_size_bits, acc_bits, a.vgpr_kib, a.lds_kib, a.waves, mfma, a.slab_k,
                                         a.stages, a.address_regs, a.pow2)
    print("Tile sizes: {}x{}".format(tn, tm))
    print("Matrix sizes: {}xKx{}".format(tn * math.ceil(a.size_n / tn), tm * math.ceil(a.size_m / tm)))
    print("Wavefront grid: {}x{} ({}x{} per wavefront, {} accumulator registers of {})".format(
        wm, wn, tn // wm, tm // wn, acc_regs, 512 // max(1, a.waves // 4)))
    print("Operand elements streamed per output element and k: {:.6f} (1/TN + 1/TM; the reference's I/O model "
          "N*M*(1 + K/TN + K/TM))".format(1.0 / tn + 1.0 / tm))


if __name__ == "__main__":
    main()
