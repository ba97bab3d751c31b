"""MFMA utilisation per launch from the counter summary of the per-shape --pmc passes in tools/profile_r02.sh (one pair
of passes per opbench shape, rows labelled with the shape):  utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x
SQ_BUSY_CU_CYCLES), i.e. the share of the CUs' busy cycles in which a SIMD's matrix pipe is executing;
v_mfma_f32_32x32x2_f32 holds the pipe 64 cycles: BUSY = 64 x SQ_INSTS_VALU_MFMA_F32.  The clock the chip held is
SQ_BUSY_CU_CYCLES / 256 CUs / the launch's duration (the matrix kernels run near 2.0 GHz, not the 2.4 GHz of the peak).

usage: python tools/mfma_util.py <pmc_mfma_summary.csv>"""
import csv
import sys
from collections import defaultdict


def main():
    vals = defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1], newline="")):
        if "pairwise_distance" in r["kernel"]:
            vals[(r["shape"], r["kernel"], int(r["grid_threads"]))][r["counter"]] = float(r["mean_value"])
    print("fp32-input MFMA 32x32x2 (64 cycles each); 256 CUs x 4 SIMDs")
    print(f"{'opbench shape':22s} {'kernel':44s} {'workgroups':>10s} {'MFMA instr':>12s} {'MFMA busy cyc':>14s} "
          f"{'CU busy cyc':>12s} {'MFMA util':>9s} {'waves/SIMD':>10s}")
    for (shape, kernel, grid), c in sorted(vals.items()):
        busy = c["SQ_BUSY_CU_CYCLES"]
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * busy)
        occ = c["SQ_WAVE_CYCLES"] / busy  # resident wave-cycles per busy CU cycle, per SIMD
        print(f"{shape:22s} {kernel.replace('v3d::', ''):44s} {grid // 256:10d} {c['SQ_INSTS_VALU_MFMA_F32']:12.0f} "
              f"{c['SQ_VALU_MFMA_BUSY_CYCLES']:14.0f} {busy:12.0f} {util:9.3f} {occ:10.2f}")
    print("(utilisation is a ratio of cycles: against the 157 TFLOP/s peak at 2.4 GHz the achieved fraction is lower by the "
          "clock ratio and by the time no workgroup is resident)")


if __name__ == "__main__":
    main()
