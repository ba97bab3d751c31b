"""MFMA utilisation per launch from the counter summary (tools/rocpd_export.py counters) of the two --pmc passes in
tools/profile_r02.sh:  utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles =
GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back").  v_mfma_f32_32x32x2_f32 holds the
pipe 64 cycles: BUSY = 64 x SQ_INSTS_VALU_MFMA_F32.

usage: python tools/mfma_util.py <pmc_mfma_summary.csv>"""
import csv
import sys
from collections import defaultdict

SHAPES = {  # grid threads of v3d::pairwise_distance_kernel<false, true> (256 per 128 x 128 tile) -> the launch it is
    1048576: "pairwise_distance B1 N8192 M8192 C256 (opbench): 64 x 64 tiles",
    4194304: "pairwise_distance B1 N16384 M16384 C64 (opbench): 128 x 128 tiles",
    524288: "pairwise_distance B32 N1024 M1024 C64 (BASELINE config 3; also inside knn_feat k20): 32 x 8 x 8 tiles",
    1601536: "knn_feat B1 N100000 M12500 C256 k1 (config 5, one-GPU shard): chunk 100000 x 1024 = 782 x 8 tiles",
    400384: "same call, last chunk 100000 x 212: 782 x 2 tiles",
    501760: "knn_feat B1 N12500 M100000 C256 k3: chunk 12500 x 2560 = 98 x 20 tiles",
    50176: "same call, last chunk 12500 x 160: 98 x 2 tiles",
}


def main():
    vals = defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1], newline="")):
        if "pairwise_distance_kernel" in r["kernel"]:
            vals[int(r["grid_threads"])][r["counter"]] = float(r["mean_value"])
    print("kernel v3d::pairwise_distance_kernel<false, true>  (fp32-input MFMA 32x32x2, 64 cycles each; 256 CUs x 4 SIMDs)")
    print(f"{'launch':96s} {'MFMA instr':>12s} {'MFMA busy cyc':>14s} {'kernel cyc':>11s} {'MFMA util':>9s} {'wave occupancy':>14s}")
    for grid, c in sorted(vals.items()):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        # SQ_WAVE_CYCLES counts quad-cycles per resident wave: x4 / (kernel cycles x 1024 SIMDs) = waves resident per SIMD
        occ = c["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc)
        print(f"{SHAPES.get(grid, str(grid)):96s} {c['SQ_INSTS_VALU_MFMA_F32']:12.0f} {c['SQ_VALU_MFMA_BUSY_CYCLES']:14.0f} "
              f"{cyc:11.0f} {util:9.3f} {occ:14.2f}")
    print("(counter passes serialise the launches and run a little slower than the timed runs; utilisation is a ratio of "
          "cycles, not of wall time)")


if __name__ == "__main__":
    main()
