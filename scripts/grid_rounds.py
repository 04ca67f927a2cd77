"""Resident-set accounting of every kernel in a rocprofv3 --kernel-trace CSV: how many workgroups one launch has, how many the
chip holds at once (256 CUs x the occupancy its registers / LDS / workgroup size allow) and the resulting number of rounds.  A
launch at 2.04 rounds runs three rounds, the last one on 4 % of the chip: the first thing to check before tuning a kernel body.
    python scripts/grid_rounds.py <dir with *_kernel_trace.csv> [min_total_us]"""
import csv, glob, os, sys
from collections import defaultdict

CUS, SIMDS, REGS, LDS_CU, WAVES_SIMD = 256, 4, 512, 160 * 1024, 8


def main():
    d = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    f = (glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) or [d])[0]
    agg = defaultdict(lambda: [0, 0.0, None])
    for r in csv.DictReader(open(f)):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        blocks = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // wg
        regs = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
        lds = int(r.get("LDS_Block_Size", 0) or 0)
        key = (r["Kernel_Name"][:90], blocks, wg, regs, lds)
        a = agg[key]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = []
    for (name, blocks, wg, regs, lds), (n, us, _) in agg.items():
        waves_wg = (wg + 63) // 64
        occ_regs = min(WAVES_SIMD, REGS // max(regs, 1)) if regs else WAVES_SIMD           # waves per SIMD
        by_regs = occ_regs * SIMDS // waves_wg
        by_lds = LDS_CU // lds if lds else 99
        per_cu = max(1, min(by_regs, by_lds, (WAVES_SIMD * SIMDS) // waves_wg))
        rounds = blocks / (per_cu * CUS)
        rows.append((us, n, name, blocks, wg, regs, lds, per_cu, rounds))
    rows.sort(reverse=True)
    print(f"{'total us':>9} {'calls':>5} {'us/call':>8} {'blocks':>7} {'wg':>4} {'regs':>4} {'lds':>6} {'wg/CU':>5} {'rounds':>7}  kernel")
    for us, n, name, blocks, wg, regs, lds, per_cu, rounds in rows:
        if us < min_us:
            continue
        frac = rounds - int(rounds)
        flag = " <-- tail round" if rounds > 1 and 0 < frac < 0.35 else ""
        print(f"{us:9.0f} {n:5d} {us / n:8.1f} {blocks:7d} {wg:4d} {regs:4d} {lds:6d} {per_cu:5d} {rounds:7.2f}  {name}{flag}")


if __name__ == "__main__":
    main()
