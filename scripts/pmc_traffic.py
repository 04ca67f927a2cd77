#!/usr/bin/env python
"""Join two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, guide MI355X_MICROARCH.md 'HBM') and a --kernel-trace
--stats pass of the same bench command into profiles/r02_pmc_traffic.json: per kernel the HBM-side bytes per launch
(FETCH_SIZE x 2 on gfx950 for wide coalesced reads + WRITE_SIZE, both reported in KB) and the rate at the traced duration.
Each entry records the sha256 of the kernel's source file; bench.py reports `traffic` only while that still matches.

    python scripts/pmc_traffic.py <fetch_dir> <write_dir> <stats_csv> <out_json>
"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {  # bench.py key -> (substring of the kernel name, source file, algorithmic bytes per launch at B = 8 or None)
    "fused_block": ("fused_qkv_attn_kernel<true, true", "fused_attn.hip", 6 * 1920 * 64 * 192 * 4 * 2),   # six blocks per launch
    "conv64": ("conv_c64_persistent_kernel<4, 0, 3>", "conv3x3_c64.hip", 2 * 8 * 720 * 1280 * 64 * 2),
    "tail": ("tail_fused_kernel", "tail_fused.hip", 8 * 3 * (720 * 1280 + 1440 * 2560 + 1080 * 1920) * 4),
    "patch_unembed": ("gemm_panel2_kernel<1, 4>", "gemm_tokens.hip", None),
    "patch_embed": ("patch_embed_kernel<3, 2, 3>", "gemm_tokens.hip", None),
    "branch_a_5x5": ("conv_c64_persistent_kernel<1, 1, 5>", "conv3x3_c64.hip", None),
    "conv1": ("conv3x3_c3_persistent_kernel", "conv_thin.hip", None),
    "decoder_conv2": ("conv_c64_persistent_kernel<1, 1, 3>", "conv3x3_c64.hip", None),
}


def counter_means(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, stats_csv, out = sys.argv[1:5]
    fetch, write = counter_means(fetch_dir, "FETCH_SIZE"), counter_means(write_dir, "WRITE_SIZE")
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(stats_csv))}
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 --mode infer "
                    "--no-cpu-baseline`; counters are in KB; reads x 2 (gfx950 tallies a 128-B request as 64 B), writes exact; "
                    "durations from the --kernel-trace --stats pass"}
    for key, (sub, src, alg) in KEYS.items():
        names = [n for n in fetch if sub in n]
        if not names:
            continue
        n = names[0]
        rd, wr = fetch[n][0] * 1024 * 2, write.get(n, (0.0, 0))[0] * 1024
        ns = next((v for k, v in dur.items() if sub in k), None)
        with open(os.path.join(ROOT, "transformerupscaler_amd", "csrc", src), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        res[key] = {"kernel": n[:160], "launches_sampled": fetch[n][1], "read_bytes_x2_corrected": rd, "write_bytes": wr,
                    "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg, "avg_ns": ns,
                    "TBps": (rd + wr) / ns / 1e3 if ns else None, "source": src, "source_sha256": sha}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k != "_note":
            print(k, f"{v['traffic_bytes_per_launch'] / 1e6:.1f} MB", v["TBps"])


if __name__ == "__main__":
    main()
