#!/usr/bin/env python
"""Join two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, guide MI355X_MICROARCH.md 'HBM') and a --kernel-trace
--stats pass of the same bench command into profiles/r04_pmc_traffic_<mode>.json: per kernel the HBM-side bytes per launch
(FETCH_SIZE x 2 on gfx950 for wide coalesced reads + WRITE_SIZE, both reported in KB) and the rate at the traced duration.
Each entry records the sha256 of the kernel's source file; bench.py reports `traffic` only while that still matches.

    python scripts/pmc_traffic.py <fetch_dir> <write_dir> <stats_csv> <out_json> [infer|x4|train|rt]
"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MB = 1e6
# bench.py key -> (substring of the kernel name, source file, algorithmic bytes per launch = every operand read once + every result
# written once at the mode's batch).  720p = 720 x 1280 LR, NHWC bf16 64-channel maps = 118 MB per image.
F64 = 720 * 1280 * 64 * 2            # one 64-channel bf16 map of a 720p image
KEYS = {
    "infer": {   # B = 8, 2x 720p -> 1080p
        # six blocks per launch.  Algorithmic bytes as SURVEY 8(d) counts them: the residual stream read + written per block; the launch's own
        # minimum is the stream in once + out once + the weights (0.19 GB + 5.3 MB); the kernel parks the stream once per block
        "stream_block": ("blocks_stream_kernel", "block_stream.hip", 6 * 1920 * 64 * 192 * 4 * 2),
        "conv64": ("conv_c64_persistent_kernel<4, 0, 3>", "conv3x3_c64.hip", 2 * 8 * F64),
        "tail": ("tail_stream_r2_kernel<true>", "tail_stream.hip", 8 * 3 * (720 * 1280 + 1440 * 2560 + 1080 * 1920) * 4),
        "patch_unembed": ("gemm_panel2_kernel<0, 4>", "gemm_tokens.hip", 1920 * 64 * 192 * 2 + 2 * 8 * F64),      # bf16 tokens (the streamed block kernel's out_bf16), skip in, map out
        "patch_embed": ("patch_embed_kernel<3, 2, 3>", "gemm_tokens.hip", 8 * F64 + 1920 * 64 * 192 * 4),
        "branch_a_5x5": ("bra_rows_persistent_kernel", "conv3x3_c64.hip", 8 * F64 + 8 * 3 * 1440 * 2560 * 4),
        "conv1": ("conv3x3_c3_persistent_kernel", "conv_thin.hip", 8 * 3 * 720 * 1280 * 4 + 8 * F64),
        "decoder_conv2": ("conv3_thin_rows_kernel", "conv3x3_c64.hip", 8 * F64 + 8 * 3 * 720 * 1280 * 4),
    },
    "x4": {      # B = 4, 4x 540p -> 2160p: the tail's last stage runs 1080p -> 2160p without a Resize
        "tail": ("tail_stream_r2_kernel<false>", "tail_stream.hip", 4 * 3 * (1080 * 1920 + 2 * 2160 * 3840) * 4),
        "stream_block": ("blocks_stream_kernel", "block_stream.hip", 6 * 540 * 64 * 192 * 4 * 2),        # 4 x 135 windows
        "branch_a_5x5": ("bra_rows_persistent_kernel", "conv3x3_c64.hip", 4 * 1080 * 1920 * 64 * 2 + 4 * 3 * 2160 * 3840 * 4),
    },
    "train": {   # B = 4, 2x 720p -> 1080p training step
        "window_attn_bwd": ("window_attn_bwd_kernel<12>", "attention_bwd.hip", 960 * 64 * (576 * 2 + 2 * 192 * 2 + 576 * 2) + 960 * 12 * 64 * 4),      # qkv, d att, att, lse in; d qkv out
        "conv64": ("conv_c64_persistent_kernel<4, 0, 3>", "conv3x3_c64.hip", 2 * 4 * F64),
        "conv64_wgrad": ("conv3x3_wgrad_c64_kernel", "conv_bwd.hip", 2 * 4 * F64),
        "conv_thin_wgrad": ("conv3x3_wgrad_thin_kernel", "conv_bwd.hip", 4 * F64 + 4 * 3 * 720 * 1280 * 4),                 # the 64-channel map + the 3-channel fp32 gradient
        "bra_wgrad": ("bra_wgrad_kernel", "branch_a_train.hip", 4 * F64 + 4 * 720 * 1280 * 16 * 2),                         # feat + g12 (16 bf16 per pixel)
        "patch_wgrad_wide": ("gemm_wgrad_wide_kernel<4, 2, true, 8>", "gemm_wgrad.hip", 4 * F64 + 960 * 64 * 192 * 2 + 192 * 4096 * 4),     # the map once, the bf16 token rows once, the fp32 output
        "feat_grad_combine": ("feat_grad_combine_kernel", "conv_bwd.hip", 5 * 4 * F64),          # only when H or W is not a multiple of 8
        "pe_bwd_merge": ("gemm_panel2_kernel<1, 6>", "gemm_tokens.hip", 960 * 64 * 192 * 4 + 4 * 4 * F64),      # tokens in; two adds, the gate map in, the merged gradient out
    },
    "rt": {      # B = 2, ResidualTransformer 6x training step
        "rt_attn_fwd": ("rt_attention_kernel<true>", "rt_kernels.hip", 2 * 3600 * (384 + 128) * 2 + 2 * 8 * 3600 * 4),
        "rt_attn_bwd_dq": ("rt_attn_bwd_dq_kernel<true>", "rt_kernels.hip", 2 * 3600 * (384 + 128 + 128) * 2),
        "rt_attn_bwd_dkv": ("rt_attn_bwd_dkv_kernel<true>", "rt_kernels.hip", 2 * 3600 * (384 + 128 + 256) * 2),
        "bicubic_bwd_rows": ("rt_bicubic_bwd_rows_band_kernel", "rt_kernels.hip", 2 * 3 * 4320 * 7680 * 4 * 2),
        "bicubic_sum": ("rt_bicubic_sum_sep_kernel", "rt_kernels.hip", 2 * 3 * 4320 * 7680 * 4),
        "l1_partial": ("l1_partial_kernel", "loss.hip", 2 * 2 * 3 * 4320 * 7680 * 4),
    },
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
    mode = sys.argv[5] if len(sys.argv) > 5 else "infer"
    fetch, write = counter_means(fetch_dir, "FETCH_SIZE"), counter_means(write_dir, "WRITE_SIZE")
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(stats_csv))}
    res = {"_note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 --mode {mode} "
                    "--no-cpu-baseline`; counters are in KB; reads x 2 (gfx950 tallies a 128-B request as 64 B), writes exact; "
                    "durations from the --kernel-trace --stats pass"}
    for key, (sub, src, alg) in KEYS[mode].items():
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
