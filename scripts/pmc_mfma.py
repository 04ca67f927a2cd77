#!/usr/bin/env python
"""Counter-based MFMA evidence for the whole-block kernel (SURVEY 8(d) "rocprof MFMA utilisation").

Joins rocprofv3 --pmc passes (each collected in its own run with --kernel-trace only) of
`bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline` into profiles/r04_pmc_mfma.json:

    python scripts/pmc_mfma.py <out_json> <stats_csv> <pmc_dir> [<pmc_dir> ...] [--mode infer|train|rt]

Per kernel of interest: the raw counter means per launch, and derived from them
  * mfma_util            = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)  -- rocprof's own `MfmaUtil` expression
                           (GRBM_GUI_ACTIVE is summed over the 8 XCDs; the matrix-pipe busy cycles over every SIMD of the chip)
  * clock_GHz            = GRBM_GUI_ACTIVE / 8 / duration  (guide: reads high on dispatches shorter than ~0.3 ms)
  * mfma_cycles_expected = SQ_INSTS_VALU_MFMA_MOPS_BF16-derived or instruction-count * 16 (v_mfma_f32_16x16x32_bf16 = 16 cycles)
  * wave-cycle split     = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES, VALU share, co-execution
"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS_BY_MODE = {
    "infer": {
        "stream_block": ("blocks_stream_kernel", "block_stream.hip"),
        "fused_block": ("fused_qkv_attn_kernel<true, true", "fused_attn.hip"),
        "conv64": ("conv_c64_persistent_kernel<4, 0, 3>", "conv3x3_c64.hip"),
        "branch_a_5x5": ("bra_rows_persistent_kernel", "conv3x3_c64.hip"),
        "patch_embed": ("patch_embed_kernel<3, 2, 3>", "gemm_tokens.hip"),
        "patch_unembed": ("gemm_panel2_kernel<0, 4>", "gemm_tokens.hip"),          # bf16 tokens from the streamed block kernel
    },
    "train": {   # FastTransformer training step (BASELINE configs[2] per rank)
        "conv64": ("conv_c64_persistent_kernel<4, 0, 3>", "conv3x3_c64.hip"),
        "conv64_wgrad": ("conv3x3_wgrad_c64_kernel", "conv_bwd.hip"),
        "gemm_wgrad": ("gemm_wgrad_kernel<0, 0>", "gemm_wgrad.hip"),
        "bra_wgrad": ("bra_wgrad_kernel", "branch_a_train.hip"),
        "conv_thin_wgrad": ("conv3x3_wgrad_thin_kernel", "conv_bwd.hip"),
        "patch_wgrad_wide": ("gemm_wgrad_wide_kernel<4, 2, true, 8>", "gemm_wgrad.hip"),
        "window_attn_bwd": ("window_attn_bwd_kernel<12>", "attention_bwd.hip"),
        "window_attn_fwd": ("window_attn_kernel<12", "attention.hip"),
        "fused_mlp": ("fused_mlp_v2_kernel", "fused_blocks.hip"),
    },
    "rt": {      # ResidualTransformer 6x training step (configs[4] per rank)
        "rt_attn_fwd": ("rt_attention_kernel<true>", "rt_kernels.hip"),
        "rt_attn_bwd_dq": ("rt_attn_bwd_dq_kernel<true>", "rt_kernels.hip"),
        "rt_attn_bwd_dkv": ("rt_attn_bwd_dkv_kernel<true>", "rt_kernels.hip"),
    },
}
SIMDS = 256 * 4


def counters(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    argv = list(sys.argv[1:])
    mode = "infer"
    if "--mode" in argv:
        i = argv.index("--mode"); mode = argv[i + 1]; del argv[i:i + 2]
    KERNELS = KERNELS_BY_MODE[mode]
    out, stats_csv, dirs = argv[0], argv[1], argv[2:]
    acc = counters(dirs)
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(stats_csv))}
    res = {"_note": "rocprofv3 --pmc <counters> --kernel-trace (each pass its own run) of `bench.py --steps 3 --warmup 1 --mode " + mode + " "
                    "--no-cpu-baseline`; values are means per launch; durations from the --kernel-trace --stats pass of the same command. "
                    "Profiled passes run at a lower clock than untraced ones (guide, DVFS item 2): read ratios, not absolute times."}
    for key, (sub, src) in KERNELS.items():
        names = [n for n in acc if sub in n]
        if not names:
            continue
        n = names[0]
        c = {k: sum(v) / len(v) for k, v in acc[n].items()}
        ns = next((v for k, v in dur.items() if sub in k), None)
        ent = {"kernel": n[:160], "launches_sampled": max(len(v) for v in acc[n].values()), "avg_ns_stats_pass": ns, "counters": c}
        gui = c.get("GRBM_GUI_ACTIVE")
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        d = {}
        if gui and busy is not None:
            d["mfma_util"] = busy / (gui / 8 * SIMDS)
        if gui and ns:
            d["clock_GHz_from_GRBM_GUI_ACTIVE"] = gui / 8 / ns
        if busy is not None and ns:
            d["mfma_busy_cycles_per_simd"] = busy / SIMDS
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if k in c:
                    d[k.lower() + "_share_of_wave_cycles"] = c[k] / wc
        if busy and "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
            d["valu_mfma_coexec_share_of_mfma_busy"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / busy
        if "SQ_INSTS_MFMA" in c:
            d["mfma_instructions"] = c["SQ_INSTS_MFMA"]
            if busy:
                d["busy_cycles_per_mfma_instruction"] = busy / c["SQ_INSTS_MFMA"]
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            d["valu_instructions_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c and ns:
            # one MOPS unit = 512 FLOP (rocprof's convention for the MFMA_MOPS counters)
            d["mfma_tflops_executed_bf16"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / ns / 1e3
        ent["derived"] = d
        with open(os.path.join(ROOT, "transformerupscaler_amd", "csrc", src), "rb") as f:
            ent["source"], ent["source_sha256"] = src, hashlib.sha256(f.read()).hexdigest()
        res[key] = ent
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k != "_note":
            print(k, json.dumps(v["derived"]))


if __name__ == "__main__":
    main()
