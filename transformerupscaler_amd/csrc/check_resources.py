#!/usr/bin/env python3
"""Build-time guard for the kernels whose LDS-DMA hand-off is ordered by hand-counted `s_waitcnt vmcnt(N)`.

Scratch (register spill) loads and stores count on the same `vmcnt` as the LDS-DMA pieces those waits count, so a build of
such a kernel with scratch > 0 can pass a barrier before its weights have landed: wrong results at full size, masked at small
sizes (DESIGN 5b records exactly that for the whole-block kernel).  The Makefile compiles every object with
`-Rpass-analysis=kernel-resource-usage`, keeps the remarks in build/<file>.remarks and runs this script before linking:

    check_resources.py build/*.remarks  ->  build/resource_usage.json, exit 1 if a guarded kernel reports scratch

Diagnostic instantiations (the `STAMPS = true` builds, which spill by design and are never launched by the product path)
are exempt.  The JSON also records the compiler version, which bench.py copies into its line.
"""
import json
import os
import re
import subprocess
import sys

# (source file, kernel function names) -> every instantiation must have ScratchSize == 0
GUARDED = [
    ("fused_attn.hip", ["fused_qkv_attn_kernel"]),
    ("block_stream.hip", ["blocks_stream_kernel"]),
    ("fused_blocks.hip", ["fused_mlp_v2_kernel"]),
    ("conv3x3_c64.hip", ["conv_c64_persistent_kernel", "bra_rows_persistent_kernel", "conv3_thin_rows_kernel"]),
    ("conv_thin.hip", ["conv3x3_c3_persistent_kernel"]),
    ("gemm_tokens.hip", ["gemm_panel2_kernel", "patch_embed_kernel"]),
    # no counted hand-off here: guarded because the unrolled halo-row loops sit at 240 / 202 registers and a build that hoists their
    # read addresses out of the tile loop spills (seen twice while they were written)
    ("conv_bwd.hip", ["conv3x3_wgrad_c64_kernel", "conv3x3_wgrad_thin_kernel"]),
    ("branch_a_train.hip", ["bra_wgrad_kernel"]),
    # counted vmcnt between its LDS-DMA stages, asm fragment reads with counted lgkmcnt
    ("gemm_wgrad.hip", ["gemm_wgrad_wide_kernel"]),
]
# diagnostic template instantiations, never launched by the product path: fused_qkv_attn_kernel<PROJ, MLP, STAMPS = true>, the
# timing ablations fused_mlp_v2_kernel<ABL != 0>
EXEMPT = re.compile(r"fused_qkv_attn_kernel<true, true, true>|fused_qkv_attn_kernelILb1ELb1ELb1E|fused_mlp_v2_kernelILi[1-9]|blocks_stream_kernel<true>|blocks_stream_kernelILb1E")
# Known spills that sit OUTSIDE the counted hand-off (checked in the ISA): allowed up to the recorded size, so growth still fails.
#  * conv_c64_persistent_kernel<4,0,3>: two per-lane source pointers of the first tile's prefetch, spilled at kernel entry and
#    reloaded once per cout pass BEFORE that pass's first DMA -- older than every DMA piece a counted wait covers;
#  * fused_mlp_v2_kernel<0> (the stand-alone MLP half, not on the default inference path): LayerNorm staging values spilled
#    before the chunk loop.
# Extra vmcnt-counted operations YOUNGER than a DMA piece make a counted wait stricter (slower), never weaker; the failure mode the
# guard exists for is a build whose spill reloads land inside the head / chunk loops, which these two caps would catch as growth.
ALLOWED_BYTES = {"fused_mlp_v2_kernelILi0E": 72}

FIELDS = {
    "sgprs": r"TotalSGPRs: (\d+)", "vgprs": r" VGPRs: (\d+)", "agprs": r"AGPRs: (\d+)",
    "scratch_bytes_per_lane": r"ScratchSize \[bytes/lane\]: (\d+)", "waves_per_simd": r"Occupancy \[waves/SIMD\]: (\d+)",
    "sgpr_spill": r"SGPRs Spill: (\d+)", "vgpr_spill": r"VGPRs Spill: (\d+)",
}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:                       # noqa: BLE001 -- no demangler: patterns below are tried on the mangled name too
        return {n: n for n in names}


def parse(path):
    txt = open(path, errors="replace").read()
    parts = re.split(r"remark: Function Name: (\S+)", txt)
    recs = {}
    for name, body in zip(parts[1::2], parts[2::2]):
        rec = {}
        for k, pat in FIELDS.items():
            m = re.search(pat, body)
            if m:
                rec[k] = int(m.group(1))
        recs[name] = rec
    return recs


def main(argv):
    out_path = None
    files = []
    for a in argv:
        if a.startswith("--out="):
            out_path = a[6:]
        else:
            files.append(a)
    report, bad = {}, []
    for f in sorted(files):
        src = os.path.basename(f).replace(".remarks", ".hip")
        recs = parse(f)
        names = demangle(list(recs))
        for mangled, rec in recs.items():
            nice = names.get(mangled, mangled)
            rec["source"] = src
            guarded = any(src == gsrc and any(fn in mangled for fn in fns) for gsrc, fns in GUARDED) \
                and not (EXEMPT.search(nice) or EXEMPT.search(mangled))
            rec["guarded_no_scratch"] = bool(guarded)
            report[nice] = rec
            cap = max([v for k, v in ALLOWED_BYTES.items() if k in mangled] + [0])
            rec["scratch_cap_bytes_per_lane"] = cap
            if guarded and rec.get("scratch_bytes_per_lane", 0) > cap:
                bad.append((src, nice, rec["scratch_bytes_per_lane"]))
    try:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        m = re.search(r"HIP version: (\S+)", ver)
        c = re.search(r"clang version (\S+)", ver)
        compiler = {"hip_version": m.group(1) if m else None, "clang_version": c.group(1) if c else None}
    except Exception:                       # noqa: BLE001
        compiler = {"hip_version": None, "clang_version": None}
    if out_path:
        with open(out_path, "w") as fh:
            json.dump({"compiler": compiler, "kernels": report}, fh, indent=1, sort_keys=True)
    if bad:
        for src, nice, sc in bad:
            sys.stderr.write(f"check_resources: {src}: {nice} uses {sc} B/lane of scratch; its hand-counted vmcnt waits "
                             "would be wrong (scratch traffic counts on vmcnt). Reduce register pressure; do not ship this build.\n")
        return 1
    n_guard = sum(1 for r in report.values() if r["guarded_no_scratch"])
    print(f"check_resources: {len(report)} kernels, {n_guard} guarded (scratch 0, or within a recorded cap), compiler {compiler}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
