"""EXPERIMENT: issue cost of vector instructions on one SIMD, alone and beside MFMAs (csrc/experiments/valu_rate_exp.hip)."""
import ctypes, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "transformerupscaler_amd", "libtupscale_valu_rate_exp.so"))
P = ctypes.c_void_p
lib.tup_exp_valu_rate.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
names = ["v_fma_f32", "v_pk_fma_f16", "v_fma_f16", "v_pk_mul_f16", "v_cvt_pk_f16_f32", "v_exp_f32", "v_pk_max_f16", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_mul_f32", "s_nop 0", "s_add_u32", "ds_read_b128", "v_mov_b32", "v_fmac_f32", "v_pk_fmac_f16", "v_add_f32", "s_waitcnt (no wait)", "v_pk_fma_f32", "v_pk_mul_f32", "v_fmac_f32 (sgpr src)"]
out = torch.zeros(2048, dtype=torch.int64, device="cuda")
reps = 256
def run(kind, nf8, mf, waves):
    out.zero_()
    for _ in range(2):
        e = lib.tup_exp_valu_rate(out.data_ptr(), kind, nf8, mf, waves, reps, torch.cuda.current_stream().cuda_stream)
        assert e == 0, e
        torch.cuda.synchronize()
    return out[:waves].cpu().tolist()
base1 = run(0, 0, 1, 4); base2 = run(0, 0, 1, 8)
print(f"bare MFMAs: one wave/SIMD {max(base1) / (4 * reps):.1f} cycles per MFMA; two waves/SIMD {max(base2) / (4 * reps):.1f} per MFMA of each wave")
for k, n in enumerate(names):
    a1 = run(k, 1, 0, 4); a2 = run(k, 1, 0, 8)
    m1 = run(k, 1, 1, 4); m2 = run(k, 1, 1, 8); m1b = run(k, 2, 1, 4); m2b = run(k, 2, 1, 8)
    few = "  ".join(f"{c}: {max(run(k, 100 + c, 1, 4)) / (4 * reps):5.1f} / {max(run(k, 100 + c, 1, 8)) / (4 * reps):5.1f}" for c in (2, 4, 6))
    print(f"{n:20s} MFMA + c fillers, gap of one wave (1 wave/SIMD / 2 waves/SIMD): {few}")
    print(f"{n:20s} alone: {max(a1) / (32 * reps):5.2f} cyc/instr (1 wave/SIMD) {max(a2) / (32 * reps):5.2f} (2 waves, each)   "
          f"| MFMA + 8 fillers: {max(m1) / (4 * reps):6.1f} / gap (1 wave) {max(m2) / (4 * reps):6.1f} (2 waves)   | MFMA + 16: {max(m1b) / (4 * reps):6.1f} {max(m2b) / (4 * reps):6.1f}")
