// The six WindowTransformerBlocks of a forward (reference models/FastTransformer/model.py:153-172, the loop at :288-289) as ONE
// launch built on v_mfma_f32_32x32x16_{bf16,f16}, with every vector instruction of LayerNorm / softmax / GELU placed BETWEEN the
// matrix instructions of the same wave (tup_blocks_stream_fwd; round 4's replacement of fused_attn.hip's whole-block kernel).
//
// Why this shape (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): an MFMA holds the SIMD's vector issue for 8 cycles whatever
// its size, so a 32x32x16 MFMA (32 cycles of matrix pipe) leaves 24 cycles = 5-6 vector instructions of the SAME wave that cost
// nothing, where a 16x16x32 MFMA (16 cycles) leaves 2.  The block needs ~3.4 vector instructions per 16 cycles of matrix work:
// on 16x16x32 tiles that cannot hide, on 32x32x16 tiles almost all of it can -- if the instruction stream offers the fillers right
// behind each MFMA.  fused_attn.hip runs its phases back to back (an MFMA phase, then a softmax or GELU phase) and leaves the overlap to
// the other wave of the SIMD; its own ablations cap that structure at 0.34 of the MFMA peak.  Here the stream is
// software-pipelined by hand: the softmax of head h rides between the qkv MFMAs of head h + 1, the GELU of hidden chunk c - 1 between
// the fc1 MFMAs of chunk c and the fc2 MFMAs of chunk c - 2, and the order is pinned with sched_barrier fences.
//
// Workgroup = 8 waves = 4 windows (two waves per SIMD, 256 registers each); wave w owns the 32 tokens of half hf = w & 1 of window
// w >> 1: token r = l & 31 on the lane, lane half h = l >> 5.  Layouts (packing.pack_stream_block states them from the other side):
//   * accumulator tile [32 rows][32 tokens]: register i of lane (r, h) = row rho(i, h) = (i & 3) + 8 (i >> 2) + 4 h, token r;
//   * residual stream R[6]: tile rt row rho = channel 32 rt + 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3): a lane's registers of a
//     tile are 16 consecutive channels; registers 8u .. 8u+7 pack into the B fragment of K-step 2 rt + u of the product that follows;
//   * q | k tile of a head: rows 0-15 = q (pre-scaled by 1/4), rows 16-31 = k: registers 0-7 pack into the B fragment of S^T = K Q^T,
//     registers 8-15 into the A fragment of the wave's own 32 keys; the other 32 keys' fragment crosses LDS as it is (1 KB);
//   * v of a head pair comes out of the product with the operands swapped (A = tokens, B = weights): [token rows][32 channels on the
//     lanes], i.e. already the A operand of O^T = V^T P^T; the 16 lanes of the pair's other head are set to 1.0, so the same
//     MFMAs leave the softmax row sums in the other 16 rows (no reduction instructions);
//   * O^T's registers pack into the B fragment of K-step "head" of the proj; the hidden tile's into the K-steps of mlp.2.
// LDS (one workgroup per CU): tables | region A: q|k ring [2] + v ring [2] (12 KB tiles, byte-exact images, linear DMA) | region B:
// K / V fragment exchange | the attention outputs of heads 0-7 (8 KB per wave; heads 8-11 stay in registers).  The proj weight
// then takes regions A + B, the MLP weight ring (6 x 24 KB: mlp.0 chunk + mlp.2 chunk of 32 hidden units) all of it.
// The residual stream leaves the registers once per block (stored after LayerNorm1, re-read into the proj accumulators).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BS_NT = 512;
constexpr int TILE = 12288;                               // [3 k-tiles][32 rows][128 B]
constexpr int L_TAB = 0;                                  // 1536 floats
constexpr int L_A = 8192, L_WQK = L_A, L_WV = L_A + 2 * TILE;
constexpr int L_B = L_A + 4 * TILE;                       // 57344
constexpr int L_KX = L_B, L_VX = L_B + 16384;             // [2 parities][8 waves][1 KB] | [8 waves][2 KB]
constexpr int L_OF = L_B + 32768;                         // 90112: [8 waves][8 heads][1 KB]
constexpr int BS_LDS = L_OF + 65536;                      // 155648
constexpr int CHUNK = 2 * TILE;                           // MLP ring slot: mlp.0 tile + mlp.2 tile
static_assert(BS_LDS <= 163840, "LDS");
static_assert(L_A + 6 * CHUNK <= BS_LDS && L_A + 3 * CHUNK >= L_B + 2 * TILE, "MLP ring: six slots; slots 3-5 (the first chunks) clear of the proj tiles");
// table offsets (floats)
constexpr int T_QKB = 0, T_B1 = 384, T_BP = 1152, T_B2 = 1344;
// LDS offsets of the six proj weight tiles (see the slots that request them)
constexpr int PROJ_T[6] = {L_WV, L_WQK, L_WV + TILE, L_WQK + TILE, L_B, L_B + TILE};

struct StreamBlock { const char* wqk; const char* wv; const char* wproj; const char* w1; const char* w2; const float* tab; const float* sbias; };
// one slot of the MLP half: fc1 of chunk c (F1) | GELU of chunk c - 1 (GE) | fc2 of chunk c - 2 (F2); DM: the slot also requests weights
template <int PAR_, bool F1_, bool GE_, bool F2_, bool DM_> struct MlpSlot {
    static constexpr int PAR = PAR_, NM = (F1_ && F2_) ? 24 : 12;
    static constexpr bool F1 = F1_, GE = GE_, F2 = F2_, DM = DM_;
    static constexpr bool LNS = !F1_ && !GE_ && F2_;       // the last slot: tile t's accumulators are final behind MFMA 2 t + 1
};
constexpr int BS_MAX_BLK = 8;
struct StreamTable { StreamBlock b[BS_MAX_BLK]; };

template <class F, int... I> TUP_DEVICE void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> TUP_DEVICE void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

TUP_DEVICE f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
TUP_DEVICE f32x16 mfma32h(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

TUP_DEVICE void lds_write_b128(uint32_t addr, bf16x8 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
TUP_DEVICE void lds_write_b128_off(uint32_t addr, bf16x8 v, int off) { asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(off) : "memory"); }
TUP_DEVICE void lds_write_b32(uint32_t addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
TUP_DEVICE uint32_t lds_read_b32_off(uint32_t addr, int off) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off));
    return v;
}
TUP_DEVICE f32x4 lds_read_f4_off(uint32_t addr, int off) { return __builtin_bit_cast(f32x4, lds_read_b128_asm_off(addr, off)); }
template <int N> TUP_DEVICE void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
TUP_DEVICE void barrier_all() {          // every LDS write of this wave is done, every DMA piece it issued has landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

TUP_DEVICE __amdgpu_buffer_rsrc_t bs_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
// one 1 KB piece: LDS dst (wave-uniform) + lane * 16  <-  src + soff + lane * 16
TUP_DEVICE void bs_dma(__amdgpu_buffer_rsrc_t r, char* lds_dst, uint32_t lane16, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)lane16, soff, 0, 0);
}

// The same piece inside an MFMA stream: requested only by the waves whose LDS base has bit 31 clear, with the branch INSIDE the asm
// statement -- a C++ `if` around the builtin splits the stream into basic blocks, across which hipcc moves the GELU arithmetic out of
// its gaps unless every value is pinned (and every pin next to a packed instruction costs an s_nop) -- and with the piece's offset in
// the instruction's immediate field, which moves the LDS address and the source address alike: one LDS base and one source offset
// per group of up to four pieces instead of four scalar additions per piece.  (hipcc does not see the request: nothing but the
// hand-counted barrier_all() waits for it.  The predicate rides in the LDS base because a separate scalar operand that lives
// across the block loop came out of register allocation as a VGPR.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int OFF> TUP_DEVICE void bs_dma_on(__amdgpu_buffer_rsrc_t rs, uint32_t lds_base, uint32_t lane16, uint32_t soff) {
    static_assert(OFF >= 0 && OFF < 4096, "12-bit immediate offset");
    // (s_nop: a scalar write of M0 needs one wait state before an LDS-DMA instruction reads it -- hipcc pads its own the same way)
    asm volatile("s_bitcmp1_b32 %0, 31\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:%4 lds\n1:"
                 :: "s"(lds_base), "v"(lane16), "s"(rs), "s"(soff), "n"(OFF) : "memory", "scc", "m0");
}
#pragma clang diagnostic pop

TUP_DEVICE bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    return __builtin_bit_cast(bf16x8, u32x4{pack_bf16x2(a0, a1), pack_bf16x2(a2, a3), pack_bf16x2(a4, a5), pack_bf16x2(a6, a7)});
}
template <int O> TUP_DEVICE bf16x8 pack8_regs(const f32x16& v) { return pack8(v[O], v[O + 1], v[O + 2], v[O + 3], v[O + 4], v[O + 5], v[O + 6], v[O + 7]); }
TUP_DEVICE float half_sum(float v) {       // v(lane l) + v(lane l ^ 32)
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, opaque_copy(q[1]));
}
TUP_DEVICE float half_max(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return vmax(__builtin_bit_cast(float, q[0]), __builtin_bit_cast(float, opaque_copy(q[1])));
}
TUP_DEVICE float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }      // v_max3_f32

// LayerNorm (scale / shift folded into the Linear that follows) of the wave's 32 tokens: residual tiles -> the 12 B fragments.
// sum and sum of squares in one sweep (four partial chains), the two lane halves of a token joined by permlane32 swaps.
struct LnStats { float s[2], q[2]; };
TUP_DEVICE void ln_stats_tile(LnStats& st, const f32x16& t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { st.s[i & 1] += t[i]; st.q[i & 1] = __builtin_fmaf(t[i], t[i], st.q[i & 1]); }
}
// the same as micro-operations (op 2 i: sum, op 2 i + 1: sum of squares of register i), for the gaps of the proj MFMAs; every result is
// pinned (an empty asm) so that neither the SLP vectoriser pairs them up nor the optimiser moves them out of their gap
template <int K> TUP_DEVICE void ln_stats_op(LnStats& st, const f32x16& t) {
    constexpr int i = K >> 1;
    if constexpr ((K & 1) == 0) { st.s[i & 1] += t[i]; asm volatile("" : "+v"(st.s[i & 1])); }
    else { st.q[i & 1] = __builtin_fmaf(t[i], t[i], st.q[i & 1]); asm volatile("" : "+v"(st.q[i & 1])); }
}
template <int K0, int K1> TUP_DEVICE void ln_stats_ops(LnStats& st, const f32x16& t) {
    static_for<(K1 > K0 ? K1 - K0 : 0)>([&](auto k) { ln_stats_op<K0 + decltype(k)::value>(st, t); });
}
TUP_DEVICE void ln_finish(const LnStats& st, float& rstd, float& shift) {
    const float sum = half_sum(st.s[0] + st.s[1]), sq = half_sum(st.q[0] + st.q[1]);
    const float mean = sum * (1.0f / 192);
    rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, sq * (1.0f / 192)), 0.f) + 1e-5f);
    shift = -mean * rstd;
}
template <int U> TUP_DEVICE bf16x8 ln_frag(const f32x16& t, float rstd, float shift) {
    constexpr int O = 8 * U;
    return pack8(__builtin_fmaf(t[O], rstd, shift), __builtin_fmaf(t[O + 1], rstd, shift), __builtin_fmaf(t[O + 2], rstd, shift),
                 __builtin_fmaf(t[O + 3], rstd, shift), __builtin_fmaf(t[O + 4], rstd, shift), __builtin_fmaf(t[O + 5], rstd, shift),
                 __builtin_fmaf(t[O + 6], rstd, shift), __builtin_fmaf(t[O + 7], rstd, shift));
}

// ---- softmax of one head as a list of micro-operations (so that the caller can hand them out between MFMAs) ----
// S[0] / S[1]: the 32 own-key / 32 partner-key scores of query r in this lane half (the other half of the keys sits in lane l ^ 32).
// The scores arrive in log2 units (q rows and the relative position bias carry log2(e), packing.pack_stream_block).
// ops 0..15: four max3 chains over 8 values each; 16, 17: join; 18: other lane half; 19: the B fragment of "- max" (bf16: the
// shift of a softmax is free to be any number near the maximum) -- the subtraction itself is one MFMA per score tile, a K-step
// whose A operand is a column of ones (32 vector instructions per head and lane otherwise); 20 + k (k < 32): e = exp2(s) in place;
// 52 + f (f < 4): the four fragments of P (bf16).
constexpr int SM_MAX_OPS = 20, SM_OPS = 56;
struct SmState { float c[4]; bf16x8 nmB; };
template <int K> TUP_DEVICE void softmax_op(SmState& st, f32x16 (&S)[2], bf16x8 (&P)[4], bool h0) {
    if constexpr (K < 16) {
        constexpr int ch = K & 3, step = K >> 2;          // chain ch covers S[ch >> 1] registers 8 (ch & 1) .. +7
        constexpr int t = ch >> 1, o = 8 * (ch & 1);
        if constexpr (step == 0) st.c[ch] = max3(S[t][o], S[t][o + 1], S[t][o + 2]);
        else if constexpr (step == 1) st.c[ch] = max3(st.c[ch], S[t][o + 3], S[t][o + 4]);
        else if constexpr (step == 2) st.c[ch] = max3(st.c[ch], S[t][o + 5], S[t][o + 6]);
        else st.c[ch] = vmax(st.c[ch], S[t][o + 7]);
    } else if constexpr (K == 16) st.c[0] = max3(st.c[0], st.c[1], st.c[2]);
    else if constexpr (K == 17) st.c[0] = vmax(st.c[0], st.c[3]);
    else if constexpr (K == 18) st.c[0] = half_max(st.c[0]);
    else if constexpr (K == 19) {
        // K column 0 (lane half 0, element 0) of the B fragment = - max of this lane's query column
        const uint32_t nm = pack_bf16x2(-st.c[0], 0.f);
        st.nmB = __builtin_bit_cast(bf16x8, u32x4{h0 ? nm : 0u, 0u, 0u, 0u});
    } else if constexpr (K < 52) {
        constexpr int k = K - 20, t = k >> 4, i = k & 15;
        S[t][i] = __builtin_amdgcn_exp2f(S[t][i]);
    } else {
        constexpr int f = K - 52;
        if constexpr ((f & 1) == 0) P[f] = pack8_regs<0>(S[f >> 1]); else P[f] = pack8_regs<8>(S[f >> 1]);
    }
}
template <int K0, int K1> TUP_DEVICE void softmax_ops(SmState& st, f32x16 (&S)[2], bf16x8 (&P)[4], bool h0) {
    static_for<(K1 > K0 ? K1 - K0 : 0)>([&](auto k) { softmax_op<K0 + decltype(k)::value>(st, S, P, h0); });
}

// ---- GELU of a hidden chunk (16 accumulator values per lane = 8 fp16 pairs) as 64 micro-operations: two batches of four chains ----
// u = x / 4 in (the 1/4 rides in the packed mlp.0 weight and bias), gelu(x) / 4 = u Phi out (mlp.2 carries the 4), with
//     Phi(x) ~ clamp01(1/2 + u q(u^2 - 1/2)),  q of degree 4,
// in packed fp16.  The clamp is the clamp bit of the last fma and there is none on the input: q is fitted (minimax on the GELU's
// own error over |x| <= 12, scripts/fit_gelu.py) with a positive leading coefficient, so that beyond |x| ~ 3.6 the argument runs
// monotonically out of [0, 1] (overflow to +-inf included) and the clamp returns the exact 0 or 1.  fp16 evaluation: max |error|
// 3.9e-3, rms 8.0e-4 under N(0, 1.5) (rounds 3-4's degree 5 with an explicit max / min on the input: 4.2e-3 / 7.3e-4, 11 instead
// of 8 instructions per pair of values; the bf16 hidden tile of rounds 1-2 had 1.6e-2 / 2.4e-3).  Every vector instruction of this
// kernel is charged on the SIMD's vector port: the 88 instructions of the previous form cost 8-11 % of the launch
// although four of them sat behind every MFMA.  op 32 b + k: k < 4 convert pair k; then step-major: k = 4 + 4 t + i, step t < 7 of
// chain i (u^2 - 1/2; the Horner start; three Horner steps; Phi; u Phi).
// Pinning: hipcc sinks a chain nothing in its slot consumes out of the MFMA stream (a GELU phase of its own in the loop latch;
// sched_barrier binds only the machine scheduler), so a step's results pass through an empty volatile asm behind the NEXT MFMA.
constexpr int GELU_OPS = 64;                  // = its instructions: 2 x (4 conversions + 7 steps x 4 packed pairs)
struct GeluState { h2 x[4], sv[4], q[4]; };
// One value of a step's four is pinned: with the slot one basic block that holds the whole step in its gap, and the pinned value's
// consumer is the LAST operation of the next step -- three instructions away from the asm, so no hazard s_nop is owed.
#define TUP_PIN4(A) asm volatile("" : "+v"(A[3]))
template <int K> TUP_DEVICE void gelu_op(GeluState& g, const f32x16& acc, bf16x8 (&hf)[2]) {
    constexpr int b = K / 32, k = K % 32;
    if constexpr (k < 4) {
        g.x[k] = __builtin_convertvector(f32x2{acc[8 * b + 2 * k], acc[8 * b + 2 * k + 1]}, h2);
    } else {
        constexpr int t = (k - 4) >> 2, i = (k - 4) & 3;
        constexpr float C[3] = {0.82089258f, -0.67178146f, 0.70430058f};
        const h2 zero = {(_Float16)0.0f, (_Float16)0.0f}, one = {(_Float16)1.0f, (_Float16)1.0f};
        if constexpr (t == 0) g.sv[i] = __builtin_elementwise_fma(g.x[i], g.x[i], h2{(_Float16)-0.5f, (_Float16)-0.5f});
        else if constexpr (t == 1) g.q[i] = __builtin_elementwise_fma(g.sv[i], h2{(_Float16)2.74597983f, (_Float16)2.74597983f}, h2{(_Float16)-1.28256022f, (_Float16)-1.28256022f});
        else if constexpr (t < 5) g.q[i] = __builtin_elementwise_fma(g.q[i], g.sv[i], h2{(_Float16)C[t - 2], (_Float16)C[t - 2]});
        // Phi = the fma's result clamped to [0, 1]: the instruction's clamp bit
        else if constexpr (t == 5) g.q[i] = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_elementwise_fma(g.x[i], g.q[i], h2{(_Float16)0.5f, (_Float16)0.5f}), zero), one);
        else g.x[i] = g.x[i] * g.q[i];
    }
}
// the pins of the steps that operation K completes (issued by the caller behind the gap's MFMA: a pin directly behind a packed
// instruction whose result it names costs an s_nop as well)
template <int K> TUP_DEVICE void gelu_pin(GeluState& g, bf16x8 (&hf)[2]) {
    constexpr int b = K / 32, k = K % 32;
    if constexpr (k >= 4 && ((k - 4) & 3) == 3) {
        constexpr int t = (k - 4) >> 2;
        if constexpr (t == 0) TUP_PIN4(g.sv);
        else if constexpr (t < 6) TUP_PIN4(g.q);
        else {
            TUP_PIN4(g.x);
            u32x4 pk;
#pragma unroll
            for (int n = 0; n < 4; ++n) pk[n] = __builtin_bit_cast(uint32_t, g.x[n]);
            hf[b] = __builtin_bit_cast(bf16x8, pk);
        }
    }
}
template <int K0, int K1> TUP_DEVICE void gelu_pins(GeluState& g, bf16x8 (&hf)[2]) {
    static_for<(K1 > K0 ? K1 - K0 : 0)>([&](auto k) { gelu_pin<K0 + decltype(k)::value>(g, hf); });
}
template <int K0, int K1> TUP_DEVICE void gelu_ops(GeluState& g, const f32x16& acc, bf16x8 (&hf)[2]) {
    static_for<(K1 > K0 ? K1 - K0 : 0)>([&](auto k) { gelu_op<K0 + decltype(k)::value>(g, acc, hf); });
}


// Diagnostic build only (`make diag`, STAMPS = true): s_memtime at phase boundaries, summed per phase kind over the launch, for the
// eight waves of two recorded workgroups; the values leave through a buffer nothing else reads.
constexpr int BS_NPH = 20;
enum { Q_PRO = 0, Q_BAR_PRO, Q_A0, Q_BAR_A0, Q_SLOT, Q_BAR_SLOT, Q_PROJ_PRE, Q_BAR_X, Q_PROJ, Q_LN2, Q_BAR_Y, Q_MLP, Q_BAR_MLP, Q_TOTAL, Q_S_DMA, Q_S_LOOP, Q_S_PV, Q_M_DMA };
__device__ unsigned long long tup_bs_stamps[2][8][BS_NPH];
TUP_DEVICE unsigned long long bs_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <bool STAMPS>
__global__ __launch_bounds__(BS_NT, 2) void blocks_stream_kernel(float* __restrict__ xio, bf16_t* __restrict__ xout16, int nwin, const StreamTable tbl, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sbase = lds_addr(smem);
    f32x16 R[6];
    bf16x8 tf[12];
    LnStats ln1{};       // LayerNorm1's sums over tiles 0-3 of the NEXT block's input ride in the gaps of the last MLP slot
    // (Measured and removed: a start stagger of the first round's workgroups in eight phases, against the chip-wide bursts of the
    // parking stores: 849.0 vs 850.4 us -- the parking is bound by each CU's own issue rate)
    unsigned long long ph[BS_NPH] = {}, tprev = 0, tstart = 0;
    if constexpr (STAMPS) tprev = tstart = bs_now();
#define BS_STAMP(K) do { if constexpr (STAMPS) { const unsigned long long t_ = bs_now(); ph[K] += t_ - tprev; tprev = t_; } } while (0)

#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
    // thread coordinates from an opaque copy of threadIdx per block (as loop invariants hipcc hoists and spills what derives from them)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < BS_NT);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int hf = wave & 1;
    const int win = blockIdx.x * 4 + (wave >> 1);
    const bool active = win < nwin;
    const int row0 = (active ? win : nwin - 1) * 64 + 32 * hf;
    // The wave's 32 token rows are one contiguous 24 KB stretch of x that no other wave touches.  Inside the launch it is the wave's
    // PARKING area for the residual stream, in lane-linear order (register quad q of all lanes = 1 KB at q * 1 KB + lane * 16): a
    // store / load instruction then moves one contiguous KB.  (In the window layout a lane's 64 bytes are a row apart from its
    // neighbour's: 24 such stores per wave ran at 7 bytes per cycle and CU -- 28 k cycles per block in the stamps.)  The window
    // layout is read at the start and written at the end of the launch, through an LDS transpose (full 128-byte lines per row).
    // (addressed as a scalar base + a 32-bit lane offset: as 64-bit per-lane pointers every access beyond the instruction's immediate
    // offset range cost two or three vector additions)
    typedef __attribute__((address_space(1))) char* gptr_b;
    gptr_b xpark_b = (gptr_b)(xio + (size_t)row0 * 192);
    asm volatile("" : "+s"(xpark_b));
    const uint32_t lane16u = (uint32_t)lane * 16;
    // the block's pointers as opaque scalar values: left as kernel-argument loads, hipcc re-loads them in front of every DMA piece
    // (s_load + s_waitcnt lgkmcnt(0): a wait that also drains the LDS fragment reads in flight)
    struct { const char* wqk; const char* wv; const char* wproj; const char* wmlp; const float* tab; const float* sbias; } bp;
    {
        const StreamBlock& kb = tbl.b[blk];
        bp.wqk = kb.wqk; bp.wv = kb.wv; bp.wproj = kb.wproj; bp.wmlp = wave < 2 ? kb.w1 : kb.w2; bp.tab = kb.tab; bp.sbias = kb.sbias;
        asm volatile("" : "+s"(bp.wqk), "+s"(bp.wv), "+s"(bp.wproj), "+s"(bp.wmlp), "+s"(bp.tab), "+s"(bp.sbias));
    }
    const uint32_t lane16 = (uint32_t)lane * 16;
    // per-lane LDS offsets of the A fragments: tile row r, 16-byte chunk 2 t + h of a 128-byte row, swizzled by (row >> 1) & 7
    uint32_t woff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) woff[t] = sbase + (uint32_t)(r * 128 + (((2 * t + h) ^ ((r >> 1) & 7)) << 4));
    const uint32_t tabh = sbase + L_TAB + (uint32_t)h * 64;                      // [..][h][16 floats]
    const uint32_t tabr = sbase + L_TAB + (uint32_t)r * 4;                       // bias K-step words [..][32 rows]
    const uint32_t kx_wr = sbase + L_KX + (uint32_t)wave * 1024 + lane16, kx_rd = sbase + L_KX + (uint32_t)(wave ^ 1) * 1024 + lane16;
    const uint32_t vx_wr = sbase + L_VX + (uint32_t)wave * 2048 + lane16, vx_rd = sbase + L_VX + (uint32_t)(wave ^ 1) * 2048 + lane16;
    const uint32_t of_ad = sbase + L_OF + (uint32_t)wave * 8192 + lane16;
    // the B fragment of a bias K-step: K columns 0 and 1 (lane half 0, elements 0 and 1) are 1.0
    const bf16x8 onesB = __builtin_bit_cast(bf16x8, u32x4{h == 0 ? 0x3f803f80u : 0u, 0u, 0u, 0u});
    // ... and the A fragment whose K column 0 is 1.0 in every row (the "- max" K-step of the softmax)
    const bf16x8 onesA = __builtin_bit_cast(bf16x8, u32x4{h == 0 ? 0x00003f80u : 0u, 0u, 0u, 0u});

    // a 12 KB tile by the four waves of one half of the workgroup (three pieces each)
    // ... and one of those three pieces (u): inside the MFMA streams the pieces go out one at a time, a few MFMAs apart -- requested
    // together at the top of a slot, the 24-48 pieces of the eight waves queue up in the CU's one address path while every wave
    // waits at its issue (0.4-0.55 k cycles per MLP slot in the stamps)
    // Inside the slots all pieces are requested by waves 0-3: they are the older wave of each SIMD, win its arbitration and reach
    // every barrier ~0.8 k cycles ahead of waves 4-7 (bar_slot / bar_mlp in the stamps) -- the issue stalls of the DMA come out of
    // that slack instead of out of the critical waves' streams.
    auto dma_tile_piece = [&](const char* src, int tile, int lds_off, int half, int u) {
        (void)half;
        if (wave < 4) { const int pc = wave * 3 + u; bs_dma(bs_rsrc(src), smem + lds_off + pc * 1024, lane16, tile * TILE + pc * 1024); }
    };
    auto dma_tile = [&](const char* src, int tile, int lds_off, int half) {
        if ((wave >> 2) == half) {
            const __amdgpu_buffer_rsrc_t rs = bs_rsrc(src);
#pragma unroll
            for (int u = 0; u < 3; ++u) { const int pc = (wave & 3) * 3 + u; bs_dma(rs, smem + lds_off + pc * 1024, lane16, tile * TILE + pc * 1024); }
        }
    };

    if (blk == 0) {
        // x -> R: instruction j of tile rt fetches rows 8j .. 8j+7, 128 bytes each (lane L: row 8j + (L >> 3), 16-byte chunk L & 7);
        // the tile is turned in a 4 KB LDS scratch of the wave (the attention-output region, unused yet)
        const float* xg = xio + (size_t)(row0 + (lane >> 3)) * 192 + 4 * (lane & 7);
        char* scr = smem + L_OF + wave * 8192;
        f32x4 tmp[6][4];
#pragma unroll
        for (int rt = 0; rt < 6; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmp[rt][j] = *reinterpret_cast<const f32x4*>(xg + (size_t)(8 * j) * 192 + 32 * rt);
#pragma unroll
        for (int rt = 0; rt < 6; ++rt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(scr + j * 1024 + lane * 16) = tmp[rt][j];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(scr + r * 128 + (4 * h + m) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) R[rt][4 * m + e] = v[e];
            }
        }
    }
    // ---- block prologue: tables (requested ahead of the DMA: vmcnt retires in order), first weights on their way, LayerNorm1, the
    // residual stream out to memory ----
    // (the laundered pointers have lost their address space: without the casts their loads are flat_load -- counted on lgkmcnt as
    // well, out of order, under the hand-counted LDS waits)
    typedef const __attribute__((address_space(1))) float* gptr_f;
    typedef const __attribute__((address_space(1))) f32x4* gptr_f4;
    const gptr_f tabg = (gptr_f)bp.tab;
    const float tb0 = tabg[tid], tb1 = tabg[512 + tid], tb2 = tabg[1024 + tid];
    __syncthreads();                                       // everyone is done with the previous block's LDS
    {
        const uint32_t ta = sbase + L_TAB + (uint32_t)tid * 4;
        lds_write_b32(ta, tb0); lds_write_b32(ta + 2048, tb1); lds_write_b32(ta + 4096, tb2);
    }
    FENCE();
    dma_tile(bp.wqk, 0, L_WQK, 0);
    dma_tile(bp.wv, 0, L_WV, 1);
    dma_tile(bp.wqk, 1, L_WQK + TILE, 0);
    dma_tile(bp.wv, 1, L_WV + TILE, 1);
    // LayerNorm1: the sums of tiles 0-3 came with the previous block's last slot (block 0: here), tiles 4 and 5 here; the fragments
    // themselves are formed in the gaps of A(0)'s MFMAs, one K-step ahead
    float ln_rstd, ln_shift;
    {
        if (blk == 0) {
            ln1 = LnStats{};
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) ln_stats_tile(ln1, R[rt]);
        }
        ln_stats_tile(ln1, R[4]); ln_stats_tile(ln1, R[5]);
        ln_finish(ln1, ln_rstd, ln_shift);
        tf[0] = ln_frag<0>(R[0], ln_rstd, ln_shift);
        ln1 = LnStats{};
    }
    // (the residual stream is parked from inside A(0): one 1 KB store behind each of its 24 MFMAs -- requested together, the CU's 192
    // stores sat ~8 k cycles in the issue queue)
    FENCE();

    // ---- attention half ----
    f32x16 accqk, accv, S[2], O;
    bf16x8 Qf, Kfo, Kfp, Vfo[2], Vfp[2], P[4], Ofr[4];
    // relative position bias of (head, query half hf, key half kt): 4 KB each, 64 bytes per lane; scalar base per (head, kt)
    typedef const __attribute__((address_space(1))) char* gptr_c;
    const uint32_t lane64u = (uint32_t)lane * 64;
    auto load_sbias = [&](int hd) {                         // S[0] <- bias of (own keys), S[1] <- (partner's keys)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kt = t == 0 ? hf : 1 - hf;
            gptr_c sb = (gptr_c)bp.sbias + (size_t)(((hd * 2 + hf) * 2 + kt) * 4096);
            asm volatile("" : "+s"(sb));
            const gptr_f4 p4 = (gptr_f4)(sb + lane64u);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 v = p4[m];
#pragma unroll
                for (int e = 0; e < 4; ++e) S[t][4 * m + e] = v[e];
            }
        }
    };
    // weight fragment of K-step s of the tile at LDS offset T
#define WFRAG(T, s) lds_read_b128_asm_off(woff[(s) & 3], (T) + ((s) >> 2) * 4096)
#define BIASQ(OFF, m) lds_read_f4_off(tabh, ((OFF) + 4 * (m)) * 4)
    auto acc_from4 = [](f32x4 a, f32x4 b, f32x4 c, f32x4 d) {
        return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    load_sbias(0);
    BS_STAMP(Q_PRO);
    barrier_all();                                          // tables, first tiles
    BS_STAMP(Q_BAR_PRO);
    // ---- A(0): q|k of head 0 and v of pair 0 (all 12 K-steps), no softmax to carry yet ----
    {
        const f32x4 b0 = BIASQ(T_QKB + 0, 0), b1 = BIASQ(T_QKB + 0, 1), b2 = BIASQ(T_QKB + 0, 2), b3 = BIASQ(T_QKB + 0, 3);
        bf16x8 wq[4];
        wq[0] = WFRAG(L_WQK, 0); wq[1] = WFRAG(L_WQK, 1); wq[2] = WFRAG(L_WQK, 2);
        lds_wait<3>();
        FENCE();
        accqk = acc_from4(b0, b1, b2, b3);
        accv = zero16;
        static_for<24>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            // fragment g + 3 (g < 12: q|k tile K-step g; else v tile K-step g - 12)
            if constexpr (g + 3 < 24) {
                if constexpr (g + 3 < 12) wq[(g + 3) & 3] = WFRAG(L_WQK, g + 3); else wq[(g + 3) & 3] = WFRAG(L_WV, g + 3 - 12);
                lds_wait<3>();
            } else lds_wait<23 - g>();
            FENCE();
            if constexpr (g < 12) accqk = mfma32(wq[g & 3], tf[g], accqk);
            else accv = mfma32(tf[g - 12], wq[g & 3], accv);
            if constexpr (g + 1 < 12) tf[g + 1] = ln_frag<(g + 1) & 1>(R[(g + 1) >> 1], ln_rstd, ln_shift);
            {
                constexpr int rt = g >> 2, m = g & 3;
                gptr_b pb = xpark_b + rt * 4096;         // one scalar base per 4 KB (the immediate offset's range)
                asm volatile("" : "+s"(pb));
                if (active) *(__attribute__((address_space(1))) f32x4*)(pb + m * 1024 + lane16u) = f32x4{R[rt][4 * m], R[rt][4 * m + 1], R[rt][4 * m + 2], R[rt][4 * m + 3]};
            }
            FENCE();
        });
        Qf = pack8_regs<0>(accqk); Kfo = pack8_regs<8>(accqk);
        Vfo[0] = pack8_regs<0>(accv); Vfo[1] = pack8_regs<8>(accv);
        lds_write_b128(kx_wr, Kfo);
        lds_write_b128(vx_wr, Vfo[0]); lds_write_b128_off(vx_wr, Vfo[1], 1024);
        FENCE();
        BS_STAMP(Q_A0);
        barrier_all();
        BS_STAMP(Q_BAR_A0);
    }

    // ---- slots: head hd's S^T, softmax, PV with head hd + 1's q|k (and half of the next pair's v) between them ----
    static_for<12>([&](auto hd_) {
        constexpr int hd = decltype(hd_)::value;
        constexpr int NQK = hd + 1 < 12 ? 12 : 0;                       // q|k K-steps of head hd + 1
        constexpr int VP = hd / 2 + 1;                                  // the pair whose v is on its way in this slot
        constexpr int NV = VP < 6 ? 6 : 0, V0 = (hd & 1) * 6;           // its K-steps V0 .. V0 + 5
        constexpr int NG = NQK + NV;
        constexpr int TQK = L_WQK + ((hd + 1) & 1) * TILE, TV = L_WV + (VP & 1) * TILE;
        // weights two heads ahead into the ring slot head hd's consumers left at the last barrier
        // (the pieces themselves go out behind MFMAs 1, 4 and 7 of the loop below)
        constexpr bool DQK = hd + 2 < 12, DV = (hd & 1) == 0 && hd / 2 + 2 < 6;
        // the proj weight's first four tiles take the ring slots as they fall free: slot 8 -> v ring 0, slot 10 -> q|k ring 0 and v ring 1,
        // slot 11 -> q|k ring 1 (tiles 4 and 5 follow into region B behind the last slot's barrier)
        constexpr int PJ0 = hd == 8 ? 0 : hd == 10 ? 1 : hd == 11 ? 3 : -1, PJ1 = hd == 10 ? 2 : -1;
        Kfp = lds_read_b128_asm_off(kx_rd, (hd & 1) * 8192);
        if constexpr ((hd & 1) == 0) { Vfp[0] = lds_read_b128_asm_off(vx_rd, 0); Vfp[1] = lds_read_b128_asm_off(vx_rd, 1024); }
        lds_wait<0>();
        FENCE();
        BS_STAMP(Q_S_DMA);
        // S^T goes in a few MFMAs into the slot: its accumulator input (the relative position bias, requested before the last
        // barrier, which did not wait for it) has that much longer to arrive
        // (no later than the slot's first DMA piece: hipcc's wait for the bias loads in front of the S^T MFMAs is vmcnt(0) when an
        // LDS-DMA stands between them, i.e. a wait for pieces requested a moment ago)
        constexpr int GS = -1;
        if constexpr (GS < 0) { S[0] = mfma32(Kfo, Qf, S[0]); S[1] = mfma32(Kfp, Qf, S[1]); FENCE(); }
        SmState sm;
        if constexpr (NG > 0) {
            const f32x4 b0 = BIASQ(T_QKB + (hd + 1) * 32, 0), b1 = BIASQ(T_QKB + (hd + 1) * 32, 1), b2 = BIASQ(T_QKB + (hd + 1) * 32, 2),
                        b3 = BIASQ(T_QKB + (hd + 1) * 32, 3);
            bf16x8 wq[4];
            auto frag = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g < NQK) return WFRAG(TQK, g); else return WFRAG(TV, V0 + g - NQK);
            };
            wq[0] = frag(std::integral_constant<int, 0>{}); wq[1] = frag(std::integral_constant<int, 1>{}); wq[2] = frag(std::integral_constant<int, 2>{});
            lds_wait<2>();
            FENCE();
            if constexpr (NQK > 0) accqk = acc_from4(b0, b1, b2, b3);
            if constexpr (NV > 0 && V0 == 0) accv = zero16;
            // Gap g: MFMA g, the request of fragment g + 3 right behind it (an inline-asm ds_read behind packed / converting vector
            // instructions costs an s_nop), this gap's vector work, then the wait that makes fragment g + 1 ready.
            // Softmax operations: none behind the first two MFMAs (S^T is still in the pipe), then evenly
            static_for<NG>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g < NQK) accqk = mfma32(wq[g & 3], tf[g], accqk);
                else accv = mfma32(tf[V0 + g - NQK], wq[g & 3], accv);
                if constexpr (g == 1 || g == 4 || g == 7) {
                    if constexpr (DQK) dma_tile_piece(bp.wqk, hd + 2, L_WQK + (hd & 1) * TILE, hd & 1, g / 3);
                    if constexpr (DV) dma_tile_piece(bp.wv, hd / 2 + 2, L_WV + ((hd / 2) & 1) * TILE, 1 - (hd & 1), g / 3);
                    if constexpr (PJ0 >= 0) dma_tile_piece(bp.wproj, PJ0, PROJ_T[PJ0], PJ0 & 1, g / 3);
                    if constexpr (PJ1 >= 0) dma_tile_piece(bp.wproj, PJ1, PROJ_T[PJ1], PJ1 & 1, g / 3);
                }
                if constexpr (g == GS) { S[0] = mfma32(Kfo, Qf, S[0]); S[1] = mfma32(Kfp, Qf, S[1]); }
                // maximum in gaps G0 .. G0 + 3, the two "- max" MFMAs behind MFMA G0 + 4, exp2 and the bf16 fragments from G0 + 7 on
                constexpr int G0 = GS + 3, GM = G0 + 4, GE = GM + 3, NGV = NG - GE;
                if constexpr (g >= G0 && g < GM) softmax_ops<(SM_MAX_OPS * (g - G0)) / 4, (SM_MAX_OPS * (g - G0 + 1)) / 4>(sm, S, P, h == 0);
                if constexpr (g == GM) { S[0] = mfma32(onesA, sm.nmB, S[0]); S[1] = mfma32(onesA, sm.nmB, S[1]); }
                if constexpr (g >= GE) softmax_ops<SM_MAX_OPS + ((SM_OPS - SM_MAX_OPS) * (g - GE)) / NGV, SM_MAX_OPS + ((SM_OPS - SM_MAX_OPS) * (g - GE + 1)) / NGV>(sm, S, P, h == 0);
                if constexpr (g + 3 < NG) wq[(g + 3) & 3] = frag(std::integral_constant<int, g + 3>{});
                if constexpr (g + 1 < NG) lds_wait<(NG - 2 - g < 2 ? NG - 2 - g : 2)>();
                FENCE();
            });
        } else {
            if constexpr (PJ0 >= 0) dma_tile(bp.wproj, PJ0, PROJ_T[PJ0], PJ0 & 1);
            softmax_ops<0, SM_MAX_OPS>(sm, S, P, h == 0);
            FENCE();
            S[0] = mfma32(onesA, sm.nmB, S[0]); S[1] = mfma32(onesA, sm.nmB, S[1]);
            FENCE();
            softmax_ops<SM_MAX_OPS, SM_OPS>(sm, S, P, h == 0);
            FENCE();
        }
        BS_STAMP(Q_S_LOOP);
        // S's registers are free: the next head's relative position bias lands in them while PV runs
        if constexpr (hd + 1 < 12) load_sbias(hd + 1);
        FENCE();
        // O^T (+ row sums) = [V^T ; 1] P^T: the pair's other head's 16 lanes become ones
        {
            const bool mine = (hd & 1) == 0 ? r < 16 : r >= 16;
            auto sel = [&](bf16x8 v) {
                const u32x4 u = __builtin_bit_cast(u32x4, v);
                return __builtin_bit_cast(bf16x8, u32x4{mine ? u[0] : 0x3f803f80u, mine ? u[1] : 0x3f803f80u, mine ? u[2] : 0x3f803f80u, mine ? u[3] : 0x3f803f80u});
            };
            const bf16x8 v0 = sel(Vfo[0]), v1 = sel(Vfo[1]);
            FENCE();
            O = mfma32(v0, P[0], zero16);
            O = mfma32(v1, P[1], O);
            const bf16x8 v2 = sel(Vfp[0]), v3 = sel(Vfp[1]);
            FENCE();
            O = mfma32(v2, P[2], O);
            O = mfma32(v3, P[3], O);
            FENCE();
        }
        BS_STAMP(Q_S_PV);
        // head hd + 1's fragments (own registers; K and, when a pair is complete, V also to the partner through LDS)
        if constexpr (NQK > 0) {
            Qf = pack8_regs<0>(accqk); Kfo = pack8_regs<8>(accqk);
            lds_write_b128_off(kx_wr, Kfo, ((hd + 1) & 1) * 8192);
        }
        if constexpr (NV > 0 && V0 == 6) {
            Vfo[0] = pack8_regs<0>(accv); Vfo[1] = pack8_regs<8>(accv);
            lds_write_b128(vx_wr, Vfo[0]); lds_write_b128_off(vx_wr, Vfo[1], 1024);
        }
        FENCE();
        {   // normalise and park head hd's output: heads 0-7 in LDS, 8-11 in registers
            constexpr int c0 = (hd & 1) == 0 ? 0 : 8, s0 = (hd & 1) == 0 ? 8 : 0;
            const float inv = __builtin_amdgcn_rcpf(O[s0]);
            const bf16x8 of = pack8(O[c0] * inv, O[c0 + 1] * inv, O[c0 + 2] * inv, O[c0 + 3] * inv, O[c0 + 4] * inv, O[c0 + 5] * inv,
                                    O[c0 + 6] * inv, O[c0 + 7] * inv);
            if constexpr (hd < 8) lds_write_b128_off(of_ad, of, hd * 1024); else Ofr[hd - 8] = of;
        }
        FENCE();
        BS_STAMP(Q_SLOT);
        // everything this wave requested at the top of the slot has landed; the next head's 8 bias loads (younger) may still be on their way
        if constexpr (hd + 1 < 12) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else barrier_all();
        BS_STAMP(Q_BAR_SLOT);
    });

    // ---- proj + residual; LayerNorm2 ----
    // tiles 4 and 5 of the proj weight into region B (its readers passed the last barrier); tiles 0-3 are in place
    dma_tile(bp.wproj, 4, PROJ_T[4], 0);
    dma_tile(bp.wproj, 5, PROJ_T[5], 1);
    bf16x8 Ofl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) Ofl[i] = lds_read_b128_asm_off(of_ad, i * 1024);
    // the parked residual stream straight into the proj accumulators, two tiles ahead of the MFMAs that accumulate onto it (all 24
    // loads at once sat 8 k cycles in the issue queue of the critical waves: the CU's 192 KB come back at the L2 / fabric rate)
    auto load_park = [&](auto rt_, auto m_) {
        constexpr int rt = decltype(rt_)::value, m = decltype(m_)::value;
        gptr_b pb = xpark_b + rt * 4096;
        asm volatile("" : "+s"(pb));
        const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(pb + m * 1024 + lane16u);
#pragma unroll
        for (int e = 0; e < 4; ++e) R[rt][4 * m + e] = v[e];
    };
    static_for<8>([&](auto i_) { load_park(std::integral_constant<int, decltype(i_)::value / 4>{}, std::integral_constant<int, decltype(i_)::value % 4>{}); });
    lds_wait<0>();
    FENCE();
    BS_STAMP(Q_PROJ_PRE);
    // an MLP chunk = 24 pieces (< 12: the mlp.0 tile), six per wave of waves 0-3 (waves 0, 1: mlp.0; 2, 3: mlp.2)
    const int mw = wave * 6;
    auto dma_chunk_piece = [&](int c, int lds_off, int u) {
        if (wave < 4) { const int pc = mw + u; bs_dma(bs_rsrc(bp.wmlp), smem + lds_off + pc * 1024, lane16, c * TILE + (wave < 2 ? pc : pc - 12) * 1024); }
    };
    // ... and inside the slots (bs_dma_on): pieces u = 0..3 and 4, 5 of a chunk off two bases each; waves 4-7 (bit 31): none
    const __amdgpu_buffer_rsrc_t rs_mlp = bs_rsrc(bp.wmlp);
    const uint32_t mlp_lds0 = __builtin_amdgcn_readfirstlane(sbase + (uint32_t)(mw * 1024) + (wave < 4 ? 0u : 0x80000000u));
    const uint32_t mlp_src0 = __builtin_amdgcn_readfirstlane((uint32_t)((wave < 2 ? mw : mw - 12) * 1024));
    auto dma_chunk = [&](int c, int lds_off) {
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_chunk_piece(c, lds_off, u);
    };
    // MLP weight ring: six 24 KB slots from L_A; chunk c sits in slot (c + 3) % 6, so the first chunks land beyond the proj tiles
    auto ring_off = [&](int c) { return L_A + ((c + 3) % 6) * CHUNK; };
    {
        LnStats st{};
        static_for<6>([&](auto rt_) {
            constexpr int rt = decltype(rt_)::value;
            if constexpr (rt == 4) {
                // tiles 4 and 5 have landed everywhere; everyone holds its attention outputs in registers, so the first two MLP chunks
                // may take the attention-output region
                BS_STAMP(Q_PROJ);
                barrier_all();
                BS_STAMP(Q_BAR_X);
                dma_chunk(0, ring_off(0));
                dma_chunk(1, ring_off(1));
            }
            // (tiles 4 and 5 sit beyond the 16-bit immediate offset of a ds_read: their own base registers)
            constexpr int TB = rt < 4 ? 0 : L_B, T = PROJ_T[rt] - TB;
            uint32_t wb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wb[t] = woff[t] + TB;
#define WFRAGB(T, s) lds_read_b128_asm_off(wb[(s) & 3], (T) + ((s) >> 2) * 4096)
            bf16x8 wq[4];
            wq[0] = WFRAGB(T, 0); wq[1] = WFRAGB(T, 1); wq[2] = WFRAGB(T, 2);
            const uint32_t bw = lds_read_b32_off(tabr, (T_BP + rt * 32) * 4);
            lds_wait<0>();
            FENCE();
            static_for<12>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g + 3 < 12) { wq[(g + 3) & 3] = WFRAGB(T, g + 3); lds_wait<3>(); } else lds_wait<11 - g>();
                FENCE();
                if constexpr (g < 8) R[rt] = mfma32(wq[g & 3], Ofl[g], R[rt]); else R[rt] = mfma32(wq[g & 3], Ofr[g - 8], R[rt]);
                if constexpr (rt + 2 < 6 && g % 3 == 1) load_park(std::integral_constant<int, rt + 2>{}, std::integral_constant<int, g / 3>{});
                // LayerNorm2's sums over the previous tile ride in the gaps (from gap 3 on: that tile's last MFMA has to drain)
                if constexpr (rt > 0 && g >= 3) ln_stats_ops<(32 * (g - 3)) / 9, (32 * (g - 2)) / 9>(st, R[rt - 1]);
                FENCE();
            });
            R[rt] = mfma32(__builtin_bit_cast(bf16x8, u32x4{bw, 0u, 0u, 0u}), onesB, R[rt]);
            FENCE();
        });
        BS_STAMP(Q_PROJ);
        ln_stats_tile(st, R[5]);
        float rstd, shift;
        ln_finish(st, rstd, shift);
#pragma unroll
        for (int rt = 0; rt < 6; ++rt) { tf[2 * rt] = ln_frag<0>(R[rt], rstd, shift); tf[2 * rt + 1] = ln_frag<1>(R[rt], rstd, shift); }
        // + mlp.2's bias (a bias K-step per tile)
        static_for<6>([&](auto rt_) {
            constexpr int rt = decltype(rt_)::value;
            const uint32_t bw = lds_read_b32_off(tabr, (T_B2 + rt * 32) * 4);
            lds_wait<0>();
            FENCE();
            R[rt] = mfma32(__builtin_bit_cast(bf16x8, u32x4{bw, 0u, 0u, 0u}), onesB, R[rt]);
        });
    }
    FENCE();
    BS_STAMP(Q_LN2);
    barrier_all();                                          // everyone is done with the proj weight; chunks 0 and 1 landed
    BS_STAMP(Q_BAR_Y);

    // ---- MLP half: slot c = fc1 of chunk c | GELU of chunk c - 1 | fc2 of chunk c - 2 ----
    {
        f32x16 acc1[2];
        bf16x8 hfr[2][2];
        GeluState gs;
        uint32_t w2off[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) w2off[u] = sbase + (uint32_t)(r * 64 + (((2 * u + h) ^ ((r >> 2) & 3)) << 4));
        int ri = 3;                                      // ring slot of the chunk whose fc1 the next pair starts with (chunk 0: slot 3)
        // A PAIR of slots (chunks c, c + 1) is one stream of MFMAs with ONE fragment pipeline: the ring of fragment reads runs through
        // the boundary between the two slots, and the second slot's mlp.0 bias is requested from inside the first once the GELU
        // conversions have read the accumulators it goes into.  (Per slot: ~40 instructions of addresses, bias reads and a pipeline
        // start less; the 0.33 k cycles per slot the stamps showed for them were mostly hidden by the SIMD's other wave.)
        // DM (first slot only): the pair also requests the chunks c + 2 and c + 3 (twelve pieces per wave of waves 0-3)
        auto pair = [&](auto A_, auto B_, int c) {
            using A = decltype(A_); using B = decltype(B_);
            static_assert(!B::DM && A::PAR == 0 && B::PAR == 1, "");
            constexpr int NA = A::NM, NB = B::NM, NT = NA + NB;
            // ring slots of the chunks c - 2 .. c + 3 from the carried index ri = (c + 3) % 6 (a modulo per offset was ~30 scalar
            // instructions at the top of every slot)
            auto wrap = [](int v) { return v >= 6 ? v - 6 : v; };
            // LDS bases and source offsets of the pieces this pair requests: chunk c + 2 -> ring slot ri + 2, chunk c + 3 -> ri + 3
            uint32_t dl[2] = {0, 0}, ds[2] = {0, 0};
            if constexpr (A::DM) {
                dl[0] = mlp_lds0 + (uint32_t)(L_A + wrap(ri + 2) * CHUNK); dl[1] = mlp_lds0 + (uint32_t)(L_A + wrap(ri + 3) * CHUNK);
                ds[0] = mlp_src0 + (uint32_t)((c + 2) * TILE); ds[1] = ds[0] + TILE;
            }
            // fragment addresses: [slot][..] = mlp.0 tile of chunk c / c + 1, mlp.2 tile of chunk c - 2 / c - 1
            uint32_t a1[2][4], a2[2][2];
            {
                const uint32_t o1[2] = {(uint32_t)(L_A + ri * CHUNK), (uint32_t)(L_A + wrap(ri + 1) * CHUNK)};
                const uint32_t o2[2] = {(uint32_t)(L_A + wrap(ri + 4) * CHUNK + TILE), (uint32_t)(L_A + wrap(ri + 5) * CHUNK + TILE)};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) a1[q][t] = woff[t] + o1[q];
#pragma unroll
                    for (int u = 0; u < 2; ++u) a2[q][u] = w2off[u] + o2[q];
                }
            }
            ri = wrap(ri + 2);
            const uint32_t tb = tabh + (uint32_t)((T_B1 + c * 32) * 4);
            BS_STAMP(Q_M_DMA);
            // fragment list: position g -> slot g / NA, n = g % NA; in a slot: (fc1 K-step n / 2) on even n, (fc2 (rt, u) = (n / 4, (n / 2) & 1))
            // on odd n (a slot with only one of the two: K-step n)
            auto frag = [&](auto g_) {
                constexpr int g = decltype(g_)::value, q = g < NA ? 0 : 1, n = g < NA ? g : g - NA;
                constexpr bool F1 = q == 0 ? A::F1 : B::F1, F2 = q == 0 ? A::F2 : B::F2;
                constexpr bool is1 = F1 && (!F2 || (n & 1) == 0);
                constexpr int k = (F1 && F2) ? n / 2 : n;
#ifdef TUP_BSX_NOLDS          // timing experiment (wrong results): the MLP slots without their LDS fragment reads
                return __builtin_bit_cast(bf16x8, u32x4{a1[q][k & 3], a2[q][k & 1], (uint32_t)n, 0u});
#else
                if constexpr (is1) return lds_read_b128_asm_off(a1[q][k & 3], (k >> 2) * 4096);
                else return lds_read_b128_asm_off(a2[q][k & 1], (k >> 1) * 2048);
#endif
            };
            // fragments LA MFMAs ahead (ring of LA + 1): with eight waves reading 1 KB per MFMA the LDS runs half busy and a read takes
            // longer than the three gaps the attention loops allow
            constexpr int LA = 5, RS = LA + 1;
            // the second slot's bias: requested in gap NBIAS of the first, behind that gap's fragment (the GELU conversions -- operations
            // 0-3 and 32-35 of the slot's list -- are done with the accumulators)
            constexpr int NBIAS = A::GE ? (35 * NA) / GELU_OPS + 1 : 1;
            static_assert(!B::F1 || NBIAS + LA + 2 < NA, "the bias has landed before the second slot starts");
            bf16x8 wq[RS];
            f32x4 b0, b1, b2, b3, c0, c1, c2, c3;
            if constexpr (A::F1) { b0 = lds_read_f4_off(tb, 0); b1 = lds_read_f4_off(tb, 16); b2 = lds_read_f4_off(tb, 32); b3 = lds_read_f4_off(tb, 48); }
            static_for<LA>([&](auto i_) { wq[decltype(i_)::value] = frag(i_); });
            lds_wait<LA - 1>();
            FENCE();
            if constexpr (A::F1) acc1[0] = acc_from4(b0, b1, b2, b3);
            static_for<NT>([&](auto g_) {
                constexpr int g = decltype(g_)::value, q = g < NA ? 0 : 1, n = g < NA ? g : g - NA;
                constexpr int NM = q == 0 ? NA : NB, PAR = q;
                constexpr bool F1 = q == 0 ? A::F1 : B::F1, F2 = q == 0 ? A::F2 : B::F2, GE = q == 0 ? A::GE : B::GE;
                constexpr bool DM = q == 0 && A::DM, LNS = q == 0 ? A::LNS : B::LNS;
                constexpr bool is1 = F1 && (!F2 || (n & 1) == 0);
                constexpr int k = (F1 && F2) ? n / 2 : n;
                if constexpr (g == NA && B::F1) acc1[1] = acc_from4(c0, c1, c2, c3);
                if constexpr (is1) acc1[PAR] = mfma32(wq[g % RS], tf[k], acc1[PAR]);
                else R[k >> 1] = mfma32h(wq[g % RS], hfr[PAR][k & 1], R[k >> 1]);
                FENCE();                                   // the MFMA heads its gap (left to itself hipcc pairs the MFMAs of two gaps up)
                // Order of a gap: the MFMA; the pins of the PREVIOUS gap's GELU results (an asm right behind the packed instruction whose
                // result it names, or right in front of one that reads it, costs an s_nop); the LDS request; the wait; this gap's GELU work
#ifndef TUP_BSX_NOGELU
                if constexpr (GE && n > 0) gelu_pins<(GELU_OPS * (n - 1)) / NM, (GELU_OPS * n) / NM>(gs, hfr[1 - PAR]);
#endif
                if constexpr (g + LA < NT) wq[(g + LA) % RS] = frag(std::integral_constant<int, g + LA>{});
                if constexpr (B::F1 && g == NBIAS) {
                    c0 = lds_read_f4_off(tb, 128); c1 = lds_read_f4_off(tb, 128 + 16); c2 = lds_read_f4_off(tb, 128 + 32); c3 = lds_read_f4_off(tb, 128 + 48);
                }
                // (one wait per TWO gaps: at even g the fragments of MFMAs g + 1 and g + 2 are made ready -- every instruction of a gap is an
                // issue slot of the SIMD; the four bias reads count while they are younger than the fragment of MFMA g + 2)
                if constexpr ((g & 1) == 0 && g + 1 < NT) {
                    constexpr int FR = NT - 3 - g < LA - 2 ? (NT - 3 - g < 0 ? 0 : NT - 3 - g) : LA - 2;
                    constexpr int BI = (B::F1 && g + 2 - LA <= NBIAS && NBIAS <= g) ? 4 : 0;
                    lds_wait<FR + BI>();
                }
#ifndef TUP_BSX_NOGELU        // timing experiment (wrong results): ... without the GELU arithmetic (fc1's accumulators kept alive)
                if constexpr (GE) gelu_ops<(GELU_OPS * n) / NM, (GELU_OPS * (n + 1)) / NM>(gs, acc1[1 - PAR], hfr[1 - PAR]);
#else
                if constexpr (GE && n == NM - 1) { asm volatile("" :: "v"(acc1[1 - PAR])); asm volatile("" : "+v"(hfr[1 - PAR][0]), "+v"(hfr[1 - PAR][1])); }
#endif
#ifndef TUP_BSX_NODMA         // ... without the weight DMA requests (stale weights)
                if constexpr (DM && n % (NM / 12) == NM / 12 - 1) {
                    constexpr int pi = n / (NM / 12);
                    constexpr int dq = pi / 6, u = pi % 6;
                    bs_dma_on<(u & 3) * 1024>(rs_mlp, dl[dq] + (u >> 2) * 4096, lane16, ds[dq] + (u >> 2) * 4096);
                }
#endif
                if constexpr (LNS && n >= 4) ln_stats_ops<16 * (n & 1), 16 * (n & 1) + 16>(ln1, R[(n - 4) >> 1]);
#ifndef TUP_BSX_NOGELU
                if constexpr (GE && n == NM - 1) gelu_pins<(GELU_OPS * n) / NM, GELU_OPS>(gs, hfr[1 - PAR]);
#endif
                FENCE();
            });
            BS_STAMP(Q_MLP);
        };
        // One barrier per pair: at the start of pair k (slots 2k, 2k + 1) the chunks 2k + 2 and 2k + 3 are requested into the ring
        // slots of chunks 2k - 4 and 2k - 3 (whose mlp.2 tiles were last read in slots 2k - 2 and 2k - 1); they are needed a pair later.
        auto pair_end = [&]() { barrier_all(); BS_STAMP(Q_BAR_MLP); };
        pair(MlpSlot<0, true, false, false, true>{}, MlpSlot<1, true, true, false, false>{}, 0);
        pair_end();
#pragma unroll 1
        for (int c = 2; c < 22; c += 2) {
            pair(MlpSlot<0, true, true, true, true>{}, MlpSlot<1, true, true, true, false>{}, c);
            pair_end();
        }
        pair(MlpSlot<0, true, true, true, false>{}, MlpSlot<1, true, true, true, false>{}, 22);
        pair(MlpSlot<0, false, true, true, false>{}, MlpSlot<1, false, false, true, false>{}, 24);
        BS_STAMP(Q_MLP);
    }
    }   // blk
    if (xout16 != nullptr) {
        // R -> bf16 tokens [nwin * 64][192] in the window layout (the operand patch_unembed's GEMM rounds x to anyway; x itself then
        // only holds the parked stream): two passes of three tiles through the wave's LDS scratch as a linear image of 32 half rows
        // of 192 bytes, stored 16 bytes per lane
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
        const int win = blockIdx.x * 4 + (wave >> 1);
        char* scr = smem + L_OF + wave * 8192;
        char* og = reinterpret_cast<char*>(xout16) + (size_t)((win < nwin ? win : nwin - 1) * 64 + 32 * (wave & 1)) * 384;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const f32x16& a = R[3 * q + t];
                *reinterpret_cast<bf16x8*>(scr + r * 192 + t * 64 + h * 32) = pack8(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
                *reinterpret_cast<bf16x8*>(scr + r * 192 + t * 64 + h * 32 + 16) = pack8(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int b = j * 1024 + lane * 16, row = b / 192, col = b - row * 192;
                const u32x4 v = *reinterpret_cast<const u32x4*>(scr + b);
                if (win < nwin) *reinterpret_cast<u32x4*>(og + (size_t)row * 384 + q * 192 + col) = v;
            }
        }
    } else {   // R -> x in the window layout, through the wave's LDS scratch (every LDS reader of the launch is behind the last barrier)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
        const int win = blockIdx.x * 4 + (wave >> 1);
        char* scr = smem + L_OF + wave * 8192;
        float* xg = xio + (size_t)((win < nwin ? win : nwin - 1) * 64 + 32 * (wave & 1) + (lane >> 3)) * 192 + 4 * (lane & 7);
#pragma unroll
        for (int rt = 0; rt < 6; ++rt) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
                *reinterpret_cast<f32x4*>(scr + r * 128 + (4 * h + m) * 16) = f32x4{R[rt][4 * m], R[rt][4 * m + 1], R[rt][4 * m + 2], R[rt][4 * m + 3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(scr + j * 1024 + lane * 16);
                if (win < nwin) *reinterpret_cast<f32x4*>(xg + (size_t)(8 * j) * 192 + 32 * rt) = v;
            }
        }
    }
    if constexpr (STAMPS) {
        ph[Q_TOTAL] = bs_now() - tstart;
        const int b = blockIdx.x, rec = b == 0 ? 0 : (b == 300 ? 1 : -1);
        if (rec >= 0 && (threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < BS_NPH; ++k) tup_bs_stamps[rec][threadIdx.x >> 6][k] = ph[k];
    }
#undef BS_STAMP
#undef WFRAG
#undef WFRAGB
#undef BIASQ
}

}  // namespace

// nblk (<= 8) consecutive WindowTransformerBlocks in one launch (the loop of model.py:288-289): x fp32 [nwin * 64][192] in window
// order; table: HOST array [nblk][7] of device pointers = the tensors of packing.pack_stream_block (wqk, wv, wproj, w1, w2, tab,
// sbias).  out_bf16 == nullptr: the result replaces x.  out_bf16 != nullptr: the result is written there as bf16 [nwin * 64][192]
// (round to nearest even of the same values) and x is left holding the kernel's parked intermediate -- for a consumer that rounds
// the tokens to bf16 anyway (tup_patch_unembed_fwd), which then reads half the bytes.
extern "C" int tup_blocks_stream_fwd(float* x, void* out_bf16, const void* const* table, int nblk, int nwin, void* stream)
{
    if (nwin <= 0 || nblk <= 0) return 0;
    if (nblk > BS_MAX_BLK || table == nullptr) return (int)hipErrorInvalidValue;
    StreamTable t{};
    for (int i = 0; i < nblk; ++i) {
        const void* const* r = table + (size_t)i * 7;
        t.b[i] = StreamBlock{(const char*)r[0], (const char*)r[1], (const char*)r[2], (const char*)r[3], (const char*)r[4],
                             (const float*)r[5], (const float*)r[6]};
    }
    const int nwg = (nwin + 3) / 4;
#ifdef TUP_DIAG
    if (getenv("TUP_BS_STAMPS")) {
        TUP_SET_DYN_LDS(blocks_stream_kernel<true>, BS_LDS);
        blocks_stream_kernel<true><<<dim3(nwg), dim3(BS_NT), BS_LDS, reinterpret_cast<hipStream_t>(stream)>>>(x, reinterpret_cast<bf16_t*>(out_bf16), nwin, t, nblk);
        TUP_CHECK_LAUNCH();
        return 0;
    }
#endif
    TUP_SET_DYN_LDS(blocks_stream_kernel<false>, BS_LDS);
    blocks_stream_kernel<false><<<dim3(nwg), dim3(BS_NT), BS_LDS, reinterpret_cast<hipStream_t>(stream)>>>(x, reinterpret_cast<bf16_t*>(out_bf16), nwin, t, nblk);
    TUP_CHECK_LAUNCH();
    return 0;
}

#ifdef TUP_DIAG
// Timing experiments only (`make diag`): per-phase cycle sums of the last TUP_BS_STAMPS=1 launch, [2 workgroups][8 waves][16 phases].
extern "C" int tup_debug_bs_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tup_bs_stamps), sizeof(unsigned long long) * 2 * 8 * BS_NPH);
}
#endif
