// Output tail for a LAST final_upscale stage of r = 2 (scales 2 and 4) as a register-streaming stencil, no LDS, no barriers (gfx950):
//     t1  = PixelShuffle(2)(Conv2d(3, 12, 3)(x))            final_upscale        models/FastTransformer/utils.py:62-63,74-75
//     sum = Conv2d(3, 3, 3)(t1) + upscaled_input            final_upscale_conv   models/FastTransformer/model.py:316-320
//     out = clamp(sum, 0, 1) (optional)                                          model.py:327
// RESIZE = true appends the antialiased Resize of model.py:323-325 (tap tables of at most 4 taps per axis, i.e. down-scaling by
// up to 1.5) and the clamp: the pre-resize sums never leave the wave.  Vertically the last 6 HR rows of a lane's two columns sit in
// a wave-private LDS ring (round 4: 36 registers less = three waves per SIMD instead of two; a lane reads back only what it wrote)
// and an output row is emitted as soon as its last input row exists; horizontally the vertically filtered row goes
// through a 1.5 KB wave-private LDS line from which every lane gathers the taps of its (up to two) output columns.
//
// Why not tiles: the tiled kernel (tail_fused.hip) walks the stencil chain through three LDS tiles with a barrier between stages; a
// tile spends 40 % of its life waiting for its window loads and the rest in short latency-bound phases (VALU floor 3.3 k of its
// 28 k cycles).  Here a WAVE owns 64 consecutive LR columns (60 produce output, 2 + 2 are halo) and marches down the LR rows of
// a band: a lane keeps its column's 3-row LR window and its 5-row t1 window in registers, gets its horizontal neighbours with
// DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) and the 405 wave-uniform weights as scalar operands; per LR row and
// lane 324 + 324 FMAs, 3 + 6 loads (requested one row ahead) and 6 stores; nothing is recomputed except two t1 rows per band.
// HBM sees x, upscaled_input and out once each: bound = HBM / VALU issue, about equal.
#include "common.h"

namespace {

constexpr int TS_COLS = 60;         // output LR columns per wave (lanes 2..61)
constexpr int TS_BAND_MIN = 12;     // LR rows per band: chosen by the launcher (a multiple of 3: the register windows rotate with period 3)

struct TailStreamParams {
    const float* x;            // [B][3][H][W]
    const float* wfu_t;        // [27][12]: k = cin*9 + ky*3 + kx, o = c*4 + i*2 + j (PixelShuffle: out[c][2h+i][2w+j] = conv[c*4+i*2+j][h][w])
    const float* bfu;          // [12]
    const float* wfc_t;        // [27][4]: k as above, o < 3 (+ one pad)
    const float* bfc;          // [3]
    const float* ui;           // [B][3][2H][2W]
    float* out;                // [B][3][2H][2W]
    int B, H, W, nstrip, nband, band_h, clamp01;
    // RESIZE: tap tables (as tup_resize_aa_fwd), ownership of output columns per strip / output rows per band, strip stride (LR
    // columns; 60 minus the horizontal tap reach) and the LR rows a band runs past its end for the vertical taps
    const int* ymin; const int* ysize; const float* yw; int KY;
    const int* xmin; const int* xsize; const float* xw; int KX;
    const int* oxb; const int* oyb;
    int Ho, Wo, sc, ext;
};

// Neighbour exchange by DPP wave shifts, as volatile asm: the values are fetched where they are used (a handful of transient
// registers); as intrinsics hipcc hoists every exchange of a stage to its top and spills.  hipcc does not see into the asm, so
// the VALU-write -> DPP-read hazard (2 wait states) is handled INSIDE the asm: `s_nop 1` in front of the DPP move, whatever hipcc
// schedules ahead of it (hipcc sinks the instruction that defines a value down to its first use, i.e. right in front of the
// exchange: found as wrong first rows of every band; rounds 3's fix pinned the producers a weight group earlier, which held only
// as long as hipcc kept that order).  The pins stay (they keep the nops from ever waiting); tests/test_hip_kernels.py checks the
// band seams.
TUP_DEVICE float from_left(float v) {     // lane l <- lane l - 1 (0 into lane 0)
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
}
TUP_DEVICE float from_right(float v) {    // lane l <- lane l + 1 (0 into lane 63)
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
}

// (the pointers are separate __restrict__ parameters: the weight reads inside the row loop stay scalar loads only while hipcc can
// prove that the kernel's own stores do not alias them)
template <bool RESIZE>
__global__ __launch_bounds__(256, 3) void tail_stream_r2_kernel(
    const float* __restrict__ x_arg, const float* __restrict__ wfu_arg, const float* __restrict__ bfu_arg,
    const float* __restrict__ wfc_arg, const float* __restrict__ bfc_arg, const float* __restrict__ ui_arg, float* __restrict__ out_arg,
    const TailStreamParams p)
{
    // The 405 weights + 15 biases are wave-uniform, but ~100 SGPRs cannot hold them (as scalar loads hipcc keeps 16-dword bursts
    // in flight and spills ~300 SGPRs through v_writelane / v_readlane, 18 % of the loop's VALU issues).  They live in LDS and are
    // read with same-address (broadcast, conflict-free) ds_read_b128 through a hand-pipelined ring of 4 x 12 VGPRs: the weight
    // stream of an iteration is 36 groups of 12 floats -- group g < 27 = tap g of the 3 -> 12 conv (12 outputs), group 27 + j = taps
    // 3j .. 3j+2 of the 3 -> 3 conv (4 floats each) -- followed by bfu [12] and bfc [3 + pad].  The reads are inline asm (hipcc
    // would hoist all of them to the top of the stage and spill 200 VGPRs); waits are counted by hand.  The only barrier of the
    // kernel follows the copy.
    __shared__ __attribute__((aligned(16))) float wl[36 * 12 + 12 + 4];
    __shared__ __attribute__((aligned(16))) float hb[RESIZE ? 4 : 1][3][128];          // RESIZE: one vertically filtered HR line per wave
    __shared__ __attribute__((aligned(16))) float xwl[RESIZE ? 4 : 1][2][64][4];       // RESIZE: the lanes' horizontal tap weights (per wave, pass, lane)
    // RESIZE: the last six HR rows of the sums (slot = HR row % 6 = 2 (LR row % 3) + i), per wave, channel and lane -- in LDS, not in 36
    // registers: a lane only ever reads what it wrote itself; an output row reads its (at most four) tap rows
    __shared__ __attribute__((aligned(16))) f32x2 vrl[RESIZE ? 4 : 1][6][3][64];
    for (int i = threadIdx.x; i < 27 * 12; i += 256) wl[i] = wfu_arg[i];
    if (threadIdx.x < 27 * 4) wl[324 + threadIdx.x] = wfc_arg[threadIdx.x];
    if (threadIdx.x < 12) wl[432 + threadIdx.x] = bfu_arg[threadIdx.x];
    if (threadIdx.x < 4) wl[444 + threadIdx.x] = threadIdx.x < 3 ? bfc_arg[threadIdx.x] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (wid >= p.B * p.nband * p.nstrip) return;
    const int strip = wid % p.nstrip, band = (wid / p.nstrip) % p.nband, b = wid / (p.nstrip * p.nband);
    const int H = p.H, W = p.W, Hs = 2 * H, Ws = 2 * W;
    const int xcol = strip * (RESIZE ? p.sc : TS_COLS) - 2 + lane;
    const bool colok = xcol >= 0 && xcol < W;
    const bool store_lane = lane >= 2 && lane < 2 + TS_COLS && xcol < W;
    const int y0 = band * p.band_h, y1 = min(y0 + p.band_h, H);
    const int y_end = RESIZE ? min(y1 + p.ext, H) : y1;             // RESIZE: the rows under the vertical taps of the band's last output rows
    // every address = wave-uniform row pointer (SALU) + one per-lane element offset, clamped so that masked lanes stay in range
    // Buffer addressing: resource descriptor (the tensor) + wave-uniform byte offset of the row (SGPR, SALU arithmetic) + one per-lane
    // byte offset (VGPR, fixed for the whole kernel): no vector address arithmetic in the row loop.  The launcher refuses tensors
    // of 2 GB or more (32-bit offsets).
    const unsigned xc = colok ? (unsigned)xcol : 0u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_arg), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ui_arg), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out_arg, 0, 0x7fffffff, 0x00020000);
    const unsigned vx = xc * 4u, vu = xc * 8u;

    // ---- register windows, indexed by (LR row + 3) % 3 ----
    float X[3][3];                  // [slot][channel] LR values of this lane's column (neighbours by DPP at use)
    float T[3][3][2][2];            // [slot][channel][HR row within the LR row][col 2x | col 2x+1]; the neighbours' columns 2x-1 and
                                    // 2x+2 are fetched by DPP where stage C uses a row (storing them cost 30 registers and tipped the
                                    // kernel into scratch, whose reloads wait on vmcnt behind the prefetched rows)
    // Loads in flight land in register PAIRS of their own: hipcc pairs neighbouring VGPRs into v_pk_fma_f32 operands, and a pair
    // with a pending load in its other half makes the instruction wait for that load (vmcnt(0) in the middle of stage B).
    f32x2 xn[3];                    // [0] = LR row r + 2 of channel c, requested at the top of iteration r
    f32x2 un[1][3][2];              // upscaled_input (HR rows 2q, 2q+1): requested at the end of iteration q (stage C of q - 1 has read the previous rows), used in q + 1

    // RESIZE state: the lane's output columns of the two gather passes with their first tap (as an index into the wave's LDS line),
    // the band's next output row (the last six HR rows of the sums and the lanes' tap weights live in LDS: vrl, xwl)
    int oxl[2] = {0, 0}, xi[2] = {0, 0};
    bool hval[2] = {false, false};
    int oy = 0, oy_end = 0;
    const int wv = threadIdx.x >> 6;
    if constexpr (RESIZE) {
        const int ox0 = p.oxb[strip], ox1 = p.oxb[strip + 1];
        const int hx0 = 2 * (strip * p.sc - 2);                           // HR column of LDS index 0 (lane 0's first column)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ox = ox0 + lane + 64 * q;
            hval[q] = ox < ox1;
            oxl[q] = hval[q] ? ox : ox0;
            xi[q] = hval[q] ? p.xmin[oxl[q]] - hx0 : 4;
            // (the four tap weights of the pass live in LDS, not in eight registers for the whole band: one ds_read_b128 per pass and
            // output row; the kernel is bound by what a wave can issue, and 168 registers = a third wave per SIMD)
            f32x4 wq4;
#pragma unroll
            for (int t = 0; t < 4; ++t) wq4[t] = (hval[q] && t < p.KX) ? p.xw[oxl[q] * p.KX + t] : 0.f;
            *reinterpret_cast<f32x4*>(&xwl[wv][q][lane][0]) = wq4;
        }
        oy = p.oyb[band];
        oy_end = p.oyb[band + 1];
    }
    int nym = 0, nyn = 0;                 // RESIZE: first tap row, tap count and weights of output row `oy` (wave-uniform)
    float nyw[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch_row_taps = [&](int row) __attribute__((always_inline)) {
        const int rr = row < oy_end ? row : oy_end - 1;
        nym = p.ymin[rr]; nyn = p.ysize[rr];
#pragma unroll
        for (int a = 0; a < 4; ++a) nyw[a] = a < p.KY ? p.yw[rr * p.KY + a] : 0.f;
    };
    if constexpr (RESIZE) { if (oy < oy_end) fetch_row_taps(oy); }

    auto load_x_row = [&](int row, f32x2 (&dst)[3]) __attribute__((always_inline)) {
        const bool rok = row >= 0 && row < H;                     // wave-uniform
        const int rr = rok ? row : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dst[c][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx, (((b * 3 + c) * H + rr) * W) * 4, 0));
    };
    auto place_x_row = [&](int row, int slot, const f32x2 (&src)[3]) __attribute__((always_inline)) {
        const bool ok = colok && row >= 0 && row < H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            X[slot][c] = ok ? src[c][0] : 0.f;
            asm volatile("" : "+v"(X[slot][c]));        // materialised HERE: see from_left
        }
    };
    auto load_ui_rows = [&](int q, int buf) __attribute__((always_inline)) {         // HR rows 2q, 2q+1 of this lane's column pair
        const int qq = (q >= 0 && q < H) ? q : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                un[buf][c][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ru, vu, (((b * 3 + c) * Hs + 2 * qq + i) * Ws) * 4, 0));
    };

    const uint32_t wbase = lds_addr(wl);
    f32x4 ring[3][3];
    auto rd = [&](int g) __attribute__((always_inline)) {       // group g (0..35) or the biases (g = 36: bfu, + bfc in slot [3][..] of `bias`)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[g % 3][j]) : "v"(wbase), "i"((g * 12 + j * 4) * 4));
    };
    // bfu (three reads at the top of stage B) and bfc (one read in front of the last group request of stage B: live for one group
    // instead of for the whole stage)
    f32x4 bias[4];
    auto rd_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias[j]) : "v"(wbase), "i"((432 + j * 4) * 4));
    };
    auto rd_bias3 = [&]() __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias[3]) : "v"(wbase), "i"((432 + 12) * 4));
    };

    // One LR row: stage B = T(r) from the LR rows r-1, r, r+1 (slots SM, S, SP), zero outside the image (the 3 -> 3 conv zero-pads
    // t1); stage C (doC) = HR rows 2q, 2q+1 of LR row q = r - 1 from t1 rows 2q-1 .. 2q+2 = T(q-1) row 1 (slot SP), T(q) rows 0, 1
    // (slot SM), T(q+1) row 0 (slot S), + upscaled_input (buffer ub) + bias, [clamp], store.
    auto stages = [&](int r, int SM, int S, int SP, bool doC, int ub) __attribute__((always_inline)) {
        rd_bias();
        rd(0); rd(1);
        lds_wait<6>();
        __builtin_amdgcn_sched_barrier(0);
        // RULE for the asynchronous reads: hipcc believes an asm output is written AT the asm statement.  An output that is never
        // used afterwards is dead to it from that point: the register goes to the next value computed, and the data lands on
        // top of that value later.  So every read's destination is "used" (really, or by an empty asm) BEHIND the wait that
        // covers it.
        // explicit pairs (v_pk_fma_f32): left to the SLP vectoriser the scalar chains of ALL taps are gathered into one block of
        // packed FMAs behind the last read -- every weight group live at once, 280 spilled registers
        f32x2 acc[6];                   // pair j = outputs 2j, 2j+1 = (channel j >> 1, HR row j & 1), columns 2x, 2x+1
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = f32x2{bias[j >> 1][2 * (j & 1)], bias[j >> 1][2 * (j & 1) + 1]};
        const int slots[3] = {SM, S, SP};
#pragma unroll
        for (int g = 0; g < 27; ++g) {
            // (the groups of the 3 -> 3 conv are requested whether or not stage C runs: a conditional request costs hipcc's register
            //  allocation more than the six wasted reads of a band's first two rows)
            // (request order at g = 26: groups 26 [being waited for], 27, bfc, 28 -- ten reads, the four oldest done = group 26)
            if (g == 26) rd_bias3();
            rd(g + 2); lds_wait<6>();
            __builtin_amdgcn_sched_barrier(0);
            const float xo = X[slots[(g / 3) % 3]][g / 9];
            const float v = g % 3 == 0 ? from_left(xo) : g % 3 == 2 ? from_right(xo) : xo;
            const f32x2 vv = {v, v};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x4 wq = ring[g % 3][j >> 1];
                acc[j] = __builtin_elementwise_fma(f32x2{wq[2 * (j & 1)], wq[2 * (j & 1) + 1]}, vv, acc[j]);
                asm volatile("" : "+v"(acc[j]));        // pins the FMA between this group's reads and the next: hipcc otherwise sinks
                                                        // whole taps below the (volatile) read sequence and spills the weights it still needs
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // Without stage C the two groups requested ahead (27, 28) are never used: drain, and keep their registers reserved until
        // then (RULE above; found as wrong first rows of every band but the first -- the last taps' FMAs had been given the
        // registers of the pending reads).
        if (!doC) {
            lds_wait<0>();
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(ring[27 % 3][j]), "v"(ring[28 % 3][j]));      // groups 27, 28: see RULE
            asm volatile("" ::"v"(bias[3]));                                                                 // ... and bfc
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool ok = colok && r >= 0 && r < H;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float a = ok ? acc[c * 2 + i][0] : 0.f, bq = ok ? acc[c * 2 + i][1] : 0.f;
                T[S][c][i][0] = a;
                T[S][c][i][1] = bq;
                asm volatile("" : "+v"(T[S][c][i][0]), "+v"(T[S][c][i][1]));       // materialised HERE: see from_left
            }
        if (!doC) return;
        const int q = r - 1;
        lds_wait<3>();                  // outstanding: group 27, bfc, group 28 -> the first two have landed
        __builtin_amdgcn_sched_barrier(0);
        f32x2 ac[2][3];                 // [HR row][channel] = columns 2x, 2x+1
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int o = 0; o < 3; ++o) ac[i][o] = un[ub][o][i] + f32x2{bias[3][o], bias[3][o]};
#pragma unroll
        for (int g = 27; g < 36; ++g) {
            if (g + 2 < 36) { rd(g + 2); lds_wait<6>(); }
            else if (g + 1 < 36) lds_wait<3>();
            else lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            const int c = (g - 27) / 3, ky = (g - 27) % 3;
            // t1 row of (output row i, tap row ky): i = 0 -> T(q-1)[1], T(q)[0], T(q)[1];  i = 1 -> T(q)[0], T(q)[1], T(q+1)[0]
            const int s0 = ky == 0 ? SP : SM, h0 = ky == 0 ? 1 : ky == 1 ? 0 : 1;      // (slot, HR row) of the t1 row under output row 0
            const int s1 = ky == 2 ? S : SM, h1 = ky == 0 ? 0 : ky == 1 ? 1 : 0;       // ... under output row 1
            // [left neighbour's column 2x-1 | 2x | 2x+1 | right neighbour's column 2x+2]
            const float a0[4] = {from_left(T[s0][c][h0][1]), T[s0][c][h0][0], T[s0][c][h0][1], from_right(T[s0][c][h0][0])};
            const float a1[4] = {from_left(T[s1][c][h1][1]), T[s1][c][h1][0], T[s1][c][h1][1], from_right(T[s1][c][h1][0])};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x2 t0 = {a0[kx], a0[kx + 1]}, t1v = {a1[kx], a1[kx + 1]};
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float w = ring[g % 3][kx][o];
                    const f32x2 ww = {w, w};
                    ac[0][o] = __builtin_elementwise_fma(ww, t0, ac[0][o]);
                    ac[1][o] = __builtin_elementwise_fma(ww, t1v, ac[1][o]);
                    asm volatile("" : "+v"(ac[0][o]), "+v"(ac[1][o]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!RESIZE) {
            if (store_lane) {
#pragma unroll
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x2 v = ac[i][o];
                        if (p.clamp01) { v[0] = fminf(fmaxf(v[0], 0.f), 1.f); v[1] = fminf(fmaxf(v[1], 0.f), 1.f); }
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ro, vu, (((b * 3 + o) * Hs + 2 * q + i) * Ws) * 4, 0);
                    }
            }
        } else {
            // HR rows 2q, 2q+1 enter the ring (q % 3 = SM); then every output row of this band whose last tap row now exists
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int o = 0; o < 3; ++o) vrl[wv][2 * SM + i][o][lane] = ac[i][o];
            while (oy < oy_end) {
                const int ym = nym, yn = nyn;                          // this row's taps were fetched an emission ago (scalar loads:
                if (ym + yn - 1 > 2 * q + 1) break;                    // their latency would otherwise sit in front of every row)
                // the row's taps: ring slot (wave-uniform) and weight; a tap beyond the row's count reads the first tap's row with weight 0
                // (an unwritten slot may hold anything)
                int sl[4];
                float wa[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    sl[a] = a < yn ? (ym + a) % 6 : ym % 6;
                    wa[a] = a < yn ? nyw[a] : 0.f;
                }
                fetch_row_taps(oy + 1);
                f32x2 vr[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    vr[c] = f32x2{wa[0], wa[0]} * vrl[wv][sl[0]][c][lane];
#pragma unroll
                    for (int a = 1; a < 4; ++a) vr[c] = __builtin_elementwise_fma(f32x2{wa[a], wa[a]}, vrl[wv][sl[a]][c][lane], vr[c]);
                    *reinterpret_cast<f32x2*>(&hb[wv][c][2 * lane]) = vr[c];
                }
#pragma unroll
                for (int pq = 0; pq < 2; ++pq)
                    if (hval[pq]) {
                        const f32x4 xwt = *reinterpret_cast<const f32x4*>(&xwl[wv][pq][lane][0]);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float o = xwt[0] * hb[wv][c][xi[pq]];
#pragma unroll
                            for (int t = 1; t < 4; ++t) o = __builtin_fmaf(xwt[t], hb[wv][c][xi[pq] + t], o);
                            if (p.clamp01) o = fminf(fmaxf(o, 0.f), 1.f);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ro, (unsigned)oxl[pq] * 4u,
                                                                  (((b * 3 + c) * p.Ho + oy) * p.Wo) * 4, 0);
                        }
                    }
                ++oy;
            }
        }
    };

    // iteration r (S = (r + 3) % 3): request LR row r + 2; T(r); the output rows of LR row r - 1 (their upscaled_input was
    // requested an iteration ago); request the upscaled_input rows of LR row r; finally LR row r + 2 takes the slot of row r - 1
    auto iteration = [&](int r, int S) __attribute__((always_inline)) {
        const int SM = (S + 2) % 3, SP = (S + 1) % 3;
        load_x_row(r + 2, xn);
        stages(r, SM, S, SP, r - 1 >= y0, 0);                  // T(r-2) lives in slot (r - 2 + 3) % 3 = SP, T(r-1) in SM, T(r) in S
        load_ui_rows(r, 0);
        place_x_row(r + 2, SM, xn);
    };

    // ---- prologue: LR rows y0-2, y0-1, y0 (for T(y0-1)); y0 % 3 == 0 (band_h is a multiple of 3), so the first iteration
    //      r = y0 - 1 has S = 2 ----
    {
        f32x2 t[3];
        load_x_row(y0 - 2, t); place_x_row(y0 - 2, 1, t);       // (y0 - 2 + 3) % 3 = 1
        load_x_row(y0 - 1, t); place_x_row(y0 - 1, 2, t);
        load_x_row(y0, t);     place_x_row(y0, 0, t);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 2; ++e) { T[s][c][i][e] = 0.f; asm volatile("" : "+v"(T[s][c][i][e])); }
    }
    int r = y0 - 1;
    iteration(r, 2); ++r;
    while (r <= y_end) {
        iteration(r, 0); if (++r > y_end) break;
        iteration(r, 1); if (++r > y_end) break;
        iteration(r, 2); ++r;
    }
}

}  // namespace

namespace {
int launch_tail_stream(TailStreamParams& p, const float* x, const float* wfu_t, const float* bfu, const float* wfc_t, const float* bfc,
                       const float* ui, float* out, bool resize, void* stream)
{
    const long long waves = (long long)p.B * p.nstrip * p.nband;
    if (waves > (1ll << 30) || (long long)p.B * 3 * 4 * p.H * p.W >= (1ll << 29)) return (int)hipErrorInvalidValue;     // 32-bit byte offsets
    const dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (resize) tail_stream_r2_kernel<true><<<grid, dim3(256), 0, s>>>(x, wfu_t, bfu, wfc_t, bfc, ui, out, p);
    else tail_stream_r2_kernel<false><<<grid, dim3(256), 0, s>>>(x, wfu_t, bfu, wfc_t, bfc, ui, out, p);
    TUP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// x fp32 [B][3][H][W]; wfu_t fp32 [27][12], bfu [12]; wfc_t fp32 [27][4], bfc [3] (packing.pack_planar_t);
// ui / out fp32 [B][3][2H][2W].  r = 2 only; any H, W >= 1.
extern "C" int tup_tail_stream_r2_fwd(const float* x, const float* wfu_t, const float* bfu, const float* wfc_t, const float* bfc,
                                      const float* ui, float* out, int B, int H, int W, int clamp01, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    TailStreamParams p{};
    p.B = B; p.H = H; p.W = W; p.clamp01 = clamp01;
    p.nstrip = (W + TS_COLS - 1) / TS_COLS;
    // bands: as many as fill the chip once at three waves per SIMD (256 CUs x 4 SIMDs x 3), each band a multiple of 3 rows and at
    // least TS_BAND_MIN (a band recomputes two t1 rows at its top)
    {
        const long long cols = (long long)B * p.nstrip;
        long long nb = (256 * 4 * 3) / (cols > 0 ? cols : 1);
        if (nb < 1) nb = 1;
        int bh = (int)((H + nb - 1) / nb);
        bh = (bh + 2) / 3 * 3;
        if (bh < TS_BAND_MIN) bh = TS_BAND_MIN;
        p.band_h = bh;
        p.nband = (H + bh - 1) / bh;
    }
    return launch_tail_stream(p, x, wfu_t, bfu, wfc_t, bfc, ui, out, false, stream);
}

// The same followed by the antialiased Resize to Ho x Wo (model.py:323-325) and the clamp: out fp32 [B][3][Ho][Wo].
// Tap tables as tup_resize_aa_fwd (every tap count <= 4).  The caller fixes the decomposition and says who owns what:
// sc = LR columns a strip advances by (<= 60 - ceil((max taps - 1) / 2)), band_h = LR rows per band (a multiple of 3), ext = LR rows a
// band runs past its end (ceil((max taps - 1) / 2)); oxb int [nstrip + 1], oyb int [nband + 1]: strip s owns the output columns
// [oxb[s], oxb[s+1]) = those whose first tap lies in its HR columns [2 s sc, 2 (s+1) sc), band k the output rows [oyb[k], oyb[k+1])
// whose first tap lies in its HR rows [2 k band_h, 2 (k+1) band_h); nstrip = ceil(W / sc), nband = ceil(H / band_h).
extern "C" int tup_tail_stream_r2_resize_fwd(const float* x, const float* wfu_t, const float* bfu, const float* wfc_t, const float* bfc,
                                             const float* ui, float* out, const int* ymin, const int* ysize, const float* yw, int KY,
                                             const int* xmin, const int* xsize, const float* xw, int KX, const int* oxb, const int* oyb,
                                             int B, int H, int W, int Ho, int Wo, int sc, int band_h, int ext, int clamp01, void* stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (sc < 1 || sc > TS_COLS || band_h < 3 || band_h % 3 != 0 || ext < 0 || ext > 2 || Ho < 1 || Wo < 1 || KX < 1 || KY < 1)
        return (int)hipErrorInvalidValue;
    if ((long long)B * 3 * Ho * Wo >= (1ll << 29)) return (int)hipErrorInvalidValue;
    TailStreamParams p{};
    p.B = B; p.H = H; p.W = W; p.clamp01 = clamp01;
    p.nstrip = (W + sc - 1) / sc;
    p.band_h = band_h;
    p.nband = (H + band_h - 1) / band_h;
    p.ymin = ymin; p.ysize = ysize; p.yw = yw; p.KY = KY; p.xmin = xmin; p.xsize = xsize; p.xw = xw; p.KX = KX;
    p.oxb = oxb; p.oyb = oyb; p.Ho = Ho; p.Wo = Wo; p.sc = sc; p.ext = ext;
    return launch_tail_stream(p, x, wfu_t, bfu, wfc_t, bfc, ui, out, true, stream);
}
