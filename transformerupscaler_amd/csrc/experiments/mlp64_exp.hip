// EXPERIMENT (not part of libtupscale_hip.so; built by `make exp`): the MLP half of the streamed block kernel in the
// one-wave-per-SIMD form DESIGN.md section 9 sizes ("Y1"): a workgroup of FOUR waves, each wave one whole window = 64 tokens = two
// 32-token tiles, so that every weight fragment read from LDS feeds TWO MFMAs and the wave has the 512-register budget.
//   x <- x + mlp.2(gelu(mlp.0(LN2(x))))          (model.py:103-104), operands = packing.pack_stream_block's w1 / w2 / tab
// Same weight ring, DMA pieces, GELU micro-operations and packed layouts as block_stream.hip (included for its helpers).
#include "../block_stream.hip"

namespace {

constexpr int M64_NT = 256;

template <int K0, int K1> TUP_DEVICE void gelu2_ops(GeluState& g, const f32x16 (&acc)[2], bf16x8 (&hf)[2][2]) {
    static_for<(K1 > K0 ? K1 - K0 : 0)>([&](auto k) {
        constexpr int K = K0 + decltype(k)::value;
        if constexpr (K < GELU_OPS) gelu_op<K>(g, acc[0], hf[0]); else gelu_op<K - GELU_OPS>(g, acc[1], hf[1]);
    });
}
template <int K0, int K1> TUP_DEVICE void gelu2_pins(GeluState& g, bf16x8 (&hf)[2][2]) {
    static_for<(K1 > K0 ? K1 - K0 : 0)>([&](auto k) {
        constexpr int K = K0 + decltype(k)::value;
        if constexpr (K < GELU_OPS) gelu_pin<K>(g, hf[0]); else gelu_pin<K - GELU_OPS>(g, hf[1]);
    });
}

__global__ __launch_bounds__(M64_NT, 1) void mlp64_kernel(float* __restrict__ xio, int nwin, const StreamBlock kb, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sbase = lds_addr(smem);
    f32x16 R[2][6];
    bf16x8 tf[2][12];
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < M64_NT);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int win = blockIdx.x * 4 + wave;
    const bool active = win < nwin;
    const int row0 = (active ? win : nwin - 1) * 64;
    struct { const char* wmlp; const float* tab; } bp;
    bp.wmlp = wave < 2 ? kb.w1 : kb.w2; bp.tab = kb.tab;
    asm volatile("" : "+s"(bp.wmlp), "+s"(bp.tab));
    const uint32_t lane16 = (uint32_t)lane * 16;
    uint32_t woff[4], w2off[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) woff[t] = sbase + (uint32_t)(r * 128 + (((2 * t + h) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
    for (int u = 0; u < 2; ++u) w2off[u] = sbase + (uint32_t)(r * 64 + (((2 * u + h) ^ ((r >> 2) & 3)) << 4));
    const uint32_t tabh = sbase + L_TAB + (uint32_t)h * 64;
    const uint32_t tabr = sbase + L_TAB + (uint32_t)r * 4;
    const bf16x8 onesB = __builtin_bit_cast(bf16x8, u32x4{h == 0 ? 0x3f803f80u : 0u, 0u, 0u, 0u});
    char* scr = smem + L_A + wave * 8192;
    {
        const float* xg = xio + (size_t)(row0 + (lane >> 3)) * 192 + 4 * (lane & 7);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int rt = 0; rt < 6; ++rt) {
                f32x4 tmp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tmp[j] = *reinterpret_cast<const f32x4*>(xg + (size_t)(32 * T + 8 * j) * 192 + 32 * rt);
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(scr + j * 1024 + lane * 16) = tmp[j];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(scr + r * 128 + (4 * h + m) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) R[T][rt][4 * m + e] = v[e];
                }
            }
    }
    typedef const __attribute__((address_space(1))) float* gptr_f;
    const gptr_f tabg = (gptr_f)bp.tab;
    // an MLP chunk = 24 pieces of 1 KB (< 12: the mlp.0 tile), six per wave (waves 0, 1: mlp.0; 2, 3: mlp.2)
    const int mw = wave * 6;
    auto dma_chunk_piece = [&](int c, int lds_off, int u) {
        const int pc = mw + u;
        bs_dma(bs_rsrc(bp.wmlp), smem + lds_off + pc * 1024, lane16, c * TILE + (wave < 2 ? pc : pc - 12) * 1024);
    };
    auto acc_from4 = [](f32x4 a, f32x4 b, f32x4 c, f32x4 d) {
        return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
    };

#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
        float tb[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) tb[i] = tabg[i * 256 + tid];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 6; ++i) lds_write_b32(sbase + L_TAB + (uint32_t)(i * 256 + tid) * 4, tb[i]);
        FENCE();
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_chunk_piece(0, L_A + 3 * CHUNK, u);
#pragma unroll
        for (int u = 0; u < 6; ++u) dma_chunk_piece(1, L_A + 4 * CHUNK, u);
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            LnStats st{};
#pragma unroll
            for (int rt = 0; rt < 6; ++rt) ln_stats_tile(st, R[T][rt]);
            float rstd, shift;
            ln_finish(st, rstd, shift);
#pragma unroll
            for (int rt = 0; rt < 6; ++rt) { tf[T][2 * rt] = ln_frag<0>(R[T][rt], rstd, shift); tf[T][2 * rt + 1] = ln_frag<1>(R[T][rt], rstd, shift); }
        }
        barrier_all();                                      // tables written, chunks 0 and 1 landed
        static_for<6>([&](auto rt_) {
            constexpr int rt = decltype(rt_)::value;
            const uint32_t bw = lds_read_b32_off(tabr, (T_B2 + rt * 32) * 4);
            lds_wait<0>();
            FENCE();
            const bf16x8 ba = __builtin_bit_cast(bf16x8, u32x4{bw, 0u, 0u, 0u});
            R[0][rt] = mfma32(ba, onesB, R[0][rt]);
            R[1][rt] = mfma32(ba, onesB, R[1][rt]);
        });
        FENCE();

        f32x16 acc1[2][2];            // [parity of the chunk][token tile]
        bf16x8 hfr[2][2][2];          // [parity][token tile][K-step]
        GeluState gs;
        int ri = 3;
        auto slot = [&](auto par_, auto f1_, auto ge_, auto f2_, auto dm_, int c) {
            constexpr int PAR = decltype(par_)::value;
            constexpr bool F1 = decltype(f1_)::value, GE = decltype(ge_)::value, F2 = decltype(f2_)::value, DM = decltype(dm_)::value;
            auto wrap = [](int v) { return v >= 6 ? v - 6 : v; };
            const int d0 = L_A + wrap(ri + 2) * CHUNK, d1 = L_A + wrap(ri + 3) * CHUNK;
            const uint32_t o1 = (uint32_t)(L_A + ri * CHUNK), o2 = (uint32_t)(L_A + wrap(ri + 4) * CHUNK + TILE);
            ri = wrap(ri + 1);
            uint32_t a1[4], a2[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) a1[t] = woff[t] + o1;
#pragma unroll
            for (int u = 0; u < 2; ++u) a2[u] = w2off[u] + o2;
            const uint32_t tbb = tabh + (uint32_t)((T_B1 + c * 32) * 4);
            constexpr int NM = (F1 && F2) ? 24 : 12;        // fragments of the slot; every fragment feeds two MFMAs
            constexpr int NG = 2 * NM;                      // MFMA gaps
            auto frag = [&](auto n_) {
                constexpr int n = decltype(n_)::value;
                constexpr bool is1 = F1 && (!F2 || (n & 1) == 0);
                constexpr int k = (F1 && F2) ? n / 2 : n;
#ifdef M64_NOLDS
                return __builtin_bit_cast(bf16x8, u32x4{a1[k & 3], a2[k & 1], (uint32_t)n, 0u});
#else
                if constexpr (is1) return lds_read_b128_asm_off(a1[k & 3], (k >> 2) * 4096);
                else return lds_read_b128_asm_off(a2[k & 1], (k >> 1) * 2048);
#endif
            };
            constexpr int LA = 3, RS = LA + 1;
            bf16x8 wq[RS];
            f32x4 b0, b1, b2, b3;
            if constexpr (F1) { b0 = lds_read_f4_off(tbb, 0); b1 = lds_read_f4_off(tbb, 16); b2 = lds_read_f4_off(tbb, 32); b3 = lds_read_f4_off(tbb, 48); }
            static_for<LA>([&](auto i_) { wq[decltype(i_)::value] = frag(i_); });
            lds_wait<LA - 1>();
            FENCE();
            if constexpr (F1) { acc1[PAR][0] = acc_from4(b0, b1, b2, b3); acc1[PAR][1] = acc1[PAR][0]; }
            static_for<NG>([&](auto j_) {
                constexpr int j = decltype(j_)::value, n = j >> 1, T = j & 1;
                constexpr bool is1 = F1 && (!F2 || (n & 1) == 0);
                constexpr int k = (F1 && F2) ? n / 2 : n;
                if constexpr (T == 0 && n > 0) { lds_wait<(NM - 1 - n < LA - 1 ? NM - 1 - n : LA - 1)>(); FENCE(); }
                if constexpr (is1) acc1[PAR][T] = mfma32(wq[n % RS], tf[T][k], acc1[PAR][T]);
                else R[T][k >> 1] = mfma32h(wq[n % RS], hfr[PAR][T][k & 1], R[T][k >> 1]);
#ifndef M64_NOGELU
                if constexpr (GE && j > 0) gelu2_pins<(2 * GELU_OPS * (j - 1)) / NG, (2 * GELU_OPS * j) / NG>(gs, hfr[1 - PAR]);
#endif
                if constexpr (T == 1 && n + LA < NM) wq[(n + LA) % RS] = frag(std::integral_constant<int, n + LA>{});
#ifndef M64_NOGELU
                if constexpr (GE) gelu2_ops<(2 * GELU_OPS * j) / NG, (2 * GELU_OPS * (j + 1)) / NG>(gs, acc1[1 - PAR], hfr[1 - PAR]);
#else
                if constexpr (GE && j == NG - 1) { asm volatile("" :: "v"(acc1[1 - PAR][0]), "v"(acc1[1 - PAR][1])); asm volatile("" : "+v"(hfr[1 - PAR][0][0]), "+v"(hfr[1 - PAR][0][1]), "+v"(hfr[1 - PAR][1][0]), "+v"(hfr[1 - PAR][1][1])); }
#endif
#ifndef M64_NODMA
                if constexpr (DM && j % (NG / 12) == NG / 12 - 1) {
                    constexpr int pi = j / (NG / 12);
                    if constexpr (pi < 6) dma_chunk_piece(c + 2, d0, pi); else dma_chunk_piece(c + 3, d1, pi - 6);
                }
#endif
#ifndef M64_NOGELU
                if constexpr (GE && j == NG - 1) gelu2_pins<(2 * GELU_OPS * j) / NG, 2 * GELU_OPS>(gs, hfr[1 - PAR]);
#endif
                FENCE();
            });
        };
        using T_ = std::true_type; using F_ = std::false_type;
        constexpr std::integral_constant<int, 0> P0{}; constexpr std::integral_constant<int, 1> P1{};
        slot(P0, T_{}, F_{}, F_{}, T_{}, 0);
        slot(P1, T_{}, T_{}, F_{}, F_{}, 1);
        barrier_all();
#pragma unroll 1
        for (int c = 2; c < 22; c += 2) {
            slot(P0, T_{}, T_{}, T_{}, T_{}, c); slot(P1, T_{}, T_{}, T_{}, F_{}, c + 1);
            barrier_all();
        }
        slot(P0, T_{}, T_{}, T_{}, F_{}, 22); slot(P1, T_{}, T_{}, T_{}, F_{}, 23);
        slot(P0, F_{}, T_{}, T_{}, F_{}, 24);
        slot(P1, F_{}, F_{}, T_{}, F_{}, 25);
    }
    __syncthreads();
    {
        float* xg = xio + (size_t)(row0 + (lane >> 3)) * 192 + 4 * (lane & 7);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int rt = 0; rt < 6; ++rt) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    *reinterpret_cast<f32x4*>(scr + r * 128 + (4 * h + m) * 16) = f32x4{R[T][rt][4 * m], R[T][rt][4 * m + 1], R[T][rt][4 * m + 2], R[T][rt][4 * m + 3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(scr + j * 1024 + lane * 16);
                    if (active) *reinterpret_cast<f32x4*>(xg + (size_t)(32 * T + 8 * j) * 192 + 32 * rt) = v;
                }
            }
    }
}

}  // namespace

// x fp32 [nwin * 64][192] in window order, in place, `reps` times over; operands: the w1 / w2 / tab tensors of pack_stream_block
extern "C" int tup_exp_mlp64(float* x, const void* w1, const void* w2, const void* tab, int nwin, int reps, void* stream)
{
    if (nwin <= 0 || reps <= 0) return 0;
    StreamBlock kb{};
    kb.w1 = (const char*)w1; kb.w2 = (const char*)w2; kb.tab = (const float*)tab;
    TUP_SET_DYN_LDS(mlp64_kernel, BS_LDS);
    mlp64_kernel<<<dim3((nwin + 3) / 4), dim3(M64_NT), BS_LDS, reinterpret_cast<hipStream_t>(stream)>>>(x, nwin, kb, reps);
    TUP_CHECK_LAUNCH();
    return 0;
}
