// Branch A of the FastTransformer in TRAINING through its exact algebraic composition (gfx950), r = 2:
//     upscaled_input = ReLU(conv3x3_{64->3, no bias}(PixelShuffle(2)(conv3x3_{64->256}(feat) + b)))      model.py:264-265, utils.py:32-40,62-63
// There is no non-linearity between the two convs (utils.py:50), so they compose into ONE 5x5-tap conv of feat with 12 outputs
// (3 channels x 4 sub-pixel phases; 16 non-zero taps per phase) -- the inference path has used that since round 1.  The same
// algebra serves the backward: with G = dL/dWc (the gradient w.r.t. the composed weights, a 12 x 25 x 64 tensor) and g12 = the
// phase-unshuffled gradient of the 3-channel HR image,
//     d feat  = 5x5 transposed conv of g12 with Wc                       (22.6 GF per image instead of 271.8 + 12.7)
//     G       = sum over pixels of g12 (x) feat windows                  (22.6 GF instead of 271.8 + 12.7)
//     dW_up, db_up, dW_3 = the chain rule through the composition (a few MFLOP),
// and the 64-channel HR tensor (472 MB at B = 4) and its gradient never exist.  The reference zero-pads that HR tensor for the
// 64->3 conv, so the 1-pixel HR border ring uses 8 weight variants without the taps that fall outside (rowmode / colmode of
// packing.compose_branch_a); ring pixels are excluded from the main kernels and handled by the two small ring kernels.
// Layouts: Wc [9 variants][12 n = c*4 + si*2 + sj][25 t = ty*5 + tx][64 ci]; feat / d feat NHWC bf16; g fp32 planar [B][3][2H][2W].
#include "common.h"
#include <type_traits>

namespace {

constexpr int NOUT = 12, NV = 9, NTAP = 25;
// G / Gb are accumulated with float atomics from every workgroup into the same ~170 k addresses: REP replicas (chosen by the
// workgroup id) cut the contention REP-fold (one replica: the ring kernel alone took 0.43 ms); the chain rule sums them
constexpr int REP = 16, GSZ = NV * NOUT * NTAP * 64, GBSZ = NV * NOUT;

TUP_DEVICE bool dropped(int mode, int d) { return (mode == 1 && d == 0) || (mode == 2 && d == 2); }

// ---- composition: one thread per (variant, n, tap, ci) ----
__global__ __launch_bounds__(256) void bra_compose_kernel(const float* __restrict__ wu, const float* __restrict__ bu,
                                                          const float* __restrict__ w3, bf16_t* __restrict__ wv, float* __restrict__ bv,
                                                          bf16_t* __restrict__ wp, bf16_t* __restrict__ wd)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NV * NOUT * NTAP * 64) return;
    const int ci = idx & 63;
    int q = idx >> 6;
    const int t = q % NTAP; q /= NTAP;
    const int n = q % NOUT, v = q / NOUT;
    const int rm = v / 3, cm = v % 3;
    const int o3 = n >> 2, si = (n >> 1) & 1, sj = n & 1, ty = t / 5, tx = t % 5;
    float acc = 0.f, bacc = 0.f;
    for (int dy = 0; dy < 3; ++dy) {
        if (dropped(rm, dy)) continue;
        const int qy = si + dy - 1, oy = qy < 0 ? -1 : (qy >> 1), si2 = qy - 2 * oy, ky = ty - 1 - oy;
        for (int dx = 0; dx < 3; ++dx) {
            if (dropped(cm, dx)) continue;
            const int qx = sj + dx - 1, ox = qx < 0 ? -1 : (qx >> 1), sj2 = qx - 2 * ox, kx = tx - 1 - ox;
            const float* w3p = w3 + ((size_t)o3 * 64 * 3 + dy) * 3 + dx;            // + ch * 9
            if (t == 0 && ci == 0)
                for (int ch = 0; ch < 64; ++ch) bacc += w3p[ch * 9] * bu[ch * 4 + si2 * 2 + sj2];
            if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
            const float* wup = wu + ((size_t)(si2 * 2 + sj2) * 64 + ci) * 9 + ky * 3 + kx;      // + ch * 4 * 64 * 9
            for (int ch = 0; ch < 64; ++ch) acc += w3p[ch * 9] * wup[(size_t)ch * 4 * 64 * 9];
        }
    }
    wv[idx] = f32_to_bf16(acc);
    if (t == 0 && ci == 0) bv[v * NOUT + n] = bacc;
    if (v == 0) {
        wp[((size_t)t * 16 + n) * 64 + ci] = f32_to_bf16(acc);                       // forward main kernel: [tap][16 rows][64]
        // input-gradient kernel: A operand rows = feat channels (row ct*16 + 4g + e = channel g*16 + ct*4 + e), K = [tap'][16 n],
        // tap' = 24 - t (the transposed conv walks the taps backwards), stored as [13 k-steps][64 rows][32 k]
        const int rho = ((ci >> 2) & 3) * 16 + (ci >> 4) * 4 + (ci & 3);
        const int kk = (24 - t) * 16 + n;
        wd[((size_t)(kk >> 5) * 64 + rho) * 32 + (kk & 31)] = f32_to_bf16(acc);
    }
}

// ---- g12m[b][y][x][16] bf16 = phase-unshuffled (g * [ui > 0]) with the HR border ring zeroed, rows 12..15 zero;
//      gb0[n] += column sums (bias gradient of the composed conv, interior variant) ----
__global__ __launch_bounds__(256) void bra_prep_kernel(const float* __restrict__ g, const float* __restrict__ ui, bf16_t* __restrict__ g12,
                                                       float* __restrict__ gb0, int B, int H, int W)
{
    __shared__ float red[4][NOUT];
    const int Hs = 2 * H, Ws = 2 * W;
    const long long total = (long long)B * H * W;
    float s[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) s[n] = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int x = (int)(p % W);
        const long long t = p / W;
        const int y = (int)(t % H), b = (int)(t / H);
        float v[NOUT];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const int Y = 2 * y + si;
                const size_t off = (((size_t)b * 3 + c) * Hs + Y) * Ws + 2 * x;
                const f32x2 gv = *reinterpret_cast<const f32x2*>(g + off), uv = *reinterpret_cast<const f32x2*>(ui + off);
                const bool ringrow = Y == 0 || Y == Hs - 1;
                v[c * 4 + si * 2 + 0] = (uv[0] > 0.f && !ringrow && x != 0) ? gv[0] : 0.f;
                v[c * 4 + si * 2 + 1] = (uv[1] > 0.f && !ringrow && x != W - 1) ? gv[1] : 0.f;
            }
        u32x4 lo, hi;
        lo[0] = pack_bf16x2(v[0], v[1]); lo[1] = pack_bf16x2(v[2], v[3]); lo[2] = pack_bf16x2(v[4], v[5]); lo[3] = pack_bf16x2(v[6], v[7]);
        hi[0] = pack_bf16x2(v[8], v[9]); hi[1] = pack_bf16x2(v[10], v[11]); hi[2] = 0u; hi[3] = 0u;
        u32x4* d = reinterpret_cast<u32x4*>(g12 + (size_t)p * 16);
        d[0] = lo; d[1] = hi;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) s[n] += v[n];
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        float t = s[n];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][n] = t;
    }
    __syncthreads();
    if (threadIdx.x < NOUT) atomicAdd(gb0 + (blockIdx.x % REP) * GBSZ + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- input gradient, main part: d feat[y][x][ci] = sum_{t', n} wd[ci][t'][n] g12[y + uy][x + ux][n], (uy, ux) = (t'/5 - 2, t'%5 - 2).
// Implicit GEMM on MFMA 16x16x32: A = wd rows (LDS-resident for the whole persistent workgroup), B = pixels, K-step = two taps
// x 16 channels read straight out of the haloed g tile (32 B per pixel).  A wave owns two tile rows (64 pixels) x all 64 channels. ----
constexpr int DTH = 8, DTW = 32, DHW = DTW + 4, DHH = DTH + 4, DNPIX = DHH * DHW;          // 12 x 36 = 432 haloed pixels
constexpr int DG_BYTES = DNPIX * 32;                                                      // 13,824
constexpr int DW_BYTES = 13 * 64 * 64;                                                    // 53,248
constexpr int DGRAD_LDS = DW_BYTES + DG_BYTES;

__global__ __launch_bounds__(256, 2) void bra_dgrad_kernel(const bf16_t* __restrict__ g12, const bf16_t* __restrict__ wd,
                                                           bf16_t* __restrict__ out, int B, int H, int W, int tilesX, int tilesY)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w_lds = smem;
    char* g_lds = smem + DW_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    // weights: [ks][row][4 chunks of 16 B], chunk XOR-swizzled by (row >> 2) & 3 (rows 64 B apart: 4-way conflicts otherwise)
    for (int i = tid; i < 13 * 64 * 4; i += 256) {
        const int ch = i & 3, row = (i >> 2) & 63, ks = i >> 8;
        *reinterpret_cast<u32x4*>(w_lds + (ks * 64 + row) * 64 + ((ch ^ ((row >> 2) & 3)) << 4)) =
            *reinterpret_cast<const u32x4*>(wd + ((size_t)(ks * 64 + row) * 32 + ch * 8));
    }
    const int ntiles = tilesX * tilesY * B;
    constexpr int GP = (DNPIX * 2 + 255) / 256;                 // 16-byte pieces per thread (864 / 256 -> 4)
    u32x4 gpre[GP];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY, b = t / tilesY;
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int idx = tid + u * 256, q = idx >> 1, half = idx & 1;
            const int yy = q / DHW, xx = q - yy * DHW;
            const int iy = ty * DTH - 2 + yy, ix = tx * DTW - 2 + xx;
            gpre[u] = u32x4{0u, 0u, 0u, 0u};
            if (idx < DNPIX * 2 && iy >= 0 && iy < H && ix >= 0 && ix < W)
                gpre[u] = *reinterpret_cast<const u32x4*>(g12 + (((size_t)b * H + iy) * W + ix) * 16 + half * 8);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    const uint32_t gbase = lds_addr(g_lds), wbase = lds_addr(w_lds);
    const int tsel = g >> 1, half = g & 1;
    const uint32_t a_off = (uint32_t)(pl * 64 + ((g ^ ((pl >> 2) & 3)) << 4));          // + ks * 4096 + ct * 1024
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                    // everyone is done reading the previous g tile (and the weights landed)
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int idx = tid + u * 256;
            if (idx < DNPIX * 2) *reinterpret_cast<u32x4*>(g_lds + idx * 16) = gpre[u];
        }
        __syncthreads();
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY, b = t / tilesY;
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        f32x4 acc[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // this lane's pixel of pixel tile pt = (r2, xh): tile row 2*wave + r2, column 16*xh + pl; halo origin (-2, -2)
        uint32_t pbase[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
            pbase[pt] = gbase + (uint32_t)(((2 * wave + (pt >> 1) + 2) * DHW + 16 * (pt & 1) + pl + 2) * 32 + half * 16);
#pragma unroll
        for (int ks = 0; ks < 13; ++ks) {
            // lanes with tsel = 0 read tap 2ks, the others tap 2ks + 1 (tap 25 does not exist: its weights are zero, any address does)
            const int t0 = 2 * ks, t1 = (2 * ks + 1 < 25) ? 2 * ks + 1 : 24;
            const int o0 = ((t0 / 5 - 2) * DHW + (t0 % 5 - 2)) * 32, o1 = ((t1 / 5 - 2) * DHW + (t1 % 5 - 2)) * 32;
            const int off = tsel ? o1 : o0;
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                af[ct] = *reinterpret_cast<const bf16x8*>(w_lds + ks * 4096 + ct * 1024 + a_off);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
                bfr[pt] = *reinterpret_cast<const bf16x8*>(g_lds + (pbase[pt] - gbase) + off);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = mfma16x16x32(af[ct], bfr[pt], acc[ct][pt]);
        }
        // lane (pl, g) holds channels g*16 + ct*4 + e of its pixel: 32 contiguous bytes
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int oy = ty * DTH + 2 * wave + (pt >> 1), ox = tx * DTW + 16 * (pt & 1) + pl;
            if (oy < H && ox < W) {
                u32x4 lo, hi;
                lo[0] = pack_bf16x2(acc[0][pt][0], acc[0][pt][1]); lo[1] = pack_bf16x2(acc[0][pt][2], acc[0][pt][3]);
                lo[2] = pack_bf16x2(acc[1][pt][0], acc[1][pt][1]); lo[3] = pack_bf16x2(acc[1][pt][2], acc[1][pt][3]);
                hi[0] = pack_bf16x2(acc[2][pt][0], acc[2][pt][1]); hi[1] = pack_bf16x2(acc[2][pt][2], acc[2][pt][3]);
                hi[2] = pack_bf16x2(acc[3][pt][0], acc[3][pt][1]); hi[3] = pack_bf16x2(acc[3][pt][2], acc[3][pt][3]);
                u32x4* d = reinterpret_cast<u32x4*>(out + (((size_t)b * H + oy) * W + ox) * 64 + g * 16);
                d[0] = lo; d[1] = hi;
            }
        }
    }
}

// ring enumeration shared by the two ring kernels: HR ring pixel k of an image (per_img = 2*Ws + 2*(Hs - 2))
TUP_DEVICE void ring_pixel(int k, int Hs, int Ws, int& Y, int& X) {
    if (k < Ws) { Y = 0; X = k; }
    else if (k < 2 * Ws) { Y = Hs - 1; X = k - Ws; }
    else { k -= 2 * Ws; Y = 1 + (k >> 1); X = (k & 1) ? Ws - 1 : 0; }
}
TUP_DEVICE int variant_of(int Y, int X, int Hs, int Ws) {
    return ((Y == 0) ? 1 : (Y == Hs - 1 ? 2 : 0)) * 3 + ((X == 0) ? 1 : (X == Ws - 1 ? 2 : 0));
}

// ---- input gradient, ring part: the LR pixels within 2 of the image border receive, on top of the main kernel's result, the
// contributions of the HR ring pixels through their weight variants.  One wave per LR frame pixel, lane = channel. ----
__global__ __launch_bounds__(256) void bra_dgrad_ring_kernel(const float* __restrict__ g, const float* __restrict__ ui,
                                                             const bf16_t* __restrict__ wv, bf16_t* __restrict__ out, int B, int H, int W)
{
    const int lane = threadIdx.x & 63;
    const int Hs = 2 * H, Ws = 2 * W;
    // frame pixels: rows 0..2 and H-3..H-1 in full, then columns 0..2 and W-3..W-1 of the remaining rows (H, W >= 6 assumed by the host)
    const int nrow = 6 * W, ncol = 6 * (H - 6), per_img = nrow + ncol;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)per_img * B) return;
    const int b = (int)(gid / per_img);
    int k = (int)(gid - (long long)b * per_img);
    int fy, fx;
    if (k < nrow) { const int rr = k / W; fy = rr < 3 ? rr : H - 6 + rr; fx = k - rr * W; }
    else { k -= nrow; const int rr = k / 6, cc = k - rr * 6; fy = 3 + rr; fx = cc < 3 ? cc : W - 6 + cc; }
    float acc = 0.f;
    for (int ly = max(fy - 2, 0); ly <= min(fy + 2, H - 1); ++ly) {
        if (ly != 0 && ly != H - 1 && fx - 2 > 0 && fx + 2 < W - 1) continue;         // no ring pixel in this LR row within reach
        for (int lx = max(fx - 2, 0); lx <= min(fx + 2, W - 1); ++lx) {
            const int tap = (fy - ly + 2) * 5 + (fx - lx + 2);
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int Y = 2 * ly + (sp >> 1), X = 2 * lx + (sp & 1);
                const int v = variant_of(Y, X, Hs, Ws);
                if (v == 0) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const size_t off = (((size_t)b * 3 + c) * Hs + Y) * Ws + X;
                    const float gv = ui[off] > 0.f ? g[off] : 0.f;
                    acc += gv * bf16_to_f32(wv[(((size_t)v * NOUT + c * 4 + sp) * NTAP + tap) * 64 + lane]);
                }
            }
        }
    }
    bf16_t* o = out + (((size_t)b * H + fy) * W + fx) * 64 + lane;
    *o = f32_to_bf16(bf16_to_f32(*o) + acc);
}

// ---- weight gradient, main part: G0[n][t][ci] += sum_px g12[px][n] feat[px + (ty-2, tx-2)][ci] (ring pixels are zero in g12).
// MFMA 16x16x32 with both operands transposed out of LDS (pixels are the contraction; one K-step = one tile row of 32 pixels),
// the scheme of conv3x3_wgrad_thin_kernel (conv_bwd.hip): wave = input-channel tile with all 25 taps in registers across the
// persistent tile loop; the loop runs over the twelve HALO rows -- the five dx fragments of halo row hy serve the taps (dy, dx) of
// output rows hy - dy against a window of five rows' g fragments, so every x fragment is read once per tile (60 instead of 200) --
// and the swizzled read addresses come from lane constants (a halo row is 36 pixels = 18 pairs: row hy shifts the phase by 2 hy).
// (Until round 4: the four waves split the taps on the 16x16x16 form and rebuilt every address from the pixel index: 400 MFMAs,
// 416 transposed reads and ~1,800 vector instructions per tile and wave, 365-379 us at 4 x 720p.) ----
constexpr int WTH = 8, WTW = 32, WHW = WTW + 4, WHH = WTH + 4, WNPIX = WHH * WHW;          // 432
constexpr int WX_BYTES = WNPIX * 128;                                                     // 55,296
constexpr int WGRAD_LDS = WX_BYTES + 256 * 32;

TUP_DEVICE s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__global__ __launch_bounds__(512, 1) void bra_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g12,
                                                           float* __restrict__ G0, int B, int H, int W, int tilesX, int tilesY)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* x_lds = smem;
    char* g_lds = smem + WX_BYTES;                              // [256 pixels][16 n] bf16
    // eight waves: input-channel tile cit = wave & 3, tap rows dy 0..2 (waves 0-3) or 3..4 (waves 4-7; one wave of each per SIMD).
    // (Four waves with all 25 taps each: 100 accumulator + 56 staging registers, hipcc spills 176-376 B per lane.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cit = wave & 3;
    const int half = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int g = lane >> 4, l16 = lane & 15;
    const int trq = l16 >> 2, trp = l16 & 3;
    f32x4 acc[15];
#pragma unroll
    for (int t = 0; t < 15; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int XP = (WNPIX * 8 + 511) / 512;                 // 7
    u32x4 xpre[XP], gpre;
    const int ntiles = tilesX * tilesY * B;
    const bool last_ok = tid + (XP - 1) * 512 < WNPIX * 8;
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY, b = t / tilesY;
        const bf16_t* xb = x + (size_t)b * H * W * 64;
        const int ty0 = ty * WTH, tx0 = tx * WTW;
        if (ty0 >= 2 && ty0 + WTH + 2 <= H && tx0 >= 2 && tx0 + WTW + 2 <= W) {        // interior tile: offsets from the thread index
            const bf16_t* xt = xb + ((size_t)(ty0 - 2) * W + (tx0 - 2)) * 64;
            const int tq = (int)opaque_copy((uint32_t)tid);                             // (not hoisted: the offsets would cost the K loop its registers)
#pragma unroll
            for (int u = 0; u < XP; ++u) {
                const int idx = min(tq + u * 512, WNPIX * 8 - 1);
                const int q = idx >> 3, c = idx & 7;
                const int yy = (q * 911) >> 15, xx = q - yy * WHW;                      // q / 36 for q < 432
                xpre[u] = u32x4{0u, 0u, 0u, 0u};
                if (u + 1 < XP || last_ok) xpre[u] = *reinterpret_cast<const u32x4*>(xt + (yy * W + xx) * 64 + c * 8);
            }
            // g tile: 256 pixels x 32 B = 512 pieces of 16 B, one per thread
            gpre = *reinterpret_cast<const u32x4*>(g12 + (((size_t)b * H + ty0 + (tq >> 6)) * W + tx0 + ((tq >> 1) & 31)) * 16 + (tq & 1) * 8);
            return;
        }
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int idx = tid + u * 512, q = idx >> 3, c = idx & 7;
            const int yy = q / WHW, xx = q - yy * WHW;
            const int iy = ty0 - 2 + yy, ix = tx0 - 2 + xx;
            xpre[u] = u32x4{0u, 0u, 0u, 0u};
            if (idx < WNPIX * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)
                xpre[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * W + ix) * 64 + c * 8);
        }
        const int oy = ty0 + (tid >> 6), ox = tx0 + ((tid >> 1) & 31);
        gpre = u32x4{0u, 0u, 0u, 0u};
        if (oy < H && ox < W) gpre = *reinterpret_cast<const u32x4*>(g12 + (((size_t)b * H + oy) * W + ox) * 16 + (tid & 1) * 8);
    };
    // lane constants of the fragment reads.  g rows are 32 bytes (16 n), not swizzled: pixel ry * 32 + 8g + trq (+ 4)
    const int gb0 = (8 * g + trq) * 32 + trp * 8;
    // x: pixel q = hy * 36 + 8g + trq + dx + 4h, logical chunk xc, byte (xcol & 7) * 2 inside it
    const int xcol = 16 * cit + 4 * trp, xc = xcol >> 3;
    int xs[5][2], xb0[5][2];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q0 = 8 * g + trq + dx + 4 * h;
            xs[dx][h] = q0 >> 1;
            xb0[dx][h] = q0 * 128 + (xcol & 7) * 2;
        }
    auto join = [](s16x4 lo, s16x4 hi) {
        const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
        return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int idx = tid + u * 512;
            if (idx < WNPIX * 8) *reinterpret_cast<u32x4*>(x_lds + swz128(idx >> 3, idx & 7)) = xpre[u];
        }
        reinterpret_cast<u32x4*>(g_lds)[tid] = gpre;
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        uint32_t xst[5][2];              // through an empty asm per tile: left visible, hipcc hoists the read addresses out of the tile loop
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
            for (int h = 0; h < 2; ++h) xst[dx][h] = opaque_copy((uint32_t)xs[dx][h]);
        // tap rows DY0 .. DY0 + NDY - 1: halo rows DY0 .. DY0 + NDY + 6; output row ry pairs with halo row ry + dy
        auto taps = [&](auto dy0c, auto ndyc) {
            constexpr int DY0 = decltype(dy0c)::value, NDY = decltype(ndyc)::value;
            bf16x8 gf[NDY];              // g fragments of output rows (hy - DY0) .. (hy - DY0 - NDY + 1), slot = row % NDY: A[n = l16][k = pixel 8g + j]
#pragma unroll
            for (int hy = DY0; hy < DY0 + NDY + WTH - 1; ++hy) {
                const int rn = hy - DY0;
                if (rn < WTH) gf[rn % NDY] = join(lds_tr16(g_lds + gb0 + rn * 1024), lds_tr16(g_lds + gb0 + 128 + rn * 1024));
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) {
                    s16x4 part[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ph = (int)((opaque_copy(xst[dx][h]) + 2 * hy) & 7u);          // (opaque per row: the addresses stay in their row)
                        part[h] = lds_tr16(x_lds + xb0[dx][h] + ((xc ^ ph) << 4) + hy * (WHW * 128));
                    }
                    const bf16x8 bfr = join(part[0], part[1]);
#pragma unroll
                    for (int d = 0; d < NDY; ++d) {
                        const int ry = hy - DY0 - d;
                        if (ry >= 0 && ry < WTH) acc[d * 5 + dx] = mfma16x16x32(gf[ry % NDY], bfr, acc[d * 5 + dx]);
                    }
                }
                asm volatile("" ::: "memory");          // a halo row at a time: left free, hipcc requests every row's fragments up front and spills
            }
        };
        if (half == 0) taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
        else taps(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
    }
    // D[row = n 4g+e][col = ci l16]; rows 12..15 are padding
    if (g < 3) {
        const int ntl = half == 0 ? 15 : 10, tap0 = half == 0 ? 0 : 15;
#pragma unroll
        for (int t = 0; t < 15; ++t)
            if (t < ntl)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    atomicAdd(G0 + (size_t)(blockIdx.x % REP) * GSZ + ((size_t)(4 * g + e) * NTAP + tap0 + t) * 64 + 16 * cit + l16, acc[t][e]);
    }
}

// ---- weight gradient, ring part: G[v][n][t][ci] (v = 1..8) and Gb[v][n] from the HR ring pixels.  A workgroup owns a run of
// RING_RUN ring pixels (thread = channel x tap group); partial sums per (variant-of-pixel, n, tap) go out as atomics -- the
// run is cut so that all its pixels share one variant.  32-pixel runs, four pixels per trip: ~1,000 workgroups at 4 x 720p.  (Measured:
// 128-pixel runs with eight pixels per trip -- a quarter of the atomics -- 492 us instead of 276: the serial length of a workgroup
// weighs more than the atomics it saves.) ----
constexpr int RING_RUN = 32, RING_PX = 4;
__global__ __launch_bounds__(256) void bra_wgrad_ring_kernel(const float* __restrict__ g, const float* __restrict__ ui,
                                                             const bf16_t* __restrict__ x, float* __restrict__ G, float* __restrict__ Gb,
                                                             int B, int H, int W)
{
    const int Hs = 2 * H, Ws = 2 * W;
    const int per_img = 2 * Ws + 2 * (Hs - 2);
    // the tap group is the wave index (scalar): pixel coordinates, variants and tap addresses are wave-uniform; only `ci` is per lane.
    // Timing split of the 276 us at 4 x 720p: 152 us without the flush atomics (10.7 M of them), i.e. 124 us of atomics.
    const int ci = threadIdx.x & 63, tg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntap = (tg == 0) ? 7 : 6;
    const int k0 = blockIdx.x * RING_RUN, nring = per_img * B;          // (the host checks that the ring fits 31 bits)
    float acc[7][3];
    int cur_v = -1, cur_sp = -1;
    auto flush = [&]() {
        if (cur_v < 0) return;
#pragma unroll
        for (int a = 0; a < 7; ++a)
            if (a < ntap)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    atomicAdd(G + (size_t)(blockIdx.x % REP) * GSZ + (((size_t)cur_v * NOUT + c * 4 + cur_sp) * NTAP + tg + 4 * a) * 64 + ci, acc[a][c]);
    };
    // two passes over the run, one per sub-pixel phase parity along the run, so that (variant, phase) changes rarely; the pixels go
    // RING_PX at a time with all their loads issued before the first use (one pixel per iteration paid a global round trip each)
    for (int pass = 0; pass < 2; ++pass) {
        for (int i0 = 0; i0 < RING_RUN; i0 += RING_PX) {
            int vv[RING_PX], spp[RING_PX];
            float gv[RING_PX][3], f[RING_PX][7];
            bool ok[RING_PX];
#pragma unroll
            for (int j = 0; j < RING_PX; ++j) {
                const int gid = k0 + i0 + j;
                ok[j] = gid < nring;
                const int b = ok[j] ? gid / per_img : 0;
                int Y, X;
                ring_pixel(ok[j] ? gid - b * per_img : 0, Hs, Ws, Y, X);
                ok[j] = ok[j] && (((Y + X) & 1) == pass);
                spp[j] = (Y & 1) * 2 + (X & 1);
                vv[j] = variant_of(Y, X, Hs, Ws);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const size_t off = (((size_t)b * 3 + c) * Hs + Y) * Ws + X;
                    gv[j][c] = ui[off] > 0.f ? g[off] : 0.f;
                }
                const int ly = Y >> 1, lx = X >> 1;
#pragma unroll
                for (int a = 0; a < 7; ++a) {
                    const int tap = tg + 4 * a;
                    const int iy = ly + tap / 5 - 2, ix = lx + tap % 5 - 2;
                    f[j][a] = (a < ntap && iy >= 0 && iy < H && ix >= 0 && ix < W)
                                  ? bf16_to_f32(x[(((size_t)b * H + iy) * W + ix) * 64 + ci]) : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < RING_PX; ++j) {
                if (!ok[j]) continue;
                if (vv[j] != cur_v || spp[j] != cur_sp) {
                    flush();
                    cur_v = vv[j]; cur_sp = spp[j];
#pragma unroll
                    for (int a = 0; a < 7; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[a][c] = 0.f;
                }
                if (threadIdx.x < 3) atomicAdd(Gb + (blockIdx.x % REP) * GBSZ + vv[j] * NOUT + threadIdx.x * 4 + spp[j], gv[j][threadIdx.x]);
#pragma unroll
                for (int a = 0; a < 7; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[a][c] = fmaf(gv[j][c], f[j][a], acc[a][c]);
            }
        }
    }
    flush();
}

// ---- chain rule: (G [9][12][25][64], Gb [9][12]) -> dW_up [256][64][3][3], db_up [256], dW_3 [3][64][3][3] ----
// dM[o3][dy][dx][si2][sj2][ky][kx][ci] = sum over variants keeping (dy, dx) and over the (si, sj) that reach (si2, sj2) of G
__global__ __launch_bounds__(256) void bra_chain_dm_kernel(const float* __restrict__ G, const float* __restrict__ Gb,
                                                           float* __restrict__ dM, float* __restrict__ dMb)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;                 // [o3 3][dy 3][dx 3][si2 2][sj2 2][ky 3][kx 3][ci 64]
    if (idx >= 3 * 9 * 4 * 9 * 64) return;
    const int ci = idx & 63;
    int q = idx >> 6;
    const int kx = q % 3; q /= 3;
    const int ky = q % 3; q /= 3;
    const int sj2 = q & 1, si2 = (q >> 1) & 1; q >>= 2;
    const int dx = q % 3; q /= 3;
    const int dy = q % 3, o3 = q / 3;
    float s = 0.f, sb = 0.f;
    for (int oy = -1; oy <= 1; ++oy) {
        const int si = si2 + 2 * oy - dy + 1;
        if (si < 0 || si > 1) continue;
        for (int ox = -1; ox <= 1; ++ox) {
            const int sj = sj2 + 2 * ox - dx + 1;
            if (sj < 0 || sj > 1) continue;
            const int n = o3 * 4 + si * 2 + sj, t = (oy + 1 + ky) * 5 + (ox + 1 + kx);
            for (int v = 0; v < NV; ++v) {
                if (dropped(v / 3, dy) || dropped(v % 3, dx)) continue;
                for (int rp = 0; rp < REP; ++rp) s += G[(size_t)rp * GSZ + (((size_t)v * NOUT + n) * NTAP + t) * 64 + ci];
                if (ci == 0 && ky == 0 && kx == 0)
                    for (int rp = 0; rp < REP; ++rp) sb += Gb[rp * GBSZ + v * NOUT + n];
            }
        }
    }
    dM[idx] = s;
    if (ci == 0 && ky == 0 && kx == 0) dMb[((o3 * 3 + dy) * 3 + dx) * 4 + si2 * 2 + sj2] = sb;
}

// dW_up[(ch, si2, sj2)][ci][ky][kx] = sum_{o3, dy, dx} w3[o3][ch][dy][dx] dM[...];  db_up likewise from dMb
__global__ __launch_bounds__(256) void bra_chain_wu_kernel(const float* __restrict__ dM, const float* __restrict__ dMb,
                                                           const float* __restrict__ w3, float* __restrict__ dwu, float* __restrict__ dbu)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;                 // [row 256][ci 64][ky 3][kx 3]
    if (idx >= 256 * 64 * 9) return;
    const int kx = idx % 3, ky = (idx / 3) % 3, ci = (idx / 9) & 63, row = idx / (9 * 64);
    const int ch = row >> 2, si2 = (row >> 1) & 1, sj2 = row & 1;
    float s = 0.f, sb = 0.f;
    for (int o3 = 0; o3 < 3; ++o3)
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const float w = w3[((o3 * 64 + ch) * 3 + dy) * 3 + dx];
                s += w * dM[(((((((size_t)o3 * 3 + dy) * 3 + dx) * 2 + si2) * 2 + sj2) * 3 + ky) * 3 + kx) * 64 + ci];
                if (ci == 0 && ky == 0 && kx == 0) sb += w * dMb[((o3 * 3 + dy) * 3 + dx) * 4 + si2 * 2 + sj2];
            }
    dwu[idx] = s;
    if (ci == 0 && ky == 0 && kx == 0) dbu[row] = sb;
}

// dW_3[o3][ch][dy][dx] = sum_{si2, sj2, ky, kx, ci} dM[...] W_up[(ch, si2, sj2)][ci][ky][kx] + sum_{si2, sj2} dMb[...] b_up[(ch, si2, sj2)]
// one wave per output element, lanes = ci
__global__ __launch_bounds__(256) void bra_chain_w3_kernel(const float* __restrict__ dM, const float* __restrict__ dMb,
                                                           const float* __restrict__ wu, const float* __restrict__ bu, float* __restrict__ dw3)
{
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);              // [o3 3][ch 64][dy 3][dx 3]
    if (o >= 3 * 64 * 9) return;
    const int dx = o % 3, dy = (o / 3) % 3, ch = (o / 9) & 63, o3 = o / (9 * 64);
    float s = 0.f;
    for (int sp = 0; sp < 4; ++sp)
        for (int k = 0; k < 9; ++k)
            s += dM[((((((size_t)o3 * 3 + dy) * 3 + dx) * 4 + sp) * 9 + k)) * 64 + lane] * wu[((size_t)(ch * 4 + sp) * 64 + lane) * 9 + k];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) {
        for (int sp = 0; sp < 4; ++sp) s += dMb[((o3 * 3 + dy) * 3 + dx) * 4 + sp] * bu[ch * 4 + sp];
        dw3[o] = s;
    }
}

int persistent_blocks(long long ntiles)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long long cap = 2LL * cus;
    return (int)(ntiles < cap ? ntiles : cap);
}

}  // namespace

// Composition of the last Upsampler stage (r = 2) with up1_conv for the training path; every output is written by the kernel
// (wp rows 12..15 and wd's 26th tap must be zero: the caller passes zero-initialised buffers).
// wu fp32 [256][64][3][3], bu fp32 [256], w3 fp32 [3][64][3][3]  ->  wv bf16 [9][12][25][64], bv fp32 [9][12] (all variants),
// wp bf16 [25][16][64] (forward main kernel, variant 0), wd bf16 [13][64][32] (input-gradient kernel, variant 0).
extern "C" int tup_bra_compose(const float* wu, const float* bu, const float* w3, void* wv, float* bv, void* wp, void* wd, void* stream)
{
    const int total = NV * NOUT * NTAP * 64;
    bra_compose_kernel<<<dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        wu, bu, w3, (bf16_t*)wv, bv, (bf16_t*)wp, (bf16_t*)wd);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Backward of the composed branch A (r = 2).  g fp32 [B][3][2H][2W] = gradient w.r.t. upscaled_input (before its ReLU mask),
// ui fp32 [B][3][2H][2W] = upscaled_input (the ReLU mask is ui > 0), feat bf16 [B][H][W][64].
// Outputs: dfeat bf16 [B][H][W][64] (written), G fp32 [16 replicas][9][12][25][64] and Gb fp32 [16][9][12] (ACCUMULATED: zero
// them first; the gradient is the sum over the replicas, taken by tup_bra_chain),
// g12 bf16 [B][H][W][16] workspace.  H, W >= 6.
extern "C" int tup_bra_backward(const float* g, const float* ui, const void* feat, const void* wd, const void* wv,
                                void* g12, void* dfeat, float* G, float* Gb, int B, int H, int W, void* stream)
{
    if (B <= 0) return 0;
    if (H < 6 || W < 6) return (int)hipErrorInvalidValue;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long npix = (long long)B * H * W;
    long long pb = (npix + 255) / 256;
    if (pb > 2048) pb = 2048;
    bra_prep_kernel<<<dim3((unsigned)pb), dim3(256), 0, s>>>(g, ui, (bf16_t*)g12, Gb, B, H, W);
    TUP_CHECK_LAUNCH();
    const int tilesX = (W + DTW - 1) / DTW, tilesY = (H + DTH - 1) / DTH;
    const long long nt = (long long)tilesX * tilesY * B;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    TUP_SET_DYN_LDS((bra_dgrad_kernel), DGRAD_LDS);
    bra_dgrad_kernel<<<dim3(persistent_blocks(nt)), dim3(256), DGRAD_LDS, s>>>((const bf16_t*)g12, (const bf16_t*)wd, (bf16_t*)dfeat,
                                                                             B, H, W, tilesX, tilesY);
    TUP_CHECK_LAUNCH();
    const long long nframe = (long long)B * (6LL * W + 6LL * (H - 6));
    bra_dgrad_ring_kernel<<<dim3((unsigned)((nframe + 3) / 4)), dim3(256), 0, s>>>(g, ui, (const bf16_t*)wv, (bf16_t*)dfeat, B, H, W);
    TUP_CHECK_LAUNCH();
    TUP_SET_DYN_LDS((bra_wgrad_kernel), WGRAD_LDS);
    bra_wgrad_kernel<<<dim3((unsigned)(nt < 256 ? nt : 256)), dim3(512), WGRAD_LDS, s>>>((const bf16_t*)feat, (const bf16_t*)g12, G, B, H, W, tilesX, tilesY);
    TUP_CHECK_LAUNCH();
    const long long nring = (long long)B * (4LL * W + 2LL * (2 * H - 2));
    if (nring > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    bra_wgrad_ring_kernel<<<dim3((unsigned)((nring + RING_RUN - 1) / RING_RUN)), dim3(256), 0, s>>>(g, ui, (const bf16_t*)feat, G, Gb, B, H, W);
    TUP_CHECK_LAUNCH();
    return 0;
}

// Chain rule through the composition: G, Gb (tup_bra_backward) -> dwu fp32 [256][64][3][3], dbu fp32 [256], dw3 fp32 [3][64][3][3]
// (all written); dM fp32 [3*9*4*9*64] and dMb fp32 [108] are workspaces.
extern "C" int tup_bra_chain(const float* G, const float* Gb, const float* wu, const float* bu, const float* w3,
                             float* dM, float* dMb, float* dwu, float* dbu, float* dw3, void* stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bra_chain_dm_kernel<<<dim3((3 * 9 * 4 * 9 * 64 + 255) / 256), dim3(256), 0, s>>>(G, Gb, dM, dMb);
    TUP_CHECK_LAUNCH();
    bra_chain_wu_kernel<<<dim3((256 * 64 * 9 + 255) / 256), dim3(256), 0, s>>>(dM, dMb, w3, dwu, dbu);
    TUP_CHECK_LAUNCH();
    bra_chain_w3_kernel<<<dim3((3 * 64 * 9 + 3) / 4), dim3(256), 0, s>>>(dM, dMb, wu, bu, dw3);
    TUP_CHECK_LAUNCH();
    return 0;
}
