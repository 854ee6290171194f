// PyRatCNN evaluator (alpharat/nn/models/cnn/model.py:123-230, cnn/blocks.py:10-79, cnn/heads.py:10-38)
// in eval mode, fused with the encoder: trunk of pre-activation residual blocks (optionally with a
// global-pooling branch), features gathered at the two player cells, DeepSet heads.
//
// One block of 256 threads evaluates a tile of CNN_TILE leaves with all activations in LDS:
//   A  [leaf][C][hw]            trunk state x (residual stream)
//   B,C[leaf][C][(h+2)(w+2)]    zero-bordered activations feeding the 3x3 convolutions
// so a 3x3 tap never needs a bounds check. Thread t owns output channel t % C and the rows
// y = t / C (mod 256/C): it keeps <= 2 rows x w columns x CNN_TILE accumulators in registers and, per
// input channel, reads a (rows+2) x (w+2) patch from LDS (broadcast across the C threads that share
// the rows) and one weight per tap from L2 (layout [ci][tap][co], coalesced over co).
// BatchNorm that follows a convolution (stem_bn, bn2) is folded into it at load time; BatchNorm in
// front of a ReLU on the residual stream (bn1, pool_bn) is a per-channel affine applied while the
// padded copy is written. The 3x3 convolutions run on v_mfma_f32_32x32x2_f32 when C is a multiple of 32
// (conv3x3_tile_mfma), otherwise on fp32 FMA (conv3x3_tile); both give the same bits.
#pragma once
// (included from nets.h after the shared helpers)

namespace arnet {

static const int CNN_TILE = 2;
static const int CNN_MAX_BLOCKS = 8;

struct CnnBlockDev {
    const float *bn1_a, *bn1_b;        // [C] affine of bn1
    const float *w1, *b1;              // conv1 with bn2 folded: [C][9][C], [C]
    const float* w2;                   // conv2 [C][9][C]
    int gpool;                         // 0 = res, else gpool channels G
    const float *pbn_a, *pbn_b;        // [C]
    const float* wp;                   // pool_conv [C][G]  (ci-major)
    const float *wl, *bl;              // pool_linear [2G][C], [C]
};

struct CnnDev {
    int width, height, hw, C, PD, HD, n_blocks;
    const float *stem_w, *stem_b;      // [5][9][C] (stem_bn folded), [C]
    CnnBlockDev blk[CNN_MAX_BLOCKS];
    const float *pe_w, *pe_b;          // player_encoder [3][PD], [PD]
    const float *cb_w, *cb_b;          // combiner [(C+PD)][HD], [HD]
    const float *hd_w, *hd_b;          // heads rows: policy 5, value 1 over 2*HD, row-major [6][2HD]
    // PooledValueHead (cnn/heads.py:40-68) instead of the point head's row 5: cat([mean, max of the trunk's output over
    // the board, h_i, agg]) -> Linear(2C + 2HD -> VH) -> ReLU -> Linear(VH -> 1); VH = 0: point head
    int VH;
    const float *pv_w0, *pv_b0;        // [2C + 2HD][VH] (transposed), [VH]
    const float *pv_w2, *pv_b2;        // [VH], [1]
    const uint8_t* maze;               // unused here (maze comes through the boards' maze_off)
    // k_cnn_mfma (trunk state in registers): MT row tiles per wavefront (0: k_cnn), L leaves per workgroup,
    // floats of the image / scratch region P
    int MT, L, p_floats;
};

// zero-bordered patch conv: out[l][co][y][x] = bias[co] + sum_ci sum_tap w[ci][tap][co] * in[l][ci][y+dy][x+dx]
// `in` has stride WP = w+2 per row and (h+2)*WP per channel. Results handed to `emit(l, y, x, value)`.
template <typename Emit>
__device__ inline void conv3x3_tile(const float* __restrict__ wt, const float* __restrict__ bias, int Cin, int C,
                                    const float* in, int in_leaf_stride, int h, int w, int tid, Emit emit) {
    const int co = tid % C, rg = tid / C, RG = NTHREADS / C;
    const int WP = w + 2, chs = (h + 2) * WP;
    if (rg >= h) return;
    const int y0 = rg, y1 = rg + RG;  // RG >= 4 and h <= 8: at most two rows per thread
    const bool two = y1 < h;
    float acc[CNN_TILE][2][8];
    const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
    for (int l = 0; l < CNN_TILE; ++l)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int x = 0; x < 8; ++x) acc[l][r][x] = bv;
    for (int ci = 0; ci < Cin; ++ci) {
        float wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = wt[((size_t)ci * 9 + t) * C + co];
#pragma unroll
        for (int l = 0; l < CNN_TILE; ++l) {
            const float* src = in + (size_t)l * in_leaf_stride + (size_t)ci * chs;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && !two) break;
                const int y = r == 0 ? y0 : y1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float* row = src + (size_t)(y + dy) * WP;
                    float v[10];
#pragma unroll
                    for (int x = 0; x < 10; ++x) v[x] = x < WP ? row[x] : 0.0f;
#pragma unroll
                    for (int x = 0; x < 8; ++x)
                        acc[l][r][x] = fmaf(wv[dy * 3 + 2], v[x + 2], fmaf(wv[dy * 3 + 1], v[x + 1], fmaf(wv[dy * 3], v[x], acc[l][r][x])));
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < CNN_TILE; ++l)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !two) break;
            const int y = r == 0 ? y0 : y1;
#pragma unroll
            for (int x = 0; x < 8; ++x)
                if (x < w) emit(l, co, y, x, acc[l][r][x]);
        }
}

// The same convolution on the matrix cores (C a multiple of 32): an implicit GEMM
//   out[m = (leaf, y, x)][n = co] = bias[co] + sum_{k = ci * 9 + tap} in[leaf][ci][y + dy][x + dx] * w[k][co]
// on v_mfma_f32_32x32x2_f32, whose accumulation is the k-ordered fmaf chain conv3x3_tile runs (ci outer,
// taps inner, bias first), so the two produce the same bits. Wavefront w takes the 32 positions
// 32w .. 32w+31 of the tile's CNN_TILE * hw (<= 128) and all output channels, 64 at a time (two accumulator
// tiles share one read of the activations); the A operand is read straight from the zero-bordered LDS
// image (no im2col), the weights stream from L2 through mfma_pass's register ring.
template <typename Emit>
__device__ inline void conv3x3_tile_mfma(const float* __restrict__ wt, const float* __restrict__ bias, int Cin, int C,
                                         const float* in, int in_leaf_stride, int h, int w, int tid, Emit emit) {
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h2 = lane >> 5;
    const int hw = h * w, M = CNN_TILE * hw, WP = w + 2, chs = (h + 2) * WP, K = Cin * 9;
    if (wave * 32 >= M) return;  // wave-uniform
    const int m = wave * 32 + r;
    const bool row_ok = m < M;
    const int mm = row_ok ? m : 0;
    const float* a_row = in + (mm / hw) * in_leaf_stride + ((mm % hw) / w) * WP + (mm % hw) % w;
    // k runs over (input channel, tap) two channels = 18 values = 9 MFMA steps at a time, so the tap of
    // step p in lane half h2 is a compile-time pattern: k_local = 2p + h2 -> channel k_local / 9, tap k_local % 9
    int off[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        const int kl = 2 * p + h2, cl = kl >= 9 ? 1 : 0, tap = kl - 9 * cl, dy = tap / 3, dx = tap - 3 * dy;
        off[p] = cl * chs + dy * WP + dx;
    }
    const int n_pairs = (Cin + 1) / 2;
    for (int n0 = 0; n0 < C; n0 += 64) {
        const bool two = n0 + 32 < C;
        const float* bp0 = wt + n0 + r;
        const float* bp1 = bp0 + (two ? 32 : 0);
        f32x16 c0, c1;
        const float b0 = bias ? bias[n0 + r] : 0.0f, b1 = bias ? bias[n0 + (two ? 32 : 0) + r] : 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            c0[v] = b0;
            c1[v] = b1;
        }
        // Weights of the next channel pair are fetched while this pair is multiplied. The nine steps of a
        // pair are straight-line code: no guards inside (every branch there starts a new basic block with
        // conservative s_waitcnt's in front of each MFMA). A step past K -- the second channel of an odd
        // last pair -- gets zero weights and reads the next, finite, channel of the LDS image: it adds 0.
        // three register buffers in rotation: while pair q is multiplied the weights of q + 1 are (mostly)
        // there and those of q + 2 are on their way -- two pairs (2 x 9 MFMA steps) of distance to the L2.
        // A fetch past the last pair touches no memory (all its k are >= K) and yields zeros.
        float w0a[9], w0b[9], w1a[9], w1b[9], w2a[9], w2b[9];
        auto fetch = [&](int q, float* A9, float* B9) {
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                const int kk = q * 18 + 2 * p + h2;
                A9[p] = kk < K ? bp0[(size_t)kk * C] : 0.0f;
                B9[p] = kk < K ? bp1[(size_t)kk * C] : 0.0f;
            }
        };
        auto multiply = [&](int q, const float* A9, const float* B9) {
            const float* ab = a_row + (size_t)(2 * q) * chs;
            float av[9];
#pragma unroll
            for (int p = 0; p < 9; ++p) av[p] = row_ok ? ab[off[p]] : 0.0f;
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], A9[p], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], B9[p], c1, 0, 0, 0);
            }
        };
        fetch(0, w0a, w0b);
        fetch(1, w1a, w1b);
        for (int q = 0; q < n_pairs; q += 3) {
            fetch(q + 2, w2a, w2b);
            multiply(q, w0a, w0b);
            if (q + 1 < n_pairs) {
                fetch(q + 3, w0a, w0b);
                multiply(q + 1, w1a, w1b);
            }
            if (q + 2 < n_pairs) {
                fetch(q + 4, w1a, w1b);
                multiply(q + 2, w2a, w2b);
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int mo = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * h2;
            if (mo < M) {
                const int l = mo / hw, cell = mo % hw, y = cell / w, x = cell % w;
                emit(l, n0 + r, y, x, c0[v]);
                if (two) emit(l, n0 + 32 + r, y, x, c1[v]);
            }
        }
    }
}

// conv3x3_tile for any board: thread (co, rg) walks the rows rg, rg + RG, ... eight columns at a time (the same
// chain per output element: bias, then input channels in order, taps row by row)
template <typename Emit>
__device__ inline void conv3x3_tile_big(const float* __restrict__ wt, const float* __restrict__ bias, int Cin, int C,
                                        const float* in, int in_leaf_stride, int h, int w, int tid, Emit emit) {
    const int co = tid % C, rg = tid / C, RG = NTHREADS / C;
    const int WP = w + 2, chs = (h + 2) * WP;
    const float bv = bias ? bias[co] : 0.0f;
    for (int y = rg; y < h; y += RG)
        for (int x0 = 0; x0 < w; x0 += 8) {
            float acc[CNN_TILE][8];
#pragma unroll
            for (int l = 0; l < CNN_TILE; ++l)
#pragma unroll
                for (int x = 0; x < 8; ++x) acc[l][x] = bv;
            for (int ci = 0; ci < Cin; ++ci) {
                float wv[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wv[t] = wt[((size_t)ci * 9 + t) * C + co];
#pragma unroll
                for (int l = 0; l < CNN_TILE; ++l) {
                    const float* src = in + (size_t)l * in_leaf_stride + (size_t)ci * chs;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const float* row = src + (size_t)(y + dy) * WP + x0;
                        float v[10];
#pragma unroll
                        for (int x = 0; x < 10; ++x) v[x] = x0 + x < WP ? row[x] : 0.0f;
#pragma unroll
                        for (int x = 0; x < 8; ++x)
                            acc[l][x] = fmaf(wv[dy * 3 + 2], v[x + 2], fmaf(wv[dy * 3 + 1], v[x + 1], fmaf(wv[dy * 3], v[x], acc[l][x])));
                    }
                }
            }
#pragma unroll
            for (int l = 0; l < CNN_TILE; ++l)
#pragma unroll
                for (int x = 0; x < 8; ++x)
                    if (x0 + x < w) emit(l, co, y, x0 + x, acc[l][x]);
        }
}

// scalar or matrix-core convolution by channel count (block-uniform)
template <typename Emit>
__device__ inline void conv3x3_any(const float* __restrict__ wt, const float* __restrict__ bias, int Cin, int C,
                                   const float* in, int in_leaf_stride, int h, int w, int tid, Emit emit) {
    if ((C & 31) == 0 && CNN_TILE * h * w <= 32 * (NTHREADS / 64)) conv3x3_tile_mfma(wt, bias, Cin, C, in, in_leaf_stride, h, w, tid, emit);
    else if (h <= 8 && w <= 8) conv3x3_tile(wt, bias, Cin, C, in, in_leaf_stride, h, w, tid, emit);
    else conv3x3_tile_big(wt, bias, Cin, C, in, in_leaf_stride, h, w, tid, emit);
}

// heads: features at the player cells, player encoder, combiner, DeepSet heads (policy, point or pooled value);
// `A` = the trunk's output [L][C][hw] in LDS. All threads of the block call this.
template <int NW>
__device__ inline void cnn_heads(const CnnDev& net, int L, int cnt, const float* A, int a_leaf, float* small,
                                 const LeafFeat* feat, int tid, uint32_t base, ar::EvalOut* out, float* logits) {
    const int C = net.C, hw = net.hw;
    const int PD = net.PD, HD = net.HD;
    float* cat = small;                    // [L][2][C+PD]
    float* hid = cat + L * 2 * (C + PD);   // [L][2][HD]
    float* hl = hid + L * 2 * HD;          // [L][12]
    for (int i = tid; i < L * 2 * (C + PD); i += NTHREADS) {
        const int l = i / (2 * (C + PD)), rem = i % (2 * (C + PD)), p = rem / (C + PD), k = rem % (C + PD);
        const LeafFeat& f = feat[l];
        float v;
        if (k < C) {
            v = A[(size_t)l * a_leaf + (size_t)k * hw + (p == 0 ? f.p1 : f.p2)];
        } else {
            const int o = k - C;
            // side = [score, mud, progress]  (model.py:153-168)
            const float s0 = f.sc[4 + p], s1 = f.sc[2 + p], s2 = f.sc[1];
            float acc = net.pe_b[o];
            acc = fmaf(net.pe_w[0 * PD + o], s0, acc);
            acc = fmaf(net.pe_w[1 * PD + o], s1, acc);
            acc = fmaf(net.pe_w[2 * PD + o], s2, acc);
            v = fmaxf(acc, 0.0f);
        }
        cat[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < L * 2 * HD; i += NTHREADS) {
        const int l = i / (2 * HD), rem = i % (2 * HD), p = rem / HD, o = rem % HD;
        const float* cv = cat + (size_t)(l * 2 + p) * (C + PD);
        float acc = net.cb_b[o];
#pragma unroll 16
        for (int k = 0; k < C + PD; ++k) acc = fmaf(net.cb_w[(size_t)k * HD + o], cv[k], acc);
        hid[i] = fmaxf(acc, 0.0f);
    }
    __syncthreads();
    for (int i = tid; i < L * 12; i += NTHREADS) {
        const int l = i / 12, r = i % 12, p = r / 6, o = r % 6;
        const float* hi = hid + (size_t)(l * 2 + p) * HD;
        const float* h0 = hid + (size_t)(l * 2) * HD;
        const float* h1 = h0 + HD;
        const float* wr = net.hd_w + (size_t)o * 2 * HD;
        float acc = net.hd_b[o];
#pragma unroll 16
        for (int k = 0; k < HD; ++k) acc = fmaf(wr[k], hi[k], acc);
#pragma unroll 16
        for (int k = 0; k < HD; ++k) acc = fmaf(wr[HD + k], h0[k] + h1[k], acc);
        hl[i] = acc;
    }
    __syncthreads();
    if (net.VH > 0) {
        const int VH = net.VH;
        float* pool = hl + L * 12;        // [L][2C]: mean, max of the trunk's output over the board
        float* vh = pool + L * 2 * C;     // [L][2][VH]
        for (int i = tid; i < L * C; i += NTHREADS) {
            const int l = i / C, c = i % C;
            const float* a = &A[(size_t)l * a_leaf + (size_t)c * hw];
            float s = 0.0f, mx = a[0];
            for (int k = 0; k < hw; ++k) {
                s += a[k];
                mx = fmaxf(mx, a[k]);
            }
            pool[l * 2 * C + c] = s / (float)hw;
            pool[l * 2 * C + C + c] = mx;
        }
        __syncthreads();
        for (int i = tid; i < L * 2 * VH; i += NTHREADS) {
            const int l = i / (2 * VH), rem = i % (2 * VH), p = rem / VH, o = rem % VH;
            const float* hi = hid + (size_t)(l * 2 + p) * HD;
            const float* h0 = hid + (size_t)(l * 2) * HD;
            const float* h1 = h0 + HD;
            const float* pl = pool + (size_t)l * 2 * C;
            float acc = net.pv_b0[o];
            for (int k = 0; k < 2 * C; ++k) acc = fmaf(net.pv_w0[(size_t)k * VH + o], pl[k], acc);
            for (int k = 0; k < HD; ++k) acc = fmaf(net.pv_w0[(size_t)(2 * C + k) * VH + o], hi[k], acc);
            for (int k = 0; k < HD; ++k) acc = fmaf(net.pv_w0[(size_t)(2 * C + HD + k) * VH + o], h0[k] + h1[k], acc);
            vh[i] = fmaxf(acc, 0.0f);
        }
        __syncthreads();
        for (int i = tid; i < L * 2; i += NTHREADS) {
            const float* v = vh + (size_t)i * VH;
            float acc = net.pv_b2[0];
            for (int k = 0; k < VH; ++k) acc = fmaf(net.pv_w2[k], v[k], acc);
            hl[(i / 2) * 12 + (i % 2) * 6 + 5] = acc;  // the value logit of leaf i / 2, player i % 2
        }
        __syncthreads();
    }
    if (tid < cnt) {
        const float* hh = hl + tid * 12;
        ar::EvalOut o;
        softmax5(hh, o.p1);
        softmax5(hh + 6, o.p2);
        o.v1 = softplusf(hh[5]);
        o.v2 = softplusf(hh[11]);
        out[base + tid] = o;
        if (logits)
            for (int k = 0; k < 5; ++k) {
                logits[(size_t)(base + tid) * 10 + k] = hh[k];
                logits[(size_t)(base + tid) * 10 + 5 + k] = hh[6 + k];
            }
    }
}

#if defined(AR_CNN_PROF)
__device__ int g_cnn_prof_done = 0;
#define CNN_T(i) do { __syncthreads(); if (prof) tp[i] = wall_clock64(); } while (0)
#else
#define CNN_T(i) ((void)0)
#endif
template <int NW>
__global__ void __launch_bounds__(NTHREADS) k_cnn(CnnDev net, const ar::LeafReq<NW>* q, const uint32_t* qcount,
                                                  uint32_t n_fixed, const char* boards, size_t board_stride,
                                                  const uint8_t* maze_pool, ar::EvalOut* out, float* logits) {
    constexpr int L = CNN_TILE;
    extern __shared__ float smem[];
    const int C = net.C, h = net.height, w = net.width, hw = net.hw;
    const int WP = w + 2, chs = (h + 2) * WP;
    const int a_leaf = C * hw, p_leaf = C * chs;
    float* A = smem;                          // [L][C][hw]
    float* Bp = A + (size_t)L * a_leaf;       // [L][C][chs] padded
    float* Cp = Bp + (size_t)L * p_leaf;      // [L][C][chs] padded
    float* small = Cp + (size_t)L * p_leaf;   // [L][4*64] scratch for pooled / head vectors
    __shared__ LeafFeat feat[L];
    const uint32_t n = qcount ? *qcount : n_fixed;
    const uint32_t base = blockIdx.x * L;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (uint32_t)L ? (n - base) : (uint32_t)L);
    const int tid = threadIdx.x;
#if defined(AR_CNN_PROF)
    const bool prof = blockIdx.x == 0 && tid == 0;
    unsigned long long tp[24];
    for (int i = 0; i < 24; ++i) tp[i] = 0;
    int tpi = 0;
#endif
    CNN_T(0);
    // zero both padded buffers once: borders stay zero, interiors are always overwritten
    for (int i = tid; i < 2 * L * p_leaf; i += NTHREADS) Bp[i] = 0.0f;
    if (tid < L) {
        const int l = tid < cnt ? tid : 0;
        const ar::LeafReq<NW>& r = q[base + l];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)r.slot * board_stride);
        leaf_features<NW>(r.st, b, hw, feat[tid]);
    }
    __syncthreads();
    // spatial input (5 channels: maze up/right/down/left, cheese) into the padded buffer Cp[l][0..4]
    for (int i = tid; i < L * 5 * hw; i += NTHREADS) {
        const int l = i / (5 * hw), rem = i % (5 * hw), c = rem / hw, cell = rem % hw;
        const int ll = l < cnt ? l : 0;
        const ar::LeafReq<NW>& r = q[base + ll];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)r.slot * board_stride);
        float v;
        if (c < 4) {
            const uint8_t cst = maze_pool[b.maze_off + (uint32_t)cell * 4u + (uint32_t)c];
            v = cst ? (float)cst / 10.0f : -1.0f;
        } else {
            v = ar::st_has_cheese(r.st, cell) ? 1.0f : 0.0f;
        }
        Cp[(size_t)l * p_leaf + (size_t)c * chs + (size_t)(cell / w + 1) * WP + (cell % w + 1)] = v;
    }
    __syncthreads();
    CNN_T(1);
    // stem: conv(5 -> C) + folded stem_bn + ReLU -> A
    conv3x3_any(net.stem_w, net.stem_b, 5, C, Cp, p_leaf, h, w, tid, [&](int l, int co, int y, int x, float v) {
        A[(size_t)l * a_leaf + (size_t)co * hw + y * w + x] = fmaxf(v, 0.0f);
    });
    __syncthreads();
    CNN_T(2);
    for (int bi = 0; bi < net.n_blocks; ++bi) {
        const CnnBlockDev& blk = net.blk[bi];
        float* pout = small;  // [L][C] pooled-branch output (gpool blocks)
        if (blk.gpool) {
            const int G = blk.gpool;
            float* pc = Cp;  // reuse as [L][G][hw] (unpadded scratch); re-zeroed below
            // pool path: relu(pool_bn(x)) -> 1x1 conv -> mean/max over cells
            for (int i = tid; i < L * G * hw; i += NTHREADS) {
                const int l = i / (G * hw), rem = i % (G * hw), g = rem / hw, cell = rem % hw;
                float acc = 0.0f;
#pragma unroll 16  // the loads of 16 channels in flight together: the chain itself is only 16 fmas
                for (int c = 0; c < C; ++c) {
                    const float xv = fmaxf(fmaf(blk.pbn_a[c], A[(size_t)l * a_leaf + (size_t)c * hw + cell], blk.pbn_b[c]), 0.0f);
                    acc = fmaf(blk.wp[(size_t)c * G + g], xv, acc);
                }
                pc[(size_t)l * G * hw + (size_t)g * hw + cell] = acc;
            }
            __syncthreads();
            float* pcat = small + L * C;  // [L][2G]
            for (int i = tid; i < L * G; i += NTHREADS) {
                const int l = i / G, g = i % G;
                const float* p = pc + (size_t)l * G * hw + (size_t)g * hw;
                float s = 0.0f, mx = p[0];
                for (int c2 = 0; c2 < hw; ++c2) {
                    s += p[c2];
                    mx = fmaxf(mx, p[c2]);
                }
                pcat[l * 2 * G + g] = s / (float)hw;
                pcat[l * 2 * G + G + g] = mx;
            }
            __syncthreads();
            for (int i = tid; i < L * C; i += NTHREADS) {
                const int l = i / C, c = i % C;
                float acc = blk.bl[c];
#pragma unroll 16
                for (int k = 0; k < 2 * G; ++k) acc = fmaf(blk.wl[(size_t)k * C + c], pcat[l * 2 * G + k], acc);
                pout[l * C + c] = acc;
            }
            __syncthreads();
            for (int i = tid; i < L * p_leaf; i += NTHREADS) Cp[i] = 0.0f;  // restore the zero borders
            __syncthreads();
        }
        CNN_T(3 + bi * 4);
        // Bp = pad(relu(bn1(A)))
#pragma unroll 4
        for (int i = tid; i < L * a_leaf; i += NTHREADS) {
            const int l = i / a_leaf, rem = i % a_leaf, c = rem / hw, cell = rem % hw;
            Bp[(size_t)l * p_leaf + (size_t)c * chs + (size_t)(cell / w + 1) * WP + (cell % w + 1)] =
                fmaxf(fmaf(blk.bn1_a[c], A[i], blk.bn1_b[c]), 0.0f);
        }
        __syncthreads();
        CNN_T(4 + bi * 4);
        // Cp = pad(relu(conv1'(Bp)))   (bn2 folded)
        conv3x3_any(blk.w1, blk.b1, C, C, Bp, p_leaf, h, w, tid, [&](int l, int co, int y, int x, float v) {
            Cp[(size_t)l * p_leaf + (size_t)co * chs + (size_t)(y + 1) * WP + (x + 1)] = fmaxf(v, 0.0f);
        });
        __syncthreads();
        CNN_T(5 + bi * 4);
        // A = conv2(Cp) [+ pooled] + A
        conv3x3_any(blk.w2, nullptr, C, C, Cp, p_leaf, h, w, tid, [&](int l, int co, int y, int x, float v) {
            float* a = &A[(size_t)l * a_leaf + (size_t)co * hw + y * w + x];
            *a = blk.gpool ? v + pout[l * C + co] + *a : v + *a;
        });
        __syncthreads();
        CNN_T(6 + bi * 4);
    }
    cnn_heads<NW>(net, L, cnt, A, a_leaf, small, feat, tid, base, out, logits);
    CNN_T(20);
#if defined(AR_CNN_PROF)
    if (prof && atomicCAS(&g_cnn_prof_done, 0, 1) == 0) {
        printf("[cnn prof, 10 ns ticks] zero+input %llu stem %llu", tp[1] - tp[0], tp[2] - tp[1]);
        for (int bi = 0; bi < net.n_blocks; ++bi)
            printf(" | blk%d pool %llu bn %llu conv1 %llu conv2 %llu", bi, tp[3 + bi * 4] - (bi ? tp[2 + bi * 4] : tp[2]),
                   tp[4 + bi * 4] - tp[3 + bi * 4], tp[5 + bi * 4] - tp[4 + bi * 4], tp[6 + bi * 4] - tp[5 + bi * 4]);
        printf(" | heads %llu\n", tp[20] - tp[2 + net.n_blocks * 4]);
    }
    (void)tpi;
#endif
}

// ---- PyRatCNN with the trunk state in registers (C = 32 or 64, boards up to 256 cells) ---------------------
// The kernel above keeps three activation images per tile in LDS (trunk state A and two zero-bordered copies:
// 108 KB at 7x7 / C = 64, one workgroup per CU, one wavefront per SIMD -- the matrix pipe idles through every
// phase that is not a convolution: 31 % busy, profiles/r03_pmc_sq_cnn.txt). Here the trunk state x lives in the
// registers of the wavefront that computes it, in the accumulator layout of the convolutions (row = board
// position, column = channel), and there is ONE zero-bordered image P: a convolution's results stay in the
// accumulators until every wavefront has finished reading P, then go over it (42 KB at 7x7 / C = 64: three
// workgroups per CU, whose phases interleave). Rows are positions m = leaf * hw + cell of the tile's L leaves;
// wavefront wv holds the 32-row tiles wv, wv + 4, ... (MT of them: 128 MT rows per workgroup), so a board above
// 128 cells is one leaf over two tiles per wavefront. The pooling branch's 1x1 convolution runs on the matrix
// cores too (its operand relu(pool_bn(x)) is laid out in P, which is free between blocks). Arithmetic, operation
// by operation, is k_cnn's: the same k-ordered chains, the same affine / ReLU / residual order.
static const int CNN_TILE_MAX = 8;

// one 3x3 convolution for T row tiles of this wavefront: c[t][j] = bias + sum_k P-window(row, k) * w[k][co]
template <int MT, int T>
__device__ inline void conv3x3_rows_mfma(const float* __restrict__ wt, const float* __restrict__ bias, int Cin, int C,
                                         const float* P, const int (&arow)[MT], const bool (&rok)[MT], int chs, int WP,
                                         int r, int h2_, bool two, f32x16 (&c)[MT][2]) {
    const int K = Cin * 9;
    // (opaque copy: what follows is the same for every convolution of the trunk, and hoisted out of the loop over the
    // blocks the per-tap offsets would sit in registers -- spilled ones -- through the whole kernel)
    int h2 = h2_;
    asm volatile("" : "+v"(h2));
    int off[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        const int kl = 2 * p + h2, cl = kl >= 9 ? 1 : 0, tap = kl - 9 * cl, dy = tap / 3, dx = tap - 3 * dy;
        off[p] = cl * chs + dy * WP + dx;
    }
    const int n_pairs = (Cin + 1) / 2;
    const float b0 = bias ? bias[r] : 0.0f, b1 = bias ? bias[(two ? 32 : 0) + r] : 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            c[t][0][v] = b0;
            c[t][1][v] = b1;
        }
    // weights of two channels (18 k values, nine MFMA steps) per fetch, two register buffers in turn (the other
    // wavefronts of the SIMD cover what one pair of distance to the L2 does not); a fetch feeds the T tiles
    float w0a[9], w0b[9], w1a[9], w1b[9];
    // (32-bit offsets from the uniform base, row clamped instead of a guarded load: no branch, no 64-bit address per tap)
    const unsigned col0 = (unsigned)r, col1 = (unsigned)(r + (two ? 32 : 0));
    auto fetch = [&](int q, float* A9, float* B9) {
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int kk = q * 18 + 2 * p + h2;
            const unsigned row = (unsigned)(kk < K ? kk : K - 1) * (unsigned)C;
            const float a = wt[row + col0], b = wt[row + col1];
            A9[p] = kk < K ? a : 0.0f;
            B9[p] = kk < K ? b : 0.0f;
        }
    };
    auto multiply = [&](int q, const float* A9, const float* B9) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float* ab = P + (arow[t] + 2 * q * chs);
            float av[9];
#pragma unroll
            for (int p = 0; p < 9; ++p) av[p] = rok[t] ? ab[off[p]] : 0.0f;
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                c[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], A9[p], c[t][0], 0, 0, 0);
                c[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], B9[p], c[t][1], 0, 0, 0);
            }
        }
    };
    fetch(0, w0a, w0b);
    for (int q = 0; q < n_pairs; q += 2) {
        fetch(q + 1, w1a, w1b);
        multiply(q, w0a, w0b);
        if (q + 1 < n_pairs) {
            fetch(q + 2, w0a, w0b);
            multiply(q + 1, w1a, w1b);
        }
    }
}

template <int NW, int MT>
__global__ void __launch_bounds__(NTHREADS, MT == 1 ? 3 : 2) k_cnn_mfma(CnnDev net, const ar::LeafReq<NW>* q, const uint32_t* qcount,
                                                       uint32_t n_fixed, const char* boards, size_t board_stride,
                                                       const uint8_t* maze_pool, ar::EvalOut* out, float* logits) {
    constexpr int ROWS = 128 * MT;
    extern __shared__ float smem[];
    const int L = net.L, C = net.C, h = net.height, w = net.width, hw = net.hw;
    const int WP = w + 2, chs = (h + 2) * WP;
    const int a_leaf = C * hw, p_leaf = C * chs, M = L * hw;
    float* P = smem;                  // [L][C][chs] zero-bordered; between convolutions: [L][C][hw] (+ [L][G][hw])
    float* small = P + net.p_floats;  // pooled / head vectors
    __shared__ LeafFeat feat[CNN_TILE_MAX];
    __shared__ int row_p[ROWS], row_u[ROWS], row_l[ROWS];  // per row: offset of its cell in P (channel 0), in [L][C][hw], leaf
    const uint32_t n = qcount ? *qcount : n_fixed;
    const uint32_t base = blockIdx.x * (uint32_t)L;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (uint32_t)L ? (n - base) : (uint32_t)L);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h2 = lane >> 5;
    const bool two = C > 32;
    for (int m = tid; m < ROWS; m += NTHREADS) {
        const int mm = m < M ? m : 0;
        const int l = mm / hw, cell = mm - l * hw, y = cell / w, x = cell - y * w;
        row_p[m] = l * p_leaf + (y + 1) * WP + (x + 1);
        row_u[m] = l * a_leaf + cell;
        row_l[m] = l;
    }
    for (int i = tid; i < L * p_leaf; i += NTHREADS) P[i] = 0.0f;
    if (tid < L) {
        const int l = tid < cnt ? tid : 0;
        const ar::LeafReq<NW>& rq = q[base + l];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)rq.slot * board_stride);
        leaf_features<NW>(rq.st, b, hw, feat[tid]);
    }
    __syncthreads();
    for (int i = tid; i < L * 5 * hw; i += NTHREADS) {
        const int l = i / (5 * hw), rem = i % (5 * hw), c = rem / hw, cell = rem % hw;
        const int ll = l < cnt ? l : 0;
        const ar::LeafReq<NW>& rq = q[base + ll];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)rq.slot * board_stride);
        float v;
        if (c < 4) {
            const uint8_t cst = maze_pool[b.maze_off + (uint32_t)cell * 4u + (uint32_t)c];
            v = cst ? (float)cst / 10.0f : -1.0f;
        } else {
            v = ar::st_has_cheese(rq.st, cell) ? 1.0f : 0.0f;
        }
        P[(size_t)l * p_leaf + (size_t)c * chs + (size_t)(cell / w + 1) * WP + (cell % w + 1)] = v;
    }
    // this lane's operand rows (window origin in P) and how many of the wavefront's tiles hold rows at all
    int arow[MT], urow[MT];
    bool rok[MT];
    int n_on = 0;  // wave-uniform
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        rok[t] = (wave + 4 * t) * 32 + r < M;
        if ((wave + 4 * t) * 32 < M) n_on = t + 1;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = (wave + 4 * t) * 32 + r;
        arow[t] = row_p[rok[t] ? m : 0] - WP - 1;
        urow[t] = row_u[rok[t] ? m : 0];
    }
    auto conv = [&](const float* wt, const float* bias, int Cin, f32x16(&c)[MT][2]) {
        if (MT == 2 && n_on == 2) conv3x3_rows_mfma<MT, MT>(wt, bias, Cin, C, P, arow, rok, chs, WP, r, h2, two, c);
        else if (n_on >= 1) conv3x3_rows_mfma<MT, 1>(wt, bias, Cin, C, P, arow, rok, chs, WP, r, h2, two, c);
    };
    // result rows of this lane: tile t, register v -> row mo
    auto row_of = [&](int t, int v) -> int { return (wave + 4 * t) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h2; };
    f32x16 xs[MT][2], acc[MT][2];
    // stem: conv(5 -> C) + folded stem_bn + ReLU
    conv(net.stem_w, net.stem_b, 5, acc);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            xs[t][0][v] = fmaxf(acc[t][0][v], 0.0f);
            xs[t][1][v] = fmaxf(acc[t][1][v], 0.0f);
        }
    __syncthreads();  // every wavefront is done reading P
    for (int bi = 0; bi < net.n_blocks; ++bi) {
        const CnnBlockDev& blk = net.blk[bi];
        float* pout = small;  // [L][C] pooled-branch output (gpool blocks)
        if (blk.gpool) {
            const int G = blk.gpool;
            float* U = P;                         // [L][C][hw] relu(pool_bn(x))
            float* pc = P + (size_t)L * a_leaf;   // [L][G][hw]
            {
                const float a0 = blk.pbn_a[r], c0 = blk.pbn_b[r], a1 = blk.pbn_a[two ? r + 32 : r], c1 = blk.pbn_b[two ? r + 32 : r];
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int mo = row_of(t, v);
                        if (t < n_on && mo < M) {
                            const int u = row_u[mo];
                            U[u + r * hw] = fmaxf(fmaf(a0, xs[t][0][v], c0), 0.0f);
                            if (two) U[u + (r + 32) * hw] = fmaxf(fmaf(a1, xs[t][1][v], c1), 0.0f);
                        }
                    }
            }
            __syncthreads();
            // 1x1 convolution C -> G on the matrix cores: pc[m][g] = sum_c U[m][c] * wp[c][g] (k = c, from 0)
            {
                const bool g0 = r < G, g1 = r + 32 < G, wide = G > 32;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][0][v] = acc[t][1][v] = 0.0f;
                if (n_on >= 1) {
#pragma unroll 4
                    for (int k = 0; k < C; k += 2) {
                        const float b0 = g0 ? blk.wp[(size_t)(k + h2) * G + r] : 0.0f;
                        const float b1 = g1 ? blk.wp[(size_t)(k + h2) * G + r + 32] : 0.0f;
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            if (t < n_on) {
                                const float a = rok[t] ? U[urow[t] + (k + h2) * hw] : 0.0f;
                                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[t][0], 0, 0, 0);
                                if (wide) acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[t][1], 0, 0, 0);
                            }
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int mo = row_of(t, v);
                        if (t < n_on && mo < M) {
                            const int l = row_l[mo], cell = row_u[mo] - l * a_leaf;
                            if (g0) pc[(size_t)l * G * hw + (size_t)r * hw + cell] = acc[t][0][v];
                            if (g1) pc[(size_t)l * G * hw + (size_t)(r + 32) * hw + cell] = acc[t][1][v];
                        }
                    }
            }
            __syncthreads();
            float* pcat = small + L * C;  // [L][2G]
            for (int i = tid; i < L * G; i += NTHREADS) {
                const int l = i / G, g = i % G;
                const float* p = pc + (size_t)l * G * hw + (size_t)g * hw;
                float s = 0.0f, mx = p[0];
                for (int c2 = 0; c2 < hw; ++c2) {
                    s += p[c2];
                    mx = fmaxf(mx, p[c2]);
                }
                pcat[l * 2 * G + g] = s / (float)hw;
                pcat[l * 2 * G + G + g] = mx;
            }
            __syncthreads();
            for (int i = tid; i < L * C; i += NTHREADS) {
                const int l = i / C, c = i % C;
                float a = blk.bl[c];
#pragma unroll 16
                for (int k = 0; k < 2 * G; ++k) a = fmaf(blk.wl[(size_t)k * C + c], pcat[l * 2 * G + k], a);
                pout[l * C + c] = a;
            }
            for (int i = tid; i < L * p_leaf; i += NTHREADS) P[i] = 0.0f;  // the zero borders again
            __syncthreads();
        }
        // P = pad(relu(bn1(x)))
        {
            const float a0 = blk.bn1_a[r], c0 = blk.bn1_b[r], a1 = blk.bn1_a[two ? r + 32 : r], c1 = blk.bn1_b[two ? r + 32 : r];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int mo = row_of(t, v);
                    if (t < n_on && mo < M) {
                        const int p = row_p[mo];
                        P[p + r * chs] = fmaxf(fmaf(a0, xs[t][0][v], c0), 0.0f);
                        if (two) P[p + (r + 32) * chs] = fmaxf(fmaf(a1, xs[t][1][v], c1), 0.0f);
                    }
                }
        }
        __syncthreads();
        conv(blk.w1, blk.b1, C, acc);  // bn2 folded
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int mo = row_of(t, v);
                if (t < n_on && mo < M) {
                    const int p = row_p[mo];
                    P[p + r * chs] = fmaxf(acc[t][0][v], 0.0f);
                    if (two) P[p + (r + 32) * chs] = fmaxf(acc[t][1][v], 0.0f);
                }
            }
        __syncthreads();
        conv(blk.w2, nullptr, C, acc);
        // x = conv2 [+ pooled] + x
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (blk.gpool) {
                    const int mo = row_of(t, v), l = row_l[mo < M ? mo : 0];
                    xs[t][0][v] = acc[t][0][v] + pout[l * C + r] + xs[t][0][v];
                    xs[t][1][v] = acc[t][1][v] + pout[l * C + (two ? r + 32 : r)] + xs[t][1][v];
                } else {
                    xs[t][0][v] = acc[t][0][v] + xs[t][0][v];
                    xs[t][1][v] = acc[t][1][v] + xs[t][1][v];
                }
            }
        __syncthreads();
    }
    // the trunk's output as [L][C][hw] for the heads
    float* A = P;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int mo = row_of(t, v);
            if (t < n_on && mo < M) {
                const int u = row_u[mo];
                A[u + r * hw] = xs[t][0][v];
                if (two) A[u + (r + 32) * hw] = xs[t][1][v];
            }
        }
    __syncthreads();
    cnn_heads<NW>(net, L, cnt, A, a_leaf, small, feat, tid, base, out, logits);
}

// ---- host: blob -> device weights ---------------------------------------------------------------
struct CnnHost {
    CnnDev dev;
    size_t smem = 0;
};

// conv weight [co][ci][3][3] (optionally scaled per co) -> [ci][tap][co]
inline std::vector<float> conv_t(const std::vector<float>& w, int co_n, int ci_n, const std::vector<double>* scale) {
    std::vector<float> t((size_t)ci_n * 9 * co_n);
    for (int co = 0; co < co_n; ++co)
        for (int ci = 0; ci < ci_n; ++ci)
            for (int k = 0; k < 9; ++k)
                t[((size_t)ci * 9 + k) * co_n + co] =
                    (float)((double)w[((size_t)co * ci_n + ci) * 9 + k] * (scale ? (*scale)[co] : 1.0));
    return t;
}
inline bool bn_affine(const Blob& b, const std::string& p, int n, std::vector<double>& a, std::vector<double>& c,
                      std::string& err) {
    const std::vector<float>*g = b.get(p + ".weight"), *be = b.get(p + ".bias"), *m = b.get(p + ".running_mean"),
                      *v = b.get(p + ".running_var");
    if (!g || !be || !m || !v || (int)g->size() != n) {
        err = "weight blob lacks " + p;
        return false;
    }
    a.resize(n);
    c.resize(n);
    for (int i = 0; i < n; ++i) {
        a[i] = (double)(*g)[i] / sqrt((double)(*v)[i] + 1e-5);
        c[i] = (double)(*be)[i] - (double)(*m)[i] * a[i];
    }
    return true;
}

}  // namespace arnet
