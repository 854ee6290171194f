// Policy/value heads as HIP kernels: PyRatMLP and SymmetricMLP (fp32), fused with the flat
// observation encoder. Replaces FlatEncoder + OnnxBackend / TensorrtBackend
// (crates/alpharat-sampling/src/flat_encoder.rs:52-125, backends/onnx.rs:176-246,
// backends/tensorrt.rs:423 ff.) and restates `model.predict()` of
// alpharat/nn/models/mlp.py:120-153 and alpharat/nn/models/symmetric.py:124-229 in eval mode.
//
// What the kernels exploit (none of it changes results beyond fp32 summation order):
//   * BatchNorm (eval) is folded into the preceding Linear at load time.
//   * The observation is never materialised for the network: its maze block is the same for every
//     leaf of a maze, so its product with the first layer is a per-maze constant vector; the
//     player one-hots select one weight column each; the cheese mask selects <= n_cheese columns;
//     six scalars scale six columns. First layer cost: ~(8 + n_cheese) x H instead of D x H MACs.
//   * Weights are stored transposed ([in][out]) so the 256 threads of a block (one per output
//     neuron) read them coalesced; a block evaluates a tile of leaves so each weight fetched from
//     L2 is reused across the tile; activations live in LDS and are read as broadcasts.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/alpharat_hip.h"
#include "dev_search.h"

static int nets_fail(int code, const std::string& msg);

namespace arnet {

enum { ARCH_MLP = 0, ARCH_SYMMETRIC = 1, ARCH_CNN = 2 };
static const int TILE_MLP = 32;  // leaves per block
static const int TILE_SYM = 16;
static const int NTHREADS = 256;

// Device-side weight views (all fp32, BN folded, transposed to [in][out])
struct NetDev {
    int arch, width, height, hw, H;
    // MLP: l1 = trunk.0 (+trunk.1), l2 = trunk.4 (+trunk.5), head rows: p1[5] p2[5] v[2]
    // SYM: l1 = shared_encoder, lp = player_encoder, l2 = trunk.0 (K = 2H), l3 = trunk.4,
    //      head rows: policy[5] value[1] over K = 2H
    const float *w1t, *b1;   // [D1][H]
    const float *wpt, *bp;   // [hw+2][H]        (symmetric only)
    const float *w2t, *b2;   // [K2][H]
    const float *w3t, *b3;   // [H][H]           (symmetric only)
    const float *wh, *bh;    // [n_head][Kh] row-major
    int n_head, Kh;
    const float* cmaze;      // [n_mazes][H] first-layer maze contribution incl. bias
    int n_mazes;
};

struct Blob {
    uint32_t arch = 0, width = 0, height = 0;
    std::map<std::string, std::vector<float>> t;
    std::map<std::string, std::vector<uint32_t>> dims;
    bool load(const char* path, std::string& err) {
        FILE* f = fopen(path, "rb");
        if (!f) {
            err = std::string("cannot open weight blob ") + path;
            return false;
        }
        char magic[8];
        uint32_t n = 0;
        bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "ARNET001", 8) == 0 && fread(&arch, 4, 1, f) == 1 &&
                  fread(&width, 4, 1, f) == 1 && fread(&height, 4, 1, f) == 1 && fread(&n, 4, 1, f) == 1;
        for (uint32_t i = 0; ok && i < n; ++i) {
            uint32_t nl = 0, nd = 0;
            ok = fread(&nl, 4, 1, f) == 1 && nl < 4096;
            std::string name(ok ? nl : 0, '\0');
            ok = ok && fread(&name[0], 1, nl, f) == nl && fread(&nd, 4, 1, f) == 1 && nd <= 8;
            std::vector<uint32_t> d(ok ? nd : 0);
            size_t cnt = 1;
            for (uint32_t k = 0; ok && k < nd; ++k) {
                ok = fread(&d[k], 4, 1, f) == 1;
                cnt *= d[k];
            }
            if (ok) {
                std::vector<float> v(cnt);
                ok = fread(v.data(), 4, cnt, f) == cnt;
                t[name] = std::move(v);
                dims[name] = d;
            }
        }
        fclose(f);
        if (!ok) err = std::string("malformed weight blob ") + path;
        return ok;
    }
    const std::vector<float>* get(const std::string& k) const {
        auto it = t.find(k);
        return it == t.end() ? nullptr : &it->second;
    }
};

// Linear (+ eval BatchNorm) -> transposed weights [in][out] and bias [out]
inline bool fold_linear(const Blob& b, const std::string& lin, const std::string& bn, std::vector<float>& wt,
                        std::vector<float>& bias, uint32_t& in, uint32_t& out, std::string& err) {
    const std::vector<float>* w = b.get(lin + ".weight");
    const std::vector<float>* bi = b.get(lin + ".bias");
    if (!w || b.dims.at(lin + ".weight").size() != 2) {
        err = "weight blob lacks " + lin + ".weight";
        return false;
    }
    out = b.dims.at(lin + ".weight")[0];
    in = b.dims.at(lin + ".weight")[1];
    std::vector<double> scale(out, 1.0), shift(out, 0.0);
    if (!bn.empty()) {
        const std::vector<float>*g = b.get(bn + ".weight"), *be = b.get(bn + ".bias"), *m = b.get(bn + ".running_mean"),
                          *v = b.get(bn + ".running_var");
        if (!g || !be || !m || !v) {
            err = "weight blob lacks " + bn + " statistics";
            return false;
        }
        for (uint32_t o = 0; o < out; ++o) {
            scale[o] = (double)(*g)[o] / sqrt((double)(*v)[o] + 1e-5);
            shift[o] = (double)(*be)[o] - (double)(*m)[o] * scale[o];
        }
    }
    wt.assign((size_t)in * out, 0.0f);
    bias.assign(out, 0.0f);
    for (uint32_t o = 0; o < out; ++o) {
        for (uint32_t i = 0; i < in; ++i) wt[(size_t)i * out + o] = (float)((double)(*w)[(size_t)o * in + i] * scale[o]);
        bias[o] = (float)((bi ? (double)(*bi)[o] : 0.0) * scale[o] + shift[o]);
    }
    return true;
}

// ---- device helpers ---------------------------------------------------------------------------
// one leaf's sparse first-layer inputs
struct LeafFeat {
    int p1, p2;          // cells
    float sc[6];         // score_diff, progress, p1_mud, p2_mud, p1_score, p2_score (flat_encoder.rs:114-123)
    int maze_id;
};

template <int NW>
__device__ inline void leaf_features(const ar::State<NW>& st, const ar::Board& b, int hw, LeafFeat& f) {
    f.p1 = st.p1;
    f.p2 = st.p2;
    f.sc[0] = st.s1 - st.s2;
    f.sc[1] = b.max_turns > 0 ? (float)st.turn / (float)b.max_turns : 0.0f;
    f.sc[2] = (float)st.m1 / 10.0f;
    f.sc[3] = (float)st.m2 / 10.0f;
    f.sc[4] = st.s1 / 10.0f;
    f.sc[5] = st.s2 / 10.0f;
    f.maze_id = (int)(b.maze_off / (uint32_t)(hw * 4));
}

// acc[l] += sum_k wt[k*H + n] * act[l*ld + k]   (act in LDS, broadcast reads)
template <int L>
__device__ inline void dense_acc(const float* __restrict__ wt, int K, int H, int n, const float* act, int ld,
                                 float* acc) {
    for (int k = 0; k < K; k += 4) {
        const float w0 = wt[(size_t)(k + 0) * H + n], w1 = wt[(size_t)(k + 1) * H + n];
        const float w2 = wt[(size_t)(k + 2) * H + n], w3 = wt[(size_t)(k + 3) * H + n];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float4 a = *(const float4*)(act + (size_t)l * ld + k);
            acc[l] = fmaf(w3, a.w, fmaf(w2, a.z, fmaf(w1, a.y, fmaf(w0, a.x, acc[l]))));
        }
    }
}

// The same product on the matrix cores for a 32-leaf tile: out[32][H] = relu(bias + act[32][K] x wt[K][H]).
// v_mfma_f32_32x32x2_f32 accumulates each output element as the k-ordered chain
// fma(a_k1, b_k1, fma(a_k0, b_k0, c)) with one rounding per product (cdna_hip_programming.md, "FP32-input
// MFMA"), i.e. bit for bit what dense_acc computes, at twice the VALU rate and with one VGPR per operand.
// Each wavefront owns column tiles of 32; two tiles share one pass over K (one LDS read of the
// activations feeds both, two independent accumulator chains cover the weight loads' latency).
// Operand lanes: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31];
// result register v of lane l is out[(v & 3) + 8 (v >> 2) + 4 (l >> 5)][l & 31].
typedef float f32x16 __attribute__((ext_vector_type(16)));
// one pass over K for two column tiles; weights run eight k-steps ahead of the MFMAs in a register ring
__device__ inline void mfma_tile_pair(const float* ap, const float* bp0, const float* bp1, int K, int H, f32x16& c0,
                                      f32x16& c1) {
    float wa[8], wb[8], na[8], nb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wa[j] = 2 * j < K ? bp0[(size_t)(2 * j) * H] : 0.0f;
        wb[j] = 2 * j < K ? bp1[(size_t)(2 * j) * H] : 0.0f;
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        const bool more = k0 + 16 < K;
        if (more) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 16 + 2 * j;
                na[j] = k < K ? bp0[(size_t)k * H] : 0.0f;
                nb[j] = k < K ? bp1[(size_t)k * H] : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (k0 + 2 * j < K) {  // K is even and wave-uniform
                const float a = ap[k0 + 2 * j];
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wa[j], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j], c1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            wa[j] = na[j];
            wb[j] = nb[j];
        }
    }
}
// `out` may be `act` itself when every wavefront has at most one tile pair (H <= 64 * waves): the block
// then synchronises between the last read of `act` and the first write. All threads must call this.
__device__ inline void dense_mfma32_relu(const float* __restrict__ wt, const float* __restrict__ bias, int K, int H,
                                         const float* act, int ld, float* out, int tid, int n_threads) {
    const int wave = tid >> 6, n_waves = n_threads >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const float* ap = act + (size_t)r * ld + h;
    const bool in_place = out == act;
    for (int n0 = wave * 64; n0 < H || in_place; n0 += n_waves * 64) {
        const bool has = n0 < H;
        const bool two = n0 + 32 < H;  // wave-uniform
        f32x16 c0, c1;
        if (has) {
            const float* bp0 = wt + (size_t)h * H + n0 + r;
            const float* bp1 = bp0 + (two ? 32 : 0);
            const float b0 = bias[n0 + r], b1 = bias[n0 + (two ? 32 : 0) + r];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                c0[v] = b0;
                c1[v] = b1;
            }
            mfma_tile_pair(ap, bp0, bp1, K, H, c0, c1);
        }
        if (in_place) __syncthreads();
        if (has) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v & 3) + 8 * (v >> 2) + 4 * h;
                out[(size_t)i * ld + n0 + r] = fmaxf(c0[v], 0.0f);
                if (two) out[(size_t)i * ld + n0 + 32 + r] = fmaxf(c1[v], 0.0f);
            }
        }
        if (in_place) break;
    }
}

// Head rows for a 32-leaf tile on v_mfma_f32_16x16x4_f32: out[32][n_out <= 16] = bias + act[32][K] x w[n_out][K]^T,
// wavefront w takes leaves 16w .. 16w+15. Operand lanes: A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15]; result register v of lane l is out[4 (l >> 4) + v][l & 15].
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline void heads_mfma16(const float* __restrict__ w, const float* __restrict__ bias, int n_out, int K,
                                    const float* act, int ld, float* out, int tid, int n_tiles = 2) {
    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    if (wave >= n_tiles) return;
    const bool col = r < n_out;
    const float* ap = act + (size_t)(wave * 16 + r) * ld + q;
    const float* bp = w + (size_t)(col ? r : 0) * K + q;
    const float b = col ? bias[r] : 0.0f;
    f32x4 c = {b, b, b, b};
#pragma unroll 8
    for (int k = 0; k < K; k += 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k], col ? bp[k] : 0.0f, c, 0, 0, 0);
    if (col)
#pragma unroll
        for (int v = 0; v < 4; ++v) out[(wave * 16 + 4 * q + v) * n_out + r] = c[v];
}

// 32 columns of one weight row, as eight 16-byte loads
__device__ inline void row_load(float4* acc, const float* __restrict__ row) {
    const float4* p = (const float4*)row;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = p[j];
}
__device__ inline void row_acc(float4* acc, const float4* t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j].x += t[j].x;
        acc[j].y += t[j].y;
        acc[j].z += t[j].z;
        acc[j].w += t[j].w;
    }
}
__device__ inline void row_add(float4* acc, const float* __restrict__ row) {
    const float4* p = (const float4*)row;
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = p[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j].x += t[j].x;
        acc[j].y += t[j].y;
        acc[j].z += t[j].z;
        acc[j].w += t[j].w;
    }
}
__device__ inline void row_fma(float4* acc, float s, const float* __restrict__ row) {
    const float4* p = (const float4*)row;
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = p[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j].x = fmaf(s, t[j].x, acc[j].x);
        acc[j].y = fmaf(s, t[j].y, acc[j].y);
        acc[j].z = fmaf(s, t[j].z, acc[j].z);
        acc[j].w = fmaf(s, t[j].w, acc[j].w);
    }
}

// the MLP kernel keeps one activation buffer when a wavefront's accumulators hold its share of a layer
__host__ __device__ inline bool mlp_in_place(int H) { return (H & 31) == 0 && H <= 64 * (NTHREADS / 64); }
__host__ __device__ inline size_t mlp_smem_bytes(int H) {
    return (size_t)(mlp_in_place(H) ? 1 : 2) * TILE_MLP * (H + 4) * 4;
}

__device__ inline void softmax5(const float* l, float* p) {
    float mx = l[0];
    for (int i = 1; i < 5; ++i) mx = fmaxf(mx, l[i]);
    float e[5], s = 0.0f;
    for (int i = 0; i < 5; ++i) {
        e[i] = expf(l[i] - mx);
        s += e[i];
    }
    for (int i = 0; i < 5; ++i) p[i] = e[i] / s;
}
__device__ inline float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// per-maze first-layer constant: cmaze[m][n] = b1[n] + sum_{i < hw*4} maze_val(i) * w1t[i][n]
// `ids` (optional): the mazes to (re)compute, one block each; without it block m computes maze m
__global__ void k_maze_const(const float* w1t, const float* b1, int H, int hw, const uint8_t* maze_pool, int n_mazes,
                             float* cmaze, const uint32_t* ids = nullptr) {
    if ((int)blockIdx.x >= n_mazes) return;
    const int m = ids ? (int)ids[blockIdx.x] : (int)blockIdx.x;
    const uint8_t* cost = maze_pool + (size_t)m * hw * 4;
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        float acc = b1[n];
        for (int i = 0; i < hw * 4; ++i) {
            const uint8_t c = cost[i];
            const float v = c ? (float)c / 10.0f : -1.0f;
            acc = fmaf(v, w1t[(size_t)i * H + n], acc);
        }
        cmaze[(size_t)m * H + n] = acc;
    }
}

// ---- PyRatMLP ---------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NTHREADS) k_mlp(NetDev net, const ar::LeafReq<NW>* q, const uint32_t* qcount,
                                                  uint32_t n_fixed, const char* boards, size_t board_stride,
                                                  ar::EvalOut* out, float* logits) {
    constexpr int L = TILE_MLP;
    extern __shared__ float smem[];
    const int H = net.H, hw = net.hw, ld = H + 4;
    float* a1 = smem;  // [L][ld]
    // second-layer output: written over a1 when the MFMA path holds a whole layer in accumulators
    // (mlp_smem_bytes sizes the allocation to match), which halves the LDS a block needs
    float* a2 = mlp_in_place(H) ? a1 : smem + (size_t)L * ld;
    __shared__ LeafFeat feat[L];
    __shared__ unsigned long long cheese[L][4];
    __shared__ float hl[L * 12];
    const uint32_t n = qcount ? *qcount : n_fixed;
    const uint32_t base = blockIdx.x * L;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (uint32_t)L ? (n - base) : (uint32_t)L);
    const int tid = threadIdx.x;
    if (tid < L) {
        const int l = tid < cnt ? tid : 0;
        const ar::LeafReq<NW>& r = q[base + l];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)r.slot * board_stride);
        leaf_features<NW>(r.st, b, hw, feat[tid]);
        for (int k = 0; k < 4; ++k) cheese[tid][k] = k < NW ? r.st.cheese[k] : 0ULL;
    }
    __syncthreads();
    const float* w1p1 = net.w1t + (size_t)(hw * 4) * H;
    const float* w1p2 = net.w1t + (size_t)(hw * 5) * H;
    const float* w1ch = net.w1t + (size_t)(hw * 6) * H;
    const float* w1sc = net.w1t + (size_t)(hw * 7) * H;
    const bool wide = (H & 31) == 0;  // block-uniform: the paths that work on 32-column pieces
    if (wide) {
        // first layer: the observation is one-hot except six scalars, so a leaf's pre-activation is a sum
        // of weight rows (maze constant, p1 cell, p2 cell, one row per cheese, in that order) plus six
        // scaled rows. One thread sums 32 columns of one leaf: every row is eight independent 16-byte
        // loads, so the row fetches overlap instead of queueing behind each other.
        const int T = H >> 5;
        for (int item = tid; item < L * T; item += NTHREADS) {
            const int l = item / T, n0 = (item - l * T) << 5;
            const LeafFeat& f = feat[l];
            float4 acc[8], t[8], u[8];
            row_load(acc, net.cmaze + (size_t)f.maze_id * H + n0);
            row_load(t, w1p1 + (size_t)f.p1 * H + n0);
            row_load(u, w1p2 + (size_t)f.p2 * H + n0);
            row_acc(acc, t);
            // cheese rows, each fetched while the previous one is added
            bool have = false;
            for (int wd = 0; wd < 4; ++wd) {
                unsigned long long m = cheese[l][wd];
                while (m) {
                    const int c = __ffsll((long long)m) - 1 + 64 * wd;
                    m &= m - 1;
                    row_load(t, w1ch + (size_t)c * H + n0);
                    row_acc(acc, u);  // the row fetched one step earlier (p2 the first time)
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] = t[j];
                    have = true;
                }
            }
            (void)have;
            row_acc(acc, u);
            for (int s = 0; s < 6; ++s) row_fma(acc, f.sc[s], w1sc + (size_t)s * H + n0);
            float4* dst = (float4*)(a1 + (size_t)l * ld + n0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                dst[j] = make_float4(fmaxf(acc[j].x, 0.0f), fmaxf(acc[j].y, 0.0f), fmaxf(acc[j].z, 0.0f),
                                     fmaxf(acc[j].w, 0.0f));
        }
    } else {
        for (int nn = tid; nn < H; nn += NTHREADS) {
            float ws[6];
            for (int s = 0; s < 6; ++s) ws[s] = w1sc[(size_t)s * H + nn];
            for (int l = 0; l < L; ++l) {
                const LeafFeat& f = feat[l];
                float acc = net.cmaze[(size_t)f.maze_id * H + nn];
                acc += w1p1[(size_t)f.p1 * H + nn];
                acc += w1p2[(size_t)f.p2 * H + nn];
                for (int wd = 0; wd < 4; ++wd) {
                    unsigned long long m = cheese[l][wd];
                    while (m) {
                        const int c = __ffsll((long long)m) - 1 + 64 * wd;
                        m &= m - 1;
                        acc += w1ch[(size_t)c * H + nn];
                    }
                }
                for (int s = 0; s < 6; ++s) acc = fmaf(f.sc[s], ws[s], acc);
                a1[(size_t)l * ld + nn] = fmaxf(acc, 0.0f);
            }
        }
    }
    __syncthreads();
    if (wide) {
        static_assert(L == 32, "the MFMA tile is 32 leaves");
        dense_mfma32_relu(net.w2t, net.b2, H, H, a1, ld, a2, tid, NTHREADS);
    } else {
        for (int nn = tid; nn < H; nn += NTHREADS) {
            float acc[L];
#pragma unroll
            for (int l = 0; l < L; ++l) acc[l] = net.b2[nn];
            dense_acc<L>(net.w2t, H, H, nn, a1, ld, acc);
#pragma unroll
            for (int l = 0; l < L; ++l) a2[(size_t)l * ld + nn] = fmaxf(acc[l], 0.0f);
        }
    }
    __syncthreads();
    // heads: 12 dot products per leaf
    if (wide) {
        heads_mfma16(net.wh, net.bh, 12, H, a2, ld, hl, tid);
    } else {
        for (int idx = tid; idx < L * 12; idx += NTHREADS) {
            const int l = idx / 12, o = idx % 12;
            const float* w = net.wh + (size_t)o * H;
            float acc = net.bh[o];
            for (int k = 0; k < H; ++k) acc = fmaf(w[k], a2[(size_t)l * ld + k], acc);
            hl[l * 12 + o] = acc;
        }
    }
    __syncthreads();
    if (tid < cnt) {
        const float* h = hl + tid * 12;
        ar::EvalOut o;
        softmax5(h, o.p1);
        softmax5(h + 5, o.p2);
        o.v1 = softplusf(h[10]);
        o.v2 = softplusf(h[11]);
        out[base + tid] = o;
        if (logits)
            for (int k = 0; k < 10; ++k) logits[(size_t)(base + tid) * 10 + k] = h[k];
    }
}

// ---- PyRatMLP, all three layers on the matrix cores (hidden width a multiple of 32, at most 256) ----
// One block evaluates 32*MT leaves. The first layer is the same k-ordered sum as in k_mlp written as a
// dense product over the non-maze part of the observation: x = [p1 one-hot | p2 one-hot | cheese mask |
// six scalars] (K1 = 3 hw + 6), accumulator initialised with the per-maze constant. Products with x = 0
// add nothing and x = 1 adds the weight row itself, so the chain is the one k_mlp's row sums compute --
// but the weights stream through once per block (K1 rows) instead of once per leaf (one row per set
// feature), which is what bounded k_mlp: ~19 KB of L2 reads per leaf against ~2.4 KB here.
// The operand x is generated in registers from the leaf's cells and cheese mask; nothing is staged.
// R = k-steps the weight loads run ahead of the MFMAs (register ring of 2 x 2R values): 8 covers the L2
// latency when a SIMD holds several wavefronts (the MLP), the CNN's single wavefront per SIMD needs more
template <int MT, int R = 8, class AFn>
__device__ inline void mfma_pass(const float* bp0, bool two, int K, int H, int h, AFn a_of, f32x16 (&c)[MT][2]) {
    const float* bp1 = bp0 + (two ? 32 : 0);
    float wa[R], wb[R], na[R], nb[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = 2 * j;
        wa[j] = k + h < K ? bp0[(size_t)k * H] : 0.0f;
        wb[j] = k + h < K ? bp1[(size_t)k * H] : 0.0f;
    }
    // whole chunks of R steps are straight-line code (a guard per step would start a new basic block, and
    // with it conservative s_waitcnt's, in front of every MFMA); only the last, partial chunk is guarded.
    // While the chunk after the current one lies wholly inside K its loads carry no guard either: a guarded
    // load is a branch around the instruction, and the compiler then waits for every load in flight before
    // the first MFMA of the chunk instead of counting them.
    const int K_main = K - K % (2 * R);
    int k0 = 0;
    for (; k0 + 4 * R <= K; k0 += 2 * R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int k = k0 + 2 * R + 2 * j;
            na[j] = bp0[(size_t)k * H];
            nb[j] = bp1[(size_t)k * H];
        }
        // the loads go out before the chunk's MFMAs (left to itself the scheduler sinks them to the end of
        // the chunk, where the next iteration waits for them at once)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < R; ++j) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float a = a_of(k0 + 2 * j, t);
                c[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wa[j], c[t][0], 0, 0, 0);
                c[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j], c[t][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            wa[j] = na[j];
            wb[j] = nb[j];
        }
    }
    for (; k0 < K_main; k0 += 2 * R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {  // next chunk (a load past K yields 0 without touching memory)
            const int k = k0 + 2 * R + 2 * j;
            na[j] = k + h < K ? bp0[(size_t)k * H] : 0.0f;
            nb[j] = k + h < K ? bp1[(size_t)k * H] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float a = a_of(k0 + 2 * j, t);
                c[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wa[j], c[t][0], 0, 0, 0);
                c[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j], c[t][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            wa[j] = na[j];
            wb[j] = nb[j];
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = k0 + 2 * j;
        if (k < K) {  // wave-uniform
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float a = a_of(k, t);
                c[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wa[j], c[t][0], 0, 0, 0);
                c[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[j], c[t][1], 0, 0, 0);
            }
        }
    }
}

// (amdgpu_num_vgpr: with the limit the compiler keeps the accumulators in ordinary vector registers and needs 148 in all,
// without it 130 + 64 accumulator registers = 196 (200 allocated): two such wavefronts leave a SIMD 112 of its 512
// registers, and a tree-reuse wavefront (k_advance: 128) on a SIMD then kept the CU's second evaluator workgroup out for
// as long as it ran. A/B of the two builds: +1.3 % simulations/s.)
template <int NW, int MT, int FL>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_num_vgpr(128))) k_mlp_mfma(NetDev net, const ar::LeafReq<NW>* q, const uint32_t* qcount,
                                                       uint32_t n_fixed, const char* boards, size_t board_stride,
                                                       ar::EvalOut* out, float* logits) {
    constexpr int L = 32 * MT;
    extern __shared__ float smem[];
    const int H = net.H, hw = net.hw, ld = H + 4;
    float* act = smem;  // [L][ld], both hidden layers in turn
    __shared__ LeafFeat feat[L];
    __shared__ unsigned long long cheese[L][4];
    __shared__ float hl[L * 12];
    const uint32_t n = qcount ? *qcount : n_fixed;
    const uint32_t base = blockIdx.x * L;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (uint32_t)L ? (n - base) : (uint32_t)L);
    const int tid = threadIdx.x;
    if (tid < L) {
        const int l = tid < cnt ? tid : 0;
        const ar::LeafReq<NW>& r = q[base + l];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)r.slot * board_stride);
        leaf_features<NW>(r.st, b, hw, feat[tid]);
        for (int k = 0; k < 4; ++k) cheese[tid][k] = k < NW ? r.st.cheese[k] : 0ULL;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int n0 = wave * 64;
    const bool has = n0 < H, two = n0 + 32 < H;  // wave-uniform; H <= 64 * waves
    const int K1 = 3 * hw + 6, K1e = (K1 + 1) & ~1;
    // The first layer's operand x, one row of K1e floats per leaf, laid out in `act` (which nothing else uses
    // until the layer's results are written there) when it fits: the k-loop then is the second layer's loop,
    // one LDS read per MFMA pair. Forming x in registers from the leaf's cells and cheese mask (the path kept
    // for boards whose K1 exceeds a row of `act`) costs a dozen vector instructions and several branches per
    // k-step and held the first layer at a third of the matrix pipe's rate.
    // FL (first layer): 0 = x staged, the product over all K1 rows; 1 = x formed in registers (boards whose K1e exceeds a
    // row of `act`; AR_MLP_FL=1); 2 = the p1 / p2 rows summed into the start of the chain, cheese and scalars on the matrix
    // cores (below)
    constexpr bool staged = FL == 0;
    if (FL == 2) {
        // x is one-hot over the p1 cells and over the p2 cells: the first 2 hw steps of a leaf's chain add exactly one
        // weight row each (fma(1, w, acc) = acc + w, fma(0, w, acc) = acc), so the chain can start from
        //   (maze constant + row of the p1 cell) + row of the p2 cell
        // -- the same bits -- and the matrix cores only take the hw + 6 cheese and scalar rows: 28 k-steps instead of 77
        // at 7x7. The starts are summed 16 bytes at a time into `act` and picked up from there in accumulator layout.
        const int H4 = H >> 2;
        for (int item = tid; item < L * H4; item += NTHREADS) {
            const int l = item / H4, c4 = item - l * H4;
            const LeafFeat& f = feat[l];
            const float4 a = ((const float4*)(net.cmaze + (size_t)f.maze_id * H))[c4];
            const float4 b = ((const float4*)(net.w1t + (size_t)(4 * hw + f.p1) * H))[c4];
            const float4 c = ((const float4*)(net.w1t + (size_t)(5 * hw + f.p2) * H))[c4];
            float4 sum;
            sum.x = (a.x + b.x) + c.x;
            sum.y = (a.y + b.y) + c.y;
            sum.z = (a.z + b.z) + c.z;
            sum.w = (a.w + b.w) + c.w;
            *(float4*)(act + (size_t)l * ld + 4 * c4) = sum;
        }
        __syncthreads();
    }
    if (staged) {
        constexpr int TPL = NTHREADS / L;  // threads per leaf
        const int l = tid / TPL;
        const LeafFeat& f = feat[l];
        float* x = act + (size_t)l * ld;
        for (int kk = tid % TPL; kk < K1e; kk += TPL) {
            float v;
            if (kk < hw) v = kk == f.p1 ? 1.0f : 0.0f;
            else if (kk < 2 * hw) v = kk - hw == f.p2 ? 1.0f : 0.0f;
            else if (kk < 3 * hw) {
                const int bit = kk - 2 * hw;
                v = (cheese[l][bit >> 6] >> (bit & 63)) & 1ULL ? 1.0f : 0.0f;
            } else v = kk < K1 ? f.sc[kk - 3 * hw] : 0.0f;
            x[kk] = v;
        }
        __syncthreads();
    }
    f32x16 c[MT][2];
    if (FL == 2) {
        if (has) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                    c[t][0][v] = act[(size_t)i * ld + n0 + r];
                    c[t][1][v] = act[(size_t)i * ld + n0 + (two ? 32 : 0) + r];
                }
                __builtin_amdgcn_sched_barrier(0);  // (one row tile's 32 reads at a time)
            }
        }
        __syncthreads();  // the starts are in registers: `act` now takes the operand
        {
            // x over [cheese | six scalars], K2e floats per leaf at the head of its row of `act`
            const int K2 = hw + 6, K2e = (K2 + 1) & ~1;
            constexpr int TPL = NTHREADS / L;  // threads per leaf
            const int l = tid / TPL;
            const LeafFeat& f = feat[l];
            float* x = act + (size_t)l * ld;
            for (int kk = tid % TPL; kk < K2e; kk += TPL) {
                float v;
                if (kk < hw) v = (cheese[l][kk >> 6] >> (kk & 63)) & 1ULL ? 1.0f : 0.0f;
                else v = kk < K2 ? f.sc[kk - hw] : 0.0f;
                x[kk] = v;
            }
        }
        __syncthreads();
        if (has) {
            const float* xp = act + (size_t)r * ld + h;
            auto x_lds = [&](int k, int t) -> float { return xp[(size_t)(32 * t) * ld + k]; };
            mfma_pass<MT, 8>(net.w1t + (size_t)(6 * hw + h) * H + n0 + r, two, hw + 6, H, h, x_lds, c);
        }
    } else if (has) {
        // first layer
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                const float* cm = net.cmaze + (size_t)feat[i].maze_id * H + n0 + r;
                c[t][0][v] = cm[0];
                c[t][1][v] = cm[two ? 32 : 0];
            }
        const float* w1 = net.w1t + (size_t)(4 * hw + h) * H + n0 + r;
        if (staged) {
            const float* xp = act + (size_t)r * ld + h;
            auto x_lds = [&](int k, int t) -> float { return xp[(size_t)(32 * t) * ld + k]; };
            mfma_pass<MT, 8>(w1, two, K1, H, h, x_lds, c);
        } else {
            int p1[MT], p2[MT];
            unsigned long long ch[MT][NW];
            float sc[MT][6];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const LeafFeat& f = feat[32 * t + r];
                p1[t] = f.p1;
                p2[t] = f.p2;
                for (int s6 = 0; s6 < 6; ++s6) sc[t][s6] = f.sc[s6];
                for (int w = 0; w < NW; ++w) ch[t][w] = cheese[32 * t + r][w];
            }
            auto x_of = [&](int k, int t) -> float {
                const int kk = k + h;
                if (kk < hw) return kk == p1[t] ? 1.0f : 0.0f;
                if (kk < 2 * hw) return kk - hw == p2[t] ? 1.0f : 0.0f;
                if (kk < 3 * hw) {
                    const int bit = kk - 2 * hw;
                    unsigned long long word = ch[t][0];
#pragma unroll
                    for (int w = 1; w < NW; ++w) word = (bit >> 6) == w ? ch[t][w] : word;
                    return (word >> (bit & 63)) & 1ULL ? 1.0f : 0.0f;
                }
                const int s6 = kk - 3 * hw;
                float v = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; ++j) v = s6 == j ? sc[t][j] : v;
                return v;
            };
            mfma_pass<MT, 8>(w1, two, K1, H, h, x_of, c);
        }
    }
    if (FL != 1) __syncthreads();  // every wavefront is done reading x before the results go over it
    if (has) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                act[(size_t)i * ld + n0 + r] = fmaxf(c[t][0][v], 0.0f);
                if (two) act[(size_t)i * ld + n0 + 32 + r] = fmaxf(c[t][1][v], 0.0f);
            }
    }
    __syncthreads();
    if (has) {
        // second layer, result held in the accumulators until every wavefront is done reading `act`
        const float b0 = net.b2[n0 + r], b1 = net.b2[n0 + (two ? 32 : 0) + r];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                c[t][0][v] = b0;
                c[t][1][v] = b1;
            }
        const float* ap = act + (size_t)r * ld + h;
        auto a_of = [&](int k, int t) -> float { return ap[(size_t)(32 * t) * ld + k]; };
        mfma_pass<MT, 8>(net.w2t + (size_t)h * H + n0 + r, two, H, H, h, a_of, c);
    }
    __syncthreads();
    if (has) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                act[(size_t)i * ld + n0 + r] = fmaxf(c[t][0][v], 0.0f);
                if (two) act[(size_t)i * ld + n0 + 32 + r] = fmaxf(c[t][1][v], 0.0f);
            }
    }
    __syncthreads();
    heads_mfma16(net.wh, net.bh, 12, H, act, ld, hl, tid, 2 * MT);
    __syncthreads();
    if (tid < cnt) {
        const float* hh = hl + tid * 12;
        ar::EvalOut o;
        softmax5(hh, o.p1);
        softmax5(hh + 5, o.p2);
        o.v1 = softplusf(hh[10]);
        o.v2 = softplusf(hh[11]);
        out[base + tid] = o;
        if (logits)
            for (int k = 0; k < 10; ++k) logits[(size_t)(base + tid) * 10 + k] = hh[k];
    }
}

__host__ __device__ inline bool mlp_all_mfma(int H) { return (H & 31) == 0 && H <= 64 * (NTHREADS / 64); }
static const int MLP_MFMA_MT = 2;  // 64 leaves per block

// ---- SymmetricMLP -----------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NTHREADS) k_symmetric(NetDev net, const ar::LeafReq<NW>* q, const uint32_t* qcount,
                                                        uint32_t n_fixed, const char* boards, size_t board_stride,
                                                        ar::EvalOut* out, float* logits) {
    constexpr int L = TILE_SYM;
    extern __shared__ float smem[];
    const int H = net.H, hw = net.hw, ld = H + 4;
    float* sh = smem;                        // shared encoding   [L][ld]
    float* pe = sh + (size_t)L * ld;         // player encoding   [L][ld]
    float* t1 = pe + (size_t)L * ld;         // trunk layer 1     [L][ld]
    float* h0 = t1 + (size_t)L * ld;         // h1, then h2       [2][L][ld]
    __shared__ LeafFeat feat[L];
    __shared__ unsigned long long cheese[L][4];
    const uint32_t n = qcount ? *qcount : n_fixed;
    const uint32_t base = blockIdx.x * L;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (uint32_t)L ? (n - base) : (uint32_t)L);
    const int tid = threadIdx.x;
    if (tid < L) {
        const int l = tid < cnt ? tid : 0;
        const ar::LeafReq<NW>& r = q[base + l];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)r.slot * board_stride);
        leaf_features<NW>(r.st, b, hw, feat[tid]);
        for (int k = 0; k < 4; ++k) cheese[tid][k] = k < NW ? r.st.cheese[k] : 0ULL;
    }
    __syncthreads();
    // shared encoder input = [maze hw*4 | cheese hw | progress]
    const float* wch = net.w1t + (size_t)(hw * 4) * H;
    const float* wpr = net.w1t + (size_t)(hw * 5) * H;
    for (int nn = tid; nn < H; nn += NTHREADS) {
        const float wprog = wpr[nn];
        for (int l = 0; l < L; ++l) {
            const LeafFeat& f = feat[l];
            float acc = net.cmaze[(size_t)f.maze_id * H + nn];
            for (int wd = 0; wd < 4; ++wd) {
                unsigned long long m = cheese[l][wd];
                while (m) {
                    const int c = __ffsll((long long)m) - 1 + 64 * wd;
                    m &= m - 1;
                    acc += wch[(size_t)c * H + nn];
                }
            }
            acc = fmaf(f.sc[1], wprog, acc);
            sh[(size_t)l * ld + nn] = fmaxf(acc, 0.0f);
        }
    }
    __syncthreads();
    for (int p = 0; p < 2; ++p) {
        // player encoder input = [pos one-hot hw | mud | score]
        for (int nn = tid; nn < H; nn += NTHREADS) {
            const float wm = net.wpt[(size_t)hw * H + nn], wsco = net.wpt[(size_t)(hw + 1) * H + nn], bb = net.bp[nn];
            for (int l = 0; l < L; ++l) {
                const LeafFeat& f = feat[l];
                float acc = bb + net.wpt[(size_t)(p == 0 ? f.p1 : f.p2) * H + nn];
                acc = fmaf(f.sc[2 + p], wm, acc);
                acc = fmaf(f.sc[4 + p], wsco, acc);
                pe[(size_t)l * ld + nn] = fmaxf(acc, 0.0f);
            }
        }
        __syncthreads();
        for (int nn = tid; nn < H; nn += NTHREADS) {
            float acc[L];
#pragma unroll
            for (int l = 0; l < L; ++l) acc[l] = net.b2[nn];
            dense_acc<L>(net.w2t, H, H, nn, sh, ld, acc);
            dense_acc<L>(net.w2t + (size_t)H * H, H, H, nn, pe, ld, acc);
#pragma unroll
            for (int l = 0; l < L; ++l) t1[(size_t)l * ld + nn] = fmaxf(acc[l], 0.0f);
        }
        __syncthreads();
        float* hp = h0 + (size_t)p * L * ld;
        for (int nn = tid; nn < H; nn += NTHREADS) {
            float acc[L];
#pragma unroll
            for (int l = 0; l < L; ++l) acc[l] = net.b3[nn];
            dense_acc<L>(net.w3t, H, H, nn, t1, ld, acc);
#pragma unroll
            for (int l = 0; l < L; ++l) hp[(size_t)l * ld + nn] = fmaxf(acc[l], 0.0f);
        }
        __syncthreads();
    }
    // heads on cat(h_i, h1 + h2): 6 rows (policy 5, value 1) x 2 players
    float* hl = sh;  // reuse [L][12]
    for (int idx = tid; idx < L * 12; idx += NTHREADS) {
        const int l = idx / 12, r = idx % 12, p = r / 6, o = r % 6;
        const float* w = net.wh + (size_t)o * (2 * H);
        const float* hi = h0 + (size_t)p * L * ld + (size_t)l * ld;
        const float* ha = h0 + (size_t)l * ld;
        const float* hb = h0 + (size_t)L * ld + (size_t)l * ld;
        float acc = net.bh[o];
        for (int k = 0; k < H; ++k) acc = fmaf(w[k], hi[k], acc);
        for (int k = 0; k < H; ++k) acc = fmaf(w[H + k], ha[k] + hb[k], acc);
        hl[l * 12 + r] = acc;
    }
    __syncthreads();
    if (tid < cnt) {
        const float* h = hl + tid * 12;
        ar::EvalOut o;
        softmax5(h, o.p1);
        softmax5(h + 6, o.p2);
        o.v1 = softplusf(h[5]);
        o.v2 = softplusf(h[11]);
        out[base + tid] = o;
        if (logits)
            for (int k = 0; k < 5; ++k) {
                logits[(size_t)(base + tid) * 10 + k] = h[k];
                logits[(size_t)(base + tid) * 10 + 5 + k] = h[6 + k];
            }
    }
}

// ---- the same with BOTH players of a leaf in one pass ------------------------------------------------------------
// k_symmetric_mfma streams the trunk's weights (the bulk: (2H + H) x H) through once per player per 32-leaf block,
// ~49 KB of L2 reads per leaf. Here a block's rows are (player, leaf) pairs -- 64 rows for 32 leaves -- so the player
// encoder, both trunk layers and the heads run once over two row tiles and the trunk weights are read once per block:
// half the L2 traffic, four independent accumulator chains per wavefront. Per output the k-ordered chain is unchanged
// (same bits as the FMA loops). LDS: ONE [64][ld] buffer (67 KB at H = 256: two blocks per CU) that holds in turn the
// shared encoder's operand, the shared encoding (rows 0..31, read by the first half of trunk 1), the player encodings
// (read by the second half), trunk 1 and h; results wait in the accumulators until the buffer's readers are done.
template <int NW>
__global__ void __launch_bounds__(NTHREADS) k_symmetric_mfma2(NetDev net, const ar::LeafReq<NW>* q, const uint32_t* qcount,
                                                              uint32_t n_fixed, const char* boards, size_t board_stride,
                                                              ar::EvalOut* out, float* logits) {
    constexpr int L = 32;
    extern __shared__ float smem[];
    const int H = net.H, hw = net.hw, ld = H + 4;
    float* bufB = smem;  // [2 L][ld]: in turn the shared encoder's operand and output (rows 0..31), the player encodings,
                         // trunk 1 and h (row = player * L + leaf)
    __shared__ LeafFeat feat[L];
    __shared__ unsigned long long cheese[L][4];
    __shared__ float hl[L * 12];
    const uint32_t n = qcount ? *qcount : n_fixed;
    const uint32_t base = blockIdx.x * L;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (uint32_t)L ? (n - base) : (uint32_t)L);
    const int tid = threadIdx.x;
    if (tid < L) {
        const int l = tid < cnt ? tid : 0;
        const ar::LeafReq<NW>& r = q[base + l];
        const ar::Board& b = *(const ar::Board*)(boards + (size_t)r.slot * board_stride);
        leaf_features<NW>(r.st, b, hw, feat[tid]);
        for (int k = 0; k < 4; ++k) cheese[tid][k] = k < NW ? r.st.cheese[k] : 0ULL;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int n0 = wave * 64;
    const bool has = n0 < H, two = n0 + 32 < H;  // wave-uniform; H <= 64 * waves
    const int c1 = two ? 32 : 0;
    f32x16 c[2][2];  // [row tile = player][column tile]
    auto store_relu2 = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                bufB[(size_t)i * ld + n0 + r] = fmaxf(c[t][0][v], 0.0f);
                if (two) bufB[(size_t)i * ld + n0 + 32 + r] = fmaxf(c[t][1][v], 0.0f);
            }
    };
    auto init_bias2 = [&](const float* bias) {
        const float b0 = bias[n0 + r], b1 = bias[n0 + c1 + r];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                c[t][0][v] = b0;
                c[t][1][v] = b1;
            }
    };
    // ---- shared encoder (one row tile: the 32 leaves): operand x = [cheese | progress] staged at the head of rows
    // 0..31 (the k-loop then reads it like any activation: one LDS read per MFMA pair, no per-step selects or branches)
    const int Ks = hw + 1, Kse = (Ks + 1) & ~1;
    {
        constexpr int TPL = NTHREADS / L;
        const int l = tid / TPL;
        float* x = bufB + (size_t)l * ld;
        for (int kk = tid % TPL; kk < Kse; kk += TPL)
            x[kk] = kk < hw ? ((cheese[l][kk >> 6] >> (kk & 63)) & 1ULL ? 1.0f : 0.0f) : kk == hw ? feat[l].sc[1] : 0.0f;
    }
    __syncthreads();
    f32x16 cs[1][2];
    if (has) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = (v & 3) + 8 * (v >> 2) + 4 * h;
            const float* cm = net.cmaze + (size_t)feat[i].maze_id * H + n0 + r;
            cs[0][0][v] = cm[0];
            cs[0][1][v] = cm[c1];
        }
        const float* xp = bufB + (size_t)r * ld + h;
        auto x_sh = [&](int k, int) -> float { return xp[k]; };
        mfma_pass<1, 8>(net.w1t + (size_t)(4 * hw + h) * H + n0 + r, two, Ks, H, h, x_sh, cs);
    }
    __syncthreads();  // every wavefront is done reading x
    if (has) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = (v & 3) + 8 * (v >> 2) + 4 * h;
            bufB[(size_t)i * ld + n0 + r] = fmaxf(cs[0][0][v], 0.0f);
            if (two) bufB[(size_t)i * ld + n0 + 32 + r] = fmaxf(cs[0][1][v], 0.0f);
        }
    }
    __syncthreads();
    if (has) {
        // ---- trunk 1, first half: k over the shared encoding (row r of rows 0..31 for both players)
        init_bias2(net.b2);
        const float* ap = bufB + (size_t)r * ld + h;
        auto a_sh = [&](int k, int) -> float { return ap[k]; };
        mfma_pass<2, 8>(net.w2t + (size_t)h * H + n0 + r, two, H, H, h, a_sh, c);
    }
    __syncthreads();  // the shared encoding has been read: the buffer takes the player encodings
    {
        // ---- player encoder, both players (row = player * L + leaf). Its input is one-hot over the player's cell plus
        // two scalars, so the k-ordered chain bias, fma(x_k, w_k, .) is: bias + row of the cell (the zero products add
        // nothing, the 1 adds the row), then the two scaled rows -- three weight rows per output instead of hw + 2 steps
        // on the matrix cores, the same bits. 16 bytes of columns per item.
        const int H4 = H >> 2;
        const float4* bias4 = (const float4*)net.bp;
        const float4* wm = (const float4*)(net.wpt + (size_t)hw * H);
        const float4* ws = (const float4*)(net.wpt + (size_t)(hw + 1) * H);
        for (int item = tid; item < 2 * L * H4; item += NTHREADS) {
            const int row = item / H4, c4 = item - row * H4, t = row / L, l = row - t * L;
            const LeafFeat& f = feat[l];
            const float4 wc = ((const float4*)(net.wpt + (size_t)(t == 0 ? f.p1 : f.p2) * H))[c4];
            const float4 b = bias4[c4], m = wm[c4], s4 = ws[c4];
            const float mud = f.sc[2 + t], score = f.sc[4 + t];
            float4 o;
            o.x = fmaxf(fmaf(score, s4.x, fmaf(mud, m.x, b.x + wc.x)), 0.0f);
            o.y = fmaxf(fmaf(score, s4.y, fmaf(mud, m.y, b.y + wc.y)), 0.0f);
            o.z = fmaxf(fmaf(score, s4.z, fmaf(mud, m.z, b.z + wc.z)), 0.0f);
            o.w = fmaxf(fmaf(score, s4.w, fmaf(mud, m.w, b.w + wc.w)), 0.0f);
            *(float4*)(bufB + (size_t)row * ld + 4 * c4) = o;
        }
    }
    __syncthreads();
    if (has) {
        // ---- trunk 1, second half: k over the player encodings
        const float* bp = bufB + (size_t)r * ld + h;
        auto a_pe = [&](int k, int t) -> float { return bp[(size_t)(32 * t) * ld + k]; };
        mfma_pass<2, 8>(net.w2t + (size_t)(H + h) * H + n0 + r, two, H, H, h, a_pe, c);
    }
    __syncthreads();
    if (has) store_relu2();
    __syncthreads();
    if (has) {
        // ---- trunk 2
        init_bias2(net.b3);
        const float* bp = bufB + (size_t)r * ld + h;
        auto a_t1 = [&](int k, int t) -> float { return bp[(size_t)(32 * t) * ld + k]; };
        mfma_pass<2, 8>(net.w3t + (size_t)h * H + n0 + r, two, H, H, h, a_t1, c);
    }
    __syncthreads();
    if (has) store_relu2();  // h: rows 0..31 player 1, rows 32..63 player 2
    __syncthreads();
    // ---- heads on cat(h_p, h_1 + h_2): wavefront = (player, 16-leaf half); rows: policy 5, value 1
    {
        const int pl = wave >> 1, half = wave & 1, rr = lane & 15, qq = lane >> 4;
        const bool col = rr < 6;
        const float* ha = bufB + (size_t)(half * 16 + rr) * ld + qq;
        const float* hb = bufB + (size_t)(32 + half * 16 + rr) * ld + qq;
        const float* hp = pl == 0 ? ha : hb;
        const float* wp = net.wh + (size_t)(col ? rr : 0) * (2 * H) + qq;
        const float b = col ? net.bh[rr] : 0.0f;
        f32x4 acc = {b, b, b, b};
#pragma unroll 8
        for (int k = 0; k < H; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[k], col ? wp[k] : 0.0f, acc, 0, 0, 0);
#pragma unroll 8
        for (int k = 0; k < H; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[k] + hb[k], col ? wp[H + k] : 0.0f, acc, 0, 0, 0);
        if (col)
#pragma unroll
            for (int v = 0; v < 4; ++v) hl[(half * 16 + 4 * qq + v) * 12 + pl * 6 + rr] = acc[v];
    }
    __syncthreads();
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

__host__ __device__ inline bool symmetric_mfma_ok(int H) { return (H & 31) == 0 && H <= 64 * (NTHREADS / 64); }

// flat observation (flat_encoder.rs:52-125), one block per position
template <int NW>
__global__ void k_encode(const ar::LeafReq<NW>* q, uint32_t n, const ar::Board* boards, const uint8_t* maze_pool,
                         float* obs, int obs_stride) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const ar::State<NW>& st = q[i].st;
    const ar::Board& b = boards[q[i].slot];
    const int hw = b.width * b.height;
    const uint8_t* cost = maze_pool + b.maze_off;
    float* o = obs + (size_t)i * obs_stride;
    for (int k = threadIdx.x; k < hw * 4; k += blockDim.x) o[k] = cost[k] ? (float)cost[k] / 10.0f : -1.0f;
    for (int k = threadIdx.x; k < hw; k += blockDim.x) {
        o[hw * 4 + k] = k == st.p1 ? 1.0f : 0.0f;
        o[hw * 5 + k] = k == st.p2 ? 1.0f : 0.0f;
        o[hw * 6 + k] = ar::st_has_cheese(st, k) ? 1.0f : 0.0f;
    }
    if (threadIdx.x == 0) {
        float* s = o + hw * 7;
        s[0] = st.s1 - st.s2;
        s[1] = b.max_turns > 0 ? (float)st.turn / (float)b.max_turns : 0.0f;
        s[2] = (float)st.m1 / 10.0f;
        s[3] = (float)st.m2 / 10.0f;
        s[4] = st.s1 / 10.0f;
        s[5] = st.s2 / 10.0f;
    }
}

}  // namespace arnet

#include "nets_cnn.h"

// ---- host object --------------------------------------------------------------------------------
struct ArNet {
    int device = 0;
    arnet::NetDev dev;
    arnet::CnnDev cnn;
    std::vector<void*> allocs;
    float* cmaze = nullptr;
    const uint8_t* bound_pool = nullptr;
    int bound_mazes = 0;
    size_t smem = 0;
    // ar_net_evaluate's own buffers (requests, boards, mazes, outputs, logits) and the maze bytes they hold
    struct Scratch {
        void* p = nullptr;
        size_t bytes = 0;
    } scratch[5];
    std::vector<uint8_t> scratch_mazes;

    ~ArNet() {
        for (void* p : allocs) hipFree(p);
        if (cmaze) hipFree(cmaze);
        for (Scratch& sc : scratch)
            if (sc.p) hipFree(sc.p);
    }
    const void* upload_raw(const void* src, size_t bytes, bool& ok) {
        void* d = nullptr;
        if (hipMalloc(&d, bytes + 16) != hipSuccess) {
            ok = false;
            return nullptr;
        }
        allocs.push_back(d);
        if (hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        return d;
    }
    const float* upload(const std::vector<float>& v, bool& ok) {
        float* d = nullptr;
        if (hipMalloc((void**)&d, v.size() * 4 + 16) != hipSuccess) {
            ok = false;
            return nullptr;
        }
        allocs.push_back(d);
        if (hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        return d;
    }
};

static int net_build(const arnet::Blob& b, ArNet* net) {
    using namespace arnet;
    std::string err;
    NetDev& d = net->dev;
    memset(&d, 0, sizeof d);
    d.arch = (int)b.arch;
    d.width = (int)b.width;
    d.height = (int)b.height;
    d.hw = d.width * d.height;
    bool ok = true;
    std::vector<float> wt, bias;
    uint32_t in = 0, out = 0;
    if (b.arch == ARCH_MLP) {
        if (!fold_linear(b, "trunk.0", "trunk.1", wt, bias, in, out, err)) return nets_fail(AR_E_BACKEND, err);
        if ((int)in != d.hw * 7 + 6) return nets_fail(AR_E_BACKEND, "MLP input width does not match the board size");
        d.H = (int)out;
        d.w1t = net->upload(wt, ok);
        d.b1 = net->upload(bias, ok);
        if (!fold_linear(b, "trunk.4", "trunk.5", wt, bias, in, out, err)) return nets_fail(AR_E_BACKEND, err);
        d.w2t = net->upload(wt, ok);
        d.b2 = net->upload(bias, ok);
        std::vector<float> wh((size_t)12 * d.H), bh(12);
        const char* heads[3] = {"policy_p1_head", "policy_p2_head", "value_head"};
        const int rows[3] = {5, 5, 2};
        int r0 = 0;
        for (int h = 0; h < 3; ++h) {
            const std::vector<float>*w = b.get(std::string(heads[h]) + ".weight"), *bi = b.get(std::string(heads[h]) + ".bias");
            if (!w || !bi || (int)w->size() != rows[h] * d.H) return nets_fail(AR_E_BACKEND, std::string("bad head ") + heads[h]);
            memcpy(&wh[(size_t)r0 * d.H], w->data(), w->size() * 4);
            memcpy(&bh[r0], bi->data(), bi->size() * 4);
            r0 += rows[h];
        }
        d.wh = net->upload(wh, ok);
        d.bh = net->upload(bh, ok);
        d.n_head = 12;
        d.Kh = d.H;
        net->smem = mlp_smem_bytes(d.H);
    } else if (b.arch == ARCH_SYMMETRIC) {
        if (!fold_linear(b, "shared_encoder.0", "shared_encoder.1", wt, bias, in, out, err)) return nets_fail(AR_E_BACKEND, err);
        if ((int)in != d.hw * 5 + 1) return nets_fail(AR_E_BACKEND, "SymmetricMLP input width does not match the board size");
        d.H = (int)out;
        d.w1t = net->upload(wt, ok);
        d.b1 = net->upload(bias, ok);
        if (!fold_linear(b, "player_encoder.0", "player_encoder.1", wt, bias, in, out, err)) return nets_fail(AR_E_BACKEND, err);
        d.wpt = net->upload(wt, ok);
        d.bp = net->upload(bias, ok);
        if (!fold_linear(b, "trunk.0", "trunk.1", wt, bias, in, out, err)) return nets_fail(AR_E_BACKEND, err);
        d.w2t = net->upload(wt, ok);
        d.b2 = net->upload(bias, ok);
        if (!fold_linear(b, "trunk.4", "trunk.5", wt, bias, in, out, err)) return nets_fail(AR_E_BACKEND, err);
        d.w3t = net->upload(wt, ok);
        d.b3 = net->upload(bias, ok);
        const std::vector<float>*pw = b.get("policy_head.weight"), *pb = b.get("policy_head.bias"),
                          *vw = b.get("value_head.weight"), *vb = b.get("value_head.bias");
        if (!pw || !pb || !vw || !vb) return nets_fail(AR_E_BACKEND, "weight blob lacks the symmetric heads");
        std::vector<float> wh((size_t)6 * 2 * d.H), bh(6);
        memcpy(wh.data(), pw->data(), pw->size() * 4);
        memcpy(&wh[(size_t)5 * 2 * d.H], vw->data(), vw->size() * 4);
        memcpy(bh.data(), pb->data(), 20);
        bh[5] = (*vb)[0];
        d.wh = net->upload(wh, ok);
        d.bh = net->upload(bh, ok);
        d.n_head = 6;
        d.Kh = 2 * d.H;
        net->smem = (size_t)5 * TILE_SYM * (d.H + 4) * 4;
    } else if (b.arch == ARCH_CNN) {
        CnnDev& c = net->cnn;
        memset(&c, 0, sizeof c);
        c.width = d.width;
        c.height = d.height;
        c.hw = d.hw;
        const std::vector<float>* sw = b.get("stem.weight");
        if (!sw || b.dims.at("stem.weight").size() != 4) return nets_fail(AR_E_BACKEND, "weight blob lacks stem.weight");
        c.C = (int)b.dims.at("stem.weight")[0];
        if (b.dims.at("stem.weight")[1] != 5 || (c.C != 16 && c.C != 32 && c.C != 64))
            return nets_fail(AR_E_BACKEND, "CNN trunk must have 5 input planes and 16, 32 or 64 channels");
        // C = 32 / 64: trunk state in registers (k_cnn_mfma), L leaves (L * hw <= 128 MT rows) per workgroup, chosen so
        // that the image stays under 64 KB (several workgroups per CU) unless a single leaf needs more; C = 16: k_cnn
        c.MT = 0;
        c.L = CNN_TILE;
        const size_t chs = (size_t)(c.height + 2) * (c.width + 2);
        if (c.C % 32 == 0 && c.hw <= 256 && !getenv("AR_CNN_LDS")) {  // (AR_CNN_LDS: the three-image kernel, A/B knob; boards <= 8x8)
            c.MT = c.hw > 128 ? 2 : 1;
            c.L = 128 * c.MT / c.hw;
            if (c.L > CNN_TILE_MAX) c.L = CNN_TILE_MAX;
            while (c.L > 1 && (size_t)c.L * c.C * chs * 4 > 64 * 1024) c.L -= 1;
            if (const char* e = getenv("AR_CNN_L")) {  // tuning knob: leaves per workgroup (more rows per tile, fewer workgroups per CU)
                const int want = atoi(e);
                if (want >= 1 && want <= CNN_TILE_MAX && want * c.hw <= 256) {
                    c.L = want;
                    c.MT = want * c.hw > 128 ? 2 : 1;
                }
            }
        } else if (c.C % 32 == 0 && (c.width > 8 || c.height > 8)) {
            return nets_fail(AR_E_BACKEND, "AR_CNN_LDS: the three-image CNN kernel handles 32 / 64 channels on boards up to 8x8");
        }
        const size_t TL = (size_t)c.L;
        size_t g_max = 0;
        std::vector<double> sa, sb;
        if (!bn_affine(b, "stem_bn", c.C, sa, sb, err)) return nets_fail(AR_E_BACKEND, err);
        c.stem_w = net->upload(conv_t(*sw, c.C, 5, &sa), ok);
        c.stem_b = net->upload(std::vector<float>(sb.begin(), sb.end()), ok);
        size_t small_floats = 0;
        for (int bi = 0;; ++bi) {
            const std::string p = "blocks." + std::to_string(bi);
            const std::vector<float>*w1 = b.get(p + ".conv1.weight"), *w2 = b.get(p + ".conv2.weight");
            if (!w1) break;
            if (bi >= CNN_MAX_BLOCKS) return nets_fail(AR_E_BACKEND, "too many trunk blocks");
            if (!w2) return nets_fail(AR_E_BACKEND, "weight blob lacks " + p + ".conv2.weight");
            CnnBlockDev& k = c.blk[bi];
            std::vector<double> a1, b1, a2, b2;
            if (!bn_affine(b, p + ".bn1", c.C, a1, b1, err) || !bn_affine(b, p + ".bn2", c.C, a2, b2, err))
                return nets_fail(AR_E_BACKEND, err);
            k.bn1_a = net->upload(std::vector<float>(a1.begin(), a1.end()), ok);
            k.bn1_b = net->upload(std::vector<float>(b1.begin(), b1.end()), ok);
            k.w1 = net->upload(conv_t(*w1, c.C, c.C, &a2), ok);
            k.b1 = net->upload(std::vector<float>(b2.begin(), b2.end()), ok);
            k.w2 = net->upload(conv_t(*w2, c.C, c.C, nullptr), ok);
            k.gpool = 0;
            if (const std::vector<float>* pw = b.get(p + ".pool_conv.weight")) {
                const int G = (int)b.dims.at(p + ".pool_conv.weight")[0];
                k.gpool = G;
                std::vector<double> pa, pb;
                if (!bn_affine(b, p + ".pool_bn", c.C, pa, pb, err)) return nets_fail(AR_E_BACKEND, err);
                k.pbn_a = net->upload(std::vector<float>(pa.begin(), pa.end()), ok);
                k.pbn_b = net->upload(std::vector<float>(pb.begin(), pb.end()), ok);
                std::vector<float> wpt((size_t)c.C * G);
                for (int g = 0; g < G; ++g)
                    for (int ci = 0; ci < c.C; ++ci) wpt[(size_t)ci * G + g] = (*pw)[(size_t)g * c.C + ci];
                k.wp = net->upload(wpt, ok);
                const std::vector<float>*lw = b.get(p + ".pool_linear.weight"), *lb = b.get(p + ".pool_linear.bias");
                if (!lw || !lb) return nets_fail(AR_E_BACKEND, "weight blob lacks " + p + ".pool_linear");
                std::vector<float> wlt((size_t)2 * G * c.C);
                for (int o = 0; o < c.C; ++o)
                    for (int kk = 0; kk < 2 * G; ++kk) wlt[(size_t)kk * c.C + o] = (*lw)[(size_t)o * 2 * G + kk];
                k.wl = net->upload(wlt, ok);
                k.bl = net->upload(*lb, ok);
                if (TL * (c.C + 2 * G) > small_floats) small_floats = TL * (c.C + 2 * G);
                if ((size_t)G > g_max) g_max = (size_t)G;
                if (c.MT && G > 64) return nets_fail(AR_E_BACKEND, "gpool_channels above 64");
            }
            c.n_blocks = bi + 1;
        }
        if (!fold_linear(b, "player_encoder.0", "", wt, bias, in, out, err) || in != 3) return nets_fail(AR_E_BACKEND, "bad player_encoder");
        c.PD = (int)out;
        c.pe_w = net->upload(wt, ok);
        c.pe_b = net->upload(bias, ok);
        if (!fold_linear(b, "combiner.0", "", wt, bias, in, out, err) || (int)in != c.C + c.PD) return nets_fail(AR_E_BACKEND, "bad combiner");
        c.HD = (int)out;
        c.cb_w = net->upload(wt, ok);
        c.cb_b = net->upload(bias, ok);
        const std::vector<float>*pw = b.get("policy_head.linear.weight"), *pb = b.get("policy_head.linear.bias"),
                          *vw = b.get("value_head.linear.weight"), *vb = b.get("value_head.linear.bias");
        const bool pooled = b.get("value_head.mlp.0.weight") != nullptr;  // PooledValueHead (value_head.type "pooled")
        if (!pw || !pb || (!pooled && (!vw || !vb))) return nets_fail(AR_E_BACKEND, "unknown policy / value head in the weight blob");
        std::vector<float> wh((size_t)6 * 2 * c.HD, 0.0f), bh(6, 0.0f);
        memcpy(wh.data(), pw->data(), pw->size() * 4);
        memcpy(bh.data(), pb->data(), 20);
        c.VH = 0;
        if (pooled) {
            if (!fold_linear(b, "value_head.mlp.0", "", wt, bias, in, out, err) || (int)in != 2 * c.C + 2 * c.HD)
                return nets_fail(AR_E_BACKEND, "bad pooled value head");
            c.VH = (int)out;
            c.pv_w0 = net->upload(wt, ok);
            c.pv_b0 = net->upload(bias, ok);
            const std::vector<float>*w2 = b.get("value_head.mlp.2.weight"), *b2 = b.get("value_head.mlp.2.bias");
            if (!w2 || !b2 || (int)w2->size() != c.VH) return nets_fail(AR_E_BACKEND, "bad pooled value head");
            c.pv_w2 = net->upload(*w2, ok);
            c.pv_b2 = net->upload(*b2, ok);
        } else {
            memcpy(&wh[(size_t)5 * 2 * c.HD], vw->data(), vw->size() * 4);
            bh[5] = (*vb)[0];
        }
        c.hd_w = net->upload(wh, ok);
        c.hd_b = net->upload(bh, ok);
        const size_t head_floats = TL * (2 * (c.C + c.PD) + 2 * c.HD + 12 + 2 * c.C + 2 * c.VH);
        if (head_floats > small_floats) small_floats = head_floats;
        if (c.MT) {
            size_t pf = TL * c.C * chs;  // the image; between blocks [L][C][hw] + [L][G][hw]
            if (TL * (c.C + g_max) * c.hw > pf) pf = TL * (c.C + g_max) * c.hw;
            c.p_floats = (int)pf;
            net->smem = (pf + small_floats + 64) * 4;
        } else {
            net->smem = (TL * c.C * c.hw + 2 * TL * c.C * chs + small_floats + 64) * 4;
        }
        d.H = 4;  // unused by the CNN path
    } else {
        return nets_fail(AR_E_BACKEND, "unknown architecture id in the weight blob");
    }
    if (d.H % 4 != 0 || d.H > 1024) return nets_fail(AR_E_BACKEND, "hidden_dim must be a multiple of 4 and <= 1024");
    if (!ok) return nets_fail(AR_E_NOMEM, "device allocation failed while loading weights");
    if (net->smem > 150 * 1024) return nets_fail(AR_E_BACKEND, "hidden_dim too large for the LDS tile");
    return AR_OK;
}

// per-maze first-layer constants for a pool of `n_mazes` cost tables of this net's board size
// (always recomputed: a pool at the same address with the same count may hold other mazes -- an engine that
// died and a new one whose allocation landed on the same block; the kernel is tiny)
static int net_bind_mazes(ArNet* net, const uint8_t* d_maze_pool, int n_mazes, hipStream_t stream) {
    if (net->dev.arch == arnet::ARCH_CNN) {  // the CNN reads the maze planes itself
        net->bound_pool = d_maze_pool;
        net->bound_mazes = n_mazes;
        return AR_OK;
    }
    if (net->cmaze) hipFree(net->cmaze);
    net->cmaze = nullptr;
    if (hipMalloc((void**)&net->cmaze, (size_t)n_mazes * net->dev.H * 4) != hipSuccess)
        return nets_fail(AR_E_NOMEM, "device allocation failed (maze constants)");
    hipLaunchKernelGGL(arnet::k_maze_const, dim3(n_mazes), dim3(256), 0, stream, net->dev.w1t, net->dev.b1, net->dev.H,
                       net->dev.hw, d_maze_pool, n_mazes, net->cmaze);
    if (hipGetLastError() != hipSuccess) return nets_fail(AR_E_DEVICE, "k_maze_const launch failed");
    net->dev.cmaze = net->cmaze;
    net->dev.n_mazes = n_mazes;
    net->bound_pool = d_maze_pool;
    net->bound_mazes = n_mazes;
    return AR_OK;
}

// mazes `d_ids[0..n)` of the bound pool changed (a slot got a new game): refresh their first-layer constants
static int net_rebind_mazes(ArNet* net, const uint32_t* d_ids, int n, hipStream_t stream) {
    if (n <= 0 || net->dev.arch == arnet::ARCH_CNN || !net->cmaze) return AR_OK;
    hipLaunchKernelGGL(arnet::k_maze_const, dim3(n), dim3(256), 0, stream, net->dev.w1t, net->dev.b1, net->dev.H,
                       net->dev.hw, net->bound_pool, n, net->cmaze, d_ids);
    if (hipGetLastError() != hipSuccess) return nets_fail(AR_E_DEVICE, "k_maze_const launch failed");
    return AR_OK;
}

template <int NW>
static int net_launch(ArNet* net, const ar::LeafReq<NW>* q, const uint32_t* qcount, uint32_t n_max, const char* boards,
                      size_t board_stride, ar::EvalOut* out, float* logits, hipStream_t stream) {
    using namespace arnet;
    if (n_max == 0) return AR_OK;
    const bool mlp_mfma = net->dev.arch == ARCH_MLP && mlp_all_mfma(net->dev.H);
    static const int mlp_mt = getenv("AR_MLP_MT") && atoi(getenv("AR_MLP_MT")) == 1 ? 1 : MLP_MFMA_MT;  // tuning knob
    // (k_symmetric_mfma2 stages the shared encoder's operand, hw + 1 values, at the head of a row of its buffer)
    const bool sym_mfma = net->dev.arch == ARCH_SYMMETRIC && symmetric_mfma_ok(net->dev.H) && !getenv("AR_SYM_FMA") &&
                          ((net->dev.hw + 2) & ~1) <= net->dev.H + 4;
    const int tile = mlp_mfma ? 32 * mlp_mt : net->dev.arch == ARCH_MLP ? TILE_MLP : net->dev.arch == ARCH_CNN ? net->cnn.L
                                                                        : sym_mfma ? 32 : TILE_SYM;
    const uint32_t blocks = (n_max + tile - 1) / tile;
    if (net->dev.arch == ARCH_CNN && net->cnn.MT) {
        const void* fn = net->cnn.MT == 2 ? (const void*)k_cnn_mfma<NW, 2> : (const void*)k_cnn_mfma<NW, 1>;
        if (net->smem > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)net->smem) != hipSuccess)
            return nets_fail(AR_E_DEVICE, "cannot reserve LDS for the CNN kernel");
        if (net->cnn.MT == 2)
            hipLaunchKernelGGL((k_cnn_mfma<NW, 2>), dim3(blocks), dim3(NTHREADS), net->smem, stream, net->cnn, q, qcount, n_max,
                               boards, board_stride, net->bound_pool, out, logits);
        else
            hipLaunchKernelGGL((k_cnn_mfma<NW, 1>), dim3(blocks), dim3(NTHREADS), net->smem, stream, net->cnn, q, qcount, n_max,
                               boards, board_stride, net->bound_pool, out, logits);
    } else if (net->dev.arch == ARCH_CNN) {
        if (net->smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_cnn<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)net->smem) != hipSuccess)
            return nets_fail(AR_E_DEVICE, "cannot reserve LDS for the CNN kernel");
        hipLaunchKernelGGL(k_cnn<NW>, dim3(blocks), dim3(NTHREADS), net->smem, stream, net->cnn, q, qcount, n_max, boards,
                           board_stride, net->bound_pool, out, logits);
    } else if (mlp_mfma) {
        const size_t smem = (size_t)32 * mlp_mt * (net->dev.H + 4) * 4;
        // first-layer variant (k_mlp_mfma's FL): 2 unless AR_MLP_FL says otherwise (read per launch: a test toggles it);
        // 0 needs the whole observation's non-maze part in a row of `act`
        int fl = getenv("AR_MLP_FL") ? atoi(getenv("AR_MLP_FL")) : 2;
        const int K1e = (3 * net->dev.hw + 6 + 1) & ~1;
        const int K2e = (net->dev.hw + 6 + 1) & ~1;
        if (fl < 0 || fl > 2) fl = 2;
        if (fl == 2 && K2e > net->dev.H + 4) fl = 1;  // (the staged operand is a row of `act`: narrow hidden layers on big boards)
        if (fl == 0 && K1e > net->dev.H + 4) fl = 1;
        const void* fns[2][3] = {{(const void*)k_mlp_mfma<NW, 1, 0>, (const void*)k_mlp_mfma<NW, 1, 1>, (const void*)k_mlp_mfma<NW, 1, 2>},
                                 {(const void*)k_mlp_mfma<NW, 2, 0>, (const void*)k_mlp_mfma<NW, 2, 1>, (const void*)k_mlp_mfma<NW, 2, 2>}};
        const void* fn = fns[mlp_mt == 1 ? 0 : 1][fl];
        if (smem > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return nets_fail(AR_E_DEVICE, "cannot reserve LDS for the MLP kernel");
#define AR_MLP_LAUNCH(MT_, FL_)                                                                                          \
    hipLaunchKernelGGL((k_mlp_mfma<NW, MT_, FL_>), dim3(blocks), dim3(NTHREADS), smem, stream, net->dev, q, qcount, n_max, \
                       boards, board_stride, out, logits)
        if (mlp_mt == 1) {
            if (fl == 0) AR_MLP_LAUNCH(1, 0);
            else if (fl == 1) AR_MLP_LAUNCH(1, 1);
            else AR_MLP_LAUNCH(1, 2);
        } else {
            if (fl == 0) AR_MLP_LAUNCH(2, 0);
            else if (fl == 1) AR_MLP_LAUNCH(2, 1);
            else AR_MLP_LAUNCH(2, 2);
        }
#undef AR_MLP_LAUNCH
    } else if (net->dev.arch == ARCH_MLP) {
        if (net->smem > 48 * 1024 &&
            hipFuncSetAttribute((const void*)k_mlp<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)net->smem) !=
                hipSuccess)
            return nets_fail(AR_E_DEVICE, "cannot reserve LDS for the MLP kernel");
        hipLaunchKernelGGL(k_mlp<NW>, dim3(blocks), dim3(NTHREADS), net->smem, stream, net->dev, q, qcount, n_max, boards,
                           board_stride, out, logits);
    } else if (sym_mfma) {
        const size_t smem = (size_t)2 * 32 * (net->dev.H + 4) * 4;
        if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_symmetric_mfma2<NW>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return nets_fail(AR_E_DEVICE, "cannot reserve LDS for the SymmetricMLP kernel");
        hipLaunchKernelGGL(k_symmetric_mfma2<NW>, dim3(blocks), dim3(NTHREADS), smem, stream, net->dev, q, qcount, n_max, boards,
                           board_stride, out, logits);
    } else {
        if (net->smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_symmetric<NW>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)net->smem) != hipSuccess)
            return nets_fail(AR_E_DEVICE, "cannot reserve LDS for the SymmetricMLP kernel");
        hipLaunchKernelGGL(k_symmetric<NW>, dim3(blocks), dim3(NTHREADS), net->smem, stream, net->dev, q, qcount, n_max,
                           boards, board_stride, out, logits);
    }
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return nets_fail(AR_E_DEVICE, std::string("network kernel launch failed: ") + hipGetErrorString(le));
    return AR_OK;
}

// evaluator step of the self-play loop: leaves in `queue[0 .. *queue_count)` -> ev_out
template <int NW>
static int net_forward_queue(ArNet* net, const ar::LeafReq<NW>* queue, const uint32_t* queue_count, uint32_t n_max,
                             const ar::Slot<NW>* slots, const uint8_t* maze_pool, ar::EvalOut* ev_out,
                             hipStream_t stream) {
    (void)maze_pool;
    return net_launch<NW>(net, queue, queue_count, n_max, (const char*)&slots[0].board, sizeof(ar::Slot<NW>), ev_out,
                          nullptr, stream);
}
