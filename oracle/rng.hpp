// TEST INFRASTRUCTURE ONLY -- CPU oracle. Not linked into, imported by or called from the
// product path (alpharat_amd/). Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// Restatement of the random-number algorithms the reference's search consumes. The crates
// themselves are third-party and absent from /root/reference (Cargo.lock:1124-1132 rand 0.8.5,
// :1171 rand_distr 0.4.3), so these are restated from the published algorithms and anchored on
// the reference's call sites:
//   - SmallRng::seed_from_u64           crates/alpharat-mcts/src/bindings.rs:262-265
//   - rng.gen_range(0..tie_count) (u32) crates/alpharat-mcts/src/search.rs:527
//   - WeightedIndex::<f32>              crates/alpharat-sampling/src/selfplay.rs:474-479
//   - rng.sample(Gamma::new(a, 1.0))    crates/alpharat-mcts/src/search.rs:410-418
// PARITY UNPINNED: no reference test records any of these streams.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

#include "zig_norm_tables.inc"
static const double ZIG_NORM_X[257] = AR_ZIG_NORM_X_INIT;
static const double ZIG_NORM_F[257] = AR_ZIG_NORM_F_INIT;

// rand 0.8.5 SmallRng on 64-bit targets = Xoshiro256PlusPlus.
struct SmallRng {
    uint64_t s[4];

    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

    // SmallRng (0.8.5) does not forward seed_from_u64 to the xoshiro type, so the rand_core 0.6
    // default applies: a PCG32 stream fills the 32 seed bytes, 4 at a time, little endian.
    static SmallRng seed_from_u64(uint64_t state) {
        const uint64_t MUL = 6364136223846793005ULL;
        const uint64_t INC = 11634580027462260723ULL;
        uint8_t seed[32];
        for (int c = 0; c < 8; ++c) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * c + 0] = (uint8_t)(x);
            seed[4 * c + 1] = (uint8_t)(x >> 8);
            seed[4 * c + 2] = (uint8_t)(x >> 16);
            seed[4 * c + 3] = (uint8_t)(x >> 24);
        }
        return from_seed(seed);
    }

    // Xoshiro256PlusPlus::from_seed: an all-zero seed is replaced by seed_from_u64(0) of the
    // xoshiro type itself (SplitMix64); otherwise four little-endian u64 words.
    static SmallRng from_seed(const uint8_t seed[32]) {
        bool all_zero = true;
        for (int i = 0; i < 32; ++i) all_zero = all_zero && seed[i] == 0;
        SmallRng r;
        if (all_zero) {
            uint64_t st = 0;
            for (int i = 0; i < 4; ++i) {
                st += 0x9e3779b97f4a7c15ULL;
                uint64_t z = st;
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
                z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
                r.s[i] = z ^ (z >> 31);
            }
            return r;
        }
        for (int i = 0; i < 4; ++i) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; --b) w = (w << 8) | seed[8 * i + b];
            r.s[i] = w;
        }
        return r;
    }

    inline uint64_t next_u64() {
        uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    // upper half: the low bits of xoshiro have linear dependencies
    inline uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }

    // rand 0.8.5 UniformInt<u32>::sample_single(0, n): widening multiply with the
    // "conservative zone" rejection. n > 0.
    inline uint32_t gen_range_u32(uint32_t n) {
        uint32_t zone = (n << __builtin_clz(n)) - 1;
        for (;;) {
            uint32_t v = next_u32();
            uint64_t m = (uint64_t)v * (uint64_t)n;
            uint32_t lo = (uint32_t)m;
            if (lo <= zone) return (uint32_t)(m >> 32);
        }
    }

    // Standard f64 in [0,1): 53 random bits.
    inline double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

    // Open01 f64 in (0,1): 52 bits into a [1,2) float, minus (1 - eps/2).
    inline double gen_open01_f64() {
        uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ULL;
        double v;
        std::memcpy(&v, &bits, 8);
        return v - (1.0 - 2.220446049250313e-16 / 2.0);
    }
};

// rand 0.8.5 WeightedIndex<f32>::new + sample over exactly five weights.
// Returns -1 for the error cases (negative / NaN weight, all zero) -- the caller falls back to
// STAY (selfplay.rs:474-479).
inline int weighted_index5_sample(const float w[5], SmallRng& rng) {
    float cumulative[4];
    if (!(w[0] >= 0.0f)) return -1;
    float total = w[0];
    for (int i = 1; i < 5; ++i) {
        if (!(w[i] >= 0.0f)) return -1;
        cumulative[i - 1] = total;
        total += w[i];
    }
    if (total == 0.0f) return -1;
    // UniformFloat<f32>::new(0, total): scale shrinks by one ulp while scale*max_rand+low >= high
    float low = 0.0f, high = total;
    const float max_rand = 0.99999988079071044921875f;  // (0x7FFFFF mantissa in [1,2)) - 1
    float scale = high - low;
    for (;;) {
        if (!(scale * max_rand + low >= high)) break;
        uint32_t b;
        std::memcpy(&b, &scale, 4);
        b -= 1;
        std::memcpy(&scale, &b, 4);
    }
    uint32_t bits = (rng.next_u32() >> 9) | 0x3F800000u;
    float v12;
    std::memcpy(&v12, &bits, 4);
    float chosen = (v12 - 1.0f) * scale + low;
    // partition_point(|w| w <= chosen)
    int idx = 0;
    while (idx < 4 && cumulative[idx] <= chosen) ++idx;
    return idx;
}

// rand_distr 0.4.3 StandardNormal (f64) by the ziggurat method.
inline double sample_standard_normal(SmallRng& rng) {
    for (;;) {
        uint64_t bits = rng.next_u64();
        unsigned i = (unsigned)(bits & 0xff);
        uint64_t fb = (bits >> 12) | 0x4000000000000000ULL;  // exponent 1 -> [2,4)
        double u;
        std::memcpy(&u, &fb, 8);
        u -= 3.0;
        double x = u * ZIG_NORM_X[i];
        double test_x = std::fabs(x);
        if (test_x < ZIG_NORM_X[i + 1]) return x;
        if (i == 0) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                double x_ = rng.gen_open01_f64();
                double y_ = rng.gen_open01_f64();
                xx = std::log(x_) / AR_ZIG_NORM_R;
                yy = std::log(y_);
            }
            return u < 0.0 ? xx - AR_ZIG_NORM_R : AR_ZIG_NORM_R - xx;
        }
        if (ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * rng.gen_f64() < std::exp(-x * x / 2.0))
            return x;
    }
}

// rand_distr 0.4.3 Gamma(shape >= 1, scale = 1): Marsaglia-Tsang. The reference only ever asks
// for shape = concentration / n_outcomes with n_outcomes in 2..5 (search.rs:407).
struct GammaLarge {
    double scale, c, d;
    bool ok;
    static GammaLarge make(double shape, double scale) {
        GammaLarge g;
        g.ok = shape >= 1.0 && scale > 0.0;
        g.scale = scale;
        g.d = shape - 1.0 / 3.0;
        g.c = 1.0 / std::sqrt(9.0 * g.d);
        return g;
    }
    double sample(SmallRng& rng) const {
        for (;;) {
            double x = sample_standard_normal(rng);
            double v_cbrt = 1.0 + c * x;
            if (v_cbrt <= 0.0) continue;
            double v = v_cbrt * v_cbrt * v_cbrt;
            double u = rng.gen_open01_f64();
            double x_sqr = x * x;
            if (u < 1.0 - 0.0331 * x_sqr * x_sqr || std::log(u) < 0.5 * x_sqr + d * (1.0 - v + std::log(v)))
                return d * v * scale;
        }
    }
};

}  // namespace oracle
