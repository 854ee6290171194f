// Per-game random stream on the device: the generator and the three distributions the search
// draws from, matching what the reference's Rust uses (rand 0.8.5 SmallRng = xoshiro256++,
// UniformInt<u32> single sampling, WeightedIndex<f32>, rand_distr 0.4.3 Gamma/StandardNormal).
// Call sites mirrored: search.rs:527 (tie break), search.rs:410-418 (Dirichlet), selfplay.rs:474-479.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AR_HD __host__ __device__ inline
#else
#define AR_HD inline
#endif
#include <math.h>

namespace ar {

struct Rng {
    uint64_t a, b, c, d;
};

AR_HD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

AR_HD uint64_t rng_u64(Rng& r) {
    const uint64_t out = rotl64(r.a + r.d, 23) + r.a;
    const uint64_t t = r.b << 17;
    r.c ^= r.a;
    r.d ^= r.b;
    r.b ^= r.c;
    r.a ^= r.d;
    r.c ^= t;
    r.d = rotl64(r.d, 45);
    return out;
}
AR_HD uint32_t rng_u32(Rng& r) { return (uint32_t)(rng_u64(r) >> 32); }

// seed_from_u64 as rand_core 0.6 defines it for generators that do not override it: eight PCG32
// outputs become the 32 seed bytes (little endian), i.e. word k = out[2k] | out[2k+1] << 32.
AR_HD void rng_seed(Rng& r, uint64_t seed) {
    uint64_t st = seed;
    uint64_t w[4];
    for (int k = 0; k < 4; ++k) {
        uint32_t half[2];
        for (int j = 0; j < 2; ++j) {
            st = st * 6364136223846793005ULL + 11634580027462260723ULL;
            const uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27);
            const uint32_t rot = (uint32_t)(st >> 59);
            half[j] = (xs >> rot) | (xs << ((32u - rot) & 31u));
        }
        w[k] = (uint64_t)half[0] | ((uint64_t)half[1] << 32);
    }
    if ((w[0] | w[1] | w[2] | w[3]) == 0) {  // xoshiro refuses the all-zero state: SplitMix64(0)
        uint64_t z = 0;
        for (int k = 0; k < 4; ++k) {
            z += 0x9e3779b97f4a7c15ULL;
            uint64_t v = z;
            v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ULL;
            v = (v ^ (v >> 27)) * 0x94d049bb133111ebULL;
            w[k] = v ^ (v >> 31);
        }
    }
    r.a = w[0];
    r.b = w[1];
    r.c = w[2];
    r.d = w[3];
}

AR_HD int clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)v);
#else
    return __builtin_clz(v);
#endif
}

// gen_range(0..n), n >= 1
AR_HD uint32_t rng_below(Rng& r, uint32_t n) {
    const uint32_t zone = (n << clz32(n)) - 1u;
    for (;;) {
        const uint64_t wide = (uint64_t)rng_u32(r) * (uint64_t)n;
        if ((uint32_t)wide <= zone) return (uint32_t)(wide >> 32);
    }
}

AR_HD float bits_to_f32(uint32_t b) {
    union {
        uint32_t u;
        float f;
    } x;
    x.u = b;
    return x.f;
}
AR_HD uint32_t f32_to_bits(float f) {
    union {
        uint32_t u;
        float f;
    } x;
    x.f = f;
    return x.u;
}
AR_HD double bits_to_f64(uint64_t b) {
    union {
        uint64_t u;
        double f;
    } x;
    x.u = b;
    return x.f;
}

// WeightedIndex<f32> over five weights; -1 when the constructor would fail.
AR_HD int rng_weighted5(Rng& r, const float* w) {
    float prefix[4];
    float total = 0.0f;
    for (int i = 0; i < 5; ++i) {
        if (!(w[i] >= 0.0f)) return -1;
        if (i == 0) {
            total = w[0];
        } else {
            prefix[i - 1] = total;
            total += w[i];
        }
    }
    if (total == 0.0f) return -1;
    float scale = total;  // Uniform::new(0, total)
    while (scale * 0.99999988079071044921875f >= total) scale = bits_to_f32(f32_to_bits(scale) - 1u);
    const float u01 = bits_to_f32((rng_u32(r) >> 9) | 0x3F800000u) - 1.0f;
    const float pick = u01 * scale + 0.0f;
    int k = 0;
    while (k < 4 && prefix[k] <= pick) ++k;
    return k;
}

AR_HD double rng_open01(Rng& r) {
    return bits_to_f64((rng_u64(r) >> 12) | 0x3FF0000000000000ULL) - (1.0 - 1.1102230246251565e-16);
}
AR_HD double rng_unit_f64(Rng& r) { return (double)(rng_u64(r) >> 11) * 1.1102230246251565e-16; }

struct ZigTables {
    double x[257];
    double f[257];
};
#define AR_ZIG_R 3.654152885361008796

AR_HD double rng_normal(Rng& r, const ZigTables* zt) {
    for (;;) {
        const uint64_t bits = rng_u64(r);
        const int layer = (int)(bits & 0xffu);
        const double u = bits_to_f64((bits >> 12) | 0x4000000000000000ULL) - 3.0;
        const double x = u * zt->x[layer];
        if (fabs(x) < zt->x[layer + 1]) return x;
        if (layer == 0) {
            double tx = 1.0, ty = 0.0;
            while (-2.0 * ty < tx * tx) {
                const double e1 = rng_open01(r);
                const double e2 = rng_open01(r);
                tx = log(e1) / AR_ZIG_R;
                ty = log(e2);
            }
            return u < 0.0 ? tx - AR_ZIG_R : AR_ZIG_R - tx;
        }
        const double f1 = zt->f[layer + 1], f0 = zt->f[layer];
        if (f1 + (f0 - f1) * rng_unit_f64(r) < exp(-x * x / 2.0)) return x;
    }
}

// Gamma(shape >= 1, 1): Marsaglia-Tsang squeeze
AR_HD double rng_gamma(Rng& r, double shape, const ZigTables* zt) {
    const double d = shape - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        const double x = rng_normal(r, zt);
        const double t = 1.0 + c * x;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        const double u = rng_open01(r);
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2 || log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * 1.0;
    }
}

}  // namespace ar
