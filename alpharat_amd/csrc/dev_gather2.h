// The gather (search.rs:576-817, 961-999) with TWO LANES PER GAME: lane 0 of a pair is player 1, lane 1 is player 2;
// a wavefront holds 32 games. Same node records, same arithmetic in the same order, same random draws as the other two
// gathers (gather_round in dev_search.h: one lane per game; gather8_round in dev_gather8.h: eight lanes per game) --
// identical trees, batch entries and counters; which one runs is a scheduling choice (AR_GATHER).
//
// Why a third shape. Counters of the first two on the bench workload (profiles/r02_pmc_sq_gather_ab.txt): the
// lane-per-game kernel issues 0.30 G vector instructions per launch and spends 5/6 of its time waiting for memory
// (one wavefront per SIMD is all the resident games give it); the eight-lane kernel keeps three wavefronts per SIMD
// busy but issues 1.03 G -- everything that is not per outcome is replicated eight times, the per-outcome work runs
// on five lanes of eight, twice (two players) -- and is bound by exactly that: its vector ALUs are busy 63 % of a
// launch. What one game's round costs is fixed; what a WAVEFRONT's round costs is the union of the branches its games
// are in, nearly the whole round for eight games as for sixty-four. So: the fewest lanes per game that still share a
// record's load and hide its latency. With a lane per PLAYER
//   * the allocation state of a node is the lane kernel's HalfAlloc, one per lane (half_init / half_take are shared
//     with it); best / second-best scans are local, the two players meet through one DPP exchange per allocation step;
//   * a record arrives in one round trip as with eight lanes: five edge groups per lane, the headers from the same
//     address, the child table by quarters (lane 0: four, lane 1: three) into LDS, where the visits allocated to the
//     25 child slots (vtp) live as well -- dynamically indexed, no select chains, no registers;
//   * position, masks, counters are replicated twice instead of eight times; the random stream and the root position
//     live in LDS.
// A wavefront's round then serves 32 games for about the instructions the eight-lane kernel spends on 8.
#pragma once
#include "dev_gather8.h"

#if defined(__HIPCC__)
namespace ar {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the other lane of the pair (DPP quad_perm [1,0,3,2]); both lanes of a pair are always active together
__device__ inline uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }

// a pair's LDS: written by one lane, read by the other inside the same wavefront (in-order LDS, no barrier needed);
// every access is volatile so that the compiler never keeps a copy in a register across the other lane's write
template <int NW>
struct alignas(16) PairShared {
    uint32_t kid[28];  // child table of the current node (25 used)
    uint32_t vtp[28];  // visits allocated to its child slots in this gather (search.rs:757-798)
    uint32_t rng[8];   // the game's random stream: only tie breaks draw from it (search.rs:511-532)
    State<NW> root_st;
    uint32_t pad[(76 * 4 - (28 + 28 + 8) * 4 - (int)sizeof(State<NW>)) / 4 > 0 ? (76 * 4 - (28 + 28 + 8) * 4 - (int)sizeof(State<NW>)) / 4 : 4];
};

__device__ inline Rng pair_rng_load(const volatile uint32_t* w) {
    Rng r;
    r.a = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    r.b = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    r.c = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
    r.d = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    return r;
}
__device__ inline void pair_rng_store(volatile uint32_t* w, const Rng& r) {
    w[0] = (uint32_t)r.a;
    w[1] = (uint32_t)(r.a >> 32);
    w[2] = (uint32_t)r.b;
    w[3] = (uint32_t)(r.b >> 32);
    w[4] = (uint32_t)r.c;
    w[5] = (uint32_t)(r.c >> 32);
    w[6] = (uint32_t)r.d;
    w[7] = (uint32_t)(r.d >> 32);
}

template <int NW>
struct Pair {
    // ---- replicated in both lanes ----
    bool done;
    uint32_t batch;
    long long left;
    uint32_t depth, node, mask, omap0, omap1, pick_mv;
    bool have_pick;
    State<NW> work;  // position at the current node
    uint32_t alloc_left;
    uint32_t hi, cap, root, node_count, n_proc, n_coll, b_nn, b_term, b_coll, error, batch_active;
    uint32_t d_new, d_visits, rounds;
    // ---- this lane's player ----
    HalfAlloc h;
};

// A gather cut off at the launch's round limit (scheduling only: the longest walks of a launch would otherwise hold
// every other game's evaluation and backup back), kept in the game's scratch until the next launch continues it.
template <int NW>
struct alignas(16) PairParked {
    uint32_t kid[28], vtp[28];
    HalfAlloc h[2];
    uint32_t batch, depth, node, mask, omap0, omap1, pick_mv, have_pick, alloc_left;
    uint32_t n_proc, n_coll, b_nn, b_term, b_coll, error, pad;
    long long left;
    State<NW> work;
};

// (field by field: a struct copy through a run-time index goes through the stack)
__device__ inline void half_copy(HalfAlloc& d, const HalfAlloc& s) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        d.score[k] = s.score[k];
        d.util[k] = s.util[k];
        d.num[k] = s.num[k];
        d.ns[k] = s.ns[k];
        d.add[k] = s.add[k];
        d.nif0[k] = s.nif0[k];
    }
    d.forced = s.forced;
    d.n = s.n;
}

// search.rs:463-554 estimated_visits_to_change_best_half on this lane's player (half_best of dev_search.h, with the
// random stream in LDS and the two players' tie passes in player order: the reference calls P1 then P2, search.rs:777-778)
template <int NW>
__device__ inline void pair_best(const HalfAlloc& h, PairShared<NW>& sh, uint32_t p, uint32_t& best_out, uint32_t& vtc_out) {
    const float NEG_INF = -__builtin_inff();
    const uint32_t n = h.n;
    uint32_t best = 0;
    float best_score = NEG_INF, second = NEG_INF;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        const float sc = h.score[i];
        const bool in = i < n;
        const bool gt = in && sc > best_score;
        const bool gt2 = in && !gt && sc > second;
        second = gt ? best_score : (gt2 ? sc : second);
        best_score = gt ? sc : best_score;
        best = gt ? i : best;
    }
    bool any_tie = false;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) any_tie = any_tie || (i < n && i != best && fabsf(h.score[i] - best_score) < 1e-12f);
    if (any_tie) {
        // player 1's pass, then player 2's: the second reads the stream the first has written back. Two blocks with a
        // scheduling barrier between them -- a loop over the player with `if (p == pl)` inside would be one execution
        // per thread, which the compiler may legally run for both lanes at once.
        auto tie_pass = [&]() {
            Rng rng = pair_rng_load(sh.rng);
            uint32_t ties = 1;
#pragma unroll
            for (uint32_t i = 0; i < 5; ++i) {
                if (i < n && i != best && fabsf(h.score[i] - best_score) < 1e-12f) {
                    ties += 1;
                    if (rng_below(rng, ties) == 0) best = i;
                }
            }
            pair_rng_store(sh.rng, rng);
        };
        if (p == 0) tie_pass();
        __builtin_amdgcn_wave_barrier();
        if (p == 1) tie_pass();
    }
    best_out = best;
    vtc_out = 0xFFFFFFFFu;
    const float best_util = pick5(h.util, best);
    if (second <= NEG_INF) return;
    if (best_util >= second) return;
    const float denom = second - best_util;
    if (denom <= 0.0f) return;
    const float n1 = (float)pick5u(h.ns, best) + 1.0f;
    float vtc = pick5(h.num, best) / denom - n1 + 1.0f;
    if (!(vtc > 1.0f)) vtc = 1.0f;
    const uint32_t k = vtc >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)vtc;
    vtc_out = k > 1 ? k : 1;
}

// One round of a pair's gather: the same decisions in the same order as gather_round / gather8_round.
template <int NW>
__device__ inline void gather2_round(Pair<NW>& o, PairShared<NW>& sh, const OutcomeTable& otab, const Board& board,
                                     const OctMem<NW>& m, const SearchCfg& cfg, uint32_t p) {
    if (o.done) return;
    o.rounds += 1;
    volatile uint32_t* kid = sh.kid;
    volatile uint32_t* vtp = sh.vtp;
    volatile u32x4* kid4 = (volatile u32x4*)sh.kid;
    volatile u32x4* vtp4 = (volatile u32x4*)sh.vtp;
    const uint32_t q0 = p ? 4u : 0u, qn = p ? 3u : 4u;  // this lane's quarters of a 25-entry table
    if (o.alloc_left == 0) {
        if (o.mask == 0 && o.depth > 0) {
            // level exhausted: backtrack (search.rs:728-734)
            o.depth -= 1;
            const LevelO<NW>& L = m.levels()[o.depth];
            o.node = L.node;
            o.mask = L.mask;
            o.omap0 = L.omap0;
            o.omap1 = L.omap1;
            o.work = L.saved;
            const u32x4* lv = (const u32x4*)L.vtp;
            const u32x4* lk = (const u32x4*)&m.kids[L.node];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                if (j < qn) {
                    vtp4[q0 + j] = lv[q0 + j];
                    kid4[q0 + j] = lk[q0 + j];
                }
            }
        } else {
            uint32_t rec = NIL, visits_in = 0;
            bool from_pick = false;
            uint32_t emit_node = NIL, emit_kind = PROC_NONE, coll_mv = 0;
            State<NW> pos;  // the position at the node that is looked at (the current node's stays in o.work)
            if (o.mask == 0) {
                // search.rs:981-999 outer gather loop around pick_nodes_to_extend
                if (o.have_pick) {
                    o.b_coll += o.pick_mv;
                    o.left -= (long long)o.pick_mv;
                    o.have_pick = false;
                }
                if (!(o.n_proc < o.batch && o.left > 0)) {
                    o.done = true;
                    o.batch_active = 1;
                    return;
                }
                uint32_t budget = (uint32_t)o.left;
                if (o.batch - o.n_proc < budget) budget = o.batch - o.n_proc;
                o.pick_mv = 0;
                o.have_pick = true;
                pos = sh.root_st;
                rec = o.root;
                visits_in = budget;
                from_pick = true;
            } else {
                const uint32_t idx = (uint32_t)lowest_bit(o.mask);
                o.mask &= o.mask - 1;
                const uint32_t k = vtp[idx];
                const uint32_t child = kid[idx];
                const uint32_t o1 = idx / 5, o2 = idx % 5;
                float r1, r2;
                pos = o.work;
                st_step(board, m.cost(), pos, outcome_action(o.omap0, o1), outcome_action(o.omap1, o2), r1, r2);
                if (child == NIL) {
                    // new leaf: shell creation + claim are stores only (tree.rs:107-148, search.rs:675-701)
                    if (o.hi >= o.cap) {  // excluded by the capacity check at the start of the gather
                        o.error = 3;
                    } else {
                        const uint32_t nid = o.hi++;
                        const bool over = st_over(board, pos);
                        // the record's 16-byte groups: 0..4 player 1's edges, 5..9 player 2's (all zero), 10..12 the
                        // headers; the child table's seven groups are all NIL. Lane p: its player's edges, its
                        // quarters of the table; headers h0, h1 by lane 0, h2 by lane 1.
                        const u32x4 zero = {0u, 0u, 0u, 0u}, nil4 = {NIL, NIL, NIL, NIL};
                        const uint32_t k1 = outcome_key(m.cost(), pos.p1, pos.m1), k2 = outcome_key(m.cost(), pos.p2, pos.m2);
                        u32x4 ga, gb;  // lane 0: h0, h1; lane 1: h2, (unused)
                        if (p == 0) {
                            ga = (u32x4){0u, 0u, 0u, 1u};  // v1 0, v2 0, visits 0, nif 1 (try_start_score_update on a fresh node)
                            gb = (u32x4){__float_as_uint((float)(pos.remaining > 1 ? pos.remaining : 1)), __float_as_uint(r1),
                                         __float_as_uint(r2), o.node};
                        } else {
                            ga = (u32x4){otab.omap[k1], otab.omap[k2], otab.n[k1] | (otab.n[k2] << 8) | (o1 << 16) | (o2 << 24),
                                         over ? 1u : 0u};
                            gb = zero;
                        }
                        u32x4* S = (u32x4*)&m.stats[nid];
                        u32x4* K = (u32x4*)&m.kids[nid];
#pragma unroll
                        for (uint32_t j = 0; j < 5; ++j) S[5 * p + j] = zero;
                        S[p ? 12 : 10] = ga;
                        if (p == 0) S[11] = gb;
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j)
                            if (j < qn) K[q0 + j] = nil4;
                        if (p == 0) m.kids[o.node].c[idx] = nid;
                        o.node_count += 1;
                        o.d_new += 1;
                        emit_node = nid;
                        emit_kind = over ? PROC_TERMINAL : PROC_EVAL;
                        coll_mv = k > 1 ? k - 1 : 0;
                    }
                } else {
                    rec = child;
                    visits_in = k;
                }
            }
            if (rec != NIL) {
                // the record of `rec`: this player's five edge groups, the three headers (both lanes, one address),
                // this lane's quarters of the child table -- one round trip
                const NodeStats& N = m.stats[rec];
                Edge E[5];
#pragma unroll
                for (uint32_t j = 0; j < 5; ++j) E[j] = N.e[p][j];
                const NodeH0 a = N.h0;
                const NodeH1 b = N.h1;
                const NodeH2 c = N.h2;
                u32x4 kin[4];
                {
                    const u32x4* lk = (const u32x4*)&m.kids[rec];
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) kin[j] = j < qn ? lk[q0 + j] : (u32x4){NIL, NIL, NIL, NIL};
                }
                if (a.visits == 0 || c.terminal != 0) {
                    // leaf or terminal (search.rs:591-636 for the root, :675-706 for a child)
                    emit_node = rec;
                    if (!(a.visits == 0 && a.nif > 0)) {  // try_start_score_update
                        const bool term = c.terminal != 0 || st_over(board, pos);
                        if (p == 0) {
                            m.stats[rec].h0.nif = a.nif + 1;
                            if (term && a.visits == 0) m.stats[rec].h2.terminal = 1;
                        }
                        emit_kind = term ? PROC_TERMINAL : PROC_EVAL;
                        coll_mv = visits_in > 1 ? visits_in - 1 : 0;
                    } else {
                        coll_mv = visits_in;
                    }
                } else if (!from_pick && o.depth >= m.max_depth) {
                    o.error = 4;
                } else {
                    // visited interior node: route the visits through it (search.rs:639 / :707-725)
                    if (p == 0) m.stats[rec].h0.nif = a.nif + visits_in;
                    if (!from_pick && o.mask != 0) {  // siblings still wait: keep the parent level for the way back
                        LevelO<NW>& L = m.levels()[o.depth];
                        if (p == 0) {
                            L.node = o.node;
                            L.mask = o.mask;
                            L.omap0 = o.omap0;
                            L.omap1 = o.omap1;
                            L.saved = o.work;
                        }
                        u32x4* lv = (u32x4*)L.vtp;
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j)
                            if (j < qn) lv[q0 + j] = vtp4[q0 + j];
                        o.depth += 1;
                    }
                    if (from_pick) o.depth = 0;
                    o.work = pos;
                    // build_gather_level set-up (search.rs:742-774) for this lane's player
                    const uint32_t cv = a.visits > 0 ? a.visits - 1 : 0;
                    half_init(o.h, E, meta_n(c.meta, (int)p), p ? a.v2 : a.v1, b.scale, cv, cfg, from_pick);
                    o.node = rec;
                    o.omap0 = c.omap[0];
                    o.omap1 = c.omap[1];
                    o.mask = 0;
                    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        if (j < qn) {
                            kid4[q0 + j] = kin[j];
                            vtp4[q0 + j] = zero;
                        }
                    }
                    o.d_visits += 1;
                    o.alloc_left = visits_in;
                }
            }
            if (emit_kind != PROC_NONE) {
                if (o.n_proc >= cfg.batch_size) {
                    o.error = 1;
                } else {
                    const uint32_t i = o.n_proc++;
                    if (p == 0) {
                        ProcEntry pe;
                        pe.node = emit_node;
                        pe.kind = emit_kind;
                        m.proc()[i] = pe;
                    }
                    if (emit_kind == PROC_EVAL) {
                        const uint32_t j = o.b_nn++;
                        if (p == 1) m.leaves()[j] = pos;
                    } else {
                        o.b_term += 1;
                    }
                }
            }
            if (coll_mv) {
                o.pick_mv += coll_mv;
                if (o.n_coll >= m.coll_cap) {
                    o.error = 2;
                } else {
                    if (p == 0) {
                        CollEntry ce;
                        ce.node = emit_node;
                        ce.mv = coll_mv;
                        m.coll()[o.n_coll] = ce;
                    }
                    o.n_coll += 1;
                }
            }
        }
    }
    if (o.alloc_left > 0) {
        for (uint32_t it = 0; it < cfg.alloc_per_round && o.alloc_left > 0; ++it) {  // search.rs:775-798, no memory traffic
            uint32_t b, c;
            pair_best(o.h, sh, p, b, c);
            const uint32_t b_other = pair_swap(b), c_other = pair_swap(c);
            uint32_t k = o.alloc_left;
            if (c < k) k = c;
            if (c_other < k) k = c_other;
            if (k < 1) k = 1;
            const uint32_t flat = p ? b_other * 5 + b : b * 5 + b_other;
            if (p == 0) vtp[flat] = vtp[flat] + k;
            o.mask |= 1u << flat;
            half_take(o.h, b, k);
            o.alloc_left -= k;
        }
        if (o.alloc_left == 0) {
            NodeStats& W = m.stats[o.node];  // search.rs:800-814: write the virtual-loss deltas back
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j)
                if (o.h.add[j]) W.e[p][j].nif = o.h.nif0[j] + o.h.add[j];
        }
    }
}

}  // namespace ar
#endif  // __HIPCC__
