// The gather (search.rs:576-817, 961-999) with EIGHT LANES PER GAME: an "octet" of a wavefront walks one game's
// tree together, a wavefront holds eight games. Same node records, same arithmetic in the same order, same
// random draws as the lane-per-game gather of dev_search.h (gather_round) -- the two produce identical trees, batch
// entries and counters; which one runs is a scheduling choice (AR_GATHER).
//
// What the octet shares out:
//   * octet lane i < 5 owns outcome i of BOTH players: it loads the two 16-byte edge records e[0][i], e[1][i] of the
//     node that is entered, computes that outcome's q_norm / exploration numerator / forced flag / score
//     (search.rs:478-498) and keeps them while the node's visits are allocated; the five scores of a player meet in
//     every lane through five lane broadcasts (ds_swizzle), the best / second-best scan, the tie pass and its random
//     draws (search.rs:500-532) are then computed redundantly by all eight lanes on identical values, so no result
//     has to be sent back;
//   * octet lane j < 7 owns child slots 4j..4j+3: one 16-byte load brings its quarter of the 25-entry child table,
//     four registers hold the visits allocated to those slots (vtp, search.rs:757-798);
//   * the three header groups are loaded by all lanes from the same address (one transaction); a new node's
//     thirteen + seven groups are stored one group per lane.
// A node record therefore arrives with six load instructions per lane instead of twenty, the per-outcome divisions
// and square roots run once per lane instead of ten times, and eight lanes wait for a game's memory round trips
// instead of one -- so a SIMD keeps several wavefronts in flight where the lane-per-game layout, bounded by the HBM
// the trees take, has one.
// Everything that is not per outcome / per child slot (position, masks, counters, the random stream) is replicated:
// all lanes of an octet execute the same control flow on the same values.
#pragma once
#include "dev_search.h"

#if defined(__HIPCC__)
namespace ar {

// value of octet lane I (compile-time) in every lane of the octet: ds_swizzle, bit-mask mode, src = (lane & 0x18) | I
template <int I>
__device__ inline uint32_t oct_get(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (I << 5) | 0x18);
}
template <int I>
__device__ inline float oct_getf(float v) {
    return __uint_as_float(oct_get<I>(__float_as_uint(v)));
}
// value of octet lane `i` (run-time, the same in all lanes of the octet)
__device__ inline uint32_t oct_pick(uint32_t v, uint32_t i) {
    return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 56u) | i), 64);
}

// a level of the DFS kept while a deeper one is processed (the octet's form of Level<NW>)
template <int NW>
struct alignas(16) LevelO {
    uint32_t node, mask, omap0, omap1;
    uint32_t vtp[28];  // lane j < 7 keeps the visits of child slots 4j..4j+3 (16-byte aligned quarters)
    State<NW> saved;
};

// node.rs:251-283 compute_outcomes for the 17 effective-action maps a cell can have (each of the four moves open or
// blocked; or stuck in mud: everything is STAY), tabulated once per block: a new node's two maps are two LDS reads
// instead of two runs of pack_outcomes (~50 instructions each, executed by the whole wavefront whenever ANY of its
// eight games creates a node, i.e. in most rounds)
struct OutcomeTable {
    uint32_t omap[17];
    uint32_t n[17];
};
__device__ inline void outcome_table_fill(OutcomeTable& t) {  // lanes 0..16 of the block
    const uint32_t k = threadIdx.x;
    if (k < 17) {
        uint32_t eff = 4u << 12;
        if (k == 16) {
            eff = 4u | (4u << 3) | (4u << 6) | (4u << 9) | (4u << 12);
        } else {
            eff |= (k & 1u) ? 4u : 0u;            // bit d set: direction d is blocked -> STAY
            eff |= ((k & 2u) ? 4u : 1u) << 3;
            eff |= ((k & 4u) ? 4u : 2u) << 6;
            eff |= ((k & 8u) ? 4u : 3u) << 9;
        }
        uint32_t om, nn;
        pack_outcomes(eff, om, nn);
        t.omap[k] = om;
        t.n[k] = nn;
    }
}
// the table key of a player at `cell` with mud timer `mud` (same case split as eff_actions, dev_engine.h)
__device__ inline uint32_t outcome_key(const uint8_t* cost, uint8_t cell, uint8_t mud) {
    if (mud > 0) return 16u;
    const uint32_t c = cell_costs(cost, cell);
    return ((c & 0xffu) ? 0u : 1u) | (((c >> 8) & 0xffu) ? 0u : 2u) | (((c >> 16) & 0xffu) ? 0u : 4u) | (((c >> 24) & 0xffu) ? 0u : 8u);
}

// what an octet touches rarely lives in LDS (coherent inside the wavefront, no registers): the game's random
// stream -- only tie breaks draw from it, search.rs:511-532 -- and the root position, read once per pick
template <int NW>
struct OctShared {
    Rng rng;
    State<NW> root_st;
};

// one game's memory as the octet addresses it: two 64-bit arena pointers, everything else as 32-bit offsets
// from kernel-argument bases (scratch block, maze pool), so the address arithmetic stays in scalar registers
template <int NW>
struct OctMem {
    NodeStats* stats;
    unsigned char* scratch;  // uniform base of all games' scratch
    const uint8_t* maze;     // uniform base of the maze pool
    uint32_t s_off;          // this game's scratch block
    uint32_t maze_off;       // this game's cost table
    uint32_t proc_off, coll_off, levels_off, leaf_off;  // SlotLayout offsets (uniform)
    uint32_t coll_cap, max_depth;
    __device__ const uint8_t* cost() const { return maze + maze_off; }
    __device__ ProcEntry* proc() const { return (ProcEntry*)(scratch + (s_off + proc_off)); }
    __device__ CollEntry* coll() const { return (CollEntry*)(scratch + (s_off + coll_off)); }
    __device__ LevelO<NW>* levels() const { return (LevelO<NW>*)(scratch + (s_off + levels_off)); }
    __device__ State<NW>* leaves() const { return (State<NW>*)(scratch + (s_off + leaf_off)); }
};

template <int NW>
struct Oct {
    // ---- replicated ----
    bool done;
    uint32_t batch;
    long long left;
    uint32_t depth, node, mask, omap0, omap1, pick_mv;
    bool have_pick;
    State<NW> work;  // position at the current node
    uint32_t alloc_left;
    uint32_t n1, n2;  // outcomes of the node being allocated
    // mirrored slot fields
    uint32_t hi, cap, root, node_count, n_proc, n_coll, b_nn, b_term, b_coll, error, batch_active;
    uint32_t d_new, d_visits;  // new nodes / node records entered in this gather
    uint32_t rounds;           // rounds of this gather
    // ---- per octet lane ----
    float sc[2], util[2], num[2];  // lane i: outcome i of player 0 / 1
    uint32_t ns[2], add[2], nif0[2], forced;
    uint32_t kid[4], vtp[4];  // lane j: child slots 4j..4j+3
};

__device__ inline uint32_t sel4(const uint32_t* a, uint32_t c) { return c == 0 ? a[0] : c == 1 ? a[1] : c == 2 ? a[2] : a[3]; }

// search.rs:500-532 on five scores that every lane holds: best and second-best in outcome order, then the tie
// pass. The random stream is only fetched (from LDS) when some outcome ties with the best one -- exactly the
// cases in which the reference draws.
template <int NW>
__device__ inline void best_of5(const float* s, uint32_t n, OctShared<NW>& sh, uint32_t& best_out, float& second_out) {
    const float NEG_INF = -__builtin_inff();
    uint32_t best = 0;
    float best_score = NEG_INF, second = NEG_INF;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        if (i < n) {
            const float sc = s[i];
            if (sc > best_score) {
                second = best_score;
                best_score = sc;
                best = i;
            } else if (sc > second) {
                second = sc;
            }
        }
    }
    bool any_tie = false;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) any_tie = any_tie || (i < n && i != best && fabsf(s[i] - best_score) < 1e-12f);
    if (any_tie) {
        Rng rng = sh.rng;
        uint32_t ties = 1;
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) {
            if (i < n && i != best && fabsf(s[i] - best_score) < 1e-12f) {
                ties += 1;
                if (rng_below(rng, ties) == 0) best = i;
            }
        }
        // every lane stores the (identical) new state: each thread then reads back what it wrote itself, so the
        // compiler cannot keep a stale copy in registers for the lanes that would otherwise only read
        sh.rng = rng;
    }
    best_out = best;
    second_out = second;
}
// this lane's outcome as the best one: search.rs:534-553
__device__ inline uint32_t vtc_of(float util, float num, uint32_t ns, float second) {
    const float NEG_INF = -__builtin_inff();
    if (second <= NEG_INF) return 0xFFFFFFFFu;
    if (util >= second) return 0xFFFFFFFFu;
    const float denom = second - util;
    if (denom <= 0.0f) return 0xFFFFFFFFu;
    const float n1 = (float)ns + 1.0f;
    float vtc = num / denom - n1 + 1.0f;
    if (!(vtc > 1.0f)) vtc = 1.0f;
    const uint32_t k = vtc >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)vtc;
    return k > 1 ? k : 1;
}

// One round of an octet's gather: the same decisions in the same order as gather_round (dev_search.h).
template <int NW>
__device__ inline void gather8_round(Oct<NW>& o, OctShared<NW>& sh, const OutcomeTable& otab, const Board& board,
                                     const OctMem<NW>& m, const SearchCfg& cfg, uint32_t ol) {
    if (o.done) return;
    o.rounds += 1;
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
            if (ol < 7) {
                const uint4 v = *(const uint4*)&L.vtp[4 * ol];
                o.vtp[0] = v.x;
                o.vtp[1] = v.y;
                o.vtp[2] = v.z;
                o.vtp[3] = v.w;
                const uint4 k = *((const uint4*)&m.stats[L.node] + NODE_KID_GROUP + ol);
                o.kid[0] = k.x;
                o.kid[1] = k.y;
                o.kid[2] = k.z;
                o.kid[3] = k.w;
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
                const uint32_t k = oct_pick(sel4(o.vtp, idx & 3u), idx >> 2);
                const uint32_t o1 = idx / 5, o2 = idx % 5;
                float r1, r2;
                pos = o.work;
                st_step(board, m.cost(), pos, outcome_action(o.omap0, o1), outcome_action(o.omap1, o2), r1, r2);
                const uint32_t child = oct_pick(sel4(o.kid, idx & 3u), idx >> 2);
                if (child == NIL) {
                    // new leaf: shell creation + claim are stores only (tree.rs:107-148, search.rs:675-701)
                    if (o.hi >= o.cap) {  // excluded by the capacity check at the start of the gather
                        o.error = 3;
                    } else {
                        const uint32_t nid = o.hi++;
                        const bool over = st_over(board, pos);
                        // one 16-byte group per lane: the edge groups 0..9 are zero (prior 0, q 0, visits 0, nif 0),
                        // groups 10..12 the headers; the child table's seven groups are all NIL
                        uint4 g0 = make_uint4(0u, 0u, 0u, 0u), g1 = g0;
                        if (ol == 2) {  // h0: v1 0, v2 0, visits 0, nif 1 (try_start_score_update on a fresh node)
                            g1.w = 1u;
                        } else if (ol == 3) {  // h1: scale, edge rewards, parent
                            g1.x = __float_as_uint((float)(pos.remaining > 1 ? pos.remaining : 1));
                            g1.y = __float_as_uint(r1);
                            g1.z = __float_as_uint(r2);
                            g1.w = o.node;
                        } else if (ol == 4) {  // h2: outcome maps, counts, terminal flag
                            const uint32_t k1 = outcome_key(m.cost(), pos.p1, pos.m1), k2 = outcome_key(m.cost(), pos.p2, pos.m2);
                            g1.x = otab.omap[k1];
                            g1.y = otab.omap[k2];
                            g1.z = otab.n[k1] | (otab.n[k2] << 8) | (o1 << 16) | (o2 << 24);
                            g1.w = over ? 1u : 0u;
                        }
                        uint4* S = (uint4*)&m.stats[nid];
                        S[ol] = g0;
                        if (ol < 5) S[8 + ol] = g1;
                        if (ol < 7) S[NODE_KID_GROUP + ol] = make_uint4(NIL, NIL, NIL, NIL);
                        if (ol == 0) m.stats[o.node].c[idx] = nid;
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
                // the record of `rec`: two edge groups for lanes 0..4, the three headers for everybody, a quarter of
                // the child table for lanes 0..6 -- one round trip
                const NodeStats& N = m.stats[rec];
                Edge E1, E2;
                E1.prior = E1.q = E2.prior = E2.q = 0.0f;
                E1.visits = E1.nif = E2.visits = E2.nif = 0;
                if (ol < 5) {
                    E1 = N.e[0][ol];
                    E2 = N.e[1][ol];
                }
                const NodeH0 a = N.h0;
                const NodeH1 b = N.h1;
                const NodeH2 c = N.h2;
                uint4 kin = make_uint4(NIL, NIL, NIL, NIL);
                if (ol < 7) kin = *((const uint4*)&N + NODE_KID_GROUP + ol);
                if (a.visits == 0 || c.terminal != 0) {
                    // leaf or terminal (search.rs:591-636 for the root, :675-706 for a child)
                    emit_node = rec;
                    if (!(a.visits == 0 && a.nif > 0)) {  // try_start_score_update
                        const bool term = c.terminal != 0 || st_over(board, pos);
                        if (ol == 0) {
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
                    if (ol == 0) m.stats[rec].h0.nif = a.nif + visits_in;
                    if (!from_pick && o.mask != 0) {  // siblings still wait: keep the parent level for the way back
                        LevelO<NW>& L = m.levels()[o.depth];
                        if (ol == 0) {
                            L.node = o.node;
                            L.mask = o.mask;
                            L.omap0 = o.omap0;
                            L.omap1 = o.omap1;
                            L.saved = o.work;
                        }
                        if (ol < 7) *(uint4*)&L.vtp[4 * ol] = make_uint4(o.vtp[0], o.vtp[1], o.vtp[2], o.vtp[3]);
                        o.depth += 1;
                    }
                    if (from_pick) o.depth = 0;
                    o.work = pos;
                    // build_gather_level set-up (search.rs:742-774): this lane's outcome for both players
                    const uint32_t cv = a.visits > 0 ? a.visits - 1 : 0;
                    const uint32_t n1 = meta_n(c.meta, 0), n2 = meta_n(c.meta, 1);
                    const float c1 = (ol < n1 && E1.visits > 0) ? E1.prior : 0.0f;
                    const float c2 = (ol < n2 && E2.visits > 0) ? E2.prior : 0.0f;
                    // prior mass of the visited outcomes, summed in outcome order (x + 0.0f == x: an outcome that
                    // does not count adds nothing)
                    float mass1 = 0.0f, mass2 = 0.0f;
                    mass1 += oct_getf<0>(c1);
                    mass2 += oct_getf<0>(c2);
                    mass1 += oct_getf<1>(c1);
                    mass2 += oct_getf<1>(c2);
                    mass1 += oct_getf<2>(c1);
                    mass2 += oct_getf<2>(c2);
                    mass1 += oct_getf<3>(c1);
                    mass2 += oct_getf<3>(c2);
                    mass1 += oct_getf<4>(c1);
                    mass2 += oct_getf<4>(c2);
                    const float fpu1 = a.v1 - cfg.fpu_reduction * b.scale * sqrtf(mass1);
                    const float fpu2 = a.v2 - cfg.fpu_reduction * b.scale * sqrtf(mass2);
                    const float sqrt_total = sqrtf((float)(cv > 1 ? cv : 1));
                    o.forced = 0;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const Edge& e = pl == 0 ? E1 : E2;
                        const uint32_t n = pl == 0 ? n1 : n2;
                        const bool live = ol < n;
                        const float q = e.visits > 0 ? e.q : (pl == 0 ? fpu1 : fpu2);
                        o.util[pl] = q / b.scale;
                        o.num[pl] = cfg.c_puct * e.prior * sqrt_total;
                        o.ns[pl] = live ? e.visits + e.nif : 0;
                        o.nif0[pl] = e.nif;
                        o.add[pl] = 0;
                        float sc = o.util[pl] + o.num[pl] / (1.0f + (float)o.ns[pl]);
                        if (live && from_pick && cfg.force_k > 0.0f && e.prior > 0.0f) {
                            const float threshold = sqrtf(cfg.force_k * e.prior * (float)cv);
                            if ((float)e.visits < threshold) {
                                sc = 1e20f;
                                o.forced |= 1u << pl;
                            }
                        }
                        o.sc[pl] = sc;
                    }
                    o.n1 = n1;
                    o.n2 = n2;
                    o.node = rec;
                    o.omap0 = c.omap[0];
                    o.omap1 = c.omap[1];
                    o.mask = 0;
                    o.vtp[0] = o.vtp[1] = o.vtp[2] = o.vtp[3] = 0;
                    o.kid[0] = kin.x;
                    o.kid[1] = kin.y;
                    o.kid[2] = kin.z;
                    o.kid[3] = kin.w;
                    o.d_visits += 1;
                    o.alloc_left = visits_in;
                }
            }
            if (emit_kind != PROC_NONE) {
                if (o.n_proc >= cfg.batch_size) {
                    o.error = 1;
                } else {
                    const uint32_t i = o.n_proc++;
                    if (ol == 0) {
                        ProcEntry pe;
                        pe.node = emit_node;
                        pe.kind = (emit_kind == PROC_EVAL ? emit_kind | (o.b_nn << 8) : emit_kind) | (i << 16);
                        m.proc()[i] = pe;
                    }
                    if (emit_kind == PROC_EVAL) {
                        const uint32_t j = o.b_nn++;
                        if (ol == 1) m.leaves()[j] = pos;
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
                    if (ol == 0) {
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
            uint32_t b1 = 0, b2 = 0, c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu;
            if (o.n1 > 1) {  // (a single outcome never changes: search.rs:470-472)
                float s1[5], sec1;
                s1[0] = oct_getf<0>(o.sc[0]);
                s1[1] = oct_getf<1>(o.sc[0]);
                s1[2] = oct_getf<2>(o.sc[0]);
                s1[3] = oct_getf<3>(o.sc[0]);
                s1[4] = oct_getf<4>(o.sc[0]);
                best_of5(s1, o.n1, sh, b1, sec1);
                c1 = oct_pick(vtc_of(o.util[0], o.num[0], o.ns[0], sec1), b1);
            }
            if (o.n2 > 1) {
                float s2[5], sec2;
                s2[0] = oct_getf<0>(o.sc[1]);
                s2[1] = oct_getf<1>(o.sc[1]);
                s2[2] = oct_getf<2>(o.sc[1]);
                s2[3] = oct_getf<3>(o.sc[1]);
                s2[4] = oct_getf<4>(o.sc[1]);
                best_of5(s2, o.n2, sh, b2, sec2);
                c2 = oct_pick(vtc_of(o.util[1], o.num[1], o.ns[1], sec2), b2);
            }
            uint32_t k = o.alloc_left;
            if (c1 < k) k = c1;
            if (c2 < k) k = c2;
            if (k < 1) k = 1;
            const uint32_t flat = b1 * 5 + b2;
            {
                const bool mine = ol == (flat >> 2);
                const uint32_t w = flat & 3u;
                o.vtp[0] += (mine && w == 0) ? k : 0u;
                o.vtp[1] += (mine && w == 1) ? k : 0u;
                o.vtp[2] += (mine && w == 2) ? k : 0u;
                o.vtp[3] += (mine && w == 3) ? k : 0u;
            }
            o.mask |= 1u << flat;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {  // half_take: the chosen outcome's started count and score
                const uint32_t b = pl == 0 ? b1 : b2;
                const uint32_t nsb = o.ns[pl] + k;
                const float sc = o.util[pl] + o.num[pl] / (1.0f + (float)nsb);
                const bool hit = ol == b;
                o.ns[pl] = hit ? nsb : o.ns[pl];
                o.add[pl] += hit ? k : 0u;
                o.sc[pl] = (hit && !((o.forced >> pl) & 1u)) ? sc : o.sc[pl];
            }
            o.alloc_left -= k;
        }
        if (o.alloc_left == 0 && ol < 5) {
            NodeStats& W = m.stats[o.node];  // search.rs:800-814: write the virtual-loss deltas back
            if (o.add[0]) W.e[0][ol].nif = o.nif0[0] + o.add[0];
            if (o.add[1]) W.e[1][ol].nif = o.nif0[1] + o.add[1];
        }
    }
}

}  // namespace ar
#endif  // __HIPCC__
