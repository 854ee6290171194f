// The gather (search.rs:576-817, 961-999) with FOUR LANES PER GAME: a wavefront holds sixteen games. Same node records,
// same arithmetic in the same order, same random draws as the other gathers (gather_round: one lane per game,
// gather8_round: eight, gather2_round: two) -- identical trees, batch entries and counters; which one runs is a
// scheduling choice (AR_GATHER).
//
// Where it sits between the measured shapes ("what bounds the tree kernels"): eight lanes per game
// are bound by vector-instruction issue -- everything that is not per outcome is executed by eight lanes, and the
// per-outcome work uses five of them; two lanes per game issue a third of those instructions and are bound by their
// memory instructions, every one of which is 64 separate line accesses. With four lanes
//   * the replicated work is halved per game against the eight-lane kernel (sixteen games per wavefront);
//   * a record still arrives as whole lines: the four lanes of a game read four consecutive 16-byte groups per
//     instruction (edge groups 0..3 of a player are one 64-byte line), eight load instructions per record;
//   * lane q owns outcome q of BOTH players (q = 0..3) and lane p < 2 also outcome 4 of player p ("the fifth"): the
//     per-outcome arithmetic runs three times per wavefront instruction stream (player 1, player 2, the fifths)
//     instead of twice; the five scores of a player meet in every lane through five DPP quad broadcasts (no LDS);
//   * lane q owns the child-table quarters q and q + 4 (child slots 4q..4q+3 and 16+4q..): eight registers of child
//     ids, eight of allocated visits.
#pragma once
#include "dev_gather8.h"

#if defined(__HIPCC__)
namespace ar {

// value of quad lane I (compile-time) in every lane of the quad: DPP quad_perm [I, I, I, I]
template <int I>
__device__ inline uint32_t quad_get(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, I | (I << 2) | (I << 4) | (I << 6), 0xF, 0xF, true);
}
template <int I>
__device__ inline float quad_getf(float v) {
    return __uint_as_float(quad_get<I>(__float_as_uint(v)));
}
// value of quad lane `i` (run-time, the same in all lanes of the quad)
__device__ inline uint32_t quad_pick(uint32_t v, uint32_t i) {
    return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 60u) | i), 64);
}

// (an OR of masked words, never a conditional chain over the index: the optimizer turns such a chain into a dynamically
// indexed access, which puts the whole lane state into scratch memory)
__device__ inline uint32_t sel8(const uint32_t* a, uint32_t c) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) r |= (c == j) ? a[j] : 0u;
    return r;
}

template <int NW>
struct Quad {
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
    // ---- per quad lane q: outcome q of player 0 / 1 ----
    float sc[2], util[2], num[2];
    uint32_t ns[2], add[2], nif0[2], forced;
    // ---- lane p < 2: outcome 4 of player p ----
    float sc5, util5, num5;
    uint32_t ns5, add5, nif05, forced5;
    // ---- lane q: child slots 4q..4q+3 ([0..3]) and 16+4q..16+4q+3 ([4..7], q < 3) ----
    uint32_t kid[8], vtp[8];
};

// the five scores of player PL in every lane of the quad
template <int PL, int NW>
__device__ inline void quad_scores(const Quad<NW>& o, float* s) {
    s[0] = quad_getf<0>(o.sc[PL]);
    s[1] = quad_getf<1>(o.sc[PL]);
    s[2] = quad_getf<2>(o.sc[PL]);
    s[3] = quad_getf<3>(o.sc[PL]);
    s[4] = quad_getf<PL>(o.sc5);
}

// One round of a quad's gather: the same decisions in the same order as gather_round / gather8_round.
template <int NW>
__device__ inline void gather4_round(Quad<NW>& o, OctShared<NW>& sh, const OutcomeTable& otab, const Board& board,
                                     const OctMem<NW>& m, const SearchCfg& cfg, uint32_t ql) {
    if (o.done) return;
    o.rounds += 1;
    const bool has_hi = ql < 3;  // this lane's second quarter (4 + ql) exists
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
            {
                const uint4 v = *(const uint4*)&L.vtp[4 * ql];
                o.vtp[0] = v.x;
                o.vtp[1] = v.y;
                o.vtp[2] = v.z;
                o.vtp[3] = v.w;
                const uint4 k = *((const uint4*)&m.kids[L.node] + ql);
                o.kid[0] = k.x;
                o.kid[1] = k.y;
                o.kid[2] = k.z;
                o.kid[3] = k.w;
            }
            if (has_hi) {
                const uint4 v = *(const uint4*)&L.vtp[16 + 4 * ql];
                o.vtp[4] = v.x;
                o.vtp[5] = v.y;
                o.vtp[6] = v.z;
                o.vtp[7] = v.w;
                const uint4 k = *((const uint4*)&m.kids[L.node] + 4 + ql);
                o.kid[4] = k.x;
                o.kid[5] = k.y;
                o.kid[6] = k.z;
                o.kid[7] = k.w;
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
                // child slot idx: quarter idx / 4 lives in lane (idx / 4) & 3, registers 4 * (idx / 16) + idx % 4
                const uint32_t owner = (idx >> 2) & 3u, reg = ((idx >> 4) << 2) | (idx & 3u);
                const uint32_t k = quad_pick(sel8(o.vtp, reg), owner);
                const uint32_t o1 = idx / 5, o2 = idx % 5;
                float r1, r2;
                pos = o.work;
                st_step(board, m.cost(), pos, outcome_action(o.omap0, o1), outcome_action(o.omap1, o2), r1, r2);
                const uint32_t child = quad_pick(sel8(o.kid, reg), owner);
                if (child == NIL) {
                    // new leaf: shell creation + claim are stores only (tree.rs:107-148, search.rs:675-701)
                    if (o.hi >= o.cap) {  // excluded by the capacity check at the start of the gather
                        o.error = 3;
                    } else {
                        const uint32_t nid = o.hi++;
                        const bool over = st_over(board, pos);
                        // the record's 16-byte groups: 0..9 the edges (all zero), 10..12 the headers; lane q stores groups
                        // q, q + 4, q + 8 (and lane 0 group 12): h0 = group 10 (lane 2), h1 = 11 (lane 3), h2 = 12 (lane 0).
                        // The child table's seven groups are all NIL: lane q stores quarters q and q + 4.
                        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
                        uint4 g8 = zero;  // this lane's group q + 8
                        if (ql == 2) {    // h0: v1 0, v2 0, visits 0, nif 1 (try_start_score_update on a fresh node)
                            g8.w = 1u;
                        } else if (ql == 3) {  // h1: scale, edge rewards, parent
                            g8.x = __float_as_uint((float)(pos.remaining > 1 ? pos.remaining : 1));
                            g8.y = __float_as_uint(r1);
                            g8.z = __float_as_uint(r2);
                            g8.w = o.node;
                        }
                        const uint32_t k1 = outcome_key(m.cost(), pos.p1, pos.m1), k2 = outcome_key(m.cost(), pos.p2, pos.m2);
                        uint4* S = (uint4*)&m.stats[nid];
                        S[ql] = zero;
                        S[4 + ql] = zero;
                        S[8 + ql] = g8;
                        if (ql == 0)  // h2: outcome maps, counts, terminal flag
                            S[12] = make_uint4(otab.omap[k1], otab.omap[k2], otab.n[k1] | (otab.n[k2] << 8) | (o1 << 16) | (o2 << 24),
                                               over ? 1u : 0u);
                        uint4* K = (uint4*)&m.kids[nid];
                        K[ql] = make_uint4(NIL, NIL, NIL, NIL);
                        if (has_hi) K[4 + ql] = make_uint4(NIL, NIL, NIL, NIL);
                        if (ql == 0) m.kids[o.node].c[idx] = nid;
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
                // the record of `rec`: edge q of both players, lanes 0 / 1 the fifth edge of player 0 / 1, the three
                // headers for everybody, two quarters of the child table -- one round trip
                const NodeStats& N = m.stats[rec];
                const Edge E1 = N.e[0][ql], E2 = N.e[1][ql];
                Edge E5;
                E5.prior = E5.q = 0.0f;
                E5.visits = E5.nif = 0;
                if (ql < 2) E5 = N.e[ql][4];
                const NodeH0 a = N.h0;
                const NodeH1 b = N.h1;
                const NodeH2 c = N.h2;
                const uint4 kin0 = *((const uint4*)&m.kids[rec] + ql);
                uint4 kin1 = make_uint4(NIL, NIL, NIL, NIL);
                if (has_hi) kin1 = *((const uint4*)&m.kids[rec] + 4 + ql);
                if (a.visits == 0 || c.terminal != 0) {
                    // leaf or terminal (search.rs:591-636 for the root, :675-706 for a child)
                    emit_node = rec;
                    if (!(a.visits == 0 && a.nif > 0)) {  // try_start_score_update
                        const bool term = c.terminal != 0 || st_over(board, pos);
                        if (ql == 0) {
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
                    if (ql == 0) m.stats[rec].h0.nif = a.nif + visits_in;
                    if (!from_pick && o.mask != 0) {  // siblings still wait: keep the parent level for the way back
                        LevelO<NW>& L = m.levels()[o.depth];
                        if (ql == 0) {
                            L.node = o.node;
                            L.mask = o.mask;
                            L.omap0 = o.omap0;
                            L.omap1 = o.omap1;
                            L.saved = o.work;
                        }
                        *(uint4*)&L.vtp[4 * ql] = make_uint4(o.vtp[0], o.vtp[1], o.vtp[2], o.vtp[3]);
                        if (has_hi) *(uint4*)&L.vtp[16 + 4 * ql] = make_uint4(o.vtp[4], o.vtp[5], o.vtp[6], o.vtp[7]);
                        o.depth += 1;
                    }
                    if (from_pick) o.depth = 0;
                    o.work = pos;
                    // build_gather_level set-up (search.rs:742-774): this lane's outcomes
                    const uint32_t cv = a.visits > 0 ? a.visits - 1 : 0;
                    const uint32_t n1 = meta_n(c.meta, 0), n2 = meta_n(c.meta, 1);
                    const uint32_t n_mine = ql == 0 ? n1 : n2;  // (lanes 0 / 1: the player whose fifth outcome this lane owns)
                    const float c1 = (ql < n1 && E1.visits > 0) ? E1.prior : 0.0f;
                    const float c2 = (ql < n2 && E2.visits > 0) ? E2.prior : 0.0f;
                    const float c5 = (ql < 2 && 4u < n_mine && E5.visits > 0) ? E5.prior : 0.0f;
                    // prior mass of the visited outcomes, summed in outcome order (x + 0.0f == x: an outcome that
                    // does not count adds nothing)
                    float mass1 = 0.0f, mass2 = 0.0f;
                    mass1 += quad_getf<0>(c1);
                    mass2 += quad_getf<0>(c2);
                    mass1 += quad_getf<1>(c1);
                    mass2 += quad_getf<1>(c2);
                    mass1 += quad_getf<2>(c1);
                    mass2 += quad_getf<2>(c2);
                    mass1 += quad_getf<3>(c1);
                    mass2 += quad_getf<3>(c2);
                    mass1 += quad_getf<0>(c5);
                    mass2 += quad_getf<1>(c5);
                    const float fpu1 = a.v1 - cfg.fpu_reduction * b.scale * sqrtf(mass1);
                    const float fpu2 = a.v2 - cfg.fpu_reduction * b.scale * sqrtf(mass2);
                    const float sqrt_total = sqrtf((float)(cv > 1 ? cv : 1));
                    o.forced = 0;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const Edge& e = pl == 0 ? E1 : E2;
                        const uint32_t n = pl == 0 ? n1 : n2;
                        const bool live = ql < n;
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
                    {  // the fifth outcome of player ql (lanes 0 and 1)
                        const bool live = ql < 2 && 4u < n_mine;
                        const float q = E5.visits > 0 ? E5.q : (ql == 0 ? fpu1 : fpu2);
                        o.util5 = q / b.scale;
                        o.num5 = cfg.c_puct * E5.prior * sqrt_total;
                        o.ns5 = live ? E5.visits + E5.nif : 0;
                        o.nif05 = E5.nif;
                        o.add5 = 0;
                        o.forced5 = 0;
                        float sc = o.util5 + o.num5 / (1.0f + (float)o.ns5);
                        if (live && from_pick && cfg.force_k > 0.0f && E5.prior > 0.0f) {
                            const float threshold = sqrtf(cfg.force_k * E5.prior * (float)cv);
                            if ((float)E5.visits < threshold) {
                                sc = 1e20f;
                                o.forced5 = 1;
                            }
                        }
                        o.sc5 = sc;
                    }
                    o.n1 = n1;
                    o.n2 = n2;
                    o.node = rec;
                    o.omap0 = c.omap[0];
                    o.omap1 = c.omap[1];
                    o.mask = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o.vtp[j] = 0;
                    o.kid[0] = kin0.x;
                    o.kid[1] = kin0.y;
                    o.kid[2] = kin0.z;
                    o.kid[3] = kin0.w;
                    o.kid[4] = kin1.x;
                    o.kid[5] = kin1.y;
                    o.kid[6] = kin1.z;
                    o.kid[7] = kin1.w;
                    o.d_visits += 1;
                    o.alloc_left = visits_in;
                }
            }
            if (emit_kind != PROC_NONE) {
                if (o.n_proc >= cfg.batch_size) {
                    o.error = 1;
                } else {
                    const uint32_t i = o.n_proc++;
                    if (ql == 0) {
                        ProcEntry pe;
                        pe.node = emit_node;
                        pe.kind = emit_kind;
                        m.proc()[i] = pe;
                    }
                    if (emit_kind == PROC_EVAL) {
                        const uint32_t j = o.b_nn++;
                        if (ql == 1) m.leaves()[j] = pos;
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
                    if (ql == 0) {
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
                quad_scores<0>(o, s1);
                best_of5(s1, o.n1, sh, b1, sec1);
                const uint32_t va = vtc_of(o.util[0], o.num[0], o.ns[0], sec1), v5 = vtc_of(o.util5, o.num5, o.ns5, sec1);
                const uint32_t from_a = quad_pick(va, b1 & 3u), from_5 = quad_get<0>(v5);
                c1 = b1 == 4 ? from_5 : from_a;
            }
            if (o.n2 > 1) {
                float s2[5], sec2;
                quad_scores<1>(o, s2);
                best_of5(s2, o.n2, sh, b2, sec2);
                const uint32_t va = vtc_of(o.util[1], o.num[1], o.ns[1], sec2), v5 = vtc_of(o.util5, o.num5, o.ns5, sec2);
                const uint32_t from_a = quad_pick(va, b2 & 3u), from_5 = quad_get<1>(v5);
                c2 = b2 == 4 ? from_5 : from_a;
            }
            uint32_t k = o.alloc_left;
            if (c1 < k) k = c1;
            if (c2 < k) k = c2;
            if (k < 1) k = 1;
            const uint32_t flat = b1 * 5 + b2;
            {
                const bool mine = ql == ((flat >> 2) & 3u);
                const uint32_t reg = ((flat >> 4) << 2) | (flat & 3u);
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) o.vtp[j] += (mine && reg == j) ? k : 0u;
            }
            o.mask |= 1u << flat;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {  // half_take: the chosen outcome's started count and score
                const uint32_t b = pl == 0 ? b1 : b2;
                const uint32_t nsb = o.ns[pl] + k;
                const float sc = o.util[pl] + o.num[pl] / (1.0f + (float)nsb);
                const bool hit = ql == b;  // (b == 4: nobody here)
                o.ns[pl] = hit ? nsb : o.ns[pl];
                o.add[pl] += hit ? k : 0u;
                o.sc[pl] = (hit && !((o.forced >> pl) & 1u)) ? sc : o.sc[pl];
            }
            {  // the fifths: lane p, when player p's best is outcome 4
                const uint32_t b = ql == 0 ? b1 : b2;
                const uint32_t nsb = o.ns5 + k;
                const float sc = o.util5 + o.num5 / (1.0f + (float)nsb);
                const bool hit = ql < 2 && b == 4;
                o.ns5 = hit ? nsb : o.ns5;
                o.add5 += hit ? k : 0u;
                o.sc5 = (hit && !o.forced5) ? sc : o.sc5;
            }
            o.alloc_left -= k;
        }
        if (o.alloc_left == 0) {
            NodeStats& W = m.stats[o.node];  // search.rs:800-814: write the virtual-loss deltas back
            if (o.add[0]) W.e[0][ql].nif = o.nif0[0] + o.add[0];
            if (o.add[1]) W.e[1][ql].nif = o.nif0[1] + o.add[1];
            if (ql < 2 && o.add5) W.e[ql][4].nif = o.nif05 + o.add5;
        }
    }
}

}  // namespace ar
#endif  // __HIPCC__
