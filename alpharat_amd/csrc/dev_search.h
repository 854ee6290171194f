// Decoupled-PUCT tree search on device-resident node arenas: the MI355X-side counterpart of
// crates/alpharat-mcts/src/{node,tree,search}.rs and the per-move part of
// crates/alpharat-sampling/src/selfplay.rs:515-598.
//
// Layout (one arena per game, all in HBM):
//   NodeStats  256 B / node, 16-B edge records {prior, q, visits, in_flight} x 5 outcomes x 2
//              players, then the node header. Everything select and backup touch is in these two
//              128-B lines.
//   NodeKids   128 B / node, a 25-slot child table indexed by (p1_outcome*5 + p2_outcome)
//              (replaces the reference's linked list + find_child walk, tree.rs:52-63).
// Node ids are arena indices; the live tree occupies [lo, hi) and new nodes bump `hi`. Moving the
// root (tree reuse, tree.rs:283-295) copies the kept subtree in DFS order into the free part of
// the same arena, which also yields the recounted node_count the collision budget depends on.
//
// All arithmetic keeps the reference's f32 operation order; build with -ffp-contract=off.
#pragma once
#include "dev_engine.h"

namespace ar {

static const uint32_t NIL = 0xFFFFFFFFu;

struct Edge {
    float prior;
    float q;
    uint32_t visits;
    uint32_t nif;  // n_in_flight (virtual loss)
};

struct alignas(128) NodeStats {
    Edge e[2][5];        // [player][outcome]                     160 B
    float v1, v2;        // Welford means
    uint32_t visits;     // total_visits
    uint32_t nif;        // n_in_flight
    float scale;         // value_scale = max(remaining_cheese, 1) at creation
    float r1, r2;        // edge rewards from the parent
    uint32_t parent;     // NIL for the root
    uint32_t omap[2];    // per player: outcome->action 3 bits each (bits 0..14), action->outcome (bits 15..29)
    uint8_t n[2];        // n_outcomes
    uint8_t po[2];       // parent_outcome
    uint32_t terminal;   // is_terminal
    uint32_t pad[12];
};
struct alignas(128) NodeKids {
    uint32_t c[25];
    uint32_t pad[7];
};

struct SearchCfg {
    float c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration;
    uint32_t coll_min, coll_max, coll_start, coll_end;
    float coll_power;
    uint32_t n_sims, batch_size;
};

enum { SLOT_EMPTY = 0, SLOT_ACTIVE = 1, SLOT_DONE = 2, SLOT_STALL = 3, SLOT_FAILED = 4 };
enum { PROC_TERMINAL = 0, PROC_EVAL = 1 };

struct ProcEntry {
    uint32_t node;
    uint32_t kind;
};
struct CollEntry {
    uint32_t node;
    uint32_t mv;
};
struct EvalOut {
    float p1[5], p2[5];
    float v1, v2;
};

template <int NW>
struct Level {
    uint32_t node;
    uint16_t next_idx, last_idx;
    uint16_t vtp[25];
    uint16_t pad;
    State<NW> saved;  // position before the move that led to the next level
};
struct CopyFrame {
    uint32_t old_id, new_id, slot;
};

// search.rs:304-325 plus the sampled actions of the move
struct MoveResult {
    float policy[2][5];
    float value[2];
    float visit_counts[2][5];
    float prior[2][5];
    uint32_t total_visits, nn_evals, terminals, collisions;
};

template <int NW>
struct PosRec {  // selfplay.rs:80-102 PositionRecord
    State<NW> st;
    MoveResult res;
    uint8_t a1, a2;
    uint8_t pad[6];
};

template <int NW>
struct Slot {
    Board board;
    State<NW> st;
    Rng rng;
    uint32_t game_index;
    uint32_t status;
    // arena
    NodeStats* stats;
    NodeKids* kids;
    uint32_t cap, lo, hi, root, node_count;
    uint32_t pending_root;  // subtree to keep when the slot is stalled for a bigger arena
    uint32_t need_nodes;    // capacity the stalled slot asks for
    // current search
    uint32_t remaining;
    uint32_t s_nn, s_term, s_coll;
    // current batch
    uint32_t n_proc, n_coll, eval_base, b_nn, b_term, b_coll, batch_active;
    // game totals
    uint32_t n_pos;
    uint32_t single_search;  // 1: ar_search mode -- stop after one search, do not move
    uint64_t t_sims, t_nn, t_term, t_coll;
    uint64_t nv_gather, nv_backup, new_nodes;
    // per-slot scratch
    ProcEntry* proc;      // [batch_size]
    CollEntry* coll;      // [coll_cap]
    Level<NW>* levels;    // [max_depth]
    CopyFrame* frames;    // [max_depth]
    EvalOut* ev_local;    // [batch_size] evaluator outputs when the evaluator runs inline
    State<NW>* leaf_local;// [batch_size] leaf positions (host-callback evaluator)
    PosRec<NW>* pos;      // [max_turns]
    uint32_t coll_cap, max_depth;
    MoveResult last;      // result of the last finished search
    uint32_t error;       // non-zero: internal capacity violation (bug guard)
    uint32_t pad0;
};

// A leaf waiting for the device-wide evaluator (replaces MuxBackend's request queue, mux.rs:170-289)
template <int NW>
struct LeafReq {
    State<NW> st;
    uint32_t slot;
    uint32_t pad;
};

// ---- node.rs:251-283 compute_outcomes, packed ------------------------------------------------
AR_HD void pack_outcomes(uint32_t eff, uint32_t& omap, uint8_t& n_out) {
    uint32_t present = 0;
    for (int a = 0; a < 5; ++a) present |= 1u << ((eff >> (3 * a)) & 7u);
    uint32_t m = 0;
    uint32_t rank[5];
    uint32_t cnt = 0;
    for (uint32_t act = 0; act < 5; ++act) {
        rank[act] = cnt;
        if (present & (1u << act)) {
            m |= act << (3 * cnt);
            ++cnt;
        }
    }
    for (int a = 0; a < 5; ++a) m |= rank[(eff >> (3 * a)) & 7u] << (15 + 3 * a);
    omap = m;
    n_out = (uint8_t)cnt;
}
AR_HD uint32_t outcome_action(uint32_t omap, uint32_t idx) { return (omap >> (3 * idx)) & 7u; }
AR_HD uint32_t action_outcome(uint32_t omap, uint32_t act) { return (omap >> (15 + 3 * act)) & 7u; }

// node.rs:173-179 set_prior: clear, scatter-add in action order
AR_HD void set_prior(NodeStats& nd, int pl, const float* p5) {
    for (int i = 0; i < 5; ++i) nd.e[pl][i].prior = 0.0f;
    for (uint32_t a = 0; a < 5; ++a) nd.e[pl][action_outcome(nd.omap[pl], a)].prior += p5[a];
}
// tree.rs:69-84 smart_uniform_prior from a packed effective-action map
AR_HD void uniform_prior(uint32_t eff, float* p5) {
    uint32_t present = 0;
    for (int a = 0; a < 5; ++a) present |= 1u << ((eff >> (3 * a)) & 7u);
    uint32_t cnt = 0;
    for (int a = 0; a < 5; ++a) cnt += (present >> a) & 1u;
    const float p = 1.0f / (float)cnt;
    for (int a = 0; a < 5; ++a) p5[a] = (present >> a) & 1u ? p : 0.0f;
}

AR_HD void init_shell(NodeStats& nd, NodeKids& kd, uint32_t eff1, uint32_t eff2, uint16_t remaining,
                      uint32_t parent, uint32_t o1, uint32_t o2, float r1, float r2) {
    for (int pl = 0; pl < 2; ++pl)
        for (int i = 0; i < 5; ++i) {
            nd.e[pl][i].prior = 0.0f;
            nd.e[pl][i].q = 0.0f;
            nd.e[pl][i].visits = 0;
            nd.e[pl][i].nif = 0;
        }
    nd.v1 = 0.0f;
    nd.v2 = 0.0f;
    nd.visits = 0;
    nd.nif = 0;
    nd.scale = (float)(remaining > 1 ? remaining : 1);
    nd.r1 = r1;
    nd.r2 = r2;
    nd.parent = parent;
    pack_outcomes(eff1, nd.omap[0], nd.n[0]);
    pack_outcomes(eff2, nd.omap[1], nd.n[1]);
    nd.po[0] = (uint8_t)o1;
    nd.po[1] = (uint8_t)o2;
    nd.terminal = 0;
    for (int i = 0; i < 25; ++i) kd.c[i] = NIL;
}

// tree.rs:351-365 alloc_root at arena index `at`
template <int NW>
AR_HD void make_root(Slot<NW>& s, const uint8_t* cost, uint32_t at) {
    const uint32_t e1 = eff_actions(cost, s.st.p1, s.st.m1), e2 = eff_actions(cost, s.st.p2, s.st.m2);
    NodeStats& nd = s.stats[at];
    init_shell(nd, s.kids[at], e1, e2, s.st.remaining, NIL, 0, 0, 0.0f, 0.0f);
    float p[5];
    uniform_prior(e1, p);
    set_prior(nd, 0, p);
    uniform_prior(e2, p);
    set_prior(nd, 1, p);
    s.root = at;
    s.lo = at;
    s.hi = at + 1;
    s.node_count = 1;
}

// ---- search.rs:120-152 ----------------------------------------------------------------------
AR_HD float fpu_of(const Edge* e, int n, float node_value, float scale, float fpu_reduction) {
    float mass = 0.0f;
    for (int i = 0; i < n; ++i)
        if (e[i].visits > 0) mass += e[i].prior;
    return node_value - fpu_reduction * scale * sqrtf(mass);
}

// search.rs:463-554 estimated_visits_to_change_best_half
AR_HD void visits_to_change_best(const Edge* e, int n, float node_value, float scale, uint32_t children_visits,
                                 const SearchCfg& cfg, bool is_root, const uint32_t* nstarted, Rng& rng,
                                 uint32_t& best_out, uint32_t& vtc_out) {
    if (n <= 1) {
        best_out = 0;
        vtc_out = 0xFFFFFFFFu;
        return;
    }
    const float fpu = fpu_of(e, n, node_value, scale, cfg.fpu_reduction);
    const float sqrt_total = sqrtf((float)(children_visits > 1 ? children_visits : 1));
    const float NEG_INF = -__builtin_inff();
    float score[5], util[5];
    for (int i = 0; i < n; ++i) {
        const float q = e[i].visits > 0 ? e[i].q : fpu;
        const float qn = q / scale;
        float sc = qn + cfg.c_puct * e[i].prior * sqrt_total / (1.0f + (float)nstarted[i]);
        if (is_root && cfg.force_k > 0.0f && e[i].prior > 0.0f) {
            const float threshold = sqrtf(cfg.force_k * e[i].prior * (float)children_visits);
            if ((float)e[i].visits < threshold) sc = 1e20f;
        }
        score[i] = sc;
        util[i] = qn;
    }
    uint32_t best = 0;
    float best_score = NEG_INF, best_util = NEG_INF, second = NEG_INF;
    for (int i = 0; i < n; ++i) {
        if (score[i] > best_score) {
            second = best_score;
            best_score = score[i];
            best = (uint32_t)i;
            best_util = util[i];
        } else if (score[i] > second) {
            second = score[i];
        }
    }
    uint32_t ties = 1;
    for (int i = 0; i < n; ++i) {
        if ((uint32_t)i == best) continue;
        if (fabsf(score[i] - best_score) < 1e-12f) {
            ties += 1;
            if (rng_below(rng, ties) == 0) {
                best = (uint32_t)i;
                best_util = util[i];
            }
        }
    }
    best_out = best;
    vtc_out = 0xFFFFFFFFu;
    if (second <= NEG_INF) return;
    if (best_util >= second) return;
    const float denom = second - best_util;
    if (denom <= 0.0f) return;
    const float n1 = (float)nstarted[best] + 1.0f;
    float vtc = cfg.c_puct * e[best].prior * sqrt_total / denom - n1 + 1.0f;
    if (!(vtc > 1.0f)) vtc = 1.0f;
    const uint32_t k = vtc >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)vtc;
    vtc_out = k > 1 ? k : 1;
}

// search.rs:742-817 build_gather_level
template <int NW>
AR_HD void build_level(NodeStats& nd, uint32_t node, uint32_t cur_limit, const SearchCfg& cfg, bool is_root,
                       Rng& rng, Level<NW>& lv) {
    const int n1 = nd.n[0], n2 = nd.n[1];
    const uint32_t cv = nd.visits > 0 ? nd.visits - 1 : 0;
    const float scale = nd.scale, v1 = nd.v1, v2 = nd.v2;
    Edge e1[5], e2[5];
    uint32_t ns1[5], ns2[5], add1[5], add2[5];
    for (int i = 0; i < 5; ++i) {
        e1[i] = nd.e[0][i];
        e2[i] = nd.e[1][i];
        ns1[i] = i < n1 ? e1[i].visits + e1[i].nif : 0;
        ns2[i] = i < n2 ? e2[i].visits + e2[i].nif : 0;
        add1[i] = 0;
        add2[i] = 0;
    }
    lv.node = node;
    for (int i = 0; i < 25; ++i) lv.vtp[i] = 0;
    uint32_t remaining = cur_limit, last = 0;
    while (remaining > 0) {
        uint32_t b1, b2, c1, c2;
        visits_to_change_best(e1, n1, v1, scale, cv, cfg, is_root, ns1, rng, b1, c1);
        visits_to_change_best(e2, n2, v2, scale, cv, cfg, is_root, ns2, rng, b2, c2);
        uint32_t k = remaining;
        if (c1 < k) k = c1;
        if (c2 < k) k = c2;
        if (k < 1) k = 1;
        const uint32_t flat = b1 * 5 + b2;
        lv.vtp[flat] = (uint16_t)(lv.vtp[flat] + k);
        ns1[b1] += k;
        ns2[b2] += k;
        add1[b1] += k;
        add2[b2] += k;
        remaining -= k;
        if (flat > last) last = flat;
    }
    for (int i = 0; i < n1; ++i)
        if (add1[i]) nd.e[0][i].nif += add1[i];
    for (int j = 0; j < n2; ++j)
        if (add2[j]) nd.e[1][j].nif += add2[j];
    lv.next_idx = 0;
    lv.last_idx = (uint16_t)last;
}

AR_HD bool try_start(NodeStats& nd) {  // node.rs:388-394
    if (nd.visits == 0 && nd.nif > 0) return false;
    nd.nif += 1;
    return true;
}

// search.rs:437-450
AR_HD uint32_t collisions_left(uint32_t node_count, const SearchCfg& c) {
    if (node_count >= c.coll_end) return c.coll_max;
    if (node_count <= c.coll_start) return c.coll_min;
    const float ratio = (float)(node_count - c.coll_start) / (float)(c.coll_end - c.coll_start);
    const float scaled = (float)c.coll_min + ((float)c.coll_max - (float)c.coll_min) * powf(ratio, c.coll_power);
    const float r = roundf(scaled);
    uint32_t v = r <= 0.0f ? 0u : (r >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)r);
    if (v < c.coll_min) v = c.coll_min;
    if (v > c.coll_max) v = c.coll_max;
    return v;
}

// How a gathered leaf is handed to the evaluator.
//  EVAL_UNIFORM: SmartUniform computed on the spot into ev_local (backend.rs:92-103)
//  EVAL_STORE:   position kept in leaf_local; the step kernel then appends the batch's leaves to
//                the device-wide leaf queue (network evaluators) or the host reads them back
//                (predict_fn callback)
enum { EVAL_UNIFORM = 0, EVAL_STORE = 1 };

struct GatherCtx {
    const uint8_t* cost;
    int eval_mode;
};

template <int NW>
AR_HD void push_proc(Slot<NW>& s, const GatherCtx& cx, const SearchCfg& cfg, uint32_t node, uint32_t kind,
                     const State<NW>& leaf) {
    if (s.n_proc >= cfg.batch_size) {
        s.error = 1;
        return;
    }
    const uint32_t i = s.n_proc++;
    s.proc[i].node = node;
    s.proc[i].kind = kind;
    if (kind == PROC_EVAL) {
        const uint32_t j = s.b_nn++;
        if (cx.eval_mode == EVAL_UNIFORM) {
            EvalOut& o = s.ev_local[j];
            uniform_prior(eff_actions(cx.cost, leaf.p1, leaf.m1), o.p1);
            uniform_prior(eff_actions(cx.cost, leaf.p2, leaf.m2), o.p2);
            o.v1 = 0.0f;
            o.v2 = 0.0f;
        } else {
            s.leaf_local[j] = leaf;
        }
    } else {
        s.b_term += 1;
    }
}
template <int NW>
AR_HD void push_coll(Slot<NW>& s, uint32_t node, uint32_t mv) {
    if (s.n_coll >= s.coll_cap) {
        s.error = 2;
        return;
    }
    s.coll[s.n_coll].node = node;
    s.coll[s.n_coll].mv = mv;
    s.n_coll += 1;
}

// search.rs:576-738 pick_nodes_to_extend; returns the collision multivisits it produced
template <int NW>
AR_HD uint32_t pick_nodes(Slot<NW>& s, const GatherCtx& cx, const SearchCfg& cfg, uint32_t budget) {
    const uint32_t root = s.root;
    NodeStats& R = s.stats[root];
    uint32_t coll_mv = 0;
    State<NW> work = s.st;
    if (R.visits == 0 || R.terminal) {
        if (R.visits == 0 && !R.terminal) {
            if (try_start(R)) {
                if (st_over(s.board, work)) {
                    R.terminal = 1;
                    push_proc(s, cx, cfg, root, PROC_TERMINAL, work);
                } else {
                    push_proc(s, cx, cfg, root, PROC_EVAL, work);
                }
                if (budget > 1) {
                    push_coll(s, root, budget - 1);
                    coll_mv += budget - 1;
                }
            } else {
                push_coll(s, root, budget);
                coll_mv += budget;
            }
        } else {
            if (R.visits == 0) R.terminal = 1;
            if (try_start(R)) {
                push_proc(s, cx, cfg, root, PROC_TERMINAL, work);
                if (budget > 1) {
                    push_coll(s, root, budget - 1);
                    coll_mv += budget - 1;
                }
            } else {
                push_coll(s, root, budget);
                coll_mv += budget;
            }
        }
        return coll_mv;
    }

    R.nif += budget;
    build_level(R, root, budget, cfg, true, s.rng, s.levels[0]);
    s.nv_gather += 1;
    uint32_t depth = 1;
    while (depth > 0) {
        Level<NW>& L = s.levels[depth - 1];
        bool descended = false;
        while (L.next_idx <= L.last_idx) {
            const uint32_t idx = L.next_idx;
            L.next_idx += 1;
            const uint32_t k = L.vtp[idx];
            if (k == 0) continue;
            const uint32_t o1 = idx / 5, o2 = idx % 5;
            const uint32_t node = L.node;
            NodeStats& N = s.stats[node];
            const uint32_t act1 = outcome_action(N.omap[0], o1), act2 = outcome_action(N.omap[1], o2);
            const State<NW> before = work;
            float r1, r2;
            st_step(s.board, cx.cost, work, act1, act2, r1, r2);
            uint32_t child = s.kids[node].c[idx];
            if (child == NIL) {
                if (s.hi >= s.cap) {  // guarded by the pre-batch capacity check
                    s.error = 3;
                    work = before;
                    continue;
                }
                child = s.hi++;
                init_shell(s.stats[child], s.kids[child], eff_actions(cx.cost, work.p1, work.m1),
                           eff_actions(cx.cost, work.p2, work.m2), work.remaining, node, o1, o2, r1, r2);
                s.kids[node].c[idx] = child;
                s.node_count += 1;
                s.new_nodes += 1;
            }
            NodeStats& C = s.stats[child];
            if (C.visits == 0 || C.terminal) {
                if (try_start(C)) {
                    if (C.terminal || st_over(s.board, work)) {
                        if (C.visits == 0) C.terminal = 1;
                        push_proc(s, cx, cfg, child, PROC_TERMINAL, work);
                    } else {
                        push_proc(s, cx, cfg, child, PROC_EVAL, work);
                    }
                    if (k > 1) {
                        push_coll(s, child, k - 1);
                        coll_mv += k - 1;
                    }
                } else {
                    push_coll(s, child, k);
                    coll_mv += k;
                }
                work = before;
            } else {
                C.nif += k;  // try_start_score_update (always succeeds on a visited node) + k-1
                L.saved = before;
                if (depth >= s.max_depth) {
                    s.error = 4;
                    work = before;
                    continue;
                }
                build_level(C, child, k, cfg, false, s.rng, s.levels[depth]);
                s.nv_gather += 1;
                depth += 1;
                descended = true;
                break;
            }
        }
        if (!descended) {
            depth -= 1;
            if (depth > 0) work = s.levels[depth - 1].saved;
        }
    }
    return coll_mv;
}

// node.rs:444-457 + node.rs:82-85
AR_HD void finalize_node(NodeStats& nd, float q1, float q2, uint32_t mv) {
    nd.visits += mv;
    const float n = (float)nd.visits, w = (float)mv;
    nd.v1 += (q1 - nd.v1) * w / n;
    nd.v2 += (q2 - nd.v2) * w / n;
    nd.nif -= mv;
}
AR_HD void edge_update(Edge& e, float value, uint32_t mv) {
    e.visits += mv;
    e.q += (value - e.q) * (float)mv / (float)e.visits;
    e.nif -= mv;
}

// search.rs:826-852 backup_and_finalize
template <int NW>
AR_HD void backup_path(Slot<NW>& s, uint32_t leaf, float g1, float g2, uint32_t mv) {
    finalize_node(s.stats[leaf], g1, g2, mv);
    s.nv_backup += 1;
    float v1 = g1, v2 = g2;
    uint32_t cur = leaf;
    for (;;) {
        const NodeStats& C = s.stats[cur];
        const uint32_t parent = C.parent;
        if (parent == NIL) break;
        const float q1 = C.r1 + v1, q2 = C.r2 + v2;
        const uint32_t a1 = C.po[0], a2 = C.po[1];
        NodeStats& P = s.stats[parent];
        finalize_node(P, q1, q2, mv);
        edge_update(P.e[0][a1], q1, mv);
        edge_update(P.e[1][a2], q2, mv);
        s.nv_backup += 1;
        v1 = q1;
        v2 = q2;
        cur = parent;
    }
}

// search.rs:860-889 cancel_shared_collisions (the root has no parent, so the walk ends there)
template <int NW>
AR_HD void cancel_collisions(Slot<NW>& s) {
    for (uint32_t i = 0; i < s.n_coll; ++i) {
        const uint32_t mv = s.coll[i].mv;
        uint32_t cur = s.coll[i].node;
        for (;;) {
            const NodeStats& C = s.stats[cur];
            const uint32_t parent = C.parent;
            if (parent == NIL) break;
            NodeStats& P = s.stats[parent];
            P.nif -= mv;
            P.e[0][C.po[0]].nif -= mv;
            P.e[1][C.po[1]].nif -= mv;
            cur = parent;
        }
    }
}
// search.rs:899-910 + :945-955: revert a gathered batch after an evaluator failure
template <int NW>
AR_HD void cancel_batch(Slot<NW>& s) {
    for (uint32_t i = 0; i < s.n_proc; ++i) {
        uint32_t cur = s.proc[i].node;
        s.stats[cur].nif -= 1;
        for (;;) {
            const NodeStats& C = s.stats[cur];
            const uint32_t parent = C.parent;
            if (parent == NIL) break;
            NodeStats& P = s.stats[parent];
            P.nif -= 1;
            P.e[0][C.po[0]].nif -= 1;
            P.e[1][C.po[1]].nif -= 1;
            cur = parent;
        }
    }
    cancel_collisions(s);
    s.n_proc = 0;
    s.n_coll = 0;
    s.batch_active = 0;
}

// search.rs:400-429 apply_dirichlet_noise (shape = concentration / n >= 1 path of rand_distr Gamma)
AR_HD bool dirichlet_noise(NodeStats& nd, int pl, float epsilon, float concentration, Rng& rng, const ZigTables* zt) {
    const int n = nd.n[pl];
    if (n <= 1) return true;
    const double alpha = (double)(concentration / (float)n);
    if (!(alpha > 0.0)) return true;
    if (alpha <= 1.0) return false;
    float noise[5];
    float total = 0.0f;
    for (int i = 0; i < n; ++i) {
        noise[i] = (float)rng_gamma(rng, alpha, zt);
        total += noise[i];
    }
    if (total < 1.17549435e-38f) return true;
    for (int i = 0; i < n; ++i)
        nd.e[pl][i].prior = nd.e[pl][i].prior * (1.0f - epsilon) + epsilon * noise[i] / total;
    return true;
}

// ---- one simulate_batch, split at the evaluator boundary (search.rs:961-1073) ----------------
// gather: returns false when the arena cannot take a full batch (slot stalls untouched)
template <int NW>
AR_HD bool gather_batch(Slot<NW>& s, const GatherCtx& cx, const SearchCfg& cfg) {
    const uint32_t batch = s.remaining < cfg.batch_size ? s.remaining : cfg.batch_size;
    if (s.hi + batch > s.cap) {
        s.status = SLOT_STALL;
        s.pending_root = s.root;
        s.need_nodes = s.node_count + cfg.n_sims + 2 * cfg.batch_size;
        return false;
    }
    long long left = (long long)(int32_t)collisions_left(s.node_count, cfg);
    s.n_proc = 0;
    s.n_coll = 0;
    s.b_nn = 0;
    s.b_term = 0;
    s.b_coll = 0;
    while (s.n_proc < batch && left > 0) {
        uint32_t budget = (uint32_t)left;
        if (batch - s.n_proc < budget) budget = batch - s.n_proc;
        const uint32_t mv = pick_nodes(s, cx, cfg, budget);
        s.b_coll += mv;
        left -= (long long)mv;
    }
    s.batch_active = 1;
    return true;
}

// backup: `ev` holds b_nn results in gather order. Returns true when the search is complete.
template <int NW>
AR_HD bool backup_batch(Slot<NW>& s, const SearchCfg& cfg, const EvalOut* ev, const ZigTables* zt) {
    uint32_t j = 0;
    for (uint32_t i = 0; i < s.n_proc; ++i) {
        const uint32_t node = s.proc[i].node;
        if (s.proc[i].kind == PROC_EVAL) {
            const EvalOut& o = ev[j++];
            NodeStats& nd = s.stats[node];
            set_prior(nd, 0, o.p1);
            set_prior(nd, 1, o.p2);
            if (node == s.root && cfg.noise_epsilon > 0.0f) {
                const bool ok1 = dirichlet_noise(nd, 0, cfg.noise_epsilon, cfg.noise_concentration, s.rng, zt);
                const bool ok2 = dirichlet_noise(nd, 1, cfg.noise_epsilon, cfg.noise_concentration, s.rng, zt);
                if (!ok1 || !ok2) s.error = 5;
            }
            backup_path(s, node, o.v1, o.v2, 1);
        } else {
            backup_path(s, node, 0.0f, 0.0f, 1);
        }
    }
    cancel_collisions(s);
    s.s_nn += s.b_nn;
    s.s_term += s.b_term;
    s.s_coll += s.b_coll;
    uint32_t produced = s.b_nn + s.b_term;
    if (produced < 1) produced = 1;
    s.remaining = s.remaining > produced ? s.remaining - produced : 0;
    s.n_proc = 0;
    s.n_coll = 0;
    s.batch_active = 0;
    return s.remaining == 0;
}

// ---- search.rs:249-296, 1079-1177: result extraction -----------------------------------------
AR_HD void extract_player(const NodeStats& nd, int pl, float node_value, const SearchCfg& cfg, float* policy,
                          float* visit_counts, float* prior5, float& value) {
    const int n = nd.n[pl];
    const Edge* e = nd.e[pl];
    const uint32_t cv = nd.visits > 0 ? nd.visits - 1 : 0;
    for (int i = 0; i < 5; ++i) {
        policy[i] = 0.0f;
        visit_counts[i] = 0.0f;
        prior5[i] = 0.0f;
    }
    for (int i = 0; i < n; ++i) prior5[outcome_action(nd.omap[pl], i)] = e[i].prior;
    if (n == 0) {
        value = node_value;
        return;
    }
    const float fpu = fpu_of(e, n, node_value, nd.scale, cfg.fpu_reduction);
    float q[5], raw[5], qn[5], pruned[5];
    for (int i = 0; i < 5; ++i) {
        q[i] = 0.0f;
        raw[i] = 0.0f;
        qn[i] = 0.0f;
        pruned[i] = 0.0f;
    }
    for (int i = 0; i < n; ++i) {
        q[i] = e[i].visits > 0 ? e[i].q : fpu;
        raw[i] = (float)e[i].visits;
        qn[i] = q[i] / nd.scale;
    }
    if (n == 1) {
        pruned[0] = raw[0];
    } else {
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (raw[i] > raw[best]) best = i;
        const float sqrt_total = sqrtf((float)(cv > 1 ? cv : 1));
        const float puct_star = qn[best] + cfg.c_puct * e[best].prior * sqrt_total / (1.0f + raw[best]);
        for (int i = 0; i < n; ++i) {
            if (i == best || qn[i] >= puct_star) {
                pruned[i] = raw[i];
            } else {
                const float denom = puct_star - qn[i];
                if (denom <= 0.0f) {
                    pruned[i] = raw[i];
                } else {
                    float nmin = cfg.c_puct * e[i].prior * sqrt_total / denom - 1.0f;
                    if (!(nmin > 0.0f)) nmin = 0.0f;
                    pruned[i] = raw[i] < nmin ? raw[i] : nmin;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) visit_counts[outcome_action(nd.omap[pl], i)] = pruned[i];
    float sum = 0.0f;
    for (int i = 0; i < 5; ++i) sum += visit_counts[i];
    if (sum > 0.0f) {
        for (int i = 0; i < 5; ++i) policy[i] = visit_counts[i] / sum;
    } else {
        for (int i = 0; i < 5; ++i) policy[i] = prior5[i];
    }
    float vs = 0.0f;
    for (int i = 0; i < n; ++i) vs += raw[i];
    if (vs > 0.0f) {
        float dot = 0.0f;
        for (int i = 0; i < n; ++i) dot += q[i] * raw[i];
        value = dot / vs;
    } else {
        value = node_value;
    }
}

template <int NW>
AR_HD void extract_result(const Slot<NW>& s, const SearchCfg& cfg, MoveResult& r) {
    const NodeStats& R = s.stats[s.root];
    extract_player(R, 0, R.v1, cfg, r.policy[0], r.visit_counts[0], r.prior[0], r.value[0]);
    extract_player(R, 1, R.v2, cfg, r.policy[1], r.visit_counts[1], r.prior[1], r.value[1]);
    r.total_visits = R.visits;
    r.nn_evals = s.s_nn;
    r.terminals = s.s_term;
    r.collisions = s.s_coll;
}

// ---- tree reuse: tree.rs:283-302 --------------------------------------------------------------
// Copy the subtree under `src_root` of (src_stats, src_kids) to dst[dst_at ...) in DFS pre-order.
// Returns the number of nodes copied (the reference's count_subtree_nodes, tree.rs:209-226).
AR_HD uint32_t copy_subtree(const NodeStats* src_stats, const NodeKids* src_kids, uint32_t src_root,
                            NodeStats* dst_stats, NodeKids* dst_kids, uint32_t dst_at, CopyFrame* frames,
                            uint32_t max_depth, uint32_t& error) {
    uint32_t next = dst_at;
    dst_stats[next] = src_stats[src_root];
    dst_stats[next].parent = NIL;
    for (int i = 0; i < 25; ++i) dst_kids[next].c[i] = NIL;
    frames[0].old_id = src_root;
    frames[0].new_id = next;
    frames[0].slot = 0;
    next += 1;
    uint32_t depth = 1;
    while (depth > 0) {
        CopyFrame& f = frames[depth - 1];
        uint32_t child = NIL, sl = f.slot;
        while (sl < 25) {
            child = src_kids[f.old_id].c[sl];
            if (child != NIL) break;
            ++sl;
        }
        if (sl >= 25) {
            depth -= 1;
            continue;
        }
        f.slot = sl + 1;
        const uint32_t nn = next++;
        dst_stats[nn] = src_stats[child];
        dst_stats[nn].parent = f.new_id;
        for (int i = 0; i < 25; ++i) dst_kids[nn].c[i] = NIL;
        dst_kids[f.new_id].c[sl] = nn;
        if (depth >= max_depth) {
            error = 6;
            continue;
        }
        frames[depth].old_id = child;
        frames[depth].new_id = nn;
        frames[depth].slot = 0;
        depth += 1;
    }
    return next - dst_at;
}

// After the real move (a1, a2): keep the matching child's subtree or start a fresh root.
template <int NW>
AR_HD void advance_or_reinit(Slot<NW>& s, const uint8_t* cost, const SearchCfg& cfg, uint32_t a1, uint32_t a2) {
    const NodeStats& R = s.stats[s.root];
    const uint32_t i = action_outcome(R.omap[0], a1), j = action_outcome(R.omap[1], a2);
    const uint32_t child = s.kids[s.root].c[i * 5 + j];
    if (child == NIL) {
        make_root(s, cost, 0);  // reinit (tree.rs:298-302)
        return;
    }
    // upper bound on the kept subtree: every node was created by one visit of that child
    uint32_t bound = s.stats[child].visits + 1;
    if (bound > s.node_count) bound = s.node_count;
    uint32_t at;
    if (s.lo >= bound) at = 0;
    else if (s.cap - s.hi >= bound) at = s.hi;
    else {
        s.status = SLOT_STALL;
        s.pending_root = child;
        s.need_nodes = bound + cfg.n_sims + 2 * cfg.batch_size;
        return;
    }
    const uint32_t cnt = copy_subtree(s.stats, s.kids, child, s.stats, s.kids, at, s.frames, s.max_depth, s.error);
    s.root = at;
    s.lo = at;
    s.hi = at + cnt;
    s.node_count = cnt;
}

// The end of one self-play turn (selfplay.rs:538-565): extract, sample both actions, record, move,
// reuse the tree. Returns true when the game is over.
template <int NW>
AR_HD bool finish_move(Slot<NW>& s, const uint8_t* cost, const SearchCfg& cfg) {
    extract_result(s, cfg, s.last);
    if (s.single_search) {
        s.status = SLOT_DONE;
        return true;
    }
    s.t_sims += s.last.total_visits;
    s.t_nn += s.last.nn_evals;
    s.t_term += s.last.terminals;
    s.t_coll += s.last.collisions;
    int a1 = rng_weighted5(s.rng, s.last.policy[0]);
    if (a1 < 0) a1 = 4;
    int a2 = rng_weighted5(s.rng, s.last.policy[1]);
    if (a2 < 0) a2 = 4;
    if (s.n_pos < s.board.max_turns) {
        PosRec<NW>& p = s.pos[s.n_pos];
        p.st = s.st;
        p.res = s.last;
        p.a1 = (uint8_t)a1;
        p.a2 = (uint8_t)a2;
    } else {
        s.error = 7;
    }
    s.n_pos += 1;
    float r1, r2;
    st_step(s.board, cost, s.st, (uint32_t)a1, (uint32_t)a2, r1, r2);
    if (st_over(s.board, s.st)) {
        s.status = SLOT_DONE;
        return true;
    }
    s.remaining = cfg.n_sims;
    s.s_nn = 0;
    s.s_term = 0;
    s.s_coll = 0;
    advance_or_reinit(s, cost, cfg, (uint32_t)a1, (uint32_t)a2);
    return false;
}

// Start a game in a slot (selfplay.rs:526-535). The caller has filled board, st, rng, arena and
// scratch pointers.
template <int NW>
AR_HD void start_game(Slot<NW>& s, const uint8_t* cost, const SearchCfg& cfg) {
    s.n_pos = 0;
    s.t_sims = s.t_nn = s.t_term = s.t_coll = 0;
    s.nv_gather = s.nv_backup = s.new_nodes = 0;
    s.s_nn = s.s_term = s.s_coll = 0;
    s.n_proc = s.n_coll = 0;
    s.b_nn = s.b_term = s.b_coll = 0;
    s.batch_active = 0;
    s.error = 0;
    s.remaining = cfg.n_sims;
    make_root(s, cost, 0);
    if (!s.single_search && st_over(s.board, s.st)) s.status = SLOT_DONE;  // while !check_game_over()
    else s.status = SLOT_ACTIVE;
}

}  // namespace ar

// ---- arena growth ------------------------------------------------------------------------------
// A stalled slot (gather_batch / advance_or_reinit found no room) is moved by the host runtime
// into a bigger arena: the subtree it asked to keep is copied to the front of the new arena.
namespace ar {
template <int NW>
AR_HD void migrate_slot(Slot<NW>& s, NodeStats* new_stats, NodeKids* new_kids, uint32_t new_cap) {
    const uint32_t cnt =
        copy_subtree(s.stats, s.kids, s.pending_root, new_stats, new_kids, 0, s.frames, s.max_depth, s.error);
    s.stats = new_stats;
    s.kids = new_kids;
    s.cap = new_cap;
    s.root = 0;
    s.lo = 0;
    s.hi = cnt;
    s.node_count = cnt;
    s.status = SLOT_ACTIVE;
}
}  // namespace ar
