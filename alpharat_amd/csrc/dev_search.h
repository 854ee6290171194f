// Decoupled-PUCT tree search on device-resident node arenas: the MI355X-side counterpart of
// crates/alpharat-mcts/src/{node,tree,search}.rs and the per-move part of
// crates/alpharat-sampling/src/selfplay.rs:515-598.
//
// Layout (one arena per game, all in HBM):
//   NodeStats  320 B / node = twenty 16-B groups in five consecutive 64-byte lines: ten edge records
//              {prior, q, visits, in_flight} ([player][outcome]), three header groups, and the 25-slot
//              child table indexed by (p1_outcome*5 + p2_outcome) (replaces the reference's linked
//              list + find_child walk, tree.rs:52-63). Every access below is a whole 16-B group, so a
//              lane moves a node with dwordx4 loads/stores; a visit touches one contiguous record
//              (rounds 1-2 kept the child table in a second array: six lines in two places per visit).
// Node ids are arena indices; the live tree is [0, hi) and new nodes bump `hi`, so a parent's id is
// always smaller than its children's. Moving the root (tree reuse, tree.rs:283-295) is an in-place
// sliding compaction in id order done by a whole wavefront (advance_tree_*), which also yields the
// recounted node_count the collision budget depends on.
//
// One lane walks one game. The reference's nested loops (allocation loop inside the DFS inside the
// gather loop, then the backup walks) are flattened into one loop over small states
// (gather_machine / backup_machine) so the 64 lanes of a wavefront, each at a different place of a
// different tree, execute the same instructions most of the time. The per-node allocation state
// (scores, started counts, the 25 visit allocations) lives in registers; it is spilled to the
// per-game level stack only while a deeper level is being processed.
//
// All arithmetic keeps the reference's f32 operation order; build with -ffp-contract=off.
#pragma once
#include "dev_engine.h"

namespace ar {

static const uint32_t NIL = 0xFFFFFFFFu;

struct alignas(16) Edge {
    float prior;
    float q;
    uint32_t visits;
    uint32_t nif;  // n_in_flight (virtual loss)
};
struct alignas(16) NodeH0 {
    float v1, v2;     // Welford means
    uint32_t visits;  // total_visits
    uint32_t nif;     // n_in_flight
};
struct alignas(16) NodeH1 {
    float scale;      // value_scale = max(remaining_cheese, 1) at creation
    float r1, r2;     // edge rewards from the parent
    uint32_t parent;  // NIL for the root
};
struct alignas(16) NodeH2 {
    uint32_t omap[2];  // per player: outcome->action 3 bits each (bits 0..14), action->outcome (bits 15..29)
    uint32_t meta;     // n_outcomes p1 | p2 << 8 | parent_outcome p1 << 16 | p2 << 24
    uint32_t terminal;
};
struct alignas(64) NodeStats {
    Edge e[2][5];
    NodeH0 h0;
    NodeH1 h1;
    NodeH2 h2;
    uint32_t c[25];  // child ids, NIL = no child yet
    uint32_t pad[3];
};
static_assert(sizeof(NodeStats) == 320, "a node record is twenty 16-byte groups");
enum { NODE_GROUPS = 20, NODE_KID_GROUP = 13 };  // groups per record; first group of the child table

AR_HD uint32_t meta_n(uint32_t meta, int pl) { return (meta >> (8 * pl)) & 0xffu; }
AR_HD uint32_t meta_po(uint32_t meta, int pl) { return (meta >> (16 + 8 * pl)) & 0xffu; }

struct SearchCfg {
    float c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration;
    uint32_t coll_min, coll_max, coll_start, coll_end;
    float coll_power;
    uint32_t n_sims, batch_size;
    uint32_t alloc_per_round;  // scheduling only: allocation-loop steps one gather round runs (gather_round)
};

enum { SLOT_EMPTY = 0, SLOT_ACTIVE = 1, SLOT_DONE = 2, SLOT_STALL = 3, SLOT_FAILED = 4, SLOT_ADVANCE = 5 };
enum { PROC_TERMINAL = 0, PROC_EVAL = 1 };

struct ProcEntry {
    uint32_t node;
    // PROC_TERMINAL or PROC_EVAL | (index of the leaf's evaluation in this batch's requests) << 8 | order key << 16: the
    // backup takes a batch's entries by increasing key (the depth-first gathers write them in that order: key = index;
    // the work-queue gather writes them as their leaves are reached: key = pick number, visit slot)
    uint32_t kind;
};
AR_HD uint32_t proc_kind(uint32_t k) { return k & 0xffu; }
AR_HD uint32_t proc_eval_index(uint32_t k) { return (k >> 8) & 0xffu; }
AR_HD uint32_t proc_key(uint32_t k) { return k >> 16; }
struct CollEntry {
    uint32_t node;
    uint32_t mv;
};
struct EvalOut {
    float p1[5], p2[5];
    float v1, v2;
};

// a level of the gather DFS while a deeper level is being processed (search.rs:561-569 GatherLevel)
template <int NW>
struct alignas(16) Level {
    uint32_t node;
    uint32_t mask;      // child slots with allocated visits not yet processed
    uint32_t omap[2];   // node's outcome->action maps
    uint32_t vtp[13];   // 25 x u16 allocated visits, two per word
    uint32_t pad[3];
    State<NW> saved;    // position at this node
};

struct MoveResult {  // search.rs:304-325
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

// (128-byte aligned: neighbouring slots can be owned by kernels on different streams and XCDs)
template <int NW>
struct alignas(128) Slot {
    Board board;
    State<NW> st;
    Rng rng;
    uint32_t game_index;
    uint32_t status;
    // arena: byte offsets from the arena base the kernels receive as an argument (so the compiler
    // can prove the accesses are to global memory)
    long long stats_off, fwd_off;
    uint32_t cap, hi, root, node_count;
    uint32_t pending_root;  // SLOT_ADVANCE: child to keep (NIL = fresh root)
    uint32_t need_nodes;    // capacity a stalled slot asks for
    uint32_t release_grown; // 1: the slot went back to its pool share; the host may free the grown arena
    uint32_t pool_blk;      // 0: first arena or host-grown; else ((class + 1) << 24) | block index in the overflow pool
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
    MoveResult last;  // result of the last finished search
    uint32_t error;   // non-zero: internal capacity violation (bug guard)
    uint32_t gather_pending;  // 1: a gather was cut off at the round limit (lane kernel: its state is in Mem::glane;
                              // work-queue kernel: between two pick_nodes_to_extend calls, counters below)
    uint32_t g_rounds;        // rounds the game's last complete gather took (scheduling hint)
    int32_t g_left;           // parked work-queue gather: collision budget left, number of the last pick
    uint32_t g_pick;
};

// resolved addresses of one game's memory (built per kernel from kernel arguments + slot offsets)
template <int NW>
struct Mem {
    NodeStats* stats;
    uint32_t* fwd;          // [cap] new ids during the compaction
    ProcEntry* proc;        // [batch_size]
    CollEntry* coll;        // [coll_cap]
    Level<NW>* levels;      // [max_depth]
    EvalOut* ev_local;      // [batch_size] evaluator outputs when the evaluator runs inline
    State<NW>* leaf_local;  // [batch_size] leaf positions for evaluators outside the walk
    PosRec<NW>* pos;        // [max_turns]
    void* glane;            // GatherLane<NW> of a gather cut off at the round limit
    const uint8_t* cost;    // this game's maze
    uint32_t coll_cap, max_depth;
};

// A leaf waiting for the device-wide evaluator (replaces MuxBackend's request queue, mux.rs:170-289)
template <int NW>
struct LeafReq {
    State<NW> st;
    uint32_t slot;
    uint32_t pad;
};

// ---- node.rs:251-283 compute_outcomes, packed ------------------------------------------------
AR_HD void pack_outcomes(uint32_t eff, uint32_t& omap, uint32_t& n_out) {
    uint32_t present = 0;
    for (int a = 0; a < 5; ++a) present |= 1u << ((eff >> (3 * a)) & 7u);
    uint32_t m = 0, cnt = 0, rank_packed = 0;
    for (uint32_t act = 0; act < 5; ++act) {
        rank_packed |= cnt << (3 * act);
        if (present & (1u << act)) {
            m |= act << (3 * cnt);
            ++cnt;
        }
    }
    for (int a = 0; a < 5; ++a) m |= ((rank_packed >> (3 * ((eff >> (3 * a)) & 7u))) & 7u) << (15 + 3 * a);
    omap = m;
    n_out = cnt;
}
AR_HD uint32_t outcome_action(uint32_t omap, uint32_t idx) { return (omap >> (3 * idx)) & 7u; }
AR_HD uint32_t action_outcome(uint32_t omap, uint32_t act) { return (omap >> (15 + 3 * act)) & 7u; }

// tree.rs:69-84 smart_uniform_prior from a packed effective-action map
AR_HD void uniform_prior(uint32_t eff, float* p5) {
    uint32_t present = 0;
    for (int a = 0; a < 5; ++a) present |= 1u << ((eff >> (3 * a)) & 7u);
    uint32_t cnt = 0;
    for (int a = 0; a < 5; ++a) cnt += (present >> a) & 1u;
    const float p = 1.0f / (float)cnt;
    for (int a = 0; a < 5; ++a) p5[a] = (present >> a) & 1u ? p : 0.0f;
}

// node.rs:173-179 set_prior: scatter-add the 5 action priors into outcome slots in action order
AR_HD void reduce_prior(uint32_t omap, const float* p5, float* out5) {
    for (int i = 0; i < 5; ++i) out5[i] = 0.0f;
    for (uint32_t a = 0; a < 5; ++a) {
        const uint32_t o = action_outcome(omap, a);
        for (uint32_t i = 0; i < 5; ++i)
            if (i == o) out5[i] += p5[a];
    }
}

// shell node (tree.rs:107-148 extend_node + :199 edge rewards): all stores, no loads
AR_HD void init_shell(NodeStats& nd, uint32_t eff1, uint32_t eff2, uint16_t remaining,
                      uint32_t parent, uint32_t o1, uint32_t o2, float r1, float r2) {
    Edge z;
    z.prior = 0.0f;
    z.q = 0.0f;
    z.visits = 0;
    z.nif = 0;
    for (int pl = 0; pl < 2; ++pl)
        for (int i = 0; i < 5; ++i) nd.e[pl][i] = z;
    NodeH0 h0;
    h0.v1 = 0.0f;
    h0.v2 = 0.0f;
    h0.visits = 0;
    h0.nif = 0;
    nd.h0 = h0;
    NodeH1 h1;
    h1.scale = (float)(remaining > 1 ? remaining : 1);
    h1.r1 = r1;
    h1.r2 = r2;
    h1.parent = parent;
    nd.h1 = h1;
    NodeH2 h2;
    uint32_t n1, n2;
    pack_outcomes(eff1, h2.omap[0], n1);
    pack_outcomes(eff2, h2.omap[1], n2);
    h2.meta = n1 | (n2 << 8) | (o1 << 16) | (o2 << 24);
    h2.terminal = 0;
    nd.h2 = h2;
    for (int i = 0; i < 25; ++i) nd.c[i] = NIL;
    nd.pad[0] = nd.pad[1] = nd.pad[2] = NIL;
}

// tree.rs:351-365 alloc_root at arena index 0 (also MCTSTree::reinit, tree.rs:298-302)
template <int NW>
AR_HD void make_root(Slot<NW>& s, const Mem<NW>& m) {
    const uint32_t e1 = eff_actions(m.cost, s.st.p1, s.st.m1), e2 = eff_actions(m.cost, s.st.p2, s.st.m2);
    NodeStats& nd = m.stats[0];
    init_shell(nd, e1, e2, s.st.remaining, NIL, 0, 0, 0.0f, 0.0f);
    uint32_t om[2], nn[2];
    pack_outcomes(e1, om[0], nn[0]);
    pack_outcomes(e2, om[1], nn[1]);
    float p5[5], red[5];
    for (int pl = 0; pl < 2; ++pl) {
        uniform_prior(pl == 0 ? e1 : e2, p5);
        reduce_prior(om[pl], p5, red);
        for (int i = 0; i < 5; ++i) {
            Edge e;
            e.prior = red[i];
            e.q = 0.0f;
            e.visits = 0;
            e.nif = 0;
            nd.e[pl][i] = e;
        }
    }
    s.root = 0;
    s.hi = 1;
    s.node_count = 1;
}

// search.rs:437-450
AR_HD uint32_t collisions_left(uint32_t node_count, const SearchCfg& c) {
    if (node_count >= c.coll_end) return c.coll_max;
    if (node_count <= c.coll_start) return c.coll_min;
    const float ratio = (float)(node_count - c.coll_start) / (float)(c.coll_end - c.coll_start);
    // (x^1 = x in every libm; spelled out because the device's pow is a long routine that every game would run)
    const float scaled = (float)c.coll_min + ((float)c.coll_max - (float)c.coll_min) * (c.coll_power == 1.0f ? ratio : powf(ratio, c.coll_power));
    const float r = roundf(scaled);
    uint32_t v = r <= 0.0f ? 0u : (r >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)r);
    if (v < c.coll_min) v = c.coll_min;
    if (v > c.coll_max) v = c.coll_max;
    return v;
}

// How a gathered leaf is handed to the evaluator.
//  EVAL_UNIFORM: SmartUniform computed on the spot into ev_local (backend.rs:92-103)
//  EVAL_STORE:   position kept in leaf_local; the step kernel then appends the batch's leaves to the
//                device-wide leaf queue (network evaluators) or the host reads them back (predict_fn)
enum { EVAL_UNIFORM = 0, EVAL_STORE = 1 };

// ---- per-player allocation state of the node being expanded (registers) ----------------------
// Only the started count of the chosen outcome changes between two allocation steps
// (search.rs:775-798), so q_norm, the exploration numerator c_puct*prior*sqrt_total and the forced
// flag are computed once per node; each step recomputes one score (same expression, same bits).
struct HalfAlloc {
    float score[5], util[5], num[5];
    uint32_t ns[5], add[5], nif0[5];
    uint32_t forced;  // bit i: forced-playout score (search.rs:493-498)
    uint32_t n;
};

AR_HD float pick5(const float* a, uint32_t i) {
    return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : i == 3 ? a[3] : a[4];
}
AR_HD uint32_t pick5u(const uint32_t* a, uint32_t i) {
    return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : i == 3 ? a[3] : a[4];
}

// node entry: search.rs:478-498 (fpu, sqrt_total, per-outcome score) from the loaded edges
AR_HD void half_init(HalfAlloc& h, const Edge* e, uint32_t n, float node_value, float scale, uint32_t cv,
                     const SearchCfg& cfg, bool is_root) {
    h.n = n;
    h.forced = 0;
    float mass = 0.0f;
    for (uint32_t i = 0; i < 5; ++i)
        if (i < n && e[i].visits > 0) mass += e[i].prior;
    const float fpu = node_value - cfg.fpu_reduction * scale * sqrtf(mass);
    const float sqrt_total = sqrtf((float)(cv > 1 ? cv : 1));
    for (uint32_t i = 0; i < 5; ++i) {
        const bool live = i < n;
        const float q = e[i].visits > 0 ? e[i].q : fpu;
        h.util[i] = q / scale;
        h.num[i] = cfg.c_puct * e[i].prior * sqrt_total;
        h.ns[i] = live ? e[i].visits + e[i].nif : 0;
        h.nif0[i] = e[i].nif;
        h.add[i] = 0;
        float sc = h.util[i] + h.num[i] / (1.0f + (float)h.ns[i]);
        if (live && is_root && cfg.force_k > 0.0f && e[i].prior > 0.0f) {
            const float threshold = sqrtf(cfg.force_k * e[i].prior * (float)cv);
            if ((float)e[i].visits < threshold) {
                sc = 1e20f;
                h.forced |= 1u << i;
            }
        }
        h.score[i] = sc;
    }
}

// search.rs:463-554 estimated_visits_to_change_best_half on the cached scores
AR_HD void half_best(const HalfAlloc& h, Rng& rng, uint32_t& best_out, uint32_t& vtc_out) {
    const uint32_t n = h.n;
    if (n <= 1) {
        best_out = 0;
        vtc_out = 0xFFFFFFFFu;
        return;
    }
    const float NEG_INF = -__builtin_inff();
    uint32_t best = 0;
    float best_score = NEG_INF, best_util = NEG_INF, second = NEG_INF;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        if (i < n) {
            const float sc = h.score[i];
            if (sc > best_score) {
                second = best_score;
                best_score = sc;
                best = i;
                best_util = h.util[i];
            } else if (sc > second) {
                second = sc;
            }
        }
    }
    uint32_t ties = 1;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        if (i < n && i != best && fabsf(h.score[i] - best_score) < 1e-12f) {
            ties += 1;
            if (rng_below(rng, ties) == 0) {
                best = i;
                best_util = h.util[i];
            }
        }
    }
    best_out = best;
    vtc_out = 0xFFFFFFFFu;
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

AR_HD void half_take(HalfAlloc& h, uint32_t b, uint32_t k) {
    // one division for the chosen outcome, then scatter with selects (no per-index divisions)
    const uint32_t nsb = pick5u(h.ns, b) + k;
    const float sc = pick5(h.util, b) + pick5(h.num, b) / (1.0f + (float)nsb);
    const bool keep_forced = (h.forced >> b) & 1u;
    for (uint32_t i = 0; i < 5; ++i) {
        const bool hit = i == b;
        h.ns[i] = hit ? nsb : h.ns[i];
        h.add[i] += hit ? k : 0u;
        h.score[i] = (hit && !keep_forced) ? sc : h.score[i];
    }
}

// (selects over all 13 words, never a conditional access: a conditional one is turned into a
// dynamically indexed access by the optimizer, which forces the whole lane state into scratch memory)
AR_HD uint32_t vtp_get(const uint32_t* w, uint32_t idx) {
    uint32_t word = 0;
    const uint32_t x = idx >> 1;
#pragma unroll
    for (uint32_t j = 0; j < 13; ++j) word |= (j == x) ? w[j] : 0u;
    return (word >> (16 * (idx & 1u))) & 0xffffu;
}
AR_HD void vtp_add(uint32_t* w, uint32_t idx, uint32_t k) {
    const uint32_t x = idx >> 1, v = k << (16 * (idx & 1u));
#pragma unroll
    for (uint32_t j = 0; j < 13; ++j) w[j] += (j == x) ? v : 0u;
}

AR_HD int lowest_bit(uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)m) - 1;
#else
    return __builtin_ctz(m);
#endif
}

// ---- gather: search.rs:961-999 (outer loop) + 576-738 (pick_nodes_to_extend) + 742-817 ---------
enum {
    G_PICK = 0,   // start one pick_nodes_to_extend call (or finish the gather)
    G_ALLOC = 1,  // (unused: the allocation loop runs inside G_ENTER)
    G_CHILD = 2,  // process the next child slot that received visits (or pop)
    G_ENTER = 3,  // load a node for expansion and set up its allocation state
    G_DONE = 4
};

// Which state the wavefront runs next: the one most lanes are waiting in. Lanes in other states sit
// the round out, so every executed instruction serves as many lanes as possible; a lane's own
// sequence of states -- and therefore its result -- does not depend on the election.
#if defined(AR_STATS) && defined(__HIPCC__)
__device__ unsigned long long g_round_stats[32];  // [machine*16 + state*2 + {rounds, lanes}] , [30] alive, [31] rounds
// [0..63] lanes by rounds/4 of one gather, [64..127] wavefronts by (max rounds)/4, [128..135] lane-rounds by
// path: pop, pick, new leaf, leaf claim, interior, allocation steps, rounds in allocation only, done-at-pick
__device__ unsigned long long g_gather_hist[136];
// wavefront clocks of the gather loop by (max rounds)/4: [b] = sum of clocks, [64 + b] = wavefronts
__device__ unsigned long long g_gather_clk[128];
#if defined(AR_STATS_PATHS)  // per-lane path counters: heavy (contended atomics), distorts timings
#define AR_COUNT(i) atomicAdd(&g_gather_hist[i], 1ULL)
#else
#define AR_COUNT(i) ((void)0)
#endif
#else
#define AR_COUNT(i) ((void)0)
#endif
AR_HD uint32_t elect_state(uint32_t state, uint32_t n_states, uint32_t done_state, uint32_t machine = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t best = done_state;
    int best_count = 0;
    for (uint32_t st = 0; st < n_states; ++st) {
        const int c = __popcll(__ballot(state == st));
        if (c > best_count) {
            best_count = c;
            best = st;
        }
    }
#if defined(AR_STATS_PATHS)
    if (best != done_state) {
        const int alive = __popcll(__ballot(state != done_state));
        if (__popcll(__ballot(1) & ((1ULL << (threadIdx.x & 63)) - 1ULL)) == 0) {
            atomicAdd(&g_round_stats[machine * 12 + best * 2], 1ULL);
            atomicAdd(&g_round_stats[machine * 12 + best * 2 + 1], (unsigned long long)best_count);
            atomicAdd(&g_round_stats[30], (unsigned long long)alive);
            atomicAdd(&g_round_stats[31], 1ULL);
        }
    }
#endif
    (void)machine;
    return best;
#else
    (void)n_states;
    (void)done_state;
    return state;
#endif
}

template <int NW>
AR_HD void emit_proc(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, int eval_mode, uint32_t node, uint32_t kind,
                     const State<NW>& leaf) {
    if (s.n_proc >= cfg.batch_size) {
        s.error = 1;
        return;
    }
    const uint32_t i = s.n_proc++;
    ProcEntry pe;
    pe.node = node;
    pe.kind = (kind == PROC_EVAL ? kind | (s.b_nn << 8) : kind) | (i << 16);
    m.proc[i] = pe;
    if (kind == PROC_EVAL) {
        const uint32_t j = s.b_nn++;
        if (eval_mode == EVAL_UNIFORM) {
            EvalOut o;
            uniform_prior(eff_actions(m.cost, leaf.p1, leaf.m1), o.p1);
            uniform_prior(eff_actions(m.cost, leaf.p2, leaf.m2), o.p2);
            o.v1 = 0.0f;
            o.v2 = 0.0f;
            m.ev_local[j] = o;
        } else {
            m.leaf_local[j] = leaf;
        }
    } else {
        s.b_term += 1;
    }
}
template <int NW>
AR_HD void emit_coll(Slot<NW>& s, const Mem<NW>& m, uint32_t node, uint32_t mv, uint32_t& pick_mv) {
    pick_mv += mv;
    if (s.n_coll >= m.coll_cap) {
        s.error = 2;
        return;
    }
    CollEntry c;
    c.node = node;
    c.mv = mv;
    m.coll[s.n_coll] = c;
    s.n_coll += 1;
}

// ---- per-lane gather state (registers) ---------------------------------------------------------
template <int NW>
struct GatherLane {
    uint32_t state;
    uint32_t batch;        // productive entries wanted this batch
    long long left;        // collision budget left (search.rs:970)
    uint32_t depth;        // levels kept on the stack below the current node
    uint32_t node;         // current node
    uint32_t mask;         // child slots of the current node still to process
    uint32_t omap0, omap1;
    uint32_t vtp[13];      // 25 x u16 visits allocated to the child slots
    uint32_t pick_mv;      // collision multivisits of the running pick_nodes_to_extend call
    uint32_t enter, enter_budget;
    bool have_pick, enter_root;
    int eval_mode;
    State<NW> work;        // position at the current node
    uint32_t alloc_left;   // visits of the current node still to allocate (0: not in an allocation)
    HalfAlloc h1, h2;      // allocation state of the current node
    uint32_t kid[25];      // child table of the current node, loaded together with its record
};

// Starts one simulate_batch. Returns false when the arena cannot take a full batch (the slot
// stalls untouched and the host moves it to a bigger arena).
template <int NW>
AR_HD bool gather_begin(GatherLane<NW>& g, Slot<NW>& s, const SearchCfg& cfg, int eval_mode) {
    g.state = G_DONE;
    g.batch = s.remaining < cfg.batch_size ? s.remaining : cfg.batch_size;
    if (s.hi + g.batch > s.cap) {
        s.status = SLOT_STALL;
        s.need_nodes = s.hi + cfg.n_sims + 2 * cfg.batch_size;
        return false;
    }
    g.left = (long long)(int32_t)collisions_left(s.node_count, cfg);
    s.n_proc = 0;
    s.n_coll = 0;
    s.b_nn = 0;
    s.b_term = 0;
    s.b_coll = 0;
    g.state = G_PICK;
    g.depth = 0;
    g.node = 0;
    g.mask = 0;
    g.omap0 = g.omap1 = 0;
    for (int j = 0; j < 13; ++j) g.vtp[j] = 0;
    g.pick_mv = 0;
    g.enter = NIL;
    g.enter_budget = 0;
    g.have_pick = false;
    g.enter_root = false;
    g.eval_mode = eval_mode;
    g.work = s.st;
    g.alloc_left = 0;
    for (int j = 0; j < 25; ++j) g.kid[j] = NIL;
    return true;
}

// One round of a lane's gather. Every lane that is not done runs the same two-part sequence, so the
// wavefront does not split by state:
//   D. (lanes not in the middle of an allocation) decide: pop a finished level (rare: only after a
//      split allocation), start the next pick_nodes_to_extend call at the root, or take the next child
//      slot of the current node (step the position, one load of the child id; a missing child is
//      created and becomes a leaf right here, stores only). Then load the whole record of the node to
//      look at -- the root or the existing child -- in one round trip and classify it: unvisited /
//      terminal -> a batch entry (or collision); visited interior -> add the virtual loss, keep the
//      level if siblings still wait, set up the allocation from the registers just loaded. The batch
//      entry / collision record of whichever branch produced one is written once, after the branches.
//   A. (lanes with visits still to allocate, including those that just set one up) at most
//      cfg.alloc_per_round steps of the allocation loop (search.rs:775-798); the virtual-loss
//      write-back when the last visit is placed.
// The allocation loop of a node takes 1..batch steps (the root takes the most), so running it to the
// end inside one round made every lane of the wavefront wait for the longest one; with the cap a
// long allocation spreads over a few rounds and only that lane waits.
// A descent costs one round (<= two dependent memory trips) per tree level.
enum { PROC_NONE = 0xFFu };
template <int NW>
AR_HD void gather_round(GatherLane<NW>& g, Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg) {
    if (g.state == G_DONE) return;
#if defined(AR_STATS_PATHS) && defined(__HIPCC__)
    if (g.alloc_left != 0) AR_COUNT(134);
#endif
    if (g.alloc_left == 0) {
        const State<NW> before = g.work;
        if (g.mask == 0 && g.depth > 0) {
            // level exhausted: backtrack (search.rs:728-734)
            AR_COUNT(128);
            g.depth -= 1;
            const Level<NW>& L = m.levels[g.depth];
            g.node = L.node;
            g.mask = L.mask;
            g.omap0 = L.omap[0];
            g.omap1 = L.omap[1];
            for (int j = 0; j < 13; ++j) g.vtp[j] = L.vtp[j];
            g.work = L.saved;
            const NodeStats& K = m.stats[L.node];
            for (int j = 0; j < 25; ++j) g.kid[j] = K.c[j];
        } else {
            uint32_t rec = NIL;      // node whose record is inspected
            uint32_t visits_in = 0;  // visits routed to it (pick budget or k)
            bool from_pick = false;
            uint32_t emit_node = NIL, emit_kind = PROC_NONE, coll_mv = 0;
            bool restore = false;
            if (g.mask == 0) {
                // search.rs:981-999 outer gather loop around pick_nodes_to_extend
                if (g.have_pick) {
                    s.b_coll += g.pick_mv;
                    g.left -= (long long)g.pick_mv;
                    g.have_pick = false;
                }
                if (!(s.n_proc < g.batch && g.left > 0)) {
                    AR_COUNT(135);
                    g.state = G_DONE;
                    s.batch_active = 1;
                    return;
                }
                AR_COUNT(129);
                uint32_t budget = (uint32_t)g.left;
                if (g.batch - s.n_proc < budget) budget = g.batch - s.n_proc;
                g.pick_mv = 0;
                g.have_pick = true;
                g.work = s.st;
                rec = s.root;
                visits_in = budget;
                from_pick = true;
            } else {
                const uint32_t idx = (uint32_t)lowest_bit(g.mask);
                g.mask &= g.mask - 1;
                const uint32_t k = vtp_get(g.vtp, idx);
                const uint32_t o1 = idx / 5, o2 = idx % 5;
                float r1, r2;
                st_step(s.board, m.cost, g.work, outcome_action(g.omap0, o1), outcome_action(g.omap1, o2), r1, r2);
                // the child id comes from the table loaded with the parent's record: no dependent load here
                uint32_t child = 0;
#pragma unroll
                for (uint32_t j = 0; j < 25; ++j) child |= (j == idx) ? g.kid[j] : 0u;
                if (child == NIL) {
                    // new leaf: shell creation + claim are stores only (tree.rs:107-148, search.rs:675-701)
                    AR_COUNT(130);
                    if (s.hi >= s.cap) {  // excluded by the capacity check in gather_begin
                        s.error = 3;
                    } else {
                        const uint32_t nid = s.hi++;
                        init_shell(m.stats[nid], eff_actions(m.cost, g.work.p1, g.work.m1),
                                   eff_actions(m.cost, g.work.p2, g.work.m2), g.work.remaining, g.node, o1, o2, r1, r2);
                        m.stats[g.node].c[idx] = nid;
                        s.node_count += 1;
                        s.new_nodes += 1;
                        m.stats[nid].h0.nif = 1;  // try_start_score_update on a fresh node
                        const bool over = st_over(s.board, g.work);
                        if (over) m.stats[nid].h2.terminal = 1;
                        emit_node = nid;
                        emit_kind = over ? PROC_TERMINAL : PROC_EVAL;
                        coll_mv = k > 1 ? k - 1 : 0;
                    }
                    restore = true;
                } else {
                    rec = child;
                    visits_in = k;
                }
            }
            if (rec != NIL) {
                // the record of `rec`, one round trip
                const NodeStats& N = m.stats[rec];
                Edge e1[5], e2[5];
                for (int i = 0; i < 5; ++i) {
                    e1[i] = N.e[0][i];
                    e2[i] = N.e[1][i];
                }
                const NodeH0 a = N.h0;
                const NodeH1 b = N.h1;
                const NodeH2 c = N.h2;
                // its child table rides along in the same round trip (used if the node is expanded)
                uint32_t kid_in[25];
                for (int j = 0; j < 25; ++j) kid_in[j] = N.c[j];
                if (a.visits == 0 || c.terminal != 0) {
                    // leaf or terminal (search.rs:591-636 for the root, :675-706 for a child)
                    AR_COUNT(131);
                    emit_node = rec;
                    if (!(a.visits == 0 && a.nif > 0)) {  // try_start_score_update
                        m.stats[rec].h0.nif = a.nif + 1;
                        const bool term = c.terminal != 0 || st_over(s.board, g.work);
                        if (term && a.visits == 0) m.stats[rec].h2.terminal = 1;
                        emit_kind = term ? PROC_TERMINAL : PROC_EVAL;
                        coll_mv = visits_in > 1 ? visits_in - 1 : 0;
                    } else {
                        coll_mv = visits_in;
                    }
                    restore = true;  // (a root pick leaves work == s.st; mask stays 0 -> next round picks again)
                } else if (!from_pick && g.depth >= m.max_depth) {
                    s.error = 4;
                    restore = true;
                } else {
                    // visited interior node: route the visits through it (search.rs:639 / :707-725)
                    AR_COUNT(132);
                    m.stats[rec].h0.nif = a.nif + visits_in;
                    if (!from_pick && g.mask != 0) {  // siblings still wait: keep the parent level for the way back
                        Level<NW>& L = m.levels[g.depth];
                        L.node = g.node;
                        L.mask = g.mask;
                        L.omap[0] = g.omap0;
                        L.omap[1] = g.omap1;
                        for (int j = 0; j < 13; ++j) L.vtp[j] = g.vtp[j];
                        L.saved = before;
                        g.depth += 1;
                    }
                    if (from_pick) g.depth = 0;
                    // set up build_gather_level (search.rs:742-817) from the registers just loaded
                    const uint32_t cv = a.visits > 0 ? a.visits - 1 : 0;
                    half_init(g.h1, e1, meta_n(c.meta, 0), a.v1, b.scale, cv, cfg, from_pick);
                    half_init(g.h2, e2, meta_n(c.meta, 1), a.v2, b.scale, cv, cfg, from_pick);
                    g.node = rec;
                    g.omap0 = c.omap[0];
                    g.omap1 = c.omap[1];
                    g.mask = 0;
                    for (int j = 0; j < 13; ++j) g.vtp[j] = 0;
                    for (int j = 0; j < 25; ++j) g.kid[j] = kid_in[j];
                    s.nv_gather += 1;
                    g.alloc_left = visits_in;
                }
            }
            if (emit_kind != PROC_NONE) emit_proc(s, m, cfg, g.eval_mode, emit_node, emit_kind, g.work);
            if (coll_mv) emit_coll(s, m, emit_node, coll_mv, g.pick_mv);
            if (restore) g.work = before;
        }
    }
    if (g.alloc_left > 0) {
        for (uint32_t it = 0; it < cfg.alloc_per_round && g.alloc_left > 0; ++it) {  // search.rs:775-798, no memory traffic
            AR_COUNT(133);
            uint32_t b1, b2, c1, c2;
            half_best(g.h1, s.rng, b1, c1);
            half_best(g.h2, s.rng, b2, c2);
            uint32_t k = g.alloc_left;
            if (c1 < k) k = c1;
            if (c2 < k) k = c2;
            if (k < 1) k = 1;
            const uint32_t flat = b1 * 5 + b2;
            vtp_add(g.vtp, flat, k);
            g.mask |= 1u << flat;
            half_take(g.h1, b1, k);
            half_take(g.h2, b2, k);
            g.alloc_left -= k;
        }
        if (g.alloc_left == 0) {
            NodeStats& W = m.stats[g.node];  // search.rs:800-814: write the virtual-loss deltas back
            for (uint32_t i = 0; i < 5; ++i) {
                if (g.h1.add[i]) W.e[0][i].nif = g.h1.nif0[i] + g.h1.add[i];
                if (g.h2.add[i]) W.e[1][i].nif = g.h2.nif0[i] + g.h2.add[i];
            }
        }
    }
}

// gather of one simulate_batch for every lane of the wavefront (split kernels: evaluator outside)
template <int NW>
AR_HD bool gather_machine(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, int eval_mode) {
    GatherLane<NW> g;
    const bool ok = gather_begin(g, s, cfg, eval_mode);
    for (;;) {
        const uint32_t run = elect_state(g.state, G_DONE, G_DONE);
        if (run == G_DONE) break;
        if (g.state == run) gather_round(g, s, m, cfg);
    }
    return ok;
}

// The same with a limit on the rounds one call may run. The time of a gather kernel is the time of
// its slowest lane, and the number of rounds a batch takes varies several-fold between games, so
// without a limit most lanes sit finished while a few complete theirs. A lane that hits the limit
// parks its state in the slot's scratch and resumes at the next call; only complete batches go on to
// the evaluator and the backup. Per-game results do not depend on where the cuts fall.
enum { GATHER_STALLED = 0, GATHER_COMPLETE = 1, GATHER_PENDING = 2 };
template <int NW>
AR_HD int gather_machine_limited(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, int eval_mode,
                                 uint32_t max_rounds) {
    GatherLane<NW> g;
    GatherLane<NW>* parked = (GatherLane<NW>*)m.glane;
    bool ok = true;
    if (s.gather_pending) g = *parked;
    else ok = gather_begin(g, s, cfg, eval_mode);
    uint32_t my_rounds = 0;
#if defined(AR_STATS) && defined(__HIPCC__)
    const unsigned long long clk0 = wall_clock64();
#endif
    for (uint32_t r = 0; r < max_rounds; ++r) {
        const uint32_t run = elect_state(g.state, G_DONE, G_DONE);
        if (run == G_DONE) break;
        if (g.state == run) {
            gather_round(g, s, m, cfg);
            my_rounds += 1;
        }
#if defined(AR_STATS) && defined(__HIPCC__)
        // checkpoints: clocks since the loop began after 1, 2, 4, 8, 16, 32, 64, 128 wavefront rounds
        if (((r + 1) & r) == 0 && r < 128 && (threadIdx.x & 63) == 0) {
            const int cp = 31 - __clz((int)(r + 1));
            atomicAdd(&g_gather_clk[112 + cp], wall_clock64() - clk0);
            atomicAdd(&g_gather_clk[120 + cp], 1ULL);
        }
#endif
    }
#if defined(AR_STATS) && defined(__HIPCC__)
    {
        uint32_t wave_max = my_rounds;
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)wave_max, off, 64);
            wave_max = o > wave_max ? o : wave_max;
        }
#if defined(AR_STATS_PATHS)
        if (ok) atomicAdd(&g_gather_hist[my_rounds / 4 < 63 ? my_rounds / 4 : 63], 1ULL);
#endif
        if ((threadIdx.x & 63) == 0) {
            const uint32_t b = wave_max / 8 < 47 ? wave_max / 8 : 47;  // clk: [0..47] sums, [64..111] counts by rounds/8
            atomicAdd(&g_gather_hist[64 + (wave_max / 4 < 63 ? wave_max / 4 : 63)], 1ULL);
            atomicAdd(&g_gather_clk[b], wall_clock64() - clk0);
            atomicAdd(&g_gather_clk[64 + b], 1ULL);
        }
    }
#endif
    if (!ok) return GATHER_STALLED;
    if (g.state != G_DONE) {
        *parked = g;
        s.gather_pending = 1;
        s.g_rounds += my_rounds;
        return GATHER_PENDING;
    }
    s.g_rounds = (s.gather_pending ? s.g_rounds : 0u) + my_rounds;
    s.gather_pending = 0;
    return GATHER_COMPLETE;
}

// ---- backup: search.rs:1027-1066 ---------------------------------------------------------------
// node.rs:444-457 on a loaded header
AR_HD void finalize_h0(NodeH0& a, float q1, float q2, uint32_t mv) {
    a.visits += mv;
    const float n = (float)a.visits, w = (float)mv;
    a.v1 += (q1 - a.v1) * w / n;
    a.v2 += (q2 - a.v2) * w / n;
    a.nif -= mv;
}
// node.rs:82-85 + revert_virtual_loss_multi
AR_HD void edge_update(Edge& e, float value, uint32_t mv) {
    e.visits += mv;
    e.q += (value - e.q) * (float)mv / (float)e.visits;
    e.nif -= mv;
}

// search.rs:400-429 apply_dirichlet_noise on the reduced priors (shape = concentration/n > 1 only)
AR_HD bool dirichlet_mix(float* prior, uint32_t n, float epsilon, float concentration, Rng& rng, const ZigTables* zt) {
    if (n <= 1) return true;
    const double alpha = (double)(concentration / (float)n);
    if (!(alpha > 0.0)) return true;
    if (alpha <= 1.0) return false;
    float noise[5];
    float total = 0.0f;
    for (uint32_t i = 0; i < 5; ++i) {
        noise[i] = 0.0f;
        if (i < n) {
            noise[i] = (float)rng_gamma(rng, alpha, zt);
            total += noise[i];
        }
    }
    if (total < 1.17549435e-38f) return true;
    for (uint32_t i = 0; i < 5; ++i)
        if (i < n) prior[i] = prior[i] * (1.0f - epsilon) + epsilon * noise[i] / total;
    return true;
}

enum { B_ENTRY = 0, B_LEVEL = 1, B_CANCEL = 2, B_CANCEL_LEVEL = 3, B_DONE = 4 };

struct BackupLane {
    uint32_t state;
    uint32_t i, j;  // proc / collision index, eval index
    uint32_t parent, po, mv;
    float v1, v2, cr1, cr2;
};

AR_HD void backup_begin(BackupLane& b) {
    b.state = B_ENTRY;
    b.i = b.j = 0;
    b.parent = b.po = b.mv = 0;
    b.v1 = b.v2 = b.cr1 = b.cr2 = 0.0f;
}

// One round of the lane's current backup state. `ev` holds b_nn results in gather order.
template <int NW>
AR_HD void backup_round(BackupLane& b, Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, const EvalOut* ev,
                        const ZigTables* zt) {
    if (b.state == B_LEVEL) {
        // search.rs:834-851 one ancestor: everything of the parent arrives in one round trip
        NodeStats& P = m.stats[b.parent];
        NodeH0 a = P.h0;
        const NodeH1 h = P.h1;
        const NodeH2 c = P.h2;
        const uint32_t a1 = b.po & 0xffu, a2 = b.po >> 8;
        Edge e1 = P.e[0][a1], e2 = P.e[1][a2];
        const float q1 = b.cr1 + b.v1, q2 = b.cr2 + b.v2;
        finalize_h0(a, q1, q2, 1);
        edge_update(e1, q1, 1);
        edge_update(e2, q2, 1);
        P.h0 = a;
        P.e[0][a1] = e1;
        P.e[1][a2] = e2;
        s.nv_backup += 1;
        b.v1 = q1;
        b.v2 = q2;
        b.cr1 = h.r1;
        b.cr2 = h.r2;
        b.po = meta_po(c.meta, 0) | (meta_po(c.meta, 1) << 8);
        b.parent = h.parent;
        if (b.parent == NIL) b.state = B_ENTRY;
    } else if (b.state == B_ENTRY) {
        if (b.i >= s.n_proc) {
            b.i = 0;
            b.state = B_CANCEL;
            return;
        }
        const ProcEntry pe = m.proc[b.i++];
        NodeStats& N = m.stats[pe.node];
        NodeH0 a = N.h0;
        const NodeH1 h = N.h1;
        const NodeH2 c = N.h2;
        float g1 = 0.0f, g2 = 0.0f;
        if (proc_kind(pe.kind) == PROC_EVAL) {
            const EvalOut o = ev[proc_eval_index(pe.kind)];
            float red1[5], red2[5];
            reduce_prior(c.omap[0], o.p1, red1);  // populate_node (tree.rs:156-173)
            reduce_prior(c.omap[1], o.p2, red2);
            if (pe.node == s.root && cfg.noise_epsilon > 0.0f) {  // search.rs:1036-1050
                const bool ok1 =
                    dirichlet_mix(red1, meta_n(c.meta, 0), cfg.noise_epsilon, cfg.noise_concentration, s.rng, zt);
                const bool ok2 =
                    dirichlet_mix(red2, meta_n(c.meta, 1), cfg.noise_epsilon, cfg.noise_concentration, s.rng, zt);
                if (!ok1 || !ok2) s.error = 5;
            }
            for (int k = 0; k < 5; ++k) {
                N.e[0][k].prior = red1[k];
                N.e[1][k].prior = red2[k];
            }
            g1 = o.v1;
            g2 = o.v2;
        }
        finalize_h0(a, g1, g2, 1);  // leaf: finalize_score_update
        N.h0 = a;
        s.nv_backup += 1;
        b.v1 = g1;
        b.v2 = g2;
        b.cr1 = h.r1;
        b.cr2 = h.r2;
        b.po = meta_po(c.meta, 0) | (meta_po(c.meta, 1) << 8);
        b.parent = h.parent;
        if (b.parent != NIL) b.state = B_LEVEL;
    } else if (b.state == B_CANCEL) {
        // search.rs:860-889 cancel_shared_collisions
        if (b.i >= s.n_coll) {
            b.state = B_DONE;
            return;
        }
        const CollEntry ce = m.coll[b.i++];
        b.mv = ce.mv;
        const NodeH1 h = m.stats[ce.node].h1;
        const NodeH2 c = m.stats[ce.node].h2;
        b.po = meta_po(c.meta, 0) | (meta_po(c.meta, 1) << 8);
        b.parent = h.parent;
        if (b.parent != NIL) b.state = B_CANCEL_LEVEL;
    } else if (b.state == B_CANCEL_LEVEL) {
        NodeStats& P = m.stats[b.parent];
        const NodeH1 h = P.h1;
        const NodeH2 c = P.h2;
        const uint32_t a1 = b.po & 0xffu, a2 = b.po >> 8;
        P.h0.nif -= b.mv;
        P.e[0][a1].nif -= b.mv;
        P.e[1][a2].nif -= b.mv;
        b.po = meta_po(c.meta, 0) | (meta_po(c.meta, 1) << 8);
        b.parent = h.parent;
        if (b.parent == NIL) b.state = B_CANCEL;
    }
}

// search.rs:378-381 after the backup of one batch. Returns true when the search is complete.
template <int NW>
AR_HD bool batch_end(Slot<NW>& s) {
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

// The batch entries in backup order (by key; a no-op for the depth-first gathers, whose entries are written in order).
template <int NW>
AR_HD void proc_sort(const Slot<NW>& s, const Mem<NW>& m) {
    for (uint32_t i = 1; i < s.n_proc; ++i) {
        const ProcEntry pe = m.proc[i];
        uint32_t j = i;
        while (j > 0 && proc_key(m.proc[j - 1].kind) > proc_key(pe.kind)) {
            m.proc[j] = m.proc[j - 1];
            j -= 1;
        }
        if (j != i) m.proc[j] = pe;
    }
}

// backup of one simulate_batch for every lane of the wavefront (split kernels)
template <int NW>
AR_HD bool backup_machine(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, const EvalOut* ev, const ZigTables* zt) {
    proc_sort(s, m);
    BackupLane b;
    backup_begin(b);
    for (;;) {
        const uint32_t run = elect_state(b.state, B_DONE, B_DONE, 1);
        if (run == B_DONE) break;
        if (b.state == run) backup_round(b, s, m, cfg, ev, zt);
    }
    return batch_end(s);
}

template <int NW>
AR_HD void finish_move(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg);

// Fused form for evaluators computed inside the gather (SmartUniform): every lane runs up to `iters`
// whole simulate_batch cycles (gather -> backup -> bookkeeping) back to back without waiting for
// the other lanes between phases; the wavefront elects over the union of gather and backup states.
// A lane leaves when it has done its batches, finished a search (tree reuse is a separate kernel),
// or stalled for arena space.
template <int NW>
AR_HD void fused_machine(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, const ZigTables* zt, int iters) {
    enum { OFF_B = 8, ST_DONE = 16 };
    GatherLane<NW> g;
    BackupLane b;
    backup_begin(b);
    int left_iters = iters;
    uint32_t st = ST_DONE;  // combined state: gather states 0..3, backup states 8..11
    if (s.status == SLOT_ACTIVE && left_iters > 0 && gather_begin(g, s, cfg, EVAL_UNIFORM)) st = g.state;
    else g.state = G_DONE;
    for (;;) {
        const uint32_t run = elect_state(st, ST_DONE, ST_DONE, 0);
        if (run == ST_DONE) break;
        if (st != run) continue;
        if (st < OFF_B) {
            gather_round(g, s, m, cfg);
            if (g.state == G_DONE) {
                backup_begin(b);
                st = OFF_B + b.state;
            } else {
                st = g.state;
            }
        } else {
            backup_round(b, s, m, cfg, m.ev_local, zt);
            if (b.state == B_DONE) {
                st = ST_DONE;
                left_iters -= 1;
                if (batch_end(s)) finish_move(s, m, cfg);
                if (s.status == SLOT_ACTIVE && left_iters > 0 && gather_begin(g, s, cfg, EVAL_UNIFORM)) st = g.state;
            } else {
                st = OFF_B + b.state;
            }
        }
    }
}

// search.rs:899-910 + :945-955: revert a gathered batch after an evaluator failure
template <int NW>
AR_HD void cancel_batch(Slot<NW>& s, const Mem<NW>& m) {
    for (uint32_t i = 0; i < s.n_proc; ++i) {
        uint32_t cur = m.proc[i].node;
        m.stats[cur].h0.nif -= 1;
        for (;;) {
            const uint32_t parent = m.stats[cur].h1.parent;
            if (parent == NIL) break;
            const uint32_t meta = m.stats[cur].h2.meta;
            NodeStats& P = m.stats[parent];
            P.h0.nif -= 1;
            P.e[0][meta_po(meta, 0)].nif -= 1;
            P.e[1][meta_po(meta, 1)].nif -= 1;
            cur = parent;
        }
    }
    for (uint32_t i = 0; i < s.n_coll; ++i) {
        const uint32_t mv = m.coll[i].mv;
        uint32_t cur = m.coll[i].node;
        for (;;) {
            const uint32_t parent = m.stats[cur].h1.parent;
            if (parent == NIL) break;
            const uint32_t meta = m.stats[cur].h2.meta;
            NodeStats& P = m.stats[parent];
            P.h0.nif -= mv;
            P.e[0][meta_po(meta, 0)].nif -= mv;
            P.e[1][meta_po(meta, 1)].nif -= mv;
            cur = parent;
        }
    }
    s.n_proc = 0;
    s.n_coll = 0;
    s.batch_active = 0;
}

// ---- search.rs:249-296, 1079-1177: result extraction -----------------------------------------
AR_HD void extract_player(const Edge* e, uint32_t n, uint32_t omap, float node_value, float scale, uint32_t visits,
                          const SearchCfg& cfg, float* policy, float* visit_counts, float* prior5, float& value) {
    const uint32_t cv = visits > 0 ? visits - 1 : 0;
    for (int i = 0; i < 5; ++i) {
        policy[i] = 0.0f;
        visit_counts[i] = 0.0f;
        prior5[i] = 0.0f;
    }
    float mass = 0.0f;
    for (uint32_t i = 0; i < 5; ++i)
        if (i < n && e[i].visits > 0) mass += e[i].prior;
    const float fpu = node_value - cfg.fpu_reduction * scale * sqrtf(mass);
    float q[5], raw[5], qn[5], pruned[5];
    for (uint32_t i = 0; i < 5; ++i) {
        const bool live = i < n;
        q[i] = live ? (e[i].visits > 0 ? e[i].q : fpu) : 0.0f;
        raw[i] = live ? (float)e[i].visits : 0.0f;
        qn[i] = live ? q[i] / scale : 0.0f;
        pruned[i] = 0.0f;
    }
    if (n == 1) {
        pruned[0] = raw[0];
    } else {
        uint32_t best = 0;
        float best_v = raw[0], best_qn = qn[0], best_prior = e[0].prior;
        for (uint32_t i = 1; i < 5; ++i)
            if (i < n && raw[i] > best_v) {
                best = i;
                best_v = raw[i];
                best_qn = qn[i];
                best_prior = e[i].prior;
            }
        const float sqrt_total = sqrtf((float)(cv > 1 ? cv : 1));
        const float puct_star = best_qn + cfg.c_puct * best_prior * sqrt_total / (1.0f + best_v);
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) {
            if (i < n) {
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
    }
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        if (i < n) {
            const uint32_t act = outcome_action(omap, i);
#pragma unroll
            for (uint32_t a = 0; a < 5; ++a)
                if (a == act) {
                    visit_counts[a] = pruned[i];
                    prior5[a] = e[i].prior;
                }
        }
    }
    float sum = 0.0f;
    for (int i = 0; i < 5; ++i) sum += visit_counts[i];
    if (sum > 0.0f) {
        for (int i = 0; i < 5; ++i) policy[i] = visit_counts[i] / sum;
    } else {
        for (int i = 0; i < 5; ++i) policy[i] = prior5[i];
    }
    float vs = 0.0f;
    for (uint32_t i = 0; i < 5; ++i)
        if (i < n) vs += raw[i];
    if (vs > 0.0f) {
        float dot = 0.0f;
        for (uint32_t i = 0; i < 5; ++i)
            if (i < n) dot += q[i] * raw[i];
        value = dot / vs;
    } else {
        value = node_value;
    }
}

template <int NW>
AR_HD void extract_result(const Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg, MoveResult& r) {
    const NodeStats& R = m.stats[s.root];
    Edge e1[5], e2[5];
    for (int i = 0; i < 5; ++i) {
        e1[i] = R.e[0][i];
        e2[i] = R.e[1][i];
    }
    const NodeH0 a = R.h0;
    const NodeH1 b = R.h1;
    const NodeH2 c = R.h2;
    extract_player(e1, meta_n(c.meta, 0), c.omap[0], a.v1, b.scale, a.visits, cfg, r.policy[0], r.visit_counts[0],
                   r.prior[0], r.value[0]);
    extract_player(e2, meta_n(c.meta, 1), c.omap[1], a.v2, b.scale, a.visits, cfg, r.policy[1], r.visit_counts[1],
                   r.prior[1], r.value[1]);
    r.total_visits = a.visits;
    r.nn_evals = s.s_nn;
    r.terminals = s.s_term;
    r.collisions = s.s_coll;
}

// The end of one self-play turn (selfplay.rs:538-565) up to the tree reuse: extract, sample both
// actions, record, move. Leaves the slot in SLOT_ADVANCE (tree to be re-rooted by advance_tree_*)
// or SLOT_DONE (game over / single search): the tree work is done by a whole wavefront afterwards.
template <int NW>
AR_HD void finish_move(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg) {
    extract_result(s, m, cfg, s.last);
    if (s.single_search) {
        s.status = SLOT_DONE;
        return;
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
        PosRec<NW>& p = m.pos[s.n_pos];
        p.st = s.st;
        p.res = s.last;
        p.a1 = (uint8_t)a1;
        p.a2 = (uint8_t)a2;
    } else {
        s.error = 7;
    }
    s.n_pos += 1;
    // tree.rs:284-285: the child to keep, looked up before the position changes
    const NodeH2 c = m.stats[s.root].h2;
    const uint32_t ci = action_outcome(c.omap[0], (uint32_t)a1) * 5 + action_outcome(c.omap[1], (uint32_t)a2);
    const uint32_t child = m.stats[s.root].c[ci];
    float r1, r2;
    st_step(s.board, m.cost, s.st, (uint32_t)a1, (uint32_t)a2, r1, r2);
    if (st_over(s.board, s.st)) {
        s.status = SLOT_DONE;
        return;
    }
    s.remaining = cfg.n_sims;
    s.s_nn = 0;
    s.s_term = 0;
    s.s_coll = 0;
    s.pending_root = child;
    s.status = SLOT_ADVANCE;
}

// ---- tree reuse: tree.rs:283-302 as an in-place sliding compaction ------------------------------
// Ids grow from parent to child, so one pass in id order decides what is kept (keep[i] =
// keep[parent[i]]), assigns new ids (prefix count of kept nodes) and -- because new id <= old id --
// nodes can move in place. fwd[] holds the new id of every old node (NIL = dropped).
// This scalar form is what one lane does (CPU harness, growth path); the kernel runs the same two
// passes with a wavefront per game. Both give the same tree.
template <int NW>
AR_HD void advance_tree_scalar(Slot<NW>& s, const Mem<NW>& m) {
    const uint32_t keep_root = s.pending_root;
    if (keep_root == NIL) {
        make_root(s, m);
        s.status = SLOT_ACTIVE;
        return;
    }
    const uint32_t hi = s.hi;
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < hi; ++i) {
        bool keep = i == keep_root;
        if (!keep && i > keep_root) {
            const uint32_t p = m.stats[i].h1.parent;
            keep = p != NIL && m.fwd[p] != NIL;
        }
        m.fwd[i] = keep ? cnt++ : NIL;
    }
    for (uint32_t i = keep_root; i < hi; ++i) {
        const uint32_t ni = m.fwd[i];
        if (ni == NIL) continue;
        NodeStats nd = m.stats[i];
        nd.h1.parent = i == keep_root ? NIL : m.fwd[nd.h1.parent];
        for (int c = 0; c < 25; ++c)
            if (nd.c[c] != NIL) nd.c[c] = m.fwd[nd.c[c]];
        m.stats[ni] = nd;
    }
    s.root = 0;
    s.hi = cnt;
    s.node_count = cnt;  // tree.rs:290 count_subtree_nodes
    s.status = SLOT_ACTIVE;
}

// Start a game in a slot (selfplay.rs:526-535). The caller has filled board, st, rng and the arena.
template <int NW>
AR_HD void start_game_header(Slot<NW>& s, const SearchCfg& cfg) {
    s.n_pos = 0;
    s.t_sims = s.t_nn = s.t_term = s.t_coll = 0;
    s.nv_gather = s.nv_backup = s.new_nodes = 0;
    s.s_nn = s.s_term = s.s_coll = 0;
    s.n_proc = s.n_coll = 0;
    s.b_nn = s.b_term = s.b_coll = 0;
    s.batch_active = 0;
    s.error = 0;
    s.gather_pending = 0;
    s.g_rounds = 0;
    s.remaining = cfg.n_sims;
}
template <int NW>
AR_HD void start_game(Slot<NW>& s, const Mem<NW>& m, const SearchCfg& cfg) {
    start_game_header(s, cfg);
    make_root(s, m);
    if (!s.single_search && st_over(s.board, s.st)) s.status = SLOT_DONE;  // while !check_game_over()
    else s.status = SLOT_ACTIVE;
}

}  // namespace ar
