// The gather (search.rs:576-817, 961-999) as a WORK QUEUE OVER TREE LEVELS: a wavefront serves several games at once,
// and every node a pick_nodes_to_extend call reaches is an independent work item that any lane may take.
//
// Why. The reference walks a pick depth-first: allocate the budget at the root, go down into the first child that got
// visits, allocate there, ... -- about a hundred node visits in sequence per batch of sixteen descents, each a
// dependent memory trip (the gathers of rounds 1-2 were that sequence, one game per lane or per group of lanes: 2.6-2.9 ms
// per launch whatever the shape). But inside one pick the subtrees below different children do not
// depend on each other: visits only flow DOWN (a child's budget is fixed once its parent's allocation is done,
// search.rs:757-798), every node is reached at most once, and all writes of a pick (virtual losses, claims, new nodes)
// are to the node itself. What the depth-first order fixes is only
//   (a) the order of the batch entries and collision records (backup order matters: f32 running means),
//   (b) the order of the draws from the game's random stream (tie breaks, search.rs:511-532).
// So a pick is run level by level instead: all nodes of a level at once, ~12 dependent trips per pick instead of ~80.
//   * The `budget` visits of a pick are numbered 0..budget-1 in depth-first order; a node that routes k visits owns k
//     consecutive numbers ("visit slots") starting at its slot t, and its children, in child-slot order, take
//     t, t + k_0, t + k_0 + k_1, ... So an entry's slot is its position in the depth-first order and nothing ever moves.
//     For (a) a batch entry is written the moment its leaf is reached, in arrival order, with the key (pick number,
//     slot); the backup takes the entries in key order (proc_sort / backup16).
//   * (b): an allocation step that has to draw may only do so when every earlier slot is final (leaf reached), i.e.
//     when everything the depth-first walk would have done before it is done. Otherwise the entry is put back in the
//     queue untouched (nothing of an entry is written before its allocation is complete) and runs again later. With
//     network priors ties are rare below the root; with uniform priors the queue degenerates to the sequential order.
//   * Work items of all the wavefront's games share one ring in LDS; each pass the 64 lanes take the next 64 items.
//     The children of one parent are always taken in the same pass, so the parent's position (one record per slot in
//     LDS) is read by all of them before the first child, which inherits the parent's slot, overwrites it.
// Same arithmetic in the same order per node, same draws, same entry order as gather_round (dev_search.h): trees,
// batch entries and counters are identical; only the ids of the nodes created inside one pick are handed out in
// arrival order instead of depth-first order (ids are never compared across siblings; parent id < child id holds).
//
// The per-entry logic is written __host__ __device__ in three phases (fetch / visit / publish, + the end of a pick) so
// that tests/hostsim can run it on the CPU, one lane after the other per phase, against the oracle.
#pragma once
#include "dev_search.h"

namespace ar {

enum { GW_SLOTS = 16 };  // visit slots per pick (= largest batch size served)
enum { GW_SPILL = 7 };   // record reference of a stub: 0..R-1 = one of the game's records in LDS, GW_SPILL = the game's scratch in global memory

#if defined(__HIP_DEVICE_COMPILE__)
#define GW_ATOMIC_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GW_ATOMIC_OR(p, v) __hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GW_ATOMIC_CAS(p, expect, v) atomicCAS((p), (expect), (v))
#else
AR_HD uint32_t gw_host_add(uint32_t* p, uint32_t v) {
    const uint32_t o = *p;
    *p = o + v;
    return o;
}
AR_HD uint32_t gw_host_or(uint32_t* p, uint32_t v) {
    const uint32_t o = *p;
    *p = o | v;
    return o;
}
#define GW_ATOMIC_ADD(p, v) gw_host_add((p), (v))
AR_HD uint32_t gw_host_cas(uint32_t* p, uint32_t expect, uint32_t v) {
    const uint32_t o = *p;
    if (o == expect) *p = v;
    return o;
}
#define GW_ATOMIC_OR(p, v) gw_host_or((p), (v))
#define GW_ATOMIC_CAS(p, expect, v) gw_host_cas((p), (expect), (v))
#endif

struct alignas(16) GwU4 {
    uint32_t x, y, z, w;
};

// node.rs:251-283 compute_outcomes for the 17 effective-action maps a cell can have, tabulated once per block
// (device); the CPU harness computes them directly
struct GwOutcomeTable {
    uint32_t omap[17];
    uint32_t n[17];
};
AR_HD void gw_outcome_entry(uint32_t k, uint32_t& omap, uint32_t& n) {
    uint32_t eff = 4u << 12;
    if (k == 16) {
        eff = 4u | (4u << 3) | (4u << 6) | (4u << 9) | (4u << 12);
    } else {
        eff |= (k & 1u) ? 4u : 0u;  // bit d set: direction d is blocked -> STAY
        eff |= ((k & 2u) ? 4u : 1u) << 3;
        eff |= ((k & 4u) ? 4u : 2u) << 6;
        eff |= ((k & 8u) ? 4u : 3u) << 9;
    }
    pack_outcomes(eff, omap, n);
}
AR_HD uint32_t gw_outcome_key(const uint8_t* cost, uint8_t cell, uint8_t mud) {
    if (mud > 0) return 16u;
    const uint32_t c = cell_costs(cost, cell);
    return ((c & 0xffu) ? 0u : 1u) | (((c >> 8) & 0xffu) ? 0u : 2u) | (((c >> 16) & 0xffu) ? 0u : 4u) |
           (((c >> 24) & 0xffu) ? 0u : 8u);
}

// the position of an entry as its children (or the entry itself, when it is run again) find it
template <int NW>
struct alignas(8) GwRec {
    State<NW> pos;
    uint32_t omap0, omap1;  // the node's outcome -> action maps (children step with them)
    uint32_t node;
    uint32_t pad;
};

// one game as the wavefront sees it
template <int NW>
struct alignas(8) GwGame {
    Rng rng;  // only tie breaks draw from it (search.rs:511-532)
    State<NW> root_st;
    Board board;
    long long stats_off;  // arena of the game, relative to the arena base
    uint32_t slot;        // slot index (NIL: no game in this context)
    uint32_t root, cap;
    // (updated with LDS atomics by the lanes that work for the game)
    uint32_t hi, node_count, b_nn, d_new, d_visits, n_proc, n_coll, b_term, pick_mv;
    uint32_t final_mask;  // visit slots of the running pick that are final
    uint32_t b_coll;
    uint32_t next_game;   // (kernel) how many games this context has taken
    int32_t left;  // collision budget left (search.rs:970)
    uint8_t batch, budget, pick, error;  // pick: number of the running pick_nodes_to_extend call
    uint8_t running;  // 1: gather in progress
    uint8_t stalled, began;
    uint8_t parked;   // 1: the launch's pass limit was reached between two picks: the gather goes on in the next launch
#if defined(AR_STATS)
    uint32_t dbg_start;  // pass of the wavefront in which the game began
#endif
};

// Everything a wavefront keeps about its G games. A parent's position record is needed from the pass that publishes its
// children to the pass that takes them (the next one, as a rule); a game rarely has more than two or three such records
// alive, so it owns R of them in LDS and a parent that finds them all taken puts its record in the game's scratch in
// global memory instead (the slot's idle level-stack area; the children then read it from there).
template <int NW, int G, int R>
struct GwShared {
    enum { RING = G * GW_SLOTS };  // every entry of every game fits (G a power of two)
    GwGame<NW> game[G];
    GwRec<NW> rec[G][R];
    uint32_t rec_owner[G][R];         // 0: free; else slot + 1 of the entry whose record it is
    uint32_t stub[G][GW_SLOTS];
    uint32_t stub_node[G][GW_SLOTS];  // CHILD stubs: the child's node id (NIL: to be created), set by the parent
    uint16_t ring[RING];
    uint32_t tail;
};

// stub of a queued entry: bit 0 child (position = parent's + one move), bit 1 first level of a pick, idx, visits, where
// the record is (its own for a re-queued entry, the parent's for a child): reference (3 bits) and slot of its owner
AR_HD uint32_t gw_stub(bool child, bool from_pick, uint32_t idx, uint32_t k, uint32_t recref, uint32_t rslot) {
    return (child ? 1u : 0u) | (from_pick ? 2u : 0u) | (idx << 2) | (k << 7) | (recref << 12) | (rslot << 15);
}
// ring item: game context (bits 0..6; bit 7 is the kernel's "context wants a game" flag), visit slot, siblings behind it
AR_HD uint16_t gw_item(uint32_t g, uint32_t t, uint32_t rem) { return (uint16_t)(g | (t << 8) | (rem << 12)); }

// what a lane carries through the phases of one pass (registers on the device)
template <int NW>
struct GwLane {
    bool active;
    uint32_t g, t, k, idx;
    bool child, from_pick;
    uint32_t free_ref;  // record in LDS this entry releases (it was the last to read it), or GW_SPILL
    State<NW> pos;
    uint32_t node, parent;
    float r1, r2;
    // results of the visit
    bool is_final, wait, interior;
    uint32_t fin_node, fin_kind, coll_mv, arr;
    uint32_t omap0, omap1, mask;
    uint32_t vtp[5];  // visits allocated to child slot (o1, o2): field o2 (6 bits) of word o1
    uint32_t kid[25];  // the node's child table (it arrives with the record: the children need no trip of their own for their ids)
#if defined(AR_STATS)
    uint32_t dbg_steps;               // allocation steps of this entry
    unsigned long long dbg_t[3];      // 100 MHz clock: record arrived, set-up done, allocation done
#endif
};

// Element `i` of a small array that lives in registers, as an OR over masked elements: a chain of `i == 0 ? a[0] : ...`
// selects is rewritten by the optimizer into one load through a selected pointer, which pins the whole lane state in
// scratch memory; a select between an element and zero is not.
AR_HD uint32_t gw_sel5u(const uint32_t* a, uint32_t i) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t j = 0; j < 5; ++j) r |= (i == j) ? a[j] : 0u;
    return r;
}
AR_HD float gw_sel5f(const float* a, uint32_t i) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t j = 0; j < 5; ++j) r |= (i == j) ? f32_to_bits(a[j]) : 0u;
    return bits_to_f32(r);
}
// kid[idx]: the row of five first (masks), then the column
AR_HD uint32_t gw_pick25(const uint32_t* a, uint32_t idx) {
    const uint32_t row = idx / 5, col = idx % 5;
    uint32_t r[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (uint32_t q = 0; q < 5; ++q) {
        const uint32_t mk = row == q ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 5; ++j) r[j] |= a[5 * q + j] & mk;
    }
    return gw_sel5u(r, col);
}

// ---- per-player allocation state (the lane kernel's HalfAlloc with the added-visits counters packed) ------------
struct GwHalf {
    float score[5], util[5], num[5];
    uint32_t ns[5], nif0[5];
    uint32_t add;     // 5 x 6 bits: visits added to outcome i in this allocation
    uint32_t forced;  // bit i: forced-playout score (search.rs:493-498)
    uint32_t n;
};

// search.rs:478-498 (same expressions as half_init, dev_search.h)
AR_HD void gw_half_init(GwHalf& h, const Edge* e, uint32_t n, float node_value, float scale, uint32_t cv,
                        const SearchCfg& cfg, bool is_root) {
    h.n = n;
    h.forced = 0;
    h.add = 0;
    float mass = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i)
        if (i < n && e[i].visits > 0) mass += e[i].prior;
    const float fpu = node_value - cfg.fpu_reduction * scale * sqrtf(mass);
    const float sqrt_total = sqrtf((float)(cv > 1 ? cv : 1));
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        const bool live = i < n;
        const float q = e[i].visits > 0 ? e[i].q : fpu;
        h.util[i] = q / scale;
        h.num[i] = cfg.c_puct * e[i].prior * sqrt_total;
        h.ns[i] = live ? e[i].visits + e[i].nif : 0;
        h.nif0[i] = e[i].nif;
        h.score[i] = h.util[i] + h.num[i] / (1.0f + (float)h.ns[i]);
    }
    if (is_root && cfg.force_k > 0.0f) {
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) {
            if (i < n && e[i].prior > 0.0f) {
                const float threshold = sqrtf(cfg.force_k * e[i].prior * (float)cv);
                if ((float)e[i].visits < threshold) {
                    h.score[i] = 1e20f;
                    h.forced |= 1u << i;
                }
            }
        }
    }
}

// search.rs:463-554 estimated_visits_to_change_best_half on the cached scores (half_best, dev_search.h). A tie pass
// draws from the game's stream only when `may_draw`; otherwise `wait` is set and nothing has been consumed.
AR_HD void gw_half_best(const GwHalf& h, bool may_draw, Rng& rng, bool& wait, bool& drew, uint32_t& best_out,
                        uint32_t& vtc_out) {
    const uint32_t n = h.n;
    best_out = 0;
    vtc_out = 0xFFFFFFFFu;
    if (n <= 1) return;
    const float NEG_INF = -__builtin_inff();
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
    // (the first outcome the tie pass below would count: it is tested before any draw has moved `best`)
    bool any_tie = false;
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) any_tie = any_tie || (i < n && i != best && fabsf(h.score[i] - best_score) < 1e-12f);
    if (any_tie) {
        if (!may_draw) {
            wait = true;
            return;
        }
        drew = true;
        uint32_t ties = 1;
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) {
            if (i < n && i != best && fabsf(h.score[i] - best_score) < 1e-12f) {  // (`best` moves with the draws, as in the reference)
                ties += 1;
                if (rng_below(rng, ties) == 0) best = i;
            }
        }
    }
    best_out = best;
    const float best_util = gw_sel5f(h.util, best);
    if (second <= NEG_INF) return;
    if (best_util >= second) return;
    const float denom = second - best_util;
    if (denom <= 0.0f) return;
    const float n1 = (float)gw_sel5u(h.ns, best) + 1.0f;
    float vtc = gw_sel5f(h.num, best) / denom - n1 + 1.0f;
    if (!(vtc > 1.0f)) vtc = 1.0f;
    const uint32_t k = vtc >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)vtc;
    vtc_out = k > 1 ? k : 1;
}

AR_HD void gw_half_take(GwHalf& h, uint32_t b, uint32_t k) {
    const uint32_t nsb = gw_sel5u(h.ns, b) + k;
    const float sc = gw_sel5f(h.util, b) + gw_sel5f(h.num, b) / (1.0f + (float)nsb);
    const bool keep_forced = (h.forced >> b) & 1u;
    h.add += k << (6u * b);
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
        const bool hit = i == b;
        h.ns[i] = hit ? nsb : h.ns[i];
        h.score[i] = (hit && !keep_forced) ? sc : h.score[i];
    }
}

// One game's addresses: the arena from the game's offset, the per-slot scratch from the layout
template <int NW>
struct GwMem {
    unsigned char* arena;
    unsigned char* scratch;
    const uint8_t* maze;  // cost tables (LDS copy when the run has one shared maze)
    size_t slot_bytes;
    uint32_t proc_off, coll_off, leaf_off, spill_off, coll_cap;
    AR_HD NodeStats* stats(const GwGame<NW>& G) const { return (NodeStats*)(arena + G.stats_off); }
    AR_HD const uint8_t* cost(const GwGame<NW>& G) const { return maze + G.board.maze_off; }
    AR_HD ProcEntry* proc(const GwGame<NW>& G) const { return (ProcEntry*)(scratch + (size_t)G.slot * slot_bytes + proc_off); }
    AR_HD CollEntry* coll(const GwGame<NW>& G) const { return (CollEntry*)(scratch + (size_t)G.slot * slot_bytes + coll_off); }
    AR_HD State<NW>* leaves(const GwGame<NW>& G) const { return (State<NW>*)(scratch + (size_t)G.slot * slot_bytes + leaf_off); }
    AR_HD GwRec<NW>* spill(const GwGame<NW>& G) const { return (GwRec<NW>*)(scratch + (size_t)G.slot * slot_bytes + spill_off); }
};

// Starts the next pick_nodes_to_extend call of a game (search.rs:981-999), or ends its gather. Returns true when an
// item for the root was queued (the root's position is the game's: it needs no record).
template <int NW>
AR_HD bool gw_next_pick(GwGame<NW>& G, uint32_t* stub, uint16_t* ring, uint32_t ring_mask, uint32_t* tail, uint32_t g) {
    if (!(G.n_proc < G.batch && G.left > 0)) {
        G.running = 0;
        return false;
    }
    uint32_t budget = (uint32_t)G.left;
    if (G.batch - G.n_proc < budget) budget = G.batch - G.n_proc;
    G.budget = budget;
    G.pick += 1;
    G.pick_mv = 0;
    G.final_mask = 0;
    stub[0] = gw_stub(false, true, 0, budget, GW_SPILL, 0);
    const uint32_t at = GW_ATOMIC_ADD(tail, 1u);
    ring[at & ring_mask] = gw_item(g, 0, 0);
    return true;
}

// ---- phase 1: take an item; the entry's position; the id of the child it stands for -------------------------------
template <int NW, int R>
AR_HD void gw_fetch(GwLane<NW>& ln, uint32_t item, const GwGame<NW>* games, const GwRec<NW>* rec /*[G][R]*/,
                    const uint32_t* stub /*[G][SLOTS]*/, const uint32_t* stub_node /*[G][SLOTS]*/, const GwMem<NW>& m) {
    ln.g = item & 0x7fu;
    ln.t = (item >> 8) & 0xfu;
    const uint32_t rem = (item >> 12) & 0xfu;
    const uint32_t st = stub[ln.g * GW_SLOTS + ln.t];
    ln.child = st & 1u;
    ln.from_pick = (st >> 1) & 1u;
    ln.idx = (st >> 2) & 31u;
    ln.k = (st >> 7) & 31u;
    const uint32_t ref = (st >> 12) & 7u, rslot = (st >> 15) & 15u;
    const GwGame<NW>& G = games[ln.g];
    ln.parent = NIL;
    ln.r1 = ln.r2 = 0.0f;
    ln.free_ref = GW_SPILL;
    if (ln.from_pick) {
        ln.pos = G.root_st;
        ln.node = G.root;
        return;
    }
    GwRec<NW> Rc;
    if (ref < (uint32_t)R) Rc = rec[ln.g * R + ref];
    else Rc = m.spill(G)[rslot];
    // the last reader of a record in LDS releases it: the last of the siblings, or the re-queued entry itself
    if (ref < (uint32_t)R && (!ln.child || rem == 0)) ln.free_ref = ref;
    ln.pos = Rc.pos;
    ln.node = Rc.node;
    if (ln.child) {
        ln.parent = Rc.node;
        const uint32_t o1 = ln.idx / 5, o2 = ln.idx % 5;
        st_step(G.board, m.cost(G), ln.pos, outcome_action(Rc.omap0, o1), outcome_action(Rc.omap1, o2), ln.r1, ln.r2);
        ln.node = stub_node[ln.g * GW_SLOTS + ln.t];  // NIL: no such child yet
    }
}

// ---- phase 2: look at the node (search.rs:591-636 root, :675-725 child, :742-817 build_gather_level) ---------------
template <int NW>
AR_HD void gw_visit(GwLane<NW>& ln, GwGame<NW>& G, uint32_t* rec_owner /*[R] of the game*/, const GwMem<NW>& m,
                    const SearchCfg& cfg, const GwOutcomeTable* otab) {
    if (ln.free_ref != GW_SPILL) rec_owner[ln.free_ref] = 0;  // (every reader of it ran phase 1 of this pass)
    ln.is_final = ln.wait = ln.interior = false;
    ln.fin_node = NIL;
    ln.fin_kind = PROC_NONE;
    ln.coll_mv = 0;
    ln.arr = 0;
    ln.mask = 0;
    NodeStats* stats = m.stats(G);
    const uint8_t* cost = m.cost(G);
    if (ln.child && ln.node == NIL) {
        // new leaf: shell creation + claim are stores only (tree.rs:107-148, search.rs:675-701)
        const uint32_t nid = GW_ATOMIC_ADD(&G.hi, 1u);
        ln.is_final = true;
        if (nid >= G.cap) {  // excluded by the capacity check at the start of the gather
            G.error = 3;
            ln.coll_mv = ln.k;
            return;
        }
        const bool over = st_over(G.board, ln.pos);
        uint32_t om1, om2, n1, n2;
#if defined(__HIP_DEVICE_COMPILE__)
        {
            const uint32_t k1 = gw_outcome_key(cost, ln.pos.p1, ln.pos.m1), k2 = gw_outcome_key(cost, ln.pos.p2, ln.pos.m2);
            om1 = otab->omap[k1];
            n1 = otab->n[k1];
            om2 = otab->omap[k2];
            n2 = otab->n[k2];
        }
#else
        (void)otab;
        pack_outcomes(eff_actions(cost, ln.pos.p1, ln.pos.m1), om1, n1);
        pack_outcomes(eff_actions(cost, ln.pos.p2, ln.pos.m2), om2, n2);
#endif
        GwU4* S = (GwU4*)&stats[nid];
        const GwU4 zero = {0u, 0u, 0u, 0u}, nil4 = {NIL, NIL, NIL, NIL};
#pragma unroll
        for (int j = 0; j < 10; ++j) S[j] = zero;  // edges: prior 0, q 0, visits 0, in flight 0
        const GwU4 h0 = {0u, 0u, 0u, 1u};  // v1 0, v2 0, visits 0, in flight 1 (try_start_score_update on a fresh node)
        S[10] = h0;
        const GwU4 h1 = {f32_to_bits((float)(ln.pos.remaining > 1 ? ln.pos.remaining : 1)), f32_to_bits(ln.r1),
                         f32_to_bits(ln.r2), ln.parent};
        S[11] = h1;
        const GwU4 h2 = {om1, om2, n1 | (n2 << 8) | ((ln.idx / 5) << 16) | ((ln.idx % 5) << 24), over ? 1u : 0u};
        S[12] = h2;
#pragma unroll
        for (int j = NODE_KID_GROUP; j < NODE_GROUPS; ++j) S[j] = nil4;
        stats[ln.parent].c[ln.idx] = nid;
        GW_ATOMIC_ADD(&G.node_count, 1u);
        GW_ATOMIC_ADD(&G.d_new, 1u);
        ln.fin_node = nid;
        ln.fin_kind = over ? PROC_TERMINAL : PROC_EVAL;
        ln.coll_mv = ln.k > 1 ? ln.k - 1 : 0;
    } else {
        // the record of the node: thirteen 16-byte groups, one round trip
        const NodeStats& N = stats[ln.node];
        Edge e1[5], e2[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            e1[i] = N.e[0][i];
            e2[i] = N.e[1][i];
        }
        const NodeH0 a = N.h0;
        const NodeH1 b = N.h1;
        const NodeH2 c = N.h2;
#if defined(AR_STATS) && defined(__HIP_DEVICE_COMPILE__)
        ln.dbg_steps = 0;
        ln.dbg_t[0] = ln.dbg_t[1] = ln.dbg_t[2] = (a.visits + c.terminal + e1[4].visits + e2[4].visits) ? wall_clock64() : wall_clock64() + 0;
#endif
        // its child table rides along in the same round trip (used if the node turns out to be interior)
        uint32_t kid_in[25];
#pragma unroll
        for (int j = 0; j < 25; ++j) kid_in[j] = N.c[j];
        if (a.visits == 0 || c.terminal != 0) {
            // leaf or terminal (search.rs:591-636 for the root, :675-706 for a child)
            ln.is_final = true;
            ln.fin_node = ln.node;
            if (!(a.visits == 0 && a.nif > 0)) {  // try_start_score_update
                stats[ln.node].h0.nif = a.nif + 1;
                const bool term = c.terminal != 0 || st_over(G.board, ln.pos);
                if (term && a.visits == 0) stats[ln.node].h2.terminal = 1;
                ln.fin_kind = term ? PROC_TERMINAL : PROC_EVAL;
                ln.coll_mv = ln.k > 1 ? ln.k - 1 : 0;
            } else {
                ln.coll_mv = ln.k;
            }
        } else {
            // visited interior node: split its visits among the children (build_gather_level, search.rs:742-817)
            const uint32_t cv = a.visits > 0 ? a.visits - 1 : 0;
            GwHalf h1, h2;
            gw_half_init(h1, e1, meta_n(c.meta, 0), a.v1, b.scale, cv, cfg, ln.from_pick);
            gw_half_init(h2, e2, meta_n(c.meta, 1), a.v2, b.scale, cv, cfg, ln.from_pick);
            // a draw is in depth-first order only when every earlier visit slot of the pick is final
            const uint32_t earlier = (1u << ln.t) - 1u;
            const bool may_draw = (G.final_mask & earlier) == earlier;
            uint32_t left = ln.k, mask = 0;
            uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
            bool wait = false;
#if defined(AR_STATS) && defined(__HIP_DEVICE_COMPILE__)
            ln.dbg_t[1] = (h1.score[0] + h2.score[0] + h1.score[4] + h2.score[4]) != 12345.0f ? wall_clock64() : 0;
#endif
            Rng rng = G.rng;
            bool drew = false;
            while (left > 0) {
                uint32_t b1, b2, c1, c2;
                gw_half_best(h1, may_draw, rng, wait, drew, b1, c1);
                if (wait) break;
                gw_half_best(h2, may_draw, rng, wait, drew, b2, c2);
                if (wait) break;
                uint32_t kk = left;
                if (c1 < kk) kk = c1;
                if (c2 < kk) kk = c2;
                if (kk < 1) kk = 1;
                const uint32_t add = kk << (6u * b2);
                v0 += b1 == 0 ? add : 0u;
                v1 += b1 == 1 ? add : 0u;
                v2 += b1 == 2 ? add : 0u;
                v3 += b1 == 3 ? add : 0u;
                v4 += b1 == 4 ? add : 0u;
                mask |= 1u << (b1 * 5 + b2);
                gw_half_take(h1, b1, kk);
                gw_half_take(h2, b2, kk);
                left -= kk;
#if defined(AR_STATS) && defined(__HIP_DEVICE_COMPILE__)
                ln.dbg_steps += 1;
#endif
            }
#if defined(AR_STATS) && defined(__HIP_DEVICE_COMPILE__)
            ln.dbg_t[2] = (mask + left) != 0xFFFFFFF0u ? wall_clock64() : 0;
#endif
            if (wait) {
                ln.wait = true;  // nothing was written, nothing was drawn: the entry runs again later
                return;
            }
            if (drew) G.rng = rng;  // (only the first unfinished entry of a game can have drawn: one writer per pass)
            // the visits enter the node (search.rs:639 / :711-716) and the edges they were routed to (:800-814)
            NodeStats& W = stats[ln.node];
            W.h0.nif = a.nif + ln.k;
#pragma unroll
            for (uint32_t i = 0; i < 5; ++i) {
                const uint32_t a1 = (h1.add >> (6u * i)) & 63u, a2 = (h2.add >> (6u * i)) & 63u;
                if (a1) W.e[0][i].nif = h1.nif0[i] + a1;
                if (a2) W.e[1][i].nif = h2.nif0[i] + a2;
            }
            GW_ATOMIC_ADD(&G.d_visits, 1u);
            ln.interior = true;
            ln.omap0 = c.omap[0];
            ln.omap1 = c.omap[1];
            ln.mask = mask;
            ln.vtp[0] = v0;
            ln.vtp[1] = v1;
            ln.vtp[2] = v2;
            ln.vtp[3] = v3;
            ln.vtp[4] = v4;
#pragma unroll
            for (int j = 0; j < 25; ++j) ln.kid[j] = kid_in[j];
        }
    }
    if (ln.is_final) {
        // the batch entry (search.rs:598-633, 681-701) and the collision record, in arrival order; the entry carries its
        // place in the depth-first order (pick number, visit slot) and the index of its evaluation request
        if (ln.fin_kind != PROC_NONE) {
            uint32_t word = ln.fin_kind;
            if (ln.fin_kind == PROC_EVAL) {
                ln.arr = GW_ATOMIC_ADD(&G.b_nn, 1u);
                m.leaves(G)[ln.arr] = ln.pos;
                word |= ln.arr << 8;
            } else {
                GW_ATOMIC_ADD(&G.b_term, 1u);
            }
            const uint32_t i = GW_ATOMIC_ADD(&G.n_proc, 1u);
            if (i >= cfg.batch_size) {
                G.error = 1;
            } else {
                ProcEntry pe;
                pe.node = ln.fin_node;
                pe.kind = word | ((((uint32_t)G.pick << 4) | ln.t) << 16);
                m.proc(G)[i] = pe;
            }
        }
        if (ln.coll_mv) {
            GW_ATOMIC_ADD(&G.pick_mv, ln.coll_mv);
            const uint32_t j = GW_ATOMIC_ADD(&G.n_coll, 1u);
            if (j >= m.coll_cap) {
                G.error = 2;
            } else {
                CollEntry ce;
                ce.node = ln.fin_node;
                ce.mv = ln.coll_mv;
                m.coll(G)[j] = ce;
            }
        }
    }
}

// ---- phase 3: publish: children into the queue / the entry back into the queue / the final record -----------------
// Returns true for the lane whose entry completed the pick (it runs gw_finish_pick).
template <int NW, int R>
AR_HD bool gw_publish(const GwLane<NW>& ln, GwGame<NW>& G, GwRec<NW>* rec /*[R] of the game*/, uint32_t* rec_owner,
                      GwRec<NW>* spill /*[SLOTS] of the game, global*/, uint32_t* stub, uint32_t* stub_node,
                      uint16_t* ring, uint32_t ring_mask, uint32_t* tail) {
    if (ln.is_final) {
        const uint32_t bits = ((1u << ln.k) - 1u) << ln.t;
        const uint32_t old = GW_ATOMIC_OR(&G.final_mask, bits);
        const uint32_t full = (1u << G.budget) - 1u;
        return (old | bits) == full && old != full;
    }
    // the entry's record (its children step from it; a re-queued entry starts from it again): one of the game's
    // records in LDS if one is free, else the game's scratch
    GwRec<NW> r;
    r.pos = ln.pos;
    r.omap0 = ln.interior ? ln.omap0 : 0u;
    r.omap1 = ln.interior ? ln.omap1 : 0u;
    r.node = ln.node;
    r.pad = 0;
    uint32_t ref = GW_SPILL;
    for (uint32_t j = 0; j < (uint32_t)R; ++j) {
        if (ref == GW_SPILL && GW_ATOMIC_CAS(&rec_owner[j], 0u, ln.t + 1u) == 0u) ref = j;
    }
    if (ref != GW_SPILL) rec[ref] = r;
    else spill[ln.t] = r;
    if (ln.wait) {
        stub[ln.t] = gw_stub(false, false, 0, ln.k, ref, ln.t);
        const uint32_t at = GW_ATOMIC_ADD(tail, 1u);
        ring[at & ring_mask] = gw_item(ln.g, ln.t, 0);
        return false;
    }
    // children in child-slot order take consecutive runs of this entry's visit slots
    uint32_t mm = ln.mask;
    uint32_t cnt = 0;
    for (uint32_t x = mm; x; x &= x - 1) cnt += 1;
    uint32_t at = GW_ATOMIC_ADD(tail, cnt);
    uint32_t slot = ln.t, left = cnt;
    while (mm) {
        const uint32_t idx = (uint32_t)lowest_bit(mm);
        mm &= mm - 1;
        const uint32_t kj = (gw_sel5u(ln.vtp, idx / 5) >> (6u * (idx % 5))) & 63u;
        left -= 1;
        stub[slot] = gw_stub(true, false, idx, kj, ref, ln.t);
        stub_node[slot] = gw_pick25(ln.kid, idx);
        ring[at & ring_mask] = gw_item(ln.g, slot, left);
        at += 1;
        slot += kj;
    }
    return false;
}

// ---- the end of a pick_nodes_to_extend call: the outer loop of simulate_batch (search.rs:981-999) ------------------
// `go_on` false: the launch has used up its passes; a gather that is not complete is parked here, between two picks (all
// its entries are final, everything it did is in the tree), and the next launch continues it. Which launch runs which pick
// does not change what a pick does.
template <int NW>
AR_HD void gw_finish_pick(GwGame<NW>& G, uint32_t* stub, uint16_t* ring, uint32_t ring_mask, uint32_t* tail, uint32_t g,
                          bool go_on) {
    G.b_coll += G.pick_mv;
    G.left -= (int32_t)G.pick_mv;
    G.pick_mv = 0;
    if (!go_on && G.n_proc < G.batch && G.left > 0) {
        G.running = 0;
        G.parked = 1;
        return;
    }
    gw_next_pick(G, stub, ring, ring_mask, tail, g);
}

// Starts one simulate_batch's gather for a game context (gather_begin, dev_search.h). `S` is the game's slot.
template <int NW>
AR_HD void gw_begin(GwGame<NW>& G, const Slot<NW>& S, uint32_t slot, const SearchCfg& cfg) {
    G.rng = S.rng;
    G.root_st = S.st;
    G.board = S.board;
    G.stats_off = S.stats_off;
    G.slot = slot;
    G.root = S.root;
    G.hi = S.hi;
    G.cap = S.cap;
    G.node_count = S.node_count;
    G.n_proc = G.n_coll = G.b_nn = G.b_term = G.b_coll = 0;
    G.parked = 0;
    G.batch = S.remaining < cfg.batch_size ? S.remaining : cfg.batch_size;
    G.budget = 0;
    G.pick = 0;
    G.pick_mv = 0;
    G.final_mask = 0;
    G.error = 0;
    G.d_new = G.d_visits = 0;
    G.running = 0;
    G.stalled = 0;
    G.began = 0;
    G.left = 0;
    if (G.hi + G.batch > G.cap && !S.gather_pending) {  // (a parked gather passed this check when it began)
        G.stalled = 1;
        return;
    }
    G.began = 1;
    G.running = 1;
    if (S.gather_pending) {  // a parked gather: where it stood after its last pick
        G.n_proc = S.n_proc;
        G.n_coll = S.n_coll;
        G.b_nn = S.b_nn;
        G.b_term = S.b_term;
        G.b_coll = S.b_coll;
        G.left = S.g_left;
        G.pick = (uint8_t)S.g_pick;
        return;
    }
    G.left = (int32_t)collisions_left(G.node_count, cfg);
}

// The end of a game's gather: what the slot header takes back (the tail of k_gather8).
template <int NW>
AR_HD void gw_end(const GwGame<NW>& G, Slot<NW>& S, const SearchCfg& cfg) {
    if (G.stalled) {
        S.need_nodes = S.hi + cfg.n_sims + 2 * cfg.batch_size;
        return;
    }
    if (!G.began) return;
    S.hi = G.hi;
    S.node_count = G.node_count;
    S.new_nodes += G.d_new;
    S.nv_gather += G.d_visits;
    S.n_proc = G.n_proc < cfg.batch_size ? G.n_proc : cfg.batch_size;  // (the error paths keep counting)
    S.n_coll = G.n_coll;
    S.b_nn = G.b_nn;
    S.b_term = G.b_term;
    S.b_coll = G.b_coll;
    S.rng = G.rng;
    S.g_rounds = 0;
    if (G.parked) {
        S.gather_pending = 1;
        S.g_left = G.left;
        S.g_pick = G.pick;
        if (G.error) S.error = G.error;
        return;
    }
    S.batch_active = 1;
    S.gather_pending = 0;
    if (G.error) S.error = G.error;
    else if (G.running) S.error = 8;  // the queue ran dry with the gather unfinished (a bug guard)
}

}  // namespace ar
