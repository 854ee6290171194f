// TEST HARNESS -- compiles the product's device-side search logic (alpharat_amd/csrc/dev_*.h,
// written __host__ __device__) for the CPU so its control flow, arena management and arithmetic
// can be checked against the oracle in the GPU-less build container. It is NOT a CPU fallback:
// nothing in alpharat_amd/ loads this file, and libalpharat_hip.so has no CPU path.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../alpharat_amd/csrc/slot_layout.h"
#include "../../alpharat_amd/csrc/dev_gatherw.h"
#include "../../alpharat_amd/csrc/zig_norm_tables.inc"

using namespace ar;

static const ZigTables g_zig = {AR_ZIG_NORM_X_INIT, AR_ZIG_NORM_F_INIT};

extern "C" {

struct HsCfg {
    float c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration;
    uint32_t coll_min, coll_max, coll_start, coll_end;
    float coll_power;
};
struct HsGame {
    uint8_t width, height;
    uint16_t max_turns, turn;
    uint8_t p1_x, p1_y, p2_x, p2_y, p1_mud, p2_mud;
    float p1_score, p2_score;
    const uint8_t* cost;    // [hw*4]
    const uint8_t* cheese;  // [hw]
};

struct HsRun {
    Slot<4> slot;
    std::vector<unsigned char> scratch;
    std::vector<unsigned char> arena;  // [node records | fwd]
    std::vector<uint8_t> cost;
    SearchCfg cfg;
    SlotLayout L;
    uint32_t grows = 0;
    uint64_t wide_passes = 0, wide_gathers = 0;
    Mem<4> mem() { return resolve_mem<4>(slot, arena.data(), scratch.data(), 0, L, cost.data()); }
    const Mem<4> cmem() const {
        return resolve_mem<4>(slot, const_cast<unsigned char*>(arena.data()),
                              const_cast<unsigned char*>(scratch.data()), 0, L, cost.data());
    }
};

static SearchCfg to_cfg(const HsCfg* c, uint32_t n_sims, uint32_t batch) {
    SearchCfg s;
    s.c_puct = c->c_puct;
    s.fpu_reduction = c->fpu_reduction;
    s.force_k = c->force_k;
    s.noise_epsilon = c->noise_epsilon;
    s.noise_concentration = c->noise_concentration;
    s.coll_min = c->coll_min;
    s.coll_max = c->coll_max;
    s.coll_start = c->coll_start;
    s.coll_end = c->coll_end;
    s.coll_power = c->coll_power;
    s.n_sims = n_sims;
    s.batch_size = batch;
    s.alloc_per_round = 2;
    return s;
}

static void set_arena(HsRun* r, uint32_t cap) {
    r->slot.cap = cap;
    r->slot.stats_off = 0;
    r->slot.fwd_off = (long long)((size_t)cap * sizeof(NodeStats));
}

// what the runtime does for a stalled slot: a doubled arena, live nodes copied across unchanged
static void grow(HsRun* r) {
    uint32_t ncap = r->slot.cap * 2;
    while (ncap < r->slot.need_nodes) ncap *= 2;
    std::vector<unsigned char> na(arena_bytes(ncap) + 256);
    const uint32_t hi = r->slot.hi;
    std::memcpy(na.data(), r->arena.data(), (size_t)hi * sizeof(NodeStats));
    r->arena.swap(na);
    set_arena(r, ncap);
    r->slot.status = SLOT_ACTIVE;
    r->grows += 1;
}

// The work-queue gather (dev_gatherw.h) run the way a wavefront runs it: every pass takes a run of queued items (never
// splitting the children of one parent), then phase by phase, lane after lane. `sched` varies how many items a pass
// takes (1 = as many as 64 lanes can; otherwise a seeded random cut), so the result is checked not to depend on it.
}  // extern "C"
template <int R>
static bool wide_gather(HsRun* r, uint64_t& sched, uint32_t pass_limit) {
    typedef GwShared<4, 1, R> Sh;
    static thread_local Sh sh;
    const uint32_t ring_mask = Sh::RING - 1;
    Slot<4>& s = r->slot;
    GwMem<4> m;
    m.arena = r->arena.data();
    m.scratch = r->scratch.data();
    m.maze = r->cost.data();
    m.slot_bytes = r->L.total;
    m.proc_off = (uint32_t)r->L.proc_off;
    m.coll_off = (uint32_t)r->L.coll_off;
    m.leaf_off = (uint32_t)r->L.leaf_off;
    m.spill_off = (uint32_t)r->L.levels_off;
    m.coll_cap = r->L.coll_cap;
    GwGame<4>& G = sh.game[0];
    gw_begin(G, s, 0, r->cfg);
    sh.tail = 0;
    for (int j = 0; j < R; ++j) sh.rec_owner[0][j] = 0;
    uint32_t head = 0;
    if (G.began) gw_next_pick(G, sh.stub[0], sh.ring, ring_mask, &sh.tail, 0);
    GwLane<4> lanes[64];
    bool finisher[64];
    uint64_t passes = 0;
    while (head != sh.tail) {
        uint32_t n = sh.tail - head, take = n < 64 ? n : 64;
        if (sched != 1) {
            sched = sched * 6364136223846793005ULL + 1442695040888963407ULL;
            take = 1 + (uint32_t)((sched >> 33) % take);
        }
        // whole sibling groups only
        while (true) {
            const uint32_t last = sh.ring[(head + take - 1) & ring_mask];
            const uint32_t rem = (last >> 12) & 15u;
            if (rem == 0) break;
            take += rem;
        }
        if (take > 64) {  // a group that does not fit behind the cut: stop in front of it
            uint32_t t2 = 0;
            while (true) {
                const uint32_t it = sh.ring[(head + t2) & ring_mask];
                const uint32_t rem = (it >> 12) & 15u;
                if (t2 + rem + 1 > 64) break;
                t2 += rem + 1;
            }
            take = t2;
        }
        for (uint32_t l = 0; l < take; ++l) {
            lanes[l].active = true;
            gw_fetch<4, R>(lanes[l], sh.ring[(head + l) & ring_mask], sh.game, &sh.rec[0][0], &sh.stub[0][0], &sh.stub_node[0][0], m);
        }
        head += take;
        for (uint32_t l = 0; l < take; ++l)
            gw_visit(lanes[l], sh.game[lanes[l].g], sh.rec_owner[lanes[l].g], m, r->cfg, (const GwOutcomeTable*)nullptr);
        for (uint32_t l = 0; l < take; ++l) {
            const uint32_t g = lanes[l].g;
            finisher[l] = gw_publish<4, R>(lanes[l], sh.game[g], sh.rec[g], sh.rec_owner[g], m.spill(sh.game[g]), sh.stub[g],
                                           sh.stub_node[g], sh.ring, ring_mask, &sh.tail);
        }
        for (uint32_t l = 0; l < take; ++l)
            if (finisher[l]) {
                const uint32_t g = lanes[l].g;
                gw_finish_pick(sh.game[g], sh.stub[g], sh.ring, ring_mask, &sh.tail, g, passes + 1 < pass_limit);
            }
        passes += 1;
    }
    r->wide_passes += passes;
    r->wide_gathers += 1;
    gw_end(G, s, r->cfg);
    if (G.stalled) {
        s.status = SLOT_STALL;
        return false;
    }
    return !G.parked;  // (parked at the pass limit: the next call continues the gather)
}
extern "C" {

// A stand-in for a network in the CPU harness: priors and values that are a fixed hash of the position, so that (unlike
// uniform priors) outcomes almost never tie -- the regime the network-driven self-play runs in. tests/_hostsim.py
// (hashed_eval) states the same function for the oracle's callback backend.
static void hashed_eval(const State<4>& lf, EvalOut& o) {
    uint32_t x = (uint32_t)lf.p1 * 7919u + (uint32_t)lf.p2 * 104729u + (uint32_t)lf.turn * 1299709u;
    for (int c = 0; c < 256; ++c)
        if (st_has_cheese(lf, c)) x += (uint32_t)(c + 1) * 15485863u;
    for (int pl = 0; pl < 2; ++pl) {
        float w[5], tot = 0.0f;
        for (uint32_t a = 0; a < 5; ++a) {
            const uint32_t h = (x + a * 40503u + (uint32_t)pl * 7u) * 2654435761u;
            w[a] = (float)(1u + ((h >> 8) % 1000u));
            tot += w[a];
        }
        for (int a = 0; a < 5; ++a) (pl == 0 ? o.p1 : o.p2)[a] = w[a] / tot;
    }
    o.v1 = (float)((x * 2246822519u >> 10) % 64u) / 16.0f;
    o.v2 = (float)((x * 3266489917u >> 10) % 64u) / 16.0f;
}

// eval_mode 0: SmartUniform inline; 1: leaves stored, harness evaluates (uniform priors + constant
// values v1/v2) -- exercises the split gather / evaluate / backup path; 3: the same with the gather
// limited to 7 rounds per call (gather_machine_limited).
void* hs_run(const HsGame* g, const HsCfg* c, uint32_t n_sims, uint32_t batch, uint64_t seed, int single,
             int eval_mode, float v1, float v2, uint32_t arena_nodes) {
    HsRun* r = new HsRun();
    r->cfg = to_cfg(c, n_sims, batch);
    // scheduling knob only: vary it across the modes so every cut position of the allocation loop is exercised
    r->cfg.alloc_per_round = eval_mode == 3 ? 1 : eval_mode == 1 ? 3 : 2;
    const int hw = g->width * g->height;
    r->cost.assign(g->cost, g->cost + hw * 4);
    r->L = make_layout<4>(r->cfg, g->max_turns);
    r->scratch.assign(r->L.total, 0);
    uint32_t cap = arena_nodes ? arena_nodes : initial_arena_nodes(r->cfg);
    r->arena.assign(arena_bytes(cap) + 256, 0);
    Slot<4>& s = r->slot;
    std::memset(&s, 0, sizeof s);
    set_arena(r, cap);
    s.board.width = g->width;
    s.board.height = g->height;
    s.board.max_turns = g->max_turns;
    s.board.maze_off = 0;
    uint16_t rem = 0;
    for (int k = 0; k < 4; ++k) s.st.cheese[k] = 0;
    for (int i = 0; i < hw; ++i)
        if (g->cheese[i]) {
            s.st.cheese[i >> 6] |= 1ULL << (i & 63);
            ++rem;
        }
    s.st.remaining = rem;
    s.st.s1 = g->p1_score;
    s.st.s2 = g->p2_score;
    s.board.total_cheese = (uint16_t)(g->p1_score + g->p2_score + (float)rem);
    s.st.turn = g->turn;
    s.st.p1 = (uint8_t)(g->p1_y * g->width + g->p1_x);
    s.st.p2 = (uint8_t)(g->p2_y * g->width + g->p2_x);
    s.st.m1 = g->p1_mud;
    s.st.m2 = g->p2_mud;
    rng_seed(s.rng, seed);
    s.single_search = single;
    start_game(s, r->mem(), r->cfg);
    const int mode = eval_mode == 0 ? EVAL_UNIFORM : EVAL_STORE;
    std::vector<EvalOut> ev(batch);
    uint64_t wide_sched = (eval_mode == 5 || eval_mode == 7 || eval_mode == 9) ? (seed | 2) : 1;
    const bool hashed = eval_mode >= 6;  // 6, 7, 9: work-queue gather; 8: lane gather -- evaluations from hashed_eval
    while (s.status == SLOT_ACTIVE || s.status == SLOT_STALL || s.status == SLOT_ADVANCE) {
        if (s.status == SLOT_STALL) {
            grow(r);
            continue;
        }
        Mem<4> m = r->mem();
        if (s.status == SLOT_ADVANCE) {
            advance_tree_scalar(s, m);
            continue;
        }
        if (eval_mode == 2) {  // the fused kernel body: several batches per call, SmartUniform inline
            fused_machine(s, m, r->cfg, &g_zig, 3);
            continue;
        }
        if (eval_mode == 9) {  // the work-queue gather with one position record per game in "LDS": the others go through the scratch
            if (!wide_gather<1>(r, wide_sched, 6)) continue;  // (and a limit of six passes per call: gathers are parked and resumed)
        } else if (eval_mode == 4 || eval_mode == 5 || eval_mode == 6 || eval_mode == 7) {  // the work-queue gather (dev_gatherw.h); 5, 7: random cuts of the queue
            if (!wide_gather<4>(r, wide_sched, 0xFFFFFFFFu)) continue;
        } else if (eval_mode == 3) {  // the self-play kernel's gather: cut off every few rounds, parked, resumed
            const int got = gather_machine_limited(s, m, r->cfg, mode, 7);
            if (got != GATHER_COMPLETE) continue;
        } else if (!gather_machine(s, m, r->cfg, mode)) continue;
        const EvalOut* evp = m.ev_local;
        if (eval_mode != 0) {
            for (uint32_t j = 0; j < s.b_nn; ++j) {
                const State<4>& lf = m.leaf_local[j];
                if (hashed) {
                    hashed_eval(lf, ev[j]);
                    continue;
                }
                uniform_prior(eff_actions(m.cost, lf.p1, lf.m1), ev[j].p1);
                uniform_prior(eff_actions(m.cost, lf.p2, lf.m2), ev[j].p2);
                ev[j].v1 = v1;
                ev[j].v2 = v2;
            }
            evp = ev.data();
        }
        if (backup_machine(s, m, r->cfg, evp, &g_zig)) finish_move(s, m, r->cfg);
    }
    return r;
}
void hs_free(void* p) { delete (HsRun*)p; }

// header: [n_positions, status, error, grows, node_count, result-unused...]
void hs_header(const void* p, uint64_t out[14], float fs[2]) {
    const HsRun* r = (const HsRun*)p;
    const Slot<4>& s = r->slot;
    out[0] = s.n_pos;
    out[1] = s.status;
    out[2] = s.error;
    out[3] = r->grows;
    out[4] = s.node_count;
    out[5] = s.t_sims;
    out[6] = s.t_nn;
    out[7] = s.t_term;
    out[8] = s.t_coll;
    out[9] = s.nv_gather;
    out[10] = s.nv_backup;
    out[11] = s.new_nodes;
    out[12] = r->wide_passes;
    out[13] = r->wide_gathers;
    fs[0] = s.st.s1;
    fs[1] = s.st.s2;
}
// final position: [p1x,p1y,p2x,p2y,turn,remaining], final cheese mask
void hs_final(const void* p, int32_t out[6], uint8_t* mask) {
    const HsRun* r = (const HsRun*)p;
    const Slot<4>& s = r->slot;
    int w = s.board.width;
    out[0] = s.st.p1 % w;
    out[1] = s.st.p1 / w;
    out[2] = s.st.p2 % w;
    out[3] = s.st.p2 / w;
    out[4] = s.st.turn;
    out[5] = s.st.remaining;
    for (int i = 0; i < s.board.width * s.board.height; ++i) mask[i] = st_has_cheese(s.st, i);
}
static void fill_floats(const MoveResult& m, float* F) {
    F[2] = m.value[0];
    F[3] = m.value[1];
    std::memcpy(F + 4, m.visit_counts[0], 20);
    std::memcpy(F + 9, m.visit_counts[1], 20);
    std::memcpy(F + 14, m.prior[0], 20);
    std::memcpy(F + 19, m.prior[1], 20);
    std::memcpy(F + 24, m.policy[0], 20);
    std::memcpy(F + 29, m.policy[1], 20);
}
// same per-position layout as the oracle driver (tests/_oracle.py play_game)
void hs_positions(const void* p, int32_t* ints, float* floats, uint8_t* masks) {
    const HsRun* r = (const HsRun*)p;
    const Slot<4>& s = r->slot;
    int w = s.board.width, hw = s.board.width * s.board.height;
    for (uint32_t i = 0; i < s.n_pos; ++i) {
        const PosRec<4>& q = r->cmem().pos[i];
        int32_t* I = ints + (size_t)i * 9;
        I[0] = q.st.p1 % w;
        I[1] = q.st.p1 / w;
        I[2] = q.st.p2 % w;
        I[3] = q.st.p2 / w;
        I[4] = q.st.m1;
        I[5] = q.st.m2;
        I[6] = q.st.turn;
        I[7] = q.a1;
        I[8] = q.a2;
        float* F = floats + (size_t)i * 34;
        F[0] = q.st.s1;
        F[1] = q.st.s2;
        fill_floats(q.res, F);
        for (int c = 0; c < hw; ++c) masks[(size_t)i * hw + c] = st_has_cheese(q.st, c);
    }
}
// last search result (single-search mode): floats as positions' F[2..34), counters
void hs_last(const void* p, float* F34, uint32_t cnt[4]) {
    const HsRun* r = (const HsRun*)p;
    const MoveResult& m = r->slot.last;
    F34[0] = F34[1] = 0;
    fill_floats(m, F34);
    cnt[0] = m.total_visits;
    cnt[1] = m.nn_evals;
    cnt[2] = m.terminals;
    cnt[3] = m.collisions;
}
// canonical tree dump, same 43-word rows as the oracle's or_tree_dump
uint32_t hs_tree_dump(const void* p, uint32_t* out, uint32_t max_nodes) {
    const HsRun* r = (const HsRun*)p;
    const Slot<4>& s = r->slot;
    const Mem<4> m = r->cmem();
    struct It {
        uint32_t id, depth;
    };
    std::vector<It> st{{s.root, 0}};
    uint32_t count = 0;
    while (!st.empty()) {
        It it = st.back();
        st.pop_back();
        const NodeStats& n = m.stats[it.id];
        if (count < max_nodes) {
            uint32_t* o = out + (size_t)count * 43;
            o[0] = it.depth;
            o[1] = it.depth ? meta_po(n.h2.meta, 0) : 0;
            o[2] = it.depth ? meta_po(n.h2.meta, 1) : 0;
            o[3] = n.h0.visits;
            o[4] = n.h0.nif;
            o[5] = n.h2.terminal;
            o[6] = meta_n(n.h2.meta, 0);
            o[7] = meta_n(n.h2.meta, 1);
            o[8] = f32_to_bits(n.h0.v1);
            o[9] = f32_to_bits(n.h0.v2);
            o[10] = f32_to_bits(n.h1.scale);
            o[11] = f32_to_bits(it.depth ? n.h1.r1 : 0.0f);
            o[12] = f32_to_bits(it.depth ? n.h1.r2 : 0.0f);
            for (int i = 0; i < 5; ++i) {
                o[13 + i * 3] = f32_to_bits(n.e[0][i].prior);
                o[14 + i * 3] = f32_to_bits(n.e[0][i].q);
                o[15 + i * 3] = n.e[0][i].visits;
                o[28 + i * 3] = f32_to_bits(n.e[1][i].prior);
                o[29 + i * 3] = f32_to_bits(n.e[1][i].q);
                o[30 + i * 3] = n.e[1][i].visits;
            }
        }
        ++count;
        for (int i = 24; i >= 0; --i)
            if (m.stats[it.id].c[i] != NIL) st.push_back({m.stats[it.id].c[i], it.depth + 1});
    }
    return count;
}

}  // extern "C"
