// TEST INFRASTRUCTURE ONLY -- CPU oracle (see rng.hpp header).
// C entry points so tests/, smoke() and bench.py's cpu_baseline leg can drive the oracle through
// ctypes. Nothing in alpharat_amd/ links or loads this library.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mcts.hpp"
#include "nets.hpp"
#include "selfplay.hpp"

using namespace oracle;

extern "C" {

struct OrSearchConfig {
    float c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration;
    uint32_t collision_limit_min, collision_limit_max, collision_scaling_start, collision_scaling_end;
    float collision_scaling_power;
};

struct OrSearchResult {
    float policy_p1[5], policy_p2[5];
    float value_p1, value_p2;
    float visit_counts_p1[5], visit_counts_p2[5];
    float prior_p1[5], prior_p2[5];
    uint32_t total_visits, nn_evals, terminals, collisions;
};

static SearchConfig to_cfg(const OrSearchConfig* c) {
    SearchConfig s;
    s.c_puct = c->c_puct;
    s.fpu_reduction = c->fpu_reduction;
    s.force_k = c->force_k;
    s.noise_epsilon = c->noise_epsilon;
    s.noise_concentration = c->noise_concentration;
    s.collision_limit_min = c->collision_limit_min;
    s.collision_limit_max = c->collision_limit_max;
    s.collision_scaling_start = c->collision_scaling_start;
    s.collision_scaling_end = c->collision_scaling_end;
    s.collision_scaling_power = c->collision_scaling_power;
    return s;
}

static thread_local std::string g_err;
const char* or_last_error() { return g_err.c_str(); }

// ---- games ----------------------------------------------------------------------------------
void* or_game_new(uint8_t w, uint8_t h, uint16_t max_turns) {
    GameState* g = new GameState();
    g->init_open(w, h, max_turns);
    return g;
}
void* or_game_clone(const void* g) { return new GameState(*(const GameState*)g); }
void or_game_free(void* g) { delete (GameState*)g; }
void or_game_set_positions(void* gp, uint8_t x1, uint8_t y1, uint8_t x2, uint8_t y2) {
    GameState* g = (GameState*)gp;
    g->player1.x = x1;
    g->player1.y = y1;
    g->player2.x = x2;
    g->player2.y = y2;
}
int or_game_add_wall(void* g, int x1, int y1, int x2, int y2) { return ((GameState*)g)->add_wall(x1, y1, x2, y2); }
int or_game_add_mud(void* g, int x1, int y1, int x2, int y2, int v) {
    return ((GameState*)g)->add_mud(x1, y1, x2, y2, (uint8_t)v);
}
void or_game_add_cheese(void* g, int x, int y) { ((GameState*)g)->add_cheese(x, y); }
int or_game_random_cheese(void* g, uint16_t count, int symmetric, uint64_t seed) {
    return make_cheese(*(GameState*)g, count, symmetric != 0, seed);
}
void or_game_random_maze(void* g, float wall_density, float mud_density, int symmetric, uint64_t seed) {
    make_maze(*(GameState*)g, wall_density, mud_density, symmetric != 0, seed);
}
void or_game_cost(const void* gp, uint8_t* out) {  // [cells * 4]: 0 wall / edge, 1 open, >= 2 mud
    const GameState* g = (const GameState*)gp;
    std::memcpy(out, g->cost, (size_t)g->cells() * 4);
}
void or_game_make_move(void* g, uint8_t d1, uint8_t d2) { ((GameState*)g)->make_move(d1, d2); }
int or_game_over(const void* g) { return ((const GameState*)g)->check_game_over(); }
// state: [p1x,p1y,p2x,p2y,p1mud,p2mud,turn,remaining] + scores
void or_game_state(const void* gp, int32_t out_i[8], float out_f[2]) {
    const GameState* g = (const GameState*)gp;
    out_i[0] = g->player1.x;
    out_i[1] = g->player1.y;
    out_i[2] = g->player2.x;
    out_i[3] = g->player2.y;
    out_i[4] = g->player1.mud_timer;
    out_i[5] = g->player2.mud_timer;
    out_i[6] = g->turn;
    out_i[7] = g->remaining_cheese;
    out_f[0] = g->player1.score;
    out_f[1] = g->player2.score;
}
void or_game_cheese_mask(const void* gp, uint8_t* out) {
    const GameState* g = (const GameState*)gp;
    for (int i = 0; i < g->cells(); ++i) out[i] = g->has_cheese(i);
}
void or_game_maze(const void* gp, int8_t* out) {
    std::vector<int8_t> m = build_maze_array(*(const GameState*)gp);
    std::memcpy(out, m.data(), m.size());
}
void or_game_effective_actions(const void* gp, uint8_t out1[5], uint8_t out2[5]) {
    ((const GameState*)gp)->effective_actions_p1(out1);
    ((const GameState*)gp)->effective_actions_p2(out2);
}
void or_encode(const void* gp, float* out) { encode_flat(*(const GameState*)gp, out); }

// ---- nets -----------------------------------------------------------------------------------
void* or_net_load(const char* path) {
    NetBlob* nb = new NetBlob();
    if (!nb->load(path, g_err)) {
        delete nb;
        return nullptr;
    }
    return nb;
}
void or_net_free(void* n) { delete (NetBlob*)n; }
// out per sample: logits_p1[5] logits_p2[5] policy_p1[5] policy_p2[5] v1 v2  (22 floats)
int or_net_forward(const void* n, const float* obs, int count, int obs_dim, float* out) {
    for (int i = 0; i < count; ++i) {
        NetOut o;
        if (!net_forward(*(const NetBlob*)n, obs + (size_t)i * obs_dim, o, g_err)) return -1;
        float* r = out + (size_t)i * 22;
        std::memcpy(r, o.logits_p1, 20);
        std::memcpy(r + 5, o.logits_p2, 20);
        std::memcpy(r + 10, o.policy_p1, 20);
        std::memcpy(r + 15, o.policy_p2, 20);
        r[20] = o.value_p1;
        r[21] = o.value_p2;
    }
    return 0;
}

// A host evaluator supplied by the test: the reference's Backend trait object (backend.rs:75-82) as a C
// callback. The parity tests plug the product's HIP evaluator in here (through its C-ABI), so the oracle's
// search sees bit for bit the network outputs the device pipeline sees and whole-game records can be
// compared byte for byte. leaves[i] = {p1x, p1y, p2x, p2y, p1_mud, p2_mud, turn, pad, p1_score, p2_score
// (float bits), cheese mask bits 0..255 as 8 x u32}; out[i] = policy_p1[5] policy_p2[5] v1 v2.
struct OrLeaf {
    uint8_t p1x, p1y, p2x, p2y, p1_mud, p2_mud;
    uint16_t turn;
    float p1_score, p2_score;
    uint32_t cheese[8];
};
typedef int (*OrEvalFn)(void* user, const OrLeaf* leaves, uint32_t n, float* out12);
struct OrCallback {
    OrEvalFn fn;
    void* user;
};

// backend_kind: 0 SmartUniform, 1 constant values (value args), 2 net blob, 3 always failing,
// 4 host callback (`net` points to an OrCallback)
static Backend make_backend(int kind, float v1, float v2, const void* net) {
    if (kind == 0) return smart_uniform_backend();
    if (kind == 1) return smart_uniform_backend(v1, v2);
    if (kind == 4) {
        const OrCallback cb = *(const OrCallback*)net;
        return [cb](const std::vector<const GameState*>& games, std::vector<EvalResult>& out, std::string& e) {
            std::vector<OrLeaf> leaves(games.size());
            std::vector<float> res(games.size() * 12);
            for (size_t i = 0; i < games.size(); ++i) {
                const GameState& g = *games[i];
                OrLeaf& l = leaves[i];
                std::memset(&l, 0, sizeof l);
                l.p1x = g.player1.x;
                l.p1y = g.player1.y;
                l.p2x = g.player2.x;
                l.p2y = g.player2.y;
                l.p1_mud = g.player1.mud_timer;
                l.p2_mud = g.player2.mud_timer;
                l.turn = g.turn;
                l.p1_score = g.player1.score;
                l.p2_score = g.player2.score;
                for (int c = 0; c < g.cells(); ++c)
                    if (g.has_cheese(c)) l.cheese[c >> 5] |= 1u << (c & 31);
            }
            if (cb.fn(cb.user, leaves.data(), (uint32_t)leaves.size(), res.data()) != 0) {
                e = "host evaluator callback failed";
                return false;
            }
            out.resize(games.size());
            for (size_t i = 0; i < games.size(); ++i) {
                std::memcpy(out[i].policy_p1, &res[i * 12], 20);
                std::memcpy(out[i].policy_p2, &res[i * 12 + 5], 20);
                out[i].value_p1 = res[i * 12 + 10];
                out[i].value_p2 = res[i * 12 + 11];
            }
            return true;
        };
    }
    if (kind == 3)
        return [](const std::vector<const GameState*>&, std::vector<EvalResult>&, std::string& e) {
            e = "test failure";
            return false;
        };
    const NetBlob* nb = (const NetBlob*)net;
    return [nb](const std::vector<const GameState*>& games, std::vector<EvalResult>& out, std::string& e) {
        out.resize(games.size());
        std::vector<float> obs;
        for (size_t i = 0; i < games.size(); ++i) {
            obs.resize((size_t)obs_dim(games[i]->width, games[i]->height));
            encode_flat(*games[i], obs.data());
            NetOut o;
            if (!net_forward(*nb, obs.data(), o, e)) return false;
            for (int k = 0; k < 5; ++k) {
                out[i].policy_p1[k] = o.policy_p1[k];
                out[i].policy_p2[k] = o.policy_p2[k];
            }
            out[i].value_p1 = o.value_p1;
            out[i].value_p2 = o.value_p2;
            if (!std::isfinite(o.value_p1) || !std::isfinite(o.value_p2)) {
                e = "NaN/Inf in network output";
                return false;
            }
        }
        return true;
    };
}

static void fill_result(const SearchResult& r, OrSearchResult* out) {
    std::memcpy(out->policy_p1, r.policy_p1, 20);
    std::memcpy(out->policy_p2, r.policy_p2, 20);
    out->value_p1 = r.value_p1;
    out->value_p2 = r.value_p2;
    std::memcpy(out->visit_counts_p1, r.visit_counts_p1, 20);
    std::memcpy(out->visit_counts_p2, r.visit_counts_p2, 20);
    std::memcpy(out->prior_p1, r.prior_p1, 20);
    std::memcpy(out->prior_p2, r.prior_p2, 20);
    out->total_visits = r.total_visits;
    out->nn_evals = r.nn_evals;
    out->terminals = r.terminals;
    out->collisions = r.collisions;
}

// ---- trees (kept alive across calls so tests can inspect invariants / reuse) ---------------
struct OrTree {
    MCTSTree tree;
    explicit OrTree(const GameState& g) : tree(g) {}
};
void* or_tree_new(const void* game) { return new OrTree(*(const GameState*)game); }
void or_tree_free(void* t) { delete (OrTree*)t; }
int or_tree_advance(void* t, const void* game_after, uint8_t a1, uint8_t a2) {
    OrTree* tr = (OrTree*)t;
    if (tr->tree.advance_root(a1, a2)) return 1;
    tr->tree.reinit(*(const GameState*)game_after);
    return 0;
}
uint32_t or_tree_node_count(const void* t) { return ((const OrTree*)t)->tree.node_count; }
// walk: total nodes, sum n_in_flight (nodes+edges), count of nodes violating
// sum(p1 edge visits) == total_visits-1 (interior, non-terminal nodes with children)
void or_tree_check(const void* t, uint64_t out[4]) {
    const Node* root = ((const OrTree*)t)->tree.root;
    uint64_t nodes = 0, inflight = 0, bad = 0, terminal_with_children = 0;
    std::vector<const Node*> st{root};
    while (!st.empty()) {
        const Node* n = st.back();
        st.pop_back();
        ++nodes;
        inflight += n->n_in_flight;
        uint32_t s1 = 0, s2 = 0;
        for (int i = 0; i < n->p1.n_outcomes; ++i) {
            inflight += n->p1.edges[i].n_in_flight;
            s1 += n->p1.edges[i].visits;
        }
        for (int i = 0; i < n->p2.n_outcomes; ++i) {
            inflight += n->p2.edges[i].n_in_flight;
            s2 += n->p2.edges[i].visits;
        }
        if (n->first_child && n->total_visits > 0 && (s1 != n->total_visits - 1 || s2 != n->total_visits - 1)) ++bad;
        if (n->is_terminal && n->first_child) ++terminal_with_children;
        for (const Node* c = n->first_child; c; c = c->next_sibling) st.push_back(c);
    }
    out[0] = nodes;
    out[1] = inflight;
    out[2] = bad;
    out[3] = terminal_with_children;
}

// Canonical tree dump for parity with the device arena: pre-order DFS visiting children in
// increasing (po1*5+po2) order. Per node 40 u32 words:
// [depth, po1, po2, total_visits, n_in_flight, is_terminal, n1, n2, bits(v1), bits(v2),
//  bits(value_scale), bits(edge_r1), bits(edge_r2), p1: 5x(bits prior, bits q, visits) , p2: same ] (13+30=43)
uint32_t or_tree_dump(const void* t, uint32_t* out, uint32_t max_nodes) {
    const Node* root = ((const OrTree*)t)->tree.root;
    struct It {
        const Node* n;
        uint32_t depth;
    };
    std::vector<It> st{{root, 0}};
    uint32_t count = 0;
    auto bits = [](float f) {
        uint32_t b;
        std::memcpy(&b, &f, 4);
        return b;
    };
    while (!st.empty()) {
        It it = st.back();
        st.pop_back();
        if (count < max_nodes) {
            uint32_t* o = out + (size_t)count * 43;
            const Node* n = it.n;
            o[0] = it.depth;
            o[1] = it.depth ? n->po1 : 0;  // the root's link to its old parent is not observable
            o[2] = it.depth ? n->po2 : 0;
            o[3] = n->total_visits;
            o[4] = n->n_in_flight;
            o[5] = n->is_terminal;
            o[6] = n->p1.n_outcomes;
            o[7] = n->p2.n_outcomes;
            o[8] = bits(n->v1);
            o[9] = bits(n->v2);
            o[10] = bits(n->value_scale);
            o[11] = bits(it.depth ? n->edge_r1 : 0.0f);
            o[12] = bits(it.depth ? n->edge_r2 : 0.0f);
            for (int i = 0; i < 5; ++i) {
                o[13 + i * 3] = bits(n->p1.prior[i]);
                o[14 + i * 3] = bits(n->p1.edges[i].q);
                o[15 + i * 3] = n->p1.edges[i].visits;
                o[28 + i * 3] = bits(n->p2.prior[i]);
                o[29 + i * 3] = bits(n->p2.edges[i].q);
                o[30 + i * 3] = n->p2.edges[i].visits;
            }
        }
        ++count;
        // push children in decreasing slot order so they pop in increasing order
        const Node* kids[25];
        for (int i = 0; i < 25; ++i) kids[i] = nullptr;
        for (const Node* c = it.n->first_child; c; c = c->next_sibling) kids[c->po1 * 5 + c->po2] = c;
        for (int i = 24; i >= 0; --i)
            if (kids[i]) st.push_back({kids[i], it.depth + 1});
    }
    return count;
}

// rng state travels with the caller: 4 u64 words
void or_rng_seed(uint64_t seed, uint64_t state[4]) {
    SmallRng r = SmallRng::seed_from_u64(seed);
    std::memcpy(state, r.s, 32);
}
uint64_t or_rng_next_u64(uint64_t state[4]) {
    SmallRng r;
    std::memcpy(r.s, state, 32);
    uint64_t v = r.next_u64();
    std::memcpy(state, r.s, 32);
    return v;
}
uint32_t or_rng_gen_range(uint64_t state[4], uint32_t n) {
    SmallRng r;
    std::memcpy(r.s, state, 32);
    uint32_t v = r.gen_range_u32(n);
    std::memcpy(state, r.s, 32);
    return v;
}
int or_rng_weighted5(uint64_t state[4], const float w[5]) {
    SmallRng r;
    std::memcpy(r.s, state, 32);
    int v = weighted_index5_sample(w, r);
    std::memcpy(state, r.s, 32);
    return v;
}
double or_rng_gamma(uint64_t state[4], double shape) {
    SmallRng r;
    std::memcpy(r.s, state, 32);
    double v = GammaLarge::make(shape, 1.0).sample(r);
    std::memcpy(state, r.s, 32);
    return v;
}
double or_rng_normal(uint64_t state[4]) {
    SmallRng r;
    std::memcpy(r.s, state, 32);
    double v = sample_standard_normal(r);
    std::memcpy(state, r.s, 32);
    return v;
}

// run_search on a persistent tree. rng_state in/out.
int or_search(void* tree, const void* game, const OrSearchConfig* cfg, uint32_t n_sims, uint32_t batch,
              uint64_t rng_state[4], int backend_kind, float v1, float v2, const void* net, OrSearchResult* out,
              uint64_t counters[3]) {
    OrTree* tr = (OrTree*)tree;
    SmallRng rng;
    std::memcpy(rng.s, rng_state, 32);
    SearchResult r;
    SearchCounters ctr;
    g_err.clear();
    bool ok = run_search(tr->tree, *(const GameState*)game, make_backend(backend_kind, v1, v2, net), to_cfg(cfg),
                         n_sims, batch, rng, r, g_err, &ctr);
    std::memcpy(rng_state, rng.s, 32);
    if (!ok) return -1;
    fill_result(r, out);
    if (counters) {
        counters[0] = ctr.gather_node_visits;
        counters[1] = ctr.backup_node_visits;
        counters[2] = ctr.new_nodes;
    }
    return 0;
}

// ---- play_game ------------------------------------------------------------------------------
void* or_play_game(const void* game, const OrSearchConfig* cfg, uint32_t n_sims, uint32_t batch, uint64_t rng_seed,
                   int backend_kind, const void* net, uint32_t game_index) {
    GameRecord* rec = new GameRecord();
    SmallRng rng = SmallRng::seed_from_u64(rng_seed);
    g_err.clear();
    if (!play_game(*(const GameState*)game, make_backend(backend_kind, 0, 0, net), to_cfg(cfg), n_sims, batch, rng,
                   game_index, *rec, g_err)) {
        delete rec;
        return nullptr;
    }
    return rec;
}
void or_record_free(void* r) { delete (GameRecord*)r; }
// header: [n_positions, width, height, max_turns, result, cheese_available, game_index]
// sums: [total_simulations, nn_evals, terminals, collisions, gather_node_visits, backup_node_visits, new_nodes]
void or_record_header(const void* rp, int32_t hdr[7], uint64_t sums[7], float final_scores[2]) {
    const GameRecord* r = (const GameRecord*)rp;
    hdr[0] = (int32_t)r->positions.size();
    hdr[1] = r->width;
    hdr[2] = r->height;
    hdr[3] = r->max_turns;
    hdr[4] = r->result;
    hdr[5] = r->cheese_available;
    hdr[6] = (int32_t)r->game_index;
    sums[0] = r->total_simulations;
    sums[1] = r->total_nn_evals;
    sums[2] = r->total_terminals;
    sums[3] = r->total_collisions;
    sums[4] = r->counters.gather_node_visits;
    sums[5] = r->counters.backup_node_visits;
    sums[6] = r->counters.new_nodes;
    final_scores[0] = r->final_p1_score;
    final_scores[1] = r->final_p2_score;
}
void or_record_game_arrays(const void* rp, int8_t* maze, uint8_t* initial_cheese, uint8_t* cheese_outcomes) {
    const GameRecord* r = (const GameRecord*)rp;
    std::memcpy(maze, r->maze.data(), r->maze.size());
    std::memcpy(initial_cheese, r->initial_cheese.data(), r->initial_cheese.size());
    std::memcpy(cheese_outcomes, r->cheese_outcomes.data(), r->cheese_outcomes.size());
}
// per position: ints [p1x,p1y,p2x,p2y,p1mud,p2mud,turn,a1,a2] (9), floats [p1_score,p2_score,value_p1,value_p2,
// visit_p1[5],visit_p2[5],prior_p1[5],prior_p2[5],policy_p1[5],policy_p2[5]] (34), cheese_mask[hw]
void or_record_positions(const void* rp, int32_t* ints, float* floats, uint8_t* masks) {
    const GameRecord* r = (const GameRecord*)rp;
    size_t hw = (size_t)r->width * r->height;
    for (size_t i = 0; i < r->positions.size(); ++i) {
        const PositionRecord& p = r->positions[i];
        int32_t* I = ints + i * 9;
        I[0] = p.p1_pos[0];
        I[1] = p.p1_pos[1];
        I[2] = p.p2_pos[0];
        I[3] = p.p2_pos[1];
        I[4] = p.p1_mud;
        I[5] = p.p2_mud;
        I[6] = p.turn;
        I[7] = p.action_p1;
        I[8] = p.action_p2;
        float* F = floats + i * 34;
        F[0] = p.p1_score;
        F[1] = p.p2_score;
        F[2] = p.value_p1;
        F[3] = p.value_p2;
        std::memcpy(F + 4, p.visit_counts_p1, 20);
        std::memcpy(F + 9, p.visit_counts_p2, 20);
        std::memcpy(F + 14, p.prior_p1, 20);
        std::memcpy(F + 19, p.prior_p2, 20);
        std::memcpy(F + 24, p.policy_p1, 20);
        std::memcpy(F + 29, p.policy_p2, 20);
        std::memcpy(masks + i * hw, p.cheese_mask.data(), hw);
    }
}

// ---- CPU baseline: the reference's worker-thread structure (selfplay.rs:609-703): one game per
// OS thread claimed by atomic index. Game i: cheese seed game_seed_base+i, search rng seed
// rng_seed_base+i (the reference seeds from entropy; fixed seeds here so runs are repeatable).
// `max_secs` > 0 bounds the sample: after that long no thread claims another game (the games in progress
// are finished). Every thread notes when its own last game ended, so the caller can add up per-thread rates
// (work of thread t / busy time of thread t) instead of dividing by the time of the slowest thread.
// out: [games, positions, simulations, nn_evals, terminals, collisions, gather_nv, backup_nv, new_nodes]
// thread_secs (optional, [threads]) and thread_sims (optional, [threads]): busy time and simulations per thread
int or_selfplay_bench(uint8_t w, uint8_t h, uint16_t cheese, uint16_t max_turns, uint32_t n_games,
                      const OrSearchConfig* cfg, uint32_t n_sims, uint32_t batch, uint32_t threads,
                      uint64_t game_seed_base, uint64_t rng_seed_base, int backend_kind, const void* net,
                      uint64_t out[9], double* elapsed_secs, double max_secs, double* thread_secs,
                      uint64_t* thread_sims) {
    std::atomic<uint32_t> next{0};
    std::atomic<int> failed{0};
    std::vector<std::vector<uint64_t>> acc(threads, std::vector<uint64_t>(9, 0));
    std::vector<double> busy(threads, 0.0);
    SearchConfig sc = to_cfg(cfg);
    auto t0 = std::chrono::steady_clock::now();
    auto secs = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            Backend be = make_backend(backend_kind, 0, 0, net);
            std::string err;
            for (;;) {
                if (max_secs > 0.0 && secs() >= max_secs) break;
                uint32_t i = next.fetch_add(1);
                if (i >= n_games) break;
                GameState g;
                g.init_open(w, h, max_turns);
                if (!make_cheese(g, cheese, true, game_seed_base + i)) {
                    failed = 1;
                    break;
                }
                SmallRng rng = SmallRng::seed_from_u64(rng_seed_base + i);
                GameRecord rec;
                if (!play_game(g, be, sc, n_sims, batch, rng, i, rec, err)) {
                    failed = 1;
                    break;
                }
                uint64_t* a = acc[t].data();
                a[0] += 1;
                a[1] += rec.positions.size();
                a[2] += rec.total_simulations;
                a[3] += rec.total_nn_evals;
                a[4] += rec.total_terminals;
                a[5] += rec.total_collisions;
                a[6] += rec.counters.gather_node_visits;
                a[7] += rec.counters.backup_node_visits;
                a[8] += rec.counters.new_nodes;
                busy[t] = secs();
            }
        });
    for (auto& th : pool) th.join();
    *elapsed_secs = secs();
    for (int k = 0; k < 9; ++k) {
        out[k] = 0;
        for (uint32_t t = 0; t < threads; ++t) out[k] += acc[t][k];
    }
    for (uint32_t t = 0; t < threads; ++t) {
        if (thread_secs) thread_secs[t] = busy[t];
        if (thread_sims) thread_sims[t] = acc[t][2];
    }
    return failed ? -1 : 0;
}

}  // extern "C"
