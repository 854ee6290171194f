// TEST INFRASTRUCTURE ONLY -- CPU oracle (see rng.hpp header).
//
// Restatement of the per-game self-play loop and its records:
//   crates/alpharat-sampling/src/selfplay.rs:374-409  build_maze_array / build_cheese_mask
//   crates/alpharat-sampling/src/selfplay.rs:415-471  compute_cheese_outcomes
//   crates/alpharat-sampling/src/selfplay.rs:474-479  sample_action
//   crates/alpharat-sampling/src/selfplay.rs:486-512  record_position
//   crates/alpharat-sampling/src/selfplay.rs:515-598  play_game
//   crates/alpharat-sampling/src/flat_encoder.rs:52-125  FlatEncoder::encode_into
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mcts.hpp"

namespace oracle {

struct PositionRecord {
    uint8_t p1_pos[2], p2_pos[2];
    float p1_score, p2_score;
    uint8_t p1_mud, p2_mud;
    uint16_t turn;
    std::vector<uint8_t> cheese_mask;
    float value_p1, value_p2;
    float visit_counts_p1[5], visit_counts_p2[5];
    float prior_p1[5], prior_p2[5];
    float policy_p1[5], policy_p2[5];
    uint8_t action_p1, action_p2;
};

struct GameRecord {
    uint8_t width, height;
    uint16_t max_turns;
    std::vector<int8_t> maze;
    std::vector<uint8_t> initial_cheese;
    std::vector<PositionRecord> positions;
    float final_p1_score, final_p2_score;
    uint8_t result;  // 0 draw, 1 P1, 2 P2 (selfplay.rs:62-66)
    uint64_t total_simulations;
    uint16_t cheese_available;
    uint32_t game_index;
    std::vector<uint8_t> cheese_outcomes;  // P1Win 0 / Simultaneous 1 / Uncollected 2 / P2Win 3
    uint64_t total_nn_evals, total_terminals, total_collisions;
    SearchCounters counters;
};

// selfplay.rs:374-392
inline std::vector<int8_t> build_maze_array(const GameState& g) {
    int n = g.cells();
    std::vector<int8_t> maze((size_t)n * 4, (int8_t)-1);
    for (int c = 0; c < n; ++c)
        for (int d = 0; d < 4; ++d) {
            uint8_t v = g.cost[(size_t)c * 4 + d];
            if (v) maze[(size_t)c * 4 + d] = (int8_t)v;
        }
    return maze;
}

// selfplay.rs:395-409
inline std::vector<uint8_t> build_cheese_mask(const GameState& g) {
    int n = g.cells();
    std::vector<uint8_t> m((size_t)n, 0);
    for (int i = 0; i < n; ++i) m[i] = g.has_cheese(i) ? 1 : 0;
    return m;
}

// selfplay.rs:474-479
inline uint8_t sample_action(const float policy[5], SmallRng& rng) {
    int i = weighted_index5_sample(policy, rng);
    return i < 0 ? (uint8_t)4 : (uint8_t)i;
}

// selfplay.rs:415-471
inline std::vector<uint8_t> compute_cheese_outcomes(const std::vector<PositionRecord>& positions,
                                                    const GameState& game) {
    int w = game.width;
    size_t size = (size_t)game.cells();
    std::vector<uint8_t> outcomes(size, (uint8_t)2);
    size_t n = positions.size();
    std::vector<uint8_t> final_mask = build_cheese_mask(game);
    for (size_t i = 0; i < n; ++i) {
        const std::vector<uint8_t>& cur = positions[i].cheese_mask;
        const std::vector<uint8_t>& next = (i + 1 < n) ? positions[i + 1].cheese_mask : final_mask;
        uint8_t np1[2], np2[2];
        if (i + 1 < n) {
            np1[0] = positions[i + 1].p1_pos[0];
            np1[1] = positions[i + 1].p1_pos[1];
            np2[0] = positions[i + 1].p2_pos[0];
            np2[1] = positions[i + 1].p2_pos[1];
        } else {
            np1[0] = game.player1.x;
            np1[1] = game.player1.y;
            np2[0] = game.player2.x;
            np2[1] = game.player2.y;
        }
        for (size_t idx = 0; idx < size; ++idx) {
            if (cur[idx] == 1 && next[idx] == 0) {
                uint8_t x = (uint8_t)(idx % w), y = (uint8_t)(idx / w);
                bool p1_there = np1[0] == x && np1[1] == y;
                bool p2_there = np2[0] == x && np2[1] == y;
                outcomes[idx] = (p1_there && p2_there) ? 1 : p1_there ? 0 : p2_there ? 3 : 2;
            }
        }
    }
    return outcomes;
}

// selfplay.rs:515-598
inline bool play_game(GameState game, const Backend& backend, const SearchConfig& cfg, uint32_t n_sims,
                      uint32_t batch_size, SmallRng& rng, uint32_t game_index, GameRecord& rec,
                      std::string& err) {
    rec.width = game.width;
    rec.height = game.height;
    rec.max_turns = game.max_turns;
    rec.maze = build_maze_array(game);
    rec.initial_cheese = build_cheese_mask(game);
    rec.cheese_available = game.remaining_cheese;
    rec.positions.clear();
    rec.total_simulations = rec.total_nn_evals = rec.total_terminals = rec.total_collisions = 0;
    rec.counters = SearchCounters();
    MCTSTree tree(game);
    while (!game.check_game_over()) {
        SearchResult r;
        if (!run_search(tree, game, backend, cfg, n_sims, batch_size, rng, r, err, &rec.counters)) return false;
        rec.total_simulations += r.total_visits;
        rec.total_nn_evals += r.nn_evals;
        rec.total_terminals += r.terminals;
        rec.total_collisions += r.collisions;
        uint8_t a1 = sample_action(r.policy_p1, rng);
        uint8_t a2 = sample_action(r.policy_p2, rng);
        PositionRecord p;
        p.p1_pos[0] = game.player1.x;
        p.p1_pos[1] = game.player1.y;
        p.p2_pos[0] = game.player2.x;
        p.p2_pos[1] = game.player2.y;
        p.p1_score = game.player1.score;
        p.p2_score = game.player2.score;
        p.p1_mud = game.player1.mud_timer;
        p.p2_mud = game.player2.mud_timer;
        p.turn = game.turn;
        p.cheese_mask = build_cheese_mask(game);
        p.value_p1 = r.value_p1;
        p.value_p2 = r.value_p2;
        for (int i = 0; i < 5; ++i) {
            p.visit_counts_p1[i] = r.visit_counts_p1[i];
            p.visit_counts_p2[i] = r.visit_counts_p2[i];
            p.prior_p1[i] = r.prior_p1[i];
            p.prior_p2[i] = r.prior_p2[i];
            p.policy_p1[i] = r.policy_p1[i];
            p.policy_p2[i] = r.policy_p2[i];
        }
        p.action_p1 = a1;
        p.action_p2 = a2;
        rec.positions.push_back(std::move(p));
        game.make_move(a1, a2);
        if (!tree.advance_root(a1, a2)) tree.reinit(game);
    }
    rec.final_p1_score = game.player1.score;
    rec.final_p2_score = game.player2.score;
    rec.result = rec.final_p1_score > rec.final_p2_score ? 1 : rec.final_p2_score > rec.final_p1_score ? 2 : 0;
    rec.cheese_outcomes = compute_cheese_outcomes(rec.positions, game);
    rec.game_index = game_index;
    return true;
}

// flat_encoder.rs:52-125
inline int obs_dim(int w, int h) { return w * h * 7 + 6; }
inline void encode_flat(const GameState& g, float* out) {
    const int spatial = g.cells();
    for (int i = 0; i < spatial * 4; ++i) {
        uint8_t c = g.cost[i];
        out[i] = c ? (float)c / 10.0f : -1.0f;
    }
    float* p1 = out + spatial * 4;
    float* p2 = out + spatial * 5;
    float* ch = out + spatial * 6;
    for (int j = 0; j < spatial; ++j) {
        p1[j] = 0.0f;
        p2[j] = 0.0f;
        ch[j] = g.has_cheese(j) ? 1.0f : 0.0f;
    }
    p1[g.idx(g.player1.x, g.player1.y)] = 1.0f;
    p2[g.idx(g.player2.x, g.player2.y)] = 1.0f;
    float* s = out + spatial * 7;
    float s1 = g.player1.score, s2 = g.player2.score;
    s[0] = s1 - s2;
    s[1] = g.max_turns > 0 ? (float)g.turn / (float)g.max_turns : 0.0f;
    s[2] = (float)g.player1.mud_timer / 10.0f;
    s[3] = (float)g.player2.mud_timer / 10.0f;
    s[4] = s1 / 10.0f;
    s[5] = s2 / 10.0f;
}

}  // namespace oracle
