// TEST INFRASTRUCTURE ONLY -- CPU oracle (see rng.hpp header).
//
// Restatement of the PyRat rules the reference's search and self-play call into. The engine is
// the third-party crate `pyrat-rust` 0.2.0 (git mintiti/pyrat-rust @ 8d10747, Cargo.lock:1064-1066)
// and is NOT under /root/reference, so the rules are restated from the behaviour the reference
// pins at its call sites:
//   - Direction UP0(+y) RIGHT1(+x) DOWN2(-y) LEFT3(-x) STAY4; cell index y*w+x
//       crates/alpharat-mcts-python/python/pyrat_engine/core/types.py:21-53
//   - effective_actions: blocked -> STAY, in mud -> all STAY
//       crates/alpharat-mcts-python/python/pyrat_engine/core/game.pyi:337-378
//   - scoring 1.0 solo / 0.5 shared              crates/alpharat-mcts/src/tree.rs:934-999
//   - game over: turn limit, no cheese, majority alpharat/eval/game.py:31-44
//   - entering a cost-N mud edge: position = target, mud_timer = N
//       crates/alpharat-sampling/tests/fixtures/mud_stuck_5x5.json
// PARITY UNPINNED for: the mud countdown (here: one tick per later turn, no move, no cheese
// pickup while the timer is > 0), and the random cheese sampler (here: our own, see make_cheese).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "rng.hpp"

namespace oracle {

enum : uint8_t { UP = 0, RIGHT = 1, DOWN = 2, LEFT = 3, STAY = 4 };

struct Player {
    uint8_t x = 0, y = 0;
    float score = 0.0f;
    uint8_t mud_timer = 0;
};

struct MoveUndo {
    Player p1, p2;
    uint16_t turn;
    int collected[2];  // cell indices re-inserted on undo, -1 = none
};

struct GameState {
    uint8_t width = 0, height = 0;
    uint16_t max_turns = 0, turn = 0;
    Player player1, player2;
    // cheese bitset over cells, idx = y*w+x (boards up to 256 cells)
    uint64_t cheese_bits[4] = {0, 0, 0, 0};
    uint16_t remaining_cheese = 0;
    uint16_t total_cheese = 0;
    // cost[cell*4 + dir]: 0 = wall / board edge, 1 = open passage, >= 2 = mud.
    // Immutable once play starts, so clones of a state share it (the reference clones a
    // GameState per leaf, search.rs:695).
    std::shared_ptr<std::vector<uint8_t>> maze;
    const uint8_t* cost = nullptr;

    bool has_cheese(int i) const { return (cheese_bits[i >> 6] >> (i & 63)) & 1; }
    void set_cheese(int i, bool v) {
        if (v) cheese_bits[i >> 6] |= (1ULL << (i & 63));
        else cheese_bits[i >> 6] &= ~(1ULL << (i & 63));
    }

    int cells() const { return (int)width * (int)height; }
    int idx(int x, int y) const { return y * width + x; }

    void init_open(uint8_t w, uint8_t h, uint16_t mt) {
        width = w;
        height = h;
        max_turns = mt;
        turn = 0;
        for (int k = 0; k < 4; ++k) cheese_bits[k] = 0;
        remaining_cheese = total_cheese = 0;
        maze = std::make_shared<std::vector<uint8_t>>((size_t)w * h * 4, (uint8_t)0);
        cost = maze->data();
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                uint8_t* c = &(*maze)[(size_t)idx(x, y) * 4];
                c[UP] = y + 1 < h;
                c[RIGHT] = x + 1 < w;
                c[DOWN] = y > 0;
                c[LEFT] = x > 0;
            }
        player1 = Player();
        player2 = Player();
        player2.x = w - 1;
        player2.y = h - 1;
    }

    static int dir_between(int x1, int y1, int x2, int y2) {
        if (x2 == x1 && y2 == y1 + 1) return UP;
        if (x2 == x1 + 1 && y2 == y1) return RIGHT;
        if (x2 == x1 && y2 == y1 - 1) return DOWN;
        if (x2 == x1 - 1 && y2 == y1) return LEFT;
        return -1;
    }
    // wall / mud between two adjacent cells, both directions
    bool set_edge(int x1, int y1, int x2, int y2, uint8_t value) {
        int d = dir_between(x1, y1, x2, y2);
        if (d < 0) return false;
        if (maze.use_count() > 1) {  // copy on write (construction time only)
            maze = std::make_shared<std::vector<uint8_t>>(*maze);
            cost = maze->data();
        }
        (*maze)[(size_t)idx(x1, y1) * 4 + d] = value;
        (*maze)[(size_t)idx(x2, y2) * 4 + ((d + 2) & 3)] = value;
        return true;
    }
    bool add_wall(int x1, int y1, int x2, int y2) { return set_edge(x1, y1, x2, y2, 0); }
    bool add_mud(int x1, int y1, int x2, int y2, uint8_t v) { return set_edge(x1, y1, x2, y2, v); }
    void add_cheese(int x, int y) {
        int i = idx(x, y);
        if (!has_cheese(i)) {
            set_cheese(i, true);
            ++remaining_cheese;
            ++total_cheese;
        }
    }

    float player1_score() const { return player1.score; }
    float player2_score() const { return player2.score; }

    void effective_actions(const Player& p, uint8_t out[5]) const {
        if (p.mud_timer > 0) {
            for (int a = 0; a < 5; ++a) out[a] = STAY;
            return;
        }
        const uint8_t* c = &cost[(size_t)idx(p.x, p.y) * 4];
        for (int a = 0; a < 4; ++a) out[a] = c[a] ? (uint8_t)a : (uint8_t)STAY;
        out[4] = STAY;
    }
    void effective_actions_p1(uint8_t out[5]) const { effective_actions(player1, out); }
    void effective_actions_p2(uint8_t out[5]) const { effective_actions(player2, out); }

    bool check_game_over() const {
        if (turn >= max_turns) return true;
        if (remaining_cheese == 0) return true;
        float half = (float)total_cheese / 2.0f;
        return player1.score > half || player2.score > half;
    }

    void move_player(Player& p, uint8_t d) {
        if (p.mud_timer > 0) {
            p.mud_timer -= 1;
            return;
        }
        if (d >= 4) return;
        uint8_t c = cost[(size_t)idx(p.x, p.y) * 4 + d];
        if (c == 0) return;  // blocked: counts as a miss in the engine, position unchanged
        switch (d) {
            case UP: p.y += 1; break;
            case RIGHT: p.x += 1; break;
            case DOWN: p.y -= 1; break;
            default: p.x -= 1; break;
        }
        if (c >= 2) p.mud_timer = c;
    }

    MoveUndo make_move(uint8_t d1, uint8_t d2) {
        MoveUndo u;
        u.p1 = player1;
        u.p2 = player2;
        u.turn = turn;
        u.collected[0] = u.collected[1] = -1;
        move_player(player1, d1);
        move_player(player2, d2);
        bool c1 = player1.mud_timer == 0, c2 = player2.mud_timer == 0;
        int i1 = idx(player1.x, player1.y), i2 = idx(player2.x, player2.y);
        if (c1 && c2 && i1 == i2) {
            if (has_cheese(i1)) {
                set_cheese(i1, false);
                --remaining_cheese;
                player1.score += 0.5f;
                player2.score += 0.5f;
                u.collected[0] = i1;
            }
        } else {
            if (c1 && has_cheese(i1)) {
                set_cheese(i1, false);
                --remaining_cheese;
                player1.score += 1.0f;
                u.collected[0] = i1;
            }
            if (c2 && has_cheese(i2)) {
                set_cheese(i2, false);
                --remaining_cheese;
                player2.score += 1.0f;
                u.collected[1] = i2;
            }
        }
        turn += 1;
        return u;
    }

    void unmake_move(const MoveUndo& u) {
        for (int k = 0; k < 2; ++k)
            if (u.collected[k] >= 0) {
                set_cheese(u.collected[k], true);
                ++remaining_cheese;
            }
        player1 = u.p1;
        player2 = u.p2;
        turn = u.turn;
    }
};

// Our own seeded cheese sampler (the engine's `with_random_cheese` is absent and the reference
// creates games unseeded, bindings.rs(sampling):528-532 -- nothing to match).
// Documented in DESIGN.md "game generation"; the product's host code implements the same rule
// independently. Symmetric: 180-degree pairs (i, N-1-i); the centre cell is used when the count
// is odd. Start cells never hold cheese.
inline bool make_cheese(GameState& g, uint16_t count, bool symmetric, uint64_t seed) {
    SmallRng rng = SmallRng::seed_from_u64(seed);
    const int n = g.cells();
    const int s1 = g.idx(g.player1.x, g.player1.y), s2 = g.idx(g.player2.x, g.player2.y);
    std::vector<int> cand;
    if (symmetric) {
        int centre = -1;
        for (int i = 0; i < n; ++i) {
            int j = n - 1 - i;
            if (i == s1 || i == s2 || j == s1 || j == s2) continue;
            if (i < j) cand.push_back(i);
            else if (i == j) centre = i;
        }
        int need = count;
        if (need & 1) {
            if (centre < 0) return false;
            g.add_cheese(centre % g.width, centre / g.width);
            need -= 1;
        }
        int pairs = need / 2;
        if (pairs > (int)cand.size()) return false;
        for (int i = (int)cand.size() - 1; i >= 1; --i) {
            int j = (int)rng.gen_range_u32((uint32_t)i + 1);
            int t = cand[i];
            cand[i] = cand[j];
            cand[j] = t;
        }
        for (int k = 0; k < pairs; ++k) {
            int i = cand[k], j = n - 1 - i;
            g.add_cheese(i % g.width, i / g.width);
            g.add_cheese(j % g.width, j / g.width);
        }
        return true;
    }
    for (int i = 0; i < n; ++i)
        if (i != s1 && i != s2) cand.push_back(i);
    if ((int)count > (int)cand.size()) return false;
    for (int i = (int)cand.size() - 1; i >= 1; --i) {
        int j = (int)rng.gen_range_u32((uint32_t)i + 1);
        int t = cand[i];
        cand[i] = cand[j];
        cand[j] = t;
    }
    for (int k = 0; k < count; ++k) g.add_cheese(cand[k] % g.width, cand[k] / g.width);
    return true;
}


// Random walls + mud. The reference delegates this to the pyrat-rust engine (GameBuilder::with_random_maze /
// with_classic_maze, bindings.rs:505-517), whose generator and RNG (rand 0.10) are not in the container:
// PARITY UNPINNED. This is our own generator, specified in DESIGN.md "game generation" and implemented
// twice (here and in alpharat_hip.hip::generate_maze):
//   edges in the order (cell 0..n-1: RIGHT edge, then UP edge); with symmetry an edge and its 180-degree
//   image (cells i -> n-1-i) are decided together, the one with the smaller index deciding;
//   1. every deciding edge becomes a wall when below(2^24) < (u32)(wall_density * 2^24);
//   2. the walled deciding edges are shuffled (Fisher-Yates, gen_range) and opened again, in that order,
//      whenever they (or their image) join two components, until the maze is connected;
//   3. every open deciding edge gets mud when below(2^24) < (u32)(mud_density * 2^24), cost 2 + below(2).
// Stream: SmallRng::seed_from_u64(seed ^ 0x6D617A65).
inline void make_maze(GameState& g, float wall_density, float mud_density, bool symmetric, uint64_t seed) {
    SmallRng rng = SmallRng::seed_from_u64(seed ^ 0x6D617A65ULL);
    const int w = g.width, h = g.height, n = w * h;
    struct E {
        int a, b;
    };
    std::vector<E> edges;
    for (int i = 0; i < n; ++i) {
        const int x = i % w, y = i / w;
        if (x + 1 < w) edges.push_back({i, i + 1});
        if (y + 1 < h) edges.push_back({i, i + w});
    }
    auto find_edge = [&](int a, int b) {
        for (size_t k = 0; k < edges.size(); ++k)
            if (edges[k].a == a && edges[k].b == b) return (int)k;
        return -1;
    };
    std::vector<int> image(edges.size());
    for (size_t k = 0; k < edges.size(); ++k) image[k] = symmetric ? find_edge(n - 1 - edges[k].b, n - 1 - edges[k].a) : (int)k;
    std::vector<uint8_t> value(edges.size(), 1);  // 0 wall, 1 open, >= 2 mud
    std::vector<int> deciding;
    for (size_t k = 0; k < edges.size(); ++k)
        if ((int)k <= image[k]) deciding.push_back((int)k);
    const uint32_t wall_thr = (uint32_t)(wall_density * 16777216.0f), mud_thr = (uint32_t)(mud_density * 16777216.0f);
    for (int k : deciding)
        if (rng.gen_range_u32(1u << 24) < wall_thr) value[k] = value[image[k]] = 0;
    std::vector<int> parent(n);
    for (int i = 0; i < n; ++i) parent[i] = i;
    auto root = [&](int v) {
        while (parent[v] != v) v = parent[v] = parent[parent[v]];
        return v;
    };
    int comps = n;
    auto join = [&](int a, int b) {
        a = root(a);
        b = root(b);
        if (a == b) return false;
        parent[a < b ? b : a] = a < b ? a : b;
        comps -= 1;
        return true;
    };
    for (size_t k = 0; k < edges.size(); ++k)
        if (value[k]) join(edges[k].a, edges[k].b);
    std::vector<int> walled;
    for (int k : deciding)
        if (!value[k]) walled.push_back(k);
    for (int i = (int)walled.size() - 1; i >= 1; --i) {
        const int j = (int)rng.gen_range_u32((uint32_t)i + 1);
        const int t = walled[i];
        walled[i] = walled[j];
        walled[j] = t;
    }
    for (int k : walled) {
        if (comps == 1) break;
        const int m = image[k];
        const bool split = root(edges[k].a) != root(edges[k].b) || root(edges[m].a) != root(edges[m].b);
        if (!split) continue;
        value[k] = value[m] = 1;
        join(edges[k].a, edges[k].b);
        join(edges[m].a, edges[m].b);
    }
    for (int k : deciding)
        if (value[k] && rng.gen_range_u32(1u << 24) < mud_thr) value[k] = value[image[k]] = (uint8_t)(2 + rng.gen_range_u32(2));
    for (size_t k = 0; k < edges.size(); ++k) {
        const int a = edges[k].a, b = edges[k].b;
        g.set_edge(a % w, a / w, b % w, b / w, value[k]);
    }
}

}  // namespace oracle
