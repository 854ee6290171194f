// PyRat position and step on the device. Replaces the calls the reference's search makes into
// pyrat::GameState (make_move / unmake_move / effective_actions_p{1,2} / check_game_over /
// scores / cheese; call sites search.rs:586,596,665-667,678,706,723,732, tree.rs:91-92,132-136).
//
// A position is a small POD: cheese is a bitmask over cells (idx = y*w + x), players are cell
// indices, so the step is bit-ops and the DFS "unmake" is restoring a saved copy.
// NW = number of 64-bit cheese words (1 for boards up to 64 cells: 5x5, 7x7; 4 up to 256 cells).
#pragma once
#include "dev_rng.h"

namespace ar {

enum { DIR_UP = 0, DIR_RIGHT = 1, DIR_DOWN = 2, DIR_LEFT = 3, DIR_STAY = 4 };

// Board constants shared by every position of one game (kept per game slot).
struct Board {
    uint16_t width, height;
    uint16_t max_turns;
    uint16_t total_cheese;  // for the majority rule: p1 + p2 + remaining
    uint32_t maze_off;      // byte offset of this game's cost table in the maze pool
};

template <int NW>
struct State {
    uint64_t cheese[NW];
    float s1, s2;
    uint16_t turn;
    uint16_t remaining;
    uint8_t p1, p2;  // cell indices
    uint8_t m1, m2;  // mud timers
};

// (the word is picked with selects, never a dynamic index: that would push the lane state into scratch memory)
template <int NW>
AR_HD bool st_has_cheese(const State<NW>& s, int cell) {
    uint64_t word = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) word |= (NW == 1 || (cell >> 6) == w) ? s.cheese[w] : 0ULL;
    return (word >> (cell & 63)) & 1ULL;
}
template <int NW>
AR_HD void st_take_cheese(State<NW>& s, int cell) {
    const uint64_t bit = 1ULL << (cell & 63);
#pragma unroll
    for (int w = 0; w < NW; ++w) s.cheese[w] &= (cell >> 6) == w ? ~bit : ~0ULL;
    s.remaining -= 1;
}

AR_HD bool board_over(const Board& b, uint16_t turn, uint16_t remaining, float s1, float s2) {
    if (turn >= b.max_turns) return true;
    if (remaining == 0) return true;
    const float half = (float)b.total_cheese / 2.0f;
    return s1 > half || s2 > half;
}
template <int NW>
AR_HD bool st_over(const Board& b, const State<NW>& s) {
    return board_over(b, s.turn, s.remaining, s.s1, s.s2);
}

// effective action map packed 3 bits per action (action a -> bits [3a, 3a+3))
// the four direction costs of a cell in one 32-bit load (the table is 4-byte aligned per cell)
AR_HD uint32_t cell_costs(const uint8_t* cost, uint32_t cell) { return ((const uint32_t*)cost)[cell]; }

AR_HD uint32_t eff_actions(const uint8_t* cost, uint8_t cell, uint8_t mud) {
    if (mud > 0) return 4u | (4u << 3) | (4u << 6) | (4u << 9) | (4u << 12);
    const uint32_t c = cell_costs(cost, cell);
    uint32_t e = 4u << 12;
    e |= ((c & 0xffu) ? 0u : 4u);
    e |= (((c >> 8) & 0xffu) ? 1u : 4u) << 3;
    e |= (((c >> 16) & 0xffu) ? 2u : 4u) << 6;
    e |= (((c >> 24) & 0xffu) ? 3u : 4u) << 9;
    return e;
}

AR_HD void move_one(const uint8_t* cost, uint32_t width, uint8_t& cell, uint8_t& mud, uint32_t dir) {
    if (mud > 0) {
        mud -= 1;
        return;
    }
    if (dir >= 4u) return;
    const uint8_t c = (uint8_t)((cell_costs(cost, cell) >> (8u * dir)) & 0xffu);
    if (c == 0) return;
    const int delta = dir == DIR_UP ? (int)width : dir == DIR_RIGHT ? 1 : dir == DIR_DOWN ? -(int)width : -1;
    cell = (uint8_t)((int)cell + delta);
    if (c >= 2) mud = c;
}

// One simultaneous move. Rewards are the score deltas (tree.rs:89-94 compute_rewards).
template <int NW>
AR_HD void st_step(const Board& b, const uint8_t* cost, State<NW>& s, uint32_t d1, uint32_t d2, float& r1,
                   float& r2) {
    move_one(cost, b.width, s.p1, s.m1, d1);
    move_one(cost, b.width, s.p2, s.m2, d2);
    r1 = 0.0f;
    r2 = 0.0f;
    const bool free1 = s.m1 == 0, free2 = s.m2 == 0;
    if (free1 && free2 && s.p1 == s.p2) {
        if (st_has_cheese(s, s.p1)) {
            st_take_cheese(s, s.p1);
            r1 = 0.5f;
            r2 = 0.5f;
        }
    } else {
        if (free1 && st_has_cheese(s, s.p1)) {
            st_take_cheese(s, s.p1);
            r1 = 1.0f;
        }
        if (free2 && st_has_cheese(s, s.p2)) {
            st_take_cheese(s, s.p2);
            r2 = 1.0f;
        }
    }
    // score after = score before + delta; the reference computes the delta as (after - before),
    // which is exact for these half-integer scores, so carrying the delta directly is identical
    s.s1 += r1;
    s.s2 += r2;
    s.turn += 1;
}

}  // namespace ar
