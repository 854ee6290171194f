// Per-slot scratch carving shared by the HIP runtime and the CPU logic harness under tests/.
#pragma once
#include <stddef.h>

#include "dev_search.h"

namespace ar {

struct SlotLayout {
    size_t proc_off, coll_off, levels_off, ev_off, leaf_off, pos_off, glane_off, total;
    uint32_t coll_cap, max_depth;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <int NW>
inline SlotLayout make_layout(const SearchCfg& cfg, uint32_t max_turns) {
    SlotLayout L;
    L.coll_cap = cfg.coll_max + cfg.batch_size + 1;
    L.max_depth = max_turns + 2;
    size_t off = 0;
    L.proc_off = off;
    off = align_up(off + sizeof(ProcEntry) * cfg.batch_size, 64);
    L.coll_off = off;
    off = align_up(off + sizeof(CollEntry) * L.coll_cap, 64);
    L.levels_off = off;
    // (the octet gather's level record, dev_gather8.h LevelO: header 16 B + 28 visit words + position, 16-byte
    // aligned, is the larger one)
    const size_t level_o = align_up(16 + 112 + sizeof(State<NW>), 16);
    const size_t level_bytes = sizeof(Level<NW>) > level_o ? sizeof(Level<NW>) : level_o;
    // (the work-queue gather, dev_gatherw.h, keeps up to sixteen spilled position records here instead: State + 16 bytes each)
    const size_t spill_bytes = 16 * align_up(sizeof(State<NW>) + 16, 8);
    const size_t levels_total = level_bytes * L.max_depth > spill_bytes ? level_bytes * L.max_depth : spill_bytes;
    off = align_up(off + levels_total, 64);
    L.ev_off = off;
    off = align_up(off + sizeof(EvalOut) * cfg.batch_size, 64);
    L.leaf_off = off;
    off = align_up(off + sizeof(State<NW>) * cfg.batch_size, 64);
    L.pos_off = off;
    off = align_up(off + sizeof(PosRec<NW>) * (max_turns > 0 ? max_turns : 1), 64);
    L.glane_off = off;
    // (a parked gather: the lane kernel's GatherLane)
    off = align_up(off + sizeof(GatherLane<NW>), 256);
    L.total = off;
    return L;
}

// All of one game's addresses from the three bases the kernels receive as arguments
// (arena_base, scratch_base, maze_pool): the pointers are derived from kernel arguments, never
// loaded from memory, so the compiler addresses them as global memory.
template <int NW>
AR_HD Mem<NW> resolve_mem(const Slot<NW>& s, unsigned char* arena_base, unsigned char* scratch_base, uint32_t slot_id,
                          const SlotLayout& L, const uint8_t* maze_pool) {
    Mem<NW> m;
    m.stats = (NodeStats*)(arena_base + s.stats_off);
    m.fwd = (uint32_t*)(arena_base + s.fwd_off);
    unsigned char* base = scratch_base + (size_t)slot_id * L.total;
    m.proc = (ProcEntry*)(base + L.proc_off);
    m.coll = (CollEntry*)(base + L.coll_off);
    m.levels = (Level<NW>*)(base + L.levels_off);
    m.ev_local = (EvalOut*)(base + L.ev_off);
    m.leaf_local = (State<NW>*)(base + L.leaf_off);
    m.pos = (PosRec<NW>*)(base + L.pos_off);
    m.glane = (void*)(base + L.glane_off);
    m.cost = maze_pool + s.board.maze_off;
    m.coll_cap = L.coll_cap;
    m.max_depth = L.max_depth;
    return m;
}

inline uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
// arena of a fresh game (the smallest size class): the root plus one full search, as k_advance sizes it (arena_need),
// in whole groups of 64 nodes (an arena is then a whole number of 256-byte units)
inline uint32_t initial_arena_nodes(const SearchCfg& cfg) {
    const uint32_t want = 1 + cfg.n_sims + 2 * cfg.batch_size + 64;
    const uint32_t p = (want + 63) / 64 * 64;
    return p < 256 ? 256 : p;
}
// bytes of one arena of `cap` nodes: [node records | fwd]
// (rounded up to a whole number of 64-byte lines, so consecutive arenas keep the records' alignment)
AR_HD size_t arena_bytes(uint32_t cap) {
    return ((size_t)cap * (sizeof(NodeStats) + sizeof(uint32_t)) + 63) & ~(size_t)63;
}

}  // namespace ar
