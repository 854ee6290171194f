// Per-slot scratch carving shared by the HIP runtime and the CPU logic harness under tests/.
#pragma once
#include <stddef.h>

#include "dev_search.h"

namespace ar {

struct SlotLayout {
    size_t proc_off, coll_off, levels_off, frames_off, ev_off, leaf_off, pos_off, total;
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
    off = align_up(off + sizeof(Level<NW>) * L.max_depth, 64);
    L.frames_off = off;
    off = align_up(off + sizeof(CopyFrame) * L.max_depth, 64);
    L.ev_off = off;
    off = align_up(off + sizeof(EvalOut) * cfg.batch_size, 64);
    L.leaf_off = off;
    off = align_up(off + sizeof(State<NW>) * cfg.batch_size, 64);
    L.pos_off = off;
    off = align_up(off + sizeof(PosRec<NW>) * (max_turns > 0 ? max_turns : 1), 256);
    L.total = off;
    return L;
}

template <int NW>
AR_HD void bind_scratch(Slot<NW>& s, unsigned char* base, const SlotLayout& L) {
    s.proc = (ProcEntry*)(base + L.proc_off);
    s.coll = (CollEntry*)(base + L.coll_off);
    s.levels = (Level<NW>*)(base + L.levels_off);
    s.frames = (CopyFrame*)(base + L.frames_off);
    s.ev_local = (EvalOut*)(base + L.ev_off);
    s.leaf_local = (State<NW>*)(base + L.leaf_off);
    s.pos = (PosRec<NW>*)(base + L.pos_off);
    s.coll_cap = L.coll_cap;
    s.max_depth = L.max_depth;
}

inline uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
// first arena of a game: room for the kept subtree plus one search, twice over
inline uint32_t initial_arena_nodes(const SearchCfg& cfg) {
    uint32_t want = 2 * (cfg.n_sims + 2 * cfg.batch_size) + 64;
    uint32_t p = next_pow2(want);
    return p < 256 ? 256 : p;
}

}  // namespace ar
