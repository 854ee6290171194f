// The backup (search.rs:1027-1066: populate / finalize the gathered leaves, walk the values up, cancel the
// collisions) with SIXTEEN LANES PER GAME: lane w of a group walks batch entry w, a wavefront holds four games.
//
// The lane-per-game backup (backup_round, dev_search.h) is one dependent memory trip per tree level per entry:
// ~16 entries x ~10 levels = ~170 trips per batch in sequence, and nothing else to run meanwhile. Entries cannot
// simply be walked at the same time: paths share their upper part, and a shared node's running means must take the
// entries' values in batch order (f32 Welford updates do not commute). Here the walks overlap as far as that order
// allows:
//   1. every lane follows the parent links of its own entry from the leaf to the root and notes the path in LDS
//      (node, which outcome pair of it leads down) -- reads only, all entries at once: depth trips instead of 16 x depth;
//   2. the updates then run as a pipeline: lane w starts late enough that it reaches every depth strictly after lane
//      w - 1 did (end round E_w = max(E_{w-1} + 1, D_w), D = path length), so two entries never meet at a node in the same
//      round and always arrive in batch order; a round is: load the node's header and the two edges, apply
//      finalize_score_update / update_multivisit exactly as backup_round does, store, workgroup-scope fence (the four
//      games of a block share one wavefront: its L1 is coherent for it);
//      the leaf itself is level 0 of the path (populate_node + first finalize), so a terminal node that was claimed
//      twice in one batch is also served in order;
//   3. collisions only subtract integers from in-flight counters: lanes take them 16 at a time and walk up with atomic
//      subtractions, after the updates are done.
// Same arithmetic, same order per node as backup_round: trees are bit-identical. End-of-batch bookkeeping (batch_end)
// is done by lane 0; the end of a move (extraction, sampling, record, step: finish_move) is left to k_finish.
#pragma once
#include "dev_search.h"

#if defined(__HIPCC__)
namespace ar {

// one step of a path: node id (26 bits) | outcome pair of THIS node that leads to the level below (p1 3 bits, p2 3 bits)
typedef uint32_t PathStep;
enum { PATH_LDS_STEPS = 20 };  // steps kept in LDS per lane; deeper ones go to the game's (idle) level-stack scratch
__device__ inline PathStep path_pack(uint32_t node, uint32_t o1, uint32_t o2) { return node | (o1 << 26) | (o2 << 29); }

__device__ inline uint32_t grp_pick(uint32_t v, uint32_t w) {  // value of group lane `w` (16 lanes per game)
    return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 48u) | w), 64);
}

// One game's batch. `path` = this lane's LDS path buffer [path_cap]; all 16 lanes of the group call this together
// (idle groups call it with active = false and only take part in the wave-wide loops' votes).
// A batch that evaluates the ROOT (a fresh tree's first batch: one entry) draws Dirichlet noise (search.rs:1036-1050,
// the Gamma sampler is a hundred registers of code): such games are left to the lane-per-game kernel, which the host
// launches behind this one for whatever still has a batch to back up.
template <int NW>
__device__ inline bool backup16_wants(const Slot<NW>& S, const Mem<NW>& m, const SearchCfg& cfg) {
    if (!(cfg.noise_epsilon > 0.0f)) return true;
    const ProcEntry pe = m.proc[0];
    return !(S.n_proc >= 1 && pe.node == S.root && proc_kind(pe.kind) == PROC_EVAL);
}

template <int NW>
__device__ inline void backup16(bool active, Slot<NW>& S, const Mem<NW>& m, const SearchCfg& cfg, const EvalOut* ev,
                                PathStep* lds_path, uint32_t path_cap, uint32_t w) {
    const uint32_t n_proc = active ? S.n_proc : 0u, n_coll = active ? S.n_coll : 0u;
    // steps beyond the LDS part: lane w's slice of the level-stack scratch (not in use between gather and gather)
    PathStep* deep_path = (PathStep*)m.levels + (size_t)w * path_cap;
    auto path_at = [&](uint32_t d) -> PathStep& { return d < PATH_LDS_STEPS ? lds_path[d] : deep_path[d - PATH_LDS_STEPS]; };
    uint32_t err = 0;
    uint32_t nv = 0;  // node records updated (the lane kernel's nv_backup)
    // eval index of entry e = number of EVAL entries before it: prefix over the proc list (kinds are read by all)
    for (uint32_t base = 0; __any(base < n_proc); base += 16) {
        const uint32_t e = base + w;
        const bool mine = e < n_proc;
        // ---- 1. the path of entry e
        uint32_t kind = PROC_TERMINAL, D = 0, leaf = NIL, ev_idx = 0;
        {
            // lane w takes the w-th entry in backup order: rank by key among the chunk's entries (a chunk of the
            // depth-first gathers' entries is in order already; the work-queue gather writes at most 16, in arrival order)
            ProcEntry pe;
            pe.node = NIL;
            pe.kind = 0xFFFF0000u;
            if (mine) pe = m.proc[e];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < 16; ++j) {
                const uint32_t kj = grp_pick(pe.kind, j) >> 16;
                rank += (kj < (pe.kind >> 16) || (kj == (pe.kind >> 16) && j < w)) ? 1u : 0u;
            }
            uint32_t src = 0;
            for (uint32_t j = 0; j < 16; ++j) src = grp_pick(rank, j) == w ? j : src;
            pe.node = grp_pick(pe.node, src);
            pe.kind = grp_pick(pe.kind, src);
            if (mine) {  // (entries without a batch entry sort last: lanes >= n_proc - base get those)
                kind = proc_kind(pe.kind);
                leaf = pe.node;
                ev_idx = proc_eval_index(pe.kind);  // where the evaluator put this leaf's outputs
            }
        }
        {
            uint32_t cur = leaf, po = 0;
            bool walking = mine;
            while (__any(walking)) {
                if (walking) {
                    if (D >= path_cap) {
                        err = 6;
                        walking = false;
                    } else {
                        const uint32_t parent = m.stats[cur].h1.parent;
                        const uint32_t meta = m.stats[cur].h2.meta;
                        path_at(D) = path_pack(cur, po & 7u, po >> 3);
                        D += 1;
                        po = meta_po(meta, 0) | (meta_po(meta, 1) << 3);
                        cur = parent;
                        if (cur == NIL) walking = false;
                    }
                }
            }
        }
        // ---- 2. the schedule: E_w = max(E_{w-1} + 1, D_w) over the lanes that have an entry, in lane order
        uint32_t E = 0, t0 = 0;
        {
            uint32_t prev = 0;
            for (uint32_t j = 0; j < 16; ++j) {
                const uint32_t Dj = grp_pick(D, j);
                uint32_t Ej = prev;
                if (Dj > 0) {
                    Ej = prev + 1 > Dj ? prev + 1 : Dj;
                    prev = Ej;
                }
                if (j == w) {
                    E = Ej;
                    t0 = Ej - Dj;
                }
            }
            (void)E;
        }
        // ---- the pipeline: at round r lane w works on path[r - t0] while t0 <= r < t0 + D
        float v1 = 0.0f, v2 = 0.0f, cr1 = 0.0f, cr2 = 0.0f;  // value carried up, edge rewards of the level below
        for (uint32_t r = 0; __any(mine && r < t0 + D); ++r) {
            if (mine && r >= t0 && r < t0 + D && err == 0) {
                const uint32_t lvl = r - t0;
                const PathStep ps = path_at(lvl);
                NodeStats& N = m.stats[ps & 0x3FFFFFFu];
                NodeH0 a = N.h0;
                const NodeH1 h = N.h1;
                if (lvl == 0) {
                    // the gathered leaf: populate_node (tree.rs:156-173) + first finalize (backup_round B_ENTRY)
                    float g1 = 0.0f, g2 = 0.0f;
                    if (kind == PROC_EVAL) {
                        const NodeH2 c = N.h2;
                        const EvalOut o = ev[ev_idx];
                        float red1[5], red2[5];
                        reduce_prior(c.omap[0], o.p1, red1);
                        reduce_prior(c.omap[1], o.p2, red2);
                        for (int k = 0; k < 5; ++k) {
                            N.e[0][k].prior = red1[k];
                            N.e[1][k].prior = red2[k];
                        }
                        g1 = o.v1;
                        g2 = o.v2;
                    }
                    finalize_h0(a, g1, g2, 1);
                    N.h0 = a;
                    v1 = g1;
                    v2 = g2;
                } else {
                    // one ancestor (backup_round B_LEVEL, search.rs:834-851)
                    const uint32_t a1 = (ps >> 26) & 7u, a2 = ps >> 29;
                    Edge e1 = N.e[0][a1], e2 = N.e[1][a2];
                    const float q1 = cr1 + v1, q2 = cr2 + v2;
                    finalize_h0(a, q1, q2, 1);
                    edge_update(e1, q1, 1);
                    edge_update(e2, q2, 1);
                    N.h0 = a;
                    N.e[0][a1] = e1;
                    N.e[1][a2] = e2;
                    v1 = q1;
                    v2 = q2;
                }
                cr1 = h.r1;
                cr2 = h.r2;
                nv += 1;
            }
            // the next round's lanes read what this round's lanes wrote (one wavefront, one L1: a workgroup-scope
            // release / acquire is a wait for the stores, not a cache write-back)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    // ---- 3. collisions (search.rs:860-889): integer subtractions, any order, atomics where paths meet
    for (uint32_t base = 0; __any(base < n_coll); base += 16) {
        const uint32_t e = base + w;
        uint32_t cur = NIL, mv = 0;
        if (e < n_coll) {
            const CollEntry ce = m.coll[e];
            cur = ce.node;
            mv = ce.mv;
        }
        bool walking = cur != NIL;
        while (__any(walking)) {
            if (walking) {
                const NodeH1 h = m.stats[cur].h1;
                const NodeH2 c = m.stats[cur].h2;
                if (h.parent == NIL) {
                    walking = false;
                } else {
                    NodeStats& P = m.stats[h.parent];
                    atomicSub(&P.h0.nif, mv);
                    atomicSub(&P.e[0][meta_po(c.meta, 0)].nif, mv);
                    atomicSub(&P.e[1][meta_po(c.meta, 1)].nif, mv);
                    cur = h.parent;
                }
            }
        }
    }
    // counters: sum over the group's lanes, written by lane 0
    for (int off = 8; off > 0; off >>= 1) {
        nv += (uint32_t)__shfl_xor((int)nv, off, 64);
        err |= (uint32_t)__shfl_xor((int)err, off, 64);
    }
    if (active && w == 0) {
        S.nv_backup += nv;
        if (err) S.error = err;
    }
}

}  // namespace ar
#endif
