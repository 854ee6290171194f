// libalpharat_hip.so -- HIP kernels + host runtime + C-ABI (include/alpharat_hip.h).
//
// Host runtime = the MI355X counterpart of crates/alpharat-sampling/src/selfplay.rs:609-808
// (game_worker_loop / run_self_play_to_disk): instead of N OS threads each walking one tree,
// every resident game advances one simulate_batch per kernel step; finished games are drained to
// a writer thread and their slots refilled from the pending game list.
//
// gfx950 only. No CPU path: every entry point needs a HIP device and fails with AR_E_DEVICE
// without one.
#include <errno.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/alpharat_hip.h"
#include "nets.h"
#include "npz_writer.h"
#include "slot_layout.h"
#include "dev_gather8.h"
#include "dev_gatherw.h"
#include "dev_backup16.h"
#include "zig_norm_tables.inc"

using namespace ar;

// ------------------------------------------------------------------------------------------------
// hardware queues
// ------------------------------------------------------------------------------------------------
// The step pipeline of a large run keeps four streams busy at once (two groups of games x {walk + evaluator, tree
// reuse}; group 0 runs on the engine's own stream). The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default:
// streams that share a queue serialise, and the pipeline runs 1.1-1.7x slower than with one queue per stream
// (measured). The variable is read when the runtime initialises, so it is set when this
// library is loaded -- unless the host has chosen a value itself. A process that brought the runtime up earlier
// (e.g. torch imported first) has to export it itself: bench.py does.
// What is known about the queues is what the variable said when this library was loaded: a value set later (by this
// constructor, or by anyone) does not reach a runtime that is already up. AR_HW_QUEUES_UNKNOWN=1 (set by
// alpharat_amd/_lib.py when torch has initialised a device before the library is loaded) says exactly that.
static int g_hw_queues_at_load = 0;
__attribute__((constructor)) static void ar_more_hw_queues() {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    if (getenv("AR_HW_QUEUES_UNKNOWN")) {
        g_hw_queues_at_load = e ? atoi(e) : 4;  // whatever the runtime came up with: not ours to change now
        return;
    }
    if (!e) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    e = getenv("GPU_MAX_HW_QUEUES");
    g_hw_queues_at_load = e ? atoi(e) : 4;
}
static int hw_queues() { return g_hw_queues_at_load; }

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;
static int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
static int nets_fail(int code, const std::string& msg) { return fail(code, msg); }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(AR_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int NW>
struct GameInit {
    Board board;
    State<NW> st;
    uint64_t rng_seed;
    uint32_t game_index;
    uint32_t slot;
    uint32_t single;
    uint32_t reset_arena;  // 1: the slot's grown arena was released, go back to its pool share
};

template <int NW>
struct DoneInfo {
    uint32_t slot, game_index, n_pos, error;
    State<NW> final_st;
    uint64_t t_sims, t_nn, t_term, t_coll, nv_gather, nv_backup, new_nodes;
    MoveResult last;
};

// Where the trees live. One allocation per device holds every game's arena, cut into PAGES of 768 nodes (243 KB with
// the nodes' share of the re-rooting table); a game owns one run of consecutive pages at a time, exactly as many as its
// kept tree plus one full search needs, and changes run when tree reuse re-roots the tree (k_advance rewrites every kept
// node anyway). A resident game so costs what its tree needs -- not a fixed first arena plus overflow blocks (rounds 1-2:
// 3.7 MB per game for 1.7 MB of live nodes, which capped the device at 65536 games).
// Which pages are taken is a bitmap, 64 pages to a word; a run lies inside one word (at most 64 pages = 49152 nodes;
// bigger trees get an arena from the host), so finding n free pages is bit arithmetic on one word and claiming them one
// atomic OR -- and pages that come back merge with their free neighbours by themselves (size classes with free lists
// were tried first: a session's games start in step, every class sees its peak demand once, and the region ends up
// cut into blocks of sizes nobody wants any more).
// Kernels only CLAIM pages and only APPEND runs they leave to the return lists; k_pool_merge (ordered after the kernels
// that could still read those pages) clears their bits -- so no page is handed out while an earlier kernel may read it.
// The region is cut into ZONES, one per 1024 consecutive slots, each with its own bitmap: a wavefront (and a CU) works on
// consecutive slots, and their trees then lie within a few GB of each other. With one pool for the whole device the
// trees of neighbouring slots ended up anywhere in 260 GB and every kernel that walks trees ran 2-3x slower, more so
// the longer the run (address translation: the big allocation is mapped with large fragments, and a CU that touches a
// few of them is served from its translation cache).
enum { POOL_PAGE_NODES = 768, POOL_WORD_PAGES = 64, POOL_ZONE_SLOTS = 1024 };
struct ArenaPool {
    unsigned long long* bits;   // [zones x zone_words] bit set: page taken
    unsigned long long* ret;    // [zones x ret_cap] runs left since the last merge: page index within the zone | pages << 32
    uint32_t* ret_n;            // [zones]
    uint32_t zone_words, zones, ret_cap;
    uint32_t page_nodes;        // nodes per page (768; the AR_ARENA_NODES test knob makes pages small)
    uint32_t fresh_pages;       // pages of a fresh game's arena
    unsigned long long page_bytes, zone_bytes;  // zone z is [z * zone_bytes, (z + 1) * zone_bytes) of the region
};

// The cost tables are read once or twice in every round of a gather (the step into the next child), in front of the
// round's record fetch: with one maze shared by all games (open mazes) -- or a handful, in batched searches -- the
// whole pool is copied into LDS at kernel start, which takes a dependent L2 round trip out of every round.
enum { MAZE_STAGE_BYTES = 4096 };

// the three bases every tree kernel receives; all per-game addresses derive from them
struct Bases {
    unsigned char* arena;    // pool of first arenas; grown arenas are addressed relative to it too
    unsigned char* scratch;  // S x layout.total
    const uint8_t* maze;     // cost tables
    uint32_t maze_stage;     // bytes of the whole maze pool when it is small enough to sit in LDS (else 0)
    SlotLayout L;
    ArenaPool pool;
};

// pages for `need` nodes; 0 when no run can hold them (more than one word of pages)
__device__ inline uint32_t pool_pages_for(const ArenaPool& P, uint32_t need) {
    const uint32_t n = (need + P.page_nodes - 1) / P.page_nodes;
    return n <= POOL_WORD_PAGES ? (n ? n : 1u) : 0u;
}
// A run of `n` free pages in the slot's zone: first fit from a start word that differs from slot to slot (games that
// look at the same time then look at different words). Returns the first page's index within the zone, NIL if none.
// Called by one thread.
__device__ inline uint32_t pool_claim(const ArenaPool& P, uint32_t slot, uint32_t n) {
    const uint32_t zone = slot / POOL_ZONE_SLOTS;
    unsigned long long* words = P.bits + (size_t)zone * P.zone_words;
    const unsigned long long run = n >= 64 ? ~0ULL : ((1ULL << n) - 1ULL);
    // (the start words cover the low end of the zone only: a zone that is far from full keeps its trees close together)
    uint32_t w = (slot * 7u) % (P.zone_words < 96u ? P.zone_words : 96u);
    for (uint32_t tries = 0; tries < P.zone_words; ++tries, w = w + 1 == P.zone_words ? 0 : w + 1) {
        for (int again = 0; again < 4; ++again) {  // (a few retries on a word others are claiming from too)
            const unsigned long long taken = __hip_atomic_load(&words[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long m = ~taken;  // bit p survives when pages p .. p + n - 1 are all free
            for (uint32_t len = 1; len < n && m;) {
                const uint32_t sh = len < n - len ? len : n - len;
                m &= m >> sh;
                len += sh;
            }
            if (!m) break;
            const uint32_t pos = (uint32_t)__ffsll((long long)m) - 1u;
            const unsigned long long mask = run << pos;
            const unsigned long long old = atomicOr(&words[w], mask);
            if ((old & mask) == 0) return w * POOL_WORD_PAGES + pos;
            atomicAnd(&words[w], ~(mask & ~old));  // someone was faster on part of it: give back what was ours, look again
        }
    }
    return NIL;
}
// where a slot's arena fields point when it owns `n` pages from page `first` of its zone
template <int NW>
__device__ inline void slot_at_pages(Slot<NW>& s, const Bases& B, uint32_t slot, uint32_t first, uint32_t n) {
    const uint32_t cap = n * B.pool.page_nodes;
    const long long base = (long long)((unsigned long long)(slot / POOL_ZONE_SLOTS) * B.pool.zone_bytes + (unsigned long long)first * B.pool.page_bytes);
    s.cap = cap;
    s.stats_off = base;
    s.fwd_off = base + (long long)cap * (long long)sizeof(NodeStats);
    s.pool_blk = n;
}
// the arena a slot leaves behind: pages go to the return list of their zone, host-grown arenas are flagged
template <int NW>
__device__ inline void leave_arena(const Slot<NW>& old, Slot<NW>& s, const Bases& B) {
    if (old.pool_blk) {
        const uint32_t zone = (uint32_t)((unsigned long long)old.stats_off / B.pool.zone_bytes);
        const uint32_t first = (uint32_t)(((unsigned long long)old.stats_off - (unsigned long long)zone * B.pool.zone_bytes) / B.pool.page_bytes);
        const uint32_t at = atomicAdd(&B.pool.ret_n[zone], 1u);
        if (at < B.pool.ret_cap) B.pool.ret[(size_t)zone * B.pool.ret_cap + at] = (unsigned long long)first | ((unsigned long long)old.pool_blk << 32);
        // (the list holds a run per slot of the zone and then some: more cannot be left between two merges)
    } else if (old.cap != 0) {
        s.release_grown = 1;
    }
}
// nodes a game needs room for when its kept tree has `cnt` nodes: the tree plus one full search
__device__ inline uint32_t arena_need(uint32_t cnt, const SearchCfg& cfg) { return cnt + cfg.n_sims + 2 * cfg.batch_size + 64; }

__global__ void k_pool_merge(ArenaPool P, uint32_t zone_first, uint32_t zone_count) {  // one block per zone
    if (blockIdx.x >= zone_count) return;
    const uint32_t zone = zone_first + blockIdx.x;
    if (zone >= P.zones) return;
    uint32_t n = P.ret_n[zone];
    if (n > P.ret_cap) n = P.ret_cap;
    unsigned long long* words = P.bits + (size_t)zone * P.zone_words;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long r = P.ret[(size_t)zone * P.ret_cap + i];
        const uint32_t first = (uint32_t)r, pages = (uint32_t)(r >> 32);
        const unsigned long long run = pages >= 64 ? ~0ULL : ((1ULL << pages) - 1ULL);
        atomicAnd(&words[first / POOL_WORD_PAGES], ~(run << (first % POOL_WORD_PAGES)));
    }
    __syncthreads();
    if (threadIdx.x == 0) P.ret_n[zone] = 0;
}

// Slot ownership across the two streams of a group. The tree walks (k_gather, k_backup) own a slot
// while it is ACTIVE; k_advance, on the side stream, owns it from the backup that ended a move (or the
// gather that stalled) until the compaction is done. No kernel may take over a slot from a kernel that
// is still running -- only a kernel boundary orders the writes (and writes back / invalidates the
// per-XCD L2s), and in-kernel agent-scope fences cost a full L2 write-back each. So every hand-off
// status carries the parity of the step that wrote it, and a kernel only accepts the parity whose
// writer the stream order has already completed:
//   backup(t) / gather(t) write ADVANCE / STALL tagged t&1  -> accepted by advance(t) only (launched after
//                                                             backup(t); advance(t-1) may still be running
//                                                             next to backup(t) and ignores this parity);
//   advance(t) writes READY tagged t&1                       -> accepted by gather(t+2), which waits for
//                                                             advance(t)'s event; gather(t+1) ignores it.
// Without the side stream everything runs in stream order with phase 0 and READY = ACTIVE.
enum { SLOT_STALL_B = 6, SLOT_ADVANCE_B = 7, SLOT_READY_A = 8, SLOT_READY_B = 9 };
__device__ inline uint32_t tag_status(uint32_t st, uint32_t phase) {
    if (phase == 0) return st;
    return st == SLOT_STALL ? (uint32_t)SLOT_STALL_B : st == SLOT_ADVANCE ? (uint32_t)SLOT_ADVANCE_B : st;
}

// per-game mazes (generated mazes): maze i of the upload goes to the pool entry of the slot game i starts in
template <int NW>
__global__ void k_place_mazes(const GameInit<NW>* init, uint32_t n, const uint8_t* stage, uint32_t stride, uint8_t* pool,
                              uint32_t* ids_out) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t slot = init[i].slot;
    for (uint32_t b = threadIdx.x; b < stride; b += blockDim.x) pool[(size_t)slot * stride + b] = stage[(size_t)i * stride + b];
    if (threadIdx.x == 0) ids_out[i] = slot;
}

template <int NW>
__global__ void k_init_games(Slot<NW>* slots, const GameInit<NW>* init, uint32_t n, Bases B, SearchCfg cfg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GameInit<NW> gi = init[i];
    Slot<NW> s = slots[gi.slot];
    // the block of the game that was here goes back (a host-grown arena was freed by the host already); the new game
    // takes one for a fresh tree
    if (s.pool_blk) {
        Slot<NW> unused = s;
        leave_arena(s, unused, B);
    }
    s.cap = 0;
    s.pool_blk = 0;
    s.board = gi.board;
    s.st = gi.st;
    rng_seed(s.rng, gi.rng_seed);
    s.game_index = gi.game_index;
    s.single_search = gi.single;
    const uint32_t off = pool_claim(B.pool, gi.slot, B.pool.fresh_pages);
    if (off != NIL) {
        slot_at_pages(s, B, gi.slot, off, B.pool.fresh_pages);
        const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, gi.slot, B.L, B.maze);
        start_game(s, m, cfg);
    } else {
        // no block right now: the game waits as a tree to be re-rooted from nothing; k_advance gives it its arena
        // and its root when blocks have come back
        start_game_header(s, cfg);
        s.hi = 0;
        s.root = 0;
        s.node_count = 0;
        s.stats_off = s.fwd_off = 0;
        s.pending_root = NIL;
        s.status = (!s.single_search && st_over(s.board, s.st)) ? (uint32_t)SLOT_DONE : (uint32_t)SLOT_ADVANCE;
    }
    slots[gi.slot] = s;
}

// One lane walks one game's tree: `iters` x (gather -> evaluate -> backup [-> end of turn]).
// SmartUniform is evaluated inside the gather, so the whole simulate_batch is one pass. The slot
// header is copied to registers for the duration of the kernel.
template <int NW>
__global__ void __launch_bounds__(64) k_step_uniform(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B,
                                                     const ZigTables* zt, int iters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    if (slots[i].status != SLOT_ACTIVE) return;
    Slot<NW> s = slots[i];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, i, B.L, B.maze);
    // (fused_machine -- lanes free-running across batch boundaries -- measured slower: it spreads the
    // wavefront over twice as many states per round)
    for (int it = 0; it < iters; ++it) {
        if (s.status != SLOT_ACTIVE) break;
        if (!gather_machine(s, m, cfg, EVAL_UNIFORM)) break;
        if (backup_machine(s, m, cfg, m.ev_local, zt)) finish_move(s, m, cfg);
    }
    slots[i] = s;
}

// Split form for evaluators that run outside the tree walk (networks, host callbacks).
template <int NW>
__global__ void __launch_bounds__(64) k_gather(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B,
                                               LeafReq<NW>* queue, uint32_t* queue_count, uint32_t max_rounds,
                                               uint32_t lanes, uint32_t first, uint32_t phase,
                                               uint32_t accept_ready) {
    // slots [first, n_slots) -- one group of games; `lanes` games per wavefront (<= 64)
    __shared__ uint32_t lds_maze[MAZE_STAGE_BYTES / 4];
    if (B.maze_stage) {
        for (uint32_t w = threadIdx.x; w < B.maze_stage / 4; w += blockDim.x) lds_maze[w] = ((const uint32_t*)B.maze)[w];
        __syncthreads();
    }
    if (threadIdx.x >= lanes) return;
    const uint32_t i = first + blockIdx.x * lanes + threadIdx.x;
    if (i >= n_slots) return;
    {
        const uint32_t st = slots[i].status;
        if (st != SLOT_ACTIVE && st != accept_ready) return;
    }
    Slot<NW> s = slots[i];
    s.status = SLOT_ACTIVE;
    Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, i, B.L, B.maze);
    if (B.maze_stage) m.cost = (const uint8_t*)lds_maze + s.board.maze_off;
    // a batch that was already gathered and still waits for its backup is left alone
    const int got = s.batch_active ? GATHER_PENDING : gather_machine_limited(s, m, cfg, EVAL_STORE, max_rounds);
    if (got == GATHER_COMPLETE && queue != nullptr && s.b_nn > 0) {
        const uint32_t base = atomicAdd(queue_count, s.b_nn);
        s.eval_base = base;
        for (uint32_t j = 0; j < s.b_nn; ++j) {
            LeafReq<NW> r;
            r.st = m.leaf_local[j];
            r.slot = i;
            r.pad = 0;
            queue[base + j] = r;
        }
    }
    s.status = tag_status(s.status, phase);
    slots[i] = s;
}

// The same gather with eight lanes per game (dev_gather8.h): a wavefront holds eight games, an octet of lanes walks
// one tree together. Launched with ceil(n / 8) blocks of 64 threads. Trees, batch entries, counters and the leaf
// queue contents per game are identical to k_gather's; only the order in which games append to the queue differs.
template <int NW, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_gather8(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B,
                                                LeafReq<NW>* queue, uint32_t* queue_count, uint32_t first,
                                                uint32_t phase, uint32_t accept_ready) {
    const uint32_t ol = threadIdx.x & 7u;
    const uint32_t i = first + blockIdx.x * 8u + (threadIdx.x >> 3);
    __shared__ uint32_t lds_maze[MAZE_STAGE_BYTES / 4];
    bool run = i < n_slots;
    if (run) {
        const uint32_t st = slots[i].status;
        run = st == SLOT_ACTIVE || st == accept_ready;
    }
    const uint32_t ii = i < n_slots ? i : first;  // idle octets read a valid slot and store nothing
    Slot<NW>& S = slots[ii];
    OctMem<NW> m;
    m.stats = (NodeStats*)(B.arena + S.stats_off);
    m.scratch = B.scratch;
    m.maze = B.maze;
    if (B.maze_stage) {
        for (uint32_t w = threadIdx.x; w < B.maze_stage / 4; w += blockDim.x) lds_maze[w] = ((const uint32_t*)B.maze)[w];
        m.maze = (const uint8_t*)lds_maze;  // (the barrier below, in front of the rounds, orders these writes)
    }
    m.s_off = (uint32_t)((size_t)ii * B.L.total);  // (setup() keeps all games' scratch below 4 GB for this kernel)
    m.maze_off = S.board.maze_off;
    m.proc_off = (uint32_t)B.L.proc_off;
    m.coll_off = (uint32_t)B.L.coll_off;
    m.levels_off = (uint32_t)B.L.levels_off;
    m.leaf_off = (uint32_t)B.L.leaf_off;
    m.coll_cap = B.L.coll_cap;
    m.max_depth = B.L.max_depth;
    const Board board = S.board;
    __shared__ OctShared<NW> shared[8];
    __shared__ OutcomeTable otab;
    outcome_table_fill(otab);
    OctShared<NW>& sh = shared[threadIdx.x >> 3];
    Oct<NW> o;
    o.done = true;
    o.alloc_left = 0;
    o.error = 0;
    o.d_new = o.d_visits = 0;
    o.rounds = 0;
    o.batch_active = S.batch_active;
    bool stalled = false;
    // a batch that was already gathered and still waits for its backup is left alone
    const bool begin = run && o.batch_active == 0;
    if (begin) {
        // gather_begin (dev_search.h)
        o.hi = S.hi;
        o.cap = S.cap;
        o.root = S.root;
        o.node_count = S.node_count;
        const uint32_t remaining = S.remaining;
        o.batch = remaining < cfg.batch_size ? remaining : cfg.batch_size;
        if (o.hi + o.batch > o.cap) {
            stalled = true;
        } else {
            o.left = (long long)(int32_t)collisions_left(o.node_count, cfg);
            o.n_proc = o.n_coll = o.b_nn = o.b_term = o.b_coll = 0;
            o.depth = 0;
            o.node = 0;
            o.mask = 0;
            o.omap0 = o.omap1 = 0;
            o.pick_mv = 0;
            o.have_pick = false;
            o.work = S.st;
            sh.rng = S.rng;  // (all eight lanes store the same values: see best_of5)
            sh.root_st = S.st;
            o.n1 = o.n2 = 0;
            o.forced = 0;
            for (int k = 0; k < 2; ++k) {
                o.sc[k] = o.util[k] = o.num[k] = 0.0f;
                o.ns[k] = o.add[k] = o.nif0[k] = 0;
            }
            for (int k = 0; k < 4; ++k) {
                o.kid[k] = NIL;
                o.vtp[k] = 0;
            }
            o.done = false;
        }
    }
    __syncthreads();  // (one wavefront: the LDS writes above are visible to its other lanes)
    for (uint32_t guard = 0; guard < (1u << 22); ++guard) {  // (every game's gather ends; the bound is a fuse)
        if (!__any(!o.done)) break;
        gather8_round(o, sh, otab, board, m, cfg, ol);
    }
    __syncthreads();
    if (!run) return;
    if (begin && !stalled && !o.done) o.error = 8;  // the fuse blew
    uint32_t status = SLOT_ACTIVE;
    if (stalled) status = SLOT_STALL;
    if (begin && !stalled) {
        uint32_t base = 0;
        const bool complete = o.done && o.batch_active != 0;
        if (complete && queue != nullptr && o.b_nn > 0) {
            if (ol == 0) base = atomicAdd(queue_count, o.b_nn);
            base = oct_pick(base, 0);
            for (uint32_t j = ol; j < o.b_nn; j += 8) {
                LeafReq<NW> r;
                r.st = m.leaves()[j];
                r.slot = i;
                r.pad = 0;
                queue[base + j] = r;
            }
        }
        if (ol == 0) {
            S.hi = o.hi;
            S.node_count = o.node_count;
            S.new_nodes += o.d_new;
            S.nv_gather += o.d_visits;
            S.n_proc = o.n_proc;
            S.n_coll = o.n_coll;
            S.b_nn = o.b_nn;
            S.b_term = o.b_term;
            S.b_coll = o.b_coll;
            S.batch_active = o.batch_active;
            S.eval_base = base;
            S.rng = sh.rng;
            S.gather_pending = 0;
            S.g_rounds = o.rounds;
            if (o.error) S.error = o.error;
        }
    }
    if (ol == 0) {
        if (stalled) S.need_nodes = S.hi + cfg.n_sims + 2 * cfg.batch_size;
        S.status = tag_status(status, phase);
    }
}


// The gather as a work queue over tree levels (dev_gatherw.h): a wavefront serves G game contexts at once, every node
// a pick reaches is an item in the wavefront's LDS ring, and each pass the 64 lanes take the next 64 items -- of
// whichever games they are. The grid is persistent: context c of wavefront b walks game c of the sets (of G consecutive
// slots) j * gridDim.x + b, j = 0, 1, ... one after the other, so the lanes stay busy until the launch runs out of games.
// With a pass limit the sets of turn j >= 1 only get the passes their wavefront's first set leaves over, so the numbering of
// the sets is rotated from launch to launch (`rot`: set v of this launch is set (v + rot) mod n_sets; the host advances rot
// by n_sets - gridDim.x per launch, the width of the second turn): every game waits its share of the launches instead of
// the same eighth of the games waiting in all of them. Which wavefront walks a game never changes what its walk does.
// Leaves are left in the games' scratch (k_pack_leaves appends them to the evaluator queue).
// Ring item: context (bits 0..6), BEGIN flag (bit 7: the context wants a game), visit slot, siblings behind it.
template <int NW, int G, int R>
__global__ void __launch_bounds__(64) k_gatherw(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B, uint32_t first,
                                                uint32_t phase, uint32_t accept_ready, uint32_t pass_limit, uint32_t rot) {
    static_assert((G & (G - 1)) == 0 && G <= 64, "the ring is a power of two; a context number fits a ring item");
    static_assert(sizeof(GwRec<NW>) == (sizeof(State<NW>) + 16 + 7) / 8 * 8, "slot_layout.h sizes the spill area");
    typedef GwShared<NW, G, R> Sh;
    __shared__ Sh sh;
    const uint32_t ring_mask = Sh::RING - 1;
    __shared__ GwOutcomeTable otab;
    extern __shared__ uint32_t lds_maze[];  // the shared maze (B.maze_stage bytes), if the run has one
    const uint32_t L = threadIdx.x;
    GwMem<NW> m;
    m.arena = B.arena;
    m.scratch = B.scratch;
    m.maze = B.maze;
    if (B.maze_stage) {
        for (uint32_t w = L; w < B.maze_stage / 4; w += 64) lds_maze[w] = ((const uint32_t*)B.maze)[w];
        m.maze = (const uint8_t*)lds_maze;
    }
    m.slot_bytes = B.L.total;
    m.proc_off = (uint32_t)B.L.proc_off;
    m.coll_off = (uint32_t)B.L.coll_off;
    m.leaf_off = (uint32_t)B.L.leaf_off;
    m.spill_off = (uint32_t)B.L.levels_off;
    m.coll_cap = B.L.coll_cap;
    if (L < 17) gw_outcome_entry(L, otab.omap[L], otab.n[L]);
    if (L == 0) sh.tail = G;
    for (uint32_t c = L; c < (uint32_t)G; c += 64) {
        sh.game[c].slot = NIL;
        sh.game[c].running = 0;
        sh.game[c].next_game = 0;
        for (int j = 0; j < R; ++j) sh.rec_owner[c][j] = 0;
        sh.ring[c] = (uint16_t)(c | 0x80u);  // every context starts by asking for a game
    }
    uint32_t head = 0;
#if defined(AR_STATS)
    // per-wavefront counters (lane 0): passes, items, begins, pick ends, waits, interior, final; 100 MHz clocks of the four phases
    unsigned long long gs_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gs_clk[5] = {0, 0, 0, 0, 0}, gs_hist[17] = {0}, gs_x[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long gs_t0 = wall_clock64();
#define GW_STAT(...) __VA_ARGS__
#else
#define GW_STAT(...)
#endif
    for (uint32_t guard = 0; guard < (1u << 24); ++guard) {  // (the queue drains; the bound is a fuse)
        __syncthreads();
        const uint32_t n = *(volatile uint32_t*)&sh.tail - head;
        if (n == 0) break;
        GW_STAT(const unsigned long long gs_a = wall_clock64();)
        uint32_t take = n < 64u ? n : 64u;
        const uint32_t item = L < take ? (uint32_t)sh.ring[(head + L) & ring_mask] : 0u;
        {
            // the children of one parent are taken in the same pass (they all read the parent's position first)
            const unsigned long long split = __ballot(L < take && L + ((item >> 12) & 15u) >= 64u);
            if (split) take = (uint32_t)__ffsll((long long)split) - 1u;
        }
        head += take;
        const bool mine = L < take;
        const bool begin = mine && (item & 0x80u);
        GW_STAT(gs_cnt[0] += 1; gs_cnt[1] += take; gs_cnt[2] += (unsigned long long)__popcll(__ballot(begin)); gs_hist[take / 4] += 1;)
        const uint32_t g = item & 0x7fu;
        GwLane<NW> ln;
        ln.active = mine && !begin;
        ln.g = g;
        // ---- phase 1
        uint32_t slot_i = NIL;
        bool skip = false;  // a context beyond the end of a ragged last set: nothing to walk this turn, but the next may have
        if (begin && guard < pass_limit) {  // (past the limit no game is begun: its turn comes with the next launch)
            const uint32_t turn = sh.game[g].next_game;
            sh.game[g].next_game = turn + 1u;
            const uint32_t n_sets = (n_slots - first + (uint32_t)G - 1u) / (uint32_t)G;
            const uint32_t v = turn * gridDim.x + blockIdx.x;
            if (v < n_sets) {
                uint32_t set = v + rot;
                if (set >= n_sets) set -= n_sets;
                slot_i = first + set * (uint32_t)G + g;
                skip = slot_i >= n_slots;
            }
        }
        if (ln.active) gw_fetch<NW, R>(ln, item, sh.game, &sh.rec[0][0], &sh.stub[0][0], &sh.stub_node[0][0], m);
        // ---- phase 2
        GW_STAT(const unsigned long long gs_b = wall_clock64();)
        bool started = false, retire = false;
        if (begin) {
            if (slot_i >= n_slots) {
                retire = !skip;
            } else {
                Slot<NW>& S = slots[slot_i];
                const uint32_t st = S.status;
                const bool run = st == SLOT_ACTIVE || st == accept_ready;
                if (run) {
                    uint32_t status = SLOT_ACTIVE;
                    if (S.batch_active == 0) {  // (a batch that still waits for its backup is left alone)
                        GwGame<NW>& Gm = sh.game[g];
                        gw_begin(Gm, S, slot_i, cfg);
                        started = Gm.began != 0;
                        GW_STAT(Gm.dbg_start = guard;)
                        if (Gm.stalled) {
                            gw_end(Gm, S, cfg);
                            status = SLOT_STALL;
                        }
                    }
                    if (!started) S.status = tag_status(status, phase);
                }
            }
        }
        if (ln.active) gw_visit(ln, sh.game[g], sh.rec_owner[g], m, cfg, &otab);
        __syncthreads();
        GW_STAT(const unsigned long long gs_c = wall_clock64();
                gs_cnt[4] += (unsigned long long)__popcll(__ballot(ln.active && ln.wait));
                gs_cnt[5] += (unsigned long long)__popcll(__ballot(ln.active && ln.interior));
                gs_cnt[6] += (unsigned long long)__popcll(__ballot(ln.active && ln.is_final));
                {
                    // the slowest lane of the pass: allocation steps; clocks from the start of the phase to record arrival /
                    // set-up / end of allocation (interior entries only)
                    const bool in = ln.active && ln.interior;
                    uint32_t st = in ? ln.dbg_steps : 0u, t0 = in ? (uint32_t)(ln.dbg_t[0] - gs_b) : 0u,
                             t1 = in ? (uint32_t)(ln.dbg_t[1] - gs_b) : 0u, t2 = in ? (uint32_t)(ln.dbg_t[2] - gs_b) : 0u;
                    uint32_t ssum = st;
                    for (int off = 32; off > 0; off >>= 1) {
                        st = max(st, (uint32_t)__shfl_xor((int)st, off, 64));
                        ssum += (uint32_t)__shfl_xor((int)ssum, off, 64);
                        t0 = max(t0, (uint32_t)__shfl_xor((int)t0, off, 64));
                        t1 = max(t1, (uint32_t)__shfl_xor((int)t1, off, 64));
                        t2 = max(t2, (uint32_t)__shfl_xor((int)t2, off, 64));
                    }
                    gs_x[0] += st; gs_x[1] += ssum; gs_x[2] += t0; gs_x[3] += t1; gs_x[4] += t2;
                    gs_x[5] += (unsigned long long)__popcll(__ballot(ln.active && ln.from_pick));
                })
        // ---- phase 3
        bool finisher = false;
        if (ln.active)
            finisher = gw_publish<NW, R>(ln, sh.game[g], sh.rec[g], sh.rec_owner[g], m.spill(sh.game[g]), sh.stub[g], sh.stub_node[g],
                                         sh.ring, ring_mask, &sh.tail);
        __syncthreads();
        GW_STAT(const unsigned long long gs_d = wall_clock64(); gs_cnt[3] += (unsigned long long)__popcll(__ballot(finisher));)
        // ---- phase 4: the end of a pick / the first pick of a new game / the next game for a context
        if (finisher) gw_finish_pick(sh.game[g], sh.stub[g], sh.ring, ring_mask, &sh.tail, g, guard + 1u < pass_limit);
        if (started) gw_next_pick(sh.game[g], sh.stub[g], sh.ring, ring_mask, &sh.tail, g);
        if (finisher || started) {
            GwGame<NW>& Gm = sh.game[g];
            if (!Gm.running) {  // the game's gather is complete (or parked): hand the slot back, ask for another game
                GW_STAT(atomicAdd(&g_gather_clk[64 + min((guard - Gm.dbg_start) / 8u, 47u)], 1ULL);)
                Slot<NW>& S = slots[Gm.slot];
                gw_end(Gm, S, cfg);
                S.status = tag_status(SLOT_ACTIVE, phase);
                Gm.slot = NIL;
                sh.ring[atomicAdd(&sh.tail, 1u) & ring_mask] = (uint16_t)(g | 0x80u);
            }
        } else if (begin && !retire) {
            sh.ring[atomicAdd(&sh.tail, 1u) & ring_mask] = (uint16_t)(g | 0x80u);  // not a game to walk: try the next one
        }
        GW_STAT(const unsigned long long gs_e = wall_clock64(); gs_clk[0] += gs_b - gs_a; gs_clk[1] += gs_c - gs_b;
                gs_clk[2] += gs_d - gs_c; gs_clk[3] += gs_e - gs_d;)
    }
#if defined(AR_STATS)
    if (L == 0) {
        gs_clk[4] = wall_clock64() - gs_t0;
        atomicAdd(&g_gather_clk[0], 1ULL);
        for (int k = 0; k < 7; ++k) atomicAdd(&g_gather_clk[1 + k], gs_cnt[k]);
        for (int k = 0; k < 5; ++k) atomicAdd(&g_gather_clk[8 + k], gs_clk[k]);
        for (int k = 0; k < 17; ++k) atomicAdd(&g_gather_clk[16 + k], gs_hist[k]);
        for (int k = 0; k < 6; ++k) atomicAdd(&g_gather_clk[40 + k], gs_x[k]);
    }
#endif
}

// The evaluation requests of the games that have just gathered a batch, appended to the device-wide leaf queue
// (replaces MuxBackend's request queue, mux.rs:170-289): one atomic per wavefront reserves the run, every lane copies its
// game's leaves. Only the order of games in the queue depends on the scheduling.
template <int NW>
__global__ void __launch_bounds__(64) k_pack_leaves(Slot<NW>* slots, uint32_t n_slots, Bases B, uint32_t first, LeafReq<NW>* queue,
                                                    uint32_t* queue_count) {
    const uint32_t i = first + blockIdx.x * 64u + threadIdx.x;
    uint32_t nn = 0;
    if (i < n_slots && slots[i].status == SLOT_ACTIVE && slots[i].batch_active != 0) nn = slots[i].b_nn;
    uint32_t incl = nn;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
        if ((threadIdx.x & 63u) >= (uint32_t)off) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    uint32_t base = 0;
    if (threadIdx.x == 0 && total) base = atomicAdd(queue_count, total);
    base = (uint32_t)__shfl((int)base, 0, 64) + incl - nn;
    if (nn == 0) return;
    slots[i].eval_base = base;
    const State<NW>* leaves = (const State<NW>*)(B.scratch + (size_t)i * B.L.total + B.L.leaf_off);
    for (uint32_t j = 0; j < nn; ++j) {
        LeafReq<NW> r;
        r.st = leaves[j];
        r.slot = i;
        r.pad = 0;
        queue[base + j] = r;
    }
}

template <int NW>
__global__ void __launch_bounds__(64) k_backup(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B,
                                               const ZigTables* zt, const EvalOut* ev_queue, uint32_t lanes,
                                               uint32_t first, uint32_t phase) {
    if (threadIdx.x >= lanes) return;
    const uint32_t i = first + blockIdx.x * lanes + threadIdx.x;
    if (i >= n_slots) return;
    if (slots[i].status != SLOT_ACTIVE || !slots[i].batch_active) return;
    Slot<NW> s = slots[i];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, i, B.L, B.maze);
    const EvalOut* ev = ev_queue ? ev_queue + s.eval_base : m.ev_local;
    if (backup_machine(s, m, cfg, ev, zt)) finish_move(s, m, cfg);
    s.status = tag_status(s.status, phase);
    slots[i] = s;
}

// The backup with sixteen lanes per game (dev_backup16.h): ceil(n / 4) blocks of 64 threads, dynamic LDS = 64 paths
// of `path_cap` steps. Ends with batch_end; a search that is complete is left in the state (ACTIVE, no batch,
// remaining == 0) for k_finish.
template <int NW>
__global__ void __launch_bounds__(64) k_backup16(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B,
                                                 const ZigTables* zt, const EvalOut* ev_queue, uint32_t first,
                                                 uint32_t path_cap) {
    extern __shared__ PathStep lds_path[];
    const uint32_t w = threadIdx.x & 15u;
    const uint32_t i = first + blockIdx.x * 4u + (threadIdx.x >> 4);
    bool active = i < n_slots && slots[i].status == SLOT_ACTIVE && slots[i].batch_active != 0;
    const uint32_t ii = i < n_slots ? i : first;
    Slot<NW>& S = slots[ii];
    const Mem<NW> m = resolve_mem<NW>(S, B.arena, B.scratch, ii, B.L, B.maze);
    active = active && backup16_wants(S, m, cfg);
    const EvalOut* ev = ev_queue ? ev_queue + S.eval_base : m.ev_local;
    (void)zt;
    backup16<NW>(active, S, m, cfg, ev, lds_path + (size_t)threadIdx.x * PATH_LDS_STEPS, path_cap, w);
    if (active && w == 0) batch_end(S);
}
// the end of a move (selfplay.rs:538-565 up to the tree reuse) for the games whose search k_backup16 completed
template <int NW>
__global__ void __launch_bounds__(64) k_finish(Slot<NW>* slots, uint32_t n_slots, SearchCfg cfg, Bases B, uint32_t first,
                                               uint32_t phase) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    if (slots[i].status != SLOT_ACTIVE || slots[i].batch_active != 0 || slots[i].remaining != 0) return;
    Slot<NW> s = slots[i];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, i, B.L, B.maze);
    finish_move(s, m, cfg);
    s.status = tag_status(s.status, phase);
    slots[i] = s;
}

// Tree reuse (tree.rs:283-302): one wavefront per game re-roots the tree by the sliding compaction
// described in dev_search.h (advance_tree_scalar is the one-lane statement of the same passes).
// Pass 1 decides keep/drop and new ids 64 nodes at a time (parents inside the same 64 are resolved
// by lane shuffles, the running count by ballot + popcount); pass 2 moves kept nodes.
// The compaction is also where a game changes arena: once the kept tree is counted, the slot picks
// the smallest arena (its first arena or an overflow-pool class) that holds the kept tree plus one
// full search, and pass 2 writes there. So a tree never runs out of room in the middle of a search,
// and a tree that shrank gives its block back. A slot that stalled anyway (pool empty at the time)
// is retried here on every launch: its nodes are copied unchanged into a bigger block.
template <int NW>
__global__ void __launch_bounds__(64) k_advance(Slot<NW>* slots, uint32_t n_slots, Bases B, SearchCfg cfg,
                                                uint32_t first_slot, uint32_t phase, uint32_t ready_status) {
    const uint32_t slot = first_slot + blockIdx.x;
    if (slot >= n_slots) return;
    const uint32_t status = slots[slot].status;
    if (status != tag_status(SLOT_ADVANCE, phase) && status != tag_status(SLOT_STALL, phase)) return;
    const uint32_t lane = threadIdx.x;
    Slot<NW> s = slots[slot];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, slot, B.L, B.maze);

    if (status == tag_status(SLOT_STALL, phase)) {
        const uint32_t want = pool_pages_for(B.pool, s.need_nodes > s.cap + 1 ? s.need_nodes : s.cap + 1);
        uint32_t idx = NIL;
        if (lane == 0 && want) idx = pool_claim(B.pool, slot, want);
        idx = (uint32_t)__shfl((int)idx, 0, 64);
        if (idx == NIL) return;  // nothing free: the host sees the stall and decides
        Slot<NW> d = s;
        slot_at_pages(d, B, slot, idx, want);
        const Mem<NW> md = resolve_mem<NW>(d, B.arena, B.scratch, slot, B.L, B.maze);
        const uint4* ss = (const uint4*)m.stats;
        uint4* ds = (uint4*)md.stats;
        for (uint32_t w = lane; w < s.hi * (uint32_t)(sizeof(NodeStats) / 16); w += 64) ds[w] = ss[w];
        if (lane == 0) {
            leave_arena(s, d, B);
            d.status = ready_status;
            slots[slot] = d;
        }
        return;
    }

    const uint32_t keep_root = s.pending_root;
    if (keep_root == NIL) {  // fresh root: the pages a fresh tree wants (the run it is in, if that is as many)
        if (lane == 0) {
            Slot<NW> d = s;
            const uint32_t want = B.pool.fresh_pages;
            bool ok = s.cap != 0;
            if (s.pool_blk != want) {  // (a bigger run, none, or an arena from the host)
                const uint32_t off = pool_claim(B.pool, slot, want);
                if (off != NIL) {
                    slot_at_pages(d, B, slot, off, want);
                    leave_arena(s, d, B);
                    ok = true;
                }
            }
            if (ok) {  // (an arena too small for a whole search stalls at its first gather and grows, like any other)
                make_root(d, resolve_mem<NW>(d, B.arena, B.scratch, slot, B.L, B.maze));
                d.status = ready_status;
                slots[slot] = d;
            }  // else: no arena yet; the slot stays as it is and the next launch of this parity tries again
        }
        return;
    }
    const uint32_t hi = s.hi;
    uint32_t cnt = 0;
    const uint32_t first = keep_root & ~63u;
    for (uint32_t base = 0; base < first; base += 64) {  // everything before the kept root is dropped
        const uint32_t i = base + lane;
        if (i < hi) __hip_atomic_store(&m.fwd[i], NIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (uint32_t base = first; base < hi; base += 64) {
        const uint32_t i = base + lane;
        // 0 unknown, 1 keep, 2 drop
        uint32_t st = 2, p = NIL;
        if (i < hi) {
            if (i == keep_root) st = 1;
            else if (i > keep_root) {
                p = m.stats[i].h1.parent;
                if (p == NIL) st = 2;
                else if (p < base)
                    st = __hip_atomic_load(&m.fwd[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != NIL ? 1u : 2u;
                else st = 0;
            }
        }
        // parents inside this group of 64: propagate along the id order (parent lane < child lane)
        for (int round = 0; round < 64; ++round) {
            const uint32_t src = (st == 0) ? (p - base) : lane;
            const uint32_t pst = (uint32_t)__shfl((int)st, (int)src, 64);
            if (st == 0 && pst != 0) st = pst;
            if (!__any(st == 0)) break;
        }
        const bool keep = st == 1;
        const unsigned long long bal = __ballot(keep);
        const uint32_t before = (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
        if (i < hi)
            __hip_atomic_store(&m.fwd[i], keep ? cnt + before : NIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cnt += (uint32_t)__popcll(bal);
        // the next group's lanes read this group's new ids: the stores must have left the wavefront
        // (one wavefront on one CU: workgroup scope is enough, and an agent-scope fence would write
        // back the XCD's whole L2 -- 550 times per step)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // destination arena: the smallest that holds the kept tree plus one search
    const uint32_t want = pool_pages_for(B.pool, arena_need(cnt, cfg));  // (0: beyond a run; the tree stays where it is)
    Slot<NW> d = s;
    bool moved = false;
    if (want != 0 && want != s.pool_blk) {  // (pool_blk 0: an arena from the host)
        uint32_t idx = NIL;
        if (lane == 0) idx = pool_claim(B.pool, slot, want);
        idx = (uint32_t)__shfl((int)idx, 0, 64);
        if (idx != NIL) {
            slot_at_pages(d, B, slot, idx, want);
            moved = true;
        }
    }
    const Mem<NW> md = resolve_mem<NW>(d, B.arena, B.scratch, slot, B.L, B.maze);

    for (uint32_t base = first; base < hi; base += 64) {
        const uint32_t i = base + lane;
        uint32_t ni = NIL;
        if (i < hi) ni = __hip_atomic_load(&m.fwd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        NodeStats nd;
        if (ni != NIL) {
            nd = m.stats[i];
            nd.h1.parent = i == keep_root ? NIL
                                          : __hip_atomic_load(&m.fwd[nd.h1.parent], __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int c = 0; c < 25; ++c)
                if (nd.c[c] != NIL)
                    nd.c[c] = __hip_atomic_load(&m.fwd[nd.c[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();  // every lane has read its node before any lane overwrites a source
        if (ni != NIL) md.stats[ni] = nd;
        __syncthreads();
    }
    if (lane == 0) {
        if (moved) leave_arena(s, d, B);
        d.root = 0;
        d.hi = cnt;
        d.node_count = cnt;
        d.status = ready_status;
        slots[slot] = d;
    }
}

// evaluator failure: revert the gathered batch (search.rs:919-955)
template <int NW>
__global__ void k_cancel(Slot<NW>* slots, uint32_t n_slots, Bases B) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    Slot<NW> s = slots[i];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, i, B.L, B.maze);
    if (s.batch_active || s.gather_pending) cancel_batch(s, m);  // (a parked gather holds claims and virtual losses too)
    s.gather_pending = 0;
    if (s.status == SLOT_ACTIVE) s.status = SLOT_FAILED;
    slots[i] = s;
}

// counts[0] done, [1] stalled, [2] active (incl. waiting for k_advance), [3] errors; lists hold slot ids.
// live[0..9): sums over the games that sit in a slot right now (finished-but-not-drained ones included) of
// positions, simulations, nn evals, terminals, collisions, gather / backup node-visits, new nodes, and the
// number of such games -- the host adds them to the totals of the games already drained, so a session can
// report the work done inside a window of steps (per finished move) without waiting for games to end.
enum { LIVE_N = 9 };
template <int NW>
__global__ void k_scan(const Slot<NW>* slots, uint32_t n_slots, uint32_t* counts, uint32_t* done_list,
                       uint32_t* stall_list, uint32_t* release_list, ArenaPool P, unsigned long long* live) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_slots;
    if (in && P.bits && i < P.zones * P.zone_words) {  // for the host's bookkeeping: free pages of the region
        const uint32_t free_pages = (uint32_t)__popcll(~P.bits[i]);
        if (free_pages) atomicAdd(&counts[5], free_pages);
    }
    unsigned long long v[LIVE_N];
    for (int k = 0; k < LIVE_N; ++k) v[k] = 0;
    if (in) {
        const Slot<NW>& s = slots[i];
        const uint32_t st = s.status;
        if (s.error) atomicAdd(&counts[3], 1u);
        if (s.release_grown) release_list[atomicAdd(&counts[4], 1u)] = i;
        if (st == SLOT_DONE) done_list[atomicAdd(&counts[0], 1u)] = i;
        else if (st == SLOT_STALL || st == SLOT_STALL_B) stall_list[atomicAdd(&counts[1], 1u)] = i;
        else if (st == SLOT_ACTIVE || st == SLOT_ADVANCE || st == SLOT_ADVANCE_B || st == SLOT_READY_A || st == SLOT_READY_B)
            atomicAdd(&counts[2], 1u);
        if (st != SLOT_EMPTY && live != nullptr) {
            v[0] = s.n_pos;
            v[1] = s.t_sims;
            v[2] = s.t_nn;
            v[3] = s.t_term;
            v[4] = s.t_coll;
            v[5] = s.nv_gather;
            v[6] = s.nv_backup;
            v[7] = s.new_nodes;
            v[8] = 1;
        }
        // what a game that has been running for a while holds in tree pages (the host sizes the resident set by it)
        if (live != nullptr && st != SLOT_EMPTY && st != SLOT_DONE && s.pool_blk != 0 && s.n_pos >= 4) {
            atomicAdd(&live[12], (unsigned long long)s.pool_blk);
            atomicAdd(&live[13], 1ULL);
        }
    }
    if (live == nullptr) return;
    for (int k = 0; k < LIVE_N; ++k) {  // one atomic per wavefront and counter
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += (unsigned long long)__shfl_xor((long long)x, off, 64);
        if ((threadIdx.x & 63u) == 0 && x) atomicAdd(&live[k], x);
    }
}

// one block per finished game: header by thread 0, position records copied by the whole block
template <int NW>
__global__ void k_pack_done(Slot<NW>* slots, const uint32_t* done_list, uint32_t n_done, DoneInfo<NW>* info,
                            PosRec<NW>* staging, uint32_t max_turns, Bases B) {
    const uint32_t d = blockIdx.x;
    if (d >= n_done) return;
    const uint32_t slot = done_list[d];
    Slot<NW>& s = slots[slot];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, slot, B.L, B.maze);
    if (threadIdx.x == 0) {
        DoneInfo<NW>& o = info[d];
        o.slot = slot;
        o.game_index = s.game_index;
        o.n_pos = s.n_pos;
        o.error = s.error;
        o.final_st = s.st;
        o.t_sims = s.t_sims;
        o.t_nn = s.t_nn;
        o.t_term = s.t_term;
        o.t_coll = s.t_coll;
        o.nv_gather = s.nv_gather;
        o.nv_backup = s.nv_backup;
        o.new_nodes = s.new_nodes;
        o.last = s.last;
    }
    const uint32_t n = s.n_pos < max_turns ? s.n_pos : max_turns;
    const uint32_t words = n * (uint32_t)(sizeof(PosRec<NW>) / 4);
    const uint32_t* src = (const uint32_t*)m.pos;
    uint32_t* dst = (uint32_t*)(staging + (size_t)d * max_turns);
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    __syncthreads();
    if (threadIdx.x == 0) {
        // the finished game's tree pages go back now (the slot may stay empty for a while: the host admits new games by
        // what the zone has free); an arena from the host is flagged for release
        Slot<NW> unused = s;
        leave_arena(s, unused, B);
        if (unused.release_grown) s.release_grown = 1;
        s.cap = 0;
        s.pool_blk = 0;
        s.status = SLOT_EMPTY;
    }
}

// arena growth: what a stalled slot reports, and its new home once the host has copied the nodes
struct StallInfo {
    uint32_t slot, cap, hi, need;
    long long stats_off;
};
struct GrowReq {
    uint32_t slot, cap;
    long long stats_off, fwd_off;
};
template <int NW>
__global__ void k_read_stall(const Slot<NW>* slots, const uint32_t* stall_list, uint32_t n, StallInfo* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Slot<NW>& s = slots[stall_list[i]];
    StallInfo o;
    o.slot = stall_list[i];
    o.cap = s.cap;
    o.hi = s.hi;
    o.need = s.need_nodes;
    o.stats_off = s.stats_off;
    out[i] = o;
}
template <int NW>
__global__ void k_clear_release(Slot<NW>* slots, const uint32_t* list, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[list[i]].release_grown = 0;
}
template <int NW>
__global__ void k_apply_grow(Slot<NW>* slots, const GrowReq* req, uint32_t n, Bases B) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Slot<NW>& s = slots[req[i].slot];
    if (s.pool_blk) {
        Slot<NW> unused = s;
        leave_arena(s, unused, B);
    }
    s.pool_blk = 0;
    s.cap = req[i].cap;
    s.stats_off = req[i].stats_off;
    s.fwd_off = req[i].fwd_off;
    s.status = SLOT_ACTIVE;
}

// ------------------------------------------------------------------------------------------------
// NN-evaluation cache (cached_backend.rs:70-130, nn_cache.rs): positions that were evaluated before
// skip the network. The reference keeps one FIFO table per worker thread keyed by a 64-bit FNV-1a
// hash of everything the network sees; here one device-wide open-addressing table serves all games,
// indexed by the same kind of hash but matched on the full position, so a hit is exact. A hit returns
// bit for bit what the network kernel would compute again (it is deterministic per leaf), so game
// records do not depend on the cache, only the work does.
//   k_cache_probe: every queued leaf looks itself up (<= CACHE_PROBES slots from its home slot); hits
//     get their result at once, misses are compacted into a second queue for the network;
//   k_cache_fill:  results of the misses go to their place in the batch and into the table (first empty
//     slot of the probe window, else the home slot is overwritten). Writers claim a slot by CAS on its
//     tag; readers only run in k_cache_probe, i.e. after a kernel boundary, and never see a slot mid-write
//     (the cache forces one group of games: one probe / fill pair in flight at a time).
// The maze is part of the key like in the reference (walls and mud are hashed, cached_backend.rs:130-199):
// with one shared maze the key is its pool offset; with generated mazes (one pool entry per slot, rewritten
// for every new game) it is the game the position belongs to -- only that game has that maze.
enum { CACHE_PROBES = 8, CACHE_EMPTY = 0, CACHE_VALID = 1, CACHE_BUSY = 2 };
template <int NW>
struct alignas(16) CacheEntry {
    State<NW> st;
    uint32_t maze_off;
    uint32_t tag;
    uint32_t maze_game;  // 0: shared maze; game index + 1 with one maze per game
    EvalOut ev;
};
template <int NW>
__device__ inline uint64_t position_hash(const State<NW>& st, uint32_t maze_off, uint32_t maze_game) {  // FNV-1a, cached_backend.rs:135-199
    uint64_t h = 0xcbf29ce484222325ULL;
    auto mix = [&](uint64_t v) {
        h ^= v;
        h *= 0x100000001b3ULL;
    };
    mix(st.p1);
    mix(st.p2);
    mix(__float_as_uint(st.s1));
    mix(__float_as_uint(st.s2));
    mix(st.m1);
    mix(st.m2);
    mix(st.turn);
    mix(maze_off);
    mix(maze_game);
    for (int w = 0; w < NW; ++w) mix(st.cheese[w]);
    return h;
}
template <int NW>
__device__ inline bool same_position(const State<NW>& a, const State<NW>& b) {
    bool eq = a.p1 == b.p1 && a.p2 == b.p2 && a.m1 == b.m1 && a.m2 == b.m2 && a.turn == b.turn &&
              __float_as_uint(a.s1) == __float_as_uint(b.s1) && __float_as_uint(a.s2) == __float_as_uint(b.s2);
    for (int w = 0; w < NW; ++w) eq = eq && a.cheese[w] == b.cheese[w];
    return eq;
}
template <int NW>
__global__ void k_cache_probe(const LeafReq<NW>* queue, const uint32_t* n_ptr, const Slot<NW>* slots,
                              const CacheEntry<NW>* table, uint64_t mask, EvalOut* ev_out, LeafReq<NW>* miss_queue,
                              uint32_t* miss_map, uint32_t* miss_count, unsigned long long* counters,
                              uint32_t maze_per_game) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < *n_ptr;
    LeafReq<NW> r = queue[in ? i : 0];
    bool hit = false;
    if (in) {
        const uint32_t maze_off = slots[r.slot].board.maze_off;
        const uint32_t maze_game = maze_per_game ? slots[r.slot].game_index + 1u : 0u;
        const uint64_t h = position_hash(r.st, maze_off, maze_game);
        for (int p = 0; p < CACHE_PROBES && !hit; ++p) {
            const CacheEntry<NW>& e = table[(h + (uint64_t)p) & mask];
            if (e.tag == CACHE_EMPTY) break;
            if (e.tag == CACHE_VALID && e.maze_off == maze_off && e.maze_game == maze_game && same_position(e.st, r.st)) {
                ev_out[i] = e.ev;
                hit = true;
            }
        }
    }
    // one atomic per wavefront for the counters and for the misses' places in the second queue
    const unsigned long long hits = __ballot(in && hit), misses = __ballot(in && !hit);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) {
        if (hits) atomicAdd(&counters[0], (unsigned long long)__popcll(hits));
        if (misses) {
            atomicAdd(&counters[1], (unsigned long long)__popcll(misses));
            base = atomicAdd(miss_count, (uint32_t)__popcll(misses));
        }
    }
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (in && !hit) {
        const uint32_t j = base + (uint32_t)__popcll(misses & ((1ULL << lane) - 1ULL));
        miss_queue[j] = r;
        miss_map[j] = i;
    }
}
template <int NW>
__global__ void k_cache_fill(const LeafReq<NW>* miss_queue, const uint32_t* miss_map, const uint32_t* miss_count,
                             const EvalOut* ev_miss, const Slot<NW>* slots, CacheEntry<NW>* table, uint64_t mask,
                             EvalOut* ev_out, uint32_t maze_per_game) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *miss_count) return;
    const LeafReq<NW> r = miss_queue[j];
    const EvalOut ev = ev_miss[j];
    ev_out[miss_map[j]] = ev;
    const uint32_t maze_off = slots[r.slot].board.maze_off;
    const uint32_t maze_game = maze_per_game ? slots[r.slot].game_index + 1u : 0u;
    const uint64_t h = position_hash(r.st, maze_off, maze_game);
    // first empty slot of the window; a position that is already there (or being written by a twin in
    // this batch) is left alone; a full window overwrites the home slot
    for (int p = 0; p <= CACHE_PROBES; ++p) {
        const bool evict = p == CACHE_PROBES;
        CacheEntry<NW>& e = table[(h + (uint64_t)(evict ? 0 : p)) & mask];
        const uint32_t tag = e.tag;
        if (!evict && tag == CACHE_VALID) {
            if (e.maze_off == maze_off && e.maze_game == maze_game && same_position(e.st, r.st)) return;
            continue;
        }
        if (!evict && tag == CACHE_BUSY) continue;
        if (atomicCAS(&e.tag, evict ? (uint32_t)CACHE_VALID : (uint32_t)CACHE_EMPTY, (uint32_t)CACHE_BUSY) !=
            (evict ? (uint32_t)CACHE_VALID : (uint32_t)CACHE_EMPTY)) {
            if (evict) return;
            continue;
        }
        e.st = r.st;
        e.maze_off = maze_off;
        e.maze_game = maze_game;
        e.ev = ev;
        __threadfence();
        atomicExch(&e.tag, (uint32_t)CACHE_VALID);
        return;
    }
}

// SmartUniformBackend (backend.rs:92-103) as an evaluator of the leaf queue: the split pipeline (k_gather8 -> this ->
// k_backup16) then serves uniform-prior runs too; the values are what the fused k_step_uniform computes inline
// (emit_proc, EVAL_UNIFORM).
template <int NW>
__global__ void k_uniform_eval(const LeafReq<NW>* queue, const uint32_t* n_ptr, const Slot<NW>* slots, const uint8_t* maze,
                               EvalOut* ev_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    const LeafReq<NW> r = queue[i];
    const uint8_t* cost = maze + slots[r.slot].board.maze_off;
    EvalOut o;
    uniform_prior(eff_actions(cost, r.st.p1, r.st.m1), o.p1);
    uniform_prior(eff_actions(cost, r.st.p2, r.st.m2), o.p2);
    o.v1 = 0.0f;
    o.v2 = 0.0f;
    ev_out[i] = o;
}

// host-callback evaluator support: leaves out, results in (single-slot searches)
template <int NW>
__global__ void k_export_leaves(const Slot<NW>* slots, uint32_t slot, Bases B, State<NW>* out, uint32_t* n_out) {
    const Slot<NW>& s = slots[slot];
    const Mem<NW> m = resolve_mem<NW>(s, B.arena, B.scratch, slot, B.L, B.maze);
    if (threadIdx.x == 0) *n_out = s.batch_active ? s.b_nn : 0;
    for (uint32_t j = threadIdx.x; j < s.b_nn; j += blockDim.x) out[j] = m.leaf_local[j];
}
template <int NW>
__global__ void k_import_evals(const Slot<NW>* slots, uint32_t slot, Bases B, const EvalOut* in, uint32_t n) {
    const Mem<NW> m = resolve_mem<NW>(slots[slot], B.arena, B.scratch, slot, B.L, B.maze);
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) m.ev_local[j] = in[j];
}

// ------------------------------------------------------------------------------------------------
// host: game generation (our own seeded sampler, DESIGN.md "game generation")
// ------------------------------------------------------------------------------------------------
namespace {

struct HostRng {  // same generator family as the device stream, host copy for game generation
    Rng r;
    explicit HostRng(uint64_t seed) { rng_seed(r, seed); }
    uint32_t below(uint32_t n) { return rng_below(r, n); }
};

struct HostGame {
    uint8_t width, height;
    uint16_t max_turns, turn;
    uint8_t p1, p2, m1, m2;
    float s1, s2;
    std::vector<uint8_t> cheese;  // [hw]
    uint16_t total_cheese;
    std::vector<uint8_t> cost;    // [hw * 4] this game's generated maze; empty = the run's shared maze
};

// 180-degree-symmetric cheese: shuffle the pair representatives (i < N-1-i, start cells excluded)
// with Fisher-Yates driven by gen_range, take count/2 pairs; an odd count uses the centre cell.
bool place_cheese(HostGame& g, uint32_t count, bool symmetric, uint64_t seed, std::string& err) {
    const int n = g.width * g.height;
    g.cheese.assign(n, 0);
    HostRng rng(seed);
    std::vector<int> cand;
    uint32_t need = count;
    if (symmetric) {
        int centre = -1;
        for (int i = 0; i < n; ++i) {
            const int j = n - 1 - i;
            if (i == g.p1 || i == g.p2 || j == g.p1 || j == g.p2) continue;
            if (i < j) cand.push_back(i);
            else if (i == j) centre = i;
        }
        if (need & 1u) {
            if (centre < 0) {
                err = "odd symmetric cheese count needs a free centre cell";
                return false;
            }
            g.cheese[centre] = 1;
            need -= 1;
        }
        if (need / 2 > cand.size()) {
            err = "cheese_count does not fit the board";
            return false;
        }
        for (int i = (int)cand.size() - 1; i >= 1; --i) std::swap(cand[i], cand[rng.below((uint32_t)i + 1)]);
        for (uint32_t k = 0; k < need / 2; ++k) {
            g.cheese[cand[k]] = 1;
            g.cheese[n - 1 - cand[k]] = 1;
        }
    } else {
        for (int i = 0; i < n; ++i)
            if (i != g.p1 && i != g.p2) cand.push_back(i);
        if (need > cand.size()) {
            err = "cheese_count does not fit the board";
            return false;
        }
        for (int i = (int)cand.size() - 1; i >= 1; --i) std::swap(cand[i], cand[rng.below((uint32_t)i + 1)]);
        for (uint32_t k = 0; k < need; ++k) g.cheese[cand[k]] = 1;
    }
    g.total_cheese = (uint16_t)count;
    return true;
}

std::vector<uint8_t> open_maze_cost(int w, int h) {
    std::vector<uint8_t> c((size_t)w * h * 4, 0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t* p = &c[((size_t)y * w + x) * 4];
            p[DIR_UP] = y + 1 < h;
            p[DIR_RIGHT] = x + 1 < w;
            p[DIR_DOWN] = y > 0;
            p[DIR_LEFT] = x > 0;
        }
    return c;
}

// Random walls + mud: the reference delegates to the pyrat-rust engine (absent, own RNG), so this is our
// own generator -- specification in DESIGN.md "game generation", second implementation in
// oracle/pyrat_engine.hpp::make_maze (the two are compared game by game in the parity tests).
std::vector<uint8_t> generate_maze(int w, int h, float wall_density, float mud_density, bool symmetric, uint64_t seed) {
    HostRng rng(seed ^ 0x6D617A65ULL);
    const int n = w * h;
    std::vector<std::pair<int, int>> ed;  // (a, b), a < b; per cell: RIGHT then UP
    std::vector<int> right_of(n, -1), up_of(n, -1);
    for (int i = 0; i < n; ++i) {
        if (i % w + 1 < w) {
            right_of[i] = (int)ed.size();
            ed.emplace_back(i, i + 1);
        }
        if (i / w + 1 < h) {
            up_of[i] = (int)ed.size();
            ed.emplace_back(i, i + w);
        }
    }
    const int ne = (int)ed.size();
    std::vector<int> image(ne);
    for (int k = 0; k < ne; ++k) {
        if (!symmetric) {
            image[k] = k;
            continue;
        }
        const int a = n - 1 - ed[k].second, b = n - 1 - ed[k].first;  // a < b again
        image[k] = b == a + 1 ? right_of[a] : up_of[a];
    }
    std::vector<uint8_t> val(ne, 1);
    const uint32_t wall_thr = (uint32_t)(wall_density * 16777216.0f), mud_thr = (uint32_t)(mud_density * 16777216.0f);
    for (int k = 0; k < ne; ++k)
        if (k <= image[k] && rng.below(1u << 24) < wall_thr) val[k] = val[image[k]] = 0;
    std::vector<int> up(n);
    for (int i = 0; i < n; ++i) up[i] = i;
    int comps = n;
    auto root = [&](int v) {
        while (up[v] != v) v = up[v] = up[up[v]];
        return v;
    };
    auto join = [&](int a, int b) {
        a = root(a);
        b = root(b);
        if (a == b) return;
        if (a < b) up[b] = a;
        else up[a] = b;
        --comps;
    };
    for (int k = 0; k < ne; ++k)
        if (val[k]) join(ed[k].first, ed[k].second);
    std::vector<int> closed;
    for (int k = 0; k < ne; ++k)
        if (k <= image[k] && !val[k]) closed.push_back(k);
    for (int i = (int)closed.size() - 1; i >= 1; --i) std::swap(closed[i], closed[rng.below((uint32_t)i + 1)]);
    for (int k : closed) {
        if (comps == 1) break;
        const int m = image[k];
        if (root(ed[k].first) == root(ed[k].second) && root(ed[m].first) == root(ed[m].second)) continue;
        val[k] = val[m] = 1;
        join(ed[k].first, ed[k].second);
        join(ed[m].first, ed[m].second);
    }
    for (int k = 0; k < ne; ++k)
        if (k <= image[k] && val[k] && rng.below(1u << 24) < mud_thr) val[k] = val[image[k]] = (uint8_t)(2 + rng.below(2));
    std::vector<uint8_t> c((size_t)n * 4, 0);
    for (int k = 0; k < ne; ++k) {
        const int a = ed[k].first, b = ed[k].second;
        const int d = b == a + 1 ? DIR_RIGHT : DIR_UP;
        c[(size_t)a * 4 + d] = val[k];
        c[(size_t)b * 4 + ((d + 2) & 3)] = val[k];
    }
    return c;
}

template <int NW>
void fill_state(const HostGame& g, Board& b, State<NW>& st, uint32_t maze_off) {
    b.width = g.width;
    b.height = g.height;
    b.max_turns = g.max_turns;
    b.total_cheese = g.total_cheese;
    b.maze_off = maze_off;
    for (int k = 0; k < NW; ++k) st.cheese[k] = 0;
    uint16_t rem = 0;
    for (size_t i = 0; i < g.cheese.size(); ++i)
        if (g.cheese[i]) {
            st.cheese[NW == 1 ? 0 : (i >> 6)] |= 1ULL << (i & 63);
            ++rem;
        }
    st.remaining = rem;
    st.s1 = g.s1;
    st.s2 = g.s2;
    st.turn = g.turn;
    st.p1 = g.p1;
    st.p2 = g.p2;
    st.m1 = g.m1;
    st.m2 = g.m2;
}

SearchCfg to_cfg(const ArSearchConfig& c, uint32_t sims, uint32_t batch) {
    SearchCfg s;
    s.c_puct = c.c_puct;
    s.fpu_reduction = c.fpu_reduction;
    s.force_k = c.force_k;
    s.noise_epsilon = c.noise_epsilon;
    s.noise_concentration = c.noise_concentration;
    s.coll_min = c.collision_limit_min;
    s.coll_max = c.collision_limit_max;
    s.coll_start = c.collision_scaling_start;
    s.coll_end = c.collision_scaling_end;
    s.coll_power = c.collision_scaling_power;
    s.n_sims = sims;
    s.batch_size = batch;
    s.alloc_per_round = 1;  // measured best on the bench workload; results do not depend on it
    if (const char* e = getenv("AR_ALLOC_PER_ROUND"))
        if (atoi(e) >= 1) s.alloc_per_round = (uint32_t)atoi(e);
    return s;
}

int check_cfg(const SearchCfg& c) {
    if (c.n_sims == 0) return fail(AR_E_INVALID, "simulations must be > 0");
    if (c.batch_size == 0 || c.batch_size > 4096) return fail(AR_E_INVALID, "batch_size must be in 1..4096");
    if (c.coll_max > 65536) return fail(AR_E_INVALID, "collision_limit_max must be <= 65536");
    if (c.noise_epsilon > 0.0f && !(c.noise_concentration / 5.0f > 1.0f))
        return fail(AR_E_INVALID, "noise_concentration must exceed 5 (Gamma shape >= 1 path only)");
    return AR_OK;
}

// ------------------------------------------------------------------------------------------------
// host: the engine -- device-resident slots and the step loop
// ------------------------------------------------------------------------------------------------
struct GameRecordHost {  // owning twin of ArGameRecordView
    uint8_t width, height;
    uint16_t max_turns;
    uint32_t game_index, n;
    std::vector<int8_t> maze;
    std::vector<uint8_t> initial_cheese, cheese_outcomes;
    float final_p1, final_p2;
    uint8_t result;
    uint16_t cheese_available;
    uint64_t sims, nn, term, coll;
    std::vector<uint8_t> p1_pos, p2_pos, p1_mud, p2_mud, cheese_mask, a1, a2;
    std::vector<float> p1_score, p2_score, v1, v2, vc1, vc2, pr1, pr2, po1, po2;
    std::vector<uint16_t> turn;
    ArGameRecordView view() const {
        ArGameRecordView v;
        memset(&v, 0, sizeof v);
        v.width = width;
        v.height = height;
        v.max_turns = max_turns;
        v.game_index = game_index;
        v.n_positions = n;
        v.maze = maze.data();
        v.initial_cheese = initial_cheese.data();
        v.cheese_outcomes = cheese_outcomes.data();
        v.final_p1_score = final_p1;
        v.final_p2_score = final_p2;
        v.result = result;
        v.cheese_available = cheese_available;
        v.total_simulations = sims;
        v.total_nn_evals = nn;
        v.total_terminals = term;
        v.total_collisions = coll;
        v.p1_pos = p1_pos.data();
        v.p2_pos = p2_pos.data();
        v.p1_score = p1_score.data();
        v.p2_score = p2_score.data();
        v.p1_mud = p1_mud.data();
        v.p2_mud = p2_mud.data();
        v.turn = turn.data();
        v.cheese_mask = cheese_mask.data();
        v.value_p1 = v1.data();
        v.value_p2 = v2.data();
        v.visit_counts_p1 = vc1.data();
        v.visit_counts_p2 = vc2.data();
        v.prior_p1 = pr1.data();
        v.prior_p2 = pr2.data();
        v.policy_p1 = po1.data();
        v.policy_p2 = po2.data();
        v.action_p1 = a1.data();
        v.action_p2 = a2.data();
        return v;
    }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        n = count;
        return hipMalloc((void**)&p, sizeof(T) * (count ? count : 1));
    }
    ~DevBuf() {
        if (p) hipFree(p);
    }
};
template <typename T>
struct PinBuf {
    T* p = nullptr;
    hipError_t alloc(size_t count) { return hipHostMalloc((void**)&p, sizeof(T) * (count ? count : 1)); }
    ~PinBuf() {
        if (p) hipHostFree(p);
    }
};

// The arena allocation (first arenas + overflow pool) is by far the largest one, and the driver
// clears device memory when it hands it out (~30 ms per GB measured, 6-7 s for a full MI355X). One
// block per device is therefore kept between calls: a training loop that calls ar_selfplay_run once
// per iteration pays for it once. ar_release_device_memory() (or AR_NO_ARENA_CACHE=1) gives it back.
struct ArenaBlock {
    unsigned char* p = nullptr;
    size_t bytes = 0;
};
enum { MAX_DEVICES = 64 };
static std::mutex g_arena_mu;
static ArenaBlock g_arena_cache[MAX_DEVICES];

static size_t arena_cached_bytes(int dev) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    return dev >= 0 && dev < MAX_DEVICES ? g_arena_cache[dev].bytes : 0;
}
static hipError_t arena_acquire(int dev, size_t bytes, ArenaBlock& out) {
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        if (dev >= 0 && dev < MAX_DEVICES) {
            ArenaBlock& c = g_arena_cache[dev];
            if (c.p && c.bytes >= bytes) {
                out = c;
                c = ArenaBlock();
                return hipSuccess;
            }
            if (c.p) {  // too small for this call: make room for the new one
                hipFree(c.p);
                c = ArenaBlock();
            }
        }
    }
    out.bytes = bytes;
    return hipMalloc((void**)&out.p, bytes);
}
static void arena_release(int dev, ArenaBlock b) {
    if (!b.p) return;
    if (!getenv("AR_NO_ARENA_CACHE") && dev >= 0 && dev < MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        ArenaBlock& c = g_arena_cache[dev];
        if (!c.p) {
            c = b;
            return;
        }
        if (c.bytes < b.bytes) std::swap(c, b);  // keep the bigger of the two
    }
    hipFree(b.p);
}
struct ArenaHold {
    unsigned char* p = nullptr;
    ArenaBlock blk;
    int dev = -1;
    hipError_t alloc(int device, size_t bytes) {
        dev = device;
        const hipError_t e = arena_acquire(device, bytes, blk);
        p = blk.p;
        return e;
    }
    ~ArenaHold() { arena_release(dev, blk); }
};

// Which gather kernel the network path uses when AR_GATHER does not say: measured on the bench workload (DESIGN.md
// section 7, profiles/r02_gather_ab.md) the eight-lanes-per-game kernel is faster per launch at every size -- 1.9x /
// 1.7x at 1024 / 8192 resident games, where its eight-fold wavefront count fills SIMDs the lane-per-game kernel leaves
// to one wavefront or none, and still ahead at 65536 once the shared maze sits in LDS and three wavefronts share a SIMD.
static bool default_gather8(uint32_t) { return true; }
// its register budget: two wavefronts per SIMD without spills up to 32768 games, three (168 VGPRs) above
static int default_gather8_wpe(uint32_t resident_games) { return resident_games <= 32768u ? 2 : 3; }
// measured (BASELINE config 2: 5x5, 1000 sims, 4096 games: 71.1 M vs 41.5 M simulations/s through the split pipeline;
// 7x7 / 1897 sims at 65536 games: 1180 M vs 1251 M): few games want k_gather8's wavefront count, many the fused kernel
static bool default_uniform_queue(uint32_t resident_games) { return resident_games <= 16384u; }

#if defined(AR_STATS)
static void* g_dbg_slots;
static uint32_t g_dbg_S, g_dbg_nw;
template <int NW>
__global__ void k_dbg_tree_hist(const Slot<NW>* slots, uint32_t n, unsigned long long* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Slot<NW>& s = slots[i];
    if (s.status == SLOT_EMPTY || s.status == SLOT_DONE) return;
    atomicAdd(&out[min(s.hi / 512u, 63u)], 1ULL);
    atomicAdd(&out[64 + min(s.cap / 512u, 63u)], 1ULL);
    atomicAdd(&out[128], (unsigned long long)s.hi);
    atomicAdd(&out[129], (unsigned long long)s.cap);
    atomicAdd(&out[130], 1ULL);
}
#endif
template <int NW>
struct Engine {
    int dev = 0;
    hipStream_t stream = nullptr;
    uint32_t S = 0;
    SearchCfg cfg;
    SlotLayout L;
    uint32_t max_turns = 0, cap0 = 0;
    DevBuf<Slot<NW>> slots;
    DevBuf<unsigned char> scratch;
    ArenaHold arena;
    DevBuf<uint8_t> maze;
    DevBuf<ZigTables> zig;
    DevBuf<uint32_t> counts, done_list, stall_list, release_list;
    DevBuf<StallInfo> stall_info;
    DevBuf<GameInit<NW>> init;
    DevBuf<DoneInfo<NW>> info;
    DevBuf<PosRec<NW>> staging;
    DevBuf<GrowReq> grow;
    DevBuf<LeafReq<NW>> queue;
    DevBuf<EvalOut> ev_queue;
    DevBuf<uint32_t> queue_count;
    PinBuf<uint32_t> h_counts, h_release;
    DevBuf<unsigned long long> live;  // k_scan's sums over the resident games (LIVE_N values)
    PinBuf<unsigned long long> h_live;
    PinBuf<StallInfo> h_stall;
    PinBuf<DoneInfo<NW>> h_info;
    PinBuf<PosRec<NW>> h_staging;
    std::vector<void*> slot_grown;  // per slot: the arena the host allocated after a stall (nullptr = none)
    // host-allocated arenas are recycled by size instead of going back to the driver (hipMalloc + hipFree of
    // a few MB cost ~0.3 ms together, thousands of times per run when the overflow pool runs dry)
    std::vector<uint32_t> slot_grown_cap;
    std::multimap<uint32_t, void*> spare_arenas;  // cap -> block
    void retire_arena(uint32_t slot) {
        if (slot_grown[slot]) spare_arenas.emplace(slot_grown_cap[slot], slot_grown[slot]);
        slot_grown[slot] = nullptr;
    }
    // generated mazes: one pool entry per slot, rewritten when a new game starts there (open mazes: one shared entry)
    bool per_slot_maze = false;
    uint32_t maze_stride = 0;
    DevBuf<uint8_t> maze_stage;
    DevBuf<uint32_t> maze_ids;
    // NN-evaluation cache (k_cache_probe / k_cache_fill); cache_entries == 0: off
    uint64_t cache_entries = 0;
    DevBuf<CacheEntry<NW>> cache_table;
    DevBuf<LeafReq<NW>> miss_queue;
    DevBuf<uint32_t> miss_map, miss_count;
    DevBuf<EvalOut> ev_miss;
    DevBuf<unsigned long long> cache_counters;
    ArenaPool pool = {};            // overflow blocks handed out by the kernels themselves
    uint32_t lanes = 64;  // games per wavefront in k_gather / k_backup
    uint32_t backup_lanes = 64;  // games per wavefront in the network path's k_backup (AR_BACKUP_LANES)
    bool backup16 = false;       // network path: the sixteen-lanes-per-game backup (k_backup16 + k_finish) instead of k_backup
    bool uniform_queue = false;  // SmartUniform through the split pipeline (leaf queue + k_uniform_eval) instead of k_step_uniform
    bool use_queue() const { return net != nullptr || uniform_queue; }
    bool gather8 = false; // network path: the eight-lanes-per-game gather (k_gather8) instead of k_gather
    int gather8_wpe = 2;  // its register budget: 2 wavefronts per SIMD (no spills) or 4 (128 VGPRs, spills to scratch)
    // network path: the gather as a work queue over tree levels (k_gatherw, dev_gatherw.h); a persistent grid of `gatherw_waves`
    bool gatherw = false;
    uint32_t gatherw_waves = 2048;
    // passes one launch may run; gathers that are not complete then are parked between two picks and go on in the next
    // launch. A game needs 50 passes in the median, 110 at the 90th and 160 at the 99th percentile, a few need 300+
    // (profiles/r03_gw_stats.txt): without a limit every launch lasts as long as its slowest game. Measured at 131072
    // resident games: 64 / 80 / 96 / 128 / 192 passes -> 638 / 646 / 643 / 615 / 598 M simulations/s; with the faster
    // evaluator and the late tree reuse of the round's end, over windows of five generations of games: 32 / 48 / 64 / 96 ->
    // 726 / 732 / 730 / 721 M (profiles/r03_sweeps.md; shorter windows sample one phase of the games and mislead).
    uint32_t gatherw_passes = 64;
    uint32_t gather_rounds = 0xFFFFFFFFu;  // rounds one k_gather launch may run per lane (self-play sets a limit)
    size_t region_bytes = 0;
    uint32_t region_low_mb = 0xFFFFFFFFu;  // least the region had left: MB never carved + MB in free blocks
    DevBuf<unsigned long long> pool_bits;
    DevBuf<int> pool_ctr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_order = nullptr;  // ev_order: side streams start after the main stream's refills
    double device_ms = 0.0;
    uint64_t steps = 0;
    uint64_t grows = 0;
    bool timed = false;
    ArNet* net = nullptr;

    ~Engine() {
        if (stream) hipStreamSynchronize(stream);
        for (Group& g : groups) {
            if (g.stream && g.stream != stream) {
                hipStreamSynchronize(g.stream);
                hipStreamDestroy(g.stream);
            }
            if (g.adv_stream) {
                hipStreamSynchronize(g.adv_stream);
                hipStreamDestroy(g.adv_stream);
            }
            if (g.done) hipEventDestroy(g.done);
            if (g.backed_up) hipEventDestroy(g.backed_up);
            if (g.gathered) hipEventDestroy(g.gathered);
            if (g.adv_ev[0]) hipEventDestroy(g.adv_ev[0]);
            if (g.adv_ev[1]) hipEventDestroy(g.adv_ev[1]);
            if (g.adv_done) hipEventDestroy(g.adv_done);
        }
        for (void* p : slot_grown)
            if (p) hipFree(p);
        for (auto& kv : spare_arenas) hipFree(kv.second);
        for (hipEvent_t e : gather_ev) hipEventDestroy(e);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
        if (ev_order) hipEventDestroy(ev_order);
        if (stream) hipStreamDestroy(stream);
    }

    Bases bases() const {
        Bases b;
        b.arena = arena.p;
        b.scratch = scratch.p;
        b.maze = maze.p;
        b.maze_stage = (!per_slot_maze && maze.n > 0 && maze.n <= MAZE_STAGE_BYTES && maze.n % 4 == 0) ? (uint32_t)maze.n : 0u;
        b.L = L;
        b.pool = pool;
        return b;
    }

    // bytes of device memory setup() takes besides the arenas, per resident game
    static size_t per_game_overhead(const SearchCfg& c, uint32_t mt, bool need_queue) {
        size_t b = make_layout<NW>(c, mt).total + sizeof(Slot<NW>) + sizeof(PosRec<NW>) * mt + sizeof(GameInit<NW>) +
                   sizeof(DoneInfo<NW>) + sizeof(StallInfo) + sizeof(GrowReq) + 16;
        if (need_queue) b += (size_t)c.batch_size * (sizeof(LeafReq<NW>) + sizeof(EvalOut));
        return b;
    }

    // `tree_bytes`: device memory for the trees (0 = one smallest block per game: every growth goes through the host)
    int setup(int device, uint32_t n_slots, const SearchCfg& c, uint32_t mt, const std::vector<uint8_t>& maze_bytes,
              uint32_t arena_nodes, bool need_queue, size_t tree_bytes = 0) {
        dev = device;
        S = n_slots;
        cfg = c;
        max_turns = mt;
        HIP_TRY(hipSetDevice(dev));
        HIP_TRY(hipStreamCreate(&stream));
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventCreateWithFlags(&ev_order, hipEventDisableTiming));
        L = make_layout<NW>(cfg, max_turns);
        static_assert(sizeof(LevelO<NW>) <= (16 + 112 + sizeof(State<NW>) + 15) / 16 * 16, "slot_layout.h sizes the level stack");
        // which gather kernel walks the trees of the network path (results are identical): AR_GATHER=lane | octet
        gather8 = default_gather8(S);
        gather8_wpe = default_gather8_wpe(S);
        // which backup kernel (results are identical): AR_BACKUP=lane | group; the group kernel keeps 64 paths in LDS
        // (the deep part of its paths lives in the level-stack scratch: 16 lanes x path_cap steps must fit there)
        backup16 = (size_t)16 * (L.max_depth + 2) * sizeof(PathStep) <= (size_t)L.max_depth * (16 + 112 + sizeof(State<NW>));
        if (const char* e = getenv("AR_BACKUP")) backup16 = backup16 && std::string(e) != "lane";
        if (const char* e = getenv("AR_GATHER")) gather8 = std::string(e).rfind("octet", 0) == 0;
        if (const char* e = getenv("AR_GATHER"))
            if (std::string(e).rfind("octet", 0) == 0)
                gather8_wpe = std::string(e) == "octet4" ? 4 : std::string(e) == "octet3" ? 3 : std::string(e) == "octet2" ? 2 : gather8_wpe;
        // the work-queue gather serves batches of up to GW_SLOTS descents; with uniform priors nearly every allocation step
        // draws a tie break, which puts its items in sequence again: those runs keep the eight-lane kernel
        gatherw = need_queue && net != nullptr && cfg.batch_size <= GW_SLOTS;
        if (const char* e = getenv("AR_GATHER")) gatherw = std::string(e) == "wide" && need_queue && cfg.batch_size <= GW_SLOTS;
        if (gatherw) {
            int cus = 0;
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            // two wavefronts per SIMD (212 registers each), 32 (16 on boards above 64 cells) games each -- less one per CU: the
            // grid is persistent, and the other group's small kernels (k_finish, k_backup: 130-220 registers) then find
            // a SIMD with room instead of waiting for a gather wavefront to end (655 vs 643 M simulations/s)
            gatherw_waves = (uint32_t)(cus > 0 ? cus : 256) * 7u;
            if (const char* e = getenv("AR_GW_WAVES"))
                if (atoi(e) > 0) gatherw_waves = (uint32_t)atoi(e);
            if (const char* e = getenv("AR_GW_PASSES")) gatherw_passes = atoi(e) > 0 ? (uint32_t)atoi(e) : 0xFFFFFFFFu;  // 0: no limit
        }
        cap0 = arena_nodes ? (uint32_t)align_up(arena_nodes, 64) : initial_arena_nodes(cfg);  // (arenas are whole 256-byte units)
        slot_grown.assign(S, nullptr);
        slot_grown_cap.assign(S, 0u);
        HIP_TRY(slots.alloc(S));
        HIP_TRY(hipMemsetAsync(slots.p, 0, sizeof(Slot<NW>) * S, stream));
#if defined(AR_STATS)
        g_dbg_slots = slots.p;
        g_dbg_S = S;
        g_dbg_nw = NW;
#endif
        HIP_TRY(scratch.alloc((size_t)S * L.total));
        if ((size_t)S * L.total >= ((size_t)1 << 32)) gather8 = false;  // k_gather8 addresses scratch with 32-bit offsets
        // the tree region: pages, a bitmap per zone of 1024 slots; at least a fresh game's pages for every slot
        size_t region = tree_bytes;
        {
            pool.page_nodes = arena_nodes ? cap0 : (uint32_t)POOL_PAGE_NODES;  // (test knob: small pages, fresh games on a single one)
            pool.fresh_pages = arena_nodes ? 1u : (initial_arena_nodes(cfg) + POOL_PAGE_NODES - 1) / POOL_PAGE_NODES;
            if (pool.fresh_pages > POOL_WORD_PAGES) return fail(AR_E_INVALID, "simulations per move beyond what one run of tree pages holds");
            cap0 = pool.fresh_pages * pool.page_nodes;
            pool.page_bytes = arena_bytes(pool.page_nodes);
            const size_t word_bytes = (size_t)POOL_WORD_PAGES * pool.page_bytes;
            pool.zones = (S + POOL_ZONE_SLOTS - 1) / POOL_ZONE_SLOTS;
            {
                // (every zone holds a fresh arena per slot of a full zone, with room for runs that do not pack a word)
                const size_t floor_b = (size_t)(S < POOL_ZONE_SLOTS ? S : (uint32_t)POOL_ZONE_SLOTS) * pool.fresh_pages * pool.page_bytes * 5 / 4 + word_bytes;
                const size_t share = region / pool.zones;
                const size_t zb = share > floor_b ? share : floor_b;
                pool.zone_words = (uint32_t)((zb + word_bytes - 1) / word_bytes);
            }
            pool.zone_bytes = (unsigned long long)pool.zone_words * word_bytes;
            region = (size_t)pool.zone_bytes * pool.zones;
            region_bytes = region;
            pool.ret_cap = 2 * POOL_ZONE_SLOTS + 64;
            HIP_TRY(pool_bits.alloc((size_t)pool.zones * pool.zone_words + (size_t)pool.zones * pool.ret_cap));
            HIP_TRY(hipMemsetAsync(pool_bits.p, 0, 8 * (size_t)pool.zones * pool.zone_words, stream));
            HIP_TRY(pool_ctr.alloc(pool.zones + 2));
            HIP_TRY(hipMemsetAsync(pool_ctr.p, 0, sizeof(int) * (pool.zones + 2), stream));
            pool.bits = pool_bits.p;
            pool.ret = pool_bits.p + (size_t)pool.zones * pool.zone_words;
            pool.ret_n = (uint32_t*)pool_ctr.p;
        }
        const size_t pool_end = region;
        HIP_TRY(arena.alloc(dev, pool_end + 256));
        HIP_TRY(maze.alloc(maze_bytes.size()));
        HIP_TRY(hipMemcpyAsync(maze.p, maze_bytes.data(), maze_bytes.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(zig.alloc(1));
        static const ZigTables host_zig = {AR_ZIG_NORM_X_INIT, AR_ZIG_NORM_F_INIT};
        HIP_TRY(hipMemcpyAsync(zig.p, &host_zig, sizeof host_zig, hipMemcpyHostToDevice, stream));
        HIP_TRY(counts.alloc(8));
        HIP_TRY(release_list.alloc(S));
        HIP_TRY(h_release.alloc(S));
        HIP_TRY(done_list.alloc(S));
        HIP_TRY(stall_list.alloc(S));
        HIP_TRY(stall_info.alloc(S));
        HIP_TRY(init.alloc(S));
        HIP_TRY(info.alloc(S));
        HIP_TRY(staging.alloc((size_t)S * max_turns));
        HIP_TRY(grow.alloc(S));
        HIP_TRY(h_counts.alloc(8));
        HIP_TRY(live.alloc(16));
        HIP_TRY(h_live.alloc(16));
        HIP_TRY(h_stall.alloc(S));
        HIP_TRY(h_info.alloc(S));
        HIP_TRY(h_staging.alloc((size_t)S * max_turns));
        if (need_queue) {
            HIP_TRY(queue.alloc((size_t)S * cfg.batch_size));
            HIP_TRY(ev_queue.alloc((size_t)S * cfg.batch_size));
            HIP_TRY(queue_count.alloc(64));
        }
        if (need_queue && cache_entries) {
            HIP_TRY(cache_table.alloc(cache_entries));
            HIP_TRY(hipMemsetAsync(cache_table.p, 0, sizeof(CacheEntry<NW>) * cache_entries, stream));
            HIP_TRY(miss_queue.alloc((size_t)S * cfg.batch_size));
            HIP_TRY(miss_map.alloc((size_t)S * cfg.batch_size));
            HIP_TRY(ev_miss.alloc((size_t)S * cfg.batch_size));
            HIP_TRY(miss_count.alloc(64));
            HIP_TRY(cache_counters.alloc(2));
            HIP_TRY(hipMemsetAsync(cache_counters.p, 0, 16, stream));
        }
        if (net != nullptr) {
            const size_t per = (size_t)net->dev.hw * 4;
            if (maze_bytes.size() % per != 0) return fail(AR_E_INVALID, "maze pool does not match the network's board size");
            if (int rc = net_bind_mazes(net, maze.p, (int)(maze_bytes.size() / per), stream)) return rc;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return AR_OK;
    }

    uint32_t grid(uint32_t n) const { return (n + 63) / 64; }

    // `mazes`: games.size() x maze_stride bytes when per_slot_maze
    int start_games(std::vector<GameInit<NW>>& games, const std::vector<uint8_t>* mazes = nullptr) {
        if (games.empty()) return AR_OK;
        for (GameInit<NW>& g : games) {  // a new game starts in the slot's first arena again
            g.reset_arena = 0;
            if (slot_grown[g.slot]) {
                retire_arena(g.slot);  // the slot was drained: no kernel touches that arena any more
                g.reset_arena = 1;
            }
        }
        HIP_TRY(hipMemcpyAsync(init.p, games.data(), sizeof(GameInit<NW>) * games.size(), hipMemcpyHostToDevice, stream));
        if (per_slot_maze) {
            if (!mazes || mazes->size() != games.size() * (size_t)maze_stride) return fail(AR_E_INVALID, "internal: mazes missing");
            if (!maze_stage.p) {
                HIP_TRY(maze_stage.alloc((size_t)S * maze_stride));
                HIP_TRY(maze_ids.alloc(S));
            }
            HIP_TRY(hipMemcpyAsync(maze_stage.p, mazes->data(), mazes->size(), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_place_mazes<NW>, dim3((uint32_t)games.size()), dim3(64), 0, stream, init.p,
                               (uint32_t)games.size(), maze_stage.p, maze_stride, maze.p, maze_ids.p);
            if (net != nullptr)
                if (int rc = net_rebind_mazes(net, maze_ids.p, (int)games.size(), stream)) return rc;
        }
        hipLaunchKernelGGL(k_init_games<NW>, dim3(grid((uint32_t)games.size())), dim3(64), 0, stream, slots.p, init.p,
                           (uint32_t)games.size(), bases(), cfg);
        HIP_TRY(hipGetLastError());
        // init.p is reused by the next call: wait for the copy + kernel
        HIP_TRY(hipStreamSynchronize(stream));
        return AR_OK;
    }

    // Games are split into `n_groups` contiguous slot ranges, each with its own stream and evaluator
    // queue. The groups run the same gather -> evaluate -> backup -> advance sequence independently, so
    // the compute-bound network kernel of one group overlaps the latency-bound tree walks of the others.
    // Tree reuse (k_advance) runs on a side stream per group: its duration is that of the one slowest
    // compaction, and the games it works on sit out the next step anyway, so the next gather does not
    // wait for it (ownership of a slot passes through its status word, see tag_status).
    struct Group {
        hipStream_t stream = nullptr, adv_stream = nullptr;
        hipEvent_t done = nullptr, backed_up = nullptr, adv_done = nullptr, adv_ev[2] = {nullptr, nullptr};
        hipEvent_t gathered = nullptr;  // the group's last gather launch is complete
        uint32_t first = 0, end = 0;  // slots [first, end)
        uint64_t step = 0;
        bool adv_pending = false;  // (late tree reuse) the last step's k_advance has not been launched yet
        uint32_t adv_phase = 0;
    };
    // HIP events around every k_gather launch of group 0 (on the stream it runs on); read back in scan()
    std::vector<hipEvent_t> gather_ev;
    size_t gather_ev_used = 0;
    double gather_ms = 0.0;
    uint64_t gather_launches = 0;
    bool overlap_advance = true;
    bool merge_per_group = false;
    bool stagger_gathers = false;
    int cu_split = 0;  // 1: groups on the low / high half of the CU mask bits, 2: even / odd bits
    bool adv_late = true;   // AR_ADV_LATE=0: the tree reuse beside the group's next tree walk, as before
    std::vector<Group> groups;
    int make_groups(uint32_t n) {
        if (n < 1) n = 1;
        if (n > S) n = S;
        groups.resize(n);
        // groups of whole pool zones when there are enough games: a group's tree reuse then hands blocks out and takes
        // them back in its own zones only, and can put returned blocks back on the free stacks at every step
        merge_per_group = S / n >= POOL_ZONE_SLOTS;
        auto cut = [&](uint32_t g) {
            uint32_t at = (uint32_t)((uint64_t)S * g / n);
            if (merge_per_group && g > 0 && g < n) at = at / POOL_ZONE_SLOTS * POOL_ZONE_SLOTS;
            return at;
        };
        for (uint32_t g = 0; g < n; ++g) {
            groups[g].first = cut(g);
            groups[g].end = cut(g + 1);
            // group 0 runs on the engine's own stream: with two groups that makes four streams (two step streams, two
            // tree-reuse side streams), which is what the runtime maps to hardware queues by default (GPU_MAX_HW_QUEUES
            // = 4); a fifth stream shares a queue with another and the pipeline collapses (measured: 321 M against
            // 565 M simulations/s)
            HIP_TRY(hipEventCreateWithFlags(&groups[g].gathered, hipEventDisableTiming));
            // (AR_CUMASK=lohi|evenodd, off by default) every group on its own half of the compute units: the streams of
            // group g are created with a CU mask, so that kernels of different groups never share a CU
            uint32_t mask[8];
            const bool masked = cu_split != 0 && n == 2;
            for (int w = 0; w < 8; ++w)
                mask[w] = cu_split == 2 ? (g == 0 ? 0x55555555u : 0xAAAAAAAAu) : ((w < 4) == (g == 0) ? 0xFFFFFFFFu : 0u);
            if (g > 0 || masked) {
                if (masked)
                    HIP_TRY(hipExtStreamCreateWithCUMask(&groups[g].stream, 8, mask));
                else
                    HIP_TRY(hipStreamCreateWithFlags(&groups[g].stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&groups[g].done, hipEventDisableTiming));
            } else {
                groups[g].stream = stream;
            }
            if (overlap_advance) {
                if (masked)
                    HIP_TRY(hipExtStreamCreateWithCUMask(&groups[g].adv_stream, 8, mask));
                else
                    HIP_TRY(hipStreamCreateWithFlags(&groups[g].adv_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&groups[g].backed_up, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&groups[g].adv_done, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&groups[g].adv_ev[0], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&groups[g].adv_ev[1], hipEventDisableTiming));
            }
        }
        return AR_OK;
    }
    int group_step(Group& g) {
        const uint32_t n = g.end - g.first, gi = (uint32_t)(&g - groups.data());
        const bool side = g.adv_stream != nullptr;
        const uint32_t phase = side ? (uint32_t)(g.step & 1) : 0u;
        const uint32_t ready = side ? (uint32_t)SLOT_READY_A + phase : (uint32_t)SLOT_ACTIVE;
        if (side) HIP_TRY(hipStreamWaitEvent(g.stream, g.adv_ev[phase], 0));  // advance(step - 2) is complete
        LeafReq<NW>* q = queue.p + (size_t)g.first * cfg.batch_size;
        EvalOut* ev = ev_queue.p + (size_t)g.first * cfg.batch_size;
        uint32_t* qc = queue_count.p + gi;
        HIP_TRY(hipMemsetAsync(qc, 0, 4, g.stream));
        // (AR_STAGGER=1, off by default) One gather at a time: a group's tree walk starts when the previous group's (in launch
        // order) has ended. Left to themselves the groups drift into step -- both walk, then both evaluate (kernel trace,
        // profiles/r03_timeline_131k.txt) -- so stages do not overlap across groups. Held apart they do not overlap either:
        // the persistent gather fills every SIMD's registers (2 x 212 of 512) and the evaluator's wavefronts (141) wait for
        // it to end: 588 vs 643 M simulations/s, also with half the gather wavefronts (570 M).
        if (stagger_gathers && groups.size() > 1)
            HIP_TRY(hipStreamWaitEvent(g.stream, groups[(gi + groups.size() - 1) % groups.size()].gathered, 0));
        const bool timed_launch = true;  // every group's gather launch is timed on its own stream
        if (timed_launch) {
            while (gather_ev.size() < gather_ev_used + 2) {
                hipEvent_t e = nullptr;
                HIP_TRY(hipEventCreate(&e));
                gather_ev.push_back(e);
            }
            HIP_TRY(hipEventRecord(gather_ev[gather_ev_used], g.stream));
        }
        if (gatherw) {
            // a persistent grid: at most gatherw_waves wavefronts, and at small sizes about four games per wavefront
            constexpr int GWG = NW == 1 ? 32 : 16;
            uint32_t waves = (n + 3) / 4;
            if (waves > gatherw_waves) waves = gatherw_waves;
            const uint32_t n_sets = (n + GWG - 1) / GWG, second_turn = n_sets > waves ? n_sets - waves : 0u;
            const uint32_t rot = second_turn ? (uint32_t)((g.step * (uint64_t)second_turn) % n_sets) : 0u;
            hipLaunchKernelGGL((k_gatherw<NW, GWG, 4>), dim3(waves), dim3(64), (size_t)bases().maze_stage, g.stream, slots.p, g.end, cfg,
                               bases(), g.first, phase, ready, gatherw_passes, rot);
        } else if (gather8 && gather8_wpe == 3)
            hipLaunchKernelGGL((k_gather8<NW, 3>), dim3((n + 7) / 8), dim3(64), 0, g.stream, slots.p, g.end, cfg, bases(), q,
                               qc, g.first, phase, ready);
        else if (gather8 && gather8_wpe == 4)
            hipLaunchKernelGGL((k_gather8<NW, 4>), dim3((n + 7) / 8), dim3(64), 0, g.stream, slots.p, g.end, cfg, bases(), q,
                               qc, g.first, phase, ready);
        else if (gather8)
            hipLaunchKernelGGL((k_gather8<NW, 2>), dim3((n + 7) / 8), dim3(64), 0, g.stream, slots.p, g.end, cfg, bases(), q,
                               qc, g.first, phase, ready);
        else
            hipLaunchKernelGGL(k_gather<NW>, dim3((n + lanes - 1) / lanes), dim3(64), 0, g.stream, slots.p, g.end, cfg,
                               bases(), q, qc, gather_rounds, lanes, g.first, phase, ready);
        if (timed_launch) {
            HIP_TRY(hipEventRecord(gather_ev[gather_ev_used + 1], g.stream));
            gather_ev_used += 2;
        }
        HIP_TRY(hipEventRecord(g.gathered, g.stream));
        if (g.adv_pending)
            if (int rc = launch_late_advance(g)) return rc;
        if (gatherw)
            hipLaunchKernelGGL(k_pack_leaves<NW>, dim3((n + 63) / 64), dim3(64), 0, g.stream, slots.p, g.end, bases(), g.first, q, qc);
        const uint32_t n_max = (uint32_t)((size_t)n * cfg.batch_size);
        if (cache_entries && net != nullptr) {
            LeafReq<NW>* mq = miss_queue.p + (size_t)g.first * cfg.batch_size;
            uint32_t* mm = miss_map.p + (size_t)g.first * cfg.batch_size;
            EvalOut* em = ev_miss.p + (size_t)g.first * cfg.batch_size;
            uint32_t* mc = miss_count.p + gi;
            HIP_TRY(hipMemsetAsync(mc, 0, 4, g.stream));
            hipLaunchKernelGGL(k_cache_probe<NW>, dim3((n_max + 255) / 256), dim3(256), 0, g.stream, q, qc, slots.p,
                               cache_table.p, cache_entries - 1, ev, mq, mm, mc, cache_counters.p, per_slot_maze ? 1u : 0u);
            if (int rc = net_forward_queue<NW>(net, mq, mc, n_max, slots.p, maze.p, em, g.stream)) return rc;
            hipLaunchKernelGGL(k_cache_fill<NW>, dim3((n_max + 255) / 256), dim3(256), 0, g.stream, mq, mm, mc, em, slots.p,
                               cache_table.p, cache_entries - 1, ev, per_slot_maze ? 1u : 0u);
        } else if (net == nullptr) {
            hipLaunchKernelGGL(k_uniform_eval<NW>, dim3((n_max + 255) / 256), dim3(256), 0, g.stream, q, qc, slots.p, maze.p, ev);
        } else if (int rc = net_forward_queue<NW>(net, q, qc, n_max, slots.p, maze.p, ev, g.stream)) {
            return rc;
        }
        if (backup16) {
            const uint32_t path_cap = L.max_depth + 2;
            hipLaunchKernelGGL(k_backup16<NW>, dim3((n + 3) / 4), dim3(64), (size_t)64 * PATH_LDS_STEPS * sizeof(PathStep),
                               g.stream, slots.p, g.end, cfg, bases(), zig.p, ev, g.first, path_cap);
            hipLaunchKernelGGL(k_finish<NW>, dim3(grid(n)), dim3(64), 0, g.stream, slots.p, g.end, cfg, bases(), g.first, phase);
            // (fresh roots that draw Dirichlet noise were left alone above; k_finish ignores a slot with a batch pending)
            if (cfg.noise_epsilon > 0.0f)
                hipLaunchKernelGGL(k_backup<NW>, dim3((n + 63) / 64), dim3(64), 0, g.stream, slots.p, g.end, cfg, bases(), zig.p,
                                   ev, 64u, g.first, phase);
        } else
            hipLaunchKernelGGL(k_backup<NW>, dim3((n + backup_lanes - 1) / backup_lanes), dim3(64), 0, g.stream, slots.p, g.end,
                               cfg, bases(), zig.p, ev, backup_lanes, g.first, phase);
        // blocks the group's trees left at the last step go back on the free stacks (nothing reads them any more, and
        // this stream is the only one that pops or returns in the group's zones)
        if (side && adv_late) {
            // The tree reuse of this step is launched behind the NEXT gather of the group (launch_late_advance),
            // so that it runs beside the evaluator, which leaves the memory system alone, instead of beside the tree walk,
            // which it slows down by a fifth (profiles/r03_sweeps.md: the gather launch with and without k_advance beside it)
            HIP_TRY(hipEventRecord(g.backed_up, g.stream));
            g.adv_pending = true;
            g.adv_phase = phase;
            g.step += 1;
            return AR_OK;
        }
        if (merge_per_group && pool.bits) {
            const uint32_t z0 = g.first / POOL_ZONE_SLOTS, z1 = (g.end + POOL_ZONE_SLOTS - 1) / POOL_ZONE_SLOTS;
            if (side) {
                HIP_TRY(hipEventRecord(g.backed_up, g.stream));
                HIP_TRY(hipStreamWaitEvent(g.adv_stream, g.backed_up, 0));
            }
            hipLaunchKernelGGL(k_pool_merge, dim3(z1 - z0), dim3(64), 0, side ? g.adv_stream : g.stream, pool, z0, z1 - z0);
        }
        if (side) {
            HIP_TRY(hipEventRecord(g.backed_up, g.stream));
            HIP_TRY(hipStreamWaitEvent(g.adv_stream, g.backed_up, 0));
            hipLaunchKernelGGL(k_advance<NW>, dim3(n), dim3(64), 0, g.adv_stream, slots.p, g.end, bases(), cfg, g.first,
                               phase, ready);
            HIP_TRY(hipEventRecord(g.adv_ev[phase], g.adv_stream));
        } else {
            hipLaunchKernelGGL(k_advance<NW>, dim3(n), dim3(64), 0, g.stream, slots.p, g.end, bases(), cfg, g.first, 0u,
                               (uint32_t)SLOT_ACTIVE);
        }
        g.step += 1;
        return AR_OK;
    }

    // tree reuse of the group's previous step, on the side stream: after that step's backup and (when a gather has been
    // launched since) after that gather
    int launch_late_advance(Group& g) {
        const uint32_t n = g.end - g.first;
        HIP_TRY(hipStreamWaitEvent(g.adv_stream, g.backed_up, 0));
        HIP_TRY(hipStreamWaitEvent(g.adv_stream, g.gathered, 0));
        if (merge_per_group && pool.bits) {
            const uint32_t z0 = g.first / POOL_ZONE_SLOTS, z1 = (g.end + POOL_ZONE_SLOTS - 1) / POOL_ZONE_SLOTS;
            hipLaunchKernelGGL(k_pool_merge, dim3(z1 - z0), dim3(64), 0, g.adv_stream, pool, z0, z1 - z0);
        }
        hipLaunchKernelGGL(k_advance<NW>, dim3(n), dim3(64), 0, g.adv_stream, slots.p, g.end, bases(), cfg, g.first, g.adv_phase,
                           (uint32_t)SLOT_READY_A + g.adv_phase);
        HIP_TRY(hipEventRecord(g.adv_ev[g.adv_phase], g.adv_stream));
        g.adv_pending = false;
        return AR_OK;
    }

    void launch_gather(bool to_queue) {
        hipLaunchKernelGGL(k_gather<NW>, dim3((S + lanes - 1) / lanes), dim3(64), 0, stream, slots.p, S, cfg, bases(),
                           to_queue ? queue.p : (LeafReq<NW>*)nullptr, to_queue ? queue_count.p : (uint32_t*)nullptr,
                           gather_rounds, lanes, 0u, 0u, (uint32_t)SLOT_ACTIVE);
    }
    void launch_backup(bool from_queue) {
        hipLaunchKernelGGL(k_backup<NW>, dim3((S + lanes - 1) / lanes), dim3(64), 0, stream, slots.p, S, cfg, bases(),
                           zig.p, from_queue ? ev_queue.p : (const EvalOut*)nullptr, lanes, 0u, 0u);
    }
    void launch_advance() {
        if (pool.bits) hipLaunchKernelGGL(k_pool_merge, dim3(pool.zones), dim3(64), 0, stream, pool, 0u, pool.zones);
        hipLaunchKernelGGL(k_advance<NW>, dim3(S), dim3(64), 0, stream, slots.p, S, bases(), cfg, 0u, 0u, (uint32_t)SLOT_ACTIVE);
    }
    void launch_cancel() { hipLaunchKernelGGL(k_cancel<NW>, dim3(grid(S)), dim3(64), 0, stream, slots.p, S, bases()); }

    // `n_launch` rounds of {`iters` simulate_batch per game, then tree reuse for the games that moved},
    // timed with HIP events on our stream
    int run_steps(int n_launch, int iters) {
        if (!timed) HIP_TRY(hipEventRecord(ev0, stream));  // several calls between two scans are timed as one interval
        for (int k = 0; k < n_launch; ++k) {
            if (!use_queue()) {
                hipLaunchKernelGGL(k_step_uniform<NW>, dim3(grid(S)), dim3(64), 0, stream, slots.p, S, cfg, bases(), zig.p,
                                   iters);
                launch_advance();
                steps += (uint64_t)iters;
            } else {
                steps += (uint64_t)iters;
            }
        }
        if (use_queue()) {
            if (groups.empty())
                if (int rc = make_groups(1)) return rc;
            HIP_TRY(hipEventRecord(ev_order, stream));
            for (const Group& g : groups) {
                if (g.stream != stream) HIP_TRY(hipStreamWaitEvent(g.stream, ev_order, 0));
                if (g.adv_stream) HIP_TRY(hipStreamWaitEvent(g.adv_stream, ev_order, 0));
            }
            for (int k = 0; k < n_launch * iters; ++k)
                for (Group& g : groups)
                    if (int rc = group_step(g)) return rc;
            for (Group& g : groups)
                if (g.adv_pending)
                    if (int rc = launch_late_advance(g)) return rc;
            for (const Group& g : groups) {
                if (g.stream != stream) {
                    HIP_TRY(hipEventRecord(g.done, g.stream));
                    HIP_TRY(hipStreamWaitEvent(stream, g.done, 0));
                }
                if (g.adv_stream) {  // the host's scan sees every slot at rest
                    HIP_TRY(hipEventRecord(g.adv_done, g.adv_stream));
                    HIP_TRY(hipStreamWaitEvent(stream, g.adv_done, 0));
                }
            }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev1, stream));
        timed = true;
        return AR_OK;
    }

    // after run_steps: status lists on the host. Also accumulates device time.
    int scan(uint32_t out_counts[4]) {
        if (pool.bits) hipLaunchKernelGGL(k_pool_merge, dim3(pool.zones), dim3(64), 0, stream, pool, 0u, pool.zones);
        HIP_TRY(hipMemsetAsync(counts.p, 0, 32, stream));
        HIP_TRY(hipMemsetAsync(live.p, 0, 8 * 16, stream));
        hipLaunchKernelGGL(k_scan<NW>, dim3(grid(S)), dim3(64), 0, stream, slots.p, S, counts.p, done_list.p,
                           stall_list.p, release_list.p, pool, live.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_counts.p, counts.p, 32, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_live.p, live.p, 8 * 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (const uint32_t n_rel = h_counts.p[4]) {  // slots that moved back to their pool share
            HIP_TRY(hipMemcpyAsync(h_release.p, release_list.p, 4 * n_rel, hipMemcpyDeviceToHost, stream));
            hipLaunchKernelGGL(k_clear_release<NW>, dim3(grid(n_rel)), dim3(64), 0, stream, slots.p, release_list.p, n_rel);
            HIP_TRY(hipStreamSynchronize(stream));
            for (uint32_t i = 0; i < n_rel; ++i) retire_arena(h_release.p[i]);
        }
        if (timed) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) device_ms += ms;
            timed = false;
        }
        for (size_t i = 0; i + 1 < gather_ev_used; i += 2) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, gather_ev[i], gather_ev[i + 1]) == hipSuccess) {
                gather_ms += ms;
                gather_launches += 1;
            }
        }
        gather_ev_used = 0;
        for (int i = 0; i < 4; ++i) out_counts[i] = h_counts.p[i];
        {
            const uint32_t left_mb = (uint32_t)(((unsigned long long)h_counts.p[5] * pool.page_bytes) >> 20);
            if (left_mb < region_low_mb) region_low_mb = left_mb;
        }
        return AR_OK;
    }

    // stalled games get a doubled arena: live nodes are copied across unchanged (ids are arena-relative)
    // `may_wait`: other games are still running, so a stalled game that cannot be given memory now
    // simply stays stalled (k_advance retries it from the overflow pool on every launch)
    int handle_stalls(uint32_t n_stall, bool may_wait = false) {
        if (n_stall == 0) return AR_OK;
        hipLaunchKernelGGL(k_read_stall<NW>, dim3(grid(n_stall)), dim3(64), 0, stream, slots.p, stall_list.p, n_stall,
                           stall_info.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_stall.p, stall_info.p, sizeof(StallInfo) * n_stall, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        std::vector<GrowReq> reqs(n_stall);
        std::vector<std::pair<uint32_t, void*>> old_arenas;
        for (uint32_t i = 0; i < n_stall; ++i) {
            const StallInfo& si = h_stall.p[i];
            uint32_t ncap = si.cap * 2;
            while (ncap < si.need) ncap *= 2;
            unsigned char* na = nullptr;
            const auto spare = spare_arenas.find(ncap);
            if (spare != spare_arenas.end()) {
                na = (unsigned char*)spare->second;
                spare_arenas.erase(spare);
            }
            // keep a reserve: the runtime allocates kernel scratch and queues from the same memory
            size_t free_b = 0, total_b = 0;
            const bool room = na != nullptr || (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                                                free_b > arena_bytes(ncap) + ((size_t)3 << 30));
            if (!room || (na == nullptr && hipMalloc((void**)&na, arena_bytes(ncap) + 256) != hipSuccess)) {
                (void)hipGetLastError();
                if (may_wait) {
                    n_stall = i;  // the rest waits for blocks to come back
                    reqs.resize(i);
                    break;
                }
                return fail(AR_E_NOMEM, "out of device memory while growing a tree arena to " + std::to_string(ncap) +
                                            " nodes");
            }
            if (slot_grown[si.slot]) old_arenas.emplace_back(slot_grown_cap[si.slot], slot_grown[si.slot]);
            slot_grown[si.slot] = na;
            slot_grown_cap[si.slot] = ncap;
            GrowReq r;
            r.slot = si.slot;
            r.cap = ncap;
            r.stats_off = (long long)(na - arena.p);
            r.fwd_off = r.stats_off + (long long)ncap * (long long)sizeof(NodeStats);
            HIP_TRY(hipMemcpyAsync(na, arena.p + si.stats_off, (size_t)si.hi * sizeof(NodeStats), hipMemcpyDeviceToDevice,
                                   stream));
            reqs[i] = r;
        }
        if (n_stall == 0) return AR_OK;
        HIP_TRY(hipMemcpyAsync(grow.p, reqs.data(), sizeof(GrowReq) * n_stall, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_apply_grow<NW>, dim3(grid(n_stall)), dim3(64), 0, stream, slots.p, grow.p, n_stall, bases());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& kv : old_arenas) spare_arenas.emplace(kv.first, kv.second);  // the copies are done (sync above)
        grows += n_stall;
        return AR_OK;
    }

    // copies the finished games' headers + positions to pinned host memory and frees their slots
    int drain(uint32_t n_done) {
        if (n_done == 0) return AR_OK;
        hipLaunchKernelGGL(k_pack_done<NW>, dim3(n_done), dim3(128), 0, stream, slots.p, done_list.p, n_done, info.p,
                           staging.p, max_turns, bases());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_info.p, info.p, sizeof(DoneInfo<NW>) * n_done, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_staging.p, staging.p, sizeof(PosRec<NW>) * (size_t)n_done * max_turns,
                               hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return AR_OK;
    }
};

std::vector<int8_t> maze_array(const std::vector<uint8_t>& cost) {
    std::vector<int8_t> m(cost.size());
    for (size_t i = 0; i < cost.size(); ++i) m[i] = cost[i] ? (int8_t)cost[i] : (int8_t)-1;
    return m;
}

template <int NW>
void record_from_device(const DoneInfo<NW>& d, const PosRec<NW>* pos, const HostGame& g0,
                        const std::vector<uint8_t>& cost, GameRecordHost& r) {
    const int w = g0.width, hw = g0.width * g0.height;
    r.width = g0.width;
    r.height = g0.height;
    r.max_turns = g0.max_turns;
    r.game_index = d.game_index;
    r.n = d.n_pos;
    r.maze = maze_array(cost);
    r.initial_cheese = g0.cheese;
    r.cheese_available = g0.total_cheese;
    r.final_p1 = d.final_st.s1;
    r.final_p2 = d.final_st.s2;
    r.result = r.final_p1 > r.final_p2 ? 1 : r.final_p2 > r.final_p1 ? 2 : 0;
    r.sims = d.t_sims;
    r.nn = d.t_nn;
    r.term = d.t_term;
    r.coll = d.t_coll;
    const size_t n = r.n;
    r.p1_pos.resize(n * 2);
    r.p2_pos.resize(n * 2);
    r.p1_mud.resize(n);
    r.p2_mud.resize(n);
    r.cheese_mask.resize(n * hw);
    r.a1.resize(n);
    r.a2.resize(n);
    r.p1_score.resize(n);
    r.p2_score.resize(n);
    r.v1.resize(n);
    r.v2.resize(n);
    r.vc1.resize(n * 5);
    r.vc2.resize(n * 5);
    r.pr1.resize(n * 5);
    r.pr2.resize(n * 5);
    r.po1.resize(n * 5);
    r.po2.resize(n * 5);
    r.turn.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const PosRec<NW>& p = pos[i];
        r.p1_pos[i * 2] = p.st.p1 % w;
        r.p1_pos[i * 2 + 1] = p.st.p1 / w;
        r.p2_pos[i * 2] = p.st.p2 % w;
        r.p2_pos[i * 2 + 1] = p.st.p2 / w;
        r.p1_mud[i] = p.st.m1;
        r.p2_mud[i] = p.st.m2;
        r.turn[i] = p.st.turn;
        r.p1_score[i] = p.st.s1;
        r.p2_score[i] = p.st.s2;
        for (int c = 0; c < hw; ++c) r.cheese_mask[i * hw + c] = st_has_cheese(p.st, c) ? 1 : 0;
        r.v1[i] = p.res.value[0];
        r.v2[i] = p.res.value[1];
        memcpy(&r.vc1[i * 5], p.res.visit_counts[0], 20);
        memcpy(&r.vc2[i * 5], p.res.visit_counts[1], 20);
        memcpy(&r.pr1[i * 5], p.res.prior[0], 20);
        memcpy(&r.pr2[i * 5], p.res.prior[1], 20);
        memcpy(&r.po1[i * 5], p.res.policy[0], 20);
        memcpy(&r.po2[i * 5], p.res.policy[1], 20);
        r.a1[i] = p.a1;
        r.a2[i] = p.a2;
    }
    // selfplay.rs:415-471 compute_cheese_outcomes: diff consecutive masks against the next positions
    r.cheese_outcomes.assign(hw, 2);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* cur = &r.cheese_mask[i * hw];
        uint8_t n1, n2;
        std::vector<uint8_t> last_mask;
        const uint8_t* next;
        if (i + 1 < n) {
            next = &r.cheese_mask[(i + 1) * hw];
            n1 = pos[i + 1].st.p1;
            n2 = pos[i + 1].st.p2;
        } else {
            last_mask.resize(hw);
            for (int c = 0; c < hw; ++c) last_mask[c] = st_has_cheese(d.final_st, c) ? 1 : 0;
            next = last_mask.data();
            n1 = d.final_st.p1;
            n2 = d.final_st.p2;
        }
        for (int c = 0; c < hw; ++c)
            if (cur[c] == 1 && next[c] == 0) r.cheese_outcomes[c] = (n1 == c && n2 == c) ? 1 : n1 == c ? 0 : n2 == c ? 3 : 2;
    }
}

std::string uuid4() {
    static std::mutex m;
    static std::mt19937_64 gen{std::random_device{}()};
    std::lock_guard<std::mutex> lk(m);
    uint64_t a = gen(), b = gen();
    a = (a & 0xFFFFFFFFFFFF0FFFULL) | 0x0000000000004000ULL;
    b = (b & 0x3FFFFFFFFFFFFFFFULL) | 0x8000000000000000ULL;
    char buf[40];
    snprintf(buf, sizeof buf, "%08x-%04x-%04x-%04x-%012llx", (uint32_t)(a >> 32), (uint32_t)((a >> 16) & 0xFFFF),
             (uint32_t)(a & 0xFFFF), (uint32_t)(b >> 48), (unsigned long long)(b & 0xFFFFFFFFFFFFULL));
    return buf;
}

// writer thread: recording.rs:170-224 BundleWriter behind an unbounded queue (selfplay.rs:742-757)
struct BundleSink {
    std::string dir;
    uint32_t max_games = 32;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::shared_ptr<GameRecordHost>> q;
    bool closed = false;
    std::string error;
    std::thread th;
    std::vector<std::shared_ptr<GameRecordHost>> buf;
    std::vector<std::string> paths;

    void start() {
        th = std::thread([this]() {
            for (;;) {
                std::shared_ptr<GameRecordHost> g;
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [this] { return closed || !q.empty(); });
                    if (q.empty()) break;
                    g = q.front();
                    q.pop_front();
                }
                buf.push_back(g);
                if (buf.size() >= max_games) flush();
            }
            flush();
        });
    }
    void flush() {
        if (buf.empty() || !error.empty()) {
            buf.clear();
            return;
        }
        std::vector<ArGameRecordView> views;
        for (auto& g : buf) views.push_back(g->view());
        std::string path = dir + "/bundle_" + uuid4() + ".npz", err;
        if (!write_bundle(views.data(), (uint32_t)views.size(), path, err)) error = err;
        else paths.push_back(path);
        buf.clear();
    }
    void push(std::shared_ptr<GameRecordHost> g) {
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(g));
        }
        cv.notify_one();
    }
    void finish() {
        {
            std::lock_guard<std::mutex> lk(m);
            closed = true;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
};

int parse_device(const char* device, int device_index, int& out) {
    std::string d = device ? device : "auto";
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(AR_E_DEVICE, "no HIP device visible: libalpharat_hip has no CPU path");
    if (d == "auto" || d == "hip" || d == "mi355x" || d == "cuda" || d == "rocm") out = device_index;
    else if (d.rfind("hip:", 0) == 0) out = atoi(d.c_str() + 4);
    else return fail(AR_E_INVALID, "device '" + d + "' is not available in the HIP sampler (use auto | hip | hip:N)");
    if (out < 0 || out >= n) return fail(AR_E_INVALID, "device index " + std::to_string(out) + " out of range");
    return AR_OK;
}

// ------------------------------------------------------------------------------------------------
// self-play driver
// ------------------------------------------------------------------------------------------------
// Round limit of one gather launch: a few rounds per wanted descent covers the typical batch (a
// descent is one round per tree level) and cuts the long tail; measured in DESIGN.md section 7.
static uint32_t default_gather_rounds(const SearchCfg&) { return 0xFFFFFFFFu; }

// A self-play run as an object that outlives one call: the engine, its resident games and the supply of
// new games stay alive between ar_selfplay_step calls, so a caller can run the sampler in bounded slices
// (the benchmark's "step") or to the end (ar_selfplay_run = open + step until finished + close).
// The counterpart in the reference is the worker pool of run_self_play_to_disk (selfplay.rs:721-808), which
// lives for one call only; the slicing is an extension.
struct SessionBase {
    virtual ~SessionBase() {}
    virtual void info(ArSessionInfo* out) const = 0;
    virtual int step(uint32_t batch_steps, ArSelfPlayStats* window, int* finished_out) = 0;
    virtual int close(ArSelfPlayStats* total) = 0;
};

template <int NW>
struct SelfPlaySession : SessionBase {
    ArSelfPlayParams p;  // strings are copied below; the pointers in here are not used after open()
    SearchCfg cfg;
    std::vector<uint8_t> cost;
    int hw = 0;
    uint64_t gseed = 0, rseed = 0;
    bool random_pos = false, gen_maze = false, maze_sym = true, unbounded = false;
    float wall_d = 0.0f, mud_d = 0.0f;
    uint32_t S = 0;
    Engine<NW> eng;
    ArNet* net = nullptr;  // owned by the ArSelfPlaySession wrapper (freed after the engine)
    BundleSink writer;
    bool to_disk = false;
    ArProgress* progress = nullptr;
    ArGameSink sink = nullptr;
    void* sink_user = nullptr;
    ArSelfPlayStats st;  // finished games
    unsigned long long live[LIVE_N] = {0};  // resident games at the last scan, minus the ones drained since
    std::vector<HostGame> slot_game;
    uint64_t next_game = 0, finished = 0;
    std::chrono::steady_clock::time_point t0;
    std::string err;
    bool timing = false, failed = false;
    double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    void info(ArSessionInfo* out) const override {
        out->resident_games = S;
        out->groups = (uint32_t)(eng.groups.empty() ? 1 : eng.groups.size());
        out->gather_kind = !eng.use_queue() ? 3u : eng.gatherw ? 2u : eng.gather8 ? 1u : 0u;
        out->gather_pass_limit = eng.gatherw ? eng.gatherw_passes : 0xFFFFFFFFu;
        out->tree_region_bytes = eng.region_bytes;
        out->host_grown_arenas = eng.grows;
        out->idle_slots = (uint32_t)idle_slots.size();
        out->tree_pages_per_game = (float)pages_per_game;
    }
    ~SelfPlaySession() override {
        if (to_disk) writer.finish();
        if (eng.stream) hipStreamSynchronize(eng.stream);
        // the evaluator outlives every kernel that reads its weights: ~Engine synchronises its streams, so
        // the engine has to go first -- done explicitly in close(); here only as a last resort
    }

    bool make_game(uint32_t index, HostGame& g) {
        if (gen_maze) g.cost = generate_maze(p.width, p.height, wall_d, mud_d, maze_sym, gseed + index);
        g.width = p.width;
        g.height = p.height;
        g.max_turns = p.max_turns;
        g.turn = 0;
        g.m1 = g.m2 = 0;
        g.s1 = g.s2 = 0.0f;
        g.p1 = 0;
        g.p2 = (uint8_t)(hw - 1);
        if (random_pos) {
            HostRng rng((gseed + index) ^ 0x9E3779B97F4A7C15ULL);
            g.p1 = (uint8_t)rng.below((uint32_t)hw);
            g.p2 = (uint8_t)(hw - 1 - g.p1);
            if (!p.cheese_symmetric) g.p2 = (uint8_t)rng.below((uint32_t)hw);
        }
        return place_cheese(g, p.cheese_count, p.cheese_symmetric != 0, gseed + index, err);
    }

    bool supply_left() const { return unbounded || next_game < p.num_games; }
    bool all_finished() const { return !unbounded && finished >= p.num_games; }

    // How many games are resident is bounded by what their trees need, and that depends on the run (the network's priors
    // decide how much of a tree survives a move): a slot is only given a new game while its zone of the tree region would
    // stay under 88 % with every resident game at the size games of this run have after a few moves (measured by k_scan;
    // four fresh arenas per game until there is a measurement). Slots that have to wait are tried again at every visit.
    // Without this, 131072 games with a network that keeps big subtrees fill the region before the first generation has
    // finished, nothing can grow any more, and the run crawls (24 M instead of 370 M simulations/s, measured).
    std::vector<uint8_t> slot_busy;
    std::vector<uint32_t> zone_busy, idle_slots;
    double pages_per_game = 0.0;
    bool admit(uint32_t slot) const {
        const uint32_t z = slot / POOL_ZONE_SLOTS;
        const double zone_pages = (double)eng.pool.zone_words * POOL_WORD_PAGES;
        return ((double)zone_busy[z] + 1.0) * pages_per_game <= 0.88 * zone_pages || zone_busy[z] == 0;
    }
    void leave_slot(uint32_t slot) {
        if (slot < slot_busy.size() && slot_busy[slot]) {
            slot_busy[slot] = 0;
            zone_busy[slot / POOL_ZONE_SLOTS] -= 1;
        }
    }
    int refill(const std::vector<uint32_t>& free_slots) {
        std::vector<GameInit<NW>> inits;
        std::vector<uint8_t> mazes;
        std::vector<uint32_t> candidates;
        candidates.swap(idle_slots);
        candidates.insert(candidates.end(), free_slots.begin(), free_slots.end());
        for (uint32_t sl : candidates) {
            if (!supply_left()) break;
            if (!admit(sl)) {
                idle_slots.push_back(sl);
                continue;
            }
            slot_busy[sl] = 1;
            zone_busy[sl / POOL_ZONE_SLOTS] += 1;
            const uint32_t index = p.first_game_index + (uint32_t)next_game;  // wraps in an unbounded session
            HostGame g;
            if (!make_game(index, g)) return fail(AR_E_INVALID, err);
            GameInit<NW> gi;
            memset(&gi, 0, sizeof gi);
            fill_state<NW>(g, gi.board, gi.st, gen_maze ? sl * eng.maze_stride : 0u);
            if (gen_maze) mazes.insert(mazes.end(), g.cost.begin(), g.cost.end());
            gi.rng_seed = rseed + index;
            gi.game_index = index;
            gi.slot = sl;
            gi.single = 0;
            inits.push_back(gi);
            slot_game[sl] = std::move(g);
            ++next_game;
        }
        return eng.start_games(inits, gen_maze ? &mazes : nullptr);
    }

    int open(const ArSelfPlayParams& params, int device, ArNet* owned_net, ArProgress* prog, ArGameSink sk, void* sk_user) {
        p = params;
        net = owned_net;
        progress = prog;
        sink = sk;
        sink_user = sk_user;
        memset(&st, 0, sizeof st);
        st.min_turns = 0xFFFFFFFFu;
        cfg = to_cfg(p.search, p.simulations, p.batch_size);
        if (int rc = check_cfg(cfg)) return rc;
        cost = open_maze_cost(p.width, p.height);
        hw = p.width * p.height;
        unbounded = p.num_games == 0xFFFFFFFFu;

        gseed = p.game_seed_base;
        rseed = p.rng_seed_base;
        if (!p.has_seed) {  // reference behaviour: entropy (selfplay.rs:622, bindings.rs:530-532)
            std::random_device rd;
            gseed = ((uint64_t)rd() << 32) | rd();
            rseed = ((uint64_t)rd() << 32) | rd();
        }
        random_pos = p.positions && std::string(p.positions) == "random";
        // "classic" = the engine's default walls and mud (bindings.rs:507 with_classic_maze; taken here as density
        // 0.7 / 0.1, symmetric -- the defaults of the binding's own signature), "random" = the caller's parameters
        const std::string maze_type = p.maze_type ? p.maze_type : "open";
        gen_maze = maze_type != "open";
        wall_d = maze_type == "classic" ? 0.7f : p.wall_density;
        mud_d = maze_type == "classic" ? 0.1f : p.mud_density;
        maze_sym = maze_type == "classic" ? true : p.maze_symmetric != 0;
        to_disk = p.output_dir != nullptr;
        if (to_disk) {
            writer.dir = p.output_dir;
            writer.max_games = p.max_games_per_bundle ? p.max_games_per_bundle : 32;
        }
        p.maze_type = p.positions = p.output_dir = p.weights_path = p.device = nullptr;  // caller-owned strings

        // resident games: bounded by the request, the caller's hint and the arena footprint
        S = p.concurrent_games ? p.concurrent_games : 16384;
        if (S > p.num_games) S = p.num_games;
        size_t pool_bytes = 0;
        const uint32_t arena_nodes = getenv("AR_ARENA_NODES") ? (uint32_t)atoi(getenv("AR_ARENA_NODES")) : 0u;  // test knob
        t0 = std::chrono::steady_clock::now();
        if (S == 0) return AR_OK;
        {
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(device) != hipSuccess) return fail(AR_E_DEVICE, "hipSetDevice failed");
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            free_b += arena_cached_bytes(device);  // the block kept from the previous call is ours to reuse
            if (const char* e = getenv("AR_MEM_FRACTION"))  // several processes on one device (bench rehearsals): each
                if (atof(e) > 0.0 && atof(e) <= 1.0) free_b = (size_t)((double)free_b * atof(e));  // takes its share
            // SmartUniform runs: the fused step kernel, or the split pipeline of the network path with k_uniform_eval as
            // its evaluator (AR_UNIFORM=fused | queue; results are identical)
            eng.uniform_queue = net == nullptr && default_uniform_queue(S);
            if (const char* e = getenv("AR_UNIFORM")) eng.uniform_queue = net == nullptr && std::string(e) == "queue";
            // device memory per resident game: its scratch and queue share, and its tree. A tree lives in a block of the
            // smallest size class (4/3 apart) that holds it plus one full search; averaged over a game's life that is about
            // twice a fresh game's block at the tuned 7x7 configuration (measured: mean tree top 2150 nodes at 1897
            // simulations per move), which is what the resident count is sized by -- games beyond what the region can
            // serve would only wait for blocks.
            const size_t fresh = arena_bytes(arena_nodes ? arena_nodes : initial_arena_nodes(cfg));
            const size_t overhead = Engine<NW>::per_game_overhead(cfg, p.max_turns, net != nullptr || eng.uniform_queue);
            const size_t reserve = (size_t)4 << 30;  // the evaluator's buffers, arenas beyond the largest class (host-allocated)
            const size_t usable = free_b > reserve ? free_b - reserve : 0;
            const size_t per_game = 2 * fresh + overhead;
            if ((size_t)S * per_game > usable) S = (uint32_t)(usable / per_game);
            if (S == 0) return fail(AR_E_NOMEM, "not enough device memory for a single game arena");
            pool_bytes = (usable - (size_t)S * overhead) / 100 * 95;
        }

        // AR_TIMING=1 prints where the host wall time of this run went (stderr)
        timing = getenv("AR_TIMING") != nullptr;
        auto tp = std::chrono::steady_clock::now();
        auto since = [](std::chrono::steady_clock::time_point a) {
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        };
        eng.net = net;
        if (p.cache_size && net != nullptr) {
            // the reference sizes one table per worker thread (capacity x 1.9 slots, nn_cache.rs:49); one table
            // serves every game here, so it gets the threads' worth, rounded up to a power of two
            const uint64_t want = (uint64_t)((double)p.cache_size * (p.num_threads ? p.num_threads : 1) * 1.9) + 1;
            uint64_t e = 1u << 16;
            while (e < want && e < (1ULL << 26)) e <<= 1;
            eng.cache_entries = e;
        }
        if (getenv("AR_NO_POOL")) pool_bytes = 0;  // test knob: one smallest block per game, every growth goes through the host path
        if (const char* e = getenv("AR_TREE_GB"))  // test knob: a small tree region (games wait for blocks, trees move between classes)
            if (atof(e) > 0.0) pool_bytes = (size_t)(atof(e) * 1073741824.0);
        eng.per_slot_maze = gen_maze;
        eng.maze_stride = (uint32_t)hw * 4u;
        {
            // generated mazes: one pool entry per slot (filled when a game starts there); open: the one shared maze
            const std::vector<uint8_t> pool_init = gen_maze ? std::vector<uint8_t>((size_t)S * hw * 4, (uint8_t)0) : cost;
            if (int rc = eng.setup(device, S, cfg, p.max_turns, pool_init, arena_nodes, net != nullptr || eng.uniform_queue, pool_bytes)) return rc;
        }
        tm[0] = since(tp);
        // rounds per gather launch (dev_search.h gather_machine_limited); AR_GATHER_ROUNDS overrides, 0 = no limit
        eng.gather_rounds = default_gather_rounds(cfg);
        if (const char* e = getenv("AR_GATHER_ROUNDS")) eng.gather_rounds = atoi(e) > 0 ? (uint32_t)atoi(e) : 0xFFFFFFFFu;
        {
            // groups of games pipelined against each other (Engine::group_step); AR_GROUPS overrides
            // two groups from 8192 games up (one group's evaluator runs beside the other group's tree walks: +6..18 %,
            // DESIGN.md section 7) -- when every stream can have a hardware queue of its own
            uint32_t ng = (S >= 8192u && hw_queues() >= 8) ? 2u : 1u;
            if (const char* e = getenv("AR_GROUPS"))
                if (atoi(e) >= 1 && atoi(e) <= 64) ng = (uint32_t)atoi(e);
            if (eng.cache_entries) ng = 1;  // one probe/fill pair in flight at a time: a reader never overlaps an eviction
            if (getenv("AR_NO_ADVANCE_OVERLAP")) eng.overlap_advance = false;
            if (getenv("AR_STAGGER")) eng.stagger_gathers = true;
            if (const char* e = getenv("AR_ADV_LATE")) eng.adv_late = atoi(e) != 0;
            if (const char* e = getenv("AR_CUMASK")) eng.cu_split = std::string(e) == "lohi" ? 1 : std::string(e) == "evenodd" ? 2 : 0;
            if (int rc = eng.make_groups(ng)) return rc;
        }
        if (const char* e = getenv("AR_LANES_PER_WAVE"))
            if (atoi(e) >= 1 && atoi(e) <= 64) eng.lanes = eng.backup_lanes = (uint32_t)atoi(e);
        if (const char* e = getenv("AR_BACKUP_LANES"))
            if (atoi(e) >= 1 && atoi(e) <= 64) eng.backup_lanes = (uint32_t)atoi(e);
        if (eng.gather_rounds != 0xFFFFFFFFu) eng.gather8 = eng.gatherw = false;  // the round limit parks the walk: lane kernel only
        if (to_disk) writer.start();
        slot_game.resize(S);
        slot_busy.assign(S, 0);
        zone_busy.assign(eng.pool.zones, 0u);
        pages_per_game = 4.0 * (double)eng.pool.fresh_pages;
        t0 = std::chrono::steady_clock::now();
        tp = t0;
        std::vector<uint32_t> all(S);
        for (uint32_t i = 0; i < S; ++i) all[i] = i;
        if (int rc = refill(all)) return rc;
        tm[1] = since(tp);
        return AR_OK;
    }

    // `c` simulate_batch steps for every resident game
    int run_chunk(uint32_t c) {
        const uint32_t iters = 4;  // batches one k_step_uniform launch runs per lane before the tree reuse
        if (c / iters)
            if (int rc = eng.run_steps((int)(c / iters), (int)iters)) return rc;
        if (c % iters)
            if (int rc = eng.run_steps(1, (int)(c % iters))) return rc;
        return AR_OK;
    }

    // one finished game: stats, progress, sink, writer (SelfPlayStats::add_game, selfplay.rs:190-210)
    void account(const DoneInfo<NW>& di, const PosRec<NW>* pos) {
        const float f1 = di.final_st.s1, f2 = di.final_st.s2;
        const HostGame& hg = slot_game[di.slot];
        st.total_games += 1;
        st.total_positions += di.n_pos;
        st.total_simulations += di.t_sims;
        st.total_nn_evals += di.t_nn;
        st.total_terminals += di.t_term;
        st.total_collisions += di.t_coll;
        st.total_cheese_collected += f1 + f2;
        st.total_cheese_available += hg.total_cheese;
        if (di.n_pos < st.min_turns) st.min_turns = di.n_pos;
        if (di.n_pos > st.max_turns) st.max_turns = di.n_pos;
        if (f1 > f2) st.p1_wins += 1;
        else if (f2 > f1) st.p2_wins += 1;
        else st.draws += 1;
        st.gather_node_visits += di.nv_gather;
        st.backup_node_visits += di.nv_backup;
        st.new_nodes += di.new_nodes;
        // the game leaves the resident set: its part of the last scan's sums moves to the finished totals
        const unsigned long long part[LIVE_N] = {di.n_pos, di.t_sims, di.t_nn, di.t_term, di.t_coll, di.nv_gather,
                                                di.nv_backup, di.new_nodes, 1};
        for (int k = 0; k < LIVE_N; ++k) live[k] = live[k] >= part[k] ? live[k] - part[k] : 0;
        if (progress) {  // selfplay.rs:637-645
            __atomic_fetch_add(&progress->positions_completed, (uint64_t)di.n_pos, __ATOMIC_RELAXED);
            __atomic_fetch_add(&progress->simulations_completed, (uint64_t)di.t_sims, __ATOMIC_RELAXED);
            __atomic_fetch_add(&progress->nn_evals_completed, (uint64_t)di.t_nn, __ATOMIC_RELAXED);
            __atomic_fetch_add(&progress->games_completed, 1u, __ATOMIC_RELAXED);
        }
        if (sink || to_disk) {  // the record itself is only built for someone who wants it
            auto rec = std::make_shared<GameRecordHost>();
            record_from_device<NW>(di, pos, hg, hg.cost.empty() ? cost : hg.cost, *rec);
            if (sink) {
                ArGameRecordView v = rec->view();
                sink(sink_user, &v);
            }
            if (to_disk && rec->n > 0) writer.push(rec);
        }
    }

    // cumulative figures: finished games + what the resident games have done so far (per finished move)
    void totals(ArSelfPlayStats& o) const {
        o = st;
        o.total_positions += live[0];
        o.total_simulations += live[1];
        o.total_nn_evals += live[2];
        o.total_terminals += live[3];
        o.total_collisions += live[4];
        o.gather_node_visits += live[5];
        o.backup_node_visits += live[6];
        o.new_nodes += live[7];
        o.device_secs = eng.device_ms / 1000.0;
        o.steps = eng.steps;
        o.gather_secs = eng.gather_ms / 1000.0;
        o.gather_launches = eng.gather_launches;
    }
    int read_cache_counters(ArSelfPlayStats& o) {
        if (!eng.cache_entries) return AR_OK;
        unsigned long long hm[2] = {0, 0};
        HIP_TRY(hipMemcpy(hm, eng.cache_counters.p, sizeof hm, hipMemcpyDeviceToHost));
        o.cache_hits = hm[0];
        o.cache_misses = hm[1];
        return AR_OK;
    }

    int step(uint32_t batch_steps, ArSelfPlayStats* window, int* finished_out) override {
        if (failed) return fail(AR_E_INVALID, "the session has failed earlier; close it");
        if (unbounded && batch_steps == 0xFFFFFFFFu)
            return fail(AR_E_INVALID, "a session with an endless supply of games cannot be run to its end: pass a number of batch steps");
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(now() - a).count(); };
        const auto w0 = now();
        ArSelfPlayStats before;
        totals(before);
        if (int rc = read_cache_counters(before)) return rc;
        uint32_t win_min = 0xFFFFFFFFu, win_max = 0;
        const ArSelfPlayStats st_before = st;
        int rc = AR_OK;
        // steps between host visits: a game needs ~n_sims/batch steps per move, so a few dozen steps of
        // latency on refills and arena growth costs little
        const uint32_t visit_every = 32;
        uint64_t left = batch_steps;
        while (S > 0 && !all_finished() && left > 0) {
            const uint32_t chunk = left < visit_every ? (uint32_t)left : visit_every;
            if (batch_steps != 0xFFFFFFFFu) left -= chunk;
            auto tp = now();
            if ((rc = run_chunk(chunk)) != AR_OK) break;
            tm[2] += since(tp);
            tp = now();
            uint32_t c[4];
            if ((rc = eng.scan(c)) != AR_OK) break;
            for (int k = 0; k < LIVE_N; ++k) live[k] = eng.h_live.p[k];
            if (eng.h_live.p[13] >= 256)  // tree pages of a game that has played a few moves, in this run
                pages_per_game = std::max((double)eng.pool.fresh_pages, 1.1 * (double)eng.h_live.p[12] / (double)eng.h_live.p[13]);
            tm[3] += since(tp);
            tp = now();
            if (c[3] != 0) {
                rc = fail(AR_E_DEVICE, "internal capacity guard tripped in a tree kernel (slot.error != 0)");
                break;
            }
            if (c[1] && (rc = eng.handle_stalls(c[1], c[2] > 0)) != AR_OK) break;
            tm[4] += since(tp);
            tp = now();
            if (c[0]) {
                const uint32_t n_done = c[0];
                if ((rc = eng.drain(n_done)) != AR_OK) break;
                tm[5] += since(tp);
                tp = now();
                std::vector<uint32_t> free_slots;
                for (uint32_t d = 0; d < n_done; ++d) {
                    const DoneInfo<NW>& di = eng.h_info.p[d];
                    account(di, eng.h_staging.p + (size_t)d * p.max_turns);
                    if (di.n_pos < win_min) win_min = di.n_pos;
                    if (di.n_pos > win_max) win_max = di.n_pos;
                    free_slots.push_back(di.slot);
                    leave_slot(di.slot);
                    ++finished;
                }
                tm[6] += since(tp);
                tp = now();
                if ((rc = refill(free_slots)) != AR_OK) break;
                tm[7] += since(tp);
            } else if (!idle_slots.empty() && supply_left()) {
                if ((rc = refill(std::vector<uint32_t>())) != AR_OK) break;  // (slots that waited for room in their zone)
            }
            if (c[0] == 0 && c[1] == 0 && c[2] == 0 && !all_finished() && !supply_left()) {
                rc = fail(AR_E_DEVICE, "self-play stalled: no active games left but not all games finished");
                break;
            }
        }
        if (rc != AR_OK) failed = true;
        if (window) {
            ArSelfPlayStats a;
            totals(a);
            if (rc == AR_OK) rc = read_cache_counters(a);
            ArSelfPlayStats& w = *window;
            memset(&w, 0, sizeof w);
            w.total_games = a.total_games - before.total_games;
            w.total_positions = a.total_positions - before.total_positions;
            w.total_simulations = a.total_simulations - before.total_simulations;
            w.total_nn_evals = a.total_nn_evals - before.total_nn_evals;
            w.total_terminals = a.total_terminals - before.total_terminals;
            w.total_collisions = a.total_collisions - before.total_collisions;
            w.gather_node_visits = a.gather_node_visits - before.gather_node_visits;
            w.backup_node_visits = a.backup_node_visits - before.backup_node_visits;
            w.new_nodes = a.new_nodes - before.new_nodes;
            w.p1_wins = st.p1_wins - st_before.p1_wins;
            w.p2_wins = st.p2_wins - st_before.p2_wins;
            w.draws = st.draws - st_before.draws;
            w.total_cheese_collected = st.total_cheese_collected - st_before.total_cheese_collected;
            w.total_cheese_available = st.total_cheese_available - st_before.total_cheese_available;
            w.min_turns = w.total_games ? win_min : 0;
            w.max_turns = win_max;
            w.cache_hits = a.cache_hits - before.cache_hits;
            w.cache_misses = a.cache_misses - before.cache_misses;
            w.device_secs = a.device_secs - before.device_secs;
            w.steps = a.steps - before.steps;
            w.gather_secs = a.gather_secs - before.gather_secs;
            w.gather_launches = a.gather_launches - before.gather_launches;
            w.elapsed_secs = since(w0);
        }
        if (finished_out) *finished_out = (S == 0 || all_finished()) ? 1 : 0;
        return rc;
    }

    // Ends the run: bundles flushed, totals of the FINISHED games returned (an unbounded session that is
    // closed with games in flight simply drops them, like a sampler that is interrupted).
    int close(ArSelfPlayStats* total) override {
        int rc = AR_OK;
        if (to_disk) {
            writer.finish();
            to_disk = false;
            if (!writer.error.empty()) rc = fail(AR_E_IO, writer.error);
        }
        ArSelfPlayStats o = st;
        o.elapsed_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (o.total_games == 0) o.min_turns = 0;
        o.device_secs = eng.device_ms / 1000.0;
        o.steps = eng.steps;
        o.gather_secs = eng.gather_ms / 1000.0;
        o.gather_launches = eng.gather_launches;
        if (S > 0 && rc == AR_OK) rc = read_cache_counters(o);
        if (timing)
            fprintf(stderr,
                    "[ar timing] games=%llu resident=%u wall=%.3fs device=%.3fs | setup %.3f first-fill %.3f launch %.3f "
                    "scan-wait %.3f grow %.3f (%llu) drain %.3f records %.3f refill %.3f | tree region %.1f GB, least left "
                    "%.1f GB\n",
                    (unsigned long long)finished, S, o.elapsed_secs, o.device_secs, tm[0], tm[1], tm[2], tm[3], tm[4],
                    (unsigned long long)eng.grows, tm[5], tm[6], tm[7], (double)eng.region_bytes / 1073741824.0,
                    eng.region_low_mb == 0xFFFFFFFFu ? -1.0 : (double)eng.region_low_mb / 1024.0);
        if (total) *total = o;
        return rc;
    }
};

// ------------------------------------------------------------------------------------------------
// single searches
// ------------------------------------------------------------------------------------------------
bool host_game_from_spec(const ArGameSpec& g, HostGame& h, std::vector<uint8_t>& cost, std::string& err) {
    const int hw = g.width * g.height;
    if (g.width == 0 || g.height == 0 || hw > 256) {
        err = "board must have 1..256 cells";
        return false;
    }
    if (g.p1_x >= g.width || g.p2_x >= g.width || g.p1_y >= g.height || g.p2_y >= g.height) {
        err = "player position outside the board";
        return false;
    }
    if (!g.cheese) {
        err = "cheese mask is required";
        return false;
    }
    h.width = g.width;
    h.height = g.height;
    h.max_turns = g.max_turns;
    h.turn = g.turn;
    h.p1 = (uint8_t)(g.p1_y * g.width + g.p1_x);
    h.p2 = (uint8_t)(g.p2_y * g.width + g.p2_x);
    h.m1 = g.p1_mud;
    h.m2 = g.p2_mud;
    h.s1 = g.p1_score;
    h.s2 = g.p2_score;
    h.cheese.assign(g.cheese, g.cheese + hw);
    uint32_t rem = 0;
    for (int i = 0; i < hw; ++i) rem += h.cheese[i] ? 1 : 0;
    h.total_cheese = (uint16_t)(g.p1_score + g.p2_score + (float)rem);
    if (g.cost) cost.assign(g.cost, g.cost + (size_t)hw * 4);
    else cost = open_maze_cost(g.width, g.height);
    return true;
}

void fill_result(const MoveResult& m, ArSearchResult* o) {
    memcpy(o->policy_p1, m.policy[0], 20);
    memcpy(o->policy_p2, m.policy[1], 20);
    o->value_p1 = m.value[0];
    o->value_p2 = m.value[1];
    memcpy(o->visit_counts_p1, m.visit_counts[0], 20);
    memcpy(o->visit_counts_p2, m.visit_counts[1], 20);
    memcpy(o->prior_p1, m.prior[0], 20);
    memcpy(o->prior_p2, m.prior[1], 20);
    o->total_visits = m.total_visits;
    o->nn_evals = m.nn_evals;
    o->terminals = m.terminals;
    o->collisions = m.collisions;
}

template <int NW>
int search_impl(const ArGameSpec* games, uint32_t n, const SearchCfg& cfg, const uint64_t* seeds,
                ArPredictFn predict_fn, void* user, ArNet* net, int device, ArSearchResult* out) {
    std::string err;
    std::vector<HostGame> hg(n);
    std::vector<uint8_t> mazes;
    std::vector<uint32_t> maze_off(n);
    uint16_t max_turns = 1;
    for (uint32_t i = 0; i < n; ++i) {
        std::vector<uint8_t> c;
        if (!host_game_from_spec(games[i], hg[i], c, err)) return fail(AR_E_INVALID, err);
        maze_off[i] = (uint32_t)mazes.size();
        mazes.insert(mazes.end(), c.begin(), c.end());
        if (hg[i].max_turns > max_turns) max_turns = hg[i].max_turns;
    }
    Engine<NW> eng;
    eng.net = predict_fn ? nullptr : net;
    eng.uniform_queue = !predict_fn && net == nullptr && default_uniform_queue(n);
    if (const char* e = getenv("AR_UNIFORM")) eng.uniform_queue = !predict_fn && net == nullptr && std::string(e) == "queue";
    if (eng.net != nullptr)
        for (uint32_t i = 0; i < n; ++i)
            if (games[i].width != net->dev.width || games[i].height != net->dev.height)
                return fail(AR_E_INVALID, "game size does not match the network's board size");
    if (int rc = eng.setup(device, n, cfg, max_turns, mazes, 0, eng.use_queue())) return rc;
    std::vector<GameInit<NW>> inits(n);
    std::random_device rd;
    for (uint32_t i = 0; i < n; ++i) {
        memset(&inits[i], 0, sizeof inits[i]);
        fill_state<NW>(hg[i], inits[i].board, inits[i].st, maze_off[i]);
        inits[i].rng_seed = seeds ? seeds[i] : (((uint64_t)rd() << 32) | rd());
        inits[i].game_index = i;
        inits[i].slot = i;
        inits[i].single = 1;
    }
    if (int rc = eng.start_games(inits)) return rc;

    if (predict_fn) {
        // host evaluator per leaf batch (PyCallbackBackend, bindings.rs:119-161); n == 1
        DevBuf<State<NW>> d_leaves;
        DevBuf<EvalOut> d_ev;
        DevBuf<uint32_t> d_n;
        HIP_TRY(d_leaves.alloc(cfg.batch_size));
        HIP_TRY(d_ev.alloc(cfg.batch_size));
        HIP_TRY(d_n.alloc(1));
        std::vector<State<NW>> leaves(cfg.batch_size);
        std::vector<ArLeaf> al(cfg.batch_size);
        std::vector<EvalOut> ev(cfg.batch_size);
        std::vector<float> pp1(cfg.batch_size * 5), pp2(cfg.batch_size * 5), pv1(cfg.batch_size), pv2(cfg.batch_size);
        const int w = hg[0].width;
        for (;;) {
            uint32_t c[4];
            if (int rc = eng.scan(c)) return rc;
            if (c[3]) return fail(AR_E_DEVICE, "internal capacity guard tripped in a tree kernel");
            if (c[1]) {
                if (int rc = eng.handle_stalls(c[1])) return rc;
                continue;
            }
            if (c[2] == 0) break;
            eng.launch_gather(false);
            hipLaunchKernelGGL(k_export_leaves<NW>, dim3(1), dim3(64), 0, eng.stream, eng.slots.p, 0u, eng.bases(),
                               d_leaves.p, d_n.p);
            uint32_t nl = 0;
            HIP_TRY(hipMemcpyAsync(&nl, d_n.p, 4, hipMemcpyDeviceToHost, eng.stream));
            HIP_TRY(hipMemcpyAsync(leaves.data(), d_leaves.p, sizeof(State<NW>) * cfg.batch_size, hipMemcpyDeviceToHost,
                                   eng.stream));
            HIP_TRY(hipStreamSynchronize(eng.stream));
            if (nl > 0) {
                for (uint32_t j = 0; j < nl; ++j) {
                    const State<NW>& s = leaves[j];
                    ArLeaf& a = al[j];
                    memset(&a, 0, sizeof a);
                    a.p1_x = s.p1 % w;
                    a.p1_y = s.p1 / w;
                    a.p2_x = s.p2 % w;
                    a.p2_y = s.p2 / w;
                    a.p1_mud = s.m1;
                    a.p2_mud = s.m2;
                    a.turn = s.turn;
                    a.p1_score = s.s1;
                    a.p2_score = s.s2;
                    for (int k = 0; k < NW; ++k) a.cheese_bits[k] = s.cheese[k];
                }
                if (predict_fn(user, al.data(), nl, pp1.data(), pp2.data(), pv1.data(), pv2.data()) != 0) {
                    eng.launch_cancel();
                    hipStreamSynchronize(eng.stream);
                    return fail(AR_E_BACKEND, "predict_fn raised an exception");
                }
                for (uint32_t j = 0; j < nl; ++j) {
                    memcpy(ev[j].p1, &pp1[j * 5], 20);
                    memcpy(ev[j].p2, &pp2[j * 5], 20);
                    ev[j].v1 = pv1[j];
                    ev[j].v2 = pv2[j];
                }
                HIP_TRY(hipMemcpyAsync(d_ev.p, ev.data(), sizeof(EvalOut) * nl, hipMemcpyHostToDevice, eng.stream));
                hipLaunchKernelGGL(k_import_evals<NW>, dim3(1), dim3(64), 0, eng.stream, eng.slots.p, 0u, eng.bases(),
                                   d_ev.p, nl);
            }
            eng.launch_backup(false);
            HIP_TRY(hipGetLastError());
        }
    } else {
        for (;;) {
            if (int rc = eng.run_steps(4, 8)) return rc;
            uint32_t c[4];
            if (int rc = eng.scan(c)) return rc;
            if (c[3]) return fail(AR_E_DEVICE, "internal capacity guard tripped in a tree kernel");
            if (c[1]) {
                if (int rc = eng.handle_stalls(c[1])) return rc;
            }
            if (c[2] == 0 && c[1] == 0) break;
        }
    }
    uint32_t c[4];
    if (int rc = eng.scan(c)) return rc;
    if (c[0] != n) return fail(AR_E_DEVICE, "search ended with unfinished slots");
    if (int rc = eng.drain(n)) return rc;
    for (uint32_t d = 0; d < n; ++d) fill_result(eng.h_info.p[d].last, &out[eng.h_info.p[d].game_index]);
    return AR_OK;
}

}  // namespace

// the evaluator must outlive the engine's streams: the session is destroyed first, then the net
struct ArSelfPlaySession {
    SessionBase* impl = nullptr;
    ArNet* net = nullptr;
    ~ArSelfPlaySession() {
        delete impl;
        if (net) ar_net_free(net);
    }
};

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* ar_version(void) { return "alpharat_hip 0.1.0 (gfx950)"; }

size_t ar_last_error(char* buf, size_t cap) {
    if (buf && cap) {
        size_t n = g_error.size() < cap - 1 ? g_error.size() : cap - 1;
        memcpy(buf, g_error.data(), n);
        buf[n] = 0;
    }
    return g_error.size();
}

int ar_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return fail(AR_E_DEVICE, "hipGetDeviceCount failed (no HIP device?)");
    return n;
}

// ---- game generation, host only (no device needed): what ar_selfplay_run draws for game `seed` ----------
int ar_generate_maze(uint8_t width, uint8_t height, float wall_density, float mud_density, int symmetric, uint64_t seed,
                     uint8_t* cost_out) {
    if (!cost_out || width == 0 || height == 0 || (int)width * height > 256) return fail(AR_E_INVALID, "bad board");
    const std::vector<uint8_t> c = generate_maze(width, height, wall_density, mud_density, symmetric != 0, seed);
    memcpy(cost_out, c.data(), c.size());
    return AR_OK;
}
int ar_generate_cheese(uint8_t width, uint8_t height, uint8_t p1_cell, uint8_t p2_cell, uint16_t count, int symmetric,
                       uint64_t seed, uint8_t* cheese_out) {
    if (!cheese_out || width == 0 || height == 0 || (int)width * height > 256) return fail(AR_E_INVALID, "bad board");
    HostGame g;
    g.width = width;
    g.height = height;
    g.p1 = p1_cell;
    g.p2 = p2_cell;
    std::string err;
    if (!place_cheese(g, count, symmetric != 0, seed, err)) return fail(AR_E_INVALID, err);
    memcpy(cheese_out, g.cheese.data(), g.cheese.size());
    return AR_OK;
}

#if defined(AR_STATS)
// sizes of the resident trees: [0..64) games by live-tree top (hi) / 512, [64..128) by arena capacity / 512,
// [128] sum of hi, [129] sum of cap, [130] games
int ar_debug_tree_hist(unsigned long long* out131) {
    HIP_TRY(hipDeviceSynchronize());
    if (!g_dbg_slots) return fail(AR_E_INVALID, "no session");
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 131 * 8));
    HIP_TRY(hipMemset(d, 0, 131 * 8));
    if (g_dbg_nw == 1) hipLaunchKernelGGL(k_dbg_tree_hist<1>, dim3((g_dbg_S + 255) / 256), dim3(256), 0, 0, (const Slot<1>*)g_dbg_slots, g_dbg_S, d);
    else hipLaunchKernelGGL(k_dbg_tree_hist<4>, dim3((g_dbg_S + 255) / 256), dim3(256), 0, 0, (const Slot<4>*)g_dbg_slots, g_dbg_S, d);
    HIP_TRY(hipMemcpy(out131, d, 131 * 8, hipMemcpyDeviceToHost));
    hipFree(d);
    return AR_OK;
}
int ar_debug_round_stats(unsigned long long* out32) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(ar::g_round_stats), sizeof(unsigned long long) * 32));
    unsigned long long z[32] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ar::g_round_stats), z, sizeof z));
    return AR_OK;
}
int ar_debug_gather_hist(unsigned long long* out136) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out136, HIP_SYMBOL(ar::g_gather_hist), sizeof(unsigned long long) * 136));
    unsigned long long z[136] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ar::g_gather_hist), z, sizeof z));
    return AR_OK;
}
int ar_debug_gather_clk(unsigned long long* out128) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out128, HIP_SYMBOL(ar::g_gather_clk), sizeof(unsigned long long) * 128));
    unsigned long long z[128] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ar::g_gather_clk), z, sizeof z));
    return AR_OK;
}
#endif

int ar_release_device_memory(int device) {
    int dev = 0;
    if (int rc = parse_device("hip", device, dev)) return rc;
    HIP_TRY(hipSetDevice(dev));
    ArenaBlock b;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        if (dev < MAX_DEVICES) std::swap(b, g_arena_cache[dev]);
    }
    if (b.p) HIP_TRY(hipFree(b.p));
    return AR_OK;
}

int ar_device_sync(int device) {
    int dev = 0;
    if (int rc = parse_device("hip", device, dev)) return rc;
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipDeviceSynchronize());
    return AR_OK;
}

int ar_search_many(const ArGameSpec* games, uint32_t n, const ArSearchConfig* cfg, uint32_t simulations,
                   uint32_t batch_size, const uint64_t* seeds, ArNet* net, int device, ArSearchResult* out) {
    if (!games || !cfg || !out || n == 0) return fail(AR_E_INVALID, "null argument");
    int dev = 0;
    if (int rc = parse_device("hip", device, dev)) return rc;
    SearchCfg c = to_cfg(*cfg, simulations, batch_size);
    if (int rc = check_cfg(c)) return rc;
    bool small = true;
    for (uint32_t i = 0; i < n; ++i) small = small && (int)games[i].width * games[i].height <= 64;
    return small ? search_impl<1>(games, n, c, seeds, nullptr, nullptr, net, dev, out)
                 : search_impl<4>(games, n, c, seeds, nullptr, nullptr, net, dev, out);
}

int ar_search(const ArGameSpec* game, const ArSearchConfig* cfg, uint32_t simulations, uint32_t batch_size,
              const uint64_t* seed, ArPredictFn predict_fn, void* user, ArNet* net, int device, ArSearchResult* out) {
    if (!game || !cfg || !out) return fail(AR_E_INVALID, "null argument");
    int dev = 0;
    if (int rc = parse_device("hip", device, dev)) return rc;
    SearchCfg c = to_cfg(*cfg, simulations, batch_size);
    if (int rc = check_cfg(c)) return rc;
    const bool small = (int)game->width * game->height <= 64;
    return small ? search_impl<1>(game, 1, c, seed, predict_fn, user, net, dev, out)
                 : search_impl<4>(game, 1, c, seed, predict_fn, user, net, dev, out);
}

int ar_selfplay_open(const ArSelfPlayParams* p, ArProgress* progress, ArGameSink sink, void* sink_user,
                     ArSelfPlaySession** out) {
    if (!p || !out) return fail(AR_E_INVALID, "null argument");
    *out = nullptr;
    const std::string mt = p->maze_type ? p->maze_type : "open";
    if (mt != "open") {
        if (mt != "classic" && mt != "random") return fail(AR_E_INVALID, "unknown maze_type: " + mt);
    }
    const std::string pos = p->positions ? p->positions : "corners";
    if (pos != "corners" && pos != "random") return fail(AR_E_INVALID, "unknown positions: " + pos);
    if (p->width == 0 || p->height == 0 || (int)p->width * p->height > 256)
        return fail(AR_E_INVALID, "board must have 1..256 cells");
    int dev = 0;
    if (int rc = parse_device(p->device, p->device_index, dev)) return rc;
    std::unique_ptr<ArSelfPlaySession> s(new ArSelfPlaySession());
    if (p->weights_path) {
        if (int rc = ar_net_load(p->weights_path, dev, &s->net)) return rc;
        // the reference fails on the ONNX input shape when the model was trained for another board
        if (s->net->dev.width != p->width || s->net->dev.height != p->height)
            return fail(AR_E_INVALID, "the network was built for a " + std::to_string(s->net->dev.width) + "x" +
                                          std::to_string(s->net->dev.height) + " board, the games are " +
                                          std::to_string(p->width) + "x" + std::to_string(p->height));
    }
    int rc;
    if ((int)p->width * p->height <= 64) {
        auto* impl = new SelfPlaySession<1>();
        s->impl = impl;
        rc = impl->open(*p, dev, s->net, progress, sink, sink_user);
    } else {
        auto* impl = new SelfPlaySession<4>();
        s->impl = impl;
        rc = impl->open(*p, dev, s->net, progress, sink, sink_user);
    }
    if (rc != AR_OK) return rc;
    *out = s.release();
    return AR_OK;
}

int ar_selfplay_step(ArSelfPlaySession* s, uint32_t batch_steps, ArSelfPlayStats* window, int* finished) {
    if (!s || !s->impl) return fail(AR_E_INVALID, "null session");
    return s->impl->step(batch_steps, window, finished);
}

int ar_selfplay_info(const ArSelfPlaySession* s, ArSessionInfo* out) {
    if (!s || !s->impl || !out) return fail(AR_E_INVALID, "null argument");
    s->impl->info(out);
    return AR_OK;
}

int ar_selfplay_close(ArSelfPlaySession* s, ArSelfPlayStats* total) {
    if (!s) return AR_OK;
    const int rc = s->impl ? s->impl->close(total) : AR_OK;
    delete s;
    return rc;
}

int ar_selfplay_run(const ArSelfPlayParams* p, ArProgress* progress, ArGameSink sink, void* sink_user,
                    ArSelfPlayStats* out) {
    if (!p || !out) return fail(AR_E_INVALID, "null argument");
    if (p->num_games == 0xFFFFFFFFu) return fail(AR_E_INVALID, "num_games = UINT32_MAX (unbounded) needs a session");
    memset(out, 0, sizeof *out);
    ArSelfPlaySession* s = nullptr;
    if (int rc = ar_selfplay_open(p, progress, sink, sink_user, &s)) return rc;
    int finished = 0;
    const int rc = ar_selfplay_step(s, 0xFFFFFFFFu, nullptr, &finished);
    const std::string msg = g_error;
    const int rc2 = ar_selfplay_close(s, out);
    if (rc != AR_OK) return fail(rc, msg);
    return rc2;
}

int ar_net_load(const char* blob_path, int device, ArNet** out) {
    if (!blob_path || !out) return fail(AR_E_INVALID, "null argument");
    *out = nullptr;
    int dev = 0;
    if (int rc = parse_device("hip", device, dev)) return rc;
    HIP_TRY(hipSetDevice(dev));
    arnet::Blob blob;
    std::string err;
    if (!blob.load(blob_path, err)) return fail(AR_E_IO, err);
    std::unique_ptr<ArNet> net(new ArNet());
    net->device = dev;
    if (int rc = net_build(blob, net.get())) return rc;
    *out = net.release();
    return AR_OK;
}

void ar_net_free(ArNet* net) { delete net; }

}  // extern "C"

namespace {
template <int NW>
int specs_to_device(const ArGameSpec* games, uint32_t n, std::vector<LeafReq<NW>>& reqs, std::vector<Board>& boards,
                    std::vector<uint8_t>& mazes) {
    std::string err;
    reqs.resize(n);
    boards.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        HostGame hg;
        std::vector<uint8_t> c;
        if (!host_game_from_spec(games[i], hg, c, err)) return fail(AR_E_INVALID, err);
        memset(&reqs[i], 0, sizeof reqs[i]);
        fill_state<NW>(hg, boards[i], reqs[i].st, (uint32_t)mazes.size());
        reqs[i].slot = i;
        mazes.insert(mazes.end(), c.begin(), c.end());
    }
    return AR_OK;
}

// grow-only device buffer kept inside the ArNet between ar_net_evaluate calls (callers such as a predict_fn
// adaptor evaluate a handful of positions thousands of times: five allocations per call cost more than the network)
static int scratch_reserve(ArNet* net, int which, size_t bytes, void** out) {
    ArNet::Scratch& sc = net->scratch[which];
    if (sc.bytes < bytes) {
        if (sc.p) hipFree(sc.p);
        sc.p = nullptr;
        sc.bytes = 0;
        const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
        if (hipMalloc(&sc.p, want) != hipSuccess) return fail(AR_E_NOMEM, "device allocation failed (evaluator scratch)");
        sc.bytes = want;
    }
    *out = sc.p;
    return AR_OK;
}

template <int NW>
int net_evaluate_impl(ArNet* net, const ArGameSpec* games, uint32_t n, float* pp1, float* pp2, float* pv1, float* pv2,
                      float* lg1, float* lg2) {
    std::vector<LeafReq<NW>> reqs;
    std::vector<Board> boards;
    std::vector<uint8_t> mazes;
    for (uint32_t i = 0; i < n; ++i)
        if (games[i].width != net->dev.width || games[i].height != net->dev.height)
            return fail(AR_E_INVALID, "game size does not match the network's board size");
    if (int rc = specs_to_device<NW>(games, n, reqs, boards, mazes)) return rc;
    // positions on the same maze share one pool entry (the usual case: a batch of leaves of one game)
    const size_t per = (size_t)net->dev.hw * 4;
    bool one_maze = true;
    for (uint32_t i = 1; i < n && one_maze; ++i) one_maze = memcmp(&mazes[i * per], &mazes[0], per) == 0;
    if (one_maze) {
        mazes.resize(per);
        for (uint32_t i = 0; i < n; ++i) boards[i].maze_off = 0;
    }
    const int n_mazes = one_maze ? 1 : (int)n;
    HIP_TRY(hipSetDevice(net->device));
    LeafReq<NW>* d_req = nullptr;
    Board* d_boards = nullptr;
    uint8_t* d_maze = nullptr;
    EvalOut* d_out = nullptr;
    float* d_logits = nullptr;
    if (int rc = scratch_reserve(net, 0, sizeof(LeafReq<NW>) * n, (void**)&d_req)) return rc;
    if (int rc = scratch_reserve(net, 1, sizeof(Board) * n, (void**)&d_boards)) return rc;
    if (int rc = scratch_reserve(net, 2, mazes.size(), (void**)&d_maze)) return rc;
    if (int rc = scratch_reserve(net, 3, sizeof(EvalOut) * n, (void**)&d_out)) return rc;
    if (int rc = scratch_reserve(net, 4, (size_t)n * 40, (void**)&d_logits)) return rc;
    HIP_TRY(hipMemcpy(d_req, reqs.data(), sizeof(LeafReq<NW>) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_boards, boards.data(), sizeof(Board) * n, hipMemcpyHostToDevice));
    // the maze constants are only recomputed when the mazes' bytes (not just their address) differ from the last call's
    if (net->scratch_mazes != mazes || net->bound_pool != d_maze || net->bound_mazes != n_mazes) {
        HIP_TRY(hipMemcpy(d_maze, mazes.data(), mazes.size(), hipMemcpyHostToDevice));
        if (int rc = net_bind_mazes(net, d_maze, n_mazes, nullptr)) return rc;
        net->scratch_mazes = mazes;
    }
    int rc = net_launch<NW>(net, d_req, nullptr, n, (const char*)d_boards, sizeof(Board), d_out, d_logits, nullptr);
    if (rc != AR_OK) return rc;
    std::vector<EvalOut> out(n);
    std::vector<float> logits((size_t)n * 10);
    HIP_TRY(hipMemcpy(out.data(), d_out, sizeof(EvalOut) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(logits.data(), d_logits, logits.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        memcpy(pp1 + (size_t)i * 5, out[i].p1, 20);
        memcpy(pp2 + (size_t)i * 5, out[i].p2, 20);
        pv1[i] = out[i].v1;
        pv2[i] = out[i].v2;
        if (lg1) memcpy(lg1 + (size_t)i * 5, &logits[(size_t)i * 10], 20);
        if (lg2) memcpy(lg2 + (size_t)i * 5, &logits[(size_t)i * 10 + 5], 20);
    }
    return AR_OK;
}

template <int NW>
int encode_impl(const ArGameSpec* games, uint32_t n, float* obs) {
    std::vector<LeafReq<NW>> reqs;
    std::vector<Board> boards;
    std::vector<uint8_t> mazes;
    if (int rc = specs_to_device<NW>(games, n, reqs, boards, mazes)) return rc;
    int dim = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const int d = games[i].width * games[i].height * 7 + 6;
        if (i && d != dim) return fail(AR_E_INVALID, "ar_encode: all games must have the same board size");
        dim = d;
    }
    DevBuf<LeafReq<NW>> d_req;
    DevBuf<Board> d_boards;
    DevBuf<uint8_t> d_maze;
    DevBuf<float> d_obs;
    HIP_TRY(d_req.alloc(n));
    HIP_TRY(d_boards.alloc(n));
    HIP_TRY(d_maze.alloc(mazes.size()));
    HIP_TRY(d_obs.alloc((size_t)n * dim));
    HIP_TRY(hipMemcpy(d_req.p, reqs.data(), sizeof(LeafReq<NW>) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_boards.p, boards.data(), sizeof(Board) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_maze.p, mazes.data(), mazes.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(arnet::k_encode<NW>, dim3(n), dim3(128), 0, nullptr, d_req.p, n, d_boards.p, d_maze.p, d_obs.p, dim);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(obs, d_obs.p, (size_t)n * dim * 4, hipMemcpyDeviceToHost));
    return AR_OK;
}
}  // namespace

extern "C" {

int ar_net_evaluate(ArNet* net, const ArGameSpec* games, uint32_t n, float* policy_p1, float* policy_p2,
                    float* value_p1, float* value_p2, float* logits_p1, float* logits_p2) {
    if (!net || !games || !policy_p1 || !policy_p2 || !value_p1 || !value_p2) return fail(AR_E_INVALID, "null argument");
    if (n == 0) return AR_OK;
    return net->dev.hw <= 64
               ? net_evaluate_impl<1>(net, games, n, policy_p1, policy_p2, value_p1, value_p2, logits_p1, logits_p2)
               : net_evaluate_impl<4>(net, games, n, policy_p1, policy_p2, value_p1, value_p2, logits_p1, logits_p2);
}

int ar_encode(const ArGameSpec* games, uint32_t n, int device, float* obs) {
    if (!games || !obs) return fail(AR_E_INVALID, "null argument");
    if (n == 0) return AR_OK;
    int dev = 0;
    if (int rc = parse_device("hip", device, dev)) return rc;
    HIP_TRY(hipSetDevice(dev));
    bool small = true;
    for (uint32_t i = 0; i < n; ++i) small = small && (int)games[i].width * games[i].height <= 64;
    return small ? encode_impl<1>(games, n, obs) : encode_impl<4>(games, n, obs);
}

int ar_write_bundle(const ArGameRecordView* games, uint32_t n, const char* path) {
    if (!games || !path) return fail(AR_E_INVALID, "null argument");
    std::string err;
    if (!write_bundle(games, n, path, err)) return fail(AR_E_IO, err);
    return AR_OK;
}

}  // extern "C"
