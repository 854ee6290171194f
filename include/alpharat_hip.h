/* alpharat_hip.h -- C-ABI of libalpharat_hip.so, the MI355X-native self-play MCTS sampler.
 *
 * These entry points are what the reference's FFI for the sampling hot path binds today through
 * PyO3 (module pyrat_engine._core, crates/alpharat-mcts-python/src/lib.rs:10-39). Each function
 * names the reference interface it replaces. Conventions kept from the reference: the callee owns
 * all game / tree memory, the caller passes scalars, plain pointers and UTF-8 paths, results are
 * returned by value in caller-owned structs; functions return 0 on success and a negative AR_E_*
 * code on failure (message via ar_last_error), and never throw across the boundary.
 * No torch types appear here; the library needs a HIP device and fails loudly without one.
 */
#ifndef ALPHARAT_HIP_H
#define ALPHARAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AR_ABI_VERSION 1

enum {
    AR_OK = 0,
    AR_E_INVALID = -1,  /* bad argument (reference: ValueError / panic on unknown maze_type) */
    AR_E_BACKEND = -2,  /* evaluator failure (reference: BackendError -> PyRuntimeError)       */
    AR_E_IO = -3,       /* bundle / weight file I/O (reference: SelfPlayError::Io -> PyIOError) */
    AR_E_DEVICE = -4,   /* no HIP device, HIP API error, kernel fault                           */
    AR_E_NOMEM = -5     /* device arena / host allocation failure                               */
};

/* crates/alpharat-mcts/src/search.rs:18-58  SearchConfig (same fields, same defaults) */
typedef struct ArSearchConfig {
    float c_puct;               /* 1.5   */
    float fpu_reduction;        /* 0.2   */
    float force_k;              /* 2.0   */
    float noise_epsilon;        /* 0.0   */
    float noise_concentration;  /* 10.83 */
    uint32_t collision_limit_min;     /* 1     */
    uint32_t collision_limit_max;     /* 256   */
    uint32_t collision_scaling_start; /* 800   */
    uint32_t collision_scaling_end;   /* 50000 */
    float collision_scaling_power;    /* 1.0   */
} ArSearchConfig;

/* A full PyRat position: what `rust_mcts_search(game: PyRat, ...)` clones out of the engine
 * (crates/alpharat-mcts/src/bindings.rs:248). Cells are y-major, idx = y*width + x; directions
 * UP0 RIGHT1 DOWN2 LEFT3. cost[cell*4+dir]: 0 wall / board edge, 1 open, >=2 mud cost. */
typedef struct ArGameSpec {
    uint8_t width, height;
    uint16_t max_turns;
    uint16_t turn;
    uint8_t p1_x, p1_y, p2_x, p2_y;
    uint8_t p1_mud, p2_mud;
    float p1_score, p2_score;
    const uint8_t* cost;   /* [height*width*4]; NULL = open maze */
    const uint8_t* cheese; /* [height*width] 0/1 */
} ArGameSpec;

/* crates/alpharat-mcts/src/search.rs:304-325 SearchResult / bindings.rs:26-99 PySearchResult */
typedef struct ArSearchResult {
    float policy_p1[5], policy_p2[5];
    float value_p1, value_p2;
    float visit_counts_p1[5], visit_counts_p2[5];
    float prior_p1[5], prior_p2[5];
    uint32_t total_visits, nn_evals, terminals, collisions;
} ArSearchResult;

/* Leaf position handed to a host evaluator callback: what PyCallbackBackend wraps as PyRat
 * objects for `predict_fn(list[PyRat])` (crates/alpharat-mcts/src/bindings.rs:131-160). */
typedef struct ArLeaf {
    uint8_t p1_x, p1_y, p2_x, p2_y;
    uint8_t p1_mud, p2_mud;
    uint16_t turn;
    float p1_score, p2_score;
    uint64_t cheese_bits[4]; /* bit idx = y*width + x */
} ArLeaf;

/* predict_fn: fill policy_p1[n*5], policy_p2[n*5], value_p1[n], value_p2[n]; n <= batch_size.
 * Return 0, or non-zero to fail the search (-> AR_E_BACKEND, virtual losses reverted as in
 * search.rs:919-955). */
typedef int (*ArPredictFn)(void* user, const ArLeaf* leaves, uint32_t n, float* policy_p1, float* policy_p2,
                           float* value_p1, float* value_p2);

typedef struct ArNet ArNet; /* device-resident policy/value head (weights + workspace) */

const char* ar_version(void);
/* copies the calling thread's last error message; returns its length */
size_t ar_last_error(char* buf, size_t cap);
/* number of visible HIP devices, or AR_E_DEVICE */
int ar_device_count(void);
/* Game generation, host only (works without a device): the walls + mud (`cost_out[cells * 4]`: 0 wall / edge,
 * 1 open, >= 2 mud; directions UP RIGHT DOWN LEFT) and the cheese mask (`cheese_out[cells]`) that
 * ar_selfplay_run draws for the game seeded `seed` (= game_seed_base + game index). The reference leaves
 * both to the pyrat-rust engine (bindings.rs:502-532), which is not available: own generators, DESIGN.md
 * "game generation". Player cells are y * width + x. */
int ar_generate_maze(uint8_t width, uint8_t height, float wall_density, float mud_density, int symmetric, uint64_t seed,
                     uint8_t* cost_out);
int ar_generate_cheese(uint8_t width, uint8_t height, uint8_t p1_cell, uint8_t p2_cell, uint16_t count, int symmetric,
                       uint64_t seed, uint8_t* cheese_out);
/* hipDeviceSynchronize on `device` (benchmark bracketing) */
int ar_device_sync(int device);
/* The library keeps one tree-arena allocation per device between calls (the driver clears device
 * memory on allocation, seconds for a full MI355X; the reference's sampler likewise keeps its
 * thread pool and backend alive across play_games calls, selfplay.rs:600-660). This frees it, e.g.
 * before a training step that needs the HBM. AR_NO_ARENA_CACHE=1 in the environment disables the cache. */
int ar_release_device_memory(int device);

/* ---- evaluator: replaces OnnxBackend / TensorrtBackend (+ FlatEncoder) -----------------------
 * crates/alpharat-sampling/src/backends/onnx.rs:176-246, tensorrt.rs:423 ff., trt_shim.cpp:53-324.
 * `blob_path` is the weight blob written by alpharat_amd/weights.py from a .pt checkpoint. */
int ar_net_load(const char* blob_path, int device, ArNet** out);
void ar_net_free(ArNet* net);
/* Encode + forward `n` positions on the device. Outputs are host arrays:
 * policy_p1[n*5], policy_p2[n*5] (softmax), value_p1[n], value_p2[n] (softplus);
 * logits_p1/logits_p2 may be NULL. == model.predict(FlatObservationBuilder.build(...)) */
int ar_net_evaluate(ArNet* net, const ArGameSpec* games, uint32_t n, float* policy_p1, float* policy_p2,
                    float* value_p1, float* value_p2, float* logits_p1, float* logits_p2);
/* Device FlatEncoder (flat_encoder.rs:52-125): obs[n * (w*h*7+6)] on the host. */
int ar_encode(const ArGameSpec* games, uint32_t n, int device, float* obs);

/* ---- rust_mcts_search (crates/alpharat-mcts/src/bindings.rs:228-304) --------------------------
 * One search on a fresh tree. seed == NULL -> entropy. Evaluator: `net` if non-NULL, else
 * `predict_fn` if non-NULL (host callback per leaf batch), else SmartUniform. */
int ar_search(const ArGameSpec* game, const ArSearchConfig* cfg, uint32_t simulations, uint32_t batch_size,
              const uint64_t* seed, ArPredictFn predict_fn, void* user, ArNet* net, int device,
              ArSearchResult* out);
/* `n` independent searches in one launch set (evaluation-time callers: tournaments). seeds[n]. */
int ar_search_many(const ArGameSpec* games, uint32_t n, const ArSearchConfig* cfg, uint32_t simulations,
                   uint32_t batch_size, const uint64_t* seeds, ArNet* net, int device, ArSearchResult* out);

/* ---- rust_self_play (crates/alpharat-sampling/src/bindings.rs:268-483) ----------------------- */
typedef struct ArSelfPlayParams {
    /* game (make_games, bindings.rs:489-533) */
    uint8_t width, height;
    uint16_t cheese_count, max_turns;
    uint32_t num_games;
    int cheese_symmetric;       /* default 1 */
    const char* maze_type;      /* "open" | "classic" | "random" (generated mazes: our own seeded generator) */
    const char* positions;      /* "corners" | "random" */
    float wall_density, mud_density;
    int maze_symmetric;
    /* search */
    uint32_t simulations, batch_size;
    ArSearchConfig search;
    /* sampling */
    uint32_t num_threads;          /* accepted for drop-in compatibility; the device runs every game */
    const char* output_dir;        /* bundles go to output_dir/bundle_<uuid>.npz; NULL = keep in memory */
    uint32_t max_games_per_bundle; /* 32 */
    const char* weights_path;      /* NULL = SmartUniform; replaces onnx_model_path */
    const char* device;            /* "auto" | "hip" | "hip:N" | "mi355x"; anything else: AR_E_INVALID */
    uint32_t mux_max_batch_size;   /* accepted, unused: leaves are batched device-wide */
    uint64_t cache_size;           /* NN-eval cache entries, 0 = off */
    /* extensions (not in the reference signature) */
    int has_seed;                  /* 0: entropy (reference behaviour), 1: game i uses seeds below */
    uint64_t game_seed_base;       /* cheese layout of game i: game_seed_base + i */
    uint64_t rng_seed_base;        /* search/sampling stream of game i: rng_seed_base + i */
    uint32_t first_game_index;     /* shard offset: this call plays games first..first+num_games */
    uint32_t concurrent_games;     /* device-resident games (0 = choose from HBM size) */
    int device_index;              /* used when device == "auto"/"hip" */
} ArSelfPlayParams;

/* crates/alpharat-sampling/src/selfplay.rs:136-158 (+ derived rates computed by the caller) */
typedef struct ArSelfPlayStats {
    uint32_t total_games;
    uint64_t total_positions, total_simulations;
    double elapsed_secs;
    uint32_t p1_wins, p2_wins, draws;
    float total_cheese_collected;
    uint32_t total_cheese_available;
    uint32_t min_turns, max_turns;
    uint64_t total_nn_evals, total_terminals, total_collisions;
    uint64_t cache_hits, cache_misses;
    /* instrumentation for the roofline: tree levels traversed */
    uint64_t gather_node_visits, backup_node_visits, new_nodes;
    double device_secs; /* time inside step kernels (HIP events) */
    uint64_t steps;     /* batch steps launched */
    double gather_secs; /* time inside the gather kernel alone: HIP events around every launch, on its stream */
    uint64_t gather_launches;
} ArSelfPlayStats;

/* crates/alpharat-sampling/src/selfplay.rs:343-359: live counters in caller-owned memory, written
 * with relaxed atomic stores while ar_selfplay_run is blocking on another thread. */
typedef struct ArProgress {
    volatile uint32_t games_completed;
    volatile uint64_t positions_completed, simulations_completed, nn_evals_completed;
} ArProgress;

/* One finished game as the bundle writer sees it (selfplay.rs:80-132); arrays are only valid
 * during the callback. */
typedef struct ArGameRecordView {
    uint8_t width, height;
    uint16_t max_turns;
    uint32_t game_index, n_positions;
    const int8_t* maze;            /* [h*w*4] */
    const uint8_t* initial_cheese; /* [h*w]   */
    const uint8_t* cheese_outcomes;/* [h*w]   */
    float final_p1_score, final_p2_score;
    uint8_t result;                /* 0 draw 1 P1 2 P2 */
    uint16_t cheese_available;
    uint64_t total_simulations, total_nn_evals, total_terminals, total_collisions;
    /* position arrays, n_positions rows */
    const uint8_t* p1_pos;  /* [n*2] */
    const uint8_t* p2_pos;  /* [n*2] */
    const float* p1_score;  const float* p2_score;
    const uint8_t* p1_mud;  const uint8_t* p2_mud;
    const uint16_t* turn;
    const uint8_t* cheese_mask;   /* [n*h*w] */
    const float* value_p1;  const float* value_p2;
    const float* visit_counts_p1; const float* visit_counts_p2; /* [n*5] */
    const float* prior_p1;  const float* prior_p2;
    const float* policy_p1; const float* policy_p2;
    const uint8_t* action_p1; const uint8_t* action_p2;
} ArGameRecordView;
typedef void (*ArGameSink)(void* user, const ArGameRecordView* game);

/* Blocking. Plays params->num_games games, streams bundles to output_dir (atomic tmp->rename,
 * recording.rs:121-161) and/or hands each finished game to `sink`. */
int ar_selfplay_run(const ArSelfPlayParams* params, ArProgress* progress, ArGameSink sink, void* sink_user,
                    ArSelfPlayStats* out);

/* ---- the same run in slices (extension; the reference's worker pool lives for one call, selfplay.rs:721-808).
 * A session keeps the device engine, its resident games and the supply of new games alive between calls:
 *   ar_selfplay_open   sets up `concurrent_games` resident games (num_games == UINT32_MAX: the supply never
 *                      ends; game indices continue from first_game_index and wrap);
 *   ar_selfplay_step   runs `batch_steps` simulate_batch steps (search.rs:961) for every resident game
 *                      (UINT32_MAX: until all games are finished), with finished games drained to the sink /
 *                      bundle writer and their slots refilled, and reports in `window` what was done inside
 *                      this call: games finished, and positions / simulations / evaluations / node-visits per
 *                      finished MOVE (games still in flight count with the moves they completed);
 *                      *finished = 1 when no game is left;
 *   ar_selfplay_close  flushes the bundles, returns the totals of the finished games and frees everything.
 * ar_selfplay_run(p, ...) == open; step(UINT32_MAX); close. */
typedef struct ArSelfPlaySession ArSelfPlaySession;
int ar_selfplay_open(const ArSelfPlayParams* params, ArProgress* progress, ArGameSink sink, void* sink_user,
                     ArSelfPlaySession** out);
int ar_selfplay_step(ArSelfPlaySession* session, uint32_t batch_steps, ArSelfPlayStats* window, int* finished);
int ar_selfplay_close(ArSelfPlaySession* session, ArSelfPlayStats* total);

/* What the library chose for a session (the resident count may be smaller than `concurrent_games`: it is bounded by the
 * device memory the trees need). gather_kind: 0 one lane per game (k_gather), 1 eight lanes per game (k_gather8), 2 the
 * work queue over tree levels (k_gatherw), 3 fused SmartUniform step (k_step_uniform). */
typedef struct ArSessionInfo {
    uint32_t resident_games, groups, gather_kind;
    uint32_t gather_pass_limit; /* k_gatherw: passes per launch (UINT32_MAX: no limit) */
    uint64_t tree_region_bytes; /* device memory set aside for the trees */
    uint64_t host_grown_arenas; /* trees that outgrew a run of pages and were moved to an arena of their own */
    uint32_t idle_slots;        /* slots waiting for room in their zone of the tree region (resident games = resident_games - idle_slots) */
    float tree_pages_per_game;  /* what the resident set is sized by: pages a game of this run holds after a few moves */
} ArSessionInfo;
int ar_selfplay_info(const ArSelfPlaySession* session, ArSessionInfo* out);

/* Bundle writer on its own (recording.rs:23-162 write_bundle): used by the known-answer test. */
int ar_write_bundle(const ArGameRecordView* games, uint32_t n, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* ALPHARAT_HIP_H */
