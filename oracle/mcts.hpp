// TEST INFRASTRUCTURE ONLY -- CPU oracle (see rng.hpp header).
//
// Restatement of the reference's decoupled-PUCT search: outcome-indexed nodes, the tree
// lifecycle and the LC0-style batched gather / evaluate / backup loop.
//   node.rs   -> HalfEdge, HalfNode, compute_outcomes, Node
//   tree.rs   -> find_child, smart_uniform_prior, extend_node, populate_node,
//                find_or_extend_child, MCTSTree (advance_root / reinit / recount)
//   search.rs -> compute_fpu, puct_score, estimated_visits_to_change_best_half,
//                build_gather_level, pick_nodes_to_extend, backup_and_finalize,
//                cancel_shared_collisions, calculate_collisions_left, simulate_batch,
//                apply_dirichlet_noise, compute_pruned_visits, extract_half, run_search
// Each function cites the reference lines it follows. Floating-point operation order is kept
// exactly (f32 everywhere the reference uses f32; compile with -ffp-contract=off).
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "pyrat_engine.hpp"
#include "rng.hpp"

namespace oracle {

// ---- node.rs:56-121 -------------------------------------------------------------------------
struct HalfEdge {
    float q = 0.0f;
    uint32_t visits = 0;
    uint32_t n_in_flight = 0;
    // node.rs:82-85
    void update_multivisit(float value, uint32_t count) {
        visits += count;
        q += (value - q) * (float)count / (float)visits;
    }
    uint32_t n_started() const { return visits + n_in_flight; }
};

// ---- node.rs:251-283 ------------------------------------------------------------------------
inline void compute_outcomes(const uint8_t effective[5], uint8_t outcomes[5], uint8_t& n_out,
                             uint8_t action_to_idx[5]) {
    uint8_t unique[5] = {0, 0, 0, 0, 0};
    int n = 0;
    for (int k = 0; k < 5; ++k) {
        uint8_t val = effective[k];
        int pos = 0;
        while (pos < n && unique[pos] < val) ++pos;
        if (pos < n && unique[pos] == val) continue;
        for (int i = n; i > pos; --i) unique[i] = unique[i - 1];
        unique[pos] = val;
        ++n;
    }
    for (int a = 0; a < 5; ++a) {
        int i = 0;
        while (i < n && unique[i] < effective[a]) ++i;
        action_to_idx[a] = (uint8_t)i;
    }
    for (int i = 0; i < 5; ++i) outcomes[i] = unique[i];
    n_out = (uint8_t)n;
}

// ---- node.rs:130-241 ------------------------------------------------------------------------
struct HalfNode {
    float prior[5] = {0, 0, 0, 0, 0};
    HalfEdge edges[5];
    uint8_t outcomes[5] = {0, 0, 0, 0, 0};
    uint8_t action_to_idx[5] = {0, 0, 0, 0, 0};
    uint8_t n_outcomes = 0;

    static HalfNode new_shell(const uint8_t effective[5]) {
        HalfNode h;
        compute_outcomes(effective, h.outcomes, h.n_outcomes, h.action_to_idx);
        return h;
    }
    // node.rs:173-179: clear, then scatter-add in action order 0..4
    void set_prior(const float prior5[5]) {
        for (int i = 0; i < 5; ++i) prior[i] = 0.0f;
        for (int a = 0; a < 5; ++a) prior[action_to_idx[a]] += prior5[a];
    }
    void expand_prior(float out[5]) const {
        for (int i = 0; i < 5; ++i) out[i] = 0.0f;
        for (int i = 0; i < n_outcomes; ++i) out[outcomes[i]] = prior[i];
    }
};

// ---- node.rs:289-458 ------------------------------------------------------------------------
struct Node {
    HalfNode p1, p2;
    float v1 = 0.0f, v2 = 0.0f;
    uint32_t total_visits = 0;
    uint32_t n_in_flight = 0;
    float value_scale = 0.0f;
    float edge_r1 = 0.0f, edge_r2 = 0.0f;
    Node* first_child = nullptr;
    Node* next_sibling = nullptr;
    Node* parent = nullptr;
    uint8_t po1 = 0, po2 = 0;  // parent_outcome
    bool is_terminal = false;

    uint32_t children_visits() const { return total_visits > 0 ? total_visits - 1 : 0; }
    // node.rs:388-394
    bool try_start_score_update() {
        if (total_visits == 0 && n_in_flight > 0) return false;
        n_in_flight += 1;
        return true;
    }
    // node.rs:444-457
    void finalize_score_update(float q1, float q2, uint32_t multivisit) {
        total_visits += multivisit;
        float n = (float)total_visits;
        float w = (float)multivisit;
        v1 += (q1 - v1) * w / n;
        v2 += (q2 - v2) * w / n;
        n_in_flight -= multivisit;
    }
};

inline void free_subtree(Node* root) {
    if (!root) return;
    std::vector<Node*> stack;
    stack.push_back(root);
    while (!stack.empty()) {
        Node* n = stack.back();
        stack.pop_back();
        if (n->first_child) stack.push_back(n->first_child);
        if (n->next_sibling) stack.push_back(n->next_sibling);
        delete n;
    }
}

// ---- tree.rs:52-63 --------------------------------------------------------------------------
inline Node* find_child(Node* parent, uint8_t i, uint8_t j) {
    for (Node* c = parent->first_child; c; c = c->next_sibling)
        if (c->po1 == i && c->po2 == j) return c;
    return nullptr;
}

// ---- tree.rs:69-84 --------------------------------------------------------------------------
inline void smart_uniform_prior(const uint8_t effective[5], float prior[5]) {
    bool seen[5] = {false, false, false, false, false};
    int count = 0;
    for (int k = 0; k < 5; ++k)
        if (!seen[effective[k]]) {
            seen[effective[k]] = true;
            ++count;
        }
    float p = 1.0f / (float)count;
    for (int k = 0; k < 5; ++k) prior[k] = 0.0f;
    for (int k = 0; k < 5; ++k) prior[effective[k]] = p;
}

// ---- backend.rs:57-62, 75-82 ----------------------------------------------------------------
struct EvalResult {
    float policy_p1[5], policy_p2[5];
    float value_p1, value_p2;
};
// evaluate_batch(&[&GameState]) -> Result<Vec<EvalResult>, BackendError>; false = error
using Backend = std::function<bool(const std::vector<const GameState*>&, std::vector<EvalResult>&, std::string&)>;

// backend.rs:92-103 (and ConstantValueBackend :114-129 when v1/v2 are non-zero)
inline Backend smart_uniform_backend(float v1 = 0.0f, float v2 = 0.0f) {
    return [v1, v2](const std::vector<const GameState*>& games, std::vector<EvalResult>& out, std::string&) {
        out.resize(games.size());
        for (size_t i = 0; i < games.size(); ++i) {
            uint8_t e1[5], e2[5];
            games[i]->effective_actions_p1(e1);
            games[i]->effective_actions_p2(e2);
            smart_uniform_prior(e1, out[i].policy_p1);
            smart_uniform_prior(e2, out[i].policy_p2);
            out[i].value_p1 = v1;
            out[i].value_p2 = v2;
        }
        return true;
    };
}

// ---- tree.rs:238-365 ------------------------------------------------------------------------
struct MCTSTree {
    Node* root = nullptr;
    uint32_t node_count = 0;

    static Node* alloc_root(const GameState& game) {  // tree.rs:351-365
        uint8_t e1[5], e2[5];
        float pr1[5], pr2[5];
        game.effective_actions_p1(e1);
        game.effective_actions_p2(e2);
        smart_uniform_prior(e1, pr1);
        smart_uniform_prior(e2, pr2);
        Node* n = new Node();
        n->p1 = HalfNode::new_shell(e1);
        n->p1.set_prior(pr1);
        n->p2 = HalfNode::new_shell(e2);
        n->p2.set_prior(pr2);
        uint16_t rc = game.remaining_cheese;
        n->value_scale = (float)(rc > 1 ? rc : 1);
        return n;
    }
    explicit MCTSTree(const GameState& game) : root(alloc_root(game)), node_count(1) {}
    ~MCTSTree() { free_subtree(root); }
    MCTSTree(const MCTSTree&) = delete;
    MCTSTree& operator=(const MCTSTree&) = delete;

    static uint32_t count_subtree_nodes(const Node* r) {  // tree.rs:209-226
        uint32_t count = 1;
        std::vector<const Node*> stack;
        if (r->first_child) stack.push_back(r->first_child);
        while (!stack.empty()) {
            const Node* n = stack.back();
            stack.pop_back();
            ++count;
            if (n->first_child) stack.push_back(n->first_child);
            if (n->next_sibling) stack.push_back(n->next_sibling);
        }
        return count;
    }

    // tree.rs:283-295, detach_child :305-339
    bool advance_root(uint8_t a1, uint8_t a2) {
        uint8_t i = root->p1.action_to_idx[a1];
        uint8_t j = root->p2.action_to_idx[a2];
        Node** link = &root->first_child;
        Node* found = nullptr;
        while (*link) {
            if ((*link)->po1 == i && (*link)->po2 == j) {
                found = *link;
                *link = found->next_sibling;
                found->next_sibling = nullptr;
                found->parent = nullptr;
                break;
            }
            link = &(*link)->next_sibling;
        }
        if (!found) return false;
        free_subtree(root);
        root = found;
        node_count = count_subtree_nodes(root);
        return true;
    }
    void reinit(const GameState& game) {  // tree.rs:298-302
        free_subtree(root);
        root = alloc_root(game);
        node_count = 1;
    }
};

// ---- tree.rs:107-201 ------------------------------------------------------------------------
inline Node* find_or_extend_child(Node* parent, uint8_t o1, uint8_t o2, const GameState& game, float r1,
                                  float r2, bool& is_new) {
    if (Node* c = find_child(parent, o1, o2)) {
        is_new = false;
        return c;
    }
    uint8_t e1[5], e2[5];
    game.effective_actions_p1(e1);
    game.effective_actions_p2(e2);
    Node* n = new Node();
    n->p1 = HalfNode::new_shell(e1);
    n->p2 = HalfNode::new_shell(e2);
    uint16_t rc = game.remaining_cheese;
    n->value_scale = (float)(rc > 1 ? rc : 1);
    n->parent = parent;
    n->po1 = o1;
    n->po2 = o2;
    n->next_sibling = parent->first_child;  // prepend, tree.rs:139-145
    parent->first_child = n;
    n->edge_r1 = r1;
    n->edge_r2 = r2;
    is_new = true;
    return n;
}

// ---- search.rs:18-58 ------------------------------------------------------------------------
struct SearchConfig {
    float c_puct = 1.5f;
    float fpu_reduction = 0.2f;
    float force_k = 2.0f;
    float noise_epsilon = 0.0f;
    float noise_concentration = 10.83f;
    uint32_t collision_limit_min = 1;
    uint32_t collision_limit_max = 256;
    uint32_t collision_scaling_start = 800;
    uint32_t collision_scaling_end = 50000;
    float collision_scaling_power = 1.0f;
};

// ---- search.rs:304-325 ----------------------------------------------------------------------
struct SearchResult {
    float policy_p1[5], policy_p2[5];
    float value_p1, value_p2;
    float visit_counts_p1[5], visit_counts_p2[5];
    float prior_p1[5], prior_p2[5];
    uint32_t total_visits, nn_evals, terminals, collisions;
};

// Instrumentation the reference does not have: node-visits (tree levels traversed by allocated
// visits in gather, and levels walked in backup) so algorithmic bytes can be priced.
struct SearchCounters {
    uint64_t gather_node_visits = 0;  // build_gather_level calls weighted by nothing: one per level
    uint64_t backup_node_visits = 0;  // nodes touched by backup_and_finalize
    uint64_t new_nodes = 0;
};

static const float FORCED_PLAYOUT_SCORE = 1e20f;  // search.rs:10

// search.rs:120-128
inline float compute_fpu(const HalfNode& half, float node_value, float value_scale, float fpu_reduction) {
    float visited_prior_mass = 0.0f;
    for (int i = 0; i < half.n_outcomes; ++i)
        if (half.edges[i].visits > 0) visited_prior_mass += half.prior[i];
    return node_value - fpu_reduction * value_scale * std::sqrt(visited_prior_mass);
}

// search.rs:139-152
inline void puct_score(const HalfEdge& edge, float prior, float fpu, float value_scale, float c_puct,
                       float sqrt_total, float nstarted, float& score, float& q_norm) {
    float q = edge.visits > 0 ? edge.q : fpu;
    q_norm = q / value_scale;
    float exploration = c_puct * prior * sqrt_total / (1.0f + nstarted);
    score = q_norm + exploration;
}

// search.rs:463-554
inline void estimated_visits_to_change_best_half(const HalfNode& half, float node_value, float value_scale,
                                                 uint32_t children_visits, const SearchConfig& config,
                                                 bool is_root, const uint32_t nstarted[5], SmallRng& rng,
                                                 uint8_t& best_out, uint32_t& vtc_out) {
    const int n = half.n_outcomes;
    if (n <= 1) {
        best_out = 0;
        vtc_out = UINT32_MAX;
        return;
    }
    const float fpu = compute_fpu(half, node_value, value_scale, config.fpu_reduction);
    const uint32_t cv1 = children_visits > 1 ? children_visits : 1;
    const float sqrt_total = std::sqrt((float)cv1);
    const float c_puct = config.c_puct;
    const float NEG_INF = -std::numeric_limits<float>::infinity();

    uint8_t best_idx = 0;
    float best_score = NEG_INF, best_utility = NEG_INF, second_best_score = NEG_INF;

    auto scored = [&](int i, float& score, float& q_norm) {
        const HalfEdge& edge = half.edges[i];
        float prior = half.prior[i];
        puct_score(edge, prior, fpu, value_scale, c_puct, sqrt_total, (float)nstarted[i], score, q_norm);
        if (is_root && config.force_k > 0.0f && prior > 0.0f) {
            float threshold = std::sqrt(config.force_k * prior * (float)children_visits);
            if ((float)edge.visits < threshold) score = FORCED_PLAYOUT_SCORE;
        }
    };

    for (int i = 0; i < n; ++i) {
        float score, q_norm;
        scored(i, score, q_norm);
        if (score > best_score) {
            second_best_score = best_score;
            best_score = score;
            best_idx = (uint8_t)i;
            best_utility = q_norm;
        } else if (score > second_best_score) {
            second_best_score = score;
        }
    }

    // ties: reservoir sampling in a second pass (search.rs:511-532)
    uint32_t tie_count = 1;
    for (int i = 0; i < n; ++i) {
        if ((uint8_t)i == best_idx) continue;
        float score, q_norm;
        scored(i, score, q_norm);
        if (std::fabs(score - best_score) < 1e-12f) {
            tie_count += 1;
            if (rng.gen_range_u32(tie_count) == 0) {
                best_idx = (uint8_t)i;
                best_utility = q_norm;
            }
        }
    }

    best_out = best_idx;
    if (second_best_score <= NEG_INF) {
        vtc_out = UINT32_MAX;
        return;
    }
    if (best_utility >= second_best_score) {
        vtc_out = UINT32_MAX;
        return;
    }
    float prior_best = half.prior[best_idx];
    float n1 = (float)nstarted[best_idx] + 1.0f;
    float denom = second_best_score - best_utility;
    if (denom <= 0.0f) {
        vtc_out = UINT32_MAX;
        return;
    }
    float vtc = c_puct * prior_best * sqrt_total / denom - n1 + 1.0f;
    if (!(vtc > 1.0f)) vtc = 1.0f;  // f32::max(1.0): NaN -> 1.0
    // Rust `as u32` saturates
    uint32_t k = vtc >= 4294967296.0f ? UINT32_MAX : (uint32_t)vtc;
    vtc_out = k > 1 ? k : 1;
}

// search.rs:561-569
struct GatherLevel {
    Node* node;
    uint32_t vtp[25];
    int next_idx, last_idx;
};

// search.rs:742-817
inline GatherLevel build_gather_level(Node* node, uint32_t cur_limit, const SearchConfig& config, bool is_root,
                                      SmallRng& rng) {
    const int n1 = node->p1.n_outcomes, n2 = node->p2.n_outcomes;
    const uint32_t children_visits = node->children_visits();
    const float value_scale = node->value_scale;
    const float v1 = node->v1, v2 = node->v2;
    uint32_t ns_p1[5] = {0, 0, 0, 0, 0}, ns_p2[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n1; ++i) ns_p1[i] = node->p1.edges[i].n_started();
    for (int j = 0; j < n2; ++j) ns_p2[j] = node->p2.edges[j].n_started();
    uint32_t orig1[5], orig2[5];
    for (int i = 0; i < 5; ++i) {
        orig1[i] = ns_p1[i];
        orig2[i] = ns_p2[i];
    }
    GatherLevel lvl;
    lvl.node = node;
    for (int i = 0; i < 25; ++i) lvl.vtp[i] = 0;
    uint32_t remaining = cur_limit;
    int last_idx = 0;
    while (remaining > 0) {
        uint8_t best1, best2;
        uint32_t vtcb1, vtcb2;
        estimated_visits_to_change_best_half(node->p1, v1, value_scale, children_visits, config, is_root, ns_p1,
                                             rng, best1, vtcb1);
        estimated_visits_to_change_best_half(node->p2, v2, value_scale, children_visits, config, is_root, ns_p2,
                                             rng, best2, vtcb2);
        uint32_t k = remaining;
        if (vtcb1 < k) k = vtcb1;
        if (vtcb2 < k) k = vtcb2;
        if (k < 1) k = 1;
        int flat = (int)best1 * 5 + (int)best2;
        lvl.vtp[flat] += k;
        ns_p1[best1] += k;
        ns_p2[best2] += k;
        remaining -= k;
        if (flat > last_idx && lvl.vtp[flat] > 0) last_idx = flat;
    }
    for (int i = 0; i < n1; ++i) {
        uint32_t delta = ns_p1[i] - orig1[i];
        if (delta > 0) node->p1.edges[i].n_in_flight += delta;
    }
    for (int j = 0; j < n2; ++j) {
        uint32_t delta = ns_p2[j] - orig2[j];
        if (delta > 0) node->p2.edges[j].n_in_flight += delta;
    }
    lvl.next_idx = 0;
    lvl.last_idx = last_idx;
    return lvl;
}

// search.rs:339-351
struct NodeToProcess {
    Node* node;
    bool needs_eval;  // false = Terminal
    GameState game_state;
    uint32_t multivisit;
};
struct Collision {
    Node* node;
    uint32_t multivisit;
};

// search.rs:576-738
inline void pick_nodes_to_extend(MCTSTree& tree, const GameState& game, const SearchConfig& config,
                                 uint32_t budget, SmallRng& rng, std::vector<NodeToProcess>& to_process,
                                 std::vector<Collision>& shared_collisions, SearchCounters* ctr) {
    Node* root = tree.root;
    GameState work_game = game;
    std::vector<MoveUndo> undos;
    const uint32_t cur_limit = budget;

    if (root->total_visits == 0 || root->is_terminal) {
        if (root->total_visits == 0 && !root->is_terminal) {
            if (root->try_start_score_update()) {
                if (work_game.check_game_over()) {
                    root->is_terminal = true;  // populate_node(root, None)
                    to_process.push_back(NodeToProcess{root, false, GameState(), 1});
                } else {
                    to_process.push_back(NodeToProcess{root, true, work_game, 1});
                }
                if (cur_limit > 1) shared_collisions.push_back(Collision{root, cur_limit - 1});
            } else {
                shared_collisions.push_back(Collision{root, cur_limit});
            }
        } else {
            if (root->total_visits == 0) root->is_terminal = true;
            if (root->try_start_score_update()) {
                to_process.push_back(NodeToProcess{root, false, GameState(), 1});
                if (cur_limit > 1) shared_collisions.push_back(Collision{root, cur_limit - 1});
            } else {
                shared_collisions.push_back(Collision{root, cur_limit});
            }
        }
        return;
    }

    root->n_in_flight += cur_limit;
    std::vector<GatherLevel> levels;
    levels.push_back(build_gather_level(root, cur_limit, config, true, rng));
    if (ctr) ctr->gather_node_visits += 1;

    while (!levels.empty()) {
        bool found_child = false;
        while (levels.back().next_idx <= levels.back().last_idx) {
            GatherLevel& level = levels.back();
            int idx = level.next_idx;
            level.next_idx += 1;
            if (level.vtp[idx] == 0) continue;
            uint8_t a1 = (uint8_t)(idx / 5), a2 = (uint8_t)(idx % 5);
            uint32_t k = level.vtp[idx];
            Node* lnode = level.node;
            uint8_t act1 = lnode->p1.outcomes[a1], act2 = lnode->p2.outcomes[a2];
            float sb1 = work_game.player1_score(), sb2 = work_game.player2_score();
            MoveUndo undo = work_game.make_move(act1, act2);
            float r1 = work_game.player1_score() - sb1;  // compute_rewards, tree.rs:89-94
            float r2 = work_game.player2_score() - sb2;

            bool is_new;
            Node* child = find_or_extend_child(lnode, a1, a2, work_game, r1, r2, is_new);
            if (is_new) {
                tree.node_count += 1;
                if (ctr) ctr->new_nodes += 1;
            }

            if (child->total_visits == 0 || child->is_terminal) {
                if (child->try_start_score_update()) {
                    if (child->is_terminal || work_game.check_game_over()) {
                        if (child->total_visits == 0) child->is_terminal = true;
                        to_process.push_back(NodeToProcess{child, false, GameState(), 1});
                        if (k > 1) shared_collisions.push_back(Collision{child, k - 1});
                    } else {
                        to_process.push_back(NodeToProcess{child, true, work_game, 1});
                        if (k > 1) shared_collisions.push_back(Collision{child, k - 1});
                    }
                } else {
                    shared_collisions.push_back(Collision{child, k});
                }
                work_game.unmake_move(undo);
            } else {
                if (child->try_start_score_update()) {
                    if (k > 1) child->n_in_flight += k - 1;
                    undos.push_back(undo);
                    GatherLevel child_level = build_gather_level(child, k, config, false, rng);
                    if (ctr) ctr->gather_node_visits += 1;
                    levels.push_back(child_level);  // invalidates `level`
                    found_child = true;
                    break;
                } else {
                    shared_collisions.push_back(Collision{child, k});
                    work_game.unmake_move(undo);
                }
            }
        }
        if (!found_child) {
            levels.pop_back();
            if (!undos.empty()) {
                work_game.unmake_move(undos.back());
                undos.pop_back();
            }
        }
    }
}

// search.rs:826-852
inline void backup_and_finalize(Node* leaf, float g1, float g2, uint32_t multivisit, SearchCounters* ctr) {
    leaf->finalize_score_update(g1, g2, multivisit);
    if (ctr) ctr->backup_node_visits += 1;
    float v1 = g1, v2 = g2;
    Node* current = leaf;
    while (Node* parent = current->parent) {
        uint8_t a1 = current->po1, a2 = current->po2;
        float q1 = current->edge_r1 + v1;
        float q2 = current->edge_r2 + v2;
        parent->finalize_score_update(q1, q2, multivisit);
        parent->p1.edges[a1].update_multivisit(q1, multivisit);
        parent->p2.edges[a2].update_multivisit(q2, multivisit);
        parent->p1.edges[a1].n_in_flight -= multivisit;
        parent->p2.edges[a2].n_in_flight -= multivisit;
        if (ctr) ctr->backup_node_visits += 1;
        v1 = q1;
        v2 = q2;
        current = parent;
    }
}

// search.rs:860-889
inline void cancel_shared_collisions(const std::vector<Collision>& collisions, Node* root) {
    for (const Collision& c : collisions) {
        Node* current = c.node;
        while (Node* parent = current->parent) {
            parent->n_in_flight -= c.multivisit;
            parent->p1.edges[current->po1].n_in_flight -= c.multivisit;
            parent->p2.edges[current->po2].n_in_flight -= c.multivisit;
            if (parent == root) break;
            current = parent;
        }
    }
}

// search.rs:899-910
inline void cancel_leaf_and_path(Node* leaf, uint32_t multivisit) {
    leaf->n_in_flight -= multivisit;
    Node* current = leaf;
    while (Node* parent = current->parent) {
        parent->n_in_flight -= multivisit;
        parent->p1.edges[current->po1].n_in_flight -= multivisit;
        parent->p2.edges[current->po2].n_in_flight -= multivisit;
        current = parent;
    }
}

// search.rs:437-450
inline uint32_t calculate_collisions_left(uint32_t tree_node_count, const SearchConfig& c) {
    if (tree_node_count >= c.collision_scaling_end) return c.collision_limit_max;
    if (tree_node_count <= c.collision_scaling_start) return c.collision_limit_min;
    float ratio = (float)(tree_node_count - c.collision_scaling_start) /
                  (float)(c.collision_scaling_end - c.collision_scaling_start);
    float scaled = (float)c.collision_limit_min +
                   ((float)c.collision_limit_max - (float)c.collision_limit_min) *
                       std::pow(ratio, c.collision_scaling_power);
    float r = std::round(scaled);  // f32::round: half away from zero
    uint32_t v = r <= 0.0f ? 0u : (r >= 4294967296.0f ? UINT32_MAX : (uint32_t)r);
    if (v < c.collision_limit_min) v = c.collision_limit_min;
    if (v > c.collision_limit_max) v = c.collision_limit_max;
    return v;
}

// search.rs:400-429. Only shapes >= 1 are implemented (concentration / n_outcomes >= 1);
// smaller shapes take a different sampler in rand_distr and leave the priors untouched here
// with `ok=false` so the caller can flag it.
inline bool apply_dirichlet_noise(HalfNode& half, float epsilon, float concentration, SmallRng& rng) {
    int n = half.n_outcomes;
    if (n <= 1) return true;
    double alpha = (double)(concentration / (float)n);
    if (!(alpha > 0.0)) return true;  // Gamma::new error -> return
    if (alpha < 1.0 || alpha == 1.0) return false;
    GammaLarge g = GammaLarge::make(alpha, 1.0);
    float noise[5] = {0, 0, 0, 0, 0};
    float total = 0.0f;
    for (int i = 0; i < n; ++i) {
        noise[i] = (float)g.sample(rng);
        total += noise[i];
    }
    if (total < std::numeric_limits<float>::min()) return true;
    for (int i = 0; i < n; ++i) half.prior[i] = half.prior[i] * (1.0f - epsilon) + epsilon * noise[i] / total;
    return true;
}

struct BatchStats {
    uint32_t nn_evals = 0, terminals = 0, collisions = 0;
};

// search.rs:961-1073
inline bool simulate_batch(MCTSTree& tree, const GameState& game, const Backend& backend,
                           const SearchConfig& config, uint32_t batch_size, SmallRng& rng, BatchStats& out,
                           std::string& err, SearchCounters* ctr) {
    Node* root = tree.root;
    int64_t collisions_left = (int32_t)calculate_collisions_left(tree.node_count, config);
    std::vector<NodeToProcess> all_to_process;
    std::vector<Collision> all_collisions;
    uint32_t minibatch_size = 0, terminals = 0;

    while (minibatch_size < batch_size && collisions_left > 0) {
        uint32_t budget = (uint32_t)collisions_left;
        if (batch_size - minibatch_size < budget) budget = batch_size - minibatch_size;
        std::vector<NodeToProcess> to_process;
        std::vector<Collision> shared;
        pick_nodes_to_extend(tree, game, config, budget, rng, to_process, shared, ctr);
        for (NodeToProcess& e : to_process) {
            if (!e.needs_eval) terminals += e.multivisit;
            minibatch_size += 1;
            all_to_process.push_back(std::move(e));
        }
        for (const Collision& c : shared) {
            collisions_left -= (int32_t)c.multivisit;
            all_collisions.push_back(c);
        }
    }

    uint32_t nn_evals = 0, total_collisions = 0;
    std::vector<const GameState*> states;
    for (const NodeToProcess& e : all_to_process)
        if (e.needs_eval) {
            ++nn_evals;
            states.push_back(&e.game_state);
        }
    for (const Collision& c : all_collisions) total_collisions += c.multivisit;

    std::vector<EvalResult> evals;
    if (!states.empty()) {
        if (!backend(states, evals, err) || evals.size() != states.size()) {
            // GatherCleanupGuard, search.rs:945-955
            for (const NodeToProcess& e : all_to_process) cancel_leaf_and_path(e.node, e.multivisit);
            cancel_shared_collisions(all_collisions, root);
            if (err.empty()) err = "backend returned a wrong number of results";
            return false;
        }
    }

    size_t eval_idx = 0;
    for (const NodeToProcess& e : all_to_process) {
        if (e.needs_eval) {
            const EvalResult& ev = evals[eval_idx++];
            e.node->p1.set_prior(ev.policy_p1);  // populate_node(Some)
            e.node->p2.set_prior(ev.policy_p2);
            if (e.node == root && config.noise_epsilon > 0.0f) {
                bool ok1 = apply_dirichlet_noise(e.node->p1, config.noise_epsilon, config.noise_concentration, rng);
                bool ok2 = apply_dirichlet_noise(e.node->p2, config.noise_epsilon, config.noise_concentration, rng);
                if (!ok1 || !ok2) {
                    err = "oracle: Dirichlet noise with concentration/n_outcomes <= 1 is not restated";
                    return false;
                }
            }
            backup_and_finalize(e.node, ev.value_p1, ev.value_p2, e.multivisit, ctr);
        } else {
            backup_and_finalize(e.node, 0.0f, 0.0f, e.multivisit, ctr);
        }
    }
    cancel_shared_collisions(all_collisions, root);
    out.nn_evals = nn_evals;
    out.terminals = terminals;
    out.collisions = total_collisions;
    return true;
}

// search.rs:249-296
inline void compute_pruned_visits(const float* q_norm, const float* prior, const float* visits, int n,
                                  uint32_t parent_visits, float c_puct, float result[5]) {
    for (int i = 0; i < 5; ++i) result[i] = 0.0f;
    if (n <= 1) {
        if (n == 1) result[0] = visits[0];
        return;
    }
    int best_idx = 0;
    float best_visits = visits[0];
    for (int i = 1; i < n; ++i)
        if (visits[i] > best_visits) {
            best_visits = visits[i];
            best_idx = i;
        }
    uint32_t pv = parent_visits > 1 ? parent_visits : 1;
    float sqrt_total = std::sqrt((float)pv);
    float puct_star = q_norm[best_idx] + c_puct * prior[best_idx] * sqrt_total / (1.0f + visits[best_idx]);
    for (int i = 0; i < n; ++i) {
        if (i == best_idx || q_norm[i] >= puct_star) {
            result[i] = visits[i];
        } else {
            float denom = puct_star - q_norm[i];
            if (denom <= 0.0f) {
                result[i] = visits[i];
            } else {
                float n_min = c_puct * prior[i] * sqrt_total / denom - 1.0f;
                if (!(n_min > 0.0f)) n_min = 0.0f;  // f32::max(0.0)
                result[i] = visits[i] < n_min ? visits[i] : n_min;  // f32::min
            }
        }
    }
}

// search.rs:1116-1177
inline void extract_half(const HalfNode& half, float node_value, float value_scale, uint32_t children_visits,
                         const SearchConfig& config, float policy[5], float visit_counts[5], float& value) {
    int n = half.n_outcomes;
    for (int i = 0; i < 5; ++i) policy[i] = visit_counts[i] = 0.0f;
    if (n == 0) {
        value = node_value;
        return;
    }
    float fpu = compute_fpu(half, node_value, value_scale, config.fpu_reduction);
    float q[5] = {0, 0, 0, 0, 0}, raw[5] = {0, 0, 0, 0, 0}, prior[5] = {0, 0, 0, 0, 0}, qn[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const HalfEdge& e = half.edges[i];
        q[i] = e.visits > 0 ? e.q : fpu;
        raw[i] = (float)e.visits;
        prior[i] = half.prior[i];
        qn[i] = q[i] / value_scale;
    }
    float pruned[5];
    compute_pruned_visits(qn, prior, raw, n, children_visits, config.c_puct, pruned);
    for (int i = 0; i < n; ++i) visit_counts[half.outcomes[i]] = pruned[i];
    float sum = 0.0f;  // iter().sum::<f32>() starts from 0.0 and adds in order
    for (int i = 0; i < 5; ++i) {
        policy[i] = visit_counts[i];
        sum += policy[i];
    }
    if (sum > 0.0f) {
        for (int i = 0; i < 5; ++i) policy[i] /= sum;
    } else {
        half.expand_prior(policy);
    }
    float visit_sum = 0.0f;
    for (int i = 0; i < n; ++i) visit_sum += raw[i];
    if (visit_sum > 0.0f) {
        float dot = 0.0f;
        for (int i = 0; i < n; ++i) dot += q[i] * raw[i];
        value = dot / visit_sum;
    } else {
        value = node_value;
    }
}

// search.rs:1079-1111
inline SearchResult extract_result(const Node* root, const SearchConfig& config) {
    SearchResult r;
    r.total_visits = root->total_visits;
    uint32_t cv = root->children_visits();
    extract_half(root->p1, root->v1, root->value_scale, cv, config, r.policy_p1, r.visit_counts_p1, r.value_p1);
    extract_half(root->p2, root->v2, root->value_scale, cv, config, r.policy_p2, r.visit_counts_p2, r.value_p2);
    root->p1.expand_prior(r.prior_p1);
    root->p2.expand_prior(r.prior_p2);
    r.nn_evals = r.terminals = r.collisions = 0;
    return r;
}

// search.rs:362-390
inline bool run_search(MCTSTree& tree, const GameState& game, const Backend& backend, const SearchConfig& config,
                       uint32_t n_sims, uint32_t batch_size, SmallRng& rng, SearchResult& result,
                       std::string& err, SearchCounters* ctr = nullptr) {
    uint32_t remaining = n_sims;
    uint32_t nn = 0, term = 0, coll = 0;
    while (remaining > 0) {
        BatchStats b;
        uint32_t bs = remaining < batch_size ? remaining : batch_size;
        if (!simulate_batch(tree, game, backend, config, bs, rng, b, err, ctr)) return false;
        nn += b.nn_evals;
        term += b.terminals;
        coll += b.collisions;
        uint32_t produced = b.nn_evals + b.terminals;
        if (produced < 1) produced = 1;
        remaining = remaining > produced ? remaining - produced : 0;
    }
    result = extract_result(tree.root, config);
    result.nn_evals = nn;
    result.terminals = term;
    result.collisions = coll;
    return true;
}

}  // namespace oracle
