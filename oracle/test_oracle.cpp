// TEST INFRASTRUCTURE ONLY -- pins the CPU oracle against the exact known answers the
// reference's own unit tests hold for this path. Each check names the
// reference test it restates. Run by tests/test_oracle_known_answers.py; exit code 0 = all pass.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "mcts.hpp"
#include "selfplay.hpp"

using namespace oracle;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        ++g_checks;                                                          \
        if (!(cond)) {                                                       \
            ++g_fail;                                                        \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);      \
        }                                                                    \
    } while (0)
#define NEAR(a, b, tol) CHECK(std::fabs((double)(a) - (double)(b)) < (tol))

static HalfNode half_new(const float prior5[5], const uint8_t eff[5]) {
    HalfNode h = HalfNode::new_shell(eff);
    h.set_prior(prior5);
    return h;
}
static Node* open_node(const float p1[5], const float p2[5]) {
    static const uint8_t open[5] = {0, 1, 2, 3, 4};
    Node* n = new Node();
    n->p1 = half_new(p1, open);
    n->p2 = half_new(p2, open);
    return n;
}
static GameState open5(int x1, int y1, int x2, int y2, std::initializer_list<std::pair<int, int>> cheese,
                       uint16_t max_turns = 100) {
    GameState g;
    g.init_open(5, 5, max_turns);
    g.player1.x = x1;
    g.player1.y = y1;
    g.player2.x = x2;
    g.player2.y = y2;
    for (auto c : cheese) g.add_cheese(c.first, c.second);
    return g;
}

static void test_node() {
    // node.rs:496-543 outcomes_*
    uint8_t out[5], a2i[5], n;
    const uint8_t e_open[5] = {0, 1, 2, 3, 4}, e_wall[5] = {4, 1, 2, 3, 4}, e_corner[5] = {4, 1, 2, 4, 4},
                  e_mud[5] = {4, 4, 4, 4, 4};
    compute_outcomes(e_open, out, n, a2i);
    CHECK(n == 5);
    for (int a = 0; a < 5; ++a) CHECK(out[a] == a && a2i[a] == a);
    compute_outcomes(e_wall, out, n, a2i);
    CHECK(n == 4 && out[0] == 1 && out[1] == 2 && out[2] == 3 && out[3] == 4);
    CHECK(a2i[0] == a2i[4] && out[a2i[0]] == 4);
    compute_outcomes(e_corner, out, n, a2i);
    CHECK(n == 3 && out[0] == 1 && out[1] == 2 && out[2] == 4 && a2i[0] == a2i[3] && a2i[0] == a2i[4]);
    compute_outcomes(e_mud, out, n, a2i);
    CHECK(n == 1 && out[0] == 4);
    for (int a = 0; a < 5; ++a) CHECK(a2i[a] == 0);

    // node.rs:548-586 prior_reduction_*
    const float u[5] = {0.2f, 0.2f, 0.2f, 0.2f, 0.2f};
    HalfNode h = half_new(u, e_wall);
    CHECK(h.n_outcomes == 4);
    NEAR(h.prior[h.action_to_idx[4]], 0.4, 1e-6);
    const float nu[5] = {0.1f, 0.3f, 0.2f, 0.15f, 0.25f};
    h = half_new(nu, e_wall);
    NEAR(h.prior[h.action_to_idx[4]], 0.35, 1e-6);
    float tot = 0;
    for (int i = 0; i < h.n_outcomes; ++i) tot += h.prior[i];
    NEAR(tot, 1.0, 1e-6);
    const float nm[5] = {0.1f, 0.2f, 0.3f, 0.15f, 0.25f};
    h = half_new(nm, e_mud);
    CHECK(h.n_outcomes == 1);
    NEAR(h.prior[0], 1.0, 1e-6);
    // expand_prior round trip (node.rs expand_prior_one_wall)
    h = half_new(nu, e_wall);
    float ex[5];
    h.expand_prior(ex);
    CHECK(ex[0] == 0.0f);
    NEAR(ex[4], 0.35, 1e-6);
    NEAR(ex[1], 0.3, 1e-6);

    // node.rs:986-1071 multivisit Welford equivalence
    Node a, b;
    a.n_in_flight = 5;
    a.finalize_score_update(3.0f, 5.0f, 5);
    b.n_in_flight = 5;
    for (int i = 0; i < 5; ++i) b.finalize_score_update(3.0f, 5.0f, 1);
    CHECK(a.total_visits == 5 && b.total_visits == 5 && a.n_in_flight == 0);
    NEAR(a.v1, b.v1, 1e-6);
    NEAR(a.v2, b.v2, 1e-6);
    HalfEdge em, es;
    em.update_multivisit(7.0f, 4);
    for (int i = 0; i < 4; ++i) es.update_multivisit(7.0f, 1);
    CHECK(em.visits == es.visits);
    NEAR(em.q, es.q, 1e-6);
    Node c;
    c.n_in_flight = 5;
    c.finalize_score_update(2.0f, 2.0f, 3);
    c.finalize_score_update(8.0f, 8.0f, 2);
    NEAR(c.v1, 4.4, 1e-5);
    // node.rs edge_welford_sequence / marginal q
    HalfEdge e;
    e.update_multivisit(10.0f, 1);
    e.update_multivisit(20.0f, 1);
    e.update_multivisit(30.0f, 1);
    CHECK(e.visits == 3);
    NEAR(e.q, 20.0, 1e-5);
    HalfEdge e2;
    e2.update_multivisit(5.0f, 1);
    e2.update_multivisit(4.5f, 1);
    e2.update_multivisit(5.0f, 1);
    NEAR(e2.q, 14.5 / 3.0, 1e-5);
    // node.rs try_start_score_update semantics (:388-394)
    Node t;
    CHECK(t.try_start_score_update());
    CHECK(!t.try_start_score_update());
    t.finalize_score_update(0, 0, 1);
    CHECK(t.try_start_score_update() && t.try_start_score_update());
}

static void test_tree() {
    // tree.rs:432-461 smart_uniform_*
    float p[5];
    const uint8_t e_open[5] = {0, 1, 2, 3, 4}, e_wall[5] = {4, 1, 2, 3, 4}, e_mud[5] = {4, 4, 4, 4, 4};
    smart_uniform_prior(e_open, p);
    for (int i = 0; i < 5; ++i) NEAR(p[i], 0.2, 1e-6);
    smart_uniform_prior(e_wall, p);
    CHECK(p[0] == 0.0f);
    for (int i = 1; i < 5; ++i) NEAR(p[i], 0.25, 1e-6);
    smart_uniform_prior(e_mud, p);
    for (int i = 0; i < 4; ++i) CHECK(p[i] == 0.0f);
    NEAR(p[4], 1.0, 1e-6);

    // tree.rs:934-999 compute_rewards_*
    {
        GameState g = open5(0, 0, 4, 4, {{1, 0}});
        g.make_move(RIGHT, STAY);
        NEAR(g.player1.score, 1.0, 1e-6);
        NEAR(g.player2.score, 0.0, 1e-6);
        GameState g2 = open5(0, 0, 4, 4, {{1, 0}});
        g2.make_move(UP, STAY);
        NEAR(g2.player1.score, 0.0, 1e-6);
        GameState g3 = open5(0, 0, 2, 0, {{1, 0}});
        g3.make_move(RIGHT, LEFT);
        NEAR(g3.player1.score, 0.5, 1e-6);
        NEAR(g3.player2.score, 0.5, 1e-6);
        GameState g4 = open5(0, 0, 4, 0, {{1, 0}, {3, 0}});
        g4.make_move(RIGHT, LEFT);
        NEAR(g4.player1.score, 1.0, 1e-6);
        NEAR(g4.player2.score, 1.0, 1e-6);
        GameState g5 = open5(0, 0, 2, 0, {{1, 0}});
        g5.make_move(STAY, LEFT);
        NEAR(g5.player1.score, 0.0, 1e-6);
        NEAR(g5.player2.score, 1.0, 1e-6);
    }
    // tree.rs root init: value_scale = max(remaining, 1), priors smart-uniform, corner outcomes
    {
        GameState g = open5(0, 0, 4, 4, {{0, 0}, {1, 1}, {3, 3}});
        MCTSTree tr(g);
        NEAR(tr.root->value_scale, 3.0, 1e-6);
        CHECK(tr.root->p1.n_outcomes == 3 && tr.root->p2.n_outcomes == 3 && tr.node_count == 1);
        CHECK(tr.root->total_visits == 0);
        NEAR(tr.root->p1.prior[0], 1.0 / 3.0, 1e-6);
    }
    // effective actions: test_util.rs mud_game_p1_stuck, backend.rs:232-256
    {
        GameState g;
        g.init_open(5, 5, 100);
        g.player1.x = 2;
        g.player1.y = 2;
        g.add_mud(2, 2, 2, 3, 3);
        g.add_cheese(0, 0);
        g.make_move(UP, STAY);
        CHECK(g.player1.mud_timer > 0);
        CHECK(g.player1.x == 2 && g.player1.y == 3);
        uint8_t e1[5];
        g.effective_actions_p1(e1);
        for (int i = 0; i < 5; ++i) CHECK(e1[i] == 4);
    }
    // undo restores exactly (search relies on it, search.rs:706,723,732)
    {
        GameState g = open5(0, 0, 2, 0, {{1, 0}, {3, 3}});
        GameState before = g;
        MoveUndo u = g.make_move(RIGHT, LEFT);
        CHECK(g.remaining_cheese == 1 && g.turn == 1);
        g.unmake_move(u);
        CHECK(g.turn == 0 && g.remaining_cheese == 2 && g.player1.x == 0 && g.player2.x == 2);
        CHECK(std::memcmp(g.cheese_bits, before.cheese_bits, sizeof g.cheese_bits) == 0);
        CHECK(g.player1.score == 0.0f && g.player2.score == 0.0f);
    }
    // test_util.rs:105-118 terminal_game: turn >= max_turns => over
    {
        GameState g = open5(0, 0, 0, 1, {{4, 4}}, 1);
        CHECK(!g.check_game_over());
        g.make_move(STAY, STAY);
        CHECK(g.check_game_over());
    }
}

static void test_search_units() {
    SearchConfig cfg;
    // search.rs:1773-1875 prune_*
    {
        float q[5] = {0.5f, 0.3f, 0.8f, 0.2f, 0.1f}, pr[5] = {0.2f, 0.2f, 0.2f, 0.2f, 0.2f},
              v[5] = {10, 5, 20, 3, 2}, r[5];
        compute_pruned_visits(q, pr, v, 5, 40, 1.5f, r);
        NEAR(r[2], 20.0, 1e-6);
        for (int i = 0; i < 5; ++i) CHECK(r[i] <= v[i] + 1e-6f && r[i] >= 0.0f);
        float q2[5] = {0.5f, 0.3f, 0.8f, 0.95f, 0.1f}, v2[5] = {10, 5, 20, 18, 2};
        compute_pruned_visits(q2, pr, v2, 5, 55, 1.5f, r);
        NEAR(r[3], 18.0, 1e-6);
        float q3[5] = {0.8f, 0.1f, 0, 0, 0}, v3[5] = {50, 20, 10, 10, 10};
        compute_pruned_visits(q3, pr, v3, 5, 100, 1.5f, r);
        NEAR(r[0], 50.0, 1e-6);
        for (int i = 2; i < 5; ++i) CHECK(r[i] <= v3[i]);
        // exact cap: n_min = c*p*sqrt(N)/(puct*-q) - 1 with puct* = 0.8 + 1.5*0.2*10/51
        float ps = 0.8f + 1.5f * 0.2f * 10.0f / 51.0f;
        float nmin = 1.5f * 0.2f * 10.0f / (ps - 0.0f) - 1.0f;
        NEAR(r[2], nmin < 10.0f ? nmin : 10.0f, 1e-6);
        float q4[5] = {0.9f, 0, 0, 0, 0}, v4[5] = {50, 1, 1, 1, 1};
        compute_pruned_visits(q4, pr, v4, 5, 54, 1.5f, r);
        for (int i = 0; i < 5; ++i) CHECK(r[i] >= 0.0f);
        float q5[1] = {0.5f}, p5[1] = {1.0f}, v5[1] = {42.0f};
        compute_pruned_visits(q5, p5, v5, 1, 42, 1.5f, r);
        NEAR(r[0], 42.0, 1e-6);
    }
    const float u[5] = {0.2f, 0.2f, 0.2f, 0.2f, 0.2f};
    // search.rs:3193-3233 backup_finalize_single_level
    {
        Node* root = open_node(u, u);
        Node* child = open_node(u, u);
        child->parent = root;
        child->po1 = 0;
        child->po2 = 1;
        child->edge_r1 = 1.0f;
        child->edge_r2 = 0.5f;
        child->value_scale = root->value_scale = 5.0f;
        root->first_child = child;
        root->n_in_flight = 1;
        child->n_in_flight = 1;
        root->p1.edges[0].n_in_flight = 1;
        root->p2.edges[1].n_in_flight = 1;
        backup_and_finalize(child, 3.0f, 2.0f, 1, nullptr);
        CHECK(child->total_visits == 1 && root->total_visits == 1);
        NEAR(child->v1, 3.0, 1e-6);
        NEAR(child->v2, 2.0, 1e-6);
        NEAR(root->v1, 4.0, 1e-6);
        NEAR(root->v2, 2.5, 1e-6);
        CHECK(child->n_in_flight == 0 && root->n_in_flight == 0);
        CHECK(root->p1.edges[0].n_in_flight == 0 && root->p2.edges[1].n_in_flight == 0);
        CHECK(root->p1.edges[0].visits == 1 && root->p2.edges[1].visits == 1);
        NEAR(root->p1.edges[0].q, 4.0, 1e-6);
        free_subtree(root);
    }
    // search.rs:3235-3276 backup_finalize_two_level
    {
        Node* root = open_node(u, u);
        Node* mid = open_node(u, u);
        Node* leaf = open_node(u, u);
        leaf->parent = mid;
        leaf->po1 = 1;
        leaf->po2 = 2;
        leaf->edge_r1 = 0.5f;
        leaf->edge_r2 = 1.0f;
        mid->parent = root;
        mid->edge_r1 = 1.0f;
        mid->edge_r2 = 0.5f;
        mid->first_child = leaf;
        root->first_child = mid;
        root->n_in_flight = mid->n_in_flight = leaf->n_in_flight = 1;
        root->p1.edges[0].n_in_flight = root->p2.edges[0].n_in_flight = 1;
        mid->p1.edges[1].n_in_flight = mid->p2.edges[2].n_in_flight = 1;
        backup_and_finalize(leaf, 2.0f, 3.0f, 1, nullptr);
        NEAR(leaf->v1, 2.0, 1e-6);
        NEAR(mid->v1, 2.5, 1e-6);
        NEAR(root->v1, 3.5, 1e-6);
        NEAR(mid->v2, 4.0, 1e-6);
        NEAR(root->v2, 4.5, 1e-6);
        CHECK(leaf->n_in_flight == 0 && mid->n_in_flight == 0 && root->n_in_flight == 0);
        CHECK(root->p1.edges[0].n_in_flight == 0 && mid->p1.edges[1].n_in_flight == 0);
        free_subtree(root);
    }
    // search.rs:3278-3311 backup_finalize_multivisit
    {
        Node* root = open_node(u, u);
        Node* child = open_node(u, u);
        child->parent = root;
        root->first_child = child;
        root->n_in_flight = child->n_in_flight = 3;
        root->p1.edges[0].n_in_flight = root->p2.edges[0].n_in_flight = 3;
        backup_and_finalize(child, 4.0f, 2.0f, 3, nullptr);
        CHECK(child->total_visits == 3 && root->total_visits == 3);
        NEAR(child->v1, 4.0, 1e-6);
        NEAR(root->v1, 4.0, 1e-6);
        CHECK(child->n_in_flight == 0 && root->n_in_flight == 0 && root->p1.edges[0].visits == 3);
        free_subtree(root);
    }
    // search.rs:3403-3477 collision_cancel_*
    {
        Node* root = open_node(u, u);
        Node* child = open_node(u, u);
        child->parent = root;
        child->po1 = 2;
        child->po2 = 3;
        root->first_child = child;
        root->n_in_flight = 5;
        root->p1.edges[2].n_in_flight = root->p2.edges[3].n_in_flight = 5;
        cancel_shared_collisions({Collision{child, 5}}, root);
        CHECK(root->n_in_flight == 0 && root->p1.edges[2].n_in_flight == 0 && root->p2.edges[3].n_in_flight == 0);
        root->n_in_flight = 5;
        root->p1.edges[2].n_in_flight = root->p2.edges[3].n_in_flight = 5;
        cancel_shared_collisions({Collision{child, 2}}, root);
        CHECK(root->n_in_flight == 3 && root->p1.edges[2].n_in_flight == 3 && root->p2.edges[3].n_in_flight == 3);
        free_subtree(root);
        Node* r2 = open_node(u, u);
        Node* mid = open_node(u, u);
        Node* leaf = open_node(u, u);
        leaf->parent = mid;
        leaf->po1 = 1;
        leaf->po2 = 0;
        mid->parent = r2;
        mid->first_child = leaf;
        r2->first_child = mid;
        r2->n_in_flight = mid->n_in_flight = 3;
        r2->p1.edges[0].n_in_flight = r2->p2.edges[0].n_in_flight = 3;
        mid->p1.edges[1].n_in_flight = mid->p2.edges[0].n_in_flight = 3;
        cancel_shared_collisions({Collision{leaf, 3}}, r2);
        CHECK(r2->n_in_flight == 0 && mid->n_in_flight == 0 && r2->p1.edges[0].n_in_flight == 0 &&
              mid->p1.edges[1].n_in_flight == 0);
        free_subtree(r2);
    }
    // search.rs:3578-3655 collisions_left_*
    {
        SearchConfig c;
        c.collision_limit_min = 2;
        c.collision_limit_max = 128;
        CHECK(calculate_collisions_left(0, c) == 2 && calculate_collisions_left(799, c) == 2 &&
              calculate_collisions_left(800, c) == 2);
        CHECK(calculate_collisions_left(50000, c) == 128 && calculate_collisions_left(100000, c) == 128);
        SearchConfig l;
        l.collision_limit_min = 0;
        l.collision_limit_max = 100;
        l.collision_scaling_start = 0;
        l.collision_scaling_end = 100;
        CHECK(calculate_collisions_left(50, l) == 50 && calculate_collisions_left(25, l) == 25);
        l.collision_scaling_power = 2.0f;
        CHECK(calculate_collisions_left(50, l) == 25 && calculate_collisions_left(100, l) == 100);
        SearchConfig q;
        q.collision_limit_min = 5;
        q.collision_limit_max = 200;
        q.collision_scaling_start = q.collision_scaling_end = 1000;
        CHECK(calculate_collisions_left(999, q) == 5 && calculate_collisions_left(1000, q) == 200 &&
              calculate_collisions_left(1001, q) == 200);
        // defaults give 1 below 800 nodes and ~7 at 1897 nodes
        SearchConfig d;
        CHECK(calculate_collisions_left(1897, d) == 7);
    }
    // search.rs:3372-3399 estimated_vtcb_*
    {
        const uint8_t stuck[5] = {4, 4, 4, 4, 4}, open[5] = {0, 1, 2, 3, 4};
        HalfNode hs = half_new(u, stuck);
        uint32_t ns[5] = {0, 0, 0, 0, 0};
        SmallRng r = SmallRng::seed_from_u64(42);
        uint8_t best;
        uint32_t vt;
        estimated_visits_to_change_best_half(hs, 2.0f, 5.0f, 1, cfg, false, ns, r, best, vt);
        CHECK(best == 0 && vt == UINT32_MAX);
        HalfNode ho = half_new(u, open);
        estimated_visits_to_change_best_half(ho, 2.0f, 5.0f, 0, cfg, false, ns, r, best, vt);
        CHECK(vt == 1);
    }
    // search.rs:3316-3370 pick_distribute_*
    {
        Node* n = open_node(u, u);
        n->value_scale = 5.0f;
        n->n_in_flight = 1;
        n->finalize_score_update(2.0f, 2.0f, 1);
        SmallRng r = SmallRng::seed_from_u64(42);
        GatherLevel lvl = build_gather_level(n, 10, cfg, false, r);
        uint32_t tot = 0, nz = 0, vl1 = 0, vl2 = 0;
        for (int i = 0; i < 25; ++i) {
            tot += lvl.vtp[i];
            nz += lvl.vtp[i] > 0;
        }
        for (int i = 0; i < 5; ++i) {
            vl1 += n->p1.edges[i].n_in_flight;
            vl2 += n->p2.edges[i].n_in_flight;
        }
        CHECK(tot == 10 && nz > 1 && vl1 == 10 && vl2 == 10);
        delete n;
        const float dom[5] = {0.8f, 0.05f, 0.05f, 0.05f, 0.05f};
        Node* m = open_node(dom, u);
        m->value_scale = 5.0f;
        m->n_in_flight = 1;
        m->finalize_score_update(2.0f, 2.0f, 1);
        GatherLevel l2 = build_gather_level(m, 20, cfg, false, r);
        uint32_t p1v0 = 0;
        for (int a2 = 0; a2 < 5; ++a2) p1v0 += l2.vtp[a2];
        CHECK(p1v0 > 10);
        delete m;
    }
}

static void walk(const Node* n, const std::function<void(const Node*)>& f) {
    f(n);
    for (const Node* c = n->first_child; c; c = c->next_sibling) walk(c, f);
}

static void test_search_invariants() {
    SearchConfig cfg;
    Backend be = smart_uniform_backend();
    std::string err;
    auto run = [&](GameState g, uint32_t sims, uint32_t batch, SearchResult& r, MCTSTree& tr) {
        SmallRng rng = SmallRng::seed_from_u64(123);
        bool ok = run_search(tr, g, be, cfg, sims, batch, rng, r, err);
        CHECK(ok);
    };
    // search.rs:2371-2390 search_root_evaluation
    {
        GameState g = open5(1, 1, 3, 3, {{2, 2}, {3, 3}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, 1, 1, r, tr);
        CHECK(r.total_visits == 1);
        NEAR(r.value_p1, 0, 1e-6);
        NEAR(r.value_p2, 0, 1e-6);
    }
    // search.rs:2394-2423 search_first_expansion
    {
        GameState g = open5(1, 1, 3, 3, {{2, 2}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, 2, 1, r, tr);
        CHECK(r.total_visits == 2 && tr.root->first_child != nullptr);
        bool found = false;
        for (Node* c = tr.root->first_child; c; c = c->next_sibling) found = found || c->total_visits == 1;
        CHECK(found);
    }
    // search.rs:2438-2484 search_invariants_after_50_sims, :2751 n_in_flight zero
    {
        GameState g = open5(2, 2, 2, 2, {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, 50, 1, r, tr);
        CHECK(r.total_visits == 50);
        walk(tr.root, [&](const Node* n) {
            CHECK(n->n_in_flight == 0);
            uint32_t s = 0;
            for (int j = 0; j < n->p1.n_outcomes; ++j) {
                s += n->p1.edges[j].visits;
                CHECK(std::isfinite(n->p1.edges[j].q) && n->p1.edges[j].n_in_flight == 0);
            }
            if (n->total_visits > 0 && n->first_child) CHECK(s == n->total_visits - 1);
        });
    }
    // search.rs:2488-2508 search_corridor
    {
        GameState g;
        g.init_open(5, 5, 100);
        for (int x = 0; x < 5; ++x) g.add_wall(x, 0, x, 1);
        g.player2.x = 4;
        g.player2.y = 0;
        g.add_cheese(2, 0);
        MCTSTree tr(g);
        SearchResult r;
        run(g, 50, 4, r, tr);
        CHECK(r.policy_p1[0] == 0.0f && r.policy_p1[2] == 0.0f && r.policy_p1[1] > 0.3f);
    }
    // search.rs:2512-2528 search_adjacent_cheese
    {
        GameState g = open5(0, 0, 4, 4, {{1, 0}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, 100, 4, r, tr);
        CHECK(r.policy_p1[1] > 0.5f);
    }
    // search.rs:2532-2552 search_terminal_mid_tree, :3089 batch_short_game_completes
    for (uint32_t batch : {1u, 8u}) {
        GameState g = open5(0, 0, 2, 0, {{1, 0}}, 3);
        MCTSTree tr(g);
        SearchResult r;
        run(g, 50, batch, r, tr);
        if (batch == 1) CHECK(r.total_visits >= 50);
        CHECK(r.total_visits > 0);
        walk(tr.root, [&](const Node* n) {
            if (n->is_terminal) CHECK(n->first_child == nullptr);
            CHECK(n->n_in_flight == 0);
        });
    }
    // search.rs:2556-2567 search_terminal_root; :3691 terminal_root_exact_accounting
    {
        GameState g = open5(0, 0, 0, 1, {{4, 4}}, 1);
        g.make_move(STAY, STAY);
        MCTSTree tr(g);
        SearchResult r;
        run(g, 10, 4, r, tr);
        NEAR(r.value_p1, 0, 1e-6);
        NEAR(r.value_p2, 0, 1e-6);
        CHECK(tr.root->is_terminal && tr.root->n_in_flight == 0);
    }
    // search.rs:2571-2588 search_mud_position
    {
        GameState g;
        g.init_open(5, 5, 100);
        g.player1.x = 2;
        g.player1.y = 2;
        g.add_mud(2, 2, 2, 3, 3);
        g.add_cheese(0, 0);
        g.make_move(UP, STAY);
        MCTSTree tr(g);
        SearchResult r;
        run(g, 20, 4, r, tr);
        NEAR(r.policy_p1[4], 1.0, 1e-6);
        for (int a = 0; a < 4; ++a) CHECK(r.policy_p1[a] == 0.0f);
    }
    // search.rs:2625-2640 search_blocked_actions_zero, :2592 policy sums
    {
        GameState g = open5(0, 0, 4, 4, {{2, 2}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, 30, 4, r, tr);
        CHECK(r.policy_p1[2] == 0.0f && r.policy_p1[3] == 0.0f);
        float s1 = 0, s2 = 0;
        for (int i = 0; i < 5; ++i) {
            s1 += r.policy_p1[i];
            s2 += r.policy_p2[i];
        }
        NEAR(s1, 1.0, 1e-5);
        NEAR(s2, 1.0, 1e-5);
    }
    // search.rs:3113-3135 batch_unvisited_root_one_eval; :3138 ooo_no_change_batch_size_1
    {
        GameState g = open5(1, 1, 3, 3, {{2, 2}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, 1, 10, r, tr);
        CHECK(r.total_visits == 1);
        GameState g2 = open5(0, 0, 4, 4, {{2, 2}});
        MCTSTree tr2(g2);
        run(g2, 20, 1, r, tr2);
        CHECK(r.total_visits == 20);
    }
    // test_search.py:131-135 total_visits == n_sims for {10,50,100,200} at batch 8 (fresh tree:
    // below 800 nodes the collision budget is 1, so a batch never loses a visit)
    for (uint32_t sims : {10u, 50u, 100u, 200u}) {
        GameState g = open5(0, 0, 4, 4, {{2, 2}, {1, 3}, {3, 1}});
        MCTSTree tr(g);
        SearchResult r;
        run(g, sims, 8, r, tr);
        CHECK(r.total_visits == sims);
    }
    // search.rs:2822-2852 nonzero backend root value; :3722 backend_error_cleanup_warm_tree
    {
        GameState g = open5(1, 1, 3, 3, {{2, 2}, {0, 4}});
        MCTSTree tr(g);
        SearchResult r;
        SmallRng rng = SmallRng::seed_from_u64(123);
        CHECK(run_search(tr, g, smart_uniform_backend(1.5f, 0.5f), cfg, 1, 1, rng, r, err));
        NEAR(r.value_p1, 1.5, 1e-6);
        NEAR(r.value_p2, 0.5, 1e-6);
        CHECK(run_search(tr, g, smart_uniform_backend(1.5f, 0.5f), cfg, 40, 4, rng, r, err));
        Backend failing = [](const std::vector<const GameState*>&, std::vector<EvalResult>&, std::string& e) {
            e = "test failure";
            return false;
        };
        CHECK(!run_search(tr, g, failing, cfg, 8, 4, rng, r, err));
        walk(tr.root, [&](const Node* n) {
            CHECK(n->n_in_flight == 0);
            for (int j = 0; j < 5; ++j) CHECK(n->p1.edges[j].n_in_flight == 0 && n->p2.edges[j].n_in_flight == 0);
        });
    }
    // search.rs:2956-3086 noise_*: disabled leaves priors alone; enabled modifies, sums to 1,
    // deterministic under a seed; single outcome no-op
    {
        GameState g = open5(2, 2, 1, 1, {{0, 0}, {4, 4}});
        SearchConfig nz = cfg;
        nz.noise_epsilon = 0.25f;
        float pri[2][5];
        for (int rep = 0; rep < 2; ++rep) {
            MCTSTree tr(g);
            SearchResult r;
            SmallRng rng = SmallRng::seed_from_u64(7);
            CHECK(run_search(tr, g, be, nz, 20, 4, rng, r, err));
            float s = 0;
            bool moved = false;
            for (int i = 0; i < 5; ++i) {
                pri[rep][i] = r.prior_p1[i];
                s += r.prior_p1[i];
                moved = moved || std::fabs(r.prior_p1[i] - 0.2f) > 1e-4f;
            }
            NEAR(s, 1.0, 1e-5);
            CHECK(moved);
        }
        for (int i = 0; i < 5; ++i) CHECK(pri[0][i] == pri[1][i]);
        const uint8_t stuck[5] = {4, 4, 4, 4, 4};
        const float u[5] = {0.2f, 0.2f, 0.2f, 0.2f, 0.2f};
        HalfNode h = half_new(u, stuck);
        SmallRng rng = SmallRng::seed_from_u64(7);
        uint64_t s0 = rng.s[0];
        CHECK(apply_dirichlet_noise(h, 0.25f, 10.83f, rng));
        NEAR(h.prior[0], 1.0, 1e-6);
        CHECK(rng.s[0] == s0);
    }
}

static void test_selfplay() {
    // selfplay.rs:878-931 maze array; :955-1026 play_game invariants
    GameState g;
    g.init_open(5, 5, 30);
    CHECK(make_cheese(g, 5, true, 3));
    CHECK(g.remaining_cheese == 5 && g.has_cheese(12));
    for (int i = 0; i < 25; ++i) CHECK(g.has_cheese(i) == g.has_cheese(24 - i));
    CHECK(!g.has_cheese(0) && !g.has_cheese(24));
    std::vector<int8_t> mz = build_maze_array(g);
    CHECK(mz[0] == 1 && mz[1] == 1 && mz[2] == -1 && mz[3] == -1);
    CHECK(mz[24 * 4 + 0] == -1 && mz[24 * 4 + 1] == -1 && mz[24 * 4 + 2] == 1 && mz[24 * 4 + 3] == 1);
    SearchConfig cfg;
    GameRecord rec;
    std::string err;
    SmallRng rng = SmallRng::seed_from_u64(99);
    CHECK(play_game(g, smart_uniform_backend(), cfg, 50, 8, rng, 7, rec, err));
    CHECK(!rec.positions.empty() && rec.positions.size() <= 30 && rec.game_index == 7);
    CHECK(rec.cheese_available == 5);
    float collected = rec.final_p1_score + rec.final_p2_score;
    int outc = 0;
    for (uint8_t o : rec.cheese_outcomes) outc += o != 2;
    CHECK((float)outc == collected);
    for (size_t i = 0; i < rec.positions.size(); ++i) {
        const PositionRecord& p = rec.positions[i];
        CHECK(p.turn == i);
        float s = 0;
        for (int a = 0; a < 5; ++a) s += p.policy_p1[a];
        NEAR(s, 1.0, 1e-5);
        CHECK(p.policy_p1[p.action_p1] > 0.0f && p.policy_p2[p.action_p2] > 0.0f);
    }
    CHECK(rec.result == (rec.final_p1_score > rec.final_p2_score ? 1 : rec.final_p2_score > rec.final_p1_score ? 2 : 0));
    // determinism under a seed
    GameRecord rec2;
    SmallRng rng2 = SmallRng::seed_from_u64(99);
    CHECK(play_game(g, smart_uniform_backend(), cfg, 50, 8, rng2, 7, rec2, err));
    CHECK(rec2.positions.size() == rec.positions.size() && rec2.total_simulations == rec.total_simulations);
    // sample_action: all-zero -> STAY (selfplay.rs:474-479)
    const float z[5] = {0, 0, 0, 0, 0};
    CHECK(sample_action(z, rng) == 4);
    const float one[5] = {0, 0, 1.0f, 0, 0};
    for (int i = 0; i < 20; ++i) CHECK(sample_action(one, rng) == 2);
}

int main() {
    test_node();
    test_tree();
    test_search_units();
    test_search_invariants();
    test_selfplay();
    std::printf("%d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
