// Host-side machinery of the shim that needs no GPU: the write-back's fork-join pool (every index visited exactly once, whatever the
// number of threads, with late and early workers), the on_destroy hooks that replace the per-update scan for destroyed bodies /
// constraints (island_manager.cpp:24-27 is the reference's form of the same), the host form of update_presentation / snap_presentation.
#include <edyn/edyn.hpp>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
#define REQUIRE_VOID(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

int main() {
    // ---- worker_pool
    for (unsigned threads : {1u, 2u, 5u, 8u}) {
        edyn::detail::worker_pool pool(threads);
        REQUIRE(pool.size() == threads);
        std::vector<std::atomic<uint32_t>> hits(70000);
        uint64_t expect = 0;
        for (int round = 0; round < 400; ++round) {
            const uint32_t count = round % 7 == 0 ? 1u + (uint32_t)round : 1000u + 157u * (uint32_t)round, grain = round % 3 == 0 ? 64u : 512u;
            if (round % 5 == 0) pool.prewake();
            if (round % 50 == 49) std::this_thread::sleep_for(std::chrono::milliseconds(5));   // the workers fall asleep in between
            std::atomic<uint32_t> max_worker{0};
            pool.parallel_for(count, grain, [&](uint32_t b, uint32_t e, unsigned w) {
                for (uint32_t i = b; i < e; ++i) hits[i].fetch_add(1, std::memory_order_relaxed);
                uint32_t m = max_worker.load(); while (w > m && !max_worker.compare_exchange_weak(m, w)) {}
            });
            REQUIRE(max_worker.load() < pool.size());
            expect += count;
            for (uint32_t i = 0; i < count; ++i) if (hits[i].load(std::memory_order_relaxed) == 0) { std::printf("index %u missed in round %d\n", i, round); return 1; }
        }
        uint64_t total = 0;
        for (auto &h : hits) total += h.load();
        REQUIRE(total == expect);
    }
    // ---- entity_table (the contact entities' look-up tables): open addressing with backward-shift deletion against std::unordered_map
    {
        edyn::detail::entity_table table;
        std::unordered_map<uint64_t, entt::entity> ref;
        uint64_t x = 88172645463325252ull;
        auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        for (int round = 0; round < 200000; ++round) {
            const uint64_t key = (rnd() % 5000) * 0x100000004ull + (rnd() % 3 == 0 ? 0 : (rnd() & 3));   // clustered keys, key 0 included
            const int op = (int)(rnd() % 3);
            if (op == 0) { const auto e = static_cast<entt::entity>((uint32_t)(rnd() % 100000)); table[key] = e; ref[key] = e; }
            else if (op == 1) {
                auto it = table.find(key); auto rt = ref.find(key);
                REQUIRE((it == table.end()) == (rt == ref.end()));
                if (rt != ref.end()) { REQUIRE(it->second == rt->second && edyn::detail::entity_table::key_of(*it) == key); table.erase(it); ref.erase(rt); }
            } else {
                auto it = table.find(key); auto rt = ref.find(key);
                REQUIRE((it == table.end()) == (rt == ref.end()));
                if (rt != ref.end()) REQUIRE(it->second == rt->second);
            }
            REQUIRE(table.size() == ref.size());
        }
        size_t seen = 0;
        for (auto &kv : table) { auto rt = ref.find(edyn::detail::entity_table::key_of(kv)); REQUIRE(rt != ref.end() && rt->second == kv.second); ++seen; }
        REQUIRE(seen == ref.size());
        table.clear();
        REQUIRE(table.size() == 0 && table.find(1) == table.end());
    }
    // ---- removal hooks: nothing is walked while nothing was destroyed
    entt::registry registry;
    auto cfg = edyn::init_config{};
    cfg.num_worker_threads = 3;
    edyn::attach(registry, cfg);
    auto &s = registry.ctx().get<edyn::detail::gpu_stepper>();
    auto def = edyn::rigidbody_def{};
    def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
    std::vector<entt::entity> bodies;
    for (int i = 0; i < 2000; ++i) { def.position = {1.5f * i, 1, 0}; def.angvel = {0.1f * (i % 7), 0.3f, 2.0f * (i % 3)}; def.linvel = {1, 0.5f * (i % 5), 0}; bodies.push_back(edyn::make_rigidbody(registry, def)); }
    auto con = edyn::make_constraint<edyn::point_constraint>(registry, bodies[0], bodies[1]);
    REQUIRE(s.hooks_connected);
    s.removal_pending = false;
    registry.get<edyn::position>(bodies[3]).x = 7;   // an edit is not a removal
    REQUIRE(!s.removal_pending);
    registry.destroy(bodies[5]);
    REQUIRE(s.removal_pending);
    edyn::detail::sync_removed(registry, s);   // (no device context yet: only the lists are updated)
    REQUIRE(!s.removal_pending && s.bodies[5] == entt::null && s.bodies[4] == bodies[4]);
    registry.destroy(con);
    REQUIRE(s.removal_pending);
    edyn::detail::sync_removed(registry, s);
    REQUIRE(s.constraints[0] == entt::null);
    s.removal_pending = false;
    edyn::clear_rigidbody(registry, bodies[9]);
    REQUIRE(s.removal_pending);
    edyn::detail::sync_removed(registry, s);
    REQUIRE(s.bodies[9] == entt::null && registry.valid(bodies[9]));
    // a joint whose body is destroyed goes with it
    auto con2 = edyn::make_constraint<edyn::hinge_constraint>(registry, bodies[20], bodies[21]);
    registry.destroy(bodies[21]);
    edyn::detail::sync_removed(registry, s);
    REQUIRE(s.constraints[1] == entt::null && !registry.valid(con2));

    // ---- registry-side helpers of the reference's util headers (no device needed: the stepper uploads lazily)
    {
        entt::entity b50 = bodies[50];
        REQUIRE(edyn::validate_rigidbody(registry, b50));
        entt::entity plain = registry.create();
        REQUIRE(!edyn::validate_rigidbody(registry, plain));
        auto c1 = edyn::make_constraint<edyn::point_constraint>(registry, bodies[50], bodies[51]);
        auto c2 = edyn::make_constraint<edyn::distance_constraint>(registry, bodies[50], bodies[52]);
        int edges = 0; std::vector<entt::entity> nb;
        edyn::visit_edges(registry, bodies[50], [&](entt::entity) { ++edges; });
        edyn::visit_neighbors(registry, bodies[50], [&](entt::entity o) { nb.push_back(o); });
        REQUIRE(edges == 2 && nb.size() == 2 && ((nb[0] == bodies[51] && nb[1] == bodies[52]) || (nb[0] == bodies[52] && nb[1] == bodies[51])));
        int first_only = 0;
        edyn::visit_edges(registry, bodies[50], [&](entt::entity) { ++first_only; return false; });   // a bool functor stops the walk
        REQUIRE(first_only == 1);
        s.removal_pending = false;
        edyn::clear_constraint(registry, c1);
        REQUIRE(s.removal_pending && registry.valid(c1) && !registry.all_of<edyn::point_constraint>(c1));
        edyn::detail::sync_removed(registry, s);
        edges = 0; edyn::visit_edges(registry, bodies[50], [&](entt::entity e) { ++edges; REQUIRE_VOID(e == c2); });
        REQUIRE(edges == 1);
        edyn::exclude_collision(registry, bodies[60], bodies[61]);
        edyn::exclude_collision(registry, edyn::entity_pair{bodies[60], bodies[62]});
        edyn::exclude_collision(registry, bodies[63], bodies[64]);
        REQUIRE(s.exclusions.size() == 3);
        edyn::clear_collision_exclusion(registry, bodies[60]);
        REQUIRE(s.exclusions.size() == 1 && s.exclusions[0][0] == registry.get<edyn::detail::body_index>(bodies[63]).value);
        REQUIRE(!edyn::manifold_exists(registry, bodies[60], bodies[61]) && edyn::get_manifold_entity(registry, edyn::entity_pair{bodies[60], bodies[61]}) == entt::null);
        // origin helpers on a body with a centre-of-mass offset
        auto od = edyn::rigidbody_def{};
        od.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
        od.position = {1, 2, 3};
        od.center_of_mass = edyn::vector3{0.25f, 0, 0};
        const auto ob = edyn::make_rigidbody(registry, od);
        const auto o0 = edyn::get_rigidbody_origin(registry, ob);
        REQUIRE(o0.x == 1 && o0.y == 2 && o0.z == 3 && registry.get<edyn::position>(ob).x == 1.25f);
        edyn::set_rigidbody_origin(registry, ob, {5, 2, 3});
        REQUIRE(registry.get<edyn::position>(ob).x == 5.25f && edyn::get_rigidbody_origin(registry, ob).x == 5 && s.state_dirty);
        registry.get<edyn::position>(ob).x = 7.25f;
        edyn::rigidbody_update_origin(registry, ob);
        REQUIRE(edyn::get_rigidbody_origin(registry, ob).x == 7);
        registry.get<edyn::present_position>(ob) = edyn::present_position{{9.25f, 2, 3}};
        REQUIRE(edyn::get_rigidbody_present_origin(registry, ob).x == 9);
    }
    // ---- host presentation: pos + v dt, integrate(orn, w, dt); sleeping bodies and bodies without present_* are left alone
    registry.emplace<edyn::sleeping_tag>(bodies[30]);
    registry.get<edyn::present_position>(bodies[30]).x = -99;
    const float dt = -0.004f;
    edyn::detail::update_presentation(registry, s, dt);
    bool ok = true;
    for (size_t i = 0; i < bodies.size(); ++i) {
        if (s.bodies[i] == entt::null) continue;
        const auto e = bodies[i];
        const auto &p = registry.get<edyn::position>(e); const auto &v = registry.get<edyn::linvel>(e);
        const auto &pp = registry.get<edyn::present_position>(e); const auto &po = registry.get<edyn::present_orientation>(e);
        if (i == 30) { ok = ok && pp.x == -99; continue; }
        ok = ok && pp.x == p.x + v.x * dt && pp.y == p.y + v.y * dt && pp.z == p.z + v.z * dt;
        const auto o = edyn::detail::integrate(registry.get<edyn::orientation>(e), registry.get<edyn::angvel>(e), dt);
        ok = ok && po.x == o.x && po.y == o.y && po.z == o.z && po.w == o.w && std::fabs(o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w - 1) < 1e-6f;
    }
    REQUIRE(ok);
    edyn::detail::snap_presentation(registry, s);
    for (size_t i = 0; i < bodies.size(); ++i) {
        if (s.bodies[i] == entt::null) continue;
        const auto &p = registry.get<edyn::position>(bodies[i]); const auto &pp = registry.get<edyn::present_position>(bodies[i]);
        ok = ok && pp.x == p.x && pp.y == p.y && pp.z == p.z;
    }
    REQUIRE(ok);
    edyn::detach(registry);
    registry.destroy(bodies[40]);   // the hooks are gone with the stepper: nothing dangles
    std::printf("HOST_LOGIC_OK\n");
    return 0;
}
