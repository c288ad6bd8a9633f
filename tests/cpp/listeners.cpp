// The shim over an EnTT-API registry (the `__has_include(<entt/entt.hpp>)` branch of include/edyn/edyn.hpp): built ONLY with
// -I oracle/entt_min (test-only use of the checker's EnTT subset - EnTT 3.15 itself is not in this image), never with the bundled
// mini registry. What an application of the reference does with signals keeps working:
//  * registry.on_construct<edyn::contact_point>() / on_destroy<...>() listeners see every contact point the device reports
//    (the reference's collision events: docs/Design.md:135-139, narrowphase.cpp:111-130, collision_util.cpp:311-388);
//  * on_construct / on_destroy<edyn::contact_manifold> likewise (constraint_util.cpp:60-102);
//  * registry.destroy(body) - the entity's own on_destroy hooks run at once, its manifolds and points follow at the next update
//    (island_manager.cpp:47-115 semantics);
//  * entt::scoped_connection releases a listener.
#include <edyn/edyn.hpp>
#include <cstdio>
#include <set>
#include <vector>

#if !__has_include(<entt/entt.hpp>)
#error "listeners.cpp exercises the shim's EnTT branch: build it with -I oracle/entt_min (make entt)"
#endif

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

struct observer {
    size_t points_made = 0, points_gone = 0, manifolds_made = 0, manifolds_gone = 0, bodies_gone = 0;
    std::set<entt::entity> live_points;
    bool point_had_its_data = true;
    void on_point(entt::registry &r, entt::entity e) {
        ++points_made;
        live_points.insert(e);
        point_had_its_data = point_had_its_data && r.all_of<edyn::contact_point>(e);
    }
    void off_point(entt::registry &, entt::entity e) { ++points_gone; live_points.erase(e); }
    void on_manifold(entt::registry &, entt::entity) { ++manifolds_made; }
    void off_manifold(entt::registry &, entt::entity) { ++manifolds_gone; }
    void off_body(entt::registry &, entt::entity) { ++bodies_gone; }
};

template <typename T> static size_t count(entt::registry &registry) {
    size_t n = 0;
    for (auto e : registry.view<T>()) { (void)e; ++n; }
    return n;
}

int main() {
    entt::registry registry;
    observer obs;
    registry.on_construct<edyn::contact_point>().connect<&observer::on_point>(obs);
    registry.on_destroy<edyn::contact_point>().connect<&observer::off_point>(obs);
    registry.on_construct<edyn::contact_manifold>().connect<&observer::on_manifold>(obs);
    registry.on_destroy<edyn::contact_manifold>().connect<&observer::off_manifold>(obs);
    registry.on_destroy<edyn::rigidbody_tag>().connect<&observer::off_body>(obs);

    edyn::attach(registry, edyn::init_config{});
    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    edyn::make_rigidbody(registry, floor_def);
    std::vector<entt::entity> boxes;
    for (int i = 0; i < 12; ++i) {
        auto def = edyn::rigidbody_def{};
        def.mass = 1;
        def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
        def.position = {1.5f * (i % 4), 0.7f + 1.2f * (i / 4), 0};   // three layers that fall onto each other
        def.sleeping_disabled = true;
        boxes.push_back(edyn::make_rigidbody(registry, def));
    }
    double t = 0;
    for (int i = 0; i < 120; ++i) { t += 1.0 / 60; edyn::update(registry, t); }

    // every point and manifold in the registry was announced, every vanished one retracted
    REQUIRE(obs.points_made > 12 * 4 - 1 && obs.manifolds_made >= 12);
    REQUIRE(obs.point_had_its_data);
    REQUIRE(obs.points_made - obs.points_gone == count<edyn::contact_point>(registry));
    REQUIRE(obs.manifolds_made - obs.manifolds_gone == count<edyn::contact_manifold>(registry));
    REQUIRE(obs.live_points.size() == count<edyn::contact_point>(registry));
    for (auto e : obs.live_points) REQUIRE(registry.valid(e) && registry.all_of<edyn::contact_point_list>(e));
    const auto device = edyn::get_contact_manifolds(registry);
    size_t device_points = 0;
    for (auto &m : device) device_points += m.num_points;
    REQUIRE(device.size() == count<edyn::contact_manifold>(registry) && device_points == count<edyn::contact_point>(registry));

    // registry.destroy(body): the body's own hook at once; its manifolds and points with the next update
    const size_t gone_before = obs.points_gone, mgone_before = obs.manifolds_gone;
    registry.destroy(boxes[8]);   // the top box of a column: one manifold (with the box below)
    REQUIRE(obs.bodies_gone == 1 && !registry.valid(boxes[8]));
    t += 1.0 / 60; edyn::update(registry, t);
    REQUIRE(obs.manifolds_gone > mgone_before && obs.points_gone > gone_before);
    REQUIRE(obs.points_made - obs.points_gone == count<edyn::contact_point>(registry));
    REQUIRE(obs.manifolds_made - obs.manifolds_gone == count<edyn::contact_manifold>(registry));

    // a scoped connection stops listening when it goes out of scope
    size_t scoped_hits = 0;
    struct counter { size_t *n; void hit(entt::registry &, entt::entity) { ++*n; } } ctr{&scoped_hits};
    {
        entt::scoped_connection conn = registry.on_construct<edyn::contact_point>().connect<&counter::hit>(ctr);
        auto def = edyn::rigidbody_def{};
        def.mass = 1; def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}}; def.position = {8.0f, 0.52f, 0}; def.sleeping_disabled = true;
        edyn::make_rigidbody(registry, def);
        for (int i = 0; i < 5; ++i) { t += 1.0 / 60; edyn::update(registry, t); }
        REQUIRE(scoped_hits >= 4);   // the new box met the floor: four points
    }
    const size_t frozen = scoped_hits;
    {
        auto def = edyn::rigidbody_def{};
        def.mass = 1; def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}}; def.position = {11.0f, 0.52f, 0}; def.sleeping_disabled = true;
        edyn::make_rigidbody(registry, def);
        for (int i = 0; i < 5; ++i) { t += 1.0 / 60; edyn::update(registry, t); }
    }
    REQUIRE(scoped_hits == frozen);

    // detach destroys the engine-created entities (edyn.cpp:148-197): the listeners see them go
    edyn::detach(registry);
    REQUIRE(count<edyn::contact_point>(registry) == 0 && count<edyn::contact_manifold>(registry) == 0);
    REQUIRE(obs.points_made == obs.points_gone && obs.manifolds_made == obs.manifolds_gone);
    std::printf("LISTENERS_OK points %zu manifolds %zu\n", obs.points_made, obs.manifolds_made);
    return 0;
}
