// Contact entities and the asynchronous execution mode through the C++ shim:
//  * contact_manifold / contact_point entities appear and disappear in the registry as the device reports them
//    (make_contact_manifold constraint_util.cpp:60-102, create_contact_point collision_util.cpp:311-388);
//  * execution_mode::asynchronous hands the registry the previous update's state while the next one runs
//    (simulation_worker.cpp:406-444): its trajectory is the synchronous one, one update late.
#include <edyn/edyn.hpp>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static void build(entt::registry &registry, std::vector<entt::entity> &boxes) {
    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    edyn::make_rigidbody(registry, floor_def);
    for (int i = 0; i < 27; ++i) {
        auto def = edyn::rigidbody_def{};
        def.mass = 1;
        def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
        def.position = {1.02f * (i % 3), 0.505f + 1.005f * ((i / 3) % 3), 1.02f * (i / 9)};
        def.sleeping_disabled = true;
        boxes.push_back(edyn::make_rigidbody(registry, def));
    }
}
template <typename T> static size_t count(entt::registry &registry) {
    size_t n = 0;
    registry.view<T>().each([&](auto, auto &) { ++n; });
    return n;
}

int main() {
    // ---- contact entities
    entt::registry registry;
    auto cfg = edyn::init_config{};
    cfg.contact_point_data = true;
    edyn::attach(registry, cfg);
    std::vector<entt::entity> boxes;
    build(registry, boxes);
    double t = 0;
    for (int i = 0; i < 90; ++i) { t += 1.0 / 60; edyn::update(registry, t); }
    const auto manifolds = edyn::get_contact_manifolds(registry);   // straight from the device
    size_t device_points = 0;
    for (auto &m : manifolds) device_points += m.num_points;
    REQUIRE(!manifolds.empty() && device_points > 27);
    REQUIRE(count<edyn::contact_manifold>(registry) == manifolds.size());
    REQUIRE(count<edyn::contact_point>(registry) == device_points);
    size_t listed = 0, penetrating = 0;
    registry.view<edyn::contact_manifold>().each([&](auto, edyn::contact_manifold &m) { listed += m.num_points; });
    REQUIRE(listed == device_points);
    bool parents_ok = true, data_ok = true;
    registry.view<edyn::contact_point_list>().each([&](auto e, edyn::contact_point_list &l) {
        parents_ok = parents_ok && registry.all_of<edyn::contact_manifold>(l.parent);
        const auto &cp = registry.get<edyn::contact_point>(e);
        const float len = std::sqrt(cp.normal.x * cp.normal.x + cp.normal.y * cp.normal.y + cp.normal.z * cp.normal.z);
        data_ok = data_ok && std::fabs(len - 1.0f) < 1e-4f;   // contact_point_data: the normals were read back
        if (registry.get<edyn::contact_point_geometry>(e).distance < 0.02f) ++penetrating;
    });
    REQUIRE(parents_ok && data_ok && penetrating > 0);
    // util/contact_manifold_util.hpp, util/constraint_util.hpp visit_*: box 3 rests on box 0 (same column, one layer up)
    REQUIRE(edyn::manifold_exists(registry, boxes[3], boxes[0]) && edyn::manifold_exists(registry, boxes[0], boxes[3]));
    REQUIRE(registry.all_of<edyn::contact_manifold>(edyn::get_manifold_entity(registry, boxes[3], boxes[0])));
    REQUIRE(!edyn::manifold_exists(registry, boxes[0], boxes[26]));
    bool below_seen = false; size_t neighbours = 0;
    edyn::visit_neighbors(registry, boxes[3], [&](entt::entity o) { ++neighbours; below_seen = below_seen || o == boxes[0]; });
    REQUIRE(below_seen && neighbours >= 2);
    // destroying a body takes its manifolds and points with it (at the next update)
    registry.destroy(boxes.back());
    t += 1.0 / 60; edyn::update(registry, t);
    const auto after = edyn::get_contact_manifolds(registry);
    size_t after_points = 0;
    for (auto &m : after) after_points += m.num_points;
    REQUIRE(after.size() < manifolds.size());
    REQUIRE(count<edyn::contact_manifold>(registry) == after.size() && count<edyn::contact_point>(registry) == after_points);

    // ---- asynchronous mode: the synchronous trajectory, one update late
    entt::registry sync_reg, async_reg;
    auto acfg = edyn::init_config{};
    edyn::attach(sync_reg, acfg);
    acfg.execution_mode = edyn::execution_mode::asynchronous;
    edyn::attach(async_reg, acfg);
    std::vector<entt::entity> sb, ab;
    build(sync_reg, sb); build(async_reg, ab);
    std::vector<edyn::position> history;
    double ts = 0;
    for (int i = 0; i < 40; ++i) {
        ts += 1.0 / 60;
        edyn::update(sync_reg, ts); edyn::update(async_reg, ts);
        history.push_back(sync_reg.get<edyn::position>(sb[13]));
        if (i > 0) {
            const auto &p = async_reg.get<edyn::position>(ab[13]);
            REQUIRE(p.x == history[i - 1].x && p.y == history[i - 1].y && p.z == history[i - 1].z);
        }
    }
    REQUIRE(count<edyn::contact_manifold>(async_reg) > 0);
    // ---- asynchronous mode with the user editing between updates: a body made mid-run and an impulse applied mid-run
    // must reach the device as they would in synchronous mode - the registry lags one update, the simulation does not
    {
        entt::registry sreg, areg;
        auto c2 = edyn::init_config{};
        edyn::attach(sreg, c2);
        c2.execution_mode = edyn::execution_mode::asynchronous;
        edyn::attach(areg, c2);
        std::vector<entt::entity> s2, a2;
        build(sreg, s2); build(areg, a2);
        entt::entity s_new = entt::null, a_new = entt::null;
        std::vector<edyn::position> hist13, hist_new;
        double t2 = 0;
        for (int i = 0; i < 60; ++i) {
            t2 += 1.0 / 60;
            edyn::update(sreg, t2); edyn::update(areg, t2);
            hist13.push_back(sreg.get<edyn::position>(s2[13]));
            if (s_new != entt::null) hist_new.push_back(sreg.get<edyn::position>(s_new));
            if (i == 20) {   // between two updates: a new box above the pile, an impulse on box 13 (both worlds alike)
                auto def = edyn::rigidbody_def{};
                def.mass = 1; def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}}; def.position = {1.02f, 6.0f, 1.02f}; def.sleeping_disabled = true;
                s_new = edyn::make_rigidbody(sreg, def); a_new = edyn::make_rigidbody(areg, def);
                edyn::rigidbody_apply_impulse(sreg, s2[13], {0, 0, 3.0f}, {0, 0, 0});
                edyn::rigidbody_apply_impulse(areg, a2[13], {0, 0, 3.0f}, {0, 0, 0});
            }
            if (i >= 22) {   // one update late again (the velocity edit is carried as an increment: equal up to its rounding)
                const auto &p = areg.get<edyn::position>(a2[13]);
                REQUIRE(std::fabs(p.x - hist13[i - 1].x) < 1e-4f && std::fabs(p.y - hist13[i - 1].y) < 1e-4f && std::fabs(p.z - hist13[i - 1].z) < 1e-4f);
                const auto &pn = areg.get<edyn::position>(a_new);
                const auto &hn = hist_new[hist_new.size() - 2];
                REQUIRE(std::isfinite(pn.y) && std::fabs(pn.x - hn.x) < 1e-4f && std::fabs(pn.y - hn.y) < 1e-4f && std::fabs(pn.z - hn.z) < 1e-4f);
            }
        }
        REQUIRE(std::fabs(sreg.get<edyn::position>(s2[13]).z - 1.02f) > 0.05f);   // the impulse did move it
        const auto &qn = areg.get<edyn::orientation>(a_new);
        REQUIRE(std::fabs(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w - 1.0f) < 1e-4f);   // never a zero quaternion
    }
    // ---- contact_extras through the shim: material::roll_friction stops a rolling sphere, the default lets it roll on
    entt::registry roll_reg;
    edyn::attach(roll_reg, edyn::init_config{});
    {
        auto floor_def = edyn::rigidbody_def{};
        floor_def.kind = edyn::rigidbody_kind::rb_static;
        floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
        floor_def.material = edyn::material{};
        floor_def.material->friction = 1;
        edyn::make_rigidbody(roll_reg, floor_def);
    }
    entt::entity balls[2];
    for (int i = 0; i < 2; ++i) {
        auto def = edyn::rigidbody_def{};
        def.mass = 1;
        def.shape = edyn::sphere_shape{0.5f};
        def.position = {0, 0.5f, 3.0f * i};
        def.linvel = {2, 0, 0};
        def.angvel = {0, 0, -4};   // rolling without slipping: v = w x r
        def.sleeping_disabled = true;
        def.material = edyn::material{};
        def.material->friction = 1;
        if (i == 1) def.material->roll_friction = 0.2f;
        balls[i] = edyn::make_rigidbody(roll_reg, def);
    }
    double tr = 0;
    for (int i = 0; i < 180; ++i) { tr += 1.0 / 60; edyn::update(roll_reg, tr); }
    const float v_free = roll_reg.get<edyn::linvel>(balls[0]).x, v_braked = roll_reg.get<edyn::linvel>(balls[1]).x;
    REQUIRE(v_free > 1.5f);            // nothing slows a sphere rolling on a plane
    REQUIRE(std::fabs(v_braked) < 0.2f);   // rolling friction brought this one to rest
    std::printf("contacts OK: %zu manifolds, %zu points mirrored; asynchronous mode lags one update, bit-identical\n", manifolds.size(), device_points);
    return 0;
}
