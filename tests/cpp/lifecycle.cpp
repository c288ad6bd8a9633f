// Life-cycle of a running world through the C++ drop-in shim: everything the reference lets a user do between two
// edyn::update calls without losing the simulation state.
//   registry.destroy(body) / registry.destroy(constraint) / edyn::clear_rigidbody      src/edyn/edyn.cpp:148-197 hooks, island_manager.cpp:47-115
//   set_solver_*_iterations / set_gravity / set_fixed_dt on a running world            solver_iteration_config.cpp:9-75, gravity_util.cpp:12-20
//   make_constraint<T>(registry, entity, body0, body1, setup...) and the zero-setup form   util/constraint_util.hpp:38-54
//   exclude_collision                                                                   util/exclude_collision.hpp:20-47
//   capacity growth while running (contacts carried over), update(registry) without a time argument
#include <edyn/edyn.hpp>
#include <edyn/util/ragdoll.hpp>
#include <cmath>
#include <cstdio>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++failures; } } while (0)

int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    entt::registry registry;
    auto config = edyn::init_config{};
    config.num_solver_velocity_iterations = 10;
    config.max_bodies = 8;   // small on purpose: the world outgrows it below
    edyn::attach(registry, config);

    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    edyn::make_rigidbody(registry, floor_def);

    auto def = edyn::rigidbody_def{};
    def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
    def.sleeping_disabled = true;
    entt::entity stack[3];
    for (int i = 0; i < 3; ++i) { def.position = {0, 0.5f + 1.0f * i, 0}; stack[i] = edyn::make_rigidbody(registry, def); }

    double t = 0;
    auto run = [&](int frames) { for (int i = 0; i < frames; ++i) { t += 1.0 / 60 + 1e-6; edyn::update(registry, t); } };
    run(60);
    CHECK(edyn::get_contact_manifolds(registry).size() == 3);
    CHECK(std::fabs(registry.get<edyn::position>(stack[2]).y - 2.5f) < 2e-2f);

    // settings on the running world: the stack must not twitch (contacts and warm-start impulses survive)
    edyn::set_solver_velocity_iterations(registry, 14);
    edyn::set_solver_position_iterations(registry, 2);
    run(1);
    CHECK(std::fabs(registry.get<edyn::linvel>(stack[2]).y) < 0.02f);
    CHECK(edyn::get_solver_velocity_iterations(registry) == 14);
    edyn::set_gravity(registry, {0, -4.9f, 0});
    run(30);
    CHECK(std::fabs(registry.get<edyn::position>(stack[2]).y - 2.5f) < 2e-2f);
    edyn::set_gravity(registry, edyn::gravity_earth);

    // destroy the middle box: the top one falls onto the bottom one
    registry.destroy(stack[1]);
    run(90);
    CHECK(std::fabs(registry.get<edyn::position>(stack[2]).y - 1.5f) < 3e-2f);
    CHECK(edyn::get_contact_manifolds(registry).size() == 2);

    // outgrow max_bodies: eight more boxes in a row; the resting pair keeps its contacts through the re-allocation
    for (int i = 0; i < 8; ++i) { def.position = {3.0f + 1.5f * i, 0.5f, 0}; edyn::make_rigidbody(registry, def); }
    run(1);
    CHECK(std::fabs(registry.get<edyn::linvel>(stack[2]).y) < 0.05f);
    run(60);
    CHECK(edyn::get_contact_manifolds(registry).size() == 10);

    // constraints: entity form with two setup functions, zero-setup form, destruction
    auto anchor_def = edyn::rigidbody_def{};
    anchor_def.kind = edyn::rigidbody_kind::rb_static;
    anchor_def.position = {-5, 5, 0};
    auto anchor = edyn::make_rigidbody(registry, anchor_def);
    auto bob_def = edyn::rigidbody_def{};
    bob_def.position = {-4, 5, 0};
    bob_def.sleeping_disabled = true;
    bob_def.inertia = edyn::matrix3x3{{edyn::vector3{0.01f, 0, 0}, edyn::vector3{0, 0.01f, 0}, edyn::vector3{0, 0, 0.01f}}};
    auto bob = edyn::make_rigidbody(registry, bob_def);
    auto hinge_entity = registry.create();
    edyn::make_constraint<edyn::hinge_constraint>(registry, hinge_entity, anchor, bob,
        [](edyn::hinge_constraint &h) { h.pivot = {edyn::vector3{0, 0, 0}, edyn::vector3{-1, 0, 0}}; },
        [](edyn::hinge_constraint &h) { h.set_axes({0, 0, 1}, {0, 0, 1}); h.angle_min = -0.4f; h.angle_max = 0.4f; });
    auto loose = edyn::make_constraint<edyn::point_constraint>(registry, anchor, bob);   // pivots at the origins: removed right away
    registry.destroy(loose);
    run(120);
    {
        const auto &pb = registry.get<edyn::position>(bob);
        const float L = std::sqrt((pb.x + 5) * (pb.x + 5) + (pb.y - 5) * (pb.y - 5) + pb.z * pb.z);
        const float angle = std::atan2(5 - pb.y, pb.x + 5);   // 0 = horizontal start, positive = swung down
        CHECK(std::fabs(L - 1.0f) < 3e-2f);
        CHECK(angle < 0.4f + 0.06f);                          // the limit holds the pendulum up
    }
    registry.destroy(hinge_entity);                           // now it simply falls
    run(60);
    CHECK(registry.get<edyn::position>(bob).y < 3.5f);

    // exclude_collision: a box dropped onto an excluded partner falls through it to the floor
    def.position = {-10, 0.5f, 0}; auto lower = edyn::make_rigidbody(registry, def);
    def.position = {-10, 1.6f, 0}; auto upper = edyn::make_rigidbody(registry, def);
    edyn::exclude_collision(registry, lower, upper);
    run(90);
    CHECK(std::fabs(registry.get<edyn::position>(upper).y - 0.5f) < 3e-2f);

    // set_should_collide (collision/should_collide.hpp:18): a user predicate decides about NEW manifolds - a ghost box falls through a
    // solid one although nothing excludes the pair; with the default back, the next ghost lands on it
    {
        static entt::entity ghost;
        def.position = {-14, 0.5f, 0}; auto solid = edyn::make_rigidbody(registry, def);
        def.position = {-14, 1.6f, 0}; ghost = edyn::make_rigidbody(registry, def);
        static int asked = 0;
        edyn::set_should_collide(registry, [](const entt::registry &r, entt::entity a, entt::entity b) {
            ++asked;
            const bool floor = !r.all_of<edyn::dynamic_tag>(a) || !r.all_of<edyn::dynamic_tag>(b);
            return edyn::should_collide_default(r, a, b) && (floor || (a != ghost && b != ghost));
        });
        run(90);
        CHECK(asked > 0);
        CHECK(std::fabs(registry.get<edyn::position>(ghost).y - 0.5f) < 3e-2f);      // through the solid box, onto the floor
        CHECK(std::fabs(registry.get<edyn::position>(solid).y - 0.5f) < 3e-2f);
        edyn::set_should_collide(registry, &edyn::should_collide_default);
        def.position = {-18, 0.5f, 0}; edyn::make_rigidbody(registry, def);
        def.position = {-18, 1.6f, 0}; auto lands = edyn::make_rigidbody(registry, def);
        run(90);
        CHECK(std::fabs(registry.get<edyn::position>(lands).y - 1.5f) < 3e-2f);
    }

    // clear_rigidbody keeps the entity but takes the body out of the world
    edyn::clear_rigidbody(registry, lower);
    CHECK(registry.valid(lower));
    run(5);
    for (auto &m : edyn::get_contact_manifolds(registry)) CHECK(m.body[0] != lower && m.body[1] != lower);

    // step callbacks: called once before and once after every fixed step, with the registry current in between
    static int pre_calls = 0, post_calls = 0;
    static float last_seen_y = 0;
    static entt::entity watched;
    watched = upper;
    edyn::set_pre_step_callback(registry, [](entt::registry &) { ++pre_calls; });
    edyn::set_post_step_callback(registry, [](entt::registry &r) { ++post_calls; last_seen_y = r.get<edyn::position>(watched).y; });
    run(7);
    CHECK(pre_calls == 7 && post_calls == 7);
    CHECK(last_seen_y == registry.get<edyn::position>(upper).y);
    edyn::set_pre_step_callback(registry, nullptr);
    edyn::set_post_step_callback(registry, nullptr);

    // edits of a body between updates (util/rigidbody.hpp:105-260)
    {
        auto bdef = edyn::rigidbody_def{};
        bdef.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
        bdef.position = {20, 0.5f, 0};
        const entt::entity puck = edyn::make_rigidbody(registry, bdef);   // sleeping enabled
        run(30);
        // an impulse through the centre of mass: dv = J / m; a torque impulse about y: dw = I^-1 L with I = m/6 for the unit cube
        edyn::rigidbody_apply_impulse(registry, puck, {0, 3, 0}, {0, 0, 0});
        CHECK(std::fabs(registry.get<edyn::linvel>(puck).y - 3.0f) < 1e-5f);
        edyn::rigidbody_apply_torque_impulse(registry, puck, {0, 0.5f, 0});
        CHECK(std::fabs(registry.get<edyn::angvel>(puck).y - 3.0f) < 1e-4f);
        run(10);
        CHECK(registry.get<edyn::position>(puck).y > 0.8f);                    // it took off ...
        CHECK(std::fabs(registry.get<edyn::angvel>(puck).y - 3.0f) < 0.05f);   // ... spinning
        run(240);   // lands, settles and falls asleep (the stack has sleeping disabled, this body not)
        CHECK(std::fabs(registry.get<edyn::position>(puck).y - 0.5f) < 2e-2f);
        CHECK(registry.all_of<edyn::sleeping_tag>(puck));
        edyn::wake_up_entity(registry, puck);
        run(1);
        CHECK(!registry.all_of<edyn::sleeping_tag>(puck));
        // a heavier body presses harder on the floor: the normal impulses follow the new mass, the contact itself survives the edit
        edyn::set_rigidbody_mass(registry, puck, 4.0f);
        run(30);
        float weight = 0;
        {
            auto &s = registry.ctx().get<edyn::detail::gpu_stepper>();
            uint32_t nm = 0; edynhip_num_manifolds(s.ctx, &nm);
            std::vector<edynhip_manifold> recs(nm); edynhip_get_manifolds(s.ctx, recs.data(), nm, &nm);
            const uint32_t pi = registry.get<edyn::detail::body_index>(puck).value;
            for (auto &r : recs) if (r.body[0] == pi || r.body[1] == pi) for (uint32_t k = 0; k < r.num_points; ++k) weight += r.pt[k].normal_impulse;
        }
        CHECK(std::fabs(weight - 4.0f * 9.8f / 60.0f) < 0.02f);
        // friction: pushed sideways with mu = 0.5 it stops within a metre; on ice (mu = 0) it keeps going
        edyn::set_rigidbody_friction(registry, puck, 0.0f);
        run(2);
        edyn::rigidbody_apply_impulse(registry, puck, {8, 0, 0}, {0, 0, 0});   // 2 m/s
        run(60);
        CHECK(registry.get<edyn::position>(puck).x > 21.8f && std::fabs(registry.get<edyn::linvel>(puck).x - 2.0f) < 0.05f);
        // another shape and another kind on a running world: the puck becomes a ball, then a static obstacle that no longer falls
        edyn::rigidbody_set_shape(registry, puck, edyn::shapes_variant_t{edyn::sphere_shape{0.5f}});
        run(30);
        CHECK(edyn::rigidbody_has_shape(registry, puck) && registry.all_of<edyn::sphere_shape>(puck) && !registry.all_of<edyn::box_shape>(puck));
        edyn::rigidbody_set_kind(registry, puck, edyn::rigidbody_kind::rb_static);
        const float frozen_x = registry.get<edyn::position>(puck).x;
        run(30);
        CHECK(registry.all_of<edyn::static_tag>(puck) && registry.get<edyn::position>(puck).x == frozen_x);
        // a kinematic platform moved by the user: its velocity is what takes it there in dt
        auto kdef = edyn::rigidbody_def{};
        kdef.kind = edyn::rigidbody_kind::rb_kinematic;
        kdef.shape = edyn::box_shape{{1, 0.1f, 1}};
        kdef.position = {40, 1, 0};
        const entt::entity lift = edyn::make_rigidbody(registry, kdef);
        run(1);
        edyn::set_kinematic_position(registry, lift, {40, 1.5f, 0}, 0.5f);
        CHECK(std::fabs(registry.get<edyn::linvel>(lift).y - 1.0f) < 1e-6f && registry.get<edyn::position>(lift).y == 1.5f);
        const float h = 0.5f * 0.3f;
        edyn::set_kinematic_orientation(registry, lift, {0, std::sin(h), 0, std::cos(h)}, 0.1f);   // 0.3 rad in 0.1 s
        CHECK(std::fabs(registry.get<edyn::angvel>(lift).y - 3.0f) < 1e-3f);
        run(2);
    }

    // rigidbody_def::center_of_mass: a sphere loaded at the bottom, laid on its side, rolls back upright and stays there
    {
        auto wdef = edyn::rigidbody_def{};
        wdef.shape = edyn::sphere_shape{0.5f};
        wdef.center_of_mass = edyn::vector3{0, -0.35f, 0};
        wdef.material->friction = 1.0f;
        wdef.material->roll_friction = 0.03f;   // rolling resistance damps the rocking
        wdef.sleeping_disabled = true;
        const float h = 0.5f * 1.2f;   // tilted 1.2 rad about z
        wdef.orientation = {0, 0, std::sin(h), std::cos(h)};
        wdef.position = {60, 0.5f, 0};   // the ORIGIN (centre of the sphere)
        const entt::entity weeble = edyn::make_rigidbody(registry, wdef);
        const auto com0 = registry.get<edyn::position>(weeble);
        CHECK(std::fabs(com0.y - (0.5f - 0.35f * std::cos(1.2f))) < 1e-5f && std::fabs(com0.x - (60 + 0.35f * std::sin(1.2f))) < 1e-5f);
        float lowest = 1, highest = 0;
        for (int k = 0; k < 900; ++k) { run(1); const float y = registry.get<edyn::position>(weeble).y; lowest = std::fmin(lowest, y); highest = std::fmax(highest, y); }
        std::printf("weeble: centre of mass y in [%.3f, %.3f], final %.3f, q.w %.4f\n", lowest, highest, registry.get<edyn::position>(weeble).y, registry.get<edyn::orientation>(weeble).w);
        const auto q = registry.get<edyn::orientation>(weeble);
        const auto o = edyn::get_rigidbody_origin(registry, weeble);
        const auto p = registry.get<edyn::position>(weeble);
        CHECK(highest < com0.y + 1e-3f && lowest < 0.16f);               // it never gained height, and swung through the upright pose
        CHECK(std::fabs(q.w) > 0.98f || std::fabs(q.y) > 0.98f);         // upright again (it may have turned about the vertical)
        CHECK(std::fabs(o.y - 0.5f) < 2e-2f && std::fabs(p.y - 0.15f) < 2e-2f);   // origin at the sphere's centre, centre of mass 0.35 below it
    }

    // update(registry) without a time: the monotonic clock drives the accumulator - or the user's time source
    edyn::update(registry);
    edyn::update(registry);
    static double fake_now = 1000.0;
    edyn::set_time_source(registry, [] { return fake_now; });
    CHECK(edyn::get_time(registry) == 1000.0);
    CHECK(edyn::get_execution_mode(registry) == edyn::execution_mode::sequential && edyn::get_max_steps_per_update(registry) >= 1);
    edyn::detach(registry);
    // ---- a rag doll (21 collision exclusions between its limbs, 36 constraints) in a world that outgrows its capacity and then has
    // a body's friction edited (both re-create the device context): the exclusions, the joints' warm-start impulses and tracked
    // angles must travel - no manifold may ever appear between two bodies of the figure (they used to be dropped)
    {
        entt::registry world;
        auto cfg = edyn::init_config{};
        cfg.max_bodies = 32;   // the figure fits; the crowd added below does not
        edyn::attach(world, cfg);
        auto ground = edyn::rigidbody_def{};
        ground.kind = edyn::rigidbody_kind::rb_static;
        ground.shape = edyn::plane_shape{{0, 1, 0}, 0};
        edyn::make_rigidbody(world, ground);
        edyn::ragdoll_simple_def rd;
        rd.position = {0, 1.2f, 0};
        const edyn::ragdoll_entities rag = edyn::make_ragdoll(world, rd);
        (void)rag;
        auto &st = world.ctx().get<edyn::detail::gpu_stepper>();
        const size_t figure_first = 1, figure_end = st.bodies.size();
        auto is_limb = [&](entt::entity e) {
            for (size_t i = figure_first; i < figure_end; ++i) if (st.bodies[i] == e) return true;
            return false;
        };
        auto excluded_pair_touching = [&]() {   // a manifold between two limbs that exclude each other
            int bad = 0;
            for (auto &m : edyn::get_contact_manifolds(world)) {
                if (!is_limb(m.body[0]) || !is_limb(m.body[1])) continue;
                const uint32_t a = world.get<edyn::detail::body_index>(m.body[0]).value, b = world.get<edyn::detail::body_index>(m.body[1]).value;
                for (auto &ex : st.exclusions) if ((ex[0] == a && ex[1] == b) || (ex[0] == b && ex[1] == a)) ++bad;
            }
            return bad;
        };
        double tw = 0;
        auto step = [&](int frames) { for (int i = 0; i < frames; ++i) { tw += 1.0 / 60; edyn::update(world, tw); } };
        step(45);   // the figure collapses onto the floor: its limbs fold over each other
        CHECK(st.exclusions.size() >= 20);
        CHECK(excluded_pair_touching() == 0);
        auto crowd = edyn::rigidbody_def{};
        crowd.shape = edyn::box_shape{{0.25f, 0.25f, 0.25f}};
        for (int i = 0; i < 40; ++i) { crowd.position = {4.0f + 0.7f * (i % 8), 0.25f + 0.6f * (i / 8), 3.0f}; edyn::make_rigidbody(world, crowd); }
        step(45);   // capacity growth happened in the first of these updates
        CHECK(st.capacity > 32);
        CHECK(excluded_pair_touching() == 0);
        edyn::set_rigidbody_friction(world, st.bodies[figure_first], 0.9f);   // re-creates the context once more
        step(45);
        CHECK(excluded_pair_touching() == 0);
        const auto &pelvis = world.get<edyn::position>(st.bodies[figure_first]);
        CHECK(std::isfinite(pelvis.x) && std::isfinite(pelvis.y) && pelvis.y > 0.0f && pelvis.y < 1.5f);   // lying on the floor, in one piece
        // removing an exclusion takes it off the list for good (it is not replayed into the next context)
        const size_t before = st.exclusions.size();
        edyn::remove_collision_exclusion(world, st.bodies[st.exclusions[0][0]], st.bodies[st.exclusions[0][1]]);
        CHECK(st.exclusions.size() == before - 1);
        step(2);
    }
    // --- a paused stepper: update() snaps the presentation to the state (stepper_sequential.cpp:38-43), step_simulation() takes exactly one
    //     step and leaves the presentation alone (:121-147); step callbacks in asynchronous mode take the synchronous write-back, contact
    //     entities included (round 6: the event prefetch is on in every mode)
    {
        entt::registry world;
        auto cfg = edyn::init_config{};
        cfg.execution_mode = edyn::execution_mode::asynchronous;
        edyn::attach(world, cfg);
        auto floor_def = edyn::rigidbody_def{};
        floor_def.kind = edyn::rigidbody_kind::rb_static;
        floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
        edyn::make_rigidbody(world, floor_def);
        auto def = edyn::rigidbody_def{};
        def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
        def.position = {0, 0.6f, 0};
        const auto box = edyn::make_rigidbody(world, def);
        static int pre_calls = 0, post_calls = 0;
        pre_calls = post_calls = 0;
        edyn::set_pre_step_callback(world, [](entt::registry &) { ++pre_calls; });
        edyn::set_post_step_callback(world, [](entt::registry &) { ++post_calls; });
        double tw = 0;
        for (int i = 0; i < 30; ++i) { tw += 1.0 / 60; edyn::update(world, tw); }
        CHECK(pre_calls == post_calls && pre_calls >= 29);
        size_t points = 0;
        world.view<edyn::contact_point>().each([&](auto, auto &) { ++points; });
        CHECK(points >= 1);   // the callbacks' synchronous write-back built the contact entities
        CHECK(std::fabs(world.get<edyn::position>(box).y - 0.5f) < 0.02f);
        edyn::set_pre_step_callback(world, nullptr); edyn::set_post_step_callback(world, nullptr);
        edyn::set_paused(world, true);
        world.get<edyn::present_position>(box).y = 42.0f;
        edyn::update(world, tw + 1.0);   // paused: no step, the presentation snaps to the state
        CHECK(world.get<edyn::present_position>(box).y == world.get<edyn::position>(box).y);
        edyn::rigidbody_apply_impulse(world, box, {0, 3.0f, 0}, {0, 0, 0});
        world.get<edyn::present_position>(box).y = 42.0f;
        const float y0 = world.get<edyn::position>(box).y;
        edyn::step_simulation(world, tw + 2.0);
        edyn::step_simulation(world, tw + 2.0 + 1.0 / 60);
        CHECK(world.get<edyn::position>(box).y > y0 + 0.01f);            // two steps were taken (asynchronous: the first one has arrived)
        CHECK(world.get<edyn::present_position>(box).y == 42.0f);        // ... and the presentation was not touched
        edyn::detach(world);
    }
    std::printf(failures == 0 ? "LIFECYCLE_OK\n" : "LIFECYCLE_FAIL\n");
    return failures == 0 ? 0 : 1;
}
