// The reference's README main loop (README.md:126-139) and examples/hello_world, ported verbatim in structure:
// attach -> make_rigidbody -> update loop -> read components -> detach. Links only libedynhip.so.
#include <edyn/edyn.hpp>
#include <cstdio>
#include <cmath>

int main() {
    entt::registry registry;
    auto config = edyn::init_config{};
    config.num_solver_velocity_iterations = 10;
    edyn::attach(registry, config);

    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    edyn::make_rigidbody(registry, floor_def);

    auto def = edyn::rigidbody_def{};
    def.kind = edyn::rigidbody_kind::rb_dynamic;
    def.position = {0, 3, 0};
    def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
    def.mass = 1;
    auto box = edyn::make_rigidbody(registry, def);
    def.position = {0.1f, 4.2f, 0};
    auto box2 = edyn::make_rigidbody(registry, def);

    // a pendulum on a hinge next to the stack
    auto anchor_def = edyn::rigidbody_def{};
    anchor_def.kind = edyn::rigidbody_kind::rb_static;
    anchor_def.position = {5, 5, 0};
    auto anchor = edyn::make_rigidbody(registry, anchor_def);
    auto bob_def = edyn::rigidbody_def{};
    bob_def.position = {6, 5, 0};
    bob_def.inertia = edyn::matrix3x3{{edyn::vector3{0.01f, 0, 0}, edyn::vector3{0, 0.01f, 0}, edyn::vector3{0, 0, 0.01f}}};
    auto bob = edyn::make_rigidbody(registry, bob_def);
    edyn::make_constraint<edyn::hinge_constraint>(registry, anchor, bob, [](edyn::hinge_constraint &h) {
        h.pivot = {edyn::vector3{0, 0, 0}, edyn::vector3{-1, 0, 0}};
        h.set_axes({0, 0, 1}, {0, 0, 1});
    });

    double t = 0;
    for (int i = 0; i < 240; ++i) {
        t += 1.0 / 60 + 1e-6;
        edyn::update(registry, t);
    }
    const auto &p = registry.get<edyn::position>(box);
    const auto &p2 = registry.get<edyn::position>(box2);
    const auto &pb = registry.get<edyn::position>(bob);
    const auto &v = registry.get<edyn::linvel>(box);
    std::printf("box  pos (%.4f, %.4f, %.4f) vel (%.4f, %.4f, %.4f)\n", p.x, p.y, p.z, v.x, v.y, v.z);
    std::printf("box2 pos (%.4f, %.4f, %.4f)\n", p2.x, p2.y, p2.z);
    std::printf("bob  pos (%.4f, %.4f, %.4f)\n", pb.x, pb.y, pb.z);
    // what a renderer draws: present_position trails the simulated state by up to one fixed step
    const auto &pp = registry.get<edyn::present_position>(box);
    std::printf("box  present (%.4f, %.4f, %.4f)\n", pp.x, pp.y, pp.z);
    auto manifolds = edyn::get_contact_manifolds(registry);
    std::printf("manifolds %zu\n", manifolds.size());
    bool ok = std::fabs(pp.y - p.y) < 1e-3f && std::fabs(p.y - 0.5f) < 5e-3f && std::fabs(p2.y - 1.5f) < 1e-2f && manifolds.size() == 2;
    const float L = std::sqrt((pb.x - 5) * (pb.x - 5) + (pb.y - 5) * (pb.y - 5) + pb.z * pb.z);
    ok = ok && std::fabs(L - 1.0f) < 3e-2f && pb.y < 5.0f;
    // a third box created while the world is running: the stepper appends it (edynhip_add_bodies) and the resting
    // contacts of the first two keep their cached impulses, so the stack does not twitch.
    def.position = {0.05f, 2.6f, 0};
    auto box3 = edyn::make_rigidbody(registry, def);
    t += 1.0 / 60 + 1e-6;
    edyn::update(registry, t);
    const float jolt = std::fabs(registry.get<edyn::linvel>(box).y);
    for (int i = 0; i < 180; ++i) {
        t += 1.0 / 60 + 1e-6;
        edyn::update(registry, t);
    }
    const auto &p3 = registry.get<edyn::position>(box3);
    std::printf("box3 pos (%.4f, %.4f, %.4f) jolt %.5f manifolds %zu\n", p3.x, p3.y, p3.z, jolt, edyn::get_contact_manifolds(registry).size());
    ok = ok && std::fabs(p3.y - 2.5f) < 2e-2f && jolt < 0.02f && edyn::get_contact_manifolds(registry).size() == 3;
    edyn::detach(registry);
    std::printf(ok ? "HELLO_WORLD_OK\n" : "HELLO_WORLD_FAIL\n");
    return ok ? 0 : 1;
}
