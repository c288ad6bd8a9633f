// edyn::attach with init_config::devices: ONE registry, one simulation, several GPUs (two shards on the one GPU of the test box).
// The application code is the reference's (attach / make_rigidbody / make_constraint / update / registry.destroy); the trajectory
// must equal the same registry program on one device, bit for bit - islands never exchange data inside a step (solver.cpp:408-428).
#include <edyn/edyn.hpp>
#include <cmath>
#include <cstdio>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static void build(entt::registry &registry, std::vector<entt::entity> &bodies, entt::entity &roller) {
    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    edyn::make_rigidbody(registry, floor_def);
    for (int site = 0; site < 4; ++site)
        for (int k = 0; k < 8; ++k) {
            auto def = edyn::rigidbody_def{};
            def.mass = 1;
            def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
            def.position = {7.0f * site + 1.02f * (k % 2), 0.505f + 1.005f * ((k / 2) % 2), 1.02f * (k / 4)};
            def.sleeping_disabled = true;
            bodies.push_back(edyn::make_rigidbody(registry, def));
        }
    // a hinge door on the last pile's first box, and a sphere rolling from the first pile towards the second
    auto post = edyn::rigidbody_def{}; post.kind = edyn::rigidbody_kind::rb_static; post.position = {30.0f, 2.0f, 0};
    auto door = edyn::rigidbody_def{}; door.mass = 1; door.position = {31.0f, 2.0f, 0}; door.shape = edyn::box_shape{{0.5f, 0.5f, 0.1f}}; door.sleeping_disabled = true;
    const auto e_post = edyn::make_rigidbody(registry, post), e_door = edyn::make_rigidbody(registry, door);
    edyn::make_constraint<edyn::hinge_constraint>(registry, e_post, e_door, [](edyn::hinge_constraint &h) {
        h.pivot[0] = {0, 0, 0}; h.pivot[1] = {-1.0f, 0, 0}; h.set_axes({0, 0, 1}, {0, 0, 1});
    });
    bodies.push_back(e_door);
    auto ball = edyn::rigidbody_def{};
    ball.mass = 1; ball.shape = edyn::sphere_shape{0.5f}; ball.position = {3.2f, 0.5f, 0.5f}; ball.linvel = {4.0f, 0, 0}; ball.sleeping_disabled = true;
    roller = edyn::make_rigidbody(registry, ball);
    bodies.push_back(roller);
}

int main() {
    entt::registry one, many;
    auto cfg = edyn::init_config{};
    cfg.num_solver_velocity_iterations = 10;
    cfg.materialize_contacts = false;
    edyn::attach(one, cfg);
    cfg.devices = {0, 0};
    edyn::attach(many, cfg);
    std::vector<entt::entity> b1, b2;
    entt::entity r1, r2;
    build(one, b1, r1); build(many, b2, r2);
    double t = 0;
    float door_lowest = 2.0f;
    for (int step = 0; step < 150; ++step) {
        t += 1.0 / 60;
        edyn::update(one, t); edyn::update(many, t);
        door_lowest = std::fmin(door_lowest, many.get<edyn::position>(b2[b2.size() - 2]).y);
        for (size_t k = 0; k < b1.size(); ++k) {
            if (!one.valid(b1[k])) continue;
            const auto &p = one.get<edyn::position>(b1[k]), &q = many.get<edyn::position>(b2[k]);
            const auto &o = one.get<edyn::orientation>(b1[k]), &u = many.get<edyn::orientation>(b2[k]);
            const auto &v = one.get<edyn::linvel>(b1[k]), &w = many.get<edyn::linvel>(b2[k]);
            // bit for bit up to the edit at step 100; the rebuilt multi-device world then starts its contacts without the warm start the
            // single-device world keeps, so from there the two agree physically (a settled scene: within 5 cm), not in the last bits
            const bool same = p.x == q.x && p.y == q.y && p.z == q.z && o.x == u.x && o.y == u.y && o.z == u.z && o.w == u.w && v.x == w.x && v.y == w.y && v.z == w.z;
            const bool close = std::fabs(p.x - q.x) < 0.05f && std::fabs(p.y - q.y) < 0.05f && std::fabs(p.z - q.z) < 0.05f;
            if (step <= 100 ? !same : !close) {
                std::printf("FAILED: body %zu differs at step %d (%g %g %g vs %g %g %g)\n", k, step, p.x, p.y, p.z, q.x, q.y, q.z);
                return 1;
            }
        }
        if (step == 100) { one.destroy(b1[3]); many.destroy(b2[3]); }   // an edit of the running worlds: the multi-device world is rebuilt from the registry
    }
    REQUIRE(one.get<edyn::position>(r1).x > 6.0f);                        // the ball did reach the second pile
    REQUIRE(door_lowest < 1.3f);                                          // the door swung down on its hinge (a pendulum of length 1 from y = 2)
    std::printf("MULTI_SHIM_OK\n");
    return 0;
}
