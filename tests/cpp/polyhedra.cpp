// polyhedron_shape through the C++ shim, written as code against the reference is written
// (test/edyn/collision/test_collision.cpp:92-103: make_box_mesh, initialize, polyhedron_shape{mesh}):
// convex polyhedra - a box mesh shared by several bodies, an off-centre wedge - dropped on a floor among boxes and spheres;
// more bodies sharing the mesh are created while the world runs (the context is re-created: the meshes go up again).
// Prints every body's final transform; tests/test_cpp_shim.py compares them with the same scene stepped through the C ABI from Python.
#include <edyn/edyn.hpp>
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    entt::registry registry;
    auto cfg = edyn::init_config{};
    cfg.num_solver_velocity_iterations = 10;
    edyn::attach(registry, cfg);

    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    std::vector<entt::entity> bodies{edyn::make_rigidbody(registry, floor_def)};

    auto cube = std::make_shared<edyn::convex_mesh>();
    edyn::make_box_mesh({0.5f, 0.5f, 0.5f}, cube->vertices, cube->indices, cube->faces);
    cube->initialize();
    auto wedge = std::make_shared<edyn::convex_mesh>();
    wedge->vertices = {{-0.5f, -0.25f, -0.3f}, {0.5f, -0.25f, -0.3f}, {0.5f, -0.25f, 0.3f}, {-0.5f, -0.25f, 0.3f}, {-0.5f, 0.25f, -0.3f}, {-0.5f, 0.25f, 0.3f}};
    wedge->indices = {0, 1, 2, 3, 0, 4, 1, 3, 2, 5, 0, 3, 5, 4, 1, 4, 5, 2};
    wedge->faces = {0, 4, 4, 3, 7, 3, 10, 4, 14, 4};
    wedge->initialize();
    REQUIRE(std::fabs(wedge->vertices[0].x + 0.5f) > 0.05f);   // initialize() moved the vertices to the centroid

    auto drop = [&](edyn::shapes_variant_t shape, float x, float y, float z) {
        auto def = edyn::rigidbody_def{};
        def.mass = 2;
        def.shape = shape;
        def.position = {x, y, z};
        def.sleeping_disabled = true;
        bodies.push_back(edyn::make_rigidbody(registry, def));
    };
    for (int i = 0; i < 3; ++i) drop(edyn::polyhedron_shape{cube}, 0.0f, 0.52f + 1.03f * i, 0.0f);       // a stack of polyhedral cubes
    for (int i = 0; i < 3; ++i) drop(edyn::box_shape{{0.5f, 0.5f, 0.5f}}, 3.0f, 0.52f + 1.03f * i, 0.0f);   // the same stack of boxes
    drop(edyn::polyhedron_shape{wedge}, -2.0f, 0.6f, 0.5f);
    drop(edyn::sphere_shape{0.3f}, -2.2f, 1.4f, 0.5f);                                                      // rolls down the wedge
    drop(edyn::polyhedron_shape{wedge}, 0.1f, 3.8f, 0.05f);                                                 // lands on the cube stack

    double t = 0;
    for (int i = 0; i < 120; ++i) { t += 1.0 / 60; edyn::update(registry, t); }
    for (int i = 0; i < 40; ++i) drop(edyn::polyhedron_shape{cube}, 6.0f + 1.2f * (i % 8), 0.6f + 1.1f * (i / 8), 2.0f);   // grows past the capacity
    for (int i = 0; i < 120; ++i) { t += 1.0 / 60; edyn::update(registry, t); }

    for (size_t i = 0; i < bodies.size(); ++i) {
        const auto &p = registry.get<edyn::position>(bodies[i]);
        const auto &q = registry.get<edyn::orientation>(bodies[i]);
        std::printf("body %zu pos %.9g %.9g %.9g orn %.9g %.9g %.9g %.9g\n", i, p.x, p.y, p.z, q.x, q.y, q.z, q.w);
    }
    // the polyhedral cubes rest like the boxes beside them
    for (int i = 0; i < 3; ++i) {
        const float yp = registry.get<edyn::position>(bodies[1 + i]).y, yb = registry.get<edyn::position>(bodies[4 + i]).y;
        REQUIRE(std::fabs(yp - (0.5f + i)) < 0.03f && std::fabs(yb - (0.5f + i)) < 0.03f);
    }
    REQUIRE(registry.get<edyn::position>(bodies[9]).y > 2.9f);      // the wedge stayed on top of the stack
    REQUIRE(registry.get<edyn::position>(bodies[8]).y < 0.35f);     // the sphere left the wedge for the floor
    for (size_t i = 10; i < bodies.size(); ++i) REQUIRE(registry.get<edyn::position>(bodies[i]).y > 0.45f);
    edyn::detach(registry);
    std::printf("POLYHEDRA_OK\n");
    return 0;
}
