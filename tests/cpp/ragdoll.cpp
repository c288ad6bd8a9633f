// edyn::make_ragdoll through the shim: dumps the figure it builds (every body and constraint as registry components) for
// tests/test_cpp_shim.py to compare with the real engine's own rag doll (tests/golden/ragdoll_*.npz), and - with a GPU
// (argument "run") - drops it on a floor. Usage: ragdoll capsule|box|cylinder [run]
#include <edyn/util/ragdoll.hpp>
#include <cstdio>
#include <cstring>
#include <cmath>

static void v3(const char *tag, const edyn::vector3 &v) { std::printf(" %s %.9g %.9g %.9g", tag, v.x, v.y, v.z); }
static void m3(const char *tag, const edyn::matrix3x3 &m) {
    std::printf(" %s", tag);
    for (int r = 0; r < 3; ++r) std::printf(" %.9g %.9g %.9g", m.row[r].x, m.row[r].y, m.row[r].z);
}

int main(int argc, char **argv) {
    const bool box = argc > 1 && !std::strcmp(argv[1], "box");
    const bool cyl = argc > 1 && !std::strcmp(argv[1], "cylinder");
    const bool run = argc > 2 && !std::strcmp(argv[2], "run");
    entt::registry registry;
    edyn::attach(registry);
    if (run) {
        edyn::rigidbody_def floor;
        floor.kind = edyn::rigidbody_kind::rb_static;
        floor.shape = edyn::plane_shape{{0, 1, 0}, 0};
        edyn::make_rigidbody(registry, floor);
    }
    edyn::ragdoll_simple_def def;
    def.shape_type = box ? edyn::ragdoll_shape_type::box : cyl ? edyn::ragdoll_shape_type::cylinder : edyn::ragdoll_shape_type::capsule;
    if (run) def.position = {0, 1.3f, 0};
    const edyn::ragdoll_entities rag = edyn::make_ragdoll(registry, def);
    auto &s = registry.ctx().get<edyn::detail::gpu_stepper>();
    const uint32_t first = run ? 1u : 0u;
    if (!run) {
        for (uint32_t i = first; i < s.bodies.size(); ++i) {
            const entt::entity e = s.bodies[i];
            std::printf("body %u mass %.9g", i - first, registry.get<edyn::mass>(e).s);
            v3("pos", registry.get<edyn::position>(e));
            const auto &q = registry.get<edyn::orientation>(e);
            std::printf(" orn %.9g %.9g %.9g %.9g", q.x, q.y, q.z, q.w);
            if (auto *b = registry.try_get<edyn::box_shape>(e)) { std::printf(" shape 1"); v3("param", b->half_extents); }
            else if (auto *c = registry.try_get<edyn::capsule_shape>(e)) std::printf(" shape 4 param %.9g %.9g %d", c->radius, c->half_length, (int)c->axis);
            else if (auto *y = registry.try_get<edyn::cylinder_shape>(e)) std::printf(" shape 5 param %.9g %.9g %d", y->radius, y->half_length, (int)y->axis);
            else { std::printf(" shape 0 param 0 0 0"); m3("inertia", registry.get<edyn::inertia>(e)); }
            const auto &mat = registry.get<edyn::material>(e);
            std::printf(" friction %.9g restitution %.9g\n", mat.friction, mat.restitution);
        }
        for (uint32_t j = 0; j < s.constraints.size(); ++j) {
            const entt::entity e = s.constraints[j];
            const int kind = s.constraint_kind[j];
            const edyn::constraint_base *cb = edyn::detail::constraint_of(registry, e, kind);
            std::printf("joint %u kind %d bodies %u %u", j, kind, registry.get<edyn::detail::body_index>(cb->body[0]).value - first,
                        registry.get<edyn::detail::body_index>(cb->body[1]).value - first);
            if (kind == EDYNHIP_JOINT_HINGE) {
                const auto &h = registry.get<edyn::hinge_constraint>(e);
                v3("pivotA", h.pivot[0]); v3("pivotB", h.pivot[1]); v3("axisA", h.axis[0]); v3("axisB", h.axis[1]);
                std::printf(" p %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g", h.angle_min, h.angle_max, h.limit_restitution, h.bump_stop_angle,
                            h.bump_stop_stiffness, h.torque, h.speed, h.rest_angle, h.stiffness, h.damping);
            } else if (kind == EDYNHIP_JOINT_CONE) {
                const auto &c = registry.get<edyn::cone_constraint>(e);
                v3("pivotA", c.pivot[0]); v3("pivotB", c.pivot[1]); m3("frameA", c.frame);
                std::printf(" p %.9g %.9g %.9g %.9g %.9g", c.span_tan[0], c.span_tan[1], c.restitution, c.bump_stop_stiffness, c.bump_stop_length);
            } else {
                const auto &c = registry.get<edyn::cvjoint_constraint>(e);
                v3("pivotA", c.pivot[0]); v3("pivotB", c.pivot[1]); m3("frameA", c.frame[0]); m3("frameB", c.frame[1]);
                std::printf(" p %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g", c.twist_min, c.twist_max, c.twist_restitution,
                            c.twist_bump_stop_angle, c.twist_bump_stop_stiffness, c.twist_friction_torque, c.twist_rest_angle, c.twist_stiffness, c.twist_damping,
                            c.rest_direction.x, c.rest_direction.y, c.rest_direction.z, c.bend_stiffness, c.bend_friction_torque, c.bend_damping);
            }
            std::printf("\n");
        }
        for (auto &x : s.exclusions) std::printf("exclude %u %u\n", x[0] - first, x[1] - first);
        std::printf("entities %d %d\n", (int)(rag.head != rag.hand_right), (int)(rag.wrist_right_constraint != rag.hip_torso_lower_constraint));
        std::printf("RAGDOLL_DUMP_OK\n");
        return 0;
    }
    // drop it: one second of simulation, every part ends up lying on the floor, finite and still connected
    for (int i = 1; i <= 120; ++i) edyn::update(registry, i / 60.0);
    float top = 0; bool finite = true;
    for (uint32_t i = first; i < s.bodies.size(); ++i) {
        const auto &p = registry.get<edyn::position>(s.bodies[i]);
        finite = finite && std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z);
        top = std::fmax(top, p.y);
    }
    const auto &head = registry.get<edyn::position>(rag.head), &neck = registry.get<edyn::position>(rag.neck);
    const float gap = std::sqrt((head.x - neck.x) * (head.x - neck.x) + (head.y - neck.y) * (head.y - neck.y) + (head.z - neck.z) * (head.z - neck.z));
    std::printf("top %.4f head-neck %.4f finite %d\n", top, gap, (int)finite);
    if (!finite || top > 0.7f || top < 0.03f || gap > 0.25f) { std::printf("RAGDOLL_RUN_FAILED\n"); return 1; }
    std::printf("RAGDOLL_RUN_OK\n");
    return 0;
}
