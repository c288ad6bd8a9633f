// Code written against the reference includes its headers piecemeal; every one of those paths resolves to the shim.
#include <edyn/config/execution_mode.hpp>
#include <edyn/math/vector3.hpp>
#include <edyn/math/quaternion.hpp>
#include <edyn/math/matrix3x3.hpp>
#include <edyn/comp/position.hpp>
#include <edyn/comp/orientation.hpp>
#include <edyn/comp/linvel.hpp>
#include <edyn/comp/material.hpp>
#include <edyn/comp/tag.hpp>
#include <edyn/shapes/box_shape.hpp>
#include <edyn/shapes/capsule_shape.hpp>
#include <edyn/constraints/hinge_constraint.hpp>
#include <edyn/constraints/cvjoint_constraint.hpp>
#include <edyn/collision/contact_manifold.hpp>
#include <edyn/collision/should_collide.hpp>
#include <edyn/util/rigidbody.hpp>
#include <edyn/util/constraint_util.hpp>
#include <edyn/util/exclude_collision.hpp>
#include <edyn/util/gravity_util.hpp>
#include <edyn/util/insert_material_mixing.hpp>
#include <edyn/util/ragdoll.hpp>
#include <edyn/edyn.hpp>
#include <cstdio>

int main() {
    entt::registry registry;
    edyn::attach(registry);
    auto def = edyn::rigidbody_def{};
    def.shape = edyn::capsule_shape{0.2f, 0.3f, edyn::coordinate_axis::y};
    const auto a = edyn::make_rigidbody(registry, def), b = edyn::make_rigidbody(registry, def);
    edyn::make_constraint<edyn::hinge_constraint>(registry, a, b, [](edyn::hinge_constraint &h) { h.set_axes({1, 0, 0}, {1, 0, 0}); });
    edyn::exclude_collision(registry, a, b);
    edyn::set_gravity(registry, {0, -1.62f, 0});
    std::printf("INCLUDES_OK %d\n", (int)registry.all_of<edyn::position, edyn::orientation, edyn::linvel>(a));
    return 0;
}
