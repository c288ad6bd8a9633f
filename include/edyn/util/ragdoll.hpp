// edyn::make_ragdoll for the MI355X stepper shim - the reference's rag doll builder (include/edyn/util/ragdoll.hpp:10-128,
// src/edyn/util/ragdoll.cpp:16-914) written as tables of parts and joints over this shim's make_rigidbody / make_constraint /
// exclude_collision. The figure it builds is the reference's: 22 bodies (two shapeless fore-arm twist bodies, two shapeless
// shoulders), 36 constraints (cone + cvjoint pairs sharing an entity, hinges), 21 collision exclusions - checked field by field
// against the real engine's own rag doll (tests/golden/ragdoll_*.npz, exported from the engine) by tests/test_cpp_shim.py.
// box, capsule and cylinder figures (ragdoll_shape_type), as the reference builds them.
#ifndef EDYN_HIP_UTIL_RAGDOLL_HPP
#define EDYN_HIP_UTIL_RAGDOLL_HPP

#include <edyn/edyn.hpp>
#include <cmath>

namespace edyn {

enum class ragdoll_shape_type { box, capsule, cylinder };

struct ragdoll_simple_def {
    vector3 position{vector3_zero};
    quaternion orientation{quaternion_identity};
    scalar height{scalar(1.7)};
    scalar weight{scalar(72)};
    scalar restitution{0};
    scalar friction{scalar(0.5)};
    ragdoll_shape_type shape_type{ragdoll_shape_type::capsule};
};

struct ragdoll_def {
    vector3 position{vector3_zero};
    quaternion orientation{quaternion_identity};
    scalar head_mass, neck_mass, torso_upper_mass, torso_middle_mass, torso_lower_mass, hip_mass, leg_upper_mass, leg_lower_mass, foot_mass,
           shoulder_mass, arm_upper_mass, arm_lower_mass, hand_mass;
    vector3 head_size, neck_size, torso_upper_size, torso_middle_size, torso_lower_size, hip_size, leg_upper_size, leg_lower_size, foot_size,
            arm_upper_size, arm_lower_size, hand_size;
    scalar restitution{0};
    scalar friction{scalar(0.5)};
    ragdoll_shape_type shape_type{ragdoll_shape_type::capsule};
};

struct ragdoll_entities {
    entt::entity head, neck, torso_upper, torso_middle, torso_lower, hip;
    entt::entity leg_upper_left, leg_upper_right, leg_lower_left, leg_lower_right, foot_left, foot_right;
    entt::entity shoulder_left, shoulder_right, arm_upper_left, arm_upper_right, arm_lower_left, arm_lower_right;
    entt::entity arm_twist_left, arm_twist_right, hand_left, hand_right;
    entt::entity hip_torso_lower_constraint, torso_lower_torso_middle_constraint, torso_middle_torso_upper_constraint,
                 torso_upper_neck_constraint, neck_head_constraint;
    entt::entity hip_upper_leg_left_constraint, hip_upper_leg_right_constraint, knee_left_hinge, knee_right_hinge;
    entt::entity ankle_left_constraint, ankle_right_constraint;
    entt::entity torso_upper_shoulder_left_constraint, torso_upper_shoulder_right_constraint;
    entt::entity shoulder_arm_upper_left_constraint, shoulder_arm_upper_right_constraint;
    entt::entity elbow_left_hinge, elbow_right_hinge, arm_twist_left_hinge, arm_twist_right_hinge;
    entt::entity wrist_left_constraint, wrist_right_constraint;
};

namespace detail::rag {
constexpr scalar kPi = scalar(3.1415926535897932384626433832795029);
inline scalar rad(scalar degrees) { return degrees * kPi / scalar(180); }
inline scalar nm_per_rad(scalar nm_per_degree) { return nm_per_degree * (scalar(180) / kPi); }   // math.hpp:30 to_Nm_per_radian
inline vector3 operator+(vector3 a, vector3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vector3 operator*(vector3 a, scalar s) { return {a.x * s, a.y * s, a.z * s}; }
inline vector3 neg(vector3 a) { return {-a.x, -a.y, -a.z}; }
inline vector3 cross(vector3 a, vector3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline quaternion mul(quaternion a, quaternion b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline vector3 rotate(quaternion q, vector3 v) {
    const vector3 u{q.x, q.y, q.z};
    const vector3 t = cross(u, v) * scalar(2);
    return v + t * q.w + cross(u, t);
}
inline quaternion axis_angle(vector3 axis, scalar angle) {
    const scalar len = std::sqrt(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z), s = std::sin(angle / 2) / len;
    return {axis.x * s, axis.y * s, axis.z * s, std::cos(angle / 2)};
}
inline matrix3x3 columns(vector3 a, vector3 b, vector3 c) { return {{vector3{a.x, b.x, c.x}, vector3{a.y, b.y, c.y}, vector3{a.z, b.z, c.z}}}; }
inline matrix3x3 diagonal(vector3 d) { return {{vector3{d.x, 0, 0}, vector3{0, d.y, 0}, vector3{0, 0, d.z}}}; }
inline vector3 box_inertia(scalar mass, vector3 extents) {   // moment_of_inertia.cpp:11-17
    return vector3{extents.y * extents.y + extents.z * extents.z, extents.z * extents.z + extents.x * extents.x,
                   extents.x * extents.x + extents.y * extents.y} * (scalar(1) / scalar(12) * mass);
}
// solid capsule, as the reference computes it (moment_of_inertia.cpp:65-90: the cylinder term arrives permuted for the axis and is
// read as (axial, transverse) - reproduced, the device does the same: capi.hip k_init_bodies)
inline vector3 cylinder_inertia(scalar mass, scalar radius, scalar half_length, coordinate_axis axis) {   // moment_of_inertia.cpp:27-44
    const scalar len = half_length * 2;
    const scalar xx = scalar(0.5) * mass * radius * radius, yy = scalar(1) / scalar(12) * mass * (scalar(3) * radius * radius + len * len);
    return axis == coordinate_axis::x ? vector3{xx, yy, yy} : axis == coordinate_axis::y ? vector3{yy, xx, yy} : vector3{yy, yy, xx};
}
inline vector3 capsule_inertia(scalar mass, scalar radius, scalar half_length, coordinate_axis axis) {
    const scalar len = half_length * 2;
    const scalar cyl_vol = kPi * radius * radius * len, sph_vol = kPi * radius * radius * radius * scalar(4) / scalar(3);
    const scalar cyl_mass = mass * cyl_vol / (cyl_vol + sph_vol), sph_mass = mass * sph_vol / (cyl_vol + sph_vol);
    const scalar cyl_xx = scalar(0.5) * cyl_mass * radius * radius, cyl_yy = scalar(1) / scalar(12) * cyl_mass * (scalar(3) * radius * radius + len * len);
    const scalar sph = scalar(0.4) * sph_mass * radius * radius;
    const vector3 cyl = axis == coordinate_axis::x ? vector3{cyl_xx, cyl_yy, cyl_yy} : axis == coordinate_axis::y ? vector3{cyl_yy, cyl_xx, cyl_yy} : vector3{cyl_yy, cyl_yy, cyl_xx};
    const scalar t = scalar(4) * len + scalar(3) * radius;
    const scalar xx = sph + cyl.x, yy = sph + sph_mass * t * t / scalar(64) + cyl.y;
    return axis == coordinate_axis::x ? vector3{xx, yy, yy} : axis == coordinate_axis::y ? vector3{yy, xx, yy} : vector3{yy, yy, xx};
}
}  // namespace detail::rag

inline ragdoll_def make_ragdoll_def_from_simple(const ragdoll_simple_def &simple) {   // ragdoll.cpp:16-63
    using namespace detail::rag;
    ragdoll_def d{};
    d.position = simple.position; d.orientation = simple.orientation;
    d.friction = simple.friction; d.restitution = simple.restitution; d.shape_type = simple.shape_type;
    // body-part masses as 72nds of the weight
    const scalar w = simple.weight;
    d.head_mass = w * 4 / 72; d.neck_mass = w * 2 / 72; d.torso_upper_mass = w * 7 / 72; d.torso_middle_mass = w * 6 / 72;
    d.torso_lower_mass = w * 5 / 72; d.hip_mass = w * 3 / 72; d.leg_upper_mass = w * 8 / 72; d.leg_lower_mass = w * 7 / 72;
    d.foot_mass = w * 1 / 72; d.shoulder_mass = w * scalar(1.5) / 72; d.arm_upper_mass = w * scalar(2.5) / 72;
    d.arm_lower_mass = w * 2 / 72; d.hand_mass = w * scalar(0.5) / 72;
    // half sizes of a 1.70 m figure; width and depth scale at 80 % of the rate of the height
    const scalar v = simple.height / scalar(1.7), h = scalar(0.2) + v * scalar(0.8);
    auto sized = [](vector3 scale, vector3 half) { return vector3{scale.x * 2 * half.x, scale.y * 2 * half.y, scale.z * 2 * half.z}; };
    const vector3 up{h, v, h}, out{v, h, h};   // parts along the spine / legs, and the arms (held sideways)
    d.head_size = sized(up, {scalar(0.075), scalar(0.09), scalar(0.105)});
    d.neck_size = sized(up, {scalar(0.06), scalar(0.065), scalar(0.06)});
    d.torso_upper_size = sized(up, {scalar(0.17), scalar(0.108), scalar(0.095)});
    d.torso_middle_size = sized(up, {scalar(0.151), scalar(0.084), scalar(0.07)});
    d.torso_lower_size = sized(up, {scalar(0.155), scalar(0.065), scalar(0.086)});
    d.hip_size = sized(up, {scalar(0.17), scalar(0.07), scalar(0.1)});
    d.leg_upper_size = sized(up, {scalar(0.075), scalar(0.205), scalar(0.075)});
    d.leg_lower_size = sized(up, {scalar(0.06), scalar(0.205), scalar(0.06)});
    d.foot_size = sized(up, {scalar(0.05), scalar(0.04), scalar(0.13)});
    d.arm_upper_size = sized(out, {scalar(0.135), scalar(0.05), scalar(0.05)});
    d.arm_lower_size = sized(out, {scalar(0.135), scalar(0.04), scalar(0.04)});
    d.hand_size = sized(out, {scalar(0.065), scalar(0.045), scalar(0.045)});
    return d;
}

inline ragdoll_entities make_ragdoll(entt::registry &registry, const ragdoll_def &r) {   // ragdoll.cpp:69-914
    using namespace detail::rag;
    ragdoll_entities e{};
    const bool capsules = r.shape_type == ragdoll_shape_type::capsule, cylinders = r.shape_type == ragdoll_shape_type::cylinder;
    const quaternion turned = mul(r.orientation, axis_angle({0, 0, 1}, kPi));   // right arm: the same parts turned about z
    const vector3 X{1, 0, 0}, Y{0, 1, 0}, Z{0, 0, 1};

    // ---- bodies. `along` is the part's long axis (the capsule's axis); shapeless parts carry an explicit inertia.
    enum class shape_mode { shaped, shapeless_like_shape, shapeless_box };
    auto part = [&](scalar mass, vector3 size, coordinate_axis along, vector3 local, bool flip, shape_mode mode = shape_mode::shaped) {
        rigidbody_def def;
        def.material->restitution = r.restitution; def.material->friction = r.friction;
        def.mass = mass;
        def.position = r.position + rotate(r.orientation, local);
        def.orientation = flip ? turned : r.orientation;
        const scalar across = along == coordinate_axis::y ? size.x : along == coordinate_axis::x ? size.y : size.x;
        const scalar length = along == coordinate_axis::y ? size.y : along == coordinate_axis::x ? size.x : size.z;
        if (mode == shape_mode::shapeless_box) def.inertia = diagonal(box_inertia(mass, size));
        else if (mode == shape_mode::shapeless_like_shape)
            def.inertia = diagonal(capsules ? capsule_inertia(mass, across / 2, (length - across) / 2, along)
                                   : cylinders ? cylinder_inertia(mass, across / 2, length / 2, along) : box_inertia(mass, size));
        else if (capsules) def.shape = capsule_shape{across / 2, (length - across) / 2, along};
        else if (cylinders) def.shape = cylinder_shape{across / 2, length / 2, along};   // ragdoll.cpp:94-96 and the like
        else def.shape = box_shape{size * scalar(0.5)};
        return make_rigidbody(registry, def);
    };
    const scalar hip_top = r.hip_size.y / 2;
    const scalar y_tl = r.torso_lower_size.y, y_tm = r.torso_middle_size.y, y_tu = r.torso_upper_size.y;
    const coordinate_axis ax = coordinate_axis::x, ay = coordinate_axis::y, az = coordinate_axis::z;
    e.head = part(r.head_mass, r.head_size, ay, {0, r.head_size.y / 2 + r.neck_size.y * scalar(0.627) + y_tu + y_tm + y_tl + hip_top, scalar(-0.025)}, false);
    e.neck = part(r.neck_mass, r.neck_size, ay, {0, r.neck_size.y * scalar(0.627) / 2 + y_tu + y_tm + y_tl + hip_top, 0}, false);
    e.torso_upper = part(r.torso_upper_mass, r.torso_upper_size, ax, {0, y_tu / 2 + y_tm + y_tl + hip_top, 0}, false);
    e.torso_middle = part(r.torso_middle_mass, r.torso_middle_size, ax, {0, y_tm / 2 + y_tl + hip_top, 0}, false);
    e.torso_lower = part(r.torso_lower_mass, r.torso_lower_size, ax, {0, y_tl / 2 + hip_top, 0}, false);
    e.hip = part(r.hip_mass, r.hip_size, ax, {0, 0, 0}, false);
    const scalar leg_x = r.hip_size.x / 2 - (r.leg_upper_size.x - scalar(0.0072)) / 2;
    entt::entity *leg_upper[2] = {&e.leg_upper_left, &e.leg_upper_right}, *leg_lower[2] = {&e.leg_lower_left, &e.leg_lower_right};
    entt::entity *foot[2] = {&e.foot_left, &e.foot_right}, *shoulder[2] = {&e.shoulder_left, &e.shoulder_right};
    entt::entity *arm_upper[2] = {&e.arm_upper_left, &e.arm_upper_right}, *arm_lower[2] = {&e.arm_lower_left, &e.arm_lower_right};
    entt::entity *arm_twist[2] = {&e.arm_twist_left, &e.arm_twist_right}, *hand[2] = {&e.hand_left, &e.hand_right};
    const scalar sign[2] = {1, -1};   // left, right
    for (int i = 0; i < 2; ++i) *leg_upper[i] = part(r.leg_upper_mass, r.leg_upper_size, ay, {leg_x * sign[i], -r.leg_upper_size.y / 2, 0}, false);
    for (int i = 0; i < 2; ++i) *leg_lower[i] = part(r.leg_lower_mass, r.leg_lower_size, ay, {leg_x * sign[i], -(r.leg_upper_size.y + r.leg_lower_size.y / 2), 0}, false);
    for (int i = 0; i < 2; ++i)
        *foot[i] = part(r.foot_mass, r.foot_size, az, {leg_x * sign[i], -(r.leg_upper_size.y + r.leg_lower_size.y + r.foot_size.y / 2), -r.leg_lower_size.z / 2}, false);
    const scalar chest_top = y_tu + y_tm + y_tl + hip_top, arm_y = chest_top - r.arm_upper_size.y / 2, half_chest = r.torso_upper_size.x / 2;
    const vector3 shoulder_size{r.torso_upper_size.x * scalar(0.352), r.arm_upper_size.y, r.arm_upper_size.z};
    for (int i = 0; i < 2; ++i) *shoulder[i] = part(r.shoulder_mass, shoulder_size, ax, {half_chest * scalar(0.65) * sign[i], arm_y, 0}, i == 1, shape_mode::shapeless_box);
    for (int i = 0; i < 2; ++i) *arm_upper[i] = part(r.arm_upper_mass, r.arm_upper_size, ax, {(half_chest + r.arm_upper_size.x / 2) * sign[i], arm_y, 0}, i == 1);
    for (int i = 0; i < 2; ++i) {   // the fore-arm's mass is shared with a shapeless twin that carries the twist
        const vector3 at{(half_chest + r.arm_upper_size.x + r.arm_lower_size.x / 2) * sign[i], arm_y, 0};
        *arm_lower[i] = part(r.arm_lower_mass / 2, r.arm_lower_size, ax, at, i == 1);
        *arm_twist[i] = part(r.arm_lower_mass / 2, r.arm_lower_size, ax, at, i == 1, shape_mode::shapeless_like_shape);
    }
    for (int i = 0; i < 2; ++i)
        *hand[i] = part(r.hand_mass, r.hand_size, ax, {(half_chest + r.arm_upper_size.x + r.arm_lower_size.x + r.hand_size.x / 2) * sign[i],
                                                      arm_y - (r.hand_size.y - r.arm_lower_size.y) / 2, -(r.hand_size.z - r.arm_lower_size.z) / 2}, i == 1);

    // ---- neighbours along the skeleton do not collide (ragdoll.cpp:446-466)
    const std::array<entt::entity, 2> no_contact[] = {
        {e.hip, e.torso_lower}, {e.torso_middle, e.torso_lower}, {e.torso_middle, e.torso_upper}, {e.neck, e.torso_upper}, {e.neck, e.head},
        {e.hip, e.leg_upper_left}, {e.hip, e.leg_upper_right}, {e.torso_lower, e.leg_upper_left}, {e.torso_lower, e.leg_upper_right},
        {e.leg_lower_left, e.leg_upper_left}, {e.leg_lower_right, e.leg_upper_right}, {e.leg_lower_left, e.foot_left}, {e.leg_lower_right, e.foot_right},
        {e.torso_upper, e.shoulder_left}, {e.torso_upper, e.shoulder_right}, {e.torso_upper, e.arm_upper_left}, {e.torso_upper, e.arm_upper_right},
        {e.arm_lower_left, e.arm_upper_left}, {e.arm_lower_right, e.arm_upper_right}, {e.arm_lower_left, e.hand_left}, {e.arm_lower_right, e.hand_right}};
    for (const auto &pair : no_contact) exclude_collision(registry, pair[0], pair[1]);

    // ---- joints: a cone (swing limit with a bump stop) and a cvjoint (pivot, twist limit, friction) on ONE entity
    struct swing { vector3 pivot_a, cone_pivot_b; matrix3x3 cone_frame; scalar span_deg[2], stop_stiffness, stop_length; };
    struct twist { vector3 pivot_b; matrix3x3 frame_a, frame_b; scalar limit_deg[2]; bool limited; scalar stop_deg, twist_friction, bend_friction, damping_twist, damping_bend; };
    auto ball = [&](entt::entity a, entt::entity b, const swing &s, const twist &t) {
        const entt::entity con = registry.create();
        make_constraint<cone_constraint>(registry, con, a, b, [&](cone_constraint &c) {
            c.pivot = {s.pivot_a, s.cone_pivot_b}; c.frame = s.cone_frame;
            c.span_tan = {std::tan(rad(s.span_deg[0])), std::tan(rad(s.span_deg[1]))};
            c.bump_stop_stiffness = s.stop_stiffness; c.bump_stop_length = s.stop_length;
        });
        make_constraint<cvjoint_constraint>(registry, con, a, b, [&](cvjoint_constraint &c) {
            c.pivot = {s.pivot_a, t.pivot_b}; c.frame = {t.frame_a, t.frame_b};
            if (t.limited) {
                c.twist_min = rad(t.limit_deg[0]); c.twist_max = rad(t.limit_deg[1]);
                c.twist_bump_stop_angle = rad(t.stop_deg); c.twist_bump_stop_stiffness = nm_per_rad(5);
            }
            c.twist_friction_torque = nm_per_rad(t.twist_friction); c.bend_friction_torque = nm_per_rad(t.bend_friction);
            c.twist_damping = nm_per_rad(t.damping_twist); c.bend_damping = nm_per_rad(t.damping_bend);
        });
        return con;
    };
    auto limited = [](vector3 pivot_b, matrix3x3 fa, matrix3x3 fb, scalar lo, scalar hi, scalar stop_deg = 4) {
        return twist{pivot_b, fa, fb, {lo, hi}, true, stop_deg, scalar(0.02), scalar(0.02), scalar(0.2), scalar(0.2)};
    };
    auto loose = [](vector3 pivot_b, matrix3x3 fa, matrix3x3 fb, scalar friction, scalar damping) {
        return twist{pivot_b, fa, fb, {0, 0}, false, 0, 0, friction, 0, damping};
    };
    // the spine: cones open upwards, twist about the vertical
    const matrix3x3 upward = columns(Y, neg(X), Z);
    struct vertebra { entt::entity *con, a, b; scalar a_height, b_height, b_cone_z, cv_pivot_y, cv_pivot_z, span[2], stiffness, twist_deg; };
    const vertebra spine[] = {
        {&e.hip_torso_lower_constraint, e.hip, e.torso_lower, r.hip_size.y, y_tl, 0, -y_tl / 2, 0, {10, 20}, 5000, 12},
        {&e.torso_lower_torso_middle_constraint, e.torso_lower, e.torso_middle, y_tl, y_tm, 0, -y_tm / 2, 0, {16, 30}, 5000, 18},
        {&e.torso_middle_torso_upper_constraint, e.torso_middle, e.torso_upper, y_tm, y_tu, 0, -y_tu / 2, 0, {18, 32}, 5000, 10},
        {&e.torso_upper_neck_constraint, e.torso_upper, e.neck, y_tu, r.neck_size.y, 0, -r.neck_size.y * scalar(0.33), 0, {16, 32}, 3000, 30},
        {&e.neck_head_constraint, e.neck, e.head, r.neck_size.y, r.head_size.y, scalar(0.025), -(r.head_size.y / 2 - r.neck_size.y * scalar(0.2)), scalar(0.025), {16, 32}, 5000, 30}};
    for (const vertebra &v : spine)
        *v.con = ball(v.a, v.b, swing{{0, v.a_height / 2, 0}, {0, v.b_height, v.b_cone_z}, upward, {v.span[0], v.span[1]}, v.stiffness, scalar(0.05)},
                      limited({0, v.cv_pivot_y, v.cv_pivot_z}, upward, upward, -v.twist_deg, v.twist_deg));
    // hips: the cone points down, forward and a little outwards
    const matrix3x3 downward = columns(Y, X, neg(Z));
    entt::entity *hip_joint[2] = {&e.hip_upper_leg_left_constraint, &e.hip_upper_leg_right_constraint};
    for (int i = 0; i < 2; ++i) {
        const quaternion q = mul(axis_angle(X, rad(50)), axis_angle(Z, rad(10 * sign[i])));
        *hip_joint[i] = ball(e.hip, *leg_upper[i],
                             swing{{sign[i] * leg_x, 0, 0}, {0, -r.leg_upper_size.y, 0}, columns(rotate(q, neg(Y)), rotate(q, X), rotate(q, neg(Z))), {45, 70}, 5000, scalar(0.05)},
                             limited({0, r.leg_upper_size.y / 2, 0}, downward, downward, i == 0 ? -80 : -15, i == 0 ? 15 : 80));
    }
    auto hinge = [&](entt::entity a, entt::entity b, vector3 pa, vector3 pb, vector3 axis, scalar lo, scalar hi, scalar damping, scalar torque, scalar stop_stiffness) {
        return make_constraint<hinge_constraint>(registry, a, b, [&](hinge_constraint &h) {
            h.pivot = {pa, pb}; h.set_axes(axis, axis);
            h.angle_min = lo; h.angle_max = hi; h.damping = damping; h.torque = torque;
            h.bump_stop_angle = rad(10); h.bump_stop_stiffness = stop_stiffness;
        });
    };
    entt::entity *knee[2] = {&e.knee_left_hinge, &e.knee_right_hinge};
    for (int i = 0; i < 2; ++i)
        *knee[i] = hinge(*leg_upper[i], *leg_lower[i], {0, -r.leg_upper_size.y / 2, 0}, {0, r.leg_lower_size.y / 2, 0}, X, rad(-140), 0, 2, 1, 30);
    entt::entity *ankle[2] = {&e.ankle_left_constraint, &e.ankle_right_constraint};
    for (int i = 0; i < 2; ++i) {
        const quaternion q = axis_angle(X, rad(5));
        *ankle[i] = ball(*leg_lower[i], *foot[i],
                         swing{{0, -r.leg_lower_size.y / 2, 0}, {0, -r.foot_size.y, 0}, columns(rotate(q, neg(Y)), rotate(q, neg(X)), rotate(q, Z)), {24, 50}, 3000, scalar(0.03)},
                         loose({0, r.foot_size.y / 2, r.leg_lower_size.z / 2}, downward, downward, scalar(0.005), scalar(0.05)));
    }
    // shoulder girdle and arms
    entt::entity *girdle[2] = {&e.torso_upper_shoulder_left_constraint, &e.torso_upper_shoulder_right_constraint};
    for (int i = 0; i < 2; ++i) {
        const scalar s = sign[i];
        const quaternion q = mul(axis_angle(Z, rad(15 * s)), axis_angle(Y, rad(15 * s)));
        const vector3 at{(half_chest * scalar(0.65) - shoulder_size.x / 2) * s, y_tu / 2 - r.arm_upper_size.y / 2, 0};
        *girdle[i] = ball(e.torso_upper, *shoulder[i],
                          swing{at, {shoulder_size.x, 0, 0}, columns(rotate(q, X * s), rotate(q, Y * s), rotate(q, Z)), {30, 40}, 3000, scalar(0.03)},
                          limited({-shoulder_size.x / 2, 0, 0}, columns(X * s, Y * s, Z), matrix3x3_identity, -5, 5, 2));
    }
    entt::entity *arm_joint[2] = {&e.shoulder_arm_upper_left_constraint, &e.shoulder_arm_upper_right_constraint};
    for (int i = 0; i < 2; ++i) {
        const quaternion q = axis_angle({0, 1, -sign[i]}, rad(45));
        *arm_joint[i] = ball(*shoulder[i], *arm_upper[i],
                             swing{{shoulder_size.x / 2, 0, 0}, {r.arm_upper_size.x, 0, 0}, columns(rotate(q, X), rotate(q, Y), rotate(q, Z)), {45, 45}, 3000, scalar(0.03)},
                             limited({-r.arm_upper_size.x / 2, 0, 0}, matrix3x3_identity, matrix3x3_identity, -45, 45));
    }
    entt::entity *elbow[2] = {&e.elbow_left_hinge, &e.elbow_right_hinge}, *forearm[2] = {&e.arm_twist_left_hinge, &e.arm_twist_right_hinge};
    for (int i = 0; i < 2; ++i)
        *elbow[i] = hinge(*arm_upper[i], *arm_lower[i], {r.arm_upper_size.x / 2, 0, 0}, {-r.arm_lower_size.x / 2, 0, 0}, Y, 0, rad(140), scalar(0.1), scalar(0.02), nm_per_rad(5));
    for (int i = 0; i < 2; ++i)
        *forearm[i] = hinge(*arm_lower[i], *arm_twist[i], {0, 0, 0}, {0, 0, 0}, X, -kPi / 2, kPi / 2, scalar(0.1), scalar(0.02), nm_per_rad(5));
    entt::entity *wrist[2] = {&e.wrist_left_constraint, &e.wrist_right_constraint};
    for (int i = 0; i < 2; ++i)
        *wrist[i] = ball(*arm_twist[i], *hand[i],
                         swing{{r.arm_lower_size.x / 2, 0, 0}, {r.hand_size.x, 0, 0}, matrix3x3_identity, {80, 30}, 2000, scalar(0.03)},
                         loose({-r.hand_size.x / 2, (r.hand_size.y - r.arm_lower_size.y) / 2, (r.hand_size.z - r.arm_lower_size.z) / 2},
                               matrix3x3_identity, matrix3x3_identity, scalar(0.004), scalar(0.02)));
    return e;
}

inline ragdoll_entities make_ragdoll(entt::registry &registry, const ragdoll_simple_def &def) {   // ragdoll.cpp:65-67
    return make_ragdoll(registry, make_ragdoll_def_from_simple(def));
}

}  // namespace edyn

#endif
