// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by or called from the product
// path (edyn_amd/, include/edynhip.h). Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it.
//
// Scalar fp32 restatement of the reference's math primitives. Operation order follows the
// reference expression by expression so that float rounding is identical (build with
// -ffp-contract=off, no fast-math).
//   vec3   : /root/reference/include/edyn/math/vector3.hpp:12-320
//   quat   : /root/reference/include/edyn/math/quaternion.hpp:9-248, src/edyn/math/quaternion.cpp:7-22
//   mat3   : /root/reference/include/edyn/math/matrix3x3.hpp:12-265
//   xform  : /root/reference/include/edyn/math/transform.hpp
#pragma once
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <algorithm>

namespace orc {

constexpr float kEps = FLT_EPSILON;          // EDYN_EPSILON, math/scalar.hpp:17
constexpr float kScalarMax = FLT_MAX;        // EDYN_SCALAR_MAX
constexpr float kLarge = 1e18f;              // large_scalar, math/constants.hpp:17
constexpr float kGravitationalConstant = 6.674e-11f;   // math/constants.hpp
constexpr float kPi = 3.1415926535897932384626433832795029f;
constexpr float kPi2 = kPi * 2.0f;
constexpr float kHalfSqrt2 = 0.7071067811865475244008443621048490f;

struct vec3 {
    float x, y, z;
    float &operator[](size_t i) { return (&x)[i]; }
    float operator[](size_t i) const { return (&x)[i]; }
};
struct vec2 { float x, y; };

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 &operator+=(vec3 &a, vec3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline vec3 &operator-=(vec3 &a, vec3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
inline vec3 &operator*=(vec3 &a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
// vector3.hpp:112-118: v /= s multiplies by the reciprocal (NOT three divisions).
inline vec3 &operator/=(vec3 &a, float s) { float z = 1.0f / s; a.x *= z; a.y *= z; a.z *= z; return a; }
inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(vec3 a, vec3 b) { return !(a == b); }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length_sqr(vec3 a) { return dot(a, a); }
inline float length(vec3 a) { return std::sqrt(length_sqr(a)); }
inline float distance_sqr(vec3 a, vec3 b) { return length_sqr(a - b); }
inline vec3 normalize(vec3 a) { return a / length(a); }
inline bool try_normalize(vec3 &v) {   // vector3.hpp:233-241
    float l2 = length_sqr(v);
    if (l2 > 1e-18) { v /= std::sqrt(l2); return true; }
    return false;
}
inline vec3 project_plane(vec3 p, vec3 q, vec3 n) { return p - n * dot(p - q, n); }
inline vec3 vmin(vec3 a, vec3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline vec3 vmax(vec3 a, vec3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline vec3 vabs(vec3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline size_t max_index(vec3 v) {      // vector3.hpp:293-305
    float m = v.x; size_t i = 0;
    if (v.y > m) { m = v.y; i = 1; }
    if (v.z > m) { i = 2; }
    return i;
}
inline size_t max_index_abs(vec3 v) { return max_index(vabs(v)); }
inline vec3 lerp(vec3 a, vec3 b, float s) { return a * (1.0f - s) + b * s; }   // math.hpp:68-71
inline float clamp_unit(float s) { return std::min(std::max(s, 0.0f), 1.0f); }
inline float square(float s) { return s * s; }

inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator-(vec2 a) { return {-a.x, -a.y}; }

struct quat { float x, y, z, w; };
inline quat operator*(quat q, quat r) {
    return {q.w * r.x + q.x * r.w + q.y * r.z - q.z * r.y,
            q.w * r.y + q.y * r.w + q.z * r.x - q.x * r.z,
            q.w * r.z + q.z * r.w + q.x * r.y - q.y * r.x,
            q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z};
}
inline quat operator*(quat q, float s) { return {q.x * s, q.y * s, q.z * s, q.w * s}; }
inline quat operator/(quat q, float s) { return {q.x / s, q.y / s, q.z / s, q.w / s}; }
inline quat operator+(quat a, quat b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline float length_sqr(quat q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
inline quat normalize(quat q) { return q / std::sqrt(length_sqr(q)); }
inline quat conjugate(quat q) { return {-q.x, -q.y, -q.z, q.w}; }
// quaternion.hpp:145-149
inline vec3 rotate(quat q, vec3 v) {
    vec3 r{q.x, q.y, q.z};
    return v + cross(2.0f * r, cross(r, v) + q.w * v);
}
// sin/cos: the reference calls std::sin/std::cos on floats, whose last bit depends on the C library. Both the
// oracle and the GPU evaluate them in double precision and round once, i.e. correctly rounded fp32 (equal to
// glibc's sinf/cosf except in rare 1-ulp cases), so that CPU and GPU agree bit for bit over long horizons.
// g_libm_trig (test switch, default off): evaluate them exactly as the reference does - std::sin/std::cos on float,
// i.e. the C library's sinf/cosf - so that the restatement can be compared bit for bit with the real engine
// (oracle/_ref) in scenes that spin; the default remains the library-independent, correctly rounded value.
inline bool g_libm_trig = false;
inline float sin_cr(float x) { return g_libm_trig ? std::sin(x) : (float)std::sin((double)x); }
inline float cos_cr(float x) { return g_libm_trig ? std::cos(x) : (float)std::cos((double)x); }
// quaternion.cpp:7-22 (exponential map; Taylor branch for |w| < 0.001)
inline quat integrate(quat q, vec3 w, float dt) {
    const float ws = length(w);
    const float half = 0.5f;
    float t;
    if (ws < 0.001f) {
        const float k = 1.0f / 48.0f;
        t = half * dt - dt * dt * dt * k * ws * ws;
    } else {
        t = sin_cr(half * ws * dt) / ws;
    }
    quat r{w.x * t, w.y * t, w.z * t, cos_cr(half * ws * dt)};
    return normalize(r * q);
}
// quaternion.hpp:244-246: quaternion{w,0} * q * 0.5
inline quat quaternion_derivative(quat q, vec3 w) { return (quat{w.x, w.y, w.z, 0.0f} * q) * 0.5f; }

struct mat3 {
    vec3 row[3];
    vec3 &operator[](size_t i) { return row[i]; }
    const vec3 &operator[](size_t i) const { return row[i]; }
    vec3 column(size_t i) const { return {row[0][i], row[1][i], row[2][i]}; }
    float column_dot(size_t i, vec3 v) const { return row[0][i] * v.x + row[1][i] * v.y + row[2][i] * v.z; }
};
constexpr mat3 kMat3Zero{{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
constexpr mat3 kMat3Identity{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
inline mat3 operator*(const mat3 &m, const mat3 &n) {   // matrix3x3.hpp:61-67
    return {{{n.column_dot(0, m.row[0]), n.column_dot(1, m.row[0]), n.column_dot(2, m.row[0])},
             {n.column_dot(0, m.row[1]), n.column_dot(1, m.row[1]), n.column_dot(2, m.row[1])},
             {n.column_dot(0, m.row[2]), n.column_dot(1, m.row[2]), n.column_dot(2, m.row[2])}}};
}
inline vec3 operator*(const mat3 &m, vec3 v) { return {dot(m.row[0], v), dot(m.row[1], v), dot(m.row[2], v)}; }
inline vec3 operator*(vec3 v, const mat3 &m) { return {m.column_dot(0, v), m.column_dot(1, v), m.column_dot(2, v)}; }
inline mat3 transpose(const mat3 &m) { return {{m.column(0), m.column(1), m.column(2)}}; }
inline mat3 mat3_columns(vec3 a, vec3 b, vec3 c) { return {{{a.x, b.x, c.x}, {a.y, b.y, c.y}, {a.z, b.z, c.z}}}; }
inline mat3 diagonal(vec3 v) { return {{{v.x, 0, 0}, {0, v.y, 0}, {0, 0, v.z}}}; }
inline mat3 skew(vec3 v) { return {{{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}}}; }
inline mat3 to_mat3(quat q) {   // matrix3x3.hpp:252-265
    float d = length_sqr(q);
    float s = 2 / d;
    float xs = q.x * s, ys = q.y * s, zs = q.z * s;
    float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    return {{{1 - (yy + zz), xy - wz, xz + wy},
             {xy + wz, 1 - (xx + zz), yz - wx},
             {xz - wy, yz + wx, 1 - (xx + yy)}}};
}
inline mat3 inverse_symmetric(const mat3 &m) {   // matrix3x3.hpp:190-218
    float det = dot(m.row[0], cross(m.row[1], m.row[2]));
    float di = 1.0f / det;
    float a11 = m[0][0], a12 = m[0][1], a13 = m[0][2];
    float a22 = m[1][1], a23 = m[1][2];
    float a33 = m[2][2];
    mat3 r{};
    r[0][0] = di * (a22 * a33 - a23 * a23);
    r[0][1] = di * (a13 * a23 - a12 * a33);
    r[0][2] = di * (a12 * a23 - a13 * a22);
    r[1][0] = r[0][1];
    r[1][1] = di * (a11 * a33 - a13 * a13);
    r[1][2] = di * (a12 * a13 - a11 * a23);
    r[2][0] = r[0][2];
    r[2][1] = r[1][2];
    r[2][2] = di * (a11 * a22 - a12 * a12);
    return r;
}

inline vec3 to_world(vec3 p, vec3 pos, quat orn) { return pos + rotate(orn, p); }
inline vec3 to_object(vec3 p, vec3 pos, quat orn) { return rotate(conjugate(orn), p - pos); }
inline vec3 to_object(vec3 p, vec3 pos, const mat3 &basis) { return (p - pos) * basis; }
inline vec3 to_world(vec3 p, vec3 pos, const mat3 &basis) { return pos + basis * p; }   // transform.hpp:33-35

struct aabb {
    vec3 min, max;
    aabb inset(vec3 v) const { return {min + v, max - v}; }   // comp/aabb.hpp:16-18
    bool contains(vec3 p) const {
        return min.x <= p.x && min.y <= p.y && min.z <= p.z && p.x <= max.x && p.y <= max.y && p.z <= max.z;
    }
    bool contains(const aabb &b) const { return contains(b.min) && contains(b.max); }
    float area() const {
        vec3 d = max - min;
        return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
    }
};
inline bool intersect(const aabb &a, const aabb &b) {   // geom.cpp:762-770
    return (a.min.x <= b.max.x) && (a.max.x >= b.min.x) && (a.min.y <= b.max.y) && (a.max.y >= b.min.y) &&
           (a.min.z <= b.max.z) && (a.max.z >= b.min.z);
}
inline aabb enclosing(const aabb &a, const aabb &b) { return {vmin(a.min, b.min), vmax(a.max, b.max)}; }

}  // namespace orc
