// Device-side fp32 vector math for the gfx950 kernels.
// Expression order matches the reference's math headers (include/edyn/math/vector3.hpp,
// quaternion.hpp, matrix3x3.hpp, transform.hpp) so that, built with -ffp-contract=off and IEEE
// divide/sqrt, the kernels round exactly like the reference's scalar C++.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#define DI __device__ __forceinline__

namespace dm {

constexpr float kEps = FLT_EPSILON;
constexpr float kScalarMax = FLT_MAX;
constexpr float kLarge = 1e18f;
constexpr float kPi = 3.1415926535897932384626433832795029f;   // math/constants.hpp:10-12
constexpr float kPi2 = kPi * 2.0f;
constexpr float kHalfSqrt2 = 0.7071067811865475244008443621048490f;

struct f3 {
    float x, y, z;
    DI float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }          // (used by the cylinder routines, dcylinder.hpp)
    DI float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct f2 { float x, y; };
struct q4 { float x, y, z, w; };
struct m3 { f3 r0, r1, r2; };

DI f3 mk3(float x, float y, float z) { return {x, y, z}; }
DI f3 from4(const float4 &v) { return {v.x, v.y, v.z}; }
DI q4 q_from4(const float4 &v) { return {v.x, v.y, v.z, v.w}; }
DI float4 to4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
DI float4 to4(q4 q) { return make_float4(q.x, q.y, q.z, q.w); }

DI f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DI f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DI f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
DI f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DI f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DI f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
DI f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
DI f3 &operator+=(f3 &a, f3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
DI f3 &operator*=(f3 &a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
DI f3 div_recip(f3 a, float s) { float z = 1.0f / s; return {a.x * z, a.y * z, a.z * z}; }   // vector3 operator/=
DI bool eq(f3 a, f3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DI f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DI float length_sqr(f3 a) { return dot(a, a); }
DI float length(f3 a) { return sqrtf(length_sqr(a)); }
DI f3 normalize(f3 a) { return a / length(a); }   // vector3.hpp:231-235 (true division)
DI float distance_sqr(f3 a, f3 b) { return length_sqr(a - b); }
DI f3 project_plane(f3 p, f3 q, f3 n) { return p - n * dot(p - q, n); }
DI f3 lerp(f3 a, f3 b, float s) { return a * (1.0f - s) + b * s; }
DI float clamp_unit(float s) { return fminf(fmaxf(s, 0.0f), 1.0f); }
DI float square(float s) { return s * s; }
DI float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
DI bool try_normalize(f3 &v) {
    float l2 = length_sqr(v);
    if ((double)l2 > 1e-18) { v = div_recip(v, sqrtf(l2)); return true; }
    return false;
}
DI f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
DI f2 operator-(f2 a) { return {-a.x, -a.y}; }

DI q4 operator*(q4 q, q4 r) {
    return {q.w * r.x + q.x * r.w + q.y * r.z - q.z * r.y,
            q.w * r.y + q.y * r.w + q.z * r.x - q.x * r.z,
            q.w * r.z + q.z * r.w + q.x * r.y - q.y * r.x,
            q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z};
}
DI q4 operator*(q4 q, float s) { return {q.x * s, q.y * s, q.z * s, q.w * s}; }
DI q4 operator/(q4 q, float s) { return {q.x / s, q.y / s, q.z / s, q.w / s}; }
DI q4 operator+(q4 a, q4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
DI float length_sqr(q4 q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
DI q4 normalize(q4 q) { return q / sqrtf(length_sqr(q)); }
DI q4 conjugate(q4 q) { return {-q.x, -q.y, -q.z, q.w}; }
DI f3 rotate(q4 q, f3 v) {
    f3 r{q.x, q.y, q.z};
    return v + cross(2.0f * r, cross(r, v) + q.w * v);
}
// Correctly rounded fp32 sin/cos (double evaluation, one rounding): libm-independent, so the device agrees bit for
// bit with a CPU evaluation of the same formula (the reference's std::sin(float) depends on the C library's last bit).
DI float sin_cr(float x) { return (float)sin((double)x); }
DI float cos_cr(float x) { return (float)cos((double)x); }
// src/edyn/math/quaternion.cpp:7-22
DI q4 integrate(q4 q, f3 w, float dt) {
    const float ws = length(w);
    float t;
    if (ws < 0.001f) {
        const float k = 1.0f / 48.0f;
        t = 0.5f * dt - dt * dt * dt * k * ws * ws;
    } else {
        t = sin_cr(0.5f * ws * dt) / ws;
    }
    q4 r{w.x * t, w.y * t, w.z * t, cos_cr(0.5f * ws * dt)};
    return normalize(r * q);
}
DI q4 quaternion_derivative(q4 q, f3 w) { return (q4{w.x, w.y, w.z, 0.0f} * q) * 0.5f; }

DI float col_dot(const m3 &m, int i, f3 v) {
    return comp(m.r0, i) * v.x + comp(m.r1, i) * v.y + comp(m.r2, i) * v.z;
}
DI f3 mul(const m3 &m, f3 v) { return {dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)}; }       // m * v
DI f3 mul(f3 v, const m3 &m) { return {col_dot(m, 0, v), col_dot(m, 1, v), col_dot(m, 2, v)}; }   // v * m
DI m3 mul(const m3 &m, const m3 &n) {
    return {{col_dot(n, 0, m.r0), col_dot(n, 1, m.r0), col_dot(n, 2, m.r0)},
            {col_dot(n, 0, m.r1), col_dot(n, 1, m.r1), col_dot(n, 2, m.r1)},
            {col_dot(n, 0, m.r2), col_dot(n, 1, m.r2), col_dot(n, 2, m.r2)}};
}
DI m3 transpose(const m3 &m) { return {{m.r0.x, m.r1.x, m.r2.x}, {m.r0.y, m.r1.y, m.r2.y}, {m.r0.z, m.r1.z, m.r2.z}}; }
DI m3 m3_columns(f3 a, f3 b, f3 c) { return {{a.x, b.x, c.x}, {a.y, b.y, c.y}, {a.z, b.z, c.z}}; }
DI m3 m3_zero() { return {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; }
DI m3 to_m3(q4 q) {
    float d = length_sqr(q);
    float s = 2 / d;
    float xs = q.x * s, ys = q.y * s, zs = q.z * s;
    float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    return {{1 - (yy + zz), xy - wz, xz + wy}, {xy + wz, 1 - (xx + zz), yz - wx}, {xz - wy, yz + wx, 1 - (xx + yy)}};
}
DI f3 to_world(f3 p, f3 pos, q4 orn) { return pos + rotate(orn, p); }
DI f3 to_object(f3 p, f3 pos, q4 orn) { return rotate(conjugate(orn), p - pos); }
DI f3 to_object(f3 p, f3 pos, const m3 &basis) { return mul(p - pos, basis); }

// AABB predicates (include/edyn/comp/aabb.hpp:16-18, src/edyn/math/geom.cpp:762-770)
struct box3 { f3 mn, mx; };
DI box3 inset(const box3 &b, float v) { return {{b.mn.x + v, b.mn.y + v, b.mn.z + v}, {b.mx.x - v, b.mx.y - v, b.mx.z - v}}; }
DI bool intersect(const box3 &a, const box3 &b) {
    return (a.mn.x <= b.mx.x) && (a.mx.x >= b.mn.x) && (a.mn.y <= b.mx.y) && (a.mx.y >= b.mn.y) &&
           (a.mn.z <= b.mx.z) && (a.mx.z >= b.mn.z);
}

DI void plane_space(f3 n, f3 &p, f3 &q) {   // src/edyn/math/geom.cpp:730-754
    if (fabsf(n.z) > kHalfSqrt2) {
        float a = n.y * n.y + n.z * n.z;
        float k = 1.0f / sqrtf(a);
        p.x = 0; p.y = -n.z * k; p.z = n.y * k;
        q.x = a * k; q.y = -n.x * p.z; q.z = n.x * p.y;
    } else {
        float a = n.x * n.x + n.y * n.y;
        float k = 1.0f / sqrtf(a);
        p.x = -n.y * k; p.y = n.x * k; p.z = 0;
        q.x = -n.z * p.y; q.y = n.z * p.x; q.z = a * k;
    }
}

}  // namespace dm
