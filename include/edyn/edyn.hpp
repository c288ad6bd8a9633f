// edyn.hpp — drop-in C++ shim that keeps Edyn's registry-facing stepping API on top of the MI355X C-ABI.
//
// Same names, argument meaning and ownership as the reference:
//   edyn::init_config, attach, detach, update, step_simulation, set_paused, is_paused, get/set_fixed_dt,
//   set_max_steps_per_update, get/set_solver_*_iterations, get/set_gravity            include/edyn/edyn.hpp:39-186,
//                                                                                      config/solver_iteration_config.hpp:13-66, util/gravity_util.hpp:15-23
//   edyn::rigidbody_def, rigidbody_kind, make_rigidbody                                include/edyn/util/rigidbody.hpp:22-93
//   edyn::make_constraint<point_constraint|hinge_constraint>(registry, [entity,] body0, body1, setup...)   include/edyn/util/constraint_util.hpp:38-54
//   edyn::exclude_collision, remove_collision_exclusion, clear_rigidbody                include/edyn/util/exclude_collision.hpp:20-47, util/rigidbody.hpp:95-103
//   edyn::set_should_collide, should_collide_default                                   include/edyn/collision/should_collide.hpp:8-18 (host predicate: slow path)
//   registry.destroy(body / constraint entity) on a running world                      noticed at the next update (island_manager.cpp:47-115 semantics)
//   components: position, orientation, linvel, angvel, mass, mass_inv, inertia, material, box_shape, sphere_shape,
//               plane_shape, AABB, dynamic_tag / kinematic_tag / static_tag, rigidbody_tag, contact_manifold (read-only view)
//
// Semantics: the registry stays caller-owned; like the reference's sequential stepper, all mutation happens inside
// update()/step_simulation() on the calling thread. Bodies and constraints created since the last step are uploaded
// at the next step; after a step position/orientation/linvel/angvel of every rigid body are written back. Direct
// writes by the user to those four components are picked up after edyn::refresh(registry) (full re-upload of the
// state) — the analogue of registry.patch in the reference's asynchronous mode. Contact manifolds are materialised
// lazily by edyn::get_contact_manifolds(registry). Uses the real EnTT when <entt/entt.hpp> is available, otherwise the
// bundled minimal registry (include/edyn/detail/mini_entt.hpp).
#pragma once
#if __has_include(<entt/entt.hpp>)
#include <entt/entt.hpp>
#else
#include "detail/mini_entt.hpp"
#endif
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <variant>
#include <vector>
#include "../edynhip.h"

namespace edyn {

using scalar = float;
struct vector3 { scalar x{}, y{}, z{}; };
struct quaternion { scalar x{}, y{}, z{}, w{1}; };
struct matrix3x3 { std::array<vector3, 3> row{}; };
inline constexpr vector3 vector3_zero{0, 0, 0};
inline constexpr vector3 gravity_earth{0, scalar(-9.8), 0};   // math/constants.hpp:24
inline constexpr quaternion quaternion_identity{0, 0, 0, 1};

// components (comp/*.hpp) — thin wrappers, as in the reference
struct position : vector3 {};
struct orientation : quaternion {};
struct linvel : vector3 {};
struct angvel : vector3 {};
struct present_position : vector3 {};      // comp/present_position.hpp: what a renderer should draw (interpolated)
struct present_orientation : quaternion {};
struct mass { scalar s; };
struct mass_inv { scalar s; };
struct inertia : matrix3x3 {};
struct gravity : vector3 {};
struct center_of_mass : vector3 {};   // comp/center_of_mass.hpp: offset of the centre of mass in the shape's frame
struct origin : vector3 {};           // comp/origin.hpp: where the shape sits; `position` is the centre of mass (kept in sync after every update)
inline constexpr scalar large_scalar = scalar(1e18);   // math/constants.hpp:17
struct material_base {   // comp/material.hpp:15-22; the last four select contact_extras_constraint (rolling / spinning friction, soft contacts)
    scalar restitution{0}, friction{scalar(0.5)}, spin_friction{0}, roll_friction{0}, stiffness{large_scalar}, damping{large_scalar};
};
struct material : material_base {   // :27-31: optional identifier for the material mix table
    using id_type = uint16_t;
    static constexpr id_type UnassignedID = 0xFFFF;
    id_type id{UnassignedID};
};
struct AABB { vector3 min, max; };
struct dynamic_tag {};
struct kinematic_tag {};
struct static_tag {};
struct procedural_tag {};
struct rigidbody_tag {};
struct sleeping_tag {};            // comp/tag.hpp: the body's island is asleep (kept in sync after every update)
struct sleeping_disabled_tag {};   // comp/tag.hpp
struct collision_filter { uint64_t group{~0ull}, mask{~0ull}; };

struct box_shape { vector3 half_extents; };
struct sphere_shape { scalar radius; };
struct plane_shape { vector3 normal; scalar constant; };
enum class coordinate_axis : unsigned char { x, y, z };                                       // math/coordinate_axis.hpp
struct capsule_shape { scalar radius; scalar half_length; coordinate_axis axis{coordinate_axis::x}; };   // shapes/capsule_shape.hpp:17-30
struct cylinder_shape { scalar radius; scalar half_length; coordinate_axis axis{coordinate_axis::x}; };  // shapes/cylinder_shape.hpp:22-25
/// shapes/convex_mesh.hpp:17-70: vertices, the faces' vertex indices (counter-clockwise seen from outside), faces = (first index, count)
/// pairs. initialize() moves the vertices so that the centroid is the origin (convex_mesh.cpp:32-38, shape_util.cpp:351-391); face normals,
/// edges, adjacency and the relevant faces / edges are derived on upload (edynhip_create_convex_mesh) and live with the device context.
struct convex_mesh {
    std::vector<vector3> vertices;
    std::vector<uint32_t> indices, faces;
    bool initialized{false};
    size_t num_faces() const { return faces.size() / 2; }
    void initialize() {
        if (initialized) return;
        vector3 c{0, 0, 0};
        scalar volume = 0;
        for (size_t f = 0; f < num_faces(); ++f) {
            const uint32_t first = faces[2 * f], count = faces[2 * f + 1];
            const vector3 v0 = vertices[indices[first]];
            for (uint32_t j = 1; j + 1 < count; ++j) {
                const vector3 v1 = vertices[indices[first + j]], v2 = vertices[indices[first + j + 1]];
                const vector3 a{v1.x - v0.x, v1.y - v0.y, v1.z - v0.z}, b{v2.x - v1.x, v2.y - v1.y, v2.z - v1.z};
                const vector3 n{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
                volume += v0.x * n.x + v0.y * n.y + v0.z * n.z;
                const vector3 vx{v0.x + v1.x, v1.x + v2.x, v2.x + v0.x}, vy{v0.y + v1.y, v1.y + v2.y, v2.y + v0.y}, vz{v0.z + v1.z, v1.z + v2.z, v2.z + v0.z};
                c.x += n.x * (vx.x * vx.x + vx.y * vx.y + vx.z * vx.z);
                c.y += n.y * (vy.x * vy.x + vy.y * vy.y + vy.z * vy.z);
                c.z += n.z * (vz.x * vz.x + vz.y * vz.y + vz.z * vz.z);
            }
        }
        volume /= 6;
        const scalar z = scalar(1) / (24 * 2 * volume);
        c = {c.x * z, c.y * z, c.z * z};
        for (auto &v : vertices) v = {v.x - c.x, v.y - c.y, v.z - c.z};
        initialized = true;
    }
};
struct polyhedron_shape {   // shapes/polyhedron_shape.hpp:11-43 (the rotated mesh is the device's business)
    std::shared_ptr<convex_mesh> mesh;
    polyhedron_shape() = default;
    polyhedron_shape(std::shared_ptr<convex_mesh> m) : mesh(std::move(m)) {}
};
/// util/shape_util.hpp make_box_mesh (shape_util.cpp:12-38)
inline void make_box_mesh(const vector3 &he, std::vector<vector3> &vertices, std::vector<uint32_t> &indices, std::vector<uint32_t> &faces) {
    const scalar x = he.x, y = he.y, z = he.z;
    const vector3 v[8] = {{-x, -y, -z}, {x, -y, -z}, {x, -y, z}, {-x, -y, z}, {-x, y, -z}, {x, y, -z}, {x, y, z}, {-x, y, z}};
    vertices.insert(vertices.end(), v, v + 8);
    const uint32_t idx[24] = {0, 1, 2, 3, 7, 6, 5, 4, 4, 5, 1, 0, 6, 7, 3, 2, 7, 4, 0, 3, 5, 6, 2, 1};
    indices.insert(indices.end(), idx, idx + 24);
    for (uint32_t f = 0; f < 6; ++f) { faces.push_back(4 * f); faces.push_back(4); }
}
using shapes_variant_t = std::variant<box_shape, sphere_shape, plane_shape, capsule_shape, cylinder_shape, polyhedron_shape>;

enum class rigidbody_kind : uint8_t { rb_dynamic, rb_kinematic, rb_static };   // util/rigidbody.hpp:22-27

struct rigidbody_def {   // util/rigidbody.hpp:29-81 (hot-path fields)
    rigidbody_kind kind{rigidbody_kind::rb_dynamic};
    vector3 position{vector3_zero};
    quaternion orientation{quaternion_identity};
    scalar mass{1};
    std::optional<matrix3x3> inertia;
    vector3 linvel{vector3_zero};
    vector3 angvel{vector3_zero};
    std::optional<vector3> center_of_mass;   // (position is then the origin, as in the reference: util/rigidbody.cpp:85-87)
    std::optional<vector3> gravity;
    std::optional<shapes_variant_t> shape;
    std::optional<edyn::material> material{edyn::material{}};
    uint64_t collision_group{~0ull};
    uint64_t collision_mask{~0ull};
    bool sleeping_disabled{false};
    bool presentation{true};
};

struct constraint_base { std::array<entt::entity, 2> body; };
struct point_constraint : constraint_base {                                                   // constraints/point_constraint.hpp:21-34
    std::array<vector3, 2> pivot;
    scalar friction_torque{};
};
struct distance_constraint : constraint_base {                                                // constraints/distance_constraint.hpp
    std::array<vector3, 2> pivot;
    scalar distance{0};
};
struct soft_distance_constraint : constraint_base {                                           // constraints/soft_distance_constraint.hpp
    std::array<vector3, 2> pivot;
    scalar distance{0}, stiffness{scalar(1e10)}, damping{scalar(1e10)};
};
struct generic_constraint : constraint_base {                                                 // constraints/generic_constraint.hpp:18-76
    struct linear_dof {
        bool limit_enabled{true};
        scalar offset_min{}, offset_max{}, limit_restitution{}, bump_stop_length{}, bump_stop_stiffness{}, friction_force{}, rest_offset{},
               spring_stiffness{}, damping{};
    };
    struct angular_dof {
        bool limit_enabled{true};
        scalar angle_min{}, angle_max{}, limit_restitution{}, bump_stop_angle{}, bump_stop_stiffness{}, friction_torque{}, rest_angle{},
               spring_stiffness{}, damping{};
    };
    std::array<vector3, 2> pivot;
    std::array<matrix3x3, 2> frame{matrix3x3{{vector3{1, 0, 0}, vector3{0, 1, 0}, vector3{0, 0, 1}}}, matrix3x3{{vector3{1, 0, 0}, vector3{0, 1, 0}, vector3{0, 0, 1}}}};
    std::array<linear_dof, 3> linear_dofs;
    std::array<angular_dof, 3> angular_dofs;
};
struct null_constraint : constraint_base {};                                                  // constraints/null_constraint.hpp: no rows, one island
struct gravity_constraint : constraint_base {};                                               // constraints/gravity_constraint.hpp (Newtonian attraction)
inline constexpr matrix3x3 matrix3x3_identity{{vector3{1, 0, 0}, vector3{0, 1, 0}, vector3{0, 0, 1}}};
struct cone_constraint : constraint_base {                                                    // constraints/cone_constraint.hpp:19-49
    std::array<vector3, 2> pivot;
    matrix3x3 frame{matrix3x3_identity};   // in body 0; first column = the cone direction
    std::array<scalar, 2> span_tan{1, 1};
    scalar restitution{}, bump_stop_stiffness{}, bump_stop_length{};
};
struct cvjoint_constraint : constraint_base {                                                 // constraints/cvjoint_constraint.hpp:20-102
    std::array<vector3, 2> pivot;
    std::array<matrix3x3, 2> frame{matrix3x3_identity, matrix3x3_identity};   // first column = the twist axis
    scalar twist_min{}, twist_max{}, twist_restitution{}, twist_bump_stop_angle{}, twist_bump_stop_stiffness{}, twist_friction_torque{},
           twist_rest_angle{}, twist_stiffness{}, twist_damping{};
    vector3 rest_direction{};
    scalar bend_stiffness{}, bend_friction_torque{}, bend_damping{};
};
struct hinge_constraint : constraint_base {                                                   // constraints/hinge_constraint.hpp:22-93
    std::array<vector3, 2> pivot;
    std::array<vector3, 2> axis{vector3{1, 0, 0}, vector3{1, 0, 0}};
    scalar angle_min{}, angle_max{}, limit_restitution{};     // limits are active when angle_min < angle_max
    scalar bump_stop_angle{}, bump_stop_stiffness{};
    scalar torque{}, speed{};
    scalar rest_angle{}, stiffness{}, damping{};
    void set_axes(const vector3 &axisA, const vector3 &axisB) { axis = {axisA, axisB}; }
};
struct contact_manifold { std::array<entt::entity, 2> body; unsigned num_points; };           // collision/contact_manifold.hpp:14-22
// Contact points are entities of their own, as in the reference (collision/contact_point.hpp:17-58): created and destroyed
// in the registry as the device reports them (init_config::materialize_contacts), so registry.view<contact_point>() and,
// with EnTT, on_construct / on_destroy<contact_point> work as they do against the reference.
struct contact_point { vector3 pivotA, pivotB, normal; };
struct contact_point_geometry { vector3 local_normal; scalar distance{0}; int normal_attachment{0}; };
struct contact_point_impulse { scalar normal_impulse{0}; std::array<scalar, 2> friction_impulse{0, 0}; };
struct contact_point_list { entt::entity parent; uint64_t id; };   // parent manifold; id: the device's point id

enum class execution_mode : uint8_t { sequential, sequential_multithreaded, asynchronous };
struct init_config {   // edyn.hpp:39-60 + settings.hpp:21-57
    // Host threads that carry the device's results into the registry (the write-back of position / orientation / linvel / angvel /
    // present_* / origin is a parallel loop over the bodies). The reference's field of the same name sizes its job dispatcher
    // (edyn.hpp:39-42: 0 = hardware_concurrency); here 0 = min(16, hardware_concurrency / 2), 1 = no threads are started.
    size_t num_worker_threads{0};
    edyn::execution_mode execution_mode{execution_mode::sequential};   // asynchronous: the registry receives each update's result during the next one (below)
    scalar fixed_dt{scalar(1.0 / 60)};
    unsigned num_solver_velocity_iterations{8};
    unsigned num_solver_position_iterations{3};
    unsigned max_steps_per_update{10};
    vector3 gravity{gravity_earth};
    int device{0};
    // More than one entry: ONE simulation over these GPUs at island granularity (edynhip.h "Multi-GPU world": one shard per device,
    // islands partitioned by load, re-partitioned when islands of different shards meet) - the reference's island parallelism
    // (solver.cpp:408-428) across devices. Bodies of every shape, every constraint type, exclusions and settings are supported;
    // contact entities, contact_extras materials / the mix table, asynchronous mode and step callbacks are single-device features
    // (attach throws stepper_error EDYNHIP_ERR_UNSUPPORTED when combined); an edit of a running multi-device world (bodies made or
    // destroyed, edyn::refresh) rebuilds the world from the registry - correct, but the contacts' warm start is lost.
    std::vector<int> devices{};
    unsigned max_bodies{0};      // 0 = sized at the first upload (count + 25 % head-room)
    unsigned max_manifolds{0};
    bool island_sleeping{true};  // the reference always sleeps islands (bodies opt out with sleeping_disabled)
    // contact_manifold / contact_point entities in the registry, kept in step with the device through its event list.
    // contact_point_data: also refresh every live point's pivots / normal / distance / impulses after each update (a full
    // read-back of the manifolds; off = call edyn::refresh_contact_points(registry) when the data is needed).
    bool materialize_contacts{true};
    bool contact_point_data{false};
    // Contact arithmetic (edynhip.h EDYNHIP_FLAG_FUSED_VELOCITY_ROWS / EDYNHIP_FLAG_BLOCK_POSITION). Default: every contact row and every
    // contact position correction with the reference's operations in the reference's order; the two switches opt in to faster forms
    // of the same equations (stated deviations: DESIGN.md section 4).
    bool fused_velocity_rows{false};
    bool block_position{false};
    // EDYNHIP_FLAG_EXCLUSIVE_DEVICE: a promise that this stepper is the only user of its GPU while it steps - the resident-grid
    // solver kernels are then launched plainly instead of cooperatively (edynhip.h). Off = always safe.
    bool exclusive_device{false};
};

/// Host time the shim itself spent around the device calls, accumulated over updates (edyn::get_shim_timings / reset_shim_timings):
/// what edyn::update costs beyond edynhip_step. Milliseconds; `step_call` is the host time inside edynhip_step(_timed) - the part
/// of the GPU step during which the host waits for the step's counters -, `state_wait` the wait for the rest of the step + the copy.
struct shim_timings {
    double sync_removed{0}, upload{0}, step_call{0}, state_wait{0}, write_back{0}, contacts{0}, presentation{0}, total{0};
    uint64_t updates{0}, steps{0}, contact_events{0};   // contact_events: manifold / point creations and destructions carried into the registry
};

class stepper_error : public std::runtime_error {
public:
    stepper_error(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};

namespace detail {
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
// A small fork-join pool for the registry write-back (the reference's counterpart: the job dispatcher's parallel_for, which its
// multithreaded stepper uses for the same kind of per-entity loops). parallel_for cuts [0, count) into chunks of `grain`; the
// caller and the workers take chunks from one atomic ticket until none is left, so a worker that wakes up late costs nothing -
// the others have done its share. Workers spin briefly for the next job, then sleep; prewake() tells sleeping workers that a job
// is about to follow (they then spin for it for up to ~3 ms instead of being woken when it is already there).
class worker_pool {
public:
    explicit worker_pool(unsigned threads) {
        for (unsigned k = 1; k < threads; ++k) workers_.emplace_back([this, k] { loop(k); });
    }
    ~worker_pool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    worker_pool(const worker_pool &) = delete;
    worker_pool &operator=(const worker_pool &) = delete;
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    void prewake() {
        if (workers_.empty()) return;
        // (seq_cst on both sides of the sleepers_ / hint_ and sleepers_ / ticket_ pairs: "publish, then look for sleepers" against "register as a
        //  sleeper, then look for work" is a store-load pattern - with release / acquire alone both sides may miss each other and a worker sleeps through a job)
        hint_.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers_.load(std::memory_order_seq_cst)) { { std::lock_guard<std::mutex> l(m_); } cv_.notify_all(); }
    }
    /// f(begin, end, worker) for every chunk; worker in [0, size()); returns when every chunk is done
    template <typename F>
    void parallel_for(uint32_t count, uint32_t grain, F &&f) {
        if (count == 0) return;
        const uint32_t chunks = (count + grain - 1) / grain;
        if (workers_.empty() || chunks == 1) { f(0u, count, 0u); return; }
        using Fn = std::remove_reference_t<F>;
        run_ = [](void *ctx, uint32_t b, uint32_t e, unsigned w) { (*static_cast<Fn *>(ctx))(b, e, w); };
        ctx_ = &f; count_ = count; grain_ = grain; chunks_.store(chunks, std::memory_order_relaxed);
        remaining_.store(chunks, std::memory_order_relaxed);
        const uint64_t seq = ++seq_;
        ticket_.store(seq << 32, std::memory_order_seq_cst);   // opens the job: (sequence, next chunk)
        if (sleepers_.load(std::memory_order_seq_cst)) { { std::lock_guard<std::mutex> l(m_); } cv_.notify_all(); }
        take_chunks(seq, 0u);
        while (remaining_.load(std::memory_order_acquire) != 0) cpu_relax();
        ticket_.store((seq << 32) | 0xFFFFFFFFull, std::memory_order_release);   // closed: a straggler finds no chunk and no stale field is read
    }
private:
    void take_chunks(uint64_t seq, unsigned worker) {
        uint64_t t = ticket_.load(std::memory_order_acquire);
        for (;;) {
            if ((t >> 32) != seq) return;
            const uint32_t c = (uint32_t)t;
            if (c == 0xFFFFFFFFu || c >= chunks_.load(std::memory_order_relaxed)) return;   // (the job's fields change only between a close and the next open)
            if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
            const uint32_t b = c * grain_;
            run_(ctx_, b, std::min(count_, b + grain_), worker);
            remaining_.fetch_sub(1, std::memory_order_acq_rel);
            t = ticket_.load(std::memory_order_acquire);
        }
    }
    void loop(unsigned worker) {
        uint64_t seen = 0, seen_hint = 0;
        auto spin_until = std::chrono::steady_clock::now();
        for (;;) {
            const uint64_t t = ticket_.load(std::memory_order_acquire);
            if ((t >> 32) != seen && (uint32_t)t != 0xFFFFFFFFu) {
                seen = t >> 32;
                take_chunks(seen, worker);
                spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(50);
                continue;
            }
            const uint64_t h = hint_.load(std::memory_order_acquire);
            if (h != seen_hint) { seen_hint = h; spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(3000); }
            if (std::chrono::steady_clock::now() < spin_until) { for (int k = 0; k < 32; ++k) cpu_relax(); continue; }
            std::unique_lock<std::mutex> l(m_);
            if (stop_) return;
            sleepers_.fetch_add(1, std::memory_order_seq_cst);
            cv_.wait(l, [&] {
                const uint64_t now = ticket_.load(std::memory_order_seq_cst);
                return stop_ || ((now >> 32) != seen && (uint32_t)now != 0xFFFFFFFFu) || hint_.load(std::memory_order_seq_cst) != seen_hint;
            });
            sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            if (stop_) return;
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_; std::condition_variable cv_; bool stop_{false};
    std::atomic<uint64_t> ticket_{0xFFFFFFFFull}, hint_{0};
    std::atomic<uint32_t> remaining_{0}, sleepers_{0};
    uint64_t seq_{0};
    void (*run_)(void *, uint32_t, uint32_t, unsigned){nullptr}; void *ctx_{nullptr}; uint32_t count_{0}, grain_{1}; std::atomic<uint32_t> chunks_{0};
};

// uint64 key -> entity, open addressing with linear probing and backward-shift deletion: the contact entities' look-up tables (half a
// million live points on the headline pile, a thousand insertions and removals per update) - one cache line per operation and no node
// allocations, where std::unordered_map pays a malloc / free and two dependent misses each. The interface is the subset of
// std::unordered_map the shim uses (find / end / erase(iterator) / operator[] / clear / range-for over `first`, `second`).
class entity_table {
public:
    struct slot { uint64_t first; entt::entity second; };
    class iterator {
    public:
        iterator(slot *p, slot *e) : p_(p), e_(e) { skip(); }
        slot &operator*() const { return *p_; }
        slot *operator->() const { return p_; }
        iterator &operator++() { ++p_; skip(); return *this; }
        bool operator!=(const iterator &o) const { return p_ != o.p_; }
        bool operator==(const iterator &o) const { return p_ == o.p_; }
    private:
        friend class entity_table;
        void skip() { while (p_ != e_ && p_->first == empty_key) ++p_; }
        slot *p_, *e_;
    };
    iterator begin() { return iterator(slots_.data(), slots_.data() + slots_.size()); }
    iterator end() { return iterator(slots_.data() + slots_.size(), slots_.data() + slots_.size()); }
    size_t size() const { return count_; }
    void clear() { slots_.clear(); count_ = 0; }
    iterator find(uint64_t key) {
        if (slots_.empty()) return end();
        const uint64_t k = key + 1;   // 0 marks an empty slot (no point id is 2^64 - 1)
        for (size_t i = home(k);; i = (i + 1) & mask()) {
            if (slots_[i].first == k) return at(i);
            if (slots_[i].first == empty_key) return end();
        }
    }
    entt::entity &operator[](uint64_t key) {
        if ((count_ + 1) * 4 > slots_.size() * 3) grow();
        const uint64_t k = key + 1;
        for (size_t i = home(k);; i = (i + 1) & mask()) {
            if (slots_[i].first == k) return slots_[i].second;
            if (slots_[i].first == empty_key) { slots_[i].first = k; slots_[i].second = entt::null; ++count_; return slots_[i].second; }
        }
    }
    void erase(iterator it) {   // backward-shift deletion: no tombstones, probe sequences stay short however long the table lives
        size_t i = (size_t)(it.p_ - slots_.data());
        for (size_t j = (i + 1) & mask();; j = (j + 1) & mask()) {
            if (slots_[j].first == empty_key) break;
            const size_t h = home(slots_[j].first);
            if (((j - h) & mask()) >= ((j - i) & mask())) { slots_[i] = slots_[j]; i = j; }   // j's element may move back to i: i lies on its probe path
        }
        slots_[i].first = empty_key;
        --count_;
    }
    /// the key a slot was stored under (`first` holds key + 1)
    static uint64_t key_of(const slot &s) { return s.first - 1; }
private:
    static constexpr uint64_t empty_key = 0;
    size_t mask() const { return slots_.size() - 1; }
    size_t home(uint64_t k) const { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return (size_t)k & mask(); }
    iterator at(size_t i) { iterator it(slots_.data() + slots_.size(), slots_.data() + slots_.size()); it.p_ = slots_.data() + i; return it; }
    void grow() {
        std::vector<slot> old;
        old.swap(slots_);
        slots_.assign(old.empty() ? 1024 : old.size() * 2, slot{empty_key, entt::null});
        count_ = 0;
        for (const slot &s : old) if (s.first != empty_key) (*this)[s.first - 1] = s.second;
    }
    std::vector<slot> slots_;
    size_t count_{0};
};

// The analogue of stepper_sequential in registry.ctx() (edyn.cpp:117-123).
struct gpu_stepper {
    init_config cfg;
    edynhip_ctx *ctx{nullptr};
    edynhip_world *world{nullptr};             // init_config::devices names more than one GPU: the multi-GPU world instead of `ctx`
    bool multi() const { return cfg.devices.size() > 1; }
    std::vector<entt::entity> bodies;          // body index -> entity (creation order); entt::null = destroyed (the index stays reserved)
    std::vector<entt::entity> constraints;     // joint index -> entity, same convention
    std::vector<uint8_t> constraint_kind;      // joint index -> EDYNHIP_JOINT_*: one entity may carry several constraint types (make_ragdoll: cone + cvjoint)
    std::vector<std::array<uint32_t, 2>> exclusions;   // every active exclude_collision pair (body indices): replayed into a re-created context
    bool (*should_collide)(const entt::registry &, entt::entity, entt::entity){nullptr};   // edyn::set_should_collide: the user's predicate (nullptr = should_collide_default on the device)
    entt::registry *filter_registry{nullptr};        // what the predicate is called with
    size_t exclusions_uploaded{0};                     // how many of them the current device context already holds (a prefix)
    std::vector<float> shadow;                         // asynchronous mode: the state this shim last wrote into the registry (13 floats per body) - what differs was edited by the user
    unsigned snapshot_bodies{0};                       // bodies the snapshot in flight covers
    bool scene_dirty{true}, state_dirty{false}, paused{false}, params_dirty{false};
    double accumulated{0}, last_time{0};
    unsigned capacity{0}, joint_capacity{0};
    unsigned uploaded_bodies{0}, uploaded_constraints{0};   // what the device context already holds
    entity_table manifold_entities;   // (body index A << 32 | body index B) -> contact_manifold entity
    entity_table point_entities;      // device point id -> contact_point entity
    double (*time_func)(){nullptr};                 // settings.time_func (edyn::set_time_source); nullptr = the monotonic clock
    void (*pre_step)(entt::registry &){nullptr};    // settings.pre_step_callback / post_step_callback (context/step_callback.hpp)
    void (*post_step)(entt::registry &){nullptr};
    struct mixing { uint32_t id0, id1; float v[6]; };
    std::vector<mixing> mixings;   // insert_material_mixing calls, replayed into a (re)created context
    bool refresh_friction{false};  // set_rigidbody_friction: the carried contact points take the bodies' current materials (rigidbody.cpp:324-350)
    std::vector<std::pair<std::shared_ptr<convex_mesh>, uint32_t>> meshes;   // convex meshes the current device context holds, with their ids
    std::vector<uint32_t> reshaped;   // bodies whose shape / kind changed: their contacts are detected afresh by the re-created context
    bool recreate{false};          // a body's mass / inertia / material was edited: the next upload re-creates the context (contacts, joints and sleep state are carried)
    bool contacts_resync{false};   // the context was re-created (capacity growth): point ids changed, rebuild the contact entities
    bool snapshot_pending{false};                                   // asynchronous mode: a snapshot of the previous update is in flight
    shim_timings tm;                                                // host time per phase (edyn::get_shim_timings)
    std::unique_ptr<worker_pool> pool;                              // the write-back's host threads (init_config::num_worker_threads)
    std::vector<std::vector<uint32_t>> sleep_changes;               // per worker: bodies whose sleeping flag differs from the registry's tag
    std::vector<uint8_t> asleep_shadow;                             // body index -> the registry carries sleeping_tag (as this shim left it)
    bool removal_pending{true};                                     // an on_destroy hook fired (or nothing is known yet): sync_removed has work
    bool records_pending{false};                                    // a record snapshot of the previous update is in flight (asynchronous mode)
    float present_dt{0};                                            // update_presentation's interpolation_dt of the update in progress
    bool prefetch_on{false};                                        // the context copies its contact events ahead of the step's state (edynhip_set_event_prefetch): what write_back reads
    bool present_this_call{true};                                   // edyn::update refreshes present_*; edyn::step_simulation does not (stepper_sequential.cpp:121-147 never calls update_presentation)
    bool hooks_connected{false};
    bool host_presentation{false};                                  // this update's presentation transforms were not part of a write-back: compute them on the host
    uint32_t events_stale_through{0};                               // asynchronous mode: snapshots up to this step index carry events a rebuild already covered
    ~gpu_stepper() { if (ctx) edynhip_destroy(ctx); if (world) edynhip_world_destroy(world); }
};
struct body_index { uint32_t value; };
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct phase_timer {   // adds the time between construction and destruction to one field of shim_timings
    double &acc; double t0;
    explicit phase_timer(double &a) : acc(a), t0(now_ms()) {}
    ~phase_timer() { acc += now_ms() - t0; }
};
/// edynhip_pair_filter -> the user's should_collide_func: body indices back to the entities they were made from
inline int pair_filter_trampoline(void *user, uint32_t body, uint32_t other) {
    auto &s = *static_cast<gpu_stepper *>(user);
    if (!s.should_collide || !s.filter_registry || body >= s.bodies.size() || other >= s.bodies.size()) return 1;
    return s.should_collide(*s.filter_registry, s.bodies[body], s.bodies[other]) ? 1 : 0;
}
template <typename T> constexpr int joint_kind_of() {
    if constexpr (std::is_same_v<T, point_constraint>) return EDYNHIP_JOINT_POINT;
    else if constexpr (std::is_same_v<T, hinge_constraint>) return EDYNHIP_JOINT_HINGE;
    else if constexpr (std::is_same_v<T, distance_constraint>) return EDYNHIP_JOINT_DISTANCE;
    else if constexpr (std::is_same_v<T, soft_distance_constraint>) return EDYNHIP_JOINT_SOFT_DISTANCE;
    else if constexpr (std::is_same_v<T, cone_constraint>) return EDYNHIP_JOINT_CONE;
    else if constexpr (std::is_same_v<T, cvjoint_constraint>) return EDYNHIP_JOINT_CVJOINT;
    else if constexpr (std::is_same_v<T, gravity_constraint>) return EDYNHIP_JOINT_GRAVITY;
    else if constexpr (std::is_same_v<T, null_constraint>) return EDYNHIP_JOINT_NULL;
    else return EDYNHIP_JOINT_GENERIC;
}
// the constraint component of kind `kind` on `e`, or nullptr (also when the entity itself is gone)
inline constraint_base *constraint_of(entt::registry &registry, entt::entity e, int kind) {
    if (!registry.valid(e)) return nullptr;
    switch (kind) {
    case EDYNHIP_JOINT_POINT: return registry.try_get<point_constraint>(e);
    case EDYNHIP_JOINT_HINGE: return registry.try_get<hinge_constraint>(e);
    case EDYNHIP_JOINT_DISTANCE: return registry.try_get<distance_constraint>(e);
    case EDYNHIP_JOINT_SOFT_DISTANCE: return registry.try_get<soft_distance_constraint>(e);
    case EDYNHIP_JOINT_CONE: return registry.try_get<cone_constraint>(e);
    case EDYNHIP_JOINT_CVJOINT: return registry.try_get<cvjoint_constraint>(e);
    case EDYNHIP_JOINT_GRAVITY: return registry.try_get<gravity_constraint>(e);
    case EDYNHIP_JOINT_NULL: return registry.try_get<null_constraint>(e);
    default: return registry.try_get<generic_constraint>(e);
    }
}

inline void check(gpu_stepper &s, int rc) {
    if (rc != EDYNHIP_OK) throw stepper_error(rc, std::string("edynhip: ") + (s.world ? edynhip_world_last_error(s.world) : edynhip_last_error(s.ctx)));
}

// Joint definitions [first, end) for edynhip_set_joints / edynhip_add_joints. A constraint destroyed before it was ever
// uploaded keeps its index: it goes up as a placeholder and is listed in `dead` so that the caller removes it again before
// any step runs.
inline void joint_arrays(entt::registry &registry, gpu_stepper &s, uint32_t first, std::vector<int32_t> &jt, std::vector<uint32_t> &jb,
                         std::vector<float> &jp, std::vector<float> &ja, std::vector<float> &jq, std::vector<uint32_t> &dead) {
    const uint32_t nj = (uint32_t)s.constraints.size() - first;
    jt.assign(nj, EDYNHIP_JOINT_POINT); jb.assign(2 * nj, 0); jp.assign(6 * nj, 0.f); ja.assign(6 * nj, 0.f); jq.assign(10 * nj, 0.f);
    for (uint32_t j = 0; j < nj; ++j) {
        const entt::entity e = s.constraints[first + j];
        if (e == entt::null) { dead.push_back(first + j); continue; }
        auto fill = [&](const constraint_base &cb, const std::array<vector3, 2> &pv) {
            for (int k = 0; k < 2; ++k) {
                jb[2 * j + k] = registry.get<body_index>(cb.body[k]).value;
                jp[6 * j + 3 * k] = pv[k].x; jp[6 * j + 3 * k + 1] = pv[k].y; jp[6 * j + 3 * k + 2] = pv[k].z;
            }
        };
        const int kind = s.constraint_kind[first + j];
        if (kind == EDYNHIP_JOINT_POINT) { auto *pc = &registry.get<point_constraint>(e); jt[j] = kind; fill(*pc, pc->pivot); jq[10 * j] = pc->friction_torque; }
        else if (kind == EDYNHIP_JOINT_DISTANCE) { auto *dc = &registry.get<distance_constraint>(e); jt[j] = kind; fill(*dc, dc->pivot); jq[10 * j] = dc->distance; }
        else if (kind == EDYNHIP_JOINT_SOFT_DISTANCE) {
            auto *sc = &registry.get<soft_distance_constraint>(e);
            jt[j] = kind; fill(*sc, sc->pivot); jq[10 * j] = sc->distance; jq[10 * j + 1] = sc->stiffness; jq[10 * j + 2] = sc->damping;
        } else if (kind == EDYNHIP_JOINT_GRAVITY) { jt[j] = kind; fill(registry.get<gravity_constraint>(e), std::array<vector3, 2>{}); }
        else if (kind == EDYNHIP_JOINT_NULL) { jt[j] = kind; fill(registry.get<null_constraint>(e), std::array<vector3, 2>{}); }
        else if (kind == EDYNHIP_JOINT_GENERIC) { auto *ge = &registry.get<generic_constraint>(e); jt[j] = kind; fill(*ge, ge->pivot); }   // (definition follows)
        else if (kind == EDYNHIP_JOINT_CONE) { auto *cc = &registry.get<cone_constraint>(e); jt[j] = kind; fill(*cc, cc->pivot); }      // frames / parameters follow
        else if (kind == EDYNHIP_JOINT_CVJOINT) { auto *cv = &registry.get<cvjoint_constraint>(e); jt[j] = kind; fill(*cv, cv->pivot); }  // (define_frames below)
        else {
            auto &hc = registry.get<hinge_constraint>(e);
            jt[j] = EDYNHIP_JOINT_HINGE; fill(hc, hc.pivot);
            for (int k = 0; k < 2; ++k) { ja[6 * j + 3 * k] = hc.axis[k].x; ja[6 * j + 3 * k + 1] = hc.axis[k].y; ja[6 * j + 3 * k + 2] = hc.axis[k].z; }
            const float q[10] = {hc.angle_min, hc.angle_max, hc.limit_restitution, hc.bump_stop_angle, hc.bump_stop_stiffness, hc.torque, hc.speed,
                                 hc.rest_angle, hc.stiffness, hc.damping};
            for (int k = 0; k < 10; ++k) jq[10 * j + k] = q[k];
        }
    }
}

// The per-body arrays of edynhip_bodies for bodies [first, first + n) of the stepper's list, from the registry's components.
struct body_arrays {
    std::vector<int32_t> kind, stype;
    std::vector<float> pos, orn, lv, av, m, I, sp, fr, re, g, com, xspin, xroll, xstiff, xdamp;
    std::vector<uint8_t> hasI, nosleep;
    std::vector<uint32_t> mat_ids, dead;
    std::vector<uint64_t> grp, msk;
    bool any_com{false}, any_extras{false}, any_ids{false};
    explicit body_arrays(uint32_t n)
        : kind(n), stype(n), pos(3 * (size_t)n), orn(4 * (size_t)n), lv(3 * (size_t)n), av(3 * (size_t)n), m(n, 1.f), I(9 * (size_t)n, 0.f), sp(4 * (size_t)n, 0.f), fr(n, 0.5f),
          re(n, 0.f), g(3 * (size_t)n, 0.f), com(3 * (size_t)n, 0.f), xspin(n, 0.f), xroll(n, 0.f), xstiff(n, float(large_scalar)), xdamp(n, float(large_scalar)), hasI(n, 0),
          nosleep(n, 0), mat_ids(n, 0xFFFFu), grp(n, ~0ull), msk(n, ~0ull) {}
    edynhip_bodies view() const {
        return edynhip_bodies{kind.data(), pos.data(), orn.data(), lv.data(), av.data(), m.data(), I.data(), hasI.data(), stype.data(), sp.data(),
                              fr.data(), re.data(), grp.data(), msk.data(), g.data(), nosleep.data(), any_com ? com.data() : nullptr};
    }
};
// mesh_id(polyhedron_shape) -> the id of its mesh in the device context (created there on first sight)
template <typename MeshId>
inline void fill_body_arrays(entt::registry &registry, gpu_stepper &s, uint32_t first, uint32_t n, body_arrays &A, MeshId &&mesh_id) {
    auto &kind = A.kind; auto &stype = A.stype; auto &pos = A.pos; auto &orn = A.orn; auto &lv = A.lv; auto &av = A.av; auto &m = A.m; auto &I = A.I;
    auto &sp = A.sp; auto &fr = A.fr; auto &re = A.re; auto &g = A.g; auto &com = A.com; auto &xspin = A.xspin; auto &xroll = A.xroll; auto &xstiff = A.xstiff;
    auto &xdamp = A.xdamp; auto &hasI = A.hasI; auto &nosleep = A.nosleep; auto &mat_ids = A.mat_ids; auto &dead = A.dead; auto &grp = A.grp; auto &msk = A.msk;
    bool &any_com = A.any_com; bool &any_extras = A.any_extras; bool &any_ids = A.any_ids;
    for (uint32_t i = 0; i < n; ++i) {
        const entt::entity e = s.bodies[first + i];
        if (e == entt::null) {   // destroyed before it was ever uploaded (or before a re-upload): a shapeless static placeholder
            kind[i] = EDYNHIP_KIND_STATIC; stype[i] = EDYNHIP_SHAPE_NONE; orn[4 * i + 3] = 1.f; dead.push_back(first + i);
            continue;
        }
        kind[i] = registry.all_of<dynamic_tag>(e) ? EDYNHIP_KIND_DYNAMIC : registry.all_of<kinematic_tag>(e) ? EDYNHIP_KIND_KINEMATIC : EDYNHIP_KIND_STATIC;
        const auto &p = registry.get<position>(e); const auto &q = registry.get<orientation>(e);
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        orn[4 * i] = q.x; orn[4 * i + 1] = q.y; orn[4 * i + 2] = q.z; orn[4 * i + 3] = q.w;
        if (auto *v = registry.try_get<linvel>(e)) { lv[3 * i] = v->x; lv[3 * i + 1] = v->y; lv[3 * i + 2] = v->z; }
        if (auto *w = registry.try_get<angvel>(e)) { av[3 * i] = w->x; av[3 * i + 1] = w->y; av[3 * i + 2] = w->z; }
        if (auto *cm = registry.try_get<center_of_mass>(e)) {   // the device takes the ORIGIN and the offset, and moves position / velocity itself
            const auto &o = registry.get<origin>(e);
            const vector3 r{p.x - o.x, p.y - o.y, p.z - o.z};   // centre of mass - origin
            pos[3 * i] = o.x; pos[3 * i + 1] = o.y; pos[3 * i + 2] = o.z;
            lv[3 * i] -= av[3 * i + 1] * r.z - av[3 * i + 2] * r.y; lv[3 * i + 1] -= av[3 * i + 2] * r.x - av[3 * i] * r.z; lv[3 * i + 2] -= av[3 * i] * r.y - av[3 * i + 1] * r.x;
            com[3 * i] = cm->x; com[3 * i + 1] = cm->y; com[3 * i + 2] = cm->z; any_com = true;
        }
        if (auto *ms = registry.try_get<mass>(e)) m[i] = ms->s;
        if (auto *in = registry.try_get<inertia>(e)) {
            hasI[i] = 1;
            for (int r = 0; r < 3; ++r) { I[9 * i + 3 * r] = in->row[r].x; I[9 * i + 3 * r + 1] = in->row[r].y; I[9 * i + 3 * r + 2] = in->row[r].z; }
        }
        if (auto *b = registry.try_get<box_shape>(e)) { stype[i] = EDYNHIP_SHAPE_BOX; sp[4 * i] = b->half_extents.x; sp[4 * i + 1] = b->half_extents.y; sp[4 * i + 2] = b->half_extents.z; }
        else if (auto *sh = registry.try_get<sphere_shape>(e)) { stype[i] = EDYNHIP_SHAPE_SPHERE; sp[4 * i] = sh->radius; }
        else if (auto *cs = registry.try_get<capsule_shape>(e)) { stype[i] = EDYNHIP_SHAPE_CAPSULE; sp[4 * i] = cs->radius; sp[4 * i + 1] = cs->half_length; sp[4 * i + 2] = (float)(int)cs->axis; }
        else if (auto *cy = registry.try_get<cylinder_shape>(e)) { stype[i] = EDYNHIP_SHAPE_CYLINDER; sp[4 * i] = cy->radius; sp[4 * i + 1] = cy->half_length; sp[4 * i + 2] = (float)(int)cy->axis; }
        else if (auto *ph = registry.try_get<polyhedron_shape>(e)) {   // the mesh goes up once per context, however many bodies share it
            const uint32_t id = mesh_id(*ph);
            stype[i] = EDYNHIP_SHAPE_POLYHEDRON; sp[4 * i] = (float)id;
        }
        else if (auto *pl = registry.try_get<plane_shape>(e)) { stype[i] = EDYNHIP_SHAPE_PLANE; sp[4 * i] = pl->normal.x; sp[4 * i + 1] = pl->normal.y; sp[4 * i + 2] = pl->normal.z; sp[4 * i + 3] = pl->constant; }
        else stype[i] = EDYNHIP_SHAPE_NONE;
        if (auto *mt = registry.try_get<material>(e)) {
            fr[i] = mt->friction; re[i] = mt->restitution;
            xspin[i] = mt->spin_friction; xroll[i] = mt->roll_friction; xstiff[i] = mt->stiffness; xdamp[i] = mt->damping;
            any_extras = any_extras || mt->spin_friction > 0 || mt->roll_friction > 0 || mt->stiffness < large_scalar || mt->damping < large_scalar;
            mat_ids[i] = mt->id; any_ids = any_ids || mt->id != material::UnassignedID;
        }
        if (auto *f = registry.try_get<collision_filter>(e)) { grp[i] = f->group; msk[i] = f->mask; }
        if (registry.all_of<sleeping_disabled_tag>(e)) nosleep[i] = 1;
        if (auto *gr = registry.try_get<gravity>(e)) { g[3 * i] = gr->x; g[3 * i + 1] = gr->y; g[3 * i + 2] = gr->z; }
    }
}

// frames and parameter blocks of the cone / cvjoint / generic constraints [first_joint, nj): set_def(joint, frame_a, frame_b, params, generic)
template <typename SetDef>
inline void upload_joint_defs(entt::registry &registry, gpu_stepper &s, uint32_t first_joint, uint32_t nj, SetDef &&set_def) {
    for (uint32_t j = first_joint; j < nj; ++j) {
        const entt::entity e = s.constraints[j];
        if (e == entt::null) continue;
        auto rows9 = [](const matrix3x3 &m, float *o) { for (int r = 0; r < 3; ++r) { o[3 * r] = m.row[r].x; o[3 * r + 1] = m.row[r].y; o[3 * r + 2] = m.row[r].z; } };
        float fa[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, fb[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, q[16] = {0};
        const int kind = s.constraint_kind[j];
        if (kind == EDYNHIP_JOINT_CONE) {
            auto *cc = &registry.get<cone_constraint>(e);
            rows9(cc->frame, fa);
            q[0] = cc->span_tan[0]; q[1] = cc->span_tan[1]; q[2] = cc->restitution; q[3] = cc->bump_stop_stiffness; q[4] = cc->bump_stop_length;
            set_def(j, fa, fb, q, false);
        } else if (kind == EDYNHIP_JOINT_GENERIC) {
            auto *ge = &registry.get<generic_constraint>(e);
            rows9(ge->frame[0], fa); rows9(ge->frame[1], fb);
            float dof[60];
            for (int d = 0; d < 3; ++d) {
                const auto &l = ge->linear_dofs[d]; const auto &a = ge->angular_dofs[d];
                const float lv[10] = {l.limit_enabled ? 1.f : 0.f, l.offset_min, l.offset_max, l.limit_restitution, l.bump_stop_length, l.bump_stop_stiffness,
                                      l.friction_force, l.rest_offset, l.spring_stiffness, l.damping};
                const float av[10] = {a.limit_enabled ? 1.f : 0.f, a.angle_min, a.angle_max, a.limit_restitution, a.bump_stop_angle, a.bump_stop_stiffness,
                                      a.friction_torque, a.rest_angle, a.spring_stiffness, a.damping};
                for (int k = 0; k < 10; ++k) { dof[10 * d + k] = lv[k]; dof[10 * (3 + d) + k] = av[k]; }
            }
            set_def(j, fa, fb, dof, true);
        } else if (kind == EDYNHIP_JOINT_CVJOINT) {
            auto *cv = &registry.get<cvjoint_constraint>(e);
            rows9(cv->frame[0], fa); rows9(cv->frame[1], fb);
            const float v[15] = {cv->twist_min, cv->twist_max, cv->twist_restitution, cv->twist_bump_stop_angle, cv->twist_bump_stop_stiffness,
                                 cv->twist_friction_torque, cv->twist_rest_angle, cv->twist_stiffness, cv->twist_damping,
                                 cv->rest_direction.x, cv->rest_direction.y, cv->rest_direction.z, cv->bend_stiffness, cv->bend_friction_torque, cv->bend_damping};
            for (int k = 0; k < 15; ++k) q[k] = v[k];
            set_def(j, fa, fb, q, false);
        }
    }
}

// init_config::devices names several GPUs: the whole scene goes to the multi-GPU world (edynhip_world_*), which takes a scene once -
// so every (re)upload builds a new world from the registry's current components.
inline void upload_scene_multi(entt::registry &registry, gpu_stepper &s) {
    const uint32_t total = (uint32_t)s.bodies.size(), nj = (uint32_t)s.constraints.size();
    if (s.world) { edynhip_world_destroy(s.world); s.world = nullptr; }
    if (!s.mixings.empty()) throw stepper_error(EDYNHIP_ERR_UNSUPPORTED, "edyn: the material mix table is a single-device feature (init_config::devices)");
    edynhip_config c{};
    c.max_manifolds = s.cfg.max_manifolds;
    c.fixed_dt = s.cfg.fixed_dt;
    c.num_velocity_iterations = s.cfg.num_solver_velocity_iterations;
    c.num_position_iterations = s.cfg.num_solver_position_iterations;
    c.gravity[0] = s.cfg.gravity.x; c.gravity[1] = s.cfg.gravity.y; c.gravity[2] = s.cfg.gravity.z;
    c.flags = (s.cfg.island_sleeping ? EDYNHIP_FLAG_SLEEPING : 0u) | (s.cfg.fused_velocity_rows ? EDYNHIP_FLAG_FUSED_VELOCITY_ROWS : 0u) | (s.cfg.block_position ? EDYNHIP_FLAG_BLOCK_POSITION : 0u);
    std::vector<int32_t> devs(s.cfg.devices.begin(), s.cfg.devices.end());
    int st = 0;
    s.world = edynhip_world_create(&c, devs.data(), (uint32_t)devs.size(), &st);
    if (!s.world) throw stepper_error(st, std::string("edynhip_world_create: ") + edynhip_world_last_error(nullptr));
    if (s.should_collide) check(s, edynhip_world_set_pair_filter(s.world, &pair_filter_trampoline, &s));   // a rebuilt world asks the same predicate
    s.meshes.clear();
    body_arrays A(total);
    fill_body_arrays(registry, s, 0, total, A, [&](const polyhedron_shape &ph) {
        uint32_t id = ~0u;
        for (auto &known : s.meshes) if (known.first == ph.mesh) id = known.second;
        if (id == ~0u) {
            const convex_mesh &cm = *ph.mesh;
            std::vector<float> mv(3 * cm.vertices.size());
            for (size_t k = 0; k < cm.vertices.size(); ++k) { mv[3 * k] = cm.vertices[k].x; mv[3 * k + 1] = cm.vertices[k].y; mv[3 * k + 2] = cm.vertices[k].z; }
            check(s, edynhip_world_create_convex_mesh(s.world, (uint32_t)cm.vertices.size(), mv.data(), (uint32_t)cm.indices.size(), cm.indices.data(),
                                                      (uint32_t)cm.num_faces(), cm.faces.data(), cm.initialized ? EDYNHIP_MESH_INITIALIZED : 0u, &id));
            s.meshes.emplace_back(ph.mesh, id);
        }
        return id;
    });
    if (A.any_extras || A.any_ids) throw stepper_error(EDYNHIP_ERR_UNSUPPORTED, "edyn: contact_extras materials and material ids are single-device features (init_config::devices)");
    const edynhip_bodies b = A.view();   // destroyed bodies went up as shapeless static placeholders: they take part in nothing
    check(s, edynhip_world_set_bodies(s.world, total, &b));
    std::vector<int32_t> jt; std::vector<uint32_t> jb; std::vector<float> jp, ja, jq;
    std::vector<uint32_t> dead_joints;
    joint_arrays(registry, s, 0, jt, jb, jp, ja, jq, dead_joints);
    for (uint32_t j : dead_joints) { jt[j] = EDYNHIP_JOINT_NULL; jb[2 * j] = jb[2 * j + 1] = 0; }   // a destroyed constraint keeps its index: a rowless self-edge
    edynhip_joints js{jt.data(), jb.data(), jp.data(), ja.data(), jq.data()};
    check(s, edynhip_world_set_joints(s.world, nj, nj ? &js : nullptr));
    upload_joint_defs(registry, s, 0, nj, [&](uint32_t j, const float *fa, const float *fb, const float *q, bool generic) {
        check(s, edynhip_world_set_joint_definition(s.world, j, fa, fb, q, generic ? 1 : 0));
    });
    for (const auto &e : s.exclusions) check(s, edynhip_world_exclude_collision(s.world, e[0], e[1]));
    s.uploaded_bodies = total; s.uploaded_constraints = nj; s.exclusions_uploaded = s.exclusions.size();
    s.scene_dirty = false; s.state_dirty = false;
}

constexpr uint32_t prefetch_events_max = 8192;   // contact events that travel ahead of a step's state (sequential modes); more = read after the step
inline void upload_scene(entt::registry &registry, gpu_stepper &s) {
    const uint32_t total = (uint32_t)s.bodies.size();
    const uint32_t nj = (uint32_t)s.constraints.size();
    std::vector<edynhip_manifold> carried;   // contact state carried over a capacity growth (indices are stable)
    std::vector<float> carried_impulses, carried_angles;   // joints: 24 applied-impulse slots + the tracked angle, by joint index
    std::vector<uint8_t> carried_asleep;                   // sleeping tags by body index
    std::vector<uint32_t> carried_labels; std::vector<double> carried_since; double carried_clock = 0;   // island labels + sleep timers by label + their clock
    bool regrown = false;
    if (!s.ctx || total > s.capacity || nj > s.joint_capacity || s.recreate) {
        s.recreate = false;
        if (s.ctx) {
            uint32_t m = 0;
            check(s, edynhip_num_manifolds(s.ctx, &m));
            carried.resize(m);
            if (m) check(s, edynhip_get_manifolds(s.ctx, carried.data(), m, &m));
            if (!s.reshaped.empty()) {   // rigidbody_set_shape / set_kind: the old contacts of those bodies mean nothing to the new shape / kind
                carried.erase(std::remove_if(carried.begin(), carried.end(), [&](const edynhip_manifold &rec) {
                    return std::find(s.reshaped.begin(), s.reshaped.end(), rec.body[0]) != s.reshaped.end() || std::find(s.reshaped.begin(), s.reshaped.end(), rec.body[1]) != s.reshaped.end(); }), carried.end());
                s.reshaped.clear();
            }
            if (s.refresh_friction) {   // set_rigidbody_friction: existing points take the mixed friction of the current materials, except
                for (auto &rec : carried) {   // pairs combined through the material mix table (rigidbody.cpp:324-350)
                    const entt::entity ea = s.bodies[rec.body[0]], eb = s.bodies[rec.body[1]];
                    if (ea == entt::null || eb == entt::null) continue;
                    const material *ma = registry.try_get<material>(ea), *mb = registry.try_get<material>(eb);
                    if (!ma || !mb) continue;
                    bool tabled = false;
                    for (auto &mx : s.mixings) tabled = tabled || (mx.id0 == ma->id && mx.id1 == mb->id) || (mx.id0 == mb->id && mx.id1 == ma->id);
                    if (tabled) continue;
                    for (uint32_t k = 0; k < rec.num_points; ++k) rec.pt[k].friction = std::sqrt(ma->friction * mb->friction);
                }
                s.refresh_friction = false;
            }
            // joints' applied impulses (warm start) and tracked angles, and the sleeping tags, travel too (by index: both are stable)
            if (s.uploaded_constraints) {
                carried_impulses.assign((size_t)24 * s.uploaded_constraints, 0.f);
                std::vector<float> imp10((size_t)10 * s.uploaded_constraints, 0.f);
                check(s, edynhip_get_joint_slot_impulses(s.ctx, carried_impulses.data()));
                check(s, edynhip_get_joint_impulses(s.ctx, imp10.data()));
                carried_angles.resize(s.uploaded_constraints);
                for (uint32_t j = 0; j < s.uploaded_constraints; ++j) carried_angles[j] = imp10[(size_t)10 * j + 9];
            }
            if (s.cfg.island_sleeping && s.uploaded_bodies) {
                carried_asleep.assign(s.uploaded_bodies, 0);
                check(s, edynhip_get_asleep(s.ctx, carried_asleep.data()));
                carried_labels.assign(s.uploaded_bodies, 0); carried_since.assign(s.uploaded_bodies, -1.0);   // the islands' sleep timers go on in the new context
                check(s, edynhip_get_sleep_timers(s.ctx, carried_labels.data(), carried_since.data(), &carried_clock));
            }
            edynhip_destroy(s.ctx); s.ctx = nullptr;
            s.exclusions_uploaded = 0;
            regrown = true;
            s.contacts_resync = true; s.snapshot_pending = false; s.records_pending = false; s.events_stale_through = 0;
        }
        s.uploaded_bodies = s.uploaded_constraints = 0;
        edynhip_config c{};
        c.device = s.cfg.device;
        c.max_bodies = s.cfg.max_bodies > total ? s.cfg.max_bodies : total + total / 2 + 16;
        c.max_manifolds = s.cfg.max_manifolds;
        c.max_joints = nj + nj / 2 + 16;
        c.fixed_dt = s.cfg.fixed_dt;
        c.num_velocity_iterations = s.cfg.num_solver_velocity_iterations;
        c.num_position_iterations = s.cfg.num_solver_position_iterations;
        c.gravity[0] = s.cfg.gravity.x; c.gravity[1] = s.cfg.gravity.y; c.gravity[2] = s.cfg.gravity.z;
        c.flags = (s.cfg.island_sleeping ? EDYNHIP_FLAG_SLEEPING : 0u) | (s.cfg.materialize_contacts ? EDYNHIP_FLAG_CONTACT_EVENTS : 0u) |
                  (s.cfg.fused_velocity_rows ? EDYNHIP_FLAG_FUSED_VELOCITY_ROWS : 0u) | (s.cfg.block_position ? EDYNHIP_FLAG_BLOCK_POSITION : 0u) |
                  (s.cfg.exclusive_device ? EDYNHIP_FLAG_EXCLUSIVE_DEVICE : 0u);
        int st = 0;
        s.ctx = edynhip_create(&c, &st);
        s.meshes.clear();   // meshes belong to the context
        if (!s.ctx) throw stepper_error(st, std::string("edynhip_create: ") + edynhip_last_error(nullptr));
        if (s.should_collide) check(s, edynhip_set_pair_filter(s.ctx, &pair_filter_trampoline, &s));   // a re-created context asks the same predicate
        // contact entities follow the narrowphase, not the end of the step (edynhip.h "Contact-event prefetch"): the host builds them while the solve runs.
        // Asynchronous mode reads its events with the record snapshots and only takes the synchronous write-back for step callbacks: the
        // prefetch (a pack kernel and up to 200 KB of copy per step call) is switched on there when a callback is first seen (run_steps)
        s.prefetch_on = s.cfg.materialize_contacts && (s.cfg.execution_mode != execution_mode::asynchronous || s.pre_step || s.post_step);
        if (s.prefetch_on) check(s, edynhip_set_event_prefetch(s.ctx, prefetch_events_max));
        s.capacity = c.max_bodies; s.joint_capacity = c.max_joints;
        s.params_dirty = false;
    }
    // Bodies created since the last upload are appended (edynhip_add_bodies): the running contact state of the others stays.
    const uint32_t first = s.uploaded_bodies, n = total - first;
    body_arrays A(n);
    fill_body_arrays(registry, s, first, n, A, [&](const polyhedron_shape &ph) {   // the mesh goes up once per context, however many bodies share it
        uint32_t id = ~0u;
        for (auto &known : s.meshes) if (known.first == ph.mesh) id = known.second;
        if (id == ~0u) {
            const convex_mesh &cm = *ph.mesh;
            std::vector<float> mv(3 * cm.vertices.size());
            for (size_t k = 0; k < cm.vertices.size(); ++k) { mv[3 * k] = cm.vertices[k].x; mv[3 * k + 1] = cm.vertices[k].y; mv[3 * k + 2] = cm.vertices[k].z; }
            check(s, edynhip_create_convex_mesh(s.ctx, (uint32_t)cm.vertices.size(), mv.data(), (uint32_t)cm.indices.size(), cm.indices.data(),
                                                (uint32_t)cm.num_faces(), cm.faces.data(), cm.initialized ? EDYNHIP_MESH_INITIALIZED : 0u, &id));
            s.meshes.emplace_back(ph.mesh, id);
        }
        return id;
    });
    const edynhip_bodies b = A.view();
    auto &xspin = A.xspin; auto &xroll = A.xroll; auto &xstiff = A.xstiff; auto &xdamp = A.xdamp; auto &mat_ids = A.mat_ids; auto &dead = A.dead;
    const bool any_extras = A.any_extras, any_ids = A.any_ids;
    if (first == 0) check(s, edynhip_set_bodies(s.ctx, n, &b));
    else if (n) check(s, edynhip_add_bodies(s.ctx, n, &b));
    if (any_extras) check(s, edynhip_set_material_extras(s.ctx, first, n, xspin.data(), xroll.data(), xstiff.data(), xdamp.data()));
    if (first == 0) for (auto &m : s.mixings) check(s, edynhip_insert_material_mixing(s.ctx, m.id0, m.id1, m.v));   // a (re)created context
    if (any_ids) check(s, edynhip_set_material_ids(s.ctx, first, n, mat_ids.data()));
    if (!dead.empty()) check(s, edynhip_remove_bodies(s.ctx, (uint32_t)dead.size(), dead.data()));
    if (s.cfg.execution_mode == execution_mode::asynchronous) {   // the registry state these bodies went up with is what "unedited" means for them
        if (s.shadow.size() < (size_t)13 * total) s.shadow.resize((size_t)13 * total, 0.f);
        for (uint32_t i = first; i < total; ++i) {
            const entt::entity e = s.bodies[i];
            if (e == entt::null) continue;
            float *sh = &s.shadow[(size_t)13 * i];
            const auto &p = registry.get<position>(e); const auto &q = registry.get<orientation>(e);
            sh[0] = p.x; sh[1] = p.y; sh[2] = p.z; sh[3] = q.x; sh[4] = q.y; sh[5] = q.z; sh[6] = q.w;
            if (auto *v = registry.try_get<linvel>(e)) { sh[7] = v->x; sh[8] = v->y; sh[9] = v->z; }
            if (auto *w = registry.try_get<angvel>(e)) { sh[10] = w->x; sh[11] = w->y; sh[12] = w->z; }
        }
    }
    s.uploaded_bodies = total;
    if (regrown && !carried.empty()) check(s, edynhip_set_manifolds(s.ctx, carried.data(), (uint32_t)carried.size()));
    // joints: everything after a (re)creation of the context, otherwise only the ones made since the last upload
    std::vector<int32_t> jt; std::vector<uint32_t> jb; std::vector<float> jp, ja, jq;
    std::vector<uint32_t> dead_joints;
    if (first == 0) {
        joint_arrays(registry, s, 0, jt, jb, jp, ja, jq, dead_joints);
        edynhip_joints js{jt.data(), jb.data(), jp.data(), ja.data(), jq.data()};
        check(s, edynhip_set_joints(s.ctx, nj, nj ? &js : nullptr));
    } else if (nj > s.uploaded_constraints) {
        joint_arrays(registry, s, s.uploaded_constraints, jt, jb, jp, ja, jq, dead_joints);
        edynhip_joints js{jt.data(), jb.data(), jp.data(), ja.data(), jq.data()};
        uint32_t first_joint = 0;
        check(s, edynhip_add_joints(s.ctx, nj - s.uploaded_constraints, &js, &first_joint));
    }
    upload_joint_defs(registry, s, first == 0 ? 0u : s.uploaded_constraints, nj, [&](uint32_t j, const float *fa, const float *fb, const float *q, bool generic) {
        check(s, generic ? edynhip_set_generic_definition(s.ctx, j, fa, fb, q) : edynhip_set_joint_definition(s.ctx, j, fa, fb, q));
    });
    if (regrown && !carried_impulses.empty()) {   // joints created since then start from zero impulses, like any new joint
        // (a joint made just now keeps the angle reset_angle gave it: its carried slot is not applied)
        const uint32_t had = (uint32_t)std::min<size_t>(carried_angles.size(), nj);
        carried_impulses.resize((size_t)24 * nj, 0.f); carried_angles.resize(nj, 0.f);
        std::vector<float> now10((size_t)10 * nj, 0.f);
        check(s, edynhip_get_joint_impulses(s.ctx, now10.data()));
        for (uint32_t j = 0; j < nj; ++j) if (j >= had || s.constraints[j] == entt::null) carried_angles[j] = now10[(size_t)10 * j + 9];
        check(s, edynhip_set_joint_warm_start(s.ctx, carried_impulses.data(), carried_angles.data()));
    }
    if (!dead_joints.empty()) check(s, edynhip_remove_joints(s.ctx, (uint32_t)dead_joints.size(), dead_joints.data()));
    s.uploaded_constraints = nj;
    // collision exclusions: the whole list into a (re)created context, the new ones otherwise
    for (size_t k = s.exclusions_uploaded; k < s.exclusions.size(); ++k) check(s, edynhip_exclude_collision(s.ctx, s.exclusions[k][0], s.exclusions[k][1]));
    s.exclusions_uploaded = s.exclusions.size();
    if (regrown && !carried_asleep.empty()) {
        carried_asleep.resize(total, 0);
        check(s, edynhip_set_asleep(s.ctx, carried_asleep.data()));
        const uint32_t had = (uint32_t)carried_labels.size();
        carried_labels.resize(total); carried_since.resize(total, -1.0);
        for (uint32_t i = had; i < total; ++i) carried_labels[i] = i;   // bodies made since: islands of their own, no timer
        check(s, edynhip_set_sleep_timers(s.ctx, carried_labels.data(), carried_since.data(), carried_clock));
    }
    s.scene_dirty = false;
    if (first == 0) s.state_dirty = false;
}

inline void upload_state(entt::registry &registry, gpu_stepper &s) {
    const uint32_t n = (uint32_t)s.bodies.size();
    std::vector<float> pos(3 * n), orn(4 * n), lv(3 * n, 0.f), av(3 * n, 0.f);
    for (uint32_t i = 0; i < n; ++i) {
        const entt::entity e = s.bodies[i];
        if (e == entt::null) { orn[4 * i + 3] = 1.f; continue; }
        const auto &p = registry.get<position>(e); const auto &q = registry.get<orientation>(e);
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        orn[4 * i] = q.x; orn[4 * i + 1] = q.y; orn[4 * i + 2] = q.z; orn[4 * i + 3] = q.w;
        if (auto *v = registry.try_get<linvel>(e)) { lv[3 * i] = v->x; lv[3 * i + 1] = v->y; lv[3 * i + 2] = v->z; }
        if (auto *w = registry.try_get<angvel>(e)) { av[3 * i] = w->x; av[3 * i + 1] = w->y; av[3 * i + 2] = w->z; }
    }
    check(s, edynhip_set_state(s.ctx, pos.data(), orn.data(), lv.data(), av.data()));
    s.state_dirty = false;
}

// ---- the registry write-back. The device hands over one 96-byte record per body in pinned host memory (edynhip_snapshot_records /
// edynhip_snapshot_map): state, presentation transforms (update_presentation.cpp:56-84 evaluated on the device), origin
// (update_origins.cpp:13-15) and flags. The loop below is all the host does: it reads the records in place and writes the
// components through direct pool handles (registry.storage<T>(): one sparse look-up per component instead of a type-indexed
// registry.get), in parallel over the bodies; sleeping_tag changes (island_manager.cpp:541-565) are collected per worker and
// applied serially afterwards - emplace / remove are the only operations that change a pool's layout.
inline worker_pool &pool_of(gpu_stepper &s) {
    if (!s.pool) {
        size_t n = s.cfg.num_worker_threads;
        if (const char *e = std::getenv("EDYN_NUM_WORKER_THREADS")) n = (size_t)std::atoi(e);
        if (n == 0) n = std::min<size_t>(16, std::max<size_t>(1, std::thread::hardware_concurrency() / 2));
        s.pool = std::make_unique<worker_pool>((unsigned)n);
        s.sleep_changes.assign(s.pool->size(), {});
    }
    return *s.pool;
}
constexpr uint32_t write_back_grain = 512;   // bodies per chunk of the parallel loops
inline void import_records(entt::registry &registry, gpu_stepper &s, const edynhip_record_view &view, bool presentation) {
    const uint32_t n = (uint32_t)std::min<size_t>(s.bodies.size(), view.num_bodies);
    if (n == 0) return;
    const bool async = s.cfg.execution_mode == execution_mode::asynchronous;
    if (async && s.shadow.size() < (size_t)13 * s.bodies.size()) s.shadow.resize((size_t)13 * s.bodies.size(), 0.f);
    if (s.asleep_shadow.size() < s.bodies.size()) s.asleep_shadow.resize(s.bodies.size(), 0);
    auto &sp = registry.storage<position>(); auto &sq = registry.storage<orientation>();
    auto &sv = registry.storage<linvel>(); auto &sw = registry.storage<angvel>();
    auto &spp = registry.storage<present_position>(); auto &spo = registry.storage<present_orientation>();
    auto &sorg = registry.storage<origin>();
    worker_pool &pool = pool_of(s);
    const edynhip_body_record *recs = view.records;
    pool.parallel_for(n, write_back_grain, [&](uint32_t begin, uint32_t end, unsigned worker) {
        auto &changes = s.sleep_changes[worker];
        for (uint32_t i = begin; i < end; ++i) {
            const entt::entity e = s.bodies[i];
            const edynhip_body_record &r = recs[i];
            if (e == entt::null || !(r.flags & EDYNHIP_RECORD_DYNAMIC) || (r.flags & EDYNHIP_RECORD_REMOVED)) continue;   // static / kinematic: the registry is the authority
            if (!sp.contains(e) || !sv.contains(e)) continue;   // stripped by the user since the last update (noticed by sync_removed next time)
            auto &p = sp.get(e); p.x = r.pos[0]; p.y = r.pos[1]; p.z = r.pos[2];
            auto &q = sq.get(e); q.x = r.orn[0]; q.y = r.orn[1]; q.z = r.orn[2]; q.w = r.orn[3];
            auto &v = sv.get(e); v.x = r.linvel[0]; v.y = r.linvel[1]; v.z = r.linvel[2];
            auto &w = sw.get(e); w.x = r.angvel[0]; w.y = r.angvel[1]; w.z = r.angvel[2];
            if ((r.flags & EDYNHIP_RECORD_HAS_ORIGIN) && sorg.contains(e)) { auto &o = sorg.get(e); o.x = r.origin[0]; o.y = r.origin[1]; o.z = r.origin[2]; }
            const uint8_t asleep = (r.flags & EDYNHIP_RECORD_ASLEEP) ? 1 : 0;
            if (presentation && !asleep && spp.contains(e) && spo.contains(e)) {   // update_presentation's views exclude sleeping entities
                auto &pp = spp.get(e); pp.x = r.present_pos[0]; pp.y = r.present_pos[1]; pp.z = r.present_pos[2];
                auto &po = spo.get(e); po.x = r.present_orn[0]; po.y = r.present_orn[1]; po.z = r.present_orn[2]; po.w = r.present_orn[3];
            }
            if (asleep != s.asleep_shadow[i]) changes.push_back(i);
            if (async) std::memcpy(&s.shadow[(size_t)13 * i], r.pos, 13 * sizeof(float));   // pos orn linvel angvel are the record's first 13 floats
        }
    });
    if (s.cfg.island_sleeping)
        for (auto &changes : s.sleep_changes) {
            for (uint32_t i : changes) {
                const entt::entity e = s.bodies[i];
                const bool asleep = (recs[i].flags & EDYNHIP_RECORD_ASLEEP) != 0, tagged = registry.all_of<sleeping_tag>(e);
                if (asleep && !tagged) registry.emplace<sleeping_tag>(e);
                else if (!asleep && tagged) registry.remove<sleeping_tag>(e);
                s.asleep_shadow[i] = asleep ? 1 : 0;
            }
            changes.clear();
        }
    else for (auto &changes : s.sleep_changes) changes.clear();
}
// Contact events that travelled with a record snapshot -> contact entities (sync_contacts below applies them).
inline void apply_contact_events(entt::registry &registry, gpu_stepper &s, std::vector<edynhip_contact_event> &ev, int rc);
// sequential write-back: enqueue the pack + copy behind the step, wait for it, import. `events_max`: how many contact events travel along.
inline void write_back(entt::registry &registry, gpu_stepper &s, bool presentation) {
    { phase_timer t(s.tm.state_wait); check(s, edynhip_snapshot_records(s.ctx, s.present_dt, 0u, EDYNHIP_SNAPSHOT_DIRECT)); }   // the pack, enqueued right behind the step, straight into pinned memory: this update waits for it
    if (s.cfg.materialize_contacts) {
        // Contact points become entities while the step's solve is still running on the device - where the reference creates them too:
        // inside the step, by the narrowphase, before the solver moves anything (narrowphase.cpp:21-40, collision_util.cpp:311-430).
        phase_timer t(s.tm.contacts);
        const edynhip_contact_event *events = nullptr;
        uint32_t num = 0, total = 0;
        check(s, edynhip_prefetched_events(s.ctx, &events, &num, &total));
        std::vector<edynhip_contact_event> ev;
        int rc = EDYNHIP_OK;
        if (total > num) {   // more than travel ahead (a pile hitting the ground): the whole list, once the step is through
            uint32_t cnt = 0;
            ev.resize(total);
            rc = edynhip_get_contact_events(s.ctx, ev.data(), (uint32_t)ev.size(), &cnt);
            ev.resize(rc == EDYNHIP_OK ? cnt : 0);
        } else if (num) ev.assign(events, events + num);
        apply_contact_events(registry, s, ev, rc);
    }
    pool_of(s).prewake();
    edynhip_record_view view{};
    { phase_timer t(s.tm.state_wait); check(s, edynhip_snapshot_map(s.ctx, &view)); }
    { phase_timer t(s.tm.write_back); import_records(registry, s, view, presentation); }
}

// registry.destroy(entity) / clear_rigidbody on bodies and constraints since the last update: the reference reacts through
// on_destroy hooks (island_manager.cpp:24-27); this shim notices at the next update that the entity is gone (or no longer
// carries the component that made it a body / a joint) and removes it from the device world, keeping every other index.
inline void on_stepper_entity_gone(gpu_stepper &s, entt::registry &, entt::entity) { s.removal_pending = true; }
template <typename... T> inline void connect_removal_hooks(entt::registry &registry, gpu_stepper &s) {
    (registry.template on_destroy<T>().template connect<&on_stepper_entity_gone>(s), ...);
}
template <typename... T> inline void disconnect_removal_hooks(entt::registry &registry, gpu_stepper &s) {
    (registry.template on_destroy<T>().template disconnect<&on_stepper_entity_gone>(s), ...);
}
inline void sync_removed(entt::registry &registry, gpu_stepper &s) {
    // The reference reacts to destroyed bodies / constraints through on_destroy hooks (island_manager.cpp:24-27); so does this shim:
    // the hooks (attach) raise `removal_pending`, and only then is the list of bodies and constraints walked.
    if (s.hooks_connected && !s.removal_pending) return;
    s.removal_pending = false;
    std::vector<uint32_t> gone_bodies, gone_joints;
    for (uint32_t i = 0; i < (uint32_t)s.bodies.size(); ++i) {
        const entt::entity e = s.bodies[i];
        if (e == entt::null) continue;
        const bool alive = registry.valid(e) && registry.all_of<rigidbody_tag, body_index>(e) && registry.get<body_index>(e).value == i;
        if (!alive) { s.bodies[i] = entt::null; if (i < s.uploaded_bodies) gone_bodies.push_back(i); }
    }
    for (uint32_t j = 0; j < (uint32_t)s.constraints.size(); ++j) {
        const entt::entity e = s.constraints[j];
        if (e == entt::null) continue;
        const constraint_base *cb = constraint_of(registry, e, s.constraint_kind[j]);
        bool alive = cb != nullptr;
        if (alive)   // a joint whose body was destroyed goes with it
            for (int k = 0; k < 2; ++k) if (!registry.valid(cb->body[k]) || !registry.all_of<body_index>(cb->body[k])) alive = false;
        if (!alive) {
            s.constraints[j] = entt::null;
            if (cb != nullptr) registry.destroy(e);   // its body is gone (an entity carrying two constraints: the second sees it gone)
            if (j < s.uploaded_constraints) gone_joints.push_back(j);
        }
    }
    if (!s.ctx) return;
    if (!gone_joints.empty()) check(s, edynhip_remove_joints(s.ctx, (uint32_t)gone_joints.size(), gone_joints.data()));
    if (!gone_bodies.empty()) check(s, edynhip_remove_bodies(s.ctx, (uint32_t)gone_bodies.size(), gone_bodies.data()));
}
inline void apply_params(gpu_stepper &s) {   // settings on the running context: nothing is re-created, no contact state is lost
    if (!s.ctx || !s.params_dirty) { s.params_dirty = false; return; }
    edynhip_params p{};
    p.fixed_dt = s.cfg.fixed_dt;
    p.num_velocity_iterations = s.cfg.num_solver_velocity_iterations;
    p.num_position_iterations = s.cfg.num_solver_position_iterations;
    p.gravity[0] = s.cfg.gravity.x; p.gravity[1] = s.cfg.gravity.y; p.gravity[2] = s.cfg.gravity.z;
    check(s, edynhip_set_params(s.ctx, &p));
    s.params_dirty = false;
}
// Contact entities. The device reports what changed (edynhip_get_contact_events); the registry follows: a
// contact_manifold entity per overlapping pair (make_contact_manifold, constraint_util.cpp:60-102), a contact_point entity
// per point (create_contact_point, collision_util.cpp:311-388), destroyed when the device says so.
inline uint64_t manifold_key(uint32_t a, uint32_t b) { return ((uint64_t)a << 32) | b; }
inline void refresh_contact_points(entt::registry &registry, gpu_stepper &s) {
    uint32_t n = 0;
    check(s, edynhip_num_manifolds(s.ctx, &n));
    if (n == 0) return;
    std::vector<edynhip_manifold> recs(n);
    std::vector<uint64_t> ids((size_t)4 * n);
    check(s, edynhip_get_manifolds(s.ctx, recs.data(), n, &n));
    check(s, edynhip_get_point_ids(s.ctx, ids.data(), n, &n));
    for (uint32_t m = 0; m < n; ++m)
        for (uint32_t k = 0; k < recs[m].num_points; ++k) {
            auto it = s.point_entities.find(ids[(size_t)4 * m + k]);
            if (it == s.point_entities.end()) continue;
            const edynhip_point &p = recs[m].pt[k];
            registry.get<contact_point>(it->second) = {{p.pivotA[0], p.pivotA[1], p.pivotA[2]}, {p.pivotB[0], p.pivotB[1], p.pivotB[2]}, {p.normal[0], p.normal[1], p.normal[2]}};
            registry.get<contact_point_geometry>(it->second) = {{p.local_normal[0], p.local_normal[1], p.local_normal[2]}, p.distance, p.attachment};
            registry.get<contact_point_impulse>(it->second) = {p.normal_impulse, {p.friction_impulse[0], p.friction_impulse[1]}};
        }
}
inline void apply_contact_events(entt::registry &registry, gpu_stepper &s, std::vector<edynhip_contact_event> &ev, int rc) {
    if (!s.cfg.materialize_contacts || !s.ctx) return;
    s.tm.contact_events += ev.size();
    if (ev.empty() && rc == EDYNHIP_OK && !s.contacts_resync) { if (s.cfg.contact_point_data) refresh_contact_points(registry, s); return; }
    auto body_of = [&](uint32_t i) { return i < s.bodies.size() ? s.bodies[i] : entt::entity{entt::null}; };
    // direct pool handles: a thousand points come and go per update on the headline pile - no type-indexed pool look-up per component
    auto &pool_manifold = registry.storage<contact_manifold>(); auto &pool_list = registry.storage<contact_point_list>();
    auto &pool_geometry = registry.storage<contact_point_geometry>(); auto &pool_impulse = registry.storage<contact_point_impulse>();
    auto &pool_point = registry.storage<contact_point>();
    auto make_manifold = [&](uint32_t a, uint32_t b) {
        const entt::entity e = registry.create();
        pool_manifold.emplace(e, contact_manifold{{body_of(a), body_of(b)}, 0u});
        s.manifold_entities[manifold_key(a, b)] = e;
        return e;
    };
    auto make_point = [&](uint32_t a, uint32_t b, uint64_t id) {
        auto mit = s.manifold_entities.find(manifold_key(a, b));
        const entt::entity parent = mit != s.manifold_entities.end() ? mit->second : make_manifold(a, b);
        const entt::entity e = registry.create();
        pool_list.emplace(e, contact_point_list{parent, id});
        pool_geometry.emplace(e, contact_point_geometry{});
        pool_impulse.emplace(e, contact_point_impulse{});
        pool_point.emplace(e, contact_point{});   // last, as create_contact_point does: a listener finds the other components
        ++pool_manifold.get(parent).num_points;
        s.point_entities[id] = e;
    };
    if (rc == EDYNHIP_ERR_CAPACITY || s.contacts_resync) {   // more events than the context holds, or a re-created context: rebuild from the manifolds
        s.contacts_resync = false;
        for (auto &kv : s.point_entities) registry.destroy(kv.second);
        for (auto &kv : s.manifold_entities) registry.destroy(kv.second);
        s.point_entities.clear(); s.manifold_entities.clear();
        uint32_t m = 0;
        check(s, edynhip_num_manifolds(s.ctx, &m));
        std::vector<edynhip_manifold> recs(m);
        std::vector<uint64_t> ids((size_t)4 * m);
        if (m) { check(s, edynhip_get_manifolds(s.ctx, recs.data(), m, &m)); check(s, edynhip_get_point_ids(s.ctx, ids.data(), m, &m)); }
        for (uint32_t i = 0; i < m; ++i) {
            make_manifold(recs[i].body[0], recs[i].body[1]);
            for (uint32_t k = 0; k < recs[i].num_points; ++k) make_point(recs[i].body[0], recs[i].body[1], ids[(size_t)4 * i + k]);
        }
    } else {
        check(s, rc);
        // within a step: ends before beginnings (a replaced point leaves before its successor arrives); steps in order
        auto rank = [](uint32_t t) { return t == EDYNHIP_EVENT_POINT_DESTROYED ? 0 : t == EDYNHIP_EVENT_MANIFOLD_DESTROYED ? 1 : t == EDYNHIP_EVENT_MANIFOLD_CREATED ? 2 : 3; };
        std::stable_sort(ev.begin(), ev.end(), [&](const edynhip_contact_event &x, const edynhip_contact_event &y) {
            return x.step != y.step ? x.step < y.step : rank(x.type) < rank(y.type);
        });
        for (const auto &e : ev) {
            switch (e.type) {
            case EDYNHIP_EVENT_MANIFOLD_CREATED: make_manifold(e.body[0], e.body[1]); break;
            case EDYNHIP_EVENT_POINT_CREATED: make_point(e.body[0], e.body[1], e.point_id); break;
            case EDYNHIP_EVENT_POINT_DESTROYED: {
                auto it = s.point_entities.find(e.point_id);
                if (it == s.point_entities.end()) break;
                const entt::entity parent = pool_list.get(it->second).parent;
                if (pool_manifold.contains(parent)) { auto &m = pool_manifold.get(parent); if (m.num_points) --m.num_points; }
                registry.destroy(it->second);
                s.point_entities.erase(it);
                break;
            }
            case EDYNHIP_EVENT_MANIFOLD_DESTROYED: {
                auto it = s.manifold_entities.find(manifold_key(e.body[0], e.body[1]));
                if (it == s.manifold_entities.end()) break;
                registry.destroy(it->second);
                s.manifold_entities.erase(it);
                break;
            }
            default: break;
            }
        }
    }
    if (s.cfg.contact_point_data) refresh_contact_points(registry, s);
}
inline void sync_contacts(entt::registry &registry, gpu_stepper &s) {   // the events straight from the device (no record snapshot at hand)
    if (!s.cfg.materialize_contacts || !s.ctx) return;
    uint32_t n = 0;
    int rc = edynhip_get_contact_events(s.ctx, nullptr, 0, &n);
    std::vector<edynhip_contact_event> ev(n);
    if (rc == EDYNHIP_OK && n) rc = edynhip_get_contact_events(s.ctx, ev.data(), n, &n);
    apply_contact_events(registry, s, ev, rc);
}
inline void import_state(entt::registry &registry, gpu_stepper &s, const std::vector<float> &pos, const std::vector<float> &orn,
                         const std::vector<float> &lv, const std::vector<float> &av, uint32_t count) {
    // `count` = the bodies the arrays cover (a snapshot taken before bodies were appended covers fewer than exist now)
    const uint32_t n = (uint32_t)std::min<size_t>(s.bodies.size(), count);
    if (s.shadow.size() < (size_t)13 * s.bodies.size()) s.shadow.resize((size_t)13 * s.bodies.size(), 0.f);
    for (uint32_t i = 0; i < n; ++i) {
        const entt::entity e = s.bodies[i];
        if (e == entt::null || !registry.valid(e) || !registry.all_of<dynamic_tag>(e)) continue;
        auto &p = registry.get<position>(e); p.x = pos[3 * i]; p.y = pos[3 * i + 1]; p.z = pos[3 * i + 2];
        auto &q = registry.get<orientation>(e); q.x = orn[4 * i]; q.y = orn[4 * i + 1]; q.z = orn[4 * i + 2]; q.w = orn[4 * i + 3];
        auto &v = registry.get<linvel>(e); v.x = lv[3 * i]; v.y = lv[3 * i + 1]; v.z = lv[3 * i + 2];
        auto &w = registry.get<angvel>(e); w.x = av[3 * i]; w.y = av[3 * i + 1]; w.z = av[3 * i + 2];
        float *sh = &s.shadow[(size_t)13 * i];
        sh[0] = p.x; sh[1] = p.y; sh[2] = p.z; sh[3] = q.x; sh[4] = q.y; sh[5] = q.z; sh[6] = q.w;
        sh[7] = v.x; sh[8] = v.y; sh[9] = v.z; sh[10] = w.x; sh[11] = w.y; sh[12] = w.z;
    }
}
// Asynchronous mode with user edits pending (edyn::refresh, rigidbody_apply_impulse, set_kinematic_* between two updates): the
// registry is one update behind the device, so neither "import the snapshot" (the edits would be overwritten) nor "upload the
// registry" (every other body would be rewound by an update) is right. The reference sends such edits to the simulation worker,
// which applies them to ITS current state (simulation_worker.cpp, registry operations). Here: fetch the device's current state
// (this one update waits for the GPU), find the bodies whose registry state differs from what this shim last wrote there (the
// shadow) - those were edited -, carry a velocity edit over as an increment on the current velocity (an impulse stays an
// impulse) and a transform edit as the new transform, bring the registry up to date, and let upload_state push the result.
inline void merge_user_edits(entt::registry &registry, gpu_stepper &s) {
    const uint32_t n = std::min<uint32_t>((uint32_t)s.bodies.size(), s.uploaded_bodies);
    if (n == 0) return;
    std::vector<float> pos(3 * (size_t)s.uploaded_bodies), orn(4 * (size_t)s.uploaded_bodies), lv(3 * (size_t)s.uploaded_bodies), av(3 * (size_t)s.uploaded_bodies);
    check(s, edynhip_get_state(s.ctx, pos.data(), orn.data(), lv.data(), av.data()));
    if (s.shadow.size() < (size_t)13 * s.bodies.size()) s.shadow.resize((size_t)13 * s.bodies.size(), 0.f);
    for (uint32_t i = 0; i < n; ++i) {
        const entt::entity e = s.bodies[i];
        if (e == entt::null || !registry.valid(e) || !registry.all_of<dynamic_tag>(e)) continue;   // static / kinematic: the registry is the authority
        const float *sh = &s.shadow[(size_t)13 * i];
        auto &p = registry.get<position>(e); auto &q = registry.get<orientation>(e);
        auto &v = registry.get<linvel>(e); auto &w = registry.get<angvel>(e);
        const bool moved = p.x != sh[0] || p.y != sh[1] || p.z != sh[2] || q.x != sh[3] || q.y != sh[4] || q.z != sh[5] || q.w != sh[6];
        if (moved) { pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z; orn[4 * i] = q.x; orn[4 * i + 1] = q.y; orn[4 * i + 2] = q.z; orn[4 * i + 3] = q.w; }
        lv[3 * i] += v.x - sh[7]; lv[3 * i + 1] += v.y - sh[8]; lv[3 * i + 2] += v.z - sh[9];
        av[3 * i] += w.x - sh[10]; av[3 * i + 1] += w.y - sh[11]; av[3 * i + 2] += w.z - sh[12];
    }
    import_state(registry, s, pos, orn, lv, av, n);   // the registry (and the shadow) now hold the current state plus the edits
}
// asynchronous mode: the record snapshot of the previous update -> registry (state, presentation, sleeping tags, contact entities)
inline void import_pending_records(entt::registry &registry, gpu_stepper &s, const edynhip_record_view &view, uint32_t steps_in_flight) {
    { phase_timer t(s.tm.write_back); import_records(registry, s, view, s.present_this_call); }
    if (!s.cfg.materialize_contacts) return;
    phase_timer t(s.tm.contacts);
    std::vector<edynhip_contact_event> ev;
    if (view.step_index <= s.events_stale_through) {
        // a rebuild from the manifolds (below) already covered these steps: their events would be applied twice
    } else if (view.total_events > view.num_events) {
        // more events than travel with a snapshot (a pile hitting the ground): the device's list has been reset by the steps enqueued
        // since, so the contact entities are rebuilt from the CURRENT manifolds - which include the steps in flight, whose events are
        // then skipped when their snapshot arrives
        s.contacts_resync = true;
        s.events_stale_through = view.step_index + steps_in_flight;
    } else if (view.num_events) ev.assign(view.events, view.events + view.num_events);
    apply_contact_events(registry, s, ev, EDYNHIP_OK);
}
inline void run_steps(entt::registry &registry, gpu_stepper &s, unsigned steps, bool timed = false, double first_time = 0, double step_dt = 0) {
    const bool async = s.cfg.execution_mode == execution_mode::asynchronous;
    if (async && s.records_pending && (s.state_dirty || s.scene_dirty || steps == 0 || s.pre_step || s.post_step)) {
        // execution_mode::asynchronous, the cases that cannot overlap: user edits pending (the snapshot is superseded by the device's
        // current state + the edits), a scene change (the context may be re-created), no step to overlap with
        if (s.state_dirty) {
            { phase_timer t(s.tm.write_back); merge_user_edits(registry, s); }
            { phase_timer t(s.tm.contacts); sync_contacts(registry, s); }
        } else {
            edynhip_record_view view{};
            { phase_timer t(s.tm.state_wait); check(s, edynhip_snapshot_map(s.ctx, &view)); }
            import_pending_records(registry, s, view, 0u);
        }
        s.records_pending = false;
    }
    if (s.multi()) {   // init_config::devices: the multi-GPU world (see upload_scene_multi)
        if (async || s.pre_step || s.post_step || s.cfg.contact_point_data)
            throw stepper_error(EDYNHIP_ERR_UNSUPPORTED, "edyn: asynchronous mode, step callbacks and contact entities are single-device features (init_config::devices)");
        const size_t gone_before = (size_t)std::count(s.bodies.begin(), s.bodies.end(), entt::entity{entt::null}) + (size_t)std::count(s.constraints.begin(), s.constraints.end(), entt::entity{entt::null});
        sync_removed(registry, s);
        const size_t gone_after = (size_t)std::count(s.bodies.begin(), s.bodies.end(), entt::entity{entt::null}) + (size_t)std::count(s.constraints.begin(), s.constraints.end(), entt::entity{entt::null});
        if (gone_after != gone_before || s.params_dirty) s.scene_dirty = true;   // settings travel with the world's creation
        s.params_dirty = false;
        if (s.bodies.empty()) return;
        if (s.scene_dirty || s.state_dirty || !s.world) upload_scene_multi(registry, s);
        if (steps == 0) return;
        check(s, edynhip_world_step(s.world, steps));
        const uint32_t n = (uint32_t)s.bodies.size();
        std::vector<float> pos(3 * (size_t)n), orn(4 * (size_t)n), lv(3 * (size_t)n), av(3 * (size_t)n);
        check(s, edynhip_world_get_state(s.world, pos.data(), orn.data(), lv.data(), av.data()));
        import_state(registry, s, pos, orn, lv, av, n);
        s.host_presentation = true;   // (the multi-GPU world hands back plain state arrays)
        return;
    }
    { phase_timer t(s.tm.sync_removed); sync_removed(registry, s); }
    {
        phase_timer t(s.tm.upload);
        if (s.scene_dirty) upload_scene(registry, s);
        if (s.state_dirty) upload_state(registry, s);   // also after an append: edits made in the same frame are not lost
        apply_params(s);
    }
    if (steps == 0 || s.bodies.empty()) { s.host_presentation = true; return; }   // time went on without a step: presentation from the registry's state
    s.tm.steps += steps;
    if (s.pre_step || s.post_step) {
        // step callbacks (stepper_sequential.cpp:76-78,97-99) see the registry between steps: one step per launch, the state
        // written back after each, edits made by a callback (followed by edyn::refresh) uploaded before the next
        if (s.cfg.materialize_contacts && !s.prefetch_on) { check(s, edynhip_set_event_prefetch(s.ctx, prefetch_events_max)); s.prefetch_on = true; }   // (callbacks set on a running asynchronous world)
        for (unsigned k = 0; k < steps; ++k) {
            if (s.pre_step) s.pre_step(registry);
            if (s.state_dirty) upload_state(registry, s);
            check(s, timed ? edynhip_step_timed(s.ctx, 1, first_time + step_dt * k, step_dt) : edynhip_step(s.ctx, 1));
            write_back(registry, s, s.present_this_call && k + 1 == steps);
            if (s.post_step) s.post_step(registry);
        }
        return;
    }
    { phase_timer t(s.tm.step_call); check(s, timed ? edynhip_step_timed(s.ctx, steps, first_time, step_dt) : edynhip_step(s.ctx, steps)); }
    if (async) {
        // The registry receives the PREVIOUS update's result while this update's steps run on the device (the simulation worker's
        // snapshots, simulation_worker.cpp:406-444): its copy finished long ago - it was enqueued before these steps -, so nothing
        // here waits for the GPU, and the GPU does not wait for the host loop below.
        edynhip_record_view prev{};
        const bool have_prev = s.records_pending;
        if (have_prev) { phase_timer t(s.tm.state_wait); check(s, edynhip_snapshot_map(s.ctx, &prev)); }
        // every contact event of these steps travels with the snapshot (as many as a slot holds)
        check(s, edynhip_snapshot_records(s.ctx, s.present_dt, s.cfg.materialize_contacts ? 0xFFFFFFFFu : 0u, 0u));   // (copy engine, side stream: nobody waits for it)
        s.records_pending = true;
        if (have_prev) { pool_of(s).prewake(); import_pending_records(registry, s, prev, steps); }
        return;
    }
    write_back(registry, s, s.present_this_call);
}
// update_presentation (src/edyn/sys/update_presentation.cpp:56-84), local simulation (no discontinuities): transforms are
// extrapolated from the last simulated state to `presentation_delay` = fixed_dt behind the current time.
inline quaternion integrate(const quaternion &q, const vector3 &w, scalar dt) {   // math/quaternion.cpp:7-22
    const scalar ws = std::sqrt(w.x * w.x + w.y * w.y + w.z * w.z);
    const scalar half = scalar(0.5);
    scalar t;
    if (ws < scalar(0.001)) t = half * dt - dt * dt * dt * (scalar(1) / scalar(48)) * ws * ws;
    else t = (scalar)std::sin((double)(half * ws * dt)) / ws;   // sin / cos through double, rounded once: what the device's integrate() computes
    const quaternion r{w.x * t, w.y * t, w.z * t, (scalar)std::cos((double)(half * ws * dt))};
    quaternion o{r.w * q.x + r.x * q.w + r.y * q.z - r.z * q.y, r.w * q.y + r.y * q.w + r.z * q.x - r.x * q.z,
                 r.w * q.z + r.z * q.w + r.x * q.y - r.y * q.x, r.w * q.w - r.x * q.x - r.y * q.y - r.z * q.z};
    const scalar l = std::sqrt(o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w);
    o.x /= l; o.y /= l; o.z /= l; o.w /= l;
    return o;
}
/// interpolation_dt of update_presentation (update_presentation.cpp:70-71) for the state the registry will hold after this update
inline scalar presentation_dt(const gpu_stepper &s, double time) {
    const double sim_time = time - s.accumulated;   // get_simulation_timestamp() once m_last_time = time
    return std::min(static_cast<scalar>(time - s.cfg.fixed_dt - sim_time), s.cfg.fixed_dt);
}
// The host form of update_presentation: used when the update ran no step (time went on, the state did not) and by the multi-GPU
// world; an update that steps gets the same transforms from the device with the state (edynhip_body_record::present_*).
inline void update_presentation(entt::registry &registry, gpu_stepper &s, scalar idt) {
    const uint32_t n = (uint32_t)s.bodies.size();
    auto &sp = registry.storage<position>(); auto &sq = registry.storage<orientation>();
    auto &sv = registry.storage<linvel>(); auto &sw = registry.storage<angvel>();
    auto &spp = registry.storage<present_position>(); auto &spo = registry.storage<present_orientation>();
    auto &sleeping = registry.storage<sleeping_tag>();
    pool_of(s).parallel_for(n, write_back_grain, [&](uint32_t begin, uint32_t end, unsigned) {
        for (uint32_t i = begin; i < end; ++i) {
            const entt::entity e = s.bodies[i];
            if (e == entt::null || !spp.contains(e) || !spo.contains(e) || sleeping.contains(e)) continue;   // destroyed bodies keep their index
            if (!sp.contains(e) || !sq.contains(e) || !sv.contains(e) || !sw.contains(e)) continue;
            const auto &p = sp.get(e); const auto &q = sq.get(e); const auto &v = sv.get(e); const auto &w = sw.get(e);
            auto &pp = spp.get(e);
            pp.x = p.x + v.x * idt; pp.y = p.y + v.y * idt; pp.z = p.z + v.z * idt;
            const quaternion o = integrate(q, w, idt);
            auto &po = spo.get(e);
            po.x = o.x; po.y = o.y; po.z = o.z; po.w = o.w;
        }
    });
}
/// snap_presentation (update_presentation.cpp:86-92): what a paused stepper does instead (stepper_sequential.cpp:38-43)
inline void snap_presentation(entt::registry &registry, gpu_stepper &s) {
    auto &sp = registry.storage<position>(); auto &sq = registry.storage<orientation>();
    auto &spp = registry.storage<present_position>(); auto &spo = registry.storage<present_orientation>();
    pool_of(s).parallel_for((uint32_t)s.bodies.size(), write_back_grain, [&](uint32_t begin, uint32_t end, unsigned) {
        for (uint32_t i = begin; i < end; ++i) {
            const entt::entity e = s.bodies[i];
            if (e == entt::null || !spp.contains(e) || !spo.contains(e) || !sp.contains(e) || !sq.contains(e)) continue;
            const auto &p = sp.get(e); const auto &q = sq.get(e);
            auto &pp = spp.get(e); pp.x = p.x; pp.y = p.y; pp.z = p.z;
            auto &po = spo.get(e); po.x = q.x; po.y = q.y; po.z = q.z; po.w = q.w;
        }
    });
}
}  // namespace detail

// ---- edyn.hpp:66-150
inline void attach(entt::registry &registry, const init_config &config = {}) {
    auto &s = registry.ctx().emplace<detail::gpu_stepper>();
    s.cfg = config;
    // destroyed bodies / constraints are noticed through on_destroy hooks, like the reference's island manager (island_manager.cpp:24-27)
    detail::connect_removal_hooks<rigidbody_tag, detail::body_index, point_constraint, hinge_constraint, distance_constraint, soft_distance_constraint,
                                  generic_constraint, null_constraint, gravity_constraint, cone_constraint, cvjoint_constraint>(registry, s);
    s.hooks_connected = true;
}
inline void detach(entt::registry &registry) {   // edyn.cpp:148-197: the stepper goes, and every entity the engine created with it
    if (auto *s = registry.ctx().find<detail::gpu_stepper>()) {
        if (s->hooks_connected)
            detail::disconnect_removal_hooks<rigidbody_tag, detail::body_index, point_constraint, hinge_constraint, distance_constraint, soft_distance_constraint,
                                             generic_constraint, null_constraint, gravity_constraint, cone_constraint, cvjoint_constraint>(registry, *s);
        s->hooks_connected = false;
        for (auto &kv : s->point_entities) if (registry.valid(kv.second)) registry.destroy(kv.second);
        for (auto &kv : s->manifold_entities) if (registry.valid(kv.second)) registry.destroy(kv.second);
        s->point_entities.clear(); s->manifold_entities.clear();
    }
    registry.ctx().erase<detail::gpu_stepper>();
}
inline scalar get_fixed_dt(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().cfg.fixed_dt; }
inline void set_fixed_dt(entt::registry &registry, scalar dt) { auto &s = registry.ctx().get<detail::gpu_stepper>(); s.cfg.fixed_dt = dt; s.params_dirty = true; }
inline void set_max_steps_per_update(entt::registry &registry, unsigned n) { registry.ctx().get<detail::gpu_stepper>().cfg.max_steps_per_update = n; }
inline unsigned get_max_steps_per_update(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().cfg.max_steps_per_update; }
inline execution_mode get_execution_mode(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().cfg.execution_mode; }
inline bool is_paused(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().paused; }
inline void set_paused(entt::registry &registry, bool paused) { auto &s = registry.ctx().get<detail::gpu_stepper>(); s.paused = paused; s.accumulated = 0; }
inline vector3 get_gravity(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().cfg.gravity; }
inline void set_gravity(entt::registry &registry, vector3 g) {   // gravity_util.cpp:12-20: the setting and every body's gravity component
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    s.cfg.gravity = g; s.params_dirty = true;
    for (const entt::entity e : s.bodies) if (e != entt::null && registry.valid(e)) if (auto *gr = registry.try_get<gravity>(e)) { gr->x = g.x; gr->y = g.y; gr->z = g.z; }
}
inline unsigned get_solver_velocity_iterations(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().cfg.num_solver_velocity_iterations; }
inline void set_solver_velocity_iterations(entt::registry &registry, unsigned n) { auto &s = registry.ctx().get<detail::gpu_stepper>(); s.cfg.num_solver_velocity_iterations = n; s.params_dirty = true; }
inline unsigned get_solver_position_iterations(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().cfg.num_solver_position_iterations; }
inline void set_solver_position_iterations(entt::registry &registry, unsigned n) { auto &s = registry.ctx().get<detail::gpu_stepper>(); s.cfg.num_solver_position_iterations = n; s.params_dirty = true; }
/// Tell the stepper that position/orientation/linvel/angvel were edited by the user (registry.patch analogue).
/// edyn::set_pre_step_callback / set_post_step_callback (edyn.hpp, context/step_callback.hpp): called before / after every fixed step.
using step_callback_t = void (*)(entt::registry &);
inline void set_pre_step_callback(entt::registry &registry, step_callback_t func) { registry.ctx().get<detail::gpu_stepper>().pre_step = func; }
inline void set_post_step_callback(entt::registry &registry, step_callback_t func) { registry.ctx().get<detail::gpu_stepper>().post_step = func; }
inline void refresh(entt::registry &registry) { registry.ctx().get<detail::gpu_stepper>().state_dirty = true; }

/// stepper_sequential::update (stepper_sequential.cpp:28-119): fixed-dt accumulator; when more steps are due than
/// max_steps_per_update, the steps that do run carry stretched time stamps (:60-66; they feed the island sleep timers, the
/// solver always integrates with fixed_dt).
inline void update(entt::registry &registry, double time) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    if (s.paused) { detail::run_steps(registry, s, 0); detail::snap_presentation(registry, s); return; }
    detail::phase_timer whole(s.tm.total);
    ++s.tm.updates;
    const double sim_time = s.last_time - s.accumulated;   // get_simulation_timestamp() before this update
    const double elapsed = std::max(time - s.last_time, 0.0);
    s.accumulated += elapsed;
    const double dt = s.cfg.fixed_dt;
    const auto num_steps = static_cast<uint64_t>(std::floor(s.accumulated / dt));
    const double advance_dt = static_cast<double>(num_steps) * dt;
    s.accumulated -= advance_dt;
    uint64_t effective_steps = num_steps;
    double step_dt = dt;
    if (effective_steps > s.cfg.max_steps_per_update) {
        effective_steps = s.cfg.max_steps_per_update;
        step_dt = advance_dt / static_cast<double>(effective_steps);
    }
    s.present_dt = detail::presentation_dt(s, time);   // travels to the device with the write-back request
    s.host_presentation = false;
    detail::run_steps(registry, s, (unsigned)effective_steps, true, sim_time, step_dt);
    s.last_time = time;
    if (s.host_presentation) { detail::phase_timer t(s.tm.presentation); detail::update_presentation(registry, s, s.present_dt); }
}
/// Host-side cost of the shim per phase since attach / the last reset (shim_timings above).
inline shim_timings get_shim_timings(entt::registry &registry) { return registry.ctx().get<detail::gpu_stepper>().tm; }
inline void reset_shim_timings(entt::registry &registry) { registry.ctx().get<detail::gpu_stepper>().tm = shim_timings{}; }
namespace detail {
inline double performance_time() {   // time/time.hpp performance_time(): seconds on a monotonic clock
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace detail
/// edyn::update(registry) (edyn.hpp:124, edyn.cpp:234-238): the time comes from settings.time_func (a monotonic clock).
inline double get_time(entt::registry &registry) {   // edyn.hpp:171, edyn.cpp:289-293
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    return s.time_func ? s.time_func() : detail::performance_time();
}
inline void set_time_source(entt::registry &registry, double (*time_func)(void)) { registry.ctx().get<detail::gpu_stepper>().time_func = time_func; }   // edyn.hpp:164
inline void update(entt::registry &registry) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const double now = get_time(registry);
    if (s.last_time == 0 && s.accumulated == 0) s.last_time = now;   // attach() stamps the stepper with the current time
    update(registry, now);
}
/// stepper_sequential::step_simulation (stepper_sequential.cpp:121-147): exactly one step; requires paused.
inline void step_simulation(entt::registry &registry, double time) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    s.last_time = time;
    struct restore { bool &flag; ~restore() { flag = true; } } guard{s.present_this_call};   // (also when a step throws)
    s.present_this_call = false;
    detail::run_steps(registry, s, 1, true, time, s.cfg.fixed_dt);
}
/// edyn::step_simulation(registry) (edyn.hpp:142).
inline void step_simulation(entt::registry &registry) { step_simulation(registry, get_time(registry)); }

// ---- util/rigidbody.hpp:84-93, rigidbody.cpp:47-191
inline void make_rigidbody(entt::entity entity, entt::registry &registry, const rigidbody_def &def) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    registry.emplace<position>(entity, position{def.position});
    registry.emplace<orientation>(entity, orientation{def.orientation});
    if (def.kind == rigidbody_kind::rb_dynamic) {
        registry.emplace<mass>(entity, mass{def.mass});
        registry.emplace<mass_inv>(entity, mass_inv{scalar(1) / def.mass});
        if (def.inertia) registry.emplace<inertia>(entity, inertia{*def.inertia});
    }
    if (def.kind != rigidbody_kind::rb_static) {
        registry.emplace<linvel>(entity, linvel{def.linvel});
        registry.emplace<angvel>(entity, angvel{def.angvel});
    }
    if (def.kind == rigidbody_kind::rb_dynamic && def.presentation) {   // rigidbody.cpp:71-75
        registry.emplace<present_position>(entity, present_position{def.position});
        registry.emplace<present_orientation>(entity, present_orientation{def.orientation});
    }
    if (def.center_of_mass && (def.center_of_mass->x != 0 || def.center_of_mass->y != 0 || def.center_of_mass->z != 0)) {   // apply_center_of_mass, rigidbody.cpp:517-548
        const vector3 c = *def.center_of_mass;
        const quaternion &q = def.orientation;
        const vector3 u{q.x, q.y, q.z};
        const vector3 t{2 * (u.y * c.z - u.z * c.y), 2 * (u.z * c.x - u.x * c.z), 2 * (u.x * c.y - u.y * c.x)};
        const vector3 rc{c.x + q.w * t.x + (u.y * t.z - u.z * t.y), c.y + q.w * t.y + (u.z * t.x - u.x * t.z), c.z + q.w * t.z + (u.x * t.y - u.y * t.x)};   // rotate(orn, com)
        registry.emplace<center_of_mass>(entity, center_of_mass{c});
        registry.emplace<origin>(entity, origin{def.position});
        auto &p = registry.get<position>(entity); p.x += rc.x; p.y += rc.y; p.z += rc.z;
        if (def.kind != rigidbody_kind::rb_static) {
            auto &v = registry.get<linvel>(entity); const auto &w = registry.get<angvel>(entity);
            v.x += w.y * rc.z - w.z * rc.y; v.y += w.z * rc.x - w.x * rc.z; v.z += w.x * rc.y - w.y * rc.x;
        }
    }
    const vector3 g = def.gravity ? *def.gravity : s.cfg.gravity;
    if (def.kind == rigidbody_kind::rb_dynamic) registry.emplace<gravity>(entity, gravity{g});
    if (def.material) registry.emplace<material>(entity, *def.material);
    if (def.shape) {
        std::visit([&](auto &&sh) { registry.emplace<std::decay_t<decltype(sh)>>(entity, sh); }, *def.shape);
        if (def.collision_group != ~0ull || def.collision_mask != ~0ull) registry.emplace<collision_filter>(entity, collision_filter{def.collision_group, def.collision_mask});
    }
    if (def.sleeping_disabled) registry.emplace<sleeping_disabled_tag>(entity);
    switch (def.kind) {
    case rigidbody_kind::rb_dynamic: registry.emplace<dynamic_tag>(entity); registry.emplace<procedural_tag>(entity); break;
    case rigidbody_kind::rb_kinematic: registry.emplace<kinematic_tag>(entity); break;
    case rigidbody_kind::rb_static: registry.emplace<static_tag>(entity); break;
    }
    registry.emplace<detail::body_index>(entity, detail::body_index{(uint32_t)s.bodies.size()});
    s.bodies.push_back(entity);
    s.scene_dirty = true;
    registry.emplace<rigidbody_tag>(entity);
}
inline entt::entity make_rigidbody(entt::registry &registry, const rigidbody_def &def) {
    auto e = registry.create();
    make_rigidbody(e, registry, def);
    return e;
}

// ---- util/constraint_util.hpp:38-54: make_constraint<T>(registry, entity, body0, body1, setup...) and the entity-creating form
template <typename T, typename... SetupFunc>
void make_constraint(entt::registry &registry, entt::entity entity, entt::entity body0, entt::entity body1, SetupFunc... setup) {
    static_assert(std::is_same_v<T, point_constraint> || std::is_same_v<T, hinge_constraint> || std::is_same_v<T, distance_constraint> ||
                      std::is_same_v<T, soft_distance_constraint> || std::is_same_v<T, cone_constraint> || std::is_same_v<T, cvjoint_constraint> ||
                      std::is_same_v<T, gravity_constraint> || std::is_same_v<T, generic_constraint> || std::is_same_v<T, null_constraint>,
                  "unknown constraint type");
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    auto &con = registry.emplace<T>(entity);
    con.body = {body0, body1};
    (setup(con), ...);
    s.constraints.push_back(entity);
    s.constraint_kind.push_back((uint8_t)detail::joint_kind_of<T>());
    s.scene_dirty = true;
}
template <typename T, typename... SetupFunc>
entt::entity make_constraint(entt::registry &registry, entt::entity body0, entt::entity body1, SetupFunc... setup) {
    auto e = registry.create();
    make_constraint<T>(registry, e, body0, body1, setup...);
    return e;
}

// ---- collision/should_collide.hpp:8-18: the user's predicate that replaces should_collide_default for NEW manifolds (a host callback:
// steps that have new candidate pairs take the slow path of edynhip_set_pair_filter; worlds over several devices: edynhip_world_set_pair_filter).
using should_collide_func_t = bool (*)(const entt::registry &, entt::entity, entt::entity);
/// collision groups / masks and exclusion lists, evaluated on the host (should_collide.cpp:11-57) - for predicates that extend the default
inline bool should_collide_default(const entt::registry &registry, entt::entity first, entt::entity second) {
    if (first == second) return false;
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const uint32_t a = registry.get<detail::body_index>(first).value, b = registry.get<detail::body_index>(second).value;
    uint64_t ga = ~0ull, ma = ~0ull, gb = ~0ull, mb = ~0ull;
    if (auto *f = registry.try_get<collision_filter>(first)) { ga = f->group; ma = f->mask; }
    if (auto *f = registry.try_get<collision_filter>(second)) { gb = f->group; mb = f->mask; }
    if ((ga & mb) == 0 || (gb & ma) == 0) return false;
    for (auto &ex : s.exclusions) if ((ex[0] == a && ex[1] == b) || (ex[0] == b && ex[1] == a)) return false;
    return true;
}
inline void set_should_collide(entt::registry &registry, should_collide_func_t func) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    if (func == &should_collide_default) func = nullptr;   // the device's own test
    s.should_collide = func; s.filter_registry = &registry;
    if (s.ctx) detail::check(s, edynhip_set_pair_filter(s.ctx, func ? &detail::pair_filter_trampoline : nullptr, &s));
    if (s.world) detail::check(s, edynhip_world_set_pair_filter(s.world, func ? &detail::pair_filter_trampoline : nullptr, &s));   // global body indices = the shim's
}

// ---- util/exclude_collision.hpp:20-47
inline void exclude_collision(entt::registry &registry, entt::entity first, entt::entity second) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const uint32_t a = registry.get<detail::body_index>(first).value, b = registry.get<detail::body_index>(second).value;
    // recorded for good (a re-created context gets the whole list again); sent at once when the device already holds everything
    // recorded before it, else with the next upload
    s.exclusions.push_back({a, b});
    if (s.ctx && a < s.uploaded_bodies && b < s.uploaded_bodies && !s.scene_dirty && s.exclusions_uploaded + 1 == s.exclusions.size()) {
        detail::check(s, edynhip_exclude_collision(s.ctx, a, b));
        s.exclusions_uploaded = s.exclusions.size();
    } else s.scene_dirty = true;
}
inline void remove_collision_exclusion(entt::registry &registry, entt::entity first, entt::entity second) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const uint32_t a = registry.get<detail::body_index>(first).value, b = registry.get<detail::body_index>(second).value;
    bool on_device = false;
    for (size_t k = s.exclusions.size(); k; --k) {
        auto &ex = s.exclusions[k - 1];
        if ((ex[0] == a && ex[1] == b) || (ex[0] == b && ex[1] == a)) {
            if (k - 1 < s.exclusions_uploaded) { on_device = true; --s.exclusions_uploaded; }
            s.exclusions.erase(s.exclusions.begin() + (k - 1));
        }
    }
    if (on_device && s.ctx && a < s.uploaded_bodies && b < s.uploaded_bodies) detail::check(s, edynhip_remove_collision_exclusion(s.ctx, a, b));
}
/// util/exclude_collision.hpp:23 (entity_pair form) and :38 clear_collision_exclusion (exclude_collision.cpp:59-69): every exclusion `entity` takes part in
using entity_pair = std::pair<entt::entity, entt::entity>;   // core/entity_pair.hpp
inline void exclude_collision(entt::registry &registry, entity_pair entities) { exclude_collision(registry, entities.first, entities.second); }
inline void clear_collision_exclusion(entt::registry &registry, entt::entity entity) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const auto *bi = registry.try_get<detail::body_index>(entity);
    if (!bi) return;
    std::vector<entt::entity> others;
    for (const auto &ex : s.exclusions) if (ex[0] == bi->value || ex[1] == bi->value) others.push_back(s.bodies[ex[0] == bi->value ? ex[1] : ex[0]]);
    for (const entt::entity o : others) if (o != entt::null) remove_collision_exclusion(registry, entity, o);
}
/// util/constraint_util.hpp:61 (constraint_util.cpp:53-58): the entity stops being a constraint (it lives on); the stepper drops the joint at the next update
inline void clear_constraint(entt::registry &registry, entt::entity entity) {
    registry.remove<point_constraint>(entity); registry.remove<hinge_constraint>(entity); registry.remove<distance_constraint>(entity);
    registry.remove<soft_distance_constraint>(entity); registry.remove<generic_constraint>(entity); registry.remove<null_constraint>(entity);
    registry.remove<gravity_constraint>(entity); registry.remove<cone_constraint>(entity); registry.remove<cvjoint_constraint>(entity);
    registry.ctx().get<detail::gpu_stepper>().removal_pending = true;   // (also without signals: the bundled registry's sinks cover destroy / remove alike)
}
/// util/rigidbody.hpp:140 (rigidbody.cpp:289-298): does the entity carry what a rigid body of this stepper needs?
inline bool validate_rigidbody(entt::registry &registry, entt::entity &entity) {
    if (!registry.valid(entity) || !registry.all_of<rigidbody_tag, detail::body_index>(entity)) return false;
    if (!registry.any_of<dynamic_tag, kinematic_tag, static_tag>(entity)) return false;
    if (!registry.all_of<position, orientation>(entity)) return false;
    if (!registry.all_of<static_tag>(entity) && !registry.all_of<linvel, angvel>(entity)) return false;
    if (registry.all_of<dynamic_tag>(entity) && !registry.all_of<mass, mass_inv>(entity)) return false;
    return true;
}
/// util/contact_manifold_util.hpp:19-35: is there a contact manifold between the two bodies / which entity is it (init_config::materialize_contacts)
inline entt::entity get_manifold_entity(entt::registry &registry, entt::entity first, entt::entity second) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const auto *a = registry.try_get<detail::body_index>(first), *b = registry.try_get<detail::body_index>(second);
    if (!a || !b) return entt::null;
    for (const uint64_t key : {detail::manifold_key(a->value, b->value), detail::manifold_key(b->value, a->value)}) {
        auto it = s.manifold_entities.find(key);
        if (it != s.manifold_entities.end()) return it->second;
    }
    return entt::null;
}
inline entt::entity get_manifold_entity(entt::registry &registry, entity_pair entities) { return get_manifold_entity(registry, entities.first, entities.second); }
inline bool manifold_exists(entt::registry &registry, entt::entity first, entt::entity second) { return get_manifold_entity(registry, first, second) != entt::null; }
inline bool manifold_exists(entt::registry &registry, entity_pair entities) { return manifold_exists(registry, entities.first, entities.second); }
/// util/constraint_util.hpp:73-105: the edges of a body in the island graph - its constraint entities and (with materialize_contacts) its contact manifold
/// entities - and the bodies at their other ends. func(entity), or bool func(entity) returning false to stop.
template <typename Func> void visit_edges(entt::registry &registry, entt::entity entity, Func func) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    auto call = [&](entt::entity e) { if constexpr (std::is_invocable_r_v<bool, Func, entt::entity>) return func(e); else { func(e); return true; } };
    for (uint32_t j = 0; j < (uint32_t)s.constraints.size(); ++j) {
        const entt::entity c = s.constraints[j];
        if (c == entt::null) continue;
        const constraint_base *cb = detail::constraint_of(registry, c, s.constraint_kind[j]);
        if (cb && (cb->body[0] == entity || cb->body[1] == entity) && !call(c)) return;
    }
    const auto *bi = registry.try_get<detail::body_index>(entity);
    if (!bi) return;
    for (auto &kv : s.manifold_entities) {
        const uint64_t key = detail::entity_table::key_of(kv);
        if (((uint32_t)(key >> 32) == bi->value || (uint32_t)key == bi->value) && !call(kv.second)) return;
    }
}
template <typename Func> void visit_neighbors(entt::registry &registry, entt::entity entity, Func func) {
    visit_edges(registry, entity, [&](entt::entity edge) {
        entt::entity other = entt::null;
        if (auto *m = registry.try_get<contact_manifold>(edge)) other = m->body[0] == entity ? m->body[1] : m->body[0];
        else {
            auto &s = registry.ctx().get<detail::gpu_stepper>();
            for (uint32_t j = 0; j < (uint32_t)s.constraints.size() && other == entt::null; ++j)
                if (s.constraints[j] == edge) if (const constraint_base *cb = detail::constraint_of(registry, edge, s.constraint_kind[j])) other = cb->body[0] == entity ? cb->body[1] : cb->body[0];
        }
        if (other != entt::null) func(other);
    });
}
/// util/rigidbody.hpp:95-103, rigidbody.cpp:193-226: strips everything make_rigidbody assigned; the entity itself lives on.
inline void clear_rigidbody(entt::registry &registry, entt::entity entity) {
    registry.remove<rigidbody_tag>(entity); registry.remove<dynamic_tag>(entity); registry.remove<kinematic_tag>(entity);
    registry.remove<static_tag>(entity); registry.remove<procedural_tag>(entity); registry.remove<sleeping_disabled_tag>(entity);
    registry.remove<sleeping_tag>(entity); registry.remove<collision_filter>(entity); registry.remove<box_shape>(entity);
    registry.remove<sphere_shape>(entity); registry.remove<plane_shape>(entity); registry.remove<capsule_shape>(entity); registry.remove<cylinder_shape>(entity); registry.remove<polyhedron_shape>(entity); registry.remove<material>(entity);
    registry.remove<gravity>(entity); registry.remove<center_of_mass>(entity); registry.remove<origin>(entity); registry.remove<linvel>(entity); registry.remove<angvel>(entity); registry.remove<mass>(entity);
    registry.remove<mass_inv>(entity); registry.remove<inertia>(entity); registry.remove<present_position>(entity);
    registry.remove<present_orientation>(entity); registry.remove<position>(entity); registry.remove<orientation>(entity);
    registry.remove<detail::body_index>(entity);   // the stepper drops the body from the device world at the next update
}

// ---- util/rigidbody.hpp:105-260: edits of a body between updates. Velocity / transform edits go to the device with the next update
// (they mark the state dirty like edyn::refresh); mass / inertia / friction edits re-create the device context at the next update,
// carrying contacts (warm-start impulses), joints (applied impulses, tracked angles), collision exclusions and sleeping tags over (island
// sleep timers of awake islands restart) - rare operations, kept simple.
namespace detail {
inline matrix3x3 mat_mul(const matrix3x3 &a, const matrix3x3 &b) {
    matrix3x3 r{};
    auto el = [](const vector3 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; };
    for (int i = 0; i < 3; ++i) {
        scalar o[3];
        for (int k = 0; k < 3; ++k) o[k] = el(a.row[i], 0) * el(b.row[0], k) + el(a.row[i], 1) * el(b.row[1], k) + el(a.row[i], 2) * el(b.row[2], k);
        r.row[i] = {o[0], o[1], o[2]};
    }
    return r;
}
inline matrix3x3 transposed(const matrix3x3 &m) { return {{vector3{m.row[0].x, m.row[1].x, m.row[2].x}, vector3{m.row[0].y, m.row[1].y, m.row[2].y}, vector3{m.row[0].z, m.row[1].z, m.row[2].z}}}; }
inline matrix3x3 rotation_matrix(const quaternion &q) {   // math/quaternion.hpp to_matrix3x3
    const scalar d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = scalar(2) / d;
    const scalar xs = q.x * s, ys = q.y * s, zs = q.z * s, wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const scalar xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    return {{vector3{1 - (yy + zz), xy - wz, xz + wy}, vector3{xy + wz, 1 - (xx + zz), yz - wx}, vector3{xz - wy, yz + wx, 1 - (xx + yy)}}};
}
// inverse inertia tensor in world space of a dynamic body (zero for the others): the user's inertia component, or the solid
// shape's (moment_of_inertia.cpp - the capsule as the reference computes it, see capi.hip k_init_bodies)
inline matrix3x3 inertia_world_inv_of(entt::registry &registry, entt::entity e) {
    matrix3x3 zero{};
    if (!registry.all_of<dynamic_tag>(e)) return zero;
    const scalar m = registry.get<mass>(e).s;
    matrix3x3 I{};
    if (auto *in = registry.try_get<inertia>(e)) I = *in;
    else {
        vector3 d{large_scalar, large_scalar, large_scalar};
        if (auto *b = registry.try_get<box_shape>(e)) {
            const vector3 x{b->half_extents.x * 2, b->half_extents.y * 2, b->half_extents.z * 2};
            const scalar k = scalar(1) / scalar(12) * m;
            d = {k * (x.y * x.y + x.z * x.z), k * (x.z * x.z + x.x * x.x), k * (x.x * x.x + x.y * x.y)};
        } else if (auto *sp = registry.try_get<sphere_shape>(e)) {
            const scalar v = scalar(0.4) * m * sp->radius * sp->radius; d = {v, v, v};
        } else if (auto *c = registry.try_get<capsule_shape>(e)) {
            const scalar pi = scalar(3.1415926535897932384626433832795029), len = c->half_length * 2, r = c->radius;
            const scalar cv = pi * r * r * len, sv = pi * r * r * r * scalar(4) / scalar(3), cm = m * cv / (cv + sv), sm = m * sv / (cv + sv);
            const scalar cxx = scalar(0.5) * cm * r * r, cyy = scalar(1) / scalar(12) * cm * (scalar(3) * r * r + len * len), si = scalar(0.4) * sm * r * r;
            const int ax = (int)c->axis;
            const scalar cyl_x = ax == 0 ? cxx : cyy, cyl_y = ax == 1 ? cxx : cyy, tt = scalar(4) * len + scalar(3) * r;
            const scalar xx = si + cyl_x, yy = si + sm * tt * tt / scalar(64) + cyl_y;
            d = {ax == 0 ? xx : yy, ax == 1 ? xx : yy, ax == 2 ? xx : yy};
        } else if (auto *cy = registry.try_get<cylinder_shape>(e)) {   // moment_of_inertia.cpp:27-44
            const scalar len = cy->half_length * 2, r = cy->radius;
            const scalar xx = scalar(0.5) * m * r * r, yy = scalar(1) / scalar(12) * m * (scalar(3) * r * r + len * len);
            const int ax = (int)cy->axis;
            d = {ax == 0 ? xx : yy, ax == 1 ? xx : yy, ax == 2 ? xx : yy};
        }
        I = {{vector3{d.x, 0, 0}, vector3{0, d.y, 0}, vector3{0, 0, d.z}}};
        if (auto *ph = registry.try_get<polyhedron_shape>(e)) {   // moment_of_inertia_polyhedron (moment_of_inertia.cpp:93-157)
            const convex_mesh &cm = *ph->mesh;
            scalar vol = 0, xx = 0, yy = 0, zz = 0, yz = 0, zx = 0, xy = 0;
            for (size_t f = 0; f < cm.num_faces(); ++f) {
                const uint32_t first = cm.faces[2 * f], count = cm.faces[2 * f + 1];
                const vector3 v0 = cm.vertices[cm.indices[first]];
                for (uint32_t j = 1; j + 1 < count; ++j) {
                    const vector3 v1 = cm.vertices[cm.indices[first + j]], v2 = cm.vertices[cm.indices[first + j + 1]];
                    const scalar pd = v0.x * (v1.y * v2.z - v1.z * v2.y) + v0.y * (v1.z * v2.x - v1.x * v2.z) + v0.z * (v1.x * v2.y - v1.y * v2.x);
                    vol += pd;
                    const vector3 v3{v0.x + v1.x + v2.x, v0.y + v1.y + v2.y, v0.z + v1.z + v2.z};
                    xx += pd * (v0.x * v0.x + v1.x * v1.x + v2.x * v2.x + v3.x * v3.x);
                    yy += pd * (v0.y * v0.y + v1.y * v1.y + v2.y * v2.y + v3.y * v3.y);
                    zz += pd * (v0.z * v0.z + v1.z * v1.z + v2.z * v2.z + v3.z * v3.z);
                    yz += pd * (v0.y * v0.z + v1.y * v1.z + v2.y * v2.z + v3.y * v3.z);
                    zx += pd * (v0.z * v0.x + v1.z * v1.x + v2.z * v2.x + v3.z * v3.x);
                    xy += pd * (v0.x * v0.y + v1.x * v1.y + v2.x * v2.y + v3.x * v3.y);
                }
            }
            const scalar r = m / (vol / scalar(6)) / scalar(120);
            I = {{vector3{(yy + zz) * r, xy * r, zx * r}, vector3{xy * r, (zz + xx) * r, yz * r}, vector3{zx * r, yz * r, (xx + yy) * r}}};
        }
    }
    // inverse_matrix_symmetric (matrix3x3.hpp:190-218)
    const vector3 &r0 = I.row[0], &r1 = I.row[1], &r2 = I.row[2];
    const scalar det = r0.x * (r1.y * r2.z - r1.z * r2.y) + r0.y * (r1.z * r2.x - r1.x * r2.z) + r0.z * (r1.x * r2.y - r1.y * r2.x), di = scalar(1) / det;
    const scalar a11 = r0.x, a12 = r0.y, a13 = r0.z, a22 = r1.y, a23 = r1.z, a33 = r2.z;
    const scalar i11 = di * (a22 * a33 - a23 * a23), i12 = di * (a13 * a23 - a12 * a33), i13 = di * (a12 * a23 - a13 * a22);
    const scalar i22 = di * (a11 * a33 - a13 * a13), i23 = di * (a12 * a13 - a11 * a23), i33 = di * (a11 * a22 - a12 * a12);
    const matrix3x3 inv{{vector3{i11, i12, i13}, vector3{i12, i22, i23}, vector3{i13, i23, i33}}};
    const matrix3x3 basis = rotation_matrix(registry.get<orientation>(e));
    return mat_mul(mat_mul(basis, inv), transposed(basis));   // update_inertias.cpp:12-24
}
inline vector3 mat_vec(const matrix3x3 &m, const vector3 &v) {
    return {m.row[0].x * v.x + m.row[0].y * v.y + m.row[0].z * v.z, m.row[1].x * v.x + m.row[1].y * v.y + m.row[1].z * v.z, m.row[2].x * v.x + m.row[2].y * v.y + m.row[2].z * v.z};
}
}  // namespace detail
/// util/rigidbody.hpp:191: where the body's shape sits (its position is the centre of mass)
inline vector3 get_rigidbody_origin(entt::registry &registry, entt::entity entity) {
    if (auto *o = registry.try_get<origin>(entity)) return *o;
    return registry.get<position>(entity);
}
namespace detail {
inline vector3 to_world_space(const vector3 &c, const vector3 &p, const quaternion &q) {   // math/transform.hpp: p + rotate(q, c)
    const vector3 u{q.x, q.y, q.z};
    const vector3 t{2 * (u.y * c.z - u.z * c.y), 2 * (u.z * c.x - u.x * c.z), 2 * (u.x * c.y - u.y * c.x)};
    return {p.x + c.x + q.w * t.x + (u.y * t.z - u.z * t.y), p.y + c.y + q.w * t.y + (u.z * t.x - u.x * t.z), p.z + c.z + q.w * t.z + (u.x * t.y - u.y * t.x)};
}
}  // namespace detail
/// util/rigidbody.hpp:201 (rigidbody.cpp:382-391): move the body so that its ORIGIN is at `origin` (edyn::refresh is implied: the device takes the edit at the next update)
inline void set_rigidbody_origin(entt::registry &registry, entt::entity entity, const vector3 &org) {
    auto &p = registry.get<position>(entity);
    if (auto *cm = registry.try_get<center_of_mass>(entity)) {
        const vector3 w = detail::to_world_space(*cm, org, registry.get<orientation>(entity));
        p.x = w.x; p.y = w.y; p.z = w.z;
        auto &o = registry.get<origin>(entity); o.x = org.x; o.y = org.y; o.z = org.z;
    } else { p.x = org.x; p.y = org.y; p.z = org.z; }
    registry.ctx().get<detail::gpu_stepper>().state_dirty = true;
}
/// util/rigidbody.hpp:219 (rigidbody.cpp:403-407): recompute `origin` from position / orientation / centre of mass after an edit of the transform
inline void rigidbody_update_origin(entt::registry &registry, entt::entity entity) {
    const auto &cm = registry.get<center_of_mass>(entity);
    const vector3 w = detail::to_world_space(vector3{-cm.x, -cm.y, -cm.z}, registry.get<position>(entity), registry.get<orientation>(entity));
    auto &o = registry.get<origin>(entity); o.x = w.x; o.y = w.y; o.z = w.z;
}
/// util/rigidbody.hpp:198 (rigidbody.cpp:393-401): where a renderer draws the body's shape
inline vector3 get_rigidbody_present_origin(entt::registry &registry, entt::entity entity) {
    const auto &pp = registry.get<present_position>(entity);
    if (auto *cm = registry.try_get<center_of_mass>(entity)) return detail::to_world_space(vector3{-cm->x, -cm->y, -cm->z}, pp, registry.get<present_orientation>(entity));
    return pp;
}
/// util/rigidbody.hpp:182, rigidbody.cpp:364-370,517-548: the body's centre of mass moves (in the shape's frame); its origin stays
inline void set_center_of_mass(entt::registry &registry, entt::entity entity, const vector3 &com) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    const auto &q = registry.get<orientation>(entity);
    auto rot = [&](vector3 c) {
        const vector3 u{q.x, q.y, q.z};
        const vector3 t{2 * (u.y * c.z - u.z * c.y), 2 * (u.z * c.x - u.x * c.z), 2 * (u.x * c.y - u.y * c.x)};
        return vector3{c.x + q.w * t.x + (u.y * t.z - u.z * t.y), c.y + q.w * t.y + (u.z * t.x - u.x * t.z), c.z + q.w * t.z + (u.x * t.y - u.y * t.x)};
    };
    auto &p = registry.get<position>(entity);
    vector3 org{p.x, p.y, p.z};
    if (auto *old = registry.try_get<center_of_mass>(entity)) { const vector3 r = rot({-old->x, -old->y, -old->z}); org = {p.x + r.x, p.y + r.y, p.z + r.z}; }
    const vector3 rc = rot(com), cw{org.x + rc.x, org.y + rc.y, org.z + rc.z}, d{cw.x - p.x, cw.y - p.y, cw.z - p.z};
    if (auto *v = registry.try_get<linvel>(entity)) {
        const auto &w = registry.get<angvel>(entity);
        v->x += w.y * d.z - w.z * d.y; v->y += w.z * d.x - w.x * d.z; v->z += w.x * d.y - w.y * d.x;
    }
    p.x = cw.x; p.y = cw.y; p.z = cw.z;
    const bool has = com.x != 0 || com.y != 0 || com.z != 0;
    if (has) { registry.emplace_or_replace<center_of_mass>(entity, center_of_mass{com}); registry.emplace_or_replace<origin>(entity, origin{org}); }
    else { registry.remove<center_of_mass>(entity); registry.remove<origin>(entity); }
    const auto *bi = registry.try_get<detail::body_index>(entity);
    if (bi && s.ctx && bi->value < s.uploaded_bodies && !s.scene_dirty) {
        const float c3[3] = {com.x, com.y, com.z};
        detail::check(s, edynhip_set_center_of_mass(s.ctx, bi->value, c3));   // the device does the same on its copy
    }
}
/// rigidbody.cpp:228-246
inline void rigidbody_apply_impulse(entt::registry &registry, entt::entity entity, const vector3 &impulse, const vector3 &rel_location) {
    if (!registry.all_of<dynamic_tag>(entity)) return;
    const scalar m_inv = registry.get<mass_inv>(entity).s;
    auto &v = registry.get<linvel>(entity); auto &w = registry.get<angvel>(entity);
    v.x += impulse.x * m_inv; v.y += impulse.y * m_inv; v.z += impulse.z * m_inv;
    const vector3 t{rel_location.y * impulse.z - rel_location.z * impulse.y, rel_location.z * impulse.x - rel_location.x * impulse.z, rel_location.x * impulse.y - rel_location.y * impulse.x};
    const vector3 dw = detail::mat_vec(detail::inertia_world_inv_of(registry, entity), t);
    w.x += dw.x; w.y += dw.y; w.z += dw.z;
    registry.ctx().get<detail::gpu_stepper>().state_dirty = true;
}
inline void rigidbody_apply_torque_impulse(entt::registry &registry, entt::entity entity, const vector3 &torque_impulse) {
    if (!registry.all_of<dynamic_tag>(entity)) return;
    auto &w = registry.get<angvel>(entity);
    const vector3 dw = detail::mat_vec(detail::inertia_world_inv_of(registry, entity), torque_impulse);
    w.x += dw.x; w.y += dw.y; w.z += dw.z;
    registry.ctx().get<detail::gpu_stepper>().state_dirty = true;
}
/// rigidbody.cpp:248-290: a kinematic body is moved by giving it the velocity that takes it there in `dt`
inline void set_kinematic_position(entt::registry &registry, entt::entity entity, const vector3 &pos, scalar dt) {
    auto &cur = registry.get<position>(entity); auto &vel = registry.get<linvel>(entity);
    vel.x = (pos.x - cur.x) / dt; vel.y = (pos.y - cur.y) / dt; vel.z = (pos.z - cur.z) / dt;
    cur.x = pos.x; cur.y = pos.y; cur.z = pos.z;
    registry.ctx().get<detail::gpu_stepper>().state_dirty = true;
}
inline void set_kinematic_orientation(entt::registry &registry, entt::entity entity, const quaternion &orn, scalar dt) {
    auto &cur = registry.get<orientation>(entity); auto &vel = registry.get<angvel>(entity);
    const quaternion c{-cur.x, -cur.y, -cur.z, cur.w};   // r = orn * conjugate(cur): the rotation from the current orientation to the new one
    const quaternion r{orn.w * c.x + orn.x * c.w + orn.y * c.z - orn.z * c.y, orn.w * c.y - orn.x * c.z + orn.y * c.w + orn.z * c.x,
                       orn.w * c.z + orn.x * c.y - orn.y * c.x + orn.z * c.w, orn.w * c.w - orn.x * c.x - orn.y * c.y - orn.z * c.z};
    const scalar ws = std::acos(r.w) / (scalar(0.5) * dt);   // the inverse of quaternion integrate (math/quaternion.cpp:7-22)
    const scalar t = ws < scalar(0.001) ? scalar(0.5) * dt - dt * dt * dt * (scalar(1) / scalar(48)) * ws * ws : std::sin(scalar(0.5) * ws * dt) / ws;
    vel.x = r.x / t; vel.y = r.y / t; vel.z = r.z / t;
    cur.x = orn.x; cur.y = orn.y; cur.z = orn.z; cur.w = orn.w;
    registry.ctx().get<detail::gpu_stepper>().state_dirty = true;
}
/// rigidbody.cpp:301-352
inline void set_rigidbody_mass(entt::registry &registry, entt::entity entity, scalar m) {
    registry.get<mass>(entity).s = m; registry.get<mass_inv>(entity).s = scalar(1) / m;
    auto &s = registry.ctx().get<detail::gpu_stepper>(); s.recreate = s.scene_dirty = true;
}
inline void set_rigidbody_inertia(entt::registry &registry, entt::entity entity, const matrix3x3 &I) {
    registry.emplace_or_replace<inertia>(entity, inertia{I});
    auto &s = registry.ctx().get<detail::gpu_stepper>(); s.recreate = s.scene_dirty = true;
}
inline void set_rigidbody_friction(entt::registry &registry, entt::entity entity, scalar friction) {   // (existing contact points take the new value with the re-created context)
    registry.get<material>(entity).friction = friction;
    auto &s = registry.ctx().get<detail::gpu_stepper>(); s.recreate = s.scene_dirty = s.refresh_friction = true;
}
/// util/rigidbody.hpp:235-257 (rigidbody.cpp:417-515): another shape / no shape, another kind - through a re-created context like the edits above
inline bool rigidbody_has_shape(entt::registry &registry, entt::entity entity) { return registry.any_of<box_shape, sphere_shape, plane_shape, capsule_shape, cylinder_shape, polyhedron_shape>(entity); }
inline void rigidbody_set_shape(entt::registry &registry, entt::entity entity, std::optional<shapes_variant_t> shape_opt) {
    registry.remove<box_shape>(entity); registry.remove<sphere_shape>(entity); registry.remove<plane_shape>(entity); registry.remove<capsule_shape>(entity); registry.remove<cylinder_shape>(entity); registry.remove<polyhedron_shape>(entity);
    if (shape_opt) std::visit([&](auto &&sh) { registry.emplace<std::decay_t<decltype(sh)>>(entity, sh); }, *shape_opt);
    auto &s = registry.ctx().get<detail::gpu_stepper>(); s.recreate = s.scene_dirty = true;
    if (auto *bi = registry.try_get<detail::body_index>(entity)) s.reshaped.push_back(bi->value);
}
inline void rigidbody_set_kind(entt::registry &registry, entt::entity entity, rigidbody_kind kind) {
    registry.remove<dynamic_tag>(entity); registry.remove<procedural_tag>(entity); registry.remove<kinematic_tag>(entity); registry.remove<static_tag>(entity);
    if (kind == rigidbody_kind::rb_dynamic) {
        registry.emplace<dynamic_tag>(entity); registry.emplace<procedural_tag>(entity);
        if (!registry.all_of<mass>(entity)) { registry.emplace<mass>(entity, mass{1}); registry.emplace<mass_inv>(entity, mass_inv{1}); }
        if (!registry.all_of<gravity>(entity)) registry.emplace<gravity>(entity, gravity{registry.ctx().get<detail::gpu_stepper>().cfg.gravity});
    } else if (kind == rigidbody_kind::rb_kinematic) registry.emplace<kinematic_tag>(entity);
    else registry.emplace<static_tag>(entity);
    if (kind != rigidbody_kind::rb_static) {
        if (!registry.all_of<linvel>(entity)) registry.emplace<linvel>(entity, linvel{});
        if (!registry.all_of<angvel>(entity)) registry.emplace<angvel>(entity, angvel{});
    } else {   // static bodies do not move (rigidbody.cpp:480-486 zeroes the velocities)
        if (auto *v = registry.try_get<linvel>(entity)) *v = linvel{};
        if (auto *w = registry.try_get<angvel>(entity)) *w = angvel{};
    }
    auto &s = registry.ctx().get<detail::gpu_stepper>(); s.recreate = s.scene_dirty = true;
    if (auto *bi = registry.try_get<detail::body_index>(entity)) s.reshaped.push_back(bi->value);
}
/// rigidbody.cpp:409-415 -> island_manager wake_up_island
inline void wake_up_entity(entt::registry &registry, entt::entity entity) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    if (auto *bi = registry.try_get<detail::body_index>(entity); bi && s.ctx && bi->value < s.uploaded_bodies && !s.scene_dirty)
        detail::check(s, edynhip_wake_bodies(s.ctx, 1, &bi->value));
}

/// util/island_util.hpp:20-26: in the sequential stepper wake_up_entity IS wake_up_island_resident (rigidbody.cpp:409-415)
inline void wake_up_island_resident(entt::registry &registry, entt::entity entity) { wake_up_entity(registry, entity); }
inline void wake_up_island_residents(entt::registry &registry, const std::vector<entt::entity> &entities) { for (const entt::entity e : entities) wake_up_entity(registry, e); }

/// edyn::insert_material_mixing (util/insert_material_mixing.hpp:17): the material of contacts between bodies whose materials carry
/// these ids. As in the reference the lookup is sensitive to the order in which the two bodies meet (edynhip.h).
inline void insert_material_mixing(entt::registry &registry, material::id_type id0, material::id_type id1, const material_base &m) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    detail::gpu_stepper::mixing e{id0, id1, {m.restitution, m.friction, m.spin_friction, m.roll_friction, m.stiffness, m.damping}};
    s.mixings.push_back(e);
    if (s.ctx) detail::check(s, edynhip_insert_material_mixing(s.ctx, id0, id1, e.v));
}

/// Refreshes pivots / normal / distance / impulses of every contact_point entity from the device (one read-back).
inline void refresh_contact_points(entt::registry &registry) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    if (s.ctx && s.cfg.materialize_contacts) detail::refresh_contact_points(registry, s);
}

/// Current contact manifolds (body pair + point count), materialised on demand.
inline std::vector<contact_manifold> get_contact_manifolds(entt::registry &registry) {
    auto &s = registry.ctx().get<detail::gpu_stepper>();
    std::vector<contact_manifold> out;
    if (!s.ctx) return out;
    uint32_t n = 0;
    detail::check(s, edynhip_num_manifolds(s.ctx, &n));
    std::vector<edynhip_manifold> recs(n);
    if (n) detail::check(s, edynhip_get_manifolds(s.ctx, recs.data(), n, &n));
    out.reserve(n);
    for (auto &r : recs) if (s.bodies[r.body[0]] != entt::null && s.bodies[r.body[1]] != entt::null) out.push_back({{s.bodies[r.body[0]], s.bodies[r.body[1]]}, r.num_points});
    return out;
}

}  // namespace edyn
