// What the DROP-IN costs: edyn::update(registry, t) over the registry on the headline pile (BASELINE.json's 32k-box pile, 10 SI
// iterations), built with edyn::make_rigidbody, against the raw edynhip_step rate on the very same context.
// The reference's loop has no write-back at all - the registry IS its storage (stepper_sequential.cpp:28-119); this shim pays for
// the device -> registry copy, update_presentation (update_presentation.cpp:56-84), the sleeping tags and the contact entities.
//
//   bench_update [n=32] [settle=120] [steps=300] [modes=all]     prints one JSON line per run and a final "BENCH_UPDATE {...}" line
//
// Each run: attach -> 1 + n^3 x make_rigidbody -> `settle` updates -> `steps` timed updates (one fixed step per update) -> the
// host-side breakdown of the timed updates (edyn::get_shim_timings) -> the raw rate: edynhip_step(ctx, steps) on the same context.
#include <edyn/edyn.hpp>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// SplitMix64 in counter mode, the jitter source of every benchmark scene (edyn_amd/scenes.py splitmix64_uniform)
static float splitmix_uniform(uint64_t i) {
    uint64_t z = 0x9E3779B97F4A7C15ull + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)((double)(z >> 40) / (double)(1 << 24));
}

// the brick-offset lattice of unit boxes on a static plane (edyn_amd/scenes.py box_pile): layer k shifted by 0.5 (k & 1) in x and z,
// horizontal pitch 1.02, vertical pitch 1.005, lowest centre 0.505, +-0.005 m and +-0.02 rad of yaw jitter
static void build_pile(entt::registry &registry, int n, std::vector<entt::entity> &bodies) {
    auto floor_def = edyn::rigidbody_def{};
    floor_def.kind = edyn::rigidbody_kind::rb_static;
    floor_def.shape = edyn::plane_shape{{0, 1, 0}, 0};
    bodies.push_back(edyn::make_rigidbody(registry, floor_def));
    uint64_t b = 0;
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j, ++b) {
                const double off = 0.5 * (k & 1);
                float x = (float)((i - (n - 1) / 2.0) * 1.02 + off), y = (float)(0.505 + k * 1.005), z = (float)((j - (n - 1) / 2.0) * 1.02 + off);
                x += (splitmix_uniform(3 * b) * 2 - 1) * 0.005f;
                z += (splitmix_uniform(3 * b + 1) * 2 - 1) * 0.005f;
                const float half = (splitmix_uniform(3 * b + 2) * 2 - 1) * 0.02f * 0.5f;
                auto def = edyn::rigidbody_def{};
                def.mass = 1;
                def.shape = edyn::box_shape{{0.5f, 0.5f, 0.5f}};
                def.position = {x, y, z};
                def.orientation = {0, std::sin(half), 0, std::cos(half)};
                def.sleeping_disabled = true;
                bodies.push_back(edyn::make_rigidbody(registry, def));
            }
}

struct run_result {
    std::string name;
    double update_steps_per_sec{0}, raw_steps_per_sec{0}, host_ms_per_update{0};
    edyn::shim_timings tm;
    size_t bodies{0}, contact_points{0}, contact_entities{0};
    bool finite{true};
    float top_y{0};
};

static run_result run(const char *name, int n, int settle, int steps, edyn::execution_mode mode, bool exclusive, bool contacts) {
    run_result r;
    r.name = name;
    entt::registry registry;
    auto cfg = edyn::init_config{};
    cfg.execution_mode = mode;
    cfg.num_solver_velocity_iterations = 10;
    cfg.num_solver_position_iterations = 3;
    cfg.exclusive_device = exclusive;
    cfg.materialize_contacts = contacts;
    edyn::attach(registry, cfg);
    std::vector<entt::entity> bodies;
    build_pile(registry, n, bodies);
    r.bodies = bodies.size();
    const double dt = (double)cfg.fixed_dt;
    double t = 0.5 * dt;   // half a step of slack on either side of the accumulator's floor(): exactly one step per update
    edyn::update(registry, t);   // (zero steps: stamps the stepper)
    for (int k = 0; k < settle; ++k) { t += dt; edyn::update(registry, t); }
    edyn::reset_shim_timings(registry);
    auto &st = registry.ctx().get<edyn::detail::gpu_stepper>();
    edynhip_synchronize(st.ctx);
    const double t0 = now_s();
    for (int k = 0; k < steps; ++k) { t += dt; edyn::update(registry, t); }
    if (mode == edyn::execution_mode::asynchronous) edynhip_synchronize(st.ctx);   // the last update's step is still in flight
    const double t1 = now_s();
    r.tm = edyn::get_shim_timings(registry);
    r.update_steps_per_sec = (double)r.tm.steps / (t1 - t0);
    // host time per update that is NOT the device step: everything but the step call and the wait for the state
    r.host_ms_per_update = (r.tm.sync_removed + r.tm.upload + r.tm.write_back + r.tm.contacts + r.tm.presentation) / (double)std::max<uint64_t>(r.tm.updates, 1);
    float top = -1e30f;
    for (size_t i = 1; i < bodies.size(); ++i) {
        const auto &p = registry.get<edyn::position>(bodies[i]);
        const auto &pp = registry.get<edyn::present_position>(bodies[i]);
        r.finite = r.finite && std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z) && std::isfinite(pp.y);
        top = std::max(top, p.y);
    }
    r.top_y = top;
    edynhip_stats stats{};
    edynhip_get_stats(st.ctx, &stats);
    r.contact_points = stats.num_points;
    size_t ce = 0;
    registry.view<edyn::contact_point>().each([&](auto, auto &) { ++ce; });
    r.contact_entities = ce;
    // the raw rate on the same context, same regime: edynhip_step with the state resident (what bench.py's `value` is)
    edynhip_synchronize(st.ctx);
    const double r0 = now_s();
    edynhip_step(st.ctx, (uint32_t)steps);
    edynhip_synchronize(st.ctx);
    const double r1 = now_s();
    r.raw_steps_per_sec = steps / (r1 - r0);
    edyn::detach(registry);
    return r;
}

static void print(const run_result &r) {
    const double u = (double)std::max<uint64_t>(r.tm.updates, 1);
    std::printf("{\"run\": \"%s\", \"bodies\": %zu, \"contact_points\": %zu, \"contact_point_entities\": %zu, \"updates\": %llu, \"steps\": %llu, "
                "\"contact_events_per_update\": %.1f, \"update_steps_per_sec\": %.1f, \"raw_steps_per_sec\": %.1f, \"ratio\": %.3f, \"host_ms_per_update\": %.4f, "
                "\"ms_per_update\": {\"total\": %.4f, \"sync_removed\": %.4f, \"upload\": %.4f, \"step_call\": %.4f, \"state_wait\": %.4f, \"write_back\": %.4f, "
                "\"contacts\": %.4f, \"presentation\": %.4f}, \"finite\": %s, \"top_y\": %.3f}\n",
                r.name.c_str(), r.bodies, r.contact_points, r.contact_entities, (unsigned long long)r.tm.updates, (unsigned long long)r.tm.steps,
                (double)r.tm.contact_events / u, r.update_steps_per_sec, r.raw_steps_per_sec, r.update_steps_per_sec / r.raw_steps_per_sec, r.host_ms_per_update,
                r.tm.total / u, r.tm.sync_removed / u, r.tm.upload / u, r.tm.step_call / u, r.tm.state_wait / u, r.tm.write_back / u, r.tm.contacts / u,
                r.tm.presentation / u, r.finite ? "true" : "false", r.top_y);
    std::fflush(stdout);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 32;
    const int settle = argc > 2 ? std::atoi(argv[2]) : 120;
    const int steps = argc > 3 ? std::atoi(argv[3]) : 300;
    const std::string modes = argc > 4 ? argv[4] : "all";
    struct spec { const char *name; edyn::execution_mode mode; bool exclusive, contacts; };
    const spec specs[] = {
        {"sequential", edyn::execution_mode::sequential, false, true},
        {"sequential_exclusive", edyn::execution_mode::sequential, true, true},
        {"asynchronous", edyn::execution_mode::asynchronous, false, true},
        {"asynchronous_exclusive", edyn::execution_mode::asynchronous, true, true},
        {"sequential_exclusive_no_contact_entities", edyn::execution_mode::sequential, true, false},
    };
    bool ok = true;
    std::vector<run_result> results;
    try {
        for (const auto &sp : specs) {
            if (modes != "all" && modes != sp.name) continue;
            results.push_back(run(sp.name, n, settle, steps, sp.mode, sp.exclusive, sp.contacts));
            print(results.back());
            const auto &r = results.back();
            ok = ok && r.finite && r.tm.steps == (uint64_t)steps && r.top_y > 0.4f * n && r.top_y < 1.1f * n + 1;
            if (sp.contacts) ok = ok && r.contact_entities > 0;
        }
    } catch (const std::exception &e) {
        std::printf("bench_update: %s\n", e.what());
        return 2;
    }
    std::printf(ok ? "BENCH_UPDATE_OK\n" : "BENCH_UPDATE_FAIL\n");
    return ok ? 0 : 1;
}
