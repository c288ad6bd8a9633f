// edynhip_world: ONE simulation over several GPUs of a node, behind the C-ABI (include/edynhip.h "Multi-GPU world").
//
// The reference's unit of parallelism is the island: solver.cpp:408-428 hands every island to a worker, islands share only
// non-procedural bodies, which a step never writes (comp/island.hpp:34-41). Here a shard = one GPU = one edynhip_ctx that owns the
// dynamic bodies of its islands plus a replica of every non-dynamic body and steps them with NO data-path exchange; after every
// step the shard's packed state goes to pinned host memory (N independent device-to-host copies - the registry write-back of
// SURVEY 8e; processes that own one GPU each gather with RCCL instead, edyn_amd/parallel.py) and lands in the world's global arrays.
//
// What keeps the partition valid: islands of different shards must not come within the manifold-creation margin of each other
// unnoticed. Each shard reduces, ON THE DEVICE, (a) per island the union of its bodies' AABBs (k_island_box_*), when a check is
// due, and (b) every step, how far any body's AABB has grown out of the AABB it had at the last check (k_shard_growth: one float
// per shard and step). The host sweeps the island boxes of all shards (sort along x) for cross-shard pairs and remembers the
// smallest gap; until twice the largest growth has eaten that gap no check is needed - a resting world checks every few hundred
// steps, a world in motion as often as its motion requires, and no contact between two shards can appear unseen. When a gap closes
// the islands are re-partitioned (welded where their boxes overlap, longest-processing-time-first over the summed weights) and
// every shard is rebuilt with what its islands carry: contact manifolds (warm-start impulses, colours), the joints' applied impulses
// and tracked angles, sleeping tags, collision exclusions, joint definitions, meshes.
//
// Body order inside a shard is ascending global index, so canonical pair keys, island labels (lowest index) and the colouring keep
// their relative order: a shard computes for its islands exactly what one context computes for the whole world
// (tests/cpp/multi.cpp: bit-equal through a forced and an approach-triggered re-partition).
#include "ctx.hpp"
#include <chrono>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>

namespace eh {

// ------------------------------------------------------------------------------------------------ host-only pieces
// Longest-processing-time-first: islands by descending weight (ties: ascending label), each to the lightest rank so far (ties: the
// lowest rank). rank_of[i] = -1 for bodies that are not dynamic (replicated on every rank).
static void partition_islands(uint32_t n, const uint32_t *labels, const int32_t *kind, const double *weights, uint32_t world_size, int32_t *rank_of) {
    std::vector<uint32_t> isl;
    isl.reserve(1024);
    for (uint32_t i = 0; i < n; ++i) { rank_of[i] = -1; if (kind[i] == EDYNHIP_KIND_DYNAMIC) isl.push_back(labels[i]); }
    std::sort(isl.begin(), isl.end());
    isl.erase(std::unique(isl.begin(), isl.end()), isl.end());
    const uint32_t m = (uint32_t)isl.size();
    if (m == 0) return;
    auto index_of = [&](uint32_t label) { return (uint32_t)(std::lower_bound(isl.begin(), isl.end(), label) - isl.begin()); };
    std::vector<double> w(m, 0.0);
    for (uint32_t i = 0; i < n; ++i) if (kind[i] == EDYNHIP_KIND_DYNAMIC) w[index_of(labels[i])] += weights ? weights[i] : 1.0;
    std::vector<uint32_t> order(m);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return w[a] != w[b] ? w[a] > w[b] : isl[a] < isl[b]; });
    std::vector<double> load(world_size, 0.0);
    std::vector<int32_t> owner(m, 0);
    for (uint32_t k : order) {
        uint32_t r = 0;
        for (uint32_t q = 1; q < world_size; ++q) if (load[q] < load[r]) r = q;
        owner[k] = (int32_t)r;
        load[r] += w[k];
    }
    for (uint32_t i = 0; i < n; ++i) if (kind[i] == EDYNHIP_KIND_DYNAMIC) rank_of[i] = owner[index_of(labels[i])];
}

// The world's own placement rule (round 5): islands in the order of a space-filling curve through their box centres (30-bit Morton
// keys over the scene's bounds), cut into world_size runs of equal weight. Neighbours in space mostly share a shard, so islands that
// meet later mostly meet inside a shard - the longest-processing-time rule above scatters neighbours over the shards, and every meeting
// of two of them then costs a re-partition (a field of collapsing piles: 18 in 40 steps). A shard's load is within one island of the mean.
// aabb6[n][6]: the bodies' boxes; shapeless bodies count with a point at the origin of their island's other bodies (or 0).
static void partition_islands_spatial(uint32_t n, const uint32_t *labels, const int32_t *kind, const double *weights, const float *aabb6, uint32_t world_size, int32_t *rank_of) {
    std::vector<uint32_t> isl;
    for (uint32_t i = 0; i < n; ++i) { rank_of[i] = -1; if (kind[i] == EDYNHIP_KIND_DYNAMIC) isl.push_back(labels[i]); }
    std::sort(isl.begin(), isl.end());
    isl.erase(std::unique(isl.begin(), isl.end()), isl.end());
    const uint32_t m = (uint32_t)isl.size();
    if (m == 0) return;
    auto index_of = [&](uint32_t label) { return (uint32_t)(std::lower_bound(isl.begin(), isl.end(), label) - isl.begin()); };
    std::vector<double> w(m, 0.0), cx(3 * (size_t)m, 0.0), cw(m, 0.0);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < n; ++i) {
        if (kind[i] != EDYNHIP_KIND_DYNAMIC) continue;
        const uint32_t k = index_of(labels[i]);
        w[k] += weights ? weights[i] : 1.0;
        const float *b = aabb6 + 6 * (size_t)i;
        bool finite = true;
        for (int d = 0; d < 6; ++d) finite = finite && std::isfinite(b[d]) && std::fabs(b[d]) < 1.0e30f;
        if (!finite || (b[0] == 0 && b[3] == 0 && b[1] == 0 && b[4] == 0 && b[2] == 0 && b[5] == 0)) continue;   // no shape: no place of its own
        for (int d = 0; d < 3; ++d) { cx[3 * (size_t)k + d] += 0.5 * ((double)b[d] + (double)b[3 + d]); }
        cw[k] += 1.0;
    }
    // islands without any shaped body have no place: they stay out of the bounds the Morton keys are quantised over (a scene far from the
    // origin would otherwise lose its resolution to a phantom island at 0,0,0) and take the key of the scene's lower corner
    for (uint32_t k = 0; k < m; ++k) for (int d = 0; d < 3; ++d) {
        if (cw[k] <= 0) continue;
        cx[3 * (size_t)k + d] /= cw[k];
        lo[d] = std::min(lo[d], cx[3 * (size_t)k + d]); hi[d] = std::max(hi[d], cx[3 * (size_t)k + d]);
    }
    for (uint32_t k = 0; k < m; ++k) if (cw[k] <= 0) for (int d = 0; d < 3; ++d) cx[3 * (size_t)k + d] = lo[d] <= hi[d] ? lo[d] : 0.0;
    for (int d = 0; d < 3; ++d) if (lo[d] > hi[d]) lo[d] = hi[d] = 0.0;
    auto spread = [](uint32_t v) { v &= 0x3FFu; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u; return v; };
    std::vector<uint32_t> key(m), order(m);
    for (uint32_t k = 0; k < m; ++k) {
        uint32_t q[3];
        for (int d = 0; d < 3; ++d) { const double ext = hi[d] - lo[d]; q[d] = ext > 0 ? (uint32_t)std::min(1023.0, (cx[3 * (size_t)k + d] - lo[d]) / ext * 1024.0) : 0u; }
        key[k] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
    }
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : isl[a] < isl[b]; });
    double total = 0, before = 0;
    for (double x : w) total += x;
    std::vector<int32_t> owner(m, 0);
    for (uint32_t k : order) {   // the run an island falls into is decided by the middle of its own weight interval
        const double mid = before + 0.5 * w[k];
        owner[k] = (int32_t)std::min<double>(world_size - 1, std::floor(mid / total * world_size));
        before += w[k];
    }
    for (uint32_t i = 0; i < n; ++i) if (kind[i] == EDYNHIP_KIND_DYNAMIC) rank_of[i] = owner[index_of(labels[i])];
}

struct IslandBox { float lo[3], hi[3]; uint32_t label; int32_t owner; };
// Sweep along x over boxes grown by `reach` on every side: calls hit(a, b, gap) for every pair whose grown boxes overlap, gap = the
// largest per-axis distance between the ORIGINAL boxes (0 when they overlap). cross_only: pairs of different owners only.
template <typename Hit>
static void sweep_boxes(std::vector<IslandBox> &boxes, float reach, bool cross_only, Hit &&hit) {
    // Sweep along x; the boxes still open at the sweep front are kept in BUCKETS along z, so a box only meets the open boxes of the z
    // cells it reaches. (Round 6: with one list of open boxes a field of piles on a 64 x 64 grid - 262 144 body boxes, every pile of a
    // column open at once - cost ~260 M pair tests, 0.8 s of every re-partition's 1.2 s.) A pair that shares several cells is tested in
    // the first of them only.
    std::sort(boxes.begin(), boxes.end(), [](const IslandBox &a, const IslandBox &b) { return a.lo[0] != b.lo[0] ? a.lo[0] < b.lo[0] : a.label < b.label; });
    const uint32_t nb = (uint32_t)boxes.size();
    if (nb < 2) return;
    float zlo = 3.0e38f, zhi = -3.0e38f;
    double ext = 0;
    uint32_t finite = 0;
    for (const IslandBox &b : boxes) {
        if (!(b.lo[2] <= b.hi[2]) || !std::isfinite(b.lo[2]) || !std::isfinite(b.hi[2])) continue;
        zlo = std::min(zlo, b.lo[2]); zhi = std::max(zhi, b.hi[2]); ext += (double)b.hi[2] - (double)b.lo[2]; ++finite;
    }
    const float mean_ext = finite ? (float)(ext / finite) : 0.0f;
    const float span = finite ? zhi - zlo : 0.0f;
    const uint32_t cells = span > 0 ? (uint32_t)std::min(4096.0f, std::max(1.0f, span / std::max(mean_ext + 2 * reach, 1e-3f))) : 1u;
    const float inv = cells > 1 ? (float)cells / span : 0.0f;
    auto cell_of = [&](float z) { if (!(z > zlo)) return 0u; const float c = (z - zlo) * inv; return c >= (float)(cells - 1) ? cells - 1 : (uint32_t)c; };   // (NaN and -inf land in cell 0, +inf in the last)
    std::vector<std::vector<uint32_t>> open(cells);
    std::vector<uint32_t> first_cell(nb);
    for (uint32_t k = 0; k < nb; ++k) {
        const IslandBox &bk = boxes[k];
        const bool sane = bk.lo[2] <= bk.hi[2];
        const uint32_t q0 = sane ? cell_of(bk.lo[2] - 2 * reach) : 0u, q1 = sane ? cell_of(bk.hi[2] + 2 * reach) : cells - 1;
        for (uint32_t c = q0; c <= q1; ++c) {
            std::vector<uint32_t> &bucket = open[c];
            size_t keep = 0;
            for (uint32_t a : bucket) {
                const IslandBox &ba = boxes[a];
                if (!(ba.hi[0] + 2 * reach >= bk.lo[0])) continue;   // the sweep front has passed it: drop it from this bucket
                bucket[keep++] = a;
                if (c != std::max(first_cell[a], q0)) continue;       // this pair is tested in the first cell the two share
                if (cross_only && ba.owner == bk.owner) continue;
                float gap = 0;
                for (int d = 0; d < 3; ++d) gap = std::max(gap, std::max(bk.lo[d] - ba.hi[d], ba.lo[d] - bk.hi[d]));
                if (gap < 2 * reach) hit(ba, bk, gap);
            }
            bucket.resize(keep);
        }
        const uint32_t i0 = sane ? cell_of(bk.lo[2]) : 0u, i1 = sane ? cell_of(bk.hi[2]) : cells - 1;
        first_cell[k] = i0;
        for (uint32_t c = i0; c <= i1; ++c) open[c].push_back(k);
    }
}

// ------------------------------------------------------------------------------------------------ device pieces
__device__ __forceinline__ uint32_t ordered_bits(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float from_ordered_bits(uint32_t u) {
    const uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f; memcpy(&f, &v, 4); return f;
}
__device__ __forceinline__ bool shard_body_counts(uint32_t fl) { return (fl & BF_KIND_MASK) == EDYNHIP_KIND_DYNAMIC && !(fl & BF_REMOVED) && (fl & BF_SHAPE_MASK) != 0; }

// (b) above: the largest distance by which a dynamic body's AABB sticks out of the AABB recorded at the last check (0 = nothing moved out)
__global__ void k_shard_growth(uint32_t n, Bodies b, const float4 *__restrict__ amin0, const float4 *__restrict__ amax0, uint32_t *out_bits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float g = 0.0f;
    if (i < n && shard_body_counts(b.flags[i])) {
        const float4 lo = b.amin[i], hi = b.amax[i], lo0 = amin0[i], hi0 = amax0[i];
        g = fmaxf(fmaxf(fmaxf(lo0.x - lo.x, lo0.y - lo.y), lo0.z - lo.z), fmaxf(fmaxf(hi.x - hi0.x, hi.y - hi0.y), hi.z - hi0.z));
        g = fmaxf(g, 0.0f);
        if (!(g == g)) g = 3.0e38f;   // a NaN box: force a check (and let the check fail loudly)
    }
    for (int off = 32; off > 0; off >>= 1) g = fmaxf(g, __shfl_xor(g, off));
    if ((threadIdx.x & 63) == 0 && g > 0.0f) atomicMax(out_bits, __float_as_uint(g));   // non-negative floats order like their bits
}
__global__ void k_island_box_clear(uint32_t n, uint32_t *lo, uint32_t *hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) { lo[i] = 0xFFFFFFFFu; hi[i] = 0u; }
}
// (a) above, also records the boxes the growth is measured against
__global__ void k_island_box_reduce(uint32_t n, Bodies b, uint32_t *lo, uint32_t *hi, float4 *amin0, float4 *amax0, bool per_body) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 mn = b.amin[i], mx = b.amax[i];
    amin0[i] = mn; amax0[i] = mx;
    if (!shard_body_counts(b.flags[i])) return;
    uint32_t l = per_body ? i : b.island[i];
    if (l >= n) l = i;
    atomicMin(&lo[3 * l], ordered_bits(mn.x)); atomicMin(&lo[3 * l + 1], ordered_bits(mn.y)); atomicMin(&lo[3 * l + 2], ordered_bits(mn.z));
    atomicMax(&hi[3 * l], ordered_bits(mx.x)); atomicMax(&hi[3 * l + 1], ordered_bits(mx.y)); atomicMax(&hi[3 * l + 2], ordered_bits(mx.z));
}
__global__ void k_island_box_compact(uint32_t n, const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi, uint32_t *count, uint32_t *out_label, float *out_box) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n || lo[3 * l] == 0xFFFFFFFFu) return;   // no shaped dynamic body carries this label
    const uint32_t k = atomicAdd(count, 1u);
    out_label[k] = l;
    for (int d = 0; d < 3; ++d) { out_box[6 * k + d] = from_ordered_bits(lo[3 * l + d]); out_box[6 * k + 3 + d] = from_ordered_bits(hi[3 * l + d]); }
}

}  // namespace eh

using namespace eh;

// ------------------------------------------------------------------------------------------------ the world
namespace {

struct HostScene {   // deep copy of what the caller described, in global indices
    uint32_t n = 0;
    std::vector<int32_t> kind, shape_type;
    std::vector<float> pos, orn, linvel, angvel, mass, inertia, shape_param, friction, restitution, gravity, com;
    std::vector<uint8_t> has_inertia, sleeping_disabled;
    std::vector<uint64_t> group, mask;
    uint32_t nj = 0;
    std::vector<int32_t> jtype;
    std::vector<uint32_t> jbody;
    std::vector<float> jpivot, jaxis, jparams;
    struct Def { uint32_t joint; bool generic; std::vector<float> fa, fb, p; };
    std::vector<Def> defs;
    std::vector<std::array<uint32_t, 2>> exclusions;
    struct Mesh { std::vector<float> v; std::vector<uint32_t> idx, faces; uint32_t flags; };
    std::vector<Mesh> meshes;
};

struct Shard {
    edynhip_ctx *ctx = nullptr;
    int device = 0;
    std::vector<uint32_t> local_ids;      // local index -> global index (ascending)
    std::vector<int32_t> to_local;        // global index -> local index or -1
    std::vector<uint32_t> owned_local;    // local indices of the bodies whose state this shard contributes
    std::vector<uint32_t> local_joints;   // local joint index -> global joint index
    float *pack_dev = nullptr, *pack_host = nullptr;   // [n_local][13]
    // monitor scratch on the device: [0] growth bits, [1] island count | per label 3 lo + 3 hi ordered bits | out labels | out boxes | amin0 | amax0
    uint32_t *mon_dev = nullptr, *mon_host = nullptr;
    uint32_t cap = 0, joint_cap = 0, meshes_created = 0;
    int rc = EDYNHIP_OK;
    std::string err;
    // collected for a re-partition
    std::vector<uint32_t> labels; std::vector<float> aabb; std::vector<edynhip_manifold> manifolds; std::vector<float> imp24, imp10; std::vector<uint8_t> asleep;
    std::vector<uint32_t> sleep_label; std::vector<double> sleep_since; double sleep_clock = 0;
};

class Pool {   // one persistent host thread per shard (a context is single-threaded; its step spins on its own counters)
public:
    explicit Pool(uint32_t n) : n_(n), serial_(getenv("EDYNHIP_WORLD_SERIAL") != nullptr) {   // (developer knob: every shard on the caller's thread)
        if (!serial_) for (uint32_t r = 0; r < n; ++r) threads_.emplace_back([this, r] { loop(r); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run(const std::function<void(uint32_t)> &fn) {
        if (n_ == 1) { fn(0); return; }
        if (serial_) { for (uint32_t r = 0; r < n_; ++r) fn(r); return; }
        { std::lock_guard<std::mutex> g(m_); fn_ = &fn; pending_ = n_; ++gen_; }
        cv_.notify_all();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
private:
    void loop(uint32_t r) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(uint32_t)> *fn;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            (*fn)(r);
            { std::lock_guard<std::mutex> g(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    uint32_t n_;
    bool serial_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(uint32_t)> *fn_ = nullptr;
    uint32_t pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

constexpr float kCreationMargin = 0.026f;   // broadphase.hpp:18: a manifold exists once two AABBs are this close; both boxes are grown by it
constexpr float kHorizon = 1.0f;            // cross-shard pairs further apart than this are not tracked individually

}  // namespace

struct edynhip_world {
    edynhip_config cfg{};
    std::vector<int> devices;
    std::vector<Shard> shards;
    std::unique_ptr<Pool> pool;
    HostScene scene;
    std::vector<int32_t> rank_of;
    std::vector<float> pos, orn, linvel, angvel;   // the gathered state, global order
    // settings.should_collide_func on a world over several devices (broadphase.cpp:143-153): the user's predicate speaks GLOBAL body
    // indices; every shard's context asks it through a thunk that maps its local indices back (ShardFilter, one per shard, stable
    // across re-partitions). The shards step on their own host threads: the calls are serialised (the reference's sequential stepper
    // asks from one thread).
    edynhip_pair_filter filter = nullptr;
    void *filter_user = nullptr;
    std::mutex filter_mutex;
    struct ShardFilter { edynhip_world *w; uint32_t shard; };
    std::vector<ShardFilter> shard_filters;
    bool built = false;
    bool stepped = false;                          // stepped since the scene was last described (set_bodies): scene-description calls are refused then
    float budget = 0.0f;                           // how much two islands of different shards may still approach before a check is due
    edynhip_world_stats stats{};
    std::string err;
    int fail(int code, const std::string &what) { err = what; return code; }
};

namespace {

int shard_error(edynhip_world *w) {
    for (uint32_t r = 0; r < w->shards.size(); ++r)
        if (w->shards[r].rc != EDYNHIP_OK) return w->fail(w->shards[r].rc, "shard " + std::to_string(r) + ": " + w->shards[r].err);
    return EDYNHIP_OK;
}
#define SH_TRY(s, expr) do { int r__ = (expr); if (r__ != EDYNHIP_OK) { (s).rc = r__; (s).err = (s).ctx ? edynhip_last_error((s).ctx) : edynhip_last_error(nullptr); return; } } while (0)
#define SH_HIP(s, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (s).rc = EDYNHIP_ERR_HIP; (s).err = std::string(#call) + ": " + hipGetErrorString(e__); return; } } while (0)

void free_shard(Shard &s) {
    if (s.ctx) { (void)hipSetDevice(s.device); (void)edynhip_synchronize(s.ctx); }
    if (s.pack_dev) (void)hipFree(s.pack_dev);
    if (s.pack_host) (void)hipHostFree(s.pack_host);
    if (s.mon_dev) (void)hipFree(s.mon_dev);
    if (s.mon_host) (void)hipHostFree(s.mon_host);
    if (s.ctx) edynhip_destroy(s.ctx);
    s.ctx = nullptr; s.pack_dev = s.pack_host = nullptr; s.mon_dev = s.mon_host = nullptr;
    s.cap = s.joint_cap = s.meshes_created = 0;
}

template <typename T>
std::vector<T> take(const std::vector<T> &src, const std::vector<uint32_t> &ids, size_t width) {
    std::vector<T> out;
    if (src.empty()) return out;
    out.resize(ids.size() * width);
    for (size_t k = 0; k < ids.size(); ++k) std::memcpy(&out[k * width], &src[(size_t)ids[k] * width], width * sizeof(T));
    return out;
}

// layout of the monitor scratch (Shard::mon_dev, edynhip_get_island_boxes), in 32-bit words, for `cap` bodies:
//   [0] growth bits  [1] island count | per label 3 lo + 3 hi ordered bits | out labels | out boxes (6 floats) | pad | amin0 | amax0 (float4 each)
struct MonLayout {
    size_t lo, hi, out_label, out_box, amin0, amax0, words;
    explicit MonLayout(size_t cap) {
        lo = 2; hi = lo + 3 * cap; out_label = hi + 3 * cap; out_box = out_label + cap;
        amin0 = (out_box + 6 * cap + 3) & ~(size_t)3; amax0 = amin0 + 4 * cap; words = amax0 + 4 * cap;
    }
};

struct Carry {   // what travels with the islands through a re-partition (global indices)
    std::vector<edynhip_manifold> manifolds;   // canonical order
    std::vector<float> imp24, angle;           // per global joint
    std::vector<uint8_t> asleep;               // per global body
    std::vector<uint32_t> label;               // per global body: its island (global index of the island's lowest body) - with
    std::vector<double> since;                 // ... the islands' sleep timers by that label, and the clock they are measured on (ADVICE r04:
    double clock = 0;                          //     every re-partition used to restart every island's timer)
    bool any = false;
    // sticky re-partitions: the manifolds already sorted out per NEW shard (global indices, canonical order) - the bodies of a shard that
    // stay keep their relative order, so their manifolds are taken over in place and only the few that arrive are merged in; `manifolds`
    // above (one merged, sorted list of everything) is then empty. Consumed (moved from) by build_shard.
    mutable std::vector<std::vector<edynhip_manifold>> per_shard;
    // island boxes of the state the shards are built from, owners = the new partition: the approach check after the rebuild sweeps these
    // (thousands) instead of every body's box (hundreds of thousands)
    std::vector<IslandBox> island_boxes;
};

int shard_filter_thunk(void *user, uint32_t body, uint32_t other) {
    auto *sf = static_cast<edynhip_world::ShardFilter *>(user);
    edynhip_world *w = sf->w;
    const Shard &s = w->shards[sf->shard];
    std::lock_guard<std::mutex> lock(w->filter_mutex);
    return w->filter(w->filter_user, s.local_ids[body], s.local_ids[other]);
}

// developer aid (EDYNHIP_WORLD_TRACE=1): wall time of the phases of a re-partition, to stderr
struct PhaseTrace {
    bool on; std::chrono::steady_clock::time_point t0; const char *what;
    explicit PhaseTrace(const char *w_) : on(getenv("EDYNHIP_WORLD_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), what(w_) {}
    void mark(const char *phase) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[edynhip_world] %s: %s %.2f ms\n", what, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
// (Re)builds shard r for the partition w->rank_of from the scene, the current global state and what the islands carry.
void build_shard(edynhip_world *w, uint32_t r, const Carry &carry, bool from_state) {
    Shard &s = w->shards[r];
    const HostScene &sc = w->scene;
    s.rc = EDYNHIP_OK; s.err.clear();
    PhaseTrace trace("build_shard");
    SH_HIP(s, hipSetDevice(s.device));
    const uint32_t n = sc.n;
    s.local_ids.clear(); s.owned_local.clear();
    s.to_local.assign(n, -1);
    for (uint32_t i = 0; i < n; ++i)
        if (w->rank_of[i] == (int32_t)r || w->rank_of[i] < 0) {
            s.to_local[i] = (int32_t)s.local_ids.size();
            if (w->rank_of[i] == (int32_t)r || r == 0) s.owned_local.push_back((uint32_t)s.local_ids.size());   // shard 0 also reports the replicated bodies
            s.local_ids.push_back(i);
        }
    const uint32_t nl = (uint32_t)s.local_ids.size();
    // joints whose bodies are here and of which this shard owns one (a joint to a replicated anchor lives on one shard)
    s.local_joints.clear();
    for (uint32_t g = 0; g < sc.nj; ++g) {
        const uint32_t a = sc.jbody[2 * g], b = sc.jbody[2 * g + 1];
        // a joint is an island edge: a partition that puts its two dynamic bodies on different shards is wrong - never drop it silently
        if (w->rank_of[a] >= 0 && w->rank_of[b] >= 0 && w->rank_of[a] != w->rank_of[b]) {
            s.rc = EDYNHIP_ERR_INTERNAL; s.err = "partition splits joint " + std::to_string(g) + " (bodies " + std::to_string(a) + ", " + std::to_string(b) + ") over two shards";
            return;
        }
        if (s.to_local[a] >= 0 && s.to_local[b] >= 0 && (w->rank_of[a] == (int32_t)r || w->rank_of[b] == (int32_t)r)) s.local_joints.push_back(g);
    }
    edynhip_config cfg = w->cfg;
    cfg.device = s.device;
    if (w->cfg.max_manifolds) cfg.max_manifolds = w->cfg.max_manifolds; else cfg.max_manifolds = 0;
    // A shard that is rebuilt keeps its context when the new body / joint set fits the old capacities: edynhip_set_bodies starts a new
    // world on an existing context (manifolds, joints, exclusions, sleep state, clocks - everything but the meshes, which are the scene's
    // and stay), and a context's ~100 device allocations - over a gigabyte for a 32k-body shard - are neither freed nor made again
    // (round 6: the frees and allocations were ~0.1 s of every re-partition). A context that is too small is replaced with head-room.
    const bool reuse = s.ctx != nullptr && nl <= s.cap && (uint32_t)s.local_joints.size() <= s.joint_cap && (uint32_t)sc.meshes.size() == s.meshes_created;
    if (!reuse) {
        free_shard(s);
        trace.mark("free the old context");
        cfg.max_bodies = std::max<uint32_t>(nl + nl / 4 + 64, 64);
        cfg.max_joints = std::max<uint32_t>((uint32_t)s.local_joints.size() + (uint32_t)s.local_joints.size() / 4 + 16, 16);
        int status = 0;
        s.ctx = edynhip_create(&cfg, &status);
        if (!s.ctx) { s.rc = status; s.err = edynhip_last_error(nullptr); return; }
        s.cap = cfg.max_bodies; s.joint_cap = cfg.max_joints;
        trace.mark("edynhip_create");
        for (const HostScene::Mesh &m : sc.meshes) {
            uint32_t id = 0;
            SH_TRY(s, edynhip_create_convex_mesh(s.ctx, (uint32_t)m.v.size() / 3, m.v.data(), (uint32_t)m.idx.size(), m.idx.data(), (uint32_t)m.faces.size() / 2, m.faces.data(), m.flags, &id));
        }
        s.meshes_created = (uint32_t)sc.meshes.size();
    } else {
        SH_TRY(s, edynhip_synchronize(s.ctx));
    }
    // bodies: the scene's definitions with the CURRENT state
    const std::vector<float> &P = from_state ? w->pos : sc.pos, &Q = from_state ? w->orn : sc.orn, &V = from_state ? w->linvel : sc.linvel, &W = from_state ? w->angvel : sc.angvel;
    auto kind = take(sc.kind, s.local_ids, 1), shape_type = take(sc.shape_type, s.local_ids, 1);
    auto pos = take(P, s.local_ids, 3), orn = take(Q, s.local_ids, 4), lv = take(V, s.local_ids, 3), av = take(W, s.local_ids, 3);
    auto mass = take(sc.mass, s.local_ids, 1), inertia = take(sc.inertia, s.local_ids, 9), sp = take(sc.shape_param, s.local_ids, 4);
    auto fr = take(sc.friction, s.local_ids, 1), re = take(sc.restitution, s.local_ids, 1), gr = take(sc.gravity, s.local_ids, 3), com = take(sc.com, s.local_ids, 3);
    auto hi = take(sc.has_inertia, s.local_ids, 1), sd = take(sc.sleeping_disabled, s.local_ids, 1);
    auto group = take(sc.group, s.local_ids, 1), mask = take(sc.mask, s.local_ids, 1);
    edynhip_bodies b{};
    b.kind = kind.data(); b.pos = pos.data(); b.orn = orn.data(); b.linvel = lv.data(); b.angvel = av.data(); b.mass = mass.data();
    b.inertia = inertia.empty() ? nullptr : inertia.data(); b.has_inertia = hi.empty() ? nullptr : hi.data();
    b.shape_type = shape_type.data(); b.shape_param = sp.data(); b.friction = fr.data(); b.restitution = re.data();
    b.group = group.empty() ? nullptr : group.data(); b.mask = mask.empty() ? nullptr : mask.data();
    b.gravity = gr.empty() ? nullptr : gr.data(); b.sleeping_disabled = sd.empty() ? nullptr : sd.data();
    b.center_of_mass = com.empty() ? nullptr : com.data();
    SH_TRY(s, edynhip_set_bodies(s.ctx, nl, &b));
    trace.mark("take + edynhip_set_bodies");
    if (from_state && !com.empty()) {   // set_bodies read `pos` as the origin of bodies with an offset: put the centre-of-mass state back ...
        SH_TRY(s, edynhip_set_state(s.ctx, pos.data(), orn.data(), lv.data(), av.data()));
        // ... and with it what set_bodies derived from the misread position: origins, AABBs (displaced by R com otherwise - the first
        // broadphase / narrowphase after a re-partition would place those shapes wrong) and world inertias. Before the sleep tags below:
        // the derivation skips sleeping bodies
        SH_TRY(s, edynhip_refresh_derived(s.ctx));
    }
    // joints, in local indices
    const uint32_t njl = (uint32_t)s.local_joints.size();
    if (njl) {
        auto jt = take(sc.jtype, s.local_joints, 1);
        auto jb = take(sc.jbody, s.local_joints, 2);
        for (uint32_t &x : jb) x = (uint32_t)s.to_local[x];
        auto jp = take(sc.jpivot, s.local_joints, 6), ja = take(sc.jaxis, s.local_joints, 6), jq = take(sc.jparams, s.local_joints, 10);
        edynhip_joints j{};
        j.type = jt.data(); j.body = jb.data(); j.pivot = jp.data(); j.axis = ja.empty() ? nullptr : ja.data(); j.params = jq.empty() ? nullptr : jq.data();
        SH_TRY(s, edynhip_set_joints(s.ctx, njl, &j));
        for (const HostScene::Def &d : sc.defs) {
            const auto it = std::lower_bound(s.local_joints.begin(), s.local_joints.end(), d.joint);
            if (it == s.local_joints.end() || *it != d.joint) continue;
            const uint32_t lj = (uint32_t)(it - s.local_joints.begin());
            if (d.generic) SH_TRY(s, edynhip_set_generic_definition(s.ctx, lj, d.fa.data(), d.fb.data(), d.p.data()));
            else SH_TRY(s, edynhip_set_joint_definition(s.ctx, lj, d.fa.data(), d.fb.data(), d.p.data()));
        }
    }
    for (const auto &e : sc.exclusions)
        if (s.to_local[e[0]] >= 0 && s.to_local[e[1]] >= 0 && (w->rank_of[e[0]] == (int32_t)r || w->rank_of[e[1]] == (int32_t)r))
            SH_TRY(s, edynhip_exclude_collision(s.ctx, (uint32_t)s.to_local[e[0]], (uint32_t)s.to_local[e[1]]));
    if (w->filter) SH_TRY(s, edynhip_set_pair_filter(s.ctx, &shard_filter_thunk, &w->shard_filters[r]));
    if (carry.any) {
        std::vector<edynhip_manifold> mine;
        if (!carry.per_shard.empty()) {
            mine.swap(carry.per_shard[r]);
            for (edynhip_manifold &m : mine) { m.body[0] = (uint32_t)s.to_local[m.body[0]]; m.body[1] = (uint32_t)s.to_local[m.body[1]]; }
        } else
        for (const edynhip_manifold &m : carry.manifolds)
            if (w->rank_of[m.body[0]] == (int32_t)r || w->rank_of[m.body[1]] == (int32_t)r) {
                mine.push_back(m);
                mine.back().body[0] = (uint32_t)s.to_local[m.body[0]]; mine.back().body[1] = (uint32_t)s.to_local[m.body[1]];   // a monotone map: the canonical order is kept
            }
        trace.mark("joints, exclusions, select carried manifolds");
        if (!mine.empty()) SH_TRY(s, edynhip_set_manifolds(s.ctx, mine.data(), (uint32_t)mine.size()));
        trace.mark("edynhip_set_manifolds");
        if (njl && !carry.imp24.empty()) {
            auto i24 = take(carry.imp24, s.local_joints, 24), ang = take(carry.angle, s.local_joints, 1);
            SH_TRY(s, edynhip_set_joint_warm_start(s.ctx, i24.data(), ang.data()));
        }
        if (!carry.asleep.empty() && std::any_of(carry.asleep.begin(), carry.asleep.end(), [](uint8_t a) { return a != 0; })) {
            auto as = take(carry.asleep, s.local_ids, 1);
            SH_TRY(s, edynhip_set_asleep(s.ctx, as.data()));
        }
        if (!carry.label.empty()) {   // the islands' sleep timers go on where they were (edynhip_set_sleep_timers), in local indices
            std::vector<uint32_t> lab(nl);
            std::vector<double> since(nl, -1.0);
            for (uint32_t l = 0; l < nl; ++l) {
                const uint32_t g = s.local_ids[l], root = carry.label[g];
                const int32_t lr = w->rank_of[g] == (int32_t)r && root < n ? s.to_local[root] : -1;   // an island lives on one shard: its root is local
                lab[l] = lr >= 0 ? (uint32_t)lr : l;
                if (lr >= 0 && root == g) since[l] = carry.since[g];
            }
            SH_TRY(s, edynhip_set_sleep_timers(s.ctx, lab.data(), since.data(), carry.clock));
        }
    }
    // gather and monitor buffers (sized by the context's capacity: they live as long as the context does)
    const size_t words = MonLayout(s.cap).words;
    if (!s.pack_dev) {
        SH_HIP(s, hipMalloc((void **)&s.pack_dev, (size_t)std::max<uint32_t>(s.cap, 1) * 13 * sizeof(float)));
        SH_HIP(s, hipHostMalloc((void **)&s.pack_host, (size_t)std::max<uint32_t>(s.cap, 1) * 13 * sizeof(float), hipHostMallocDefault));
        SH_HIP(s, hipMalloc((void **)&s.mon_dev, words * sizeof(uint32_t)));
        SH_HIP(s, hipHostMalloc((void **)&s.mon_host, (2 + 7 * (size_t)s.cap) * sizeof(uint32_t), hipHostMallocDefault));
    }
    SH_HIP(s, hipMemset(s.mon_dev, 0, words * sizeof(uint32_t)));
    trace.mark("warm start, sleep state, gather / monitor buffers");
}

// the shard's state into the world's global arrays (its own bodies; shard 0: also the replicated ones)
void scatter_state(edynhip_world *w, Shard &s) {
    for (uint32_t l : s.owned_local) {
        const float *row = s.pack_host + 13 * (size_t)l;
        const uint32_t g = s.local_ids[l];
        std::memcpy(&w->pos[3 * (size_t)g], row, 12); std::memcpy(&w->orn[4 * (size_t)g], row + 3, 16);
        std::memcpy(&w->linvel[3 * (size_t)g], row + 7, 12); std::memcpy(&w->angvel[3 * (size_t)g], row + 10, 12);
    }
}

void gather_shard(edynhip_world *w, uint32_t r, bool with_growth) {
    Shard &s = w->shards[r];
    if (s.rc != EDYNHIP_OK) return;
    const uint32_t nl = (uint32_t)s.local_ids.size();
    if (nl == 0) return;
    SH_HIP(s, hipSetDevice(s.device));
    hipStream_t st = s.ctx->stream;
    const MonLayout L(s.cap);
    if (with_growth)
        hipLaunchKernelGGL(k_shard_growth, dim3((nl + 255) / 256), dim3(256), 0, st, nl, s.ctx->b, (const float4 *)(s.mon_dev + L.amin0),
                           (const float4 *)(s.mon_dev + L.amax0), s.mon_dev);
    SH_TRY(s, edynhip_pack_state_device(s.ctx, s.pack_dev, 0, nl));
    SH_HIP(s, hipMemcpyAsync(s.pack_host, s.pack_dev, (size_t)nl * 13 * sizeof(float), hipMemcpyDeviceToHost, st));
    SH_HIP(s, hipMemcpyAsync(s.mon_host, s.mon_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SH_HIP(s, hipStreamSynchronize(st));
    scatter_state(w, s);
}

// island boxes of shard r, reduced on its device; also records the AABBs the growth is measured against and clears the growth.
// per_body: every body is its own "island" (right after a shard was built its device labels are not computed yet; body boxes are
// the exact criterion anyway - island boxes are the cheaper, more conservative aggregate).
void island_boxes_shard(edynhip_world *w, uint32_t r, bool per_body) {
    Shard &s = w->shards[r];
    if (s.rc != EDYNHIP_OK) return;
    const uint32_t nl = (uint32_t)s.local_ids.size();
    s.mon_host[1] = 0;
    if (nl == 0) return;
    SH_HIP(s, hipSetDevice(s.device));
    hipStream_t st = s.ctx->stream;
    const MonLayout L(s.cap);
    uint32_t *d = s.mon_dev;
    SH_HIP(s, hipMemsetAsync(d, 0, 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_island_box_clear, dim3((3 * nl + 255) / 256), dim3(256), 0, st, nl, d + L.lo, d + L.hi);
    hipLaunchKernelGGL(k_island_box_reduce, dim3((nl + 255) / 256), dim3(256), 0, st, nl, s.ctx->b, d + L.lo, d + L.hi, (float4 *)(d + L.amin0), (float4 *)(d + L.amax0), per_body);
    hipLaunchKernelGGL(k_island_box_compact, dim3((nl + 255) / 256), dim3(256), 0, st, nl, d + L.lo, d + L.hi, d + 1, d + L.out_label, (float *)(d + L.out_box));
    SH_HIP(s, hipMemcpyAsync(s.mon_host, d, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SH_HIP(s, hipStreamSynchronize(st));
    const uint32_t k = s.mon_host[1];
    if (k) {
        SH_HIP(s, hipMemcpyAsync(s.mon_host + 2, d + L.out_label, (size_t)k * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        SH_HIP(s, hipMemcpyAsync(s.mon_host + 2 + s.cap, d + L.out_box, (size_t)k * 6 * sizeof(float), hipMemcpyDeviceToHost, st));
        SH_HIP(s, hipStreamSynchronize(st));
    }
}

// The approach check on the device-reduced island boxes. Returns true when islands of different shards are within the creation
// margin of each other; otherwise w->budget = how far they may still approach.
int approach_check(edynhip_world *w, bool &close, bool per_body = false, const std::vector<IslandBox> *host_boxes = nullptr) {
    w->pool->run([w, per_body](uint32_t r) { island_boxes_shard(w, r, per_body); });
    EH_TRY(shard_error(w));
    std::vector<IslandBox> boxes;
    if (host_boxes) {   // the caller knows the islands of this state (a re-partition): their boxes, owned by the shard their bodies went to
        boxes = *host_boxes;
        for (IslandBox &b : boxes) b.owner = b.label < w->rank_of.size() ? w->rank_of[b.label] : -1;
    } else
    for (uint32_t r = 0; r < w->shards.size(); ++r) {
        Shard &s = w->shards[r];
        const uint32_t k = s.local_ids.empty() ? 0 : s.mon_host[1];
        const float *bx = (const float *)(s.mon_host + 2 + s.cap);
        for (uint32_t q = 0; q < k; ++q) {
            const uint32_t lead = s.local_ids[s.mon_host[2 + q]];
            if (w->rank_of[lead] != (int32_t)r) continue;   // (cannot happen: replicated bodies are not dynamic)
            IslandBox b;
            for (int d = 0; d < 3; ++d) { b.lo[d] = bx[6 * q + d]; b.hi[d] = bx[6 * q + 3 + d]; }
            b.label = lead; b.owner = (int32_t)r;
            boxes.push_back(b);
        }
    }
    float gmin = kHorizon;
    sweep_boxes(boxes, 0.5f * kHorizon, true, [&](const IslandBox &, const IslandBox &, float gap) { gmin = std::min(gmin, gap); });
    ++w->stats.approach_checks;
    close = !(gmin >= 2 * kCreationMargin);   // also true for NaN
    w->budget = close ? 0.0f : gmin - 2 * kCreationMargin;
    return EDYNHIP_OK;
}

// What the islands carry, per shard, in two parts: LIGHT - island labels, AABBs, sleeping tags and timers (28-45 bytes per body: what a
// re-partition decides on) - and HEAVY - the manifolds with their warm-start impulses and the joints' applied impulses (336 bytes per
// manifold: only worth reading from the shards that are rebuilt).
void collect_shard(edynhip_world *w, uint32_t r, bool light, bool heavy) {
    Shard &s = w->shards[r];
    if (s.rc != EDYNHIP_OK) return;
    const uint32_t nl = (uint32_t)s.local_ids.size();
    if (light) { s.labels.assign(nl, 0); s.aabb.assign((size_t)nl * 6, 0.f); s.asleep.assign(nl, 0); }
    s.manifolds.clear(); s.imp24.clear(); s.imp10.clear();
    if (nl == 0) return;
    SH_HIP(s, hipSetDevice(s.device));
    if (light) {
        SH_TRY(s, edynhip_get_derived(s.ctx, s.aabb.data(), nullptr, s.labels.data()));
        if (w->cfg.flags & EDYNHIP_FLAG_SLEEPING) {
            SH_TRY(s, edynhip_get_asleep(s.ctx, s.asleep.data()));
            s.sleep_label.assign(nl, 0); s.sleep_since.assign(nl, -1.0);
            SH_TRY(s, edynhip_get_sleep_timers(s.ctx, s.sleep_label.data(), s.sleep_since.data(), &s.sleep_clock));
        }
    }
    if (!heavy) return;
    uint32_t nm = 0;
    SH_TRY(s, edynhip_num_manifolds(s.ctx, &nm));
    s.manifolds.resize(nm);
    if (nm) SH_TRY(s, edynhip_get_manifolds(s.ctx, s.manifolds.data(), nm, &nm));
    s.manifolds.resize(nm);
    const uint32_t njl = (uint32_t)s.local_joints.size();
    if (njl) {
        s.imp24.assign((size_t)njl * 24, 0.f); s.imp10.assign((size_t)njl * 10, 0.f);
        SH_TRY(s, edynhip_get_joint_slot_impulses(s.ctx, s.imp24.data()));
        SH_TRY(s, edynhip_get_joint_impulses(s.ctx, s.imp10.data()));
    }
}

inline uint64_t canonical_key(const HostScene &sc, uint32_t a, uint32_t b) {   // (owner << 32) | other: the owner is the dynamic body, the higher index of two
    const bool da = sc.kind[a] == EDYNHIP_KIND_DYNAMIC, db = sc.kind[b] == EDYNHIP_KIND_DYNAMIC;
    const uint32_t owner = (da && db) ? std::max(a, b) : (da ? a : b), other = owner == a ? b : a;
    return ((uint64_t)owner << 32) | other;
}

// gathers the manifolds of all shards, each once, global indices, canonical order
void merge_manifolds(edynhip_world *w, std::vector<edynhip_manifold> &out) {
    out.clear();
    for (uint32_t r = 0; r < w->shards.size(); ++r) {
        Shard &s = w->shards[r];
        for (edynhip_manifold m : s.manifolds) {
            m.body[0] = s.local_ids[m.body[0]]; m.body[1] = s.local_ids[m.body[1]];
            const int32_t ra = w->rank_of[m.body[0]], rb = w->rank_of[m.body[1]];
            if (ra == (int32_t)r || (ra < 0 && rb == (int32_t)r)) out.push_back(m);
        }
    }
    // canonical order: sort (key, position) pairs and move every 336-byte record once (a stable sort of the records themselves, with
    // the key recomputed in every comparison, was 0.15-0.3 s of a re-partition)
    std::vector<std::pair<uint64_t, uint32_t>> order(out.size());
    for (uint32_t k = 0; k < out.size(); ++k) order[k] = {canonical_key(w->scene, out[k].body[0], out[k].body[1]), k};
    std::sort(order.begin(), order.end());   // (ties keep their gathering order: the position is the second key)
    std::vector<edynhip_manifold> sorted(out.size());
    for (uint32_t k = 0; k < out.size(); ++k) sorted[k] = out[order[k].second];
    out.swap(sorted);
}

// `only` (or nullptr = all): the shards to build; the others keep their contexts - their bodies, and therefore their local indices, are unchanged.
int rebuild(edynhip_world *w, const Carry &carry, bool from_state, const std::vector<uint8_t> *only = nullptr) {
    PhaseTrace trace("rebuild");
    w->pool->run([&](uint32_t r) { if (!only || (*only)[r]) build_shard(w, r, carry, from_state); });
    EH_TRY(shard_error(w));
    trace.mark("build the shards (in parallel)");
    w->pool->run([w](uint32_t r) { gather_shard(w, r, false); });
    EH_TRY(shard_error(w));
    trace.mark("gather");
    for (uint32_t r = 0; r < w->shards.size(); ++r) w->stats.bodies_per_shard[r < 16 ? r : 15] = (uint32_t)w->shards[r].owned_local.size();
    w->built = true;
    bool close = false;
    // a fresh partition keeps close islands together: the check sets the budget and records the reference boxes the growth is measured against
    const int rc_check = approach_check(w, close, true, carry.island_boxes.empty() ? nullptr : &carry.island_boxes);
    trace.mark(carry.island_boxes.empty() ? "approach check on body boxes" : "approach check (reference boxes on the devices, island boxes from the host)");
    return rc_check;
}

// Re-partition. FULL (edynhip_world_repartition, on request): everything the islands carry is read from every shard, the islands are
// balanced afresh (longest processing time first over their contact rows) and every shard is rebuilt. STICKY (the approach check found
// islands of different shards within the creation margin of each other - what a step does by itself): the islands stay where they are, a
// group of islands that has to live together moves to the shard that already holds most of it, and only the shards that gain or lose a
// body are read in full and rebuilt - the cost of an island meeting its neighbour follows the two shards involved, not the world
// (round 5: a collapsing 262 144-box scene re-partitioned 18 times in 160 steps at 8 s each when every meeting rebuilt all shards from an
// unrelated partition; scripts/multi_overhead.py).
int repartition(edynhip_world *w, bool sticky) {
    const HostScene &sc = w->scene;
    const uint32_t n = sc.n, W = (uint32_t)w->shards.size();
    PhaseTrace trace(sticky ? "sticky re-partition" : "full re-partition");
    w->pool->run([w, sticky](uint32_t r) { collect_shard(w, r, true, !sticky); });
    EH_TRY(shard_error(w));
    trace.mark("collect light state of every shard");
    std::vector<uint32_t> labels(n);
    std::iota(labels.begin(), labels.end(), 0u);
    std::vector<float> aabb((size_t)n * 6, 0.f);
    std::vector<double> weights(n, 1.0);
    Carry carry;
    carry.any = true;
    if (sc.nj) { carry.imp24.assign((size_t)sc.nj * 24, 0.f); carry.angle.assign(sc.nj, 0.f); }
    if (w->cfg.flags & EDYNHIP_FLAG_SLEEPING) { carry.asleep.assign(n, 0); carry.label.resize(n); std::iota(carry.label.begin(), carry.label.end(), 0u); carry.since.assign(n, -1.0); }
    for (uint32_t r = 0; r < W; ++r) {
        Shard &s = w->shards[r];
        if (!carry.label.empty() && !s.local_ids.empty()) carry.clock = std::max(carry.clock, s.sleep_clock);   // (every shard has taken the same steps)
        for (uint32_t l = 0; l < s.local_ids.size(); ++l) {
            const uint32_t g = s.local_ids[l];
            if (w->rank_of[g] != (int32_t)r) continue;
            labels[g] = s.local_ids[s.labels[l]];
            std::memcpy(&aabb[6 * (size_t)g], &s.aabb[6 * (size_t)l], 24);
            if (!carry.asleep.empty()) carry.asleep[g] = s.asleep[l];
            if (!carry.label.empty() && s.sleep_label[l] < s.local_ids.size()) {
                carry.label[g] = s.local_ids[s.sleep_label[l]];
                if (s.sleep_label[l] == l) carry.since[g] = s.sleep_since[l];
            }
        }
    }
    // what travels with the bodies of the shards read in full (all of them / later: the ones that change)
    auto take_heavy = [&](const std::vector<uint8_t> *only) {
        for (uint32_t r = 0; r < W; ++r) {
            Shard &s = w->shards[r];
            if (only && !(*only)[r]) { s.manifolds.clear(); continue; }
            for (const edynhip_manifold &m : s.manifolds) {   // weight = 1 + the contact points the body takes part in (balance the rows)
                const uint32_t a = s.local_ids[m.body[0]], b = s.local_ids[m.body[1]];
                if (w->rank_of[a] == (int32_t)r) weights[a] += m.num_points;
                if (w->rank_of[b] == (int32_t)r) weights[b] += m.num_points;
            }
            for (uint32_t lj = 0; lj < s.local_joints.size() && !s.imp24.empty(); ++lj) {
                const uint32_t g = s.local_joints[lj];
                std::memcpy(&carry.imp24[(size_t)g * 24], &s.imp24[(size_t)lj * 24], 24 * sizeof(float));
                carry.angle[g] = s.imp10[(size_t)lj * 10 + 9];
            }
        }
        merge_manifolds(w, carry.manifolds);
    };
    if (!sticky) take_heavy(nullptr);
    // islands whose boxes overlap (grown by the creation margin) must end up on one shard: weld them - all such pairs, also those the
    // last partition had co-located (they may still be separate islands, and the partitioner would be free to split them)
    std::vector<IslandBox> boxes;
    {
        std::vector<uint32_t> isl;
        for (uint32_t i = 0; i < n; ++i) if (sc.kind[i] == EDYNHIP_KIND_DYNAMIC) isl.push_back(labels[i]);
        std::sort(isl.begin(), isl.end()); isl.erase(std::unique(isl.begin(), isl.end()), isl.end());
        boxes.resize(isl.size());
        for (size_t k = 0; k < isl.size(); ++k) { for (int d = 0; d < 3; ++d) { boxes[k].lo[d] = 3.0e38f; boxes[k].hi[d] = -3.0e38f; } boxes[k].label = isl[k]; boxes[k].owner = 0; }
        for (uint32_t i = 0; i < n; ++i) {
            if (sc.kind[i] != EDYNHIP_KIND_DYNAMIC || sc.shape_type[i] == EDYNHIP_SHAPE_NONE) continue;   // a shapeless body touches nothing
            IslandBox &b = boxes[std::lower_bound(isl.begin(), isl.end(), labels[i]) - isl.begin()];
            for (int d = 0; d < 3; ++d) { b.lo[d] = std::min(b.lo[d], aabb[6 * (size_t)i + d]); b.hi[d] = std::max(b.hi[d], aabb[6 * (size_t)i + 3 + d]); }
        }
    }
    std::vector<uint32_t> parent(n);
    std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    // The device labels are those of the shards' LAST island stage: a shard that has not stepped since it was built (a re-partition
    // before the first step, or right after a step that re-partitioned by itself) still has the labels it was built with. The edges the
    // island manager connects bodies through are known on the host - every joint and every carried manifold between two dynamic bodies
    // (island_manager.cpp:117-247) - so they are united here, whatever the labels say: cheap, and a joint or a carried manifold can
    // never end up with its bodies on two shards. (A sticky re-partition follows a step of every shard: labels are current.)
    auto unite = [&](uint32_t a, uint32_t b) {
        if (sc.kind[a] != EDYNHIP_KIND_DYNAMIC || sc.kind[b] != EDYNHIP_KIND_DYNAMIC) return;
        const uint32_t ra = find(labels[a]), rb = find(labels[b]);
        if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
    };
    for (uint32_t g = 0; g < sc.nj; ++g) unite(sc.jbody[2 * g], sc.jbody[2 * g + 1]);
    for (const edynhip_manifold &m : carry.manifolds) unite(m.body[0], m.body[1]);
    sweep_boxes(boxes, kCreationMargin, false, [&](const IslandBox &a, const IslandBox &b, float) {
        const uint32_t ra = find(a.label), rb = find(b.label);
        if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
    });
    std::vector<uint32_t> welded(n);
    for (uint32_t i = 0; i < n; ++i) welded[i] = find(labels[i]);
    ++w->stats.repartitions;
    trace.mark("island boxes, welding");
    if (!sticky) {
        w->rank_of.assign(n, -1);
        partition_islands_spatial(n, welded.data(), sc.kind.data(), weights.data(), aabb.data(), W, w->rank_of.data());
        return rebuild(w, carry, true);
    }
    // sticky: a group that spans shards goes to the shard holding most of its bodies (ties: the lower shard); everybody else stays
    std::vector<int32_t> next = w->rank_of;
    {
        std::vector<uint32_t> order;
        for (uint32_t i = 0; i < n; ++i) if (sc.kind[i] == EDYNHIP_KIND_DYNAMIC) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return welded[a] != welded[b] ? welded[a] < welded[b] : a < b; });
        std::vector<uint32_t> count(W);
        for (size_t a = 0; a < order.size();) {
            size_t e = a;
            std::fill(count.begin(), count.end(), 0u);
            while (e < order.size() && welded[order[e]] == welded[order[a]]) { ++count[(uint32_t)w->rank_of[order[e]]]; ++e; }
            uint32_t target = 0, spans = 0;
            for (uint32_t r = 0; r < W; ++r) { if (count[r]) ++spans; if (count[r] > count[target]) target = r; }
            if (spans > 1) for (size_t k = a; k < e; ++k) next[order[k]] = (int32_t)target;
            a = e;
        }
    }
    // Sticky moves only ever add to the shard that holds most of a group: a collapsing or merging scene would drift onto a few shards.
    // Once the heaviest shard carries more than 1.5 x the mean, this meeting takes the full, spatially balanced re-partition instead
    // (the price is one rebuild of every shard, the gain every later step of the slowest shard).
    if (W > 1) {
        std::vector<uint32_t> load(W, 0);
        uint32_t total = 0;
        for (uint32_t i = 0; i < n; ++i) if (next[i] >= 0) { ++load[(uint32_t)next[i]]; ++total; }
        const uint32_t heaviest = *std::max_element(load.begin(), load.end());
        if (total >= 8 * W && (double)heaviest * W > 1.5 * (double)total) {
            ++w->stats.rebalances; --w->stats.repartitions;   // (counted again by the full pass)
            return repartition(w, false);
        }
    }
    std::vector<uint8_t> changed(W, 0);
    for (uint32_t i = 0; i < n; ++i) if (next[i] != w->rank_of[i]) { changed[(uint32_t)next[i]] = 1; changed[(uint32_t)w->rank_of[i]] = 1; }
    if (std::none_of(changed.begin(), changed.end(), [](uint8_t c) { return c != 0; })) {
        // nothing has to move (the close islands already share a shard): only the budget is taken again
        bool close = false;
        EH_TRY(approach_check(w, close, true));
        return EDYNHIP_OK;
    }
    trace.mark("sticky targets");
    w->pool->run([w, &changed](uint32_t r) { if (changed[r]) collect_shard(w, r, false, true); });
    EH_TRY(shard_error(w));
    trace.mark("collect manifolds / joint impulses of the changed shards");
    // joints' applied impulses and angles of the changed shards
    for (uint32_t r = 0; r < W; ++r) {
        Shard &s = w->shards[r];
        if (!changed[r]) continue;
        for (uint32_t lj = 0; lj < s.local_joints.size() && !s.imp24.empty(); ++lj) {
            const uint32_t g = s.local_joints[lj];
            std::memcpy(&carry.imp24[(size_t)g * 24], &s.imp24[(size_t)lj * 24], 24 * sizeof(float));
            carry.angle[g] = s.imp10[(size_t)lj * 10 + 9];
        }
    }
    // Manifolds: a body that stays in its shard keeps its place among the others that stay (local indices ascend with the global ones), so
    // the manifolds that stay ARE in canonical order already; only those whose island moves are taken out, sorted among themselves and
    // merged into their new shard's list. No global sort, one copy per record at most (the merged, sorted list of every changed shard's
    // manifolds was 0.1-0.3 s of a re-partition).
    std::vector<std::vector<edynhip_manifold>> leaving(W);
    w->pool->run([&](uint32_t q) {
        if (!changed[q]) return;
        Shard &s = w->shards[q];
        size_t keep = 0;
        for (size_t k = 0; k < s.manifolds.size(); ++k) {
            edynhip_manifold m = s.manifolds[k];
            m.body[0] = s.local_ids[m.body[0]]; m.body[1] = s.local_ids[m.body[1]];
            const uint32_t dyn = w->rank_of[m.body[0]] >= 0 ? m.body[0] : m.body[1];
            if (w->rank_of[dyn] != (int32_t)q) continue;          // (not this shard's to report: cannot happen - a manifold lives where its dynamic body does)
            if (next[dyn] == (int32_t)q) s.manifolds[keep++] = m; else leaving[q].push_back(m);
        }
        s.manifolds.resize(keep);
    });
    std::vector<std::vector<edynhip_manifold>> incoming(W);
    for (uint32_t q = 0; q < W; ++q)
        for (const edynhip_manifold &m : leaving[q]) {
            const uint32_t dyn = w->rank_of[m.body[0]] >= 0 ? m.body[0] : m.body[1];
            incoming[(uint32_t)next[dyn]].push_back(m);
        }
    carry.per_shard.assign(W, {});
    w->pool->run([&](uint32_t r) {
        if (!changed[r]) return;
        Shard &s = w->shards[r];
        auto key_less = [&](const edynhip_manifold &x, const edynhip_manifold &y) { return canonical_key(sc, x.body[0], x.body[1]) < canonical_key(sc, y.body[0], y.body[1]); };
        if (incoming[r].empty()) { carry.per_shard[r].swap(s.manifolds); return; }
        std::sort(incoming[r].begin(), incoming[r].end(), key_less);
        carry.per_shard[r].resize(s.manifolds.size() + incoming[r].size());
        std::merge(s.manifolds.begin(), s.manifolds.end(), incoming[r].begin(), incoming[r].end(), carry.per_shard[r].begin(), key_less);
        s.manifolds.clear(); s.manifolds.shrink_to_fit();
    });
    trace.mark("sort out the carried manifolds per new shard");
    carry.island_boxes = boxes;
    w->rank_of = next;
    const int rc_rebuild = rebuild(w, carry, true, &changed);
    trace.mark("rebuild changed shards (+ gather, approach check)");
    return rc_rebuild;
}

int ensure_built(edynhip_world *w) {
    if (w->built) return EDYNHIP_OK;
    const HostScene &sc = w->scene;
    const uint32_t n = sc.n, W = (uint32_t)w->shards.size();
    if (n == 0) return w->fail(EDYNHIP_ERR_INVALID, "edynhip_world: no bodies");
    // The islands of the initial state, estimated on the HOST (round 6; until round 5 a probe context on devices[0] held the WHOLE scene and
    // ran broadphase, narrowphase and the island stage - a sharded scene had to fit one GPU, and a quarter-million-body probe cost seconds).
    // An island edge exists once a manifold exists (constraint_util.cpp:76-78: every manifold owns a null_constraint edge), i.e. once two
    // AABBs come within the creation margin (broadphase.hpp:15-18), or through a joint (island_manager.cpp:117-247). The partition only
    // has to keep every TRUE island on one shard, so a conservative edge set is as good as the exact one - it merely co-locates islands
    // that are about to meet anyway: every dynamic shaped body gets the box of its bounding sphere (no collision filters, no exclusions
    // applied), boxes within the margin and jointed bodies are united, the label of a component is its lowest body index, as on the device.
    std::vector<uint32_t> labels(n);
    std::vector<float> aabb0((size_t)n * 6, 0.f);
    {
        std::vector<float> mesh_radius(sc.meshes.size(), 0.f);
        for (size_t k = 0; k < sc.meshes.size(); ++k) {
            const std::vector<float> &v = sc.meshes[k].v;
            double c[3] = {0, 0, 0};
            const size_t nv = v.size() / 3;
            for (size_t i = 0; i < nv; ++i) for (int d = 0; d < 3; ++d) c[d] += v[3 * i + d];
            for (int d = 0; d < 3; ++d) c[d] /= (double)std::max<size_t>(nv, 1);
            double r2 = 0;
            for (size_t i = 0; i < nv; ++i) { double q = 0; for (int d = 0; d < 3; ++d) q += (v[3 * i + d] - c[d]) * (v[3 * i + d] - c[d]); r2 = std::max(r2, q); }
            mesh_radius[k] = 2.0f * (float)std::sqrt(r2);   // (the centroid the library shifts to lies inside the hull: twice the radius about the mean covers it)
        }
        std::vector<IslandBox> boxes;
        for (uint32_t i = 0; i < n; ++i) {
            if (sc.kind[i] != EDYNHIP_KIND_DYNAMIC) continue;
            const float *p = &sc.shape_param[4 * (size_t)i];
            float r = -1.0f;
            switch (sc.shape_type[i]) {
            case EDYNHIP_SHAPE_BOX: r = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); break;
            case EDYNHIP_SHAPE_SPHERE: r = p[0]; break;
            case EDYNHIP_SHAPE_CAPSULE: r = p[0] + p[1]; break;
            case EDYNHIP_SHAPE_CYLINDER: r = std::sqrt(p[0] * p[0] + p[1] * p[1]); break;
            case EDYNHIP_SHAPE_POLYHEDRON: { const size_t id = (size_t)p[0]; r = id < mesh_radius.size() ? mesh_radius[id] : 0.0f; break; }
            default: break;   // no shape (or a plane, which is never dynamic in a sensible scene): touches nothing
            }
            if (r < 0) continue;
            if (!sc.com.empty()) r += std::sqrt(sc.com[3 * (size_t)i] * sc.com[3 * (size_t)i] + sc.com[3 * (size_t)i + 1] * sc.com[3 * (size_t)i + 1] + sc.com[3 * (size_t)i + 2] * sc.com[3 * (size_t)i + 2]);
            IslandBox b;
            for (int d = 0; d < 3; ++d) { b.lo[d] = sc.pos[3 * (size_t)i + d] - r; b.hi[d] = sc.pos[3 * (size_t)i + d] + r; aabb0[6 * (size_t)i + d] = b.lo[d]; aabb0[6 * (size_t)i + 3 + d] = b.hi[d]; }
            b.label = i; b.owner = 0;
            boxes.push_back(b);
        }
        std::vector<uint32_t> parent(n);
        std::iota(parent.begin(), parent.end(), 0u);
        auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        auto unite = [&](uint32_t a, uint32_t b) { const uint32_t ra = find(a), rb = find(b); if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb); };
        sweep_boxes(boxes, kCreationMargin, false, [&](const IslandBox &a, const IslandBox &b, float) { unite(a.label, b.label); });
        for (uint32_t g = 0; g < sc.nj; ++g) {
            const uint32_t a = sc.jbody[2 * g], b = sc.jbody[2 * g + 1];
            if (sc.kind[a] == EDYNHIP_KIND_DYNAMIC && sc.kind[b] == EDYNHIP_KIND_DYNAMIC) unite(a, b);
        }
        for (uint32_t i = 0; i < n; ++i) labels[i] = find(i);
    }
    w->rank_of.assign(n, -1);
    partition_islands_spatial(n, labels.data(), sc.kind.data(), nullptr, aabb0.data(), W, w->rank_of.data());
    w->pos = sc.pos; w->orn = sc.orn; w->linvel = sc.linvel; w->angvel = sc.angvel;
    Carry none;
    return rebuild(w, none, false);
}

template <typename T> void copy_in(std::vector<T> &dst, const T *src, size_t count) { if (src) dst.assign(src, src + count); else dst.clear(); }

}  // namespace

extern "C" {

int edynhip_partition_islands(uint32_t n, const uint32_t *labels, const int32_t *kind, const double *weights, uint32_t world_size, int32_t *rank_of) {
    if ((n && (!labels || !kind || !rank_of)) || world_size == 0) return EDYNHIP_ERR_INVALID;
    partition_islands(n, labels, kind, weights, world_size, rank_of);
    return EDYNHIP_OK;
}

int edynhip_island_boxes_overlap(uint32_t num_islands, const float *boxes6, const uint32_t *labels, const int32_t *owner, float margin, int any_owner,
                                 uint32_t *pairs, uint32_t capacity, uint32_t *num_pairs) {
    if ((num_islands && (!boxes6 || !labels || !owner)) || !num_pairs) return EDYNHIP_ERR_INVALID;
    std::vector<IslandBox> boxes(num_islands);
    for (uint32_t k = 0; k < num_islands; ++k) {
        for (int d = 0; d < 3; ++d) { boxes[k].lo[d] = boxes6[6 * k + d]; boxes[k].hi[d] = boxes6[6 * k + 3 + d]; }
        boxes[k].label = labels[k]; boxes[k].owner = owner[k];
    }
    std::vector<std::pair<uint32_t, uint32_t>> hits;
    sweep_boxes(boxes, margin, !any_owner, [&](const IslandBox &a, const IslandBox &b, float) { hits.emplace_back(std::min(a.label, b.label), std::max(a.label, b.label)); });
    std::sort(hits.begin(), hits.end());
    hits.erase(std::unique(hits.begin(), hits.end()), hits.end());
    *num_pairs = (uint32_t)hits.size();
    if (pairs) {
        if (capacity < hits.size()) return EDYNHIP_ERR_CAPACITY;
        for (size_t k = 0; k < hits.size(); ++k) { pairs[2 * k] = hits[k].first; pairs[2 * k + 1] = hits[k].second; }
    }
    return EDYNHIP_OK;
}

int edynhip_get_island_boxes(edynhip_ctx *c, uint32_t *labels, float *boxes6, uint32_t capacity, uint32_t *num_islands) {
    if (!c || !num_islands) return EDYNHIP_ERR_INVALID;
    const uint32_t n = c->b.n;
    *num_islands = 0;
    if (n == 0) return EDYNHIP_OK;
    EH_HIP(c, hipSetDevice(c->device));
    uint32_t *d = nullptr;
    const MonLayout L(n);
    EH_HIP(c, hipMalloc((void **)&d, L.words * sizeof(uint32_t)));
    uint32_t *count = d + 1, *out_label = d + L.out_label;
    float *out_box = (float *)(d + L.out_box);
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(uint32_t), c->stream);
    hipLaunchKernelGGL(k_island_box_clear, dim3((3 * n + 255) / 256), dim3(256), 0, c->stream, n, d + L.lo, d + L.hi);
    hipLaunchKernelGGL(k_island_box_reduce, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->b, d + L.lo, d + L.hi, (float4 *)(d + L.amin0), (float4 *)(d + L.amax0), false);
    hipLaunchKernelGGL(k_island_box_compact, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d + L.lo, d + L.hi, count, out_label, out_box);
    uint32_t k = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&k, count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    int rc = EDYNHIP_OK;
    if (e == hipSuccess) {
        *num_islands = k;
        if (labels && boxes6) {
            if (capacity < k) rc = set_error(c, EDYNHIP_ERR_CAPACITY, "edynhip_get_island_boxes: capacity");
            else if (k) {
                e = hipMemcpyAsync(labels, out_label, (size_t)k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(boxes6, out_box, (size_t)k * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            }
        }
    }
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(c, EDYNHIP_ERR_HIP, "edynhip_get_island_boxes", e);
    return rc;
}

static std::string g_world_create_error;

edynhip_world *edynhip_world_create(const edynhip_config *cfg, const int32_t *devices, uint32_t num_devices, int *status_out) {
    auto fail = [&](int code, const char *msg) -> edynhip_world * { g_world_create_error = msg; if (status_out) *status_out = code; return nullptr; };
    if (!cfg || !devices || num_devices == 0 || num_devices > 64) return fail(EDYNHIP_ERR_INVALID, "edynhip_world_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(EDYNHIP_ERR_NO_DEVICE, "edynhip_world_create: no HIP device (this library has no CPU fallback)");
    for (uint32_t r = 0; r < num_devices; ++r) if (devices[r] < 0 || devices[r] >= ndev) return fail(EDYNHIP_ERR_INVALID, "edynhip_world_create: device ordinal out of range");
    edynhip_world *w = new edynhip_world();
    w->cfg = *cfg;
    w->devices.assign(devices, devices + num_devices);
    w->shards.resize(num_devices);
    for (uint32_t r = 0; r < num_devices; ++r) w->shards[r].device = devices[r];
    // several shards on ONE device (functional tests) must not promise that device to each of them
    for (uint32_t r = 0; r < num_devices; ++r) for (uint32_t q = 0; q < r; ++q) if (devices[q] == devices[r]) w->cfg.flags &= ~(uint32_t)EDYNHIP_FLAG_EXCLUSIVE_DEVICE;
    w->pool.reset(new Pool(num_devices));
    w->shard_filters.resize(num_devices);
    for (uint32_t r = 0; r < num_devices; ++r) w->shard_filters[r] = edynhip_world::ShardFilter{w, r};
    w->stats.num_shards = num_devices;
    if (status_out) *status_out = EDYNHIP_OK;
    return w;
}

void edynhip_world_destroy(edynhip_world *w) {
    if (!w) return;
    for (Shard &s : w->shards) free_shard(s);
    delete w;
}

const char *edynhip_world_last_error(const edynhip_world *w) { return w ? w->err.c_str() : g_world_create_error.c_str(); }

int edynhip_world_create_convex_mesh(edynhip_world *w, uint32_t num_vertices, const float *vertices, uint32_t num_indices, const uint32_t *indices,
                                     uint32_t num_faces, const uint32_t *faces, uint32_t flags, uint32_t *mesh_id) {
    if (!w || !vertices || !indices || !faces || !mesh_id) return EDYNHIP_ERR_INVALID;
    if (w->built) return w->fail(EDYNHIP_ERR_UNSUPPORTED, "edynhip_world_create_convex_mesh: meshes are created before the bodies");
    HostScene::Mesh m;
    m.v.assign(vertices, vertices + 3 * (size_t)num_vertices); m.idx.assign(indices, indices + num_indices); m.faces.assign(faces, faces + 2 * (size_t)num_faces); m.flags = flags;
    w->scene.meshes.push_back(std::move(m));
    *mesh_id = (uint32_t)w->scene.meshes.size() - 1;
    return EDYNHIP_OK;
}

int edynhip_world_set_bodies(edynhip_world *w, uint32_t n, const edynhip_bodies *in) {
    if (!w || !in || n == 0) return EDYNHIP_ERR_INVALID;
    if (!in->kind || !in->pos || !in->orn || !in->linvel || !in->angvel || !in->mass || !in->shape_type || !in->shape_param || !in->friction || !in->restitution)
        return w->fail(EDYNHIP_ERR_INVALID, "edynhip_world_set_bodies: missing array");
    HostScene &sc = w->scene;
    sc.n = n;
    copy_in(sc.kind, in->kind, n); copy_in(sc.shape_type, in->shape_type, n);
    copy_in(sc.pos, in->pos, 3 * (size_t)n); copy_in(sc.orn, in->orn, 4 * (size_t)n); copy_in(sc.linvel, in->linvel, 3 * (size_t)n); copy_in(sc.angvel, in->angvel, 3 * (size_t)n);
    copy_in(sc.mass, in->mass, n); copy_in(sc.shape_param, in->shape_param, 4 * (size_t)n); copy_in(sc.friction, in->friction, n); copy_in(sc.restitution, in->restitution, n);
    copy_in(sc.inertia, in->has_inertia ? in->inertia : nullptr, 9 * (size_t)n); copy_in(sc.has_inertia, in->inertia ? in->has_inertia : nullptr, n);
    copy_in(sc.group, in->group, n); copy_in(sc.mask, in->mask, n); copy_in(sc.gravity, in->gravity, 3 * (size_t)n);
    copy_in(sc.sleeping_disabled, in->sleeping_disabled, n); copy_in(sc.com, in->center_of_mass, 3 * (size_t)n);
    sc.nj = 0; sc.jtype.clear(); sc.jbody.clear(); sc.jpivot.clear(); sc.jaxis.clear(); sc.jparams.clear(); sc.defs.clear(); sc.exclusions.clear();
    w->built = false; w->stepped = false;
    w->stats.num_bodies = n;
    return EDYNHIP_OK;
}

int edynhip_world_set_joints(edynhip_world *w, uint32_t n, const edynhip_joints *in) {
    if (!w || (n && (!in || !in->type || !in->body || !in->pivot))) return EDYNHIP_ERR_INVALID;
    // The scene description is what the world is (re)built from - with the INITIAL state of edynhip_world_set_bodies. Once the world has
    // stepped, such a rebuild would silently reset the simulation: refused (describe the scene again with edynhip_world_set_bodies -
    // which takes the current state - as the C++ shim does for an edit of a running multi-device world)
    if (w->stepped) return w->fail(EDYNHIP_ERR_UNSUPPORTED, "edynhip_world_set_joints: the world has been stepped; describe the scene again (edynhip_world_set_bodies with the current state) before editing it");
    HostScene &sc = w->scene;
    for (uint32_t k = 0; k < 2 * n; ++k) if (in->body[k] >= sc.n) return w->fail(EDYNHIP_ERR_INVALID, "edynhip_world_set_joints: body index out of range");
    sc.nj = n;
    copy_in(sc.jtype, n ? in->type : nullptr, n); copy_in(sc.jbody, n ? in->body : nullptr, 2 * (size_t)n); copy_in(sc.jpivot, n ? in->pivot : nullptr, 6 * (size_t)n);
    if (n && in->axis) copy_in(sc.jaxis, in->axis, 6 * (size_t)n); else sc.jaxis.assign(6 * (size_t)n, 0.f);
    if (n && in->params) copy_in(sc.jparams, in->params, 10 * (size_t)n); else sc.jparams.assign(10 * (size_t)n, 0.f);
    sc.defs.clear();
    w->built = false;
    return EDYNHIP_OK;
}

int edynhip_world_set_joint_definition(edynhip_world *w, uint32_t joint, const float *frame_a9, const float *frame_b9, const float *params, int generic) {
    if (!w || !frame_a9 || !frame_b9 || !params || joint >= w->scene.nj) return EDYNHIP_ERR_INVALID;
    if (w->stepped) return w->fail(EDYNHIP_ERR_UNSUPPORTED, "edynhip_world_set_joint_definition: the world has been stepped; describe the scene again first (see edynhip_world_set_joints)");
    HostScene::Def d;
    d.joint = joint; d.generic = generic != 0;
    d.fa.assign(frame_a9, frame_a9 + 9); d.fb.assign(frame_b9, frame_b9 + 9); d.p.assign(params, params + (generic ? 60 : 16));
    w->scene.defs.push_back(std::move(d));
    w->built = false;
    return EDYNHIP_OK;
}

int edynhip_world_exclude_collision(edynhip_world *w, uint32_t a, uint32_t b) {
    if (!w || a >= w->scene.n || b >= w->scene.n) return EDYNHIP_ERR_INVALID;
    if (w->stepped) return w->fail(EDYNHIP_ERR_UNSUPPORTED, "edynhip_world_exclude_collision: the world has been stepped; describe the scene again first (see edynhip_world_set_joints)");
    w->scene.exclusions.push_back({a, b});
    w->built = false;
    return EDYNHIP_OK;
}

int edynhip_world_set_pair_filter(edynhip_world *w, edynhip_pair_filter filter, void *user) {
    if (!w) return EDYNHIP_ERR_INVALID;
    w->filter = filter; w->filter_user = user;
    if (!w->built) return EDYNHIP_OK;   // installed when the shards are built
    for (uint32_t r = 0; r < w->shards.size(); ++r) {
        Shard &s = w->shards[r];
        if (!s.ctx) continue;
        const int rc = edynhip_set_pair_filter(s.ctx, filter ? &shard_filter_thunk : nullptr, &w->shard_filters[r]);
        if (rc != EDYNHIP_OK) return w->fail(rc, std::string("edynhip_world_set_pair_filter: shard ") + std::to_string(r) + ": " + edynhip_last_error(s.ctx));
    }
    return EDYNHIP_OK;
}

int edynhip_world_default_should_collide(edynhip_world *w, uint32_t a, uint32_t b) {   // should_collide_default in global indices
    if (!w || a >= w->scene.n || b >= w->scene.n) return EDYNHIP_ERR_INVALID;
    if (a == b) return 0;
    const HostScene &sc = w->scene;
    const uint64_t ga = sc.group.empty() ? ~0ull : sc.group[a], gb = sc.group.empty() ? ~0ull : sc.group[b];
    const uint64_t ma = sc.mask.empty() ? ~0ull : sc.mask[a], mb = sc.mask.empty() ? ~0ull : sc.mask[b];
    if ((ga & mb) == 0 || (gb & ma) == 0) return 0;
    for (const auto &e : sc.exclusions) if ((e[0] == a && e[1] == b) || (e[0] == b && e[1] == a)) return 0;
    return 1;
}

int edynhip_world_step(edynhip_world *w, uint32_t nsteps) {
    if (!w) return EDYNHIP_ERR_INVALID;
    EH_TRY(ensure_built(w));
    for (uint32_t k = 0; k < nsteps; ++k) {
        w->pool->run([w](uint32_t r) {
            Shard &s = w->shards[r];
            if (s.rc != EDYNHIP_OK || s.local_ids.empty()) return;
            if (hipSetDevice(s.device) != hipSuccess) { s.rc = EDYNHIP_ERR_HIP; s.err = "hipSetDevice"; return; }
            SH_TRY(s, edynhip_step(s.ctx, 1));
            gather_shard(w, r, true);
        });
        EH_TRY(shard_error(w));
        ++w->stats.steps;
        w->stepped = true;
        if (w->shards.size() < 2) continue;
        float growth = 0.0f;
        for (Shard &s : w->shards) if (!s.local_ids.empty()) { float g; std::memcpy(&g, s.mon_host, 4); growth = std::max(growth, g); }
        if (2 * growth < w->budget) continue;   // nobody has moved far enough out of the boxes of the last check
        bool close = false;
        EH_TRY(approach_check(w, close));
        if (close) EH_TRY(repartition(w, true));
    }
    return EDYNHIP_OK;
}

int edynhip_world_repartition(edynhip_world *w) {
    if (!w) return EDYNHIP_ERR_INVALID;
    EH_TRY(ensure_built(w));
    return repartition(w, false);
}

int edynhip_world_get_state(edynhip_world *w, float *pos, float *orn, float *linvel, float *angvel) {
    if (!w) return EDYNHIP_ERR_INVALID;
    EH_TRY(ensure_built(w));
    const size_t n = w->scene.n;
    if (pos) std::memcpy(pos, w->pos.data(), 3 * n * sizeof(float));
    if (orn) std::memcpy(orn, w->orn.data(), 4 * n * sizeof(float));
    if (linvel) std::memcpy(linvel, w->linvel.data(), 3 * n * sizeof(float));
    if (angvel) std::memcpy(angvel, w->angvel.data(), 3 * n * sizeof(float));
    return EDYNHIP_OK;
}

int edynhip_world_get_partition(edynhip_world *w, int32_t *rank_of) {
    if (!w || !rank_of) return EDYNHIP_ERR_INVALID;
    EH_TRY(ensure_built(w));
    std::memcpy(rank_of, w->rank_of.data(), w->scene.n * sizeof(int32_t));
    return EDYNHIP_OK;
}

int edynhip_world_get_manifolds(edynhip_world *w, edynhip_manifold *out, uint32_t capacity, uint32_t *n) {
    if (!w || !n) return EDYNHIP_ERR_INVALID;
    EH_TRY(ensure_built(w));
    w->pool->run([w](uint32_t r) { collect_shard(w, r, false, true); });
    EH_TRY(shard_error(w));
    std::vector<edynhip_manifold> all;
    merge_manifolds(w, all);
    *n = (uint32_t)all.size();
    if (out) {
        if (capacity < all.size()) return w->fail(EDYNHIP_ERR_CAPACITY, "edynhip_world_get_manifolds: capacity");
        if (!all.empty()) std::memcpy(out, all.data(), all.size() * sizeof(edynhip_manifold));
    }
    return EDYNHIP_OK;
}

int edynhip_world_get_stats(edynhip_world *w, edynhip_world_stats *out) {
    if (!w || !out) return EDYNHIP_ERR_INVALID;
    *out = w->stats;
    return EDYNHIP_OK;
}

edynhip_ctx *edynhip_world_context(edynhip_world *w, uint32_t shard) {
    if (!w || shard >= w->shards.size() || ensure_built(w) != EDYNHIP_OK) return nullptr;
    return w->shards[shard].ctx;
}

}  // extern "C"
