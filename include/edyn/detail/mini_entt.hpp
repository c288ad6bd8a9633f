// Minimal EnTT-compatible registry, used ONLY when the real EnTT (<entt/entt.hpp>) is not on the include path.
// It covers the subset of the API the edyn:: shim and typical user loops need:
// create / destroy / valid, emplace / emplace_or_replace / get / try_get / all_of / any_of / remove,
// view<Ts...>().each(fn) (+ begin/end over entities), ctx().emplace / get / find / erase / contains.
// With EnTT 3.15 installed this header is never included and the caller's own registry type is used.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <tuple>
#include <typeindex>
#include <unordered_map>
#include <utility>
#include <vector>

namespace entt {

enum class entity : std::uint32_t {};
inline constexpr entity null{static_cast<entity>(0xFFFFFFFFu)};

class registry;
namespace detail {
// on_destroy listeners (the subset of entt::sigh / entt::sink the shim uses): free functions with a payload,
// registry.on_destroy<T>().connect<&fn>(payload) with fn(Payload &, registry &, entity), called BEFORE the component goes away.
struct listener { void (*call)(void *payload, registry &, entity); void *payload; const void *id; };
template <auto Candidate> struct candidate_id { static inline const char tag = 0; };
struct pool_base {
    virtual ~pool_base() = default;
    virtual void remove(entity e) = 0;
    virtual bool contains(entity e) const = 0;
    std::vector<listener> on_destroy;
    registry *owner{nullptr};
};
class destroy_sink {
public:
    explicit destroy_sink(pool_base &p) : p_(p) {}
    template <auto Candidate, typename Payload> void connect(Payload &payload) {
        disconnect<Candidate>(payload);
        p_.on_destroy.push_back({[](void *pl, registry &r, entity e) { Candidate(*static_cast<Payload *>(pl), r, e); }, &payload, &candidate_id<Candidate>::tag});
    }
    template <auto Candidate, typename Payload> void disconnect(Payload &payload) {
        for (std::size_t k = p_.on_destroy.size(); k-- > 0;)
            if (p_.on_destroy[k].payload == &payload && p_.on_destroy[k].id == &candidate_id<Candidate>::tag) p_.on_destroy.erase(p_.on_destroy.begin() + (std::ptrdiff_t)k);
    }
private:
    pool_base &p_;
};
template <typename T>
struct pool final : pool_base {
    std::vector<std::uint32_t> sparse;   // entity -> packed index + 1
    std::vector<entity> packed;
    std::vector<T> data;
    bool contains(entity e) const override {
        auto i = static_cast<std::uint32_t>(e);
        return i < sparse.size() && sparse[i] != 0;
    }
    template <typename... A>
    T &emplace(entity e, A &&...a) {
        auto i = static_cast<std::uint32_t>(e);
        if (i >= sparse.size()) sparse.resize(i + 1, 0);
        if (sparse[i]) { data[sparse[i] - 1] = T{std::forward<A>(a)...}; return data[sparse[i] - 1]; }
        packed.push_back(e);
        data.push_back(T{std::forward<A>(a)...});
        sparse[i] = static_cast<std::uint32_t>(packed.size());
        return data.back();
    }
    T &get(entity e) { return data[sparse[static_cast<std::uint32_t>(e)] - 1]; }
    void remove(entity e) override {   // swap-and-pop, like entt::sparse_set
        if (!contains(e)) return;
        for (std::size_t k = 0; k < on_destroy.size(); ++k) on_destroy[k].call(on_destroy[k].payload, *owner, e);
        auto i = static_cast<std::uint32_t>(e);
        std::uint32_t at = sparse[i] - 1, last = static_cast<std::uint32_t>(packed.size() - 1);
        if (at != last) {
            packed[at] = packed[last];
            data[at] = std::move(data[last]);
            sparse[static_cast<std::uint32_t>(packed[at])] = at + 1;
        }
        packed.pop_back(); data.pop_back(); sparse[i] = 0;
    }
};
}  // namespace detail

class registry {
public:
    class context {
    public:
        template <typename T, typename... A>
        T &emplace(A &&...a) {
            auto &slot = vars_[std::type_index(typeid(T))];
            slot = std::shared_ptr<void>(new T(std::forward<A>(a)...), [](void *p) { delete static_cast<T *>(p); });
            return *static_cast<T *>(slot.get());
        }
        template <typename T> T &get() { return *static_cast<T *>(vars_.at(std::type_index(typeid(T))).get()); }
        template <typename T> const T &get() const { return *static_cast<const T *>(vars_.at(std::type_index(typeid(T))).get()); }
        template <typename T> T *find() {
            auto it = vars_.find(std::type_index(typeid(T)));
            return it == vars_.end() ? nullptr : static_cast<T *>(it->second.get());
        }
        template <typename T> bool contains() const { return vars_.count(std::type_index(typeid(T))) != 0; }
        template <typename T> void erase() { vars_.erase(std::type_index(typeid(T))); }
    private:
        std::unordered_map<std::type_index, std::shared_ptr<void>> vars_;
    };

    entity create() {
        if (!free_.empty()) { entity e = free_.back(); free_.pop_back(); alive_[static_cast<std::uint32_t>(e)] = true; return e; }
        alive_.push_back(true);
        return static_cast<entity>(alive_.size() - 1);
    }
    bool valid(entity e) const { auto i = static_cast<std::uint32_t>(e); return i < alive_.size() && alive_[i]; }
    void destroy(entity e) {
        if (!valid(e)) return;
        for (auto &kv : pools_) kv.second->remove(e);
        alive_[static_cast<std::uint32_t>(e)] = false;
        free_.push_back(e);
    }
    template <typename T, typename... A> T &emplace(entity e, A &&...a) { return assure<T>().emplace(e, std::forward<A>(a)...); }
    template <typename T, typename... A> T &emplace_or_replace(entity e, A &&...a) { return assure<T>().emplace(e, std::forward<A>(a)...); }
    template <typename T> T &get(entity e) { return assure<T>().get(e); }
    template <typename T> T *try_get(entity e) { auto &p = assure<T>(); return p.contains(e) ? &p.get(e) : nullptr; }
    template <typename... T> bool all_of(entity e) { return (assure<T>().contains(e) && ...); }
    template <typename... T> bool any_of(entity e) { return (assure<T>().contains(e) || ...); }
    template <typename T> void remove(entity e) { assure<T>().remove(e); }
    template <typename T> detail::pool<T> &storage() { return assure<T>(); }             // direct pool handle: contains(e) / get(e)
    template <typename T> detail::destroy_sink on_destroy() { return detail::destroy_sink(assure<T>()); }
    context &ctx() { return ctx_; }
    // read-only access through a const registry (what a should_collide predicate is handed): same pools, nothing is created that a
    // non-const call would not create
    template <typename T> const T &get(entity e) const { return const_cast<registry *>(this)->template get<T>(e); }
    template <typename T> const T *try_get(entity e) const { return const_cast<registry *>(this)->template try_get<T>(e); }
    template <typename... T> bool all_of(entity e) const { return const_cast<registry *>(this)->template all_of<T...>(e); }
    template <typename... T> bool any_of(entity e) const { return const_cast<registry *>(this)->template any_of<T...>(e); }
    const context &ctx() const { return ctx_; }

    template <typename First, typename... Rest>
    class basic_view {
    public:
        explicit basic_view(registry &r) : r_(r) {}
        template <typename F>
        void each(F f) {
            auto &lead = r_.assure<First>();
            for (std::size_t k = lead.packed.size(); k-- > 0;) {   // back to front, like EnTT
                entity e = lead.packed[k];
                if ((r_.assure<Rest>().contains(e) && ...)) {
                    if constexpr (std::is_invocable_v<F, entity, First &, Rest &...>) f(e, lead.data[k], r_.assure<Rest>().get(e)...);
                    else f(lead.data[k], r_.assure<Rest>().get(e)...);
                }
            }
        }
        std::vector<entity> entities() {
            std::vector<entity> out;
            auto &lead = r_.assure<First>();
            for (std::size_t k = lead.packed.size(); k-- > 0;)
                if ((r_.assure<Rest>().contains(lead.packed[k]) && ...)) out.push_back(lead.packed[k]);
            return out;
        }
        template <typename T> T &get(entity e) { return r_.assure<T>().get(e); }
        bool contains(entity e) { return r_.all_of<First, Rest...>(e); }
    private:
        registry &r_;
    };
    template <typename... T> basic_view<T...> view() { return basic_view<T...>(*this); }

private:
    template <typename T>
    detail::pool<T> &assure() {
        auto &slot = pools_[std::type_index(typeid(T))];
        if (!slot) { slot = std::make_unique<detail::pool<T>>(); slot->owner = this; }
        return *static_cast<detail::pool<T> *>(slot.get());
    }
    std::unordered_map<std::type_index, std::unique_ptr<detail::pool_base>> pools_;
    std::vector<bool> alive_;
    std::vector<entity> free_;
    context ctx_;
};

}  // namespace entt
