// ORACLE — TEST INFRASTRUCTURE ONLY.
// A from-scratch, single-header implementation of the small part of the EnTT 3.15 API that the
// reference's simulation path uses: entity ids, sparse_set, storage with
// construct/update/destroy signals, views with exclusion lists, registry + context variables,
// sigh/sink/scoped_connection, delegate, any, type_id. EnTT itself is not in this image (no network);
// this file exists so that the REAL reference translation units under /root/reference can be compiled
// where they lie into oracle/_ref/libedynref.so and used as the checker (never as the product).
//
// Semantics that the reference's behaviour depends on are kept as in EnTT:
//   * packed arrays iterate from the LAST element to the first; erase = swap-and-pop;
//   * a multi-type view leads with the smallest pool (first one on ties);
//   * destroying an entity removes it from the pools in reverse order of pool creation;
//   * on_destroy fires before the component goes away, on_construct after it exists;
//   * signal listeners run newest first;
//   * entity ids are 20-bit index + 12-bit version, destroyed ids are recycled LIFO.
#ifndef ENTT_MIN_HPP
#define ENTT_MIN_HPP

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iterator>
#include <memory>
#include <new>
#include <string_view>
#include <tuple>
#include <type_traits>
#include <typeindex>
#include <unordered_map>
#include <utility>
#include <vector>

// Misuse (a component that is not there, a dead entity) stops with a backtrace instead of corrupting the run.
#define ENTT_MIN_ASSERT(cond) ((cond) ? (void)0 : ::entt::internal::assert_fail(#cond, __LINE__))

namespace entt {

namespace internal {
[[noreturn]] inline void assert_fail(const char *what, int line) {
    std::fprintf(stderr, "entt_min: assertion `%s' failed at entt_min.hpp:%d\n", what, line);
    void *frames[48];
    backtrace_symbols_fd(frames, backtrace(frames, 48), 2);
    std::abort();
}
}  // namespace internal

using id_type = std::uint32_t;

// ------------------------------------------------------------------------------------------- entity
enum class entity : std::uint32_t {};

namespace internal {
constexpr std::uint32_t entity_mask = 0xFFFFFu;
constexpr std::uint32_t version_mask = 0xFFFu;
constexpr std::uint32_t version_shift = 20u;
}  // namespace internal

constexpr std::uint32_t to_integral(entity e) noexcept { return static_cast<std::uint32_t>(e); }
constexpr std::uint32_t to_entity(entity e) noexcept { return to_integral(e) & internal::entity_mask; }
constexpr std::uint32_t to_version(entity e) noexcept {
    return (to_integral(e) >> internal::version_shift) & internal::version_mask;
}

struct null_t {
    constexpr operator entity() const noexcept { return entity{internal::entity_mask | (internal::version_mask << internal::version_shift)}; }
    constexpr bool operator==(null_t) const noexcept { return true; }
    constexpr bool operator!=(null_t) const noexcept { return false; }
    constexpr bool operator==(entity e) const noexcept { return to_entity(e) == internal::entity_mask; }
    constexpr bool operator!=(entity e) const noexcept { return !(*this == e); }
};
constexpr bool operator==(entity e, null_t n) noexcept { return n == e; }
constexpr bool operator!=(entity e, null_t n) noexcept { return !(n == e); }
inline constexpr null_t null{};

struct tombstone_t {
    constexpr operator entity() const noexcept { return entity{internal::entity_mask | (internal::version_mask << internal::version_shift)}; }
    constexpr bool operator==(entity e) const noexcept { return to_version(e) == internal::version_mask; }
    constexpr bool operator!=(entity e) const noexcept { return !(*this == e); }
};
constexpr bool operator==(entity e, tombstone_t t) noexcept { return t == e; }
constexpr bool operator!=(entity e, tombstone_t t) noexcept { return !(t == e); }
inline constexpr tombstone_t tombstone{};

// ------------------------------------------------------------------------------------------ type ids
namespace internal {
inline id_type next_type_index() {
    static id_type counter = 0;
    return counter++;
}
}  // namespace internal

template <typename T>
struct type_index {
    static id_type value() noexcept {
        static const id_type v = internal::next_type_index();
        return v;
    }
};

template <typename T>
struct type_hash {
    static id_type value() noexcept {
        // FNV-1a over the mangled name: stable inside one process, which is all that is needed here.
        static const id_type v = [] {
            const char *n = typeid(T).name();
            std::uint32_t h = 2166136261u;
            while (*n) { h ^= static_cast<unsigned char>(*n++); h *= 16777619u; }
            return h;
        }();
        return v;
    }
};

struct type_info {
    id_type seq, identifier;
    std::string_view alias;
    id_type index() const noexcept { return seq; }
    id_type hash() const noexcept { return identifier; }
    std::string_view name() const noexcept { return alias; }
    bool operator==(const type_info &o) const noexcept { return identifier == o.identifier; }
    bool operator!=(const type_info &o) const noexcept { return identifier != o.identifier; }
    bool operator<(const type_info &o) const noexcept { return seq < o.seq; }
};

template <typename T>
const type_info &type_id() noexcept {
    using U = std::remove_cv_t<std::remove_reference_t<T>>;
    static const type_info info{type_index<U>::value(), type_hash<U>::value(), typeid(U).name()};
    return info;
}
template <typename T>
const type_info &type_id(T &&) noexcept { return type_id<std::remove_cv_t<std::remove_reference_t<T>>>(); }

struct identity {
    template <typename T>
    constexpr T &&operator()(T &&v) const noexcept { return std::forward<T>(v); }
};

template <typename K, typename V, typename H = std::hash<K>, typename E = std::equal_to<K>>
using dense_map = std::unordered_map<K, V, H, E>;

// hashed_string / literals: only the spelling "name"_hs is needed.
struct hashed_string {
    id_type h;
    constexpr hashed_string(const char *s) : h(2166136261u) { while (*s) { h ^= static_cast<unsigned char>(*s++); h *= 16777619u; } }
    constexpr operator id_type() const noexcept { return h; }
    constexpr id_type value() const noexcept { return h; }
};
namespace literals {
constexpr hashed_string operator""_hs(const char *s, std::size_t) { return hashed_string{s}; }
}  // namespace literals

// ---------------------------------------------------------------------------------------------- any
// Owning, move-only type-erased value (what the reference's message queues need).
class any {
    struct base {
        virtual ~base() = default;
        virtual void *ptr() noexcept = 0;
        virtual const std::type_info &ti() const noexcept = 0;
        virtual const type_info &info() const noexcept = 0;
    };
    template <typename T>
    struct model final : base {
        T value;
        template <typename... Args>
        explicit model(std::in_place_t, Args &&...args) : value(make(std::forward<Args>(args)...)) {}
        template <typename... Args>
        static T make(Args &&...args) {
            if constexpr (std::is_aggregate_v<T>) return T{std::forward<Args>(args)...};
            else return T(std::forward<Args>(args)...);
        }
        void *ptr() noexcept override { return &value; }
        const std::type_info &ti() const noexcept override { return typeid(T); }
        const type_info &info() const noexcept override { return type_id<T>(); }
    };
    std::unique_ptr<base> m_value;
public:
    any() = default;
    template <typename T, typename... Args>
    explicit any(std::in_place_type_t<T>, Args &&...args) : m_value(new model<T>(std::in_place, std::forward<Args>(args)...)) {}
    template <typename T, typename = std::enable_if_t<!std::is_same_v<std::decay_t<T>, any>>>
    any(T &&v) : m_value(new model<std::decay_t<T>>(std::in_place, std::forward<T>(v))) {}
    any(any &&) noexcept = default;
    any &operator=(any &&) noexcept = default;
    explicit operator bool() const noexcept { return static_cast<bool>(m_value); }
    const type_info &type() const noexcept { return m_value ? m_value->info() : type_id<void>(); }
    template <typename T> T *try_as() noexcept { return m_value && m_value->ti() == typeid(T) ? static_cast<T *>(m_value->ptr()) : nullptr; }
    template <typename T> const T *try_as() const noexcept { return m_value && m_value->ti() == typeid(T) ? static_cast<const T *>(m_value->ptr()) : nullptr; }
};
template <typename T> T *any_cast(any *a) noexcept { return a->template try_as<T>(); }
template <typename T> const T *any_cast(const any *a) noexcept { return a->template try_as<T>(); }
template <typename T> T any_cast(any &a) { return *a.template try_as<std::remove_cv_t<std::remove_reference_t<T>>>(); }

// ----------------------------------------------------------------------------------------- delegate
template <auto>
struct connect_arg_t { explicit connect_arg_t() = default; };
template <auto Candidate>
inline constexpr connect_arg_t<Candidate> connect_arg{};

template <typename>
class delegate;

namespace internal {
// EnTT lets a candidate ignore trailing arguments of the delegate's signature: call it with the
// longest prefix of the arguments it accepts.
template <std::size_t N, typename F, typename Lead, typename ArgTuple, typename = std::make_index_sequence<N>>
struct prefix_invocable;
template <std::size_t N, typename F, typename... Lead, typename ArgTuple, std::size_t... I>
struct prefix_invocable<N, F, std::tuple<Lead...>, ArgTuple, std::index_sequence<I...>>
    : std::is_invocable<F, Lead..., std::tuple_element_t<I, ArgTuple>...> {};

template <typename F, typename Lead, typename ArgTuple, std::size_t N = std::tuple_size_v<ArgTuple>>
constexpr std::size_t usable_arity() {
    if constexpr (prefix_invocable<N, F, Lead, ArgTuple>::value) return N;
    else if constexpr (N == 0u) return static_cast<std::size_t>(-1);
    else return usable_arity<F, Lead, ArgTuple, N - 1u>();
}
template <auto Candidate, typename ArgTuple, std::size_t... I, typename... Lead>
decltype(auto) invoke_prefix(std::index_sequence<I...>, ArgTuple &&args, Lead &&...lead) {
    return std::invoke(Candidate, std::forward<Lead>(lead)..., std::get<I>(std::forward<ArgTuple>(args))...);
}
}  // namespace internal

template <typename Ret, typename... Args>
class delegate<Ret(Args...)> {
    using fn_t = Ret(const void *, Args...);
    fn_t *m_fn{nullptr};
    const void *m_instance{nullptr};

    template <auto Candidate, typename... Lead>
    static Ret call(std::tuple<Args &&...> args, Lead &&...lead) {
        constexpr auto n = internal::usable_arity<decltype(Candidate), std::tuple<Lead...>, std::tuple<Args...>>();
        static_assert(n != static_cast<std::size_t>(-1), "candidate cannot be invoked with the delegate's arguments");
        return Ret(internal::invoke_prefix<Candidate>(std::make_index_sequence<n>{}, std::move(args), std::forward<Lead>(lead)...));
    }

public:
    delegate() = default;
    template <auto Candidate>
    delegate(connect_arg_t<Candidate>) { connect<Candidate>(); }
    template <auto Candidate, typename Type>
    delegate(connect_arg_t<Candidate>, Type &&instance) { connect<Candidate>(std::forward<Type>(instance)); }

    template <auto Candidate>
    void connect() {
        m_instance = nullptr;
        m_fn = [](const void *, Args... args) -> Ret { return call<Candidate>(std::forward_as_tuple(std::forward<Args>(args)...)); };
    }
    template <auto Candidate, typename Type>
    void connect(Type &instance) {
        m_instance = &instance;
        m_fn = [](const void *payload, Args... args) -> Ret {
            Type *curr = static_cast<Type *>(const_cast<void *>(payload));
            return call<Candidate>(std::forward_as_tuple(std::forward<Args>(args)...), *curr);
        };
    }
    template <auto Candidate, typename Type>
    void connect(Type *instance) {
        m_instance = instance;
        m_fn = [](const void *payload, Args... args) -> Ret {
            Type *curr = static_cast<Type *>(const_cast<void *>(payload));
            return call<Candidate>(std::forward_as_tuple(std::forward<Args>(args)...), curr);
        };
    }
    void reset() noexcept { m_fn = nullptr; m_instance = nullptr; }
    const void *data() const noexcept { return m_instance; }
    Ret operator()(Args... args) const { return m_fn(m_instance, std::forward<Args>(args)...); }
    explicit operator bool() const noexcept { return m_fn != nullptr; }
    bool operator==(const delegate &o) const noexcept { return m_fn == o.m_fn && m_instance == o.m_instance; }
    bool operator!=(const delegate &o) const noexcept { return !(*this == o); }
};

// --------------------------------------------------------------------------------------------- sigh
template <typename>
class sigh;
template <typename>
class sink;

class connection {
    template <typename> friend class sink;
    std::function<void()> m_disconnect;
public:
    connection() = default;
    explicit connection(std::function<void()> fn) : m_disconnect(std::move(fn)) {}
    explicit operator bool() const noexcept { return static_cast<bool>(m_disconnect); }
    void release() {
        if (m_disconnect) { m_disconnect(); m_disconnect = nullptr; }
    }
};

struct scoped_connection {
    scoped_connection() = default;
    scoped_connection(const connection &c) : conn(c) {}
    scoped_connection(const scoped_connection &) = delete;
    scoped_connection(scoped_connection &&o) noexcept : conn(std::exchange(o.conn, connection{})) {}
    scoped_connection &operator=(const scoped_connection &) = delete;
    scoped_connection &operator=(scoped_connection &&o) noexcept {
        conn.release();
        conn = std::exchange(o.conn, connection{});
        return *this;
    }
    scoped_connection &operator=(connection c) {
        conn.release();
        conn = std::move(c);
        return *this;
    }
    ~scoped_connection() { conn.release(); }
    explicit operator bool() const noexcept { return static_cast<bool>(conn); }
    void release() { conn.release(); }
private:
    connection conn;
};

template <typename Ret, typename... Args>
class sigh<Ret(Args...)> {
    friend class sink<sigh<Ret(Args...)>>;
    // Shared so that a connection handle outliving the signal stays harmless.
    std::shared_ptr<std::vector<delegate<Ret(Args...)>>> m_calls = std::make_shared<std::vector<delegate<Ret(Args...)>>>();
public:
    using sink_type = sink<sigh<Ret(Args...)>>;
    std::size_t size() const noexcept { return m_calls->size(); }
    bool empty() const noexcept { return m_calls->empty(); }
    void publish(Args... args) const {
        // Newest listener first, like EnTT.
        for (auto pos = m_calls->size(); pos; --pos) {
            (*m_calls)[pos - 1u](args...);
            if (pos > m_calls->size()) pos = m_calls->size() + 1u;  // a listener disconnected something
        }
    }
};

template <typename Ret, typename... Args>
class sink<sigh<Ret(Args...)>> {
    using signal_type = sigh<Ret(Args...)>;
    using delegate_type = delegate<Ret(Args...)>;
    signal_type *m_signal;

    connection make_connection(const delegate_type &call) {
        std::weak_ptr<std::vector<delegate_type>> weak = m_signal->m_calls;
        return connection{[weak, call]() {
            if (auto calls = weak.lock()) {
                calls->erase(std::remove(calls->begin(), calls->end(), call), calls->end());
            }
        }};
    }
    void drop(const delegate_type &call) {
        auto &calls = *m_signal->m_calls;
        calls.erase(std::remove(calls.begin(), calls.end(), call), calls.end());
    }

public:
    sink(signal_type &ref) noexcept : m_signal(&ref) {}
    bool empty() const noexcept { return m_signal->empty(); }

    template <auto Candidate, typename... Type>
    connection connect(Type &&...instance) {
        delegate_type call;
        call.template connect<Candidate>(std::forward<Type>(instance)...);
        drop(call);
        m_signal->m_calls->push_back(call);
        return make_connection(call);
    }
    template <auto Candidate, typename... Type>
    void disconnect(Type &&...instance) {
        delegate_type call;
        call.template connect<Candidate>(std::forward<Type>(instance)...);
        drop(call);
    }
    template <typename Type>
    void disconnect(Type &instance) {
        auto &calls = *m_signal->m_calls;
        const void *ptr = &instance;
        calls.erase(std::remove_if(calls.begin(), calls.end(), [ptr](const delegate_type &d) { return d.data() == ptr; }), calls.end());
    }
    template <typename Type>
    void disconnect(Type *instance) { if (instance) disconnect(*instance); }
    void disconnect() { m_signal->m_calls->clear(); }
};

template <typename Ret, typename... Args>
sink(sigh<Ret(Args...)> &) -> sink<sigh<Ret(Args...)>>;

// --------------------------------------------------------------------------------------- sparse_set
enum class deletion_policy : std::uint8_t { swap_and_pop = 0u, in_place = 1u, swap_only = 2u };

template <typename... Type> struct exclude_t { explicit constexpr exclude_t() = default; };
template <typename... Type> inline constexpr exclude_t<Type...> exclude{};
template <typename... Type> struct get_t { explicit constexpr get_t() = default; };
template <typename... Type> inline constexpr get_t<Type...> get{};

class sparse_set {
    static constexpr std::uint32_t npos = 0xFFFFFFFFu;

protected:
    std::vector<entity> m_packed;
    std::vector<std::uint32_t> m_sparse;

    virtual void swap_payload(std::size_t, std::size_t) {}
    virtual void pop_payload(std::size_t) {}
    virtual void about_to_pop(entity) {}

    void raw_push(entity e) {
        const auto idx = to_entity(e);
        if (idx >= m_sparse.size()) m_sparse.resize(std::max<std::size_t>(idx + 1u, m_sparse.size() * 2u), npos);
        ENTT_MIN_ASSERT(m_sparse[idx] == npos);
        m_sparse[idx] = static_cast<std::uint32_t>(m_packed.size());
        m_packed.push_back(e);
    }
    void raw_swap_and_pop(std::size_t pos) {
        const auto last = m_packed.size() - 1u;
        const auto idx = to_entity(m_packed[pos]);
        if (pos != last) {
            swap_payload(pos, last);
            m_packed[pos] = m_packed[last];
            m_sparse[to_entity(m_packed[pos])] = static_cast<std::uint32_t>(pos);
        }
        pop_payload(last);
        m_sparse[idx] = npos;
        m_packed.pop_back();
    }

public:
    using entity_type = entity;
    using size_type = std::size_t;

    // Index-based, walks the packed array backwards; survives push_back and erasing the current element.
    class iterator {
        const std::vector<entity> *m_vec{nullptr};
        std::ptrdiff_t m_offset{0};
    public:
        using value_type = entity;
        using pointer = const entity *;
        using reference = const entity &;
        using difference_type = std::ptrdiff_t;
        using iterator_category = std::random_access_iterator_tag;
        iterator() = default;
        iterator(const std::vector<entity> &v, std::ptrdiff_t off) : m_vec(&v), m_offset(off) {}
        iterator &operator++() noexcept { --m_offset; return *this; }
        iterator operator++(int) noexcept { auto c = *this; --m_offset; return c; }
        iterator &operator--() noexcept { ++m_offset; return *this; }
        iterator operator--(int) noexcept { auto c = *this; ++m_offset; return c; }
        iterator &operator+=(difference_type n) noexcept { m_offset -= n; return *this; }
        iterator &operator-=(difference_type n) noexcept { m_offset += n; return *this; }
        iterator operator+(difference_type n) const noexcept { return iterator{*m_vec, m_offset - n}; }
        iterator operator-(difference_type n) const noexcept { return iterator{*m_vec, m_offset + n}; }
        difference_type operator-(const iterator &o) const noexcept { return o.m_offset - m_offset; }
        reference operator[](difference_type n) const noexcept { return (*m_vec)[static_cast<std::size_t>(m_offset - n - 1)]; }
        reference operator*() const noexcept { return (*m_vec)[static_cast<std::size_t>(m_offset - 1)]; }
        pointer operator->() const noexcept { return &**this; }
        bool operator==(const iterator &o) const noexcept { return m_offset == o.m_offset; }
        bool operator!=(const iterator &o) const noexcept { return m_offset != o.m_offset; }
        bool operator<(const iterator &o) const noexcept { return m_offset > o.m_offset; }
        bool operator>(const iterator &o) const noexcept { return m_offset < o.m_offset; }
        bool operator<=(const iterator &o) const noexcept { return m_offset >= o.m_offset; }
        bool operator>=(const iterator &o) const noexcept { return m_offset <= o.m_offset; }
        difference_type index() const noexcept { return m_offset - 1; }
    };
    using const_iterator = iterator;
    using reverse_iterator = std::vector<entity>::const_iterator;

    sparse_set() = default;
    sparse_set(const sparse_set &o) : m_packed(o.m_packed), m_sparse(o.m_sparse) {}
    sparse_set(sparse_set &&) noexcept = default;
    sparse_set &operator=(const sparse_set &o) { m_packed = o.m_packed; m_sparse = o.m_sparse; return *this; }
    sparse_set &operator=(sparse_set &&) noexcept = default;
    virtual ~sparse_set() = default;

    size_type size() const noexcept { return m_packed.size(); }
    bool empty() const noexcept { return m_packed.empty(); }
    const entity *data() const noexcept { return m_packed.data(); }
    void reserve(size_type n) { m_packed.reserve(n); }
    size_type capacity() const noexcept { return m_packed.capacity(); }
    void shrink_to_fit() {}
    size_type free_list() const noexcept { return m_packed.size(); }

    iterator begin() const noexcept { return iterator{m_packed, static_cast<std::ptrdiff_t>(m_packed.size())}; }
    iterator end() const noexcept { return iterator{m_packed, 0}; }
    iterator cbegin() const noexcept { return begin(); }
    iterator cend() const noexcept { return end(); }
    reverse_iterator rbegin() const noexcept { return m_packed.cbegin(); }
    reverse_iterator rend() const noexcept { return m_packed.cend(); }

    bool contains(entity e) const noexcept {
        const auto idx = to_entity(e);
        return idx < m_sparse.size() && m_sparse[idx] != npos && m_packed[m_sparse[idx]] == e;
    }
    size_type index(entity e) const noexcept { ENTT_MIN_ASSERT(contains(e)); return m_sparse[to_entity(e)]; }
    iterator find(entity e) const noexcept { return contains(e) ? iterator{m_packed, static_cast<std::ptrdiff_t>(index(e) + 1u)} : end(); }
    entity operator[](size_type pos) const noexcept { return m_packed[pos]; }
    entity at(size_type pos) const noexcept { return pos < m_packed.size() ? m_packed[pos] : entity{null}; }

    iterator push(entity e) { raw_push(e); return iterator{m_packed, static_cast<std::ptrdiff_t>(m_packed.size())}; }
    template <typename It>
    iterator push(It first, It last) { for (; first != last; ++first) raw_push(*first); return begin(); }
    // Older spellings.
    iterator emplace(entity e) { return push(e); }
    template <typename It>
    iterator insert(It first, It last) { return push(first, last); }

    void erase(entity e) {
        ENTT_MIN_ASSERT(contains(e));
        about_to_pop(e);
        raw_swap_and_pop(index(e));
    }
    template <typename It>
    void erase(It first, It last) { for (; first != last; ++first) erase(*first); }
    bool remove(entity e) { return contains(e) ? (erase(e), true) : false; }
    template <typename It>
    size_type remove(It first, It last) { size_type n = 0; for (; first != last; ++first) n += remove(*first); return n; }

    virtual void clear() {
        for (auto pos = m_packed.size(); pos; --pos) about_to_pop(m_packed[pos - 1u]);
        for (auto pos = m_packed.size(); pos; --pos) pop_payload(pos - 1u);
        m_packed.clear();
        std::fill(m_sparse.begin(), m_sparse.end(), npos);
    }
    void swap_elements(entity a, entity b) {
        const auto pa = index(a), pb = index(b);
        swap_payload(pa, pb);
        std::swap(m_packed[pa], m_packed[pb]);
        m_sparse[to_entity(a)] = static_cast<std::uint32_t>(pb);
        m_sparse[to_entity(b)] = static_cast<std::uint32_t>(pa);
    }
    deletion_policy policy() const noexcept { return deletion_policy::swap_and_pop; }
};
template <typename = entity>
using basic_sparse_set = sparse_set;

// ------------------------------------------------------------------------------------------ storage
class registry;

namespace internal {
template <typename T>
inline constexpr bool is_empty_component = std::is_empty_v<T>;
}

// Component pool: entities in a sparse set + values in fixed-size pages (stable addresses while the
// pool grows, like EnTT). Empty types carry no payload. Signals take (registry&, entity).
template <typename T>
class storage : public sparse_set {
    static_assert(!std::is_const_v<T>, "pools are keyed by the non-const type");
    static constexpr std::size_t page_size = 1024u;
    static constexpr bool has_payload = !internal::is_empty_component<T>;
    using slot_t = std::aligned_storage_t<has_payload ? sizeof(T) : 1u, has_payload ? alignof(T) : 1u>;

    std::vector<std::unique_ptr<slot_t[]>> m_pages;
    registry *m_owner{nullptr};
    sigh<void(registry &, entity)> m_construction, m_update, m_destruction;

    T *slot(std::size_t pos) const noexcept {
        return std::launder(reinterpret_cast<T *>(&m_pages[pos / page_size][pos % page_size]));
    }
    void assure_slot(std::size_t pos) {
        while (pos / page_size >= m_pages.size()) m_pages.emplace_back(new slot_t[page_size]);
    }

protected:
    void swap_payload(std::size_t a, std::size_t b) override {
        if constexpr (has_payload) {
            // Only used as "move b into a's place" (swap-and-pop) or as a true swap.
            using std::swap;
            swap(*slot(a), *slot(b));
        }
    }
    void pop_payload(std::size_t pos) override {
        if constexpr (has_payload) slot(pos)->~T();
    }
    void about_to_pop(entity e) override {
        if (m_owner && !m_destruction.empty()) m_destruction.publish(*m_owner, e);
    }

public:
    using value_type = T;
    using element_type = T;

    storage() = default;
    storage(const storage &) = delete;
    storage &operator=(const storage &) = delete;
    storage(storage &&) noexcept = default;
    storage &operator=(storage &&) noexcept = default;
    ~storage() override {
        if constexpr (has_payload) for (std::size_t pos = 0; pos < m_packed.size(); ++pos) slot(pos)->~T();
    }

    void bind(registry &owner) noexcept { m_owner = &owner; }
    auto on_construct() noexcept { return sink<sigh<void(registry &, entity)>>{m_construction}; }
    auto on_update() noexcept { return sink<sigh<void(registry &, entity)>>{m_update}; }
    auto on_destroy() noexcept { return sink<sigh<void(registry &, entity)>>{m_destruction}; }

    template <typename... Args>
    decltype(auto) emplace(entity e, Args &&...args) {
        if constexpr (has_payload) {
            const auto pos = m_packed.size();
            assure_slot(pos);
            if constexpr (std::is_aggregate_v<T> && (sizeof...(Args) != 0u || !std::is_default_constructible_v<T>)) {
                ::new (static_cast<void *>(&m_pages[pos / page_size][pos % page_size])) T{std::forward<Args>(args)...};
            } else {
                ::new (static_cast<void *>(&m_pages[pos / page_size][pos % page_size])) T(std::forward<Args>(args)...);
            }
            raw_push(e);
            if (m_owner && !m_construction.empty()) m_construction.publish(*m_owner, e);
            return static_cast<T &>(*slot(index(e)));
        } else {
            raw_push(e);
            if (m_owner && !m_construction.empty()) m_construction.publish(*m_owner, e);
        }
    }
    template <typename It>
    void insert(It first, It last, const T &value = {}) { for (; first != last; ++first) emplace(*first, value); }

    template <typename... Func>
    decltype(auto) patch(entity e, Func &&...func) {
        if constexpr (has_payload) {
            T &elem = *slot(index(e));
            (std::forward<Func>(func)(elem), ...);
            if (m_owner && !m_update.empty()) m_update.publish(*m_owner, e);
            return static_cast<T &>(*slot(index(e)));
        } else {
            if (m_owner && !m_update.empty()) m_update.publish(*m_owner, e);
        }
    }

    template <bool P = has_payload, typename = std::enable_if_t<P>>
    T &get(entity e) const noexcept { return *slot(index(e)); }
    auto get_as_tuple(entity e) const noexcept {
        if constexpr (has_payload) return std::forward_as_tuple(*slot(index(e)));
        else return std::make_tuple();
    }
    T *raw_at(std::size_t pos) const noexcept { return slot(pos); }
};

template <typename T>
using storage_for_t = std::conditional_t<std::is_const_v<T>, const storage<std::remove_const_t<T>>, storage<std::remove_const_t<T>>>;

// --------------------------------------------------------------------------------------------- view
template <typename, typename>
class basic_view;

namespace internal {
template <typename S> using storage_value_t = typename std::remove_const_t<S>::value_type;
template <typename S> using storage_elem_t = std::conditional_t<std::is_const_v<S>, const storage_value_t<S>, storage_value_t<S>>;
template <typename S> inline constexpr bool storage_has_payload = !is_empty_component<storage_value_t<S>>;

template <typename T, typename... Ts>
struct index_of;
template <typename T, typename... Ts>
struct index_of<T, T, Ts...> : std::integral_constant<std::size_t, 0> {};
template <typename T, typename U, typename... Ts>
struct index_of<T, U, Ts...> : std::integral_constant<std::size_t, 1 + index_of<T, Ts...>::value> {};
}  // namespace internal

template <typename... Get, typename... Exclude>
class basic_view<get_t<Get...>, exclude_t<Exclude...>> {
    static_assert(sizeof...(Get) > 0u);
    std::tuple<Get *...> m_pools{};
    std::tuple<Exclude *...> m_filter{};
    const sparse_set *m_lead{nullptr};

    template <typename T>
    static constexpr std::size_t pool_index = internal::index_of<std::remove_const_t<T>, internal::storage_value_t<Get>...>::value;

    static bool accepts(const std::tuple<Get *...> &pools, const std::tuple<Exclude *...> &filter, entity e) noexcept {
        return std::apply([e](auto *...p) { return ((p && p->contains(e)) && ...); }, pools) &&
               std::apply([e](auto *...p) { return (!(p && p->contains(e)) && ...); }, filter);
    }
    bool accepts(entity e) const noexcept { return accepts(m_pools, m_filter, e); }
    void pick_lead() noexcept {
        const sparse_set *best = nullptr;
        bool any_missing = false;
        std::apply([&](auto *...p) {
            (([&] {
                 if (!p) { any_missing = true; return; }
                 if (!best || p->size() < best->size()) best = p;
             }()),
             ...);
        }, m_pools);
        m_lead = any_missing ? nullptr : best;
    }
    static auto get_all(const std::tuple<Get *...> &pools, entity e) {
        return std::apply([e](auto *...p) { return std::tuple_cat(tuple_for(p, e)...); }, pools);
    }
    template <typename S>
    static auto tuple_for(S *pool, entity e) {
        if constexpr (internal::storage_has_payload<S>) return std::tuple<internal::storage_elem_t<S> &>(pool->get(e));
        else return std::make_tuple();
    }

public:
    using entity_type = entity;
    using size_type = std::size_t;

    class iterator {
        friend class basic_view;
        // Pool pointers by value: an iterator stays valid after the view object it came from is gone.
        std::tuple<Get *...> m_pools{};
        std::tuple<Exclude *...> m_filter{};
        sparse_set::iterator m_it{}, m_end{};
        void settle() { while (m_it != m_end && !basic_view::accepts(m_pools, m_filter, *m_it)) ++m_it; }
    public:
        using value_type = entity;
        using pointer = const entity *;
        using reference = const entity &;
        using difference_type = std::ptrdiff_t;
        using iterator_category = std::forward_iterator_tag;
        iterator() = default;
        iterator(const basic_view &v, sparse_set::iterator it, sparse_set::iterator end) : m_pools(v.m_pools), m_filter(v.m_filter), m_it(it), m_end(end) { settle(); }
        iterator &operator++() { ++m_it; settle(); return *this; }
        iterator operator++(int) { auto c = *this; ++*this; return c; }
        reference operator*() const noexcept { return *m_it; }
        pointer operator->() const noexcept { return &*m_it; }
        bool operator==(const iterator &o) const noexcept { return m_it == o.m_it; }
        bool operator!=(const iterator &o) const noexcept { return m_it != o.m_it; }
    };

    class iterable {
        // Pool pointers by value: `for (auto [e, c] : registry.view<T>().each())` outlives the temporary view.
        std::tuple<Get *...> m_pools;
        iterator m_first, m_last;
    public:
        class it {
            std::tuple<Get *...> m_pools;
            iterator m_inner;
        public:
            using value_type = decltype(std::tuple_cat(std::make_tuple(entity{}), basic_view::get_all(std::declval<const std::tuple<Get *...> &>(), entity{})));
            using difference_type = std::ptrdiff_t;
            using iterator_category = std::input_iterator_tag;
            using pointer = void;
            using reference = value_type;
            it(const std::tuple<Get *...> &pools, iterator i) : m_pools(pools), m_inner(i) {}
            it &operator++() { ++m_inner; return *this; }
            it operator++(int) { auto c = *this; ++*this; return c; }
            value_type operator*() const { return std::tuple_cat(std::make_tuple(*m_inner), basic_view::get_all(m_pools, *m_inner)); }
            bool operator==(const it &o) const noexcept { return m_inner == o.m_inner; }
            bool operator!=(const it &o) const noexcept { return m_inner != o.m_inner; }
        };
        iterable(const std::tuple<Get *...> &pools, iterator first, iterator last) : m_pools(pools), m_first(first), m_last(last) {}
        it begin() const { return it{m_pools, m_first}; }
        it end() const { return it{m_pools, m_last}; }
    };

    basic_view() = default;
    basic_view(Get &...pools, Exclude &...filter) : m_pools(&pools...), m_filter(&filter...) { pick_lead(); }
    basic_view(std::tuple<Get *...> pools, std::tuple<Exclude *...> filter) : m_pools(pools), m_filter(filter) { pick_lead(); }

    explicit operator bool() const noexcept { return m_lead != nullptr; }
    void refresh() noexcept { pick_lead(); }
    const sparse_set *handle() const noexcept { return m_lead; }

    template <typename T>
    auto *storage() const noexcept { return std::get<pool_index<T>>(m_pools); }
    template <std::size_t I>
    auto *storage() const noexcept { return std::get<I>(m_pools); }

    template <typename T>
    void use() noexcept { m_lead = std::get<pool_index<T>>(m_pools); }

    size_type size_hint() const noexcept { return m_lead ? m_lead->size() : 0u; }
    template <std::size_t N = sizeof...(Get) + sizeof...(Exclude), typename = std::enable_if_t<N == 1u>>
    size_type size() const noexcept { return m_lead ? m_lead->size() : 0u; }
    template <std::size_t N = sizeof...(Get) + sizeof...(Exclude), typename = std::enable_if_t<N == 1u>>
    bool empty() const noexcept { return !m_lead || m_lead->empty(); }

    iterator begin() const noexcept { return m_lead ? iterator{*this, m_lead->begin(), m_lead->end()} : iterator{}; }
    iterator end() const noexcept { return m_lead ? iterator{*this, m_lead->end(), m_lead->end()} : iterator{}; }
    entity front() const noexcept { auto it = begin(); return it != end() ? *it : entity{null}; }
    entity back() const noexcept {
        entity last = null;
        for (auto it = begin(), e = end(); it != e; ++it) last = *it;
        return last;
    }
    iterator find(entity e) const noexcept {
        if (!contains(e)) return end();
        return iterator{*this, m_lead->find(e), m_lead->end()};
    }
    bool contains(entity e) const noexcept { return m_lead && accepts(e); }
    explicit operator bool() noexcept { return m_lead != nullptr; }

    template <typename... T>
    decltype(auto) get(entity e) const {
        if constexpr (sizeof...(T) == 0u) {
            return std::apply([e](auto *...p) { return std::tuple_cat(tuple_for(p, e)...); }, m_pools);
        } else if constexpr (sizeof...(T) == 1u) {
            return (std::get<pool_index<T>>(m_pools)->get(e), ...);
        } else {
            return std::tuple_cat(tuple_for(std::get<pool_index<T>>(m_pools), e)...);
        }
    }
    template <std::size_t I, std::size_t... Is>
    decltype(auto) get(entity e) const {
        if constexpr (sizeof...(Is) == 0u) return std::get<I>(m_pools)->get(e);
        else return std::tuple_cat(tuple_for(std::get<I>(m_pools), e), tuple_for(std::get<Is>(m_pools), e)...);
    }
    decltype(auto) operator[](entity e) const { return get(e); }

    template <typename Func>
    void each(Func func) const {
        for (auto it = begin(), last = end(); it != last; ++it) {
            const entity e = *it;
            if constexpr (is_applicable<Func, decltype(std::tuple_cat(std::make_tuple(e), get(e)))>::value) {
                std::apply(func, std::tuple_cat(std::make_tuple(e), get(e)));
            } else {
                std::apply(func, get(e));
            }
        }
    }
    iterable each() const noexcept { return iterable{m_pools, begin(), end()}; }

    template <typename... OGet, typename... OExclude>
    auto operator|(const basic_view<get_t<OGet...>, exclude_t<OExclude...>> &other) const noexcept {
        return basic_view<get_t<Get..., OGet...>, exclude_t<Exclude..., OExclude...>>{
            std::tuple_cat(m_pools, other.pools()), std::tuple_cat(m_filter, other.filter())};
    }
    const std::tuple<Get *...> &pools() const noexcept { return m_pools; }
    const std::tuple<Exclude *...> &filter() const noexcept { return m_filter; }

private:
    template <typename F, typename Tuple>
    struct is_applicable : std::false_type {};
    template <typename F, typename... A>
    struct is_applicable<F, std::tuple<A...>> : std::is_invocable<F, A...> {};
};

// ----------------------------------------------------------------------------------------- registry
namespace internal {
class context {
    struct holder { virtual ~holder() = default; };
    template <typename T>
    struct holder_of : holder {
        T value;
        template <typename... Args>
        explicit holder_of(std::in_place_t, Args &&...args) : value(make(std::forward<Args>(args)...)) {}
        template <typename... Args>
        static T make(Args &&...args) {
            if constexpr (std::is_aggregate_v<T> && sizeof...(Args) != 0u) return T{std::forward<Args>(args)...};
            else return T(std::forward<Args>(args)...);
        }
    };
    // Insertion order is kept so that the variables die newest first.
    std::vector<std::pair<id_type, std::unique_ptr<holder>>> m_vars;
    auto locate(id_type id) { return std::find_if(m_vars.begin(), m_vars.end(), [id](auto &p) { return p.first == id; }); }
    auto locate(id_type id) const { return std::find_if(m_vars.begin(), m_vars.end(), [id](auto &p) { return p.first == id; }); }

public:
    context() = default;
    explicit context(const std::allocator<entity> &) {}
    context(const context &) = delete;
    context &operator=(const context &) = delete;
    ~context() { while (!m_vars.empty()) m_vars.pop_back(); }

    template <typename T, typename... Args>
    T &emplace(Args &&...args) { return emplace_as<T>(type_id<T>().hash(), std::forward<Args>(args)...); }
    template <typename T, typename... Args>
    T &emplace_as(id_type id, Args &&...args) {
        auto it = locate(id);
        if (it == m_vars.end()) {
            auto ptr = std::make_unique<holder_of<T>>(std::in_place, std::forward<Args>(args)...);
            auto *raw = ptr.get();
            m_vars.emplace_back(id, std::move(ptr));
            return raw->value;
        }
        return static_cast<holder_of<T> *>(it->second.get())->value;
    }
    template <typename T>
    T &insert_or_assign(T &&value) {
        using U = std::remove_cv_t<std::remove_reference_t<T>>;
        const auto id = type_id<U>().hash();
        auto it = locate(id);
        if (it != m_vars.end()) m_vars.erase(it);
        return emplace<U>(std::forward<T>(value));
    }
    template <typename T>
    bool erase(id_type id = type_id<T>().hash()) {
        auto it = locate(id);
        if (it == m_vars.end()) return false;
        auto victim = std::move(it->second);
        m_vars.erase(it);
        return true;
    }
    template <typename T>
    T &get(id_type id = type_id<T>().hash()) {
        auto it = locate(id);
        ENTT_MIN_ASSERT(it != m_vars.end());
        return static_cast<holder_of<std::remove_const_t<T>> *>(it->second.get())->value;
    }
    template <typename T>
    const T &get(id_type id = type_id<T>().hash()) const {
        auto it = locate(id);
        ENTT_MIN_ASSERT(it != m_vars.end());
        return static_cast<const holder_of<std::remove_const_t<T>> *>(it->second.get())->value;
    }
    template <typename T>
    T *find(id_type id = type_id<T>().hash()) {
        auto it = locate(id);
        return it == m_vars.end() || !it->second ? nullptr : &static_cast<holder_of<std::remove_const_t<T>> *>(it->second.get())->value;
    }
    template <typename T>
    const T *find(id_type id = type_id<T>().hash()) const {
        auto it = locate(id);
        return it == m_vars.end() || !it->second ? nullptr : &static_cast<const holder_of<std::remove_const_t<T>> *>(it->second.get())->value;
    }
    template <typename T>
    bool contains(id_type id = type_id<T>().hash()) const {
        auto it = locate(id);
        return it != m_vars.end() && it->second;
    }
};
}  // namespace internal

template <typename T>
using pool_t = storage<T>;

class registry {
    // Entities: ids in use occupy [0, m_alive) of m_entities, destroyed ones follow (recycled LIFO).
    std::vector<entity> m_entities;
    std::vector<std::uint32_t> m_where;  // entity index -> position in m_entities
    std::size_t m_alive{0};

    std::vector<std::pair<id_type, std::unique_ptr<sparse_set>>> m_pools;  // creation order
    std::unordered_map<id_type, std::size_t> m_pool_lookup;
    internal::context m_vars;

    template <typename T>
    pool_t<T> &assure() {
        static_assert(!std::is_const_v<T>);
        const auto id = type_id<T>().hash();
        auto it = m_pool_lookup.find(id);
        if (it == m_pool_lookup.end()) {
            auto pool = std::make_unique<pool_t<T>>();
            pool->bind(*this);
            m_pool_lookup.emplace(id, m_pools.size());
            m_pools.emplace_back(id, std::move(pool));
            return static_cast<pool_t<T> &>(*m_pools.back().second);
        }
        return static_cast<pool_t<T> &>(*m_pools[it->second].second);
    }
    template <typename T>
    const pool_t<T> *find_pool() const {
        auto it = m_pool_lookup.find(type_id<T>().hash());
        return it == m_pool_lookup.end() ? nullptr : static_cast<const pool_t<T> *>(m_pools[it->second].second.get());
    }

public:
    using entity_type = entity;
    using size_type = std::size_t;
    using version_type = std::uint32_t;
    using context = internal::context;
    using allocator_type = std::allocator<entity>;
    template <typename T>
    using storage_for_type = storage_for_t<T>;

    registry() = default;
    registry(const registry &) = delete;
    registry &operator=(const registry &) = delete;
    registry(registry &&) = delete;
    // m_vars is the last member: context variables (which hold signal connections into the pools) die first.
    ~registry() = default;

    // ---- entities
    entity create() {
        if (m_alive < m_entities.size()) return m_entities[m_alive++];
        const auto idx = static_cast<std::uint32_t>(m_entities.size());
        ENTT_MIN_ASSERT(idx < internal::entity_mask);
        const entity e{idx};
        m_entities.push_back(e);
        m_where.push_back(idx);
        ++m_alive;
        return e;
    }
    entity create(entity) { return create(); }
    template <typename It>
    void create(It first, It last) { for (; first != last; ++first) *first = create(); }

    bool valid(entity e) const noexcept {
        const auto idx = to_entity(e);
        return idx < m_where.size() && m_where[idx] < m_alive && m_entities[m_where[idx]] == e;
    }
    version_type current(entity e) const noexcept {
        const auto idx = to_entity(e);
        return idx < m_where.size() ? to_version(m_entities[m_where[idx]]) : internal::version_mask;
    }
    version_type destroy(entity e) {
        ENTT_MIN_ASSERT(valid(e));
        for (auto pos = m_pools.size(); pos; --pos) m_pools[pos - 1u].second->remove(e);
        // Move to the head of the free region with the version bumped.
        const auto idx = to_entity(e);
        const auto pos = m_where[idx];
        const auto last = m_alive - 1u;
        std::uint32_t ver = (to_version(e) + 1u) & internal::version_mask;
        if (ver == internal::version_mask) ver = 0;
        const entity bumped{idx | (ver << internal::version_shift)};
        m_entities[pos] = m_entities[last];
        m_where[to_entity(m_entities[pos])] = pos;
        m_entities[last] = bumped;
        m_where[idx] = static_cast<std::uint32_t>(last);
        --m_alive;
        return ver;
    }
    template <typename It>
    void destroy(It first, It last) {
        std::vector<entity> victims(first, last);
        for (auto e : victims) destroy(e);
    }

    // ---- pools
    template <typename T>
    auto &storage() { return assure<std::remove_const_t<T>>(); }
    template <typename T>
    const auto *storage() const { return find_pool<std::remove_const_t<T>>(); }

    template <typename T, typename... Args>
    decltype(auto) emplace(entity e, Args &&...args) {
        ENTT_MIN_ASSERT(valid(e));
        return assure<T>().emplace(e, std::forward<Args>(args)...);
    }
    template <typename T, typename It>
    void insert(It first, It last, const T &value = {}) { assure<T>().insert(first, last, value); }

    template <typename T, typename... Args>
    decltype(auto) replace(entity e, Args &&...args) {
        if constexpr (internal::is_empty_component<T>) return assure<T>().patch(e);
        else return assure<T>().patch(e, [&](T &curr) { curr = T{std::forward<Args>(args)...}; });
    }
    template <typename T, typename... Args>
    decltype(auto) emplace_or_replace(entity e, Args &&...args) {
        auto &pool = assure<T>();
        if (pool.contains(e)) return replace<T>(e, std::forward<Args>(args)...);
        return pool.emplace(e, std::forward<Args>(args)...);
    }
    template <typename T, typename... Func>
    decltype(auto) patch(entity e, Func &&...func) { return assure<T>().patch(e, std::forward<Func>(func)...); }

    template <typename T, typename... Other>
    size_type remove(entity e) { return (assure<T>().remove(e) + ... + assure<Other>().remove(e)); }
    template <typename T, typename... Other, typename It>
    size_type remove(It first, It last) {
        size_type n = 0;
        std::vector<entity> victims(first, last);
        for (auto e : victims) n += remove<T, Other...>(e);
        return n;
    }
    template <typename T, typename... Other>
    void erase(entity e) { (assure<T>().erase(e), (assure<Other>().erase(e), ...)); }
    template <typename T, typename... Other, typename It>
    void erase(It first, It last) {
        std::vector<entity> victims(first, last);
        for (auto e : victims) erase<T, Other...>(e);
    }

    template <typename... T>
    void clear() {
        if constexpr (sizeof...(T) == 0u) {
            for (auto pos = m_pools.size(); pos; --pos) m_pools[pos - 1u].second->clear();
            while (m_alive) destroy(m_entities[m_alive - 1u]);
        } else {
            (assure<T>().clear(), ...);
        }
    }
    bool orphan(entity e) const {
        for (auto &p : m_pools) if (p.second->contains(e)) return false;
        return true;
    }

    template <typename... T>
    bool all_of(entity e) const {
        return (([&] { auto *p = find_pool<std::remove_const_t<T>>(); return p && p->contains(e); }()) && ...);
    }
    template <typename... T>
    bool any_of(entity e) const {
        return (([&] { auto *p = find_pool<std::remove_const_t<T>>(); return p && p->contains(e); }()) || ...);
    }

    template <typename... T>
    decltype(auto) get(entity e) {
        if constexpr (sizeof...(T) == 1u) return (assure<std::remove_const_t<T>>().get(e), ...);
        else return std::forward_as_tuple(assure<std::remove_const_t<T>>().get(e)...);
    }
    template <typename... T>
    decltype(auto) get(entity e) const {
        if constexpr (sizeof...(T) == 1u) return (static_cast<const T &>(find_pool<std::remove_const_t<T>>()->get(e)), ...);
        else return std::forward_as_tuple(static_cast<const T &>(find_pool<std::remove_const_t<T>>()->get(e))...);
    }
    template <typename T, typename... Args>
    T &get_or_emplace(entity e, Args &&...args) {
        auto &pool = assure<T>();
        return pool.contains(e) ? pool.get(e) : pool.emplace(e, std::forward<Args>(args)...);
    }
    template <typename... T>
    auto try_get(entity e) {
        if constexpr (sizeof...(T) == 1u) {
            return ([&]() -> T * {
                auto &pool = assure<std::remove_const_t<T>>();
                return pool.contains(e) ? &pool.get(e) : nullptr;
            }(), ...);
        } else {
            return std::make_tuple(try_get<T>(e)...);
        }
    }
    template <typename... T>
    auto try_get(entity e) const {
        if constexpr (sizeof...(T) == 1u) {
            return ([&]() -> const T * {
                auto *pool = find_pool<std::remove_const_t<T>>();
                return pool && pool->contains(e) ? &pool->get(e) : nullptr;
            }(), ...);
        } else {
            return std::make_tuple(try_get<T>(e)...);
        }
    }

    // ---- views
    template <typename T, typename... Other, typename... Exclude>
    basic_view<get_t<storage_for_t<T>, storage_for_t<Other>...>, exclude_t<storage_for_t<Exclude>...>>
    view(exclude_t<Exclude...> = exclude_t<>{}) {
        return {assure<std::remove_const_t<T>>(), assure<std::remove_const_t<Other>>()..., assure<std::remove_const_t<Exclude>>()...};
    }
    template <typename T, typename... Other, typename... Exclude>
    basic_view<get_t<const pool_t<std::remove_const_t<T>>, const pool_t<std::remove_const_t<Other>>...>,
               exclude_t<const pool_t<std::remove_const_t<Exclude>>...>>
    view(exclude_t<Exclude...> = exclude_t<>{}) const {
        return {std::make_tuple(find_pool<std::remove_const_t<T>>(), find_pool<std::remove_const_t<Other>>()...),
                std::make_tuple(find_pool<std::remove_const_t<Exclude>>()...)};
    }

    // ---- signals
    template <typename T>
    auto on_construct() { return assure<T>().on_construct(); }
    template <typename T>
    auto on_update() { return assure<T>().on_update(); }
    template <typename T>
    auto on_destroy() { return assure<T>().on_destroy(); }

    // ---- context variables
    context &ctx() noexcept { return m_vars; }
    const context &ctx() const noexcept { return m_vars; }

    template <typename Func>
    void each_entity(Func func) const { for (std::size_t i = m_alive; i; --i) func(m_entities[i - 1u]); }
    size_type alive() const noexcept { return m_alive; }
};

template <typename = entity>
using basic_registry = registry;

template <typename... T>
using view = basic_view<get_t<storage_for_t<T>...>, exclude_t<>>;

// --------------------------------------------------------------------------------------------- meta
// The reference uses entt::meta only to remap entity-valued members when registry contents are shipped
// between registries (replication / networking — not on the simulation path). This stub keeps those
// headers compiling; nothing is ever registered, so every resolve() yields an invalid type.
struct as_ref_t {};
struct meta_type;
struct meta_any;
struct meta_handle {
    meta_handle() = default;
    template <typename T> meta_handle(T &) {}
};
struct meta_sequence_container {
    struct iterator {
        using value_type = meta_any; using difference_type = std::ptrdiff_t; using pointer = void; using reference = void;
        using iterator_category = std::input_iterator_tag;
        inline meta_any operator*() const;
        iterator &operator++() { return *this; }
        bool operator==(const iterator &) const { return true; }
        bool operator!=(const iterator &) const { return false; }
    };
    inline meta_type value_type() const;
    std::size_t size() const { return 0; }
    iterator begin() { return {}; }
    iterator end() { return {}; }
    bool clear() { return true; }
    inline meta_any operator[](std::size_t);
    explicit operator bool() const { return false; }
};
struct meta_associative_container {
    struct iterator {
        using value_type = std::pair<meta_any, meta_any>; using difference_type = std::ptrdiff_t; using pointer = void; using reference = void;
        using iterator_category = std::input_iterator_tag;
        inline std::pair<meta_any, meta_any> operator*() const;
        iterator &operator++() { return *this; }
        bool operator==(const iterator &) const { return true; }
        bool operator!=(const iterator &) const { return false; }
    };
    inline meta_type key_type() const;
    inline meta_type mapped_type() const;
    inline meta_type value_type() const;
    std::size_t size() const { return 0; }
    iterator begin() { return {}; }
    iterator end() { return {}; }
    explicit operator bool() const { return false; }
};
struct meta_data;
struct meta_type {
    struct range {
        struct iterator {
            using value_type = std::pair<id_type, meta_data>; using difference_type = std::ptrdiff_t; using pointer = void; using reference = void;
            using iterator_category = std::input_iterator_tag;
            inline std::pair<id_type, meta_data> operator*() const;
            iterator &operator++() { return *this; }
            bool operator==(const iterator &) const { return true; }
            bool operator!=(const iterator &) const { return false; }
        };
        iterator begin() const { return {}; }
        iterator end() const { return {}; }
    };
    explicit operator bool() const noexcept { return false; }
    bool operator==(const meta_type &) const noexcept { return false; }
    bool operator!=(const meta_type &) const noexcept { return true; }
    id_type id() const noexcept { return 0; }
    bool is_sequence_container() const noexcept { return false; }
    bool is_associative_container() const noexcept { return false; }
    range data() const noexcept { return {}; }
};
struct meta_any {
    meta_any() = default;
    template <typename T, typename = std::enable_if_t<!std::is_same_v<std::decay_t<T>, meta_any>>> meta_any(T &&) {}
    meta_type type() const noexcept { return {}; }
    template <typename T> T cast() const { return T{}; }
    template <typename T> T *try_cast() { return nullptr; }
    template <typename T> bool assign(T &&) { return false; }
    meta_sequence_container as_sequence_container() { return {}; }
    meta_associative_container as_associative_container() { return {}; }
    explicit operator bool() const noexcept { return false; }
};
struct meta_data {
    meta_type type() const noexcept { return {}; }
    meta_any get(meta_handle) const { return {}; }
    template <typename T> bool set(meta_handle, T &&) const { return false; }
};
inline meta_any meta_sequence_container::iterator::operator*() const { return {}; }
inline meta_any meta_sequence_container::operator[](std::size_t) { return {}; }
inline meta_type meta_sequence_container::value_type() const { return {}; }
inline std::pair<meta_any, meta_any> meta_associative_container::iterator::operator*() const { return {}; }
inline meta_type meta_associative_container::key_type() const { return {}; }
inline meta_type meta_associative_container::mapped_type() const { return {}; }
inline meta_type meta_associative_container::value_type() const { return {}; }
inline std::pair<id_type, meta_data> meta_type::range::iterator::operator*() const { return {}; }
template <typename T> meta_type resolve() noexcept { return {}; }
inline meta_type resolve(id_type) noexcept { return {}; }
inline meta_type resolve(const type_info &) noexcept { return {}; }
template <typename T>
struct meta_factory {
    template <auto, typename...> meta_factory &data(id_type) { return *this; }
    meta_factory &type(id_type) { return *this; }
};
template <typename...> struct meta_sequence_container_traits;
template <typename...> struct meta_associative_container_traits;

}  // namespace entt

#endif  // ENTT_MIN_HPP
