// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header).
// Incremental bounding-volume tree used by the reference broadphase: a Box2D-style dynamic AABB tree
// with fattened leaves, surface-area-heuristic sibling selection and height-balancing rotations.
// Restates the behaviour of
//   /root/reference/src/edyn/collision/dynamic_tree.cpp:41-78   create / move (fat margin 0.1, re-insert only
//                                                                 when the tight box leaves the fat box)
//   /root/reference/src/edyn/collision/dynamic_tree.cpp:80-123  best sibling (SAH descent)
//   /root/reference/src/edyn/collision/dynamic_tree.cpp:125-339 insert / remove / refit / balance
//   /root/reference/include/edyn/collision/query_tree.hpp:9-42  stack DFS query
// The pair set produced by the broadphase does not depend on the tree shape (the final predicate
// uses the true AABBs), so this structure matters for CPU-baseline timing, not for parity.
#pragma once
#include <vector>
#include "omath.hpp"

namespace orc {

class DynTree {
public:
    static constexpr uint32_t NIL = 0xFFFFFFFFu;
    static constexpr float kFat = 0.1f;   // dynamic_tree.hpp:24 (aabb_inset = -0.1)

    uint32_t create(const aabb &box, uint32_t payload) {
        uint32_t id = alloc();
        n_[id].payload = payload;
        n_[id].box = box.inset({-kFat, -kFat, -kFat});
        link_leaf(id);
        return id;
    }
    void destroy(uint32_t id) { unlink_leaf(id); release(id); }
    bool move(uint32_t id, const aabb &box) {
        if (n_[id].box.contains(box)) return false;
        unlink_leaf(id);
        n_[id].box = box.inset({-kFat, -kFat, -kFat});
        link_leaf(id);
        return true;
    }
    uint32_t payload(uint32_t id) const { return n_[id].payload; }
    template <typename F>
    void query(const aabb &q, F &&visit) const {
        stack_.clear();
        stack_.push_back(root_);
        while (!stack_.empty()) {
            uint32_t id = stack_.back();
            stack_.pop_back();
            if (id == NIL) continue;
            const Node &nd = n_[id];
            if (!intersect(nd.box, q)) continue;
            if (nd.is_leaf()) visit(id);
            else { stack_.push_back(nd.c1); stack_.push_back(nd.c2); }
        }
    }
    size_t node_count() const { return n_.size(); }

private:
    struct Node {
        uint32_t parent = NIL;   // doubles as free-list link
        uint32_t c1 = NIL, c2 = NIL;
        uint32_t payload = NIL;
        aabb box{};
        int height = 0;
        bool is_leaf() const { return c1 == NIL; }
    };
    std::vector<Node> n_;
    uint32_t root_ = NIL, free_ = NIL;
    mutable std::vector<uint32_t> stack_;

    uint32_t alloc() {
        uint32_t id;
        if (free_ == NIL) { id = (uint32_t)n_.size(); n_.emplace_back(); }
        else { id = free_; free_ = n_[id].parent; }
        n_[id] = Node{};
        return id;
    }
    void release(uint32_t id) { n_[id].parent = free_; n_[id].height = -1; free_ = id; }

    uint32_t pick_sibling(const aabb &box) const {
        uint32_t id = root_;
        while (!n_[id].is_leaf()) {
            const Node &nd = n_[id];
            float merged = enclosing(nd.box, box).area();
            float cost_here = 2.0f * merged;
            float inherit = 2.0f * (merged - nd.box.area());
            auto descend_cost = [&](uint32_t c) {
                const Node &ch = n_[c];
                float a = enclosing(ch.box, box).area();
                return ch.is_leaf() ? a + inherit : (a - ch.box.area()) + inherit;
            };
            float k1 = descend_cost(nd.c1), k2 = descend_cost(nd.c2);
            if (cost_here < k1 && cost_here < k2) break;
            id = k1 < k2 ? nd.c1 : nd.c2;
        }
        return id;
    }
    void link_leaf(uint32_t leaf) {
        if (root_ == NIL) { root_ = leaf; n_[leaf].parent = NIL; return; }
        const aabb lbox = n_[leaf].box;
        const uint32_t sib = pick_sibling(lbox);
        const uint32_t gp = n_[sib].parent;
        const uint32_t par = alloc();
        n_[par].parent = gp;
        n_[par].box = enclosing(n_[sib].box, lbox);
        n_[par].height = n_[sib].height + 1;
        n_[par].c1 = sib;
        n_[par].c2 = leaf;
        n_[sib].parent = par;
        n_[leaf].parent = par;
        if (gp == NIL) root_ = par;
        else if (n_[gp].c1 == sib) n_[gp].c1 = par;
        else n_[gp].c2 = par;
        fix_upwards(par);
    }
    void unlink_leaf(uint32_t leaf) {
        if (leaf == root_) { root_ = NIL; return; }
        const uint32_t par = n_[leaf].parent;
        const uint32_t sib = n_[par].c1 == leaf ? n_[par].c2 : n_[par].c1;
        const uint32_t gp = n_[par].parent;
        if (gp == NIL) {
            root_ = sib;
            n_[sib].parent = NIL;
            release(par);
        } else {
            if (n_[gp].c1 == par) n_[gp].c1 = sib; else n_[gp].c2 = sib;
            n_[sib].parent = gp;
            release(par);
            fix_upwards(gp);
        }
    }
    void fix_upwards(uint32_t id) {
        while (id != NIL) {
            id = rebalance(id);
            Node &nd = n_[id];
            nd.box = enclosing(n_[nd.c1].box, n_[nd.c2].box);
            nd.height = std::max(n_[nd.c1].height, n_[nd.c2].height) + 1;
            id = nd.parent;
        }
    }
    // Promote child `up` (the taller one) above `a`; `other` is a's remaining child.
    uint32_t rotate_up(uint32_t a, uint32_t up, uint32_t other, bool up_is_c2) {
        const uint32_t g1 = n_[up].c1, g2 = n_[up].c2;
        n_[up].c1 = a;
        n_[up].parent = n_[a].parent;
        n_[a].parent = up;
        const uint32_t pp = n_[up].parent;
        if (pp != NIL) { if (n_[pp].c1 == a) n_[pp].c1 = up; else n_[pp].c2 = up; }
        else root_ = up;
        const bool first_taller = n_[g1].height > n_[g2].height;
        const uint32_t keep = first_taller ? g1 : g2;   // stays under `up`
        const uint32_t give = first_taller ? g2 : g1;   // handed down to `a`
        n_[up].c2 = keep;
        if (up_is_c2) n_[a].c2 = give; else n_[a].c1 = give;
        n_[give].parent = a;
        n_[a].box = enclosing(n_[other].box, n_[give].box);
        n_[up].box = enclosing(n_[a].box, n_[keep].box);
        n_[a].height = std::max(n_[other].height, n_[give].height) + 1;
        n_[up].height = std::max(n_[a].height, n_[keep].height) + 1;
        return up;
    }
    uint32_t rebalance(uint32_t a) {
        if (n_[a].is_leaf() || n_[a].height < 2) return a;
        const uint32_t b = n_[a].c1, c = n_[a].c2;
        const int skew = n_[c].height - n_[b].height;
        if (skew > 1) return rotate_up(a, c, b, true);
        if (skew < -1) return rotate_up(a, b, c, false);
        return a;
    }
};

}  // namespace orc
