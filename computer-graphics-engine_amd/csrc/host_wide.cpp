// host_wide.cpp — the traversal structure of our own: a 4-wide SAH tree over the scene's triangles.
//
// Nothing here has a counterpart in the reference: the reference's tree (median split by depth-cycled axis,
// 16 levels, src/bounding_volume_hierarchy.cpp:74-147) is built by host_scene.cpp because its LEAVES and the
// primitive order inside them decide results. Which leaves a ray visits does not depend on the tree above them
// (rt_wide.h), so the kernels walk this tree instead: binned SAH (16 bins per axis), leaves of up to four
// triangles, binary tree collapsed to four children per node (the child with the largest surface area is
// opened first). Every slot keeps the triangle's index in the reference's visit order.
#include "host_scene.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace rt {
namespace {

struct WTri {
    float lo[3], hi[3], c[3];
    uint32_t src; // index into HostBvh::records
};
struct WBox {
    float lo[3], hi[3];
    void reset()
    {
        for (int a = 0; a < 3; a++)
            lo[a] = FLT_MAX, hi[a] = -FLT_MAX;
    }
    void add(const float* l, const float* h)
    {
        for (int a = 0; a < 3; a++)
            lo[a] = std::min(lo[a], l[a]), hi[a] = std::max(hi[a], h[a]);
    }
    void add_point(const float* p)
    {
        for (int a = 0; a < 3; a++)
            lo[a] = std::min(lo[a], p[a]), hi[a] = std::max(hi[a], p[a]);
    }
    double area() const
    {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x < 0 ? 0.0 : 2.0 * (x * y + y * z + z * x);
    }
};
struct BNode { // binary build node
    WBox box;
    int32_t left = -1, right = -1; // children (internal)
    uint32_t beg = 0, cnt = 0;     // triangle range (leaf)
};

constexpr int kBins = 16;
constexpr uint32_t kLeafMax = 4;

struct WideBuilder {
    std::vector<WTri> tris;
    std::vector<BNode> bn;

    int32_t build(uint32_t beg, uint32_t end)
    {
        BNode node;
        node.box.reset();
        WBox cb;
        cb.reset();
        for (uint32_t i = beg; i < end; i++) {
            node.box.add(tris[i].lo, tris[i].hi);
            cb.add_point(tris[i].c);
        }
        const uint32_t n = end - beg;
        const int32_t me = (int32_t)bn.size();
        bn.push_back(node);
        if (n <= kLeafMax) {
            bn[me].beg = beg, bn[me].cnt = n;
            return me;
        }
        // binned SAH over the three axes
        int best_axis = -1, best_bin = 0;
        double best_cost = DBL_MAX;
        for (int a = 0; a < 3; a++) {
            const float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f))
                continue;
            WBox bb[kBins];
            uint32_t cnt[kBins] = {};
            for (auto& b : bb)
                b.reset();
            const float k = (float)kBins * (1.0f - 1e-6f) / ext;
            for (uint32_t i = beg; i < end; i++) {
                int b = (int)((tris[i].c[a] - cb.lo[a]) * k);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                cnt[b]++;
                bb[b].add(tris[i].lo, tris[i].hi);
            }
            double ra[kBins];
            uint32_t rc[kBins];
            WBox acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                if (cnt[b])
                    acc.add(bb[b].lo, bb[b].hi);
                c += cnt[b];
                ra[b] = acc.area(), rc[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                if (cnt[b])
                    acc.add(bb[b].lo, bb[b].hi);
                c += cnt[b];
                if (c == 0 || rc[b + 1] == 0)
                    continue;
                // leaves hold up to four triangles: cost in units of leaf groups
                const double cost = acc.area() * std::ceil(c / 4.0) + ra[b + 1] * std::ceil(rc[b + 1] / 4.0);
                if (cost < best_cost)
                    best_cost = cost, best_axis = a, best_bin = b;
            }
        }
        uint32_t mid;
        if (best_axis < 0) { // all centroids coincide: split by position in the range
            mid = beg + n / 2;
        } else {
            const int a = best_axis;
            const float k = (float)kBins * (1.0f - 1e-6f) / (cb.hi[a] - cb.lo[a]);
            const float lo = cb.lo[a];
            auto it = std::partition(tris.begin() + beg, tris.begin() + end, [&](const WTri& t) {
                int b = (int)((t.c[a] - lo) * k);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - tris.begin());
            if (mid == beg || mid == end)
                mid = beg + n / 2;
        }
        const int32_t l = build(beg, mid);
        const int32_t r = build(mid, end);
        bn[me].left = l, bn[me].right = r;
        return me;
    }
};

} // namespace

void HostBvh::build_wide()
{
    wide_ok = false;
    wide_nodes.clear(), wide_slot_src.clear(), wide_index.clear();
    const size_t n = records.size();
    if (n == 0 || ref_leaf_of.size() != n)
        return;
    WideBuilder b;
    b.tris.resize(n);
    for (size_t i = 0; i < n; i++) {
        const PrimRecord& r = records[i];
        if (r.kind_mat & 0x80000000u)
            return; // spheres: no wide tree
        WTri& t = b.tris[i];
        for (int a = 0; a < 3; a++) {
            t.lo[a] = std::min(r.v0[a], std::min(r.v1[a], r.v2[a]));
            t.hi[a] = std::max(r.v0[a], std::max(r.v1[a], r.v2[a]));
            t.c[a] = 0.5f * (t.lo[a] + t.hi[a]);
            if (!std::isfinite(t.lo[a]) || !std::isfinite(t.hi[a]))
                return; // NaN / infinite vertices: the exact-topology traversal handles them
        }
        t.src = (uint32_t)i;
    }
    b.bn.reserve(2 * n / 2 + 16);
    const int32_t root = b.build(0, (uint32_t)n);
    if (b.bn[root].left < 0)
        return; // four triangles or fewer: not worth a tree (the root of the reference tree is a leaf too)

    // collapse to four children per node; nodes in depth-first order, leaf groups in the order they are met
    struct Item {
        int32_t bnode;
        uint32_t wnode, depth;
    };
    std::vector<Item> todo;
    wide_nodes.emplace_back();
    todo.push_back({ root, 0u, 1u });
    uint32_t n_groups = 0;
    wide_depth = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        wide_depth = std::max(wide_depth, it.depth);
        int32_t kids[4] = { b.bn[it.bnode].left, b.bn[it.bnode].right, -1, -1 };
        int nk = 2;
        while (nk < 4) { // open the internal child with the largest surface area
            int pick = -1;
            double best = -1.0;
            for (int k = 0; k < nk; k++)
                if (b.bn[kids[k]].left >= 0 && b.bn[kids[k]].box.area() > best)
                    best = b.bn[kids[k]].box.area(), pick = k;
            if (pick < 0)
                break;
            const int32_t open = kids[pick];
            kids[pick] = b.bn[open].left;
            kids[nk++] = b.bn[open].right;
        }
        WideNode wn;
        memset(&wn, 0, sizeof(wn));
        for (int k = 0; k < 4; k++) {
            wn.ref[k] = kWideEmpty;
            wn.lox[k] = wn.loy[k] = wn.loz[k] = FLT_MAX; // an empty slot is masked by its reference word
            wn.hix[k] = wn.hiy[k] = wn.hiz[k] = FLT_MAX;
        }
        for (int k = 0; k < nk; k++) {
            const BNode& c = b.bn[kids[k]];
            wn.lox[k] = c.box.lo[0], wn.loy[k] = c.box.lo[1], wn.loz[k] = c.box.lo[2];
            wn.hix[k] = c.box.hi[0], wn.hiy[k] = c.box.hi[1], wn.hiz[k] = c.box.hi[2];
            if (c.left < 0) { // leaf group
                if (n_groups >= kWideMaxGroups)
                    return;
                wn.ref[k] = wide_make_leaf(n_groups, c.cnt);
                for (uint32_t j = 0; j < 4; j++) {
                    if (j < c.cnt) {
                        const uint32_t src = b.tris[c.beg + j].src;
                        wide_slot_src.push_back(src);
                        wide_index.push_back(src);
                    } else {
                        wide_slot_src.push_back(0xffffffffu);
                        wide_index.push_back(0u);
                    }
                }
                n_groups++;
            } else {
                if (wide_nodes.size() >= kWideMaxNodes)
                    return;
                wn.ref[k] = (uint32_t)wide_nodes.size();
                wide_nodes.emplace_back();
                todo.push_back({ kids[k], wn.ref[k], it.depth + 1 });
            }
        }
        wide_nodes[it.wnode] = wn;
    }
    wide_ok = true;
}

// Structural self-check of the wide tree (host only; rt_bvh_wide_check): every triangle of `records` sits in exactly
// one slot, every child box encloses the bounds of every triangle below it, child words are well-formed.
// stats: [0] nodes, [1] leaf groups, [2] depth, [3] triangles found, [4] empty child slots, [5] internal children
std::string HostBvh::check_wide(uint64_t stats[8]) const
{
    for (int i = 0; i < 8; i++)
        stats[i] = 0;
    if (!wide_ok)
        return "";
    const size_t n = records.size();
    if (wide_index.size() != wide_slot_src.size() || wide_slot_src.size() % 4 != 0 || ref_leaf_of.size() != n)
        return "array sizes";
    std::vector<uint8_t> seen(n, 0);
    struct Item {
        uint32_t node, depth;
    };
    std::vector<Item> todo { { 0u, 1u } };
    std::vector<uint8_t> visited(wide_nodes.size(), 0);
    // bounds of the triangles below a node, bottom-up: second pass over a post-order list
    std::vector<uint32_t> order;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.node >= wide_nodes.size() || visited[it.node])
            return "node index out of range or reached twice";
        visited[it.node] = 1;
        order.push_back(it.node);
        stats[2] = std::max<uint64_t>(stats[2], it.depth);
        for (int k = 0; k < 4; k++) {
            const uint32_t r = wide_nodes[it.node].ref[k];
            if (r == kWideEmpty)
                stats[4]++;
            else if (!wide_is_leaf(r))
                stats[5]++, todo.push_back({ r, it.depth + 1 });
        }
    }
    stats[0] = order.size();
    if (order.size() != wide_nodes.size())
        return "unreachable nodes";
    std::vector<WBox> below(wide_nodes.size());
    for (size_t oi = order.size(); oi-- > 0;) { // children come after their parent in `order`
        const uint32_t ni = order[oi];
        const WideNode& wn = wide_nodes[ni];
        WBox all;
        all.reset();
        for (int k = 0; k < 4; k++) {
            const uint32_t r = wn.ref[k];
            if (r == kWideEmpty)
                continue;
            WBox cb;
            cb.reset();
            if (wide_is_leaf(r)) {
                const uint32_t g = wide_leaf_group(r), cnt = wide_leaf_count(r);
                if ((size_t)g * 4 + 4 > wide_slot_src.size())
                    return "group out of range";
                stats[1]++;
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t src = wide_slot_src[(size_t)g * 4 + j];
                    if (j >= cnt) {
                        if (src != 0xffffffffu)
                            return "padding slot holds a triangle";
                        continue;
                    }
                    if (src >= n || seen[src] || wide_index[(size_t)g * 4 + j] != src)
                        return "triangle missing, repeated or mis-indexed";
                    seen[src] = 1;
                    stats[3]++;
                    const PrimRecord& pr = records[src];
                    cb.add_point(pr.v0), cb.add_point(pr.v1), cb.add_point(pr.v2);
                }
            } else {
                cb = below[r];
            }
            const float lo[3] = { wn.lox[k], wn.loy[k], wn.loz[k] }, hi[3] = { wn.hix[k], wn.hiy[k], wn.hiz[k] };
            for (int a = 0; a < 3; a++)
                if (!(lo[a] <= cb.lo[a]) || !(hi[a] >= cb.hi[a]))
                    return "a child box does not enclose the triangles below it";
            all.add(cb.lo, cb.hi);
        }
        below[ni] = all;
    }
    if (stats[3] != n)
        return "not every triangle is in the tree";
    return "";
}

} // namespace rt
