// host_scene.cpp — scene flattening, BVH construction and HBM record writer.
//
// Replaces (host side, C++ as in the reference):
//   BoundingVolumeHierarchy::BoundingVolumeHierarchy  src/bounding_volume_hierarchy.cpp:149-194
//   createBVH / splitStandard / splitSAHBinning        :130-147 / :74-78 / :97-128
//   getBoundingBox / boundingBoxSurfaceArea            :22-68
// The split sequence (std::nth_element / std::sort calls, comparator results)
// is the reference's, so the primitive permutation — which decides hit indices
// on ties — is reproduced; node boxes are built bottom-up (min/max are exact,
// so unions equal the reference's per-node rescans).
#include "host_scene.h"

#include <algorithm>
#include <limits>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace rt {
namespace {

struct F3 {
    float x, y, z;
};
inline F3 ld(const float* p) { return F3 { p[0], p[1], p[2] }; }
inline void st(float* p, F3 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }
inline F3 sub(F3 a, F3 b) { return F3 { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline F3 add(F3 a, F3 b) { return F3 { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline F3 scale(F3 a, float s) { return F3 { a.x * s, a.y * s, a.z * s }; }
// glm dot: (x + y) + z ; glm cross term order (func_geometric.inl:48-79)
inline float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline F3 cross3(F3 a, F3 b) { return F3 { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline F3 unit(F3 v) { return scale(v, 1.0f / std::sqrt(dot3(v, v))); }

struct BuildPrim {
    float c[3];       // Primitive::center (bvh.h:22)
    float lo[3], hi[3];
    int32_t id;
};

struct Builder {
    std::vector<BuildPrim> prims;
    HostBvh* out;
    bool sah;

    static constexpr uint32_t kMaxDepth = 16; // MAX_DEPTH, bvh.h:43
    static constexpr uint32_t kBins = 8;      // NUM_OF_BINS, bvh.h:44

    void bounds(size_t beg, size_t end, float lo[3], float hi[3]) const
    {
        for (int a = 0; a < 3; a++)
            lo[a] = prims[beg].lo[a], hi[a] = prims[beg].hi[a];
        for (size_t i = beg + 1; i < end; i++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::min(lo[a], prims[i].lo[a]);
                hi[a] = std::max(hi[a], prims[i].hi[a]);
            }
    }

    // areaLeft*(split-beg) + areaRight*(end-split), bvh.cpp:80-88
    float split_cost(size_t beg, size_t end, size_t split) const
    {
        float area[2];
        const size_t r[3] = { beg, split, end };
        for (int s = 0; s < 2; s++) {
            float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
            if (r[s] != r[s + 1])
                bounds(r[s], r[s + 1], lo, hi);
            const float lx = hi[0] - lo[0], ly = hi[1] - lo[1], lz = hi[2] - lo[2];
            area[s] = 2 * (lx * ly + ly * lz + lz * lx);
        }
        return area[0] * (split - beg) + area[1] * (end - split);
    }

    size_t split_median(size_t beg, size_t end, uint32_t depth)
    {
        const size_t mid = beg + (end - beg) / 2;
        const int axis = (int)(depth % 3);
        std::nth_element(prims.begin() + beg, prims.begin() + mid, prims.begin() + end,
            [axis](const BuildPrim& a, const BuildPrim& b) { return a.c[axis] < b.c[axis]; });
        return mid;
    }

    size_t split_sah(size_t beg, size_t end)
    {
        const size_t skip = std::max<size_t>(1, (end - beg) / kBins);
        size_t best_split = beg, best_axis = 0;
        float best_cost = FLT_MAX;
        for (int axis = 0; axis < 3; axis++) {
            std::sort(prims.begin() + beg, prims.begin() + end,
                [axis](const BuildPrim& a, const BuildPrim& b) { return a.c[axis] < b.c[axis]; });
            for (size_t split = beg + skip; split < end; split += skip) {
                const float cost = split_cost(beg, end, split);
                if (cost < best_cost)
                    best_cost = cost, best_split = split, best_axis = (size_t)axis;
            }
        }
        const int axis = (int)best_axis;
        std::sort(prims.begin() + beg, prims.begin() + end,
            [axis](const BuildPrim& a, const BuildPrim& b) { return a.c[axis] < b.c[axis]; });
        return best_split;
    }

    // returns the index of the created node in out->nodes (post-order)
    int32_t create(size_t beg, size_t end, uint32_t depth)
    {
        out->num_levels = std::max(out->num_levels, (int)depth + 1);
        RefNode node;
        node.depth = (int32_t)depth, node.beg = (int32_t)beg, node.end = (int32_t)end;
        if (depth + 1 == kMaxDepth || beg + 1 == end) {
            bounds(beg, end, node.lo, node.hi);
            node.leaf = 1, node.left = node.right = -1;
            out->num_leaves++;
            out->max_leaf = std::max(out->max_leaf, (uint32_t)(end - beg));
            out->nodes.push_back(node);
            return (int32_t)out->nodes.size() - 1;
        }
        const size_t mid = sah ? split_sah(beg, end) : split_median(beg, end, depth);
        const int32_t left = create(beg, mid, depth + 1);
        const int32_t right = create(mid, end, depth + 1);
        const RefNode &l = out->nodes[left], &r = out->nodes[right];
        for (int a = 0; a < 3; a++) {
            node.lo[a] = std::min(l.lo[a], r.lo[a]);
            node.hi[a] = std::max(l.hi[a], r.hi[a]);
        }
        node.leaf = 0, node.left = left, node.right = right;
        out->nodes.push_back(node);
        return (int32_t)out->nodes.size() - 1;
    }
};

void fill_material(MaterialRec& m, const rt_material& s)
{
    memset(&m, 0, sizeof(m));
    memcpy(m.kd, s.kd, 12);
    memcpy(m.ks, s.ks, 12);
    m.shininess = s.shininess;
    m.transparency = s.transparency;
    m.texture = s.texture;
}

} // namespace

std::string HostScene::convert_lights(const rt_light_desc* lights, uint32_t n, std::vector<LightRec>& out)
{
    out.clear();
    for (uint32_t i = 0; i < n; i++) {
        if (lights[i].kind < RT_LIGHT_POINT || lights[i].kind > RT_LIGHT_PARALLELOGRAM)
            return "light " + std::to_string(i) + ": unknown kind";
        LightRec L;
        memset(&L, 0, sizeof(L));
        L.kind = lights[i].kind;
        memcpy(L.v, lights[i].v, sizeof(L.v));
        out.push_back(L);
    }
    return "";
}

std::string HostScene::load(const rt_scene_desc& d)
{
    if ((d.n_meshes && !d.meshes) || (d.n_spheres && !d.spheres) || (d.n_lights && !d.lights)
        || (d.n_textures && !d.textures))
        return "null array with non-zero count";
    for (uint32_t t = 0; t < d.n_textures; t++) {
        const rt_texture_desc& td = d.textures[t];
        if (!td.pixels || td.width <= 0 || td.height <= 0)
            return "texture " + std::to_string(t) + ": empty";
        TextureRec rec { (uint32_t)(texels.size() / 3), td.width, td.height, 0 };
        textures.push_back(rec);
        texels.insert(texels.end(), td.pixels, td.pixels + (size_t)td.width * td.height * 3);
    }
    for (uint32_t m = 0; m < d.n_meshes; m++) {
        const rt_mesh_desc& md = d.meshes[m];
        if (!md.positions || (md.n_triangles && !md.triangles))
            return "mesh " + std::to_string(m) + ": null positions/triangles";
        if (md.material.texture >= (int32_t)d.n_textures)
            return "mesh " + std::to_string(m) + ": texture index out of range";
        HostMesh hm { (uint32_t)(positions.size() / 3), n_triangles, md.n_vertices, md.n_triangles };
        positions.insert(positions.end(), md.positions, md.positions + 3 * (size_t)md.n_vertices);
        if (md.normals)
            normals.insert(normals.end(), md.normals, md.normals + 3 * (size_t)md.n_vertices);
        else
            normals.resize(normals.size() + 3 * (size_t)md.n_vertices, 0.0f);
        if (md.texcoords)
            texcoords.insert(texcoords.end(), md.texcoords, md.texcoords + 2 * (size_t)md.n_vertices);
        else
            texcoords.resize(texcoords.size() + 2 * (size_t)md.n_vertices, 0.0f);
        for (size_t i = 0; i < 3 * (size_t)md.n_triangles; i++) {
            if (md.triangles[i] >= md.n_vertices)
                return "mesh " + std::to_string(m) + ": vertex index out of range";
            tri_vidx.push_back(hm.vertex_base + md.triangles[i]);
        }
        n_triangles += md.n_triangles;
        meshes.push_back(hm);
        MaterialRec mr;
        fill_material(mr, md.material);
        if (mr.transparency != 1.0f)
            any_non_opaque = true;
        materials.push_back(mr);
    }
    for (uint32_t s = 0; s < d.n_spheres; s++) {
        spheres.push_back(d.spheres[s]);
        MaterialRec mr;
        fill_material(mr, d.spheres[s].material);
        mr.texture = -1;
        if (mr.transparency != 1.0f)
            any_non_opaque = true;
        materials.push_back(mr);
    }
    for (float v : positions)
        extent = std::max(extent, std::fabs(v));
    for (const rt_sphere_desc& sp : spheres)
        for (int a = 0; a < 3; a++)
            extent = std::max(extent, std::fabs(sp.center[a]) + std::fabs(sp.radius));
    return convert_lights(d.lights, d.n_lights, lights);
}

std::string HostBvh::build(const HostScene& scene, bool sah_binning)
{
    const uint32_t n = scene.n_prims();
    if (n == 0)
        return "empty scene (the reference throws bad_optional_access, bvh.cpp:132)";
    if (n >= (1u << 25))
        return "more than 2^25 primitives: a leaf count no longer fits the count word (rt_layout.h)";
    Builder b;
    b.out = this;
    b.sah = sah_binning;
    b.prims.resize(n);
    // bvh.cpp:158-172 — triangles mesh by mesh, then spheres
    for (uint32_t t = 0; t < scene.n_triangles; t++) {
        BuildPrim& p = b.prims[t];
        const F3 a = ld(&scene.positions[3 * (size_t)scene.tri_vidx[3 * (size_t)t]]);
        const F3 c1 = ld(&scene.positions[3 * (size_t)scene.tri_vidx[3 * (size_t)t + 1]]);
        const F3 c2 = ld(&scene.positions[3 * (size_t)scene.tri_vidx[3 * (size_t)t + 2]]);
        const F3 s = add(add(a, c1), c2); // triangleCenter: (a + b + c) / 3.f
        p.c[0] = s.x / 3.f, p.c[1] = s.y / 3.f, p.c[2] = s.z / 3.f;
        p.lo[0] = std::min({ a.x, c1.x, c2.x }), p.hi[0] = std::max({ a.x, c1.x, c2.x });
        p.lo[1] = std::min({ a.y, c1.y, c2.y }), p.hi[1] = std::max({ a.y, c1.y, c2.y });
        p.lo[2] = std::min({ a.z, c1.z, c2.z }), p.hi[2] = std::max({ a.z, c1.z, c2.z });
        p.id = (int32_t)t;
    }
    for (uint32_t s = 0; s < scene.spheres.size(); s++) {
        BuildPrim& p = b.prims[scene.n_triangles + s];
        const rt_sphere_desc& sp = scene.spheres[s];
        for (int a = 0; a < 3; a++) {
            p.c[a] = sp.center[a];
            p.lo[a] = sp.center[a] - sp.radius;
            p.hi[a] = sp.center[a] + sp.radius;
        }
        p.id = (int32_t)(scene.n_triangles + s);
    }
    nodes.clear();
    nodes.reserve(2 * (size_t)n);
    num_levels = num_leaves = 0;
    max_leaf = 0;
    const int32_t root = b.create(0, n, 0);

    // permutation (reference array order) + fetch records in that order
    prim_order.resize(n);
    std::vector<PrimRecord> flat(n);
    std::vector<PrimRecord>& records = flat;
    // which mesh does a triangle prim id belong to
    std::vector<uint32_t> mesh_of(scene.n_triangles);
    for (uint32_t m = 0; m < scene.meshes.size(); m++)
        for (uint32_t t = 0; t < scene.meshes[m].n_triangles; t++)
            mesh_of[scene.meshes[m].prim_base + t] = m;
    for (uint32_t i = 0; i < n; i++) {
        const int32_t id = b.prims[i].id;
        prim_order[i] = id;
        PrimRecord& r = records[i];
        memset(&r, 0, sizeof(r));
        r.prim_id = id;
        if ((uint32_t)id < scene.n_triangles) {
            const F3 v0 = ld(&scene.positions[3 * (size_t)scene.tri_vidx[3 * (size_t)id]]);
            const F3 v1 = ld(&scene.positions[3 * (size_t)scene.tri_vidx[3 * (size_t)id + 1]]);
            const F3 v2 = ld(&scene.positions[3 * (size_t)scene.tri_vidx[3 * (size_t)id + 2]]);
            const F3 nn = unit(cross3(sub(v1, v0), sub(v2, v0))); // trianglePlane (Appendix A.2)
            st(r.n, nn);
            r.D = dot3(v0, nn);
            st(r.e0, cross3(sub(v2, v0), nn)); // pointInTriangle edge vectors (Appendix A.1)
            st(r.e1, cross3(sub(v0, v1), nn));
            st(r.e2, cross3(sub(v1, v2), nn));
            st(r.v0, v0), st(r.v1, v1), st(r.v2, v2);
            r.kind_mat = mesh_of[id];
        } else {
            const uint32_t s = (uint32_t)id - scene.n_triangles;
            memcpy(r.v0, scene.spheres[s].center, 12);
            r.D = scene.spheres[s].radius;
            r.kind_mat = 0x80000000u | ((uint32_t)scene.meshes.size() + s);
        }
    }

    // Device order = the order in which the reference's traversal reaches the
    // primitives when every box is hit: the stack pops the RIGHT child first
    // (bvh.cpp:354-355), leaf primitives in array order (:272-293). With this
    // order "tested later" == "larger index", which is all the order-free
    // closest-hit reduction of rt_device.h needs.
    std::vector<uint32_t> leaf_dev_beg(nodes.size(), 0);
    this->records.clear();
    this->records.reserve(n);
    ref_leaf_of.clear(), wide_leaf_box.clear();
    {
        std::vector<int32_t> stack { root };
        while (!stack.empty()) {
            const int32_t i = stack.back();
            stack.pop_back();
            const RefNode& nd = nodes[i];
            if (nd.leaf) {
                leaf_dev_beg[i] = (uint32_t)this->records.size();
                LeafBox lb;
                memset(&lb, 0, sizeof(lb));
                memcpy(lb.lo, nd.lo, 12), memcpy(lb.hi, nd.hi, 12);
                for (int32_t p = nd.beg; p < nd.end; p++) {
                    this->records.push_back(flat[p]);
                    ref_leaf_of.push_back((uint32_t)wide_leaf_box.size());
                }
                wide_leaf_box.push_back(lb);
            } else {
                stack.push_back(nd.left);
                stack.push_back(nd.right);
            }
        }
    }
    records_flat.swap(flat);

    // NodePair per internal node; child reference of a node
    std::vector<int32_t> pair_of(nodes.size(), -1);
    pairs.clear();
    for (size_t i = 0; i < nodes.size(); i++)
        if (!nodes[i].leaf) {
            pair_of[i] = (int32_t)pairs.size();
            pairs.emplace_back();
        }
    // Culling groups of the leaves (rt_layout.h GroupRecord): triangles only, <= 32 per leaf
    groups.clear();
    grouped_src.clear();
    std::vector<uint32_t> leaf_group_beg(nodes.size(), 0xffffffffu);
    if (scene.spheres.empty()) {
        struct Tri {
            float c[3];
            uint32_t rank;
        };
        std::vector<Tri> tri;
        // order a leaf's triangles so that consecutive runs of four are spatially compact:
        // recursive median split of the centroids along their longest axis, cut at a multiple of 4
        auto order = [&](auto&& self, Tri* a, uint32_t cnt) -> void {
            if (cnt <= 4)
                return;
            float lo[3] = { a[0].c[0], a[0].c[1], a[0].c[2] }, hi[3] = { lo[0], lo[1], lo[2] };
            for (uint32_t i = 1; i < cnt; i++)
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(lo[k], a[i].c[k]);
                    hi[k] = std::max(hi[k], a[i].c[k]);
                }
            int ax = 0;
            for (int k = 1; k < 3; k++)
                if (hi[k] - lo[k] > hi[ax] - lo[ax])
                    ax = k;
            const uint32_t mid = ((cnt / 2 + 3) / 4) * 4; // 4 <= mid < cnt for cnt >= 5
            std::nth_element(a, a + mid, a + cnt, [ax](const Tri& x, const Tri& y) {
                return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.rank < y.rank);
            });
            self(self, a, mid);
            self(self, a + mid, cnt - mid);
        };
        for (size_t i = 0; i < nodes.size(); i++) {
            const RefNode& nd = nodes[i];
            const uint32_t cnt = nd.leaf ? (uint32_t)(nd.end - nd.beg) : 0u;
            if (!nd.leaf || cnt == 0 || cnt > kGroupLeafMax)
                continue;
            const uint32_t ref = leaf_dev_beg[i];
            tri.resize(cnt);
            for (uint32_t k = 0; k < cnt; k++) {
                const PrimRecord& r = this->records[ref + k];
                for (int a = 0; a < 3; a++)
                    tri[k].c[a] = (r.v0[a] + r.v1[a] + r.v2[a]) * (1.0f / 3.0f);
                tri[k].rank = k;
            }
            order(order, tri.data(), cnt);
            if (groups.size() + (cnt + 3) / 4 >= (1u << 25))
                break; // count word holds 25 bits of group index: the rest stays ungrouped
            leaf_group_beg[i] = (uint32_t)groups.size();
            for (uint32_t g = 0; g * 4 < cnt; g++) {
                GroupRecord gr;
                memset(&gr, 0, sizeof(gr));
                for (int a = 0; a < 3; a++)
                    gr.lo[a] = std::numeric_limits<float>::infinity(), gr.hi[a] = -std::numeric_limits<float>::infinity();
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t k = g * 4 + j;
                    if (k >= cnt) {
                        gr.ranks |= 31u << (5 * j);
                        grouped_src.push_back(0xffffffffu);
                        continue;
                    }
                    const PrimRecord& r = this->records[ref + tri[k].rank];
                    for (int a = 0; a < 3; a++) {
                        gr.lo[a] = std::min(gr.lo[a], std::min(r.v0[a], std::min(r.v1[a], r.v2[a])));
                        gr.hi[a] = std::max(gr.hi[a], std::max(r.v0[a], std::max(r.v1[a], r.v2[a])));
                    }
                    gr.ranks |= tri[k].rank << (5 * j);
                    grouped_src.push_back(ref + tri[k].rank);
                }
                groups.push_back(gr);
            }
        }
    }
    if (groups.empty()) { // keep the device arrays non-empty
        groups.emplace_back();
        grouped_src.assign(4, 0xffffffffu);
    }
    auto child_ref = [&](int32_t node, uint32_t& ref, uint32_t& cnt) {
        const RefNode& c = nodes[node];
        if (c.leaf) {
            ref = leaf_dev_beg[node];
            const uint32_t n_leaf = (uint32_t)(c.end - c.beg);
            cnt = leaf_group_beg[node] != 0xffffffffu ? (n_leaf | (leaf_group_beg[node] << 6)) : cw_make_big(n_leaf);
        } else
            ref = (uint32_t)pair_of[node], cnt = 0;
    };
    for (size_t i = 0; i < nodes.size(); i++) {
        if (nodes[i].leaf)
            continue;
        NodePair& p = pairs[pair_of[i]];
        const RefNode &l = nodes[nodes[i].left], &r = nodes[nodes[i].right];
        memcpy(p.l_lo, l.lo, 12), memcpy(p.l_hi, l.hi, 12);
        memcpy(p.r_lo, r.lo, 12), memcpy(p.r_hi, r.hi, 12);
        child_ref(nodes[i].left, p.l_ref, p.l_cnt);
        child_ref(nodes[i].right, p.r_ref, p.r_cnt);
    }
    child_ref(root, root_ref, root_cnt);
    if (root_cnt != 0u) // a root that is a leaf takes the ungrouped path (one range per ray)
        root_cnt = cw_make_big((uint32_t)(nodes[root].end - nodes[root].beg));
    bounds_finite = true;
    for (const NodePair& p : pairs)
        for (int a = 0; a < 3; a++)
            bounds_finite = bounds_finite && std::isfinite(p.l_lo[a]) && std::isfinite(p.l_hi[a])
                && std::isfinite(p.r_lo[a]) && std::isfinite(p.r_hi[a]);
    if (pairs.empty())
        pairs.emplace_back(); // keep the device array non-empty
    // MAX_DEPTH = 16 caps a tree at 2^15 - 1 internal nodes; the traversal's packed deque
    // entry (rt_device.h pack_entry) relies on it
    if (pairs.size() > (1u << 15))
        return "internal error: more than 32768 internal nodes";
    // the wide tree: triangles only, and a root that is an internal node (a leaf root is never box-tested,
    // bvh.cpp:313 — such scenes are tiny and stay on the exact-topology traversal)
    wide_ok = false;
    if (scene.spheres.empty() && !nodes[root].leaf && bounds_finite)
        build_wide();
    return "";
}

} // namespace rt
