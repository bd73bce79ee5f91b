// host_scene.h — host-side (C++) scene copy, BVH builder and HBM layout writer.
// Compiled with -ffp-contract=off: every float op here is a separately rounded
// binary32 op, like the reference's baseline x86-64 build.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_whitted.h"
#include "rt_layout.h"

namespace rt {

struct HostMesh {
    uint32_t vertex_base; // index of this mesh's first vertex in the global arrays
    uint32_t prim_base;   // prim id of this mesh's first triangle
    uint32_t n_vertices, n_triangles;
};

// Deep copy of an rt_scene_desc in the flattened form the kernels consume.
struct HostScene {
    std::vector<HostMesh> meshes;
    std::vector<float> positions;   // global vertices, xyz
    std::vector<float> normals;     // xyz
    std::vector<float> texcoords;   // uv
    std::vector<uint32_t> tri_vidx; // 3 global vertex indices per triangle, prim-id order
    std::vector<rt_sphere_desc> spheres;
    std::vector<MaterialRec> materials; // one per mesh, then one per sphere
    std::vector<TextureRec> textures;
    std::vector<float> texels;
    std::vector<LightRec> lights;
    uint32_t n_triangles = 0;
    float extent = 0.0f;         // largest |coordinate| of any vertex / sphere bound
    bool any_non_opaque = false; // some material.transparency != 1 (src/render.cpp:122)

    // returns "" on success, otherwise what is wrong with the descriptor
    std::string load(const rt_scene_desc& d);
    static std::string convert_lights(const rt_light_desc* lights, uint32_t n, std::vector<LightRec>& out);
    uint32_t n_prims() const { return n_triangles + (uint32_t)spheres.size(); }
};

// Reference-shaped node (src/bounding_volume_hierarchy.h:31-41), kept for
// rt_bvh_export / numLevels / numLeaves parity.
struct RefNode {
    float lo[3], hi[3];
    int32_t leaf, depth, beg, end, left, right;
};

struct HostBvh {
    std::vector<RefNode> nodes;      // post-order, root = last
    std::vector<int32_t> prim_order; // permuted position → prim id
    std::vector<NodePair> pairs;     // one per internal node
    bool bounds_finite = true;       // every bound of every pair is a finite float
    std::vector<PrimRecord> records;      // the reference's visit order (leaves right-first) — BVH mode
    std::vector<PrimRecord> records_flat; // the reference's array order — brute-force mode
    // culling groups (rt_layout.h GroupRecord): slot i of the grouped leaf array holds
    // records[grouped_src[i]] (0xffffffff = padding)
    std::vector<GroupRecord> groups;
    std::vector<uint32_t> grouped_src;
    // the wide tree over the triangles (rt_layout.h WideNode; host_wide.cpp); wide_ok false: not built
    std::vector<WideNode> wide_nodes;
    std::vector<uint32_t> wide_slot_src; // slot -> index into `records` (0xffffffff = padding)
    std::vector<uint32_t> wide_index;    // per slot: index into `records` (0 for padding)
    std::vector<LeafBox> wide_leaf_box;  // per reference leaf, in visit order
    std::vector<uint32_t> ref_leaf_of;   // per entry of `records`: its reference leaf
    bool wide_ok = false;
    uint32_t wide_depth = 0;
    void build_wide(); // host_wide.cpp; needs records
    std::string check_wide(uint64_t stats[8]) const; // structural self-check ("" = consistent)
    uint32_t root_ref = 0, root_cnt = 0; // root_cnt: count word (rt_layout.h)
    int num_levels = 0, num_leaves = 0;
    uint32_t max_leaf = 0;

    // ≙ BoundingVolumeHierarchy::BoundingVolumeHierarchy (bvh.cpp:149-194)
    std::string build(const HostScene& scene, bool sah_binning);
};

} // namespace rt
