// rt_layout.h — HBM data layout shared by the host builder and the HIP kernels.
//
// Everything a WAVE fetches during traversal is laid out as one contiguous,
// 32-byte-aligned record per fetch so that a wave-uniform index turns into
// scalar (SMEM) loads: one 64-byte NodePair per internal node visited, one
// 96-byte PrimRecord per leaf primitive tested. Per-LANE gathers (vertex
// attributes, materials, texels) happen once per shaded hit and use plain
// arrays.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h> // float4

namespace rt {

#if defined(__HIPCC__)
#define RT_LAYOUT_FN __host__ __device__ inline
#else
#define RT_LAYOUT_FN inline
#endif

// Children of one internal node (src/bounding_volume_hierarchy.cpp:331-355 reads
// nodes[left].aabb and nodes[right].aabb of the popped parent). Child reference:
// cnt == 0 → internal node, ref = index of ITS NodePair; cnt > 0 → leaf holding
// permuted primitives [ref, ref + cnt).
struct alignas(64) NodePair {
    float l_lo[3], l_hi[3];
    float r_lo[3], r_hi[3];
    uint32_t l_ref, l_cnt;
    uint32_t r_ref, r_cnt;
};
static_assert(sizeof(NodePair) == 64, "NodePair must be one 64-byte record");

enum : uint32_t { PRIM_TRIANGLE = 0u, PRIM_SPHERE = 1u };

// One primitive. Two device arrays hold these records: `prims` in the order in
// which the reference's traversal would reach them if every box were hit
// (leaves right-first, src/bounding_volume_hierarchy.cpp:354-355; array order
// inside a leaf), so that "tested later" == "larger index" (rt_device.h), and
// `prims_flat` in the reference's own array order for the brute-force path. For a triangle the per-primitive part
// of intersectRayWithTriangle is hoisted here: it
// depends on the vertices only, and is computed on the host with the same
// operation order, so per-ray results stay bit-identical:
//   n  = normalize(cross(v1 - v0, v2 - v0)),  D = dot(v0, n)
//   e0 = cross(v2 - v0, n), e1 = cross(v0 - v1, n), e2 = cross(v1 - v2, n)
// For a sphere: v0 = center, D = radius.
struct alignas(32) PrimRecord {
    float n[3];
    float D;
    float e0[3];
    float e1[3];
    float e2[3];
    float v0[3];
    float v1[3];
    float v2[3];
    int32_t prim_id;   // pre-permutation id (triangles mesh-major, then spheres)
    uint32_t kind_mat; // bit 31: PRIM_SPHERE; bits 0..30: material slot (mesh index, or n_meshes + sphere index)
};
static_assert(sizeof(PrimRecord) == 96, "PrimRecord must be 96 bytes");

// What the LEAF phase loads per primitive tested: 48 bytes = three 16-byte words
// (measured on MI355X, tools/ta_bench.hip: a wave's loads cost per byte and per line
// touched, and a 96-byte stride costs twice a 48-byte one, so the record holds only
// what cannot be recomputed cheaply). D and the edge vectors e0..e2 are recomputed per
// test from (v0, v1, v2, n) with the host's operation order, so they are bit-identical
// to the PrimRecord fields. A sphere is tagged by n == (0, 0, 0) — a triangle's normal
// is unit length, or NaN/inf when degenerate, never all zero — with v0 = centre,
// v1[0] = radius.
struct alignas(16) LeafRecord {
    float v0[3], v1[3], v2[3], n[3];
};
static_assert(sizeof(LeafRecord) == 48, "LeafRecord must be 48 bytes");

// Culling groups inside a leaf. The reference's leaves hold up to MAX-depth-limited runs of
// primitives (25 on the bench scene) and its traversal tests every one of them. In the
// pruned policy the leaf phase first tests the ray against the bounds of GROUPS of four
// leaf triangles (conservatively, grown by the pruning slack — rt_device.h "Pruning": a
// candidate's point lies within rho of its triangle, hence of the group's box) and runs the
// exact triangle test only for the groups it may touch. The leaf's triangles are re-ordered
// into spatially compact groups (host_scene.cpp), so each slot carries its RANK in the
// reference's order: keys and reported indices use leaf_ref + rank, which keeps the
// "tested later wins ties" rule intact. 32 bytes per group.
struct alignas(16) GroupRecord {
    float lo[3], hi[3];
    uint32_t ranks; // four 5-bit ranks (slot j at bits 5j..5j+4), 31 = empty slot
    uint32_t pad;
};
static_assert(sizeof(GroupRecord) == 32, "GroupRecord must be 32 bytes");
constexpr uint32_t kGroupLeafMax = 32; // leaves with more primitives (or spheres) are not grouped

// Child count word of a NodePair / the root reference:
//   0                      internal node (ref = NodePair index)
//   bit 31 set             ungrouped leaf: count = (word & 0x7fffffff) >> 6 (bits 5:0 are zero, so the grouped
//                          leaves' `word & 63` sees no group in it), primitives leaf[ref .. ref + count)
//   otherwise              grouped leaf: count = word & 63 (1..32), first group = word >> 6;
//                          group g covers slots (first + g) * 4 + [0, 4) of leaf_grouped
RT_LAYOUT_FN bool cw_is_leaf(uint32_t w) { return w != 0u; }
RT_LAYOUT_FN bool cw_is_big(uint32_t w) { return (w >> 31) != 0u; }
RT_LAYOUT_FN uint32_t cw_big_count(uint32_t w) { return (w & 0x7fffffffu) >> 6; }
RT_LAYOUT_FN uint32_t cw_make_big(uint32_t count) { return 0x80000000u | (count << 6); }

// ---- The traversal structure of our own (rt_wide.h) ------------------------------------------------------
// The reference's visit set does not depend on the tree ABOVE its leaves: its box test is monotone under box
// inclusion (rt_wide.h has the argument and its three exceptions), so a leaf is visited exactly when its OWN
// box passes the exact test. The kernels therefore walk a 4-wide SAH tree built over the TRIANGLES (not the
// reference's median-split tree) with cheap conservative box tests, run the reference's exact triangle test
// on the leaves' triangles, and check a would-be candidate against the exact test of the reference leaf it
// lives in (the winner only: rt_wide.h). Tie ranks (the index in the reference's visit order) ride along per slot.
//
// WideNode: four children, SoA so that one child is one lane of a float4. 128 bytes, 128-byte aligned.
// Child reference word:
//   0xffffffff          empty slot
//   bit 31 set          leaf group: bits 30:29 = triangles - 1 (1..4), bits 23:0 = group index;
//                       group g holds slots [4g, 4g + count) of `wide_leaf` / `wide_index`
//   otherwise           internal node: index into the WideNode array (< 2^26)
struct alignas(128) WideNode {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t ref[4];
    uint32_t pad[4];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be 128 bytes");
constexpr uint32_t kWideEmpty = 0xffffffffu;
constexpr uint32_t kWideMaxNodes = 1u << 26, kWideMaxGroups = 1u << 24;
RT_LAYOUT_FN bool wide_is_leaf(uint32_t ref) { return (ref >> 31) != 0u; }
RT_LAYOUT_FN uint32_t wide_leaf_group(uint32_t ref) { return ref & 0xffffffu; }
RT_LAYOUT_FN uint32_t wide_leaf_count(uint32_t ref) { return ((ref >> 29) & 3u) + 1u; }
RT_LAYOUT_FN uint32_t wide_make_leaf(uint32_t group, uint32_t count) { return 0x80000000u | ((count - 1u) << 29) | group; }

// Bounds of one reference leaf (what bvh.cpp:331-352 tests before the leaf is scanned), 32 bytes
struct alignas(16) LeafBox {
    float lo[3], hi[3];
    uint32_t pad[2];
};
static_assert(sizeof(LeafBox) == 32, "LeafBox");

// Material slot (framework mesh.h:22-34 without the shared_ptr)
struct alignas(16) MaterialRec {
    float kd[3];
    float shininess;
    float ks[3];
    float transparency;
    int32_t texture; // -1 none
    int32_t pad[3];
};
static_assert(sizeof(MaterialRec) == 48, "MaterialRec");

struct TextureRec {
    uint32_t offset; // first texel (in float3 units) in the texel pool
    int32_t width, height;
    int32_t pad;
};

struct LightRec {
    int32_t kind;
    float v[7][3];
    int32_t pad[2];
};
static_assert(sizeof(LightRec) == 96, "LightRec");

// Device view of a built scene + BVH (all pointers into HBM).
struct SceneView {
    const NodePair* nodes;
    const PrimRecord* prims;      // BVH mode: the reference's visit order (right-first DFS over the leaves)
    const PrimRecord* prims_flat; // brute-force mode (enableAccelStructure off): the reference's array order
    // LeafRecord arrays in the same two orders, as three 16-byte words per primitive
    const float4* leaf;
    const float4* leaf_flat;
    const float4* leaf_grouped; // LeafRecords in group order (4 slots per group, padded)
    const float4* groups;       // GroupRecord array, two 16-byte words each
    uint32_t n_groups;
    const uint32_t* tri_vidx;   // prim_id*3 → global vertex index (triangles only)
    const float* vnormal;       // global vertex normals, xyz
    const float* vtexcoord;     // global vertex texcoords, uv
    const MaterialRec* materials;
    const TextureRec* textures;
    const float* texels;        // rgb pool
    const LightRec* lights;
    uint32_t n_lights;
    uint32_t root_ref, root_cnt; // child reference of the root (never box-tested: bvh.cpp:313)
    uint32_t n_prims;
    uint32_t n_pairs; // internal nodes (NodePair records)
    uint32_t bounds_finite; // no node bound is NaN / infinite (rt_device.h: box_pair_hit fast path)
    uint32_t prune_ok; // 0 when the scene holds spheres (rt_device.h: Pruning)
    float extent;      // largest |coordinate| of any vertex
    float scene_lo[3], scene_hi[3]; // bounds of the root node (all primitives)
    unsigned long long* stamps; // diagnostic builds (-DRT_STAMPS) only, else NULL
#ifdef RT_DIAG // diagnostic builds: second copies at other addresses (tools/diag.sh)
    const NodePair* diag_nodes;
    const float4* diag_leaf;
    const void* diag_wnodes; // second copies of wide_nodes / wide_leaf
    const void* diag_wleaf;
#endif
    uint32_t leaf_flush; // leaf-queue fill at which the wave switches to the leaf phase (tuning, 2..64)
    // the wide tree (rt_wide.h); wide_ok == 0: not built for this scene (spheres, a root that is a leaf, too large)
    const WideNode* wide_nodes;
    const float4* wide_leaf;      // LeafRecords, four slots per group (padding slots are zero)
    const uint32_t* wide_index;    // per slot: the triangle's position in `prims` (the reference's visit order)
    const uint32_t* wide_ref_leaf; // per entry of `prims`: the reference leaf it lives in
    const LeafBox* wide_leaf_box;  // per reference leaf
    uint32_t wide_n_nodes, wide_n_groups, wide_n_ref_leaves, wide_ok;
    uint32_t wide_prefetch; // tuning: touch the next node's line at the end of a node step
};

} // namespace rt
