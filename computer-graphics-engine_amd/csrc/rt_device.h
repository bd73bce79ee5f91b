// rt_device.h — device-side building blocks of the Whitted path (gfx950).
//
// One wavefront lane = one ray. A WAVE walks the BVH as a packet: the visit
// order of the reference's traversal (src/bounding_volume_hierarchy.cpp:312-362)
// is fixed — push left, push right, pop right first — and never depends on the
// ray or on the closest hit so far (ray.t is reset to FLT_MAX around every box
// test, :334-352). The set of nodes a ray visits is therefore purely geometric,
// and every lane's visit sequence is a subsequence of one fixed DFS order. The
// wave walks the UNION of its lanes' visit sets in that order, carrying a
// 64-bit lane mask per stack entry; a lane takes part in a node exactly when
// the reference would have visited it for that lane's ray, so per-ray results
// (t bits, last-accepted primitive on ties) are those of the scalar reference.
// What the packet buys on CDNA4: node and primitive records are fetched once
// per wave with scalar (SMEM) loads through the constant address space instead
// of 64 divergent vector loads, control flow is wave-uniform, and the traversal
// stack is 4 VGPRs used as a lane-indexed array (slot i lives in lane i).
//
// Arithmetic: every float op is a separately rounded IEEE binary32 op in the
// reference's order (file built with -ffp-contract=off; '/' and sqrtf are
// correctly rounded on gfx950 with hipcc defaults).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_layout.h"

namespace rt {

#define RT_CONST_AS __attribute__((address_space(4)))
#define RT_DEV __device__ __forceinline__

constexpr float kFltMax = 3.402823466e+38f;
constexpr float kFltMin = 1.175494351e-38f;

struct V3 {
    float x, y, z;
};
RT_DEV V3 mk(float x, float y, float z) { return V3 { x, y, z }; }
RT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_DEV V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
RT_DEV V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// glm dot: (x + y) + z; glm cross; normalize = v * (1 / sqrt(dot)); length = sqrt(dot)
RT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
RT_DEV V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
RT_DEV float length(V3 v) { return sqrtf(dot(v, v)); }
RT_DEV bool all_zero(V3 v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }
// second operand wins when the compare is false/unordered (x86 minss/maxss)
RT_DEV float sel_min(float a, float b) { return a < b ? a : b; }
RT_DEV float sel_max(float a, float b) { return a > b ? a : b; }

struct RayL { // one lane's ray
    V3 o, d;
};

// ---- wave helpers ---------------------------------------------------------
RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV uint64_t ballot(bool p) { return __ballot(p); }
RT_DEV bool lane_in(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
RT_DEV uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
RT_DEV uint32_t prefix_count(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Lane-indexed wave stack: entry i = {ref, cnt, mask} stored in lane i of four VGPRs.
struct WaveStack {
    uint32_t ref, cnt, mlo, mhi;
    RT_DEV void push(uint32_t lane, uint32_t sp, uint32_t r, uint32_t c, uint64_t m)
    {
        const bool here = lane == sp;
        ref = here ? r : ref;
        cnt = here ? c : cnt;
        mlo = here ? (uint32_t)m : mlo;
        mhi = here ? (uint32_t)(m >> 32) : mhi;
    }
    RT_DEV void pop(uint32_t sp, uint32_t& r, uint32_t& c, uint64_t& m) const
    {
        r = (uint32_t)__builtin_amdgcn_readlane((int)ref, (int)sp);
        c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)sp);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)mlo, (int)sp);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)mhi, (int)sp);
        m = ((uint64_t)hi << 32) | lo;
    }
};

struct LaneCounters {
    uint32_t box = 0, tri = 0, sph = 0;     // lane-level tests
    uint32_t node_fetch = 0, tri_fetch = 0; // wave-level record fetches (same value in every lane)
};

// ---- primitive tests (decls src/intersect.h:5-16) -----

// A.4 intersectRayWithShape(AxisAlignedBox) with ray.t = FLT_MAX (bvh.cpp:336,344):
// only the boolean matters. lo/hi are wave-uniform (SGPR) box bounds.
RT_DEV bool box_hit(const float RT_CONST_AS* lo, const float RT_CONST_AS* hi, const RayL& r)
{
    float tl[3], th[3];
    const float o[3] = { r.o.x, r.o.y, r.o.z };
    const float d[3] = { r.d.x, r.d.y, r.d.z };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (d[a] == 0.0f) {
            tl[a] = kFltMin;
            th[a] = kFltMax;
        } else {
            tl[a] = (lo[a] - o[a]) / d[a];
            th[a] = (hi[a] - o[a]) / d[a];
        }
    }
    float tin = sel_max(sel_min(th[2], tl[2]), sel_max(sel_min(th[1], tl[1]), sel_min(th[0], tl[0])));
    float tout = sel_min(sel_max(th[2], tl[2]), sel_min(sel_max(th[1], tl[1]), sel_max(th[0], tl[0])));
    if (tin < 0.0f) {
        if (!(tout > 0.0f))
            return false;
        tin = tout;
        tout = kFltMax;
    }
    if (tin > tout || tin < 0.0f || tin > kFltMax)
        return false;
    return true;
}

// A.1 + A.2 with the per-triangle part (n, D, edge vectors) read from the record.
// Returns true and updates t when the reference would accept (0 <= t' <= t, inside).
RT_DEV bool tri_hit(const PrimRecord RT_CONST_AS* rec, const RayL& r, float& t)
{
    const V3 n = mk(rec->n[0], rec->n[1], rec->n[2]);
    const float tt = (rec->D - dot(r.o, n)) / dot(r.d, n);
    if (!(tt >= 0.0f))
        return false;
    if (!(t >= tt))
        return false;
    const V3 p = r.d * tt + r.o;
    if (!(dot(mk(rec->e0[0], rec->e0[1], rec->e0[2]), p - mk(rec->v0[0], rec->v0[1], rec->v0[2])) >= 0.0f))
        return false;
    if (!(dot(mk(rec->e1[0], rec->e1[1], rec->e1[2]), p - mk(rec->v1[0], rec->v1[1], rec->v1[2])) >= 0.0f))
        return false;
    if (!(dot(mk(rec->e2[0], rec->e2[1], rec->e2[2]), p - mk(rec->v2[0], rec->v2[1], rec->v2[2])) >= 0.0f))
        return false;
    t = tt;
    return true;
}

// A.3 intersectRayWithShape(Sphere): assumes |direction| = 1, strict t < ray.t
RT_DEV bool sphere_hit(const PrimRecord RT_CONST_AS* rec, const RayL& r, float& t)
{
    const V3 c = mk(rec->v0[0], rec->v0[1], rec->v0[2]);
    const float radius = rec->D;
    const V3 oc = r.o - c;
    const float h = dot(r.d, oc);
    const float B = h + h;
    const float C = dot(oc, oc) - radius * radius;
    const float disc = B * B - C * 4.0f;
    if (disc < 0.0f)
        return false;
    float t0, t1;
    if (disc == 0.0f) {
        t0 = t1 = 0.5f * (-B);
    } else {
        const float q = sqrtf(disc);
        t1 = (q - B) * 0.5f;
        t0 = ((-B) - q) * 0.5f;
    }
    float tt = kFltMax;
    if (t1 >= 0.0f) {
        if (t0 >= 0.0f)
            tt = sel_min(t0, t1);
        else if (t1 <= kFltMax)
            tt = t1;
    } else if (t0 >= 0.0f && t0 < kFltMax) {
        tt = t0;
    }
    if (tt < 0.0f || tt >= t)
        return false;
    t = tt;
    return true;
}

// ---- packet traversal -------------------------------------------------------
// ANY = false: closest hit, ≙ the traversal part of BoundingVolumeHierarchy::intersect
//   (bvh.cpp:299-362) + getIntersecting (:272-293). `t` holds ray.t in/out, `pos`
//   receives the permuted index of the last accepted primitive (-1: none).
// ANY = true: shadow query — the reference runs the same closest-hit traversal and
//   only uses the boolean (light.cpp:61-67); a lane leaves the packet at its first
//   accepted primitive. `t` is the ray's t limit (1.0 for shadow rays).
// `active` selects the lanes that carry a ray; all lanes of the wave must call.
// use_bvh = features.enableAccelStructure; when false every primitive is tested
// in permuted order (bvh.cpp:303-305).
template <bool ANY, bool COUNT>
RT_DEV void traverse(const SceneView& sv, bool use_bvh, bool active, const RayL& ray, float& t, int32_t& pos,
    LaneCounters& lc)
{
    pos = -1;
    uint64_t alive = ballot(active);
    if (alive == 0)
        return;
    const uint32_t lane = lane_id();
    WaveStack st { 0u, 0u, 0u, 0u };
    uint32_t sp = 0;
    uint32_t ref = use_bvh ? sv.root_ref : 0u;
    uint32_t cnt = use_bvh ? sv.root_cnt : sv.n_prims;
    uint64_t mask = alive;
    const NodePair RT_CONST_AS* nodes = (const NodePair RT_CONST_AS*)sv.nodes;
    const PrimRecord RT_CONST_AS* prims = (const PrimRecord RT_CONST_AS*)sv.prims;

    for (;;) {
        if (cnt == 0u) {
            // internal node: test both children's boxes for the lanes in `mask`
            const NodePair RT_CONST_AS* np = nodes + ref;
            const bool in = lane_in(mask);
            bool hl = false, hr = false;
            if (in) {
                hl = box_hit(np->l_lo, np->l_hi, ray);
                hr = box_hit(np->r_lo, np->r_hi, ray);
                if (COUNT)
                    lc.box += 2;
            }
            if (COUNT)
                lc.node_fetch++;
            const uint64_t ml = ballot(hl), mr = ballot(hr);
            const uint32_t l_ref = np->l_ref, l_cnt = np->l_cnt, r_ref = np->r_ref, r_cnt = np->r_cnt;
            if (mr) { // right child is popped first (bvh.cpp:354-355)
                if (ml) {
                    st.push(lane, sp, l_ref, l_cnt, ml);
                    sp++;
                }
                ref = r_ref, cnt = r_cnt, mask = mr;
                continue;
            }
            if (ml) {
                ref = l_ref, cnt = l_cnt, mask = ml;
                continue;
            }
        } else {
            // leaf: every primitive of [ref, ref+cnt) in order, last accepted wins (bvh.cpp:272-293)
            const bool in = lane_in(mask);
            for (uint32_t i = 0; i < cnt; i++) {
                const PrimRecord RT_CONST_AS* rec = prims + (ref + i);
                bool hit = false;
                if (COUNT)
                    lc.tri_fetch++;
                if ((rec->kind_mat & 0x80000000u) == 0u) {
                    if (in && (!ANY || pos < 0)) {
                        hit = tri_hit(rec, ray, t);
                        if (COUNT)
                            lc.tri++;
                    }
                } else {
                    if (in && (!ANY || pos < 0)) {
                        hit = sphere_hit(rec, ray, t);
                        if (COUNT)
                            lc.sph++;
                    }
                }
                if (hit)
                    pos = (int32_t)(ref + i);
                if (ANY) {
                    alive &= ~ballot(hit);
                    if ((mask & alive) == 0)
                        break;
                }
            }
        }
        // pop the next entry that still has live lanes
        for (;;) {
            if (sp == 0u)
                return;
            sp--;
            st.pop(sp, ref, cnt, mask);
            if (ANY)
                mask &= alive;
            if (mask != 0)
                break;
        }
    }
}

} // namespace rt
