// rt_device.h — device-side building blocks of the Whitted path (gfx950).
//
// One wavefront lane = one ray. Why the traversal below may reorder work and
// still return the reference's bits:
//
//   * The reference never prunes: ray.t is reset to FLT_MAX around every box test
//     (src/bounding_volume_hierarchy.cpp:334-352), so the set of leaves a ray
//     visits is purely geometric — it does not depend on hits found so far.
//   * Within that set the reference tests primitives in a FIXED order (push left,
//     push right, pop right first; leaf primitives in array order) and accepts a
//     triangle when 0 <= t <= ray.t and the point is inside (Appendix A.2), the
//     last accepted one being reported (bvh.cpp:272-293). The inside test and t
//     depend on the triangle and the ray only, so the final result is: ray.t =
//     min t over the candidates (inside, 0 <= t <= initial ray.t), and the hit
//     primitive is the LAST candidate in the fixed order that attains that
//     minimum (an equal t is accepted again: "<="). Spheres use a strict "<"
//     (A.3), so among equal minima a sphere only wins when no triangle attains
//     the minimum, and then the FIRST such sphere does.
//
// Hence closest-hit = min over candidates of the key (t, tie rank) — a reduction
// that is independent of evaluation order. The host stores primitives in the
// reference's visit order (rt_layout.h), so "later in the fixed order" is simply
// "larger array index", and the kernel is free to:
//   - walk each ray's tree depth-first with a per-lane stack in LDS (node phase),
//   - park every (ray, leaf) pair it finds in a per-wave LDS queue, and
//   - test leaves in a separate phase where HALF A WAVE works on ONE pair, one
//     triangle per lane: consecutive lanes read consecutive 96-byte records
//     (coalesced), the ray is an LDS broadcast, and a candidate lane publishes
//     its key with one 64-bit LDS atomic-min. Shadow rays (any-hit) stop as soon
//     as their slot holds a key.
//
// Arithmetic: every float op is a separately rounded IEEE binary32 op in the
// reference's order (file built with -ffp-contract=off; '/' and sqrtf are
// correctly rounded on gfx950 with hipcc defaults).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_layout.h"

namespace rt {

#define RT_DEV __device__ __forceinline__
// constant address space: a wave-uniform index into such a pointer becomes scalar (SMEM) loads
#define RT_CONST_AS __attribute__((address_space(4)))

constexpr float kFltMax = 3.402823466e+38f;
constexpr float kFltMin = 1.175494351e-38f;
constexpr unsigned long long kNoHit = ~0ull;

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
RT_DEV V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

struct RayL { // one lane's ray
    V3 o, d;
};

// ---- wave helpers ---------------------------------------------------------
RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV uint64_t ballot(bool p) { return __ballot(p); }
RT_DEV uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
RT_DEV uint32_t prefix_count(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
RT_DEV uint32_t read_lane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// lanes of one wave exchange data through LDS: keep the compiler from moving LDS
// accesses across the hand-over (the hardware runs one wave's LDS ops in order)
RT_DEV void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct LaneCounters {
    uint32_t box = 0, tri = 0, sph = 0;     // lane-level tests
    uint32_t node_fetch = 0, tri_fetch = 0; // lane-level record fetches (64 B / 96 B each)
};

// Per-wave LDS working set (8.25 KiB): ray table, best-key slots, the per-lane
// deques (slot [level][lane]: conflict-free; entry = node pair index << 6 | ray
// slot) and the (ray, leaf) pair queue.
constexpr uint32_t kStackLevels = 16; // MAX_DEPTH = 16 levels ⇒ at most 15 internal levels pending
constexpr uint32_t kLeafQueue = 192;  // < 64 left over + at most 128 appended per node step
struct alignas(16) WaveLds {
    float4 rayA[64]; // ox oy oz dx
    float4 rayB[64]; // dy dz t0 -
    unsigned long long best[64];
    uint32_t stack[kStackLevels][64];
    uint32_t lq_ref[kLeafQueue];
    uint32_t lq_rc[kLeafQueue]; // ray slot (6 bits) | count << 6
    uint32_t xfer[64];          // work-stealing hand-over slots
};
static_assert(sizeof(WaveLds) == 8448, "WaveLds");

// ---- primitive tests (decls src/intersect.h:5-16) -----

// Correctly rounded n / d from a correctly rounded reciprocal y = RN(1/d):
//   q0 = RN(n*y);  r = n - d*q0 (exact, one FMA);  q = RN(q0 + r*y)
// equals RN(n/d) (Markstein's theorem) as long as nothing under/overflows on the
// way. The slab test divides TWELVE numerators by the same three ray-direction
// components, so the reciprocals are computed once per ray (IEEE '/') and each
// quotient costs 3 instructions instead of the ~11 of the full division sequence.
// `ok` is the per-axis guard computed with the reciprocal (RayInv): it is false
// for extreme |d| and for the all-ones significand that the theorem excludes;
// extreme or zero numerators also take the plain '/' path. The FMAs here are
// explicit — they do not change any result bit (rt_debug_div_check verifies
// q == n / d exhaustively on the device) and -ffp-contract=off stays in force
// for everything else.
struct RayInv {
    float y[3]; // RN(1 / d[a])
    bool ok[3];
};
RT_DEV RayInv make_ray_inv(const RayL& r)
{
    RayInv v;
    const float d[3] = { r.d.x, r.d.y, r.d.z };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        v.y[a] = 1.0f / d[a];
        const uint32_t b = __float_as_uint(d[a]);
        const uint32_t e = (b >> 23) & 0xffu;
        v.ok[a] = e >= 127u - 40u && e <= 127u + 40u && (b & 0x7fffffu) != 0x7fffffu;
    }
    return v;
}
RT_DEV float div_by_ray(float n, float d, float y, bool ok)
{
    const uint32_t e = (__float_as_uint(n) >> 23) & 0xffu;
    if (ok && e >= 127u - 60u && e <= 127u + 60u) {
        const float q0 = n * y;
        const float r = __builtin_fmaf(-d, q0, n);
        return __builtin_fmaf(r, y, q0);
    }
    return n / d;
}

// A.4 intersectRayWithShape(AxisAlignedBox) with ray.t = FLT_MAX (bvh.cpp:336,344):
// only the boolean matters.
RT_DEV bool box_hit(const float lo[3], const float hi[3], const RayL& r, const RayInv& inv)
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
            tl[a] = div_by_ray(lo[a] - o[a], d[a], inv.y[a], inv.ok[a]);
            th[a] = div_by_ray(hi[a] - o[a], d[a], inv.y[a], inv.ok[a]);
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

// A.2: t of the ray with the triangle's plane (n, D precomputed on the host with
// the same operation order — rt_layout.h)
RT_DEV float tri_plane_t(const PrimRecord& rec, const RayL& r)
{
    const V3 n = mk(rec.n[0], rec.n[1], rec.n[2]);
    return (rec.D - dot(r.o, n)) / dot(r.d, n);
}

// A.1 + A.2: is this triangle a candidate for a ray whose limit is t0?
// (0 <= t <= t0 and the three inclusive edge tests, in the reference's order)
RT_DEV bool tri_candidate(const PrimRecord& rec, const RayL& r, float t0, float& tt)
{
    tt = tri_plane_t(rec, r);
    if (!(tt >= 0.0f))
        return false;
    if (!(t0 >= tt))
        return false;
    const V3 p = r.d * tt + r.o;
    if (!(dot(ld3(rec.e0), p - ld3(rec.v0)) >= 0.0f))
        return false;
    if (!(dot(ld3(rec.e1), p - ld3(rec.v1)) >= 0.0f))
        return false;
    if (!(dot(ld3(rec.e2), p - ld3(rec.v2)) >= 0.0f))
        return false;
    return true;
}

// A.3 intersectRayWithShape(Sphere): assumes |direction| = 1, strict t < ray.t
RT_DEV bool sphere_candidate(const PrimRecord& rec, const RayL& r, float t0, float& tt)
{
    const V3 c = ld3(rec.v0);
    const float radius = rec.D;
    const V3 oc = r.o - c;
    const float h = dot(r.d, oc);
    const float B = h + h;
    const float C = dot(oc, oc) - radius * radius;
    const float disc = B * B - C * 4.0f;
    tt = kFltMax;
    if (disc < 0.0f)
        return false;
    float t0r, t1r;
    if (disc == 0.0f) {
        t0r = t1r = 0.5f * (-B);
    } else {
        const float q = sqrtf(disc);
        t1r = (q - B) * 0.5f;
        t0r = ((-B) - q) * 0.5f;
    }
    if (t1r >= 0.0f) {
        if (t0r >= 0.0f)
            tt = sel_min(t0r, t1r);
        else if (t1r <= kFltMax)
            tt = t1r;
    } else if (t0r >= 0.0f && t0r < kFltMax) {
        tt = t0r;
    }
    if (tt < 0.0f || tt >= t0)
        return false;
    return true;
}

// 64-bit reduction key: smaller = better. High word: t bits (t >= 0, so unsigned
// order = float order; -0.0f is folded onto +0.0f because the reference compares
// them equal). Low word: triangles first (bit 31 clear), larger index = smaller
// key ("last accepted wins"); spheres after them, smaller index first.
RT_DEV unsigned long long make_key(float tt, uint32_t index, bool sphere)
{
    const uint32_t tb = tt == 0.0f ? 0u : __float_as_uint(tt);
    const uint32_t low = sphere ? (0x80000000u | index) : (0x7fffffffu - index);
    return ((unsigned long long)tb << 32) | low;
}
RT_DEV uint32_t key_index(unsigned long long key)
{
    const uint32_t low = (uint32_t)key;
    return (low & 0x80000000u) ? (low & 0x7fffffffu) : (0x7fffffffu - low);
}

// ---- leaf phase -----------------------------------------------------------------
// Entries [first, first + n) of the wave's leaf queue, n <= 64: two entries per
// step, one per half-wave, one primitive per lane. The record fetch of step k+1
// is issued before step k is tested (software pipeline: the loop is otherwise a
// chain of dependent ~1 µs gathers).
struct LeafSlot {
    PrimRecord rec;
    uint32_t r, ref, cnt, cmax; // ray slot, first primitive, this half's count, max of both halves
    bool live;                  // this lane holds a record of the first 32 primitives
};

template <bool ANY>
RT_DEV void leaf_fetch(const WaveLds& L, const PrimRecord* __restrict__ prims, uint32_t my_ref, uint32_t my_rc,
    uint32_t k, uint32_t n, bool upper, uint32_t sub, LeafSlot& s)
{
    const uint32_t ref0 = read_lane(my_ref, k), rc0 = read_lane(my_rc, k);
    const uint32_t k1 = k + 1 < n ? k + 1 : k;
    const uint32_t ref1 = read_lane(my_ref, k1), rc1 = (k + 1 < n) ? read_lane(my_rc, k1) : 0u;
    s.ref = upper ? ref1 : ref0;
    const uint32_t rc = upper ? rc1 : rc0;
    s.r = rc & 63u;
    s.cnt = rc >> 6;
    const uint32_t c0 = rc0 >> 6, c1 = rc1 >> 6;
    s.cmax = c0 > c1 ? c0 : c1;
    s.live = sub < s.cnt;
    if (ANY && s.live && L.best[s.r] != kNoHit)
        s.live = false; // this shadow ray is already blocked
    if (s.live)
        s.rec = prims[s.ref + sub];
}

template <bool COUNT>
RT_DEV void leaf_test_one(WaveLds& L, const PrimRecord& rec, uint32_t r, uint32_t index, LaneCounters& lc)
{
    const float4 a = L.rayA[r], b = L.rayB[r];
    RayL ray;
    ray.o = mk(a.x, a.y, a.z);
    ray.d = mk(a.w, b.x, b.y);
    const float t0 = b.z;
    float tt;
    bool cand;
    const bool sphere = (rec.kind_mat & 0x80000000u) != 0u;
    if (!sphere) {
        cand = tri_candidate(rec, ray, t0, tt);
        if (COUNT)
            lc.tri++;
    } else {
        cand = sphere_candidate(rec, ray, t0, tt);
        if (COUNT)
            lc.sph++;
    }
    if (COUNT)
        lc.tri_fetch++;
    if (cand)
        atomicMin(&L.best[r], make_key(tt, index, sphere));
}

template <bool ANY, bool COUNT>
RT_DEV void leaf_phase(WaveLds& L, const PrimRecord* __restrict__ prims, uint32_t first, uint32_t n, uint32_t lane,
    LaneCounters& lc)
{
    uint32_t my_ref = 0, my_rc = 0;
    if (lane < n) {
        my_ref = L.lq_ref[first + lane];
        my_rc = L.lq_rc[first + lane];
    }
    const bool upper = lane >= 32u;
    const uint32_t sub = lane & 31u;
    LeafSlot cur, nxt;
    leaf_fetch<ANY>(L, prims, my_ref, my_rc, 0u, n, upper, sub, cur);
    for (uint32_t k = 0; k < n; k += 2) {
        nxt.live = false;
        nxt.cmax = 0;
        if (k + 2 < n)
            leaf_fetch<ANY>(L, prims, my_ref, my_rc, k + 2, n, upper, sub, nxt);
        if (cur.live)
            leaf_test_one<COUNT>(L, cur.rec, cur.r, cur.ref + sub, lc);
        // leaves larger than 32 primitives (brute-force mode, very large meshes)
        for (uint32_t base = 32u; base < cur.cmax; base += 32u) {
            const uint32_t idx = base + sub;
            if (idx < cur.cnt && !(ANY && L.best[cur.r] != kNoHit)) {
                const PrimRecord rec = prims[cur.ref + idx];
                leaf_test_one<COUNT>(L, rec, cur.r, cur.ref + idx, lc);
            }
        }
        cur = nxt;
    }
}

// ---- traversal ------------------------------------------------------------------
// ANY = false: closest hit ≙ BoundingVolumeHierarchy::intersect's traversal
//   (bvh.cpp:299-362) + getIntersecting (:272-293): `t` holds ray.t in/out, `pos`
//   the array index of the reported primitive (-1: none).
// ANY = true: shadow query (light.cpp:61-67 only uses the boolean): `t` is the
//   limit (1.0), pos >= 0 iff something blocks the ray.
// `active` selects the lanes that carry a ray; ALL lanes of the wave must call.
// use_bvh = features.enableAccelStructure; when false the whole array is one leaf
// in the reference's own order (bvh.cpp:303-305; sv.prims_flat).
//
// Node phase = depth-first walk with WORK STEALING inside the wave: every lane is
// a worker with a private deque of (ray slot, node) entries in LDS; a lane whose
// deque is empty takes the OLDEST entry (the largest pending subtree) of a lane
// that holds at least two. Because the closest-hit reduction is order-free, any
// lane may walk any part of any ray's tree; lanes that carry no ray of their own
// (sparse reflection levels, finished rays) pick up work immediately, so a wave's
// time is its total work / 64 rather than its longest ray.
template <bool ANY, bool COUNT>
RT_DEV void traverse(const SceneView& sv, bool use_bvh, bool active, const RayL& ray, float& t, int32_t& pos,
    LaneCounters& lc, WaveLds& L)
{
    pos = -1;
    if (ballot(active) == 0)
        return;
    const uint32_t lane = lane_id();
    const PrimRecord* __restrict__ prims = use_bvh ? sv.prims : sv.prims_flat;
    L.rayA[lane] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
    L.rayB[lane] = make_float4(ray.d.y, ray.d.z, t, 0.0f);
    L.best[lane] = kNoHit;

    uint32_t bot = 0, sp = 0; // this lane's deque = stack[bot .. sp)
    uint32_t lq_n = 0;        // leaf-queue fill (wave-uniform)
    RayL cur = ray;           // ray of the entry being processed (cached in registers)
    RayInv cur_inv = make_ray_inv(cur);
    uint32_t cur_r = lane;
    const bool root_is_leaf = !use_bvh || sv.root_cnt != 0u;
    if (root_is_leaf) {
        const uint64_t m = ballot(active);
        if (active) {
            const uint32_t o = prefix_count(m);
            L.lq_ref[o] = use_bvh ? sv.root_ref : 0u;
            L.lq_rc[o] = lane | ((use_bvh ? sv.root_cnt : sv.n_prims) << 6);
        }
        lq_n = (uint32_t)__builtin_popcountll(m);
    } else if (active) {
        L.stack[0][lane] = (sv.root_ref << 6) | lane; // the root's own box is never tested (bvh.cpp:313)
        sp = 1;
    }
    wave_lds_fence();

    for (;;) {
        const uint64_t busy = ballot(sp > bot);
        if (busy == 0)
            break;
        // ---- steal: idle lanes take the oldest entry of lanes holding >= 2 ----------
        if (busy != ~0ull) {
            const bool donor = sp - bot >= 2u;
            const uint64_t donors = ballot(donor);
            if (donors) {
                const uint64_t idle = ~busy;
                const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
                const uint32_t n_don = (uint32_t)__builtin_popcountll(donors);
                const uint32_t n_x = n_idle < n_don ? n_idle : n_don;
                const bool is_idle = sp == bot;
                if (donor) {
                    const uint32_t k = prefix_count(donors);
                    if (k < n_x) {
                        L.xfer[k] = L.stack[bot][lane];
                        bot++;
                    }
                }
                wave_lds_fence();
                if (is_idle) {
                    const uint32_t k = prefix_count(idle);
                    if (k < n_x) {
                        L.stack[0][lane] = L.xfer[k];
                        bot = 0;
                        sp = 1;
                    }
                }
                wave_lds_fence();
            }
        }
        // ---- node phase: every lane with work pops one internal node ---------------
        bool leaf_l = false, leaf_r = false;
        uint32_t l_ref = 0, l_cnt = 0, r_ref = 0, r_cnt = 0, r = 0;
        if (sp > bot) {
            sp--;
            const uint32_t e = L.stack[sp][lane];
            if (sp == bot)
                sp = bot = 0;
            r = e & 63u;
            if (!(ANY && L.best[r] != kNoHit)) { // a blocked shadow ray needs no more tests
                if (r != cur_r) {
                    const float4 a = L.rayA[r], b = L.rayB[r];
                    cur.o = mk(a.x, a.y, a.z);
                    cur.d = mk(a.w, b.x, b.y);
                    cur_inv = make_ray_inv(cur);
                    cur_r = r;
                }
                const NodePair* np = sv.nodes + (e >> 6);
                const float4 q0 = *reinterpret_cast<const float4*>(&np->l_lo[0]); // l_lo xyz, l_hi x
                const float4 q1 = *reinterpret_cast<const float4*>(&np->l_hi[1]); // l_hi yz, r_lo xy
                const float4 q2 = *reinterpret_cast<const float4*>(&np->r_lo[2]); // r_lo z, r_hi xyz
                const uint4 q3 = *reinterpret_cast<const uint4*>(&np->l_ref);
                const float llo[3] = { q0.x, q0.y, q0.z }, lhi[3] = { q0.w, q1.x, q1.y };
                const float rlo[3] = { q1.z, q1.w, q2.x }, rhi[3] = { q2.y, q2.z, q2.w };
                const bool hl = box_hit(llo, lhi, cur, cur_inv);
                const bool hr = box_hit(rlo, rhi, cur, cur_inv);
                l_ref = q3.x, l_cnt = q3.y, r_ref = q3.z, r_cnt = q3.w;
                if (COUNT) {
                    lc.box += 2;
                    lc.node_fetch++;
                }
                if (hl && l_cnt == 0u) {
                    L.stack[sp][lane] = (l_ref << 6) | r;
                    sp++;
                }
                if (hr && r_cnt == 0u) {
                    L.stack[sp][lane] = (r_ref << 6) | r;
                    sp++;
                }
                leaf_l = hl && l_cnt != 0u;
                leaf_r = hr && r_cnt != 0u;
            }
        }
        const uint64_t ml = ballot(leaf_l), mr = ballot(leaf_r);
        if (ml | mr) {
            if (leaf_l) {
                const uint32_t o = lq_n + prefix_count(ml);
                L.lq_ref[o] = l_ref;
                L.lq_rc[o] = r | (l_cnt << 6);
            }
            lq_n += (uint32_t)__builtin_popcountll(ml);
            if (leaf_r) {
                const uint32_t o = lq_n + prefix_count(mr);
                L.lq_ref[o] = r_ref;
                L.lq_rc[o] = r | (r_cnt << 6);
            }
            lq_n += (uint32_t)__builtin_popcountll(mr);
            if (lq_n >= 64u) {
                wave_lds_fence();
                do {
                    lq_n -= 64u;
                    leaf_phase<ANY, COUNT>(L, prims, lq_n, 64u, lane, lc);
                } while (lq_n >= 64u);
                wave_lds_fence();
            }
        }
    }
    if (lq_n) {
        wave_lds_fence();
        leaf_phase<ANY, COUNT>(L, prims, 0u, lq_n, lane, lc);
    }
    wave_lds_fence();
    const unsigned long long key = L.best[lane];
    if (active && key != kNoHit) {
        pos = (int32_t)key_index(key);
        if (!ANY) { // the reported primitive's own t, with its exact bits (-0.0f included)
            const PrimRecord rec = prims[pos];
            float tt;
            if (rec.kind_mat & 0x80000000u)
                sphere_candidate(rec, ray, t, tt);
            else
                tt = tri_plane_t(rec, ray);
            t = tt;
        }
    }
    wave_lds_fence(); // the caller may reuse L for the next query
}

} // namespace rt
