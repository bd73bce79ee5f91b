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
// "larger array index", and the traversal (rt_traverse.h) is free to walk the
// tree in any order, to split one ray's tree over several lanes, and to test
// leaves in a separate phase; a candidate publishes its key with one 64-bit LDS
// atomic-min into its ray's slot. Shadow rays (any-hit) stop as soon as their
// slot holds a key.
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

// Diagnostic build only (-DRT_STAMPS): per-phase shader-clock sums of lane 0 of
// every wave, written to SceneView::stamps (never to an output). Not compiled
// into the shipped library.
#ifdef RT_STAMPS
#define RT_STAMP(var) const unsigned long long var = clock64()
#define RT_STAMP_ADD(slot, a, b) stamp_acc[slot] += (b) - (a), stamp_n[slot]++
#else
#define RT_STAMP(var)
#define RT_STAMP_ADD(slot, a, b)
#endif

// 128-bit buffer loads through a resource descriptor: a 32-bit byte offset per lane instead
// of 64-bit address arithmetic, hardware range check, and exactly one dwordx4 per call
// (the compiler splits and overlaps plain float4 loads whose components it tracks one by one).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t BufRsrc;
RT_DEV BufRsrc make_rsrc(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
RT_DEV float4 buf_load16(BufRsrc r, uint32_t byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct LaneCounters {
    uint32_t box = 0, tri = 0, sph = 0;     // lane-level tests
    uint32_t grp = 0;                       // lane-level group-box tests (leaf phase)
    uint32_t node_fetch = 0, tri_fetch = 0; // lane-level record fetches (64 B / 96 B each)
    uint32_t exact_rays = 0;                // rays this lane's slot sent through the exact-topology engine
#ifdef RT_TIMELINE
    uint32_t tl_node_iters = 0, tl_leaf_steps = 0, tl_traversals = 0; // wave-uniform step counts (diagnostic)
#endif
};

// Per-wave LDS working set (10 240 B): ray table and best-key slots, shared by the two traversal engines, and
// a union of their private parts. Exact-topology engine (rt_traverse.h): per-lane deques (slot [level][lane]:
// conflict-free; entry = node pair index << 6 | ray slot, plus the entry's distance lower bound), the
// (ray, leaf) pair queue and the surviving-group list. Wide engine (rt_wide.h): per-lane deques of
// (node, ray) entries with a 16-bit lower bound each, the (ray, leaf group) queue and the mask of rays that must
// take the exact engine.
constexpr uint32_t kStackLevels = 16; // MAX_DEPTH = 16 levels ⇒ at most 15 internal levels pending
constexpr uint32_t kLeafFlushMax = 36; // largest leaf_flush threshold (SceneView::leaf_flush)
constexpr uint32_t kLeafQueue = 164;   // < kLeafFlushMax left over + at most 128 appended per node step
constexpr uint32_t kWideStack = 13;      // deque slots per lane; a lane that needs more sends its ray to the exact engine
constexpr uint32_t kWideFlushMax = 28;   // largest flush threshold of the wide engine's leaf queue
constexpr uint32_t kWideLeafQueue = 284; // < kWideFlushMax left over + at most 256 appended per node step
struct alignas(16) WaveLds {
    float4 rayA[64]; // ox oy oz dx
    float4 rayB[64]; // dy dz t0 guard bits (RayInv::ok[0..2], all_ok)
    float4 rayC[64]; // RN(1/dx) RN(1/dy) RN(1/dz) pruning slack — RayInv, computed once per ray
    unsigned long long best[64];
    uint32_t xfer[64]; // work-stealing hand-over slots
    union {
        struct { // exact-topology engine
            uint32_t stack[kStackLevels][64]; // packed entries, see pack_entry()
            uint32_t lq_ref[kLeafQueue];
            uint32_t lq_cw[kLeafQueue]; // count word of the leaf (rt_layout.h); lq_ref holds ref | ray slot << 26
            float lq_lb[kLeafQueue];
            uint32_t surv[80];          // leaf phase: surviving (pair, group) entries, < 16 left over + 64 new
        };
        struct { // wide engine
            uint32_t wstack[kWideStack][64];    // node index << 6 | ray slot
            uint16_t wstack_lb[kWideStack][64]; // the entry's lower bound: top 16 bits of the float, i.e. rounded down
            uint32_t wlq[kWideLeafQueue];       // leaf group index | (triangles - 1) << 24 | ray slot << 26
            uint16_t wxfer_lb[64];
            uint32_t wfallback[2];              // ray slots that must be traced by the exact engine
        };
    };
};
static_assert(sizeof(WaveLds) <= 10240, "WaveLds"); // one wave per workgroup: 16 of them fit a CU's 160 KiB (163 840 B)

// A deque entry in 32 bits: [31:21] lower bound, [20:6] node pair index, [5:0] ray slot.
// MAX_DEPTH = 16 levels caps a tree at 32 767 internal nodes (15 bits). The bound keeps
// the top 11 bits of the non-negative float (8 exponent + 3 mantissa bits), truncated,
// i.e. rounded DOWN — still a valid lower bound, at most 12.5 % looser.
RT_DEV uint32_t pack_entry(uint32_t node, uint32_t ray, float lb)
{
    const uint32_t b = lb > 0.0f ? (__float_as_uint(lb) >> 20) : 0u; // NaN / negative: no bound
    return (b << 21) | (node << 6) | ray;
}
RT_DEV float entry_lb(uint32_t e) { return __uint_as_float((e >> 21) << 20); }
RT_DEV uint32_t entry_node(uint32_t e) { return (e >> 6) & 0x7fffu; }
RT_DEV uint32_t entry_ray(uint32_t e) { return e & 63u; }

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
    uint32_t guard; // bit a: axis a may use the reciprocal; bit 3: all three may AND |origin| + the scene's
                    // largest |coordinate| < 2^60 (then no component of d is zero, no slab quotient can be
                    // NaN, and every numerator bound - origin of a finite box is below 2^61)
    RT_DEV bool ok(int a) const { return (guard >> a) & 1u; }
    RT_DEV bool all_ok() const { return (guard & 8u) != 0u; }
    float slack; // rho * max_a |1/d[a]|: how far a slab plane moved by rho shifts t (pruning bound)
};
// `extent` = largest |coordinate| of the scene (SceneView::extent).
RT_DEV RayInv make_ray_inv(const RayL& r, float extent)
{
    RayInv v;
    v.guard = 0u;
    const float d[3] = { r.d.x, r.d.y, r.d.z };
    float ymax = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        v.y[a] = 1.0f / d[a];
        const uint32_t b = __float_as_uint(d[a]);
        const uint32_t e = (b >> 23) & 0xffu;
        if (e >= 127u - 40u && e <= 127u + 40u && (b & 0x7fffffu) != 0x7fffffu)
            v.guard |= 1u << a;
        if (d[a] != 0.0f)
            ymax = fmaxf(ymax, fabsf(v.y[a]));
    }
    // rho (see "Pruning" below): 2^-16 of the largest coordinate magnitude in play
    const float S = fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fmaxf(fabsf(r.o.z), extent));
    v.slack = (S * 1.52587890625e-05f) * ymax;
    const float omax = fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z));
    const bool o_small = r.o.x == r.o.x && r.o.y == r.o.y && r.o.z == r.o.z && omax + extent < 0x1p60f;
    if (v.guard == 7u && o_small)
        v.guard = 15u;
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
// only the boolean matters for the reference's visit set. `lb` is the pruning
// lower bound derived from the same slab quotients (see "Pruning").
//
// Pruning. The reference visits every leaf whose ancestors' boxes pass this test
// and keeps the candidate with the smallest t. A subtree can be skipped without
// changing that result when no primitive in it can be a candidate with
// t <= limit (limit = min(initial ray.t, best t so far); the comparison is strict,
// so equal-t ties are never cut). For a triangle T inside box B, a candidate's
// point p = fl(d*t + o) passed the three edge tests and t = fl(num/den): p lies
// within a few ulps of the scene scale S of T (relative rounding of num/den keeps
// o + d*t within c*u*S of T's plane, the edge tests pin its in-plane position to
// within 4u*|p - v| of T), hence inside B grown by rho = 2^-16 * S on every side —
// 2^8 times the worst rounding bound. So t >= the ray's exact entry distance into
// B grown by rho >= tin * (1 - 2^-20) - rho * max|1/d| =: lb, with tin the TRUE entry
// distance computed below (before the reference's "origin inside" rewrite).
// Spheres are excluded (their test assumes |d| = 1): scenes with spheres do not
// prune. Proven against the unpruned traversal in tests/test_gpu_parity.py.
// the six slab quotients of one box, guarded division per quotient (any input)
RT_DEV void slab_quotients(const float lo[3], const float hi[3], const RayL& r, const RayInv& inv, float tl[3],
    float th[3])
{
    const float o[3] = { r.o.x, r.o.y, r.o.z };
    const float d[3] = { r.d.x, r.d.y, r.d.z };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (d[a] == 0.0f) {
            tl[a] = kFltMin;
            th[a] = kFltMax;
        } else {
            tl[a] = div_by_ray(lo[a] - o[a], d[a], inv.y[a], inv.ok(a));
            th[a] = div_by_ray(hi[a] - o[a], d[a], inv.y[a], inv.ok(a));
        }
    }
}

// the reference's decision on the six quotients, and the pruning bound
RT_DEV bool slab_decide(const float tl[3], const float th[3], float slack, float& lb)
{
    float tin = sel_max(sel_min(th[2], tl[2]), sel_max(sel_min(th[1], tl[1]), sel_min(th[0], tl[0])));
    float tout = sel_min(sel_max(th[2], tl[2]), sel_min(sel_max(th[1], tl[1]), sel_max(th[0], tl[0])));
    lb = tin * 0.99999904632568359375f - slack; // (1 - 2^-20); negative or NaN never prunes
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

RT_DEV bool box_hit(const float lo[3], const float hi[3], const RayL& r, const RayInv& inv, float& lb)
{
    float tl[3], th[3];
    slab_quotients(lo, hi, r, inv, tl, th);
    return slab_decide(tl, th, inv.slack, lb);
}

// Both boxes of a node pair. The twelve quotients share ONE guard: when the ray's three
// reciprocals are usable (all_ok) and every numerator's magnitude lies in
// [2^-60, 2^61) — the same windows div_by_ray checks one by one — all twelve take the
// three-instruction path without a branch each (`bounds_finite`: no node bound of the
// tree is NaN or infinite — SceneView::bounds_finite); otherwise (origin exactly on a slab
// plane, an axis-parallel ray, extreme values) the pair falls back to the guarded
// per-quotient code. Same results either way, ~half the instructions on the hot path.
RT_DEV void box_pair_hit(const float llo[3], const float lhi[3], const float rlo[3], const float rhi[3],
    const RayL& r, const RayInv& inv, bool bounds_finite, bool& hl, bool& hr, float& lb_l, float& lb_r)
{
    const float o[3] = { r.o.x, r.o.y, r.o.z };
    const float d[3] = { r.d.x, r.d.y, r.d.z };
    float n[12];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        n[a] = llo[a] - o[a];
        n[3 + a] = lhi[a] - o[a];
        n[6 + a] = rlo[a] - o[a];
        n[9 + a] = rhi[a] - o[a];
    }
    // the numerators' upper bound is the ray's (RayInv::all_ok: finite boxes lie within the scene's extent);
    // the lower one is per box
    float amin = fabsf(n[0]);
#pragma unroll
    for (int i = 1; i < 12; i++)
        amin = fminf(amin, fabsf(n[i]));
    if (inv.all_ok() && bounds_finite && amin >= 0x1p-60f) {
        // Every quotient is finite here, so IEEE min / max (v_min_f32, v_max3_f32: one
        // instruction each) agree with the reference's compare-and-select up to the sign
        // of a zero, which no comparison below can see; slab_decide() reduces to
        //   tin < 0 ? tout > 0 : tin <= tout.
        float q[12];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float y = inv.y[a], nd = -d[a];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float q0 = n[3 * k + a] * y;
                q[3 * k + a] = __builtin_fmaf(__builtin_fmaf(nd, q0, n[3 * k + a]), y, q0);
            }
        }
        const float tin_l = fmaxf(fmaxf(fminf(q[0], q[3]), fminf(q[1], q[4])), fminf(q[2], q[5]));
        const float tout_l = fminf(fminf(fmaxf(q[0], q[3]), fmaxf(q[1], q[4])), fmaxf(q[2], q[5]));
        const float tin_r = fmaxf(fmaxf(fminf(q[6], q[9]), fminf(q[7], q[10])), fminf(q[8], q[11]));
        const float tout_r = fminf(fminf(fmaxf(q[6], q[9]), fmaxf(q[7], q[10])), fmaxf(q[8], q[11]));
        lb_l = tin_l * 0.99999904632568359375f - inv.slack;
        lb_r = tin_r * 0.99999904632568359375f - inv.slack;
        // = tin < 0 ? tout > 0 : tin <= tout (no NaN here), written as lane-mask logic: three compares and
        // scalar and / or instead of value selects
        hl = (tin_l <= tout_l) & ((tout_l > 0.0f) | (tin_l >= 0.0f));
        hr = (tin_r <= tout_r) & ((tout_r > 0.0f) | (tin_r >= 0.0f));
    } else {
        float tl_l[3], th_l[3], tl_r[3], th_r[3];
        slab_quotients(llo, lhi, r, inv, tl_l, th_l);
        slab_quotients(rlo, rhi, r, inv, tl_r, th_r);
        hl = slab_decide(tl_l, th_l, inv.slack, lb_l);
        hr = slab_decide(tl_r, th_r, inv.slack, lb_r);
    }
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
    if (!(t0 >= tt)) // t0 may be any value in [best t so far, initial ray.t]: "<=" keeps equal-t ties
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

// The same test from the leaf phase's 48-byte record (v0, v1, v2, n): D and the edge
// vectors are recomputed with the host's operation order (host_scene.cpp), so every
// intermediate equals the PrimRecord field bit for bit.
RT_DEV bool tri_candidate_leaf(V3 v0, V3 v1, V3 v2, V3 n, const RayL& r, float t0, float& tt)
{
    const float D = dot(v0, n);
    tt = (D - dot(r.o, n)) / dot(r.d, n);
    // The reference's early returns (A.1, A.2) as one conjunction: with up to 64 triangles in flight some lane
    // nearly always reaches the last edge test, so branches only cost; every comparison is false for NaN as there.
    const V3 p = r.d * tt + r.o;
    const float e0 = dot(cross(v2 - v0, n), p - v0), e1 = dot(cross(v0 - v1, n), p - v1);
    const float e2 = dot(cross(v1 - v2, n), p - v2);
    return (tt >= 0.0f) & (t0 >= tt) & (e0 >= 0.0f) & (e1 >= 0.0f) & (e2 >= 0.0f);
}

// A.3 intersectRayWithShape(Sphere): assumes |direction| = 1, strict t < ray.t
RT_DEV bool sphere_candidate(V3 c, float radius, const RayL& r, float t0, float& tt)
{
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

RT_DEV bool sphere_candidate(const PrimRecord& rec, const RayL& r, float t0, float& tt)
{
    return sphere_candidate(ld3(rec.v0), rec.D, r, t0, tt);
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


// limit of ray slot r for pruning: min(initial ray.t, best t so far) — both are
// >= 0, so the unsigned order of their bit patterns is their float order, and the
// empty slot (all ones) never wins. A negative / NaN initial t has no candidates.
RT_DEV float ray_limit(const WaveLds& L, uint32_t r)
{
    const uint32_t t0b = __float_as_uint(L.rayB[r].z);
    const uint32_t bb = (uint32_t)(L.best[r] >> 32);
    return __uint_as_float(t0b < bb ? t0b : bb);
}

} // namespace rt
