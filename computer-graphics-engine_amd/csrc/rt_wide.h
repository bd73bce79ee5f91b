// rt_wide.h — the wide engine: traversal of our own 4-wide SAH tree (rt_layout.h WideNode, host_wide.cpp)
// that returns the bits of the reference's traversal of ITS tree.
//
// Why a different tree may be walked. For a ray whose three direction components are non-zero, finite and of
// moderate magnitude and whose origin is finite (RayInv::all_ok) every slab quotient of the reference's box
// test is finite, and the test reduces to
//     T(box) = tin < 0 ? tout > 0 : tin <= tout        (ray.t = FLT_MAX at the call, bvh.cpp:336,344)
// with tin = max over the axes of the nearer quotient, tout = min of the farther one. RN(x - o) and RN(n / d)
// are monotone, so for a child box inside its parent's box every nearer quotient of the parent is <= the
// child's and every farther one >=: tin_p <= tin_c, tout_p >= tout_c. Hence T(child) implies T(parent),
// except when tout_p == 0 with tin_p < 0 — and then 0 <= tin_c <= tout_c <= tout_p = 0. The reference's node
// boxes are exact unions of their primitives' bounds, every box is inside its parent's, and the root's own box
// is never tested (bvh.cpp:313). So for such a ray
//     the reference visits a leaf  <=>  T(leaf's own box), unless that test passes with tout == 0,
// whatever the tree above the leaves looks like. The wide engine therefore
//   * walks the wide tree with a CONSERVATIVE box test: one FMA per slab plane, the interval widened by the
//     pruning slack (rt_device.h "Pruning": a candidate's point lies within rho = 2^-16 x scene scale of its
//     triangle, hence of every box around it; the FMA form adds at most 2^-22 x scale x |1/d| of rounding,
//     1/64 of the slack), pruned by min(initial ray.t, best t so far) like the exact engine;
//   * runs the reference's exact triangle test (Appendix A.1/A.2) on the triangles of the leaf groups reached;
//   * keeps, per ray, the smallest key among ALL triangles that pass that test (wide_index gives a slot's index in
//     the reference's visit order, so ties resolve as there) and prunes with it;
//   * at the end checks the WINNER against the exact box test of the reference leaf it lives in (wide_ref_leaf,
//     LeafBox) — the reference's "is this leaf visited". If the winner's leaf is visited, the winner is the
//     reference's result: every triangle of a visited leaf that passes the test was a contender (pruning only cut
//     subtrees whose lower bound exceeded some contender's t >= the winner's), and the contenders are a superset
//     of the reference's candidates, so the smallest key among them, being one of the reference's candidates, is
//     the smallest of those. (A shadow ray's winner is "some blocker": valid as soon as its leaf is visited.)
//   * sends a ray to the exact-topology engine (rt_traverse.h) when the winner's leaf fails that test or passes it
//     with tout == 0, when the ray is not all_ok (zero / tiny / huge / NaN components), or when a lane's deque
//     would overflow. Rare: a ray has to graze the corner of a leaf box for the first, sit on a box face for the second.
#pragma once
#include "rt_device.h"

namespace rt {

template <int MODE>
RT_DEV bool wide_any_hit_slot(uint32_t r)
{
    return MODE == 1 /* Q_ANY */ || (MODE == 2 /* Q_MIXED */ && r >= 32u);
}

RT_DEV uint32_t lb_to_u16(float lb) { return lb > 0.0f ? (__float_as_uint(lb) >> 16) : 0u; } // NaN / negative: no bound
RT_DEV float u16_to_lb(uint32_t h) { return __uint_as_float(h << 16); }

template <int MODE, bool COUNT>
struct WideWalk {
    const SceneView& sv;
    WaveLds& L;
    LaneCounters& lc;
    uint32_t lane;
    BufRsrc nodes, leaf, index;
#ifdef RT_DIAG
    BufRsrc nodes2, leaf2; // second copies (diagnostic builds: tools/ab.py)
#endif
    uint32_t lq_n = 0; // wave-uniform fill of the leaf queue
    uint32_t flush_at;
#ifdef RT_STAMPS
    // diagnostic build: shader-clock sums of this wave's phases (tools/stamps.py)
    unsigned long long st_steal = 0, st_node = 0, st_leaf = 0, n_node = 0, n_leaf = 0;
    unsigned long long st_pop = 0, st_load = 0, st_alu = 0, st_push = 0;
#endif

    RT_DEV void flag_fallback(uint32_t r) { atomicOr(&L.wfallback[r >> 5], 1u << (r & 31u)); }

    // ---- leaf phase: four lanes per (ray, leaf group) pair, one triangle per lane ------------------------
    RT_DEV void flush_leaves()
    {
#ifdef RT_STAMPS
        const unsigned long long tl0 = clock64();
#endif
        wave_lds_fence();
        for (uint32_t base = 0; base < lq_n; base += 16u) {
            const uint32_t p = base + (lane >> 2), j = lane & 3u;
            if (p < lq_n) {
                const uint32_t e = L.wlq[p], r = e >> 26, g = e & 0xffffffu, cnt = ((e >> 24) & 3u) + 1u;
                const bool any = wide_any_hit_slot<MODE>(r);
                const float4 rb = L.rayB[r];
                const unsigned long long bk = L.best[r];
                const uint32_t t0b = __float_as_uint(rb.z), bb = (uint32_t)(bk >> 32);
                const float lim = __uint_as_float((any || t0b < bb) ? t0b : bb);
                const bool done = any & (bk != kNoHit);
                if (j < cnt && !done) {
                    const uint32_t slot = g * 4u + j, off = slot * 48u;
                    // the index rides along with the triangle (issued first, so that it costs no round trip of its
                    // own when the triangle turns out to be a contender; the asm keeps the load from sinking there)
                    uint32_t idx = __builtin_amdgcn_raw_buffer_load_b32(index, slot * 4u, 0, 0);
                    const float4 c0 = buf_load16(leaf, off), c1 = buf_load16(leaf, off + 16u), c2 = buf_load16(leaf, off + 32u);
                    asm volatile("" : "+v"(idx));
#if defined(RT_DIAG) && RT_DIAG == 12 // the same three loads from a second copy of the leaf array
                    {
                        const float4 d0 = buf_load16(leaf2, off), d1 = buf_load16(leaf2, off + 16u), d2 = buf_load16(leaf2, off + 32u);
                        asm volatile("" ::"v"(d0.x), "v"(d1.y), "v"(d2.z));
                    }
#endif
                    const float4 a = L.rayA[r];
                    RayL ray;
                    ray.o = mk(a.x, a.y, a.z), ray.d = mk(a.w, rb.x, rb.y);
                    const V3 v0 = mk(c0.x, c0.y, c0.z), v1 = mk(c0.w, c1.x, c1.y);
                    const V3 v2 = mk(c1.z, c1.w, c2.x), n = mk(c2.y, c2.z, c2.w);
                    float tt;
                    // `lim` lies in [best t so far, initial ray.t]: "<=" keeps equal-t ties (rt_device.h)
                    if (tri_candidate_leaf(v0, v1, v2, n, ray, lim, tt))
                        atomicMin(&L.best[r], make_key(tt, idx, false));
                    if (COUNT)
                        lc.tri++, lc.tri_fetch++;
                }
            }
#ifdef RT_TIMELINE
            lc.tl_leaf_steps++;
#endif
            wave_lds_fence();
        }
        lq_n = 0;
#ifdef RT_STAMPS
        st_leaf += clock64() - tl0, n_leaf++;
#endif
    }

    // the lanes' hit leaf children (up to four each) appended to the leaf queue: one prefix sum over the lanes'
    // counts (three ballots: the bits of a count 0..4) instead of one ballot + prefix per child
    RT_DEV void queue_leaves4(uint32_t leafm, uint32_t r, const uint32_t ref[4])
    {
        const uint32_t c = (uint32_t)__builtin_popcount(leafm); // leafm: bit k = child k is a hit leaf group
        const uint64_t any = ballot(c != 0u);
        if (any == 0)
            return;
        const uint64_t b0 = ballot((c & 1u) != 0u), b1 = ballot((c & 2u) != 0u), b2 = ballot((c & 4u) != 0u);
        uint32_t o = lq_n + prefix_count(b0) + 2u * prefix_count(b1) + 4u * prefix_count(b2);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (leafm & (1u << k)) { // ref: bit 31, count - 1 at 30:29, group at 23:0 -> group | (count - 1) << 24 | ray << 26
                L.wlq[o] = (ref[k] & 0xffffffu) | (((ref[k] >> 29) & 3u) << 24) | (r << 26);
                o++;
            }
        }
        lq_n += (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) + 4u * (uint32_t)__builtin_popcountll(b2);
    }

    // ---- node phase: private deques + work stealing, as in the exact engine -------------------------------
    // Lanes with `seed` start at the root with their own ray. Returns the mask of ray slots for the exact engine.
    RT_DEV uint64_t run(bool seed)
    {
        if (lane < 2u)
            L.wfallback[lane] = 0u;
        wave_lds_fence();
        uint32_t bot = 0, sp = 0;
        bool have = seed;
        uint32_t e = lane; // node 0, ray = lane
        float e_lb = 0.0f;
        uint32_t touched = 0u; // the dword the previous step requested from this step's node (prefetch)
        const bool prefetch = sv.wide_prefetch != 0u;
        for (;;) {
#ifdef RT_STAMPS
            const unsigned long long tn0 = clock64();
#endif
            if (!have && sp > bot) {
                sp--;
                e = L.wstack[sp][lane];
                e_lb = u16_to_lb(L.wstack_lb[sp][lane]);
                if (sp == bot)
                    sp = bot = 0;
                have = true;
            }
            const uint64_t busy = ballot(have);
            if (busy == 0)
                break;
#ifdef RT_TIMELINE
            lc.tl_node_iters++;
#endif
            if (busy != ~0ull) { // idle lanes take the OLDEST deque entry (largest subtree) of a busy lane
                const bool donor = sp > bot;
                const uint64_t donors = ballot(donor);
                if (donors) {
                    const uint64_t idle = ~busy;
                    const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
                    const uint32_t n_don = (uint32_t)__builtin_popcountll(donors);
                    const uint32_t n_x = n_idle < n_don ? n_idle : n_don;
                    if (donor) {
                        const uint32_t k = prefix_count(donors);
                        if (k < n_x) {
                            L.xfer[k] = L.wstack[bot][lane];
                            L.wxfer_lb[k] = L.wstack_lb[bot][lane];
                            bot++;
                            if (sp == bot)
                                sp = bot = 0;
                        }
                    }
                    wave_lds_fence();
                    if (!have) {
                        const uint32_t k = prefix_count(idle);
                        if (k < n_x) {
                            e = L.xfer[k];
                            e_lb = u16_to_lb(L.wxfer_lb[k]);
                            have = true;
                        }
                    }
                    wave_lds_fence();
                }
            }
#ifdef RT_STAMPS
            const unsigned long long tn1 = clock64(); // pop + steal done
            unsigned long long tn2 = tn1, tn3 = tn1;
#endif
            uint32_t leafm = 0u; // bit k: child k is a hit leaf group (one VGPR across the step's branches, not four lane masks)
            uint32_t ref[4] = { 0u, 0u, 0u, 0u };
            uint32_t r = 0;
            if (have) {
                have = false;
                r = e & 63u;
                const bool any = wide_any_hit_slot<MODE>(r);
                const unsigned long long bk = L.best[r];
                const float4 ra = L.rayA[r], rb = L.rayB[r], rc = L.rayC[r];
                const uint32_t t0b = __float_as_uint(rb.z), bb = (uint32_t)(bk >> 32);
                const float lim = __uint_as_float((any || t0b < bb) ? t0b : bb); // shadow rays: t0; else min(t0, best t)
                const bool skip = (any & (bk != kNoHit)) | (!any & (e_lb > lim));
                if (!skip) {
                    const uint32_t noff = (e >> 6) * 128u;
                    const float4 lox = buf_load16(nodes, noff), loy = buf_load16(nodes, noff + 16u);
                    const float4 loz = buf_load16(nodes, noff + 32u), hix = buf_load16(nodes, noff + 48u);
                    const float4 hiy = buf_load16(nodes, noff + 64u), hiz = buf_load16(nodes, noff + 80u);
                    const float4 rf = buf_load16(nodes, noff + 96u);
#if defined(RT_DIAG) && RT_DIAG == 11 // the same seven loads from a second copy of the node array
                    {
                        const float4 d0 = buf_load16(nodes2, noff), d1 = buf_load16(nodes2, noff + 16u), d2 = buf_load16(nodes2, noff + 32u);
                        const float4 d3 = buf_load16(nodes2, noff + 48u), d4 = buf_load16(nodes2, noff + 64u), d5 = buf_load16(nodes2, noff + 80u);
                        const float4 d6 = buf_load16(nodes2, noff + 96u);
                        asm volatile("" ::"v"(d0.x), "v"(d1.y), "v"(d2.z), "v"(d3.w), "v"(d4.x), "v"(d5.y), "v"(d6.z));
                    }
#endif
#if defined(RT_DIAG) && RT_DIAG == 14 // the ray-slot reads once more (the neighbouring slot), result unused
                    {
                        const uint32_t r2 = r ^ 1u;
                        const float4 xa = L.rayA[r2], xb = L.rayB[r2], xc = L.rayC[r2];
                        const unsigned long long xk = L.best[r2];
                        asm volatile("" ::"v"(xa.x), "v"(xb.y), "v"(xc.z), "v"((uint32_t)xk));
                    }
#endif
                    asm volatile("" ::"v"(touched)); // the touch is over by now (loads return in order)
#ifdef RT_STAMPS
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the stamp after the data has arrived
                    tn2 = clock64();
#endif
                    ref[0] = __float_as_uint(rf.x), ref[1] = __float_as_uint(rf.y);
                    ref[2] = __float_as_uint(rf.z), ref[3] = __float_as_uint(rf.w);
                    // t = b * (1/d) - o * (1/d): one FMA per slab plane; s covers its rounding and the box growth rho
                    const float s = rc.w * 1.0625f;
                    const float ox = ra.x * rc.x, oy = ra.y * rc.y, oz = ra.z * rc.z;
                    const float s2 = s + s, ms = -s, lims = lim + s;
                    const float bx0[4] = { lox.x, lox.y, lox.z, lox.w }, bx1[4] = { hix.x, hix.y, hix.z, hix.w };
                    const float by0[4] = { loy.x, loy.y, loy.z, loy.w }, by1[4] = { hiy.x, hiy.y, hiy.z, hiy.w };
                    const float bz0[4] = { loz.x, loz.y, loz.z, loz.w }, bz1[4] = { hiz.x, hiz.y, hiz.z, hiz.w };
                    uint32_t inm = 0u; // bit k: child k is a hit internal node
                    float lbk[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float tx0 = __builtin_fmaf(bx0[k], rc.x, -ox), tx1 = __builtin_fmaf(bx1[k], rc.x, -ox);
                        const float ty0 = __builtin_fmaf(by0[k], rc.y, -oy), ty1 = __builtin_fmaf(by1[k], rc.y, -oy);
                        const float tz0 = __builtin_fmaf(bz0[k], rc.z, -oz), tz1 = __builtin_fmaf(bz1[k], rc.z, -oz);
                        const float tin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
                        const float tout = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
                        // may the ray pass within the slack of the box at a distance in [0, lim]? (positive comparisons:
                        // a NaN — impossible for an all_ok ray and a finite box — would drop the child, never a real one)
                        const bool hit = (tin <= tout + s2) & (tout >= ms) & (tin <= lims) & (ref[k] != kWideEmpty);
                        lbk[k] = fmaxf(tin - s, 0.0f); // a bound below zero says no more than zero does
                        const uint32_t hb = hit ? (1u << k) : 0u;
                        leafm |= wide_is_leaf(ref[k]) ? hb : 0u;
                        inm |= wide_is_leaf(ref[k]) ? 0u : hb;
                    }
#if defined(RT_DIAG) && RT_DIAG == 13 // the four box tests once more (boxes shifted), result unused
                    {
                        float acc = 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float tx0 = __builtin_fmaf(bx0[k] + 1.0f, rc.x, -ox), tx1 = __builtin_fmaf(bx1[k] + 1.0f, rc.x, -ox);
                            const float ty0 = __builtin_fmaf(by0[k] + 1.0f, rc.y, -oy), ty1 = __builtin_fmaf(by1[k] + 1.0f, rc.y, -oy);
                            const float tz0 = __builtin_fmaf(bz0[k] + 1.0f, rc.z, -oz), tz1 = __builtin_fmaf(bz1[k] + 1.0f, rc.z, -oz);
                            const float tin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
                            const float tout = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
                            acc += ((tin <= tout + s2) & (tout >= ms) & (tin <= lims)) ? tin : tout;
                        }
                        asm volatile("" ::"v"(acc));
                    }
#endif
                    if (COUNT) {
                        lc.node_fetch++;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            lc.box += ref[k] != kWideEmpty ? 1u : 0u;
                    }
#ifdef RT_STAMPS
                    tn3 = clock64();
#endif
                    // the nearest internal child becomes the current entry, the others are put aside. Nearest = smallest
                    // packed key (bound's bits with the child number in the two lowest: non-negative floats order as integers)
                    uint32_t nkey = 0xffffffffu;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t key = (__float_as_uint(lbk[k]) & ~3u) | (uint32_t)k;
                        nkey = (inm & (1u << k)) ? (key < nkey ? key : nkey) : nkey;
                    }
                    int near = inm ? (int)(nkey & 3u) : -1;
                    const float near_lb = __uint_as_float(nkey & ~3u); // the bound, two mantissa bits lower: still a lower bound
                    // room for the (up to three) others? otherwise the exact engine traces this ray — its entries here are
                    // dropped, the result is discarded anyway
                    if (sp + 3u > kWideStack && (uint32_t)__builtin_popcount(inm) - (inm ? 1u : 0u) + sp > kWideStack) {
                        flag_fallback(r);
                        near = -1;
                        inm = 0u;
                    }
                    const uint32_t pushm = near >= 0 ? inm & ~(1u << near) : 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (pushm & (1u << k)) {
                            L.wstack[sp][lane] = (ref[k] << 6) | r;
                            L.wstack_lb[sp][lane] = (uint16_t)lb_to_u16(lbk[k]);
                            sp++;
                        }
                    }
                    if (near >= 0) {
                        const uint32_t nref = near == 0 ? ref[0] : (near == 1 ? ref[1] : (near == 2 ? ref[2] : ref[3]));
                        e = (nref << 6) | r;
                        e_lb = near_lb;
                        have = true;
                        if (prefetch) // bring the next node's line towards this CU while the step's bookkeeping runs
                            touched = __builtin_amdgcn_raw_buffer_load_b32(nodes, nref * 128u, 0, 0);
                    }
                }
            }
            queue_leaves4(leafm, r, ref);
#ifdef RT_STAMPS
            {
                const unsigned long long tn4 = clock64();
                st_steal += tn1 - tn0, st_load += tn2 - tn1, st_alu += tn3 - tn2, st_push += tn4 - tn3;
                st_node += tn4 - tn0, n_node++;
            }
#endif
            if (lq_n >= flush_at)
                flush_leaves();
        }
        if (lq_n)
            flush_leaves();
        wave_lds_fence();
        return (uint64_t)L.wfallback[0] | ((uint64_t)L.wfallback[1] << 32);
    }
};

// The winner of ray slot `lane` (L.best) against the reference's own test of the leaf it lives in: true when the
// reference visits that leaf for certain (exact A.4 on the leaf's box passes with tout != 0), i.e. the winner stands.
template <bool COUNT>
RT_DEV bool wide_winner_stands(const SceneView& sv, const RayL& ray, const RayInv& inv, unsigned long long key, LaneCounters& lc)
{
    const uint32_t ref_leaf = sv.wide_ref_leaf[key_index(key)];
    const float4* bp = reinterpret_cast<const float4*>(sv.wide_leaf_box + ref_leaf);
    const float4 b0 = bp[0], b1 = bp[1];
    const float lo[3] = { b0.x, b0.y, b0.z }, hi[3] = { b0.w, b1.x, b1.y };
    float tl[3], th[3];
    slab_quotients(lo, hi, ray, inv, tl, th); // the reference's six quotients, bit for bit
    // A.4 with every quotient finite (all_ok ray, finite box): tin < 0 ? tout > 0 : tin <= tout
    const float tin = sel_max(sel_min(th[2], tl[2]), sel_max(sel_min(th[1], tl[1]), sel_min(th[0], tl[0])));
    const float tout = sel_min(sel_max(th[2], tl[2]), sel_min(sel_max(th[1], tl[1]), sel_max(th[0], tl[0])));
    if (COUNT)
        lc.box++;
    return (tin < 0.0f ? tout > 0.0f : tin <= tout) && tout != 0.0f;
}

// Does triangle `pos` (index in the reference's visit order) block this shadow ray, i.e. is it one of the
// reference's candidates with 0 <= t <= 1? Exactly that when it answers true: the reference's triangle test passes
// (Appendix A.1/A.2, the precomputed plane and edges of PrimRecord) AND the reference visits the triangle's leaf
// (exact box test of its reference leaf, header comment) — then BvhInterface::intersect reports a hit whatever else
// lies along the ray. false = "not known": the caller traverses. Used to try the previous sample's blocker first
// for the many shadow rays of an area light (light.cpp:139-157); any-hit results do not depend on the order in which
// candidates are looked at.
template <bool COUNT>
RT_DEV bool wide_known_blocker(const SceneView& sv, const RayL& ray, uint32_t pos, LaneCounters& lc)
{
    const RayInv inv = make_ray_inv(ray, sv.extent);
    if (!inv.all_ok())
        return false;
    const PrimRecord rec = sv.prims[pos];
    float tt;
    if (COUNT)
        lc.tri++;
    if (!tri_candidate(rec, ray, 1.0f, tt))
        return false;
    return wide_winner_stands<COUNT>(sv, ray, inv, make_key(tt, pos, false), lc);
}

// Ray slots with `fast` set (all_ok rays: the caller checks) are traced through the wide tree (closest hit /
// any hit per slot as MODE says); on return L.best holds their keys, and the returned mask names the slots whose
// result must be discarded and traced by the exact engine instead. `ray` / `inv`: the lane's own ray (slot = lane).
// ALL lanes of the wave must call.
template <int MODE, bool COUNT>
RT_DEV uint64_t traverse_wide(const SceneView& sv, bool fast, const RayL& ray, const RayInv& inv, LaneCounters& lc,
    WaveLds& L, uint32_t lane)
{
    uint32_t flush_at = sv.leaf_flush < kWideFlushMax ? sv.leaf_flush : kWideFlushMax;
    WideWalk<MODE, COUNT> w { sv, L, lc, lane, make_rsrc(sv.wide_nodes, sv.wide_n_nodes * 128u),
        make_rsrc(sv.wide_leaf, sv.wide_n_groups * 4u * 48u), make_rsrc(sv.wide_index, sv.wide_n_groups * 4u * 4u),
#ifdef RT_DIAG
        make_rsrc(sv.diag_wnodes, sv.wide_n_nodes * 128u), make_rsrc(sv.diag_wleaf, sv.wide_n_groups * 4u * 48u),
#endif
        0u, flush_at < 1u ? 1u : flush_at };
#ifdef RT_STAMPS
    const unsigned long long stamp_begin = clock64();
#endif
    uint64_t back = w.run(fast);
    const unsigned long long key = L.best[lane];
    bool redo = false;
    if (fast && key != kNoHit)
        redo = !wide_winner_stands<COUNT>(sv, ray, inv, key, lc);
#ifdef RT_STAMPS
    if (sv.stamps && lane == 0) { // slots 32..: the wide engine (closest / mixed: 32, any hit: 40, node-step split: 48)
        const unsigned long long total = clock64() - stamp_begin;
        const int base = MODE == 1 ? 40 : 32;
        atomicAdd(&sv.stamps[base + 0], w.st_steal);
        atomicAdd(&sv.stamps[base + 1], w.st_node);
        atomicAdd(&sv.stamps[base + 2], w.st_leaf);
        atomicAdd(&sv.stamps[base + 3], total);
        atomicAdd(&sv.stamps[base + 4], w.n_node);
        atomicAdd(&sv.stamps[base + 5], w.n_leaf);
        atomicAdd(&sv.stamps[base + 6], 1ull);
        atomicMax(&sv.stamps[base + 7], total);
        atomicAdd(&sv.stamps[48], w.st_load);
        atomicAdd(&sv.stamps[49], w.st_alu);
        atomicAdd(&sv.stamps[50], w.st_push);
        atomicAdd(&sv.stamps[51], w.n_node);
    }
#endif
    return back | ballot(redo);
}

} // namespace rt
