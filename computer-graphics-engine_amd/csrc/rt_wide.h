// rt_wide.h — the wide engine: traversal of our own 8-wide SAH tree (rt_layout.h WideNode, host_wide.cpp)
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
//   * walks the wide tree with a CONSERVATIVE box test: the children's bounds are 8-bit offsets inside the node's
//     box, rounded outward (rt_layout.h WideNode), one conversion + one FMA per slab plane, the interval widened
//     by the pruning slack (rt_device.h "Pruning": a candidate's point lies within rho = 2^-16 x scene scale of
//     its triangle, hence of every box around it; the roundings of the FMA form stay below 1/32 of the slack),
//     pruned by min(initial ray.t, best t so far) like the exact engine;
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
    uint32_t lq_n = 0; // wave-uniform fill of the leaf queue
    uint32_t flush_at;

    RT_DEV void flag_fallback(uint32_t r) { atomicOr(&L.wfallback[r >> 5], 1u << (r & 31u)); }

    // ---- leaf phase: four lanes per (ray, leaf group) pair, one triangle per lane ------------------------
    RT_DEV void flush_leaves()
    {
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
    }

    // Hit leaf children k0 .. k0 + 3 of every lane (`four`: their hit bits) appended to the leaf queue: one prefix sum
    // over the lanes' counts (three ballots: the bits of a count 0..4) instead of one ballot + prefix per child.
    // Child k is the (leaf children before k)-th group from base_group; entry: group | (triangles - 1) << 24 | ray << 26.
    RT_DEV void queue_leaves4(uint32_t four, int k0, uint32_t r, uint32_t base_group, uint32_t counts, uint32_t leaf_mask)
    {
        const uint32_t c = (uint32_t)__builtin_popcount(four);
        const uint64_t any = ballot(c != 0u);
        if (any == 0)
            return;
        const uint64_t b0 = ballot((c & 1u) != 0u), b1 = ballot((c & 2u) != 0u), b2 = ballot((c & 4u) != 0u);
        uint32_t o = lq_n + prefix_count(b0) + 2u * prefix_count(b1) + 4u * prefix_count(b2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((four >> j) & 1u) {
                const int k = k0 + j;
                const uint32_t group = (base_group & 0xffffffu) + (uint32_t)__builtin_popcount(leaf_mask & ((1u << k) - 1u));
                L.wlq[o] = group | (((counts >> (2 * k)) & 3u) << 24) | (r << 26);
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
            uint32_t leaves = 0u;       // mask of this lane's hit leaf children
            uint32_t base_group = 0u, counts = 0u, leaf_mask = 0u;
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
                    const float4 w0 = buf_load16(nodes, noff), w1 = buf_load16(nodes, noff + 16u), w2 = buf_load16(nodes, noff + 32u);
                    const float4 w3 = buf_load16(nodes, noff + 48u), w4 = buf_load16(nodes, noff + 64u);
                    asm volatile("" ::"v"(touched)); // the touch is over by now (loads return in order)
                    // Child plane: b = origin + q * 2^(exp - 127); t = b * (1/d) - o * (1/d) = q * (scale / d) + (origin - o) / d:
                    // one conversion and one FMA per plane. s covers the roundings (each below 2^-21 x scene scale x |1/d|,
                    // 1/32 of the slack) and the box growth rho.
                    const float s = rc.w * 1.0625f;
                    const uint32_t ex = __float_as_uint(w0.w);
                    const float sx = __uint_as_float((ex & 0xffu) << 23) * rc.x, sy = __uint_as_float(((ex >> 8) & 0xffu) << 23) * rc.y;
                    const float sz = __uint_as_float(((ex >> 16) & 0xffu) << 23) * rc.z;
                    const float ox = (w0.x - ra.x) * rc.x, oy = (w0.y - ra.y) * rc.y, oz = (w0.z - ra.z) * rc.z;
                    const float s2 = s + s, ms = -s, lims = lim + s;
                    // qlo x: w1.x w1.y, y: w1.z w1.w, z: w2.x w2.y; qhi x: w2.z w2.w, y: w3.x w3.y, z: w3.z w3.w
                    const uint32_t qx0[2] = { __float_as_uint(w1.x), __float_as_uint(w1.y) }, qy0[2] = { __float_as_uint(w1.z), __float_as_uint(w1.w) };
                    const uint32_t qz0[2] = { __float_as_uint(w2.x), __float_as_uint(w2.y) }, qx1[2] = { __float_as_uint(w2.z), __float_as_uint(w2.w) };
                    const uint32_t qy1[2] = { __float_as_uint(w3.x), __float_as_uint(w3.y) }, qz1[2] = { __float_as_uint(w3.z), __float_as_uint(w3.w) };
                    const uint32_t base_node = __float_as_uint(w4.x), int_mask = ex >> 24;
                    base_group = __float_as_uint(w4.y), counts = __float_as_uint(w4.z);
                    leaf_mask = base_group >> 24;
                    auto byte_f = [](uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xffu); }; // v_cvt_f32_ubyteK
                    uint32_t hits = 0u; // children whose (widened) box the ray may pass within [0, lim]
                    float lbk[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int w = k >> 2, b = k & 3;
                        const float tx0 = __builtin_fmaf(byte_f(qx0[w], b), sx, ox), tx1 = __builtin_fmaf(byte_f(qx1[w], b), sx, ox);
                        const float ty0 = __builtin_fmaf(byte_f(qy0[w], b), sy, oy), ty1 = __builtin_fmaf(byte_f(qy1[w], b), sy, oy);
                        const float tz0 = __builtin_fmaf(byte_f(qz0[w], b), sz, oz), tz1 = __builtin_fmaf(byte_f(qz1[w], b), sz, oz);
                        const float tin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
                        const float tout = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
                        // positive comparisons: a NaN — impossible for an all_ok ray and a finite box — would drop the
                        // child, never a real one; empty children are in neither kind mask
                        const bool hit = (tin <= tout + s2) & (tout >= ms) & (tin <= lims);
                        hits |= hit ? (1u << k) : 0u;
                        lbk[k] = tin - s;
                    }
                    leaves = hits & leaf_mask;
                    uint32_t inner = hits & int_mask;
                    if (COUNT) {
                        lc.node_fetch++;
                        lc.box += (uint32_t)__builtin_popcount(int_mask | leaf_mask);
                    }
                    // the nearest internal child becomes the current entry, the others are put aside
                    int near = -1;
                    float near_lb = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const bool better = ((inner >> k) & 1u) & ((near < 0) | (lbk[k] < near_lb));
                        near = better ? k : near;
                        near_lb = better ? lbk[k] : near_lb;
                    }
                    // room for the (up to seven) others? otherwise the exact engine traces this ray — its entries here
                    // are dropped, the result is discarded anyway
                    const uint32_t n_push = (uint32_t)__builtin_popcount(inner) - (near >= 0 ? 1u : 0u);
                    if (sp + n_push > kWideStack) {
                        flag_fallback(r);
                        near = -1;
                        inner = 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (((inner >> k) & 1u) & (k != near)) {
                            const uint32_t node = base_node + (uint32_t)__builtin_popcount(int_mask & ((1u << k) - 1u));
                            L.wstack[sp][lane] = (node << 6) | r;
                            L.wstack_lb[sp][lane] = (uint16_t)lb_to_u16(lbk[k]);
                            sp++;
                        }
                    }
                    if (near >= 0) {
                        const uint32_t nnode = base_node + (uint32_t)__builtin_popcount(int_mask & ((1u << near) - 1u));
                        e = (nnode << 6) | r;
                        e_lb = near_lb;
                        have = true;
                        if (prefetch) // bring the next node's line towards this CU while the step's bookkeeping runs
                            touched = __builtin_amdgcn_raw_buffer_load_b32(nodes, nnode * 128u, 0, 0);
                    }
                }
            }
            // the hit leaf children join the leaf queue, four child slots at a time (capacity: 256 new entries + a
            // remainder below the flush threshold)
            queue_leaves4(leaves & 15u, 0, r, base_group, counts, leaf_mask);
            if (lq_n >= flush_at)
                flush_leaves();
            queue_leaves4(leaves >> 4, 4, r, base_group, counts, leaf_mask);
            if (lq_n >= flush_at)
                flush_leaves();
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
        make_rsrc(sv.wide_leaf, sv.wide_n_groups * 4u * 48u), make_rsrc(sv.wide_index, sv.wide_n_groups * 4u * 4u), 0u,
        flush_at < 1u ? 1u : flush_at };
    uint64_t back = w.run(fast);
    const unsigned long long key = L.best[lane];
    bool redo = false;
    if (fast && key != kNoHit)
        redo = !wide_winner_stands<COUNT>(sv, ray, inv, key, lc);
    return back | ballot(redo);
}

} // namespace rt
