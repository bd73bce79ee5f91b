// rt_traverse.h — BVH traversal of one wave's rays (closest hit / shadow query).
//
// Building blocks and the order-free reduction argument: rt_device.h.
//
//  NODE phase. Every lane is a worker with a private deque of (ray slot, node
//    pair) entries in LDS. A worker pops its newest entry, gathers the 64-byte
//    NodePair, runs the two slab tests for that ray and pushes the internal
//    children it must descend into (nearer child on top). A lane whose deque is
//    empty takes the OLDEST entry (the largest pending subtree) of a lane that
//    holds at least two, so lanes that carry no ray of their own (sparse
//    reflection levels, finished rays) pick up work at once and a wave's time is
//    its total work / 64 rather than its longest ray.
//    (A wave-level "packet" walk with scalar node loads was measured too: on the
//    819 200-triangle bench scene it is 2.5x slower — its visits form one serial
//    chain per wave — and was dropped; profiles/r01_notes.md.)
//  LEAF phase. Leaf children are parked as (ray, leaf) pairs in a per-wave queue. A flush
//    first tests the ray against the bounds of the leaf's GROUPS of four triangles (eight
//    lanes per pair, conservative, pruned policy only), then runs the exact triangle test
//    on the surviving groups, four lanes each (rt_layout.h GroupRecord).
#pragma once
#include "rt_device.h"
#include "rt_wide.h"

namespace rt {

// Query mode of a traversal: closest hit, any hit (shadow query), or MIXED — ray slots 0..31
// closest, slots 32..63 any hit (k_trace_level traces a level's shadow rays together with the
// next level's reflection rays).
enum : int { Q_CLOSEST = 0, Q_ANY = 1, Q_MIXED = 2 };
template <int MODE>
RT_DEV bool any_hit_slot(uint32_t r)
{
    return MODE == Q_ANY || (MODE == Q_MIXED && r >= 32u);
}

// ---- leaf phase -----------------------------------------------------------------
#if defined(RT_DIAG) && RT_DIAG == 2
__device__ const float4* g_diag_leaf;
#endif

// the three 16-byte words of LeafRecord number `slot`
RT_DEV void load_leaf(BufRsrc leaf, uint32_t slot, float4 c[3])
{
    const uint32_t off = slot * 48u;
#pragma unroll
    for (int k = 0; k < 3; k++)
        c[k] = buf_load16(leaf, off + 16u * k);
}

// one primitive against ray slot r; `index` = its position in the reference's order (key, report)
template <bool COUNT>
RT_DEV void leaf_test_one(WaveLds& L, const float4 c[3], uint32_t r, uint32_t index, bool spheres, LaneCounters& lc)
{
    const float4 a = L.rayA[r], b = L.rayB[r];
    RayL ray;
    ray.o = mk(a.x, a.y, a.z);
    ray.d = mk(a.w, b.x, b.y);
    const float t0 = b.z;
    const V3 v0 = mk(c[0].x, c[0].y, c[0].z), v1 = mk(c[0].w, c[1].x, c[1].y);
    const V3 v2 = mk(c[1].z, c[1].w, c[2].x), n = mk(c[2].y, c[2].z, c[2].w);
    float tt;
    bool cand;
    // `spheres` is wave-uniform (the scene holds any); the tag is n == 0 (rt_layout.h)
    const bool sphere = spheres && n.x == 0.0f && n.y == 0.0f && n.z == 0.0f;
    if (!sphere) {
        cand = tri_candidate_leaf(v0, v1, v2, n, ray, t0, tt);
        if (COUNT)
            lc.tri++;
    } else {
        cand = sphere_candidate(v0, v1.x, ray, t0, tt);
        if (COUNT)
            lc.sph++;
    }
    if (COUNT)
        lc.tri_fetch++;
    if (cand)
        atomicMin(&L.best[r], make_key(tt, index, sphere));
}

// Ungrouped range [ref, ref + cnt) of `leaf` against ray slot r, the whole wave on it, one
// primitive per lane: leaves above kGroupLeafMax primitives, scenes with spheres, a root that
// is a leaf, and the brute-force mode (enableAccelStructure off: the whole array).
template <int MODE, bool COUNT>
RT_DEV void flat_range(WaveLds& L, BufRsrc leaf, bool spheres, uint32_t r, uint32_t ref, uint32_t cnt, uint32_t lane,
    LaneCounters& lc)
{
    for (uint32_t base = 0; base < cnt; base += 64u) {
        if (any_hit_slot<MODE>(r) && L.best[r] != kNoHit)
            break;
        const uint32_t i = base + lane;
        if (i < cnt) {
            float4 c[3];
            load_leaf(leaf, ref + i, c);
            leaf_test_one<COUNT>(L, c, r, ref + i, spheres, lc);
        }
        wave_lds_fence();
    }
}

// Conservative ray / group-box overlap: may the ray pass within the pruning slack of the box
// at a distance in [0, lim]? Quotients by reciprocal multiplication (<= 1.5 ulp off), every
// bound moved outward by 2^-20 relative and by the slack; any NaN compares false = "keep".
RT_DEV bool group_may_hit(float4 g0, float4 g1, float4 ra, float4 rb, float4 rc, float lim)
{
    const float tl0 = (g0.x - ra.x) * rc.x, th0 = (g0.w - ra.x) * rc.x;
    const float tl1 = (g0.y - ra.y) * rc.y, th1 = (g1.x - ra.y) * rc.y;
    const float tl2 = (g0.z - ra.z) * rc.z, th2 = (g1.y - ra.z) * rc.z;
    const float tin = fmaxf(fmaxf(fminf(tl0, th0), fminf(tl1, th1)), fminf(tl2, th2));
    const float tout = fminf(fminf(fmaxf(tl0, th0), fmaxf(tl1, th1)), fmaxf(tl2, th2));
    const float lo = tin - fabsf(tin) * 9.5367431640625e-07f - rc.w;
    const float hi = tout + fabsf(tout) * 9.5367431640625e-07f + rc.w;
    return !(lo > hi) && !(hi < 0.0f) && !(lo > lim);
}

// Entries [first, first + n) of the wave's leaf queue (grouped leaves only).
//  Stage A: 8 lanes per (ray, leaf) pair, one GROUP of four triangles per lane: conservative
//           box test (pruned policy), survivors appended to an LDS list together with the
//           group's four ranks.
//  Stage B: 4 lanes per surviving group, one triangle per lane: the reference's exact test.
template <int MODE, bool COUNT>
RT_DEV void leaf_phase(WaveLds& L, BufRsrc leafg, BufRsrc groups, uint32_t first, uint32_t n, uint32_t lane, bool prune,
    LaneCounters& lc)
{
    auto stage_b = [&](uint32_t base, uint32_t take) {
        const uint32_t s = lane >> 2, j = lane & 3u;
        if (s < take) {
            const uint32_t e = L.surv[base + s];
            const uint32_t rank = (e >> (11u + 5u * j)) & 31u;
            const uint32_t pi = (e >> 3) & 0xffu, g = e & 7u;
            const uint32_t refw = L.lq_ref[pi], cw = L.lq_cw[pi];
            const uint32_t r = refw >> 26;
            const bool done = any_hit_slot<MODE>(r) & (L.best[r] != kNoHit); // read for every slot: no divergent branch
            if (rank != 31u && !done) {
                float4 c[3];
                load_leaf(leafg, ((cw >> 6) + g) * 4u + j, c);
#if defined(RT_DIAG) && RT_DIAG == 2 // the same three loads from a second copy of the leaf array
                {
                    float4 dd[3];
                    load_leaf(make_rsrc(g_diag_leaf, 0x7fffffffu), ((cw >> 6) + g) * 4u + j, dd);
                    asm volatile("" ::"v"(dd[0].x), "v"(dd[1].y), "v"(dd[2].z));
                }
#endif
                leaf_test_one<COUNT>(L, c, r, (refw & 0x3ffffffu) + rank, false, lc);
            }
        }
#ifdef RT_TIMELINE
        lc.tl_leaf_steps++;
#endif
        wave_lds_fence();
    };
    uint32_t fill = 0;
    for (uint32_t k = 0; k < n; k += 8u) {
        const uint32_t p = k + (lane >> 3), g = lane & 7u;
        bool alive = false;
        uint32_t entry = 0;
        if (p < n) {
            const uint32_t refw = L.lq_ref[first + p], cw = L.lq_cw[first + p];
            const float lb = L.lq_lb[first + p];
            const uint32_t r = refw >> 26;
            // one read of the ray's state, the same for closest-hit and any-hit slots (no divergent branch)
            const bool any = any_hit_slot<MODE>(r);
            const float4 rb = L.rayB[r];
            const unsigned long long bk = L.best[r];
            const uint32_t t0b = __float_as_uint(rb.z), bb = (uint32_t)(bk >> 32);
            const float lim = __uint_as_float((any || t0b < bb) ? t0b : bb); // ray_limit(), t0 for shadow rays
            const bool skip = (any & (bk != kNoHit)) | (!any & prune & (lb > lim));
            if (g * 4u < (cw & 63u) && !skip) {
                const uint32_t off = ((cw >> 6) + g) * 32u;
                const float4 g0 = buf_load16(groups, off), g1 = buf_load16(groups, off + 16u);
                alive = true;
                if (prune) {
                    if (__float_as_uint(rb.w) & 8u) // reciprocals usable on all three axes, finite origin
                        alive = group_may_hit(g0, g1, L.rayA[r], rb, L.rayC[r], lim);
                    if (COUNT)
                        lc.grp++;
                }
                entry = ((first + p) << 3) | g | (__float_as_uint(g1.z) << 11);
            }
        }
        const uint64_t m = ballot(alive);
        if (alive)
            L.surv[fill + prefix_count(m)] = entry;
        fill += (uint32_t)__builtin_popcountll(m);
#ifdef RT_TIMELINE
        lc.tl_leaf_steps++;
#endif
        wave_lds_fence();
        while (fill >= 16u) {
            fill -= 16u;
            stage_b(fill, 16u);
        }
    }
    if (fill)
        stage_b(0u, fill);
}

// State shared by the engines of one traverse() call.
template <int MODE, bool COUNT>
struct Walk {
    const SceneView& sv;
    WaveLds& L;
    BufRsrc leaf, nodes; // ungrouped LeafRecord array (visit or array order) / NodePair array
    BufRsrc leafg, groups; // grouped LeafRecord array / GroupRecord array
    bool spheres; // the scene holds spheres (wave-uniform)
    LaneCounters& lc;
    uint32_t lane;
    bool prune;
    uint32_t lq_n; // leaf-queue fill (wave-uniform)
    // the current entry's ray (load_ray)
    RayL cur;
    RayInv cur_inv;
#ifdef RT_STAMPS
    unsigned long long st_node = 0, st_leaf = 0, st_steal = 0, n_node = 0, n_leaf = 0;
    unsigned long long st_pop = 0, st_load = 0, st_alu = 0, st_push = 0; // finer split of the node step
#endif

    // Ray slot r of the LDS ray table into registers; returns the bits of its initial limit t0. Read on every
    // node step: skipping the reads when the lane stays on one ray costs more (a divergent branch) than they do.
    RT_DEV uint32_t load_ray(uint32_t r)
    {
        const float4 a = L.rayA[r], b = L.rayB[r], c = L.rayC[r];
        cur.o = mk(a.x, a.y, a.z);
        cur.d = mk(a.w, b.x, b.y);
        cur_inv.y[0] = c.x, cur_inv.y[1] = c.y, cur_inv.y[2] = c.z, cur_inv.slack = c.w;
        cur_inv.guard = __float_as_uint(b.w);
        return __float_as_uint(b.z);
    }

    // append the lanes' (ray r, leaf ref / count word, lb) pairs for which `want` is set (grouped and
    // ungrouped leaves alike: flush_leaves tells them apart)
    RT_DEV void queue_leaves(bool want, uint32_t r, uint32_t ref, uint32_t cw, float lb)
    {
        const uint64_t m = ballot(want);
        if (m) {
            if (want) {
                const uint32_t o = lq_n + prefix_count(m);
                L.lq_ref[o] = ref | (r << 26);
                L.lq_cw[o] = cw;
                L.lq_lb[o] = lb;
            }
            lq_n += (uint32_t)__builtin_popcountll(m);
        }
    }

    // Ungrouped leaves among the queue's entries (rt_layout.h count word: leaves above kGroupLeafMax
    // primitives, scenes with spheres, a root that is a leaf, the brute-force mode): the whole wave on each,
    // one primitive per lane. Kept out of the node loop — most scenes have none.
    RT_DEV void flush_ungrouped()
    {
        for (uint32_t base = 0; base < lq_n; base += 64u) {
            const uint32_t i = base + lane;
            uint64_t big = ballot(i < lq_n && cw_is_big(L.lq_cw[i]));
            while (big) {
                const uint32_t k = base + (uint32_t)__builtin_ctzll(big);
                big &= big - 1;
                const uint32_t refw = L.lq_ref[k], rr = refw >> 26;
                if (any_hit_slot<MODE>(rr) ? (L.best[rr] != kNoHit) : (prune && L.lq_lb[k] > ray_limit(L, rr)))
                    continue;
                flat_range<MODE, COUNT>(L, leaf, spheres, rr, refw & 0x3ffffffu, cw_big_count(L.lq_cw[k]), lane, lc);
            }
        }
    }

    // Flush early: a leaf step keeps all 64 lanes busy with any two entries, and the
    // sooner a leaf is tested the sooner its hit prunes the rest of the tree.
    RT_DEV void flush_leaves(uint32_t threshold)
    {
        if (lq_n >= threshold && lq_n > 0u) {
#ifdef RT_STAMPS
            const unsigned long long ts = clock64();
#endif
            wave_lds_fence();
            flush_ungrouped(); // the grouped phases below see no group in those entries (count word bits 5:0 = 0)
            while (lq_n > 0u) {
                const uint32_t n = lq_n < 64u ? lq_n : 64u;
                lq_n -= n;
                leaf_phase<MODE, COUNT>(L, leafg, groups, lq_n, n, lane, prune, lc);
            }
            wave_lds_fence();
#ifdef RT_STAMPS
            st_leaf += clock64() - ts, n_leaf++;
#endif
        }
    }

    // Box tests of one node pair for ray slot r (already cached): returns which
    // children to descend into, with their lower bounds.
    // `lim`: the ray's pruning limit (shadow rays: the initial limit t0 = 1; else min(t0, best t so far))
    RT_DEV void test_pair(const float llo[3], const float lhi[3], const float rlo[3], const float rhi[3], float lim,
        bool& hl, bool& hr, float& lb_l, float& lb_r)
    {
        box_pair_hit(llo, lhi, rlo, rhi, cur, cur_inv, sv.bounds_finite != 0u, hl, hr, lb_l, lb_r);
        if (COUNT) {
            lc.box += 2;
            lc.node_fetch++;
        }
        if (prune) {
            hl = hl & !(lb_l > lim);
            hr = hr & !(lb_r > lim);
        }
    }

    // ---- LANE engine: private deques in rows [row0, ...) of L.stack, work stealing ----
    // On entry lanes with `seed` hold one entry (node pair `node`, own ray, bound lb).
    RT_DEV void lane_engine(uint32_t row0, bool seed, uint32_t node, float seed_lb)
    {
        // The lane's CURRENT entry lives in a register; its deque in LDS (rows row0 + [bot .. sp))
        // holds only what it has put aside (far children) — that is what thieves take.
        uint32_t bot = 0, sp = 0;
        bool have = seed;
        uint32_t e = seed ? pack_entry(node, lane, seed_lb) : 0u;
        for (;;) {
            // lanes without a current entry take the newest one of their own deque
            if (!have && sp > bot) {
                sp--;
                e = L.stack[row0 + sp][lane];
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
#ifdef RT_STAMPS
            const unsigned long long ts0 = clock64();
#endif
            if (busy != ~0ull) { // steal: idle lanes take the OLDEST deque entry (largest subtree) of a busy lane
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
                            L.xfer[k] = L.stack[row0 + bot][lane];
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
                            have = true;
                        }
                    }
                    wave_lds_fence();
                }
            }
#ifdef RT_STAMPS
            const unsigned long long ts1 = clock64();
            st_steal += ts1 - ts0;
#endif
            bool leaf_l = false, leaf_r = false;
            uint32_t l_ref = 0, l_cnt = 0, r_ref = 0, r_cnt = 0, r = 0;
            float lb_l = 0.0f, lb_r = 0.0f;
            if (have) {
                have = false;
                const float e_lb = entry_lb(e);
                r = entry_ray(e);
                // one read of the ray's state serves the skip test and the pruning of the children
                const bool any = any_hit_slot<MODE>(r);
                const unsigned long long bk = L.best[r];
                const uint32_t t0b = load_ray(r), bb = (uint32_t)(bk >> 32);
                const float lim = __uint_as_float((any || t0b < bb) ? t0b : bb); // ray_limit(), t0 for shadow rays
                const bool skip = (any & (bk != kNoHit)) | (!any & prune & (e_lb > lim));
                if (!skip) {
                    const uint32_t noff = entry_node(e) * 64u;
                    const float4 q0 = buf_load16(nodes, noff);       // l_lo xyz, l_hi x
                    const float4 q1 = buf_load16(nodes, noff + 16u); // l_hi yz, r_lo xy
                    const float4 q2 = buf_load16(nodes, noff + 32u); // r_lo z, r_hi xyz
                    const float4 q3f = buf_load16(nodes, noff + 48u);
                    const uint4 q3 = make_uint4(__float_as_uint(q3f.x), __float_as_uint(q3f.y), __float_as_uint(q3f.z),
                        __float_as_uint(q3f.w));
#if defined(RT_DIAG) && RT_DIAG == 1 // the same four loads from a second copy of the node array
                    {
                        const float4* vp = reinterpret_cast<const float4*>(sv.diag_nodes + entry_node(e));
                        const float4 d0 = vp[0], d1 = vp[1], d2 = vp[2], d3 = vp[3];
                        asm volatile("" ::"v"(d0.x), "v"(d1.y), "v"(d2.z), "v"(d3.w));
                    }
#endif
                    const float llo[3] = { q0.x, q0.y, q0.z }, lhi[3] = { q0.w, q1.x, q1.y };
                    const float rlo[3] = { q1.z, q1.w, q2.x }, rhi[3] = { q2.y, q2.z, q2.w };
                    bool hl, hr;
                    test_pair(llo, lhi, rlo, rhi, lim, hl, hr, lb_l, lb_r);
#if defined(RT_DIAG) && RT_DIAG == 3 // the slab arithmetic once more (children swapped), result unused
                    {
                        bool xl, xr;
                        float yl, yr;
                        const float a0[3] = { rlo[0] + 1.0f, rlo[1], rlo[2] }, a1[3] = { rhi[0] + 1.0f, rhi[1], rhi[2] };
                        const float b0[3] = { llo[0] + 1.0f, llo[1], llo[2] }, b1[3] = { lhi[0] + 1.0f, lhi[1], lhi[2] };
                        xl = box_hit(a0, a1, cur, cur_inv, yl);
                        xr = box_hit(b0, b1, cur, cur_inv, yr);
                        asm volatile("" ::"v"(yl), "v"(yr), "v"((int)xl), "v"((int)xr));
                    }
#endif
                    l_ref = q3.x, l_cnt = q3.y, r_ref = q3.z, r_cnt = q3.w;
                    const bool push_l = hl && l_cnt == 0u, push_r = hr && r_cnt == 0u;
                    // the nearer child becomes the current entry (finds close hits early), the
                    // farther one is put aside
                    const uint32_t e_l = pack_entry(l_ref, r, lb_l), e_r = pack_entry(r_ref, r, lb_r);
                    const bool both = push_l && push_r, l_far = lb_l > lb_r;
                    if (both) {
                        L.stack[row0 + sp][lane] = l_far ? e_l : e_r;
                        sp++;
                    }
                    if (push_l || push_r) {
                        e = both ? (l_far ? e_r : e_l) : (push_l ? e_l : e_r);
                        have = true;
                    }
                    leaf_l = hl && l_cnt != 0u; // l_cnt / r_cnt: count words (rt_layout.h)
                    leaf_r = hr && r_cnt != 0u;
                }
            }
            queue_leaves(leaf_l, r, l_ref, l_cnt, lb_l);
            queue_leaves(leaf_r, r, r_ref, r_cnt, lb_r);
#ifdef RT_STAMPS
            st_node += clock64() - ts1, n_node++;
#endif
            flush_leaves(sv.leaf_flush);
        }
    }
};

// ---- traversal ------------------------------------------------------------------
// MODE Q_CLOSEST: closest hit ≙ BoundingVolumeHierarchy::intersect's traversal
//   (bvh.cpp:299-362) + getIntersecting (:272-293): `t` holds ray.t in/out, `pos`
//   the array index of the reported primitive (-1: none).
// MODE Q_ANY: shadow query (light.cpp:61-67 only uses the boolean): `t` is the
//   limit (1.0), pos >= 0 iff something blocks the ray.
// `active` selects the lanes that carry a ray; ALL lanes of the wave must call.
// use_bvh = features.enableAccelStructure; when false the whole array is one leaf
// in the reference's own order (bvh.cpp:303-305; sv.leaf_flat).
// NO_SPHERES: the caller knows the scene holds triangles only (lean kernel variant): the sphere
// test is not compiled in.
template <int MODE, bool COUNT, bool NO_SPHERES = false>
RT_DEV void traverse(const SceneView& sv, bool use_bvh, bool prune, bool active, const RayL& ray, float& t,
    int32_t& pos, LaneCounters& lc, WaveLds& L)
{
    pos = -1;
    const uint64_t act = ballot(active);
    if (act == 0)
        return;
#ifdef RT_TIMELINE
    lc.tl_traversals++;
#endif
    const uint32_t lane = lane_id();
    prune = prune && use_bvh && sv.prune_ok != 0u;
    L.rayA[lane] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
    const RayInv inv = make_ray_inv(ray, sv.extent);
    L.rayB[lane] = make_float4(ray.d.y, ray.d.z, t, __uint_as_float(inv.guard));
    L.rayC[lane] = make_float4(inv.y[0], inv.y[1], inv.y[2], inv.slack);
    L.best[lane] = kNoHit;
    Walk<MODE, COUNT> w { sv, L, make_rsrc(use_bvh ? sv.leaf : sv.leaf_flat, sv.n_prims * 48u),
        make_rsrc(sv.nodes, sv.n_pairs * 64u), make_rsrc(sv.leaf_grouped, sv.n_groups * 4u * 48u),
        make_rsrc(sv.groups, sv.n_groups * 32u), !NO_SPHERES && sv.prune_ok == 0u, lc, lane, prune, 0u, ray, inv };
#ifdef RT_STAMPS
    const unsigned long long stamp_begin = clock64();
#endif

    if (!use_bvh || sv.root_cnt != 0u) { // the root is a leaf / brute force: one ungrouped range per ray
        if (COUNT && active)
            lc.exact_rays++;
        w.queue_leaves(active, lane, use_bvh ? sv.root_ref : 0u, use_bvh ? sv.root_cnt : cw_make_big(sv.n_prims), 0.0f);
    } else if (prune && sv.wide_ok != 0u) {
        // The wide engine (rt_wide.h) for the rays it is proven for; the exact-topology engine for the others
        // (zero / tiny / huge / NaN components) and for those the wide engine hands back.
        const bool fast = active && inv.all_ok();
        const uint64_t back = traverse_wide<MODE, COUNT>(sv, fast, ray, inv, lc, L, lane);
        const bool slow = active && (!fast || ((back >> lane) & 1ull) != 0ull);
        if (COUNT && slow)
            lc.exact_rays++;
        if (ballot(slow)) {
            if (slow)
                L.best[lane] = kNoHit;
            wave_lds_fence();
            w.lane_engine(0u, slow, sv.root_ref, 0.0f);
        }
    } else { // the root's own box is never tested (bvh.cpp:313)
        if (COUNT && active)
            lc.exact_rays++;
        w.lane_engine(0u, active, sv.root_ref, 0.0f);
    }
    w.flush_leaves(1u);
    wave_lds_fence();
    const unsigned long long key = L.best[lane];
    if (active && key != kNoHit) {
        pos = (int32_t)key_index(key);
        if (!any_hit_slot<MODE>(lane)) { // the reported primitive's own t, with its exact bits (-0.0f included)
            const PrimRecord rec = (use_bvh ? sv.prims : sv.prims_flat)[pos];
            float tt;
            if (!NO_SPHERES && (rec.kind_mat & 0x80000000u))
                sphere_candidate(rec, ray, t, tt);
            else
                tt = tri_plane_t(rec, ray);
            t = tt;
        }
    }
    wave_lds_fence(); // the caller may reuse L for the next query
#ifdef RT_STAMPS
    if (sv.stamps && lane == 0) {
        const unsigned long long total = clock64() - stamp_begin;
        const int base = MODE == Q_ANY ? 8 : 0;
        atomicAdd(&sv.stamps[base + 0], w.st_steal);
        atomicAdd(&sv.stamps[base + 1], w.st_node);
        atomicAdd(&sv.stamps[base + 2], w.st_leaf);
        atomicAdd(&sv.stamps[base + 3], total);
        atomicAdd(&sv.stamps[base + 4], w.n_node);
        atomicAdd(&sv.stamps[base + 5], w.n_leaf);
        atomicAdd(&sv.stamps[base + 6], 1ull);
        atomicMax(&sv.stamps[base + 7], total);
        if (MODE != Q_ANY) {
            atomicAdd(&sv.stamps[16], w.st_pop);
            atomicAdd(&sv.stamps[17], w.st_load);
            atomicAdd(&sv.stamps[18], w.st_alu);
            atomicAdd(&sv.stamps[19], w.st_push);
        }
    }
#endif
}

// Two queries in one traversal, packed by ray SLOT: up to 32 closest-hit rays in slots 0..31 and up to 32 any-hit
// (shadow) rays in slots 32..63, whatever lanes they come from — a lane may bring one of each (its next-level
// reflection ray and its shadow ray). k_trace_level uses it for a level's first shadow rays + the next level's
// reflection rays whenever both kinds fit: half as many dependent traversals per path tree, and sparse levels fill
// the wave twice as well. Requires use_bvh and an internal root (the caller checks); slots must be distinct.
// Engines as in traverse(): the wide engine for all_ok rays, the exact-topology engine for the others and for the
// slots the wide engine hands back; the winner checks run in the lanes that own the rays.
template <bool COUNT, bool NO_SPHERES>
RT_DEV void traverse_pair(const SceneView& sv, bool prune, bool want_c, const RayL& ray_c, uint32_t slot_c, float& t_c,
    int32_t& pos_c, bool want_a, const RayL& ray_a, uint32_t slot_a, float t_a, int32_t& pos_a, LaneCounters& lc, WaveLds& L)
{
    pos_c = pos_a = -1;
    if (ballot(want_c || want_a) == 0)
        return;
#ifdef RT_TIMELINE
    lc.tl_traversals++;
#endif
    const uint32_t lane = lane_id();
    prune = prune && sv.prune_ok != 0u;
    const bool wide = prune && sv.wide_ok != 0u;
    const RayInv inv_c = make_ray_inv(ray_c, sv.extent), inv_a = make_ray_inv(ray_a, sv.extent);
    const bool fast_c = want_c && wide && inv_c.all_ok(), fast_a = want_a && wide && inv_a.all_ok();
    // slot table in L.xfer (free between the engines' runs): 0 empty, 1 wide engine, 2 exact-topology engine
    L.xfer[lane] = 0u;
    wave_lds_fence();
    if (want_c) {
        L.rayA[slot_c] = make_float4(ray_c.o.x, ray_c.o.y, ray_c.o.z, ray_c.d.x);
        L.rayB[slot_c] = make_float4(ray_c.d.y, ray_c.d.z, t_c, __uint_as_float(inv_c.guard));
        L.rayC[slot_c] = make_float4(inv_c.y[0], inv_c.y[1], inv_c.y[2], inv_c.slack);
        L.best[slot_c] = kNoHit;
        L.xfer[slot_c] = fast_c ? 1u : 2u;
    }
    if (want_a) {
        L.rayA[slot_a] = make_float4(ray_a.o.x, ray_a.o.y, ray_a.o.z, ray_a.d.x);
        L.rayB[slot_a] = make_float4(ray_a.d.y, ray_a.d.z, t_a, __uint_as_float(inv_a.guard));
        L.rayC[slot_a] = make_float4(inv_a.y[0], inv_a.y[1], inv_a.y[2], inv_a.slack);
        L.best[slot_a] = kNoHit;
        L.xfer[slot_a] = fast_a ? 1u : 2u;
    }
    wave_lds_fence();
    const uint32_t mine = L.xfer[lane]; // the state of the slot this lane seeds
    wave_lds_fence();
    bool redo_c = want_c && !fast_c, redo_a = want_a && !fast_a;
    if (wide) {
        uint32_t flush_at = sv.leaf_flush < kWideFlushMax ? sv.leaf_flush : kWideFlushMax;
        WideWalk<Q_MIXED, COUNT> ww { sv, L, lc, lane, make_rsrc(sv.wide_nodes, sv.wide_n_nodes * 128u),
            make_rsrc(sv.wide_leaf, sv.wide_n_groups * 4u * 48u), make_rsrc(sv.wide_index, sv.wide_n_groups * 4u * 4u),
#ifdef RT_DIAG
            make_rsrc(sv.diag_wnodes, sv.wide_n_nodes * 128u), make_rsrc(sv.diag_wleaf, sv.wide_n_groups * 4u * 48u),
#endif
            0u, flush_at < 1u ? 1u : flush_at };
        const uint64_t back = ww.run(mine == 1u);
        if (fast_c) {
            const unsigned long long key = L.best[slot_c];
            redo_c = ((back >> slot_c) & 1ull) != 0ull || (key != kNoHit && !wide_winner_stands<COUNT>(sv, ray_c, inv_c, key, lc));
        }
        if (fast_a) {
            const unsigned long long key = L.best[slot_a];
            redo_a = ((back >> slot_a) & 1ull) != 0ull || (key != kNoHit && !wide_winner_stands<COUNT>(sv, ray_a, inv_a, key, lc));
        }
        wave_lds_fence();
        L.xfer[lane] = 0u;
        wave_lds_fence();
    }
    if (redo_c)
        L.best[slot_c] = kNoHit, L.xfer[slot_c] = 2u;
    if (redo_a)
        L.best[slot_a] = kNoHit, L.xfer[slot_a] = 2u;
    if (COUNT)
        lc.exact_rays += (redo_c ? 1u : 0u) + (redo_a ? 1u : 0u);
    wave_lds_fence();
    const bool slow = L.xfer[lane] == 2u;
    wave_lds_fence();
    if (ballot(slow)) {
        Walk<Q_MIXED, COUNT> w { sv, L, make_rsrc(sv.leaf, sv.n_prims * 48u), make_rsrc(sv.nodes, sv.n_pairs * 64u),
            make_rsrc(sv.leaf_grouped, sv.n_groups * 4u * 48u), make_rsrc(sv.groups, sv.n_groups * 32u),
            !NO_SPHERES && sv.prune_ok == 0u, lc, lane, prune, 0u, ray_c, inv_c };
        w.lane_engine(0u, slow, sv.root_ref, 0.0f);
        w.flush_leaves(1u);
        wave_lds_fence();
    }
    if (want_c) {
        const unsigned long long key = L.best[slot_c];
        if (key != kNoHit) { // the reported primitive's own t, with its exact bits (-0.0f included)
            pos_c = (int32_t)key_index(key);
            const PrimRecord rec = sv.prims[pos_c];
            float tt;
            if (!NO_SPHERES && (rec.kind_mat & 0x80000000u))
                sphere_candidate(rec, ray_c, t_c, tt);
            else
                tt = tri_plane_t(rec, ray_c);
            t_c = tt;
        }
    }
    if (want_a) {
        const unsigned long long key = L.best[slot_a];
        if (key != kNoHit)
            pos_a = (int32_t)key_index(key);
    }
    wave_lds_fence(); // the caller may reuse L for the next query
}

} // namespace rt
