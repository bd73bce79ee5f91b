// rt_stream.h — the streaming walk: one ray per lane from fetch to result, new rays taken from a queue as lanes
// finish (persistent waves).
//
// The per-lane walk of rt_wide.h (WideWalk::run) serves a FIXED set of rays — a tile's — and keeps its lanes busy
// by moving (node, ray) entries between lanes: every step pays for that generality (entries carry a ray slot, the
// ray's constants are read from the LDS table each step, idle lanes steal through LDS hand-overs, ballots and prefix
// sums decide who does what), and a wave lasts as long as its tile's longest path tree. Here a lane OWNS its ray:
// the ray's constants live in registers, the deque entries are plain node references, nobody steals — and a lane
// whose ray is finished takes the next ray of the launch's queue (one atomic per refill of a wave), so a wave never
// idles on a long ray and the launch ends when the queue is empty and the last rays are done, not after the
// heaviest tile's recursion.
//
// Same tree, same conservative box test, same leaf phase (WideWalk::flush_leaves: four lanes per (ray, leaf group)
// pair, exact triangle test, 64-bit atomicMin of the key in LDS), same pruning rule as WideWalk::run — so the set of
// contenders per ray and the argument at the top of rt_wide.h carry over unchanged: the smallest key among the
// contenders, checked against the reference's own test of the leaf it lives in (wide_winner_stands); rays that fail
// the check, overflow their deque or are not all_ok are handed to the exact-topology engine by the caller.
#pragma once
#include "rt_wide.h"

namespace rt {

// A source of rays for StreamWalk::run. All lanes call both functions together.
//   bool fetch(bool need, RayL& ray, float& t0, uint32_t& id)   lanes with `need` ask for a ray; true where one was given
//   void emit(bool done, uint32_t id, const RayL& ray, unsigned long long key, bool redo)
//        lanes with `done` hand back their ray's result: the winning key (kNoHit: none) or redo = the exact engine must trace it
template <int MODE, bool COUNT>
struct StreamWalk {
    WideWalk<MODE, COUNT> w; // leaf queue + leaf phase, buffer descriptors, counters

    // refill when at least this many lanes have nothing to do (fewer: they wait — a refill costs every lane the
    // ray set-up code)
#ifndef RT_REFILL_AT
#define RT_REFILL_AT 16
#endif
    static constexpr uint32_t kRefillAt = RT_REFILL_AT;

    template <class Src>
    RT_DEV void run(Src& src)
    {
        const SceneView& sv = w.sv;
        WaveLds& L = w.L;
        const uint32_t lane = w.lane;
        constexpr bool any = MODE == 1;
        // the lane's ray
        bool live = false;  // holds a ray that is not finished
        bool fast = false;  // ... which the wide engine may trace (all_ok)
        bool over = false;  // ... whose deque overflowed: result void, the exact engine traces it
        bool pend = false;  // ... with leaf pairs in the queue since the last leaf phase
        uint32_t id = 0u;
        RayL ray;
        ray.o = mk(0, 0, 0), ray.d = mk(0, 0, 1);
        RayInv inv = make_ray_inv(ray, sv.extent);
        float rx = 0, ry = 0, rz = 0, ox = 0, oy = 0, oz = 0, s = 0, t0 = 0, lim = 0;
        // the walk: current entry in registers, the others in the lane's column of L.wstack
        bool have = false;
        uint32_t node = 0u, sp = 0u;
        float e_lb = 0.0f;
        unsigned long long tk_refill = 0, tk_node = 0, tk_flush = 0, tk0 = wall_clock64(); // diagnostic (100 MHz ticks)
        for (;;) {
            if (COUNT)
                w.lc.grp++; // diagnostic: iterations of the wave (same in every lane)
            const unsigned long long tka = wall_clock64();
            // ---- rays that are finished leave, new ones come in ------------------------------------------------
            const bool walking = live && (have || sp > 0u || pend);
            const uint64_t busy = ballot(walking);
            if ((uint32_t)__builtin_popcountll(~busy) >= kRefillAt || busy == 0ull) {
                if (COUNT)
                    w.lc.sph++; // diagnostic: refills
                const bool done = live && !walking;
                if (ballot(done)) {
                    unsigned long long key = kNoHit;
                    bool redo = false;
                    if (done) {
                        key = L.best[lane];
                        redo = !fast || over;
#ifndef RT_STREAM_NOCHECK
                        if (!redo && key != kNoHit)
                            redo = !wide_winner_stands<COUNT>(sv, ray, inv, key, w.lc);
#endif
                    }
                    src.emit(done, id, ray, key, redo);
                    if (done)
                        live = false;
                }
                const bool got = src.fetch(!live, ray, t0, id);
                if (got) {
                    inv = make_ray_inv(ray, sv.extent);
                    fast = inv.all_ok();
                    live = true, over = false, pend = false;
                    rx = inv.y[0], ry = inv.y[1], rz = inv.y[2];
                    ox = ray.o.x * rx, oy = ray.o.y * ry, oz = ray.o.z * rz;
                    s = inv.slack * 1.0625f;
                    lim = t0;
                    L.rayA[lane] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
                    L.rayB[lane] = make_float4(ray.d.y, ray.d.z, t0, __uint_as_float(inv.guard));
                    L.best[lane] = kNoHit;
                    have = fast, node = 0u, e_lb = 0.0f, sp = 0u; // rays that are not all_ok finish at once (redo)
                }
                if (ballot(live) == 0ull)
                    break;
                wave_lds_fence();
            }
            const unsigned long long tkb = wall_clock64();
            tk_refill += tkb - tka;
            // ---- one node step per lane --------------------------------------------------------------------------
            if (!have && sp > 0u) {
                sp--;
                node = L.wstack[sp][lane];
                e_lb = u16_to_lb(L.wstack_lb[sp][lane]);
                have = true;
            }
#ifdef RT_TIMELINE
            w.lc.tl_node_iters++;
#endif
            uint32_t leafm = 0u;
            uint32_t ref[4] = { 0u, 0u, 0u, 0u };
            if (have) {
                have = false;
                const bool skip = any ? lim < 0.0f : e_lb > lim; // any hit: lim < 0 once the ray is blocked
                if (!skip) {
                    const uint32_t noff = node * 128u;
                    const float4 lox = buf_load16(w.nodes, noff), loy = buf_load16(w.nodes, noff + 16u);
                    const float4 loz = buf_load16(w.nodes, noff + 32u), hix = buf_load16(w.nodes, noff + 48u);
                    const float4 hiy = buf_load16(w.nodes, noff + 64u), hiz = buf_load16(w.nodes, noff + 80u);
                    const float4 rf = buf_load16(w.nodes, noff + 96u);
                    ref[0] = __float_as_uint(rf.x), ref[1] = __float_as_uint(rf.y);
                    ref[2] = __float_as_uint(rf.z), ref[3] = __float_as_uint(rf.w);
                    const float s2 = s + s, ms = -s, lims = lim + s;
                    const float bx0[4] = { lox.x, lox.y, lox.z, lox.w }, bx1[4] = { hix.x, hix.y, hix.z, hix.w };
                    const float by0[4] = { loy.x, loy.y, loy.z, loy.w }, by1[4] = { hiy.x, hiy.y, hiy.z, hiy.w };
                    const float bz0[4] = { loz.x, loz.y, loz.z, loz.w }, bz1[4] = { hiz.x, hiz.y, hiz.z, hiz.w };
                    uint32_t inm = 0u;
                    float lbk[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float tx0 = __builtin_fmaf(bx0[k], rx, -ox), tx1 = __builtin_fmaf(bx1[k], rx, -ox);
                        const float ty0 = __builtin_fmaf(by0[k], ry, -oy), ty1 = __builtin_fmaf(by1[k], ry, -oy);
                        const float tz0 = __builtin_fmaf(bz0[k], rz, -oz), tz1 = __builtin_fmaf(bz1[k], rz, -oz);
                        const float tin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
                        const float tout = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
                        const bool hit = (tin <= tout + s2) & (tout >= ms) & (tin <= lims) & (ref[k] != kWideEmpty);
                        lbk[k] = fmaxf(tin - s, 0.0f);
                        const uint32_t hb = hit ? (1u << k) : 0u;
                        leafm |= wide_is_leaf(ref[k]) ? hb : 0u;
                        inm |= wide_is_leaf(ref[k]) ? 0u : hb;
                    }
                    if (COUNT) {
                        w.lc.node_fetch++;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            w.lc.box += ref[k] != kWideEmpty ? 1u : 0u;
                    }
                    uint32_t nkey = 0xffffffffu;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t key = (__float_as_uint(lbk[k]) & ~3u) | (uint32_t)k;
                        nkey = (inm & (1u << k)) ? (key < nkey ? key : nkey) : nkey;
                    }
                    int near = inm ? (int)(nkey & 3u) : -1;
                    if (sp + 3u > kWideStack && (uint32_t)__builtin_popcount(inm) - (inm ? 1u : 0u) + sp > kWideStack) {
                        over = true; // no room: the ray's result is void (the exact engine traces it), its walk ends
                        near = -1, inm = 0u, leafm = 0u, sp = 0u;
                    }
                    const uint32_t pushm = near >= 0 ? inm & ~(1u << near) : 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (pushm & (1u << k)) {
                            L.wstack[sp][lane] = ref[k];
                            L.wstack_lb[sp][lane] = (uint16_t)lb_to_u16(lbk[k]);
                            sp++;
                        }
                    }
                    if (near >= 0) {
                        node = near == 0 ? ref[0] : (near == 1 ? ref[1] : (near == 2 ? ref[2] : ref[3]));
                        e_lb = __uint_as_float(nkey & ~3u);
                        have = true;
                    }
                }
            }
            pend = pend || leafm != 0u;
            w.queue_leaves4(leafm, lane, ref);
            const unsigned long long tkc = wall_clock64();
            tk_node += tkc - tkb;
            // ---- leaf phase: when the queue is long enough, or when nobody has a node to look at ---------------------
            if (w.lq_n >= w.flush_at || (w.lq_n != 0u && ballot(have || sp > 0u) == 0ull)) {
                if (COUNT)
                    w.lc.exact_rays++; // diagnostic: leaf phases
                w.flush_leaves();
                pend = false;
                const unsigned long long bk = L.best[lane];
                if (any) {
                    if (bk != kNoHit)
                        lim = -1.0f, sp = 0u, have = false; // blocked: the walk is over
                } else {
                    const uint32_t t0b = __float_as_uint(t0), bb = (uint32_t)(bk >> 32);
                    lim = __uint_as_float(t0b < bb ? t0b : bb);
                }
            }
            tk_flush += wall_clock64() - tkc;
        }
        if (COUNT) {
            w.lc.tri_fetch = (uint32_t)tk_refill, w.lc.empty_ticks_node = (uint32_t)tk_node, w.lc.empty_ticks_flush = (uint32_t)tk_flush;
            w.lc.empty_ticks_total = (uint32_t)(wall_clock64() - tk0);
        }
    }
};

} // namespace rt
