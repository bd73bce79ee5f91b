// rt_kernels.hip — HIP kernels of the Whitted render path for gfx950 (MI355X).
//
//   k_trace_level  one level of the flattened recursion of recursiveRayTrace
//                  (src/render.cpp:27-150): ray fetch (camera tile at level 0 —
//                  render.cpp:286-289,317 + trackball.cpp:101-110 — or the
//                  compacted reflection queue), closest-hit packet traversal
//                  (bvh.cpp:299-362), hit attributes (bvh.cpp:365-426), the light
//                  loop with shadow packets (light.cpp:49-73,108-165), Phong
//                  (shading.cpp:7-37), reflection spawn (shading.cpp:40-62) with
//                  wave-ballot compaction into the next level's queue.
//   k_fold         back-to-front colour fold `Lo = (direct + r) + r`
//                  (render.cpp:100,118) and the framebuffer write
//                  (render.cpp:322-323 → screen.cpp:41-47).
//   k_intersect    ≙ BvhInterface::intersect on a batch / the primary rays.
//   k_deinterleave multi-GPU strip gather → row-flipped frame.
//
// Each reflection ray is traversed ONCE: the reference traces the same ray
// twice (render.cpp:100 and :118) and adds the two identical results one after
// the other, which the fold reproduces as (direct + r) + r.
#include "rt_kernels.h"
#include "rt_shade.h"

namespace rt {

namespace {

#ifndef RT_BLOCK
#define RT_BLOCK 64
#endif
// One wave per workgroup: waves share nothing, and a multi-wave workgroup keeps its LDS
// (so its CU slot) until its slowest wave ends — with tiles as uneven as "knot" vs
// "background" that left ~60 % of the wave slots empty (SQ_WAVE_CYCLES, profiles/r01_notes.md).
constexpr int kBlock = RT_BLOCK;
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4 // __launch_bounds__ 2nd argument: 128 VGPRs (31 dwords spilled), 4 waves per SIMD: +3 % (tools/ab.py)
#endif
constexpr uint32_t kMinRaysPerWave = 4;

RT_DEV V3 quat_rotate(const float* q, V3 v)
{
    // glm/detail/type_quat.inl:347-354: v + ((uv * q.w) + uuv) * 2
    const V3 qv = mk(q[1], q[2], q[3]);
    const V3 uv = cross(qv, v);
    const V3 uuv = cross(qv, uv);
    return v + ((uv * q[0]) + uuv) * 2.0f;
}

// Trackball::generateRay (trackball.cpp:101-110) for the normalised position (px, py)
RT_DEV RayL generate_ray(const CameraArgs& c, float px, float py)
{
    const V3 csd = normalize(mk(-px * c.half_w, py * c.half_h, 1.0f));
    RayL r;
    r.o = mk(c.origin[0], c.origin[1], c.origin[2]);
    r.d = quat_rotate(c.quat, csd);
    return r;
}

// Primary ray of virtual column xv (= pixel xv / spp, sample xv % spp) of row y.
// Plain path: render.cpp:286-289,317. Samples: getRaySamples (:211-229) and getDOFRays
// (:245-276) with the counter-based generator in place of std::mt19937 (streams 62 / 63,
// keyed by pixel and sample — the oracle draws the same numbers).
RT_DEV RayL camera_ray(const CameraArgs& c, int32_t xv, int32_t y)
{
    const int32_t x = c.spp == 1 ? xv : xv / c.spp;
    const float px = float(x) / float(c.width) * 2.0f - 1.0f;
    const float py = float(y) / float(c.height) * 2.0f - 1.0f;
    if (c.spp == 1 && c.ms_n == 0 && c.dof_n == 0)
        return generate_ray(c, px, py);
    const int32_t s = xv - x * c.spp;
    const uint32_t pixel = (uint32_t)y * (uint32_t)c.width + (uint32_t)x;
    const int32_t n = c.ms_n;
    if (s < n * n) {
        const float pixelBoxX = (1 / float(c.width) * 2.f) / float(n), pixelBoxY = (1 / float(c.height) * 2.f) / float(n);
        const int32_t i = s / n, j = s - i * n;
        const float newPosX = px + float(i) * pixelBoxX, newPosY = py + float(j) * pixelBoxY;
        const float jx = rand01(rand31(c.seed, pixel, 62u, 0u, (uint32_t)s, 0u)) * pixelBoxX;
        const float jy = rand01(rand31(c.seed, pixel, 62u, 0u, (uint32_t)s, 1u)) * pixelBoxY;
        return generate_ray(c, newPosX + jx, newPosY + jy);
    }
    const uint32_t k = (uint32_t)(s - n * n);
    // getPlane (:231-236): normal = normalize(lookAt - position), D = dist - sqrt(dot(position, position))
    const V3 position = mk(c.origin[0], c.origin[1], c.origin[2]);
    const V3 N = normalize(mk(c.look_at[0], c.look_at[1], c.look_at[2]) - position);
    const float focalD = c.focus - sqrtf(dot(position, position));
    const RayL ray = generate_ray(c, px, py);
    V3 v1 = normalize(mk(-N.y, N.x, 0.0f));
    if (N.x == 0.f && N.y == 0.f)
        v1 = normalize(mk(N.z, 0.0f, -N.x));
    const V3 v2 = normalize(cross(N, v1));
    const float tf = (focalD - dot(N, ray.o)) / dot(N, ray.d); // getIntersection (:238-243)
    const V3 focalPoint = ray.o + tf * ray.d;
    const float r = sqrtf(rand01(rand31(c.seed, pixel, 63u, 0u, k, 0u)) * c.blur);
    const float theta = rand01(rand31(c.seed, pixel, 63u, 0u, k, 1u)) * (3.14159265358979323846264338327950288f * 2);
    const float ct = (float)cos((double)theta), st = (float)sin((double)theta); // as the oracle: via double
    RayL out;
    out.o = ray.o + r * ct * v1 + r * st * v2;
    out.d = focalPoint - out.o;
    return out;
}

// lane → pixel of the wave's tile (2^tile_wlog x 2^tile_hlog pixels; lanes past the
// tile carry no ray and start as work thieves); false for lanes outside the frame /
// row list. Small tiles spread a dense region over more waves: the frame time of a
// centred object is the time of its heaviest tile.
// Wave -> primary tile. With super_log = s > 0 the tiles are grouped into super-tiles
// of 2^s x 2^s tiles and whole super-tiles are dealt round-robin to the 8 XCDs
// (workgroup b runs on XCD b % 8): every XCD's L2 then serves the triangles behind
// 1/8 of the screen instead of the whole visible scene. Purely a locality choice —
// item / pixel indices do not depend on it.
RT_DEV bool tile_pixel(const CameraArgs& c, uint32_t wave, uint32_t lane, int32_t& x, int32_t& y, int32_t& lrow)
{
    const uint32_t tw = 1u << c.tile_wlog, th = 1u << c.tile_hlog;
    const uint32_t tiles_x = ((uint32_t)c.vwidth + tw - 1u) >> c.tile_wlog;
    uint32_t tx, ty;
    if (c.super_log == 0) {
        if (c.order) {
            if (c.xcd_order) { // launch position -> the (wave / 8)-th tile of XCD (wave % 8)'s list
                // the hardware deals WORKGROUPS round-robin to the XCDs: wave index == workgroup index only with
                // one-wave workgroups (the lists and launch_waves = 8 x the longest list are built on that)
                static_assert(kBlock == 64, "the XCD-aware tile order assumes one wave per workgroup");
                const uint32_t xl = wave & 7u, k = wave >> 3;
                if (k >= c.xcd_count[xl]) {
                    x = y = lrow = 0;
                    return false;
                }
                wave = c.order[c.xcd_base[xl] + k];
            } else {
                wave = c.order[wave];
            }
        }
        if (c.tiles_w) { // launched sub-grid
            ty = wave / c.tiles_w, tx = wave - ty * c.tiles_w;
            tx += c.tile_x0, ty += c.tile_y0;
        } else {
            ty = wave / tiles_x, tx = wave - ty * tiles_x;
        }
    } else {
        const uint32_t s = c.super_log, wpb = kBlock / 64;
        const uint32_t bpc = (1u << (2 * s)) / wpb; // workgroups per super-tile
        const uint32_t b = wave / wpb, wi = wave - b * wpb;
        const uint32_t xcd = b & 7u, j = b >> 3;
        const uint32_t chunk = (j / bpc) * 8u + xcd, k = (j % bpc) * wpb + wi;
        const uint32_t sx = (tiles_x + (1u << s) - 1u) >> s;
        const uint32_t scy = chunk / sx, scx = chunk - scy * sx;
        tx = (scx << s) + (k & ((1u << s) - 1u));
        ty = (scy << s) + (k >> s);
    }
    x = (int32_t)(tx * tw + (lane & (tw - 1u)));
    lrow = (int32_t)(ty * th + (lane >> c.tile_wlog));
    const bool ok = lane < tw * th && tx < tiles_x && x < c.vwidth && lrow < c.n_rows;
    y = 0;
    if (ok)
        y = c.rows ? c.rows[lrow] : c.row0 + lrow;
    return ok;
}

RT_DEV float wave_min_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
RT_DEV float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Can no primary ray of this tile come near the scene? The tile's rays lie inside the pyramid
// spanned by the four corner positions of its pixels' bounding rectangle (the direction is
// linear in the screen position before normalisation); if all eight corners of the scene
// bounds — grown by 2^-10 of the scene scale, three orders above the pruning slack and the
// rounding of these dot products — lie outside one of the pyramid's side planes, no ray can
// produce a candidate (same lemma as pruning: a candidate's point lies within rho of its
// triangle). Lanes 0..31 evaluate plane (lane & 3) against box corner (lane >> 2). Wave-uniform.
RT_DEV bool tile_misses_scene(const CameraArgs& c, const SceneView& sv, bool active, int32_t x, int32_t y)
{
    const uint32_t lane = lane_id();
    const float px = float(x) / float(c.width) * 2.0f - 1.0f, py = float(y) / float(c.height) * 2.0f - 1.0f;
    const float inf = __builtin_inff();
    const float px0 = wave_min_f(active ? px : inf), px1 = wave_max_f(active ? px : -inf);
    const float py0 = wave_min_f(active ? py : inf), py1 = wave_max_f(active ? py : -inf);
    auto corner = [&](uint32_t k) {
        const float cx = (k == 0u || k == 3u) ? px0 : px1, cy = k < 2u ? py0 : py1;
        return quat_rotate(c.quat, mk(-cx * c.half_w, cy * c.half_h, 1.0f));
    };
    const uint32_t k = lane & 3u;
    const V3 c0 = corner(k), c1 = corner((k + 1u) & 3u);
    const V3 mid = quat_rotate(c.quat, mk(-(0.5f * (px0 + px1)) * c.half_w, (0.5f * (py0 + py1)) * c.half_h, 1.0f));
    V3 n = cross(c0, c1);
    if (dot(n, mid) < 0.0f)
        n = -n; // inside = positive side
    const float S = fmaxf(sv.extent, fmaxf(fabsf(c.origin[0]), fmaxf(fabsf(c.origin[1]), fabsf(c.origin[2]))));
    const float g = S * 0.0009765625f;
    const uint32_t bc = lane >> 2;
    const V3 P = mk((bc & 1u) ? sv.scene_hi[0] + g : sv.scene_lo[0] - g, (bc & 2u) ? sv.scene_hi[1] + g : sv.scene_lo[1] - g,
        (bc & 4u) ? sv.scene_hi[2] + g : sv.scene_lo[2] - g);
    const float side = dot(n, P - mk(c.origin[0], c.origin[1], c.origin[2]));
    const uint32_t outside = (uint32_t)(ballot(lane < 32u && side < 0.0f) & 0xffffffffull); // NaN: not outside
    bool miss = false;
#pragma unroll
    for (uint32_t p = 0; p < 4u; p++)
        miss = miss || ((outside >> p) & 0x11111111u) == 0x11111111u;
    return miss;
}

RT_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

struct RayCounts {
    uint32_t closest = 0, shadow = 0, hits = 0, queued = 0, culled = 0, skipped = 0;
};

RT_DEV void flush_counters(rt_counters* out, const LaneCounters& lc, const RayCounts& rc, int32_t level, uint32_t lane)
{
    const uint32_t box = wave_sum(lc.box), tri = wave_sum(lc.tri), sph = wave_sum(lc.sph);
    const uint32_t closest = wave_sum(rc.closest), shadow = wave_sum(rc.shadow);
    const uint32_t hits = wave_sum(rc.hits), queued = wave_sum(rc.queued);
    const uint32_t node_fetch = wave_sum(lc.node_fetch), tri_fetch = wave_sum(lc.tri_fetch), grp = wave_sum(lc.grp);
    const uint32_t culled = wave_sum(rc.culled), exact = wave_sum(lc.exact_rays), skipped = wave_sum(rc.skipped);
    if (lane == 0) {
        auto add = [](uint64_t* p, uint64_t v) {
            if (v)
                atomicAdd((unsigned long long*)p, (unsigned long long)v);
        };
        add(&out->rays, (uint64_t)closest + shadow);
        add(level == 0 ? &out->rays_primary : &out->rays_reflect, closest);
        add(&out->rays_shadow, shadow);
        add(&out->rays_ref_equiv, ((uint64_t)closest + shadow + skipped) << (level > 0 ? level : 0));
        add(&out->rays_shadow_skipped, skipped);
        add(&out->box_tests, box);
        add(&out->tri_tests, tri);
        add(&out->sphere_tests, sph);
        add(&out->node_fetches, node_fetch);
        add(&out->tri_fetches, tri_fetch);
        add(&out->hits_shaded, hits);
        add(&out->rays_queued, queued);
        add(&out->group_tests, grp);
        add(&out->rays_culled, culled);
        add(&out->rays_exact, exact);
    }
}

// bvh.intersect's HitInfo (bvh.cpp:365-426) and computeLightContribution (light.cpp:108-165) for the
// lanes with `hit`: returns Lo (the albedo when shading is off); ALL lanes of the wave must call
// (the shadow rays are traced by the whole wave).
// GENERAL: also compiles the shadow query of enableTransparency (closest blocker's transparency) —
// only k_trace_general needs it, the plain kernel stays one traversal instantiation smaller.
// Position and colour of sample s of light L for the pixel with RNG key `pixel`
// (point light; sampleSegmentLight, light.cpp:19-26; sampleParallelogramLight, light.cpp:30-45)
RT_DEV void light_sample(const LightRec RT_CONST_AS* L, int32_t kind, const LevelArgs& a, uint32_t pixel, int32_t level,
    uint32_t li, int32_t s, V3& lpos, V3& lcol)
{
    if (kind == RT_LIGHT_POINT) {
        lpos = mk(L->v[0][0], L->v[0][1], L->v[0][2]);
        lcol = mk(L->v[1][0], L->v[1][1], L->v[1][2]);
    } else if (kind == RT_LIGHT_SEGMENT) {
        const V3 e0 = mk(L->v[0][0], L->v[0][1], L->v[0][2]), e1 = mk(L->v[1][0], L->v[1][1], L->v[1][2]);
        const V3 c0 = mk(L->v[2][0], L->v[2][1], L->v[2][2]), c1 = mk(L->v[3][0], L->v[3][1], L->v[3][2]);
        const float r = rand01(rand31(a.seed, pixel, (uint32_t)level, li, (uint32_t)s, 0u));
        const float weight = ((float)s + r) / (float)a.segment_n;
        lpos = (e1 - e0) * weight + e0;
        lcol = weight * c1 + (1 - weight) * c0;
    } else {
        const int32_t n = a.parallelogram_n;
        const int32_t i = s / n, k = s - i * n;
        const V3 v0 = mk(L->v[0][0], L->v[0][1], L->v[0][2]);
        const V3 e01 = mk(L->v[1][0], L->v[1][1], L->v[1][2]), e02 = mk(L->v[2][0], L->v[2][1], L->v[2][2]);
        const V3 c0 = mk(L->v[3][0], L->v[3][1], L->v[3][2]), c1 = mk(L->v[4][0], L->v[4][1], L->v[4][2]);
        const V3 c2 = mk(L->v[5][0], L->v[5][1], L->v[5][2]), c3 = mk(L->v[6][0], L->v[6][1], L->v[6][2]);
        const float horRandom = rand01(rand31(a.seed, pixel, (uint32_t)level, li, (uint32_t)s, 0u));
        const float verRandom = rand01(rand31(a.seed, pixel, (uint32_t)level, li, (uint32_t)s, 1u));
        const float horWeight = ((float)i + horRandom) / (float)n;
        const float verWeight = ((float)k + verRandom) / (float)n;
        const V3 horVector = horWeight * e01;
        const V3 verVector = verWeight * e02;
        lpos = v0 + horVector + verVector;
        const V3 bottomColor = horWeight * c1 + (1 - horWeight) * c0;
        const V3 topColor = horWeight * c3 + (1 - horWeight) * c2;
        lcol = verWeight * topColor + (1 - verWeight) * bottomColor;
    }
}

// computeLightContribution multiplies a light sample's colour by its visibility (light.cpp:121,135,153). Where
// every component of the colour is +-0 or NaN — a surface facing away from the light: diffuse = kd * Lc * 0, no
// specular term — the product has the same bits for visibility 0 and 1, so the sample's shadow ray is not traced
// (it is reported as rays_shadow_skipped, not as a traversed ray). Exact; rt_set_option("skip_unlit", 0) traces them.
RT_DEV bool visibility_is_moot(V3 c)
{
    return (c.x == 0.0f || c.x != c.x) && (c.y == 0.0f || c.y != c.y) && (c.z == 0.0f || c.z != c.z);
}

// The next level's closest-hit query, traced TOGETHER with this level's first shadow query (traverse_pair): the
// reflection rays take ray slots 0..31 and the shadow rays slots 32..63 by their rank among the wave's rays of
// that kind, whatever lanes hold them — taken level by level whenever each kind fits its 32 slots (sparse deeper
// levels of 64-pixel tiles, every level of tiles of up to 32 pixels). One traversal with twice the rays instead of
// two: better lane use in the node phase and half as many serial traversals per path tree.
struct MergeNext {
    bool enabled = false; // the launch runs fused (and merge_next is on)
    bool done = false;    // the merged traversal ran: pos / t hold the next level's closest hit
    bool spawn = false;   // this lane has a reflection ray
    RayL refl;
    int32_t pos = -1;
    float t = kFltMax;
};

// LEAN: the kernel variant for the common plain case — point lights only (or soft shadows off),
// triangles only, no normal interpolation, no textures; none of that code is compiled in (the
// area-light code alone costs the point-light path 2 % through register allocation).
// NOSPH: the scene holds triangles only (no sphere test in the shadow traversals)
template <bool COUNT, bool GENERAL, bool LEAN, bool NOSPH = LEAN>
RT_DEV V3 shade_hit(const LevelArgs& a, bool hit, const RayL& ray, float t, int32_t pos, int32_t level, uint32_t pixel,
    const HitL& h, LaneCounters& lc, RayCounts& rc, WaveLds& wl, MergeNext* mg)
{
    const SceneView& sv = a.sv;
    const uint32_t flags = a.flags;
    const bool use_bvh = (flags & F_ACCEL) != 0;
    const bool prune = (flags & F_NO_PRUNE) == 0;
    (void)pos;
    // ---- computeLightContribution (light.cpp:108-165) ---------------------------
    V3 Lo = h.kd; // !enableShading: the albedo (light.cpp:161-164)
    if (flags & F_SHADING) {
        V3 result = mk(0.0f, 0.0f, 0.0f);
        const LightRec RT_CONST_AS* lights = (const LightRec RT_CONST_AS*)sv.lights;
        // Area lights: the wave's n_hit hit points take 64 / n_hit shadow samples each per traversal — lane
        // q + j * n_hit carries the ray of the q-th hit's sample s0 + j, whatever lanes the hits sit in
        const uint64_t hit_mask = ballot(hit);
        const uint32_t n_hit = (uint32_t)__builtin_popcountll(hit_mask);
        const uint32_t batch = n_hit ? 64u / n_hit : 1u;
        for (uint32_t li = 0; li < sv.n_lights; li++) {
            const LightRec RT_CONST_AS* L = lights + li;
            const int32_t kind = L->kind;
            const bool is_point = kind == RT_LIGHT_POINT;
            if (!is_point && (LEAN || !(flags & F_SOFT_SHADOW)))
                continue; // segment / parallelogram lights only contribute with soft shadows
            const bool trace_shadow = is_point ? (flags & F_HARD_SHADOW) != 0 : true;
            const int32_t n = a.parallelogram_n;
            const int32_t nsamp = is_point ? 1 : (kind == RT_LIGHT_SEGMENT ? a.segment_n : n * n);
            V3 acc = mk(0.0f, 0.0f, 0.0f);
            // One any-hit traversal serves batch samples of every hit; the pixel lane then adds the
            // contributions in the reference's order.
            int32_t s_begin = 0;
            if (!LEAN && !GENERAL && !is_point && batch > 1u) {
                const uint32_t lane = lane_id();
                const uint32_t q_mine = prefix_count(hit_mask); // this lane's rank among the hits (if it is one)
                wave_lds_fence();
                if (hit)
                    wl.xfer[q_mine] = lane; // rank -> lane (xfer is free between traversals)
                wave_lds_fence();
                const uint32_t kk = lane / n_hit;
                const int src = (int)wl.xfer[lane % n_hit];
                wave_lds_fence();
                RayL rs;
                rs.o = mk(__shfl(ray.o.x, src, 64), __shfl(ray.o.y, src, 64), __shfl(ray.o.z, src, 64));
                rs.d = mk(__shfl(ray.d.x, src, 64), __shfl(ray.d.y, src, 64), __shfl(ray.d.z, src, 64));
                const float ts = __shfl(t, src, 64);
                const uint32_t pix_s = (uint32_t)__shfl((int)pixel, src, 64);
                const bool try_known = use_bvh && prune && sv.wide_ok != 0u;
                int32_t known = -1; // what blocked this lane's previous sample is tried first (wide_known_blocker)
                HitL hs; // the hit this lane's samples belong to (for visibility_is_moot)
                hs.normal = mk(__shfl(h.normal.x, src, 64), __shfl(h.normal.y, src, 64), __shfl(h.normal.z, src, 64));
                hs.kd = mk(__shfl(h.kd.x, src, 64), __shfl(h.kd.y, src, 64), __shfl(h.kd.z, src, 64));
                hs.ks = mk(__shfl(h.ks.x, src, 64), __shfl(h.ks.y, src, 64), __shfl(h.ks.z, src, 64));
                hs.shininess = __shfl(h.shininess, src, 64), hs.transparency = 1.0f, hs.texture = -1;
                for (int32_t s0 = 0; s0 < nsamp; s0 += (int32_t)batch) {
                    const int32_t s_mine = s0 + (int32_t)kk;
                    const bool mine = kk < batch && s_mine < nsamp;
                    V3 lp, lcq;
                    light_sample(L, kind, a, pix_s, level, li, mine ? s_mine : 0, lp, lcq);
                    const RayL sray = shadow_ray(rs, ts, lp);
                    float st = 1.0f;
                    int32_t spos;
                    const bool moot = a.skip_unlit != 0u && visibility_is_moot(compute_shading(lp, lcq, rs, ts, hs));
                    bool need = mine && !moot, by_known = false;
                    if (COUNT && mine) { // counted where the ray is traced (the lane that carries the sample)
                        if (moot)
                            rc.skipped++;
                        else
                            rc.shadow++;
                    }
                    if (try_known && ballot(need && known >= 0)) {
                        if (need && known >= 0 && wide_known_blocker<COUNT>(sv, sray, (uint32_t)known, lc))
                            need = false, by_known = true;
                    }
                    traverse<Q_ANY, COUNT, NOSPH>(sv, use_bvh, prune, need, sray, st, spos, lc, wl);
                    if (spos >= 0)
                        known = spos;
                    const int blocked = (spos >= 0 || by_known) ? 1 : 0;
                    for (uint32_t j = 0; j < batch && s0 + (int32_t)j < nsamp; j++) {
                        const float visibility = __shfl(blocked, (int)((q_mine + j * n_hit) & 63u), 64) ? 0.0f : 1.0f;
                        V3 lpos, lcol;
                        light_sample(L, kind, a, pixel, level, li, s0 + (int32_t)j, lpos, lcol);
                        V3 color = mk(0.0f, 0.0f, 0.0f);
                        if (hit)
                            color = compute_shading(lpos, lcol, ray, t, h);
                        acc = acc + color * visibility;
                    }
                }
                s_begin = nsamp;
            }
            // area lights: what blocked a pixel's previous sample is tried first for the next one (wide_known_blocker)
            const bool try_blocker = !LEAN && !GENERAL && !is_point && use_bvh && prune && sv.wide_ok != 0u;
            int32_t last_blocker = -1;
            for (int32_t s = s_begin; s < nsamp; s++) {
                V3 lpos, lcol;
                light_sample(L, kind, a, pixel, level, li, s, lpos, lcol);
                float visibility = 1.0f;
                V3 color = mk(0.0f, 0.0f, 0.0f);
                if (hit)
                    color = compute_shading(lpos, lcol, ray, t, h);
                if (trace_shadow) { // testVisibilityLightSample, light.cpp:49-73
                    const RayL sray = shadow_ray(ray, t, lpos);
                    float st = 1.0f;
                    int32_t spos;
                    // with enableTransparency the visibility is a material's transparency, not 0 / 1: every ray is traced
                    const bool moot = a.skip_unlit != 0u && !(GENERAL && (flags & F_TRANSPARENCY)) && visibility_is_moot(color);
                    bool need = hit && !moot;
                    if (COUNT && hit) {
                        if (moot)
                            rc.skipped++;
                        else
                            rc.shadow++;
                    }
                    if (try_blocker && ballot(need && last_blocker >= 0)) {
                        if (need && last_blocker >= 0 && wide_known_blocker<COUNT>(sv, sray, (uint32_t)last_blocker, lc)) {
                            visibility = 0.0f;
                            need = false;
                        }
                    }
                    // this level's first shadow rays + the next level's reflection rays in ONE traversal when each kind
                    // fits its 32 ray slots (traverse_pair): rays go to slots by their rank, whatever lanes hold them
                    const uint64_t m_c = (!GENERAL && mg && mg->enabled && !mg->done) ? ballot(mg->spawn) : 0ull;
                    const uint64_t m_a = ballot(need);
                    if (!GENERAL && mg && mg->enabled && !mg->done && use_bvh && sv.root_cnt == 0u
                        && __builtin_popcountll(m_c) <= 32 && __builtin_popcountll(m_a) <= 32) {
                        float tq = kFltMax;
                        int32_t pq, pa;
                        traverse_pair<COUNT, NOSPH>(sv, prune, mg->spawn, mg->refl, prefix_count(m_c), tq, pq, need, sray,
                            32u + prefix_count(m_a), 1.0f, pa, lc, wl);
                        mg->pos = pq, mg->t = tq, mg->done = true;
                        if (pa >= 0) {
                            visibility = 0.0f;
                            last_blocker = pa;
                        }
                    } else if (!GENERAL || !(flags & F_TRANSPARENCY)) {
                        traverse<Q_ANY, COUNT, NOSPH>(sv, use_bvh, prune, need, sray, st, spos, lc, wl);
                        if (spos >= 0) {
                            visibility = 0.0f;
                            last_blocker = spos;
                        }
                    } else { // light.cpp:66-69: the CLOSEST blocker's material.transparency
                        traverse<Q_CLOSEST, COUNT>(sv, use_bvh, prune, hit, sray, st, spos, lc, wl);
                        if (spos >= 0)
                            visibility = hit_transparency(sv, flags, spos);
                    }
                }
                if (is_point)
                    result = result + color * visibility;
                else
                    acc = acc + color * visibility;
            }
            if (!is_point) {
                const float denom = kind == RT_LIGHT_SEGMENT ? (float)a.segment_n : (float)n * (float)n;
                result = result + acc / denom;
            }
        }
        Lo = result;
    }
    return Lo;
}

// ---------------------------------------------------------------------------------------------
// k_trace_general — recursiveRayTrace (src/render.cpp:27-147) with ALL its branches: the
// transparency-only chain (:36-49), glossy reflections (:56-99) and the `transparency != 1`
// branch (:101-130). The fused level loop of k_trace_level follows ONE continuation per level;
// these branches make up to three calls per invocation (or raysPerReflection of them), so every
// lane runs the recursion as an explicit state machine with its frames in global memory, and the
// wave traces one pending ray per lane per step. Same float operations in the same order as the
// oracle; glossy samples from rt_rand31 stream 61 keyed by the pixel and a draw counter.
// Used only when those flags / materials are present — the plain path stays in k_trace_level.
// ---------------------------------------------------------------------------------------------
enum : int32_t { GS_A, GS_B, GS_BG, GS_C, GS_C2, GS_END, GS_RET, GS_RESUME, GS_DONE };

struct GFrame {
    V3 Lo, ks, normal, total, w, u, v, gorigin, r2o, r2d;
    float transp, shin, a, r2t;
    int32_t depth, pc, gi, valid2;
};
RT_DEV void gframe_store(float* f, const GFrame& x)
{
    const float v[kFrameWords] = { x.Lo.x, x.Lo.y, x.Lo.z, x.ks.x, x.ks.y, x.ks.z, x.normal.x, x.normal.y, x.normal.z,
        x.total.x, x.total.y, x.total.z, x.w.x, x.w.y, x.w.z, x.u.x, x.u.y, x.u.z, x.v.x, x.v.y, x.v.z, x.gorigin.x,
        x.gorigin.y, x.gorigin.z, x.r2o.x, x.r2o.y, x.r2o.z, x.r2d.x, x.r2d.y, x.r2d.z, x.transp, x.shin, x.a, x.r2t,
        __int_as_float(x.depth), __int_as_float(x.pc), __int_as_float(x.gi), __int_as_float(x.valid2), 0.0f, 0.0f };
#pragma unroll
    for (int i = 0; i < kFrameWords; i++)
        f[i] = v[i];
}
RT_DEV void gframe_load(const float* f, GFrame& x)
{
    float v[kFrameWords];
#pragma unroll
    for (int i = 0; i < kFrameWords; i++)
        v[i] = f[i];
    x.Lo = mk(v[0], v[1], v[2]), x.ks = mk(v[3], v[4], v[5]), x.normal = mk(v[6], v[7], v[8]);
    x.total = mk(v[9], v[10], v[11]), x.w = mk(v[12], v[13], v[14]), x.u = mk(v[15], v[16], v[17]);
    x.v = mk(v[18], v[19], v[20]), x.gorigin = mk(v[21], v[22], v[23]), x.r2o = mk(v[24], v[25], v[26]);
    x.r2d = mk(v[27], v[28], v[29]), x.transp = v[30], x.shin = v[31], x.a = v[32], x.r2t = v[33];
    x.depth = __float_as_int(v[34]), x.pc = __float_as_int(v[35]), x.gi = __float_as_int(v[36]);
    x.valid2 = __float_as_int(v[37]);
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock, RT_WAVES_PER_SIMD) void k_trace_general(const LevelArgs a)
{
    const uint32_t lane = lane_id();
    const uint32_t slot = uni(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    const uint32_t n_slots = gridDim.x * (kBlock / 64);
    const SceneView& sv = a.sv;
    const uint32_t flags = a.flags;
    const bool use_bvh = (flags & F_ACCEL) != 0, prune = (flags & F_NO_PRUNE) == 0;
    const bool fR = (flags & F_RECURSIVE) != 0, fT = (flags & F_TRANSPARENCY) != 0, fG = (flags & F_GLOSSY) != 0;
    __shared__ WaveLds lds[kBlock / 64];
    WaveLds& wl = lds[threadIdx.x >> 6];
    float* const my_frames = a.frames + ((size_t)slot * 64u + lane) * (size_t)a.max_frames * kFrameWords;
    uint32_t n_units;
    if (a.source == SRC_CAMERA) {
        const uint32_t tw = 1u << a.cam.tile_wlog, th = 1u << a.cam.tile_hlog;
        n_units = a.cam.tiles_w ? a.cam.tiles_w * a.cam.tiles_h
                                : (((uint32_t)a.cam.vwidth + tw - 1u) >> a.cam.tile_wlog) * (((uint32_t)a.cam.n_rows + th - 1u) >> a.cam.tile_hlog);
        if (a.cam.order && a.cam.xcd_order)
            n_units = a.cam.launch_waves; // eight lists of tiles, padded to the longest (tile_pixel)
    } else {
        n_units = (a.in_count_imm + 63u) / 64u;
    }
    for (uint32_t unit = slot; unit < n_units; unit += n_slots) {
        // ---- this lane's primary ray ------------------------------------------------------
        bool active = false;
        RayL ray;
        ray.o = mk(0, 0, 0), ray.d = mk(0, 0, -1);
        float tlim = kFltMax;
        uint32_t item = 0, pixel = 0;
        if (a.source == SRC_CAMERA) {
            int32_t x, y, lrow;
            active = tile_pixel(a.cam, unit, lane, x, y, lrow);
            if (active) {
                ray = camera_ray(a.cam, x, y);
                item = (uint32_t)lrow * (uint32_t)a.cam.vwidth + (uint32_t)x;
                pixel = (uint32_t)y * (uint32_t)a.cam.vwidth + (uint32_t)x;
            }
        } else {
            const uint32_t i = unit * 64u + lane;
            active = i < a.in_count_imm;
            if (active) {
                const float* q = a.in.ray;
                const size_t cap = a.in.cap;
                ray.o = mk(q[i], q[cap + i], q[2 * cap + i]);
                ray.d = mk(q[3 * cap + i], q[4 * cap + i], q[5 * cap + i]);
                if (a.in.t)
                    tlim = a.in.t[i];
                item = a.in.item ? a.in.item[i] : i;
                pixel = a.in.pixel ? a.in.pixel[i] : i;
            }
        }
        if (ballot(active) == 0)
            continue;
        int32_t op = active ? 1 : 0; // 0 none, 1 trace + shade (a recursiveRayTrace invocation), 2 probe
        int32_t rayDepth = a.depth, sp = 0;
        uint32_t draws = 0, equiv_extra = 0;
        V3 result = mk(0.0f, 0.0f, 0.0f);
        for (;;) {
            const bool want = op != 0;
            if (ballot(want) == 0)
                break;
            LaneCounters lc;
            RayCounts rc;
            if (COUNT && want)
                rc.closest = 1;
            float t = tlim;
            int32_t pos;
            traverse<Q_CLOSEST, COUNT>(sv, use_bvh, prune, want, ray, t, pos, lc, wl);
            const bool full = op == 1;
            const bool hit = full && pos >= 0;
            const int32_t level = a.depth - rayDepth;
            HitL h;
            h.normal = mk(0, 0, 1), h.kd = mk(0, 0, 0), h.ks = mk(0, 0, 0);
            h.shininess = 1.0f, h.transparency = 1.0f, h.texture = -1;
            if (hit)
                resolve_hit(sv, flags, ray, t, pos, h);
            const V3 Lo0 = shade_hit<COUNT, true, false>(a, hit, ray, t, pos, level, pixel, h, lc, rc, wl, nullptr);
            if (COUNT && hit)
                rc.hits = 1;
            if (COUNT)
                flush_counters(a.counters, lc, rc, 0, lane);
            int32_t stage = GS_DONE;
            V3 val = mk(0.0f, 0.0f, 0.0f);
            GFrame fv;
            fv.Lo = Lo0, fv.ks = h.ks, fv.normal = h.normal, fv.total = mk(0, 0, 0), fv.w = fv.u = fv.v = mk(0, 0, 0);
            fv.gorigin = mk(0, 0, 0), fv.r2o = mk(0, 0, 0), fv.r2d = mk(0, 0, 0), fv.transp = h.transparency;
            fv.shin = h.shininess, fv.a = 0.0f, fv.r2t = kFltMax, fv.depth = rayDepth, fv.pc = 0, fv.gi = 0, fv.valid2 = 0;
            RayL refl;
            refl.o = mk(0, 0, 0), refl.d = mk(0, 0, 0);
            bool refl_ok = false;
            if (op == 2) { // the probe of render.cpp:44: its hit distance is the callee's initial ray.t
                tlim = t;
                op = 1;
            } else if (full) {
                if (!hit) {
                    stage = GS_RET;
                } else {
                    refl_ok = compute_reflection_ray(ray, t, h, refl); // :34
                    if (fR && !fG) { // the ray of :103-110, needed after the first calls return
                        if (fT && h.transparency != 1.0f) {
                            fv.r2o = (0.00001f + t) * ray.d + ray.o;
                            fv.r2d = ray.d;
                            fv.r2t = kFltMax;
                            fv.valid2 = 1;
                        } else {
                            fv.r2o = refl.o, fv.r2d = refl.d;
                            fv.r2t = refl_ok ? kFltMax : 0.0f; // the sentinel Ray{0, 0, 0} of shading.cpp
                            fv.valid2 = refl_ok ? 1 : 0;
                        }
                    }
                    stage = GS_A;
                }
            }
            // ---- per-lane control: runs until the lane has a new ray to trace or is finished -----
            // call(): push the frame and make `child` the lane's pending ray
            RayL child;
            child.o = mk(0, 0, 0), child.d = mk(0, 0, 0);
            float child_t = kFltMax;
            int32_t child_depth = 0;
            bool child_probe = false;
            while (stage != GS_DONE) {
                bool do_call = false;
                switch (stage) {
                case GS_A:
                    if (fT && !fR && !fG) { // :36-49
                        if (fv.transp == 1.0f || fv.depth <= 0) {
                            val = fv.Lo;
                            stage = GS_RET;
                            break;
                        }
                        fv.Lo = fv.Lo * fv.transp;
                        child.o = (0.00001f + t) * ray.d + ray.o;
                        child.d = ray.d;
                        child_t = kFltMax, child_depth = fv.depth - 1, child_probe = true;
                        fv.pc = 1;
                        do_call = true;
                    } else {
                        stage = GS_B;
                    }
                    break;
                case GS_B:
                    if (refl_ok && !(fv.depth < 1) && fR) { // :52-54
                        if (fG && fv.shin != 0.0f) { // :56-71
                            const V3 w = normalize(refl.d);
                            V3 tt = w;
                            float mn = tt.x;
                            int minIdx = 0;
                            if (tt.y < mn) {
                                mn = tt.y;
                                minIdx = 1;
                            }
                            if (tt.z < mn) {
                                mn = tt.z;
                                minIdx = 2;
                            }
                            if (minIdx == 0)
                                tt.x = 1.0f;
                            else if (minIdx == 1)
                                tt.y = 1.0f;
                            else
                                tt.z = 1.0f;
                            const V3 u = cross(tt, w) / length(cross(tt, w));
                            fv.w = w, fv.u = u, fv.v = cross(w, u);
                            fv.a = (1 / fv.shin) * a.alpha_modifier;
                            fv.gorigin = refl.o;
                            fv.total = mk(.0f, .0f, .0f);
                            fv.gi = 0;
                            stage = GS_BG;
                        } else { // :97
                            child = refl, child_t = kFltMax, child_depth = fv.depth - 1, child_probe = false;
                            fv.pc = 2;
                            do_call = true;
                        }
                    } else {
                        stage = GS_C;
                    }
                    break;
                case GS_BG: { // the sample loop of :82-95
                    bool called = false;
                    while (fv.gi < a.rays_per_reflection) {
                        const float randOne = rand01(rand31(a.seed, pixel, 61u, 0u, draws & 0x1ffffu, 0u));
                        const float randTwo = rand01(rand31(a.seed, pixel, 61u, 0u, draws & 0x1ffffu, 1u));
                        draws++;
                        fv.gi++;
                        const float weightU = -fv.a / 2 + randOne * fv.a;
                        const float weightV = -fv.a / 2 + randTwo * fv.a;
                        V3 gr = fv.w + weightU * fv.u + weightV * fv.v;
                        gr = normalize(gr);
                        if (dot(fv.normal, gr) > 0) {
                            child.o = fv.gorigin, child.d = gr, child_t = kFltMax;
                            child_depth = fv.depth - 1 < a.glossy_cap ? fv.depth - 1 : a.glossy_cap;
                            child_probe = false;
                            fv.pc = 3;
                            called = true;
                            break;
                        }
                    }
                    if (called) {
                        do_call = true;
                    } else {
                        fv.total = fv.total / float(a.rays_per_reflection);
                        fv.Lo = fv.Lo + fv.total;
                        stage = GS_C;
                    }
                    break;
                }
                case GS_C:
                    if (fR && !fG) { // :101-119
                        if (fv.valid2 && fv.depth > 0) {
                            child.o = fv.r2o, child.d = fv.r2d, child_t = fv.r2t, child_depth = fv.depth - 1;
                            child_probe = false;
                            fv.pc = 4;
                            do_call = true;
                        } else {
                            stage = GS_C2;
                        }
                    } else {
                        stage = GS_END;
                    }
                    break;
                case GS_C2:
                    if (fv.transp != 1.0f) { // :121-122, no depth test
                        child.o = fv.r2o, child.d = fv.r2d, child_t = fv.r2t, child_depth = fv.depth - 1;
                        child_probe = false;
                        fv.pc = 5;
                        do_call = true;
                    } else {
                        stage = GS_END;
                    }
                    break;
                case GS_END:
                    val = fv.Lo;
                    stage = GS_RET;
                    break;
                case GS_RET: // `val` is this invocation's return value
                    if (sp == 0) {
                        result = val;
                        op = 0;
                        stage = GS_DONE;
                    } else {
                        sp--;
                        gframe_load(my_frames + (size_t)sp * kFrameWords, fv);
                        stage = GS_RESUME;
                    }
                    break;
                case GS_RESUME:
                    if (fv.pc == 1) {
                        fv.Lo = fv.Lo + (1.0f - fv.transp) * val;
                        stage = GS_END;
                    } else if (fv.pc == 2) {
                        fv.Lo = fv.Lo + val;
                        stage = GS_C;
                    } else if (fv.pc == 3) {
                        fv.total = fv.total + val * fv.ks;
                        stage = GS_BG;
                    } else if (fv.pc == 4) {
                        fv.Lo = fv.Lo + val;
                        stage = GS_C2;
                    } else {
                        val = fv.transp * val + (1 - fv.transp) * fv.Lo; // :123
                        // the intersect of :125-127 has no effect and is not performed: counted as a
                        // reference-equivalent ray only
                        if (COUNT)
                            equiv_extra++;
                        stage = GS_RET;
                    }
                    break;
                default:
                    stage = GS_DONE;
                    break;
                }
                if (do_call) {
                    if (sp + 1 > a.depth + kExtraFrames) { // recursion guard (oracle: same): the callee returns vec3(0)
                        val = mk(0.0f, 0.0f, 0.0f);
                        stage = GS_RESUME;
                    } else {
                        gframe_store(my_frames + (size_t)sp * kFrameWords, fv);
                        sp++;
                        ray = child, tlim = child_t, rayDepth = child_depth;
                        op = child_probe ? 2 : 1;
                        stage = GS_DONE;
                    }
                }
            }
        }
        if (COUNT) {
            const uint32_t extra = wave_sum(equiv_extra);
            if (lane == 0 && extra)
                atomicAdd((unsigned long long*)&a.counters->rays_ref_equiv, (unsigned long long)extra);
        }
        if (active) { // one "level" holding the finished colour: k_fold adds vec3(0) and divides by 1
            a.state[item] = 1u;
            float* d = a.direct + 3 * (size_t)item;
            d[0] = result.x, d[1] = result.y, d[2] = result.z;
        }
    }
}

// MULTI: several frames in one launch (CameraArgs::n_frames > 1): the tile's frame comes from its row entries,
// the camera pose from LevelArgs::poses — a variant of its own, the one-frame kernels carry none of it.
// VAR: which code the variant carries. Register allocation of the traversal loops is sensitive to everything else in
// the kernel (the area-light code alone costs the point-light case 3.5 %, the general variant spills 280 bytes per
// lane — at 4K that scratch write-back was 40 % of a C5 frame's HBM traffic), so the common cases get kernels of their own:
//   0  everything: spheres, area lights with soft shadows, normal interpolation, textures (and the counting kernel)
//   1  lean: point lights only (or soft shadows off), triangles only, flat normals, no textures (C2, C3)
//   2  lean + hit attributes: as 1 with normal interpolation and / or textures (C5)
// (A fourth variant — area lights on triangles with flat normals, C4 — measured the same as variant 0: 2.456 vs 2.444 ms.)
template <bool COUNT, int VAR, bool MULTI = false>
__global__ __launch_bounds__(kBlock, RT_WAVES_PER_SIMD) void k_trace_level(const LevelArgs a)
{
    constexpr bool LEAN = VAR == 1 || VAR == 2; // no area-light code
    constexpr bool NOSPH = VAR != 0;            // triangles only
    constexpr bool FLAT = VAR == 1;             // flat normals, no textures
    const uint32_t lane = lane_id();
    const uint32_t wave = uni(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    const SceneView& sv = a.sv;
    const uint32_t flags = a.flags;
    const bool use_bvh = (flags & F_ACCEL) != 0;
    const bool prune = (flags & F_NO_PRUNE) == 0;

    // ---- fetch this lane's ray ------------------------------------------------
    bool active = false, would_cull = false;
    RayL ray;
    ray.o = mk(0, 0, 0), ray.d = mk(0, 0, -1);
    float t = kFltMax;
    uint32_t item = 0, pixel = 0;
    if (a.source == SRC_CAMERA) {
        int32_t x, y, lrow;
        active = tile_pixel(a.cam, wave, lane, x, y, lrow);
        if (active)
            item = (uint32_t)lrow * (uint32_t)a.cam.vwidth + (uint32_t)x; // x: virtual column (pixel * spp + sample)
        CameraArgs cam = a.cam;
        if (MULTI) { // the tile's frame (wave-uniform: frame_rows is a multiple of the tile height) and its pose
            const uint64_t m = ballot(active);
            if (m == 0)
                return;
            const uint32_t f = uni((uint32_t)read_lane((uint32_t)y, (uint32_t)__builtin_ctzll(m)) >> 16);
            y &= 0xffff;
            const CameraPose& ps = a.poses[f < (uint32_t)kMaxFramesPerLaunch ? f : 0u];
#pragma unroll
            for (int k = 0; k < 3; k++)
                cam.origin[k] = ps.origin[k];
#pragma unroll
            for (int k = 0; k < 4; k++)
                cam.quat[k] = ps.quat[k];
        }
        if (!COUNT && a.tile_cull && a.level == 0 && cam.spp == 1 && cam.dof_n == 0 && use_bvh && prune
            && sv.prune_ok != 0u && tile_misses_scene(cam, sv, active, x, y)) {
            if (active)
                a.state[item] = 0u; // every primary ray of the tile misses: vec3(0) (render.cpp:149-150)
            return;
        }
        if (COUNT && a.tile_cull_count && a.level == 0 && cam.spp == 1 && cam.dof_n == 0 && use_bvh && prune
            && sv.prune_ok != 0u)
            would_cull = tile_misses_scene(cam, sv, active, x, y); // counted, and traced all the same
        if (active) {
            ray = camera_ray(cam, x, y);
            pixel = (uint32_t)y * (uint32_t)cam.vwidth + (uint32_t)x;  // RNG key of the light samples
        }
    } else {
        const uint32_t count = a.source == SRC_QUEUE ? *a.in_count : a.in_count_imm;
        // Sparse levels: give each wave only R < 64 rays, so that the wave's other
        // lanes start idle and share those rays' trees by work stealing (rt_device.h).
        // R is the largest power of two that still leaves >= min_waves busy waves,
        // raised again if the grid would not cover the queue.
        const uint32_t n_waves = gridDim.x * (kBlock / 64);
        uint32_t R = 64u;
        while (R > kMinRaysPerWave && (count + R - 1u) / R < a.min_waves)
            R >>= 1;
        while ((uint64_t)R * n_waves < count)
            R <<= 1;
        const uint32_t i = wave * R + lane;
        active = lane < R && i < count;
        if (active) {
            const float* q = a.in.ray;
            const size_t cap = a.in.cap;
            ray.o = mk(q[i], q[cap + i], q[2 * cap + i]);
            ray.d = mk(q[3 * cap + i], q[4 * cap + i], q[5 * cap + i]);
            if (a.in.t)
                t = a.in.t[i];
            item = a.in.item ? a.in.item[i] : i;
            pixel = a.in.pixel ? a.in.pixel[i] : i;
        }
    }
#ifdef RT_TIMELINE // diagnostic: (start, end, HW_ID) of every wave, 100 MHz constant clock
    const unsigned long long tl_begin = wall_clock64();
    struct TimelineEnd {
        unsigned long long* p;
        unsigned long long t0;
        uint32_t wave, lane;
        const LaneCounters* lc;
        __device__ ~TimelineEnd()
        {
            if (p && lc && lane == 0 && wave < (1u << 17)) { // lc == NULL: the wave left before it had any work
                uint32_t hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                uint32_t xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                p[3 * (size_t)wave] = t0;
                p[3 * (size_t)wave + 1] = wall_clock64();
                // word 2: hw id (16 bits of it) | xcc | node iterations | leaf steps | traversals
                p[3 * (size_t)wave + 2] = (unsigned long long)((hw >> 8) & 0x3ffu) | ((unsigned long long)(xcc & 0xfu) << 10)
                    | ((unsigned long long)(lc->tl_node_iters & 0xfffffu) << 14)
                    | ((unsigned long long)(lc->tl_leaf_steps & 0xfffffu) << 34)
                    | ((unsigned long long)(lc->tl_traversals & 0x3ffu) << 54);
            }
        }
    } tl_end { a.sv.stamps, tl_begin, wave, lane, nullptr };
#endif
    if (ballot(active) == 0)
        return;

    LaneCounters lc;
#ifdef RT_TIMELINE
    tl_end.lc = &lc;
#endif
    RayCounts rc;
    if (COUNT && active)
        rc.closest = 1, rc.culled = would_cull ? 1u : 0u;
    __shared__ WaveLds lds[kBlock / 64];
    WaveLds& wl = lds[threadIdx.x >> 6];

    const int32_t first_level = a.level;
    int32_t level = first_level;
    MergeNext mg;
    mg.refl.o = mk(0, 0, 0), mg.refl.d = mk(0, 0, 0);
    // merged traversals (traverse_pair) need a fused launch; they are taken level by level where the rays fit
    mg.enabled = a.fused != 0u && a.merge_next != 0u;
    for (;; level++) {
    // ---- closest hit (bvh.intersect, render.cpp:31) — unless the previous level's merged
    //      traversal (MergeNext) already found it
    int32_t pos;
    if (mg.done) {
        pos = mg.pos;
        t = mg.t;
        mg.done = false;
    } else {
        traverse<Q_CLOSEST, COUNT, NOSPH>(sv, use_bvh, prune, active, ray, t, pos, lc, wl);
    }
    const bool hit = active && pos >= 0;
    if (active) {
        if (hit)
            a.state[item] = (uint32_t)level + 1u;
        else if (level == 0)
            a.state[item] = 0u;
        else
            a.state[item] |= 0x100u; // the reflected ray left the scene: Lo += vec3(0) twice
    }

    HitL h;
    h.normal = mk(0, 0, 1), h.kd = mk(0, 0, 0), h.ks = mk(0, 0, 0);
    h.shininess = 1.0f, h.transparency = 1.0f, h.texture = -1;
    if (hit)
        resolve_hit<FLAT>(sv, flags, ray, t, pos, h);
    if (COUNT && hit)
        rc.hits = 1;

    // ---- reflection spawn (render.cpp:34,52-54,100,113-119): known before the lights are
    //      evaluated, so its ray can share a traversal with this level's shadow rays
    bool spawn = false;
    RayL refl;
    refl.o = mk(0, 0, 0), refl.d = mk(0, 0, 0);
    if (hit && (flags & F_RECURSIVE) && level < a.depth)
        spawn = compute_reflection_ray(ray, t, h, refl);
    mg.spawn = spawn, mg.refl = refl;

    const V3 Lo = shade_hit<COUNT, false, LEAN, NOSPH>(a, hit, ray, t, pos, level, pixel, h, lc, rc, wl, &mg);
    if (hit) {
        float* d = a.direct + (size_t)(level - first_level) * a.level_stride + 3 * (size_t)item;
        d[0] = Lo.x, d[1] = Lo.y, d[2] = Lo.z;
    }
    const uint64_t smask = ballot(spawn);
    if (COUNT && spawn)
        rc.queued = 1;
    if (COUNT) {
        flush_counters(a.counters, lc, rc, level, lane);
        lc = LaneCounters();
        rc = RayCounts();
    }
    if (a.fused) {
        // The wave keeps its own reflection rays and traces them at once: lanes whose
        // path ended idle and pick up subtrees of the others by work stealing. No global
        // queue, no launch per level, and deep sparse levels overlap with everything else.
        if (smask == 0)
            break;
        active = spawn;
        ray = refl;
        t = kFltMax;
        if (COUNT && active)
            rc.closest = 1;
        continue;
    }
    if (smask) {
        uint32_t base = 0;
        if (lane == 0)
            base = atomicAdd(a.out_count, (uint32_t)__builtin_popcountll(smask));
        base = uni(base);
        if (spawn) {
            const uint32_t o = base + prefix_count(smask);
            if (o < a.out.cap) {
                float* q = a.out.ray;
                const size_t cap = a.out.cap;
                q[o] = refl.o.x, q[cap + o] = refl.o.y, q[2 * cap + o] = refl.o.z;
                q[3 * cap + o] = refl.d.x, q[4 * cap + o] = refl.d.y, q[5 * cap + o] = refl.d.z;
                a.out.item[o] = item;
                a.out.pixel[o] = pixel;
            }
        }
    }
    break;
    }
}

__global__ __launch_bounds__(256) void k_fold(const FoldArgs a)
{
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    if (item >= a.n_items)
        return;
    uint32_t s = 0u;
    bool traced = true;
    if (a.tiles_w) { // only the tiles inside the scene's screen rectangle were launched
        const uint32_t lrow = item / a.vwidth, xv = item - lrow * a.vwidth;
        const uint32_t tx = xv >> a.tile_wlog, ty = lrow >> a.tile_hlog;
        traced = tx >= a.tile_x0 && tx - a.tile_x0 < a.tiles_w && ty >= a.tile_y0 && ty - a.tile_y0 < a.tiles_h;
    }
    if (traced)
        s = a.state[item];
    const int32_t n = (int32_t)(s & 0xffu);
    V3 c = mk(0.0f, 0.0f, 0.0f);
    if (n > 0) {
        c = ld3(a.direct + (size_t)(n - 1) * a.level_stride + 3 * (size_t)item);
        if (s & 0x100u) { // deepest reflection missed: Lo += vec3(0), twice (render.cpp:100,118,148)
            c = c + mk(0.0f, 0.0f, 0.0f);
            c = c + mk(0.0f, 0.0f, 0.0f);
        }
        for (int32_t j = n - 2; j >= 0; j--) {
            const V3 d = ld3(a.direct + (size_t)j * a.level_stride + 3 * (size_t)item);
            c = (d + c) + c;
        }
    }
    // render.cpp:283,318,322: colorSum = vec3(0) + colour; finalColor = colorSum / float(1)
    V3 colorSum = mk(0.0f, 0.0f, 0.0f) + c;
    const V3 finalColor = colorSum / 1.0f;
    size_t o = item;
    if (a.width > 0 && !a.packed) {
        const int32_t lrow = (int32_t)(item / (uint32_t)a.width);
        const int32_t x = (int32_t)(item - (uint32_t)lrow * (uint32_t)a.width);
        int32_t y = a.rows ? a.rows[lrow] : a.row0 + lrow;
        if (a.n_frames > 1)
            y &= 0xffff;
        o = (size_t)(a.height - 1 - y) * a.width + x; // Screen::setPixel, screen.cpp:45
    }
    float* p = a.out + 3 * o;
    if (a.n_frames > 1) { // frame f of the launch: its items follow frame f - 1's, its pixels start at f * frame_stride
        const uint32_t f = item / a.frame_items;
        if (a.width > 0 && a.packed)
            p -= 3 * (size_t)f * a.frame_items;
        p += (size_t)f * a.frame_stride;
    }
    p[0] = finalColor.x, p[1] = finalColor.y, p[2] = finalColor.z;
}

// renderRayTracing's per-pixel combination of its samples (src/render.cpp:292-322)
__global__ __launch_bounds__(256) void k_resolve(const ResolveArgs a)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= a.n_pixels)
        return;
    const float* sm = a.samples + 3 * (size_t)p * (size_t)a.spp;
    V3 colorSum = mk(0.f, 0.f, 0.f);
    uint32_t weight = 0;
    int32_t s = 0;
    if (a.ms_n > 0) {
        V3 color = mk(0.f, 0.f, 0.f);
        for (int32_t i = 0; i < a.ms_n * a.ms_n; i++, s++)
            color = color + ld3(sm + 3 * s);
        color = color / float(a.ms_n * a.ms_n);
        colorSum = colorSum + color;
        weight++;
    }
    if (a.dof_n > 0) {
        V3 color = mk(0.f, 0.f, 0.f);
        for (int32_t i = 0; i < a.dof_n; i++, s++)
            color = color + ld3(sm + 3 * s);
        color = color / float(a.dof_n);
        colorSum = colorSum + color * 3.f;
        weight += 3;
    }
    if (a.ms_n == 0 && a.dof_n == 0) {
        colorSum = colorSum + ld3(sm);
        weight++;
    }
    const V3 finalColor = colorSum / float(weight);
    size_t o = p;
    if (!a.packed) {
        const int32_t lrow = (int32_t)(p / (uint32_t)a.width);
        const int32_t x = (int32_t)(p - (uint32_t)lrow * (uint32_t)a.width);
        const int32_t y = a.rows ? a.rows[lrow] : a.row0 + lrow;
        o = (size_t)(a.height - 1 - y) * a.width + x; // Screen::setPixel, screen.cpp:45
    }
    a.out[3 * o] = finalColor.x, a.out[3 * o + 1] = finalColor.y, a.out[3 * o + 2] = finalColor.z;
}

// renderBloomFilter (src/render.cpp:157-196): one thread per pixel (x, y), x < W - 1, y < H - 1
__global__ __launch_bounds__(256) void k_bloom(const BloomArgs a)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const int32_t W = a.width, H = a.height;
    if (id >= (uint32_t)(W - 1) * (uint32_t)(H - 1))
        return;
    const int32_t y = (int32_t)(id / (uint32_t)(W - 1)), x = (int32_t)(id - (uint32_t)y * (uint32_t)(W - 1));
    const size_t idx = (size_t)(H - 1 - y) * W + x; // Screen::indexAt
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    for (int k = -1; k < 2; k++) {
        for (int j = -1; j < 2; j++) {
            if (!(x + k < 0 || x + k > W - 1 || y + j < 0 || y + j > H - 1)) {
                const size_t ti = (size_t)(H - 1 - (y + j)) * W + (x + k);
                V3 c = ld3(a.src + 3 * ti);
                // threshold pass (:165-170): brightness in double, kept as float
                const float brightness = (float)(0.2126 * (double)c.x + 0.7152 * (double)c.y + 0.0722 * (double)c.z);
                if (brightness < a.threshold)
                    c = mk(0.0f, 0.0f, 0.0f);
                sum = sum + c * a.w[k + 1][j + 1];
            }
        }
    }
    const V3 orig = ld3(a.src + 3 * idx);
    const V3 newColor = orig + sum * a.scalar;
    const V3 outc = a.debug_option == 0 ? newColor : (a.debug_option == 1 ? sum * a.scalar : orig);
    a.frame[3 * idx] = outc.x, a.frame[3 * idx + 1] = outc.y, a.frame[3 * idx + 2] = outc.z;
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_intersect(const IntersectArgs a)
{
    const uint32_t lane = lane_id();
    const uint32_t wave = uni(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    bool active = false;
    RayL ray;
    ray.o = mk(0, 0, 0), ray.d = mk(0, 0, -1);
    float t = kFltMax;
    size_t out_idx = 0;
    if (a.source == SRC_CAMERA) {
        int32_t x, y, lrow;
        active = tile_pixel(a.cam, wave, lane, x, y, lrow);
        if (active) {
            ray = camera_ray(a.cam, x, y);
            out_idx = (size_t)(a.cam.height - 1 - y) * a.cam.width + x;
        }
    } else {
        const uint32_t i = wave * 64u + lane;
        active = i < a.n_rays;
        if (active) {
            const float* q = a.in.ray;
            const size_t cap = a.in.cap;
            ray.o = mk(q[i], q[cap + i], q[2 * cap + i]);
            ray.d = mk(q[3 * cap + i], q[4 * cap + i], q[5 * cap + i]);
            t = a.in.t[i];
            out_idx = i;
        }
    }
    if (ballot(active) == 0)
        return;
    LaneCounters lc;
    RayCounts rc;
    if (COUNT && active)
        rc.closest = 1;
    __shared__ WaveLds lds[kBlock / 64];
    int32_t pos;
    traverse<Q_CLOSEST, COUNT>(a.sv, (a.flags & F_ACCEL) != 0, (a.flags & F_NO_PRUNE) == 0, active, ray, t, pos, lc,
        lds[threadIdx.x >> 6]);
    if (active) {
        const bool hit = pos >= 0;
        HitL h;
        h.normal = mk(0, 0, 0), h.kd = mk(0, 0, 0), h.ks = mk(0, 0, 0);
        h.shininess = 1.0f, h.transparency = 1.0f, h.texture = -1;
        int32_t prim = -1;
        if (hit)
            prim = resolve_hit(a.sv, a.flags, ray, t, pos, h);
        if (a.t_bits)
            a.t_bits[out_idx] = __float_as_uint(t);
        if (a.prim)
            a.prim[out_idx] = prim;
        if (a.t_out)
            a.t_out[out_idx] = t;
        if (a.hits) {
            rt_hit o;
            o.hit = hit ? 1 : 0;
            o.prim = prim;
            o.t = t;
            o.normal[0] = h.normal.x, o.normal[1] = h.normal.y, o.normal[2] = h.normal.z;
            o.material.kd[0] = h.kd.x, o.material.kd[1] = h.kd.y, o.material.kd[2] = h.kd.z;
            o.material.ks[0] = h.ks.x, o.material.ks[1] = h.ks.y, o.material.ks[2] = h.ks.z;
            o.material.shininess = h.shininess;
            o.material.transparency = h.transparency;
            o.material.texture = h.texture;
            a.hits[out_idx] = o;
        }
    }
    if (COUNT)
        flush_counters(a.counters, lc, rc, 0, lane);
}

// gathered: n_ranks blocks of batch frames of rows_max*W*3 floats; strip s (strip_rows image rows) of
// every frame was rendered by rank s % n_ranks as its local strip s / n_ranks. The first n_frames
// frames of the batch are written, row-flipped, to frames[n_frames][H][W][3]. One thread moves VEC
// floats of a row (VEC = 4 when a row is a whole number of float4s and the pointers are aligned).
template <int VEC>
__global__ __launch_bounds__(256) void k_deinterleave(const float* __restrict__ gathered, float* __restrict__ frames,
    int32_t width, int32_t height, int32_t strip_rows, int32_t n_ranks, int32_t rows_max, int32_t n_frames,
    int32_t batch)
{
    const size_t row_units = (size_t)width * 3 / VEC;
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x; // (frame, y, unit)
    if (idx >= row_units * height * n_frames)
        return;
    const int32_t fy = (int32_t)(idx / row_units);
    const size_t u = idx - (size_t)fy * row_units;
    const int32_t b = fy / height;
    const int32_t y = fy - b * height;
    const int32_t strip = y / strip_rows;
    const int32_t rank = strip % n_ranks;
    const int32_t lrow = (strip / n_ranks) * strip_rows + (y - strip * strip_rows);
    const float* src = gathered + (((size_t)rank * batch + b) * rows_max + lrow) * (size_t)width * 3 + VEC * u;
    float* dst = frames + ((size_t)b * height + (height - 1 - y)) * (size_t)width * 3 + VEC * u;
    if (VEC == 4)
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    else
        dst[0] = src[0];
}

// rt_debug_div_check: div_by_ray(n, d, RN(1/d)) against the device's own n / d on
// pseudo-random operands. mode 0: raw random bit patterns (extremes, NaN/inf →
// guard paths); 1: moderate exponents (the fast path); 2: d with an all-ones
// significand; 3: slab-test shaped operands (n = bound - origin with cancellation).
__global__ __launch_bounds__(256) void k_div_check(uint32_t seed, uint32_t mode, uint32_t per_thread,
    unsigned long long* out /* [0] mismatches, [1] fast-path count, [2],[3] first mismatching n, d bits */)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t h = pcg_hash(seed ^ (tid * 0x9E3779B9u));
    unsigned long long bad = 0, fast = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        h = pcg_hash(h + i);
        uint32_t nb = h;
        h = pcg_hash(h ^ 0x85ebca6bu);
        uint32_t db = h;
        if (mode == 1u) { // exponents in [2^-20, 2^20)
            nb = (nb & 0x807fffffu) | ((107u + (nb >> 23) % 40u) << 23);
            db = (db & 0x807fffffu) | ((107u + (db >> 23) % 40u) << 23);
        } else if (mode == 2u) {
            nb = (nb & 0x807fffffu) | ((107u + (nb >> 23) % 40u) << 23);
            db = (db & 0x80000000u) | ((107u + (db >> 23) % 40u) << 23) | 0x7fffffu;
        }
        float n = __uint_as_float(nb), d = __uint_as_float(db);
        if (mode == 3u) {
            const float o = (float)(int32_t)(nb & 0xffffu) / 4096.0f - 8.0f;
            const float b = o + (float)(int32_t)((nb >> 16) & 0xffu) * __uint_as_float(0x33800000u + ((db & 7u) << 23));
            n = b - o;
            d = __uint_as_float((db & 0x807fffffu) | ((100u + (db >> 23) % 28u) << 23));
        }
        if (d == 0.0f)
            continue;
        RayL r;
        r.o = mk(0, 0, 0), r.d = mk(d, d, d);
        const RayInv inv = make_ray_inv(r, 1.0f);
        const float q = div_by_ray(n, d, inv.y[0], inv.ok(0));
        const float ref = n / d;
        const uint32_t ne = (__float_as_uint(n) >> 23) & 0xffu;
        if (inv.ok(0) && ne >= 127u - 60u && ne <= 127u + 60u)
            fast++;
        const bool same = __float_as_uint(q) == __float_as_uint(ref) || (q != q && ref != ref);
        if (!same) {
            if (bad == 0 && atomicCAS(&out[2], 0ull, (unsigned long long)__float_as_uint(n) | 1ull << 63) == 0ull)
                out[3] = __float_as_uint(d);
            bad++;
        }
    }
    if (bad)
        atomicAdd(&out[0], bad);
    atomicAdd(&out[1], fast);
}

// Dispatch order of the launched tiles, built on the device (no host round trip when the launched grid
// changes with the camera): rings around the centre of the launched grid, innermost first, as a counting
// sort — ring sizes (k_tile_ring_count), exclusive scan (k_tile_ring_scan), scatter (k_tile_ring_scatter).
// The order inside a ring is whatever the atomics give: a scheduling choice, results do not depend on it.
RT_DEV uint32_t tile_ring(const TileOrderArgs& a, uint32_t tile)
{
    const int64_t ty = tile / a.gw, tx = tile - (uint32_t)ty * a.gw;
    const int64_t dx = tx - a.cx, dy = ty % a.period - a.cy;
    const int64_t r = max((dx < 0 ? -dx : dx) * a.tw, (dy < 0 ? -dy : dy) * a.th);
    int64_t q = (r + a.step - 1) / a.step;
    q = q < (int64_t)a.n_rings - 1 ? q : (int64_t)a.n_rings - 1;
    if (a.xcd_order) // eight lists, one per XCD; in a list frame after frame (a frame's share of the scene fits the
                     // XCD's L2, eight frames' shares do not), each frame centre first
        q += ((int64_t)tile_xcd((uint32_t)tx + a.tile_x0, (uint32_t)(ty % a.period) + a.tile_y0, a.xcd_bxlog, a.xcd_bylog)
                     * a.n_frames + (a.frame_major ? ty / a.period : 0)) * a.n_rings;
    return (uint32_t)q;
}
// One atomic per (wave, ring) instead of one per tile: the 64 consecutive tiles of a wave lie in a handful of
// rings, and hundreds of thousands of single adds to a few hundred counters serialise in their L2 channels
// (~50 ns each). `base`: what the leader's add returned (scatter) — lanes of the same ring take consecutive places.
RT_DEV uint32_t ring_add(uint32_t* rings, bool valid, uint32_t ring)
{
    uint32_t place = 0;
    uint64_t todo = ballot(valid);
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t r0 = read_lane(ring, leader);
        const uint64_t same = ballot(valid && ring == r0) & todo;
        uint32_t base = 0;
        if (lane_id() == leader)
            base = atomicAdd(&rings[r0], (uint32_t)__builtin_popcountll(same));
        base = read_lane(base, leader);
        if (valid && ring == r0)
            place = base + prefix_count(same);
        todo &= ~same;
    }
    return place;
}
__global__ __launch_bounds__(256) void k_tile_ring_count(const TileOrderArgs a)
{
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    const bool valid = tile < a.n_tiles;
    ring_add(a.rings, valid, valid ? tile_ring(a, tile) : 0u);
}
__global__ __launch_bounds__(1024) void k_tile_ring_scan(const TileOrderArgs a)
{
    __shared__ uint32_t part[1024];
    const uint32_t n_keys = a.n_rings * (a.xcd_order ? 8u * a.n_frames : 1u);
    const uint32_t per = (n_keys + 1023u) / 1024u, b = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t i = b; i < b + per && i < n_keys; i++)
        sum += a.rings[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = b; i < b + per && i < n_keys; i++) {
        const uint32_t c = a.rings[i];
        a.rings[i] = run;
        run += c;
    }
}
__global__ __launch_bounds__(256) void k_tile_ring_scatter(const TileOrderArgs a)
{
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    const bool valid = tile < a.n_tiles;
    const uint32_t place = ring_add(a.rings, valid, valid ? tile_ring(a, tile) : 0u);
    if (valid)
        a.order[place] = tile;
}

// The same order in ONE launch: workgroup x builds list x (the tiles of XCD x's screen blocks; without the XCD
// order there is one list of all tiles) with its counters in LDS — count, scan, scatter with two barriers instead
// of two kernel boundaries and a memset (a lone frame spends ~60 us in those, tools/timeline of the serial frame).
// The places inside a ring are whatever the LDS atomics give.
__global__ __launch_bounds__(1024) void k_tile_order_lists(const TileOrderArgs a)
{
    extern __shared__ uint32_t lds_keys[];
    const uint32_t list = blockIdx.x, nt = blockDim.x, tid = threadIdx.x;
    const uint32_t kpl = a.n_rings * (a.xcd_order ? a.n_frames : 1u); // keys per list
    uint32_t* hist = lds_keys;
    uint32_t* part = lds_keys + kpl;
    if (list == 0 && a.zero)
        for (uint32_t i = tid; i < a.n_zero; i += nt)
            a.zero[i] = 0u;
    for (uint32_t i = tid; i < kpl; i += nt)
        hist[i] = 0u;
    __syncthreads();
    // the list's tiles: in tile row ty the blocks whose column bx has (bx + 3 by) % 8 == list, i.e. every eighth
    const uint32_t gh = a.n_tiles / a.gw;
    const uint32_t B = 1u << a.xcd_bxlog, bx_lo = a.tile_x0 >> a.xcd_bxlog;
    const uint32_t nb = ((a.tile_x0 + a.gw - 1u) >> a.xcd_bxlog) - bx_lo + 1u;
    const uint32_t per_row = a.xcd_order ? ((nb + 7u) / 8u) << a.xcd_bxlog : a.gw;
    const uint32_t items = gh * per_row;
    auto tile_of = [&](uint32_t it, uint32_t& tile) {
        const uint32_t ty = it / per_row, r = it - ty * per_row;
        if (!a.xcd_order) {
            tile = it;
            return true;
        }
        const uint32_t by = ((uint32_t)(ty % (uint32_t)a.period) + a.tile_y0) >> a.xcd_bylog;
        const uint32_t bx = bx_lo + ((list - 3u * by - bx_lo) & 7u) + 8u * (r >> a.xcd_bxlog);
        const uint32_t atx = (bx << a.xcd_bxlog) + (r & (B - 1u));
        tile = ty * a.gw + (atx - a.tile_x0);
        return atx >= a.tile_x0 && atx < a.tile_x0 + a.gw;
    };
    const uint32_t key0 = a.xcd_order ? list * kpl : 0u;
    for (uint32_t it = tid; it < items; it += nt) {
        uint32_t tile;
        if (tile_of(it, tile)) {
            const uint32_t k = tile_ring(a, tile) - key0;
            if (k < kpl)
                atomicAdd(&hist[k], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the kpl counters
    const uint32_t per = (kpl + nt - 1u) / nt, b = tid * per;
    uint32_t sum = 0;
    for (uint32_t i = b; i < b + per && i < kpl; i++)
        sum += hist[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < nt; o <<= 1) {
        const uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum + a.list_base[list];
    for (uint32_t i = b; i < b + per && i < kpl; i++) {
        const uint32_t c = hist[i];
        hist[i] = run;
        run += c;
    }
    __syncthreads();
    for (uint32_t it = tid; it < items; it += nt) {
        uint32_t tile;
        if (tile_of(it, tile)) {
            const uint32_t k = tile_ring(a, tile) - key0;
            const uint32_t place = k < kpl ? atomicAdd(&hist[k], 1u) : a.n_tiles;
            if (place < a.n_tiles) // holds by construction (the host's list lengths are the lists'); never write outside
                a.order[place] = tile;
        }
    }
}

uint32_t grid_for_waves(uint64_t waves) { return (uint32_t)((waves + (kBlock / 64) - 1) / (kBlock / 64)); }

uint64_t camera_waves(const CameraArgs& c)
{
    const uint32_t tw = 1u << c.tile_wlog, th = 1u << c.tile_hlog;
    const uint64_t tiles_x = (c.vwidth + tw - 1) / tw, tiles_y = (c.n_rows + th - 1) / th;
    if (c.super_log == 0)
        return c.order && c.xcd_order ? (uint64_t)c.launch_waves : (c.tiles_w ? (uint64_t)c.tiles_w * c.tiles_h : tiles_x * tiles_y);
    const uint32_t s = c.super_log;
    const uint64_t chunks = ((tiles_x + (1u << s) - 1) >> s) * ((tiles_y + (1u << s) - 1) >> s);
    return ((chunks + 7) / 8 * 8) << (2 * s); // whole super-tiles, a multiple of 8 of them (tile_pixel)
}

} // namespace

#if defined(RT_DIAG) && RT_DIAG == 2
void set_diag_leaf(const float4* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_leaf), &p, sizeof(p)); }
#endif
uint32_t g_lds_pad = 0; // tuning experiment: extra dynamic LDS per block lowers occupancy

uint32_t trace_level_groups(const LevelArgs& a)
{
    uint64_t waves;
    if (a.source == SRC_CAMERA)
        waves = camera_waves(a.cam);
    else if (a.source == SRC_QUEUE)
        waves = ((uint64_t)a.in.cap + 63) / 64;
    else
        waves = ((uint64_t)a.in_count_imm + 63) / 64;
    return waves ? grid_for_waves(waves) : 0u;
}

void launch_trace_level(const LevelArgs& a, bool count, hipStream_t s)
{
    uint64_t waves;
    if (a.source == SRC_CAMERA)
        waves = camera_waves(a.cam);
    else if (a.source == SRC_QUEUE)
        waves = ((uint64_t)a.in.cap + 63) / 64; // worst case; waves past *in_count exit at once
    else
        waves = ((uint64_t)a.in_count_imm + 63) / 64;
    if (waves == 0)
        return;
    const uint32_t grid = grid_for_waves(waves);
    const bool multi = a.source == SRC_CAMERA && a.cam.n_frames > 1;
#define RT_LAUNCH_TRACE(VAR)                                                                                              \
    do {                                                                                                                  \
        if (multi)                                                                                                        \
            hipLaunchKernelGGL((k_trace_level<false, VAR, true>), dim3(grid), dim3(kBlock), g_lds_pad, s, a);            \
        else                                                                                                              \
            hipLaunchKernelGGL((k_trace_level<false, VAR, false>), dim3(grid), dim3(kBlock), g_lds_pad, s, a);           \
    } while (0)
    if (count)
        hipLaunchKernelGGL((k_trace_level<true, 0>), dim3(grid), dim3(kBlock), 0, s, a);
    else if (a.variant == 1)
        RT_LAUNCH_TRACE(1);
    else if (a.variant == 2)
        RT_LAUNCH_TRACE(2);
    else
        RT_LAUNCH_TRACE(0);
#undef RT_LAUNCH_TRACE
}

uint32_t trace_general_waves(const LevelArgs& a)
{
    uint64_t units = a.source == SRC_CAMERA ? camera_waves(a.cam) : ((uint64_t)a.in_count_imm + 63) / 64;
    const uint64_t cap = 8192; // frame stacks are allocated per wave slot, not per pixel
    return (uint32_t)(units < cap ? units : cap);
}

void launch_trace_general(const LevelArgs& a, bool count, hipStream_t s)
{
    const uint32_t waves = trace_general_waves(a);
    if (waves == 0)
        return;
    const uint32_t grid = grid_for_waves(waves);
    if (count)
        hipLaunchKernelGGL(k_trace_general<true>, dim3(grid), dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(k_trace_general<false>, dim3(grid), dim3(kBlock), 0, s, a);
}

void launch_tile_order(const TileOrderArgs& a, hipStream_t s)
{
    if (a.n_tiles == 0)
        return;
    if (a.one_launch) {
        // a lone frame gets a whole CU's worth of threads; several frames per launch run beside other launches' trace
        // kernels, where a workgroup of 16 waves would wait for a CU to drain
        const uint32_t nt = a.n_frames > 1 ? 256u : 1024u;
        const uint32_t kpl = a.n_rings * (a.xcd_order ? a.n_frames : 1u);
        hipLaunchKernelGGL(k_tile_order_lists, dim3(a.xcd_order ? 8u : 1u), dim3(nt), (kpl + nt) * sizeof(uint32_t), s, a);
        return;
    }
    (void)hipMemsetAsync(a.rings, 0, (size_t)a.n_rings * (a.xcd_order ? 8u * a.n_frames : 1u) * sizeof(uint32_t), s);
    const uint32_t grid = (a.n_tiles + 255u) / 256u;
    hipLaunchKernelGGL(k_tile_ring_count, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_tile_ring_scan, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_tile_ring_scatter, dim3(grid), dim3(256), 0, s, a);
}

void launch_fold(const FoldArgs& a, hipStream_t s)
{
    if (a.n_items == 0)
        return;
    hipLaunchKernelGGL(k_fold, dim3((a.n_items + 255) / 256), dim3(256), 0, s, a);
}

void launch_resolve(const ResolveArgs& a, hipStream_t s)
{
    if (a.n_pixels)
        hipLaunchKernelGGL(k_resolve, dim3((a.n_pixels + 255) / 256), dim3(256), 0, s, a);
}

void launch_bloom(const BloomArgs& a, hipStream_t s)
{
    if (a.width > 1 && a.height > 1) {
        const uint32_t n = (uint32_t)(a.width - 1) * (uint32_t)(a.height - 1);
        hipLaunchKernelGGL(k_bloom, dim3((n + 255) / 256), dim3(256), 0, s, a);
    }
}

void launch_intersect(const IntersectArgs& a, bool count, hipStream_t s)
{
    const uint64_t waves = a.source == SRC_CAMERA ? camera_waves(a.cam) : ((uint64_t)a.n_rays + 63) / 64;
    if (waves == 0)
        return;
    const uint32_t grid = grid_for_waves(waves);
    if (count)
        hipLaunchKernelGGL(k_intersect<true>, dim3(grid), dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(k_intersect<false>, dim3(grid), dim3(kBlock), 0, s, a);
}

void launch_div_check(uint32_t seed, uint32_t mode, uint32_t blocks, uint32_t per_thread, unsigned long long* out,
    hipStream_t s)
{
    hipLaunchKernelGGL(k_div_check, dim3(blocks), dim3(256), 0, s, seed, mode, per_thread, out);
}

void launch_deinterleave(const float* gathered, float* frames, int32_t width, int32_t height, int32_t strip_rows,
    int32_t n_ranks, int32_t n_frames, int32_t batch, hipStream_t s)
{
    const int32_t n_strips = (height + strip_rows - 1) / strip_rows;
    const int32_t rows_max = ((n_strips + n_ranks - 1) / n_ranks) * strip_rows;
    const bool vec = (width * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(gathered) & 15) == 0
        && (reinterpret_cast<uintptr_t>(frames) & 15) == 0;
    const size_t n = (size_t)width * 3 / (vec ? 4 : 1) * height * n_frames;
    if (vec)
        hipLaunchKernelGGL(k_deinterleave<4>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, gathered, frames,
            width, height, strip_rows, n_ranks, rows_max, n_frames, batch);
    else
        hipLaunchKernelGGL(k_deinterleave<1>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, gathered, frames,
            width, height, strip_rows, n_ranks, rows_max, n_frames, batch);
}

} // namespace rt
