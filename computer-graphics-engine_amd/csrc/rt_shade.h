// rt_shade.h — per-lane hit attribute resolution, Phong shading, light sampling
// and reflection rays. Each function names the reference code it restates;
// operation order is the reference's (see rt_device.h on arithmetic).
#pragma once
#include "rt_traverse.h"

namespace rt {

enum : uint32_t {
    F_SHADING = 1u << 0,
    F_RECURSIVE = 1u << 1,
    F_HARD_SHADOW = 1u << 2,
    F_SOFT_SHADOW = 1u << 3,
    F_NORMAL_INTERP = 1u << 4,
    F_TEXTURE = 1u << 5,
    F_ACCEL = 1u << 6,
    F_BILINEAR = 1u << 7,
    F_NO_PRUNE = 1u << 8, // reference policy: visit every leaf whose boxes are hit (rt_set_option "unpruned")
    F_TRANSPARENCY = 1u << 9, // extra.enableTransparency (render.cpp:36-49,108-111; light.cpp:66-69)
    F_GLOSSY = 1u << 10,      // extra.enableGlossyReflection (render.cpp:56-99)
};

struct HitL { // one lane's HitInfo (src/common.h:14-19) without the unused members
    V3 normal;
    V3 kd, ks;
    float shininess, transparency;
    int32_t texture;
};

// computeBarycentricCoord — src/interpolate.cpp:4-17
RT_DEV V3 barycentric(V3 v0, V3 v1, V3 v2, V3 p)
{
    const V3 a = v1 - v0, b = v2 - v0, c = p - v0;
    const float d00 = dot(a, a), d01 = dot(a, b), d11 = dot(b, b), d20 = dot(c, a), d21 = dot(c, b);
    const float denom = d00 * d11 - d01 * d01;
    const float v = (d11 * d20 - d01 * d21) / denom;
    const float w = (d00 * d21 - d01 * d20) / denom;
    const float u = 1.0f - v - w;
    return mk(u, v, w);
}

// x86 cvttss2si semantics of the reference's float → int conversions
RT_DEV int32_t f2i_x86(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f))
        return (int32_t)0x80000000;
    return (int32_t)f;
}

// acquireTexel — src/texture.cpp:8-48 (enableTextureMapping is known set here)
RT_DEV V3 acquire_texel(const SceneView& sv, int32_t tex, float u, float v, bool bilinear)
{
    const TextureRec tr = sv.textures[tex];
    const float* px = sv.texels + 3 * (size_t)tr.offset;
    const float fx = u * (float)tr.width;
    const float fy = (1.0f - v) * (float)tr.height;
    if (!bilinear) {
        int32_t i = f2i_x86(fx < 0.0f ? 0.0f : fx); // std::max(x, 0.0f)
        int32_t j = f2i_x86(fy < 0.0f ? 0.0f : fy);
        i = i < tr.width - 1 ? i : tr.width - 1;
        j = j < tr.height - 1 ? j : tr.height - 1;
        // INT_MIN (NaN / overflow) indexes out of bounds in the reference (undefined);
        // clamp to 0 to stay memory-safe, like the oracle.
        i = i < 0 ? 0 : i;
        j = j < 0 ? 0 : j;
        return ld3(px + 3 * ((size_t)j * tr.width + i));
    }
    const float wmax = (float)tr.width - 1, hmax = (float)tr.height - 1;
    auto clampf = [](float x, float lo, float hi) { return x < lo ? lo : (hi < x ? hi : x); };
    const float yUp = clampf(ceilf(fy), 0.0f, hmax), xUp = clampf(ceilf(fx), 0.0f, wmax);
    const float yDown = clampf(floorf(fy), 0.0f, hmax), xDown = clampf(floorf(fx), 0.0f, wmax);
    const float xSplit = fmodf(fx, 1.0f), ySplit = fmodf(fy, 1.0f);
    const float w = (float)tr.width;
    const V3 xdyd = ld3(px + 3 * (size_t)(yDown * w + xDown));
    const V3 xdyu = ld3(px + 3 * (size_t)(yUp * w + xDown));
    const V3 xuyd = ld3(px + 3 * (size_t)(yDown * w + xUp));
    const V3 xuyu = ld3(px + 3 * (size_t)(yUp * w + xUp));
    const V3 x1 = (1 - xSplit) * xdyd + xSplit * xuyd;
    const V3 x2 = (1 - xSplit) * xdyu + xSplit * xuyu;
    return (1 - ySplit) * x1 + ySplit * x2;
}

// material.transparency of the primitive at array position `pos` (shadow rays with enableTransparency)
RT_DEV float hit_transparency(const SceneView& sv, uint32_t flags, int32_t pos)
{
    const PrimRecord* rec = ((flags & F_ACCEL) ? sv.prims : sv.prims_flat) + pos;
    return sv.materials[rec->kind_mat & 0x7fffffffu].transparency;
}

// Hit attribute resolution — src/bounding_volume_hierarchy.cpp:365-426.
// `pos` is the array index of the reported primitive; returns its prim id.
// LEAN: triangles only, no normal interpolation, no textures (the lean kernel variant)
template <bool LEAN = false>
RT_DEV int32_t resolve_hit(const SceneView& sv, uint32_t flags, const RayL& ray, float t, int32_t pos, HitL& h)
{
    const PrimRecord* rec = ((flags & F_ACCEL) ? sv.prims : sv.prims_flat) + pos;
    const int32_t prim_id = rec->prim_id;
    const uint32_t km = rec->kind_mat;
    const MaterialRec* m = sv.materials + (km & 0x7fffffffu);
    h.kd = ld3(m->kd), h.ks = ld3(m->ks);
    h.shininess = m->shininess, h.transparency = m->transparency, h.texture = m->texture;
    if (!LEAN && (km & 0x80000000u)) { // sphere (:400-403, :421-423)
        const V3 p = ray.o + ray.d * t;
        h.normal = normalize(p - ld3(rec->v0));
        return prim_id;
    }
    const bool interp = (flags & F_NORMAL_INTERP) != 0;
    const bool textured = (flags & F_TEXTURE) != 0 && h.texture >= 0;
    if (LEAN || (!interp && !textured)) {
        h.normal = ld3(rec->n); // normalize(cross(v2 - v1, v3 - v1)) == the plane normal (:395-397)
        return prim_id;
    }
    const V3 v0 = ld3(rec->v0), v1 = ld3(rec->v1), v2 = ld3(rec->v2);
    const V3 bc = barycentric(v0, v1, v2, ray.o + ray.d * t);
    const uint32_t* vi = sv.tri_vidx + 3 * (size_t)prim_id;
    const uint32_t i0 = vi[0], i1 = vi[1], i2 = vi[2];
    if (interp) { // :380-393 + interpolateNormal (interpolate.cpp:19-23)
        const V3 n0 = ld3(sv.vnormal + 3 * (size_t)i0), n1 = ld3(sv.vnormal + 3 * (size_t)i1),
                 n2 = ld3(sv.vnormal + 3 * (size_t)i2);
        V3 n = normalize((n0 * bc.x + n1 * bc.y + n2 * bc.z) / 3.0f);
        if (dot(n, ray.d) > 0.0f)
            n = -n;
        h.normal = n;
    } else {
        h.normal = ld3(rec->n);
    }
    if (textured) { // :413-419 + interpolateTexCoord (interpolate.cpp:25-28)
        const float* t0 = sv.vtexcoord + 2 * (size_t)i0;
        const float* t1 = sv.vtexcoord + 2 * (size_t)i1;
        const float* t2 = sv.vtexcoord + 2 * (size_t)i2;
        const float u = (bc.x * t0[0] + bc.y * t1[0]) + bc.z * t2[0];
        const float v = (bc.x * t0[1] + bc.y * t1[1]) + bc.z * t2[1];
        h.kd = acquire_texel(sv, h.texture, u, v, (flags & F_BILINEAR) != 0);
    }
    return prim_id;
}

// computeShading — src/shading.cpp:7-37 (enableShading known set)
RT_DEV V3 compute_shading(V3 lightPosition, V3 lightColor, const RayL& ray, float t, const HitL& h)
{
    const V3 newNormal = normalize(h.normal);
    const V3 light = normalize(lightPosition - (ray.d * t + ray.o));
    float d = dot(newNormal, light);
    if (d < 0.0f)
        d = 0.0f;
    const V3 diffuse = h.kd * lightColor * d;
    const V3 camera = normalize(ray.d);
    d = 0.0f;
    if (dot(newNormal, light) > 0.0f && dot(newNormal, camera) > 0.0f) {
        const V3 reflection = (2.0f * dot(light, newNormal)) * newNormal - light;
        d = powf(dot(camera, reflection), h.shininess);
    }
    const V3 specular = h.ks * lightColor * d;
    return diffuse + specular;
}

// computeReflectionRay — src/shading.cpp:40-62. Returns false for the ks == 0
// sentinel (Ray{0,0,t=0}, render.cpp:52,116), true with the reflected ray otherwise.
RT_DEV bool compute_reflection_ray(const RayL& ray, float t, const HitL& h, RayL& out)
{
    if (all_zero(h.ks))
        return false;
    const V3 point = t * ray.d + ray.o;
    const V3 n = normalize(h.normal);
    const V3 r = normalize(-ray.d);
    const V3 reflect = normalize((2.0f * dot(n, r)) * n - r);
    out.o = point + 0.00001f * n;
    out.d = reflect;
    return true;
}

// Area-light RNG (replaces rand(), src/light.cpp:21,32-33); same function as
// rt_rand31 of the C ABI.
RT_DEV uint32_t pcg_hash(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
RT_DEV uint32_t rand31(uint32_t seed, uint32_t pixel, uint32_t level, uint32_t light, uint32_t sample, uint32_t dim)
{
    uint32_t h = pcg_hash(seed ^ 0x9E3779B9u);
    h = pcg_hash(h + pixel);
    h = pcg_hash(h ^ ((level << 26) | (light << 18) | (sample << 1) | dim));
    return h & 0x7fffffffu;
}
RT_DEV float rand01(uint32_t r31) { return (float)(int32_t)r31 / 2147483648.0f; } // (float)rand() / RAND_MAX

// Shadow ray of testVisibilityLightSample — src/light.cpp:49-60:
// t *= |d|; d = normalize(d); p = o + d*(t - 1e-5); Ray{p, samplePos - p, 1}
RT_DEV RayL shadow_ray(const RayL& ray, float t, V3 samplePos)
{
    const float tl = t * length(ray.d);
    const V3 dn = normalize(ray.d);
    const V3 p = ray.o + dn * (tl - .00001f);
    RayL s;
    s.o = p;
    s.d = samplePos - p;
    return s;
}

} // namespace rt
