/*
 * ref_harness.cpp — container-only caller of the REAL reference code.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md). Compiled by oracle/Makefile
 * together with reference translation units taken BY PATH from /root/reference
 * (never copied) into oracle/_ref/libcge_ref.so. Those TUs are the ones whose
 * code is complete in source form and needs nothing the image lacks:
 *   src/scene.cpp, framework/src/mesh.cpp, framework/src/image.cpp (+ vendored
 *   tinyobjloader, stb_image, fmt), src/shading.cpp, src/interpolate.cpp,
 *   src/texture.cpp, src/light.cpp, src/bounding_volume_hierarchy.cpp,
 *   src/bvh_interface.cpp.
 *   src/bvh_interface.cpp, src/render.cpp (compiles with the same include path:
 *   render.h / screen.h / trackball.h pull in no GL header).
 * NOT built / NOT loaded: prebuilt/libIntersect_* (prebuilt machine code),
 * src/draw.cpp, src/screen.cpp, framework/src/{window,trackball}.cpp (they
 * include framework/opengl_includes.h → GL/glu.h, absent from this image). The
 * library therefore has unresolved symbols (draw*, intersectRayWith*, Screen::*,
 * Trackball::*); it is loaded with RTLD_LAZY and this harness only calls
 * functions that never reach them:
 *   - scene loading, BVH *construction* (no traversal), shading, reflection
 *     ray, barycentrics, interpolation, texel fetch, the point-light loop of
 *     computeLightContribution with shadows off, the two area-light samplers
 *     (src/light.cpp:19-45, rand() replayed through srand), weightsGaussian and
 *     getIntersection of src/render.cpp.
 * oracle/ref_bmp.cpp (its own TU) compiles the vendored stb_image_write.h by path
 * for the BMP byte layout of Screen::writeBitmapToFile (src/screen.cpp:49-60).
 * The camera functions below are the harness's own use of the vendored glm
 * (framework/src/trackball.cpp itself cannot be built): they pin the glm
 * operation order that the oracle restates by hand.
 */
#include <cstdint>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "bounding_volume_hierarchy.h"
#include "bvh_interface.h"
#include "interpolate.h"
#include "light.h"
#include "render.h"
#include "scene.h"
#include "shading.h"
#include "texture.h"
#include <framework/image.h>
#include <glm/gtc/quaternion.hpp>

#include "../include/rt_whitted.h"

// Definitions with external linkage in the reference whose headers do not declare them (light.h declares
// three-argument samplers that nothing defines; render.h omits getIntersection): declared here with the
// signatures of the definitions at src/light.cpp:19,30 and src/render.cpp:238.
void sampleSegmentLight(const SegmentLight& segmentLight, glm::vec3& position, glm::vec3& color, float index,
    float sampleSize);
void sampleParallelogramLight(const ParallelogramLight& parallelogramLight, glm::vec3& position, glm::vec3& color,
    float horizontalIndex, float verticalIndex, float sampleSizeA, float sampleSizeB);
glm::vec3 getIntersection(const Ray& ray, const Plane& plane);

static Features toFeatures(const rt_features* f)
{
    Features o;
    static_assert(sizeof(Features) == sizeof(rt_features), "Features layout");
    memcpy(&o, f, sizeof(o));
    return o;
}
static glm::vec3 g3(const float* p) { return glm::vec3(p[0], p[1], p[2]); }

extern "C" {

int ref_sizeof(int what)
{
    switch (what) {
    case 0: return (int)sizeof(Features);
    case 1: return (int)sizeof(Ray);
    case 2: return (int)sizeof(HitInfo);
    case 3: return (int)sizeof(Material);
    case 4: return (int)sizeof(Vertex);
    case 5: return (int)sizeof(Sphere);
    case 6: return (int)sizeof(AxisAlignedBox);
    }
    return -1;
}

// ---- scenes (src/scene.cpp:5-92) -------------------------------------------
Scene* ref_scene_load(int sceneType, const char* dataDir)
{
    try {
        return new Scene(loadScenePrebuilt((SceneType)sceneType, dataDir));
    } catch (...) {
        return nullptr;
    }
}
void ref_scene_free(Scene* s) { delete s; }
void ref_scene_counts(const Scene* s, int* meshes, int* spheres, int* lights)
{
    *meshes = (int)s->meshes.size();
    *spheres = (int)s->spheres.size();
    *lights = (int)s->lights.size();
}
void ref_mesh_counts(const Scene* s, int m, int* nv, int* nt, int* tw, int* th)
{
    const Mesh& mesh = s->meshes[m];
    *nv = (int)mesh.vertices.size();
    *nt = (int)mesh.triangles.size();
    *tw = mesh.material.kdTexture ? mesh.material.kdTexture->width : 0;
    *th = mesh.material.kdTexture ? mesh.material.kdTexture->height : 0;
}
// material8 = kd[3], ks[3], shininess, transparency
void ref_mesh_copy(const Scene* s, int m, float* pos, float* nrm, float* uv, uint32_t* tri, float* material8, float* texels)
{
    const Mesh& mesh = s->meshes[m];
    for (size_t i = 0; i < mesh.vertices.size(); i++) {
        memcpy(pos + 3 * i, &mesh.vertices[i].position, 12);
        memcpy(nrm + 3 * i, &mesh.vertices[i].normal, 12);
        memcpy(uv + 2 * i, &mesh.vertices[i].texCoord, 8);
    }
    for (size_t i = 0; i < mesh.triangles.size(); i++) {
        tri[3 * i] = mesh.triangles[i].x, tri[3 * i + 1] = mesh.triangles[i].y, tri[3 * i + 2] = mesh.triangles[i].z;
    }
    memcpy(material8, &mesh.material.kd, 12);
    memcpy(material8 + 3, &mesh.material.ks, 12);
    material8[6] = mesh.material.shininess;
    material8[7] = mesh.material.transparency;
    if (texels && mesh.material.kdTexture)
        memcpy(texels, mesh.material.kdTexture->pixels.data(), mesh.material.kdTexture->pixels.size() * 12);
}
void ref_sphere_copy(const Scene* s, int i, float* center_radius4, float* material8)
{
    const Sphere& sp = s->spheres[i];
    memcpy(center_radius4, &sp.center, 12);
    center_radius4[3] = sp.radius;
    memcpy(material8, &sp.material.kd, 12);
    memcpy(material8 + 3, &sp.material.ks, 12);
    material8[6] = sp.material.shininess;
    material8[7] = sp.material.transparency;
}
void ref_light_copy(const Scene* s, int i, int* kind, float* v21)
{
    memset(v21, 0, 21 * sizeof(float));
    const auto& L = s->lights[i];
    if (std::holds_alternative<PointLight>(L)) {
        *kind = RT_LIGHT_POINT;
        memcpy(v21, &std::get<PointLight>(L), sizeof(PointLight));
    } else if (std::holds_alternative<SegmentLight>(L)) {
        *kind = RT_LIGHT_SEGMENT;
        memcpy(v21, &std::get<SegmentLight>(L), sizeof(SegmentLight));
    } else {
        *kind = RT_LIGHT_PARALLELOGRAM;
        memcpy(v21, &std::get<ParallelogramLight>(L), sizeof(ParallelogramLight));
    }
}

// ---- BVH construction (src/bounding_volume_hierarchy.cpp:149-194) ----------
BoundingVolumeHierarchy* ref_bvh_build(Scene* s, const rt_features* f)
{
    Features feat = toFeatures(f);
    return new BoundingVolumeHierarchy(s, feat);
}
void ref_bvh_free(BoundingVolumeHierarchy* b) { delete b; }
void ref_bvh_counts(const BoundingVolumeHierarchy* b, int* nodes, int* prims, int* levels, int* leaves, int* root)
{
    *nodes = (int)b->nodes.size(); // private members: harness TU is built with -fno-access-control
    *prims = (int)b->primitives.size();
    *levels = b->numLevels();
    *leaves = b->numLeaves();
    *root = (int)b->root;
}
// same shape as rt_bvh_export / orc_bvh_export
void ref_bvh_export(const BoundingVolumeHierarchy* b, float* node_aabb, int32_t* node_meta, int32_t* prim_order)
{
    for (size_t i = 0; i < b->nodes.size(); i++) {
        const Node& n = b->nodes[i];
        memcpy(node_aabb + 6 * i, &n.aabb.lower, 12);
        memcpy(node_aabb + 6 * i + 3, &n.aabb.upper, 12);
        int32_t* m = node_meta + 6 * i;
        bool leaf = n.data[0] == 1;
        m[0] = leaf, m[1] = (int32_t)n.data[1], m[2] = (int32_t)n.data[2], m[3] = (int32_t)n.data[3];
        m[4] = leaf ? -1 : (int32_t)n.data[4], m[5] = leaf ? -1 : (int32_t)n.data[5];
    }
    // prim id = index in the pre-permutation list (:158-172): recover it from
    // (mesh, vertex triple); identical triples take ids in first-come order.
    const Scene* s = b->m_pScene;
    std::vector<std::map<std::tuple<size_t, size_t, size_t>, std::vector<int32_t>>> maps(s->meshes.size());
    int32_t id = 0;
    for (size_t m = 0; m < s->meshes.size(); m++)
        for (const auto& t : s->meshes[m].triangles)
            maps[m][{ t.x, t.y, t.z }].push_back(id++);
    for (auto& mm : maps)
        for (auto& kv : mm)
            std::reverse(kv.second.begin(), kv.second.end());
    for (size_t i = 0; i < b->primitives.size(); i++) {
        const Primitive& p = b->primitives[i];
        if (std::holds_alternative<TrianglePrim>(p.p)) {
            const TrianglePrim& t = std::get<TrianglePrim>(p.p);
            auto& v = maps[t.meshIdx][{ t.v1, t.v2, t.v3 }];
            prim_order[i] = v.back();
            v.pop_back();
        } else {
            prim_order[i] = id + (int32_t)std::get<SpherePrim>(p.p).sphereIdx;
        }
    }
}

// ---- shading / interpolation / texture -------------------------------------
static HitInfo makeHit(const float normal[3], const rt_material* m)
{
    HitInfo h;
    h.normal = g3(normal);
    h.material.kd = g3(m->kd);
    h.material.ks = g3(m->ks);
    h.material.shininess = m->shininess;
    h.material.transparency = m->transparency;
    return h;
}
static Ray makeRay(const rt_ray* r) { return Ray { g3(r->origin), g3(r->direction), r->t }; }

void ref_compute_shading(const float lightPos[3], const float lightColor[3], const rt_features* f, const rt_ray* ray,
    const float normal[3], const rt_material* m, float out[3])
{
    glm::vec3 c = computeShading(g3(lightPos), g3(lightColor), toFeatures(f), makeRay(ray), makeHit(normal, m));
    memcpy(out, &c, 12);
}
void ref_compute_reflection_ray(const rt_ray* ray, const float normal[3], const rt_material* m, rt_ray* out)
{
    Ray r = computeReflectionRay(makeRay(ray), makeHit(normal, m));
    memcpy(out->origin, &r.origin, 12);
    memcpy(out->direction, &r.direction, 12);
    out->t = r.t;
}
void ref_barycentric(const float v0[3], const float v1[3], const float v2[3], const float p[3], float out[3])
{
    glm::vec3 b = computeBarycentricCoord(g3(v0), g3(v1), g3(v2), g3(p));
    memcpy(out, &b, 12);
}
void ref_interpolate_normal(const float n0[3], const float n1[3], const float n2[3], const float bc[3], float out[3])
{
    glm::vec3 n = interpolateNormal(g3(n0), g3(n1), g3(n2), g3(bc));
    memcpy(out, &n, 12);
}
void ref_interpolate_texcoord(const float t0[2], const float t1[2], const float t2[2], const float bc[3], float out[2])
{
    glm::vec2 t = interpolateTexCoord(glm::vec2(t0[0], t0[1]), glm::vec2(t1[0], t1[1]), glm::vec2(t2[0], t2[1]), g3(bc));
    memcpy(out, &t, 8);
}
// texel of mesh m's kdTexture
void ref_acquire_texel(const Scene* s, int m, const float tc[2], const rt_features* f, float out[3])
{
    glm::vec3 c = acquireTexel(*s->meshes[m].material.kdTexture, glm::vec2(tc[0], tc[1]), toFeatures(f));
    memcpy(out, &c, 12);
}
// computeLightContribution (src/light.cpp:108-165) — ONLY legal with
// enableHardShadow = enableSoftShadow = 0 (anything else reaches bvh.intersect
// → the unresolved prebuilt intersection symbols).
int ref_light_contribution_noshadow(Scene* s, const rt_features* f, const rt_ray* ray, const float normal[3],
    const rt_material* m, float out[3])
{
    if (f->enableHardShadow || f->enableSoftShadow)
        return 1;
    Features feat = toFeatures(f);
    static std::map<Scene*, BvhInterface*> cache;
    auto it = cache.find(s);
    if (it == cache.end())
        it = cache.emplace(s, new BvhInterface(s, feat)).first;
    glm::vec3 c = computeLightContribution(*s, *it->second, feat, makeRay(ray), makeHit(normal, m));
    memcpy(out, &c, 12);
    return 0;
}

// ---- area-light samplers (src/light.cpp:19-45). They draw from the process-global rand():
// srand(seed) → record the draws → srand(seed) again → call the reference, so the caller knows
// exactly which rand() values the reference consumed (glibc: RAND_MAX = 2147483647).
void ref_sample_segment_light(const float v[4][3], unsigned seed, float index, float sampleSize, float pos[3],
    float col[3], int32_t* drawn)
{
    SegmentLight L { g3(v[0]), g3(v[1]), g3(v[2]), g3(v[3]) };
    srand(seed);
    drawn[0] = rand();
    srand(seed);
    glm::vec3 p, c;
    sampleSegmentLight(L, p, c, index, sampleSize);
    memcpy(pos, &p, 12);
    memcpy(col, &c, 12);
}
void ref_sample_parallelogram_light(const float v[7][3], unsigned seed, float horizontalIndex, float verticalIndex,
    float sampleSizeA, float sampleSizeB, float pos[3], float col[3], int32_t* drawn)
{
    ParallelogramLight L { g3(v[0]), g3(v[1]), g3(v[2]), g3(v[3]), g3(v[4]), g3(v[5]), g3(v[6]) };
    srand(seed);
    drawn[0] = rand();
    drawn[1] = rand();
    srand(seed);
    glm::vec3 p, c;
    sampleParallelogramLight(L, p, c, horizontalIndex, verticalIndex, sampleSizeA, sampleSizeB);
    memcpy(pos, &p, 12);
    memcpy(col, &c, 12);
}
int ref_rand_max(void) { return RAND_MAX; }

// ---- src/render.cpp: weightsGaussian (:198-210), getIntersection (:238-243) ----
void ref_weights_gaussian(float sigma, float out[9])
{
    const glm::mat3 m = weightsGaussian(sigma);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            out[3 * i + k] = m[i][k];
}
void ref_get_intersection(const rt_ray* ray, float D, const float normal[3], float out[3])
{
    Plane pl;
    pl.D = D;
    pl.normal = g3(normal);
    const glm::vec3 p = getIntersection(makeRay(ray), pl);
    memcpy(out, &p, 12);
}

// ---- camera math through the vendored glm (harness restatement of
// framework/src/trackball.cpp:23-31,64-74,101-110) ---------------------------
void ref_glm_camera(float fovy, float aspect, const float look_at[3], float distance, const float rot[3], rt_camera* out)
{
    out->half_h = std::tan(fovy / 2.0f);
    out->half_w = aspect * out->half_h;
    glm::vec3 euler = g3(rot);
    glm::quat q = glm::quat(euler);
    out->quat[0] = q.w, out->quat[1] = q.x, out->quat[2] = q.y, out->quat[3] = q.z;
    glm::vec3 p = g3(look_at) + glm::quat(euler) * glm::vec3(0, 0, -distance);
    memcpy(out->origin, &p, 12);
    memcpy(out->look_at, look_at, 12);
}
void ref_glm_generate_ray(float half_w, float half_h, const float rot[3], const float origin[3], float px, float py, rt_ray* out)
{
    const glm::vec3 cameraSpaceDirection = glm::normalize(glm::vec3(-px * half_w, py * half_h, 1.0f));
    glm::vec3 d = glm::quat(g3(rot)) * cameraSpaceDirection;
    memcpy(out->origin, origin, 12);
    memcpy(out->direction, &d, 12);
    out->t = std::numeric_limits<float>::max();
}
float ref_glm_radians(float deg) { return glm::radians(deg); }

} // extern "C"
