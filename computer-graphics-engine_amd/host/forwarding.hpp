// forwarding.hpp — the forwarding code behind the reference's render-path call surface, written against
// the REFERENCE'S OWN type names only (Scene, Mesh, Vertex, Material, Image, Sphere, the three light structs,
// Features, Ray, HitInfo, Screen). It is included by
//   reference_api.cpp       with this repo's layout-compatible re-declarations (reference_api.hpp): built and
//                           run on the GPU box, where the reference tree is absent (tests/test_host_shim.py);
//   reference_drop_in.cpp   with the reference's real headers (/root/reference/src/{render,bvh_interface,common,
//                           scene,screen,light}.h, framework/{trackball,ray,mesh,image}.h): what a maintainer
//                           compiles in place of src/render.cpp, src/bvh_interface.cpp and
//                           src/bounding_volume_hierarchy.{h,cpp}; compiled in the container by
//                           tests/test_reference_headers.py.
// No rendering arithmetic lives here: flatten the Scene once, keep opaque handles, call the C ABI.
// The including file provides, before the #include: the reference types above, and the tunables
// segmentLightSamples, parallelogramLightDirectionSamples, raysPerPixelSide, samplesDoF, focusPlaneDistance,
// blurStrength, bloomScalar, bloomThreshold, bloomDebugOption, glossyReflectionsCap, raysPerReflection,
// alphaModifier (src/light.cpp:12-13, src/render.cpp:14-25) and areaLightSeed.
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

namespace rt_fwd {

// the boundary structs are the reference's, byte for byte
static_assert(sizeof(Features) == sizeof(rt_features) && sizeof(Features) == 17, "Features ≙ rt_features");
static_assert(sizeof(Ray) == sizeof(rt_ray) && sizeof(Ray) == 28, "Ray ≙ rt_ray");
static_assert(offsetof(Ray, origin) == offsetof(rt_ray, origin) && offsetof(Ray, direction) == offsetof(rt_ray, direction)
        && offsetof(Ray, t) == offsetof(rt_ray, t),
    "Ray ≙ rt_ray");
static_assert(offsetof(Features, enableShading) == offsetof(rt_features, enableShading)
        && offsetof(Features, enableAccelStructure) == offsetof(rt_features, enableAccelStructure)
        && offsetof(Features, extra) == offsetof(rt_features, enableEnvironmentMapping),
    "Features ≙ rt_features");
// Material: kd, ks, shininess, transparency lead both structs (the reference's ends in a shared_ptr<Image>,
// the C struct in a texture index)
static_assert(offsetof(Material, kd) == offsetof(rt_material, kd) && offsetof(Material, ks) == offsetof(rt_material, ks)
        && offsetof(Material, shininess) == offsetof(rt_material, shininess)
        && offsetof(Material, transparency) == offsetof(rt_material, transparency),
    "Material ≙ rt_material (leading members)");
static_assert(sizeof(glm::vec3) == 12, "Screen::pixels() must be tightly packed fp32 RGB");

[[noreturn]] inline void fail(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}
inline void check(int rc, const char* what)
{
    if (rc != RT_OK)
        fail(what);
}

inline void put3(float* dst, const glm::vec3& v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; }

inline std::vector<rt_light_desc> flatten_lights(const Scene& scene)
{
    std::vector<rt_light_desc> out;
    for (const auto& l : scene.lights) {
        rt_light_desc d;
        memset(&d, 0, sizeof(d));
        if (std::holds_alternative<PointLight>(l)) {
            const auto& p = std::get<PointLight>(l);
            d.kind = RT_LIGHT_POINT;
            put3(d.v[0], p.position), put3(d.v[1], p.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& s = std::get<SegmentLight>(l);
            d.kind = RT_LIGHT_SEGMENT;
            put3(d.v[0], s.endpoint0), put3(d.v[1], s.endpoint1), put3(d.v[2], s.color0), put3(d.v[3], s.color1);
        } else {
            const auto& p = std::get<ParallelogramLight>(l);
            d.kind = RT_LIGHT_PARALLELOGRAM;
            put3(d.v[0], p.v0), put3(d.v[1], p.edge01), put3(d.v[2], p.edge02);
            put3(d.v[3], p.color0), put3(d.v[4], p.color1), put3(d.v[5], p.color2), put3(d.v[6], p.color3);
        }
        out.push_back(d);
    }
    return out;
}

inline rt_render_params params_for(int depth)
{
    rt_render_params p;
    p.depth = depth;
    p.parallelogram_n = parallelogramLightDirectionSamples;
    p.segment_n = segmentLightSamples;
    p.seed = areaLightSeed;
    // the extras' globals of src/render.cpp:14-22
    p.rays_per_pixel_side = raysPerPixelSide;
    p.samples_dof = samplesDoF;
    p.focus_plane_distance = focusPlaneDistance;
    p.blur_strength = blurStrength;
    p.bloom_scalar = bloomScalar;
    p.bloom_threshold = bloomThreshold;
    p.bloom_debug_option = bloomDebugOption;
    p.glossy_reflections_cap = glossyReflectionsCap;
    p.rays_per_reflection = raysPerReflection;
    p.alpha_modifier = alphaModifier;
    return p;
}

// What stands behind a BvhInterface: the device scene + BVH handles of the C ABI — one pair per shard. With
// several GPUs visible every GPU gets a copy of the scene and its BVH, and a frame is rendered by all of them
// (rt_render_sharded: strips of 8 rows dealt round-robin, gathered on the first GPU) — the reference's OpenMP row
// loop (src/render.cpp:277-281) spread over the node's GPUs. RT_WHITTED_SHARDS=n overrides the shard count
// (shard s lives on device s % visible devices; two shards on one GPU rehearse the path on a one-GPU box).
struct Handles {
    std::vector<rt_scene*> scenes; // [0] serves the single-ray calls
    std::vector<rt_bvh*> bvhs;
    rt_scene* scene = nullptr;     // = scenes[0]
    rt_bvh* bvh = nullptr;         // = bvhs[0]
    Scene* source = nullptr; // borrowed, like m_pScene (bounding_volume_hierarchy.h:86)
    std::vector<rt_light_desc> lights;
    std::vector<std::shared_ptr<Image>> textures; // texture index → the mesh's kdTexture
    std::vector<int> material_texture;            // per mesh / sphere material slot
    Handles() = default;
    Handles(const Handles&) = delete;
    Handles& operator=(const Handles&) = delete;
    ~Handles()
    {
        for (rt_bvh* b : bvhs)
            rt_bvh_destroy(b);
        for (rt_scene* sc : scenes)
            rt_scene_destroy(sc);
    }
    // the GUI edits scene.lights between frames (src/main.cpp:296-368)
    void sync_lights()
    {
        std::vector<rt_light_desc> now = flatten_lights(*source);
        if (now.size() != lights.size()
            || (!now.empty() && memcmp(now.data(), lights.data(), now.size() * sizeof(rt_light_desc)) != 0)) {
            for (rt_scene* sc : scenes)
                check(rt_scene_update_lights(sc, now.data(), (uint32_t)now.size()), "rt_scene_update_lights");
            lights.swap(now);
        }
    }
    static int shard_count(int& n_devices)
    {
        n_devices = 0;
        check(rt_device_info(&n_devices, nullptr, 0), "rt_device_info");
        int n = n_devices;
        if (const char* e = std::getenv("RT_WHITTED_SHARDS"))
            n = std::atoi(e);
        return n < 1 ? 1 : (n > 64 ? 64 : n);
    }

    // ≙ BoundingVolumeHierarchy::BoundingVolumeHierarchy(Scene*, const Features&) (bvh.cpp:149-194): flatten the
    // scene into the C ABI's descriptors, upload it, build the BVH
    void build(Scene* pScene, const Features& features)
    {
    Handles& im = *this;
    im.source = pScene;
    // SoA copies of every mesh (the C ABI wants separate position/normal/uv arrays)
    struct Flat {
        std::vector<float> pos, nrm, uv;
        std::vector<uint32_t> tri;
    };
    std::vector<Flat> flats(pScene->meshes.size());
    std::vector<rt_mesh_desc> meshes(pScene->meshes.size());
    std::vector<rt_texture_desc> textures;
    std::vector<std::vector<float>> texel_store;
    for (size_t m = 0; m < pScene->meshes.size(); m++) {
        const Mesh& mesh = pScene->meshes[m];
        Flat& f = flats[m];
        for (const Vertex& v : mesh.vertices) {
            f.pos.insert(f.pos.end(), { v.position.x, v.position.y, v.position.z });
            f.nrm.insert(f.nrm.end(), { v.normal.x, v.normal.y, v.normal.z });
            f.uv.insert(f.uv.end(), { v.texCoord.x, v.texCoord.y });
        }
        for (const glm::uvec3& t : mesh.triangles)
            f.tri.insert(f.tri.end(), { t.x, t.y, t.z });
        rt_mesh_desc& d = meshes[m];
        memset(&d, 0, sizeof(d));
        d.positions = f.pos.data(), d.normals = f.nrm.data(), d.texcoords = f.uv.data(), d.triangles = f.tri.data();
        d.n_vertices = (uint32_t)mesh.vertices.size(), d.n_triangles = (uint32_t)mesh.triangles.size();
        put3(d.material.kd, mesh.material.kd), put3(d.material.ks, mesh.material.ks);
        d.material.shininess = mesh.material.shininess;
        d.material.transparency = mesh.material.transparency;
        d.material.texture = -1;
        if (mesh.material.kdTexture) {
            int idx = -1;
            for (size_t k = 0; k < im.textures.size(); k++)
                if (im.textures[k] == mesh.material.kdTexture)
                    idx = (int)k;
            if (idx < 0) {
                const Image& img = *mesh.material.kdTexture;
                std::vector<float> px;
                for (const glm::vec3& c : img.pixels)
                    px.insert(px.end(), { c.x, c.y, c.z });
                texel_store.push_back(std::move(px));
                textures.push_back(rt_texture_desc { nullptr, img.width, img.height });
                im.textures.push_back(mesh.material.kdTexture);
                idx = (int)im.textures.size() - 1;
            }
            d.material.texture = idx;
        }
        im.material_texture.push_back(d.material.texture);
    }
    for (size_t k = 0; k < textures.size(); k++)
        textures[k].pixels = texel_store[k].data();
    std::vector<rt_sphere_desc> spheres(pScene->spheres.size());
    for (size_t s = 0; s < pScene->spheres.size(); s++) {
        const Sphere& sp = pScene->spheres[s];
        rt_sphere_desc& d = spheres[s];
        memset(&d, 0, sizeof(d));
        put3(d.center, sp.center);
        d.radius = sp.radius;
        put3(d.material.kd, sp.material.kd), put3(d.material.ks, sp.material.ks);
        d.material.shininess = sp.material.shininess;
        d.material.transparency = sp.material.transparency;
        d.material.texture = -1;
        im.material_texture.push_back(-1);
    }
    im.lights = flatten_lights(*pScene);
    rt_scene_desc desc;
    desc.meshes = meshes.data(), desc.n_meshes = (uint32_t)meshes.size();
    desc.spheres = spheres.data(), desc.n_spheres = (uint32_t)spheres.size();
    desc.lights = im.lights.data(), desc.n_lights = (uint32_t)im.lights.size();
    desc.textures = textures.data(), desc.n_textures = (uint32_t)textures.size();
    int n_devices = 0;
    const int n_shards = shard_count(n_devices);
    for (int sh = 0; sh < n_shards; sh++) { // one copy of the scene + BVH per shard, on the shard's device
        check(rt_set_device(sh % n_devices), "rt_set_device");
        rt_scene* sc = nullptr;
        rt_bvh* bv = nullptr;
        check(rt_scene_create(&desc, &sc), "rt_scene_create");
        im.scenes.push_back(sc);
        check(rt_bvh_build(sc, reinterpret_cast<const rt_features*>(&features), &bv), "rt_bvh_build");
        im.bvhs.push_back(bv);
    }
    check(rt_set_device(0), "rt_set_device");
    im.scene = im.scenes[0], im.bvh = im.bvhs[0];
}

    // ≙ BoundingVolumeHierarchy::intersect (bvh.cpp:299-427) for one ray
    bool intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
    {
    rt_ray r;
    put3(r.origin, ray.origin), put3(r.direction, ray.direction);
    r.t = ray.t;
    rt_hit h;
    check(rt_intersect_batch(bvh, &r, 1, reinterpret_cast<const rt_features*>(&features), &h, nullptr),
        "rt_intersect_batch");
    ray.t = r.t;
    if (!h.hit)
        return false; // hitInfo untouched, like bvh.cpp:365
    hitInfo.normal = glm::vec3(h.normal[0], h.normal[1], h.normal[2]);
    hitInfo.material.kd = glm::vec3(h.material.kd[0], h.material.kd[1], h.material.kd[2]);
    hitInfo.material.ks = glm::vec3(h.material.ks[0], h.material.ks[1], h.material.ks[2]);
    hitInfo.material.shininess = h.material.shininess;
    hitInfo.material.transparency = h.material.transparency;
    hitInfo.material.kdTexture = h.material.texture >= 0 ? textures[(size_t)h.material.texture] : nullptr;
    return true;
}

    // ≙ renderRayTracing (render.cpp:273-329): the whole frame into screen.pixels()
    void render(const rt_camera& cam, Screen& screen, const Features& features)
    {
        sync_lights();
        const glm::ivec2 res = screen.resolution();
        const rt_render_params p = params_for(5); // the literal rayDepth of render.cpp:298,308,318
        if (bvhs.size() > 1) // every shard renders its strips, the first one's device assembles the frame
            check(rt_render_sharded(bvhs.data(), (int32_t)bvhs.size(), &cam, reinterpret_cast<const rt_features*>(&features), &p,
                      res.x, res.y, 8, reinterpret_cast<float*>(screen.pixels().data()), 0),
                "rt_render_sharded");
        else
            check(rt_render(bvh, &cam, reinterpret_cast<const rt_features*>(&features), &p, res.x, res.y, 0, res.y,
                      reinterpret_cast<float*>(screen.pixels().data()), nullptr),
                "rt_render");
    }

    // ≙ getFinalColor (render.cpp:152-155) for a batch of rays
    void final_colors(const std::vector<Ray>& rays, const Features& features, int rayDepth, std::vector<glm::vec3>& out)
    {
        sync_lights();
        out.resize(rays.size());
        const rt_render_params p = params_for(rayDepth);
        check(rt_trace_batch(bvh, reinterpret_cast<const rt_ray*>(rays.data()), (uint32_t)rays.size(), nullptr,
                  reinterpret_cast<const rt_features*>(&features), &p, reinterpret_cast<float*>(out.data()), nullptr),
            "rt_trace_batch");
    }
};

// Trackball (framework/src/trackball.cpp:23-31,71-74,101-110) → the camera record of the C ABI
inline rt_camera camera_of(float fovy, float aspect, const glm::vec3& lookAt, float distance, const glm::vec3& rotation)
{
    rt_camera cam;
    const float look[3] = { lookAt.x, lookAt.y, lookAt.z };
    const float rot[3] = { rotation.x, rotation.y, rotation.z };
    check(rt_camera_from_trackball(fovy, aspect, look, distance, rot, &cam), "rt_camera_from_trackball");
    return cam;
}

} // namespace rt_fwd
