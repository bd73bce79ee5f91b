// reference_api.hpp — host-side C++ mirror of the reference's call surface for the
// render path, forwarding to the C ABI (include/rt_whitted.h).
//
// Same names, argument meaning and error behaviour as the reference:
//   renderRayTracing(scene, camera, bvh, screen, features)      src/render.h:32  (render.cpp:273-329)
//   getFinalColor(scene, bvh, ray, features, rayDepth = 0)      src/render.h:35  (render.cpp:152-155)
//   BvhInterface(Scene*, const Features&), intersect, numLevels, numLeaves
//                                                               src/bvh_interface.h:12-49
//   Scene / Mesh / Vertex / Material / Sphere / lights / Features / HitInfo / Ray /
//   Image / Screen / Trackball (the members the path reads)     src/common.h, src/scene.h,
//                                                               framework/{ray,mesh,image,trackball}.h, src/screen.h
//   extern int parallelogramLightDirectionSamples, segmentLightSamples   src/light.h:9-10
//
// A maintainer of the reference keeps the reference's own headers and replaces
// the bodies of src/render.cpp / src/bvh_interface.cpp with the forwarding code of
// reference_api.cpp; this header exists so that the same code
// compiles and is tested here, where the reference tree and glm are absent:
// define RT_SHIM_HAVE_GLM to use the real glm types instead of the layout-
// compatible stand-ins below.
#pragma once
#include <cstdint>
#include <filesystem>
#include <limits>
#include <memory>
#include <variant>
#include <vector>

#include "../../include/rt_whitted.h"

#ifdef RT_SHIM_HAVE_GLM
#include <glm/vec2.hpp>
#include <glm/vec3.hpp>
#else
namespace glm {
struct vec2 {
    float x = 0, y = 0;
    vec2() = default;
    vec2(float a, float b) : x(a), y(b) {}
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct uvec3 {
    unsigned x = 0, y = 0, z = 0;
};
struct ivec2 {
    int x = 0, y = 0;
};
} // namespace glm
#endif

// framework/include/framework/ray.h:9-13
struct Ray {
    glm::vec3 origin { 0.0f };
    glm::vec3 direction { 0.0f, 0.0f, -1.0f };
    float t { std::numeric_limits<float>::max() };
};

// framework/include/framework/image.h:11-18 (pixels supplied by the caller here)
struct Image {
    int width = 0, height = 0;
    std::vector<glm::vec3> pixels;
};

// framework/include/framework/mesh.h:14-43
struct Vertex {
    glm::vec3 position;
    glm::vec3 normal;
    glm::vec2 texCoord;
};
struct Material {
    glm::vec3 kd;
    glm::vec3 ks { 0.0f };
    float shininess { 1.0f };
    float transparency { 1.0f };
    std::shared_ptr<Image> kdTexture;
};
struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<glm::uvec3> triangles;
    Material material;
};

// src/common.h:14-77
struct HitInfo {
    glm::vec3 normal;
    glm::vec3 barycentricCoord;
    glm::vec2 texCoord;
    Material material;
};
struct Sphere {
    glm::vec3 center { 0.0f };
    float radius = 1.0f;
    Material material;
};
struct PointLight {
    glm::vec3 position, color;
};
struct SegmentLight {
    glm::vec3 endpoint0, endpoint1, color0, color1;
};
struct ParallelogramLight {
    glm::vec3 v0, edge01, edge02, color0, color1, color2, color3;
};
struct ExtraFeatures {
    bool enableEnvironmentMapping = false;
    bool enableBvhSahBinning = false;
    bool enableMotionBlur = false;
    bool enableBloomEffect = false;
    bool enableBilinearTextureFiltering = false;
    bool enableMipmapTextureFiltering = false;
    bool enableMultipleRaysPerPixel = false;
    bool enableGlossyReflection = false;
    bool enableTransparency = false;
    bool enableDepthOfField = false;
};
struct Features {
    bool enableShading = false;
    bool enableRecursive = false;
    bool enableHardShadow = false;
    bool enableSoftShadow = false;
    bool enableNormalInterp = false;
    bool enableTextureMapping = false;
    bool enableAccelStructure = false;
    ExtraFeatures extra = {};
};
static_assert(sizeof(Features) == sizeof(rt_features), "Features must keep the reference's 17-byte layout");
static_assert(sizeof(Ray) == sizeof(rt_ray), "Ray must keep the reference's 28-byte layout");

// src/scene.h:28-33 (SceneType is only a tag in the reference)
struct Scene {
    int type = 0;
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>> lights;
};

// src/screen.h — the framebuffer part (setPixel/indexAt/pixels, screen.cpp:41-47,114-127)
class Screen {
public:
    explicit Screen(const glm::ivec2& resolution)
        : m_resolution(resolution)
        , m_textureData(size_t(resolution.x) * size_t(resolution.y), glm::vec3(0.0f))
    {
    }
    void setPixel(int x, int y, const glm::vec3& color) { m_textureData[size_t(indexAt(x, y))] = color; }
    [[nodiscard]] glm::ivec2 resolution() const { return m_resolution; }
    [[nodiscard]] const std::vector<glm::vec3>& pixels() const { return m_textureData; }
    [[nodiscard]] std::vector<glm::vec3>& pixels() { return m_textureData; }
    [[nodiscard]] int indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; }

private:
    glm::ivec2 m_resolution;
    std::vector<glm::vec3> m_textureData;
};

// framework/include/framework/trackball.h — the members the render path reads. The
// reference couples it to a Window only to get the aspect ratio (trackball.cpp:27).
class Trackball {
public:
    Trackball(float aspectRatio, float fovy, float distanceFromLookAt = 4.0f, float rotationX = 0.0f, float rotationY = 0.0f)
        : m_aspect(aspectRatio), m_fovy(fovy), m_distanceFromLookAt(distanceFromLookAt), m_rotationEulerAngles(rotationX, rotationY, 0.0f)
    {
    }
    void setCamera(const glm::vec3 lookAt, const glm::vec3 rotations, const float dist)
    {
        m_lookAt = lookAt, m_rotationEulerAngles = rotations, m_distanceFromLookAt = dist;
    }
    [[nodiscard]] glm::vec3 position() const; // trackball.cpp:71-74
    // what generateRay (trackball.cpp:101-110) needs, evaluated per pixel on the GPU
    [[nodiscard]] rt_camera cameraArgs() const;

private:
    float m_aspect, m_fovy;
    glm::vec3 m_lookAt { 0.0f };
    float m_distanceFromLookAt;
    glm::vec3 m_rotationEulerAngles { 0.0f };
};

// src/bvh_interface.h:12-49. Like the reference's (bvh_interface.cpp:6-9: a raw
// `new` that is never deleted, shallow copies) copies share one implementation;
// here it is reference-counted and freed with the last copy.
class BvhInterface {
public:
    BvhInterface(Scene* pScene, const Features& features);
    [[nodiscard]] int numLevels() const;
    [[nodiscard]] int numLeaves() const;
    bool intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const;

    // (shim internals)
    struct Impl;
    [[nodiscard]] const std::shared_ptr<Impl>& impl() const { return m_impl; }

private:
    std::shared_ptr<Impl> m_impl;
};

// src/light.h:9-10 — read at render time, like the reference's globals
extern int segmentLightSamples;
extern int parallelogramLightDirectionSamples;
// area-light RNG seed (the reference uses the process-global rand())
extern unsigned int areaLightSeed;
// src/render.h:11-16,25-28 — the extras' sliders, read at render time
extern float bloomScalar;
extern float bloomThreshold;
extern int bloomDebugOption;
extern int raysPerPixelSide;
extern int samplesDoF;
extern float focusPlaneDistance;
extern float blurStrength;
extern int raysPerReflection;
extern float alphaModifier;
extern int glossyReflectionsCap; // file-local in the reference (src/render.cpp:23); exposed here for tests

// src/render.h:32,35
void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features);
glm::vec3 getFinalColor(const Scene& scene, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth = 0);
// batch form of the same call (what a GPU caller should use instead of a loop)
void getFinalColorBatch(const Scene& scene, const BvhInterface& bvh, const std::vector<Ray>& rays, const Features& features,
    int rayDepth, std::vector<glm::vec3>& out);
