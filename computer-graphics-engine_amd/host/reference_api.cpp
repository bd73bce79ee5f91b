// reference_api.cpp — reference_api.hpp's call surface on top of forwarding.hpp (the forwarding code a maintainer
// drops into the reference: see reference_drop_in.cpp for the same code behind the reference's real headers).
#include "reference_api.hpp"

#include <cmath>

int segmentLightSamples = 25;              // src/light.cpp:12
int parallelogramLightDirectionSamples = 5; // src/light.cpp:13
unsigned int areaLightSeed = 0;
int raysPerPixelSide = 3;         // src/render.cpp:14
int samplesDoF = 5;               // :15
float focusPlaneDistance = 3.f;   // :16
float blurStrength = .005f;       // :17
float bloomScalar = .3f;          // :19
float bloomThreshold = .4f;       // :20
int bloomDebugOption = 0;         // :21
int glossyReflectionsCap = 3;     // :23
int raysPerReflection = 40;       // :24
float alphaModifier = 1.f;        // :25

#include "forwarding.hpp"

struct BvhInterface::Impl : rt_fwd::Handles {
};

BvhInterface::BvhInterface(Scene* pScene, const Features& features)
    : m_impl(std::make_shared<Impl>())
{
    m_impl->build(pScene, features);
}

int BvhInterface::numLevels() const { return rt_bvh_num_levels(m_impl->bvh); }
int BvhInterface::numLeaves() const { return rt_bvh_num_leaves(m_impl->bvh); }

bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    return m_impl->intersect(ray, hitInfo, features);
}

rt_camera Trackball::cameraArgs() const
{
    return rt_fwd::camera_of(m_fovy, m_aspect, m_lookAt, m_distanceFromLookAt, m_rotationEulerAngles);
}

glm::vec3 Trackball::position() const
{
    const rt_camera cam = cameraArgs();
    return glm::vec3(cam.origin[0], cam.origin[1], cam.origin[2]);
}

void renderRayTracing(const Scene& /*scene*/, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    bvh.impl()->render(camera.cameraArgs(), screen, features);
}

void getFinalColorBatch(const Scene& /*scene*/, const BvhInterface& bvh, const std::vector<Ray>& rays, const Features& features,
    int rayDepth, std::vector<glm::vec3>& out)
{
    bvh.impl()->final_colors(rays, features, rayDepth, out);
}

glm::vec3 getFinalColor(const Scene& scene, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth)
{
    std::vector<glm::vec3> out;
    getFinalColorBatch(scene, bvh, { ray }, features, rayDepth, out);
    return out[0];
}
