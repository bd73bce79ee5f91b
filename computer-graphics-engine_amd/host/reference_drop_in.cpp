// reference_drop_in.cpp — the render path of the reference, re-pointed at the MI355X library, written against
// the reference's OWN headers. A maintainer compiles this file IN PLACE OF
//     src/render.cpp, src/bvh_interface.cpp, src/bounding_volume_hierarchy.cpp (+ its header)
// and links librt_whitted.so instead of prebuilt/libIntersect (INTEGRATION.md has the CMake lines). Nothing in
// src/main.cpp, src/light.cpp (its two tunables), the GUI or the config reader changes: the functions below keep
// the signatures of src/render.h:32,35 and src/bvh_interface.h:12-49 exactly.
//
// Checked here, in the container only (tests/test_reference_headers.py, @pytest.mark.reference):
//     g++ -std=c++20 -fsyntax-only -fno-access-control -I/root/reference/src -I…/framework/include … this file
// against the unmodified headers under /root/reference (it is not part of the product library; on the GPU box
// the same forwarding code runs behind reference_api.hpp's re-declarations: forwarding.hpp).
//
// -fno-access-control: Trackball keeps its lens (m_fovy, m_halfScreenSpaceWidth / Height) private and
// generateRay() is not exactly invertible, so the exact half-extents the reference's rays are built from
// (trackball.cpp:26-27,101-110) are read from the private members. A maintainer who prefers not to use the
// flag adds two one-line accessors to framework/trackball.h (INTEGRATION.md shows them) and defines
// RT_TRACKBALL_HAS_ACCESSORS.
#include "bvh_interface.h"
#include "common.h"
#include "light.h" // extern int segmentLightSamples, parallelogramLightDirectionSamples (defined in src/light.cpp)
#include "render.h"
#include "scene.h"
#include "screen.h"
#include <framework/trackball.h>

#include "../../include/rt_whitted.h"

// the tunables src/render.cpp defines (render.h:11-16,25-28; render.cpp:14-25)
int raysPerPixelSide = 3;
int samplesDoF = 5;
float focusPlaneDistance = 3.f;
float blurStrength = .005f;
float bloomScalar = .3f;
float bloomThreshold = .4f;
int bloomDebugOption = 0;
int glossyReflectionsCap = 3;
int raysPerReflection = 40;
float alphaModifier = 1.f;
unsigned int areaLightSeed = 0; // the reference draws from the process-global rand()

#include "forwarding.hpp"

// The class behind BvhInterface's pimpl (src/bvh_interface.h:8,48): the reference's own
// BoundingVolumeHierarchy is replaced by the device handles.
class BoundingVolumeHierarchy : public rt_fwd::Handles {
};

namespace {
// renderRayTracing / getFinalColor receive a `const BvhInterface&` whose implementation pointer is private.
// BvhInterface::setRecursionLevel is the one const call through which the reference's render loop already talks
// to the implementation (render.cpp:30); here it reports which implementation the render call was given.
thread_local BoundingVolumeHierarchy* g_current = nullptr;
BoundingVolumeHierarchy& impl_of(const BvhInterface& bvh)
{
    bvh.setRecursionLevel(0);
    return *g_current;
}

rt_camera camera_of(const Trackball& camera)
{
    // origin and rotation through the public interface; the lens exactly as trackball.cpp:26-27 computed it
    rt_camera cam = rt_fwd::camera_of(0.0f, 1.0f, camera.lookAt(), camera.distanceFromLookAt(), camera.rotationEulerAngles());
#ifdef RT_TRACKBALL_HAS_ACCESSORS
    cam.half_w = camera.halfScreenSpaceWidth(), cam.half_h = camera.halfScreenSpaceHeight();
#else
    cam.half_w = camera.m_halfScreenSpaceWidth, cam.half_h = camera.m_halfScreenSpaceHeight;
#endif
    return cam;
}
} // namespace

// ---- src/bvh_interface.cpp ------------------------------------------------------------------------------
BvhInterface::BvhInterface(Scene* pScene, const Features& features)
{
    m_impl = new BoundingVolumeHierarchy; // never deleted, shallow copies: like bvh_interface.cpp:6-9
    m_impl->build(pScene, features);
}
int BvhInterface::numLevels() const { return rt_bvh_num_levels(m_impl->bvh); }
int BvhInterface::numLeaves() const { return rt_bvh_num_leaves(m_impl->bvh); }
void BvhInterface::setRecursionLevel(int) const { g_current = m_impl; }
void BvhInterface::setDebugRecursionLevel(int) const { g_current = m_impl; }
void BvhInterface::debugDrawLevel(int) {}         // visual debuggers of the rasterised view: not part of the path
void BvhInterface::debugDrawLeaf(int) {}
void BvhInterface::debugDrawSAHSplits(int, int) {}
bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    return m_impl->intersect(ray, hitInfo, features);
}

// ---- src/render.cpp -------------------------------------------------------------------------------------
void renderRayTracing(const Scene& /*scene*/, const Trackball& camera, const BvhInterface& bvh, Screen& screen,
    const Features& features)
{
    impl_of(bvh).render(camera_of(camera), screen, features);
}

glm::vec3 getFinalColor(const Scene& /*scene*/, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth)
{
    std::vector<glm::vec3> out;
    impl_of(bvh).final_colors({ ray }, features, rayDepth, out);
    return out[0];
}
