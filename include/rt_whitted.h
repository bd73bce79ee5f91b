/*
 * rt_whitted.h — C ABI of the MI355X-native Whitted render path.
 *
 * This is the drop-in boundary: plain C, plain pointers and
 * sizes, no C++/torch types. Every entry point names the reference interface
 * it replaces (paths relative to the reference tree). The reference-side
 * binding a maintainer would add is shown in INTEGRATION.md; a C++ shim with
 * the reference's own names (renderRayTracing / getFinalColor / BvhInterface)
 * lives in computer-graphics-engine_amd/host/reference_api.hpp.
 *
 * Conventions
 *   - every function returns an int status (RT_OK == 0); no exception crosses
 *     the boundary; rt_last_error() gives a thread-local message.
 *   - "prim id" = index of the primitive in the PRE-permutation list of
 *     src/bounding_volume_hierarchy.cpp:158-172: triangles mesh-major in
 *     scene.meshes order, then mesh.triangles order, followed by spheres.
 *     Miss = -1.
 *   - framebuffers are W*H*3 fp32, row-flipped like Screen::setPixel
 *     (src/screen.cpp:41-47): pixel (x,y) lives at ((H-1-y)*W + x)*3.
 *   - pointers named *_dev are device (HBM) pointers, everything else is host
 *     memory. Scene data is copied at rt_scene_create (the reference borrows a
 *     Scene*, src/bounding_volume_hierarchy.h:86).
 */
#ifndef RT_WHITTED_H
#define RT_WHITTED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument / descriptor */
    RT_ERR_UNSUPPORTED = 2,  /* a Features flag outside the hot-path scope */
    RT_ERR_DEVICE = 3,       /* HIP runtime error (message in rt_last_error) */
    RT_ERR_NOMEM = 4
};

/* ---- Features: byte-for-byte the layout of `Features` + `ExtraFeatures`
 * (src/common.h:54-77, 17 one-byte bools) so a `const Features&` can be
 * passed as `(const rt_features*)&features`. ------------------------------ */
typedef struct rt_features {
    uint8_t enableShading;
    uint8_t enableRecursive;
    uint8_t enableHardShadow;
    uint8_t enableSoftShadow;
    uint8_t enableNormalInterp;
    uint8_t enableTextureMapping;
    uint8_t enableAccelStructure;
    /* ExtraFeatures (src/common.h:54-65) */
    uint8_t enableEnvironmentMapping;
    uint8_t enableBvhSahBinning;
    uint8_t enableMotionBlur;
    uint8_t enableBloomEffect;
    uint8_t enableBilinearTextureFiltering;
    uint8_t enableMipmapTextureFiltering;
    uint8_t enableMultipleRaysPerPixel;
    uint8_t enableGlossyReflection;
    uint8_t enableTransparency;
    uint8_t enableDepthOfField;
} rt_features;

/* ---- Ray: layout of framework/include/framework/ray.h:9-13 (28 bytes) ---- */
typedef struct rt_ray {
    float origin[3];
    float direction[3];
    float t;
} rt_ray;

/* ---- Material without the shared_ptr<Image> (framework mesh.h:22-34) ----- */
typedef struct rt_material {
    float kd[3];
    float ks[3];
    float shininess;
    float transparency;
    int32_t texture; /* index into rt_scene_desc.textures, -1 = none (kdTexture) */
} rt_material;

/* ---- Mesh (framework mesh.h:36-43): SoA view of Vertex{position,normal,
 * texCoord} + uvec3 triangles + Material ---------------------------------- */
typedef struct rt_mesh_desc {
    const float* positions;    /* n_vertices * 3 */
    const float* normals;      /* n_vertices * 3 */
    const float* texcoords;    /* n_vertices * 2 */
    const uint32_t* triangles; /* n_triangles * 3 vertex indices */
    uint32_t n_vertices;
    uint32_t n_triangles;
    rt_material material;
} rt_mesh_desc;

/* ---- Sphere (src/common.h:31-35) ---------------------------------------- */
typedef struct rt_sphere_desc {
    float center[3];
    float radius;
    rt_material material; /* texture ignored (spheres are never textured) */
} rt_sphere_desc;

/* ---- Lights: tagged union of PointLight / SegmentLight / ParallelogramLight
 * (src/common.h:37-52). v[] holds the struct's vec3 members in declaration
 * order: point {position,color}; segment {endpoint0,endpoint1,color0,color1};
 * parallelogram {v0,edge01,edge02,color0,color1,color2,color3}. ------------ */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_SEGMENT = 1, RT_LIGHT_PARALLELOGRAM = 2 };
typedef struct rt_light_desc {
    int32_t kind;
    float v[7][3];
} rt_light_desc;

/* ---- Image (framework image.h:11-18): fp32 RGB, row-major ---------------- */
typedef struct rt_texture_desc {
    const float* pixels; /* width*height*3 */
    int32_t width;
    int32_t height;
} rt_texture_desc;

/* ---- Scene (src/scene.h:28-33) ------------------------------------------ */
typedef struct rt_scene_desc {
    const rt_mesh_desc* meshes;
    uint32_t n_meshes;
    const rt_sphere_desc* spheres;
    uint32_t n_spheres;
    const rt_light_desc* lights;
    uint32_t n_lights;
    const rt_texture_desc* textures;
    uint32_t n_textures;
} rt_scene_desc;

/* ---- Camera: what Trackball::generateRay / position() need
 * (framework/src/trackball.cpp:101-110, 71-74, 26-27) ---------------------- */
typedef struct rt_camera {
    float origin[3]; /* Trackball::position() */
    float quat[4];   /* glm::quat(eulerAngles) as (w,x,y,z) */
    float half_w;    /* m_halfScreenSpaceWidth  = aspect * tan(fovy/2) */
    float half_h;    /* m_halfScreenSpaceHeight = tan(fovy/2) */
    float look_at[3]; /* Trackball::lookAt() — depth of field only (getPlane, src/render.cpp:225-230) */
} rt_camera;

/* ---- per-ray hit record, ≙ the observable result of
 * BvhInterface::intersect (src/bvh_interface.h:45): return value, ray.t,
 * HitInfo.normal, HitInfo.material (kd after texture lookup) ---------------- */
typedef struct rt_hit {
    int32_t hit;   /* 0 / 1 */
    int32_t prim;  /* prim id (see above), -1 on miss */
    float t;       /* ray.t after the call */
    float normal[3];
    rt_material material; /* .texture = source mesh's texture index */
} rt_hit;

/* ---- explicit render parameters that are globals / literals in the
 * reference: recursion depth (literal 5 at src/render.cpp:318),
 * parallelogramLightDirectionSamples / segmentLightSamples
 * (src/light.cpp:12-13) and the area-light RNG seed (reference: rand()) ---- */
typedef struct rt_render_params {
    int32_t depth;            /* rayDepth passed to getFinalColor */
    int32_t parallelogram_n;  /* samples per side (reference default 5) */
    int32_t segment_n;        /* samples (reference default 25) */
    uint32_t seed;            /* counter-based RNG seed */
    /* the extras' globals (src/render.cpp:14-22); read only when their feature flag is set */
    int32_t rays_per_pixel_side;  /* raysPerPixelSide (3): enableMultipleRaysPerPixel traces n x n jittered rays */
    int32_t samples_dof;          /* samplesDoF (5): enableDepthOfField */
    float focus_plane_distance;   /* focusPlaneDistance (3.f) */
    float blur_strength;          /* blurStrength (.005f): sampling radius on the lens plane */
    float bloom_scalar;           /* bloomScalar (.3f): enableBloomEffect */
    float bloom_threshold;        /* bloomThreshold (.4f) */
    int32_t bloom_debug_option;   /* bloomDebugOption (0 = final image, 1 = bloom only, else unfiltered) */
    int32_t glossy_reflections_cap; /* glossyReflectionsCap (3): enableGlossyReflection */
    int32_t rays_per_reflection;  /* raysPerReflection (40) */
    float alpha_modifier;         /* alphaModifier (1.f) */
} rt_render_params;

/* ---- ray / test counters of one call (what SURVEY.md §8(d) prices) ------- */
typedef struct rt_counters {
    uint64_t rays;        /* BvhInterface::intersect call equivalents traced */
    uint64_t rays_primary;
    uint64_t rays_shadow;
    uint64_t rays_reflect;   /* distinct reflection rays traversed */
    uint64_t rays_ref_equiv; /* what the reference would count (reflections x2^k) */
    uint64_t box_tests;      /* ray-AABB slab tests (lane-level) */
    uint64_t tri_tests;      /* ray-triangle tests (lane-level) */
    uint64_t sphere_tests;
    uint64_t node_fetches;   /* wave-level internal-node fetches (64 B each) */
    uint64_t tri_fetches;    /* wave-level triangle fetches */
    uint64_t hits_shaded;    /* closest hits whose attributes were resolved and shaded */
    uint64_t rays_queued;    /* reflection rays written to a device queue */
    uint64_t group_tests;    /* conservative ray / triangle-group box tests of the leaf phase (pruned policy only) */
    uint64_t rays_culled;    /* primary rays (counted in `rays`) of tiles the timed launch ends in its prologue or never
                              * launches: their ray pyramid misses the scene bounds, so they never touch the BVH */
    uint64_t rays_shadow_skipped; /* shadow-ray call equivalents NOT traced (and not counted in `rays` / `rays_shadow`):
                              * the light's colour at the hit is exactly zero (surface facing away: every component
                              * +-0 or NaN), so colour x visibility has the same bits whatever the visibility */
    uint64_t rays_exact;     /* rays traced by the exact-topology engine: all of them where the wide tree does not
                              * apply (spheres, unpruned policy, brute force, tiny scenes), else the rays that are not
                              * all_ok or that the wide engine handed back (csrc/rt_wide.h) */
} rt_counters;

typedef struct rt_scene rt_scene; /* opaque */
typedef struct rt_bvh rt_bvh;     /* opaque */

/* thread-local description of the last failure on the calling thread */
const char* rt_last_error(void);

/* library / device info: returns RT_ERR_DEVICE when no gfx950 device or the
 * HIP runtime is unusable. The product path never falls back to the CPU. */
int rt_device_info(int* device_count, char* name_buf, size_t name_buf_len);

/* Process-wide options. "unpruned" = 1: traverse with the reference's own policy
 * (every leaf whose boxes are hit is tested, src/bounding_volume_hierarchy.cpp:334-355);
 * 0 (default): additionally skip subtrees and leaf triangle groups that provably
 * cannot hold the closest candidate (csrc/rt_device.h "Pruning", csrc/rt_layout.h
 * GroupRecord) — results are identical bit for bit.
 * Scheduling / tuning options, none of which changes any result: "fused" (1: all
 * recursion levels in one launch), "fused_from", "tile_wlog" / "tile_hlog" (primary
 * tile = 2^w x 2^h pixels per wave, w + h <= 6; setting either switches "tile_auto" off), "tile_auto" (1, the default:
 * one-frame launches pick the tile by what the frame is — a row of 32 or 64 pixels for point lights on triangles, 4 x 4
 * for area lights), "tile_order" (1: centre-first dispatch),
 * "leaf_flush", "min_waves", "super_log", "tile_cull", "launch_rect", "merge_next", "wide" (0: the
 * exact-topology engine alone), "xcd_order" / "xcd_block" (screen blocks of 2^v pixels a side per XCD),
 * "frame_major", "order_one_launch" (the tile order built in one launch: 1 for lone frames, 2 always, 0 never), "skip_unlit"
 * (0: trace the shadow rays of unlit hits too), "prefetch", "host_bands" (rt_render into a host buffer: the frame
 * goes out as that many bands of rows whose copies back overlap the later bands' rendering; 1: one launch and one
 * copy; 0, the default: 4 for frames as cheap to render as to copy, else 1). Unknown names return RT_ERR_INVALID. */
int rt_set_option(const char* name, int value);

/* Select the HIP device used by subsequent creates on this thread. */
int rt_set_device(int device);

/* -- Scene ≙ `Scene` value + loadScene* results (src/scene.h:28-33). Copies
 *    everything (host flatten + device upload). ----------------------------- */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);
/* The GUI moves lights between frames (src/main.cpp:296-368). */
int rt_scene_update_lights(rt_scene* scene, const rt_light_desc* lights, uint32_t n_lights);

/* -- BVH ≙ BvhInterface::BvhInterface(Scene*, const Features&)
 *    (src/bvh_interface.cpp:6-9 → src/bounding_volume_hierarchy.cpp:149-194).
 *    Host C++ build (same split sequence), SoA flatten, device upload. ------ */
int rt_bvh_build(rt_scene* scene, const rt_features* features, rt_bvh** out);
int rt_bvh_destroy(rt_bvh* bvh);
/* ≙ BvhInterface::numLevels()/numLeaves() (src/bvh_interface.h:20,24) */
int rt_bvh_num_levels(const rt_bvh* bvh);
int rt_bvh_num_leaves(const rt_bvh* bvh);
int rt_bvh_num_nodes(const rt_bvh* bvh);
int rt_bvh_num_prims(const rt_bvh* bvh);
/* Host-only structural check of the traversal tree the kernels walk (csrc/rt_wide.h: our own 4-wide SAH tree over
 * the triangles; the reference's tree above is what decides results): every triangle in exactly one leaf slot, every
 * child box enclosing the triangles below it. stats: [0] nodes [1] leaf groups [2] depth [3] triangles [4] empty
 * child slots [5] internal children [6] 1 if the scene has a wide tree (0: spheres, tiny scene: exact engine only). */
int rt_bvh_wide_check(const rt_bvh* bvh, uint64_t stats[8]);

/* Debug/parity export of the reference-shaped tree: post-order nodes
 * (root = last). node_aabb: n*6 floats (lower,upper); node_meta: n*6 int32
 * {leaf, depth, beg, end, left, right} (left/right = -1 for leaves);
 * prim_order: n_prims int32 = prim id stored at each permuted position.
 * Any pointer may be NULL. */
int rt_bvh_export(const rt_bvh* bvh, float* node_aabb, int32_t* node_meta, int32_t* prim_order);

/* -- Camera helper ≙ Trackball ctor + setCamera + position()
 *    (framework/src/trackball.cpp:23-31,64-74). Angles in radians. ---------- */
int rt_camera_from_trackball(float fovy, float aspect, const float look_at[3], float distance,
                             const float rotation_euler[3], rt_camera* out);

/* -- Bloom ≙ renderBloomFilter (src/render.cpp:157-196) on a complete W x H frame in device
 *    memory (row-flipped, as rt_render_device writes it). rt_render applies it itself when
 *    features.extra.enableBloomEffect is set; callers that assemble a frame from strips
 *    (rt_render_device with a row list, multi-GPU) call it after the frame is whole.
 *    The random samples of enableMultipleRaysPerPixel / enableDepthOfField (std::mt19937 seeded
 *    from std::random_device in the reference, src/render.cpp:213-216,246-249) come from the
 *    counter-based generator rt_rand31 keyed by (seed, pixel, sample). ---------------------- */
int rt_bloom_device(float* rgb_dev, int32_t width, int32_t height, const rt_render_params* params, void* hip_stream);

/* -- Frame ≙ renderRayTracing (src/render.cpp:273-329, plain path :316-323)
 *    with explicit depth. Renders image rows [row_begin,row_end) of a W x H
 *    frame (row = the reference's loop variable y, bottom-up) and writes them
 *    at their row-flipped position of the FULL frame buffer `rgb` (host). ---- */
int rt_render(const rt_bvh* bvh, const rt_camera* camera, const rt_features* features,
              const rt_render_params* params, int32_t width, int32_t height,
              int32_t row_begin, int32_t row_end, float* rgb, rt_counters* counters);

/* Same, framebuffer resident in HBM; enqueued on `hip_stream` (a hipStream_t,
 * NULL = default stream) without host synchronisation. `rows` lists the image
 * rows this call renders (host array, n_rows entries; NULL = all rows
 * [0,height)); row i of the list is written to packed row i of `rgb_dev` when
 * `packed` != 0, otherwise to its row-flipped place in a full W*H frame.
 * counters_dev (device, may be NULL) receives an rt_counters. */
int rt_render_device(const rt_bvh* bvh, const rt_camera* camera, const rt_features* features,
                     const rt_render_params* params, int32_t width, int32_t height,
                     const int32_t* rows, int32_t n_rows, int32_t packed, float* rgb_dev,
                     rt_counters* counters_dev, void* hip_stream);

/* The part of a frame that can differ from the background: rect = {x0, y0, x1, y1} (columns [x0, x1), image rows
 * — the reference's loop variable y, src/render.cpp:283 — [y0, y1)) such that every pixel outside it is vec3(0)
 * in the frames rt_render / rt_render_device compute for these cameras (the union over them; before the bloom filter): the rays of those
 * pixels cannot reach the scene's bounds, and the render calls do not even launch their tiles. Conservative, aligned
 * outward to 8 pixels; the whole frame when no such statement can be made (depth of field, spheres, a camera inside
 * the scene's bounds, acceleration structure off). Host arithmetic only. What a multi-GPU host uses to gather only
 * the part of its strips that carries information (sharding.StripRenderer). */
int rt_frame_rect(const rt_bvh* bvh, const rt_camera* cameras, int32_t n_cameras, const rt_features* features,
                  const rt_render_params* params, int32_t width, int32_t height, int32_t rect[4]);

/* -- Several frames (cameras) in ONE launch, framebuffers in HBM: frame k from cameras[k] goes to
 *    rgb_dev + k * frame_stride_floats (0 = tight: rows_per_frame * W * 3 packed, else H * W * 3 row-flipped);
 *    rows / n_rows / packed as in rt_render_device, the same for every frame. What the reference does
 *    with one std::thread per camera (src/main.cpp:512-531) and what a rank of a multi-GPU job does with its
 *    small share of successive frames: a launch of several frames has no idle tail between them. Up to 8
 *    frames share a launch when the pixel loop is the plain one (one ray per pixel, no bloom, no glossy /
 *    transparency recursion), the cameras share one lens and a frame's rows are whole tiles; otherwise the
 *    frames are rendered one after the other — same results either way. ------------------------------ */
int rt_render_device_frames(const rt_bvh* bvh, const rt_camera* cameras, int32_t n_frames,
                            const rt_features* features, const rt_render_params* params, int32_t width,
                            int32_t height, const int32_t* rows, int32_t n_rows, int32_t packed, float* rgb_dev,
                            uint64_t frame_stride_floats, void* hip_stream);

/* -- One frame on SEVERAL devices from ONE process (the C++ host's multi-GPU path; replaces the reference's
 *    in-process OpenMP row loop, src/render.cpp:277-281): bvhs[s] is a BVH handle built on the device shard s
 *    renders on (rt_set_device(d) before rt_scene_create / rt_bvh_build; the scene and BVH are replicated, one
 *    copy per device; two shards may share a device). The frame's strips of `strip_rows` image rows are dealt
 *    round-robin to the shards (strip k -> shard k % n_shards: balances an object in the middle of the frame),
 *    every shard renders its strips into a packed buffer on its own device and stream, the packed strips are
 *    gathered on shard 0's device by peer copies (hipMemcpyPeerAsync — xGMI between the GPUs of a node) and
 *    de-interleaved there into the reference's row-flipped frame. The area-light RNG is keyed by the global pixel
 *    id: the image does not depend on n_shards. renderBloomFilter (whole-frame) runs on the assembled frame.
 *    rgb: width * height * 3 floats — host memory when rgb_on_device == 0, else memory of shard 0's device.
 *    Enqueues all shards before it waits for any; returns when the frame is complete. ------------------------ */
int rt_render_sharded(rt_bvh* const* bvhs, int32_t n_shards, const rt_camera* camera, const rt_features* features,
                      const rt_render_params* params, int32_t width, int32_t height, int32_t strip_rows, float* rgb,
                      int32_t rgb_on_device);

/* -- The same into host memory: n_frames row-flipped frames of width * height * 3 floats, one after the other
 *    in rgb (what n calls of rt_render would write, one camera each). ----------------------------------- */
int rt_render_frames(const rt_bvh* bvh, const rt_camera* cameras, int32_t n_frames, const rt_features* features,
                     const rt_render_params* params, int32_t width, int32_t height, float* rgb);

/* Primary-ray hit records of a frame (parity instrument: bit patterns of t and
 * prim ids at full resolution). t_bits/prim: W*H entries, row-flipped like the
 * framebuffer. Host pointers. */
int rt_render_hits(const rt_bvh* bvh, const rt_camera* camera, const rt_features* features,
                   int32_t width, int32_t height, uint32_t* t_bits, int32_t* prim);

/* -- Batch of rays ≙ getFinalColor (src/render.cpp:152-155): rgb[i] =
 *    getFinalColor(scene, bvh, rays[i], features, params->depth). pixel_ids
 *    (may be NULL → i) key the area-light RNG. Host pointers. --------------- */
int rt_trace_batch(const rt_bvh* bvh, const rt_ray* rays, uint32_t n_rays, const uint32_t* pixel_ids,
                   const rt_features* features, const rt_render_params* params, float* rgb,
                   rt_counters* counters);

/* -- Batch of rays ≙ BvhInterface::intersect (src/bvh_interface.cpp:62 →
 *    src/bounding_volume_hierarchy.cpp:299-427): rays[i].t is updated in
 *    place, hits[i] filled. Host pointers. ---------------------------------- */
int rt_intersect_batch(const rt_bvh* bvh, rt_ray* rays, uint32_t n_rays, const rt_features* features,
                       rt_hit* hits, rt_counters* counters);

/* -- Scene ingestion ≙ loadMesh / centerAndScaleToUnitMesh
 *    (framework/src/mesh.cpp:52-176, tinyobjloader semantics), Image::Image
 *    (framework/src/image.cpp:13-35, PNG → fp32 RGB) and the preset scenes /
 *    loadSceneFromFile (src/scene.cpp:5-103). Host only (no GPU needed). The loaded
 *    scene owns its arrays; rt_loaded_scene_desc fills a descriptor that points
 *    into it (valid until rt_loaded_scene_destroy) and can go to rt_scene_create. */
typedef struct rt_loaded_scene rt_loaded_scene;
int rt_load_obj(const char* path, int center_and_normalize, const rt_light_desc* lights, uint32_t n_lights,
                rt_loaded_scene** out);
int rt_load_preset(int scene_type /* SceneType, src/scene.h:14-25 */, const char* data_dir, rt_loaded_scene** out);
int rt_loaded_scene_desc(rt_loaded_scene* scene, rt_scene_desc* out);
int rt_loaded_scene_destroy(rt_loaded_scene* scene);
const char* rt_io_last_error(void);
/* ≙ Screen::writeBitmapToFile (src/screen.cpp:49-60, row f2): the row-flipped fp32 frame
 * of rt_render → stb_image_write's 32-bit V4 BMP (clamp, *255, truncate; NaN → 0). */
int rt_write_bmp(const char* path, const float* rgb, int32_t width, int32_t height);

/* -- Instrumentation for the roofline figure (bench.py): when enabled on a bvh,
 *    every k_trace_level launch of later rt_render_device calls is bracketed by
 *    hipEvents recorded on the launch stream. rt_profile_read synchronises the
 *    stream and returns the summed kernel milliseconds and the launch count since
 *    the last reset. Off by default; costs two event records per launch. ------- */
int rt_profile_enable(rt_bvh* bvh, int enable);
int rt_profile_read(rt_bvh* bvh, void* hip_stream, double* trace_kernel_ms, uint64_t* trace_kernel_launches,
                    int reset);
/* The same events one by one: launch_ms[i] = duration of the i-th trace-kernel launch on `hip_stream` since the
 * last reset (the first `cap` of them), *n_launches = how many there were. */
int rt_profile_read_launches(rt_bvh* bvh, void* hip_stream, double* launch_ms, uint64_t cap, uint64_t* n_launches,
                             int reset);

/* -- Device self-check of the slab test's reciprocal-based exact division
 *    (csrc/rt_device.h div_by_ray) against the device's own IEEE '/', on
 *    n_samples pseudo-random operand pairs. mode: 0 raw bit patterns, 1 moderate
 *    exponents, 2 divisors with an all-ones significand, 3 slab-shaped operands.
 *    Returns the number of differing quotients (must be 0) and how many samples
 *    took the fast path. ------------------------------------------------------- */
int rt_debug_div_check(uint32_t seed, uint32_t mode, uint64_t n_samples, uint64_t* mismatches,
                       uint64_t* fast_path_samples);

/* -- Device self-check of the tile dispatch order (a scheduling choice, but one that must start every tile of
 *    the launched grid exactly once): builds the order of a launch of n_frames frames of n_rows / n_frames rows
 *    and `width` columns, tiles of 2^tile_wlog x 2^tile_hlog pixels, launched sub-grid rect = {tile_x0, tile_y0,
 *    tiles_w, tiles_h} (NULL: the whole grid), exactly as the render calls do under the current options, and
 *    copies it to order_out (host, `cap` entries). info: [0] grid width, [1] grid height in tiles, [2] XCD order
 *    in use, [3] launch positions, [4..11] first place of XCD list x, [12..19] its length. ------------------- */
int rt_debug_tile_order(int32_t width, int32_t n_rows, int32_t n_frames, int32_t tile_wlog, int32_t tile_hlog,
                        const int32_t rect[4], uint32_t* order_out, uint64_t cap, uint32_t info[20]);

/* -- Multi-GPU strip de-interleave: gathered packed strips
 *    of all ranks → the reference's row-flipped frame. gathered_dev holds
 *    n_ranks blocks of rows_per_rank_max*W*3 floats; strip s (strip_rows image
 *    rows) belongs to rank s % n_ranks. Device pointers, enqueued on stream. - */
int rt_deinterleave_strips(const float* gathered_dev, float* frame_dev, int32_t width, int32_t height,
                           int32_t strip_rows, int32_t n_ranks, void* hip_stream);

/* -- The same for a BATCH of frames exchanged in one all-gather (a rank's share of a frame is small at
 *    8 ranks; several frames per exchange keep the exchange off the critical path): gathered_dev holds
 *    n_ranks blocks of batch_frames frames of rows_per_rank_max*W*3 floats; the first n_frames of them
 *    are written to frames_dev[n_frames][H][W][3], each row-flipped as above. One launch. ------------- */
int rt_deinterleave_frames(const float* gathered_dev, float* frames_dev, int32_t width, int32_t height,
                           int32_t strip_rows, int32_t n_ranks, int32_t n_frames, int32_t batch_frames,
                           void* hip_stream);

/* The area-light RNG (replaces rand(), src/light.cpp:21,32-33): 31-bit value
 * for (seed, pixel id, recursion level, light index, sample index, dim). */
uint32_t rt_rand31(uint32_t seed, uint32_t pixel, uint32_t level, uint32_t light, uint32_t sample,
                   uint32_t dim);

#ifdef __cplusplus
}
#endif
#endif /* RT_WHITTED_H */
