// rt_capi.hip — the C ABI of include/rt_whitted.h on top of the HIP kernels.
//
// No CPU fallback exists anywhere in this library: every compute entry point
// needs a HIP device and returns RT_ERR_DEVICE (with rt_last_error()) otherwise.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_whitted.h"
#include "host_scene.h"
#include "rt_kernels.h"
#include "rt_shade.h"

namespace {

thread_local std::string g_error;
// host-output entry points share the default-stream scratch of a bvh: one at a time
std::mutex g_host_call_mu;
// rt_set_option("unpruned", 1): traverse with the reference's own policy (no pruning)
std::atomic<int> g_unpruned { 0 };
// rt_set_option("leaf_flush", n): leaf-queue fill that triggers the leaf phase (tuning)
std::atomic<int> g_leaf_flush { 16 };
// rt_set_option("host_bands", n): rt_render into a HOST buffer renders the frame as n bands of rows on alternating
// streams and copies every band back while the later ones render (1: one launch, then one copy; 0 = automatic:
// 4 bands where the copy back weighs as much as the render — one ray per pixel, point lights —, 1 where the render
// dominates and bands would only add launch tails: area lights with soft shadows, several samples per pixel, glossy
// reflections, transparency)
std::atomic<int> g_host_bands { 0 };
// rt_set_option("wide", 0): traverse the reference-topology tree only (the wide tree stays built)
std::atomic<int> g_wide { 1 };
// rt_set_option("xcd_order", 1): the tile dispatch order deals screen blocks to the XCDs (rt_kernels.h CameraArgs)
// rt_set_option("prefetch", 1): the wide engine touches the next node's line at the end of a step
std::atomic<int> g_xcd_order { 1 }, g_prefetch { 0 }, g_xcd_block { 4 }, g_frame_major { 1 }, g_skip_unlit { 1 }, g_order_one_launch { 1 };
// rt_set_option("tile_wlog"/"tile_hlog"/"min_waves"): primary tile shape, sparse-level packing (tuning)
// "tile_auto" 1 (default; switched off by setting tile_wlog / tile_hlog): ONE-frame camera launches pick their tile
// shape by what the frame is — a row of 32 x 1 pixels (64 x 1 from 4 Mpixel up) for point lights on triangles: a
// row mixes cheap and expensive path trees in one wave where a square tile inside the object holds only expensive
// ones, and 32 - 64 rays per wave cost fewer instructions per ray than 16 (lone C3 frame 0.56 -> 0.49 ms over the
// orbit, C5 2.42 -> 1.47, C2 0.153 -> 0.12 against 4 x 4); 4 x 4 for area lights, whose 64 samples per hit make a
// wave of 32 or 64 hits last too long (C4 3.5 ms against 4.3 / 5.8) — profiles/r03_notes.md §7
std::atomic<int> g_tile_auto { 1 };
std::atomic<int> g_tile_wlog { 2 }, g_tile_hlog { 2 }, g_min_waves { 4096 }, g_super_log { 0 }, g_fused { 1 }, g_tile_order { 1 }, g_fused_from { 0 }, g_merge_next { 1 }, g_tile_cull { 1 }, g_launch_rect { 1 };
unsigned long long* g_stamps = nullptr; // diagnostic builds: rt_debug_stamps

int fail(int code, const std::string& msg)
{
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr, n = 0;
    }
    hipError_t reserve(size_t count)
    {
        if (count <= n)
            return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess)
            n = count;
        return e;
    }
    hipError_t upload(const std::vector<T>& v)
    {
        hipError_t e = reserve(std::max<size_t>(v.size(), 1));
        if (e != hipSuccess || v.empty())
            return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// per-(bvh, stream) scratch: level colours, per-item state, two ray queues
struct Scratch {
    DevBuf<float> direct;
    DevBuf<uint32_t> state;
    DevBuf<float> qray[2];
    DevBuf<uint32_t> qitem[2], qpixel[2];
    DevBuf<uint32_t> counts;
    DevBuf<int32_t> rows;
    std::vector<int32_t> rows_host;
    std::vector<rt::CameraPose> poses; // rt_render_device_frames: the frames' poses of the launch being enqueued
    DevBuf<uint32_t> order; // tile dispatch order (tile_order_for)
    DevBuf<uint32_t> order_rings; // its ring counters
    uint32_t xcd_count[8] = {};   // XCD-aware order: lengths of the eight lists of the cached order
    DevBuf<float> samples;   // per-sample colours of multi-sample frames (k_resolve input)
    DevBuf<float> frames;    // k_trace_general: per-wave-slot recursion frames
    DevBuf<float> bloom_src; // the frame before the bloom filter
    std::vector<int64_t> order_key;
    DevBuf<float> frame;      // device frame for host-output calls
    DevBuf<float> in_ray, in_t; // uploaded user rays
    DevBuf<uint32_t> in_pixel;
    DevBuf<rt_hit> hits;
    DevBuf<uint32_t> t_bits;
    DevBuf<int32_t> prim;
    DevBuf<rt_counters> counters;
    // rt_profile_*: event pairs around k_trace_level launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    ~Scratch()
    {
        for (auto& e : events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
};

} // namespace

struct rt_scene {
    int device = 0;
    rt::HostScene host;
    DevBuf<float> vnormal, vtexcoord, texels;
    DevBuf<uint32_t> tri_vidx;
    DevBuf<rt::MaterialRec> materials;
    DevBuf<rt::TextureRec> textures;
    DevBuf<rt::LightRec> lights;
    uint32_t n_lights = 0;
    std::mutex mu;
};

struct rt_bvh {
    rt_scene* scene = nullptr;
    rt::HostBvh host;
    DevBuf<rt::NodePair> nodes;
    DevBuf<rt::PrimRecord> prims, prims_flat;
    DevBuf<rt::LeafRecord> leaf, leaf_flat, leaf_grouped;
    DevBuf<rt::GroupRecord> groups;
    DevBuf<rt::WideNode> wide_nodes;
    DevBuf<rt::LeafRecord> wide_leaf;
    DevBuf<uint32_t> wide_index, wide_ref_leaf;
    DevBuf<rt::LeafBox> wide_leaf_box;
#ifdef RT_DIAG
    DevBuf<rt::NodePair> diag_nodes;
    DevBuf<rt::LeafRecord> diag_leaf;
    DevBuf<rt::WideNode> diag_wnodes;
    DevBuf<rt::LeafRecord> diag_wleaf;
#endif
    std::mutex mu;
    bool profile = false;
    // rt_render_sharded: this handle's own stream on its device, the packed strips it renders, and (shard 0) the
    // gathered strips + assembled frame
    hipStream_t shard_stream = nullptr;
    hipEvent_t shard_done = nullptr;
    DevBuf<float> shard_packed, shard_gathered, shard_frame;
    // rt_render into a host buffer: row bands on two render streams, their copies back on a third (host_bands)
    hipStream_t host_stream[3] = { nullptr, nullptr, nullptr };
    hipEvent_t host_done[8] = {};
    DevBuf<float> host_frame;
    ~rt_bvh()
    {
        for (auto& e : host_done)
            if (e)
                (void)hipEventDestroy(e);
        for (auto& s_ : host_stream)
            if (s_)
                (void)hipStreamDestroy(s_);
        if (shard_done)
            (void)hipEventDestroy(shard_done);
        if (shard_stream)
            (void)hipStreamDestroy(shard_stream);
    }
    std::map<void*, std::unique_ptr<Scratch>> scratch; // keyed by stream
    Scratch& scratch_for(void* stream)
    {
        std::lock_guard<std::mutex> g(mu);
        auto& s = scratch[stream];
        if (!s)
            s.reset(new Scratch);
        return *s;
    }
    rt::SceneView view() const
    {
        rt::SceneView v;
        v.nodes = nodes.p;
        v.prims = prims.p;
        v.prims_flat = prims_flat.p;
        v.leaf = reinterpret_cast<const float4*>(leaf.p);
        v.leaf_flat = reinterpret_cast<const float4*>(leaf_flat.p);
        v.leaf_grouped = reinterpret_cast<const float4*>(leaf_grouped.p);
        v.groups = reinterpret_cast<const float4*>(groups.p);
        v.n_groups = (uint32_t)host.groups.size();
        v.tri_vidx = scene->tri_vidx.p;
        v.vnormal = scene->vnormal.p;
        v.vtexcoord = scene->vtexcoord.p;
        v.materials = scene->materials.p;
        v.textures = scene->textures.p;
        v.texels = scene->texels.p;
        v.lights = scene->lights.p;
        v.n_lights = scene->n_lights;
        v.root_ref = host.root_ref;
        v.root_cnt = host.root_cnt;
        v.n_prims = (uint32_t)host.records.size();
        v.n_pairs = (uint32_t)host.pairs.size();
        v.bounds_finite = host.bounds_finite ? 1u : 0u;
        v.prune_ok = scene->host.spheres.empty() ? 1u : 0u;
        v.extent = scene->host.extent;
        if (!host.nodes.empty()) { // post-order: the root is last
            memcpy(v.scene_lo, host.nodes.back().lo, 12);
            memcpy(v.scene_hi, host.nodes.back().hi, 12);
        } else {
            memset(v.scene_lo, 0, 12), memset(v.scene_hi, 0, 12);
        }
        v.leaf_flush = (uint32_t)g_leaf_flush.load();
        v.wide_nodes = wide_nodes.p;
        v.wide_leaf = reinterpret_cast<const float4*>(wide_leaf.p);
        v.wide_index = wide_index.p;
        v.wide_ref_leaf = wide_ref_leaf.p;
        v.wide_leaf_box = wide_leaf_box.p;
        v.wide_n_nodes = (uint32_t)host.wide_nodes.size();
        v.wide_n_groups = (uint32_t)(host.wide_slot_src.size() / 4);
        v.wide_n_ref_leaves = (uint32_t)host.wide_leaf_box.size();
        v.wide_ok = host.wide_ok && g_wide.load() ? 1u : 0u;
        v.wide_prefetch = g_prefetch.load() ? 1u : 0u;
        v.stamps = g_stamps;
#ifdef RT_DIAG
        v.diag_nodes = diag_nodes.p;
        v.diag_leaf = reinterpret_cast<const float4*>(diag_leaf.p);
        v.diag_wnodes = diag_wnodes.p, v.diag_wleaf = diag_wleaf.p;
#endif
        return v;
    }
};

namespace {

uint32_t feature_flags(const rt_features& f)
{
    uint32_t fl = 0;
    if (f.enableShading) fl |= rt::F_SHADING;
    if (f.enableRecursive) fl |= rt::F_RECURSIVE;
    if (f.enableHardShadow) fl |= rt::F_HARD_SHADOW;
    if (f.enableSoftShadow) fl |= rt::F_SOFT_SHADOW;
    if (f.enableNormalInterp) fl |= rt::F_NORMAL_INTERP;
    if (f.enableTextureMapping) fl |= rt::F_TEXTURE;
    if (f.enableAccelStructure) fl |= rt::F_ACCEL;
    if (f.enableBilinearTextureFiltering) fl |= rt::F_BILINEAR;
    if (g_unpruned.load()) fl |= rt::F_NO_PRUNE;
    if (f.enableTransparency) fl |= rt::F_TRANSPARENCY;
    if (f.enableGlossyReflection) fl |= rt::F_GLOSSY;
    return fl;
}

// Flags outside the hot-path scope fail loudly instead of rendering something else.
// (enableEnvironmentMapping / enableMotionBlur / enableMipmapTextureFiltering are
// never read by the reference's render path and are ignored like it does.)
int check_features(const rt_features* f, const rt_bvh* bvh)
{
    if (!f)
        return fail(RT_ERR_INVALID, "features == NULL");
    (void)bvh;
    return RT_OK;
}

int check_params(const rt_render_params* p)
{
    if (!p)
        return fail(RT_ERR_INVALID, "params == NULL");
    if (p->depth < 0 || p->depth > 61) // RNG streams 62 / 63 belong to the pixel samples
        return fail(RT_ERR_INVALID, "depth must be in [0, 61]");
    if (p->parallelogram_n < 0 || p->parallelogram_n > 181 || p->segment_n < 0 || p->segment_n > 32767)
        return fail(RT_ERR_INVALID, "light sample counts out of range");
    return RT_OK;
}

// the leaf phase's 48-byte view of the primitive records (rt_layout.h)
std::vector<rt::LeafRecord> leaf_records(const std::vector<rt::PrimRecord>& recs)
{
    std::vector<rt::LeafRecord> out(recs.size());
    for (size_t i = 0; i < recs.size(); i++) {
        const rt::PrimRecord& r = recs[i];
        rt::LeafRecord& l = out[i];
        memcpy(l.v0, r.v0, 12), memcpy(l.v1, r.v1, 12), memcpy(l.v2, r.v2, 12), memcpy(l.n, r.n, 12);
        if (r.kind_mat & 0x80000000u) {
            l.v1[0] = r.D; // radius
            l.n[0] = l.n[1] = l.n[2] = 0.0f;
        }
    }
    return out;
}

rt::CameraArgs camera_args(const rt_camera& c, int32_t w, int32_t h)
{
    rt::CameraArgs a;
    memset(&a, 0, sizeof(a));
    memcpy(a.origin, c.origin, 12);
    memcpy(a.quat, c.quat, 16);
    a.half_w = c.half_w, a.half_h = c.half_h;
    a.width = w, a.height = h;
    a.rows = nullptr, a.row0 = 0, a.n_rows = h;
    a.tile_wlog = (uint32_t)g_tile_wlog.load(), a.tile_hlog = (uint32_t)g_tile_hlog.load();
    a.super_log = (uint32_t)g_super_log.load();
    a.order = nullptr;
    a.vwidth = w, a.spp = 1, a.ms_n = 0, a.dof_n = 0;
    a.tile_x0 = a.tile_y0 = a.tiles_w = a.tiles_h = 0;
    memcpy(a.look_at, c.look_at, 12);
    a.focus = 0.0f, a.blur = 0.0f, a.seed = 0u;
    return a;
}

// the pixel loop's samples (render.cpp:295-314): n x n jittered rays and / or samplesDoF lens rays
void camera_samples(rt::CameraArgs& a, const rt_features& f, const rt_render_params& p)
{
    a.ms_n = f.enableMultipleRaysPerPixel ? p.rays_per_pixel_side : 0;
    a.dof_n = f.enableDepthOfField ? p.samples_dof : 0;
    a.spp = a.ms_n * a.ms_n + a.dof_n;
    if (a.spp == 0)
        a.spp = 1;
    a.vwidth = a.width * a.spp;
    a.focus = p.focus_plane_distance, a.blur = p.blur_strength, a.seed = p.seed;
}

// Dispatch order of the primary tiles: nearest to the image centre first. Workgroups start
// in index order, and with the recursion fused into one launch a tile's whole path tree is
// one wave's serial work; tiles that see geometry (usually around the centre) must start
// early or their chains form a tail with most of the chip idle (tools/timeline.py: half of
// the launch on the bench scene). Purely a scheduling choice — results do not depend on it.
// Launch only the tiles that can see the scene: the screen-space bounding rectangle of the
// (grown) scene bounds, two pixels wider on every side. Starting a one-wave workgroup costs
// ~1.9 ns of dispatcher time even if it exits at once (129 600 empty workgroups: 0.24 ms,
// tools/empty_frame.py), so tiles that cannot produce a hit are not launched at all; their
// pixels keep state 0 (= vec3(0)) from a memset. Conservative like tile_misses_scene, which
// still runs on the launched tiles. Returns false when the whole grid has to be launched.
bool host_launch_rect(const rt_bvh* bvh, rt::CameraArgs& cam)
{
    if (!g_tile_cull.load() || !g_launch_rect.load() || cam.super_log != 0 || cam.dof_n != 0 || bvh->host.nodes.empty()
        || !bvh->scene->host.spheres.empty())
        return false;
    const rt::RefNode& root = bvh->host.nodes.back();
    double S = bvh->scene->host.extent;
    for (int a = 0; a < 3; a++)
        S = std::max(S, (double)std::fabs(cam.origin[a]));
    const double g = S * 0.0009765625;
    // inverse camera rotation: conj(q) * v * q
    const double qw = cam.quat[0], qx = -cam.quat[1], qy = -cam.quat[2], qz = -cam.quat[3];
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int k = 0; k < 8; k++) {
        const double P[3] = { ((k & 1) ? root.hi[0] + g : root.lo[0] - g) - cam.origin[0],
            ((k & 2) ? root.hi[1] + g : root.lo[1] - g) - cam.origin[1],
            ((k & 4) ? root.hi[2] + g : root.lo[2] - g) - cam.origin[2] };
        const double uv[3] = { qy * P[2] - P[1] * qz, qz * P[0] - P[2] * qx, qx * P[1] - P[0] * qy };
        const double uuv[3] = { qy * uv[2] - uv[1] * qz, qz * uv[0] - uv[2] * qx, qx * uv[1] - uv[0] * qy };
        const double c[3] = { P[0] + (uv[0] * qw + uuv[0]) * 2.0, P[1] + (uv[1] * qw + uuv[1]) * 2.0,
            P[2] + (uv[2] * qw + uuv[2]) * 2.0 };
        if (!(c[2] > 1e-6 * S)) // a corner at or behind the camera plane: no finite rectangle
            return false;
        const double px = -c[0] / (c[2] * cam.half_w), py = c[1] / (c[2] * cam.half_h);
        const double x = (px + 1.0) * 0.5 * cam.width, y = (py + 1.0) * 0.5 * cam.height;
        xmin = std::min(xmin, x), xmax = std::max(xmax, x), ymin = std::min(ymin, y), ymax = std::max(ymax, y);
    }
    if (!(xmin == xmin) || !(ymin == ymin))
        return false;
    const int64_t tw = 1 << cam.tile_wlog, th = 1 << cam.tile_hlog;
    const int64_t tiles_x = (cam.vwidth + tw - 1) / tw, tiles_y = (cam.n_rows + th - 1) / th;
    int64_t x0 = (int64_t)std::floor(xmin) - 2, x1 = (int64_t)std::ceil(xmax) + 2; // pixel columns, inclusive
    int64_t y0 = (int64_t)std::floor(ymin) - 2, y1 = (int64_t)std::ceil(ymax) + 2;
    x0 = std::max<int64_t>(x0, 0), x1 = std::min<int64_t>(x1, cam.width - 1);
    y0 = std::max<int64_t>(y0, 0), y1 = std::min<int64_t>(y1, cam.height - 1);
    int64_t tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    if (x0 <= x1 && y0 <= y1) {
        tx0 = (x0 * cam.spp) / tw, tx1 = ((x1 + 1) * cam.spp - 1) / tw;
        if (cam.rows) { // an arbitrary row list: keep every tile row
            ty0 = 0, ty1 = tiles_y - 1;
        } else {
            const int64_t l0 = std::max<int64_t>(y0 - cam.row0, 0), l1 = std::min<int64_t>(y1 - cam.row0, cam.n_rows - 1);
            if (l0 <= l1)
                ty0 = l0 / th, ty1 = l1 / th;
        }
    }
    cam.tile_x0 = (uint32_t)tx0, cam.tile_y0 = (uint32_t)ty0;
    cam.tiles_w = (uint32_t)std::max<int64_t>(tx1 - tx0 + 1, 0), cam.tiles_h = (uint32_t)std::max<int64_t>(ty1 - ty0 + 1, 0);
    if (cam.tiles_w == 0 || cam.tiles_h == 0) { // nothing can be hit: one (culled) tile keeps the launch path uniform
        cam.tile_x0 = cam.tile_y0 = 0;
        cam.tiles_w = cam.tiles_h = 1;
    }
    return !(cam.tiles_w == (uint32_t)tiles_x && cam.tiles_h == (uint32_t)tiles_y);
}

// Dispatch order of the launched tiles: rings around the centre of the launched grid, innermost
// first (O(n), no sort: the grid changes with the camera). Workgroups start in index order, and
// with the recursion fused into one launch a tile's whole path tree is one wave's serial work;
// tiles that see geometry (around the centre of the scene's screen rectangle) must start early
// or their chains form a tail with most of the chip idle (tools/timeline.py). Purely a
// scheduling choice — results do not depend on it.
// zero / n_zero: words the caller wants cleared on `stream` before its trace launch; *zeroed tells whether the
// order launch did that on its way.
int tile_order_for(Scratch& sc, rt::CameraArgs& cam, hipStream_t stream, uint32_t* zero, uint32_t n_zero, bool* zeroed)
{
    *zeroed = false;
    if (!g_tile_order.load() || cam.super_log != 0)
        return RT_OK;
    const int64_t tw = 1 << cam.tile_wlog, th = 1 << cam.tile_hlog;
    const int64_t gw = cam.tiles_w ? cam.tiles_w : (cam.vwidth + tw - 1) / tw;
    const int64_t gh = cam.tiles_w ? cam.tiles_h : (cam.n_rows + th - 1) / th;
    // tiles are tw x th pixels: measure rings in pixels so that they are round on the screen
    // ring centre: the image centre, in tiles of the launched grid (may lie outside it)
    // several frames in the launch: every frame's rows get the rings of ONE frame (period = a frame's tile rows)
    const int64_t frame_rows = cam.n_frames > 1 ? cam.frame_rows : cam.n_rows;
    const int64_t period = cam.n_frames > 1 ? frame_rows / th : gh + (int64_t)cam.tile_y0;
    const int64_t lrow_mid = cam.rows ? frame_rows / 2 : std::min<int64_t>(std::max<int64_t>(cam.height / 2 - cam.row0, 0), cam.n_rows - 1);
    const int64_t cx = (cam.vwidth / 2) / tw - (int64_t)cam.tile_x0, cy = lrow_mid / th - (int64_t)cam.tile_y0;
    const int64_t xcd = g_xcd_order.load() ? 1 : 0;
    // screen blocks of 2^xcd_block pixels a side, in tiles
    const uint32_t bpl = (uint32_t)g_xcd_block.load();
    const uint32_t bxlog = bpl > cam.tile_wlog ? bpl - cam.tile_wlog : 0u, bylog = bpl > cam.tile_hlog ? bpl - cam.tile_hlog : 0u;
    const std::vector<int64_t> key { gw, gh, tw, th, cx, cy, period, xcd, (int64_t)cam.tile_x0, (int64_t)cam.tile_y0, (int64_t)bxlog, (int64_t)bylog, (int64_t)g_frame_major.load(), (int64_t)g_order_one_launch.load() };
    if (key != sc.order_key || !sc.order.p) {
        // built on the device (k_tile_ring_*): the launched grid follows the camera, and a host-built order
        // would put an upload + synchronisation into every launch of a moving camera
        const size_t n = (size_t)(gw * gh);
        const int64_t rmax = std::max(std::max(std::llabs(cx), std::llabs(gw - 1 - cx)) * tw,
            std::max(std::llabs(cy), std::llabs(std::min(gh, period) - 1 - cy)) * th);
        // rings four tiles wide: the order inside a ring is arbitrary anyway, and the 64 consecutive tiles of a
        // sorting wave then fall into a handful of rings (one atomic each) instead of one ring per tile
        const int64_t step = 4 * std::min(tw, th);
        rt::TileOrderArgs o;
        o.gw = (uint32_t)gw, o.n_tiles = (uint32_t)n;
        o.xcd_order = (uint32_t)xcd, o.xcd_bxlog = bxlog, o.xcd_bylog = bylog, o.tile_x0 = cam.tile_x0, o.tile_y0 = cam.tile_y0;
        o.tw = tw, o.th = th, o.cx = cx, o.cy = cy, o.period = period, o.step = step;
        o.n_rings = (uint32_t)(rmax / step + 2);
        o.n_frames = (uint32_t)std::max<int64_t>(1, (gh + period - 1) / period);
        o.frame_major = g_frame_major.load() ? 1u : 0u;
        // sized for the whole tile grid once: a reallocation frees memory, which waits for the device
        HIP_TRY(sc.order.reserve((size_t)(((cam.vwidth + tw - 1) / tw) * ((cam.n_rows + th - 1) / th))));
        HIP_TRY(sc.order_rings.reserve(8u * (size_t)rt::kMaxFramesPerLaunch * ((size_t)cam.vwidth + (size_t)cam.n_rows + 4u)));
        if (n > sc.order.n || 8u * (size_t)o.n_frames * (size_t)o.n_rings > sc.order_rings.n)
            return fail(RT_ERR_INVALID, "internal error: tile order scratch too small");
        o.rings = sc.order_rings.p, o.order = sc.order.p;
        // XCD-aware order: the lengths of the eight lists (tile_xcd over the launched grid). A block column's XCD is
        // (bx + 3 by) % 8: count the tiles of a row per residue of bx once, then rotate that by 3 by per tile row.
        memset(sc.xcd_count, 0, sizeof(sc.xcd_count));
        if (xcd) {
            const int64_t B = (int64_t)1 << bxlog;
            uint32_t per_res[8] = {};
            for (int64_t tx = 0; tx < gw;) {
                const int64_t atx = tx + (int64_t)cam.tile_x0;
                const int64_t run = std::min<int64_t>(B - (atx & (B - 1)), gw - tx);
                per_res[(atx >> bxlog) & 7] += (uint32_t)run;
                tx += run;
            }
            for (int64_t ty = 0; ty < gh; ty++) {
                const uint32_t by = ((uint32_t)(ty % period) + cam.tile_y0) >> bylog;
                for (uint32_t c = 0; c < 8; c++)
                    sc.xcd_count[rt::tile_xcd(c << bxlog, by << bylog, bxlog, bylog)] += per_res[c];
            }
        }
        const uint64_t keys_per_list = (uint64_t)o.n_rings * (xcd ? o.n_frames : 1u);
        // one launch for a lone frame (it waits for its tile order: count / scan / scatter + a memset cost it ~30 us
        // more); launches of several frames run beside other launches, and there the three kernels — which queue up
        // behind the running trace kernel and so stagger the launches — measured 1 % better (profiles/r02_notes.md)
        const int one = g_order_one_launch.load();
        o.one_launch = (one == 2 || (one == 1 && o.n_frames == 1)) && keys_per_list <= rt::kTileOrderLdsKeys ? 1u : 0u;
        o.zero = nullptr, o.n_zero = 0;
        for (uint32_t k = 0, base = 0; k < 8; k++) {
            o.list_base[k] = xcd ? base : 0u;
            base += sc.xcd_count[k];
        }
        if (o.one_launch && zero)
            o.zero = zero, o.n_zero = n_zero, *zeroed = true;
        rt::launch_tile_order(o, stream);
        HIP_TRY(hipGetLastError());
        sc.order_key = key;
    }
    cam.xcd_order = (uint32_t)xcd;
    if (xcd) {
        uint32_t base = 0, longest = 0;
        for (int k = 0; k < 8; k++) {
            cam.xcd_base[k] = base, cam.xcd_count[k] = sc.xcd_count[k];
            base += sc.xcd_count[k];
            longest = std::max(longest, sc.xcd_count[k]);
        }
        cam.launch_waves = 8u * longest;
    }
    cam.order = sc.order.p;
    return RT_OK;
}

// Enqueue the whole flattened recursion for n_items rays on `stream`:
// level kernels, then the fold into fold.out.
int enqueue_trace(rt_bvh* bvh, Scratch& sc, const rt_features& f, const rt_render_params& p, int32_t source,
    const rt::CameraArgs& cam, const rt::RaySoA& user_rays, uint32_t n_items, rt::FoldArgs fold,
    rt_counters* counters_dev, hipStream_t stream)
{
    const uint32_t flags = feature_flags(f);
    const int32_t n_levels = (flags & rt::F_RECURSIVE) ? p.depth + 1 : 1;
    HIP_TRY(sc.direct.reserve((size_t)n_levels * n_items * 3));
    HIP_TRY(sc.state.reserve(n_items));
    HIP_TRY(sc.counts.reserve((size_t)n_levels + 1));
    if (n_levels > 1)
        for (int b = 0; b < 2; b++) {
            HIP_TRY(sc.qray[b].reserve((size_t)n_items * 6));
            HIP_TRY(sc.qitem[b].reserve(n_items));
            HIP_TRY(sc.qpixel[b].reserve(n_items));
        }
    if (counters_dev)
        HIP_TRY(hipMemsetAsync(counters_dev, 0, sizeof(rt_counters), stream));

    rt::LevelArgs a;
    memset(&a, 0, sizeof(a));
    a.sv = bvh->view();
    a.flags = flags;
    a.depth = n_levels - 1;
    a.parallelogram_n = p.parallelogram_n, a.segment_n = p.segment_n, a.seed = p.seed;
    a.min_waves = (uint32_t)g_min_waves.load();
    const bool fused = g_fused.load() != 0;
    // fused_from = k: levels below k get a launch each (rays re-packed through the device
    // queue between them), level k's launch then runs all deeper levels inside its waves
    const int32_t fused_from = fused ? std::min(g_fused_from.load(), n_levels - 1) : n_levels;
    a.merge_next = g_merge_next.load() ? 1u : 0u;
    bool area_lights = false;
    if (f.enableSoftShadow)
        for (const rt::LightRec& l : bvh->scene->host.lights)
            if (l.kind != RT_LIGHT_POINT)
                area_lights = true;
    const bool tris_only = bvh->scene->host.spheres.empty(), attrs = f.enableNormalInterp || f.enableTextureMapping;
    a.variant = tris_only && !area_lights ? (attrs ? 2u : 1u) : 0u;
    a.tile_cull = g_tile_cull.load() && !counters_dev ? 1u : 0u; // the counting pass traces every primary ray
    a.tile_cull_count = g_tile_cull.load() && counters_dev ? 1u : 0u;
    a.skip_unlit = g_skip_unlit.load() ? 1u : 0u;
    a.level_stride = (size_t)n_items * 3;
    a.cam = cam;
    if (g_tile_auto.load() && source == rt::SRC_CAMERA && cam.n_frames <= 1 && cam.super_log == 0) {
        // one frame in the launch: the tile shape that suits the frame (g_tile_auto)
        // point lights on triangles: a row of 32 pixels per wave (64 from 4 Mpixel up); area lights, spheres: 4 x 4
        const bool rows = a.variant != 0u, big = (int64_t)cam.vwidth * cam.height >= (int64_t)4 << 20;
        a.cam.tile_wlog = rows ? (big ? 6u : 5u) : 2u, a.cam.tile_hlog = rows ? 0u : 2u;
        a.sv.leaf_flush = rows ? (big ? 32u : 24u) : 16u;
        if (a.sv.leaf_flush > rt::kWideFlushMax && a.sv.wide_ok)
            a.sv.leaf_flush = rt::kWideFlushMax;
    }
    bool rect = false, counts_zeroed = false;
    if (source == rt::SRC_CAMERA && !counters_dev && (flags & rt::F_ACCEL) && !(flags & rt::F_NO_PRUNE)) {
        rect = host_launch_rect(bvh, a.cam);
        if (cam.n_frames > 1) { // the union of the frames' rectangles (their row ranges are all kept: a row list)
            for (int32_t k = 0; k < cam.n_frames && k < rt::kMaxFramesPerLaunch; k++) {
                a.poses[k] = sc.poses[(size_t)k];
                rt::CameraArgs ck = cam;
                memcpy(ck.origin, a.poses[k].origin, sizeof(ck.origin));
                memcpy(ck.quat, a.poses[k].quat, sizeof(ck.quat));
                const bool rk = host_launch_rect(bvh, ck);
                if (k == 0)
                    a.cam.tile_x0 = ck.tile_x0, a.cam.tiles_w = ck.tiles_w, a.cam.tile_y0 = ck.tile_y0, a.cam.tiles_h = ck.tiles_h;
                rect = (k == 0 ? rk : rect && rk);
                if (rect && k > 0) {
                    const uint32_t x0 = std::min(a.cam.tile_x0, ck.tile_x0);
                    const uint32_t x1 = std::max(a.cam.tile_x0 + a.cam.tiles_w, ck.tile_x0 + ck.tiles_w);
                    a.cam.tile_x0 = x0, a.cam.tiles_w = x1 - x0;
                }
            }
            if (rect) { // every tile row of every frame
                const uint32_t th = 1u << a.cam.tile_hlog;
                a.cam.tile_y0 = 0, a.cam.tiles_h = ((uint32_t)a.cam.n_rows + th - 1u) / th;
            }
        }
    } else if (cam.n_frames > 1) {
        for (int32_t k = 0; k < cam.n_frames && k < rt::kMaxFramesPerLaunch; k++)
            a.poses[k] = sc.poses[(size_t)k];
    }
    // recursiveRayTrace's branching variants (render.cpp:36-49 transparency chain, :56-99 glossy,
    // :101-130 non-opaque materials under enableRecursive) run as per-lane state machines (k_trace_general)
    const bool general = f.enableGlossyReflection || f.enableTransparency
        || (f.enableRecursive && bvh->scene->host.any_non_opaque);
    if (source == rt::SRC_CAMERA) {
        // the counting pass traces every primary ray; otherwise launch only the tiles that can see the scene
        if (rect) {
            // k_fold treats the items of tiles outside the launched grid as untraced (vec3(0))
            fold.vwidth = (uint32_t)a.cam.vwidth, fold.tile_wlog = a.cam.tile_wlog, fold.tile_hlog = a.cam.tile_hlog;
            fold.tile_x0 = a.cam.tile_x0, fold.tile_y0 = a.cam.tile_y0;
            fold.tiles_w = a.cam.tiles_w, fold.tiles_h = a.cam.tiles_h;
        } else {
            a.cam.tile_x0 = a.cam.tile_y0 = a.cam.tiles_w = a.cam.tiles_h = 0;
        }
        const int rc = tile_order_for(sc, a.cam, stream, sc.counts.p, (uint32_t)n_levels + 1u, &counts_zeroed);
        if (rc != RT_OK)
            return rc;
    }
    if (!counts_zeroed)
        HIP_TRY(hipMemsetAsync(sc.counts.p, 0, ((size_t)n_levels + 1) * sizeof(uint32_t), stream));
    a.state = sc.state.p;
    a.n_items = n_items;
    a.counters = counters_dev;
    // recursiveRayTrace's branching variants (render.cpp:36-49 transparency chain, :56-99 glossy,
    // :101-130 non-opaque materials under enableRecursive) run as per-lane state machines
    if (general) {
        if (f.enableGlossyReflection && (p.rays_per_reflection < 1 || p.rays_per_reflection > 4096 || p.glossy_reflections_cap < 0))
            return fail(RT_ERR_INVALID, "rays_per_reflection must be in [1, 4096], glossy_reflections_cap >= 0");
        a.level = 0, a.fused = 0;
        a.depth = p.depth;
        a.direct = sc.direct.p;
        a.source = source, a.in = user_rays, a.in_count = nullptr, a.in_count_imm = n_items;
        a.max_frames = p.depth + rt::kExtraFrames + 1;
        a.glossy_cap = p.glossy_reflections_cap, a.rays_per_reflection = p.rays_per_reflection;
        a.alpha_modifier = p.alpha_modifier;
        const uint32_t waves = rt::trace_general_waves(a);
        HIP_TRY(sc.frames.reserve((size_t)waves * 64 * (size_t)a.max_frames * rt::kFrameWords));
        a.frames = sc.frames.p;
        if (bvh->profile) {
            if (sc.events_used == sc.events.size()) {
                hipEvent_t e0, e1;
                HIP_TRY(hipEventCreate(&e0));
                HIP_TRY(hipEventCreate(&e1));
                sc.events.emplace_back(e0, e1);
            }
            HIP_TRY(hipEventRecord(sc.events[sc.events_used].first, stream));
        }
        rt::launch_trace_general(a, counters_dev != nullptr, stream);
        if (bvh->profile) {
            HIP_TRY(hipEventRecord(sc.events[sc.events_used].second, stream));
            sc.events_used++;
        }
    }
    for (int32_t level = 0; !general && level <= std::min(fused_from, n_levels - 1); level++) {
        a.level = level;
        a.fused = level == fused_from ? 1u : 0u;
        a.direct = sc.direct.p + (size_t)level * n_items * 3;
        if (level == 0) {
            a.source = source;
            a.in = user_rays;
            a.in_count = nullptr;
            a.in_count_imm = n_items;
        } else {
            const int b = (level - 1) & 1;
            a.source = rt::SRC_QUEUE;
            a.in = rt::RaySoA { sc.qray[b].p, nullptr, sc.qitem[b].p, sc.qpixel[b].p, n_items };
            a.in_count = sc.counts.p + level;
        }
        const int ob = level & 1;
        a.out = rt::RaySoA { sc.qray[ob].p, nullptr, sc.qitem[ob].p, sc.qpixel[ob].p, n_items };
        a.out_count = sc.counts.p + level + 1;
        if (bvh->profile) {
            if (sc.events_used == sc.events.size()) {
                hipEvent_t e0, e1;
                HIP_TRY(hipEventCreate(&e0));
                HIP_TRY(hipEventCreate(&e1));
                sc.events.emplace_back(e0, e1);
            }
            HIP_TRY(hipEventRecord(sc.events[sc.events_used].first, stream));
        }
        rt::launch_trace_level(a, counters_dev != nullptr, stream);
        if (bvh->profile) {
            HIP_TRY(hipEventRecord(sc.events[sc.events_used].second, stream));
            sc.events_used++;
        }
    }
    fold.direct = sc.direct.p;
    fold.state = sc.state.p;
    fold.n_items = n_items;
    fold.level_stride = (size_t)n_items * 3;
    rt::launch_fold(fold, stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// AoS rt_ray[] → SoA upload
int upload_rays(Scratch& sc, const rt_ray* rays, uint32_t n, const uint32_t* pixel_ids, hipStream_t stream,
    rt::RaySoA& out)
{
    std::vector<float> soa((size_t)n * 6), tt(n);
    for (uint32_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            soa[(size_t)k * n + i] = rays[i].origin[k];
            soa[(size_t)(3 + k) * n + i] = rays[i].direction[k];
        }
        tt[i] = rays[i].t;
    }
    HIP_TRY(sc.in_ray.reserve((size_t)n * 6));
    HIP_TRY(sc.in_t.reserve(n));
    HIP_TRY(hipMemcpyAsync(sc.in_ray.p, soa.data(), soa.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(sc.in_t.p, tt.data(), tt.size() * 4, hipMemcpyHostToDevice, stream));
    out = rt::RaySoA { sc.in_ray.p, sc.in_t.p, nullptr, nullptr, n };
    if (pixel_ids) {
        HIP_TRY(sc.in_pixel.reserve(n));
        HIP_TRY(hipMemcpyAsync(sc.in_pixel.p, pixel_ids, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        out.pixel = sc.in_pixel.p;
    }
    HIP_TRY(hipStreamSynchronize(stream)); // the staging vectors die at return
    return RT_OK;
}

} // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_info(int* device_count, char* name_buf, size_t name_buf_len)
{
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_count)
        *device_count = n;
    if (n == 0)
        return fail(RT_ERR_DEVICE, "no HIP device");
    if (name_buf && name_buf_len) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        snprintf(name_buf, name_buf_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return RT_OK;
}

int rt_set_option(const char* name, int value)
{
    if (name && strcmp(name, "unpruned") == 0) {
        g_unpruned.store(value != 0);
        return RT_OK;
    }
    if (name && (strcmp(name, "tile_wlog") == 0 || strcmp(name, "tile_hlog") == 0)) {
        // a tile is one wave's primary rays: at most 64 pixels; its height divides the 8-row strips
        const bool w_ = name[5] == 'w';
        if (value < 0 || value > (w_ ? 6 : 3) || value + (w_ ? g_tile_hlog : g_tile_wlog).load() > 6)
            return fail(RT_ERR_INVALID, "rt_set_option: tile_wlog in [0, 6], tile_hlog in [0, 3], tile_wlog + tile_hlog <= 6");
        (w_ ? g_tile_wlog : g_tile_hlog).store(value);
        g_tile_auto.store(0); // an explicit shape is meant
        return RT_OK;
    }
    if (name && strcmp(name, "fused") == 0) { // 1: all recursion levels in one launch (default); 0: one launch per level
        g_fused.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "launch_rect") == 0) { // 1: launch only the tiles inside the scene's screen rectangle
        g_launch_rect.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "tile_cull") == 0) { // 1: a primary tile whose rays cannot reach the scene bounds is finished at once
        g_tile_cull.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "merge_next") == 0) { // 1: shadow rays of level k share a traversal with the reflection rays of level k + 1
        g_merge_next.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "fused_from") == 0) { // first recursion level that runs fused (0 = one launch per frame)
        if (value < 0 || value > 64)
            return fail(RT_ERR_INVALID, "rt_set_option: fused_from out of range");
        g_fused_from.store(value);
        return RT_OK;
    }
    if (name && strcmp(name, "tile_order") == 0) { // 1: centre-first dispatch of the primary tiles (default); 0: row-major
        g_tile_order.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "super_log") == 0) {
        if (value < 0 || value > 6)
            return fail(RT_ERR_INVALID, "rt_set_option: super_log must be in [0, 6]");
        g_super_log.store(value);
        return RT_OK;
    }
    if (name && strcmp(name, "min_waves") == 0) {
        if (value < 1)
            return fail(RT_ERR_INVALID, "rt_set_option: min_waves must be >= 1");
        g_min_waves.store(value);
        return RT_OK;
    }
    if (name && strcmp(name, "lds_pad") == 0) { // tuning experiment: extra dynamic LDS lowers occupancy
        rt::g_lds_pad = (uint32_t)value;
        return RT_OK;
    }
    if (name && strcmp(name, "wide") == 0) {
        g_wide.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "xcd_order") == 0) {
        g_xcd_order.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "xcd_block") == 0) { // screen blocks of 2^v x 2^v pixels go to one XCD
        if (value < 0 || value > 10)
            return fail(RT_ERR_INVALID, "rt_set_option: xcd_block must be in [0, 10]");
        g_xcd_block.store(value);
        return RT_OK;
    }
    if (name && strcmp(name, "frame_major") == 0) { // several frames per launch: an XCD's tile list frame after frame (1) or centre first over all frames (0)
        g_frame_major.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "order_one_launch") == 0) { // the tile order in one launch (counters in LDS): 1 for lone frames, 2 for every launch; 0: count / scan / scatter kernels
        if (value < 0 || value > 2)
            return fail(RT_ERR_INVALID, "rt_set_option: order_one_launch must be 0, 1 (lone frames) or 2 (every launch)");
        g_order_one_launch.store(value);
        return RT_OK;
    }
    if (name && strcmp(name, "skip_unlit") == 0) { // 0: trace the shadow rays of unlit hits too (what the reference does; same bits)
        g_skip_unlit.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "prefetch") == 0) {
        g_prefetch.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "host_bands") == 0) {
        if (value < 0 || value > 8)
            return fail(RT_ERR_INVALID, "rt_set_option: host_bands must be in [0, 8] (0 = automatic)");
        g_host_bands.store(value);
        return RT_OK;
    }
    if (name && strcmp(name, "tile_auto") == 0) {
        g_tile_auto.store(value != 0);
        return RT_OK;
    }
    if (name && strcmp(name, "leaf_flush") == 0) {
        if (value < 1 || value > (int)rt::kLeafFlushMax)
            return fail(RT_ERR_INVALID, "rt_set_option: leaf_flush must be in [1, 36]");
        g_leaf_flush.store(value);
        return RT_OK;
    }
    return fail(RT_ERR_INVALID, std::string("rt_set_option: unknown option ") + (name ? name : "(null)"));
}

int rt_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* desc, rt_scene** out)
{
    if (!desc || !out)
        return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    std::unique_ptr<rt_scene> s(new rt_scene);
    const std::string err = s->host.load(*desc);
    if (!err.empty())
        return fail(RT_ERR_INVALID, "rt_scene_create: " + err);
    HIP_TRY(hipGetDevice(&s->device));
    HIP_TRY(s->vnormal.upload(s->host.normals));
    HIP_TRY(s->vtexcoord.upload(s->host.texcoords));
    HIP_TRY(s->texels.upload(s->host.texels));
    HIP_TRY(s->tri_vidx.upload(s->host.tri_vidx));
    HIP_TRY(s->materials.upload(s->host.materials));
    HIP_TRY(s->textures.upload(s->host.textures));
    HIP_TRY(s->lights.upload(s->host.lights));
    s->n_lights = (uint32_t)s->host.lights.size();
    *out = s.release();
    return RT_OK;
}

int rt_scene_destroy(rt_scene* scene)
{
    delete scene;
    return RT_OK;
}

int rt_scene_update_lights(rt_scene* scene, const rt_light_desc* lights, uint32_t n_lights)
{
    if (!scene || (n_lights && !lights))
        return fail(RT_ERR_INVALID, "rt_scene_update_lights: null argument");
    std::vector<rt::LightRec> recs;
    const std::string err = rt::HostScene::convert_lights(lights, n_lights, recs);
    if (!err.empty())
        return fail(RT_ERR_INVALID, "rt_scene_update_lights: " + err);
    std::lock_guard<std::mutex> g(scene->mu);
    HIP_TRY(hipDeviceSynchronize()); // frames in flight still read the old lights
    if (recs.size() > scene->lights.n) {
        HIP_TRY(scene->lights.upload(recs));
    } else if (!recs.empty()) {
        HIP_TRY(hipMemcpy(scene->lights.p, recs.data(), recs.size() * sizeof(rt::LightRec), hipMemcpyHostToDevice));
    }
    scene->host.lights = recs;
    scene->n_lights = n_lights;
    return RT_OK;
}

int rt_bvh_build(rt_scene* scene, const rt_features* features, rt_bvh** out)
{
    if (!scene || !features || !out)
        return fail(RT_ERR_INVALID, "rt_bvh_build: null argument");
    std::unique_ptr<rt_bvh> b(new rt_bvh);
    b->scene = scene;
    const std::string err = b->host.build(scene->host, features->enableBvhSahBinning != 0);
    if (!err.empty())
        return fail(RT_ERR_INVALID, "rt_bvh_build: " + err);
    HIP_TRY(b->nodes.upload(b->host.pairs));
    HIP_TRY(b->prims.upload(b->host.records));
    HIP_TRY(b->prims_flat.upload(b->host.records_flat));
    HIP_TRY(b->leaf.upload(leaf_records(b->host.records)));
    HIP_TRY(b->leaf_flat.upload(leaf_records(b->host.records_flat)));
    {
        std::vector<rt::PrimRecord> g(b->host.grouped_src.size());
        for (size_t i = 0; i < g.size(); i++)
            if (b->host.grouped_src[i] != 0xffffffffu)
                g[i] = b->host.records[b->host.grouped_src[i]];
            else
                memset(&g[i], 0, sizeof(g[i]));
        HIP_TRY(b->leaf_grouped.upload(leaf_records(g)));
        HIP_TRY(b->groups.upload(b->host.groups));
    }
    if (b->host.wide_ok) { // the wide tree (rt_wide.h): nodes, leaf records per slot, slot meta, reference leaf boxes
        std::vector<rt::PrimRecord> g(b->host.wide_slot_src.size());
        for (size_t i = 0; i < g.size(); i++)
            if (b->host.wide_slot_src[i] != 0xffffffffu)
                g[i] = b->host.records[b->host.wide_slot_src[i]];
            else
                memset(&g[i], 0, sizeof(g[i]));
        HIP_TRY(b->wide_leaf.upload(leaf_records(g)));
        HIP_TRY(b->wide_nodes.upload(b->host.wide_nodes));
#ifdef RT_DIAG
        HIP_TRY(b->diag_wleaf.upload(leaf_records(g)));
        HIP_TRY(b->diag_wnodes.upload(b->host.wide_nodes));
#endif
        HIP_TRY(b->wide_index.upload(b->host.wide_index));
        HIP_TRY(b->wide_ref_leaf.upload(b->host.ref_leaf_of));
        HIP_TRY(b->wide_leaf_box.upload(b->host.wide_leaf_box));
    }
#ifdef RT_DIAG
    HIP_TRY(b->diag_nodes.upload(b->host.pairs));
    HIP_TRY(b->diag_leaf.upload(leaf_records(b->host.records)));
#if RT_DIAG == 2
    rt::set_diag_leaf(reinterpret_cast<const float4*>(b->diag_leaf.p));
#endif
#endif
    *out = b.release();
    return RT_OK;
}

int rt_bvh_destroy(rt_bvh* bvh)
{
    if (bvh)
        (void)hipDeviceSynchronize();
    delete bvh;
    return RT_OK;
}

int rt_bvh_num_levels(const rt_bvh* bvh) { return bvh ? bvh->host.num_levels : 0; }
int rt_bvh_num_leaves(const rt_bvh* bvh) { return bvh ? bvh->host.num_leaves : 0; }
int rt_bvh_num_nodes(const rt_bvh* bvh) { return bvh ? (int)bvh->host.nodes.size() : 0; }
int rt_bvh_num_prims(const rt_bvh* bvh) { return bvh ? (int)bvh->host.prim_order.size() : 0; }

int rt_bvh_wide_check(const rt_bvh* bvh, uint64_t stats[8])
{
    if (!bvh || !stats)
        return fail(RT_ERR_INVALID, "rt_bvh_wide_check: null argument");
    const std::string err = bvh->host.check_wide(stats);
    stats[6] = bvh->host.wide_ok ? 1u : 0u;
    if (!err.empty())
        return fail(RT_ERR_INVALID, "rt_bvh_wide_check: " + err);
    return RT_OK;
}

int rt_bvh_export(const rt_bvh* bvh, float* node_aabb, int32_t* node_meta, int32_t* prim_order)
{
    if (!bvh)
        return fail(RT_ERR_INVALID, "rt_bvh_export: null bvh");
    for (size_t i = 0; i < bvh->host.nodes.size(); i++) {
        const rt::RefNode& n = bvh->host.nodes[i];
        if (node_aabb) {
            memcpy(node_aabb + 6 * i, n.lo, 12);
            memcpy(node_aabb + 6 * i + 3, n.hi, 12);
        }
        if (node_meta) {
            int32_t* m = node_meta + 6 * i;
            m[0] = n.leaf, m[1] = n.depth, m[2] = n.beg, m[3] = n.end, m[4] = n.left, m[5] = n.right;
        }
    }
    if (prim_order)
        memcpy(prim_order, bvh->host.prim_order.data(), bvh->host.prim_order.size() * sizeof(int32_t));
    return RT_OK;
}

int rt_camera_from_trackball(float fovy, float aspect, const float look_at[3], float distance,
    const float rot[3], rt_camera* out)
{
    if (!look_at || !rot || !out)
        return fail(RT_ERR_INVALID, "rt_camera_from_trackball: null argument");
    // Trackball ctor (trackball.cpp:26-27)
    out->half_h = std::tan(fovy / 2.0f);
    out->half_w = aspect * out->half_h;
    // glm::quat(vec3 eulerAngle), type_quat.inl:208-217
    const float cx = std::cos(rot[0] * 0.5f), cy = std::cos(rot[1] * 0.5f), cz = std::cos(rot[2] * 0.5f);
    const float sx = std::sin(rot[0] * 0.5f), sy = std::sin(rot[1] * 0.5f), sz = std::sin(rot[2] * 0.5f);
    const float w = cx * cy * cz + sx * sy * sz;
    const float x = sx * cy * cz - cx * sy * sz;
    const float y = cx * sy * cz + sx * cy * sz;
    const float z = cx * cy * sz - sx * sy * cz;
    out->quat[0] = w, out->quat[1] = x, out->quat[2] = y, out->quat[3] = z;
    // position() = lookAt + quat * vec3(0, 0, -dist) (trackball.cpp:71-74); quat*vec: type_quat.inl:347-354
    const float v[3] = { 0.0f, 0.0f, -distance };
    const float uv[3] = { y * v[2] - v[1] * z, z * v[0] - v[2] * x, x * v[1] - v[0] * y };
    const float uuv[3] = { y * uv[2] - uv[1] * z, z * uv[0] - uv[2] * x, x * uv[1] - uv[0] * y };
    for (int k = 0; k < 3; k++)
        out->origin[k] = look_at[k] + (v[k] + ((uv[k] * w) + uuv[k]) * 2.0f);
    memcpy(out->look_at, look_at, 12);
    return RT_OK;
}

namespace {

// weightsGaussian(sigma), src/render.cpp:198-210 (answer[i + 1][k + 1], then / sum). The
// reference's unqualified exp() on a float argument is ::exp(double) with <cmath> alone.
void weights_gaussian(float sigma, float out[3][3])
{
    float sum = 0.0f;
    for (int i = -1; i < 2; i++) {
        for (int k = -1; k < 2; k++) {
            const float weight = (float)(::exp((double)(-(i * i + k * k) / (2 * sigma * sigma))) / (2 * 3.1415 * sigma * sigma));
            out[i + 1][k + 1] = weight;
            sum += weight;
        }
    }
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            out[i][k] = out[i][k] / sum;
}

// renderBloomFilter on a whole row-flipped frame in device memory
int enqueue_bloom(Scratch& sc, float* frame_dev, int32_t width, int32_t height, const rt_render_params& p,
    hipStream_t stream)
{
    const size_t floats = (size_t)width * height * 3;
    HIP_TRY(sc.bloom_src.reserve(floats));
    HIP_TRY(hipMemcpyAsync(sc.bloom_src.p, frame_dev, floats * sizeof(float), hipMemcpyDeviceToDevice, stream));
    rt::BloomArgs b;
    b.src = sc.bloom_src.p, b.frame = frame_dev, b.width = width, b.height = height;
    weights_gaussian(1.0f, b.w);
    b.scalar = p.bloom_scalar, b.threshold = p.bloom_threshold, b.debug_option = p.bloom_debug_option;
    rt::launch_bloom(b, stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ≙ the pixel loop of renderRayTracing (render.cpp:283-324) for the rows described by `cam`
// (+ renderBloomFilter when the whole frame is rendered). `fold` carries the output mapping.
int enqueue_frame(rt_bvh* bvh, Scratch& sc, const rt_features& f, const rt_render_params& p, rt::CameraArgs cam,
    rt::FoldArgs fold, bool whole_frame, rt_counters* counters_dev, hipStream_t stream)
{
    if (f.enableMultipleRaysPerPixel && (p.rays_per_pixel_side < 1 || p.rays_per_pixel_side > 32))
        return fail(RT_ERR_INVALID, "rays_per_pixel_side must be in [1, 32]");
    if (f.enableDepthOfField && (p.samples_dof < 1 || p.samples_dof > 1024))
        return fail(RT_ERR_INVALID, "samples_dof must be in [1, 1024]");
    if (f.enableBloomEffect && !whole_frame)
        return fail(RT_ERR_UNSUPPORTED,
            "enableBloomEffect filters the WHOLE frame (render.cpp:326-328): render the rows without it and call "
            "rt_bloom_device once the frame is assembled");
    camera_samples(cam, f, p);
    const uint64_t n_items64 = (uint64_t)cam.n_rows * (uint64_t)cam.vwidth;
    if (n_items64 > (1ull << 30) || (uint64_t)cam.height * (uint64_t)cam.vwidth > 0xffffffffull)
        return fail(RT_ERR_INVALID, "too many samples: rows x width x samples per pixel must be <= 2^30");
    const uint32_t n_items = (uint32_t)n_items64;
    int rc;
    if (cam.spp == 1) {
        rc = enqueue_trace(bvh, sc, f, p, rt::SRC_CAMERA, cam, rt::RaySoA {}, n_items, fold, counters_dev, stream);
    } else {
        HIP_TRY(sc.samples.reserve((size_t)n_items * 3));
        rt::FoldArgs sf;
        memset(&sf, 0, sizeof(sf));
        sf.packed = 1, sf.out = sc.samples.p;
        rc = enqueue_trace(bvh, sc, f, p, rt::SRC_CAMERA, cam, rt::RaySoA {}, n_items, sf, counters_dev, stream);
        if (rc != RT_OK)
            return rc;
        rt::ResolveArgs r;
        r.samples = sc.samples.p, r.n_pixels = (uint32_t)cam.n_rows * (uint32_t)cam.width;
        r.spp = cam.spp, r.ms_n = cam.ms_n, r.dof_n = cam.dof_n;
        r.width = fold.width, r.height = fold.height, r.row0 = fold.row0, r.packed = fold.packed, r.rows = fold.rows;
        r.out = fold.out;
        rt::launch_resolve(r, stream);
        HIP_TRY(hipGetLastError());
    }
    if (rc == RT_OK && f.enableBloomEffect)
        rc = enqueue_bloom(sc, fold.out, cam.width, cam.height, p, stream);
    return rc;
}

} // namespace

int rt_bloom_device(float* rgb_dev, int32_t width, int32_t height, const rt_render_params* params, void* hip_stream)
{
    if (!rgb_dev || !params)
        return fail(RT_ERR_INVALID, "rt_bloom_device: null argument");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30)
        return fail(RT_ERR_INVALID, "rt_bloom_device: bad resolution");
    static std::mutex mu;
    static std::map<void*, std::unique_ptr<Scratch>> scratch; // keyed by stream (no BVH handle involved)
    Scratch* sc;
    {
        std::lock_guard<std::mutex> g(mu);
        auto& s = scratch[hip_stream];
        if (!s)
            s.reset(new Scratch);
        sc = s.get();
    }
    return enqueue_bloom(*sc, rgb_dev, width, height, *params, (hipStream_t)hip_stream);
}

int rt_render_device(const rt_bvh* cbvh, const rt_camera* camera, const rt_features* features,
    const rt_render_params* params, int32_t width, int32_t height, const int32_t* rows, int32_t n_rows,
    int32_t packed, float* rgb_dev, rt_counters* counters_dev, void* hip_stream)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !camera || !rgb_dev)
        return fail(RT_ERR_INVALID, "rt_render_device: null argument");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30)
        return fail(RT_ERR_INVALID, "rt_render_device: bad resolution");
    int rc = check_features(features, bvh);
    if (rc == RT_OK)
        rc = check_params(params);
    if (rc != RT_OK)
        return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    Scratch& sc = bvh->scratch_for(hip_stream);
    rt::CameraArgs cam = camera_args(*camera, width, height);
    if (rows) {
        if (n_rows < 0)
            return fail(RT_ERR_INVALID, "rt_render_device: n_rows < 0");
        for (int32_t i = 0; i < n_rows; i++)
            if (rows[i] < 0 || rows[i] >= height)
                return fail(RT_ERR_INVALID, "rt_render_device: row out of range");
        if (sc.rows_host.size() != (size_t)n_rows || memcmp(sc.rows_host.data(), rows, (size_t)n_rows * 4) != 0) {
            sc.rows_host.assign(rows, rows + n_rows);
            HIP_TRY(sc.rows.reserve((size_t)n_rows));
            if (n_rows)
                HIP_TRY(hipMemcpyAsync(sc.rows.p, sc.rows_host.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice,
                    stream));
        }
        cam.rows = sc.rows.p;
        cam.n_rows = n_rows;
    }
    if (cam.n_rows == 0)
        return RT_OK;
    rt::FoldArgs fold;
    memset(&fold, 0, sizeof(fold));
    fold.width = width, fold.height = height, fold.row0 = cam.row0, fold.packed = packed, fold.rows = cam.rows;
    fold.out = rgb_dev;
    return enqueue_frame(bvh, sc, *features, *params, cam, fold, rows == nullptr && !packed, counters_dev, stream);
}

int rt_frame_rect(const rt_bvh* cbvh, const rt_camera* cameras, int32_t n_cameras, const rt_features* features,
    const rt_render_params* params, int32_t width, int32_t height, int32_t rect[4])
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !cameras || !rect || n_cameras < 1)
        return fail(RT_ERR_INVALID, "rt_frame_rect: null argument");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30)
        return fail(RT_ERR_INVALID, "rt_frame_rect: bad resolution");
    int rc = check_features(features, bvh);
    if (rc == RT_OK)
        rc = check_params(params);
    if (rc != RT_OK)
        return rc;
    rect[0] = 0, rect[1] = 0, rect[2] = width, rect[3] = height;
    // exactly the condition under which the render calls leave tiles unlaunched (enqueue_trace) — the pixels of
    // those tiles are the ones k_fold fills with vec3(0) without looking at anything
    const uint32_t flags = feature_flags(*features);
    if (!(flags & rt::F_ACCEL) || (flags & rt::F_NO_PRUNE))
        return RT_OK;
    int64_t x0 = width, y0 = height, x1 = 0, y1 = 0;
    for (int32_t k = 0; k < n_cameras; k++) {
        rt::CameraArgs cam = camera_args(cameras[k], width, height);
        camera_samples(cam, *features, *params);
        cam.tile_wlog = cam.tile_hlog = 3; // the coarsest tile the library launches: the rectangle is aligned outward to it
        if (!host_launch_rect(bvh, cam))
            return RT_OK; // the whole frame
        const int64_t spp = cam.spp;
        x0 = std::min<int64_t>(x0, ((int64_t)cam.tile_x0 * 8) / spp);
        x1 = std::max<int64_t>(x1, std::min<int64_t>(width, (((int64_t)cam.tile_x0 + cam.tiles_w) * 8 + spp - 1) / spp));
        y0 = std::min<int64_t>(y0, (int64_t)cam.tile_y0 * 8);
        y1 = std::max<int64_t>(y1, std::min<int64_t>(height, ((int64_t)cam.tile_y0 + cam.tiles_h) * 8));
    }
    rect[0] = (int32_t)x0, rect[1] = (int32_t)y0, rect[2] = (int32_t)x1, rect[3] = (int32_t)y1;
    return RT_OK;
}

int rt_render_device_frames(const rt_bvh* cbvh, const rt_camera* cameras, int32_t n_frames,
    const rt_features* features, const rt_render_params* params, int32_t width, int32_t height, const int32_t* rows,
    int32_t n_rows, int32_t packed, float* rgb_dev, uint64_t frame_stride_floats, void* hip_stream)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !cameras || !rgb_dev || !features || !params)
        return fail(RT_ERR_INVALID, "rt_render_device_frames: null argument");
    if (n_frames < 1 || width <= 0 || height <= 0 || (rows && n_rows < 0))
        return fail(RT_ERR_INVALID, "rt_render_device_frames: bad frame count / resolution / row count");
    const int32_t fr = rows ? n_rows : height; // rows of one frame
    const size_t tight = (size_t)(packed ? fr : height) * (size_t)width * 3;
    const size_t frame_floats = frame_stride_floats ? (size_t)frame_stride_floats : tight;
    if (frame_floats < tight)
        return fail(RT_ERR_INVALID, "rt_render_device_frames: frame stride smaller than a frame");
    // One launch holds the frames when the pixel loop is the plain one (one ray per pixel, no bloom, the fused
    // level kernel) and a frame's rows are whole tiles; anything else renders the frames one after the other.
    const rt_features& f = *features;
    const uint32_t th = 1u << g_tile_hlog.load();
    bool one_launch = n_frames >= 2 && n_frames <= rt::kMaxFramesPerLaunch && height < 65536 && fr > 0 && fr % (int32_t)th == 0
        && !f.enableMultipleRaysPerPixel && !f.enableDepthOfField && !f.enableBloomEffect && !f.enableGlossyReflection
        && !f.enableTransparency && !(f.enableRecursive && bvh->scene->host.any_non_opaque) && g_fused.load() != 0
        && g_fused_from.load() == 0 && g_super_log.load() == 0
        && (uint64_t)n_frames * (uint64_t)fr * (uint64_t)width <= (1ull << 30);
    for (int32_t k = 1; one_launch && k < n_frames; k++) // one lens for all frames (the pose is per frame)
        one_launch = cameras[k].half_w == cameras[0].half_w && cameras[k].half_h == cameras[0].half_h;
    if (!one_launch) {
        for (int32_t k = 0; k < n_frames; k++) {
            const int rc = rt_render_device(cbvh, cameras + k, features, params, width, height, rows, n_rows, packed,
                rgb_dev + (size_t)k * frame_floats, nullptr, hip_stream);
            if (rc != RT_OK)
                return rc;
        }
        return RT_OK;
    }
    int rc = check_features(features, bvh);
    if (rc == RT_OK)
        rc = check_params(params);
    if (rc != RT_OK)
        return rc;
    for (int32_t i = 0; rows && i < n_rows; i++)
        if (rows[i] < 0 || rows[i] >= height)
            return fail(RT_ERR_INVALID, "rt_render_device_frames: row out of range");
    hipStream_t stream = (hipStream_t)hip_stream;
    Scratch& sc = bvh->scratch_for(hip_stream);
    // the launch's row list: frame k's rows, entry = row | k << 16
    std::vector<int32_t> all((size_t)n_frames * (size_t)fr);
    for (int32_t k = 0; k < n_frames; k++)
        for (int32_t i = 0; i < fr; i++)
            all[(size_t)k * fr + i] = (rows ? rows[i] : i) | (k << 16);
    if (sc.rows_host != all) {
        sc.rows_host.swap(all);
        HIP_TRY(sc.rows.reserve(sc.rows_host.size()));
        HIP_TRY(hipMemcpyAsync(sc.rows.p, sc.rows_host.data(), sc.rows_host.size() * 4, hipMemcpyHostToDevice, stream));
    }
    rt::CameraArgs cam = camera_args(cameras[0], width, height);
    cam.rows = sc.rows.p, cam.n_rows = n_frames * fr, cam.n_frames = n_frames, cam.frame_rows = fr;
    sc.poses.resize((size_t)n_frames);
    for (int32_t k = 0; k < n_frames; k++) {
        const rt::CameraArgs ck = camera_args(cameras[k], width, height);
        memcpy(sc.poses[(size_t)k].origin, ck.origin, sizeof(ck.origin));
        memcpy(sc.poses[(size_t)k].quat, ck.quat, sizeof(ck.quat));
    }
    rt::FoldArgs fold;
    memset(&fold, 0, sizeof(fold));
    fold.width = width, fold.height = height, fold.row0 = 0, fold.packed = packed, fold.rows = cam.rows;
    fold.n_frames = n_frames, fold.frame_items = (uint32_t)fr * (uint32_t)width, fold.frame_stride = frame_floats;
    fold.out = rgb_dev;
    return enqueue_frame(bvh, sc, *features, *params, cam, fold, false, nullptr, stream);
}

int rt_render(const rt_bvh* cbvh, const rt_camera* camera, const rt_features* features,
    const rt_render_params* params, int32_t width, int32_t height, int32_t row_begin, int32_t row_end, float* rgb,
    rt_counters* counters)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !camera || !rgb)
        return fail(RT_ERR_INVALID, "rt_render: null argument");
    if (width <= 0 || height <= 0 || row_begin < 0 || row_end > height || row_begin > row_end)
        return fail(RT_ERR_INVALID, "rt_render: bad resolution / row range");
    if (row_begin == row_end)
        return RT_OK;
    int rc = check_features(features, bvh);
    if (rc == RT_OK)
        rc = check_params(params);
    if (rc != RT_OK)
        return rc;
    std::lock_guard<std::mutex> g(g_host_call_mu);
    const size_t frame_floats = (size_t)width * height * 3;
    // the handle's device, whatever the calling thread's current device is (a fresh std::thread starts on device 0)
    int caller_device = 0;
    HIP_TRY(hipGetDevice(&caller_device));
    struct RestoreDevice {
        int d, mine;
        ~RestoreDevice()
        {
            if (d != mine)
                (void)hipSetDevice(d);
        }
    } restore_device { caller_device, bvh->scene->device };
    if (caller_device != bvh->scene->device)
        HIP_TRY(hipSetDevice(bvh->scene->device));
    // A frame for the caller's HOST buffer (what renderRayTracing -> Screen asks for): the copy back costs as much
    // as the render, so the frame goes out as bands of rows on two alternating streams and every band is copied
    // back (third stream) while the later bands render. Same launches per band as rt_render_device with a row range:
    // same bits. Not with counters (one counting launch) and not with the whole-frame bloom filter.
    int32_t n_bands = g_host_bands.load();
    if (n_bands == 0) {
        bool heavy = features->enableMultipleRaysPerPixel || features->enableDepthOfField || features->enableGlossyReflection
            || features->enableTransparency;
        if (features->enableSoftShadow)
            for (const rt::LightRec& l : bvh->scene->host.lights)
                heavy = heavy || l.kind != RT_LIGHT_POINT;
        n_bands = heavy ? 1 : ((int64_t)width * height >= (int64_t)6 << 20 ? 8 : 4); // 4K frames: the copy dominates, finer bands
    }
    if (!counters && !features->enableBloomEffect && n_bands > 1 && row_end - row_begin >= 64 * n_bands) {
        for (auto& s_ : bvh->host_stream)
            if (!s_)
                HIP_TRY(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
        for (int32_t b = 0; b < n_bands; b++)
            if (!bvh->host_done[b])
                HIP_TRY(hipEventCreateWithFlags(&bvh->host_done[b], hipEventDisableTiming));
        HIP_TRY(bvh->host_frame.reserve(frame_floats));
        int32_t y_edge[9];
        for (int32_t b = 0; b <= n_bands; b++) // band edges on multiples of 8 rows (tile heights divide 8)
            y_edge[b] = b == n_bands ? row_end : row_begin + (int32_t)(((int64_t)(row_end - row_begin) * b / n_bands) & ~7ll);
        auto drain = [&](int code) {
            for (auto s_ : bvh->host_stream)
                (void)hipStreamSynchronize(s_);
            return code;
        };
        for (int32_t b = 0; b < n_bands; b++) {
            hipStream_t s_ = bvh->host_stream[b & 1];
            Scratch& scb = bvh->scratch_for(s_);
            rt::CameraArgs cam = camera_args(*camera, width, height);
            cam.row0 = y_edge[b], cam.n_rows = y_edge[b + 1] - y_edge[b];
            rt::FoldArgs fold;
            memset(&fold, 0, sizeof(fold));
            fold.width = width, fold.height = height, fold.row0 = y_edge[b], fold.packed = 0, fold.rows = nullptr;
            fold.out = bvh->host_frame.p;
            rc = enqueue_frame(bvh, scb, *features, *params, cam, fold, false, nullptr, s_);
            if (rc != RT_OK)
                return drain(rc);
            if (hipEventRecord(bvh->host_done[b], s_) != hipSuccess)
                return drain(fail(RT_ERR_DEVICE, "rt_render: hipEventRecord"));
        }
        hipStream_t cs = bvh->host_stream[2];
        for (int32_t b = 0; b < n_bands; b++) { // rows [y0, y1) occupy flipped rows [H - y1, H - y0)
            const size_t off = (size_t)(height - y_edge[b + 1]) * width * 3;
            const size_t bytes = (size_t)(y_edge[b + 1] - y_edge[b]) * width * 3 * sizeof(float);
            if (hipStreamWaitEvent(cs, bvh->host_done[b], 0) != hipSuccess
                || hipMemcpyAsync(rgb + off, bvh->host_frame.p + off, bytes, hipMemcpyDeviceToHost, cs) != hipSuccess)
                return drain(fail(RT_ERR_DEVICE, "rt_render: copy of a band of rows back to the host"));
        }
        HIP_TRY(hipStreamSynchronize(cs));
        return RT_OK;
    }
    Scratch& sc = bvh->scratch_for(nullptr);
    HIP_TRY(sc.frame.reserve(frame_floats));
    if (counters)
        HIP_TRY(sc.counters.reserve(1));
    rt::CameraArgs cam = camera_args(*camera, width, height);
    cam.row0 = row_begin, cam.n_rows = row_end - row_begin;
    rt::FoldArgs fold;
    memset(&fold, 0, sizeof(fold));
    fold.width = width, fold.height = height, fold.row0 = row_begin, fold.packed = 0, fold.rows = nullptr;
    fold.out = sc.frame.p;
    rc = enqueue_frame(bvh, sc, *features, *params, cam, fold, row_begin == 0 && row_end == height,
        counters ? sc.counters.p : nullptr, nullptr);
    if (rc != RT_OK)
        return rc;
    // rows [row_begin,row_end) occupy flipped rows [H-row_end, H-row_begin)
    const size_t off = (size_t)(height - row_end) * width * 3;
    HIP_TRY(hipMemcpy(rgb + off, sc.frame.p + off, (size_t)cam.n_rows * width * 3 * sizeof(float),
        hipMemcpyDeviceToHost));
    if (counters)
        HIP_TRY(hipMemcpy(counters, sc.counters.p, sizeof(rt_counters), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render_frames(const rt_bvh* cbvh, const rt_camera* cameras, int32_t n_frames, const rt_features* features,
    const rt_render_params* params, int32_t width, int32_t height, float* rgb)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !cameras || !rgb || !features || !params)
        return fail(RT_ERR_INVALID, "rt_render_frames: null argument");
    if (n_frames < 1 || width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30)
        return fail(RT_ERR_INVALID, "rt_render_frames: bad frame count / resolution");
    std::lock_guard<std::mutex> g(g_host_call_mu);
    Scratch& sc = bvh->scratch_for(nullptr);
    const size_t frame_floats = (size_t)width * height * 3;
    // groups of up to 8 frames: one launch each (or one launch per frame when the frames
    // cannot share one, rt_render_device_frames), one copy back per group
    constexpr int32_t kGroup = 8; // whole frames: eight fill the chip, and the staging buffer stays at 8 frames
    for (int32_t k0 = 0; k0 < n_frames; k0 += kGroup) {
        const int32_t n = std::min<int32_t>(kGroup, n_frames - k0);
        HIP_TRY(sc.frame.reserve(frame_floats * (size_t)n));
        const int rc = rt_render_device_frames(cbvh, cameras + k0, n, features, params, width, height, nullptr, height, 0,
            sc.frame.p, 0, nullptr);
        if (rc != RT_OK)
            return rc;
        HIP_TRY(hipMemcpy(rgb + (size_t)k0 * frame_floats, sc.frame.p, frame_floats * (size_t)n * sizeof(float),
            hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_render_sharded(rt_bvh* const* bvhs, int32_t n_shards, const rt_camera* camera, const rt_features* features,
    const rt_render_params* params, int32_t width, int32_t height, int32_t strip_rows, float* rgb, int32_t rgb_on_device)
{
    if (!bvhs || n_shards < 1 || !camera || !rgb)
        return fail(RT_ERR_INVALID, "rt_render_sharded: null argument / no shard");
    if (width <= 0 || height <= 0 || strip_rows <= 0 || (int64_t)width * height > (int64_t)1 << 30)
        return fail(RT_ERR_INVALID, "rt_render_sharded: bad resolution / strip height");
    for (int32_t s = 0; s < n_shards; s++)
        if (!bvhs[s])
            return fail(RT_ERR_INVALID, "rt_render_sharded: null bvh");
    int rc = check_features(features, bvhs[0]);
    if (rc == RT_OK)
        rc = check_params(params);
    if (rc != RT_OK)
        return rc;
    std::lock_guard<std::mutex> g(g_host_call_mu);
    int caller_device = 0;
    HIP_TRY(hipGetDevice(&caller_device));
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore { caller_device };
    const int dev0 = bvhs[0]->scene->device;
    const int32_t n_strips = (height + strip_rows - 1) / strip_rows;
    const int32_t rows_max = ((n_strips + n_shards - 1) / n_shards) * strip_rows;
    const size_t shard_floats = (size_t)rows_max * width * 3, frame_floats = (size_t)width * height * 3;
    // the strips are rendered without the whole-frame filter; it runs on the assembled frame
    rt_features f_strips = *features;
    f_strips.enableBloomEffect = 0;

    rt_bvh* b0 = bvhs[0];
    HIP_TRY(hipSetDevice(dev0));
    if (!b0->shard_stream)
        HIP_TRY(hipStreamCreateWithFlags(&b0->shard_stream, hipStreamNonBlocking));
    HIP_TRY(b0->shard_gathered.reserve(shard_floats * (size_t)n_shards));
    HIP_TRY(b0->shard_frame.reserve(frame_floats));
    // ---- every shard: its strips into its packed buffer, then a peer copy into shard 0's gathered buffer ----
    // an error return must not leave shards in flight behind it: wait for what was enqueued so far
    int32_t enqueued = 0;
    auto drain = [&](int code) {
        for (int32_t q = 0; q < enqueued; q++)
            if (bvhs[q]->shard_stream) {
                (void)hipSetDevice(bvhs[q]->scene->device);
                (void)hipStreamSynchronize(bvhs[q]->shard_stream);
            }
        return code;
    };
#define SHARD_TRY(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return drain(fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)));       \
    } while (0)
    for (int32_t s = 0; s < n_shards; s++) {
        rt_bvh* b = bvhs[s];
        const int dev = b->scene->device;
        SHARD_TRY(hipSetDevice(dev));
        if (!b->shard_stream) {
            SHARD_TRY(hipStreamCreateWithFlags(&b->shard_stream, hipStreamNonBlocking));
            if (dev != dev0) {
                // first use of this shard: peer access dev -> dev0 (the copy below runs on dev's stream) and
                // dev0 -> dev (shard 0's stream waits on this device's event); ONCE per shard handle, an
                // already-enabled pair is not an error
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev, dev0) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(dev0, 0);
                can = 0;
                if (hipSetDevice(dev0) == hipSuccess && hipDeviceCanAccessPeer(&can, dev0, dev) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(dev, 0);
                (void)hipGetLastError();
                SHARD_TRY(hipSetDevice(dev));
            }
        }
        if (!b->shard_done)
            SHARD_TRY(hipEventCreateWithFlags(&b->shard_done, hipEventDisableTiming));
        SHARD_TRY(b->shard_packed.reserve(shard_floats));
        std::vector<int32_t> rows;
        for (int32_t k = s; k < n_strips; k += n_shards)
            for (int32_t y = k * strip_rows; y < std::min((k + 1) * strip_rows, height); y++)
                rows.push_back(y);
        enqueued = s + 1;
        if (!rows.empty()) {
            rc = rt_render_device(b, camera, &f_strips, params, width, height, rows.data(), (int32_t)rows.size(), 1,
                b->shard_packed.p, nullptr, b->shard_stream);
            if (rc != RT_OK)
                return drain(rc);
            SHARD_TRY(hipMemcpyPeerAsync(b0->shard_gathered.p + (size_t)s * shard_floats, dev0, b->shard_packed.p, dev,
                rows.size() * (size_t)width * 3 * sizeof(float), b->shard_stream));
        }
        SHARD_TRY(hipEventRecord(b->shard_done, b->shard_stream));
    }
    // ---- shard 0's device: wait for every shard's copy, de-interleave, filter, hand the frame over ----------
    SHARD_TRY(hipSetDevice(dev0));
    for (int32_t s = 1; s < n_shards; s++)
        SHARD_TRY(hipStreamWaitEvent(b0->shard_stream, bvhs[s]->shard_done, 0));
    float* frame = rgb_on_device ? rgb : b0->shard_frame.p;
    rt::launch_deinterleave(b0->shard_gathered.p, frame, width, height, strip_rows, n_shards, 1, 1, b0->shard_stream);
    SHARD_TRY(hipGetLastError());
    if (features->enableBloomEffect) {
        rc = enqueue_bloom(b0->scratch_for(b0->shard_stream), frame, width, height, *params, b0->shard_stream);
        if (rc != RT_OK)
            return drain(rc);
    }
    if (!rgb_on_device)
        SHARD_TRY(hipMemcpyAsync(rgb, frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, b0->shard_stream));
    SHARD_TRY(hipStreamSynchronize(b0->shard_stream));
    return RT_OK;
#undef SHARD_TRY
}

int rt_render_hits(const rt_bvh* cbvh, const rt_camera* camera, const rt_features* features, int32_t width,
    int32_t height, uint32_t* t_bits, int32_t* prim)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !camera || !features)
        return fail(RT_ERR_INVALID, "rt_render_hits: null argument");
    if (width <= 0 || height <= 0)
        return fail(RT_ERR_INVALID, "rt_render_hits: bad resolution");
    std::lock_guard<std::mutex> g(g_host_call_mu);
    Scratch& sc = bvh->scratch_for(nullptr);
    const size_t n = (size_t)width * height;
    HIP_TRY(sc.t_bits.reserve(n));
    HIP_TRY(sc.prim.reserve(n));
    rt::IntersectArgs a;
    memset(&a, 0, sizeof(a));
    a.sv = bvh->view();
    a.flags = feature_flags(*features);
    a.source = rt::SRC_CAMERA;
    a.cam = camera_args(*camera, width, height);
    a.t_bits = sc.t_bits.p;
    a.prim = sc.prim.p;
    rt::launch_intersect(a, false, nullptr);
    HIP_TRY(hipGetLastError());
    if (t_bits)
        HIP_TRY(hipMemcpy(t_bits, sc.t_bits.p, n * 4, hipMemcpyDeviceToHost));
    if (prim)
        HIP_TRY(hipMemcpy(prim, sc.prim.p, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_trace_batch(const rt_bvh* cbvh, const rt_ray* rays, uint32_t n_rays, const uint32_t* pixel_ids,
    const rt_features* features, const rt_render_params* params, float* rgb, rt_counters* counters)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || (n_rays && (!rays || !rgb)))
        return fail(RT_ERR_INVALID, "rt_trace_batch: null argument");
    int rc = check_features(features, bvh);
    if (rc == RT_OK)
        rc = check_params(params);
    if (rc != RT_OK || n_rays == 0)
        return rc;
    std::lock_guard<std::mutex> g(g_host_call_mu);
    Scratch& sc = bvh->scratch_for(nullptr);
    rt::RaySoA in;
    rc = upload_rays(sc, rays, n_rays, pixel_ids, nullptr, in);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(sc.frame.reserve((size_t)n_rays * 3));
    if (counters)
        HIP_TRY(sc.counters.reserve(1));
    rt::FoldArgs fold;
    memset(&fold, 0, sizeof(fold));
    fold.out = sc.frame.p; // width == 0 → out[item]
    rt::CameraArgs cam;
    memset(&cam, 0, sizeof(cam));
    rc = enqueue_trace(bvh, sc, *features, *params, rt::SRC_RAYS, cam, in, n_rays, fold,
        counters ? sc.counters.p : nullptr, nullptr);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(hipMemcpy(rgb, sc.frame.p, (size_t)n_rays * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (counters)
        HIP_TRY(hipMemcpy(counters, sc.counters.p, sizeof(rt_counters), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_intersect_batch(const rt_bvh* cbvh, rt_ray* rays, uint32_t n_rays, const rt_features* features, rt_hit* hits,
    rt_counters* counters)
{
    rt_bvh* bvh = const_cast<rt_bvh*>(cbvh);
    if (!bvh || !features || (n_rays && !rays))
        return fail(RT_ERR_INVALID, "rt_intersect_batch: null argument");
    if (n_rays == 0)
        return RT_OK;
    std::lock_guard<std::mutex> g(g_host_call_mu);
    Scratch& sc = bvh->scratch_for(nullptr);
    rt::RaySoA in;
    int rc = upload_rays(sc, rays, n_rays, nullptr, nullptr, in);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(sc.hits.reserve(n_rays));
    HIP_TRY(sc.frame.reserve(n_rays));
    if (counters) {
        HIP_TRY(sc.counters.reserve(1));
        HIP_TRY(hipMemset(sc.counters.p, 0, sizeof(rt_counters)));
    }
    rt::IntersectArgs a;
    memset(&a, 0, sizeof(a));
    a.sv = bvh->view();
    a.flags = feature_flags(*features);
    a.source = rt::SRC_RAYS;
    a.in = in;
    a.n_rays = n_rays;
    a.hits = sc.hits.p;
    a.t_out = sc.frame.p;
    a.counters = counters ? sc.counters.p : nullptr;
    rt::launch_intersect(a, counters != nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    std::vector<float> tt(n_rays);
    HIP_TRY(hipMemcpy(tt.data(), sc.frame.p, (size_t)n_rays * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_rays; i++)
        rays[i].t = tt[i];
    if (hits)
        HIP_TRY(hipMemcpy(hits, sc.hits.p, (size_t)n_rays * sizeof(rt_hit), hipMemcpyDeviceToHost));
    if (counters)
        HIP_TRY(hipMemcpy(counters, sc.counters.p, sizeof(rt_counters), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_tile_order(int32_t width, int32_t n_rows, int32_t n_frames, int32_t tile_wlog, int32_t tile_hlog,
    const int32_t rect[4], uint32_t* order_out, uint64_t cap, uint32_t info[20])
{
    if (!order_out || !info || width <= 0 || n_rows <= 0 || n_frames < 1 || n_frames > rt::kMaxFramesPerLaunch || tile_wlog < 0
        || tile_wlog > 3 || tile_hlog < 0 || tile_hlog > 3 || n_rows % (n_frames << tile_hlog) != 0)
        return fail(RT_ERR_INVALID, "rt_debug_tile_order: bad argument");
    Scratch sc;
    rt::CameraArgs cam;
    memset(&cam, 0, sizeof(cam));
    cam.width = width, cam.vwidth = width, cam.spp = 1, cam.height = n_rows / n_frames, cam.n_rows = n_rows;
    cam.tile_wlog = (uint32_t)tile_wlog, cam.tile_hlog = (uint32_t)tile_hlog;
    cam.n_frames = n_frames, cam.frame_rows = n_rows / n_frames;
    std::vector<int32_t> rows_stub(1, 0);
    if (n_frames > 1)
        cam.rows = rows_stub.data(); // only tested for being a row list
    if (rect)
        cam.tile_x0 = (uint32_t)rect[0], cam.tile_y0 = (uint32_t)rect[1], cam.tiles_w = (uint32_t)rect[2], cam.tiles_h = (uint32_t)rect[3];
    bool zeroed = false;
    const int rc = tile_order_for(sc, cam, nullptr, nullptr, 0, &zeroed);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    const uint32_t tw = 1u << tile_wlog, th = 1u << tile_hlog;
    const uint64_t gw = cam.tiles_w ? cam.tiles_w : ((uint64_t)width + tw - 1) / tw;
    const uint64_t gh = cam.tiles_w ? cam.tiles_h : ((uint64_t)n_rows + th - 1) / th;
    if (!cam.order || gw * gh > cap)
        return fail(RT_ERR_INVALID, "rt_debug_tile_order: no order built, or the output is too small");
    HIP_TRY(hipMemcpy(order_out, cam.order, gw * gh * sizeof(uint32_t), hipMemcpyDeviceToHost));
    info[0] = (uint32_t)gw, info[1] = (uint32_t)gh, info[2] = cam.xcd_order, info[3] = cam.launch_waves;
    for (int k = 0; k < 8; k++)
        info[4 + k] = cam.xcd_base[k], info[12 + k] = cam.xcd_count[k];
    return RT_OK;
}

int rt_debug_div_check(uint32_t seed, uint32_t mode, uint64_t n_samples, uint64_t* mismatches,
    uint64_t* fast_path_samples)
{
    DevBuf<unsigned long long> out;
    HIP_TRY(out.reserve(4));
    HIP_TRY(hipMemset(out.p, 0, 4 * sizeof(unsigned long long)));
    const uint32_t per_thread = 256;
    const uint64_t threads = (n_samples + per_thread - 1) / per_thread;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, (threads + 255) / 256);
    rt::launch_div_check(seed, mode, blocks, per_thread, out.p, nullptr);
    HIP_TRY(hipGetLastError());
    unsigned long long h[4];
    HIP_TRY(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
    if (mismatches)
        *mismatches = h[0];
    if (fast_path_samples)
        *fast_path_samples = h[1];
    if (h[0])
        g_error = "div_by_ray != n / d, first at n bits " + std::to_string((uint32_t)h[2]) + ", d bits "
            + std::to_string((uint32_t)h[3]);
    return RT_OK;
}

// Diagnostic (library built with -DRT_STAMPS): zero / read the 24 phase-clock slots.
// With -DRT_TIMELINE the buffer holds 3 words (start, end, hardware id) for each of
// up to 2^17 waves of the last k_trace_level launch instead.
#ifdef RT_TIMELINE
constexpr size_t kStampWords = 3u << 17;
#else
constexpr size_t kStampWords = 64; // 0..23 exact-topology engine, 32..51 wide engine (rt_wide.h)
#endif
int rt_debug_stamps(int reset, unsigned long long* out)
{
    if (!g_stamps)
        HIP_TRY(hipMalloc((void**)&g_stamps, kStampWords * sizeof(unsigned long long)));
    HIP_TRY(hipDeviceSynchronize());
    if (out)
        HIP_TRY(hipMemcpy(out, g_stamps, kStampWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset)
        HIP_TRY(hipMemset(g_stamps, 0, kStampWords * sizeof(unsigned long long)));
    return RT_OK;
}

int rt_profile_enable(rt_bvh* bvh, int enable)
{
    if (!bvh)
        return fail(RT_ERR_INVALID, "rt_profile_enable: null bvh");
    bvh->profile = enable != 0;
    return RT_OK;
}

int rt_profile_read(rt_bvh* bvh, void* hip_stream, double* trace_kernel_ms, uint64_t* trace_kernel_launches,
    int reset)
{
    if (!bvh)
        return fail(RT_ERR_INVALID, "rt_profile_read: null bvh");
    Scratch& sc = bvh->scratch_for(hip_stream);
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    double ms = 0.0;
    for (size_t i = 0; i < sc.events_used; i++) {
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, sc.events[i].first, sc.events[i].second));
        ms += t;
    }
    if (trace_kernel_ms)
        *trace_kernel_ms = ms;
    if (trace_kernel_launches)
        *trace_kernel_launches = sc.events_used;
    if (reset)
        sc.events_used = 0;
    return RT_OK;
}

int rt_profile_read_launches(rt_bvh* bvh, void* hip_stream, double* launch_ms, uint64_t cap, uint64_t* n_launches,
    int reset)
{
    if (!bvh || (cap && !launch_ms))
        return fail(RT_ERR_INVALID, "rt_profile_read_launches: null argument");
    Scratch& sc = bvh->scratch_for(hip_stream);
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    for (size_t i = 0; i < sc.events_used && i < cap; i++) {
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, sc.events[i].first, sc.events[i].second));
        launch_ms[i] = t;
    }
    if (n_launches)
        *n_launches = sc.events_used;
    if (reset)
        sc.events_used = 0;
    return RT_OK;
}

int rt_deinterleave_strips(const float* gathered_dev, float* frame_dev, int32_t width, int32_t height,
    int32_t strip_rows, int32_t n_ranks, void* hip_stream)
{
    if (!gathered_dev || !frame_dev || width <= 0 || height <= 0 || strip_rows <= 0 || n_ranks <= 0)
        return fail(RT_ERR_INVALID, "rt_deinterleave_strips: bad argument");
    rt::launch_deinterleave(gathered_dev, frame_dev, width, height, strip_rows, n_ranks, 1, 1, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_deinterleave_frames(const float* gathered_dev, float* frames_dev, int32_t width, int32_t height,
    int32_t strip_rows, int32_t n_ranks, int32_t n_frames, int32_t batch_frames, void* hip_stream)
{
    if (!gathered_dev || !frames_dev || width <= 0 || height <= 0 || strip_rows <= 0 || n_ranks <= 0 || n_frames <= 0
        || batch_frames < n_frames)
        return fail(RT_ERR_INVALID, "rt_deinterleave_frames: bad argument");
    if ((uint64_t)width * 3 * (uint64_t)height * (uint64_t)n_frames / 256 >= 0x7fffffffull)
        return fail(RT_ERR_INVALID, "rt_deinterleave_frames: too many pixels for one launch");
    rt::launch_deinterleave(gathered_dev, frames_dev, width, height, strip_rows, n_ranks, n_frames, batch_frames,
        (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

uint32_t rt_rand31(uint32_t seed, uint32_t pixel, uint32_t level, uint32_t light, uint32_t sample, uint32_t dim)
{
    auto pcg = [](uint32_t v) {
        const uint32_t state = v * 747796405u + 2891336453u;
        const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
        return (word >> 22u) ^ word;
    };
    uint32_t h = pcg(seed ^ 0x9E3779B9u);
    h = pcg(h + pixel);
    h = pcg(h ^ ((level << 26) | (light << 18) | (sample << 1) | dim));
    return h & 0x7fffffffu;
}

} // extern "C"
