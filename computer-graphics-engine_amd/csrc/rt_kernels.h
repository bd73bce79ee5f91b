// rt_kernels.h — launch interface between the C ABI (rt_capi.hip) and the HIP
// kernels (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rt_whitted.h"
#include "rt_layout.h"

namespace rt {

enum : int32_t { SRC_CAMERA = 0, SRC_QUEUE = 1, SRC_RAYS = 2 };

// Ray queue / user ray batch in SoA form: 6 float arrays of `cap` entries
// (ox, oy, oz, dx, dy, dz) at ray[k*cap + i], optional t, item and pixel ids.
struct RaySoA {
    float* ray;
    float* t;        // NULL → FLT_MAX
    uint32_t* item;  // NULL → i
    uint32_t* pixel; // NULL → i
    uint32_t cap;
};

struct CameraArgs {
    float origin[3], quat[4], half_w, half_h;
    int32_t width, height;
    const int32_t* rows; // device list of image rows, NULL → row0 + local row
    int32_t row0, n_rows;
    uint32_t tile_wlog, tile_hlog; // primary tile = 2^wlog x 2^hlog pixels per wave (<= 64)
    uint32_t super_log;            // > 0: super-tiles of 2^s x 2^s tiles dealt to the XCDs (tile_pixel)
    const uint32_t* order;         // NULL, or launch position -> tile index (heavy-first dispatch order)
    // XCD-aware order (xcd_order != 0): workgroup b runs on XCD b % 8, and `order` holds eight lists — list x = the
    // tiles of the screen blocks dealt to XCD x (tile_xcd), centre first — so that every XCD's L2 serves the part of
    // the scene behind its own blocks; launch position b = the (b / 8)-th tile of list b % 8
    uint32_t xcd_order, launch_waves; // launch_waves: 8 x the longest list
    uint32_t xcd_base[8], xcd_count[8];
    // launched sub-grid of tiles (host_launch_rect): tiles [tile_x0, tile_x0 + tiles_w) x [tile_y0, tile_y0 + tiles_h);
    // tiles_w == 0 means the whole grid
    uint32_t tile_x0, tile_y0, tiles_w, tiles_h;
    // samples per pixel (src/render.cpp:295-314): ms_n x ms_n jittered rays, then dof_n lens rays;
    // the launch sees a VIRTUAL image of vwidth = width * spp columns, column x * spp + s = sample s of pixel x
    int32_t vwidth, spp, ms_n, dof_n;
    float look_at[3], focus, blur;
    uint32_t seed;
    // Several frames in one launch (rt_render_device_frames): the row list holds frame f's rows one frame after
    // the other, entry = row | f << 16; frame_rows entries per frame (a multiple of the tile height, so a tile
    // never straddles two frames); frame f's origin / rotation are poses[f]. n_frames <= 1: a plain launch.
    int32_t n_frames, frame_rows;
};
constexpr int32_t kMaxFramesPerLaunch = 32; // a rank's small share of a frame at 8 ranks: 32 shares = 4 frames' worth of work per launch
struct CameraPose {
    float origin[3], quat[4];
};

// One recursion level of the flattened Whitted recursion (k_trace_level).
struct LevelArgs {
    SceneView sv;
    uint32_t flags;
    int32_t level, depth;
    int32_t parallelogram_n, segment_n;
    uint32_t seed;
    int32_t source;
    CameraArgs cam;
    CameraPose poses[kMaxFramesPerLaunch]; // cam.n_frames > 1 only
    RaySoA in;
    const uint32_t* in_count; // device count for SRC_QUEUE
    uint32_t in_count_imm;    // count for SRC_RAYS
    uint32_t min_waves;       // sparse levels: fewer rays per wave until this many waves are busy
    uint32_t fused;           // 1: the wave traces its own reflection rays, all levels in this launch
    size_t level_stride;      // floats between the `direct` planes of consecutive levels
    uint32_t merge_next;      // 1: trace a level's first shadow rays together with the next level's reflection rays
    uint32_t tile_cull;       // 1: primary tiles that cannot reach the scene bounds skip the traversal
    uint32_t skip_unlit;      // 1: no shadow ray where the light's colour at the hit is exactly zero (shade_hit)
    uint32_t tile_cull_count; // counting pass: 1 = count (and still trace) the primary rays tile_cull would skip
    uint32_t variant;         // which k_trace_level variant renders the launch (rt_kernels.hip: 0 everything, 1 lean,
                              //    2 lean + hit attributes)
    RaySoA out;
    uint32_t* out_count;
    float* direct;   // this level's direct-light colours, n_items * 3
    uint32_t* state; // per item: bits 0..7 = levels that hit, bit 8 = chain ended on a miss
    uint32_t n_items;
    rt_counters* counters; // NULL unless counting build
    // k_trace_general (transparency / glossy recursion): per-lane frame stacks, the extras' parameters
    float* frames;
    int32_t max_frames; // depth + kExtraFrames + 1
    int32_t glossy_cap, rays_per_reflection;
    float alpha_modifier;
};

constexpr int32_t kExtraFrames = 10; // frames past `depth` for the un-depth-limited call of render.cpp:122 (oracle: same)
constexpr int32_t kFrameWords = 40;

struct FoldArgs {
    const float* direct; // (depth + 1) level slices of n_items * 3
    const uint32_t* state;
    uint32_t n_items;
    size_t level_stride; // floats between level slices
    // output mapping: batch (width == 0): out[item]; frame: item = local_row*W + x →
    // packed: same index; flipped: (H-1-y)*W + x with y = rows ? rows[local_row] : row0 + local_row
    int32_t width, height, row0, packed;
    const int32_t* rows;
    int32_t n_frames; // > 1: rows entries are row | frame << 16, the frames follow each other in `out`
    uint32_t frame_items;  // items (rows x width) of one frame of the launch
    size_t frame_stride;   // floats between the frames in `out`
    float* out;
    // launched sub-grid (CameraArgs::tiles_w != 0): items of tiles outside it were never traced = vec3(0)
    uint32_t vwidth, tile_wlog, tile_hlog, tile_x0, tile_y0, tiles_w, tiles_h;
};

// Per-pixel combination of the samples (renderRayTracing, src/render.cpp:295-322) — k_resolve
struct ResolveArgs {
    const float* samples; // n_pixels * spp colours, sample-major inside a pixel
    uint32_t n_pixels;
    int32_t spp, ms_n, dof_n;
    int32_t width, height, row0, packed;
    const int32_t* rows;
    float* out;
};

// renderBloomFilter (src/render.cpp:157-196) — k_bloom
struct BloomArgs {
    const float* src; // copy of the frame before the filter
    float* frame;
    int32_t width, height;
    float w[3][3];    // weightsGaussian(1.0f), computed on the host
    float scalar, threshold;
    int32_t debug_option;
};

struct IntersectArgs {
    SceneView sv;
    uint32_t flags;
    int32_t source; // SRC_CAMERA (primary rays of a frame) or SRC_RAYS
    CameraArgs cam;
    RaySoA in;
    uint32_t n_rays;
    // outputs (any may be NULL); for SRC_CAMERA indices are row-flipped frame positions
    uint32_t* t_bits;
    int32_t* prim;
    rt_hit* hits;
    float* t_out; // updated ray.t per ray (SRC_RAYS)
    rt_counters* counters;
};

// Dispatch order of the launched tile grid (k_tile_ring_*): rings around (cx, cy), measured in pixels
// XCD of the tile at (tx, ty) of the tile grid (ty within its frame): screen blocks of 2^bxlog x 2^bylog tiles go
// to one XCD; shared by host (list lengths) and device
inline __host__ __device__ uint32_t tile_xcd(uint32_t tx, uint32_t ty, uint32_t bxlog, uint32_t bylog)
{
    return ((tx >> bxlog) + 3u * (ty >> bylog)) & 7u;
}
struct TileOrderArgs {
    uint32_t gw, n_tiles; // launched grid width in tiles, tiles in it
    uint32_t xcd_order, xcd_bxlog, xcd_bylog, tile_x0, tile_y0; // xcd_order: sort key = (XCD of the tile, frame, ring)
    uint32_t n_frames;                                         // frames in the launch (tile rows / period)
    uint32_t frame_major; // 1: an XCD's list goes frame after frame (each centre first); 0: centre first over all frames
    int64_t tw, th, cx, cy, period, step;
    uint32_t n_rings;
    uint32_t* rings; // n_rings counters (scratch)
    uint32_t* order; // out: launch position -> tile index
    // one-launch form (k_tile_order_lists: one workgroup per list, counters in LDS)
    uint32_t one_launch;   // keys per list fit the LDS
    uint32_t list_base[8]; // first place of list x in `order`
    uint32_t* zero;        // NULL, or n_zero words the launch clears on its way (the level counters of the trace launch)
    uint32_t n_zero;
};
constexpr uint32_t kTileOrderLdsKeys = 12288; // one-launch form: sort keys per list that fit the LDS (48 KB)

extern uint32_t g_lds_pad;
#if defined(RT_DIAG) && RT_DIAG == 2
void set_diag_leaf(const float4* p);
#endif
uint32_t trace_level_groups(const LevelArgs& a); // workgroups launch_trace_level will start
void launch_trace_level(const LevelArgs& a, bool count, hipStream_t s);
uint32_t trace_general_waves(const LevelArgs& a); // persistent waves of launch_trace_general (frame stacks are per wave slot)
void launch_trace_general(const LevelArgs& a, bool count, hipStream_t s);
void launch_tile_order(const TileOrderArgs& a, hipStream_t s);
void launch_fold(const FoldArgs& a, hipStream_t s);
void launch_resolve(const ResolveArgs& a, hipStream_t s);
void launch_bloom(const BloomArgs& a, hipStream_t s);
void launch_intersect(const IntersectArgs& a, bool count, hipStream_t s);
void launch_div_check(uint32_t seed, uint32_t mode, uint32_t blocks, uint32_t per_thread, unsigned long long* out,
    hipStream_t s);
void launch_deinterleave(const float* gathered, float* frames, int32_t width, int32_t height, int32_t strip_rows,
    int32_t n_ranks, int32_t n_frames, int32_t batch, hipStream_t s);

} // namespace rt
