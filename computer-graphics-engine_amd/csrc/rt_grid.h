// rt_grid.h — empty-space skipping: how far along a ray can a candidate be at all?
//
// The reference's result for a ray is decided by its CANDIDATES (rt_device.h: triangles that pass the exact test
// with 0 <= t <= ray.t). A candidate's point lies within rho of its triangle (rt_device.h "Pruning"), so it lies
// in a voxel that the host marked for that triangle (host_wide.cpp build_grid: the triangle's bounds grown by a
// quarter cell + rho). Walking the ray through the voxel grid therefore bounds where candidates can be: none
// beyond the last marked voxel the ray passes, none at all when it passes no marked voxel. Lowering a ray's limit
// to that bound (or dropping the ray) cannot remove a candidate — the traversal finds the same candidates, the
// same minimum key, the same bits — but it takes the far part of the tree off every ray that ends in free space:
// an unoccluded shadow ray stops at the skin of the surface it leaves instead of crossing the scene (84 box tests
// on the bench scene), a primary or reflection ray that misses everything is not traversed at all.
//
// The walk is the usual 3-D DDA, one voxel per step, the step's plane crossing computed afresh from the voxel
// index (no accumulated error). In float arithmetic it can differ from the exact walk only where two crossings
// nearly coincide (the ray passes within a few ulps of a voxel edge); the voxel it then leaves out is touched by
// the ray for a length of a few ulps next to voxels it does visit, and the quarter cell by which the host grew the
// marked bounds covers that a thousand times over; the limit returned is moved out by 1/64 of a cell as well.
#pragma once
#include "rt_device.h"

namespace rt {

// The limit to use for `ray` instead of t0 (<= t0), or a negative value when the ray cannot have a candidate.
// Only for rays with three non-zero finite direction components (RayInv::all_ok) and an origin the grid was built
// for; others keep t0. `steps` (diagnostic) counts voxels visited.
RT_DEV float grid_limit(const SceneView& sv, const RayL& r, const RayInv& inv, float t0)
{
    const int32_t n = (int32_t)sv.grid_n;
    const float omax = fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z));
    if (n == 0 || !inv.all_ok() || !(omax <= sv.grid_smax) || !(t0 > 0.0f))
        return t0;
    const float o[3] = { r.o.x, r.o.y, r.o.z }, d[3] = { r.d.x, r.d.y, r.d.z };
    // the part of [0, t0] inside the grid box, taken a little too large (clamped voxel indices make up for it)
    float ta = 0.0f, tb = t0;
    float oy[3]; // o * (1 / d) per axis
#pragma unroll
    for (int a = 0; a < 3; a++) {
        oy[a] = o[a] * inv.y[a];
        const float hi = sv.grid_lo[a] + sv.grid_cell[a] * (float)n;
        const float q0 = __builtin_fmaf(sv.grid_lo[a], inv.y[a], -oy[a]), q1 = __builtin_fmaf(hi, inv.y[a], -oy[a]);
        ta = fmaxf(ta, fminf(q0, q1));
        tb = fminf(tb, fmaxf(q0, q1));
    }
    // how far 1/64 of the smallest cell is in t, for the axis along which the ray moves slowest
    const float cmin = fminf(fminf(sv.grid_cell[0], sv.grid_cell[1]), sv.grid_cell[2]);
    const float ymax = fmaxf(fmaxf(fabsf(inv.y[0]), fabsf(inv.y[1])), fabsf(inv.y[2]));
    const float dt = cmin * 0.015625f * ymax;
    if (!(ta <= tb + dt))
        return -1.0f; // the ray misses the grid box (all triangles lie inside it, margins included)
    int32_t iv[3], st[3];
    float edge[3]; // index offset of the plane through which the ray leaves a voxel
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float p = __builtin_fmaf(d[a], ta, o[a]);
        int32_t i = (int32_t)floorf((p - sv.grid_lo[a]) * sv.grid_inv_cell[a]);
        i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
        iv[a] = i;
        st[a] = d[a] > 0.0f ? 1 : -1;
        edge[a] = d[a] > 0.0f ? 1.0f : 0.0f;
    }
    float t_last = -1.0f;
    const uint32_t* bits = sv.grid_bits;
    // one voxel per step; the occupancy word of a voxel is used one step later than it is requested, so that the
    // load's latency overlaps the next step's arithmetic (the walk itself does not depend on the bits)
    uint32_t word = 0u, bit = 0u;
    float t_exit_prev = 0.0f;
    bool pending = false;
    for (int32_t guard = 0; guard < 3 * n + 3; guard++) {
        const uint32_t idx = ((uint32_t)iv[2] * (uint32_t)n + (uint32_t)iv[1]) * (uint32_t)n + (uint32_t)iv[0];
        const uint32_t w_now = bits[idx >> 5];
        float tq[3];
#pragma unroll
        for (int a = 0; a < 3; a++)
            tq[a] = __builtin_fmaf(__builtin_fmaf((float)iv[a] + edge[a], sv.grid_cell[a], sv.grid_lo[a]), inv.y[a], -oy[a]);
        const float tn = fminf(fminf(tq[0], tq[1]), tq[2]);
        if (pending && ((word >> bit) & 1u))
            t_last = t_exit_prev;
        word = w_now, bit = idx & 31u, t_exit_prev = tn, pending = true;
        if (!(tn < tb))
            break;
        // leave through the nearest plane (a tie: any of them — the voxels left out are touched in a point)
        if (tq[0] <= tq[1] && tq[0] <= tq[2])
            iv[0] += st[0];
        else if (tq[1] <= tq[2])
            iv[1] += st[1];
        else
            iv[2] += st[2];
        if ((uint32_t)iv[0] >= (uint32_t)n || (uint32_t)iv[1] >= (uint32_t)n || (uint32_t)iv[2] >= (uint32_t)n)
            break;
    }
    if (pending && ((word >> bit) & 1u))
        t_last = t_exit_prev;
    if (t_last < 0.0f)
        return -1.0f;
    const float lim = t_last + dt;
    return lim < t0 ? lim : t0;
}

} // namespace rt
