"""Randomised parity campaign: random triangle soups / tessellated surfaces / coplanar sheets / spheres,
random materials, textures, lights, feature flags, depths, cameras and odd resolutions — the HIP path
against the CPU oracle: hit records bit-identical, colours within the stated tolerance with identical
NaN positions, ray counts equal. FUZZ_CASES (default 60) and FUZZ_SEED (default 1) size the run;
the round's long campaign is recorded in profiles/r01_notes.md."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-5, 1e-6   # tests/test_gpu_parity.py


def material(rng):
    ks = rng.uniform(0, 0.8, 3) if rng.random() < 0.6 else np.zeros(3)
    tr = 1.0 if rng.random() < 0.6 else float(rng.choice([0.0, 0.25, 0.5, 0.9]))
    return abi.MaterialData(tuple(rng.uniform(0.1, 1, 3)), tuple(ks), float(rng.choice([0.0, 1.0, 8.0, 60.0, 250.0])), tr, -1)


def soup(rng, n, spread, size):
    tri = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32) * np.float32(size) \
        + (rng.uniform(-1, 1, (n, 1, 3)) * spread).astype(np.float32)
    return tri


def surface(rng, n):
    """a tessellated height field: shared vertices, hits on edges, smooth normals"""
    k = max(2, int(np.sqrt(n / 2)))
    xs = np.linspace(-1, 1, k + 1, dtype=np.float32)
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    gz = (0.3 * np.sin(3 * gx + rng.uniform(0, 6)) * np.cos(2 * gy + rng.uniform(0, 6))).astype(np.float32)
    P = np.stack([gx, gz, gy], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)])


def make_scene(rng):
    meshes = []
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 4)
        n = int(rng.choice([1, 2, 7, 33, 200, 1500, 9000]))
        if kind == 0:
            tri = soup(rng, n, rng.choice([0.2, 1.0, 2.0]), rng.choice([0.02, 0.2, 0.8]))
        elif kind == 1:
            tri = surface(rng, n)
        elif kind == 2:   # duplicates + degenerates
            tri = soup(rng, max(n // 2, 1), 1.0, 0.3)
            tri = np.concatenate([tri, tri[::2]])
            tri[::5, 2] = tri[::5, 1]
        else:             # axis-aligned coplanar sheets (flat boxes, rays in the plane)
            tri = soup(rng, n, 1.0, 0.4)
            tri[:, :, int(rng.integers(0, 3))] = np.float32(rng.choice([-0.5, 0.0, 0.25]))
        pos = tri.reshape(-1, 3)
        nrm = rng.normal(size=pos.shape).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        uv = rng.uniform(-0.2, 1.2, (len(pos), 2)).astype(np.float32)
        idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
        m = material(rng)
        meshes.append(abi.MeshData(pos, nrm, uv, idx, m))
    spheres = []
    if rng.random() < 0.35:
        for _ in range(int(rng.integers(1, 6))):
            spheres.append(abi.SphereData(tuple(rng.uniform(-1, 1, 3)), float(rng.uniform(0.05, 0.6)), material(rng)))
    textures = []
    if rng.random() < 0.5:
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        textures.append(rng.uniform(0, 1, (h, w, 3)).astype(np.float32))
        for m in meshes[::2]:
            m.material.texture = 0
    lights = []
    for _ in range(int(rng.integers(1, 4))):
        k = rng.integers(0, 3)
        p = rng.uniform(-2.5, 2.5, 3)
        if k == 0:
            lights.append(abi.LightData.point(tuple(p), tuple(rng.uniform(0.2, 1, 3))))
        elif k == 1:
            lights.append(abi.LightData.segment(tuple(p), tuple(p + rng.uniform(-0.5, 0.5, 3)), tuple(rng.uniform(0, 1, 3)),
                                                tuple(rng.uniform(0, 1, 3))))
        else:
            lights.append(abi.LightData.parallelogram(tuple(p), tuple(rng.uniform(-0.5, 0.5, 3)), tuple(rng.uniform(-0.5, 0.5, 3)),
                                                      *[tuple(rng.uniform(0, 1, 3)) for _ in range(4)]))
    return abi.SceneData(meshes=meshes, spheres=spheres, lights=lights, textures=textures, name="fuzz")


def make_flags(rng, scene):
    f = dict(enableShading=int(rng.random() < 0.85), enableAccelStructure=int(rng.random() < 0.9),
             enableBvhSahBinning=int(rng.random() < 0.3), enableHardShadow=int(rng.random() < 0.6),
             enableSoftShadow=int(rng.random() < 0.35), enableRecursive=int(rng.random() < 0.6),
             enableNormalInterp=int(rng.random() < 0.4), enableTextureMapping=int(bool(scene.textures) and rng.random() < 0.7),
             enableBilinearTextureFiltering=int(rng.random() < 0.3), enableTransparency=int(rng.random() < 0.25),
             enableGlossyReflection=int(rng.random() < 0.15), enableMultipleRaysPerPixel=int(rng.random() < 0.15),
             enableDepthOfField=int(rng.random() < 0.1), enableBloomEffect=int(rng.random() < 0.15))
    p = dict(depth=int(rng.integers(0, 6)), parallelogram_n=int(rng.integers(1, 4)), segment_n=int(rng.integers(1, 5)),
             seed=int(rng.integers(0, 1 << 30)), rays_per_pixel_side=int(rng.integers(1, 3)), samples_dof=int(rng.integers(1, 3)),
             rays_per_reflection=int(rng.integers(1, 3)), bloom_threshold=float(rng.uniform(0.05, 0.6)))
    if not f["enableAccelStructure"] and scene.n_triangles > 2000:
        p["depth"] = min(p["depth"], 2)   # brute force x path trees: minutes of oracle time per case otherwise
    if f["enableRecursive"] and f["enableTransparency"]:
        p["depth"] = min(p["depth"], 3)   # up to three calls per hit (render.cpp:97-122): 3^depth paths per ray
    return f, p


def _progress(name, case, n_cases, bad, t_start):
    """campaigns (FUZZ_CASES in the thousands) report every 50 cases to FUZZ_PROGRESS (a file under gpurun_out/ keeps
    a long run from looking hung) and to stdout"""
    if (case + 1) % 50 == 0 or case + 1 == n_cases:
        line = f"{name}: {case + 1}/{n_cases} cases, {len(bad)} mismatches, {time.time() - t_start:.0f} s"
        print(line, flush=True)
        path = os.environ.get("FUZZ_PROGRESS")
        if path:
            with open(path, "a") as fh:
                fh.write(line + "\n")


def test_random_scenes_match_oracle(rt, oracle, abi):
    globals()["abi"] = abi
    n_cases = int(os.environ.get("FUZZ_CASES", "60"))
    seed0 = int(os.environ.get("FUZZ_SEED", "1"))
    orc = oracle
    bad = []
    t_start = time.time()
    rays_all = rays_exact = wide_cases = 0
    for case in range(n_cases):
        seed = seed0 + case
        rng = np.random.default_rng(seed)
        scene = make_scene(rng)
        flags, params = make_flags(rng, scene)
        f, p = abi.Features.make(**flags), abi.RenderParams.make(**params)
        W, H = int(rng.integers(1, 150)), int(rng.integers(1, 110))
        cam_args = (float(rng.uniform(20, 80)), W / H, tuple(rng.uniform(-0.3, 0.3, 3)), float(rng.uniform(0.5, 6)),
                    tuple(rng.uniform(-180, 180, 3) * np.array([1, 1, 0.2])))
        msg = []
        try:
            bvh, ob = rt.bvh(rt.scene(scene), f), orc.build(scene, f)
            rgb, cnt = rt.render(bvh, rt.camera(*cam_args), f, p, W, H, count=True)
            rays_all, rays_exact = rays_all + cnt.rays, rays_exact + cnt.rays_exact
            wide_cases += 1 if cnt.rays_exact < cnt.rays else 0
            want_hits = not (flags["enableMultipleRaysPerPixel"] or flags["enableDepthOfField"])
            orgb, otb, opr, ocnt = orc.render(ob, orc.camera(*cam_args), f, p, W, H, want_hits=want_hits)
            if want_hits:
                tb, pr = rt.render_hits(bvh, rt.camera(*cam_args), f, W, H)
                if not (np.array_equal(tb, otb) and np.array_equal(pr, opr)):
                    msg.append(f"hit records differ at {(pr != opr).sum()} prim / {(tb != otb).sum()} t")
            ng, nw = np.isnan(rgb), np.isnan(orgb)
            if not np.array_equal(ng, nw):
                msg.append(f"NaN positions differ {ng.sum()} vs {nw.sum()}")
            else:
                ok = ~nw
                err = np.abs(rgb[ok].astype(np.float64) - orgb[ok].astype(np.float64))
                lim = RTOL * np.abs(orgb[ok].astype(np.float64)) + ATOL
                if (err > lim).any():
                    msg.append(f"{(err > lim).sum()} colour values off, max error {err.max():.3e}")
            if cnt.rays_ref_equiv != ocnt.rays:
                msg.append(f"ray counts {cnt.rays_ref_equiv} vs {ocnt.rays}")
        except Exception as e:  # noqa: BLE001 - report the case and go on
            msg.append(f"exception {type(e).__name__}: {e}")
        if msg:
            on = [k for k, v in flags.items() if v]
            bad.append(f"seed {seed} {W}x{H} tris {scene.n_triangles} spheres {len(scene.spheres)} {on} {params}: "
                       + "; ".join(msg))
            print("MISMATCH", bad[-1], flush=True)
        _progress("random scenes", case, n_cases, bad, t_start)
    print(f"random scenes: {rays_all} rays, {rays_all - rays_exact} through the wide engine ({wide_cases} of {n_cases} cases), "
          f"{rays_exact} through the exact-topology engine", flush=True)
    assert wide_cases > n_cases // 10, "the campaign must exercise the wide engine"
    assert not bad, f"{len(bad)} of {n_cases} random cases differ:\n" + "\n".join(bad[:10])


def adversarial_rays(abi, rng, scene, n):
    """Rays that aim at the special cases of the slab / triangle / sphere tests: origins exactly ON vertices and on
    the coordinates of box faces, directions with exact zeros (the FLT_MIN / FLT_MAX slab quirk), directions along
    triangle edges, denormal and huge components, zero and negative limits, infinities and NaNs."""
    rays = np.zeros(n, abi.RAY_DTYPE)
    verts = np.concatenate([m.positions for m in scene.meshes])
    o = (rng.normal(size=(n, 3)) * 2).astype(np.float32)
    tgt = verts[rng.integers(0, len(verts), n)] + (rng.normal(size=(n, 3)) * 0.05).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    kind = rng.integers(0, 13, n)
    sel = kind == 1                                    # origin exactly on a vertex
    o[sel] = verts[rng.integers(0, len(verts), sel.sum())]
    sel = kind == 2                                    # one origin coordinate equal to a vertex coordinate (a box face)
    ax = rng.integers(0, 3, n)
    o[sel, ax[sel]] = verts[rng.integers(0, len(verts), sel.sum()), ax[sel]]
    sel = kind == 3                                    # aimed exactly at a vertex
    d[sel] = verts[rng.integers(0, len(verts), sel.sum())] - o[sel]
    sel = kind == 4                                    # one zero component
    d[sel, ax[sel]] = 0.0
    sel = kind == 5                                    # axis-parallel
    keep = rng.integers(0, 3, n)
    for a in range(3):
        d[sel & (keep != a), a] = 0.0
    sel = kind == 6                                    # along an edge, starting on it
    tri = rng.integers(0, len(verts) // 3, sel.sum()) * 3
    o[sel] = verts[tri]
    d[sel] = verts[tri + 1] - verts[tri]
    sel = kind == 7                                    # denormal / tiny components
    d[sel, ax[sel]] = np.float32(rng.choice([1e-42, -1e-42, 1e-38, 1e-30]))
    sel = kind == 8                                    # huge components
    d[sel] *= np.float32(1e30)
    sel = kind == 9                                    # negative zero components
    d[sel, ax[sel]] = np.float32(-0.0)
    sel = kind == 12                                   # far origins, either side of the 2^60 numerator guard
    far = rng.choice(np.float32([1e15, 9e17, 2e18, 1e25, 3e38]), sel.sum())
    o[sel, ax[sel]] = far * np.sign(o[sel, ax[sel]])
    d[sel] = (tgt[sel] - o[sel]).astype(np.float32)
    rays["origin"], rays["direction"] = o, d
    t = np.full(n, 3.402823466e+38, np.float32)
    lim = rng.integers(0, 10, n)
    t[lim == 0] = rng.uniform(0.0, 3.0, (lim == 0).sum()).astype(np.float32)
    t[lim == 1] = 0.0
    t[lim == 2] = np.float32(-1.0)
    t[(kind == 10) & (lim == 3)] = np.inf
    rays["t"] = t
    sel = kind == 11                                   # non-finite inputs
    bad = rng.integers(0, 4, n)
    rays["direction"][sel & (bad == 0), 0] = np.nan
    rays["origin"][sel & (bad == 1), 1] = np.inf
    rays["direction"][sel & (bad == 2), 2] = -np.inf
    rays["direction"][sel & (bad == 3)] = 0.0
    return rays


def test_adversarial_rays_intersect_and_trace(rt, oracle, abi):
    """≙ BvhInterface::intersect and getFinalColor on rays built to sit on the special cases."""
    from conftest import bits, nan_aware_equal_bits
    globals()["abi"] = abi
    n_cases = int(os.environ.get("FUZZ_RAY_CASES", "30"))
    seed0 = int(os.environ.get("FUZZ_SEED", "1"))
    bad = []
    n_hits = n_rays = n_lit = 0
    t_start = time.time()
    for case in range(n_cases):
        seed = 77000 + seed0 + case
        rng = np.random.default_rng(seed)
        scene = make_scene(rng)
        flags, params = make_flags(rng, scene)
        for k in ("enableMultipleRaysPerPixel", "enableDepthOfField", "enableBloomEffect"):  # pixel-loop features
            flags[k] = 0
        f, p = abi.Features.make(**flags), abi.RenderParams.make(**params)
        bvh, ob = rt.bvh(rt.scene(scene), f), oracle.build(scene, f)
        rays = adversarial_rays(abi, rng, scene, 3000 if flags["enableAccelStructure"] else 600)
        msg = []
        got_rays, got = rt.intersect_batch(bvh, rays, f)
        want_rays, want, _ = oracle.intersect_batch(ob, rays, f)
        if not np.array_equal(got["hit"], want["hit"]):
            msg.append(f"hit flags differ at {(got['hit'] != want['hit']).sum()}")
        elif not np.array_equal(got["prim"], want["prim"]):
            msg.append(f"prim ids differ at {(got['prim'] != want['prim']).sum()}")
        elif not np.array_equal(bits(got_rays["t"]), bits(want_rays["t"])):
            msg.append(f"t differs at {(bits(got_rays['t']) != bits(want_rays['t'])).sum()}")
        else:
            hit = want["hit"] == 1
            # NaN normals (a hit point at 1e25: inf - inf in the barycentrics): x86 and the GPU sign their
            # default NaN differently, NaN POSITIONS must agree
            if not nan_aware_equal_bits(got["normal"][hit], want["normal"][hit]):
                msg.append("normals differ")
        n_hits, n_rays = n_hits + int(want["hit"].sum()), n_rays + len(rays)
        rgb = rt.trace_batch(bvh, rays, f, p)
        orgb, _ = oracle.trace_batch(ob, rays, f, p)
        n_lit += int((np.nan_to_num(orgb) != 0).any(axis=1).sum())
        ng, nw = np.isnan(rgb), np.isnan(orgb)
        if not np.array_equal(ng, nw):
            msg.append(f"trace: NaN positions differ {ng.sum()} vs {nw.sum()}")
        else:
            ok = ~nw & np.isfinite(orgb)
            err = np.abs(rgb[ok].astype(np.float64) - orgb[ok].astype(np.float64))
            lim = RTOL * np.abs(orgb[ok].astype(np.float64)) + ATOL
            if (err > lim).any():
                msg.append(f"trace: {(err > lim).sum()} colour values off, max error {err.max():.3e}")
            inf_ok = np.array_equal(rgb[~nw & ~np.isfinite(orgb)], orgb[~nw & ~np.isfinite(orgb)])
            if not inf_ok:
                msg.append("trace: infinite colour values differ")
        if msg:
            on = [k for k, v in flags.items() if v]
            bad.append(f"seed {seed} tris {scene.n_triangles} spheres {len(scene.spheres)} {on} {params}: " + "; ".join(msg))
            print("MISMATCH", bad[-1], flush=True)
        _progress("adversarial rays", case, n_cases, bad, t_start)
    print(f"adversarial rays: {n_rays} rays, {n_hits} hits, {n_lit} non-black colours", flush=True)
    assert n_hits > n_rays // 20 and n_lit > n_rays // 50, "the rays must hit and light something"
    assert not bad, f"{len(bad)} of {n_cases} adversarial-ray cases differ:\n" + "\n".join(bad[:10])
