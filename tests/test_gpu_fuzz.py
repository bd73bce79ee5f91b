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
    return f, p


def test_random_scenes_match_oracle(rt, oracle, abi):
    globals()["abi"] = abi
    n_cases = int(os.environ.get("FUZZ_CASES", "60"))
    seed0 = int(os.environ.get("FUZZ_SEED", "1"))
    orc = oracle
    bad = []
    t_start = time.time()
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
        if case % 50 == 49:
            print(f"... {case + 1} cases, {len(bad)} bad, {time.time() - t_start:.0f} s", flush=True)
    assert not bad, f"{len(bad)} of {n_cases} random cases differ:\n" + "\n".join(bad[:10])
