// shim_selftest.cpp — drives the reference-named C++ surface (reference_api.hpp)
// the way src/main.cpp drives the reference: load a Scene, build a BvhInterface,
// renderRayTracing into a Screen, getFinalColor / bvh.intersect for single rays.
// tests/test_host_shim.py feeds it a scene file and compares every output bit
// with the same calls made through the Python binding of the C ABI.
//
//   shim_selftest <scene.bin> <out.bin> <width> <height>
//
// scene.bin (little endian): u32 n_meshes; per mesh: u32 nv, u32 nt, f32 pos[3nv],
// nrm[3nv], uv[2nv], u32 tri[3nt], f32 material[8] (kd, ks, shininess,
// transparency), u32 tex_w, tex_h, f32 texels[3*w*h]; u32 n_lights; per light:
// i32 kind, f32 v[21]; 17 feature bytes; f32 camera[9] (fovy, aspect, lookAt[3],
// dist, rot[3]); u32 n_rays; f32 rays[7*n_rays]; i32 ray_depth.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "reference_api.hpp"

template <class T>
static void rd(std::ifstream& f, T* p, size_t n)
{
    f.read(reinterpret_cast<char*>(p), (std::streamsize)(sizeof(T) * n));
    if (!f)
        throw std::runtime_error("scene file truncated");
}

int main(int argc, char** argv)
{
    if (argc != 5) {
        std::fprintf(stderr, "usage: %s scene.bin out.bin W H\n", argv[0]);
        return 2;
    }
    try {
        std::ifstream in(argv[1], std::ios::binary);
        Scene scene;
        uint32_t n_meshes;
        rd(in, &n_meshes, 1);
        for (uint32_t m = 0; m < n_meshes; m++) {
            uint32_t nv, nt;
            rd(in, &nv, 1), rd(in, &nt, 1);
            std::vector<float> pos(3 * nv), nrm(3 * nv), uv(2 * nv), mat(8);
            std::vector<uint32_t> tri(3 * nt);
            rd(in, pos.data(), pos.size()), rd(in, nrm.data(), nrm.size()), rd(in, uv.data(), uv.size());
            rd(in, tri.data(), tri.size()), rd(in, mat.data(), 8);
            Mesh mesh;
            for (uint32_t i = 0; i < nv; i++)
                mesh.vertices.push_back(Vertex { glm::vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                    glm::vec3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), glm::vec2(uv[2 * i], uv[2 * i + 1]) });
            for (uint32_t i = 0; i < nt; i++) {
                glm::uvec3 t;
                t.x = tri[3 * i], t.y = tri[3 * i + 1], t.z = tri[3 * i + 2];
                mesh.triangles.push_back(t);
            }
            mesh.material.kd = glm::vec3(mat[0], mat[1], mat[2]);
            mesh.material.ks = glm::vec3(mat[3], mat[4], mat[5]);
            mesh.material.shininess = mat[6], mesh.material.transparency = mat[7];
            uint32_t tw, th;
            rd(in, &tw, 1), rd(in, &th, 1);
            if (tw && th) {
                auto img = std::make_shared<Image>();
                img->width = (int)tw, img->height = (int)th;
                std::vector<float> px(3 * (size_t)tw * th);
                rd(in, px.data(), px.size());
                for (size_t i = 0; i < (size_t)tw * th; i++)
                    img->pixels.emplace_back(px[3 * i], px[3 * i + 1], px[3 * i + 2]);
                mesh.material.kdTexture = img;
            }
            scene.meshes.push_back(std::move(mesh));
        }
        uint32_t n_lights;
        rd(in, &n_lights, 1);
        for (uint32_t l = 0; l < n_lights; l++) {
            int32_t kind;
            float v[21];
            rd(in, &kind, 1), rd(in, v, 21);
            auto g = [&](int i) { return glm::vec3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); };
            if (kind == RT_LIGHT_POINT)
                scene.lights.emplace_back(PointLight { g(0), g(1) });
            else if (kind == RT_LIGHT_SEGMENT)
                scene.lights.emplace_back(SegmentLight { g(0), g(1), g(2), g(3) });
            else
                scene.lights.emplace_back(ParallelogramLight { g(0), g(1), g(2), g(3), g(4), g(5), g(6) });
        }
        Features features;
        rd(in, reinterpret_cast<uint8_t*>(&features), 17);
        float cam[9];
        rd(in, cam, 9);
        uint32_t n_rays;
        rd(in, &n_rays, 1);
        std::vector<Ray> rays(n_rays);
        rd(in, reinterpret_cast<float*>(rays.data()), 7 * (size_t)n_rays);
        int32_t ray_depth;
        rd(in, &ray_depth, 1);

        const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
        // src/main.cpp:512-520
        BvhInterface bvh { &scene, features };
        glm::ivec2 res;
        res.x = W, res.y = H;
        Screen screen { res };
        Trackball camera { cam[1], cam[0], cam[5] };
        camera.setCamera(glm::vec3(cam[2], cam[3], cam[4]), glm::vec3(cam[6], cam[7], cam[8]), cam[5]);
        renderRayTracing(scene, camera, bvh, screen, features);

        std::ofstream out(argv[2], std::ios::binary);
        const int32_t hdr[4] = { W, H, bvh.numLevels(), bvh.numLeaves() };
        out.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
        out.write(reinterpret_cast<const char*>(screen.pixels().data()), (std::streamsize)(sizeof(glm::vec3) * screen.pixels().size()));
        // single-ray API (src/main.cpp:398): getFinalColor + bvh.intersect
        for (const Ray& r : rays) {
            const glm::vec3 c = getFinalColor(scene, bvh, r, features, ray_depth);
            Ray rr = r;
            HitInfo h;
            h.normal = glm::vec3(0.0f);
            const bool hit = bvh.intersect(rr, h, features);
            const float rec[8] = { c.x, c.y, c.z, hit ? 1.0f : 0.0f, rr.t, h.normal.x, h.normal.y, h.normal.z };
            out.write(reinterpret_cast<const char*>(rec), sizeof(rec));
        }
        std::printf("shim_selftest: %dx%d frame, %u single rays, levels %d leaves %d\n", W, H, n_rays, hdr[2], hdr[3]);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "shim_selftest failed: %s\n", e.what());
        return 1;
    }
}
