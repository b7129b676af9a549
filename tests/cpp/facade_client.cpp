// A planner-side client of include/mlmap_facade.hpp, built by tests/test_gpu_boundary.py with g++ (no HIP headers, no
// torch: only the C ABI) and run as a fresh child process: it integrates frames read from a file through the facade's
// reference-named methods and prints its answers, which the test compares with the ctypes path and the oracle.
// Vec3 below stands for Eigen::Vector3d (anything with operator[](int) -> double works with the facade's templates).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlmap_facade.hpp"

struct Vec3 {
    double d[3];
    double operator[](int i) const { return d[i]; }
};
struct Vec3I {
    int d[3];
    int operator[](int i) const { return d[i]; }
};

static bool read_exact(FILE *f, void *p, size_t n) { return std::fread(p, 1, n, f) == n; }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    mlm_config cfg;
    int32_t hdr[4]; // frames, width, height, positions
    if (!read_exact(f, &cfg, sizeof cfg) || !read_exact(f, hdr, sizeof hdr)) return 2;
    const int n_frames = hdr[0], W = hdr[1], H = hdr[2], n_pos = hdr[3];
    try {
        mlmap_hip::mlmap map;
        mlm_limits lim{};
        lim.max_blocks = 8192;
        lim.max_points = W * H;
        map.init_map(cfg, 0, &lim);
        std::vector<uint16_t> img((size_t)W * H);
        for (int k = 0; k < n_frames; ++k) {
            double pose[7];
            if (!read_exact(f, pose, sizeof pose) || !read_exact(f, img.data(), img.size() * 2)) return 2;
            map.set_depth_image(img.data(), W, H);
            map.set_pose(pose, pose + 4);
            map.project_depth();
            map.update_map();
        }
        if (!map.has_data || !map.map_updated) return 3;
        const double ct[3] = {0.0, 0.0, 1.5};
        map.inflate_map(Vec3{{ct[0], ct[1], ct[2]}});
        std::vector<double> pos((size_t)n_pos * 3);
        if (!read_exact(f, pos.data(), pos.size() * 8)) return 2;
        for (int i = 0; i < n_pos; ++i) {
            const Vec3 p{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}};
            const mlmap_hip::Vec3d g = map.getOddGrad(p);
            const mlmap_hip::Vec3d g2 = map.getOddGrad(p, 2);
            std::printf("%d %d %d %a %a %a %a %a %a %a\n", map.getOccupancy(p), map.getOccupancy(p, 0.15f), map.getInflateOccupancy(p),
                        (double)map.getOdd(p), g[0], g[1], g[2], g2[0], g2[1], g2[2]);
        }
        // getOdd(glb_id, subbox_id) and setFree_map_in_bound
        std::printf("at %a %a\n", (double)map.getOdd(Vec3I{{0, 0, 1}}, 7), (double)map.getOdd(Vec3I{{40, 40, 40}}, 0));
        map.setFree_map_in_bound(Vec3{{0.5, -0.5, 1.0}}, Vec3{{1.0, 0.5, 1.5}});
        std::printf("free %d %d\n", map.getOccupancy(Vec3{{0.75, 0.0, 1.25}}), (int)(map.getOccupancy(Vec3{{0.75, 0.0, 1.25}}) == mlmap_hip::mlmap::FREE));
        // the visualisers' loops over local_map->observed_group_map, written as in src/rviz_vis.cpp:280-321, against the snapshot
        {
            const auto snap = map.local_map_snapshot();
            const mlmap_hip::local_map_view *localmap = snap.get();
            size_t n_o = 0, n_f = 0, n_infl = 0, n_front = 0;
            double sx = 0, sy = 0, sz = 0, sl = 0;
            for (auto iter = localmap->observed_group_map.begin(); iter != localmap->observed_group_map.end(); iter++) {
                int subbox_id = 0;
                for (auto it = iter->second.inflate_occupancy.begin(); it != iter->second.inflate_occupancy.end(); it++) {
                    if (*it == 'o') {
                        const mlmap_hip::PointF p = localmap->subbox_id2xyz_glb(iter->first, subbox_id);
                        sx += p.x, sy += p.y, sz += p.z;
                        ++n_infl;
                    }
                    subbox_id++;
                }
                for (auto it = iter->second.frontier.begin(); it != iter->second.frontier.end(); it++) ++n_front;
                for (char c : iter->second.occupancy) n_o += c == 'o', n_f += c == 'f';
                for (float l : iter->second.log_odds) sl += l;
            }
            std::printf("snapshot %zu %zu %zu %zu %zu %a %a %a %a\n", localmap->observed_group_map.size(), n_o, n_f, n_infl, n_front, sx, sy, sz, sl);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "facade_client: %s\n", e.what());
        return 1;
    }
    std::fclose(f);
    return 0;
}
