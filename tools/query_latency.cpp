// query_latency — what ONE position costs through the drop-in query interface, measured the way a planner would call it: a C++
// program (g++, no HIP headers: only include/mlmap_facade.hpp over the C ABI) that integrates a few frames and then calls
// getOccupancy / getOdd / getOddGrad / getOccupancy(pos, inflate) / getInflateOccupancy one position per call, with a
// std::chrono clock around every call.  The reference answers these with one hash lookup on the host (include/mlmap.h:170-295).
// Input: the blob tests/test_gpu_boundary.py::test_cpp_facade_client_process writes (mlm_config, {frames, W, H, positions},
// frames as pose[7] + uint16 image, positions as doubles).  Output: one JSON object on stdout.
// Built by mlmapping_amd/csrc/Makefile into mlmapping_amd/lib/mlm_query_latency; run by bench.py (extra.single_query_us).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "mlmap_facade.hpp"

struct Vec3 {
    double d[3];
    double operator[](int i) const { return d[i]; }
};
static bool read_exact(FILE *f, void *p, size_t n) { return std::fread(p, 1, n, f) == n; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Stat {
    double p50, p99, mean;
};
static Stat stat_of(std::vector<double> &v) {
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    return Stat{v[v.size() / 2], v[(size_t)(v.size() * 0.99)], s / v.size()};
}

// `--callback <blob>`: only the reference's callback (src/mlmap.cpp:463-532), `calls` times after 20 untimed ones, every third followed by
// inflate_map like the shipped 10 Hz timer (launch files with mlmapping_apply_inflate) when the blob says so.  Blob: mlm_config,
// {frames, W, H, is_f32, calls, inflate_every, poses}, double camera2odom_latency, the images (float32 metres or uint16 millimetres), the
// poses (q w,x,y,z then t; call r uses image r % frames and pose r % poses).  Output: one JSON object.
static int callback_mode(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return 2;
    mlm_config cfg;
    int32_t hdr[7];
    double latency = 0;
    if (!read_exact(f, &cfg, sizeof cfg) || !read_exact(f, hdr, sizeof hdr) || !read_exact(f, &latency, sizeof latency)) return 2;
    const int n_frames = hdr[0], W = hdr[1], H = hdr[2], is_f32 = hdr[3], calls = hdr[4], inflate_every = hdr[5], n_poses = hdr[6];
    const size_t px_bytes = is_f32 ? 4 : 2;
    try {
        mlmap_hip::mlmap map;
        mlm_limits lim{};
        lim.max_blocks = 16384;
        lim.max_points = W * H;
        lim.max_batch = 2;
        map.init_map(cfg, 0, &lim);
        std::vector<std::vector<unsigned char>> imgs((size_t)n_frames, std::vector<unsigned char>((size_t)W * H * px_bytes));
        std::vector<double> poses((size_t)n_poses * 7);
        for (int k = 0; k < n_frames; ++k)
            if (!read_exact(f, imgs[(size_t)k].data(), imgs[(size_t)k].size())) return 2;
        if (n_poses < 1 || !read_exact(f, poses.data(), poses.size() * 8)) return 2;
        std::fclose(f);
        const double zero3[3] = {0, 0, 0};
        std::vector<double> t_cb;
        for (int r = 0; r < calls + 20; ++r) {
            const int k = r % n_frames;
            const double *pose = &poses[(size_t)(r % n_poses) * 7];
            double T[7];
            const double t0 = now_us();
            if (mlm_integrate_callback(map.handle(), imgs[(size_t)k].data(), is_f32, W, H, 0.0, pose + 4, pose, zero3, 0.0, zero3, 0.0, latency, 1,
                                       T) != MLM_OK)
                throw std::runtime_error(mlm_last_error(map.handle()));
            if (inflate_every > 0 && r % inflate_every == inflate_every - 1 && mlm_inflate_map(map.handle(), T + 4) != MLM_OK)
                throw std::runtime_error(mlm_last_error(map.handle()));
            if (r >= 20) t_cb.push_back(now_us() - t0);
        }
        const Stat cb = stat_of(t_cb);
        std::printf("{\"p50\": %.2f, \"p99\": %.2f, \"mean\": %.2f, \"calls\": %zu}\n", cb.p50, cb.p99, cb.mean, t_cb.size());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "query_latency: %s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (argc >= 3 && std::string(argv[1]) == "--callback") return callback_mode(argv[2]);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    mlm_config cfg;
    int32_t hdr[4];
    if (!read_exact(f, &cfg, sizeof cfg) || !read_exact(f, hdr, sizeof hdr)) return 2;
    const int n_frames = hdr[0], W = hdr[1], H = hdr[2], n_pos = hdr[3];
    try {
        mlmap_hip::mlmap map;
        mlm_limits lim{};
        lim.max_blocks = 16384;
        lim.max_points = W * H;
        lim.max_batch = 2;
        map.init_map(cfg, 0, &lim);
        std::vector<std::vector<uint16_t>> imgs((size_t)n_frames, std::vector<uint16_t>((size_t)W * H));
        std::vector<double> poses((size_t)n_frames * 7);
        for (int k = 0; k < n_frames; ++k)
            if (!read_exact(f, &poses[(size_t)k * 7], 7 * 8) || !read_exact(f, imgs[(size_t)k].data(), imgs[(size_t)k].size() * 2)) return 2;
        std::vector<double> pos((size_t)n_pos * 3);
        if (!read_exact(f, pos.data(), pos.size() * 8)) return 2;
        auto integrate = [&](int k) {
            map.set_depth_image(imgs[(size_t)k].data(), W, H);
            map.set_pose(&poses[(size_t)k * 7], &poses[(size_t)k * 7] + 4);
            map.project_depth();
            map.update_map();
        };
        for (int k = 0; k + 1 < n_frames; ++k) integrate(k);
        const double ct[3] = {poses[4], poses[5], poses[6]};
        map.inflate_map(Vec3{{ct[0], ct[1], ct[2]}});
        // the first query after the map changed pays for bringing the host mirror up to date; the very first one also pins the planes
        long long sink = 0;
        auto first_after = [&](std::vector<double> &out, int reps) {
            for (int r = 0; r < reps; ++r) {
                integrate(n_frames - 1);
                const Vec3 p{{pos[0], pos[1], pos[2]}};
                const double t0 = now_us();
                sink += map.getOccupancy(p);
                out.push_back(now_us() - t0);
            }
        };
        std::vector<double> ever, first, first_async;
        first_after(ever, 1);
        first_after(first, 200);
        mlm_set_async(map.handle(), 1); // (asynchronous integrate calls: the query itself waits for what was submitted)
        first_after(first_async, 3);    // (warm-up of the mode)
        first_async.clear();
        first_after(first_async, 100);
        mlm_set_async(map.handle(), 0);
        // steady state: the map does not change between the calls
        std::vector<double> t_occ, t_odd, t_grad, t_occ_i, t_infl, t_clock;
        double fsink = 0;
        for (int rep = 0; rep < 4; ++rep)
            for (int i = 0; i < n_pos; ++i) {
                const Vec3 p{{pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]}};
                double t0 = now_us();
                sink += map.getOccupancy(p);
                double t1 = now_us();
                fsink += map.getOdd(p);
                double t2 = now_us();
                const mlmap_hip::Vec3d g = map.getOddGrad(p);
                double t3 = now_us();
                sink += map.getOccupancy(p, 0.15f);
                double t4 = now_us();
                sink += map.getInflateOccupancy(p);
                double t5 = now_us();
                double t6 = now_us();
                fsink += g[0];
                if (rep == 0) continue; // (warm-up pass)
                t_occ.push_back(t1 - t0);
                t_odd.push_back(t2 - t1);
                t_grad.push_back(t3 - t2);
                t_occ_i.push_back(t4 - t3);
                t_infl.push_back(t5 - t4);
                t_clock.push_back(t6 - t5);
            }
        // the reference's own call: depth_odom_input_callback with its 500 rand() samples (src/mlmap.cpp:463-507), one frame per call, from
        // this C++ process (no interpreter between the clock and the library)
        std::vector<double> t_cb;
        {
            const double zero3[3] = {0, 0, 0};
            for (int r = 0; r < 320; ++r) {
                const int k = r % n_frames;
                const double t0 = now_us();
                if (mlm_integrate_callback(map.handle(), imgs[(size_t)k].data(), 0, W, H, 0.0, &poses[(size_t)k * 7 + 4], &poses[(size_t)k * 7], zero3, 0.0, zero3, 0.0, 0.0, 1,
                                           nullptr) != MLM_OK)
                    throw std::runtime_error(mlm_last_error(map.handle()));
                if (r >= 20) t_cb.push_back(now_us() - t0);
            }
        }
        const Stat cb = stat_of(t_cb);
        mlm_frame_stats st{};
        mlm_get_frame_stats(map.handle(), &st);
        const Stat a = stat_of(t_occ), b = stat_of(t_odd), c = stat_of(t_grad), d = stat_of(t_occ_i), e = stat_of(t_infl), k = stat_of(t_clock), fi = stat_of(first), fa = stat_of(first_async);
        std::printf("{\"calls_per_kind\": %zu, \"clock_overhead_us\": %.4f, "
                    "\"getOccupancy\": {\"p50\": %.4f, \"p99\": %.4f, \"mean\": %.4f}, "
                    "\"getOdd\": {\"p50\": %.4f, \"p99\": %.4f, \"mean\": %.4f}, "
                    "\"getOddGrad\": {\"p50\": %.4f, \"p99\": %.4f, \"mean\": %.4f}, "
                    "\"getOccupancy_inflate\": {\"p50\": %.4f, \"p99\": %.4f, \"mean\": %.4f}, "
                    "\"getInflateOccupancy\": {\"p50\": %.4f, \"p99\": %.4f, \"mean\": %.4f}, "
                    "\"first_query_after_integrate\": {\"p50\": %.2f, \"p99\": %.2f, \"mean\": %.2f, \"samples\": %zu, \"includes\": \"the wait for the integrate call's own refresh of the host mirror\"}, "
                    "\"first_query_after_async_integrate\": {\"p50\": %.2f, \"p99\": %.2f, \"mean\": %.2f, \"samples\": %zu, \"includes\": \"the frame itself: the query drains what was submitted\"}, "
                    "\"first_query_ever_us\": %.2f, "
                    "\"callback_sampled500_cpp\": {\"p50\": %.2f, \"p99\": %.2f, \"mean\": %.2f, \"calls\": %zu}, "
                    "\"n_host_queries\": %lld, \"n_mirror_refreshes\": %lld, \"n_mirror_blocks\": %lld, \"n_blocks\": %lld, \"sink\": %lld}\n",
                    t_occ.size(), k.p50, a.p50, a.p99, a.mean, b.p50, b.p99, b.mean, c.p50, c.p99, c.mean, d.p50, d.p99, d.mean, e.p50, e.p99, e.mean,
                    fi.p50, fi.p99, fi.mean, first.size(), fa.p50, fa.p99, fa.mean, first_async.size(), ever[0], cb.p50, cb.p99, cb.mean, t_cb.size(), (long long)st.n_host_queries, (long long)st.n_mirror_refreshes, (long long)st.n_mirror_blocks,
                    (long long)st.n_blocks, sink + (long long)fsink);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "query_latency: %s\n", e.what());
        return 1;
    }
    std::fclose(f);
    return 0;
}
