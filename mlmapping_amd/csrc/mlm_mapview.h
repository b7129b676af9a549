// mlm_mapview.h — the reference's query inlines (include/mlmap.h:142-295) over a HOST copy of the block planes: pure host code, no HIP.
// The library's host mirror (mlm_mirror.h) answers small query batches through it; tests/test_mapview.py builds it for the CPU with
// -fsanitize=address,undefined and holds every query kind to the oracle bit for bit.
//
// Layout mirrored: block slot s holds cells [s * cells, (s + 1) * cells) of log_odds / occupancy / inflate_occupancy, `collapsed[s]`
// marks a released block (vectors of size 1 in the reference: element 0 answers, mlmap.h:183-184,221-222); block keys are packed
// like the device's (three 21-bit biased indices) into an open-addressed table.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mlm_host {

struct MapView {
    // geometry (map_local.cpp:56-62)
    double d_sub = 0, d_glb = 0, d_sub_half = 0;
    int n = 0, cells = 0;
    // the mirrored planes (not owned)
    const float *lo = nullptr;
    const uint8_t *occ = nullptr, *infl = nullptr, *col = nullptr;
    // block key -> slot
    std::vector<unsigned long long> tk;
    std::vector<int> ts;
    uint32_t tmask = 0;
    static constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;
    enum { FREE = 1, OCCUPIED = 0, UNKNOWN = -1 }; // mlmap.h:109-114

    static uint32_t mix(unsigned long long k) { // (the device's mlm_mix)
        k ^= k >> 33;
        k *= 0xff51afd7ed558ccdull;
        k ^= k >> 33;
        k *= 0xc4ceb9fe1a85ec53ull;
        k ^= k >> 33;
        return (uint32_t)k;
    }
    // a block index beyond the key's 21 bits per axis names no block of the map
    static bool key_ok(int gx, int gy, int gz) {
        return !((((unsigned int)gx + (1u << 20)) >> 21) || (((unsigned int)gy + (1u << 20)) >> 21) || (((unsigned int)gz + (1u << 20)) >> 21));
    }
    static unsigned long long pack(int gx, int gy, int gz) {
        return ((unsigned long long)(((uint32_t)gx + (1u << 20)) & 0x1FFFFFu) << 42) | ((unsigned long long)(((uint32_t)gy + (1u << 20)) & 0x1FFFFFu) << 21) |
               (unsigned long long)(((uint32_t)gz + (1u << 20)) & 0x1FFFFFu);
    }
    void table_reset(size_t cap_blocks) {
        size_t ht = 1024;
        while (ht < cap_blocks * 4) ht <<= 1;
        tk.assign(ht, kEmpty);
        ts.assign(ht, -1);
        tmask = (uint32_t)(ht - 1);
    }
    void table_clear() {
        tk.clear();
        ts.clear();
        tmask = 0;
    }
    void table_insert(int gx, int gy, int gz, int slot) {
        const unsigned long long key = pack(gx, gy, gz);
        uint32_t p = mix(key) & tmask;
        while (tk[p] != kEmpty) p = (p + 1) & tmask;
        tk[p] = key;
        ts[p] = slot;
    }
    int find(int gx, int gy, int gz) const {
        if (!key_ok(gx, gy, gz) || !tmask) return -1;
        const unsigned long long key = pack(gx, gy, gz);
        for (uint32_t p = mix(key) & tmask;; p = (p + 1) & tmask) {
            const unsigned long long k = tk[p];
            if (k == key) return ts[p];
            if (k == kEmpty) return -1;
        }
    }

    static int cvt_int(double v) { // x86 cvttsd2si (mlm_cvt_int)
        if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
        return (int)v;
    }
    static int mul(int a, int b) { return (int)((unsigned int)a * (unsigned int)b); } // (wraps like the hardware; INT_MIN * n is UB in C++)
    // get_global_idx / get_subbox_id, map_local.h:148-152,167-173 — two independent divisions per axis; a cell coordinate outside
    // [0, n) maps to id 0 (operator[] default-inserts in the reference)
    void voxel_of(double x, double y, double z, int g[3], int &cid) const {
        g[0] = cvt_int(std::floor(x / d_glb));
        g[1] = cvt_int(std::floor(y / d_glb));
        g[2] = cvt_int(std::floor(z / d_glb));
        const int cx = cvt_int(std::floor(x / d_sub) - mul(g[0], n));
        const int cy = cvt_int(std::floor(y / d_sub) - mul(g[1], n));
        const int cz = cvt_int(std::floor(z / d_sub) - mul(g[2], n));
        if (cx < 0 || cy < 0 || cz < 0 || cx >= n || cy >= n || cz >= n) cid = 0;
        else cid = cz * n * n + cy * n + cx;
    }
    // getOccupancy, mlmap.h:170-193
    int occupancy(double x, double y, double z) const {
        int g[3], cid;
        voxel_of(x, y, z, g, cid);
        const int slot = find(g[0], g[1], g[2]);
        if (slot < 0) return UNKNOWN;
        if (col[slot]) cid = 0; // occupancy.size() == 1 -> occupancy[0], mlmap.h:183-184
        const uint8_t r = occ[(size_t)slot * cells + cid];
        return r == 'o' ? OCCUPIED : (r == 'f' ? FREE : UNKNOWN);
    }
    // logit_inv, mlmap.h:40: pow(10, x) / (1 + pow(10, x)) in double, narrowed by getOdd's float return
    static float logit_inv(float L) {
        const double p = std::pow(10.0, (double)L);
        return (float)(p / (1 + p));
    }
    // getOdd(glb_id, subbox_id), mlmap.h:227-235
    float odd_at(int gx, int gy, int gz, int cid) const {
        const int slot = find(gx, gy, gz);
        if (slot < 0) return 0.5f;
        if (col[slot]) cid = 0; // log_odds.size() == 1 -> log_odds[0], mlmap.h:221-222
        return logit_inv(lo[(size_t)slot * cells + cid]);
    }
    // one step along direction dir of subbox_neighbors (map_local.cpp:77-120): order +z,-z,+y,-y,+x,-x
    void neighbor(int dir, int g[3], int &cid) const {
        int c[3];
        c[2] = cid / (n * n);
        c[1] = (cid - c[2] * n * n) / n;
        c[0] = cid - c[2] * n * n - c[1] * n;
        const int axis = 2 - dir / 2, step = (dir & 1) ? -1 : 1;
        c[axis] += step;
        if (c[axis] >= n) { // (block indices wrap like the hardware's: a NaN position sits at INT_MIN)
            g[axis] = (int)((unsigned int)g[axis] + 1u);
            c[axis] = 0;
        } else if (c[axis] < 0) {
            g[axis] = (int)((unsigned int)g[axis] - 1u);
            c[axis] = n - 1;
        }
        cid = c[2] * n * n + c[1] * n + c[0];
    }
    // mode 0: getOccupancy  1: getOccupancy(pos, inflate)  2: getInflateOccupancy  3: getOdd  4: getOddGrad (k_query's modes)
    void answer(int mode, const double *pos, int count, float inflate, int max_iter, void *out) const {
        for (int i = 0; i < count; ++i) {
            const double x = pos[3 * (size_t)i], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
            if (mode == 0) {
                ((int8_t *)out)[i] = (int8_t)occupancy(x, y, z);
            } else if (mode == 1) {
                // the 19-point stencil in the reference's order, mlmap.h:142-169; Vec3(+-inflate) promotes the float to double
                const double f = inflate;
                static const int8_t o[19][3] = {{0, 0, 0},  {0, 0, 1},   {0, 0, -1}, {0, 1, 0},   {0, -1, 0}, {1, 0, 0},  {-1, 0, 0},
                                                {-1, 1, 0}, {-1, -1, 0}, {1, 1, 0},  {1, -1, 0},  {0, -1, 1}, {0, -1, -1}, {0, 1, 1},
                                                {0, 1, -1}, {-1, 0, 1},  {-1, 0, -1}, {1, 0, 1},  {1, 0, -1}};
                int res = FREE;
                for (int k = 0; k < 19 && res == FREE; ++k)
                    if (occupancy(x + (o[k][0] ? (o[k][0] > 0 ? f : -f) : 0.0), y + (o[k][1] ? (o[k][1] > 0 ? f : -f) : 0.0),
                                  z + (o[k][2] ? (o[k][2] > 0 ? f : -f) : 0.0)) == OCCUPIED)
                        res = OCCUPIED;
                ((int8_t *)out)[i] = (int8_t)res;
            } else if (mode == 2) { // getInflateOccupancy, mlmap.h:195-211
                int g[3], cid;
                voxel_of(x, y, z, g, cid);
                const int slot = find(g[0], g[1], g[2]);
                int res = UNKNOWN;
                if (slot >= 0 && !col[slot] && infl[(size_t)slot * cells + cid] == 'o') res = OCCUPIED;
                ((int8_t *)out)[i] = (int8_t)res;
            } else if (mode == 3) {
                int g[3], cid;
                voxel_of(x, y, z, g, cid);
                ((float *)out)[i] = odd_at(g[0], g[1], g[2], cid);
            } else { // getOddGrad, mlmap.h:237-295
                int g[3], cid;
                voxel_of(x, y, z, g, cid);
                float min_odd = odd_at(g[0], g[1], g[2], cid);
                const float ori_odd = min_odd;
                int ng[6][3], ncid[6], mg[3] = {0, 0, 0}, mcid = 0;
                bool flag = false;
                for (int iter = 0; iter < max_iter && !flag; ++iter)
                    for (int d = 0; d < 6; ++d) {
                        if (iter == 0) {
                            ng[d][0] = g[0], ng[d][1] = g[1], ng[d][2] = g[2];
                            ncid[d] = cid;
                        }
                        neighbor(d, ng[d], ncid[d]); // (keeps searching along the original direction, mlmap.h:267-268)
                        const float tmp = odd_at(ng[d][0], ng[d][1], ng[d][2], ncid[d]);
                        if (tmp < min_odd) {
                            min_odd = tmp;
                            mg[0] = ng[d][0], mg[1] = ng[d][1], mg[2] = ng[d][2];
                            mcid = ncid[d];
                            flag = true;
                        }
                    }
                double r[3] = {0.0, 0.0, 0.0};
                if (flag) { // subbox_id2xyz_glb_vec, map_local.h:208-213
                    const int cz = mcid / (n * n), cy = (mcid - cz * n * n) / n, cx = mcid - cz * n * n - cy * n;
                    const double s = (double)(ori_odd - min_odd);
                    r[0] = ((mg[0] * d_glb + cx * d_sub + d_sub_half) - x) * s;
                    r[1] = ((mg[1] * d_glb + cy * d_sub + d_sub_half) - y) * s;
                    r[2] = ((mg[2] * d_glb + cz * d_sub + d_sub_half) - z) * s;
                }
                ((double *)out)[3 * (size_t)i] = r[0];
                ((double *)out)[3 * (size_t)i + 1] = r[1];
                ((double *)out)[3 * (size_t)i + 2] = r[2];
            }
        }
    }
};

// Where the device map may differ from the mirror: up to kMax boxes of block indices (inclusive bounds) or "anywhere".  The rule the
// refresh relies on: a block inside ANY box ever marked since the last clear() is covered() — boxes are only merged into larger ones.
struct DirtyBoxes {
    static constexpr int kMax = 16;
    bool all = true;
    int n = 0;
    int lo[kMax][3], hi[kMax][3];
    void mark_all() {
        all = true;
        n = 0;
    }
    void clear() {
        all = false;
        n = 0;
    }
    void mark(const int blo[3], const int bhi[3]) {
        if (all) return;
        for (int k = 0; k < n; ++k) { // already covered?
            bool in = true;
            for (int a = 0; a < 3; ++a) in = in && blo[a] >= lo[k][a] && bhi[a] <= hi[k][a];
            if (in) return;
        }
        if (n == kMax) { // the list is full: one box around everything recorded so far
            for (int k = 1; k < n; ++k)
                for (int a = 0; a < 3; ++a) {
                    lo[0][a] = lo[k][a] < lo[0][a] ? lo[k][a] : lo[0][a];
                    hi[0][a] = hi[k][a] > hi[0][a] ? hi[k][a] : hi[0][a];
                }
            n = 1;
        }
        for (int a = 0; a < 3; ++a) {
            lo[n][a] = blo[a];
            hi[n][a] = bhi[a];
        }
        n++;
    }
    // a box of world coordinates with one block of margin each side; anything not finite or beyond the key range: "anywhere"
    void mark_world(const double wlo[3], const double whi[3], double d_glb) {
        int blo[3], bhi[3];
        for (int a = 0; a < 3; ++a) {
            const double l = std::floor(wlo[a] / d_glb) - 1.0, u = std::floor(whi[a] / d_glb) + 1.0;
            if (!(l > -1048000.0 && u < 1048000.0 && l <= u)) { // (NaN fails)
                mark_all();
                return;
            }
            blo[a] = (int)l;
            bhi[a] = (int)u;
        }
        mark(blo, bhi);
    }
    bool covered(int gx, int gy, int gz) const { // (what k_mirror_refresh evaluates per block)
        if (all) return true;
        for (int k = 0; k < n; ++k)
            if (gx >= lo[k][0] && gx <= hi[k][0] && gy >= lo[k][1] && gy <= hi[k][1] && gz >= lo[k][2] && gz <= hi[k][2]) return true;
        return false;
    }
};

} // namespace mlm_host
