// Test driver for mlmapping_amd/csrc/mlm_mapview.h — the reference's query inlines (include/mlmap.h:142-295) over a host copy of the
// block planes, the code behind the library's single-position queries — built by tests/test_mapview.py with g++
// -fsanitize=address,undefined (no HIP, no GPU).  Input blob: d_sub, n, n_blocks, n_pos, max_iter, inflate; keys [n_blocks*3] i32;
// collapsed [n_blocks] u8; log_odds [n_blocks*cells] f32; occ, infl [n_blocks*cells] u8; positions [n_pos*3] f64; glb ids
// [n_pos*3] i32 + cell ids [n_pos] i32.  Output: per position "occ occ_inflate inflate_occ odd grad[3] odd_at" as hex floats.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlm_mapview.h"

static bool rd(FILE *f, void *p, size_t n) { return std::fread(p, 1, n, f) == n; }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    double d_sub;
    int32_t hdr[4]; // n, n_blocks, n_pos, max_iter
    float inflate;
    if (!rd(f, &d_sub, 8) || !rd(f, hdr, sizeof hdr) || !rd(f, &inflate, 4)) return 2;
    const int n = hdr[0], nb = hdr[1], np = hdr[2], max_iter = hdr[3], C = n * n * n;
    std::vector<int32_t> keys((size_t)nb * 3), glb((size_t)np * 3), cid((size_t)np);
    std::vector<uint8_t> col((size_t)nb), occ((size_t)nb * C), infl((size_t)nb * C);
    std::vector<float> lo((size_t)nb * C);
    std::vector<double> pos((size_t)np * 3);
    if (!rd(f, keys.data(), keys.size() * 4) || !rd(f, col.data(), col.size()) || !rd(f, lo.data(), lo.size() * 4) || !rd(f, occ.data(), occ.size()) ||
        !rd(f, infl.data(), infl.size()) || !rd(f, pos.data(), pos.size() * 8) || !rd(f, glb.data(), glb.size() * 4) || !rd(f, cid.data(), cid.size() * 4))
        return 2;
    std::fclose(f);
    mlm_host::MapView v;
    v.d_sub = d_sub;
    v.n = n;
    v.cells = C;
    v.d_glb = d_sub * n; // map_local.cpp:60
    v.d_sub_half = d_sub * 0.5;
    v.lo = lo.data(), v.occ = occ.data(), v.infl = infl.data(), v.col = col.data();
    // an empty view answers "unknown" / 0.5 everywhere
    {
        int8_t o = 7;
        float p = 0;
        const double q[3] = {0.1, 0.2, 0.3};
        v.answer(0, q, 1, 0.f, 0, &o);
        v.answer(3, q, 1, 0.f, 0, &p);
        if (o != mlm_host::MapView::UNKNOWN || p != 0.5f) return 3;
    }
    // DirtyBoxes: whatever is marked stays covered (boxes only merge into larger ones), "anywhere" for non-finite / far boxes
    {
        uint64_t st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13, st ^= st >> 7, st ^= st << 17; return st; };
        for (int trial = 0; trial < 200; ++trial) {
            mlm_host::DirtyBoxes B;
            B.clear();
            std::vector<int> pts;
            const int marks = 1 + (int)(rnd() % 60);
            for (int m = 0; m < marks; ++m) {
                int lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = (int)(rnd() % 200) - 100;
                    hi[a] = lo[a] + (int)(rnd() % 12);
                }
                B.mark(lo, hi);
                for (int k = 0; k < 4; ++k)
                    for (int a = 0; a < 3; ++a) pts.push_back(lo[a] + (int)(rnd() % (uint64_t)(hi[a] - lo[a] + 1)));
                for (size_t i = 0; i + 2 < pts.size(); i += 3)
                    if (!B.covered(pts[i], pts[i + 1], pts[i + 2])) return 4;
            }
            if (B.n > mlm_host::DirtyBoxes::kMax || B.all) return 4;
        }
        mlm_host::DirtyBoxes B;
        B.clear();
        const double a0[3] = {0.0, 0.0, 0.0}, a1[3] = {1.0, 2.0, 3.0}, bad[3] = {0.0, std::nan(""), 0.0}, far[3] = {3e9, 0.0, 0.0};
        B.mark_world(a0, a1, 1.0);
        if (B.all || !B.covered(-1, 3, 4) || B.covered(-2, 0, 0)) return 4; // one block of margin each side
        B.mark_world(a0, bad, 1.0);
        if (!B.all) return 4;
        B.clear();
        B.mark_world(a0, far, 1.0);
        if (!B.all) return 4;
    }
    v.table_reset((size_t)nb);
    for (int b = 0; b < nb; ++b) v.table_insert(keys[3 * (size_t)b], keys[3 * (size_t)b + 1], keys[3 * (size_t)b + 2], b);
    for (int i = 0; i < np; ++i) { // one position per call, like a planner
        int8_t o0, o1, o2;
        float odd;
        double g[3];
        v.answer(0, &pos[3 * (size_t)i], 1, 0.f, 0, &o0);
        v.answer(1, &pos[3 * (size_t)i], 1, inflate, 0, &o1);
        v.answer(2, &pos[3 * (size_t)i], 1, 0.f, 0, &o2);
        v.answer(3, &pos[3 * (size_t)i], 1, 0.f, 0, &odd);
        v.answer(4, &pos[3 * (size_t)i], 1, 0.f, max_iter, g);
        std::printf("%d %d %d %a %a %a %a %a\n", o0, o1, o2, (double)odd, g[0], g[1], g[2],
                    (double)v.odd_at(glb[3 * (size_t)i], glb[3 * (size_t)i + 1], glb[3 * (size_t)i + 2], cid[(size_t)i]));
    }
    return 0;
}
