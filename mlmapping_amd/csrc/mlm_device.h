// mlm_device.h — device helpers shared by the map-update kernels.
//
// Index arithmetic is FP64 in the reference's operation order; the TU is built with -ffp-contract=off so
// no mul+add is fused (the reference is an SSE2 build: CMakeLists.txt:4, SURVEY.md §7 hard part 2).
#pragma once
#include <hip/hip_runtime.h>
#include "mlm_types.h"
#include "mlm_host.h" // mlm_cv_f32_to_u16 (host + device)

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// x86-64 static_cast<int>(double) == cvttsd2si: NaN / out-of-range -> INT_MIN.  gfx950 v_cvt_i32_f64 saturates
// and maps NaN to 0, so the x86 behaviour the reference silently relies on is spelled out.
// Pointers loaded from MlmDev / MlmFrame are generic to the compiler, which then emits FLAT instructions (they count
// on lgkmcnt as well as vmcnt, so every LDS / scalar-load wait also waits for them).  All of them point to device
// memory: mlm_gp() re-types one as an address-space-1 pointer so that GLOBAL instructions are emitted; the g_atomic_*
// wrappers are the device-scope relaxed atomics of HIP's atomicOr/atomicAdd/atomicMin on such pointers.
#define MLM_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ MLM_GLOBAL T *mlm_gp(T *p) { return (MLM_GLOBAL T *)p; }
// A word every lane of the workgroup reads at the same address and that no kernel-mate writes before the read: loaded through
// the scalar cache (address space 4 makes the compiler emit s_load) — a hit there answers in a fraction of a vector load's time.
// (the scalar cache is invalidated at every kernel start: what an earlier kernel wrote, atomics included, is seen)
__device__ __forceinline__ unsigned int mlm_uniform_word(const unsigned int *p) {
    return *(const __attribute__((address_space(4))) unsigned int *)(unsigned long long)p;
}
__device__ __forceinline__ double mlm_uniform_f64(const double *p) { return *(const __attribute__((address_space(4))) double *)(unsigned long long)p; }
// The same for LDS: a generic pointer to __shared__ memory (e.g. a volatile one used for wave-synchronous exchange)
// compiles to FLAT instructions with 64-bit address arithmetic; mlm_lp() re-types it as address space 3 (DS instructions).
#define MLM_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ MLM_LDS T *mlm_lp(T *p) { return (MLM_LDS T *)p; }
template <class T> __device__ __forceinline__ T g_atomic_or(MLM_GLOBAL T *p, T v) {
    return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ T g_atomic_add(MLM_GLOBAL T *p, T v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ T g_atomic_max(MLM_GLOBAL T *p, T v) {
    return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ T g_atomic_min(MLM_GLOBAL T *p, T v) {
    return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Broadcast of one lane's value when the source lane is wave-uniform: v_readlane_b32 (the general __shfl is a
// ds_bpermute_b32, i.e. an LDS round trip per call).
__device__ __forceinline__ int mlm_readlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t mlm_readlane(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
typedef unsigned int mlm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mlm_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int mlm_u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ int mlm_cvt_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

// QuaternionBase::_transformVector (Eigen generic path): v + w*(2 q x v) + q x (2 q x v); so3.cpp:80-84
__device__ __forceinline__ void mlm_quat_rot(const double q[4], double vx, double vy, double vz, double &ox,
                                             double &oy, double &oz) {
    const double qx = q[1], qy = q[2], qz = q[3], qw = q[0];
    double ux = qy * vz - qz * vy;
    double uy = qz * vx - qx * vz;
    double uz = qx * vy - qy * vx;
    ux = ux + ux;
    uy = uy + uy;
    uz = uz + uz;
    const double cx = qy * uz - qz * uy;
    const double cy = qz * ux - qx * uz;
    const double cz = qx * uy - qy * ux;
    ox = (vx + qw * ux) + cx;
    oy = (vy + qw * uy) + cy;
    oz = (vz + qw * uz) + cz;
}

// map_awareness.h:115-118
__device__ __forceinline__ double mlm_fast_atan(double x) { return x * (45 - (x - 1) * (14 + 3.83 * x)); }
// map_awareness.h:86-113 (deg2rad is the unparenthesised macro M_PI / 180, map_awareness.h:7)
__device__ __forceinline__ double mlm_fast_atan2(double y, double x) {
    const double input = y / x;
    const double a_input = fabs(input);
    double res;
    if (a_input > 1)
        res = copysign(M_PI / 180 * (90 - mlm_fast_atan(1 / a_input)), input);
    else
        res = copysign(M_PI / 180 * mlm_fast_atan(a_input), input);
    if (x > 0) return res;
    if (y >= 0) return res + M_PI;
    return res - M_PI;
}

// Index of a coordinate: the reference computes (int)(v / d) or floor(v / d) with a correctly rounded FP64 division.
// q = v * (1/d) differs from RN(v / d) by less than 4e-16 |q| (one rounding in 1/d, one in the product, one in the
// division), so wherever q is farther than 4e-15 max(1, |q|) from every integer, trunc and floor of q and of the true
// quotient agree, and the division (three quarter-rate instructions + a dozen FMAs) is skipped.  `exact` is cleared
// when q is too close to an integer (or not finite / huge): the caller then takes the division.
__device__ __forceinline__ double mlm_quot(double v, double inv_d, bool &exact) {
    const double q = v * inv_d;
    const double r = q - rint(q);
    if (!(fabs(r) > 4e-15 * fmax(1.0, fabs(q)) && fabs(q) < 1e9)) exact = false; // NaN fails the comparison
    return q;
}

// xyz2RhoPhiZwithBoderCheck, map_awareness.cpp:84-107
__device__ __forceinline__ bool mlm_bin_point(const MlmDev &P, double x, double y, double z, int &rho_idx,
                                              int &phi_idx, int &z_idx, bool &can_do_cast) {
    const double rho = sqrt(x * x + y * y);
    double phi = mlm_fast_atan2(y, x);
    if (phi < 0) phi += 2 * M_PI;
    const double zz = z - P.z_border_min;
    bool exact = true;
    const double qr = mlm_quot(rho, P.inv_dRho, exact), qp = mlm_quot(phi, P.inv_dPhi, exact), qz = mlm_quot(zz, P.inv_dZ, exact);
    rho_idx = (int)qr;
    phi_idx = (int)qp;
    z_idx = (int)floor(qz);
    if (!__all(exact)) { // rare: the whole wave skips the divisions otherwise
        if (!exact) {
            rho_idx = mlm_cvt_int(rho / P.dRho);
            phi_idx = mlm_cvt_int(phi / P.dPhi);
            z_idx = mlm_cvt_int(floor(zz / P.dZ));
        }
    }
    can_do_cast = (rho_idx >= 0 && phi_idx >= 0 && phi_idx < P.nPhi);
    return can_do_cast && z_idx >= 0 && rho_idx < P.nRho && z_idx < P.nZ;
}

// ---- The same bins WITHOUT the correctly rounded divisions and square root (k_bin_sectors, a third of its FP64 instructions).
// Only the three integer indices leave the chain projection -> pose -> sqrt / atan2 -> quotient, so the chain may be evaluated
// with cheaper forms — products with reciprocals, a matrix instead of the quaternion formula, fused multiply-adds, v_rcp_f64 /
// v_rsq_f64 refined once — as long as every quotient is PROVABLY on the same side of every integer as the reference's.  Error
// budget (u = 2^-53, lengths in metres):
//   * the sensor-frame coordinates: m * RN(1/f) against RN(m / f): <= 3.1 u |v|;
//   * rotation and translation: the reference rounds each of 33 operations on intermediates <= g |v| (g = MlmFrame::m_gain, 3 for a
//     unit quaternion); here the same linear map is a matrix (entries from the host, good to a few u g) applied with FMAs; the two
//     results differ by at most ~50 u (g |v| + |t|): with S = g |v|_1 + |t_ls|_1 + 1, E = 1e-12 S is two hundred times that and is
//     what the margin uses for |dx|, |dy|, |dz|;
//   * the approximate reciprocal and square root: relative error <= 4e-12 after ONE refinement of a seed good to 2^-20
//     (tests/test_gpu_parity.py::test_fast_bin_seeds measures the seeds on the device: v_rcp_f64 / v_rsq_f64 deliver far more);
//   * rho = sqrt(x^2 + y^2): |d rho| <= |dx| + |dy| + 1e-11 rho;   z: |dz| <= E;
//   * phi: the reference's polynomial in t = min(|x|,|y|) / max(|x|,|y|) is continuous over its branch (|y/x| = 1) and over the
//     quadrant joins except at phi = 0 = 2 pi, an integer of the quotient; its slope is at most 1.1 rad per unit t;
//     |dt| <= (|dx| + |dy|) / max(|x|,|y|) + 4e-12 <= 2.9 E / rho + 4e-12.  (x = +0 EXACTLY is a value of its own — the reference
//     returns 3 pi / 2 for y > 0 there, pi / 2 on either side —: a lane within 4 E of x = 0 is never sure.)
// A lane whose quotient lies nearer to an integer than that budget (with safety factors, see the margin) is not `sure`; the
// caller then evaluates the reference's own sequence for the wave (mlm_bin_point) — about one wave in 10^5.
__device__ __forceinline__ double mlm_rcp_approx(double b) {
    const double r = __builtin_amdgcn_rcp(b);
    return fma(r, fma(-b, r, 1.0), r);
}
__device__ __forceinline__ double mlm_sqrt_approx(double a) {
    const double r = __builtin_amdgcn_rsq(a);
    const double g = a * r, h = 0.5 * r;
    return fma(g, fma(-h, g, 0.5), g);
}
// (vx, vy, vz): the point in the sensor frame; v_l1 >= |vx| + |vy| + |vz|
__device__ __forceinline__ bool mlm_bin_point_fast(const MlmDev &P, const MlmFrame &F, double vx, double vy, double vz, double v_l1, int &rho_idx,
                                                   int &phi_idx, int &z_idx, bool &can_do_cast, bool &sure) {
    // p_l = T_ls p_s as a matrix product with fused multiply-adds: nothing here has to round like the reference
    const double x = fma(F.m_ls[0], vx, fma(F.m_ls[1], vy, fma(F.m_ls[2], vz, F.t_ls[0])));
    const double y = fma(F.m_ls[3], vx, fma(F.m_ls[4], vy, fma(F.m_ls[5], vz, F.t_ls[1])));
    const double z = fma(F.m_ls[6], vx, fma(F.m_ls[7], vy, fma(F.m_ls[8], vz, F.t_ls[2])));
    const double E = 1e-12 * fma(F.m_gain, v_l1, F.t_l1);
    const double rho = mlm_sqrt_approx(fma(x, x, y * y));
    // fast_atan2 (map_awareness.h:86-113) on t = smaller / larger component
    const double ax = fabs(x), ay = fabs(y);
    const bool steep = ay > ax; // |y / x| > 1
    const double t = (steep ? ax : ay) * mlm_rcp_approx(steep ? ay : ax);
    double deg = mlm_fast_atan(t);
    if (steep) deg = 90 - deg;
    double res = M_PI / 180 * deg;
    if ((x < 0) != (y < 0)) res = -res; // sign of y / x
    double phi = x > 0 ? res : (y >= 0 ? res + M_PI : res - M_PI);
    if (phi < 0) phi += 2 * M_PI;
    const double zz = z - P.z_border_min;
    const double qr = rho * P.inv_dRho, qp = phi * P.inv_dPhi, qz = zz * P.inv_dZ;
    // ONE margin for the three quotients, in their units: the largest of the three budgets above — 4 E (1 + 1 / rho) covers 3 E
    // (rho), 2 E (z) and 4 E / rho (phi), 2e-11 (1 + rho) the approximate operations' share — times the largest of the three
    // 1 / cell-size factors, plus the quotients' own roundings
    const double M = (4.0 * E * (1.0 + mlm_rcp_approx(rho)) + 2e-11 * (1.0 + rho)) * P.inv_d_max + 2e-12 * (((1.0 + qr) + qp) + fabs(qz));
    // (NaN and infinities fail the comparisons: a point on the axis, x = y = 0, goes to the exact sequence)
    sure = fabs(qr - rint(qr)) > M && fabs(qp - rint(qp)) > M && fabs(qz - rint(qz)) > M && qr < 1e9 && fabs(qz) < 1e9 && rho > 1e-9 &&
           ax > 4.0 * E; // (x = +0 exactly is a value of its own in the reference: 3 pi / 2 for y > 0)
    rho_idx = (int)qr;
    phi_idx = (int)qp;
    z_idx = (int)floor(qz);
    can_do_cast = (rho_idx >= 0 && phi_idx >= 0 && phi_idx < P.nPhi);
    return can_do_cast && z_idx >= 0 && rho_idx < P.nRho && z_idx < P.nZ;
}

// VectorHasher (map_awareness.h:31-41): 32-bit wrap-around int arithmetic, arithmetic >>, result sign-extended
// to size_t by libstdc++; bucket = code % bucket_count (std::__detail::_Mod_range_hashing).
__device__ __forceinline__ uint32_t mlm_hash_step(uint32_t h, int v) {
    return h ^ ((uint32_t)v + 0x9e3779b9u + (h << 6) + (uint32_t)((int32_t)h >> 2));
}
__device__ __forceinline__ unsigned long long mlm_hash_rpz(int rho, int phi, int z) {
    uint32_t h = 3u;
    h = mlm_hash_step(h, rho);
    h = mlm_hash_step(h, phi);
    h = mlm_hash_step(h, z);
    return (unsigned long long)(long long)(int32_t)h;
}

// cell-centre of an awareness cell (fill loop map_awareness.cpp:57-62) moved to world by T_wa (pure translation:
// map_awareness.cpp:184, map_local.cpp:151,180)
__device__ __forceinline__ void mlm_cell_center_w(const MlmDev &P, const double t_wa[3], int rho, int phi, int z,
                                                  double &wx, double &wy, double &wz) {
    const double center_z = P.z_border_min + (P.dZ / 2) + (z * P.dZ);
    const double center_rho = P.dRho / 2 + (rho * P.dRho);
    wx = center_rho * P.cos_phi[phi] + t_wa[0];
    wy = center_rho * P.sin_phi[phi] + t_wa[1];
    wz = center_z + t_wa[2];
}

// get_global_idx / get_subbox_id, map_local.h:148-152,167-173: two independent divisions per axis; a cell
// coordinate outside [0,n) maps to id 0 (operator[] default-inserts in the reference).
__device__ __forceinline__ void mlm_voxel_of(const MlmDev &P, double x, double y, double z, int &gx, int &gy,
                                             int &gz, int &cid, int *c3 = nullptr) {
    bool exact = true;
    const double qgx = mlm_quot(x, P.inv_d_glb, exact), qgy = mlm_quot(y, P.inv_d_glb, exact), qgz = mlm_quot(z, P.inv_d_glb, exact);
    const double qsx = mlm_quot(x, P.inv_d_sub, exact), qsy = mlm_quot(y, P.inv_d_sub, exact), qsz = mlm_quot(z, P.inv_d_sub, exact);
    gx = (int)floor(qgx);
    gy = (int)floor(qgy);
    gz = (int)floor(qgz);
    int cx = (int)(floor(qsx) - gx * P.n);
    int cy = (int)(floor(qsy) - gy * P.n);
    int cz = (int)(floor(qsz) - gz * P.n);
    if (!__all(exact)) { // rare: the whole wave skips the six divisions otherwise
        if (!exact) {
            gx = mlm_cvt_int(floor(x / P.d_glb));
            gy = mlm_cvt_int(floor(y / P.d_glb));
            gz = mlm_cvt_int(floor(z / P.d_glb));
            cx = mlm_cvt_int(floor(x / P.d_sub) - gx * P.n);
            cy = mlm_cvt_int(floor(y / P.d_sub) - gy * P.n);
            cz = mlm_cvt_int(floor(z / P.d_sub) - gz * P.n);
        }
    }
    if (cx < 0 || cy < 0 || cz < 0 || cx >= P.n || cy >= P.n || cz >= P.n) {
        cid = 0;
        cx = cy = cz = 0;
    } else {
        cid = cz * P.n * P.n + cy * P.n + cx;
    }
    if (c3) { // cell coordinates inside the block, consistent with cid
        c3[0] = cx;
        c3[1] = cy;
        c3[2] = cz;
    }
}

// One axis of mlm_voxel_of (the axes are independent: a quotient that passes mlm_quot's check has the integer part of the
// true division, so recomputing only the failing axis gives what recomputing all six gives)
__device__ __forceinline__ void mlm_voxel_axis(const MlmDev &P, double x, int &g, int &c) {
    bool exact = true;
    const double qg = mlm_quot(x, P.inv_d_glb, exact), qs = mlm_quot(x, P.inv_d_sub, exact);
    g = (int)floor(qg);
    c = (int)(floor(qs) - g * P.n);
    if (!exact) {
        g = mlm_cvt_int(floor(x / P.d_glb));
        c = mlm_cvt_int(floor(x / P.d_sub) - g * P.n);
    }
}
// ... and their combination: block indices, cell id and in-block coordinates from the three (block index, cell coordinate) pairs
__device__ __forceinline__ void mlm_voxel_combine(const MlmDev &P, int &cx, int &cy, int &cz, int &cid) {
    if (cx < 0 || cy < 0 || cz < 0 || cx >= P.n || cy >= P.n || cz >= P.n) {
        cid = 0;
        cx = cy = cz = 0;
    } else {
        cid = cz * P.n * P.n + cy * P.n + cx;
    }
}

// ---- hashed block table -------------------------------------------------------------------------------------
// key: three 21-bit biased block indices.  Blocks are never freed (the reference only ever "collapses" them).
__device__ __forceinline__ unsigned long long mlm_pack_key(int gx, int gy, int gz) {
    return ((unsigned long long)((uint32_t)(gx + (1 << 20)) & 0x1FFFFFu) << 42) |
           ((unsigned long long)((uint32_t)(gy + (1 << 20)) & 0x1FFFFFu) << 21) |
           (unsigned long long)((uint32_t)(gz + (1 << 20)) & 0x1FFFFFu);
}
__device__ __forceinline__ uint32_t mlm_mix(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}
__device__ __forceinline__ void mlm_unpack_key(unsigned long long key, int &gx, int &gy, int &gz) {
    gx = (int)((key >> 42) & 0x1FFFFFu) - (1 << 20);
    gy = (int)((key >> 21) & 0x1FFFFFu) - (1 << 20);
    gz = (int)(key & 0x1FFFFFu) - (1 << 20);
}
// lookup only; -1 if absent
__device__ __forceinline__ int mlm_block_find_k(const MlmDev &P, unsigned long long key) {
    // (key and slot of a probe position are fetched together: one round trip when the first probe decides, the usual case
    // with a table four times the pool)
    uint32_t h = mlm_mix(key) & P.ht_mask;
    for (uint32_t probe = 0; probe <= P.ht_mask; ++probe) {
        const unsigned long long k = mlm_gp(P.ht_keys)[h];
        const int s = mlm_gp(P.ht_slot)[h];
        asm volatile("" ::"v"(s)); // (keeps the slot's load next to the key's: the compiler otherwise sinks it behind the comparison — a second trip)
        if (k == key) return s;
        if (k == MLM_HT_EMPTY) return -1;
        h = (h + 1) & P.ht_mask;
    }
    return -1;
}
__device__ __forceinline__ int mlm_block_find(const MlmDev &P, int gx, int gy, int gz) {
    // (a block index beyond the key's 21 bits per axis — a query a million blocks from the origin — names no block of the map: absent,
    // like observed_group_map.find; it must not alias a block inside the range)
    if (((unsigned int)gx + (1u << 20)) >> 21 || ((unsigned int)gy + (1u << 20)) >> 21 || ((unsigned int)gz + (1u << 20)) >> 21) return -1;
    return mlm_block_find_k(P, mlm_pack_key(gx, gy, gz));
}
// allocate_ram (map_local.h:215-231): find or create.  One loop whose divergent arms reconverge every
// iteration, so a lane waiting for another lane's slot publication can never starve it.
__device__ __forceinline__ int mlm_block_find_or_insert_k(const MlmDev &P, unsigned long long key) {
    uint32_t h = mlm_mix(key) & P.ht_mask;
    int slot = -1;
    bool done = false;
    uint32_t probes = 0;
    while (!done) {
        const unsigned long long k = __hip_atomic_load(&P.ht_keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == key) {
            const int s = __hip_atomic_load(&P.ht_slot[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s != -1) {
                slot = s;
                done = true;
            }
        } else if (k == MLM_HT_EMPTY) {
            const unsigned long long prev = atomicCAS(&P.ht_keys[h], MLM_HT_EMPTY, key);
            if (prev == MLM_HT_EMPTY) {
                int s = (int)atomicAdd(&P.g->n_blocks, 1u);
                if (s >= P.max_blocks) {
                    atomicOr(&P.g->err, 1u);
                    s = -2; // published as "pool full"
                } else {
                    int gx, gy, gz;
                    mlm_unpack_key(key, gx, gy, gz);
                    P.block_keys[3 * s + 0] = gx;
                    P.block_keys[3 * s + 1] = gy;
                    P.block_keys[3 * s + 2] = gz;
                }
                __hip_atomic_store(&P.ht_slot[h], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                slot = s;
                done = true;
            }
            // else: someone else claimed this entry — re-examine it next iteration
        } else {
            h = (h + 1) & P.ht_mask;
            if (++probes > P.ht_mask) {
                atomicOr(&P.g->err, 1u);
                slot = -2;
                done = true;
            }
        }
    }
    return slot;
}
// block of a packed key: plain-load lookup first (blocks almost always exist already), CAS insert otherwise
__device__ __forceinline__ int mlm_block_slot(const MlmDev &P, unsigned long long key) {
    int slot = mlm_block_find_k(P, key);
    if (slot == -1) slot = mlm_block_find_or_insert_k(P, key);
    return slot;
}

// wave64-aggregated append: returns this lane's position (only meaningful where `has`)
__device__ __forceinline__ unsigned int mlm_wave_append(unsigned int *counter, bool has) {
    const unsigned long long mask = __ballot(has);
    if (mask == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(mask));
    base = mlm_readlane(base, leader);
    return base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
}

// Block-aggregated append into one of MLM_RAY_LISTS sub-lists (chosen by blockIdx, so that blocks of different
// XCDs hit different counter cache lines): one global atomic per block.  Every thread of the block must call it
// at the same point.  s_cnt needs blockDim/64 entries.  Returns the position inside sub-list (blockIdx.x & 7).
__device__ __forceinline__ unsigned int mlm_block_append(unsigned int (*ctr)[32], bool flag, unsigned int *s_cnt,
                                                         unsigned int *s_base) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_cnt[wid] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int w = 0; w < nw; ++w) tot += s_cnt[w];
        *s_base = tot ? atomicAdd(&ctr[blockIdx.x & 7][0], tot) : 0u;
    }
    __syncthreads();
    unsigned int pos = *s_base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) pos += s_cnt[w];
    return pos;
}
