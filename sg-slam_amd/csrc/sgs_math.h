// Host+device fp64 rigid-body maths for the SG-SLAM hot path (the subset of Sophus SE3d / Eigen the
// reference touches: F/Registration.cpp:87,112,169; F/Deskew.cpp:41-44; F/Threshold.cpp:30;
// P/SemGraphSLAM.cpp:61-63,79,157,165).  Compiled with -ffp-contract=off so the same expression
// order yields the same bits on host and device.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#define SGS_HD __host__ __device__ inline

namespace sgs {

struct D3 {
    double x, y, z;
};
SGS_HD D3 d3(double x, double y, double z) { return D3{x, y, z}; }
SGS_HD D3 add(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SGS_HD D3 sub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SGS_HD D3 scale(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
SGS_HD D3 crs(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// 3-vector reductions associate as x + (y + z), the order Eigen's unrolled redux uses.
SGS_HD double sq3(D3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }

// rigid transform: unit quaternion (w,x,y,z) + translation
struct Rigid {
    double qw, qx, qy, qz;
    D3 t;
};
SGS_HD Rigid rigid_identity() { return Rigid{1, 0, 0, 0, D3{0, 0, 0}}; }
SGS_HD Rigid rigid_from_pose7(const double *p) { return Rigid{p[3], p[0], p[1], p[2], D3{p[4], p[5], p[6]}}; }
SGS_HD void rigid_to_pose7(const Rigid &T, double *p) {
    p[0] = T.qx; p[1] = T.qy; p[2] = T.qz; p[3] = T.qw;
    p[4] = T.t.x; p[5] = T.t.y; p[6] = T.t.z;
}
SGS_HD D3 rotate(const Rigid &T, D3 p) {
    const D3 qv{T.qx, T.qy, T.qz};
    D3 uv = crs(qv, p);
    uv = add(uv, uv);
    return add(add(p, scale(T.qw, uv)), crs(qv, uv));
}
SGS_HD D3 apply(const Rigid &T, D3 p) { return add(rotate(T, p), T.t); }
SGS_HD Rigid compose(const Rigid &a, const Rigid &b) {
    Rigid r;
    r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
    r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
    r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
    r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
    const double n = sqrt(r.qw * r.qw + r.qx * r.qx + r.qy * r.qy + r.qz * r.qz);
    r.qw /= n; r.qx /= n; r.qy /= n; r.qz /= n;
    r.t = add(a.t, rotate(a, b.t));
    return r;
}
SGS_HD Rigid invert(const Rigid &T) {
    Rigid r{T.qw, -T.qx, -T.qy, -T.qz, D3{0, 0, 0}};
    const D3 rt = rotate(r, T.t);
    r.t = D3{-rt.x, -rt.y, -rt.z};
    return r;
}

constexpr double kTiny = 1e-10;

// exp: tangent [v; w] -> rigid
SGS_HD Rigid rigid_exp(const double *a) {
    const D3 v{a[0], a[1], a[2]}, w{a[3], a[4], a[5]};
    const double th2 = sq3(w);
    double theta, im, re;
    if (th2 < kTiny * kTiny) {
        theta = 0;
        const double th4 = th2 * th2;
        im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        re = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        const double h = 0.5 * theta;
        im = sin(h) / theta;
        re = cos(h);
    }
    Rigid T;
    T.qw = re; T.qx = im * w.x; T.qy = im * w.y; T.qz = im * w.z;
    const double n = sqrt(T.qw * T.qw + T.qx * T.qx + T.qy * T.qy + T.qz * T.qz);
    T.qw /= n; T.qx /= n; T.qy /= n; T.qz /= n;
    // t = V v with V = I + c1 W + c2 W^2 (W = hat(w)); for theta -> 0, V = R
    if (theta < kTiny) {
        T.t = rotate(T, v);
    } else {
        const double c1 = (1.0 - cos(theta)) / th2;
        const double c2 = (theta - sin(theta)) / (th2 * theta);
        const D3 wv = crs(w, v);
        const D3 wwv = crs(w, wv);
        T.t = add(v, add(scale(c1, wv), scale(c2, wwv)));
    }
    return T;
}
// log: rigid -> tangent [v; w]
SGS_HD void rigid_log(const Rigid &T, double *out) {
    const double n2 = T.qx * T.qx + (T.qy * T.qy + T.qz * T.qz);
    double k, theta;
    if (n2 < kTiny * kTiny) {
        k = 2.0 / T.qw - (2.0 / 3.0) * n2 / (T.qw * T.qw * T.qw);
        theta = 2.0 * n2 / T.qw;
    } else {
        const double n = sqrt(n2);
        const double at = (T.qw < 0) ? atan2(-n, -T.qw) : atan2(n, T.qw);
        k = 2.0 * at / n;
        theta = k * n;
    }
    const D3 w{k * T.qx, k * T.qy, k * T.qz};
    double c;
    if (fabs(theta) < kTiny) {
        c = 1.0 / 12.0;
    } else {
        const double h = 0.5 * theta;
        c = (1.0 - theta * cos(h) / (2.0 * sin(h))) / (theta * theta);
    }
    // v = (I - W/2 + c W^2) t
    const D3 wt = crs(w, T.t);
    const D3 wwt = crs(w, wt);
    const D3 v = add(sub(T.t, scale(0.5, wt)), scale(c, wwt));
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
    out[3] = w.x; out[4] = w.y; out[5] = w.z;
}
SGS_HD double rigid_rotation_angle(const Rigid &T) {
    const double n = sqrt(T.qx * T.qx + (T.qy * T.qy + T.qz * T.qz));
    return 2.0 * atan2(n, fabs(T.qw));
}

// Symmetric 6x6 solve A x = b by LDL^T with diagonal pivoting; pivots at or below DBL_MIN are
// treated as zero (pseudo-inverse), so an all-zero system yields x = 0 as Eigen's LDLT does
// (F/Registration.cpp:111).  A is overwritten.  Every index is a compile-time constant (the pivot
// is dispatched through if-chains over constant cases), so the 6x6 system lives in registers on the
// GPU instead of in scratch memory: a runtime-indexed private array costs a memory round trip per
// access.  Each column is scaled by ONE reciprocal of its pivot (an fp64 divide is a ~30-instruction
// dependent chain on CDNA and this runs on a single lane); that differs from Eigen's per-element
// divide by an ulp, far inside the 1e-4 m / 1e-5 rad pose bar.
namespace detail {
template <int K, int P>
SGS_HD void sym_swap(double (&A)[6][6]) {
#pragma unroll
    for (int j = 0; j < K; ++j) { const double t = A[K][j]; A[K][j] = A[P][j]; A[P][j] = t; }
#pragma unroll
    for (int i = P + 1; i < 6; ++i) { const double t = A[i][K]; A[i][K] = A[i][P]; A[i][P] = t; }
    { const double t = A[K][K]; A[K][K] = A[P][P]; A[P][P] = t; }
#pragma unroll
    for (int i = K + 1; i < P; ++i) { const double t = A[i][K]; A[i][K] = A[P][i]; A[P][i] = t; }
}
template <int K>
SGS_HD void pivot_swap(double (&A)[6][6], int p) {
    if constexpr (K + 1 < 6) { if (p == K + 1) sym_swap<K, K + 1>(A); }
    if constexpr (K + 2 < 6) { if (p == K + 2) sym_swap<K, K + 2>(A); }
    if constexpr (K + 3 < 6) { if (p == K + 3) sym_swap<K, K + 3>(A); }
    if constexpr (K + 4 < 6) { if (p == K + 4) sym_swap<K, K + 4>(A); }
    if constexpr (K + 5 < 6) { if (p == K + 5) sym_swap<K, K + 5>(A); }
}
// swap y[K] <-> y[p] (p >= K) as branch-free selects over constant indices: an if-chain here gets
// its stores sunk into one store through a phi of addresses, which turns y into a scratch array
template <int K>
SGS_HD void vec_swap(double (&y)[6], int p) {
#pragma unroll
    for (int j = K + 1; j < 6; ++j) {
        const bool hit = (p == j);
        const double yk = y[K], yj = y[j];
        y[K] = hit ? yj : yk;
        y[j] = hit ? yk : yj;
    }
}
template <int K>
SGS_HD void ldlt_column(double (&A)[6][6], int (&tr)[6], double (&inv)[6], bool &zero) {
    tr[K] = K;
    inv[K] = 0.0;
    if (zero) return;
    int p = K;
    double best = fabs(A[K][K]);
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
        const double v = fabs(A[i][i]);
        if (v > best) { best = v; p = i; }
    }
    tr[K] = p;
    pivot_swap<K>(A, p);
    double tmp[6];
#pragma unroll
    for (int j = 0; j < K; ++j) tmp[j] = A[j][j] * A[K][j];
    if constexpr (K > 0) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) s += A[K][j] * tmp[j];
        A[K][K] -= s;
#pragma unroll
        for (int i = K + 1; i < 6; ++i) {
            double s2 = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) s2 += A[i][j] * tmp[j];
            A[i][K] -= s2;
        }
    }
    const double akk = A[K][K];
    if (K == 0 && !(fabs(akk) > 0.0)) { zero = true; return; }
    if (fabs(akk) > 0.0) {
        const double r = 1.0 / akk;
        if (fabs(akk) > 2.2250738585072014e-308) inv[K] = r;
#pragma unroll
        for (int i = K + 1; i < 6; ++i) A[i][K] *= r;
    }
}
}  // namespace detail

SGS_HD void solve6_static(double (&A)[6][6], const double (&b)[6], double (&x)[6]) {
    int tr[6];
    double inv[6];
    bool zero = false;
    detail::ldlt_column<0>(A, tr, inv, zero);
    detail::ldlt_column<1>(A, tr, inv, zero);
    detail::ldlt_column<2>(A, tr, inv, zero);
    detail::ldlt_column<3>(A, tr, inv, zero);
    detail::ldlt_column<4>(A, tr, inv, zero);
    detail::ldlt_column<5>(A, tr, inv, zero);
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = b[i];
    detail::vec_swap<0>(y, tr[0]); detail::vec_swap<1>(y, tr[1]); detail::vec_swap<2>(y, tr[2]);
    detail::vec_swap<3>(y, tr[3]); detail::vec_swap<4>(y, tr[4]);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] *= inv[i];  // inv == 0 where |D_i| <= DBL_MIN (pseudo-inverse)
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) y[i] -= A[j][i] * y[j];
    detail::vec_swap<4>(y, tr[4]); detail::vec_swap<3>(y, tr[3]); detail::vec_swap<2>(y, tr[2]);
    detail::vec_swap<1>(y, tr[1]); detail::vec_swap<0>(y, tr[0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// 1/sqrt(s) for s = |q|^2 of a product of unit quaternions (s = 1 + e, |e| ~ 1e-15): the series
// 1 - e/2 + 3e^2/8 is exact to double precision there; anything farther from 1 takes the sqrt.
SGS_HD double inv_sqrt_near1(double s) {
    const double e = s - 1.0;
    if (fabs(e) < 1e-5) return 1.0 + e * (-0.5 + e * (0.375 - 0.3125 * e));
    return 1.0 / sqrt(s);
}

// exp for the Gauss-Newton updates on the device.  The updates are small (|omega| well below 0.1 rad
// after the first iteration), so below |omega|^2 = 1e-2 every trigonometric coefficient is a short
// even power series in theta (no sqrt, no sincos, no cancellation; truncation < 1e-19 relative);
// above it, one sincos of theta/2 plus double-angle identities.  Differs from rigid_exp by rounding.
SGS_HD Rigid rigid_exp_small(D3 v, D3 w) {
    const double th2 = sq3(w);
    double im, re, c1, c2;
    if (th2 < 1e-2) {
        const double h2 = 0.25 * th2;
        // sin(h)/theta = 0.5 * sin(h)/h ; cos(h)
        im = 0.5 * (1.0 + h2 * (-1.0 / 6 + h2 * (1.0 / 120 + h2 * (-1.0 / 5040 + h2 * (1.0 / 362880 +
                   h2 * (-1.0 / 39916800 + h2 * (1.0 / 6227020800.0)))))));
        re = 1.0 + h2 * (-1.0 / 2 + h2 * (1.0 / 24 + h2 * (-1.0 / 720 + h2 * (1.0 / 40320 +
                   h2 * (-1.0 / 3628800 + h2 * (1.0 / 479001600.0))))));
        // (1 - cos t)/t^2 ; (t - sin t)/t^3
        c1 = 1.0 / 2 + th2 * (-1.0 / 24 + th2 * (1.0 / 720 + th2 * (-1.0 / 40320 + th2 * (1.0 / 3628800 +
                   th2 * (-1.0 / 479001600.0 + th2 * (1.0 / 87178291200.0))))));
        c2 = 1.0 / 6 + th2 * (-1.0 / 120 + th2 * (1.0 / 5040 + th2 * (-1.0 / 362880 + th2 * (1.0 / 39916800 +
                   th2 * (-1.0 / 6227020800.0 + th2 * (1.0 / 1307674368000.0))))));
    } else {
        const double theta = sqrt(th2);
        double sh, ch;
        sincos(0.5 * theta, &sh, &ch);
        im = sh / theta;
        re = ch;
        c1 = 2.0 * sh * sh / th2;                          // 1 - cos(theta) = 2 sin^2(theta/2)
        c2 = (theta - 2.0 * sh * ch) / (th2 * theta);
    }
    Rigid T;
    T.qw = re; T.qx = im * w.x; T.qy = im * w.y; T.qz = im * w.z;
    const double rn = inv_sqrt_near1(T.qw * T.qw + T.qx * T.qx + T.qy * T.qy + T.qz * T.qz);
    T.qw *= rn; T.qx *= rn; T.qy *= rn; T.qz *= rn;
    const D3 wv = crs(w, v);
    const D3 wwv = crs(w, wv);
    T.t = add(v, add(scale(c1, wv), scale(c2, wwv)));
    return T;
}
// compose() with the renormalisation as one reciprocal (device solve tail)
SGS_HD Rigid compose_fast(const Rigid &a, const Rigid &b) {
    Rigid r;
    r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
    r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
    r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
    r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
    const double rn = inv_sqrt_near1(r.qw * r.qw + r.qx * r.qx + r.qy * r.qy + r.qz * r.qz);
    r.qw *= rn; r.qx *= rn; r.qy *= rn; r.qz *= rn;
    r.t = add(a.t, rotate(a, b.t));
    return r;
}

// ---- voxel keys ----
struct K3 {
    int32_t x, y, z;
};
SGS_HD K3 voxel_of(double x, double y, double z, double voxel_size) {
    return K3{(int32_t)(x / voxel_size), (int32_t)(y / voxel_size), (int32_t)(z / voxel_size)};
}
// reference hash (core/Coreutils.h:40-45): used ONLY to reproduce tsl::robin_map's emission order
SGS_HD uint32_t ref_hash(K3 k) {
    return ((1u << 20) - 1u) & ((uint32_t)k.x * 73856093u ^ (uint32_t)k.y * 19349663u ^ (uint32_t)k.z * 83492791u);
}
constexpr int32_t kKeyBias = 1 << 20;  // keys are packed 21 bits per axis
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr uint64_t kTombKey = 0xFFFFFFFFFFFFFFFEull;
SGS_HD bool key_in_range(K3 k) {
    return k.x >= -kKeyBias && k.x < kKeyBias && k.y >= -kKeyBias && k.y < kKeyBias && k.z >= -kKeyBias && k.z < kKeyBias;
}
SGS_HD uint64_t pack_key(K3 k) {
    return ((uint64_t)(uint32_t)(k.x + kKeyBias) << 42) | ((uint64_t)(uint32_t)(k.y + kKeyBias) << 21) |
           (uint64_t)(uint32_t)(k.z + kKeyBias);
}
SGS_HD K3 unpack_key(uint64_t p) {
    return K3{(int32_t)((p >> 42) & 0x1FFFFF) - kKeyBias, (int32_t)((p >> 21) & 0x1FFFFF) - kKeyBias,
              (int32_t)(p & 0x1FFFFF) - kKeyBias};
}
// slot hash of the GPU-resident map (free choice: murmur3 finaliser on the packed key)
SGS_HD uint32_t map_hash(uint64_t p) {
    p ^= p >> 33;
    p *= 0xff51afd7ed558ccdull;
    p ^= p >> 33;
    p *= 0xc4ceb9fe1a85ec53ull;
    p ^= p >> 33;
    return (uint32_t)p;
}
// Owner of a voxel when the map is sharded by key over `world` ranks (BASELINE.json configs[4], SURVEY.md
// 8e row 2): coarse blocks of 2^b x 2^b voxel columns (z is not split), hashed to a rank.  Arithmetic shift =
// floor division, so the blocks tile negative coordinates too.
SGS_HD int shard_owner(K3 k, int block_log2, int world) {
    const uint32_t bx = (uint32_t)(k.x >> block_log2), by = (uint32_t)(k.y >> block_log2);
    return (int)(((bx * 73856093u) ^ (by * 19349663u)) % (uint32_t)world);
}
// a rank stores a voxel when it owns it or any of its neighbours (one-voxel halo): a query whose centre
// voxel the rank owns then finds all 27 neighbour voxels locally
SGS_HD bool shard_stores(K3 k, int block_log2, int world, int rank) {
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            if (shard_owner(K3{k.x + dx, k.y + dy, k.z}, block_log2, world) == rank) return true;
    return false;
}
SGS_HD double semantic_weight(int label) {  // F/Registration.hpp:21-56
    if (label == 16 || label == 18 || label == 19) return 1.2;
    if (label >= 0 && label < 20) return 1.0;
    return 0.0;
}

}  // namespace sgs
