// ORACLE (test infrastructure, NOT product code): CPU restatement of the third-party maths the
// SG-SLAM odometry hot path delegates to Eigen 3.4.90 / Sophus 1.22.11 (neither is vendored under
// /root/reference nor installed here; formulas restated from those libraries' published algorithms,
// see SURVEY.md Appendix A).  PARITY UNPINNED: the reference ships no tests / golden vectors, and
// cannot be compiled in this environment.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace sgo {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(double s, const Vec3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(const Vec3 &a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen's unrolled non-vectorised reduction of a 3-vector splits as x0 + (x1 + x2)
// (redux_novec_unroller<.., 0, 3>: HalfLength = 1).  Used by .norm()/.squaredNorm() everywhere on
// the path (F/Preprocessing.cpp:116, F/VoxelHashMap.cpp:86,112, F/Registration.cpp:102).
inline double sqnorm(const Vec3 &a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
inline double norm(const Vec3 &a) { return std::sqrt(sqnorm(a)); }
inline double dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }

struct Mat3 {
    double m[3][3];
    static Mat3 zero() {
        Mat3 r;
        std::memset(r.m, 0, sizeof(r.m));
        return r;
    }
    static Mat3 identity() {
        Mat3 r = zero();
        r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
        return r;
    }
};
inline Mat3 operator*(const Mat3 &a, const Mat3 &b) {
    Mat3 r = Mat3::zero();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline Vec3 operator*(const Mat3 &a, const Vec3 &v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
            a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
// Sophus SO3::hat
inline Mat3 hat(const Vec3 &w) {
    Mat3 r = Mat3::zero();
    r.m[0][1] = -w.z; r.m[0][2] = w.y;
    r.m[1][0] = w.z;  r.m[1][2] = -w.x;
    r.m[2][0] = -w.y; r.m[2][1] = w.x;
    return r;
}

constexpr double kEps = 1e-10;  // Sophus::Constants<double>::epsilon()

struct Quat {  // unit quaternion, (w; x,y,z)
    double w = 1, x = 0, y = 0, z = 0;
};
inline void normalize(Quat &q) {
    const double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
// Sophus SO3 group product (explicit Hamilton product, then renormalise).
inline Quat qmul(const Quat &a, const Quat &b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    normalize(r);
    return r;
}
// Sophus SO3 * point:  p + w*uv + qv x uv,  uv = 2 (qv x p)
inline Vec3 qrot(const Quat &q, const Vec3 &p) {
    const Vec3 qv{q.x, q.y, q.z};
    Vec3 uv = cross(qv, p);
    uv = uv + uv;
    return p + q.w * uv + cross(qv, uv);
}
inline Mat3 qmat(const Quat &q) {  // Eigen toRotationMatrix
    Mat3 R;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz;       R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;       R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;       R.m[2][1] = tyz + twx;       R.m[2][2] = 1 - (txx + tyy);
    return R;
}

struct SE3 {
    Quat q;
    Vec3 t;
};
inline SE3 operator*(const SE3 &a, const SE3 &b) {
    SE3 r;
    r.q = qmul(a.q, b.q);
    r.t = a.t + qrot(a.q, b.t);
    return r;
}
inline Vec3 operator*(const SE3 &T, const Vec3 &p) { return qrot(T.q, p) + T.t; }
inline SE3 inverse(const SE3 &T) {
    SE3 r;
    r.q = Quat{T.q.w, -T.q.x, -T.q.y, -T.q.z};
    const Vec3 rt = qrot(r.q, T.t);
    r.t = Vec3{-rt.x, -rt.y, -rt.z};
    return r;
}

// Sophus SO3::expAndTheta
inline Quat so3_exp(const Vec3 &omega, double *theta_out) {
    const double theta_sq = sqnorm(omega);
    double imag, real, theta;
    if (theta_sq < kEps * kEps) {
        theta = 0;
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        theta = std::sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    if (theta_out) *theta_out = theta;
    Quat q{real, imag * omega.x, imag * omega.y, imag * omega.z};
    normalize(q);
    return q;
}
// Sophus SO3::logAndTheta
inline Vec3 so3_log(const Quat &q, double *theta_out) {
    const double sq_n = q.x * q.x + (q.y * q.y + q.z * q.z);
    const double w = q.w;
    double k, theta;
    if (sq_n < kEps * kEps) {
        const double sq_w = w * w;
        k = 2.0 / w - (2.0 / 3.0) * sq_n / (w * sq_w);
        theta = 2.0 * sq_n / w;
    } else {
        const double n = std::sqrt(sq_n);
        const double a = (w < 0) ? std::atan2(-n, -w) : std::atan2(n, w);
        k = 2.0 * a / n;
        theta = k * n;
    }
    if (theta_out) *theta_out = theta;
    return {k * q.x, k * q.y, k * q.z};
}

struct Vec6 {
    double v[6] = {0, 0, 0, 0, 0, 0};
};
// Sophus SE3::exp, tangent = [upsilon; omega]
inline SE3 se3_exp(const Vec6 &a) {
    const Vec3 ups{a.v[0], a.v[1], a.v[2]}, omega{a.v[3], a.v[4], a.v[5]};
    double theta;
    SE3 T;
    T.q = so3_exp(omega, &theta);
    const Mat3 Om = hat(omega);
    const Mat3 Om2 = Om * Om;
    Mat3 V;
    if (theta < kEps) {
        V = qmat(T.q);
    } else {
        const double th2 = theta * theta;
        const double c1 = (1.0 - std::cos(theta)) / th2;
        const double c2 = (theta - std::sin(theta)) / (th2 * theta);
        V = Mat3::identity();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) V.m[i][j] += c1 * Om.m[i][j] + c2 * Om2.m[i][j];
    }
    T.t = V * ups;
    return T;
}
// Sophus SE3::log
inline Vec6 se3_log(const SE3 &T) {
    double theta;
    const Vec3 omega = so3_log(T.q, &theta);
    const Mat3 Om = hat(omega);
    const Mat3 Om2 = Om * Om;
    Mat3 Vi = Mat3::identity();
    double c;
    if (std::fabs(theta) < kEps) {
        c = 1.0 / 12.0;
    } else {
        const double half = 0.5 * theta;
        c = (1.0 - theta * std::cos(half) / (2.0 * std::sin(half))) / (theta * theta);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Vi.m[i][j] += -0.5 * Om.m[i][j] + c * Om2.m[i][j];
    const Vec3 ups = Vi * T.t;
    Vec6 r;
    r.v[0] = ups.x; r.v[1] = ups.y; r.v[2] = ups.z;
    r.v[3] = omega.x; r.v[4] = omega.y; r.v[5] = omega.z;
    return r;
}
inline double norm6(const Vec6 &a) {
    double s = 0;
    for (int i = 0; i < 6; ++i) s += a.v[i] * a.v[i];
    return std::sqrt(s);
}
// Eigen::AngleAxisd(R).angle(): 2*atan2(|q.vec|, |q.w|)
inline double rotation_angle(const Quat &q) {
    const double n = std::sqrt(q.x * q.x + (q.y * q.y + q.z * q.z));
    return 2.0 * std::atan2(n, std::fabs(q.w));
}

// Eigen::LDLT (lower, diagonal pivoting) + solve with the pseudo-inverse rule for tiny pivots.
// Used by AlignClouds: x = JTJ.ldlt().solve(-JTr)  (F/Registration.cpp:111).
inline void ldlt6_solve(const double Ain[6][6], const double b[6], double x[6]) {
    const int n = 6;
    double A[6][6];
    std::memcpy(A, Ain, sizeof(A));
    int perm[6];
    for (int i = 0; i < n; ++i) perm[i] = i;
    int transp[6];
    bool all_zero = false;
    for (int k = 0; k < n; ++k) {
        if (all_zero) { transp[k] = k; continue; }
        int p = k;
        double big = std::fabs(A[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A[i][i]) > big) { big = std::fabs(A[i][i]); p = i; }
        transp[k] = p;
        if (p != k) {
            // symmetric swap of rows/cols k and p, lower triangle storage
            for (int j = 0; j < k; ++j) std::swap(A[k][j], A[p][j]);
            for (int i = p + 1; i < n; ++i) std::swap(A[i][k], A[i][p]);
            std::swap(A[k][k], A[p][p]);
            for (int i = k + 1; i < p; ++i) std::swap(A[i][k], A[p][i]);
        }
        // A(k,k) -= sum_j L(k,j)^2 D(j);  A21 -= A20 * (D .* L(k,:))
        double temp[6];
        for (int j = 0; j < k; ++j) temp[j] = A[j][j] * A[k][j];
        if (k > 0) {
            double s = 0;
            for (int j = 0; j < k; ++j) s += A[k][j] * temp[j];
            A[k][k] -= s;
            for (int i = k + 1; i < n; ++i) {
                double s2 = 0;
                for (int j = 0; j < k; ++j) s2 += A[i][j] * temp[j];
                A[i][k] -= s2;
            }
        }
        const double akk = A[k][k];
        const bool valid = std::fabs(akk) > 0.0;
        if (k == 0 && !valid) {
            all_zero = true;
            for (int j = 1; j < n; ++j) transp[j] = j;
            continue;
        }
        if (valid)
            for (int i = k + 1; i < n; ++i) A[i][k] /= akk;
    }
    (void)perm;
    double y[6];
    for (int i = 0; i < n; ++i) y[i] = b[i];
    for (int k = 0; k < n; ++k) std::swap(y[k], y[transp[k]]);          // P b
    for (int i = 0; i < n; ++i)                                         // L^-1
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
    const double tol = std::numeric_limits<double>::min();
    for (int i = 0; i < n; ++i) {                                       // D^+ (pseudo-inverse)
        if (std::fabs(A[i][i]) > tol) y[i] /= A[i][i];
        else y[i] = 0;
    }
    for (int i = n - 1; i >= 0; --i)                                    // L^-T
        for (int j = i + 1; j < n; ++j) y[i] -= A[j][i] * y[j];
    for (int k = n - 1; k >= 0; --k) std::swap(y[k], y[transp[k]]);     // P^T
    for (int i = 0; i < n; ++i) x[i] = y[i];
}

// Symmetric eigendecomposition (Householder tridiagonalisation + implicit QL, the classic
// tred2/tql2 pair), eigenvalues ascending like Eigen::SelfAdjointEigenSolver.  Eigenvectors of
// repeated eigenvalues are NOT unique across solvers.  a: n*n row-major, in/out
// (columns = eigenvectors on return), d: eigenvalues.
inline void sym_eig(int n, std::vector<double> &a, std::vector<double> &d) {
    d.assign(n, 0.0);
    if (n == 0) return;
    std::vector<double> e(n, 0.0);
    auto A = [&](int i, int j) -> double & { return a[(size_t)i * n + j]; };
    // tred2
    for (int i = n - 1; i > 0; --i) {
        const int l = i - 1;
        double h = 0, scale = 0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += std::fabs(A(i, k));
            if (scale == 0.0) {
                e[i] = A(i, l);
            } else {
                for (int k = 0; k <= l; ++k) {
                    A(i, k) /= scale;
                    h += A(i, k) * A(i, k);
                }
                double f = A(i, l);
                double g = (f >= 0 ? -std::sqrt(h) : std::sqrt(h));
                e[i] = scale * g;
                h -= f * g;
                A(i, l) = f - g;
                f = 0;
                for (int j = 0; j <= l; ++j) {
                    A(j, i) = A(i, j) / h;
                    g = 0;
                    for (int k = 0; k <= j; ++k) g += A(j, k) * A(i, k);
                    for (int k = j + 1; k <= l; ++k) g += A(k, j) * A(i, k);
                    e[j] = g / h;
                    f += e[j] * A(i, j);
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = A(i, j);
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k) A(j, k) -= (f * e[k] + g * A(i, k));
                }
            }
        } else {
            e[i] = A(i, l);
        }
        d[i] = h;
    }
    d[0] = 0;
    e[0] = 0;
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0;
                for (int k = 0; k <= l; ++k) g += A(i, k) * A(k, j);
                for (int k = 0; k <= l; ++k) A(k, j) -= g * A(k, i);
            }
        }
        d[i] = A(i, i);
        A(i, i) = 1;
        for (int j = 0; j <= l; ++j) A(j, i) = A(i, j) = 0;
    }
    // tql2
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= std::numeric_limits<double>::epsilon() * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) break;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1, c = 1, p = 0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    e[i + 1] = (r = std::hypot(f, g));
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    for (int k = 0; k < n; ++k) {
                        f = A(k, i + 1);
                        A(k, i + 1) = s * A(k, i) + c * f;
                        A(k, i) = c * A(k, i) - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0;
            }
        } while (m != l);
    }
    // sort ascending (selection sort on columns, like Eigen)
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j)
            if (d[j] < d[k]) k = j;
        if (k != i) {
            std::swap(d[i], d[k]);
            for (int r = 0; r < n; ++r) std::swap(A(r, i), A(r, k));
        }
    }
}

}  // namespace sgo
