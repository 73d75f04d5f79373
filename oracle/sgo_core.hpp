// ORACLE (test infrastructure, NOT product code): CPU restatement, stage by stage, of the SG-SLAM
// per-scan odometry hot path.  Every function cites the reference file:line it follows
// (paths relative to /root/reference/cpp/semgraph_slam/).  fp64 throughout, "reference-shaped"
// data structures (robin-hood hash map of per-voxel point vectors, per-query candidate gather),
// OpenMP standing in for TBB exactly where the reference uses tbb::parallel_*.
// PARITY UNPINNED: the reference ships no tests/golden vectors and cannot be compiled here
// (Eigen/Sophus/TBB/tsl absent, no network).  Deliberate, documented divergences from the
// reference's undefined behaviour: (1) a query whose 27 voxels hold no point yields NO
// correspondence (reference returns an uninitialised vector, F/VoxelHashMap.cpp:83,93);
// (2) labels outside 0..25 get semantic weight 0 (reference: unordered_map::operator[] inserts 0).
#pragma once
#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sgo_math.hpp"
#include "sgo_robin.hpp"

namespace sgo {

struct Cloud {  // V3d_i  (core/Coreutils.h:34)
    std::vector<Vec3> p;
    std::vector<int> l;
    size_t size() const { return p.size(); }
};
struct P4 {  // Eigen::Vector4d (x,y,z,label)
    double x, y, z, l;
};

// ---- voxel key: C++ double->int truncation of coord / voxel_size in fp64 ----------------------
// frontend/Preprocessing.cpp:77, frontend/VoxelHashMap.cpp:55-57,167
inline Key3 voxel_key(double x, double y, double z, double voxel_size) {
    return Key3{static_cast<int32_t>(x / voxel_size), static_cast<int32_t>(y / voxel_size),
                static_cast<int32_t>(z / voxel_size)};
}

// ---- frontend/Preprocessing.cpp:107-125 PreprocessSemantic ------------------------------------
inline Cloud preprocess_semantic(const std::vector<Vec3> &frame, const std::vector<int> &label,
                                 double max_range, double min_range) {
    Cloud out;
    for (size_t i = 0; i < frame.size(); ++i) {
        const double n = norm(frame[i]);
        if (n < max_range && n > min_range && label[i] < 20) {
            out.p.push_back(frame[i]);
            out.l.push_back(label[i]);
        }
    }
    return out;
}

// ---- frontend/Preprocessing.cpp:127-137 CorrectKITTIScan --------------------------------------
// AngleAxisd(a, normalized(p x z)) * p, Rodrigues matrix as Eigen builds it
// (AngleAxis::toRotationMatrix), zero rotation vector left as is by normalized().
inline Mat3 angle_axis_matrix(double angle, const Vec3 &axis) {
    const double s = std::sin(angle), c = std::cos(angle);
    const Vec3 cos1_axis = (1.0 - c) * axis;
    Mat3 R;
    double tmp;
    tmp = cos1_axis.x * axis.y; R.m[0][1] = tmp - s * axis.z; R.m[1][0] = tmp + s * axis.z;
    tmp = cos1_axis.x * axis.z; R.m[0][2] = tmp + s * axis.y; R.m[2][0] = tmp - s * axis.y;
    tmp = cos1_axis.y * axis.z; R.m[1][2] = tmp - s * axis.x; R.m[2][1] = tmp + s * axis.x;
    R.m[0][0] = cos1_axis.x * axis.x + c;
    R.m[1][1] = cos1_axis.y * axis.y + c;
    R.m[2][2] = cos1_axis.z * axis.z + c;
    return R;
}
inline std::vector<Vec3> correct_kitti_scan(const std::vector<Vec3> &frame) {
    const double VERTICAL_ANGLE_OFFSET = (0.205 * M_PI) / 180.0;
    std::vector<Vec3> out(frame.size());
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)frame.size(); ++i) {
        const Vec3 &pt = frame[i];
        Vec3 rv = cross(pt, Vec3(0., 0., 1.));
        const double z = sqnorm(rv);
        if (z > 0) rv = rv / std::sqrt(z);  // Eigen normalized(): only when squaredNorm > 0
        out[i] = angle_axis_matrix(VERTICAL_ANGLE_OFFSET, rv) * pt;
    }
    return out;
}

// ---- frontend/Deskew.cpp:37-48 DeSkewScan -----------------------------------------------------
inline std::vector<Vec3> deskew_scan(const std::vector<Vec3> &frame, const std::vector<double> &ts,
                                     const SE3 &start_pose, const SE3 &finish_pose) {
    const Vec6 delta = se3_log(inverse(start_pose) * finish_pose);
    std::vector<Vec3> out(frame.size());
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)frame.size(); ++i) {
        Vec6 a;
        for (int k = 0; k < 6; ++k) a.v[k] = (ts[i] - 0.5) * delta.v[k];
        out[i] = se3_exp(a) * frame[i];
    }
    return out;
}

// ---- frontend/Preprocessing.cpp:70-94 VoxelDownsampleSemantic ---------------------------------
// order_mode 0 = robin (reference-faithful: emit in tsl::robin_map bucket order),
//            1 = canonical (ascending first-occurrence index; a debugging aid, not the reference).
struct DownsampleInfo {
    int64_t buckets = 0, max_dist = 0, grow_by_probe = 0, wrapped = 0;
};
inline Cloud voxel_downsample_semantic(const Cloud &frame, double voxel_size, int order_mode,
                                       std::vector<int> *index_out, DownsampleInfo *info,
                                       int64_t probe_limit = 4096) {
    RobinMap<int> grid(probe_limit);
    grid.reserve(frame.size());
    for (int i = 0; i < (int)frame.size(); ++i) {
        const Key3 v = voxel_key(frame.p[i].x, frame.p[i].y, frame.p[i].z, voxel_size);
        if (grid.contains(v)) continue;
        grid.insert(v, i);
    }
    std::vector<int> idx;
    idx.reserve(grid.size());
    for (const auto &b : grid.buckets())
        if (b.dist >= 0) idx.push_back(b.value);
    if (order_mode == 1) std::sort(idx.begin(), idx.end());
    Cloud out;
    out.p.reserve(idx.size());
    out.l.reserve(idx.size());
    for (int i : idx) {
        out.p.push_back(frame.p[i]);
        out.l.push_back(frame.l[i]);
    }
    if (info) {
        grid.compute_wrapped();
        info->buckets = (int64_t)grid.bucket_count();
        info->max_dist = grid.stats().max_dist;
        info->grow_by_probe = grid.stats().grow_by_probe;
        info->wrapped = grid.stats().wrapped;
    }
    if (index_out) *index_out = std::move(idx);
    return out;
}

// ---- frontend/VoxelHashMap.{hpp,cpp} ----------------------------------------------------------
struct VoxelBlock {  // VoxelHashMap.hpp:43-56
    std::vector<P4> points;
    int num_points = 0;
    void AddPoint(const P4 &pt) {
        if (points.size() < (size_t)num_points) points.push_back(pt);
    }
};

struct Correspondences {
    std::vector<P4> src, tgt;
    std::vector<int> query_index;  // oracle extra: which query produced each pair
    int64_t candidates = 0;        // oracle extra: total candidate points visited (for C-bar)
};

struct VoxelHashMap {
    double voxel_size_, max_distance_;
    int max_points_per_voxel_;
    int prune_mode_ = 0;  // 0 = pure predicate (what the GPU implements), 1 = reference erase-while-iterate
    RobinMap<VoxelBlock> map_;

    VoxelHashMap(double vs, double md, int cap)
        : voxel_size_(vs), max_distance_(md), max_points_per_voxel_(cap) {}
    void Clear() { map_.clear(); }
    bool Empty() const { return map_.empty(); }

    // VoxelHashMap.cpp:164-176
    void AddPoints(const std::vector<P4> &points) {
        for (const P4 &pt : points) {
            const Key3 v = voxel_key(pt.x, pt.y, pt.z, voxel_size_);
            const int64_t f = map_.find(v);
            if (f >= 0) {
                map_.value_at(f).AddPoint(pt);
            } else {
                VoxelBlock b;
                b.points.push_back(pt);
                b.num_points = max_points_per_voxel_;
                map_.insert(v, std::move(b));
            }
        }
    }
    // VoxelHashMap.cpp:178-186
    void RemovePointsFarFromLocation(const Vec3 &origin) {
        const double max_d2 = max_distance_ * max_distance_;
        if (prune_mode_ == 0) {
            std::vector<Key3> doomed;
            for (const auto &b : map_.buckets()) {
                if (b.dist < 0) continue;
                const P4 &f = b.value.points.front();
                if (sqnorm(Vec3(f.x, f.y, f.z) - origin) > max_d2) doomed.push_back(b.key);
            }
            for (const Key3 &k : doomed) map_.erase(k);
        } else {
            // range-for with erase(key) inside: after a backward-shift erase the iterator still
            // points at the same bucket and ++it skips whatever was shifted into it.
            const size_t nb = map_.bucket_count();
            for (size_t i = 0; i < nb; ++i) {
                const auto &b = map_.buckets()[i];
                if (b.dist < 0) continue;
                const P4 &f = b.value.points.front();
                if (sqnorm(Vec3(f.x, f.y, f.z) - origin) > max_d2) map_.erase(b.key);
            }
        }
    }
    // VoxelHashMap.cpp:146-162
    void Update(const std::vector<P4> &points, const Vec3 &origin) {
        AddPoints(points);
        RemovePointsFarFromLocation(origin);
    }
    void Update(const std::vector<P4> &points, const SE3 &pose) {
        std::vector<P4> tr(points.size());
        for (size_t i = 0; i < points.size(); ++i) {
            const Vec3 w = pose * Vec3(points[i].x, points[i].y, points[i].z);
            tr[i] = P4{w.x, w.y, w.z, points[i].l};
        }
        Update(tr, pose.t);
    }
    // VoxelHashMap.cpp:133-144
    std::vector<P4> Pointcloud() const {
        std::vector<P4> out;
        for (const auto &b : map_.buckets())
            if (b.dist >= 0)
                for (const P4 &p : b.value.points) out.push_back(p);
        return out;
    }

    // VoxelHashMap.cpp:54-94: nearest of all points in the 27 neighbouring voxels, x-outer/z-inner
    // voxel order then insertion order, strict '<' keeps the first minimum.
    bool closest_neighbor(const P4 &q, P4 *best, int64_t *ncand) const {
        const int kx = static_cast<int>(q.x / voxel_size_);
        const int ky = static_cast<int>(q.y / voxel_size_);
        const int kz = static_cast<int>(q.z / voxel_size_);
        std::vector<P4> neigh;
        neigh.reserve(27 * (size_t)max_points_per_voxel_);
        for (int i = kx - 1; i < kx + 1 + 1; ++i)
            for (int j = ky - 1; j < ky + 1 + 1; ++j)
                for (int k = kz - 1; k < kz + 1 + 1; ++k) {
                    const int64_t f = map_.find(Key3{i, j, k});
                    if (f >= 0)
                        for (const P4 &p : map_.value_at(f).points) neigh.push_back(p);
                }
        double best_d2 = std::numeric_limits<double>::max();
        bool found = false;
        for (const P4 &n : neigh) {
            const double d2 = sqnorm(Vec3(n.x, n.y, n.z) - Vec3(q.x, q.y, q.z));
            if (d2 < best_d2) {
                *best = n;
                best_d2 = d2;
                found = true;
            }
        }
        if (ncand) *ncand = (int64_t)neigh.size();
        return found;
    }
    // VoxelHashMap.cpp:51-131 (tbb::parallel_reduce with order-preserving concatenation)
    Correspondences GetCorrespondences(const std::vector<P4> &points, double max_dist) const {
        const long n = (long)points.size();
        const int nt = omp_get_max_threads();
        std::vector<Correspondences> part((size_t)nt);
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const long lo = n * t / nt, hi = n * (t + 1) / nt;
            Correspondences &c = part[(size_t)t];
            for (long i = lo; i < hi; ++i) {
                P4 best;
                int64_t nc = 0;
                const bool found = closest_neighbor(points[(size_t)i], &best, &nc);
                c.candidates += nc;
                if (!found) continue;  // divergence (1), see header
                const Vec3 d = Vec3(best.x, best.y, best.z) -
                               Vec3(points[(size_t)i].x, points[(size_t)i].y, points[(size_t)i].z);
                if (norm(d) < max_dist) {
                    c.src.push_back(points[(size_t)i]);
                    c.tgt.push_back(best);
                    c.query_index.push_back((int)i);
                }
            }
        }
        Correspondences out;
        for (auto &c : part) {
            out.src.insert(out.src.end(), c.src.begin(), c.src.end());
            out.tgt.insert(out.tgt.end(), c.tgt.begin(), c.tgt.end());
            out.query_index.insert(out.query_index.end(), c.query_index.begin(), c.query_index.end());
            out.candidates += c.candidates;
        }
        return out;
    }
};

// ---- frontend/Registration.hpp:21-56 weights_semantic -----------------------------------------
inline double semantic_weight(int label) {
    if (label < 0 || label > 25) return 0.0;  // divergence (2), see header
    if (label == 16 || label == 18 || label == 19) return 1.2;
    if (label >= 20) return 0.0;
    return 1.0;
}

struct LinearSystem {
    double JTJ[6][6];
    double JTr[6];
    LinearSystem() {
        std::memset(JTJ, 0, sizeof(JTJ));
        std::memset(JTr, 0, sizeof(JTr));
    }
};

// ---- frontend/Registration.cpp:71-113 AlignClouds (the reduction part) ------------------------
// J = [I | -hat(s)], w = th^2/(th+|r|^2)^2, JTJ += J^T w J, JTr += J^T w r * w_sem(label).
// plain = the loop-closure variant (loopclosure/FastIcp.cpp:137-170): no robust kernel, no class weight.
inline LinearSystem build_linear_system(const std::vector<P4> &src, const std::vector<P4> &tgt,
                                        double th, bool plain = false) {
    const long n = (long)src.size();
    const int nt = omp_get_max_threads();
    std::vector<LinearSystem> part((size_t)nt);
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const long lo = n * t / nt, hi = n * (t + 1) / nt;
        LinearSystem &L = part[(size_t)t];
        for (long i = lo; i < hi; ++i) {
            const Vec3 s(src[(size_t)i].x, src[(size_t)i].y, src[(size_t)i].z);
            const Vec3 tg(tgt[(size_t)i].x, tgt[(size_t)i].y, tgt[(size_t)i].z);
            const Vec3 r = s - tg;
            const int label = (int)src[(size_t)i].l;
            double J[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
            const Mat3 H = hat(s);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) J[a][3 + b] = -1.0 * H.m[a][b];
            const double r2 = sqnorm(r);
            const double w = plain ? 1.0 : (th * th) / ((th + r2) * (th + r2));
            const double ws = plain ? 1.0 : semantic_weight(label);
            const double rv[3] = {r.x, r.y, r.z};
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) {
                    double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += (J[k][a] * w) * J[k][b];
                    L.JTJ[a][b] += acc;
                }
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += (J[k][a] * w) * rv[k];
                L.JTr[a] += acc * ws;
            }
        }
    }
    LinearSystem out;
    for (const auto &L : part)
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) out.JTJ[a][b] += L.JTJ[a][b];
            out.JTr[a] += L.JTr[a];
        }
    return out;
}
inline SE3 solve_linear_system(const LinearSystem &L) {
    double rhs[6], x[6];
    for (int i = 0; i < 6; ++i) rhs[i] = -L.JTr[i];
    ldlt6_solve(L.JTJ, rhs, x);
    Vec6 v;
    for (int i = 0; i < 6; ++i) v.v[i] = x[i];
    return se3_exp(v);
}

// ---- frontend/Registration.cpp:147-177 RegisterFrameSemantic ----------------------------------
struct IcpInfo {
    int iterations = 0;
    int64_t candidates = 0;       // summed over iterations
    int64_t correspondences = 0;  // summed over iterations
};
inline SE3 register_frame_semantic(const std::vector<P4> &frame, const VoxelHashMap &map,
                                   const SE3 &initial_guess, double max_corr, double kernel,
                                   IcpInfo *info, bool plain = false) {
    if (map.Empty()) return initial_guess;
    std::vector<P4> source = frame;
    auto transform = [](const SE3 &T, std::vector<P4> &pts) {
        for (auto &p : pts) {
            const Vec3 w = T * Vec3(p.x, p.y, p.z);
            p.x = w.x; p.y = w.y; p.z = w.z;
        }
    };
    transform(initial_guess, source);
    SE3 T_icp;
    for (int j = 0; j < 500; ++j) {
        const Correspondences c = map.GetCorrespondences(source, max_corr);
        const SE3 est = solve_linear_system(build_linear_system(c.src, c.tgt, kernel, plain));
        transform(est, source);
        T_icp = est * T_icp;
        if (info) {
            info->iterations = j + 1;
            info->candidates += c.candidates;
            info->correspondences += (int64_t)c.src.size();
        }
        if (norm6(se3_log(est)) < 0.0001) break;
    }
    return T_icp * initial_guess;
}

// ---- loopclosure/FastIcp.{hpp,cpp}: loop-pair refinement ----------------------
// A private copy of the map + ICP above with four differences: the voxel size is held in a FLOAT
// (FastIcp.hpp:91, so keys are trunc(p / (double)(float)size), FastIcp.cpp:35,49-51), the per-voxel cap is
// the constructor's (5 at the only call site, LoopClosure.cpp:69), AlignClouds has no robust kernel and no
// class weight (FastIcp.cpp:146-152), and the gate is the caller's 0.5 m (FastIcp.cpp:215).  get_trans
// (FastIcp.cpp:205-220): map <- cloud B, register cloud A from the initial transform.
inline SE3 fast_icp_get_trans(const std::vector<P4> &cloud_a, const std::vector<P4> &cloud_b, const SE3 &init,
                              float voxel_size_map, int max_points_per_voxel, double max_corr, IcpInfo *info) {
    VoxelHashMap map((double)voxel_size_map, 0.0, max_points_per_voxel);
    map.AddPoints(cloud_b);
    return register_frame_semantic(cloud_a, map, init, max_corr, 0.0, info, true);
}

// ---- frontend/Threshold.{hpp,cpp} -------------------------------------------------------------
struct AdaptiveThreshold {
    double initial_threshold_, min_motion_th_, max_range_;
    double model_error_sse2_ = 0;
    int num_samples_ = 0;
    SE3 model_deviation_;
    AdaptiveThreshold(double it, double mm, double mr)
        : initial_threshold_(it), min_motion_th_(mm), max_range_(mr) {}
    void UpdateModelDeviation(const SE3 &d) { model_deviation_ = d; }
    double ComputeThreshold() {  // Threshold.cpp:29-50
        const double theta = rotation_angle(model_deviation_.q);
        const double delta_rot = 2.0 * max_range_ * std::sin(theta / 2.0);
        const double delta_trans = norm(model_deviation_.t);
        const double model_error = delta_trans + delta_rot;
        if (model_error > min_motion_th_) {
            model_error_sse2_ += model_error * model_error;
            num_samples_++;
        }
        if (num_samples_ < 1) return initial_threshold_;
        return std::sqrt(model_error_sse2_ / num_samples_);
    }
};

// ---- frontend/SemanticCluster.{hpp,cpp} -------------------------------------------------------
struct Bbox {  // core/Coreutils.h:48-56
    Vec3 center, dimension;
    double theta = 0.0;
    int label = -1;
    double score = 0.0;
    int points_num = 0;
};
struct PointAPR {
    double azimuth, polar_angle, range;
};
struct SemanticCluster {
    double deltaA_, deltaR_, deltaP_;
    double min_range_ = std::numeric_limits<double>::max();
    double max_range_ = std::numeric_limits<double>::min();
    double min_azimuth_ = -24.8 * M_PI / 180;
    double max_azimuth_ = 2 * M_PI / 180;
    int length_ = 0, width_ = 0, height_ = 0;
    SemanticCluster(double dA, double dR, double dP) : deltaA_(dA), deltaR_(dR), deltaP_(dP) {}

    static double polar_angle_cal(double x, double y) {  // SemanticCluster.cpp:10-20
        double t = 0;
        if (x == 0 && y == 0) t = 0;
        else if (y >= 0) t = std::atan2(y, x);
        else if (y < 0) t = std::atan2(y, x) + 2 * M_PI;
        return t;
    }
    std::vector<PointAPR> calculateAPR(const std::vector<Vec3> &cloud) {  // :22-42
        std::vector<PointAPR> v;
        v.reserve(cloud.size());
        for (size_t i = 0; i < cloud.size(); ++i) {
            PointAPR par;
            par.polar_angle = polar_angle_cal(cloud[i].x, cloud[i].y);
            par.range = std::sqrt(cloud[i].x * cloud[i].x + cloud[i].y * cloud[i].y);
            par.azimuth = std::atan2(cloud[i].z, par.range);
            if (par.range < min_range_) min_range_ = par.range;
            if (par.range > max_range_) max_range_ = par.range;
            v.push_back(par);
        }
        length_ = int((max_range_ - min_range_) / deltaR_) + 1;
        width_ = int(std::round(360 / deltaP_));
        height_ = int(((max_azimuth_ - min_azimuth_) * 180 / M_PI) / deltaA_) + 1;
        return v;
    }
    void cell_of(const PointAPR &a, int *pol, int *rng, int *azi) const {  // :50-52 / :115-118
        *azi = int(((a.azimuth - min_azimuth_) * 180 / M_PI) / deltaA_);
        *pol = int(a.polar_angle * 180 / M_PI / deltaP_);
        *rng = int((a.range - min_range_) / deltaR_);
    }
    std::unordered_map<int, std::vector<int>> build_hash_table(const std::vector<PointAPR> &vapr) {
        std::unordered_map<int, std::vector<int>> m;  // :44-70
        for (int i = 0; i < (int)vapr.size(); ++i) {
            int pol, rng, azi;
            cell_of(vapr[(size_t)i], &pol, &rng, &azi);
            const int voxel_index = (pol * length_ + rng) + azi * length_ * width_;
            m[voxel_index].push_back(i);
        }
        return m;
    }
    void find_neighbors(int polar, int range, int azimuth, std::vector<int> &out) const {  // :72-99
        for (int z = azimuth - 1; z <= azimuth + 1; z++) {
            if (z < 0 || z > (height_ - 1)) continue;
            for (int y = range - 1; y <= range + 1; y++) {
                if (y < 0 || y > (length_ - 1)) continue;
                for (int x = polar - 1; x <= polar + 1; x++) {
                    int px = x;
                    if (x < 0) px = width_ - 1;
                    if (x > (width_ - 1)) px = 0;
                    out.push_back((px * length_ + y) + z * length_ * width_);
                }
            }
        }
    }
    // :109-174, literal (including the O(n) relabel in mergeClusters :101-107)
    std::vector<int> cluster(const std::unordered_map<int, std::vector<int>> &map_in,
                             const std::vector<PointAPR> &vapr) const {
        int current_cluster = 0;
        std::vector<int> ci(vapr.size(), -1);
        for (size_t i = 0; i < vapr.size(); ++i) {
            if (ci[i] != -1) continue;
            int pol, rng, azi;
            cell_of(vapr[i], &pol, &rng, &azi);
            const int voxel_index = (pol * length_ + rng) + azi * length_ * width_;
            std::vector<int> neigh;
            if (map_in.find(voxel_index) != map_in.end()) {
                std::vector<int> nid;
                find_neighbors(pol, rng, azi, nid);
                for (int id : nid) {
                    auto it = map_in.find(id);
                    if (it != map_in.end())
                        for (int j : it->second) neigh.push_back(j);
                }
            }
            for (size_t j = 0; j < neigh.size(); ++j) {
                const int oc = ci[i];
                const int nc = ci[(size_t)neigh[j]];
                if (oc != -1 && nc != -1) {
                    if (oc != nc)
                        for (auto &c : ci)
                            if (c == oc) c = nc;
                } else {
                    if (nc != -1) ci[i] = nc;
                    else if (oc != -1) ci[(size_t)neigh[j]] = oc;
                }
            }
            if (ci[i] == -1) {
                current_cluster++;
                ci[i] = current_cluster;
                for (int s : neigh) ci[(size_t)s] = current_cluster;
            }
        }
        return ci;
    }
};

// :205-360 ClusterPoints.  bbox output order follows libstdc++ std::unordered_map<int,...>
// iteration (this oracle is compiled against the same libstdc++ family, so it uses the container
// itself rather than emulating it).
inline std::vector<Bbox> cluster_points(const std::vector<Vec3> &frame, const std::vector<int> &label,
                                        Cloud &background, Cloud &foreground, double dA, double dR,
                                        double dP, std::vector<int> *point_cluster_ids = nullptr) {
    std::vector<std::vector<Vec3>> pc(26);
    std::vector<std::vector<int>> pc_src(26);
    for (size_t i = 0; i < frame.size(); ++i) {
        const int L = label[i];
        if (L == 1 || L == 4 || L == 5 || L == 16 || L == 18) {
            pc[(size_t)L].push_back(frame[i]);
            pc_src[(size_t)L].push_back((int)i);
            foreground.p.push_back(frame[i]);
            foreground.l.push_back(L);
        }
        if (L == 14 || L == 15 || L == 9 || L == 13) {
            background.p.push_back(frame[i]);
            background.l.push_back(L);
        }
    }
    if (point_cluster_ids) point_cluster_ids->assign(frame.size(), 0);
    std::vector<Bbox> boxes;
    for (size_t i = 0; i < 26; ++i) {
        if (pc[i].empty()) continue;
        SemanticCluster sc(dA, dR, dP);  // the i==13 branch (:257-261) is dead: 13 never routed
        const auto apr = sc.calculateAPR(pc[i]);
        const auto table = sc.build_hash_table(apr);
        const auto ci = sc.cluster(table, apr);
        if (point_cluster_ids)
            for (size_t j = 0; j < ci.size(); ++j) (*point_cluster_ids)[(size_t)pc_src[i][j]] = ci[j];
        struct label_box {
            int num;
            double d[6];
        };
        std::unordered_map<int, label_box> hist;
        for (size_t j = 0; j < ci.size(); ++j) {
            const Vec3 &p = pc[i][j];
            auto it = hist.find(ci[j]);
            if (it == hist.end()) {
                label_box &b = hist[ci[j]];
                b.d[0] = p.x; b.d[1] = p.y; b.d[2] = p.z;
                b.d[3] = p.x; b.d[4] = p.y; b.d[5] = p.z;
                b.num = 1;
            } else {
                label_box &b = it->second;
                b.num += 1;
                if (p.x < b.d[0]) b.d[0] = p.x; else if (p.x > b.d[3]) b.d[3] = p.x;
                if (p.y < b.d[1]) b.d[1] = p.y; else if (p.y > b.d[4]) b.d[4] = p.y;
                if (p.z < b.d[2]) b.d[2] = p.z; else if (p.z > b.d[5]) b.d[5] = p.z;
            }
        }
        std::vector<std::pair<int, label_box>> tr(hist.begin(), hist.end());
        for (size_t m = 0; m < tr.size(); ++m) {
            const label_box &b = tr[m].second;
            if (b.num < 10) continue;
            Bbox bb;
            if (i == 1 || i == 4 || i == 5) bb.label = 1;
            else if (i == 16) bb.label = 2;
            else if (i == 18) bb.label = 3;
            bb.center = Vec3((b.d[0] + b.d[3]) / 2, (b.d[1] + b.d[4]) / 2, (b.d[2] + b.d[5]) / 2);
            bb.dimension = Vec3(b.d[3] - b.d[0], b.d[4] - b.d[1], b.d[5] - b.d[2]);
            bb.points_num = b.num;
            if (bb.label == 1) {
                if (bb.dimension.x < 0.5 || bb.dimension.y < 0.5 || bb.dimension.z < 0.3) continue;
            } else if (bb.label == 2 || bb.label == 3) {
                if (bb.dimension.z < 0.3) continue;
            }
            boxes.push_back(bb);
        }
    }
    return boxes;
}

// ---- frontend/SemGraph.cpp:14-108 BuildGraph, :215-240 MatrixDecomposing ----------------------
struct Graph {  // core/Coreutils.h:59-78 (fields the hot path fills)
    std::vector<int> node_labels, points_num;
    std::vector<Vec3> node_centers, node_dimensions;
    std::vector<std::vector<Vec3>> node_sub_triangles;
    std::vector<std::vector<float>> node_desc;
    std::vector<std::pair<int, int>> edges;
    std::vector<double> edge_value, edge_weights;
    std::vector<double> edge_matrix;  // N*N row-major
    std::vector<double> eigenvalues;  // oracle extra (ascending), for the "simple spectrum" test
    int vehicle_num = 0, trunk_num = 0, pole_like_num = 0;
};
inline std::vector<double> matrix_decomposing(const std::vector<double> &adj, int N, int dim,
                                              std::vector<double> *evals_out) {
    std::vector<double> a = adj, d;
    sym_eig(N, a, d);
    if (evals_out) *evals_out = d;
    std::vector<float> abs_evals((size_t)N);
    for (int i = 0; i < N; ++i) abs_evals[(size_t)i] = (float)std::fabs(d[(size_t)i]);
    std::vector<int> idx((size_t)N);
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int i, int j) { return abs_evals[(size_t)i] > abs_evals[(size_t)j]; });
    std::vector<double> out((size_t)N * dim, 0.0);  // N x dim row-major
    const int iter = std::min(dim, N);
    for (int c = 0; c < iter; ++c)
        for (int r = 0; r < N; ++r)
            out[(size_t)r * dim + c] = std::fabs(a[(size_t)r * N + idx[(size_t)c]]);
    return out;
}
inline Graph build_graph(const std::vector<Bbox> &boxes, double edge_dis_th, int subinterval,
                         int graph_node_dimension, double subgraph_edge_th) {
    Graph g;
    const int sub_interval_value = int(edge_dis_th / subinterval);
    const int N = (int)boxes.size();
    std::vector<double> Adj((size_t)N * N, 0.0), Edge((size_t)N * N, 0.0);
    const int LD = subinterval * 3;
    std::vector<double> Local((size_t)N * LD, 0.0);
    for (int i = 0; i < N; ++i) {
        g.node_labels.push_back(boxes[(size_t)i].label);
        g.node_centers.push_back(boxes[(size_t)i].center);
        g.node_dimensions.push_back(boxes[(size_t)i].dimension);
        g.points_num.push_back(boxes[(size_t)i].points_num);
        if (boxes[(size_t)i].label == 1) g.vehicle_num++;
        else if (boxes[(size_t)i].label == 2) g.trunk_num++;
        else if (boxes[(size_t)i].label == 3) g.pole_like_num++;
        for (int j = 0; j < N; ++j) {
            const double edge = norm(boxes[(size_t)i].center - boxes[(size_t)j].center);
            if (edge < edge_dis_th) {
                Adj[(size_t)i * N + j] = 1;
                Edge[(size_t)i * N + j] = edge;
                if (j >= i + 1) {
                    g.edges.emplace_back(i, j);
                    g.edge_value.push_back(edge);
                    g.edge_weights.push_back((edge_dis_th - edge) / edge_dis_th);
                }
                const int lj = boxes[(size_t)j].label;
                const int bin = int(edge / sub_interval_value);
                if (lj == 1) Local[(size_t)i * LD + bin]++;
                else if (lj == 2) Local[(size_t)i * LD + subinterval + bin]++;
                else if (lj == 3) Local[(size_t)i * LD + subinterval * 2 + bin]++;
            }
        }
    }
    if (N == 0) return g;
    std::vector<double> Global = matrix_decomposing(Adj, N, graph_node_dimension, &g.eigenvalues);
    for (int i = 0; i < N; ++i) {
        double s2 = 0;
        for (int c = 0; c < LD; ++c) s2 += Local[(size_t)i * LD + c] * Local[(size_t)i * LD + c];
        const double nrm = std::sqrt(s2);
        double sum = 0;
        for (int c = 0; c < graph_node_dimension; ++c) sum += Global[(size_t)i * graph_node_dimension + c];
        std::vector<float> desc;
        desc.reserve((size_t)LD + graph_node_dimension);
        for (int c = 0; c < LD; ++c) desc.push_back((float)(Local[(size_t)i * LD + c] / nrm));
        for (int c = 0; c < graph_node_dimension; ++c)
            desc.push_back((float)(Global[(size_t)i * graph_node_dimension + c] / sum));
        g.node_desc.push_back(std::move(desc));
    }
    for (int i = 0; i < N; ++i) {
        g.node_sub_triangles.emplace_back();
        std::vector<int> ind;
        for (int m = 0; m < N; ++m) {
            const double e = Edge[(size_t)i * N + m];
            if (e < subgraph_edge_th && e != 0) ind.push_back(m);
        }
        if (ind.size() > 2) {
            for (size_t m = 0; m + 1 < ind.size(); ++m)
                for (size_t n = m + 1; n < ind.size(); ++n) {
                    float t[3];
                    t[0] = (float)norm(g.node_centers[(size_t)i] - g.node_centers[(size_t)ind[m]]);
                    t[1] = (float)norm(g.node_centers[(size_t)i] - g.node_centers[(size_t)ind[n]]);
                    t[2] = (float)norm(g.node_centers[(size_t)ind[m]] - g.node_centers[(size_t)ind[n]]);
                    std::sort(t, t + 3);
                    g.node_sub_triangles[(size_t)i].push_back(Vec3(t[0], t[1], t[2]));
                }
        }
    }
    g.edge_matrix = Edge;
    return g;
}

// ---- loopclosure/LoopClosure.{hpp,cpp}: scan descriptor handed to the back-end -------
// fastAtan2 (LoopClosure.cpp:438-457, constants LoopClosure.hpp:38-41: FLOAT literals times 180/pi in double),
// degrees in [0, 360].
inline double fast_atan2_deg(double y, double x) {
    const double k = (double)(180 / 3.1415926535897932384626433832795);
    const double p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
                 p7 = -0.04432655554792128f * k;
    const double ax = std::abs(x), ay = std::abs(y);
    double a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (double)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (double)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
inline float float_dot_self(const std::vector<float> &v) {  // std::inner_product(..., 0.0f): float accumulator
    float acc = 0.0f;
    for (float x : v) acc = acc + x * x;
    return acc;
}
// GenGraphDescriptors (LoopClosure.cpp:93-140): 6 label-pair histograms of edge length + 3 instance counts
inline std::vector<float> gen_graph_descriptors(const Graph &g, double edge_dis_th, int subinterval) {
    const float siv = float(edge_dis_th / subinterval);
    std::vector<float> d((size_t)subinterval * 6 + 3, 0.f);
    for (size_t i = 0; i < g.edges.size(); ++i) {
        const int a = g.node_labels[(size_t)g.edges[i].first], b = g.node_labels[(size_t)g.edges[i].second];
        const int bin = int(g.edge_value[i] / siv) + 3;
        int blk = -1;
        if (a == 1 && b == 1) blk = 0;
        else if (a == 2 && b == 2) blk = 1;
        else if (a == 3 && b == 3) blk = 2;
        else if ((a == 3 && b == 2) || (a == 2 && b == 3)) blk = 3;
        else if ((a == 1 && b == 2) || (a == 2 && b == 1)) blk = 4;
        else if ((a == 1 && b == 3) || (a == 3 && b == 1)) blk = 5;
        if (blk >= 0) d[(size_t)(subinterval * blk + bin)]++;
    }
    const double ss_edge = float_dot_self(d);
    float n_edge = float(std::sqrt(ss_edge));
    if (n_edge == 0) n_edge = 0.000001f;
    for (float &x : d) x /= n_edge;
    float n_cnt = float(std::sqrt(g.vehicle_num * g.vehicle_num + g.trunk_num * g.trunk_num + g.pole_like_num * g.pole_like_num));
    if (n_cnt == 0) n_cnt = 0.000001f;
    d[0] = ((float)g.vehicle_num) / n_cnt;
    d[1] = ((float)g.trunk_num) / n_cnt;
    d[2] = ((float)g.pole_like_num) / n_cnt;
    const float ss_all = float_dot_self(d);
    float n_all = float(std::sqrt(ss_all));
    if (n_all == 0) n_all = 0.000001f;
    for (float &x : d) x /= n_all;
    return d;
}
// GenBackDescriptors (LoopClosure.cpp:145-188): ring x sector occupancy (24 x 360, 5..50 m) of building (13) and
// road (9) points, summed per ring, L2-normalised WITHOUT a zero guard (no such points -> NaN, as the reference).
inline std::vector<float> gen_back_descriptors(const Cloud &back) {
    static const int order_vec[26] = {0, 22, 0, 0, 21, 20, 0, 0, 0, 10, 11, 12, 13, 15, 16, 14, 17, 9, 18, 19, 0, 0, 0, 0, 0, 0};
    const double max_dis = 50, min_dis = 5;
    const int rings = 24, sectors = 360;
    const double ring_step = (max_dis - min_dis) / rings, sector_step = 360. / sectors;
    std::vector<char> build((size_t)rings * sectors, 0), road((size_t)rings * sectors, 0);
    for (size_t i = 0; i < back.p.size(); ++i) {
        const int label = back.l[i];
        if (label < 0 || label > 25 || !(order_vec[label] > 0)) continue;
        const Vec3 &p = back.p[i];
        const double distance = std::sqrt(p.x * p.x + p.y * p.y);
        if (distance >= max_dis || distance < min_dis) continue;
        const int sector_id = int(fast_atan2_deg(p.y, p.x) / sector_step);
        const int ring_id = int((distance - min_dis) / ring_step);
        if (ring_id >= rings || ring_id < 0) continue;
        if (sector_id >= sectors || sector_id < 0) continue;
        if (label == 13) build[(size_t)ring_id * sectors + sector_id] = 1;
        else if (label == 9) road[(size_t)ring_id * sectors + sector_id] = 1;
    }
    std::vector<float> d;
    for (const auto *m : {&build, &road})
        for (int r = 0; r < rings; ++r) {
            float sum = 0.f;  // 0/1 entries: exact in any order
            for (int c = 0; c < sectors; ++c) sum += (float)(*m)[(size_t)r * sectors + c];
            d.push_back(sum);
        }
    const float ss = float_dot_self(d);
    const float nrm = std::sqrt(ss);
    for (float &x : d) x /= nrm;
    return d;
}
// GenScanDescriptors (LoopClosure.cpp:15-32).  `cluster_cloud` is the 0.2 m cloud ClusterPoints was given: its
// background part (labels 14, 15, 9, 13; SemanticCluster.cpp:234-242) is graph.back_points (SemGraphSLAM.cpp:50).
inline std::vector<float> gen_scan_descriptors(const std::vector<Bbox> &boxes, const Cloud &cluster_cloud, double edge_dis_th,
                                               int subinterval) {
    Graph g;
    for (size_t i = 0; i < boxes.size(); ++i) {
        g.node_labels.push_back(boxes[i].label);
        if (boxes[i].label == 1) g.vehicle_num++;
        else if (boxes[i].label == 2) g.trunk_num++;
        else if (boxes[i].label == 3) g.pole_like_num++;
        for (size_t j = i + 1; j < boxes.size(); ++j) {
            const double edge = norm(boxes[i].center - boxes[j].center);
            if (edge < edge_dis_th) {
                g.edges.emplace_back((int)i, (int)j);
                g.edge_value.push_back(edge);
            }
        }
    }
    Cloud back;
    for (size_t i = 0; i < cluster_cloud.p.size(); ++i) {
        const int l = cluster_cloud.l[i];
        if (l == 14 || l == 15 || l == 9 || l == 13) { back.p.push_back(cluster_cloud.p[i]); back.l.push_back(l); }
    }
    std::vector<float> d = gen_graph_descriptors(g, edge_dis_th, subinterval);
    const std::vector<float> b = gen_back_descriptors(back);
    d.insert(d.end(), b.begin(), b.end());
    const float ss = float_dot_self(d);
    float nrm = std::sqrt(ss);
    if (nrm == 0) nrm = 0.000001f;
    for (float &x : d) x /= nrm;
    return d;
}

// ---- frontend/GraphMap.{hpp,cpp} + core/CommonUtils.cpp: instance tracking and relocalization ----
struct InsNode {  // core/Coreutils.h:81-87
    int label = 0, id = 0;
    Vec3 pose, dimension;
    int points_num = 0;
};
struct InsMatch {  // one entry of VTbii (GraphMap.hpp:23): (match flag, id in the frame graph, id in the local map)
    bool flag;
    int frame_id, map_id;
};
// GraphMap.cpp:11-40
inline bool check_sub_triangle(const std::vector<Vec3> &t1, const std::vector<Vec3> &t2) {
    int matched = 0;
    if (t1.size() < 2 && t2.size() < 2) return true;
    for (const Vec3 &a : t1)
        for (const Vec3 &b : t2)
            if (std::abs(a.x - b.x) < 0.3 && std::abs(a.y - b.y) < 0.3 && std::abs(a.z - b.z) < 0.3) matched++;
    if (t1.size() > 5 && t2.size() > 5) return matched >= 3;
    return matched >= 1;
}
// squared L2 distance as nanoflann's L2_Adaptor accumulates it (core/nanoflann.hpp, float, four terms at a time)
inline float desc_dist2(const std::vector<float> &a, const std::vector<float> &b) {
    float result = 0.f;
    size_t d = 0;
    const size_t n = a.size();
    while (d + 3 < n) {
        const float d0 = a[d] - b[d], d1 = a[d + 1] - b[d + 1], d2 = a[d + 2] - b[d + 2], d3 = a[d + 3] - b[d + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        d += 4;
    }
    while (d < n) {
        const float d0 = a[d] - b[d];
        result += d0 * d0;
        ++d;
    }
    return result;
}
// GraphMap.cpp:43-82 FindCorrespondencesKDtree: the 5 nearest map nodes in descriptor space (exact search, so a
// sorted brute-force scan gives the kd-tree's answer; equal distances keep the lower index), first one with the
// same label and a compatible sub-triangle set wins.  With fewer than 5 map nodes the unused result slots of the
// reference stay at index 0.
inline std::vector<std::pair<int, int>> find_correspondences_knn(const Graph &frame, const Graph &map, int k) {
    std::vector<std::pair<int, int>> out;
    const int M = (int)map.node_desc.size();
    for (int i = 0; i < (int)frame.node_desc.size(); ++i) {
        std::vector<std::pair<float, int>> d((size_t)M);
        for (int j = 0; j < M; ++j) d[(size_t)j] = {desc_dist2(frame.node_desc[(size_t)i], map.node_desc[(size_t)j]), j};
        std::stable_sort(d.begin(), d.end(), [](const std::pair<float, int> &a, const std::pair<float, int> &b) { return a.first < b.first; });
        for (int m = 0; m < k; ++m) {
            const int j = m < M ? d[(size_t)m].second : 0;
            if (frame.node_labels[(size_t)i] == map.node_labels[(size_t)j] &&
                check_sub_triangle(frame.node_sub_triangles[(size_t)i], map.node_sub_triangles[(size_t)j])) {
                out.emplace_back(i, j);
                break;
            }
        }
    }
    return out;
}
// CommonUtils.cpp:264-316 SolveSVD: Kabsch on the centred pairs, R = V E U^T.  The 3x3 SVD is a one-sided (Hestenes)
// Jacobi iteration on W; a rank-deficient W (three points are always coplanar after
// centring) has its missing directions completed by cross products - the determinant rule then fixes the sign,
// so every completion gives the same R.  Fewer than two independent directions (repeated sample indices) has no
// defined answer in the reference either (whatever JacobiSVD leaves); here it returns R = I.
struct Iso3 {  // Eigen::Isometry3d as rotation matrix + translation
    Mat3 R;
    Vec3 t;
};
inline Iso3 solve_svd(const std::vector<Vec3> &n1, const std::vector<Vec3> &n2) {
    const int N = (int)n1.size();
    Vec3 s1(0, 0, 0), s2(0, 0, 0);
    for (int i = 0; i < N; ++i) { s1 = s1 + n1[(size_t)i]; s2 = s2 + n2[(size_t)i]; }
    const Vec3 m1 = s1 / (double)N, m2 = s2 / (double)N;
    double W[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < N; ++i) {
        const Vec3 a = n1[(size_t)i] - m1, b = n2[(size_t)i] - m2;
        const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) W[r][c] += av[r] * bv[c];
    }
    // one-sided (Hestenes) Jacobi SVD of W: rotate column pairs of A = W V until they are orthogonal; then the
    // column norms are the singular values, the normalised columns U, and V the accumulated rotations
    double A[3][3], V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            A[r][c] = W[r][c];
            V[r][c] = r == c ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += A[k][p] * A[k][p];
                    beta += A[k][q] * A[k][q];
                    gamma += A[k][p] * A[k][q];
                }
                if (gamma == 0.0 || std::abs(gamma) <= 1e-16 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::abs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
    double sv[3];
    for (int j = 0; j < 3; ++j) sv[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return sv[a] > sv[b]; });
    Vec3 v[3], u[3];
    for (int k = 0; k < 3; ++k) v[k] = Vec3(V[0][ord[k]], V[1][ord[k]], V[2][ord[k]]);
    const double smax = sv[ord[0]];
    int rank = 0;
    for (int k = 0; k < 3; ++k) {
        const double sk = sv[ord[k]];
        if (!(sk > 1e-9 * smax && sk > 1e-150)) break;
        u[k] = Vec3(A[0][ord[k]] / sk, A[1][ord[k]] / sk, A[2][ord[k]] / sk);
        rank = k + 1;
    }
    Iso3 T;
    if (rank < 2) {
        T.R = Mat3();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T.R.m[i][j] = i == j ? 1.0 : 0.0;
    } else {
        if (rank == 2) {
            u[2] = cross(u[0], u[1]);
            v[2] = cross(v[0], v[1]);
        }
        // R = V E U^T with E = diag(1, 1, det(V U^T))
        auto build = [&](double e) {
            Mat3 R;
            const double vv[3][3] = {{v[0].x, v[1].x, v[2].x}, {v[0].y, v[1].y, v[2].y}, {v[0].z, v[1].z, v[2].z}};
            const double uu[3][3] = {{u[0].x, u[1].x, u[2].x}, {u[0].y, u[1].y, u[2].y}, {u[0].z, u[1].z, u[2].z}};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R.m[r][c] = vv[r][0] * uu[c][0] + vv[r][1] * uu[c][1] + e * vv[r][2] * uu[c][2];
            return R;
        };
        Mat3 R = build(1.0);
        const double det = R.m[0][0] * (R.m[1][1] * R.m[2][2] - R.m[1][2] * R.m[2][1]) - R.m[0][1] * (R.m[1][0] * R.m[2][2] - R.m[1][2] * R.m[2][0]) +
                           R.m[0][2] * (R.m[1][0] * R.m[2][1] - R.m[1][1] * R.m[2][0]);
        if (det < 0) R = build(-1.0);
        T.R = R;
    }
    T.t = m2 - T.R * m1;
    return T;
}
// CommonUtils.cpp:115-204 RansacAlignment(match1, match2, graph1, graph2, max_iter, best_inlier): random triplets of
// the matches, inliers = frame nodes that land within one 0.2 m voxel (held in a float) of some map node.
// std::random_device in the reference; a caller-given seed here (same engine and distribution).
inline std::pair<Iso3, double> ransac_alignment(const std::vector<Vec3> &m1, const std::vector<Vec3> &m2, const std::vector<Vec3> &nodes1,
                                                const std::vector<Vec3> &nodes2, int max_iter, uint32_t seed, int *best_inlier) {
    std::mt19937 gen(seed);
    std::uniform_int_distribution<> dis(0, (int)m1.size() - 1);
    Iso3 final_T;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) final_T.R.m[i][j] = i == j ? 1.0 : 0.0;
    final_T.t = Vec3(0, 0, 0);
    const float voxel_size = 0.2f;
    RobinMap<int> occupied;
    for (const Vec3 &p : nodes2) {
        const Key3 k{(int)(p.x / voxel_size), (int)(p.y / voxel_size), (int)(p.z / voxel_size)};
        if (occupied.find(k) < 0) occupied.insert(k, 1);
    }
    for (int it = 0; it < max_iter; ++it) {
        const int i1 = dis(gen), i2 = dis(gen), i3 = dis(gen);
        const Iso3 T = solve_svd({m1[(size_t)i1], m1[(size_t)i2], m1[(size_t)i3]}, {m2[(size_t)i1], m2[(size_t)i2], m2[(size_t)i3]});
        int inliers = 0;
        for (const Vec3 &p : nodes1) {
            const Vec3 q = T.R * p + T.t;
            const int kx = (int)(q.x / voxel_size), ky = (int)(q.y / voxel_size), kz = (int)(q.z / voxel_size);
            bool hit = false;
            for (int a = kx - 1; a <= kx + 1 && !hit; ++a)
                for (int b = ky - 1; b <= ky + 1 && !hit; ++b)
                    for (int c = kz - 1; c <= kz + 1 && !hit; ++c) hit = occupied.find(Key3{a, b, c}) >= 0;
            if (hit) inliers++;
        }
        if (inliers > *best_inlier) {
            *best_inlier = inliers;
            final_T = T;
        }
    }
    const double score = (float)*best_inlier / nodes1.size();
    return {final_T, score};
}
// Eigen's Quaternion(Matrix3) (what Sophus::SE3d(Matrix4d) does to the rotation part)
inline Quat quat_from_matrix(const Mat3 &R) {
    Quat q;
    const double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q.w = 0.5 * s;
        s = 0.5 / s;
        q.x = (R.m[2][1] - R.m[1][2]) * s; q.y = (R.m[0][2] - R.m[2][0]) * s; q.z = (R.m[1][0] - R.m[0][1]) * s;
    } else {
        int i = 0;
        if (R.m[1][1] > R.m[0][0]) i = 1;
        if (R.m[2][2] > R.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
        double v[3];
        v[i] = 0.5 * s;
        s = 0.5 / s;
        q.w = (R.m[k][j] - R.m[j][k]) * s;
        v[j] = (R.m[j][i] + R.m[i][j]) * s;
        v[k] = (R.m[k][i] + R.m[i][k]) * s;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    normalize(q);
    return q;
}
inline Graph build_graph(const std::vector<Bbox> &, double, int, int, double);
struct GraphMap {  // GraphMap.hpp:28-71
    int instance_id_ = 0;
    double voxel_size_ = 1.5;
    double edge_dis_th_, subgraph_edge_th_;
    int subinterval_, graph_node_dimension_;
    double nn_vehicle_, nn_pole_, max_distance_;
    Graph InsGraphMap;
    std::vector<InsNode> instance_in_localmap;
    GraphMap(double edge_dis_th, double subgraph_edge_th, int subinterval, int dim, double nn_vehicle, double nn_pole, double max_distance)
        : edge_dis_th_(edge_dis_th), subgraph_edge_th_(subgraph_edge_th), subinterval_(subinterval), graph_node_dimension_(dim),
          nn_vehicle_(nn_vehicle), nn_pole_(nn_pole), max_distance_(max_distance) {}
    // GraphMap.cpp:88-113
    std::vector<InsMatch> FindInsMatch(const Graph &graph) const {
        std::vector<InsMatch> out;
        if (InsGraphMap.node_labels.empty()) {
            for (int i = 0; i < (int)graph.node_labels.size(); ++i) out.push_back({false, i, -1});
            return out;
        }
        const auto corr = find_correspondences_knn(graph, InsGraphMap, 5);
        std::vector<bool> is_match(graph.node_labels.size(), false);
        for (const auto &c : corr) {
            out.push_back({true, c.first, c.second});
            is_match[(size_t)c.first] = true;
        }
        for (int i = 0; i < (int)is_match.size(); ++i)
            if (!is_match[(size_t)i]) out.push_back({false, i, -1});
        return out;
    }
    // GraphMap.cpp:227-262
    bool IsExistNearestNeighbor(const RobinMap<std::vector<int>> &hash, const Vec3 &p, int label, double th) const {
        const int kx = (int)(p.x / voxel_size_), ky = (int)(p.y / voxel_size_), kz = (int)(p.z / voxel_size_);
        for (int i = kx - 1; i <= kx + 1; ++i)
            for (int j = ky - 1; j <= ky + 1; ++j)
                for (int k = kz - 1; k <= kz + 1; ++k) {
                    const int64_t f = hash.find(Key3{i, j, k});
                    if (f < 0) continue;
                    for (int idx : hash.value_at(f)) {
                        if (idx < 0 || idx >= (int)instance_in_localmap.size()) continue;
                        if (instance_in_localmap[(size_t)idx].label != label) continue;
                        if (norm(p - instance_in_localmap[(size_t)idx].pose) < th) return true;
                    }
                }
        return false;
    }
    // GraphMap.cpp:149-222.  new_instance: frame node ids that became new map instances (Graph::new_instance)
    void Update(const Graph &graph, const std::vector<InsMatch> &matches, const SE3 &pose, std::vector<int> *new_instance) {
        RobinMap<std::vector<int>> hash;  // built once, from the instances present BEFORE this update (as the reference)
        for (int idx = 0; idx < (int)instance_in_localmap.size(); ++idx) {
            const Vec3 &p = instance_in_localmap[(size_t)idx].pose;
            const Key3 k{(int)(p.x / voxel_size_), (int)(p.y / voxel_size_), (int)(p.z / voxel_size_)};
            const int64_t f = hash.find(k);
            if (f >= 0) hash.value_at(f).push_back(idx);
            else hash.insert(k, std::vector<int>{idx});
        }
        for (const InsMatch &m : matches) {
            bool matched = m.flag;
            const int label = graph.node_labels[(size_t)m.frame_id];
            const double th = label == 1 ? nn_vehicle_ : nn_pole_;
            if (matched) {
                const Vec3 pose_ins = pose * graph.node_centers[(size_t)m.frame_id];
                InsNode &ins = instance_in_localmap[(size_t)m.map_id];
                if (norm(pose_ins - ins.pose) < th) {
                    ins.pose = (ins.pose + pose_ins) / 2.0;
                    ins.dimension = (ins.dimension + graph.node_dimensions[(size_t)m.frame_id]) / 2.0;
                } else {
                    matched = false;
                }
            }
            if (!matched) {
                const Vec3 &dim = graph.node_dimensions[(size_t)m.frame_id];
                if (label == 1) {
                    if (graph.points_num[(size_t)m.frame_id] < 50) continue;
                    if (dim.x < 1 || dim.y < 1 || dim.z < 1) continue;
                } else if (label == 2 || label == 3) {
                    if (dim.z < 1) continue;
                }
                InsNode ins;
                ins.pose = pose * graph.node_centers[(size_t)m.frame_id];
                ins.label = label;
                ins.dimension = dim;
                ins.points_num = graph.points_num[(size_t)m.frame_id];
                ins.id = instance_id_++;
                if (IsExistNearestNeighbor(hash, ins.pose, ins.label, nn_pole_)) continue;
                if (new_instance) new_instance->push_back(m.frame_id);
                instance_in_localmap.push_back(ins);
            }
        }
        const double max_d2 = max_distance_ * max_distance_;
        std::vector<InsNode> kept;
        for (const InsNode &ins : instance_in_localmap)
            if (!(sqnorm(ins.pose - pose.t) > max_d2)) kept.push_back(ins);
        instance_in_localmap.swap(kept);
        // ReBuildGraph (SemGraph.cpp:113-208) is BuildGraph on the instances
        std::vector<Bbox> boxes;
        for (const InsNode &ins : instance_in_localmap) {
            Bbox b;
            b.center = ins.pose; b.dimension = ins.dimension; b.label = ins.label; b.points_num = ins.points_num;
            boxes.push_back(b);
        }
        InsGraphMap = build_graph(boxes, edge_dis_th_, subinterval_, graph_node_dimension_, subgraph_edge_th_);
    }
    // GraphMap.cpp:118-144
    std::pair<SE3, bool> Relocalization(const Graph &graph, const std::vector<InsMatch> &matches, double inlier_rate_th, uint32_t seed,
                                        double *score_out = nullptr) const {
        std::vector<Vec3> m1, m2;
        for (const InsMatch &m : matches)
            if (m.flag) {
                m1.push_back(graph.node_centers[(size_t)m.frame_id]);
                m2.push_back(InsGraphMap.node_centers[(size_t)m.map_id]);
            }
        SE3 T;
        bool ok = false;
        double score = 0;
        if (m1.size() > 3) {
            int best = 0;
            const auto r = ransac_alignment(m1, m2, graph.node_centers, InsGraphMap.node_centers, 500, seed, &best);
            score = r.second;
            if (score > inlier_rate_th) ok = true;
            T.q = quat_from_matrix(r.first.R);
            T.t = r.first.t;
        }
        if (score_out) *score_out = score;
        return {T, ok};
    }
};

// ---- pipeline/SemGraphSLAM.{hpp,cpp} ----------------------------------------------------------
struct Config {  // SemGraphSLAM.hpp:25-64
    double voxel_size = 1.0, voxel_size_cluster = 0.2, max_range = 100.0, min_range = 5.0;
    int max_points_per_voxel = 20;
    double min_motion_th = 0.1, initial_threshold = 2.0;
    double model_deviation_trans = 0.1, model_deviation_rot = 0.01, inlier_rate_th = 0.4;
    bool deskew = false, relocalization_enable = false;
    double deltaA = 2, deltaR = 1, deltaP = 2;
    double edge_dis_th = 60, subgraph_edge_th = 20;
    int subinterval = 30, graph_node_dimension = 30;
    double nn_vehicle = 1.0, nn_pole = 1.0, max_local_graph_map_range = 60.0;
    int order_mode = 0, prune_mode = 0;
    uint32_t reloc_seed = 20251031u;  // stands in for std::random_device in RansacAlignment
};
struct ScanOutput {
    bool relocalized = false;
    Cloud cluster_cloud, frame_downsample, source, cropped;
    std::vector<Bbox> boxes;
    Graph graph;
    SE3 pose, initial_guess;
    double sigma = 0;
    IcpInfo icp;
    double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // pre, downsample, cluster, graph, icp, update
};
struct SemGraphSLAM {
    Config cfg;
    std::vector<SE3> poses_;
    VoxelHashMap local_map_;
    GraphMap local_graph_map_;
    AdaptiveThreshold adaptive_threshold_;
    bool do_graph = true;
    int relocalizations = 0, relocalization_attempts = 0;
    explicit SemGraphSLAM(const Config &c)
        : cfg(c),
          local_map_(c.voxel_size, c.max_range, c.max_points_per_voxel),
          local_graph_map_(c.edge_dis_th, c.subgraph_edge_th, c.subinterval, c.graph_node_dimension, c.nn_vehicle, c.nn_pole,
                           c.max_local_graph_map_range),
          adaptive_threshold_(c.initial_threshold, c.min_motion_th, c.max_range) {
        local_map_.prune_mode_ = c.prune_mode;
    }
    bool HasMoved() const {  // SemGraphSLAM.cpp:155-159
        if (poses_.empty()) return false;
        const double motion = norm((inverse(poses_.front()) * poses_.back()).t);
        return motion > 5.0 * cfg.min_motion_th;
    }
    double GetAdaptiveThreshold() {  // :147-153
        if (!HasMoved()) return cfg.initial_threshold;
        return adaptive_threshold_.ComputeThreshold();
    }
    SE3 GetPredictionModel() const {  // :161-166
        const size_t N = poses_.size();
        if (N < 2) return SE3();
        return inverse(poses_[N - 2]) * poses_[N - 1];
    }
    static std::vector<P4> Fuse(const Cloud &c) {  // :135-144
        std::vector<P4> out(c.size());
        for (size_t i = 0; i < c.size(); ++i) out[i] = P4{c.p[i].x, c.p[i].y, c.p[i].z, (double)c.l[i]};
        return out;
    }
    static double now_ms() { return omp_get_wtime() * 1e3; }
    // :16-121 mainProcess.  GraphMap::FindInsMatch / Update and the relocalization branch (:54, :81-107, :116) run
    // when relocalization_enable is set; they cannot influence the pose otherwise and are skipped then.
    ScanOutput mainProcess(const std::vector<Vec3> &frame, const std::vector<int> &frame_label,
                           const std::vector<double> &timestamps, bool is_kitti) {
        ScanOutput o;
        double t0 = now_ms();
        std::vector<Vec3> deskew_frame = frame;
        if (cfg.deskew && !timestamps.empty()) {
            const size_t N = poses_.size();
            if (N > 2) deskew_frame = deskew_scan(frame, timestamps, poses_[N - 2], poses_[N - 1]);
        }
        if (is_kitti) deskew_frame = correct_kitti_scan(frame);
        o.cropped = preprocess_semantic(deskew_frame, frame_label, cfg.max_range, cfg.min_range);
        double t1 = now_ms();
        o.stage_ms[0] = t1 - t0;
        // :124-132 VoxelizeSemantic
        o.cluster_cloud = voxel_downsample_semantic(o.cropped, cfg.voxel_size_cluster, cfg.order_mode, nullptr, nullptr);
        o.frame_downsample = voxel_downsample_semantic(o.cropped, cfg.voxel_size * 0.5, cfg.order_mode, nullptr, nullptr);
        o.source = voxel_downsample_semantic(o.frame_downsample, cfg.voxel_size * 1.5, cfg.order_mode, nullptr, nullptr);
        double t2 = now_ms();
        o.stage_ms[1] = t2 - t1;
        if (do_graph) {
            Cloud bg, fg;
            o.boxes = cluster_points(o.cluster_cloud.p, o.cluster_cloud.l, bg, fg, cfg.deltaA, cfg.deltaR, cfg.deltaP);
            double t3 = now_ms();
            o.stage_ms[2] = t3 - t2;
            o.graph = build_graph(o.boxes, cfg.edge_dis_th, cfg.subinterval, cfg.graph_node_dimension, cfg.subgraph_edge_th);
            o.stage_ms[3] = now_ms() - t3;
        }
        double t4 = now_ms();
        const double sigma = GetAdaptiveThreshold();
        const SE3 prediction = GetPredictionModel();
        const SE3 last_pose = !poses_.empty() ? poses_.back() : SE3();
        const SE3 initial_guess = last_pose * prediction;
        const std::vector<P4> source_4d = Fuse(o.source);
        const std::vector<P4> frame_ds_4d = Fuse(o.frame_downsample);
        const bool track = cfg.relocalization_enable && do_graph;
        std::vector<InsMatch> frame2map;
        if (track) frame2map = local_graph_map_.FindInsMatch(o.graph);  // :54
        SE3 new_pose = register_frame_semantic(source_4d, local_map_, initial_guess, 3.0 * sigma, sigma / 3.0, &o.icp);
        double t5 = now_ms();
        o.stage_ms[4] = t5 - t4;
        SE3 model_deviation = inverse(initial_guess) * new_pose;
        if (track && poses_.size() > 2) {  // :83-107
            double theta = 0;
            so3_log(model_deviation.q, &theta);
            if (norm(model_deviation.t) > cfg.model_deviation_trans || std::abs(theta) > cfg.model_deviation_rot) {
                relocalization_attempts++;
                const auto rel = local_graph_map_.Relocalization(o.graph, frame2map, cfg.inlier_rate_th, cfg.reloc_seed + (uint32_t)poses_.size());
                if (rel.second) {
                    IcpInfo again;
                    new_pose = register_frame_semantic(source_4d, local_map_, rel.first, 3.0 * sigma, sigma / 3.0, &again);
                    model_deviation = SE3();
                    relocalizations++;
                    o.relocalized = true;
                }
            }
        }
        adaptive_threshold_.UpdateModelDeviation(model_deviation);
        local_map_.Update(frame_ds_4d, new_pose);
        if (track) local_graph_map_.Update(o.graph, frame2map, new_pose, nullptr);  // :116
        poses_.push_back(new_pose);
        o.stage_ms[5] = now_ms() - t5;
        o.pose = new_pose;
        o.initial_guess = initial_guess;
        o.sigma = sigma;
        return o;
    }
};

}  // namespace sgo
