// Header-only C++ facade: the reference's own names and signatures (namespace graph_slam) on top of
// the C ABI in include/sgs.h.  A maintainer of nubot-nudt/SG-SLAM swaps these declarations in for
//   semgraph_slam/frontend/Preprocessing.hpp:38-52, Deskew.hpp:33-36, VoxelHashMap.hpp:37-83,
//   Registration.hpp:62-66, SemanticCluster.hpp:81-87, SemGraph.hpp:18
// and links libsgs_hip.so; SemGraphSLAM::mainProcess (pipeline/SemGraphSLAM.cpp:16-121) compiles unchanged.  Graph,
// Bbox, InsNode (core/Coreutils.h:48-87) and AdaptiveThreshold (frontend/Threshold.hpp:30-53) are declared member for
// member, so the reference's other users of a Graph (loopclosure/LoopClosure.cpp:328, frontend/GraphMap.cpp:202,
// pipeline/SemGraphMapping.cpp:54) see the members they touch.
//
// With Eigen + Sophus available the facade uses them (SGS_FACADE_WITH_EIGEN, auto-detected).  Without
// (this build container has neither) it is instantiated on the small POD types below so that it is
// still compiled and exercised by tests/test_facade.cpp.
//
// Value semantics like the reference: inputs by const&, fresh vectors returned by value.  Errors: the
// reference has no error channel (silent degenerate cases, F/Registration.cpp:152); the facade logs to
// stderr and returns empty/identity results, it never throws.
#pragma once
#include <array>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "sgs.h"

#if !defined(SGS_FACADE_WITH_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<sophus/se3.hpp>)
#define SGS_FACADE_WITH_EIGEN 1
#endif
#endif

#if SGS_FACADE_WITH_EIGEN
#include <Eigen/Core>
#include <sophus/se3.hpp>
#endif

namespace graph_slam {

#if SGS_FACADE_WITH_EIGEN
using Vec3 = Eigen::Vector3d;
using Vec4 = Eigen::Vector4d;
using Pose = Sophus::SE3d;
inline void pose_to7(const Pose &T, double *p) {
    const auto q = T.unit_quaternion();
    p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); p[3] = q.w();
    p[4] = T.translation().x(); p[5] = T.translation().y(); p[6] = T.translation().z();
}
inline Pose pose_from7(const double *p) {
    return Pose(Eigen::Quaterniond(p[3], p[0], p[1], p[2]), Eigen::Vector3d(p[4], p[5], p[6]));
}
#else
struct Vec3 {
    double v[3];
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    const double *data() const { return v; }
    double *data() { return v; }
};
struct Vec4 {
    double v[4];
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    const double *data() const { return v; }
    double *data() { return v; }
};
struct Pose {  // [qx,qy,qz,qw,tx,ty,tz]
    double p[7] = {0, 0, 0, 1, 0, 0, 0};
};
inline void pose_to7(const Pose &T, double *p) { for (int i = 0; i < 7; ++i) p[i] = T.p[i]; }
inline Pose pose_from7(const double *p) { Pose T; for (int i = 0; i < 7; ++i) T.p[i] = p[i]; return T; }
#endif
static_assert(sizeof(Vec3) == 3 * sizeof(double) && sizeof(Vec4) == 4 * sizeof(double), "dense xyz / xyzl layout expected");

// core/Coreutils.h:34-56
typedef std::pair<std::vector<Vec3>, std::vector<int>> V3d_i;
typedef std::vector<Vec4> V4d;
typedef std::vector<Vec3> V3d;
struct Bbox {
    Vec3 center;
    Vec3 dimension;
    double theta = 0.0;
    int label = -1;
    double score = 0.0;
    int points_num = 0;
};
#if SGS_FACADE_WITH_EIGEN
using MatrixXd = Eigen::MatrixXd;
#else
// The subset of Eigen::MatrixXd the reference's users of Graph touch (core/Coreutils.h:71-72; SemGraph.cpp:84-86,105;
// loopclosure/LoopClosure.cpp:328-331: `auto e = g.edge_matrix.row(i); e.size(); e[m]`), column-major like Eigen.
class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(long rows, long cols) : r_(rows), c_(cols), v_((size_t)(rows * cols), 0.0) {}
    static MatrixXd Zero(long rows, long cols) { return MatrixXd(rows, cols); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
    double &operator()(long i, long j) { return v_[(size_t)(j * r_ + i)]; }
    double operator()(long i, long j) const { return v_[(size_t)(j * r_ + i)]; }
    const double *data() const { return v_.data(); }
    double *data() { return v_.data(); }
    void resize(long rows, long cols) { r_ = rows; c_ = cols; v_.assign((size_t)(rows * cols), 0.0); }
    void setZero() { v_.assign(v_.size(), 0.0); }
    struct ConstRow {
        const MatrixXd *m;
        long i;
        long size() const { return m->cols(); }
        double operator[](long j) const { return (*m)(i, j); }
        double operator()(long j) const { return (*m)(i, j); }
    };
    ConstRow row(long i) const { return ConstRow{this, i}; }
private:
    long r_ = 0, c_ = 0;
    std::vector<double> v_;
};
#endif
// core/Coreutils.h:59-78, member for member
struct Graph {
    std::vector<int> node_labels;
    std::vector<int> points_num;
    std::vector<Vec3> node_centers;
    std::vector<std::vector<Vec3>> node_sub_triangles;
    std::vector<Vec3> node_dimensions;
    std::vector<std::vector<float>> node_desc;
    std::vector<std::pair<int, int>> edges;
    std::vector<int> new_instance;   // filled by GraphMap::Update (frontend/GraphMap.cpp:202)
    std::vector<double> edge_value;
    std::vector<double> edge_weights;
    MatrixXd graph_matrix;           // never assigned by the reference's BuildGraph either (SemGraph.cpp:14-108)
    MatrixXd edge_matrix;            // N x N centre distances below edge_dis_th, 0 elsewhere (SemGraph.cpp:41,105)
    V3d_i back_points;
    V3d_i front_points;
    int vehicle_num = 0;
    int trunk_num = 0;
    int pole_like_num = 0;
};
// core/Coreutils.h:81-87
struct InsNode {
    int label;
    int id;
    Vec3 pose;
    Vec3 dimension;
    int points_num;
};

// frontend/Threshold.hpp:30-53 + Threshold.cpp:29-50: scalar host code, restated on the facade's Pose (it needs no device).
struct AdaptiveThreshold {
    explicit AdaptiveThreshold(double initial_threshold, double min_motion_th, double max_range)
        : initial_threshold_(initial_threshold), min_motion_th_(min_motion_th), max_range_(max_range) {}
    /// Update the current belief of the deviation from the prediction model
    inline void UpdateModelDeviation(const Pose &current_deviation) { model_deviation_ = current_deviation; }
    /// Returns the KISS-ICP adaptive threshold used in registration
    double ComputeThreshold() {
        double p[7];
        pose_to7(model_deviation_, p);
        // Eigen::AngleAxisd(R).angle() = 2 atan2(|q.vec|, |q.w|); model error = |t| + 2 max_range sin(theta / 2)
        const double theta = 2.0 * std::atan2(std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]), std::fabs(p[3]));
        const double delta_rot = 2.0 * max_range_ * std::sin(theta / 2.0);
        const double delta_trans = std::sqrt(p[4] * p[4] + (p[5] * p[5] + p[6] * p[6]));
        const double model_error = delta_trans + delta_rot;
        if (model_error > min_motion_th_) {
            model_error_sse2_ += model_error * model_error;
            num_samples_++;
        }
        if (num_samples_ < 1) return initial_threshold_;
        return std::sqrt(model_error_sse2_ / num_samples_);
    }
    // configurable parameters
    double initial_threshold_;
    double min_motion_th_;
    double max_range_;
    // Local cache for computation
    double model_error_sse2_ = 0;
    int num_samples_ = 0;
    Pose model_deviation_ = Pose();
};

// One device context per thread of use (the reference's VoxelHashMap is not thread safe either).
class HipContext {
public:
    static HipContext &instance() {
        static HipContext c;
        return c;
    }
    sgs_ctx *get(const sgs_config *cfg = nullptr) {
        if (!ctx_) {
            sgs_config c;
            if (cfg) c = *cfg; else sgs_preset("kitti", &c);
            if (sgs_create(&c, 0, &ctx_) != SGS_OK) {
                std::fprintf(stderr, "[graph_slam_hip] sgs_create failed: no HIP device or out of memory\n");
                ctx_ = nullptr;
            }
        }
        return ctx_;
    }
    ~HipContext() { if (ctx_) sgs_destroy(ctx_); }
private:
    sgs_ctx *ctx_ = nullptr;
};
inline bool ok(sgs_ctx *c, int r, const char *what) {
    if (r == SGS_OK) return true;
    std::fprintf(stderr, "[graph_slam_hip] %s: %s\n", what, c ? sgs_last_error(c) : "no context");
    return false;
}

// ---- Preprocessing.hpp ----
inline V3d_i PreprocessSemantic(const std::vector<Vec3> &frame, const std::vector<int> &label, double max_range,
                                double min_range) {
    V3d_i out;
    sgs_ctx *c = HipContext::instance().get();
    if (!c || frame.empty()) return out;
    const int n = (int)frame.size();
    out.first.resize(frame.size());
    out.second.resize(frame.size());
    int m = 0;
    if (!ok(c, sgs_preprocess(c, frame[0].data(), label.data(), n, max_range, min_range, out.first[0].data(),
                              out.second.data(), n, &m), "PreprocessSemantic"))
        m = 0;
    out.first.resize((size_t)m);
    out.second.resize((size_t)m);
    return out;
}
inline std::vector<Vec3> CorrectKITTIScan(const std::vector<Vec3> &frame) {
    std::vector<Vec3> out(frame.size());
    sgs_ctx *c = HipContext::instance().get();
    if (c && !frame.empty()) ok(c, sgs_correct_kitti(c, frame[0].data(), (int)frame.size(), out[0].data()), "CorrectKITTIScan");
    return out;
}
inline V3d_i VoxelDownsampleSemantic(const V3d_i &frame, double voxel_size) {
    V3d_i out;
    sgs_ctx *c = HipContext::instance().get();
    if (!c || frame.first.empty()) return out;
    const int n = (int)frame.first.size();
    out.first.resize((size_t)n);
    out.second.resize((size_t)n);
    int m = 0;
    if (!ok(c, sgs_voxel_downsample(c, frame.first[0].data(), frame.second.data(), n, voxel_size, out.first[0].data(),
                                    out.second.data(), nullptr, n, &m), "VoxelDownsampleSemantic"))
        m = 0;
    out.first.resize((size_t)m);
    out.second.resize((size_t)m);
    return out;
}
// ---- Deskew.hpp ----
inline std::vector<Vec3> DeSkewScan(const std::vector<Vec3> &frame, const std::vector<double> &timestamps,
                                    const Pose &start_pose, const Pose &finish_pose) {
    std::vector<Vec3> out(frame.size());
    sgs_ctx *c = HipContext::instance().get();
    double a[7], b[7];
    pose_to7(start_pose, a);
    pose_to7(finish_pose, b);
    if (c && !frame.empty())
        ok(c, sgs_deskew(c, frame[0].data(), timestamps.data(), (int)frame.size(), a, b, out[0].data()), "DeSkewScan");
    return out;
}

// ---- VoxelHashMap.hpp:37-83 ----
struct VoxelHashMap {
    using Vector4dVectorTuple = std::tuple<V4d, V4d>;
    explicit VoxelHashMap(double voxel_size, double max_distance, int max_points_per_voxel)
        : voxel_size_(voxel_size), max_distance_(max_distance), max_points_per_voxel_(max_points_per_voxel) {
        sgs_config cfg;
        sgs_preset("kitti", &cfg);
        cfg.voxel_size = voxel_size;
        cfg.max_range = max_distance;
        cfg.max_points_per_voxel = max_points_per_voxel;
        sgs_ctx *c = nullptr;
        if (sgs_create(&cfg, 0, &c) == SGS_OK) ctx_.reset(c, sgs_destroy);
        else std::fprintf(stderr, "[graph_slam_hip] VoxelHashMap: sgs_create failed\n");
    }
    Vector4dVectorTuple GetCorrespondences(const V4d &points, double max_correspondance_distance) const {
        V4d src(points.size()), tgt(points.size());
        int m = 0;
        if (ctx_ && !points.empty())
            ok(ctx_.get(), sgs_map_get_correspondences(ctx_.get(), points[0].data(), (int)points.size(), max_correspondance_distance,
                                                       src[0].data(), tgt[0].data(), nullptr, (int)points.size(), &m), "GetCorrespondences");
        src.resize((size_t)m);
        tgt.resize((size_t)m);
        return std::make_tuple(src, tgt);
    }
    inline void Clear() { if (ctx_) sgs_map_clear(ctx_.get()); }
    inline bool Empty() const {
        int e = 1;
        if (ctx_) sgs_map_empty(ctx_.get(), &e);
        return e != 0;
    }
    void Update(const V4d &points, const Vec3 &origin) {
        AddPoints(points);
        RemovePointsFarFromLocation(origin);
    }
    void Update(const V4d &points, const Pose &pose) {
        double p[7];
        pose_to7(pose, p);
        if (ctx_) ok(ctx_.get(), sgs_map_update(ctx_.get(), points.empty() ? nullptr : points[0].data(), (int)points.size(), p), "Update");
    }
    void AddPoints(const V4d &points) {
        if (ctx_ && !points.empty()) ok(ctx_.get(), sgs_map_add_points(ctx_.get(), points[0].data(), (int)points.size()), "AddPoints");
    }
    void RemovePointsFarFromLocation(const Vec3 &origin) {
        if (ctx_) ok(ctx_.get(), sgs_map_prune(ctx_.get(), origin.data()), "RemovePointsFarFromLocation");
    }
    std::vector<Vec4> Pointcloud() const {
        std::vector<Vec4> out;
        if (!ctx_) return out;
        int v = 0, np = 0;
        sgs_map_size(ctx_.get(), &v, &np);
        out.resize((size_t)np);
        int n = 0;
        if (np) ok(ctx_.get(), sgs_map_pointcloud(ctx_.get(), out[0].data(), np, &n), "Pointcloud");
        out.resize((size_t)n);
        return out;
    }
    sgs_ctx *ctx() const { return ctx_.get(); }
    double voxel_size_;
    double max_distance_;
    int max_points_per_voxel_;
private:
    std::shared_ptr<sgs_ctx> ctx_;
};

// ---- Registration.hpp:62-66 ----
inline Pose RegisterFrameSemantic(const std::vector<Vec4> &frame, const VoxelHashMap &voxel_map, const Pose &initial_guess,
                                  double max_correspondence_distance, double kernel) {
    double g[7], out[7];
    pose_to7(initial_guess, g);
    int it = 0;
    if (!voxel_map.ctx() ||
        !ok(voxel_map.ctx(), sgs_register_frame(voxel_map.ctx(), frame.empty() ? nullptr : frame[0].data(), (int)frame.size(), g,
                                                max_correspondence_distance, kernel, out, &it), "RegisterFrameSemantic"))
        return initial_guess;
    return pose_from7(out);
}

// ---- SemanticCluster.hpp:81-87 ----
inline std::vector<Bbox> ClusterPoints(const std::vector<Vec3> &frame, const std::vector<int> &frame_label, V3d_i &background_points,
                                       V3d_i &ClusterSemPoints, double delta_azimuth, double delta_range, double delta_polar) {
    std::vector<Bbox> out;
    // foreground / background splits are plain order-preserving label filters (SemanticCluster.cpp:222-244)
    for (size_t i = 0; i < frame.size(); ++i) {
        const int L = frame_label[i];
        if (L == 1 || L == 4 || L == 5 || L == 16 || L == 18) { ClusterSemPoints.first.push_back(frame[i]); ClusterSemPoints.second.push_back(L); }
        if (L == 14 || L == 15 || L == 9 || L == 13) { background_points.first.push_back(frame[i]); background_points.second.push_back(L); }
    }
    sgs_ctx *c = HipContext::instance().get();
    if (!c || frame.empty()) return out;
    std::vector<double> boxes(8 * 1024);
    int nb = 0, nfg = 0, nbg = 0;
    if (!ok(c, sgs_cluster_points(c, frame[0].data(), frame_label.data(), (int)frame.size(), delta_azimuth, delta_range, delta_polar,
                                  boxes.data(), 1024, &nb, nullptr, &nfg, &nbg), "ClusterPoints"))
        return out;
    out.resize((size_t)nb);
    for (int i = 0; i < nb; ++i) {
        const double *b = &boxes[8 * (size_t)i];
        for (int k = 0; k < 3; ++k) { out[(size_t)i].center[k] = b[k]; out[(size_t)i].dimension[k] = b[3 + k]; }
        out[(size_t)i].label = (int)b[6];
        out[(size_t)i].points_num = (int)b[7];
    }
    return out;
}

// ---- SemGraph.hpp:18 ----
inline Graph BuildGraph(const std::vector<Bbox> &cluster_boxes, double edge_dis_th, int subinterval, int graph_node_dimension,
                        double subgraph_edge_th) {
    Graph g;
    const int nb = (int)cluster_boxes.size();
    std::vector<double> boxes(8 * (size_t)nb);
    for (int i = 0; i < nb; ++i) {
        double *b = &boxes[8 * (size_t)i];
        for (int k = 0; k < 3; ++k) { b[k] = cluster_boxes[(size_t)i].center[k]; b[3 + k] = cluster_boxes[(size_t)i].dimension[k]; }
        b[6] = cluster_boxes[(size_t)i].label;
        b[7] = cluster_boxes[(size_t)i].points_num;
        g.node_labels.push_back(cluster_boxes[(size_t)i].label);
        g.node_centers.push_back(cluster_boxes[(size_t)i].center);
        g.node_dimensions.push_back(cluster_boxes[(size_t)i].dimension);
        g.points_num.push_back(cluster_boxes[(size_t)i].points_num);
    }
    sgs_ctx *c = HipContext::instance().get();
    if (!c || nb == 0) return g;
    const int D = 3 * subinterval + graph_node_dimension;
    std::vector<float> desc((size_t)nb * D);
    std::vector<double> em((size_t)nb * nb, 0.0);  // row-major from the ABI
    const int ecap = nb * nb, tcap = nb * nb * nb;
    std::vector<int32_t> edges(2 * (size_t)ecap), tri_counts((size_t)nb);
    std::vector<double> ev((size_t)ecap), ew((size_t)ecap), tri(3 * (size_t)tcap);
    int ne = 0, nt = 0;
    int32_t cc[3] = {0, 0, 0};
    if (!ok(c, sgs_build_graph(c, boxes.data(), nb, edge_dis_th, subinterval, graph_node_dimension, subgraph_edge_th, desc.data(),
                               em.data(), nullptr, edges.data(), ev.data(), ew.data(), ecap, &ne, tri_counts.data(),
                               tri.data(), tcap, &nt, cc), "BuildGraph"))
        return g;
    g.edge_matrix = MatrixXd::Zero(nb, nb);
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) g.edge_matrix(i, j) = em[(size_t)i * nb + j];
    g.vehicle_num = cc[0]; g.trunk_num = cc[1]; g.pole_like_num = cc[2];
    for (int i = 0; i < nb; ++i) g.node_desc.emplace_back(desc.begin() + (size_t)i * D, desc.begin() + (size_t)(i + 1) * D);
    for (int i = 0; i < ne; ++i) {
        g.edges.emplace_back(edges[2 * (size_t)i], edges[2 * (size_t)i + 1]);
        g.edge_value.push_back(ev[(size_t)i]);
        g.edge_weights.push_back(ew[(size_t)i]);
    }
    size_t t = 0;
    for (int i = 0; i < nb; ++i) {
        g.node_sub_triangles.emplace_back();
        for (int k = 0; k < tri_counts[(size_t)i]; ++k, ++t) {
            Vec3 v;
            v[0] = tri[3 * t]; v[1] = tri[3 * t + 1]; v[2] = tri[3 * t + 2];
            g.node_sub_triangles.back().push_back(v);
        }
    }
    return g;
}

// ---- loopclosure/LoopClosure.hpp:50 GenScanDescriptors(graph, edge_dis_th, subinterval): 6*subinterval+3 graph part +
// 48 background part.  graph.back_points is what ClusterPoints returned as background_points (SemGraphSLAM.cpp:50).
inline std::vector<float> GenScanDescriptors(const Graph &graph, double edge_dis_th, int subinterval) {
    std::vector<float> out((size_t)(6 * subinterval + 3 + 48), 0.f);
    sgs_ctx *c = HipContext::instance().get();
    if (!c) return out;
    const int nb = (int)graph.node_labels.size();
    std::vector<double> boxes(8 * (size_t)nb);
    for (int i = 0; i < nb; ++i) {
        double *b = &boxes[8 * (size_t)i];
        for (int k = 0; k < 3; ++k) { b[k] = graph.node_centers[(size_t)i][k]; b[3 + k] = graph.node_dimensions[(size_t)i][k]; }
        b[6] = graph.node_labels[(size_t)i];
        b[7] = graph.points_num[(size_t)i];
    }
    int n = 0;
    const Vec3 none{};
    ok(c, sgs_gen_scan_descriptor(c, boxes.data(), nb, graph.back_points.first.empty() ? none.data() : graph.back_points.first[0].data(),
                                  graph.back_points.second.data(), (int)graph.back_points.first.size(), edge_dis_th, subinterval,
                                  out.data(), (int)out.size(), &n), "GenScanDescriptors");
    return out;
}

// ---- loopclosure/FastIcp.hpp:40-105.  get_trans takes / returns poses here; with Eigen the reference's Matrix4d
// overload is provided on top.
class FastIcp {
public:
    FastIcp(float voxel_size_map, int max_points_per_voxel) : voxel_(voxel_size_map), cap_(max_points_per_voxel) {}
    Pose get_trans(const V3d_i &cloudA, const V3d_i &cloudB, const Pose &initTrans) {
        sgs_ctx *c = HipContext::instance().get();
        if (!c) return initTrans;
        auto fuse = [](const V3d_i &f) {
            std::vector<double> v(4 * f.first.size());
            for (size_t i = 0; i < f.first.size(); ++i) { v[4 * i] = f.first[i][0]; v[4 * i + 1] = f.first[i][1]; v[4 * i + 2] = f.first[i][2]; v[4 * i + 3] = f.second[i]; }
            return v;
        };
        const std::vector<double> a = fuse(cloudA), b = fuse(cloudB);
        double g[7], out[7];
        pose_to7(initTrans, g);
        int it = 0;
        if (!ok(c, sgs_fast_icp(c, a.data(), (int)cloudA.first.size(), b.data(), (int)cloudB.first.size(), g, voxel_, cap_, 0.5, out, &it),
                "FastIcp::get_trans"))
            return initTrans;
        return pose_from7(out);
    }
#if SGS_FACADE_WITH_EIGEN
    Eigen::Matrix4d get_trans(const V3d_i &cloudA, const V3d_i &cloudB, const Eigen::Matrix4d &initTrans) {
        const Pose T = get_trans(cloudA, cloudB, Pose(initTrans.block<3, 3>(0, 0), initTrans.block<3, 1>(0, 3)));
        Eigen::Matrix4d M = Eigen::Matrix4d::Identity();
        M.block<3, 3>(0, 0) = T.rotationMatrix();
        M.block<3, 1>(0, 3) = T.translation();
        return M;
    }
#endif
private:
    float voxel_;
    int cap_;
};

// ---- frontend/GraphMap.hpp:23-71 (the instance map lives in the context; Graph::new_instance is the second result)
using VTbii = std::vector<std::tuple<bool, int, int>>;
struct GraphMap {
    // FindInsMatch + Update in one step, as mainProcess uses them (SemGraphSLAM.cpp:54,116); like GraphMap::Update
    // (GraphMap.cpp:202) the overload that takes the graph by non-const reference records the new instances in it
    VTbii Track(Graph &graph, const Pose &pose) {
        std::vector<int> fresh;
        VTbii out = Track(static_cast<const Graph &>(graph), pose, &fresh);
        for (int id : fresh) graph.new_instance.emplace_back(id);
        return out;
    }
    VTbii Track(const Graph &graph, const Pose &pose, std::vector<int> *new_instance = nullptr) {
        VTbii out;
        sgs_ctx *c = HipContext::instance().get();
        const int nb = (int)graph.node_labels.size();
        if (!c) return out;
        std::vector<double> boxes(8 * (size_t)nb);
        for (int i = 0; i < nb; ++i) {
            double *b = &boxes[8 * (size_t)i];
            for (int k = 0; k < 3; ++k) { b[k] = graph.node_centers[(size_t)i][k]; b[3 + k] = graph.node_dimensions[(size_t)i][k]; }
            b[6] = graph.node_labels[(size_t)i];
            b[7] = graph.points_num[(size_t)i];
        }
        std::vector<int32_t> m(3 * (size_t)nb + 3), fresh((size_t)nb + 1);
        double p7[7];
        pose_to7(pose, p7);
        int nn = 0;
        if (!ok(c, sgs_graph_map_track(c, boxes.data(), nb, p7, m.data(), fresh.data(), &nn), "GraphMap::Track")) return out;
        for (int i = 0; i < nb; ++i) out.emplace_back(m[3 * (size_t)i] != 0, m[3 * (size_t)i + 1], m[3 * (size_t)i + 2]);
        if (new_instance) new_instance->assign(fresh.begin(), fresh.begin() + nn);
        return out;
    }
    size_t size() const {
        sgs_ctx *c = HipContext::instance().get();
        int n = 0;
        if (c) sgs_graph_map_instances(c, nullptr, 0, &n);
        return (size_t)n;
    }
};

}  // namespace graph_slam
