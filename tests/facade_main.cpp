// Compiles the header-only facade (reference names/signatures) against the C ABI and, when a GPU is
// present, runs a tiny registration through it.  Built and run by tests/test_facade.py.
#include <cmath>
#include <cstdio>
#include <random>

#include "graph_slam_hip.hpp"

using namespace graph_slam;

// Compile-time fidelity of the Graph / AdaptiveThreshold declarations: the access patterns of the reference's OTHER
// users of these types (not only mainProcess) must build against the facade.
static int graph_users_compile(Graph &graph, const Graph &graph1, double subgraph_edge_th) {
    // loopclosure/LoopClosure.cpp:328-332
    auto edge_list1 = graph1.edge_matrix.row(0);
    std::vector<int> indices1;
    for (int m = 0; m < edge_list1.size(); m++) {
        if (edge_list1[m] < subgraph_edge_th && edge_list1[m] != 0) indices1.push_back(m);
    }
    // frontend/GraphMap.cpp:202, pipeline/SemGraphMapping.cpp:54-57
    graph.new_instance.emplace_back(0);  // (the caller passes a graph with at least one node)
    int acc = 0;
    for (size_t i = 0; i < graph.new_instance.size(); i++) acc += graph.new_instance[i] + graph.node_labels[(size_t)graph.new_instance[i]] * 0;
    // core/Coreutils.h:71-72: both matrices exist and index as (row, col)
    graph.graph_matrix = MatrixXd::Zero(2, 2);
    graph.graph_matrix(1, 0) = 1.0;
    InsNode node{1, 7, graph.node_centers.empty() ? Vec3{} : graph.node_centers[0], Vec3{}, 12};
    return (int)indices1.size() + acc + node.id + (int)graph.edge_matrix.rows() + (int)graph.edge_matrix.cols();
}

int main() {
    {   // AdaptiveThreshold (frontend/Threshold.hpp:30-53): host-only, checked with or without a device
        AdaptiveThreshold th(2.0, 0.1, 100.0);
        if (th.ComputeThreshold() != 2.0) return 10;   // no sample yet: the initial threshold
        double d7[7] = {0, 0, std::sin(0.005), std::cos(0.005), 0.3, 0.4, 0.0};  // 0.01 rad about z, |t| = 0.5
        th.UpdateModelDeviation(pose_from7(d7));
        const double want = 0.5 + 2.0 * 100.0 * std::sin(0.005);
        if (std::fabs(th.ComputeThreshold() - want) > 1e-12 || th.num_samples_ != 1) return 11;
        double small7[7] = {0, 0, 0, 1, 0.01, 0, 0};  // below min_motion_th: not a sample, the RMS stays
        th.UpdateModelDeviation(pose_from7(small7));
        if (std::fabs(th.ComputeThreshold() - want) > 1e-12 || th.num_samples_ != 1) return 12;
        MatrixXd M = MatrixXd::Zero(3, 4);
        M(2, 3) = 5.0;
        if (M.rows() != 3 || M.cols() != 4 || M.row(2)[3] != 5.0 || M.row(2).size() != 4) return 13;
    }
    std::mt19937 rng(5);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    V4d scene(6000);
    for (auto &p : scene) { p[0] = 40 * u(rng); p[1] = 40 * u(rng); p[2] = 6 * u(rng); p[3] = 0; }
    VoxelHashMap map(1.0, 100.0, 20);
    if (!map.ctx()) { std::printf("facade: no device\n"); return 0; }
    if (!map.Empty()) return 2;
    map.AddPoints(scene);
    if (map.Empty()) return 3;
    // source = every 5th scene point shifted by a known translation; identity guess
    V4d src;
    for (size_t i = 0; i < scene.size(); i += 5) { Vec4 q = scene[i]; q[0] -= 0.3; q[1] += 0.2; q[2] -= 0.05; src.push_back(q); }
    Pose guess;
    const Pose T = RegisterFrameSemantic(src, map, guess, 3.0, 0.3);
    double p[7];
    pose_to7(T, p);
    std::printf("facade pose t = %.6f %.6f %.6f\n", p[4], p[5], p[6]);
    const bool good = std::fabs(p[4] - 0.3) < 1e-5 && std::fabs(p[5] + 0.2) < 1e-5 && std::fabs(p[6] - 0.05) < 1e-5;
    // preprocessing + downsample + cluster + graph through the reference's free functions
    std::vector<Vec3> frame(3000);
    std::vector<int> label(3000);
    for (size_t i = 0; i < frame.size(); ++i) { frame[i][0] = 60 * u(rng) - 30; frame[i][1] = 60 * u(rng) - 30; frame[i][2] = 2 * u(rng) - 1.5; label[i] = (int)(rng() % 26); }
    const V3d_i cropped = PreprocessSemantic(frame, label, 100.0, 5.0);
    const V3d_i ds = VoxelDownsampleSemantic(cropped, 0.5);
    V3d_i bg, fg;
    const std::vector<Bbox> boxes = ClusterPoints(ds.first, ds.second, bg, fg, 2.0, 0.5, 2.0);
    const Graph g = BuildGraph(boxes, 60.0, 30, 30, 20.0);
    {   // edge_matrix is the reference's N x N matrix: symmetric, zero diagonal, the listed edges at their places
        const long N = (long)boxes.size();
        if (g.edge_matrix.rows() != N || g.edge_matrix.cols() != N) { std::printf("facade FAILED: edge_matrix shape\n"); return 5; }
        for (size_t e = 0; e < g.edges.size(); ++e)
            if (g.edge_matrix(g.edges[e].first, g.edges[e].second) != g.edge_value[e] ||
                std::fabs(g.edge_matrix(g.edges[e].second, g.edges[e].first) - g.edge_value[e]) > 1e-12) { std::printf("facade FAILED: edge_matrix entries\n"); return 5; }
        for (long i = 0; i < N; ++i)
            if (g.edge_matrix(i, i) != 0.0) { std::printf("facade FAILED: edge_matrix diagonal\n"); return 5; }
    }
    std::printf("facade cropped %zu ds %zu boxes %zu graph nodes %zu map points %zu\n", cropped.first.size(), ds.first.size(),
                boxes.size(), g.node_desc.size(), map.Pointcloud().size());
    // back-end hooks: scan descriptor, loop-pair refinement, instance tracking
    Graph g2 = g;
    g2.back_points = bg;
    const std::vector<float> sd = GenScanDescriptors(g2, 60.0, 30);
    V3d_i A, B;
    for (size_t i = 0; i < scene.size(); i += 3) { Vec3 q; q[0] = scene[i][0]; q[1] = scene[i][1]; q[2] = scene[i][2]; B.first.push_back(q); B.second.push_back(0); }
    for (size_t i = 0; i < B.first.size(); i += 2) { Vec3 q = B.first[i]; q[0] -= 0.1; q[1] += 0.05; A.first.push_back(q); A.second.push_back(0); }
    FastIcp ficp(1.0f, 5);
    double f[7];
    pose_to7(ficp.get_trans(A, B, Pose()), f);
    const bool ficp_ok = std::fabs(f[4] - 0.1) < 1e-6 && std::fabs(f[5] + 0.05) < 1e-6;
    GraphMap gm;
    std::vector<int> fresh;
    const VTbii first = gm.Track(g2, Pose(), &fresh);
    Graph g3 = g2;
    const VTbii second = gm.Track(g3, Pose());  // non-const overload: fills g3.new_instance like GraphMap::Update
    if (!boxes.empty() && graph_users_compile(g3, g2, 20.0) < 0) return 6;
    size_t matched = 0;
    for (const auto &m : second) matched += std::get<0>(m) ? 1 : 0;
    std::printf("facade scan descriptor %zu floats, fast icp t = %.6f %.6f, graph map %zu instances, %zu new, %zu of %zu matched again\n", sd.size(),
                f[4], f[5], gm.size(), fresh.size(), matched, second.size());
    if (sd.size() != 231 || !ficp_ok || first.size() != boxes.size() || gm.size() != fresh.size() || matched < gm.size()) { std::printf("facade FAILED\n"); return 4; }
    std::printf(good && !cropped.first.empty() && ds.first.size() <= cropped.first.size() ? "facade ok\n" : "facade FAILED\n");
    return good ? 0 : 1;
}
