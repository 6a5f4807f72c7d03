// Second host-mirror demo: LucasKanadeTracker, RegularizationGraph and DeformableTriangulation through the C++ classes of
// nrs_views.hpp (same method names as modules/matching/lucas_kanade_tracker.h:55-70, modules/map/regularization_graph.h:
// 66-75, modules/optimization/g2o_optimization.h:34-37) on a binary blob; tests/test_gpu_host_mirror.py compares its
// output with the ctypes path bit for bit.  Blob format as host_demo.cpp.
#include <cstdio>
#include <cstring>
#include <fstream>
#include "nrs_views.hpp"

namespace {
template <class T>
std::vector<T> rd(std::ifstream& f) {
    int64_t bytes = 0;
    f.read(reinterpret_cast<char*>(&bytes), 8);
    std::vector<T> v((size_t)bytes / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), bytes);
    if (!f) throw std::runtime_error("short read");
    return v;
}
template <class T>
void wr(std::ofstream& f, const T* p, size_t n) {
    const int64_t bytes = (int64_t)(n * sizeof(T));
    f.write(reinterpret_cast<const char*>(&bytes), 8);
    f.write(reinterpret_cast<const char*>(p), bytes);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: host_demo2 <in.blob> <out.blob>\n"); return 2; }
    try {
        std::ifstream in(argv[1], std::ios::binary);
        if (!in) throw std::runtime_error("cannot open input");
        std::ofstream out(argv[2], std::ios::binary);
        nrs_host::Engine eng;
        // ---- LucasKanadeTracker: SetReferenceImage, Track, template round trip
        const auto wh = rd<int32_t>(in);
        const auto im0 = rd<uint8_t>(in), im1 = rd<uint8_t>(in);
        auto pts = rd<float>(in);
        nrs_host::LucasKanadeTracker klt(eng, 21, 4, 10, 1e-4f, 1e-4f);
        klt.SetReferenceImage(im0.data(), wh[0], wh[1], wh[0], pts);
        std::vector<int32_t> st(pts.size() / 2, NRS_TRACKED_WITH_3D);
        std::vector<float> next = pts;
        const int good = klt.Track(im1.data(), wh[0], wh[1], wh[0], next, st, true, 0.7f);
        wr(out, next.data(), next.size());
        wr(out, st.data(), st.size());
        const int32_t g32 = good;
        wr(out, &g32, 1);
        const nrs_host::PhotometricInformation ph = klt.GetPhotometricInformationOfPoint(3);
        klt.InsertPhotometricInformation(ph);
        const int32_t npts = klt.size();
        wr(out, &npts, 1);
        wr(out, ph.gray.data(), ph.gray.size());
        // ---- RegularizationGraph: all-pairs initialisation, UpdateVertex, GetEdges
        const auto gi = rd<float>(in);                      // sigma, stretch
        const auto pos0 = rd<float>(in), pos1 = rd<float>(in);
        const auto upd = rd<int32_t>(in);
        const int n = (int)(pos0.size() / 3);
        nrs_host::RegularizationGraph rg(eng, n, gi[0], gi[1]);
        std::vector<int32_t> all(n);
        for (int i = 0; i < n; ++i) all[i] = i;
        rg.AddEdges(pos0, all, all);
        const std::vector<int32_t> goodc = rg.UpdateVertices(pos1, upd);
        wr(out, goodc.data(), goodc.size());
        const auto nb = rg.GetEdges(all);
        std::vector<int32_t> flat;
        std::vector<float> fw;
        for (const auto& row : nb) {
            flat.push_back((int32_t)row.size());
            for (const auto& e : row) { flat.push_back(e.id); flat.push_back(e.status); fw.push_back(e.weight); fw.push_back(e.first_distance); }
        }
        wr(out, flat.data(), flat.size());
        wr(out, fw.data(), fw.size());
        // ---- DeformableTriangulation on a flat temporal buffer
        const auto model = rd<int32_t>(in);
        const auto prm = rd<float>(in);
        nrs_host::CameraView cam = model[0] == NRS_CAM_PINHOLE ? nrs_host::CameraView::PinHole(prm[0], prm[1], prm[2], prm[3])
                                                               : nrs_host::CameraView::KannalaBrandt8(prm.data());
        nrs_host::TemporalBufferView tb;
        const auto dims = rd<int32_t>(in);
        tb.n_frames = dims[0]; tb.n_ids = dims[1];
        tb.poses = rd<float>(in); tb.has_kp = rd<uint8_t>(in); tb.kp_xy = rd<float>(in); tb.has_lm = rd<uint8_t>(in);
        tb.lm_xyz = rd<float>(in); tb.last_status = rd<int32_t>(in);
        const auto cand = rd<int32_t>(in);
        std::vector<int32_t> tst;
        std::vector<float> txyz;
        eng.DeformableTriangulation(cam, tb, cand, tst, txyz);
        wr(out, tst.data(), tst.size());
        wr(out, txyz.data(), txyz.size());
        // ---- skinned pose-and-deformation: node selection, then CameraPoseAndDeformationOptimization on the device-resident graph
        const auto m2 = rd<int32_t>(in);                     // camera model, number of nodes
        const auto prm2 = rd<float>(in);
        nrs_host::CameraView cam2 = m2[0] == NRS_CAM_PINHOLE ? nrs_host::CameraView::PinHole(prm2[0], prm2[1], prm2[2], prm2[3])
                                                             : nrs_host::CameraView::KannalaBrandt8(prm2.data());
        nrs_host::FrameView f;
        f.uv = rd<float>(in); f.pos = rd<float>(in); f.status = rd<int32_t>(in); f.map_index = rd<int32_t>(in);
        const auto qt = rd<double>(in);
        std::memcpy(f.pose_qt, qt.data(), sizeof(double) * 7);
        nrs_host::MapView mv;
        mv.last_world_position = rd<float>(in);
        const auto sss = rd<float>(in);                      // sigma, stretch threshold, scale
        mv.sigma = sss[0]; mv.stretch_th = sss[1]; mv.scale = sss[2];
        const int np = (int)(mv.last_world_position.size() / 3);
        nrs_host::RegularizationGraph rg2(eng, np, mv.sigma, mv.stretch_th);
        std::vector<int32_t> all2(np);
        for (int i = 0; i < np; ++i) all2[i] = i;
        rg2.AddEdges(mv.last_world_position, all2, all2);
        std::vector<uint8_t> eligible(np, 0);
        for (size_t i = 0; i < f.status.size(); ++i)
            if (f.status[i] == NRS_TRACKED_WITH_3D && f.map_index[i] >= 0) eligible[f.map_index[i]] = 1;
        const std::vector<int32_t> nodes = eng.SelectGraphNodes(mv.last_world_position, eligible, m2[1]);
        std::vector<uint8_t> is_node(np, 0);
        for (int32_t id : nodes) is_node[id] = 1;
        for (size_t i = 0; i < f.status.size(); ++i)
            if (f.status[i] == NRS_TRACKED_WITH_3D && f.map_index[i] >= 0 && !is_node[f.map_index[i]]) f.status[i] = NRS_TRACKED;   // carried by stage 2
        const std::vector<int32_t> lost = eng.CameraPoseAndDeformationOptimization(cam2, f, mv, rg2, 256);
        wr(out, nodes.data(), nodes.size());
        wr(out, f.pose_qt, 7);
        wr(out, f.status.data(), f.status.size());
        wr(out, lost.data(), lost.size());
        wr(out, mv.last_world_position.data(), mv.last_world_position.size());
        std::printf("host_demo2: ok (%d tracked, %d graph points, %zu candidates, %zu nodes, %zu skinned)\n", good, n, cand.size(), nodes.size(), lost.size());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_demo2: %s\n", e.what());
        return 1;
    }
}
