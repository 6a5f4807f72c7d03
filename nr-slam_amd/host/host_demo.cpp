// Runs the three optimisation entry points of the reference through the C++ host mirror
// (nrs_views.hpp: same names and argument meaning as modules/optimization/g2o_optimization.h:27-40)
// on a problem read from a binary blob, and writes the results to another blob.  It exists to
// exercise the C++ side of the boundary end to end (tests/test_gpu_host_mirror.py compares its
// output with the ctypes path bit for bit); it is not part of the library.
//
// Blob format: a sequence of arrays, each stored as int64 byte count followed by the raw bytes.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
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
    if (argc != 3) { std::fprintf(stderr, "usage: host_demo <in.blob> <out.blob>\n"); return 2; }
    try {
        std::ifstream in(argv[1], std::ios::binary);
        if (!in) throw std::runtime_error("cannot open input");
        const auto model = rd<int32_t>(in);
        const auto prm = rd<float>(in);
        nrs_host::CameraView cam = model[0] == NRS_CAM_PINHOLE ? nrs_host::CameraView::PinHole(prm[0], prm[1], prm[2], prm[3])
                                                               : nrs_host::CameraView::KannalaBrandt8(prm.data());
        // ---- frame + map (a1, a2)
        nrs_host::FrameView f;
        f.uv = rd<float>(in); f.pos = rd<float>(in); f.status = rd<int32_t>(in); f.map_index = rd<int32_t>(in);
        const auto qt = rd<double>(in);
        std::memcpy(f.pose_qt, qt.data(), sizeof(double) * 7);
        nrs_host::MapView m;
        m.rowptr = rd<int32_t>(in); m.col = rd<int32_t>(in); m.eid = rd<int32_t>(in);
        m.e_w = rd<float>(in); m.e_d0 = rd<float>(in); m.e_max = rd<float>(in); m.e_min = rd<float>(in);
        m.e_status = rd<int32_t>(in); m.last_world_position = rd<float>(in);
        const auto sss = rd<float>(in);
        m.sigma = sss[0]; m.stretch_th = sss[1]; m.scale = sss[2];
        // ---- BA window (a3) with its own map view
        nrs_host::KeyFrameWindow w;
        w.poses_qt = rd<double>(in); w.kf_rowptr = rd<int32_t>(in); w.kf_pt = rd<int32_t>(in);
        w.lm_uv = rd<float>(in); w.lm_xyz = rd<float>(in);
        nrs_host::MapView mb;
        mb.rowptr = rd<int32_t>(in); mb.col = rd<int32_t>(in); mb.eid = rd<int32_t>(in);
        mb.e_w = rd<float>(in); mb.e_d0 = rd<float>(in); mb.e_max = rd<float>(in); mb.e_min = rd<float>(in);
        mb.e_status = rd<int32_t>(in);
        const auto sb = rd<float>(in);
        mb.sigma = sb[0]; mb.stretch_th = sb[1]; mb.scale = sb[2];

        nrs_host::Engine eng;
        std::ofstream out(argv[2], std::ios::binary);
        // a1: CameraPoseOptimization on a copy of the frame
        nrs_host::FrameView f1 = f;
        eng.CameraPoseOptimization(cam, f1);
        wr(out, f1.pose_qt, 7);
        // a2: CameraPoseAndDeformationOptimization
        const std::vector<int32_t> lost = eng.CameraPoseAndDeformationOptimization(cam, f, m);
        wr(out, f.pose_qt, 7);
        wr(out, f.pos.data(), f.pos.size());
        wr(out, f.status.data(), f.status.size());
        wr(out, lost.data(), lost.size());
        wr(out, m.last_world_position.data(), m.last_world_position.size());
        wr(out, m.e_status.data(), m.e_status.size());
        // a3: LocalDeformableBundleAdjustment, then its embedded form on a copy of the window (every 6th map point a node)
        nrs_host::KeyFrameWindow we = w;
        eng.LocalDeformableBundleAdjustment(cam, w, mb);
        wr(out, w.poses_qt.data(), w.poses_qt.size());
        wr(out, w.lm_xyz.data(), w.lm_xyz.size());
        std::vector<uint8_t> is_node(mb.rowptr.size() - 1, 0);
        for (size_t p = 0; p < is_node.size(); p += 6) is_node[p] = 1;
        eng.LocalDeformableBundleAdjustmentEmbedded(cam, we, mb, is_node);
        wr(out, we.poses_qt.data(), we.poses_qt.size());
        wr(out, we.lm_xyz.data(), we.lm_xyz.size());
        std::printf("host_demo: ok (%zu frame slots, %zu lost, %zu BA landmarks)\n", f.status.size(), lost.size(), w.lm_xyz.size() / 3);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_demo: %s\n", e.what());
        return 1;
    }
}
