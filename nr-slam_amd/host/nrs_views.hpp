// Host-side mirror of the reference's module interface for the hot path, on flat views.
//
// The reference functions take Frame / Map / KeyFrame objects (modules/optimization/
// g2o_optimization.h:27-40).  A drop-in keeps those signatures and does three things: flatten the
// containers into the views below, call the C ABI, write the results back.  This header holds the
// compilable, reference-type-free half (views + calls, same names / argument meaning / error
// behaviour as the reference); INTEGRATION.md shows the ~40-line flattening stubs that use the
// reference's own accessors.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/nrs.h"

namespace nrs_host {

struct CameraView {                       // CameraModel::GetParameters() (calibration/camera_model.h:60-62)
    nrs_camera cam;
    static CameraView PinHole(float fx, float fy, float cx, float cy) {
        CameraView v{};
        v.cam.model = NRS_CAM_PINHOLE;
        v.cam.params[0] = fx; v.cam.params[1] = fy; v.cam.params[2] = cx; v.cam.params[3] = cy;
        return v;
    }
    static CameraView KannalaBrandt8(const float p[8]) {
        CameraView v{};
        v.cam.model = NRS_CAM_KB8;
        for (int i = 0; i < 8; ++i) v.cam.params[i] = p[i];
        return v;
    }
};

// Frame arrays in index order (map/frame.h:108-123): keypoints, landmark positions, statuses and
// the map-point index of every slot (-1 when the slot has no map point).
struct FrameView {
    std::vector<float> uv;                // n x 2   cv::KeyPoint::pt
    std::vector<float> pos;               // n x 3   LandmarkPositions()
    std::vector<int32_t> status;          // n       LandmarkStatuses()
    std::vector<int32_t> map_index;       // n       MapPointIdToIndex() inverted
    double pose_qt[7];                    // CameraTransformationWorld() as (unit quaternion xyzw, translation)
    float deformation_magnitude = 0.f;    // SetDeformationMaginitud
};

// RegularizationGraph (map/regularization_graph.h:49-96) + MapPoint::GetLastWorldPosition
struct MapView {
    std::vector<int32_t> rowptr, col, eid;
    std::vector<float> e_w, e_d0, e_max, e_min;
    std::vector<int32_t> e_status;
    std::vector<float> last_world_position;   // n_points x 3
    float sigma = 0.f, stretch_th = 1.1f, scale = 1.f;
    nrs_graph c_graph() {
        nrs_graph g;
        g.n_points = (int32_t)rowptr.size() - 1;
        g.rowptr = rowptr.data(); g.col = col.data(); g.eid = eid.data();
        g.n_edges = (int32_t)e_w.size();
        g.e_w = e_w.data(); g.e_d0 = e_d0.data(); g.e_max = e_max.data(); g.e_min = e_min.data();
        g.e_status = e_status.data();
        g.sigma = sigma; g.stretch_th = stretch_th;
        return g;
    }
};

struct KeyFrameWindow {                   // the BA window, oldest keyframe first (OPT:894-952)
    std::vector<double> poses_qt;         // n_kf x 7
    std::vector<int32_t> kf_rowptr, kf_pt;  // TRACKED_WITH_3D observations per keyframe -> map index
    std::vector<float> lm_uv, lm_xyz;     // per (keyframe, point) landmark, keyframe-major
};

class Engine {                            // owns one nrs_ctx; not thread-safe, like the reference's callers
public:
    explicit Engine(int device = -1) {
        nrs_options opt{};
        opt.device = device;
        if (nrs_create(&ctx_, &opt) != NRS_OK) throw std::runtime_error("nrs_create: no usable HIP device");
    }
    ~Engine() { nrs_destroy(ctx_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // void CameraPoseOptimization(Frame&, const Sophus::SE3f&)   g2o_optimization.h:27
    void CameraPoseOptimization(const CameraView& cam, FrameView& f) {
        std::vector<float> uv, X;
        for (size_t i = 0; i < f.status.size(); ++i)
            if (f.status[i] == NRS_TRACKED_WITH_3D) {
                uv.insert(uv.end(), {f.uv[2 * i], f.uv[2 * i + 1]});
                X.insert(X.end(), {f.pos[3 * i], f.pos[3 * i + 1], f.pos[3 * i + 2]});
            }
        check(nrs_pose_only_solve(ctx_, &cam.cam, (int32_t)(uv.size() / 2), uv.data(), X.data(), f.pose_qt, nullptr, nullptr));
    }

    // absl::flat_hash_set<ID> CameraPoseAndDeformationOptimization(Frame&, shared_ptr<Map>, const SE3f&, float)
    // g2o_optimization.h:29-32.  Returns the re-located lost map points.
    std::vector<int32_t> CameraPoseAndDeformationOptimization(const CameraView& cam, FrameView& f, MapView& m) {
        nrs_graph g = m.c_graph();
        std::vector<int32_t> lost((size_t)g.n_points);
        int32_t n_lost = 0;
        check(nrs_track_deform_solve(ctx_, &cam.cam, &g, m.last_world_position.data(), (int32_t)f.status.size(),
                                     f.map_index.data(), f.status.data(), f.uv.data(), f.pos.data(), f.pose_qt, m.scale,
                                     &f.deformation_magnitude, &n_lost, lost.data(), nullptr));
        lost.resize((size_t)n_lost);
        return lost;
    }

    // void LocalDeformableBundleAdjustment(shared_ptr<Map>, float scale)   g2o_optimization.h:39-40
    void LocalDeformableBundleAdjustment(const CameraView& cam, KeyFrameWindow& w, MapView& m, int iterations = 5) {
        const int32_t n_kf = (int32_t)w.kf_rowptr.size() - 1;
        if (n_kf < 3) return;                                              // OPT:922-924
        nrs_graph g = m.c_graph();
        std::vector<int32_t> orp((size_t)g.n_points + 1), ocol(m.col.size()), oeid(m.col.size());
        check(nrs_graph_select_neighbours(ctx_, &g, orp.data(), ocol.data(), oeid.data()));
        std::vector<float> nw(ocol.size()), nd0(ocol.size());
        std::vector<int32_t> nst(ocol.size());
        for (int32_t a = 0; a < orp.back(); ++a) { nw[a] = m.e_w[oeid[a]]; nd0[a] = m.e_d0[oeid[a]]; nst[a] = m.e_status[oeid[a]]; }
        int32_t ns = 0, nd = 0;
        check(nrs_dba_build_edges(n_kf, w.kf_rowptr.data(), w.kf_pt.data(), g.n_points, orp.data(), ocol.data(), nw.data(),
                                  nd0.data(), nst.data(), &ns, nullptr, nullptr, &nd, nullptr, nullptr));
        std::vector<int32_t> sp(2 * (size_t)ns), dm(4 * (size_t)nd), lm_kf(w.kf_pt.size());
        std::vector<float> d0((size_t)ns), dw((size_t)nd);
        check(nrs_dba_build_edges(n_kf, w.kf_rowptr.data(), w.kf_pt.data(), g.n_points, orp.data(), ocol.data(), nw.data(),
                                  nd0.data(), nst.data(), &ns, sp.data(), d0.data(), &nd, dm.data(), dw.data()));
        for (int32_t k = 0; k < n_kf; ++k)
            for (int32_t l = w.kf_rowptr[k]; l < w.kf_rowptr[k + 1]; ++l) lm_kf[l] = k;
        check(nrs_dba_solve(ctx_, &cam.cam, n_kf, w.poses_qt.data(), (int32_t)lm_kf.size(), w.lm_xyz.data(), lm_kf.data(),
                            w.lm_uv.data(), ns, sp.data(), d0.data(), nd, dm.data(), dw.data(), m.scale, iterations, nullptr));
    }

    // ShiTomasi::Extract(const cv::Mat&, std::vector<cv::KeyPoint>&)  features/shi_tomasi.h:45 followed by the
    // mask filter of Tracking::ExtractFeatures (tracking.cc:118-134): `keypoints` holds the frame's keypoints on
    // entry and the NEW ones (x, y) on return, `ids` their class ids.  The extractor's buffers live in the context
    // like the reference object's members; ShiTomasi(Options) = ConfigureShiTomasi.
    void ConfigureShiTomasi(int non_max_suppression_window = 5) { check(nrs_shi_configure(ctx_, non_max_suppression_window)); }
    void ExtractFeatures(const uint8_t* im, int w, int h, int stride, const uint8_t* mask, int mask_stride,
                         std::vector<float>& keypoints, std::vector<int32_t>& ids) {
        std::vector<float> held = keypoints;
        int32_t n = 0;
        std::vector<float> xy(2 * 4096);
        std::vector<int32_t> id(4096);
        for (;;) {
            check(nrs_shi_extract(ctx_, im, w, h, stride, mask, mask_stride, (int32_t)(held.size() / 2), held.data(),
                                  (int32_t)id.size(), xy.data(), id.data(), &n));
            if ((size_t)n <= id.size()) break;
            // truncated: the call has consumed ids and left its marks, like a reference Extract would; a larger
            // buffer cannot replay it.  4096 new corners per 640x480 keyframe are never reached (31x31 exclusion).
            throw std::runtime_error("nrs_shi_extract: more keypoints than the shim's buffer");
        }
        keypoints.assign(xy.begin(), xy.begin() + 2 * (size_t)n);
        ids.assign(id.begin(), id.begin() + n);
    }

    nrs_ctx* raw() { return ctx_; }

private:
    void check(int rc) {
        // the reference signals trouble with LOG(FATAL) (regularization_graph.cc:93,125); the shim throws
        if (rc != NRS_OK) throw std::runtime_error(std::string("nrs: ") + nrs_last_error(ctx_));
    }
    nrs_ctx* ctx_ = nullptr;
};

}  // namespace nrs_host
