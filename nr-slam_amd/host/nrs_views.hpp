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

// TemporalBuffer flattened (map/temporal_buffer.h:43-63): n_frames snapshots, oldest first; per snapshot the camera pose
// and, per keypoint id, the keypoint (keypoint_tracks) and the landmark position (mapppoint_tracks_) when present.
// This is the wire form nrs_triangulate_batch reads (and what tests/golden stores for f2).
struct TemporalBufferView {
    int32_t n_frames = 0, n_ids = 0;
    std::vector<float> poses;             // n_frames x 7   Snapshot::camera_transform_world (qx qy qz qw tx ty tz)
    std::vector<uint8_t> has_kp, has_lm;  // n_frames x n_ids
    std::vector<float> kp_xy;             // n_frames x n_ids x 2
    std::vector<float> lm_xyz;            // n_frames x n_ids x 3
    std::vector<int32_t> last_status;     // n_ids          keypoint_tracks_status of the last snapshot
};

// KeyFrame snapshot (map/keyframe.cc:26-55): what a keyframe keeps of its frame, i.e. one block of KeyFrameWindow
struct KeyFrameView {
    double pose_qt[7];
    std::vector<int32_t> map_index;       // TRACKED_WITH_3D observations in keyframe index order
    std::vector<float> uv, xyz;
    void append_to(KeyFrameWindow& w) const;
};

class Engine {                            // owns one nrs_ctx; not thread-safe, like the reference's callers
public:
    explicit Engine(int device = -1) {
        nrs_options opt;
        nrs_options_init(&opt);
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

    // The same with the map's graph kept as a device-resident RegularizationGraph (all-pairs density): GetEdges / UpdateVertex
    // of OPT:252,468,496 are served there.  cap_per_point = how much of each sorted neighbour list is fetched.
    inline std::vector<int32_t> CameraPoseAndDeformationOptimization(const CameraView& cam, FrameView& f, MapView& m, class RegularizationGraph& graph,
                                                                     int cap_per_point = 64);

    // Skinned mode (include/nrs.h N2): M farthest-point graph nodes among the eligible map points (pick order)
    std::vector<int32_t> SelectGraphNodes(const std::vector<float>& positions, const std::vector<uint8_t>& eligible, int n_nodes) {
        std::vector<int32_t> nodes((size_t)n_nodes);
        check(nrs_skin_select_nodes(ctx_, (int32_t)(positions.size() / 3), positions.data(), eligible.empty() ? nullptr : eligible.data(), n_nodes, nodes.data()));
        return nodes;
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

    // The EMBEDDED form of the same call (include/nrs.h N2b; BASELINE configs[1] "points x graph nodes x keyframes"): is_node flags the
    // map points that carry vertices (e.g. SelectGraphNodes once per map); the other observations of the window are skinned to the
    // <= 11 node copies of their keyframe their walk accepts and keep their reprojection edges.  The walks see GetEdges lists with
    // everything but the nodes passed over: here they are cut out of the flat graph's ordered lists.  w.lm_xyz comes back with the
    // node copies' and the skinned points' new positions.
    void LocalDeformableBundleAdjustmentEmbedded(const CameraView& cam, KeyFrameWindow& w, MapView& m, const std::vector<uint8_t>& is_node, int iterations = 5) {
        const int32_t n_kf = (int32_t)w.kf_rowptr.size() - 1;
        if (n_kf < 3) return;                                              // OPT:922-924
        nrs_graph g = m.c_graph();
        std::vector<int32_t> orp((size_t)g.n_points + 1), ocol(m.col.size()), oeid(m.col.size());
        check(nrs_graph_select_neighbours(ctx_, &g, orp.data(), ocol.data(), oeid.data()));
        std::vector<int32_t> nrp((size_t)g.n_points + 1, 0), ncol, nst;
        std::vector<float> nw, nd0;
        for (int32_t p = 0; p < g.n_points; ++p) {                           // node neighbours only, GetEdges order kept
            for (int32_t a = orp[p]; a < orp[p + 1]; ++a)
                if (is_node[ocol[a]]) { ncol.push_back(ocol[a]); nw.push_back(m.e_w[oeid[a]]); nd0.push_back(m.e_d0[oeid[a]]); nst.push_back(m.e_status[oeid[a]]); }
            nrp[p + 1] = (int32_t)ncol.size();
        }
        if (ncol.empty()) { ncol.push_back(0); nw.push_back(0.f); nd0.push_back(0.f); nst.push_back(0); }
        int32_t nl = 0, ns = 0, nd = 0, nk = 0;
        check(nrs_dba_build_edges_embedded(n_kf, w.kf_rowptr.data(), w.kf_pt.data(), g.n_points, is_node.data(), nrp.data(), ncol.data(), nw.data(), nd0.data(),
                                           nst.data(), &nl, nullptr, &ns, nullptr, nullptr, &nd, nullptr, nullptr, &nk, nullptr, nullptr, nullptr));
        std::vector<int32_t> lm_obs((size_t)nl + 1), sp(2 * (size_t)ns + 2), dm(4 * (size_t)nd + 4), sk_obs((size_t)nk + 1), sk_node(11 * (size_t)nk + 11);
        std::vector<float> d0((size_t)ns + 1), dw((size_t)nd + 1);
        std::vector<double> sk_om(11 * (size_t)nk + 11);
        check(nrs_dba_build_edges_embedded(n_kf, w.kf_rowptr.data(), w.kf_pt.data(), g.n_points, is_node.data(), nrp.data(), ncol.data(), nw.data(), nd0.data(),
                                           nst.data(), &nl, lm_obs.data(), &ns, sp.data(), d0.data(), &nd, dm.data(), dw.data(), &nk, sk_obs.data(), sk_node.data(), sk_om.data()));
        std::vector<int32_t> obs_kf(w.kf_pt.size());
        for (int32_t k = 0; k < n_kf; ++k)
            for (int32_t l = w.kf_rowptr[k]; l < w.kf_rowptr[k + 1]; ++l) obs_kf[l] = k;
        std::vector<float> lxyz(3 * (size_t)nl + 3), luv(2 * (size_t)nl + 2), sxyz(3 * (size_t)nk + 3), suv(2 * (size_t)nk + 2);
        std::vector<int32_t> lkf((size_t)nl + 1), skf((size_t)nk + 1);
        for (int32_t i = 0; i < nl; ++i) { const int32_t o = lm_obs[i]; lkf[i] = obs_kf[o]; for (int c = 0; c < 3; ++c) lxyz[3 * i + c] = w.lm_xyz[3 * o + c]; luv[2 * i] = w.lm_uv[2 * o]; luv[2 * i + 1] = w.lm_uv[2 * o + 1]; }
        for (int32_t i = 0; i < nk; ++i) { const int32_t o = sk_obs[i]; skf[i] = obs_kf[o]; for (int c = 0; c < 3; ++c) sxyz[3 * i + c] = w.lm_xyz[3 * o + c]; suv[2 * i] = w.lm_uv[2 * o]; suv[2 * i + 1] = w.lm_uv[2 * o + 1]; }
        check(nrs_dba_solve_embedded(ctx_, &cam.cam, n_kf, w.poses_qt.data(), nl, lxyz.data(), lkf.data(), luv.data(), ns, sp.data(), d0.data(), nd, dm.data(), dw.data(),
                                     nk, skf.data(), suv.data(), sxyz.data(), sk_node.data(), sk_om.data(), m.scale, iterations, nullptr));
        for (int32_t i = 0; i < nl; ++i) for (int c = 0; c < 3; ++c) w.lm_xyz[3 * (size_t)lm_obs[i] + c] = lxyz[3 * i + c];      // OPT:1145-1160
        for (int32_t i = 0; i < nk; ++i) for (int c = 0; c < 3; ++c) w.lm_xyz[3 * (size_t)sk_obs[i] + c] = sxyz[3 * i + c];
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

    // absl::StatusOr<Eigen::Vector3f> DeformableTriangulation(TemporalBuffer&, int candidate_id, shared_ptr<CameraModel>, float)
    // g2o_optimization.h:34-37, for every candidate of Mapping::LandmarkTriangulation at once (mapping.cc:65-116):
    // status[i] = 0 (ok) or the InternalError code of include/nrs.h; xyz = the triangulated positions.
    void DeformableTriangulation(const CameraView& cam, const TemporalBufferView& tb, const std::vector<int32_t>& candidate_ids,
                                 std::vector<int32_t>& status, std::vector<float>& xyz, int min_track = 5) {
        status.assign(candidate_ids.size(), 0);
        xyz.assign(3 * candidate_ids.size(), 0.f);
        check(nrs_triangulate_batch(ctx_, &cam.cam, tb.n_frames, tb.poses.data(), tb.n_ids, tb.has_kp.data(), tb.kp_xy.data(),
                                    tb.has_lm.data(), tb.lm_xyz.data(), tb.last_status.data(), (int32_t)candidate_ids.size(),
                                    candidate_ids.data(), min_track, status.data(), xyz.data(), nullptr));
    }

    nrs_ctx* raw() { return ctx_; }
    void check_rc(int rc) { check(rc); }

private:
    void check(int rc) {
        // the reference signals trouble with LOG(FATAL) (regularization_graph.cc:93,125); the shim throws
        if (rc != NRS_OK) throw std::runtime_error(std::string("nrs: ") + nrs_last_error(ctx_));
    }
    nrs_ctx* ctx_ = nullptr;
};

inline void KeyFrameView::append_to(KeyFrameWindow& w) const {
    if (w.kf_rowptr.empty()) w.kf_rowptr.push_back(0);
    w.poses_qt.insert(w.poses_qt.end(), pose_qt, pose_qt + 7);
    w.kf_pt.insert(w.kf_pt.end(), map_index.begin(), map_index.end());
    w.lm_uv.insert(w.lm_uv.end(), uv.begin(), uv.end());
    w.lm_xyz.insert(w.lm_xyz.end(), xyz.begin(), xyz.end());
    w.kf_rowptr.push_back((int32_t)w.kf_pt.size());
}

// LucasKanadeTracker (modules/matching/lucas_kanade_tracker.h:55-70): same method names and argument meaning; the
// template caches the reference keeps as public members (Iref_, Idref_, vMeanI_, vMeanI2_, prevPts_) live in the context.
struct PhotometricInformation {           // lucas_kanade_tracker.h:38-45, all pyramid levels of one point
    float xy[2];
    std::vector<int16_t> gray, grad;
    std::vector<float> mean;
    std::vector<uint8_t> valid;
};

class LucasKanadeTracker {
public:
    // LucasKanadeTracker(cv::Size winSize, int maxLevel, int maxIters, float epsilon, float minEigThreshold)
    LucasKanadeTracker(Engine& e, int win_size = 21, int max_level = 4, int max_iters = 10, float epsilon = 1e-4f, float min_eig = 1e-4f)
        : e_(e), levels_(max_level + 1) {
        nrs_klt_config cfg{win_size, max_level, max_iters, epsilon, min_eig};
        e_.check_rc(nrs_klt_configure(e_.raw(), &cfg));
    }
    // void SetReferenceImage(cv::Mat& refIm, std::vector<cv::KeyPoint>& refPts, cv::Mat mask)
    void SetReferenceImage(const uint8_t* im, int w, int h, int stride, const std::vector<float>& ref_pts, const uint8_t* mask = nullptr) {
        e_.check_rc(nrs_klt_set_reference(e_.raw(), im, w, h, stride, mask, (int32_t)(ref_pts.size() / 2), ref_pts.data()));
    }
    // int Track(cv::Mat& newIm, std::vector<cv::KeyPoint>& nextPts, std::vector<LandmarkStatus>& status,
    //           const bool bInitialFlow, const float minSSIM, cv::Mat mask)
    int Track(const uint8_t* im, int w, int h, int stride, std::vector<float>& next_pts, std::vector<int32_t>& status,
              bool initial_flow, float min_ssim) {
        int32_t good = 0;
        e_.check_rc(nrs_klt_track(e_.raw(), im, w, h, stride, (int32_t)status.size(), next_pts.data(), status.data(), initial_flow ? 1 : 0,
                                  min_ssim, &good, nullptr));
        return good;
    }
    PhotometricInformation GetPhotometricInformationOfPoint(int idx) {
        PhotometricInformation p;
        p.gray.resize((size_t)levels_ * 441); p.grad.resize((size_t)levels_ * 882); p.mean.resize((size_t)levels_ * 2); p.valid.resize((size_t)levels_);
        e_.check_rc(nrs_klt_get_template(e_.raw(), idx, p.xy, p.gray.data(), p.grad.data(), p.mean.data(), p.valid.data()));
        return p;
    }
    void InsertPhotometricInformation(const PhotometricInformation& p) {
        e_.check_rc(nrs_klt_insert_template(e_.raw(), p.xy, p.gray.data(), p.grad.data(), p.mean.data(), p.valid.data()));
    }
    // (no reference counterpart: the photometric information of map points kept in device memory by id instead of in MapPoint --
    // ArchivePhotometricInformation at a keyframe, InsertArchived where the reference inserts a stored PhotometricInformation)
    void ArchivePhotometricInformation(const std::vector<int32_t>& slots, const std::vector<int32_t>& ids) {
        e_.check_rc(nrs_klt_archive_templates(e_.raw(), (int32_t)slots.size(), slots.data(), ids.data()));
    }
    void InsertArchived(LucasKanadeTracker& from, const std::vector<int32_t>& ids, const std::vector<float>& xy) {
        e_.check_rc(nrs_klt_insert_archived(e_.raw(), from.e_.raw(), (int32_t)ids.size(), ids.data(), xy.data()));
    }
    void clear() { e_.check_rc(nrs_klt_clear(e_.raw())); }
    int size() { return nrs_klt_num_points(e_.raw()); }

private:
    Engine& e_;
    int levels_;
};

// RegularizationGraph (modules/map/regularization_graph.h:34-96) at the reference's all-pairs density, device resident.
// Point "ids" are indices 0 .. capacity-1 (the shim keeps the MapPoint ID <-> index map, as it does for frames).
class RegularizationGraph {
public:
    struct Neighbour { int32_t id; float weight, first_distance; int32_t status; };
    RegularizationGraph(Engine& e, int capacity, float weight_sigma, float streching_th) : e_(e), cap_(capacity) {
        e_.check_rc(nrs_rgraph_create(e_.raw(), capacity, weight_sigma, streching_th, &g_));
    }
    ~RegularizationGraph() { nrs_rgraph_destroy(g_); }
    RegularizationGraph(const RegularizationGraph&) = delete;
    RegularizationGraph& operator=(const RegularizationGraph&) = delete;
    void SetSigma(float sigma) { e_.check_rc(nrs_rgraph_set_sigma(g_, sigma)); }
    float GetMinWeightAllowed() const { return nrs_rgraph_min_weight(g_); }
    // AddEdge(id, other, relative_position) for every (new, other) pair; positions = capacity x 3, by index
    void AddEdges(const std::vector<float>& positions, const std::vector<int32_t>& new_ids, const std::vector<int32_t>& other_ids) {
        e_.check_rc(nrs_rgraph_add_edges(g_, positions.data(), (int32_t)new_ids.size(), new_ids.data(), (int32_t)other_ids.size(), other_ids.data()));
    }
    // int UpdateVertex(ID) for each listed vertex: returns the good-connection counts
    std::vector<int32_t> UpdateVertices(const std::vector<float>& last_world_positions, const std::vector<int32_t>& ids) {
        std::vector<int32_t> good(ids.size());
        e_.check_rc(nrs_rgraph_update(g_, last_world_positions.data(), (int32_t)ids.size(), ids.data(), good.data()));
        return good;
    }
    // std::vector<std::pair<ID, shared_ptr<Edge>>> GetEdges(ID) for each listed vertex
    std::vector<std::vector<Neighbour>> GetEdges(const std::vector<int32_t>& ids, int cap_per_point = 256) {
        const size_t n = ids.size(), no = n * (size_t)cap_per_point;
        std::vector<int32_t> cnt(n), col(no), st(no);
        std::vector<float> w(no), d0(no);
        e_.check_rc(nrs_rgraph_get_edges(g_, (int32_t)n, ids.data(), cap_per_point, cnt.data(), col.data(), w.data(), d0.data(), st.data()));
        std::vector<std::vector<Neighbour>> out(n);
        for (size_t r = 0; r < n; ++r)
            for (int k = 0; k < cnt[r] && k < cap_per_point; ++k) out[r].push_back({col[r * cap_per_point + k], w[r * cap_per_point + k], d0[r * cap_per_point + k], st[r * cap_per_point + k]});
        return out;
    }

    nrs_rgraph* raw() { return g_; }
    int capacity() const { return cap_; }

private:
    Engine& e_;
    int cap_;
    nrs_rgraph* g_ = nullptr;
};

inline std::vector<int32_t> Engine::CameraPoseAndDeformationOptimization(const CameraView& cam, FrameView& f, MapView& m, RegularizationGraph& graph,
                                                                         int cap_per_point) {
    std::vector<int32_t> lost((size_t)graph.capacity());
    int32_t n_lost = 0;
    check(nrs_track_deform_solve_rg(ctx_, &cam.cam, graph.raw(), graph.capacity(), cap_per_point, m.last_world_position.data(), (int32_t)f.status.size(),
                                    f.map_index.data(), f.status.data(), f.uv.data(), f.pos.data(), f.pose_qt, m.scale,
                                    &f.deformation_magnitude, &n_lost, lost.data(), nullptr));
    lost.resize((size_t)n_lost);
    return lost;
}

}  // namespace nrs_host
