// instancefusion_amd/host/ifx_ferns.hpp -- the keyframe data base of the reference's global loop closure (EF/Ferns.h, EF/Ferns.cpp) as host code
// above the C-ABI: random fern codes over a 1/8 resample of the model prediction, co-occurrence search, keyframe store, the ICP alignment of the
// current view against the best keyframe and the photometric check.  Its two GPU contacts are C-ABI calls: ifx_fern_frame (the four Resize passes)
// and ifx_track_maps (RGBDOdometry on two small renders, on a handle of its own created at fern resolution).  The maths on the read-back images is
// plain host code, as in the reference, and can be driven without a GPU through addFrameMaps / findFrameMaps (tests/cpp/ferns_check.cpp).
//
// Not carried over: the instance-aware ICP variant of findFrame (EF/Ferns.cpp:225-556) -- it sits behind `if(false)` in the reference, so instICP is
// always false, smallInstanceTable is never written and the plain ElasticFusion gates (:644) decide.
#ifndef IFX_FERNS_HPP_
#define IFX_FERNS_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "ifx_c_api.h"

// Matrix4f: the row-major 4 x 4 of ifx_host.hpp (typedef FernsT<Matrix4f> Ferns there)
template <class Matrix4f>
class FernsT {
public:
    struct SurfaceConstraint {   // EF/Ferns.h:52-64
        float sourcePoint[4];
        float targetPoint[4];
    };
    struct Fern {   // EF/Ferns.h:82-91
        int pos[2];
        int rgbd[4];
        std::vector<int> ids[16];
    };
    struct Frame {   // EF/Ferns.h:95-169
        std::vector<unsigned char> codes;
        int goodCodes = 0;
        int id = 0;
        Matrix4f pose;
        int srcTime = 0;
        std::vector<unsigned char> initRgb, initInst;   // num x 3
        std::vector<float> initVerts, initNorms;        // num x 4
        int num = 0;
    };
    // RGBDOdometry on two vertex / normal maps of fern resolution: model (keyframe, in the frame of pose16) against current; pose16 in = start, out =
    // estimate; diag8[0] = lastICPError, diag8[1] = lastICPCount
    using Tracker = std::function<void(const float* modelV, const float* modelN, const float* curV, const float* curN, float* pose16, float* diag8)>;

    // EF/Ferns.cpp:20-54.  fullWidth / fullHeight / intrinsics: what the reference reads from its singletons.  The reference seeds with time(0).
    FernsT(int n, int maxDepth, float photoThresh, int fullWidth, int fullHeight, float fx, float fy, float cx, float cy, int device = 0,
           uint32_t seed = (uint32_t)std::time(0))
        : num(n), factor(8), width(fullWidth / 8), height(fullHeight / 8), maxDepth(maxDepth), photoThresh(photoThresh), lastClosest(-1), badCode(255),
          fx_(fx / 8), fy_(fy / 8), cx_(cx / 8), cy_(cy / 8), device_(device)
    {
        random.seed(seed);
        generateFerns();
    }
    ~FernsT()
    {
        if (tracker_handle_) ifx_destroy(tracker_handle_);
    }
    FernsT(const FernsT&) = delete;
    FernsT& operator=(const FernsT&) = delete;

    void setTracker(Tracker t) { tracker_ = std::move(t); }   // tests: a stand-in for the GPU tracker

    // ---- EF/Ferns.cpp:83-172
    bool addFrame(ifx_t* h, const Matrix4f& pose, int srcTime, float threshold)
    {
        readBack(h);
        return addFrameMaps(img_.data(), verts_.data(), norms_.data(), inst_.data(), pose, srcTime, threshold);
    }
    bool addFrameMaps(const unsigned char* img, const float* verts, const float* norms, const unsigned char* inst, const Matrix4f& pose, int srcTime,
                      float threshold)
    {
        const int np = width * height;
        std::unique_ptr<Frame> frame(new Frame());
        frame->id = (int)frames.size();
        frame->pose = pose;
        frame->srcTime = srcTime;
        frame->num = np;
        frame->initRgb.assign(img, img + (size_t)np * 3);
        frame->initInst.assign(inst, inst + (size_t)np * 3);
        frame->initVerts.assign(verts, verts + (size_t)np * 4);
        frame->initNorms.assign(norms, norms + (size_t)np * 4);
        std::vector<int> coOccurrences(frames.size(), 0);
        encode(img, verts, *frame, coOccurrences);

        float minimum = std::numeric_limits<float>::max();
        if (frame->goodCodes > 0) {
            for (size_t i = 0; i < frames.size(); i++) {
                float maxCo = (float)std::min(frame->goodCodes, frames[i]->goodCodes);
                float dissim = (float)(maxCo - coOccurrences[i]) / (float)maxCo;
                if (dissim < minimum) minimum = dissim;
            }
        }
        if ((minimum > threshold || frames.size() == 0) && frame->goodCodes > 0) {
            for (int i = 0; i < num; i++)
                if (frame->codes[i] != badCode) conservatory[i].ids[frame->codes[i]].push_back(frame->id);
            frames.push_back(std::move(frame));
            return true;
        }
        return false;
    }

    // ---- EF/Ferns.cpp:174-700.  Call it inside the fern callback of the handle (ifx_set_fern_callback): ifx_fern_frame then delivers the
    // predict() at the tracked pose.  Returns the recovery pose (identity when nothing matched); lastClosest != -1 when constraints were produced.
    Matrix4f findFrame(std::vector<SurfaceConstraint>& constraints, const Matrix4f& currPose, ifx_t* h, int* smallInstanceTable, const int time,
                       const bool lost)
    {
        (void)smallInstanceTable;   // written by the disabled instance-aware branch only
        readBack(h);
        return findFrameMaps(constraints, currPose, img_.data(), verts_.data(), norms_.data(), time, lost);
    }
    Matrix4f findFrameMaps(std::vector<SurfaceConstraint>& constraints, const Matrix4f& currPose, const unsigned char* imgSmall, const float* vertSmall,
                           const float* normSmall, const int time, const bool lost)
    {
        lastClosest = -1;
        lastICPError = lastICPCount = lastPhotoError = 0;
        Frame frame;
        std::vector<int> coOccurrences(frames.size(), 0);
        encode(imgSmall, vertSmall, frame, coOccurrences);

        float minimum = std::numeric_limits<float>::max();
        int minId = -1;
        for (size_t i = 0; i < frames.size(); i++) {
            float maxCo = (float)std::min(frame.goodCodes, frames[i]->goodCodes);
            float dissim = (float)(maxCo - coOccurrences[i]) / (float)maxCo;
            if (dissim < minimum && time - frames[i]->srcTime > minTimeGap) {
                minimum = dissim;
                minId = (int)i;
            }
        }
        lastDissimilarity = minimum;
        lastCandidate = minId;

        Matrix4f estPose = Matrix4f::Identity();
        if (minId != -1 && blockHDAware(&frame, frames[minId].get()) > 0.3) {
            const Frame& fern = *frames[minId];
            const Matrix4f fernPose = fern.pose;
            // :558-592 -- initICPModel(fern maps, fernPose), initICP(current maps), getIncrementalTransformation(fernPose, icpWeight 100, no pyramid,
            // no fast odometry, no SO(3)); the colour side is commented out in the reference
            float pose16[16], diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            std::memcpy(pose16, fernPose.data(), 64);
            track(fern.initVerts.data(), fern.initNorms.data(), vertSmall, normSmall, pose16, diag);
            std::memcpy(estPose.data(), pose16, 64);
            lastICPError = diag[0];
            lastICPCount = diag[1];

            const float photoError = photometricCheck(vertSmall, imgSmall, estPose, fernPose, fern.initRgb.data());
            lastPhotoError = photoError;
            const int icpCountThresh = lost ? 1400 : 2400;
            if (lastICPError < 0.0003 && lastICPCount > icpCountThresh && photoError < photoThresh) {   // :644
                lastClosest = minId;
                const int step = num / 50 > 0 ? num / 50 : 1;   // (the reference loops forever with fewer than 50 ferns)
                for (int i = 0; i < num; i += step) {
                    const float* v = &vertSmall[((size_t)conservatory[i].pos[1] * width + conservatory[i].pos[0]) * 4];
                    if (v[2] > 0 && int(v[2] * 1000.0f) < maxDepth) {
                        SurfaceConstraint c;
                        mulPoint(currPose, v, c.sourcePoint);   // worldRawPoint
                        mulPoint(estPose, v, c.targetPoint);    // worldModelPoint
                        constraints.push_back(c);
                    }
                }
            }
        }
        return estPose;
    }

    std::vector<Fern> conservatory;
    std::vector<std::unique_ptr<Frame>> frames;

    const int num;
    std::mt19937 random;
    const int factor;
    const int width;
    const int height;
    const int maxDepth;
    const float photoThresh;
    int lastClosest;
    int minTimeGap = 300;   // EF/Ferns.cpp:238: a keyframe is a candidate when it is more than 300 frames old
    const unsigned char badCode;
    // diagnostics of the last findFrame (rgbd.lastICPError / lastICPCount in the reference)
    float lastICPError = 0, lastICPCount = 0, lastPhotoError = 0, lastDissimilarity = 0;
    int lastCandidate = -1;

    // :791-808
    float blockHDAware(const Frame* f1, const Frame* f2) const
    {
        int count = 0;
        float val = 0;
        for (int i = 0; i < num; i++) {
            if (f1->codes[i] != badCode && f2->codes[i] != badCode) {
                count++;
                if (f1->codes[i] == f2->codes[i]) val += 1.0f;
            }
        }
        return val / (float)count;
    }
    // :777-789
    float blockHD(const Frame* f1, const Frame* f2) const
    {
        float sum = 0.0f;
        for (int i = 0; i < num; i++) sum += f1->codes[i] == f2->codes[i];
        return sum / (float)num;
    }

private:
    // :65-81 -- same generator, same distributions, same draw order
    void generateFerns()
    {
        std::uniform_int_distribution<int32_t> widthDist(0, width - 1), heightDist(0, height - 1), rgbDist(0, 255), dDist(400, maxDepth);
        for (int i = 0; i < num; i++) {
            Fern f;
            f.pos[0] = widthDist(random);
            f.pos[1] = heightDist(random);
            f.rgbd[0] = rgbDist(random);
            f.rgbd[1] = rgbDist(random);
            f.rgbd[2] = rgbDist(random);
            f.rgbd[3] = dDist(random);
            conservatory.push_back(f);
        }
    }
    // the code of every fern and the co-occurrence count of every stored keyframe (:108-131, :204-227)
    void encode(const unsigned char* img, const float* verts, Frame& frame, std::vector<int>& coOccurrences) const
    {
        frame.codes.assign((size_t)num, badCode);
        frame.goodCodes = 0;
        for (int i = 0; i < num; i++) {
            const Fern& f = conservatory[i];
            const size_t k = (size_t)f.pos[1] * width + f.pos[0];
            unsigned char code = badCode;
            if (verts[k * 4 + 2] > 0) {
                const unsigned char* pix = &img[k * 3];
                code = (unsigned char)((pix[0] > f.rgbd[0]) << 3 | (pix[1] > f.rgbd[1]) << 2 | (pix[2] > f.rgbd[2]) << 1 | (int(verts[k * 4 + 2] * 1000.0f) > f.rgbd[3]));
                frame.goodCodes++;
                for (size_t j = 0; j < f.ids[code].size(); j++) coOccurrences[f.ids[code][j]]++;
            }
            frame.codes[i] = code;
        }
    }
    // :702-775.  fernPose^-1 as a rigid-body inverse (the reference takes Eigen's general 4 x 4 inverse of the same matrix)
    float photometricCheck(const float* vertSmall, const unsigned char* imgSmall, const Matrix4f& estPose, const Matrix4f& fernPose, const unsigned char* fernRgb) const
    {
        const float invfx = 1.0f / fx_, invfy = 1.0f / fy_;
        float inv[16], diff[16];
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) inv[r * 4 + c] = fernPose(c, r);
            inv[r * 4 + 3] = -(fernPose(0, r) * fernPose(0, 3) + fernPose(1, r) * fernPose(1, 3) + fernPose(2, r) * fernPose(2, 3));
        }
        inv[12] = inv[13] = inv[14] = 0; inv[15] = 1;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float s = 0;
                for (int k = 0; k < 4; k++) s += inv[r * 4 + k] * estPose(k, c);
                diff[r * 4 + c] = s;
            }
        float photoSum = 0;
        int photoCount = 0;
        for (int i = 0; i < num; i++) {
            const size_t k = (size_t)conservatory[i].pos[1] * width + conservatory[i].pos[0];
            const float* v = &vertSmall[k * 4];
            if (v[2] > 0 && int(v[2] * 1000.0f) < maxDepth) {
                float w[3];
                for (int r = 0; r < 3; r++) w[r] = diff[r * 4] * v[0] + diff[r * 4 + 1] * v[1] + diff[r * 4 + 2] * v[2] + diff[r * 4 + 3];
                const int cx = (int)(w[0] * (1 / invfx) / w[2] + cx_), cy = (int)(w[1] * (1 / invfy) / w[2] + cy_);
                if (cx >= 0 && cy >= 0 && cx < width && cy < height) {
                    const unsigned char* f = &fernRgb[((size_t)cy * width + cx) * 3];
                    if (f[0] > 0 || f[1] > 0 || f[2] > 0) {
                        const unsigned char* p = &imgSmall[k * 3];
                        photoSum += std::abs((int)f[0] - (int)p[0]);
                        photoSum += std::abs((int)f[1] - (int)p[1]);
                        photoSum += std::abs((int)f[2] - (int)p[2]);
                        photoCount++;
                    }
                }
            }
        }
        return photoSum / float(photoCount);   // NaN without a sample, which fails the gate, as in the reference
    }
    static void mulPoint(const Matrix4f& m, const float* v, float* out4)
    {
        for (int r = 0; r < 4; r++) out4[r] = m(r, 0) * v[0] + m(r, 1) * v[1] + m(r, 2) * v[2] + m(r, 3) * 1.0f;
    }
    void readBack(ifx_t* h)
    {
        const size_t np = (size_t)width * height;
        img_.resize(np * 3); inst_.resize(np * 3); verts_.resize(np * 4); norms_.resize(np * 4);
        const int n = ifx_fern_frame(h, img_.data(), verts_.data(), norms_.data(), inst_.data());
        if (n != (int)np) throw std::runtime_error(std::string("ifx_fern_frame: ") + (n < 0 ? ifx_last_error(h) : "unexpected sample count"));
    }
    // the RGBDOdometry member of the reference (EF/Ferns.cpp:33-38): a handle of fern resolution, ICP weight 100, single scale, created on first use
    void track(const float* mv, const float* mn, const float* cv, const float* cn, float* pose16, float* diag8)
    {
        if (tracker_) { tracker_(mv, mn, cv, cn, pose16, diag8); return; }
        if (trackerUnavailable_) return;   // diag stays zero: lastICPCount = 0 fails the gate, no match
        if (!tracker_handle_) {
            ifx_config c;
            std::memset(&c, 0, sizeof(c));
            c.width = width; c.height = height; c.fx = fx_; c.fy = fy_; c.cx = cx_; c.cy = cy_;
            c.time_delta = 200; c.confidence = 10.f; c.depth_cut = (float)maxDepth / 1000.0f; c.max_depth_processed = 20.f;
            c.icp_weight = 100.f; c.pyramid = 0; c.fast_odom = 0; c.so3 = 0; c.max_surfels = 1024; c.device = device_; c.n_ranks = 1; c.rank = 0;
            if (ifx_create(&c, &tracker_handle_) != IFX_OK) {   // e.g. a 320 x 240 stream: 40 x 30 is not a size the tracker's pyramids take
                tracker_handle_ = nullptr;
                trackerUnavailable_ = true;
                std::fprintf(stderr, "Ferns: no tracker at %d x %d (%s): keyframes are kept, matches cannot be verified\n", width, height, ifx_global_error());
                return;
            }
        }
        if (ifx_track_maps(tracker_handle_, mv, mn, nullptr, cv, cn, nullptr, pose16, diag8) != IFX_OK)
            throw std::runtime_error(std::string("ifx_track_maps: ") + ifx_last_error(tracker_handle_));
    }

    const float fx_, fy_, cx_, cy_;
    const int device_;
    Tracker tracker_;
    ifx_t* tracker_handle_ = nullptr;
    bool trackerUnavailable_ = false;
    std::vector<unsigned char> img_, inst_;
    std::vector<float> verts_, norms_;
};

#endif
