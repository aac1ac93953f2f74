// ifx_host.hpp -- the C++ host side above the C-ABI of libifx.so (include/ifx_c_api.h): the reference's own class surface
// for the hot path, so that its main program (IF/main.cpp:49-330) compiles against these classes with the GUI removed.
//
//   Resolution / Intrinsics      EF/Utils/Resolution.h, EF/Utils/Intrinsics.h (singletons set once by main, IF/main.cpp:46-47)
//   ElasticFusion                EF/ElasticFusion.h:44-330 (constructor arguments, processFrame, getters / setters, savePly)
//   ElasticFusionInterface       IF/map_interface/ElasticFusionInterface.h:42-160 (Init, ProcessFrame, map / id accessors)
//   InstanceFusion               IF/Core/InstanceFusion.h:66-110 (whetherDoSegmentation, ProcessSegmentation, instance tables)
//   MaskSource / MaskReplay      the Mask-RCNN bridge (IF/Core/InstanceFusion.cpp:241-402) replaced by replayed masks
//   LogReader / RawLogReader / PNGLogReader   IF/utilities/{LogReader.h,RawLogReader.cpp,PNGLogReader.cpp}
//
// Citation prefixes as in the C header: EF/ = elasticfusionpublic/Core/src/, IF/ = src/ of the reference tree.
// Header-only, C++17, needs zlib (as the reference's readers do).  No Eigen / OpenCV / Pangolin / CUDA: poses are a plain
// row-major 4x4, images are raw pointers.  Errors: the reference prints and exits; these classes throw std::runtime_error.
// Everything numerical happens in libifx.so on the GPU; there is no CPU fallback here either.
#ifndef IFX_HOST_HPP_
#define IFX_HOST_HPP_

#include <zlib.h>
#include <sys/stat.h>

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <functional>
#include <iomanip>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "ifx_c_api.h"
#include "ifx_jpeg.hpp"

typedef unsigned char* ImagePtr;    // IF/utilities/Types.h
typedef unsigned short* DepthPtr;

// ------------------------------------------------------------------------------------------------ small maths
struct Matrix4f {   // row-major camera-to-world pose; stands in for Eigen::Matrix4f at this boundary
    float m[16];
    static Matrix4f Identity()
    {
        Matrix4f r;
        for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.f : 0.f;
        return r;
    }
    float& operator()(int r, int c) { return m[r * 4 + c]; }
    const float& operator()(int r, int c) const { return m[r * 4 + c]; }
    const float* data() const { return m; }
    float* data() { return m; }
};

#include "ifx_deformation.hpp"
#include "ifx_ferns.hpp"
typedef FernsT<Matrix4f> Ferns;   // EF/Ferns.h

// ------------------------------------------------------------------------------------------------ singletons
class Resolution {   // EF/Utils/Resolution.h:27-68
public:
    static const Resolution& getInstance(int width = 0, int height = 0)
    {
        static const Resolution instance(width, height);
        return instance;
    }
    const int& width() const { return w_; }
    const int& height() const { return h_; }
    const int& cols() const { return w_; }
    const int& rows() const { return h_; }
    const int& numPixels() const { return n_; }

private:
    Resolution(int w, int h) : w_(w), h_(h), n_(w * h)
    {
        if (w <= 0 || h <= 0) throw std::runtime_error("Resolution::getInstance: first call must give width and height");
    }
    int w_, h_, n_;
};

class Intrinsics {   // EF/Utils/Intrinsics.h:25-62
public:
    static const Intrinsics& getInstance(float fx = 0, float fy = 0, float cx = 0, float cy = 0)
    {
        static const Intrinsics instance(fx, fy, cx, cy);
        return instance;
    }
    const float& fx() const { return fx_; }
    const float& fy() const { return fy_; }
    const float& cx() const { return cx_; }
    const float& cy() const { return cy_; }

private:
    Intrinsics(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy)
    {
        if (fx == 0 || fy == 0) throw std::runtime_error("Intrinsics::getInstance: first call must give fx, fy, cx, cy");
    }
    float fx_, fy_, cx_, cy_;
};

struct ClassColour {   // IF/utilities/Types.h (name + colour of one instance slot)
    ClassColour() : name(""), r(0), g(0), b(0) {}
    ClassColour(std::string n, int r_, int g_, int b_) : name(std::move(n)), r(r_), g(g_), b(b_) {}
    std::string name;
    int r, g, b;
};

// COCO-81 names the Mask-RCNN bridge indexes with class_ids (IF/Core/InstanceFusion.h:48-64)
inline const char* const* ifx_coco_class_names()
{
    static const char* const names[81] = {
        "BG", "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant", "stop sign",
        "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella",
        "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard",
        "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange",
        "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet", "tv", "laptop",
        "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors",
        "teddy bear", "hair drier", "toothbrush"};
    return names;
}

// ------------------------------------------------------------------------------------------------ spatially sharded map (one process per GPU)
// How this process takes part in a spatially sharded map (BASELINE configurations 4 / 5): `ranks` processes, this one is `rank`, each
// on its own GPU, every one fed the same frames.  The collectives of a frame are enqueued by libifx.so itself on a RCCL communicator
// (ifx_owner_init_comm); what the host has to do is carry rank 0's 128-byte ncclUniqueId to the other ranks.  `idFile` is the transport for hosts without
// MPI: rank 0 writes the id there (atomically: temporary name + rename), the others wait for the file.  ranks = -1: a world of one on the sharded path.
struct Sharding {
    int ranks = 1, rank = 0;
    std::string idFile;
    bool on() const { return ranks > 1 || ranks == -1; }
    // the id: drawn by rank 0 (or a world of one), read from idFile by the others (waits up to timeoutSeconds for it)
    // idFile must be a path of THIS run (a file a previous run left behind would hand the other ranks a dead id, and ncclCommInitRank would wait for ever): the file is
    // 128 id bytes + an 8-byte nonce; `nonce` is a number the launcher gives every rank of one run (ifx_replay --shard-nonce; 0: none).  Rank 0 removes a stale file before
    // anything else; the other ranks accept a file only if its nonce is theirs.  Without a nonce a file that APPEARS (or changes) while a rank waits is this run's; one that was
    // already there at the rank's first look may be a previous run's (rank 0 has not removed it yet) or this run's (this rank started late): it is taken when it is still there,
    // unchanged, after `staleGraceSeconds` AND was written within the last `freshSeconds` (generous: the ranks of one run start within a quarter of an hour of each
    // other; a rank that started more than 2 s after rank 0 wrote the file used to wait 120 s and give up).  An older leftover is never taken: if rank 0 of this run is
    // more than the grace period behind (slow log open, GPU initialisation), the rank keeps waiting for rank 0 to replace the file and, failing that, throws after
    // `timeoutSeconds` -- it does not walk into ncclCommInitRank with a dead id, which blocks for ever.  A leftover YOUNGER than `freshSeconds` can
    // only be told from this run's file by a nonce: nonceFromLauncher() derives one from the job id the common launchers export.
    uint64_t nonce = 0;
    int staleGraceSeconds = 3;
    int freshSeconds = 900;
    // a number every rank of ONE launch shares and other launches do not: FNV-1a of the launcher's job id (torchrun, Slurm, Open MPI / PMIx, MPICH); 0 when none is set
    static uint64_t nonceFromLauncher()
    {
        std::string key;
        for (const char* name : {"IFX_SHARD_NONCE", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "SLURM_STEP_ID", "OMPI_MCA_ess_base_jobid", "PMIX_NAMESPACE", "PMI_JOBID"}) {
            const char* v = std::getenv(name);
            if (v && *v && std::string(v) != "none") { key += name; key += '='; key += v; key += ';'; }
        }
        if (key.empty()) return 0;
        uint64_t hsh = 1469598103934665603ull;
        for (unsigned char ch : key) { hsh ^= ch; hsh *= 1099511628211ull; }
        return hsh ? hsh : 1;
    }
    std::vector<uint8_t> uniqueId(int timeoutSeconds = 120) const
    {
        std::vector<uint8_t> id(128, 0);
        if (ranks == -1 || rank == 0) {
            if (ranks > 1) {
                if (idFile.empty()) throw std::runtime_error("Sharding: idFile is needed to hand the ncclUniqueId to the other ranks");
                std::remove(idFile.c_str());   // whatever a previous run left
            }
            if (ifx_comm_unique_id(id.data()) != IFX_OK) throw std::runtime_error("ifx_comm_unique_id failed (RCCL not loadable?)");
            if (ranks > 1) {
                const std::string tmp = idFile + ".tmp";
                { std::ofstream f(tmp, std::ios::binary); f.write((const char*)id.data(), 128); f.write((const char*)&nonce, 8); if (!f) throw std::runtime_error("cannot write " + tmp); }
                if (std::rename(tmp.c_str(), idFile.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
            }
            return id;
        }
        if (idFile.empty()) throw std::runtime_error("Sharding: idFile is needed to receive the ncclUniqueId of rank 0");
        std::vector<uint8_t> first_seen;   // the file's 136 bytes at this rank's first look (empty: there was none)
        bool looked = false;
        for (int waited = 0; waited < timeoutSeconds * 20; waited++) {
            std::ifstream f(idFile, std::ios::binary);
            uint64_t got = 0;
            const bool whole = f && f.read((char*)id.data(), 128) && f.gcount() == 128 && f.read((char*)&got, 8) && f.gcount() == 8;
            if (whole && got == nonce) {
                if (nonce != 0) return id;
                std::vector<uint8_t> now(id);
                now.insert(now.end(), (const uint8_t*)&got, (const uint8_t*)&got + 8);
                if (!looked) first_seen = now;                                         // it was there before this rank looked: this run's, or a previous run's
                else if (now != first_seen) return id;                                  // appeared or changed while waiting: rank 0 of THIS run wrote it
                if (waited >= staleGraceSeconds * 20) {                                // still there, unchanged: rank 0 would have removed a stale one long ago --
                    struct stat sb;                                                     // unless rank 0 is late itself: only a RECENT file can be this run's
                    if (::stat(idFile.c_str(), &sb) == 0 && std::difftime(std::time(nullptr), sb.st_mtime) <= (double)freshSeconds) return id;
                }
            }
            looked = true;
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        throw std::runtime_error("Sharding: no ncclUniqueId of this run in " + idFile + " after waiting (stale file? give every rank the same --shard-nonce)");
    }
};

// ------------------------------------------------------------------------------------------------ ElasticFusion
class ElasticFusion {
public:
    // EF/ElasticFusion.h:47-62, same order and defaults.  closeLoops turns the local loop-closure DETECTION on (ifx_set_loop_closure with
    // countThresh / errThresh / covThresh: INACTIVE prediction + model-to-model tracking + gates; candidates are counted, the deformation
    // they trigger in the reference is outside the path) and the fern data base (ferns(500, depthCut * 1000, photoThresh),
    // EF/ElasticFusion.cpp:49): findFrame every frame inside the handle's fern callback, addFrame with fernThresh after every frame.  A fern match
    // needs the reference's global graph optimiser to become a deformation: that is the caller's (setFernHandler).  reloc is kept for the getter only.
    ElasticFusion(const int timeDelta = 200, const int countThresh = 35000, const float errThresh = 5e-05, const float covThresh = 1e-05,
                  const bool closeLoops = true, const bool iclnuim = false, const bool reloc = false, const float photoThresh = 115,
                  const float confidence = 10, const float depthCut = 3, const float icpThresh = 10, const bool fastOdom = false,
                  const float fernThresh = 0.3095, const bool so3 = true, const bool frameToFrameRGB = false, const std::string fileName = "",
                  const int maxSurfels = 6 * 1000 * 1000, const int device = 0, const Sharding& sharding = Sharding())
        : saveFilename(fileName), closeLoops_(closeLoops), iclnuim_(iclnuim), reloc_(reloc), frameToFrameRGB_(frameToFrameRGB),
          countThresh_(countThresh), errThresh_(errThresh), covThresh_(covThresh), photoThresh_(photoThresh), fernThresh_(fernThresh), sharding_(sharding)
    {
        if (frameToFrameRGB) throw std::runtime_error("ElasticFusion: frameToFrameRGB is not part of the MI355X path");
        if (sharding.on() && closeLoops) throw std::runtime_error("ElasticFusion: loop closure is not available on a spatially sharded map (pass closeLoops = false)");
        std::memset(&cfg_, 0, sizeof(cfg_));
        cfg_.width = Resolution::getInstance().width();
        cfg_.height = Resolution::getInstance().height();
        cfg_.fx = Intrinsics::getInstance().fx();
        cfg_.fy = Intrinsics::getInstance().fy();
        cfg_.cx = Intrinsics::getInstance().cx();
        cfg_.cy = Intrinsics::getInstance().cy();
        cfg_.time_delta = timeDelta;
        cfg_.confidence = confidence;
        cfg_.depth_cut = depthCut;
        cfg_.max_depth_processed = 20.f;   // EF/ElasticFusion.cpp:73
        cfg_.icp_weight = icpThresh;
        cfg_.pyramid = 1;
        cfg_.fast_odom = fastOdom ? 1 : 0;
        cfg_.so3 = so3 ? 1 : 0;
        cfg_.max_surfels = maxSurfels;
        cfg_.device = device;
        cfg_.n_ranks = sharding.on() ? sharding.ranks : 1;
        cfg_.rank = sharding.on() ? sharding.rank : 0;
        if (ifx_create(&cfg_, &h_) != IFX_OK) throw std::runtime_error(std::string("ifx_create: ") + ifx_global_error());
        if (sharding.on()) {   // the library gets its communicator here; from now on a frame is one call, its exchanges enqueued by libifx.so (csrc/ifx_comm.hip)
            const std::vector<uint8_t> id = sharding.uniqueId();
            if (ifx_owner_init_comm(h_, id.data()) != IFX_OK) {
                const std::string e = ifx_last_error(h_);
                ifx_destroy(h_);
                throw std::runtime_error("ifx_owner_init_comm: " + e);
            }
        }
        if (closeLoops && ifx_set_loop_closure(h_, 1, countThresh, errThresh, covThresh) != IFX_OK) {
            const std::string e = ifx_last_error(h_);
            ifx_destroy(h_);
            throw std::runtime_error("ifx_set_loop_closure: " + e);
        }
        if (closeLoops) {
            if (ifx_set_loop_closure_callback(h_, &ElasticFusion::lcTrampoline, this) != IFX_OK) {
                const std::string e = ifx_last_error(h_);
                ifx_destroy(h_);
                throw std::runtime_error("ifx_set_loop_closure_callback: " + e);
            }
            ferns_.reset(new Ferns(500, (int)(depthCut * 1000), photoThresh, cfg_.width, cfg_.height, cfg_.fx, cfg_.fy, cfg_.cx, cfg_.cy, device));
            if (ifx_set_fern_callback(h_, &ElasticFusion::fernTrampoline, this) != IFX_OK) {
                const std::string e = ifx_last_error(h_);
                ifx_destroy(h_);
                throw std::runtime_error("ifx_set_fern_callback: " + e);
            }
        }
        currPose_ = Matrix4f::Identity();
    }
    ElasticFusion(const ElasticFusion&) = delete;
    ElasticFusion& operator=(const ElasticFusion&) = delete;

    // EF/ElasticFusion.cpp:99-136: the destructor writes the trajectory when a file name was given
    virtual ~ElasticFusion()
    {
        if (h_) {
            try {
                if (!saveFilename.empty()) savePoses();
            } catch (...) {
            }
            ifx_destroy(h_);
        }
    }

    // EF/ElasticFusion.h:75-82.  smallInstanceTable (96 x 5) is handed to Ferns::findFrame, whose only writer of it is disabled in the reference.  instanceGT (H x W bytes,
    // ScanNet ground truth) is stored per new surfel for InstanceFusion::evaluateAndSave.  bootstrap: inPose is the tracker's initial guess (currPose * inPose), not a replacement.
    void processFrame(const unsigned char* rgb, const unsigned short* depth, const int64_t& timestamp, int* smallInstanceTable,
                      const unsigned char* instanceGT = NULL, const Matrix4f* inPose = 0, const float weightMultiplier = 1.f,
                      const bool bootstrap = false)
    {
        smallInstanceTable_ = smallInstanceTable;
        if (instanceGT || hadInstanceGT_) {   // EF/ElasticFusion.cpp:285-291: the ground-truth instance image of this frame (new surfels remember the id under their pixel)
            if (ifx_set_instance_gt(h_, instanceGT) != IFX_OK) throw std::runtime_error(std::string("ifx_set_instance_gt: ") + ifx_last_error(h_));
            hadInstanceGT_ = instanceGT != NULL;
        }
        if (bootstrap && !inPose) throw std::runtime_error("ElasticFusion::processFrame: bootstrap needs inPose");   // assert(inPose), EF/ElasticFusion.cpp:354
        int r;
        if (sharding_.on()) {   // every rank is fed the same frame; poses, images and -- merged by creation number -- the map equal the unsharded run
            if (inPose || bootstrap || weightMultiplier != 1.f) throw std::runtime_error("ElasticFusion::processFrame: inPose / bootstrap / weightMultiplier are not offered on a sharded map");
            r = ifx_owner_process_frame(h_, rgb, depth, timestamp, currPose_.data());
            if (r < 0) throw std::runtime_error(std::string("ifx_owner_process_frame: ") + ifx_last_error(h_));
        } else {
            r = ifx_process_frame_ex(h_, rgb, depth, timestamp, smallInstanceTable, inPose ? inPose->data() : nullptr, weightMultiplier, bootstrap ? 1 : 0, currPose_.data());
            if (r < 0) throw std::runtime_error(std::string("ifx_process_frame_ex: ") + ifx_last_error(h_));
        }
        lost_ = (r == 1);
        if (closeLoops_) {   // poseMatches of the reference (EF/ElasticFusion.h:118): here only counted
            float lc[24];
            if (ifx_loop_closure_diag(h_, lc) == IFX_OK) loopCandidates_ = (int)lc[23];
        }
        if (ferns_ && !lost_) {   // processFerns(), EF/ElasticFusion.cpp:713-727: the read-back is enqueued behind the frame, the data base takes it in
            flushPendingFernFrame();   // before the next lookup (flushPendingFernFrame), so the host does not wait for the frame here
            if (ifx_fern_frame_async(h_) < 0) throw std::runtime_error(std::string("ifx_fern_frame_async: ") + ifx_last_error(h_));
            fernPending_ = true; fernPendingPose_ = currPose_; fernPendingTick_ = tick_;
        }
        tick_++;
        poseGraph_.push_back(currPose_);
        poseLogTimes_.push_back(timestamp);
    }

    const Matrix4f& getCurrPose() { return currPose_; }          // EF/ElasticFusion.cpp:1346
    const bool& getLost() { return lost_; }
    const int& getTick() { return tick_; }
    const int& getTimeDelta() { return cfg_.time_delta; }
    const float& getConfidenceThreshold() { return cfg_.confidence; }
    const float& getMaxDepthProcessed() { return cfg_.max_depth_processed; }
    const int& getDeforms() { return deforms_; }                  // local deformations applied (EF/ElasticFusion.h:99)
    // frames whose local loop-closure candidate passed the reference's gates (each would have deformed the map in the reference);
    // 0 at the end of a run = the trajectory and map are what the reference computes with closeLoops and an empty fern data base
    const int& getLoopClosureCandidates() { return loopCandidates_; }
    const int& getFernDeforms() { return fernDeforms_; }          // global (fern) deformations applied
    // closeLoops deforms the map on an accepted loop closure, as the reference does (localDeformation / globalDeformation, ifx_deformation.hpp); switched
    // off, loop closures are only detected and counted, and every frame is tracked and fused as if none had fired
    void setDeformOnLoopClosure(bool on)
    {
        deformOnLoopClosure_ = on;   // without a consumer of the verdict inside the frame the model-to-model tracker runs on its own stream under the map passes
        const bool need = lcHandler_ || (closeLoops_ && on);
        if (ifx_set_loop_closure_callback(h_, need ? &ElasticFusion::lcTrampoline : nullptr, this) != IFX_OK) throw std::runtime_error(ifx_last_error(h_));
    }
    Deformation& getLocalDeformation() { return localDeformation_; }
    Deformation& getGlobalDeformation() { return globalDeformation_; }
    bool getCloseLoops() const { return closeLoops_; }
    int getCountThresh() const { return countThresh_; }
    float getErrThresh() const { return errThresh_; }
    float getCovThresh() const { return covThresh_; }
    float getPhotoThresh() const { return photoThresh_; }
    float getFernThresh() const { return fernThresh_; }

    // the setters of EF/ElasticFusion.h:141-197 that the path honours; they take effect from the next frame
    void setIcpWeight(const float& val) { option("icp_weight_x1000", (int)std::lround(val * 1000.f)); cfg_.icp_weight = val; }
    void setPyramid(const bool& val) { option("pyramid", val ? 1 : 0); cfg_.pyramid = val; }
    void setFastOdom(const bool& val) { option("fast_odom", val ? 1 : 0); cfg_.fast_odom = val; }
    void setSo3(const bool& val) { option("so3", val ? 1 : 0); cfg_.so3 = val; }

    // ---- hooks of the deformation an accepted local loop-closure candidate triggers (EF/ElasticFusion.cpp:566-613).  The graph optimisation
    // (Deformation::constrain -> DeformationGraph::optimiseGraphSparse, Eigen + cholmod) is host code of the reference and stays with the caller:
    // the handler runs inside processFrame right after the gates and typically does
    //     auto c = ef.loopClosureConstraints();  auto v = ef.sampleGraphModel();      // what resize.vertex/time and sampleGraphModel delivered
    //     ... localDeformation.addConstraint(c.src, c.dst, tick, c.times, ...); localDeformation.constrain(..., rawGraph, ...) ...
    //     ef.setDeformation(rawGraph, false);  ef.adoptEstimatedPose();               // GlobalModel::clean(..., rawGraph, ...), currPose = estPose
    struct LoopClosureCandidate {
        float icpError, icpCount, covMax;
        Matrix4f estPose;
    };
    struct Constraints {
        std::vector<float> src, dst;   // n x 3 each: worldRawPoint, worldModelPoint
        std::vector<int32_t> times;    // n
    };
    // ---- global loop closure (EF/ElasticFusion.cpp:457-514).  Ferns::findFrame runs every frame; when it produced constraints (lastClosest != -1) the
    // handler stands where the reference adds them to globalDeformation and calls constrain(ferns.frames, rawGraph, tick, true, poseGraph, true):
    //     handler(ef, constraints, recoveryPose, fernSrcTime) -> rawGraph (empty = the optimiser refused; the local detection then runs as usual)
    // A non-empty graph is applied by this frame's clean as a fern deformation and currPose becomes the recovery pose.
    typedef std::function<std::vector<float>(ElasticFusion&, const std::vector<Ferns::SurfaceConstraint>&, const Matrix4f& recoveryPose, int fernSrcTime)> FernHandler;
    void setFernHandler(FernHandler fn) { fernHandler_ = std::move(fn); }
    Ferns* ferns()
    {
        flushPendingFernFrame();
        return ferns_.get();
    }
    // Ferns::addFrame of the last processed frame, if it is still outstanding
    void flushPendingFernFrame()
    {
        if (!fernPending_) return;
        fernPending_ = false;
        const size_t np = (size_t)ferns_->width * ferns_->height;
        fernImg_.resize(np * 3); fernInst_.resize(np * 3); fernVerts_.resize(np * 4); fernNorms_.resize(np * 4);
        if (ifx_fern_frame_fetch(h_, fernImg_.data(), fernVerts_.data(), fernNorms_.data(), fernInst_.data()) < 0)
            throw std::runtime_error(std::string("ifx_fern_frame_fetch: ") + ifx_last_error(h_));
        ferns_->addFrameMaps(fernImg_.data(), fernVerts_.data(), fernNorms_.data(), fernInst_.data(), fernPendingPose_, fernPendingTick_, fernThresh_);
    }
    // the reference seeds its fern table with time(0) (EF/Ferns.cpp:52); a fixed seed makes a run repeatable.  Empties the data base.
    void resetFerns(uint32_t seed)
    {
        if (!ferns_) return;
        fernPending_ = false;
        ferns_.reset(new Ferns(500, (int)(cfg_.depth_cut * 1000), photoThresh_, cfg_.width, cfg_.height, cfg_.fx, cfg_.fy, cfg_.cx, cfg_.cy, cfg_.device, seed));
    }
    int getFernMatches() const { return fernMatches_; }     // frames on which findFrame produced constraints
    void adoptPose(const Matrix4f& pose)
    {
        if (ifx_adopt_pose(h_, pose.data()) != IFX_OK) throw std::runtime_error(std::string("ifx_adopt_pose: ") + ifx_last_error(h_));
    }

    void setLoopClosureHandler(std::function<void(ElasticFusion&, const LoopClosureCandidate&)> fn)
    {
        lcHandler_ = std::move(fn);   // replaces the built-in handler (defaultLoopClosure); an empty function restores it
        if (ifx_set_loop_closure_callback(h_, (lcHandler_ || (closeLoops_ && deformOnLoopClosure_)) ? &ElasticFusion::lcTrampoline : nullptr, this) != IFX_OK)
            throw std::runtime_error(ifx_last_error(h_));
    }
    std::vector<float> sampleGraphModel()   // x, y, z, init time of every 5000th surfel (Deformation::sampleGraphModel)
    {
        std::vector<float> v((size_t)(cfg_.max_surfels / 5000 + 2) * 4);
        const int n = ifx_sample_graph_model(h_, v.data(), (int)(v.size() / 4));
        if (n < 0) throw std::runtime_error(std::string("ifx_sample_graph_model: ") + ifx_last_error(h_));
        v.resize((size_t)n * 4);
        return v;
    }
    Constraints loopClosureConstraints()
    {
        const int cap = (cfg_.width / 20) * (cfg_.height / 20);
        Constraints c;
        c.src.resize((size_t)cap * 3); c.dst.resize((size_t)cap * 3); c.times.resize((size_t)cap);
        const int n = ifx_loop_closure_constraints(h_, c.src.data(), c.dst.data(), c.times.data(), cap);
        if (n < 0) throw std::runtime_error(std::string("ifx_loop_closure_constraints: ") + ifx_last_error(h_));
        c.src.resize((size_t)n * 3); c.dst.resize((size_t)n * 3); c.times.resize((size_t)n);
        return c;
    }
    void setDeformation(const std::vector<float>& rawGraph, bool isFern)   // 16 floats per node, as Deformation::constrain fills rawGraph
    {
        if (ifx_set_deformation(h_, rawGraph.data(), (int)(rawGraph.size() / 16), isFern ? 1 : 0) != IFX_OK)
            throw std::runtime_error(std::string("ifx_set_deformation: ") + ifx_last_error(h_));
        (isFern ? fernDeforms_ : deforms_) += !rawGraph.empty();
    }
    void adoptEstimatedPose()
    {
        if (ifx_adopt_estimated_pose(h_) != IFX_OK) throw std::runtime_error(std::string("ifx_adopt_estimated_pose: ") + ifx_last_error(h_));
    }

    int getMapSurfelCount() { return ifx_map_count(h_); }
    ifx_t* handle() { return h_; }
    const ifx_config& config() const { return cfg_; }

    // One surfel as the exporters see it
    struct HostMap {
        int n = 0;
        std::vector<float> pc, nr, col, tm, ic;
    };
    HostMap downloadMap()
    {
        HostMap m;
        const int slots = ifx_map_slots(h_);
        m.pc.resize((size_t)slots * 4);
        m.nr.resize((size_t)slots * 4);
        m.col.resize((size_t)slots * 2);
        m.tm.resize((size_t)slots * 2);
        m.ic.resize((size_t)slots * 4);
        m.n = ifx_map_download(h_, slots, m.pc.data(), m.nr.data(), m.col.data(), m.tm.data(), m.ic.data(), nullptr);
        if (m.n < 0) throw std::runtime_error(std::string("ifx_map_download: ") + ifx_last_error(h_));
        return m;
    }

    // EF/ElasticFusion.cpp:796-894 (ResultModel.ply) and :896-990 (ResultModel_Instance.ply): stable surfels only, normals negated,
    // binary little endian: x y z | r g b | nx ny nz | radius.  Returns the number of vertices written.
    int savePly() { return savePlyTo(saveFilename + ".ply", false); }
    int savePlyInstance() { return savePlyTo(saveFilename + "_Instance.ply", true); }
    int savePlyTo(const std::string& path, bool instanceColours)
    {
        const HostMap m = downloadMap();
        std::vector<int> keep;
        for (int i = 0; i < m.n; i++)
            if (m.pc[(size_t)i * 4 + 3] > cfg_.confidence) keep.push_back(i);
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot write " + path);
        f << "ply\nformat binary_little_endian 1.0\nelement vertex " << keep.size()
          << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
             "property float nx\nproperty float ny\nproperty float nz\nproperty float radius\nend_header\n";
        for (int i : keep) {
            const float* p = &m.pc[(size_t)i * 4];
            const float* q = &m.nr[(size_t)i * 4];
            const int64_t c = (int64_t)m.col[(size_t)i * 2 + (instanceColours ? 1 : 0)];
            const unsigned char rgb[3] = {(unsigned char)((c >> 16) & 0xFF), (unsigned char)((c >> 8) & 0xFF), (unsigned char)(c & 0xFF)};
            const float nn[4] = {-q[0], -q[1], -q[2], q[3]};
            f.write((const char*)p, 12);
            f.write((const char*)rgb, 3);
            f.write((const char*)nn, 16);
        }
        return (int)keep.size();
    }

    // EF/ElasticFusion.cpp:104-128: "<seconds> tx ty tz qx qy qz qw" per frame, stream default float formatting
    void savePoses() { savePosesTo(saveFilename + ".freiburg"); }
    void savePosesTo(const std::string& path)
    {
        std::ofstream f(path);
        if (!f) throw std::runtime_error("cannot write " + path);
        for (size_t i = 0; i < poseGraph_.size(); i++) {
            std::stringstream strs;
            if (iclnuim_) strs << std::setprecision(6) << std::fixed << (double)poseLogTimes_[i] << " ";
            else strs << std::setprecision(6) << std::fixed << (double)poseLogTimes_[i] / 1000000.0 << " ";
            const Matrix4f& P = poseGraph_[i];
            float q[4];
            quaternion(P, q);
            f << strs.str() << P(0, 3) << " " << P(1, 3) << " " << P(2, 3) << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
        }
    }
    const std::vector<Matrix4f>& getPoseGraph() const { return poseGraph_; }

    // Eigen::Quaternionf(Matrix3f) (Shepperd's method); out = x y z w
    static void quaternion(const Matrix4f& R, float* out)
    {
        float t = R(0, 0) + R(1, 1) + R(2, 2);
        if (t > 0) {
            float s = std::sqrt(t + 1.0f);
            out[3] = 0.5f * s;
            s = 0.5f / s;
            out[0] = (R(2, 1) - R(1, 2)) * s;
            out[1] = (R(0, 2) - R(2, 0)) * s;
            out[2] = (R(1, 0) - R(0, 1)) * s;
            return;
        }
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        float s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0f);
        out[i] = 0.5f * s;
        s = 0.5f / s;
        out[3] = (R(k, j) - R(j, k)) * s;
        out[j] = (R(j, i) + R(i, j)) * s;
        out[k] = (R(k, i) + R(i, k)) * s;
    }

    std::string saveFilename;

private:
    static int lcTrampoline(ifx_t*, const float* lc, void* user)
    {
        ElasticFusion* self = static_cast<ElasticFusion*>(user);
        LoopClosureCandidate c;
        c.icpError = lc[2]; c.icpCount = lc[3]; c.covMax = lc[22];
        std::memcpy(c.estPose.data(), lc + 6, 64);
        try {
            if (self->lcHandler_) self->lcHandler_(*self, c);
            else if (self->deformOnLoopClosure_) self->defaultLoopClosure();
        } catch (const std::exception& e) {   // never unwind through the C frames of libifx.so
            std::fprintf(stderr, "loop-closure handler: %s\n", e.what());
            return IFX_E_STATE;
        }
        return IFX_OK;
    }
    std::function<void(ElasticFusion&, const LoopClosureCandidate&)> lcHandler_;
    // inside ifx_process_frame, after predict() at the tracked pose (ifx_set_fern_callback)
    static int fernTrampoline(ifx_t* h, void* user)
    {
        ElasticFusion* self = static_cast<ElasticFusion*>(user);
        try {
            self->flushPendingFernFrame();
            Matrix4f tracked;
            if (ifx_get_pose(h, tracked.data()) < 0) return IFX_E_STATE;
            std::vector<Ferns::SurfaceConstraint> constraints;
            const Matrix4f recoveryPose = self->ferns_->findFrame(constraints, tracked, h, self->smallInstanceTable_, self->tick_, self->lost_);
            if (self->ferns_->lastClosest == -1) return 0;
            self->fernMatches_++;
            const int fernSrcTime = self->ferns_->frames[self->ferns_->lastClosest]->srcTime;
            if (!self->fernHandler_ && !self->deformOnLoopClosure_) return 0;
            const std::vector<float> rawGraph = self->fernHandler_ ? self->fernHandler_(*self, constraints, recoveryPose, fernSrcTime)
                                                                   : self->defaultFernClosure(constraints, fernSrcTime);
            if (rawGraph.empty()) return 0;
            self->setDeformation(rawGraph, true);
            self->adoptPose(recoveryPose);
            return 1;
        } catch (const std::exception& e) {   // never unwind through the C frames of libifx.so
            std::fprintf(stderr, "fern data base: %s\n", e.what());
            return IFX_E_STATE;
        }
    }
    // ---- what the reference does on an accepted loop closure, with its own optimiser (ifx_deformation.hpp)
    std::vector<Deformation::TimedPose> fernPoses()
    {
        std::vector<Deformation::TimedPose> v;
        if (ferns_)
            for (auto& f : ferns_->frames) v.push_back({(uint64_t)f->srcTime, f->pose.data()});
        return v;
    }
    void sampleGraphs()   // EF/ElasticFusion.cpp:704-708 at the end of the previous frame = the map as this frame found it
    {
        localDeformation_.sampleGraphModel(sampleGraphModel());
        globalDeformation_.sampleGraphFrom(localDeformation_);
    }
    // local loop closure, EF/ElasticFusion.cpp:566-613
    void defaultLoopClosure()
    {
        const Constraints c = loopClosureConstraints();
        sampleGraphs();
        for (size_t i = 0; i < c.times.size(); i++) localDeformation_.addConstraint(&c.src[i * 3], &c.dst[i * 3], (uint64_t)tick_, (uint64_t)c.times[i], deforms_ == 0);
        std::vector<Deformation::TimedPose> fp = fernPoses(), none;
        std::vector<float> rawGraph;
        std::vector<Deformation::Constraint> newRelativeCons;
        if (localDeformation_.constrain(fp, rawGraph, tick_, false, none, false, &newRelativeCons)) {
            if (!rawGraph.empty()) setDeformation(rawGraph, false);
            adoptEstimatedPose();
            const size_t step = std::max<size_t>(newRelativeCons.size() / 3, 1);
            for (size_t i = 0; i < newRelativeCons.size(); i += step) relativeCons_.push_back(newRelativeCons[i]);
        }
    }
    // global loop closure, EF/ElasticFusion.cpp:486-512; an empty graph = refused
    std::vector<float> defaultFernClosure(const std::vector<Ferns::SurfaceConstraint>& constraints, int fernSrcTime)
    {
        sampleGraphs();
        for (const auto& c : constraints) globalDeformation_.addConstraint(c.sourcePoint, c.targetPoint, (uint64_t)tick_, (uint64_t)fernSrcTime, true);
        for (const auto& rc : relativeCons_) globalDeformation_.addConstraint(rc);
        std::vector<Deformation::TimedPose> fp = fernPoses(), pg;
        for (size_t i = 0; i < poseGraph_.size(); i++) pg.push_back({(uint64_t)(i + 1), poseGraph_[i].data()});
        std::vector<float> rawGraph;
        if (!globalDeformation_.constrain(fp, rawGraph, tick_, true, pg, true)) rawGraph.clear();
        return rawGraph;
    }
    Deformation localDeformation_, globalDeformation_;
    std::vector<Deformation::Constraint> relativeCons_;
    bool deformOnLoopClosure_ = true;
    int fernDeforms_ = 0;
    std::unique_ptr<Ferns> ferns_;
    bool fernPending_ = false;
    Matrix4f fernPendingPose_;
    int fernPendingTick_ = 0;
    std::vector<unsigned char> fernImg_, fernInst_;
    std::vector<float> fernVerts_, fernNorms_;
    FernHandler fernHandler_;
    int fernMatches_ = 0;
    int* smallInstanceTable_ = nullptr;
    void option(const char* name, int v)
    {
        if (ifx_set_option(h_, name, v) != IFX_OK) throw std::runtime_error(std::string("ifx_set_option(") + name + "): " + ifx_last_error(h_));
    }
    ifx_config cfg_;
    ifx_t* h_ = nullptr;
public:
    const Sharding& sharding() const { return sharding_; }
private:
    Matrix4f currPose_;
    bool lost_ = false;
    int tick_ = 1;   // EF/ElasticFusion.cpp:48
    int deforms_ = 0;
    int loopCandidates_ = 0;
    bool hadInstanceGT_ = false;
    bool closeLoops_, iclnuim_, reloc_, frameToFrameRGB_;
    int countThresh_;
    float errThresh_, covThresh_, photoThresh_, fernThresh_;
    Sharding sharding_;
    std::vector<Matrix4f> poseGraph_;
    std::vector<int64_t> poseLogTimes_;
};

// ------------------------------------------------------------------------------------------------ ElasticFusionInterface
class ElasticFusionInterface {   // IF/map_interface/ElasticFusionInterface.h:42-160
public:
    ElasticFusionInterface() : height_(Resolution::getInstance().height()), width_(Resolution::getInstance().width()) {}
    virtual ~ElasticFusionInterface() {}

    int height() const { return height_; }
    int width() const { return width_; }

    // IF/map_interface/ElasticFusionInterface.cpp:27-58 (the GL context and colour look-up go away; the constants stay)
    // closeLoops / confidence: the reference passes true / 10; exposed for replays that want the plain pipeline or a quicker map
    virtual bool Init(std::vector<ClassColour> class_colour_lookup, int maxSurfels = 6 * 1000 * 1000, int device = 0,
                      const std::string& fileName = "./ResultModel", bool closeLoops = true, float confidence = 10.f, const Sharding& sharding = Sharding())
    {
        class_colour_lookup_ = std::move(class_colour_lookup);
        try {
            elastic_fusion_.reset(new ElasticFusion(200, 35000, 5e-05, 1e-05, closeLoops, false, false, 115, confidence, 12, 10, false, 0.3095, true, false, fileName,
                                                    maxSurfels, device, sharding));
        } catch (const std::exception& e) {
            std::fprintf(stderr, "ElasticFusionInterface::Init: %s\n", e.what());
            return false;
        }
        surfel_ids_.resize((size_t)width_ * height_);
        initialised_ = true;
        return true;
    }

    // IF/map_interface/ElasticFusionInterface.cpp:63-73: returns !getLost()
    virtual bool ProcessFrame(const ImagePtr rgb, const DepthPtr depth, const int64_t timestamp, int* smallInstanceTable,
                              const unsigned char* instanceGT)
    {
        if (!elastic_fusion_) return false;
        elastic_fusion_->processFrame(rgb, depth, timestamp, smallInstanceTable, instanceGT);
        return !elastic_fusion_->getLost();
    }

    const std::vector<int>& getSurfelIdsAfterFusionCpu()
    {
        const int r = ifx_image_download(handle(), "ids_after", surfel_ids_.data(), (int64_t)surfel_ids_.size() * 4);
        if (r < 0) throw std::runtime_error(std::string("ifx_image_download: ") + ifx_last_error(handle()));
        return surfel_ids_;
    }
    // the R32I texture object becomes a linear H*W int32 device buffer (0 = empty)
    const int32_t* getSurfelIdsAfterFusionGpu() { return ifx_ids_after(handle()); }
    // the 256-B AoS float* into the GL vertex buffer becomes the struct-of-arrays view
    ifx_soa_view getMapSurfelsGpu()
    {
        ifx_soa_view v;
        std::memset(&v, 0, sizeof(v));
        if (elastic_fusion_) ifx_map_view(handle(), &v);
        return v;
    }
    int getMapSurfelCount() { return elastic_fusion_ ? elastic_fusion_->getMapSurfelCount() : 0; }
    Matrix4f getCurrPose() { return elastic_fusion_->getCurrPose(); }
    void setTrackingOnly(const bool) {}   // GUI switch of the reference (IF/main.cpp:131); no counterpart
    void SavePly()
    {
        elastic_fusion_->savePly();
        elastic_fusion_->savePlyInstance();
    }
    ElasticFusion& elasticFusion() { return *elastic_fusion_; }
    ifx_t* handle() { return elastic_fusion_ ? elastic_fusion_->handle() : nullptr; }

private:
    bool initialised_ = false;
    int height_, width_;
    std::unique_ptr<ElasticFusion> elastic_fusion_;
    std::vector<int> surfel_ids_;
    std::vector<ClassColour> class_colour_lookup_;
};

// ------------------------------------------------------------------------------------------------ tiny .npz reader (masks)
namespace ifx_detail {
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct NpyArray {
    std::string descr;            // "|u1", "<i4", "<i8", ...
    std::vector<int64_t> shape;
    std::vector<unsigned char> bytes;
    int64_t size() const
    {
        int64_t n = 1;
        for (int64_t s : shape) n *= s;
        return n;
    }
};

inline std::vector<unsigned char> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<unsigned char> b((size_t)n);
    if (n && !f.read((char*)b.data(), n)) throw std::runtime_error("cannot read " + path);
    return b;
}

// .npy v1/v2/v3 header: magic, version, header length, python dict literal with 'descr', 'fortran_order', 'shape'
inline NpyArray parse_npy(const unsigned char* p, size_t n, const std::string& what)
{
    if (n < 10 || std::memcmp(p, "\x93NUMPY", 6) != 0) throw std::runtime_error(what + ": not a .npy member");
    const int major = p[6];
    size_t hlen, off;
    if (major == 1) { hlen = rd16(p + 8); off = 10; }
    else { hlen = rd32(p + 8); off = 12; }
    if (off + hlen > n) throw std::runtime_error(what + ": truncated .npy header");
    const std::string hdr((const char*)p + off, hlen);
    NpyArray a;
    size_t k = hdr.find("'descr'");
    if (k == std::string::npos) throw std::runtime_error(what + ": no descr");
    k = hdr.find('\'', hdr.find(':', k));
    a.descr = hdr.substr(k + 1, hdr.find('\'', k + 1) - k - 1);
    if (hdr.find("'fortran_order': True") != std::string::npos) throw std::runtime_error(what + ": fortran order not supported");
    k = hdr.find('(', hdr.find("'shape'"));
    const size_t e = hdr.find(')', k);
    std::string sh = hdr.substr(k + 1, e - k - 1);
    std::replace(sh.begin(), sh.end(), ',', ' ');
    std::stringstream ss(sh);
    int64_t v;
    while (ss >> v) a.shape.push_back(v);
    a.bytes.assign(p + off + hlen, p + n);
    return a;
}

// members of a .npz (zip, stored or deflated) found through the central directory
inline bool npz_member(const std::vector<unsigned char>& z, const std::string& name, NpyArray* out, const std::string& what)
{
    if (z.size() < 22) throw std::runtime_error(what + ": not a zip file");
    size_t eocd = std::string::npos;
    for (size_t i = z.size() - 22 + 1; i-- > 0;) {
        if (rd32(&z[i]) == 0x06054b50u) { eocd = i; break; }
        if (z.size() - i > 22 + 65535) break;
    }
    if (eocd == std::string::npos) throw std::runtime_error(what + ": no zip end record");
    const int entries = rd16(&z[eocd + 10]);
    size_t p = rd32(&z[eocd + 16]);
    for (int i = 0; i < entries; i++) {
        if (p + 46 > z.size() || rd32(&z[p]) != 0x02014b50u) throw std::runtime_error(what + ": bad central directory");
        const int method = rd16(&z[p + 10]);
        const uint32_t csize = rd32(&z[p + 20]), usize = rd32(&z[p + 24]);
        const int nlen = rd16(&z[p + 28]), xlen = rd16(&z[p + 30]), clen = rd16(&z[p + 32]);
        const uint32_t lho = rd32(&z[p + 42]);
        const std::string nm((const char*)&z[p + 46], nlen);
        p += 46 + (size_t)nlen + xlen + clen;
        if (nm != name + ".npy") continue;
        if (csize == 0xFFFFFFFFu || usize == 0xFFFFFFFFu || lho == 0xFFFFFFFFu) throw std::runtime_error(what + ": zip64-sized member");
        if (lho + 30 > z.size() || rd32(&z[lho]) != 0x04034b50u) throw std::runtime_error(what + ": bad local header");
        const size_t data = lho + 30 + rd16(&z[lho + 26]) + rd16(&z[lho + 28]);
        if (data + csize > z.size()) throw std::runtime_error(what + ": truncated member");
        if (method == 0) {
            *out = parse_npy(&z[data], usize, what);
        } else if (method == 8) {
            std::vector<unsigned char> u(usize);
            z_stream s;
            std::memset(&s, 0, sizeof(s));
            if (inflateInit2(&s, -15) != Z_OK) throw std::runtime_error("inflateInit2");
            s.next_in = const_cast<unsigned char*>(&z[data]);
            s.avail_in = csize;
            s.next_out = u.data();
            s.avail_out = usize;
            const int r = inflate(&s, Z_FINISH);
            inflateEnd(&s);
            if (r != Z_STREAM_END) throw std::runtime_error(what + ": inflate failed");
            *out = parse_npy(u.data(), usize, what);
        } else
            throw std::runtime_error(what + ": unsupported zip method");
        return true;
    }
    return false;
}
}   // namespace ifx_detail

// ------------------------------------------------------------------------------------------------ mask provider
// What the Mask-RCNN bridge hands to InstanceFusion (IF/Core/InstanceFusion.cpp:322-402): n masks of the frame's size, 255 inside,
// sorted by area descending (build/mask_ori.py:117), and n COCO class indices.
struct MaskResult {
    int n = 0;
    std::vector<unsigned char> masks;   // n x H x W
    std::vector<int32_t> class_ids;     // n
};

class MaskSource {
public:
    virtual ~MaskSource() {}
    // false: the network has nothing for this frame (no segmentation happens)
    virtual bool detect(int frame, const ImagePtr rgb, int width, int height, MaskResult* out) = 0;
};

// Replays DIR/<frame, 6 digits>.npz with `masks` (n x H x W uint8) and `class_ids` (n integers) -- the files tools/run_log.py reads.
class MaskReplay : public MaskSource {
public:
    explicit MaskReplay(std::string dir) : dir_(std::move(dir)) {}
    bool detect(int frame, const ImagePtr, int width, int height, MaskResult* out) override
    {
        char name[32];
        std::snprintf(name, sizeof(name), "/%06d.npz", frame);
        const std::string path = dir_ + name;
        {
            std::ifstream probe(path, std::ios::binary);
            if (!probe) return false;
        }
        const std::vector<unsigned char> z = ifx_detail::read_file(path);
        ifx_detail::NpyArray m, c;
        if (!ifx_detail::npz_member(z, "masks", &m, path) || !ifx_detail::npz_member(z, "class_ids", &c, path))
            throw std::runtime_error(path + ": needs members masks and class_ids");
        if (m.descr != "|u1" || m.shape.size() != 3) throw std::runtime_error(path + ": masks must be uint8 n x H x W");
        // the bridge leaves masks of another shape uninitialised (IF/Core/InstanceFusion.cpp:342-346); a replay file of the wrong size is an error
        if (m.shape[1] != height || m.shape[2] != width) throw std::runtime_error(path + ": mask size differs from the frame size");
        out->n = (int)m.shape[0];
        if ((int64_t)m.bytes.size() < m.size()) throw std::runtime_error(path + ": masks truncated");
        out->masks.assign(m.bytes.begin(), m.bytes.begin() + m.size());
        out->class_ids.resize((size_t)out->n);
        if (c.size() != out->n) throw std::runtime_error(path + ": class_ids length differs from the number of masks");
        for (int i = 0; i < out->n; i++) {
            if (c.descr == "<i4") { int32_t v; std::memcpy(&v, &c.bytes[(size_t)i * 4], 4); out->class_ids[i] = v; }
            else if (c.descr == "<i8") { int64_t v; std::memcpy(&v, &c.bytes[(size_t)i * 8], 8); out->class_ids[i] = (int32_t)v; }
            else throw std::runtime_error(path + ": class_ids must be int32 or int64");
        }
        return true;
    }

private:
    std::string dir_;
};

// ------------------------------------------------------------------------------------------------ InstanceFusion
class InstanceFusion {
public:
    // IF/Core/InstanceFusion.h:68: (instanceNum, width, height, useMulThread, netType).  The worker thread of the bridge
    // (IF/Core/InstanceFusion.cpp:82-109) is unused by the reference's own main (IF/main.cpp:83) and is not reproduced.
    InstanceFusion(int num, int w, int h, bool useMulThread = false, int netType = 0, std::shared_ptr<MaskSource> source = nullptr)
        : instanceNum(num), width(w), height(h), source_(std::move(source))
    {
        (void)netType;
        if (num != IFX_NUM_INSTANCES) throw std::runtime_error("InstanceFusion: libifx.so is built for 96 instance slots");
        if (useMulThread) throw std::runtime_error("InstanceFusion: useMulThread is not supported");
    }
    void setMaskSource(std::shared_ptr<MaskSource> s) { source_ = std::move(s); }
    void setSuperpixelRefinement(bool on) { superpixels_ = on; }   // processInstance steps -1_1..-1_3 (IF/Core/InstanceFusion.cpp:722-738)

    int getInstanceNum() { return instanceNum; }
    int getSurfelSize() { return 64; }   // floats per surfel record of the reference (EF/Shaders/Vertex.cpp:48); the store here is SoA

    // IF/Core/InstanceFusion.cpp:192-238
    bool whetherDoSegmentation(const std::unique_ptr<ElasticFusionInterface>& map, int frame_num)
    {
        const int r = ifx_should_segment(map->handle(), frame_num);
        if (r < 0) throw std::runtime_error(std::string("ifx_should_segment: ") + ifx_last_error(map->handle()));
        return r == 1;
    }

    // IF/Core/InstanceFusion.cpp:241-270: detect() + getCurrentResults() -> processInstance()
    void ProcessSegmentation(const ImagePtr rgb, const DepthPtr depth, const std::unique_ptr<ElasticFusionInterface>& map, int frame_num, bool isflann)
    {
        if (!source_) throw std::runtime_error("InstanceFusion::ProcessSegmentation: no mask source");
        MaskResult res;
        if (!source_->detect(frame_num, rgb, width, height, &res)) return;
        const int flags = (isflann ? 1 : 0) | (superpixels_ ? 2 : 0);
        const bool sharded = map->elasticFusion().sharding().on();   // the owners' boxes / model depth / eviction statistics are merged inside the library
        const int r = sharded ? ifx_owner_process_segmentation(map->handle(), rgb, depth, res.masks.data(), res.class_ids.data(), res.n, frame_num, flags)
                              : ifx_process_segmentation(map->handle(), rgb, depth, res.masks.data(), res.class_ids.data(), res.n, frame_num, flags);
        if (r < 0) throw std::runtime_error(std::string(sharded ? "ifx_owner_process_segmentation: " : "ifx_process_segmentation: ") + ifx_last_error(map->handle()));
        handle_ = map->handle();
        segmentations_++;
    }

    // IF/Core/InstanceTable.cpp:336-445: precision / recall against the ground-truth ids the surfels carry (processFrame's instanceGT); appends to `path` the
    // rows the reference appends to ./temp/Precision_Recall_RAW.txt.  inUse: instance slots in use (+ evicted ones) as the reference's summary row reports them.
    void evaluateAndSave(const std::unique_ptr<ElasticFusionInterface>& map, const std::string& fileName, const std::string& path = "./Precision_Recall_RAW.txt")
    {
        std::vector<int32_t> instPointNum((size_t)instanceNum), gtPointNum(256), inst_gt_Map((size_t)256 * instanceNum);
        if (ifx_precision_recall(map->handle(), instPointNum.data(), gtPointNum.data(), inst_gt_Map.data()) != IFX_OK)
            throw std::runtime_error(std::string("ifx_precision_recall: ") + ifx_last_error(map->handle()));
        bool hasFile = false;
        {
            std::ifstream probe(path);
            hasFile = probe.good() && probe.peek() != std::ifstream::traits_type::eof();
        }
        std::ofstream ff(path, std::ios::app);
        if (!ff) throw std::runtime_error("cannot write " + path);
        if (!hasFile) ff << "fileName,class,gtPointNum,instPointNum,intersectionNum";
        ff << std::endl;
        int howManyGT = 0, inUse = 0;
        for (int gtID = 0; gtID < 256; gtID++) howManyGT += gtPointNum[(size_t)gtID] > 6000;
        const std::vector<ClassColour> table = getInstanceTable();
        for (const ClassColour& c : table) inUse += !c.name.empty();
        ff << ",,,,,,," << howManyGT << "," << inUse << std::endl;
        for (int gtID = 0; gtID < 256; gtID++) {
            if (gtPointNum[(size_t)gtID] <= 150) continue;
            int maxInstanceInGTNum = 0, maxInstanceID = -1;
            for (int i = 0; i < instanceNum; i++)
                if (inst_gt_Map[(size_t)gtID * instanceNum + i] > maxInstanceInGTNum) { maxInstanceInGTNum = inst_gt_Map[(size_t)gtID * instanceNum + i]; maxInstanceID = i; }
            if (maxInstanceInGTNum <= 150) continue;
            bool flag = true;
            for (int other = 0; other < 256; other++)
                if (inst_gt_Map[(size_t)other * instanceNum + maxInstanceID] > inst_gt_Map[(size_t)gtID * instanceNum + maxInstanceID]) flag = false;
            if (flag)
                ff << fileName << "," << table[(size_t)maxInstanceID].name << "," << gtPointNum[(size_t)gtID] << "," << instPointNum[(size_t)maxInstanceID] << ","
                   << inst_gt_Map[(size_t)gtID * instanceNum + maxInstanceID] << std::endl;
        }
    }

    // IF/Core/InstanceFusion.cpp:1232-1252 (the image the GUI shows as "segmentation"; the 2-D boxes are not drawn): H x W x 4 floats
    const std::vector<float>& renderProjectMap(const std::unique_ptr<ElasticFusionInterface>& map, bool /*drawBBox*/ = false)
    {
        projectColor_.resize((size_t)width * height * 4);
        if (ifx_render_project_map(map->handle(), projectColor_.data(), nullptr) != IFX_OK) throw std::runtime_error(std::string("ifx_render_project_map: ") + ifx_last_error(map->handle()));
        return projectColor_;
    }
    const std::vector<float>& getProjectColorMap() const { return projectColor_; }

    // IF/Core/InstanceFusion.h:87: one entry per instance slot; name = COCO class of the slot ("" = unused slot)
    std::vector<ClassColour> getInstanceTable()
    {
        std::vector<ClassColour> t((size_t)instanceNum);
        if (!handle_) return t;
        int32_t lc[IFX_NUM_INSTANCES * 5];
        if (ifx_loop_closure_instance_table(handle_, lc) != IFX_OK) throw std::runtime_error(ifx_last_error(handle_));
        for (int i = 0; i < instanceNum; i++) {
            const int cls = lc[i * 5 + 3];
            t[(size_t)i] = ClassColour(cls >= 0 && cls < 81 ? ifx_coco_class_names()[cls] : "", lc[i * 5 + 0], lc[i * 5 + 1], lc[i * 5 + 2]);
        }
        return t;
    }
    // IF/Core/InstanceTable.cpp:98-121: int[96*5] = r, g, b, class, index
    void getLoopClosureInstanceTable(int* out_table)
    {
        if (!handle_) {
            for (int i = 0; i < instanceNum * 5; i++) out_table[i] = (i % 5 == 4) ? i / 5 : ((i % 5 == 3) ? -1 : 0);
            return;
        }
        if (ifx_loop_closure_instance_table(handle_, out_table) != IFX_OK) throw std::runtime_error(ifx_last_error(handle_));
    }
    void bindMap(const std::unique_ptr<ElasticFusionInterface>& map) { handle_ = map->handle(); }
    // bestIDInEachSurfel (IF/Core/InstanceFusionCuda.cu:1158-1200) of the live surfels, map order
    std::vector<int32_t> getSurfelLabels(const std::unique_ptr<ElasticFusionInterface>& map)
    {
        std::vector<int32_t> l((size_t)ifx_map_slots(map->handle()));
        const int n = ifx_labels(map->handle(), l.data(), (int)l.size());
        if (n < 0) throw std::runtime_error(std::string("ifx_labels: ") + ifx_last_error(map->handle()));
        l.resize((size_t)n);
        return l;
    }
    int segmentationCalls() const { return segmentations_; }

    // IF/Core/InstanceFusion.cpp:1261-1457: 3-D boxes of the instances (map3DBBox: 6 floats per instance, ground frame when bboxType) + the frames
    void computeMapBoundingBox(const std::unique_ptr<ElasticFusionInterface>& map, bool bboxType)
    {
        map3DBBox.assign((size_t)instanceNum * 6, 0.f);
        instcMatrix.assign((size_t)instanceNum * 16, 0.f);
        if (ifx_map_bounding_boxes(map->handle(), bboxType ? 1 : 0, ratio3DBBox, map3DBBox.data(), groundNormal, gcMatrix, instcMatrix.data(), nullptr) < 0)
            throw std::runtime_error(std::string("ifx_map_bounding_boxes: ") + ifx_last_error(map->handle()));
    }
    // IF/Core/InstanceFusion.cpp:1459-1590: the surfels of every instance as {slot, position and normal in the box frame, r, g, b} records
    void getInstancePointCloud(const std::unique_ptr<ElasticFusionInterface>& map, bool bboxType)
    {
        std::vector<int32_t> counts((size_t)instanceNum, 0);
        if (ifx_instance_point_cloud(map->handle(), bboxType ? 1 : 0, counts.data(), -1, nullptr, 0) < 0) throw std::runtime_error(std::string("ifx_instance_point_cloud: ") + ifx_last_error(map->handle()));
        instSurfels.assign((size_t)instanceNum, std::vector<float>());
        for (int q = 0; q < instanceNum; q++) {
            if (counts[q] <= 0) continue;
            instSurfels[q].resize((size_t)counts[q] * 10);
            const int n = ifx_instance_point_cloud(map->handle(), bboxType ? 1 : 0, counts.data(), q, instSurfels[q].data(), counts[q]);
            if (n < 0) throw std::runtime_error(std::string("ifx_instance_point_cloud: ") + ifx_last_error(map->handle()));
            instSurfels[q].resize((size_t)n * 10);
        }
    }
    const float ratio3DBBox = 1000000;                      // IF/Core/InstanceFusion.h:250
    std::vector<float> map3DBBox, instcMatrix;
    float groundNormal[3] = {0, -1, 0}, gcMatrix[16] = {0};
    std::vector<std::vector<float>> instSurfels;

    // stage methods, same names as the reference (IF/Core/InstanceFusion_superpixel.cpp:713-772, :40-225, :651-710; IF/Core/InstanceFusion.cpp:470-593)
    int gSLICrInterface(const std::unique_ptr<ElasticFusionInterface>& map, const ImagePtr rgb, int* segMask) { return ifx_slic_segment(map->handle(), rgb, segMask); }
    int mergeSuperPixel(const std::unique_ptr<ElasticFusionInterface>& map, const DepthPtr depth, int* segMask, int* finalSPixel)
    {
        return ifx_merge_superpixels(map->handle(), depth, segMask, finalSPixel, nullptr);
    }
    int maskSuperPixelFilter_OverSeg(const std::unique_ptr<ElasticFusionInterface>& map, const int* finalSPixel, unsigned char* masks, int n)
    {
        return ifx_mask_superpixel_filter(map->handle(), finalSPixel, masks, n);
    }
    int maskGeometricFilter(const std::unique_ptr<ElasticFusionInterface>& map, const DepthPtr projectDepthMap, unsigned char* masks, const unsigned char* oriMasks, int n,
                            unsigned char* unavailable)
    {
        return ifx_mask_geometric_filter(map->handle(), projectDepthMap, masks, oriMasks, n, unavailable);
    }

private:
    int instanceNum, width, height;
    std::shared_ptr<MaskSource> source_;
    ifx_t* handle_ = nullptr;
    bool superpixels_ = true;
    int segmentations_ = 0;
    std::vector<float> projectColor_;
};

// ------------------------------------------------------------------------------------------------ log readers
class LogReader {   // IF/utilities/LogReader.h:30-100
public:
    LogReader(std::string file, bool flipColors)
        : flipColors(flipColors), timestamp(0), depth(nullptr), rgb(nullptr), currentFrame(0), file(std::move(file)),
          width(Resolution::getInstance().width()), height(Resolution::getInstance().height()), numPixels(width * height)
    {
    }
    virtual ~LogReader() {}
    virtual void getNext() = 0;
    virtual int getNumFrames() = 0;
    virtual bool hasMore() = 0;
    virtual bool rewound() = 0;
    virtual void getBack() = 0;
    virtual void fastForward(int frame) = 0;
    virtual const std::string getFile() { return file; }
    virtual void setAuto(bool) {}
    // (beyond the reference) the frame the NEXT getNext() will deliver, if the reader already holds it decoded: the buffers getNext() will publish, so that a caller can
    // announce them (ifx_hint_next_frame) before it processes the current frame.  false: no look-ahead in this reader / at the end of the log.
    virtual bool peekNext(const unsigned char*& /*rgbNext*/, const unsigned short*& /*depthNext*/) { return false; }

    bool flipColors;
    int64_t timestamp;
    unsigned short* depth;
    unsigned char* rgb;
    int currentFrame;

protected:
    const std::string file;
    int width, height, numPixels;
};

// IF/utilities/RawLogReader.cpp: int32 frame count; per frame int64 timestamp, int32 depthSize, int32 imageSize, depth (raw or zlib),
// colour (raw or baseline JPEG).
class RawLogReader : public LogReader {
public:
    RawLogReader(std::string file, bool flipColors) : LogReader(std::move(file), flipColors)
    {
        fp_ = std::fopen(this->file.c_str(), "rb");
        if (!fp_) throw std::runtime_error("cannot open " + this->file);
        int32_t n = 0;
        if (std::fread(&n, 4, 1, fp_) != 1) throw std::runtime_error(this->file + ": empty log");
        numFrames_ = n;
        cur_.depth.resize((size_t)numPixels);
        cur_.rgb.resize((size_t)numPixels * 3);
        depth = cur_.depth.data();
        rgb = cur_.rgb.data();
    }
    ~RawLogReader() override
    {
        stopReadAhead();
        if (fp_) std::fclose(fp_);
    }
    int getNumFrames() override { return numFrames_; }
    bool hasMore() override { return currentFrame + 1 < numFrames_; }   // RawLogReader.cpp:134-137: the last frame is never delivered
    bool rewound() override { return false; }

    // Decoding ahead: the records of the next `frames` frames are read and inflated / JPEG-decoded by `threads` workers while the caller processes the
    // current one (a 640 x 480 zlib + JPEG record takes ~5 ms on one core, a frame on the GPU ~1 ms).  Same frames in the same order; getBack and
    // fastForward fall back to the synchronous path.
    void setReadAhead(int frames, int threads)
    {
        stopReadAhead();
        if (frames <= 0 || threads <= 0) return;
        ring_.resize((size_t)frames);
        for (Frame& f : ring_) { f.depth.resize((size_t)numPixels); f.rgb.resize((size_t)numPixels * 3); f.state = 0; }
        head_ = tail_ = 0;
        nextToRead_ = currentFrame + 1;   // the value currentFrame takes when that frame is delivered
        quit_ = false;
        for (int i = 0; i < threads; i++) workers_.emplace_back([this] { work(); });
    }
    void getNext() override
    {
        if (workers_.empty()) {
            filePointers_.push_back(std::ftell(fp_));
            readRecord(cur_);
            decode(cur_);
            publish();
            return;
        }
        fill();
        Frame* f;
        {
            std::unique_lock<std::mutex> lk(mu_);
            f = &ring_[head_ % ring_.size()];
            done_.wait(lk, [&] { return f->state >= 2; });
        }
        if (f->state == 3) throw std::runtime_error(f->error);
        filePointers_.push_back(f->filePos);
        std::swap(cur_.depth, f->depth);
        std::swap(cur_.rgb, f->rgb);
        cur_.timestamp = f->timestamp;
        {
            std::lock_guard<std::mutex> lk(mu_);
            f->state = 0;
            head_++;
        }
        publish();
        fill();   // keep the workers busy while the caller processes this frame
    }
    bool peekNext(const unsigned char*& rgbNext, const unsigned short*& depthNext) override
    {
        if (workers_.empty() || !hasMore()) return false;
        fill();
        Frame* f;
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (tail_ == head_) return false;
            f = &ring_[head_ % ring_.size()];
            done_.wait(lk, [&] { return f->state >= 2; });
        }
        if (f->state == 3) return false;   // (getNext() reports the error)
        rgbNext = f->rgb.data();           // getNext() swaps these vectors into the published frame: the same addresses
        depthNext = f->depth.data();
        return true;
    }
    void getBack() override
    {
        if (filePointers_.empty()) throw std::runtime_error("RawLogReader::getBack: at the start");
        stopReadAhead();
        std::fseek(fp_, filePointers_.back(), SEEK_SET);
        filePointers_.pop_back();
        readRecord(cur_);
        decode(cur_);
        publish();
    }
    void fastForward(int frame) override
    {
        stopReadAhead();
        while (currentFrame < frame && hasMore()) {
            filePointers_.push_back(std::ftell(fp_));
            int64_t ts;
            int32_t ds, is;
            if (std::fread(&ts, 8, 1, fp_) != 1 || std::fread(&ds, 4, 1, fp_) != 1 || std::fread(&is, 4, 1, fp_) != 1) throw std::runtime_error(file + ": truncated");
            std::fseek(fp_, (long)ds + is, SEEK_CUR);
            currentFrame++;
        }
    }

private:
    struct Frame {
        long filePos = 0;
        int64_t timestamp = 0;
        std::vector<unsigned char> dz, iz, jpeg;   // the record as stored; scratch of the JPEG decoder
        std::vector<unsigned short> depth;
        std::vector<unsigned char> rgb;
        int state = 0;   // 0 free, 1 queued, 2 decoded, 3 failed
        std::string error;
    };
    void publish()
    {
        depth = cur_.depth.data();
        rgb = cur_.rgb.data();
        timestamp = cur_.timestamp;
        currentFrame++;
    }
    void readRecord(Frame& f)
    {
        int32_t ds = 0, is = 0;
        f.filePos = std::ftell(fp_);
        if (std::fread(&f.timestamp, 8, 1, fp_) != 1 || std::fread(&ds, 4, 1, fp_) != 1) throw std::runtime_error(file + ": truncated frame header");
        // (the record is: timestamp, depth size, image size, depth bytes, image bytes)
        if (std::fread(&is, 4, 1, fp_) != 1 || ds < 0 || is < 0) throw std::runtime_error(file + ": truncated frame header");
        f.dz.resize((size_t)ds);
        f.iz.resize((size_t)is);
        if (ds && std::fread(f.dz.data(), (size_t)ds, 1, fp_) != 1) throw std::runtime_error(file + ": truncated depth");
        if (is && std::fread(f.iz.data(), (size_t)is, 1, fp_) != 1) throw std::runtime_error(file + ": truncated colour");
    }
    void decode(Frame& f) const
    {
        if ((int64_t)f.dz.size() == (int64_t)numPixels * 2)
            std::memcpy(f.depth.data(), f.dz.data(), (size_t)numPixels * 2);
        else {
            uLongf len = (uLongf)numPixels * 2;
            if (uncompress((Bytef*)f.depth.data(), &len, f.dz.data(), (uLong)f.dz.size()) != Z_OK || len != (uLongf)numPixels * 2)
                throw std::runtime_error(file + ": depth does not inflate to the frame size");
        }
        if ((int64_t)f.iz.size() == (int64_t)numPixels * 3)
            std::memcpy(f.rgb.data(), f.iz.data(), (size_t)numPixels * 3);
        else if (!f.iz.empty()) {   // RawLogReader.cpp:96-106: JPEG (cvDecodeImage in the reference; libjpeg's default decompression path restated in ifx_jpeg.hpp)
            int jw = 0, jh = 0;
            ifx_jpeg::decode(f.iz.data(), f.iz.size(), f.jpeg, jw, jh);
            if (jw != width || jh != height) throw std::runtime_error(file + ": JPEG frame size differs from Resolution");
            std::memcpy(f.rgb.data(), f.jpeg.data(), (size_t)numPixels * 3);
        } else
            std::memset(f.rgb.data(), 0, (size_t)numPixels * 3);   // RawLogReader.cpp:107-110
        if (flipColors)
            for (int i = 0; i < numPixels; i++) std::swap(f.rgb[(size_t)i * 3], f.rgb[(size_t)i * 3 + 2]);
    }
    // reads the records of the frames to come into free ring slots (the consumer thread does the file I/O, the workers the decoding)
    void fill()
    {
        while (nextToRead_ < numFrames_) {   // frames 1 .. numFrames - 1 are delivered (hasMore)
            Frame* f;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (tail_ - head_ >= ring_.size()) return;
                f = &ring_[tail_ % ring_.size()];
            }
            readRecord(*f);
            nextToRead_++;
            {
                std::lock_guard<std::mutex> lk(mu_);
                f->state = 1;
                tail_++;
            }
            work_.notify_one();
        }
    }
    void work()
    {
        for (;;) {
            Frame* f = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                work_.wait(lk, [&] { return quit_ || next_ < tail_; });
                if (quit_) return;
                f = &ring_[next_ % ring_.size()];
                next_++;
            }
            int st = 2;
            try {
                decode(*f);
            } catch (const std::exception& e) {
                f->error = e.what();
                st = 3;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                f->state = st;
            }
            done_.notify_all();
        }
    }
    // ends the read-ahead and puts the file position behind the last frame the caller has seen
    void stopReadAhead()
    {
        if (workers_.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        work_.notify_all();
        for (std::thread& t : workers_) t.join();
        workers_.clear();
        if (tail_ > head_) std::fseek(fp_, ring_[head_ % ring_.size()].filePos, SEEK_SET);
        ring_.clear();
        head_ = tail_ = next_ = 0;
    }
    std::FILE* fp_ = nullptr;
    int numFrames_ = 0;
    std::vector<long> filePointers_;
    Frame cur_;
    std::vector<Frame> ring_;
    size_t head_ = 0, tail_ = 0, next_ = 0;   // consumed / read / handed to a worker
    int nextToRead_ = 0;
    bool quit_ = false;
    std::mutex mu_;
    std::condition_variable work_, done_;
    std::vector<std::thread> workers_;
};

namespace ifx_detail {
// Minimal PNG decoder for what RGB-D data sets hold: 8-bit RGB / RGBA / grey and 16-bit grey, non-interlaced (zlib does the inflating).
struct PngImage {
    int w = 0, h = 0, channels = 0, bits = 0;
    std::vector<unsigned char> data;   // rows of w*channels samples; 16-bit samples big endian as in the file
};
inline PngImage decode_png(const std::vector<unsigned char>& f, const std::string& what)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) throw std::runtime_error(what + ": not a PNG file");
    PngImage im;
    std::vector<unsigned char> idat;
    int ctype = -1, interlace = 0;
    size_t p = 8;
    while (p + 12 <= f.size()) {
        const uint32_t len = ((uint32_t)f[p] << 24) | ((uint32_t)f[p + 1] << 16) | ((uint32_t)f[p + 2] << 8) | f[p + 3];
        const std::string type((const char*)&f[p + 4], 4);
        const unsigned char* d = &f[p + 8];
        if (p + 12 + len > f.size()) throw std::runtime_error(what + ": truncated chunk");
        if (type == "IHDR") {
            im.w = (int)(((uint32_t)d[0] << 24) | ((uint32_t)d[1] << 16) | ((uint32_t)d[2] << 8) | d[3]);
            im.h = (int)(((uint32_t)d[4] << 24) | ((uint32_t)d[5] << 16) | ((uint32_t)d[6] << 8) | d[7]);
            im.bits = d[8];
            ctype = d[9];
            interlace = d[12];
        } else if (type == "IDAT")
            idat.insert(idat.end(), d, d + len);
        else if (type == "IEND")
            break;
        p += 12 + (size_t)len;
    }
    if (interlace) throw std::runtime_error(what + ": interlaced PNG not supported");
    if (ctype == 0) im.channels = 1;
    else if (ctype == 2) im.channels = 3;
    else if (ctype == 4) im.channels = 2;
    else if (ctype == 6) im.channels = 4;
    else throw std::runtime_error(what + ": palette PNG not supported");
    if (im.bits != 8 && im.bits != 16) throw std::runtime_error(what + ": only 8- and 16-bit PNG samples are supported");
    const size_t bpp = (size_t)im.channels * im.bits / 8, stride = bpp * im.w;
    std::vector<unsigned char> raw((stride + 1) * im.h);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) throw std::runtime_error(what + ": PNG data does not inflate");
    im.data.resize(stride * im.h);
    for (int y = 0; y < im.h; y++) {
        const unsigned char ft = raw[(stride + 1) * y];
        const unsigned char* in = &raw[(stride + 1) * y + 1];
        unsigned char* out = &im.data[stride * y];
        const unsigned char* up = y ? out - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pr = 0;
            switch (ft) {
            case 0: pr = 0; break;
            case 1: pr = a; break;
            case 2: pr = b; break;
            case 3: pr = (a + b) >> 1; break;
            case 4: {
                const int q = a + b - c, pa = std::abs(q - a), pb = std::abs(q - b), pc = std::abs(q - c);
                pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: throw std::runtime_error(what + ": bad PNG filter");
            }
            out[i] = (unsigned char)(in[i] + pr);
        }
    }
    return im;
}
}   // namespace ifx_detail

// IF/utilities/PNGLogReader.cpp:29-212: `data.txt`, one frame per line: "timestamp depth_path rgb_path ...".
class PNGLogReader : public LogReader {
public:
    PNGLogReader(std::string file) : LogReader(std::move(file), true)
    {
        std::ifstream f(this->file);
        if (!f) throw std::runtime_error("cannot open " + this->file);
        const size_t slash = this->file.find_last_of('/');
        const std::string base = slash == std::string::npos ? "" : this->file.substr(0, slash + 1);
        std::string line;
        while (std::getline(f, line)) {
            std::stringstream ss(line);
            Frame fr;
            std::string d, c;
            if (!(ss >> fr.timestamp >> d >> c)) continue;
            fr.depth_path = (d[0] == '/') ? d : base + d;
            fr.rgb_path = (c[0] == '/') ? c : base + c;
            frames_.push_back(fr);
        }
        depthBuf_.resize((size_t)numPixels);
        rgbBuf_.resize((size_t)numPixels * 3);
        depth = depthBuf_.data();
        rgb = rgbBuf_.data();
    }
    int getNumFrames() override { return (int)frames_.size(); }
    bool hasMore() override { return currentFrame < (int)frames_.size(); }   // PNGLogReader.cpp:231-234: lastGot + 1 < size, every frame is delivered
    bool rewound() override { return false; }
    void getBack() override { throw std::runtime_error("PNGLogReader::getBack: not supported"); }   // PNGLogReader.cpp: empty
    void fastForward(int frame) override { currentFrame = std::min(frame, (int)frames_.size()); }
    void getNext() override
    {
        if (currentFrame >= (int)frames_.size()) throw std::runtime_error("PNGLogReader::getNext: past the end");
        const Frame& fr = frames_[(size_t)currentFrame];
        timestamp = fr.timestamp;
        const ifx_detail::PngImage c = ifx_detail::decode_png(ifx_detail::read_file(fr.rgb_path), fr.rgb_path);
        const ifx_detail::PngImage d = ifx_detail::decode_png(ifx_detail::read_file(fr.depth_path), fr.depth_path);
        if (c.w != width || c.h != height || d.w != width || d.h != height) throw std::runtime_error(fr.rgb_path + ": image size differs from Resolution");
        if (c.bits != 8 || c.channels < 3) throw std::runtime_error(fr.rgb_path + ": colour image must be 8-bit RGB(A)");
        if (d.channels != 1) throw std::runtime_error(fr.depth_path + ": depth image must be single channel");
        for (int i = 0; i < numPixels; i++) {
            std::memcpy(&rgbBuf_[(size_t)i * 3], &c.data[(size_t)i * c.channels], 3);   // imread's BGR + flipColors = RGB
            depthBuf_[(size_t)i] = d.bits == 16 ? (unsigned short)((d.data[(size_t)i * 2] << 8) | d.data[(size_t)i * 2 + 1]) : d.data[(size_t)i];
        }
        currentFrame++;
    }

private:
    struct Frame {
        int64_t timestamp;
        std::string depth_path, rgb_path;
    };
    std::vector<Frame> frames_;
    std::vector<unsigned short> depthBuf_;
    std::vector<unsigned char> rgbBuf_;
};

#endif   // IFX_HOST_HPP_
