// ifx_replay -- the reference's main program (IF/main.cpp:49-330) without its GUI, on the C++ host classes of ifx_host.hpp:
// log reader -> ElasticFusionInterface::ProcessFrame -> InstanceFusion::whetherDoSegmentation -> ProcessSegmentation (masks replayed
// from files instead of the Mask-RCNN bridge, kNN smoothing when the last one is more than flann_skip_frames old) -> at the end
// ResultModel.freiburg (ElasticFusion's destructor, EF/ElasticFusion.cpp:99-136), ResultModel.ply and ResultModel_Instance.ply
// (map->SavePly(), IF/main.cpp:300-305), and optionally the per-surfel instance labels.
//
//   ifx_replay LOG.klg|data.txt [--width 640 --height 480 --fx 528 --fy 528 --cx 320 --cy 240] [--masks DIR] [--out PREFIX]
//              [--max-frames N] [--max-surfels N] [--no-superpixels] [--labels FILE] [--flip-colors] [--device K] [--gt-dir DIR --eval FILE]
#include <chrono>
#include <cstdlib>
#include <iostream>

#include "ifx_host.hpp"

namespace {
const int instanceNum = 96;        // IF/main.cpp:31
const int cnn_start_frames = 0;    // IF/main.cpp:34-36 (cnn_skip_frames is dead once whetherDoSegmentation decides)
int flann_skip_frames = 40;       // IF/main.cpp:36; --flann-every overrides it

struct Args {
    std::string log, masks, out = "./ResultModel", labels, gt_dir, eval_file;
    int width = 640, height = 480, max_frames = 0, max_surfels = 6 * 1000 * 1000, device = 0;
    float fx = 528.f, fy = 528.f, cx = 320.f, cy = 240.f;
    bool superpixels = true, flip = false, close_loops = true, deform = true, lookahead = true;
    int decode_threads = 4;
    float confidence = 10.f;
    Sharding shard;   // --shard-ranks G --shard-rank r --shard-id FILE: this process is rank r of G over one spatially sharded map (one process per GPU)
};

int usage(const char* argv0)
{
    std::fprintf(stderr,
                 "usage: %s LOG.klg|data.txt [--width W --height H --fx F --fy F --cx C --cy C] [--masks DIR] [--out PREFIX]\n"
                 "       [--max-frames N] [--max-surfels N] [--no-superpixels] [--labels FILE] [--flip-colors] [--flann-every N] [--device K] [--no-close-loops] [--detect-only] [--decode-threads N] [--no-lookahead] [--confidence C]\n"
                 "       [--gt-dir DIR (DIR/<frame, 6 digits>.png, 8-bit instance ground truth)] [--eval FILE (precision / recall rows, needs --gt-dir)]\n"
                 "       [--shard-ranks G --shard-rank r --shard-id FILE [--shard-nonce N] (one process per GPU over one spatially sharded map; every rank replays the same log; -1: a world of one)]\n",
                 argv0);
    return 2;
}
}   // namespace

int main(int argc, char** argv)
{
    Args a;
    for (int i = 1; i < argc; i++) {
        const std::string s = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
            return argv[++i];
        };
        if (s == "--width") a.width = std::atoi(val("--width"));
        else if (s == "--height") a.height = std::atoi(val("--height"));
        else if (s == "--fx") a.fx = (float)std::atof(val("--fx"));
        else if (s == "--fy") a.fy = (float)std::atof(val("--fy"));
        else if (s == "--cx") a.cx = (float)std::atof(val("--cx"));
        else if (s == "--cy") a.cy = (float)std::atof(val("--cy"));
        else if (s == "--masks") a.masks = val("--masks");
        else if (s == "--out") a.out = val("--out");
        else if (s == "--labels") a.labels = val("--labels");
        else if (s == "--gt-dir") a.gt_dir = val("--gt-dir");      // DIR/<frame, 6 digits>.png: 8-bit instance ground truth (hasInstanceGroundTruth, IF/main.cpp:73)
        else if (s == "--eval") a.eval_file = val("--eval");
        else if (s == "--max-frames") a.max_frames = std::atoi(val("--max-frames"));
        else if (s == "--max-surfels") a.max_surfels = std::atoi(val("--max-surfels"));
        else if (s == "--flann-every") flann_skip_frames = std::atoi(val("--flann-every"));
        else if (s == "--device") a.device = std::atoi(val("--device"));
        else if (s == "--no-superpixels") a.superpixels = false;
        else if (s == "--no-close-loops") a.close_loops = false;
        else if (s == "--decode-threads") a.decode_threads = std::atoi(val("--decode-threads"));   // 0: records are decoded when asked for, as in the reference
        else if (s == "--detect-only") a.deform = false;   // loop closures are found and counted, the map is never deformed
        else if (s == "--confidence") a.confidence = (float)std::atof(val("--confidence"));
        else if (s == "--flip-colors") a.flip = true;
        else if (s == "--no-lookahead") a.lookahead = false;   // frames are handed over one at a time, as the reference's loop does
        else if (s == "--shard-ranks") a.shard.ranks = std::atoi(val("--shard-ranks"));   // -1: a world of one on the sharded path
        else if (s == "--shard-rank") a.shard.rank = std::atoi(val("--shard-rank"));
        else if (s == "--shard-nonce") a.shard.nonce = std::strtoull(val("--shard-nonce"), nullptr, 10);   // the same number for every rank of one run: a stale id file is never accepted
        else if (s == "--shard-id") a.shard.idFile = val("--shard-id");                   // where rank 0 leaves the ncclUniqueId for the others
        else if (s == "--help" || s == "-h") { usage(argv[0]); return 0; }
        else if (!s.empty() && s[0] == '-') { std::fprintf(stderr, "unknown option %s\n", s.c_str()); return usage(argv[0]); }
        else a.log = s;
    }
    if (a.log.empty()) return usage(argv[0]);

    try {
        Resolution::getInstance(a.width, a.height);            // IF/main.cpp:46-47
        Intrinsics::getInstance(a.fx, a.fy, a.cx, a.cy);

        std::unique_ptr<LogReader> log_reader;                 // IF/main.cpp:60-75
        if (a.log.size() > 4 && a.log.substr(a.log.size() - 4) == ".txt") log_reader.reset(new PNGLogReader(a.log));
        else {
            RawLogReader* raw = new RawLogReader(a.log, a.flip);
            log_reader.reset(raw);
            raw->setReadAhead(2 * a.decode_threads, a.decode_threads);   // records are inflated / JPEG-decoded ahead of the frame loop
        }

        std::unique_ptr<InstanceFusion> instancefusion(new InstanceFusion(instanceNum, a.width, a.height, false, 0));
        instancefusion->setSuperpixelRefinement(a.superpixels);
        if (!a.masks.empty()) instancefusion->setMaskSource(std::make_shared<MaskReplay>(a.masks));

        std::unique_ptr<ElasticFusionInterface> map(new ElasticFusionInterface());
        if (a.shard.on()) a.close_loops = false;   // (not available on a sharded map)
        if (a.shard.ranks > 1 && a.shard.nonce == 0) a.shard.nonce = Sharding::nonceFromLauncher();   // (the launcher's job id tells this run's id file from a leftover; 0 when it exports none)
        if (!map->Init(instancefusion->getInstanceTable(), a.max_surfels, a.device, a.out, a.close_loops, a.confidence, a.shard)) {
            std::cout << "ElasticFusionInterface init failure" << std::endl;
            return 1;
        }
        instancefusion->bindMap(map);
        // frames only (no masks, so no whetherDoSegmentation between the frames; no loop closing): ProcessFrame may return with the pose while the frame's map passes finish
        // under the decoding / copying of the next frame (ifx_c_api.h: "host_entry_async")
        if (a.masks.empty() && !a.close_loops && !a.shard.on()) ifx_set_option(map->handle(), "host_entry_async", 1);
        // one stream into a sharded map: the frames exchange the id keys of the lattice whetherDoSegmentation samples, a segmentation call completes the image (ifx_c_api.h: "own_lazy_ids")
        if (a.shard.on()) ifx_set_option(map->handle(), "own_lazy_ids", 1);
        if (!a.deform) map->elasticFusion().setDeformOnLoopClosure(false);

        int frame_Fusion = 0, lastTimeFlann = -1;
        std::vector<int> instanceTableLoopClosure((size_t)instancefusion->getInstanceNum() * 5);
        const auto t0 = std::chrono::steady_clock::now();
        while (log_reader->hasMore() && (a.max_frames <= 0 || frame_Fusion < a.max_frames)) {   // IF/main.cpp:108-307
            log_reader->getNext();
            instancefusion->getLoopClosureInstanceTable(instanceTableLoopClosure.data());
            const unsigned char* instanceGT = NULL;
            std::vector<unsigned char> gtBuf;
            if (!a.gt_dir.empty()) {
                char name[32];
                std::snprintf(name, sizeof(name), "/%06d.png", frame_Fusion);
                std::ifstream probe(a.gt_dir + name, std::ios::binary);
                if (probe) {
                    const ifx_detail::PngImage g = ifx_detail::decode_png(ifx_detail::read_file(a.gt_dir + name), a.gt_dir + name);
                    if (g.w != a.width || g.h != a.height || g.channels != 1 || g.bits != 8) throw std::runtime_error(a.gt_dir + name + ": ground truth must be 8-bit grey of the frame size");
                    gtBuf = g.data;
                    instanceGT = gtBuf.data();
                }
            }
            {   // the reader's read-ahead already holds the next frame: announced before this one is processed (its transfer, filter, pyramids and tracker run beside / behind this frame)
                const unsigned char* rgbNext = nullptr;
                const unsigned short* depthNext = nullptr;
                if (a.lookahead && !a.shard.on() && log_reader->peekNext(rgbNext, depthNext)) ifx_hint_next_frame(map->handle(), rgbNext, depthNext);
            }
            if (!map->ProcessFrame(log_reader->rgb, log_reader->depth, log_reader->timestamp, instanceTableLoopClosure.data(), instanceGT)) {
                std::cout << "Elastic fusion lost!" << a.log << std::endl;
                return 1;
            }
            if (frame_Fusion >= cnn_start_frames && !a.masks.empty() && instancefusion->whetherDoSegmentation(map, frame_Fusion)) {
                bool flannFlag = false;
                if (frame_Fusion - lastTimeFlann > flann_skip_frames) {
                    lastTimeFlann = frame_Fusion;
                    flannFlag = true;
                }
                instancefusion->ProcessSegmentation(log_reader->rgb, log_reader->depth, map, frame_Fusion, flannFlag);
            }
            frame_Fusion++;
        }
        ifx_sync(map->handle());
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        const int n_geo = map->elasticFusion().savePly();
        const int n_ins = map->elasticFusion().savePlyInstance();
        if (!a.labels.empty()) {   // bestIDInEachSurfel of the live surfels, one int32 each (what evaluateAndSave consumes, IF/Core/InstanceFusion.h:79)
            const std::vector<int32_t> l = instancefusion->getSurfelLabels(map);
            std::ofstream f(a.labels, std::ios::binary);
            f.write((const char*)l.data(), (std::streamsize)l.size() * 4);
        }
        if (!a.eval_file.empty()) instancefusion->evaluateAndSave(map, a.log, a.eval_file);   // IF/main.cpp:340
        const Matrix4f P = map->getCurrPose();
        int32_t la[3] = {0, 0, 0};
        ifx_lookahead_stats(map->handle(), la, 0);
        std::printf("%d frames announced ahead (%d with their tracker run ahead), ", la[2], la[1]);
        std::printf("%d frames in %.2f s (%.1f frames/s incl. log decoding), %d segmentation calls, %d surfels, %d stable -> %s.ply / _Instance.ply (%d), "
                    "last position %.6f %.6f %.6f, %d local loop-closure candidates, %d fern keyframes, %d fern matches, %d local / %d global deformations\n",
                    frame_Fusion, dt, frame_Fusion / (dt > 0 ? dt : 1), instancefusion->segmentationCalls(), map->getMapSurfelCount(), n_geo, a.out.c_str(), n_ins,
                    P(0, 3), P(1, 3), P(2, 3), map->elasticFusion().getLoopClosureCandidates(),
                    map->elasticFusion().ferns() ? (int)map->elasticFusion().ferns()->frames.size() : 0, map->elasticFusion().getFernMatches(),
                    map->elasticFusion().getDeforms(), map->elasticFusion().getFernDeforms());
        map.reset();   // ~ElasticFusion writes PREFIX.freiburg
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ifx_replay: %s\n", e.what());
        return 1;
    }
    return 0;
}
