// Test helper (CPU only): drives the readers of instancefusion_amd/host/ifx_host.hpp and dumps what they deliver as raw bytes,
// so that tests/test_host_cpp.py can compare them with what the Python readers (instancefusion_amd/logio.py) deliver.
//   host_readers_check klg|png LOG W H OUT     -> per frame: int64 timestamp, depth u16[H*W], rgb u8[H*W*3]
//   host_readers_check npz DIR FRAME W H OUT   -> int32 n, int32 class_ids[n], masks u8[n*H*W]   (n = -1: no file)
//   host_readers_check quat OUT                -> quaternions of three fixed rotations (float x y z w)
#include <cstdio>
#include <cstdlib>

#include "ifx_host.hpp"
#include "ifx_deformation.hpp"

int main(int argc, char** argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "klg" || mode == "png") {
            Resolution::getInstance(std::atoi(argv[3]), std::atoi(argv[4]));
            std::unique_ptr<LogReader> r;
            if (mode == "klg") {
                RawLogReader* raw = new RawLogReader(argv[2], false);
                r.reset(raw);
                if (argc > 6) raw->setReadAhead(std::atoi(argv[6]), std::atoi(argv[7]));   // klg FILE W H OUT [AHEAD THREADS]
            }
            else r.reset(new PNGLogReader(argv[2]));
            std::ofstream f(argv[5], std::ios::binary);
            const size_t P = (size_t)Resolution::getInstance().numPixels();
            std::printf("%d\n", r->getNumFrames());
            int delivered = 0;
            const int back_at = argc > 8 ? std::atoi(argv[8]) : -1;   // ... [AHEAD THREADS BACK_AT]: getBack() after that many frames, then carry on
            while (r->hasMore()) {
                if (delivered == back_at) r->getBack();
                else r->getNext();
                delivered++;
                f.write((const char*)&r->timestamp, 8);
                f.write((const char*)r->depth, (std::streamsize)P * 2);
                f.write((const char*)r->rgb, (std::streamsize)P * 3);
            }
            return 0;
        }
        if (mode == "klgpeek") {   // klgpeek FILE W H OUT AHEAD THREADS: peekNext() before every getNext(); the dump holds what was PEEKED, "peeked N same M" on stdout
            Resolution::getInstance(std::atoi(argv[3]), std::atoi(argv[4]));
            RawLogReader r(argv[2], false);
            r.setReadAhead(std::atoi(argv[6]), std::atoi(argv[7]));
            std::ofstream f(argv[5], std::ios::binary);
            const size_t P = (size_t)Resolution::getInstance().numPixels();
            int peeked = 0, same = 0, frames = 0;
            std::vector<unsigned char> rgbCopy(P * 3);
            std::vector<unsigned short> depthCopy(P);
            while (r.hasMore()) {
                const unsigned char* pr = nullptr;
                const unsigned short* pd = nullptr;
                const bool ok = r.peekNext(pr, pd);
                if (ok) { std::memcpy(rgbCopy.data(), pr, P * 3); std::memcpy(depthCopy.data(), pd, P * 2); peeked++; }   // (what a caller would hand to ifx_hint_next_frame)
                r.getNext();
                frames++;
                if (ok && pr == r.rgb && pd == r.depth && std::memcmp(rgbCopy.data(), r.rgb, P * 3) == 0 && std::memcmp(depthCopy.data(), r.depth, P * 2) == 0) same++;
                f.write((const char*)&r.timestamp, 8);
                f.write((const char*)(ok ? depthCopy.data() : r.depth), (std::streamsize)P * 2);
                f.write((const char*)(ok ? rgbCopy.data() : r.rgb), (std::streamsize)P * 3);
            }
            std::printf("peeked %d same %d frames %d\n", peeked, same, frames);
            return 0;
        }
        if (mode == "npz") {
            MaskReplay rp(argv[2]);
            MaskResult res;
            std::ofstream f(argv[6], std::ios::binary);
            int32_t n = -1;
            if (rp.detect(std::atoi(argv[3]), nullptr, std::atoi(argv[4]), std::atoi(argv[5]), &res)) n = res.n;
            f.write((const char*)&n, 4);
            if (n > 0) {
                f.write((const char*)res.class_ids.data(), (std::streamsize)n * 4);
                f.write((const char*)res.masks.data(), (std::streamsize)res.masks.size());
            }
            return 0;
        }
        if (mode == "jpg") {   // jpg FILE OUT -> int32 w, h, rgb
            const std::vector<unsigned char> f = ifx_detail::read_file(argv[2]);
            std::vector<uint8_t> rgb;
            int w = 0, h = 0;
            ifx_jpeg::decode(f.data(), f.size(), rgb, w, h);
            std::ofstream o(argv[3], std::ios::binary);
            const int32_t hd[2] = {w, h};
            o.write((const char*)hd, 8);
            o.write((const char*)rgb.data(), (std::streamsize)rgb.size());
            return 0;
        }
        if (mode == "ferns") {   // ferns IN OUT: the host maths of the fern data base (ifx_ferns.hpp) on given read-back images, with a stand-in tracker
            // IN: int32 fullW, fullH, nFerns, maxDepth, seed, minGap, nOps; f32 fx, fy, cx, cy, photoThresh; per op: int32 kind (0 add, 1 find), time; f32 threshold;
            //     pose16; img np*3; inst np*3; verts np*4; norms np*4; (find) tracker answer: f32 dpose3 (added to the keyframe's translation), diag8
            std::ifstream in(argv[2], std::ios::binary);
            std::ofstream f(argv[3], std::ios::binary);
            int32_t hd[7];
            float fl[5];
            in.read((char*)hd, sizeof(hd));
            in.read((char*)fl, sizeof(fl));
            Ferns ferns(hd[2], hd[3], fl[4], hd[0], hd[1], fl[0], fl[1], fl[2], fl[3], 0, (uint32_t)hd[4]);
            ferns.minTimeGap = hd[5];
            for (auto& fe : ferns.conservatory) {
                const int32_t row[6] = {fe.pos[0], fe.pos[1], fe.rgbd[0], fe.rgbd[1], fe.rgbd[2], fe.rgbd[3]};
                f.write((const char*)row, sizeof(row));
            }
            const size_t np = (size_t)ferns.width * ferns.height;
            std::vector<unsigned char> img(np * 3), inst(np * 3);
            std::vector<float> verts(np * 4), norms(np * 4);
            for (int op = 0; op < hd[6]; op++) {
                int32_t kt[2];
                float thr;
                Matrix4f pose;
                in.read((char*)kt, 8); in.read((char*)&thr, 4); in.read((char*)pose.data(), 64);
                in.read((char*)img.data(), (std::streamsize)np * 3); in.read((char*)inst.data(), (std::streamsize)np * 3);
                in.read((char*)verts.data(), (std::streamsize)np * 16); in.read((char*)norms.data(), (std::streamsize)np * 16);
                if (kt[0] == 0) {
                    const int32_t ok = ferns.addFrameMaps(img.data(), verts.data(), norms.data(), inst.data(), pose, kt[1], thr) ? 1 : 0;
                    const int32_t nf = (int32_t)ferns.frames.size();
                    f.write((const char*)&ok, 4); f.write((const char*)&nf, 4);
                } else {
                    float ans[11];
                    in.read((char*)ans, sizeof(ans));
                    ferns.setTracker([&](const float*, const float*, const float*, const float*, float* p16, float* d8) {
                        p16[3] += ans[0]; p16[7] += ans[1]; p16[11] += ans[2];
                        std::memcpy(d8, ans + 3, 32);
                    });
                    std::vector<Ferns::SurfaceConstraint> cons;
                    const Matrix4f est = ferns.findFrameMaps(cons, pose, img.data(), verts.data(), norms.data(), kt[1], false);
                    const int32_t r[3] = {ferns.lastCandidate, ferns.lastClosest, (int32_t)cons.size()};
                    const float d[2] = {ferns.lastDissimilarity, ferns.lastPhotoError};
                    f.write((const char*)r, 12); f.write((const char*)d, 8); f.write((const char*)est.data(), 64);
                    for (auto& c : cons) { f.write((const char*)c.sourcePoint, 16); f.write((const char*)c.targetPoint, 16); }
                }
            }
            return 0;
        }
        if (mode == "fernrun") {   // fernrun KLG W H FX FY CX CY GAP CONFIDENCE rigid|builtin TIMEDELTA (GPU): the data base inside real frames, with a rigid stand-in for the graph optimiser
            Resolution::getInstance(std::atoi(argv[3]), std::atoi(argv[4]));
            Intrinsics::getInstance((float)std::atof(argv[5]), (float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8]));
            RawLogReader log(argv[2], false);
            ElasticFusion ef(std::atoi(argv[12]), 35000, 5e-05, 1e-05, true, false, false, 115, (float)std::atof(argv[10]), 12, 10, false, 0.3095, true, false, "", 2000000, 0);
            ef.resetFerns(12345u);
            ef.ferns()->minTimeGap = std::atoi(argv[9]);
            float lastShift = 0;
            if (std::string(argv[11]) == "rigid") ef.setFernHandler([&](ElasticFusion& e, const std::vector<Ferns::SurfaceConstraint>& cons, const Matrix4f& recovery, int srcTime) {
                (void)recovery; (void)srcTime;
                std::vector<float> nodes = e.sampleGraphModel(), g;
                float mean[3] = {0, 0, 0};
                for (auto& c : cons)
                    for (int k = 0; k < 3; k++) mean[k] += (c.targetPoint[k] - c.sourcePoint[k]) / (float)cons.size();
                lastShift = std::sqrt(mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
                if (nodes.size() / 4 < 4 || cons.empty()) return g;
                for (size_t i = 0; i < nodes.size() / 4; i++) {
                    const float n[16] = {nodes[i * 4], nodes[i * 4 + 1], nodes[i * 4 + 2], 1, 0, 0, 0, 1, 0, 0, 0, 1, mean[0], mean[1], mean[2], nodes[i * 4 + 3]};
                    g.insert(g.end(), n, n + 16);
                }
                return g;
            });
            int table[96 * 5] = {0};
            while (log.hasMore()) {
                log.getNext();
                ef.processFrame(log.rgb, log.depth, log.timestamp, table);
                const Ferns& f = *ef.ferns();
                const Matrix4f& P = ef.getCurrPose();
                std::printf("tick %d keyframes %d candidate %d closest %d icpErr %g icpCount %g photo %g matches %d deforms %d shift %g local %d lccand %d consErr %g pos %.6f %.6f %.6f surfels %d\n",
                            ef.getTick() - 1, (int)f.frames.size(), f.lastCandidate, f.lastClosest, f.lastICPError, f.lastICPCount, f.lastPhotoError, ef.getFernMatches(),
                            ef.getFernDeforms(), lastShift, ef.getDeforms(), ef.getLoopClosureCandidates(), ef.getLocalDeformation().lastMeanConsError, P(0, 3), P(1, 3), P(2, 3),
                            ef.getMapSurfelCount());
            }
            return 0;
        }
        if (mode == "deform") {   // deform IN OUT: Deformation::constrain on given nodes, constraints and poses (no GPU)
            std::ifstream in(argv[2], std::ios::binary);
            std::ofstream f(argv[3], std::ios::binary);
            int32_t hd[7];   // nodes, constraints, poses, fernMatch, relaxGraph, lastDeformTime, time
            in.read((char*)hd, sizeof(hd));
            std::vector<float> xyzt((size_t)hd[0] * 4);
            in.read((char*)xyzt.data(), (std::streamsize)xyzt.size() * 4);
            Deformation d;
            d.sampleGraphModel(xyzt);
            d.lastDeformTime = (uint64_t)hd[5];
            for (int i = 0; i < hd[1]; i++) {
                float st[6];
                int32_t q[4];
                in.read((char*)st, 24); in.read((char*)q, 16);
                if (q[2]) {
                    Deformation::Constraint c;
                    std::memcpy(c.src, st, 12); std::memcpy(c.target, st + 3, 12);
                    c.srcTime = (uint64_t)q[0]; c.targetTime = (uint64_t)q[1]; c.relative = true; c.pin = false; c.srcPointPoolId = c.tarPointPoolId = -1;
                    d.addConstraint(c);
                } else d.addConstraint(st, st + 3, (uint64_t)q[0], (uint64_t)q[1], q[3] != 0);
            }
            std::vector<Matrix4f> poses((size_t)hd[2]);
            std::vector<Deformation::TimedPose> tp, none;
            for (int i = 0; i < hd[2]; i++) {
                int32_t t;
                in.read((char*)&t, 4); in.read((char*)poses[i].data(), 64);
                tp.push_back({(uint64_t)t, poses[i].data()});
            }
            std::vector<float> rawGraph;
            std::vector<Deformation::Constraint> rel;
            const int32_t ok = d.constrain(tp, rawGraph, hd[6], hd[3] != 0, none, hd[4] != 0, &rel) ? 1 : 0;
            const float e[2] = {d.lastError, d.lastMeanConsError};
            const int32_t ng = (int32_t)rawGraph.size(), nr = (int32_t)rel.size();
            f.write((const char*)&ok, 4); f.write((const char*)e, 8); f.write((const char*)&ng, 4);
            f.write((const char*)rawGraph.data(), (std::streamsize)ng * 4);
            for (auto& P : poses) f.write((const char*)P.data(), 64);
            f.write((const char*)&nr, 4);
            for (auto& c : rel) { f.write((const char*)c.src, 12); f.write((const char*)c.target, 12); }
            const int32_t ldt = (int32_t)d.lastDeformTime;
            f.write((const char*)&ldt, 4);
            return 0;
        }
        if (mode == "shardid") {   // shardid FILE NONCE TIMEOUT: the reading side of Sharding::uniqueId (rank 1 of 2; no GPU) -- prints "id <first byte>" or "refused"
            Sharding sh; sh.ranks = 2; sh.rank = 1; sh.idFile = argv[2]; sh.nonce = std::strtoull(argv[3], nullptr, 10);
            try { const std::vector<uint8_t> id = sh.uniqueId(std::atoi(argv[4])); std::printf("id %d\n", (int)id[0]); }
            catch (const std::exception& e) { std::printf("refused: %s\n", e.what()); }
            return 0;
        }
        if (mode == "quat") {
            std::ofstream f(argv[2], std::ios::binary);
            const float R[3][9] = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, -1, 0, 1, 0, 0, 0, 0, 1}, {-1, 0, 0, 0, -0.6f, 0.8f, 0, 0.8f, 0.6f}};
            for (int k = 0; k < 3; k++) {
                Matrix4f P = Matrix4f::Identity();
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) P(i, j) = R[k][i * 3 + j];
                float q[4];
                ElasticFusion::quaternion(P, q);
                f.write((const char*)q, 16);
            }
            return 0;
        }
        if (mode == "api") {   // compile-time check of the loop-closure hooks of the class surface (no GPU: construction must fail loudly)
            Resolution::getInstance(64, 48);
            Intrinsics::getInstance(50.f, 50.f, 32.f, 24.f);
            try {
                ElasticFusion ef(200, 35000, 5e-05f, 1e-05f, true);
                ef.setLoopClosureHandler([](ElasticFusion& e, const ElasticFusion::LoopClosureCandidate&) {
                    ElasticFusion::Constraints c = e.loopClosureConstraints();
                    std::vector<float> nodes = e.sampleGraphModel(), rawGraph;
                    (void)c;
                    e.setDeformation(rawGraph, false);
                    e.adoptEstimatedPose();
                });
                ef.setFernHandler([](ElasticFusion& e, const std::vector<Ferns::SurfaceConstraint>& cons, const Matrix4f& recoveryPose, int fernSrcTime) {
                    for (const auto& c : cons) e.getGlobalDeformation().addConstraint(c.sourcePoint, c.targetPoint, (uint64_t)e.getTick(), (uint64_t)fernSrcTime, true);
                    (void)recoveryPose;
                    return std::vector<float>();
                });
                ef.resetFerns(1u);
                ef.setDeformOnLoopClosure(false);
                (void)ef.getFernMatches(); (void)ef.getFernDeforms(); (void)ef.getLocalDeformation().lastDeformTime; (void)ef.ferns()->frames.size();
                std::printf("created\n");
            } catch (const std::exception& e) {
                std::printf("refused: %s\n", e.what());
            }
            return 0;
        }
        std::fprintf(stderr, "bad mode\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
