// Test helper (CPU only): drives the readers of instancefusion_amd/host/ifx_host.hpp and dumps what they deliver as raw bytes,
// so that tests/test_host_cpp.py can compare them with what the Python readers (instancefusion_amd/logio.py) deliver.
//   host_readers_check klg|png LOG W H OUT     -> per frame: int64 timestamp, depth u16[H*W], rgb u8[H*W*3]
//   host_readers_check npz DIR FRAME W H OUT   -> int32 n, int32 class_ids[n], masks u8[n*H*W]   (n = -1: no file)
//   host_readers_check quat OUT                -> quaternions of three fixed rotations (float x y z w)
#include <cstdio>
#include <cstdlib>

#include "ifx_host.hpp"

int main(int argc, char** argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "klg" || mode == "png") {
            Resolution::getInstance(std::atoi(argv[3]), std::atoi(argv[4]));
            std::unique_ptr<LogReader> r;
            if (mode == "klg") r.reset(new RawLogReader(argv[2], false));
            else r.reset(new PNGLogReader(argv[2]));
            std::ofstream f(argv[5], std::ios::binary);
            const size_t P = (size_t)Resolution::getInstance().numPixels();
            std::printf("%d\n", r->getNumFrames());
            while (r->hasMore()) {
                r->getNext();
                f.write((const char*)&r->timestamp, 8);
                f.write((const char*)r->depth, (std::streamsize)P * 2);
                f.write((const char*)r->rgb, (std::streamsize)P * 3);
            }
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
