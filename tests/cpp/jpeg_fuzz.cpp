// jpeg_fuzz.cpp -- malformed-input test of instancefusion_amd/host/ifx_jpeg.hpp, built with AddressSanitizer / UBSan on the CPU
// (tests/test_host_cpp.py::test_jpeg_decoder_survives_malformed_input).  Takes a valid baseline JPEG, applies crafted and seeded random
// corruptions, and requires of every variant: decode() returns an image or throws std::runtime_error -- no out-of-bounds access, no crash.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>
#include "ifx_jpeg.hpp"

static int tried = 0, threw = 0, decoded = 0;
static void attempt(const std::vector<uint8_t>& d)
{
    std::vector<uint8_t> rgb;
    int w = 0, h = 0;
    tried++;
    try {
        ifx_jpeg::decode(d.data(), d.size(), rgb, w, h);
        if ((size_t)w * h * 3 != rgb.size()) { std::fprintf(stderr, "inconsistent output size\n"); std::exit(2); }
        decoded++;
    } catch (const std::runtime_error&) { threw++; }
}
static size_t find_marker(const std::vector<uint8_t>& d, int m)
{
    for (size_t p = 2; p + 3 < d.size(); p++)
        if (d[p] == 0xFF && d[p + 1] == m) return p;
    return 0;
}
int main(int argc, char** argv)
{
    if (argc < 3) return 1;
    std::ifstream f(argv[1], std::ios::binary);
    const std::vector<uint8_t> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int rounds = std::atoi(argv[2]);
    attempt(good);
    if (decoded != 1) { std::fprintf(stderr, "the valid file did not decode\n"); return 2; }
    // ---- crafted: the cases of the round-1 review
    const size_t dht = find_marker(good, 0xC4), sof = find_marker(good, 0xC0), sos = find_marker(good, 0xDA), dqt = find_marker(good, 0xDB);
    if (!dht || !sof || !sos || !dqt) { std::fprintf(stderr, "markers not found\n"); return 2; }
    { auto d = good; d[dht + 5] = 3; attempt(d); }                                  // three codes of length 1: not a prefix code
    { auto d = good; for (int k = 0; k < 16; k++) d[dht + 5 + k] = 255; attempt(d); }   // code counts far beyond the segment
    { auto d = good; d[sof + 4 + 6 + 2] = 200; attempt(d); }                       // quantisation table selector 200
    { auto d = good; d[sof + 4 + 6 + 1] = 0x00; attempt(d); }                      // sampling factors 0 x 0
    { auto d = good; d[sof + 4 + 6 + 1] = 0x44; attempt(d); }                      // sampling factors 4 x 4
    { auto d = good; d[sos + 4 + 2] = 0xFF; attempt(d); }                          // entropy table selectors 15 / 15
    { auto d = good; d[sof + 2] = 0; d[sof + 3] = 8; attempt(d); }                 // frame header shorter than its fields
    { auto d = good; d[dqt + 2] = 0; d[dqt + 3] = 10; attempt(d); }                // quantisation table cut short
    { auto d = good; d[sof + 4 + 1] = 0xFF; d[sof + 4 + 2] = 0xFF; d[sof + 4 + 3] = 0xFF; d[sof + 4 + 4] = 0xFF; attempt(d); }   // 65535 x 65535
    { auto d = good; d[sof + 4 + 1] = 0; d[sof + 4 + 2] = 0; attempt(d); }         // height 0
    for (size_t cut : {size_t(3), size_t(20), dht + 3, sof + 6, sos + 5, good.size() / 2, good.size() - 1}) { std::vector<uint8_t> d(good.begin(), good.begin() + cut); attempt(d); }
    // ---- seeded random corruption: byte flips in the headers, in the entropy-coded data, and both
    std::mt19937 rng(12345);
    for (int r = 0; r < rounds; r++) {
        auto d = good;
        const int n = 1 + (int)(rng() % 8);
        for (int k = 0; k < n; k++) {
            const size_t lim = (r % 3 == 0) ? sos + 16 : d.size();
            d[rng() % lim] = (uint8_t)rng();
        }
        if (r % 7 == 0) d.resize(rng() % d.size());
        attempt(d);
    }
    std::printf("%d variants: %d decoded, %d refused\n", tried, decoded, threw);
    return 0;
}
