// oracle/ref_slic.cpp -- TEST INFRASTRUCTURE ONLY.  Thin C driver around the REFERENCE's own gSLICr
// per-pixel functions (#included from /root/reference/src/gSLICr at build time, never copied into this
// repository): rgb2xyz / cvt_img_space_shared, init_cluster_centers_shared, find_center_association_shared,
// finalize_reduction_result_shared and supress_local_lable of gSLICr_Lib/engines/gSLICr_seg_engine_shared.h.
// Built by `make -C oracle ref` into oracle/_ref/libref_slic.so; tests/test_oracle_slic.py checks
// oracle/orc_slic.c::orc_slic_segment against it label by label.
//
// What is NOT reference code here: the loops that call those functions (the reference launches them
// from gSLICr_seg_engine_GPU.cu, which needs CUDA) and the 16x16-block tree reduction of
// Update_Cluster_Center_device (:203-290), restated below in the same summation order.
#include <cstdint>
#include <cstring>
#include <vector>

#include "gSLICr_Lib/engines/gSLICr_seg_engine_shared.h"

using gSLICr::Vector2i;
using gSLICr::Vector4f;
using gSLICr::Vector4u;
using gSLICr::objects::spixel_info;

extern "C" int ref_slic_segment(const uint8_t* rgb, int w, int h, int spixel_size, float coh_weight, int iters, int32_t* seg)
{
    const int P = w * h, BD = 16;
    Vector2i img(w, h), map(w / spixel_size, h / spixel_size);
    if (map.x < 1 || map.y < 1) return -1;
    std::vector<Vector4u> in(P);
    std::vector<Vector4f> cvt(P);
    for (int i = 0; i < P; i++) {  // gSLICrInterface + imageCV2SLIC: x <- channel 0, y <- channel 1, z <- channel 2
        in[i].x = rgb[i * 3]; in[i].y = rgb[i * 3 + 1]; in[i].z = rgb[i * 3 + 2]; in[i].w = 0;
        cvt[i] = Vector4f(0, 0, 0, 0);
    }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) cvt_img_space_shared(in.data(), cvt.data(), img, x, y, gSLICr::XYZ);
    std::vector<spixel_info> centers(map.x * map.y);
    for (int y = 0; y < map.y; y++) for (int x = 0; x < map.x; x++) init_cluster_centers_shared(cvt.data(), centers.data(), map, img, spixel_size, x, y);
    float max_xy = 1.0f / (1.4242f * spixel_size), max_col = 5.0f / 1.7321f;
    max_col *= max_col; max_xy *= max_xy;
    std::vector<int> idx(P, 0), tmp(P, 0);
    auto assoc = [&]() {
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
            find_center_association_shared(cvt.data(), centers.data(), idx.data(), map, img, spixel_size, coh_weight, x, y, max_xy, max_col);
    };
    const int per_line = spixel_size * 3 / BD, per_center = (spixel_size * spixel_size * 9 + BD * BD - 1) / (BD * BD);
    std::vector<spixel_info> accum((size_t)map.x * map.y * per_center);
    auto update = [&]() {
        Vector4f col[256];
        gSLICr::Vector2f xy[256];
        int cnt[256];
        for (int gy = 0; gy < map.y; gy++) for (int gx = 0; gx < map.x; gx++) {
            int id = gy * map.x + gx;
            for (int z = 0; z < per_center; z++) {
                int bx = z % per_line, by = z / per_line;
                for (int t = 0; t < 256; t++) {
                    col[t] = Vector4f(0, 0, 0, 0); xy[t] = gSLICr::Vector2f(0, 0); cnt[t] = 0;
                    int xo = bx * BD + (t % BD), yo = by * BD + (t / BD);
                    if (xo < spixel_size * 3 && yo < spixel_size * 3) {
                        int xi = gx * spixel_size - spixel_size + xo, yi = gy * spixel_size - spixel_size + yo;
                        if (xi >= 0 && xi < w && yi >= 0 && yi < h && idx[yi * w + xi] == id) {
                            col[t] = cvt[yi * w + xi]; xy[t] = gSLICr::Vector2f((float)xi, (float)yi); cnt[t] = 1;
                        }
                    }
                }
                for (int st = 128; st >= 1; st >>= 1)
                    for (int t = 0; t < st; t++) { col[t] += col[t + st]; xy[t] += xy[t + st]; cnt[t] += cnt[t + st]; }
                spixel_info& a = accum[(size_t)id * per_center + z];
                a.center = xy[0]; a.color_info = col[0]; a.no_pixels = cnt[0];
            }
        }
        for (int y = 0; y < map.y; y++) for (int x = 0; x < map.x; x++) finalize_reduction_result_shared(accum.data(), centers.data(), map, per_center, x, y);
    };
    assoc();
    for (int i = 0; i < iters; i++) { update(); assoc(); }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) supress_local_lable(idx.data(), tmp.data(), img, x, y);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) supress_local_lable(tmp.data(), idx.data(), img, x, y);
    std::memcpy(seg, idx.data(), (size_t)P * sizeof(int));
    return map.x * map.y;
}

// the colour conversion alone, for a per-pixel check
extern "C" void ref_rgb2xyz(const uint8_t* rgb, int n, float* out)
{
    for (int i = 0; i < n; i++) {
        Vector4u p; p.x = rgb[i * 3]; p.y = rgb[i * 3 + 1]; p.z = rgb[i * 3 + 2]; p.w = 0;
        Vector4f o(0, 0, 0, 0);
        rgb2xyz(p, o);
        out[i * 3] = o.x; out[i * 3 + 1] = o.y; out[i * 3 + 2] = o.z;
    }
}
