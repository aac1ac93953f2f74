// oracle/ref_knn.cpp -- TEST INFRASTRUCTURE ONLY.  Thin C driver around the REFERENCE's own vendored FLANN 1.8.4
// (#included from /root/reference/deps/flann-1.8.4/src/cpp at build time, never copied into this repository).
// The reference searches with flann::KDTreeCuda3dIndex (IF/Core/InstanceFusion.cpp:1081-1125; CUDA, not buildable here); the same library's
// exact CPU index (KDTreeSingleIndex, checks = unlimited) returns the same k nearest neighbours up to the order of equidistant points.
// Built by `make -C oracle ref` into oracle/_ref/libref_knn.so; tests/test_oracle_cpu.py checks orc_knn_vote's neighbour lists against it and
// tools/make_golden.py stores its answer for a seeded point set in tests/golden/knn_ref.npz for the GPU box.
#include <flann/flann.hpp>

#include <cstdint>

extern "C" int ref_knn(const float* xyz, int n, int k, int leaf_max_size, int32_t* idx_out, float* dist_out)
{
    flann::Matrix<float> data(const_cast<float*>(xyz), n, 3);
    flann::Index<flann::L2<float> > index(data, flann::KDTreeSingleIndexParams(leaf_max_size));   // leaf_max_size 64 as at the reference's call site
    index.buildIndex();
    flann::Matrix<int> I(idx_out, n, k);
    flann::Matrix<float> D(dist_out, n, k);
    flann::SearchParams sp(flann::FLANN_CHECKS_UNLIMITED);
    sp.sorted = true;
    index.knnSearch(data, I, D, k, sp);
    return 0;
}
