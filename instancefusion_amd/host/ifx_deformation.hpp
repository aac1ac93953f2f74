// instancefusion_amd/host/ifx_deformation.hpp -- the deformation-graph optimiser of the reference's loop closures as host code above the C-ABI:
// Deformation::addConstraint / constrain (EF/Deformation.cpp:68-220) over DeformationGraph (EF/Utils/DeformationGraph.cpp): an embedded deformation
// graph (one affine 3 x 3 + translation per node, nodes = every 5000th surfel in time order) fitted by Gauss-Newton to
//     E = wRot * sum_j ||R_j^T R_j - I||  +  wReg * sum_j sum_{n in N(j)} ||R_j (g_n - g_j) + g_j + t_j - (g_n + t_n)||^2  +  wCon * sum_c ||phi(v_c) - q_c||^2
// with wRot = 1, wReg = 10, wCon = 100, sequential connectivity (k = 4) and time-window vertex weights.  The reference solves the normal equations with
// CHOLMOD; here they are assembled into a skyline (envelope) matrix in natural node order -- the graph is a chain in time, so the envelope is a narrow band
// plus the few rows that relative constraints couple to an earlier part of the chain -- and factorised by a plain f64 skyline Cholesky.
// Host code only (no GPU): what it consumes comes from ifx_sample_graph_model / ifx_loop_closure_constraints / Ferns::findFrame, what it produces goes to
// ifx_set_deformation.  Checked against an independent dense numpy restatement (oracle/orc_deformation.py, tests/test_host_cpp.py).
#ifndef IFX_DEFORMATION_HPP_
#define IFX_DEFORMATION_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

class DeformationGraph {
public:
    static const int k = 4;               // EF/Deformation.cpp:23: def(4, &pointPool)
    static const int numVariables = 12;   // rotation (column-major 3 x 3) + translation
    struct Node {
        float position[3];
        float rotation[9];   // column-major, as Eigen stores it (the variable order of the reference's Jacobian)
        float translation[3];
        uint64_t time;
        bool enabled;
        std::vector<int> neighbours;
    };
    struct Weight {
        double weight;
        int node;
    };
    typedef std::array<Weight, k> WeightMap;
    struct Constraint {
        int vertexId;
        bool relative;
        int targetId;
        float targetPosition[3];
    };

    bool isInit() const { return initialised; }
    std::vector<Node>& getGraph() { return nodes; }

    // DeformationGraph::initialiseGraph + connectGraphSeq (:57-95, :256-292)
    void initialiseGraph(const std::vector<std::array<float, 3>>& points, const std::vector<uint64_t>& times)
    {
        nodes.assign(points.size(), Node());
        for (size_t i = 0; i < points.size(); i++) {
            Node& n = nodes[i];
            std::memcpy(n.position, points[i].data(), 12);
            resetNode(n);
            n.time = times[i];
            n.enabled = true;
        }
        const int N = (int)nodes.size();
        for (int i = 0; i < N; i++) {
            if (i < k / 2) {
                for (int n = 0; n < k + 1; n++)
                    if (n != i) nodes[i].neighbours.push_back(n);
            } else if (i < N - k / 2) {
                for (int n = 0; n < k / 2; n++) {
                    nodes[i].neighbours.push_back(i - (n + 1));
                    nodes[i].neighbours.push_back(i + (n + 1));
                }
            } else {
                for (int n = N - (k + 1); n < N; n++)
                    if (n != i) nodes[i].neighbours.push_back(n);
            }
        }
        initialised = (int)nodes.size() > k;
    }

    // weightVerticesSeq / setPosesSeq (:137-254, :294-409): the node nearest in time, up to 20 nodes back from it (forward only when the chain starts
    // there), the k nearest of those in space with weights (1 - d / d_{k+1})^2, normalised, ordered by node
    WeightMap weigh(const float* p, uint64_t time) const
    {
        const int N = (int)nodes.size();
        int imin = 0, imax = N - 1, imid = (imin + imax) / 2;
        while (imax >= imin) {
            imid = (imin + imax) / 2;
            if (nodes[imid].time < time) imin = imid + 1;
            else if (nodes[imid].time > time) imax = imid - 1;
            else break;
        }
        imin = std::min(imin, N - 1);
        // (imax can reach -1 when the time lies before the first node: the reference reads sampledGraphTimes[-1] there; the first node is meant)
        const int imaxc = std::max(imax, 0);
        auto dist = [&](int i) { return std::llabs((long long)nodes[i].time - (long long)time); };
        int found;
        if (dist(imin) <= dist(imid) && dist(imin) <= dist(imaxc)) found = imin;
        else if (dist(imid) <= dist(imin) && dist(imid) <= dist(imaxc)) found = imid;
        else found = imaxc;
        std::vector<std::pair<float, int>> near;
        unsigned back = 0;
        for (int j = found; j >= 0; j--) {
            near.push_back(std::make_pair(norm3(nodes[j].position, p), j));
            if (++back == 20) break;
        }
        if (back != 20)
            for (int j = found + 1; j < N; j++) {
                near.push_back(std::make_pair(norm3(nodes[j].position, p), j));
                if (++back == 20) break;
            }
        std::stable_sort(near.begin(), near.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
        const double dMax = near[k].first;
        WeightMap m;
        double sum = 0;
        for (int j = 0; j < k; j++) {
            m[j].weight = std::pow(1.0f - norm3(nodes[near[j].second].position, p) / dMax, 2);
            m[j].node = near[j].second;
            sum += m[j].weight;
        }
        for (int j = 0; j < k; j++) m[j].weight /= sum;
        std::sort(m.begin(), m.end(), [](const Weight& a, const Weight& b) { return a.node < b.node; });
        return m;
    }

    void setVertices(std::vector<std::array<float, 3>>* pool, const std::vector<uint64_t>& times)   // appendVertices on an empty pool (:97-104)
    {
        source = pool;
        vertexMap.clear();
        for (size_t i = 0; i < pool->size(); i++) vertexMap.push_back(weigh((*pool)[i].data(), times[i]));
    }
    void clearConstraints() { constraints.clear(); }
    void addConstraint(int vertexId, const float* target)
    {
        Constraint c;
        c.vertexId = vertexId; c.relative = false; c.targetId = -1;
        std::memcpy(c.targetPosition, target, 12);
        constraints.push_back(c);
    }
    void addRelativeConstraint(int vertexId, int targetId)
    {
        Constraint c;
        c.vertexId = vertexId; c.relative = true; c.targetId = targetId;
        c.targetPosition[0] = c.targetPosition[1] = c.targetPosition[2] = 0;
        constraints.push_back(c);
    }

    // computeVertexPosition (:996-1013)
    void deform(const WeightMap& m, const float* src, float* out) const
    {
        float p[3] = {0, 0, 0};
        for (int i = 0; i < k; i++) {
            const Node& n = nodes[m[i].node];
            const float d[3] = {src[0] - n.position[0], src[1] - n.position[1], src[2] - n.position[2]};
            for (int r = 0; r < 3; r++) {
                const float v = (n.rotation[r] * d[0] + n.rotation[3 + r] * d[1] + n.rotation[6 + r] * d[2]) + n.position[r] + n.translation[r];
                p[r] += (float)m[i].weight * v;   // (Eigen converts the double weight to the vector's scalar type first)
            }
        }
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    }
    // nonRelativeConstraintError (:1015-1030)
    float nonRelativeConstraintError() const
    {
        float result = 0;
        for (const Constraint& c : constraints)
            if (!c.relative) {
                float p[3];
                deform(vertexMap[c.vertexId], (*source)[c.vertexId].data(), p);
                result += norm3(p, c.targetPosition);
            }
        return result / (float)constraints.size();
    }

    // optimiseGraphSparse (:461-539): at most three Gauss-Newton steps with the reference's stopping rules
    bool optimiseGraphSparse(float& error, float& meanConsErr, const bool fernMatch, const uint64_t lastDeformTime)
    {
        meanConsErr = nonRelativeConstraintError();
        if (fernMatch && meanConsErr < 0.06) return false;
        int numCols = 0;
        for (Node& n : nodes) {
            n.enabled = n.time > lastDeformTime;
            if (n.enabled) numCols += numVariables;
        }
        int col = 0;
        column.assign(nodes.size(), -1);
        for (size_t i = 0; i < nodes.size(); i++)
            if (nodes[i].enabled) { column[i] = col; col += numVariables; }
        std::vector<Row> rows;
        linearise(rows);
        error = (float)squaredNorm(rows);
        double lastError = error;
        int iter = 0;
        solveFailed = false;
        while (iter++ < 3) {
            std::vector<double> delta;
            if (!solve(rows, numCols, delta)) { solveFailed = true; break; }
            int z = 0;   // applyDeltaSparse (:964-994)
            double dn = 0;
            for (Node& n : nodes)
                if (n.enabled) {
                    for (int v = 0; v < 9; v++) n.rotation[v] += (float)delta[z + v];
                    for (int v = 0; v < 3; v++) n.translation[v] += (float)delta[z + 9 + v];
                    z += numVariables;
                }
            for (double d : delta) dn += d * d;
            linearise(rows);
            error = (float)squaredNorm(rows);
            const double errorDiff = error - lastError;
            if (error > lastError || std::sqrt(dn) < 1e-2 || error < 1e-3 || std::fabs(errorDiff) < 1e-5 * error || (iter == 1 && fernMatch && error > 10.0f)) break;
            lastError = error;
        }
        meanConsErr = nonRelativeConstraintError();
        return true;
    }
    bool solveFailed = false;   // the normal equations were not positive definite (the reference would hand CHOLMOD the same matrix)

    // applyGraphToPoses (:106-135); poses: row-major 4 x 4.  The rotation is re-orthonormalised as U V^T of its SVD: here by the polar iteration.
    void applyGraphToPoses(const std::vector<float*>& poses, const std::vector<uint64_t>& times) const
    {
        for (size_t i = 0; i < poses.size(); i++) {
            float* P = poses[i];
            const float t[3] = {P[3], P[7], P[11]};
            const WeightMap m = weigh(t, times[i]);
            float np[3];
            deform(m, t, np);
            float R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // row-major blend of the node rotations
            for (int j = 0; j < k; j++) {
                const Node& n = nodes[m[j].node];
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) R[r * 3 + c] += (float)m[j].weight * n.rotation[c * 3 + r];
            }
            double M[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) M[r * 3 + c] = (double)R[r * 3] * P[c] + (double)R[r * 3 + 1] * P[4 + c] + (double)R[r * 3 + 2] * P[8 + c];
            polar(M);
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) P[r * 4 + c] = (float)M[r * 3 + c];
                P[r * 4 + 3] = np[r];
            }
        }
    }
    void applyGraphToVertices()   // :411-420
    {
        for (size_t i = 0; i < source->size(); i++) {
            float p[3];
            deform(vertexMap[i], (*source)[i].data(), p);
            std::memcpy((*source)[i].data(), p, 12);
        }
    }

private:
    struct Row {
        double r;
        int n;
        int col[8 * 4];
        double val[8 * 4];
        void add(int c, double v)
        {
            for (int i = 0; i < n; i++)
                if (col[i] == c) { val[i] += v; return; }
            col[n] = c; val[n] = v; n++;
        }
    };
    static void resetNode(Node& n)
    {
        for (int i = 0; i < 9; i++) n.rotation[i] = (i % 4 == 0) ? 1.f : 0.f;
        n.translation[0] = n.translation[1] = n.translation[2] = 0.f;
    }
    static float norm3(const float* a, const float* b)
    {
        const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
        return std::sqrt(x * x + y * y + z * z);
    }
    static double squaredNorm(const std::vector<Row>& rows)
    {
        double s = 0;
        for (const Row& r : rows) s += r.r * r.r;
        return s;
    }
    bool influences(const WeightMap& m) const
    {
        for (int i = 0; i < k; i++)
            if (nodes[m[i].node].enabled) return true;
        return false;
    }
    // residual and Jacobian of every active term at the current state (sparseResidual :846-953, sparseJacobian :541-844)
    void linearise(std::vector<Row>& rows) const
    {
        rows.clear();
        const double sReg = std::sqrt(10.0), sCon = std::sqrt(100.0);
        Row row;
        // rotation: the six conditions of R^T R = I, weight 1
        for (size_t j = 0; j < nodes.size(); j++) {
            const Node& n = nodes[j];
            if (!n.enabled) continue;
            const float* R = n.rotation;
            const int c0 = column[j];
            const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
            for (int q = 0; q < 3; q++) {   // c_a . c_b
                const int a = pairs[q][0], b = pairs[q][1];
                row.n = 0;
                row.r = R[a * 3] * R[b * 3] + R[a * 3 + 1] * R[b * 3 + 1] + R[a * 3 + 2] * R[b * 3 + 2];
                for (int i = 0; i < 3; i++) { row.add(c0 + a * 3 + i, R[b * 3 + i]); row.add(c0 + b * 3 + i, R[a * 3 + i]); }
                rows.push_back(row);
            }
            for (int a = 0; a < 3; a++) {   // c_a . c_a - 1
                row.n = 0;
                row.r = (R[a * 3] * R[a * 3] + R[a * 3 + 1] * R[a * 3 + 1] + R[a * 3 + 2] * R[a * 3 + 2]) - 1.0;
                for (int i = 0; i < 3; i++) row.add(c0 + a * 3 + i, 2.0 * R[a * 3 + i]);
                rows.push_back(row);
            }
        }
        // regularisation: every (node, neighbour) pair with at least one enabled end
        for (size_t j = 0; j < nodes.size(); j++) {
            const Node& a = nodes[j];
            for (int nb : a.neighbours) {
                const Node& b = nodes[nb];
                if (!a.enabled && !b.enabled) continue;
                const float d[3] = {b.position[0] - a.position[0], b.position[1] - a.position[1], b.position[2] - a.position[2]};
                for (int r = 0; r < 3; r++) {
                    row.n = 0;
                    const float v = (a.rotation[r] * d[0] + a.rotation[3 + r] * d[1] + a.rotation[6 + r] * d[2]) + a.position[r] + a.translation[r] - (b.position[r] + b.translation[r]);
                    row.r = (double)v * sReg;
                    if (a.enabled) {
                        const int c0 = column[j];
                        row.add(c0 + r, d[0] * sReg); row.add(c0 + 3 + r, d[1] * sReg); row.add(c0 + 6 + r, d[2] * sReg); row.add(c0 + 9 + r, sReg);
                    }
                    if (b.enabled) row.add(column[nb] + 9 + r, -sReg);
                    rows.push_back(row);
                }
            }
        }
        // constraints
        for (const Constraint& c : constraints) {
            const WeightMap& m = vertexMap[c.vertexId];
            bool active = influences(m);
            if (c.relative && !active) active = influences(vertexMap[c.targetId]);
            if (!active) continue;
            const float* src = (*source)[c.vertexId].data();
            float p[3], q[3];
            deform(m, src, p);
            if (c.relative) deform(vertexMap[c.targetId], (*source)[c.targetId].data(), q);
            else std::memcpy(q, c.targetPosition, 12);
            for (int r = 0; r < 3; r++) {
                row.n = 0;
                row.r = (double)(p[r] - q[r]) * sCon;
                addVertexTerms(row, m, src, r, sCon);
                if (c.relative) addVertexTerms(row, vertexMap[c.targetId], (*source)[c.targetId].data(), r, -sCon);
                rows.push_back(row);
            }
        }
    }
    void addVertexTerms(Row& row, const WeightMap& m, const float* v, int r, double s) const
    {
        for (int i = 0; i < k; i++) {
            const Node& n = nodes[m[i].node];
            if (!n.enabled) continue;
            const int c0 = column[m[i].node];
            const double w = m[i].weight * s;
            row.add(c0 + r, w * (v[0] - n.position[0]));
            row.add(c0 + 3 + r, w * (v[1] - n.position[1]));
            row.add(c0 + 6 + r, w * (v[2] - n.position[2]));
            row.add(c0 + 9 + r, w);
        }
    }
    // (J^T J) delta = -J^T r in a skyline matrix, Cholesky in place
    bool solve(const std::vector<Row>& rows, int n, std::vector<double>& delta) const
    {
        if (n == 0) return false;
        std::vector<int> first(n);
        for (int i = 0; i < n; i++) first[i] = i;
        for (const Row& r : rows) {
            int lo = n;
            for (int i = 0; i < r.n; i++) lo = std::min(lo, r.col[i]);
            for (int i = 0; i < r.n; i++) first[r.col[i]] = std::min(first[r.col[i]], lo);
        }
        std::vector<size_t> start(n + 1, 0);
        for (int i = 0; i < n; i++) start[i + 1] = start[i] + (size_t)(i - first[i] + 1);
        std::vector<double> A(start[n], 0.0);
        auto at = [&](int i, int j) -> double& { return A[start[i] + (size_t)(j - first[i])]; };
        delta.assign(n, 0.0);
        for (const Row& r : rows)
            for (int a = 0; a < r.n; a++) {
                delta[r.col[a]] -= r.val[a] * r.r;
                for (int b = 0; b < r.n; b++)
                    if (r.col[b] <= r.col[a]) at(r.col[a], r.col[b]) += r.val[a] * r.val[b];
            }
        for (int i = 0; i < n; i++) {
            for (int j = first[i]; j <= i; j++) {
                double s = at(i, j);
                for (int q = std::max(first[i], first[j]); q < j; q++) s -= at(i, q) * at(j, q);
                if (j < i) at(i, j) = s / at(j, j);
                else {
                    if (!(s > 0)) return false;
                    at(i, i) = std::sqrt(s);
                }
            }
        }
        for (int i = 0; i < n; i++) {   // L y = b
            double s = delta[i];
            for (int q = first[i]; q < i; q++) s -= at(i, q) * delta[q];
            delta[i] = s / at(i, i);
        }
        for (int i = n - 1; i >= 0; i--) {   // L^T x = y
            delta[i] /= at(i, i);
            for (int q = first[i]; q < i; q++) delta[q] -= at(i, q) * delta[i];
        }
        return true;
    }
    // M <- the orthogonal factor of its polar decomposition (= U V^T of the SVD for a non-singular M): Newton iteration X <- (X + X^-T) / 2
    static void polar(double* M)
    {
        for (int it = 0; it < 30; it++) {
            const double c[9] = {M[4] * M[8] - M[5] * M[7], M[5] * M[6] - M[3] * M[8], M[3] * M[7] - M[4] * M[6], M[2] * M[7] - M[1] * M[8], M[0] * M[8] - M[2] * M[6],
                                 M[1] * M[6] - M[0] * M[7], M[1] * M[5] - M[2] * M[4], M[2] * M[3] - M[0] * M[5], M[0] * M[4] - M[1] * M[3]};   // cofactors = det * X^-T
            const double det = M[0] * c[0] + M[1] * c[1] + M[2] * c[2];
            if (std::fabs(det) < 1e-30) return;
            double diff = 0;
            for (int i = 0; i < 9; i++) {
                const double x = 0.5 * (M[i] + c[i] / det);
                diff = std::max(diff, std::fabs(x - M[i]));
                M[i] = x;
            }
            if (diff < 1e-15) break;
        }
    }

    bool initialised = false;
    std::vector<Node> nodes;
    std::vector<int> column;
    std::vector<WeightMap> vertexMap;
    std::vector<std::array<float, 3>>* source = nullptr;
    std::vector<Constraint> constraints;
};

// EF/Deformation.{h,cpp} without its GL half (the sampling pass is ifx_sample_graph_model)
class Deformation {
public:
    struct Constraint {   // EF/Deformation.h:57-88
        float src[3], target[3];
        uint64_t srcTime, targetTime;
        bool relative, pin;
        int srcPointPoolId, tarPointPoolId;
    };
    struct TimedPose {   // one entry of ferns.frames / poseGraph: updated in place when the graph is applied
        uint64_t time;
        float* pose16;   // row-major 4 x 4
    };

    // Deformation::sampleGraphModel (:260-337): xyzt = what ifx_sample_graph_model delivered; (re)initialises the graph when there are more than k nodes
    void sampleGraphModel(const std::vector<float>& xyzt)
    {
        vertices = xyzt;
        const size_t count = xyzt.size() / 4;
        if ((int)count > DeformationGraph::k) init(xyzt, 1);
    }
    // Deformation::sampleGraphFrom (:222-258): every 5th node of the local graph for the global one
    void sampleGraphFrom(const Deformation& other)
    {
        const int count = (int)(other.vertices.size() / 4);
        if (count / 5 > DeformationGraph::k) init(other.vertices, 5);
    }
    // :68-87
    void addConstraint(const Constraint& c) { constraints.push_back(c); }
    void addConstraint(const float* src, const float* target, uint64_t srcTime, uint64_t targetTime, bool pinConstraints)
    {
        constraints.push_back(make(src, target, srcTime, targetTime, false, false));
        if (pinConstraints) constraints.push_back(make(target, target, targetTime, targetTime, false, true));
    }
    // :89-220.  ferns / poseGraph: their poses are deformed with the map when the graph is accepted
    bool constrain(std::vector<TimedPose>& ferns, std::vector<float>& rawGraph, int time, const bool fernMatch, std::vector<TimedPose>& poseGraph, const bool relaxGraph,
                   std::vector<Constraint>* newRelativeCons = nullptr)
    {
        if (!def.isInit()) return false;
        std::vector<float*> rawPoses;
        std::vector<uint64_t> times;
        for (TimedPose& p : ferns) { times.push_back(p.time); rawPoses.push_back(p.pose16); }
        if (fernMatch)
            for (TimedPose& p : poseGraph) { times.push_back(p.time); rawPoses.push_back(p.pose16); }
        std::vector<std::array<float, 3>> pointPool;
        std::vector<uint64_t> vertexTimes;
        for (Constraint& c : constraints) {
            pointPool.push_back({c.src[0], c.src[1], c.src[2]});
            vertexTimes.push_back(c.srcTime);
            c.srcPointPoolId = (int)pointPool.size() - 1;
            if (c.relative) {
                pointPool.push_back({c.target[0], c.target[1], c.target[2]});
                vertexTimes.push_back(c.targetTime);
                c.tarPointPoolId = (int)pointPool.size() - 1;
            }
        }
        def.setVertices(&pointPool, vertexTimes);
        def.clearConstraints();
        for (const Constraint& c : constraints) {
            if (c.relative) def.addRelativeConstraint(c.srcPointPoolId, c.tarPointPoolId);
            else def.addConstraint(c.srcPointPoolId, c.target);
        }
        lastError = lastMeanConsError = 0;
        const bool optimised = !constraints.empty() && def.optimiseGraphSparse(lastError, lastMeanConsError, fernMatch, (fernMatch || relaxGraph) ? 0 : lastDeformTime);
        bool poseUpdated = false;
        if (!constraints.empty() && !def.solveFailed && (!fernMatch || (optimised && lastMeanConsError < 0.0003 && lastError < 0.12))) {
            def.applyGraphToPoses(rawPoses, times);
            def.applyGraphToVertices();
            if (!fernMatch && newRelativeCons) {
                newRelativeCons->clear();
                for (const Constraint& c : constraints)
                    if (!c.relative && !c.pin) newRelativeCons->push_back(make(pointPool[c.srcPointPoolId].data(), c.target, c.srcTime, c.targetTime, true, false));
            }
            std::vector<DeformationGraph::Node>& g = def.getGraph();
            rawGraph.resize(g.size() * 16);   // 16 floats per node: position, rotation (column-major), translation, time
            for (size_t i = 0; i < g.size(); i++) {
                std::memcpy(&rawGraph[i * 16], g[i].position, 12);
                std::memcpy(&rawGraph[i * 16 + 3], g[i].rotation, 36);
                std::memcpy(&rawGraph[i * 16 + 12], g[i].translation, 12);
                rawGraph[i * 16 + 15] = (float)g[i].time;
            }
            if (!fernMatch && !relaxGraph) lastDeformTime = (uint64_t)time;
            poseUpdated = true;
        }
        constraints.clear();
        return poseUpdated;
    }

    DeformationGraph def;
    uint64_t lastDeformTime = 0;
    float lastError = 0, lastMeanConsError = 0;   // of the last constrain()
    std::vector<float> vertices;                   // x, y, z, time of the sampled nodes

private:
    void init(const std::vector<float>& xyzt, int rate)
    {
        std::vector<std::array<float, 3>> pts;
        std::vector<uint64_t> times;
        for (size_t i = 0; i < xyzt.size() / 4; i += (size_t)rate) {
            pts.push_back({xyzt[i * 4], xyzt[i * 4 + 1], xyzt[i * 4 + 2]});
            times.push_back((uint64_t)xyzt[i * 4 + 3]);
        }
        if (rate != 1) {
            vertices.clear();
            for (size_t i = 0; i < pts.size(); i++) { vertices.insert(vertices.end(), pts[i].begin(), pts[i].end()); vertices.push_back((float)times[i]); }
        }
        def.initialiseGraph(pts, times);
    }
    static Constraint make(const float* src, const float* target, uint64_t st, uint64_t tt, bool relative, bool pin)
    {
        Constraint c;
        std::memcpy(c.src, src, 12); std::memcpy(c.target, target, 12);
        c.srcTime = st; c.targetTime = tt; c.relative = relative; c.pin = pin; c.srcPointPoolId = c.tarPointPoolId = -1;
        return c;
    }
    std::vector<Constraint> constraints;
};

#endif
