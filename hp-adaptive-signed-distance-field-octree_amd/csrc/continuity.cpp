// Continuity post-process of Octree::Create (Octree.cpp:341-344, 1250-1762), host side.
//
// The north_star keeps this step on the host (the reference hands it to Eigen's sparse CG); it is the step
// right behind the GPU build and part of Create() whenever config.continuity.enforce is set (the default,
// Config.cpp:9).  What the reference does with a polling thread pool, a std::map, a triplet list and Eigen is
// restated here as:
//
//   1. face pairs      NodeProc / FaceProc (:1549-1612): one traversal from the root lists every pair of
//                      face-adjacent leaves (ordered along the face normal), in the reference's job order.
//   2. row assembly    the matrix is built row-block by row-block, one leaf at a time, in parallel: a leaf's rows
//                      only receive contributions of the faces it touches, so every thread owns its rows and
//                      duplicates are summed in face order -- no triplet list, no sort, no atomics, and the
//                      result does not depend on the thread count.
//                      equal depths   EvaluateSharedFaceIntegralAnalytically (:1459-1546), same arithmetic
//                      else           EvaluateSharedFaceIntegralNumerically (:1250-1456).  The integrand is a
//                                     product of one-dimensional factors, so the (maxDegree+1)^2-point tensor
//                                     quadrature is evaluated as two 1-D quadratures per pair of orders
//                                     (same value up to rounding; entries with |(f32)v| <= EPSILON_F32 are
//                                     dropped as in :1337, :1391, :1448)
//   3. solve           (M + strength I) x = strength c, initial guess strength c (:1724-1756), preconditioned
//                      CG with Eigen's stopping rule |r|^2 < tol^2 |b|^2.  Eigen's IncompleteCholesky is not
//                      restated (Eigen is unpinned in the reference and absent here): the preconditioner is
//                      Jacobi, so the solution agrees with the reference's to solver tolerance only.
//                      Reductions run over fixed chunks in a fixed order: identical for any thread count,
//                      hence identical on every rank of a sharded Create.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "block_check.hpp"
#include "continuity.hpp"
#include "launch.hpp"
#include "runtime.hpp"
#include "tables.hpp"

namespace hpsdf {
namespace {

constexpr float kEpsF32 = 0.000001f;  // Include/Utility/Literals.h:14
constexpr uint64_t kLeafMarker = ~0ull;

// ---- a small persistent worker pool (the reference's ContinuityThreadPool, minus the polling) -------------
class Pool {
   public:
    explicit Pool(unsigned n) : n_(n < 1 ? 1 : n) {
        for (unsigned t = 1; t < n_; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            epoch_.fetch_add(1);
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    unsigned size() const { return n_; }
    // runs fn(chunk) for chunk in [0, nChunks), chunks handed out dynamically; returns when all are done
    void forEach(uint64_t nChunks, const std::function<void(uint64_t)>& fn) {
        if (nChunks == 0) return;
        if (n_ == 1 || nChunks == 1) {
            for (uint64_t c = 0; c < nChunks; ++c) fn(c);
            return;
        }
        fn_ = &fn;
        total_ = nChunks;
        next_.store(0, std::memory_order_relaxed);
        pending_.store(n_ - 1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> g(m_);  // pairs with the sleepers' predicate check
            epoch_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        work();
        // short spin for the stragglers (regions are microseconds long), then sleep: host containers with a CPU
        // quota punish long spinning
        for (unsigned spins = 0; spins < 256 && pending_.load(std::memory_order_acquire) != 0; ++spins) relax();
        if (pending_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        fn_ = nullptr;
    }

   private:
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    void work() {
        for (;;) {
            const uint64_t c = next_.fetch_add(1, std::memory_order_relaxed);
            if (c >= total_) break;
            (*fn_)(c);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // spin briefly for the next region (they come back to back during the solve), then sleep
            unsigned spins = 0;
            while (epoch_.load(std::memory_order_acquire) == seen && spins < 256) {
                relax();
                ++spins;
            }
            if (epoch_.load(std::memory_order_acquire) == seen) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_.load(std::memory_order_acquire) != seen; });
            }
            seen = epoch_.load(std::memory_order_acquire);
            if (stop_) return;
            work();
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> g(m_);
                done_.notify_one();
            }
        }
    }
    unsigned n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(uint64_t)>* fn_ = nullptr;
    std::atomic<uint64_t> next_{0}, epoch_{0};
    std::atomic<unsigned> pending_{0};
    uint64_t total_ = 0;
    std::atomic<bool> stop_{false};
};

struct Pair {
    uint32_t a, b;  // a has the smaller aabb.min along dim (:1593-1594)
    uint8_t dim;
};

struct View {
    const hpsdf_node* nodes;
    uint64_t nNodes;
};

// Include/HP/Utility.h:166-196
struct FaceLookup {
    unsigned v[3][4][2];
    FaceLookup() {
        for (unsigned i = 0; i < 3; ++i) {
            unsigned i0 = 0, i1 = 0;
            const unsigned modVal1 = 1u << i, modVal = 1u << (i + 1), valsPerMod = 1u << (2 - i);
            for (unsigned j = 0; j < valsPerMod; ++j) {
                for (unsigned k = 0; k < modVal1; ++k) v[i][i0++][0] = k + j * modVal;
                for (unsigned k = modVal1; k < modVal; ++k) v[i][i1++][1] = k + j * modVal;
            }
        }
    }
};

void faceProc(const View& t, const FaceLookup& L, uint64_t A, uint64_t B, unsigned dim, std::vector<Pair>& out) {
    const hpsdf_node& nA = t.nodes[A];
    const hpsdf_node& nB = t.nodes[B];
    const bool aHas = nA.child_idx != kLeafMarker, bHas = nB.child_idx != kLeafMarker;
    if (aHas || bHas) {  // :1583-1588
        for (unsigned i = 0; i < 4; ++i)
            faceProc(t, L, aHas ? nA.child_idx + L.v[dim][i][1] : A, bHas ? nB.child_idx + L.v[dim][i][0] : B, dim, out);
        return;
    }
    // :1593-1604.  A leaf pair shares exactly one face and is reached by one path from the root, and NodeProc(0)
    // visits every node, so the reference's later NodeProc(i) calls (:1677-1680) only find pairs its procMap
    // already holds: the traversal from the root alone produces its job list.
    const bool aFirst = nA.aabb_min[dim] < nB.aabb_min[dim];
    out.push_back(Pair{(uint32_t)(aFirst ? A : B), (uint32_t)(aFirst ? B : A), (uint8_t)dim});
}

void nodeProc(const View& t, const FaceLookup& L, uint64_t idx, std::vector<Pair>& out) {  // :1549-1571
    const hpsdf_node& n = t.nodes[idx];
    if (n.child_idx == kLeafMarker) return;
    for (unsigned i = 0; i < 8; ++i) nodeProc(t, L, n.child_idx + i, out);
    for (unsigned i = 0; i < 3; ++i)
        for (unsigned j = 0; j < 4; ++j) faceProc(t, L, n.child_idx + L.v[i][j][0], n.child_idx + L.v[i][j][1], i, out);
}

// Octree::LpX, :988-1004
double lpx(const Tables& T, unsigned p, double x) {
    double m2 = 0.0, m1 = 1.0;
    for (unsigned i = 1; i <= p; ++i) {
        const double l = T.recurrence[i][0] * x * m1 - T.recurrence[i][1] * m2;
        m2 = m1, m1 = l;
    }
    return m1;
}

// Per-thread scratch of the row assembly.
struct Scratch {
    std::vector<double> own;       // [nL][nL] accumulated own block
    std::vector<uint8_t> ownSet;   // entry was emitted at least once (an explicit zero stays an entry, as in Eigen)
    std::vector<double> cross;     // [nL][nO] one cross block
    std::vector<uint8_t> crossSet;
    std::vector<uint64_t> cols;    // fragment under construction
    std::vector<double> vals;
    std::vector<uint32_t> rowLen;
};

struct Incident {
    uint32_t pair;
    uint8_t side;  // 0: the leaf is the pair's a (its +dim face), 1: it is b (its -dim face)
};

// One-dimensional quadrature tables of a non-conforming face, :1264-1314.
struct NumericFace {
    unsigned n = 0, m1 = 0, m2 = 0;
    double scale12 = 0.0;
    // T[side][axis 0 = m1, 1 = m2][p][q]: LpX(p, coordinate of sample q on that side)
    double T[2][2][kMaxDegree + 1][kMaxDegree + 1];
    double w[kMaxDegree + 1];
};

void prepareNumericFace(const Tables& T, const hpsdf_node& nA, const hpsdf_node& nB, unsigned dim, NumericFace& f) {
    f.m1 = (dim + 1) % 3, f.m2 = (dim + 2) % 3;
    double scale[3];
    for (int a = 0; a < 3; ++a) {  // sharedFace = A.clamp(B), sizes * 0.5 (:1264-1265)
        const float lo = nA.aabb_min[a] > nB.aabb_min[a] ? nA.aabb_min[a] : nB.aabb_min[a];
        const float hi = nA.aabb_max[a] < nB.aabb_max[a] ? nA.aabb_max[a] : nB.aabb_max[a];
        scale[a] = (double)(hi - lo) * 0.5;
    }
    f.scale12 = scale[f.m1] * scale[f.m2];
    const unsigned maxDegree = nA.degree > nB.degree ? nA.degree : nB.degree;  // :1269
    const uint64_t gqStart = T.sumToN[maxDegree];
    f.n = (unsigned)(T.sumToN[maxDegree + 1] - gqStart);
    const unsigned depthDiff = nA.depth > nB.depth ? nA.depth - nB.depth : nB.depth - nA.depth;
    const double invDist = 1.0 / std::pow(2.0, (double)depthDiff);  // :1275
    const hpsdf_node& s = nA.depth > nB.depth ? nA : nB;           // the deeper cell
    const hpsdf_node& l = nA.depth > nB.depth ? nB : nA;
    double invT[2];
    const unsigned ms[2] = {f.m1, f.m2};
    for (int q = 0; q < 2; ++q) {  // :1278-1290
        const unsigned m = ms[q];
        const float cs = (s.aabb_min[m] + s.aabb_max[m]) / 2.0f, cl = (l.aabb_min[m] + l.aabb_max[m]) / 2.0f;
        invT[q] = (double)(cs - cl) / ((double)(s.aabb_max[m] - s.aabb_min[m]) * 0.5) * invDist;
    }
    for (unsigned q = 0; q < f.n; ++q) {
        const double r = T.roots[gqStart + q];
        f.w[q] = T.weights[gqStart + q];
        for (int ax = 0; ax < 2; ++ax) {
            // the coarser side's samples are moved onto the smaller face (:1309-1314, :1363-1372, :1417-1421)
            const double ua = nB.depth > nA.depth ? r * invDist + invT[ax] : r;
            const double ub = nA.depth > nB.depth ? r * invDist + invT[ax] : r;
            for (unsigned p = 0; p <= (unsigned)kMaxDegree; ++p) {
                f.T[0][ax][p][q] = lpx(T, p, ua);
                f.T[1][ax][p][q] = lpx(T, p, ub);
            }
        }
    }
}

struct Assembler {
    const Tables& T;
    View tree;
    const std::vector<Pair>& pairs;
    std::vector<uint64_t> incStart;  // per node
    std::vector<Incident> inc;
    double faceP[kMaxDegree + 1], faceM[kMaxDegree + 1];  // LpX(p, +1), LpX(p, -1)

    Assembler(const Tables& T_, View tr, const std::vector<Pair>& p) : T(T_), tree(tr), pairs(p) {
        for (unsigned q = 0; q <= (unsigned)kMaxDegree; ++q) faceP[q] = lpx(T, q, 1.0), faceM[q] = lpx(T, q, -1.0);
        incStart.assign(tree.nNodes + 1, 0);
        for (const Pair& pr : pairs) incStart[pr.a + 1]++, incStart[pr.b + 1]++;
        for (uint64_t i = 0; i < tree.nNodes; ++i) incStart[i + 1] += incStart[i];
        inc.resize(incStart[tree.nNodes]);
        std::vector<uint64_t> fill(incStart.begin(), incStart.end() - 1);
        for (uint32_t q = 0; q < pairs.size(); ++q) {  // pair order = the reference's job order
            inc[fill[pairs[q].a]++] = Incident{q, 0};
            inc[fill[pairs[q].b]++] = Incident{q, 1};
        }
    }

    // rows of leaf `L`: appends (col, val) per row to s.cols / s.vals, lengths to s.rowLen
    void leafRows(uint64_t L, Scratch& s) const {
        const hpsdf_node& nL = tree.nodes[L];
        const unsigned nl = (unsigned)T.coeffCount[nL.degree];
        s.own.assign((size_t)nl * nl, 0.0);
        s.ownSet.assign((size_t)nl * nl, 0);
        s.cols.clear();
        s.vals.clear();
        s.rowLen.assign(nl, 0);
        // cross blocks are kept per face until the rows are written out (columns ascending = neighbours by offset)
        struct CrossRef {
            uint64_t colStart;
            unsigned nO;
            size_t off;  // into crossPool / crossSetPool
        };
        std::vector<CrossRef> refs;
        s.cross.clear();
        s.crossSet.clear();
        for (uint64_t q = incStart[L]; q < incStart[L + 1]; ++q) {
            const Pair& pr = pairs[inc[q].pair];
            const unsigned side = inc[q].side, dim = pr.dim, m1 = (dim + 1) % 3, m2 = (dim + 2) % 3;
            const uint64_t O = side ? pr.a : pr.b;
            const hpsdf_node& nO = tree.nodes[O];
            const unsigned no = (unsigned)T.coeffCount[nO.degree];
            refs.push_back(CrossRef{nO.coeffs_start, no, s.cross.size()});
            s.cross.resize(s.cross.size() + (size_t)nl * no, 0.0);
            s.crossSet.resize(s.crossSet.size() + (size_t)nl * no, 0);
            double* X = s.cross.data() + refs.back().off;
            uint8_t* XS = s.crossSet.data() + refs.back().off;
            const double* fL = side ? faceM : faceP;  // this leaf's face value of LpX
            const double* fO = side ? faceP : faceM;
            if (nL.depth == nO.depth) {  // EvaluateSharedFaceIntegralAnalytically, :1459-1546
                for (unsigned i = 0; i < nl; ++i) {
                    const uint64_t* bi = T.basisIndex[i];
                    for (unsigned j = 0; j < nl; ++j) {
                        const uint64_t* bj = T.basisIndex[j];
                        if (bi[m1] != bj[m1] || bi[m2] != bj[m2]) continue;
                        double integral = 1.0;
                        integral *= fL[bi[dim]];
                        integral *= T.normalisedLengths[bi[dim]][nL.depth];
                        integral *= fL[bj[dim]];
                        integral *= T.normalisedLengths[bj[dim]][nL.depth];
                        s.own[(size_t)i * nl + j] += integral;
                        s.ownSet[(size_t)i * nl + j] = 1;
                    }
                    for (unsigned j = 0; j < no; ++j) {
                        const uint64_t* bj = T.basisIndex[j];
                        if (bi[m1] != bj[m1] || bi[m2] != bj[m2]) continue;
                        // :1511-1515 multiplies a's factors first, then b's; the entry is symmetric in the pair
                        const uint64_t* ba = side ? bj : bi;
                        const uint64_t* bb = side ? bi : bj;
                        const hpsdf_node& na = side ? nO : nL;
                        const hpsdf_node& nb = side ? nL : nO;
                        double integral = -1.0;
                        integral *= faceP[ba[dim]];
                        integral *= T.normalisedLengths[ba[dim]][na.depth];
                        integral *= faceM[bb[dim]];
                        integral *= T.normalisedLengths[bb[dim]][nb.depth];
                        X[(size_t)i * no + j] = integral;
                        XS[(size_t)i * no + j] = 1;
                    }
                }
            } else {  // EvaluateSharedFaceIntegralNumerically, :1250-1456
                const hpsdf_node& na = side ? nO : nL;
                const hpsdf_node& nb = side ? nL : nO;
                NumericFace f;
                prepareNumericFace(T, na, nb, dim, f);
                const unsigned sL = side, sO = side ^ 1u;  // table side of this leaf / of the other leaf
                // 1-D quadratures: I[ax][p][q] = sum_x w_x T[sI][ax][p][x] T[sJ][ax][q][x]
                double ILL[2][kMaxDegree + 1][kMaxDegree + 1], ILO[2][kMaxDegree + 1][kMaxDegree + 1];
                for (int ax = 0; ax < 2; ++ax)
                    for (unsigned p = 0; p <= nL.degree; ++p) {
                        for (unsigned r = 0; r <= nL.degree; ++r) {
                            double acc = 0.0;
                            for (unsigned x = 0; x < f.n; ++x) acc += f.w[x] * f.T[sL][ax][p][x] * f.T[sL][ax][r][x];
                            ILL[ax][p][r] = acc;
                        }
                        for (unsigned r = 0; r <= nO.degree; ++r) {
                            double acc = 0.0;
                            for (unsigned x = 0; x < f.n; ++x) acc += f.w[x] * f.T[sL][ax][p][x] * f.T[sO][ax][r][x];
                            ILO[ax][p][r] = acc;
                        }
                    }
                for (unsigned i = 0; i < nl; ++i) {
                    const uint64_t* bi = T.basisIndex[i];
                    const double wi = T.normalisedLengths[bi[0]][nL.depth] * T.normalisedLengths[bi[1]][nL.depth] *
                                      T.normalisedLengths[bi[2]][nL.depth];
                    for (unsigned j = 0; j < nl; ++j) {
                        const uint64_t* bj = T.basisIndex[j];
                        const double wj = T.normalisedLengths[bj[0]][nL.depth] * T.normalisedLengths[bj[1]][nL.depth] *
                                          T.normalisedLengths[bj[2]][nL.depth];
                        const double integral = ILL[0][bi[m1]][bj[m1]] * ILL[1][bi[m2]][bj[m2]] * (fL[bi[dim]] * fL[bj[dim]]) *
                                                (f.scale12 * (wi * wj));
                        if (std::fabs((float)integral) > kEpsF32) {  // :1337 / :1448
                            s.own[(size_t)i * nl + j] += integral;
                            s.ownSet[(size_t)i * nl + j] = 1;
                        }
                    }
                    for (unsigned j = 0; j < no; ++j) {
                        const uint64_t* bj = T.basisIndex[j];
                        const double wj = T.normalisedLengths[bj[0]][nO.depth] * T.normalisedLengths[bj[1]][nO.depth] *
                                          T.normalisedLengths[bj[2]][nO.depth];
                        const double integral = ILO[0][bi[m1]][bj[m1]] * ILO[1][bi[m2]][bj[m2]] * (fL[bi[dim]] * fO[bj[dim]]) *
                                                (f.scale12 * (wi * wj)) * -1.0;
                        if (std::fabs((float)integral) > kEpsF32) {  // :1391
                            X[(size_t)i * no + j] = integral;
                            XS[(size_t)i * no + j] = 1;
                        }
                    }
                }
            }
        }
        // write the rows: columns ascending.  Blocks of distinct leaves do not overlap, so ordering the blocks by
        // their first column orders the row.
        std::vector<unsigned> order(refs.size());
        for (unsigned k = 0; k < refs.size(); ++k) order[k] = k;
        for (unsigned a = 1; a < order.size(); ++a)  // tiny: at most 6 * 4^depthDiff neighbours
            for (unsigned b = a; b > 0 && refs[order[b]].colStart < refs[order[b - 1]].colStart; --b) std::swap(order[b], order[b - 1]);
        const uint64_t ownStart = nL.coeffs_start;
        for (unsigned i = 0; i < nl; ++i) {
            uint32_t len = 0;
            bool ownDone = false;
            auto emitOwn = [&] {
                for (unsigned j = 0; j < nl; ++j)
                    if (s.ownSet[(size_t)i * nl + j]) {
                        s.cols.push_back(ownStart + j);
                        s.vals.push_back(s.own[(size_t)i * nl + j]);
                        ++len;
                    }
                ownDone = true;
            };
            for (unsigned k : order) {
                const CrossRef& r = refs[k];
                if (!ownDone && ownStart < r.colStart) emitOwn();
                const double* X = s.cross.data() + r.off + (size_t)i * r.nO;
                const uint8_t* XS = s.crossSet.data() + r.off + (size_t)i * r.nO;
                for (unsigned j = 0; j < r.nO; ++j)
                    if (XS[j]) {
                        s.cols.push_back(r.colStart + j);
                        s.vals.push_back(X[j]);
                        ++len;
                    }
            }
            if (!ownDone) emitOwn();
            s.rowLen[i] = len;
        }
    }
};

struct Csr {
    uint64_t n = 0;
    std::vector<uint64_t> rowPtr, col;
    std::vector<double> val;
};

struct ParsedBlock {
    uint64_t nCoeffs = 0, nNodes = 0;
    double* coeffs = nullptr;
    std::vector<hpsdf_node> nodes;
    hpsdf_config cfg{};
};

int parseBlock(void* block, size_t size, ParsedBlock& out, std::string& err) {
    if (!block || size < 16 + sizeof(hpsdf_config)) {
        err = "block too small";
        return HPSDF_ERR_BAD_BLOCK;
    }
    uint8_t* p = (uint8_t*)block;
    std::memcpy(&out.nCoeffs, p, 8);
    if (out.nCoeffs > (size - 16 - sizeof(hpsdf_config)) / 8) {
        err = "coefficient count exceeds block";
        return HPSDF_ERR_BAD_BLOCK;
    }
    out.coeffs = (double*)(p + 8);
    std::memcpy(&out.nNodes, p + 8 + 8 * out.nCoeffs, 8);
    const size_t need = 8 + 8 * (size_t)out.nCoeffs + 8 + sizeof(hpsdf_node) * (size_t)out.nNodes + sizeof(hpsdf_config);
    if (out.nNodes == 0 || need != size) {
        err = "node count does not match block size";
        return HPSDF_ERR_BAD_BLOCK;
    }
    out.nodes.resize(out.nNodes);
    std::memcpy(out.nodes.data(), p + 16 + 8 * out.nCoeffs, sizeof(hpsdf_node) * out.nNodes);
    std::memcpy(&out.cfg, p + 16 + 8 * out.nCoeffs + sizeof(hpsdf_node) * out.nNodes, sizeof out.cfg);
    // untrusted bytes: walk from the root (block_check.hpp) -- children in range without wrap-around, no node reached
    // twice (nodeProc recurses along child indices), depths consistent, leaf ranges inside the store and pairwise
    // disjoint (the assembly writes one row block per leaf, in parallel)
    const Tables& T = tables();
    if (out.nodes[0].degree == kInteriorDegree && out.nNodes < 9) {
        err = "an interior root needs its 8 children";
        return HPSDF_ERR_BAD_BLOCK;
    }
    BlockTreeInfo walk;
    const int vrc = checkBlockTree(out.nodes.data(), out.nNodes, out.nCoeffs, T.coeffCount, true, true, &walk, err);
    if (vrc) return vrc;
    for (uint64_t i = 0; i < out.nNodes; ++i)
        if (!walk.reached[i]) {
            err = "node not reachable from the root";
            return HPSDF_ERR_BAD_BLOCK;
        }
    return HPSDF_OK;
}

// a chunk of leaves' rows before they are copied to their place in the CSR arrays
struct Fragment {
    std::vector<uint64_t> cols;
    std::vector<double> vals;
};

// frags: scratch the caller may keep between calls (capacity is reused)
void assemble(const ParsedBlock& b, Pool& pool, Csr& M, hpsdf_continuity_stats& st, std::vector<Fragment>& frags) {
    const Tables& T = tables();
    const View view{b.nodes.data(), b.nNodes};
    std::vector<Pair> pairs;
    static const FaceLookup L;
    nodeProc(view, L, 0, pairs);
    st.n_pairs = pairs.size();
    for (const Pair& p : pairs) (b.nodes[p.a].depth == b.nodes[p.b].depth ? st.n_pairs_analytic : st.n_pairs_numeric)++;
    const Assembler A(T, view, pairs);
    std::vector<uint64_t> leaves;
    for (uint64_t i = 0; i < b.nNodes; ++i)
        if (b.nodes[i].child_idx == kLeafMarker) leaves.push_back(i);
    // leaves in chunks; every chunk builds a private CSR fragment
    const uint64_t chunk = 32, nChunks = (leaves.size() + chunk - 1) / chunk;
    if (frags.size() < nChunks) frags.resize(nChunks);
    for (uint64_t c = 0; c < nChunks; ++c) frags[c].cols.clear(), frags[c].vals.clear();
    M.n = b.nCoeffs;
    M.rowPtr.assign(M.n + 1, 0);
    pool.forEach(nChunks, [&](uint64_t c) {
        Scratch s;
        Fragment& f = frags[c];
        for (uint64_t q = c * chunk; q < std::min<uint64_t>(leaves.size(), (c + 1) * chunk); ++q) {
            const uint64_t Lf = leaves[q];
            A.leafRows(Lf, s);
            f.cols.insert(f.cols.end(), s.cols.begin(), s.cols.end());
            f.vals.insert(f.vals.end(), s.vals.begin(), s.vals.end());
            const uint64_t r0 = b.nodes[Lf].coeffs_start;
            for (size_t i = 0; i < s.rowLen.size(); ++i) M.rowPtr[r0 + i + 1] = s.rowLen[i];  // rows are disjoint
        }
    });
    for (uint64_t r = 0; r < M.n; ++r) M.rowPtr[r + 1] += M.rowPtr[r];
    M.col.resize(M.rowPtr[M.n]);
    M.val.resize(M.rowPtr[M.n]);
    pool.forEach(nChunks, [&](uint64_t c) {
        const Fragment& f = frags[c];
        size_t off = 0;
        for (uint64_t q = c * chunk; q < std::min<uint64_t>(leaves.size(), (c + 1) * chunk); ++q) {
            const hpsdf_node& n = b.nodes[leaves[q]];
            const uint64_t r0 = n.coeffs_start, r1 = r0 + T.coeffCount[n.degree];
            const size_t len = (size_t)(M.rowPtr[r1] - M.rowPtr[r0]);
            std::memcpy(M.col.data() + M.rowPtr[r0], f.cols.data() + off, len * sizeof(uint64_t));
            std::memcpy(M.val.data() + M.rowPtr[r0], f.vals.data() + off, len * sizeof(double));
            off += len;
        }
    });
    st.nnz = M.rowPtr[M.n];
}

// ---- vector kernels over fixed chunks (deterministic for any thread count) ---------------------------------
constexpr uint64_t kVecChunk = kCgChunk;  // launch.hpp: the device solve sums over the same chunks

struct Vec {
    Pool& pool;
    uint64_t n, nChunks;
    std::vector<double> partial;
    Vec(Pool& p, uint64_t n_) : pool(p), n(n_), nChunks((n_ + kVecChunk - 1) / kVecChunk), partial(nChunks ? nChunks : 1) {}
    template <typename F>
    void each(F&& f) {  // f(begin, end)
        pool.forEach(nChunks, [&](uint64_t c) { f(c * kVecChunk, std::min(n, (c + 1) * kVecChunk)); });
    }
    double dot(const double* a, const double* b) {
        pool.forEach(nChunks, [&](uint64_t c) {
            double prod[kVecChunk];
            const uint64_t lo = c * kVecChunk, cnt = std::min(n, lo + kVecChunk) - lo;
            for (uint64_t i = 0; i < cnt; ++i) prod[i] = a[lo + i] * b[lo + i];
            partial[c] = cgChunkSum(prod, cnt);
        });
        return cgCombine(partial.data(), nChunks);
    }
    // y = (M + shift I) x
    void spmv(const Csr& M, double shift, const double* x, double* y) {
        each([&](uint64_t b, uint64_t e) {
            for (uint64_t r = b; r < e; ++r) {
                double s = shift * x[r];
                for (uint64_t q = M.rowPtr[r]; q < M.rowPtr[r + 1]; ++q) s += M.val[q] * x[M.col[q]];
                y[r] = s;
            }
        });
    }
};

double nowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

unsigned poolSize(uint64_t requested) {
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 1;
    uint64_t n = requested ? requested : hc;
    if (n > hc) n = hc;
    if (n > 32) n = 32;  // bandwidth-bound SpMV on a few MB: more threads only add barrier cost
    return (unsigned)n;
}

}  // namespace

int continuityMatrix(const void* block, size_t size, uint64_t threads, uint64_t** rowPtr, uint64_t** col, double** val,
                     hpsdf_continuity_stats* stats, std::string& err) {
    ParsedBlock b;
    int rc = parseBlock(const_cast<void*>(block), size, b, err);
    if (rc) return rc;
    Pool pool(poolSize(threads));
    Csr M;
    hpsdf_continuity_stats st;
    std::memset(&st, 0, sizeof st);
    std::vector<Fragment> frags;
    assemble(b, pool, M, st, frags);
    *rowPtr = (uint64_t*)std::malloc(sizeof(uint64_t) * (M.n + 1));
    *col = (uint64_t*)std::malloc(sizeof(uint64_t) * (M.col.size() ? M.col.size() : 1));
    *val = (double*)std::malloc(sizeof(double) * (M.val.size() ? M.val.size() : 1));
    if (!*rowPtr || !*col || !*val) {
        std::free(*rowPtr), std::free(*col), std::free(*val);
        *rowPtr = *col = nullptr, *val = nullptr;
        err = "malloc failed";
        return HPSDF_ERR_OUT_OF_MEMORY;
    }
    std::memcpy(*rowPtr, M.rowPtr.data(), sizeof(uint64_t) * (M.n + 1));
    std::memcpy(*col, M.col.data(), sizeof(uint64_t) * M.col.size());
    std::memcpy(*val, M.val.data(), sizeof(double) * M.val.size());
    if (stats) *stats = st;
    return HPSDF_OK;
}

int continuityMatrixDevice(hpsdf_ctx* ctx, const void* block, size_t size, uint64_t** rowPtr, uint64_t** col, double** val,
                           hpsdf_continuity_stats* stats, std::string& err) {
    ParsedBlock b;
    int rc = parseBlock(const_cast<void*>(block), size, b, err);
    if (rc) return rc;
    hpsdf_continuity_stats st;
    std::memset(&st, 0, sizeof st);
    ContinuityDeviceMatrix dm;
    int fallback = 0;
    rc = continuityAssembleDevice(ctx, b.nodes.data(), b.nNodes, b.nCoeffs, dm, st, &fallback, err);
    if (rc) return rc;
    if (fallback) {
        err = "this tree is left to the host assembler";
        return HPSDF_ERR_UNSUPPORTED;
    }
    std::vector<uint32_t> c32(dm.nnz ? dm.nnz : 1);
    *rowPtr = (uint64_t*)std::malloc(sizeof(uint64_t) * (dm.n + 1));
    *col = (uint64_t*)std::malloc(sizeof(uint64_t) * (dm.nnz ? dm.nnz : 1));
    *val = (double*)std::malloc(sizeof(double) * (dm.nnz ? dm.nnz : 1));
    if (!*rowPtr || !*col || !*val) {
        std::free(*rowPtr), std::free(*col), std::free(*val);
        *rowPtr = *col = nullptr, *val = nullptr;
        err = "malloc failed";
        return HPSDF_ERR_OUT_OF_MEMORY;
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);  // the assembly's last kernel runs on the context's stream
    if (e == hipSuccess) e = hipMemcpy(*rowPtr, dm.dRowPtr, sizeof(uint64_t) * (dm.n + 1), hipMemcpyDeviceToHost);
    if (e == hipSuccess && dm.nnz) e = hipMemcpy(c32.data(), dm.dCol, sizeof(uint32_t) * dm.nnz, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dm.nnz) e = hipMemcpy(*val, dm.dVal, sizeof(double) * dm.nnz, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        std::free(*rowPtr), std::free(*col), std::free(*val);
        *rowPtr = *col = nullptr, *val = nullptr;
        err = std::string("continuity matrix download: ") + hipGetErrorString(e);
        return HPSDF_ERR_HIP;
    }
    for (uint64_t q = 0; q < dm.nnz; ++q) (*col)[q] = c32[q];
    if (stats) *stats = st;
    return HPSDF_OK;
}

// Octree::PerformContinuityPostProcess, :1717-1762, in place on the serialised block
// What a context keeps between post-processes: the assembled matrix, the host solver's vectors and the device buffer
// of the device solve.
struct Keep {
    Csr M;
    std::vector<double> v[8];
    std::vector<uint32_t> col32;
    std::vector<Fragment> frags;
    std::unique_ptr<Pool> asmPool;  // the assembly's workers sleep between calls instead of being spawned and joined
    ContinuityDeviceMatrix dm;  // the matrix when it was assembled on the device (continuity_asm.hip)
    char* dBase = nullptr;
    uint64_t dCap = 0;
    int device = -1;
    uint32_t* pinFlag = nullptr;  // two words of pinned host memory the solve's check kernel writes (CgDev::hostFlag)
    uint32_t flagStamp = 0;
    ~Keep() {
        if (dBase) {
            (void)hipSetDevice(device);
            (void)hipFree(dBase);
        }
        if (pinFlag) (void)hipHostFree(pinFlag);
    }
};

// The whole solve on the device: uploads the matrix and the block's coefficients, lets cg.hip set up the system
// (right-hand side, initial guess, Jacobi diagonal, first residual, threshold -- the statements of the host path below,
// same arithmetic), runs batches of iterations until the stop flag is up, brings x and the statistics back.
// Returns 0 or an HPSDF_ERR code (err filled).
static int solveOnDevice(hpsdf_ctx* ctx, Keep& keep, const ContinuityDeviceMatrix* dm, const double* coeffs, double lambda, double tol,
                         int maxIter, double* xOut, hpsdf_continuity_stats& st, std::string& err) {
    const Csr& M = keep.M;  // (used when the host assembled: dm == nullptr)
    const uint64_t n = dm ? dm->n : M.n, nChunks = (n + kCgChunk - 1) / kCgChunk;
    if (n >= 0xFFFFFFFFull) {
        err = "continuity system too large for 32-bit column indices";
        return HPSDF_ERR_UNSUPPORTED;
    }
    const bool trace = std::getenv("HPSDF_TRACE") != nullptr;
    const double tt0 = nowMs();
    auto al = [](uint64_t b) { return (b + 255) & ~255ull; };
    const uint64_t nnz = dm ? dm->nnz : M.rowPtr[n];
    std::vector<uint32_t>& col32 = keep.col32;
    if (!dm) {
        col32.resize(nnz ? nnz : 1);
        for (uint64_t q = 0; q < nnz; ++q) col32[q] = (uint32_t)M.col[q];
    }
    const double tt1 = nowMs();
    const uint64_t vecB = al(n * 8), partB = al(nChunks * 8);
    uint64_t maxRow = dm ? dm->maxRow : 0;
    if (!dm)
        for (uint64_t r = 0; r < n; ++r) maxRow = std::max<uint64_t>(maxRow, M.rowPtr[r + 1] - M.rowPtr[r]);
    // the SpMV is cut by entries (cg.hip) unless a row is too long for a workgroup's tail or the counts outgrow 32 bits
    const uint64_t nWg = (maxRow <= kCgRowTail && nnz < 0xFFFFFFFFull) ? std::max<uint64_t>(1, (nnz + kCgEntriesPerWg - 1) / kCgEntriesPerWg) : 0;
    const uint64_t total = 9 * vecB + 3 * partB + 256 + al((nWg + 1) * 4) + (dm ? 0 : al((n + 1) * 8) + al(nnz * 4 + 4) + al(nnz * 8 + 8));
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess && (keep.dCap < total || keep.device != ctx->device)) {
        if (keep.dBase) {
            (void)hipSetDevice(keep.device);
            (void)hipFree(keep.dBase);
            (void)hipSetDevice(ctx->device);
        }
        keep.dBase = nullptr, keep.dCap = 0, keep.device = ctx->device;
        e = hipMalloc((void**)&keep.dBase, total + total / 4);
        if (e == hipSuccess) keep.dCap = total + total / 4;
    }
    if (e != hipSuccess) {
        err = std::string("continuity solve: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? HPSDF_ERR_OUT_OF_MEMORY : HPSDF_ERR_HIP;
    }
    char* base = keep.dBase;
    char* cur = base;
    auto take = [&](uint64_t bytes) {
        char* q = cur;
        cur += al(bytes);
        return q;
    };
    CgDev d;
    d.hostFlag = nullptr;
    d.n = n, d.nChunks = nChunks;
    double* dC = (double*)take(n * 8);
    d.c = dC;
    d.dinv = (double*)take(n * 8), d.rhs = (double*)take(n * 8);
    d.x = (double*)take(n * 8), d.r = (double*)take(n * 8), d.p = (double*)take(n * 8), d.z = (double*)take(n * 8);
    d.tmp = (double*)take(n * 8);
    d.partA = (double*)take(nChunks * 8), d.partB = (double*)take(nChunks * 8), d.partC = (double*)take(nChunks * 8);
    d.s = (CgScalars*)take(sizeof(CgScalars));
    d.wgRow = (uint32_t*)take((nWg + 1) * 4);
    d.nWg = (uint32_t)nWg;
    uint64_t* dRowPtr = nullptr;
    uint32_t* dCsrCol = nullptr;
    double* dCsrVal = nullptr;
    if (dm) {
        d.rowPtr = dm->dRowPtr, d.col = dm->dCol, d.val = dm->dVal;
    } else {
        dRowPtr = (uint64_t*)take((n + 1) * 8);
        dCsrCol = (uint32_t*)take(nnz * 4 + 4);
        dCsrVal = (double*)take(nnz * 8 + 8);
        d.rowPtr = dRowPtr, d.col = dCsrCol, d.val = dCsrVal;
    }
    CgScalars s;
    std::memset(&s, 0, sizeof s);
    s.lambda = lambda, s.tol = tol, s.maxIter = maxIter;
    hipStream_t stm = ctx->stream;
    auto up = [&](void* dst, const void* src, uint64_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stm);
    };
    if (!dm) {
        up(dRowPtr, M.rowPtr.data(), (n + 1) * 8);
        up(dCsrCol, col32.data(), nnz * 4), up(dCsrVal, M.val.data(), nnz * 8);
    }
    up(dC, coeffs, n * 8), up(d.s, &s, sizeof s);
    if (e == hipSuccess) e = launchCgStart(stm, d);
    if (e == hipSuccess && trace) e = hipStreamSynchronize(stm);
    const double tt2 = nowMs();
    // Batches of 16 iterations; whether the loop has ended the check kernel behind each batch writes into two words of pinned host
    // memory, which the host watches (~3 us; waking up from hipStreamSynchronize ~20, seven times a solve) -- two milliseconds of
    // watching, then the ordinary wait.
    if (e == hipSuccess && !keep.pinFlag && hipHostMalloc((void**)&keep.pinFlag, 64, hipHostMallocDefault) != hipSuccess) keep.pinFlag = nullptr;
    uint32_t* dFlag = nullptr;
    if (keep.pinFlag && hipHostGetDevicePointer((void**)&dFlag, keep.pinFlag, 0) != hipSuccess) dFlag = nullptr;
    d.hostFlag = dFlag;
    for (int k0 = 0; e == hipSuccess; k0 += 16) {
        const uint32_t stamp = ++keep.flagStamp ? keep.flagStamp : ++keep.flagStamp;
        e = launchCgIterations(stm, d, k0, 16, stamp);
        if (e != hipSuccess) break;
        bool seen = false;
        if (dFlag) {
            const volatile uint32_t* f = keep.pinFlag;
            const double limit = nowMs() + 2.0;
            while (f[1] != stamp && nowMs() < limit) __builtin_ia32_pause();
            if (f[1] == stamp) {
                std::atomic_thread_fence(std::memory_order_acquire);
                s.done = (int32_t)f[0];
                seen = true;
            }
        }
        if (!seen) {
            e = hipMemcpyAsync(&s, d.s, sizeof s, hipMemcpyDeviceToHost, stm);
            if (e == hipSuccess) e = hipStreamSynchronize(stm);
        }
        if (e != hipSuccess || s.done) break;
    }
    const double tt3 = nowMs();
    if (e == hipSuccess) e = launchCgFinish(stm, d);
    if (e == hipSuccess) e = hipMemcpyAsync(&s, d.s, sizeof s, hipMemcpyDeviceToHost, stm);
    if (e == hipSuccess) e = hipMemcpyAsync(xOut, d.x, n * 8, hipMemcpyDeviceToHost, stm);
    if (e == hipSuccess) e = hipStreamSynchronize(stm);
    if (trace)
        std::fprintf(stderr, "[continuity solve] n %llu, %llu non-zeros: 32-bit columns %.2f ms, upload + start %.2f, %d iterations %.2f, finish + download %.2f\n",
                     (unsigned long long)n, (unsigned long long)nnz, tt1 - tt0, tt2 - tt1, (int)s.it, tt3 - tt2, nowMs() - tt3);
    if (e != hipSuccess) {
        err = std::string("continuity solve: ") + hipGetErrorString(e);
        return HPSDF_ERR_HIP;
    }
    if (s.done == 3) std::fill(xOut, xOut + n, 0.0);  // zero right-hand side: the zero solution (Eigen)
    st.iterations = (uint64_t)s.it;
    st.residual = s.rhsNorm2 > 0.0 ? std::sqrt(s.resNorm2 / s.rhsNorm2) : 0.0;
    st.jump_before = s.jumpBefore;
    st.jump_after = s.jumpAfter;
    return HPSDF_OK;
}

int continuityPostProcess(void* block, size_t size, double tol, int maxIter, uint64_t threads,
                          hpsdf_continuity_stats* stats, std::string& err, hpsdf_ctx* ctx) {
    const double tEntry = nowMs();
    ParsedBlock b;
    int rc = parseBlock(block, size, b, err);
    if (rc) return rc;
    hpsdf_continuity_stats st;
    std::memset(&st, 0, sizeof st);
    if (!(b.cfg.continuity_strength > 0.0)) {
        err = "continuity.strength must be > 0 (Config.cpp:28-31)";
        return HPSDF_ERR_INVALID_ARGUMENT;
    }
    if (!(tol > 0.0)) tol = (double)kEpsF32;  // extSolver.setTolerance(EPSILON_F32), :1754
    const unsigned nThreads = poolSize(threads ? threads : b.cfg.thread_count);
    const double t0 = nowMs();
    // The matrix and the solver's vectors (some 60 MB on a 5 k-node tree) live in the context and are reused by the next
    // call: allocating, first-touching and unmapping them anew cost more than the assembly itself (7.8 -> 3.1 ms) and
    // another 3.5 ms behind the solve.  Without a context they are this call's own.
    std::shared_ptr<Keep> own;
    if (ctx) {
        if (!ctx->continuityScratch) ctx->continuityScratch = std::make_shared<Keep>();
        own = std::static_pointer_cast<Keep>(ctx->continuityScratch);
    } else {
        own = std::make_shared<Keep>();
    }
    Keep& keep = *own;
    Csr& M = keep.M;
    const bool deviceSolve = ctx && (b.nCoeffs + kCgChunk - 1) / kCgChunk <= kCgMaxChunksOnDevice;
    // The matrix is assembled where it is used: on the device when the solve runs there (continuity_asm.hip: the same
    // entries bit for bit, and no 20 MB upload); HPSDF_CONTINUITY_HOST_ASSEMBLY=1, and trees the device assembly
    // declines, take the host assembler.
    bool onDevice = false;
    if (deviceSolve) {
        const char* ha = std::getenv("HPSDF_CONTINUITY_HOST_ASSEMBLY");
        if (!(ha && ha[0] == '1')) {
            int fallback = 0;
            rc = continuityAssembleDevice(ctx, b.nodes.data(), b.nNodes, b.nCoeffs, keep.dm, st, &fallback, err);
            if (rc) return rc;
            onDevice = !fallback;
        }
    }
    if (!onDevice) {
        if (!keep.asmPool || keep.asmPool->size() != nThreads) keep.asmPool.reset(new Pool(nThreads));
        assemble(b, *keep.asmPool, M, st, keep.frags);
    }
    const double t1 = nowMs();
    if (deviceSolve) {  // the solve, all of it, on the device
        if (maxIter <= 0) maxIter = (int)std::min<uint64_t>(2 * b.nCoeffs, 0x7FFFFFFF);  // Eigen's default 2n
        std::vector<double>& xd = keep.v[0];
        xd.resize(b.nCoeffs);
        rc = solveOnDevice(ctx, keep, onDevice ? &keep.dm : nullptr, b.coeffs, b.cfg.continuity_strength, tol, maxIter, xd.data(), st, err);
        if (rc) return rc;
        std::memcpy(b.coeffs, xd.data(), sizeof(double) * b.nCoeffs);  // :1756
        st.assemble_ms = t1 - t0;
        st.solve_ms = nowMs() - t1;
        if (std::getenv("HPSDF_TRACE"))
            std::fprintf(stderr, "[continuity] parse %.2f ms, assemble %.2f ms (%s), solve %.2f ms (device)\n", t0 - tEntry, st.assemble_ms,
                         onDevice ? "device" : "host", st.solve_ms);
        if (stats) *stats = st;
        return HPSDF_OK;
    }
    // a CG region over a few hundred thousand non-zeros lasts tens of microseconds: one thread per ~0.25 M non-zeros
    // (a per-leaf dense block preconditioner was tried: 106 -> 88 iterations on sphere@1e-8, not worth its triangular solves)
    Pool pool((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nThreads, st.nnz / 250000)));
    const uint64_t n = b.nCoeffs;
    const double lambda = b.cfg.continuity_strength;
    for (auto& v : keep.v) v.assign(n, 0.0);
    std::vector<double>&rhs = keep.v[0], &x = keep.v[1], &r = keep.v[2], &p = keep.v[3], &z = keep.v[4], &tmp = keep.v[5], &dinv = keep.v[6],
                       &c = keep.v[7];
    std::memcpy(c.data(), b.coeffs, sizeof(double) * n);  // the block's doubles are 8-byte aligned after the count
    Vec V(pool, n);
    V.each([&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            rhs[i] = c[i] * lambda;  // :1738-1741
            x[i] = rhs[i];           // solveWithGuess(oldCoeffs, oldCoeffs), :1755
            double d = lambda;       // :1724-1729 regularisation on the diagonal
            for (uint64_t q = M.rowPtr[i]; q < M.rowPtr[i + 1]; ++q)
                if (M.col[q] == i) d += M.val[q];
            dinv[i] = 1.0 / d;
        }
    });
    V.spmv(M, 0.0, c.data(), tmp.data());
    st.jump_before = V.dot(c.data(), tmp.data());
    if (maxIter <= 0) maxIter = (int)std::min<uint64_t>(2 * n, 0x7FFFFFFF);  // Eigen's default 2n
    // Eigen's conjugate_gradient(): residual / threshold / update order
    V.spmv(M, lambda, x.data(), tmp.data());
    V.each([&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) r[i] = rhs[i] - tmp[i];
    });
    const double rhsNorm2 = V.dot(rhs.data(), rhs.data());
    double resNorm2 = V.dot(r.data(), r.data());
    int it = 0;
    if (rhsNorm2 == 0.0) {
        std::fill(x.begin(), x.end(), 0.0);
        resNorm2 = 0.0;
    } else {
        const double threshold = std::max(tol * tol * rhsNorm2, std::numeric_limits<double>::min());
        if (!(resNorm2 < threshold)) {
            V.each([&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; ++i) p[i] = dinv[i] * r[i];
            });
            double absNew = V.dot(r.data(), p.data());
            std::vector<double> part2(V.nChunks ? V.nChunks : 1);
            while (it < maxIter) {
                // region 1: tmp = A p and the partial sums of p . tmp
                pool.forEach(V.nChunks, [&](uint64_t c) {
                    double prod[kVecChunk];
                    const uint64_t lo = c * kVecChunk, cnt = std::min(n, lo + kVecChunk) - lo;
                    for (uint64_t i = lo; i < lo + cnt; ++i) {
                        double sacc = lambda * p[i];
                        for (uint64_t q = M.rowPtr[i]; q < M.rowPtr[i + 1]; ++q) sacc += M.val[q] * p[M.col[q]];
                        tmp[i] = sacc;
                        prod[i - lo] = p[i] * sacc;
                    }
                    V.partial[c] = cgChunkSum(prod, cnt);
                });
                const double pAp = cgCombine(V.partial.data(), V.nChunks);
                const double alpha = absNew / pAp;
                // region 2: x, r, z updates and the partial sums of r . r and r . z
                pool.forEach(V.nChunks, [&](uint64_t c) {
                    double rr[kVecChunk], rz[kVecChunk];
                    const uint64_t lo = c * kVecChunk, cnt = std::min(n, lo + kVecChunk) - lo;
                    for (uint64_t i = lo; i < lo + cnt; ++i) {
                        x[i] += alpha * p[i];
                        r[i] -= alpha * tmp[i];
                        z[i] = dinv[i] * r[i];
                        rr[i - lo] = r[i] * r[i];
                        rz[i - lo] = r[i] * z[i];
                    }
                    V.partial[c] = cgChunkSum(rr, cnt);
                    part2[c] = cgChunkSum(rz, cnt);
                });
                resNorm2 = cgCombine(V.partial.data(), V.nChunks);
                const double rz = cgCombine(part2.data(), V.nChunks);
                if (resNorm2 < threshold) break;
                const double absOld = absNew;
                absNew = rz;
                const double beta = absNew / absOld;
                // region 3
                V.each([&](uint64_t lo, uint64_t hi) {
                    for (uint64_t i = lo; i < hi; ++i) p[i] = z[i] + beta * p[i];
                });
                ++it;
            }
        }
    }
    V.spmv(M, 0.0, x.data(), tmp.data());
    st.jump_after = V.dot(x.data(), tmp.data());
    st.iterations = (uint64_t)it;
    st.residual = rhsNorm2 > 0.0 ? std::sqrt(resNorm2 / rhsNorm2) : 0.0;
    std::memcpy(b.coeffs, x.data(), sizeof(double) * n);  // :1756
    st.assemble_ms = t1 - t0;
    st.solve_ms = nowMs() - t1;
    if (std::getenv("HPSDF_TRACE"))
        std::fprintf(stderr, "[continuity] parse %.2f ms, assemble %.2f ms, solve %.2f ms (%s), pool %u threads\n", t0 - tEntry,
                     st.assemble_ms, st.solve_ms, ctx ? "device" : "host", pool.size());
    if (stats) *stats = st;
    return HPSDF_OK;
}

}  // namespace hpsdf
