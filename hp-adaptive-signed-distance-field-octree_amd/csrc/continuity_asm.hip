// Continuity post-process, the ASSEMBLY on the device: what continuity.cpp's nodeProc / faceProc / Assembler do on the host
// (Octree::NodeProc / FaceProc, Octree.cpp:1549-1612; EvaluateSharedFaceIntegralAnalytically :1459-1546 and
// ...Numerically :1250-1456), with the same arithmetic statement by statement, so the CSR arrays it leaves in HBM are the
// host assembly's bit for bit (tests/test_gpu_parity.py compares them) -- and the solve (cg.hip) starts from them without
// the 20 MB upload the host-assembled matrix needed.
//
// The host lists the face pairs by one recursive traversal and gives every leaf its incident pairs in that order (the
// order in which duplicates are summed into the leaf's own block).  Restricted to one leaf L the traversal's order has a
// closed form, so every leaf finds its own incidents independently:
//   * pairs (L, O) are discovered by the faceProc calls of the lowest common ancestor A of L and O; nodeProc visits the
//     children before it runs its own twelve faceProcs, so deeper ancestors come first;
//   * of A's faceProcs (dim 0, 1, 2; four per dim) exactly one per dim involves the child of A that holds L;
//   * inside a faceProc both sides descend together through the children on the shared face, sub-call i pairing the
//     children with transverse index i -- so L's own path fixes the other side's path until one of them is a leaf, and if
//     L ends first the other side's leaves on the face follow depth first, i = 0..3.
// One workgroup per leaf then builds that leaf's rows: the own block accumulated in HBM scratch in incident order, cross
// blocks recomputed where they are written (each is a pure function of the pair), rows counted, scanned, written with
// ascending columns (blocks ordered by the neighbours' coefficient offsets).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "continuity.hpp"
#include "device_types.hpp"
#include "launch.hpp"
#include "runtime.hpp"
#include "tables.hpp"

namespace hpsdf {

namespace {

constexpr uint64_t kLeaf = ~0ull;
constexpr int kP = kMaxDegree + 1;  // 13 orders
constexpr float kEps = 0.000001f;   // Include/Utility/Literals.h:14
constexpr int kAsmThreads = 256;
constexpr uint32_t kAsmMaxIncidents = 1024;  // per leaf (LDS order table); more: the host assembles

struct AsmIncident {
    uint32_t other;
    uint8_t dim, side, numeric, pad;  // side 0: the leaf is the pair's a (its +dim face), 1: it is b
};

struct AsmDev {
    const hpsdf_node* nodes;
    uint32_t nNodes;
    const DeviceTables* T;
    uint32_t* parent;    // [nNodes] parent * 8 + slot, root: 0xFFFFFFFF
    uint32_t* leaves;    // [nLeaves]
    uint32_t* counters;  // [0] leaves listed, [1] fallback requested, [2] pairs x 2, [3] numeric pairs x 2, [4] longest row
    uint64_t* incCount;  // [nLeaves + 1] -> exclusive scan in place
    uint64_t* ownCount;  // [nLeaves + 1] rows^2 per leaf -> exclusive scan in place
    AsmIncident* inc;
    double* own;
    uint8_t* ownSet;
    uint64_t* rowLen;  // [n + 1]: lengths, then (shifted by one) the row pointer
    uint32_t* col;
    double* val;
};

__device__ __forceinline__ uint32_t childOnFace(int dim, uint32_t t, uint32_t side) {  // FaceLookup, Utility.h:166-196
    return dim == 0 ? ((t << 1) | side) : dim == 1 ? ((t & 1u) | (side << 1) | ((t >> 1) << 2)) : (t | (side << 2));
}
__device__ __forceinline__ uint32_t transverseIndex(int dim, uint32_t slot) {
    return dim == 0 ? (slot >> 1) : dim == 1 ? ((slot & 1u) | ((slot >> 2) << 1)) : (slot & 3u);
}

// Octree::LpX, :988-1004
__device__ __forceinline__ double lpxDev(const DeviceTables* T, unsigned p, double x) {
    double m2 = 0.0, m1 = 1.0;
    for (unsigned i = 1; i <= p; ++i) {
        const double l = T->rec[i][0] * x * m1 - T->rec[i][1] * m2;
        m2 = m1, m1 = l;
    }
    return m1;
}

__global__ __launch_bounds__(256) void ca_parent_kernel(AsmDev d) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= d.nNodes) return;
    if (i == 0) d.parent[0] = 0xFFFFFFFFu;
    const hpsdf_node& n = d.nodes[i];
    if (n.child_idx == kLeaf) {
        d.leaves[atomicAdd(&d.counters[0], 1u)] = i;
    } else {
        for (uint32_t c = 0; c < 8; ++c) d.parent[(uint32_t)n.child_idx + c] = i * 8u + c;
    }
}

// the incidents of leaf L in the order the host's traversal gives them (see the head of this file)
template <typename F>
__device__ __forceinline__ void forEachIncident(const AsmDev& d, uint32_t L, F&& emit) {
    uint32_t anc[16];
    uint8_t slot[16];
    int D = 0;
    for (uint32_t cur = L; d.parent[cur] != 0xFFFFFFFFu; cur = d.parent[cur] >> 3) ++D;
    if (D > 12) return;  // (block_check.hpp bounds the depth at 11)
    {
        uint32_t cur = L;
        for (int l = D; l >= 1; --l) {
            anc[l] = cur;
            const uint32_t p = d.parent[cur];
            slot[l] = (uint8_t)(p & 7u);
            cur = p >> 3;
        }
        anc[0] = cur;
    }
    for (int a = D - 1; a >= 0; --a) {
        const uint32_t s = slot[a + 1];
        for (int dim = 0; dim < 3; ++dim) {
            const uint32_t side = (s >> dim) & 1u;
            bool touches = true;
            for (int l = a + 2; l <= D; ++l) touches = touches && (((uint32_t)(slot[l] >> dim) & 1u) == (side ^ 1u));
            if (!touches) continue;
            uint32_t Y = (uint32_t)d.nodes[anc[a]].child_idx + (s ^ (1u << dim));
            int l = a + 2;
            while (l <= D && d.nodes[Y].child_idx != kLeaf) {
                Y = (uint32_t)d.nodes[Y].child_idx + childOnFace(dim, transverseIndex(dim, slot[l]), side);
                ++l;
            }
            if (d.nodes[Y].child_idx == kLeaf) {
                emit(Y, dim, side);
                continue;
            }
            uint32_t stN[14];
            uint8_t stT[14];
            int sp = 0;
            stN[0] = Y, stT[0] = 0;
            while (sp >= 0) {
                if (stT[sp] == 4) {
                    --sp;
                    continue;
                }
                const uint32_t c = (uint32_t)d.nodes[stN[sp]].child_idx + childOnFace(dim, stT[sp]++, side);
                if (d.nodes[c].child_idx == kLeaf) {
                    emit(c, dim, side);
                } else if (sp < 13) {
                    ++sp;
                    stN[sp] = c, stT[sp] = 0;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void ca_count_incidents_kernel(AsmDev d, uint32_t nLeaves) {
    const uint32_t li = blockIdx.x * 256u + threadIdx.x;
    if (li >= nLeaves) return;
    const uint32_t L = d.leaves[li];
    uint64_t n = 0, numeric = 0;
    const uint8_t depth = d.nodes[L].depth;
    forEachIncident(d, L, [&](uint32_t O, int, uint32_t) {
        ++n;
        numeric += d.nodes[O].depth != depth ? 1u : 0u;
    });
    d.incCount[li] = n;
    const uint64_t nl = d.T->count[d.nodes[L].degree];
    d.ownCount[li] = nl * nl;
    if (n > kAsmMaxIncidents) atomicOr(&d.counters[1], 1u);
    atomicAdd(&d.counters[2], (uint32_t)n);
    atomicAdd(&d.counters[3], (uint32_t)numeric);
}

__global__ __launch_bounds__(256) void ca_fill_incidents_kernel(AsmDev d, uint32_t nLeaves) {
    const uint32_t li = blockIdx.x * 256u + threadIdx.x;
    if (li >= nLeaves) return;
    const uint32_t L = d.leaves[li];
    uint64_t at = d.incCount[li];
    const uint8_t depth = d.nodes[L].depth;
    forEachIncident(d, L, [&](uint32_t O, int dim, uint32_t side) {
        d.inc[at++] = AsmIncident{O, (uint8_t)dim, (uint8_t)side, (uint8_t)(d.nodes[O].depth != depth ? 1 : 0), 0};
    });
}

// exclusive scan of a[0 .. n] in place (a[n] receives the total), one workgroup
__global__ __launch_bounds__(1024) void ca_scan_kernel(uint64_t* a, uint32_t n) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base <= n; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n ? a[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024u; off <<= 1) {
            const uint64_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i <= n) a[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
}

// One-dimensional quadrature tables of a non-conforming face (continuity.cpp prepareNumericFace, :1264-1314) in LDS, and
// the 1-D quadratures of this leaf's orders against its own (ILL) or the other leaf's (ILO).
struct NumericLds {
    double T[2][2][kP][kP];  // [side][axis 0 = m1, 1 = m2][order][sample]
    double w[kP];
    double I[2][kP][kP];
    double scale12;
    unsigned n;
};

__device__ void numericTables(const AsmDev& d, const hpsdf_node& nA, const hpsdf_node& nB, unsigned dim, NumericLds& f) {
    __shared__ double sInvT[2], sInvDist;
    const unsigned m1 = (dim + 1) % 3, m2 = (dim + 2) % 3;
    if (threadIdx.x == 0) {
        double scale[3];
        for (int a = 0; a < 3; ++a) {
            const float lo = nA.aabb_min[a] > nB.aabb_min[a] ? nA.aabb_min[a] : nB.aabb_min[a];
            const float hi = nA.aabb_max[a] < nB.aabb_max[a] ? nA.aabb_max[a] : nB.aabb_max[a];
            scale[a] = (double)(hi - lo) * 0.5;
        }
        f.scale12 = scale[m1] * scale[m2];
        const unsigned maxDegree = nA.degree > nB.degree ? nA.degree : nB.degree;
        f.n = maxDegree + 1;
        const unsigned depthDiff = nA.depth > nB.depth ? nA.depth - nB.depth : nB.depth - nA.depth;
        const double invDist = 1.0 / (double)(1ull << depthDiff);  // 1 / pow(2, depthDiff), :1275
        sInvDist = invDist;
        const hpsdf_node& s = nA.depth > nB.depth ? nA : nB;
        const hpsdf_node& l = nA.depth > nB.depth ? nB : nA;
        const unsigned ms[2] = {m1, m2};
        for (int q = 0; q < 2; ++q) {
            const unsigned m = ms[q];
            const float cs = (s.aabb_min[m] + s.aabb_max[m]) / 2.0f, cl = (l.aabb_min[m] + l.aabb_max[m]) / 2.0f;
            sInvT[q] = (double)(cs - cl) / ((double)(s.aabb_max[m] - s.aabb_min[m]) * 0.5) * invDist;
        }
    }
    __syncthreads();
    const unsigned maxDegree = nA.degree > nB.degree ? nA.degree : nB.degree;
    const unsigned gqStart = maxDegree * (maxDegree + 1) / 2;  // Tables::sumToN
    const unsigned n = maxDegree + 1;
    for (unsigned e = threadIdx.x; e < 2u * kP * n; e += kAsmThreads) {
        const unsigned ax = e / (kP * n), p = (e / n) % kP, q = e % n;
        const double r = d.T->roots[gqStart + q];
        const double ua = nB.depth > nA.depth ? r * sInvDist + sInvT[ax] : r;
        const double ub = nA.depth > nB.depth ? r * sInvDist + sInvT[ax] : r;
        f.T[0][ax][p][q] = lpxDev(d.T, p, ua);
        f.T[1][ax][p][q] = lpxDev(d.T, p, ub);
        if (ax == 0 && p == 0) f.w[q] = d.T->weights[gqStart + q];
    }
    __syncthreads();
}
// I[ax][p][r] = sum_x w_x T[sI][ax][p][x] T[sJ][ax][r][x], p <= degI, r <= degJ
__device__ void numericQuadratures(NumericLds& f, unsigned sI, unsigned sJ, unsigned degI, unsigned degJ) {
    for (unsigned e = threadIdx.x; e < 2u * (degI + 1) * (degJ + 1); e += kAsmThreads) {
        const unsigned ax = e / ((degI + 1) * (degJ + 1)), p = (e / (degJ + 1)) % (degI + 1), r = e % (degJ + 1);
        double acc = 0.0;
        for (unsigned x = 0; x < f.n; ++x) acc += f.w[x] * f.T[sI][ax][p][x] * f.T[sJ][ax][r][x];
        f.I[ax][p][r] = acc;
    }
    __syncthreads();
}

struct LeafLds {
    double faceP[kP], faceM[kP];  // LpX(p, +1), LpX(p, -1)
    NumericLds num;
    uint16_t order[kAsmMaxIncidents + 1];  // blocks in column order: incident index, or nInc for the own block
    uint32_t nBlocks;
};

__device__ __forceinline__ void leafPrologue(const AsmDev& d, LeafLds& S) {
    if (threadIdx.x < kP) {
        S.faceP[threadIdx.x] = lpxDev(d.T, threadIdx.x, 1.0);
        S.faceM[threadIdx.x] = lpxDev(d.T, threadIdx.x, -1.0);
    }
    __syncthreads();
}

// the own block of every leaf, duplicates summed in incident order (Assembler::leafRows, first half)
__global__ __launch_bounds__(kAsmThreads) void ca_own_kernel(AsmDev d) {
    __shared__ LeafLds S;
    const uint32_t li = blockIdx.x, L = d.leaves[li];
    const hpsdf_node nL = d.nodes[L];
    const DeviceTables* T = d.T;
    const unsigned nl = T->count[nL.degree];
    double* own = d.own + d.ownCount[li];
    uint8_t* set = d.ownSet + d.ownCount[li];
    for (uint64_t e = threadIdx.x; e < (uint64_t)nl * nl; e += kAsmThreads) own[e] = 0.0, set[e] = 0;
    leafPrologue(d, S);
    for (uint64_t q = d.incCount[li]; q < d.incCount[li + 1]; ++q) {
        const AsmIncident in = d.inc[q];
        const unsigned dim = in.dim, side = in.side, m1 = (dim + 1) % 3, m2 = (dim + 2) % 3;
        const double* fL = side ? S.faceM : S.faceP;
        if (!in.numeric) {  // :1459-1546
            for (unsigned i = threadIdx.x; i < nl; i += kAsmThreads) {
                const uint8_t* bi = T->bidx[i];
                for (unsigned j = 0; j < nl; ++j) {
                    const uint8_t* bj = T->bidx[j];
                    if (bi[m1] != bj[m1] || bi[m2] != bj[m2]) continue;
                    double integral = 1.0;
                    integral *= fL[bi[dim]];
                    integral *= T->nl[bi[dim]][nL.depth];
                    integral *= fL[bj[dim]];
                    integral *= T->nl[bj[dim]][nL.depth];
                    own[(size_t)i * nl + j] += integral;
                    set[(size_t)i * nl + j] = 1;
                }
            }
        } else {  // :1250-1456
            const hpsdf_node nO = d.nodes[in.other];
            numericTables(d, side ? nO : nL, side ? nL : nO, dim, S.num);
            numericQuadratures(S.num, side, side, nL.degree, nL.degree);
            for (unsigned i = threadIdx.x; i < nl; i += kAsmThreads) {
                const uint8_t* bi = T->bidx[i];
                const double wi = T->nl[bi[0]][nL.depth] * T->nl[bi[1]][nL.depth] * T->nl[bi[2]][nL.depth];
                for (unsigned j = 0; j < nl; ++j) {
                    const uint8_t* bj = T->bidx[j];
                    const double wj = T->nl[bj[0]][nL.depth] * T->nl[bj[1]][nL.depth] * T->nl[bj[2]][nL.depth];
                    const double integral = S.num.I[0][bi[m1]][bj[m1]] * S.num.I[1][bi[m2]][bj[m2]] * (fL[bi[dim]] * fL[bj[dim]]) *
                                            (S.num.scale12 * (wi * wj));
                    if (fabsf((float)integral) > kEps) {  // :1337 / :1448
                        own[(size_t)i * nl + j] += integral;
                        set[(size_t)i * nl + j] = 1;
                    }
                }
            }
            __syncthreads();  // the tables are rebuilt by the next numeric incident
        }
    }
}

// One cross block entry (row i of L against column j of O), or "not an entry".
__device__ __forceinline__ bool crossEntry(const DeviceTables* T, const LeafLds& S, const hpsdf_node& nL, const hpsdf_node& nO,
                                           const AsmIncident& in, unsigned i, unsigned j, double& v) {
    const unsigned dim = in.dim, side = in.side, m1 = (dim + 1) % 3, m2 = (dim + 2) % 3;
    const uint8_t* bi = T->bidx[i];
    const uint8_t* bj = T->bidx[j];
    if (!in.numeric) {
        if (bi[m1] != bj[m1] || bi[m2] != bj[m2]) return false;
        // :1511-1515 multiplies a's factors first, then b's
        const uint8_t* ba = side ? bj : bi;
        const uint8_t* bb = side ? bi : bj;
        const uint8_t da = side ? nO.depth : nL.depth, db = side ? nL.depth : nO.depth;
        double integral = -1.0;
        integral *= S.faceP[ba[dim]];
        integral *= T->nl[ba[dim]][da];
        integral *= S.faceM[bb[dim]];
        integral *= T->nl[bb[dim]][db];
        v = integral;
        return true;
    }
    const double* fL = side ? S.faceM : S.faceP;
    const double* fO = side ? S.faceP : S.faceM;
    const double wi = T->nl[bi[0]][nL.depth] * T->nl[bi[1]][nL.depth] * T->nl[bi[2]][nL.depth];
    const double wj = T->nl[bj[0]][nO.depth] * T->nl[bj[1]][nO.depth] * T->nl[bj[2]][nO.depth];
    const double integral = S.num.I[0][bi[m1]][bj[m1]] * S.num.I[1][bi[m2]][bj[m2]] * (fL[bi[dim]] * fO[bj[dim]]) *
                            (S.num.scale12 * (wi * wj)) * -1.0;
    v = integral;
    return fabsf((float)integral) > kEps;  // :1391
}

// WRITE = false: row lengths; WRITE = true: columns and values at the rows' places (d.rowLen is the row pointer by then)
template <bool WRITE>
__global__ __launch_bounds__(kAsmThreads) void ca_rows_kernel(AsmDev d) {
    __shared__ LeafLds S;
    const uint32_t li = blockIdx.x, L = d.leaves[li];
    const hpsdf_node nL = d.nodes[L];
    const DeviceTables* T = d.T;
    const unsigned nl = T->count[nL.degree];
    const double* own = d.own + d.ownCount[li];
    const uint8_t* set = d.ownSet + d.ownCount[li];
    const uint64_t q0 = d.incCount[li];
    const uint32_t nInc = (uint32_t)(d.incCount[li + 1] - q0);
    if (nInc > kAsmMaxIncidents) return;  // (the host assembles this tree; counters[1] is up)
    leafPrologue(d, S);
    if (threadIdx.x == 0) {  // blocks by first column: the neighbours' offsets, the own block among them (insertion sort)
        uint32_t n = 0;
        for (uint32_t k = 0; k <= nInc; ++k) {
            const uint64_t key = k < nInc ? d.nodes[d.inc[q0 + k].other].coeffs_start : nL.coeffs_start;
            uint32_t at = n++;
            while (at > 0) {
                const uint16_t o = S.order[at - 1];
                const uint64_t ok = o < nInc ? d.nodes[d.inc[q0 + o].other].coeffs_start : nL.coeffs_start;
                if (!(key < ok)) break;
                S.order[at] = o;
                --at;
            }
            S.order[at] = (uint16_t)k;
        }
        S.nBlocks = n;
    }
    __syncthreads();
    for (unsigned i0 = 0; i0 < nl; i0 += kAsmThreads) {  // (uniform: the numeric tables are built by the whole workgroup)
        const unsigned i = i0 + threadIdx.x;
        const bool live = i < nl;
        uint64_t len = 0;
        uint64_t at = WRITE && live ? d.rowLen[nL.coeffs_start + i] : 0;
        for (uint32_t b = 0; b < S.nBlocks; ++b) {
            const uint32_t k = S.order[b];
            if (k == nInc) {
                if (live)
                    for (unsigned j = 0; j < nl; ++j)
                        if (set[(size_t)i * nl + j]) {
                            if (WRITE) d.col[at] = (uint32_t)(nL.coeffs_start + j), d.val[at] = own[(size_t)i * nl + j], ++at;
                            ++len;
                        }
                continue;
            }
            const AsmIncident in = d.inc[q0 + k];
            const hpsdf_node nO = d.nodes[in.other];
            const unsigned no = T->count[nO.degree];
            if (in.numeric) {
                numericTables(d, in.side ? nO : nL, in.side ? nL : nO, in.dim, S.num);
                numericQuadratures(S.num, in.side, in.side ^ 1u, nL.degree, nO.degree);
            }
            if (live)
                for (unsigned j = 0; j < no; ++j) {
                    double v;
                    if (crossEntry(T, S, nL, nO, in, i, j, v)) {
                        if (WRITE) d.col[at] = (uint32_t)(nO.coeffs_start + j), d.val[at] = v, ++at;
                        ++len;
                    }
                }
            if (in.numeric) __syncthreads();
        }
        if (!WRITE && live) {
            d.rowLen[nL.coeffs_start + i] = len;
            atomicMax(&d.counters[4], (uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len));
        }
    }
}

}  // namespace

ContinuityDeviceMatrix::~ContinuityDeviceMatrix() {
    if (base) {
        (void)hipSetDevice(device);
        (void)hipFree(base);
    }
    if (entries) {
        (void)hipSetDevice(device);
        (void)hipFree(entries);
    }
    if (scanTmp) {
        (void)hipSetDevice(device);
        (void)hipFree(scanTmp);
    }
}

// Assembles M for the tree `nodes` on ctx's device.  *fallback = 1: this tree is for the host assembler (a leaf with more
// incidents than the order table holds, or own blocks that would not fit the scratch budget).
int continuityAssembleDevice(hpsdf_ctx* ctx, const hpsdf_node* nodes, uint64_t nNodes, uint64_t nCoeffs, ContinuityDeviceMatrix& out,
                             hpsdf_continuity_stats& st, int* fallback, std::string& err) {
    *fallback = 0;
    if (nNodes >= 0x1FFFFFFFull || nCoeffs >= 0xFFFFFFFFull) {
        *fallback = 1;
        return HPSDF_OK;
    }
    const Tables& T = tables();
    uint64_t nLeaves = 0, ownTotal = 0;
    for (uint64_t i = 0; i < nNodes; ++i)
        if (nodes[i].child_idx == kLeaf) {
            ++nLeaves;
            ownTotal += T.coeffCount[nodes[i].degree] * T.coeffCount[nodes[i].degree];
        }
    if (nLeaves == 0 || ownTotal > (1ull << 28)) {  // 2 GB of own blocks: not worth holding on the device
        *fallback = 1;
        return HPSDF_OK;
    }
    auto al = [](uint64_t b) { return (b + 255) & ~255ull; };
    auto fail2 = [&](hipError_t e, const char* what) {
        err = std::string("continuity assembly (") + what + "): " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? HPSDF_ERR_OUT_OF_MEMORY : HPSDF_ERR_HIP;
    };
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail2(e, "device");
    hipStream_t s = ctx->stream;
    // ---- pass A: everything whose size is known up front, plus a generous guess for the incidents (6 faces, some finer)
    const uint64_t incGuess = 8 * nLeaves + 1024;
    auto sizeA = [&](uint64_t incCap) {
        return al(nNodes * sizeof(hpsdf_node)) + al(nNodes * 4) + al(nLeaves * 4) + 256 + 2 * al((nLeaves + 1) * 8) + al(incCap * sizeof(AsmIncident)) +
               al(ownTotal * 8) + al(ownTotal) + al((nCoeffs + 1) * 8);
    };
    uint64_t incCap = incGuess;
    auto ensureBase = [&](uint64_t bytes) -> hipError_t {
        if (out.cap >= bytes && out.device == ctx->device) return hipSuccess;
        if (out.base) {
            (void)hipSetDevice(out.device);
            (void)hipFree(out.base);
            (void)hipSetDevice(ctx->device);
        }
        out.base = nullptr, out.cap = 0, out.device = ctx->device;
        const hipError_t r = hipMalloc((void**)&out.base, bytes + bytes / 4);
        if (r == hipSuccess) out.cap = bytes + bytes / 4;
        return r;
    };
    e = ensureBase(sizeA(incCap));
    if (e != hipSuccess) return fail2(e, "buffers");
    AsmDev d;
    auto carve = [&](uint64_t cap) {
        char* cur = out.base;
        auto take = [&](uint64_t bytes) {
            char* q = cur;
            cur += al(bytes);
            return q;
        };
        d.nodes = (const hpsdf_node*)take(nNodes * sizeof(hpsdf_node));
        d.nNodes = (uint32_t)nNodes;
        d.T = ctx->dTables;
        d.parent = (uint32_t*)take(nNodes * 4);
        d.leaves = (uint32_t*)take(nLeaves * 4);
        d.counters = (uint32_t*)take(256);
        d.incCount = (uint64_t*)take((nLeaves + 1) * 8);
        d.ownCount = (uint64_t*)take((nLeaves + 1) * 8);
        d.inc = (AsmIncident*)take(cap * sizeof(AsmIncident));
        d.own = (double*)take(ownTotal * 8);
        d.ownSet = (uint8_t*)take(ownTotal);
        d.rowLen = (uint64_t*)take((nCoeffs + 1) * 8);
        d.col = nullptr, d.val = nullptr;
    };
    carve(incCap);
    const unsigned gN = (unsigned)((nNodes + 255) / 256), gL = (unsigned)((nLeaves + 255) / 256);
    e = hipMemcpyAsync((void*)d.nodes, nodes, nNodes * sizeof(hpsdf_node), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d.counters, 0, 256, s);
    if (e == hipSuccess) e = hipMemsetAsync(d.rowLen, 0, (nCoeffs + 1) * 8, s);
    if (e != hipSuccess) return fail2(e, "upload");
    hipLaunchKernelGGL(ca_parent_kernel, dim3(gN), dim3(256), 0, s, d);
    hipLaunchKernelGGL(ca_count_incidents_kernel, dim3(gL), dim3(256), 0, s, d, (uint32_t)nLeaves);
    hipLaunchKernelGGL(ca_scan_kernel, dim3(1), dim3(1024), 0, s, d.incCount, (uint32_t)nLeaves);
    hipLaunchKernelGGL(ca_scan_kernel, dim3(1), dim3(1024), 0, s, d.ownCount, (uint32_t)nLeaves);
    uint32_t hc[4];
    uint64_t incTotal = 0;
    e = hipMemcpyAsync(hc, d.counters, sizeof hc, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&incTotal, d.incCount + nLeaves, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail2(e, "incident count");
    if (hc[0] != nLeaves) {
        err = "continuity assembly: leaf list does not match the tree";
        return HPSDF_ERR_STATE;
    }
    if (hc[1]) {
        *fallback = 1;
        return HPSDF_OK;
    }
    if (incTotal > incCap) {  // rare (strongly graded trees): grow and start over -- the first pass is cheap
        incCap = incTotal;
        e = ensureBase(sizeA(incCap));
        if (e != hipSuccess) return fail2(e, "buffers");
        carve(incCap);
        e = hipMemcpyAsync((void*)d.nodes, nodes, nNodes * sizeof(hpsdf_node), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(d.counters, 0, 256, s);
        if (e == hipSuccess) e = hipMemsetAsync(d.rowLen, 0, (nCoeffs + 1) * 8, s);
        if (e != hipSuccess) return fail2(e, "upload");
        hipLaunchKernelGGL(ca_parent_kernel, dim3(gN), dim3(256), 0, s, d);
        hipLaunchKernelGGL(ca_count_incidents_kernel, dim3(gL), dim3(256), 0, s, d, (uint32_t)nLeaves);
        hipLaunchKernelGGL(ca_scan_kernel, dim3(1), dim3(1024), 0, s, d.incCount, (uint32_t)nLeaves);
        hipLaunchKernelGGL(ca_scan_kernel, dim3(1), dim3(1024), 0, s, d.ownCount, (uint32_t)nLeaves);
    }
    hipLaunchKernelGGL(ca_fill_incidents_kernel, dim3(gL), dim3(256), 0, s, d, (uint32_t)nLeaves);
    hipLaunchKernelGGL(ca_own_kernel, dim3((unsigned)nLeaves), dim3(kAsmThreads), 0, s, d);
    hipLaunchKernelGGL(ca_rows_kernel<false>, dim3((unsigned)nLeaves), dim3(kAsmThreads), 0, s, d);
    // row lengths -> row pointer (exclusive scan over n + 1 entries: rowLen[n] = 0 receives the total)
    {
        size_t tmpBytes = 0;
        e = rocprim::exclusive_scan(nullptr, tmpBytes, d.rowLen, d.rowLen, (uint64_t)0, (size_t)(nCoeffs + 1), rocprim::plus<uint64_t>(), s);
        if (e != hipSuccess) return fail2(e, "scan size");
        if (out.scanCap < tmpBytes) {
            if (out.scanTmp) (void)hipFree(out.scanTmp);
            out.scanTmp = nullptr, out.scanCap = 0;
            e = hipMalloc(&out.scanTmp, tmpBytes + 256);
            if (e != hipSuccess) return fail2(e, "scan buffer");
            out.scanCap = tmpBytes + 256;
        }
        e = rocprim::exclusive_scan(out.scanTmp, tmpBytes, d.rowLen, d.rowLen, (uint64_t)0, (size_t)(nCoeffs + 1), rocprim::plus<uint64_t>(), s);
        if (e != hipSuccess) return fail2(e, "scan");
    }
    uint64_t nnz = 0;
    uint32_t longest = 0;
    e = hipMemcpyAsync(&nnz, d.rowLen + nCoeffs, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&longest, d.counters + 4, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail2(e, "row pointer");
    const uint64_t entryBytes = al(nnz * 4 + 4) + al(nnz * 8 + 8);
    if (out.entryCap < entryBytes) {
        if (out.entries) (void)hipFree(out.entries);
        out.entries = nullptr, out.entryCap = 0;
        e = hipMalloc((void**)&out.entries, entryBytes + entryBytes / 4);
        if (e != hipSuccess) return fail2(e, "entries");
        out.entryCap = entryBytes + entryBytes / 4;
    }
    d.col = (uint32_t*)out.entries;
    d.val = (double*)(out.entries + al(nnz * 4 + 4));
    hipLaunchKernelGGL(ca_rows_kernel<true>, dim3((unsigned)nLeaves), dim3(kAsmThreads), 0, s, d);
    e = hipGetLastError();
    if (e != hipSuccess) return fail2(e, "kernels");
    out.n = nCoeffs, out.nnz = nnz, out.maxRow = longest;
    out.dRowPtr = d.rowLen, out.dCol = d.col, out.dVal = d.val;
    st.n_pairs = hc[2] / 2;
    st.n_pairs_numeric = hc[3] / 2;
    st.n_pairs_analytic = st.n_pairs - st.n_pairs_numeric;
    st.nnz = nnz;
    return HPSDF_OK;
}

}  // namespace hpsdf
