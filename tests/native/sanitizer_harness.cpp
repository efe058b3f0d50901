// Sanitizer harness (tests/test_sanitizers.py): the host-side native code of libhpsdf.so -- continuity post-process,
// OBJ reader, mesh preparation, round scheduler -- built with g++ -fsanitize=address,undefined and run without a GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <vector>
#include "block_check.hpp"
#include "builder.hpp"
#include "continuity.hpp"
#include "launch.hpp"
#include "runtime.hpp"
namespace hpsdf {
int loadObj(const char* path, std::vector<float>& verts, std::vector<uint64_t>& tris, std::string& err);
void setError(const std::string&) {}
int fail(int code, const std::string& msg) {
    fprintf(stderr, "fail(%d): %s\n", code, msg.c_str());
    return code;
}
int hipFail(hipError_t, const char*) { return HPSDF_ERR_HIP; }
const hpsdf_field* innermost(const hpsdf_field* f) { return f; }
int makeFieldDev(const hpsdf_ctx*, const hpsdf_field*, const double*, FieldDev*) { return HPSDF_ERR_UNSUPPORTED; }
// the GPU legs are not exercised here (no device): link-time stand-ins for the launch wrappers of kernels.hip
size_t fitLdsBytes(int, int, int) { return 0; }
FitShape fitShape(int, int, uint32_t, bool, bool) { return FitShape{1, 1, 1, 0}; }
hipError_t launchFit(hipStream_t, int, int, const FitBlock*, uint32_t, size_t, const FitTask*, double*, double*, double*,
                     const DeviceTables*, const FieldDev&, const RootMap&, const uint32_t*) { return hipErrorNoDevice; }
hipError_t launchFitMulti(hipStream_t, const FitBlock*, uint32_t, size_t, const FitTask*, double*, double*, const DeviceTables*, const FieldDev&,
                          const RootMap&, const uint32_t*) { return hipErrorNoDevice; }
hipError_t launchPack(hipStream_t, const PackItem*, uint32_t, const double*, double*) { return hipErrorNoDevice; }
hipError_t launchFitWeight(hipStream_t, const FitBlock*, uint32_t, size_t, const FitTask*, const double*, double*, const DeviceTables*, const uint32_t*) { return hipErrorNoDevice; }
hipError_t launchFitMfma(hipStream_t, int, const FitBlock*, uint32_t, const FitTask*, double*, double*, const DeviceTables*, const FieldDev&,
                         const RootMap&, const uint32_t*) { return hipErrorNoDevice; }
bool fitSplitSupports(int, int) { return false; }
static int gLeft = 0;  // capi.cpp's hpsdf_set_reduction_order(); here from HPSDF_REDUCTION_ORDER (main)
int reductionLeftAssoc(const hpsdf_ctx*) { return gLeft; }
int meshFaceRuleReference(const hpsdf_ctx*) { return 0; }
void setReductionLeftAssoc(int left) { gLeft = left != 0; }
int checkBuildLimits(const hpsdf_ctx*, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t*, uint64_t, double, double) { return HPSDF_OK; }
hipError_t launchFitMfmaLow(hipStream_t, int, const FitTask*, const uint32_t*, uint32_t, uint32_t, uint32_t, double*, const DeviceTables*, const double*,
                            const RootMap&, int) { return hipErrorNoDevice; }
hipError_t launchCgIterations(hipStream_t, const CgDev&, int, int, uint32_t) { return hipErrorNoDevice; }
hipError_t launchCgStart(hipStream_t, const CgDev&) { return hipErrorNoDevice; }
hipError_t launchCgFinish(hipStream_t, const CgDev&) { return hipErrorNoDevice; }
ContinuityDeviceMatrix::~ContinuityDeviceMatrix() {}
int continuityAssembleDevice(hpsdf_ctx*, const hpsdf_node*, uint64_t, uint64_t, ContinuityDeviceMatrix&, hpsdf_continuity_stats&, int* fallback,
                             std::string&) {
    *fallback = 1;
    return HPSDF_OK;
}
hipError_t launchMeshTriPos(hipStream_t, const float*, const uint32_t*, uint64_t, float*, const uint32_t*, float*) { return hipErrorNoDevice; }
hipError_t launchMeshSample(hipStream_t, const FitTask*, uint32_t, int, const DeviceTables*, const FieldDev&, const RootMap&, double*) {
    return hipErrorNoDevice;
}
}

// The host round scheduler (builder.cpp) driven through its injection hook: two simulated ranks, synthetic errors that
// make both P- and H-refinements occur, random coefficients; the two ranks must assemble identical blocks.
static int schedulerRun() {
    using namespace hpsdf;
    const Tables& T = tables();
    hpsdf_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.target_error_threshold = 1e-3;
    cfg.thread_count = 1;
    for (int a = 0; a < 3; ++a) cfg.root_min[a] = -0.5f, cfg.root_max[a] = 0.5f;
    std::vector<std::vector<char>> blocks;
    const int world = 2;
    hpsdf_build* b[world];
    for (int r = 0; r < world; ++r) {
        b[r] = new hpsdf_build();
        hpsdf_build_opts o{256, r, world, {0, 0}};
        if (builderBegin(b[r], &cfg, &o)) return 20;
    }
    uint64_t rng = 12345;
    auto next = [&] { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) * 0x1p-53; };
    for (int round = 0; round < 12; ++round) {
        uint64_t n0 = 0, n1 = 0;
        if (builderSelect(b[0], &n0) || builderSelect(b[1], &n1) || n0 != n1) return 21;
        if (n0 == 0) break;
        std::vector<hpsdf_job> jobs(n0);
        builderJobs(b[0], jobs.data());
        std::vector<double> headers(n0 * HPSDF_JOB_HEADER_DOUBLES);
        std::vector<double> pc(kMaxCoeffs), hc(8 * kMaxCoeffs);
        for (auto& v : pc) v = next();
        for (auto& v : hc) v = next();
        for (uint64_t j = 0; j < n0; ++j) {
            const double e = jobs[j].coarse ? 1.0 : jobs[j].err;
            const double mode = next();  // < 0.5: P wins, < 0.9: H wins, else neither improves
            headers[9 * j] = mode < 0.5 ? e * 0.013 : e * 7.3;  // never exactly 100: that value marks a coarse job (Octree.cpp:806)
            for (int c = 0; c < 8; ++c) headers[9 * j + 1 + c] = mode >= 0.5 && mode < 0.9 ? e * 0.0017 : e * 7.3;
            for (int r = 0; r < world; ++r) {
                const hpsdf_build::Slice sl = b[r]->slices[r];
                if (j >= sl.first && j < sl.first + sl.count && builderInject(b[r], j, pc.data(), hc.data())) return 22;
            }
        }
        if (builderApply(b[0], headers.data()) || builderApply(b[1], headers.data())) return 23;
    }
    for (int r = 0; r < world; ++r)
        if (builderLayout(b[r])) {
            for (size_t n = 0; n < b[r]->nodes.size(); ++n) {
                if (b[r]->nodes[n].child_idx != ~0ull) continue;
                uint32_t row = 0;
                bool bad = false;
                for (int64_t sg = b[r]->segHead[n]; sg >= 0; sg = b[r]->segs[sg].next) {
                    if (b[r]->segs[sg].rowStart != row) bad = true;
                    row = b[r]->segs[sg].rowEnd;
                }
                if (bad) {
                    fprintf(stderr, "rank %d node %zu degree %d depth %d segs:", r, n, b[r]->nodes[n].degree, b[r]->nodes[n].depth);
                    for (int64_t sg = b[r]->segHead[n]; sg >= 0; sg = b[r]->segs[sg].next)
                        fprintf(stderr, " [%u,%u) owner %d", b[r]->segs[sg].rowStart, b[r]->segs[sg].rowEnd, b[r]->segs[sg].owner);
                    fprintf(stderr, "\n");
                    break;
                }
            }
            return 24;
        }
    std::vector<std::vector<double>> packs(world);
    for (int r = 0; r < world; ++r) {
        packs[r].resize(std::max<uint64_t>(1, b[r]->packCounts[r]));
        if (builderPackHost(b[r], nullptr, packs[r].data())) return 25;
    }
    const double* pp[world] = {packs[0].data(), packs[1].data()};
    for (int r = 0; r < world; ++r) {
        void* blk = nullptr;
        size_t size = 0;
        if (builderAssemble(b[r], pp, &blk, &size)) return 26;
        blocks.emplace_back((char*)blk, (char*)blk + size);
        std::free(blk);
    }
    printf("scheduler: %llu nodes, %llu coefficients, p %llu h %llu dropped %llu\n", (unsigned long long)b[0]->stats.n_nodes,
           (unsigned long long)b[0]->nCoeffsTotal, (unsigned long long)b[0]->stats.p_refines,
           (unsigned long long)b[0]->stats.h_refines, (unsigned long long)b[0]->stats.dropped);
    const bool exercised = b[0]->stats.p_refines > 1000 && b[0]->stats.h_refines > 100;
    for (int r = 0; r < world; ++r) delete b[r];
    (void)T;
    if (blocks[0] != blocks[1]) return 27;
    return exercised ? 0 : 28;
}
int main(int argc, char** argv) {
    const bool quick = getenv("HPSDF_HARNESS_QUICK") != nullptr;  // the ThreadSanitizer run: the threaded parts in full, the single-threaded fuzz loops cut short
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blk(sz); if (fread(blk.data(), 1, sz, f) != (size_t)sz) return 2; fclose(f);
    std::string err;
    for (int thr : {1, 3, 8}) {
        std::vector<char> b = blk;
        hpsdf_continuity_stats st;
        int rc = hpsdf::continuityPostProcess(b.data(), b.size(), 0.0, 0, thr, &st, err);
        printf("continuity threads %d rc %d its %llu\n", thr, rc, (unsigned long long)st.iterations);
        if (rc) return 3;
    }
    { std::vector<char> b(blk.begin(), blk.end() - 8); hpsdf_continuity_stats st; if (!hpsdf::continuityPostProcess(b.data(), b.size(), 0, 0, 1, &st, err)) return 4; }
    // malformed blocks (untrusted bytes): each must come back as an error, under ASAN/UBSAN, not crash
    {
        auto mutate = [&](const char* what, const std::function<void(std::vector<char>&, uint64_t, uint64_t, hpsdf_node*)>& edit) -> int {
            std::vector<char> b = blk;
            uint64_t nc, nn;
            memcpy(&nc, b.data(), 8);
            memcpy(&nn, b.data() + 8 + 8 * nc, 8);
            edit(b, nc, nn, (hpsdf_node*)(b.data() + 16 + 8 * nc));
            hpsdf_continuity_stats st;
            const int rc = hpsdf::continuityPostProcess(b.data(), b.size(), 0, 0, 2, &st, err);
            printf("malformed block (%s): rc %d (%s)\n", what, rc, err.c_str());
            return rc == HPSDF_ERR_BAD_BLOCK ? 0 : 1;
        };
        int bad = 0;
        bad += mutate("child index 1<<40", [](std::vector<char>&, uint64_t, uint64_t, hpsdf_node* n) { n[0].child_idx = 1ull << 40; });
        bad += mutate("child index wraps", [](std::vector<char>&, uint64_t, uint64_t nn, hpsdf_node* n) { n[0].child_idx = nn - 3; });
        bad += mutate("child points at the root", [](std::vector<char>&, uint64_t, uint64_t, hpsdf_node* n) { n[1].child_idx = 0; });
        bad += mutate("child points at an ancestor", [](std::vector<char>&, uint64_t, uint64_t, hpsdf_node* n) { n[n[1].child_idx].child_idx = 1; });
        bad += mutate("coefficient start wraps", [](std::vector<char>&, uint64_t, uint64_t nn, hpsdf_node* n) {
            for (uint64_t i = 0; i < nn; ++i) if (n[i].child_idx == ~0ull) { n[i].coeffs_start = ~0ull - 3; break; } });
        bad += mutate("coefficient start past the store", [](std::vector<char>&, uint64_t nc, uint64_t nn, hpsdf_node* n) {
            for (uint64_t i = 0; i < nn; ++i) if (n[i].child_idx == ~0ull) { n[i].coeffs_start = nc - 1; break; } });
        bad += mutate("two leaves share coefficients", [](std::vector<char>&, uint64_t, uint64_t nn, hpsdf_node* n) {
            uint64_t first = ~0ull;
            for (uint64_t i = 0; i < nn; ++i) if (n[i].child_idx == ~0ull) { if (first == ~0ull) first = i; else { n[i].coeffs_start = n[first].coeffs_start; break; } } });
        bad += mutate("leaf degree 200", [](std::vector<char>&, uint64_t, uint64_t nn, hpsdf_node* n) {
            for (uint64_t i = 0; i < nn; ++i) if (n[i].child_idx == ~0ull) { n[i].degree = 200; break; } });
        bad += mutate("leaf depth lies", [](std::vector<char>&, uint64_t, uint64_t nn, hpsdf_node* n) {
            for (uint64_t i = 0; i < nn; ++i) if (n[i].child_idx == ~0ull) { n[i].depth = 1; break; } });
        if (bad) return 40;
        {
            // random corruption of the node array (1 to 4 bytes anywhere in it, or a field set to an awkward value): the validator
            // both deserialisers share must say yes or no without leaving the array -- 6 000 blocks, both strictness levels, under the sanitizers
            uint64_t nc, nn;
            memcpy(&nc, blk.data(), 8);
            memcpy(&nn, blk.data() + 8 + 8 * nc, 8);
            std::vector<hpsdf_node> nodes(nn), work;
            memcpy(nodes.data(), blk.data() + 16 + 8 * nc, nn * sizeof(hpsdf_node));
            unsigned long long x = 0x9E3779B97F4A7C15ull;
            auto rnd = [&]() { x ^= x << 13, x ^= x >> 7, x ^= x << 17; return x; };
            const uint64_t awkward[] = {0ull, 1ull, 8ull, nn - 1, nn, nn + 1, ~0ull, ~0ull - 7, 1ull << 32, 1ull << 63, nc, nc - 1, nc + 1};
            int stillValid = 0;
            for (int it = 0; it < (quick ? 200 : 6000); ++it) {
                work = nodes;
                for (int k = 0, nk = 1 + (int)(rnd() % 4); k < nk; ++k) {
                    hpsdf_node& n = work[rnd() % nn];
                    switch (rnd() % 5) {
                        case 0: n.child_idx = awkward[rnd() % (sizeof awkward / sizeof *awkward)]; break;
                        case 1: n.coeffs_start = awkward[rnd() % (sizeof awkward / sizeof *awkward)]; break;
                        case 2: n.degree = (uint8_t)(rnd() % 256); break;
                        case 3: n.depth = (uint8_t)(rnd() % 256); break;
                        default: ((unsigned char*)&n)[rnd() % sizeof n] = (unsigned char)(rnd() % 256); break;
                    }
                }
                hpsdf::BlockTreeInfo info;
                for (int strict = 0; strict < 2; ++strict)
                    if (hpsdf::checkBlockTree(work.data(), nn, nc, hpsdf::tables().coeffCount, strict != 0, true, &info, err) == 0) ++stillValid;
            }
            printf("block fuzz: 6000 corrupted node arrays, %d verdicts 'valid'\n", stillValid);
        }
        // one interior root and nothing else: nNodes < 9
        std::vector<char> tiny(16 + sizeof(hpsdf_node) + sizeof(hpsdf_config), 0);
        const uint64_t one = 1;
        memcpy(tiny.data() + 8, &one, 8);
        hpsdf_node* root = (hpsdf_node*)(tiny.data() + 16);
        root->child_idx = 1ull << 40;
        root->degree = 13;
        hpsdf_continuity_stats st;
        if (hpsdf::continuityPostProcess(tiny.data(), tiny.size(), 0, 0, 1, &st, err) != HPSDF_ERR_BAD_BLOCK) return 41;
        printf("malformed block (1-node interior root): %s\n", err.c_str());
    }
    std::vector<float> v; std::vector<uint64_t> t;
    int rc = hpsdf::loadObj(argv[2], v, t, err);
    printf("obj rc %d verts %zu tris %zu\n", rc, v.size() / 3, t.size() / 3);
    if (rc) return 5;
    hpsdf::HostMesh hm;
    bool ok = hpsdf::prepareMesh(v.data(), v.size() / 3, t.data(), t.size() / 3, &hm);
    printf("prepareMesh closed=%d bvh nodes %zu\n", (int)ok, hm.bvh.size());
    if (!ok) return 6;
    t.resize(t.size() - 3);
    if (hpsdf::prepareMesh(v.data(), v.size() / 3, t.data(), t.size() / 3, &hm)) return 7;  // open mesh must be rejected
    for (const char* bad : {"v 1 2\nf 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4\n", ""}) {
        FILE* g = fopen(argv[3], "wb"); fputs(bad, g); fclose(g);
        if (hpsdf::loadObj(argv[3], v, t, err) == 0) { printf("malformed OBJ accepted\n"); return 8; }
    }
    {
        // a file with a NUL in the middle of a line, and 3000 random corruptions of a small valid file (a byte changed, a run
        // deleted, a token from a list of awkward ones inserted): any status will do, the reader must only stay inside its buffers
        const char nul[] = "v 0 0 0\nv 1 0\0 0\nv 0 1 0\nf 1 2 3\n";
        FILE* g = fopen(argv[3], "wb"); fwrite(nul, 1, sizeof nul - 1, g); fclose(g);
        (void)hpsdf::loadObj(argv[3], v, t, err);
        const std::string good = "# tetrahedron\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvt 0.5 0.5\nf 1 3 2\nf 1//1 2//1 4//1\nf 2/1/1 3/1/1 4/1/1\nf -4 -1 -2\n";
        const char* tokens[] = {"-1", "0", "99999999999999999999", "-99999999999999999999", "1e999", "nan", "inf", "//", "/", "f", "v", "\r", "\t", "1/2/3/4",
                                "+", "-", ".", "e", "0x10", "f 1 2", "v 1", "\n\n", " "};
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13, x ^= x >> 7, x ^= x << 17; return x; };
        int accepted = 0;
        for (int it = 0; it < (quick ? 100 : 3000); ++it) {
            std::string m = good;
            for (int k = 0, nk = 1 + (int)(rnd() % 3); k < nk && !m.empty(); ++k) {
                const size_t at = rnd() % m.size();
                switch (rnd() % 3) {
                    case 0: m[at] = (char)(rnd() % 256); break;
                    case 1: m.erase(at, 1 + rnd() % 8); break;
                    default: m.insert(at, tokens[rnd() % (sizeof tokens / sizeof *tokens)]); break;
                }
            }
            g = fopen(argv[3], "wb"); fwrite(m.data(), 1, m.size(), g); fclose(g);
            if (hpsdf::loadObj(argv[3], v, t, err) == 0) {
                ++accepted;
                for (uint64_t i : t) if (i >= v.size() / 3) { printf("accepted OBJ with an index beyond its vertices\n"); return 9; }
            }
        }
        printf("obj fuzz: 3000 corrupted files, %d still parse\n", accepted);
    }
    if (int rc = schedulerRun()) { printf("scheduler rc %d\n", rc); return rc; }
    if (argc > 5) {
        // host_query.cpp: scalar-sized Query / QueryWithGradient calls are answered on the calling thread from the tree handle's copy of
        // the block.  Here, under the sanitizers and without a GPU: a handle filled the way hpsdf_tree_upload fills it, the points of
        // argv[4] (f64 xyz) in, values and gradients out to argv[5] -- tests/test_sanitizers.py compares them with the oracle bit for bit.
        if (const char* e = getenv("HPSDF_REDUCTION_ORDER")) hpsdf::setReductionLeftAssoc(e[0] == 'l');
        uint64_t nc, nn;
        memcpy(&nc, blk.data(), 8);
        memcpy(&nn, blk.data() + 8 + 8 * nc, 8);
        // (the arrays of the device mirror, as hpsdf_tree_upload lays them out and hpsdf_tree::hostCopies() fetches them: 8-byte node
        // records, every leaf's coefficients padded to whole 128-byte lines)
        hpsdf_tree tree;
        {
            std::vector<hpsdf_node> nodes(nn);
            memcpy(nodes.data(), blk.data() + 16 + 8 * nc, nn * sizeof(hpsdf_node));
            const double* coeffs = (const double*)(blk.data() + 8);
            tree.hRecs.resize(nn);
            for (uint64_t i = 0; i < nn; ++i) {
                if (nodes[i].child_idx != ~0ull) {
                    tree.hRecs[i].a = (uint32_t)nodes[i].child_idx, tree.hRecs[i].b = hpsdf::kInteriorTag;
                } else {
                    const uint64_t cnt = hpsdf::tables().coeffCount[nodes[i].degree];
                    tree.hRecs[i].a = (uint32_t)tree.hPadded.size(), tree.hRecs[i].b = nodes[i].degree;
                    tree.hPadded.insert(tree.hPadded.end(), coeffs + nodes[i].coeffs_start, coeffs + nodes[i].coeffs_start + cnt);
                    tree.hPadded.resize((tree.hPadded.size() + 15) & ~(size_t)15, 0.0);
                }
            }
            tree.hostReady = true;
        }
        hpsdf_config cfg;
        memcpy(&cfg, blk.data() + 16 + 8 * nc + nn * sizeof(hpsdf_node), sizeof cfg);
        for (int a = 0; a < 3; ++a) {
            tree.dev.rootCentre[a] = (double)((cfg.root_min[a] + cfg.root_max[a]) / 2.0f);
            tree.dev.rootInvSizes[a] = (double)(1.0f / (cfg.root_max[a] - cfg.root_min[a]));
        }
        FILE* pf = fopen(argv[4], "rb");
        fseek(pf, 0, SEEK_END); long psz = ftell(pf); fseek(pf, 0, SEEK_SET);
        std::vector<double> xyz(psz / 8); if (fread(xyz.data(), 8, xyz.size(), pf) != xyz.size()) return 11; fclose(pf);
        const size_t np = xyz.size() / 3;
        std::vector<double> res(4 * np, 7.0);  // [values | gradients]; gradient rows of outside points keep the 7s
        for (size_t i = 0; i < np; ++i) {
            res[i] = hpsdf::hostQueryPoint(tree, xyz.data() + 3 * i);
            double v = 0.0;
            hpsdf::hostQueryPointWithGradient(tree, xyz.data() + 3 * i, &v, res.data() + np + 3 * i, hpsdf::reductionLeftAssoc(nullptr));
            if (memcmp(&v, &res[i], 8) != 0) { printf("Query and QueryWithGradient disagree on the value of point %zu\n", i); return 12; }
        }
        FILE* of = fopen(argv[5], "wb"); fwrite(res.data(), 8, res.size(), of); fclose(of);
        printf("host query: %zu points\n", np);
        if (argc > 7) {  // rays: argv[6] = rows of (origin, direction, tMax) f64 x 7 in, argv[7] = rows of (hit, t) f64 x 2 out (t of a miss: -123)
            FILE* rf = fopen(argv[6], "rb");
            fseek(rf, 0, SEEK_END); long rsz = ftell(rf); fseek(rf, 0, SEEK_SET);
            std::vector<double> rays(rsz / 8); if (fread(rays.data(), 8, rays.size(), rf) != rays.size()) return 13; fclose(rf);
            const size_t nr = rays.size() / 7;
            std::vector<double> rr(2 * nr);
            for (size_t i = 0; i < nr; ++i) {
                double tv = -123.0;
                rr[2 * i] = hpsdf::hostQueryRay(tree, &rays[7 * i], &rays[7 * i + 3], rays[7 * i + 6], &tv) ? 1.0 : 0.0;
                rr[2 * i + 1] = tv;
            }
            FILE* wf = fopen(argv[7], "wb"); fwrite(rr.data(), 8, rr.size(), wf); fclose(wf);
            printf("host rays: %zu\n", nr);
        }
    }
    printf("OK\n");
    return 0;
}
