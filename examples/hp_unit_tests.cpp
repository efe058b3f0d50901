// The reference's own unit tests, restated at their exact settings against the drop-in headers:
//   Source/Tests/HPUnitTests.cpp:46-316  -- TestOctreeCreation, TestOctreeContinuity, TestOctreeSerialisation,
//                                           TestOctreeCopying, TestOctreeSDFOperations, TestOctreeCustomDomains
//   Source/Tests/MeshingUnitTests.cpp:110-138 -- TestBVHQuerying
// Same configs (targetErrorThreshold 1e-8, Polynomial(3) nearness weighting, continuity strength 8, root
// [-0.25,5]^3 with continuity on, threadCount = hardware_concurrency), same fields (std::function lambdas over Eigen
// vectors), same sample counts (1 000 000 box.sample() points per loop, 50 for the BVH), same tolerances (1e-2, 5e-2,
// (d1-d2)^2 <= EPSILON_F32), same object choreography (block built in an inner scope and freed by the caller; copy
// constructor, then move assignment).
//
// The check loops are the reference's own (HPUnitTests.cpp:64-75): one scalar Query(sample) per point, 1 000 000 times -- a call of a
// few points is answered on the calling thread (csrc/host_query.cpp, ~0.1 us) -- and the same points then go through ONE batched
// Query(xyz, n, out) call on the GPU, which must return the same bits.  One deliberate difference: TestBVHQuerying reads the OBJ
// given on the command line (Ramesses.obj is not shipped with the reference).
//
//   L=hp-adaptive-signed-distance-field-octree_amd/lib
//   g++ -std=c++17 -O2 -I include examples/hp_unit_tests.cpp -L $L -lhpsdf -Wl,-rpath,$PWD/$L -pthread -o examples/hp_unit_tests
//   examples/hp_unit_tests mesh.obj
#include "HP/Octree.h"
#include "Meshing/BVH.h"
#include "Meshing/Mesh.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

using namespace SDF;

static const usize kSamples = 1000000;  // HPUnitTests.cpp:64

// the reference's check loop: |Query(sample) - truth(sample)| <= tol over kSamples box samples
static bool checkLoop(const Octree& tree, const Eigen::AlignedBox3d& box, const std::function<f64(const Eigen::Vector3d&)>& truth,
                      f64 tol, const char* what) {
    std::vector<f64> xyz(3 * kSamples), one(kSamples), out(kSamples);
    f64 worst = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (usize i = 0; i < kSamples; ++i) {
        const Eigen::Vector3d sample(box.sample());
        const f64 octS = tree.Query(sample);  // HPUnitTests.cpp:67-69
        const f64 err = std::abs(octS - truth(sample));
        if (!(err <= tol)) {
            std::printf("  %s: |Query - true| = %g > %g at (%g, %g, %g)\n", what, err, tol, sample.x(), sample.y(), sample.z());
            return false;
        }
        worst = err > worst ? err : worst;
        xyz[3 * i] = sample.x(), xyz[3 * i + 1] = sample.y(), xyz[3 * i + 2] = sample.z();
        one[i] = octS;
    }
    const f64 loopSeconds = std::chrono::duration<f64>(std::chrono::steady_clock::now() - t0).count();
    tree.Query(xyz.data(), kSamples, out.data());  // the same points in one batched call on the GPU: the same bits
    for (usize i = 0; i < kSamples; ++i)
        if (std::memcmp(&one[i], &out[i], sizeof(f64)) != 0) {
            std::printf("  %s: scalar Query differs from the batched one at point %zu\n", what, (size_t)i);
            return false;
        }
    std::printf("  %s: %zu scalar Query(pt) calls in %.2f s (sampling and the exact field included), max |Query - true| = %.3e (tolerance %g); "
                "batched call: same bits\n", what, (size_t)kSamples, loopSeconds, worst, tol);
    return true;
}

static u32 hardwareThreads() { return std::thread::hardware_concurrency() != 0 ? std::thread::hardware_concurrency() : 1; }

static f64 SphereFunc(const Eigen::Vector3d& pt_, const u32) { return (pt_ - Eigen::Vector3d(0.25, 0, 0)).norm() - 0.5; }
static f64 OtherSphereFunc(const Eigen::Vector3d& pt_, const u32) { return (pt_ + Eigen::Vector3d(0.25, 0, 0)).norm() - 0.5; }
static const Eigen::AlignedBox3d kUnitBox(Eigen::Vector3d(-0.5, -0.5, -0.5), Eigen::Vector3d(0.5, 0.5, 0.5));

static Config weightedConfig(bool continuity) {  // HPUnitTests.cpp:53-58, :88-94
    Config hpConfig;
    hpConfig.targetErrorThreshold = pow(10, -8);
    hpConfig.nearnessWeighting.type = Config::NearnessWeighting::Type::Polynomial;
    hpConfig.nearnessWeighting.strength = 3.0;
    hpConfig.continuity.enforce = continuity;
    if (continuity) hpConfig.continuity.strength = 8.0;
    hpConfig.threadCount = hardwareThreads();
    return hpConfig;
}

static bool TestOctreeCreation() {  // :46-77
    Octree hpOctree;
    hpOctree.Create(weightedConfig(false), SphereFunc);
    return checkLoop(hpOctree, kUnitBox, [](const Eigen::Vector3d& p) { return SphereFunc(p, 0); }, 0.01, "creation");
}

static bool TestOctreeContinuity() {  // :80-112
    Octree hpOctree;
    hpOctree.Create(weightedConfig(true), SphereFunc);
    return checkLoop(hpOctree, kUnitBox, [](const Eigen::Vector3d& p) { return SphereFunc(p, 0); }, 0.01, "continuity");
}

static bool TestOctreeSerialisation() {  // :115-154
    MemoryBlock hpBlock;
    {
        Octree hpOctree;
        hpOctree.Create(weightedConfig(true), SphereFunc);
        hpBlock = hpOctree.ToMemoryBlock();
    }
    Octree hpOctree;
    hpOctree.FromMemoryBlock(hpBlock);
    free(hpBlock.ptr);
    return checkLoop(hpOctree, kUnitBox, [](const Eigen::Vector3d& p) { return SphereFunc(p, 0); }, 0.01, "serialisation");
}

static bool TestOctreeCopying() {  // :157-204
    Octree hpOctree;
    hpOctree.Create(weightedConfig(false), SphereFunc);
    Octree otherOctree = hpOctree;
    if (!checkLoop(otherOctree, kUnitBox, [](const Eigen::Vector3d& p) { return SphereFunc(p, 0); }, 0.01, "copy constructor")) return false;
    otherOctree = std::move(hpOctree);
    return checkLoop(otherOctree, kUnitBox, [](const Eigen::Vector3d& p) { return SphereFunc(p, 0); }, 0.01, "move assignment");
}

static bool TestOctreeSDFOperations() {  // :207-282
    Config hpConfig;
    hpConfig.targetErrorThreshold = pow(10, -8);
    hpConfig.continuity.enforce = false;
    hpConfig.threadCount = hardwareThreads();
    {
        Octree hpOctree;
        hpOctree.Create(hpConfig, SphereFunc);
        hpOctree.UnionSDF(OtherSphereFunc);
        if (!checkLoop(hpOctree, kUnitBox, [](const Eigen::Vector3d& p) { return std::min(SphereFunc(p, 0), OtherSphereFunc(p, 0)); }, 0.05, "UnionSDF"))
            return false;
    }
    {
        Octree hpOctree;
        hpOctree.Create(hpConfig, SphereFunc);
        hpOctree.IntersectSDF(OtherSphereFunc);
        if (!checkLoop(hpOctree, kUnitBox, [](const Eigen::Vector3d& p) { return std::max(SphereFunc(p, 0), OtherSphereFunc(p, 0)); }, 0.05, "IntersectSDF"))
            return false;
    }
    {
        Octree hpOctree;
        hpOctree.Create(hpConfig, SphereFunc);
        hpOctree.SubtractSDF(OtherSphereFunc);
        if (!checkLoop(hpOctree, kUnitBox, [](const Eigen::Vector3d& p) { return std::max(SphereFunc(p, 0) * -1.0, OtherSphereFunc(p, 0)); }, 0.05,
                       "SubtractSDF"))
            return false;
    }
    return true;
}

static bool TestOctreeCustomDomains() {  // :285-316
    auto Sphere075 = [](const Eigen::Vector3d& pt_, const u32) -> f64 { return (pt_ - Eigen::Vector3d(0.25, 0, 0)).norm() - 0.75; };
    Config hpConfig;
    hpConfig.targetErrorThreshold = pow(10, -8);
    hpConfig.continuity.enforce = true;
    hpConfig.continuity.strength = 8.0;
    hpConfig.threadCount = hardwareThreads();
    hpConfig.root = Eigen::AlignedBox3f(Eigen::Vector3f(-0.25, -0.25, -0.25), Eigen::Vector3f(5, 5, 5));
    Octree hpOctree;
    hpOctree.Create(hpConfig, Sphere075);
    const Eigen::AlignedBox3d box(Eigen::Vector3d(-0.25, -0.25, -0.25), Eigen::Vector3d(5, 5, 5));
    return checkLoop(hpOctree, box, [&](const Eigen::Vector3d& p) { return Sphere075(p, 0); }, 0.01, "custom domain + continuity");
}

static bool TestObjParsing(const char* objPath) {  // MeshingUnitTests.cpp:45-49
    Meshing::ObjParser objParser;
    return objParser.Load(objPath);
}

static bool TestMeshCreation(const char* objPath) {  // :52-56
    Meshing::Mesh objMesh;
    return objMesh.CreateFromObj(objPath);
}

static bool TestBVHBuilding(const char* objPath) {  // :92-107 (TestNNOctreeQuerying, :59-89, exercises the point octree of the reference's
    Meshing::Mesh objMesh;                            // CPU BVH builder, which has no counterpart here)
    if (!objMesh.CreateFromObj(objPath)) return false;
    Meshing::BVH objBVH;
    return objBVH.Create(objMesh);
}

static bool TestBVHQuerying(const char* objPath) {  // MeshingUnitTests.cpp:110-138
    Meshing::Mesh objMesh;
    if (!objMesh.CreateFromObj(objPath)) return false;
    Meshing::BVH objBVH;
    if (!objBVH.Create(objMesh)) return false;
    const Eigen::AlignedBox3f meshRoot = objMesh.CalculateMeshAABB();
    f32 worst = 0.0f;
    for (u32 i = 0; i < 50; ++i) {
        const Eigen::Vector3f sample = meshRoot.sample();
        const f32 d1 = objMesh.SignedDistanceAtPt(sample);
        const f32 d2 = objMesh.SignedDistanceAtPt(sample, objBVH);
        if ((d1 - d2) * (d1 - d2) > EPSILON_F32) {
            std::printf("  BVH: naive %g vs BVH %g\n", d1, d2);
            return false;
        }
        worst = std::abs(d1 - d2) > worst ? std::abs(d1 - d2) : worst;
    }
    std::printf("  BVH querying: 50 samples, max |naive - BVH| = %g\n", worst);
    return true;
}

int main(int argc, char** argv) {
    try {
        struct T {
            const char* name;
            std::function<bool()> run;
        };
        std::vector<T> tests = {{"TestOctreeCreation", TestOctreeCreation},           {"TestOctreeContinuity", TestOctreeContinuity},
                                {"TestOctreeSerialisation", TestOctreeSerialisation}, {"TestOctreeCopying", TestOctreeCopying},
                                {"TestOctreeSDFOperations", TestOctreeSDFOperations}, {"TestOctreeCustomDomains", TestOctreeCustomDomains}};
        if (argc > 1) {
            tests.push_back({"TestObjParsing", [&] { return TestObjParsing(argv[1]); }});
            tests.push_back({"TestMeshCreation", [&] { return TestMeshCreation(argv[1]); }});
            tests.push_back({"TestBVHBuilding", [&] { return TestBVHBuilding(argv[1]); }});
            tests.push_back({"TestBVHQuerying", [&] { return TestBVHQuerying(argv[1]); }});
        }
        int passed = 0;
        for (const T& t : tests) {
            const auto t0 = std::chrono::steady_clock::now();
            const bool ok = t.run();
            std::printf("%s: %s (%.2f s)\n", t.name, ok ? "passed" : "FAILED",
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            std::fflush(stdout);
            passed += ok ? 1 : 0;
        }
        std::printf("%d / %zu tests passed\n", passed, tests.size());
        return passed == (int)tests.size() ? 0 : 1;
    } catch (const SDF::Error& e) {
        std::printf("SDF::Error %d: %s\n", e.status, e.what());
        return e.status == HPSDF_ERR_NO_DEVICE ? 42 : 2;
    }
}
