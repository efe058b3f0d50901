// The workloads of the reference's benchmark program (Source/Tests/HPBenchmarks.cpp:25-236: creation at 1e-10 with
// exponential nearness weighting, the same with the continuity post-process, 8 M random queries, a 200^3 grid of
// queries, 8 M queries with gradient, UnionSDF at 1e-8), written against the drop-in headers exactly as a user of the
// reference would write them -- reference include paths, SDF::Octree / SDF::Config, std::function fields -- plus the
// two additive forms a throughput-minded caller switches to: fields the GPU evaluates itself and batched Query.
//
//   L=hp-adaptive-signed-distance-field-octree_amd/lib
//   g++ -std=c++17 -O2 -I include examples/hp_benchmarks.cpp -L $L -lhpsdf -Wl,-rpath,$PWD/$L -pthread -o examples/hp_benchmarks
#include "HP/Octree.h"

#include <chrono>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

using namespace SDF;

static double seconds(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

static Config benchmarkConfig(double target, bool weighting, bool continuity) {
    Config c;
    c.targetErrorThreshold = target;
    if (weighting) {
        c.nearnessWeighting.type = Config::NearnessWeighting::Exponential;
        c.nearnessWeighting.strength = 3.0;
    }
    c.continuity.enforce = continuity;
    const unsigned hc = std::thread::hardware_concurrency();  // the reference uses all of them; a GPU box has hundreds
    c.threadCount = hc == 0 ? 1 : (hc > 16 ? 16 : hc);
    return c;
}

int main() {
    try {
        auto SphereFunc = [](const Eigen::Vector3d& pt_, const u32) -> f64 { return (pt_ - Eigen::Vector3d(0.25, 0, 0)).norm() - 0.5; };
        auto OtherSphere = [](const Eigen::Vector3d& pt_, const u32) -> f64 { return (pt_ - Eigen::Vector3d(-0.25, 0, 0)).norm() - 0.5; };

        // ---- creation: the callback as the reference takes it (host threads sample, the GPU fits) ...
        {
            Octree warm;
            warm.Create(benchmarkConfig(1e-4, false, false), SDF::DeviceField::Sphere(0.25, 0, 0, 0.5));  // first-use costs
        }
        Octree tree;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 3; ++rep) {  // (first call: the pinned sample buffers of the host-sampled path grow)
            t0 = std::chrono::steady_clock::now();
            tree.Create(benchmarkConfig(1e-10, true, false), SphereFunc);
            std::printf("Creation (std::function field, sampled by %llu host threads)%s: %.1f ms\n",
                        (unsigned long long)benchmarkConfig(1e-10, true, false).threadCount, rep ? "" : ", first call", seconds(t0) * 1e3);
        }
        // ... and as a field the GPU evaluates itself (first call: the context's arena and scratch grow; then steady state)
        Octree treeDev;
        for (int rep = 0; rep < 3; ++rep) {
            t0 = std::chrono::steady_clock::now();
            treeDev.Create(benchmarkConfig(1e-10, true, false), SDF::DeviceField::Sphere(0.25, 0, 0, 0.5));
            std::printf("Creation (device field)%s: %.2f ms\n", rep ? "" : ", first call", seconds(t0) * 1e3);
        }
        Octree treeCont;
        for (int rep = 0; rep < 3; ++rep) {
            t0 = std::chrono::steady_clock::now();
            treeCont.Create(benchmarkConfig(1e-10, true, true), SDF::DeviceField::Sphere(0.25, 0, 0, 0.5));
            std::printf("Creation + continuity (device field)%s: %.2f ms, %llu CG iterations\n", rep ? "" : ", first call", seconds(t0) * 1e3,
                        (unsigned long long)treeCont.LastContinuityStats().iterations);
        }

        // ---- queries: the reference loops Query(pt); the batched form takes the same points in one call
        const usize n = 8000000;
        std::vector<double> xyz(3 * n), out(n), grad(3 * n);
        std::mt19937_64 rng(5);
        std::uniform_real_distribution<double> U(-0.5, 0.5);
        for (auto& v : xyz) v = U(rng);
        t0 = std::chrono::steady_clock::now();
        // HPBenchmarks.cpp:105-109: 8 M scalar calls.  A call of a few points is answered on the calling thread (csrc/host_query.cpp)
        const usize nScalar = n;
        double acc = 0.0;
        for (usize i = 0; i < nScalar; ++i) acc += treeDev.Query(Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        std::printf("Query(pt), one call per point: %.3f us per call (%llu calls, %.2f s, checksum %.6f)\n", seconds(t0) / nScalar * 1e6,
                    (unsigned long long)nScalar, seconds(t0), acc);
        {
            t0 = std::chrono::steady_clock::now();
            const usize nG = 1000000;
            double accG = 0.0;
            Eigen::Vector3d g(0, 0, 0);
            for (usize i = 0; i < nG; ++i) accG += treeDev.QueryWithGradient(Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), g) + g.x();
            std::printf("QueryWithGradient(pt), one call per point: %.3f us per call (%llu calls, checksum %.6f)\n", seconds(t0) / nG * 1e6,
                        (unsigned long long)nG, accG);
        }
        treeDev.Query(xyz.data(), n, out.data());
        t0 = std::chrono::steady_clock::now();
        treeDev.Query(xyz.data(), n, out.data());
        double dt = seconds(t0);
        std::printf("Query(xyz, 8 M points, host arrays): %.2f ms = %.0f Mpts/s (PCIe-bound: 32 B per point)\n", dt * 1e3, n / dt / 1e6);
        {
            const usize g = 200;
            std::vector<double> grid(3 * g * g * g), gout(g * g * g);
            usize k = 0;
            for (usize x = 0; x < g; ++x)
                for (usize y = 0; y < g; ++y)
                    for (usize z = 0; z < g; ++z) {
                        grid[k++] = -0.5 + (double)x / (g - 1), grid[k++] = -0.5 + (double)y / (g - 1), grid[k++] = -0.5 + (double)z / (g - 1);
                    }
            treeDev.Query(grid.data(), g * g * g, gout.data());
            t0 = std::chrono::steady_clock::now();
            treeDev.Query(grid.data(), g * g * g, gout.data());
            dt = seconds(t0);
            std::printf("Query(200^3 grid, host arrays): %.2f ms = %.0f Mpts/s\n", dt * 1e3, g * g * g / dt / 1e6);
        }
        treeDev.QueryWithGradient(xyz.data(), n, out.data(), grad.data());
        t0 = std::chrono::steady_clock::now();
        treeDev.QueryWithGradient(xyz.data(), n, out.data(), grad.data());
        dt = seconds(t0);
        std::printf("QueryWithGradient(8 M points, host arrays): %.2f ms = %.0f Mpts/s (56 B per point over PCIe)\n", dt * 1e3, n / dt / 1e6);

        // ---- CSG: the tree of one sphere united with the mirrored sphere, 1e-8
        Octree csg;
        csg.Create(benchmarkConfig(1e-8, false, false), SDF::DeviceField::Sphere(0.25, 0, 0, 0.5));
        t0 = std::chrono::steady_clock::now();
        csg.UnionSDF(SDF::DeviceField::Sphere(-0.25, 0, 0, 0.5));
        std::printf("UnionSDF (device field, 1e-8): %.2f ms\n", seconds(t0) * 1e3);
        Octree csg2;
        csg2.Create(benchmarkConfig(1e-8, false, false), SDF::DeviceField::Sphere(0.25, 0, 0, 0.5));
        t0 = std::chrono::steady_clock::now();
        csg2.UnionSDF(OtherSphere);
        std::printf("UnionSDF (std::function field, 1e-8): %.1f ms\n", seconds(t0) * 1e3);
        const Eigen::Vector3d p(0.0, 0.3, 0.1);
        std::printf("union at (0, 0.3, 0.1): %.9f (exact %.9f)\n", csg.Query(p), std::min(SphereFunc(p, 0), OtherSphere(p, 0)));
        return 0;
    } catch (const SDF::Error& e) {
        std::fprintf(stderr, "hpsdf error %d: %s\n", e.status, e.what());
        return 1;
    }
}
