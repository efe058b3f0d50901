// Octree::Create sharded over the GPUs of one node, from C++: one thread per GPU, one RCCL communicator, the frontier of
// every round cut into per-rank slices on the devices, one ncclAllGather per round and one at the end
// (include/hpsdf_rccl.hpp; the reference's Create is Source/HP/Octree.cpp:312-352 -- it has no multi-process form).
// Every rank ends with the identical tree; rank 0 prints it against the single-GPU build.
//
//   L=hp-adaptive-signed-distance-field-octree_amd/lib
//   hipcc -std=c++17 -O2 -I include examples/hp_create_multi_gpu.cpp -L $L -lhpsdf -lrccl -Wl,-rpath,$PWD/$L -pthread -o examples/hp_create_multi_gpu
//   examples/hp_create_multi_gpu [ranks] [targetError]      (ranks <= visible GPUs; HP_SHARE_GPU=1 puts every rank on GPU 0)
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "HP/Octree.h"
#include "hpsdf_rccl.hpp"

int main(int argc, char** argv) {
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0) {
        std::printf("no GPU\n");
        return 42;
    }
    const bool share = std::getenv("HP_SHARE_GPU") != nullptr;
    const int world = argc > 1 ? std::atoi(argv[1]) : nDev;
    const double target = argc > 2 ? std::atof(argv[2]) : 1e-7;
    if (world < 1 || world > 8 || (!share && world > nDev)) {
        std::printf("ranks must be 1..min(8, GPUs)\n");
        return 2;
    }
    if (share && world > 1) {
        std::printf("RCCL needs one GPU per rank (HP_SHARE_GPU is for world = 1 only)\n");
        return 2;
    }
    SDF::Config cfg;
    cfg.targetErrorThreshold = target;
    cfg.continuity.enforce = false;
    cfg.threadCount = 1;
    const SDF::DeviceField field = SDF::DeviceField::Union3();

    SDF::Octree single;
    single.Create(cfg, field);  // warm-up + the answer to compare with
    auto t0 = std::chrono::steady_clock::now();
    single.Create(cfg, field);
    const double oneMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    MemoryBlock want = single.ToMemoryBlock();

    std::vector<ncclComm_t> comms(world);
    std::vector<int> devs(world);
    for (int r = 0; r < world; ++r) devs[r] = r;
    if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) {
        std::printf("ncclCommInitAll failed\n");
        return 3;
    }
    if (world == 1) {
        // one GPU: the sharded build has nothing to exchange, but the adaptor itself still runs once -- an in-place
        // ncclAllGather of one rank on a stream, the call every rank of a larger world makes twice per round
        hipStream_t st;
        unsigned char* buf = nullptr;
        unsigned char host[256], back[256];
        for (int i = 0; i < 256; ++i) host[i] = (unsigned char)(i * 7);
        hpsdf_rccl::Comm comm{comms[0], 0};
        bool fine = hipStreamCreate(&st) == hipSuccess && hipMalloc((void**)&buf, 256) == hipSuccess &&
                    hipMemcpyAsync(buf, host, 256, hipMemcpyHostToDevice, st) == hipSuccess &&
                    hpsdf_rccl::AllGather(&comm, buf, 256, st) == 0 && hipMemcpyAsync(back, buf, 256, hipMemcpyDeviceToHost, st) == hipSuccess &&
                    hipStreamSynchronize(st) == hipSuccess && std::memcmp(host, back, 256) == 0;
        if (buf) (void)hipFree(buf);
        if (!fine) {
            std::printf("hpsdf_rccl::AllGather failed on one rank (RCCL status %d)\n", (int)comm.last);
            return 4;
        }
        std::printf("hpsdf_rccl::AllGather: one-rank in-place all-gather on a stream ok\n");
    }
    std::vector<int> ok(world, 0);
    std::vector<double> ms(world, 0.0);
    std::vector<std::thread> pool;
    for (int r = 0; r < world; ++r)
        pool.emplace_back([&, r] {
            try {
                hpsdf_rccl::Comm comm{world > 1 ? comms[r] : nullptr, r};
                SDF::Octree tree;
                tree.SetDevice(devs[r]);
                if (world > 1) tree.SetRanks(r, world, hpsdf_rccl::AllGather, &comm);
                tree.Create(cfg, field);  // warm-up (buffers, RCCL channels)
                const auto a = std::chrono::steady_clock::now();
                tree.Create(cfg, field);
                ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
                MemoryBlock got = tree.ToMemoryBlock();
                ok[r] = got.size == want.size && std::memcmp(got.ptr, want.ptr, want.size) == 0;
                std::free(got.ptr);
            } catch (const SDF::Error& e) {
                std::printf("rank %d: SDF::Error %d: %s\n", r, e.status, e.what());
            }
        });
    for (auto& t : pool) t.join();
    int good = 0;
    double worst = 0.0;
    for (int r = 0; r < world; ++r) good += ok[r], worst = ms[r] > worst ? ms[r] : worst;
    std::printf("union3 @ %g: 1 GPU %.3f ms; %d ranks %.3f ms (slowest rank); blocks identical to the single-GPU build on %d / %d ranks\n", target,
                oneMs, world, worst, good, world);
    for (auto& c : comms) ncclCommDestroy(c);
    std::free(want.ptr);
    return good == world ? 0 : 1;
}
