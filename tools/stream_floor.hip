// Lab (not part of the product): what the Query path's bytes cost on this GPU when NOTHING else is done with them -- 24 bytes a point
// in, 8 out, no tree.  Each variant is timed walking four distinct batches (the points come from HBM) and on one batch repeated (a
// 10 M-point batch fits the 256 MB Infinity Cache).  The figure query_kernel's time is to be read against: DESIGN.md section 5.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_floor.hip -o /tmp/stream_floor && /tmp/stream_floor [points]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

// the product's shape: one lane a point, three 8-byte loads at a stride of 24, one non-temporal 8-byte store; grid-stride tiles of 256
__global__ __launch_bounds__(256) void lane_per_point(const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(xyz[3 * i] + xyz[3 * i + 1] + xyz[3 * i + 2], &out[i]);
}
// the same bytes as wide as they go: 16 bytes a lane, fully coalesced, in and out (3 loads and 1 store per 2 points)
__global__ __launch_bounds__(256) void wide(const double2* __restrict__ xyz2, size_t nPairs, double2* __restrict__ out2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nPairs; i += (size_t)gridDim.x * 256) {
        // pair i = points 2i, 2i+1 = 48 bytes = three double2; the lanes read them at a stride of 48 (still every byte of every line)
        const double2 a = xyz2[3 * i], b = xyz2[3 * i + 1], c = xyz2[3 * i + 2];
        double2 r;
        r.x = a.x + a.y + b.x, r.y = b.y + c.x + c.y;
        __builtin_nontemporal_store(r.x, &out2[i].x);
        __builtin_nontemporal_store(r.y, &out2[i].y);
    }
}
// reads only (one value a workgroup written): the input stream alone
__global__ __launch_bounds__(256) void read_only(const double2* __restrict__ xyz2, size_t nChunks, double* __restrict__ out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nChunks; i += (size_t)gridDim.x * 256) {
        const double2 a = xyz2[i];
        s += a.x + a.y;
    }
    if (s == 12345.678) out[blockIdx.x] = s;  // (never true for this data: keeps the loads)
}

// The product's row gather beside the stream: a 512 KB table of 4096 lines of 128 bytes (L2-resident), one line a point, fetched as
// query_kernel fetches it (in step k the 8 lanes of a group bring 6 x 16 bytes of the line of the group's k-th point into LDS; two
// passes of four steps).  MODE 0: the gather alone (the line is chosen by a hash of the point's INDEX; no point is read).  MODE 1: the
// stream and the gather side by side, independent of each other (the points are read and added up, the line still comes from the
// index).  MODE 2: as the product -- the line is chosen by the point's own bits, so a tile's gather waits for its points.  MODE 3: MODE 2
// with non-temporal point loads.
template <int MODE>
__global__ __launch_bounds__(256, 7) void gather(const double* __restrict__ xyz, const char* __restrict__ table, size_t n,
                                                 double* __restrict__ out) {
    __shared__ double2 sRows[4][4][66];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane & ~7, sub = lane & 7;
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
        const size_t i = base + threadIdx.x, il = i < n ? i : n - 1;
        double acc = 0.0;
        uint32_t code = (uint32_t)((il * 2654435761ull) >> 13) & 4095u;
        if (MODE >= 1) {
            const double x = MODE == 3 ? __builtin_nontemporal_load(&xyz[3 * il]) : xyz[3 * il],
                         y = MODE == 3 ? __builtin_nontemporal_load(&xyz[3 * il + 1]) : xyz[3 * il + 1],
                         z = MODE == 3 ? __builtin_nontemporal_load(&xyz[3 * il + 2]) : xyz[3 * il + 2];
            acc = x + y + z;
            if (MODE >= 2) code = (uint32_t)(__double_as_longlong(x * 4096.0 + y * 64.0 + z) >> 30) & 4095u;
        }
        uint32_t ck[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ck[k] = __shfl(code, grp | k, 64);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const char* src = table + (size_t)ck[pass * 4 + k] * 128 + sub * 16;
                if (sub < 6)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sRows[wave][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if ((sub >> 2) == pass) {
                const double2* row = &sRows[wave][sub & 3][grp];
#pragma unroll
                for (int c = 0; c < 6; ++c) acc += row[c].x + row[c].y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (i < n) __builtin_nontemporal_store(acc, &out[i]);
    }
}

// MODE 2 with FEEDERS: the first `feeders` workgroups do not query; their first wave walks the batch's lines in address order, one lane
// a line (4 bytes asked for), staying `ahead` windows of feeders x 64 lines in front of the querying workgroups' progress (a counter
// they bump once per tile) -- the points then reach the querying CUs from the Infinity Cache / L2 instead of from HBM, and the
// long-latency misses are in other waves' queues than the row gather.
__global__ __launch_bounds__(256, 7) void gather_fed(const double* __restrict__ xyz, const char* __restrict__ table, size_t n,
                                                     double* __restrict__ out, unsigned feeders, unsigned ahead,
                                                     unsigned long long* __restrict__ progress, unsigned long long epochBase) {
    __shared__ double2 sRows[4][4][66];
    __shared__ unsigned sSink[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane & ~7, sub = lane & 7;
    if (blockIdx.x < feeders) {
        if (wave != 0) return;
        const size_t lines = (n * 24 + 127) / 128, window = (size_t)feeders * 64;
        const size_t tilesTotal = (n + 255) / 256;
        unsigned long long seen = 0;
        for (size_t w = 0; w * window < lines; ++w) {
            // window w covers lines [w * window, (w+1) * window) = points up to ((w+1) * window * 128 / 24): wait until the queriers are
            // within `ahead` windows of it
            if (w > ahead) {
                const size_t needTiles = ((w - ahead) * window * 128 / 24) / 256;  // tiles that must be done
                const size_t need = needTiles < tilesTotal ? needTiles : tilesTotal;
                while (seen < need) {  // (the counter only grows: it is read again only when what was last seen is not enough)
                    seen = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epochBase;
                    if (seen < need) __builtin_amdgcn_s_sleep(64);
                }
            }
            const size_t line = w * window + (size_t)blockIdx.x * 64 + lane;
            if (line < lines)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(xyz) + line * 128),
                                                 (__attribute__((address_space(3))) void*)&sSink[0], 4, 0, 0);
        }
        return;
    }
    // (the querying workgroups take RUNS of consecutive tiles, so that dispatch order is address order and the front the feeders
    // stay ahead of is one place in the batch, not five)
    const size_t q = blockIdx.x - feeders, nq = gridDim.x - feeders;
    const size_t tilesAll = (n + 255) / 256, run = (tilesAll + nq - 1) / nq;
    const size_t tEnd = (q + 1) * run < tilesAll ? (q + 1) * run : tilesAll;
    for (size_t tile = q * run; tile < tEnd; ++tile) {
        const size_t base = tile * 256;
        const size_t i = base + threadIdx.x, il = i < n ? i : n - 1;
        const double x = xyz[3 * il], y = xyz[3 * il + 1], z = xyz[3 * il + 2];
        double acc = x + y + z;
        const uint32_t code = (uint32_t)(__double_as_longlong(x * 4096.0 + y * 64.0 + z) >> 30) & 4095u;
        uint32_t ck[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ck[k] = __shfl(code, grp | k, 64);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const char* src = table + (size_t)ck[pass * 4 + k] * 128 + sub * 16;
                if (sub < 6)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sRows[wave][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if ((sub >> 2) == pass) {
                const double2* row = &sRows[wave][sub & 3][grp];
#pragma unroll
                for (int c = 0; c < 6; ++c) acc += row[c].x + row[c].y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (i < n) __builtin_nontemporal_store(acc, &out[i]);
        // (one workgroup in 32 reports, for 32: same-address atomics cost ~25 ns each at the L2 -- one a tile was a millisecond a launch)
        if (threadIdx.x == 0 && (q & 31) == 0) __hip_atomic_fetch_add(progress, 32ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 10000000;
    const int nb = 4, reps = 40;
    std::vector<double> h(3 * n);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : h) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        v = (double)(st >> 11) / 9007199254740992.0 - 0.5;
    }
    double* dx[nb];
    double* dout;
    for (int b = 0; b < nb; ++b) {
        CK(hipMalloc(&dx[b], 3 * n * 8));
        CK(hipMemcpy(dx[b], h.data(), 3 * n * 8, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&dout, n * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(8192), block(256);
    char* dtable;
    CK(hipMalloc(&dtable, 4096 * 128));
    CK(hipMemcpy(dtable, h.data(), 4096 * 128, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 7; ++variant) {
        for (int cyc = 1; cyc >= 0; --cyc) {
            auto launch = [&](int k) {
                const double* x = dx[cyc ? k % nb : 0];
                if (variant == 0)
                    hipLaunchKernelGGL(lane_per_point, grid, block, 0, 0, x, n, dout);
                else if (variant == 1)
                    hipLaunchKernelGGL(wide, grid, block, 0, 0, (const double2*)x, n / 2, (double2*)dout);
                else if (variant == 2)
                    hipLaunchKernelGGL(read_only, grid, block, 0, 0, (const double2*)x, 3 * n / 2, dout);
                else if (variant == 3)
                    hipLaunchKernelGGL(gather<0>, grid, block, 0, 0, x, dtable, n, dout);
                else if (variant == 4)
                    hipLaunchKernelGGL(gather<1>, grid, block, 0, 0, x, dtable, n, dout);
                else if (variant == 5)
                    hipLaunchKernelGGL(gather<2>, grid, block, 0, 0, x, dtable, n, dout);
                else
                    hipLaunchKernelGGL(gather<3>, grid, block, 0, 0, x, dtable, n, dout);
            };
            for (int k = 0; k < 8; ++k) launch(k);
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < reps; ++k) launch(k);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, bytes = variant == 2 ? 24.0 * n : variant == 3 ? 8.0 * n : 32.0 * n;
            printf("%-66s %-22s %7.1f us = %5.2f TB/s = %.3f of 8 TB/s\n",
                   variant == 0 ? "one lane a point (3 x 8 B in at stride 24, 8 B nt out)"
                   : variant == 1 ? "16 bytes a lane in and out (same 32 B a point)"
                   : variant == 2 ? "reads only, 16 bytes a lane (24 B a point)"
                   : variant == 3 ? "row gather alone (line by the point's index; 8 B out)"
                   : variant == 4 ? "stream + row gather, independent (line by the index)"
                   : variant == 5 ? "stream + row gather, dependent (line by the point: the product)"
                                  : "the same with non-temporal point loads",
                   cyc ? "four batches in turn:" : "one batch repeated:", us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
        }
    }
    unsigned long long* dprog;
    CK(hipMalloc(&dprog, 8));
    CK(hipMemset(dprog, 0, 8));
    unsigned long long epoch = 0;
    const unsigned long long tiles = (n + 255) / 256;
    unsigned long long reported = 0;  // what one launch adds to the counter: 32 for every tile of every 32nd querying workgroup
    {
        const unsigned long long run = (tiles + 8191) / 8192;
        for (unsigned long long q = 0; q < 8192; q += 32) {
            const unsigned long long a = q * run, b = (q + 1) * run < tiles ? (q + 1) * run : tiles;
            if (b > a) reported += 32 * (b - a);
        }
    }
    for (unsigned feeders : {64u}) {
        for (unsigned ahead : {2u, 8u, 32u}) {
            auto launch = [&](int k) {
                hipLaunchKernelGGL(gather_fed, dim3(8192 + feeders), block, 0, 0, dx[k % nb], dtable, n, dout, feeders, ahead, dprog, epoch);
                epoch += reported;
            };
            for (int k = 0; k < 8; ++k) launch(k);
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < reps; ++k) launch(k);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("the product's pattern with %3u feeder waves %2u windows (%5.1f MB) ahead, four batches in turn: %7.1f us\n", feeders, ahead,
                   (double)ahead * feeders * 64 * 128 / 1e6, ms * 1e3 / reps);
        }
    }
    {   // two launches side by side: the product's pattern on batch k (whose points the previous round's reader brought into the
        // Infinity Cache) while a reads-only launch on another stream brings batch k+1 in from HBM -- does the HBM stream cost the
        // gather its time when OTHER waves carry it?
        hipStream_t s0, s1;
        CK(hipStreamCreate(&s0));
        CK(hipStreamCreate(&s1));
        for (unsigned readerWgs : {256u, 1024u, 8192u}) {
            hipLaunchKernelGGL(read_only, dim3(readerWgs), block, 0, s1, (const double2*)dx[0], 3 * n / 2, dout);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            for (int k = 0; k < reps; ++k) {
                hipLaunchKernelGGL(gather<2>, grid, block, 0, s0, dx[k % nb], dtable, n, dout);
                hipLaunchKernelGGL(read_only, dim3(readerWgs), block, 0, s1, (const double2*)dx[(k + 1) % nb], 3 * n / 2, dout + n - 16384);
                CK(hipEventRecord(e1, s1));
                CK(hipStreamWaitEvent(s0, e1, 0));  // the next round starts when both are done
            }
            CK(hipEventRecord(e1, s0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("the product's pattern on a batch read ahead, beside a %4u-workgroup reader of the next batch: %7.1f us a round\n", readerWgs, ms * 1e3 / reps);
        }
    }
    return 0;
}
