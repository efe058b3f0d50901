// Lab (not part of the product): what the Query path's bytes cost on this GPU when NOTHING else is done with them -- 24 bytes a point
// in, 8 out, no tree.  Each variant is timed walking four distinct batches (the points come from HBM) and on one batch repeated (a
// 10 M-point batch fits the 256 MB Infinity Cache).  The figure query_kernel's time is to be read against: DESIGN.md section 5.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_floor.hip -o /tmp/stream_floor && /tmp/stream_floor [points]
#include <hip/hip_runtime.h>

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
    for (int variant = 0; variant < 3; ++variant) {
        for (int cyc = 1; cyc >= 0; --cyc) {
            auto launch = [&](int k) {
                const double* x = dx[cyc ? k % nb : 0];
                if (variant == 0)
                    hipLaunchKernelGGL(lane_per_point, grid, block, 0, 0, x, n, dout);
                else if (variant == 1)
                    hipLaunchKernelGGL(wide, grid, block, 0, 0, (const double2*)x, n / 2, (double2*)dout);
                else
                    hipLaunchKernelGGL(read_only, grid, block, 0, 0, (const double2*)x, 3 * n / 2, dout);
            };
            for (int k = 0; k < 8; ++k) launch(k);
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < reps; ++k) launch(k);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, bytes = variant == 2 ? 24.0 * n : 32.0 * n;
            printf("%-58s %-22s %7.1f us = %5.2f TB/s = %.3f of 8 TB/s\n",
                   variant == 0 ? "one lane a point (3 x 8 B in at stride 24, 8 B nt out)"
                   : variant == 1 ? "16 bytes a lane in and out (same 32 B a point)"
                                  : "reads only, 16 bytes a lane (24 B a point)",
                   cyc ? "four batches in turn:" : "one batch repeated:", us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
        }
    }
    return 0;
}
