// Development lab #2 (not part of the product): instruction-count and request-shape variants of the cooperative
// LDS-DMA Query kernel, on the same synthetic uniform depth-4 / degree-2 tree as tools/query_lab.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/query_lab2.hip -o /tmp/query_lab2 && /tmp/query_lab2
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

struct alignas(128) Fat {
    uint32_t a, b, pad[2];
    double c[14];
};
struct alignas(128) Row80 {  // coefficients first, no header
    double c[16];
};
struct Args {
    const Fat* fat;
    const Row80* r80;
    double nl[3];
};

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void descend4(double px, double py, double pz, uint32_t& code, double& cx, double& cy, double& cz) {
    cx = cy = cz = 0.0;
    double q = 0.25;
    uint32_t ix = 0, iy = 0, iz = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const bool ux = px >= cx, uy = py >= cy, uz = pz >= cz;
        ix = ix * 2u + (ux ? 1u : 0u), iy = iy * 2u + (uy ? 1u : 0u), iz = iz * 2u + (uz ? 1u : 0u);
        cx = ux ? cx + q : cx - q;
        cy = uy ? cy + q : cy - q;
        cz = uz ? cz + q : cz - q;
        q = q * 0.5;
    }
    code = ix + 16u * (iy + 16u * iz);
}

// cell of p along one axis on the 16-cell grid over [-0.5, 0.5]: k = clamp(floor(16 p) + 8, 0, 15) -- 16 p and its floor
// are exact, so this is the comparison chain's answer; centre = (k - 7.5) / 16 exactly.
__device__ __forceinline__ void cellOf(double p, int& k, double& c) {
    double kf = floor(p * 16.0);
    kf = fmin(fmax(kf, -8.0), 7.0);
    k = (int)kf + 8;
    c = (kf + 0.5) * 0.0625;
}

__device__ __forceinline__ double eval2(const double (&cv)[10], double ux, double uy, double uz, const double* nl) {
    double tx[3], ty[3], tz[3];
    tx[0] = ty[0] = tz[0] = nl[0];
    tx[1] = ux * nl[1], ty[1] = uy * nl[1], tz[1] = uz * nl[1];
    tx[2] = (1.5 * ux * ux - 0.5) * nl[2], ty[2] = (1.5 * uy * uy - 0.5) * nl[2], tz[2] = (1.5 * uz * uz - 0.5) * nl[2];
    const int I[10][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {1, 0, 0}, {0, 0, 2}, {0, 1, 1}, {0, 2, 0}, {1, 0, 1}, {1, 1, 0}, {2, 0, 0}};
    double f = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double lp = tx[I[i][0]];
        lp = lp * ty[I[i][1]];
        lp = lp * tz[I[i][2]];
        f = f + cv[i] * lp;
    }
    return f;
}

// IDX 0: select chain, 1: floor.  CODES 0: shfl, 1: LDS.  PF: prefetch next tile's points -- BROKEN as written (the
// asm loads are still in flight when the compiler reuses their registers: memory fault at n = 10 M); not run.
template <int IDX, int CODES, int PF>
__global__ __launch_bounds__(256, 7) void lab8(Args a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ double2 sd[4][4][66];
    __shared__ uint32_t sCode[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
    const size_t step = (size_t)gridDim.x * 256;
    size_t base = (size_t)blockIdx.x * 256;
    if (base >= n) return;
    double nx = 0, ny = 0, nz = 0;
    if (PF) {
        const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        nx = xyz[3 * i], ny = xyz[3 * i + 1], nz = xyz[3 * i + 2];
    }
    const uint32_t subOff = (uint32_t)j * 16u;
    for (; base < n; base += step) {
        double px, py, pz;
        if (PF) {
            px = nx, py = ny, pz = nz;
        } else {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        }
        uint32_t code;
        double cx, cy, cz;
        if (IDX == 0) {
            descend4(px, py, pz, code, cx, cy, cz);
        } else {
            int kx, ky, kz;
            cellOf(px, kx, cx), cellOf(py, ky, cy), cellOf(pz, kz, cz);
            code = (uint32_t)(kx + 16 * (ky + 16 * kz));
        }
        uint32_t ck[8];
        if (CODES == 1) {
            sCode[w][lane] = code;
            __builtin_amdgcn_wave_barrier();
            const uint4 c0 = *reinterpret_cast<const uint4*>(&sCode[w][grp]);
            const uint4 c1 = *reinterpret_cast<const uint4*>(&sCode[w][grp + 4]);
            ck[0] = c0.x, ck[1] = c0.y, ck[2] = c0.z, ck[3] = c0.w, ck[4] = c1.x, ck[5] = c1.y, ck[6] = c1.z, ck[7] = c1.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) ck[k] = __shfl(code, grp | k, 64);
        }
        const double* np = nullptr;
        if (PF) {
            const size_t nb = base + step;
            const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;
            np = xyz + 3 * ni;
        }
        double cv[10];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t off = (ck[half * 4 + k] << 7) + subOff;
                const char* src = reinterpret_cast<const char*>(a.fat) + off;
                if (j < 6)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sd[w][k][0], 16, 0, 0);
            }
            if (PF && half == 1) {
                asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16"
                             : "=&v"(nx), "=&v"(ny), "=&v"(nz) : "v"(np) : "memory");
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_wave_barrier();
            if ((j >> 2) == half) {
                const double2* row = &sd[w][j & 3][grp];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const double2 v = row[1 + q];
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
    }
}

// lab8 with the tile's points (1536 contiguous bytes) fetched line by line instead of three stride-24 loads that each
// touch all 12 lines: PTS 1 = two dwordx4 LDS-DMA instructions (lane-linear), PTS 2 = dwordx4 to registers + ds_write.
template <int PTS>
__global__ __launch_bounds__(256, 7) void lab8D(Args a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ double2 sd[4][4][66];
    __shared__ double2 sPts[4][96];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
    const size_t step = (size_t)gridDim.x * 256;
    const uint32_t subOff = (uint32_t)j * 16u;
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += step) {
        const size_t tile = base + (size_t)w * 64;  // this wave's 64 points
        double px, py, pz;
        if (tile + 64 <= n) {
            const char* src = reinterpret_cast<const char*>(xyz + 3 * tile) + lane * 16;
            if (PTS == 1) {
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sPts[w][0], 16, 0, 0);
                if (lane < 32)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + 1024), (LDS_AS void*)&sPts[w][64], 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                const double2 v0 = *reinterpret_cast<const double2*>(src);
                double2 v1 = make_double2(0, 0);
                if (lane < 32) v1 = *reinterpret_cast<const double2*>(src + 1024);
                sPts[w][lane] = v0;
                if (lane < 32) sPts[w][64 + lane] = v1;
            }
            __builtin_amdgcn_wave_barrier();
            const double* sp = reinterpret_cast<const double*>(&sPts[w][0]) + 3 * lane;
            px = sp[0], py = sp[1], pz = sp[2];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {
            const size_t i = tile + lane < n ? tile + lane : n - 1;
            px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        }
        uint32_t code;
        double cx, cy, cz;
        descend4(px, py, pz, code, cx, cy, cz);
        uint32_t ck[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ck[k] = __shfl(code, grp | k, 64);
        double cv[10];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t off = (ck[half * 4 + k] << 7) + subOff;
                const char* src = reinterpret_cast<const char*>(a.fat) + off;
                if (j < 6)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sd[w][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if ((j >> 2) == half) {
                const double2* row = &sd[w][j & 3][grp];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const double2 v = row[1 + q];
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        if (tile + lane < n) out[tile + lane] = r;
    }
}

// 80-byte rows, 5 lanes per row, 12 rows per DMA instruction, 6 instructions per 64 points (one pass, 6.2 KB/wave)
// or two passes of 3 (3.1 KB/wave).
template <int PASSES, int PF>
__global__ __launch_bounds__(256, 7) void lab5(Args a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    constexpr int SPP = 6 / PASSES;  // steps per pass
    __shared__ double2 sd[4][SPP][66];
    __shared__ uint32_t sCode[4][72];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rowInStep = lane / 5, chunk = lane - rowInStep * 5;  // lanes 60..63: rowInStep 12 (idle)
    const bool fetcher = lane < 60;
    const int myStep = lane / 12, myRow = lane - myStep * 12;      // where this lane's own row lands
    const size_t step = (size_t)gridDim.x * 256;
    size_t base = (size_t)blockIdx.x * 256;
    if (base >= n) return;
    double nx = 0, ny = 0, nz = 0;
    if (PF) {
        const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        nx = xyz[3 * i], ny = xyz[3 * i + 1], nz = xyz[3 * i + 2];
    }
    if (lane < 8) sCode[w][64 + lane] = 0;  // steps read past point 63
    for (; base < n; base += step) {
        double px, py, pz;
        if (PF) {
            px = nx, py = ny, pz = nz;
        } else {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        }
        int kx, ky, kz;
        double cx, cy, cz;
        cellOf(px, kx, cx), cellOf(py, ky, cy), cellOf(pz, kz, cz);
        sCode[w][lane] = (uint32_t)(kx + 16 * (ky + 16 * kz));
        __builtin_amdgcn_wave_barrier();
        uint32_t ck[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) ck[s] = sCode[w][rowInStep + 12 * s];  // lanes 60..63 read row 12: harmless
        const double* np = nullptr;
        if (PF) {
            const size_t nb = base + step;
            const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;
            np = xyz + 3 * ni;
        }
        double cv[10];
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
            for (int s = 0; s < SPP; ++s) {
                const uint32_t off = (ck[pass * SPP + s] << 7) + (uint32_t)chunk * 16u;
                const char* src = reinterpret_cast<const char*>(a.r80) + off;
                if (fetcher)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sd[w][s][0], 16, 0, 0);
            }
            if (PF && pass == PASSES - 1) {
                asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16"
                             : "=&v"(nx), "=&v"(ny), "=&v"(nz) : "v"(np) : "memory");
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_wave_barrier();
            if (myStep / SPP == pass) {
                const double2* row = &sd[w][myStep % SPP][myRow * 5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const double2 v = row[q];
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
    }
}

// 64-byte rows (c0..c7: one L2 sector) fetched by 4 lanes each, 16 rows per DMA instruction, 4 instructions per 64
// points in one pass; c8, c9 of every cell live in a 64 KB LDS table shared by the workgroup's 16 waves.
struct alignas(64) Row64 {
    double c[8];
};
struct Args64 {
    const Row64* r64;
    const double2* tails;  // [4096] (c8, c9)
    double nl[3];
};
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lab64(Args64 a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    extern __shared__ double2 smem[];
    double2* sTail = smem;                                      // [4096]
    double2* sWin = smem + 4096;                                // [WAVES][4][66]
    uint32_t* sCode = reinterpret_cast<uint32_t*>(sWin + WAVES * 4 * 66);  // [WAVES][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) sTail[i] = a.tails[i];
    __syncthreads();
    const int rowInStep = lane >> 2, chunk = lane & 3;  // 16 rows per step
    const int myStep = lane >> 4, myRow = lane & 15;
    const size_t step = (size_t)gridDim.x * WAVES * 64;
    for (size_t base = (size_t)blockIdx.x * WAVES * 64; base < n; base += step) {
        const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        int kx, ky, kz;
        double cx, cy, cz;
        cellOf(px, kx, cx), cellOf(py, ky, cy), cellOf(pz, kz, cz);
        const uint32_t code = (uint32_t)(kx + 16 * (ky + 16 * kz));
        sCode[w * 64 + lane] = code;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ck = sCode[w * 64 + rowInStep + 16 * s];
            const char* src = reinterpret_cast<const char*>(a.r64) + (ck << 6) + (uint32_t)chunk * 16u;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sWin[(w * 4 + s) * 66], 16, 0, 0);
        }
        const double2 tail = sTail[code];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double cv[10];
        const double2* row = &sWin[(w * 4 + myStep) * 66 + myRow * 4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 v = row[q];
            cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
        }
        cv[8] = tail.x, cv[9] = tail.y;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
    }
}

// lab64 with the points of the tile after next already in flight (one loop body, registers F -> N -> current as in
// labPipe1): 16 waves per CU is all the 64 KB table leaves room for, so each wave has to keep more in flight.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lab64P(Args64 a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    extern __shared__ double2 smem[];
    double2* sTail = smem;
    double2* sWin = smem + 4096;
    uint32_t* sCode = reinterpret_cast<uint32_t*>(sWin + WAVES * 4 * 66);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) sTail[i] = a.tails[i];
    __syncthreads();
    const int rowInStep = lane >> 2, chunk = lane & 3;
    const int myStep = lane >> 4, myRow = lane & 15;
    const size_t step = (size_t)gridDim.x * WAVES * 64;
    size_t base = (size_t)blockIdx.x * WAVES * 64;
    if (base >= n) return;
    double nx, ny, nz, fx, fy, fz;
    {
        const size_t i0 = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const size_t i1 = base + step + threadIdx.x < n ? base + step + threadIdx.x : n - 1;
        nx = xyz[3 * i0], ny = xyz[3 * i0 + 1], nz = xyz[3 * i0 + 2];
        fx = xyz[3 * i1], fy = xyz[3 * i1 + 1], fz = xyz[3 * i1 + 2];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx), "+v"(ny), "+v"(nz), "+v"(fx), "+v"(fy), "+v"(fz)::"memory");
    }
    for (; base < n; base += step) {
        const double px = nx, py = ny, pz = nz;
        int kx, ky, kz;
        double cx, cy, cz;
        cellOf(px, kx, cx), cellOf(py, ky, cy), cellOf(pz, kz, cz);
        const uint32_t code = (uint32_t)(kx + 16 * (ky + 16 * kz));
        sCode[w * 64 + lane] = code;
        __builtin_amdgcn_wave_barrier();
        // the loads into F issued one iteration ago have had the whole iteration; they retire here, before this tile's
        // rows are even asked for, and move to N -- so the rows' wait below covers nothing but the rows
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(fx), "+v"(fy), "+v"(fz)::"memory");
        nx = fx, ny = fy, nz = fz;
        asm volatile("" : "+v"(nx), "+v"(ny), "+v"(nz));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ck = sCode[w * 64 + rowInStep + 16 * s];
            const char* src = reinterpret_cast<const char*>(a.r64) + (ck << 6) + (uint32_t)chunk * 16u;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sWin[(w * 4 + s) * 66], 16, 0, 0);
        }
        {
            const size_t nb = base + 2 * step;
            const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;
            const double* np = xyz + 3 * ni;
            asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16"
                         : "=&v"(fx), "=&v"(fy), "=&v"(fz) : "v"(np) : "memory");
        }
        const double2 tail = sTail[code];
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double cv[10];
        const double2* row = &sWin[(w * 4 + myStep) * 66 + myRow * 4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 v = row[q];
            cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
        }
        cv[8] = tail.x, cv[9] = tail.y;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fx), "+v"(fy), "+v"(fz)::"memory");
}

// lab64P with ONE wait per tile: the rows of tile i are asked for first (its points retired a tile ago), the points of
// tile i + 2 go out right behind them into the registers tile i just vacated, and vmcnt(3) retires the rows together
// with the points of tile i + 1 (older than the rows).  Two point sets are in flight, so the loop body exists twice.
#define LAB64_BODY(PX, PY, PZ, TILE)                                                                                 \
    {                                                                                                                \
        const size_t tb_ = (TILE);                                                                                   \
        int kx, ky, kz;                                                                                              \
        double cx, cy, cz;                                                                                           \
        cellOf(PX, kx, cx), cellOf(PY, ky, cy), cellOf(PZ, kz, cz);                                                  \
        const double ux = (PX - cx) * 32.0, uy = (PY - cy) * 32.0, uz = (PZ - cz) * 32.0;                            \
        const uint32_t code = (uint32_t)(kx + 16 * (ky + 16 * kz));                                                  \
        sCode[w * 64 + lane] = code;                                                                                 \
        __builtin_amdgcn_wave_barrier();                                                                             \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                              \
            uint32_t ck = sCode[w * 64 + rowInStep + 16 * s];                                                        \
            if (AUX == 100) ck &= 15u; /* experiment: every row an L1 hit (results wrong) */                         \
            const char* src = reinterpret_cast<const char*>(a.r64) + (ck << 6) + (uint32_t)chunk * 16u;              \
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sWin[(w * 4 + s) * 66], 16, 0, AUX == 100 ? 0 : AUX); \
        }                                                                                                            \
        {                                                                                                            \
            const size_t nb = tb_ + 2 * step;                                                                        \
            const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;                                     \
            const double* np = xyz + 3 * ni;                                                                         \
            asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16" \
                         : "=&v"(PX), "=&v"(PY), "=&v"(PZ) : "v"(np) : "memory");                                    \
        }                                                                                                            \
        const double2 tail = sTail[code];                                                                            \
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                             \
        __builtin_amdgcn_wave_barrier();                                                                             \
        double cv[10];                                                                                               \
        const double2* row = &sWin[(w * 4 + myStep) * 66 + myRow * 4];                                               \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
            const double2 v = row[q];                                                                                \
            cv[2 * q] = v.x, cv[2 * q + 1] = v.y;                                                                    \
        }                                                                                                            \
        cv[8] = tail.x, cv[9] = tail.y;                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_wave_barrier();                                                                             \
        const double r = eval2(cv, ux, uy, uz, a.nl);                                                                \
        if (tb_ + threadIdx.x < n) out[tb_ + threadIdx.x] = r;                                                       \
    }

template <int WAVES, int AUX>
__global__ __launch_bounds__(WAVES * 64) void lab64P2(Args64 a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    extern __shared__ double2 smem[];
    double2* sTail = smem;
    double2* sWin = smem + 4096;
    uint32_t* sCode = reinterpret_cast<uint32_t*>(sWin + WAVES * 4 * 66);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) sTail[i] = a.tails[i];
    __syncthreads();
    const int rowInStep = lane >> 2, chunk = lane & 3;
    const int myStep = lane >> 4, myRow = lane & 15;
    const size_t step = (size_t)gridDim.x * WAVES * 64;
    size_t base = (size_t)blockIdx.x * WAVES * 64;
    if (base >= n) return;
    double ax, ay, az, bx, by, bz;
    {
        const size_t i0 = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const size_t i1 = base + step + threadIdx.x < n ? base + step + threadIdx.x : n - 1;
        ax = xyz[3 * i0], ay = xyz[3 * i0 + 1], az = xyz[3 * i0 + 2];
        bx = xyz[3 * i1], by = xyz[3 * i1 + 1], bz = xyz[3 * i1 + 2];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz)::"memory");
    }
    while (true) {
        LAB64_BODY(ax, ay, az, base)
        base += step;
        if (base >= n) break;
        // B's loads are older than the rows that wait just retired
        asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz));
        LAB64_BODY(bx, by, bz, base)
        base += step;
        if (base >= n) break;
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az));
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz)::"memory");
}

// 64-byte rows as above, the (c8, c9) tails fetched by a fifth DMA step from a compact 64 KB global table (one lane
// per point): two L2 sectors per point as today, but five DMA instructions, one wait, 5.2 KB of LDS per wave.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void lab64t(Args64 a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ double2 sWin[4][5][66];
    __shared__ uint32_t sCode[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rowInStep = lane >> 2, chunk = lane & 3;
    const int myStep = lane >> 4, myRow = lane & 15;
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += step) {
        const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        int kx, ky, kz;
        double cx, cy, cz;
        cellOf(px, kx, cx), cellOf(py, ky, cy), cellOf(pz, kz, cz);
        const uint32_t code = (uint32_t)(kx + 16 * (ky + 16 * kz));
        sCode[w][lane] = code;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ck = sCode[w][rowInStep + 16 * s];
            const char* src = reinterpret_cast<const char*>(a.r64) + (ck << 6) + (uint32_t)chunk * 16u;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sWin[w][s][0], 16, 0, 0);
        }
        {
            const char* src = reinterpret_cast<const char*>(a.tails) + (code << 4);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sWin[w][4][0], 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double cv[10];
        const double2* row = &sWin[w][myStep][myRow * 4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 v = row[q];
            cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
        }
        const double2 tail = sWin[w][4][lane];
        cv[8] = tail.x, cv[9] = tail.y;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
    }
}

// Two point sets in flight per wave: tile i is processed from set (i & 1); right after its second batch of DMAs the
// points of tile i + 2 are requested into the same set (its values are already turned into code / local coordinates).
// The wait for the second batch is vmcnt(3) (the three point loads are younger), the wait for the first batch of
// the NEXT tile is vmcnt(0), which retires them -- by then they have been in flight for an evaluation, a store and a
// whole DMA round trip.  vmcnt retires in order, so this is as far ahead as a wave can look.
#define LAB_PIPE_BODY(SX, SY, SZ, TILE_BASE)                                                                         \
    {                                                                                                                \
        const size_t base_ = (TILE_BASE);                                                                            \
        asm volatile("" : "+v"(SX), "+v"(SY), "+v"(SZ));                                                             \
        const double px = SX, py = SY, pz = SZ;                                                                      \
        uint32_t code;                                                                                               \
        double cx, cy, cz;                                                                                           \
        descend4(px, py, pz, code, cx, cy, cz);                                                                      \
        const double ux = (px - cx) * 32.0, uy = (py - cy) * 32.0, uz = (pz - cz) * 32.0;                            \
        double cv[10];                                                                                               \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                     \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                          \
                const uint32_t off = ((uint32_t)__shfl(code, grp | (half * 4 + k), 64) << 7) + (uint32_t)j * 16u;    \
                const char* src = reinterpret_cast<const char*>(a.fat) + off;                                        \
                if (j < 6) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sd[w][k][0], 16, 0, 0); \
            }                                                                                                        \
            if (half == 0) {                                                                                         \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
            } else {                                                                                                 \
                const size_t nb = base_ + 2 * step;                                                                  \
                const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;                                 \
                const double* np = xyz + 3 * ni;                                                                     \
                asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16" \
                             : "=&v"(SX), "=&v"(SY), "=&v"(SZ) : "v"(np) : "memory");                                \
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                     \
            }                                                                                                        \
            __builtin_amdgcn_wave_barrier();                                                                         \
            if ((j >> 2) == half) {                                                                                  \
                const double2* row = &sd[w][j & 3][grp];                                                             \
                _Pragma("unroll") for (int q = 0; q < 5; ++q) {                                                      \
                    const double2 v = row[1 + q];                                                                    \
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;                                                            \
                }                                                                                                    \
            }                                                                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
            __builtin_amdgcn_wave_barrier();                                                                         \
        }                                                                                                            \
        const double r = eval2(cv, ux, uy, uz, a.nl);                                                                \
        if (base_ + threadIdx.x < n) out[base_ + threadIdx.x] = r;                                                   \
    }
template <int MINW>
__global__ __launch_bounds__(256, MINW) void labPipe(Args a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ double2 sd[4][4][66];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
    const size_t step = (size_t)gridDim.x * 256;
    size_t base = (size_t)blockIdx.x * 256;
    if (base >= n) return;
    double ax, ay, az, bx, by, bz;
    {
        const size_t i0 = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const size_t i1 = base + step + threadIdx.x < n ? base + step + threadIdx.x : n - 1;
        ax = xyz[3 * i0], ay = xyz[3 * i0 + 1], az = xyz[3 * i0 + 2];
        bx = xyz[3 * i1], by = xyz[3 * i1 + 1], bz = xyz[3 * i1 + 2];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz)::"memory");
    }
    for (;;) {
        LAB_PIPE_BODY(ax, ay, az, base)
        base += step;
        if (base >= n) break;
        LAB_PIPE_BODY(bx, by, bz, base)
        base += step;
        if (base >= n) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last prefetches must land before the registers are released
}

// The same look-ahead with ONE loop body: the loads of tile i + 2 always target the registers F; they are retired by the
// vmcnt(0) of tile i + 1's first batch, and only then copied to N (the "next" set), which becomes the current points
// at the top of the following iteration.  A register is never read or moved while its load is in flight.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void labPipe1(Args a, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ double2 sd[4][4][66];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
    const size_t step = (size_t)gridDim.x * 256;
    size_t base = (size_t)blockIdx.x * 256;
    if (base >= n) return;
    double nx, ny, nz, fx, fy, fz;
    {
        const size_t i0 = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const size_t i1 = base + step + threadIdx.x < n ? base + step + threadIdx.x : n - 1;
        nx = xyz[3 * i0], ny = xyz[3 * i0 + 1], nz = xyz[3 * i0 + 2];
        fx = xyz[3 * i1], fy = xyz[3 * i1 + 1], fz = xyz[3 * i1 + 2];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx), "+v"(ny), "+v"(nz), "+v"(fx), "+v"(fy), "+v"(fz)::"memory");
    }
    for (; base < n; base += step) {
        const double px = nx, py = ny, pz = nz;  // N is complete (copied from F behind a vmcnt(0), or the prologue)
        uint32_t code;
        double cx, cy, cz;
        descend4(px, py, pz, code, cx, cy, cz);
        const double ux = (px - cx) * 32.0, uy = (py - cy) * 32.0, uz = (pz - cz) * 32.0;
        double cv[10];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t off = ((uint32_t)__shfl(code, grp | (half * 4 + k), 64) << 7) + (uint32_t)j * 16u;
                const char* src = reinterpret_cast<const char*>(a.fat) + off;
                if (j < 6) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sd[w][k][0], 16, 0, 0);
            }
            if (half == 0) {
                // retires the first batch AND the loads into F issued during the previous iteration
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(fx), "+v"(fy), "+v"(fz)::"memory");
                nx = fx, ny = fy, nz = fz;
                asm volatile("" : "+v"(nx), "+v"(ny), "+v"(nz));
            } else {
                const size_t nb = base + 2 * step;
                const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;
                const double* np = xyz + 3 * ni;
                asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16"
                             : "=&v"(fx), "=&v"(fy), "=&v"(fz) : "v"(np) : "memory");
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            }
            __builtin_amdgcn_wave_barrier();
            if ((j >> 2) == half) {
                const double2* row = &sd[w][j & 3][grp];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const double2 v = row[1 + q];
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const double r = eval2(cv, ux, uy, uz, a.nl);
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fx), "+v"(fy), "+v"(fz)::"memory");
}

// The product's structure (8-lane rows, two passes) with s_memtime stamps: where a wave-iteration's time goes.
// acc[0..4]: cycles from loop top to points arrived / codes + DMA pass 0 issued & landed / pass 1 landed / evaluated /
// stored; acc[5]: iterations.  Summed over waves with atomics (diagnostic build only).
template <int PF>
__global__ __launch_bounds__(256, 6) void labTimed(Args a, const double* __restrict__ xyz, size_t n, double* __restrict__ out,
                                                   unsigned long long* __restrict__ acc) {
    __shared__ double2 sd[4][4][66];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
    const size_t step = (size_t)gridDim.x * 256;
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, iters = 0;
    const bool timed = (blockIdx.x & 63) == 0 && w == 0;  // one wave in 256 carries the stamps: the others run undisturbed
#define STAMP() (timed ? __builtin_amdgcn_s_memtime() : 0ull)
    const unsigned long long r0 = timed ? __builtin_amdgcn_s_memrealtime() : 0ull, c0 = STAMP();
    double nx = 0, ny = 0, nz = 0;
    if (PF) {
        const size_t b0 = (size_t)blockIdx.x * 256;
        const size_t i = b0 + threadIdx.x < n ? b0 + threadIdx.x : n - 1;
        nx = xyz[3 * i], ny = xyz[3 * i + 1], nz = xyz[3 * i + 2];
    }
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += step) {
        const unsigned long long s0 = STAMP();
        double px, py, pz;
        if (PF) {
            // the prefetch of this tile's points was issued behind the previous tile's first four DMAs
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx), "+v"(ny), "+v"(nz)::"memory");
            px = nx, py = ny, pz = nz;
        } else {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(px), "+v"(py), "+v"(pz)::"memory");
        }
        const unsigned long long s1 = STAMP();
        uint32_t code;
        double cx, cy, cz;
        descend4(px, py, pz, code, cx, cy, cz);
        uint32_t ck[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ck[k] = __shfl(code, grp | k, 64);
        double cv[10];
        unsigned long long s2 = 0, s3 = 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t off = (ck[half * 4 + k] << 7) + (uint32_t)j * 16u;
                const char* src = reinterpret_cast<const char*>(a.fat) + off;
                if (j < 6) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)&sd[w][k][0], 16, 0, 0);
            }
            if (PF && half == 0) {
                const size_t nb = base + step;
                const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;
                const double* np = xyz + 3 * ni;
                asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %3, off offset:16"
                             : "=&v"(nx), "=&v"(ny), "=&v"(nz) : "v"(np) : "memory");
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx), "+v"(ny), "+v"(nz)::"memory");
            }
            __builtin_amdgcn_wave_barrier();
            if (half == 0) s2 = STAMP(); else s3 = STAMP();
            if ((j >> 2) == half) {
                const double2* row = &sd[w][j & 3][grp];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const double2 v = row[1 + q];
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0, a.nl);
        asm volatile("" : "+v"(r));
        const unsigned long long s4 = STAMP();
        if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
        const unsigned long long s5 = STAMP();
        tA += s1 - s0, tB += s2 - s1, tC += s3 - s2, tD += s4 - s3, tE += s5 - s4, ++iters;
    }
    const unsigned long long r1 = timed ? __builtin_amdgcn_s_memrealtime() : 0ull, c1 = STAMP();
#undef STAMP
    if (lane == 0 && timed) {
        atomicAdd(acc + 0, tA), atomicAdd(acc + 1, tB), atomicAdd(acc + 2, tC), atomicAdd(acc + 3, tD), atomicAdd(acc + 4, tE);
        atomicAdd(acc + 5, iters), atomicAdd(acc + 6, c1 - c0), atomicAdd(acc + 7, r1 - r0);
    }
}

template <typename K>
float timeIt(K launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 10000000;
    std::vector<double> pts(3 * n);
    uint64_t s = 12345;
    for (auto& v : pts) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        v = (double)(z >> 11) * 0x1p-53 - 0.5;
    }
    // a few points on cell boundaries / root faces: the floor index must agree with the comparison chain
    for (int i = 0; i < 3000 && (size_t)i < 3 * n; ++i) pts[i] = ((i * 7) % 17 - 8) / 16.0;
    std::vector<Fat> fat(4096);
    std::vector<Row80> r80(4096);
    for (int i = 0; i < 4096; ++i) {
        fat[i].a = i, fat[i].b = 2;
        for (int k = 0; k < 10; ++k) {
            const double c = std::sin(i * 0.37 + k);
            fat[i].c[k] = c, r80[i].c[k] = c;
        }
    }
    Args a;
    for (int j = 0; j < 3; ++j) a.nl[j] = std::sqrt((2.0 * j + 1.0) * 16.0);
    double *dx, *dout;
    Fat* dfat;
    Row80* dr80;
    CK(hipMalloc(&dx, pts.size() * 8));
    CK(hipMalloc(&dout, n * 8));
    CK(hipMalloc(&dfat, fat.size() * sizeof(Fat)));
    CK(hipMalloc(&dr80, r80.size() * sizeof(Row80)));
    CK(hipMemcpy(dx, pts.data(), pts.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfat, fat.data(), fat.size() * sizeof(Fat), hipMemcpyHostToDevice));
    CK(hipMemcpy(dr80, r80.data(), r80.size() * sizeof(Row80), hipMemcpyHostToDevice));
    a.fat = dfat, a.r80 = dr80;
    std::vector<Row64> r64(4096);
    std::vector<double2> tails(4096);
    for (int i = 0; i < 4096; ++i) {
        for (int k = 0; k < 8; ++k) r64[i].c[k] = fat[i].c[k];
        tails[i] = double2{fat[i].c[8], fat[i].c[9]};
    }
    Row64* dr64;
    double2* dtails;
    CK(hipMalloc(&dr64, r64.size() * sizeof(Row64)));
    CK(hipMalloc(&dtails, tails.size() * sizeof(double2)));
    CK(hipMemcpy(dr64, r64.data(), r64.size() * sizeof(Row64), hipMemcpyHostToDevice));
    CK(hipMemcpy(dtails, tails.data(), tails.size() * sizeof(double2), hipMemcpyHostToDevice));
    Args64 a64;
    a64.r64 = dr64, a64.tails = dtails;
    for (int j = 0; j < 3; ++j) a64.nl[j] = a.nl[j];
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 100>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P2<16, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64P<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)lab64<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<double> h(n), ref;
    auto check = [&](const char* name, float ms, int grid) {
        CK(hipMemcpy(h.data(), dout, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        if (ref.empty()) ref = h;
        size_t shown = 0;
        for (size_t i = 0; i < n; ++i) {
            if (h[i] != ref[i]) {
                if (shown < 6) printf("   mismatch at %zu (tile %zu, lane %zu): %.6g vs %.6g\n", i, i / 256, i % 256, h[i], ref[i]), ++shown;
                ++bad;
            }
        }
        printf("%-58s grid %6d : %7.1f us  %7.1f GB/s alg  mismatches %zu\n", name, grid, ms * 1e3, 32.0 * n / ms / 1e6, bad);
        fflush(stdout);
    };
#define RUN8(I, C, P, g) check("8-lane rows  idx=" #I " codes=" #C " pf=" #P, timeIt([&] { hipLaunchKernelGGL((lab8<I, C, P>), dim3(g), dim3(256), 0, 0, a, dx, n, dout); }, 10), g)
#define RUN5(PS, P, g) check("5-lane rows (80 B)  passes=" #PS " pf=" #P, timeIt([&] { hipLaunchKernelGGL((lab5<PS, P>), dim3(g), dim3(256), 0, 0, a, dx, n, dout); }, 10), g)
    const char* only = argc > 2 ? argv[2] : "";
    (void)only;
    {
        unsigned long long* dacc;
        CK(hipMalloc(&dacc, 8 * sizeof(unsigned long long)));
        for (int gi = 0; gi < 6; ++gi) {
            const int g = gi < 2 ? 8192 : (gi < 4 ? 8192 : 2048), pf = gi & 1;
            CK(hipMemset(dacc, 0, 8 * sizeof(unsigned long long)));
            hipEvent_t ev0, ev1;
            CK(hipEventCreate(&ev0));
            CK(hipEventCreate(&ev1));
            CK(hipEventRecord(ev0));
            if (pf)
                hipLaunchKernelGGL(labTimed<1>, dim3(g), dim3(256), 0, 0, a, dx, n, dout, dacc);
            else
                hipLaunchKernelGGL(labTimed<0>, dim3(g), dim3(256), 0, 0, a, dx, n, dout, dacc);
            CK(hipEventRecord(ev1));
            CK(hipDeviceSynchronize());
            float kms = 0;
            CK(hipEventElapsedTime(&kms, ev0, ev1));
            unsigned long long h8[8];
            CK(hipMemcpy(h8, dacc, sizeof h8, hipMemcpyDeviceToHost));
            const double it = (double)h8[5], mhz = (double)h8[6] / (double)h8[7] * 100.0;
            printf("timed structure pf=%d, grid %d: clock %.0f MHz; per wave-iteration (cycles): points %.0f | pass 0 (codes, issue, wait) %.0f | pass 1 %.0f | "
                   "LDS read + eval %.0f | store issue %.0f | total %.0f = %.2f us\n",
                   pf, g, mhz, h8[0] / it, h8[1] / it, h8[2] / it, h8[3] / it, h8[4] / it, (h8[0] + h8[1] + h8[2] + h8[3] + h8[4]) / it,
                   (h8[0] + h8[1] + h8[2] + h8[3] + h8[4]) / it / mhz);
            const double tw = (g + 63) / 64;  // timed waves
            printf("   kernel %.1f us by events; timed waves %.0f, iterations per wave %.2f, s_memtime ticks per wave %.0f (%.1f us)\n", kms * 1e3, tw,
                   it / tw, (double)h8[6] / tw, (double)h8[6] / tw / mhz);
        }
        fflush(stdout);
    }
    for (int g : {8192}) {
        printf("grid %d\n", g);
        fflush(stdout);
        RUN8(0, 0, 0, g);
#define RUNP(MW, blocks) check("two point sets in flight (prefetch distance 2), min waves/SIMD=" #MW, timeIt([&] { hipLaunchKernelGGL((labPipe<MW>), dim3(blocks), dim3(256), 0, 0, a, dx, n, dout); }, 10), blocks)
#define RUNP1(MW, blocks) check("look-ahead 2, one loop body, min waves/SIMD=" #MW, timeIt([&] { hipLaunchKernelGGL((labPipe1<MW>), dim3(blocks), dim3(256), 0, 0, a, dx, n, dout); }, 10), blocks)
        RUNP1(7, g);
        RUNP1(6, g);
        RUNP1(5, g);
        RUNP1(4, g);
        RUNP(4, g);
        RUN8(0, 0, 0, g);
        RUN8(1, 0, 0, g);
#define RUN8D(P, g) check("8-lane rows, points line by line, PTS=" #P, timeIt([&] { hipLaunchKernelGGL((lab8D<P>), dim3(g), dim3(256), 0, 0, a, dx, n, dout); }, 10), g)
        RUN8D(1, g);
        RUN8D(2, g);
        RUN8(0, 1, 0, g);
        RUN8(1, 1, 0, g);
#define RUN64(W, blocks) check("64-B rows + LDS tails, waves/WG=" #W, timeIt([&] { hipLaunchKernelGGL((lab64<W>), dim3(blocks), dim3(W * 64), (4096 + W * 4 * 66) * 16 + W * 64 * 4, 0, a64, dx, n, dout); }, 10), blocks)
#define RUN64T(MW, blocks) check("64-B rows + tail step (5 DMA, 1 wait), min waves/SIMD=" #MW, timeIt([&] { hipLaunchKernelGGL((lab64t<MW>), dim3(blocks), dim3(256), 0, 0, a64, dx, n, dout); }, 10), blocks)
        RUN64T(7, g);
        RUN64T(8, g);
        RUN64T(6, g);
        RUN64(16, g / 4);
        RUN64(16, 256);
#define RUN64P(W, blocks) check("64-B rows + LDS tails + points look-ahead, waves/WG=" #W, timeIt([&] { hipLaunchKernelGGL((lab64P<W>), dim3(blocks), dim3(W * 64), (4096 + W * 4 * 66) * 16 + W * 64 * 4, 0, a64, dx, n, dout); }, 10), blocks)
#define RUN64P2(W, blocks) RUN64P2A(W, 0, blocks)
#define RUN64P2A(W, AUX, blocks) check("64-B rows + LDS tails + two point sets in flight, one wait, waves/WG=" #W " aux=" #AUX, timeIt([&] { hipLaunchKernelGGL((lab64P2<W, AUX>), dim3(blocks), dim3(W * 64), (4096 + W * 4 * 66) * 16 + W * 64 * 4, 0, a64, dx, n, dout); }, 10), blocks)
        RUN64P2(16, 256);
        RUN64P2A(16, 100, 256);
        RUN64P2(16, 1024);
        RUN64P(16, 256);
        RUN64P(16, 512);
        RUN64P(16, 1024);
        RUN64P(16, 2048);
        RUN5(1, 0, g);
        RUN5(2, 0, g);
        printf("\n");
    }
    return 0;
}
