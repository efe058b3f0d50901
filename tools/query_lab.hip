// Development lab (not part of the product): times structural variants of the Query kernel on a synthetic
// uniform depth-4 / degree-2 tree with random points, to find what bounds it.  Same arithmetic as the
// product kernel (no FMA contraction), correctness is cross-checked between variants by a checksum.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/query_lab.hip -o /tmp/query_lab && /tmp/query_lab
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct alignas(128) Fat {
    uint32_t a, b, pad[2];
    double c[14];
};
struct Rec {
    uint32_t a, b;
};
struct Tree {
    const Fat* fat;
    const Rec* thin;
    const double* coeffs;  // 80-byte stride
    const double* coeffs128;  // 128-byte stride
};
__constant__ double kNl[3];   // NL[j][4]
__constant__ double kRec[6];

__device__ __forceinline__ void descend4(double px, double py, double pz, uint32_t& code, double& cx, double& cy, double& cz) {
    cx = cy = cz = 0.0;
    double q = 0.25;
    code = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const bool ux = px >= cx, uy = py >= cy, uz = pz >= cz;
        code = code * 8u + (ux ? 1u : 0u) + (uy ? 2u : 0u) + (uz ? 4u : 0u);
        cx = ux ? cx + q : cx - q;
        cy = uy ? cy + q : cy - q;
        cz = uz ? cz + q : cz - q;
        q = q * 0.5;
    }
}
__device__ __forceinline__ double eval2(const double* cv, double ux, double uy, double uz) {
    double tx[3], ty[3], tz[3];
    tx[0] = ty[0] = tz[0] = kNl[0];
    double xm2 = 0, xm1 = 1, ym2 = 0, ym1 = 1, zm2 = 0, zm1 = 1;
#pragma unroll
    for (int j = 1; j <= 2; ++j) {
        const double r0 = kRec[2 * j], r1 = kRec[2 * j + 1], nl = kNl[j];
        const double lx = r0 * ux * xm1 - r1 * xm2, ly = r0 * uy * ym1 - r1 * ym2, lz = r0 * uz * zm1 - r1 * zm2;
        xm2 = xm1, xm1 = lx, ym2 = ym1, ym1 = ly, zm2 = zm1, zm1 = lz;
        tx[j] = lx * nl, ty[j] = ly * nl, tz[j] = lz * nl;
    }
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

enum { V_STREAM = 0, V_STREAM_DESCEND, V_THIN, V_FAT, V_FAT_NT, V_THIN128, V_FAT_K2, V_FAT_K4, V_FAT_1PT, V_FAT_LDSLOAD, V_NOLOAD_FAT, V_COOP, V_COOP_DIRECT, V_FATGATHER_NOEVAL, V_EVAL_FIXEDLEAF, V_COOP_DMA, V_COOP_DMA_NOEVAL, V_COOP_DMA_HALF, V_COOP_DMA_K2, V_COOP_DMA_BIGLDS, V_COOP_DMA_QUARTER, V_COOP_DMA_EIGHTH, V_COOP_DMA_HALF_PF, V_COUNT };
const char* kNames[] = {"stream only (xyz in, 1 add, out)", "stream + descent arithmetic", "thin table + coeffs 80B stride",
                        "fat table (1 line/pt)", "fat + nontemporal stream", "thin table + coeffs 128B stride",
                        "fat, 2 pts/thread interleaved", "fat, 4 pts/thread interleaved", "fat, 1 pt/thread (no grid-stride)",
                        "fat, coalesced 16B loads via LDS", "fat, points synthesised (no HBM read)",
                        "coop: LDS point loads + 8-lane line fetch + LDS transpose", "coop line fetch, strided point loads",
                        "fat gather, no eval (sum coeffs)", "eval with fixed leaf (no gather)",
                        "coop line fetch by LDS-DMA (global_load_lds x4)", "coop LDS-DMA, no eval",
                        "coop LDS-DMA in two half passes (4 KB/wave)", "coop LDS-DMA, 2 tiles per iteration", "coop LDS-DMA + 32 KB dummy LDS",
                        "coop LDS-DMA in four passes (2 KB/wave)", "coop LDS-DMA in eight passes (1 KB/wave)",
                        "half passes + next tile's points prefetched (asm loads, vmcnt(3))"};

template <int V>
__device__ __forceinline__ double onePoint(const Tree& t, double px, double py, double pz) {
    if (V == V_STREAM) return px + py + pz;
    uint32_t code;
    double cx, cy, cz;
    descend4(px, py, pz, code, cx, cy, cz);
    const double ux = (px - cx) * 32.0, uy = (py - cy) * 32.0, uz = (pz - cz) * 32.0;
    if (V == V_STREAM_DESCEND) return ux + uy + uz + (double)code;
    if (V == V_EVAL_FIXEDLEAF) {
        double cv[10];
        const double2* c2 = reinterpret_cast<const double2*>(t.fat[code & 7].c);
#pragma unroll
        for (int i = 0; i < 5; ++i) { const double2 v = c2[i]; cv[2 * i] = v.x, cv[2 * i + 1] = v.y; }
        return eval2(cv, ux, uy, uz);
    }
    if (V == V_FATGATHER_NOEVAL) {
        const double2* c2 = reinterpret_cast<const double2*>(t.fat[code].c);
        double a = ux;
#pragma unroll
        for (int i = 0; i < 5; ++i) { const double2 v = c2[i]; a += v.x + v.y; }
        return a;
    }
    double cv[10];
    if (V == V_THIN || V == V_THIN128) {
        const Rec r = t.thin[code];
        const double2* c2 = reinterpret_cast<const double2*>((V == V_THIN ? t.coeffs + (size_t)r.a * 10 : t.coeffs128 + (size_t)r.a * 16));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double2 v = c2[i];
            cv[2 * i] = v.x, cv[2 * i + 1] = v.y;
        }
        return eval2(cv, ux, uy, uz) + (r.b == 99u ? 1.0 : 0.0);
    }
    const Fat* e = t.fat + code;
    const uint2 h = *reinterpret_cast<const uint2*>(e);
    const double2* c2 = reinterpret_cast<const double2*>(e->c);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double2 v = c2[i];
        cv[2 * i] = v.x, cv[2 * i + 1] = v.y;
    }
    return eval2(cv, ux, uy, uz) + (h.y == 99u ? 1.0 : 0.0);
}

template <int V>
__global__ __launch_bounds__(256) void lab(Tree t, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (V == V_FAT_K2 || V == V_FAT_K4) {
        constexpr int K = V == V_FAT_K2 ? 2 : 4;
        for (size_t i = i0; i < n; i += stride * K) {
            double x[K], y[K], z[K], r[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t j = i + k * stride;
                const bool ok = j < n;
                x[k] = ok ? xyz[3 * j] : 0.0, y[k] = ok ? xyz[3 * j + 1] : 0.0, z[k] = ok ? xyz[3 * j + 2] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) r[k] = onePoint<V_FAT>(t, x[k], y[k], z[k]);
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (i + k * stride < n) out[i + k * stride] = r[k];
        }
        return;
    }
    if (V == V_COOP || V == V_COOP_DIRECT) {
        __shared__ double sp[256 * 3];
        __shared__ double2 sl[4][64][8];  // per wave: 64 points x 8 chunks
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
            double px, py, pz;
            if (V == V_COOP) {
                const size_t cnt = (n - base < 256 ? n - base : 256) * 3;
                const double2* src = reinterpret_cast<const double2*>(xyz + base * 3);
                for (int q = threadIdx.x; q < (int)(cnt / 2); q += 256) reinterpret_cast<double2*>(sp)[q] = src[q];
                __syncthreads();
                px = sp[3 * threadIdx.x], py = sp[3 * threadIdx.x + 1], pz = sp[3 * threadIdx.x + 2];
            } else {
                const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
                px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            }
            uint32_t code;
            double cx, cy, cz;
            descend4(px, py, pz, code, cx, cy, cz);
            // 8 lanes fetch one point's 128-byte line per instruction: 8 lines per wave-instruction instead of 64
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t ck = __shfl(code, grp | k, 64);
                const double2 v = reinterpret_cast<const double2*>(t.fat + ck)[j];
                sl[w][grp | k][j] = v;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            const uint2 h = *reinterpret_cast<const uint2*>(&sl[w][lane][0]);
            double cv[10];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double2 v = sl[w][lane][1 + i];
                cv[2 * i] = v.x, cv[2 * i + 1] = v.y;
            }
            const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0) + (h.y == 99u ? 1.0 : 0.0);
            if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
            __syncthreads();
        }
        return;
    }
    if (V == V_COOP_DMA || V == V_COOP_DMA_NOEVAL) {
        __shared__ double2 sd[4][8][64];  // per wave: instruction k -> 64 lanes x 16 B
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            uint32_t code;
            double cx, cy, cz;
            descend4(px, py, pz, code, cx, cy, cz);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t ck = __shfl(code, grp | k, 64);
                const char* src = reinterpret_cast<const char*>(t.fat + ck) + j * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)&sd[w][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // lane (g, k=j) owns the row written by instruction k at lanes g*8..g*8+7
            const double2* row = &sd[w][j][grp];
            const uint2 h = *reinterpret_cast<const uint2*>(row);
            double cv[10];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const double2 v = row[1 + q];
                cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
            }
            double r;
            if (V == V_COOP_DMA_NOEVAL) {
                r = px - cx;
                for (int q = 0; q < 10; ++q) r += cv[q];
            } else {
                r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0) + (h.y == 99u ? 1.0 : 0.0);
            }
            if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if (V == V_COOP_DMA_BIGLDS) {
        __shared__ double2 sd[4][8][64];
        __shared__ double dummy[4096];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        if (n == 1) dummy[threadIdx.x] = 1.0;
        for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            uint32_t code;
            double cx, cy, cz;
            descend4(px, py, pz, code, cx, cy, cz);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t ck = __shfl(code, grp | k, 64);
                const char* src = reinterpret_cast<const char*>(t.fat + ck) + j * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)&sd[w][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const double2* row = &sd[w][j][grp];
            double cv[10];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const double2 v = row[1 + q];
                cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
            }
            const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0) + (n == 1 ? dummy[lane] : 0.0);
            if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if (V == V_COOP_DMA_HALF) {
        __shared__ double2 sd[4][4][64];  // 4 KB per wave
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            uint32_t code;
            double cx, cy, cz;
            descend4(px, py, pz, code, cx, cy, cz);
            double cv[10];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t ck = __shfl(code, grp | (half * 4 + k), 64);
                    const char* src = reinterpret_cast<const char*>(t.fat + ck) + j * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sd[w][k][0], 16, 0, 0);
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
            const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0);
            if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
        }
        return;
    }
    if (V == V_COOP_DMA_QUARTER || V == V_COOP_DMA_EIGHTH) {
        constexpr int S = V == V_COOP_DMA_QUARTER ? 2 : 1;  // steps per pass
        __shared__ double2 sd[4][S][64];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            uint32_t code;
            double cx, cy, cz;
            descend4(px, py, pz, code, cx, cy, cz);
            double cv[10];
#pragma unroll
            for (int pass = 0; pass < 8 / S; ++pass) {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const uint32_t ck = __shfl(code, grp | (pass * S + k), 64);
                    const char* src = reinterpret_cast<const char*>(t.fat + ck) + j * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sd[w][k][0], 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if ((j / S) == pass) {
                    const double2* row = &sd[w][j % S][grp];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const double2 v = row[1 + q];
                        cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0);
            if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
        }
        return;
    }
    if (V == V_COOP_DMA_HALF_PF) {
        __shared__ double2 sd[4][4][64];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        const size_t step = (size_t)gridDim.x * 256;
        size_t base = (size_t)blockIdx.x * 256;
        if (base >= n) return;
        double nx, ny, nz;
        {
            const size_t i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
            nx = xyz[3 * i], ny = xyz[3 * i + 1], nz = xyz[3 * i + 2];
        }
        for (; base < n; base += step) {
            const double px = nx, py = ny, pz = nz;
            uint32_t code;
            double cx, cy, cz;
            descend4(px, py, pz, code, cx, cy, cz);
            double cv[10];
            const size_t nb = base + step;
            const size_t ni = (nb + threadIdx.x < n) ? nb + threadIdx.x : n - 1;
            const double* np = xyz + 3 * ni;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t ck = __shfl(code, grp | (half * 4 + k), 64);
                    const char* src = reinterpret_cast<const char*>(t.fat + ck) + j * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sd[w][k][0], 16, 0, 0);
                }
                if (half == 0) {
                    // next tile's points: issued behind the first four DMAs, allowed to stay in flight across their wait
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
            const double r = eval2(cv, (px - cx) * 32.0, (py - cy) * 32.0, (pz - cz) * 32.0);
            if (base + threadIdx.x < n) out[base + threadIdx.x] = r;
        }
        return;
    }
    if (V == V_COOP_DMA_K2) {
        __shared__ double2 sd[4][2][8][64];  // two tiles in flight per wave
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, grp = lane & ~7, j = lane & 7;
        for (size_t base = (size_t)blockIdx.x * 512; base < n; base += (size_t)gridDim.x * 512) {
            double px[2], py[2], pz[2], cx[2], cy[2], cz[2];
            uint32_t code[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t i0 = base + h * 256 + threadIdx.x;
                const size_t i = i0 < n ? i0 : n - 1;
                px[h] = xyz[3 * i], py[h] = xyz[3 * i + 1], pz[h] = xyz[3 * i + 2];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                descend4(px[h], py[h], pz[h], code[h], cx[h], cy[h], cz[h]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t ck = __shfl(code[h], grp | k, 64);
                    const char* src = reinterpret_cast<const char*>(t.fat + ck) + j * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sd[w][h][k][0], 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2* row = &sd[w][h][j][grp];
                double cv[10];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const double2 v = row[1 + q];
                    cv[2 * q] = v.x, cv[2 * q + 1] = v.y;
                }
                const double r = eval2(cv, (px[h] - cx[h]) * 32.0, (py[h] - cy[h]) * 32.0, (pz[h] - cz[h]) * 32.0);
                const size_t i0 = base + h * 256 + threadIdx.x;
                if (i0 < n) out[i0] = r;
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if (V == V_FAT_LDSLOAD) {
        __shared__ double sp[256 * 3];
        for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
            const size_t cnt = (n - base < 256 ? n - base : 256) * 3;  // doubles in this tile
            const double2* src = reinterpret_cast<const double2*>(xyz + base * 3);
            for (int j = threadIdx.x; j < (int)(cnt / 2); j += 256) reinterpret_cast<double2*>(sp)[j] = src[j];
            __syncthreads();
            if (base + threadIdx.x < n)
                out[base + threadIdx.x] = onePoint<V_FAT>(t, sp[3 * threadIdx.x], sp[3 * threadIdx.x + 1], sp[3 * threadIdx.x + 2]);
            __syncthreads();
        }
        return;
    }
    for (size_t i = i0; i < n; i += stride) {
        double x, y, z;
        if (V == V_NOLOAD_FAT) {
            uint64_t s = i * 0x9E3779B97F4A7C15ull;
            s ^= s >> 29;
            x = (double)(s & 0xFFFFF) / 1048576.0 - 0.5, y = (double)((s >> 20) & 0xFFFFF) / 1048576.0 - 0.5,
            z = (double)((s >> 40) & 0xFFFFF) / 1048576.0 - 0.5;
            out[i] = onePoint<V_FAT>(t, x, y, z);
        } else if (V == V_FAT_NT) {
            x = __builtin_nontemporal_load(xyz + 3 * i), y = __builtin_nontemporal_load(xyz + 3 * i + 1), z = __builtin_nontemporal_load(xyz + 3 * i + 2);
            __builtin_nontemporal_store(onePoint<V_FAT>(t, x, y, z), out + i);
        } else {
            x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            out[i] = onePoint<(V == V_FAT_1PT ? V_FAT : V)>(t, x, y, z);
        }
    }
}

template <int V>
float run(const Tree& t, const double* dx, size_t n, double* dout, int blocks, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(lab<V>, dim3(blocks), dim3(256), 0, 0, t, dx, n, dout);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(lab<V>, dim3(blocks), dim3(256), 0, 0, t, dx, n, dout);
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
    std::vector<Fat> fat(4096);
    std::vector<Rec> thin(4096);
    std::vector<double> co(4096 * 10), co128(4096 * 16, 0.0);
    for (int i = 0; i < 4096; ++i) {
        fat[i].a = i, fat[i].b = 2;
        thin[i].a = i, thin[i].b = 2;
        for (int k = 0; k < 10; ++k) {
            const double c = std::sin(i * 0.37 + k);
            fat[i].c[k] = c, co[i * 10 + k] = c, co128[i * 16 + k] = c;
        }
    }
    double nl[3], rec[6] = {0, 0, 1.0, 0.0, 1.5, 0.5};
    for (int j = 0; j < 3; ++j) nl[j] = std::sqrt((2.0 * j + 1.0) * 16.0);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(kNl), nl, sizeof nl));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(kRec), rec, sizeof rec));
    double *dx, *dout, *dco, *dco128;
    Fat* dfat;
    Rec* dthin;
    CK(hipMalloc(&dx, pts.size() * 8));
    CK(hipMalloc(&dout, n * 8));
    CK(hipMalloc(&dfat, fat.size() * sizeof(Fat)));
    CK(hipMalloc(&dthin, thin.size() * sizeof(Rec)));
    CK(hipMalloc(&dco, co.size() * 8));
    CK(hipMalloc(&dco128, co128.size() * 8));
    CK(hipMemcpy(dx, pts.data(), pts.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfat, fat.data(), fat.size() * sizeof(Fat), hipMemcpyHostToDevice));
    CK(hipMemcpy(dthin, thin.data(), thin.size() * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(dco, co.data(), co.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dco128, co128.data(), co128.size() * 8, hipMemcpyHostToDevice));
    Tree t{dfat, dthin, dco, dco128};
    std::vector<double> h(n);
    auto checksum = [&]() {
        CK(hipMemcpy(h.data(), dout, n * 8, hipMemcpyDeviceToHost));
        double a = 0;
        for (size_t i = 0; i < n; i += 997) a += h[i];
        return a;
    };
    const int full = (int)((n + 255) / 256);
    const int grids[] = {2048, 4096, 8192, 16384};
#define RUN(V, blocks)                                                                                         \
    {                                                                                                          \
        float ms = run<V>(t, dx, n, dout, blocks, 10);                                                         \
        printf("%-42s grid %6d : %8.3f ms  %7.1f GB/s alg  checksum %.9g\n", kNames[V], blocks, ms, 32.0 * n / ms / 1e6, checksum()); \
    }
    for (int g : grids) {
        RUN(V_STREAM, g);
        RUN(V_STREAM_DESCEND, g);
        RUN(V_THIN, g);
        RUN(V_THIN128, g);
        RUN(V_FAT, g);
        RUN(V_FAT_NT, g);
        RUN(V_FAT_K2, g);
        RUN(V_FAT_K4, g);
        RUN(V_FAT_LDSLOAD, g);
        RUN(V_NOLOAD_FAT, g);
        RUN(V_COOP, g);
        RUN(V_COOP_DIRECT, g);
        RUN(V_FATGATHER_NOEVAL, g);
        RUN(V_EVAL_FIXEDLEAF, g);
        RUN(V_COOP_DMA, g);
        RUN(V_COOP_DMA_NOEVAL, g);
        RUN(V_COOP_DMA_HALF, g);
        RUN(V_COOP_DMA_K2, g);
        RUN(V_COOP_DMA_BIGLDS, g);
        RUN(V_COOP_DMA_QUARTER, g);
        RUN(V_COOP_DMA_EIGHTH, g);
        RUN(V_COOP_DMA_HALF_PF, g);
        printf("\n");
    }
    return 0;
}
