// Lab kernel: how fast can ONE wave add 4096 doubles one after the other (the running total of Octree.cpp:253-290 must be
// added up in the reference's order, so the chain of dependent v_add_f64 is the floor) -- and how to feed it its operands.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/chain_lab.hip -o tools/_bin/chain_lab && tools/_bin/chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 4096;

template <int K>
__device__ __forceinline__ double bcast32(double v) {  // two 32-bit DPP moves
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define BC64(K)                                                                                               \
    template <>                                                                                               \
    __device__ __forceinline__ double bcast64<K>(double v) {                                                  \
        double r;                                                                                             \
        asm("v_mov_b64_dpp %0, %1 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));     \
        return r;                                                                                             \
    }
template <int K>
__device__ __forceinline__ double bcast64(double v);
BC64(0) BC64(1) BC64(2) BC64(3) BC64(4) BC64(5) BC64(6) BC64(7) BC64(8) BC64(9) BC64(10) BC64(11) BC64(12) BC64(13) BC64(14) BC64(15)

template <int K>
__device__ __forceinline__ double viaSgpr(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), K), hi = __builtin_amdgcn_readlane(__double2hiint(v), K);
    return __hiloint2double(hi, lo);
}


// The adder's inner loop by hand: `iters` times 32 operands from LDS address `addr` (16-byte aligned), batch A = v[8:39], batch B =
// v[40:71]; the reads of one batch are in flight while the other is added (the compiler turns the same source into "read 32, add 32").
// Reads 16 operands past the last batch (they are not added).
__device__ __forceinline__ double chainAsm(double total, unsigned addr, unsigned iters) {
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b128 v[8:11], %[a]\n ds_read_b128 v[12:15], %[a] offset:16\n ds_read_b128 v[16:19], %[a] offset:32\n ds_read_b128 v[20:23], %[a] offset:48\n"
        "ds_read_b128 v[24:27], %[a] offset:64\n ds_read_b128 v[28:31], %[a] offset:80\n ds_read_b128 v[32:35], %[a] offset:96\n ds_read_b128 v[36:39], %[a] offset:112\n"
        "1:\n"
        "ds_read_b128 v[40:43], %[a] offset:128\n ds_read_b128 v[44:47], %[a] offset:144\n ds_read_b128 v[48:51], %[a] offset:160\n ds_read_b128 v[52:55], %[a] offset:176\n"
        "ds_read_b128 v[56:59], %[a] offset:192\n ds_read_b128 v[60:63], %[a] offset:208\n ds_read_b128 v[64:67], %[a] offset:224\n ds_read_b128 v[68:71], %[a] offset:240\n"
        "s_waitcnt lgkmcnt(8)\n"
        "v_add_f64 %[t], %[t], v[8:9]\n v_add_f64 %[t], %[t], v[10:11]\n v_add_f64 %[t], %[t], v[12:13]\n v_add_f64 %[t], %[t], v[14:15]\n"
        "v_add_f64 %[t], %[t], v[16:17]\n v_add_f64 %[t], %[t], v[18:19]\n v_add_f64 %[t], %[t], v[20:21]\n v_add_f64 %[t], %[t], v[22:23]\n"
        "v_add_f64 %[t], %[t], v[24:25]\n v_add_f64 %[t], %[t], v[26:27]\n v_add_f64 %[t], %[t], v[28:29]\n v_add_f64 %[t], %[t], v[30:31]\n"
        "v_add_f64 %[t], %[t], v[32:33]\n v_add_f64 %[t], %[t], v[34:35]\n v_add_f64 %[t], %[t], v[36:37]\n v_add_f64 %[t], %[t], v[38:39]\n"
        "ds_read_b128 v[8:11], %[a] offset:256\n ds_read_b128 v[12:15], %[a] offset:272\n ds_read_b128 v[16:19], %[a] offset:288\n ds_read_b128 v[20:23], %[a] offset:304\n"
        "ds_read_b128 v[24:27], %[a] offset:320\n ds_read_b128 v[28:31], %[a] offset:336\n ds_read_b128 v[32:35], %[a] offset:352\n ds_read_b128 v[36:39], %[a] offset:368\n"
        "s_waitcnt lgkmcnt(8)\n"
        "v_add_f64 %[t], %[t], v[40:41]\n v_add_f64 %[t], %[t], v[42:43]\n v_add_f64 %[t], %[t], v[44:45]\n v_add_f64 %[t], %[t], v[46:47]\n"
        "v_add_f64 %[t], %[t], v[48:49]\n v_add_f64 %[t], %[t], v[50:51]\n v_add_f64 %[t], %[t], v[52:53]\n v_add_f64 %[t], %[t], v[54:55]\n"
        "v_add_f64 %[t], %[t], v[56:57]\n v_add_f64 %[t], %[t], v[58:59]\n v_add_f64 %[t], %[t], v[60:61]\n v_add_f64 %[t], %[t], v[62:63]\n"
        "v_add_f64 %[t], %[t], v[64:65]\n v_add_f64 %[t], %[t], v[66:67]\n v_add_f64 %[t], %[t], v[68:69]\n v_add_f64 %[t], %[t], v[70:71]\n"
        "v_add_u32 %[a], 0x100, %[a]\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [t] "+v"(total), [a] "+v"(addr), [n] "+s"(iters)
        :
        : "memory", "scc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26",
          "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
          "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",
          "v69", "v70", "v71");
    return total;
}

// The same loop with the running total alternating between two register pairs of opposite bank parity: v[74:75] (banks 2, 3) is added
// to the operands in banks 0, 1 (v[8:9], v[12:13], ...) into v[72:73] (banks 0, 1), which is added to the operands in banks 2, 3 -- no
// v_add_f64 reads two sources from the same banks.
__device__ __forceinline__ double chainAsm2(double total, unsigned addr, unsigned iters) {
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n v_mov_b64 v[74:75], %[t]\n"
        "ds_read_b128 v[8:11], %[a] offset:0\n ds_read_b128 v[12:15], %[a] offset:16\n ds_read_b128 v[16:19], %[a] offset:32\n ds_read_b128 v[20:23], %[a] offset:48\n ds_read_b128 v[24:27], %[a] offset:64\n ds_read_b128 v[28:31], %[a] offset:80\n ds_read_b128 v[32:35], %[a] offset:96\n ds_read_b128 v[36:39], %[a] offset:112\n"
        "1:\n"
        "ds_read_b128 v[40:43], %[a] offset:128\n ds_read_b128 v[44:47], %[a] offset:144\n ds_read_b128 v[48:51], %[a] offset:160\n ds_read_b128 v[52:55], %[a] offset:176\n ds_read_b128 v[56:59], %[a] offset:192\n ds_read_b128 v[60:63], %[a] offset:208\n ds_read_b128 v[64:67], %[a] offset:224\n ds_read_b128 v[68:71], %[a] offset:240\n"
        "s_waitcnt lgkmcnt(8)\n"
        "v_add_f64 v[72:73], v[74:75], v[8:9]\n v_add_f64 v[74:75], v[72:73], v[10:11]\n v_add_f64 v[72:73], v[74:75], v[12:13]\n v_add_f64 v[74:75], v[72:73], v[14:15]\n v_add_f64 v[72:73], v[74:75], v[16:17]\n v_add_f64 v[74:75], v[72:73], v[18:19]\n v_add_f64 v[72:73], v[74:75], v[20:21]\n v_add_f64 v[74:75], v[72:73], v[22:23]\n v_add_f64 v[72:73], v[74:75], v[24:25]\n v_add_f64 v[74:75], v[72:73], v[26:27]\n v_add_f64 v[72:73], v[74:75], v[28:29]\n v_add_f64 v[74:75], v[72:73], v[30:31]\n v_add_f64 v[72:73], v[74:75], v[32:33]\n v_add_f64 v[74:75], v[72:73], v[34:35]\n v_add_f64 v[72:73], v[74:75], v[36:37]\n v_add_f64 v[74:75], v[72:73], v[38:39]\n"
        "ds_read_b128 v[8:11], %[a] offset:256\n ds_read_b128 v[12:15], %[a] offset:272\n ds_read_b128 v[16:19], %[a] offset:288\n ds_read_b128 v[20:23], %[a] offset:304\n ds_read_b128 v[24:27], %[a] offset:320\n ds_read_b128 v[28:31], %[a] offset:336\n ds_read_b128 v[32:35], %[a] offset:352\n ds_read_b128 v[36:39], %[a] offset:368\n"
        "s_waitcnt lgkmcnt(8)\n"
        "v_add_f64 v[72:73], v[74:75], v[40:41]\n v_add_f64 v[74:75], v[72:73], v[42:43]\n v_add_f64 v[72:73], v[74:75], v[44:45]\n v_add_f64 v[74:75], v[72:73], v[46:47]\n v_add_f64 v[72:73], v[74:75], v[48:49]\n v_add_f64 v[74:75], v[72:73], v[50:51]\n v_add_f64 v[72:73], v[74:75], v[52:53]\n v_add_f64 v[74:75], v[72:73], v[54:55]\n v_add_f64 v[72:73], v[74:75], v[56:57]\n v_add_f64 v[74:75], v[72:73], v[58:59]\n v_add_f64 v[72:73], v[74:75], v[60:61]\n v_add_f64 v[74:75], v[72:73], v[62:63]\n v_add_f64 v[72:73], v[74:75], v[64:65]\n v_add_f64 v[74:75], v[72:73], v[66:67]\n v_add_f64 v[72:73], v[74:75], v[68:69]\n v_add_f64 v[74:75], v[72:73], v[70:71]\n"
        "v_add_u32 %[a], 0x100, %[a]\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n v_mov_b64 %[t], v[74:75]\n"
        : [t] "+v"(total), [a] "+v"(addr), [n] "+s"(iters)
        :
        : "memory", "scc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    return total;
}

// Operands through the scalar cache: s_load_dwordx16 brings eight doubles into SGPRs, v_add_f64 takes them as its second source -- the
// adder's instruction stream holds the additions and one scalar load per eight of them.  Batches of 16 (two loads): A = s[36:67], B = s[68:99].
__device__ __forceinline__ double chainAsmScalar(double total, const double* src, unsigned iters) {
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n s_dcache_inv\n s_mov_b64 s[34:35], %[p]\n"
        "s_load_dwordx16 s[36:51], s[34:35], 0x0\n s_load_dwordx16 s[52:67], s[34:35], 0x40\n"
        "1:\n"
        "s_waitcnt lgkmcnt(0)\n"   // (scalar loads may return out of order: wait for all, THEN ask for the next batch)
        "s_load_dwordx16 s[68:83], s[34:35], 0x80\n s_load_dwordx16 s[84:99], s[34:35], 0xc0\n"
        "v_add_f64 %[t], %[t], s[36:37]\n v_add_f64 %[t], %[t], s[38:39]\n v_add_f64 %[t], %[t], s[40:41]\n v_add_f64 %[t], %[t], s[42:43]\n v_add_f64 %[t], %[t], s[44:45]\n v_add_f64 %[t], %[t], s[46:47]\n v_add_f64 %[t], %[t], s[48:49]\n v_add_f64 %[t], %[t], s[50:51]\n v_add_f64 %[t], %[t], s[52:53]\n v_add_f64 %[t], %[t], s[54:55]\n v_add_f64 %[t], %[t], s[56:57]\n v_add_f64 %[t], %[t], s[58:59]\n v_add_f64 %[t], %[t], s[60:61]\n v_add_f64 %[t], %[t], s[62:63]\n v_add_f64 %[t], %[t], s[64:65]\n v_add_f64 %[t], %[t], s[66:67]\n"
        "s_add_u32 s34, s34, 0x100\n s_addc_u32 s35, s35, 0\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_load_dwordx16 s[36:51], s[34:35], 0x0\n s_load_dwordx16 s[52:67], s[34:35], 0x40\n"
        "v_add_f64 %[t], %[t], s[68:69]\n v_add_f64 %[t], %[t], s[70:71]\n v_add_f64 %[t], %[t], s[72:73]\n v_add_f64 %[t], %[t], s[74:75]\n v_add_f64 %[t], %[t], s[76:77]\n v_add_f64 %[t], %[t], s[78:79]\n v_add_f64 %[t], %[t], s[80:81]\n v_add_f64 %[t], %[t], s[82:83]\n v_add_f64 %[t], %[t], s[84:85]\n v_add_f64 %[t], %[t], s[86:87]\n v_add_f64 %[t], %[t], s[88:89]\n v_add_f64 %[t], %[t], s[90:91]\n v_add_f64 %[t], %[t], s[92:93]\n v_add_f64 %[t], %[t], s[94:95]\n v_add_f64 %[t], %[t], s[96:97]\n v_add_f64 %[t], %[t], s[98:99]\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [t] "+v"(total), [n] "+s"(iters)
        : [p] "s"(src)
        : "memory", "scc", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    return total;
}

template <int V>
__global__ __launch_bounds__(64) void chain(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ ticks) {
    __shared__ __attribute__((aligned(16))) double s[N + 64];
    for (int i = threadIdx.x; i < N + 64; i += 64) s[i] = i < N ? src[i] : 0.0;
    __syncthreads();
    double total = 409600.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (V == 0) {  // as frontier.hip has it: sixteen LDS reads (every lane the same address), sixteen additions
        for (int q = 0; q < N; q += 16) {
            double o[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) o[k] = s[q + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + o[k];
        }
    } else if (V == 5) {  // two register batches of 16: the next batch's LDS reads are in flight while this one is added
        double a[16], b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = s[k];
        for (int q = 0; q < N; q += 32) {
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = s[q + 16 + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + a[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = s[q + 32 + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + b[k];
        }
    } else if (V == 6) {  // the same with batches of 8
        double a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = s[k];
        for (int q = 0; q < N; q += 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = s[q + 8 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) total = total + a[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = s[q + 16 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) total = total + b[k];
        }
    } else if (V == 7) {  // operands in scalar registers: s_load from global memory (no vector instruction but the addition)
        const double* g = src;
        for (int q = 0; q < N; q += 16) {
            double o[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) o[k] = __builtin_nontemporal_load(g + q + k);
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + o[k];
        }
    } else if (V == 8) {  // ping-pong with the instruction order pinned: reads of one batch, additions of the other
        double a[16], b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = s[k];
        for (int q = 0; q < N; q += 32) {
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = s[q + 16 + k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + a[k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = s[q + 32 + k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + b[k];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (V == 9) {  // the hand-scheduled loop
        total = chainAsm(total, (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)s, N / 32);
    } else if (V == 10) {  // ... with the total alternating between register banks
        total = chainAsm2(total, (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)s, N / 32);
    } else if (V == 11) {  // one active lane (the other 63 only repeat its work)
        if (threadIdx.x == 0) {
            for (int q = 0; q < N; q += 16) {
                double o[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) o[k] = s[q + k];
#pragma unroll
                for (int k = 0; k < 16; ++k) total = total + o[k];
            }
        }
        total = __shfl(total, 0, 64);
    } else if (V == 12) {  // operands in scalar registers, from global memory (src holds 64 doubles beyond N: see main)
        total = chainAsmScalar(total, src, N / 32);
    } else if (V == 4) {  // the floor: nothing but the dependent additions
        const double c = s[threadIdx.x & 1];
        for (int q = 0; q < N; q += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) total = total + c;
        }
    } else {
        const int l = threadIdx.x & 15;
        double cur = s[l];
        for (int q = 0; q < N; q += 16) {
            const double nxt = s[q + 16 + l];
#define STEP(K)                                                                                      \
    total = total + (V == 1 ? bcast32<K>(cur) : V == 2 ? bcast64<K>(cur) : viaSgpr<K>(cur));
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
            cur = nxt;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = total, ticks[0] = t1 - t0;
    if (threadIdx.x == 37) out[1] = total;
}

int main() {
    std::vector<double> h(N);
    unsigned long long z = 88172645463325252ull;
    double ref = 409600.0;
    for (int i = 0; i < N; ++i) {
        z ^= z << 13, z ^= z >> 7, z ^= z << 17;
        h[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 1e-5 - 100.0;
        ref = ref + h[i];
    }
    double *dSrc, *dOut;
    unsigned long long* dT;
    hipMalloc(&dSrc, (N + 64) * 8), hipMalloc(&dOut, 16), hipMalloc(&dT, 8);
    hipMemcpy(dSrc, h.data(), N * 8, hipMemcpyHostToDevice);
    const char* names[] = {"16 LDS reads + 16 adds (current)", "2 x v_mov_b32_dpp row_newbcast + add", "v_mov_b64_dpp row_newbcast + add",
                           "2 x v_readlane -> SGPR operand + add", "dependent adds alone (floor)", "two batches of 16 in registers (ping-pong)", "two batches of 8 in registers", "operands straight from global memory (uniform address)", "ping-pong, order pinned by sched_barrier", "ping-pong by hand (inline asm)", "... and the total alternating between register banks", "one active lane", "operands in SGPRs (s_load_dwordx16)"};
    for (int v = 0; v < 13; ++v) {
        double best = 1e30, got[2] = {0, 0};
        unsigned long long tk = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEvent_t a, b;
            hipEventCreate(&a), hipEventCreate(&b);
            hipEventRecord(a, 0);
            switch (v) {
                case 0: hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 1: hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 2: hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 3: hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 4: hipLaunchKernelGGL(chain<4>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 5: hipLaunchKernelGGL(chain<5>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 6: hipLaunchKernelGGL(chain<6>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 7: hipLaunchKernelGGL(chain<7>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 8: hipLaunchKernelGGL(chain<8>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 9: hipLaunchKernelGGL(chain<9>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 10: hipLaunchKernelGGL(chain<10>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                case 11: hipLaunchKernelGGL(chain<11>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
                default: hipLaunchKernelGGL(chain<12>, dim3(1), dim3(64), 0, 0, dSrc, dOut, dT); break;
            }
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
            hipMemcpy(got, dOut, 16, hipMemcpyDeviceToHost);
            hipMemcpy(&tk, dT, 8, hipMemcpyDeviceToHost);
        }
        std::printf("%-40s kernel %.1f us, chain %llu ticks of the cycle counter = %.2f per addition; result %s (lane 37 %s)\n", names[v], best * 1e3, tk,
                    (double)tk / N, v == 4 ? "-" : (got[0] == ref ? "== sequential sum" : "DIFFERS"), v == 4 ? "-" : (got[1] == ref ? "same" : "DIFFERS"));
    }
    extern int secondExperimentRun(const double*, double*, unsigned long long*, double);
    return secondExperimentRun(dSrc, dOut, dT, ref);
}

// ---- second experiment: the adder inside a workgroup shaped like fr_round_kernel's last one
template <int THREADS, int MODE>  // MODE 0: loaders preload everything, then idle at the barriers; 1: loaders fetch chunk by chunk from global memory
__global__ __launch_bounds__(THREADS) void chainWg(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ ticks, int bigLds) {
    __shared__ double sOps[2 * 2048];
    __shared__ double sPad[12288];  // 96 KB more, touched only when asked (the real kernel's 120 KB)
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    if (bigLds && tid == 5) sPad[tid] = 1.0;
    if (wave != 0 && (wave & 3u) == 0) return;
    const bool adder = wave == 0;
    const unsigned nLoad = THREADS == 256 ? 192u : 768u;
    const unsigned ltid = adder ? 0u : (THREADS == 256 ? tid - 64u : (wave - 1u - (wave >> 2)) * 64u + (tid & 63u));
    double total = 409600.0;
    const unsigned nOps = N;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    auto loadChunk = [&](unsigned first, double* dst) {
        for (unsigned k = ltid; k < 2048u && first + k < nOps; k += nLoad) dst[k] = src[first + k];
    };
    if (!adder) loadChunk(0, sOps);
    __syncthreads();
    for (unsigned c0 = 0, half = 0; c0 < nOps; c0 += 2048, half ^= 1u) {
        const unsigned n = nOps - c0 < 2048u ? nOps - c0 : 2048u;
        if (!adder) {
            if (MODE == 1 && c0 + 2048u < nOps) loadChunk(c0 + 2048u, sOps + (half ^ 1u) * 2048u);
            if (MODE == 0 && c0 == 0) loadChunk(2048u, sOps + 2048u);
        } else {
            const double* s = sOps + half * 2048u;
            unsigned q = 0;
            if (n >= 32) {
                double a[16], b[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = s[k];
                for (; q + 32 <= n; q += 32) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) b[k] = s[q + 16 + k];
#pragma unroll
                    for (int k = 0; k < 16; ++k) total = total + a[k];
#pragma unroll
                    for (int k = 0; k < 16; ++k) a[k] = s[(q + 32 + k) & 4095];
#pragma unroll
                    for (int k = 0; k < 16; ++k) total = total + b[k];
                }
            }
            for (; q < n; ++q) total = total + s[q];
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) out[0] = total, ticks[0] = t1 - t0;
}
int secondExperimentRun(const double* dSrc, double* dOut, unsigned long long* dT, double ref) {
    for (int v = 0; v < 6; ++v) {
        unsigned long long tk = 0;
        double got = 0;
        for (int rep = 0; rep < 3; ++rep) {
            switch (v) {
                case 0: hipLaunchKernelGGL((chainWg<256, 0>), dim3(1), dim3(256), 0, 0, dSrc, dOut, dT, 0); break;
                case 1: hipLaunchKernelGGL((chainWg<256, 1>), dim3(1), dim3(256), 0, 0, dSrc, dOut, dT, 0); break;
                case 2: hipLaunchKernelGGL((chainWg<1024, 0>), dim3(1), dim3(1024), 0, 0, dSrc, dOut, dT, 0); break;
                case 3: hipLaunchKernelGGL((chainWg<1024, 1>), dim3(1), dim3(1024), 0, 0, dSrc, dOut, dT, 0); break;
                case 4: hipLaunchKernelGGL((chainWg<1024, 1>), dim3(1), dim3(1024), 0, 0, dSrc, dOut, dT, 1); break;
                default: hipLaunchKernelGGL((chainWg<1024, 1>), dim3(34), dim3(1024), 0, 0, dSrc, dOut, dT, 1); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(&got, dOut, 8, hipMemcpyDeviceToHost);
            hipMemcpy(&tk, dT, 8, hipMemcpyDeviceToHost);
        }
        const char* names[] = {"256 threads, operands preloaded", "256 threads, 3 loader waves", "1024 threads (waves 4, 8, 12 gone), preloaded",
                               "1024 threads, 12 loader waves", "the same, 120 KB of LDS", "the same, 34 workgroups at once"};
        std::printf("%-48s %llu ticks from first load to last addition = %.2f per addition; %s\n", names[v], tk, (double)tk / N, got == ref ? "== sequential sum" : "DIFFERS");
    }
    return 0;
}
